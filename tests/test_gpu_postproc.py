"""f-4 post-processing on the GPU (m3t.postproc over csrc/postproc.hip) against the reference's own
smooth_predictions / concordance_cc2_np / get_smoothed_ccc.py / create_submission.py outputs (golden postproc.npz)
and against the numpy oracle on fresh data."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import m3t_oracle as O

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = float(np.abs(a.astype(np.float64) - np.asarray(b, np.float64)).max()) if a.size else 0.0
    assert a.shape == np.asarray(b).shape and err <= tol, "%s: max abs err %.3e" % (what, err)


def test_smoothing_matches_reference():
    from m3t import postproc
    g = load_golden("postproc")
    for v in [str(n) for n in g["names"]]:
        for k in ("valence", "arousal"):
            p = g["pred.%s.%s" % (k, v)]
            got = postproc.smooth_predictions(torch.from_numpy(p), 35, mode="wiener")
            assert isinstance(got, torch.Tensor) and got.dtype == torch.float64      # as np.apply_along_axis on a tensor
            _close(got, g["wiener35.%s.%s" % (k, v)], 1e-9, "wiener35 " + v)
            got = postproc.smooth_predictions(p)
            assert isinstance(got, np.ndarray) and got.dtype == np.float64
            _close(got, g["wiener13.%s.%s" % (k, v)], 1e-9, "wiener13 " + v)
            got = postproc.smooth_predictions(p, 13, mode="median")
            assert got.dtype == np.float32
            _close(got, g["median13.%s.%s" % (k, v)], 0.0, "median13 " + v)


def test_smoothed_ccc_report_matches_reference():
    from m3t import postproc
    g = load_golden("postproc")
    names = [str(n) for n in g["names"]]
    preds = {"valence_pred": {}, "arousal_pred": {}, "valence_gt": {}, "arousal_gt": {}}
    for v in names:
        for k in ("valence", "arousal"):
            preds[k + "_pred"][v] = torch.from_numpy(g["pred.%s.%s" % (k, v)])
            preds[k + "_gt"][v] = torch.from_numpy(g["gt.%s.%s" % (k, v)])
    printed = []
    rep = postproc.smoothed_ccc_report(preds, out=printed.append)
    for v in names:
        assert abs(rep["ccc_v"][v] - float(g["ccc.valence." + v])) < 1e-6, v
        assert abs(rep["ccc_a"][v] - float(g["ccc.arousal." + v])) < 1e-6, v
        assert round(rep["ccc_v"][v], 3) == round(float(g["ccc.valence." + v]), 3)
    assert abs(rep["ccc_v_all"] - float(g["ccc_all.valence"])) < 1e-6
    assert abs(rep["ccc_a_all"] - float(g["ccc_all.arousal"])) < 1e-6
    assert printed[2] == "Lowest ccc-v:" and len(printed) == 2 + 4 * (1 + len(names))


def test_run_ensemble_writes_the_reference_submission(tmp_path):
    from m3t import postproc
    g = load_golden("postproc")
    names = [str(n) for n in g["names"]]
    for tag, key in (("m1.pt", "pred"), ("m2.pt", "pred2")):
        torch.save({"valence_pred": {v: torch.from_numpy(g["%s.valence.%s" % (key, v)]) for v in names},
                    "arousal_pred": {v: torch.from_numpy(g["%s.arousal.%s" % (key, v)]) for v in names}}, str(tmp_path / tag))
    (tmp_path / "videos.txt").write_text("\n".join(names) + "\n")
    (tmp_path / "scores.txt").write_text("%s\n%s\n" % (tmp_path / "m1.pt", tmp_path / "m2.pt"))
    out = postproc.run_ensemble(str(tmp_path / "videos.txt"), str(tmp_path / "scores.txt"), out_dir=str(tmp_path / "VA-Track"))
    for v in names:
        assert open(os.path.join(out, v + ".txt")).read() == str(g["submission." + v]), v


@pytest.mark.parametrize("n,window", [(1, 13), (5, 35), (2000, 35), (777, 129)])
def test_smoothing_vs_oracle_sizes(n, window):
    from m3t import postproc
    rs = np.random.RandomState(n + window)
    x = np.tanh(np.cumsum(rs.standard_normal(n) * 0.1)).astype(np.float32)
    _close(postproc.smooth_predictions(x, window, "wiener"), O.smooth_predictions(x, window, "wiener"), 1e-9, "wiener")
    _close(postproc.smooth_predictions(x, window, "median"), O.smooth_predictions(x, window, "median"), 0.0, "median")
    y = rs.uniform(-1, 1, n).astype(np.float32)
    if n > 1:
        assert abs(postproc.concordance_cc2_np(x.astype(np.float64), y) - O.concordance_cc2_np(x.astype(np.float64), y)) < 1e-6
