"""Sliding-window stitching (m3t/stitch.py) against the reference's own validation_end / test_end
(golden: tests/golden/stitch.npz).  CPU only."""
import numpy as np
import torch

from conftest import load_golden
from m3t import stitch


def _outputs(g, mode):
    n = int(g["%s.n_items" % mode])
    items = []
    for i in range(n):
        it = {"name": str(g["%s.in.%d.name" % (mode, i)]), "start": int(g["%s.in.%d.start" % (mode, i)])}
        for k in ("v_pred", "a_pred", "v_gt", "a_gt"):
            key = "%s.in.%d.%s" % (mode, i, k)
            if key in g:
                it[k] = torch.from_numpy(g[key])
        items.append(it)
    outputs = []
    for i in range(0, n, 4):
        chunk = items[i:i + 4]
        out = {"vid_names": [c["name"] for c in chunk], "start_frames": torch.tensor([c["start"] for c in chunk])}
        for k in ("v_pred", "a_pred", "v_gt", "a_gt"):
            if k in chunk[0]:
                out[k] = [c[k] for c in chunk]
        outputs.append(out)
    return outputs


def _check(got, g, mode, key):
    ref = {k.split(".")[-1]: v for k, v in g.items() if k.startswith("%s.out.%s." % (mode, key))}
    assert sorted(got) == sorted(ref)
    for vid in ref:
        np.testing.assert_allclose(got[vid].numpy(), ref[vid], rtol=0, atol=1e-7, err_msg="%s %s %s" % (mode, key, vid))


def test_test_end_stitching():
    g = load_golden("stitch")
    pv, pa = stitch.stitch_test(_outputs(g, "test"), int(g["window"]))
    _check(pv, g, "test", "valence_pred")
    _check(pa, g, "test", "arousal_pred")


def test_validation_end_concat_and_overlap():
    g = load_golden("stitch")
    for mode, tov in (("val_cat", False), ("val_overlap", True)):
        outs = _outputs(g, mode)
        gv, ga, pv, pa = stitch.stitch_val(outs, int(g["window"]), tov)
        _check(gv, g, mode, "valence_gt"); _check(ga, g, mode, "arousal_gt")
        _check(pv, g, mode, "valence_pred"); _check(pa, g, mode, "arousal_pred")
        m = stitch.val_metrics(outs)
        for k in ("val_ccc_v", "val_ccc_a", "val_mse_v", "val_mse_a", "val_loss"):
            assert abs(float(m[k]) - float(g["%s.metric.%s" % (mode, k)])) < 1e-6, (mode, k)


def test_final_half_window_is_halved_quirk():
    """frames covered once at the tail are still divided by 2 (reference models/model.py:366)."""
    out = [{"vid_names": ["v"], "start_frames": torch.tensor([0]), "v_pred": [torch.ones(8)], "a_pred": [torch.ones(8)]}]
    pv, _ = stitch.stitch_test(out, 8)
    assert torch.equal(pv["v"], torch.tensor([1., 1., 1., 1., .5, .5, .5, .5]))
