"""Post-processing of stitched predictions: the reference's `models/utils.py:20-33`,
`get_smoothed_ccc.py` and `create_submission.py`, with the smoothing and the CCC reductions on the GPU
(csrc/postproc.hip) -- every track of a call (all videos x valence/arousal) in one launch.

Same names, arguments and file formats as the reference:
  smooth_predictions(preds, window=13, mode='wiener')          models/utils.py:29-33
  concordance_cc2_np(r1, r2)                                   models/utils.py:20-22
  smoothed_ccc_report('predictions_val.pt')                    get_smoothed_ccc.py:6-43 (prints the same lines)
  run_ensemble(eval_list, score_list)                          create_submission.py:14-39 (VA-Track/<video>.txt)
There is no CPU path: the library must be loadable and a GPU present.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .ops import lib, _stream, M3THipError


def _device():
    if not torch.cuda.is_available():
        raise M3THipError("m3t.postproc needs the GPU: the M3T path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def smooth_tracks(tracks, window, mode="wiener"):
    """tracks: list of 1-D float arrays/tensors -> list of fp64 CUDA tensors (scipy.signal.wiener / medfilt per track)."""
    if mode not in ("wiener", "median"):
        raise ValueError("mode must be 'wiener' or 'median'")
    dev = _device()
    ts = [torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t).detach().to(dev, torch.float32).reshape(-1)
          for t in tracks]
    if not ts:
        return []
    lens = [int(t.numel()) for t in ts]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device=dev)
    x = torch.cat(ts) if len(ts) > 1 else ts[0].contiguous()
    y = torch.empty(x.numel(), dtype=torch.float64, device=dev)
    if x.numel():
        rc = lib().m3t_smooth_tracks(C.c_void_p(x.data_ptr()), C.c_void_p(offs.data_ptr()), len(ts), int(window),
                                     1 if mode == "median" else 0, C.c_void_p(y.data_ptr()), _stream())
        _lib.check(rc, "m3t_smooth_tracks")
    return list(torch.split(y, lens))


def smooth_predictions(preds, window=13, mode="wiener"):
    """models/utils.py:29-33 on a 1-D track.  Like the reference (np.apply_along_axis), a numpy input gives a float64
    numpy array and a torch input gives a float64 torch tensor (on the CPU, where the reference's would be)."""
    if mode not in ("wiener", "median"):
        return None                                   # the reference falls through and returns None
    is_torch = isinstance(preds, torch.Tensor)
    if (preds.dim() if is_torch else np.asarray(preds).ndim) != 1:
        raise M3THipError("smooth_predictions: only 1-D prediction tracks are built (what the reference scripts pass)")
    out = smooth_tracks([preds], window, mode)[0].cpu()
    if not is_torch and mode == "median" and np.asarray(preds).dtype == np.float32:
        return out.numpy().astype(np.float32)          # medfilt keeps the input dtype
    return out if is_torch else out.numpy()


def _ccc(p, g, g2, p_unbiased):
    dev = _device()
    p = torch.as_tensor(np.asarray(p) if not isinstance(p, torch.Tensor) else p).detach().to(dev, torch.float64).reshape(-1).contiguous()
    g = torch.as_tensor(np.asarray(g) if not isinstance(g, torch.Tensor) else g).detach().to(dev, torch.float32).reshape(-1).contiguous()
    if g.numel() != p.numel() or p.numel() == 0:
        raise M3THipError("CCC needs two tracks of the same, non-zero length")
    g2p = None
    if g2 is not None:
        g2 = torch.as_tensor(np.asarray(g2) if not isinstance(g2, torch.Tensor) else g2).detach().to(dev, torch.float32).reshape(-1).contiguous()
        g2p = C.c_void_p(g2.data_ptr())
    out = torch.empty(2, dtype=torch.float64, device=dev)
    rc = lib().m3t_ccc_masked(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), g2p, p.numel(), int(p_unbiased),
                              C.c_void_p(out.data_ptr()), _stream())
    _lib.check(rc, "m3t_ccc_masked")
    return float(out[0].item())


def concordance_cc2_np(r1, r2):
    """models/utils.py:20-22 (biased variances; every frame counts -- labels below -1 would have to be dropped by the
    caller, as get_smoothed_ccc.py:19 does)."""
    lo = torch.as_tensor(np.asarray(r2) if not isinstance(r2, torch.Tensor) else r2)
    if bool((lo < -1).any()):
        raise M3THipError("concordance_cc2_np: drop unannotated frames (< -1) first, or use smoothed_ccc_report")
    return _ccc(r1, r2, None, False)


def smoothed_ccc_report(predictions="predictions_val.pt", window=35, mode="wiener", top=10, out=print):
    """get_smoothed_ccc.py:6-43: Wiener-35 smoothing per video, per-video CCC on the annotated frames
    (valence_gt >= -1 and arousal_gt >= -1), CCC over all videos, and the 10 lowest / highest videos per track.
    `predictions` is the dict written by validation_end (or its path).  Per-video CCCs carry the reference's quirk:
    the smoothed track there is a torch tensor, so ITS variance is the unbiased one (labels: biased); the all-video
    CCC goes through np.concatenate and is biased on both sides.  Returns the numbers it prints."""
    x = torch.load(predictions, map_location="cpu") if isinstance(predictions, (str, os.PathLike)) else predictions
    gt_v, gt_a, pred_v, pred_a = x["valence_gt"], x["arousal_gt"], x["valence_pred"], x["arousal_pred"]
    names = list(gt_v.keys())
    sm = smooth_tracks([pred_v[n] for n in names] + [pred_a[n] for n in names], window, mode)
    sv, sa = sm[:len(names)], sm[len(names):]
    ccc_v, ccc_a = {}, {}
    for i, n in enumerate(names):
        ccc_v[n] = _ccc(sv[i], gt_v[n], gt_a[n], True)
        ccc_a[n] = _ccc(sa[i], gt_a[n], gt_v[n], True)
    dev = _device()
    cat = lambda d: torch.cat([torch.as_tensor(d[n]).reshape(-1).to(dev) for n in names])
    all_v = _ccc(torch.cat(sv), cat(gt_v), cat(gt_a), False)
    all_a = _ccc(torch.cat(sa), cat(gt_a), cat(gt_v), False)
    out(all_v)
    out(all_a)
    for title, table, sign in (("Lowest ccc-v:", ccc_v, 1), ("Highest ccc-v:", ccc_v, -1),
                               ("Lowest ccc-a:", ccc_a, 1), ("Highest ccc-a:", ccc_a, -1)):
        out(title)
        for name, val in sorted(table.items(), key=lambda kv: sign * kv[1])[:top]:
            out("%s %s" % (name, val))
    return {"ccc_v": ccc_v, "ccc_a": ccc_a, "ccc_v_all": all_v, "ccc_a_all": all_a}


def run_ensemble(eval_list, score_list, out_dir="VA-Track"):
    """create_submission.py:14-39: average the per-video predictions of several `predictions_test.pt` files, smooth
    (Wiener-13, the smooth_predictions default) and write `<out_dir>/<video>.txt` with a `valence,arousal` header and
    '{:.3f},{:.3f}' rows.  eval_list / score_list: paths or open files of newline-separated names / .pt paths."""
    def lines(f):
        return open(f.name if hasattr(f, "name") else f, "r").read().splitlines()
    os.makedirs(out_dir, exist_ok=True)
    video_names, score_names = lines(eval_list), lines(score_list)
    dev = _device()
    total = {}
    for i, fname in enumerate(score_names):
        scores = torch.load(fname, map_location="cpu")
        for v in video_names:
            pair = (torch.as_tensor(scores["valence_pred"][v]).to(dev, torch.float32),
                    torch.as_tensor(scores["arousal_pred"][v]).to(dev, torch.float32))
            total[v] = pair if i == 0 else (total[v][0] + pair[0], total[v][1] + pair[1])
    nb = len(score_names)
    sm = smooth_tracks([total[v][0] / nb for v in video_names] + [total[v][1] / nb for v in video_names], 13, "wiener")
    for i, v in enumerate(video_names):
        val, aro = sm[i].cpu().numpy(), sm[len(video_names) + i].cpu().numpy()
        with open(os.path.join(out_dir, v + ".txt"), "w") as fp:
            fp.write("valence,arousal\n")
            for a, b in zip(val, aro):
                fp.write("{:.3f},{:.3f}\n".format(a, b))
    return out_dir
