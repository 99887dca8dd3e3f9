"""Sliding-window stitching of per-window predictions into per-video tracks -- the host
index arithmetic of AffWild2VA.validation_end / test_end (reference models/model.py:248-303,339-373):
windows are grouped by video, sorted by start frame, overlap-ADDED, and everything from frame `window // 2` on is
halved (stride = window/2 at test time; the last half window, covered once, is halved too -- a quirk kept on
purpose).  Pure CPU torch ops: this is evaluation glue after the hot path, not part of it."""
import torch

from models.utils import concordance_cc2, mse


def _group(outputs, keys):
    by_video = {}
    for out in outputs:
        cols = [out[k] for k in keys]
        for vid, start, *vals in zip(out["vid_names"], out["start_frames"], *cols):
            by_video.setdefault(vid, []).append((int(start), *vals))
    return {k: sorted(v, key=lambda t: t[0]) for k, v in by_video.items()}


def _overlap_add(segments, idx, window):
    nframes = segments[-1][0] + len(segments[-1][1])
    track = torch.zeros(nframes)
    for seg in segments:
        start, vals = seg[0], seg[idx]
        track[start:start + len(vals)] += vals
    track[window // 2:] /= 2.0
    return track


def stitch_test(outputs, window):
    """outputs: list of test_step dicts {'v_pred','a_pred','vid_names','start_frames'} -> (pred_v, pred_a) dicts."""
    groups = _group(outputs, ("v_pred", "a_pred"))
    pred_v = {k: _overlap_add(seg, 1, window) for k, seg in groups.items()}
    pred_a = {k: _overlap_add(seg, 2, window) for k, seg in groups.items()}
    return pred_v, pred_a


def stitch_val(outputs, window, test_on_val):
    """outputs: list of validation_step dicts -> (gt_v, gt_a, pred_v, pred_a) per-video dicts."""
    groups = _group(outputs, ("v_gt", "a_gt", "v_pred", "a_pred"))
    res = [{}, {}, {}, {}]
    for k, seg in groups.items():
        for i in range(4):
            res[i][k] = _overlap_add(seg, i + 1, window) if test_on_val else torch.cat([s[i + 1] for s in seg])
    return tuple(res)


def val_metrics(outputs):
    """Global metrics of validation_end (reference models/model.py:249-259): CCC/MSE over all valid frames."""
    cat = lambda key: torch.cat([torch.cat(x[key]) for x in outputs])
    v_gt, a_gt, v_pred, a_pred = cat("v_gt"), cat("a_gt"), cat("v_pred"), cat("a_pred")
    valid = (v_gt.abs() <= 1) & (a_gt.abs() <= 1)
    ccc_v = concordance_cc2(v_gt[valid], v_pred[valid])
    ccc_a = concordance_cc2(a_gt[valid], a_pred[valid])
    return {"val_ccc_v": ccc_v, "val_ccc_a": ccc_a, "val_mse_v": mse(v_pred[valid], v_gt[valid]),
            "val_mse_a": mse(a_pred[valid], a_gt[valid]), "val_loss": 1 - 0.5 * (ccc_v + ccc_a)}
