"""CPU ORACLE for the M3T hot path -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Nothing under m3f.pytorch_amd/ imports it: the product path is the HIP
library (m3f.pytorch_amd/csrc, C-ABI in include/m3t_hip.h) and fails loudly without it.

What this is: an explicit-math numpy restatement (forward AND hand-derived backward)
of the reference's forward/backward hot path.  Every function cites the reference
file:line it restates (paths relative to the reference repo root).

Where the arithmetic really lives: the reference is pure Python on third-party
`torch` (nn.GRU, weight_norm(nn.Conv1d), nn.Linear, F.softmax, nn.BatchNorm2d ...),
which is NOT under the reference tree and is not even pinned in its requirements.txt
(requirements.txt:1-4).  The cell equations below are therefore the published torch
definitions (torch.nn.GRU docs: gate order [r; z; n], n = tanh(W_in x + b_in +
r * (W_hn h + b_hn))), anchored on the reference's own call sites.

Pinning: the reference ships no tests/golden vectors.  The oracle is
pinned against outputs of the reference itself, imported in the build container
under torch 2.10.0 (CPU): tests/golden/gen_golden.py generated tests/golden/*.npz
(inputs, weights, outputs, input- and parameter-gradients), and
tests/test_oracle_golden.py checks every function here against them.

All functions take/return numpy arrays; parameter dicts use the reference's
state_dict key names.  dtype defaults to float64 (a tighter checker than fp32).
"""
import numpy as np

F64 = np.float64


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------- mixed precision (config C2)
# BASELINE.json config C2 asks for "bf16 activations/weights, fp32 accumulate, fp32 master weights".  The reference has
# no such mode (it is fp32 only), so there is nothing to pin against: the build DEFINES it as "every operand of a
# matmul / conv / recurrent product is rounded to bf16 (nearest even); everything else stays fp32" and this switch
# makes the oracle emulate exactly that, so the HIP path can be checked against it at fp32-accumulation tolerance.
_BF16_OPERANDS = [False]


class matmul_precision:
    def __init__(self, mode):
        assert mode in ("fp32", "bf16")
        self.on = mode == "bf16"

    def __enter__(self):
        self.prev = _BF16_OPERANDS[0]
        _BF16_OPERANDS[0] = self.on

    def __exit__(self, *exc):
        _BF16_OPERANDS[0] = self.prev
        return False


def bf16_round(a):
    """round to nearest-even bf16 (through fp32), returned in the input dtype"""
    a = np.asarray(a)
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(a.dtype).reshape(a.shape)


def _R(a):
    return bf16_round(a) if _BF16_OPERANDS[0] else a


# --------------------------------------------------------------------------- linear
def linear_fwd(x, w, b=None):
    """nn.Linear: y = x W^T + b (used at models/rnn.py:22-55, models/model.py:88)."""
    y = _R(x) @ _R(w).T
    if b is not None:
        y = y + b
    return y


def linear_bwd(dy, x, w, has_bias=True):
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    dx = (_R(dy2) @ _R(w)).reshape(x.shape)
    dw = _R(dy2).T @ _R(x2)
    db = dy2.sum(0) if has_bias else None
    return dx, dw, db


# --------------------------------------------------------------------------- BiGRU
def _gru_dir_fwd(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one layer of nn.GRU(batch_first=True) (models/rnn.py:17,75).
    x [B,T,I] -> out [B,T,H]; h0 = 0; the reverse direction scans t = T-1 .. 0."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = _R(x) @ _R(w_ih).T + b_ih               # [B,T,3H], gate order r,z,n
    out = np.zeros((B, T, H), x.dtype)
    r_s = np.zeros((B, T, H), x.dtype)
    z_s = np.zeros_like(r_s)
    n_s = np.zeros_like(r_s)
    hn_s = np.zeros_like(r_s)                    # W_hn h + b_hn (needed by backward)
    h = np.zeros((B, H), x.dtype)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gh = _R(h) @ _R(w_hh).T + b_hh
        r = _sig(gi[:, t, :H] + gh[:, :H])
        z = _sig(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        out[:, t] = h
        r_s[:, t], z_s[:, t], n_s[:, t], hn_s[:, t] = r, z, n, gh[:, 2 * H:]
    return out, h, (x, out, r_s, z_s, n_s, hn_s, reverse)


def _gru_dir_bwd(dout, dh_last, cache, w_ih, w_hh):
    """BPTT of _gru_dir_fwd (autograd of nn.GRU at models/rnn.py:75)."""
    x, out, r_s, z_s, n_s, hn_s, reverse = cache
    B, T, H = out.shape
    dx = np.zeros_like(x)
    dw_ih = np.zeros_like(w_ih)
    dw_hh = np.zeros_like(w_hh)
    db_ih = np.zeros(3 * H, x.dtype)
    db_hh = np.zeros(3 * H, x.dtype)
    dh = np.zeros((B, H), x.dtype) if dh_last is None else dh_last.copy()
    order = list(range(T - 1, -1, -1) if reverse else range(T))
    for idx in range(T - 1, -1, -1):
        t = order[idx]
        h_prev = out[:, order[idx - 1]] if idx > 0 else np.zeros((B, H), x.dtype)
        r, z, n, hn = r_s[:, t], z_s[:, t], n_s[:, t], hn_s[:, t]
        dht = dout[:, t] + dh
        dn_pre = dht * (1.0 - z) * (1.0 - n * n)
        dz_pre = dht * (h_prev - n) * z * (1.0 - z)
        dr_pre = dn_pre * hn * r * (1.0 - r)
        dgi = np.concatenate([dr_pre, dz_pre, dn_pre], 1)
        dgh = np.concatenate([dr_pre, dz_pre, dn_pre * r], 1)
        dh = dht * z + _R(dgh) @ _R(w_hh)
        dx[:, t] = _R(dgi) @ _R(w_ih)
        dw_ih += _R(dgi).T @ _R(x[:, t])
        dw_hh += _R(dgh).T @ _R(h_prev)
        db_ih += dgi.sum(0)
        db_hh += dgh.sum(0)
    return dx, dw_ih, dw_hh, db_ih, db_hh


def _sfx(layer, d):
    return "l%d%s" % (layer, "_reverse" if d else "")


def bigru_fwd(x, p, num_layers, prefix="gru."):
    """Stacked bidirectional GRU, nn.GRU(I,H,L,batch_first=True,bidirectional=True)
    (models/rnn.py:17).  Layer l>0 consumes concat[fwd,bwd] of layer l-1; no
    inter-layer dropout.  Returns out [B,T,2H], h_n [2L,B,H], caches."""
    caches, hs, inp = [], [], x
    for l in range(num_layers):
        outs = []
        for d in (0, 1):
            s = _sfx(l, d)
            o, h, c = _gru_dir_fwd(inp, p[prefix + "weight_ih_" + s], p[prefix + "weight_hh_" + s],
                                   p[prefix + "bias_ih_" + s], p[prefix + "bias_hh_" + s], bool(d))
            outs.append(o); hs.append(h); caches.append(c)
        inp = np.concatenate(outs, -1)
    return inp, np.stack(hs, 0), caches


def bigru_bwd(dout, caches, p, num_layers, prefix="gru.", dh_n=None):
    grads = {}
    H = caches[0][1].shape[-1]
    d_in = dout
    for l in range(num_layers - 1, -1, -1):
        dx_sum = None
        for d in (0, 1):
            s = _sfx(l, d)
            dhl = None if dh_n is None else dh_n[2 * l + d]
            dx, dwi, dwh, dbi, dbh = _gru_dir_bwd(d_in[..., d * H:(d + 1) * H], dhl, caches[2 * l + d],
                                                  p[prefix + "weight_ih_" + s], p[prefix + "weight_hh_" + s])
            grads[prefix + "weight_ih_" + s] = dwi
            grads[prefix + "weight_hh_" + s] = dwh
            grads[prefix + "bias_ih_" + s] = dbi
            grads[prefix + "bias_hh_" + s] = dbh
            dx_sum = dx if dx_sum is None else dx_sum + dx
        d_in = dx_sum
    return d_in, grads


def _fc_names(num_fcs, dropout):
    """Key layout of GRU.fc (models/rnn.py:20-55): nFC=1 'fc'; nFC=2 fc.0, fc.2
    (fc.0, fc.3 with dropout); nFC=3 fc.0, fc.2, fc.4 (fc.0, fc.3, fc.6 with dropout)."""
    if num_fcs == 1:
        return ["fc"]
    step = 3 if dropout else 2
    return ["fc.%d" % (i * step) for i in range(num_fcs)]


def gru_module_fwd(x, p, num_layers, num_classes, num_fcs=1, dropout=False):
    """models.rnn.GRU.forward (models/rnn.py:71-81), eval-mode dropout (identity)."""
    out, h_n, caches = bigru_fwd(x, p, num_layers)
    fc_cache = []
    y = out
    if num_classes > 0:
        names = _fc_names(num_fcs, dropout)
        for i, nm in enumerate(names):
            a = linear_fwd(y, p[nm + ".weight"], p[nm + ".bias"])
            last = i == len(names) - 1
            fc_cache.append((nm, y, a))
            y = a if last else np.maximum(a, 0.0)
    return y, h_n, (caches, fc_cache)


def gru_module_bwd(dy, cache, p, num_layers, dh_n=None):
    caches, fc_cache = cache
    grads = {}
    d = dy
    for i in range(len(fc_cache) - 1, -1, -1):
        nm, xin, a = fc_cache[i]
        if i != len(fc_cache) - 1:
            d = d * (a > 0)
        d, dw, db = linear_bwd(d, xin, p[nm + ".weight"])
        grads[nm + ".weight"], grads[nm + ".bias"] = dw, db
    dx, g = bigru_bwd(d, caches, p, num_layers, dh_n=dh_n)
    grads.update(g)
    return dx, grads


# --------------------------------------------------------------------------- TCN
def weight_norm_fwd(v, g):
    """torch.nn.utils.weight_norm(dim=0) as applied at models/tcn.py:19-20,25-26:
    w = g * v / ||v||, norm over (C_in,k) per output channel; g is [C_out,1,1]."""
    nrm = np.sqrt((v * v).sum(axis=(1, 2), keepdims=True))
    return g * v / nrm, nrm


def weight_norm_bwd(dw, v, g, nrm):
    dot = (dw * v).sum(axis=(1, 2), keepdims=True)
    dg = dot / nrm
    dv = g / nrm * (dw - v * dot / (nrm * nrm))
    return dv, dg


def causal_conv1d_fwd(x, w, b, dilation):
    """Conv1d(padding=(k-1)d, dilation=d) followed by Chomp1d((k-1)d)
    (models/tcn.py:7-13,19-21): y[b,co,t] = b[co] + sum_{ci,j} w[co,ci,j] x[b,ci,t-(k-1-j)d]
    with x[.,.,<0] = 0.  x [B,C_in,T] channel-first."""
    B, Ci, T = x.shape
    Co, _, K = w.shape
    y = np.zeros((B, Co, T), x.dtype) + (0 if b is None else b[None, :, None])
    for j in range(K):
        s = (K - 1 - j) * dilation
        if s >= T:
            continue
        y[:, :, s:] += np.einsum("oc,bct->bot", _R(w[:, :, j]), _R(x[:, :, :T - s]))
    return y


def causal_conv1d_bwd(dy, x, w, dilation):
    B, Ci, T = x.shape
    Co, _, K = w.shape
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for j in range(K):
        s = (K - 1 - j) * dilation
        if s >= T:
            continue
        dx[:, :, :T - s] += np.einsum("oc,bot->bct", _R(w[:, :, j]), _R(dy[:, :, s:]))
        dw[:, :, j] = np.einsum("bot,bct->oc", _R(dy[:, :, s:]), _R(x[:, :, :T - s]))
    db = dy.sum(axis=(0, 2))
    return dx, dw, db


def temporal_block_fwd(x, p, pre, dilation, masks=None):
    """models.tcn.TemporalBlock.forward (models/tcn.py:43-46): relu(net(x) + res).
    `masks` = (m1, m2) are optional pre-scaled dropout masks (train mode,
    models/tcn.py:23,29); None = eval mode."""
    w1, n1 = weight_norm_fwd(p[pre + "conv1.weight_v"], p[pre + "conv1.weight_g"])
    w2, n2 = weight_norm_fwd(p[pre + "conv2.weight_v"], p[pre + "conv2.weight_g"])
    a1 = causal_conv1d_fwd(x, w1, p[pre + "conv1.bias"], dilation)
    h1 = np.maximum(a1, 0)
    if masks is not None:
        h1 = h1 * masks[0]
    a2 = causal_conv1d_fwd(h1, w2, p[pre + "conv2.bias"], dilation)
    h2 = np.maximum(a2, 0)
    if masks is not None:
        h2 = h2 * masks[1]
    has_ds = (pre + "downsample.weight") in p
    res = causal_conv1d_fwd(x, p[pre + "downsample.weight"], p[pre + "downsample.bias"], 1) if has_ds else x
    s = h2 + res
    y = np.maximum(s, 0)
    return y, (x, w1, n1, w2, n2, a1, h1, a2, s, has_ds, masks)


def temporal_block_bwd(dy, cache, p, pre, dilation):
    x, w1, n1, w2, n2, a1, h1, a2, s, has_ds, masks = cache
    g = {}
    ds = dy * (s > 0)
    dh2 = ds if masks is None else ds * masks[1]
    da2 = dh2 * (a2 > 0)
    dh1, dw2, db2 = causal_conv1d_bwd(da2, h1, w2, dilation)
    if masks is not None:
        dh1 = dh1 * masks[0]
    da1 = dh1 * (a1 > 0)
    dx, dw1, db1 = causal_conv1d_bwd(da1, x, w1, dilation)
    g[pre + "conv1.weight_v"], g[pre + "conv1.weight_g"] = weight_norm_bwd(
        dw1, p[pre + "conv1.weight_v"], p[pre + "conv1.weight_g"], n1)
    g[pre + "conv2.weight_v"], g[pre + "conv2.weight_g"] = weight_norm_bwd(
        dw2, p[pre + "conv2.weight_v"], p[pre + "conv2.weight_g"], n2)
    g[pre + "conv1.bias"], g[pre + "conv2.bias"] = db1, db2
    if has_ds:
        dxr, dwd, dbd = causal_conv1d_bwd(ds, x, p[pre + "downsample.weight"], 1)
        g[pre + "downsample.weight"], g[pre + "downsample.bias"] = dwd, dbd
        dx = dx + dxr
    else:
        dx = dx + ds
    return dx, g


def tcn_fwd(x, p, num_levels, masks=None, prefix="network."):
    """models.tcn.TemporalConvNet.forward (models/tcn.py:49-64): block i has
    dilation 2**i (tcn.py:55) and left padding (k-1)*2**i (tcn.py:59)."""
    caches = []
    for i in range(num_levels):
        x, c = temporal_block_fwd(x, p, "%s%d." % (prefix, i), 2 ** i, None if masks is None else masks[i])
        caches.append(c)
    return x, caches


def tcn_bwd(dy, caches, p, prefix="network."):
    grads = {}
    for i in range(len(caches) - 1, -1, -1):
        dy, g = temporal_block_bwd(dy, caches[i], p, "%s%d." % (prefix, i), 2 ** i)
        grads.update(g)
    return dy, grads


# --------------------------------------------------------------------------- tcn_simple
def conv1d_same_fwd(x, w, b, pad):
    """nn.Conv1d(C_in, C_out, k, 1, pad) with 2*pad = k-1 (models/backbone.py:107,215:
    Conv1d(.,.,3,1,1) / Conv1d(.,.,5,1,2)): y[b,co,t] = b[co] + sum_{ci,j} w[co,ci,j] x[b,ci,t+j-pad],
    zero outside [0,T).  x [B,C_in,T] channel-first."""
    B, Ci, T = x.shape
    Co, _, K = w.shape
    y = np.zeros((B, Co, T), x.dtype) + (0 if b is None else b[None, :, None])
    for j in range(K):
        o = j - pad                               # source time = t + o
        lo, hi = max(0, -o), min(T, T - o)        # output times with the source inside the clip
        if hi > lo:
            y[:, :, lo:hi] += np.einsum("oc,bct->bot", _R(w[:, :, j]), _R(x[:, :, lo + o:hi + o]))
    return y


def conv1d_same_bwd(dy, x, w, pad):
    B, Ci, T = x.shape
    Co, _, K = w.shape
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for j in range(K):
        o = j - pad
        lo, hi = max(0, -o), min(T, T - o)
        if hi > lo:
            dx[:, :, lo + o:hi + o] += np.einsum("oc,bot->bct", _R(w[:, :, j]), _R(dy[:, :, lo:hi]))
            dw[:, :, j] = np.einsum("bot,bct->oc", _R(dy[:, :, lo:hi]), _R(x[:, :, lo + o:hi + o]))
    return dx, dw, dy.sum(axis=(0, 2))


def batchnorm1d_fwd(x, gamma, beta, run_mean, run_var, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d on [B,C,T] (models/backbone.py:108,216): per-channel statistics over (B,T);
    training: biased batch variance normalises, running stats take the unbiased one (torch
    semantics).  Returns y, cache, (new_run_mean, new_run_var)."""
    n = x.shape[0] * x.shape[2]
    if training:
        mu = x.mean(axis=(0, 2))
        var = x.var(axis=(0, 2))
        new_rm = (1 - momentum) * run_mean + momentum * mu
        new_rv = (1 - momentum) * run_var + momentum * var * n / max(n - 1, 1)
    else:
        mu, var, new_rm, new_rv = run_mean, run_var, run_mean, run_var
    invstd = 1.0 / np.sqrt(var + eps)
    xh = (x - mu[None, :, None]) * invstd[None, :, None]
    y = xh * gamma[None, :, None] + beta[None, :, None]
    return y, (xh, invstd, gamma, training), (new_rm, new_rv)


def batchnorm1d_bwd(dy, cache):
    xh, invstd, gamma, training = cache
    dbeta = dy.sum(axis=(0, 2))
    dgamma = (dy * xh).sum(axis=(0, 2))
    if training:
        n = dy.shape[0] * dy.shape[2]
        dx = (gamma * invstd)[None, :, None] * (dy - dbeta[None, :, None] / n - xh * dgamma[None, :, None] / n)
    else:
        dx = dy * (gamma * invstd)[None, :, None]
    return dx, dgamma, dbeta


def simple_tcn_fwd(x, p, pad, training, prefix=""):
    """The `tcn_simple` back-end body (models/backbone.py:107-111 k=3 / 214-222 k=5):
    Sequential(Conv1d, BatchNorm1d(512), ReLU, Conv1d, BatchNorm1d(512), ReLU); x [B,C,T].
    Sequential indices: 0 conv, 1 bn, 3 conv, 4 bn.  Returns y, caches, {running-stat name: new value}."""
    caches, stats = [], {}
    for ci, bi in ((0, 1), (3, 4)):
        w, b = p["%s%d.weight" % (prefix, ci)], p["%s%d.bias" % (prefix, ci)]
        a = conv1d_same_fwd(x, w, b, pad)
        z, bc, (rm, rv) = batchnorm1d_fwd(a, p["%s%d.weight" % (prefix, bi)], p["%s%d.bias" % (prefix, bi)],
                                          p["%s%d.running_mean" % (prefix, bi)], p["%s%d.running_var" % (prefix, bi)],
                                          training)
        stats["%s%d.running_mean" % (prefix, bi)], stats["%s%d.running_var" % (prefix, bi)] = rm, rv
        y = np.maximum(z, 0)
        caches.append((x, w, z, bc))
        x = y
    return x, caches, stats


def simple_tcn_bwd(dy, caches, pad, prefix=""):
    g = {}
    for (ci, bi), (x, w, z, bc) in zip(((3, 4), (0, 1)), caches[::-1]):
        dz = dy * (z > 0)
        da, dgam, dbet = batchnorm1d_bwd(dz, bc)
        dy, dw, db = conv1d_same_bwd(da, x, w, pad)
        g["%s%d.weight" % (prefix, bi)], g["%s%d.bias" % (prefix, bi)] = dgam, dbet
        g["%s%d.weight" % (prefix, ci)], g["%s%d.bias" % (prefix, ci)] = dw, db
    return dy, g


# --------------------------------------------------------------------------- AttFusion
def att_fuse_core_fwd(s_v, s_a, x_v, x_a):
    """The reduction of models/att_fusion.py:21-25 given the raw scorer outputs:
    h = softmax([sigmoid(s_v), sigmoid(s_a)]); f = h0 * x_v + h1 * x_a.
    Softmax index 0 is VIDEO."""
    hv, ha = _sig(s_v), _sig(s_a)
    m = np.maximum(hv, ha)
    ev, ea = np.exp(hv - m), np.exp(ha - m)
    w0, w1 = ev / (ev + ea), ea / (ev + ea)
    return w0 * x_v + w1 * x_a, (hv, ha, w0, w1, x_v, x_a)


def att_fuse_core_bwd(df, cache):
    hv, ha, w0, w1, x_v, x_a = cache
    dx_v, dx_a = w0 * df, w1 * df
    dw0 = (df * x_v).sum(-1, keepdims=True)
    dw1 = (df * x_a).sum(-1, keepdims=True)
    dot = w0 * dw0 + w1 * dw1
    dhv, dha = w0 * (dw0 - dot), w1 * (dw1 - dot)
    return dhv * hv * (1 - hv), dha * ha * (1 - ha), dx_v, dx_a


def att_fusion_fwd(x_a, x_v, p):
    """models.att_fusion.AttFusion.forward(x_a, x_v) (models/att_fusion.py:18-27).
    Argument order is (audio, video); proj_v only when the dims differ (:11-13,19-20);
    scorers are GRU(D0,hidden,1,1,1) (:15-16)."""
    use_proj = "proj_v.weight" in p
    xv_in = x_v
    if use_proj:
        x_v = linear_fwd(x_v, p["proj_v.weight"], p["proj_v.bias"])
    pv = {k[len("scorer_v."):]: v for k, v in p.items() if k.startswith("scorer_v.")}
    pa = {k[len("scorer_a."):]: v for k, v in p.items() if k.startswith("scorer_a.")}
    s_v, _, cv = gru_module_fwd(x_v, pv, 1, 1, 1)
    s_a, _, ca = gru_module_fwd(x_a, pa, 1, 1, 1)
    f, cf = att_fuse_core_fwd(s_v, s_a, x_v, x_a)
    return f, (use_proj, xv_in, pv, pa, cv, ca, cf)


def att_fusion_bwd(df, cache, p):
    use_proj, xv_in, pv, pa, cv, ca, cf = cache
    ds_v, ds_a, dx_v, dx_a = att_fuse_core_bwd(df, cf)
    dxv2, gv = gru_module_bwd(ds_v, cv, pv, 1)
    dxa2, ga = gru_module_bwd(ds_a, ca, pa, 1)
    dx_v, dx_a = dx_v + dxv2, dx_a + dxa2
    grads = {"scorer_v." + k: v for k, v in gv.items()}
    grads.update({"scorer_a." + k: v for k, v in ga.items()})
    if use_proj:
        dx_v, dw, db = linear_bwd(dx_v, xv_in, p["proj_v.weight"])
        grads["proj_v.weight"], grads["proj_v.bias"] = dw, db
    return dx_a, dx_v, grads


# --------------------------------------------------------------------------- CBAM
def channel_gate_fwd(x, p, pre="ChannelGate."):
    """models.cbam.ChannelGate.forward (models/cbam.py:51-58): avg & max over H,W,
    SHARED mlp Linear(C,C/r)-ReLU-Linear(C/r,C) (cbam.py:44-49, keys mlp.1 / mlp.3),
    scale = sigmoid(mlp(avg)+mlp(max)); y = x * scale."""
    N, C, H, W = x.shape
    flat = x.reshape(N, C, H * W)
    avg, mx = flat.mean(-1), flat.max(-1)
    arg = flat.argmax(-1)
    w1, b1, w2, b2 = (p[pre + "mlp.1.weight"], p[pre + "mlp.1.bias"],
                      p[pre + "mlp.3.weight"], p[pre + "mlp.3.bias"])
    ha, hm = avg @ w1.T + b1, mx @ w1.T + b1
    att = np.maximum(ha, 0) @ w2.T + b2 + np.maximum(hm, 0) @ w2.T + b2
    s = _sig(att)
    return x * s[:, :, None, None], (x, avg, mx, arg, ha, hm, s)


def channel_gate_bwd(dy, cache, p, pre="ChannelGate."):
    x, avg, mx, arg, ha, hm, s = cache
    N, C, H, W = x.shape
    w1, w2 = p[pre + "mlp.1.weight"], p[pre + "mlp.3.weight"]
    dx = dy * s[:, :, None, None]
    ds = (dy * x).sum(axis=(2, 3))
    datt = ds * s * (1 - s)
    g = {}
    ra, rm = np.maximum(ha, 0), np.maximum(hm, 0)
    g[pre + "mlp.3.weight"] = datt.T @ ra + datt.T @ rm
    g[pre + "mlp.3.bias"] = 2 * datt.sum(0)
    dha = (datt @ w2) * (ha > 0)
    dhm = (datt @ w2) * (hm > 0)
    g[pre + "mlp.1.weight"] = dha.T @ avg + dhm.T @ mx
    g[pre + "mlp.1.bias"] = dha.sum(0) + dhm.sum(0)
    davg, dmx = dha @ w1, dhm @ w1
    dxf = dx.reshape(N, C, H * W)
    dxf += davg[:, :, None] / (H * W)
    n_i, c_i = np.meshgrid(np.arange(N), np.arange(C), indexing="ij")
    dxf[n_i, c_i, arg] += dmx
    return dxf.reshape(N, C, H, W), g


def _conv2d_same(x, w):
    """Conv2d(2,1,5,padding=2,bias=False) (models/cbam.py:84-85 via BasicConv :19)."""
    N, Ci, H, W = x.shape
    K = w.shape[-1]
    pd = K // 2
    xp = np.pad(x, ((0, 0), (0, 0), (pd, pd), (pd, pd)))
    y = np.zeros((N, 1, H, W), x.dtype)
    for c in range(Ci):
        for i in range(K):
            for j in range(K):
                y[:, 0] += w[0, c, i, j] * xp[:, c, i:i + H, j:j + W]
    return y


def _conv2d_same_bwd(dy, x, w):
    N, Ci, H, W = x.shape
    K = w.shape[-1]
    pd = K // 2
    xp = np.pad(x, ((0, 0), (0, 0), (pd, pd), (pd, pd)))
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for c in range(Ci):
        for i in range(K):
            for j in range(K):
                dw[0, c, i, j] = (dy[:, 0] * xp[:, c, i:i + H, j:j + W]).sum()
                dxp[:, c, i:i + H, j:j + W] += w[0, c, i, j] * dy[:, 0]
    return dxp[:, :, pd:pd + H, pd:pd + W], dw


def spatial_gate_fwd(x, p, training, pre="SpatialGate."):
    """models.cbam.SpatialGate.forward (models/cbam.py:87-92): ChannelPool order is
    (max, mean) (cbam.py:67-71); 5x5 conv 2->1 no bias; BatchNorm2d(1, eps 1e-5,
    momentum 0.01) (cbam.py:20); sigmoid; x * scale.  Returns also the updated
    running stats (train mode: biased var for normalisation, unbiased for running)."""
    N, C, H, W = x.shape
    cmax, carg = x.max(1), x.argmax(1)
    cmean = x.mean(1)
    comp = np.stack([cmax, cmean], 1)
    w = p[pre + "spatial.conv.weight"]
    c = _conv2d_same(comp, w)
    gamma, beta = p[pre + "spatial.bn.weight"], p[pre + "spatial.bn.bias"]
    eps, mom = 1e-5, 0.01
    if training:
        mu, var = c.mean(), c.var()
        cnt = c.size
        new_rm = (1 - mom) * p[pre + "spatial.bn.running_mean"] + mom * mu
        new_rv = (1 - mom) * p[pre + "spatial.bn.running_var"] + mom * var * cnt / max(cnt - 1, 1)
    else:
        mu, var = p[pre + "spatial.bn.running_mean"][0], p[pre + "spatial.bn.running_var"][0]
        new_rm, new_rv = p[pre + "spatial.bn.running_mean"], p[pre + "spatial.bn.running_var"]
    inv = 1.0 / np.sqrt(var + eps)
    xh = (c - mu) * inv
    s = _sig(xh * gamma[0] + beta[0])
    y = x * s
    return y, (x, carg, comp, w, xh, inv, s, gamma, training), (np.atleast_1d(new_rm), np.atleast_1d(new_rv))


def spatial_gate_bwd(dy, cache, pre="SpatialGate."):
    x, carg, comp, w, xh, inv, s, gamma, training = cache
    N, C, H, W = x.shape
    g = {}
    dx = dy * s
    ds = (dy * x).sum(1, keepdims=True)
    dbn = ds * s * (1 - s)
    g[pre + "spatial.bn.weight"] = np.atleast_1d((dbn * xh).sum())
    g[pre + "spatial.bn.bias"] = np.atleast_1d(dbn.sum())
    dxh = dbn * gamma[0]
    if training:
        m = dxh.size
        dc = inv * (dxh - dxh.mean() - xh * (dxh * xh).sum() / m)
    else:
        dc = dxh * inv
    dcomp, dw = _conv2d_same_bwd(dc, comp, w)
    g[pre + "spatial.conv.weight"] = dw
    dx = dx + dcomp[:, 1:2] / C
    n_i, h_i, w_i = np.meshgrid(np.arange(N), np.arange(H), np.arange(W), indexing="ij")
    dx[n_i, carg, h_i, w_i] += dcomp[:, 0]
    return dx, g


def cbam_fwd(x, p, training, pre=""):
    """models.cbam.CBAM.forward (models/cbam.py:107-111): ChannelGate then SpatialGate."""
    y1, c1 = channel_gate_fwd(x, p, pre + "ChannelGate.")
    y2, c2, stats = spatial_gate_fwd(y1, p, training, pre + "SpatialGate.")
    return y2, (c1, c2), stats


def cbam_bwd(dy, cache, p, pre=""):
    c1, c2 = cache
    d1, g2 = spatial_gate_bwd(dy, c2, pre + "SpatialGate.")
    dx, g1 = channel_gate_bwd(d1, c1, p, pre + "ChannelGate.")
    g1.update(g2)
    return dx, g1


# --------------------------------------------------------------------------- losses
def concordance_cc2(r1, r2):
    """models.utils.concordance_cc2 (models/utils.py:6-17) on flat vectors:
    ccc = 2*mean((r1-m1)(r2-m2)) / (var(r1)+var(r2)+(m1-m2)^2) where torch's
    Tensor.var is UNBIASED (N-1) while the covariance is BIASED (N)."""
    n = r1.size
    m1, m2 = r1.mean(), r2.mean()
    cov = ((r1 - m1) * (r2 - m2)).mean()
    v1 = ((r1 - m1) ** 2).sum() / (n - 1)
    v2 = ((r2 - m2) ** 2).sum() / (n - 1)
    return 2 * cov / (v1 + v2 + (m1 - m2) ** 2)


def ccc_loss_fwd_bwd(y_hat, y):
    """AffWild2VA.ccc_loss (models/model.py:132-133): 1 - ccc over the WHOLE
    flattened batch.  Returns (loss, dloss/dy_hat)."""
    x, t = y_hat.reshape(-1), y.reshape(-1)
    n = x.size
    mx, mt = x.mean(), t.mean()
    cov = ((x - mx) * (t - mt)).mean()
    vx = ((x - mx) ** 2).sum() / (n - 1)
    vt = ((t - mt) ** 2).sum() / (n - 1)
    den = vx + vt + (mx - mt) ** 2
    ccc = 2 * cov / den
    dccc = 2 * (t - mt) / (n * den) - (2 * cov / den ** 2) * (2 * (x - mx) / (n - 1) + 2 * (mx - mt) / n)
    return 1 - ccc, (-dccc).reshape(y_hat.shape)


def masked_ce_fwd_bwd(logits, labels, mask):
    """AffWild2VA.ce_loss (models/model.py:139-141): per-row cross entropy times mask,
    mean over ALL B*T rows (not the valid count).  Returns (loss, dloss/dlogits)."""
    lg = logits.reshape(-1, logits.shape[-1])
    lb = labels.reshape(-1)
    mk = mask.reshape(-1).astype(lg.dtype)
    m = lg.max(-1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(lg - m).sum(-1))
    rows = np.arange(lg.shape[0])
    ce = lse - lg[rows, lb]
    loss = (ce * mk).mean()
    sm = np.exp(lg - lse[:, None])
    sm[rows, lb] -= 1.0
    return loss, (sm * (mk / lg.shape[0])[:, None]).reshape(logits.shape)


def training_loss_fwd_bwd(y_hat, valence, arousal, class_expr=None, expr_valid=None,
                          loss_lambda=0.5, mtl=True):
    """Loss assembly of AffWild2VA.training_step (models/model.py:146-182), 'ccc'
    / 'ccc_mtl': valence = y_hat[...,7] (mtl) or [...,-2]; arousal = y_hat[...,-1];
    loss = lam*L_v + (1-lam)*L_a (+ 0.8 * masked CE over y_hat[...,:7] when any
    expr label is valid, :173-177).  Returns (loss, parts, dloss/dy_hat)."""
    iv = 7 if mtl else y_hat.shape[-1] - 2
    ia = y_hat.shape[-1] - 1
    lv, gv = ccc_loss_fwd_bwd(y_hat[..., iv], valence)
    la, ga = ccc_loss_fwd_bwd(y_hat[..., ia], arousal)
    loss = loss_lambda * lv + (1 - loss_lambda) * la
    dy = np.zeros_like(y_hat)
    dy[..., iv] += loss_lambda * gv
    dy[..., ia] += (1 - loss_lambda) * ga
    parts = {"loss_v": lv, "loss_a": la}
    if mtl and expr_valid is not None and expr_valid.sum() > 0:
        le, ge = masked_ce_fwd_bwd(y_hat[..., :7], class_expr, expr_valid)
        loss = loss + 0.8 * le
        dy[..., :7] += 0.8 * ge
        parts["loss_expr"] = le
    return loss, parts, dy


# --------------------------------------------------------------------------- C3 graph
def av_feature_graph_fwd(x_a, x_v, p, num_hidden=512):
    """Feature-level restatement of AffWild2VA.forward, audiovisual/attention
    (models/model.py:108-118) with the conv towers replaced by given features
    (SURVEY.md section 8(d) config C3): audio GRU -> [B,T,512]; gru_v, gru_a on the SAME
    visual features (mirrors `se_features` passed twice, model.py:111) -> cat
    [B,T,4*num_hidden]; proj_v; AttFusion; fusion GRU with 2-layer FC head."""
    sub = lambda pre: {k[len(pre):]: v for k, v in p.items() if k.startswith(pre)}
    a, _, ca = gru_module_fwd(x_a, sub("audio."), 2, -1)
    v1, _, cv1 = gru_module_fwd(x_v, sub("visual.gru_v."), 2, -1)
    v2, _, cv2 = gru_module_fwd(x_v, sub("visual.gru_a."), 2, -1)
    vc = np.concatenate([v1, v2], -1)
    vp = linear_fwd(vc, p["proj_v.weight"], p["proj_v.bias"])
    f, cf = att_fusion_fwd(a, vp, sub("att_fuse."))
    nfc = sum(1 for k in p if k.startswith("fusion.fc") and k.endswith("weight"))
    y, _, cy = gru_module_fwd(f, sub("fusion."), 2, p[[k for k in p if k.startswith("fusion.fc")][-1]].shape[0], nfc)
    return y, (ca, cv1, cv2, vc, cf, cy, sub)


def av_feature_graph_bwd(dy, cache, p):
    ca, cv1, cv2, vc, cf, cy, sub = cache
    grads = {}
    H = vc.shape[-1] // 2
    df, g = gru_module_bwd(dy, cy, sub("fusion."), 2)
    grads.update({"fusion." + k: v for k, v in g.items()})
    da, dvp, g = att_fusion_bwd(df, cf, sub("att_fuse."))
    grads.update({"att_fuse." + k: v for k, v in g.items()})
    dvc, dw, db = linear_bwd(dvp, vc, p["proj_v.weight"])
    grads["proj_v.weight"], grads["proj_v.bias"] = dw, db
    dxv1, g = gru_module_bwd(dvc[..., :H], cv1, sub("visual.gru_v."), 2)
    grads.update({"visual.gru_v." + k: v for k, v in g.items()})
    dxv2, g = gru_module_bwd(dvc[..., H:], cv2, sub("visual.gru_a."), 2)
    grads.update({"visual.gru_a." + k: v for k, v in g.items()})
    dxa, g = gru_module_bwd(da, ca, sub("audio."), 2)
    grads.update({"audio." + k: v for k, v in g.items()})
    return dxa, dxv1 + dxv2, grads


# --------------------------------------------------------------------------- post-processing
# The reference calls scipy.signal.wiener / medfilt (third-party, unpinned: requirements.txt has no scipy line;
# models/utils.py:2,29-33).  Restated here from scipy's published algorithm (scipy/signal/_signaltools.py, `wiener`,
# `medfilt`), pinned by tests/golden/postproc.npz, which was produced by the reference's own smooth_predictions
# under scipy 1.15.3.
def _window_sums(x, window):
    """sum of x over the centred odd window, zero padded ('same' correlation with ones)."""
    n, h = x.shape[0], window // 2
    c = np.concatenate([[0.0], np.cumsum(np.concatenate([np.zeros(h), x.astype(F64), np.zeros(h)]))])
    return c[window:window + n] - c[:n]


def wiener1d(x, window):
    """scipy.signal.wiener(x, window) on a 1-D signal: local mean/variance over the zero-padded window, noise = mean
    local variance, out = mean + (1 - noise/var)(x - mean) where var >= noise, else the local mean.  x**2 is taken in
    the INPUT dtype before the fp64 window sum, as scipy does (fp32 predictions -> fp32 squares)."""
    x = np.asarray(x)
    lmean = _window_sums(x, window) / window
    lvar = _window_sums(x ** 2, window) / window - lmean ** 2
    noise = lvar.mean()
    with np.errstate(divide="ignore", invalid="ignore"):
        res = (x - lmean) * (1 - noise / lvar) + lmean
    return np.where(lvar < noise, lmean, res)


def medfilt1d(x, window):
    """scipy.signal.medfilt(x, window): median of the zero-padded centred window."""
    x = np.asarray(x)
    n, h = x.shape[0], window // 2
    xp = np.concatenate([np.zeros(h, x.dtype), x, np.zeros(h, x.dtype)])
    return np.array([np.sort(xp[i:i + window])[h] for i in range(n)], x.dtype)


def smooth_predictions(preds, window=13, mode="wiener"):
    """models/utils.py:29-33 on a 1-D prediction track."""
    return medfilt1d(preds, window) if mode == "median" else wiener1d(preds, window)


def concordance_cc2_np(r1, r2, r1_unbiased=False):
    """models/utils.py:20-22: numpy CCC with BIASED variances (np.var), unlike the torch loss (a-12 quirk).
    Dtypes are left as given (the reference mixes fp64 smoothed predictions with fp32 labels: the label mean and
    variance are fp32 reductions).
    r1_unbiased: get_smoothed_ccc.py:7-20 feeds it predictions loaded from predictions_val.pt -- torch tensors -- and
    np.apply_along_axis hands a torch tensor back, so `r1.var()` there is TORCH's unbiased variance while `r2.var()`
    (numpy labels) stays biased.  The golden vectors pin this mixed form."""
    r1, r2 = np.asarray(r1), np.asarray(r2)
    cov = ((r1 - r1.mean()) * (r2 - r2.mean())).mean()
    return 2 * cov / (r1.var(ddof=1 if r1_unbiased else 0) + r2.var() + (r1.mean() - r2.mean()) ** 2)


# --------------------------------------------------------------------------- audio front-end
# melspec_db: the reference calls librosa (process/extract_melspec.py:13-20), a third-party dependency with no pinned version
# (requirements.txt) that is absent from this image, so no END-TO-END reference output can be generated here: that part of the
# parity stays unpinned.  The pieces are restated from librosa's published algorithm (librosa.feature.melspectrogram /
# filters.mel / power_to_db defaults of 0.10: periodic Hann window centred in n_fft, center=True with zero padding, power 2,
# Slaney filterbank, ref 1.0, amin 1e-10, top_db 80) and pinned one by one (tests/test_oracle_golden.py): the mel scale and
# the filterbank on the known-answer values librosa publishes in its docstrings (hz_to_mel, mel_to_hz, mel_frequencies,
# filters.mel), the framed / windowed DFT on torch.stft (an independent implementation), power_to_db on closed forms.
# load_audio IS pinned end to end (golden audio_stack.npz from the reference's models/dataset.py:83-95).
_MEL_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    """librosa.hz_to_mel (htk=False, the Slaney / Auditory Toolbox scale): linear below 1 kHz (200/3 Hz per mel), log above"""
    f = np.asarray(f, F64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / _MEL_LOGSTEP, f * 3 / 200.0)


def mel_to_hz(m):
    m = np.asarray(m, F64)
    return np.where(m >= 15.0, 1000.0 * np.exp(_MEL_LOGSTEP * (m - 15.0)), m * 200.0 / 3)


def mel_frequencies(n_mels, fmin, fmax):
    """librosa.mel_frequencies: n_mels points evenly spaced on the mel scale"""
    return mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels))


def mel_filterbank(sr, n_fft, n_mels):
    """librosa.filters.mel(sr, n_fft, n_mels) with its defaults (fmin 0, fmax sr/2, Slaney area normalisation): [n_mels, 1 + n_fft/2]"""
    freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_frequencies(n_mels + 2, 0.0, sr / 2.0)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - freqs[None, :]
    fb = np.stack([np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1])) for i in range(n_mels)])
    return fb * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


def stft_power(y, n_fft, hop, win_length, pad_mode="constant"):
    """|STFT|^2 as librosa.stft(center=True) frames it: periodic Hann of win_length centred in n_fft, n_fft/2 padding: [frames, 1 + n_fft/2]"""
    y = np.asarray(y, F64)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win_length) / win_length)
    lp = (n_fft - win_length) // 2
    win = np.concatenate([np.zeros(lp), win, np.zeros(n_fft - win_length - lp)])
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    nf = 1 + (yp.shape[0] - n_fft) // hop
    frames = np.stack([yp[f * hop:f * hop + n_fft] * win for f in range(nf)])
    return np.abs(np.fft.rfft(frames, axis=1)) ** 2


def power_to_db(S, top_db=80.0, amin=1e-10, ref=1.0):
    db = 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, ref))
    return np.maximum(db, db.max() - top_db) if top_db is not None else db


def melspec_db(y, fps=30.0, pad_mode="constant", top_db=80.0, sr=16000, n_fft=512, win_length=400, n_mels=40):
    hop = int(1 / 3 * 1 / fps * 16000)
    power = stft_power(y, n_fft, hop, win_length, pad_mode)
    return power_to_db(power @ mel_filterbank(sr, n_fft, n_mels).T, top_db)


def load_audio(mel_spec, start_idx, w_len):
    """models/dataset.py:83-95 (context width 2: five mel frames per video frame, three mel frames per video frame)."""
    rows = []
    for i in range(w_len):
        w = mel_spec[(start_idx + i) * 3:(start_idx + i) * 3 + 5]
        if len(w) < 5:
            w = np.pad(w, ((0, 5 - len(w)), (0, 0)), "constant")
        rows.append(w.reshape(-1))
    return np.stack(rows)


# ------------------------------------------------------------------ in-kernel dropout masks (test infrastructure)
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on uint32 arrays:
    the published algorithm, restated -- the generator the HIP conv epilogues use for nn.Dropout (reference models/tcn.py:17,23,29;
    include/m3t_hip.h, m3t_conv1d_fwd).  Pinned by the Random123 known-answer vectors in tests/test_oracle_golden.py."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    for _ in range(10):
        p0, p1 = c0.astype(np.uint64) * m0, c2.astype(np.uint64) * m1
        h0, l0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        h1, l1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
        k0 = np.uint32((int(k0) + 0x9E3779B9) & 0xFFFFFFFF)
        k1 = np.uint32((int(k1) + 0xBB67AE85) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def dropout_mask(rows, C, p, seed):
    """the pre-scaled mask [rows, C] of m3t_conv1d_fwd's in-kernel dropout: element (row, col) is kept (x 1/(1-p)) iff word
    (row & 3) of philox4x32_10(counter {col, row >> 2, 0, 0}, key {seed lo, seed hi}) < (1-p) * 2^32"""
    G = (rows + 3) // 4
    g = np.broadcast_to(np.arange(G, dtype=np.uint32)[:, None], (G, C))
    col = np.broadcast_to(np.arange(C, dtype=np.uint32)[None, :], (G, C))
    z = np.zeros((G, C), dtype=np.uint32)
    words = philox4x32_10(col, g, z, z, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    keep = 1.0 - float(p)
    t = keep * 4294967296.0
    thr = 0xFFFFFFFF if t >= 4294967295.0 else int(t)
    m = np.stack([(w < np.uint32(thr)) for w in words], axis=1).reshape(G * 4, C)[:rows]      # row = 4 g + word index
    return m.astype(np.float64) * np.float64(np.float32(1.0 / keep))
