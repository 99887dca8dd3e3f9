#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported read-only from
/root/reference) on CPU under this container's torch.  Runs ONLY in the build
container; the GPU box never sees the reference.  Fixtures are data only: inputs,
weights (or the seed of the frozen recipe that regenerates them), outputs, gradients.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import os
import sys
import types
import argparse

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
REF = os.environ.get("M3T_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

# models/model.py imports cv2 (via models/dataset.py) and pytorch_lightning, both absent here.
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
_pl = types.ModuleType("pytorch_lightning")
_pl.LightningModule = nn.Module
_pl.data_loader = lambda f: f
sys.modules.setdefault("pytorch_lightning", _pl)

import warnings
warnings.filterwarnings("ignore")

from models.rnn import GRU                                   # noqa: E402  (reference)
from models.tcn import TemporalConvNet                        # noqa: E402
from models.att_fusion import AttFusion                       # noqa: E402
from models.cbam import CBAM                                  # noqa: E402
from models.resnet import ResNet, BasicBlock                  # noqa: E402
from models.utils import concordance_cc2                      # noqa: E402
from models.model import AffWild2VA                           # noqa: E402
from models.backbone import VA_3DVGGM_Split, VA_3DResNet, VA_3DVGGM      # noqa: E402

from recipe import fill_module, named_tensors, draw, grad_digest  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def pack_params(mod, prefix="p."):
    return {prefix + k: v for k, v in named_tensors(mod).items()}


def pack_grads(mod, prefix="g."):
    return {prefix + n: p.grad.detach().numpy().copy() for n, p in mod.named_parameters() if p.grad is not None}


def hp(**kw):
    parser = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False))
    ns = parser.parse_args([])
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


# ------------------------------------------------------------------ small, fully stored
def case_gru(name, args, B, T, seed, return_h=False):
    rs = np.random.RandomState(seed)
    m = fill_module(GRU(*args, return_h=return_h), seed + 1).eval()
    x = torch.from_numpy(draw(rs, (B, T, args[0]))).requires_grad_(True)
    out = m(x)
    arrs = {}
    if return_h:
        y, h = out
        ct_h = torch.from_numpy(draw(rs, tuple(h.shape)))
        arrs.update(h=h.detach().numpy(), ct_h=ct_h.numpy())
    else:
        y = out
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    loss = (y * ct).sum() + ((h * ct_h).sum() if return_h else 0)
    loss.backward()
    save(name, x=x.detach().numpy(), y=y.detach().numpy(), ct=ct.numpy(), dx=x.grad.numpy(),
         args=np.array(list(args) + [int(return_h)]), **arrs, **pack_params(m), **pack_grads(m))


def case_tcn(name, num_inputs, channels, k, B, T, seed):
    rs = np.random.RandomState(seed)
    m = fill_module(TemporalConvNet(num_inputs, channels, k), seed + 1).eval()   # eval: dropout off
    x = torch.from_numpy(draw(rs, (B, num_inputs, T))).requires_grad_(True)
    y = m(x)
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    (y * ct).sum().backward()
    sd_keys = np.array(sorted(m.state_dict().keys()))
    save(name, x=x.detach().numpy(), y=y.detach().numpy(), ct=ct.numpy(), dx=x.grad.numpy(),
         args=np.array([num_inputs, k] + list(channels)), state_dict_keys=sd_keys,
         **pack_params(m), **pack_grads(m))


def case_tcn_simple(name, which, in_dim, B, T, seed, training):
    """The `tcn_simple` temporal back-end exactly as the reference builds and calls it:
    which='split' -> VA_3DVGGM_Split.tcn_v (Conv1d k=5 pad=2, models/backbone.py:212-238,285-286),
    which='vggm'  -> VA_3DVGGM.tcn (Conv1d k=3 pad=1, models/backbone.py:106-113,139-141).
    BatchNorm1d(512) is hard-wired, so hiddenDim = 512; only this sub-module is filled and run."""
    from models.backbone import VA_3DVGGM
    rs = np.random.RandomState(seed)
    if which == "split":
        host = VA_3DVGGM_Split(inputDim=in_dim - 512, hiddenDim=512, backend="tcn_simple", split_layer=3,
                               use_mtl=True, nClasses=8)
        m = host.tcn_v
    else:
        host = VA_3DVGGM(inputDim=in_dim, hiddenDim=512, backend="tcn_simple")
        m = host.tcn
    fill_module(m, seed + 1)
    m.train(training)
    x = torch.from_numpy(draw(rs, (B, in_dim, T))).requires_grad_(True)
    h = m[0](x)
    y = m[1](h.transpose(1, 2).contiguous())
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    (y * ct).sum().backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters() if p.grad is not None}
    stats = {"rs." + n: b.detach().numpy().copy() for n, b in m.named_buffers() if b.dtype.is_floating_point}
    save(name, seed=np.array(seed), dims=np.array([B, in_dim, T, int(training)]), x=x.detach().numpy(),
         y=y.detach().numpy(), ct=ct.numpy(), dx=x.grad.numpy(), state_dict_keys=np.array(sorted(m.state_dict().keys())),
         **grads, **stats)


def case_attfusion(name, dims, hidden, B, T, seed):
    rs = np.random.RandomState(seed)
    m = fill_module(AttFusion(dims, hidden), seed + 1).eval()
    xa = torch.from_numpy(draw(rs, (B, T, dims[0]))).requires_grad_(True)
    xv = torch.from_numpy(draw(rs, (B, T, dims[1]))).requires_grad_(True)
    y = m(xa, xv)
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    (y * ct).sum().backward()
    save(name, x_a=xa.detach().numpy(), x_v=xv.detach().numpy(), y=y.detach().numpy(), ct=ct.numpy(),
         dx_a=xa.grad.numpy(), dx_v=xv.grad.numpy(), args=np.array(list(dims) + [hidden]),
         **pack_params(m), **pack_grads(m))


def case_cbam(name, C, N, H, W, seed, training):
    rs = np.random.RandomState(seed)
    m = fill_module(CBAM(C), seed + 1)
    m.train(training)
    before = named_tensors(m)
    x = torch.from_numpy(draw(rs, (N, C, H, W))).requires_grad_(True)
    y = m(x)
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    (y * ct).sum().backward()
    after = named_tensors(m)
    save(name, x=x.detach().numpy(), y=y.detach().numpy(), ct=ct.numpy(), dx=x.grad.numpy(),
         training=np.array(int(training)),
         running_mean_after=after["SpatialGate.spatial.bn.running_mean"],
         running_var_after=after["SpatialGate.spatial.bn.running_var"],
         **{"p." + k: v for k, v in before.items()}, **pack_grads(m))


def case_losses(name, seed):
    rs = np.random.RandomState(seed)
    model = AffWild2VA(hp(modality="audio", loss="ccc_mtl"))
    B, T = 3, 50
    y_hat = torch.from_numpy(draw(rs, (B, T, 9))).requires_grad_(True)
    val = torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
    aro = torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
    expr = torch.from_numpy(rs.randint(0, 7, (B, T)).astype(np.int64))
    valid = torch.from_numpy(rs.uniform(size=(B, T)) < 0.7)
    ccc_v = concordance_cc2(y_hat[..., 7].reshape(-1), val.reshape(-1), "none").squeeze()
    l_v = model.ccc_loss(y_hat[..., 7], val)
    l_a = model.ccc_loss(y_hat[..., -1], aro)
    l_e = model.ce_loss(y_hat[..., :7], expr, valid)
    loss = 0.5 * l_v + 0.5 * l_a + 0.8 * l_e
    loss.backward()
    save(name, y_hat=y_hat.detach().numpy(), valence=val.numpy(), arousal=aro.numpy(), class_expr=expr.numpy(),
         expr_valid=valid.numpy(), ccc_v=ccc_v.detach().numpy(), loss_v=l_v.detach().numpy(),
         loss_a=l_a.detach().numpy(), loss_expr=l_e.detach().numpy(), loss=loss.detach().numpy(),
         dy_hat=y_hat.grad.numpy())


def case_resnet_cbam(name, seed, training):
    """ResNet v1 [1,1,1,1] with CBAM on small maps (models/resnet.py:59-124)."""
    rs = np.random.RandomState(seed)
    m = fill_module(ResNet(BasicBlock, [1, 1, 1, 1], use_cbam=True), seed + 1)
    m.train(training)
    before = named_tensors(m)
    x = torch.from_numpy(draw(rs, (3, 64, 16, 16))).requires_grad_(True)
    y = m(x)
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    (y * ct).sum().backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters() if p.grad is not None}
    save(name, seed=np.array(seed), training=np.array(int(training)), x=x.detach().numpy(), y=y.detach().numpy(),
         ct=ct.numpy(), dx=x.grad.numpy(), param_names=np.array(sorted(before.keys())), **grads)


# ------------------------------------------------------------------ full-size, recipe weights
class RefAVFeatureGraph(nn.Module):
    """Config C3/C4 of SURVEY.md section 8(d): AffWild2VA.forward audiovisual/attention
    (reference models/model.py:108-118) with the conv towers replaced by pre-computed
    features, assembled from the reference's own classes with AffWild2VA's attribute names."""

    def __init__(self, d_a=128, d_v=256, num_hidden=512):
        super().__init__()
        self.audio = GRU(d_a, 256, 2, -1, 2)
        self.visual = nn.Module()
        self.visual.gru_v = GRU(d_v, num_hidden, 2, -1, 2)
        self.visual.gru_a = GRU(d_v, num_hidden, 2, -1, 2)
        self.proj_v = nn.Linear(num_hidden * 4, 512)
        self.att_fuse = AttFusion([512, 512], 128)
        self.fusion = GRU(512, num_hidden, 2, 9, 2)

    def forward(self, x_a, x_v):
        a = self.audio(x_a)
        v = torch.cat((self.visual.gru_v(x_v), self.visual.gru_a(x_v)), dim=-1)
        v = self.proj_v(v)
        return self.fusion(self.att_fuse(a, v))


def _mtl_loss(model, y_hat, val, aro, expr, valid):
    l_v = model.ccc_loss(y_hat[..., 7], val)
    l_a = model.ccc_loss(y_hat[..., -1], aro)
    loss = 0.5 * l_v + 0.5 * l_a
    loss = loss + 0.8 * model.ce_loss(y_hat[..., :7], expr, valid)
    return loss, l_v, l_a


def draw_audio(rs, shape, audio):
    """audio='normal': N(0,1) like every other input; audio='db': U(-80, 0) -- the scale of what the reference really feeds its
    audio branch, un-normalised librosa power_to_db log-Mel values (reference process/extract_melspec.py:13-20,
    models/dataset.py:83-95; SURVEY 8(d) "secondary run")"""
    if audio == "db":
        return rs.uniform(-80.0, 0.0, shape).astype(np.float32)
    return draw(rs, shape)


def case_c3(name, B, T, seed, d_a=128, d_v=256, nh=512, with_norm=False, audio="normal", full_grads=None):
    """full_grads = (file name, [parameter names]): the FULL gradient tensors of those parameters (before the clip) go into a fixture of
    their own (full-size backward pinned beyond norms and leading values for one parameter group)"""
    rs = np.random.RandomState(seed)
    m = fill_module(RefAVFeatureGraph(d_a, d_v, nh), seed + 1).eval()
    lossmod = AffWild2VA(hp(modality="audio", loss="ccc_mtl"))
    xa = torch.from_numpy(draw_audio(rs, (B, T, d_a), audio)).requires_grad_(True)
    xv = torch.from_numpy(draw(rs, (B, T, d_v))).requires_grad_(True)
    val = torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
    aro = torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
    expr = torch.from_numpy(rs.randint(0, 7, (B, T)).astype(np.int64))
    valid = torch.from_numpy(rs.uniform(size=(B, T)) < 0.7)
    # the ONE non-smooth operation of the graph is the ReLU of the fusion head (models/rnn.py:32-36): the reference's own
    # pre-activations that lie close to zero (|pre| < 1e-5: (frame row, unit) and value).  Another fp32 evaluation may draw a
    # different sign there; a test that knows WHERE can tell such a flip from a kernel error.
    near = {}
    hook = m.fusion.fc[0].register_forward_hook(lambda mod, i, o: near.__setitem__("pre", o.detach().reshape(-1, o.shape[-1]).clone()))
    y = m(xa, xv)
    hook.remove()
    idx = (near["pre"].abs() < 1e-5).nonzero()
    relu_near = {"relu_near_idx": idx.numpy().astype(np.int32), "relu_near_pre": near["pre"][idx[:, 0], idx[:, 1]].numpy()}
    loss, l_v, l_a = _mtl_loss(lossmod, y, val, aro, expr, valid)
    loss.backward()
    ccc_v = concordance_cc2(y[..., 7].reshape(-1), val.reshape(-1), "none").squeeze()
    ccc_a = concordance_cc2(y[..., -1].reshape(-1), aro.reshape(-1), "none").squeeze()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters()}
    if full_grads is not None:
        pd = dict(m.named_parameters())
        save(full_grads[0], seed=np.array(seed), dims=np.array([B, T, d_a, d_v, nh]),
             **{"g." + n: pd[n].grad.numpy().copy() for n in full_grads[1]})
        if full_grads[2]:
            return
    if with_norm:
        # what Lightning's gradient_clip_val=1.0 computes (reference train.py:35): the global L2 norm of all gradients
        gn = torch.nn.utils.clip_grad_norm_(list(m.parameters()), 1.0)
        grads["grad_norm"] = np.array(float(gn))
        grads.update({"gdc." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters()})      # after the clip
    save(name, seed=np.array(seed), dims=np.array([B, T, d_a, d_v, nh]), y=y.detach().numpy(),
         loss=loss.detach().numpy(), loss_v=l_v.detach().numpy(), loss_a=l_a.detach().numpy(),
         ccc_v=ccc_v.detach().numpy(), ccc_a=ccc_a.detach().numpy(),
         dx_a=grad_digest(xa.grad.numpy()), dx_v=grad_digest(xv.grad.numpy()),
         dx_a_full=xa.grad.numpy()[:, ::(60 if with_norm else 25)], dx_v_full=xv.grad.numpy()[:, ::(60 if with_norm else 25)],
         dx_t_stride=np.array(60 if with_norm else 25), **relu_near, **grads)


class RefTcnHead(nn.Module):
    """Config C1: TemporalConvNet(128,[512,512],3) + Linear(512,2) composed exactly as
    VA_3DVGGM's tcn back-end (reference models/backbone.py:107-111,139-141)."""

    def __init__(self, d_in=128, hidden=512, levels=2):
        super().__init__()
        self.tcn = nn.ModuleList([TemporalConvNet(d_in, [hidden] * levels, 3), nn.Linear(hidden, 2)])

    def forward(self, x):
        x = self.tcn[0](x).transpose(1, 2).contiguous()
        return self.tcn[1](x)


class RefTcnGru(nn.Module):
    """Config C2: TemporalConvNet(256,[512,512],3) -> transpose -> GRU(512,512,2,2,2)."""

    def __init__(self, d_in=256, hidden=512):
        super().__init__()
        self.tcn = TemporalConvNet(d_in, [hidden, hidden], 3)
        self.gru = GRU(hidden, hidden, 2, 2, 2)

    def forward(self, x):
        return self.gru(self.tcn(x).transpose(1, 2))


def case_seq_model(name, ctor, in_shape, seed, ccc_out=True, with_norm=False):
    rs = np.random.RandomState(seed)
    m = fill_module(ctor(), seed + 1).eval()
    lossmod = AffWild2VA(hp(modality="audio", loss="ccc"))
    x = torch.from_numpy(draw(rs, in_shape)).requires_grad_(True)
    y = m(x)
    B, T = y.shape[0], y.shape[1]
    val = torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
    aro = torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
    l_v = lossmod.ccc_loss(y[..., -2], val)
    l_a = lossmod.ccc_loss(y[..., -1], aro)
    loss = 0.5 * l_v + 0.5 * l_a
    loss.backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters()}
    if with_norm:
        gn = torch.nn.utils.clip_grad_norm_(list(m.parameters()), 1.0)
        grads["grad_norm"] = np.array(float(gn))
    save(name, seed=np.array(seed), in_shape=np.array(in_shape), y=y.detach().numpy(), loss=loss.detach().numpy(),
         dx=grad_digest(x.grad.numpy()),
         dx_full=(x.grad.numpy()[:, ::4, ::30] if with_norm else x.grad.numpy()[:, :, ::10]) if x.dim() == 3 else x.grad.numpy(),
         **grads)


def case_seq_model_autocast(name, ctor, in_shape, seed):
    """BASELINE configs[1] names a bf16 mode the reference does not have.  The yardstick that exists outside this repo: the
    reference's own classes under PyTorch's standard mixed precision, torch.autocast('cpu', dtype=bfloat16) (forward only;
    same seed / weights / inputs as case_seq_model): outputs and loss, next to the fp32 ones of the fp32 golden."""
    rs = np.random.RandomState(seed)
    m = fill_module(ctor(), seed + 1).eval()
    lossmod = AffWild2VA(hp(modality="audio", loss="ccc"))
    x = torch.from_numpy(draw(rs, in_shape))
    with torch.no_grad():
        y32 = m(x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y16 = m(x)
        y16 = y16.float()
        B, T = y32.shape[0], y32.shape[1]
        val = torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
        aro = torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
        loss = lambda y: 0.5 * lossmod.ccc_loss(y[..., -2], val) + 0.5 * lossmod.ccc_loss(y[..., -1], aro)
    # the BACKWARD yardstick: parameter- and input-gradient digests of the same classes with forward AND backward under
    # autocast (the loss itself in fp32, as torch's AMP recipe prescribes), next to the fp32 golden's digests
    xg = x.clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        yg = m(xg)
    loss(yg.float()).backward()
    grads = {"gd." + n: grad_digest(p.grad.float().numpy()) for n, p in m.named_parameters()}
    save(name, seed=np.array(seed), in_shape=np.array(in_shape), y_autocast=y16.numpy(), loss_autocast=loss(y16).numpy(),
         loss_fp32=loss(y32).numpy(), err_autocast=np.array(float((y16 - y32).abs().max())),
         dx=grad_digest(xg.grad.float().numpy()), **grads)


def case_affwild_audio(name, seed, audio="normal"):
    """Config C1 reference-faithful variant: AffWild2VA(modality='audio', loss='ccc_mtl')
    = GRU(200,256,2,9,2) on [4,100,200]; training_step (reference models/model.py:146-218)."""
    rs = np.random.RandomState(seed)
    m = fill_module(AffWild2VA(hp(modality="audio", loss="ccc_mtl")), seed + 1).eval()
    B, T = 4, 100
    batch = {
        "audio": torch.from_numpy(draw_audio(rs, (B, T, 200), audio)),
        "label_valence": torch.from_numpy(draw(rs, (B, T), "uniform_pm1")),
        "label_arousal": torch.from_numpy(draw(rs, (B, T), "uniform_pm1")),
        "class_expr": torch.from_numpy(rs.randint(0, 7, (B, T)).astype(np.int64)),
        "expr_valid": torch.from_numpy(rs.uniform(size=(B, T)) < 0.7),
    }
    y = m(batch)
    out = m.training_step(batch, 0)
    out["loss"].backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters()}
    save(name, seed=np.array(seed), y=y.detach().numpy(), loss=out["loss"].detach().numpy(),
         loss_v=out["log"]["loss_v"].detach().numpy(), loss_a=out["log"]["loss_a"].detach().numpy(),
         loss_expr=out["log"]["loss_expr"].detach().numpy(), acc_expr=np.array(out["progress_bar"]["acc_expr"]),
         param_names=np.array(sorted(n for n, _ in m.named_parameters())), **grads)


def case_affwild_av(name, seed, B=2, T=4):
    """Config C5 (tiny): full AffWild2VA audiovisual/attention/v2p_split/ccc_mtl on raw
    112x112 frames (reference models/model.py:101-118, models/backbone.py:273-295)."""
    rs = np.random.RandomState(seed)
    m = fill_module(AffWild2VA(hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)),
                    seed + 1).eval()
    batch = {
        "video": torch.from_numpy(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32)),
        "se_features": torch.from_numpy(draw(rs, (B, 512, T))),
        "audio": torch.from_numpy(draw(rs, (B, T, 200))),
        "label_valence": torch.from_numpy(draw(rs, (B, T), "uniform_pm1")),
        "label_arousal": torch.from_numpy(draw(rs, (B, T), "uniform_pm1")),
        "class_expr": torch.from_numpy(rs.randint(0, 7, (B, T)).astype(np.int64)),
        "expr_valid": torch.from_numpy(rs.uniform(size=(B, T)) < 0.7),
    }
    # the conv stem's OUTPUT (both private towers, [B,512,T,1,1]): lets a test feed the reference's own stem features into
    # the temporal part and so separate its error (bar 1e-4) from the MIOpen stem's
    feats = {}
    hooks = [m.visual.v_private.register_forward_hook(lambda mod, i, o: feats.__setitem__("feat_v", o.detach().numpy().copy())),
             m.visual.a_private.register_forward_hook(lambda mod, i, o: feats.__setitem__("feat_a", o.detach().numpy().copy()))]
    y = m(batch)
    for h in hooks:
        h.remove()
    out = m.training_step(batch, 0)
    out["loss"].backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters() if p.grad is not None}
    save(name, seed=np.array(seed), dims=np.array([B, T]), y=y.detach().numpy(), loss=out["loss"].detach().numpy(),
         param_names=np.array(sorted(n for n, _ in m.named_parameters())),
         state_dict_keys=np.array(sorted(m.state_dict().keys())), **feats, **grads)


def case_resnet3d(name, seed, B=2, T=3):
    """VA_3DResNet(resnet_ver='v1', use_cbam=True) visual-only (SURVEY 8(d) C5 alt),
    eval mode, 112x112 input (reference models/backbone.py:314-355)."""
    rs = np.random.RandomState(seed)
    m = fill_module(VA_3DResNet(frameLen=T, resnet_ver="v1", use_cbam=True, nClasses=2, nFCs=2), seed + 1).eval()
    x = torch.from_numpy(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32))
    x = ((x - 127.5) / 127.5).requires_grad_(True)
    y = m(x)
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    (y * ct).sum().backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters() if p.grad is not None}
    save(name, seed=np.array(seed), dims=np.array([B, T]), y=y.detach().numpy(), ct=ct.numpy(),
         dx=grad_digest(x.grad.numpy()), param_names=np.array(sorted(n for n, _ in m.named_parameters())), **grads)


def _bn_state(m):
    """every BatchNorm buffer after the step (running_mean / running_var / num_batches_tracked)"""
    return {"bn." + n: b.detach().numpy().astype(np.float64) for n, b in m.named_buffers()
            if n.split(".")[-1] in ("running_mean", "running_var", "num_batches_tracked")}


def case_affwild_av_train(name, seed, B=2, T=16):
    """Config C5 in TRAIN mode (what bench.py times): the full AffWild2VA audiovisual/attention/v2p_split/ccc_mtl
    training_step on raw frames with BatchNorm3d on BATCH statistics through the five stem groups (reference models/backbone.py:179-271,
    models/model.py:146-218): outputs, loss, every parameter-gradient digest and the BatchNorm buffers after the step.  No dropout is
    active on this path (GRU heads: dropout=False, model.py:86), so the step is deterministic."""
    rs = np.random.RandomState(seed)
    m = fill_module(AffWild2VA(hp(modality="audiovisual", fusion_type="attention", loss="ccc_mtl", window=T)), seed + 1).train()
    batch = {
        "video": torch.from_numpy(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32)),
        "se_features": torch.from_numpy(draw(rs, (B, 512, T))),
        "audio": torch.from_numpy(draw(rs, (B, T, 200))),
        "label_valence": torch.from_numpy(draw(rs, (B, T), "uniform_pm1")),
        "label_arousal": torch.from_numpy(draw(rs, (B, T), "uniform_pm1")),
        "class_expr": torch.from_numpy(rs.randint(0, 7, (B, T)).astype(np.int64)),
        "expr_valid": torch.from_numpy(rs.uniform(size=(B, T)) < 0.7),
    }
    ys = {}
    fwd = m.forward                                # (training_step calls self.forward directly: no forward hook fires)

    def tap(b):
        o = fwd(b)
        ys["y"] = o.detach().numpy().copy()
        return o
    m.forward = tap
    out = m.training_step(batch, 0)                # ONE forward: the BatchNorm buffers are updated once
    del m.forward
    out["loss"].backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters() if p.grad is not None}
    save(name, seed=np.array(seed), dims=np.array([B, T]), y=ys["y"], loss=out["loss"].detach().numpy(),
         param_names=np.array(sorted(n for n, _ in m.named_parameters())), **_bn_state(m), **grads)


def case_resnet3d_train(name, seed, B=2, T=3):
    """VA_3DResNet(resnet_ver='v1', use_cbam=True) in TRAIN mode (bench.py's aux.cbam_resnet3d): BatchNorm3d of the stem, the 20
    BatchNorm2d of the per-frame ResNet-18 and the 8 CBAM gates' BatchNorm2d(1) on batch statistics (reference models/backbone.py:327-355,
    models/resnet.py:40-56, models/cbam.py:84-93): outputs, gradient digests, the input gradient and the buffers after the step."""
    rs = np.random.RandomState(seed)
    m = fill_module(VA_3DResNet(frameLen=T, resnet_ver="v1", use_cbam=True, nClasses=2, nFCs=2), seed + 1).train()
    x = torch.from_numpy(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32))
    x = ((x - 127.5) / 127.5).requires_grad_(True)
    y = m(x)
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    (y * ct).sum().backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters() if p.grad is not None}
    save(name, seed=np.array(seed), dims=np.array([B, T]), y=y.detach().numpy(), ct=ct.numpy(),
         dx=grad_digest(x.grad.numpy()), param_names=np.array(sorted(n for n, _ in m.named_parameters())), **_bn_state(m), **grads)


def case_vggm(name, seed, backend, B=2, T=4, training=False):
    """VA_3DVGGM end to end from raw frames (reference models/backbone.py:62-161, forward :134-145): the unsplit VGG-M stem
    -> TemporalConvNet(512,[512,512],3) + Linear(512,2) (backend 'tcn': the only TemporalConvNet user) or GRU (backend 'gru').
    eval mode by default (TCN dropout off, BatchNorm on running stats)."""
    rs = np.random.RandomState(seed)
    m = fill_module(VA_3DVGGM(frameLen=T, backend=backend, nClasses=2, nFCs=2), seed + 1)
    m = m.train() if training else m.eval()
    x = torch.from_numpy(rs.randint(0, 256, (B, 3, T, 112, 112)).astype(np.float32))
    x = ((x - 127.5) / 127.5).requires_grad_(True)
    y = m(x)
    ct = torch.from_numpy(draw(rs, tuple(y.shape)))
    (y * ct).sum().backward()
    grads = {"gd." + n: grad_digest(p.grad.numpy()) for n, p in m.named_parameters() if p.grad is not None}
    save(name, seed=np.array(seed), dims=np.array([B, T]), y=y.detach().numpy(), ct=ct.numpy(),
         dx=grad_digest(x.grad.numpy()), param_names=np.array(sorted(n for n, _ in m.named_parameters())),
         state_dict_keys=np.array(sorted(m.state_dict().keys())), **grads)


def case_init_digests(name):
    """Initial weights of the reference constructors under torch.manual_seed(12345) (the
    reference's default --seed, train.py:49), as digests: pins the RNG-order of the init recipes."""
    out = {}
    ctors = {"gru": lambda: GRU(24, 16, 2, 3, 2), "tcn": lambda: TemporalConvNet(8, [12, 12], 3),
             "att": lambda: AttFusion([12, 20], 6), "cbam": lambda: CBAM(32)}
    for tag, ctor in ctors.items():
        torch.manual_seed(12345)
        m = ctor()
        for n, p in m.named_parameters():
            out["%s.%s" % (tag, n)] = grad_digest(p.detach().numpy())
    save(name, **out)


def _stitch_outputs(rs, window, with_gt):
    """Synthetic validation/test step outputs: 3 videos, windows at stride window/2, ragged last windows."""
    vids = {"vidA": 37, "vidB": 16, "vidC": 25}
    items = []
    for name, n in vids.items():
        for st in range(0, n, window // 2):
            ln = min(window, n - st)
            it = {"name": name, "start": st, "v_pred": draw(rs, (ln,)), "a_pred": draw(rs, (ln,))}
            if with_gt:
                it["v_gt"] = draw(rs, (ln,), "uniform_pm1")
                it["a_gt"] = draw(rs, (ln,), "uniform_pm1")
            items.append(it)
    order = rs.permutation(len(items))          # batches arrive shuffled across videos
    items = [items[i] for i in order]
    outputs = []
    for i in range(0, len(items), 4):
        chunk = items[i:i + 4]
        out = {"vid_names": [c["name"] for c in chunk], "start_frames": torch.tensor([c["start"] for c in chunk]),
               "v_pred": [torch.from_numpy(c["v_pred"]) for c in chunk], "a_pred": [torch.from_numpy(c["a_pred"]) for c in chunk]}
        if with_gt:
            out["v_gt"] = [torch.from_numpy(c["v_gt"]) for c in chunk]
            out["a_gt"] = [torch.from_numpy(c["a_gt"]) for c in chunk]
        outputs.append(out)
    return outputs, items


def case_stitch(name, seed):
    """validation_end / test_end of the reference (models/model.py:248-373) on synthetic window outputs."""
    import tempfile
    window = 8
    arrs = {"window": np.array(window)}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for mode in ("test", "val_cat", "val_overlap"):
                rs = np.random.RandomState(seed + len(mode))
                with_gt = mode != "test"
                outputs, items = _stitch_outputs(rs, window, with_gt)
                model = AffWild2VA(hp(modality="audio", window=window, test_on_val=(mode == "val_overlap")))
                if mode == "test":
                    model.test_end(outputs)
                    res = torch.load("predictions_test.pt")
                else:
                    ret = model.validation_end(outputs)
                    res = torch.load("predictions_val.pt")
                    for k in ("val_ccc_v", "val_ccc_a", "val_mse_v", "val_mse_a", "val_loss"):
                        arrs["%s.metric.%s" % (mode, k)] = np.array(float(ret["log"][k]))
                for key, per_video in res.items():
                    for vid, t in per_video.items():
                        arrs["%s.out.%s.%s" % (mode, key, vid)] = t.numpy()
                arrs["%s.n_items" % mode] = np.array(len(items))
                for i, it in enumerate(items):
                    arrs["%s.in.%d.name" % (mode, i)] = np.array(it["name"])
                    arrs["%s.in.%d.start" % (mode, i)] = np.array(it["start"])
                    for k in ("v_pred", "a_pred", "v_gt", "a_gt"):
                        if k in it:
                            arrs["%s.in.%d.%s" % (mode, i, k)] = it[k]
        finally:
            os.chdir(cwd)
    save(name, **arrs)


def case_postproc(name, seed):
    """f-4 post-processing: the reference's own smooth_predictions / concordance_cc2_np
    (models/utils.py:20-33) driven as get_smoothed_ccc.py:7-28 and create_submission.py:14-39 drive them."""
    import tempfile
    _mpl = types.ModuleType("matplotlib")
    _mpl.pyplot = types.ModuleType("matplotlib.pyplot")
    sys.modules.setdefault("matplotlib", _mpl)
    sys.modules.setdefault("matplotlib.pyplot", _mpl.pyplot)
    from models.utils import smooth_predictions, concordance_cc2_np      # reference
    import create_submission                                              # reference
    rs = np.random.RandomState(seed)
    vids = {"vidA": 120, "vidB": 40, "vidC": 9, "vidD": 301}            # vidC is shorter than both windows
    arrs = {"names": np.array(list(vids))}
    pred, gt = {"valence": {}, "arousal": {}}, {"valence": {}, "arousal": {}}
    for v, n in vids.items():
        for k in ("valence", "arousal"):
            walk = np.cumsum(rs.standard_normal(n) * 0.05) + 0.3 * rs.standard_normal(n)
            pred[k][v] = torch.from_numpy(np.tanh(walk).astype(np.float32))
            g = rs.uniform(-1, 1, n).astype(np.float32)
            g[rs.uniform(size=n) < 0.1] = -5.0                           # unannotated frames (get_smoothed_ccc.py:19)
            gt[k][v] = torch.from_numpy(g)
            arrs["pred.%s.%s" % (k, v)] = pred[k][v].numpy()
            arrs["gt.%s.%s" % (k, v)] = g
    # smoothing as called by the reference scripts
    for v in vids:
        for k in ("valence", "arousal"):
            arrs["wiener35.%s.%s" % (k, v)] = smooth_predictions(pred[k][v], 35, mode="wiener")
            arrs["wiener13.%s.%s" % (k, v)] = smooth_predictions(pred[k][v].numpy())
            arrs["median13.%s.%s" % (k, v)] = smooth_predictions(pred[k][v].numpy(), 13, mode="median")
    # get_smoothed_ccc.py:13-28
    allp, allg = {"valence": [], "arousal": []}, {"valence": [], "arousal": []}
    for v in vids:
        pv, pa = arrs["wiener35.valence." + v], arrs["wiener35.arousal." + v]
        gv, ga = gt["valence"][v].numpy(), gt["arousal"][v].numpy()
        valid = (gv >= -1) & (ga >= -1)
        arrs["ccc.valence." + v] = np.array(concordance_cc2_np(pv[valid], gv[valid]))
        arrs["ccc.arousal." + v] = np.array(concordance_cc2_np(pa[valid], ga[valid]))
        allp["valence"].append(pv[valid]); allp["arousal"].append(pa[valid])
        allg["valence"].append(gv[valid]); allg["arousal"].append(ga[valid])
    for k in ("valence", "arousal"):
        arrs["ccc_all." + k] = np.array(concordance_cc2_np(np.concatenate(allp[k]), np.concatenate(allg[k])))
    # create_submission.py:14-39 with a 2-model ensemble
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            second = {k: {v: torch.from_numpy((pred[k][v].numpy() * 0.8 + 0.05).astype(np.float32)) for v in vids} for k in pred}
            for v in vids:
                for k in ("valence", "arousal"):
                    arrs["pred2.%s.%s" % (k, v)] = second[k][v].numpy()
            torch.save({"valence_pred": pred["valence"], "arousal_pred": pred["arousal"]}, "m1.pt")
            torch.save({"valence_pred": second["valence"], "arousal_pred": second["arousal"]}, "m2.pt")
            open("videos.txt", "w").write("\n".join(vids) + "\n")
            open("scores.txt", "w").write("m1.pt\nm2.pt\n")
            create_submission.run_ensemble(open("videos.txt"), open("scores.txt"))
            for v in vids:
                arrs["submission." + v] = np.array(open(os.path.join("VA-Track", v + ".txt")).read())
        finally:
            os.chdir(cwd)
    save(name, **arrs)


def case_audio_stack(name, seed):
    """models/dataset.py:83-95 load_audio: the 5-frame mel context stacked per video frame, incl. the zero-padded tail."""
    import tempfile
    from models.dataset import load_audio       # reference (cv2 stubbed above)
    rs = np.random.RandomState(seed)
    mel = (rs.standard_normal((47, 40)) * 20 - 40).astype(np.float32)
    arrs = {"mel": mel}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "mel.npy")
        np.save(path, mel)
        for tag, (start, w_len) in {"head": (0, 8), "mid": (5, 6), "tail": (11, 8), "past": (16, 3)}.items():
            arrs["args." + tag] = np.array([start, w_len])
            arrs["out." + tag] = load_audio(path, start, w_len)
    save(name, **arrs)


def main():
    only = set(sys.argv[1:])

    def want(n):
        return not only or n in only

    if want("gru"):
        case_gru("gru_small", (24, 16, 2, 3, 2), 3, 13, 100)
        case_gru("gru_nofc_h", (12, 8, 2, -1), 2, 9, 110, return_h=True)
        case_gru("gru_fc3", (10, 8, 1, 2, 3), 2, 7, 120)
        case_gru("gru_scorer", (10, 8, 1, 1, 1), 4, 11, 130)
        case_gru("gru_t1", (6, 8, 2, 2, 2), 2, 1, 140)
    if want("tcn"):
        case_tcn("tcn_small", 8, [12, 12], 3, 2, 20, 200)
        case_tcn("tcn_k2_deep", 6, [6, 6, 6], 2, 3, 17, 210)
        case_tcn("tcn_short", 5, [7, 7, 7], 3, 2, 5, 220)      # T < receptive field
    if want("tcn_simple"):
        case_tcn_simple("tcn_simple_split_train", "split", 528, 3, 11, 230, True)
        case_tcn_simple("tcn_simple_split_eval", "split", 528, 2, 3, 240, False)     # T < kernel
        case_tcn_simple("tcn_simple_vggm_train", "vggm", 24, 2, 9, 250, True)
    if want("att"):
        case_attfusion("attfusion_same", [12, 12], 6, 3, 10, 300)
        case_attfusion("attfusion_proj", [12, 20], 6, 2, 9, 310)
    if want("cbam"):
        case_cbam("cbam_train", 32, 4, 7, 7, 400, True)
        case_cbam("cbam_eval", 32, 3, 5, 6, 410, False)
        case_cbam("cbam_c64", 64, 2, 14, 14, 420, True)
    if want("loss"):
        case_losses("losses", 500)
    if want("resnet"):
        case_resnet_cbam("resnet_cbam_eval", 600, False)
        case_resnet_cbam("resnet_cbam_train", 610, True)
    if want("c1"):
        case_seq_model("c1_tcn_head", lambda: RefTcnHead(128, 512, 2), (4, 128, 100), 700)
        case_affwild_audio("c1_affwild_audio", 710)
    if want("c2"):
        case_seq_model("c2_tcn_gru", lambda: RefTcnGru(256, 512), (2, 256, 300), 720)
    if want("c3"):
        case_c3("c3_av_graph", 2, 300, 12345)
        case_c3("c3_av_graph_small", 3, 17, 800, d_a=10, d_v=12, nh=512)
    if want("db"):
        # the reference's REAL audio input scale (raw dB, |x| ~ 40): saturates gates and scales the absolute error of the input
        # projection -- the 1e-4 bar must hold there too
        case_affwild_audio("c1_affwild_audio_db", 715, audio="db")
        case_c3("c3_av_graph_b32_db", 32, 300, 12345, with_norm=True, audio="db")
    if want("b32"):
        # BASELINE size: exactly the batch bench.py times (32 clips x 300 frames per GPU); y + gradient digests + the
        # clip norm only (weights and inputs are regenerated from the seed)
        case_c3("c3_av_graph_b32", 32, 300, 12345, with_norm=True)
        case_seq_model("c2_tcn_gru_b32", lambda: RefTcnGru(256, 512), (32, 256, 300), 12345, with_norm=True)
    if want("b32full"):
        # the same batch once more: the full gradient tensor of the fusion GRU's layer-0 recurrent weights ([1536, 512], 3 MB)
        case_c3("c3_av_graph_b32", 32, 300, 12345, with_norm=True,
                full_grads=("c3_av_graph_b32_gradfull", ["fusion.gru.weight_hh_l0"], True))
    if want("autocast"):
        case_seq_model_autocast("c2_tcn_gru_b32_autocast", lambda: RefTcnGru(256, 512), (32, 256, 300), 12345)
    if want("init"):
        case_init_digests("init_digests")
    if want("stitch"):
        case_stitch("stitch", 1000)
    if want("audio"):
        case_audio_stack("audio_stack", 1200)
    if want("postproc"):
        case_postproc("postproc", 1100)
    if want("c5"):
        case_affwild_av("c5_affwild_av", 900)
        case_resnet3d("c5_resnet3d_cbam", 910)
    if want("vggm"):
        case_vggm("vggm_tcn_eval", 920, "tcn")
        case_vggm("vggm_gru_eval", 930, "gru", B=2, T=3)
    if want("c5t16"):
        case_affwild_av("c5_affwild_av_t16", 940, B=2, T=16)
    if want("c5train"):
        case_affwild_av_train("c5_affwild_av_t16_train", 960, B=2, T=16)
        case_resnet3d_train("c5_resnet3d_cbam_train", 970, B=2, T=3)
    if want("c5t64"):
        # BASELINE's window of the end-to-end config (64 frames per clip): scans, stitching-free heads and the stems at the length bench.py times
        case_affwild_av("c5_affwild_av_t64", 950, B=2, T=64)


if __name__ == "__main__":
    main()
