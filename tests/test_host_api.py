"""CPU-side checks: the C-ABI library loads and exports every symbol include/m3t_hip.h declares,
the drop-in modules keep the reference's parameter names/shapes, and the product path fails
loudly (no CPU fallback).  No GPU compute here."""
import ctypes
import os
import re
import argparse

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from m3t import _lib, ops
from golden.recipe import c3_param_shapes, gru_shapes, tcn_shapes, att_fusion_shapes


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "m3t_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t)\s+(m3t_\w+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    names = _header_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libm3t_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_library_version_and_loader():
    lib = _lib.load()
    assert lib.m3t_version() == 1


def test_desc_struct_layout_matches_header(tmp_path):
    """The ctypes descriptor structures against the C compiler's view of include/m3t_hip.h: size and every field offset."""
    import subprocess
    fields = {"m3t_gru_fwd_desc": [f[0] for f in _lib.GruFwdDesc._fields_],
              "m3t_gru_bwd_desc": [f[0] for f in _lib.GruBwdDesc._fields_]}
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "m3t_hip.h"', "int main(void) {"]
    for st, names in fields.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for n in names:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, n, st, n))
    src += ["return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for st, cls in (("m3t_gru_fwd_desc", _lib.GruFwdDesc), ("m3t_gru_bwd_desc", _lib.GruBwdDesc)):
        assert int(got[st]) == ctypes.sizeof(cls), st
        for n in fields[st]:
            assert int(got["%s.%s" % (st, n)]) == getattr(cls, n).offset, (st, n)


def test_persistent_scan_prefetch_is_not_copied_in_flight(tmp_path):
    """gru_persist.hip requests the next step's activations with inline-asm loads whose completion only the NEXT gather's
    vmcnt(0) guarantees.  If hipcc ever merges such a value with a register copy (or spills it) before that wait, the copy
    reads a register still in flight -- silently wrong results.  Compile the listing and check every such block
    (tools/check_inflight.py); the sgemm planner is checked on the way (no device work)."""
    import subprocess
    import sys
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    lst = str(tmp_path / "gru_persist.s")
    csrc = os.path.join(ROOT, "m3f.pytorch_amd", "csrc")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    "-Wno-unused-function", "-fno-gpu-rdc", "-S", "--cuda-device-only", os.path.join(csrc, "gru_persist.hip"),
                    "-o", lst], check=True, capture_output=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight.py"), lst, "gru_persist_"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    checked = [int(m) for m in re.findall(r"(\d+) asm load blocks checked", r.stdout)]
    # every persistent scan kernel of the listing (fp32 / bf16x6 / fp16x3 / bf16 / producer-split, narrow and wide, with and without
    # the coalesced hand-over) carries the prologue's and both loop bodies' prefetch blocks
    n_kernels = len(set(re.findall(r"^(_ZN7m3t_gru\w*gru_persist_\w+_kernel\w+):", open(lst).read(), re.M)))
    assert n_kernels >= 29 and sum(1 for c in checked if c >= 3) == n_kernels, r.stdout      # 26 of round 3 + wide fp16x3 forward + wide producer-split backward (+ its profiling instantiation)
    assert ops.sgemm_plan(0, 9600, 1536, 1024) [0] == 1 and ops.sgemm_plan(0, 9600, 1536, 1024, exclusive=True, prec=0)[0] == 1
    assert ops.sgemm_plan(0, 9600, 1536, 1024, exclusive=True)[0] == 1       # every interior shape runs on the 128-tile kernels (the 256-tile kernel is gone)
    assert ops.sgemm_plan(0, 300, 257, 130)[0] == 0


def test_no_cpu_fallback():
    x = torch.randn(4, 8)
    w = torch.randn(3, 8)
    with pytest.raises(_lib.M3THipError):
        ops.linear(x, w, None, 0)
    from models.rnn import GRU
    with pytest.raises(_lib.M3THipError):
        GRU(8, 4, 1, 2)(torch.randn(2, 5, 8))


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libm3t_hip.so")
    with pytest.raises(_lib.M3THipError):
        _lib.load()


def _hp(**kw):
    from models.model import AffWild2VA
    ns = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False)).parse_args([])
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def test_flag_defaults_match_reference():
    ns = _hp()
    assert (ns.backbone, ns.backend, ns.modality, ns.fusion_type) == ("v2p_split", "gru", "visual", "concat")
    assert (ns.window, ns.batch_size, ns.loss, ns.num_hidden, ns.split_layer, ns.num_fc_layers) == (32, 96, "ccc_mtl", 512, 3, 2)
    assert ns.loss_lambda == 0.5 and ns.learning_rate == 5e-5 and not ns.distributed


def test_affwild_state_dict_keys_match_reference():
    from models.model import AffWild2VA
    g = load_golden("c5_affwild_av")
    m = AffWild2VA(_hp(modality="audiovisual", fusion_type="attention", window=4))
    assert sorted(m.state_dict().keys()) == sorted(g["state_dict_keys"].tolist())
    assert sorted(n for n, _ in m.named_parameters()) == sorted(g["param_names"].tolist())
    assert sum(p.numel() for p in m.parameters()) == 53579147


def test_audio_model_param_names():
    from models.model import AffWild2VA
    g = load_golden("c1_affwild_audio")
    m = AffWild2VA(_hp(modality="audio"))
    assert sorted(n for n, _ in m.named_parameters()) == sorted(g["param_names"].tolist())


def test_resnet3d_param_names():
    from models.backbone import VA_3DResNet
    g = load_golden("c5_resnet3d_cbam")
    m = VA_3DResNet(frameLen=3, resnet_ver="v1", use_cbam=True, nClasses=2, nFCs=2)
    assert sorted(n for n, _ in m.named_parameters()) == sorted(g["param_names"].tolist())


@pytest.mark.parametrize("name,backend,T", [("vggm_tcn_eval", "tcn", 4), ("vggm_gru_eval", "gru", 3)])
def test_vggm_state_dict_keys_match_reference(name, backend, T):
    """VA_3DVGGM (reference models/backbone.py:62-161): checkpoint keys and parameter names of the class itself -- the only
    user of TemporalConvNet (its `tcn.0.network.*` keys carry the weight-norm aliases) -- equal the reference's"""
    from models.backbone import VA_3DVGGM
    g = load_golden(name)
    m = VA_3DVGGM(frameLen=T, backend=backend, nClasses=2, nFCs=2)
    assert sorted(m.state_dict().keys()) == sorted(g["state_dict_keys"].tolist())
    assert sorted(n for n, _ in m.named_parameters()) == sorted(g["param_names"].tolist())


def test_tcn_state_dict_aliases():
    from models.tcn import TemporalConvNet
    g = load_golden("tcn_small")
    m = TemporalConvNet(8, [12, 12], 3)
    keys = sorted(m.state_dict().keys())
    assert keys == sorted(g["state_dict_keys"].tolist())
    assert "network.0.net.0.weight_v" in keys and "network.0.conv1.weight_v" in keys
    sd = m.state_dict()
    assert sd["network.0.net.4.weight_g"].data_ptr() == sd["network.0.conv2.weight_g"].data_ptr()


def test_module_shapes_follow_shape_tables():
    from models.rnn import GRU
    from models.tcn import TemporalConvNet
    from models.att_fusion import AttFusion
    from m3t.workloads import AVFeatureGraph
    shp = lambda m: {n: tuple(p.shape) for n, p in m.named_parameters()}
    assert shp(GRU(24, 16, 2, 3, 2)) == gru_shapes("", 24, 16, 2, 3, 2)
    assert shp(GRU(10, 8, 1, 2, 3, dropout=True)) == gru_shapes("", 10, 8, 1, 2, 3, dropout=True)
    assert shp(TemporalConvNet(8, [12, 12], 3)) == tcn_shapes("", 8, [12, 12], 3)
    assert shp(AttFusion([12, 20], 6)) == att_fusion_shapes("", [12, 20], 6)
    assert shp(AVFeatureGraph()) == c3_param_shapes()


def test_gru_init_recipe():
    """reference models/rnn.py:57-69: zero biases, per-gate orthogonal W_hh, bounded uniform W_ih."""
    from models.rnn import GRU
    torch.manual_seed(0)
    m = GRU(20, 12, 2, 2, 2)
    H = 12
    lim = np.sqrt(3.0) * np.sqrt(2.0 / (20 + 12))
    for n, p in m.gru.named_parameters():
        if "bias" in n:
            assert float(p.abs().max()) == 0.0
        elif "weight_hh" in n:
            for g0 in range(0, 3 * H, H):
                blk = p[g0:g0 + H].detach()
                assert torch.allclose(blk @ blk.t(), torch.eye(H), atol=1e-5)
        elif "weight_ih" in n:
            assert float(p.abs().max()) <= lim + 1e-7


def test_init_digests_match_reference_rng_stream():
    """Same torch.manual_seed => same initial weights as the reference constructors."""
    g = load_golden("init_digests")
    from models.rnn import GRU
    from models.tcn import TemporalConvNet
    from models.att_fusion import AttFusion
    from models.cbam import CBAM
    from golden.recipe import grad_digest
    ctors = {"gru": lambda: GRU(24, 16, 2, 3, 2), "tcn": lambda: TemporalConvNet(8, [12, 12], 3),
             "att": lambda: AttFusion([12, 20], 6), "cbam": lambda: CBAM(32)}
    for tag, ctor in ctors.items():
        torch.manual_seed(12345)
        m = ctor()
        for n, p in m.named_parameters():
            np.testing.assert_allclose(grad_digest(p.detach().numpy()), g["%s.%s" % (tag, n)], rtol=1e-6, atol=1e-7,
                                       err_msg="%s.%s" % (tag, n))


def test_flag_constants_match_the_header():
    """every M3T_* flag / error constant the ctypes layer defines carries the value include/m3t_hip.h gives it"""
    txt = open(os.path.join(ROOT, "include", "m3t_hip.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(M3T_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)\b", txt, flags=re.M)}
    checked = 0
    for name, val in vars(_lib).items():
        if name.startswith("M3T_") and isinstance(val, int) and name in defs:
            assert defs[name] == val, (name, defs[name], val)
            checked += 1
    assert checked >= 8, checked
    with pytest.raises(ValueError):
        ops.precision("fp16")
    assert ops.precision("high").flag == defs["M3T_GEMM_HIGH"] and ops.precision("bf16").flag == defs["M3T_BF16"]
