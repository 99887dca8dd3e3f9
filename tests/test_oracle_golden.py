"""Pins the numpy oracle (oracle/m3t_oracle.py) against golden vectors produced by the
reference itself (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import m3t_oracle as O

TOL = 2e-5   # goldens are fp32 torch outputs; oracle runs in fp64


def P(g, dtype=np.float64):
    return {k[2:]: v.astype(dtype) for k, v in g.items() if k.startswith("p.")}


def G(g):
    return {k[2:]: v for k, v in g.items() if k.startswith("g.")}


def close(a, b, tol=TOL, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, "%s: max abs err %.3e (scale %.2f)" % (what, err, scale)


@pytest.mark.parametrize("name", ["gru_small", "gru_nofc_h", "gru_fc3", "gru_scorer", "gru_t1"])
def test_gru(name):
    g = load_golden(name)
    I, H, L, nC, nFC, ret_h = [int(v) for v in g["args"][:6]] if len(g["args"]) == 6 else (
        int(g["args"][0]), int(g["args"][1]), int(g["args"][2]), int(g["args"][3]), 1, int(g["args"][4]))
    p = P(g)
    y, h_n, cache = O.gru_module_fwd(g["x"].astype(np.float64), p, L, nC, nFC)
    close(y, g["y"], what="y")
    dh = None
    if ret_h:
        close(h_n, g["h"], what="h_n")
        dh = g["ct_h"].astype(np.float64)
    dx, grads = O.gru_module_bwd(g["ct"].astype(np.float64), cache, p, L, dh_n=dh)
    close(dx, g["dx"], what="dx")
    ref = G(g)
    assert set(ref) == set(grads)
    for k in ref:
        close(grads[k], ref[k], what=k)


@pytest.mark.parametrize("name", ["tcn_small", "tcn_k2_deep", "tcn_short"])
def test_tcn(name):
    g = load_golden(name)
    levels = len(g["args"]) - 2
    p = P(g)
    y, caches = O.tcn_fwd(g["x"].astype(np.float64), p, levels)
    close(y, g["y"], what="y")
    dx, grads = O.tcn_bwd(g["ct"].astype(np.float64), caches, p)
    close(dx, g["dx"], what="dx")
    ref = G(g)
    assert set(ref) == set(grads)
    for k in ref:
        close(grads[k], ref[k], what=k)


@pytest.mark.parametrize("name,k,n_out", [("tcn_simple_split_train", 5, 7), ("tcn_simple_split_eval", 5, 7),
                                           ("tcn_simple_vggm_train", 3, 2)])
def test_tcn_simple(name, k, n_out):
    """tcn_simple back-end (Conv1d same-padding + BatchNorm1d + ReLU, twice, then Linear) against the
    reference module run in train and eval mode; weights regenerated from the frozen recipe."""
    from golden.recipe import fill_by_shapes, simple_tcn_shapes, grad_digest
    g = load_golden(name)
    B, in_dim, T, training = [int(v) for v in g["dims"]]
    shapes = simple_tcn_shapes(in_dim, k, n_out)
    assert sorted(list(shapes) + ["0.1.num_batches_tracked", "0.4.num_batches_tracked"]) == list(g["state_dict_keys"])
    p = {n: v.astype(np.float64) for n, v in fill_by_shapes(shapes, int(g["seed"]) + 1).items()}
    h, caches, stats = O.simple_tcn_fwd(g["x"].astype(np.float64), p, (k - 1) // 2, bool(training), prefix="0.")
    y = O.linear_fwd(h.transpose(0, 2, 1), p["1.weight"], p["1.bias"])
    close(y, g["y"], what="y")
    for n, v in stats.items():
        close(v, g["rs." + n], what=n)
    dh, dw, db = O.linear_bwd(g["ct"].astype(np.float64), h.transpose(0, 2, 1), p["1.weight"])
    dx, grads = O.simple_tcn_bwd(dh.transpose(0, 2, 1), caches, (k - 1) // 2, prefix="0.")
    close(dx, g["dx"], what="dx")
    grads["1.weight"], grads["1.bias"] = dw, db
    assert set("gd." + n for n in grads) == set(kk for kk in g if kk.startswith("gd."))
    for n, v in grads.items():
        ref, got = g["gd." + n], grad_digest(v)
        assert abs(got[0] - ref[0]) <= 5e-5 * max(1.0, ref[0]), (n, got[0], ref[0])
        close(got[2:], ref[2:], tol=5e-5, what=n)


@pytest.mark.parametrize("name", ["attfusion_same", "attfusion_proj"])
def test_att_fusion(name):
    g = load_golden(name)
    p = P(g)
    y, cache = O.att_fusion_fwd(g["x_a"].astype(np.float64), g["x_v"].astype(np.float64), p)
    close(y, g["y"], what="y")
    dxa, dxv, grads = O.att_fusion_bwd(g["ct"].astype(np.float64), cache, p)
    close(dxa, g["dx_a"], what="dx_a")
    close(dxv, g["dx_v"], what="dx_v")
    ref = G(g)
    assert set(ref) == set(grads)
    for k in ref:
        close(grads[k], ref[k], what=k)


@pytest.mark.parametrize("name", ["cbam_train", "cbam_eval", "cbam_c64"])
def test_cbam(name):
    g = load_golden(name)
    p = P(g)
    training = bool(g["training"])
    y, cache, (rm, rv) = O.cbam_fwd(g["x"].astype(np.float64), p, training)
    close(y, g["y"], what="y")
    close(rm, g["running_mean_after"], what="running_mean")
    close(rv, g["running_var_after"], what="running_var")
    dx, grads = O.cbam_bwd(g["ct"].astype(np.float64), cache, p)
    close(dx, g["dx"], what="dx")
    ref = G(g)
    assert set(ref) == set(grads)
    for k in ref:
        close(grads[k], ref[k], what=k)


def test_losses():
    g = load_golden("losses")
    yh = g["y_hat"].astype(np.float64)
    ccc = O.concordance_cc2(yh[..., 7].reshape(-1), g["valence"].astype(np.float64).reshape(-1))
    close(ccc, g["ccc_v"], what="ccc")
    loss, parts, dy = O.training_loss_fwd_bwd(yh, g["valence"].astype(np.float64), g["arousal"].astype(np.float64),
                                              g["class_expr"], g["expr_valid"])
    close(parts["loss_v"], g["loss_v"], what="loss_v")
    close(parts["loss_a"], g["loss_a"], what="loss_a")
    close(parts["loss_expr"], g["loss_expr"], what="loss_expr")
    close(loss, g["loss"], what="loss")
    close(dy, g["dy_hat"], tol=1e-6, what="dy_hat")


def test_ccc_unbiased_var_biased_cov_quirk():
    """models/utils.py:13-14: var is unbiased (N-1), covariance biased (N)."""
    rs = np.random.RandomState(0)
    a, b = rs.randn(50), rs.randn(50)
    quirk = O.concordance_cc2(a, b)
    allbiased = 2 * np.mean((a - a.mean()) * (b - b.mean())) / (a.var() + b.var() + (a.mean() - b.mean()) ** 2)
    assert abs(quirk - allbiased) > 1e-4


def _c3_params(seed, d_a, d_v, nh):
    """Regenerate the recipe weights of the C3 graph WITHOUT the reference: the recipe
    walks sorted parameter names, so a name->shape table is enough."""
    from golden.recipe import c3_param_shapes, fill_by_shapes
    return fill_by_shapes(c3_param_shapes(d_a, d_v, nh), seed + 1)


@pytest.mark.parametrize("name", ["c3_av_graph_small", "c3_av_graph"])
def test_c3_graph(name):
    from golden.recipe import draw, grad_digest
    g = load_golden(name)
    B, T, d_a, d_v, nh = [int(v) for v in g["dims"]]
    seed = int(g["seed"])
    p = {k: v.astype(np.float64) for k, v in _c3_params(seed, d_a, d_v, nh).items()}
    rs = np.random.RandomState(seed)
    xa = draw(rs, (B, T, d_a)).astype(np.float64)
    xv = draw(rs, (B, T, d_v)).astype(np.float64)
    val = draw(rs, (B, T), "uniform_pm1").astype(np.float64)
    aro = draw(rs, (B, T), "uniform_pm1").astype(np.float64)
    expr = rs.randint(0, 7, (B, T)).astype(np.int64)
    valid = rs.uniform(size=(B, T)) < 0.7
    y, cache = O.av_feature_graph_fwd(xa, xv, p)
    close(y, g["y"], tol=5e-5, what="y")
    loss, parts, dy = O.training_loss_fwd_bwd(y, val, aro, expr, valid)
    close(loss, g["loss"], tol=5e-5, what="loss")
    assert round(float(1 - parts["loss_v"]), 3) == round(float(g["ccc_v"]), 3)
    assert round(float(1 - parts["loss_a"]), 3) == round(float(g["ccc_a"]), 3)
    dxa, dxv, grads = O.av_feature_graph_bwd(dy, cache, p)
    close(dxa[:, ::25], g["dx_a_full"], tol=5e-5, what="dx_a")
    close(dxv[:, ::25], g["dx_v_full"], tol=5e-5, what="dx_v")
    for k, v in grads.items():
        ref = g["gd." + k]
        got = grad_digest(v)
        assert abs(got[0] - ref[0]) <= 2e-4 * max(1.0, ref[0]), (k, got[0], ref[0])
        close(got[2:], ref[2:], tol=2e-4, what=k)


def test_postproc_smoothing_and_ccc():
    """f-4: Wiener-35 / Wiener-13 / median-13 smoothing and the numpy CCC report against the reference's own functions."""
    g = load_golden("postproc")
    names = [str(n) for n in g["names"]]
    allp, allg = {"valence": [], "arousal": []}, {"valence": [], "arousal": []}
    for v in names:
        sm = {}
        for k in ("valence", "arousal"):
            p = g["pred.%s.%s" % (k, v)]
            sm[k] = O.smooth_predictions(p, 35, "wiener")
            close(sm[k], g["wiener35.%s.%s" % (k, v)], tol=1e-6, what="wiener35 " + v)
            close(O.smooth_predictions(p), g["wiener13.%s.%s" % (k, v)], tol=1e-6, what="wiener13 " + v)
            close(O.smooth_predictions(p, 13, "median"), g["median13.%s.%s" % (k, v)], tol=0, what="median13 " + v)
        gv, ga = g["gt.valence." + v], g["gt.arousal." + v]
        valid = (gv >= -1) & (ga >= -1)
        for k, gg in (("valence", gv), ("arousal", ga)):
            close(O.concordance_cc2_np(sm[k][valid], gg[valid], r1_unbiased=True), g["ccc.%s.%s" % (k, v)], tol=1e-12, what="ccc " + v)
            allp[k].append(sm[k][valid]); allg[k].append(gg[valid])
    for k in ("valence", "arousal"):
        close(O.concordance_cc2_np(np.concatenate(allp[k]), np.concatenate(allg[k])), g["ccc_all." + k], tol=1e-12, what="ccc_all")


def test_audio_context_stacking():
    """f-3 (pinned half): models/dataset.py:83-95 load_audio on a 47-frame mel track, incl. windows that run past its end"""
    g = load_golden("audio_stack")
    for tag in ("head", "mid", "tail", "past"):
        start, w_len = [int(v) for v in g["args." + tag]]
        close(O.load_audio(g["mel"], start, w_len), g["out." + tag], tol=0, what=tag)


def test_melspec_oracle_known_answers():
    """f-3 (UNPINNED half: librosa is absent): sanity known-answers of the restated log-Mel pipeline -- frame count,
    silence -> the amin floor, a pure tone peaks in the mel band that contains it, dynamic range capped at top_db."""
    sr, fps = 16000, 30.0
    hop = int(1 / 3 * 1 / fps * 16000)
    t = np.arange(sr) / sr
    db = O.melspec_db(np.sin(2 * np.pi * 1000.0 * t), fps)
    assert db.shape == (1 + sr // hop, 40)
    mid = db[10:-10]
    peak = np.bincount(mid.argmax(1)).argmax()
    assert 13 <= peak <= 16, peak                               # 1 kHz = mel 15 of the Slaney scale: band 14/15 of 40 up to 8 kHz
    assert db.max() - db.min() <= 80.0 + 1e-9
    assert np.allclose(O.melspec_db(np.zeros(4000), fps), -100.0)   # 10 log10(amin = 1e-10)


def test_mel_scale_and_filterbank_against_librosa_published_values():
    """f-3: the Slaney mel scale and filterbank of the restated log-Mel pipeline against the known-answer values librosa
    publishes in its own docstrings (librosa.hz_to_mel, mel_to_hz, mel_frequencies, filters.mel examples) -- the third-party
    dependency of reference process/extract_melspec.py:13-20 is absent here, its published vectors are not."""
    assert abs(float(O.hz_to_mel(60)) - 0.9) < 1e-12
    np.testing.assert_allclose(O.hz_to_mel([110, 220, 440]), [1.65, 3.3, 6.6], rtol=0, atol=1e-12)
    assert abs(float(O.mel_to_hz(3)) - 200.0) < 1e-9
    np.testing.assert_allclose(O.mel_to_hz([1, 2, 3, 4, 5]), [66.667, 133.333, 200.0, 266.667, 333.333], rtol=0, atol=5e-4)
    doc = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856,
           1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686,
           2945.799, 3216.731, 3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009,
           7754.107, 8467.272, 9246.028, 10096.408, 11025.]                     # librosa.mel_frequencies(n_mels=40), fmax = 11025
    np.testing.assert_allclose(O.mel_frequencies(40, 0.0, 11025.0), doc, rtol=0, atol=6e-4)
    fb = O.mel_filterbank(22050, 2048, 128)                                     # librosa.filters.mel(sr=22050, n_fft=2048)
    assert fb.shape == (128, 1025) and abs(fb[0, 0]) < 1e-12 and round(float(fb[0, 1]), 3) == 0.016 and fb[0, -1] == 0 and fb[1, 0] == 0
    assert fb[-1, -1] == 0 and round(float(fb[-1, -2]), 3) == 0.0                # last rows of the printed example: [0., 0., ..., 0., 0.]
    # Slaney area normalisation: every triangle integrates to ~1 over frequency (bin width sr / n_fft)
    area = O.mel_filterbank(16000, 512, 40).sum(1) * (16000 / 512)
    assert np.all(np.abs(area - 1.0) < 0.12), area


def test_stft_power_against_torch_stft_and_power_to_db_closed_forms():
    """f-3: the framing / periodic-Hann / DFT stage against torch.stft (independent implementation), both paddings, and
    power_to_db on closed forms (ref 1, amin floor, top_db cap)."""
    import torch
    rs = np.random.RandomState(5)
    y = rs.standard_normal(4000)
    for fps, pad in ((30.0, "constant"), (25.0, "reflect")):
        hop = int(1 / 3 * 1 / fps * 16000)
        ref = torch.stft(torch.from_numpy(y), n_fft=512, hop_length=hop, win_length=400,
                         window=torch.hann_window(400, periodic=True, dtype=torch.float64), center=True, pad_mode=pad,
                         return_complex=True).abs().pow(2).t().numpy()
        got = O.stft_power(y, 512, hop, 400, pad)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(O.power_to_db(np.array([1.0, 10.0, 1e-3, 0.0]), top_db=None), [0.0, 10.0, -30.0, -100.0], atol=1e-12)
    np.testing.assert_allclose(O.power_to_db(np.array([1.0, 1e-12]), top_db=80.0), [0.0, -80.0], atol=1e-12)


def test_philox_known_answers_and_dropout_mask_layout():
    """the oracle's Philox4x32-10 against the Random123 known-answer vectors (kat_vectors: counter / key all zeros, all ones,
    and the pi digits), and the (row, col) -> (counter, word) layout of the dropout mask"""
    from oracle import m3t_oracle as O
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(w[0]) for w in got) == want, (ctr, [hex(int(w[0])) for w in got])
    m = O.dropout_mask(10, 6, 0.2, 0x0123456789abcdef)
    assert m.shape == (10, 6) and set(np.unique(m)) <= {0.0, float(np.float32(1.25))}
    w = O.philox4x32_10(np.array([5], np.uint32), np.array([2], np.uint32), np.array([0], np.uint32), np.array([0], np.uint32),
                        0x89abcdef, 0x01234567)
    thr = int(0.8 * 4294967296.0)
    assert [m[8, 5] > 0, m[9, 5] > 0] == [int(w[0][0]) < thr, int(w[1][0]) < thr]       # rows 8, 9 = words 0, 1 of group 2
    big = O.dropout_mask(4096, 64, 0.2, 7)
    assert abs((big > 0).mean() - 0.8) < 0.01


def test_product_mel_filterbank_equals_the_pinned_oracle():
    """the host-side constant of the GPU log-Mel front-end (m3t/audio.py: the filterbank the GEMM multiplies by) is the oracle's
    filterbank, which the librosa-published values above pin -- no GPU needed"""
    from m3t import audio
    for sr, n_fft, n_mels in ((16000, 512, 40), (22050, 2048, 128)):
        fb = np.asarray(audio.mel_filterbank(sr, n_fft, n_mels), dtype=np.float64)
        ref = O.mel_filterbank(sr, n_fft, n_mels)
        assert fb.shape == ref.shape
        np.testing.assert_allclose(fb, ref, rtol=0, atol=2e-9 if fb.dtype == np.float64 else 1e-7)
    assert audio.hop_length(30.0) == int(1 / 3 * 1 / 30.0 * 16000) and audio.hop_length(25.0) == 213


@pytest.mark.parametrize("name,db", [("c1_affwild_audio", False), ("c1_affwild_audio_db", True)])
def test_affwild_audio_training_step(name, db):
    """AffWild2VA(modality='audio', loss='ccc_mtl') = GRU(200,256,2,9,2) + the training_step loss assembly (reference
    models/model.py:86,102-103,146-182) -- on N(0,1) inputs and on the dB scale the reference really feeds, U(-80, 0)
    (reference models/dataset.py:83-95): the oracle must be pinned there too before the GPU path is judged against it."""
    from golden.recipe import draw, grad_digest, fill_by_shapes, gru_shapes
    g = load_golden(name)
    seed = int(g["seed"])
    p = {k: v.astype(np.float64) for k, v in fill_by_shapes(gru_shapes("audio.", 200, 256, 2, 9, 2), seed + 1).items()}
    assert sorted(p) == sorted(str(n) for n in g["param_names"])
    rs = np.random.RandomState(seed)
    B, T = 4, 100
    x = rs.uniform(-80.0, 0.0, (B, T, 200)).astype(np.float32) if db else draw(rs, (B, T, 200))
    val, aro = draw(rs, (B, T), "uniform_pm1").astype(np.float64), draw(rs, (B, T), "uniform_pm1").astype(np.float64)
    expr = rs.randint(0, 7, (B, T)).astype(np.int64)
    valid = rs.uniform(size=(B, T)) < 0.7
    pa = {k[len("audio."):]: v for k, v in p.items()}
    y, _, cache = O.gru_module_fwd(x.astype(np.float64), pa, 2, 9, 2)
    close(y, g["y"], tol=2e-5, what="y")
    loss, parts, dy = O.training_loss_fwd_bwd(y, val, aro, expr, valid)
    close(loss, g["loss"], tol=2e-5, what="loss")
    close(parts["loss_v"], g["loss_v"], tol=2e-5, what="loss_v")
    close(parts["loss_expr"], g["loss_expr"], tol=2e-5, what="loss_expr")
    _, grads = O.gru_module_bwd(dy, cache, pa, 2)
    for k, v in grads.items():
        ref = g["gd.audio." + k]
        got = grad_digest(v)
        assert abs(got[0] - ref[0]) <= 2e-4 * max(1.0, ref[0]), (k, got[0], ref[0])
        close(got[2:], ref[2:], tol=2e-4, what=k)
