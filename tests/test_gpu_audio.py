"""f-3 audio front-end on the GPU (m3t.audio over csrc/audio.hip + m3t_sgemm).  The context stacking is pinned on the
reference's own load_audio (golden audio_stack.npz); the log-Mel half has no reference output (librosa is absent and
unpinned: PARITY UNPINNED) and is checked against the numpy restatement in the oracle, tolerance 2e-3 dB (fp32 DFT by
GEMM vs fp64 FFT; 1e-2 dB where bands sit near the amin floor is not needed: inputs carry a noise floor)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import m3t_oracle as O

pytestmark = pytest.mark.gpu


def test_context_stacking_matches_reference():
    from m3t import audio
    g = load_golden("audio_stack")
    for tag in ("head", "mid", "tail", "past"):
        start, w_len = [int(v) for v in g["args." + tag]]
        got = audio.load_audio(g["mel"], start, w_len)
        assert got.shape == (w_len, 200) and got.dtype == torch.float32
        assert np.array_equal(got.cpu().numpy(), g["out." + tag]), tag


@pytest.mark.parametrize("seconds,fps,pad_mode", [(2.0, 30.0, "constant"), (0.7, 25.0, "reflect"), (5.3, 29.97, "constant")])
def test_logmel_vs_oracle(seconds, fps, pad_mode):
    from m3t import audio
    rs = np.random.RandomState(int(seconds * 10))
    n = int(16000 * seconds)
    t = np.arange(n) / 16000.0
    y = (0.4 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t + 1.0) * (t > 0.3)
         + 0.05 * rs.standard_normal(n)).astype(np.float32)
    ref = O.melspec_db(y.astype(np.float64), fps, pad_mode=pad_mode)
    got = audio.melspec_db(y, fps, pad_mode=pad_mode)
    assert tuple(got.shape) == ref.shape == (1 + n // audio.hop_length(fps), 40)
    err = float(np.abs(got.cpu().numpy().astype(np.float64) - ref).max())
    assert err < 2e-3, err
    # the model input built from it: [T, 200] rows for a 16-frame window starting at video frame 3
    feats = audio.load_audio(got, 3, 16)
    assert np.allclose(feats.cpu().numpy(), O.load_audio(ref, 3, 16), atol=2e-3)


def test_logmel_silence_hits_the_floor():
    from m3t import audio
    got = audio.melspec_db(np.zeros(8000, np.float32), 30.0)
    assert torch.allclose(got, torch.full_like(got, -100.0))
