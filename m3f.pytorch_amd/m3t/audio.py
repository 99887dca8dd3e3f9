"""Audio front-end on the GPU: the log-Mel features the reference extracts offline
(`process/extract_melspec.py:8-20`) and the context stacking of `models/dataset.py:83-95` that turns them into the
`[T, 200]` rows `AffWild2VA.forward` consumes as `batch['audio']`.

  melspec_db(y, fps)               = librosa.power_to_db(librosa.feature.melspectrogram(y=y, sr=16000, n_fft=512,
                                       hop_length=int(1/3 * 1/fps * 16000), win_length=400, n_mels=40)).T   -> [frames, 40]
  load_audio(mel, start_idx, w_len) = models/dataset.py:83-95 (rows (start+i)*3 .. +5, zero padded, flattened) -> [w_len, 200]

librosa is a third-party dependency the reference does not pin (requirements.txt has no version) and it is absent from
this image, so the spectrogram half is restated from librosa's published algorithm -- periodic Hann window of
`win_length` centred in `n_fft`, `center=True` framing, power spectrum, Slaney mel filterbank (`htk=False`,
`norm='slaney'`, 0..sr/2), `power_to_db(ref=1, amin=1e-10, top_db=80)` -- and its parity is UNPINNED: it is checked
against the numpy oracle only.  `pad_mode` defaults to librosa >= 0.10's zero padding ('constant'); older librosa
reflected ('reflect').  The stacking half is pinned on the reference's own function.

Constants (window, DFT matrix, filterbank) are built once on the host in float64; the data path is HIP:
m3t_frame_window -> m3t_sgemm (DFT, fp32-accurate) -> m3t_power_spectrum -> m3t_sgemm (mel) -> m3t_power_to_db.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .ops import lib, _stream, _p, sgemm, workspace, M3THipError

SR, N_FFT, WIN, N_MELS = 16000, 512, 400, 40
_CONST = {}


def hop_length(fps):
    """process/extract_melspec.py:15"""
    return int(1 / 3 * 1 / fps * 16000)


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS):
    """librosa.filters.mel(htk=False, norm='slaney', fmin=0, fmax=sr/2) -> [n_mels, 1 + n_fft/2] float64"""
    def hz_to_mel(f):
        f = np.asarray(f, np.float64)
        mel = f / (200.0 / 3)
        logstep = math.log(6.4) / 27.0
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / logstep, mel)

    def mel_to_hz(m):
        m = np.asarray(m, np.float64)
        logstep = math.log(6.4) / 27.0
        return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), m * (200.0 / 3))

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


def _constants(device):
    key = (device.type, device.index)
    c = _CONST.get(key)
    if c is None:
        win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(WIN) / WIN)           # periodic Hann (scipy get_window fftbins=True)
        pad = (N_FFT - WIN) // 2
        win = np.concatenate([np.zeros(pad), win, np.zeros(N_FFT - WIN - pad)])
        bins = 1 + N_FFT // 2
        k, n = np.arange(bins)[None, :], np.arange(N_FFT)[:, None]
        ang = 2 * np.pi * k * n / N_FFT
        dft = np.concatenate([np.cos(ang), -np.sin(ang)], 1)                # [n_fft, 2*bins]: re | im
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
        c = (dev(win), dev(dft), dev(mel_filterbank().T))                   # mel as [bins, n_mels]
        _CONST[key] = c
    return c


def melspec_db(y, fps=30.0, pad_mode="constant", top_db=80.0):
    """y: 1-D float waveform at 16 kHz (array or tensor) -> [frames, 40] float32 CUDA tensor of log-Mel energies (dB),
    frames = 1 + len(y) // hop."""
    if not torch.cuda.is_available():
        raise M3THipError("m3t.audio needs the GPU: the M3T path has no CPU fallback")
    if pad_mode not in ("constant", "reflect"):
        raise ValueError("pad_mode must be 'constant' or 'reflect'")
    dev = torch.device("cuda", torch.cuda.current_device())
    y = torch.as_tensor(np.asarray(y) if not isinstance(y, torch.Tensor) else y).detach().to(dev, torch.float32).reshape(-1).contiguous()
    n, hop = int(y.numel()), hop_length(fps)
    if n == 0 or hop <= 0:
        raise M3THipError("melspec_db: empty waveform or non-positive hop")
    win, dft, melT = _constants(dev)
    bins = 1 + N_FFT // 2
    nf = 1 + n // hop
    frames = torch.empty(nf, N_FFT, dtype=torch.float32, device=dev)
    _lib.check(lib().m3t_frame_window(_p(y), n, N_FFT, hop, 1 if pad_mode == "reflect" else 0, _p(win), _p(frames), nf,
                                      _stream()), "m3t_frame_window")
    spec = torch.empty(nf, 2 * bins, dtype=torch.float32, device=dev)
    sgemm(0, 0, nf, 2 * bins, N_FFT, frames, 0, N_FFT, dft, 0, 2 * bins, spec, 0, 2 * bins, prec=0, exclusive=True)
    power = torch.empty(nf, bins, dtype=torch.float32, device=dev)
    _lib.check(lib().m3t_power_spectrum(_p(spec), nf, bins, _p(power), _stream()), "m3t_power_spectrum")
    mel = torch.empty(nf, N_MELS, dtype=torch.float32, device=dev)
    sgemm(0, 0, nf, N_MELS, bins, power, 0, bins, melT, 0, N_MELS, mel, 0, N_MELS, prec=0, exclusive=True)
    out = torch.empty_like(mel)
    ws = workspace(dev)
    _lib.check(lib().m3t_power_to_db(_p(mel), mel.numel(), 1e-10, float(top_db), _p(out), _p(ws), ws.numel() * 4, _stream()),
               "m3t_power_to_db")
    return out


def load_audio(mel_spec, start_idx, w_len):
    """models/dataset.py:83-95 with the .npy already in memory: mel_spec [frames, 40] (array, tensor or path) ->
    [w_len, 200] float32 CUDA tensor: row i = mel rows (start_idx + i)*3 .. +5, zero padded past the end."""
    if isinstance(mel_spec, str):
        mel_spec = np.load(mel_spec)
    dev = torch.device("cuda", torch.cuda.current_device())
    mel = torch.as_tensor(np.asarray(mel_spec) if not isinstance(mel_spec, torch.Tensor) else mel_spec).detach().to(dev, torch.float32).contiguous()
    out = torch.empty(int(w_len), 5 * mel.shape[1], dtype=torch.float32, device=dev)
    _lib.check(lib().m3t_stack_context(_p(mel), mel.shape[0], mel.shape[1], int(start_idx), int(w_len), 3, 5, _p(out), _stream()),
               "m3t_stack_context")
    return out
