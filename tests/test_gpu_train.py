"""Fused optimizer kernels vs torch.optim, the Lightning-free trainer loop, checkpoint and eval hooks (GPU)."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net():
    from models.rnn import GRU
    torch.manual_seed(4)
    return GRU(12, 16, 2, 3, 2).to(DEV)


@pytest.mark.parametrize("kind", ["adam", "sgd"])
def test_flat_optimizer_matches_torch(kind):
    from m3t.ddp import FlatGradDDP
    from m3t.optim import FlatAdam, FlatSGD
    a, b = _net(), _net()
    b.load_state_dict(a.state_dict())
    ddp = FlatGradDDP(a, max_norm=0.0, flatten_params=True)
    if kind == "adam":
        mine, ref = FlatAdam(ddp, lr=1e-3, weight_decay=1e-4), torch.optim.Adam(b.parameters(), lr=1e-3, weight_decay=1e-4)
    else:
        mine, ref = FlatSGD(ddp, lr=1e-2, momentum=0.9, weight_decay=5e-4), torch.optim.SGD(b.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
    x = torch.randn(4, 9, 12, device=DEV)
    for _ in range(5):
        ddp.zero_grad()
        a(x).square().mean().backward()
        ddp.finish()
        mine.step()
        ref.zero_grad()
        b(x).square().mean().backward()
        ref.step()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(p, q, atol=2e-6, rtol=1e-5), n


def _hp(**kw):
    from models.model import AffWild2VA
    ns = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False)).parse_args([])
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def _audio_batch(B=4, T=40, seed=0):
    rs = np.random.RandomState(seed)
    f = lambda a: torch.from_numpy(a).to(DEV)
    audio = rs.standard_normal((B, T, 200)).astype(np.float32)
    # labels the model can actually learn: smooth functions of the input
    val = np.tanh(audio[..., :20].mean(-1) * 3).astype(np.float32)
    aro = np.tanh(audio[..., 20:40].mean(-1) * 3).astype(np.float32)
    return {"audio": f(audio), "label_valence": f(val), "label_arousal": f(aro),
            "class_expr": f(rs.randint(0, 7, (B, T)).astype(np.int64)), "expr_valid": f(rs.uniform(size=(B, T)) < 0.7),
            "vid_name": ["v%d" % i for i in range(B)], "start": torch.zeros(B, dtype=torch.long),
            "length": torch.full((B,), T, dtype=torch.long)}


def test_trainer_learns_and_checkpoints(tmp_path):
    from models.model import AffWild2VA
    from m3t.trainer import Trainer
    torch.manual_seed(12345)
    model = AffWild2VA(_hp(modality="audio", loss="ccc_mtl", learning_rate=2e-3)).to(DEV)
    tr = Trainer.from_hparams(model, model.hparams)
    batch = _audio_batch()
    first = float(tr.step(batch)["loss"].detach())
    for _ in range(40):
        out = tr.step(batch)
    last = float(out["loss"].detach())
    assert np.isfinite(last) and last < first - 0.3, (first, last)
    assert float(out["grad_norm"]) >= 0
    # {'state_dict': ...} checkpoint: what the reference's eval.py loads with strict=True (eval.py:14-15)
    path = os.path.join(tmp_path, "ckpt.pt")
    tr.save_checkpoint(path)
    fresh = AffWild2VA(_hp(modality="audio", loss="ccc_mtl")).to(DEV)
    fresh.load_state_dict(torch.load(path, map_location="cpu")["state_dict"], strict=True)
    model.eval(); fresh.eval()
    assert torch.equal(fresh(batch), model(batch))


def test_eval_hooks_write_reference_prediction_files(tmp_path, monkeypatch):
    from models.model import AffWild2VA
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(1)
    model = AffWild2VA(_hp(modality="audio", loss="ccc_mtl", window=40)).to(DEV).eval()
    outs = [model.validation_step(_audio_batch(seed=s), s) for s in range(2)]
    ret = model.validation_end(outs)
    assert set(ret["log"]) == {"val_ccc_v", "val_ccc_a", "val_mse_v", "val_mse_a", "val_loss"}
    saved = torch.load("predictions_val.pt")
    assert set(saved) == {"valence_gt", "arousal_gt", "valence_pred", "arousal_pred"} and "v0" in saved["valence_pred"]
    model.test_end([model.test_step(_audio_batch(seed=3), 0)])
    assert set(torch.load("predictions_test.pt")) == {"valence_pred", "arousal_pred"}
