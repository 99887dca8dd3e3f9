"""Lightning-free training loop for AffWild2VA-style modules: what the reference gets
from `pl.Trainer(gradient_clip_val=1.0, distributed_backend='ddp')` (reference train.py:32-42) reduced to the
hot loop -- zero flat grads, training_step, backward, RCCL all-reduce + clip, fused optimizer step -- plus
`{'state_dict': ...}` checkpoints the reference's eval.py can load (reference eval.py:14-15)."""
import torch

from .ddp import FlatGradDDP
from .optim import FlatAdam, FlatSGD


class Trainer:
    def __init__(self, model, optimizer="adam", learning_rate=5e-5, gradient_clip_val=1.0, process_group=None):
        self.model = model
        self.ddp = FlatGradDDP(model, max_norm=gradient_clip_val, process_group=process_group, flatten_params=True)
        if optimizer == "adam":
            self.opt = FlatAdam(self.ddp, lr=learning_rate, weight_decay=1e-4)
        elif optimizer == "sgd":
            self.opt = FlatSGD(self.ddp, lr=learning_rate, momentum=0.9, weight_decay=5e-4)
        else:
            raise ValueError(optimizer)
        self.global_step = 0

    @classmethod
    def from_hparams(cls, model, hparams, **kw):
        return cls(model, optimizer=hparams.optimizer, learning_rate=hparams.learning_rate, **kw)

    def step(self, batch):
        """One optimisation step; returns the dict of model.training_step (loss is a device scalar)."""
        self.model.train()
        self.ddp.zero_grad()
        out = self.model.training_step(batch, self.global_step)
        out["loss"].backward()
        out["grad_norm"] = self.ddp.finish()
        self.opt.step()
        self.global_step += 1
        return out

    def fit(self, batches, max_steps=None, log_every=0):
        history = []
        for i, batch in enumerate(batches):
            if max_steps is not None and i >= max_steps:
                break
            out = self.step(batch)
            if log_every and i % log_every == 0:
                history.append(float(out["loss"].detach()))
        return history

    def save_checkpoint(self, path):
        torch.save({"state_dict": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                    "global_step": self.global_step}, path)
