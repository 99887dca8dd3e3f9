"""models.utils -- metric helpers (reference models/utils.py:6-22).

`concordance_cc2` keeps the reference's quirk: torch's Tensor.var is UNBIASED (N-1) while the
covariance term is BIASED (N).  This helper is evaluation/reporting glue on whatever device
its inputs live on; the TRAINING loss does not go through it -- AffWild2VA.training_step uses
the fused HIP kernel m3t_va_loss (loss + gradient in one pass).  Smoothing / plotting helpers
of the reference (utils.py:29-45) are post-processing and out of scope.
"""
import numpy as np


def concordance_cc2(r1, r2, reduction='mean'):
    m1 = r1.mean(dim=-1, keepdim=True)
    m2 = r2.mean(dim=-1, keepdim=True)
    cov = ((r1 - m1) * (r2 - m2)).mean(dim=-1, keepdim=True)
    denom = r1.var(dim=-1, keepdim=True) + r2.var(dim=-1, keepdim=True) + (m1 - m2) ** 2
    ccc = 2 * cov / denom
    if reduction == 'none':
        return ccc
    if reduction == 'mean':
        return ccc.mean()
    raise ValueError(reduction)


def concordance_cc2_np(r1, r2):
    r1, r2 = np.asarray(r1), np.asarray(r2)
    cov = ((r1 - r1.mean()) * (r2 - r2.mean())).mean()
    return 2 * cov / (r1.var() + r2.var() + (r1.mean() - r2.mean()) ** 2)


def mse(preds, labels):
    return sum((preds - labels) ** 2) / len(labels)
