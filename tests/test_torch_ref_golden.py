"""The CPU-baseline port (oracle/torch_ref.py) reproduces the reference's golden outputs."""
import numpy as np
import torch

from conftest import load_golden
from golden.recipe import fill_module, draw
from oracle import torch_ref as R


def test_c3_port_matches_reference_golden():
    g = load_golden("c3_av_graph_small")
    B, T, d_a, d_v, nh = [int(v) for v in g["dims"]]
    seed = int(g["seed"])
    m = fill_module(R.RefAVFeatureGraph(d_a, d_v, nh), seed + 1)
    rs = np.random.RandomState(seed)
    xa, xv = torch.from_numpy(draw(rs, (B, T, d_a))), torch.from_numpy(draw(rs, (B, T, d_v)))
    val, aro = torch.from_numpy(draw(rs, (B, T), "uniform_pm1")), torch.from_numpy(draw(rs, (B, T), "uniform_pm1"))
    expr = torch.from_numpy(rs.randint(0, 7, (B, T)).astype(np.int64))
    valid = torch.from_numpy(rs.uniform(size=(B, T)) < 0.7)
    y = m(xa, xv)
    assert np.abs(y.detach().numpy() - g["y"]).max() < 1e-5
    loss = R.mtl_loss(y, val, aro, expr, valid)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
