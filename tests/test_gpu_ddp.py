"""Single-GPU check of the fused gradient post-processing kernel behind FlatGradDDP."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,world,max_norm", [(1000003, 1, 1.0), (26397707, 8, 1.0), (4099, 2, 0.0), (77, 4, 1e9)])
def test_grad_norm_scale_kernel(n, world, max_norm):
    from m3t import ops
    torch.manual_seed(n % 97)
    g = torch.randn(n, device="cuda:0") * 3e-3
    ref = g.double() / world
    norm = ref.norm()
    if max_norm > 0:
        ref = ref * min(1.0, max_norm / (float(norm) + 1e-6))
    got_norm = ops.grad_norm_scale_(g, world, max_norm)
    assert abs(float(got_norm) - float(norm)) <= 1e-5 * max(1.0, float(norm))
    assert float((g.double() - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))


def test_flat_ddp_single_gpu_matches_plain_autograd():
    from m3t.ddp import FlatGradDDP
    from models.rnn import GRU
    torch.manual_seed(1)
    a, b = GRU(12, 16, 2, 3, 2).to("cuda:0"), GRU(12, 16, 2, 3, 2).to("cuda:0")
    b.load_state_dict(a.state_dict())
    x = torch.randn(4, 9, 12, device="cuda:0")
    ddp = FlatGradDDP(a, max_norm=0.0)
    for _ in range(2):
        ddp.zero_grad()
        a(x).square().mean().backward()
        ddp.finish()
    b(x).square().mean().backward()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-7), n
