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


def test_gradient_sinks_match_autograd_accumulation():
    """m3t.ops writes GRU / Linear weight gradients straight into FlatGradDDP's flat buffer (no AccumulateGrad kernels).
    Same gradients as the ordinary path; a parameter used twice in one graph (second gradient must be ADDED), a
    parameter nothing touches (must read zero, not last step's value), and grads reset to None by the user (sinks must
    stand down) all behave."""
    from m3t import ops
    from m3t.ddp import FlatGradDDP
    from models.rnn import GRU

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gru = GRU(12, 16, 2, 3, 2)
            self.unused = torch.nn.Parameter(torch.ones(5))

        def forward(self, x):
            return self.gru(x) + 0.5 * self.gru(x.flip(1))          # every parameter of the GRU is used twice

    torch.manual_seed(3)
    a, b = Net().to("cuda:0"), Net().to("cuda:0")
    b.load_state_dict(a.state_dict())
    x = torch.randn(4, 9, 12, device="cuda:0")
    ddp = FlatGradDDP(a, max_norm=0.0)
    assert ddp.sinks and sum(1 for e in ops._GRAD_SINKS.values() if e[3] == id(ddp)) == len(list(a.parameters()))
    for _ in range(3):                                               # stale values of earlier steps must not leak
        ddp.zero_grad()
        a(x).square().mean().backward()
        ddp.finish()
    assert not any(e[2] for k, e in ops._GRAD_SINKS.items() if e[3] == id(ddp) and e[0]() is not a.unused), "a GRU sink was not used"
    b(x).square().mean().backward()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if q.grad is None:
            assert float(p.grad.abs().max()) == 0.0, n
        else:
            assert torch.allclose(p.grad, q.grad, atol=1e-7), n
    # the user drops the flat views: the sinks stand down and autograd allocates ordinary grads
    for p in a.parameters():
        p.grad = None
    ops.arm_grad_sinks()
    a(x).square().mean().backward()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if q.grad is not None:
            assert p.grad is not None and torch.allclose(p.grad, q.grad, atol=1e-7), n
    ops.clear_grad_sinks()


def test_bench_two_ranks_on_one_gpu_gloo():
    """The N>1 bench path (torchrun launch, flat-buffer bucketed all-reduce, barrier/max timing, rank-0 JSON)
    on the single GPU of the test box: both ranks on cuda:0, gloo instead of RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, M3T_BENCH_BACKEND="gloo", M3T_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--frames", "40"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0
    assert d["cpu_baseline"] is None and d["scaling"] == "weak"
    assert d["allreduce"]["world_size"] == 2 and d["allreduce"]["ms_in_step"] is not None and "fallback" not in d


def test_shared_linear_weight_with_deferred_sink_write():
    """one nn.Linear weight used twice in a graph: the first backward node writes its gradient into the sink on the
    weight-gradient stream (joined only in finish()), the second returns a tensor that autograd ADDS onto the same slice on
    the main stream -- the two must be ordered (m3t.ops._Linear joins the stream when the sink is already taken).  Also two
    backward passes without zero_grad in between (gradient accumulation)."""
    from m3t import ops
    from m3t.ddp import FlatGradDDP

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(512, 512)

        def forward(self, x):
            h = ops.linear(x, self.lin.weight, self.lin.bias, 1)
            return ops.linear(h, self.lin.weight, self.lin.bias, 0)

    torch.manual_seed(2)
    a, b = Net().to("cuda:0"), Net().to("cuda:0")
    b.load_state_dict(a.state_dict())
    x = torch.randn(9600, 512, device="cuda:0")
    ddp = FlatGradDDP(a, max_norm=0.0)
    for _ in range(3):
        ddp.zero_grad()
        a(x).square().mean().backward()
        a(x).square().mean().backward()          # accumulation: second pass finds the sinks taken
        ddp.finish()
    for _ in range(2):
        b(x).square().mean().backward()
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=2e-5, atol=1e-7), (n, float((p.grad - q.grad).abs().max()))
    ops.clear_grad_sinks()


_DDP_CHILD = r"""
import os, sys
for p in (%r, %r, %r):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
from golden.recipe import fill_module
from m3t.ddp import FlatGradDDP, shard_indices
from m3t import ops
from models.rnn import GRU
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
torch.manual_seed(1000 + rank)                       # different initial weights per rank: construction must broadcast rank 0's
net = GRU(12, 16, 2, 3, 2).to("cuda:0")
if rank == 0:
    fill_module(net, 5)
ddp = FlatGradDDP(net, max_norm=0.05, overlap=False)      # the default schedule of the persistent-scan path: sinks + ONE all-reduce
assert ddp.sinks and not ddp.overlap and ddp.world == world
rs = np.random.RandomState(3)
x = torch.from_numpy(rs.standard_normal((8, 9, 12)).astype(np.float32)).to("cuda:0")
t = torch.from_numpy(rs.standard_normal((8, 9, 3)).astype(np.float32)).to("cuda:0")
idx = shard_indices(8, rank, world)
for _ in range(2):
    ddp.zero_grad()
    ((net(x[idx]) - t[idx]) ** 2).mean().backward()
    norm = ddp.finish()                                   # HIP finalize: 1/N + clip on the all-reduced flat buffer
torch.cuda.synchronize()
assert not any(e[2] for e in ops._GRAD_SINKS.values()), "a gradient sink was not used"
np.save(sys.argv[1] + ".%%d.npy" %% rank, torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy())
np.save(sys.argv[1] + ".norm%%d.npy" %% rank, norm.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
"""


def test_two_ranks_gru_sinks_hip_finalize_equal_mean_of_shard_gradients(tmp_path):
    """The real N > 1 combination on the one GPU of the test box: two processes (gloo, both on cuda:0, launch-per-step scans:
    two processes must not both run persistent scans on one device), GRU modules writing their weight gradients into gradient
    sinks, overlap=False (ONE all-reduce of the flat buffer after backward), the fused HIP finalize (1/N + clip), and replicas
    that seed differently (construction broadcasts rank 0's state).  Every rank must end with the clipped MEAN of the per-shard
    gradients -- what the reference's DDP optimises (train.py:35,40)."""
    import os
    import subprocess
    import sys
    import numpy as np
    from golden.recipe import fill_module
    from m3t.ddp import shard_indices
    from models.rnn import GRU
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _DDP_CHILD % (os.path.join(root, "m3f.pytorch_amd"), os.path.join(root, "tests"), root)
    script = tmp_path / "child.py"
    script.write_text(code)
    out = str(tmp_path / "g")
    env = dict(os.environ, M3T_SCAN_PERSIST="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2500:]
    g0, g1 = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    assert np.array_equal(g0, g1), "replicas diverged"
    # single-process reference: mean over ranks of the per-shard gradients, then the clip
    from m3t import ops
    ops.SCAN_PER_STEP[0] = True
    try:
        net = fill_module(GRU(12, 16, 2, 3, 2), 5).to("cuda:0")
        rs = np.random.RandomState(3)
        x = torch.from_numpy(rs.standard_normal((8, 9, 12)).astype(np.float32)).to("cuda:0")
        t = torch.from_numpy(rs.standard_normal((8, 9, 3)).astype(np.float32)).to("cuda:0")
        loss = sum(((net(x[shard_indices(8, r_, 2)]) - t[shard_indices(8, r_, 2)]) ** 2).mean() for r_ in range(2)) / 2
        loss.backward()
    finally:
        ops.SCAN_PER_STEP[0] = False
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double()
    norm = float(ref.norm())
    ref = (ref * min(1.0, 0.05 / (norm + 1e-6))).cpu().numpy()
    assert norm > 0.05, "the clip must be active in this case"
    assert abs(float(np.load(out + ".norm0.npy")[0]) - norm) <= 1e-5 * norm
    assert float(np.abs(g0 - ref).max()) <= 2e-6 * max(1.0, float(np.abs(ref).max())), float(np.abs(g0 - ref).max())


_FAULT_CHILD = r"""
import os, sys
for p in (%r, %r, %r):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
from golden.recipe import fill_module
from m3t.ddp import FlatGradDDP, shard_indices
from m3t.optim import FlatAdam
from m3t import ops, _lib
from models.rnn import GRU
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
net = fill_module(GRU(12, 16, 2, 3, 2), 5).to("cuda:0")
ddp = FlatGradDDP(net, max_norm=1.0, overlap=(os.environ["FAULT_OVERLAP"] == "1"), flatten_params=True)
opt = FlatAdam(ddp, lr=1e-2)
rs = np.random.RandomState(3)
x = torch.from_numpy(rs.standard_normal((8, 9, 12)).astype(np.float32)).to("cuda:0")
t = torch.from_numpy(rs.standard_normal((8, 9, 3)).astype(np.float32)).to("cuda:0")
idx = shard_indices(8, rank, world)

def step(inject=False):
    ddp.zero_grad()
    ((net(x[idx]) - t[idx]) ** 2).mean().backward()
    if inject:
        ops.inject_scan_error()              # "a scan of THIS rank died during backward" (stream order)
    norm = ddp.finish()
    opt.step()
    return norm

step()
torch.cuda.synchronize()
p1 = ddp.flat_params.clone()
raised, norm = False, None
try:
    norm = step(inject=(rank == 1))          # only rank 1 fails
    step()                                   # and one more step queued behind it on every rank
except _lib.M3THipError:
    raised = True
if not raised:
    try:
        ops.poll_scan_error(sync=True)
    except _lib.M3THipError:
        raised = True
torch.cuda.synchronize()
same = bool(torch.equal(p1, ddp.flat_params))
nan_norm = norm is None or not bool(torch.isfinite(norm).all())
print("RANK %%d raised=%%s params_unchanged=%%s nan_norm=%%s" %% (rank, raised, same, nan_norm), flush=True)
assert raised and same and nan_norm
dist.barrier()
step()                                       # every rank cleared its state when it raised: the job trains on
torch.cuda.synchronize()
ops.poll_scan_error()
assert not torch.equal(p1, ddp.flat_params)
np.save(sys.argv[1] + ".%%d.npy" %% rank, ddp.flat_params.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_one_ranks_dead_scan_stops_every_rank(tmp_path, overlap):
    """the all-reduce spreads a dead rank's garbage to every rank, but only the dead rank's guard used to fire.
    m3t_grad_poison / m3t_grad_dead_check carry the failure through the collective: EVERY rank's finalize sees a raised
    flag (gradients zeroed, norm NaN, fused Adam skipped), EVERY rank raises at its next poll, nobody is left waiting in a
    collective, and the replicas are still identical afterwards.  Two ranks on cuda:0 over gloo; the fault is injected by
    a kernel on rank 1 only (m3t_gru_inject_error)."""
    import os
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(_FAULT_CHILD % (os.path.join(root, "m3f.pytorch_amd"), os.path.join(root, "tests"), root))
    out = str(tmp_path / "p")
    env = dict(os.environ, M3T_SCAN_PERSIST="0", FAULT_OVERLAP=overlap)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29535", str(script), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2500:])
    assert "RANK 0 raised=True params_unchanged=True nan_norm=True" in r.stdout
    assert "RANK 1 raised=True params_unchanged=True nan_norm=True" in r.stdout
    assert np.array_equal(np.load(out + ".0.npy"), np.load(out + ".1.npy")), "replicas diverged after the fault"


_EARLY_CHILD = r"""
import os, sys
for p in (%r, %r, %r):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
from golden.recipe import fill_module
from m3t.ddp import FlatGradDDP, shard_indices
from models.rnn import GRU
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
class Two(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.enc = GRU(12, 16, 2, -1, 2)
        self.top = GRU(32, 16, 2, 3, 2)
    def forward(self, x):
        return self.top(self.enc(x))
net = fill_module(Two(), 5).to("cuda:0")
ddp = FlatGradDDP(net, bucket_order=[list(net.top.parameters()), list(net.enc.parameters())], max_norm=0.05, overlap=False)
armed = False
if os.environ["EARLY"] == "1":
    armed = ddp.early_bucket_after(net.top, resident_workgroups=192, n_cus=256, channels=32)
    assert armed
    assert not ddp.early_bucket_after(net.top, resident_workgroups=230, n_cus=256, channels=32)      # the invariant refuses: 230 + 32 > 256
    ddp._early = True
rs = np.random.RandomState(3)
x = torch.from_numpy(rs.standard_normal((8, 9, 12)).astype(np.float32)).to("cuda:0")
t = torch.from_numpy(rs.standard_normal((8, 9, 3)).astype(np.float32)).to("cuda:0")
idx = shard_indices(8, rank, world)
for _ in range(2):
    ddp.zero_grad()
    ((net(x[idx]) - t[idx]) ** 2).mean().backward()
    early_seen = getattr(ddp, "_early_handle", None) is not None
    norm = ddp.finish()
torch.cuda.synchronize()
print("RANK %%d armed=%%s early_collective_issued=%%s" %% (rank, armed, early_seen), flush=True)
np.save(sys.argv[1] + ".%%d.npy" %% rank, np.concatenate([ddp.flat.cpu().numpy(), norm.reshape(1).cpu().numpy()]))
dist.barrier()
dist.destroy_process_group()
"""


def test_early_bucket_gives_the_default_schedules_gradients(tmp_path):
    """the opt-in early bucket (FlatGradDDP.early_bucket_after / M3T_DDP_EARLY_BUCKET=1) -- bucket 0 all-reduced on a
    communication stream from a backward hook of its module, the rest of the buffer and the dead slot in finish() -- must give exactly
    the gradients (and clip norm) of the default ONE collective after backward, and its residency invariant (scan workgroups + RCCL
    channels <= CUs) must refuse to arm when it does not hold.  Two ranks on cuda:0 over gloo, launch-per-step scans."""
    import os
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(_EARLY_CHILD % (os.path.join(root, "m3f.pytorch_amd"), os.path.join(root, "tests"), root))
    res = {}
    for early, port in (("0", "29541"), ("1", "29542")):
        out = str(tmp_path / ("g" + early))
        env = dict(os.environ, M3T_SCAN_PERSIST="0", EARLY=early)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port, str(script), out]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2500:])
        if early == "1":
            assert "RANK 0 armed=True early_collective_issued=True" in r.stdout and "RANK 1 armed=True early_collective_issued=True" in r.stdout, r.stdout[-600:]
        res[early] = [np.load(out + ".%d.npy" % k) for k in (0, 1)]
    assert np.array_equal(res["0"][0], res["0"][1]) and np.array_equal(res["1"][0], res["1"][1]), "replicas diverged"
    assert np.array_equal(res["0"][0], res["1"][0]), float(np.abs(res["0"][0] - res["1"][0]).max())


_AGREED_CHILD = r"""
import os, sys, time
for p in (%r, %r, %r):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
from golden.recipe import fill_module
from m3t.ddp import FlatGradDDP, shard_indices
from m3t.optim import FlatAdam
from m3t import ops, _lib
from models.rnn import GRU
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
net = fill_module(GRU(12, 16, 2, 3, 2), 5).to("cuda:0")
ddp = FlatGradDDP(net, max_norm=1.0, overlap=False, flatten_params=True)
opt = FlatAdam(ddp, lr=1e-2)
rs = np.random.RandomState(3)
x = torch.from_numpy(rs.standard_normal((8, 9, 12)).astype(np.float32)).to("cuda:0")
t = torch.from_numpy(rs.standard_normal((8, 9, 3)).astype(np.float32)).to("cuda:0")
idx = shard_indices(8, rank, world)
raised_at, where = None, None
for k in range(6):
    try:
        ddp.zero_grad()
        if k == 2 and rank == 1:
            ops.inject_scan_error()          # a scan of THIS rank dies at the start of step 2 ...
            torch.cuda.synchronize()         # ... and this rank's HOST sees the flag at once, in the middle of the step
        if rank == 0:
            time.sleep(0.05 * (k %% 2))      # (the ranks' hosts are never in lockstep)
        where = "scans"
        loss = ((net(x[idx]) - t[idx]) ** 2).mean()      # scan calls behind the dead scan: must NOT raise on rank 1 alone
        loss.backward()
        where = "finish"
        ddp.finish()
        opt.step()
    except _lib.M3THipError:
        assert raised_at is None
        raised_at = (k, where)
print("RANK %%d raised_at=%%s" %% (rank, raised_at), flush=True)
ddp.agree_on_scan_error()                    # clean again on every rank
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_dead_scan_is_raised_at_the_same_step_on_every_rank(tmp_path):
    """the scan error flag is host-mapped and every rank's host sees it at a different point.  Raised from wherever
    a host happens to notice it (a scan call returning M3T_ESPIN mid-step), the failing rank leaves the step without issuing the
    step's collective and its peers wait there forever.  With several ranks nothing raises mid-step any more
    (m3t_gru_error_defer); the all-reduced dead slot of step k is read by every rank at the start of finish() of step k + 1, before
    that step's collective: both ranks raise at step 3, in finish(), although rank 1's host knew in the middle of step 2."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(_AGREED_CHILD % (os.path.join(root, "m3f.pytorch_amd"), os.path.join(root, "tests"), root))
    env = dict(os.environ, M3T_SCAN_PERSIST="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2500:])
    assert "RANK 0 raised_at=(3, 'finish')" in r.stdout and "RANK 1 raised_at=(3, 'finish')" in r.stdout, r.stdout[-800:]


def test_rccl_allreduce_in_stream_order_between_persistent_scans():
    """The 8-GPU default is persistent scans + RCCL (backend 'nccl') + ONE all-reduce after backward.  A 1-GPU box cannot
    host two RCCL ranks, but it can run that exact sequence in a 1-rank RCCL group: the real ncclAllReduce kernel on the
    real backend, queued between the persistent scans of consecutive steps (collective_when_alone=True), with the poison /
    dead-check kernels around it.  Results must equal the run without the collective, no scan may spin out, and the
    persistent path must really have run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
for p in (%r, %r, %r):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from m3t import ops, _lib
from m3t.workloads import AVFeatureGraph, make_c3_step
from golden.recipe import fill_module
sys.path.insert(0, %r)
from bench import synth_batch
B, T = 16, 48
res = []
for alone in (False, True):
    m = fill_module(AVFeatureGraph(128, 256, 512), 7).to("cuda:0")
    ddp, step = make_c3_step(m, synth_batch(B, T, 128, 256, torch.device("cuda", 0), 0), max_norm=1.0, collective_when_alone=alone)
    n0 = _lib.load().m3t_gru_persist_count()
    for _ in range(4):
        loss, _, y = step()
    torch.cuda.synchronize()
    ops.poll_scan_error()
    assert _lib.load().m3t_gru_persist_count() - n0 == 4 * 14, "the persistent scans did not run"
    res.append((float(loss), float(ddp.last_norm), ddp.flat.clone()))
    ddp.close()
assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and torch.equal(res[0][2], res[1][2]), (res[0][:2], res[1][:2])
print("RCCL-BETWEEN-SCANS-OK owner=%%d" %% ops.persist_owner(), flush=True)
dist.destroy_process_group()
""" % (os.path.join(root, "m3f.pytorch_amd"), os.path.join(root, "tests"), root, root)
    env = dict(os.environ, M3T_SCAN_LOCK="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2500:])
    assert "RCCL-BETWEEN-SCANS-OK" in r.stdout


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with NO torchrun and no WORLD_SIZE must start its two ranks itself (a child
    `python -m torch.distributed.run`, issued before the parent touches the GPU), each rank a supervisor + worker pair, and print
    ONE JSON line with n_gpus = 2 -- weak scaling (the default) and `--scaling strong` (the global batch split over the ranks).
    Both ranks on cuda:0 over gloo (M3T_BENCH_BACKEND / M3T_BENCH_ONE_DEVICE: the 1-GPU stand-in for the RCCL run)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(M3T_BENCH_BACKEND="gloo", M3T_BENCH_ONE_DEVICE="1", MASTER_PORT="29561")
    for extra, batch, per_gpu in (([], 4, 4), (["--scaling", "strong"], 4, 2)):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", str(batch),
               "--frames", "40"] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None
        assert d["scaling"] == ("strong" if extra else "weak")
        assert d["config"]["clips_per_gpu"] == per_gpu and d["config"]["global_batch"] == 2 * per_gpu
        assert d["allreduce"]["world_size"] == 2 and d["allreduce"]["ms_in_step"] is not None and "fallback" not in d


def test_bench_rejects_a_world_size_that_is_not_gpus():
    """--gpus must equal the number of ranks: a launcher that starts 2 ranks for `--gpus 1` exits non-zero on every rank"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, M3T_BENCH_BACKEND="gloo", M3T_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)


def test_bench_falls_back_to_per_step_scans_in_fresh_workers_when_a_scan_gives_up():
    """The 8-GPU default (persistent scans + RCCL) has never run on real multi-GPU hardware, so bench.py plans for its failure:
    a worker that sees M3T_ESPIN exits with a code its supervisor knows, every supervisor ends its own worker, and fresh workers
    run the launch-per-step scans with overlapped gradient buckets; the JSON line says so.  Here rank 1's first attempt gets an
    injected scan failure (M3T_BENCH_INJECT_FAULT, raised by a kernel exactly as a dying scan raises it)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(M3T_BENCH_BACKEND="gloo", M3T_BENCH_ONE_DEVICE="1", MASTER_PORT="29581", M3T_BENCH_INJECT_FAULT="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--frames", "40"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (lines, out.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "fallback" in d and "launch-per-step" in d["fallback"], d.get("fallback")
    assert "worker failure" in out.stderr
