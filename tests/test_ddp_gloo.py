"""N>1 data-parallel path on CPU: world_size-2 gloo processes exercise the flat-buffer bucketing,
the hook-driven async all-reduce and the DistributedSampler-style sharding of m3t.ddp.  The fused
HIP norm/scale kernel is swapped for an equivalent torch finalize here (CPU has no HIP); the kernel
itself is covered by tests/test_gpu_ddp.py on one GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from m3t.ddp import FlatGradDDP, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def torch_finalize(flat, world, max_norm):
    flat.mul_(1.0 / world)
    norm = flat.norm()
    if max_norm > 0:
        flat.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
    return norm


def _net():
    torch.manual_seed(7)
    return nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 8), nn.Tanh(), nn.Linear(8, 2))


def _data():
    g = torch.Generator().manual_seed(11)
    return torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)


def _worker(rank, world, port, max_norm, overlap, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _net()
    ddp = FlatGradDDP(net, max_norm=max_norm, finalize=torch_finalize, overlap=overlap)
    assert ddp.world == world and len(ddp.buckets) == 3 and ddp.overlap == overlap
    x, y = _data()
    idx = shard_indices(x.shape[0], rank, world)
    for _ in range(2):                      # two steps: zero_grad must reset the hook counters
        ddp.zero_grad()
        ((net(x[idx]) - y[idx]) ** 2).mean().backward()
        norm = ddp.finish()
    for p in net.parameters():              # .grad must still be views of the flat buffer
        assert p.grad.data_ptr() >= ddp.flat.data_ptr()
        assert p.grad.data_ptr() < ddp.flat.data_ptr() + ddp.flat.numel() * 4
    packed = torch.cat([p.grad.reshape(-1) for b in ddp.buckets for p in b])      # the slices without their alignment padding
    gap = torch.ones(ddp.flat.numel(), dtype=torch.bool)
    for b in ddp.buckets:
        for p in b:
            gap[ddp.offsets[id(p)]:ddp.offsets[id(p)] + p.numel()] = False
    assert not gap.any() or float(ddp.flat[gap].abs().max()) == 0.0                 # the alignment padding stays zero
    out_q.put((rank, packed.tolist(), float(norm)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("max_norm,overlap", [(0.0, True), (0.05, True), (0.05, False)])
def test_two_rank_gradients_equal_mean_of_shard_gradients(max_norm, overlap):
    """both schedules: bucketed all-reduces overlapped with backward, and one all-reduce after backward (the default
    while the persistent scans are enabled)"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, max_norm, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r, torch.tensor(f), n) for r, f, n in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: mean over ranks of the per-shard mean losses (what DDP optimises)
    net = _net()
    x, y = _data()
    loss = sum(((net(x[shard_indices(8, r, world)]) - y[shard_indices(8, r, world)]) ** 2).mean() for r in range(world)) / world
    loss.backward()
    order = [p for child in reversed(list(net.children())) for p in child.parameters()]
    ref = torch.cat([p.grad.reshape(-1) for p in order])
    norm = ref.norm()
    if max_norm > 0:
        ref = ref * torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    for rank, flat, n in res:
        assert torch.allclose(flat, ref, atol=1e-6), rank
        assert abs(n - float(norm)) < 1e-5
    assert torch.equal(res[0][1], res[1][1])       # replicas stay identical


def test_shard_indices_distributed_sampler_semantics():
    assert shard_indices(8, 0, 2) == [0, 2, 4, 6] and shard_indices(8, 1, 2) == [1, 3, 5, 7]
    assert shard_indices(5, 0, 2) == [0, 2, 4] and shard_indices(5, 1, 2) == [1, 3, 0]     # wraps to pad
    from torch.utils.data.distributed import DistributedSampler
    ds = list(range(11))
    for r in range(4):
        assert shard_indices(11, r, 4) == list(DistributedSampler(ds, num_replicas=4, rank=r, shuffle=False))


def test_single_process_flat_views_and_accumulation():
    net = _net()
    ddp = FlatGradDDP(net, max_norm=0.0, finalize=torch_finalize)
    x, y = _data()
    ddp.zero_grad()
    ((net(x) - y) ** 2).mean().backward()
    ddp.finish()
    ref = _net()
    ((ref(x) - y) ** 2).mean().backward()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a.grad, b.grad, atol=1e-7)
    assert abs(float(ddp.flat.norm()) - float(torch.cat([p.grad.reshape(-1) for p in ref.parameters()]).norm())) < 1e-6


def _bcast_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # replicas that would silently diverge
    net = nn.Sequential(nn.Linear(6, 16), nn.BatchNorm1d(16), nn.Tanh(), nn.Linear(16, 2))
    with torch.no_grad():
        net[1].running_mean.add_(float(rank + 1))
    net[3].weight.requires_grad = False                 # frozen parameters are synchronised too
    ddp = FlatGradDDP(net, max_norm=0.0, finalize=torch_finalize, flatten_params=True)
    state = torch.cat([t.detach().reshape(-1).float() for t in list(net.parameters()) + list(net.buffers())])
    # the parameters are views of the flat buffer AND were broadcast
    assert all(p.data_ptr() >= ddp.flat_params.data_ptr() for p in net.parameters() if p.requires_grad)
    out_q.put((rank, state.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_construction_broadcasts_rank0_parameters_and_buffers():
    """torch DDP (the reference's Lightning 'ddp' backend, train.py:40) broadcasts rank 0's state at construction; ranks
    that seed differently must still start as identical replicas"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    torch.manual_seed(100)
    ref = nn.Sequential(nn.Linear(6, 16), nn.BatchNorm1d(16), nn.Tanh(), nn.Linear(16, 2))
    assert torch.allclose(torch.tensor(res[1][:6 * 16]), ref[0].weight.detach().reshape(-1))     # rank 0's values won


def test_trainer_epoch_schedulers_and_freeze_enc_host_logic(tmp_path):
    """m3t.trainer on CPU (no optimizer step is taken: the fused kernels need a GPU): freeze_enc happens before the flat
    buffers are built, plateau / exp schedules follow torch's own scheduler classes as the reference configures them
    (models/model.py:399-407), best-val_loss bookkeeping."""
    import argparse
    from models.model import AffWild2VA
    from m3t.trainer import Trainer
    ns = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False)).parse_args([])
    ns.modality, ns.fusion_type, ns.loss, ns.freeze_enc, ns.window = "audiovisual", "attention", "ccc_mtl", True, 4
    ns.checkpoint_path = None
    model = AffWild2VA(ns)
    tr = Trainer.from_hparams(model, ns)
    live = set(id(p) for m in (model.fusion, model.proj_v, model.att_fuse) for p in m.parameters())
    assert all(p.requires_grad == (id(p) in live) for p in model.parameters())
    assert sum(p.numel() for b in tr.ddp.buckets for p in b) == sum(p.numel() for p in model.parameters() if id(p) in live)
    assert tr.ddp.flat.numel() < 0.25 * sum(p.numel() for p in model.parameters())      # the frozen encoders are outside
    # plateau (the default): x0.5 after 3 epochs without improvement beyond the best, floor 1e-6
    ref_opt = torch.optim.Adam([nn.Parameter(torch.zeros(1))], lr=ns.learning_rate, weight_decay=1e-4)
    ref_s = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, factor=ns.decay_factor, patience=3, min_lr=1e-6)
    losses = [1.0, 0.9, 0.95, 0.93, 0.92, 0.91, 0.905, 0.95, 0.96, 0.97, 0.98, 0.99, 1.0, 1.0, 1.0, 1.0]
    best = float("inf")
    for v in losses:
        improved = tr.end_epoch(v)
        ref_s.step(v)
        assert improved == (v < best)
        best = min(best, v)
        assert tr.lr == ref_opt.param_groups[0]["lr"]
    assert tr.lr < ns.learning_rate and tr.best_val_loss == min(losses) and tr.epoch == len(losses)
    # exp: lr * decay_factor^epoch
    ns2 = argparse.Namespace(**vars(ns))
    ns2.scheduler, ns2.freeze_enc = "exp", False
    tr2 = Trainer.from_hparams(AffWild2VA(ns2), ns2)
    for e in range(1, 5):
        tr2.end_epoch(None)
        assert abs(tr2.lr - ns.learning_rate * ns.decay_factor ** e) < 1e-12


def test_two_flat_ddp_instances_keep_their_own_gradient_sinks():
    """the gradient-sink registry is per owner: constructing, arming, zeroing or closing a second FlatGradDDP in the process
    (an EMA teacher, a second trainer) must not disarm or clear the first one's sinks"""
    from m3t import ops
    a, b = _net(), _net()
    da = FlatGradDDP(a, max_norm=0.0, finalize=torch_finalize)
    n_a = len(list(a.parameters()))
    mine = lambda d: [e for e in ops._GRAD_SINKS.values() if e[3] == id(d)]
    assert len(mine(da)) == n_a
    db = FlatGradDDP(b, max_norm=0.0, finalize=torch_finalize)       # used to clear every sink of the process
    assert len(mine(da)) == n_a and len(mine(db)) == n_a
    da.zero_grad()
    assert all(e[2] for e in mine(da)) and not any(e[2] for e in mine(db))
    db.zero_grad()
    assert all(e[2] for e in mine(da)) and all(e[2] for e in mine(db))
    p0 = next(a.parameters())
    assert ops._take_sink(p0) is not None and ops._take_sink(p0) is None      # first gradient of the step only
    db.close()
    assert len(mine(da)) == n_a and not mine(db)
    del da
    import gc
    gc.collect()
    assert not any(e[0]() is p0 for e in ops._GRAD_SINKS.values())         # a collected owner takes its sinks with it


def _trainer_worker(rank, world, port, ckdir, out_q):
    import argparse
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from models.model import AffWild2VA
    from m3t.trainer import Trainer
    ns = AffWild2VA.add_model_specific_args(argparse.ArgumentParser(add_help=False)).parse_args([])
    ns.modality, ns.loss, ns.checkpoint_path = "audio", "ccc", ckdir
    torch.manual_seed(rank)
    tr = Trainer.from_hparams(AffWild2VA(ns), ns)
    lrs = []
    # per-rank validation losses that disagree about "improved" and about the plateau: the mean must decide, on every rank
    per_rank = [[1.0, 0.8, 1.2, 1.2, 1.2, 1.2, 1.2], [1.0, 1.0, 0.2, 1.2, 1.2, 1.2, 1.2]][rank]
    improved = []
    for v in per_rank:
        improved.append(tr.end_epoch(v))
        lrs.append(tr.lr)
    ck = torch.load(os.path.join(ckdir, "best.ckpt"), map_location="cpu")       # behind save_checkpoint's barrier: complete
    stray = [f for f in os.listdir(ckdir) if f != "best.ckpt"]
    out_q.put((rank, improved, lrs, tr.best_val_loss, ck["best_val_loss"], ck["epoch"], stray))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_under_ddp_means_val_loss_and_writes_checkpoints_from_rank0(tmp_path):
    """Lightning's ModelCheckpoint saves from rank 0 only and the reference's plateau scheduler sees one val_loss; here every
    rank validates its own shard, so end_epoch all-reduces (mean) the val_loss before the scheduler and the best check, rank 0
    writes best.ckpt through a temporary file + os.replace, and the others wait at a barrier."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] == [True, True, True, False, False, False, False]      # means 1.0, 0.9, 0.7, 1.2 ...
    assert res[0][1] == res[1][1] and res[0][1][-1] < res[0][1][0]                      # same lr trace, plateau fired
    assert res[0][2] == res[1][2] == 0.7 and res[0][3] == 0.7 and res[0][4] == 3
    assert res[0][5] == [] and res[1][5] == []                                          # no torn / temporary files left
