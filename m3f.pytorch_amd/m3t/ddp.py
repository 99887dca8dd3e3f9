"""Clip-level data parallelism: one process per GPU, RCCL all-reduce of ONE flat fp32 gradient
buffer over xGMI, then the global-norm clip -- what the reference gets from Lightning's
`distributed_backend='ddp'` + `gradient_clip_val=1.0` (reference train.py:32-41) and
`DistributedSampler` (reference models/model.py:416-418).

Design: parameters' .grad tensors are VIEWS into one contiguous buffer, so
the ~100 small gradient tensors of the AV graph cost one (or a few, bucketed) collectives
instead of one each.  Two schedules:
  * overlap=False (default while the persistent GRU scans are enabled): ONE all-reduce of the whole flat buffer after
    backward.  A persistent scan launch needs every workgroup of its grid resident; an RCCL kernel that holds a few
    CUs while it waits for its peers on other GPUs could keep part of a scan grid off the chip on one GPU while the
    mirror-image happens on another -- a cross-GPU cycle.  Keeping collectives and scans disjoint in time rules it
    out by construction (106 MB at xGMI rates is ~1 ms per step, less than the persistent scans save).
  * overlap=True (launch-per-step scans, M3T_SCAN_PERSIST=0): buckets follow the order in which backward finishes
    sub-modules, and each bucket's all-reduce is issued from a post-accumulate hook so it overlaps the rest of backward.
The 1/world scaling and the clip coefficient are applied by one fused HIP pass (m3t_grad_norm_scale).  There is no
other collective on the path (no SyncBN, as the reference).

A dead persistent scan on ONE rank stops EVERY rank: the buffer carries one extra slot (`dead`) through the same
all-reduce; m3t_grad_poison sets it (and NaN in flat[0]) before the collective when this rank's sticky scan-error flag is
up, m3t_grad_dead_check raises the flag on every rank after it, so every rank's finalize zeroes its gradients and every fused
optimizer step skips itself (include/m3t_hip.h, error model).  The HOST side is rank-agreed too (round 4): with
world > 1 nothing raises from a scan call or a bare poll (each rank's host sees its flag at a different point, and a rank
that raised mid-step would leave its peers in the step's collective); instead every finish() copies the ALL-REDUCED dead slot
to pinned host memory, and the next finish() -- before it issues its own collective -- reads it: the value is the same on every
rank, so every rank raises at the same step, with no collective of that step issued.  Points every rank reaches anyway
(validation, checkpoints, the end of fit) use agree_on_scan_error(): one small host-synchronous all-reduce of the local flags.
"""
import os
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """DistributedSampler(shuffle=False) semantics: pad to a multiple of world_size by wrapping,
    then rank r takes indices r, r+world, ..."""
    idx = list(range(n_items))
    total = ((n_items + world_size - 1) // world_size) * world_size
    idx += idx[: total - n_items]
    return idx[rank:total:world_size]


class FlatGradDDP:
    def __init__(self, module, bucket_order=None, max_norm=1.0, process_group=None, finalize=None,
                 flatten_params=False, overlap=None, collective_when_alone=False):
        """bucket_order: list of lists of parameters, in the order backward completes them
        (default: one bucket per top-level child, reversed registration order).
        finalize(flat, world_size, max_norm) -> norm tensor; default = fused HIP kernel."""
        self.module = module
        self.overlap = (os.environ.get("M3T_SCAN_PERSIST", "1") == "0") if overlap is None else bool(overlap)
        self.max_norm = max_norm
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # collective_when_alone: issue the all-reduce even in a 1-rank group (a 1-GPU box can then run the real RCCL call,
        # on the real backend, in stream order between persistent scans -- tests/test_gpu_ddp.py)
        self._alone_collective = bool(collective_when_alone) and self.world == 1 and dist.is_available() and dist.is_initialized()
        if self.world > 1:
            # replicas must start identical: torch DDP (what Lightning's 'ddp' wraps the reference in, train.py:40)
            # broadcasts rank 0's parameters and buffers at construction; so does this.  Frozen parameters and buffers
            # (BatchNorm running statistics) included.
            self.broadcast_state()
        params = [p for p in module.parameters() if p.requires_grad]
        if bucket_order is not None:
            # parameters frozen before construction (freeze_enc, reference models/model.py:377-386) drop out of the buckets
            bucket_order = [[p for p in b if p.requires_grad] for b in bucket_order]
            bucket_order = [b for b in bucket_order if b]
        if bucket_order is None:
            seen, bucket_order = set(), []
            for child in reversed(list(module.children())):
                b = [p for p in child.parameters() if p.requires_grad and id(p) not in seen]
                seen.update(id(p) for p in b)
                if b:
                    bucket_order.append(b)
            rest = [p for p in params if id(p) not in seen]
            if rest:
                bucket_order.append(rest)
        self.buckets = bucket_order
        assert sum(p.numel() for b in self.buckets for p in b) == sum(p.numel() for p in params), \
            "buckets must cover every trainable parameter exactly once"
        dev = params[0].device
        # every parameter's slice starts on a 128-byte boundary (a 9-element bias would otherwise misalign everything behind
        # it: kernels that write gradients in place -- gradient sinks -- take their float4 paths only on aligned outputs);
        # the padding stays zero, so norms and all-reduces are unaffected
        pad = lambda k: (k + 31) // 32 * 32
        self.offsets, off = {}, 0
        for b in self.buckets:
            for p in b:
                self.offsets[id(p)] = off
                off = pad(off + p.numel())
        n = off
        # [ gradients (n floats, 128-B aligned slices) | dead slot (1 float + padding) ]: ONE buffer, so that the slot rides
        # in the same all-reduce; self.flat is the gradient part (what norms, optimizers and tests see)
        self._buf = torch.zeros(n + 32, dtype=torch.float32, device=dev)
        self.flat = self._buf[:n]
        self.dead = self._buf[n:n + 1]
        self.ranges = []
        self._pending = []
        for bi, b in enumerate(self.buckets):
            start = self.offsets[id(b[0])]
            for p in b:
                o = self.offsets[id(p)]
                p.grad = self.flat[o:o + p.numel()].view_as(p)
            self.ranges.append((start, pad(self.offsets[id(b[-1])] + b[-1].numel())))
        # optionally the parameters themselves become views of one flat buffer (same order as the gradients),
        # so an optimizer step is ONE kernel over (flat_params, flat)
        self.flat_params = None
        if flatten_params:
            self.flat_params = torch.zeros(n, dtype=torch.float32, device=dev)
            with torch.no_grad():
                for b in self.buckets:
                    for p in b:
                        o = self.offsets[id(p)]
                        view = self.flat_params[o:o + p.numel()].view_as(p)
                        view.copy_(p.data)
                        p.data = view
        self._left = [0] * len(self.buckets)
        self._handles = []
        if self.world > 1:
            for bi, b in enumerate(self.buckets):
                for p in b:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))
        from . import ops
        if finalize is None:
            finalize = ops.grad_norm_scale_
        self._finalize = finalize
        self.last_norm = None
        # rank-agreed failure (module docstring): the all-reduced dead slot of the last finish() on its way to the host
        self._agreed = (self.world > 1 or self._alone_collective) and dev.type == "cuda"
        self._dead_host = self._dead_ev = None
        self._dead_pending = False
        if self._agreed:
            ops.defer_scan_errors_acquire()       # refcounted: the switch is process-global, several instances may be alive
            self._dead_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._dead_ev = torch.cuda.Event()
        self.ar_events = None          # bench.py: a list here collects (start, end) HIP events around the in-step all-reduce
        # gradient sinks (m3t.ops): backward writes weight gradients straight into the flat buffer instead of handing
        # them to autograd's AccumulateGrad.  Not with overlap=True: the bucket all-reduces hang on post-accumulate hooks.
        self.sinks = (not self.overlap or self.world == 1) and os.environ.get("M3T_GRAD_SINKS", "1") != "0"
        # the registry is keyed by parameter and tagged with its owner: a second FlatGradDDP in the process (EMA teacher,
        # another trainer) neither disarms nor clears this one's sinks
        if self.sinks:
            for b in self.buckets:
                for p in b:
                    ops.register_grad_sink(p, p.grad, owner=self)

    def close(self):
        """drop this instance's gradient sinks (the parameters keep their .grad views)"""
        from . import ops
        ops.clear_grad_sinks(self)
        ops.drop_weight_amax(self)
        if getattr(self, "_agreed", False):
            self._agreed = False
            try:
                ops.defer_scan_errors_release()   # deferral goes off only with the LAST live owner
            except Exception:  # noqa: BLE001  (library already gone at interpreter shutdown)
                pass
        h = getattr(self, "_early_hook", None)
        if h is not None:
            h.remove()
            self._early_hook = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def broadcast_state(self, src=0):
        """rank `src`'s parameters and buffers to every rank (construction, and after a checkpoint is loaded on one rank)"""
        if self.world <= 1:
            return
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                dist.broadcast(t.data, src=dist.get_global_rank(self.pg, src) if self.pg is not None else src, group=self.pg)

    def _make_hook(self, bi):
        def hook(_p):
            self._left[bi] -= 1
            if self._left[bi] == 0 and self.overlap:
                s, e = self.ranges[bi]
                if self.flat.is_cuda:
                    # the bucket's gradients may have been written on the library's other streams (weight-gradient streams; the side stream
                    # that runs the second conv tower / the audio stack and -- autograd keeps a node's backward on its forward's stream --
                    # their backward): the collective is ordered behind the CURRENT stream only
                    from . import ops
                    cur = ops.cur_stream(self.flat.device)
                    for st in ops.live_streams(self.flat.device):
                        cur.wait_stream(st)
                self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        return hook

    def zero_grad(self):
        self._buf.zero_()                     # one 106 MB memset (~30 us): parameters nothing writes this step read zero
        if self.sinks:
            from . import ops
            ops.arm_grad_sinks(self)          # the first gradient of a parameter this step overwrites its slice in place
        self._left = [len(b) for b in self.buckets]
        self._handles = []
        if self.flat.is_cuda:
            # the weights are final here (the optimizer ran after the previous finish()): their magnitudes for this step's fp16x3
            # contractions, every matrix in one pass (m3t.ops.measure_weight_amax) instead of one measuring launch per forward call
            from . import ops
            ops.measure_weight_amax([p for b in self.buckets for p in b], owner=self)

    def _raise_agreed(self, why):
        from . import _lib
        for h in self._handles:                  # (overlap: every rank issued the same bucket collectives during backward)
            h.wait()
        self._handles = []
        if getattr(self, "_early_handle", None) is not None:      # the step's early bucket is in flight on its own stream: join it
            self._early_handle.wait()
            torch.cuda.current_stream().wait_stream(self._early_stream)
            self._early_handle = None
        self._dead_pending = False
        _lib._recover_scan_error()               # synchronise the device, then clear the sticky state
        raise _lib.M3THipError("a persistent GRU scan gave up waiting for a peer workgroup on at least one rank (M3T_ESPIN, %s): "
                               "its gradients were discarded and every optimizer step queued behind it was skipped on every rank; "
                               "all ranks raise here, at the same step" % why)

    def _check_agreed(self):
        """the previous finish()'s all-reduced dead slot: non-zero on every rank or on none"""
        if self._dead_pending:
            self._dead_ev.synchronize()          # a step old: no wait on the healthy path
            self._dead_pending = False
            if float(self._dead_host[0]) != 0.0:
                self._raise_agreed("reported by the gradient all-reduce of the previous step")

    def agree_on_scan_error(self):
        """A point every rank reaches (validation, checkpoint, end of fit): wait for the device, all-reduce the local scan error
        flags (host-synchronous), raise on every rank if any rank's is up.  One rank: an ordinary synchronised poll."""
        from . import ops, _lib
        if not self._agreed or self.world <= 1:
            ops.poll_scan_error(sync=True, force=True)
            return
        torch.cuda.synchronize()
        self._dead_pending = False
        t = torch.tensor([1.0 if _lib.load().m3t_gru_poll_error() else 0.0], dtype=torch.float32)
        if dist.get_backend(self.pg) == "nccl":
            t = t.to(self.flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        if float(t.item()) != 0.0:
            self._raise_agreed("agreed on by all ranks at a synchronisation point")

    # ---- opt-in: the first bucket's all-reduce early, on a stream of its own (M3T_DDP_EARLY_BUCKET=1) ---------------------------
    def early_bucket_after(self, module, resident_workgroups, n_cus, channels=None):
        """VERDICT r3 item 8 (unmeasured: no multi-GPU node; opt-in so that the first 8-GPU run can A/B it against the default ONE
        collective after backward).  Bucket 0 (the fusion GRU in the C3 graph: 8.4 M of 26.4 M parameters, final ~6 ms before the
        step ends) is all-reduced on a communication stream as soon as `module`'s backward has been issued, beside the rest of
        backward.  The argument against overlapping a collective with persistent scans (module docstring) is residency: an RCCL kernel
        parked on some CUs must never keep part of a scan grid off the chip.  Hence the CHECKED invariant: the scan workgroups that
        can be resident at once during the window (`resident_workgroups`, from m3t_gru_scan_workgroups: C3 128 + 64) plus one CU per
        RCCL channel (`channels`, default NCCL_MAX_NCHANNELS or 32) must fit the device's CUs -- every scan grid then becomes resident
        whatever the collective holds.  Returns True when armed."""
        if self.world <= 1 or self.overlap or not self.flat.is_cuda:
            return False
        if channels is None:
            channels = int(os.environ.get("NCCL_MAX_NCHANNELS", "32"))
        if resident_workgroups + channels > n_cus:
            import warnings
            warnings.warn("M3T_DDP_EARLY_BUCKET refused: %d scan workgroups + %d RCCL channels do not fit %d CUs"
                          % (resident_workgroups, channels, n_cus))
            return False
        if getattr(self, "_early_hook", None) is not None:      # re-arm: one hook, one collective per slice
            self._early_hook.remove()
            self._early_hook = None
        self._early_stream = torch.cuda.Stream(device=self.flat.device)
        self._early_handle = None
        s0, e0 = self.ranges[0]
        assert s0 == 0, "the early bucket must be the head of the flat buffer (the rest, dead slot included, stays ONE range)"

        def hook(_m, _gi, _go):
            from . import ops
            cur = ops.cur_stream()
            self._early_stream.wait_stream(cur)
            for st in ops.live_streams(self.flat.device):       # the bucket's weight gradients are written on these streams
                self._early_stream.wait_stream(st)
            with torch.cuda.stream(self._early_stream):
                self._early_handle = dist.all_reduce(self.flat[s0:e0], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

        self._early_hook = module.register_full_backward_hook(hook)
        self._early = True
        return True

    def finish(self):
        """Call after backward: waits for the bucket all-reduces, then averages + clips in place."""
        from . import ops
        if self._agreed:
            self._check_agreed()                  # BEFORE this step's collective: either every rank raises here or none does
        if self.flat.is_cuda:
            # weight-gradient GEMMs that write straight into the flat buffer, and whatever else of backward ran on the library's side stream (the
            # second conv tower's backward: its weight gradients go into gradient sinks, so no AccumulateGrad node makes autograd join that
            # stream at the end of backward): ONE wait per live stream (round 6: join_wgrad's two came on top of these three)
            cur = ops.cur_stream(self.flat.device)
            for st in ops.live_streams(self.flat.device):
                cur.wait_stream(st)
            ops.wgrad_joined(self.flat.device)
        else:
            ops.join_wgrad(self.flat.device)
        hip = self.flat.is_cuda
        timed = self.ar_events is not None and hip
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if self._alone_collective:
            ops.grad_poison_(self.flat, self.dead)
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.pg)
            ops.grad_dead_check_(self.dead)
        early = self.world > 1 and not self.overlap and getattr(self, "_early", False) and self._early_handle is not None
        if early:
            # bucket 0 -- which holds flat[0] -- may still be in flight on the communication stream: join it BEFORE the poison kernel
            # writes flat[0] on this stream (a data race otherwise); the dead slot carries the failure either way
            self._early_handle.wait()
            ops.cur_stream().wait_stream(self._early_stream)
        if self.world > 1 and hip:
            ops.grad_poison_(self.flat, self.dead)          # this rank's dead scan -> NaN in flat[0], 1 in the dead slot
        if early:
            # bucket 0 left early (early_bucket_after); the rest of the buffer and the dead slot now, then join
            dist.all_reduce(self._buf[self.ranges[0][1]:], op=dist.ReduceOp.SUM, group=self.pg)
            self._early_handle = None
        elif self.world > 1 and not self.overlap:
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.pg)      # gradients + dead slot: ONE collective
        elif self.world > 1:
            for bi, left in enumerate(self._left):      # parameters that received no gradient this step
                if left > 0:
                    s, e = self.ranges[bi]
                    self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            for h in self._handles:
                h.wait()
            # (the bucket that holds flat[0] may have left before the scan died: the dead slot travels on its own here)
            dist.all_reduce(self._buf[self.flat.numel():], op=dist.ReduceOp.SUM, group=self.pg)
        if self.world > 1 and hip:
            ops.grad_dead_check_(self.dead)                 # any rank dead -> this rank's flag up: finalize zeroes, norm = NaN
        if self._agreed:
            self._dead_host.copy_(self.dead, non_blocking=True)      # the all-reduced slot: the same value on every rank
            self._dead_ev.record()
            self._dead_pending = True
        if timed:
            ev1.record()
            self.ar_events.append((ev0, ev1))
        self.last_norm = self._finalize(self.flat, self.world, self.max_norm)
        # a persistent GRU scan that died leaves garbage gradients.  Two nets: the finalize kernel reads the scan error
        # word on the device, in stream order (gradients zeroed, norm = NaN, the fused optimizer steps skip on a
        # non-finite norm) -- and this poll of the same word raises as soon as the failure is visible to the host
        # (no synchronisation here: at the latest on the next step's finish()).  Several ranks: silent here -- the next
        # finish() raises on every rank from the all-reduced slot (_check_agreed).
        ops.poll_scan_error()
        ops.drop_weight_amax(self)                # the optimizer is about to change the weights: this step's magnitude table dies here
        return self.last_norm
