"""Data-parallel gradient synchronisation over RCCL/xGMI (one process per GPU; torch.distributed backend
"nccl" IS RCCL on ROCm; "gloo" for the CPU tests and the one-GPU rehearsal).

The reference is single-process (no distributed code); it emulates a large batch with gradient
accumulation (utils/training.py:49-56) and computes BatchNorm statistics per micro-batch, so per-replica
BatchNorm + one gradient all-reduce per optimizer step is the faithful data-parallel extension.

Mechanism ("gradients live in their bucket"): parameters are grouped, in reverse registration order (~ the order
backward produces them: output, up4 .. up1, down5 .. down1), into size-capped buckets, each backed by ONE flat
buffer that lives as long as the synchroniser.  While armed, the weight-gradient kernels write their result straight
into the parameter's slice of that buffer (`ops.grad_destination`), autograd adopts the slice as `.grad`, and a
post-accumulate-grad hook counts arrivals; a bucket whose gradients are all there is all-reduced IN PLACE,
asynchronously on a side stream, so RCCL traffic (75 % of the bytes live in up1/down5, ready mid-backward)
overlaps the remaining high-resolution backward kernels.  Gradients that arrive somewhere else (BatchNorm vectors,
the head, accumulated micro-batches) are moved into their slice by one fused copy per bucket.  `sync()` (right before
optimizer.step()) flushes stragglers and makes the compute stream wait: `.grad` then IS the averaged slice -- no
flatten / un-flatten passes over the 124 MB of gradients.  xGMI is point-to-point (7 links x ~153 GB/s per GPU):
buckets are kept large (default 32 MiB) so each collective is bandwidth- not latency-bound.
"""
import torch
import torch.distributed as dist

from . import ops


class GradSync:
    def __init__(self, module, bucket_mb: float = 32.0, group=None, overlap: bool = True, broadcast: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.overlap = overlap
        self.module = module
        if self.world > 1 and broadcast:
            # replicas must start identical: rank 0's parameters and buffers win (torch DDP does the same)
            self.broadcast_parameters(module)
            self.broadcast_buffers(module)
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()
        cap = int(bucket_mb * (1 << 20))
        self.buckets, cur, size = [], [], 0
        for p in params:
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        # one flat buffer per bucket; slices start on 16-byte boundaries (the kernels store 16 bytes per lane)
        self._flat, self._slot = [], {}
        for bi, b in enumerate(self.buckets):
            al = max(1, 16 // b[0].element_size())
            off = 0
            for p in b:
                self._slot[p] = (bi, off, p.numel())
                off += (p.numel() + al - 1) // al * al
            self._flat.append(torch.zeros((off,), dtype=b[0].dtype, device=b[0].device))
        self._by_ptr = {p.data_ptr(): p for p in params}
        self._armed = False
        self._count = [0] * len(self.buckets)
        self._need = [len(b) for b in self.buckets]
        self._pending = {}           # bucket index -> work handle (or None)
        self._stream = None
        self._hooks = []
        self._next = 0               # buckets [0, _next) have been launched in this window (launches are strictly in order)
        self.timing = False          # True: record HIP events per bucket launch / completion (bench.py, rank 0)
        self._t0 = self._t1 = None
        self._trace = []
        self.collectives = 0         # all-reduces issued so far (tests: un-armed micro-steps must not communicate)
        self.direct_grads = 0        # gradients found already in their bucket slice when their bucket was launched
        if self.world > 1:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- helpers --------------------------------------------------------------------------------
    def _view(self, p):
        """A fresh dense view (the parameter's shape) of its slice of the bucket buffer."""
        bi, off, n = self._slot[p]
        return self._flat[bi][off:off + n].view(p.shape)

    def _in_place(self, p):
        bi, off, n = self._slot[p]
        g = p.grad
        f = self._flat[bi]
        return (g is not None and g.dtype == f.dtype and g.is_contiguous()
                and g.data_ptr() == f.data_ptr() + off * f.element_size())

    # -- protocol -------------------------------------------------------------------------------
    def arm(self):
        """Call before the backward of the micro-batch that ends an accumulation window."""
        self._armed = True
        self._next = 0
        self._t0 = self._t1 = None
        self._trace = []
        self._count = [0] * len(self.buckets)
        self._need = [sum(1 for p in b if p.requires_grad) for b in self.buckets]   # parameters frozen since __init__
        if self.world > 1:
            ops.grad_destination_begin(self._destination)

    def _destination(self, param):
        """ops.grad_destination callback: a FRESH view of the parameter's bucket slice (autograd adopts a gradient
        tensor nobody else references without copying it), or None when the slice cannot take this gradient."""
        if param not in self._slot:                      # a saved tensor may come back as another Python object
            param = self._by_ptr.get(param.data_ptr())
            if param is None:
                return None
        if param.grad is not None or not param.is_contiguous():
            return None
        return self._view(param)

    def _on_grad(self, p):
        if not self._armed:
            return
        bi = self._slot[p][0]
        self._count[bi] += 1
        if not self.overlap:
            return
        # Collectives are matched across ranks by ISSUE ORDER, so buckets are launched strictly in index order: bucket k
        # goes out only once buckets 0..k-1 have gone (a parameter that receives no gradient on one rank only -- a
        # data-dependent branch, an unused head -- would otherwise make that rank issue its all-reduces in another order
        # than its peers: a hang, or silently mixed buckets).  A bucket whose count never fills waits for sync(), which
        # launches the rest in the same index order on every rank.
        while self._next < len(self.buckets) and self._count[self._next] == self._need[self._next]:
            self._launch(self._next)
            self._next += 1

    # -- optional event trace of the overlap (bench.py: per-bucket launch times relative to the start of backward) -------
    def backward_begin(self):
        if self.timing and self._flat and self._flat[0].is_cuda:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record()

    def backward_end(self):
        if self.timing and self._t0 is not None:
            self._t1 = torch.cuda.Event(enable_timing=True)
            self._t1.record()

    def overlap_trace(self):
        """After sync() + a device synchronisation: per bucket, when its all-reduce was enqueued behind the backward
        kernels (compute stream) and when it had finished (side stream), in ms from backward_begin()."""
        if self._t0 is None:
            return None
        out = {"backward_ms": round(self._t0.elapsed_time(self._t1), 3) if self._t1 is not None else None, "buckets": []}
        for bi, nbytes, ev_l, ev_d in self._trace:
            out["buckets"].append({"bucket": bi, "mbytes": round(nbytes / 1e6, 2),
                                   "launched_at_ms": round(self._t0.elapsed_time(ev_l), 3),
                                   "done_at_ms": round(self._t0.elapsed_time(ev_d), 3)})
        return out

    def _launch(self, bi):
        flat = self._flat[bi]
        move_dst, move_src, zero = [], [], []
        for p in self.buckets[bi]:
            if p.grad is None:
                zero.append(self._view(p))               # unused on this rank: contributes zeros
            elif not self._in_place(p):
                move_dst.append(self._view(p))
                move_src.append(p.grad)
            else:
                self.direct_grads += 1
        side = flat.is_cuda and self.overlap
        ev_l = ev_d = None
        if self.timing and self._t0 is not None and flat.is_cuda:
            ev_l = torch.cuda.Event(enable_timing=True)
            ev_l.record()                                 # compute stream: every gradient of the bucket exists here
        if side:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
        ctx = torch.cuda.stream(self._stream) if side else _null()
        with ctx:
            if zero:
                torch._foreach_zero_(zero)
            if move_dst:
                torch._foreach_copy_(move_dst, move_src)
                if side:
                    for g in move_src:
                        g.record_stream(self._stream)
            avg = self.backend == "nccl"                  # RCCL averages in the collective; gloo has no AVG
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group,
                                   async_op=True)
            if ev_l is not None:
                if not side:
                    work.wait()
                ev_d = torch.cuda.Event(enable_timing=True)
                ev_d.record()                             # the stream the collective was enqueued on
                self._trace.append((bi, flat.numel() * flat.element_size(), ev_l, ev_d))
        self.collectives += 1
        for p, v in zip([q for q in self.buckets[bi] if q.grad is not None and not self._in_place(q)], move_dst):
            p.grad = v                                    # .grad now lives in the bucket
        self._pending[bi] = (work, avg)

    def sync(self):
        """Average gradients across ranks; returns when .grad holds the averaged values (stream-ordered)."""
        ops.grad_destination_end()
        if self.world == 1:
            self._armed = False
            return
        for bi in range(self._next if self._armed else 0, len(self.buckets)):   # the rest, in index order on every rank
            if bi not in self._pending:
                self._launch(bi)
        self._next = 0
        for bi, (work, avg) in sorted(self._pending.items()):
            work.wait()                                   # makes the current stream wait for the collective
        if self._stream is not None:
            for f in self._flat:
                if f.is_cuda:
                    torch.cuda.current_stream(f.device).wait_stream(self._stream)
                    break
        for bi, (work, avg) in self._pending.items():
            if not avg:
                self._flat[bi].div_(self.world)
        # a parameter without a local gradient still receives the other ranks' average: replicas stay identical
        for b in self.buckets:
            for p in b:
                if p.grad is None and p.requires_grad:
                    p.grad = self._view(p)
        self._pending.clear()
        self._armed = False

    def broadcast_parameters(self, module, src: int = 0):
        if self.world == 1:
            return
        with torch.no_grad():
            for p in module.parameters():
                dist.broadcast(p.data, src=src, group=self.group)
        ops.invalidate_packed_weights()

    def broadcast_buffers(self, module, src: int = 0):
        """BatchNorm running statistics are per replica during training (like the reference's per-micro-batch
        statistics); broadcast rank `src`'s before evaluation / checkpointing."""
        if self.world == 1:
            return
        for b in module.buffers():
            dist.broadcast(b, src=src, group=self.group)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        ops.grad_destination_end()


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False
