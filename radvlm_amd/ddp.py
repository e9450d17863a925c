"""Data-parallel gradient synchronisation over a flat gradient buffer (one process per GPU, RCCL over xGMI).

The partition is by samples (disjoint per rank, llava_trainer.py:129-149 semantics); the only exchange step is a
sum-all-reduce of the trainable gradients per optimizer step.  Buckets are contiguous
slices of the flat buffer handed over back-to-front as backward finalises them; each is all-reduced in place on a
side stream (GPU) so that it overlaps the rest of backward.  Averaging is folded into the loss scale
(1/world), i.e. mean over ranks of per-rank mean losses, as the reference's DDP run would compute.
On CPU tensors (gloo, tests) the same object works synchronously.
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, flat_grads, process_group=None, min_bucket_elems=0, force=False):
        self.flat = flat_grads
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        self.on_gpu = flat_grads.is_cuda
        # force: issue the per-bucket collectives even in a group of one rank -- the single-GPU rehearsal of the RCCL path (same
        # stream / event ordering, same launches next to backward's GEMMs; a one-rank all-reduce leaves the data unchanged)
        self.active = self.world > 1 or bool(force)
        self.stream = torch.cuda.Stream(device=flat_grads.device) if (self.on_gpu and self.active) else None
        self.pending = []
        self.min_bucket = min_bucket_elems
        self._lo = self._hi = None
        self.launched = []  # (start, end) of every collective issued since the last finish(); for tests/telemetry
        self.timing = None  # bench: a list collects (event before, event after) of every finish() on the compute stream
        self.last_buckets = []

    def bucket_done(self, start, end):
        """Elements [start, end) are final. Adjacent ready ranges are coalesced until >= min_bucket_elems."""
        if not self.active:
            return
        if self._lo is None:
            self._lo, self._hi = start, end
        elif end == self._lo:
            self._lo = start
        elif start == self._hi:
            self._hi = end
        else:
            self._launch()
            self._lo, self._hi = start, end
        if self._hi - self._lo >= self.min_bucket:
            self._launch()

    def _launch(self):
        if self._lo is None:
            return
        s, e = self._lo, self._hi
        self._lo = self._hi = None
        self.launched.append((s, e))
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self.pending.append(dist.all_reduce(self.flat[s:e], group=self.pg, async_op=True))
        else:
            dist.all_reduce(self.flat[s:e], group=self.pg)

    def finish(self):
        """Block the compute stream until every bucket is reduced."""
        self._launch()
        ev = None
        if self.timing is not None and self.on_gpu:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        if ev is not None:      # elapsed = how long the compute stream sat waiting for the reduced gradients (exposed communication)
            ev[1].record()
            self.timing.append(ev)
        out, self.launched = self.launched, []
        self.last_buckets = out        # telemetry: the collectives of the step that just ended (bench.py reports their count / sizes)
        return out


# ---------------------------------------------------------------------------------------------- replica consistency
def _checksum(t, chunk=1 << 26):
    """Two 63-bit integers of a tensor's BITS (sum and index-weighted sum of its 16-bit words, in chunks: no 8x temporary of a
    13 GB buffer).  Different values, a shifted slice or two swapped elements change at least one of them."""
    flat = t.detach().reshape(-1)
    if flat.element_size() % 2:
        flat = flat.view(torch.uint8).to(torch.int16)
    else:
        flat = flat.view(torch.int16)
    s0 = torch.zeros((), dtype=torch.int64, device=flat.device)
    s1 = torch.zeros((), dtype=torch.int64, device=flat.device)
    for a in range(0, flat.numel(), chunk):
        w = flat[a:a + chunk].to(torch.int64)
        s0 += w.sum()
        s1 += (w * (torch.arange(a, a + w.numel(), device=flat.device, dtype=torch.int64) % 65521 + 1)).sum()
    return torch.stack((s0 & 0x7FFFFFFFFFFFFFFF, s1 & 0x7FFFFFFFFFFFFFFF))


def assert_replicas_equal(named_tensors, process_group=None, where=""):
    """Data parallelism keeps one model per rank and never re-synchronises parameters: a rank that loaded another
    checkpoint, took another optimizer step or was hit by a faulty collective would train on silently.  All-reduce (MIN and MAX) the
    checksums of every named buffer and raise on the first one whose minimum and maximum differ.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return []
    names = [n for n, t in named_tensors if t is not None]
    sums = torch.stack([_checksum(t) for n, t in named_tensors if t is not None])          # [n, 2]
    lo, hi = sums.clone(), sums.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=process_group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=process_group)
    bad = [names[i] for i in range(len(names)) if not torch.equal(lo[i], hi[i])]
    if bad:
        raise RuntimeError(f"data-parallel replicas diverged ({where}): {bad} differ between ranks "
                           f"(rank {dist.get_rank(process_group)} of {dist.get_world_size(process_group)})")
    return names


def broadcast_from_rank0(named_tensors, process_group=None):
    """What DistributedDataParallel does when it wraps a model: rank 0's buffers become everyone's."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
    for _, t in named_tensors:
        if t is not None:
            dist.broadcast(t, src=src, group=process_group)
