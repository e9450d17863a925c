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
        return out
