"""Row-sharded FITC evaluation across the GPUs of one node: one process per GPU, training points
split into contiguous row blocks, sum-all-reduces (RCCL over xGMI via torch.distributed) of the m x m
accumulations: two per gradient evaluation, one per evidence-only evaluation.

The reference has no counterpart (single process, lib/fitc_gp.ml).  Every shard runs the same
m x m factorisations redundantly on the reduced buffers, so all ranks return identical results.

The exchange buffers carry the symmetric m x m accumulations as their upper 128-tiles only
(gprhip_ar1_len / gprhip_ar2_len), i.e. half the square.  The library enqueues on its own HIP stream; the
collective is ordered against it with events (no host-side drain): library stream -> event -> torch's
current stream (RCCL) -> event -> library stream.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .problem import F64, Problem


def shard_rows(n_total, rank, world):
    """Rows [lo, hi) owned by `rank`: contiguous blocks, sizes differ by at most one."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class ShardedProblem:
    """backend: object with the staged interface of gpr_amd.Problem (eval_pass1/eval_pass2/eval_finish/
    ar1_len/ar2_len/sync); defaults to a device Problem on `device`.  `group` is the process group.
    `timing=True` brackets every collective with events; `last_comm_ms` then holds their durations."""

    def __init__(self, cov_kind, n_total, D, d, m, rank=None, world=None, device=0, chunk_rows=0,
                 backend=None, group=None, buffer_device=None, precision=F64, timing=False):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_total = int(n_total)
        self.lo, self.hi = shard_rows(n_total, self.rank, self.world)
        if self.hi - self.lo < 1:
            raise ValueError("ShardedProblem: rank %d would own no training points" % self.rank)
        self.local = backend if backend is not None else Problem(cov_kind, self.hi - self.lo, D, d, m,
                                                                 device=device, chunk_rows=chunk_rows,
                                                                 precision=precision)
        if buffer_device is None:
            buffer_device = torch.device("cuda", device) if backend is None else torch.device("cpu")
        self.ar1 = torch.zeros(self.local.ar1_len(), dtype=torch.float64, device=buffer_device)
        self.ar2 = torch.zeros(self.local.ar2_len(), dtype=torch.float64, device=buffer_device)
        self._cuda = buffer_device.type == "cuda"
        # the collective runs whenever a process group exists (also a one-rank RCCL group: the smoke path)
        self._collective = dist.is_available() and dist.is_initialized()
        self._lib_stream = None
        if self._cuda and hasattr(self.local, "stream") and self._collective and dist.get_backend(group) == "nccl":
            self._lib_stream = torch.cuda.ExternalStream(self.local.stream(), device=buffer_device)
        self.timing = bool(timing)
        self.last_comm_ms = []
        self.collectives = 0

    @property
    def n_local(self):
        return self.hi - self.lo

    def set_inputs(self, inputs_local):
        self.local.set_inputs(inputs_local)

    def set_targets(self, targets_local):
        self.local.set_targets(targets_local)

    def _allreduce(self, buf):
        if not self._collective:
            return
        self.collectives += 1
        if self._lib_stream is None:  # gloo stand-ins (validation): host-side ordering -- drain, reduce, drain
            self.local.sync()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            if self._cuda:
                torch.cuda.synchronize(buf.device)
            return
        cur = torch.cuda.current_stream(buf.device)
        done = torch.cuda.Event()
        done.record(self._lib_stream)
        cur.wait_event(done)  # the library's pass has produced the buffer
        if self.timing:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(cur)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        if self.timing:
            t1.record(cur)
            self._pending.append((t0, t1))
        reduced = torch.cuda.Event()
        reduced.record(cur)
        self._lib_stream.wait_event(reduced)  # the library's next pass reads the reduced buffer

    def eval(self, **hypers):
        want_grad = hypers.get("want_grad", True)
        self._pending = []
        self.local.eval_pass1(self.ar1.data_ptr(), self.n_total, **hypers)
        self._allreduce(self.ar1)
        self.local.eval_pass2(self.ar1.data_ptr(), self.ar2.data_ptr())
        if want_grad:  # an evidence-only evaluation carries nothing in the second buffer
            self._allreduce(self.ar2)
        ev = self.local.eval_finish(self.ar2.data_ptr())  # drains the library's stream
        if self.timing:
            if self._cuda:
                torch.cuda.current_stream(self.ar1.device).synchronize()
            self.last_comm_ms = [t0.elapsed_time(t1) for t0, t1 in self._pending]
        return ev

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()
