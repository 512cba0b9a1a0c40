"""Row-sharded FITC evaluation across the GPUs of one node: one process per GPU, training points
split into contiguous row blocks, two sum-all-reduces (RCCL over xGMI via torch.distributed) of the
m x m accumulations per evaluation.

The reference has no counterpart (single process, lib/fitc_gp.ml).  Every shard runs the same
m x m factorisations redundantly on the reduced buffers, so all ranks return identical results.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .problem import Problem


def shard_rows(n_total, rank, world):
    """Rows [lo, hi) owned by `rank`: contiguous blocks, sizes differ by at most one."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class ShardedProblem:
    """backend: object with the staged interface of gpr_amd.Problem (eval_pass1/eval_pass2/eval_finish/
    ar1_len/ar2_len/sync); defaults to a device Problem on `device`.  `group` is the process group."""

    def __init__(self, cov_kind, n_total, D, d, m, rank=None, world=None, device=0, chunk_rows=0,
                 backend=None, group=None, buffer_device=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_total = int(n_total)
        self.lo, self.hi = shard_rows(n_total, self.rank, self.world)
        if self.hi - self.lo < 1:
            raise ValueError("ShardedProblem: rank %d would own no training points" % self.rank)
        self.local = backend if backend is not None else Problem(cov_kind, self.hi - self.lo, D, d, m,
                                                                 device=device, chunk_rows=chunk_rows)
        if buffer_device is None:
            buffer_device = torch.device("cuda", device) if backend is None else torch.device("cpu")
        self.ar1 = torch.zeros(self.local.ar1_len(), dtype=torch.float64, device=buffer_device)
        self.ar2 = torch.zeros(self.local.ar2_len(), dtype=torch.float64, device=buffer_device)
        self._cuda = buffer_device.type == "cuda"

    @property
    def n_local(self):
        return self.hi - self.lo

    def set_inputs(self, inputs_local):
        self.local.set_inputs(inputs_local)

    def set_targets(self, targets_local):
        self.local.set_targets(targets_local)

    def _allreduce(self, buf):
        # the library enqueues on its own HIP stream: drain it before RCCL reads the buffer, and
        # drain RCCL's stream before the library reads the reduced buffer
        self.local.sync()
        if self.world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        if self._cuda:
            torch.cuda.synchronize(buf.device)

    def eval(self, **hypers):
        self.local.eval_pass1(self.ar1.data_ptr(), self.n_total, **hypers)
        self._allreduce(self.ar1)
        self.local.eval_pass2(self.ar1.data_ptr(), self.ar2.data_ptr())
        self._allreduce(self.ar2)
        return self.local.eval_finish(self.ar2.data_ptr())

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()
