"""Particle sharding over the GPUs of one node (one process per GPU).

Shards are contiguous particle ranges cut on molecule boundaries, so every Drude pair
and every COM group stays on one GPU.  The only cross-GPU exchange is an element-wise int64 sum of the
<= 3 fixed-point accumulators after each reduction phase (include/vvhip.h: vvhip_step_middle_phase),
issued through torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_bounds(system, world_size: int) -> List[Tuple[int, int]]:
    """Cut [0, N) into `world_size` contiguous ranges of nearly equal size whose ends fall between molecules.

    Requires every molecule to occupy a contiguous index range (true for all shipped models); image-charge
    systems are single-GPU in the reference's configs (BASELINE.json C5) and are rejected by the plan."""
    mol = np.asarray(system.mol_id)
    n = mol.shape[0]
    if world_size == 1:
        return [(0, n)]
    change = np.nonzero(np.diff(mol) != 0)[0] + 1            # indices where a new molecule starts
    if len(np.unique(mol)) != len(change) + 1:
        raise ValueError("molecules are not contiguous in particle index: cannot shard by index range")
    starts = np.concatenate([[0], change, [n]])
    cuts = [0]
    for r in range(1, world_size):
        target = r * n / world_size
        k = int(np.argmin(np.abs(starts - target)))
        cuts.append(int(max(starts[k], cuts[-1])))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world_size)]


class _DevInt64:
    """Zero-copy view of `count` int64 values at a raw device pointer, via the CUDA array interface."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


class ShardedStepper:
    """Drives vvhip_step_middle_phase on this rank's shard and sums the fixed-point accumulators over ranks
    after every reduction phase.  With backend nccl (= RCCL) the all-reduce runs on the GPU over xGMI on the
    plan's stream; with gloo (CPU tests, or several ranks sharing one GPU) the 8..24 bytes are staged through
    the host.  int64 sums are associative, so every rank sees the same bits whatever the rank count."""

    def __init__(self, ctx, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import vvhip as H
        self.ctx, self.group, self.dist, self.H, self.C = ctx, group, dist, H, C
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.nphase = H.lib.vvhip_step_middle_phases(ctx.plan)
        self._tensors = {}

    def _acc(self, ph):
        """The accumulators alternate between two copies from step to step (thermostat double-buffering), so ask every time."""
        C, H = self.C, self.H
        p, n = C.c_void_p(), C.c_int32()
        H.check(H.lib.vvhip_accumulators(self.ctx.plan, ph, C.byref(p), C.byref(n)), self.ctx.plan)
        return p.value, n.value

    def _all_reduce(self, ph):
        import numpy as np
        import torch
        ptr, n = self._acc(ph)
        if self.backend == "nccl":
            if ptr not in self._tensors:
                self._tensors[ptr] = torch.as_tensor(_DevInt64(ptr, n), device="cuda")
            self.dist.all_reduce(self._tensors[ptr], op=self.dist.ReduceOp.SUM, group=self.group)
        else:
            H = self.H
            host = np.zeros(n, dtype=np.int64)
            H.check(H.lib.vvhip_synchronize(self.ctx.plan), self.ctx.plan)
            H.check(H.lib.vvhip_memcpy_d2h(host.ctypes.data, ptr, host.nbytes), what="d2h")
            t = torch.from_numpy(host)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            H.check(H.lib.vvhip_memcpy_h2d(ptr, host.ctypes.data, host.nbytes), what="h2d")

    def step(self, steps: int = 1):
        H, ctx = self.H, self.ctx
        for _ in range(steps):
            ctx.calcForces()
            for ph in range(self.nphase):
                H.check(H.lib.vvhip_step_middle_phase(ctx.plan, ph, 0), ctx.plan)
                if ph < self.nphase - 1 and self.backend is not None:
                    self._all_reduce(ph)
