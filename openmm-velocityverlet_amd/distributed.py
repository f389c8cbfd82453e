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


def first_contact_report(local: dict, group=None):
    """What every rank of a multi-GPU run knows about its own set-up, gathered on all ranks and checked for agreement BEFORE the first
    timed region (round-4 review, "first real multi-GPU contact": the in-core RCCL all-reduce at world > 1 and the hipIpc mailbox
    ACROSS devices have never run in the build environment, so the first run that does must at least fail with a readable report).
    `local` = {"device": int, "device_name": str, "peer_access": [bool per rank's device] or None, "rccl_ranks": int, "exchange": chosen
    mechanism, "mailbox_trial": str or None, "world": int, ...}.  Returns (ok, text): ok = every rank chose the same exchange, no two ranks
    sit on one device (unless they say they share it on purpose), every communicator counts the whole world, every device reaches every
    other where the mailbox is used; text = one line per rank + the complaints.  Works on any backend (all_gather_object)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    every = [None] * world
    if world > 1:
        dist.all_gather_object(every, local, group=group)
    else:
        every = [local]
    return check_first_contact(every)


def check_first_contact(every: List[dict]):
    """The agreement rules of first_contact_report on the gathered per-rank records (host-only: unit-tested without a process group)."""
    lines, complaints = [], []
    world = len(every)
    for r, rec in enumerate(every):
        pa = rec.get("peer_access")
        lines.append(f"rank {r}: device {rec.get('device')} ({rec.get('device_name', '?')}), exchange {rec.get('exchange')}, "
                     f"RCCL communicator ranks {rec.get('rccl_ranks')}, peer access {''.join('1' if x else '0' for x in pa) if pa is not None else 'not checked'}, "
                     f"mailbox trial: {rec.get('mailbox_trial')}")
    chosen = {rec.get("exchange") for rec in every}
    if len(chosen) != 1:
        complaints.append(f"the ranks chose different exchange mechanisms: {sorted(map(str, chosen))}")
    if any(rec.get("world") != world for rec in every):
        complaints.append(f"ranks disagree about the world size: {[rec.get('world') for rec in every]} (gathered {world} records)")
    # (a device is (host, local index): on several nodes the local indices repeat)
    devices = [(rec.get("host"), rec.get("device")) for rec in every]
    if len(set(devices)) != world and not all(rec.get("share_device") for rec in every):
        complaints.append(f"two ranks on one device without --share-device: devices {devices}")
    exch = next(iter(chosen)) if len(chosen) == 1 else None
    if exch in ("eager", "graph"):
        bad = [r for r, rec in enumerate(every) if rec.get("rccl_ranks") != world]
        if bad:
            complaints.append(f"RCCL communicator of rank(s) {bad} does not count {world} ranks")
    if exch == "mailbox" and world > 1 and not all(rec.get("share_device") for rec in every):
        if len({rec.get("host") for rec in every}) > 1:
            complaints.append("the mailbox maps the peers' boxes through hipIpc: ranks of ONE host only")
        bad = [r for r, rec in enumerate(every) if rec.get("peer_access") is not None and not all(rec["peer_access"])]
        if bad:
            complaints.append(f"hipDeviceCanAccessPeer denies a pair of devices on rank(s) {bad}: the mailbox stores into the peers' memory")
    text = "\n".join(lines + ["first contact: " + ("all ranks agree" if not complaints else "; ".join(complaints))])
    return not complaints, text


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
