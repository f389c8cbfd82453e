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
