"""oracle/make_golden.py -- TEST INFRASTRUCTURE.  Generates tests/golden/*.npz.

Run in THIS container only (it needs oracle/_ref, i.e. the reference sources under /root/reference):
    python oracle/make_golden.py
Each file holds the seeded inputs of one case (oracle/cases.py) and the outputs of the REFERENCE's own
kernels (platforms/cuda/src/kernels/*.cu compiled for the CPU by `make -C oracle ref`, one thread) after
every kernel of the sequence.  The files are data only: inputs and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O, cases  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def hardwall_massless_case(prec, seed=5):
    """Drude pairs whose parent is massless (velm.w == 0): the branch at K/middle.cu:151-173."""
    rng = np.random.default_rng(seed)
    M, R = O.MIXED[prec], O.REAL[prec]
    npairs = 12
    n = 2 * npairs
    velm = np.zeros((n, 4), dtype=M)
    velm[:, :3] = rng.standard_normal((n, 3)) * 0.3
    velm[0::2, 3] = 1.0 / 0.4            # Drude
    velm[1::2, 3] = 0.0                  # massless parent
    velm[1::2, :3] = 0.0
    posq = np.zeros((n, 4), dtype=R)
    posq[1::2, :3] = rng.uniform(0, 3, (npairs, 3))
    posq[0::2, :3] = posq[1::2, :3] + rng.normal(0, 0.02, (npairs, 3)).astype(R)
    posq[:, 3] = rng.uniform(-1, 1, n)
    corr = (rng.uniform(-1e-8, 1e-8, (n, 4))).astype(R) if prec == "mixed" else np.zeros((n, 4), dtype=R)
    corr[:, 3] = 0
    pairs = np.stack([np.arange(0, n, 2), np.arange(1, n, 2)], 1).astype(np.int32)
    return dict(velm=velm, posq=posq, posq_corr=corr, drude_pairs=pairs)


def run_hardwall_massless(K, inp):
    velm, posq, corr = inp["velm"].copy(), inp["posq"].copy(), inp["posq_corr"].copy()
    K.hard_wall(posq, corr, velm, inp["drude_pairs"], 0.001, 0.02, np.sqrt(O.BOLTZ * 1.0))
    return {"hw.velm": velm, "hw.posq": posq, "hw.corr": corr}


def main():
    if not O.have_ref():
        O.build("ref")
    os.makedirs(GOLD, exist_ok=True)
    for prec in O.PRECISIONS:
        K = O.Kernels("ref", prec)
        for name, fn in cases.CASES.items():
            inp = fn(prec)
            out = cases.run_sequence(K, inp)
            blob = {f"in.{k}": v for k, v in inp.items()}
            blob.update({f"out.{k}": v for k, v in out.items()})
            np.savez_compressed(os.path.join(GOLD, f"{name}_{prec}.npz"), **blob)
            print(f"{name}_{prec}: {len(out)} snapshots, n={inp['velm'].shape[0]}")
        inp = hardwall_massless_case(prec)
        out = run_hardwall_massless(K, inp)
        blob = {f"in.{k}": v for k, v in inp.items()}
        blob.update({f"out.{k}": v for k, v in out.items()})
        np.savez_compressed(os.path.join(GOLD, f"hwmassless_{prec}.npz"), **blob)
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    main()
