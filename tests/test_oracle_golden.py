"""The C restatement (oracle/vv_oracle.c) against golden vectors produced by the REFERENCE's own
kernels (oracle/make_golden.py).  Bit-exact: both are plain IEEE arithmetic in the same order."""
import os

import numpy as np
import pytest

from oracle import oracle as O, cases
from oracle.make_golden import run_hardwall_massless


def _load(path):
    z = np.load(path)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in.")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out.")}
    return inp, out


def _bits_equal(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("case", sorted(cases.CASES))
def test_sequence_matches_reference_golden(golden_dir, case, prec):
    inp, gold = _load(os.path.join(golden_dir, f"{case}_{prec}.npz"))
    got = cases.run_sequence(O.Kernels("oracle", prec), inp)
    assert set(got) == set(gold)
    for k in sorted(gold):
        assert _bits_equal(got[k], gold[k]), f"{case}/{prec}: snapshot {k} differs from the reference"


@pytest.mark.parametrize("prec", O.PRECISIONS)
def test_hardwall_massless_parent_matches_reference_golden(golden_dir, prec):
    inp, gold = _load(os.path.join(golden_dir, f"hwmassless_{prec}.npz"))
    got = run_hardwall_massless(O.Kernels("oracle", prec), inp)
    moved = (got["hw.velm"] != inp["velm"]).any(1).sum()
    assert moved >= 4, "the case must actually exercise the massless-parent branch"
    for k in gold:
        assert _bits_equal(got[k], gold[k]), k


@pytest.mark.parametrize("prec", O.PRECISIONS)
def test_image_mirror_properties(golden_dir, prec):
    """bit-exact x,y copies and z_img + z_par == 2*mirror."""
    inp, gold = _load(os.path.join(golden_dir, f"edl_{prec}.npz"))
    ip = inp["image_pairs"]
    posq, corr = gold["images.posq"], gold["images.corr"]
    assert np.array_equal(posq[ip[:, 0], :2].view(np.uint8), posq[ip[:, 1], :2].view(np.uint8))
    mirror = float(inp["scalars"][6])
    z = posq[:, 2].astype(np.float64) + (corr[:, 2].astype(np.float64) if prec == "mixed" else 0.0)
    tol = 1e-12 if prec != "single" else 2e-6
    assert np.allclose(z[ip[:, 0]] + z[ip[:, 1]], 2 * mirror, rtol=0, atol=tol * max(1.0, abs(mirror)))
