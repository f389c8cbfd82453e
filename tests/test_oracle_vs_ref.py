"""Direct oracle-vs-reference comparison on further seeds.  Needs oracle/_ref (built by `make -C oracle ref`
where /root/reference exists, i.e. in the build container; the .so files travel to the GPU box)."""
import numpy as np
import pytest

from oracle import oracle as O, cases
from oracle.make_golden import hardwall_massless_case, run_hardwall_massless

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (reference sources absent)")


@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("seed", [101, 202])
def test_all_kernels_bit_exact(prec, seed):
    for name, fn in cases.CASES.items():
        inp = fn(prec, seed=seed)
        a = cases.run_sequence(O.Kernels("ref", prec), inp)
        b = cases.run_sequence(O.Kernels("oracle", prec), inp)
        for k in a:
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), f"{name}/{prec}/seed{seed}: {k}"


@pytest.mark.parametrize("prec", O.PRECISIONS)
def test_hardwall_massless_bit_exact(prec):
    inp = hardwall_massless_case(prec, seed=77)
    a = run_hardwall_massless(O.Kernels("ref", prec), inp)
    b = run_hardwall_massless(O.Kernels("oracle", prec), inp)
    for k in a:
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
