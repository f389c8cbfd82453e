"""propagateNHChain (openmmapi/src/VVIntegrator.cpp:340-376): C restatement vs an independent pure-Python
statement, plus analytic identities.  NB this routine is NOT pinned by reference code (needs OpenMM headers)."""
import numpy as np
import pytest

from oracle import oracle as O


def _state(nc, rng, hot=True):
    eta = rng.normal(0, 0.1, nc)
    eta_dot = np.concatenate([rng.normal(0, 2.0, nc) if hot else np.zeros(nc), [0.0]])   # last entry stays 0 (quirk Q10)
    eta_dd = rng.normal(0, 1.0, nc)
    return eta, eta_dot, eta_dd


@pytest.mark.parametrize("nc", [1, 2, 3, 5])
@pytest.mark.parametrize("loops", [1, 3])
def test_c_matches_python(nc, loops):
    rng = np.random.default_rng(nc * 10 + loops)
    dof, T = 165000.0, 333.0
    kbt = O.BOLTZ * T
    mass = np.array([dof * kbt / 100.0] + [kbt / 100.0] * (nc - 1))
    for trial in range(20):
        eta, ed, edd = _state(nc, rng)
        ke2 = dof * kbt * rng.uniform(0.8, 1.2)
        a = [x.copy() for x in (eta, ed, edd)]
        b = [x.copy() for x in (eta, ed, edd)]
        fa = O.propagate_nh_chain(a[0], a[1], a[2], mass, ke2, dof * kbt, T, 0.001, loops)
        fb = O.propagate_nh_chain_py(b[0], b[1], b[2], mass, ke2, dof * kbt, T, 0.001, loops)
        assert fa == fb
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert a[1][-1] == 0.0


def test_equilibrium_is_a_fixed_point():
    """2KE == dof*kT and eta_dot == 0 => factor == 1 and eta_dot[0] stays 0."""
    nc, dof, T = 3, 1000.0, 300.0
    kbt = O.BOLTZ * T
    mass = np.array([dof * kbt / 100.0, kbt / 100.0, kbt / 100.0])
    eta, ed, edd = np.zeros(nc), np.zeros(nc + 1), np.zeros(nc)
    f = O.propagate_nh_chain(eta, ed, edd, mass, dof * kbt, dof * kbt, T, 0.001)
    assert f == 1.0 and ed[0] == 0.0 and eta[0] == 0.0
    # the upper chain links feel -kT/Q and start moving: that is the reference's behaviour too
    assert ed[1] < 0


def test_hot_system_is_cooled():
    nc, dof, T = 3, 1000.0, 300.0
    kbt = O.BOLTZ * T
    mass = np.array([dof * kbt / 100.0, kbt / 100.0, kbt / 100.0])
    eta, ed, edd = np.zeros(nc), np.zeros(nc + 1), np.zeros(nc)
    ke2 = 1.5 * dof * kbt
    factors = []
    for _ in range(50):
        f = O.propagate_nh_chain(eta, ed, edd, mass, ke2, dof * kbt, T, 0.001)
        ke2 *= f * f
        factors.append(f)
    assert factors[0] < 1.0 and ke2 < 1.5 * dof * kbt


def test_middle_scheme_advances_half_step():
    """Quirk Q1: one call advances the thermostat by dt/2 (dt2 = stepSize/loops/2)."""
    nc = 1
    mass = np.array([10.0])
    eta, ed, edd = np.zeros(1), np.array([0.3, 0.0]), np.zeros(1)
    O.propagate_nh_chain(eta, ed, edd, mass, 5.0, 5.0, 300.0, 0.002)
    assert abs(eta[0] - 0.3 * 0.001) < 1e-12
