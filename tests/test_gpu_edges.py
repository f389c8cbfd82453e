"""-m gpu: edge cases of the fused path against the oracle -- chain lengths that take the in-kernel chain (1..4) and the
stand-alone chain launch (8), several chain loops per step, systems without any NH particle, a single molecule, ragged
particle counts, parameter changes between steps (the reference re-reads its getters at every call), thermostat
checkpoint round trip (an extension: the reference loses the chain state on restart)."""
import ctypes as C
import importlib

import numpy as np
import pytest

from oracle import oracle as O

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, systems = pkg.vvhip, pkg.integrator, pkg.systems
pytestmark = pytest.mark.gpu


def _run_pair(spec, nsteps, middle=True, chains=3, loops=1, maxd=0.02, cos=0.0, prec="mixed", T=333.0):
    p = O.Params(temperature=T, drude_temperature=1.0, max_drude_distance=maxd, cos_acceleration=cos, use_middle_scheme=middle,
                 num_chains=chains, loops_per_step=loops)
    osys = O.OracleSystem(spec, p, prec, force_mode=1)
    it = I.VVIntegrator(T, 10.0, 1.0, 40.0, 0.001, chains, loops)
    it.setMaxDrudeDistance(maxd)
    it.setCosAcceleration(cos)
    it.setUseMiddleScheme(middle)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether")
    osys.step(nsteps)
    it.step(nsteps)
    return osys, ctx, it


def _assert_close(osys, ctx, tol=1e-9):
    x_o, x_g = osys.positions(), ctx.getPositions()
    v_o, v_g = osys.velm[:, :3].astype(np.float64), ctx.getVelocities()
    m = osys.velm[:, 3] != 0
    ex = np.abs(x_g - x_o).max() / max(np.abs(x_o).max(), 1e-30)
    ev = np.abs(v_g[m] - v_o[m]).max() / max(np.abs(v_o[m]).max(), 1e-30)
    assert ex < tol and ev < tol, (ex, ev)


@pytest.mark.parametrize("chains", [1, 2, 4, 5, 8])
@pytest.mark.parametrize("loops", [1, 3])
@pytest.mark.parametrize("middle", [True, False])
def test_chain_length_and_loops(chains, loops, middle):
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=12, seed=21)
    osys, ctx, it = _run_pair(spec, 10, middle=middle, chains=chains, loops=loops)
    try:
        _assert_close(osys, ctx)
        st, ch = ctx.getNHState(), osys.chain_state()
        for g in range(3):
            assert np.allclose(list(st.eta[g])[:chains], ch["eta"][g][:chains], rtol=1e-8, atol=1e-14)
            assert np.allclose(list(st.eta_dot[g])[:chains], ch["eta_dot"][g][:chains], rtol=1e-7, atol=1e-12)
    finally:
        ctx.close()


@pytest.mark.parametrize("n_pairs", [1, 2, 7])
def test_tiny_and_ragged_systems(n_pairs):
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=n_pairs, seed=3 + n_pairs)
    osys, ctx, it = _run_pair(spec, 8)
    try:
        assert ctx.info.num_waves >= 1 and ctx.info.num_slots_used == spec.num_atoms
        _assert_close(osys, ctx)
    finally:
        ctx.close()


def test_single_water_molecule_and_63_65_particles():
    for nmol in (1, 21, 22):                         # 3, 63 and 66 particles: around one wave
        spec = systems.spce_water(nmol, seed=nmol)
        osys, ctx, it = _run_pair(spec, 6, maxd=0.0, T=300.0)
        try:
            _assert_close(osys, ctx)
        finally:
            ctx.close()


def test_all_particles_langevin_no_nh():
    """No NH particle at all: VVIntegrator.cpp:168 never creates the NH kernel; the fused step has a single phase."""
    spec = systems.spce_water(30, seed=8)
    spec.particles_ld = list(range(spec.num_atoms))
    rnd = np.random.default_rng(5).standard_normal((2048, 4)).astype(np.float32)
    p = O.Params(temperature=300.0, max_drude_distance=0.0)
    osys = O.OracleSystem(spec, p, "mixed", random=rnd, force_mode=1)
    it = I.VVIntegrator(300.0, 10, 1.0, 40, 0.001)
    ctx = I.Context(spec, it, precision="mixed", random=rnd)
    try:
        assert H.lib.vvhip_step_middle_phases(ctx.plan) == 1 and ctx.info.num_particles_nh == 0
        osys.step(12); it.step(12)
        _assert_close(osys, ctx)
    finally:
        ctx.close()


def test_parameters_are_reread_between_steps():
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=15, seed=33)
    osys, ctx, it = _run_pair(spec, 5)
    try:
        for T, dt in ((350.0, 0.001), (320.0, 0.0005)):
            it.setTemperature(T); it.setStepSize(dt)          # reference: getters are read at every kernel call
            osys.s.temperature, osys.s.dt = T, dt
            osys.step(5); it.step(5)
        _assert_close(osys, ctx)
    finally:
        ctx.close()


def test_thermostat_checkpoint_round_trip():
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=15, seed=34)
    def fresh():
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
        return it, I.Context(spec, it, precision="mixed")
    it1, c1 = fresh()
    it1.step(10)
    state, velm, posq, corr = c1.getNHState(), c1.getVelm(), c1.getPosq(), c1.getPosqCorrection()
    it1.step(10)
    ref = (c1.getPosq(), c1.getVelm())
    c1.close()
    it2, c2 = fresh()                                         # "restart": new context, restored particle + thermostat state
    c2.velm.upload(velm); c2.posq.upload(posq); c2.posq_corr.upload(corr)
    c2.setNHState(state)
    it2.step(10)
    try:
        assert np.array_equal(c2.getPosq().view(np.uint8), ref[0].view(np.uint8))
        assert np.array_equal(c2.getVelm().view(np.uint8), ref[1].view(np.uint8))
    finally:
        c2.close()


def _big_molecule_system(seed=1):
    """Two polymers of 150 and 70 particles (with Drude pairs) plus a few small molecules: COM groups larger than a wave."""
    rng = np.random.default_rng(seed)
    masses, mol, pairs = [], [], []
    def polymer(n_heavy, n_h, m):
        for _ in range(n_heavy):
            masses.extend([11.611, 0.4]); mol.extend([m, m]); pairs.append((len(masses) - 1, len(masses) - 2))
        for _ in range(n_h):
            masses.append(1.008); mol.append(m)
    polymer(50, 50, 0)
    polymer(25, 20, 1)
    for m in range(2, 8):
        polymer(3, 2, m)
    n = len(masses)
    masses = np.array(masses)
    pos = rng.uniform(0, 3, (n, 3))
    d = np.array([p[0] for p in pairs])
    pos[d] = pos[d - 1] + rng.normal(0, 2e-4, (len(d), 3))
    vel = rng.standard_normal((n, 3)) * np.sqrt(O.BOLTZ * 300.0 / masses)[:, None]
    return systems.SystemSpec(name="polymers", masses=masses, charges=np.zeros(n), positions=pos, velocities=vel, box=np.array([3.0, 3.0, 3.0]),
                              mol_id=np.array(mol, np.int32), drude_pairs=np.array(pairs, np.int32), constraints=np.zeros((0, 2), np.int32))


@pytest.mark.parametrize("middle", [True, False])
@pytest.mark.parametrize("cos", [0.0, 0.02])
def test_molecules_larger_than_a_wave_with_com_group(middle, cos):
    spec = _big_molecule_system()
    osys, ctx, it = _run_pair(spec, 10, middle=middle, cos=cos, T=300.0)
    try:
        assert ctx.info.use_com_temp_group and ctx.info.max_cluster <= 64 and ctx.info.num_temp_groups == 3
        _assert_close(osys, ctx)
        ke_o, ke_g = osys.ke2(), np.array(list(ctx.getNHState().ke2))
        assert np.allclose(ke_g, ke_o, rtol=1e-10)
    finally:
        ctx.close()


def test_device_gaussian_generator_moments_and_graph_refresh():
    """Philox4x32-10 + Box-Muller fill of the Langevin buffer: N(0,1) moments, different numbers on every refill, and a
    Langevin system stepping through graph replay stays finite with a sane temperature."""
    spec = systems.edl_slab(num_ion_pairs=20, num_electrode=200, seed=9)
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    it.setMaxDrudeDistance(0.02)
    it.setMirrorLocation(float(spec.box[2]) / 2)
    ctx = I.Context(spec, it, precision="mixed")
    try:
        ctx.fill_random(seed=12345)
        ctx.synchronize()
        a = ctx.random.download().astype(np.float64).ravel()
        ctx.fill_random()
        ctx.synchronize()
        b = ctx.random.download().astype(np.float64).ravel()
        n = a.size
        assert abs(a.mean()) < 5 / np.sqrt(n) and abs(a.var() - 1) < 0.02 and abs((a ** 4).mean() - 3) < 0.1
        assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 5 / np.sqrt(n)
        assert not np.array_equal(a, b) and abs(np.corrcoef(a, b)[0, 1]) < 5 / np.sqrt(n)
        ctx.run_graph(200, steps_per_graph=20)
        v = ctx.getVelm()
        el = np.array(spec.particles_ld)
        T_el = (spec.masses[el, None] * v[el, :3] ** 2).sum() / (3 * len(el)) / O.BOLTZ
        assert np.isfinite(v).all() and 150 < T_el < 600, T_el
    finally:
        ctx.close()


def test_kinetic_energy_and_group_temperatures():
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=50, seed=17)
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed")
    try:
        ke0 = 0.5 * (spec.masses[:, None] * spec.velocities ** 2).sum()
        assert ctx.getKineticEnergy() == pytest.approx(ke0, rel=1e-12)
        it.step(5)
        v = ctx.getVelocities()
        assert ctx.getKineticEnergy() == pytest.approx(0.5 * (spec.masses[:, None] * v ** 2).sum(), rel=1e-12)
        T = ctx.getGroupTemperatures()
        assert 200 < T[0] < 500 and 100 < T[1] < 700 and 0.05 < T[2] < 100
        it.step(5)                                  # the query must not disturb the thermostat accumulators
        p = O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02)
        osys = O.OracleSystem(spec, p, "mixed"); osys.step(10)
        assert np.abs(ctx.getPositions() - osys.positions()).max() < 1e-11
    finally:
        ctx.close()


def test_switching_the_cos_perturbation_on_mid_run_equals_a_fresh_start():
    """setCosAcceleration at run time changes the accumulator layout (moment rows come into use) and the number of reductions
    per step: continuing must give the bits of a context that starts from the same state with the perturbation already on."""
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=50, seed=21)

    def make(cos):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
        it.setMaxDrudeDistance(0.02)
        it.setCosAcceleration(cos)
        return it, I.Context(spec, it, precision="mixed", force_provider="tether")

    it1, c1 = make(0.0)
    it1.step(7)                                                  # odd: the two accumulator / thermostat copies have swapped roles
    velm, posq, corr, nh = c1.getVelm(), c1.getPosq(), c1.getPosqCorrection(), c1.getNHState()
    it1.setCosAcceleration(0.02)
    it1.step(6)
    r1 = (c1.getVelm(), c1.getPosq(), c1.getPosqCorrection(), list(c1.getNHState().ke2), it1.getViscosity())
    c1.close()
    it2, c2 = make(0.02)
    c2.velm.upload(velm); c2.posq.upload(posq); c2.posq_corr.upload(corr)
    c2.setNHState(nh)
    c2.forces_valid = False
    it2.step(6)
    r2 = (c2.getVelm(), c2.getPosq(), c2.getPosqCorrection(), list(c2.getNHState().ke2), it2.getViscosity())
    c2.close()
    for a, b in zip(r1[:3], r2[:3]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert r1[3] == r2[3] and r1[4] == r2[4]


def test_trace_switch_does_not_change_results():
    """vvhip_set_trace (roctx ranges around the launch groups; the adapter ties it to VVIntegrator::setDebugEnabled) is observational."""
    spec = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=20, seed=5)
    res = []
    for trace in (0, 1):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
        it.setMaxDrudeDistance(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        H.check(H.lib.vvhip_set_trace(ctx.plan, trace), ctx.plan)
        it.step(5)
        res.append(ctx.getVelm())
        ctx.close()
    assert np.array_equal(res[0].view(np.uint8), res[1].view(np.uint8))
