"""-m gpu: the sticky health word (accumulator overflow), graph executables per thermostat parity, and the static mass tables."""
import importlib

import numpy as np
import pytest

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, S = pkg.vvhip, pkg.integrator, pkg.systems
pytestmark = pytest.mark.gpu


def _ctx(spec, **kw):
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    return it, I.Context(spec, it, precision="mixed", force_provider="tether", **kw)


def test_accumulator_overflow_is_reported_not_swallowed():
    """2KE beyond 1024 x the thermostat target cannot be held by the int64 fixed-point accumulators: the kernels raise a sticky flag
    in pinned host memory, vvhip_synchronize and the run loops return VVHIP_ERR_OVERFLOW instead of thermostatting on garbage."""
    spec = S.make_config("C3", 0.004)
    it, ctx = _ctx(spec)
    try:
        it.step(2)
        ctx.synchronize()
        assert ctx.status() == (False, False)
        ctx.setVelocities(ctx.getVelocities() * 1.0e4)                 # 1e8 x the kinetic energy
        it.step(1)
        with pytest.raises(H.VVHipError) as e:
            ctx.synchronize()
        assert e.value.code == H.ERR_OVERFLOW and "overflow" in str(e.value)
        assert ctx.status() == (False, True)
        for call in (lambda: ctx.run_graph(8, 8), lambda: ctx.run_eager(1)):
            with pytest.raises(H.VVHipError) as e:
                call()
            assert e.value.code == H.ERR_OVERFLOW
        ctx.status_clear()
        assert ctx.status() == (False, False)
    finally:
        ctx.close()


def test_a_nan_velocity_raises_the_flag_too():
    spec = S.make_config("C3", 0.004)
    it, ctx = _ctx(spec)
    try:
        v = ctx.getVelocities()
        v[5, 0] = np.nan
        ctx.setVelocities(v)
        it.step(1)
        with pytest.raises(H.VVHipError) as e:
            ctx.synchronize()
        assert e.value.code == H.ERR_OVERFLOW
    finally:
        ctx.close()


def test_graph_executables_per_parity_and_odd_tails_equal_eager_stepping():
    """Replays and host-launched tails in any mix (odd tails flip the thermostat parity between replays: both executables are used)
    give the bits of plain step-by-step launching."""
    spec = S.make_config("C3", 0.01)
    it_a, a = _ctx(spec)
    it_b, b = _ctx(spec)
    try:
        n = 0
        a.graph_prepare(8)                       # both parities, nothing launched: state untouched
        assert np.array_equal(a.getVelocities(), b.getVelocities())
        for steps, spg in ((3, 8), (17, 8), (8, 8), (21, 8), (5, 4), (16, 8)):
            a.run_graph(steps, spg)
            n += steps
        it_b.step(n)
        assert np.array_equal(a.getPositions(), b.getPositions()) and np.array_equal(a.getVelocities(), b.getVelocities())
        sa, sb = a.getNHState(), b.getNHState()
        assert list(sa.vscale) == list(sb.vscale) and list(sa.ke2) == list(sb.ke2)
    finally:
        a.close(); b.close()


def test_langevin_graph_tail_uses_filled_random_slices():
    """A prepared but never replayed graph must not leave the random cursor in slices the device generator has not filled: eager
    steps after vvhip_graph_prepare read the buffer the host bound, and replays refill it themselves."""
    spec = S.make_config("C5", 0.05)
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    lz = float(spec.box[2])
    it.setMirrorLocation(lz / 2)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    try:
        ctx.graph_prepare(10)
        ctx.run_eager(3)
        ctx.run_graph(23, 10)
        ctx.synchronize()
        v = ctx.getVelocities()
        m = np.asarray(spec.masses) > 0
        assert np.isfinite(v[m]).all() and np.isfinite(ctx.getPositions()).all()
        ld = np.array(spec.particles_ld)
        T = (np.asarray(spec.masses)[ld, None] * v[ld] ** 2).sum() / (3 * len(ld)) / 8.31446261815324e-3
        assert 150 < T < 600, T                  # the Langevin subset stays thermal
    finally:
        ctx.close()


def test_mass_tables_follow_velm():
    """vvhip_masses_changed refills the static tables from velm.w in front of the next launch: with unchanged masses the trajectory
    is bit-identical to an undisturbed run (graphs are re-captured), and the tables really are what kernel B scales with -- after
    swapping in a velm with different inverse masses the pair splitting follows them."""
    spec = S.make_config("C3", 0.01)
    it_a, a = _ctx(spec)
    it_b, b = _ctx(spec)
    try:
        a.run_graph(16, 8)
        H.check(H.lib.vvhip_masses_changed(a.plan), a.plan)
        a.run_graph(16, 8)
        b.run_graph(32, 8)
        assert np.array_equal(a.getVelocities(), b.getVelocities()) and np.array_equal(a.getPositions(), b.getPositions())
        # heavier Drude particles in a's velm only: without the refresh the old fractions would be used and the runs would agree
        velm = a.getVelm()
        d = spec.drude_pairs[:, 0]
        velm[d, 3] = 1.0 / 0.8
        a.velm.upload(velm)
        velb = b.getVelm()
        velb[d, 3] = 1.0 / 0.8
        b.velm.upload(velb)
        H.check(H.lib.vvhip_masses_changed(a.plan), a.plan)
        a.run_eager(4)
        b.run_eager(4)                           # b keeps the stale tables
        assert not np.array_equal(a.getVelocities(), b.getVelocities())
    finally:
        a.close(); b.close()


def test_a_solver_that_hits_its_iteration_cap_says_so():
    """A rigid triangle with an angle of ~1.2 degrees between two of its bonds: the coloured Gauss-Seidel sweeps of the general clusters
    (vv_device.inc: general_positions / general_velocities) do not converge it in their 150 rounds -- found by the random-topology fuzz of
    round 4, where GPU and oracle stopped at the cap alike and nobody was told.  Now the kernels raise sticky word [3] and the host gets
    VVHIP_ERR_CONSTRAINT from the next synchronisation; a well-conditioned copy of the same system leaves the word alone."""
    def build(angle_deg):
        spec = S.spce_water(12, seed=21)
        x = spec.positions
        u = np.array([1.0, 0.0, 0.0])
        th = np.deg2rad(angle_deg)
        x[1] = x[0] + 0.10 * u
        x[2] = x[0] + 0.13 * np.array([np.cos(th), np.sin(th), 0.0])          # scalene: not a SETTLE molecule, a general cluster of three constraints
        cons = np.array([[0, 1], [0, 2], [1, 2]], dtype=np.int32)
        spec.constraints = cons
        spec.constraint_distances = np.linalg.norm(x[cons[:, 0]] - x[cons[:, 1]], axis=1)
        return spec

    for angle, expect in ((60.0, 0), (1.2, 1)):
        spec = build(angle)
        it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.002)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        try:
            assert ctx.info.constraints_fused and ctx.info.num_general_constraints == 3
            raised = None
            try:
                it.step(2)
                ctx.synchronize()
            except H.VVHipError as e:
                raised = e
            words = ctx.status_words()
            assert words[3] == expect, (angle, words)
            if expect:
                assert raised is not None and raised.code == H.ERR_CONSTRAINT and "iteration cap" in raised.message
                ctx.status_clear()
                assert ctx.status_words() == [0, 0, 0, 0]
            else:
                assert raised is None
        finally:
            ctx.close()
