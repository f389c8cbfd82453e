"""-m gpu: kernels compiled at run time (csrc/vv_rtc.cpp).

The reference compiles its kernels per System when the integrator is created (CudaVVKernels.cpp:98-101, 639-647).  libvvhip carries the
stage sets of the BASELINE configurations compiled ahead of time; a stage set outside that list, or a thermostat chain of 1, 2 or 4
links, is compiled by hipRTC from the same source the first time a plan launches it.  Checked here:
  * the run-time kernel IS the compiled kernel: with VVHIP_RTC=2 every launch of kernels A / B takes the run-time route, and the
    trajectories equal those of the ahead-of-time kernels bit for bit (same source, same options, -ffp-contract=off above all);
  * stage sets outside the compiled list run a run-time kernel, not the generic one (vvhip_generic_launches stays 0), against the oracle;
  * a stage set met for the first time INSIDE a graph capture compiles and loads there;
  * VVHIP_RTC=0 restores the generic kernel."""
import importlib

import numpy as np
import pytest

from oracle import oracle as O

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, S = pkg.vvhip, pkg.integrator, pkg.systems
pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_mode():
    old = I.Context.rtc_mode()
    yield
    I.Context.rtc_mode(old)


def _integrator(cfg, middle, spec, chains=3):
    it = I.VVIntegrator(300.0 if cfg == "C2" else 333.0, 10.0, 1.0, 40.0, 0.002 if cfg == "C2" else 0.001, chains, 1)
    if cfg not in ("C1", "C2"):
        it.setMaxDrudeDistance(0.02)
    if cfg == "C4":
        it.setCosAcceleration(0.02)
    if cfg == "C5":
        lz = float(spec.box[2])
        it.setMirrorLocation(lz / 2)
        it.setElectricField(2.0 / lz * 2 * 1.602176634e-22)
    it.setUseMiddleScheme(middle)
    return it


def _trajectory(spec, cfg, middle, prec, mode, steps=6):
    I.Context.rtc_mode(mode)
    it = _integrator(cfg, middle, spec)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether")
    try:
        it.step(steps)
        ctx.run_graph(4, 2)
        ctx.synchronize()
        st = ctx.getNHState()
        chain = np.array([list(st.eta[g]) + list(st.eta_dot[g]) for g in range(3)])
        return ctx.getPosq().copy(), ctx.getVelm().copy(), chain, ctx.generic_launches()[0]
    finally:
        ctx.close()


@pytest.mark.parametrize("cfg,hbonds,middle,prec", [
    ("C3", False, True, "mixed"), ("C3", True, True, "mixed"), ("C3", False, False, "mixed"), ("C4", False, True, "mixed"),
    ("C5", True, True, "mixed"), ("C2", True, True, "mixed"), ("C3", True, True, "single"), ("C3", False, True, "double"),
    ("C2", True, True, "single"), ("C2", True, False, "double"),      # SETTLE (its two solvers allow contraction: the same fusions in both builds)
])
def test_run_time_kernel_equals_compiled_kernel(cfg, hbonds, middle, prec):
    spec = S.make_config(cfg, 0.08 if cfg in ("C3", "C4") else 1.0, hbonds=hbonds)
    before = I.Context.rtc_stats()
    ref = _trajectory(spec, cfg, middle, prec, 0)
    mid = I.Context.rtc_stats()
    assert mid[1:3] == before[1:3], "VVHIP_RTC=0 must not launch run-time kernels"
    rtc = _trajectory(spec, cfg, middle, prec, 2)
    after = I.Context.rtc_stats()
    # (a step -- the classic scheme: each of its halves -- is ONE launch, an instance of kernel B that also runs kernel A's stages, wherever the plan allows it)
    assert after[2] > mid[2], "VVHIP_RTC=2: the step's kernels take the run-time route"
    assert ref[3] == (0, 0)
    for a, b, what in zip(ref[:3], rtc[:3], ("posq", "velm", "chain")):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), f"{what} differs between the compiled and the run-time kernel"


@pytest.mark.parametrize("what", ["large", "sharded", "large rigid water"])
def test_run_time_kernel_equals_compiled_kernel_large_and_sharded(what):
    """The arithmetic work-item layout with the stand-alone chain launch (C3x30, 3.3 M particles) and the mailbox stage sets of a sharded plan
    (one rank of two, its own handle as the only peer): run-time kernels against the compiled ones, bit for bit."""
    D = pkg.distributed
    cfg = "C2" if "water" in what else "C3"
    spec = S.make_config("C2", 120.0, hbonds=True) if cfg == "C2" else S.make_config("C3", 30.0 if what == "large" else 0.08)
    res = []
    for mode in (0, 2):
        I.Context.rtc_mode(mode)
        it = _integrator(cfg, True, spec)
        kw = {"shard": D.shard_bounds(spec, 2)[0]} if what == "sharded" else {}
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether", **kw)
        try:
            if what == "sharded":
                ctx.mailbox_connect(ctx.mailbox_create(1, 0))
            it.step(4)
            ctx.synchronize()
            res.append((ctx.getPosq().copy(), ctx.getVelm().copy(), ctx.generic_launches()[0]))
        finally:
            ctx.close()
    assert res[0][2] == (0, 0) and res[1][2] == (0, 0)
    assert np.array_equal(res[0][0].view(np.uint8), res[1][0].view(np.uint8)) and np.array_equal(res[0][1].view(np.uint8), res[1][1].view(np.uint8))


def _oracle_pair(spec, nsteps, chains, middle, ld=False, efield=0.0):
    p = O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02, use_middle_scheme=middle, num_chains=chains, loops_per_step=1)
    osys = O.OracleSystem(spec, p, "mixed", force_mode=1)
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001, chains, 1)
    it.setMaxDrudeDistance(0.02)
    it.setUseMiddleScheme(middle)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    osys.step(nsteps)
    it.step(nsteps)
    return osys, ctx


@pytest.mark.parametrize("chains", [1, 2, 4])
@pytest.mark.parametrize("middle", [True, False])
def test_other_chain_lengths_get_their_own_kernel(chains, middle):
    """The compiled kernels carry the three-link chain; 1, 2 and 4 links used to run the generic kernel."""
    I.Context.rtc_mode(1)
    spec = S.drude_il(cells=(1, 1, 1), pairs_per_cell=40, seed=31)
    before = I.Context.rtc_stats()
    osys, ctx = _oracle_pair(spec, 10, chains, middle)
    try:
        counts, sets = ctx.generic_launches()
        assert counts == (0, 0), f"generic kernel ran {counts}, stage sets A 0x{sets[0]:x} B 0x{sets[1]:x}"
        assert I.Context.rtc_stats()[2] > before[2]
        x_o, x_g = osys.positions(), ctx.getPositions()
        v_o, v_g = osys.velm[:, :3].astype(np.float64), ctx.getVelocities()
        m = osys.velm[:, 3] != 0
        assert np.abs(x_g - x_o).max() / np.abs(x_o).max() < 1e-9 and np.abs(v_g[m] - v_o[m]).max() / np.abs(v_o[m]).max() < 1e-9
        st, ch = ctx.getNHState(), osys.chain_state()
        for g in range(3):
            assert np.allclose(list(st.eta[g])[:chains], ch["eta"][g][:chains], rtol=1e-8, atol=1e-14)
            assert np.allclose(list(st.eta_dot[g])[:chains], ch["eta_dot"][g][:chains], rtol=1e-7, atol=1e-12)
    finally:
        ctx.close()


def test_uncompiled_stage_set_inside_a_graph_capture_and_mode_zero():
    """Chain length 2 met for the first time while the step is being captured: the kernel compiles and loads inside the capture, the
    replay equals the eager run bit for bit; with VVHIP_RTC=0 the same plan counts generic launches."""
    spec = S.drude_il(cells=(1, 1, 1), pairs_per_cell=50, seed=77)
    outs = []
    for how in ("graph", "eager", "generic"):
        I.Context.rtc_mode(0 if how == "generic" else 1)
        it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001, 2, 1)
        it.setMaxDrudeDistance(0.02)
        ctx = I.Context(spec, it, precision="double", force_provider="tether")
        try:
            if how == "graph":
                ctx.run_graph(12, 4)
            else:
                it.step(12)
            ctx.synchronize()
            counts, _ = ctx.generic_launches()
            assert (counts[1] > 0) == (how == "generic")
            outs.append((ctx.getPosq().copy(), ctx.getVelm().copy()))
        finally:
            ctx.close()
    assert np.array_equal(outs[0][0].view(np.uint8), outs[1][0].view(np.uint8)) and np.array_equal(outs[0][1].view(np.uint8), outs[1][1].view(np.uint8))
    # generic vs specialised: the same arithmetic statements; the chain of a 2-link thermostat runs through the run-time switch instead of the
    # templated body -- element-wise stages agree to rounding of the scale factors
    assert np.allclose(outs[0][0], outs[2][0], rtol=0, atol=1e-11) and np.allclose(outs[0][1][:, :3], outs[2][1][:, :3], rtol=1e-10, atol=1e-12)


def _random_stage_sets(rng, n):
    """Stage-bit combinations the launchers accept on a plain NH + Drude system (no Langevin, field, images, constraints, mailbox, big molecules),
    sensible or not: the generic kernel takes any of them, so must a kernel compiled for exactly that set."""
    A = {k: getattr(H, k) for k in ("A_FE_LOAD", "A_FE_STORE", "A_COS", "A_KICK_FULL", "A_KICK_HALF", "A_POSDELTA_VV", "A_POS1", "A_BIAS", "A_KE", "A_CZ_STORE", "A_CZ_LOAD")}
    A.update(A_KE_MOM=1 << 17, A_KE_PLAIN=1 << 15, A_NOSTORE=1 << 19, A_UNBIAS_ACC=1 << 11)
    B = {k: getattr(H, k) for k in ("B_SCALE", "B_UNBIAS", "B_BIAS_REMOVE", "B_BIAS_RESTORE", "B_DRIFT_MIDDLE", "B_POS2", "B_POS3", "B_VV_KICK", "B_VV_POS", "B_HARDWALL", "B_CHAIN", "B_CZ_LOAD")}
    B.update(B_KE_MOM=1 << 15, B_KICK=1 << 17)
    out = []
    while len(out) < n:
        if rng.random() < 0.5:
            f = 0
            for k, bit in A.items():
                if rng.random() < 0.35:
                    f |= bit
            if (f & A["A_KICK_FULL"]) and (f & A["A_KICK_HALF"]):
                f &= ~A["A_KICK_HALF"]
            if (f & A["A_CZ_STORE"]) and (f & A["A_CZ_LOAD"]):
                f &= ~A["A_CZ_LOAD"]
            if f & A["A_KE_MOM"]:
                f |= A["A_KE"] | A["A_BIAS"]
                f &= ~A["A_UNBIAS_ACC"]
            if (f & A["A_KE_PLAIN"]) and (f & A["A_KE"]):
                f &= ~A["A_KE_PLAIN"]
            if f & A["A_NOSTORE"] and not f & (A["A_KICK_FULL"] | A["A_KICK_HALF"]):
                f &= ~A["A_NOSTORE"]
            if f:
                out.append((0, f))
        else:
            f = 0
            for k, bit in B.items():
                if rng.random() < 0.35:
                    f |= bit
            pos = [B["B_DRIFT_MIDDLE"], B["B_POS2"], B["B_POS3"], B["B_VV_KICK"], B["B_VV_POS"]]
            chosen = [b for b in pos if f & b]
            for b in chosen[1:]:
                f &= ~b
            if f & B["B_UNBIAS"]:
                f &= ~(B["B_BIAS_REMOVE"] | B["B_BIAS_RESTORE"])
            if (f & B["B_KE_MOM"]) and not (f & B["B_UNBIAS"] and f & B["B_CHAIN"]):
                f &= ~B["B_KE_MOM"]
            if (f & B["B_KICK"]) and (f & B["B_VV_KICK"]):
                f &= ~B["B_KICK"]
            if f:
                out.append((1, f))
    return out


def test_random_stage_sets_specialised_kernel_equals_generic_kernel():
    """A walk through the space of stage sets: 40 random combinations of stage bits, each launched on the same state once on the generic kernel
    (run-time stage bits, VVHIP_RTC=0) and once on a kernel compiled at run time for exactly that set -- every particle array, the accumulators'
    totals and the thermostat state must come out the same to the bit.  What specialisation removes from a kernel must be dead code, for every set."""
    import ctypes as C
    rng = np.random.default_rng(20260117)
    spec = S.drude_il(cells=(1, 1, 1), pairs_per_cell=30, seed=9)
    sets = _random_stage_sets(rng, 40)
    results = []
    compiled_before = I.Context.rtc_stats()[0]
    for mode in (0, 1):
        I.Context.rtc_mode(mode)
        it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
        it.setMaxDrudeDistance(0.02)
        it.setCosAcceleration(0.02)
        ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
        snap = []
        try:
            it.step(2)                                             # a physical state with forces, cos cache and thermostat history
            ctx.synchronize()
            for kernel, flags in sets:
                H.check(H.lib.vvhip_debug_launch(ctx.plan, kernel, flags, 0), ctx.plan)
                ctx.synchronize()
                acc = (C.c_double * 4)()
                H.check(H.lib.vvhip_debug_read_accumulators(ctx.plan, acc, 0), ctx.plan)
                st = ctx.getNHState()
                snap.append((ctx.getPosq().copy(), ctx.getVelm().copy(), np.array(list(acc) + [x for g in range(3) for x in list(st.eta[g]) + list(st.eta_dot[g])])))
            results.append(snap)
        finally:
            ctx.close()
    assert I.Context.rtc_stats()[0] - compiled_before >= 20, "most of these sets are outside the compiled list: they must have been compiled at run time"
    for i, ((kernel, flags), a, b) in enumerate(zip(sets, results[0], results[1])):
        for x, y, what in zip(a, b, ("posq", "velm", "accumulators and thermostat")):
            assert np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)), \
                f"launch {i}: kernel {'AB'[kernel]} stage set 0x{flags:x}: {what} differs between the generic and the specialised kernel"



@pytest.mark.parametrize("middle", [True, False])
def test_first_launch_inside_a_graph_capture(middle):
    """A plan whose very first step is enqueued by vvhip_run_graph meets its uncompiled stage set (a two-link chain) INSIDE the stream capture:
    the kernel is compiled and loaded there, the captured graph replays it, and the trajectory is the one of step-by-step launches."""
    I.Context.rtc_mode(1)
    spec = S.drude_il(cells=(1, 1, 1), pairs_per_cell=24, seed=77)
    out = []
    before = I.Context.rtc_stats()[0]
    for graph_first in (True, False):
        it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001, 2, 1)
        it.setMaxDrudeDistance(0.02)
        it.setUseMiddleScheme(middle)
        ctx = I.Context(spec, it, precision="single", force_provider="tether")      # (two links in single precision: no other test has compiled these)
        try:
            if graph_first:
                ctx.run_graph(8, 4)
                assert I.Context.rtc_stats()[0] > before, "nothing was compiled inside the capture"
            else:
                it.step(8)
            ctx.synchronize()
            assert ctx.generic_launches()[0] == (0, 0)
            out.append((ctx.getPosq().copy(), ctx.getVelm().copy(), bytes(ctx.getNHState())))
        finally:
            ctx.close()
    assert np.array_equal(out[0][0].view(np.uint8), out[1][0].view(np.uint8)) and np.array_equal(out[0][1].view(np.uint8), out[1][1].view(np.uint8))
    assert out[0][2] == out[1][2]
