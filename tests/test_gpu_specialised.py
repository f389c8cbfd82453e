"""-m gpu: every stage set that the BASELINE configurations launch has a kernel compiled for it.

The reference JIT-compiles its kernels per System (CudaVVKernels.cpp:98-101, 639-647: the defines carry the System's sizes and
precision), so each System runs code specialised for it.  Here the stage sets of the supported paths are compiled at build time (vv_kernels.hip:
SF_*); a launch whose stage set is not in the list gets a kernel compiled at run time (tests/test_gpu_rtc.py) and, failing that, the
generic kernel with run-time stage bits, 15-20 % slower.  vvhip_generic_launches counts those; this test runs every BASELINE configuration -- plain, with
the constraints the example scripts put on it, in the classic scheme, sharded with the mailbox exchange, and at the size where the
arithmetic layout and the stand-alone chain launch take over -- and requires the count to stay 0."""
import importlib
import os

import pytest

pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, D = pkg.integrator, pkg.systems, pkg.distributed
pytestmark = pytest.mark.gpu

@pytest.fixture
def no_run_time_kernels():
    """The compiled list itself covers the case: no launch took a kernel compiled at run time either."""
    if os.environ.get("VVHIP_RTC") == "2":
        pytest.skip("VVHIP_RTC=2 sends every launch to a run-time kernel: nothing to say about the compiled list")
    before = I.Context.rtc_stats()
    yield
    after = I.Context.rtc_stats()
    assert after[1:3] == before[1:3], f"launches of run-time kernels: A {after[1] - before[1]}, B {after[2] - before[2]}"


CASES = [(cfg, hb, middle) for cfg in ("C1", "C2", "C3", "C4", "C5") for hb in (False, True) for middle in (True, False)
         if not (cfg == "C1" and hb)]


def _integrator(cfg, middle):
    it = I.VVIntegrator(300.0 if cfg == "C2" else 333.0, 10.0, 1.0, 40.0, 0.002 if cfg == "C2" else 0.001)
    if cfg not in ("C1", "C2"):
        it.setMaxDrudeDistance(0.02)
    if cfg == "C4":
        it.setCosAcceleration(0.02)
    it.setUseMiddleScheme(middle)
    return it


def _edl(it, spec):
    lz = float(spec.box[2])
    it.setMirrorLocation(lz / 2)
    it.setElectricField(2.0 / lz * 2 * 1.602176634e-22)


@pytest.mark.parametrize("cfg,hbonds,middle", CASES)
def test_baseline_configurations_run_compiled_stage_sets(cfg, hbonds, middle, no_run_time_kernels):
    spec = S.make_config(cfg, hbonds=hbonds)
    it = _integrator(cfg, middle)
    if cfg == "C5":
        _edl(it, spec)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    try:
        it.step(4)
        ctx.run_graph(8, 4)
        ctx.synchronize()
        counts, sets = ctx.generic_launches()
        assert counts == (0, 0), f"{cfg} hbonds={hbonds} middle={middle}: generic kernel ran {counts} times, stage sets A 0x{sets[0]:x} B 0x{sets[1]:x}"
    finally:
        ctx.close()


@pytest.mark.parametrize("cfg,hbonds", [("C3", False), ("C3", True), ("C4", False)])
def test_sharded_plans_run_compiled_stage_sets(cfg, hbonds, no_run_time_kernels):
    """One rank of a two-rank decomposition with the mailbox exchange set up (its own handle as the only peer is enough to take the
    B_MAILBOX stage sets; the numbers are not used)."""
    spec = S.make_config(cfg, hbonds=hbonds)
    bounds = D.shard_bounds(spec, 2)
    it = _integrator(cfg, True)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", shard=bounds[0])
    try:
        h = ctx.mailbox_create(1, 0)
        ctx.mailbox_connect(h)
        it.step(4)
        ctx.synchronize()
        counts, sets = ctx.generic_launches()
        assert counts == (0, 0), f"{cfg} hbonds={hbonds} sharded: generic kernel ran {counts} times, stage sets A 0x{sets[0]:x} B 0x{sets[1]:x}"
    finally:
        ctx.close()


@pytest.mark.parametrize("cfg,scale,hbonds,middle", [("C3", 8.0, False, True), ("C3", 10.0, False, True), ("C3", 10.0, True, True), ("C3", 10.0, False, False),
                                                     ("C3", 30.0, False, True), ("C3", 30.0, True, True), ("C3", 30.0, False, False),
                                                     ("C2", 120.0, False, True), ("C2", 120.0, True, True)])
def test_large_boxes_run_compiled_stage_sets(cfg, scale, hbonds, middle, no_run_time_kernels):
    """0.9 M particles (C3x8): best-fit layout, strided grid; 1.1 M (C3x10) / 1.2 M (water x120): arithmetic layout, chain in kernel B;
    3.3 M (C3x30): arithmetic layout, capped grids, the stand-alone chain launch."""
    spec = S.make_config(cfg, scale, hbonds=hbonds)
    it = _integrator(cfg, middle)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    try:
        assert ctx.info.periodic_layout == (0 if scale < 10 else 1)
        it.step(2)
        ctx.synchronize()
        counts, sets = ctx.generic_launches()
        assert counts == (0, 0), f"{cfg}x{scale} hbonds={hbonds} middle={middle}: generic kernel ran {counts} times, stage sets A 0x{sets[0]:x} B 0x{sets[1]:x}"
    finally:
        ctx.close()


def test_an_unlisted_stage_set_is_reported():
    """Four chain links instead of three: kernel B's compiled specialisations carry the three-link chain only.  By default a kernel is
    compiled at run time for the plan (tests/test_gpu_rtc.py); with that switched off (VVHIP_RTC=0) the generic kernel runs and the plan says so."""
    spec = S.drude_il(cells=(1, 1, 1), pairs_per_cell=20, seed=2)
    old = I.Context.rtc_mode()
    try:
        for mode, expected in ((0, 3), (1, 0)):
            I.Context.rtc_mode(mode)
            it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001, numNHChains=4)
            it.setMaxDrudeDistance(0.02)
            ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
            try:
                it.step(3)
                ctx.synchronize()
                counts, sets = ctx.generic_launches()
                assert counts[1] == expected and counts[0] == 0 and (sets[1] != 0) == (expected != 0)
            finally:
                ctx.close()
    finally:
        I.Context.rtc_mode(old)
