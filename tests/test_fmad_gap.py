"""How far does the reference drift from ITSELF when its multiply-adds are contracted?

The reference compiles its kernels at run time with NVRTC's default options (CudaVVKernels.cpp:98-101: no `--fmad=false`), so on an NVIDIA
GPU every a*b+c the compiler finds becomes one fused operation.  The oracle, oracle/_ref and the product are all built with contraction
off (every product and sum rounded separately): "bit for bit with the reference" means the reference's SOURCE under IEEE-separate
rounding.  This file bounds the distance between the two readings of the same source with what this image can execute:

* oracle/_ref/libvvref_host_<prec>_fmad.so  -- the reference's whole step (its VVIntegrator.cpp, CudaVVKernels.cpp, kernels/*.cu compiled
  in place for the CPU, oracle/Makefile `reffmad`) built with -ffp-contract=fast -mfma, against the same build with contraction off;
* oracle/liboracle_<prec>_fmad.so           -- the restatement, the same two ways (classic scheme, full-size boxes).

gcc's choice of which products to fuse is not NVRTC's: these runs do not reproduce an NVIDIA GPU's bits, they measure the SIZE of the
effect -- a per-operation change of <= 1/2 ulp wherever a multiply feeds an add -- over 200 steps of every BASELINE configuration's
machinery (reduced particle numbers; C3 and C4 once at full size).  Forces follow the positions (tether + Drude spring, recomputed from
each run's own positions before every step), so rounding differences feed back through the dynamics as they would in a simulation.

Measured: mixed / double precision <= 2e-14 (positions) and <= 1e-13 (velocities) relative after 200 steps, chain
state <= 1e-15 -- nine orders of magnitude inside north_star's 1e-5.  Single precision sits at float rounding with or without
contraction (1 ulp of a position under the Drude spring).  One exception, stated below: Drude PAIRS inside the
Langevin subset (not a configuration the reference's examples build).
"""
import importlib
import os

import numpy as np
import pytest

from oracle import oracle as O, refhost as RH
from oracle.make_golden_refhost import make_spec

systems = importlib.import_module("openmm-velocityverlet_amd.systems")
STEPS = 200

have_oracle = pytest.mark.skipif(not O.have_fmad(), reason="oracle/liboracle_*_fmad.so not built (make -C oracle fmad) or no FMA on this CPU")
have_ref = pytest.mark.skipif(not (O.have_fmad() and all(RH.available(p) and RH.available(p, fmad=True) for p in O.PRECISIONS)),
                              reason="oracle/_ref/libvvref_host_*[_fmad].so not built (reference sources absent)")


def rel(a, b):
    return float(np.abs(np.asarray(a, float) - np.asarray(b, float)).max() / max(np.abs(np.asarray(b, float)).max(), 1e-300))


def normals_for(spec, params, steps):
    t = O.build_tables(spec, params)
    n = (max(len(t["normal_ld"]), 1) + 2 * max(len(t["pairs_ld"]), 1)) * steps + 5
    return np.random.default_rng(3).standard_normal((n, 4)).astype(np.float32)


def edl_as_the_example_builds_it():
    """run-edl.py's structure: Langevin on the electrode atoms only (no Drude pair in the Langevin set), images, field."""
    spec = systems.edl_slab(num_ion_pairs=3, num_electrode=10, seed=23)
    return spec, O.Params(temperature=333.0, max_drude_distance=0.02, mirror_location=float(spec.box[2]) / 2,
                          electric_field=2.0 / float(spec.box[2]) * 1.602176634e-22)


REDUCED = {     # BASELINE configuration -> the same machinery at a size the CPU steps in a second
    "C1 (non-Drude, plain NH)": lambda: make_spec("nondrude"),
    "C2 (water, plain NH)": lambda: make_spec("water"),
    "C3 (Drude IL, TGNH, hard wall)": lambda: make_spec("bulk_middle"),
    "C4 (C3 + cos acceleration)": lambda: make_spec("bulk_middle_cos"),
    "C5 (electrode slab: Langevin subset, images, field)": edl_as_the_example_builds_it,
}
# what "agree" means per precision: (positions, velocities, chain state), relative to the largest component
BOUND = {"mixed": (1e-9, 1e-9, 1e-8), "double": (1e-9, 1e-9, 1e-8),
         # float arithmetic: one ulp of a position at 2-6 nm is 2e-7 relative, and under the 209 200 kJ/mol/nm^2 Drude spring that ulp is
         # 1e-3 nm/ps of Drude velocity within a step -- with or without contraction (tests/test_gpu_steps.py: oracle vs oracle just the same)
         "single": (2e-6, 2e-4, 1e-5)}


def reference_pipeline_both_ways(spec, params, prec, steps):
    """The reference's whole step, contraction off and on, each run fed forces from ITS OWN positions before every step (middle scheme:
    the force evaluation opens the step).  Also returns whether the uncontracted run equals the oracle bit for bit all the way."""
    rnd = normals_for(spec, params, steps)
    helper = O.OracleSystem(spec, params, prec, random=rnd, force_mode=1)          # the force provider only
    o = O.OracleSystem(spec, params, prec, random=rnd, force_mode=1)
    runs = [RH.RefHost(spec, params, prec, random=rnd, fmad=f) for f in (False, True)]
    for r in runs:
        assert r.h, r.error
    for _ in range(steps):
        for r in runs:
            helper.state["posq"][:] = r.posq
            helper.tether_force()
            r.state["force"][:] = helper.force
            r._up(3, r.state["force"])
            r.step(1)
        o.step(1)
    plain, fused = runs
    same = np.array_equal(plain.velm.view(np.uint8), o.velm.view(np.uint8)) and np.array_equal(plain.posq.view(np.uint8), o.posq.view(np.uint8))
    tp, tf = plain.thermostat(), fused.thermostat()
    chain = max(rel(tf["eta"], tp["eta"]), rel(tf["eta_dot"], tp["eta_dot"])) if tp else 0.0
    out = rel(fused.positions(), plain.positions()), rel(fused.velm[:, :3], plain.velm[:, :3]), chain, same
    for r in runs:
        r.close()
    return out


def oracle_both_ways(spec, params, prec, steps, threads=1):
    rnd = normals_for(spec, params, steps)
    a = O.OracleSystem(spec, params, prec, random=rnd, force_mode=1, num_threads=threads)
    b = O.OracleSystem(spec, params, prec, random=rnd, force_mode=1, num_threads=threads, fmad=True)
    a.step(steps)
    b.step(steps)
    ca, cb = a.chain_state(), b.chain_state()
    return rel(b.positions(), a.positions()), rel(b.velm[:, :3], a.velm[:, :3]), max(rel(cb["eta"], ca["eta"]), rel(cb["eta_dot"], ca["eta_dot"]))


@have_ref
@pytest.mark.parametrize("prec", O.PRECISIONS)
@pytest.mark.parametrize("name", sorted(REDUCED))
def test_reference_step_with_contracted_multiply_adds(name, prec, capfd):
    if name.startswith("C5") and prec != "mixed":
        pytest.skip("the reference dereferences a null posqCorrection for image pairs outside mixed precision")
    spec, params = REDUCED[name]()
    ex, ev, ec, same = reference_pipeline_both_ways(spec, params, prec, STEPS)
    capfd.readouterr()                                   # the reference's initialize() prints its tables
    assert same, f"{name}/{prec}: the uncontracted reference pipeline left the oracle within {STEPS} steps of live forces"
    bx, bv, bc = BOUND[prec]
    assert ex <= bx and ev <= bv and ec <= bc, (name, prec, ex, ev, ec)
    if prec != "single":                                 # what was measured, with an order of magnitude to spare: a regression guard
        assert ex <= 2e-13 and ev <= 1e-12, (name, prec, ex, ev)


@have_oracle
@pytest.mark.parametrize("prec", ["mixed", "double"])
@pytest.mark.parametrize("name", ["bulk_classic", "bulk_classic_cos", "bulk_chain5_loops3", "bulk_nocom", "edl_classic"])
def test_restatement_with_contracted_multiply_adds_other_schemes(name, prec):
    """The classic scheme evaluates forces in mid-step, which the reference build's stand-in context cannot do with live forces: the
    restatement (bit for bit with the reference, tests/test_ref_host.py) both ways instead."""
    if name.startswith("edl") and prec != "mixed":
        pytest.skip("image pairs: mixed precision only")
    spec, params = make_spec(name)
    if name.startswith("edl"):                           # the golden configuration puts one IL molecule into the Langevin set: see the last test
        spec, params = edl_as_the_example_builds_it()
        params.use_middle_scheme = False
    ex, ev, ec = oracle_both_ways(spec, params, prec, STEPS)
    assert ex <= 1e-9 and ev <= 1e-9 and ec <= 1e-8, (name, prec, ex, ev, ec)
    assert ex <= 2e-13 and ev <= 2e-11, (name, prec, ex, ev)


@have_oracle
@pytest.mark.parametrize("cfg", ["C3", "C4"])
def test_full_size_box_with_contracted_multiply_adds(cfg):
    """The 111 000-particle box of the headline metric (C4: with the cos perturbation), mixed precision as the examples run it.
    One thread: the restatement's OpenMP reductions add in a run-dependent order, and at this size even THAT (1e-16) shows now and then --
    a mixed-precision position is posq (float) + posqCorrection (float), forces are evaluated from posq alone (as OpenMM's are), and a
    last-bit difference of the double sum occasionally rounds posq to the neighbouring float: the Drude spring then kicks that particle by
    1e-5 of the velocity scale (seen between two UNcontracted 8-thread runs: positions 9e-9, velocities 1.1e-5 at one Drude particle, every
    other particle at 1e-15).  A property of the reference's mixed precision, not of contraction; single-threaded runs are reproducible."""
    spec = systems.make_config(cfg)
    params = O.Params(temperature=333.0, max_drude_distance=0.02, cos_acceleration=0.02 if cfg == "C4" else 0.0)
    ex, ev, ec = oracle_both_ways(spec, params, "mixed", STEPS, threads=1)
    assert ex <= 1e-9 and ev <= 1e-9 and ec <= 1e-8, (cfg, ex, ev, ec)
    assert ex <= 1e-13 and ev <= 1e-13, (cfg, ex, ev)


@have_oracle
def test_langevin_drude_pairs_are_the_one_place_where_contraction_shows():
    """forceExtra is `real3` -- float also in mixed precision (CudaVVKernels.cpp:79-89).  For a Drude PAIR in the Langevin set the drag and
    noise on the relative coordinate go through that float array; one float ulp there (contracted or not: any two compilers differ by it)
    is amplified by the stiff Drude spring to ~1e-4 of the velocity scale within 200 steps.  The reference's examples put electrode atoms
    (no Drudes) into the Langevin set, so no BASELINE configuration sees it; the bound is recorded so that nobody mistakes it for a bug."""
    spec, params = make_spec("edl")                      # golden configuration: first IL molecule (8 Drude pairs) moved into the Langevin set
    ex, ev, _ = oracle_both_ways(spec, params, "mixed", STEPS)
    assert 1e-9 < ev < 2e-3 and ex < 1e-5, (ex, ev)
    others = [i for i in range(spec.num_atoms) if i not in set(spec.particles_ld)]
    rnd = normals_for(spec, params, STEPS)
    a = O.OracleSystem(spec, params, "mixed", random=rnd, force_mode=1)
    b = O.OracleSystem(spec, params, "mixed", random=rnd, force_mode=1, fmad=True)
    a.step(STEPS); b.step(STEPS)
    assert rel(b.velm[others, :3], a.velm[others, :3]) < 1e-11      # everything outside the Langevin pairs stays at double rounding


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(O.__file__), "vv_oracle.c")), reason="oracle sources absent")
def test_the_prelude_assumption_is_worth_one_float_ulp_and_nothing_at_the_headline():
    """The second thing this image cannot verify: what OpenMM 8.1.2's CudaContext prepends in MIXED precision.  Assumed (oracle/ref_prelude.h):
    `SQRT = sqrtf`, `RECIP(x) = 1.0f/(x)`.  RECIP of a double argument is a double quotient either way; SQRT is what matters -- the Drude
    separation of the hard wall and the mass roots of the Langevin stage would be float roots of double values.  `make -C oracle altprelude`
    builds the restatement under the OTHER reading (double roots), and this test measures what rests on the assumption:
      * C1-C4 as benchmarked: NOTHING -- no particle reaches the hard wall, no Langevin subset; every bit of 200 steps of the full C3 box is equal;
      * C5 (Langevin electrode): 5e-9 of the velocity scale after 200 steps (float `forceExtra` swallows most of a double root's extra digits);
      * systems built to hit the wall from the first step (the golden configurations): one float ulp per hit, amplified by the Drude spring
        to 1e-7 (middle scheme) ... 1e-4 (classic) of the velocity scale after 200 steps -- the size any change of the last float bit has there."""
    def both(spec, params, steps=STEPS):
        rnd = normals_for(spec, params, steps)
        a = O.OracleSystem(spec, params, "mixed", random=rnd, force_mode=1)
        b = O.OracleSystem(spec, params, "mixed", random=rnd, force_mode=1, fmad="altprelude")
        a.step(steps); b.step(steps)
        return a, b
    a, b = both(systems.make_config("C3"), O.Params(temperature=333.0, max_drude_distance=0.02))
    assert np.array_equal(a.velm.view(np.uint8), b.velm.view(np.uint8)) and np.array_equal(a.posq.view(np.uint8), b.posq.view(np.uint8))
    spec = systems.make_config("C5")
    p5 = O.Params(temperature=333.0, max_drude_distance=0.02, mirror_location=float(spec.box[2]) / 2, electric_field=2.0 / float(spec.box[2]) * 2 * 1.602176634e-22)
    a, b = both(spec, p5)
    assert rel(b.velm[:, :3], a.velm[:, :3]) < 1e-7 and rel(b.positions(), a.positions()) < 1e-8
    for name, bound in (("bulk_middle", 1e-5), ("bulk_classic", 1e-3)):
        a, b = both(*make_spec(name))
        ev = rel(b.velm[:, :3], a.velm[:, :3])
        assert 0 < ev < bound, (name, ev)
