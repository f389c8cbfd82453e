"""-m gpu: the product (fused HIP path through the C-ABI) against trajectories and thermostat constants recorded from the REFERENCE's
whole step -- its own VVIntegrator.cpp + CudaVVKernels.cpp + kernels/*.cu run on the CPU (tests/golden/refhost_*.npz, written by
oracle/make_golden_refhost.py in the build container) -- on the same seeded systems, static forces and injected normals.  No oracle in
between: reference output vs product output.  Tolerance after 6 steps: 1e-9 relative in mixed / double precision (measured ~1e-15),
1e-5 in single (BASELINE.json north_star); degrees of freedom, N kB T and thermostat masses bit for bit.
"""
import importlib
import os

import numpy as np
import pytest

from oracle.make_golden_refhost import CONFIGS, FULL, FULL_STEPS, GOLDEN, inputs_for, make_spec, precisions_of

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I = pkg.vvhip, pkg.integrator
pytestmark = pytest.mark.gpu
TOL = {"single": 1e-5, "mixed": 1e-9, "double": 1e-9}


def _integrator(p):
    it = I.VVIntegrator(p.temperature, p.frequency, p.drude_temperature, p.drude_frequency, p.step_size, p.num_chains, p.loops_per_step)
    it.setMaxDrudeDistance(p.max_drude_distance)
    if not p.auto_set_friction:
        it.setFriction(p.friction)
    it.setMirrorLocation(p.mirror_location)
    it.setElectricField(p.electric_field)
    it.setCosAcceleration(p.cos_acceleration)
    it.setUseMiddleScheme(p.use_middle_scheme)
    if not p.auto_set_com_temp_group:
        it.setUseCOMTempGroup(p.use_com_temp_group)
    return it


@pytest.mark.parametrize("prec", ["single", "mixed", "double"])
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_product_reproduces_reference_pipeline(name, prec):
    if prec not in precisions_of(name):
        pytest.skip("no reference run exists for image pairs outside mixed precision")
    g = np.load(os.path.join(GOLDEN, f"refhost_{name}.npz"))
    steps = int(g["steps"])
    spec, params = make_spec(name)
    rnd, force = inputs_for(spec, params, steps)
    it = _integrator(params)
    ctx = I.Context(spec, it, precision=prec, force_provider="static", random=rnd)
    try:
        ctx.force.upload(force)
        # the constants the reference's CudaModifyDrudeNoseKernel::initialize derived (CudaVVKernels.cpp:505-594)
        ntg = int(g["num_tg"])
        if ntg:
            assert ctx.info.num_temp_groups == ntg
            assert np.array_equal(np.array(list(ctx.info.dof)), g["dof"])
            assert np.array_equal(np.array(list(ctx.info.nkbt))[:ntg], g["nkbt"][:ntg])
            em = np.array([list(r) for r in ctx.info.eta_mass])
            assert np.array_equal(em[:ntg, :params.num_chains], g["eta_mass"][:ntg, :params.num_chains])
        it.step(steps)
        velm, posq = ctx.getVelm(), ctx.getPosq()
        rv, rp = g[f"velm_{prec}"], g[f"posq_{prec}"]
        massive = rv[:, 3] != 0
        ev = np.abs(velm[massive, :3].astype(np.float64) - rv[massive, :3]).max() / np.abs(rv[massive, :3]).max()
        ep = np.abs(posq[:, :3].astype(np.float64) - rp[:, :3]).max() / np.abs(rp[:, :3]).max()
        ptol = TOL[prec] if prec == "double" else max(TOL[prec], 1.2e-7)     # posq is float outside double precision: half an ulp
        assert ev < TOL[prec] and ep < ptol, f"{name}/{prec}: rel err vel {ev:.2e} pos {ep:.2e}"
        assert np.array_equal(velm[:, 3], rv[:, 3]) and np.array_equal(posq[:, 3], rp[:, 3])      # inverse masses / charges untouched
        print(f"{name}/{prec}: vs the reference's own pipeline, rel err vel {ev:.2e} pos {ep:.2e}")
    finally:
        ctx.close()


@pytest.mark.parametrize("name", sorted(FULL))
def test_product_reproduces_reference_pipeline_at_full_baseline_size(name):
    """The BASELINE configurations at full size on the reference's own topologies: product (fused path, hipGraph-free stepping) against a run
    of the reference's whole step -- every 97th particle, the group kinetic energies, the scale factors, the velocity sum."""
    g = np.load(os.path.join(GOLDEN, f"refhost_{name}.npz"))
    spec, params = FULL[name]()
    rnd, force = inputs_for(spec, params, FULL_STEPS)
    it = _integrator(params)
    ctx = I.Context(spec, it, precision="mixed", force_provider="static", random=rnd)
    try:
        ctx.force.upload(force)
        ntg = int(g["num_tg"])
        assert ctx.info.num_temp_groups == ntg and np.array_equal(np.array(list(ctx.info.dof)), g["dof"])
        assert np.array_equal(np.array(list(ctx.info.nkbt))[:ntg], g["nkbt"][:ntg])
        it.step(int(g["steps"]))
        velm, posq, corr = ctx.getVelm(), ctx.getPosq(), ctx.getPosqCorrection()
        idx = g["index"]
        rv, rp, rc = g["velm"], g["posq"], g["posq_corr"]
        massive = rv[:, 3] != 0
        ev = np.abs(velm[idx][massive, :3] - rv[massive, :3]).max() / np.abs(rv[massive, :3]).max()
        x = posq[idx][:, :3].astype(np.float64) + corr[idx][:, :3].astype(np.float64)
        xr = rp[:, :3].astype(np.float64) + rc[:, :3].astype(np.float64)
        ex = np.abs(x - xr).max() / np.abs(xr).max()
        assert ev < 1e-9 and ex < 1e-9, f"{name}: rel err vel {ev:.2e} pos {ex:.2e}"
        st = ctx.getNHState()
        assert np.allclose(np.array(list(st.ke2))[:ntg], g["ke2"][:ntg], rtol=1e-10)
        assert np.allclose(np.array(list(st.vscale))[:ntg], g["vscale"][:ntg], rtol=0, atol=1e-12)
        sv = velm[:, :3].sum(axis=0)
        assert np.allclose(sv, g["sum_velm"], rtol=0, atol=1e-9 * np.abs(velm[:, :3]).sum())
        print(f"{name}: vs the reference's own pipeline at full size, rel err vel {ev:.2e} pos {ex:.2e}")
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["bulk_middle_cos", "bulk_classic_cos"])
def test_product_follows_the_reference_through_parameter_changes(name):
    """The cos acceleration switched off (the reference keeps applying the last extra force: its kick kernels add forceExtra always and
    nothing resets the array any more, VVIntegrator.cpp:238-240 / kernels/middle.cu:11-21), on again with another value, a box change,
    another step size -- recorded from the reference's own pipeline (tests/golden/refhost_switch_*.npz), replayed on the fused path."""
    from tests.test_ref_host import run_switch_sequence
    g = np.load(os.path.join(GOLDEN, f"refhost_switch_{name}.npz"))
    spec, params = make_spec(name)
    rnd, force = inputs_for(spec, params, 12)
    it = _integrator(params)
    ctx = I.Context(spec, it, precision="mixed", force_provider="static", random=rnd)
    try:
        ctx.force.upload(force)

        def setter(what, val):
            if what == "cos_acceleration": it.setCosAcceleration(val)
            elif what == "step_size": it.setStepSize(val)
            elif what == "box": ctx.setPeriodicBoxSize(val, val, val)
        run_switch_sequence(it.step, setter)
        velm, posq, corr = ctx.getVelm(), ctx.getPosq(), ctx.getPosqCorrection()
        rv = g["velm"]
        ev = np.abs(velm[:, :3] - rv[:, :3]).max() / np.abs(rv[:, :3]).max()
        x = posq[:, :3].astype(np.float64) + corr[:, :3].astype(np.float64)
        xr = g["posq"][:, :3].astype(np.float64) + g["posq_corr"][:, :3].astype(np.float64)
        ex = np.abs(x - xr).max() / np.abs(xr).max()
        assert ev < 1e-9 and ex < 1e-9, f"{name}: rel err vel {ev:.2e} pos {ex:.2e}"
        print(f"{name}: through the parameter changes, rel err vel {ev:.2e} pos {ex:.2e}")
    finally:
        ctx.close()
