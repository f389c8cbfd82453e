"""-m gpu: the trajectory does not depend on the host's timing.

VVHIP_STALL=us:period makes every period-th launch of a plan wait on the host first, long enough for the GPU to drain.  Anything that is only
ordered by the depth of the queue -- a fill or a copy enqueued on another stream than the plan's, a host read without a synchronisation --
may then land differently, and the trajectory changes.  One race of this kind existed (a null-stream hipMemset at a switch of the cos
perturbation); it was found and is pinned by the adapter fuzz under run-time compilation, whose second-long stalls sit at
the first launch of every new stage set (tests/test_cpp_plugin.py::test_fuzz_is_insensitive_to_host_stalls) -- the patterns below did not
reproduce it when it was put back, so this file is the broader, cheaper net: sequences of steps, graph replays, queries and parameter changes,
stalled in several patterns and not at all, must give the same bits."""
import importlib
import os

import numpy as np
import pytest

pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
pytestmark = pytest.mark.gpu


def _sequence(spec, middle, cos, hbonds_tol=None):
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    it.setUseMiddleScheme(middle)
    it.setCosAcceleration(cos)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    trace = []
    try:
        it.step(3)
        trace.append(ctx.getKineticEnergy())               # run_a(A_KE_PLAIN) + read + reset of the accumulators, then straight on
        it.step(2)
        if cos:
            trace.append(it.getViscosity()[0])
        trace.extend(ctx.getGroupTemperatures())
        it.setStepSize(0.00125)
        it.step(2)
        ctx.run_graph(8, 4)
        it.setTemperature(350.0)
        trace.append(ctx.getKineticEnergy())
        ctx.run_graph(4, 2)
        if cos:
            it.setCosAcceleration(0.0)                       # the accumulator layout changes: both copies start over from zeros
            it.step(3)
            it.setCosAcceleration(0.5 * cos)
        ctx.setPeriodicBoxSize(*(1.01 * np.asarray(spec.box)))
        it.step(3)
        trace.append(ctx.getKineticEnergy())
        ctx.run_eager(4)
        ctx.synchronize()
        st = ctx.getNHState()
        chain = [x for g in range(3) for x in list(st.eta[g]) + list(st.eta_dot[g])]
        return ctx.getPosq().copy(), ctx.getVelm().copy(), np.array(trace + chain)
    finally:
        ctx.close()


@pytest.mark.parametrize("middle,cos,hbonds", [(True, 0.0, False), (True, 0.02, False), (False, 0.02, False), (False, 0.0, True), (True, 0.0, True)])
def test_trajectories_do_not_depend_on_host_stalls(middle, cos, hbonds):
    spec = S.make_config("C3", 0.03, hbonds=hbonds)
    old = os.environ.pop("VVHIP_STALL", None)
    try:
        ref = _sequence(spec, middle, cos)
        for stall in os.environ.get("VV_TEST_STALLS", "3000:1,2000:3,5000:7").split(","):
            os.environ["VVHIP_STALL"] = stall
            got = _sequence(spec, middle, cos)
            os.environ.pop("VVHIP_STALL")
            for a, b, what in zip(ref, got, ("posq", "velm", "queries and thermostat state")):
                assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), f"{what} differs with VVHIP_STALL={stall}"
    finally:
        os.environ.pop("VVHIP_STALL", None)
        if old is not None:
            os.environ["VVHIP_STALL"] = old
