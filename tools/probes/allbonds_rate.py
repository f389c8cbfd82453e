"""Step rate of the full C3 box with constraints=AllBonds / HAngles solved by the general in-wave solver (graph replay, force provider inside),
next to HBonds (hydrogen-type solver) and no constraints, one box."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
def rate(spec, n=4000):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    ctx.run_graph(400, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize(); t = time.perf_counter() - t0
    x = ctx.getPositions(); info = ctx.info
    c, d = np.asarray(spec.constraints).reshape(-1, 2), np.asarray(getattr(spec, "constraint_distances", np.zeros(0)))
    viol = float(np.abs(np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1) - d).max() / d.max()) if len(c) else 0.0
    out = (n / t, info.num_shake_clusters, info.num_general_constraints, info.constraints_fused, viol, ctx.generic_launches()[0])
    ctx.close()
    return out
print("C3 unconstrained          : %8.0f steps/s" % rate(S.make_config("C3"))[0])
r = rate(S.make_config("C3", hbonds=True)); print("C3 HBonds   (%5d clusters): %8.0f steps/s, worst constraint %.1e" % (r[1], r[0], r[4]))
for hang in (False, True):
    t0 = time.time(); spec = S.constrain_all_bonds(S.make_config("C3"), hangles=hang); tb = time.time() - t0
    r = rate(spec)
    print("C3 %s (%6d constraints, general solver; fused %d; built in %.0f s): %8.0f steps/s, worst constraint %.1e, generic launches %s" % ("HAngles " if hang else "AllBonds", r[2], r[3], tb, r[0], r[4], tuple(r[5])))
spec = S.add_virtual_sites(S.make_config("C3"), kinds=(3,), interleaved=False)
r = rate(spec)
print("C3 + a lone pair per molecule (%d virtual sites placed by kernel B): %8.0f steps/s, generic launches %s" % (len(spec.virtual_sites), r[0], tuple(r[5])))
