"""Step rate of the full C3 box with AllBonds / HAngles constraints against the relaxation factor of the general clusters' sweeps (test hook key
gc_omega_permille), and the worst constraint at the end: what vv_layout.h's GC_OMEGA_* were chosen from."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
def rate(spec, omega, n=2000):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether", tune={"gc_omega_permille": int(round(omega * 1000))})
    ctx.run_graph(200, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(n, 100); ctx.synchronize(); t = time.perf_counter() - t0
    x = ctx.getPositions(); c, d = np.asarray(spec.constraints), np.asarray(spec.constraint_distances)
    viol = float(np.abs(np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1) - d).max() / d.max())
    ctx.close()
    return n / t, viol
for hang, omegas in ((False, (1.0, 1.05, 1.1, 1.15, 1.2, 1.25, 1.3)), (True, (1.0, 1.2, 1.3, 1.35, 1.4, 1.45, 1.5, 1.6))):
    spec = S.constrain_all_bonds(S.make_config("C3"), hangles=hang)
    print("HAngles" if hang else "AllBonds", " ".join("%.2f: %.0f (%.0e)" % ((w,) + rate(spec, w)) for w in omegas))
