"""Long-run sanity of the thermostats on the headline box: group temperatures (atom, COM, Drude) over 400 000 steps."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
for hb in (False, True):
    spec = S.make_config("C3", hbonds=hb)
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    t0 = time.perf_counter()
    print("constraints" if hb else "no constraints")
    for k in range(8):
        ctx.run_graph(50000, 100); ctx.synchronize()
        T = ctx.getGroupTemperatures()
        x = ctx.getPositions()
        d = spec.drude_pairs
        r = np.linalg.norm(x[d[:, 0]] - x[d[:, 1]], axis=1).max()
        msg = f"  step {(k + 1) * 50000:7d}: T_atom {T[0]:7.2f}  T_com {T[1]:7.2f}  T_drude {T[2]:6.3f} K   max Drude distance {r:.4f} nm  finite {np.isfinite(x).all()}"
        if hb:
            c, dist = np.asarray(spec.constraints), np.asarray(spec.constraint_distances)
            rr = np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1)
            msg += f"  max |r-d|/d {np.abs(rr - dist).max() / dist[0]:.1e}"
        print(msg, flush=True)
    print(f"  {400000 / (time.perf_counter() - t0):.0f} steps/s sustained")
    ctx.close()
