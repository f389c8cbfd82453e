import importlib, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, S = pkg.vvhip, pkg.integrator, pkg.systems
spec = S.rigid_water(S.spce_water(300, seed=5))
def run(fused, steps, prec):
    it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.002)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether", tune={"fused": fused})
    it.step(steps)
    out = ctx.getPosq(), ctx.getVelm(), bytes(ctx.getNHState())
    ctx.close()
    return out
for prec in ("single", "mixed"):
    for steps in (1, 2, 3, 12):
        a, b = run(1, steps, prec), run(0, steps, prec)
        dp = np.abs(a[0].astype(np.float64) - b[0]).max(); dv = np.abs(a[1].astype(np.float64) - b[1]).max()
        nv = (a[1] != b[1]).any(axis=1).sum(); npz = (a[0] != b[0]).any(axis=1).sum()
        print(prec, steps, "steps: max |dpos|", dp, "rows", npz, " max |dvel|", dv, "rows", nv, " nh equal", a[2] == b[2])
        if nv and steps == 1:
            idx = np.nonzero((a[1] != b[1]).any(axis=1))[0][:6]
            print("   first differing particles", idx, a[1][idx], b[1][idx])
