import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from oracle import oracle as O
import test_gpu_periodic as T
pkg = importlib.import_module("openmm-velocityverlet_amd"); I = pkg.integrator
class MP:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=False): os.environ.pop(k, None)
mp = MP()
worst = 0; nper = 0
for seed in range(1000, 1160):
    spec = T._random_repeated(seed)
    maxd = 0.02 if len(spec.drude_pairs) else 0.0
    cos = 0.02 if seed % 3 == 0 else 0.0
    middle = seed % 5 != 0
    for env in ({"VVHIP_PERIODIC": "1"}, {"VVHIP_PERIODIC": "0"}):
        flag, v, x, c, ke = T._run(spec, "mixed", 4, env, mp, cos=cos, maxd=maxd, middle=middle)
        nper += flag
        p = O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=maxd, cos_acceleration=cos, use_middle_scheme=middle)
        o = O.OracleSystem(spec, p, "mixed", force_mode=1); o.step(4)
        ev = np.abs(v[:, :3] - o.velm[:, :3]).max() / np.abs(o.velm[:, :3]).max()
        ex = np.abs(x[:, :3].astype(np.float64) - o.posq[:, :3]).max() / np.abs(o.posq[:, :3]).max()
        worst = max(worst, ev)
        if not (ev < 1e-9 and ex < 2e-7): print("MISMATCH", seed, env, ev, ex, spec.num_atoms, flag)
print("fuzz done: 160 systems x 2 layouts, periodic recognised", nper, "worst rel vel err", worst)
