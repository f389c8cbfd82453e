"""Where does kernel B's time go at the headline size?  Times stage subsets through vvhip_time_kernel (100 back-to-back launches)."""
import importlib, sys, statistics
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S, H = pkg.integrator, pkg.systems, pkg.vvhip
import ctypes as C
spec = S.make_config("C3")
it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
ctx.run_graph(200, 100); ctx.synchronize()
B = dict(SCALE=1, DRIFT=16, POS3=64, HW=512, CHAIN=2048)
A = dict(KICK=32, KE=1024)
def t(kernel, flags):
    vals = []
    for _ in range(5):
        ms = C.c_double(0)
        H.check(H.lib.vvhip_time_kernel(ctx.plan, kernel, flags, 100, C.byref(ms)), ctx.plan)
        vals.append(ms.value * 1e3)
    return statistics.median(vals)
rows = [("B full (specialised)", 1, B["CHAIN"] | B["SCALE"] | B["DRIFT"] | B["HW"]),
        ("B no chain wave (scales from memory)", 1, B["SCALE"] | B["DRIFT"] | B["HW"]),
        ("B chain+scale+drift (no hard wall)", 1, B["CHAIN"] | B["SCALE"] | B["DRIFT"]),
        ("B drift only", 1, B["DRIFT"]),
        ("B pos3 only (load/store skeleton)", 1, B["POS3"]),
        ("A full (kick + KE)", 0, A["KICK"] | A["KE"]),
        ("A kick only", 0, A["KICK"]),
        ("A KE only", 0, A["KE"])]
for name, k, f in rows:
    print(f"{name:42s} {t(k, f):6.2f} us", flush=True)
ctx.close()
