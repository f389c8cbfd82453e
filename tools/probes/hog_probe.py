import importlib, os, subprocess, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo") else os.getcwd())
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_recovery as T
hog = T._hog_binary()
spec = S.make_config("C3")
for blocks in (224, 128, 64, 32, 8):
    ctx = T._context(spec, fused=True, recover=0)
    ctx.run_graph(400, 100); ctx.synchronize()
    t0 = time.perf_counter(); ctx.run_graph(60000, 100); ctx.synchronize(); base = time.perf_counter() - t0
    proc = subprocess.Popen([hog, str(blocks), "0.6", "0.15"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    print(proc.stdout.readline().strip())
    t0 = time.perf_counter(); proc.stdin.write("go\n"); proc.stdin.flush()
    ctx.run_graph(60000, 100); t1 = time.perf_counter() - t0
    l = proc.stdout.readline().strip(); t2 = time.perf_counter() - t0
    err = None
    try: ctx.synchronize()
    except Exception as e: err = str(e)[:60]
    t3 = time.perf_counter() - t0
    d = proc.stdout.readline().strip(); t4 = time.perf_counter() - t0
    proc.stdin.close(); proc.wait()
    print(f"hog {blocks} blocks: undisturbed run {base:.3f} s; enqueue {t1:.3f}, hog {l} at {t2:.3f}, our sync returned at {t3:.3f} ({err}), hog {d} at {t4:.3f}; status {ctx.status_words()}", flush=True)
    ctx.close()
