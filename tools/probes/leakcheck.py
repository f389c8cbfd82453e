"""Plan / context life cycle: free device memory before and after 300 create -> run -> destroy cycles (incl. graphs, mailbox, constraints)."""
import importlib, sys
import torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
spec = S.constrain_hydrogens(S.drude_il(cells=(1, 1, 1), pairs_per_cell=100, seed=1))
def cycle(k):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02); it.setCosAcceleration(0.02 if k % 3 == 0 else 0.0)
    ctx = I.Context(spec, it, precision=("mixed", "single", "double")[k % 3], force_provider="tether")
    if k % 4 == 0:
        h = ctx.mailbox_create(1, 0); ctx.mailbox_connect(h)
    ctx.run_graph(16, 8); it.step(2)
    ctx.close()
cycle(0); torch.cuda.synchronize()
f0 = torch.cuda.mem_get_info()[0]
for k in range(300): cycle(k)
torch.cuda.synchronize()
f1 = torch.cuda.mem_get_info()[0]
print(f"free before {f0 / 2**20:.1f} MiB, after 300 cycles {f1 / 2**20:.1f} MiB, difference {(f0 - f1) / 2**20:.2f} MiB")
assert f0 - f1 < 64 * 2**20
