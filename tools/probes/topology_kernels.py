"""Eager steps (no graph: rocprofv3's kernel trace and replayed graphs do not get along on this image) of the full C3 box without constraints, with
HBonds, AllBonds, HAngles and with a lone pair per molecule: run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations of the stage
sets behind these topologies."""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems
base = S.make_config("C3")
for name, spec in (("plain", base), ("hbonds", S.make_config("C3", hbonds=True)), ("allbonds", S.constrain_all_bonds(S.make_config("C3"))),
                   ("hangles", S.constrain_all_bonds(S.make_config("C3"), hangles=True)), ("lone pairs", S.add_virtual_sites(base, kinds=(3,), interleaved=False))):
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001); it.setMaxDrudeDistance(0.02)
    ctx = I.Context(spec, it, precision="mixed", force_provider="tether")
    it.step(300)
    ctx.synchronize()
    a = H = None
    print(name, "A 0x%x B 0x%x" % (ctx.fused_flags(0), ctx.fused_flags(1)) if hasattr(ctx, "fused_flags") else "")
    ctx.close()
