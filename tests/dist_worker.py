"""Worker for tests/test_distributed.py (launched by torch.distributed.run, backend gloo).

mode "protocol" (CPU): each rank quantises per-block partial sums of ITS shard to int64 fixed point, the ranks all-reduce
them, and every rank must hold exactly the integers a single process gets for the same blocks.
mode "gpu": two ranks share GPU 0; each integrates its molecule-aligned shard with ShardedStepper (accumulators staged through
the host for gloo) and rank 0 compares the stitched trajectory with a single-process run of the whole box."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("openmm-velocityverlet_amd")
S, D = pkg.systems, pkg.distributed


def protocol(rank, world):
    spec = S.drude_il(cells=(1, 1, 1), pairs_per_cell=40, seed=11)
    bounds = D.shard_bounds(spec, world)
    scale = 2.0 ** 30
    ke = spec.masses * (spec.velocities ** 2).sum(1)
    def blocks(b, e):                       # 64-particle "blocks" inside a shard, quantised like csrc/vv_kernels.hip block_accumulate
        q = np.zeros(256, dtype=np.int64)
        for k, s in enumerate(range(b, e, 64)):
            q[k % 256] += np.int64(np.rint(ke[s:min(s + 64, e)].sum() * scale))
        return q
    mine = torch.from_numpy(blocks(*bounds[rank]))
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)
    want = sum(blocks(*bd) for bd in bounds)
    assert np.array_equal(mine.numpy(), want), "int64 all-reduce must be exact"
    total = mine.numpy().sum() / scale
    assert abs(total - ke.sum()) < 1e-6 * ke.sum()
    # order independence: reversed rank order gives the same integers
    assert np.array_equal(sum(blocks(*bd) for bd in reversed(bounds)), want)
    if rank == 0:
        print("PROTOCOL OK", flush=True)


def first_contact(rank, world):
    """bench.py's first-contact report over a real (gloo) process group: two ranks that agree, then two that do not."""
    base = {"device": rank, "device_name": "fake MI355X", "peer_access": [True, True], "rccl_ranks": world, "exchange": "eager", "mailbox_trial": None,
            "world": world, "share_device": False}
    ok, text = D.first_contact_report(dict(base))
    assert ok and "all ranks agree" in text and text.count("rank ") >= world, text
    bad = dict(base, exchange="mailbox" if rank == 1 else "eager")                 # rank 1 fell back to another mechanism
    ok, text = D.first_contact_report(bad)
    assert not ok and "different exchange mechanisms" in text, text
    bad = dict(base, rccl_ranks=1)                                                 # a communicator that does not span the world
    ok, text = D.first_contact_report(bad)
    assert not ok and "does not count 2 ranks" in text, text
    bad = dict(base, exchange="mailbox", peer_access=[True, rank == 0])            # rank 1 cannot reach its peer
    ok, text = D.first_contact_report(bad)
    assert not ok and "hipDeviceCanAccessPeer" in text and "rank(s) [1]" in text, text
    bad = dict(base, device=0)                                                     # both ranks on device 0, not on purpose
    ok, text = D.first_contact_report(bad)
    assert not ok and "two ranks on one device" in text, text
    if rank == 0:
        print("FIRST CONTACT OK", flush=True)


def gpu(rank, world):
    I = pkg.integrator
    spec = S.drude_il(cells=(1, 1, 1), pairs_per_cell=60, seed=13)
    bounds = D.shard_bounds(spec, world)
    nsteps = 10
    def make(shard, cos):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
        it.setMaxDrudeDistance(0.02)
        it.setCosAcceleration(cos)
        return it, I.Context(spec, it, precision="mixed", force_provider="tether", shard=shard, device=0)
    for cos in (0.0, 0.02):
        it, ctx = make(bounds[rank], cos)
        st = D.ShardedStepper(ctx)
        assert st.nphase == 2          # cos: bias moment and group moments leave kernel A together (A_KE_MOM)
        st.step(nsteps)
        x, v = ctx.getPositions(), ctx.getVelocities()
        nh = ctx.getNHState()
        parts = [None] * world
        dist.all_gather_object(parts, (x, v, list(nh.ke2), list(nh.vscale)))
        ctx.close()
        if rank == 0:
            it1, ctx1 = make(None, cos)
            it1.step(nsteps)
            x1, v1, nh1 = ctx1.getPositions(), ctx1.getVelocities(), ctx1.getNHState()
            ctx1.close()
            xs = np.concatenate([p[0] for p in parts]); vs = np.concatenate([p[1] for p in parts])
            assert xs.shape == x1.shape
            ex = np.abs(xs - x1).max() / np.abs(x1).max(); ev = np.abs(vs - v1).max() / np.abs(v1).max()
            assert ex < 1e-11 and ev < 1e-11, (cos, ex, ev)
            for p in parts:                  # every rank advanced the same thermostat
                assert np.allclose(p[2], list(nh1.ke2), rtol=1e-12) and np.allclose(p[3], list(nh1.vscale), rtol=0, atol=1e-13)
            print(f"cos={cos}: sharded == single process (pos {ex:.1e}, vel {ev:.1e})", flush=True)
    if rank == 0:
        print("GPU DIST OK", flush=True)


def mailbox(rank, world):
    """Two processes on GPU 0 exchange their kinetic-energy totals through the IPC mailbox (kernel A publishes, kernel B collects):
    eager steps, hipGraph replay, both schemes, with and without in-kernel constraints -- against one process."""
    I = pkg.integrator
    base = S.drude_il(cells=(1, 1, 1), pairs_per_cell=60, seed=13)
    for label, spec, middle, cos in (("middle", base, True, 0.0), ("classic", base, False, 0.0), ("middle+cos", base, True, 0.02),
                                     ("full-size C3", S.make_config("C3"), True, 0.0),
                                     ("middle+hbonds", S.constrain_hydrogens(S.drude_il(cells=(1, 1, 1), pairs_per_cell=60, seed=13)), True, 0.0),
                                     ("middle+allbonds (general clusters)", S.constrain_all_bonds(S.bulk_Im21(cells=(1, 1, 1), pairs_per_cell=24)), True, 0.0),
                                     ("middle+lone pairs (virtual sites)", S.add_virtual_sites(base, kinds=(3, 0)), True, 0.0)):
        bounds = D.shard_bounds(spec, world)
        def make(shard):
            it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
            it.setMaxDrudeDistance(0.02)
            it.setUseMiddleScheme(middle)
            it.setCosAcceleration(cos)
            return it, I.Context(spec, it, precision="mixed", force_provider="tether", shard=shard, device=0)
        it, ctx = make(bounds[rank])
        handles = [None] * world
        dist.all_gather_object(handles, ctx.mailbox_create(world, rank))
        ctx.mailbox_connect(b"".join(handles))
        assert ctx.mailbox_status() == (True, False)
        dist.barrier()
        it.step(6)                                   # host-launched
        if middle:
            ctx.run_graph(24, steps_per_graph=8)     # replayed: the exchange lives inside the captured kernels
        else:
            it.step(24)
        x, v, nh = ctx.getPositions(), ctx.getVelocities(), ctx.getNHState()
        active, timed_out = ctx.mailbox_status()
        assert active and not timed_out, "a mailbox wait ran out"
        parts = [None] * world
        dist.all_gather_object(parts, (x, v, list(nh.ke2), list(nh.vscale)))
        dist.barrier()
        ctx.mailbox_destroy()
        ctx.close()
        if rank == 0:
            it1, ctx1 = make(None)
            it1.step(30)
            x1, v1, nh1 = ctx1.getPositions(), ctx1.getVelocities(), ctx1.getNHState()
            ctx1.close()
            xs = np.concatenate([p[0] for p in parts]); vs = np.concatenate([p[1] for p in parts])
            # the shard-local wave layout groups the block partial sums differently: last-bit differences of the fixed-point totals
            ex = np.abs(xs - x1).max() / np.abs(x1).max(); ev = np.abs(vs - v1).max() / np.abs(v1).max()
            # (general clusters: a wave sweeps until none of ITS constraints moved, so a molecule's sweep count depends on its wave mates, which
            # the shard-local layout changes: differences of the size of the constraint tolerance)
            tol = 1e-5 if "general" in label else 1e-11
            assert ex < tol and ev < tol, (label, ex, ev)
            for p in parts:
                assert p[2] == parts[0][2] and p[3] == parts[0][3], label          # all ranks: the very same thermostat bits
                assert np.allclose(p[2], list(nh1.ke2), rtol=100 * tol) and np.allclose(p[3], list(nh1.vscale), rtol=0, atol=100 * tol)
            print(f"{label}: mailbox-sharded == single process (pos {ex:.1e}, vel {ev:.1e})", flush=True)
    # a peer that never shows up: the wait is bounded, the failure is reported, nothing hangs
    import time
    bounds = D.shard_bounds(base, world)
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    it.setMaxDrudeDistance(0.02)
    ctx = I.Context(base, it, precision="mixed", force_provider="tether", shard=bounds[rank], device=0)
    handles = [None] * world
    dist.all_gather_object(handles, ctx.mailbox_create(world, rank))
    ctx.mailbox_connect(b"".join(handles))
    if rank == 0:
        t0 = time.perf_counter()
        it.step(3)
        try:                                         # the failure surfaces as an error code on the next host call that looks
            ctx.synchronize()
            raise AssertionError("vvhip_synchronize did not report the timed-out exchange")
        except pkg.vvhip.VVHipError as e:
            assert e.code == pkg.vvhip.ERR_EXCHANGE, e
        waited = time.perf_counter() - t0
        assert ctx.status() == (True, False)
        assert 3.0 < waited < 12.0, waited          # one ~5 s wait, the following steps do not wait again
        for call in (lambda: ctx.run_eager(1), lambda: ctx.run_graph(8, 8)):     # the run loops refuse to continue a void run
            try:
                call()
                raise AssertionError("a run loop started on a plan whose exchange had timed out")
            except pkg.vvhip.VVHipError as e:
                assert e.code == pkg.vvhip.ERR_EXCHANGE, e
        ctx.status_clear()
        assert ctx.status() == (False, False)
        print(f"absent peer: reported after {waited:.1f} s", flush=True)
    dist.barrier()
    ctx.mailbox_destroy()
    ctx.close()
    if rank == 0:
        print("MAILBOX OK", flush=True)


def mailbox_periodic(rank, world):
    """Mailbox exchange NEXT TO the arithmetic work-item layout of kernel B (a box large enough to get one: C3x4 = 444 000 particles,
    222 000 per rank), two processes on GPU 0, against one process.  Round 3 kept the two apart after time-outs in this very set-up;
    round 4's experiment (tools/probes/mailbox_periodic.sh, profiles/r04a_mailbox_periodic.txt) found the cause in the launch shape, not in the
    layout: two processes whose device-filling grids of polling thermostat waves cannot be resident together starve each other, with
    loaded slot words just the same.  vvhip_mailbox_connect now notices ranks that share its device and gives each its share of the CUs."""
    I = pkg.integrator
    os.environ["VVHIP_PERIODIC"] = "1"          # (a shard of 222 000 particles: the arithmetic layout is automatic only from 0.85 M lanes)
    spec = S.make_config("C3", 4.0)
    bounds = D.shard_bounds(spec, world)

    def make(shard):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
        it.setMaxDrudeDistance(0.02)
        return it, I.Context(spec, it, precision="mixed", force_provider="tether", shard=shard, device=0)
    it, ctx = make(bounds[rank])
    assert ctx.info.periodic_layout == 1
    handles = [None] * world
    dist.all_gather_object(handles, ctx.mailbox_create(world, rank))
    ctx.mailbox_connect(b"".join(handles))
    shared, arith = ctx.mailbox_layout()
    assert shared, "two ranks on GPU 0: vvhip_mailbox_connect must notice that the peer's box lives on this device"
    assert arith, "kernel B keeps the arithmetic layout next to the mailbox exchange"
    dist.barrier()
    it.step(6)
    ctx.run_graph(24, steps_per_graph=8)
    x, v, nh = ctx.getPositions(), ctx.getVelocities(), ctx.getNHState()
    active, timed_out = ctx.mailbox_status()
    assert active and not timed_out, "a mailbox wait ran out"
    generic = ctx.generic_launches() if hasattr(ctx, "generic_launches") else None
    parts = [None] * world
    dist.all_gather_object(parts, (x, v, list(nh.ke2), list(nh.vscale)))
    dist.barrier()
    ctx.mailbox_destroy()
    ctx.close()
    if rank == 0:
        it1, ctx1 = make(None)
        it1.step(30)
        x1, v1, nh1 = ctx1.getPositions(), ctx1.getVelocities(), ctx1.getNHState()
        ctx1.close()
        xs = np.concatenate([p[0] for p in parts]); vs = np.concatenate([p[1] for p in parts])
        ex = np.abs(xs - x1).max() / np.abs(x1).max(); ev = np.abs(vs - v1).max() / np.abs(v1).max()
        assert ex < 1e-11 and ev < 1e-11, (ex, ev)
        for p in parts:
            assert p[2] == parts[0][2] and p[3] == parts[0][3]
            assert np.allclose(p[2], list(nh1.ke2), rtol=1e-12) and np.allclose(p[3], list(nh1.vscale), rtol=0, atol=1e-13)
        print(f"MAILBOX PERIODIC OK ({spec.num_atoms} particles, pos {ex:.1e}, vel {ev:.1e}, generic launches {generic})", flush=True)


if __name__ == "__main__":
    dist.init_process_group(backend="gloo")
    r, w = dist.get_rank(), dist.get_world_size()
    {"protocol": protocol, "first_contact": first_contact, "gpu": gpu, "mailbox": mailbox, "mailbox_periodic": mailbox_periodic}[sys.argv[1]](r, w)
    dist.barrier()
    dist.destroy_process_group()
