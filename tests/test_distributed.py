"""N > 1 path: world_size-2 runs over gloo (CPU: the fixed-point exchange protocol; GPU: two ranks sharing the one GPU,
sharded trajectory vs single process).  On the 8-GPU node the same ShardedStepper runs over nccl (= RCCL, xGMI)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _launch(mode, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), WORKER, mode]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)


def test_fixed_point_exchange_protocol_gloo_world2():
    r = _launch("protocol", 29541)
    assert r.returncode == 0 and "PROTOCOL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_first_contact_report_gloo_world2():
    """The report bench.py writes before the first timed region of a multi-GPU run, over a real two-rank gloo group: agreement passes,
    and each kind of disagreement (exchange mechanism, communicator size, peer access, two ranks on one device) is named."""
    r = _launch("first_contact", 29547)
    assert r.returncode == 0 and "FIRST CONTACT OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_sharded_step_equals_single_process_two_ranks_one_gpu():
    r = _launch("gpu", 29542)
    assert r.returncode == 0 and "GPU DIST OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_mailbox_exchange_two_ranks_one_gpu():
    """The xGMI mailbox path (hipIpc mappings, kernel A publishes / kernel B collects, hipGraph replay) with two processes."""
    r = _launch("mailbox", 29544)
    assert r.returncode == 0 and "MAILBOX OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_mailbox_next_to_the_arithmetic_layout_two_ranks_one_gpu():
    """Round-3 verdict item: 0.44 M particles sharded over two processes that share the GPU -- kernel B with computed slot words AND the
    mailbox exchange, grids capped to each rank's share of the CUs -- equals the single-process run to 1e-11, no wait runs out."""
    r = _launch("mailbox_periodic", 29546)
    assert r.returncode == 0 and "MAILBOX PERIODIC OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["mailbox", "graph", "eager", "python"])
def test_nccl_code_path_world1(mode):
    """The exact multi-GPU path of bench.py (process group over nccl = RCCL, accumulator tensor aliasing the plan's
    device memory, all-reduce between kernel A and kernel B on torch's stream) with a single rank."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-mode", mode, "--steps", "300", "--warmup", "30",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["value"] > 100 and line["n_gpus"] == 1
    assert mode in line["config"]["parallelism"], line["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_mailbox():
    """bench.py's own N = 2 logic (handle gather, trial run, agreement between ranks, graph replay) with two processes on GPU 0;
    gloo carries the set-up traffic because RCCL refuses two ranks on one device.  Also the second series of a scaling run: the large box
    (here C3x8) sharded over the same ranks with the same exchange (config.large_n_sharded)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29545", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
           "--steps", "400", "--warmup", "100", "--no-cpu-baseline", "--large-n", "C3x8"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 100
    assert "mailbox" in line["config"]["parallelism"], line["config"]["parallelism"] + r.stderr[-2000:]
    # the large box's capped grid fills the ONE shared GPU with polling thermostat waves, the other process's kernels cannot be scheduled and
    # the bounded waits run out: the block must then be skipped on both ranks (and say so), not hang or report a number
    ls = line["config"]["large_n_sharded"]
    assert (isinstance(ls, dict) and ls["n_gpus"] == 2 and ls["steps_per_s"] > 1) or ls == "skipped (see stderr)", (ls, r.stderr[-2000:])
    # round 5: BASELINE.json's configs[3] -- the same box with the cos perturbation -- as a series of its own next to the headline
    c4 = line["config"]["c4_sharded"]
    assert isinstance(c4, dict) and c4["n_gpus"] == 2 and c4["steps_per_s"] > 100 and c4["exchange"] == "mailbox" and "cos acceleration" in c4["workload"], (c4, r.stderr[-2000:])


@pytest.mark.gpu
def test_bench_second_series_large_box_sharded_two_ranks():
    """config.large_n_sharded through the exchange that works with two ranks on one GPU (accumulators staged through the host, gloo):
    the large box sharded over the ranks of the run, the second series of a scaling curve."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29565", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--dist-mode", "python",
           "--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--large-n", "C3x8"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    ls = line["config"]["large_n_sharded"]
    assert isinstance(ls, dict) and ls["n_gpus"] == 2 and ls["steps_per_s"] > 1 and ls["exchange"] == "python", (ls, r.stderr[-2000:])
    assert "888000 particles" in ls["workload"]


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("cfg", ["C3", "C4"])
def test_shard_plans_for_the_scaling_run_host_only(cfg, world):
    """What `bench.py --gpus N` builds on every rank, without a GPU: molecule-aligned shards of the headline box, one plan per
    rank with GLOBAL thermostat constants (DOF, chain masses, total mass), every particle in exactly one shard, and the exchange
    payload the mailbox carries (3 totals; 10 with the cos perturbation in its moment form = 20 words)."""
    import importlib
    import numpy as np
    pkg = importlib.import_module("openmm-velocityverlet_amd")
    I, S, D = pkg.integrator, pkg.systems, pkg.distributed
    spec = S.make_config(cfg)
    bounds = D.shard_bounds(spec, world)
    assert bounds[0][0] == 0 and bounds[-1][1] == spec.num_atoms and all(b[1] == c[0] for b, c in zip(bounds, bounds[1:]))
    sizes = np.array([e - b for b, e in bounds])
    assert sizes.min() > 0 and (sizes.max() - sizes.min()) <= 2 * 27 + 10          # balanced to within a couple of ions
    mol = np.asarray(spec.mol_id)
    for b, e in bounds[:-1]:
        assert mol[e - 1] != mol[e]                                               # never inside a molecule (nor a Drude pair / COM group)
    infos, used = [], 0
    for r in range(world):
        it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
        it.setMaxDrudeDistance(0.02)
        if cfg == "C4":
            it.setCosAcceleration(0.02)
        plan, info, _ = I.create_plan(spec, it, shard=bounds[r])
        infos.append((list(info.dof), list(info.nkbt), [list(info.eta_mass[g])[:3] for g in range(3)], info.inv_mass_total, info.num_temp_groups))
        used += info.num_slots_used
        assert pkg.vvhip.lib.vvhip_step_middle_phases(plan) == 2                   # kick+sums | exchange | chain+scale+drift, also with cos
        pkg.vvhip.lib.vvhip_plan_destroy(plan)
    assert used == spec.num_atoms
    assert all(i == infos[0] for i in infos)
    assert np.allclose(infos[0][0], [198000.0, 17997.0, 117000.0], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("world,cfg", [(4, "C3"), (8, "C3"), (4, "C4"), (8, "C4")])
def test_bench_rank_counts_of_the_scaling_run_on_one_gpu(world, cfg):
    """bench.py's N = 4 and N = 8 logic (shard bounds, handle gather, mailbox trial, agreement, replayed graphs; with C4 the
    20-word payload of the cos moments) with all ranks on GPU 0 -- a dry run of the driver's scaling command for rank-count bugs.
    gloo carries the set-up traffic because RCCL refuses several ranks on one device; the exchange itself is the mailbox (hipIpc)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29550 + world + (1 if cfg == "C4" else 0)), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo",
           "--share-device", "--config", cfg, "--steps", "200", "--warmup", "40", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    # speed means nothing here (N processes time-slice one GPU and wait for each other's kernels: tens of steps/s); the run must
    # complete with every rank's thermostat bits equal -- bench.py exits non-zero otherwise
    assert line["n_gpus"] == world and line["value"] > 1 and line["steps"] == 200, line
    assert f"x{world}" in line["config"]["parallelism"]
    # several processes time-slicing one GPU may lose the mailbox trial (its waits are bounded); then the run falls back to the
    # per-step collective -- either way every rank finished with identical thermostat bits (bench.py checks that itself)
    assert any(k in line["config"]["parallelism"] for k in ("mailbox", "python", "eager"))


@pytest.mark.gpu
def test_distributed_code_path_costs_nothing_at_one_rank():
    """`bench.py --force-dist` at N = 1 (process group, exchange set up and chosen as at N > 1, sharded plan of the whole box) against
    the plain N = 1 run on the same GPU: the day a scaling curve is measured its N = 1 point must be the headline figure, not a slower
    code path.  Also checks what the line says about the exchange (config.exchange: chosen mechanism, RCCL's own rank count)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", HSA_ENABLE_IPC_MODE_LEGACY="0")
    vals = {}
    for label, extra in (("plain", []), ("dist", ["--force-dist"]), ("plain2", []), ("dist2", ["--force-dist"]), ("plain_long", []), ("dist_long", ["--force-dist"])):
        steps, warmup = ("4000", "400") if label.endswith("long") else ("20", "5")          # the driver's flags; one long run beside them
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", steps, "--warmup", warmup, "--no-cpu-baseline",
                            "--no-rocprof", "--large-n", "none"] + extra, capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
        vals[label] = line["value"]
        if extra:
            ex = line["config"]["exchange"]
            assert ex["chosen"] in ("mailbox", "eager", "graph", "python") and ex["process_group_ranks"] == 1
            assert ex["rccl_ranks"] in (0, 1) and isinstance(ex["log"], dict)
    plain, distv = max(vals["plain"], vals["plain2"]), max(vals["dist"], vals["dist2"])
    assert distv > 0.93 * plain, vals
    # the 4000-step form: 0.95-0.96 over a dozen boxes (kernel for kernel the two runs take the same time in rocprofv3's trace: B with the
    # exchange 5.63 us, without 5.64 us; round 5, one launch per step: the mailbox words are one more wait of ~0.2 us in a 10 us step)
    assert vals["dist_long"] > 0.90 * vals["plain_long"], vals


@pytest.mark.gpu
def test_bench_two_ranks_large_box_on_one_gpu():
    """The regime in which sharding pays (0.9 M particles: arithmetic work-item layout, capped grids, stand-alone chain launch) through
    bench.py's N = 2 logic with both ranks on GPU 0 -- a dry run of `--gpus N --config C3x8` for the scaling curve's second series."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29563", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--config", "C3x8",
           "--steps", "60", "--warmup", "20", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 1 and "x2" in line["config"]["parallelism"]
    assert line["config"]["exchange"]["chosen"] in ("mailbox", "python", "eager")
