"""Host logic of the product (vvhip_plan_create: no GPU needed) against the numpy restatement of the
reference's initialize() methods, the wave-layout invariants the kernels rely on, and the C-ABI export list."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

from oracle import oracle as O

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, systems = pkg.vvhip, pkg.integrator, pkg.systems
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _integrator(p: O.Params):
    it = I.VVIntegrator(p.temperature, p.frequency, p.drude_temperature, p.drude_frequency, p.step_size, p.num_chains, p.loops_per_step)
    it.setMaxDrudeDistance(p.max_drude_distance)
    it._cosAcceleration = p.cos_acceleration
    return it


@pytest.mark.parametrize("cfg,scale", [("C1", 1.0), ("C2", 0.1), ("C3", 0.02), ("C3", 1.0), ("C5", 0.05)])
def test_analysis_matches_reference_init(cfg, scale):
    spec = systems.make_config(cfg, scale)
    p = O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02)
    t = O.build_tables(spec, p)
    info, slots = I.plan_layout(spec, _integrator(p))
    assert info.num_particles_nh == len(t["particles_nh"]) and info.num_molecules_nh == len(t["molecules_nh"])
    assert info.num_normal_nh == len(t["normal_nh"]) and info.num_pairs_nh == len(t["pairs_nh"])
    assert info.num_normal_ld == len(t["normal_ld"]) and info.num_pairs_ld == len(t["pairs_ld"])
    assert info.num_temp_groups == t["num_tg"] and bool(info.use_com_temp_group) == t["params"].use_com_temp_group
    assert info.friction == t["params"].friction
    assert list(info.dof) == list(t["dof"])                       # same summation order => bit-equal
    assert list(info.nkbt) == list(t["nkbt"])
    assert np.array_equal(np.array([list(r) for r in info.eta_mass]), t["eta_mass"])
    assert info.inv_mass_total == pytest.approx(t["inv_mass_total"], rel=1e-15)

    # ---- wave layout invariants
    atoms, meta = slots[:, 0], slots[:, 1].astype(np.uint32)
    used = atoms >= 0
    massive = spec.masses != 0
    assert np.array_equal(np.sort(atoms[used]), np.nonzero(massive | np.isin(np.arange(spec.num_atoms), [par for _, par in spec.image_pairs]))[0])
    lane = np.arange(slots.shape[0]) % 64
    wave = np.arange(slots.shape[0]) // 64
    slot_of = -np.ones(spec.num_atoms, dtype=np.int64)
    slot_of[atoms[used]] = np.nonzero(used)[0]
    role = meta & 0xF
    partner = (meta >> 4) & 63
    for d, par in spec.drude_pairs:
        sd, sp = slot_of[d], slot_of[par]
        assert wave[sd] == wave[sp], "a Drude pair must share a wave"
        assert partner[sd] == lane[sp] and partner[sp] == lane[sd]
        assert role[sd] in (3, 6) and role[sp] in (4, 7)
    if info.use_com_temp_group:
        first, last = (meta >> 10) & 63, (meta >> 16) & 63
        nh = set(t["particles_nh"].tolist())
        for m in np.unique(spec.mol_id):
            members = [i for i in np.nonzero(spec.mol_id == m)[0] if massive[i] and i in nh]
            if not members:
                continue
            s = slot_of[members]
            assert len(set(wave[s])) == 1, "a molecule must share a wave"
            assert (first[s] == lane[s].min()).all() and (last[s] == lane[s].max()).all()
            assert ((meta[s] >> 24) & 1).sum() == 1, "exactly one COM leader per molecule"
            if m > 40:
                break


def test_reference_error_cases_keep_their_messages():
    spec = systems.make_config("C5", 0.02)
    it = I.VVIntegrator(333, 10, 1, 40, 0.001)
    it.setCosAcceleration(0.02)
    with pytest.raises(H.VVHipError, match="Langevin thermostat and periodic perturbation shouldn't be used together") as e:
        I.plan_layout(spec, it)
    assert e.value.code == H.ERR_TOPOLOGY
    it = I.VVIntegrator(333, 10, 1, 40, 0.001)
    it.addParticleLangevin(spec.image_pairs[0][1])
    with pytest.raises(H.VVHipError, match="NH and Langevin thermostat cannot be applied on the same molecule"):
        I.plan_layout(spec, it)
    spec = systems.make_config("C3", 0.01)
    it = I.VVIntegrator(333, 10, 1, 40, 0.001)
    it.addParticleLangevin(int(spec.drude_pairs[0][0]))
    with pytest.raises(H.VVHipError):
        I.plan_layout(spec, it)
    spec.constraints = np.array([[0, spec.num_atoms - 1]], np.int32)
    it = I.VVIntegrator(333, 10, 1, 40, 0.001)
    for i in np.nonzero(spec.mol_id == spec.mol_id[0])[0]:
        it.addParticleLangevin(int(i))
    with pytest.raises(H.VVHipError, match="Constrained particle pair should be in the same thermostat"):
        I.plan_layout(spec, it)


def test_shards_are_molecule_aligned_and_cover_everything():
    spec = systems.make_config("C3", 0.02)
    it = I.VVIntegrator(333, 10, 1, 40, 0.001)
    dist = importlib.import_module("openmm-velocityverlet_amd.distributed")
    bounds = dist.shard_bounds(spec, 4)
    assert bounds[0][0] == 0 and bounds[-1][1] == spec.num_atoms
    total = 0
    for b, e in bounds:
        info, slots = I.plan_layout(spec, it, shard=(b, e))
        a = slots[:, 0]
        assert a[a >= 0].min() == 0 and a.max() == e - b - 1
        total += info.num_slots_used
        full, _ = I.plan_layout(spec, it)
        assert list(info.dof) == list(full.dof)          # thermostat constants are global, not per shard
    assert total == spec.num_atoms
    with pytest.raises(H.VVHipError, match="cuts a"):
        I.plan_layout(spec, it, shard=(0, bounds[0][1] - 1))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vvhip.h")).read()
    declared = set(re.findall(r"\b(vvhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 45
    import ctypes
    lib = ctypes.CDLL(H.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in include/vvhip.h but not exported: {missing}"


def test_no_gpu_means_loud_failure_not_fallback():
    if H.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(H.VVHipError) as e:
        I.Context(systems.make_config("C1"), I.VVIntegrator(300, 10, 1, 40, 0.001))
    assert e.value.code == H.ERR_NO_DEVICE


def test_big_molecule_is_cut_into_wave_sized_chunks():
    n = 150
    spec = systems.SystemSpec(name="big", masses=np.full(n, 12.0), charges=np.zeros(n), positions=np.zeros((n, 3)), velocities=np.zeros((n, 3)),
                              box=np.ones(3), mol_id=np.zeros(n, np.int32), drude_pairs=np.zeros((0, 2), np.int32),
                              constraints=np.zeros((0, 2), np.int32))
    it = I.VVIntegrator(300, 10, 1, 40, 0.001)
    it.setUseCOMTempGroup(True)
    info, slots = I.plan_layout(spec, it)
    meta = slots[:, 1].astype(np.uint32)
    used = slots[:, 0] >= 0
    assert info.num_slots_used == n and info.num_waves == 3 and info.max_cluster <= 64
    assert ((meta[used] >> 28) & 1).all()                     # every lane knows it belongs to a big molecule
    assert ((meta >> 29) & 1).sum() == 1                      # exactly one lane adds the molecule's M V^2 / clears nothing twice
    assert ((meta >> 24) & 1).sum() == 3                      # one chunk leader per wave
    assert info.dof[1] == 0.0                                 # 3*1 molecule - 3 (CMMotionRemover) = 0 -> no COM group DOF


@pytest.mark.parametrize("use_com", [None, False])
def test_constraint_clusters_share_a_wave(use_com):
    """In-kernel SHAKE needs every constraint cluster inside one 64-lane wave (and, without the COM group, together with the
    Drude pairs hanging off its members).  Cluster admission follows OpenMM's SHAKE rule; the oracle restates it independently."""
    spec = systems.constrain_hydrogens(systems.drude_il(cells=(1, 1, 1), pairs_per_cell=25, seed=2))
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    if use_com is not None:
        it.setUseCOMTempGroup(use_com)
    info, slots = I.plan_layout(spec, it)
    atoms, params = O.build_shake(spec)
    assert info.constraints_fused and info.num_shake_clusters == len(atoms)
    wave_of = np.full(spec.num_atoms, -1)
    live = slots[:, 0] >= 0
    wave_of[slots[live, 0]] = np.nonzero(live)[0] // 64
    for row in atoms:
        members = row[row >= 0]
        assert len(set(wave_of[members])) == 1 and wave_of[members[0]] >= 0
    for d, par in np.asarray(spec.drude_pairs):
        assert wave_of[d] == wave_of[par]
    # a chain of constraints (H-O-H plus a fourth particle hanging off one H) is neither a triangle nor such a cluster: it becomes a
    # general cluster (coloured sweeps inside the wave), not silently dropped
    spec2 = systems.spce_water(4)
    spec2.constraints = np.array([(1, 0), (2, 0), (2, 3)], dtype=np.int32)
    spec2.constraint_distances = np.array([0.1, 0.1, 0.16])
    info2, _ = I.plan_layout(spec2, I.VVIntegrator(300.0, 10, 1.0, 40, 0.001))
    assert info2.constraints_fused and info2.num_shake_clusters == 0 and info2.num_settle_clusters == 0
    assert info2.num_general_constraints == 3
    with pytest.raises(O.OracleError):
        O.build_constraint_clusters(spec2)


def test_rigid_water_is_recognised_as_settle_clusters():
    spec = systems.rigid_water(systems.spce_water(40, seed=2))
    info, slots = I.plan_layout(spec, I.VVIntegrator(300.0, 10, 1.0, 40, 0.002))
    cl = O.build_constraint_clusters(spec)
    assert info.constraints_fused and info.num_settle_clusters == 40 == len(cl["settle_atoms"]) and info.num_shake_clusters == 0
    assert np.array_equal(cl["settle_atoms"][:, 0] % 3, np.zeros(40, int))            # the oxygen is the apex
    assert list(info.dof)[0] == 3 * 120 - 120 - 3
    wave_of = np.full(spec.num_atoms, -1)
    live = slots[:, 0] >= 0
    wave_of[slots[live, 0]] = np.nonzero(live)[0] // 64
    assert all(len(set(wave_of[row])) == 1 for row in cl["settle_atoms"])
    # water + a solute with HBonds in one System: both kinds side by side
    mix = systems.constrain_hydrogens(systems.drude_il(cells=(1, 1, 1), pairs_per_cell=3, seed=2))
    cm = O.build_constraint_clusters(mix)
    assert len(cm["settle_atoms"]) == 0 and len(cm["shake_atoms"]) > 0


@pytest.mark.parametrize("cfg,expect", [("C1", 1), ("C2", 1), ("C3", 1), ("C4", 1), ("C5", 0)])
def test_periodic_layout_is_found_for_runs_of_identical_molecules(cfg, expect, monkeypatch):
    """vv_host.hpp PeriodicLayout: the BASELINE bulk boxes are runs of identical molecules (C3: 12 cells of 250 cations + 250 anions), the
    electrode slab with its image particles is not.  Forced on here (auto only from ~1.1 M particles); analyze() itself verifies, lane
    for lane, that the arithmetic layout reproduces the explicit slot table before enabling it -- these are the invariants seen from outside."""
    monkeypatch.setenv("VVHIP_PERIODIC", "1")
    spec = systems.make_config(cfg)
    p = O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02 if len(spec.drude_pairs) else 0.0)
    info, slots = I.plan_layout(spec, _integrator(p))
    assert info.periodic_layout == expect
    atoms = slots[:, 0]
    if expect:
        # every wave holds consecutive particles from lane 0 on, waves in particle order
        used = atoms >= 0
        a2 = atoms.reshape(-1, 64)
        u2 = used.reshape(-1, 64)
        cnt = u2.sum(axis=1)
        assert all(u2[w, :cnt[w]].all() and not u2[w, cnt[w]:].any() for w in range(a2.shape[0]))
        assert np.array_equal(atoms[used], np.arange(spec.num_atoms))
    monkeypatch.setenv("VVHIP_PERIODIC", "0")
    info0, slots0 = I.plan_layout(spec, _integrator(p))
    assert info0.periodic_layout == 0 and info0.num_waves <= info.num_waves
    assert np.array_equal(np.sort(slots0[:, 0][slots0[:, 0] >= 0]), np.sort(atoms[atoms >= 0]))


def test_periodic_layout_auto_threshold():
    spec = systems.make_config("C3", scale=10)
    info, _ = I.plan_layout(spec, _integrator(O.Params(temperature=333.0, max_drude_distance=0.02)))
    assert info.periodic_layout == 1 and info.num_waves == 10 * 2004
    for scale in (1, 4, 8):                   # 111 000, 444 000 and 888 000 particles: best-fit packing
        info, _ = I.plan_layout(systems.make_config("C3", scale=scale), _integrator(O.Params(temperature=333.0, max_drude_distance=0.02)))
        assert info.periodic_layout == 0


def test_periodic_layout_units_must_repeat(monkeypatch):
    """Short runs: 13 molecules of one kind followed by 20 of another are two regions, not one long non-repeating unit (the first version of the
    greedy decomposition took whatever covered the most clusters)."""
    monkeypatch.setenv("VVHIP_PERIODIC", "1")
    import tests.test_gpu_periodic as G
    spec = G._four_species()
    it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    it.setMaxDrudeDistance(0.02)
    info, slots = I.plan_layout(spec, it)
    assert info.periodic_layout == 1 and info.num_waves == 3 * 4       # 3 cells x (9x7 | 13x4 | 20x3 | 30x1 atoms: one wave each)


def _bytes(plan):
    a, b = C.c_int32(0), C.c_int32(0)
    assert H.lib.vvhip_algorithmic_bytes(plan, C.byref(a), C.byref(b)) == H.OK
    return a.value, b.value


def test_algorithmic_bytes_follow_what_the_kernels_read():
    """vvhip_algorithmic_bytes (bench.py prices every launch with it) per configuration, no GPU needed: SURVEY section 8d's accounting for the
    headline path, + posq and the per-lane cos(kz) hand-over with the cos perturbation, + the positions of cluster members (their
    share of the particles) and the cluster word / parameters with in-kernel constraints."""
    def plan_of(cfg, cos=0.0, **kw):
        spec = systems.make_config(cfg, **kw)
        it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
        it.setMaxDrudeDistance(0.02)
        it.setCosAcceleration(cos)
        plan, info, keep = I.create_plan(spec, it, "mixed")
        return spec, plan, keep
    spec, plan, keep = plan_of("C3")
    assert _bytes(plan) == (62, 158)                                  # R velm 32 + R force 24 + idx 6 | + R pos 32 + W velm 32 + W pos 32
    H.lib.vvhip_plan_destroy(plan)
    spec, plan, keep = plan_of("C3", cos=0.02)
    assert _bytes(plan) == (62 + 16 + 8, 158 + 8)                     # + R posq, W cos(kz) | + R cos(kz)
    H.lib.vvhip_plan_destroy(plan)
    spec, plan, keep = plan_of("C3", hbonds=True)
    members = len(set(np.asarray(spec.constraints).reshape(-1).tolist()))
    share = (32 * members + spec.num_atoms // 2) // spec.num_atoms
    assert 10 <= share <= 20                                          # 51 000 of 111 000 particles sit in a cluster
    assert _bytes(plan) == (94 + share + 20, 134 + 20)                # A writes the velocities back, reads members' positions, both read cluster word + parameters
    H.lib.vvhip_plan_destroy(plan)


def test_tuning_hook_names():
    """vvhip_debug_tune replaces the tuning environment switches of rounds 1-3: known names are accepted before binding, unknown ones refused."""
    spec = systems.make_config("C2")
    it = I.VVIntegrator(300.0, 10.0, 1.0, 40.0, 0.002)
    plan, info, keep = I.create_plan(spec, it, "mixed")
    for key, val in (("grid_cap_a", 8), ("grid_cap_b", 8), ("split_chain_waves", 1), ("periodic_kernels", 0), ("periodic_a", 1), ("rekick", 0),
                     ("no_moments", 1), ("mass_tab_a", 1), ("mass_tab_b", 0), ("acc_store", 0), ("block_threads", 128)):
        assert H.lib.vvhip_debug_tune(plan, key.encode(), val) == H.OK, key
    assert H.lib.vvhip_debug_tune(plan, b"no_such_choice", 1) == H.ERR_INVALID
    assert b"no_such_choice" in H.lib.vvhip_last_error(plan)
    assert H.lib.vvhip_debug_tune(plan, b"block_threads", 100) == H.ERR_INVALID
    H.lib.vvhip_plan_destroy(plan)


@pytest.mark.parametrize("hangles", [False, True])
def test_general_constraint_clusters_are_coloured_lists_per_wave(hangles):
    """constraints=AllBonds / HAngles: rings and chains are neither hydrogen-type clusters nor rigid triangles.  vv::analyze then keeps every
    connected component of the constraint graph in one wave and writes the wave's constraint list into the per-lane constraint tables:
    constraint l in lane l, sorted by colour, no two constraints of one colour sharing a particle (csrc/vv_layout.h: GC_WORD_*)."""
    spec = systems.constrain_all_bonds(systems.bulk_Im21(cells=(1, 1, 1), pairs_per_cell=20), hangles=hangles)
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    plan, info, keep = I.create_plan(spec, it, "mixed")
    try:
        assert info.constraints_fused == 1 and info.num_shake_clusters == 0 and info.num_general_constraints == len(spec.constraints)
        nslots = info.num_waves * 64
        slots = np.zeros((nslots, 2), dtype=np.int32)
        assert H.lib.vvhip_plan_get_slots(plan, slots.ctypes.data, nslots) == nslots
        wave_of = {int(a): k // 64 for k, a in enumerate(slots[:, 0]) if a >= 0}
        lane_of = {int(a): k % 64 for k, a in enumerate(slots[:, 0]) if a >= 0}
        for a, b in spec.constraints:                      # both ends in one wave, both marked as constraint members
            assert wave_of[int(a)] == wave_of[int(b)]
            assert slots[wave_of[int(a)] * 64 + lane_of[int(a)], 1] & (1 << 30) and slots[wave_of[int(b)] * 64 + lane_of[int(b)], 1] & (1 << 30)
        # DOF: every constraint leaves the atom group (HOST:505-509)
        t = O.build_tables(spec, O.Params(temperature=333.0, max_drude_distance=0.02))
        assert list(info.dof) == list(t["dof"])
    finally:
        H.lib.vvhip_plan_destroy(plan)
    # the oracle colours in the same System order (greedy, smallest free colour): within a colour no particle appears twice
    atoms, params, ncol, colours = O.build_general_constraints(spec)
    assert len(atoms) == len(spec.constraints) and 2 <= ncol <= 16 and (np.diff(colours) >= 0).all()
    for c in range(ncol):
        ends = atoms[colours == c].reshape(-1)
        assert len(set(ends.tolist())) == ends.size


def test_virtual_sites_share_a_wave_with_their_parents():
    """vvhip_system_desc.virtual_sites: a site gets a lane (massless particles otherwise have none) in the wave of its parents; what cannot
    be placed in-kernel is reported (num_virtual_sites == 0: the caller keeps its own computeVirtualSites), what is malformed is refused."""
    base = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=20, seed=2)
    for interleaved in (True, False):
        spec = systems.add_virtual_sites(base, kinds=(3, 0), interleaved=interleaved)
        for use_com in (True, False):
            it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
            it.setUseCOMTempGroup(use_com)
            info, slots = I.plan_layout(spec, it)
            assert info.num_virtual_sites == len(spec.virtual_sites) == 80
            wave_of = np.full(spec.num_atoms, -1)
            live = slots[:, 0] >= 0
            wave_of[slots[live, 0]] = np.nonzero(live)[0] // 64
            for site, kind, parents, prm in spec.virtual_sites:           # placed from a parent's lane: the site itself has none
                assert wave_of[site] == -1 and wave_of[parents[0]] >= 0 and all(wave_of[q] == wave_of[parents[0]] for q in parents)
            info0, _ = I.plan_layout(base, it)
            assert list(info.dof) == list(info0.dof) and info.num_slots_used == info0.num_slots_used and info.num_waves == info0.num_waves
    # more sites than parents on a molecule: the ones no parent is free for get lanes of their own
    crowded = systems.add_virtual_sites(systems.spce_water(8), kinds=(1, 1, 0, 2))
    info, slots = I.plan_layout(crowded, I.VVIntegrator(300.0, 10, 1.0, 40, 0.001))
    assert info.num_virtual_sites == 32 and info.num_slots_used == 24 + 8
    spec = systems.add_virtual_sites(systems.spce_water(8), kinds=(1,))
    it = I.VVIntegrator(300.0, 10, 1.0, 40, 0.001)
    # a site hanging on another site: left to the caller
    s2 = systems.add_virtual_sites(systems.spce_water(8), kinds=(1,))
    s2.virtual_sites[1] = (s2.virtual_sites[1][0], 0, (s2.virtual_sites[0][0], 4), (0.5, 0.5))
    assert I.plan_layout(s2, it)[0].num_virtual_sites == 0
    # parents in two molecules far apart while molecules are kept together for the molecular temperature group: not in one wave, left to the caller
    s3 = systems.add_virtual_sites(systems.drude_il(cells=(1, 1, 1), pairs_per_cell=40, seed=2), kinds=(0,), interleaved=False)
    near, far = s3.virtual_sites[0], s3.virtual_sites[-1]
    s3.virtual_sites[0] = (near[0], 0, (near[2][0], far[2][1]), far[3])
    itc = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
    itc.setUseCOMTempGroup(True)
    assert I.plan_layout(s3, itc)[0].num_virtual_sites == 0
    # malformed: a massive site, an unknown kind, a site described twice
    for bad, text in (((0, 1, (1, 2, 4), (0.3, 0.3, 0.4)), "mass 0"), ((3, 7, (0, 1, 2), (1.0,)), "kind"), (spec.virtual_sites[0], "twice")):
        s4 = systems.add_virtual_sites(systems.spce_water(8), kinds=(1,))
        s4.virtual_sites.append(bad)
        with pytest.raises(H.VVHipError, match=text):
            I.plan_layout(s4, it)


def test_a_shard_must_not_cut_a_virtual_site_from_its_parents():
    spec = systems.add_virtual_sites(systems.spce_water(8), kinds=(1,), interleaved=False)      # sites behind the last molecule
    it = I.VVIntegrator(300.0, 10, 1.0, 40, 0.001)
    with pytest.raises(H.VVHipError, match="virtual site"):
        I.plan_layout(spec, it, shard=(0, 12))              # molecules 0..3 without their sites (which sit at 24..27)
    inter = systems.add_virtual_sites(systems.spce_water(8), kinds=(1,))                         # O H H M: sites travel with their molecule
    info, _ = I.plan_layout(inter, it, shard=(0, 16))
    assert info.num_virtual_sites == 4


def test_a_plan_that_cannot_fuse_its_constraints_says_why():
    """Round-4 advisor: a bare constraints_fused = 0 told a host nothing.  The procedural look-alike box with `constrain_all_bonds` (random
    geometry: every pair within the cut-off becomes a "bond", one component holds far more than the 64 constraints of a wave's list) names its
    reason; the reference's own model with HAngles fuses and has none.  The wave packer counts constraints as well as lanes since round 5."""
    import ctypes as C
    for spec, fused in ((systems.constrain_all_bonds(systems.drude_il(cells=(1, 1, 1), pairs_per_cell=30, seed=1)), 0),
                        (systems.constrain_all_bonds(systems.bulk_Im21(cells=(1, 1, 1), pairs_per_cell=20), hangles=True), 1)):
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
        it.setMaxDrudeDistance(0.02)
        plan, info, _keep = I.create_plan(spec, it)
        try:
            reason = H.lib.vvhip_plan_unfused_reason(plan).decode()
            assert info.constraints_fused == fused
            if fused:
                assert reason == "" and info.num_general_constraints == len(spec.constraints)
                nslots = H.lib.vvhip_plan_get_slots(plan, None, 0)
                assert nslots == info.num_waves * 64
            else:
                assert "64 constraints of a wave's list" in reason or "16 colours" in reason or "two waves" in reason, reason
        finally:
            H.lib.vvhip_plan_destroy(plan)


def test_launch_shapes_of_the_measured_regimes():
    """The launch shapes the measurements of TUNING_LOG sections 12-13 settled on, as the plan chooses them for a whole MI355X (256 CUs) before any
    device is bound: one block of seven tile waves per CU at the headline size (one launch per step), one tile wave per block for a rank's
    share of the cos-perturbed box (round 5's three-wave rule went with the shared poll in round 6), two blocks per CU past one pass, whole rounds of
    four-wave blocks in the bandwidth-bound regime."""
    def shape(cfg, scale=1.0, cos=0.0, shard=None, hbonds=False):
        spec = systems.make_config(cfg, scale, hbonds=hbonds)
        it = I.VVIntegrator(333.0, 10, 1.0, 40, 0.001)
        it.setMaxDrudeDistance(0.02 if cfg in ("C3", "C5") else 0.0)
        it.setCosAcceleration(cos)
        sh = None if shard is None else pkg.distributed.shard_bounds(spec, shard)[0]
        info, _ = I.plan_layout(spec, it, shard=sh)
        return info.num_waves, I.plan_launch_shape(spec, it, shard=sh)

    nw, s = shape("C3")
    assert nw == 1752 and s == (448, 256, 256, 7)                  # 251 blocks of 7 tile waves + the thermostat wave: the one-launch step
    nw, s = shape("C3", cos=0.02)
    assert s == (448, 256, 256, 7)                                 # C4: same shape (the thermostat wave polls the ten rows)
    nw, s = shape("C3", cos=0.02, shard=8)
    assert nw <= 256 and s == (64, 256, 256, 1)                    # one rank's eighth of C4: one tile wave per block like every small system (round 6: the
                                                                   # thermostat wave polls the ten rows alone; profiles/r06w_c4_shard_shapes.txt)
    nw, s = shape("C3", shard=8)
    assert s == (64, 256, 256, 1)                                  # ... of C3 (three rows): one tile wave per block stays the best
    nw, s = shape("C2")
    assert nw == 157 and s == (64, 256, 256, 1)
    nw, s = shape("C5")
    assert nw == 338 and s[0] == 128 and s[3] == 2                 # 169 blocks of two tile waves
    nw, s = shape("C3", 1.5)
    assert nw > 1792 and s[1] == 512 and s[3] == 0                 # past one pass: two blocks per CU, two launches
    nw, s = shape("C3", 30.0)
    assert s == (256, 2048, 512, 0)                                # 3.3 M particles: stand-alone chain, kernel A eight blocks per CU
    nw, s = shape("C3", 80.0)
    assert s == (256, 1024, 512, 0)                                # 8.9 M particles: kernel A in whole rounds (four per CU)
