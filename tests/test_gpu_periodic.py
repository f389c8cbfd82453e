"""-m gpu: the arithmetic ("periodic") work-item layout -- particle indices computed from the wave index, role words from the pattern wave
(vv_host.hpp: PeriodicLayout) -- forced on small systems (it is automatic from ~1.1 M particles) and compared with the oracle like
every other path, in kernel B alone (the default where the layout is on) and in both kernels; plus bit-equality of the trajectory with
the explicit-slot kernels on the same layout, which is what the change must preserve."""
import importlib

import numpy as np
import pytest

from oracle import oracle as O

pkg = importlib.import_module("openmm-velocityverlet_amd")
H, I, systems = pkg.vvhip, pkg.integrator, pkg.systems
pytestmark = pytest.mark.gpu


def _run(spec, prec, nsteps, env, monkeypatch, cos=0.0, maxd=0.02, T=333.0, dt=0.001, graph=False, middle=True):
    # VVHIP_PERIODIC is the library's switch for the layout; "VVHIP_PERIODIC_K" / "VVHIP_PERIODIC_A" in `env` are this file's names for the
    # per-plan hook vvhip_debug_tune("periodic_kernels" / "periodic_a"): slot words loaded instead of computed, on the same layout
    monkeypatch.delenv("VVHIP_PERIODIC", raising=False)
    tune = {}
    for k, v in env.items():
        if k == "VVHIP_PERIODIC_K": tune["periodic_kernels"] = int(v)
        elif k == "VVHIP_PERIODIC_A": tune["periodic_a"] = int(v)
        else: monkeypatch.setenv(k, v)
    it = I.VVIntegrator(T, 10.0, 1.0, 40.0, dt)
    it.setMaxDrudeDistance(maxd)
    it.setCosAcceleration(cos)
    it.setUseMiddleScheme(middle)
    ctx = I.Context(spec, it, precision=prec, force_provider="tether", tune=tune)
    try:
        if graph:
            ctx.run_graph(nsteps, nsteps)
        else:
            it.step(nsteps)
        return ctx.info.periodic_layout, ctx.getVelm(), ctx.getPosq(), ctx.getPosqCorrection(), np.array(list(ctx.getNHState().ke2))
    finally:
        ctx.close()


SYSTEMS = {
    "bulk": lambda: (systems.drude_il(cells=(1, 1, 1), pairs_per_cell=70, seed=7), dict(maxd=0.02)),               # 70 cations, 70 anions: partial last waves
    "bulk_cells": lambda: (systems.make_config("C3", scale=0.25), dict(maxd=0.02)),                                # cells of the reference topology
    "water": lambda: (systems.spce_water(500, seed=5), dict(maxd=0.0, T=300.0, dt=0.002)),
    "nondrude": lambda: (systems.nondrude_il(num_pairs=40, seed=3), dict(maxd=0.0)),
}


@pytest.mark.parametrize("prec", ["mixed", "double", "single"])
@pytest.mark.parametrize("cos", [0.0, 0.02])
@pytest.mark.parametrize("name", sorted(SYSTEMS))
def test_periodic_kernels_equal_explicit_kernels_and_oracle(name, cos, prec, monkeypatch):
    spec, kw = SYSTEMS[name]()
    nsteps = 2 if prec == "single" else 12
    flag, v_b, p_b, c_b, ke_b = _run(spec, prec, nsteps, {"VVHIP_PERIODIC": "1"}, monkeypatch, cos=cos, **kw)                                   # kernel B periodic
    assert flag == 1, "layout not recognised as periodic"
    _, v_ab, p_ab, c_ab, ke_ab = _run(spec, prec, nsteps, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_A": "1"}, monkeypatch, cos=cos, **kw)       # both kernels
    _, v_e, p_e, c_e, ke_e = _run(spec, prec, nsteps, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_K": "0"}, monkeypatch, cos=cos, **kw)           # same layout, slot words loaded
    flag0, v_0, p_0, _, _ = _run(spec, prec, nsteps, {"VVHIP_PERIODIC": "0"}, monkeypatch, cos=cos, **kw)                                      # best-fit layout
    assert flag0 == 0
    # same layout => same summation order => the very same bits, whichever way the slot words come
    for a, b in ((v_b, v_e), (p_b, p_e), (c_b, c_e), (v_ab, v_e), (p_ab, p_e), (c_ab, c_e)):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert np.array_equal(ke_b, ke_e) and np.array_equal(ke_ab, ke_e)
    # against the oracle (and the other layout: different summation order, same physics)
    p = O.Params(temperature=kw.get("T", 333.0), drude_temperature=1.0, step_size=kw.get("dt", 0.001), max_drude_distance=kw["maxd"], cos_acceleration=cos)
    osys = O.OracleSystem(spec, p, prec, force_mode=1)
    osys.step(nsteps)
    tol = 1e-5
    massive = osys.velm[:, 3] != 0
    for v, x in ((v_b, p_b), (v_0, p_0)):
        ev = np.abs(v[massive, :3].astype(np.float64) - osys.velm[massive, :3]).max() / np.abs(osys.velm[massive, :3]).max()
        ex = np.abs(x[:, :3].astype(np.float64) - osys.posq[:, :3]).max() / np.abs(osys.posq[:, :3]).max()
        assert ev < tol and ex < tol, f"{name}/{prec}/cos={cos}: rel err vel {ev:.2e} pos {ex:.2e}"


def test_periodic_kernels_under_graph_replay(monkeypatch):
    spec, kw = SYSTEMS["bulk_cells"]()
    _, v_g, p_g, c_g, _ = _run(spec, "mixed", 40, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_A": "1"}, monkeypatch, graph=True, **kw)
    _, v_e, p_e, c_e, _ = _run(spec, "mixed", 40, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_K": "0"}, monkeypatch, **kw)
    assert np.array_equal(v_g.view(np.uint8), v_e.view(np.uint8)) and np.array_equal(p_g.view(np.uint8), p_e.view(np.uint8))


@pytest.mark.parametrize("name", ["water", "nondrude"])
def test_large_system_launch_shape_without_drude_pairs(name, monkeypatch):
    """The stage sets big water / plain ionic-liquid boxes run (kernel B without thermostat wave and without hard wall), forced at a testable size."""
    monkeypatch.setitem(I.DEFAULT_TUNE, "split_chain_waves", 1)
    monkeypatch.setitem(I.DEFAULT_TUNE, "grid_cap_a", 8)
    monkeypatch.setitem(I.DEFAULT_TUNE, "grid_cap_b", 8)
    spec, kw = SYSTEMS[name]()
    for env in ({"VVHIP_PERIODIC": "1"}, {"VVHIP_PERIODIC": "0"}):
        _, v, x, c, ke = _run(spec, "mixed", 12, env, monkeypatch, **kw)
        p = O.Params(temperature=kw.get("T", 333.0), drude_temperature=1.0, step_size=kw.get("dt", 0.001), max_drude_distance=kw["maxd"])
        osys = O.OracleSystem(spec, p, "mixed", force_mode=1)
        osys.step(12)
        ev = np.abs(v[:, :3] - osys.velm[:, :3]).max() / np.abs(osys.velm[:, :3]).max()
        ex = np.abs(x[:, :3].astype(np.float64) - osys.posq[:, :3]).max() / np.abs(osys.posq[:, :3]).max()
        assert ev < 1e-9 and ex < 2e-7, f"{name}/{env}: rel err vel {ev:.2e} pos {ex:.2e}"


@pytest.mark.parametrize("cos", [0.0, 0.02])
def test_periodic_kernels_in_the_large_system_launch_shape(cos, monkeypatch):
    """What systems beyond ~0.7 M particles run -- the chain as its own launch, kernel B without a thermostat wave, 256-thread blocks
    striding over tiles -- forced at a testable size, with the periodic layout, against the oracle and against the explicit-slot kernels."""
    monkeypatch.setitem(I.DEFAULT_TUNE, "split_chain_waves", 1)
    monkeypatch.setitem(I.DEFAULT_TUNE, "grid_cap_a", 8)           # few blocks: every wave strides over several tiles
    monkeypatch.setitem(I.DEFAULT_TUNE, "grid_cap_b", 8)
    spec, kw = SYSTEMS["bulk_cells"]()
    _, v_p, p_p, c_p, ke_p = _run(spec, "mixed", 12, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_A": "1"}, monkeypatch, cos=cos, **kw)
    _, v_e, p_e, c_e, ke_e = _run(spec, "mixed", 12, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_K": "0"}, monkeypatch, cos=cos, **kw)
    assert np.array_equal(v_p.view(np.uint8), v_e.view(np.uint8)) and np.array_equal(p_p.view(np.uint8), p_e.view(np.uint8))
    assert np.array_equal(c_p.view(np.uint8), c_e.view(np.uint8)) and np.array_equal(ke_p, ke_e)
    p = O.Params(temperature=333.0, drude_temperature=1.0, step_size=0.001, max_drude_distance=kw["maxd"], cos_acceleration=cos)
    osys = O.OracleSystem(spec, p, "mixed", force_mode=1)
    osys.step(12)
    ev = np.abs(v_p[:, :3] - osys.velm[:, :3]).max() / np.abs(osys.velm[:, :3]).max()
    ex = np.abs(p_p[:, :3].astype(np.float64) - osys.posq[:, :3]).max() / np.abs(osys.posq[:, :3]).max()
    assert ev < 1e-9 and ex < 2e-7, f"rel err vel {ev:.2e} pos {ex:.2e}"


def test_periodic_layout_switches_itself_on_and_matches_the_oracle_at_that_size(monkeypatch):
    """C3 tiled ten times (1 110 000 particles): the first tiling at which the arithmetic layout is chosen without being asked for (from 1.1 M lanes;
    C3 tiled twice, 222 000 particles, must not get it)."""
    monkeypatch.delenv("VVHIP_PERIODIC", raising=False)
    it2 = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it2.setMaxDrudeDistance(0.02)
    assert I.plan_layout(systems.make_config("C3", scale=2), it2)[0].periodic_layout == 0
    spec = systems.make_config("C3", scale=10)
    flag, v, x, c, ke = _run(spec, "mixed", 6, {}, monkeypatch, maxd=0.02)
    assert flag == 1
    osys = O.OracleSystem(spec, O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02), "mixed", force_mode=1, num_threads=8)
    osys.step(6)
    ev = np.abs(v[:, :3] - osys.velm[:, :3]).max() / np.abs(osys.velm[:, :3]).max()
    ex = np.abs(x[:, :3].astype(np.float64) - osys.posq[:, :3]).max() / np.abs(osys.posq[:, :3]).max()
    assert ev < 1e-9 and ex < 2e-7, f"rel err vel {ev:.2e} pos {ex:.2e}"
    assert np.allclose(ke[:3], osys.ke2()[:3], rtol=1e-9)


def _four_species(cells=3, seed=4):
    """Four kinds of molecules (Drude pairs and plain atoms), each kind in a run, the four runs repeated cell after cell: four regions."""
    templates = [(["pair", "pair", "pair", "hydrogen"], 12.0), (["pair", "heavy", "heavy"], 14.0), (["heavy", "hydrogen", "hydrogen"], 15.9994), (["heavy"], 35.45)]
    counts = [9, 13, 20, 30]
    masses, mol_id, pairs = [], [], []
    mol = 0
    for _ in range(cells):
        for (units, heavy), cnt in zip(templates, counts):
            for _ in range(cnt):
                for u in units:
                    if u == "pair":
                        masses.extend([heavy - 0.4, 0.4]); pairs.append((len(masses) - 1, len(masses) - 2)); mol_id.extend([mol, mol])
                    else:
                        masses.append(heavy if u == "heavy" else 1.008); mol_id.append(mol)
                mol += 1
    n = len(masses)
    rng = np.random.default_rng(seed)
    masses = np.array(masses)
    pos = rng.uniform(0, 3, (n, 3))
    vel = rng.standard_normal((n, 3)) * np.sqrt(systems.BOLTZ * 333.0 / masses)[:, None]
    pairs = np.array(pairs, np.int32)
    pos[pairs[:, 0]] = pos[pairs[:, 1]] + rng.normal(0, 2e-4, (len(pairs), 3))           # Drudes next to their parents
    vel[pairs[:, 0]] = vel[pairs[:, 1]] + rng.standard_normal((len(pairs), 3)) * np.sqrt(systems.BOLTZ * 1.0 / 0.4)
    return systems.SystemSpec(name="four_species", masses=masses, charges=np.zeros(n), positions=pos, velocities=vel, box=np.array([3.0, 3.0, 3.0]),
                              mol_id=np.array(mol_id, np.int32), drude_pairs=pairs, constraints=np.zeros((0, 2), np.int32), has_cm_motion_remover=True)


@pytest.mark.parametrize("cos", [0.0, 0.02])
def test_periodic_layout_with_four_regions_per_cell(cos, monkeypatch):
    spec = _four_species()
    flag, v_p, p_p, c_p, ke_p = _run(spec, "mixed", 10, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_A": "1"}, monkeypatch, cos=cos, maxd=0.02)
    assert flag == 1
    _, v_e, p_e, c_e, ke_e = _run(spec, "mixed", 10, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_K": "0"}, monkeypatch, cos=cos, maxd=0.02)
    assert np.array_equal(v_p.view(np.uint8), v_e.view(np.uint8)) and np.array_equal(p_p.view(np.uint8), p_e.view(np.uint8)) and np.array_equal(ke_p, ke_e)
    osys = O.OracleSystem(spec, O.Params(temperature=333.0, drude_temperature=1.0, max_drude_distance=0.02, cos_acceleration=cos), "mixed", force_mode=1)
    osys.step(10)
    ev = np.abs(v_p[:, :3] - osys.velm[:, :3]).max() / np.abs(osys.velm[:, :3]).max()
    ex = np.abs(p_p[:, :3].astype(np.float64) - osys.posq[:, :3]).max() / np.abs(osys.posq[:, :3]).max()
    assert ev < 1e-9 and ex < 2e-7, f"rel err vel {ev:.2e} pos {ex:.2e}"


def _random_repeated(seed):
    rng = np.random.default_rng(seed)
    templates = []
    for _ in range(int(rng.integers(1, 4))):
        size = int(rng.choice([1, 2, 3, 4, 7, 10, 19, 27, 33, 64]))
        units, k = [], 0
        while k < size:
            if rng.random() < 0.5 and k + 2 <= size:
                units.append("pair"); k += 2
            else:
                units.append("heavy" if rng.random() < 0.6 else "hydrogen"); k += 1
        templates.append((units, 12.0 + int(rng.integers(0, 3))))
    counts = [int(rng.integers(1, 15)) for _ in templates]
    cells = int(rng.integers(1, 5))
    masses, mol_id, pairs, mol = [], [], [], 0
    for _ in range(cells):
        for (units, heavy), cnt in zip(templates, counts):
            for _ in range(cnt):
                for u in units:
                    if u == "pair":
                        masses.extend([heavy - 0.4, 0.4]); pairs.append((len(masses) - 1, len(masses) - 2)); mol_id.extend([mol, mol])
                    else:
                        masses.append(heavy if u == "heavy" else 1.008); mol_id.append(mol)
                mol += 1
    n = len(masses)
    masses = np.array(masses)
    pos = rng.uniform(0, 3, (n, 3))
    vel = rng.standard_normal((n, 3)) * np.sqrt(systems.BOLTZ * 333.0 / masses)[:, None]
    pairs = np.array(pairs, np.int32).reshape(-1, 2)
    if len(pairs):
        pos[pairs[:, 0]] = pos[pairs[:, 1]] + rng.normal(0, 2e-4, (len(pairs), 3))
        vel[pairs[:, 0]] = vel[pairs[:, 1]] + rng.standard_normal((len(pairs), 3)) * np.sqrt(systems.BOLTZ * 1.0 / 0.4)
    return systems.SystemSpec(name=f"rand{seed}", masses=masses, charges=np.zeros(n), positions=pos, velocities=vel, box=np.array([3.0, 3.0, 3.0]),
                              mol_id=np.array(mol_id, np.int32), drude_pairs=pairs, constraints=np.zeros((0, 2), np.int32), has_cm_motion_remover=True)


def test_periodic_kernels_on_random_repeated_inventories(monkeypatch):
    """Forty random inventories of repeated molecules (1-3 kinds, 1-14 of each, 1-4 cells; with and without Drude pairs, with and without the COM
    temperature group): wherever the layout is recognised, computing the slot words must not change a bit of the trajectory."""
    recognised = 0
    for seed in range(40):
        spec = _random_repeated(seed)
        maxd = 0.02 if len(spec.drude_pairs) else 0.0
        flag, v_p, p_p, c_p, ke_p = _run(spec, "mixed", 5, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_A": "1"}, monkeypatch, maxd=maxd)
        if not flag:
            continue
        recognised += 1
        _, v_e, p_e, c_e, ke_e = _run(spec, "mixed", 5, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_K": "0"}, monkeypatch, maxd=maxd)
        assert np.array_equal(v_p.view(np.uint8), v_e.view(np.uint8)) and np.array_equal(p_p.view(np.uint8), p_e.view(np.uint8)), seed
        assert np.array_equal(c_p.view(np.uint8), c_e.view(np.uint8)) and np.array_equal(ke_p, ke_e), seed
        assert np.isfinite(v_p).all() and np.isfinite(p_p).all(), seed
    assert recognised >= 25, recognised


CONSTRAINED = {
    "bulk_hbonds": lambda: (systems.make_config("C3", scale=0.25, hbonds=True), dict(maxd=0.02)),            # SHAKE clusters (C-H, CH2, CH3) next to Drude pairs
    "rigid_water": lambda: (systems.rigid_water(systems.spce_water(500, seed=5)), dict(maxd=0.0, T=300.0, dt=0.002)),      # SETTLE
}


@pytest.mark.parametrize("large_shape", [False, True])
@pytest.mark.parametrize("name", sorted(CONSTRAINED))
def test_periodic_kernels_with_in_kernel_constraints(name, large_shape, monkeypatch):
    """Constraint cluster words and parameters come from the pattern wave as well (both kernels take the arithmetic path here): bit-equal with
    the loaded tables, constraint lengths kept, oracle within the constraint tolerance."""
    if large_shape:
        monkeypatch.setitem(I.DEFAULT_TUNE, "split_chain_waves", 1)
        monkeypatch.setitem(I.DEFAULT_TUNE, "grid_cap_a", 8)
        monkeypatch.setitem(I.DEFAULT_TUNE, "grid_cap_b", 8)
    spec, kw = CONSTRAINED[name]()
    flag, v_p, p_p, c_p, ke_p = _run(spec, "mixed", 10, {"VVHIP_PERIODIC": "1"}, monkeypatch, **kw)
    assert flag == 1
    _, v_e, p_e, c_e, ke_e = _run(spec, "mixed", 10, {"VVHIP_PERIODIC": "1", "VVHIP_PERIODIC_K": "0"}, monkeypatch, **kw)
    assert np.array_equal(v_p.view(np.uint8), v_e.view(np.uint8)) and np.array_equal(p_p.view(np.uint8), p_e.view(np.uint8))
    assert np.array_equal(c_p.view(np.uint8), c_e.view(np.uint8)) and np.array_equal(ke_p, ke_e)
    x = p_p[:, :3].astype(np.float64) + c_p[:, :3].astype(np.float64)
    c, dist = np.asarray(spec.constraints), np.asarray(spec.constraint_distances)
    r = np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1)
    assert np.abs(r - dist).max() / dist.max() < 2e-5
    p = O.Params(temperature=kw.get("T", 333.0), drude_temperature=1.0, step_size=kw.get("dt", 0.001), max_drude_distance=kw["maxd"])
    osys = O.OracleSystem(spec, p, "mixed", force_mode=1)
    osys.step(10)
    ev = np.abs(v_p[:, :3] - osys.velm[:, :3]).max() / np.abs(osys.velm[:, :3]).max()
    ex = np.abs(x - osys.positions()).max() / np.abs(osys.positions()).max()
    assert ev < 1e-5 and ex < 1e-5, f"{name}: rel err vel {ev:.2e} pos {ex:.2e}"


@pytest.mark.parametrize("periodic", ["1", "0"])
def test_classic_scheme_in_the_large_system_launch_shape(periodic, monkeypatch):
    """stepVV's two thermostat applications as big boxes run them (scale + half kick + positions + hard wall, and scale alone, kernel B without a
    thermostat wave), forced at a testable size, with computed and with loaded slot words."""
    monkeypatch.setitem(I.DEFAULT_TUNE, "split_chain_waves", 1)
    monkeypatch.setitem(I.DEFAULT_TUNE, "grid_cap_a", 8)
    monkeypatch.setitem(I.DEFAULT_TUNE, "grid_cap_b", 8)
    spec, kw = SYSTEMS["bulk_cells"]()
    _, v, x, c, ke = _run(spec, "mixed", 12, {"VVHIP_PERIODIC": periodic}, monkeypatch, middle=False, **kw)
    p = O.Params(temperature=333.0, drude_temperature=1.0, step_size=0.001, max_drude_distance=kw["maxd"], use_middle_scheme=False)
    osys = O.OracleSystem(spec, p, "mixed", force_mode=1)
    osys.step(12)
    ev = np.abs(v[:, :3] - osys.velm[:, :3]).max() / np.abs(osys.velm[:, :3]).max()
    ex = np.abs(x[:, :3].astype(np.float64) - osys.posq[:, :3]).max() / np.abs(osys.posq[:, :3]).max()
    assert ev < 1e-9 and ex < 2e-7, f"rel err vel {ev:.2e} pos {ex:.2e}"
