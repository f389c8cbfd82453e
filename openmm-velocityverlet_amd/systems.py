"""Synthetic particle systems for the BASELINE.json configurations (C1..C5).

The reference's example drivers build an OpenMM ``System`` from gro/psf/prm files
(/root/reference/examples/run-bulk.py:56-75, run-edl.py:82-100).  Neither OpenMM nor
those files exist on the benchmark machine, so this module generates systems with the
same *integrator-relevant* structure procedurally: particle masses, charges, molecule
membership, Drude (Drude, parent) pairs, constraint pairs (for DOF accounting only),
Langevin / image / electrolyte particle sets.  Chemistry facts used: element masses,
the CL&Pol convention that every heavy atom carries a Drude particle of mass 0.4 taken
from its parent and placed at index parent+1 (examples/ommhelper/oplspsffile.py:1515),
c2c1im+ = 8 heavy atoms + 11 H (27 particles), dca- = 5 heavy atoms (10 particles).

Everything is deterministic: ``numpy.random.default_rng(20241008)``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

BOLTZ = (1.380649e-23 * 6.02214076e23) / 1000.0  # kJ/mol/K, as OpenMM's SimTKOpenMMRealType.h
SEED = 20241008

_M_N, _M_C, _M_H, _M_D = 14.007, 12.011, 1.008, 0.4
# particle pattern of one c2c1im+ cation (EX/models/bulk_Im21: 27 particles, 8 Drude pairs, 11 hydrogens): heavy atom followed by
# its Drude and then by its hydrogens -- three ring C-H, N-CH3, N-CH2-CH3 -- so HBonds constraints give the 11 constraints per
# cation (33 000 at C3) that SURVEY.md section 8 quotes
_CATION = "ND CDH ND CDH CDH CDHHH CDHH CDHHH".replace(" ", "")
_ANION = "NDCDNDCDND"


def _template(pattern: str, net_charge: float):
    masses, is_drude, charges = [], [], []
    heavy = {"N": _M_N, "C": _M_C}
    for ch in pattern:
        if ch == "D":
            masses[-1] -= _M_D
            masses.append(_M_D)
            is_drude.append(True)
            charges[-1] += 2.0
            charges.append(-2.0)
        elif ch == "H":
            masses.append(_M_H)
            is_drude.append(False)
            charges.append(0.1)
        else:
            masses.append(heavy[ch])
            is_drude.append(False)
            charges.append(-0.05)
    charges = np.array(charges)
    charges[0] += net_charge - charges.sum()
    return np.array(masses), np.array(is_drude), charges


@dataclass
class SystemSpec:
    """What VVIntegrator::initialize and the kernels' initialize() read from an OpenMM System."""
    name: str
    masses: np.ndarray            # float64 [N]
    charges: np.ndarray           # float64 [N]
    positions: np.ndarray         # float64 [N,3] nm
    velocities: np.ndarray        # float64 [N,3] nm/ps
    box: np.ndarray               # float64 [3] nm (orthorhombic)
    mol_id: np.ndarray            # int32 [N]  (ContextImpl::getMolecules())
    drude_pairs: np.ndarray       # int32 [Np,2] (drude, parent)  (DrudeForce::getParticleParameters p, p1)
    constraints: np.ndarray       # int32 [Nc,2]  System constraints (pairs)
    constraint_distances: np.ndarray = None   # float64 [Nc] nm; given => hydrogen-type clusters are solved in-kernel (SHAKE)
    has_cm_motion_remover: bool = True
    particles_ld: List[int] = field(default_factory=list)
    image_pairs: List[Tuple[int, int]] = field(default_factory=list)   # (image, parent)
    particles_electrolyte: List[int] = field(default_factory=list)
    # System::getVirtualSite, one (site, kind, (parents...), (parameters...)) per massless site; kinds and parameter order as
    # include/vvhip.h: VVHIP_VSITE_*
    virtual_sites: List[tuple] = field(default_factory=list)

    @property
    def num_atoms(self) -> int:
        return int(self.masses.shape[0])

    @property
    def num_molecules(self) -> int:
        return int(self.mol_id.max()) + 1 if self.num_atoms else 0


def _maxwell_boltzmann(rng, masses, is_drude_of, parent_of, T, T_drude):
    """Normal particles ~ MB(T); Drude pairs: centre of mass ~ MB(T, m1+m2), relative ~ MB(T_D, mu)."""
    n = masses.shape[0]
    v = np.zeros((n, 3))
    massive = masses > 0
    sig = np.zeros(n)
    sig[massive] = np.sqrt(BOLTZ * T / masses[massive])
    v[:] = rng.standard_normal((n, 3)) * sig[:, None]
    d = np.nonzero(is_drude_of)[0]
    if d.size:
        p = parent_of[d]
        m1, m2 = masses[d], masses[p]
        mt = m1 + m2
        mu = m1 * m2 / mt
        vcm = rng.standard_normal((d.size, 3)) * np.sqrt(BOLTZ * T / mt)[:, None]
        vrel = rng.standard_normal((d.size, 3)) * np.sqrt(BOLTZ * T_drude / mu)[:, None]   # v_d - v_p
        v[d] = vcm + vrel * (m2 / mt)[:, None]
        v[p] = vcm - vrel * (m1 / mt)[:, None]
    return v


def _assemble(name, mol_templates, counts_per_cell, cells, cell_box, T, T_drude, rng):
    """Tile `cells` unit cells; each unit cell lists all molecules of kind 0, then kind 1, ... (as conf.gro does)."""
    masses, charges, isd, molid = [], [], [], []
    nmol = 0
    mols_per_cell = sum(counts_per_cell)
    cx, cy, cz = cells
    centers = []
    for ic in range(cx * cy * cz):
        ox = np.array([ic // (cy * cz), (ic // cz) % cy, ic % cz], dtype=float) * cell_box
        side = int(np.ceil(mols_per_cell ** (1.0 / 3.0)))
        grid = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:mols_per_cell]
        grid = rng.permutation(grid)
        k = 0
        for (m, q, d), cnt in zip(mol_templates, counts_per_cell):
            for _ in range(cnt):
                masses.append(m)
                charges.append(q)
                isd.append(d)
                molid.append(np.full(m.shape[0], nmol, dtype=np.int32))
                centers.append(ox + (grid[k] + 0.5) / side * cell_box)
                nmol += 1
                k += 1
    sizes = [m.shape[0] for m in masses]
    masses = np.concatenate(masses)
    charges = np.concatenate(charges)
    isd = np.concatenate(isd)
    molid = np.concatenate(molid)
    n = masses.shape[0]
    pos = np.repeat(np.array(centers), sizes, axis=0) + rng.uniform(-0.15, 0.15, size=(n, 3))
    parent_of = np.arange(n) - 1
    d = np.nonzero(isd)[0]
    pos[d] = pos[d - 1] + rng.normal(0.0, 2e-4, size=(d.size, 3))    # ~ sqrt(kT_D / k_D) at T_D = 1 K, k_D = 209 200
    vel = _maxwell_boltzmann(rng, masses, isd, parent_of, T, T_drude)
    box = cell_box * np.array(cells, dtype=float)
    pairs = np.stack([d, d - 1], axis=1).astype(np.int32) if d.size else np.zeros((0, 2), np.int32)
    return SystemSpec(name=name, masses=masses, charges=charges, positions=pos, velocities=vel, box=box,
                      mol_id=molid, drude_pairs=pairs, constraints=np.zeros((0, 2), np.int32))


def drude_il(cells=(2, 2, 3), pairs_per_cell=250, T=333.0, T_drude=1.0, seed=SEED, name=None) -> SystemSpec:
    """C3/C4: Im21-like Drude ionic liquid; default 2x2x3 cells x 250 ion pairs = 111 000 particles,
    6 000 molecules, 39 000 Drude pairs, 33 000 plain (H) particles (SURVEY.md §8 header)."""
    rng = np.random.default_rng(seed)
    cat = _template(_CATION, +0.8)
    ani = _template(_ANION, -0.8)
    spec = _assemble(name or f"drude_il_{cells[0]}x{cells[1]}x{cells[2]}x{pairs_per_cell}",
                     [(cat[0], cat[2], cat[1]), (ani[0], ani[2], ani[1])], [pairs_per_cell, pairs_per_cell],
                     cells, np.array([3.1, 3.1, 6.1]) * (pairs_per_cell / 250.0) ** (1.0 / 3.0), T, T_drude, rng)
    return spec


def nondrude_il(num_pairs=83, T=333.0, seed=SEED) -> SystemSpec:
    """C1: non-Drude cut -- Drude masses/charges merged back into their parents; 24 particles per ion pair."""
    rng = np.random.default_rng(seed)

    def merged(pattern, q):
        m, d, c = _template(pattern, q)
        keep = ~d
        mm = m.copy()
        cc = c.copy()
        idx = np.nonzero(d)[0]
        mm[idx - 1] += m[idx]
        cc[idx - 1] += c[idx]
        return mm[keep], cc[keep], np.zeros(int(keep.sum()), bool)
    cat, ani = merged(_CATION, 0.8), merged(_ANION, -0.8)
    spec = _assemble(f"nondrude_il_{num_pairs}", [cat, ani], [num_pairs, num_pairs], (1, 1, 1),
                     np.array([3.1, 3.1, 6.1]) * (num_pairs / 250.0) ** (1.0 / 3.0), T, 1.0, rng)
    return spec


def spce_water(num_molecules=3333, T=300.0, seed=SEED) -> SystemSpec:
    """C2: SPC/E-like water, 3 particles per molecule (O, H, H); rigid-water constraints are OpenMM's,
    so the parity/bench run is unconstrained and the DOF are counted accordingly."""
    rng = np.random.default_rng(seed)
    m = np.array([15.9994, 1.008, 1.008])
    q = np.array([-0.8476, 0.4238, 0.4238])
    d = np.zeros(3, bool)
    box = (num_molecules / 33.4) ** (1.0 / 3.0)
    return _assemble(f"spce_{num_molecules}", [(m, q, d)], [num_molecules], (1, 1, 1), np.array([box] * 3), T, 1.0, rng)


def edl_slab(num_ion_pairs=511, num_electrode=2496, T=333.0, T_drude=1.0, seed=SEED) -> SystemSpec:
    """C5: Drude IL slab + Langevin electrode subset + massless image particles (examples/run-edl.py:82-100).
    Default counts follow SURVEY.md §8d C5: 2 496 electrode atoms + 18 907 IL particles + 18 907 images.
    Each image is in its parent's molecule (run-edl.py:95 bonds them; quirk Q11) and has mass 0."""
    rng = np.random.default_rng(seed)
    il = drude_il(cells=(1, 1, 1), pairs_per_cell=num_ion_pairs, T=T, T_drude=T_drude, seed=seed + 1)
    lz = 16.0
    n_il = il.num_atoms
    # squeeze the IL into the lower half of the box (z in (0.5, lz/2 - 0.5)); mirror plane at lz/2
    z = il.positions[:, 2]
    il.positions[:, 2] = 0.5 + (z - z.min()) / max(z.max() - z.min(), 1e-9) * (lz / 2 - 1.0)
    # electrode: alternating Mo / S atoms near z = 0.2, every atom its own molecule, Langevin-thermostatted
    m_el = np.where(np.arange(num_electrode) % 3 == 0, 95.94, 32.06)
    q_el = np.zeros(num_electrode)
    p_el = np.stack([rng.uniform(0, il.box[0], num_electrode), rng.uniform(0, il.box[1], num_electrode),
                     rng.uniform(0.1, 0.3, num_electrode)], axis=1)
    v_el = rng.standard_normal((num_electrode, 3)) * np.sqrt(BOLTZ * T / m_el)[:, None]
    # order as in edl_*/conf.gro: electrode, IL, images
    o_il = num_electrode
    o_img = num_electrode + n_il
    masses = np.concatenate([m_el, il.masses, np.zeros(n_il)])
    charges = np.concatenate([q_el, il.charges, -il.charges])
    pos_img = il.positions.copy()
    pos_img[:, 2] = lz - pos_img[:, 2]          # 2*mirror - z with mirror = lz/2
    positions = np.concatenate([p_el, il.positions, pos_img])
    velocities = np.concatenate([v_el, il.velocities, np.zeros((n_il, 3))])
    nmol_il = il.num_molecules
    mol_id = np.concatenate([np.arange(num_electrode, dtype=np.int32),
                             il.mol_id + num_electrode, il.mol_id + num_electrode]).astype(np.int32)
    spec = SystemSpec(name=f"edl_{num_ion_pairs}", masses=masses, charges=charges, positions=positions,
                      velocities=velocities, box=np.array([il.box[0], il.box[1], lz]), mol_id=mol_id,
                      drude_pairs=(il.drude_pairs + o_il).astype(np.int32), constraints=np.zeros((0, 2), np.int32),
                      has_cm_motion_remover=False)
    spec.particles_ld = list(range(num_electrode))
    spec.image_pairs = [(o_img + i, o_il + i) for i in range(n_il)]
    spec.particles_electrolyte = list(range(o_il, o_il + n_il))
    assert nmol_il + num_electrode == spec.num_molecules
    return spec


def constrain_hydrogens(spec: SystemSpec, distance: float = 0.109) -> SystemSpec:
    """HBonds constraints as examples/ommhelper/oplspsffile.py:952-955 asks of OpenMM: every H (mass 1.008 here) is constrained to
    the heavy particle in front of it in its molecule (the generator lists a heavy atom, its Drude, then its hydrogens), and the
    hydrogens are put at exactly `distance` from it so the initial state satisfies the constraints."""
    if spec.name.startswith(("bulk_Im21", "edl_Im21")):
        raise ValueError("the reference-derived systems carry their own HBonds constraints: make_config(..., hbonds=True)")
    m = spec.masses
    cons = []
    heavy = -1
    for i in range(spec.num_atoms):
        if i > 0 and spec.mol_id[i] != spec.mol_id[i - 1]:
            heavy = -1
        if m[i] > 1.5:
            heavy = i
        elif abs(m[i] - 1.008) < 1e-9 and heavy >= 0 and len([c for c in cons if c[1] == heavy]) < 3:
            cons.append((i, heavy))
    cons = np.array(cons, dtype=np.int32).reshape(-1, 2)
    rng = np.random.default_rng(SEED + 7)
    u = rng.standard_normal((len(cons), 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    spec.positions[cons[:, 0]] = spec.positions[cons[:, 1]] + distance * u
    # remove the velocity component along each bond (relative velocity), so the start also satisfies the velocity constraints
    rel = spec.velocities[cons[:, 0]] - spec.velocities[cons[:, 1]]
    spec.velocities[cons[:, 0]] -= (rel * u).sum(1)[:, None] * u
    spec.constraints = cons
    spec.constraint_distances = np.full(len(cons), distance)
    return spec


def constrain_all_bonds(spec: SystemSpec, cutoff: float = 0.165, hangles: bool = False) -> SystemSpec:
    """constraints=AllBonds (and HAngles) as examples/ommhelper/oplspsffile.py:948-951 can ask of OpenMM: every bond becomes a distance constraint,
    with HAngles also the H-X-H angles (as a distance between the two hydrogens).  The topology files' bond lists are not part of the
    fixtures, so bonds are found from the geometry: two real (non-Drude, massive) particles of one molecule closer than `cutoff` nm in
    the initial positions are bonded (C-H 0.109, C-C / C-N 0.13-0.15 nm in conf.gro), and the constraint length is their distance there, so
    the start satisfies the position constraints.  Rings (the imidazolium ring) and chains (dicyanamide) give constraint graphs that are
    neither hydrogen-type clusters nor rigid triangles: what OpenMM hands to CCMA and the fused kernels to their general solver."""
    is_drude = np.zeros(spec.num_atoms, dtype=bool)
    pairs = np.asarray(spec.drude_pairs).reshape(-1, 2)
    if len(pairs):
        is_drude[pairs[:, 0]] = True
    real = (~is_drude) & (spec.masses > 0)
    cons, dist = [], []
    order = np.argsort(spec.mol_id, kind="stable")
    bounds = np.flatnonzero(np.diff(spec.mol_id[order])) + 1
    for grp in np.split(order, bounds):
        idx = np.sort(grp[real[grp]])
        if idx.size < 2:
            continue
        x = spec.positions[idx]
        d = np.linalg.norm(x[:, None, :] - x[None, :, :], axis=2)
        bonded = (d < cutoff) & np.triu(np.ones_like(d, dtype=bool), 1)
        for i, j in zip(*np.nonzero(bonded)):
            cons.append((int(idx[i]), int(idx[j]))); dist.append(float(d[i, j]))
        if hangles:
            light = spec.masses[idx] < 1.5
            adj = (d < cutoff) & ~np.eye(idx.size, dtype=bool)
            for c in range(idx.size):
                hs = [h for h in np.nonzero(adj[c] & light)[0] if not light[c]]
                for p in range(len(hs)):
                    for q in range(p + 1, len(hs)):
                        cons.append((int(idx[hs[p]]), int(idx[hs[q]]))); dist.append(float(d[hs[p], hs[q]]))
    spec.constraints = np.array(cons, dtype=np.int32).reshape(-1, 2)
    spec.constraint_distances = np.array(dist, dtype=np.float64)
    # remove the bond-parallel relative velocities (a few Gauss-Seidel passes: the constraints are coupled) so that the start is near the velocity manifold
    v, m = spec.velocities, spec.masses
    ca, cb = spec.constraints[:, 0], spec.constraints[:, 1]
    r = spec.positions[ca] - spec.positions[cb]
    rr = (r * r).sum(1)
    ima, imb = 1.0 / m[ca], 1.0 / m[cb]
    for _ in range(60):                              # damped Jacobi passes over all constraints at once (vectorised: the full C3 box has 66 000 of them)
        rv = ((v[ca] - v[cb]) * r).sum(1) / rr
        np.subtract.at(v, ca, 0.5 * r * (rv * ima / (ima + imb))[:, None])
        np.add.at(v, cb, 0.5 * r * (rv * imb / (ima + imb))[:, None])
    return spec


def add_random_constraints(spec: SystemSpec, rng, max_degree: int = 6) -> SystemSpec:
    """Random distance constraints inside the molecules of `spec` (tools/probes/fuzz_constraints.py, tests/test_gpu_general_constraints.py):
    a random spanning forest over each molecule's real (massive, non-Drude) particles plus a few ring-closing edges -- chains, stars, rings,
    triangles, at most as many constraints as particles so that none is redundant -- with the present distances as lengths (the start
    satisfies the position constraints) and the bond-parallel relative velocities removed."""
    is_drude = np.zeros(spec.num_atoms, dtype=bool)
    pairs = np.asarray(spec.drude_pairs).reshape(-1, 2)
    if len(pairs):
        is_drude[pairs[:, 0]] = True
    real = (~is_drude) & (spec.masses > 0)
    cons = []
    for m in np.unique(spec.mol_id):
        idx = np.nonzero((spec.mol_id == m) & real)[0]
        if idx.size < 2 or rng.random() < 0.2:
            continue
        deg = {int(i): 0 for i in idx}
        perm = [int(i) for i in rng.permutation(idx)]
        edges = set()
        for k in range(1, len(perm)):                    # a random tree, partly cut into a forest
            if rng.random() < 0.25:
                continue
            cand = [q for q in perm[:k] if deg[q] < max_degree]
            if not cand:
                continue
            q = cand[int(rng.integers(0, len(cand)))]
            edges.add((min(perm[k], q), max(perm[k], q))); deg[perm[k]] += 1; deg[q] += 1
        for _ in range(int(rng.integers(0, 3))):         # ring closures
            a, b = (int(i) for i in rng.choice(idx, 2, replace=False))
            e = (min(a, b), max(a, b))
            if e not in edges and deg[a] < max_degree and deg[b] < max_degree and len(edges) < idx.size:
                edges.add(e); deg[a] += 1; deg[b] += 1
        cons += sorted(edges)
    cons = np.array(cons, dtype=np.int32).reshape(-1, 2)
    out = SystemSpec(name=spec.name + "+random constraints", masses=spec.masses, charges=spec.charges, positions=spec.positions.copy(),
                     velocities=spec.velocities.copy(), box=spec.box, mol_id=spec.mol_id, drude_pairs=spec.drude_pairs, constraints=cons,
                     constraint_distances=np.linalg.norm(spec.positions[cons[:, 0]] - spec.positions[cons[:, 1]], axis=1),
                     has_cm_motion_remover=spec.has_cm_motion_remover, particles_ld=list(spec.particles_ld), image_pairs=list(spec.image_pairs),
                     particles_electrolyte=list(spec.particles_electrolyte), virtual_sites=list(spec.virtual_sites))
    v, mss = out.velocities, out.masses
    ca, cb = cons[:, 0], cons[:, 1]
    r = out.positions[ca] - out.positions[cb]
    rr = (r * r).sum(1)
    ima, imb = 1.0 / mss[ca], 1.0 / mss[cb]
    for _ in range(500):                                  # Gauss-Seidel passes, one constraint after the other (small systems only; converges for any graph)
        worst = 0.0
        for k in range(len(cons)):
            a, b = ca[k], cb[k]
            rv = float(((v[a] - v[b]) * r[k]).sum() / rr[k])
            v[a] -= r[k] * (rv * ima[k] / (ima[k] + imb[k]))
            v[b] += r[k] * (rv * imb[k] / (ima[k] + imb[k]))
            worst = max(worst, abs(rv))
        if worst < 1e-12:
            break
    return out


def rigid_water(spec: SystemSpec, d_oh: float = 0.1, d_hh: float = 0.1633) -> SystemSpec:
    """Rigid three-site water as OpenMM's rigidWater=True asks for it: O-H, O-H and H-H constrained in every (O, H, H) molecule
    of `spec`; the hydrogens are put on the rigid geometry and the molecule's velocity is made rigid-body compatible (the
    bond-parallel relative velocities are removed), so the start lies on the constraint manifold."""
    n = spec.num_atoms
    assert n % 3 == 0
    o = np.arange(0, n, 3)
    rng = np.random.default_rng(SEED + 11)
    u = rng.standard_normal((len(o), 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    w = rng.standard_normal((len(o), 3)); w -= (w * u).sum(1)[:, None] * u; w /= np.linalg.norm(w, axis=1)[:, None]
    half = np.arcsin(0.5 * d_hh / d_oh)
    spec.positions[o + 1] = spec.positions[o] + d_oh * (np.cos(half) * u + np.sin(half) * w)
    spec.positions[o + 2] = spec.positions[o] + d_oh * (np.cos(half) * u - np.sin(half) * w)
    cons = np.stack([np.stack([o + 1, o], 1), np.stack([o + 2, o], 1), np.stack([o + 1, o + 2], 1)], 1).reshape(-1, 2)
    dist = np.tile([d_oh, d_oh, d_hh], len(o))
    # rigid-body velocities: v_i = V_com + omega x (r_i - r_com), with V_com the molecule's COM velocity and a random omega
    m = spec.masses.reshape(-1, 3)
    x = spec.positions.reshape(-1, 3, 3)
    v = spec.velocities.reshape(-1, 3, 3)
    com = (m[:, :, None] * x).sum(1) / m.sum(1)[:, None]
    vcom = (m[:, :, None] * v).sum(1) / m.sum(1)[:, None]
    omega = rng.standard_normal((len(o), 3)) * 10.0
    spec.velocities = (vcom[:, None, :] + np.cross(omega[:, None, :], x - com[:, None, :])).reshape(-1, 3)
    spec.constraints = cons.astype(np.int32)
    spec.constraint_distances = dist
    return spec


# ---------------------------------------------------------------------------------------------------------------------------
# The reference's own example models: tests/golden/topo_{bulk_Im21,edl_Im21}.npz hold masses, charges,
# molecule ids, Drude pairs, HBonds constraints, positions and the Langevin / image / electrolyte sets parsed from
# examples/models/{bulk_Im21,edl_Im21} by tests/golden/make_topologies.py (index and number data, no file text).  The particle order
# inside the real ions differs from the procedural look-alike above (the hydrogens of c2c1im+ come after the ring, not next to their
# carbon), which is what decides the wave packing and the SHAKE cluster shapes.
_TOPO_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


VSITE_PARAMS = {       # kind -> (number of parents, parameters): include/vvhip.h VVHIP_VSITE_*
    0: (2, (0.6, 0.4)),
    1: (3, (0.786646558, 0.106676721, 0.106676721)),                              # TIP4P-Ew's M site
    2: (3, (0.3, 0.25, -2.0)),                                                    # out of plane: w12, w13, wCross (1/nm)
    3: (3, (1.0, 0.0, 0.0, 1.0, -1.0, 0.0, 0.0, -1.0, 1.0, 0.03, 0.02, -0.01)),   # a lone pair as examples/ommhelper/oplspsffile.py:991 builds it
}


def virtual_site_position(kind, prm, p1, p2, p3=None):
    """The definitions OpenMM documents for its four site classes, in float64 (initial positions of the synthetic systems only)."""
    if kind == 0:
        return p1 * prm[0] + p2 * prm[1]
    if kind == 1:
        return p1 * prm[0] + p2 * prm[1] + p3 * prm[2]
    if kind == 2:
        a, b = p2 - p1, p3 - p1
        return p1 + a * prm[0] + b * prm[1] + np.cross(a, b) * prm[2]
    o = p1 * prm[0] + p2 * prm[1] + p3 * prm[2]
    x = p1 * prm[3] + p2 * prm[4] + p3 * prm[5]
    y = p1 * prm[6] + p2 * prm[7] + p3 * prm[8]
    z = np.cross(x, y)
    x, z = x / np.linalg.norm(x), z / np.linalg.norm(z)
    y = np.cross(z, x)
    return o + x * prm[9] + y * prm[10] + z * prm[11]


def add_virtual_sites(spec: SystemSpec, kinds=(1,), interleaved: bool = True) -> SystemSpec:
    """One massless virtual site per molecule and entry of `kinds` (molecules with fewer massive non-Drude particles than the site needs
    parents get none), hanging on the molecule's first massive non-Drude particles.  interleaved: the sites follow their molecule's last
    particle (a TIP4P-like O H H M layout, everything renumbered); else they are appended behind the last particle of the system, in
    their parent's molecule (as run-edl.py's images are)."""
    n = spec.num_atoms
    mol = np.asarray(spec.mol_id)
    drude = np.zeros(n, bool)
    if len(spec.drude_pairs):
        drude[np.asarray(spec.drude_pairs)[:, 0]] = True
    order = np.argsort(mol, kind="stable")
    bounds = np.nonzero(np.diff(mol[order], prepend=-1, append=mol.max() + 1))[0]
    new_of_old = np.zeros(n, dtype=np.int64)
    sites = []                                   # (position in the new numbering, molecule, kind, parents (old numbering))
    nxt, tail = 0, []
    contiguous = bool(np.all(np.diff(mol) >= 0))
    assert contiguous or not interleaved, "interleaved sites need molecules stored one after the other"
    for b in range(len(bounds) - 1):
        members = order[bounds[b]:bounds[b + 1]]
        for i in members:
            new_of_old[i] = nxt
            nxt += 1
        heavy = [i for i in members if spec.masses[i] != 0 and not drude[i]]
        for kind in kinds:
            npar = VSITE_PARAMS[kind][0]
            if len(heavy) < npar:
                continue
            if interleaved:
                sites.append((nxt, int(mol[members[0]]), kind, heavy[:npar]))
                nxt += 1
            else:
                tail.append((int(mol[members[0]]), kind, heavy[:npar]))
    if not interleaved:
        new_of_old = np.arange(n, dtype=np.int64)
        sites = [(n + k, m, kind, par) for k, (m, kind, par) in enumerate(tail)]
    ntot = n + len(sites)
    def spread(a, fill):
        out = np.full((ntot,) + a.shape[1:], fill, dtype=a.dtype)
        out[new_of_old] = a
        return out
    masses, charges = spread(np.asarray(spec.masses, float), 0.0), spread(np.asarray(spec.charges, float), 0.0)
    pos, vel = spread(np.asarray(spec.positions, float), 0.0), spread(np.asarray(spec.velocities, float), 0.0)
    molid = spread(mol.astype(np.int32), 0)
    vs = []
    for at, m, kind, par in sites:
        prm = VSITE_PARAMS[kind][1]
        pp = [np.asarray(spec.positions[i], float) for i in par]
        pos[at] = virtual_site_position(kind, prm, *pp)
        molid[at] = m
        charges[at] = -0.3
        vs.append((int(at), int(kind), tuple(int(new_of_old[i]) for i in par), tuple(prm)))
    ren = lambda a: new_of_old[np.asarray(a, dtype=np.int64)].astype(np.int32) if len(a) else np.asarray(a, dtype=np.int32)
    out = SystemSpec(name=spec.name + "+vsites", masses=masses, charges=charges, positions=pos, velocities=vel, box=spec.box, mol_id=molid,
                     drude_pairs=ren(spec.drude_pairs).reshape(-1, 2), constraints=ren(spec.constraints).reshape(-1, 2),
                     constraint_distances=spec.constraint_distances, has_cm_motion_remover=spec.has_cm_motion_remover,
                     particles_ld=[int(new_of_old[i]) for i in spec.particles_ld],
                     image_pairs=[(int(new_of_old[a]), int(new_of_old[b])) for a, b in spec.image_pairs],
                     particles_electrolyte=[int(new_of_old[i]) for i in spec.particles_electrolyte], virtual_sites=vs)
    return out


def add_random_virtual_sites(spec: SystemSpec, rng) -> SystemSpec:
    """Random virtual sites behind the last particle (tools/probes/fuzz_sites.py, tests/test_gpu_virtual_sites.py): on ~70 % of the molecules one to
    four sites of random kind and weights, hanging on any massive particles of the molecule (Drude particles included), several per parent."""
    n = spec.num_atoms
    mol = np.asarray(spec.mol_id)
    masses = np.asarray(spec.masses, float)
    extra = []
    for m in np.unique(mol):
        if rng.random() < 0.3:
            continue
        members = np.nonzero((mol == m) & (masses != 0))[0]
        for _ in range(int(rng.integers(1, 5))):
            kind = int(rng.integers(0, 4))
            npar = 2 if kind == 0 else 3
            if len(members) < npar:
                continue
            par = [int(i) for i in rng.choice(members, npar, replace=False)]
            if kind == 0:
                w = rng.uniform(-0.5, 1.5); prm = (w, 1 - w)
            elif kind == 1:
                a, b = rng.uniform(-0.3, 0.9, 2); prm = (a, b, 1 - a - b)
            elif kind == 2:
                prm = tuple(rng.uniform(-0.5, 0.5, 2)) + (float(rng.uniform(-3, 3)),)
            else:
                ow = rng.uniform(-0.2, 0.8, 2); xw = rng.uniform(-1, 1, 2); yw = rng.uniform(-1, 1, 2)
                prm = (ow[0], ow[1], 1 - ow.sum(), xw[0], xw[1], -xw.sum(), yw[0], yw[1], -yw.sum()) + tuple(rng.uniform(-0.05, 0.05, 3))
            extra.append((int(m), kind, par, tuple(float(x) for x in prm)))
    k = len(extra)
    pos = np.concatenate([spec.positions, np.zeros((k, 3))])
    vs = []
    for j, (m, kind, par, prm) in enumerate(extra):
        pos[n + j] = virtual_site_position(kind, prm, *[np.asarray(spec.positions[i], float) for i in par])
        vs.append((n + j, kind, tuple(par), prm))
    out = SystemSpec(name=spec.name + "+random sites", masses=np.concatenate([masses, np.zeros(k)]), charges=np.concatenate([spec.charges, np.full(k, -0.2)]),
                       positions=pos, velocities=np.concatenate([spec.velocities, np.zeros((k, 3))]), box=spec.box,
                       mol_id=np.concatenate([mol, np.array([e[0] for e in extra], dtype=mol.dtype)]).astype(np.int32), drude_pairs=spec.drude_pairs,
                       constraints=spec.constraints, constraint_distances=spec.constraint_distances, has_cm_motion_remover=spec.has_cm_motion_remover,
                       particles_ld=list(spec.particles_ld), image_pairs=list(spec.image_pairs), particles_electrolyte=list(spec.particles_electrolyte),
                       virtual_sites=vs)
    return out


def have_reference_topologies() -> bool:
    return all(os.path.exists(os.path.join(_TOPO_DIR, f)) for f in ("topo_bulk_Im21.npz", "topo_edl_Im21.npz"))


def _on_constraint_manifold(pos, vel, cons, dist):
    """Hydrogens moved along their bond to the constraint length (conf.gro has three decimals) and the bond-parallel relative velocity
    removed, so that the start satisfies position and velocity constraints."""
    if len(cons) == 0:
        return
    h, x = cons[:, 0], cons[:, 1]
    u = pos[h] - pos[x]
    u /= np.linalg.norm(u, axis=1)[:, None]
    pos[h] = pos[x] + dist[:, None] * u
    rel = vel[h] - vel[x]
    vel[h] -= (rel * u).sum(1)[:, None] * u


def bulk_Im21(cells=(2, 2, 3), pairs_per_cell=250, hbonds=False, T=333.0, T_drude=1.0, seed=SEED) -> SystemSpec:
    """C3 / C4 as SURVEY.md section 8 defines them: examples/models/bulk_Im21 (250 c2c1im+ + 250 dca-, 9 250 particles, box 3.1 x 3.1 x
    6.1 nm) tiled `cells` times -> 111 000 particles, 6 000 molecules, 39 000 Drude pairs for 2 x 2 x 3.  `pairs_per_cell` < 250 keeps the
    first ion pairs of the cell only (reduced copies for fast tests).  hbonds: the 11 X-H constraints per cation that constraints=HBonds
    puts on the System (examples/run-bulk.py), solved by the fused kernels."""
    z = np.load(os.path.join(_TOPO_DIR, "topo_bulk_Im21.npz"))
    mol0 = z["mol_id"]
    k = int(pairs_per_cell)
    assert 1 <= k <= 250
    keep = np.nonzero((mol0 < k) | ((mol0 >= 250) & (mol0 < 250 + k)))[0]        # conf.gro lists the 250 cations, then the 250 anions
    renum = -np.ones(mol0.size, dtype=np.int64)
    renum[keep] = np.arange(keep.size)
    n1 = keep.size
    masses1, charges1, pos1 = z["masses"][keep], z["charges"][keep], z["positions"][keep].astype(np.float64)
    _, mol1 = np.unique(mol0[keep], return_inverse=True)
    pairs1 = renum[z["drude_pairs"][np.isin(z["drude_pairs"][:, 0], keep)]]
    sel = np.isin(z["constraints"][:, 0], keep)
    cons1, dist1 = renum[z["constraints"][sel]], z["constraint_distances"][sel]
    box1 = z["box"].astype(np.float64)
    cx, cy, cz = cells
    ncell = cx * cy * cz
    nmol1 = int(mol1.max()) + 1
    shifts = np.array([[i, j, l] for i in range(cx) for j in range(cy) for l in range(cz)], dtype=np.float64) * box1
    masses = np.tile(masses1, ncell)
    charges = np.tile(charges1, ncell)
    pos = (pos1[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
    off = (np.arange(ncell) * n1)[:, None, None]
    pairs = (pairs1[None, :, :] + off).reshape(-1, 2).astype(np.int32)
    cons = (cons1[None, :, :] + off).reshape(-1, 2).astype(np.int32)
    dist = np.tile(dist1, ncell)
    mol = (mol1[None, :] + (np.arange(ncell) * nmol1)[:, None]).reshape(-1).astype(np.int32)
    isd = np.zeros(masses.size, dtype=bool)
    isd[pairs[:, 0]] = True
    parent_of = np.arange(masses.size) - 1
    rng = np.random.default_rng(seed)
    vel = _maxwell_boltzmann(rng, masses, isd, parent_of, T, T_drude)
    spec = SystemSpec(name=f"bulk_Im21_{cx}x{cy}x{cz}" + ("" if k == 250 else f"x{k}"), masses=masses, charges=charges, positions=pos, velocities=vel,
                      box=box1 * np.array(cells, dtype=np.float64), mol_id=mol, drude_pairs=pairs, constraints=np.zeros((0, 2), np.int32))
    if hbonds:
        _on_constraint_manifold(spec.positions, spec.velocities, cons, dist)
        spec.constraints, spec.constraint_distances = cons, dist
    return spec


def nondrude_Im21(num_pairs=83, hbonds=False, T=333.0, seed=SEED) -> SystemSpec:
    """C1 as SURVEY.md section 8d defines it: the first `num_pairs` ion pairs of examples/models/bulk_Im21 with the Drude particles
    stripped -- every Drude's mass and charge go back to its parent -- : 83 pairs = 1 992 particles (19-atom c2c1im+, 5-atom dca-), 166
    molecules, no Drude pairs, so the plain Nose-Hoover thermostat with one temperature group (VVIntegrator.cpp:106-108)."""
    z = np.load(os.path.join(_TOPO_DIR, "topo_bulk_Im21.npz"))
    mol0 = z["mol_id"]
    k = int(num_pairs)
    assert 1 <= k <= 250
    masses0, charges0 = z["masses"].astype(np.float64).copy(), z["charges"].astype(np.float64).copy()
    drudes, parents = z["drude_pairs"][:, 0], z["drude_pairs"][:, 1]
    np.add.at(masses0, parents, masses0[drudes])
    np.add.at(charges0, parents, charges0[drudes])
    is_drude = np.zeros(mol0.size, dtype=bool)
    is_drude[drudes] = True
    keep = np.nonzero(((mol0 < k) | ((mol0 >= 250) & (mol0 < 250 + k))) & ~is_drude)[0]      # conf.gro: 250 cations, then 250 anions
    renum = -np.ones(mol0.size, dtype=np.int64)
    renum[keep] = np.arange(keep.size)
    _, mol = np.unique(mol0[keep], return_inverse=True)
    masses = masses0[keep]
    rng = np.random.default_rng(seed)
    none = np.zeros(masses.size, dtype=bool)
    vel = _maxwell_boltzmann(rng, masses, none, np.arange(masses.size) - 1, T, 1.0)
    spec = SystemSpec(name=f"bulk_Im21_nondrude_{k}", masses=masses, charges=charges0[keep], positions=z["positions"][keep].astype(np.float64),
                      velocities=vel, box=z["box"].astype(np.float64), mol_id=mol.astype(np.int32), drude_pairs=np.zeros((0, 2), np.int32),
                      constraints=np.zeros((0, 2), np.int32))
    if hbonds:
        sel = np.isin(z["constraints"][:, 0], keep) & np.isin(z["constraints"][:, 1], keep)
        cons, dist = renum[z["constraints"][sel]].astype(np.int32), z["constraint_distances"][sel]
        _on_constraint_manifold(spec.positions, spec.velocities, cons, dist)
        spec.constraints, spec.constraint_distances = cons, dist
    return spec


def edl_Im21(hbonds=False, T=333.0, T_drude=1.0, seed=SEED) -> SystemSpec:
    """C5 as SURVEY.md section 8 defines it: examples/models/edl_Im21/conf.gro -- 2 496 MoS2 atoms (Langevin subset), 511 ion pairs =
    18 907 ionic-liquid particles (Nose-Hoover, electrolyte for the field), 18 907 massless images; mirror at Lz / 2 = 8 nm."""
    z = np.load(os.path.join(_TOPO_DIR, "topo_edl_Im21.npz"))
    masses, pairs = z["masses"].astype(np.float64), z["drude_pairs"].astype(np.int32)
    isd = np.zeros(masses.size, dtype=bool)
    isd[pairs[:, 0]] = True
    rng = np.random.default_rng(seed)
    vel = _maxwell_boltzmann(rng, masses, isd, np.arange(masses.size) - 1, T, T_drude)
    spec = SystemSpec(name="edl_Im21", masses=masses, charges=z["charges"].astype(np.float64), positions=z["positions"].astype(np.float64),
                      velocities=vel, box=z["box"].astype(np.float64), mol_id=z["mol_id"].astype(np.int32), drude_pairs=pairs,
                      constraints=np.zeros((0, 2), np.int32), has_cm_motion_remover=False)
    spec.particles_ld = [int(i) for i in z["particles_ld"]]
    spec.image_pairs = [(int(a), int(b)) for a, b in z["image_pairs"]]
    spec.particles_electrolyte = [int(i) for i in z["particles_electrolyte"]]
    if hbonds:
        cons, dist = z["constraints"].astype(np.int32), z["constraint_distances"].astype(np.float64)
        _on_constraint_manifold(spec.positions, spec.velocities, cons, dist)
        spec.constraints, spec.constraint_distances = cons, dist
    return spec


def make_config(name: str, scale: float = 1.0, hbonds: bool = False, synthetic: bool = False) -> SystemSpec:
    """BASELINE.json configs by id.  C1 / C3 / C4 / C5 are built from the reference's own example models (C1: the first 83 ion pairs of
    bulk_Im21 with the Drude particles merged into their parents; C3 / C4: bulk_Im21 tiled 2 x 2 x 3; C5: edl_Im21);
    `synthetic=True` (or missing fixtures) gives the procedural look-alikes instead.  `scale` > 1 tiles C3 further along z, `scale` < 1
    gives a reduced copy for fast parity tests.  hbonds: with the HBonds constraints of the example scripts (rigid water for C2)."""
    real = have_reference_topologies() and not synthetic
    if name == "C1":
        if real:
            return nondrude_Im21(max(1, min(250, int(round(83 * scale)))), hbonds=hbonds)
        spec = nondrude_il(max(2, int(round(83 * scale))))
        return constrain_hydrogens(spec) if hbonds else spec
    if name == "C2":
        spec = spce_water(max(4, int(round(3333 * scale))))
        return rigid_water(spec) if hbonds else spec
    if name in ("C3", "C4"):
        if real:
            if scale >= 1.0:
                return bulk_Im21(cells=(2, 2, 3 * int(round(scale))), hbonds=hbonds)
            return bulk_Im21(cells=(1, 1, 1), pairs_per_cell=max(2, min(250, int(round(3000 * scale)))), hbonds=hbonds)
        spec = drude_il(cells=(2, 2, 3 * int(round(scale)))) if scale >= 1.0 else drude_il(cells=(1, 1, 1), pairs_per_cell=max(2, int(round(3000 * scale))))
        return constrain_hydrogens(spec) if hbonds else spec
    if name == "C5":
        if real and scale >= 1.0:
            return edl_Im21(hbonds=hbonds)
        spec = edl_slab(max(2, int(round(511 * scale))), max(3, int(round(2496 * scale))))
        return constrain_hydrogens(spec) if hbonds else spec
    raise ValueError(f"unknown config {name!r}")
