"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end for
  * liboracle_{single,mixed,double}.so  -- our C restatement (vv_oracle.c), and
  * _ref/libvvref_{...}.so              -- the reference's own kernels built for the CPU
                                           (only where `make ref` was run: this container),
plus a numpy restatement of the reference's host-side initialisation:
  VVIntegrator::initialize            openmmapi/src/VVIntegrator.cpp:92-188
  CudaModifyDrudeNoseKernel::initialize   platforms/cuda/src/CudaVVKernels.cpp:462-594
  CudaModifyDrudeLangevinKernel::initialize  ...:775-804

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BOLTZ = (1.380649e-23 * 6.02214076e23) / 1000.0
AVOGADRO = 6.02214076e23
TG_ATOM, TG_COM, TG_DRUDE = 0, 1, 2
MAX_CHAINS = 8

PRECISIONS = ("single", "mixed", "double")
REAL = {"single": np.float32, "mixed": np.float32, "double": np.float64}
MIXED = {"single": np.float32, "mixed": np.float64, "double": np.float64}


class OracleError(Exception):
    """Stands in for OpenMMException in the restated init logic."""


@dataclass
class Params:
    """VVIntegrator's parameters with the constructor defaults of openmmapi/src/VVIntegrator.cpp:46-70."""
    temperature: float = 300.0
    frequency: float = 10.0
    drude_temperature: float = 1.0
    drude_frequency: float = 40.0
    step_size: float = 0.001
    num_chains: int = 3
    loops_per_step: int = 1
    max_drude_distance: float = 0.0
    friction: float = 5.0
    drude_friction: float = 20.0
    mirror_location: float = 0.0
    electric_field: float = 0.0          # kJ/(nm e) per particle, as the C++ API takes it (quirk Q12)
    cos_acceleration: float = 0.0
    use_com_temp_group: bool = False
    use_middle_scheme: bool = True
    auto_set_com_temp_group: bool = True
    auto_set_friction: bool = True


def build(target: str = "all") -> None:
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


def have_ref() -> bool:
    return all(os.path.exists(os.path.join(HERE, "_ref", f"libvvref_{p}.so")) for p in PRECISIONS)


# ----------------------------------------------------------------------------------------------
# host-side init, restated in numpy
# ----------------------------------------------------------------------------------------------
def build_tables(spec, params: Params) -> Dict:
    """Returns every index table / scalar the reference's initialize() methods derive."""
    n = spec.num_atoms
    p = Params(**vars(params))
    has_drude = len(spec.drude_pairs) > 0
    # API:106-121
    if not has_drude:
        if p.auto_set_com_temp_group:
            p.use_com_temp_group = False
        if p.auto_set_friction:
            p.friction = 1.0
    else:
        if p.auto_set_com_temp_group:
            p.use_com_temp_group = True
        if p.auto_set_friction:
            p.friction = 5.0
    mol_id = np.asarray(spec.mol_id, dtype=np.int32)
    nmol = spec.num_molecules
    mol_mass = np.zeros(nmol)
    np.add.at(mol_mass, mol_id, spec.masses)                       # API:130-135
    with np.errstate(divide="ignore"):
        mol_inv_mass = 1.0 / mol_mass
    is_ld = np.zeros(n, bool)
    is_ld[list(spec.particles_ld)] = True
    is_img = np.zeros(n, bool)
    is_img[[i for i, _ in spec.image_pairs]] = True
    is_nh = ~is_ld & ~is_img                                        # API:138-145
    particles_nh = np.nonzero(is_nh)[0].astype(np.int32)
    _, first = np.unique(mol_id[particles_nh], return_index=True)
    molecules_nh = mol_id[particles_nh][np.sort(first)].astype(np.int32)   # order of first appearance
    if np.isin(mol_id[is_ld], molecules_nh).any():                  # API:146-151
        raise OracleError("NH and Langevin thermostat cannot be applied on the same molecule")
    if is_ld.any() and p.cos_acceleration != 0:                     # API:154-155
        raise OracleError("Langevin thermostat and periodic perturbation shouldn't be used together")

    # HOST:483-494: particles sorted by molecule id, (count, start) per molecule
    order = np.argsort(mol_id, kind="stable").astype(np.int32)
    counts = np.bincount(mol_id, minlength=nmol).astype(np.int32)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    particles_in_molecules = np.stack([counts, starts], axis=1).astype(np.int32)

    dof = np.zeros(3)
    massive_nh = is_nh & (spec.masses != 0)
    # HOST:505-510 (summed in particle order, as the reference does)
    for i in np.nonzero(massive_nh)[0]:
        dof[TG_ATOM] += 3
        if p.use_com_temp_group:
            dof[TG_ATOM] -= 3 * spec.masses[i] * mol_inv_mass[mol_id[i]]
    nh_left = set(particles_nh.tolist())
    pairs_nh = []
    for d, par in np.asarray(spec.drude_pairs).reshape(-1, 2):      # HOST:513-528
        if is_nh[d] != is_nh[par]:
            raise OracleError("Drude particle and its parent atom should be in the same thermostat")
        if is_nh[d]:
            nh_left.discard(int(d))
            nh_left.discard(int(par))
            pairs_nh.append((int(d), int(par)))
            dof[TG_ATOM] -= 3
            dof[TG_DRUDE] += 3
    normal_nh = np.array(sorted(nh_left), dtype=np.int32)           # HOST:529 (std::set order)
    for a, b in np.asarray(spec.constraints).reshape(-1, 2):        # HOST:532-541
        if is_nh[a] != is_nh[b]:
            raise OracleError("Constrained particle pair should be in the same thermostat")
        if is_nh[a]:
            dof[TG_ATOM] -= 1
    if p.use_com_temp_group:                                        # HOST:547-558
        dof[TG_COM] = 3 * len(molecules_nh)
    if spec.has_cm_motion_remover:
        if p.use_com_temp_group:
            dof[TG_COM] -= 3
        else:
            dof[TG_ATOM] -= 3
    dof = np.maximum(dof, 0.0)                                      # HOST:563-564
    num_tg = 3                                                      # HOST:567-573
    if dof[TG_DRUDE] == 0:
        num_tg = 2
        if dof[TG_COM] == 0:
            num_tg = 1
    real_kbt = BOLTZ * p.temperature                                # HOST:583-594
    drude_kbt = BOLTZ * p.drude_temperature
    eta_mass = np.zeros((3, MAX_CHAINS))
    nkbt = np.zeros(3)
    for i in range(num_tg):
        kbt = drude_kbt if i == TG_DRUDE else real_kbt
        tg_mass = drude_kbt / p.drude_frequency ** 2 if i == TG_DRUDE else real_kbt / p.frequency ** 2
        nkbt[i] = dof[i] * kbt
        eta_mass[i, 0] = dof[i] * tg_mass
        eta_mass[i, 1:p.num_chains] = tg_mass

    # HOST:775-804
    ld_left = set(np.nonzero(is_ld)[0].tolist())
    pairs_ld = []
    for d, par in np.asarray(spec.drude_pairs).reshape(-1, 2):
        if is_ld[d] != is_ld[par]:
            raise OracleError("Drude particle and its parent atom should be in the same thermostat")
        if is_ld[d]:
            ld_left.discard(int(d))
            ld_left.discard(int(par))
            pairs_ld.append((int(d), int(par)))
    for a, b in np.asarray(spec.constraints).reshape(-1, 2):
        if is_ld[a] != is_ld[b]:
            raise OracleError("Constrained particle pair should be in the same thermostat")
    normal_ld = np.array(sorted(ld_left), dtype=np.int32)

    return dict(
        params=p, particles_nh=particles_nh, molecules_nh=molecules_nh, normal_nh=normal_nh,
        pairs_nh=np.array(pairs_nh, dtype=np.int32).reshape(-1, 2), particle_mol_id=mol_id,
        particles_in_molecules=particles_in_molecules, particles_sorted_by_mol_id=order,
        dof=dof, num_tg=num_tg, eta_mass=eta_mass, nkbt=nkbt,
        normal_ld=normal_ld, pairs_ld=np.array(pairs_ld, dtype=np.int32).reshape(-1, 2),
        num_particles_ld=int(is_ld.sum()),
        image_pairs=np.array(spec.image_pairs, dtype=np.int32).reshape(-1, 2),
        particles_electrolyte=np.array(spec.particles_electrolyte, dtype=np.int32),
        inv_mass_total=1.0 / float(np.cumsum(np.asarray(spec.masses, dtype=np.float64))[-1]),   # HOST:1028-1031: summed in particle order (np.sum adds pairwise: 1 ulp off)
        mol_inv_mass=mol_inv_mass,
    )


# ----------------------------------------------------------------------------------------------
# array layouts shared by oracle, _ref and the product's host mirror
# ----------------------------------------------------------------------------------------------
def build_constraint_clusters(spec):
    """The System's constraints sorted into what the in-kernel solvers take, by OpenMM's admission rules:
    rigid three-site molecules (three mutual constraints, an apex with two equal distances to two equal-mass partners) -> SETTLE;
    everything else must be hydrogen-type clusters (central particle + <= 3 peripherals of equal mass and distance, each
    peripheral in one constraint only) -> SHAKE.  Returns None without constraint distances, else a dict of
    shake_atoms int32 [n,4], shake_params float32 [n,4], settle_atoms int32 [m,3], settle_params float32 [m,2]."""
    cons = np.asarray(spec.constraints).reshape(-1, 2)
    dist = getattr(spec, "constraint_distances", None)
    if dist is None or len(cons) == 0:
        return None
    m = spec.masses
    deg = np.bincount(cons.reshape(-1), minlength=spec.num_atoms)
    dmap = {}
    for (a, b), d in zip(cons, dist):
        dmap[(min(a, b), max(a, b))] = float(d)
    nbr = {}
    for a, b in cons:
        nbr.setdefault(int(a), []).append(int(b))
        nbr.setdefault(int(b), []).append(int(a))
    settle_atoms, settle_params, in_settle = [], [], set()
    for a in sorted(nbr):
        if a in in_settle or deg[a] != 2:
            continue
        b, c = nbr[a]
        if deg[b] != 2 or deg[c] != 2 or (min(b, c), max(b, c)) not in dmap:
            continue
        tri = (a, b, c)
        dd = {(i, j): dmap[(min(i, j), max(i, j))] for i in tri for j in tri if i != j}
        apex = None
        for x in tri:
            y, z = [t for t in tri if t != x]
            if dd[(x, y)] == dd[(x, z)] and m[y] == m[z]:
                apex = (x, min(y, z), max(y, z))
                break
        if apex is None:
            raise OracleError("a rigid triangle without an apex of two equal bonds to equal partners")
        settle_atoms.append(list(apex))
        settle_params.append([dd[(apex[0], apex[1])], dd[(apex[1], apex[2])]])
        in_settle.update(tri)
    clusters = {}
    for (a, b), d in zip(cons, dist):
        if int(a) in in_settle:
            continue
        ctr = a if deg[a] > 1 else (b if deg[b] > 1 else (a if m[a] >= m[b] else b))
        per = b if ctr == a else a
        if deg[per] != 1:
            raise OracleError("constraint topology is neither rigid triangles nor hydrogen-type clusters")
        clusters.setdefault(int(ctr), []).append((int(per), float(d)))
    atoms, params = [], []
    for ctr, lst in clusters.items():
        if len(lst) > 3 or len({d for _, d in lst}) != 1 or len({m[p] for p, _ in lst}) != 1:
            raise OracleError("constraint topology is neither rigid triangles nor hydrogen-type clusters")
        imc, imp, d = 1.0 / m[ctr], 1.0 / m[lst[0][0]], lst[0][1]
        atoms.append([ctr] + [p for p, _ in lst] + [-1] * (3 - len(lst)))
        params.append([imc, 0.5 / (imc + imp), d * d, imp])
    return dict(shake_atoms=np.array(atoms, dtype=np.int32).reshape(-1, 4), shake_params=np.array(params, dtype=np.float32).reshape(-1, 4),
                settle_atoms=np.array(settle_atoms, dtype=np.int32).reshape(-1, 3), settle_params=np.array(settle_params, dtype=np.float32).reshape(-1, 2))


def build_general_constraints(spec):
    """Any other constraint topology (AllBonds, HAngles): every constraint goes to the coloured Gauss-Seidel solver (vvo_general_*), as
    vv::analyze decides for the product.  Colours: greedy in System order, the smallest colour none of the constraints already placed at
    either particle uses -- the rule of csrc/vv_host.cpp, so both sides sweep in the same order.  Returns atoms int32 [n,2] and params
    float32 [n,4] (d^2, 0.5/(1/m_a + 1/m_b), 1/m_a, 1/m_b), sorted by colour, the number of colours and the colour of every row."""
    cons = np.asarray(spec.constraints).reshape(-1, 2)
    dist = np.asarray(spec.constraint_distances, dtype=np.float64)
    m = spec.masses
    used = {}
    rows = []
    for (a, b), d in zip(cons, dist):
        a, b = int(a), int(b)
        taken = used.get(a, 0) | used.get(b, 0)
        colour = 0
        while (taken >> colour) & 1:
            colour += 1
        if colour >= 16:
            raise OracleError("more than 16 colours: the product does not fuse such a System either")
        used[a] = used.get(a, 0) | (1 << colour)
        used[b] = used.get(b, 0) | (1 << colour)
        ima, imb = 1.0 / m[a], 1.0 / m[b]
        rows.append((colour, a, b, d * d, 0.5 / (ima + imb), ima, imb))
    rows.sort(key=lambda r: r[0])          # stable: System order inside a colour
    atoms = np.array([[r[1], r[2]] for r in rows], dtype=np.int32).reshape(-1, 2)
    params = np.array([[r[3], r[4], r[5], r[6]] for r in rows], dtype=np.float32).reshape(-1, 4)
    colours = np.array([r[0] for r in rows], dtype=np.int32)
    return atoms, params, (int(colours.max()) + 1 if rows else 0), colours


GC_OMEGA_PLAIN, GC_OMEGA_TRIANGLES = 1.2, 1.4       # csrc/vv_layout.h: GC_OMEGA_*


def general_relaxation(spec) -> float:
    """Relaxation factor of the general clusters' sweeps, by the product's rule (csrc/vv_host.cpp): 1.4 when three constraints close a
    triangle anywhere in the System (HAngles), else 1.2."""
    adj = {}
    cons = [(int(a), int(b)) for a, b in np.asarray(spec.constraints).reshape(-1, 2)]
    for a, b in cons:
        adj.setdefault(a, set()).add(b)
        adj.setdefault(b, set()).add(a)
    for a, b in cons:
        if (adj[a] & adj[b]) - {a, b}:
            return GC_OMEGA_TRIANGLES
    return GC_OMEGA_PLAIN


def build_shake(spec):
    """(atoms, params) of the SHAKE clusters only; None without constraint distances."""
    c = build_constraint_clusters(spec)
    return None if c is None else (c["shake_atoms"], c["shake_params"])


def padded(n: int) -> int:
    return (n + 31) // 32 * 32          # OpenMM pads the atom count to a multiple of its tile size


def make_state(spec, prec: str) -> Dict[str, np.ndarray]:
    """velm / posq / posqCorrection / force in the reference's device layouts."""
    R, M = REAL[prec], MIXED[prec]
    n = spec.num_atoms
    velm = np.zeros((n, 4), dtype=M)
    velm[:, :3] = spec.velocities
    with np.errstate(divide="ignore"):
        velm[:, 3] = np.where(spec.masses != 0, 1.0 / np.where(spec.masses != 0, spec.masses, 1.0), 0.0)
    posq = np.zeros((n, 4), dtype=R)
    posq[:, :3] = spec.positions
    posq[:, 3] = spec.charges
    corr = np.zeros((n, 4), dtype=R)
    if prec == "mixed":
        corr[:, :3] = spec.positions - posq[:, :3].astype(np.float64)
    return dict(velm=velm, posq=posq, posq_corr=corr, force=np.zeros(3 * padded(n), dtype=np.int64))


def positions_of(state, prec) -> np.ndarray:
    if prec == "mixed":
        return state["posq"][:, :3].astype(np.float64) + state["posq_corr"][:, :3].astype(np.float64)
    return state["posq"][:, :3].astype(np.float64)


# ----------------------------------------------------------------------------------------------
# ctypes: our restatement
# ----------------------------------------------------------------------------------------------
class _System(C.Structure):
    _fields_ = [
        ("num_atoms", C.c_int), ("padded_num_atoms", C.c_int), ("num_molecules", C.c_int),
        ("velm", C.c_void_p), ("posq", C.c_void_p), ("posq_corr", C.c_void_p), ("force", C.c_void_p),
        ("pos_delta", C.c_void_p),
        ("force_extra", C.c_void_p), ("old_delta", C.c_void_p), ("com_velm", C.c_void_p), ("v_buffer", C.c_void_p),
        ("num_drude_pairs", C.c_int), ("drude_pairs", C.c_void_p),
        ("num_particles_nh", C.c_int), ("particles_nh", C.c_void_p),
        ("num_molecules_nh", C.c_int), ("molecules_nh", C.c_void_p),
        ("num_normal_nh", C.c_int), ("normal_nh", C.c_void_p),
        ("num_pairs_nh", C.c_int), ("pairs_nh", C.c_void_p),
        ("particle_mol_id", C.c_void_p), ("particles_in_molecules", C.c_void_p),
        ("particles_sorted_by_mol_id", C.c_void_p),
        ("num_tg", C.c_int), ("use_com_tg", C.c_int),
        ("num_chains", C.c_int), ("loops_per_step", C.c_int),
        ("eta", (C.c_double * MAX_CHAINS) * 3), ("eta_dot", (C.c_double * (MAX_CHAINS + 1)) * 3),
        ("eta_dotdot", (C.c_double * MAX_CHAINS) * 3), ("eta_mass", (C.c_double * MAX_CHAINS) * 3),
        ("tg_nkbt", C.c_double * 3), ("ke2", C.c_double * 3), ("vscale", C.c_double * 3),
        ("num_particles_ld", C.c_int),
        ("num_normal_ld", C.c_int), ("normal_ld", C.c_void_p),
        ("num_pairs_ld", C.c_int), ("pairs_ld", C.c_void_p),
        ("random", C.c_void_p), ("random_size", C.c_uint), ("random_index", C.c_uint),
        ("num_images", C.c_int), ("image_pairs", C.c_void_p),
        ("num_electrolyte", C.c_int), ("particles_electrolyte", C.c_void_p),
        ("dt", C.c_double), ("temperature", C.c_double), ("drude_temperature", C.c_double),
        ("friction", C.c_double), ("drude_friction", C.c_double), ("max_drude_distance", C.c_double),
        ("mirror", C.c_double), ("efield", C.c_double), ("cos_accel", C.c_double),
        ("box", C.c_double * 3), ("inv_mass_total", C.c_double),
        ("use_middle", C.c_int),
        ("force_mode", C.c_int), ("site", C.c_void_p), ("k_tether", C.c_double), ("k_drude", C.c_double),
        ("forces_valid", C.c_int), ("num_threads", C.c_int),
        ("num_shake", C.c_int), ("shake_atoms", C.c_void_p), ("shake_params", C.c_void_p), ("constraint_tolerance", C.c_double),
        ("num_settle", C.c_int), ("settle_atoms", C.c_void_p), ("settle_params", C.c_void_p),
        ("shake_mode", C.c_int),
        ("num_general", C.c_int), ("general_atoms", C.c_void_p), ("general_params", C.c_void_p), ("general_omega", C.c_double),
        ("num_vsites", C.c_int), ("vsite_atoms", C.c_void_p), ("vsite_params", C.c_void_p),
    ]


_LIBS: Dict[Tuple[str, str], C.CDLL] = {}


def lib(prec: str, fmad=False) -> C.CDLL:
    """fmad: True = the build with contracted multiply-adds (`make -C oracle fmad`, oracle/Makefile: OPT_FMAD) -- NVRTC's default for the
    reference's kernels; "altprelude" = SQRT / RECIP of `mixed` values in double (`make altprelude`, mixed mode).  Only
    tests/test_fmad_gap.py asks for either."""
    suffix = "_altprelude" if fmad == "altprelude" else ("_fmad" if fmad else "")
    key = ("oracle" + suffix, prec)
    if key not in _LIBS:
        path = os.path.join(HERE, f"liboracle_{prec}{suffix}.so")
        if not os.path.exists(path):
            build("altprelude" if fmad == "altprelude" else ("fmad" if fmad else "all"))
        L = C.CDLL(path)
        assert L.vvo_sizeof_system() == C.sizeof(_System), "ctypes _System out of sync with vv_oracle.h"
        L.vvo_calc_viscosity.restype = C.c_double
        _LIBS[key] = L
    return _LIBS[key]


def ref_lib(prec: str, fmad: bool = False) -> C.CDLL:
    key = ("ref" + ("_fmad" if fmad else ""), prec)
    if key not in _LIBS:
        _LIBS[key] = C.CDLL(os.path.join(HERE, "_ref", f"libvvref_{prec}{'_fmad' if fmad else ''}.so"))
    return _LIBS[key]


def have_fmad() -> bool:
    """liboracle_*_fmad.so present (or buildable) and this CPU has FMA instructions."""
    try:
        if " fma " not in open("/proc/cpuinfo").read().replace("\n", " "):
            return False
    except OSError:
        return False
    return all(os.path.exists(os.path.join(HERE, f"liboracle_{p}_fmad.so")) for p in PRECISIONS)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ct(prec):
    cr = C.c_float if REAL[prec] == np.float32 else C.c_double
    cm = C.c_float if MIXED[prec] == np.float32 else C.c_double
    return cr, cm


class OracleSystem:
    """One system + integrator configuration living in host memory, stepped by vv_oracle.c."""

    def __init__(self, spec, params: Params, prec: str = "mixed", random: np.ndarray | None = None,
                 force_mode: int = 1, k_tether: float = 1000.0, k_drude: float = 209200.0, num_threads: int = 1,
                 shake_mode: int | None = None, fmad: bool = False):
        """shake_mode: 0 = Gauss-Seidel sweeps over a constraint cluster, 1 = the cluster's constraints at once (vv_oracle.c:
        vvo_cluster_*); None follows VVHIP_SHAKE_MODE like the product (default 1).  fmad: the contracted build (lib())."""
        self.prec, self.spec = prec, spec
        self.L = lib(prec, fmad)
        R, M = REAL[prec], MIXED[prec]
        self.t = t = build_tables(spec, params)
        self.params = p = t["params"]
        n = spec.num_atoms
        self.state = make_state(spec, prec)
        self.site = self.state["posq"].copy()
        self.pos_delta = np.zeros((n, 4), dtype=M)
        self.old_delta = np.zeros((n, 4), dtype=M)
        self.force_extra = np.zeros((n, 3), dtype=R)
        self.com_velm = np.zeros((max(spec.num_molecules, 1), 4), dtype=M)
        self.v_buffer = np.zeros(max(n, 1), dtype=M)
        self.random = np.ascontiguousarray(random, dtype=np.float32) if random is not None else np.zeros((4, 4), np.float32)
        self.drude_pairs = np.ascontiguousarray(spec.drude_pairs, dtype=np.int32).reshape(-1, 2)
        s = self.s = _System()
        s.num_atoms, s.padded_num_atoms, s.num_molecules = n, padded(n), spec.num_molecules
        s.velm, s.posq, s.force = _p(self.state["velm"]), _p(self.state["posq"]), _p(self.state["force"])
        s.posq_corr = _p(self.state["posq_corr"]) if prec == "mixed" else None
        s.pos_delta, s.force_extra, s.old_delta = _p(self.pos_delta), _p(self.force_extra), _p(self.old_delta)
        s.com_velm, s.v_buffer = _p(self.com_velm), _p(self.v_buffer)
        s.num_drude_pairs, s.drude_pairs = len(self.drude_pairs), _p(self.drude_pairs)
        for name, cnt in (("particles_nh", "num_particles_nh"), ("molecules_nh", "num_molecules_nh"),
                          ("normal_nh", "num_normal_nh"), ("pairs_nh", "num_pairs_nh"), ("normal_ld", "num_normal_ld"),
                          ("pairs_ld", "num_pairs_ld"), ("image_pairs", "num_images"),
                          ("particles_electrolyte", "num_electrolyte")):
            arr = np.ascontiguousarray(t[name], dtype=np.int32)
            t[name] = arr
            setattr(s, cnt, arr.shape[0])
            setattr(s, name, _p(arr))
        for name in ("particle_mol_id", "particles_in_molecules", "particles_sorted_by_mol_id"):
            t[name] = np.ascontiguousarray(t[name], dtype=np.int32)
            setattr(s, name, _p(t[name]))
        s.num_particles_ld = t["num_particles_ld"]
        s.num_tg, s.use_com_tg = t["num_tg"], int(p.use_com_temp_group)
        s.num_chains, s.loops_per_step = p.num_chains, p.loops_per_step
        for i in range(3):
            for k in range(MAX_CHAINS):
                s.eta_mass[i][k] = t["eta_mass"][i, k]
            s.tg_nkbt[i] = t["nkbt"][i]
        s.random, s.random_size, s.random_index = _p(self.random), self.random.shape[0], 0
        s.dt, s.temperature, s.drude_temperature = p.step_size, p.temperature, p.drude_temperature
        s.friction, s.drude_friction, s.max_drude_distance = p.friction, p.drude_friction, p.max_drude_distance
        s.mirror, s.efield, s.cos_accel = p.mirror_location, p.electric_field, p.cos_acceleration
        for i in range(3):
            s.box[i] = float(spec.box[i])
        s.inv_mass_total = t["inv_mass_total"]
        s.use_middle = int(p.use_middle_scheme)
        s.force_mode, s.site, s.k_tether, s.k_drude = force_mode, _p(self.site), k_tether, k_drude
        s.forces_valid, s.num_threads = 0, num_threads
        self.general = None
        try:
            self.clusters = build_constraint_clusters(spec)
        except OracleError:                      # neither rigid triangles nor hydrogen-type clusters: the general solver takes ALL constraints
            self.clusters = None
            self.general = build_general_constraints(spec)
            s.num_general, s.general_atoms, s.general_params = len(self.general[0]), _p(self.general[0]), _p(self.general[1])
            s.general_omega = self.general_omega = general_relaxation(spec)
        vsites = list(getattr(spec, "virtual_sites", None) or [])
        if vsites:                               # (site, kind, parents, parameters) as SystemSpec.virtual_sites
            self.vsite_atoms = np.full((len(vsites), 5), -1, dtype=np.int32)
            self.vsite_params = np.zeros((len(vsites), 12), dtype=np.float64)
            for i, (site, kind, parents, prm) in enumerate(vsites):
                self.vsite_atoms[i, 0], self.vsite_atoms[i, 1] = site, kind
                self.vsite_atoms[i, 2:2 + len(parents)] = parents
                self.vsite_params[i, :len(prm)] = prm
            s.num_vsites, s.vsite_atoms, s.vsite_params = len(vsites), _p(self.vsite_atoms), _p(self.vsite_params)
        s.constraint_tolerance = 1e-5
        s.shake_mode = int(os.environ.get("VVHIP_SHAKE_MODE", "1")) if shake_mode is None else int(shake_mode)
        if self.clusters is not None:
            cl = self.clusters
            if len(cl["shake_atoms"]):
                s.num_shake, s.shake_atoms, s.shake_params = len(cl["shake_atoms"]), _p(cl["shake_atoms"]), _p(cl["shake_params"])
            if len(cl["settle_atoms"]):
                s.num_settle, s.settle_atoms, s.settle_params = len(cl["settle_atoms"]), _p(cl["settle_atoms"]), _p(cl["settle_params"])

    def step(self, n: int = 1):
        self.L.vvo_step(C.byref(self.s), n)

    def tether_force(self):
        self.L.vvo_tether_force(C.byref(self.s))

    @property
    def velm(self): return self.state["velm"]
    @property
    def posq(self): return self.state["posq"]
    @property
    def posq_corr(self): return self.state["posq_corr"]
    @property
    def force(self): return self.state["force"]

    def positions(self): return positions_of(self.state, self.prec)
    def ke2(self): return np.array(list(self.s.ke2))
    def vscale(self): return np.array(list(self.s.vscale))

    def chain_state(self):
        nc = self.params.num_chains
        return dict(eta=np.array([list(self.s.eta[i])[:nc] for i in range(3)]),
                    eta_dot=np.array([list(self.s.eta_dot[i])[:nc + 1] for i in range(3)]),
                    eta_dotdot=np.array([list(self.s.eta_dotdot[i])[:nc] for i in range(3)]))

    def viscosity(self):
        vmax = C.c_double()
        inv = self.L.vvo_calc_viscosity(C.byref(self.s), C.byref(vmax))
        return vmax.value, inv


def propagate_nh_chain(eta, eta_dot, eta_dotdot, eta_mass, ke2, ke2_target, t_target, step_size, loops_per_step=1):
    """C restatement of VVIntegrator::propagateNHChain; arrays are modified in place; returns the scale factor."""
    L = lib("double")
    nc = len(eta)
    assert len(eta_dot) == nc + 1
    f = C.c_double()
    L.vvo_propagate_nh_chain(C.c_int(nc), C.c_int(loops_per_step), C.c_double(step_size), _p(eta), _p(eta_dot),
                             _p(eta_dotdot), _p(eta_mass), C.c_double(ke2), C.c_double(ke2_target),
                             C.c_double(t_target), C.byref(f))
    return f.value


def propagate_nh_chain_py(eta, eta_dot, eta_dotdot, eta_mass, ke2, ke2_target, t_target, step_size, loops_per_step=1):
    """Independent pure-Python statement of openmmapi/src/VVIntegrator.cpp:340-376 (cross-check of the C one)."""
    from math import exp
    nc = len(eta)
    dt2 = step_size / loops_per_step / 2
    dt4 = dt2 / 2
    dt8 = dt4 / 2
    factor = 1.0
    expfac = 1.0
    eta_dotdot[0] = (ke2 - ke2_target) / eta_mass[0]
    for _ in range(loops_per_step):
        for ich in range(nc - 1, -1, -1):
            expfac = exp(-dt8 * eta_dot[ich + 1])
            eta_dot[ich] *= expfac
            eta_dot[ich] += eta_dotdot[ich] * dt4
            eta_dot[ich] *= expfac
        factor *= exp(-dt2 * eta_dot[0])
        for ich in range(nc):
            eta[ich] += dt2 * eta_dot[ich]
        eta_dotdot[0] = (ke2 * factor * factor - ke2_target) / eta_mass[0]
        eta_dot[0] *= expfac
        eta_dot[0] += eta_dotdot[0] * dt4
        eta_dot[0] *= expfac
        for ich in range(1, nc):
            expfac = exp(-dt8 * eta_dot[ich + 1])
            eta_dot[ich] *= expfac
            eta_dotdot[ich] = (eta_mass[ich - 1] * eta_dot[ich - 1] * eta_dot[ich - 1] - BOLTZ * t_target) / eta_mass[ich]
            eta_dot[ich] += eta_dotdot[ich] * dt4
            eta_dot[ich] *= expfac
    return factor


# ----------------------------------------------------------------------------------------------
# kernel-level calls, same signature for our restatement ("oracle") and the reference build ("ref")
# ----------------------------------------------------------------------------------------------
class _RefSizes(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("num_atoms", "padded_num_atoms", "num_drude_pairs", "num_particles_nh",
                                        "num_molecules_nh", "num_normal_particles_nh", "num_pairs_nh",
                                        "num_normal_particles_ld", "num_pairs_ld", "num_images",
                                        "num_particles_electrolyte")]


class Kernels:
    """Uniform kernel-level interface.  which='oracle' -> vv_oracle.c; which='ref' -> reference .cu on CPU.

    Every method takes numpy arrays in the reference's device layouts and updates them in place."""

    def __init__(self, which: str, prec: str, fmad: bool = False):
        self.which, self.prec = which, prec
        self.L = lib(prec, fmad) if which == "oracle" else ref_lib(prec, fmad)
        self.cr, self.cm = _ct(prec)
        self.M = MIXED[prec]
        self.R = REAL[prec]

    # -- helpers for the reference build ------------------------------------------------------
    def _sizes(self, **kw):
        s = _RefSizes()
        for k, v in kw.items():
            setattr(s, k, int(v))
        self.L.vvref_set_sizes(C.byref(s))

    def _dt2(self, dt):
        return np.array([0.0, dt], dtype=self.M)      # `mixed2 dt`: .y is the step size

    def _corr(self, corr):
        return _p(corr) if self.prec == "mixed" else None

    # -- middle scheme ------------------------------------------------------------------------
    def middle_vel(self, velm, force, force_extra, dt):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_integrate_middle_vel(n, force.shape[0] // 3, _p(velm), _p(force), _p(force_extra), self.cm(dt))
        else:
            self._sizes(num_atoms=n, padded_num_atoms=force.shape[0] // 3)
            self.L.integrateMiddleVel(_p(velm), _p(force), _p(force_extra), _p(self._dt2(dt)))

    def middle_pos1(self, velm, pos_delta, old_delta, dt):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_integrate_middle_pos1(n, _p(velm), _p(pos_delta), _p(old_delta), self.cm(dt))
        else:
            self._sizes(num_atoms=n)
            self.L.integrateMiddlePos1(_p(velm), _p(pos_delta), _p(old_delta), _p(self._dt2(dt)))

    def middle_pos2(self, velm, pos_delta, old_delta, dt):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_integrate_middle_pos2(n, _p(velm), _p(pos_delta), _p(old_delta), self.cm(dt))
        else:
            self._sizes(num_atoms=n)
            self.L.integrateMiddlePos2(_p(velm), _p(pos_delta), _p(old_delta), _p(self._dt2(dt)))

    def middle_pos3(self, posq, corr, pos_delta, old_delta, velm, dt):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_integrate_middle_pos3(n, _p(posq), self._corr(corr), _p(pos_delta), _p(old_delta), _p(velm), self.cm(dt))
        else:
            self._sizes(num_atoms=n)
            self.L.integrateMiddlePos3(_p(posq), self._corr(corr), _p(pos_delta), _p(old_delta), _p(velm), _p(self._dt2(dt)))

    def hard_wall(self, posq, corr, velm, drude_pairs, dt, max_dist, hw_scale, vv_module=False):
        npairs = drude_pairs.shape[0]
        if self.which == "oracle":
            self.L.vvo_apply_hard_wall(npairs, _p(posq), self._corr(corr), _p(velm), _p(drude_pairs), self.cm(dt),
                                       self.cm(max_dist), self.cm(hw_scale))
        else:
            self._sizes(num_drude_pairs=npairs)
            fn = self.L.vv_applyHardWallConstraints if vv_module else self.L.applyHardWallConstraints
            fn(_p(posq), self._corr(corr), _p(velm), _p(drude_pairs), _p(self._dt2(dt)), self.cm(max_dist), self.cm(hw_scale))

    def reset_extra_force(self, force_extra):
        n = force_extra.shape[0]
        if self.which == "oracle":
            self.L.vvo_reset_extra_force(n, _p(force_extra))
        else:
            self._sizes(num_atoms=n)
            self.L.resetExtraForce(_p(force_extra))

    # -- classic velocity Verlet --------------------------------------------------------------
    def vv_vel(self, velm, force, force_extra, pos_delta, dt, fscale, update_pos_delta):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_vv_integrate_velocities(n, force.shape[0] // 3, _p(velm), _p(force), _p(force_extra), _p(pos_delta),
                                               self.cm(dt), self.cm(fscale), int(update_pos_delta))
        else:
            self._sizes(num_atoms=n, padded_num_atoms=force.shape[0] // 3)
            self.L.velocityVerletIntegrateVelocities(_p(velm), _p(force), _p(force_extra), _p(pos_delta),
                                                     _p(self._dt2(dt)), self.cm(fscale), C.c_bool(bool(update_pos_delta)))

    def vv_pos(self, posq, corr, pos_delta, velm, dt):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_vv_integrate_positions(n, _p(posq), self._corr(corr), _p(pos_delta), _p(velm), self.cm(dt))
        else:
            self._sizes(num_atoms=n)
            self.L.velocityVerletIntegratePositions(_p(posq), self._corr(corr), _p(pos_delta), _p(velm), _p(self._dt2(dt)))

    # -- Nose-Hoover --------------------------------------------------------------------------
    def _nh_sizes(self, t):
        self._sizes(num_particles_nh=len(t["particles_nh"]), num_molecules_nh=len(t["molecules_nh"]),
                    num_normal_particles_nh=len(t["normal_nh"]), num_pairs_nh=len(t["pairs_nh"]))

    def calc_com(self, velm, com_velm, t):
        if self.which == "oracle":
            self.L.vvo_calc_com_velocities(len(t["molecules_nh"]), _p(velm), _p(com_velm), _p(t["particles_in_molecules"]),
                                           _p(t["particles_sorted_by_mol_id"]), _p(t["molecules_nh"]))
        else:
            self._nh_sizes(t)
            getattr(self.L, f"calcCOMVelocities_tg{t['num_tg']}")(
                _p(velm), _p(com_velm), _p(t["particles_in_molecules"]), _p(t["particles_sorted_by_mol_id"]), _p(t["molecules_nh"]))

    def normalize(self, velm, com_velm, t):
        if self.which == "oracle":
            self.L.vvo_normalize_velocities(len(t["particles_nh"]), _p(velm), _p(com_velm), _p(t["particle_mol_id"]), _p(t["particles_nh"]))
        else:
            self._nh_sizes(t)
            getattr(self.L, f"normalizeVelocities_tg{t['num_tg']}")(_p(velm), _p(com_velm), _p(t["particle_mol_id"]), _p(t["particles_nh"]))

    def kinetic_energies(self, velm, com_velm, t):
        """compute + sum; returns 2*KE per temperature group as an array of `mixed` [num_tg]."""
        ntg = t["num_tg"]
        out = np.zeros(3, dtype=self.M)
        if self.which == "oracle":
            self.L.vvo_compute_kinetic_energies(ntg, len(t["normal_nh"]), len(t["molecules_nh"]), len(t["pairs_nh"]), _p(velm),
                                                _p(com_velm), _p(t["normal_nh"]), _p(t["pairs_nh"]), _p(t["molecules_nh"]), _p(out))
        else:
            self._nh_sizes(t)
            bufsize = max(len(t["particles_nh"]), 1) * ntg
            buf = np.zeros(bufsize, dtype=self.M)        # the harness zeroes the buffer (quirk Q3)
            getattr(self.L, f"computeNormalizedKineticEnergies_tg{ntg}")(
                _p(velm), _p(com_velm), _p(t["normal_nh"]), _p(t["pairs_nh"]), _p(buf), _p(t["molecules_nh"]), C.c_int(bufsize))
            getattr(self.L, f"sumNormalizedKineticEnergies_tg{ntg}")(_p(buf), _p(out), C.c_int(bufsize))
        return out[:ntg].copy()

    def scale_velocity(self, velm, com_velm, t, vscale3):
        vs = np.ascontiguousarray(vscale3, dtype=self.M)
        assert vs.shape[0] == 3
        if self.which == "oracle":
            self.L.vvo_scale_velocity(len(t["normal_nh"]), len(t["pairs_nh"]), _p(velm), _p(com_velm), _p(t["particle_mol_id"]),
                                      _p(t["normal_nh"]), _p(t["pairs_nh"]), _p(vs))
        else:
            self._nh_sizes(t)
            getattr(self.L, f"scaleVelocity_tg{t['num_tg']}")(_p(velm), _p(com_velm), _p(t["particle_mol_id"]),
                                                               _p(t["normal_nh"]), _p(t["pairs_nh"]), _p(vs))

    # -- cos acceleration ---------------------------------------------------------------------
    def _ibox(self, inv_box_z):
        class R4(C.Structure):
            _fields_ = [("x", self.cr), ("y", self.cr), ("z", self.cr), ("w", self.cr)]
        return R4(0, 0, inv_box_z, 0)

    def add_cos_acceleration(self, posq, velm, force_extra, accel, inv_box_z):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_add_cos_acceleration(n, _p(posq), _p(velm), _p(force_extra), self.cr(accel), self.cr(inv_box_z))
        else:
            self._sizes(num_atoms=n)
            self.L.addCosAcceleration(_p(posq), _p(velm), _p(force_extra), self.cr(accel), self._ibox(inv_box_z))

    def calc_bias(self, posq, velm, vbuf, inv_box_z, inv_mass_total):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_calc_periodic_velocity_bias(n, _p(posq), _p(velm), _p(vbuf), self.cr(inv_box_z))
            self.L.vvo_sum_v(n, _p(vbuf), C.c_double(inv_mass_total))
        else:
            self._sizes(num_atoms=n)
            self.L.calcPeriodicVelocityBias(_p(posq), _p(velm), _p(vbuf), self._ibox(inv_box_z))
            self.L.sumV(_p(vbuf), C.c_double(inv_mass_total), C.c_int(n))

    def remove_bias(self, posq, velm, vbuf, inv_box_z):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_remove_periodic_velocity_bias(n, _p(posq), _p(velm), _p(vbuf), self.cr(inv_box_z))
        else:
            self._sizes(num_atoms=n)
            self.L.removePeriodicVelocityBias(_p(posq), _p(velm), _p(vbuf), self._ibox(inv_box_z))

    def restore_bias(self, posq, velm, vbuf, inv_box_z):
        n = velm.shape[0]
        if self.which == "oracle":
            self.L.vvo_restore_periodic_velocity_bias(n, _p(posq), _p(velm), _p(vbuf), self.cr(inv_box_z))
        else:
            self._sizes(num_atoms=n)
            self.L.restorePeriodicVelocityBias(_p(posq), _p(velm), _p(vbuf), self._ibox(inv_box_z))

    # -- Langevin / E-field / images ----------------------------------------------------------
    def langevin(self, velm, force_extra, normal, pairs, drag, randf, drag_d, randf_d, random, random_index):
        if self.which == "oracle":
            self.L.vvo_add_extra_force_drude_langevin(len(normal), len(pairs), _p(velm), _p(force_extra), _p(normal), _p(pairs),
                                                      self.cm(drag), self.cm(randf), self.cm(drag_d), self.cm(randf_d),
                                                      _p(random), C.c_uint(random_index))
        else:
            self._sizes(num_normal_particles_ld=len(normal), num_pairs_ld=len(pairs))
            self.L.addExtraForceDrudeLangevin(_p(velm), _p(force_extra), _p(normal), _p(pairs), self.cm(drag), self.cm(randf),
                                              self.cm(drag_d), self.cm(randf_d), _p(random), C.c_uint(random_index))

    def electric_field(self, posq, force_extra, particles, efscale):
        if self.which == "oracle":
            self.L.vvo_add_extra_force_electric_field(len(particles), _p(posq), _p(force_extra), _p(particles), self.cr(efscale))
        else:
            self._sizes(num_particles_electrolyte=len(particles))
            self.L.addExtraForceElectricField(_p(posq), _p(force_extra), _p(particles), self.cr(efscale))

    def update_images(self, posq, corr, image_pairs, mirror):
        if self.which == "oracle":
            self.L.vvo_update_image_positions(len(image_pairs), _p(posq), _p(corr), _p(image_pairs), self.cm(mirror))
        else:
            self._sizes(num_images=len(image_pairs))
            self.L.updateImagePositions(_p(posq), _p(corr), _p(image_pairs), self.cm(mirror))


# ----------------------------------------------------------------------------------------------
# the reference's kernels on the GPU (oracle/_ref/libvvref_gpu_mixed_tg{1,3}.so, built by `make refgpu`)
# ----------------------------------------------------------------------------------------------
def have_ref_gpu() -> bool:
    return all(os.path.exists(os.path.join(HERE, "_ref", f"libvvref_gpu_mixed_tg{t}.so")) for t in (1, 3))


class RefGpuSystem:
    """The reference's own kernel sequence (middle scheme, mixed precision) on the GPU: see ref_gpu_driver.cpp.
    Covers NH / TGNH + hard wall + cos acceleration (BASELINE C2, C3, C4); no Langevin / image kernels."""

    def __init__(self, spec, params: Params, k_tether: float = 1000.0, k_drude: float = 209200.0):
        t = build_tables(spec, params)
        p = t["params"]
        if t["num_tg"] == 2 or not p.use_middle_scheme or len(spec.particles_ld) or len(spec.image_pairs):
            raise ValueError("RefGpuSystem covers the middle scheme with 1 or 3 temperature groups, no Langevin / images")
        self.L = C.CDLL(os.path.join(HERE, "_ref", f"libvvref_gpu_mixed_tg{t['num_tg']}.so"))
        self.L.vvrefgpu_create.restype = C.c_void_p
        st = make_state(spec, "mixed")
        self.n = n = spec.num_atoms
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self._keep = [st["velm"], st["posq"], st["posq_corr"], i32(spec.drude_pairs).reshape(-1, 2), i32(t["particles_nh"]), i32(t["molecules_nh"]),
                      i32(t["normal_nh"]), i32(t["pairs_nh"]).reshape(-1, 2), i32(t["particle_mol_id"]), i32(t["particles_in_molecules"]),
                      i32(t["particles_sorted_by_mol_id"]), np.ascontiguousarray(t["eta_mass"], dtype=np.float64),
                      np.ascontiguousarray(t["nkbt"], dtype=np.float64), np.ascontiguousarray(spec.box, dtype=np.float64)]
        k = self._keep
        d = C.c_double
        self.h = C.c_void_p(self.L.vvrefgpu_create(
            n, padded(n), spec.num_molecules, _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), len(k[3]), _p(k[4]), len(k[4]), _p(k[5]), len(k[5]),
            _p(k[6]), len(k[6]), _p(k[7]), len(k[7]), _p(k[8]), _p(k[9]), _p(k[10]), int(t["num_tg"]), int(p.use_com_temp_group),
            int(p.num_chains), int(p.loops_per_step), _p(k[11]), _p(k[12]), d(p.step_size), d(p.temperature), d(p.drude_temperature),
            d(p.max_drude_distance), d(p.cos_acceleration), d(t["inv_mass_total"]), _p(k[13]), d(k_tether), d(k_drude)))

    def step(self, n):
        self.L.vvrefgpu_step(self.h, int(n))

    def sync(self):
        self.L.vvrefgpu_sync(self.h)

    def download(self):
        velm = np.zeros((self.n, 4), np.float64)
        posq = np.zeros((self.n, 4), np.float32)
        corr = np.zeros((self.n, 4), np.float32)
        ke2, vs = (C.c_double * 3)(), (C.c_double * 3)()
        self.L.vvrefgpu_download(self.h, _p(velm), _p(posq), _p(corr), ke2, vs)
        return dict(velm=velm, posq=posq, posq_corr=corr, ke2=np.array(list(ke2)), vscale=np.array(list(vs)),
                    positions=posq[:, :3].astype(np.float64) + corr[:, :3].astype(np.float64))

    def close(self):
        if self.h:
            self.L.vvrefgpu_destroy(self.h)
            self.h = None
