"""oracle/refhost.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of oracle/_ref/libvvref_host_<precision>.so (`make -C oracle refhost`, build container only): the REFERENCE's whole step
on the CPU -- its VVIntegrator.cpp, its CudaVVKernels.cpp / CudaVVKernelFactory.cpp and its kernels compiled in place against the
stand-in CUDA-platform headers of oracle/refhost/ (oracle/ref_host_shim.cpp says what that executes and what it does not).
Used by tests/test_ref_host.py and oracle/make_golden_refhost.py.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .oracle import MIXED, REAL, Params, make_state, padded, positions_of

HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def path(prec: str, fmad: bool = False) -> str:
    """fmad: the build with contracted multiply-adds (`make -C oracle reffmad`, oracle/Makefile: OPT_FMAD)."""
    return os.path.join(HERE, "_ref", f"libvvref_host_{prec}{'_fmad' if fmad else ''}.so")


def available(prec: str = "mixed", fmad: bool = False) -> bool:
    return os.path.exists(path(prec, fmad))


def load(prec: str, fmad: bool = False) -> C.CDLL:
    key = prec + ("_fmad" if fmad else "")
    prec_ = prec
    prec = key
    if prec not in _LIBS:
        L = C.CDLL(path(prec_, fmad))
        L.vvrh_create.restype = C.c_void_p
        L.vvrh_time.restype = C.c_double
        L.vvrh_time.argtypes = [C.c_void_p]
        L.vvrh_destroy.restype = None
        L.vvrh_destroy.argtypes = [C.c_void_p]
        _LIBS[prec] = L
    return _LIBS[prec]


def _ip(a):
    return a.ctypes.data_as(C.c_void_p) if a.size else None


class RefHost:
    """System + VVIntegrator + Context on the stand-in CUDA platform, for one SystemSpec / Params; initialize() has run on return.

    Forces are static (whatever ``force`` holds, int64 fixed point, 3 x padded atoms), normals are the injected ``random`` buffer."""

    def __init__(self, spec, params: Params, prec: str = "mixed", random: np.ndarray | None = None, force: np.ndarray | None = None,
                 fmad: bool = False):
        self.L, self.prec, self.spec = load(prec, fmad), prec, spec
        p = params
        m = np.ascontiguousarray(spec.masses, dtype=np.float64)
        mol = np.ascontiguousarray(spec.mol_id, dtype=np.int32)
        dr = np.ascontiguousarray(spec.drude_pairs, dtype=np.int32).reshape(-1)
        cn = np.ascontiguousarray(spec.constraints, dtype=np.int32).reshape(-1)
        ld = np.ascontiguousarray(list(spec.particles_ld), dtype=np.int32).reshape(-1)
        im = np.ascontiguousarray(spec.image_pairs, dtype=np.int32).reshape(-1)
        el = np.ascontiguousarray(list(spec.particles_electrolyte), dtype=np.int32).reshape(-1)
        par = np.array([p.temperature, p.frequency, p.drude_temperature, p.drude_frequency, p.step_size, p.max_drude_distance,
                        -1.0 if p.auto_set_friction else p.friction, -1.0 if p.drude_friction == Params().drude_friction else p.drude_friction, p.mirror_location, p.electric_field,
                        p.cos_acceleration], dtype=np.float64)
        ipar = np.array([p.num_chains, p.loops_per_step, int(p.use_middle_scheme), -1 if p.auto_set_com_temp_group else int(p.use_com_temp_group),
                         int(spec.has_cm_motion_remover), int(prec == "double"), int(prec == "mixed")], dtype=np.int32)
        box = np.ascontiguousarray(spec.box, dtype=np.float64)
        err = C.create_string_buffer(512)
        self.n = m.size
        self.h = self.L.vvrh_create(C.c_int(m.size), _ip(m), C.c_int(int(spec.num_molecules)), _ip(mol), C.c_int(dr.size // 2), _ip(dr),
                                    C.c_int(cn.size // 2), _ip(cn), C.c_int(ld.size), _ip(ld), C.c_int(im.size // 2), _ip(im), C.c_int(el.size), _ip(el),
                                    _ip(par), _ip(ipar), _ip(box), err, C.c_int(512))
        self.error = err.value.decode() if not self.h else ""
        if not self.h:
            return
        self.state = make_state(spec, prec)
        if force is not None:
            self.state["force"][:] = force
        self._up(0, self.state["velm"]); self._up(1, self.state["posq"]); self._up(3, self.state["force"])
        if prec == "mixed":
            self._up(2, self.state["posq_corr"])
        if random is not None:
            r = np.ascontiguousarray(random, dtype=np.float32).reshape(-1, 4)
            self.L.vvrh_upload(C.c_void_p(self.h), C.c_int(4), _ip(r), C.c_longlong(r.shape[0]))

    def _up(self, which, a):
        self.L.vvrh_upload(C.c_void_p(self.h), C.c_int(which), _ip(a), C.c_longlong(a.shape[0]))

    def step(self, n: int = 1):
        err = C.create_string_buffer(512)
        if self.L.vvrh_step(C.c_void_p(self.h), C.c_int(n), err, C.c_int(512)) != 0:
            raise RuntimeError(err.value.decode())
        self.L.vvrh_download(C.c_void_p(self.h), C.c_int(0), _ip(self.state["velm"]))
        self.L.vvrh_download(C.c_void_p(self.h), C.c_int(1), _ip(self.state["posq"]))
        if self.prec == "mixed":
            self.L.vvrh_download(C.c_void_p(self.h), C.c_int(2), _ip(self.state["posq_corr"]))

    def set(self, which: str, value: float):
        """A parameter change between steps through the reference's own setters (cos_acceleration, step_size, temperature) / the box edge."""
        code = {"cos_acceleration": 0, "step_size": 1, "temperature": 2, "box": 3}[which]
        assert self.L.vvrh_set(C.c_void_p(self.h), C.c_int(code), C.c_double(float(value))) == 0

    @property
    def velm(self): return self.state["velm"]
    @property
    def posq(self): return self.state["posq"]
    def positions(self): return positions_of(self.state, self.prec)

    def thermostat(self):
        """DOF, N*kB*T, chain masses / state and table sizes as CudaModifyDrudeNoseKernel holds them; None without an NH kernel."""
        dof, nkbt, ke2, vs = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)
        em, eta, ed = np.zeros((3, 16)), np.zeros((3, 16)), np.zeros((3, 16))
        cnt = np.zeros(4, np.int32)
        ntg = self.L.vvrh_thermostat(C.c_void_p(self.h), _ip(dof), _ip(nkbt), _ip(em), _ip(eta), _ip(ed), _ip(ke2), _ip(vs), _ip(cnt))
        if ntg == 0:
            return None
        return dict(num_tg=ntg, dof=dof, nkbt=nkbt, eta_mass=em, eta=eta, eta_dot=ed, ke2=ke2, vscale=vs,
                    num_particles_nh=int(cnt[0]), num_molecules_nh=int(cnt[1]), num_normal_nh=int(cnt[2]), num_pairs_nh=int(cnt[3]))

    def launches(self):
        buf = C.create_string_buffer(1 << 16)
        n = self.L.vvrh_launches(C.c_void_p(self.h), buf, C.c_int(1 << 16))
        assert n >= 0
        return [s for s in buf.value.decode().split(",") if s]

    def viscosity(self):
        a, b = C.c_double(), C.c_double()
        self.L.vvrh_viscosity(C.c_void_p(self.h), C.byref(a), C.byref(b))
        return a.value, b.value

    def time(self):
        return self.L.vvrh_time(C.c_void_p(self.h))

    def close(self):
        if self.h:
            self.L.vvrh_destroy(C.c_void_p(self.h))
            self.h = None
