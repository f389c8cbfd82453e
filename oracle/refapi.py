"""oracle/refapi.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of the C shim (oracle/ref_api_shim.cpp) around a VVIntegrator host class:
  * ``load("ref")``  -> oracle/_ref/libvvref_api.so : the REFERENCE's openmmapi/src/VVIntegrator.cpp compiled in place
                        (build container only; `make -C oracle refapi`)
  * ``load("ours")`` -> oracle/libvvours_api.so     : this repository's openmmapi/src/VVIntegrator.cpp behind the same shim
Both against the stand-in OpenMM headers of compat/.  Used by tests/test_ref_api.py and oracle/make_golden_refapi.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATHS = {"ref": os.path.join(HERE, "_ref", "libvvref_api.so"), "ours": os.path.join(HERE, "libvvours_api.so")}
_LIBS = {}


def available(which: str) -> bool:
    if which == "ours" and not os.path.exists(PATHS["ours"]):
        subprocess.run(["make", "-s", "-C", HERE, "ourapi"], check=True)
    return os.path.exists(PATHS[which])


def load(which: str) -> C.CDLL:
    if which not in _LIBS:
        if not available(which):
            raise FileNotFoundError(PATHS[which])
        L = C.CDLL(PATHS[which])
        L.vvref_api_create.restype = C.c_void_p
        L.vvref_api_destroy.argtypes = [C.c_void_p]
        L.vvref_api_destroy.restype = None
        _LIBS[which] = L
    return _LIBS[which]


def _ip(a):
    return a.ctypes.data_as(C.c_void_p) if a.size else None


def propagate(which, eta, eta_dot, eta_dotdot, eta_mass, ke2, ke2_target, t_target, step_size, loops_per_step=1):
    """VVIntegrator::propagateNHChain of the chosen build; arrays (float64) are modified in place; returns the factor."""
    L = load(which)
    f = C.c_double()
    rc = L.vvref_api_propagate(C.c_double(step_size), C.c_int(loops_per_step), C.c_int(len(eta)), _ip(eta), _ip(eta_dot), _ip(eta_dotdot),
                               _ip(eta_mass), C.c_double(ke2), C.c_double(ke2_target), C.c_double(t_target), C.byref(f))
    assert rc == 0
    return f.value


class Api:
    """A System + VVIntegrator + Context on the mock platform; initialize() has run when the constructor returns."""

    def __init__(self, which, masses, mol_id, num_molecules, drude_pairs=(), constraints=(), cmm=True, particles_ld=(), image_pairs=(),
                 electrolyte=(), cos=0.0, middle=True, use_com=-1, friction=-1.0):
        self.L = load(which)
        m = np.ascontiguousarray(masses, dtype=np.float64)
        mol = np.ascontiguousarray(mol_id, dtype=np.int32)
        dr = np.ascontiguousarray(drude_pairs, dtype=np.int32).reshape(-1)
        cn = np.ascontiguousarray(constraints, dtype=np.int32).reshape(-1)
        ld = np.ascontiguousarray(particles_ld, dtype=np.int32).reshape(-1)
        im = np.ascontiguousarray(image_pairs, dtype=np.int32).reshape(-1)
        el = np.ascontiguousarray(electrolyte, dtype=np.int32).reshape(-1)
        err = C.create_string_buffer(512)
        self.n, self.nmol = m.size, int(num_molecules)
        self.h = self.L.vvref_api_create(C.c_int(m.size), _ip(m), C.c_int(self.nmol), _ip(mol), C.c_int(dr.size // 2), _ip(dr),
                                         C.c_int(cn.size // 2), _ip(cn), C.c_int(int(cmm)), C.c_int(ld.size), _ip(ld), C.c_int(im.size // 2), _ip(im),
                                         C.c_int(el.size), _ip(el), C.c_double(cos), C.c_int(int(middle)), C.c_int(int(use_com)), C.c_double(friction),
                                         err, C.c_int(512))
        self.error = err.value.decode() if not self.h else ""

    def tables(self):
        n, nmol = self.n, self.nmol
        pnh, mnh = np.zeros(n, np.int32), np.zeros(max(nmol, 1), np.int32)
        pmol, ld, img = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        inv = np.zeros(max(nmol, 1))
        n_nh, n_mnh, use_com, fr = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        got = self.L.vvref_api_tables(C.c_void_p(self.h), _ip(pnh), C.byref(n_nh), _ip(mnh), C.byref(n_mnh), _ip(pmol), _ip(inv), C.byref(use_com),
                                      C.byref(fr), _ip(ld), _ip(img))
        assert got == nmol
        return dict(particles_nh=pnh[:n_nh.value].copy(), molecules_nh=mnh[:n_mnh.value].copy(), particle_mol_id=pmol, mol_inv_mass=inv[:nmol].copy(),
                    use_com=bool(use_com.value), friction=fr.value, is_ld=ld.astype(bool), is_image=img.astype(bool))

    def trace(self, steps, query_energy_after=False):
        out = C.create_string_buffer(1 << 16)
        rc = self.L.vvref_api_trace(C.c_void_p(self.h), C.c_int(steps), C.c_int(int(query_energy_after)), out, C.c_int(1 << 16))
        assert rc >= 0, rc
        return out.value.decode().split(",")

    def close(self):
        if self.h:
            self.L.vvref_api_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
