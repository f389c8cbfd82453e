"""ctypes binding of lib/libvvhip.so (C ABI: include/vvhip.h).  Pure plumbing: no arithmetic here.

The library is looked up in-tree (``openmm-velocityverlet_amd/lib/libvvhip.so``, built by
``__graft_entry__.build()`` / ``make -C openmm-velocityverlet_amd/csrc``).  If it is missing the
import of this module fails loudly -- there is no Python or CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# VVHIP_LIB: another build of the same library (A/B runs of two builds on one GPU box, tools/probes); default: the in-tree one
LIB_PATH = os.environ.get("VVHIP_LIB") or os.path.join(HERE, "lib", "libvvhip.so")

MAX_CHAINS = 8
SINGLE, MIXED, DOUBLE = 0, 1, 2
PRECISION = {"single": SINGLE, "mixed": MIXED, "double": DOUBLE}
REAL = {"single": np.float32, "mixed": np.float32, "double": np.float64}
MIXED_T = {"single": np.float32, "mixed": np.float64, "double": np.float64}
OK, ERR_INVALID, ERR_TOPOLOGY, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_EXCHANGE, ERR_OVERFLOW, ERR_RENDEZVOUS, ERR_CONSTRAINT = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9

# stage bits of csrc/vv_args.hpp (only the test hooks need them)
A_FE_LOAD, A_FE_STORE, A_LD, A_EF, A_COS, A_KICK_FULL, A_KICK_HALF, A_POSDELTA_VV, A_POS1, A_BIAS, A_KE, A_UNBIAS_ACC, A_COMPART, A_CZ_STORE, A_CZ_LOAD = \
    [1 << i for i in range(15)]
B_SCALE, B_UNBIAS, B_BIAS_REMOVE, B_BIAS_RESTORE, B_DRIFT_MIDDLE, B_POS2, B_POS3, B_VV_KICK, B_VV_POS, B_HARDWALL, B_IMAGE, B_CHAIN, B_CZ_LOAD = \
    [1 << i for i in range(13)]
A_MTAB, B_MTAB = 1 << 18, 1 << 16        # static mass tables: added by the library itself (vv_api.cpp run_a / run_b)
C_CHAIN, C_BIAS = 1, 2


class VVHipError(RuntimeError):
    """Stands in for OpenMMException on the Python side; `.code` is the vvhip error code."""

    def __init__(self, code, message):
        super().__init__(f"[vvhip {code}] {message}")
        self.code = code
        self.message = message


class SystemDesc(C.Structure):
    _fields_ = [("num_atoms", C.c_int32), ("padded_num_atoms", C.c_int32), ("masses", C.c_void_p), ("mol_id", C.c_void_p),
                ("num_molecules", C.c_int32), ("num_drude_pairs", C.c_int32), ("drude_pairs", C.c_void_p),
                ("num_constraints", C.c_int32), ("constraints", C.c_void_p), ("has_cm_motion_remover", C.c_int32),
                ("num_particles_ld", C.c_int32), ("particles_ld", C.c_void_p),
                ("num_image_pairs", C.c_int32), ("image_pairs", C.c_void_p),
                ("num_electrolyte", C.c_int32), ("particles_electrolyte", C.c_void_p),
                ("shard_begin", C.c_int32), ("shard_end", C.c_int32), ("constraint_distances", C.c_void_p),
                ("num_virtual_sites", C.c_int32), ("virtual_sites", C.c_void_p), ("virtual_site_params", C.c_void_p)]


VSITE_AVERAGE2, VSITE_AVERAGE3, VSITE_OUT_OF_PLANE, VSITE_LOCAL_COORDS = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [("temperature", C.c_double), ("frequency", C.c_double), ("drude_temperature", C.c_double),
                ("drude_frequency", C.c_double), ("step_size", C.c_double),
                ("num_nh_chains", C.c_int32), ("loops_per_step", C.c_int32),
                ("max_drude_distance", C.c_double), ("friction", C.c_double), ("drude_friction", C.c_double),
                ("mirror_location", C.c_double), ("electric_field", C.c_double), ("cos_acceleration", C.c_double),
                ("use_com_temp_group", C.c_int32), ("use_middle_scheme", C.c_int32),
                ("auto_set_com_temp_group", C.c_int32), ("auto_set_friction", C.c_int32), ("constraint_tolerance", C.c_double)]


class Buffers(C.Structure):
    _fields_ = [("velm", C.c_void_p), ("posq", C.c_void_p), ("posq_correction", C.c_void_p), ("force", C.c_void_p),
                ("pos_delta", C.c_void_p), ("random", C.c_void_p), ("random_size", C.c_uint32), ("stream", C.c_void_p)]


class PlanInfo(C.Structure):
    _fields_ = [("num_particles_nh", C.c_int32), ("num_molecules_nh", C.c_int32), ("num_normal_nh", C.c_int32),
                ("num_pairs_nh", C.c_int32), ("num_normal_ld", C.c_int32), ("num_pairs_ld", C.c_int32),
                ("num_images", C.c_int32), ("num_electrolyte", C.c_int32), ("num_temp_groups", C.c_int32),
                ("use_com_temp_group", C.c_int32), ("friction", C.c_double),
                ("dof", C.c_double * 3), ("nkbt", C.c_double * 3), ("eta_mass", (C.c_double * MAX_CHAINS) * 3),
                ("inv_mass_total", C.c_double), ("num_waves", C.c_int32), ("num_slots_used", C.c_int32),
                ("max_cluster", C.c_int32), ("num_shake_clusters", C.c_int32), ("constraints_fused", C.c_int32),
                ("num_settle_clusters", C.c_int32), ("periodic_layout", C.c_int32), ("num_general_constraints", C.c_int32),
                ("num_virtual_sites", C.c_int32), ("general_relaxation", C.c_double)]


class NHState(C.Structure):
    _fields_ = [("eta", (C.c_double * MAX_CHAINS) * 3), ("eta_dot", (C.c_double * (MAX_CHAINS + 1)) * 3),
                ("eta_dotdot", (C.c_double * MAX_CHAINS) * 3), ("ke2", C.c_double * 3), ("vscale", C.c_double * 3),
                ("v_bias", C.c_double)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the hot path is HIP only; there is no fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, dbl = C.c_void_p, C.c_int32, C.c_uint32, C.c_double
    P = C.POINTER
    sig = {
        "vvhip_plan_create": [P(SystemDesc), P(Params), C.c_int, P(vp), C.c_char_p, C.c_size_t],
        "vvhip_plan_get_info": [vp, P(PlanInfo)],
        "vvhip_plan_get_slots": [vp, vp, i32],
        "vvhip_bind": [vp, P(Buffers)],
        "vvhip_set_params": [vp, P(Params)],
        "vvhip_set_box": [vp, P(dbl * 3)],
        "vvhip_get_nh_state": [vp, P(NHState)],
        "vvhip_set_nh_state": [vp, P(NHState)],
        "vvhip_step_middle": [vp, u32],
        "vvhip_step_vv_first": [vp],
        "vvhip_step_vv_second": [vp, u32],
        "vvhip_step_middle_phases": [vp],
        "vvhip_algorithmic_bytes": [vp, P(i32), P(i32)],
        "vvhip_step_middle_phase": [vp, C.c_int, u32],
        "vvhip_accumulators": [vp, C.c_int, P(vp), P(i32)],
        "vvhip_reset_extra_force": [vp], "vvhip_middle_kick": [vp], "vvhip_middle_half_drift1": [vp],
        "vvhip_middle_half_drift2": [vp], "vvhip_middle_finish": [vp],
        "vvhip_vv_half_kick": [vp, C.c_int], "vvhip_vv_positions": [vp], "vvhip_scale_velocity": [vp],
        "vvhip_apply_langevin_force": [vp, u32], "vvhip_update_image_positions": [vp],
        "vvhip_apply_electric_force": [vp], "vvhip_apply_cosine_force": [vp], "vvhip_calc_velocity_bias": [vp],
        "vvhip_remove_velocity_bias": [vp], "vvhip_restore_velocity_bias": [vp],
        "vvhip_calc_viscosity": [vp, P(dbl), P(dbl)], "vvhip_compute_kinetic_energy": [vp, P(dbl)], "vvhip_force_extra": [vp, P(vp)],
        "vvhip_device_count": [P(C.c_int)], "vvhip_set_device": [C.c_int],
        "vvhip_malloc": [P(vp), C.c_size_t], "vvhip_free": [vp],
        "vvhip_memcpy_h2d": [vp, vp, C.c_size_t], "vvhip_memcpy_d2h": [vp, vp, C.c_size_t],
        "vvhip_memset": [vp, C.c_int, C.c_size_t], "vvhip_synchronize": [vp],
        "vvhip_stream_create": [P(vp)], "vvhip_stream_destroy": [vp],
        "vvhip_synth_tether_force": [vp, vp, dbl, dbl],
        "vvhip_run_graph": [vp, C.c_int, C.c_int, vp, dbl, dbl],
        "vvhip_graph_prepare": [vp, C.c_int, vp, dbl, dbl],
        "vvhip_status": [vp, P(i32), P(i32)], "vvhip_status_clear": [vp], "vvhip_recovery_count": [vp, P(C.c_int64)], "vvhip_masses_changed": [vp],
        "vvhip_status_words": [vp, P(i32 * 4)], "vvhip_fused_status": [vp, P(i32), P(C.c_int64), P(i32)],
        "vvhip_run_eager": [vp, C.c_int, vp, dbl, dbl],
        "vvhip_run_eager_unfused": [vp, C.c_int, vp, dbl, dbl],
        "vvhip_set_random_seed": [vp, C.c_uint64], "vvhip_fill_random": [vp],
        "vvhip_comm_count": [vp, P(i32)], "vvhip_peer_access": [C.c_int, C.c_int, P(i32)],
        "vvhip_comm_unique_id": [vp], "vvhip_comm_init": [vp, vp, C.c_int, C.c_int], "vvhip_comm_destroy": [vp],
        "vvhip_mailbox_create": [vp, C.c_int, C.c_int, vp], "vvhip_mailbox_connect": [vp, vp],
        "vvhip_mailbox_status": [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)], "vvhip_mailbox_destroy": [vp],
        "vvhip_mailbox_layout": [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
        "vvhip_time_kernel": [vp, C.c_int, u32, C.c_int, P(dbl)],
        "vvhip_generic_launches": [vp, P(C.c_int64 * 2), P(u32 * 2)],
        "vvhip_rtc_stats": [P(C.c_int64 * 3), P(C.c_double)], "vvhip_rtc_failures": [P(C.c_int64)],
        "vvhip_rtc_mode": [C.c_int],
        "vvhip_timing_enable": [vp, C.c_int], "vvhip_timing_read": [vp, P(dbl), P(dbl), P(dbl), P(i32 * 3)],
        "vvhip_debug_step_spans": [vp, C.c_int, vp, dbl, dbl, P(dbl * 36)], "vvhip_debug_fused_flags": [vp, C.c_int, P(u32)], "vvhip_debug_launch_shape": [vp, P(C.c_int32 * 4)],
        "vvhip_debug_launch": [vp, C.c_int, u32, u32], "vvhip_debug_tune": [vp, C.c_char_p, C.c_int],
        "vvhip_debug_read_accumulators": [vp, P(dbl * 4), C.c_int],
        "vvhip_debug_set_scales": [vp, P(dbl * 4)],
        "vvhip_debug_old_delta": [vp, P(vp)],
        "vvhip_set_trace": [vp, C.c_int],
        "vvhip_debug_span": [vp, C.c_int, C.c_uint32, C.c_int, P(C.c_double * 8)],
        "vvhip_debug_timestamps": [vp, C.c_uint32, C.c_int, P(C.c_longlong * 128)],
        "vvhip_debug_timestamps_fused": [vp, C.c_int, P(C.c_longlong * 128)],
    }
    for name, args in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # an OLDER build of the library named by VVHIP_LIB (A/B runs of a previous round's kernels, tools/probes/ab_lib_rates.sh) may lack the newest
            # entry points; the product library must export every one of them (tests/test_host_plan.py checks the header's list)
            if os.environ.get("VVHIP_LIB"):
                continue
            raise
        fn.argtypes = args
        fn.restype = C.c_int
    lib.vvhip_plan_destroy.argtypes = [vp]
    lib.vvhip_plan_destroy.restype = None
    lib.vvhip_last_error.argtypes = [vp]
    lib.vvhip_last_error.restype = C.c_char_p
    lib.vvhip_plan_unfused_reason.argtypes = [vp]
    lib.vvhip_plan_unfused_reason.restype = C.c_char_p
    return lib


lib = _load()
EXPORTS = sorted(["vvhip_plan_destroy", "vvhip_last_error"] + [n for n in dir(lib) if n.startswith("vvhip_")])


def device_count() -> int:
    n = C.c_int(0)
    lib.vvhip_device_count(C.byref(n))
    return n.value


def check(rc: int, plan=None, what: str = ""):
    if rc != OK:
        msg = lib.vvhip_last_error(plan).decode() if plan else what
        raise VVHipError(rc, msg or what)


class DeviceArray:
    """A device allocation with a host-side dtype/shape; upload/download are blocking copies."""

    def __init__(self, shape, dtype, fill=None):
        self.shape = tuple(np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        rc = lib.vvhip_malloc(C.byref(p), max(self.nbytes, 16))
        if rc != OK:
            raise VVHipError(rc, f"hipMalloc of {self.nbytes} bytes failed (is a GPU visible?)")
        self.ptr = p.value
        if fill is not None:
            self.upload(np.full(self.shape, fill, dtype=self.dtype))

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        if self.nbytes:
            rc = lib.vvhip_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes)
            if rc != OK:
                raise VVHipError(rc, "hipMemcpy H2D failed")

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            rc = lib.vvhip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes)
            if rc != OK:
                raise VVHipError(rc, "hipMemcpy D2H failed")
        return out

    def free(self):
        if self.ptr:
            lib.vvhip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
