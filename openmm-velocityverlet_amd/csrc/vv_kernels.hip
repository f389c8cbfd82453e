// vv_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// Work-item layout.  One work-item per particle; a 64-lane wave holds whole "clusters": every Drude
// pair, and (with the COM temperature group) every molecule, sits inside ONE wave (vv_host.cpp).
// Hence
//   * the Drude partner's velocity/position comes from a cross-lane shuffle, not a second gather;
//   * the molecular centre-of-mass velocity is a segmented wave scan, not the reference's serial
//     per-molecule read-modify-write in global memory (K/drudeNoseHoover.cu:15-25);
//   * per-group kinetic energies are reduced wave -> LDS -> one 64-bit fixed-point atomic per block,
//     so the global sums do not depend on arrival order (bit-reproducible) and need no second
//     "single block" pass (K/drudeNoseHoover.cu:121-151) nor a host round trip (HOST:709-746).
// Global loads are 16/32-byte per lane and, for contiguous molecules, fully coalesced.
//
// Arithmetic.  Each expression keeps the operand types and association of the reference kernel it
// replaces (cited per stage; K/ = platforms/cuda/src/kernels/) and the library is built with
// -ffp-contract=off, so element-wise results equal the CPU oracle bit for bit; only the reductions
// differ (their order, and the reciprocals that feed nothing but sums: Prec::RECIP_SUM).  No MFMA:
// there is no contraction on this path (HBM / latency / fp64-issue bound).
#include "vv_kernels.hpp"

#include <hip/hip_ext.h>

#include "vv_host.hpp"

namespace vv {

template <class T> struct Vec;
template <> struct Vec<float> { using v4 = float4; using v3 = float3; };
template <> struct Vec<double> { using v4 = double4; using v3 = double3; };

// What OpenMM's context prepends to the reference kernels (assumption stated in oracle/ref_prelude.h):
// single-precision sqrt/recip literals unless the `real` type is double.
template <class real> struct Prec;
template <> struct Prec<float> {
    template <class T> static __device__ __forceinline__ float SQRT(T x) { return sqrtf((float) x); }
    static __device__ __forceinline__ float RECIP(float x) { return 1.0f / x; }
    static __device__ __forceinline__ double RECIP(double x) { return 1.0f / x; }
    // Reciprocal for quantities that only feed REDUCTIONS (kinetic-energy sums, molecular COM): hardware estimate (24 bits) + two Newton
    // steps, within 1 ulp of the IEEE quotient (equal to it in 2^20 random samples, tools/probes/rcp_probe.cpp).  Those sums already depend on the summation order at that level, so nothing that is compared
    // bit for bit with the oracle (kick, drift, hard wall, scaling arithmetic) goes through this; an IEEE fp64 division
    // costs 14 instructions, this 5, on a kernel that is issue-bound at the headline size.
    static __device__ __forceinline__ float RECIP_SUM(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ double RECIP_SUM(double x) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); return fma(fma(-x, r, 1.0), r, r); }
};
template <> struct Prec<double> {
    static __device__ __forceinline__ double SQRT(double x) { return sqrt(x); }
    static __device__ __forceinline__ double RECIP(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double RECIP_SUM(double x) { double r = __builtin_amdgcn_rcp(x); r = fma(fma(-x, r, 1.0), r, r); return fma(fma(-x, r, 1.0), r, r); }
};

template <class V>
__device__ __forceinline__ void store_vec(V* base, int index, const V& val) { base[index] = val; }

__device__ __forceinline__ float shfl(float x, int src) { return __shfl(x, src, 64); }
__device__ __forceinline__ double shfl(double x, int src) { return __shfl(x, src, 64); }

// ---- wave-wide inclusive prefix sum on the DPP network (no LDS round trips): Hillis-Steele inside each 16-lane
// row (row_shr 1,2,4,8; lanes without a source add 0), then row 0 -> row 1 / row 2 -> row 3 (row_bcast:15) and
// rows 0+1 -> rows 2,3 (row_bcast:31).  Lane 63 ends up with the wave total.
// ROW_MASK 0xF with bound_ctrl writes EVERY lane (lanes without a source read 0), so the destination needs no initial value:
// __builtin_amdgcn_mov_dpp.  The two row_bcast steps write only some rows; there the unwritten lanes must read 0, which costs a
// zeroing move per 32-bit half (update_dpp with old = 0).  Using update_dpp everywhere, as round 1 did, put those moves into all
// 6 steps of every scan (24 instructions per 64-bit scan instead of 20).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_fetch32(int v) {
    if (ROW_MASK == 0xF) return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double x) {
    const int lo = dpp_fetch32<CTRL, ROW_MASK>(__double2loint(x));
    const int hi = dpp_fetch32<CTRL, ROW_MASK>(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fetch(float x) {
    return __int_as_float(dpp_fetch32<CTRL, ROW_MASK>(__float_as_int(x)));
}

// Arithmetic layout (vv_host.hpp: PeriodicLayout): everything a lane used to read from its slot words, from the wave index alone.
struct PeriodicWave {
    int atom0, count, region, seg0;
};
__device__ __forceinline__ PeriodicWave periodic_wave(const PeriodicArgs& q, int wave_uniform) {
    const int w = __builtin_amdgcn_readfirstlane(wave_uniform);
    const int cell = (int) __umulhi((unsigned) w, q.magic);                 // = w / wpc (checked for every wave by analyze()); 0 for one cell
    const int wl = w - cell * q.wpc;
    int ws = q.d_wave[0], as = q.d_atom_start[0], ae = q.d_atom_end[0], ss = q.d_seg[0], P = q.d_P[0], spw = q.d_spw[0], region = 0;
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const bool in = wl >= q.wave_start[k];
        region += in ? 1 : 0;
        ws += in ? q.d_wave[k] : 0; as += in ? q.d_atom_start[k] : 0; ae += in ? q.d_atom_end[k] : 0;
        ss += in ? q.d_seg[k] : 0; P += in ? q.d_P[k] : 0; spw += in ? q.d_spw[k] : 0;
    }
    const int wr = wl - ws, a0 = as + wr * P;
    PeriodicWave r;
    r.atom0 = cell * q.apc + a0;
    r.count = min(P, ae - a0);
    r.region = region;
    r.seg0 = cell * q.spc + ss + wr * spw;
    return r;
}

// number of set bits of a 64-lane mask below the calling lane
__device__ __forceinline__ unsigned lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) mask, 0u));
}

template <class T>
__device__ __forceinline__ T wave_scan(T x) {
    x += dpp_fetch<0x111, 0xF>(x);   // row_shr:1
    x += dpp_fetch<0x112, 0xF>(x);   // row_shr:2
    x += dpp_fetch<0x114, 0xF>(x);   // row_shr:4
    x += dpp_fetch<0x118, 0xF>(x);   // row_shr:8
    x += dpp_fetch<0x142, 0xA>(x);   // row_bcast:15 into rows 1 and 3
    x += dpp_fetch<0x143, 0xC>(x);   // row_bcast:31 into rows 2 and 3
    return x;
}
// Three independent scans, step by step side by side: the compiler keeps the order of the source, and a DPP read needs two wait
// states after the VALU write of its source -- alone, every step of a scan is followed by an s_nop and a dependent add; three
// interleaved scans fill each other's gaps.
template <class T>
__device__ __forceinline__ void wave_scan3(T& x, T& y, T& z) {
#define VV_SCAN_STEP(CTRL, MASK) { const T dx_ = dpp_fetch<CTRL, MASK>(x), dy_ = dpp_fetch<CTRL, MASK>(y), dz_ = dpp_fetch<CTRL, MASK>(z); x += dx_; y += dy_; z += dz_; }
    VV_SCAN_STEP(0x111, 0xF) VV_SCAN_STEP(0x112, 0xF) VV_SCAN_STEP(0x114, 0xF) VV_SCAN_STEP(0x118, 0xF) VV_SCAN_STEP(0x142, 0xA) VV_SCAN_STEP(0x143, 0xC)
#undef VV_SCAN_STEP
}
// total of a wave in lane 63
__device__ __forceinline__ double wave_sum(double x) { return wave_scan(x); }

// Total of the contiguous lane segment [first, last] for every lane of it: P[last] - P[first-1] on the wave prefix sum.
// (The prefix runs over at most 64 particles, so the subtraction costs < 3 bits; the reference itself sums serially.)
template <class T>
__device__ __forceinline__ T segment_total(T x, int lane, int first, int last) {
    const T P = wave_scan(x);
    const T hi = shfl(P, last);
    const T lo = shfl(P, first > 0 ? first - 1 : 0);
    return first > 0 ? hi - lo : hi;
}

// Block partials -> fixed-point atomics.  vals[k] is the calling thread's contribution.
template <int NV>
__device__ __forceinline__ void block_accumulate(const double (&vals)[NV], const bool (&enabled)[NV],
                                                 unsigned long long* acc, const double* scale, unsigned int* status, bool exclusive, unsigned long long acc_old) {
    __shared__ double red[16][NV];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // the three kinetic-energy sums side by side (wave_scan3), the rest one by one
    if (NV >= 3 && enabled[0] && enabled[1] && enabled[2]) {
        double s0 = vals[0], s1 = vals[1], s2 = vals[2];
        wave_scan3(s0, s1, s2);
        if (lane == 63) { red[w][0] = s0; red[w][1] = s1; red[w][2] = s2; }
    }
#pragma unroll
    for (int k = 0; k < NV; k++) {
        if (!enabled[k] || (k < 3 && enabled[0] && enabled[1] && enabled[2])) continue;           // compile-time in the specialised kernels
        double s = wave_sum(vals[k]);
        if (lane == 63) red[w][k] = s;
    }
    __syncthreads();
    // Thread k finishes quantity k.  Nothing here may index `enabled` or `scale` by the thread id: both live in constant / kernel
    // argument memory, and a lane-indexed read of those turns into a global load in front of the atomic -- two dependent memory
    // round trips at the very end of every block (seen in the ISA; ~0.5 us of kernel A's tail).  Bit mask and select chain instead.
    unsigned mask = 0;
#pragma unroll
    for (int k = 0; k < NV; k++) mask |= enabled[k] ? (1u << k) : 0u;
    if (threadIdx.x < NV && ((mask >> threadIdx.x) & 1u)) {
        double s = 0;
        for (int i = 0; i < nw; i++) s += red[i][threadIdx.x];
        double sc = 0;
#pragma unroll
        for (int k = 0; k < NV; k++)
            if (enabled[k] && (int) threadIdx.x == k) sc = scale[k];
        const double scaled = s * sc;
        // every block stays below 2^62 / blocks, so the int64 total over all blocks (and ranks <= 16 with 1024x headroom in the scale)
        // cannot wrap unnoticed; a block beyond that (or a NaN) raises the sticky flag in host memory instead of feeding the
        // thermostat garbage (vvhip_synchronize / the run loops return VVHIP_ERR_OVERFLOW)
        if (__builtin_expect(!(fabs(scaled) * (double) gridDim.x < 4611686018427387904.0), 0) && status)
            __hip_atomic_store(&status[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long q = __double2ll_rn(scaled);
        // `exclusive`: the launch has at most ACC_SLOTS blocks, so this block is the only one that touches its slot: the slot's old
        // content (requested at the kernel's top) plus q goes back as a plain store.  An agent-scope atomic of this multi-XCD part is
        // executed at the memory side, and the kernel cannot end before it has come back.
        if (exclusive) { if (q != 0) acc[threadIdx.x * ACC_SLOTS + (blockIdx.x & (ACC_SLOTS - 1))] = acc_old + (unsigned long long) q; }
        else if (q != 0) atomicAdd(&acc[threadIdx.x * ACC_SLOTS + (blockIdx.x & (ACC_SLOTS - 1))], (unsigned long long) q);
    }
}

// Sum of the ACC_SLOTS slots of quantity k (exact integer sum), computed cooperatively by one wave (DPP prefix sum,
// total taken from lane 63).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_fetch_i64(long long x) {
    const int lo = dpp_fetch32<CTRL, ROW_MASK>((int) (x & 0xFFFFFFFFll));
    const int hi = dpp_fetch32<CTRL, ROW_MASK>((int) (x >> 32));
    return ((long long) hi << 32) | (unsigned int) lo;
}
__device__ __forceinline__ long long acc_reduce_lane_sum(long long s);
// ... of values the caller has already loaded (kernel B's thermostat wave issues all its loads first)
__device__ __forceinline__ long long acc_reduce(const long long (&raw)[ACC_SLOTS / 64]) {
    long long s = 0;
#pragma unroll
    for (int j = 0; j < ACC_SLOTS / 64; j++) s += raw[j];
    return acc_reduce_lane_sum(s);
}
__device__ __forceinline__ long long acc_total(const unsigned long long* acc, int k, int lane) {
    long long s = 0;
#pragma unroll
    for (int j = 0; j < ACC_SLOTS / 64; j++) s += (long long) acc[k * ACC_SLOTS + lane + 64 * j];
    return acc_reduce_lane_sum(s);
}
__device__ __forceinline__ long long acc_reduce_lane_sum(long long s) {
    s += dpp_fetch_i64<0x111, 0xF>(s);
    s += dpp_fetch_i64<0x112, 0xF>(s);
    s += dpp_fetch_i64<0x114, 0xF>(s);
    s += dpp_fetch_i64<0x118, 0xF>(s);
    s += dpp_fetch_i64<0x142, 0xA>(s);
    s += dpp_fetch_i64<0x143, 0xC>(s);
    return __shfl(s, 63, 64);
}

template <class real>
__device__ __forceinline__ double cos_kz(real z, real inv_box_z) {
    return cos(2 * 3.1415926 * z * inv_box_z);   // literal pi and double cosine in every mode (quirk Q2)
}

// positions are posq (+ posqCorrection in mixed mode): K/middle.cu:81-96
template <class real, class mixed>
struct PosIO {
    using real4 = typename Vec<real>::v4;
    static constexpr bool kMixed = sizeof(real) != sizeof(mixed);
    // K/middle.cu:81-96: positions are posq (+ posqCorrection in mixed mode)
    static __device__ __forceinline__ void load(const void* posq, const void* corr, int i, mixed& x, mixed& y, mixed& z, mixed& w, real& zraw) {
        const real4 p1 = ((const real4*) posq)[i];
        zraw = p1.z;
        if (kMixed) {
            const real4 p2 = ((const real4*) corr)[i];
            x = p1.x + (mixed) p2.x; y = p1.y + (mixed) p2.y; z = p1.z + (mixed) p2.z; w = p1.w;
        } else {
            x = p1.x; y = p1.y; z = p1.z; w = p1.w;
        }
    }
    static __device__ __forceinline__ void store(void* posq, void* corr, int i, mixed x, mixed y, mixed z, mixed w) {
        real4 p = {(real) x, (real) y, (real) z, (real) w};
        store_vec((real4*) posq, i, p);
        if (kMixed) {
            real4 c = {(real) (x - (real) x), (real) (y - (real) y), (real) (z - (real) z), 0};
            store_vec((real4*) corr, i, c);
        }
    }
};

// ================================================================================ in-kernel SHAKE
// Constraint clusters of the kind OpenMM's SHAKE kernels take: one central particle, up to three peripheral particles of
// equal mass at equal distance.  A cluster always lies inside one wave (vv_host.cpp), so the peripherals hand their state
// to the central lane through a per-wave LDS page, the central lane iterates exactly like OpenMM's applyShakeToPositions /
// applyShakeToVelocities (Gauss-Seidel over the cluster, <= 15 sweeps, float cluster parameters, `mixed` arithmetic), and
// the peripherals pick their result up again.  OpenMM's source is not under /root/reference: this follows its published
// algorithm, parity with OpenMM itself is unpinned; the CPU oracle carries the same statement.
// Rigid three-site molecules (what OpenMM hands to SETTLE): analytic, no iteration.  Same unpinned status as the SHAKE clusters:
// OpenMM's source is not under /root/reference; the formulas below were checked numerically (|r_ij| = d_ij, r_ij.v_ij = 0 and
// conservation of the molecule's momentum to 1e-15) and the CPU oracle carries an independently written statement.
template <class mixed>
__device__ __forceinline__ void settle_positions_math(mixed m0, mixed m1, mixed distAB, mixed distBB,
        mixed apos0x, mixed apos0y, mixed apos0z, mixed apos1x, mixed apos1y, mixed apos1z, mixed apos2x, mixed apos2y, mixed apos2z,
        mixed& xp0x, mixed& xp0y, mixed& xp0z, mixed& xp1x, mixed& xp1y, mixed& xp1z, mixed& xp2x, mixed& xp2y, mixed& xp2z) {
        // Miyamoto & Kollman's SETTLE as OpenMM's settle kernel applies it to the step displacement: apos* old positions, xp* displacements,
        // particle 0 = apex (mass m0), particles 1, 2 = the two equal partners (mass m1 each); distAB apex-partner, distBB partner-partner.
        const mixed xb0 = apos1x - apos0x, yb0 = apos1y - apos0y, zb0 = apos1z - apos0z;
        const mixed xc0 = apos2x - apos0x, yc0 = apos2y - apos0y, zc0 = apos2z - apos0z;
        const mixed invTotalMass = 1 / (m0 + m1 + m1);
        const mixed xcom = (xp0x * m0 + (xb0 + xp1x) * m1 + (xc0 + xp2x) * m1) * invTotalMass;
        const mixed ycom = (xp0y * m0 + (yb0 + xp1y) * m1 + (yc0 + xp2y) * m1) * invTotalMass;
        const mixed zcom = (xp0z * m0 + (zb0 + xp1z) * m1 + (zc0 + xp2z) * m1) * invTotalMass;
        const mixed xa1 = xp0x - xcom, ya1 = xp0y - ycom, za1 = xp0z - zcom;
        const mixed xb1 = xb0 + xp1x - xcom, yb1 = yb0 + xp1y - ycom, zb1 = zb0 + xp1z - zcom;
        const mixed xc1 = xc0 + xp2x - xcom, yc1 = yc0 + xp2y - ycom, zc1 = zc0 + xp2z - zcom;
        const mixed xaksZd = yb0 * zc0 - zb0 * yc0, yaksZd = zb0 * xc0 - xb0 * zc0, zaksZd = xb0 * yc0 - yb0 * xc0;
        const mixed xaksXd = ya1 * zaksZd - za1 * yaksZd, yaksXd = za1 * xaksZd - xa1 * zaksZd, zaksXd = xa1 * yaksZd - ya1 * xaksZd;
        const mixed xaksYd = yaksZd * zaksXd - zaksZd * yaksXd, yaksYd = zaksZd * xaksXd - xaksZd * zaksXd, zaksYd = xaksZd * yaksXd - yaksZd * xaksXd;
        const mixed axlng = sqrt(xaksXd * xaksXd + yaksXd * yaksXd + zaksXd * zaksXd);
        const mixed aylng = sqrt(xaksYd * xaksYd + yaksYd * yaksYd + zaksYd * zaksYd);
        const mixed azlng = sqrt(xaksZd * xaksZd + yaksZd * yaksZd + zaksZd * zaksZd);
        const mixed trns11 = xaksXd / axlng, trns21 = yaksXd / axlng, trns31 = zaksXd / axlng;
        const mixed trns12 = xaksYd / aylng, trns22 = yaksYd / aylng, trns32 = zaksYd / aylng;
        const mixed trns13 = xaksZd / azlng, trns23 = yaksZd / azlng, trns33 = zaksZd / azlng;
        const mixed xb0d = trns11 * xb0 + trns21 * yb0 + trns31 * zb0, yb0d = trns12 * xb0 + trns22 * yb0 + trns32 * zb0;
        const mixed xc0d = trns11 * xc0 + trns21 * yc0 + trns31 * zc0, yc0d = trns12 * xc0 + trns22 * yc0 + trns32 * zc0;
        const mixed za1d = trns13 * xa1 + trns23 * ya1 + trns33 * za1;
        const mixed xb1d = trns11 * xb1 + trns21 * yb1 + trns31 * zb1, yb1d = trns12 * xb1 + trns22 * yb1 + trns32 * zb1, zb1d = trns13 * xb1 + trns23 * yb1 + trns33 * zb1;
        const mixed xc1d = trns11 * xc1 + trns21 * yc1 + trns31 * zc1, yc1d = trns12 * xc1 + trns22 * yc1 + trns32 * zc1, zc1d = trns13 * xc1 + trns23 * yc1 + trns33 * zc1;
        // step 1: A2'
        const mixed rc = 0.5f * distBB;
        mixed rb = sqrt(distAB * distAB - rc * rc);
        const mixed ra = rb * (m1 + m1) * invTotalMass;
        rb -= ra;
        const mixed sinphi = za1d / ra;
        const mixed cosphi = sqrt(1 - sinphi * sinphi);
        const mixed sinpsi = (zb1d - zc1d) / (2 * rc * cosphi);
        const mixed cospsi = sqrt(1 - sinpsi * sinpsi);
        const mixed ya2d = ra * cosphi;
        mixed xb2d = -rc * cospsi;
        const mixed yb2d = -rb * cosphi - rc * sinpsi * sinphi;
        const mixed yc2d = -rb * cosphi + rc * sinpsi * sinphi;
        const mixed xb2d2 = xb2d * xb2d;
        const mixed hh2 = 4.0f * xb2d2 + (yb2d - yc2d) * (yb2d - yc2d) + (zb1d - zc1d) * (zb1d - zc1d);
        const mixed deltx = 2.0f * xb2d + sqrt(4.0f * xb2d2 - hh2 + distBB * distBB);
        xb2d -= deltx * 0.5f;
        // step 2: A3'
        const mixed alpha = xb2d * (xb0d - xc0d) + yb0d * yb2d + yc0d * yc2d;
        const mixed beta = xb2d * (yc0d - yb0d) + xb0d * yb2d + xc0d * yc2d;
        const mixed gamma = xb0d * yb1d - xb1d * yb0d + xc0d * yc1d - xc1d * yc0d;
        const mixed al2be2 = alpha * alpha + beta * beta;
        const mixed sintheta = (alpha * gamma - beta * sqrt(al2be2 - gamma * gamma)) / al2be2;
        // step 3: A3
        const mixed costheta = sqrt(1 - sintheta * sintheta);
        const mixed xa3d = -ya2d * sintheta, ya3d = ya2d * costheta, za3d = za1d;
        const mixed xb3d = xb2d * costheta - yb2d * sintheta, yb3d = xb2d * sintheta + yb2d * costheta, zb3d = zb1d;
        const mixed xc3d = -xb2d * costheta - yc2d * sintheta, yc3d = -xb2d * sintheta + yc2d * costheta, zc3d = zc1d;
        // step 4: back to the lab frame
        const mixed xa3 = trns11 * xa3d + trns12 * ya3d + trns13 * za3d, ya3 = trns21 * xa3d + trns22 * ya3d + trns23 * za3d, za3 = trns31 * xa3d + trns32 * ya3d + trns33 * za3d;
        const mixed xb3 = trns11 * xb3d + trns12 * yb3d + trns13 * zb3d, yb3 = trns21 * xb3d + trns22 * yb3d + trns23 * zb3d, zb3 = trns31 * xb3d + trns32 * yb3d + trns33 * zb3d;
        const mixed xc3 = trns11 * xc3d + trns12 * yc3d + trns13 * zc3d, yc3 = trns21 * xc3d + trns22 * yc3d + trns23 * zc3d, zc3 = trns31 * xc3d + trns32 * yc3d + trns33 * zc3d;
        xp0x = xcom + xa3; xp0y = ycom + ya3; xp0z = zcom + za3;
        xp1x = xcom + xb3 - xb0; xp1y = ycom + yb3 - yb0; xp1z = zcom + zb3 - zb0;
        xp2x = xcom + xc3 - xc0; xp2y = ycom + yc3 - yc0; xp2z = zcom + zc3 - zc0;
}
template <class mixed>
__device__ __forceinline__ void settle_velocities_math(mixed m0, mixed m1,
        mixed apos0x, mixed apos0y, mixed apos0z, mixed apos1x, mixed apos1y, mixed apos1z, mixed apos2x, mixed apos2y, mixed apos2z,
        mixed& v0x, mixed& v0y, mixed& v0z, mixed& v1x, mixed& v1y, mixed& v1z, mixed& v2x, mixed& v2y, mixed& v2z) {
        // Velocity constraints of the rigid triangle: one multiplier per bond, d/dt |r_ij|^2 = 0 for the three bonds at once.
        // With e_ij the unit bond vectors and v_ij = (v_j - v_i).e_ij the system is linear 3x3 in (tab, tbc, tca) and is solved in closed
        // form; particle 0 = apex A (mass mA), 1 = B, 2 = C (mass mB = mC).
        mixed eABx = apos1x - apos0x, eABy = apos1y - apos0y, eABz = apos1z - apos0z;
        mixed eBCx = apos2x - apos1x, eBCy = apos2y - apos1y, eBCz = apos2z - apos1z;
        mixed eCAx = apos0x - apos2x, eCAy = apos0y - apos2y, eCAz = apos0z - apos2z;
        const mixed nAB = 1 / sqrt(eABx * eABx + eABy * eABy + eABz * eABz);
        const mixed nBC = 1 / sqrt(eBCx * eBCx + eBCy * eBCy + eBCz * eBCz);
        const mixed nCA = 1 / sqrt(eCAx * eCAx + eCAy * eCAy + eCAz * eCAz);
        eABx *= nAB; eABy *= nAB; eABz *= nAB; eBCx *= nBC; eBCy *= nBC; eBCz *= nBC; eCAx *= nCA; eCAy *= nCA; eCAz *= nCA;
        const mixed vAB = (v1x - v0x) * eABx + (v1y - v0y) * eABy + (v1z - v0z) * eABz;
        const mixed vBC = (v2x - v1x) * eBCx + (v2y - v1y) * eBCy + (v2z - v1z) * eBCz;
        const mixed vCA = (v0x - v2x) * eCAx + (v0y - v2y) * eCAy + (v0z - v2z) * eCAz;
        const mixed cA = -(eABx * eCAx + eABy * eCAy + eABz * eCAz);
        const mixed cB = -(eABx * eBCx + eABy * eBCy + eABz * eBCz);
        const mixed cC = -(eBCx * eCAx + eBCy * eCAy + eBCz * eCAz);
        const mixed s2A = 1 - cA * cA, s2B = 1 - cB * cB, s2C = 1 - cC * cC;
        const mixed mA = m0, mB = m1, mC = m1;
        const mixed mABCinv = 1 / (mA * mB * mC);
        const mixed denom = (((s2A * mB + s2B * mA) * mC + (s2A * mB * mB + 2 * (cA * cB * cC + 1) * mA * mB + s2B * mA * mA)) * mC + s2C * mA * mB * (mA + mB)) * mABCinv;
        const mixed tab = ((cB * cC * mA - cA * mB - cA * mC) * vCA + (cA * cC * mB - cB * mC - cB * mA) * vBC + (s2C * mA * mA * mB * mB * mABCinv + (mA + mB + mC)) * vAB) / denom;
        const mixed tbc = ((cA * cB * mC - cC * mB - cC * mA) * vCA + (s2A * mB * mB * mC * mC * mABCinv + (mA + mB + mC)) * vBC + (cA * cC * mB - cB * mA - cB * mC) * vAB) / denom;
        const mixed tca = ((s2B * mA * mA * mC * mC * mABCinv + (mA + mB + mC)) * vCA + (cA * cB * mC - cC * mB - cC * mA) * vBC + (cB * cC * mA - cA * mB - cA * mC) * vAB) / denom;
        const mixed iA = 1 / mA, iB = 1 / mB, iC = 1 / mC;
        v0x += (eABx * tab - eCAx * tca) * iA; v0y += (eABy * tab - eCAy * tca) * iA; v0z += (eABz * tab - eCAz * tca) * iA;
        v1x += (eBCx * tbc - eABx * tab) * iB; v1y += (eBCy * tbc - eABy * tab) * iB; v1z += (eBCz * tbc - eABz * tab) * iB;
        v2x += (eCAx * tca - eBCx * tbc) * iC; v2y += (eCAy * tca - eBCy * tbc) * iC; v2z += (eCAz * tca - eBCz * tbc) * iC;
}

// ---- constraint clusters: hand-over page and solvers
// Per-wave LDS page, component-major: page[c * 64 + lane], c = 0..2 position, 3..5 the vector being constrained (velocity or step
// displacement), 6 inverse mass (SETTLE only).  Lane-major rows of 7 values put the lanes of a wave on 32 banks two by two; this way a
// row of 64 lanes covers every bank once, for the writes and for the gathers by cluster lane alike.  A wave executes its LDS
// operations in order, so the "barriers" below are compiler fences only.
#define VV_WAVE_LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <class mixed>
__device__ __forceinline__ mixed dot3(const mixed (&a)[3], const mixed (&b)[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Cofactors and reciprocal determinant of a symmetric 3 x 3 matrix (oracle/vv_oracle.c: sym3_cofactors, same operations in the same order)
template <class mixed> struct Sym3Inv { mixed c00, c01, c02, c11, c12, c22, inv; };
template <class mixed>
__device__ __forceinline__ Sym3Inv<mixed> sym3_cofactors(mixed A00, mixed A01, mixed A02, mixed A11, mixed A12, mixed A22) {
    Sym3Inv<mixed> q;
    q.c00 = A11 * A22 - A12 * A12;
    q.c01 = A02 * A12 - A01 * A22;
    q.c02 = A01 * A12 - A02 * A11;
    q.c11 = A00 * A22 - A02 * A02;
    q.c12 = A01 * A02 - A00 * A12;
    q.c22 = A00 * A11 - A01 * A01;
    const mixed det = A00 * q.c00 + A01 * q.c01 + A02 * q.c02;
    q.inv = 1 / det;
    return q;
}
// What every lane of a hydrogen-type cluster needs: old bonds r_k = x_central - x_k and w_k = a_central - a_k (a = the velocity or the
// step displacement) of all (<= 3) constraints, gathered from the page by the lanes the cluster word lists; rows beyond np are zero.
template <class mixed>
__device__ __forceinline__ void cluster_gather(const mixed* page, unsigned word, mixed (&c)[6], mixed (&r)[3][3], mixed (&w)[3][3]) {
    // All 24 reads are unconditional and issued together (a read under `k < np` becomes a divergent block with its own wait: nine
    // serialised LDS round trips in the first version of this).  The cluster word names the CENTRAL lane for the peripherals a
    // cluster does not have, so their rows come out as c - c = 0 exactly, without a select.
    const int lc = (int) ((word >> SHAKE_WORD_CENTRAL_SHIFT) & 63u);
    const int p0 = (int) ((word >> 4) & 63u), p1 = (int) ((word >> 10) & 63u), p2 = (int) ((word >> 16) & 63u);
    mixed q[3][6];
#pragma unroll
    for (int a = 0; a < 6; a++) { c[a] = page[a * 64 + lc]; q[0][a] = page[a * 64 + p0]; q[1][a] = page[a * 64 + p1]; q[2][a] = page[a * 64 + p2]; }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int a = 0; a < 3; a++) { r[k][a] = c[a] - q[k][a]; w[k][a] = c[3 + a] - q[k][3 + a]; }
}

// Hydrogen-type clusters, ALL constraints of a cluster at once (default; oracle/vv_oracle.c: vvo_cluster_velocities_direct /
// vvo_cluster_positions_newton state the same arithmetic).  Every lane of a cluster -- central and peripheral alike -- gathers the
// cluster's bonds from the page and solves the k x k system (k <= 3) for itself: the wave executes ONE instruction stream whatever the
// number of lanes that take part, so repeating the solve in four lanes costs nothing, while a central lane sweeping over its three
// constraints one after the other (the Gauss-Seidel form below, OpenMM's iteration) is a three times longer serial chain of dependent
// fp64 operations, repeated four times until its last sweep finds nothing to correct (measured: 2.8 us of an 18.3 us step at C3 with
// its 33 000 constraints, profiles/r03a_shake_cost.txt).
//   velocities: (u_k + imc sum_m l_m r_m + imp l_k r_k) . r_k = 0 is linear in the multipliers l: closed-form solve, no iteration;
//   positions:  Newton on g_k(l) = |s_k + imc sum_m l_m r_m + imp l_k r_k|^2 - d^2 with the exact diagonal (imc + imp) b_k . r_k and the
//               off-diagonals at the old bonds; OpenMM's convergence test (|g_k| < tol d^2 for every constraint of the cluster).
// OpenMM's source is not under /root/reference: parity with OpenMM itself is unpinned either way.
template <class mixed>
__device__ __forceinline__ void cluster_velocities_direct(unsigned word, float4 prm, mixed x, mixed y, mixed z, mixed& vx, mixed& vy, mixed& vz, const mixed* page) {
    mixed c[6], r[3][3], u[3][3];
    cluster_gather<mixed>(page, word, c, r, u);
    const int np = (int) ((word >> 2) & 3u);
    const mixed imc = prm.x, imp = prm.w, ims = imc + imp;
    const mixed b0 = dot3(u[0], r[0]), b1 = dot3(u[1], r[1]), b2 = dot3(u[2], r[2]);
    const mixed A00 = ims * dot3(r[0], r[0]);
    const mixed A11 = np > 1 ? ims * dot3(r[1], r[1]) : (mixed) 1;
    const mixed A22 = np > 2 ? ims * dot3(r[2], r[2]) : (mixed) 1;
    const mixed A01 = imc * dot3(r[0], r[1]), A02 = imc * dot3(r[0], r[2]), A12 = imc * dot3(r[1], r[2]);
    const Sym3Inv<mixed> q = sym3_cofactors<mixed>(A00, A01, A02, A11, A12, A22);
    const mixed l0 = -((q.c00 * b0 + q.c01 * b1 + q.c02 * b2) * q.inv);
    const mixed l1 = -((q.c01 * b0 + q.c11 * b1 + q.c12 * b2) * q.inv);
    const mixed l2 = -((q.c02 * b0 + q.c12 * b1 + q.c22 * b2) * q.inv);
    if (word & 1u) {
        vx += imc * (l0 * r[0][0] + l1 * r[1][0] + l2 * r[2][0]);
        vy += imc * (l0 * r[0][1] + l1 * r[1][1] + l2 * r[2][1]);
        vz += imc * (l0 * r[0][2] + l1 * r[1][2] + l2 * r[2][2]);
    } else {
        const int k = (int) ((word >> SHAKE_WORD_OWN_SHIFT) & 3u);
        const mixed f = imp * (k == 0 ? l0 : (k == 1 ? l1 : l2));
        vx -= f * (c[0] - x); vy -= f * (c[1] - y); vz -= f * (c[2] - z);      // own bond: the same bits as r[k]
    }
}
template <class mixed>
__device__ __forceinline__ void cluster_positions_newton(unsigned word, float4 prm, mixed tol, mixed x, mixed y, mixed z, mixed& dx, mixed& dy, mixed& dz, const mixed* page, bool member) {
    mixed c[6], r[3][3], s[3][3], b[3][3];
    int np = 0;
    mixed imc = 0, imp = 0, ims = 0, d2 = 0, d2tol = 0, O01 = 0, O02 = 0, O12 = 0;
    if (member) {
        cluster_gather<mixed>(page, word, c, r, s);
        np = (int) ((word >> 2) & 3u);
        imc = prm.x; d2 = prm.z; imp = prm.w; ims = imc + imp; d2tol = d2 * tol;
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int a = 0; a < 3; a++) { s[k][a] = r[k][a] + s[k][a]; b[k][a] = s[k][a]; }
        O01 = imc * dot3(r[0], r[1]); O02 = imc * dot3(r[0], r[2]); O12 = imc * dot3(r[1], r[2]);
    }
    mixed l0 = 0, l1 = 0, l2 = 0, tx = 0, ty = 0, tz = 0;
    bool live = member;
    for (int iteration = 0; iteration < 15; iteration++) {
        mixed g0 = 0, g1 = 0, g2 = 0;
        if (live) {
            g0 = dot3(b[0], b[0]) - d2;
            g1 = np > 1 ? dot3(b[1], b[1]) - d2 : (mixed) 0;
            g2 = np > 2 ? dot3(b[2], b[2]) - d2 : (mixed) 0;
            live = fabs(g0) >= d2tol || fabs(g1) >= d2tol || fabs(g2) >= d2tol;      // a cluster that is within tolerance stays there
        }
        if (!__any(live)) break;
        if (live) {
            const mixed D0 = ims * dot3(b[0], r[0]);
            const mixed D1 = np > 1 ? ims * dot3(b[1], r[1]) : (mixed) 1;
            const mixed D2 = np > 2 ? ims * dot3(b[2], r[2]) : (mixed) 1;
            const Sym3Inv<mixed> q = sym3_cofactors<mixed>(D0, O01, O02, D1, O12, D2);
            const mixed h0 = 0.5f * g0, h1 = 0.5f * g1, h2 = 0.5f * g2;
            l0 -= (q.c00 * h0 + q.c01 * h1 + q.c02 * h2) * q.inv;
            l1 -= (q.c01 * h0 + q.c11 * h1 + q.c12 * h2) * q.inv;
            l2 -= (q.c02 * h0 + q.c12 * h1 + q.c22 * h2) * q.inv;
            tx = imc * (l0 * r[0][0] + l1 * r[1][0] + l2 * r[2][0]);
            ty = imc * (l0 * r[0][1] + l1 * r[1][1] + l2 * r[2][1]);
            tz = imc * (l0 * r[0][2] + l1 * r[1][2] + l2 * r[2][2]);
            const mixed f0 = imp * l0, f1 = imp * l1, f2 = imp * l2;
            b[0][0] = (s[0][0] + tx) + f0 * r[0][0]; b[0][1] = (s[0][1] + ty) + f0 * r[0][1]; b[0][2] = (s[0][2] + tz) + f0 * r[0][2];
            if (np > 1) { b[1][0] = (s[1][0] + tx) + f1 * r[1][0]; b[1][1] = (s[1][1] + ty) + f1 * r[1][1]; b[1][2] = (s[1][2] + tz) + f1 * r[1][2]; }
            if (np > 2) { b[2][0] = (s[2][0] + tx) + f2 * r[2][0]; b[2][1] = (s[2][1] + ty) + f2 * r[2][1]; b[2][2] = (s[2][2] + tz) + f2 * r[2][2]; }
        }
    }
    if (member) {
        if (word & 1u) {
            dx = dx + tx; dy = dy + ty; dz = dz + tz;
        } else {
            const int k = (int) ((word >> SHAKE_WORD_OWN_SHIFT) & 3u);
            const mixed f = imp * (k == 0 ? l0 : (k == 1 ? l1 : l2));
            dx -= f * (c[0] - x); dy -= f * (c[1] - y); dz -= f * (c[2] - z);
        }
    }
}

// Constraint clusters of the kind OpenMM's SHAKE kernels take, Gauss-Seidel form (VVHIP_SHAKE_MODE=0; generic kernels only): the
// central lane iterates like OpenMM's applyShakeToPositions / applyShakeToVelocities (sweeps over the cluster's constraints, <= 15,
// float cluster parameters, `mixed` arithmetic) and hands the peripherals their result through the page.
template <class mixed>
__device__ __forceinline__ void cluster_positions_sweeps(unsigned word, float4 prm, mixed tol, mixed x, mixed y, mixed z, mixed& dx, mixed& dy, mixed& dz, mixed* page) {
    const int np = (int) ((word >> 2) & 3u);
    const mixed invMassCentral = prm.x, avgMass = prm.y, d2 = prm.z, invMassPeripheral = prm.w;
    mixed rij[3][3], rijsq[3], ld[3], xpj[3][3];
    int pl[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        pl[k] = (int) ((word >> (4 + 6 * k)) & 63u);
        if (k < np) {
            rij[k][0] = x - page[pl[k]]; rij[k][1] = y - page[64 + pl[k]]; rij[k][2] = z - page[128 + pl[k]];
            xpj[k][0] = page[192 + pl[k]]; xpj[k][1] = page[256 + pl[k]]; xpj[k][2] = page[320 + pl[k]];
        } else {
            rij[k][0] = rij[k][1] = rij[k][2] = 0; xpj[k][0] = xpj[k][1] = xpj[k][2] = 0;
        }
        rijsq[k] = rij[k][0] * rij[k][0] + rij[k][1] * rij[k][1] + rij[k][2] * rij[k][2];
        ld[k] = d2 - rijsq[k];
    }
    mixed xpi[3] = {dx, dy, dz};
    const mixed d2tol = d2 * tol;
    bool converged = false;
    for (int iteration = 0; iteration < 15 && !converged; iteration++) {
        converged = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k < np) {
                const mixed rp0 = xpi[0] - xpj[k][0], rp1 = xpi[1] - xpj[k][1], rp2 = xpi[2] - xpj[k][2];
                const mixed rpsqij = rp0 * rp0 + rp1 * rp1 + rp2 * rp2;
                const mixed rrpr = rij[k][0] * rp0 + rij[k][1] * rp1 + rij[k][2] * rp2;
                // OpenMM's test is fabs(..) / (d2 * tol) >= 1; the product form decides the same up to the last bit of the quotient
                // (the oracle states it the same way)
                const mixed num = ld[k] - 2.0f * rrpr - rpsqij;
                const mixed acor = num * avgMass / (rrpr + rijsq[k]);
                if (fabs(num) >= d2tol) {
                    const mixed d0 = rij[k][0] * acor, d1 = rij[k][1] * acor, d2v = rij[k][2] * acor;
                    xpi[0] += d0 * invMassCentral; xpi[1] += d1 * invMassCentral; xpi[2] += d2v * invMassCentral;
                    xpj[k][0] -= d0 * invMassPeripheral; xpj[k][1] -= d1 * invMassPeripheral; xpj[k][2] -= d2v * invMassPeripheral;
                    converged = false;
                }
            }
        }
    }
    dx = xpi[0]; dy = xpi[1]; dz = xpi[2];
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (k < np) { page[192 + pl[k]] = xpj[k][0]; page[256 + pl[k]] = xpj[k][1]; page[320 + pl[k]] = xpj[k][2]; }
}
template <class mixed>
__device__ __forceinline__ void cluster_velocities_sweeps(unsigned word, float4 prm, mixed tol, mixed x, mixed y, mixed z, mixed& vx, mixed& vy, mixed& vz, mixed* page) {
    const int np = (int) ((word >> 2) & 3u);
    const mixed invMassCentral = prm.x, avgMass = prm.y, invMassPeripheral = prm.w;
    mixed rij[3][3], rijsq[3], vj[3][3];
    int pl[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        pl[k] = (int) ((word >> (4 + 6 * k)) & 63u);
        if (k < np) {
            rij[k][0] = x - page[pl[k]]; rij[k][1] = y - page[64 + pl[k]]; rij[k][2] = z - page[128 + pl[k]];
            vj[k][0] = page[192 + pl[k]]; vj[k][1] = page[256 + pl[k]]; vj[k][2] = page[320 + pl[k]];
        } else {
            rij[k][0] = rij[k][1] = rij[k][2] = 0; vj[k][0] = vj[k][1] = vj[k][2] = 0;
        }
        rijsq[k] = rij[k][0] * rij[k][0] + rij[k][1] * rij[k][1] + rij[k][2] * rij[k][2];
        rijsq[k] = k < np ? (mixed) 1 / rijsq[k] : (mixed) 0;     // the bond does not move during the sweeps: one reciprocal, not one division per visit
    }
    mixed vi[3] = {vx, vy, vz};
    bool converged = false;
    for (int iteration = 0; iteration < 15 && !converged; iteration++) {
        converged = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k < np) {
                const mixed rp0 = vi[0] - vj[k][0], rp1 = vi[1] - vj[k][1], rp2 = vi[2] - vj[k][2];
                const mixed rrpr = rp0 * rij[k][0] + rp1 * rij[k][1] + rp2 * rij[k][2];
                const mixed delta = -2.0f * avgMass * rrpr * rijsq[k];
                const mixed d0 = rij[k][0] * delta, d1 = rij[k][1] * delta, d2v = rij[k][2] * delta;
                vi[0] += d0 * invMassCentral; vi[1] += d1 * invMassCentral; vi[2] += d2v * invMassCentral;
                vj[k][0] -= d0 * invMassPeripheral; vj[k][1] -= d1 * invMassPeripheral; vj[k][2] -= d2v * invMassPeripheral;
                if (fabs(delta) > tol) converged = false;
            }
        }
    }
    vx = vi[0]; vy = vi[1]; vz = vi[2];
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (k < np) { page[192 + pl[k]] = vj[k][0]; page[256 + pl[k]] = vj[k][1]; page[320 + pl[k]] = vj[k][2]; }
}

// One call per tile: every lane of the wave walks through it.  `word` = the lane's cluster word (0: not in a cluster).  has_shake /
// has_settle: the plan holds hydrogen-type clusters / rigid three-site molecules (stage bits, compile-time constants in the
// specialised kernels: a box of ionic liquid carries no SETTLE code, a box of water no cluster solver); gs = the Gauss-Seidel form
// for the hydrogen-type clusters.  Rigid molecules: analytic, solved by the apex lane, partners' results handed back through the page.
template <class mixed>
__device__ __forceinline__ void shake_positions(int lane, unsigned word, float4 prm, mixed tol, mixed x, mixed y, mixed z, mixed invm,
                                                mixed& dx, mixed& dy, mixed& dz, mixed* page, bool has_shake, bool has_settle, bool gs) {
    const bool member = (word & 3u) != 0, settle = has_settle && (word & SHAKE_WORD_SETTLE) != 0;
    if (member) {
        page[lane] = x; page[64 + lane] = y; page[128 + lane] = z; page[192 + lane] = dx; page[256 + lane] = dy; page[320 + lane] = dz;
        if (settle) page[384 + lane] = invm;
    }
    VV_WAVE_LDS_FENCE();
    if (has_shake && !gs) cluster_positions_newton<mixed>(word, prm, tol, x, y, z, dx, dy, dz, page, member && !settle);
    if (has_settle || (has_shake && gs)) {            // someone solves for its cluster mates and hands the results back
        if ((word & 1u) && settle) {
            const int l1 = (int) ((word >> 4) & 63u), l2 = (int) ((word >> 10) & 63u);
            mixed d1x = page[192 + l1], d1y = page[256 + l1], d1z = page[320 + l1], d2x = page[192 + l2], d2y = page[256 + l2], d2z = page[320 + l2];
            settle_positions_math<mixed>((mixed) 1 / invm, (mixed) 1 / page[384 + l1], (mixed) prm.x, (mixed) prm.y, x, y, z,
                                         page[l1], page[64 + l1], page[128 + l1], page[l2], page[64 + l2], page[128 + l2], dx, dy, dz, d1x, d1y, d1z, d2x, d2y, d2z);
            page[192 + l1] = d1x; page[256 + l1] = d1y; page[320 + l1] = d1z; page[192 + l2] = d2x; page[256 + l2] = d2y; page[320 + l2] = d2z;
        } else if (has_shake && gs && (word & 1u)) {
            cluster_positions_sweeps<mixed>(word, prm, tol, x, y, z, dx, dy, dz, page);
        }
        VV_WAVE_LDS_FENCE();
        if ((word & 2u) && (settle || (has_shake && gs))) { dx = page[192 + lane]; dy = page[256 + lane]; dz = page[320 + lane]; }
    }
}

template <class mixed>
__device__ __forceinline__ void shake_velocities(int lane, unsigned word, float4 prm, mixed tol, mixed x, mixed y, mixed z, mixed invm,
                                                 mixed& vx, mixed& vy, mixed& vz, mixed* page, bool has_shake, bool has_settle, bool gs) {
    const bool member = (word & 3u) != 0, settle = has_settle && (word & SHAKE_WORD_SETTLE) != 0;
    if (member) {
        page[lane] = x; page[64 + lane] = y; page[128 + lane] = z; page[192 + lane] = vx; page[256 + lane] = vy; page[320 + lane] = vz;
        if (settle) page[384 + lane] = invm;
    }
    VV_WAVE_LDS_FENCE();
    if (has_shake && !gs && member && !settle) cluster_velocities_direct<mixed>(word, prm, x, y, z, vx, vy, vz, page);
    if (has_settle || (has_shake && gs)) {
        if ((word & 1u) && settle) {
            const int l1 = (int) ((word >> 4) & 63u), l2 = (int) ((word >> 10) & 63u);
            mixed u1x = page[192 + l1], u1y = page[256 + l1], u1z = page[320 + l1], u2x = page[192 + l2], u2y = page[256 + l2], u2z = page[320 + l2];
            settle_velocities_math<mixed>((mixed) 1 / invm, (mixed) 1 / page[384 + l1], x, y, z,
                                          page[l1], page[64 + l1], page[128 + l1], page[l2], page[64 + l2], page[128 + l2], vx, vy, vz, u1x, u1y, u1z, u2x, u2y, u2z);
            page[192 + l1] = u1x; page[256 + l1] = u1y; page[320 + l1] = u1z; page[192 + l2] = u2x; page[256 + l2] = u2y; page[320 + l2] = u2z;
        } else if (has_shake && gs && (word & 1u)) {
            cluster_velocities_sweeps<mixed>(word, prm, tol, x, y, z, vx, vy, vz, page);
        }
        VV_WAVE_LDS_FENCE();
        if ((word & 2u) && (settle || (has_shake && gs))) { vx = page[192 + lane]; vy = page[256 + lane]; vz = page[320 + lane]; }
    }
}

// ================================================================================ multi-GPU mailbox (vv_kernels.hpp: Mailbox)
// Head of kernel B, thermostat waves.  Every wave has just folded this rank's accumulators (complete: kernel A ended); block 0's
// stores the totals into slot [seq & 1][rank] of every OTHER rank's box; then every block's wave polls its own box until the other
// ranks' words carry the sequence number and adds them (int64: any order gives the same bits).  Bounded wait: after ~5 s without the peers' words the wave raises
// ctl[0] and carries on (the host reports the failure; nothing ever hangs the GPU).
__device__ __forceinline__ void mailbox_exchange(const KArgs& a, int lane, unsigned int seq, unsigned int* words, long long (&tot)[NUM_ACC]) {
    const int par = (int) (seq & 1u);
    const int nwords = a.mb.ranks * MB_WORDS;
    if (blockIdx.x == 0) {
        for (int i = lane; i < nwords; i += 64) {
            const int peer = i / MB_WORDS, w = i % MB_WORDS;
            if (peer == a.mb.rank) continue;
            long long t = 0;
#pragma unroll
            for (int k = 0; k < NUM_ACC; k++) if ((w >> 1) == k) t = tot[k];
            const unsigned int payload = (w & 1) ? (unsigned int) ((unsigned long long) t >> 32) : (unsigned int) t;
            unsigned long long* box = a.mb.peers[peer];
            __hip_atomic_store(&box[((size_t) par * a.mb.ranks + a.mb.rank) * MB_WORDS + w], ((unsigned long long) seq << 32) | payload,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const unsigned long long* box = a.mb.local + (size_t) par * nwords;
    const long long t0 = wall_clock64();
    const bool dead = a.mb.ctl[0] != 0;            // an earlier wait ran out: the run is void, do not wait again
    for (int i = lane; i < nwords; i += 64) {
        unsigned long long v;
        if (i / MB_WORDS == a.mb.rank) continue;
        for (;;) {
            v = __hip_atomic_load(&box[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned int) (v >> 32) == seq) break;
            if (dead || wall_clock64() - t0 > 500000000LL) {                                 // 100 MHz counter
                a.mb.ctl[0] = 1u;
                if (a.status) __hip_atomic_store(&a.status[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the host sees it without a sync
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        words[i] = (unsigned int) v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < NUM_ACC; k++) {
        long long s = tot[k];                      // this rank's own total never leaves the registers
        for (int r = 0; r < a.mb.ranks; r++)
            if (r != a.mb.rank) s += (long long) ((unsigned long long) words[r * MB_WORDS + 2 * k] | ((unsigned long long) words[r * MB_WORDS + 2 * k + 1] << 32));
        tot[k] = s;
    }
}

// Shader-clock stamps of the instrumented build (-DVV_KERNEL_TIMESTAMPS, tools/probes): empty macros otherwise
#include "vv_probes.inc"

// ================================================================================ kernel A
// Kernel-argument preload (gfx950: up to 16 user SGPRs are filled from the head of the kernarg segment by the packet processor
// before the wave starts; Makefile: -mllvm -amdgpu-kernarg-preload-count).  A wave's FIRST memory operation -- the load of its
// slot words, for the thermostat wave of kernel B the accumulator / state / chain-constant loads -- needs only a pointer and a
// count; passed as leading scalar parameters they are in SGPRs at wave entry, and the s_load round trip to the kernarg segment
// (one of three dependent memory round trips of ~0.4 us each in front of a tile's arithmetic, profiles/r02a_timeline_*) leaves
// the critical path; everything else in KArgs is fetched in the shadow of that first load.
template <class real, class mixed, uint32_t SF>   // SF != 0: the stage bits are a compile-time constant (hot paths)
__global__ void __launch_bounds__(512) vv_kernel_a(const int2* __restrict__ pre_slots, const int pre_nwaves, const int pre_wpb, void* pre_velm, const long long* __restrict__ pre_force,
                                                   const int pre_padded, const KArgs a) {
    using real4 = typename Vec<real>::v4;
    using real3 = typename Vec<real>::v3;
    using mixed4 = typename Vec<mixed>::v4;
    using P = Prec<real>;
    const int lane = threadIdx.x & 63;
    const uint32_t F = SF ? SF : a.flags;
    double k_atom = 0, k_com = 0, k_drude = 0, k_bias = 0;
    double m_ab[3] = {0, 0, 0}, m_bb[3] = {0, 0, 0};      // A_KE_MOM: cross and field moments of the groups atom, com, drude

    // grid-stride over 64-lane tiles (the grid is capped in launch_a): per-lane partial sums run across all tiles of the
    // block, so the block reduction and its atomics are paid once per block however large the system is
    // pre_wpb = waves per block = blockDim.x / 64, as a preloaded argument: blockDim itself is a hidden kernel argument, i.e. another
    // s_load in front of the first slot load
    VV_STAMP(threadIdx.x >> 6, 0);
    VV_SPAN_BEGIN;
    // this block's accumulator slots as they are now (block_accumulate: exclusive slots take a plain store of old + new at the end)
    unsigned long long acc_old = 0;
    if ((F & (A_KE | A_BIAS | A_KE_PLAIN)) && a.acc_exclusive && threadIdx.x < NUM_ACC) acc_old = a.acc[threadIdx.x * ACC_SLOTS + (blockIdx.x & (ACC_SLOTS - 1))];
    // (Requesting the NEXT tile's slot words one tile ahead in this grid-stride loop was measured, same box, three alternating runs:
    // 8.9 M particles A 237.7 -> 236.1 us, B 264.1 -> 267.7 us; 111 k particles A 5.27 -> 5.47 us.  Not kept.)
    // Periodic layout: the role words and segment masses of a region's waves are those of its pattern wave (the region's first wave of
    // cell 0).  The block copies the rows of all (<= 4) regions into LDS once: thousands of waves re-reading the same few cache lines
    // tile after tile queue up on the L2 channels that hold them (measured at 8.9 M particles: kernel A 110 -> 117 us).
    __shared__ unsigned sh_pat_meta[4][64];
    __shared__ double2 sh_pat_segm[4][64];
    __shared__ unsigned sh_pat_shake[4][64];        // constraint cluster words / parameters of the pattern waves (A_SHAKE_V)
    __shared__ float4 sh_pat_prm[4][64];
    if (F & A_PERIODIC) {
        for (int row = threadIdx.x >> 6; row < 4; row += pre_wpb) {
            const int ws = a.per.wave_start[row];
            unsigned m = 0;
            if (ws != 0x7fffffff) m = (unsigned) pre_slots[(size_t) ws * 64 + lane].y;
            sh_pat_meta[row][lane] = m;
            if (F & A_CONS) {
                const bool member = ws != 0x7fffffff && (m & META_SHAKE);
                sh_pat_shake[row][lane] = member ? (unsigned) a.slot_shake[(size_t) ws * 64 + lane] : 0u;
                sh_pat_prm[row][lane] = member ? a.slot_shake_param[(size_t) ws * 64 + lane] : make_float4(0, 0, 0, 0);
            }
            int ss = a.per.d_seg[0];
#pragma unroll
            for (int k = 1; k < 4; k++) ss += k <= row ? a.per.d_seg[k] : 0;
            const int below = (int) lanes_below(__ballot((m & META_COM_LEADER) != 0));
            double2 sm = {0, 0};
            if ((F & A_KE) && (m & META_COM_LEADER)) sm = ((const double2*) a.seg_mass)[ss + below];
            sh_pat_segm[row][lane] = sm;
        }
        __syncthreads();
    }
    // Arithmetic layout: a tile's particle addresses follow from the wave index alone, so the NEXT tile's velocity and force loads are
    // issued before this tile's arithmetic (two tiles in flight per wave; with loaded slot words the same was tried one tile ahead and
    // did not pay, see above).  pw_n / v_n / f*_n carry the next tile; the first tile is loaded in front of the loop.
    PeriodicWave pw_n = {0, 0, 0, 0};
    mixed4 v_n = {0, 0, 0, 0};
    long long fx_n = 0, fy_n = 0, fz_n = 0;
    const int wave_first = blockIdx.x * pre_wpb + (threadIdx.x >> 6), wave_stride = gridDim.x * pre_wpb;
    auto request_tile = [&](int w) {
        pw_n = periodic_wave(a.per, w);
        const bool in = lane < pw_n.count;
        const int at = pw_n.atom0 + lane;
        v_n = mixed4{0, 0, 0, 0}; fx_n = fy_n = fz_n = 0;
        if (in) v_n = ((const mixed4*) pre_velm)[at];
        if ((F & (A_KICK_FULL | A_KICK_HALF)) && in) { fx_n = pre_force[at]; fy_n = pre_force[at + pre_padded]; fz_n = pre_force[at + 2 * pre_padded]; }
    };
    if ((F & A_PERIODIC) && wave_first < pre_nwaves) request_tile(wave_first);
    for (int wave = wave_first; wave < pre_nwaves; wave += wave_stride) {
        int atom;
        unsigned meta;
        PeriodicWave pw = {0, 0, 0, 0};
        mixed4* velm = (mixed4*) a.velm;
        mixed4 v = {0, 0, 0, 0};
        long long fx = 0, fy = 0, fz = 0;
        int segb = 0;
        if (F & A_PERIODIC) {
            // the particle index is arithmetic: velocity and force were requested a tile ago, no slot word in front of them
            pw = pw_n; v = v_n; fx = fx_n; fy = fy_n; fz = fz_n;
            if (wave + wave_stride < pre_nwaves) request_tile(wave + wave_stride);
            const bool in = lane < pw.count;
            atom = in ? pw.atom0 + lane : -1;
            meta = in ? sh_pat_meta[pw.region][lane] : 0u;
        } else {
            // (the wave's segment base is requested FIRST: it returns with the slot word, and everything below needs only these two)
            if (F & A_KE) segb = a.seg_base[__builtin_amdgcn_readfirstlane(wave)];
            const int2 slot = pre_slots[(size_t) wave * 64 + lane];
            atom = slot.x;
            meta = (unsigned) slot.y;
        }
        const unsigned role = meta & META_ROLE_MASK;
        const int partner = (meta >> META_PARTNER_SHIFT) & 63;
        const bool act = atom >= 0;
        // ---- every load of the tile that needs nothing but the slot word, in ONE batch: unconditional loads from clamped indices
        // (idle lanes read particle 0 / their own table entry and discard it), closed by a scheduling barrier.  Written as
        // `if (act) x = array[atom]`, each load sits in a divergent block of its own, and the backend neither batches loads across
        // blocks nor hoists them above the first use of an earlier one: kernel A went through FOUR dependent memory round trips per
        // tile (slot -> velocity and force -> segment base -> segment mass), kernel B through six.
        const int ai = act ? atom : 0;
        const size_t li = (size_t) wave * 64 + lane;
        if (!(F & A_PERIODIC)) {
            v = velm[ai];
            if (F & (A_KICK_FULL | A_KICK_HALF)) { fx = a.force[ai]; fy = a.force[ai + a.padded]; fz = a.force[ai + 2 * a.padded]; }
        }
        double2 seg_mw = {0, 0};                 // (mass, 1/mass) of this lane's COM segment, one 16-byte entry per segment
        // dense index of this lane's COM segment: segments are numbered wave by wave and, inside a wave, by the lane of their leader
        // (= last) lane, so every lane of a segment counts the same leaders below itself (vv_host.cpp: seg_base; the tables end
        // with one spare entry, which the lanes behind a wave's last leader read)
        int segi = 0;
        if (F & A_KE) {
            const int below = (int) lanes_below(__ballot((meta & META_COM_LEADER) != 0));
            segi = ((F & A_PERIODIC) ? pw.seg0 : segb) + below;
            seg_mw = (F & A_PERIODIC) ? sh_pat_segm[pw.region][lane] : ((const double2*) a.seg_mass)[segi];
        }
        // Static per-lane masses (A_MTAB): m = RECIP(velm.w) and, for the members of a Drude pair, the mass fraction m / (m1 + m2), both
        // formed ONCE by vv_kernel_mass_table with the very operations the stages below used to repeat every step (IEEE quotients of the
        // mode's `mixed` type), so every value is bit-identical to the per-step one; velm.w never changes during a run.
        mixed tab_m = 0, tab_f = 0;
        if ((F & A_MTAB) && (F & (A_KE | A_BIAS | A_COS | A_LD | A_KE_PLAIN | A_COMPART))) {
            tab_m = (mixed) a.slot_m[li];
            if (F & A_KE) tab_f = (mixed) a.slot_f[li];
        }
        real4 pq = {0, 0, 0, 0};
        const bool need_pos_any = ((F & (A_COS | A_BIAS | A_UNBIAS_ACC)) && !(F & A_CZ_LOAD)) || (F & A_EF);
        if (need_pos_any) pq = ((const real4*) a.posq)[ai];
        double czl = 0;          // cos(2 pi z / Lz): evaluated at most once per launch, cached across the launches of a step
        if ((F & (A_COS | A_BIAS | A_UNBIAS_ACC)) && (F & A_CZ_LOAD)) czl = a.cosz[li];
        real3 fe = {0, 0, 0};    // extra force (VVIntegrator.cpp:238-245), accumulated in `real` like forceExtra
        if (F & A_FE_LOAD) fe = ((const real3*) a.fextra)[ai];
        // in-kernel constraints: cluster word, parameters and position of every lane (members use them)
        unsigned cons_word = 0;
        float4 cons_prm = make_float4(0, 0, 0, 0);
        real4 cons_p1 = {0, 0, 0, 0}, cons_p2 = {0, 0, 0, 0};
        if (F & A_CONS) {
            cons_word = (F & A_PERIODIC) ? sh_pat_shake[pw.region][lane] : (unsigned) a.slot_shake[li];
            cons_prm = (F & A_PERIODIC) ? sh_pat_prm[pw.region][lane] : a.slot_shake_param[li];
            cons_p1 = ((const real4*) a.posq)[ai];
            if (PosIO<real, mixed>::kMixed) cons_p2 = ((const real4*) a.corr)[ai];
        }
        // Langevin lanes: slot of the normal deviates with the batch, the deviates themselves behind it (keyed by the role word)
        int rand_slot = 0;
        if (F & A_LD) rand_slot = a.slot_rand[li];
        __builtin_amdgcn_sched_barrier(0);
        float4 rnd_a = make_float4(0, 0, 0, 0), rnd_b = make_float4(0, 0, 0, 0);
        if ((F & A_LD) && act && (role == ROLE_LD_NORMAL || role == ROLE_LD_DRUDE || role == ROLE_LD_PARENT)) {
            const unsigned ri = a.random_index + (unsigned) rand_slot;
            rnd_a = a.random[ri];
            if (role != ROLE_LD_NORMAL) rnd_b = a.random[ri + 1];
        }
        if (!act) { v = mixed4{0, 0, 0, 0}; fe = real3{0, 0, 0}; }
        if (!(act && (meta & META_COM_LEADER))) seg_mw = double2{0, 0};
        if (!(act && (meta & META_MASSIVE))) { tab_m = 0; tab_f = 0; }
        if (!(meta & META_PAIR)) tab_f = 0;
        const bool massive = act && v.w != 0;
        const mixed stepSize = (mixed) a.dt;
        // own mass: bit-exact form (feeds element-wise results) and the form for quantities that only feed reductions
        auto mass_exact = [&]() -> mixed { return (F & A_MTAB) ? tab_m : P::RECIP(v.w); };
        auto mass_sum = [&]() -> mixed { return (F & A_MTAB) ? tab_m : P::RECIP_SUM(v.w); };

        if (F & (A_COS | A_BIAS | A_UNBIAS_ACC)) {
            if (!(F & A_CZ_LOAD)) czl = cos_kz<real>(pq.z, (real) a.inv_box_z);
            if (F & A_CZ_STORE) a.cosz[li] = czl;
        }
        // ---------------- extra force (VVIntegrator.cpp:238-245), accumulated in `real` like forceExtra
        if (F & A_LD) {
            const mixed pvx = shfl(v.x, partner), pvy = shfl(v.y, partner), pvz = shfl(v.z, partner), pvw = shfl(v.w, partner);
            const mixed dragFactor = (mixed) a.drag, randFactor = (mixed) a.randf;
            const mixed dragFactorDrude = (mixed) a.drag_drude, randFactorDrude = (mixed) a.randf_drude;
            if (role == ROLE_LD_NORMAL && massive) {                        // K/drudeLangevin.cu:14-25
                const mixed mass = mass_exact();
                const mixed sqrtMass = P::SQRT(mass);
                const float4 rnd = rnd_a;
                fe.x += (-dragFactor * mass * v.x + randFactor * sqrtMass * rnd.x);
                fe.y += (-dragFactor * mass * v.y + randFactor * sqrtMass * rnd.y);
                fe.z += (-dragFactor * mass * v.z + randFactor * sqrtMass * rnd.z);
            } else if (role == ROLE_LD_DRUDE || role == ROLE_LD_PARENT) {   // K/drudeLangevin.cu:29-59
                const bool isd = role == ROLE_LD_DRUDE;
                const mixed v1x = isd ? v.x : pvx, v1y = isd ? v.y : pvy, v1z = isd ? v.z : pvz, v1w = isd ? v.w : pvw;
                const mixed v2x = isd ? pvx : v.x, v2y = isd ? pvy : v.y, v2z = isd ? pvz : v.z, v2w = isd ? pvw : v.w;
                const mixed mass1 = P::RECIP(v1w), mass2 = P::RECIP(v2w);
                const mixed totMass = mass1 + mass2;
                const mixed sqrtTotMass = P::SQRT(totMass);
                const mixed redMass = P::RECIP((mass1 + mass2) * v1w * v2w);
                const mixed sqrtRedMass = P::SQRT(redMass);
                const mixed invTotMass = P::RECIP(totMass);
                const mixed mass1fract = invTotMass * mass1, mass2fract = invTotMass * mass2;
                const mixed cmx = v1x * mass1fract + v2x * mass2fract;
                const mixed cmy = v1y * mass1fract + v2y * mass2fract;
                const mixed cmz = v1z * mass1fract + v2z * mass2fract;
                const mixed rx = v2x - v1x, ry = v2y - v1y, rz = v2z - v1z;
                const float4 rand1 = rnd_a, rand2 = rnd_b;
                real3 cmForce, relForce;
                cmForce.x = (-dragFactor * totMass * cmx + randFactor * sqrtTotMass * rand1.x);
                cmForce.y = (-dragFactor * totMass * cmy + randFactor * sqrtTotMass * rand1.y);
                cmForce.z = (-dragFactor * totMass * cmz + randFactor * sqrtTotMass * rand1.z);
                relForce.x = (-dragFactorDrude * redMass * rx + randFactorDrude * sqrtRedMass * rand2.x);
                relForce.y = (-dragFactorDrude * redMass * ry + randFactorDrude * sqrtRedMass * rand2.y);
                relForce.z = (-dragFactorDrude * redMass * rz + randFactorDrude * sqrtRedMass * rand2.z);
                if (isd) {      // scalar * real3 narrows the mass fraction to `real` (K/vectorOps.cu:427,451)
                    const real m1f = (real) mass1fract;
                    fe.x += m1f * cmForce.x - relForce.x; fe.y += m1f * cmForce.y - relForce.y; fe.z += m1f * cmForce.z - relForce.z;
                } else {
                    const real m2f = (real) mass2fract;
                    fe.x += m2f * cmForce.x + relForce.x; fe.y += m2f * cmForce.y + relForce.y; fe.z += m2f * cmForce.z + relForce.z;
                }
            }
        }
        if ((F & A_EF) && act && (meta & META_EFIELD))                      // K/electricField.cu:8-10
            fe.z += (real) a.efscale * pq.w;
        if ((F & A_COS) && massive)                                         // K/cosineAccelerate.cu:9
            fe.x += (real) a.cos_accel * czl * mass_exact();
        if ((F & A_FE_STORE) && act) ((real3*) a.fextra)[atom] = fe;

        // ---------------- kick
        VV_STAMP(threadIdx.x >> 6, 1);
        if (F & (A_KICK_FULL | A_KICK_HALF)) {
            if (massive) {
                if (F & A_KICK_FULL) {                                      // K/middle.cu:11-21
                    const mixed fscale = stepSize / (mixed) 0x100000000;
                    v.x += stepSize * v.w * fe.x + fscale * v.w * fx;
                    v.y += stepSize * v.w * fe.y + fscale * v.w * fy;
                    v.z += stepSize * v.w * fe.z + fscale * v.w * fz;
                } else {                                                    // K/velocityVerlet.cu:20-22 (0.5 is a double literal)
                    const mixed fscale = (mixed) a.fscale_vv;
                    v.x += 0.5 * stepSize * v.w * fe.x + fscale * v.w * fx;
                    v.y += 0.5 * stepSize * v.w * fe.y + fscale * v.w * fy;
                    v.z += 0.5 * stepSize * v.w * fe.z + fscale * v.w * fz;
                }
                if (!(F & (A_CONS | A_NOSTORE))) store_vec(velm, atom, v);
                if (F & A_POSDELTA_VV) {                                    // K/velocityVerlet.cu:24-26
                    mixed4 d = {stepSize * v.x, stepSize * v.y, stepSize * v.z, 0};
                    ((mixed4*) a.pos_delta)[atom] = d;
                }
            }
        }
        if (F & A_CONS) {                      // integration.applyVelocityConstraints(tol) (HOST:151, 427), clusters solved in the wave
            // one page per wave of the block, sized at launch (dynamic LDS): a static [8] cost 28 KB per block also where blocks have 4 waves
            extern __shared__ double vv_dyn_lds[];
            mixed* shake_page_a = (mixed*) vv_dyn_lds + (threadIdx.x >> 6) * (7 * 64);
            // cluster word, parameters and position came with the tile's load batch
            const bool member = act && (meta & META_SHAKE);
            const unsigned word = member ? cons_word : 0u;
            const float4 prm = cons_prm;
            mixed sx, sy, sz;
            if (PosIO<real, mixed>::kMixed) { sx = cons_p1.x + (mixed) cons_p2.x; sy = cons_p1.y + (mixed) cons_p2.y; sz = cons_p1.z + (mixed) cons_p2.z; }
            else { sx = cons_p1.x; sy = cons_p1.y; sz = cons_p1.z; }
            shake_velocities<mixed>(lane, word, prm, (mixed) a.shake_tol, sx, sy, sz, v.w, v.x, v.y, v.z, shake_page_a,
                                    (F & A_SHAKE_V) != 0, (F & A_SETTLE) != 0, (F & A_SHAKE_GS) != 0);
            if (massive) store_vec(velm, atom, v);
        }
        if ((F & A_POS1) && massive) {                                      // K/middle.cu:33-40
            const mixed halfdt = 0.5f * stepSize;
            mixed4 d = {halfdt * v.x, halfdt * v.y, halfdt * v.z, 0};
            ((mixed4*) a.pos_delta)[atom] = d;
            ((mixed4*) a.old_delta)[atom] = d;
        }

        // ---------------- periodic bias moment (K/cosineAccelerate.cu:24-27; massless -> 0)
        if ((F & A_BIAS) && massive) {
            const mixed t = mass_exact() * v.x * 2 * czl;
            k_bias += (double) t;
        }

        // ---------------- molecules larger than a wave: per-chunk partial sums into the molecule's accumulator
        if ((F & A_COMPART) && a.slot_big) {
            const bool nhb = (role == ROLE_NH_NORMAL || role == ROLE_NH_DRUDE || role == ROLE_NH_PARENT) && (meta & META_BIGMOL);
            mixed bx = v.x;
            if (F & A_UNBIAS_ACC) {
                const mixed V = (mixed) ((double) acc_total(a.acc, 3, lane) * a.acc_inv_scale[3] * a.inv_mass_total);
                if (act) bx -= V * czl;
            }
            const int first = (meta >> META_SEGFIRST_SHIFT) & 63, last = (meta >> META_SEGLAST_SHIFT) & 63;
            mixed mass = 0, mx = 0, my = 0, mz = 0;
            if (nhb && massive) { mass = mass_exact(); mx = bx * mass; my = v.y * mass; mz = v.z * mass; }
            mx = segment_total(mx, lane, first, last); my = segment_total(my, lane, first, last);
            mz = segment_total(mz, lane, first, last); mass = segment_total(mass, lane, first, last);
            if (nhb && (meta & META_COM_LEADER)) {
                unsigned long long* dst = a.bigacc + 4 * (size_t) a.slot_big[(size_t) wave * 64 + lane];
                atomicAdd(dst + 0, (unsigned long long) __double2ll_rn((double) mx * a.big_scale));
                atomicAdd(dst + 1, (unsigned long long) __double2ll_rn((double) my * a.big_scale));
                atomicAdd(dst + 2, (unsigned long long) __double2ll_rn((double) mz * a.big_scale));
                atomicAdd(dst + 3, (unsigned long long) __double2ll_rn((double) mass * a.big_scale));
            }
        }

        VV_STAMP(threadIdx.x >> 6, 2);
        // ---------------- plain kinetic energy of everything massive: sum m v^2 (OpenMM's computeKineticEnergy(0) is half of it)
        if ((F & A_KE_PLAIN) && massive) k_atom += (double) ((v.x * v.x + v.y * v.y + v.z * v.z) * mass_exact());

        // ---------------- kinetic energies of the thermostat groups (K/drudeNoseHoover.cu:33-151)
        // The reference forms, per particle, the velocity relative to the molecular centre of mass, splits every Drude pair into its
        // centre-of-mass and relative motion and squares those.  All three group sums are quadratic forms, and Koenig's theorem turns
        // them into sums that need far less per-lane work (this stage was 85 % of kernel A's instructions, and at the headline size
        // the kernel is VALU-issue bound):
        //     sum_bodies M_b (c_b - V)^2 = sum_lanes m u^2 - sum_pairs mu r^2 - sum_molecules M V^2          (2KE of group "atom")
        // with r = u_parent - u_drude (the molecular V cancels), mu the pair's reduced mass and V = P / M the molecular COM velocity
        // from the segment's momentum P.  So every thermostatted lane adds m u^2, the Drude lane of a pair adds mu r^2 to the Drude
        // group, and the LAST lane of a molecule's segment (the COM leader, vv_host.cpp) takes P = S[last] - S[first-1] from the wave
        // prefix sum of the momenta, stores V for kernel B and adds M V^2 to the COM group; the three differences are formed per lane
        // before the block reduction.  Nothing is broadcast back to the lanes and no pair COM is built.  Reductions only: the sums
        // agree with the reference's to rounding (~1e-15 relative), like every other summation order.
        if (F & A_KE) {
            const bool nh = role == ROLE_NH_NORMAL || role == ROLE_NH_DRUDE || role == ROLE_NH_PARENT;
            mixed ux = v.x;
            if (F & A_UNBIAS_ACC) {                                         // K/cosineAccelerate.cu:53-58, 69-71
                // same expression as the chain kernel writes to scales[3], so kernel B removes exactly this V
                const mixed V = (mixed) ((double) acc_total(a.acc, 3, lane) * a.acc_inv_scale[3] * a.inv_mass_total);
                if (act) ux -= V * czl;
            }
            const int first = (meta >> META_SEGFIRST_SHIFT) & 63, last = (meta >> META_SEGLAST_SHIFT) & 63;
            const bool use_com = first != last || (meta & META_COM_LEADER);
            const bool leader = use_com && (meta & META_COM_LEADER);       // the last lane of its segment
            const bool contrib = nh && massive;
            const mixed own_mass = massive ? mass_sum() : (mixed) 0;
            const mixed wx = ((F & A_KE_MOM) && act) ? (mixed) czl : (mixed) 0;      // the field of the cos perturbation, (cos kz, 0, 0)
            mixed px = 0, py = 0, pz = 0, pw = 0;
            if (contrib) {
                px = ux * own_mass; py = v.y * own_mass; pz = v.z * own_mass;
                k_atom += (double) (px * ux + py * v.y + pz * v.z);
                if (F & A_KE_MOM) { pw = wx * own_mass; m_ab[0] += (double) (px * wx); m_bb[0] += (double) (pw * wx); }
            }
            // ---- Drude pairs: relative motion (K/drudeNoseHoover.cu:97-114, pair.x = Drude); the Drude lane adds for the pair
            if (__any(role == ROLE_NH_DRUDE)) {
                const mixed ox = shfl(ux, partner), oy = shfl(v.y, partner), oz = shfl(v.z, partner);
                const mixed om = (F & A_MTAB) ? shfl(tab_f, partner) : shfl(own_mass, partner);     // partner's mass fraction / mass
                const mixed ow = (F & A_KE_MOM) ? shfl(wx, partner) : (mixed) 0;
                if (role == ROLE_NH_DRUDE) {
                    const mixed reducedMass = (F & A_MTAB) ? own_mass * om : own_mass * om * P::RECIP_SUM(own_mass + om);
                    const mixed rx = ox - ux, ry = oy - v.y, rz = oz - v.z;
                    k_drude += (double) ((rx * rx + ry * ry + rz * rz) * reducedMass);
                    if (F & A_KE_MOM) { const mixed rb = ow - wx; m_ab[2] += (double) (rx * rb * reducedMass); m_bb[2] += (double) (rb * rb * reducedMass); }
                }
            }
            // ---- molecular centre of mass (K/drudeNoseHoover.cu:5-31, 85-94): only where a wave holds COM segments at all
            if (__any(use_com)) {
                const bool in_seg = contrib && use_com;
                mixed Sx = in_seg ? px : (mixed) 0, Sy = in_seg ? py : (mixed) 0, Sz = in_seg ? pz : (mixed) 0;
                wave_scan3(Sx, Sy, Sz);
                const int prev = first > 0 ? first - 1 : 0;
                mixed Tx = Sx - shfl(Sx, prev), Ty = Sy - shfl(Sy, prev), Tz = Sz - shfl(Sz, prev);
                if (first == 0) { Tx = Sx; Ty = Sy; Tz = Sz; }
                mixed Tw = 0;
                if (F & A_KE_MOM) {      // the shuffle must run in ALL lanes: a lane reads its left neighbour segment's last lane
                    const mixed Sw = wave_scan(in_seg ? pw : (mixed) 0);
                    const mixed Lw = shfl(Sw, prev);
                    Tw = first > 0 ? Sw - Lw : Sw;
                }
                if (leader) {
                    mixed Vm = (mixed) seg_mw.x, Vw = (mixed) seg_mw.y;      // static: summed once on the host in particle order (vv_host.hpp: seg_mass)
                    if (a.slot_big && (meta & META_BIGMOL)) {     // molecule spread over several waves: totals from the accumulator
                        const unsigned long long* src = a.bigacc + 4 * (size_t) a.slot_big[(size_t) wave * 64 + lane];
                        Tx = (mixed) ((double) (long long) src[0] * a.big_inv_scale); Ty = (mixed) ((double) (long long) src[1] * a.big_inv_scale);
                        Tz = (mixed) ((double) (long long) src[2] * a.big_inv_scale); Vm = (mixed) ((double) (long long) src[3] * a.big_inv_scale);
                        Vw = P::RECIP(Vm);
                    }
                    const mixed Vx = Tx * Vw, Vy = Ty * Vw, Vz = Tz * Vw;        // V = P * RECIP(M), comVelm.w = RECIP(M)
                    const mixed4 cv = {Vx, Vy, Vz, Vw};
                    ((mixed4*) a.comv)[segi] = cv;                                // the reference's comVelm[id_mol], handed to kernel B
                    const bool counts = Vw != 0 && (!(meta & META_BIGMOL) || (meta & META_BIG_FIRST));
                    if (counts) k_com += (double) ((Vx * Vx + Vy * Vy + Vz * Vz) * Vm);
                    if (F & A_KE_MOM) {
                        const mixed Wx = Tw * Vw;                                // mass-weighted mean of cos(kz) over the molecule
                        a.comw[segi] = (double) Wx;
                        if (counts) { m_ab[1] += (double) (Vx * Wx * Vm); m_bb[1] += (double) (Wx * Wx * Vm); }
                    }
                }
            }
        }
    }
    VV_STAMP(threadIdx.x >> 6, 3);
    if (F & (A_KE | A_BIAS | A_KE_PLAIN)) {
        // group "atom" by difference (see the KE stage); with A_KE_PLAIN k_com and k_drude are zero
        const double vals[NUM_ACC] = {k_atom - k_drude - k_com, k_com, k_drude, k_bias, m_ab[0] - m_ab[2] - m_ab[1], m_ab[1], m_ab[2],
                                      m_bb[0] - m_bb[2] - m_bb[1], m_bb[1], m_bb[2]};
        const bool mom = (F & A_KE_MOM) != 0;
        const bool en[NUM_ACC] = {(F & (A_KE | A_KE_PLAIN)) != 0, (F & A_KE) != 0, (F & A_KE) != 0, (F & A_BIAS) != 0, mom, mom, mom, mom, mom, mom};
        block_accumulate<NUM_ACC>(vals, en, a.acc, a.acc_scale, a.status, a.acc_exclusive != 0, acc_old);
    }
    VV_STAMP(threadIdx.x >> 6, 4);
    VV_STAMP_DUMP(threadIdx.x >> 6);
    VV_SPAN_END;
}

// ================================================================================ NH chain
// VVIntegrator::propagateNHChain (openmmapi/src/VVIntegrator.cpp:340-376) on the device in double, so the
// reference's blocking download / upload pair (HOST:709-746) disappears.  Lane g advances temperature group g.
// exp(x) for the chain.  Its arguments are -dt/8*eta_dot and -dt/2*eta_dot: |x| << 1 in any sane run.  For
// |x| <= 2^-4 a degree-11 Taylor polynomial (Horner, 11 dependent FMAs) is exact to < 1 ulp (truncation
// x^12/12! < 1e-23 relative); anything larger goes to the library exp.
__device__ __forceinline__ double chain_exp(double x) {
    // degree-11 Taylor polynomial by Estrin's scheme (used by the stand-alone chain kernel and as the exact fallback of the
    // thermostat wave).  exp(x) = sum_{k<=11} x^k/k!, |x| <= 2^-4: truncation < 1e-23 relative.
    const double x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    const double p01 = fma(x, 1.0, 1.0), p23 = fma(x, 1.0 / 6.0, 0.5), p45 = fma(x, 1.0 / 120.0, 1.0 / 24.0);
    const double p67 = fma(x, 1.0 / 5040.0, 1.0 / 720.0), p89 = fma(x, 1.0 / 362880.0, 1.0 / 40320.0);
    const double pab = fma(x, 1.0 / 39916800.0, 1.0 / 3628800.0);
    const double q0 = fma(x2, p23, p01), q1 = fma(x2, p67, p45), q2 = fma(x2, pab, p89);
    double p = fma(x8, q2, fma(x4, q1, q0));
    if (__builtin_expect(__any(fabs(x) > 0.0625), 0)) p = fabs(x) > 0.0625 ? exp(x) : p;   // wave-uniform, practically never taken
    return p;
}
// The thermostat wave of kernel B is one serial dependency chain of fp64 operations (three lanes of one wave doing useful work) on
// the critical path of the whole kernel: ~700 cycles for a three-link chain when every exp is ONE polynomial evaluation
// (tools/probes/dpchain3_probe.cpp).  The arguments are -dt/8 eta_dot and -dt/2 eta_dot; with the Drude thermostat's 40/ps and
// dt = 1 fs the latter reaches 0.02-0.05 in ordinary runs.  Round 1 used a degree-7 polynomial (|x| <= 2^-6) with a complete re-run
// of the chain on the degree-11 one whenever an argument was larger -- which in the headline workload was EVERY step: the chain
// took 2 640 cycles (stamps, tools/probes/b_timeline.py).  Now: degree 11 in Estrin form (one level deeper than degree 7, five
// more operations), exact to < 1 ulp for |x| <= 2^-3 (truncation x^12/12! < 3e-20 relative); the caller keeps the largest biased
// exponent seen (two 32-bit operations per call) and redoes the step with the library exp only beyond that.
__device__ __forceinline__ double chain_exp_small(double x, unsigned& max_hi) {
    const unsigned hi = (unsigned) __double2hiint(x) & 0x7FFFFFFFu;
    max_hi = hi > max_hi ? hi : max_hi;
    const double x2 = x * x;
    const double p01 = x + 1.0, p23 = fma(x, 1.0 / 6.0, 0.5), p45 = fma(x, 1.0 / 120.0, 1.0 / 24.0), p67 = fma(x, 1.0 / 5040.0, 1.0 / 720.0);
    const double p89 = fma(x, 1.0 / 362880.0, 1.0 / 40320.0), pab = fma(x, 1.0 / 39916800.0, 1.0 / 3628800.0);
    const double x4 = x2 * x2, q0 = fma(x2, p23, p01), q1 = fma(x2, p67, p45), q2 = fma(x2, pab, p89);
    const double x8 = x4 * x4;
    return fma(x8, q2, fma(x4, q1, q0));
}
constexpr unsigned CHAIN_EXP_SMALL_HI = 0x3FC00000u;      // high word of 2^-3
__device__ __forceinline__ double chain_exp_wide(double x) { return exp(x); }

// CHAIN INVARIANT, stated once for both implementations below (propagate_regs: the stand-alone chain kernel; propagate_preloaded:
// kernel B's thermostat wave): eta_dot[NC], the element behind the chain's last link, is 0.  The reference sizes etaDot numChains + 1,
// initialises it to 0 and never writes the last element (API:340-376); here the plan's chain length is fixed at creation
// (vvhip_set_params keeps num_nh_chains), vvhip_bind starts both state copies from zeros and vvhip_set_nh_state zeroes everything from
// index NC on.  exp(-dt/8 * 0) is exactly 1, so BOTH implementations drop the two evaluations that take it as argument -- they cannot
// drift apart over it, whatever size regime or chain length selects between them.
// One temperature group, chain length NC known at compile time so the chain lives in registers.
// Differences from the host routine, both below 1 ulp per operation: chain_exp for exp, and multiplication by
// the reciprocal thermostat mass instead of a division.
template <int NC>
__device__ __forceinline__ double propagate_regs(const NHConst& c, int g, double ke2, const NHDevState* in, NHDevState* out) {
    double eta[NC], eta_dot[NC + 1], eta_dotdot[NC], eta_mass[NC], inv_mass[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) {
        eta[i] = in->s.eta[g][i]; eta_dot[i] = in->s.eta_dot[g][i]; eta_dotdot[i] = in->s.eta_dotdot[g][i];
        eta_mass[i] = c.eta_mass[g][i]; inv_mass[i] = c.inv_eta_mass[g][i];
    }
    eta_dot[NC] = in->s.eta_dot[g][NC];
    double factor = 1.0;
    if (g < c.num_tg && eta_mass[0] > 0) {                                       // HOST:729
        const double ke2_target = c.nkbt[g];
        double expfac = 1.0;
        const double dt2 = c.step_size / c.loops_per_step / 2;
        const double dt4 = dt2 / 2;
        const double dt8 = dt4 / 2;
        const double kT = ((1.380649e-23 * 6.02214076e23) / 1000.0) * c.temperature[g];
        eta_dotdot[0] = (ke2 - ke2_target) * inv_mass[0];
        for (int iloop = 0; iloop < c.loops_per_step; iloop++) {
#pragma unroll
            for (int ich = NC - 1; ich >= 0; ich--) {
                expfac = ich == NC - 1 ? 1.0 : chain_exp(-dt8 * eta_dot[ich + 1]);     // CHAIN INVARIANT (below): eta_dot[NC] == 0
                eta_dot[ich] *= expfac;
                eta_dot[ich] += eta_dotdot[ich] * dt4;
                eta_dot[ich] *= expfac;
            }
            factor *= chain_exp(-dt2 * eta_dot[0]);
#pragma unroll
            for (int ich = 0; ich < NC; ich++) eta[ich] += dt2 * eta_dot[ich];
            eta_dotdot[0] = (ke2 * factor * factor - ke2_target) * inv_mass[0];
            eta_dot[0] *= expfac;                                                // stale expfac on purpose (quirk Q10)
            eta_dot[0] += eta_dotdot[0] * dt4;
            eta_dot[0] *= expfac;
#pragma unroll
            for (int ich = 1; ich < NC; ich++) {
                expfac = ich == NC - 1 ? 1.0 : chain_exp(-dt8 * eta_dot[ich + 1]);
                eta_dot[ich] *= expfac;
                eta_dotdot[ich] = (eta_mass[ich - 1] * eta_dot[ich - 1] * eta_dot[ich - 1] - kT) * inv_mass[ich];
                eta_dot[ich] += eta_dotdot[ich] * dt4;
                eta_dot[ich] *= expfac;
            }
        }
    }
    if (out) {
#pragma unroll
        for (int i = 0; i < NC; i++) { out->s.eta[g][i] = eta[i]; out->s.eta_dot[g][i] = eta_dot[i]; out->s.eta_dotdot[g][i] = eta_dotdot[i]; }
        out->s.eta_dot[g][NC] = eta_dot[NC];
        out->s.ke2[g] = g < c.num_tg ? ke2 : in->s.ke2[g];
        out->s.vscale[g] = factor;
        out->scales[g] = factor;
    }
    return factor;
}

// The same update on state that the caller has already loaded (kernel B issues those loads at its very top so their
// latency overlaps the particle loads and the accumulator fold).  NC <= 4.
struct ChainRegs { double eta[4], eta_dot[5], eta_dotdot[4]; };
// `publish(factor)` is called ONCE, as soon as the scale factor of the step is final -- after the downward sweep and the exp of the
// LAST loop (API:352-358); what follows in that loop (API:360-374) only prepares the thermostat's state for the next application.
// Kernel B's thermostat wave releases the block's tile waves from inside `publish`, so the second half of the chain is off the
// kernel's critical path (the chain as a whole is ~0.85 us at the headline size, profiles/r02a_timeline_*).  publish returns false
// to abandon the evaluation (the caller then redoes it with the wide-range exp).
template <int NC, bool FAST, class Pub>
__device__ __forceinline__ double propagate_preloaded(const NHConst& c, const ChainLaneBlock& lc, double ke2, ChainRegs& r, unsigned& max_hi, Pub&& publish) {
    auto ex = [&](double x) { return FAST ? chain_exp_small(x, max_hi) : chain_exp_wide(x); };
    // eta_dot[NC] == 0 (CHAIN INVARIANT above): the two exps that take it as argument are dropped (two of the six serial ones for NC = 3)
    constexpr bool tail_zero = true;
    // Runs in the block's thermostat wave only (lanes 0..2 = the three temperature groups).  Lanes of a group that is not
    // thermostatted (HOST:729) run the same instructions on harmless values (their reciprocal masses are 0) so that `publish` is
    // reached by the whole wave; the caller discards what they computed.
    double factor = 1.0;
    const bool active = lc.active != 0;
    const double ke2_target = lc.nkbt;
    double expfac = 1.0;
    const double dt2 = lc.dt2, dt4 = lc.dt4, dt8 = lc.dt8, kT = lc.kT;        // computed once on the host exactly as API:343-347 does
    r.eta_dotdot[0] = (ke2 - ke2_target) * lc.inv_eta_mass[0];
    const int loops = c.loops_per_step;
    if (loops < 1) { publish(1.0); return 1.0; }
    for (int iloop = 0; iloop < loops; iloop++) {
#pragma unroll
        for (int ich = NC - 1; ich >= 0; ich--) {
            expfac = (tail_zero && ich == NC - 1) ? 1.0 : ex(-dt8 * r.eta_dot[ich + 1]);
            r.eta_dot[ich] *= expfac;
            r.eta_dot[ich] += r.eta_dotdot[ich] * dt4;
            r.eta_dot[ich] *= expfac;
        }
        factor *= ex(-dt2 * r.eta_dot[0]);
        if (iloop == loops - 1 && !publish(active ? factor : 1.0)) return factor;
#pragma unroll
        for (int ich = 0; ich < NC; ich++) r.eta[ich] += dt2 * r.eta_dot[ich];
        r.eta_dotdot[0] = (ke2 * factor * factor - ke2_target) * lc.inv_eta_mass[0];
        r.eta_dot[0] *= expfac;                                                  // stale expfac on purpose (quirk Q10)
        r.eta_dot[0] += r.eta_dotdot[0] * dt4;
        r.eta_dot[0] *= expfac;
#pragma unroll
        for (int ich = 1; ich < NC; ich++) {
            expfac = (tail_zero && ich == NC - 1) ? 1.0 : ex(-dt8 * r.eta_dot[ich + 1]);
            r.eta_dot[ich] *= expfac;
            r.eta_dotdot[ich] = (lc.eta_mass[ich - 1] * r.eta_dot[ich - 1] * r.eta_dot[ich - 1] - kT) * lc.inv_eta_mass[ich];
            r.eta_dot[ich] += r.eta_dotdot[ich] * dt4;
            r.eta_dot[ich] *= expfac;
        }
    }
    return factor;
}

// Kernel B only inlines chain lengths up to 4 (register budget: 8 variants would cost half the occupancy);
// longer chains take the stand-alone chain launch (vv_api.cpp decides).
template <int NC, class Pub>
__device__ __forceinline__ double propagate_small_nc(const NHConst& c, const ChainLaneBlock& lc, double ke2, ChainRegs& r, Pub&& pub) {
    const ChainRegs saved = r;
    unsigned max_hi = 0;
    bool published = false;
    // fast evaluation; the factor is only published if every exp argument so far was inside the polynomial's range
    double f = propagate_preloaded<NC, true>(c, lc, ke2, r, max_hi, [&](double fac) {
        if (__builtin_expect(__any(max_hi > CHAIN_EXP_SMALL_HI), 0)) return false;
        pub(fac);
        published = true;
        return true;
    });
    if (__builtin_expect(__any(max_hi > CHAIN_EXP_SMALL_HI), 0)) {          // an exp argument beyond 2^-3: redo with the library exp
        r = saved;
        const bool fast_factor_out = published;      // then every argument it depends on was in range: it stays the step's factor, only the state is redone
        const double f2 = propagate_preloaded<NC, false>(c, lc, ke2, r, max_hi, [&](double fac) { if (!published) { pub(fac); published = true; } return true; });
        if (!fast_factor_out) f = f2;
    }
    if (lc.active == 0) { r = saved; f = 1.0; }                             // HOST:729: a group without thermostat keeps its state, factor 1
    return f;
}
// NCT = 3: the chain length is known when the kernel is compiled (the specialised kernels are built for the integrator's default
// of three, VVIntegrator.h:62; launch_b sends other lengths to the generic kernel).  Four chain bodies less in a kernel whose code is
// fetched cold at every launch: kernel B 5.92 -> 5.74 us.
template <int NCT, class Pub>
__device__ __forceinline__ double propagate_group_small(const NHConst& c, const ChainLaneBlock& lc, double ke2, ChainRegs& r, Pub&& pub) {
    if (NCT == 3) return propagate_small_nc<3>(c, lc, ke2, r, pub);
    switch (c.num_chains) {
        case 1: return propagate_small_nc<1>(c, lc, ke2, r, pub);
        case 2: return propagate_small_nc<2>(c, lc, ke2, r, pub);
        case 3: return propagate_small_nc<3>(c, lc, ke2, r, pub);
        default: return propagate_small_nc<4>(c, lc, ke2, r, pub);
    }
}
__device__ __forceinline__ double propagate_group(const NHConst& c, int g, double ke2, const NHDevState* in, NHDevState* out) {
    switch (c.num_chains) {
        case 1: return propagate_regs<1>(c, g, ke2, in, out);
        case 2: return propagate_regs<2>(c, g, ke2, in, out);
        case 3: return propagate_regs<3>(c, g, ke2, in, out);
        case 4: return propagate_regs<4>(c, g, ke2, in, out);
        case 5: return propagate_regs<5>(c, g, ke2, in, out);
        case 6: return propagate_regs<6>(c, g, ke2, in, out);
        case 7: return propagate_regs<7>(c, g, ke2, in, out);
        default: return propagate_regs<8>(c, g, ke2, in, out);
    }
}

// Stand-alone chain launch (one wave), used by the kernel-interface-level entry points that stop between
// the reduction and its consumer (vvhip_calc_velocity_bias) and by tests.  Sole reader of the accumulators,
// so it re-zeroes what it consumed.
__global__ void __launch_bounds__(64) vv_kernel_chain(const NHConst c, NHDevState* st, unsigned long long* acc) {
    const int g = threadIdx.x;
    long long tot[NUM_ACC];
#pragma unroll
    for (int k = 0; k < NUM_ACC; k++) tot[k] = acc_total(acc, k, g);
    double sum = 0;
#pragma unroll
    for (int k = 0; k < NUM_ACC; k++)
        if (g == k) sum = (double) tot[k] * c.acc_inv_scale[k];
    __syncthreads();
    for (int j = 0; j < ACC_SLOTS / 64; j++) {
        if (c.flags & C_CHAIN) { acc[0 * ACC_SLOTS + g + 64 * j] = 0; acc[1 * ACC_SLOTS + g + 64 * j] = 0; acc[2 * ACC_SLOTS + g + 64 * j] = 0; }
        if (c.flags & C_BIAS) acc[3 * ACC_SLOTS + g + 64 * j] = 0;
    }
    if ((c.flags & C_CHAIN) && g < VVHIP_NUM_TG) propagate_group(c, g, sum, st, st);
    if ((c.flags & C_BIAS) && g == 3) {                                          // K/cosineAccelerate.cu:57-59
        st->s.v_bias = sum * c.inv_mass_total;
        st->scales[3] = sum * c.inv_mass_total;
    }
}

// ================================================================================ kernel B

template <class real, class mixed, uint32_t SF>
__global__ void __launch_bounds__(512) vv_kernel_b(const int2* __restrict__ pre_slots, const int pre_nwaves, const int pre_wpb, const unsigned long long* __restrict__ pre_acc,
                                                   const NHDevState* __restrict__ pre_nh, const ChainLaneBlock* __restrict__ pre_lane_const, const int* __restrict__ pre_seg_base,
                                                   const KArgs a) {
    using real4 = typename Vec<real>::v4;
    using real3 = typename Vec<real>::v3;
    using mixed4 = typename Vec<mixed>::v4;
    using P = Prec<real>;
    using IO = PosIO<real, mixed>;
    const int lane = threadIdx.x & 63;
    const uint32_t F = SF ? SF : a.flags;
    // Block layout.  With B_CHAIN the FIRST wave of every block is the block's thermostat wave: it folds the
    // accumulators and advances the NH chain (a ~2 us serial fp64 dependency chain) while the other waves of the
    // block load their particles and do the scale-independent preparation; one barrier joins them.  Without this,
    // every tile wave pays the chain on its own critical path (measured: 9.7 -> see DESIGN.md §7).
    const int nwb = pre_wpb;                      // = blockDim.x >> 6, preloaded (blockDim is a hidden kernel argument: an s_load)
    const bool has_cw = (F & B_CHAIN) != 0;
    // the thermostat wave is wave 0: the waves of a block start in order, and the block's critical path runs through this one
    const bool chain_wave = has_cw && (threadIdx.x >> 6) == 0;
    const int wib = (int) (threadIdx.x >> 6) - (has_cw ? 1 : 0);      // index among the block's tile waves
    const int tiles_per_block = has_cw ? nwb - 1 : nwb;
    __shared__ double sh_scales[4];
    double sc0 = 1.0, sc1 = 1.0, sc2 = 1.0, scb = 0.0;
    // Periodic layout: role words and pair mass fractions of the regions' pattern waves, copied into LDS once per block (kernel A)
    __shared__ unsigned sh_pat_meta[4][64];
    __shared__ double sh_pat_f[4][64];
    __shared__ unsigned sh_pat_shake[4][64];        // constraint cluster words / parameters of the pattern waves (B_SHAKE)
    __shared__ float4 sh_pat_prm[4][64];
    if (F & B_PERIODIC) {
        for (int row = threadIdx.x >> 6; row < 4; row += nwb) {
            const int ws = a.per.wave_start[row];
            unsigned m = 0;
            double f = 0;
            if (ws != 0x7fffffff) {
                m = (unsigned) pre_slots[(size_t) ws * 64 + lane].y;
                if ((F & B_MTAB) && (F & B_SCALE)) f = a.slot_f[(size_t) ws * 64 + lane];
            }
            sh_pat_meta[row][lane] = m;
            sh_pat_f[row][lane] = f;
            if (F & B_CONS) {
                const bool member = ws != 0x7fffffff && (m & META_SHAKE);
                sh_pat_shake[row][lane] = member ? (unsigned) a.slot_shake[(size_t) ws * 64 + lane] : 0u;
                sh_pat_prm[row][lane] = member ? a.slot_shake_param[(size_t) ws * 64 + lane] : make_float4(0, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---------------- thermostat wave: scale factors for the whole block, then done
    VV_SPAN_BEGIN;
    if (chain_wave) {
        __builtin_amdgcn_s_setprio(3);            // the block waits for this wave: let it win the issue arbitration on its SIMD
        VV_STAMP(7, 0);
        const int cg = lane < VVHIP_NUM_TG ? lane : VVHIP_NUM_TG - 1;
        // every load of this wave -- thermostat state, chain constants, the accumulator slots of all rows in use -- is issued here, in
        // front of a scheduling barrier: left to itself the backend sank part of the state loads BEHIND the first row's reduction
        // (register pressure), i.e. behind a wait for the cold accumulator loads, and the chain started a second memory round trip late
        long long raw[NUM_ACC][ACC_SLOTS / 64];
#pragma unroll
        for (int k = 0; k < NUM_ACC; k++) {
            const bool wanted = (k < 3 || (k == 3 && (F & B_UNBIAS)) || (k > 3 && (F & B_KE_MOM)));
#pragma unroll
            for (int j = 0; j < ACC_SLOTS / 64; j++) raw[k][j] = wanted ? (long long) pre_acc[k * ACC_SLOTS + lane + 64 * j] : 0ll;
        }
        ChainRegs cr;
#pragma unroll
        for (int i = 0; i < 4; i++) { cr.eta[i] = pre_nh->s.eta[cg][i]; cr.eta_dot[i] = pre_nh->s.eta_dot[cg][i]; cr.eta_dotdot[i] = pre_nh->s.eta_dotdot[cg][i]; }
        cr.eta_dot[4] = pre_nh->s.eta_dot[cg][4];
        const ChainLaneBlock lc = pre_lane_const[cg];
        const double bias_carried = pre_nh->scales[3];
        __builtin_amdgcn_sched_barrier(0);
        long long tot[NUM_ACC];
#pragma unroll
        for (int k = 0; k < NUM_ACC; k++) {
            const bool wanted = (k < 3 || (k == 3 && (F & B_UNBIAS)) || (k > 3 && (F & B_KE_MOM)));
            tot[k] = wanted ? acc_reduce(raw[k]) : 0ll;
        }
        if (F & B_MAILBOX) {                      // multi-GPU: block 0 publishes this rank's totals, every block collects all ranks'
            __shared__ unsigned int mb_words[MB_MAX_RANKS * MB_WORDS];
            mailbox_exchange(a, lane, a.nh->mb_seq + 1u, mb_words, tot);
        }
        VV_STAMP(7, 1);
        double ke2 = 0;
#pragma unroll
        for (int k = 0; k < VVHIP_NUM_TG; k++)
            if (cg == k) ke2 = (double) tot[k] * lc.acc_inv_scale;
        if (F & B_KE_MOM) {                       // 2KE of the bias-free velocities from the moments of the biased ones
            const double V = (double) tot[3] * a.chain.acc_inv_scale[3] * a.chain.inv_mass_total;
            double sab = 0, sbb = 0;
#pragma unroll
            for (int k = 0; k < VVHIP_NUM_TG; k++)
                if (cg == k) { sab = (double) tot[4 + k] * a.chain.acc_inv_scale[4]; sbb = (double) tot[7 + k] * a.chain.acc_inv_scale[7]; }
            ke2 = ke2 - 2.0 * V * sab + V * V * sbb;
        }
        double factor = 1.0;
        // the bias is requested before the chain starts (when it is carried over it is a load from the state)
        const double bias = (F & B_UNBIAS) ? (double) tot[3] * a.chain.acc_inv_scale[3] * a.chain.inv_mass_total   // K/cosineAccelerate.cu:57-59
                                           : bias_carried;                                                    // carried over unchanged
        VV_STAMP_AFTER(7, 4, ke2);
        // Hands the scale factors to the block's tile waves the moment they are final -- in the MIDDLE of the chain update (see
        // propagate_preloaded): this wave's only barrier.  The second half of the chain runs while the tile waves scale and drift.
        auto release_tiles = [&](double f) {
            if (lane < VVHIP_NUM_TG) sh_scales[lane] = f;
            if (lane == 3) sh_scales[3] = bias;
            VV_STAMP_AFTER(7, 5, f);
            __syncthreads();
        };
        factor = propagate_group_small<(SF != 0 ? 3 : 0)>(a.chain, lc, ke2, cr, release_tiles);
        VV_STAMP_AFTER(7, 2, factor);
        if (blockIdx.x == 0) {                    // one block records the advanced thermostat and clears the idle accumulator copy
            NHDevState* out = a.nh_next;
            if (lane < VVHIP_NUM_TG) {
#pragma unroll
                for (int i = 0; i < 4; i++) { out->s.eta[cg][i] = cr.eta[i]; out->s.eta_dot[cg][i] = cr.eta_dot[i]; out->s.eta_dotdot[cg][i] = cr.eta_dotdot[i]; }
                out->s.eta_dot[cg][4] = cr.eta_dot[4];
                out->s.ke2[cg] = cg < a.chain.num_tg ? ke2 : a.nh->s.ke2[cg];
                out->s.vscale[cg] = factor;
                out->scales[cg] = factor;
            }
            if (lane == 3) { out->s.v_bias = bias; out->scales[3] = bias; }
            if (lane == 4) out->mb_seq = a.nh->mb_seq + ((F & B_MAILBOX) ? 1u : 0u);
            for (int i = lane; i < a.acc_rows * ACC_SLOTS; i += 64) a.acc_next[i] = 0;
        }
        VV_STAMP(7, 3);
        VV_STAMP_DUMP(7);
        VV_SPAN_END;
        return;
    }

    // ---------------- tile waves: grid-stride over 64-lane tiles.  The grid is capped (launch_b), so at large N every block
    // pays the fold + chain once and then streams many tiles; the first tile's loads overlap the thermostat wave.
    bool need_scales = true;
    VV_STAMP(wib, 0);
    for (int wave = blockIdx.x * tiles_per_block + wib; need_scales || wave < pre_nwaves; wave += gridDim.x * tiles_per_block) {
        const bool valid = wave < pre_nwaves;
        int atom = -1;
        unsigned meta = 0;
        PeriodicWave pw = {0, 0, 0, 0};
        mixed4* velm = (mixed4*) a.velm;
        mixed4 v = {0, 0, 0, 0};
        int segb = 0;
        if (F & B_PERIODIC) {
            if (valid) {          // (uniform) particle index from the wave index: the particle loads do not wait for a slot word
                pw = periodic_wave(a.per, wave);
                const bool in = lane < pw.count;
                atom = in ? pw.atom0 + lane : -1;
                meta = in ? sh_pat_meta[pw.region][lane] : 0u;
                segb = pw.seg0;
            }
        } else if (valid) {
            // (pre_seg_base: a preloaded argument; requested FIRST, it returns with the slot word)
            if (F & B_SCALE) segb = pre_seg_base[__builtin_amdgcn_readfirstlane(wave)];
            const int2 slot = pre_slots[(size_t) wave * 64 + lane]; atom = slot.x; meta = (unsigned) slot.y;
        }
        const unsigned role = meta & META_ROLE_MASK;
        const int partner = (meta >> META_PARTNER_SHIFT) & 63;
        const bool act = atom >= 0;
        const mixed stepSize = (mixed) a.dt;
        const bool touches_pos = F & (B_DRIFT_MIDDLE | B_POS3 | B_VV_POS | B_VV_KICK | B_HARDWALL | B_IMAGE | B_UNBIAS | B_BIAS_REMOVE | B_BIAS_RESTORE);
        // ---- every load of the tile that needs nothing but the slot word (or nothing at all, with the arithmetic layout), in ONE
        // batch of unconditional loads from clamped indices closed by a scheduling barrier (see kernel A): velocity, position and
        // correction, force, the molecule's COM velocity, the pair's mass fraction, the constraint cluster words.  Six dependent
        // memory round trips per tile became two.
        const int ai = act ? atom : 0;
        const size_t li = valid ? (size_t) wave * 64 + lane : (size_t) lane;
        v = velm[ai];
        real4 p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0};
        if (touches_pos) {
            p1 = ((const real4*) a.posq)[ai];
            if (IO::kMixed) p2 = ((const real4*) a.corr)[ai];
        }
        long long kfx = 0, kfy = 0, kfz = 0;
        if (F & B_KICK) { kfx = a.force[ai]; kfy = a.force[ai + a.padded]; kfz = a.force[ai + 2 * a.padded]; }
        const unsigned long long leaders = (F & B_SCALE) ? __ballot((meta & META_COM_LEADER) != 0) : 0ull;     // in every lane: a wave-wide vote
        // COM velocity of this lane's molecule as kernel A's KE stage left it (of the bias-free velocities when a bias is
        // removed): one 32-byte entry per molecule, the same address for every lane of the segment (lanes behind the wave's last
        // leader read the tables' spare entry)
        mixed4 cv = {0, 0, 0, 0};
        double cw_raw = 0;
        if (F & B_SCALE) {
            const int segi = segb + (int) lanes_below(leaders);
            cv = ((const mixed4*) a.comv)[segi];
            if (F & B_KE_MOM) cw_raw = a.comw[segi];
        }
        // B_MTAB: the pair's mass fractions are static (vv_kernel_mass_table formed them with the operations of K/drudeNoseHoover.cu:173-180
        // on the same inverse masses, so they are the per-step values bit for bit): one 8-byte load per lane, requested with the
        // particle data, instead of two IEEE fp64 divisions per pair lane and step
        mixed tab_f = 0;
        if ((F & B_MTAB) && (F & B_SCALE)) tab_f = (F & B_PERIODIC) ? (mixed) sh_pat_f[pw.region][lane] : (mixed) a.slot_f[li];
        // cluster word and parameters of the in-kernel constraints
        unsigned shake_word = 0;
        float4 shake_prm = make_float4(0, 0, 0, 0);
        if (F & B_CONS) {
            shake_word = (F & B_PERIODIC) ? sh_pat_shake[pw.region][lane] : (unsigned) a.slot_shake[li];
            shake_prm = (F & B_PERIODIC) ? sh_pat_prm[pw.region][lane] : a.slot_shake_param[li];
        }
        // cos(2 pi z / Lz) of this lane, cached by kernel A (A_CZ_STORE)
        double cz_early = 0;
        if (F & B_CZ_LOAD) cz_early = a.cosz[li];
        int img = -1;
        if (F & B_IMAGE) img = a.slot_image[li];
        __builtin_amdgcn_sched_barrier(0);
        if (!act) v = mixed4{0, 0, 0, 0};
        if (!(act && (meta & META_SHAKE))) shake_word = 0;
        const bool massive = act && v.w != 0;
        mixed x = 0, y = 0, z = 0, q = 0;
        real zraw = 0;
        if (act && touches_pos) {          // K/middle.cu:81-96: positions are posq (+ posqCorrection in mixed mode)
            zraw = p1.z;
            if (IO::kMixed) { x = p1.x + (mixed) p2.x; y = p1.y + (mixed) p2.y; z = p1.z + (mixed) p2.z; q = p1.w; }
            else { x = p1.x; y = p1.y; z = p1.z; q = p1.w; }
        }
        // image particle of this lane's particle: its present content (the charge and the correction's w survive the mirror update)
        // is requested now, so that nothing has to be read after the position store at the end of the tile
        real4 img_p = {0, 0, 0, 0}, img_c = {0, 0, 0, 0};
        if (!((F & B_IMAGE) && act && (meta & META_HAS_IMAGE))) img = -1;
        if ((F & B_IMAGE) && img >= 0) {
            img_p = ((const real4*) a.posq)[img];
            if (IO::kMixed) img_c = ((const real4*) a.corr)[img];
        }
        // B_KICK: kernel A kept its kicked velocities in registers (A_NOSTORE); the same kick again here, from the same velm and force
        // bits with the same expression (K/middle.cu:11-21; forceExtra is zero on this path), gives the same velocities bit for bit.
        // (a massless particle's force is read and not used)
        if ((F & B_KICK) && act) {
            const long long fx = kfx, fy = kfy, fz = kfz;
            real3 fe = {0, 0, 0};
            if (F & B_UNBIAS) fe.x += (real) a.cos_accel * cz_early * P::RECIP(v.w);      // K/cosineAccelerate.cu:9, kernel A's A_COS term to the bit
            const mixed fscale = stepSize / (mixed) 0x100000000;
            if (v.w != 0) {          // K/middle.cu:11: massive particles only
                v.x += stepSize * v.w * fe.x + fscale * v.w * fx;
                v.y += stepSize * v.w * fe.y + fscale * v.w * fy;
                v.z += stepSize * v.w * fe.z + fscale * v.w * fz;
            }
        }
        const mixed4 v_old = v;          // velocity after the kick, before the thermostat (Pos1 uses it)
        bool vel_dirty = (F & B_KICK) != 0, pos_dirty = false;

        const bool nh = role == ROLE_NH_NORMAL || role == ROLE_NH_DRUDE || role == ROLE_NH_PARENT;
        const bool use_com = ((meta >> META_SEGFIRST_SHIFT) & 63) != ((meta >> META_SEGLAST_SHIFT) & 63) || (meta & META_COM_LEADER);
        mixed Vx = 0, Vy = 0, Vz = 0, Vw = 0, com_w = 0;
        if ((F & B_SCALE) && nh && use_com) {
            Vx = cv.x; Vy = cv.y; Vz = cv.z; Vw = cv.w;
            if (F & B_KE_MOM) com_w = (mixed) cw_raw;
        }

        // Factor-independent half of the scaling: velocities relative to the molecular COM, the Drude partner's over the shuffle
        // network, mass fractions, COM / relative split of the pair.  Without a bias to remove first it runs here, i.e. while
        // the tile waves of the first iteration wait for the thermostat wave.
        mixed ux = 0, uy = 0, uz = 0, cmx = 0, cmy = 0, cmz = 0, rx = 0, ry = 0, rz = 0, mass1fract = 0, mass2fract = 0;
        if (!(act && (role == ROLE_NH_DRUDE || role == ROLE_NH_PARENT))) tab_f = 0;
        auto scale_prep = [&]() {
            ux = v.x; uy = v.y; uz = v.z;
            if (nh) { ux -= Vx; uy -= Vy; uz -= Vz; }
            // each lane forms the reciprocal of its own inverse mass, the partner's arrives by shuffle: the same IEEE quotients as
            // RECIP(a1w), RECIP(a2w) of K/drudeNoseHoover.cu:173-174 with one division per lane instead of two
            const bool pair_lane = role == ROLE_NH_DRUDE || role == ROLE_NH_PARENT;
            const mixed own_m = (F & B_MTAB) ? tab_f : (pair_lane ? P::RECIP(v.w) : (mixed) 0);     // with the table: the own FRACTION travels
            const mixed px = shfl(ux, partner), py = shfl(uy, partner), pz = shfl(uz, partner), pm = shfl(own_m, partner);
            if (pair_lane) {
                const bool isd = role == ROLE_NH_DRUDE;      // velAtom1 = Drude (pair.x), velAtom2 = parent
                const mixed a1x = isd ? ux : px, a1y = isd ? uy : py, a1z = isd ? uz : pz;
                const mixed a2x = isd ? px : ux, a2y = isd ? py : uy, a2z = isd ? pz : uz;
                if (F & B_MTAB) {
                    mass1fract = isd ? own_m : pm; mass2fract = isd ? pm : own_m;
                } else {
                    const mixed mass1 = isd ? own_m : pm, mass2 = isd ? pm : own_m;
                    const mixed invTotalMass = P::RECIP(mass1 + mass2);
                    mass1fract = invTotalMass * mass1; mass2fract = invTotalMass * mass2;
                }
                cmx = a1x * mass1fract + a2x * mass2fract;
                cmy = a1y * mass1fract + a2y * mass2fract;
                cmz = a1z * mass1fract + a2z * mass2fract;
                rx = a2x - a1x; ry = a2y - a1y; rz = a2z - a1z;
            }
        };
        const bool prep_early = (F & B_SCALE) && !(F & (B_UNBIAS | B_BIAS_REMOVE));
        VV_STAMP(wib, 1);
        if (prep_early) scale_prep();
        VV_STAMP(wib, 2);

        if (need_scales) {
            need_scales = false;
            if (has_cw) {
                __syncthreads();
                sc0 = sh_scales[0]; sc1 = sh_scales[1]; sc2 = sh_scales[2]; scb = sh_scales[3];
                VV_STAMP(wib, 3);
            } else if (F & (B_SCALE | B_UNBIAS | B_BIAS_REMOVE | B_BIAS_RESTORE)) {
                sc0 = a.nh->scales[0]; sc1 = a.nh->scales[1]; sc2 = a.nh->scales[2]; scb = a.nh->scales[3];
            }
            if (!valid) break;
        }

        // ---------------- bias removal (K/cosineAccelerate.cu:63-73); cos uses posq.z (real), all particles
        double cz = 0;
        mixed Vb = 0;
        if (F & (B_UNBIAS | B_BIAS_REMOVE | B_BIAS_RESTORE)) {
            Vb = (mixed) scb;
            cz = (F & B_CZ_LOAD) ? cz_early : cos_kz<real>(zraw, (real) a.inv_box_z);
            if ((F & (B_UNBIAS | B_BIAS_REMOVE)) && act) { v.x -= Vb * cz; vel_dirty = true; }
            if (F & B_KE_MOM) Vx -= Vb * com_w;       // kernel A stored the COM velocity of the biased velocities: COM(u) = COM(v) - V COM(w)
        }

        // ---------------- NH velocity scaling (K/drudeNoseHoover.cu:157-209): the factor-independent half was prepared above
        if (F & B_SCALE) {
            if (!prep_early) scale_prep();
            const mixed vscaleAtom = (mixed) sc0, vscaleCOM = (mixed) sc1, vscaleDrude = (mixed) sc2;
            if (role == ROLE_NH_NORMAL) {
                if (massive) {
                    v.x = vscaleAtom * ux + vscaleCOM * Vx;
                    v.y = vscaleAtom * uy + vscaleCOM * Vy;
                    v.z = vscaleAtom * uz + vscaleCOM * Vz;
                    vel_dirty = true;
                }
            } else if (role == ROLE_NH_DRUDE || role == ROLE_NH_PARENT) {
                const mixed sx = vscaleAtom * cmx, sy = vscaleAtom * cmy, sz = vscaleAtom * cmz;
                const mixed tx = vscaleDrude * rx, ty = vscaleDrude * ry, tz = vscaleDrude * rz;
                if (role == ROLE_NH_DRUDE) {
                    v.x = sx - tx * mass2fract + vscaleCOM * Vx;
                    v.y = sy - ty * mass2fract + vscaleCOM * Vy;
                    v.z = sz - tz * mass2fract + vscaleCOM * Vz;
                } else {
                    v.x = sx + tx * mass1fract + vscaleCOM * Vx;
                    v.y = sy + ty * mass1fract + vscaleCOM * Vy;
                    v.z = sz + tz * mass1fract + vscaleCOM * Vz;
                }
                vel_dirty = true;
            }
        }
        if ((F & (B_UNBIAS | B_BIAS_RESTORE)) && act) { v.x += Vb * cz; vel_dirty = true; }   // K/cosineAccelerate.cu:76-85

        // ---------------- classic VV first half: kick from the stored extra force, then posDelta (K/velocityVerlet.cu:6-29)
        mixed dx = 0, dy = 0, dz = 0;
        if ((F & B_VV_KICK) && massive) {
            const real3 fe = ((const real3*) a.fextra)[atom];
            const long long fx = a.force[atom], fy = a.force[atom + a.padded], fz = a.force[atom + 2 * a.padded];
            const mixed fscale = (mixed) a.fscale_vv;
            v.x += 0.5 * stepSize * v.w * fe.x + fscale * v.w * fx;
            v.y += 0.5 * stepSize * v.w * fe.y + fscale * v.w * fy;
            v.z += 0.5 * stepSize * v.w * fe.z + fscale * v.w * fz;
            dx = stepSize * v.x; dy = stepSize * v.y; dz = stepSize * v.z;
            vel_dirty = true;
        }
        // ---------------- position updates
        if ((F & B_POS2) && massive) {                                          // K/middle.cu:51-58
            const mixed halfdt = 0.5f * stepSize;
            mixed4 d = {halfdt * v.x, halfdt * v.y, halfdt * v.z, 0};
            mixed4 pd = ((mixed4*) a.pos_delta)[atom], od = ((mixed4*) a.old_delta)[atom];
            pd.x += d.x; pd.y += d.y; pd.z += d.z; pd.w += d.w;
            od.x += d.x; od.y += d.y; od.z += d.z; od.w += d.w;
            ((mixed4*) a.pos_delta)[atom] = pd;
            ((mixed4*) a.old_delta)[atom] = od;
        }
        // per-wave LDS page of the in-kernel SHAKE (collective over the wave: every lane walks through it)
        extern __shared__ double vv_dyn_lds[];
        mixed* shake_page_b = (mixed*) vv_dyn_lds + wib * (7 * 64);      // one page per tile wave, sized at launch
        if (F & B_DRIFT_MIDDLE) {
            // Pos1 (K/middle.cu:36-38) with the pre-thermostat velocity, Pos2 (:54-56) with the scaled one.  Without constraints
            // posDelta == oldDelta and Pos3's velocity correction (K/middle.cu:77-79) adds (d - d)/dt == 0 exactly; with the
            // in-kernel SHAKE the constrained displacement differs and the correction is the constraint force's kick.
            const mixed halfdt = 0.5f * stepSize;
            mixed ddx = 0, ddy = 0, ddz = 0;
            if (massive) {
                ddx = halfdt * v_old.x; ddy = halfdt * v_old.y; ddz = halfdt * v_old.z;
                ddx += halfdt * v.x; ddy += halfdt * v.y; ddz += halfdt * v.z;
            }
            const mixed odx = ddx, ody = ddy, odz = ddz;
            if (F & B_CONS)                                                        // integration.applyConstraints(tol), HOST:176
                shake_positions<mixed>(lane, shake_word, shake_prm, (mixed) a.shake_tol, x, y, z, v.w, ddx, ddy, ddz, shake_page_b,
                                       (F & B_SHAKE) != 0, (F & B_SETTLE) != 0, (F & B_SHAKE_GS) != 0);
            if (massive) {
                const mixed invDt = (mixed) a.inv_dt_mixed;      // = 1 / stepSize, formed once on the host
                v.x += (ddx - odx) * invDt; v.y += (ddy - ody) * invDt; v.z += (ddz - odz) * invDt;
                x += ddx; y += ddy; z += ddz;
                pos_dirty = true; vel_dirty = true;
            }
        }
        if ((F & B_POS3) && massive) {                                          // K/middle.cu:70-96
            const mixed invDt = (mixed) a.inv_dt_mixed;      // = 1 / stepSize, formed once on the host
            const mixed4 d = ((const mixed4*) a.pos_delta)[atom], od = ((const mixed4*) a.old_delta)[atom];
            v.x += (d.x - od.x) * invDt; v.y += (d.y - od.y) * invDt; v.z += (d.z - od.z) * invDt;
            x += d.x; y += d.y; z += d.z;
            pos_dirty = true; vel_dirty = true;
        }
        if (F & (B_VV_POS | B_VV_KICK)) {                                       // K/velocityVerlet.cu:41-66
            if (!(F & B_VV_KICK) && massive) {
                const mixed4 d = ((const mixed4*) a.pos_delta)[atom];
                dx = d.x; dy = d.y; dz = d.z;
            }
            if (F & B_CONS)                                                        // integration.applyConstraints(tol), HOST:351
                shake_positions<mixed>(lane, shake_word, shake_prm, (mixed) a.shake_tol, x, y, z, v.w, dx, dy, dz, shake_page_b,
                                       (F & B_SHAKE) != 0, (F & B_SETTLE) != 0, (F & B_SHAKE_GS) != 0);
            if (massive) {
                const mixed invStepSize = (mixed) a.inv_dt_double; // = 1.0 / stepSize, formed once on the host
                x += dx; y += dy; z += dz;
                v.x = (mixed) (invStepSize * dx); v.y = (mixed) (invStepSize * dy); v.z = (mixed) (invStepSize * dz);
                pos_dirty = true; vel_dirty = true;
            }
        }

        // ---------------- hard wall on Drude pairs (K/middle.cu:106-221); pair.x = Drude = "1", parent = "2"
        if (F & B_HARDWALL) {
            const mixed ox = shfl(x, partner), oy = shfl(y, partner), oz = shfl(z, partner);
            const mixed ovw = shfl(v.w, partner);
            if (act && (meta & META_PAIR)) {
                const bool isd = (meta & META_IS_DRUDE) != 0;
                const mixed maxDrudeDistance = (mixed) a.max_drude, hardwallscaleDrude = (mixed) a.hw_scale;
                mixed p1x = isd ? x : ox, p1y = isd ? y : oy, p1z = isd ? z : oz;
                mixed p2x = isd ? ox : x, p2y = isd ? oy : y, p2z = isd ? oz : z;
                const mixed vel1w = isd ? v.w : ovw, vel2w = isd ? ovw : v.w;
                const mixed deltax = p1x - p2x, deltay = p1y - p2y, deltaz = p1z - p2z;
                const mixed r2 = deltax * deltax + deltay * deltay + deltaz * deltaz;
                // The reference decides on rInv * maxDrudeDistance < 1 with r = SQRT(r2), rInv = RECIP(r) (K/middle.cu:126-131): a square
                // root and an IEEE division per pair and step.  r2 <= (0.9999 max)^2 implies r <= 0.99999 max (SQRT is a float sqrt in
                // mixed mode: relative error 1.2e-7), which implies rInv * max >= 1.00001 (1 - 2^-23)^2 > 1 in every mode, i.e. "no hit"
                // without forming either; only pairs within 1e-4 of the wall or beyond it take the exact test.  Same decisions, same bits.
                mixed r = 0, rInv = 0;
                bool hit = false;
                const mixed nearWall = maxDrudeDistance * (mixed) 0.9999;
                if (__builtin_expect(r2 > nearWall * nearWall, 0)) {
                    r = P::SQRT(r2);
                    rInv = P::RECIP(r);
                    hit = rInv * maxDrudeDistance < 1;
                }
                if (__builtin_expect(hit, 0)) {      // rare: keep the hit path out of the fall-through code
                    // both lanes of a pair see the same r, so both are in here: the partner's velocity is fetched only now
                    const mixed ovx = shfl(v.x, partner), ovy = shfl(v.y, partner), ovz = shfl(v.z, partner);
                    mixed vel1x = isd ? v.x : ovx, vel1y = isd ? v.y : ovy, vel1z = isd ? v.z : ovz;
                    mixed vel2x = isd ? ovx : v.x, vel2y = isd ? ovy : v.y, vel2z = isd ? ovz : v.z;
                    const mixed bx = deltax * rInv, by = deltay * rInv, bz = deltaz * rInv;
                    const mixed mass1 = P::RECIP(vel1w), mass2 = P::RECIP(vel2w);
                    const mixed deltaR = r - maxDrudeDistance;
                    mixed deltaT = stepSize;
                    mixed dotvr1 = vel1x * bx + vel1y * by + vel1z * bz;
                    const mixed vb1x = bx * dotvr1, vb1y = by * dotvr1, vb1z = bz * dotvr1;
                    const mixed vp1x = vel1x - vb1x, vp1y = vel1y - vb1y, vp1z = vel1z - vb1z;
                    if (vel2w == 0) {                                           // massless parent (K/middle.cu:151-173)
                        if (dotvr1 != 0) deltaT = deltaR / fabs((double) dotvr1);
                        if (deltaT > stepSize) deltaT = stepSize;
                        dotvr1 = -dotvr1 * hardwallscaleDrude / (fabs((double) dotvr1) * P::SQRT(mass1));
                        const mixed dr = -deltaR + deltaT * dotvr1;
                        p1x += bx * dr; p1y += by * dr; p1z += bz * dr;
                        vel1x = vp1x + bx * dotvr1; vel1y = vp1y + by * dotvr1; vel1z = vp1z + bz * dotvr1;
                        if (isd) { x = p1x; y = p1y; z = p1z; v.x = vel1x; v.y = vel1y; v.z = vel1z; pos_dirty = true; vel_dirty = true; }
                    } else {                                                    // both move (K/middle.cu:174-218)
                        const mixed invTotalMass = P::RECIP(mass1 + mass2);
                        mixed dotvr2 = vel2x * bx + vel2y * by + vel2z * bz;
                        const mixed vb2x = bx * dotvr2, vb2y = by * dotvr2, vb2z = bz * dotvr2;
                        const mixed vp2x = vel2x - vb2x, vp2y = vel2y - vb2y, vp2z = vel2z - vb2z;
                        const mixed vbCMass = (mass1 * dotvr1 + mass2 * dotvr2) * invTotalMass;
                        dotvr1 -= vbCMass;
                        dotvr2 -= vbCMass;
                        if (dotvr1 != dotvr2) deltaT = deltaR / fabs((double) (dotvr1 - dotvr2));
                        if (deltaT > stepSize) deltaT = stepSize;
                        const mixed vBond = hardwallscaleDrude / P::SQRT(mass1);
                        dotvr1 = -dotvr1 * vBond * mass2 * invTotalMass / fabs((double) dotvr1);
                        dotvr2 = -dotvr2 * vBond * mass1 * invTotalMass / fabs((double) dotvr2);
                        const mixed dr1 = -deltaR * mass2 * invTotalMass + deltaT * dotvr1;
                        const mixed dr2 = deltaR * mass1 * invTotalMass + deltaT * dotvr2;
                        dotvr1 += vbCMass;
                        dotvr2 += vbCMass;
                        if (isd) {
                            x = p1x + bx * dr1; y = p1y + by * dr1; z = p1z + bz * dr1;
                            v.x = vp1x + bx * dotvr1; v.y = vp1y + by * dotvr1; v.z = vp1z + bz * dotvr1;
                        } else {
                            x = p2x + bx * dr2; y = p2y + by * dr2; z = p2z + bz * dr2;
                            v.x = vp2x + bx * dotvr2; v.y = vp2y + by * dotvr2; v.z = vp2z + bz * dotvr2;
                        }
                        pos_dirty = true; vel_dirty = true;
                    }
                }
            }
        }

        // ---------------- write back
        VV_STAMP(wib, 4);
        if (act && vel_dirty) store_vec(velm, atom, v);
        if (act && pos_dirty) IO::store(a.posq, a.corr, atom, x, y, z, q);
        if ((F & B_VV_KICK) && massive) {
            mixed4 d = {dx, dy, dz, 0};
            if (a.pos_delta) ((mixed4*) a.pos_delta)[atom] = d;
        }
        VV_STAMP(wib, 5);

        // ---------------- image charges (K/imageCharge.cu:10-26): x, y are bit copies of the parent's stored position, z is mirrored
        if ((F & B_IMAGE) && act && (meta & META_HAS_IMAGE)) {
            real4* posq = (real4*) a.posq;
            real4 pp, cp = {0, 0, 0, 0};
            if (pos_dirty) {                   // what IO::store has just written, conversion for conversion (no re-read needed)
                pp.x = (real) x; pp.y = (real) y; pp.z = (real) z; pp.w = (real) q;
                if (IO::kMixed) { cp.x = (real) (x - (real) x); cp.y = (real) (y - (real) y); cp.z = (real) (z - (real) z); }
            } else {                           // untouched parent (massless): its stored bits
                pp = posq[atom];
                if (IO::kMixed) cp = ((const real4*) a.corr)[atom];
            }
            real4 pi = img_p;
            pi.x = pp.x; pi.y = pp.y;
            if (IO::kMixed) {
                real4* corr = (real4*) a.corr;
                real4 ci = img_c;
                ci.x = cp.x; ci.y = cp.y;
                mixed zz = (mixed) pp.z + (mixed) cp.z;
                zz = (mixed) a.mirror * 2 - zz;
                pi.z = (real) zz;
                ci.z = (real) (zz - (real) zz);
                corr[img] = ci;
            } else {
                pi.z = 2 * (mixed) a.mirror - pp.z;
            }
            posq[img] = pi;
        }
    }   // tile loop
    VV_STAMP_DUMP(wib);
    VV_SPAN_END;
}

// ================================================================================ static mass tables
// One launch per binding: slot_m = RECIP(velm.w) and, for the two lanes of a Drude pair, slot_f = invTotalMass * own mass with
// invTotalMass = RECIP(mass1 + mass2) -- the operations of K/drudeNoseHoover.cu:173-180 / K/drudeLangevin.cu:16,36-44 in the mode's
// `mixed` type, so that the stages reading the tables get the very bits they used to recompute in every step.
template <class real, class mixed>
__global__ void __launch_bounds__(256) vv_kernel_mass_table(const void* velm_, const int2* slots, int nwaves, double* slot_m, double* slot_f) {
    using mixed4 = typename Vec<mixed>::v4;
    using P = Prec<real>;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int2 slot = slots[(size_t) wave * 64 + lane];
    const unsigned meta = (unsigned) slot.y;
    const int partner = (meta >> META_PARTNER_SHIFT) & 63;
    mixed w = 0;
    if (slot.x >= 0) w = ((const mixed4*) velm_)[slot.x].w;
    const mixed m = w != 0 ? P::RECIP(w) : (mixed) 0;
    const mixed pm = shfl(m, partner);
    mixed f = 0;
    if (slot.x >= 0 && (meta & META_PAIR)) {
        const bool isd = (meta & META_IS_DRUDE) != 0;
        const mixed mass1 = isd ? m : pm, mass2 = isd ? pm : m;        // 1 = Drude, 2 = parent: the reference's operand order
        const mixed invTotalMass = P::RECIP(mass1 + mass2);
        f = invTotalMass * m;
    }
    slot_m[(size_t) wave * 64 + lane] = (double) m;
    slot_f[(size_t) wave * 64 + lane] = (double) f;
}

// ================================================================================ stand-alone image kernel
// vvhip_update_image_positions when called on its own (ModifyImageChargeKernel::updateImagePositions).
template <class real, class mixed>
__global__ void __launch_bounds__(256) vv_kernel_images(void* posq_, void* corr_, const int2* pairs, int npairs, double mirror) {
    using real4 = typename Vec<real>::v4;
    constexpr bool kMixed = sizeof(real) != sizeof(mixed);
    real4* posq = (real4*) posq_;
    real4* corr = (real4*) corr_;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += blockDim.x * gridDim.x) {
        const int2 pr = pairs[i];
        const real4 pp = posq[pr.y];
        real4 pi = posq[pr.x];
        pi.x = pp.x; pi.y = pp.y;
        if (kMixed) {
            const real4 cp = corr[pr.y];
            real4 ci = corr[pr.x];
            ci.x = cp.x; ci.y = cp.y;
            mixed z = (mixed) pp.z + (mixed) cp.z;
            z = (mixed) mirror * 2 - z;
            pi.z = (real) z;
            ci.z = (real) (z - (real) z);
            corr[pr.x] = ci;
        } else {
            pi.z = 2 * (mixed) mirror - pp.z;
        }
        posq[pr.x] = pi;
    }
}

// ================================================================================ synthetic force provider (bench/test support)
// Mirrors oracle vvo_tether_force bit for bit: tether on massive particles, Drude-parent spring, both
// converted to fixed point by truncation and then added as integers.
template <class real, class mixed>
__global__ void __launch_bounds__(512) vv_kernel_tether(const int2* __restrict__ pre_slots, const int pre_nwaves, const int pre_wpb, const TetherArgs t) {
    using real4 = typename Vec<real>::v4;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * pre_wpb + (threadIdx.x >> 6);
    if (wave >= pre_nwaves) return;
    const int2 slot = pre_slots[(size_t) wave * 64 + lane];
    const int atom = slot.x;
    const unsigned meta = (unsigned) slot.y;
    const int partner = (meta >> META_PARTNER_SHIFT) & 63;
    const bool act = atom >= 0;
    real4 p = {0, 0, 0, 0}, s = {0, 0, 0, 0};
    const bool w = act && (meta & META_MASSIVE);       // == (velm.w != 0) without touching velm
    if (act) { p = ((const real4*) t.posq)[atom]; s = ((const real4*) t.site)[atom]; }
    const real kt = (real) t.k_tether, kd = (real) t.k_drude, scale = (real) 4294967296.0;
    real fx = 0, fy = 0, fz = 0;
    if (w) { fx = -kt * (p.x - s.x); fy = -kt * (p.y - s.y); fz = -kt * (p.z - s.z); }
    long long ix = (long long) (fx * scale), iy = (long long) (fy * scale), iz = (long long) (fz * scale);
    const real ox = shfl(p.x, partner), oy = shfl(p.y, partner), oz = shfl(p.z, partner);
    if (act && (meta & META_PAIR)) {
        const bool isd = (meta & META_IS_DRUDE) != 0;
        const real dxx = isd ? p.x - ox : ox - p.x, dyy = isd ? p.y - oy : oy - p.y, dzz = isd ? p.z - oz : oz - p.z;
        const long long sx = (long long) (-kd * dxx * scale), sy = (long long) (-kd * dyy * scale), sz = (long long) (-kd * dzz * scale);
        if (isd) { ix += sx; iy += sy; iz += sz; } else { ix -= sx; iy -= sy; iz -= sz; }
    }
    if (act) { t.force[atom] = ix; t.force[atom + t.padded] = iy; t.force[atom + 2 * t.padded] = iz; }
}

// ================================================================================ device Gaussian random numbers
// Philox4x32-10 (Salmon et al., SC'11) counter-based generator + Box-Muller; 4 normals per call = one float4 of the
// buffer the Langevin stage reads.  Not part of the reference (there the buffer is OpenMM's, CudaVVKernels.cpp:63,863).
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], const uint32_t (&k)[2]) {
    const uint64_t p0 = (uint64_t) 0xD2511F53u * c[0], p1 = (uint64_t) 0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t) (p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t) p1, n2 = (uint32_t) (p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t) p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ void __launch_bounds__(256) vv_kernel_fill_normals(float4* out, uint32_t count, uint64_t seed, const unsigned long long* epoch) {
    const unsigned long long ep = *epoch;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += blockDim.x * gridDim.x) {
        uint32_t c[4] = {i, (uint32_t) ep, (uint32_t) (ep >> 32), 0x5656u};
        uint32_t key[2] = {(uint32_t) seed, (uint32_t) (seed >> 32)};
#pragma unroll
        for (int r = 0; r < 10; r++) {
            philox_round(c, key);
            key[0] += 0x9E3779B9u; key[1] += 0xBB67AE85u;
        }
        // (0,1] uniforms from the 32-bit words, then two Box-Muller pairs
        const float u0 = ((float) c[0] + 1.0f) * 2.3283064365386963e-10f, u1 = (float) c[1] * 2.3283064365386963e-10f;
        const float u2 = ((float) c[2] + 1.0f) * 2.3283064365386963e-10f, u3 = (float) c[3] * 2.3283064365386963e-10f;
        const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
        float s0, c0, s1, c1;
        sincosf(6.283185307179586f * u1, &s0, &c0);
        sincosf(6.283185307179586f * u3, &s1, &c1);
        out[i] = make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
    }
}
__global__ void vv_kernel_bump_epoch(unsigned long long* epoch) { *epoch += 1; }

// ================================================================================ launchers
extern unsigned vv_last_grid_value;      // grid of the most recent A / B launch (instrumented builds read it back)
static inline dim3 grid_for(int nwaves, int block_threads) {
    const int wpb = block_threads / 64;
    return dim3((unsigned) ((nwaves + wpb - 1) / wpb));
}
// Launch with the dispatch's own begin / end timestamps delivered into a pair of events (hipExtLaunchKernelGGL: no barrier packets
// around the kernel, the times are those of the dispatch packet's completion signal -- what rocprofv3's kernel trace reports);
// without events, the plain launch (also the only form used inside a graph capture).
template <typename F, typename... Args>
static inline void vv_launch(F kernel, dim3 g, dim3 b, unsigned lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1, Args... args) {
    if (e0 || e1) hipExtLaunchKernelGGL(kernel, g, b, lds, s, e0, e1, 0, args...);
    else hipLaunchKernelGGL(kernel, g, b, lds, s, args...);
}
#define VV_DISPATCH(KERNEL, ...)                                                                       \
    switch (precision) {                                                                               \
        case VVHIP_SINGLE: hipLaunchKernelGGL((KERNEL<float, float>), __VA_ARGS__); break;             \
        case VVHIP_MIXED: hipLaunchKernelGGL((KERNEL<float, double>), __VA_ARGS__); break;             \
        default: hipLaunchKernelGGL((KERNEL<double, double>), __VA_ARGS__); break;                     \
    }
#define VV_DISPATCH_SF(KERNEL, SFV, ...)                                                               \
    switch (precision) {                                                                               \
        case VVHIP_SINGLE: vv_launch((KERNEL<float, float, SFV>), __VA_ARGS__); break;                 \
        case VVHIP_MIXED: vv_launch((KERNEL<float, double, SFV>), __VA_ARGS__); break;                 \
        default: vv_launch((KERNEL<double, double, SFV>), __VA_ARGS__); break;                         \
    }

// Stage-bit sets with their own compiled kernel: the fused middle step of a Drude system with / without hard wall
// (BASELINE configs C3 / C2).  Everything else runs the generic kernel with run-time bits.
constexpr uint32_t SF_A_MIDDLE = A_KICK_FULL | A_KE;
constexpr uint32_t SF_A_COS1 = A_KICK_FULL | A_COS | A_BIAS | A_CZ_STORE;          // cos acceleration (BASELINE C4): kick + bias moment
constexpr uint32_t SF_A_COS2 = A_KE | A_UNBIAS_ACC | A_CZ_LOAD;                     // ... kinetic energies of the bias-free velocities
constexpr uint32_t SF_B_COS_HW = B_CHAIN | B_SCALE | B_UNBIAS | B_CZ_LOAD | B_DRIFT_MIDDLE | B_HARDWALL;
constexpr uint32_t SF_B_MIDDLE_HW = B_CHAIN | B_SCALE | B_DRIFT_MIDDLE | B_HARDWALL;
constexpr uint32_t SF_B_MIDDLE = B_CHAIN | B_SCALE | B_DRIFT_MIDDLE;
constexpr uint32_t SF_A_EDL = A_KICK_FULL | A_LD | A_EF | A_KE;                                // electrode slab (BASELINE C5): Langevin subset + field
constexpr uint32_t SF_B_EDL = SF_B_MIDDLE_HW | B_IMAGE;                                         // ... + image mirror
constexpr uint32_t SF_A_COS_MOM = A_KICK_FULL | A_COS | A_BIAS | A_CZ_STORE | A_KE | A_KE_MOM;   // cos acceleration in one launch (moments)
constexpr uint32_t SF_B_COS_HW_MOM = SF_B_COS_HW | B_KE_MOM;
constexpr uint32_t SF_B_MIDDLE_HW_NC = B_SCALE | B_DRIFT_MIDDLE | B_HARDWALL;                // large systems: the chain runs as its own 1-wave launch in front
constexpr uint32_t SF_B_MIDDLE_HW_MB = SF_B_MIDDLE_HW | B_MAILBOX;
constexpr uint32_t SF_A_MIDDLE_SHAKE = SF_A_MIDDLE | A_SHAKE_V;                         // HBonds constraints solved in-kernel
constexpr uint32_t SF_B_MIDDLE_HW_SHAKE = SF_B_MIDDLE_HW | B_SHAKE;
constexpr uint32_t SF_B_MIDDLE_SHAKE = SF_B_MIDDLE | B_SHAKE;
constexpr uint32_t SF_A_MIDDLE_SETTLE = SF_A_MIDDLE | A_SETTLE;                       // rigid water (BASELINE C2 made physical): SETTLE only
constexpr uint32_t SF_B_MIDDLE_SETTLE = SF_B_MIDDLE | B_SETTLE;

// Further stage sets with their own compiled kernel (the generic kernel with run-time stage bits is 15-20 % slower: C5 went from
// 74 k to 88 k steps/s when it got its own pair): the classic scheme's two halves and the stages of the un-fused entry points.
constexpr uint32_t SF_A_KE = A_KE;                                              // vvhip_scale_velocity / classic first half: sums only
constexpr uint32_t SF_A_VV2 = A_KICK_HALF | A_KE;                               // classic second half
constexpr uint32_t SF_A_KICK = A_KICK_FULL;                                     // vvhip_middle_kick (forceExtra known to be zero)
constexpr uint32_t SF_A_KICK_FE = A_FE_LOAD | A_KICK_FULL;                      // ... with extra forces
constexpr uint32_t SF_A_POS1 = A_POS1;                                          // vvhip_middle_half_drift1
constexpr uint32_t SF_B_SCALE = B_CHAIN | B_SCALE;                              // vvhip_scale_velocity / classic second half
constexpr uint32_t SF_B_VV1_HW = B_CHAIN | B_SCALE | B_VV_KICK | B_HARDWALL;    // classic first half
constexpr uint32_t SF_B_VV1 = B_CHAIN | B_SCALE | B_VV_KICK;
constexpr uint32_t SF_B_POS2 = B_POS2;                                          // vvhip_middle_half_drift2
constexpr uint32_t SF_B_POS3_HW = B_POS3 | B_HARDWALL;                          // vvhip_middle_finish
constexpr uint32_t SF_B_POS3 = B_POS3;
constexpr uint32_t SF_B_MIDDLE_MB = SF_B_MIDDLE | B_MAILBOX;                    // sharded runs of systems without Drude pairs

// ... and the combinations the reference's example scripts actually run: HBonds constraints next to the cos perturbation
// (run-bulk.py) and next to the electrode machinery (run-edl.py), and the sharded variants of the constrained / perturbed box
constexpr uint32_t SF_A_COS_MOM_SHAKE = SF_A_COS_MOM | A_SHAKE_V;
constexpr uint32_t SF_B_COS_HW_MOM_SHAKE = SF_B_COS_HW_MOM | B_SHAKE;
constexpr uint32_t SF_A_EDL_SHAKE = SF_A_EDL | A_SHAKE_V;
constexpr uint32_t SF_B_EDL_SHAKE = SF_B_EDL | B_SHAKE;
constexpr uint32_t SF_A_LD = A_KICK_FULL | A_LD | A_KE;                     // a Langevin subset without the field (thermostatted wall)
constexpr uint32_t SF_A_EF = A_KICK_FULL | A_EF | A_KE;                     // a field on the bulk without Langevin particles
constexpr uint32_t SF_A_LD_SHAKE = SF_A_LD | A_SHAKE_V;
constexpr uint32_t SF_A_EF_SHAKE = SF_A_EF | A_SHAKE_V;
constexpr uint32_t SF_B_COS_HW_MOM_MB = SF_B_COS_HW_MOM | B_MAILBOX;
constexpr uint32_t SF_B_MIDDLE_HW_SHAKE_MB = SF_B_MIDDLE_HW_SHAKE | B_MAILBOX;
// the headline path without a velm round trip between the kernels (A_NOSTORE / B_KICK): plain, sharded, very large, without hard wall
constexpr uint32_t SF_A_MIDDLE_NS = SF_A_MIDDLE | A_NOSTORE;
constexpr uint32_t SF_B_MIDDLE_HW_K = SF_B_MIDDLE_HW | B_KICK;
constexpr uint32_t SF_B_MIDDLE_K = SF_B_MIDDLE | B_KICK;
constexpr uint32_t SF_B_MIDDLE_HW_NC_K = SF_B_MIDDLE_HW_NC | B_KICK;
constexpr uint32_t SF_B_MIDDLE_HW_MB_K = SF_B_MIDDLE_HW_MB | B_KICK;
constexpr uint32_t SF_B_MIDDLE_MB_K = SF_B_MIDDLE_MB | B_KICK;
constexpr uint32_t SF_A_COS_MOM_NS = SF_A_COS_MOM | A_NOSTORE;                         // ... and with the cos perturbation (BASELINE C4)
constexpr uint32_t SF_B_COS_HW_MOM_K = SF_B_COS_HW_MOM | B_KICK;
constexpr uint32_t SF_B_COS_HW_MOM_MB_K = SF_B_COS_HW_MOM_MB | B_KICK;
// ... with the arithmetic work-item layout (runs of identical molecules: every BASELINE bulk configuration)
constexpr uint32_t SF_A_MIDDLE_NS_P = SF_A_MIDDLE_NS | A_PERIODIC;
constexpr uint32_t SF_A_COS_MOM_NS_P = SF_A_COS_MOM_NS | A_PERIODIC;
constexpr uint32_t SF_B_MIDDLE_HW_K_P = SF_B_MIDDLE_HW_K | B_PERIODIC;
constexpr uint32_t SF_B_MIDDLE_K_P = SF_B_MIDDLE_K | B_PERIODIC;
constexpr uint32_t SF_B_MIDDLE_HW_NC_K_P = SF_B_MIDDLE_HW_NC_K | B_PERIODIC;
constexpr uint32_t SF_B_COS_HW_MOM_K_P = SF_B_COS_HW_MOM_K | B_PERIODIC;
// large constrained boxes (the chain as its own launch): HBonds clusters solved in kernel B without the thermostat wave
constexpr uint32_t SF_B_MIDDLE_HW_NC_SHAKE = SF_B_MIDDLE_HW_NC | B_SHAKE;
constexpr uint32_t SF_B_MIDDLE_HW_NC_SHAKE_P = SF_B_MIDDLE_HW_NC_SHAKE | B_PERIODIC;
constexpr uint32_t SF_B_MIDDLE_HW_SHAKE_P = SF_B_MIDDLE_HW_SHAKE | B_PERIODIC;
constexpr uint32_t SF_A_MIDDLE_SHAKE_P = SF_A_MIDDLE_SHAKE | A_PERIODIC;
// ... and the other stage sets large boxes produce (no thermostat wave in kernel B): without Drude pairs (water, plain ionic liquids), rigid
// water, the cos perturbation in its three-launch form -- each with loaded and with computed slot words
constexpr uint32_t SF_B_MIDDLE_NC_K = B_SCALE | B_DRIFT_MIDDLE | B_KICK;
constexpr uint32_t SF_B_MIDDLE_NC_K_P = SF_B_MIDDLE_NC_K | B_PERIODIC;
constexpr uint32_t SF_B_MIDDLE_NC_SHAKE = B_SCALE | B_DRIFT_MIDDLE | B_SHAKE;
constexpr uint32_t SF_B_MIDDLE_NC_SHAKE_P = SF_B_MIDDLE_NC_SHAKE | B_PERIODIC;
constexpr uint32_t SF_B_MIDDLE_NC_SETTLE = B_SCALE | B_DRIFT_MIDDLE | B_SETTLE;
constexpr uint32_t SF_B_MIDDLE_NC_SETTLE_P = SF_B_MIDDLE_NC_SETTLE | B_PERIODIC;
constexpr uint32_t SF_A_MIDDLE_SETTLE_P = SF_A_MIDDLE_SETTLE | A_PERIODIC;
constexpr uint32_t SF_B_COS_HW_NC = B_SCALE | B_UNBIAS | B_CZ_LOAD | B_DRIFT_MIDDLE | B_HARDWALL;
constexpr uint32_t SF_B_COS_HW_NC_P = SF_B_COS_HW_NC | B_PERIODIC;
// the classic scheme's two thermostat applications in large boxes: scale + half kick + positions (+ hard wall), and scale alone
constexpr uint32_t SF_B_VV1_HW_NC = B_SCALE | B_VV_KICK | B_HARDWALL;
constexpr uint32_t SF_B_VV1_HW_NC_P = SF_B_VV1_HW_NC | B_PERIODIC;
constexpr uint32_t SF_B_SCALE_NC = B_SCALE;
constexpr uint32_t SF_B_SCALE_NC_P = SF_B_SCALE_NC | B_PERIODIC;

// The classic scheme (two thermostat applications per step) of every BASELINE configuration with what the example scripts add to it:
// second half = half kick (+ extra forces kept for the next first half) + sums, first half = scale + half kick + positions.
constexpr uint32_t SF_A_KE_P = A_KE | A_PERIODIC;                                              // large boxes
constexpr uint32_t SF_A_VV2_P = SF_A_VV2 | A_PERIODIC;
constexpr uint32_t SF_A_VV2_SHAKE = SF_A_VV2 | A_SHAKE_V;                                      // + HBonds
constexpr uint32_t SF_A_VV2_SETTLE = SF_A_VV2 | A_SETTLE;                                      // rigid water
constexpr uint32_t SF_A_VV2_EDL = SF_A_VV2 | A_FE_STORE | A_LD | A_EF;                         // electrode slab
constexpr uint32_t SF_A_VV2_EDL_SHAKE = SF_A_VV2_EDL | A_SHAKE_V;
constexpr uint32_t SF_A_COS_MOM_VV1 = A_BIAS | A_KE | A_CZ_STORE | A_KE_MOM;                   // cos perturbation: sums of the first half
constexpr uint32_t SF_A_COS_MOM_VV2 = SF_A_COS_MOM_VV1 | A_FE_STORE | A_COS | A_KICK_HALF;     // ... half kick + sums of the second half
constexpr uint32_t SF_A_COS_MOM_VV2_SHAKE = SF_A_COS_MOM_VV2 | A_SHAKE_V;
constexpr uint32_t SF_B_VV1_HW_SHAKE = SF_B_VV1_HW | B_SHAKE;
constexpr uint32_t SF_B_VV1_SETTLE = SF_B_VV1 | B_SETTLE;
constexpr uint32_t SF_B_VV1_EDL = SF_B_VV1_HW | B_IMAGE;
constexpr uint32_t SF_B_VV1_EDL_SHAKE = SF_B_VV1_EDL | B_SHAKE;
constexpr uint32_t SF_B_COS_SCALE_MOM = B_CHAIN | B_SCALE | B_UNBIAS | B_CZ_LOAD | B_KE_MOM;   // cos perturbation: second half's scaling
constexpr uint32_t SF_B_COS_VV1_HW_MOM = SF_B_COS_SCALE_MOM | B_VV_KICK | B_HARDWALL;          // ... first half
constexpr uint32_t SF_B_COS_VV1_HW_MOM_SHAKE = SF_B_COS_VV1_HW_MOM | B_SHAKE;
constexpr uint32_t SF_B_MIDDLE_SETTLE_P = SF_B_MIDDLE_SETTLE | B_PERIODIC;                      // rigid water between 0.2 M and 0.8 M particles

// Which specialised kernels are compiled with the static mass tables; a launch whose flags disagree with the build falls through to
// the generic kernel.  Measured on MI355X (gpurun_out/r02c-e): kernel B gains at every size (two IEEE fp64 divisions per pair lane
// gone, 140 -> 101 VGPRs; C3 5.74 -> 5.29 us with the chain fix, 8.9 M particles unchanged), kernel A does not -- with the Koenig
// form of its KE stage it needs one cheap reciprocal per lane, and the 16 bytes per lane of table traffic cost it 13 % at
// 8.9 M particles (217 -> 247 us) for nothing at 111 k.  So: B with the table (VV_SF_MTAB_B=1), A without (VV_SF_MTAB_A=0).
#ifndef VV_SF_MTAB_A
#define VV_SF_MTAB_A 0
#endif
#ifndef VV_SF_MTAB_B
#define VV_SF_MTAB_B 1
#endif
constexpr uint32_t SF_AM = VV_SF_MTAB_A ? A_MTAB : 0u, SF_BM = VV_SF_MTAB_B ? B_MTAB : 0u;
bool sf_kernels_use_mass_table(int kernel) { return kernel == 0 ? VV_SF_MTAB_A != 0 : VV_SF_MTAB_B != 0; }
#define VV_TRY_SF(KERNEL, SFV) if (a.flags == ((SFV) | XM)) { VV_DISPATCH_SF(KERNEL, ((SFV) | XM), g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a); return hipGetLastError(); }

// Launches that found no compiled specialisation of their stage set and ran the generic kernel with run-time stage bits (15-20 % slower):
// counted per kernel with the last such stage set (vvhip_generic_launches reads them); VVHIP_WARN_GENERIC=1 also prints one line on
// stderr per stage set.
unsigned long long vv_generic_count[2] = {0, 0};
uint32_t vv_generic_flags[2] = {0, 0};
static void note_generic(const char* kernel, uint32_t flags) {
    vv_generic_count[kernel[0] == 'A' ? 0 : 1]++;
    vv_generic_flags[kernel[0] == 'A' ? 0 : 1] = flags;
    static const bool on = std::getenv("VVHIP_WARN_GENERIC") != nullptr;
    if (!on) return;
    static uint32_t seen[2][16];
    static int nseen[2] = {0, 0};
    const int k = kernel[0] == 'A' ? 0 : 1;
    for (int i = 0; i < nseen[k]; i++) if (seen[k][i] == flags) return;
    if (nseen[k] < 16) seen[k][nseen[k]++] = flags;
    std::fprintf(stderr, "vvhip: kernel %s runs stage set 0x%x on the generic kernel (no compiled specialisation)\n", kernel, flags);
}

hipError_t launch_a(int precision, const KArgs& a_in, int block_threads, int grid_cap, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    KArgs a = a_in;
    dim3 g = grid_for(a.nwaves, block_threads);
    if ((int) g.x > grid_cap) g.x = (unsigned) grid_cap;          // beyond that the kernel strides over tiles
    vv_last_grid_value = g.x;
    if (g.x > (unsigned) ACC_SLOTS) a.acc_exclusive = 0;          // several blocks per accumulator slot: atomics (block_accumulate)
    const dim3 b(block_threads);
    // per-wave LDS page of the in-kernel constraint solver (7 values of the mode's `mixed` type per lane)
    const unsigned lds = (a.flags & A_CONS) ? (unsigned) (block_threads / 64) * 64u * 7u * (precision == VVHIP_SINGLE ? 4u : 8u) : 0u;
    constexpr uint32_t XM = SF_AM;
#define VV_PRE_ARGS a.slots, a.nwaves, (int) (b.x >> 6), a.velm, a.force, a.padded
    VV_TRY_SF(vv_kernel_a, SF_A_MIDDLE_NS_P)
    VV_TRY_SF(vv_kernel_a, SF_A_COS_MOM_NS_P)
    VV_TRY_SF(vv_kernel_a, SF_A_MIDDLE_NS)
    VV_TRY_SF(vv_kernel_a, SF_A_COS_MOM_NS)
    VV_TRY_SF(vv_kernel_a, SF_A_MIDDLE)
    VV_TRY_SF(vv_kernel_a, SF_A_COS1)
    VV_TRY_SF(vv_kernel_a, SF_A_COS2)
    VV_TRY_SF(vv_kernel_a, SF_A_COS_MOM)
    VV_TRY_SF(vv_kernel_a, SF_A_EDL)
    VV_TRY_SF(vv_kernel_a, SF_A_MIDDLE_SHAKE)
    VV_TRY_SF(vv_kernel_a, SF_A_MIDDLE_SHAKE_P)
    VV_TRY_SF(vv_kernel_a, SF_A_MIDDLE_SETTLE)
    VV_TRY_SF(vv_kernel_a, SF_A_MIDDLE_SETTLE_P)
    VV_TRY_SF(vv_kernel_a, SF_A_COS_MOM_SHAKE)
    VV_TRY_SF(vv_kernel_a, SF_A_EDL_SHAKE)
    VV_TRY_SF(vv_kernel_a, SF_A_LD)
    VV_TRY_SF(vv_kernel_a, SF_A_EF)
    VV_TRY_SF(vv_kernel_a, SF_A_LD_SHAKE)
    VV_TRY_SF(vv_kernel_a, SF_A_EF_SHAKE)
    VV_TRY_SF(vv_kernel_a, SF_A_KE)
    VV_TRY_SF(vv_kernel_a, SF_A_VV2)
    VV_TRY_SF(vv_kernel_a, SF_A_KICK)
    VV_TRY_SF(vv_kernel_a, SF_A_KICK_FE)
    VV_TRY_SF(vv_kernel_a, SF_A_POS1)
    VV_TRY_SF(vv_kernel_a, SF_A_KE_P)
    VV_TRY_SF(vv_kernel_a, SF_A_VV2_P)
    VV_TRY_SF(vv_kernel_a, SF_A_VV2_SHAKE)
    VV_TRY_SF(vv_kernel_a, SF_A_VV2_SETTLE)
    VV_TRY_SF(vv_kernel_a, SF_A_VV2_EDL)
    VV_TRY_SF(vv_kernel_a, SF_A_VV2_EDL_SHAKE)
    VV_TRY_SF(vv_kernel_a, SF_A_COS_MOM_VV1)
    VV_TRY_SF(vv_kernel_a, SF_A_COS_MOM_VV2)
    VV_TRY_SF(vv_kernel_a, SF_A_COS_MOM_VV2_SHAKE)
    note_generic("A", a.flags);
    VV_DISPATCH_SF(vv_kernel_a, 0u, g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a);
    return hipGetLastError();
#undef VV_PRE_ARGS
}
hipError_t launch_b(int precision, const KArgs& a, int block_threads, int grid_cap, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    // block_threads counts the tile waves; B_CHAIN adds the block's thermostat wave.  Beyond grid_cap blocks the kernel strides
    // over tiles and the per-block thermostat work is amortised.
    dim3 g = grid_for(a.nwaves, block_threads);
    if ((int) g.x > grid_cap) g.x = (unsigned) grid_cap;
    vv_last_grid_value = g.x;
    const dim3 b(block_threads + ((a.flags & B_CHAIN) ? 64 : 0));
    const unsigned lds = (a.flags & B_CONS) ? (unsigned) (block_threads / 64) * 64u * 7u * (precision == VVHIP_SINGLE ? 4u : 8u) : 0u;      // one page per tile wave
    constexpr uint32_t XM = SF_BM;
#define VV_PRE_ARGS a.slots, a.nwaves, (int) (b.x >> 6), (const unsigned long long*) a.acc, a.nh, a.lane_const, a.seg_base
    if ((a.flags & B_CHAIN) && a.chain.num_chains != 3) {       // the specialised kernels carry the three-link chain only
        note_generic("B", a.flags);
        VV_DISPATCH_SF(vv_kernel_b, 0u, g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a);
        return hipGetLastError();
    }
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_NC_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_MOM_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_K)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_K)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_NC_K)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_MB_K)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_MB_K)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_MOM_K)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_MOM_MB_K)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_MOM)
    VV_TRY_SF(vv_kernel_b, SF_B_EDL)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_NC)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_MB)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_MB)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_NC_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_NC_K)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_NC_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_NC_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_NC_SHAKE_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_SETTLE)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_NC_SETTLE)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_NC_SETTLE_P)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_NC)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_NC_P)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1_HW_NC)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1_HW_NC_P)
    VV_TRY_SF(vv_kernel_b, SF_B_SCALE_NC)
    VV_TRY_SF(vv_kernel_b, SF_B_SCALE_NC_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_NC_SHAKE_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_SHAKE_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_MOM_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_EDL_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_MOM_MB)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_SHAKE_MB)
    VV_TRY_SF(vv_kernel_b, SF_B_SCALE)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1_HW)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1)
    VV_TRY_SF(vv_kernel_b, SF_B_POS2)
    VV_TRY_SF(vv_kernel_b, SF_B_POS3_HW)
    VV_TRY_SF(vv_kernel_b, SF_B_POS3)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1_HW_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1_SETTLE)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1_EDL)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1_EDL_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_SCALE_MOM)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_VV1_HW_MOM)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_VV1_HW_MOM_SHAKE)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_SETTLE_P)
    note_generic("B", a.flags);
    VV_DISPATCH_SF(vv_kernel_b, 0u, g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a);
    return hipGetLastError();
#undef VV_PRE_ARGS
}
hipError_t launch_chain(const NHConst& c, NHDevState* st, unsigned long long* acc, hipStream_t s) {
    hipLaunchKernelGGL(vv_kernel_chain, dim3(1), dim3(64), 0, s, c, st, acc);
    return hipGetLastError();
}
hipError_t launch_tether(int precision, const TetherArgs& t, int block_threads, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    switch (precision) {
        case VVHIP_SINGLE: vv_launch((vv_kernel_tether<float, float>), grid_for(t.nwaves, block_threads), dim3(block_threads), 0, s, ev0, ev1, t.slots, t.nwaves, block_threads / 64, t); break;
        case VVHIP_MIXED: vv_launch((vv_kernel_tether<float, double>), grid_for(t.nwaves, block_threads), dim3(block_threads), 0, s, ev0, ev1, t.slots, t.nwaves, block_threads / 64, t); break;
        default: vv_launch((vv_kernel_tether<double, double>), grid_for(t.nwaves, block_threads), dim3(block_threads), 0, s, ev0, ev1, t.slots, t.nwaves, block_threads / 64, t); break;
    }
    return hipGetLastError();
}
hipError_t launch_mass_table(int precision, const void* velm, const int2* slots, int nwaves, double* slot_m, double* slot_f, hipStream_t s) {
    if (nwaves <= 0) return hipSuccess;
    VV_DISPATCH(vv_kernel_mass_table, dim3((unsigned) ((nwaves + 3) / 4)), dim3(256), 0, s, velm, slots, nwaves, slot_m, slot_f);
    return hipGetLastError();
}
hipError_t launch_fill_normals(float4* out, uint32_t count, uint64_t seed, unsigned long long* epoch, hipStream_t s) {
    if (count == 0) return hipSuccess;
    const unsigned blocks = (count + 255) / 256 > 1024 ? 1024 : (count + 255) / 256;
    hipLaunchKernelGGL(vv_kernel_fill_normals, dim3(blocks), dim3(256), 0, s, out, count, seed, (const unsigned long long*) epoch);
    hipLaunchKernelGGL(vv_kernel_bump_epoch, dim3(1), dim3(1), 0, s, epoch);
    return hipGetLastError();
}
hipError_t launch_image_pairs(int precision, void* posq, void* corr, const int2* pairs, int npairs, double mirror, hipStream_t s) {
    if (npairs <= 0) return hipSuccess;
    const int blocks = (npairs + 255) / 256;
    VV_DISPATCH(vv_kernel_images, dim3(blocks), dim3(256), 0, s, posq, corr, pairs, npairs, mirror);
    return hipGetLastError();
}

}  // namespace vv
