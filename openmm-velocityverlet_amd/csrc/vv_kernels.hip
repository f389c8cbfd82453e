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
// Build: the Makefile compiles this file twice, in parallel -- VV_KERNELS_PART=1: launch_a / launch_b and the small launchers, =2: launch_fused
// (the one-launch step's instances of vv_kernel_b are half of the device code) -- and links both objects; without the macro one translation
// unit holds everything (make asm, instrumented probe builds).
#ifndef VV_KERNELS_PART
#define VV_KERNELS_PART 0
#endif
#if VV_KERNELS_PART == 2
#define VV_DEVICE_NO_PLAIN_KERNELS      // (vv_kernel_chain / vv_kernel_bump_epoch are not templates: one definition, in part 1)
#endif
#include "vv_kernels.hpp"

#include <hip/hip_ext.h>

#include "vv_host.hpp"
#include "vv_rtc.hpp"

namespace vv {

#include "vv_device.inc"

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
// The same for a kernel compiled at run time (vv_rtc.cpp): module launch with the parameter list of the kernel's signature.
static inline hipError_t vv_launch_module(hipFunction_t f, dim3 g, dim3 b, unsigned lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1, void** params) {
    if (e0 || e1) return hipExtModuleLaunchKernel(f, g.x * b.x, 1, 1, b.x, 1, 1, lds, s, params, nullptr, e0, e1, 0);
    return hipModuleLaunchKernel(f, g.x, 1, 1, b.x, 1, 1, lds, s, params, nullptr);
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
// ... next to the mailbox exchange of sharded runs (the chain stays in kernel B's head there, whatever the size)
constexpr uint32_t SF_B_MIDDLE_HW_MB_K_P = SF_B_MIDDLE_HW_MB_K | B_PERIODIC;
constexpr uint32_t SF_B_MIDDLE_MB_K_P = SF_B_MIDDLE_MB_K | B_PERIODIC;
constexpr uint32_t SF_B_COS_HW_MOM_MB_K_P = SF_B_COS_HW_MOM_MB_K | B_PERIODIC;
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
constexpr uint32_t SF_B_VV1_HW_P = SF_B_VV1_HW | B_PERIODIC;                   // classic scheme, arithmetic layout, chain in the kernel (1.1 - 2.6 M particles)
constexpr uint32_t SF_B_SCALE_P = SF_B_SCALE | B_PERIODIC;
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
constexpr uint32_t SF_B_MIDDLE_SETTLE_P = SF_B_MIDDLE_SETTLE | B_PERIODIC;                      // rigid water with the arithmetic layout and the chain in the kernel (forced layouts: automatic ones start where the chain is its own launch)

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
#if VV_KERNELS_PART != 2
bool sf_kernels_use_mass_table(int kernel) { return kernel == 0 ? VV_SF_MTAB_A != 0 : VV_SF_MTAB_B != 0; }
#endif
#define VV_TRY_SF(KERNEL, SFV) if (a.flags == ((SFV) | XM)) { VV_DISPATCH_SF(KERNEL, ((SFV) | XM), g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a); return hipGetLastError(); }

#if VV_KERNELS_PART != 2
// Launches that found no compiled specialisation of their stage set and ran the generic kernel with run-time stage bits (15-20 % slower):
// counted per kernel with the last such stage set (vvhip_generic_launches reads them); VVHIP_WARN_GENERIC=1 also prints one line on
// stderr per stage set.
std::atomic<unsigned long long> vv_generic_count[2];
static void note_generic(const char* kernel, int* route) {
    vv_generic_count[kernel[0] == 'A' ? 0 : 1]++;
    if (route) *route = ROUTE_GENERIC;
}

hipError_t launch_a(int precision, const KArgs& a_in, int block_threads, int grid_cap, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int* route) {
    KArgs a = a_in;
    if (route) *route = ROUTE_COMPILED;
    dim3 g = grid_for(a.nwaves, block_threads);
    if ((int) g.x > grid_cap) g.x = (unsigned) grid_cap;          // beyond that the kernel strides over tiles
    vv_last_grid_value = g.x;
    if (g.x > (unsigned) ACC_SLOTS) a.acc_exclusive = 0;          // several blocks per accumulator slot: atomics (block_accumulate)
    const dim3 b(block_threads);
    // per-wave LDS page of the in-kernel constraint solver (7 values of the mode's `mixed` type per lane)
    const unsigned lds = (a.flags & A_CONS) ? (unsigned) (block_threads / 64) * 64u * 7u * (precision == VVHIP_SINGLE ? 4u : 8u) : 0u;
    constexpr uint32_t XM = SF_AM;
#define VV_PRE_ARGS a.slots, a.nwaves, (int) (b.x >> 6), a.velm, a.force, a.padded
    // a kernel compiled at run time for exactly this stage set: where the list below has none (VVHIP_RTC=1, default), or always (=2)
    hipError_t rtc_error = hipSuccess;
    auto run_time_kernel = [&]() -> bool {
        hipFunction_t f = rtc_kernel('A', precision, a.flags, 3);
        if (!f) return false;
        const int2* p_slots = a.slots; int p_nwaves = a.nwaves, p_wpb = (int) (b.x >> 6), p_padded = a.padded;
        void* p_velm = a.velm; const long long* p_force = a.force;
        void* params[] = {&p_slots, &p_nwaves, &p_wpb, &p_velm, &p_force, &p_padded, &a};
        rtc_error = vv_launch_module(f, g, b, lds, s, ev0, ev1, params);
        vv_rtc_launches[0]++;
        if (route) *route = ROUTE_RUNTIME;
        return true;
    };
    if (rtc_mode() >= 2 && run_time_kernel()) return rtc_error;
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
    if (rtc_mode() == 1 && run_time_kernel()) return rtc_error;
    note_generic("A", route);
    VV_DISPATCH_SF(vv_kernel_a, 0u, g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a);
    return hipGetLastError();
#undef VV_PRE_ARGS
}
hipError_t launch_b(int precision, const KArgs& a, int block_threads, int grid_cap, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int* route) {
    if (route) *route = ROUTE_COMPILED;
    // block_threads counts the tile waves; B_CHAIN adds the block's thermostat wave.  Beyond grid_cap blocks the kernel strides
    // over tiles and the per-block thermostat work is amortised.
    dim3 g = grid_for(a.nwaves, block_threads);
    if ((int) g.x > grid_cap) g.x = (unsigned) grid_cap;
    vv_last_grid_value = g.x;
    const dim3 b(block_threads + ((a.flags & B_CHAIN) ? 64 : 0));
    const unsigned lds = (a.flags & (B_CONS | B_VSITE)) ? (unsigned) (block_threads / 64) * 64u * 7u * (precision == VVHIP_SINGLE ? 4u : 8u) : 0u;      // one page per tile wave
    constexpr uint32_t XM = SF_BM;
#define VV_PRE_ARGS a.slots, a.nwaves, (int) (b.x >> 6), (const unsigned long long*) a.acc, a.nh, a.lane_const, a.seg_base
    hipError_t rtc_error = hipSuccess;
    auto run_time_kernel = [&]() -> bool {          // as in launch_a; the thermostat wave is built for the plan's chain length
        hipFunction_t f = rtc_kernel('B', precision, a.flags, a.chain.num_chains);
        if (!f) return false;
        const int2* p_slots = a.slots; int p_nwaves = a.nwaves, p_wpb = (int) (b.x >> 6);
        const unsigned long long* p_acc = a.acc; const NHDevState* p_nh = a.nh; const ChainLaneBlock* p_lc = a.lane_const; const int* p_sb = a.seg_base;
        KArgs copy = a;
        void* params[] = {&p_slots, &p_nwaves, &p_wpb, &p_acc, &p_nh, &p_lc, &p_sb, &copy};
        rtc_error = vv_launch_module(f, g, b, lds, s, ev0, ev1, params);
        vv_rtc_launches[1]++;
        if (route) *route = ROUTE_RUNTIME;
        return true;
    };
    if (rtc_mode() >= 2 && run_time_kernel()) return rtc_error;
    if ((a.flags & B_CHAIN) && a.chain.num_chains != 3) {       // the compiled kernels carry the three-link chain only
        if (rtc_mode() == 1 && run_time_kernel()) return rtc_error;
        note_generic("B", route);
        VV_DISPATCH_SF(vv_kernel_b, 0u, g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a);
        return hipGetLastError();
    }
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_NC_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_MOM_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_HW_MB_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_MIDDLE_MB_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_COS_HW_MOM_MB_K_P)
    VV_TRY_SF(vv_kernel_b, SF_B_VV1_HW_P)
    VV_TRY_SF(vv_kernel_b, SF_B_SCALE_P)
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
    if (rtc_mode() == 1 && run_time_kernel()) return rtc_error;
    note_generic("B", route);
    VV_DISPATCH_SF(vv_kernel_b, 0u, g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a);
    return hipGetLastError();
#undef VV_PRE_ARGS
}
#endif      // VV_KERNELS_PART != 2
#if VV_KERNELS_PART != 1
// ---- fused step: kernel A's stage set `a.flags_a` and kernel B's `a.flags` in ONE launch (vv_device.inc: "fused step").  Every tile needs a
// wave of its own (it stays in registers across the rendezvous) and the blocks must be resident together: the caller (vv_api.cpp:
// use_fused) has checked the launch shape; `blocks_per_cu` != nullptr only ASKS how many blocks of this kernel a CU holds and launches
// nothing.  hipErrorNotSupported: no kernel for this pair of stage sets (the caller then takes the two-launch path).
constexpr uint32_t SF_B_COS_HW_MOM_F = B_CHAIN | B_SCALE | B_UNBIAS | B_DRIFT_MIDDLE | B_HARDWALL | B_KE_MOM;      // (no cos(kz) cache between the halves)
constexpr uint32_t SF_B_COS_SCALE_MOM_F = B_CHAIN | B_SCALE | B_UNBIAS | B_KE_MOM;                                  // classic scheme: second half / first half
constexpr uint32_t SF_B_COS_VV1_HW_MOM_F = SF_B_COS_SCALE_MOM_F | B_VV_KICK | B_HARDWALL;
hipError_t launch_fused(int precision, const KArgs& a, int block_threads, const unsigned long long* rendezvous, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int* route, int* blocks_per_cu) {
    if (route) *route = ROUTE_COMPILED;
    const dim3 g = grid_for(a.nwaves, block_threads);
    const dim3 b(block_threads + 64);                      // the tile waves + the block's thermostat wave
    if (!(a.flags & B_CHAIN) || a.flags_a == 0 || (int) g.x > ACC_SLOTS || b.x > 512) return hipErrorNotSupported;
    const unsigned lds = ((a.flags & (B_CONS | B_VSITE)) || (a.flags_a & A_CONS)) ? (unsigned) (block_threads / 64) * 64u * 7u * (precision == VVHIP_SINGLE ? 4u : 8u) : 0u;      // one page per tile wave
    vv_last_grid_value = g.x;
    constexpr uint32_t XM = SF_BM;
#define VV_PRE_ARGS a.slots, a.nwaves, (int) (b.x >> 6), rendezvous, a.nh, a.lane_const, a.seg_base
#define VV_FUSED_ONE(REAL, MIXED, SFB, SFA) { \
        if (blocks_per_cu) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, vv_kernel_b<REAL, MIXED, (SFB) | XM, SFA>, (int) b.x, lds); \
        vv_launch((vv_kernel_b<REAL, MIXED, (SFB) | XM, SFA>), g, b, lds, s, ev0, ev1, VV_PRE_ARGS, a); return hipGetLastError(); }
#define VV_TRY_FUSED(SFB, SFA) if (a.flags == ((SFB) | XM) && a.flags_a == (SFA) && a.chain.num_chains == 3) { \
        switch (precision) { case VVHIP_SINGLE: VV_FUSED_ONE(float, float, SFB, SFA) case VVHIP_MIXED: VV_FUSED_ONE(float, double, SFB, SFA) default: VV_FUSED_ONE(double, double, SFB, SFA) } }
    if (rtc_mode() < 2) {
        VV_TRY_FUSED(SF_B_MIDDLE_HW, SF_A_MIDDLE)                    // BASELINE C3
        VV_TRY_FUSED(SF_B_MIDDLE, SF_A_MIDDLE)                       // C1, C2
        VV_TRY_FUSED(SF_B_EDL, SF_A_EDL)                             // C5
        VV_TRY_FUSED(SF_B_COS_HW_MOM_F, SF_A_COS_MOM)                // C4
        VV_TRY_FUSED(SF_B_MIDDLE_HW_SHAKE, SF_A_MIDDLE_SHAKE)        // ... with their HBonds / rigid water
        VV_TRY_FUSED(SF_B_MIDDLE_SETTLE, SF_A_MIDDLE_SETTLE)
        VV_TRY_FUSED(SF_B_EDL_SHAKE, SF_A_EDL_SHAKE)
        VV_TRY_FUSED(SF_B_COS_HW_MOM_F | B_SHAKE, SF_A_COS_MOM_SHAKE)
        VV_TRY_FUSED(SF_B_MIDDLE_HW_MB, SF_A_MIDDLE)                 // sharded runs (xGMI mailbox behind the local rendezvous): C3, C4, water
        VV_TRY_FUSED(SF_B_COS_HW_MOM_F | B_MAILBOX, SF_A_COS_MOM)
        VV_TRY_FUSED(SF_B_MIDDLE_MB, SF_A_MIDDLE)
        VV_TRY_FUSED(SF_B_MIDDLE_HW_SHAKE_MB, SF_A_MIDDLE_SHAKE)
        VV_TRY_FUSED(SF_B_VV1_HW, SF_A_KE)                           // classic scheme, first half: sums + scaling + half kick + drift (C3)
        VV_TRY_FUSED(SF_B_VV1, SF_A_KE)                              // ... C1, C2
        VV_TRY_FUSED(SF_B_SCALE, SF_A_VV2)                           // classic scheme, second half: half kick + sums + scaling
        VV_TRY_FUSED(SF_B_SCALE, SF_A_KE)                            // a thermostat application on its own (vvhip_scale_velocity)
        VV_TRY_FUSED(SF_B_VV1_HW_SHAKE, SF_A_KE)                     // ... the classic scheme of the other BASELINE configurations and their constraints
        VV_TRY_FUSED(SF_B_SCALE, SF_A_VV2_SHAKE)
        VV_TRY_FUSED(SF_B_VV1_SETTLE, SF_A_KE)
        VV_TRY_FUSED(SF_B_SCALE, SF_A_VV2_SETTLE)
        VV_TRY_FUSED(SF_B_VV1_EDL, SF_A_KE)
        VV_TRY_FUSED(SF_B_SCALE, SF_A_VV2_EDL)
        VV_TRY_FUSED(SF_B_VV1_EDL_SHAKE, SF_A_KE)
        VV_TRY_FUSED(SF_B_SCALE, SF_A_VV2_EDL_SHAKE)
        VV_TRY_FUSED(SF_B_COS_VV1_HW_MOM_F, SF_A_COS_MOM_VV1)
        VV_TRY_FUSED(SF_B_COS_SCALE_MOM_F, SF_A_COS_MOM_VV2)
        VV_TRY_FUSED(SF_B_COS_VV1_HW_MOM_F | B_SHAKE, SF_A_COS_MOM_VV1)
        VV_TRY_FUSED(SF_B_COS_SCALE_MOM_F, SF_A_COS_MOM_VV2_SHAKE)
    }
#undef VV_TRY_FUSED
#undef VV_FUSED_ONE
    // any other pair of stage sets (and every pair with VVHIP_RTC=2): compiled at run time from the library's own source
    if (rtc_mode() >= 1) {
        hipFunction_t f = rtc_kernel('B', precision, a.flags, a.chain.num_chains, a.flags_a);
        if (f) {
            if (blocks_per_cu) return hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, f, (int) b.x, lds);
            const int2* p_slots = a.slots; int p_nwaves = a.nwaves, p_wpb = (int) (b.x >> 6);
            const unsigned long long* p_acc = rendezvous; const NHDevState* p_nh = a.nh; const ChainLaneBlock* p_lc = a.lane_const; const int* p_sb = a.seg_base;
            KArgs copy = a;
            void* params[] = {&p_slots, &p_nwaves, &p_wpb, &p_acc, &p_nh, &p_lc, &p_sb, &copy};
            const hipError_t e = vv_launch_module(f, g, b, lds, s, ev0, ev1, params);
            vv_rtc_launches[1]++;
            if (route) *route = ROUTE_RUNTIME;
            return e;
        }
    }
    return hipErrorNotSupported;
#undef VV_PRE_ARGS
}
#endif      // VV_KERNELS_PART != 1
#if VV_KERNELS_PART != 2
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
#endif      // VV_KERNELS_PART != 2

}  // namespace vv
