// vv_kernels.hpp -- launch entry points of the HIP kernels (gfx950).
// Device code lives in vv_device.inc (compiled through vv_kernels.hip); the C ABI (vv_api.cpp) only sees this header.
#pragma once
#include <atomic>
#include "vv_args.hpp"
#include "vv_host.hpp"

namespace vv {

inline PeriodicArgs periodic_args(const PeriodicLayout& q) {
    PeriodicArgs r{};
    r.magic = q.magic; r.wpc = q.wpc; r.apc = q.apc; r.spc = q.spc;
    for (int k = 0; k < 4; k++) {
        const bool used = k < q.nreg;
        r.wave_start[k] = used ? q.reg_wave_start[k] : 0x7fffffff;
        auto inc = [&](const int32_t* v) { return !used ? 0 : (k == 0 ? v[0] : v[k] - v[k - 1]); };
        r.d_wave[k] = inc(q.reg_wave_start); r.d_atom_start[k] = inc(q.reg_atom_start); r.d_atom_end[k] = inc(q.reg_atom_end);
        r.d_seg[k] = inc(q.reg_seg_start); r.d_P[k] = inc(q.reg_P); r.d_spw[k] = inc(q.reg_spw);
    }
    return r;
}

// launchers (precision = VVHIP_SINGLE / MIXED / DOUBLE); return hipError_t of the launch
// block_threads = 64 x tile waves per block; grid_cap = most blocks to launch (the kernels stride over tiles beyond that)
// ev0 / ev1 (optional): events that receive the dispatch's own begin / end timestamps (timing runs; never inside a graph capture)
// route (optional): which kernel the launch took -- ROUTE_COMPILED (a specialisation of the library), ROUTE_RUNTIME (hipRTC),
// ROUTE_GENERIC (run-time stage bits) -- so that the caller can count per plan (several host threads may drive plans of their own)
enum LaunchRoute { ROUTE_COMPILED = 0, ROUTE_RUNTIME = 1, ROUTE_GENERIC = 2 };
hipError_t launch_a(int precision, const KArgs& a, int block_threads, int grid_cap, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int* route = nullptr);
hipError_t launch_b(int precision, const KArgs& a, int block_threads, int grid_cap, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int* route = nullptr);
// The fused step: kernel A's stage set a.flags_a and kernel B's a.flags in one launch of vv_kernel_b<.., SFA> over `rendezvous` (an uncached
// [NUM_ACC][ACC_SLOTS] array of 8-byte words); block_threads = 64 x tile waves per block, one pass.  blocks_per_cu != nullptr: only report the
// kernel's occupancy (blocks of this shape per CU).  hipErrorNotSupported: no kernel for the pair, or a shape the rendezvous cannot take.
hipError_t launch_fused(int precision, const KArgs& a, int block_threads, const unsigned long long* rendezvous, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                        int* route = nullptr, int* blocks_per_cu = nullptr);
// launches of kernel A / B (index 0 / 1) that ran the generic kernel, process-wide (statistics only)
extern std::atomic<unsigned long long> vv_generic_count[2];
hipError_t launch_chain(const NHConst& c, NHDevState* st, unsigned long long* acc, hipStream_t s);
hipError_t launch_tether(int precision, const TetherArgs& t, int block_threads, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// Fills the static per-lane mass tables from the inverse masses in velm.w (once per binding; see A_MTAB / B_MTAB).
hipError_t launch_mass_table(int precision, const void* velm, const int2* slots, int nwaves, double* slot_m, double* slot_f, hipStream_t s);
// whether the specialised kernels A (kernel = 0) / B (1) of this build were compiled with the mass tables (they then carry A_MTAB /
// B_MTAB in their stage sets)
bool sf_kernels_use_mass_table(int kernel);
// Device Gaussian generator (stand-alone hosts; inside OpenMM the random buffer is OpenMM's): Philox4x32-10 keyed by
// `seed`, counter = (*epoch, element index); a second 1-thread launch bumps *epoch so that graph replays draw fresh numbers.
hipError_t launch_fill_normals(float4* out, uint32_t count, uint64_t seed, unsigned long long* epoch, hipStream_t s);
hipError_t launch_image_pairs(int precision, void* posq, void* corr, const int2* pairs, int npairs, double mirror,
                              hipStream_t s);

}  // namespace vv
