// vv_args.hpp -- stage bits and argument blocks of the HIP kernels (gfx950): everything the device code (vv_device.inc) and its launchers
// share.  Compiled ahead of time into libvvhip.so and, for stage sets outside the compiled list, at run time by hipRTC (vv_rtc.cpp),
// which has no host headers: nothing in here may need more than the fixed-width integer types and the HIP vector types.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>

#include <cstdint>
#endif

#include "../../include/vvhip.h"
#include "vv_layout.h"

namespace vv {

// ---- stage flags of kernel A ("produce": ends in reductions) -----------------------------------
enum : uint32_t {
    A_FE_LOAD = 1u << 0,     // start the extra force from the forceExtra array (else 0)
    A_FE_STORE = 1u << 1,    // write the extra force back to the array
    A_LD = 1u << 2,          // + Langevin drag/noise              (K/drudeLangevin.cu:2-60)
    A_EF = 1u << 3,          // + electric field force             (K/electricField.cu:2-12)
    A_COS = 1u << 4,         // + cosine acceleration force        (K/cosineAccelerate.cu:2-14)
    A_KICK_FULL = 1u << 5,   // v += dt*invM*Fe + dt/2^32*invM*F   (K/middle.cu:6-23)
    A_KICK_HALF = 1u << 6,   // v += 0.5*dt*invM*Fe + fscale*invM*F (K/velocityVerlet.cu:6-29)
    A_POSDELTA_VV = 1u << 7, // posDelta = dt*v                    (K/velocityVerlet.cu:24-26)
    A_POS1 = 1u << 8,        // posDelta = oldDelta = dt/2*v       (K/middle.cu:29-42)
    A_BIAS = 1u << 9,        // accumulate sum m*vx*2cos(kz)       (K/cosineAccelerate.cu:16-61)
    A_KE = 1u << 10,         // molecular COM + per-group sum m v^2 (K/drudeNoseHoover.cu:5-151)
    A_UNBIAS_ACC = 1u << 11, // before the KE, subtract V*cos(kz) with V taken from accumulator 3
    A_COMPART = 1u << 12,    // molecules larger than a wave: add each chunk's sum(m v), sum(m) to the molecule's accumulator
    A_CZ_STORE = 1u << 13,   // keep cos(2 pi z / Lz) of every lane for the later kernels of this step (positions do not move in between)
    A_CZ_LOAD = 1u << 14,    // ... and take it from there instead of evaluating a double-precision cosine again
    A_SHAKE_V = 1u << 16,    // velocity constraints of the hydrogen-type clusters right after the kick (OpenMM applyVelocityConstraints)
    A_SETTLE = 1u << 22,     // ... of the rigid three-site molecules (one or both bits: what the plan holds)
    A_KE_MOM = 1u << 17,     // with A_BIAS | A_KE in ONE launch: the group sums as moments Saa, Sab, Sbb of the still biased velocities
                             // (accumulators 0-2, 4-6, 7-9); kernel B combines them once V is known: 2KE = Saa - 2 V Sab + V^2 Sbb
    A_MTAB = 1u << 18,       // own mass and Drude-pair mass fraction from the static per-lane tables (slot_m, slot_f) instead of
                             // reciprocals / IEEE divisions of velm.w in every step
    A_PERIODIC = 1u << 20,   // particle index, activity and segment index of a lane from the wave index (KArgs::per), role word from the pattern wave:
                             // no slot load in front of the particle loads (vv_host.hpp: PeriodicLayout)
    A_NOSTORE = 1u << 19,    // the kicked velocities stay in registers (KE stage) and are NOT written back: kernel B repeats the kick
                             // itself (B_KICK) from velm + force -- 24 bytes of force read there instead of 32 bytes written here and
                             // 3.5 MB less dirty data behind this launch at the headline size (the kernel boundary waits for it)
    A_SHAKE_GS = 1u << 21,   // hydrogen-type clusters by Gauss-Seidel sweeps of the central lane (OpenMM's iteration; VVHIP_SHAKE_MODE=0, generic
                             // kernel only) instead of the direct solve of the cluster's velocity constraints
    A_KE_PLAIN = 1u << 15,   // sum m v^2 over every massive particle into accumulator 0 (kinetic-energy query)
    A_GCONS = 1u << 23,      // velocity constraints of general clusters (any topology inside a wave): coloured Gauss-Seidel sweeps over the wave's list
};
// ---- stage flags of kernel B ("consume": starts from the scale factors) -------------------------
enum : uint32_t {
    B_SCALE = 1u << 0,        // v = s_atom*(v-V) + s_com*V, Drude pairs split (K/drudeNoseHoover.cu:157-209)
    B_UNBIAS = 1u << 1,       // remove the periodic bias before / restore after the scaling (fused)
    B_BIAS_REMOVE = 1u << 2,  // only vx -= V cos(kz)                (K/cosineAccelerate.cu:63-73)
    B_BIAS_RESTORE = 1u << 3, // only vx += V cos(kz)                (K/cosineAccelerate.cu:76-85)
    B_DRIFT_MIDDLE = 1u << 4, // fused Pos1+Pos2+Pos3 without constraints: x += dt/2*v_old + dt/2*v_new
    B_POS2 = 1u << 5,         // posDelta += dt/2*v; oldDelta += dt/2*v (K/middle.cu:47-60)
    B_POS3 = 1u << 6,         // v += (posDelta-oldDelta)/dt; x += posDelta (K/middle.cu:66-100)
    B_VV_KICK = 1u << 7,      // half kick from the forceExtra array + posDelta = dt*v (fused classic first half)
    B_VV_POS = 1u << 8,       // x += posDelta; v = posDelta/dt        (K/velocityVerlet.cu:35-68)
    B_HARDWALL = 1u << 9,     // (K/middle.cu:106-221)
    B_IMAGE = 1u << 10,       // mirror copy to the image particle     (K/imageCharge.cu:2-28)
    B_CHAIN = 1u << 11,       // run the NH chain in the kernel head from the accumulators (else read nh->scales)
    B_CZ_LOAD = 1u << 12,     // cos(2 pi z / Lz) from the per-lane cache written by kernel A (A_CZ_STORE)
    B_SHAKE = 1u << 13,       // position constraints of the hydrogen-type clusters on the step's displacement (OpenMM applyConstraints)
    B_SETTLE = 1u << 20,      // ... of the rigid three-site molecules
    B_MAILBOX = 1u << 14,     // multi-GPU mailbox: block 0's thermostat wave stores this rank's totals into every peer, all blocks sum all ranks' totals
    B_KE_MOM = 1u << 15,      // the accumulators hold moments (A_KE_MOM): combine them with V, unbias the stored COM velocities with comw
    B_KICK = 1u << 17,        // v += dt*invM*Fe + dt/2^32*invM*F (K/middle.cu:6-23) on the freshly loaded velocities: partner of A_NOSTORE, the
                              // very expression kernel A evaluated (same operands, same order: same bits); Fe = the cos force with B_UNBIAS
    B_PERIODIC = 1u << 18,    // as A_PERIODIC
    B_MTAB = 1u << 16,        // Drude-pair mass fractions from the static per-lane table (slot_f) instead of two IEEE divisions per lane
    B_SHAKE_GS = 1u << 19,    // as A_SHAKE_GS, for the position constraints (instead of the coupled Newton iteration)
    B_GCONS = 1u << 21,       // as A_GCONS, for the position constraints
    B_VSITE = 1u << 22,       // place the plan's virtual sites after the position update (integration.computeVirtualSites(), HOST:214, 374)
};
constexpr uint32_t A_CONS = A_SHAKE_V | A_SETTLE | A_GCONS, B_CONS = B_SHAKE | B_SETTLE | B_GCONS;      // in-kernel constraints of any kind
// ---- chain kernel --------------------------------------------------------------------------------
enum : uint32_t { C_CHAIN = 1u << 0, C_BIAS = 1u << 1 };

constexpr int NUM_ACC = 10;    // fixed-point quantities: 2KE atom, 2KE com, 2KE drude, bias moment; with the cos perturbation the
                               // cross moments Sab (4-6) and the field moments Sbb (7-9) of the three groups (A_KE_MOM)
constexpr int ACC_SLOTS = 256; // each quantity is spread over 256 int64 slots (block b adds into slot b % 256):
                               // thousands of blocks adding into ONE word serialise at ~10 ns per atomic on MI355X
                               // (measured: 17 us for kernel A at 1000 blocks); integer sums stay exact and
                               // order-independent, the chain wave folds the slots.  Layout acc[quantity][slot].

// Multi-GPU accumulator exchange without a collective launch ("mailbox", one node over xGMI).  Every rank owns an UNCACHED
// device allocation box[2 parities][ranks][MB_WORDS] of 8-byte words {sequence number : 32 | payload : 32} that all peers
// have mapped through hipIpc.  At the start of kernel B the thermostat wave of block 0 -- which has just folded the rank's
// accumulators, complete since kernel A ended -- writes the NUM_ACC int64 totals as 2*NUM_ACC words into slot [seq & 1][rank] of
// every OTHER rank's box (one 8-byte store each: payload and flag travel in the same atomic word, the LL idea of RCCL).  The
// thermostat wave of every block then polls the rank's own box until the other ranks' words carry the current sequence number
// and adds them to its own fold of the local accumulators -- integers, so all ranks continue with identical bits.  Nothing is added to kernel A, no launch is
// added to the step.  Two parities suffice: a rank can only be one exchange ahead of the slowest one (it needs everybody's
// words of exchange n to finish its kernel B of exchange n).  The sequence number lives in the double-buffered NHDevState.
constexpr int MB_WORDS = 2 * NUM_ACC;
constexpr int MB_MAX_RANKS = 16;
struct Mailbox {
    unsigned long long* local;            // this rank's box
    unsigned long long* const* peers;     // device array [ranks] of every rank's box as mapped here (own entry = local)
    unsigned int* ctl;                    // [0] set when a wait on the peers ran out
    int32_t ranks, rank;
};

// Device-resident thermostat state (reference keeps it on the host: CudaVVKernels.h:206-215)
struct NHDevState {
    vvhip_nh_state s;
    double scales[4];        // what kernel B consumes: vscale[3] and the periodic bias V
    unsigned int mb_seq;     // mailbox exchanges done so far (advanced with the state by a B_MAILBOX launch)
    unsigned int rv_seq;     // fused steps done so far: the tag of the rendezvous words (advanced with the state by a fused launch)
    unsigned int rv_delay;   // fused step: units of 256 clocks a block waits between its publish and its first poll round (self-tuning, vv_device.inc)
    unsigned int rv_calm;    // ... and fused steps since a block last needed a second round
};

// Constants of the chain (HOST:577-594 fixed at init; temperatures read live as API:728 does)
struct NHConst {
    double eta_mass[VVHIP_NUM_TG][VVHIP_MAX_CHAINS];
    double inv_eta_mass[VVHIP_NUM_TG][VVHIP_MAX_CHAINS];   // 1/eta_mass (0 where the mass is 0): the chain multiplies instead of dividing
    double nkbt[VVHIP_NUM_TG];
    double temperature[VVHIP_NUM_TG];
    double step_size;
    double inv_mass_total;
    double acc_inv_scale[NUM_ACC];
    int32_t num_chains, loops_per_step, num_tg;
    uint32_t flags;
};

// Chain constants of one temperature group as the thermostat wave of kernel B loads them (lane g reads row g with ordinary vector
// loads at its very top, next to the state).  Picking lane g's row out of the kernel-argument block instead compiles to lane-indexed
// loads from the kernarg segment issued after the fold, whose latency sat on the kernel's critical path (timeline in DESIGN.md §7).
struct ChainLaneBlock {
    double eta_mass[4], inv_eta_mass[4];
    double nkbt, kT, acc_inv_scale, active;     // active != 0: the group is thermostatted (HOST:729)
    double dt2, dt4, dt8, pad_;
};

// PeriodicLayout as the kernels take it: per quantity the value of region 0 and the INCREMENTS from region to region, so that the
// value of a wave's region is base + sum of the increments of the region starts at or below the wave -- selects between a loaded
// scalar and zero.  (Selecting among the array elements themselves makes the compiler index the argument block dynamically, which
// it can only do through scratch memory: measured, kernel A 4.2 -> 5.7 us.)
struct PeriodicArgs {
    uint32_t magic;
    int32_t wpc, apc, spc;
    int32_t wave_start[4];                                    // absolute (cell-local); unused regions: INT_MAX
    int32_t d_wave[4], d_atom_start[4], d_atom_end[4], d_seg[4], d_P[4], d_spw[4];
};
// One argument block for kernels A and B (passed by value); pointer types are erased so that the
// same struct serves the three precision modes.
struct KArgs {
    // ---- first 192 bytes: what every wave needs for its first loads and kernel A for its last (three scalar-cache lines)
    const int2* slots;
    void* velm;
    void* posq;
    void* corr;
    const long long* force;
    void* comv;                    // mixed4 [num segments]: COM velocity of each COM segment (dense index, see seg_base), written by A_KE, read by B_SCALE
    unsigned long long* acc;        // accumulators of the current parity (A adds, B consumes)
    const NHDevState* nh;           // thermostat state of the current parity
    const ChainLaneBlock* lane_const;   // [VVHIP_NUM_TG] chain constants, one row per group, in device memory (kernel B's thermostat wave)
    int32_t padded;
    int32_t nwaves;
    uint32_t flags;
    uint32_t random_index;
    PeriodicArgs per;          // arithmetic work-item layout (A_PERIODIC / B_PERIODIC; vv_host.hpp: PeriodicLayout)
    double dt;                 // step size
    double inv_dt_mixed;       // 1 / dt evaluated in the mode's `mixed` type on the host (K/middle.cu:71), widened
    double inv_dt_double;      // 1.0 / dt in double (K/velocityVerlet.cu:43)
    double max_drude, hw_scale;                    // HOST:189-190
    double acc_scale[NUM_ACC];  // fixed-point scales of the accumulated quantities (kernel A's tail)
    // ---- the rest
    void* fextra;
    void* pos_delta;
    void* old_delta;
    double* cosz;                  // [64*nwaves] per-lane cos(2 pi z / Lz) cache of the current step
    const double* seg_mass;        // [2*num segments] (mass, 1/mass) of each COM segment (static; vv_host.hpp)
    const int* seg_base;           // [nwaves+1] segments in the waves before this one: segment index of a lane = seg_base[wave] + leader lanes below it
    double* comw;                  // [num segments] mass-weighted mean of cos(kz) over the same segment (A_KE_MOM -> B_KE_MOM)
    const double* slot_m;           // [64*nwaves] RECIP(velm.w) of the lane's particle in the mode's `mixed` type, widened (0 = massless / idle)
    const double* slot_f;           // [64*nwaves] Drude-pair lanes: invTotalMass * own mass (K/drudeNoseHoover.cu:173-180), else 0
    const int32_t* slot_image;
    const int32_t* slot_rand;
    const int32_t* slot_shake;      // packed cluster word per lane (vv_host.hpp: SHAKE_WORD_*), NULL without in-kernel constraints
    const float4* slot_shake_param; // every lane of a cluster: 1/m_c, 0.5/(1/m_c+1/m_p), d^2, 1/m_p (SETTLE: the two distances)
    double shake_tol;
    const int32_t* slot_big;        // big-molecule index per lane (only with molecules larger than a wave)
    unsigned long long* bigacc;     // int64 fixed point [num_big][4]: sum m vx, m vy, m vz, m
    double big_scale, big_inv_scale;
    const float4* random;
    unsigned long long* acc_next;   // other parity: zeroed by B when it runs the chain inline
    NHDevState* nh_next;            // where an inline chain writes the advanced state
    int32_t acc_rows;                  // accumulator rows in use (4, or NUM_ACC with the cos moments): what kernel B has to clear
    int32_t acc_exclusive;             // kernel A: 1 = a launch of <= ACC_SLOTS blocks stores into its slots instead of adding atomically (launch_a clears it for larger grids)
    double fscale_vv;          // 0.5*dt/2^32 computed in double on the host (HOST:306)
    double drag, randf, drag_drude, randf_drude;   // HOST:835-839
    double efscale;            // E * AVOGADRO (HOST:978)
    double cos_accel;
    double inv_box_z;
    double mirror;
    double inv_mass_total;
    double acc_inv_scale[NUM_ACC];
    NHConst chain;                  // chain constants (used by B_CHAIN)
    Mailbox mb;                     // B_MAILBOX
    unsigned int* status;           // pinned host words, system-scope stores: [0] mailbox wait timed out, [1] accumulator overflow (sticky)
    long long* dbg;                 // timestamp buffer of the instrumented build (-DVV_KERNEL_TIMESTAMPS, tools/probes), else unused
    int32_t dbg_block, dbg_pad_;
    long long* dbg_span;            // instrumented build: [2 launch parities][blocks*8 waves][2] entry / exit stamps of every wave (100 MHz wall clock)
    int32_t dbg_parity, dbg_pad2_;
    uint32_t pos_bytes;             // size of the posq (= posqCorrection) array in bytes if below 4 GB, else 0: kernel A fetches the positions of
    int32_t gc_colors;              // constraint-cluster members through a buffer resource (load_wanted), other lanes fetch nothing; gc_colors:
                                    // colours of the general clusters' sweeps (A_GCONS / B_GCONS)
    double gc_omega;                // A_GCONS / B_GCONS: relaxation factor of the sweeps (vv_layout.h: GC_OMEGA_*)
    const int2* slot_vsite;         // B_VSITE: [64*waves] (site word, vv_layout.h: VS_WORD_*; record number) of the site the lane places
    const double* vsite_params;     // B_VSITE: [12*records] weights / local position of each site (vvhip_system_desc.virtual_site_params)
    const int32_t* vsite_atom;      // B_VSITE: [records] particle index of the site: a lane that places a site for its parent stores it there
    uint32_t flags_a;               // fused step (vv_kernel_b<.., SFA>): kernel A's stage set executed by the same launch, 0 otherwise (launch_b dispatches on it)
    int32_t fused_poll_delay;       // fused step: >= 0 pins the wait between a block's publish and its first poll round (units of 256 clocks); -1: NHDevState::rv_delay
    unsigned int* rv_late_cur;      // fused step: [ACC_SLOTS] uncached words, block b's = 1 if it needed a second poll round in THIS step ...
    const unsigned int* rv_late_prev;   // ... and the previous fused step's
    int32_t fused_late_shift, fused_pad_;   // the wait grows when more than blocks >> shift blocks were late
};

struct TetherArgs {
    const void* posq;
    const void* site;
    const void* velm;
    long long* force;
    const int2* slots;
    int32_t padded, nwaves;
    double k_tether, k_drude;
    long long* dbg_span;            // instrumented build (as KArgs::dbg_span)
    int32_t dbg_parity, dbg_pad_;
};

}  // namespace vv
