// vv_api.cpp -- the C ABI of include/vvhip.h on top of vv_host (analysis) and vv_kernels (HIP).
// "HOST" = platforms/cuda/src/CudaVVKernels.cpp, "API" = openmmapi/src/VVIntegrator.cpp of the reference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is resolved at run time (see rccl_api)

#include "vv_host.hpp"
#include "vv_kernels.hpp"
#include "vv_rtc.hpp"

namespace {
// RCCL is looked up lazily so that single-GPU users never need it.  If the process already has a librccl (PyTorch
// brings its own) that copy is used -- two RCCL builds in one process is asking for trouble.
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclAllReduce) allReduce = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclCommCount) commCount = nullptr;
    decltype(&ncclGetErrorString) getErrorString = nullptr;
    bool ok = false;
};
RcclApi& rccl_api() {
    static RcclApi r;
    if (r.handle) return r;
    r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!r.handle) r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) return r;
    r.getUniqueId = (decltype(r.getUniqueId)) dlsym(r.handle, "ncclGetUniqueId");
    r.commInitRank = (decltype(r.commInitRank)) dlsym(r.handle, "ncclCommInitRank");
    r.allReduce = (decltype(r.allReduce)) dlsym(r.handle, "ncclAllReduce");
    r.commDestroy = (decltype(r.commDestroy)) dlsym(r.handle, "ncclCommDestroy");
    r.commCount = (decltype(r.commCount)) dlsym(r.handle, "ncclCommCount");
    r.getErrorString = (decltype(r.getErrorString)) dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.getUniqueId && r.commInitRank && r.allReduce && r.commDestroy;
    return r;
}

constexpr double kAvogadro = 6.02214076e23;
constexpr double kBoltz = (1.380649e-23 * kAvogadro) / 1000.0;
enum TimerClass { T_A = 0, T_B = 1, T_OTHER = 2 };
}  // namespace

// roctx ranges around every launch group (rocprofv3 --marker-trace): resolved lazily, only if VVHIP_ROCTX=1 or vvhip_set_trace(plan, 1)
struct RoctxApi {
    bool tried = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
RoctxApi& roctx_api() {
    static RoctxApi r;
    if (r.tried) return r;
    r.tried = true;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        r.push = (int (*)(const char*)) dlsym(h, "roctxRangePushA");
        r.pop = (int (*)()) dlsym(h, "roctxRangePop");
        if (r.push && r.pop) break;
        r.push = nullptr; r.pop = nullptr;
    }
    return r;
}

struct vvhip_plan {
    vv::HostPlan hp;
    std::string err;
    bool bound = false;
    vvhip_buffers buf{};
    hipStream_t stream = nullptr;
    double box[3] = {1, 1, 1};
    double acc_scale[vv::NUM_ACC], acc_inv_scale[vv::NUM_ACC];
    int block_threads = 256;       // 64 x tile waves per block, the same for the force provider, kernel A and kernel B
    int grid_cap_a = 2048, grid_cap_b = 1024;   // most blocks per launch (multiples of the CU count): see pick_launch_shape
    int split_chain_waves = 44000;   // systems with at least this many waves (~2.6 M particles) run the chain as its own launch.  Round 4, with two blocks of seven tile
                                     // waves per CU below it (profiles/r04zd_mid_sizes.txt; chain in kernel B | own launch, steps/s): 888 k particles 20.8 | 18.9 k,
                                     // 1.33 M 14.5 | 14.0 k, 1.78 M 11.4 | 11.1 k, 2.66 M 7.38 | 7.37 k, 4.4 M 4.14 | 4.20 k, 8.9 M 2.16 | 2.21 k (round 2 had it at 12 288)
    bool trace = false;            // roctx range + one stderr line per launch group (the reference's setDebugEnabled, VVIntegrator.h:417-419)
    bool fextra_dirty = false;     // forceExtra holds something since the last reset (split entry points)
    // The reference's kick kernels add forceExtra ALWAYS (K/middle.cu:11-21, K/velocityVerlet.cu:20-22) and the array is only reset in
    // steps that have a source of extra forces (API:238-240, 316-318): once the cos acceleration is set to 0 in a run without Langevin
    // particles or a field, the last cos force stays in forceExtra and every later kick keeps adding it.  The fused middle step computes
    // extra forces on the fly and leaves the array alone; `fextra_virtual` says the array SHOULD hold the cos force of the last fused
    // step.  vvhip_set_params materialises it (kernel A from the cached cos(kz)) when the acceleration goes to 0, and a fused kick
    // without sources loads the array whenever it is dirty -- the reference's behaviour to the bit, quirk included.
    bool fextra_virtual = false;
    bool fextra_external = false;  // the host asked for the pointer (vvhip_force_extra) and may write to it: never assume zeros
    bool no_moments = false;       // test hook "no_moments": keep the three-launch cos sequence (comparison runs)
    // with an arithmetic work-item layout (HostPlan::per) the kernels compute particle indices instead of loading slot words (test hook "periodic_kernels" = 0:
    // comparison runs).  Kernel A: no slot traffic (1.13 -> 1.0 x the algorithmic bytes) and the next tile's loads in flight during this tile's
    // arithmetic: 133 vs 138 us in sequence at 8.9 M particles (round 2 without the second tile in flight: 113.6 vs 115.7 back to back);
    // test hook "periodic_a" = 0 switches it off
    bool periodic_kernels = true, periodic_a = true, periodic_b = true;
    int shake_mode = 1;            // hydrogen-type constraint clusters: 1 = all constraints of a cluster at once (direct velocity solve, coupled Newton
                                   // for positions), 0 = Gauss-Seidel sweeps by the central lane (OpenMM's iteration; generic kernels) -- VVHIP_SHAKE_MODE
    // Race detection by timing (VVHIP_STALL=us[:period]): every period-th launch of this plan is preceded by a host sleep of `us` microseconds --
    // the GPU drains, anything that was only ordered by the depth of the queue (a fill or copy on another stream, a host read without a
    // synchronisation) lands differently, and the trajectory changes.  tests/test_gpu_stalls.py compares stalled and unstalled runs bit for bit.
    long stall_us = 0, stall_period = 1, stall_count = 0;
    bool acc_store = true;         // kernel A launches of <= 256 blocks store old + new into their accumulator slots instead of atomics (test hook "acc_store" = 0: atomics)
    long long generic_launches[2] = {0, 0};   // kernel A / B launches of this plan (captured ones count once) that ran the generic kernel
    uint32_t generic_flags[2] = {0, 0};       // ... and the last stage set that did (vvhip_generic_launches)
    std::vector<uint32_t> generic_seen[2];    // every stage set that did (VVHIP_WARN_GENERIC prints each once)
    bool rekick = true;            // fused middle step: kick repeated in kernel B instead of a velm store in kernel A (use_rekick)
    // One launch per step (vv_device.inc: "fused step"): kernels A and B of the middle scheme as one launch of co-resident blocks around an
    // in-kernel rendezvous.  `fused` = allowed (vvhip_debug_tune "fused": A/B comparisons and the bit-for-bit tests switch it off);
    // d_rv = the rendezvous words, [2 thermostat parities][NUM_ACC][ACC_SLOTS], uncached; fused_checked_* = the last pair of stage sets /
    // launch shape whose kernel and occupancy were looked up, fused_ok = what came of it.
    bool fused = true;
    // the wait grows when more than blocks / 2^shift blocks needed a second round (test hook "fused_late_shift").  1/16 of the blocks (shift 4,
    // the first choice) let the wait climb where the blocks finish their front unevenly (constraint clusters: 11 units against the best pinned 7);
    // half of them: C3 + HBonds 80.2 -> 81.9 k steps/s, C4 83.1 -> 84.2 k, C5 + HBonds 96.0 -> 97.4 k, C2 150.4 -> 152.4 k, C3 / C5 + 0.4 %
    // (profiles/r05s_late_shift_scan.txt)
    int fused_late_shift = 1;
    int fused_poll_delay = -1;     // >= 0: pins the wait between a block's publish and its first poll round, units of 256 clocks (test hook "fused_poll_delay"); -1: self-tuning
    unsigned long long* d_rv = nullptr;
    uint32_t fused_checked_a = 0, fused_checked_b = 0;
    int fused_checked_threads = 0, fused_checked_waves = 0;
    bool fused_ok = false;
    struct FusedCheck { uint32_t a = 0, b = 0; int threads = 0, waves = 0; bool ok = false; };
    FusedCheck fused_checks[4];    // (the classic scheme alternates between the pairs of its two halves)
    int fused_check_next = 0;
    long long fused_launches = 0;
    // Recovery from a missed rendezvous (round 6).  The one-launch step needs its blocks resident together; another process's kernel on the
    // device can break that, the blocks' bounded wait then runs out (sticky word [2]) and the step -- and every step enqueued behind it -- has
    // worked on incomplete sums.  The plan-driven loops (vvhip_run_graph / vvhip_run_eager) therefore keep a device-side copy of the physical
    // state from the entry of the first run call that is not yet known to have ended well (positions, correction, velocities, forces, extra
    // forces, both thermostat copies, the random generator's epoch: 116 B per particle in mixed precision, taken only for calls of at least
    // `min_steps` steps) together with the run calls since; the next vvhip_synchronize that finds word [2] raised puts the copy back, pins
    // the plan to two launches per step (bit for bit the same step), repeats the calls and says so once on stderr.  No multi-GPU exchange in
    // between (the other ranks would have to repeat theirs).  `vvhip_debug_tune(plan, "recover", 0)` / VVHIP_RECOVER=0 switch it off.
    struct Recovery {
        bool enabled = true, valid = false, replaying = false, in_loop = false;
        int min_steps = 64;
        void *posq = nullptr, *corr = nullptr, *velm = nullptr, *force = nullptr, *fextra = nullptr, *random = nullptr;
        vv::NHDevState* nh = nullptr;
        unsigned long long* epoch = nullptr;
        int parity = 0;
        uint32_t random_pos = 0;
        bool fextra_dirty = false, fextra_virtual = false;
        struct Run { int kind, nsteps, spg; const void* site; double kt, kd; };
        std::vector<Run> runs;
        long long recoveries = 0;
    } rec;
    // plan-owned device state
    int2* d_slots = nullptr;
    int32_t* d_slot_image = nullptr;
    int32_t* d_slot_rand = nullptr;
    int32_t* d_slot_big = nullptr;
    int32_t* d_slot_shake = nullptr;
    float4* d_slot_shake_param = nullptr;
    int2* d_slot_vsite = nullptr;
    double* d_vsite_params = nullptr;
    int32_t* d_vsite_atom = nullptr;
    unsigned long long* d_bigacc = nullptr;
    int2* d_image_pairs = nullptr;
    void* d_fextra = nullptr;
    void* d_old_delta = nullptr;
    void* d_pos_delta = nullptr;   // used when the caller does not supply one
    void* d_comv = nullptr;        // per-segment COM velocities handed from kernel A to kernel B
    double* d_slot_m = nullptr;    // static per-lane RECIP(velm.w) (vv_args.hpp: A_MTAB), filled on the device from velm.w
    double* d_slot_f = nullptr;    // static per-lane Drude-pair mass fraction (A_MTAB / B_MTAB)
    bool mass_tab_a = false, mass_tab_b = true;   // kernel A / B launches read the tables (defaults follow the build; test hooks "mass_tab_a" / "mass_tab_b" override: comparison runs)
    bool mass_tab_valid = false;   // tables match the bound velm.w (vvhip_bind / vvhip_masses_changed reset it)
    double* d_seg_mass = nullptr;  // static (mass, 1/mass) per COM segment
    int* d_seg_base = nullptr;     // per wave: COM segments in the waves before it
    double* d_comw = nullptr;      // per-segment mass-weighted mean of cos(kz) (moment form of the cos perturbation)
    double* d_cosz = nullptr;      // per-lane cos(2 pi z / Lz) of the current step
    unsigned long long* d_acc = nullptr;   // [2 parities][NUM_ACC][ACC_SLOTS]
    vv::NHDevState* d_nh = nullptr;         // [2 parities]
    int parity = 0;                         // which copy the next reduction/consumer pair uses
    unsigned long long* d_epoch = nullptr;  // refill counter of the device Gaussian generator
    uint64_t rng_seed = 0;
    uint32_t random_pos = 0;                // prepareRandomNumbers cursor for the plan-driven loops (vvhip_run_*)
    // HIP-event timing (eager launches only)
    bool timing = false;
    bool timing_kernels_only = false;   // vvhip_timing_enable(plan, 2): dispatch timestamps of kernels A and B only, nothing added to the stream
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events[3];
    std::vector<hipEvent_t> event_pool;  // events of earlier timing sessions, reused (hipEventCreate per launch would make the host the bottleneck)
    // captured graph for vvhip_run_graph
    // One executable per thermostat parity (the state is double-buffered by step parity, so a graph captured at parity q only
    // replays correctly when the plan is at parity q again).  Captured by vvhip_graph_prepare / the first vvhip_run_graph that
    // needs it, never re-captured while the key (steps, force provider) is unchanged: a run that alternates eager tails and
    // replays keeps both.
    struct GraphSlot {
        hipGraphExec_t exec = nullptr;
        int steps = 0;
        const void* site = nullptr;
        double kt = 0, kd = 0;
        uint32_t random_end = 0;               // prepareRandomNumbers cursor after the graph's last step
    };
    // (round 6: up to four graph lengths per parity -- a host that replays a short graph in front of a long one keeps both)
    static constexpr int kGraphWays = 4;
    GraphSlot graph[2][kGraphWays];
    int graph_next[2] = {0, 0};
    bool capturing = false;
    // particle sharding over GPUs: RCCL communicator for the accumulator exchange (null = single GPU)
    ncclComm_t comm = nullptr;
    int comm_ranks = 1;
    // ... or the xGMI mailbox (vv_args.hpp: Mailbox): no collective launch, works inside a captured graph
    unsigned long long* mb_local = nullptr;       // uncached, exported through hipIpc
    unsigned long long** d_mb_peers = nullptr;    // device array of the peers' mappings
    unsigned int* d_mb_ctl = nullptr;
    std::vector<void*> mb_opened;                 // hipIpcOpenMemHandle mappings to close
    int mb_ranks = 0, mb_rank = 0;
    bool mb_on = false;
    // A peer's box lives on THIS device (several ranks sharing one GPU: test set-ups): found out by vvhip_mailbox_connect.  Such ranks'
    // kernels compete for the same CUs, and device-filling grids of polling thermostat waves keep the other process's kernels off
    // the device until the bounded waits run out: every rank then launches on its share of the CUs (shared_device_cap).
    bool mb_shared_device = false;
    int dbg_seq = -1;              // instrumented build: >= 0 while vvhip_debug_step_spans numbers the launches of its steps
    int mb_device_ranks = 1;       // ranks whose boxes live on this device (this one included)
    // Sticky health word in pinned host memory, written by the kernels with system-scope stores when something goes wrong and
    // read by the host without synchronising: [0] a mailbox wait on the peers ran out (the ranks have diverged), [1] a fixed-point
    // accumulator left its range (|sum| x scale >= 2^62: the thermostat would see garbage).  Checked at the entry of the run loops
    // and in vvhip_synchronize / vvhip_status.
    unsigned int* h_status = nullptr;
    unsigned int* d_status = nullptr;             // the same words as the device sees them
    bool launch_shape_forced = false;             // a test hook fixed the launch shape ("block_threads", "grid_cap_a / b"): keep it at bind
    int num_cus = 256;                            // hipDeviceProp_t::multiProcessorCount of the bound device
    vv::ChainLaneBlock* d_lane_const = nullptr;   // [3] chain constants per temperature group (kernel B's thermostat wave)
    vv::ChainLaneBlock lane_const_host[VVHIP_NUM_TG] = {};
    bool lane_const_valid = false;
    long long* d_dbg_span = nullptr;
    int dbg_parity = 0;
    long long* d_dbg = nullptr;                   // instrumented build only (vvhip_debug_timestamps)
    int dbg_block = 0;
};

static void drop_graphs(vvhip_plan* p) {
    for (auto& row : p->graph)
        for (auto& g : row)
            if (g.exec) { (void) hipGraphExecDestroy(g.exec); g.exec = nullptr; }
}
// the slot of parity q that holds (or will hold) the graph with this key
static vvhip_plan::GraphSlot& graph_slot(vvhip_plan* p, int q, int steps, const void* site, double kt, double kd) {
    auto& row = p->graph[q & 1];
    for (auto& g : row)
        if (g.exec && g.steps == steps && g.site == site && g.kt == kt && g.kd == kd) return g;
    for (auto& g : row)
        if (!g.exec) return g;
    return row[p->graph_next[q & 1]++ % vvhip_plan::kGraphWays];
}

namespace vv { unsigned vv_last_grid_value = 0; }
static unsigned vv_last_grid() { return vv::vv_last_grid_value; }
extern "C" int vvhip_debug_read_accumulators(vvhip_plan* p, double out[4], int zero_after);

namespace {

int fail(vvhip_plan* p, int code, const std::string& msg) {
    if (p) p->err = msg;
    return code;
}
int recover_rendezvous(vvhip_plan* p);
// Work on the plan from OUTSIDE the plan-driven loops while their snapshot is still unverified (a split entry point, a parameter change, a
// fill): the calls since the snapshot are settled first -- synchronised and, if their rendezvous failed, repeated -- because what comes
// now is not in the list a recovery would repeat.
int settle_recovery(vvhip_plan* p) {
    if (!p->rec.valid || p->rec.in_loop || p->rec.replaying || p->capturing) return VVHIP_OK;
    return vvhip_synchronize(p);
}
// Sticky failures the kernels reported through the pinned status word (no synchronisation: the word lives in host memory).
int check_exchange_health(vvhip_plan* p) {
    if (!p->h_status) return VVHIP_OK;
    const unsigned int mb = __atomic_load_n(&p->h_status[0], __ATOMIC_RELAXED), ov = __atomic_load_n(&p->h_status[1], __ATOMIC_RELAXED);
    const unsigned int rv = __atomic_load_n(&p->h_status[2], __ATOMIC_RELAXED), cs = __atomic_load_n(&p->h_status[3], __ATOMIC_RELAXED);
    if (mb) return fail(p, VVHIP_ERR_EXCHANGE, "multi-GPU mailbox: a wait on the peers' thermostat totals timed out; this rank went on with incomplete sums, the run is void");
    // (a missed rendezvous first: an overflowed accumulator or an unconverged cluster next to it is what steps on incomplete sums produce)
    if (rv) {
        // (the plan is pinned to two launches per step from here on whatever else happens: the next run does not meet the same fate)
        if (p->fused) {
            p->fused = false;
            if (p->bound) (void) hipStreamSynchronize(p->stream);      // (the word is read without synchronising: replays of the graphs about to go may still be in flight)
            drop_graphs(p);
        }
        return fail(p, VVHIP_ERR_RENDEZVOUS, "fused step: the blocks of the one-launch step did not meet within 0.2 s (not resident together: another process's kernels on the device?); the thermostat went on with incomplete sums and the state since the last good synchronisation is void.  The plan now takes two launches per step (vvhip_fused_status: active = 0); vvhip_status_clear + restoring the state continues the run.  Runs of >= 64 steps through vvhip_run_graph / vvhip_run_eager recover by themselves (vvhip_debug_tune \"recover\")");
    }
    // (an unconverged constraint cluster is reported before the overflow it usually causes a step or two later)
    if (cs) return fail(p, VVHIP_ERR_CONSTRAINT, "in-kernel constraints: a cluster reached the iteration cap without converging (a degenerate geometry, or a step that is too large); positions / velocities of that cluster are not within tolerance");
    if (ov) return fail(p, VVHIP_ERR_OVERFLOW, "a fixed-point accumulator overflowed (kinetic energy beyond 1024 x the thermostat target): the thermostat input is invalid");
    return VVHIP_OK;
}
int hip_fail(vvhip_plan* p, hipError_t e, const char* what) {
    return fail(p, e == hipErrorNoDevice || e == hipErrorInvalidDevice ? VVHIP_ERR_NO_DEVICE : VVHIP_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(p, call)                                         \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return hip_fail(p, e_, #call);     \
    } while (0)
#define TRY(x)                       \
    do {                             \
        int rc_ = (x);               \
        if (rc_ != VVHIP_OK) return rc_; \
    } while (0)
#define NEED_BOUND(p)                                                                  \
    do {                                                                               \
        if (!(p)) return VVHIP_ERR_INVALID;                                            \
        if (!(p)->bound) return fail(p, VVHIP_ERR_INVALID, "vvhip_bind has not been called"); \
    } while (0)

size_t sizeof_real(int prec) { return prec == VVHIP_DOUBLE ? 8 : 4; }
size_t sizeof_mixed(int prec) { return prec == VVHIP_SINGLE ? 4 : 8; }

// 2^k fixed-point scale leaving `headroom` x `bound` below 2^62
double pick_scale(double bound, double headroom) {
    double top = std::ldexp(1.0, 62) / (std::max(bound, 1.0) * headroom);
    int k = (int) std::floor(std::log2(top));
    k = std::max(0, std::min(k, 40));
    return std::ldexp(1.0, k);
}

void fill_scales(vvhip_plan* p) {
    // One common power-of-two scale for the three 2KE sums, sized on the TOTAL thermostat target with 1024x
    // headroom: a cold group (Drude, 1 K) may transiently be orders of magnitude hotter than its own target
    // without getting anywhere near overflow, and resolution stays ~1e-13 of even the smallest group.
    const vvhip_plan_info& in = p->hp.info;
    const double total = in.nkbt[0] + in.nkbt[1] + in.nkbt[2];
    for (int g = 0; g < 3; g++) {
        p->acc_scale[g] = pick_scale(total, 1024.0);
        p->acc_inv_scale[g] = 1.0 / p->acc_scale[g];
    }
    p->acc_scale[3] = pick_scale(40.0 / in.inv_mass_total, 4.0);  // |sum m vx 2cos| <= 2 M |v|max, |v|max ~ 20 nm/ps
    p->acc_inv_scale[3] = 1.0 / p->acc_scale[3];
    // moments of the cos perturbation (A_KE_MOM): Sbb = sum m b^2 <= M (|cos| <= 1, |b| <= 2), |Sab| <= sqrt(Saa Sbb)
    const double mass = 1.0 / in.inv_mass_total;
    for (int g = 0; g < 3; g++) {
        p->acc_scale[4 + g] = pick_scale(std::sqrt(total * 1024.0 * 4.0 * mass), 4.0);
        p->acc_scale[7 + g] = pick_scale(4.0 * mass, 4.0);
        p->acc_inv_scale[4 + g] = 1.0 / p->acc_scale[4 + g];
        p->acc_inv_scale[7 + g] = 1.0 / p->acc_scale[7 + g];
    }
}

// Launch shape.  Measured on MI355X (256 CUs): what matters at the latency-bound sizes is that every CU gets the SAME number of
// blocks -- a CU with one block more than its neighbours finishes ~1.3 us later (its thermostat waves share the fp64 pipe), and the
// kernel ends with its slowest CU.  C3, 1 752 tiles: 876 blocks of 2 tiles (3.4 per CU) 69.4 k steps/s; 251 blocks of 7 tiles (one
// per CU) 74.3 k.  So: k blocks per CU, T tile waves per block (+1 thermostat wave in kernel B, whose 140 VGPRs allow 12 waves
// per CU), chosen to maximise the fill of the last pass; fewer blocks per CU and larger blocks win ties.
// Waves per CU a shape may ask for: kernel B's stage sets without the cos perturbation and without hydrogen-type / general constraint
// clusters are built with 128 VGPRs (four waves per SIMD, 16 per CU), the others with 144-162 (three per SIMD, 12 per CU); kernel A fits
// either.  Round 4 (tools/probes/shape_sweep.py, profiles/r04v_shape_sweep.txt): with 12 everywhere, 2 628 / 2 920 / 3 504 tile waves
// (166-222 k particles) ran in two passes, 66.8 / 65.7 / 64.0 k steps/s; two blocks of 6-7 tile waves per CU hold them in one, 73.3 / 70.9 /
// 67.0 k.
void pick_launch_shape(vvhip_plan* p) {
    // cus = what the bound device reports (256 on an MI355X in SPX mode; 32 per XCD partition in CPX mode); before vvhip_bind the
    // plan assumes a whole MI355X.
    const int nw = p->hp.info.num_waves, cus = p->num_cus;
    // The cos perturbation's one-launch step collects ten rows in its rendezvous, shared by the waves of a block: three tile waves per block
    // (a third of the words to poll, four waves to share the rows) beat one or two up to 3 x CUs tile waves -- one rank's eighth / quarter of C4
    // 89.1 -> 92.8 k / 89.5 -> 91.1 k steps/s; with three rows the plan's choice below stays the best (profiles/r05j_small_shape.txt)
    // (round 6: that rule is gone with the shared-out polling it served -- one tile wave per block again, C4 / 8 9.70 against 9.92 us per step,
    // profiles/r06w_c4_shard_shapes.txt)
    if (nw <= cus) { p->block_threads = 64; p->grid_cap_a = p->grid_cap_b = cus; return; }
    // bandwidth-bound regime (the chain runs as its own launch there, kernel B fits 6 waves per SIMD): tuned at 8.9 M particles
    // (kernel B: two blocks per CU, not four -- round 4, three alternating runs: 2.66 M particles 7 330 -> 7 540 steps/s, 4.4 M 4 226 -> 4 326,
    // 8.9 M 1 970 -> 2 042; kernel A's eight blocks per CU against four: 7 540 / 7 547, 4 326 / 4 272, 2 042 / 2 074)
    // Round 5 (tools/probes/large_n_shape.py, profiles/r05j_large_n_shape.txt, three rotations each on two boxes): kernel A holds 74 VGPRs = six
    // waves per SIMD, so eight blocks of four waves per CU run as one round and a third; four per CU from 5 M particles: 5.5 M 3 350 -> 3 456
    // steps/s, 8.9 M 2 116 -> 2 248 (the other box 2 032 -> 2 060), 3.3 M 6 273 -> 6 253 (eight stay there); three or two per CU lose again.
    if (nw >= p->split_chain_waves) { p->block_threads = 256; p->grid_cap_a = (nw >= 80000 ? 4 : 8) * cus; p->grid_cap_b = 2 * cus; return; }
    const int max_waves = (p->hp.params.cos_acceleration != 0 || p->hp.info.num_shake_clusters > 0 || p->hp.info.num_general_constraints > 0 ||
                           p->hp.info.num_virtual_sites > 0) ? 12 : 16;
    double best = -1;
    int bk = 1, bt = 1;
    for (int k = 1; k <= 4; k++)
        for (int t = 1; t <= 7; t++) {
            if (k * (t + 1) > max_waves) continue;
            const long cap = (long) cus * k * t;
            const long passes = (nw + cap - 1) / cap;
            // fill of the last pass; once several passes are needed, shapes with fewer than 8 tile waves per CU in flight are
            // marked down (they leave memory-level parallelism unused)
            const double fill = (double) nw / (double) (cap * passes) * (passes > 1 ? std::min(1.0, k * t / 8.0) : 1.0);
            if (fill > best + 1e-9 || (fill > best - 1e-9 && (k < bk || (k == bk && t > bt)))) { best = fill; bk = k; bt = t; }
        }
    // Past what two blocks of seven tile waves per CU hold in one pass: that very shape, strided.  The fill rule above prefers shapes whose
    // last pass is fuller, and measured they lose: 5 256 / 7 008 / 10 512 tile waves 51.0 / 39.0 / 27.0 k steps/s against 46.8 / 36.5-37.6 /
    // 26.2-26.6 k for the runners-up; with the 12-wave stage sets as well (7 008 tile waves with HBonds 28.3 k against the rule's 24.7 k, with
    // the cos perturbation 31.6 against 29.3 k; 5 256: 36.7 / 35.8 k and 41.0 / 40.4 k) (profiles/r04zd_mid_sizes.txt).
    if (nw > (long) cus * 14) { bk = 2; bt = 7; }
    // The 12-wave stage sets between 2 048 and 3 072 tile waves: the fill rule ties one block of seven with two of four and takes the former;
    // measured the latter wins (HBonds 2 628 / 2 920 tile waves 54.9 / 53.8 k against 51.3 / 50.4 k steps/s, cos 60.2 / 58.6 against 59.3 / 58.2 k)
    else if (max_waves == 12 && nw > (long) cus * 8 && nw <= (long) cus * 12) { bk = 2; bt = 4; }
    p->block_threads = 64 * bt;
    p->grid_cap_a = p->grid_cap_b = cus * bk;
}

vv::NHConst make_chain(vvhip_plan* p, uint32_t flags);
constexpr int kAccN = vv::NUM_ACC * vv::ACC_SLOTS;
constexpr int kRvCopy = 6 * kAccN;      // rendezvous words of one thermostat parity: up to 6 replicas (vv_device.inc: RV_REPLICAS) of [NUM_ACC][ACC_SLOTS]
// distance between the two parity copies: only the rows in use (4 without the cos moments)
int acc_stride(const vvhip_plan* p) { return (p->hp.params.cos_acceleration != 0 ? vv::NUM_ACC : 4) * vv::ACC_SLOTS; }

vv::KArgs make_args(vvhip_plan* p, uint32_t flags, uint32_t random_index) {
    const vvhip_params& q = p->hp.params;
    vv::KArgs a{};
    a.velm = p->buf.velm;
    a.posq = p->buf.posq;
    a.corr = p->hp.precision == VVHIP_MIXED ? p->buf.posq_correction : nullptr;
    a.force = (const long long*) p->buf.force;
    a.fextra = p->d_fextra;
    a.pos_delta = p->buf.pos_delta ? p->buf.pos_delta : p->d_pos_delta;
    a.old_delta = p->d_old_delta;
    a.comv = p->d_comv;
    a.comw = p->d_comw;
    a.seg_mass = p->d_seg_mass;
    a.seg_base = p->d_seg_base;
    a.cosz = p->d_cosz;
    a.slots = p->d_slots;
    a.slot_m = p->d_slot_m;
    a.slot_f = p->d_slot_f;
    a.slot_image = p->d_slot_image;
    a.slot_rand = p->d_slot_rand;
    a.slot_shake = p->d_slot_shake;
    a.slot_shake_param = p->d_slot_shake_param;
    a.slot_vsite = p->d_slot_vsite;
    a.vsite_params = p->d_vsite_params;
    a.vsite_atom = p->d_vsite_atom;
    a.shake_tol = q.constraint_tolerance > 0 ? q.constraint_tolerance : 1e-5;
    a.slot_big = p->d_slot_big;
    a.bigacc = p->d_bigacc;
    a.big_scale = p->hp.big_scale;
    a.big_inv_scale = 1.0 / p->hp.big_scale;
    a.random = (const float4*) p->buf.random;
    a.acc = p->d_acc + p->parity * acc_stride(p);
    a.acc_next = p->d_acc + (p->parity ^ 1) * acc_stride(p);
    a.nh = p->d_nh + p->parity;
    a.nh_next = p->d_nh + (p->parity ^ 1);
    a.chain = make_chain(p, 0);
    a.lane_const = p->d_lane_const;
    a.mb.local = p->mb_local;
    a.mb.peers = p->d_mb_peers;
    a.mb.ctl = p->d_mb_ctl;
    a.mb.ranks = p->mb_ranks;
    a.mb.rank = p->mb_rank;
    a.status = p->d_status;
    a.dbg = p->d_dbg;
    a.dbg_block = p->dbg_block;
    a.dbg_span = p->d_dbg_span;
    a.dbg_parity = p->dbg_seq >= 0 ? p->dbg_seq++ % 6 : p->dbg_parity;
    a.padded = p->hp.padded_num_atoms;
    a.gc_colors = p->hp.gc_colors;
    a.gc_omega = p->hp.gc_omega;
    {   // posq / posqCorrection as a buffer resource (kernel A's member-only position fetch): 32-bit sizes and offsets
        const unsigned long long bytes = (unsigned long long) (p->hp.shard_end - p->hp.shard_begin) * (p->hp.precision == VVHIP_DOUBLE ? 32ull : 16ull);
        a.pos_bytes = bytes < 0xFFFFFFE0ull ? (uint32_t) bytes : 0u;
    }
    a.nwaves = p->hp.info.num_waves;
    a.acc_rows = p->hp.params.cos_acceleration != 0 ? vv::NUM_ACC : 4;
    a.acc_exclusive = p->acc_store ? 1 : 0;
    a.flags = flags;
    a.random_index = random_index;
    a.per = vv::periodic_args(p->hp.per);
    a.dt = q.step_size;
    // the same IEEE quotients the kernels used to form per lane: (mixed) 1 / (mixed) dt and 1.0 / (mixed) dt
    a.inv_dt_mixed = p->hp.precision == VVHIP_SINGLE ? (double) (1.0f / (float) q.step_size) : 1.0 / q.step_size;
    a.inv_dt_double = p->hp.precision == VVHIP_SINGLE ? 1.0 / (double) (float) q.step_size : 1.0 / q.step_size;
    a.fscale_vv = 0.5 * q.step_size / (double) 0x100000000;                                    // HOST:306
    a.drag = q.friction;                                                                        // HOST:835-839
    a.randf = std::sqrt(2.0 * kBoltz * q.temperature * q.friction / q.step_size);
    a.drag_drude = q.drude_friction;
    a.randf_drude = std::sqrt(2.0 * kBoltz * q.drude_temperature * q.drude_friction / q.step_size);
    a.efscale = q.electric_field * kAvogadro;                                                   // HOST:978
    a.cos_accel = q.cos_acceleration;
    a.inv_box_z = 1.0 / p->box[2];
    a.max_drude = q.max_drude_distance;
    a.hw_scale = std::sqrt(kBoltz * q.drude_temperature);                                       // HOST:190
    a.mirror = q.mirror_location;
    a.inv_mass_total = p->hp.info.inv_mass_total;
    for (int i = 0; i < vv::NUM_ACC; i++) { a.acc_scale[i] = p->acc_scale[i]; a.acc_inv_scale[i] = p->acc_inv_scale[i]; }
    return a;
}

vv::NHConst make_chain(vvhip_plan* p, uint32_t flags) {
    const vvhip_params& q = p->hp.params;
    const vvhip_plan_info& in = p->hp.info;
    vv::NHConst c{};
    std::memcpy(c.eta_mass, in.eta_mass, sizeof(c.eta_mass));
    for (int g = 0; g < 3; g++)
        for (int i = 0; i < VVHIP_MAX_CHAINS; i++) c.inv_eta_mass[g][i] = in.eta_mass[g][i] > 0 ? 1.0 / in.eta_mass[g][i] : 0.0;
    for (int g = 0; g < 3; g++) {
        c.nkbt[g] = in.nkbt[g];
        c.temperature[g] = g == 2 ? q.drude_temperature : q.temperature;                        // HOST:728
    }
    c.step_size = q.step_size;
    c.inv_mass_total = in.inv_mass_total;
    for (int i = 0; i < vv::NUM_ACC; i++) c.acc_inv_scale[i] = p->acc_inv_scale[i];
    c.num_chains = q.num_nh_chains;
    c.loops_per_step = q.loops_per_step;
    c.num_tg = in.num_temp_groups;
    c.flags = flags;
    return c;
}

// Chain constants per temperature group for kernel B's thermostat wave; the temperatures are read live (HOST:728), so this is
// refreshed whenever the parameters change.
int upload_lane_const(vvhip_plan* p) {
    if (!p->d_lane_const) return VVHIP_OK;
    const vvhip_params& q = p->hp.params;
    const vvhip_plan_info& in = p->hp.info;
    vv::ChainLaneBlock b[VVHIP_NUM_TG] = {};
    for (int g = 0; g < VVHIP_NUM_TG; g++) {
        for (int i = 0; i < 4; i++) {
            b[g].eta_mass[i] = in.eta_mass[g][i];
            b[g].inv_eta_mass[i] = in.eta_mass[g][i] > 0 ? 1.0 / in.eta_mass[g][i] : 0.0;
        }
        b[g].nkbt = in.nkbt[g];
        b[g].kT = kBoltz * (g == 2 ? q.drude_temperature : q.temperature);
        b[g].acc_inv_scale = p->acc_inv_scale[g];
        b[g].active = (g < in.num_temp_groups && in.eta_mass[g][0] > 0) ? 1.0 : 0.0;
        b[g].dt2 = q.step_size / q.loops_per_step / 2;                            // API:343-345
        b[g].dt4 = b[g].dt2 / 2;
        b[g].dt8 = b[g].dt4 / 2;
    }
    // hosts re-send their parameters every step (the reference re-reads the getters every step): only a real change costs a copy
    if (p->lane_const_valid && std::memcmp(p->lane_const_host, b, sizeof(b)) == 0) return VVHIP_OK;
    (void) hipStreamSynchronize(p->stream);
    hipError_t e = hipMemcpy(p->d_lane_const, b, sizeof(b), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(p, e, "hipMemcpy(chain constants)");
    std::memcpy(p->lane_const_host, b, sizeof(b));
    p->lane_const_valid = true;
    return VVHIP_OK;
}

struct ScopedTimer {
    vvhip_plan* p;
    int cls;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool on, ranged = false;
    // dispatch = true: the launcher delivers the dispatch's own begin / end timestamps into e0 / e1 (kernels A and B: vv_launch);
    // otherwise the events are recorded around the enqueued work (adds two barrier packets to the stream)
    bool dispatch;
    static hipEvent_t take(vvhip_plan* p) {
        hipEvent_t e = nullptr;
        if (!p->event_pool.empty()) { e = p->event_pool.back(); p->event_pool.pop_back(); }
        else (void) hipEventCreate(&e);
        return e;
    }
    ScopedTimer(vvhip_plan* p_, int cls_, bool dispatch_ = false)
        : p(p_), cls(cls_), on(p_->timing && !p_->capturing && (cls_ != T_OTHER || !p_->timing_kernels_only)), dispatch(dispatch_) {
        if (p->trace && !p->capturing) {
            static const char* names[3] = {"vvhip kernel A (kick / extra forces / sums)", "vvhip kernel B (thermostat / drift / hard wall)", "vvhip other"};
            if (roctx_api().push) { roctx_api().push(names[cls]); ranged = true; }
        }
        if (on) {
            e0 = take(p);
            e1 = take(p);
            if (!dispatch) (void) hipEventRecord(e0, p->stream);
        }
    }
    ~ScopedTimer() {
        if (ranged) roctx_api().pop();
        if (on) {
            if (!dispatch) (void) hipEventRecord(e1, p->stream);
            p->events[cls].emplace_back(e0, e1);
        }
    }
};

// The static mass tables are filled lazily, right in front of the first stage launch that reads them (by then velm.w is what the
// host integrates with); inside a graph capture that would record the fill into every replay, so the capture entry points call this first.
static inline void debug_stall(vvhip_plan* p) {
    if (p->stall_us > 0 && !p->capturing && ++p->stall_count % p->stall_period == 0) usleep((useconds_t) p->stall_us);
}
// Ranks that SHARE a device (test set-ups; found out by vvhip_mailbox_connect) exchange through kernel B's polling thermostat waves: the
// ranks' kernels must be resident together, or the one that got the device first polls until its bounded waits run out while the
// others' launches cannot start (measured round 4, two ranks on one MI355X, 0.44 M / 0.89 M particles: device-filling grids time out
// with either work-item layout, grids of <= half the CUs per rank never do -- tools/probes/mailbox_periodic.sh).  Every rank then
// takes its share of the CUs, one block per CU.  Ranks on devices of their own keep the plan's launch shape.
// A launch that found neither a compiled nor a run-time kernel for its stage set and ran the generic one (15-20 % slower): counted per
// plan (vvhip_generic_launches); VVHIP_WARN_GENERIC=1 also prints one line per plan, kernel and stage set.
void note_generic_launch(vvhip_plan* p, int kernel, uint32_t flags) {
    p->generic_launches[kernel]++;
    // (two stage sets that alternate -- a classic step's halves -- would print on every launch if only the last one were remembered)
    bool seen = false;
    for (uint32_t f : p->generic_seen[kernel]) seen = seen || f == flags;
    if (!seen && p->generic_seen[kernel].size() < 64) p->generic_seen[kernel].push_back(flags);
    p->generic_flags[kernel] = flags;
    static const bool warn = std::getenv("VVHIP_WARN_GENERIC") != nullptr;
    if (warn && !seen) std::fprintf(stderr, "vvhip: kernel %c runs stage set 0x%x on the generic kernel (no compiled specialisation)\n", kernel == 0 ? 'A' : 'B', flags);
}
int shared_device_cap(const vvhip_plan* p, int cap) {
    if (!(p->mb_on && p->mb_shared_device) || p->launch_shape_forced) return cap;
    return std::max(1, std::min(cap, p->num_cus / std::max(1, p->mb_device_ranks)));
}
int ensure_mass_table(vvhip_plan* p) {
    if (!(p->mass_tab_a || p->mass_tab_b) || p->mass_tab_valid) return VVHIP_OK;
    if (p->capturing) return fail(p, VVHIP_ERR_INVALID, "internal: mass tables must be filled before a graph capture starts");
    HIP_TRY(p, vv::launch_mass_table(p->hp.precision, p->buf.velm, p->d_slots, p->hp.info.num_waves, p->d_slot_m, p->d_slot_f, p->stream));
    p->mass_tab_valid = true;
    return VVHIP_OK;
}
int run_a(vvhip_plan* p, uint32_t flags, uint32_t random_index) {
    TRY(settle_recovery(p));
    if (p->mass_tab_a) { flags |= vv::A_MTAB; TRY(ensure_mass_table(p)); }
    // (kernel A takes the arithmetic path where it also saves the 20 bytes per lane of constraint tables; else it does not gain, see periodic_a)
    if (p->hp.per.enabled && p->periodic_kernels && (p->periodic_a || (flags & vv::A_CONS))) flags |= vv::A_PERIODIC;
    if ((flags & vv::A_SHAKE_V) && p->shake_mode == 0) flags |= vv::A_SHAKE_GS;
    debug_stall(p);
    ScopedTimer t(p, T_A, true);
    int route = vv::ROUTE_COMPILED;
    HIP_TRY(p, vv::launch_a(p->hp.precision, make_args(p, flags, random_index), p->block_threads, shared_device_cap(p, p->grid_cap_a), p->stream, t.e0, t.e1, &route));
    if (route == vv::ROUTE_GENERIC) note_generic_launch(p, 0, flags);
    return VVHIP_OK;
}
// Kernel B takes the arithmetic layout whenever the plan has one, also next to the mailbox exchange (round 3 kept them apart after time-outs
// with two ranks on one GPU; round 4 found the cause in device-filling grids of polling waves, whatever the layout: shared_device_cap)
bool periodic_b(const vvhip_plan* p) { return p->hp.per.enabled && p->periodic_kernels && p->periodic_b; }
int run_b(vvhip_plan* p, uint32_t flags) {
    TRY(settle_recovery(p));
    if (p->mass_tab_b) { flags |= vv::B_MTAB; TRY(ensure_mass_table(p)); }
    if (periodic_b(p)) flags |= vv::B_PERIODIC;
    if ((flags & vv::B_SHAKE) && p->shake_mode == 0) flags |= vv::B_SHAKE_GS;
    debug_stall(p);
    ScopedTimer t(p, T_B, true);
    int route = vv::ROUTE_COMPILED;
    HIP_TRY(p, vv::launch_b(p->hp.precision, make_args(p, flags, 0), p->block_threads, shared_device_cap(p, p->grid_cap_b), p->stream, t.e0, t.e1, &route));
    if (route == vv::ROUTE_GENERIC) note_generic_launch(p, 1, flags);
    if (flags & vv::B_CHAIN) p->parity ^= 1;     // the advanced thermostat state now lives in the other copy
    return VVHIP_OK;
}
int run_chain(vvhip_plan* p, uint32_t flags) {
    TRY(settle_recovery(p));
    debug_stall(p);
    ScopedTimer t(p, T_OTHER);
    HIP_TRY(p, vv::launch_chain(make_chain(p, flags), p->d_nh + p->parity, p->d_acc + p->parity * acc_stride(p), p->stream));
    return VVHIP_OK;
}

// cos perturbation in two launches instead of three: kernel A accumulates the group sums as moments of the biased velocities next
// to the bias moment itself, kernel B's inline chain finishes the algebra (vv_args.hpp: A_KE_MOM).  Not with molecules larger
// than a wave or the stand-alone chain launch (long chains, very large systems), which keep the bias -> KE -> scale sequence.
bool use_moments(const vvhip_plan* p) {
    return p->hp.params.cos_acceleration != 0 && p->hp.has_nh && p->hp.num_big == 0 && p->hp.params.num_nh_chains <= 4 &&
           p->hp.info.num_waves < p->split_chain_waves && !p->no_moments;
}
// The mailbox carries the totals between the ranks' kernel-B heads (inline chain): the three kinetic-energy sums, and with the cos
// perturbation in its moment form also the bias moment and the six group moments -- everything kernel A produced, one exchange per
// thermostat application.  The three-launch cos sequence (its bias moment is consumed by another kernel A) and the stand-alone
// chain kernel still go through the collective.
bool use_mailbox(const vvhip_plan* p) {
    return p->mb_on && p->hp.params.num_nh_chains <= 4 && (p->hp.params.cos_acceleration == 0 || use_moments(p));
}

// The launch(es) that end in the per-group kinetic energies.  `first` = stage bits that must run before the KE on the
// same launch if possible (kick, extra forces).  Molecules larger than a wave need their COM summed across waves
// first (A_COMPART, its own launch after a memset of the small accumulator), so there the stages are split.
int run_ke(vvhip_plan* p, uint32_t first, uint32_t random_index, bool unbias) {
    const uint32_t ub = unbias ? (vv::A_UNBIAS_ACC | vv::A_CZ_LOAD) : 0;    // the bias launch of this step cached cos(kz)
    if (p->hp.num_big == 0) {
        if (unbias && first) { int rc = run_a(p, first, random_index); if (rc != VVHIP_OK) return rc; first = 0; }
        return run_a(p, first | vv::A_KE | ub, random_index);
    }
    if (first) { int rc = run_a(p, first, random_index); if (rc != VVHIP_OK) return rc; }
    HIP_TRY(p, hipMemsetAsync(p->d_bigacc, 0, (size_t) p->hp.num_big * 4 * sizeof(unsigned long long), p->stream));
    int rc = run_a(p, vv::A_COMPART | ub, 0);
    return rc != VVHIP_OK ? rc : run_a(p, vv::A_KE | ub, 0);
}

// Scaling kernel with the chain in its head (chain length <= 4), or the stand-alone chain launch in front of it.
int run_chain_and_b(vvhip_plan* p, uint32_t bflags, bool with_bias) {
    // Large systems: the chain registers cost kernel B half its occupancy (140 vs 74 VGPRs), which matters once the kernel is
    // bandwidth bound; there the chain runs as its own one-wave launch and B only reads the scale factors.
    const bool split = p->hp.info.num_waves >= p->split_chain_waves && !use_mailbox(p);
    if (p->hp.params.num_nh_chains <= 4 && !split) return run_b(p, vv::B_CHAIN | bflags | (use_mailbox(p) ? vv::B_MAILBOX : 0));
    int rc = run_chain(p, vv::C_CHAIN | (with_bias ? vv::C_BIAS : 0));
    return rc != VVHIP_OK ? rc : run_b(p, bflags);
}

// ---- the one-launch step (vv_device.inc: "fused step")
// Shape: the plan's own (pick_launch_shape) when it gives every tile a wave of its own on at most ACC_SLOTS blocks, one block per CU.
bool fused_shape_ok(const vvhip_plan* p) {
    const int tiles = p->block_threads / 64, nw = p->hp.info.num_waves;
    if (tiles < 1 || tiles > 7) return false;
    const int blocks = (nw + tiles - 1) / tiles;
    return blocks >= 1 && blocks <= vv::ACC_SLOTS && blocks <= std::min(p->grid_cap_b, p->grid_cap_a) && blocks <= p->num_cus;
}
// What the plan's state allows, before any kernel is looked up.  The two halves must not need anything between them: no RCCL exchange
// (sharded runs with a communicator), no stand-alone chain launch (long chains, very large systems), no partial sums of molecules larger
// than a wave, no three-launch cos sequence; ranks that share this device (test set-ups) keep the two-launch step, whose kernels need
// not be resident together.
bool fused_state_ok(const vvhip_plan* p) {
    const vv::HostPlan& hp = p->hp;
    if (!p->fused || !hp.has_nh || hp.params.num_nh_chains > 4 || hp.num_big != 0) return false;
    if (hp.info.num_waves >= p->split_chain_waves) return false;
    // sharded runs: the xGMI mailbox exchanges the ranks' totals inside the thermostat wave, right behind the local rendezvous (one wait after
    // the other, no launch in between); an RCCL all-reduce needs the kernel boundary, and ranks that share this device cannot all be resident
    if ((p->comm && !use_mailbox(p)) || (p->mb_on && (!use_mailbox(p) || p->mb_shared_device))) return false;
    if (hp.params.cos_acceleration != 0 && (p->no_moments || hp.params.num_nh_chains > 4)) return false;
    if (p->mass_tab_a || !p->mass_tab_b || p->shake_mode == 0) return false;      // (comparison builds of the two-launch kernels)
    if (hp.per.enabled && p->periodic_kernels) return false;                      // the arithmetic layout belongs to the many-pass regime
    return fused_shape_ok(p);
}
int run_fused(vvhip_plan* p, uint32_t aflags, uint32_t bflags, uint32_t random_index, bool* taken) {
    *taken = false;
    if (!fused_state_ok(p)) return VVHIP_OK;
    bflags |= vv::B_CHAIN | vv::B_MTAB | (use_mailbox(p) ? vv::B_MAILBOX : 0u);
    // kernel and occupancy of this pair of stage sets on this launch shape: looked up once
    if (p->fused_checked_a != aflags || p->fused_checked_b != bflags || p->fused_checked_threads != p->block_threads || p->fused_checked_waves != p->hp.info.num_waves) {
        for (const vvhip_plan::FusedCheck& c : p->fused_checks)
            if (c.b != 0 && c.a == aflags && c.b == bflags && c.threads == p->block_threads && c.waves == p->hp.info.num_waves) {
                p->fused_checked_a = aflags; p->fused_checked_b = bflags; p->fused_checked_threads = c.threads; p->fused_checked_waves = c.waves; p->fused_ok = c.ok;
            }
    }
    if (p->fused_checked_a != aflags || p->fused_checked_b != bflags || p->fused_checked_threads != p->block_threads || p->fused_checked_waves != p->hp.info.num_waves) {
        p->fused_checked_a = aflags; p->fused_checked_b = bflags; p->fused_checked_threads = p->block_threads; p->fused_checked_waves = p->hp.info.num_waves;
        vv::KArgs q = make_args(p, bflags, random_index);
        q.flags_a = aflags;
        int per_cu = 0;
        const hipError_t e = vv::launch_fused(p->hp.precision, q, p->block_threads, p->d_rv, p->stream, nullptr, nullptr, nullptr, &per_cu);      // (asks only; launches nothing)
        const int tiles = p->block_threads / 64, blocks = (p->hp.info.num_waves + tiles - 1) / tiles;
        p->fused_ok = e == hipSuccess && per_cu >= 1 && (long) per_cu * p->num_cus >= blocks;
        if (e != hipSuccess) (void) hipGetLastError();
        vvhip_plan::FusedCheck& c = p->fused_checks[p->fused_check_next++ & 3];
        c.a = aflags; c.b = bflags; c.threads = p->block_threads; c.waves = p->hp.info.num_waves; c.ok = p->fused_ok;
    }
    if (!p->fused_ok) return VVHIP_OK;
    TRY(settle_recovery(p));
    if (!p->fused) return VVHIP_OK;      // (settling may have pinned the plan to two launches)
    TRY(ensure_mass_table(p));
    debug_stall(p);
    ScopedTimer t(p, T_B, true);
    int route = vv::ROUTE_COMPILED;
    vv::KArgs q = make_args(p, bflags, random_index);
    q.flags_a = aflags;
    q.fused_poll_delay = p->fused_poll_delay;
    // the "a block polled twice" words of this step and of the one before (by thermostat parity), behind the two copies of the rendezvous words
    q.rv_late_cur = (unsigned int*) (p->d_rv + 2 * kRvCopy) + vv::ACC_SLOTS * p->parity;
    q.rv_late_prev = (const unsigned int*) (p->d_rv + 2 * kRvCopy) + vv::ACC_SLOTS * (p->parity ^ 1);
    q.fused_late_shift = p->fused_late_shift;
    HIP_TRY(p, vv::launch_fused(p->hp.precision, q, p->block_threads, p->d_rv + p->parity * kRvCopy, p->stream, t.e0, t.e1, &route, nullptr));
    p->parity ^= 1;            // the advanced thermostat state now lives in the other copy
    p->fused_launches++;
    *taken = true;
    return VVHIP_OK;
}

uint32_t extra_flags(const vvhip_plan* p) {
    uint32_t f = 0;
    if (p->hp.has_ld) f |= vv::A_LD;
    if (p->hp.has_ef) f |= vv::A_EF;
    if (p->hp.params.cos_acceleration != 0) f |= vv::A_COS;
    return f;
}
uint32_t tail_flags(const vvhip_plan* p) {      // what follows every position update (HOST:203-212, API:266-268)
    uint32_t f = 0;
    if (p->hp.params.max_drude_distance > 0 && p->hp.has_pairs) f |= vv::B_HARDWALL;
    if (p->hp.has_images) f |= vv::B_IMAGE;
    if (!p->hp.slot_vsite.empty()) f |= vv::B_VSITE;
    return f;
}
bool cos_on(const vvhip_plan* p) { return p->hp.params.cos_acceleration != 0; }
bool shake_on(const vvhip_plan* p) { return !p->hp.slot_shake.empty(); }
// stage bits of the in-kernel constraints the plan holds: hydrogen-type clusters and / or rigid three-site molecules
uint32_t cons_a(const vvhip_plan* p) { return (p->hp.info.num_shake_clusters > 0 ? vv::A_SHAKE_V : 0u) | (p->hp.info.num_settle_clusters > 0 ? vv::A_SETTLE : 0u) | (p->hp.info.num_general_constraints > 0 ? vv::A_GCONS : 0u); }
uint32_t cons_b(const vvhip_plan* p) { return (p->hp.info.num_shake_clusters > 0 ? vv::B_SHAKE : 0u) | (p->hp.info.num_settle_clusters > 0 ? vv::B_SETTLE : 0u) | (p->hp.info.num_general_constraints > 0 ? vv::B_GCONS : 0u); }
#define NEED_FUSABLE(p)                                                                                                   \
    do {                                                                                                                \
        if (!(p)->hp.info.constraints_fused)                                                                            \
            return fail(p, VVHIP_ERR_UNSUPPORTED, std::string("the System has constraints this backend cannot solve in-kernel") + ((p)->hp.unfused_reason.empty() ? "" : " (" + (p)->hp.unfused_reason + ")") + \
                                                  ": use the split entry points around the host's constraint solver"); \
    } while (0)

}  // namespace

extern "C" {

static void mailbox_release(vvhip_plan* p);

// ------------------------------------------------------------------------------------------ life cycle
int vvhip_plan_create(const vvhip_system_desc* system, const vvhip_params* params, int precision, vvhip_plan** plan_out,
                      char* errbuf, size_t errbuf_len) {
    auto report = [&](int code, const std::string& msg) {
        if (errbuf && errbuf_len) std::snprintf(errbuf, errbuf_len, "%s", msg.c_str());
        return code;
    };
    if (!system || !params || !plan_out) return report(VVHIP_ERR_INVALID, "null argument");
    try {
        vvhip_plan* p = new vvhip_plan();
        p->hp = vv::analyze(*system, *params, precision);
        fill_scales(p);
        if (const char* e = std::getenv("VVHIP_STALL")) {
            p->stall_us = std::atol(e);
            if (const char* c = std::strchr(e, ':')) p->stall_period = std::max(1L, std::atol(c + 1));
        }
        if (const char* e = std::getenv("VVHIP_SHAKE_MODE")) p->shake_mode = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("VVHIP_FUSED")) p->fused = std::atoi(e) != 0;
        if (const char* e = std::getenv("VVHIP_RECOVER")) p->rec.enabled = std::atoi(e) != 0;
        p->mass_tab_a = vv::sf_kernels_use_mass_table(0);
        p->mass_tab_b = vv::sf_kernels_use_mass_table(1);
        if (const char* e = std::getenv("VVHIP_ROCTX")) p->trace = std::atoi(e) != 0;
        pick_launch_shape(p);
        *plan_out = p;
        return VVHIP_OK;
    } catch (const vv::Error& e) {
        return report(e.code, e.what());
    } catch (const std::exception& e) {
        return report(VVHIP_ERR_INVALID, e.what());
    }
}

// Test / tuning hook (include/vvhip.h): the choices the measurements of TUNING_LOG.md settled, adjustable per plan so that tests can force
// code paths at sizes they can afford (the large-system launch shape on 10 000 particles, loaded instead of computed slot words, ...).
// The environment switches of rounds 1-3 (VVHIP_REKICK, VVHIP_CAP_A, ...) are gone: their experiments are closed.
int vvhip_debug_tune(vvhip_plan* p, const char* key, int value) {
    if (!p || !key) return VVHIP_ERR_INVALID;
    if (p->bound) TRY(settle_recovery(p));
    if (p->bound) HIP_TRY(p, hipStreamSynchronize(p->stream));
    const std::string k = key;
    if (k == "periodic_kernels") p->periodic_kernels = value != 0;          // 0: keep the arithmetic layout's slot order but load the slot words
    else if (k == "periodic_a") p->periodic_a = value != 0;                 // kernel A alone
    else if (k == "periodic_b") p->periodic_b = value != 0;                 // kernel B alone
    else if (k == "gc_omega_permille") p->hp.gc_omega = value / 1000.0;     // relaxation factor of the general clusters' sweeps (rate scans)
    else if (k == "rekick") p->rekick = value != 0;                         // 0: kernel A stores the kicked velocities, kernel B does not repeat the kick
    else if (k == "no_moments") p->no_moments = value != 0;                 // 1: cos perturbation as three launches (bias, sums, scale)
    else if (k == "fused") { p->fused = value != 0; p->fused_checked_b = 0; for (vvhip_plan::FusedCheck& c : p->fused_checks) c.b = 0; }
    else if (k == "recover") { p->rec.enabled = value != 0; p->rec.valid = false; p->rec.runs.clear(); }      // 0: a missed rendezvous stays fatal (VVHIP_ERR_RENDEZVOUS)
    else if (k == "recover_min_steps") p->rec.min_steps = std::max(1, value);
    else if (k == "fused_late_shift") p->fused_late_shift = std::max(0, std::min(value, 16));
    else if (k == "fused_poll_delay") p->fused_poll_delay = std::max(-1, std::min(value, 64));   // 0: the middle scheme's step as two launches (A, B) also where one would do
    else if (k == "mass_tab_a") p->mass_tab_a = value != 0;
    else if (k == "mass_tab_b") p->mass_tab_b = value != 0;
    else if (k == "acc_store") p->acc_store = value != 0;                   // 0: atomics also where a block owns its accumulator slot
    else if (k == "split_chain_waves") { p->split_chain_waves = value; if (!p->launch_shape_forced) pick_launch_shape(p); }
    else if (k == "block_threads") {
        if (value < 64 || value > 448 || value % 64) return fail(p, VVHIP_ERR_INVALID, "block_threads: a multiple of 64 in [64, 448]");
        p->block_threads = value; p->grid_cap_a = 2048; p->grid_cap_b = 1024; p->launch_shape_forced = true;
    }
    else if (k == "grid_cap_a") { if (value < 1) return VVHIP_ERR_INVALID; p->grid_cap_a = value; p->launch_shape_forced = true; }
    else if (k == "grid_cap_b") { if (value < 1) return VVHIP_ERR_INVALID; p->grid_cap_b = value; p->launch_shape_forced = true; }
    else return fail(p, VVHIP_ERR_INVALID, "vvhip_debug_tune: unknown key " + k);
    if (k == "mass_tab_a" || k == "mass_tab_b") p->mass_tab_valid = false;
    drop_graphs(p);
    return VVHIP_OK;
}

void vvhip_plan_destroy(vvhip_plan* p) {
    if (!p) return;
    if (p->bound) {
        (void) hipStreamSynchronize(p->stream);
        for (void* ptr : {(void*) p->d_slots, (void*) p->d_slot_image, (void*) p->d_slot_rand, (void*) p->d_slot_shake, (void*) p->d_slot_shake_param, (void*) p->d_slot_vsite, (void*) p->d_vsite_params, (void*) p->d_vsite_atom, (void*) p->d_slot_big, (void*) p->d_bigacc, (void*) p->d_image_pairs,
                          p->d_fextra, p->d_old_delta, p->d_pos_delta, p->d_comv, (void*) p->d_comw, (void*) p->d_seg_mass, (void*) p->d_seg_base, (void*) p->d_slot_m, (void*) p->d_slot_f, (void*) p->d_cosz, (void*) p->d_epoch, (void*) p->d_acc, (void*) p->d_rv, (void*) p->d_nh, (void*) p->d_lane_const, (void*) p->d_dbg, (void*) p->d_dbg_span,
                          p->rec.posq, p->rec.corr, p->rec.velm, p->rec.force, p->rec.fextra, p->rec.random, (void*) p->rec.nh, (void*) p->rec.epoch})
            if (ptr) (void) hipFree(ptr);
        drop_graphs(p);
        if (p->comm) (void) rccl_api().commDestroy(p->comm);
        mailbox_release(p);
        if (p->h_status) (void) hipHostFree(p->h_status);
        for (auto& v : p->events)
            for (auto& e : v) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
        for (hipEvent_t e : p->event_pool) (void) hipEventDestroy(e);
    }
    delete p;
}

const char* vvhip_last_error(const vvhip_plan* p) { return p ? p->err.c_str() : "null plan"; }

int vvhip_plan_get_info(const vvhip_plan* p, vvhip_plan_info* info) {
    if (!p || !info) return VVHIP_ERR_INVALID;
    *info = p->hp.info;
    return VVHIP_OK;
}

const char* vvhip_plan_unfused_reason(const vvhip_plan* p) { return p ? p->hp.unfused_reason.c_str() : "null plan"; }

int vvhip_plan_get_slots(const vvhip_plan* p, int32_t* slots, int32_t capacity) {
    if (!p) return VVHIP_ERR_INVALID;
    const int32_t n = p->hp.info.num_waves * 64;
    if (slots) {
        if (capacity < n) return VVHIP_ERR_INVALID;
        std::memcpy(slots, p->hp.slots.data(), (size_t) n * 2 * sizeof(int32_t));
    }
    return n;
}

int vvhip_bind(vvhip_plan* p, const vvhip_buffers* b) {
    if (!p || !b) return VVHIP_ERR_INVALID;
    if (!b->velm || !b->posq || !b->force) return fail(p, VVHIP_ERR_INVALID, "velm, posq and force must be device pointers");
    if (p->hp.precision == VVHIP_MIXED && !b->posq_correction)
        return fail(p, VVHIP_ERR_INVALID, "mixed precision needs posq_correction (the reference's image kernel dereferences it too: quirk Q5)");
    if (p->hp.has_ld && (!b->random || b->random_size == 0))
        return fail(p, VVHIP_ERR_INVALID, "Langevin particles present but no random buffer bound");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(p, VVHIP_ERR_NO_DEVICE, "no HIP device: libvvhip has no CPU path");
    if (p->bound) TRY(settle_recovery(p));
    if (p->bound) {
        // The captured graphs bake EVERY caller-owned pointer (make_args): a re-bind that swaps any of them must drop both
        // executables, or vvhip_run_graph would replay kernels on the old arrays without a word.  The stream is not part of a
        // captured node.  Only another velm array invalidates the mass tables (its inverse masses are re-read).
        const vvhip_buffers& o = p->buf;
        const bool same = b->velm == o.velm && b->posq == o.posq && b->posq_correction == o.posq_correction && b->force == o.force &&
                          b->pos_delta == o.pos_delta && b->random == o.random && b->random_size == o.random_size;
        if (b->velm != o.velm) p->mass_tab_valid = false;
        if (!same) drop_graphs(p);
    }
    // a re-bind that moves the plan to another stream: whatever the plan still has in flight on the old one (fills, steps) must be
    // complete before work enqueued on the new one can touch the same buffers
    if (p->bound && p->stream != (hipStream_t) b->stream) HIP_TRY(p, hipStreamSynchronize(p->stream));
    p->buf = *b;
    p->stream = (hipStream_t) b->stream;
    if (p->bound) return VVHIP_OK;      // re-binding only swaps the caller-owned pointers
    const vv::HostPlan& hp = p->hp;
    const size_t nloc = (size_t) (hp.shard_end - hp.shard_begin);
    const size_t nslots = (size_t) hp.info.num_waves * 64;
    const size_t rs = sizeof_real(hp.precision), ms = sizeof_mixed(hp.precision);
    HIP_TRY(p, hipMalloc((void**) &p->d_slots, nslots * sizeof(int2)));
    HIP_TRY(p, hipMemcpy(p->d_slots, hp.slots.data(), nslots * sizeof(int2), hipMemcpyHostToDevice));
    if (!hp.slot_image.empty()) {
        HIP_TRY(p, hipMalloc((void**) &p->d_slot_image, nslots * sizeof(int32_t)));
        HIP_TRY(p, hipMemcpy(p->d_slot_image, hp.slot_image.data(), nslots * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!hp.image_pairs.empty()) {
            HIP_TRY(p, hipMalloc((void**) &p->d_image_pairs, hp.image_pairs.size() * sizeof(int32_t)));
            HIP_TRY(p, hipMemcpy(p->d_image_pairs, hp.image_pairs.data(), hp.image_pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    if (!hp.slot_rand.empty()) {
        HIP_TRY(p, hipMalloc((void**) &p->d_slot_rand, nslots * sizeof(int32_t)));
        HIP_TRY(p, hipMemcpy(p->d_slot_rand, hp.slot_rand.data(), nslots * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (!hp.slot_shake.empty()) {
        HIP_TRY(p, hipMalloc((void**) &p->d_slot_shake, nslots * sizeof(int32_t)));
        HIP_TRY(p, hipMemcpy(p->d_slot_shake, hp.slot_shake.data(), nslots * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(p, hipMalloc((void**) &p->d_slot_shake_param, nslots * sizeof(float4)));
        HIP_TRY(p, hipMemcpy(p->d_slot_shake_param, hp.slot_shake_param.data(), nslots * sizeof(float4), hipMemcpyHostToDevice));
    }
    if (!hp.slot_vsite.empty()) {
        HIP_TRY(p, hipMalloc((void**) &p->d_slot_vsite, nslots * sizeof(int2)));
        HIP_TRY(p, hipMemcpy(p->d_slot_vsite, hp.slot_vsite.data(), nslots * sizeof(int2), hipMemcpyHostToDevice));
        HIP_TRY(p, hipMalloc((void**) &p->d_vsite_params, hp.vsite_params.size() * sizeof(double)));
        HIP_TRY(p, hipMemcpy(p->d_vsite_params, hp.vsite_params.data(), hp.vsite_params.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(p, hipMalloc((void**) &p->d_vsite_atom, hp.vsite_atom.size() * sizeof(int32_t)));
        HIP_TRY(p, hipMemcpy(p->d_vsite_atom, hp.vsite_atom.data(), hp.vsite_atom.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (!hp.slot_big.empty()) {
        HIP_TRY(p, hipMalloc((void**) &p->d_slot_big, nslots * sizeof(int32_t)));
        HIP_TRY(p, hipMemcpy(p->d_slot_big, hp.slot_big.data(), nslots * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(p, hipMalloc((void**) &p->d_bigacc, (size_t) hp.num_big * 4 * sizeof(unsigned long long)));
        HIP_TRY(p, hipMemsetAsync(p->d_bigacc, 0, (size_t) hp.num_big * 4 * sizeof(unsigned long long), p->stream));
    }
    HIP_TRY(p, hipMalloc(&p->d_fextra, nloc * 3 * rs));            // zero-initialised like HOST:79-89
    // (every fill of a plan buffer goes into the PLAN's stream: a plain hipMemset only enqueues on the null stream, which a non-blocking
    // stream does not wait for -- the reset of both accumulator copies at a switch of the cos perturbation (vvhip_set_params) could land a
    // step later and wipe kernel A's sums; found by the adapter fuzz when a host stall changed the timing, tests/test_cpp_plugin.py)
    HIP_TRY(p, hipMemsetAsync(p->d_fextra, 0, nloc * 3 * rs, p->stream));
    HIP_TRY(p, hipMalloc(&p->d_old_delta, nloc * 4 * ms));
    HIP_TRY(p, hipMemsetAsync(p->d_old_delta, 0, nloc * 4 * ms, p->stream));
    HIP_TRY(p, hipMalloc((void**) &p->d_cosz, nslots * sizeof(double)));
    HIP_TRY(p, hipMemsetAsync(p->d_cosz, 0, nslots * sizeof(double), p->stream));
    const size_t nseg = std::max<size_t>(hp.seg_mass.size() / 2, 1);
    HIP_TRY(p, hipMalloc(&p->d_comv, nseg * 4 * ms));
    HIP_TRY(p, hipMemsetAsync(p->d_comv, 0, nseg * 4 * ms, p->stream));
    HIP_TRY(p, hipMalloc((void**) &p->d_seg_base, hp.seg_base.size() * sizeof(int32_t)));
    HIP_TRY(p, hipMemcpy(p->d_seg_base, hp.seg_base.data(), hp.seg_base.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(p, hipMalloc((void**) &p->d_seg_mass, hp.seg_mass.size() * sizeof(double)));
    HIP_TRY(p, hipMemcpy(p->d_seg_mass, hp.seg_mass.data(), hp.seg_mass.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(p, hipMalloc((void**) &p->d_slot_m, nslots * sizeof(double)));
    HIP_TRY(p, hipMalloc((void**) &p->d_slot_f, nslots * sizeof(double)));
    HIP_TRY(p, hipMalloc((void**) &p->d_comw, nseg * sizeof(double)));
    HIP_TRY(p, hipMemsetAsync(p->d_comw, 0, nseg * sizeof(double), p->stream));
    HIP_TRY(p, hipMalloc(&p->d_pos_delta, nloc * 4 * ms));
    HIP_TRY(p, hipMemsetAsync(p->d_pos_delta, 0, nloc * 4 * ms, p->stream));
    HIP_TRY(p, hipMalloc((void**) &p->d_epoch, sizeof(unsigned long long)));
    HIP_TRY(p, hipMemsetAsync(p->d_epoch, 0, sizeof(unsigned long long), p->stream));
    HIP_TRY(p, hipMalloc((void**) &p->d_acc, 2 * kAccN * sizeof(unsigned long long)));
    HIP_TRY(p, hipMemsetAsync(p->d_acc, 0, 2 * kAccN * sizeof(unsigned long long), p->stream));
    // rendezvous words of the fused step: uncached (every block's thermostat wave polls what the other blocks -- on other XCDs, behind other
    // L2s -- have just stored); zero = "no step's word" (tags run from 1)
    // (+ ACC_SLOTS words: the two rows of "polled twice" flags; + 8: the "a rendezvous has failed" word behind them, vv_device.inc: rv_dead_word)
    HIP_TRY(p, hipExtMallocWithFlags((void**) &p->d_rv, (size_t) (2 * kRvCopy + vv::ACC_SLOTS + 8) * sizeof(unsigned long long), hipDeviceMallocUncached));
    HIP_TRY(p, hipMemsetAsync(p->d_rv, 0, (size_t) (2 * kRvCopy + vv::ACC_SLOTS + 8) * sizeof(unsigned long long), p->stream));
    HIP_TRY(p, hipMalloc((void**) &p->d_nh, 2 * sizeof(vv::NHDevState)));
    vv::NHDevState init[2] = {};
    for (int c = 0; c < 2; c++)
        for (int g = 0; g < 3; g++) { init[c].s.vscale[g] = 1.0; init[c].scales[g] = 1.0; }
    for (int c = 0; c < 2; c++) init[c].rv_delay = 6;      // where the wait of the fused step's rendezvous starts (it tunes itself from there)
    HIP_TRY(p, hipMemcpy(p->d_nh, init, sizeof(init), hipMemcpyHostToDevice));
    HIP_TRY(p, hipMalloc((void**) &p->d_lane_const, VVHIP_NUM_TG * sizeof(vv::ChainLaneBlock)));
    HIP_TRY(p, hipHostMalloc((void**) &p->h_status, 4 * sizeof(unsigned int), hipHostMallocMapped));
    std::memset(p->h_status, 0, 4 * sizeof(unsigned int));
    HIP_TRY(p, hipHostGetDevicePointer((void**) &p->d_status, p->h_status, 0));
    {   // launch shape for the device this plan is bound to (one block per CU balancing needs the real CU count)
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 &&
            prop.multiProcessorCount != p->num_cus) {
            p->num_cus = prop.multiProcessorCount;
            if (!p->launch_shape_forced) pick_launch_shape(p);
        }
    }
    p->bound = true;
    TRY(upload_lane_const(p));
    // The fills above are ordered in the plan's stream only.  A host may take vvhip_force_extra() / the plan's pos_delta pointer
    // right after this call and write through a blocking copy or another stream: bind is not on the hot path, so it returns
    // with every fill complete.
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    return VVHIP_OK;
}

int vvhip_set_params(vvhip_plan* p, const vvhip_params* q) {
    if (!p || !q) return VVHIP_ERR_INVALID;
    TRY(settle_recovery(p));
    // topology-affecting choices are frozen at plan creation (the reference bakes them into its tables/JIT defines)
    vvhip_params n = *q;
    n.use_com_temp_group = p->hp.params.use_com_temp_group;
    n.num_nh_chains = p->hp.params.num_nh_chains;
    if (p->hp.params.auto_set_friction && q->auto_set_friction) n.friction = p->hp.params.friction;
    if ((q->cos_acceleration != 0) && p->hp.has_ld)
        return fail(p, VVHIP_ERR_TOPOLOGY, "Langevin thermostat and periodic perturbation shouldn't be used together");
    const bool cos_switch = (p->hp.params.cos_acceleration != 0) != (n.cos_acceleration != 0);
    if (cos_switch && n.cos_acceleration == 0 && p->bound && p->fextra_virtual && !p->hp.has_ld && !p->hp.has_ef) {
        // forceExtra as the reference would have left it: the cos force of the last step, with the old acceleration (still in hp.params)
        // and the cos(kz) that step cached (K/cosineAccelerate.cu:9)
        TRY(run_a(p, vv::A_COS | vv::A_CZ_LOAD | vv::A_FE_STORE, 0));
        p->fextra_dirty = true;
    }
    if (cos_switch) p->fextra_virtual = false;
    p->hp.params = n;
    if (cos_switch && !p->launch_shape_forced) pick_launch_shape(p);      // (the cos stage sets of kernel B need more registers: another limit)
    drop_graphs(p);
    if (cos_switch && p->bound) {        // the accumulator copies are laid out by the rows in use: start the new layout from zeros
        HIP_TRY(p, hipStreamSynchronize(p->stream));
        HIP_TRY(p, hipMemsetAsync(p->d_acc, 0, 2 * kAccN * sizeof(unsigned long long), p->stream));
    }
    return upload_lane_const(p);
}

int vvhip_set_box(vvhip_plan* p, const double box[3]) {
    if (!p || !box) return VVHIP_ERR_INVALID;
    if (p->box[0] == box[0] && p->box[1] == box[1] && p->box[2] == box[2]) return VVHIP_OK;      // hosts re-send it every step
    for (int i = 0; i < 3; i++) p->box[i] = box[i];
    drop_graphs(p);                                  // the box is baked into the captured kernel arguments
    return VVHIP_OK;
}

int vvhip_masses_changed(vvhip_plan* p) {
    if (!p) return VVHIP_ERR_INVALID;
    p->mass_tab_valid = false;                       // refilled from velm.w in front of the next stage launch
    drop_graphs(p);
    return VVHIP_OK;
}

int vvhip_get_nh_state(vvhip_plan* p, vvhip_nh_state* out) {
    NEED_BOUND(p);
    TRY(settle_recovery(p));
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    HIP_TRY(p, hipMemcpy(out, &p->d_nh[p->parity].s, sizeof(*out), hipMemcpyDeviceToHost));
    return VVHIP_OK;
}
int vvhip_set_nh_state(vvhip_plan* p, const vvhip_nh_state* in) {
    NEED_BOUND(p);
    TRY(settle_recovery(p));
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    vvhip_nh_state st = *in;
    for (int g = 0; g < VVHIP_NUM_TG; g++)            // the chain's closing element is 0 by construction (API:340-376 never writes it)
        for (int i = std::max(0, std::min(p->hp.params.num_nh_chains, VVHIP_MAX_CHAINS)); i <= VVHIP_MAX_CHAINS; i++) st.eta_dot[g][i] = 0.0;
    HIP_TRY(p, hipMemcpy(&p->d_nh[p->parity].s, &st, sizeof(st), hipMemcpyHostToDevice));
    return VVHIP_OK;
}

// ------------------------------------------------------------------------------------------ fused path
int vvhip_step_middle_phases(const vvhip_plan* p) {
    if (!p) return VVHIP_ERR_INVALID;
    if (!p->hp.has_nh) return 1;
    return (cos_on(p) && !use_moments(p)) ? 3 : 2;
}

// The fused middle step without a velm round trip between its kernels: kernel A keeps the kicked velocities in registers, kernel B
// repeats the kick from velm + force (vv_args.hpp: A_NOSTORE / B_KICK).  Needs what A adds to the velocities beyond the plain
// kick to be absent or cheap to repeat: no Langevin subset and no field (kernel B repeats the cos force from the cached cos(kz), in
// the two-launch moment form only), no in-kernel velocity constraints; and a thermostat, i.e. the A -> B pair of one step
// (test hook "rekick" = 0 switches it off: comparison runs).
static bool use_rekick(const vvhip_plan* p) {
    const uint32_t ex = extra_flags(p);
    const bool extra_ok = ex == 0 || (ex == vv::A_COS && use_moments(p));
    const bool stale_extra = ex == 0 && (p->fextra_dirty || p->fextra_external);      // the kick must add what forceExtra holds
    return p->rekick && p->hp.has_nh && extra_ok && !stale_extra && !shake_on(p) && p->hp.num_big == 0;
}

// Algorithmic bytes per particle that kernel A / kernel B of the fused middle step must move (SURVEY section 8d's accounting: particle
// arrays + 6 bytes of index per pass): what bench.py prices the launches with.  Where a kernel takes the arithmetic work-item layout
// it loads no slot words, so no index bytes are counted for it; with the cos perturbation kernel A also reads posq (16 / 32 bytes) and
// the per-lane cos(kz) handed from kernel A to kernel B is counted on both sides (8 + 8 bytes: the step's design moves them).  With
// in-kernel constraints kernel A reads the positions of the cluster MEMBERS (their share of the particles, rounded to whole bytes) and
// both kernels read the cluster word and parameters (4 + 16 bytes per lane) wherever those come from memory, i.e. not in the
// arithmetic layout, where they are pattern rows in LDS.
// Does vvhip_step_middle take the one-launch step for this plan as it stands?  (The kernel itself is looked up at the first step; a pair
// of stage sets already found wanting says so here.)
static bool fused_active(const vvhip_plan* p) {
    if (!p->bound || !p->hp.params.use_middle_scheme || !p->hp.info.constraints_fused || !fused_state_ok(p) || (cos_on(p) && !use_moments(p))) return false;
    return !(p->fused_checked_b != 0 && !p->fused_ok);
}

int vvhip_algorithmic_bytes(const vvhip_plan* p, int32_t* bytes_a, int32_t* bytes_b) {
    if (!p || !bytes_a || !bytes_b) return VVHIP_ERR_INVALID;
    const int v = p->hp.precision == VVHIP_SINGLE ? 16 : 32;                       // velm: mixed4
    const int x = p->hp.precision == VVHIP_SINGLE ? 16 : 32;                       // posq (+ posqCorrection in mixed mode; double4 in double mode)
    const int xr = p->hp.precision == VVHIP_DOUBLE ? 32 : 16;                      // posq alone
    if (fused_active(p)) {
        // the one-launch step: everything is read once and written once -- R velm, R force, R position, W velm, W position + 6 bytes of
        // index; the cos perturbation and the constrained positions read nothing more (the positions are there), constraint clusters
        // their word and parameters, a virtual site its word.  (The cos(kz) the kernel keeps for vvhip_set_params is a hand-off of this
        // implementation, not counted: the figure stays a lower bound of what the step must move.)
        *bytes_a = 0;
        *bytes_b = v + 24 + x + v + x + 6;
        if (shake_on(p)) *bytes_b += 20;
        if (!p->hp.slot_vsite.empty()) *bytes_b += 8;
        return VVHIP_OK;
    }
    const bool per = p->hp.per.enabled && p->periodic_kernels;
    const bool per_a = per && (p->periodic_a || shake_on(p)), per_b = periodic_b(p);      // as run_a / run_b decide
    const int ia = per_a ? 0 : 6, ib = per_b ? 0 : 6;
    if (use_rekick(p)) { *bytes_a = v + 24 + ia; *bytes_b = v + 24 + x + v + x + ib; }    // A: R velm, R force;  B: R velm, R force, R pos, W velm, W pos
    else { *bytes_a = v + 24 + v + ia; *bytes_b = v + x + v + x + ib; }                   // A: R velm, R force, W velm;  B: R velm, R pos, W velm, W pos
    if (cos_on(p)) { *bytes_a += xr; if (use_moments(p)) { *bytes_a += 8; *bytes_b += 8; } }
    if (shake_on(p)) {
        long members = 0;
        for (size_t i = 0; i < p->hp.slots.size() / 2; i++)
            if (p->hp.slots[2 * i] >= 0 && ((uint32_t) p->hp.slots[2 * i + 1] & vv::META_SHAKE)) members++;
        const long n = std::max<long>(1, (long) (p->hp.shard_end - p->hp.shard_begin));
        *bytes_a += (int32_t) ((x * members + n / 2) / n);
        if (!per_a) *bytes_a += 20;
        if (!per_b) *bytes_b += 20;
    }
    if (!p->hp.slot_vsite.empty()) *bytes_b += 8;      // the site word of every lane (the 96-byte record of a site lane itself: well below a byte per particle)
    return VVHIP_OK;
}

// The stage sets of the one-launch step: phase 0's kick + sums and phase 1's scaling + drift of vvhip_step_middle_phase, without the
// hand-over bits between them (A_NOSTORE / B_KICK, the cos(kz) cache load).
static int step_middle_fused(vvhip_plan* p, uint32_t random_index, bool* taken) {
    *taken = false;
    if (!p->hp.params.use_middle_scheme || !p->hp.info.constraints_fused || !fused_state_ok(p)) return VVHIP_OK;
    const uint32_t stale = (extra_flags(p) == 0 && (p->fextra_dirty || p->fextra_external)) ? vv::A_FE_LOAD : 0u;
    uint32_t fa = vv::A_KICK_FULL | extra_flags(p) | stale | cons_a(p) | vv::A_KE;
    uint32_t fb = vv::B_SCALE | vv::B_DRIFT_MIDDLE | tail_flags(p) | cons_b(p);
    if (cos_on(p)) {
        if (!use_moments(p)) return VVHIP_OK;
        fa |= vv::A_BIAS | vv::A_CZ_STORE | vv::A_KE_MOM;
        fb |= vv::B_UNBIAS | vv::B_KE_MOM;
    }
    TRY(run_fused(p, fa, fb, random_index, taken));
    if (*taken && cos_on(p) && !p->hp.has_ld && !p->hp.has_ef) p->fextra_virtual = true;      // as phase 0 of the two-launch step
    return VVHIP_OK;
}

int vvhip_step_middle_phase(vvhip_plan* p, int phase, uint32_t random_index) {
    NEED_BOUND(p);
    NEED_FUSABLE(p);
    const bool rk = use_rekick(p);
    // no source of extra forces in this step: the kick adds whatever forceExtra still holds (see fextra_virtual); with sources the
    // forces are formed on the fly and the array is out of date from here on
    const uint32_t stale = (extra_flags(p) == 0 && (p->fextra_dirty || p->fextra_external)) ? vv::A_FE_LOAD : 0u;
    if (phase == 0 && cos_on(p) && !p->hp.has_ld && !p->hp.has_ef) p->fextra_virtual = true;
    const uint32_t kick = vv::A_KICK_FULL | extra_flags(p) | stale | cons_a(p) | (rk ? vv::A_NOSTORE : 0);
    const uint32_t drift = vv::B_DRIFT_MIDDLE | tail_flags(p) | cons_b(p) | (rk ? vv::B_KICK : 0);
    if (!p->hp.has_nh) {                                   // API:251: no NH particles, nothing to reduce
        if (phase != 0) return fail(p, VVHIP_ERR_INVALID, "phase out of range");
        // (with the cos perturbation the kick caches cos(kz) here too: vvhip_set_params rebuilds the stale forceExtra from it)
        TRY(run_a(p, kick | (cos_on(p) ? vv::A_CZ_STORE : 0u), random_index));
        return run_b(p, drift);
    }
    if (!cos_on(p)) {
        if (phase == 0) return run_ke(p, kick, random_index, false);
        if (phase == 1) return run_chain_and_b(p, vv::B_SCALE | drift, false);
    } else if (use_moments(p)) {                           // bias moment and group moments in one launch
        // (the per-lane cos(kz) travels from kernel A to kernel B: letting kernel B evaluate its own -- no 8-byte store / load per lane, ~45
        // more instructions per wave in B -- measured 74.5 k against 74.9 k steps/s at C4, profiles/r04b_ab_C4_cos_variants.txt)
        if (phase == 0) return run_a(p, kick | vv::A_BIAS | vv::A_CZ_STORE | vv::A_KE | vv::A_KE_MOM, random_index);
        if (phase == 1) return run_chain_and_b(p, vv::B_SCALE | vv::B_UNBIAS | vv::B_CZ_LOAD | vv::B_KE_MOM | drift, true);
    } else {                                               // API:252-259: bias -> remove -> scale -> restore
        if (phase == 0) return run_a(p, kick | vv::A_BIAS | vv::A_CZ_STORE, random_index);
        if (phase == 1) return run_ke(p, 0, 0, true);
        if (phase == 2) return run_chain_and_b(p, vv::B_SCALE | vv::B_UNBIAS | vv::B_CZ_LOAD | drift, true);
    }
    return fail(p, VVHIP_ERR_INVALID, "phase out of range");
}

int vvhip_accumulators(vvhip_plan* p, int phase, void** device_ptr, int32_t* count) {
    NEED_BOUND(p);
    if (!device_ptr || !count) return VVHIP_ERR_INVALID;
    unsigned long long* cur = p->d_acc + p->parity * acc_stride(p);
    if (use_moments(p)) { *device_ptr = cur; *count = vv::NUM_ACC * vv::ACC_SLOTS; }                  // everything kernel A produced
    else if (cos_on(p) && phase == 0) { *device_ptr = cur + 3 * vv::ACC_SLOTS; *count = vv::ACC_SLOTS; }   // bias moment slots only
    else { *device_ptr = cur; *count = 3 * vv::ACC_SLOTS; }                                          // the three 2KE sums
    return VVHIP_OK;
}

// Element-wise int64 sum of the accumulators of `phase` over all ranks, on the plan's stream (ncclSum is exact on
// integers, so every rank continues with identical bits).  No-op without a communicator.
static int exchange_accumulators(vvhip_plan* p, int phase) {
    if (use_mailbox(p)) return VVHIP_OK;   // kernel B exchanges the totals itself
    if (!p->comm) return VVHIP_OK;      // a 1-rank communicator still issues the collective (exercises the path on one GPU)
    void* ptr = nullptr;
    int32_t count = 0;
    int rc = vvhip_accumulators(p, phase, &ptr, &count);
    if (rc != VVHIP_OK) return rc;
    ScopedTimer t(p, T_OTHER);
    ncclResult_t e = rccl_api().allReduce(ptr, ptr, (size_t) count, ncclInt64, ncclSum, p->comm, p->stream);
    if (e != ncclSuccess) return fail(p, VVHIP_ERR_HIP, std::string("ncclAllReduce: ") + (rccl_api().getErrorString ? rccl_api().getErrorString(e) : "error"));
    return VVHIP_OK;
}

int vvhip_step_middle(vvhip_plan* p, uint32_t random_index) {
    NEED_BOUND(p);
    if (!p->hp.params.use_middle_scheme) return fail(p, VVHIP_ERR_INVALID, "plan was created for the classic scheme");
    {   // one launch where the plan allows it (bit for bit the two launches below)
        bool taken = false;
        TRY(step_middle_fused(p, random_index, &taken));
        if (taken) return VVHIP_OK;
    }
    const int n = vvhip_step_middle_phases(p);
    for (int ph = 0; ph < n; ph++) {
        TRY(vvhip_step_middle_phase(p, ph, random_index));
        if (ph < n - 1) TRY(exchange_accumulators(p, ph));
    }
    return VVHIP_OK;
}

// NH half-step used by the classic scheme (API:295-304, 327-336); `b_extra` is fused into the scaling kernel.
static int nh_half(vvhip_plan* p, uint32_t a_first, uint32_t random_index, uint32_t b_extra) {
    if (!p->hp.has_nh) {
        if (a_first) TRY(run_a(p, a_first, random_index));
        if (b_extra) TRY(run_b(p, b_extra));
        return VVHIP_OK;
    }
    // one launch per thermostat application where the plan allows it (as the middle scheme's step: sums, rendezvous, chain, scaling)
    if (!cos_on(p) || use_moments(p)) {
        bool taken = false;
        if (!cos_on(p)) TRY(run_fused(p, a_first | vv::A_KE, vv::B_SCALE | b_extra, random_index, &taken));
        else TRY(run_fused(p, a_first | vv::A_BIAS | vv::A_CZ_STORE | vv::A_KE | vv::A_KE_MOM, vv::B_SCALE | vv::B_UNBIAS | vv::B_KE_MOM | b_extra, random_index, &taken));
        if (taken) return VVHIP_OK;
    }
    if (!cos_on(p)) {
        TRY(run_ke(p, a_first, random_index, false));
        TRY(exchange_accumulators(p, 0));
        return run_chain_and_b(p, vv::B_SCALE | b_extra, false);
    }
    if (use_moments(p)) {
        TRY(run_a(p, a_first | vv::A_BIAS | vv::A_CZ_STORE | vv::A_KE | vv::A_KE_MOM, random_index));
        TRY(exchange_accumulators(p, 0));
        return run_chain_and_b(p, vv::B_SCALE | vv::B_UNBIAS | vv::B_CZ_LOAD | vv::B_KE_MOM | b_extra, true);
    }
    TRY(run_a(p, a_first | vv::A_BIAS | vv::A_CZ_STORE, random_index));
    TRY(exchange_accumulators(p, 0));
    TRY(run_ke(p, 0, 0, true));
    TRY(exchange_accumulators(p, 1));
    return run_chain_and_b(p, vv::B_SCALE | vv::B_UNBIAS | vv::B_CZ_LOAD | b_extra, true);
}

int vvhip_step_vv_first(vvhip_plan* p) {                   // API:295-310 (forces for the old positions are in `force`)
    NEED_BOUND(p);
    NEED_FUSABLE(p);
    return nh_half(p, 0, 0, vv::B_VV_KICK | tail_flags(p) | cons_b(p));
}

int vvhip_step_vv_second(vvhip_plan* p, uint32_t random_index) {   // API:316-336 (forces for the new positions)
    NEED_BOUND(p);
    uint32_t ex = extra_flags(p);
    if (ex) { ex |= vv::A_FE_STORE; p->fextra_dirty = true; }   // the first half of the NEXT step kicks with these (API:316-323)
    else if (p->fextra_dirty || p->fextra_external) ex = vv::A_FE_LOAD;      // no source: the kick adds what the array still holds (see fextra_virtual)
    NEED_FUSABLE(p);
    return nh_half(p, vv::A_KICK_HALF | ex | cons_a(p), random_index, 0);
}

// ------------------------------------------------------------------------------------------ kernel-interface level
int vvhip_reset_extra_force(vvhip_plan* p) {               // K/middle.cu:227-231
    NEED_BOUND(p);
    if (!p->fextra_dirty && !p->fextra_external) return VVHIP_OK;   // already zero (bind zeroes it; nothing has added to it since the last reset)
    p->fextra_dirty = false;
    ScopedTimer t(p, T_OTHER);
    const size_t nloc = (size_t) (p->hp.shard_end - p->hp.shard_begin);
    HIP_TRY(p, hipMemsetAsync(p->d_fextra, 0, nloc * 3 * sizeof_real(p->hp.precision), p->stream));
    return VVHIP_OK;
}
int vvhip_middle_kick(vvhip_plan* p) { NEED_BOUND(p); return run_a(p, ((p->fextra_dirty || p->fextra_external) ? vv::A_FE_LOAD : 0) | vv::A_KICK_FULL, 0); }
int vvhip_middle_half_drift1(vvhip_plan* p) { NEED_BOUND(p); return run_a(p, vv::A_POS1, 0); }
int vvhip_middle_half_drift2(vvhip_plan* p) { NEED_BOUND(p); return run_b(p, vv::B_POS2); }
int vvhip_middle_finish(vvhip_plan* p) {
    NEED_BOUND(p);
    uint32_t f = vv::B_POS3;
    if (p->hp.params.max_drude_distance > 0 && p->hp.has_pairs) f |= vv::B_HARDWALL;
    if (!p->hp.slot_vsite.empty()) f |= vv::B_VSITE;      // sites described to the plan follow EVERY position update (HOST:203-214): also on the split path
    return run_b(p, f);
}
int vvhip_vv_half_kick(vvhip_plan* p, int update_pos_delta) {
    NEED_BOUND(p);
    return run_a(p, ((p->fextra_dirty || p->fextra_external) ? vv::A_FE_LOAD : 0) | vv::A_KICK_HALF | (update_pos_delta ? vv::A_POSDELTA_VV : 0), 0);
}
int vvhip_vv_positions(vvhip_plan* p) {
    NEED_BOUND(p);
    uint32_t f = vv::B_VV_POS;
    if (p->hp.params.max_drude_distance > 0 && p->hp.has_pairs) f |= vv::B_HARDWALL;
    if (!p->hp.slot_vsite.empty()) f |= vv::B_VSITE;      // (as vvhip_middle_finish)
    return run_b(p, f);
}
int vvhip_scale_velocity(vvhip_plan* p) {                  // HOST:670-754 without the download/upload
    NEED_BOUND(p);
    if (!p->hp.has_nh) return VVHIP_OK;
    {
        bool taken = false;
        TRY(run_fused(p, vv::A_KE, vv::B_SCALE, 0, &taken));
        if (taken) return VVHIP_OK;
    }
    TRY(run_ke(p, 0, 0, false));
    return run_chain_and_b(p, vv::B_SCALE, false);
}
int vvhip_apply_langevin_force(vvhip_plan* p, uint32_t random_index) {
    NEED_BOUND(p);
    if (!p->hp.has_ld) return VVHIP_OK;
    p->fextra_dirty = true;
    return run_a(p, vv::A_FE_LOAD | vv::A_LD | vv::A_FE_STORE, random_index);
}
int vvhip_apply_electric_force(vvhip_plan* p) {
    NEED_BOUND(p);
    if (!p->hp.has_ef) return VVHIP_OK;
    p->fextra_dirty = true;
    return run_a(p, vv::A_FE_LOAD | vv::A_EF | vv::A_FE_STORE, 0);
}
int vvhip_apply_cosine_force(vvhip_plan* p) {
    NEED_BOUND(p);
    p->fextra_dirty = true;
    p->fextra_virtual = false;      // the array holds this step's cos force itself
    return run_a(p, vv::A_FE_LOAD | vv::A_COS | vv::A_FE_STORE, 0);
}
int vvhip_calc_velocity_bias(vvhip_plan* p) {              // HOST:1061-1082
    NEED_BOUND(p);
    TRY(run_a(p, vv::A_BIAS, 0));
    return run_chain(p, vv::C_BIAS);
}
int vvhip_remove_velocity_bias(vvhip_plan* p) { NEED_BOUND(p); return run_b(p, vv::B_BIAS_REMOVE); }
int vvhip_restore_velocity_bias(vvhip_plan* p) { NEED_BOUND(p); return run_b(p, vv::B_BIAS_RESTORE); }
int vvhip_calc_viscosity(vvhip_plan* p, double* v_max, double* inv_vis) {   // HOST:1112-1134, 8-byte download instead of N values
    NEED_BOUND(p);
    double v = 0;
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    HIP_TRY(p, hipMemcpy(&v, &p->d_nh[p->parity].s.v_bias, sizeof(double), hipMemcpyDeviceToHost));
    if (p->hp.precision == VVHIP_SINGLE) v = (double) (float) v;             // vMaxBuffer is `mixed`
    const double vol = p->box[0] * p->box[1] * p->box[2];
    if (v_max) *v_max = v;
    if (inv_vis)
        *inv_vis = v * vol * p->hp.info.inv_mass_total / p->hp.params.cos_acceleration * (2 * 3.1415926 / p->box[2]) *
                   (2 * 3.1415926 / p->box[2]);
    return VVHIP_OK;
}
int vvhip_compute_kinetic_energy(vvhip_plan* p, double* kinetic_energy) {   // HOST:233-235 delegates this to OpenMM; stand-alone hosts get it here
    NEED_BOUND(p);
    if (!kinetic_energy) return VVHIP_ERR_INVALID;
    // uses accumulator 0 of the current copy between two steps (it is zero there) and leaves it zero again
    TRY(run_a(p, vv::A_KE_PLAIN, 0));
    double acc[4];
    TRY(vvhip_debug_read_accumulators(p, acc, 1));
    *kinetic_energy = 0.5 * acc[0];
    return VVHIP_OK;
}
int vvhip_update_image_positions(vvhip_plan* p) {          // HOST:904-934
    NEED_BOUND(p);
    if (!p->hp.has_images) return VVHIP_OK;
    TRY(settle_recovery(p));
    ScopedTimer t(p, T_OTHER);
    HIP_TRY(p, vv::launch_image_pairs(p->hp.precision, p->buf.posq, p->buf.posq_correction, p->d_image_pairs,
                                      (int) p->hp.image_pairs.size() / 2, p->hp.params.mirror_location, p->stream));
    return VVHIP_OK;
}
int vvhip_force_extra(vvhip_plan* p, void** device_ptr) {
    NEED_BOUND(p);
    if (!device_ptr) return VVHIP_ERR_INVALID;
    p->fextra_external = true;
    *device_ptr = p->d_fextra;
    return VVHIP_OK;
}

// ------------------------------------------------------------------------------------------ stand-alone host support
int vvhip_device_count(int* count) {
    if (!count) return VVHIP_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = e == hipSuccess ? n : 0;
    return VVHIP_OK;
}
int vvhip_set_device(int device) { return hipSetDevice(device) == hipSuccess ? VVHIP_OK : VVHIP_ERR_HIP; }
int vvhip_malloc(void** ptr, size_t bytes) { return hipMalloc(ptr, bytes ? bytes : 16) == hipSuccess ? VVHIP_OK : VVHIP_ERR_HIP; }
int vvhip_free(void* ptr) { return hipFree(ptr) == hipSuccess ? VVHIP_OK : VVHIP_ERR_HIP; }
int vvhip_memcpy_h2d(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? VVHIP_OK : VVHIP_ERR_HIP; }
int vvhip_memcpy_d2h(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? VVHIP_OK : VVHIP_ERR_HIP; }
int vvhip_memset(void* dst, int value, size_t bytes) {      // complete on return (hipMemset alone only enqueues on the null stream, which the plans' streams do not wait for)
    return hipMemset(dst, value, bytes) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess ? VVHIP_OK : VVHIP_ERR_HIP;
}
int vvhip_synchronize(vvhip_plan* p) {
    NEED_BOUND(p);
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    if (p->rec.valid && !p->rec.replaying) {
        // the run calls since the snapshot have ended: well (the snapshot is dropped), or in a missed rendezvous (they are repeated).  An overflowed
        // accumulator or an unconverged constraint cluster next to it is what steps on incomplete sums produce; if one of them was there before, the
        // repeat raises it again.
        if (__atomic_load_n(&p->h_status[2], __ATOMIC_RELAXED) != 0 && !__atomic_load_n(&p->h_status[0], __ATOMIC_RELAXED)) return recover_rendezvous(p);
        p->rec.valid = false;
        p->rec.runs.clear();
    }
    return check_exchange_health(p);      // a mailbox time-out / accumulator overflow of the work just finished surfaces here
}
int vvhip_recovery_count(vvhip_plan* p, int64_t* recoveries) {
    if (!p || !recoveries) return VVHIP_ERR_INVALID;
    *recoveries = p->rec.recoveries;
    return VVHIP_OK;
}
int vvhip_status(vvhip_plan* p, int32_t* mailbox_timed_out, int32_t* accumulator_overflow) {
    NEED_BOUND(p);
    if (mailbox_timed_out) *mailbox_timed_out = (int32_t) __atomic_load_n(&p->h_status[0], __ATOMIC_RELAXED);
    if (accumulator_overflow) *accumulator_overflow = (int32_t) __atomic_load_n(&p->h_status[1], __ATOMIC_RELAXED);
    return VVHIP_OK;
}
int vvhip_status_words(vvhip_plan* p, int32_t words[4]) {
    NEED_BOUND(p);
    if (!words) return VVHIP_ERR_INVALID;
    for (int i = 0; i < 4; i++) words[i] = (int32_t) __atomic_load_n(&p->h_status[i], __ATOMIC_RELAXED);
    return VVHIP_OK;
}
int vvhip_fused_status(vvhip_plan* p, int32_t* active, int64_t* launches, int32_t* wait_units) {
    // (see fused_active)
    NEED_BOUND(p);
    if (launches) *launches = p->fused_launches;
    if (wait_units) {                  // where the self-tuning wait of the rendezvous stands (blocks: it lives in the device-resident state)
        *wait_units = p->fused_poll_delay;
        if (p->fused_poll_delay < 0) {
            HIP_TRY(p, hipStreamSynchronize(p->stream));
            unsigned int d = 0;
            HIP_TRY(p, hipMemcpy(&d, &p->d_nh[p->parity].rv_delay, sizeof(d), hipMemcpyDeviceToHost));
            *wait_units = (int32_t) d;
        }
    }
    if (active) *active = fused_active(p) ? 1 : 0;
    return VVHIP_OK;
}
int vvhip_status_clear(vvhip_plan* p) {
    NEED_BOUND(p);
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    std::memset(p->h_status, 0, 4 * sizeof(unsigned int));
    if (p->d_mb_ctl) HIP_TRY(p, hipMemsetAsync(p->d_mb_ctl, 0, 4 * sizeof(unsigned int), p->stream));
    if (p->d_rv) HIP_TRY(p, hipMemsetAsync(p->d_rv + 2 * kRvCopy + vv::ACC_SLOTS, 0, 8 * sizeof(unsigned long long), p->stream));      // "a rendezvous has failed" (vv_device.inc: rv_dead_word)
    p->rec.valid = false;          // (a snapshot from before the failure the caller has just acknowledged is nobody's to restore)
    p->rec.runs.clear();
    return VVHIP_OK;
}

int vvhip_stream_create(void** stream) {
    if (!stream) return VVHIP_ERR_INVALID;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return VVHIP_ERR_HIP;
    *stream = (void*) s;
    return VVHIP_OK;
}
int vvhip_stream_destroy(void* stream) { return hipStreamDestroy((hipStream_t) stream) == hipSuccess ? VVHIP_OK : VVHIP_ERR_HIP; }

int vvhip_synth_tether_force(vvhip_plan* p, const void* site, double k_tether, double k_drude) {
    NEED_BOUND(p);
    if (!site) return VVHIP_ERR_INVALID;
    TRY(settle_recovery(p));
    ScopedTimer t(p, T_OTHER, true);
    // (instrumented build: the provider stamps its waves only while vvhip_debug_step_spans numbers the launches -- its grid is not capped like the
    // kernels', and rows beyond the span buffer's 4096 per launch would be written past its end)
    vv::TetherArgs ta{p->buf.posq, site, p->buf.velm, (long long*) p->buf.force, p->d_slots,
                      p->hp.padded_num_atoms, p->hp.info.num_waves, k_tether, k_drude, p->dbg_seq >= 0 ? p->d_dbg_span : nullptr, p->dbg_parity, 0};
    if (p->dbg_seq >= 0) ta.dbg_parity = p->dbg_seq++ % 6;       // vvhip_debug_step_spans: every launch of the sequence stamps rows of its own
    HIP_TRY(p, vv::launch_tether(p->hp.precision, ta, p->block_threads, p->stream, t.e0, t.e1));
    return VVHIP_OK;
}

// integration.prepareRandomNumbers(n) for the plan-driven loops: hand out the next slice; when the buffer is exhausted
// enqueue a refill by the device generator and start over.  `force_refill` starts a graph with fresh numbers.
static int next_random_slice(vvhip_plan* p, uint32_t* index, bool force_refill) {
    *index = 0;
    if (!p->hp.has_ld) return VVHIP_OK;
    const vvhip_plan_info& in = p->hp.info;
    const uint32_t need = (uint32_t) std::max(in.num_normal_ld, 1) + 2u * (uint32_t) std::max(in.num_pairs_ld, 1);   // HOST:806-807,863
    if (need > p->buf.random_size) return fail(p, VVHIP_ERR_INVALID, "random buffer smaller than one step's demand");
    if (force_refill || p->random_pos + need > p->buf.random_size) {
        HIP_TRY(p, vv::launch_fill_normals((float4*) p->buf.random, p->buf.random_size, p->rng_seed, p->d_epoch, p->stream));
        p->random_pos = 0;
    }
    *index = p->random_pos;
    p->random_pos += need;
    return VVHIP_OK;
}

int vvhip_set_random_seed(vvhip_plan* p, uint64_t seed) {
    if (!p) return VVHIP_ERR_INVALID;
    p->rng_seed = seed;
    return VVHIP_OK;
}
int vvhip_fill_random(vvhip_plan* p) {
    NEED_BOUND(p);
    TRY(settle_recovery(p));
    if (!p->buf.random || !p->buf.random_size) return fail(p, VVHIP_ERR_INVALID, "no random buffer bound");
    HIP_TRY(p, vv::launch_fill_normals((float4*) p->buf.random, p->buf.random_size, p->rng_seed, p->d_epoch, p->stream));
    p->random_pos = 0;
    return VVHIP_OK;
}

// ---- recovery from a missed rendezvous (vvhip_plan::Recovery)
static size_t rec_bytes(const vvhip_plan* p, int which) {
    const vv::HostPlan& hp = p->hp;
    const size_t nloc = (size_t) (hp.shard_end - hp.shard_begin), rs = sizeof_real(hp.precision), ms = sizeof_mixed(hp.precision);
    switch (which) {
        case 0: return nloc * 4 * rs;                                  // posq
        case 1: return p->buf.posq_correction ? nloc * 4 * rs : 0;     // posqCorrection
        case 2: return nloc * 4 * ms;                                  // velm
        case 3: return (size_t) hp.padded_num_atoms * 3 * 8;           // force (planar int64)
        case 4: return nloc * 3 * rs;                                  // forceExtra
        default: return p->hp.has_ld ? (size_t) p->buf.random_size * sizeof(float4) : 0;      // the Langevin normals in use
    }
}
static int recovery_snapshot(vvhip_plan* p) {
    vvhip_plan::Recovery& r = p->rec;
    void** dst[6] = {&r.posq, &r.corr, &r.velm, &r.force, &r.fextra, &r.random};
    const void* src[6] = {p->buf.posq, p->buf.posq_correction, p->buf.velm, p->buf.force, p->d_fextra, p->buf.random};
    for (int i = 0; i < 6; i++) {
        const size_t n = rec_bytes(p, i);
        if (!n) continue;
        if (!*dst[i]) HIP_TRY(p, hipMalloc(dst[i], n));
        HIP_TRY(p, hipMemcpyAsync(*dst[i], src[i], n, hipMemcpyDeviceToDevice, p->stream));
    }
    if (!r.nh) HIP_TRY(p, hipMalloc((void**) &r.nh, 2 * sizeof(vv::NHDevState)));
    if (!r.epoch) HIP_TRY(p, hipMalloc((void**) &r.epoch, sizeof(unsigned long long)));
    HIP_TRY(p, hipMemcpyAsync(r.nh, p->d_nh, 2 * sizeof(vv::NHDevState), hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(p, hipMemcpyAsync(r.epoch, p->d_epoch, sizeof(unsigned long long), hipMemcpyDeviceToDevice, p->stream));
    r.parity = p->parity; r.random_pos = p->random_pos; r.fextra_dirty = p->fextra_dirty; r.fextra_virtual = p->fextra_virtual;
    r.runs.clear();
    r.valid = true;
    return VVHIP_OK;
}
// At the entry of a plan-driven run call: take the snapshot if there is none and the call is worth one; remember the call.
static int recovery_note_run(vvhip_plan* p, int kind, int nsteps, int spg, const void* site, double kt, double kd) {
    vvhip_plan::Recovery& r = p->rec;
    if (r.replaying || p->capturing || nsteps <= 0) return VVHIP_OK;
    if (!r.valid) {
        if (!r.enabled || nsteps < r.min_steps || !fused_state_ok(p) || p->comm || p->mb_on) return VVHIP_OK;
        TRY(recovery_snapshot(p));
    }
    r.runs.push_back({kind, nsteps, spg, site, kt, kd});
    return VVHIP_OK;
}
namespace {
int recover_rendezvous(vvhip_plan* p) {
    vvhip_plan::Recovery& r = p->rec;
    long long steps = 0;
    for (const auto& run : r.runs) steps += run.nsteps;
    std::fprintf(stderr, "libvvhip: the one-launch step's blocks did not meet within 0.2 s (another process's kernels on the device?): the last %lld step(s) "
                         "are repeated from the plan's snapshot with two launches per step, and the plan keeps two launches from here on\n", steps);
    void* src[6] = {r.posq, r.corr, r.velm, r.force, r.fextra, r.random};
    void* dst[6] = {p->buf.posq, p->buf.posq_correction, p->buf.velm, p->buf.force, p->d_fextra, const_cast<void*>(p->buf.random)};
    for (int i = 0; i < 6; i++) {
        const size_t n = rec_bytes(p, i);
        if (n && src[i]) HIP_TRY(p, hipMemcpyAsync(dst[i], src[i], n, hipMemcpyDeviceToDevice, p->stream));
    }
    HIP_TRY(p, hipMemcpyAsync(p->d_nh, r.nh, 2 * sizeof(vv::NHDevState), hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(p, hipMemcpyAsync(p->d_epoch, r.epoch, sizeof(unsigned long long), hipMemcpyDeviceToDevice, p->stream));
    // both accumulator copies are zero between steps; whatever the failed steps left in them goes
    HIP_TRY(p, hipMemsetAsync(p->d_acc, 0, 2 * kAccN * sizeof(unsigned long long), p->stream));
    if (p->d_bigacc) HIP_TRY(p, hipMemsetAsync(p->d_bigacc, 0, (size_t) p->hp.num_big * 4 * sizeof(unsigned long long), p->stream));
    HIP_TRY(p, hipMemsetAsync(p->d_rv + 2 * kRvCopy + vv::ACC_SLOTS, 0, 8 * sizeof(unsigned long long), p->stream));
    p->parity = r.parity; p->random_pos = r.random_pos; p->fextra_dirty = r.fextra_dirty; p->fextra_virtual = r.fextra_virtual;
    for (int w = 1; w < 4; w++) __atomic_store_n(&p->h_status[w], 0u, __ATOMIC_RELAXED);
    p->fused = false;
    p->fused_checked_b = 0;
    for (vvhip_plan::FusedCheck& c : p->fused_checks) c.b = 0;
    drop_graphs(p);
    r.recoveries++;
    r.valid = false;
    r.replaying = true;
    int rc = VVHIP_OK;
    const std::vector<vvhip_plan::Recovery::Run> runs = r.runs;
    r.runs.clear();
    for (const auto& run : runs) {
        rc = run.kind == 0 ? vvhip_run_graph(p, run.nsteps, run.spg, run.site, run.kt, run.kd) : vvhip_run_eager(p, run.nsteps, run.site, run.kt, run.kd);
        if (rc != VVHIP_OK) break;
    }
    r.replaying = false;
    if (rc != VVHIP_OK) return rc;
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    return check_exchange_health(p);
}
}

// One step of the plan-driven loops (vvhip_run_graph / vvhip_run_eager): force provider + fused step, in the scheme's order.
static int plan_step(vvhip_plan* p, const void* site, double k_tether, double k_drude, bool refill) {
    uint32_t ri = 0;
    TRY(next_random_slice(p, &ri, refill));
    if (p->hp.params.use_middle_scheme) {
        if (site) TRY(vvhip_synth_tether_force(p, site, k_tether, k_drude));
        return vvhip_step_middle(p, ri);
    }
    TRY(vvhip_step_vv_first(p));
    if (site) TRY(vvhip_synth_tether_force(p, site, k_tether, k_drude));
    return vvhip_step_vv_second(p, ri);
}

// Capture + instantiate + upload the graph of `steps_per_graph` steps for thermostat parity `q`, unless that slot already holds it.
// Nothing is launched: the physical state is untouched.
static int prepare_slot(vvhip_plan* p, int q, int steps_per_graph, const void* site, double k_tether, double k_drude) {
    hipStream_t s = p->stream;
    TRY(ensure_mass_table(p));                       // a one-off fill must not be recorded into the replayed graph
    vvhip_plan::GraphSlot& g = graph_slot(p, q, steps_per_graph, site, k_tether, k_drude);
    if (g.exec && g.steps == steps_per_graph && g.site == site && g.kt == k_tether && g.kd == k_drude) return VVHIP_OK;
    if (g.exec) { (void) hipStreamSynchronize(s); (void) hipGraphExecDestroy(g.exec); g.exec = nullptr; }      // (a replay of the one that goes may still be in flight)
    // The capture walks the host-side cursors (parity, Langevin random slice) through the graph's steps; they are put back
    // afterwards, because nothing has run yet.  A replay moves them to the graph's end (vvhip_run_graph).
    const int parity0 = p->parity;
    const uint32_t random0 = p->random_pos;
    const bool fextra_dirty0 = p->fextra_dirty, fextra_virtual0 = p->fextra_virtual;
    p->parity = q & 1;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { p->parity = parity0; return hip_fail(p, e, "hipStreamBeginCapture"); }
    p->capturing = true;
    int rc = VVHIP_OK;
    // Langevin: a captured graph begins with a refill of the random buffer (the device generator's epoch advances per refill), so every replay draws new numbers
    for (int i = 0; i < steps_per_graph && rc == VVHIP_OK; i++) rc = plan_step(p, site, k_tether, k_drude, i == 0 && p->hp.has_ld);
    p->capturing = false;
    e = hipStreamEndCapture(s, &graph);
    g.random_end = p->random_pos;
    p->parity = parity0; p->random_pos = random0; p->fextra_dirty = fextra_dirty0; p->fextra_virtual = fextra_virtual0;
    if (rc != VVHIP_OK) { if (graph) (void) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return hip_fail(p, e, "hipStreamEndCapture");
    e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void) hipGraphDestroy(graph);
    if (e != hipSuccess) { g.exec = nullptr; return hip_fail(p, e, "hipGraphInstantiate"); }
    (void) hipGraphUpload(g.exec, s);                // pay the first launch's set-up here, not in the caller's timed region
    g.steps = steps_per_graph; g.site = site; g.kt = k_tether; g.kd = k_drude;
    return VVHIP_OK;
}

// Both parities' executables, ready to launch.  Hosts call this outside any timed region (bench.py does, after its warm-up);
// vvhip_run_graph prepares the slot of the current parity itself when it is missing.
int vvhip_graph_prepare(vvhip_plan* p, int steps_per_graph, const void* site, double k_tether, double k_drude) {
    NEED_BOUND(p);
    if (steps_per_graph < 1) return VVHIP_ERR_INVALID;
    if (steps_per_graph % 2) steps_per_graph += 1;   // the thermostat double-buffers by step parity: a graph must hold an even number of steps
    if (!p->stream) return fail(p, VVHIP_ERR_INVALID, "graph capture needs a non-null stream in vvhip_buffers.stream");
    TRY(prepare_slot(p, p->parity, steps_per_graph, site, k_tether, k_drude));
    return prepare_slot(p, p->parity ^ 1, steps_per_graph, site, k_tether, k_drude);
}

int vvhip_run_graph(vvhip_plan* p, int nsteps, int steps_per_graph, const void* site, double k_tether, double k_drude) {
    NEED_BOUND(p);
    if (nsteps < 0 || steps_per_graph < 1) return VVHIP_ERR_INVALID;
    if (steps_per_graph % 2) steps_per_graph += 1;
    TRY(check_exchange_health(p));
    hipStream_t s = p->stream;
    if (!s) return fail(p, VVHIP_ERR_INVALID, "graph capture needs a non-null stream in vvhip_buffers.stream");
    TRY(recovery_note_run(p, 0, nsteps, steps_per_graph, site, k_tether, k_drude));
    struct InLoop { vvhip_plan* p; bool was; InLoop(vvhip_plan* q) : p(q), was(q->rec.in_loop) { p->rec.in_loop = true; } ~InLoop() { p->rec.in_loop = was; } } in_loop(p);
    const bool middle = p->hp.params.use_middle_scheme;
    // classic scheme (API:272-338): every step is first half -> forces -> second half, and the first half needs the forces of the
    // current positions; they are (re)computed once per call here, outside the replayed part
    if (!middle && site && nsteps > 0) TRY(vvhip_synth_tether_force(p, site, k_tether, k_drude));
    int done = 0;
    if (nsteps >= steps_per_graph) {
        TRY(prepare_slot(p, p->parity, steps_per_graph, site, k_tether, k_drude));     // no-op when the slot of this parity is ready
        const vvhip_plan::GraphSlot& g = graph_slot(p, p->parity, steps_per_graph, site, k_tether, k_drude);
        for (; done + steps_per_graph <= nsteps; done += steps_per_graph) HIP_TRY(p, hipGraphLaunch(g.exec, s));
        p->random_pos = g.random_end;                // an even number of steps: the parity is where it was
        if (!middle && extra_flags(p)) p->fextra_dirty = true;
        if (middle && cos_on(p) && !p->hp.has_ld && !p->hp.has_ef) p->fextra_virtual = true;    // what the replayed steps' phase 0 would have set
    }
    for (; done < nsteps; done++) TRY(plan_step(p, site, k_tether, k_drude, false));
    return VVHIP_OK;
}

// The stage bits vvhip_step_middle launches kernel A (kernel = 0) / kernel B with for this plan (timing and probe entry points)
int vvhip_debug_launch_shape(const vvhip_plan* p, int32_t shape[4]) {
    if (!p || !shape) return VVHIP_ERR_INVALID;
    shape[0] = p->block_threads; shape[1] = p->grid_cap_a; shape[2] = p->grid_cap_b;
    shape[3] = fused_shape_ok(p) ? p->block_threads / 64 : 0;
    return VVHIP_OK;
}
int vvhip_debug_fused_flags(vvhip_plan* p, int kernel, uint32_t* flags) {
    NEED_BOUND(p);
    if (!flags) return VVHIP_ERR_INVALID;
    const uint32_t mom_a = use_moments(p) ? (vv::A_KE | vv::A_KE_MOM) : 0, mom_b = use_moments(p) ? vv::B_KE_MOM : 0;
    const bool rk = use_rekick(p);
    if (kernel == 0) *flags = vv::A_KICK_FULL | (rk ? vv::A_NOSTORE : 0) | extra_flags(p) | cons_a(p) | (p->hp.has_nh ? (cos_on(p) ? (vv::A_BIAS | vv::A_CZ_STORE | mom_a) : vv::A_KE) : 0);
    const bool split = p->hp.info.num_waves >= p->split_chain_waves && !use_mailbox(p);    // as run_chain_and_b decides
    if (kernel != 0) *flags = vv::B_DRIFT_MIDDLE | (rk ? vv::B_KICK : 0) | tail_flags(p) | cons_b(p) | (p->hp.has_nh ? (((p->hp.params.num_nh_chains <= 4 && !split) ? vv::B_CHAIN : 0) | vv::B_SCALE | (cos_on(p) ? (vv::B_UNBIAS | vv::B_CZ_LOAD | mom_b) : 0)) : 0);
    return VVHIP_OK;
}
int vvhip_time_kernel(vvhip_plan* p, int kernel, uint32_t flags, int reps, double* ms_per_launch) {
    NEED_BOUND(p);
    if (reps < 1 || !ms_per_launch) return VVHIP_ERR_INVALID;
    if (flags == 0xFFFFFFFFu) TRY(vvhip_debug_fused_flags(p, kernel, &flags));     // the stage bits vvhip_step_middle uses for this plan
    hipEvent_t e0, e1;
    HIP_TRY(p, hipEventCreate(&e0));
    HIP_TRY(p, hipEventCreate(&e1));
    const int parity = p->parity;
    const bool was_timing = p->timing;
    p->timing = false;
    int rc = VVHIP_OK;
    for (int i = 0; i < 3 && rc == VVHIP_OK; i++) { p->parity = parity; rc = kernel == 0 ? run_a(p, flags, 0) : run_b(p, flags); }
    HIP_TRY(p, hipEventRecord(e0, p->stream));
    for (int i = 0; i < reps && rc == VVHIP_OK; i++) { p->parity = parity; rc = kernel == 0 ? run_a(p, flags, 0) : run_b(p, flags); }
    HIP_TRY(p, hipEventRecord(e1, p->stream));
    p->parity = parity;
    p->timing = was_timing;
    if (rc != VVHIP_OK) return rc;
    HIP_TRY(p, hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(p, hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    *ms_per_launch = (double) ms / reps;
    return VVHIP_OK;
}

// Instrumented build (-DVV_KERNEL_TIMESTAMPS): one launch of kernel B with `flags`, shader-clock stamps of block `block`:
// out[w*16 + k] for tile waves w = 0.. (k = 0 entry, 1 loads arrived, 2 prep done, 3 scales received, 4 compute done, 5 stores
// drained) and w = 7 for the thermostat wave (0 entry, 1 accumulators folded, 2 chain done, 3 after the barrier).
int vvhip_debug_timestamps(vvhip_plan* p, uint32_t flags, int block, long long out[128]) {
    // bit 31 of `flags` selects kernel A (stamps: 0 entry, 1 velm arrived, 2 kicked + stored, 3 tile loop done, 4 sums added)
    NEED_BOUND(p);
#ifndef VV_KERNEL_TIMESTAMPS
    (void) flags; (void) block; (void) out;
    return fail(p, VVHIP_ERR_UNSUPPORTED, "not an instrumented build");
#else
    if (!p->d_dbg) HIP_TRY(p, hipMalloc((void**) &p->d_dbg, 128 * sizeof(long long)));
    HIP_TRY(p, hipMemsetAsync(p->d_dbg, 0, 128 * sizeof(long long), p->stream));
    p->dbg_block = block;
    const int parity = p->parity;
    int rc = (flags & 0x80000000u) ? run_a(p, flags & 0x7FFFFFFFu, 0) : run_b(p, flags);
    p->parity = parity;
    if (rc != VVHIP_OK) return rc;
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    HIP_TRY(p, hipMemcpy(out, p->d_dbg, 128 * sizeof(long long), hipMemcpyDeviceToHost));
    return VVHIP_OK;
#endif
}

// Instrumented build: ONE real step of the one-launch path (it advances the state), shader-clock stamps of block `block`: tile waves
// w = 0..6: 0 entry, 6 loads arrived + extra forces, 7 kick + sums done, 8 partials in LDS, 9 behind barrier 1, 1 / 2 preparation, 3 scales
// received, 4 compute done, 5 stores drained; thermostat wave (w = 7): 0 entry, 6 at barrier 1, 7 behind it, 8 published, 9 all blocks' words
// held, 10 = poll rounds (a count, not a time), 1 folded, 4 ke2, 5 released, 2 chain done, 3 state stored.
int vvhip_debug_timestamps_fused(vvhip_plan* p, int block, long long out[128]) {
    NEED_BOUND(p);
#ifndef VV_KERNEL_TIMESTAMPS
    (void) block; (void) out;
    return fail(p, VVHIP_ERR_UNSUPPORTED, "not an instrumented build");
#else
    if (!p->d_dbg) HIP_TRY(p, hipMalloc((void**) &p->d_dbg, 128 * sizeof(long long)));
    HIP_TRY(p, hipMemsetAsync(p->d_dbg, 0, 128 * sizeof(long long), p->stream));
    p->dbg_block = block;
    bool taken = false;
    TRY(step_middle_fused(p, 0, &taken));
    if (!taken) return fail(p, VVHIP_ERR_UNSUPPORTED, "the plan does not take the one-launch step");
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    HIP_TRY(p, hipMemcpy(out, p->d_dbg, 128 * sizeof(long long), hipMemcpyDeviceToHost));
    return VVHIP_OK;
#endif
}

// Instrumented build: `reps` back-to-back launches of kernel A (kernel = 0) or B (1) with `flags`; every wave stamps the 100 MHz
// wall clock at entry and (after draining its memory operations) at exit.  out[0] = first entry -> last exit of the last launch,
// out[1] = last exit of the launch before -> first entry of the last launch, out[2] = median wave entry - first entry,
// out[3] = median wave lifetime (all ns).
int vvhip_debug_span(vvhip_plan* p, int kernel, uint32_t flags, int reps, double out[8]) {
    NEED_BOUND(p);
#ifndef VV_KERNEL_TIMESTAMPS
    (void) kernel; (void) flags; (void) reps; (void) out;
    return fail(p, VVHIP_ERR_UNSUPPORTED, "not an instrumented build");
#else
    const size_t per = (size_t) 4096 * 8 * 2;
    if (!p->d_dbg_span) HIP_TRY(p, hipMalloc((void**) &p->d_dbg_span, 6 * per * sizeof(long long)));
    HIP_TRY(p, hipMemsetAsync(p->d_dbg_span, 0, 6 * per * sizeof(long long), p->stream));
    const int parity = p->parity;
    int rc = VVHIP_OK;
    for (int i = 0; i < reps && rc == VVHIP_OK; i++) { p->parity = parity; p->dbg_parity = i & 1; rc = kernel == 0 ? run_a(p, flags, 0) : run_b(p, flags); }
    p->parity = parity;
    const int last = (reps - 1) & 1;
    p->dbg_parity = 0;
    long long* keep = p->d_dbg_span;
    p->d_dbg_span = nullptr;                       // later launches run unstamped
    if (rc != VVHIP_OK) { p->d_dbg_span = keep; return rc; }
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    std::vector<long long> h(2 * per);
    HIP_TRY(p, hipMemcpy(h.data(), keep, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    p->d_dbg_span = keep;
    // the grid size is not known here: the launch of parity q used rows [q * grid, (q+1) * grid); find them by scanning non-zero pairs
    std::vector<long long> in[2], ex[2];
    std::vector<int> blk;
    const int grid = (int) vv_last_grid();
    for (int q = 0; q < 2; q++)
        for (size_t r = 0; r < (size_t) grid * 8; r++) {
            const long long a0 = h[(((size_t) q * 4096) * 8 + r) * 2], a1 = h[(((size_t) q * 4096) * 8 + r) * 2 + 1];
            if (a0 && a1) { in[q].push_back(a0); ex[q].push_back(a1); if (q == ((reps - 1) & 1)) blk.push_back((int) (r / 8)); }
        }
    if (in[0].empty() || in[1].empty()) return fail(p, VVHIP_ERR_INVALID, "no stamps recorded");
    const int L = last, P = 1 - last;
    const long long first_in = *std::min_element(in[L].begin(), in[L].end()), last_out = *std::max_element(ex[L].begin(), ex[L].end());
    const long long prev_out = *std::max_element(ex[P].begin(), ex[P].end());
    std::vector<long long> rel, life;
    for (size_t k = 0; k < in[L].size(); k++) { rel.push_back(in[L][k] - first_in); life.push_back(ex[L][k] - in[L][k]); }
    size_t worst = 0;
    for (size_t k = 0; k < in[L].size(); k++) if (ex[L][k] > ex[L][worst]) worst = k;
    out[6] = (double) blk[worst]; out[7] = (double) (in[L][worst] - first_in) * 10.0;
    std::sort(rel.begin(), rel.end()); std::sort(life.begin(), life.end());
    out[4] = (double) life[life.size() * 9 / 10] * 10.0; out[5] = (double) life.back() * 10.0;
    out[0] = (double) (last_out - first_in) * 10.0; out[1] = (double) (first_in - prev_out) * 10.0;
    out[2] = (double) rel[rel.size() / 2] * 10.0; out[3] = (double) life[life.size() / 2] * 10.0;
    return VVHIP_OK;
#endif
}

// Instrumented build: `nsteps` consecutive fused steps enqueued from here (force provider -> kernel A -> kernel B; middle scheme), every wave
// of the LAST TWO steps stamping the 100 MHz wall clock at entry and (memory operations drained) at exit.  For the six launches
// (provider, A, B of the step before the last; provider, A, B of the last) out[l*6 ..] = first wave in, median wave in, last wave in,
// first wave out, median wave out, last wave out, in ns after the first entry of the first of them.  What a kernel costs IN ITS PLACE:
// ramp, body, tail and the gap to its neighbours, none of which a profiler's per-kernel duration separates.
int vvhip_debug_step_spans(vvhip_plan* p, int nsteps, const void* site, double k_tether, double k_drude, double out[36]) {
    NEED_BOUND(p);
#ifndef VV_KERNEL_TIMESTAMPS
    (void) nsteps; (void) site; (void) k_tether; (void) k_drude; (void) out;
    return fail(p, VVHIP_ERR_UNSUPPORTED, "not an instrumented build");
#else
    if (nsteps < 2 || !site || !out || !p->hp.params.use_middle_scheme || vvhip_step_middle_phases(p) != 2) return VVHIP_ERR_INVALID;
    const size_t per = (size_t) 4096 * 8 * 2;
    if (!p->d_dbg_span) HIP_TRY(p, hipMalloc((void**) &p->d_dbg_span, 6 * per * sizeof(long long)));
    TRY(ensure_mass_table(p));
    for (int i = 0; i < nsteps - 2; i++) TRY(plan_step(p, site, k_tether, k_drude, false));      // warm: same launches, rows overwritten below
    HIP_TRY(p, hipMemsetAsync(p->d_dbg_span, 0, 6 * per * sizeof(long long), p->stream));
    long long* keep = p->d_dbg_span;
    p->dbg_seq = 0;
    int rc = VVHIP_OK;
    const long long fused_before = p->fused_launches;
    for (int i = 0; i < 2 && rc == VVHIP_OK; i++) rc = plan_step(p, site, k_tether, k_drude, false);
    p->dbg_seq = -1;
    p->d_dbg_span = nullptr;
    if (rc != VVHIP_OK) { p->d_dbg_span = keep; return rc; }
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    std::vector<long long> h(6 * per);
    HIP_TRY(p, hipMemcpy(h.data(), keep, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    p->d_dbg_span = keep;
    long long t0 = 0;
    // (the one-launch step: provider + one kernel per step, four launches; rows 4 and 5 stay zero)
    const int nlaunch = p->fused_launches > fused_before ? 4 : 6;
    for (int l = 0; l < 36; l++) out[l] = 0;
    for (int l = 0; l < nlaunch; l++) {
        std::vector<long long> in, ex;
        for (size_t r = 0; r < (size_t) 4096 * 8; r++) {
            const long long a0 = h[((size_t) l * 4096 * 8 + r) * 2], a1 = h[((size_t) l * 4096 * 8 + r) * 2 + 1];
            if (a0 && a1) { in.push_back(a0); ex.push_back(a1); }
        }
        if (in.empty()) return fail(p, VVHIP_ERR_INVALID, "no stamps recorded for one of the launches");
        std::sort(in.begin(), in.end()); std::sort(ex.begin(), ex.end());
        if (l == 0) t0 = in.front();
        const long long v[6] = {in.front(), in[in.size() / 2], in.back(), ex.front(), ex[ex.size() / 2], ex.back()};
        for (int k = 0; k < 6; k++) out[l * 6 + k] = (double) (v[k] - t0) * 10.0;
    }
    return VVHIP_OK;
#endif
}

int vvhip_comm_unique_id(void* id128) {
    if (!id128) return VVHIP_ERR_INVALID;
    RcclApi& r = rccl_api();
    if (!r.ok) return VVHIP_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r.getUniqueId(&id) != ncclSuccess) return VVHIP_ERR_HIP;
    std::memcpy(id128, &id, sizeof(id));
    return VVHIP_OK;
}
int vvhip_comm_init(vvhip_plan* p, const void* id128, int nranks, int rank) {
    NEED_BOUND(p);
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(p, VVHIP_ERR_INVALID, "bad communicator arguments");
    RcclApi& r = rccl_api();
    if (!r.ok) return fail(p, VVHIP_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    if (p->comm) { (void) r.commDestroy(p->comm); p->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t e = r.commInitRank(&p->comm, nranks, id, rank);
    if (e != ncclSuccess) { p->comm = nullptr; return fail(p, VVHIP_ERR_HIP, std::string("ncclCommInitRank: ") + (r.getErrorString ? r.getErrorString(e) : "error")); }
    p->comm_ranks = nranks;
    drop_graphs(p);
    return VVHIP_OK;
}
int vvhip_comm_count(vvhip_plan* p, int32_t* ranks) {
    if (!p || !ranks) return VVHIP_ERR_INVALID;
    *ranks = 0;
    if (!p->comm) return VVHIP_OK;                   // no communicator: 0
    RcclApi& r = rccl_api();
    int n = p->comm_ranks;
    if (r.commCount && r.commCount(p->comm, &n) != ncclSuccess) return fail(p, VVHIP_ERR_HIP, "ncclCommCount failed");
    *ranks = n;
    return VVHIP_OK;
}
int vvhip_peer_access(int device, int peer_device, int32_t* can_access) {
    if (!can_access) return VVHIP_ERR_INVALID;
    int can = 0;
    if (device == peer_device) { *can_access = 1; return VVHIP_OK; }
    if (hipDeviceCanAccessPeer(&can, device, peer_device) != hipSuccess) return VVHIP_ERR_HIP;
    *can_access = can;
    return VVHIP_OK;
}
// ---- xGMI mailbox (include/vvhip.h): create -> exchange the 64-byte handles by any means -> connect
static void mailbox_release(vvhip_plan* p) {
    p->mb_on = false;
    p->mb_shared_device = false;
    p->mb_device_ranks = 1;
    for (void* m : p->mb_opened) (void) hipIpcCloseMemHandle(m);
    p->mb_opened.clear();
    if (p->d_mb_peers) { (void) hipFree(p->d_mb_peers); p->d_mb_peers = nullptr; }
    if (p->d_mb_ctl) { (void) hipFree(p->d_mb_ctl); p->d_mb_ctl = nullptr; }
    if (p->mb_local) { (void) hipFree(p->mb_local); p->mb_local = nullptr; }
    p->mb_ranks = 0;
}
int vvhip_mailbox_create(vvhip_plan* p, int nranks, int rank, void* handle64) {
    NEED_BOUND(p);
    if (!handle64 || nranks < 1 || nranks > vv::MB_MAX_RANKS || rank < 0 || rank >= nranks)
        return fail(p, VVHIP_ERR_INVALID, "mailbox: 1 <= ranks <= 16, 0 <= rank < ranks");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI hands the IPC handle over as 64 bytes");
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    mailbox_release(p);
    drop_graphs(p);
    const size_t bytes = (size_t) 2 * nranks * vv::MB_WORDS * sizeof(unsigned long long);
    // uncached: peers' stores land in this GPU's memory over xGMI and must be seen by loads that would otherwise hit in L2
    HIP_TRY(p, hipExtMallocWithFlags((void**) &p->mb_local, std::max(bytes, (size_t) 4096), hipDeviceMallocUncached));
    HIP_TRY(p, hipMemsetAsync(p->mb_local, 0, std::max(bytes, (size_t) 4096), p->stream));
    HIP_TRY(p, hipMalloc((void**) &p->d_mb_ctl, 4 * sizeof(unsigned int)));
    HIP_TRY(p, hipMemsetAsync(p->d_mb_ctl, 0, 4 * sizeof(unsigned int), p->stream));
    HIP_TRY(p, hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    HIP_TRY(p, hipIpcGetMemHandle(&h, p->mb_local));
    std::memcpy(handle64, &h, 64);
    p->mb_ranks = nranks;
    p->mb_rank = rank;
    return VVHIP_OK;
}
int vvhip_mailbox_connect(vvhip_plan* p, const void* handles) {
    NEED_BOUND(p);
    if (!p->mb_local || !handles) return fail(p, VVHIP_ERR_INVALID, "vvhip_mailbox_create has not been called");
    std::vector<unsigned long long*> peers((size_t) p->mb_ranks, nullptr);
    // A second connect: what vvhip_mailbox_destroy does first -- captured step graphs carry the OLD peer table's address and the peers' box
    // addresses in their kernel arguments (a replay after the free below would read unmapped memory), launches still in flight use them too,
    // and whether the step may be one launch depends on who shares the device (re-evaluated: fused_checked_*).
    TRY(settle_recovery(p));
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    drop_graphs(p);
    p->fused_checked_b = 0;
    for (vvhip_plan::FusedCheck& c : p->fused_checks) c.b = 0;
    p->mb_shared_device = false;       // (a second connect must not count the first one's ranks again)
    p->mb_device_ranks = 1;
    for (void* m : p->mb_opened) (void) hipIpcCloseMemHandle(m);
    p->mb_opened.clear();
    if (p->d_mb_peers) { (void) hipFree(p->d_mb_peers); p->d_mb_peers = nullptr; }
    for (int r = 0; r < p->mb_ranks; r++) {
        if (r == p->mb_rank) { peers[r] = p->mb_local; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char*) handles + (size_t) r * 64, 64);
        void* m = nullptr;
        HIP_TRY(p, hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
        p->mb_opened.push_back(m);
        peers[r] = (unsigned long long*) m;
        // whose memory is it?  A box on this very device means that rank shares the GPU with this one
        hipPointerAttribute_t attr;
        int dev = -1;
        if (hipGetDevice(&dev) == hipSuccess && hipPointerGetAttributes(&attr, m) == hipSuccess && attr.device == dev) { p->mb_shared_device = true; p->mb_device_ranks++; }
        else (void) hipGetLastError();
    }
    HIP_TRY(p, hipMalloc((void**) &p->d_mb_peers, peers.size() * sizeof(void*)));
    HIP_TRY(p, hipMemcpy(p->d_mb_peers, peers.data(), peers.size() * sizeof(void*), hipMemcpyHostToDevice));
    p->mb_on = true;
    return VVHIP_OK;
}
int vvhip_mailbox_status(vvhip_plan* p, int32_t* active, int32_t* timed_out) {
    NEED_BOUND(p);
    if (active) *active = use_mailbox(p) ? 1 : 0;
    if (timed_out) {
        *timed_out = 0;
        if (p->d_mb_ctl) {
            unsigned int ctl[4];
            HIP_TRY(p, hipStreamSynchronize(p->stream));
            HIP_TRY(p, hipMemcpy(ctl, p->d_mb_ctl, sizeof ctl, hipMemcpyDeviceToHost));
            *timed_out = (int32_t) ctl[0];
        }
    }
    return VVHIP_OK;
}
int vvhip_mailbox_layout(vvhip_plan* p, int32_t* shared_device, int32_t* arithmetic_layout) {
    NEED_BOUND(p);
    if (shared_device) *shared_device = p->mb_shared_device ? 1 : 0;
    if (arithmetic_layout) *arithmetic_layout = (use_mailbox(p) && periodic_b(p)) ? 1 : 0;
    return VVHIP_OK;
}
int vvhip_mailbox_destroy(vvhip_plan* p) {
    NEED_BOUND(p);
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    drop_graphs(p);
    mailbox_release(p);
    return VVHIP_OK;
}

int vvhip_comm_destroy(vvhip_plan* p) {
    if (!p) return VVHIP_ERR_INVALID;
    if (p->comm) { (void) hipStreamSynchronize(p->stream); (void) rccl_api().commDestroy(p->comm); p->comm = nullptr; p->comm_ranks = 1; }
    return VVHIP_OK;
}
int vvhip_run_eager(vvhip_plan* p, int nsteps, const void* site, double k_tether, double k_drude) {
    NEED_BOUND(p);
    if (nsteps < 0) return VVHIP_ERR_INVALID;
    TRY(check_exchange_health(p));
    TRY(recovery_note_run(p, 1, nsteps, 0, site, k_tether, k_drude));
    struct InLoop { vvhip_plan* p; bool was; InLoop(vvhip_plan* q) : p(q), was(q->rec.in_loop) { p->rec.in_loop = true; } ~InLoop() { p->rec.in_loop = was; } } in_loop(p);
    if (!p->hp.params.use_middle_scheme && site && nsteps > 0) TRY(vvhip_synth_tether_force(p, site, k_tether, k_drude));   // see vvhip_run_graph
    for (int i = 0; i < nsteps; i++) TRY(plan_step(p, site, k_tether, k_drude, false));
    return VVHIP_OK;
}

// The same steps through the per-KernelImpl entry points in VVIntegrator::stepMiddle's order (API:237-268) -- what the OpenMM adapter
// issues when constraints it cannot fuse force OpenMM's solver between the stages (the solver's own launches are not included).
int vvhip_run_eager_unfused(vvhip_plan* p, int nsteps, const void* site, double k_tether, double k_drude) {
    NEED_BOUND(p);
    if (nsteps < 0) return VVHIP_ERR_INVALID;
    if (!p->hp.params.use_middle_scheme) return fail(p, VVHIP_ERR_INVALID, "plan was created for the classic scheme");
    for (int i = 0; i < nsteps; i++) {
        uint32_t ri = 0;
        TRY(next_random_slice(p, &ri, false));
        if (site) TRY(vvhip_synth_tether_force(p, site, k_tether, k_drude));
        TRY(vvhip_reset_extra_force(p));
        if (p->hp.has_ld) TRY(vvhip_apply_langevin_force(p, ri));
        if (p->hp.has_ef) TRY(vvhip_apply_electric_force(p));
        if (cos_on(p)) TRY(vvhip_apply_cosine_force(p));
        TRY(vvhip_middle_kick(p));                  // (applyVelocityConstraints would run here)
        TRY(vvhip_middle_half_drift1(p));
        if (p->hp.has_nh) {
            if (cos_on(p)) { TRY(vvhip_calc_velocity_bias(p)); TRY(vvhip_remove_velocity_bias(p)); }
            TRY(vvhip_scale_velocity(p));
            if (cos_on(p)) TRY(vvhip_restore_velocity_bias(p));
        }
        TRY(vvhip_middle_half_drift2(p));           // (applyConstraints would run here)
        TRY(vvhip_middle_finish(p));
        if (p->hp.has_images) TRY(vvhip_update_image_positions(p));
    }
    return VVHIP_OK;
}

int vvhip_set_trace(vvhip_plan* p, int enable) {
    if (!p) return VVHIP_ERR_INVALID;
    p->trace = enable != 0;
    return VVHIP_OK;
}
int vvhip_generic_launches(vvhip_plan* p, int64_t counts[2], uint32_t stage_sets[2]) {
    if (!p || !counts) return VVHIP_ERR_INVALID;
    for (int k = 0; k < 2; k++) { counts[k] = p->generic_launches[k]; if (stage_sets) stage_sets[k] = p->generic_flags[k]; }
    return VVHIP_OK;
}
int vvhip_rtc_mode(int mode) { return vv::set_rtc_mode(mode); }
int vvhip_rtc_stats(int64_t counts[3], double* compile_seconds) {
    if (!counts) return VVHIP_ERR_INVALID;
    counts[0] = (int64_t) vv::vv_rtc_compiled.load(); counts[1] = (int64_t) vv::vv_rtc_launches[0].load(); counts[2] = (int64_t) vv::vv_rtc_launches[1].load();
    if (compile_seconds) *compile_seconds = vv::vv_rtc_compile_seconds;
    return VVHIP_OK;
}
int vvhip_rtc_failures(int64_t* failed) {
    if (!failed) return VVHIP_ERR_INVALID;
    *failed = (int64_t) vv::vv_rtc_failed.load();
    return VVHIP_OK;
}
int vvhip_timing_enable(vvhip_plan* p, int enable) {
    if (!p) return VVHIP_ERR_INVALID;
    p->timing = enable != 0;
    p->timing_kernels_only = enable == 2;
    if (enable > 2) {                  // enable = n > 2: as 2, with n events prepared now (a timed run of n / 2 launches creates none)
        p->timing_kernels_only = true;
        for (int i = (int) p->event_pool.size(); i < enable; i++) { hipEvent_t e = nullptr; if (hipEventCreate(&e) == hipSuccess) p->event_pool.push_back(e); }
    }
    return VVHIP_OK;
}
int vvhip_timing_read(vvhip_plan* p, double* ms_a, double* ms_b, double* ms_other, int32_t* launches) {
    NEED_BOUND(p);
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    double tot[3] = {0, 0, 0};
    int32_t n[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
        for (auto& e : p->events[c]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { tot[c] += ms; n[c]++; }
            p->event_pool.push_back(e.first);
            p->event_pool.push_back(e.second);
        }
        p->events[c].clear();
    }
    if (ms_a) *ms_a = tot[0];
    if (ms_b) *ms_b = tot[1];
    if (ms_other) *ms_other = tot[2];
    if (launches) { launches[0] = n[0]; launches[1] = n[1]; launches[2] = n[2]; }
    return VVHIP_OK;
}

// ------------------------------------------------------------------------------------------ test hooks
int vvhip_debug_launch(vvhip_plan* p, int kernel, uint32_t flags, uint32_t random_index) {
    NEED_BOUND(p);
    if (kernel == 0) return run_a(p, flags, random_index);
    if (kernel == 1) return run_b(p, flags);
    if (kernel == 2) return run_chain(p, flags);
    return VVHIP_ERR_INVALID;
}
int vvhip_debug_read_accumulators(vvhip_plan* p, double out[4], int zero_after) {
    NEED_BOUND(p);
    static long long raw[vv::NUM_ACC * vv::ACC_SLOTS];
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    HIP_TRY(p, hipMemcpy(raw, p->d_acc + p->parity * acc_stride(p), 4 * vv::ACC_SLOTS * sizeof(long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) {                 // the ABI hands out the three group sums and the bias moment
        long long s = 0;
        for (int j = 0; j < vv::ACC_SLOTS; j++) s += raw[i * vv::ACC_SLOTS + j];
        out[i] = (double) s * p->acc_inv_scale[i];
    }
    if (zero_after) HIP_TRY(p, hipMemsetAsync(p->d_acc + p->parity * acc_stride(p), 0, 4 * vv::ACC_SLOTS * sizeof(long long), p->stream));
    return VVHIP_OK;
}
int vvhip_debug_set_scales(vvhip_plan* p, const double scales[4]) {
    NEED_BOUND(p);
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    HIP_TRY(p, hipMemcpy(p->d_nh[p->parity].scales, scales, 4 * sizeof(double), hipMemcpyHostToDevice));
    return VVHIP_OK;
}

int vvhip_debug_old_delta(vvhip_plan* p, void** device_ptr) {
    NEED_BOUND(p);
    if (!device_ptr) return VVHIP_ERR_INVALID;
    *device_ptr = p->d_old_delta;
    return VVHIP_OK;
}

}  // extern "C"
