/*
 * vvhip.h -- C ABI of libvvhip.so: the MI355X-native (HIP, gfx950) implementation of the per-step
 * hot path of z-gong/openmm-velocityVerlet.
 *
 * Boundary.  In the reference the hot path sits behind seven OpenMM KernelImpl interfaces
 * (openmmapi/include/openmm/VVKernels.h:48-270) that a platform plugin implements in C++
 * (platforms/cuda/include/CudaVVKernels.h, platforms/cuda/src/CudaVVKernels.cpp).  This header is
 * the same surface flattened to C: one entry point per KernelImpl virtual (split where the
 * reference method calls OpenMM's constraint solvers in the middle), raw device pointers instead
 * of CudaArray, plain scalars instead of VVIntegrator getters.  The OpenMM-facing C++ adapters that
 * call it (HipVVKernelFactory, Hip*Kernel) live in platforms/hip/ and are described in
 * INTEGRATION.md.  Citations below are reference file:line under /root/reference; "HOST" is
 * platforms/cuda/src/CudaVVKernels.cpp, "API" is openmmapi/src/VVIntegrator.cpp.
 *
 * Conventions.  Every function returns VVHIP_OK (0) or a negative error code; the text of the last
 * error of a plan is available from vvhip_last_error().  Errors that the reference raises as
 * OpenMMException keep the reference's message text.  Nothing here falls back to a CPU path.
 * All launches go to the hipStream_t given in vvhip_buffers.stream and never synchronise with the
 * host unless the function's comment says so.  A plan is not re-entrant (as the reference: one
 * integrator <-> one context, API:93-94).
 */
#ifndef VVHIP_H
#define VVHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VVHIP_VERSION 1
#define VVHIP_MAX_CHAINS 8
#define VVHIP_NUM_TG 3            /* TG_ATOM, TG_COM, TG_DRUDE (HOST:49) */

enum {
    VVHIP_OK = 0,
    VVHIP_ERR_INVALID = -1,       /* bad argument / API misuse */
    VVHIP_ERR_TOPOLOGY = -2,      /* the reference's OpenMMException cases (API:149,155; HOST:519,537,787,801) */
    VVHIP_ERR_UNSUPPORTED = -3,   /* valid for the reference, not implemented here (message says what) */
    VVHIP_ERR_HIP = -4,           /* a HIP runtime call failed */
    VVHIP_ERR_NO_DEVICE = -5,     /* no usable GPU: the product has no CPU path */
    VVHIP_ERR_EXCHANGE = -6,      /* multi-GPU: a mailbox wait on the peers timed out; the ranks have diverged, the run is void (sticky) */
    VVHIP_ERR_OVERFLOW = -7,      /* a fixed-point accumulator left its range (2KE > 1024 x the thermostat target); sticky */
    VVHIP_ERR_RENDEZVOUS = -8,    /* one-launch step: the blocks did not meet in the in-kernel rendezvous (not resident together); run void; sticky */
    VVHIP_ERR_CONSTRAINT = -9     /* in-kernel constraints: a cluster hit its iteration cap without converging; sticky */
};

/* OpenMM's CudaPrecision / HipPrecision property (examples/run-bulk.py:78) */
enum { VVHIP_SINGLE = 0, VVHIP_MIXED = 1, VVHIP_DOUBLE = 2 };

typedef struct vvhip_plan vvhip_plan;

/* What the reference's initialize() methods read from OpenMM::System, DrudeForce and VVIntegrator
 * (API:92-188, HOST:56-117, 462-667, 761-824, 878-902, 940-969, 998-1035). Host pointers, copied. */
typedef struct {
    int32_t num_atoms;
    int32_t padded_num_atoms;          /* cu.getPaddedNumAtoms(): stride of the planar force buffer */
    const double* masses;              /* System::getParticleMass            [num_atoms]            */
    const int32_t* mol_id;             /* ContextImpl::getMolecules() as particle -> molecule id    */
    int32_t num_molecules;
    int32_t num_drude_pairs;
    const int32_t* drude_pairs;        /* (drude, parent) = DrudeForce p, p1 (HOST:68-73)   [2*n]   */
    int32_t num_constraints;
    const int32_t* constraints;        /* System::getConstraintParameters, DOF accounting   [2*n]   */
    int32_t has_cm_motion_remover;     /* HOST:550-558                                              */
    int32_t num_particles_ld;
    const int32_t* particles_ld;       /* VVIntegrator::addParticleLangevin                         */
    int32_t num_image_pairs;
    const int32_t* image_pairs;        /* (image, parent) = VVIntegrator::addImagePair      [2*n]   */
    int32_t num_electrolyte;
    const int32_t* particles_electrolyte; /* VVIntegrator::addParticleElectrolyte                   */
    /* particle shard owned by this process: atoms [shard_begin, shard_end) of the system described
     * above; must not cut a molecule or a Drude pair.  Device arrays passed to vvhip_bind are indexed
     * from shard_begin.  0,0 = the whole system. */
    int32_t shard_begin, shard_end;
    /* Optional.  System::getConstraintParameters distances [num_constraints] (nm).  When given, the fused steps solve the constraints
     * inside kernels A and B: hydrogen-type clusters (one central particle + up to three peripheral particles of equal mass and
     * distance: what OpenMM's own SHAKE kernel takes, e.g. all X-H bonds), rigid three-site molecules (SETTLE), anything else as
     * general clusters (vvhip_plan_info.num_general_constraints), as long as every connected set of constraints fits one 64-lane wave.
     * NULL: constraints only enter the DOF count and the fused steps refuse to run if there are any (use the split entry points around
     * the host's solver). */
    const double* constraint_distances;
    /* Optional.  System::getVirtualSite for every massless site (OpenMM's TwoParticleAverageSite, ThreeParticleAverageSite,
     * OutOfPlaneSite, LocalCoordinatesSite with three parents -- what examples/ommhelper/oplspsffile.py:982-991 creates for lone pairs).
     * Kernel B then places the sites right after its position update, where the reference calls integration.computeVirtualSites()
     * (HOST:214, 374), and vvhip_plan_info.num_virtual_sites says so; the caller need not launch OpenMM's kernel.  A site is placed from
     * the lane of one of its parents and stored by index (x, y, z; the charge in posq.w stays), so sites cost no work items.
     *   virtual_sites       [5*n]  site particle, kind (VVHIP_VSITE_*), parents 1, 2, 3 (parent 3 = -1 for a two-particle average)
     *   virtual_site_params [12*n] AVERAGE2: w1 w2; AVERAGE3: w1 w2 w3; OUT_OF_PLANE: w12 w13 wCross;
     *                              LOCAL_COORDS: origin weights[3], x weights[3], y weights[3], local position[3] */
    int32_t num_virtual_sites;
    const int32_t* virtual_sites;
    const double* virtual_site_params;
} vvhip_system_desc;

enum { VVHIP_VSITE_AVERAGE2 = 0, VVHIP_VSITE_AVERAGE3 = 1, VVHIP_VSITE_OUT_OF_PLANE = 2, VVHIP_VSITE_LOCAL_COORDS = 3 };

/* VVIntegrator's parameters (openmmapi/include/openmm/VVIntegrator.h:70-431).  The reference reads
 * them through getters at every kernel call; here they are re-read whenever vvhip_set_params is
 * called.  Thermostat masses / DOF are fixed at plan creation, as in the reference (HOST:583-594). */
typedef struct {
    double temperature, frequency, drude_temperature, drude_frequency, step_size;
    int32_t num_nh_chains, loops_per_step;
    double max_drude_distance;
    double friction, drude_friction;
    double mirror_location;
    double electric_field;             /* kJ/(nm e) per particle, as VVIntegrator::setElectricField  */
    double cos_acceleration;
    int32_t use_com_temp_group, use_middle_scheme;
    int32_t auto_set_com_temp_group;   /* 1 = constructor default not overridden (API:67,106-121)    */
    int32_t auto_set_friction;
    double constraint_tolerance;       /* Integrator::getConstraintTolerance (relative, 1e-5 by default); used by the in-kernel SHAKE */
} vvhip_params;

/* Device arrays owned by the caller (OpenMM's HipContext / HipIntegrationUtilities in the plugin). */
typedef struct {
    void* velm;            /* mixed4 [n]: xyz = velocity, w = 1/mass       cu.getVelm()            */
    void* posq;            /* real4  [n]: xyz = position, w = charge       cu.getPosq()            */
    void* posq_correction; /* real4  [n], mixed mode only, else NULL       cu.getPosqCorrection()  */
    void* force;           /* int64  [3*padded]: planar x|y|z, x 2^32      cu.getForce()           */
    void* pos_delta;       /* mixed4 [n], only for the split (constraint) entry points; may be NULL */
    const void* random;    /* float4 [random_size] N(0,1)                  integration.getRandom() */
    uint32_t random_size;
    void* stream;          /* hipStream_t; NULL = the null stream                                  */
} vvhip_buffers;

/* Results of the host-side analysis (readable without a GPU). */
typedef struct {
    int32_t num_particles_nh, num_molecules_nh, num_normal_nh, num_pairs_nh;
    int32_t num_normal_ld, num_pairs_ld, num_images, num_electrolyte;
    int32_t num_temp_groups;                     /* HOST:567-573 */
    int32_t use_com_temp_group;                  /* after the auto rule, API:106-121 */
    double friction;                             /* after the auto rule */
    double dof[VVHIP_NUM_TG], nkbt[VVHIP_NUM_TG];
    double eta_mass[VVHIP_NUM_TG][VVHIP_MAX_CHAINS];
    double inv_mass_total;                       /* HOST:1028-1031 */
    int32_t num_waves, num_slots_used;           /* work-item layout: 64 slots per wave */
    int32_t max_cluster;                         /* largest set of particles that must share a wave */
    int32_t num_shake_clusters;                  /* hydrogen-type constraint clusters solved in-kernel (SHAKE; 0 if none / not possible) */
    int32_t constraints_fused;                   /* 1: no constraints, or all of them are handled in-kernel (hydrogen-type clusters, rigid three-site
                                                    molecules, or general clusters: num_general_constraints) => fused steps are valid */
    int32_t num_settle_clusters;                 /* rigid three-site molecules solved in-kernel (SETTLE) */
    int32_t periodic_layout;                     /* 1: the work-item layout is arithmetic (runs of identical molecules): the fused kernels compute
                                                    particle indices instead of loading them (vv_host.hpp: PeriodicLayout) */
    int32_t num_general_constraints;             /* constraints outside the two rules above (AllBonds, HAngles: chains, rings, triangles) that the
                                                    fused kernels relax by coloured Gauss-Seidel sweeps inside the wave of their molecule; 0 if the
                                                    System has none, or if a component of the constraint graph does not fit one wave */
    int32_t num_virtual_sites;                   /* virtual sites placed by kernel B itself (0: none given, or one of them cannot share a wave with
                                                    its parents -- then the caller runs its own computeVirtualSites after every step, as before) */
    double general_relaxation;                   /* relaxation factor of the general clusters' sweeps (1.2; 1.4 when constraints close triangles); 0 if none */
} vvhip_plan_info;

/* Nose-Hoover chain state + last reduction results (HOST: CudaVVKernels.h:206-215).  The reference
 * keeps these on the host and loses them on restart; here they live on the device and can be saved. */
typedef struct {
    double eta[VVHIP_NUM_TG][VVHIP_MAX_CHAINS];
    double eta_dot[VVHIP_NUM_TG][VVHIP_MAX_CHAINS + 1];
    double eta_dotdot[VVHIP_NUM_TG][VVHIP_MAX_CHAINS];
    double ke2[VVHIP_NUM_TG];                    /* last 2*KE per temperature group */
    double vscale[VVHIP_NUM_TG];                 /* last velocity scale factors     */
    double v_bias;                               /* last periodic velocity bias V (vMaxBuffer[0]) */
} vvhip_nh_state;

/* ---------------------------------------------------------------- life cycle */
/* Host-only: partitions particles (NH / Langevin / image), checks the reference's conflict rules,
 * counts DOF, sizes the thermostat chains and lays particles out in 64-lane waves so that every
 * Drude pair and (with the COM temperature group) every molecule sits inside one wave.
 * Replaces VVIntegrator::initialize's table building (API:123-155) and the initialize() methods of
 * all seven Cuda*Kernel classes.  Does not touch the GPU. */
int vvhip_plan_create(const vvhip_system_desc* system, const vvhip_params* params, int precision,
                      vvhip_plan** plan_out, char* errbuf, size_t errbuf_len);
void vvhip_plan_destroy(vvhip_plan* plan);
const char* vvhip_last_error(const vvhip_plan* plan);
int vvhip_plan_get_info(const vvhip_plan* plan, vvhip_plan_info* info);
/* Copies the wave layout: slots[2*i] = particle index (shard-relative, -1 = idle lane),
 * slots[2*i+1] = packed role word.  `capacity` in slots; returns the number of slots or <0. */
/* Why vvhip_plan_info.constraints_fused is 0 for this plan ("" when it is 1 or the System has no constraints): what of the constraint topology
 * did not fit the in-kernel solvers -- a component with more than the 64 constraints of a wave's list, a particle in more constraints than the 16
 * colours of the sweeps, a constraint across two waves.  The fused steps then refuse (VVHIP_ERR_UNSUPPORTED, with this text) and the split
 * entry points leave the gaps for the host's own solver. */
const char* vvhip_plan_unfused_reason(const vvhip_plan* plan);
int vvhip_plan_get_slots(const vvhip_plan* plan, int32_t* slots, int32_t capacity);

/* Binds device arrays and allocates the plan's own device state (forceExtra, oldDelta, accumulators,
 * chain state, layout tables) on the current device.  First call that needs a GPU. */
int vvhip_bind(vvhip_plan* plan, const vvhip_buffers* buffers);
int vvhip_set_params(vvhip_plan* plan, const vvhip_params* params);
int vvhip_set_box(vvhip_plan* plan, const double box[3]);      /* cu.getPeriodicBoxSize() (HOST:1057,1129) */
/* The plan caches, per work-item, the mass RECIP(velm.w) and the Drude-pair mass fractions (filled on the device from velm.w in
 * front of the first launch that needs them; the reference recomputes them in every kernel, K/drudeNoseHoover.cu:173-180).  A host
 * that rewrites the inverse masses in velm.w (OpenMM does so only in Context::reinitialize, which rebuilds the kernels anyway)
 * calls this; binding a different velm array does it implicitly. */
int vvhip_masses_changed(vvhip_plan* plan);
/* Blocking copies of the thermostat state (checkpoint / tests). */
int vvhip_get_nh_state(vvhip_plan* plan, vvhip_nh_state* out);
int vvhip_set_nh_state(vvhip_plan* plan, const vvhip_nh_state* in);

/* ---------------------------------------------------------------- fused path
 * One whole VVIntegrator step between two force evaluations.  Constraints are solved inside the two kernels when
 * vvhip_plan_info.constraints_fused says so (otherwise the fused steps refuse to run and the split entry points below go around the
 * host's solver); virtual sites are placed by kernel B when described (vvhip_system_desc.virtual_sites), otherwise the caller's own
 * computeVirtualSites follows the step as it follows the reference's.  Forces for the step must already be in `force`.
 *
 * Middle scheme (API:232-270 after calcForcesAndEnergy): 2 launches
 *   pass A: extra forces (Langevin, E-field, cos) + full kick (+ velocity constraints) + molecular COM + per-group
 *           2KE -> fixed-point accumulators; with cos acceleration also the bias moment and the group sums as
 *           moments of the still biased velocities, so that no launch is needed between bias and 2KE
 *   pass B: NH chain (device) + velocity scaling + bias remove/restore + both half drifts (+ position
 *           constraints) + hard wall + image mirror
 * Only molecules larger than one 64-lane wave, chains longer than 4 and systems beyond ~2.6 M particles take more
 * launches (per-molecule COM accumulation, stand-alone chain kernel, the three-launch cos sequence).
 * Classic scheme: vvhip_step_vv_first() = API:295-310, vvhip_step_vv_second() = API:316-336.
 * `random_index` = integration.prepareRandomNumbers(...) for this step (HOST:863), ignored unless
 * Langevin particles exist. */
int vvhip_step_middle(vvhip_plan* plan, uint32_t random_index);
int vvhip_step_vv_first(vvhip_plan* plan);
int vvhip_step_vv_second(vvhip_plan* plan, uint32_t random_index);
/* vvhip_step_middle cut at its global reductions, for hosts that shard particles over GPUs and run a
 * collective in between.  A step has vvhip_step_middle_phases() phases (1 without NH particles, else 2; 3 only
 * where the cos perturbation cannot use its moment form, see above).  After every phase but the last, the host sums
 * the accumulator range reported by vvhip_accumulators(phase) element-wise over ranks (ncclSum on
 * int64; fixed point makes the result independent of rank order) on the plan's stream:
 *     for (ph = 0; ph < P; ph++) { vvhip_step_middle_phase(plan, ph, ri); if (ph < P-1) all_reduce(acc(ph)); }
 * That all-reduce (64 int64 slots per reduced quantity, <= 1.5 KB, latency-bound) is the only cross-GPU
 * exchange per thermostat application. */
int vvhip_step_middle_phases(const vvhip_plan* plan);
/* Algorithmic bytes per particle moved by kernel A / kernel B of this plan's fused middle step (particle arrays + 6 bytes of index per
 * pass, SURVEY section 8d's accounting).  Plans without extra forces and in-kernel constraints keep the kicked velocities in registers in
 * kernel A and repeat the kick in kernel B: 62 + 158 bytes in mixed precision instead of 94 + 134. */
int vvhip_algorithmic_bytes(const vvhip_plan* plan, int32_t* bytes_a, int32_t* bytes_b);
int vvhip_step_middle_phase(vvhip_plan* plan, int phase, uint32_t random_index);
int vvhip_accumulators(vvhip_plan* plan, int phase, void** device_ptr, int32_t* count);

/* Built-in exchange.  Instead of running the collective itself, a host may hand the plan an RCCL communicator:
 * rank 0 calls vvhip_comm_unique_id(), ships the 128 bytes to the other ranks by any means (bench.py uses
 * torch.distributed), every rank calls vvhip_comm_init(); from then on vvhip_step_middle / vvhip_run_graph /
 * vvhip_run_eager issue ncclAllReduce(int64, sum) on the plan's stream between the phases themselves, so a whole
 * sharded step (or a captured graph of steps) needs no host involvement.  librccl is resolved at run time. */
int vvhip_comm_unique_id(void* id128);
int vvhip_comm_init(vvhip_plan* plan, const void* id128, int nranks, int rank);
int vvhip_comm_destroy(vvhip_plan* plan);
/* Ranks of the plan's communicator as RCCL itself reports them (ncclCommCount; 0 without a communicator): what bench.py prints as
 * config.exchange.rccl_ranks, so that a scaling line says how many ranks the collective really spanned. */
int vvhip_comm_count(vvhip_plan* plan, int32_t* ranks);
/* hipDeviceCanAccessPeer(device, peer_device): whether the mailbox's hipIpc mappings of a peer's box can be direct (xGMI / PCIe P2P). */
int vvhip_peer_access(int device, int peer_device, int32_t* can_access);

/* Exchange without a collective launch, for the ranks of ONE node ("mailbox" over xGMI peer mappings).  Every rank calls
 * vvhip_mailbox_create (allocates its uncached box, returns a 64-byte hipIpc handle), the host gathers all handles in rank
 * order (ranks * 64 bytes, any transport) and every rank calls vvhip_mailbox_connect.  From then on the thermostat wave of block 0
 * of kernel B -- which has just folded the rank's accumulators, complete since kernel A ended -- stores the rank's int64 totals into
 * every peer's box (one 8-byte {sequence, payload} word each) and the thermostat wave of every block polls the rank's own box and
 * adds the peers' words: a sharded step stays two launches, replayable from a hipGraph, and all ranks continue with identical
 * bits.  Carries the three kinetic-energy totals, and with the cos perturbation in its moment form also the bias moment and the
 * six group moments (10 totals = 20 words); chain lengths <= 4 (vvhip_mailbox_status: active); everything else keeps using the
 * RCCL communicator if one is set.  A rank that does not hear from its peers within ~5 s raises a sticky flag and carries on
 * with incomplete sums instead of hanging the GPU; the flag surfaces as VVHIP_ERR_EXCHANGE from vvhip_synchronize and from the
 * next vvhip_run_graph / vvhip_run_eager (see vvhip_status), so a host can never take a figure from a diverged run.
 * No reference counterpart (the reference is single-GPU: CudaVVKernelFactory.cpp:68). */
int vvhip_mailbox_create(vvhip_plan* plan, int nranks, int rank, void* handle64);
int vvhip_mailbox_connect(vvhip_plan* plan, const void* handles);
int vvhip_mailbox_status(vvhip_plan* plan, int32_t* active, int32_t* timed_out);
int vvhip_mailbox_destroy(vvhip_plan* plan);
/* After vvhip_mailbox_connect: *shared_device = 1 if a peer's box lives on this rank's own device (several ranks on one GPU: test
 * set-ups) -- kernel B then keeps the explicit work-item layout next to the exchange; *arithmetic_layout = 1 if kernel B of this plan
 * takes the arithmetic layout together with the mailbox exchange. */
int vvhip_mailbox_layout(vvhip_plan* plan, int32_t* shared_device, int32_t* arithmetic_layout);

/* ---------------------------------------------------------------- kernel-interface level
 * One entry per KernelImpl virtual, for use inside OpenMM where constraint solvers run between them. */
/* IntegrateMiddleStepKernel (VVKernels.h:50-86; HOST:119-235) */
int vvhip_reset_extra_force(vvhip_plan* plan);                 /* resetExtraForce                      */
int vvhip_middle_kick(vvhip_plan* plan);                       /* firstIntegrate, before applyVelocityConstraints (HOST:144-148) */
int vvhip_middle_half_drift1(vvhip_plan* plan);                /* firstIntegrate, after  (HOST:154-158) */
int vvhip_middle_half_drift2(vvhip_plan* plan);                /* secondIntegrate, before applyConstraints (HOST:169-173) */
int vvhip_middle_finish(vvhip_plan* plan);                     /* secondIntegrate, after: Pos3 + hard wall (HOST:179-212) */
/* IntegrateVVStepKernel (VVKernels.h:94-130; HOST:296-442) */
int vvhip_vv_half_kick(vvhip_plan* plan, int update_pos_delta);/* HOST:341-348 / 417-424 */
int vvhip_vv_positions(vvhip_plan* plan);                      /* HOST:355-372: positions + hard wall */
/* ModifyDrudeNoseKernel (VVKernels.h:138-157; HOST:670-754): 2 launches (sums | chain + scaling), no host round trip */
int vvhip_scale_velocity(vvhip_plan* plan);
/* ModifyDrudeLangevinKernel (VVKernels.h:165-184; HOST:826-872) */
int vvhip_apply_langevin_force(vvhip_plan* plan, uint32_t random_index);
/* ModifyImageChargeKernel (VVKernels.h:192-211; HOST:904-934) */
int vvhip_update_image_positions(vvhip_plan* plan);
/* ModifyElectricFieldKernel (VVKernels.h:219-238; HOST:971-992) */
int vvhip_apply_electric_force(vvhip_plan* plan);
/* ModifyCosineAccelerateKernel (VVKernels.h:246-269; HOST:1037-1134) */
int vvhip_apply_cosine_force(vvhip_plan* plan);
int vvhip_calc_velocity_bias(vvhip_plan* plan);
int vvhip_remove_velocity_bias(vvhip_plan* plan);
int vvhip_restore_velocity_bias(vvhip_plan* plan);
int vvhip_calc_viscosity(vvhip_plan* plan, double* v_max, double* inv_viscosity);   /* blocks; 8-byte copy */
/* 1/2 sum m v^2 over all massive particles of this shard (blocks; the reference forwards computeKineticEnergy to OpenMM's
 * integration utilities, CudaVVKernels.cpp:233-235 -- a stand-alone host has no such service). kJ/mol. */
int vvhip_compute_kinetic_energy(vvhip_plan* plan, double* kinetic_energy);
/* Device pointer of the plan-owned forceExtra array (real3[n]); getForceExtra() of the reference
 * (CudaVVKernels.h:86-88). */
int vvhip_force_extra(vvhip_plan* plan, void** device_ptr);

/* ---------------------------------------------------------------- stand-alone host support
 * NOT part of the reference path: lets a host without OpenMM (tests, bench.py) own device memory and
 * produce forces.  vvhip_synth_tether_force writes F = -k_t (x - site) on every massive particle plus a
 * Drude-parent spring -k_d (x_d - x_p), in OpenMM's fixed-point planar layout. */
int vvhip_device_count(int* count);
int vvhip_set_device(int device);
int vvhip_malloc(void** ptr, size_t bytes);
int vvhip_free(void* ptr);
int vvhip_memcpy_h2d(void* dst, const void* src, size_t bytes);
int vvhip_memcpy_d2h(void* dst, const void* src, size_t bytes);
int vvhip_memset(void* dst, int value, size_t bytes);
int vvhip_synchronize(vvhip_plan* plan);         /* hipStreamSynchronize + the sticky health check below */
/* Sticky health flags the kernels raise in pinned host memory (read without synchronising): a mailbox wait on the peers ran out
 * (VVHIP_ERR_EXCHANGE), a fixed-point accumulator left its range (VVHIP_ERR_OVERFLOW).  vvhip_run_graph / vvhip_run_eager refuse
 * to start and vvhip_synchronize returns the error once a flag is up; vvhip_status_clear resets them (after the host has
 * restored a valid state). */
int vvhip_status(vvhip_plan* plan, int32_t* mailbox_timed_out, int32_t* accumulator_overflow);
/* All four words: [0] mailbox wait ran out, [1] accumulator overflow, [2] the one-launch step's in-kernel rendezvous ran out
 * (VVHIP_ERR_RENDEZVOUS), [3] an in-kernel constraint cluster reached its iteration cap unconverged (VVHIP_ERR_CONSTRAINT). */
int vvhip_status_words(vvhip_plan* plan, int32_t words[4]);
int vvhip_status_clear(vvhip_plan* plan);
/* A missed rendezvous of the one-launch step ([2]: its blocks were not resident together within 0.2 s -- another process's kernels on the
 * device) is the one failure the plan-driven loops repair themselves.  vvhip_run_graph / vvhip_run_eager calls of >= 64 steps (test hook
 * "recover_min_steps") start from a device-side snapshot of the physical state (positions, correction, velocities, forces, extra forces,
 * both thermostat copies, the random generator's state), kept with the list of run calls since until a vvhip_synchronize has seen them end
 * well; the vvhip_synchronize that finds [2] raised instead puts the snapshot back, pins the plan to two launches per step (bit for bit
 * the same step; vvhip_fused_status: active = 0), repeats the calls, prints one line on stderr and returns VVHIP_OK -- the trajectory is
 * the one an undisturbed run gives.  Anything else that touches the plan in between (a split entry point, vvhip_set_params, ...) settles
 * the pending calls first.  Not in multi-GPU runs (every rank would have to repeat).  Without a snapshot (short calls, vvhip_step_* driven
 * by the host, "recover" = 0 / VVHIP_RECOVER=0) the failure is VVHIP_ERR_RENDEZVOUS as before, and the plan is pinned to two launches
 * all the same.  *recoveries: how often it has happened to this plan.  Reference contract: one context per device,
 * platforms/cuda/src/CudaVVKernelFactory.cpp:68. */
int vvhip_recovery_count(vvhip_plan* plan, int64_t* recoveries);
/* The middle scheme's step as ONE launch (kernels A and B around an in-kernel rendezvous of co-resident blocks): *active = 1 if
 * vvhip_step_middle takes it for this plan as it stands (a thermostat with <= 4 links, every tile a wave of its own on <= 256
 * co-resident blocks, no RCCL exchange between the halves, a kernel for the plan's pair of stage sets), *launches = fused launches
 * so far (captured ones count once), *wait_units = where the self-tuning wait between a block's publish and its first poll round
 * stands (units of 256 shader clocks; reading it synchronises).  vvhip_debug_tune(plan, "fused", 0) forces the two-launch step
 * (bit-identical results); "fused_poll_delay" >= 0 pins the wait.  Any of the three pointers may be NULL.
 * The classic scheme's two thermostat applications per step (vvhip_step_vv_first / _second) and vvhip_scale_velocity take the same kernel
 * under the same conditions -- sums, rendezvous, chain, scaling (+ half kick and drift / half kick in front) in one launch each, two per
 * classic step instead of four; they count in *launches, *active speaks of the middle scheme only. */
int vvhip_fused_status(vvhip_plan* plan, int32_t* active, int64_t* launches, int32_t* wait_units);
int vvhip_stream_create(void** stream);          /* hipStreamCreateWithFlags(non-blocking) */
int vvhip_stream_destroy(void* stream);
int vvhip_synth_tether_force(vvhip_plan* plan, const void* site /* real4[n] */, double k_tether, double k_drude);
/* Replays hipGraphs of `steps_per_graph` whole steps (optionally with the synthetic force kernel in front of each; both
 * schemes): floor(nsteps / steps_per_graph) replays, the remainder enqueued step by step; returns after enqueueing.  The
 * graph of the current thermostat parity is captured on first use (see vvhip_graph_prepare to do that up front). */
int vvhip_run_graph(vvhip_plan* plan, int nsteps, int steps_per_graph, const void* site, double k_tether, double k_drude);
/* Captures, instantiates and uploads that graph for BOTH thermostat parities WITHOUT launching anything (no-op for a parity
 * whose executable is already there).  The plan keeps one executable per parity, so a host that calls this after its warm-up
 * never pays a capture inside a timed region, whatever mix of replays and odd tails follows; vvhip_run_graph replays floor(nsteps / steps_per_graph) times and enqueues the remainder step by step. */
int vvhip_graph_prepare(vvhip_plan* plan, int steps_per_graph, const void* site, double k_tether, double k_drude);
/* Device Gaussian generator for stand-alone hosts (inside OpenMM the buffer and its refills are OpenMM's): Philox4x32-10 +
 * Box-Muller.  vvhip_fill_random refills the bound random buffer; the vvhip_run_* loops refill it themselves whenever a step's
 * slice (max(normalLD,1) + 2 max(pairsLD,1) float4, HOST:806-807,863) no longer fits, and at the start of every captured graph. */
int vvhip_set_random_seed(vvhip_plan* plan, uint64_t seed);
int vvhip_fill_random(vvhip_plan* plan);
/* The same steps enqueued one by one from C (no graph): fallback when graph capture is not wanted. */
int vvhip_run_eager(vvhip_plan* plan, int nsteps, const void* site, double k_tether, double k_drude);
/* ... and through the per-KernelImpl entry points in VVIntegrator::stepMiddle's order: the launches of the un-fused drop-in path
 * (what a host with its own constraint solver between the stages pays, the solver's launches excluded). */
int vvhip_run_eager_unfused(vvhip_plan* plan, int nsteps, const void* site, double k_tether, double k_drude);
/* HIP-event timing of the dominant kernels on the plan's stream, for bench.py's roofline block. */
/* Average duration of `reps` back-to-back launches of one stage kernel (0 = A, 1 = B) with the given stage bits,
 * bracketed by two HIP events on the plan's stream.  Destroys the physical state (timing only). */
int vvhip_time_kernel(vvhip_plan* plan, int kernel, uint32_t flags, int reps, double* ms_per_launch);
/* Tracing: roctx ranges around every launch group (visible with rocprofv3 --marker-trace); the OpenMM adapter switches it on with
 * VVIntegrator::setDebugEnabled and prints the reference's per-call lines itself (VVIntegrator.h:417-419, CudaVVKernels.cpp:57,120,...).
 * Also VVHIP_ROCTX=1 in the environment.  libroctx64 is resolved at run time. */
int vvhip_set_trace(vvhip_plan* plan, int enable);
/* Every stage set a plan launches runs a kernel compiled for exactly its stage bits.  The reference compiles its kernels per System at
 * run time (CudaVVKernels.cpp:98-101, 639-647); here the stage sets of the BASELINE configurations (+- constraints, classic scheme,
 * sharded, large boxes) are compiled into the library (vv_kernels.hip: SF_*), and a stage set outside that list -- or a thermostat chain
 * of 1, 2 or 4 links instead of the integrator's default 3 -- is compiled by hipRTC from the same source with the same options the first
 * time a plan launches it (csrc/vv_rtc.cpp; ~1 s per kernel, once per process; VVHIP_RTC below).  Only if that is switched off or fails
 * (libhiprtc.so missing; one line on stderr) does the generic kernel with run-time stage bits run, 15-20 % slower:
 * counts[0 / 1] = enqueued launches of kernel A / B of this plan that did (a launch captured into a graph counts once), stage_sets = the
 * last stage bits that did.  tests/test_gpu_specialised.py asserts 0 for every BASELINE configuration +- constraints, classic scheme,
 * sharded, and for stage sets outside the compiled list.  VVHIP_WARN_GENERIC=1 prints them as they happen. */
int vvhip_generic_launches(vvhip_plan* plan, int64_t counts[2], uint32_t stage_sets[2]);
/* Process-wide: counts[0] = kernels compiled at run time so far, counts[1 / 2] = enqueued launches of kernel A / B that ran one;
 * compile_seconds (optional) = time spent in the compiler. */
int vvhip_rtc_stats(int64_t counts[3], double* compile_seconds);
/* Process-wide: stage sets whose run-time compilation FAILED (no compiler library, a compile error); their launches run the generic
 * kernel (15-20 % slower) for the rest of the process.  One line on stderr each; hosts that report performance should look here. */
int vvhip_rtc_failures(int64_t* failed);
/* Sets VVHIP_RTC's value for the launches that follow (process-wide; graphs captured earlier keep their kernels) and returns the previous
 * one; mode < 0 only returns it. */
int vvhip_rtc_mode(int mode);
/* Per-launch timing of eager (not captured) launches, summed per class until vvhip_timing_read.  Kernels A and B: the dispatch's own
 * begin / end timestamps (hipExtLaunchKernel with start / stop events: nothing is added to the stream; what rocprofv3's kernel trace
 * reports).  enable = 1: everything else ("other": force provider, chain launch, collectives) bracketed by recorded events as well;
 * enable = 2: kernels A and B only, so that the stream holds exactly what an untimed run enqueues; enable = n > 2: as 2, with n events
 * created now instead of during the timed launches.  0 switches it off. */
int vvhip_timing_enable(vvhip_plan* plan, int enable);
int vvhip_timing_read(vvhip_plan* plan, double* ms_pass_a, double* ms_pass_b, double* ms_other, int32_t launches[3]);

/* ---------------------------------------------------------------- environment (read once, at vvhip_plan_create)
 * Behaviour switches, all optional.  (The tuning switches of rounds 1-3 -- launch shape, mass tables, velocity round trip, cos moments --
 * are closed experiments, TUNING_LOG.md; what tests still need of them is the per-plan hook vvhip_debug_tune below.)
 *   VVHIP_PERIODIC=1|0        arithmetic work-item layout (vv_host.hpp: PeriodicLayout) always / never; default: from 1.1 M lanes, when the
 *                             system is runs of identical molecules (vvhip_plan_info.periodic_layout tells)
 *   VVHIP_PERIODIC_DEBUG=1    the decomposition into regions and why the layout was (not) enabled, on stderr
 *   VVHIP_FUSED=0             the middle scheme's step (and each thermostat application of the classic scheme) as TWO launches also where the one-launch step (kernels A and B around an in-kernel
 *                             rendezvous of co-resident blocks; vvhip_fused_status) would be taken -- for a GPU that other processes compute
 *                             on at the same time: blocks that are not resident together meet the rendezvous' 0.2 s bound
 *                             (VVHIP_ERR_RENDEZVOUS).  Same results bit for bit
 *   VVHIP_SHAKE_MODE=0        hydrogen-type constraint clusters by Gauss-Seidel sweeps of the central lane (OpenMM's iteration; generic
 *                             kernels) instead of the direct velocity solve / coupled Newton iteration of all lanes of a cluster
 *   VVHIP_ROCTX=1             roctx ranges (see vvhip_set_trace)
 *   VVHIP_RTC=0|1|2           run-time compilation of kernels A / B (vvhip_generic_launches): never / for stage sets without a compiled
 *                             kernel (default) / for every launch (tests: the run-time kernel against the compiled one, bit for bit)
 *   VVHIP_STALL=us[:period]   race detection by timing: every period-th launch of a plan is preceded by a host sleep of `us` microseconds (the GPU
 *                             drains; whatever was only ordered by the depth of the queue lands differently); tests/test_gpu_stalls.py
 *   VVHIP_RTC_VERBOSE=1       one line on stderr per kernel compiled at run time;  VVHIP_RTC_DENY="A:0x441,B:*": these stage sets stay on the
 *                             generic kernel (bisecting)
 *   VVHIP_WARN_GENERIC=1      one line on stderr per stage set that runs on the generic kernel (15-20 % slower)
 * and in the OpenMM adapters (platforms/hip): VVHIP_PLUGIN_DEFER=0 -- run every KernelImpl call as its own launch(es) instead of answering a
 * completed stage-by-stage sequence with the fused step. */

/* ---------------------------------------------------------------- test hooks
 * Used by tests/ to drive single stages against the oracle; not needed by an integrating host.
 * kernel: 0 = A (produce), 1 = B (consume), 2 = chain; flags are the stage bits of csrc/vv_args.hpp. */
int vvhip_debug_launch(vvhip_plan* plan, int kernel, uint32_t flags, uint32_t random_index);
/* One of the plan's tuning choices by name (call between vvhip_plan_create and vvhip_bind; later calls drop the captured graphs):
 * "grid_cap_a" / "grid_cap_b" (most blocks per launch), "block_threads", "split_chain_waves" (the chain becomes its own launch from n waves
 * on), "periodic_kernels" / "periodic_a" (0: load slot words although the layout is arithmetic), "rekick", "no_moments", "mass_tab_a" /
 * "periodic_b", "mass_tab_b", "acc_store", "fused" (0: the middle scheme's step as two launches also where one would do), "fused_poll_delay" (>= 0 pins the wait in front of the rendezvous' first poll round, units of 256 clocks; -1: self-tuning), "fused_late_shift", "gc_omega_permille" (relaxation factor of the general clusters' sweeps x 1000, for rate scans).  Tests
 * use it to run large-system code paths at small sizes. */
int vvhip_debug_tune(vvhip_plan* plan, const char* key, int value);
int vvhip_debug_read_accumulators(vvhip_plan* plan, double out[4], int zero_after);  /* blocks */
int vvhip_debug_set_scales(vvhip_plan* plan, const double scales[4]);                /* vscale[3], bias V; blocks */
int vvhip_debug_timestamps(vvhip_plan* plan, uint32_t flags, int block, long long out[128]);  /* instrumented builds only (tools/probes) */
int vvhip_debug_timestamps_fused(vvhip_plan* plan, int block, long long out[128]);            /* ... one real one-launch step, stamped */
int vvhip_debug_span(vvhip_plan* plan, int kernel, uint32_t flags, int reps, double out[8]);   /* instrumented builds only */
int vvhip_debug_fused_flags(vvhip_plan* plan, int kernel, uint32_t* flags);          /* stage bits of the fused middle step's kernel A (0) / B (1) */
/* The launch shape the plan chose (host-only: before vvhip_bind a whole MI355X of 256 CUs is assumed): shape[0] = threads of the tile waves of a
 * block (the one-launch step and kernel B add the block's thermostat wave), shape[1] / shape[2] = most blocks per launch of kernel A / kernel B,
 * shape[3] = tile waves per block of the one-launch step, 0 where the launch shape rules it out (its other conditions: vvhip_fused_status). */
int vvhip_debug_launch_shape(const vvhip_plan* plan, int32_t shape[4]);
int vvhip_debug_step_spans(vvhip_plan* plan, int nsteps, const void* site, double k_tether, double k_drude, double out[36]);   /* instrumented builds only */
int vvhip_debug_old_delta(vvhip_plan* plan, void** device_ptr);                       /* plan-owned oldDelta (mixed4[n]) */

#ifdef __cplusplus
}
#endif
#endif /* VVHIP_H */
