/*
 * oracle/vv_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the per-step velocity-Verlet / Nose-Hoover /
 * TGNH / Langevin / cos-acceleration / image-charge path of
 * z-gong/openmm-velocityVerlet.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product (libvvhip) never does.
 *
 * Pinning: every kernel-level function here is checked bit-for-bit against the
 * reference's own kernel sources compiled in place by `make ref`
 * (oracle/_ref/libvvref_*.so; see ref_prelude.h for how and under which stated
 * assumptions), by tests/test_oracle_vs_ref.py in this container, and against the
 * golden vectors those runs produced (tests/golden/*.npz) everywhere else.
 * vvo_propagate_nh_chain is checked bit-for-bit against the reference's own
 * VVIntegrator::propagateNHChain (openmmapi/src/VVIntegrator.cpp:340-376, compiled in
 * place by `make refapi` against the stand-in OpenMM headers of compat/:
 * oracle/_ref/libvvref_api.so) through tests/golden/refapi_chain.npz and live in this
 * container (tests/test_ref_api.py).  The whole step (vvo_step: sequencing, host constants, DOF accounting)
 * is checked bit-for-bit against the reference's WHOLE pipeline -- its VVIntegrator.cpp, CudaVVKernels.cpp,
 * CudaVVKernelFactory.cpp and kernels compiled in place on stand-in CUDA-platform classes (`make refhost`,
 * oracle/_ref/libvvref_host_*.so; tests/test_ref_host.py, goldens tests/golden/refhost_*) -- for twelve small
 * configurations in three precisions and for every BASELINE configuration at full size.  All reference builds need a stand-in for
 * something OpenMM supplies (the JIT prelude; the headers; the CUDA platform classes), so by the tier's rule the
 * oracle stays "parity unpinned" formally -- see DESIGN.md section 2 for what executed
 * reference code stands behind which statement.  vvo_tether_force is our own
 * synthetic force provider; the SHAKE / SETTLE statements follow OpenMM's published
 * algorithm (its source is not under /root/reference).
 *
 * Build once per precision mode (oracle/Makefile):
 *   VVO_SINGLE real=float  mixed=float
 *   VVO_MIXED  real=float  mixed=double   (reference default, examples/run-bulk.py:78)
 *   VVO_DOUBLE real=double mixed=double
 */
#ifndef VV_ORACLE_H
#define VV_ORACLE_H

#if defined(VVO_DOUBLE)
typedef double vvo_real;  typedef double vvo_mixed;
#elif defined(VVO_MIXED)
typedef float  vvo_real;  typedef double vvo_mixed;
#elif defined(VVO_SINGLE)
typedef float  vvo_real;  typedef float  vvo_mixed;
#else
#error "define one of VVO_SINGLE / VVO_MIXED / VVO_DOUBLE"
#endif

typedef struct { vvo_real x, y, z, w; } vvo_real4;
typedef struct { vvo_real x, y, z; } vvo_real3;
typedef struct { vvo_mixed x, y, z, w; } vvo_mixed4;
typedef struct { float x, y, z, w; } vvo_float4;
typedef struct { int x, y; } vvo_int2;

#define VVO_MAX_CHAINS 8
#define VVO_NUM_TG_MAX 3

/* One simulated system + integrator configuration.  All arrays are caller-owned. */
typedef struct {
    /* sizes */
    int num_atoms, padded_num_atoms, num_molecules;
    /* OpenMM-owned state (CudaVVKernels.cpp:144-147,179-184) */
    vvo_mixed4* velm;          /* xyz = v, w = 1/m            [num_atoms]          */
    vvo_real4*  posq;          /* xyz = x, w = charge         [num_atoms]          */
    vvo_real4*  posq_corr;     /* mixed mode only, else NULL  [num_atoms]          */
    long long*  force;         /* planar x|y|z, fixed point x 2^32 [3*padded]      */
    vvo_mixed4* pos_delta;     /*                             [num_atoms]          */
    /* plugin-owned state */
    vvo_real3*  force_extra;   /* [num_atoms]  */
    vvo_mixed4* old_delta;     /* [num_atoms]  */
    vvo_mixed4* com_velm;      /* [num_molecules] */
    vvo_mixed*  v_buffer;      /* [num_atoms] cos-acceleration bias buffer */
    /* Drude pairs for the hard wall: x = Drude, y = parent (CudaVVKernels.cpp:68-73) */
    int num_drude_pairs;  const vvo_int2* drude_pairs;
    /* NH tables (CudaVVKernels.cpp:483-529) */
    int num_particles_nh; const int* particles_nh;
    int num_molecules_nh; const int* molecules_nh;
    int num_normal_nh;    const int* normal_nh;
    int num_pairs_nh;     const vvo_int2* pairs_nh;
    const int* particle_mol_id;            /* [num_atoms]     */
    const vvo_int2* particles_in_molecules;/* (count,start) [num_molecules] */
    const int* particles_sorted_by_mol_id; /* [num_atoms]     */
    int num_tg;                            /* 1..3 (CudaVVKernels.cpp:567-573) */
    int use_com_tg;
    /* NH chain state, double on the host in the reference (CudaVVKernels.h:206-207) */
    int num_chains, loops_per_step;
    double eta[VVO_NUM_TG_MAX][VVO_MAX_CHAINS];
    double eta_dot[VVO_NUM_TG_MAX][VVO_MAX_CHAINS + 1];
    double eta_dotdot[VVO_NUM_TG_MAX][VVO_MAX_CHAINS];
    double eta_mass[VVO_NUM_TG_MAX][VVO_MAX_CHAINS];
    double tg_nkbt[VVO_NUM_TG_MAX];
    double ke2[VVO_NUM_TG_MAX];            /* out: last 2*KE per group   */
    double vscale[VVO_NUM_TG_MAX];         /* out: last scale factors    */
    /* Langevin subset (CudaVVKernels.cpp:775-804) */
    int num_particles_ld;
    int num_normal_ld;    const int* normal_ld;
    int num_pairs_ld;     const vvo_int2* pairs_ld;
    const vvo_float4* random; unsigned int random_size, random_index;
    /* image charges / E-field */
    int num_images;       const vvo_int2* image_pairs;   /* x = image, y = parent */
    int num_electrolyte;  const int* particles_electrolyte;
    /* integrator parameters (VVIntegrator.h) */
    double dt, temperature, drude_temperature;
    double friction, drude_friction, max_drude_distance;
    double mirror, efield, cos_accel;
    double box[3];
    double inv_mass_total;
    int use_middle;
    /* synthetic force provider (ours; 0 = forces are whatever the caller left in `force`) */
    int force_mode;                        /* 1 = tether */
    const vvo_real4* site;                 /* tether anchor positions [num_atoms] */
    double k_tether, k_drude;
    /* classic-VV bookkeeping (VVIntegrator.cpp:286-292) */
    int forces_valid;
    int num_threads;                       /* OpenMP threads for the step drivers */
    /* constraints solved on the path: OpenMM-style SHAKE clusters, see vvo_shake_positions */
    int num_shake;  const int* shake_atoms;   /* [4*n]: central, up to three peripherals (-1 = none) */
    const float* shake_params;                /* [4*n]: 1/m_c, 0.5/(1/m_c + 1/m_p), d^2, 1/m_p       */
    double constraint_tolerance;
    /* rigid three-site molecules (SETTLE), see vvo_settle_positions */
    int num_settle; const int* settle_atoms;  /* [3*n]: apex, partner, partner */
    const float* settle_params;               /* [2*n]: apex-partner distance, partner-partner distance */
    int shake_mode;                           /* 0: Gauss-Seidel sweeps over the cluster (OpenMM's iteration); 1: the cluster's constraints
                                                 at once (direct solve for velocities, coupled Newton for positions), see vvo_cluster_* */
    /* general constraint clusters (any topology: AllBonds, HAngles), see vvo_general_positions: constraints SORTED BY COLOUR, two
     * constraints of one colour never share a particle */
    int num_general; const int* general_atoms;     /* [2*n]: a, b */
    const float* general_params;                   /* [4*n]: d^2, 0.5/(1/m_a + 1/m_b), 1/m_a, 1/m_b */
    double general_omega;                          /* relaxation factor of those sweeps: every update is taken omega times (1.2; 1.4 when the
                                                      constraints close triangles: csrc/vv_layout.h GC_OMEGA_*, the product's rule) */
    /* virtual sites, see vvo_compute_virtual_sites */
    int num_vsites; const int* vsite_atoms;        /* [5*n]: site, kind (0 average of two, 1 average of three, 2 out of plane, 3 local
                                                      coordinates), parents 1, 2, 3 */
    const double* vsite_params;                    /* [12*n]: weights / local position, include/vvhip.h: virtual_site_params */
} vvo_system;

#ifdef __cplusplus
extern "C" {
#endif
int vvo_sizeof_real(void);
int vvo_sizeof_mixed(void);
int vvo_sizeof_system(void);

/* ---- kernel-level restatements (one per reference __global__) ---- */
void vvo_integrate_middle_vel(int n, int padded, vvo_mixed4* velm, const long long* force,
                              const vvo_real3* force_extra, vvo_mixed dt);
void vvo_integrate_middle_pos1(int n, const vvo_mixed4* velm, vvo_mixed4* pos_delta, vvo_mixed4* old_delta, vvo_mixed dt);
void vvo_integrate_middle_pos2(int n, const vvo_mixed4* velm, vvo_mixed4* pos_delta, vvo_mixed4* old_delta, vvo_mixed dt);
void vvo_integrate_middle_pos3(int n, vvo_real4* posq, vvo_real4* posq_corr, const vvo_mixed4* pos_delta,
                               const vvo_mixed4* old_delta, vvo_mixed4* velm, vvo_mixed dt);
void vvo_apply_hard_wall(int npairs, vvo_real4* posq, vvo_real4* posq_corr, vvo_mixed4* velm,
                         const vvo_int2* drude_pairs, vvo_mixed dt, vvo_mixed max_drude_distance,
                         vvo_mixed hardwall_scale_drude);
void vvo_reset_extra_force(int n, vvo_real3* force_extra);
void vvo_vv_integrate_velocities(int n, int padded, vvo_mixed4* velm, const long long* force,
                                 const vvo_real3* force_extra, vvo_mixed4* pos_delta, vvo_mixed dt,
                                 vvo_mixed fscale, int update_pos_delta);
void vvo_vv_integrate_positions(int n, vvo_real4* posq, vvo_real4* posq_corr, const vvo_mixed4* pos_delta,
                                vvo_mixed4* velm, vvo_mixed dt);
void vvo_calc_com_velocities(int nmol_nh, const vvo_mixed4* velm, vvo_mixed4* com_velm,
                             const vvo_int2* particles_in_molecules, const int* particles_sorted_by_mol_id,
                             const int* molecules_nh);
void vvo_normalize_velocities(int nnh, vvo_mixed4* velm, const vvo_mixed4* com_velm,
                              const int* particle_mol_id, const int* particles_nh);
void vvo_compute_kinetic_energies(int num_tg, int n_normal, int nmol_nh, int n_pairs, const vvo_mixed4* velm,
                                  const vvo_mixed4* com_velm, const int* normal, const vvo_int2* pairs,
                                  const int* molecules_nh, vvo_mixed* ke_out /*[num_tg]*/);
void vvo_scale_velocity(int n_normal, int n_pairs, vvo_mixed4* velm, const vvo_mixed4* com_velm,
                        const int* particle_mol_id, const int* normal, const vvo_int2* pairs,
                        const vvo_mixed* vscale /*[3]*/);
void vvo_add_cos_acceleration(int n, const vvo_real4* posq, const vvo_mixed4* velm, vvo_real3* force_extra,
                              vvo_real acceleration, vvo_real inv_box_z);
void vvo_calc_periodic_velocity_bias(int n, const vvo_real4* posq, const vvo_mixed4* velm, vvo_mixed* v_buffer,
                                     vvo_real inv_box_z);
void vvo_sum_v(int n, vvo_mixed* v_buffer, double inv_mass_total);
void vvo_remove_periodic_velocity_bias(int n, const vvo_real4* posq, vvo_mixed4* velm, const vvo_mixed* v_buffer,
                                       vvo_real inv_box_z);
void vvo_restore_periodic_velocity_bias(int n, const vvo_real4* posq, vvo_mixed4* velm, const vvo_mixed* v_buffer,
                                        vvo_real inv_box_z);
void vvo_add_extra_force_drude_langevin(int n_normal, int n_pairs, const vvo_mixed4* velm, vvo_real3* force_extra,
                                        const int* normal, const vvo_int2* pairs, vvo_mixed drag, vvo_mixed randf,
                                        vvo_mixed drag_drude, vvo_mixed randf_drude, const vvo_float4* random,
                                        unsigned int random_index);
void vvo_add_extra_force_electric_field(int n_el, const vvo_real4* posq, vvo_real3* force_extra,
                                        const int* particles_electrolyte, vvo_real efscale);
void vvo_update_image_positions(int n_img, vvo_real4* posq, vvo_real4* posq_corr, const vvo_int2* image_pairs,
                                vvo_mixed mirror);

/* ---- host-side restatements ---- */
/* OpenMM's SHAKE for hydrogen-type clusters (applyShakeToPositions / applyShakeToVelocities of its integration utilities; that
 * source is NOT under /root/reference, so this follows the published algorithm from the call sites HOST:151,176,351,427 and is
 * not pinned against OpenMM itself).  Positions: acts on the step displacement posDelta, old positions from posq(+corr). */
void vvo_shake_positions(int nclusters, const int* atoms, const float* params, vvo_mixed tol, const vvo_real4* posq,
                         const vvo_real4* posq_corr, vvo_mixed4* pos_delta);
void vvo_shake_velocities(int nclusters, const int* atoms, const float* params, vvo_mixed tol, const vvo_real4* posq,
                          const vvo_real4* posq_corr, vvo_mixed4* velm);
/* The same clusters with all their constraints solved together (vv_oracle.c; what the product runs unless VVHIP_SHAKE_MODE=0). */
void vvo_cluster_velocities_direct(int nclusters, const int* atoms, const float* params, const vvo_real4* posq,
                                   const vvo_real4* posq_corr, vvo_mixed4* velm);
void vvo_cluster_positions_newton(int nclusters, const int* atoms, const float* params, vvo_mixed tol, const vvo_real4* posq,
                                  const vvo_real4* posq_corr, vvo_mixed4* pos_delta);
/* SETTLE (Miyamoto & Kollman 1992) for rigid three-site molecules, on the step displacement / on the velocities; masses from
 * velm.w.  Written independently of the device code (vector form, the velocity multipliers by Cramer's rule); same unpinned status
 * as vvo_shake_*: OpenMM's source is not under /root/reference. */
void vvo_compute_virtual_sites(int n, const int* sites, const double* params, vvo_real4* posq, vvo_real4* posq_corr);
void vvo_settle_positions(int n, const int* atoms, const float* params, const vvo_real4* posq, const vvo_real4* posq_corr,
                          const vvo_mixed4* velm, vvo_mixed4* pos_delta);
void vvo_settle_velocities(int n, const int* atoms, const vvo_real4* posq, const vvo_real4* posq_corr, vvo_mixed4* velm);
void vvo_propagate_nh_chain(int num_chains, int loops_per_step, double step_size, double* eta, double* eta_dot,
                            double* eta_dotdot, const double* eta_mass, double ke2, double ke2_target,
                            double t_target, double* factor);
void vvo_nh_scale_velocity(vvo_system* s);     /* CudaModifyDrudeNoseKernel::scaleVelocity */
void vvo_tether_force(vvo_system* s);          /* ours: synthetic force provider           */
void vvo_step(vvo_system* s, int steps);       /* VVIntegrator::step (middle or classic)   */
double vvo_calc_viscosity(const vvo_system* s, double* vmax_out);
#ifdef __cplusplus
}
#endif
#endif
