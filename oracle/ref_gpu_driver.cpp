/*
 * oracle/ref_gpu_driver.cpp -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Host driver that runs the REFERENCE's kernels (built from /root/reference by `make refgpu`, see ref_gpu_prelude.h)
 * on the GPU in the reference's own per-step order for the middle scheme:
 *   VVIntegrator::stepMiddle            openmmapi/src/VVIntegrator.cpp:232-270
 *   Cuda*Kernel host methods            platforms/cuda/src/CudaVVKernels.cpp:119-231, 670-754, 1037-1110  ("HOST")
 * including what makes the reference slow: >= 10 launches per step, two single-block reductions, and the blocking
 * download of the kinetic energies -> host Nose-Hoover chain -> upload of the scale factors (HOST:709-746).
 * Launch shape follows OpenMM's executeKernel (not vendored; from memory): blocks of 64 threads, grid =
 * min(ceil(work/64), 4 x #CU) -- grid-stride loops in the kernels make the result independent of that choice.
 * Constraints, virtual sites and atom reordering are OpenMM's and identity here, as in the oracle.
 * The synthetic force provider is this repository's (same arithmetic as vvo_tether_force).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ref_gpu_prelude.h"

// ---- the reference's kernels (defined in the per-file translation units)
extern "C" {
__global__ void integrateMiddleVel(mixed4*, const long long*, const real3*, const mixed2*);
__global__ void integrateMiddlePos1(const mixed4*, mixed4*, mixed4*, const mixed2*);
__global__ void integrateMiddlePos2(const mixed4*, mixed4*, mixed4*, const mixed2*);
__global__ void integrateMiddlePos3(real4*, real4*, const mixed4*, const mixed4*, mixed4*, const mixed2*);
__global__ void applyHardWallConstraints(real4*, real4*, mixed4*, const int2*, const mixed2*, const mixed, const mixed);
__global__ void resetExtraForce(real3*);
__global__ void calcCOMVelocities(const mixed4*, mixed4*, const int2*, const int*, const int*);
__global__ void normalizeVelocities(mixed4*, const mixed4*, const int*, const int*);
__global__ void computeNormalizedKineticEnergies(const mixed4*, const mixed4*, const int*, const int2*, mixed*, const int*, int);
__global__ void sumNormalizedKineticEnergies(mixed*, mixed*, int);
__global__ void scaleVelocity(mixed4*, const mixed4*, const int*, const int*, const int2*, const mixed*);
__global__ void addCosAcceleration(const real4*, const mixed4*, real3*, real, const real4);
__global__ void calcPeriodicVelocityBias(const real4*, const mixed4*, mixed*, const real4);
__global__ void sumV(mixed*, double, int);
__global__ void removePeriodicVelocityBias(const real4*, mixed4*, const mixed*, const real4);
__global__ void restorePeriodicVelocityBias(const real4*, mixed4*, const mixed*, const real4);
void vvref_set_sizes_middle(const vvref_sizes_t*);
void vvref_set_sizes_nh(const vvref_sizes_t*);
void vvref_set_sizes_cos(const vvref_sizes_t*);
}

namespace {
const double kBoltz = (1.380649e-23 * 6.02214076e23) / 1000.0;

// ours: synthetic force provider, same arithmetic as oracle/vv_oracle.c vvo_tether_force (two kernels: the spring adds integers)
__global__ void tether_site(int n, int P, const real4* posq, const real4* site, const mixed4* velm, long long* force, real kt) {
    const real scale = (real) 4294967296.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x) {
        real fx = 0, fy = 0, fz = 0;
        if (velm[i].w != 0) { fx = -kt * (posq[i].x - site[i].x); fy = -kt * (posq[i].y - site[i].y); fz = -kt * (posq[i].z - site[i].z); }
        force[i] = (long long) (fx * scale); force[i + P] = (long long) (fy * scale); force[i + 2 * P] = (long long) (fz * scale);
    }
}
__global__ void tether_spring(int npairs, int P, const real4* posq, const int2* pairs, long long* force, real kd) {
    const real scale = (real) 4294967296.0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < npairs; k += blockDim.x * gridDim.x) {
        const int d = pairs[k].x, p = pairs[k].y;
        const long long ix = (long long) (-kd * (posq[d].x - posq[p].x) * scale), iy = (long long) (-kd * (posq[d].y - posq[p].y) * scale),
                        iz = (long long) (-kd * (posq[d].z - posq[p].z) * scale);
        force[d] += ix; force[d + P] += iy; force[d + 2 * P] += iz;       // every particle is in at most one pair
        force[p] -= ix; force[p + P] -= iy; force[p + 2 * P] -= iz;
    }
}

template <class T> T* dalloc(size_t n) { T* p = nullptr; (void) hipMalloc((void**) &p, std::max<size_t>(n, 1) * sizeof(T)); (void) hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)); return p; }
template <class T> T* dupload(const T* h, size_t n) { T* p = dalloc<T>(n); if (n) (void) hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice); return p; }
}  // namespace

struct vvrefgpu {
    int n, P, nmol, npairs_all, nnh, nmol_nh, nnormal, npairs_nh, num_tg, use_com, num_chains, loops;
    double dt, T, Td, maxd, cosacc, inv_mass_total, box[3], kt, kd;
    double eta[3][8], eta_dot[3][9], eta_dotdot[3][8], eta_mass[3][8], nkbt[3], ke2[3], vscale[3];
    mixed4 *velm, *pos_delta, *old_delta, *com;
    real4 *posq, *corr, *site;
    long long* force;
    real3* fextra;
    mixed *kebuf, *ke, *vs, *vbuf;
    mixed2* stepsize;
    int2 *drude_pairs, *pairs_nh, *pim;
    int *particles_nh, *molecules_nh, *normal_nh, *mol_id, *sorted;
    int grid_cap;
    hipStream_t stream;
};

static int grid(const vvrefgpu* r, int work) { return std::max(1, std::min((work + 63) / 64, r->grid_cap)); }

// host Nose-Hoover chain: restatement of VVIntegrator::propagateNHChain (openmmapi/src/VVIntegrator.cpp:340-376), as oracle/vv_oracle.c
static void propagate(vvrefgpu* r, int g, double ke2, double T, double* factor_out) {
    const int nc = r->num_chains;
    double *eta = r->eta[g], *ed = r->eta_dot[g], *edd = r->eta_dotdot[g], *mass = r->eta_mass[g];
    double expfac = 1.0, factor = 1.0;
    const double dt2 = r->dt / r->loops / 2, dt4 = dt2 / 2, dt8 = dt4 / 2, target = r->nkbt[g];
    edd[0] = (ke2 - target) / mass[0];
    for (int l = 0; l < r->loops; l++) {
        for (int i = nc - 1; i >= 0; i--) { expfac = std::exp(-dt8 * ed[i + 1]); ed[i] *= expfac; ed[i] += edd[i] * dt4; ed[i] *= expfac; }
        factor *= std::exp(-dt2 * ed[0]);
        for (int i = 0; i < nc; i++) eta[i] += dt2 * ed[i];
        edd[0] = (ke2 * factor * factor - target) / mass[0];
        ed[0] *= expfac; ed[0] += edd[0] * dt4; ed[0] *= expfac;
        for (int i = 1; i < nc; i++) {
            expfac = std::exp(-dt8 * ed[i + 1]); ed[i] *= expfac;
            edd[i] = (mass[i - 1] * ed[i - 1] * ed[i - 1] - kBoltz * T) / mass[i];
            ed[i] += edd[i] * dt4; ed[i] *= expfac;
        }
    }
    *factor_out = factor;
}

extern "C" {

vvrefgpu* vvrefgpu_create(int n, int P, int nmol, const void* velm, const void* posq, const void* corr, const int* drude_pairs, int npairs_all,
                          const int* particles_nh, int nnh, const int* molecules_nh, int nmol_nh, const int* normal_nh, int nnormal,
                          const int* pairs_nh, int npairs_nh, const int* mol_id, const int* pim, const int* sorted, int num_tg, int use_com,
                          int num_chains, int loops, const double* eta_mass /*[3][8]*/, const double* nkbt, double dt, double T, double Td,
                          double maxd, double cosacc, double inv_mass_total, const double* box, double kt, double kd) {
    vvrefgpu* r = new vvrefgpu();
    std::memset(r, 0, sizeof(*r));
    r->n = n; r->P = P; r->nmol = nmol; r->npairs_all = npairs_all; r->nnh = nnh; r->nmol_nh = nmol_nh; r->nnormal = nnormal; r->npairs_nh = npairs_nh;
    r->num_tg = num_tg; r->use_com = use_com; r->num_chains = num_chains; r->loops = loops;
    r->dt = dt; r->T = T; r->Td = Td; r->maxd = maxd; r->cosacc = cosacc; r->inv_mass_total = inv_mass_total; r->kt = kt; r->kd = kd;
    for (int i = 0; i < 3; i++) { r->box[i] = box[i]; r->nkbt[i] = nkbt[i]; for (int k = 0; k < 8; k++) r->eta_mass[i][k] = eta_mass[i * 8 + k]; }
    r->velm = dupload((const mixed4*) velm, n); r->posq = dupload((const real4*) posq, n); r->corr = dupload((const real4*) corr, n);
    r->site = dupload((const real4*) posq, n);
    r->pos_delta = dalloc<mixed4>(n); r->old_delta = dalloc<mixed4>(n); r->com = dalloc<mixed4>(nmol);
    r->force = dalloc<long long>((size_t) 3 * P); r->fextra = dalloc<real3>(n);       // zero-initialised (HOST:79-89)
    r->kebuf = dalloc<mixed>((size_t) std::max(nnh, 1) * num_tg);                   // fresh allocation == zero (quirk Q3)
    r->ke = dalloc<mixed>(3); r->vs = dalloc<mixed>(3); r->vbuf = dalloc<mixed>(n);
    const mixed2 ss = {0, (mixed) dt};
    r->stepsize = dupload(&ss, 1);
    r->drude_pairs = dupload((const int2*) drude_pairs, npairs_all); r->pairs_nh = dupload((const int2*) pairs_nh, npairs_nh);
    r->pim = dupload((const int2*) pim, nmol);
    r->particles_nh = dupload(particles_nh, nnh); r->molecules_nh = dupload(molecules_nh, nmol_nh); r->normal_nh = dupload(normal_nh, nnormal);
    r->mol_id = dupload(mol_id, n); r->sorted = dupload(sorted, n);
    hipDeviceProp_t prop;
    (void) hipGetDeviceProperties(&prop, 0);
    r->grid_cap = 4 * prop.multiProcessorCount;
    (void) hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    vvref_sizes_t s = {n, P, npairs_all, nnh, nmol_nh, nnormal, npairs_nh, 0, 0, 0, 0};
    vvref_set_sizes_middle(&s); vvref_set_sizes_nh(&s); vvref_set_sizes_cos(&s);
    (void) hipDeviceSynchronize();
    return r;
}

void vvrefgpu_step(vvrefgpu* r, int nsteps) {
    hipStream_t st = r->stream;
    const int n = r->n;
    const real4 invBox = {(real) (1.0 / r->box[0]), (real) (1.0 / r->box[1]), (real) (1.0 / r->box[2]), 0};
    const mixed hw = (mixed) std::sqrt(kBoltz * r->Td);
    for (int s = 0; s < nsteps; s++) {
        // context->calcForcesAndEnergy: synthetic provider
        hipLaunchKernelGGL(tether_site, dim3(grid(r, n)), dim3(64), 0, st, n, r->P, r->posq, r->site, r->velm, r->force, (real) r->kt);
        if (r->npairs_all) hipLaunchKernelGGL(tether_spring, dim3(grid(r, r->npairs_all)), dim3(64), 0, st, r->npairs_all, r->P, r->posq, r->drude_pairs, r->force, (real) r->kd);
        if (r->cosacc != 0) {                                                     // VVIntegrator.cpp:238-245
            hipLaunchKernelGGL(resetExtraForce, dim3(grid(r, n)), dim3(64), 0, st, r->fextra);
            hipLaunchKernelGGL(addCosAcceleration, dim3(grid(r, n)), dim3(64), 0, st, r->posq, r->velm, r->fextra, (real) r->cosacc, invBox);
        }
        hipLaunchKernelGGL(integrateMiddleVel, dim3(grid(r, n)), dim3(64), 0, st, r->velm, r->force, r->fextra, r->stepsize);        // HOST:144-148
        hipLaunchKernelGGL(integrateMiddlePos1, dim3(grid(r, n)), dim3(64), 0, st, r->velm, r->pos_delta, r->old_delta, r->stepsize);   // HOST:154-158
        if (r->nnh > 0) {                                                         // VVIntegrator.cpp:251-260
            if (r->cosacc != 0) {                                                 // HOST:1061-1095
                hipLaunchKernelGGL(calcPeriodicVelocityBias, dim3(grid(r, n)), dim3(64), 0, st, r->posq, r->velm, r->vbuf, invBox);
                hipLaunchKernelGGL(sumV, dim3(1), dim3(512), 512 * sizeof(mixed), st, r->vbuf, r->inv_mass_total, n);
                hipLaunchKernelGGL(removePeriodicVelocityBias, dim3(grid(r, n)), dim3(64), 0, st, r->posq, r->velm, r->vbuf, invBox);
            }
            if (r->use_com) {                                                     // HOST:676-689
                hipLaunchKernelGGL(calcCOMVelocities, dim3(grid(r, r->nmol_nh)), dim3(64), 0, st, r->velm, r->com, r->pim, r->sorted, r->molecules_nh);
                hipLaunchKernelGGL(normalizeVelocities, dim3(grid(r, r->nnh)), dim3(64), 0, st, r->velm, r->com, r->mol_id, r->particles_nh);
            }
            const int bufsize = std::max(r->nnh, 1) * r->num_tg;
            hipLaunchKernelGGL(computeNormalizedKineticEnergies, dim3(grid(r, r->nnh)), dim3(64), 0, st, r->velm, r->com, r->normal_nh, r->pairs_nh, r->kebuf, r->molecules_nh, bufsize);
            hipLaunchKernelGGL(sumNormalizedKineticEnergies, dim3(1), dim3(512), 512 * r->num_tg * sizeof(mixed), st, r->kebuf, r->ke, bufsize);   // HOST:702-707
            mixed keh[3] = {0, 0, 0};
            (void) hipMemcpyAsync(keh, r->ke, r->num_tg * sizeof(mixed), hipMemcpyDeviceToHost, st);                                               // HOST:709-716
            (void) hipStreamSynchronize(st);                                      //   ... blocking
            mixed vsh[3] = {1, 1, 1};
            for (int g = 0; g < r->num_tg; g++) {                                 // HOST:726-733
                r->ke2[g] = (double) keh[g];
                double f = 1.0;
                if (r->eta_mass[g][0] > 0) propagate(r, g, r->ke2[g], g == 2 ? r->Td : r->T, &f);
                r->vscale[g] = f;
                vsh[g] = (mixed) f;
            }
            (void) hipMemcpyAsync(r->vs, vsh, 3 * sizeof(mixed), hipMemcpyHostToDevice, st);                                                       // HOST:741-746
            hipLaunchKernelGGL(scaleVelocity, dim3(grid(r, r->nnh)), dim3(64), 0, st, r->velm, r->com, r->mol_id, r->normal_nh, r->pairs_nh, r->vs);
            if (r->cosacc != 0)
                hipLaunchKernelGGL(restorePeriodicVelocityBias, dim3(grid(r, n)), dim3(64), 0, st, r->posq, r->velm, r->vbuf, invBox);
        }
        hipLaunchKernelGGL(integrateMiddlePos2, dim3(grid(r, n)), dim3(64), 0, st, r->velm, r->pos_delta, r->old_delta, r->stepsize);    // HOST:169-173
        hipLaunchKernelGGL(integrateMiddlePos3, dim3(grid(r, n)), dim3(64), 0, st, r->posq, r->corr, r->pos_delta, r->old_delta, r->velm, r->stepsize);   // HOST:179-185
        if (r->maxd > 0 && r->npairs_all)
            hipLaunchKernelGGL(applyHardWallConstraints, dim3(grid(r, r->npairs_all)), dim3(64), 0, st, r->posq, r->corr, r->velm, r->drude_pairs, r->stepsize, (mixed) r->maxd, hw);
    }
}

void vvrefgpu_sync(vvrefgpu* r) { (void) hipStreamSynchronize(r->stream); }
void vvrefgpu_download(vvrefgpu* r, void* velm, void* posq, void* corr, double* ke2, double* vscale) {
    (void) hipStreamSynchronize(r->stream);
    (void) hipMemcpy(velm, r->velm, (size_t) r->n * sizeof(mixed4), hipMemcpyDeviceToHost);
    (void) hipMemcpy(posq, r->posq, (size_t) r->n * sizeof(real4), hipMemcpyDeviceToHost);
    (void) hipMemcpy(corr, r->corr, (size_t) r->n * sizeof(real4), hipMemcpyDeviceToHost);
    for (int g = 0; g < 3; g++) { ke2[g] = r->ke2[g]; vscale[g] = r->vscale[g]; }
}
void vvrefgpu_destroy(vvrefgpu* r) {
    if (!r) return;
    (void) hipStreamSynchronize(r->stream);
    for (void* p : {(void*) r->velm, (void*) r->pos_delta, (void*) r->old_delta, (void*) r->com, (void*) r->posq, (void*) r->corr, (void*) r->site,
                    (void*) r->force, (void*) r->fextra, (void*) r->kebuf, (void*) r->ke, (void*) r->vs, (void*) r->vbuf, (void*) r->stepsize,
                    (void*) r->drude_pairs, (void*) r->pairs_nh, (void*) r->pim, (void*) r->particles_nh, (void*) r->molecules_nh,
                    (void*) r->normal_nh, (void*) r->mol_id, (void*) r->sorted})
        (void) hipFree(p);
    (void) hipStreamDestroy(r->stream);
    delete r;
}
}  // extern "C"
