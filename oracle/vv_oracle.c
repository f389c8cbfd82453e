/*
 * oracle/vv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See vv_oracle.h.
 *
 * Abbreviations in the citations below (all under /root/reference):
 *   K/   = platforms/cuda/src/kernels/
 *   HOST = platforms/cuda/src/CudaVVKernels.cpp
 *   API  = openmmapi/src/VVIntegrator.cpp
 *
 * Every arithmetic expression keeps the operand types, the association order and
 * the implicit conversions the reference expression has under C/C++ rules, so that
 * with -ffp-contract=off the results are bit-identical to the reference kernels
 * executed as one thread (oracle/_ref).  "serial" below means exactly that
 * one-thread order; the OpenMP variants (used only for the timed CPU baseline with
 * >1 thread) change the order of reductions and nothing else.
 */
#include "vv_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef vvo_real real;
typedef vvo_mixed mixed;
typedef vvo_real4 real4;
typedef vvo_real3 real3;
typedef vvo_mixed4 mixed4;
typedef vvo_int2 int2;

/* What OpenMM's CudaContext prepends (see ref_prelude.h for the stated assumption). */
#if defined(VVO_DOUBLE) || defined(VVO_ALT_PRELUDE)
/* VVO_ALT_PRELUDE (`make altprelude`, mixed mode only): the OTHER reading of what OpenMM 8.1.2 prepends in mixed precision -- SQRT and
 * RECIP in double wherever their argument is `mixed` -- built to measure how much rests on the assumption stated in ref_prelude.h
 * (tests/test_fmad_gap.py::test_the_prelude_assumption_is_worth_one_float_ulp). */
#define SQRT sqrt
#define RECIP(x) (1.0/(x))
#else
#define SQRT sqrtf
#define RECIP(x) (1.0f/(x))
#endif
#if defined(VVO_MIXED)
#define USE_MIXED_PRECISION 1
#endif

/* SimTKOpenMMRealType.h (OpenMM 8.1.2, not vendored): BOLTZ = RGAS/KILO, RGAS = BOLTZMANN*AVOGADRO */
#define VVO_AVOGADRO (6.02214076e23)
#define VVO_BOLTZ ((1.380649e-23 * VVO_AVOGADRO) / 1000.0)
enum { TG_ATOM = 0, TG_COM = 1, TG_DRUDE = 2 };

static int g_threads = 1;
#define PAR_FOR _Pragma("omp parallel for schedule(static) if(g_threads > 1) num_threads(g_threads)")

int vvo_sizeof_real(void) { return (int) sizeof(real); }
int vvo_sizeof_mixed(void) { return (int) sizeof(mixed); }
int vvo_sizeof_system(void) { return (int) sizeof(vvo_system); }

/* ------------------------------------------------------------------ K/middle.cu:6-23 */
void vvo_integrate_middle_vel(int n, int padded, mixed4* velm, const long long* force,
                              const real3* force_extra, mixed dt) {
    mixed stepSize = dt;
    mixed fscale = stepSize / (mixed) 0x100000000;
    PAR_FOR
    for (int index = 0; index < n; index++) {
        mixed4 velocity = velm[index];
        if (velocity.w != 0) {
            velocity.x += stepSize * velocity.w * force_extra[index].x + fscale * velocity.w * force[index];
            velocity.y += stepSize * velocity.w * force_extra[index].y + fscale * velocity.w * force[index + padded];
            velocity.z += stepSize * velocity.w * force_extra[index].z + fscale * velocity.w * force[index + padded * 2];
            velm[index] = velocity;
        }
    }
}

/* ------------------------------------------------------------------ K/middle.cu:29-42 */
void vvo_integrate_middle_pos1(int n, const mixed4* velm, mixed4* pos_delta, mixed4* old_delta, mixed dt) {
    mixed halfdt = 0.5f * dt;
    PAR_FOR
    for (int index = 0; index < n; index++) {
        mixed4 velocity = velm[index];
        if (velocity.w != 0) {
            mixed4 delta = { halfdt * velocity.x, halfdt * velocity.y, halfdt * velocity.z, 0 };
            pos_delta[index] = delta;
            old_delta[index] = delta;
        }
    }
}

/* ------------------------------------------------------------------ K/middle.cu:47-60
 * `posDelta[index] += delta` is vectorOps.cu's 4-component operator+= (K/vectorOps.cu:255,267). */
void vvo_integrate_middle_pos2(int n, const mixed4* velm, mixed4* pos_delta, mixed4* old_delta, mixed dt) {
    mixed halfdt = 0.5f * dt;
    PAR_FOR
    for (int index = 0; index < n; index++) {
        mixed4 velocity = velm[index];
        if (velocity.w != 0) {
            mixed4 delta = { halfdt * velocity.x, halfdt * velocity.y, halfdt * velocity.z, 0 };
            pos_delta[index].x += delta.x; pos_delta[index].y += delta.y;
            pos_delta[index].z += delta.z; pos_delta[index].w += delta.w;
            old_delta[index].x += delta.x; old_delta[index].y += delta.y;
            old_delta[index].z += delta.z; old_delta[index].w += delta.w;
        }
    }
}

/* position read-modify-write shared by Pos3 / VV positions / hard wall (K/middle.cu:81-96) */
static inline void load_pos(const real4* posq, const real4* posq_corr, int i, mixed* x, mixed* y, mixed* z, mixed* w) {
#ifdef USE_MIXED_PRECISION
    real4 pos1 = posq[i];
    real4 pos2 = posq_corr[i];
    *x = pos1.x + (mixed) pos2.x; *y = pos1.y + (mixed) pos2.y; *z = pos1.z + (mixed) pos2.z; *w = pos1.w;
#else
    (void) posq_corr;
    real4 pos = posq[i];
    *x = pos.x; *y = pos.y; *z = pos.z; *w = pos.w;
#endif
}
static inline void store_pos(real4* posq, real4* posq_corr, int i, mixed x, mixed y, mixed z, mixed w) {
#ifdef USE_MIXED_PRECISION
    real4 p = { (real) x, (real) y, (real) z, (real) w };
    real4 c = { (real) (x - (real) x), (real) (y - (real) y), (real) (z - (real) z), 0 };
    posq[i] = p;
    posq_corr[i] = c;
#else
    (void) posq_corr;
    real4 p = { (real) x, (real) y, (real) z, (real) w };
    posq[i] = p;
#endif
}

/* ------------------------------------------------------------------ K/middle.cu:66-100 */
void vvo_integrate_middle_pos3(int n, real4* posq, real4* posq_corr, const mixed4* pos_delta,
                               const mixed4* old_delta, mixed4* velm, mixed dt) {
    mixed invDt = 1 / dt;
    PAR_FOR
    for (int index = 0; index < n; index++) {
        mixed4 velocity = velm[index];
        if (velocity.w != 0.0) {
            mixed4 delta = pos_delta[index];
            velocity.x += (delta.x - old_delta[index].x) * invDt;
            velocity.y += (delta.y - old_delta[index].y) * invDt;
            velocity.z += (delta.z - old_delta[index].z) * invDt;
            velm[index] = velocity;
#ifdef USE_MIXED_PRECISION
            mixed px, py, pz, pw;
            load_pos(posq, posq_corr, index, &px, &py, &pz, &pw);
            px += delta.x; py += delta.y; pz += delta.z;
            store_pos(posq, posq_corr, index, px, py, pz, pw);
#else
            real4 pos = posq[index];          /* real += mixed, evaluated in the promoted type */
            pos.x += delta.x; pos.y += delta.y; pos.z += delta.z;
            posq[index] = pos;
#endif
        }
    }
}

/* ------------------------------------------------------------------ K/middle.cu:106-221
 * (= K/velocityVerlet.cu:74-189).  x = Drude, y = parent. */
void vvo_apply_hard_wall(int npairs, real4* posq, real4* posq_corr, mixed4* velm, const int2* drude_pairs,
                         mixed dt, const mixed maxDrudeDistance, const mixed hardwallscaleDrude) {
    mixed stepSize = dt;
    PAR_FOR
    for (int i = 0; i < npairs; i++) {
        int2 particles = drude_pairs[i];
        mixed4 pos1, pos2;
        load_pos(posq, posq_corr, particles.x, &pos1.x, &pos1.y, &pos1.z, &pos1.w);
        load_pos(posq, posq_corr, particles.y, &pos2.x, &pos2.y, &pos2.z, &pos2.w);
        mixed4 delta = { pos1.x - pos2.x, pos1.y - pos2.y, pos1.z - pos2.z, pos1.w - pos2.w };
        mixed r = SQRT(delta.x * delta.x + delta.y * delta.y + delta.z * delta.z);
        mixed rInv = RECIP(r);
        if (rInv * maxDrudeDistance < 1) {
            mixed4 bondDir = { delta.x * rInv, delta.y * rInv, delta.z * rInv, delta.w * rInv };
            mixed4 vel1 = velm[particles.x];
            mixed4 vel2 = velm[particles.y];
            mixed mass1 = RECIP(vel1.w);
            mixed mass2 = RECIP(vel2.w);
            mixed deltaR = r - maxDrudeDistance;
            mixed deltaT = stepSize;
            mixed dotvr1 = vel1.x * bondDir.x + vel1.y * bondDir.y + vel1.z * bondDir.z;
            mixed4 vb1 = { bondDir.x * dotvr1, bondDir.y * dotvr1, bondDir.z * dotvr1, bondDir.w * dotvr1 };
            mixed4 vp1 = { vel1.x - vb1.x, vel1.y - vb1.y, vel1.z - vb1.z, vel1.w - vb1.w };
            if (vel2.w == 0) {
                /* massless parent: move only the Drude particle (K/middle.cu:151-173) */
                if (dotvr1 != 0)
                    deltaT = deltaR / fabs(dotvr1);
                if (deltaT > stepSize)
                    deltaT = stepSize;
                dotvr1 = -dotvr1 * hardwallscaleDrude / (fabs(dotvr1) * SQRT(mass1));
                mixed dr = -deltaR + deltaT * dotvr1;
                pos1.x += bondDir.x * dr;
                pos1.y += bondDir.y * dr;
                pos1.z += bondDir.z * dr;
                store_pos(posq, posq_corr, particles.x, pos1.x, pos1.y, pos1.z, pos1.w);
                vel1.x = vp1.x + bondDir.x * dotvr1;
                vel1.y = vp1.y + bondDir.y * dotvr1;
                vel1.z = vp1.z + bondDir.z * dotvr1;
                velm[particles.x] = vel1;
            }
            else {
                /* move both particles (K/middle.cu:174-218) */
                mixed invTotalMass = RECIP(mass1 + mass2);
                mixed dotvr2 = vel2.x * bondDir.x + vel2.y * bondDir.y + vel2.z * bondDir.z;
                mixed4 vb2 = { bondDir.x * dotvr2, bondDir.y * dotvr2, bondDir.z * dotvr2, bondDir.w * dotvr2 };
                mixed4 vp2 = { vel2.x - vb2.x, vel2.y - vb2.y, vel2.z - vb2.z, vel2.w - vb2.w };
                mixed vbCMass = (mass1 * dotvr1 + mass2 * dotvr2) * invTotalMass;
                dotvr1 -= vbCMass;
                dotvr2 -= vbCMass;
                if (dotvr1 != dotvr2)
                    deltaT = deltaR / fabs(dotvr1 - dotvr2);
                if (deltaT > stepSize)
                    deltaT = stepSize;
                mixed vBond = hardwallscaleDrude / SQRT(mass1);
                dotvr1 = -dotvr1 * vBond * mass2 * invTotalMass / fabs(dotvr1);
                dotvr2 = -dotvr2 * vBond * mass1 * invTotalMass / fabs(dotvr2);
                mixed dr1 = -deltaR * mass2 * invTotalMass + deltaT * dotvr1;
                mixed dr2 = deltaR * mass1 * invTotalMass + deltaT * dotvr2;
                dotvr1 += vbCMass;
                dotvr2 += vbCMass;
                pos1.x += bondDir.x * dr1;
                pos1.y += bondDir.y * dr1;
                pos1.z += bondDir.z * dr1;
                pos2.x += bondDir.x * dr2;
                pos2.y += bondDir.y * dr2;
                pos2.z += bondDir.z * dr2;
                store_pos(posq, posq_corr, particles.x, pos1.x, pos1.y, pos1.z, pos1.w);
                store_pos(posq, posq_corr, particles.y, pos2.x, pos2.y, pos2.z, pos2.w);
                vel1.x = vp1.x + bondDir.x * dotvr1;
                vel1.y = vp1.y + bondDir.y * dotvr1;
                vel1.z = vp1.z + bondDir.z * dotvr1;
                vel2.x = vp2.x + bondDir.x * dotvr2;
                vel2.y = vp2.y + bondDir.y * dotvr2;
                vel2.z = vp2.z + bondDir.z * dotvr2;
                velm[particles.x] = vel1;
                velm[particles.y] = vel2;
            }
        }
    }
}

/* ------------------------------------------------------------------ K/middle.cu:227-231 */
void vvo_reset_extra_force(int n, real3* force_extra) {
    PAR_FOR
    for (int i = 0; i < n; i++) {
        real3 z = { 0, 0, 0 };
        force_extra[i] = z;
    }
}

/* ------------------------------------------------------------------ K/velocityVerlet.cu:6-29
 * NB the literal 0.5 is a double in every precision mode. */
void vvo_vv_integrate_velocities(int n, int padded, mixed4* velm, const long long* force, const real3* force_extra,
                                 mixed4* pos_delta, mixed dt, const mixed fscale, int updatePosDelta) {
    mixed stepSize = dt;
    PAR_FOR
    for (int index = 0; index < n; index++) {
        mixed4 velocity = velm[index];
        if (velocity.w != 0) {
            velocity.x += 0.5 * stepSize * velocity.w * force_extra[index].x + fscale * velocity.w * force[index];
            velocity.y += 0.5 * stepSize * velocity.w * force_extra[index].y + fscale * velocity.w * force[index + padded];
            velocity.z += 0.5 * stepSize * velocity.w * force_extra[index].z + fscale * velocity.w * force[index + padded * 2];
            velm[index] = velocity;
            if (updatePosDelta) {
                mixed4 d = { stepSize * velocity.x, stepSize * velocity.y, stepSize * velocity.z, 0 };
                pos_delta[index] = d;
            }
        }
    }
}

/* ------------------------------------------------------------------ K/velocityVerlet.cu:35-68 */
void vvo_vv_integrate_positions(int n, real4* posq, real4* posq_corr, const mixed4* pos_delta, mixed4* velm, mixed dt) {
    mixed invStepSize = 1.0 / dt;
    PAR_FOR
    for (int index = 0; index < n; index++) {
        mixed4 vel = velm[index];
        if (vel.w != 0) {
            mixed4 delta = pos_delta[index];
#ifdef USE_MIXED_PRECISION
            mixed px, py, pz, pw;
            load_pos(posq, posq_corr, index, &px, &py, &pz, &pw);
            px += delta.x; py += delta.y; pz += delta.z;
#else
            real4 pos = posq[index];
            pos.x += delta.x; pos.y += delta.y; pos.z += delta.z;
#endif
            vel.x = (mixed) (invStepSize * delta.x);
            vel.y = (mixed) (invStepSize * delta.y);
            vel.z = (mixed) (invStepSize * delta.z);
#ifdef USE_MIXED_PRECISION
            store_pos(posq, posq_corr, index, px, py, pz, pw);
#else
            posq[index] = pos;
#endif
            velm[index] = vel;
        }
    }
}

/* ------------------------------------------------------------------ K/drudeNoseHoover.cu:5-31
 * accumulates in comVelm memory, in particlesSortedByMolId order. */
void vvo_calc_com_velocities(int nmol_nh, const mixed4* velm, mixed4* comVelm, const int2* particlesInMolecules,
                             const int* particlesSortedByMolId, const int* moleculesNH) {
    PAR_FOR
    for (int i = 0; i < nmol_nh; i++) {
        int id_mol = moleculesNH[i];
        mixed4 zero = { 0, 0, 0, 0 };
        comVelm[id_mol] = zero;
        mixed comMass = 0.0;
        for (int j = 0; j < particlesInMolecules[id_mol].x; j++) {
            int index = particlesSortedByMolId[particlesInMolecules[id_mol].y + j];
            mixed4 velocity = velm[index];
            if (velocity.w != 0) {
                mixed mass = RECIP(velocity.w);
                comVelm[id_mol].x += velocity.x * mass;
                comVelm[id_mol].y += velocity.y * mass;
                comVelm[id_mol].z += velocity.z * mass;
                comMass += mass;
            }
        }
        comVelm[id_mol].w = RECIP(comMass);
        comVelm[id_mol].x *= comVelm[id_mol].w;
        comVelm[id_mol].y *= comVelm[id_mol].w;
        comVelm[id_mol].z *= comVelm[id_mol].w;
    }
}

/* ------------------------------------------------------------------ K/drudeNoseHoover.cu:37-49 */
void vvo_normalize_velocities(int nnh, mixed4* velm, const mixed4* comVelm, const int* particleMolId,
                              const int* particlesNH) {
    PAR_FOR
    for (int i = 0; i < nnh; i++) {
        int index = particlesNH[i];
        int id_mol = particleMolId[index];
        velm[index].x -= comVelm[id_mol].x;
        velm[index].y -= comVelm[id_mol].y;
        velm[index].z -= comVelm[id_mol].z;
    }
}

/* ------------------------------------------------------------------ K/drudeNoseHoover.cu:55-151
 * computeNormalizedKineticEnergies + sumNormalizedKineticEnergies.  The values are
 * sum(m v^2) = 2*KE.  Serial order: normal particles, then molecules, then pairs,
 * each added straight into the running per-group sum (one thread => slot 0). */
static inline void ke_pair_terms(const mixed4* velm, int2 pair, mixed* atom, mixed* drude) {
    mixed4 velocity1 = velm[pair.x];
    mixed4 velocity2 = velm[pair.y];
    mixed mass1 = RECIP(velocity1.w);
    mixed mass2 = RECIP(velocity2.w);
    mixed invTotalMass = RECIP(mass1 + mass2);
    mixed invReducedMass = (mass1 + mass2) * velocity1.w * velocity2.w;
    mixed mass1fract = invTotalMass * mass1;
    mixed mass2fract = invTotalMass * mass2;
    mixed cx = velocity1.x * mass1fract + velocity2.x * mass2fract;
    mixed cy = velocity1.y * mass1fract + velocity2.y * mass2fract;
    mixed cz = velocity1.z * mass1fract + velocity2.z * mass2fract;
    mixed rx = velocity1.x - velocity2.x, ry = velocity1.y - velocity2.y, rz = velocity1.z - velocity2.z;
    *atom = (cx * cx + cy * cy + cz * cz) * (mass1 + mass2);
    *drude = (rx * rx + ry * ry + rz * rz) / invReducedMass;
}
void vvo_compute_kinetic_energies(int num_tg, int n_normal, int nmol_nh, int n_pairs, const mixed4* velm,
                                  const mixed4* comVelm, const int* normalParticles, const int2* pairParticles,
                                  const int* moleculesNH, mixed* ke_out) {
    mixed keAtom = 0, keCom = 0, keDrude = 0;
    if (g_threads <= 1) {
        for (int i = 0; i < n_normal; i++) {
            mixed4 velocity = velm[normalParticles[i]];
            if (velocity.w != 0)
                keAtom += (velocity.x * velocity.x + velocity.y * velocity.y + velocity.z * velocity.z) / velocity.w;
        }
        if (num_tg > TG_COM) {
            for (int i = 0; i < nmol_nh; i++) {
                mixed4 velocity = comVelm[moleculesNH[i]];
                if (velocity.w != 0)
                    keCom += (velocity.x * velocity.x + velocity.y * velocity.y + velocity.z * velocity.z) / velocity.w;
            }
        }
        for (int i = 0; i < n_pairs; i++) {
            mixed a, d;
            ke_pair_terms(velm, pairParticles[i], &a, &d);
            keAtom += a;
            keDrude += d;
        }
    }
    else {
        mixed a0 = 0, c0 = 0, a1 = 0, d1 = 0;
        #pragma omp parallel for schedule(static) reduction(+:a0) num_threads(g_threads)
        for (int i = 0; i < n_normal; i++) {
            mixed4 velocity = velm[normalParticles[i]];
            if (velocity.w != 0)
                a0 += (velocity.x * velocity.x + velocity.y * velocity.y + velocity.z * velocity.z) / velocity.w;
        }
        if (num_tg > TG_COM) {
            #pragma omp parallel for schedule(static) reduction(+:c0) num_threads(g_threads)
            for (int i = 0; i < nmol_nh; i++) {
                mixed4 velocity = comVelm[moleculesNH[i]];
                if (velocity.w != 0)
                    c0 += (velocity.x * velocity.x + velocity.y * velocity.y + velocity.z * velocity.z) / velocity.w;
            }
        }
        #pragma omp parallel for schedule(static) reduction(+:a1,d1) num_threads(g_threads)
        for (int i = 0; i < n_pairs; i++) {
            mixed a, d;
            ke_pair_terms(velm, pairParticles[i], &a, &d);
            a1 += a;
            d1 += d;
        }
        keAtom = a0 + a1; keCom = c0; keDrude = d1;
    }
    ke_out[TG_ATOM] = keAtom;
    if (num_tg > TG_COM) ke_out[TG_COM] = keCom;
    if (num_tg > TG_DRUDE) ke_out[TG_DRUDE] = keDrude;
}

/* ------------------------------------------------------------------ K/drudeNoseHoover.cu:157-209
 * vscale always has 3 entries here (the reference reads [1],[2] even when NUM_TG < 3: quirk Q4). */
void vvo_scale_velocity(int n_normal, int n_pairs, mixed4* velm, const mixed4* comVelm, const int* particleMolId,
                        const int* normalParticles, const int2* pairParticles, const mixed* vscaleFactors) {
    mixed vscaleAtom = vscaleFactors[0];
    mixed vscaleCOM = vscaleFactors[1];
    mixed vscaleDrude = vscaleFactors[2];
    PAR_FOR
    for (int i = 0; i < n_normal; i++) {
        int index = normalParticles[i];
        int id_mol = particleMolId[index];
        mixed4 velCOM = comVelm[id_mol];
        if (velm[index].w != 0) {
            velm[index].x = vscaleAtom * velm[index].x + vscaleCOM * velCOM.x;
            velm[index].y = vscaleAtom * velm[index].y + vscaleCOM * velCOM.y;
            velm[index].z = vscaleAtom * velm[index].z + vscaleCOM * velCOM.z;
        }
    }
    PAR_FOR
    for (int i = 0; i < n_pairs; i++) {
        int2 particles = pairParticles[i];
        int id_mol = particleMolId[particles.x];
        mixed4 velAtom1 = velm[particles.x];
        mixed4 velAtom2 = velm[particles.y];
        mixed4 velCOM = comVelm[id_mol];
        mixed mass1 = RECIP(velAtom1.w);
        mixed mass2 = RECIP(velAtom2.w);
        mixed invTotalMass = RECIP(mass1 + mass2);
        mixed mass1fract = invTotalMass * mass1;
        mixed mass2fract = invTotalMass * mass2;
        mixed cmx = velAtom1.x * mass1fract + velAtom2.x * mass2fract;
        mixed cmy = velAtom1.y * mass1fract + velAtom2.y * mass2fract;
        mixed cmz = velAtom1.z * mass1fract + velAtom2.z * mass2fract;
        mixed rx = velAtom2.x - velAtom1.x, ry = velAtom2.y - velAtom1.y, rz = velAtom2.z - velAtom1.z;
        cmx = vscaleAtom * cmx; cmy = vscaleAtom * cmy; cmz = vscaleAtom * cmz;
        rx = vscaleDrude * rx; ry = vscaleDrude * ry; rz = vscaleDrude * rz;
        velAtom1.x = cmx - rx * mass2fract + vscaleCOM * velCOM.x;
        velAtom1.y = cmy - ry * mass2fract + vscaleCOM * velCOM.y;
        velAtom1.z = cmz - rz * mass2fract + vscaleCOM * velCOM.z;
        velAtom2.x = cmx + rx * mass1fract + vscaleCOM * velCOM.x;
        velAtom2.y = cmy + ry * mass1fract + vscaleCOM * velCOM.y;
        velAtom2.z = cmz + rz * mass1fract + vscaleCOM * velCOM.z;
        velm[particles.x] = velAtom1;
        velm[particles.y] = velAtom2;
    }
}

/* ------------------------------------------------------------------ K/cosineAccelerate.cu:2-14
 * pi is the literal 3.1415926 and the cosine is evaluated in double in every mode (quirk Q2). */
void vvo_add_cos_acceleration(int n, const real4* posq, const mixed4* velm, real3* force_extra, real acceleration,
                              real invBoxSizeZ) {
    PAR_FOR
    for (int index = 0; index < n; index++)
        force_extra[index].x += acceleration * cos(2 * 3.1415926 * posq[index].z * invBoxSizeZ) * RECIP(velm[index].w);
}

/* ------------------------------------------------------------------ K/cosineAccelerate.cu:16-32 */
void vvo_calc_periodic_velocity_bias(int n, const real4* posq, const mixed4* velm, mixed* VBuffer, real invBoxSizeZ) {
    PAR_FOR
    for (int index = 0; index < n; index++) {
        if (velm[index].w == 0)
            VBuffer[index] = 0;
        else
            VBuffer[index] = RECIP(velm[index].w) * velm[index].x * 2 * cos(2 * 3.1415926 * posq[index].z * invBoxSizeZ);
    }
}

/* ------------------------------------------------------------------ K/cosineAccelerate.cu:34-61 */
void vvo_sum_v(int n, mixed* VBuffer, double invMassTotal) {
    mixed temp = 0;
    if (g_threads <= 1) {
        for (int index = 0; index < n; index++)
            temp += VBuffer[index];
    }
    else {
        mixed t = 0;
        #pragma omp parallel for schedule(static) reduction(+:t) num_threads(g_threads)
        for (int index = 0; index < n; index++)
            t += VBuffer[index];
        temp = t;
    }
    VBuffer[0] = temp * invMassTotal;
}

/* ------------------------------------------------------------------ K/cosineAccelerate.cu:63-85 */
void vvo_remove_periodic_velocity_bias(int n, const real4* posq, mixed4* velm, const mixed* VBuffer, real invBoxSizeZ) {
    mixed V = VBuffer[0];
    PAR_FOR
    for (int index = 0; index < n; index++)
        velm[index].x -= V * cos(2 * 3.1415926 * posq[index].z * invBoxSizeZ);
}
void vvo_restore_periodic_velocity_bias(int n, const real4* posq, mixed4* velm, const mixed* VBuffer, real invBoxSizeZ) {
    mixed V = VBuffer[0];
    PAR_FOR
    for (int index = 0; index < n; index++)
        velm[index].x += V * cos(2 * 3.1415926 * posq[index].z * invBoxSizeZ);
}

/* ------------------------------------------------------------------ K/drudeLangevin.cu:2-60
 * `forceExtra[p] += mass1fract * cmForce - relForce` resolves to the float (or double)
 * scalar*real3 overload (K/vectorOps.cu:427,451): mass1fract is narrowed to `real` first. */
void vvo_add_extra_force_drude_langevin(int n_normal, int n_pairs, const mixed4* velm, real3* forceExtra,
                                        const int* normalParticles, const int2* pairParticles, mixed dragFactor,
                                        mixed randFactor, mixed dragFactorDrude, mixed randFactorDrude,
                                        const vvo_float4* random, unsigned int randomIndex) {
    PAR_FOR
    for (int i = 0; i < n_normal; i++) {
        int index = normalParticles[i];
        mixed4 velocity = velm[index];
        if (velocity.w != 0) {
            mixed mass = RECIP(velocity.w);
            mixed sqrtMass = SQRT(mass);
            vvo_float4 rand = random[randomIndex + i];
            forceExtra[index].x += (-dragFactor * mass * velocity.x + randFactor * sqrtMass * rand.x);
            forceExtra[index].y += (-dragFactor * mass * velocity.y + randFactor * sqrtMass * rand.y);
            forceExtra[index].z += (-dragFactor * mass * velocity.z + randFactor * sqrtMass * rand.z);
        }
    }
    randomIndex += n_normal;
    /* pairs are independent unless a particle appears in two pairs (never: HOST:788-792) */
    PAR_FOR
    for (int i = 0; i < n_pairs; i++) {
        int2 particles = pairParticles[i];
        mixed4 velocity1 = velm[particles.x];
        mixed4 velocity2 = velm[particles.y];
        mixed mass1 = RECIP(velocity1.w);
        mixed mass2 = RECIP(velocity2.w);
        mixed totMass = mass1 + mass2;
        mixed sqrtTotMass = SQRT(totMass);
        mixed redMass = RECIP((mass1 + mass2) * velocity1.w * velocity2.w);
        mixed sqrtRedMass = SQRT(redMass);
        mixed invTotMass = RECIP(totMass);
        mixed mass1fract = invTotMass * mass1;
        mixed mass2fract = invTotMass * mass2;
        mixed cmx = velocity1.x * mass1fract + velocity2.x * mass2fract;
        mixed cmy = velocity1.y * mass1fract + velocity2.y * mass2fract;
        mixed cmz = velocity1.z * mass1fract + velocity2.z * mass2fract;
        mixed rx = velocity2.x - velocity1.x, ry = velocity2.y - velocity1.y, rz = velocity2.z - velocity1.z;
        real3 cmForce, relForce;
        vvo_float4 rand1 = random[randomIndex + 2 * i];
        vvo_float4 rand2 = random[randomIndex + 2 * i + 1];
        cmForce.x = (-dragFactor * totMass * cmx + randFactor * sqrtTotMass * rand1.x);
        cmForce.y = (-dragFactor * totMass * cmy + randFactor * sqrtTotMass * rand1.y);
        cmForce.z = (-dragFactor * totMass * cmz + randFactor * sqrtTotMass * rand1.z);
        relForce.x = (-dragFactorDrude * redMass * rx + randFactorDrude * sqrtRedMass * rand2.x);
        relForce.y = (-dragFactorDrude * redMass * ry + randFactorDrude * sqrtRedMass * rand2.y);
        relForce.z = (-dragFactorDrude * redMass * rz + randFactorDrude * sqrtRedMass * rand2.z);
        real m1f = (real) mass1fract, m2f = (real) mass2fract;
        real3 f1 = { m1f * cmForce.x - relForce.x, m1f * cmForce.y - relForce.y, m1f * cmForce.z - relForce.z };
        real3 f2 = { m2f * cmForce.x + relForce.x, m2f * cmForce.y + relForce.y, m2f * cmForce.z + relForce.z };
        forceExtra[particles.x].x += f1.x; forceExtra[particles.x].y += f1.y; forceExtra[particles.x].z += f1.z;
        forceExtra[particles.y].x += f2.x; forceExtra[particles.y].y += f2.y; forceExtra[particles.y].z += f2.z;
    }
}

/* ------------------------------------------------------------------ K/electricField.cu:2-12 */
void vvo_add_extra_force_electric_field(int n_el, const real4* posq, real3* forceExtra, const int* particlesElectrolyte,
                                        real efscale) {
    PAR_FOR
    for (int i = 0; i < n_el; i++) {
        int index = particlesElectrolyte[i];
        real charge = posq[index].w;
        forceExtra[index].z += efscale * charge;
    }
}

/* ------------------------------------------------------------------ K/imageCharge.cu:2-28
 * x and y are plain copies (bit-exact); the reference dereferences posqCorrection even
 * when the host passed 0 (quirk Q5) -- here a NULL posq_corr is simply skipped. */
void vvo_update_image_positions(int n_img, real4* posq, real4* posq_corr, const int2* imagePairs, mixed mirror) {
    PAR_FOR
    for (int i = 0; i < n_img; i++) {
        int2 pair = imagePairs[i];
        int index_img = pair.x;
        int index_par = pair.y;
        posq[index_img].x = posq[index_par].x;
        posq[index_img].y = posq[index_par].y;
        if (posq_corr) {
            posq_corr[index_img].x = posq_corr[index_par].x;
            posq_corr[index_img].y = posq_corr[index_par].y;
        }
#ifdef USE_MIXED_PRECISION
        mixed z = (mixed) posq[index_par].z + (mixed) posq_corr[index_par].z;
        z = mirror * 2 - z;
        posq[index_img].z = (real) z;
        posq_corr[index_img].z = (real) (z - (real) z);
#else
        posq[index_img].z = 2 * mirror - posq[index_par].z;
#endif
    }
}

/* ------------------------------------------------------------------ OpenMM-style SHAKE clusters (see vv_oracle.h)
 * One central particle i, up to three peripherals j of equal mass and distance; Gauss-Seidel sweeps (<= 15) until every
 * constraint of the cluster is within tol (relative, on d^2) resp. every velocity correction is below tol. */
void vvo_shake_positions(int nclusters, const int* atoms, const float* params, mixed tol, const real4* posq,
                         const real4* posq_corr, mixed4* pos_delta) {
    PAR_FOR
    for (int c = 0; c < nclusters; c++) {
        const int ic = atoms[4 * c];
        const mixed invMassCentral = params[4 * c], avgMass = params[4 * c + 1], d2 = params[4 * c + 2], invMassPeripheral = params[4 * c + 3];
        mixed x, y, z, w;
        load_pos(posq, posq_corr, ic, &x, &y, &z, &w);
        mixed rij[3][3], rijsq[3], ld[3], xpj[3][3];
        int np = 0;
        for (int k = 0; k < 3; k++) {
            const int j = atoms[4 * c + 1 + k];
            if (j < 0) break;
            mixed px, py, pz, pw;
            load_pos(posq, posq_corr, j, &px, &py, &pz, &pw);
            rij[k][0] = x - px; rij[k][1] = y - py; rij[k][2] = z - pz;
            xpj[k][0] = pos_delta[j].x; xpj[k][1] = pos_delta[j].y; xpj[k][2] = pos_delta[j].z;
            rijsq[k] = rij[k][0] * rij[k][0] + rij[k][1] * rij[k][1] + rij[k][2] * rij[k][2];
            ld[k] = d2 - rijsq[k];
            np++;
        }
        mixed xpi[3] = { pos_delta[ic].x, pos_delta[ic].y, pos_delta[ic].z };
        const mixed d2tol = d2 * tol;
        int converged = 0;
        for (int iteration = 0; iteration < 15 && !converged; iteration++) {
            converged = 1;
            for (int k = 0; k < np; k++) {
                const mixed rp0 = xpi[0] - xpj[k][0], rp1 = xpi[1] - xpj[k][1], rp2 = xpi[2] - xpj[k][2];
                const mixed rpsqij = rp0 * rp0 + rp1 * rp1 + rp2 * rp2;
                const mixed rrpr = rij[k][0] * rp0 + rij[k][1] * rp1 + rij[k][2] * rp2;
                const mixed num = ld[k] - 2.0f * rrpr - rpsqij;       /* convergence test in product form: |num| >= d2*tol */
                if (fabs(num) >= d2tol) {
                    const mixed acor = num * avgMass / (rrpr + rijsq[k]);
                    const mixed d0 = rij[k][0] * acor, d1 = rij[k][1] * acor, d2v = rij[k][2] * acor;
                    xpi[0] += d0 * invMassCentral; xpi[1] += d1 * invMassCentral; xpi[2] += d2v * invMassCentral;
                    xpj[k][0] -= d0 * invMassPeripheral; xpj[k][1] -= d1 * invMassPeripheral; xpj[k][2] -= d2v * invMassPeripheral;
                    converged = 0;
                }
            }
        }
        pos_delta[ic].x = xpi[0]; pos_delta[ic].y = xpi[1]; pos_delta[ic].z = xpi[2];
        for (int k = 0; k < np; k++) {
            const int j = atoms[4 * c + 1 + k];
            pos_delta[j].x = xpj[k][0]; pos_delta[j].y = xpj[k][1]; pos_delta[j].z = xpj[k][2];
        }
    }
}

void vvo_shake_velocities(int nclusters, const int* atoms, const float* params, mixed tol, const real4* posq,
                          const real4* posq_corr, mixed4* velm) {
    PAR_FOR
    for (int c = 0; c < nclusters; c++) {
        const int ic = atoms[4 * c];
        const mixed invMassCentral = params[4 * c], avgMass = params[4 * c + 1], invMassPeripheral = params[4 * c + 3];
        mixed x, y, z, w;
        load_pos(posq, posq_corr, ic, &x, &y, &z, &w);
        mixed rij[3][3], rijsq[3], vj[3][3];
        int np = 0;
        for (int k = 0; k < 3; k++) {
            const int j = atoms[4 * c + 1 + k];
            if (j < 0) break;
            mixed px, py, pz, pw;
            load_pos(posq, posq_corr, j, &px, &py, &pz, &pw);
            rij[k][0] = x - px; rij[k][1] = y - py; rij[k][2] = z - pz;
            vj[k][0] = velm[j].x; vj[k][1] = velm[j].y; vj[k][2] = velm[j].z;
            rijsq[k] = 1 / (rij[k][0] * rij[k][0] + rij[k][1] * rij[k][1] + rij[k][2] * rij[k][2]);   /* reciprocal once per bond */
            np++;
        }
        mixed vi[3] = { velm[ic].x, velm[ic].y, velm[ic].z };
        int converged = 0;
        for (int iteration = 0; iteration < 15 && !converged; iteration++) {
            converged = 1;
            for (int k = 0; k < np; k++) {
                const mixed rp0 = vi[0] - vj[k][0], rp1 = vi[1] - vj[k][1], rp2 = vi[2] - vj[k][2];
                const mixed rrpr = rp0 * rij[k][0] + rp1 * rij[k][1] + rp2 * rij[k][2];
                const mixed delta = -2.0f * avgMass * rrpr * rijsq[k];
                const mixed d0 = rij[k][0] * delta, d1 = rij[k][1] * delta, d2v = rij[k][2] * delta;
                vi[0] += d0 * invMassCentral; vi[1] += d1 * invMassCentral; vi[2] += d2v * invMassCentral;
                vj[k][0] -= d0 * invMassPeripheral; vj[k][1] -= d1 * invMassPeripheral; vj[k][2] -= d2v * invMassPeripheral;
                if (fabs(delta) > tol) converged = 0;
            }
        }
        velm[ic].x = vi[0]; velm[ic].y = vi[1]; velm[ic].z = vi[2];
        for (int k = 0; k < np; k++) {
            const int j = atoms[4 * c + 1 + k];
            velm[j].x = vj[k][0]; velm[j].y = vj[k][1]; velm[j].z = vj[k][2];
        }
    }
}

/* ------------------------------------------------------------------ general constraint clusters (ours; the product: vv_device.inc general_*)
 * Any constraint topology (AllBonds, HAngles: chains, rings, triangles -- what OpenMM hands to CCMA between the reference's launches,
 * HOST:151,176,351,427).  Coloured Gauss-Seidel: the constraints arrive sorted by colour (constraints of one colour share no particle,
 * so their order inside a colour does not matter -- the product relaxes them side by side); a sweep visits every constraint once with
 * OpenMM's SHAKE update (as vvo_shake_positions / vvo_shake_velocities), a constraint inside its tolerance is left alone, and sweeps
 * repeat until nothing moved (<= 150).  The product sweeps wave by wave until THAT wave's constraints rest: the same values, because a
 * resting constraint is not touched by further sweeps. */
void vvo_general_positions(int n, const int* atoms, const float* params, mixed tol, mixed omega, const real4* posq, const real4* posq_corr, mixed4* pos_delta) {
    for (int iteration = 0; iteration < 150; iteration++) {
        int moved = 0;
        for (int k = 0; k < n; k++) {
            const int a = atoms[2 * k], b = atoms[2 * k + 1];
            const mixed d2 = params[4 * k], avgMass = params[4 * k + 1], ima = params[4 * k + 2], imb = params[4 * k + 3], d2tol = d2 * tol;
            mixed ax, ay, az, aw, bx, by, bz, bw;
            load_pos(posq, posq_corr, a, &ax, &ay, &az, &aw);
            load_pos(posq, posq_corr, b, &bx, &by, &bz, &bw);
            const mixed r0 = ax - bx, r1 = ay - by, r2 = az - bz;
            const mixed rsq = r0 * r0 + r1 * r1 + r2 * r2, ld = d2 - rsq;
            const mixed rp0 = pos_delta[a].x - pos_delta[b].x, rp1 = pos_delta[a].y - pos_delta[b].y, rp2 = pos_delta[a].z - pos_delta[b].z;
            const mixed rpsq = rp0 * rp0 + rp1 * rp1 + rp2 * rp2;
            const mixed rrpr = r0 * rp0 + r1 * rp1 + r2 * rp2;
            const mixed num = ld - 2.0f * rrpr - rpsq;
            if (fabs(num) >= d2tol) {
                const mixed acor = omega * (num * avgMass / (rrpr + rsq));      /* omega: successive over-relaxation of the sweep, see vvo_system */
                const mixed e0 = r0 * acor, e1 = r1 * acor, e2 = r2 * acor;
                pos_delta[a].x = pos_delta[a].x + e0 * ima; pos_delta[a].y = pos_delta[a].y + e1 * ima; pos_delta[a].z = pos_delta[a].z + e2 * ima;
                pos_delta[b].x = pos_delta[b].x - e0 * imb; pos_delta[b].y = pos_delta[b].y - e1 * imb; pos_delta[b].z = pos_delta[b].z - e2 * imb;
                moved = 1;
            }
        }
        if (!moved) break;
    }
}
void vvo_general_velocities(int n, const int* atoms, const float* params, mixed tol, mixed omega, const real4* posq, const real4* posq_corr, mixed4* velm) {
    for (int iteration = 0; iteration < 150; iteration++) {
        int moved = 0;
        for (int k = 0; k < n; k++) {
            const int a = atoms[2 * k], b = atoms[2 * k + 1];
            const mixed avgMass = params[4 * k + 1], ima = params[4 * k + 2], imb = params[4 * k + 3];
            mixed ax, ay, az, aw, bx, by, bz, bw;
            load_pos(posq, posq_corr, a, &ax, &ay, &az, &aw);
            load_pos(posq, posq_corr, b, &bx, &by, &bz, &bw);
            const mixed r0 = ax - bx, r1 = ay - by, r2 = az - bz;
            const mixed rinv = (mixed) 1 / (r0 * r0 + r1 * r1 + r2 * r2);
            const mixed rrpr = (velm[a].x - velm[b].x) * r0 + (velm[a].y - velm[b].y) * r1 + (velm[a].z - velm[b].z) * r2;
            const mixed delta = -2.0f * avgMass * rrpr * rinv;
            if (fabs(delta) > tol) {
                const mixed od = omega * delta;
                const mixed e0 = r0 * od, e1 = r1 * od, e2 = r2 * od;
                velm[a].x = velm[a].x + e0 * ima; velm[a].y = velm[a].y + e1 * ima; velm[a].z = velm[a].z + e2 * ima;
                velm[b].x = velm[b].x - e0 * imb; velm[b].y = velm[b].y - e1 * imb; velm[b].z = velm[b].z - e2 * imb;
                moved = 1;
            }
        }
        if (!moved) break;
    }
}

/* ------------------------------------------------------------------ integration.computeVirtualSites(), HOST:214, 374
 * OpenMM's kernel is not under /root/reference (third-party arithmetic on the path).  Restated from the definitions OpenMM
 * documents for its four site classes -- TwoParticleAverageSite: w1 r1 + w2 r2; ThreeParticleAverageSite: w1 r1 + w2 r2 + w3 r3;
 * OutOfPlaneSite: r1 + w12 r12 + w13 r13 + wCross (r12 x r13); LocalCoordinatesSite: origin and two directions as weighted sums of the
 * parents, z = x cross y, x and z normalised, y = z cross x, site = origin + local position in that frame -- with the parents read
 * back from posq (+ posqCorrection) as every OpenMM kernel reads positions, weights in `real`, arithmetic in `mixed`, the site's
 * charge kept.  Parity with OpenMM's own kernel: unpinned. */
void vvo_compute_virtual_sites(int n, const int* sites, const double* params, real4* posq, real4* posq_corr) {
    for (int k = 0; k < n; k++) {
        const int site = sites[5 * k], kind = sites[5 * k + 1], a1 = sites[5 * k + 2], a2 = sites[5 * k + 3], a3 = kind == 0 ? a1 : sites[5 * k + 4];
        real w[12];
        for (int j = 0; j < 12; j++) w[j] = (real) params[12 * k + j];
        mixed x, y, z, q, p1x, p1y, p1z, p2x, p2y, p2z, p3x, p3y, p3z, unused;
        load_pos(posq, posq_corr, site, &x, &y, &z, &q);
        load_pos(posq, posq_corr, a1, &p1x, &p1y, &p1z, &unused);
        load_pos(posq, posq_corr, a2, &p2x, &p2y, &p2z, &unused);
        load_pos(posq, posq_corr, a3, &p3x, &p3y, &p3z, &unused);
        if (kind == 0) {
            x = p1x * w[0] + p2x * w[1]; y = p1y * w[0] + p2y * w[1]; z = p1z * w[0] + p2z * w[1];
        } else if (kind == 1) {
            x = p1x * w[0] + p2x * w[1] + p3x * w[2]; y = p1y * w[0] + p2y * w[1] + p3y * w[2]; z = p1z * w[0] + p2z * w[1] + p3z * w[2];
        } else if (kind == 2) {
            const mixed ax = p2x - p1x, ay = p2y - p1y, az = p2z - p1z, bx = p3x - p1x, by = p3y - p1y, bz = p3z - p1z;
            const mixed cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
            x = p1x + ax * w[0] + bx * w[1] + cx * w[2]; y = p1y + ay * w[0] + by * w[1] + cy * w[2]; z = p1z + az * w[0] + bz * w[1] + cz * w[2];
        } else {
            const mixed ox = p1x * w[0] + p2x * w[1] + p3x * w[2], oy = p1y * w[0] + p2y * w[1] + p3y * w[2], oz = p1z * w[0] + p2z * w[1] + p3z * w[2];
            mixed xx = p1x * w[3] + p2x * w[4] + p3x * w[5], xy = p1y * w[3] + p2y * w[4] + p3y * w[5], xz = p1z * w[3] + p2z * w[4] + p3z * w[5];
            mixed yx = p1x * w[6] + p2x * w[7] + p3x * w[8], yy = p1y * w[6] + p2y * w[7] + p3y * w[8], yz = p1z * w[6] + p2z * w[7] + p3z * w[8];
            mixed zx = xy * yz - xz * yy, zy = xz * yx - xx * yz, zz = xx * yy - xy * yx;
#ifdef VVO_SINGLE          /* the square root of the `mixed` type */
            const mixed normX = sqrtf(xx * xx + xy * xy + xz * xz), normZ = sqrtf(zx * zx + zy * zy + zz * zz);
#else
            const mixed normX = sqrt(xx * xx + xy * xy + xz * xz), normZ = sqrt(zx * zx + zy * zy + zz * zz);
#endif
            const mixed invX = normX > 0 ? (mixed) 1 / normX : (mixed) 0, invZ = normZ > 0 ? (mixed) 1 / normZ : (mixed) 0;
            xx *= invX; xy *= invX; xz *= invX;
            zx *= invZ; zy *= invZ; zz *= invZ;
            yx = zy * xz - zz * xy; yy = zz * xx - zx * xz; yz = zx * xy - zy * xx;
            x = ox + xx * w[9] + yx * w[10] + zx * w[11]; y = oy + xy * w[9] + yy * w[10] + zy * w[11]; z = oz + xz * w[9] + yz * w[10] + zz * w[11];
        }
        store_pos(posq, posq_corr, site, x, y, z, q);
    }
}

/* ------------------------------------------------------------------ the same clusters, all constraints of a cluster at once
 * (shake_mode 1, what the product runs by default; the Gauss-Seidel sweeps above stay as shake_mode 0).
 * Multipliers l_k, one per constraint: the central particle moves by imc * sum_m l_m r_m, peripheral k by -imp * l_k r_k, with
 * r_k = x_central - x_k the bond BEFORE the step (OpenMM's rij) and imc / imp the float inverse masses of the cluster parameters.
 *   velocities: (u_k + imc sum_m l_m r_m + imp l_k r_k) . r_k = 0 is LINEAR in l: one symmetric k x k system (k <= 3), solved in
 *               closed form (cofactors); no iteration, no tolerance.
 *   positions:  g_k(l) = |s_k + imc sum_m l_m r_m + imp l_k r_k|^2 - d^2 = 0, s_k the unconstrained new bond.  Newton on the k x k
 *               system with the exact diagonal (imc + imp) b_k . r_k (b_k the current bond: OpenMM's rrpr + rijsq) and the
 *               off-diagonals imc r_k . r_m taken at the old bonds (they differ from imc b_k . r_m by the bond's rotation during one
 *               step, a few percent of an entry that is itself 3 % of the diagonal for C-H clusters): one corrective iteration
 *               where the sweeps need four.  Same convergence test as OpenMM's: |d^2 - |b_k|^2| < tol d^2 for every constraint.
 * Unused rows (np < 3) are identity rows.  Every product, sum and fused multiply-add in the order written: the device code repeats them verbatim. */
/* Explicit fused multiply-adds, the same calls in the same places as the device code (vv_device.inc: vfma): C's fma / fmaf is the one
 * correctly rounded operation v_fma_f64 / v_fma_f32 is, so the two agree bit for bit although everything else is built without contraction. */
#if defined(VVO_SINGLE)
#define FMA(a, b, c) fmaf((a), (b), (c))
#else
#define FMA(a, b, c) fma((a), (b), (c))
#endif
#define DOT3(a, b) FMA((a)[2], (b)[2], FMA((a)[1], (b)[1], (a)[0] * (b)[0]))
typedef struct { mixed c00, c01, c02, c11, c12, c22, inv; } sym3inv;
static inline sym3inv sym3_cofactors(mixed A00, mixed A01, mixed A02, mixed A11, mixed A12, mixed A22) {
    sym3inv q;
    q.c00 = FMA(A11, A22, -(A12 * A12));
    q.c01 = FMA(A02, A12, -(A01 * A22));
    q.c02 = FMA(A01, A12, -(A02 * A11));
    q.c11 = FMA(A00, A22, -(A02 * A02));
    q.c12 = FMA(A01, A02, -(A00 * A12));
    q.c22 = FMA(A00, A11, -(A01 * A01));
    const mixed det = FMA(A02, q.c02, FMA(A01, q.c01, A00 * q.c00));
    q.inv = 1 / det;
    return q;
}
void vvo_cluster_velocities_direct(int nclusters, const int* atoms, const float* params, const real4* posq, const real4* posq_corr, mixed4* velm) {
    PAR_FOR
    for (int c = 0; c < nclusters; c++) {
        const int ic = atoms[4 * c];
        const mixed imc = params[4 * c], imp = params[4 * c + 3];
        mixed xc[3], w;
        load_pos(posq, posq_corr, ic, &xc[0], &xc[1], &xc[2], &w);
        const mixed vc[3] = { velm[ic].x, velm[ic].y, velm[ic].z };
        mixed r[3][3], b[3];
        int np = 0;
        for (int k = 0; k < 3; k++) {
            const int j = atoms[4 * c + 1 + k];
            if (j >= 0 && np == k) {
                mixed xp[3];
                load_pos(posq, posq_corr, j, &xp[0], &xp[1], &xp[2], &w);
                const mixed u[3] = { vc[0] - velm[j].x, vc[1] - velm[j].y, vc[2] - velm[j].z };
                for (int a = 0; a < 3; a++) r[k][a] = xc[a] - xp[a];
                b[k] = DOT3(u, r[k]);
                np++;
            } else {
                r[k][0] = r[k][1] = r[k][2] = 0; b[k] = 0;
            }
        }
        const mixed ims = imc + imp;
        const mixed A00 = ims * DOT3(r[0], r[0]);
        const mixed A11 = np > 1 ? ims * DOT3(r[1], r[1]) : (mixed) 1;
        const mixed A22 = np > 2 ? ims * DOT3(r[2], r[2]) : (mixed) 1;
        const mixed A01 = imc * DOT3(r[0], r[1]), A02 = imc * DOT3(r[0], r[2]), A12 = imc * DOT3(r[1], r[2]);
        const sym3inv q = sym3_cofactors(A00, A01, A02, A11, A12, A22);
        mixed l[3];
        l[0] = -(FMA(q.c02, b[2], FMA(q.c01, b[1], q.c00 * b[0])) * q.inv);
        l[1] = -(FMA(q.c12, b[2], FMA(q.c11, b[1], q.c01 * b[0])) * q.inv);
        l[2] = -(FMA(q.c22, b[2], FMA(q.c12, b[1], q.c02 * b[0])) * q.inv);
        mixed t[3];
        for (int a = 0; a < 3; a++) t[a] = FMA(l[2], r[2][a], FMA(l[1], r[1][a], l[0] * r[0][a]));
        velm[ic].x = FMA(imc, t[0], velm[ic].x); velm[ic].y = FMA(imc, t[1], velm[ic].y); velm[ic].z = FMA(imc, t[2], velm[ic].z);
        for (int k = 0; k < np; k++) {
            const int j = atoms[4 * c + 1 + k];
            const mixed f = imp * l[k];
            velm[j].x = FMA(-f, r[k][0], velm[j].x); velm[j].y = FMA(-f, r[k][1], velm[j].y); velm[j].z = FMA(-f, r[k][2], velm[j].z);
        }
    }
}
void vvo_cluster_positions_newton(int nclusters, const int* atoms, const float* params, mixed tol, const real4* posq,
                                  const real4* posq_corr, mixed4* pos_delta) {
    PAR_FOR
    for (int c = 0; c < nclusters; c++) {
        const int ic = atoms[4 * c];
        const mixed imc = params[4 * c], d2 = params[4 * c + 2], imp = params[4 * c + 3];
        mixed xc[3], w;
        load_pos(posq, posq_corr, ic, &xc[0], &xc[1], &xc[2], &w);
        const mixed dc[3] = { pos_delta[ic].x, pos_delta[ic].y, pos_delta[ic].z };
        mixed r[3][3], s[3][3], b[3][3];
        int np = 0;
        for (int k = 0; k < 3; k++) {
            const int j = atoms[4 * c + 1 + k];
            if (j >= 0 && np == k) {
                mixed xp[3];
                load_pos(posq, posq_corr, j, &xp[0], &xp[1], &xp[2], &w);
                const mixed dk[3] = { pos_delta[j].x, pos_delta[j].y, pos_delta[j].z };
                for (int a = 0; a < 3; a++) { r[k][a] = xc[a] - xp[a]; s[k][a] = r[k][a] + (dc[a] - dk[a]); }
                np++;
            } else {
                for (int a = 0; a < 3; a++) r[k][a] = s[k][a] = 0;
            }
            for (int a = 0; a < 3; a++) b[k][a] = s[k][a];
        }
        const mixed ims = imc + imp, d2tol = d2 * tol;
        const mixed O01 = imc * DOT3(r[0], r[1]), O02 = imc * DOT3(r[0], r[2]), O12 = imc * DOT3(r[1], r[2]);
        mixed l[3] = { 0, 0, 0 }, t[3] = { 0, 0, 0 };
        for (int iteration = 0; iteration < 15; iteration++) {
            mixed g[3];
            int active = 0;
            for (int k = 0; k < 3; k++) {
                g[k] = k < np ? DOT3(b[k], b[k]) - d2 : (mixed) 0;
                if (fabs(g[k]) >= d2tol) active = 1;
            }
            if (!active) break;
            const mixed D0 = ims * DOT3(b[0], r[0]);
            const mixed D1 = np > 1 ? ims * DOT3(b[1], r[1]) : (mixed) 1;
            const mixed D2 = np > 2 ? ims * DOT3(b[2], r[2]) : (mixed) 1;
            const sym3inv q = sym3_cofactors(D0, O01, O02, D1, O12, D2);
            const mixed h0 = 0.5f * g[0], h1 = 0.5f * g[1], h2 = 0.5f * g[2];
            l[0] = FMA(-FMA(q.c02, h2, FMA(q.c01, h1, q.c00 * h0)), q.inv, l[0]);
            l[1] = FMA(-FMA(q.c12, h2, FMA(q.c11, h1, q.c01 * h0)), q.inv, l[1]);
            l[2] = FMA(-FMA(q.c22, h2, FMA(q.c12, h1, q.c02 * h0)), q.inv, l[2]);
            for (int a = 0; a < 3; a++) t[a] = imc * FMA(l[2], r[2][a], FMA(l[1], r[1][a], l[0] * r[0][a]));
            for (int k = 0; k < np; k++) {
                const mixed f = imp * l[k];
                for (int a = 0; a < 3; a++) b[k][a] = FMA(f, r[k][a], s[k][a] + t[a]);
            }
        }
        pos_delta[ic].x = dc[0] + t[0]; pos_delta[ic].y = dc[1] + t[1]; pos_delta[ic].z = dc[2] + t[2];
        for (int k = 0; k < np; k++) {
            const int j = atoms[4 * c + 1 + k];
            const mixed f = imp * l[k];
            pos_delta[j].x = FMA(-f, r[k][0], pos_delta[j].x); pos_delta[j].y = FMA(-f, r[k][1], pos_delta[j].y); pos_delta[j].z = FMA(-f, r[k][2], pos_delta[j].z);
        }
    }
}

/* ------------------------------------------------------------------ SETTLE for rigid three-site molecules (see vv_oracle.h) */
typedef struct { mixed x, y, z; } v3;
static inline v3 v3_make(mixed x, mixed y, mixed z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, mixed s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline mixed v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) { return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 v3_unit(v3 a) { return v3_scale(a, 1 / sqrt(v3_dot(a, a))); }
static inline v3 v3_pos(const real4* posq, const real4* corr, int i) { mixed x, y, z, w; load_pos(posq, corr, i, &x, &y, &z, &w); return v3_make(x, y, z); }

void vvo_settle_positions(int n, const int* atoms, const float* params, const real4* posq, const real4* posq_corr,
                          const mixed4* velm, mixed4* pos_delta) {
    PAR_FOR
    for (int c = 0; c < n; c++) {
        const int ia = atoms[3 * c], ib = atoms[3 * c + 1], ic = atoms[3 * c + 2];
        const mixed dAB = params[2 * c], dBB = params[2 * c + 1];
        const mixed mA = 1 / velm[ia].w, mB = 1 / velm[ib].w;
        const mixed invM = 1 / (mA + mB + mB);
        const v3 a0 = v3_pos(posq, posq_corr, ia);
        const v3 b0 = v3_sub(v3_pos(posq, posq_corr, ib), a0), c0 = v3_sub(v3_pos(posq, posq_corr, ic), a0);
        const v3 da = v3_make(pos_delta[ia].x, pos_delta[ia].y, pos_delta[ia].z);
        const v3 db = v3_make(pos_delta[ib].x, pos_delta[ib].y, pos_delta[ib].z);
        const v3 dc = v3_make(pos_delta[ic].x, pos_delta[ic].y, pos_delta[ic].z);
        /* unconstrained new positions relative to the old apex, and their centre of mass */
        const v3 pa = da, pb = v3_add(b0, db), pc = v3_add(c0, dc);
        const v3 com = v3_scale(v3_add(v3_scale(pa, mA), v3_scale(v3_add(pb, pc), mB)), invM);
        const v3 a1 = v3_sub(pa, com), b1 = v3_sub(pb, com), c1 = v3_sub(pc, com);
        /* frame: z normal to the OLD triangle, x normal to z and to the new apex vector, y completes it */
        const v3 zax = v3_cross(b0, c0), xax = v3_cross(a1, zax), yax = v3_cross(zax, xax);
        const v3 ex = v3_unit(xax), ey = v3_unit(yax), ez = v3_unit(zax);
        const mixed xb0d = v3_dot(ex, b0), yb0d = v3_dot(ey, b0), xc0d = v3_dot(ex, c0), yc0d = v3_dot(ey, c0);
        const mixed za1d = v3_dot(ez, a1);
        const mixed xb1d = v3_dot(ex, b1), yb1d = v3_dot(ey, b1), zb1d = v3_dot(ez, b1);
        const mixed xc1d = v3_dot(ex, c1), yc1d = v3_dot(ey, c1), zc1d = v3_dot(ez, c1);
        /* canonical triangle: apex at (0, ra), partners at (-+rc, -rb) */
        const mixed rc = dBB / 2;
        const mixed height = sqrt(dAB * dAB - rc * rc);
        const mixed ra = height * (mB + mB) * invM, rb = height - ra;
        /* tilt angles phi, psi from the z displacements */
        const mixed sinphi = za1d / ra, cosphi = sqrt(1 - sinphi * sinphi);
        const mixed sinpsi = (zb1d - zc1d) / (2 * rc * cosphi), cospsi = sqrt(1 - sinpsi * sinpsi);
        const mixed ya2d = ra * cosphi;
        mixed xb2d = -rc * cospsi;
        const mixed yb2d = -rb * cosphi - rc * sinpsi * sinphi, yc2d = -rb * cosphi + rc * sinpsi * sinphi;
        const mixed hh2 = 4 * xb2d * xb2d + (yb2d - yc2d) * (yb2d - yc2d) + (zb1d - zc1d) * (zb1d - zc1d);
        xb2d -= (2 * xb2d + sqrt(4 * xb2d * xb2d - hh2 + dBB * dBB)) / 2;
        /* in-plane rotation theta from the angular-momentum condition */
        const mixed alpha = xb2d * (xb0d - xc0d) + yb0d * yb2d + yc0d * yc2d;
        const mixed beta = xb2d * (yc0d - yb0d) + xb0d * yb2d + xc0d * yc2d;
        const mixed gamma = xb0d * yb1d - xb1d * yb0d + xc0d * yc1d - xc1d * yc0d;
        const mixed ab2 = alpha * alpha + beta * beta;
        const mixed sintheta = (alpha * gamma - beta * sqrt(ab2 - gamma * gamma)) / ab2, costheta = sqrt(1 - sintheta * sintheta);
        const v3 a3d = v3_make(-ya2d * sintheta, ya2d * costheta, za1d);
        const v3 b3d = v3_make(xb2d * costheta - yb2d * sintheta, xb2d * sintheta + yb2d * costheta, zb1d);
        const v3 c3d = v3_make(-xb2d * costheta - yc2d * sintheta, -xb2d * sintheta + yc2d * costheta, zc1d);
        const v3 a3 = v3_add(v3_add(v3_scale(ex, a3d.x), v3_scale(ey, a3d.y)), v3_scale(ez, a3d.z));
        const v3 b3 = v3_add(v3_add(v3_scale(ex, b3d.x), v3_scale(ey, b3d.y)), v3_scale(ez, b3d.z));
        const v3 c3 = v3_add(v3_add(v3_scale(ex, c3d.x), v3_scale(ey, c3d.y)), v3_scale(ez, c3d.z));
        const v3 na = v3_add(com, a3), nb = v3_sub(v3_add(com, b3), b0), nc = v3_sub(v3_add(com, c3), c0);
        pos_delta[ia].x = na.x; pos_delta[ia].y = na.y; pos_delta[ia].z = na.z;
        pos_delta[ib].x = nb.x; pos_delta[ib].y = nb.y; pos_delta[ib].z = nb.z;
        pos_delta[ic].x = nc.x; pos_delta[ic].y = nc.y; pos_delta[ic].z = nc.z;
    }
}

void vvo_settle_velocities(int n, const int* atoms, const real4* posq, const real4* posq_corr, mixed4* velm) {
    PAR_FOR
    for (int c = 0; c < n; c++) {
        const int ia = atoms[3 * c], ib = atoms[3 * c + 1], ic = atoms[3 * c + 2];
        const mixed wA = velm[ia].w, wB = velm[ib].w, wC = velm[ic].w;                 /* inverse masses */
        const v3 pa = v3_pos(posq, posq_corr, ia), pb = v3_pos(posq, posq_corr, ib), pc = v3_pos(posq, posq_corr, ic);
        const v3 eAB = v3_unit(v3_sub(pb, pa)), eBC = v3_unit(v3_sub(pc, pb)), eCA = v3_unit(v3_sub(pa, pc));
        v3 va = v3_make(velm[ia].x, velm[ia].y, velm[ia].z), vb = v3_make(velm[ib].x, velm[ib].y, velm[ib].z), vc = v3_make(velm[ic].x, velm[ic].y, velm[ic].z);
        const mixed rAB = v3_dot(v3_sub(vb, va), eAB), rBC = v3_dot(v3_sub(vc, vb), eBC), rCA = v3_dot(v3_sub(va, vc), eCA);
        const mixed cA = -v3_dot(eAB, eCA), cB = -v3_dot(eAB, eBC), cC = -v3_dot(eBC, eCA);
        /* [ wA+wB  cB wB  cA wA ] [tab]   [rAB]
         * [ cB wB  wB+wC  cC wC ] [tbc] = [rBC]      (symmetric; solved by Cramer's rule)
         * [ cA wA  cC wC  wC+wA ] [tca]   [rCA] */
        const mixed m11 = wA + wB, m12 = cB * wB, m13 = cA * wA, m22 = wB + wC, m23 = cC * wC, m33 = wC + wA;
        const mixed det = m11 * (m22 * m33 - m23 * m23) - m12 * (m12 * m33 - m23 * m13) + m13 * (m12 * m23 - m22 * m13);
        const mixed tab = (rAB * (m22 * m33 - m23 * m23) - m12 * (rBC * m33 - m23 * rCA) + m13 * (rBC * m23 - m22 * rCA)) / det;
        const mixed tbc = (m11 * (rBC * m33 - m23 * rCA) - rAB * (m12 * m33 - m23 * m13) + m13 * (m12 * rCA - rBC * m13)) / det;
        const mixed tca = (m11 * (m22 * rCA - rBC * m23) - m12 * (m12 * rCA - rBC * m13) + rAB * (m12 * m23 - m22 * m13)) / det;
        va = v3_add(va, v3_scale(v3_sub(v3_scale(eAB, tab), v3_scale(eCA, tca)), wA));
        vb = v3_add(vb, v3_scale(v3_sub(v3_scale(eBC, tbc), v3_scale(eAB, tab)), wB));
        vc = v3_add(vc, v3_scale(v3_sub(v3_scale(eCA, tca), v3_scale(eBC, tbc)), wC));
        velm[ia].x = va.x; velm[ia].y = va.y; velm[ia].z = va.z;
        velm[ib].x = vb.x; velm[ib].y = vb.y; velm[ib].z = vb.z;
        velm[ic].x = vc.x; velm[ic].y = vc.y; velm[ic].z = vc.z;
    }
}

/* ------------------------------------------------------------------ API:340-376 (host, double) */
void vvo_propagate_nh_chain(int numNHChains, int loopsPerStep, double stepSize, double* eta, double* eta_dot,
                            double* eta_dotdot, const double* eta_mass, double ke2, double ke2_target,
                            double t_target, double* factor_out) {
    double expfac = 1.0;
    double dt2 = stepSize / loopsPerStep / 2;
    double dt4 = dt2 / 2;
    double dt8 = dt4 / 2;
    double factor = 1.0;
    eta_dotdot[0] = (ke2 - ke2_target) / eta_mass[0];
    for (int iloop = 0; iloop < loopsPerStep; iloop++) {
        for (int ich = numNHChains - 1; ich >= 0; ich--) {
            expfac = exp(-dt8 * eta_dot[ich + 1]);
            eta_dot[ich] *= expfac;
            eta_dot[ich] += eta_dotdot[ich] * dt4;
            eta_dot[ich] *= expfac;
        }
        factor *= exp(-dt2 * eta_dot[0]);
        for (int ich = 0; ich < numNHChains; ich++)
            eta[ich] += dt2 * eta_dot[ich];
        eta_dotdot[0] = (ke2 * factor * factor - ke2_target) / eta_mass[0];
        eta_dot[0] *= expfac;            /* stale expfac from ich == 0 above (quirk Q10) */
        eta_dot[0] += eta_dotdot[0] * dt4;
        eta_dot[0] *= expfac;
        for (int ich = 1; ich < numNHChains; ich++) {
            expfac = exp(-dt8 * eta_dot[ich + 1]);
            eta_dot[ich] *= expfac;
            eta_dotdot[ich] = (eta_mass[ich - 1] * eta_dot[ich - 1] * eta_dot[ich - 1] - VVO_BOLTZ * t_target) / eta_mass[ich];
            eta_dot[ich] += eta_dotdot[ich] * dt4;
            eta_dot[ich] *= expfac;
        }
    }
    *factor_out = factor;
}

/* ------------------------------------------------------------------ HOST:670-754 */
void vvo_nh_scale_velocity(vvo_system* s) {
    if (s->use_com_tg) {
        vvo_calc_com_velocities(s->num_molecules_nh, s->velm, s->com_velm, s->particles_in_molecules,
                                s->particles_sorted_by_mol_id, s->molecules_nh);
        vvo_normalize_velocities(s->num_particles_nh, s->velm, s->com_velm, s->particle_mol_id, s->particles_nh);
    }
    mixed ke[VVO_NUM_TG_MAX] = { 0, 0, 0 };
    vvo_compute_kinetic_energies(s->num_tg, s->num_normal_nh, s->num_molecules_nh, s->num_pairs_nh, s->velm,
                                 s->com_velm, s->normal_nh, s->pairs_nh, s->molecules_nh, ke);
    double vs[VVO_NUM_TG_MAX] = { 1.0, 1.0, 1.0 };
    for (int itg = 0; itg < s->num_tg; itg++) {
        s->ke2[itg] = (double) ke[itg];           /* HOST:709-716: downloaded as float in single mode */
        const double T = itg == TG_DRUDE ? s->drude_temperature : s->temperature;
        if (s->eta_mass[itg][0] > 0)
            vvo_propagate_nh_chain(s->num_chains, s->loops_per_step, s->dt, s->eta[itg], s->eta_dot[itg],
                                   s->eta_dotdot[itg], s->eta_mass[itg], s->ke2[itg], s->tg_nkbt[itg], T, &vs[itg]);
    }
    mixed vsm[VVO_NUM_TG_MAX];
    for (int i = 0; i < VVO_NUM_TG_MAX; i++) {
        s->vscale[i] = vs[i];
        vsm[i] = (mixed) vs[i];                   /* HOST:741-746: uploaded as float in single mode */
    }
    vvo_scale_velocity(s->num_normal_nh, s->num_pairs_nh, s->velm, s->com_velm, s->particle_mol_id, s->normal_nh,
                       s->pairs_nh, vsm);
}

/* ------------------------------------------------------------------ ours: synthetic force provider.
 * F_i = -k_t (x_i - site_i) for every massive particle, plus a Drude-parent spring
 * -k_D (x_d - x_p) on each Drude pair; evaluated in `real` from posq (as OpenMM forces are),
 * written as fixed point x 2^32 with truncation toward zero, planar x|y|z.
 * The product's device kernel (vvhip_synth_tether_force) must give identical int64 values. */
void vvo_tether_force(vvo_system* s) {
    const int n = s->num_atoms, P = s->padded_num_atoms;
    const real kt = (real) s->k_tether, kd = (real) s->k_drude;
    const real scale = (real) 4294967296.0;
    PAR_FOR
    for (int i = 0; i < n; i++) {
        real fx = 0, fy = 0, fz = 0;
        if (s->velm[i].w != 0) {
            fx = -kt * (s->posq[i].x - s->site[i].x);
            fy = -kt * (s->posq[i].y - s->site[i].y);
            fz = -kt * (s->posq[i].z - s->site[i].z);
        }
        s->force[i] = (long long) (fx * scale);
        s->force[i + P] = (long long) (fy * scale);
        s->force[i + 2 * P] = (long long) (fz * scale);
    }
    PAR_FOR
    for (int k = 0; k < s->num_drude_pairs; k++) {
        int d = s->drude_pairs[k].x, p = s->drude_pairs[k].y;
        real sx = -kd * (s->posq[d].x - s->posq[p].x);
        real sy = -kd * (s->posq[d].y - s->posq[p].y);
        real sz = -kd * (s->posq[d].z - s->posq[p].z);
        long long ix = (long long) (sx * scale), iy = (long long) (sy * scale), iz = (long long) (sz * scale);
        s->force[d] += ix; s->force[d + P] += iy; s->force[d + 2 * P] += iz;
        s->force[p] -= ix; s->force[p + P] -= iy; s->force[p + 2 * P] -= iz;
    }
}

/* ------------------------------------------------------------------ step sequencing */
static void calc_forces(vvo_system* s) {           /* context->calcForcesAndEnergy: OpenMM's, out of scope */
    if (s->force_mode == 1)
        vvo_tether_force(s);
}
static unsigned int prepare_random_numbers(vvo_system* s, unsigned int numValues) {
    /* CudaIntegrationUtilities::prepareRandomNumbers (OpenMM, not vendored): hand out a slice of the
     * buffer, "regenerate" (here: rewind the caller's fixed buffer) when it is exhausted. */
    if (s->random_index + numValues <= s->random_size) {
        unsigned int old = s->random_index;
        s->random_index += numValues;
        return old;
    }
    s->random_index = numValues;
    return 0;
}
static void apply_extra_forces(vvo_system* s) {    /* API:238-245 / API:316-323 */
    const int n = s->num_atoms;
    const int any = s->num_particles_ld > 0 || s->num_electrolyte > 0 || s->cos_accel != 0;
    if (any)
        vvo_reset_extra_force(n, s->force_extra);
    if (s->num_particles_ld > 0) {                 /* HOST:826-872 */
        double stepSize = s->dt;
        double dragFactor = s->friction;
        double randFactor = sqrt(2.0 * VVO_BOLTZ * s->temperature * dragFactor / stepSize);
        double dragFactorDrude = s->drude_friction;
        double randFactorDrude = sqrt(2.0 * VVO_BOLTZ * s->drude_temperature * dragFactorDrude / stepSize);
        /* array sizes are max(size,1) in the reference (HOST:806-807,863) */
        unsigned int nn = s->num_normal_ld > 1 ? s->num_normal_ld : 1;
        unsigned int np = s->num_pairs_ld > 1 ? s->num_pairs_ld : 1;
        unsigned int randomIndex = prepare_random_numbers(s, nn + 2 * np);
        vvo_add_extra_force_drude_langevin(s->num_normal_ld, s->num_pairs_ld, s->velm, s->force_extra, s->normal_ld,
                                           s->pairs_ld, (mixed) dragFactor, (mixed) randFactor, (mixed) dragFactorDrude,
                                           (mixed) randFactorDrude, s->random, randomIndex);
    }
    if (s->num_electrolyte > 0) {                  /* HOST:971-992 */
        double efscale = s->efield * VVO_AVOGADRO;
        vvo_add_extra_force_electric_field(s->num_electrolyte, s->posq, s->force_extra, s->particles_electrolyte,
                                           (real) efscale);
    }
    if (s->cos_accel != 0)                         /* HOST:1037-1059 */
        vvo_add_cos_acceleration(n, s->posq, s->velm, s->force_extra, (real) s->cos_accel, (real) (1.0 / s->box[2]));
}
static void nh_half(vvo_system* s) {               /* API:251-260 / 295-304 / 327-336 */
    if (s->num_particles_nh <= 0)
        return;
    const int n = s->num_atoms;
    const real ibz = (real) (1.0 / s->box[2]);
    if (s->cos_accel != 0) {
        vvo_calc_periodic_velocity_bias(n, s->posq, s->velm, s->v_buffer, ibz);
        vvo_sum_v(n, s->v_buffer, s->inv_mass_total);
        vvo_remove_periodic_velocity_bias(n, s->posq, s->velm, s->v_buffer, ibz);
    }
    vvo_nh_scale_velocity(s);
    if (s->cos_accel != 0)
        vvo_restore_periodic_velocity_bias(n, s->posq, s->velm, s->v_buffer, ibz);
}
static void hard_wall(vvo_system* s) {             /* HOST:189-212 / 307-372 */
    if (s->max_drude_distance > 0 && s->num_drude_pairs > 0) {
        double hardwallScaleDrude = sqrt(VVO_BOLTZ * s->drude_temperature);
        vvo_apply_hard_wall(s->num_drude_pairs, s->posq, s->posq_corr, s->velm, s->drude_pairs, (mixed) s->dt,
                            (mixed) s->max_drude_distance, (mixed) hardwallScaleDrude);
    }
}
static void shake_v(vvo_system* s) {                /* shake_mode 0: Gauss-Seidel sweeps, 1: all constraints of a cluster at once */
    if (s->shake_mode == 0) vvo_shake_velocities(s->num_shake, s->shake_atoms, s->shake_params, (mixed) s->constraint_tolerance, s->posq, s->posq_corr, s->velm);
    else vvo_cluster_velocities_direct(s->num_shake, s->shake_atoms, s->shake_params, s->posq, s->posq_corr, s->velm);
}
static void shake_x(vvo_system* s) {
    if (s->shake_mode == 0) vvo_shake_positions(s->num_shake, s->shake_atoms, s->shake_params, (mixed) s->constraint_tolerance, s->posq, s->posq_corr, s->pos_delta);
    else vvo_cluster_positions_newton(s->num_shake, s->shake_atoms, s->shake_params, (mixed) s->constraint_tolerance, s->posq, s->posq_corr, s->pos_delta);
}
static void step_middle(vvo_system* s) {           /* API:232-270; constraints/virtual sites/reorder are OpenMM's */
    const int n = s->num_atoms;
    calc_forces(s);
    apply_extra_forces(s);
    vvo_integrate_middle_vel(n, s->padded_num_atoms, s->velm, s->force, s->force_extra, (mixed) s->dt);  /* HOST:144-148 */
    if (s->num_shake > 0)   /* integration.applyVelocityConstraints, HOST:151 */
        shake_v(s);
    if (s->num_general > 0) vvo_general_velocities(s->num_general, s->general_atoms, s->general_params, (mixed) s->constraint_tolerance, (mixed) s->general_omega, s->posq, s->posq_corr, s->velm);
    if (s->num_settle > 0) vvo_settle_velocities(s->num_settle, s->settle_atoms, s->posq, s->posq_corr, s->velm);
    vvo_integrate_middle_pos1(n, s->velm, s->pos_delta, s->old_delta, (mixed) s->dt);                     /* HOST:154-158 */
    nh_half(s);
    vvo_integrate_middle_pos2(n, s->velm, s->pos_delta, s->old_delta, (mixed) s->dt);                     /* HOST:169-173 */
    if (s->num_shake > 0)   /* integration.applyConstraints, HOST:176 */
        shake_x(s);
    if (s->num_general > 0) vvo_general_positions(s->num_general, s->general_atoms, s->general_params, (mixed) s->constraint_tolerance, (mixed) s->general_omega, s->posq, s->posq_corr, s->pos_delta);
    if (s->num_settle > 0) vvo_settle_positions(s->num_settle, s->settle_atoms, s->settle_params, s->posq, s->posq_corr, s->velm, s->pos_delta);
    vvo_integrate_middle_pos3(n, s->posq, s->posq_corr, s->pos_delta, s->old_delta, s->velm, (mixed) s->dt); /* HOST:179-185 */
    hard_wall(s);
    if (s->num_vsites > 0) vvo_compute_virtual_sites(s->num_vsites, s->vsite_atoms, s->vsite_params, s->posq, s->posq_corr);   /* HOST:214, 374 */
    if (s->num_images > 0)
        vvo_update_image_positions(s->num_images, s->posq, s->posq_corr, s->image_pairs, (mixed) s->mirror);
}
static void step_vv(vvo_system* s) {               /* API:272-338 */
    const int n = s->num_atoms;
    if (!s->forces_valid) {
        calc_forces(s);
        s->forces_valid = 1;
    }
    nh_half(s);
    double fscale = 0.5 * s->dt / (double) 0x100000000;                                                   /* HOST:306 */
    vvo_vv_integrate_velocities(n, s->padded_num_atoms, s->velm, s->force, s->force_extra, s->pos_delta,
                                (mixed) s->dt, (mixed) fscale, 1);                                        /* HOST:341-348 */
    if (s->num_shake > 0)   /* HOST:351 */
        shake_x(s);
    if (s->num_general > 0) vvo_general_positions(s->num_general, s->general_atoms, s->general_params, (mixed) s->constraint_tolerance, (mixed) s->general_omega, s->posq, s->posq_corr, s->pos_delta);
    if (s->num_settle > 0) vvo_settle_positions(s->num_settle, s->settle_atoms, s->settle_params, s->posq, s->posq_corr, s->velm, s->pos_delta);
    vvo_vv_integrate_positions(n, s->posq, s->posq_corr, s->pos_delta, s->velm, (mixed) s->dt);           /* HOST:355-360 */
    hard_wall(s);
    if (s->num_vsites > 0) vvo_compute_virtual_sites(s->num_vsites, s->vsite_atoms, s->vsite_params, s->posq, s->posq_corr);   /* HOST:214, 374 */
    if (s->num_images > 0)
        vvo_update_image_positions(s->num_images, s->posq, s->posq_corr, s->image_pairs, (mixed) s->mirror);
    calc_forces(s);
    s->forces_valid = 1;
    apply_extra_forces(s);
    vvo_vv_integrate_velocities(n, s->padded_num_atoms, s->velm, s->force, s->force_extra, s->pos_delta,
                                (mixed) s->dt, (mixed) fscale, 0);                                        /* HOST:417-424 */
    if (s->num_shake > 0)   /* HOST:427 */
        shake_v(s);
    if (s->num_general > 0) vvo_general_velocities(s->num_general, s->general_atoms, s->general_params, (mixed) s->constraint_tolerance, (mixed) s->general_omega, s->posq, s->posq_corr, s->velm);
    if (s->num_settle > 0) vvo_settle_velocities(s->num_settle, s->settle_atoms, s->posq, s->posq_corr, s->velm);
    nh_half(s);
}
void vvo_step(vvo_system* s, int steps) {
    g_threads = s->num_threads > 1 ? s->num_threads : 1;
    for (int i = 0; i < steps; i++) {
        if (s->use_middle)
            step_middle(s);
        else
            step_vv(s);
    }
    g_threads = 1;
}

/* ------------------------------------------------------------------ HOST:1112-1134 */
double vvo_calc_viscosity(const vvo_system* s, double* vmax_out) {
    double vMax = (double) s->v_buffer[0];
    double vol = s->box[0] * s->box[1] * s->box[2];
    if (vmax_out) *vmax_out = vMax;
    return vMax * vol * s->inv_mass_total / s->cos_accel * (2 * 3.1415926 / s->box[2]) * (2 * 3.1415926 / s->box[2]);
}
