// vv_layout.h -- the role word of a lane and the cluster word of the in-kernel constraints: what the host analysis (vv_host.cpp) writes
// into the slot tables and the kernels (vv_device.inc) read.  No includes: this header is also compiled at run time (vv_rtc.cpp).
#pragma once

namespace vv {

// ---- role word of one lane ("slot"): what this particle needs from the kernels -----------------
enum Role : uint32_t {
    ROLE_NONE = 0,       // no integration work (idle lane, or a massless particle kept only as image parent)
    ROLE_PLAIN = 1,      // massive, neither NH nor Langevin (a massive image particle): kick + drift only
    ROLE_NH_NORMAL = 2,  // NH thermostat, not in a Drude pair      (normalParticlesNH, HOST:529)
    ROLE_NH_DRUDE = 3,   // NH, Drude particle of a pair (pair.x)   (pairParticlesNH,   HOST:523)
    ROLE_NH_PARENT = 4,  // NH, parent atom of a pair (pair.y)
    ROLE_LD_NORMAL = 5,  // Langevin subset                          (normalParticlesLD, HOST:804)
    ROLE_LD_DRUDE = 6,   //                                          (pairParticlesLD,   HOST:791)
    ROLE_LD_PARENT = 7,
};
constexpr uint32_t META_ROLE_MASK = 0xF;
constexpr int META_PARTNER_SHIFT = 4;    // 6 bits: lane of the Drude partner (own lane if none)
constexpr int META_SEGFIRST_SHIFT = 10;  // 6 bits: first lane of this lane's COM segment (molecule)
constexpr int META_SEGLAST_SHIFT = 16;   // 6 bits: last lane of the segment
constexpr uint32_t META_EFIELD = 1u << 22;     // particle is in particlesElectrolyte
constexpr uint32_t META_HAS_IMAGE = 1u << 23;  // particle is the parent of an image particle
constexpr uint32_t META_COM_LEADER = 1u << 24; // lane that adds its molecule's M*V^2 to TG_COM
constexpr uint32_t META_PAIR = 1u << 25;       // member of a DrudeForce pair (hard wall applies)
constexpr uint32_t META_IS_DRUDE = 1u << 26;   // the Drude (pair.x) of that pair
constexpr uint32_t META_MASSIVE = 1u << 27;    // mass != 0 (velm.w != 0)
constexpr uint32_t META_BIGMOL = 1u << 28;     // lane belongs to a molecule too large for one wave: its COM comes from bigacc
// slot_shake word of a lane that belongs to a constraint cluster (0 otherwise); every member carries the whole cluster, so that each
// lane can gather its mates' data itself: bit 0 central (apex) lane, bit 1 peripheral lane, bits 2-3 number of peripherals np,
// bits 4-9 / 10-15 / 16-21 lanes of peripherals 0 / 1 / 2 (unused ones: the central lane), bits 22-27 lane of the central particle,
// bits 28-29 the lane's own index among the peripherals, bit 30 the cluster is a rigid triangle (SETTLE)
constexpr int SHAKE_WORD_CENTRAL_SHIFT = 22;
constexpr int SHAKE_WORD_OWN_SHIFT = 28;
constexpr uint32_t SHAKE_WORD_SETTLE = 1u << 30;
constexpr uint32_t META_SHAKE = 1u << 30;      // member of an in-kernel constraint cluster: the kernels fetch its cluster word, parameters and
                                               // position in the same round of loads as the velocity, not after reading the cluster word
// General constraint clusters (any topology inside one wave): slot_shake then holds the WAVE's constraint list, constraint l in lane l:
// bits 0-5 lane of particle a, 6-11 lane of particle b, 12-15 colour (constraints of one colour share no particle), bit 31 valid;
// slot_shake_param: d^2, 0.5 / (1/m_a + 1/m_b), 1/m_a, 1/m_b
constexpr uint32_t GC_WORD_VALID = 1u << 31;
// Relaxation factor of the general clusters' sweeps (successive over-relaxation of the Gauss-Seidel SHAKE update): every update is taken
// GC_OMEGA times.  Chains and rings converge fastest slightly above 1; clusters that contain TRIANGLES of constraints (HAngles: H-X-H as an
// H-H distance) are stiff and want more.  Full C3 box, steps/s against the factor (tools/probes/gc_omega_scan.py, round 4, every run
// ends with all constraints within tolerance): AllBonds 25.0 / 29.2 / 32.9 / 32.0 / 30.2 k at 1.0 / 1.1 / 1.2 / 1.25 / 1.3;
// HAngles 8.9 / 14.5 / 17.3 / 17.7 / 16.4 / 13.4 k at 1.0 / 1.3 / 1.4 / 1.45 / 1.5 / 1.6.
constexpr double GC_OMEGA_PLAIN = 1.2, GC_OMEGA_TRIANGLES = 1.4;
// virtual-site word of the lane that places a site: lanes of parents 1, 2, 3 in bits 0-5, 6-11, 12-17 | kind (VS_*, = VVHIP_VSITE_*) << 18 | bit 31 valid
constexpr uint32_t VS_WORD_VALID = 1u << 31;
constexpr uint32_t VS_WORD_HOSTED = 1u << 30;      // the lane belongs to one of the site's parents: the site itself has no lane and is stored by index
constexpr int VS_AVERAGE2 = 0, VS_AVERAGE3 = 1, VS_OUT_OF_PLANE = 2, VS_LOCAL_COORDS = 3;
constexpr uint32_t META_BIG_FIRST = 1u << 29;  // leader of the FIRST chunk of such a molecule (adds M*V^2 once, clears bigacc)

inline uint32_t meta_role(uint32_t m) { return m & META_ROLE_MASK; }

// ---- cos(x) for the cos-acceleration stages (K/cosineAccelerate.cu:8, 26, 70, 82: `cos(2*3.1415926*z*invBoxZ)`, double in every mode)
// The library cosine of the device (ocml) spends ~150 instructions per lane on an argument reduction that copes with |x| up to 1e308;
// the argument here is 2 pi z / Lz, a few turns at most.  One Cody-Waite step with the tail kept (k pi/2 subtracted as a double-double:
// product and rounding error of k * PIO2_1 by fma, the second part of pi/2 by another), then the two fdlibm kernels (k_sin.c / k_cos.c
// polynomials, the tail passed as their `y`) and a select on k mod 4: ~45 instructions.  Only +, *, fma and rint, all correctly
// rounded on gfx950 and on the host alike, so the host check (tests/cpp/cos_check.cpp: 2^22 arguments as the kernels form them, 2^22
// spread over |x| <= 1024 with half of them pushed next to multiples of pi/2, against quad precision) holds for the device bit for bit:
// worst error 0.79 ulp.  `ok` = false where the short reduction is not enough -- |x| > 1024, or the argument within 2^-36 of a multiple
// of pi/2, where the third part of pi/2 starts to matter -- the caller then takes the library cosine (wave-uniform branch, practically
// never taken: probability ~1e-11 per lane for positions that are not adversarial).
#if defined(__HIP__) || defined(__HIPCC_RTC__)
#define VV_HOST_DEVICE __host__ __device__
#else
#define VV_HOST_DEVICE
#endif
VV_HOST_DEVICE inline double cos_short_range(double x, bool& ok) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632679489655800e+00;    // the double next to pi/2
    const double PIO2_1T = 6.12323399573676603587e-17;   // pi/2 - PIO2_1, rounded
    const double k = __builtin_rint(x * INV_PIO2);
    const double p = k * PIO2_1, pe = __builtin_fma(k, PIO2_1, -p);      // p + pe = k * PIO2_1 exactly
    const double r0 = x - p;                                              // exact (x / p in [1/2, 2], or p = 0)
    const double t = __builtin_fma(-k, PIO2_1T, -pe);
    const double rh = r0 + t;
    const double bb = rh - r0;
    const double rl = (r0 - (rh - bb)) + (t - bb);                        // rh + rl = r0 + t (two-sum)
    ok = __builtin_fabs(x) <= 1024.0 && __builtin_fabs(rh) >= 0x1p-36;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = rh * rh, w = z * z;
    const double rs = __builtin_fma(z, __builtin_fma(z, S4, S3), S2) + z * w * __builtin_fma(z, S6, S5);
    const double v = z * rh;
    const double sn = rh - ((z * (0.5 * rl - v * rs) - rl) - v * S1);
    const double rc = z * __builtin_fma(z, __builtin_fma(z, C3, C2), C1) + (w * w) * __builtin_fma(z, __builtin_fma(z, C6, C5), C4);
    const double hz = 0.5 * z;
    const double ww = 1.0 - hz;
    const double cs = ww + (((1.0 - ww) - hz) + (z * rc - rh * rl));
    const int n = (int) k & 3;                                            // cos(r + n pi/2): cos, -sin, -cos, sin
    const double res = (n & 1) ? sn : cs;
    return (n == 1 || n == 2) ? -res : res;
}

}  // namespace vv
