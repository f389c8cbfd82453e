// Host check of vv::cos_short_range (csrc/vv_layout.h): the function is +, *, fma and rint only, so what it returns here is what it
// returns on gfx950, bit for bit.  Reference: libquadmath.  Prints the worst error in ulp of the correctly rounded result and the
// number of arguments sent to the library cosine; exit code 1 beyond 1 ulp.  Build: g++ -O2 -ffp-contract=off -mfma (tests/test_cos_poly.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#include <quadmath.h>
#include "vv_layout.h"

static double ulp_of(double y) { int e; std::frexp(y, &e); return std::ldexp(1.0, e - 53); }
static uint64_t s = 88172645463325252ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

int main() {
    double worst = 0;
    long n = 0, fallbacks = 0, differs = 0;
    auto probe = [&](double x) {
        bool ok;
        double f = vv::cos_short_range(x, ok);
        if (!ok) { f = std::cos(x); fallbacks++; }
        const __float128 ref = cosq((__float128) x);
        const double err = std::fabs((double) ((__float128) f - ref)) / ulp_of((double) ref);
        if (err > worst) worst = err;
        if (f != std::cos(x)) differs++;
        n++;
    };
    // (a) arguments as the kernels form them: 2 * 3.1415926 * z * invBoxZ with float z in [-1.5, 2.5] box lengths, float 1 / Lz
    const float boxes[] = {18.3f, 4.64f, 16.0f, 6.2f, 3.3f};
    for (float box : boxes) {
        const float ib = (float) (1.0 / box);
        for (long i = 0; i < (1 << 20) * 4 / 5; i++) {
            const float zf = (float) (((double) (rnd() >> 11) / 9007199254740992.0) * 4.0 - 1.5) * box;
            probe(2 * 3.1415926 * zf * ib);
        }
    }
    // (b) |x| <= bound, half of the arguments pushed next to a multiple of pi/2 (relative distance 1e-3 ... 1e-16, log-uniform)
    for (double bound : {16.0, 256.0, 1024.0, 4096.0}) {
        for (long i = 0; i < (1 << 20); i++) {
            const uint64_t r = rnd();
            double x = ((double) (r >> 11) / 9007199254740992.0 * 2 - 1) * bound;
            if (i & 1) x = std::rint(x / 1.5707963267948966) * 1.5707963267948966 * (1 + (((r >> 20) & 1) ? 1 : -1) * std::pow(10.0, -3 - 13.0 * (double) (r & 1023) / 1023.0));
            probe(x);
        }
    }
    std::printf("samples %ld worst_ulp %.4f fallbacks %ld differs_from_host_libm %ld\n", n, worst, fallbacks, differs);
    return worst <= 1.0 ? 0 : 1;
}
