/*
 * oracle/ref_glue.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Storage that the reference kernels expect from their launcher when they are
 * built for the CPU by oracle/Makefile (see ref_prelude.h):
 *   - vvref_sizes : the host `defines` map as run-time ints;
 *   - temp[]      : backing store for `extern __shared__ mixed temp[]`
 *                   (drudeNoseHoover.cu:127, cosineAccelerate.cu:42); one thread
 *                   only touches temp[0..NUM_TG-1].
 */
#include "ref_prelude.h"

extern "C" {
vvref_sizes_t vvref_sizes = {};
void vvref_set_sizes(const vvref_sizes_t* s) { vvref_sizes = *s; }
int vvref_sizeof_real(void) { return (int) sizeof(real); }
int vvref_sizeof_mixed(void) { return (int) sizeof(mixed); }
}
mixed temp[64];
