/*
 * oracle/ref_gpu_prelude.h -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Same role as ref_prelude.h, for building the reference's own kernels (platforms/cuda/src/kernels/*.cu, compiled where
 * they lie, unmodified) as gfx950 device code with hipcc, so that the reference's per-step kernel sequence can be run
 * on the same MI355X next to the product (oracle/ref_gpu_driver.cpp) -- a second parity pin and an apples-to-apples
 * timing baseline.  HIP already provides the CUDA language built-ins (__global__, blockIdx, float4, make_float4,
 * __syncthreads, extern __shared__); this header adds only what OpenMM's CudaContext::createModule prepends at JIT
 * time: the precision typedefs / RECIP / SQRT (assumptions as listed in ref_prelude.h) and the host `defines` map, here a
 * per-translation-unit __constant__ struct so one binary serves every system size.
 */
#pragma once
#include <hip/hip_runtime.h>

#if defined(VVREF_MIXED)
typedef float real;    typedef float2 real2;   typedef float3 real3;   typedef float4 real4;
typedef double mixed;  typedef double2 mixed2; typedef double3 mixed3; typedef double4 mixed4;
#define make_real2 make_float2
#define make_real3 make_float3
#define make_real4 make_float4
#define make_mixed2 make_double2
#define make_mixed3 make_double3
#define make_mixed4 make_double4
#define USE_MIXED_PRECISION 1
#define SQRT sqrtf
#define RSQRT rsqrtf
#define RECIP(x) (1.0f/(x))
#elif defined(VVREF_SINGLE)
typedef float real;    typedef float2 real2;   typedef float3 real3;   typedef float4 real4;
typedef float mixed;   typedef float2 mixed2;  typedef float3 mixed3;  typedef float4 mixed4;
#define make_real2 make_float2
#define make_real3 make_float3
#define make_real4 make_float4
#define make_mixed2 make_float2
#define make_mixed3 make_float3
#define make_mixed4 make_float4
#define SQRT sqrtf
#define RSQRT rsqrtf
#define RECIP(x) (1.0f/(x))
#else
#error "define VVREF_MIXED or VVREF_SINGLE"
#endif

struct vvref_sizes_t {
    int num_atoms, padded_num_atoms, num_drude_pairs;
    int num_particles_nh, num_molecules_nh, num_normal_particles_nh, num_pairs_nh;
    int num_normal_particles_ld, num_pairs_ld;
    int num_images, num_particles_electrolyte;
};
#ifdef VVREF_TU                       /* a reference kernel file: gets its own copy of the defines + a setter */
static __constant__ vvref_sizes_t vvref_sizes;
#define VVREF_CAT2(a, b) a##b
#define VVREF_CAT(a, b) VVREF_CAT2(a, b)
extern "C" void VVREF_CAT(vvref_set_sizes_, VVREF_TU)(const vvref_sizes_t* s) {
    (void) hipMemcpyToSymbol(HIP_SYMBOL(vvref_sizes), s, sizeof(*s));
}
#define NUM_ATOMS                 (vvref_sizes.num_atoms)
#define PADDED_NUM_ATOMS          (vvref_sizes.padded_num_atoms)
#define NUM_DRUDE_PAIRS           (vvref_sizes.num_drude_pairs)
#define NUM_PARTICLES_NH          (vvref_sizes.num_particles_nh)
#define NUM_MOLECULES_NH          (vvref_sizes.num_molecules_nh)
#define NUM_NORMAL_PARTICLES_NH   (vvref_sizes.num_normal_particles_nh)
#define NUM_PAIRS_NH              (vvref_sizes.num_pairs_nh)
#define NUM_NORMAL_PARTICLES_LD   (vvref_sizes.num_normal_particles_ld)
#define NUM_PAIRS_LD              (vvref_sizes.num_pairs_ld)
#define NUM_IMAGES                (vvref_sizes.num_images)
#define NUM_PARTICLES_ELECTROLYTE (vvref_sizes.num_particles_electrolyte)
#define TG_ATOM 0
#define TG_COM 1
#define TG_DRUDE 2
#endif
