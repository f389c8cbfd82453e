// vv_host.hpp -- host-side analysis shared by libvvhip (C ABI) and the openmmapi layer. No HIP here.
//
// Restates, for the HIP backend, what the reference spreads over VVIntegrator::initialize
// (openmmapi/src/VVIntegrator.cpp:92-188) and the initialize() methods of the seven Cuda*Kernel
// classes (platforms/cuda/src/CudaVVKernels.cpp:56-117, 462-667, 761-824, 878-902, 940-969,
// 998-1035) -- "HOST" below -- and then lays the particles out for one-work-item-per-particle
// kernels on 64-lane wavefronts.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vvhip.h"
#include "vv_layout.h"

namespace vv {

// Stands in for OpenMMException inside the core; carries the vvhip error code.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};


struct ResolvedParams : vvhip_params {};

// Arithmetic ("periodic") work-item layout.  Most systems are a few runs of identical molecules (250 cations, 250 anions, ...), often
// repeated cell after cell.  Then the slot table is a pure function of the wave index: a wave of region r holds reg_P[r] consecutive
// particles (whole repeat units), cell c of the system starts apc particles / wpc waves / spc COM segments after cell c - 1, and the
// role words of every wave of a region equal those of the region's first wave in cell 0.  The kernels then compute the particle index
// instead of loading it -- no dependent memory round trip in front of the particle loads, no 8 bytes per lane of index traffic -- and
// read the role words of that one pattern wave.  analyze() only enables this when the formula reproduces the explicit slot table
// (which is always built and stays the reference for every other kernel) lane for lane.
struct PeriodicLayout {
    int32_t enabled = 0, nreg = 0, ncells = 1;
    int32_t apc = 0, wpc = 0, spc = 0;     // particles / waves / COM segments per cell
    uint32_t magic = 0;                    // cell = umulhi(wave, magic) (= wave / wpc for every wave of the plan); 0 when ncells == 1
    int32_t reg_wave_start[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};   // cell-local first wave of region r (unused: INT_MAX)
    int32_t reg_atom_start[4] = {0, 0, 0, 0}, reg_atom_end[4] = {0, 0, 0, 0};       // cell-local particle range of region r
    int32_t reg_seg_start[4] = {0, 0, 0, 0};                                        // cell-local first COM segment of region r
    int32_t reg_P[4] = {0, 0, 0, 0}, reg_spw[4] = {0, 0, 0, 0};                     // particles / COM segments per full wave
};

struct HostPlan {
    PeriodicLayout per;
    int precision = VVHIP_MIXED;
    vvhip_params params{};          // after the auto rules of API:106-121
    vvhip_plan_info info{};
    int32_t num_atoms = 0, padded_num_atoms = 0;
    int32_t shard_begin = 0, shard_end = 0;
    bool has_nh = false, has_ld = false, has_ef = false, has_images = false, has_pairs = false;
    // work-item layout: 64 lanes per wave; slots[2*i] = shard-relative particle index or -1, slots[2*i+1] = meta
    std::vector<int32_t> slots;
    std::vector<int32_t> slot_image;   // [64*waves] shard-relative image particle of this lane's particle, or -1
    // [2*64*waves] at the slot of a COM segment's first lane: mass of the segment's thermostatted particles, summed in particle order as
    // the reference does (K/drudeNoseHoover.cu:15-25), and its reciprocal -- static, so kernel A need not scan the masses every step
    std::vector<double> seg_mass;      // (mass, 1/mass) per COM segment, DENSE: entry seg_base[wave] + (number of COM-leader lanes below the segment's leader lane)
    std::vector<int32_t> seg_base;     // [num_waves + 1] number of COM segments in the waves before this one (last entry: the total)
    std::vector<int32_t> slot_rand;    // [64*waves] offset into the Langevin slice of the random buffer, or -1
    // in-kernel SHAKE (hydrogen-type clusters): per lane a packed word and, for central lanes, OpenMM-style cluster parameters
    //   word: bit0 central, bit1 peripheral, bits2-3 = #peripherals (central) or own index (peripheral),
    //         bits 4-9 / 10-15 / 16-21 = lanes of the peripherals (central) or bits 4-9 = lane of the central (peripheral)
    //   param float4: x = 1/m_central, y = 0.5/(1/m_central + 1/m_peripheral), z = d^2, w = 1/m_peripheral
    //   rigid three-site molecules (SETTLE): bit 30 set on the apex' word, two peripherals; param x = apex-partner distance,
    //   y = partner-partner distance (masses are taken from velm.w)
    std::vector<int32_t> slot_shake;
    std::vector<float> slot_shake_param;
    double gc_omega = 1.0;                     // ... and their relaxation factor (vv_layout.h: GC_OMEGA_*)
    int gc_colors = 0;                         // general constraint clusters: colours of the wave-level Gauss-Seidel sweeps (0 = none); the constraint
                                               // list of a wave then lives in slot_shake / slot_shake_param, one constraint per lane (vv_layout.h: GC_WORD_*)
    std::vector<int32_t> slot_vsite;           // [2*64*waves] (site word, record) of the lanes that place a virtual site (vv_layout.h: VS_WORD_*); empty = none in-kernel
    std::vector<double> vsite_params;          // [12*records]
    std::vector<int32_t> vsite_atom;           // [records] shard-relative particle index of the site (where a hosting lane stores it)
    std::vector<int32_t> slot_big;     // [64*waves] index of the lane's big molecule, or -1 (empty when there is none)
    std::string unfused_reason;        // why info.constraints_fused is 0 (general clusters: what did not fit a wave); empty otherwise
    int32_t num_big = 0;               // molecules with more than 64 thermostatted particles (COM temperature group only)
    double big_scale = 1.0;            // fixed-point scale of their sum(m v) accumulators
    std::vector<int32_t> image_pairs;  // (image, parent) shard-relative, for the stand-alone image kernel
    // reference-style tables, kept for inspection / tests (global particle indices)
    std::vector<int32_t> particles_nh, molecules_nh, normal_nh, pairs_nh, normal_ld, pairs_ld;
};

// Throws vv::Error.  `sys` pointers are only read during the call.
HostPlan analyze(const vvhip_system_desc& sys, const vvhip_params& params, int precision);

// CudaModifyDrudeNoseKernel::initialize's chain sizing for changed parameters is NOT redone on
// set_params (the reference fixes etaMass / NkbT at initialize, HOST:583-594).

}  // namespace vv
