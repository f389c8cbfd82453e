// oracle/ref_host_kernels.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// The launcher behind oracle/refhost/CudaContext.h::executeKernel: the reference's host code (CudaVVKernels.cpp, compiled in place)
// hands over a kernel name, the `defines` map of its module and CUDA-style `void** args`; this file sets the size macros the kernels
// read (oracle/ref_prelude.h: the host `defines` as run-time ints) and calls the reference's OWN kernel, compiled for the CPU from
// K/*.cu by oracle/Makefile (one block of one thread: every grid-stride loop covers the whole range).  The trampolines only unpack
// the arguments in the order and with the types of the kernels' signatures (K/middle.cu:6,29,47,66,106,227; K/velocityVerlet.cu:6,35,74,195;
// K/drudeNoseHoover.cu:5,37,55,121,157; K/cosineAccelerate.cu:2,16,34,63,76; K/drudeLangevin.cu:2; K/electricField.cu:2;
// K/imageCharge.cu:2); no arithmetic happens here.  Built with -include ref_prelude.h.
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>

extern "C" {
// ---- the reference's kernels (symbols renamed on the command line where two .cu files define the same name, oracle/Makefile)
void integrateMiddleVel(mixed4*, const long long*, const real3*, const mixed2*);
void integrateMiddlePos1(const mixed4*, mixed4*, mixed4*, const mixed2*);
void integrateMiddlePos2(const mixed4*, mixed4*, mixed4*, const mixed2*);
void integrateMiddlePos3(real4*, real4*, const mixed4*, const mixed4*, mixed4*, const mixed2*);
void applyHardWallConstraints(real4*, real4*, mixed4*, const int2*, const mixed2*, const mixed, const mixed);
void resetExtraForce(real3*);
void velocityVerletIntegrateVelocities(mixed4*, const long long*, const real3*, mixed4*, const mixed2*, const mixed, bool);
void velocityVerletIntegratePositions(real4*, real4*, const mixed4*, mixed4*, const mixed2*);
void vv_applyHardWallConstraints(real4*, real4*, mixed4*, const int2*, const mixed2*, mixed, mixed);
void vv_resetExtraForce(real3*);
#define VVRH_NH(TG)                                                                                                            \
    void calcCOMVelocities_tg##TG(const mixed4*, mixed4*, const int2*, const int*, const int*);                                \
    void normalizeVelocities_tg##TG(mixed4*, const mixed4*, const int*, const int*);                                           \
    void computeNormalizedKineticEnergies_tg##TG(const mixed4*, const mixed4*, const int*, const int2*, mixed*, const int*, int); \
    void sumNormalizedKineticEnergies_tg##TG(mixed*, mixed*, int);                                                             \
    void scaleVelocity_tg##TG(mixed4*, const mixed4*, const int*, const int*, const int2*, const mixed*);
VVRH_NH(1) VVRH_NH(2) VVRH_NH(3)
void addCosAcceleration(const real4*, const mixed4*, real3*, real, const real4);
void calcPeriodicVelocityBias(const real4*, const mixed4*, mixed*, const real4);
void sumV(mixed*, double, int);
void removePeriodicVelocityBias(const real4*, mixed4*, const mixed*, const real4);
void restorePeriodicVelocityBias(const real4*, mixed4*, const mixed*, const real4);
void addExtraForceDrudeLangevin(const mixed4*, real3*, const int*, const int2*, mixed, mixed, mixed, mixed, const float4*, unsigned int);
void addExtraForceElectricField(real4*, real3*, const int*, real);
void updateImagePositions(real4*, real4*, const int2*, mixed);
}

extern "C" { vvref_sizes_t vvref_sizes = {}; }   // the size macros of ref_prelude.h read this (this build has no ref_glue.cpp)
mixed temp[64];                                  // backing store of the kernels' `extern __shared__ mixed temp[]` (1 thread: temp[0..NUM_TG-1])

namespace {
template <class T> T* ptr(void** a, int i) { return (T*) (size_t) *(unsigned long long*) a[i]; }      // args[i] -> CUdeviceptr -> host address
template <class T> T val(void** a, int i) { return *(T*) a[i]; }

void set_sizes(const std::map<std::string, std::string>& d) {
    struct { const char* key; int* field; } table[] = {
        {"NUM_ATOMS", &vvref_sizes.num_atoms}, {"PADDED_NUM_ATOMS", &vvref_sizes.padded_num_atoms}, {"NUM_DRUDE_PAIRS", &vvref_sizes.num_drude_pairs},
        {"NUM_PARTICLES_NH", &vvref_sizes.num_particles_nh}, {"NUM_MOLECULES_NH", &vvref_sizes.num_molecules_nh},
        {"NUM_NORMAL_PARTICLES_NH", &vvref_sizes.num_normal_particles_nh}, {"NUM_PAIRS_NH", &vvref_sizes.num_pairs_nh},
        {"NUM_NORMAL_PARTICLES_LD", &vvref_sizes.num_normal_particles_ld}, {"NUM_PAIRS_LD", &vvref_sizes.num_pairs_ld},
        {"NUM_IMAGES", &vvref_sizes.num_images}, {"NUM_PARTICLES_ELECTROLYTE", &vvref_sizes.num_particles_electrolyte}};
    for (auto& t : table) {
        auto it = d.find(t.key);
        if (it != d.end()) *t.field = std::atoi(it->second.c_str());
    }
}
}  // namespace

extern "C" void vvrh_launch(const char* source, const char* name_, const std::map<std::string, std::string>* defines, void** a) {
    set_sizes(*defines);
    const std::string src(source), name(name_);
    auto unit = [&](const char* u) { return src.find(std::string("[") + u + "]") != std::string::npos; };
    if (unit("middle")) {
        if (name == "integrateMiddleVel") return integrateMiddleVel(ptr<mixed4>(a, 0), ptr<const long long>(a, 1), ptr<const real3>(a, 2), ptr<const mixed2>(a, 3));
        if (name == "integrateMiddlePos1") return integrateMiddlePos1(ptr<const mixed4>(a, 0), ptr<mixed4>(a, 1), ptr<mixed4>(a, 2), ptr<const mixed2>(a, 3));
        if (name == "integrateMiddlePos2") return integrateMiddlePos2(ptr<const mixed4>(a, 0), ptr<mixed4>(a, 1), ptr<mixed4>(a, 2), ptr<const mixed2>(a, 3));
        if (name == "integrateMiddlePos3") return integrateMiddlePos3(ptr<real4>(a, 0), ptr<real4>(a, 1), ptr<const mixed4>(a, 2), ptr<const mixed4>(a, 3), ptr<mixed4>(a, 4), ptr<const mixed2>(a, 5));
        if (name == "applyHardWallConstraints") return applyHardWallConstraints(ptr<real4>(a, 0), ptr<real4>(a, 1), ptr<mixed4>(a, 2), ptr<const int2>(a, 3), ptr<const mixed2>(a, 4), val<mixed>(a, 5), val<mixed>(a, 6));
        if (name == "resetExtraForce") return resetExtraForce(ptr<real3>(a, 0));
    }
    if (unit("velocityVerlet")) {
        if (name == "velocityVerletIntegrateVelocities") return velocityVerletIntegrateVelocities(ptr<mixed4>(a, 0), ptr<const long long>(a, 1), ptr<const real3>(a, 2), ptr<mixed4>(a, 3), ptr<const mixed2>(a, 4), val<mixed>(a, 5), val<bool>(a, 6));
        if (name == "velocityVerletIntegratePositions") return velocityVerletIntegratePositions(ptr<real4>(a, 0), ptr<real4>(a, 1), ptr<const mixed4>(a, 2), ptr<mixed4>(a, 3), ptr<const mixed2>(a, 4));
        if (name == "applyHardWallConstraints") return vv_applyHardWallConstraints(ptr<real4>(a, 0), ptr<real4>(a, 1), ptr<mixed4>(a, 2), ptr<const int2>(a, 3), ptr<const mixed2>(a, 4), val<mixed>(a, 5), val<mixed>(a, 6));
        if (name == "resetExtraForce") return vv_resetExtraForce(ptr<real3>(a, 0));
    }
    if (unit("drudeNoseHoover")) {
        const int tg = std::atoi(defines->at("NUM_TG").c_str());
#define VVRH_NH_CALLS(TG)                                                                                                                                   \
        if (tg == TG) {                                                                                                                                     \
            if (name == "calcCOMVelocities") return calcCOMVelocities_tg##TG(ptr<const mixed4>(a, 0), ptr<mixed4>(a, 1), ptr<const int2>(a, 2), ptr<const int>(a, 3), ptr<const int>(a, 4)); \
            if (name == "normalizeVelocities") return normalizeVelocities_tg##TG(ptr<mixed4>(a, 0), ptr<const mixed4>(a, 1), ptr<const int>(a, 2), ptr<const int>(a, 3)); \
            if (name == "computeNormalizedKineticEnergies") return computeNormalizedKineticEnergies_tg##TG(ptr<const mixed4>(a, 0), ptr<const mixed4>(a, 1), ptr<const int>(a, 2), ptr<const int2>(a, 3), ptr<mixed>(a, 4), ptr<const int>(a, 5), val<int>(a, 6)); \
            if (name == "sumNormalizedKineticEnergies") return sumNormalizedKineticEnergies_tg##TG(ptr<mixed>(a, 0), ptr<mixed>(a, 1), val<int>(a, 2));   \
            if (name == "scaleVelocity") return scaleVelocity_tg##TG(ptr<mixed4>(a, 0), ptr<const mixed4>(a, 1), ptr<const int>(a, 2), ptr<const int>(a, 3), ptr<const int2>(a, 4), ptr<const mixed>(a, 5)); \
        }
        VVRH_NH_CALLS(1) VVRH_NH_CALLS(2) VVRH_NH_CALLS(3)
    }
    if (unit("cosineAccelerate")) {
        if (name == "addCosAcceleration") return addCosAcceleration(ptr<const real4>(a, 0), ptr<const mixed4>(a, 1), ptr<real3>(a, 2), val<real>(a, 3), val<real4>(a, 4));
        if (name == "calcPeriodicVelocityBias") return calcPeriodicVelocityBias(ptr<const real4>(a, 0), ptr<const mixed4>(a, 1), ptr<mixed>(a, 2), val<real4>(a, 3));
        if (name == "sumV") return sumV(ptr<mixed>(a, 0), val<double>(a, 1), val<int>(a, 2));
        if (name == "removePeriodicVelocityBias") return removePeriodicVelocityBias(ptr<const real4>(a, 0), ptr<mixed4>(a, 1), ptr<const mixed>(a, 2), val<real4>(a, 3));
        if (name == "restorePeriodicVelocityBias") return restorePeriodicVelocityBias(ptr<const real4>(a, 0), ptr<mixed4>(a, 1), ptr<const mixed>(a, 2), val<real4>(a, 3));
    }
    if (unit("drudeLangevin") && name == "addExtraForceDrudeLangevin")
        return addExtraForceDrudeLangevin(ptr<const mixed4>(a, 0), ptr<real3>(a, 1), ptr<const int>(a, 2), ptr<const int2>(a, 3), val<mixed>(a, 4), val<mixed>(a, 5), val<mixed>(a, 6), val<mixed>(a, 7), ptr<const float4>(a, 8), val<unsigned int>(a, 9));
    if (unit("electricField") && name == "addExtraForceElectricField")
        return addExtraForceElectricField(ptr<real4>(a, 0), ptr<real3>(a, 1), ptr<const int>(a, 2), val<real>(a, 3));
    if (unit("imageCharge") && name == "updateImagePositions")
        return updateImagePositions(ptr<real4>(a, 0), ptr<real4>(a, 1), ptr<const int2>(a, 2), val<mixed>(a, 3));
    throw std::runtime_error("vvrh_launch: no kernel '" + name + "' in module '" + src + "'");
}
