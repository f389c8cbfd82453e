// oracle/ref_host_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C entry points that run the REFERENCE's whole step on the CPU: its VVIntegrator (openmmapi/src/VVIntegrator.cpp), its seven CUDA
// kernel classes and their factory (platforms/cuda/src/CudaVVKernels.cpp, CudaVVKernelFactory.cpp) and its kernels (platforms/cuda/src/
// kernels/*.cu) -- all compiled IN PLACE by oracle/Makefile (target `refhost`, output oracle/_ref/libvvref_host_<precision>.so), against
// the stand-in OpenMM / CUDA-platform headers of compat/ and oracle/refhost/ (host-memory "device" arrays, launches routed to the
// CPU-compiled kernels by ref_host_kernels.cpp).  What this executes that no other build does: the reference HOST code of the platform
// layer -- table building, degree-of-freedom accounting and thermostat masses (CudaVVKernels.cpp:462-667), the per-call constants
// (fscale, randFactor, efscale, hard-wall scale, 1/total mass), the launch order and arguments of every step, the blocking
// kinetic-energy download -> propagateNHChain -> upload of scaleVelocity (:670-754).  Stand-in headers => a cross-check, not "the
// reference as shipped" (DESIGN.md section 2).  Forces are whatever the caller put into the force buffer (static): the force
// provider is not part of the path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#include "OpenMMCompat.h"         // (everything standard / HIP first: the access override below must only see the reference's classes)

#define private public            // test-only: read the thermostat constants the kernel classes keep private
#define protected public
#include "CudaVVKernels.h"
#include "CudaVVKernelSources.h"
#include "openmm/VVIntegrator.h"
#undef private
#undef protected

#include <cstdio>
#include <cstring>
#include <sstream>

#include "openmm/internal/ContextImpl.h"

using namespace OpenMM;

extern "C" void registerKernelFactories();       // the reference's own (CudaVVKernelFactory.cpp:40)

// what the generated CudaVVKernelSources would hold: here only the names of the translation units (oracle/refhost/CudaVVKernelSources.h)
namespace OpenMM {
const std::string CudaVVKernelSources::vectorOps = "[vectorOps]";
const std::string CudaVVKernelSources::middle = "[middle]";
const std::string CudaVVKernelSources::velocityVerlet = "[velocityVerlet]";
const std::string CudaVVKernelSources::drudeNoseHoover = "[drudeNoseHoover]";
const std::string CudaVVKernelSources::drudeLangevin = "[drudeLangevin]";
const std::string CudaVVKernelSources::imageCharge = "[imageCharge]";
const std::string CudaVVKernelSources::electricField = "[electricField]";
const std::string CudaVVKernelSources::cosineAccelerate = "[cosineAccelerate]";
}  // namespace OpenMM

namespace {
struct Handle {
    System system;
    CudaPlatform::PlatformData pd;
    CudaContext* cu = nullptr;
    VVIntegrator* integrator = nullptr;
    Context* context = nullptr;
    ~Handle() { delete context; delete integrator; delete cu; }
};
CudaPlatform* the_platform() {
    static CudaPlatform* p = nullptr;
    if (!p) {
        p = new CudaPlatform();
        Platform::registerPlatform(p);
        registerKernelFactories();               // Platform::getPlatformByName("CUDA") -> CudaVVKernelFactory for the seven kernel names
    }
    return p;
}
}  // namespace

extern "C" {

int vvrh_precision_is_double(void);              // ref_host_kernels side knows real / mixed

// params: temperature, frequency, drudeTemperature, drudeFrequency, stepSize, maxDrudeDistance, friction (< 0: auto), drudeFriction (< 0: default),
//         mirror, electricField, cosAcceleration;  iparams: numChains, loopsPerStep, middle, useCOM (-1 auto), cmm, useDouble, useMixed
void* vvrh_create(int n, const double* masses, int nmol, const int* mol_id, int ndrude, const int* drude, int ncons, const int* cons,
                  int nld, const int* ld, int nimg, const int* img, int nel, const int* el, const double* params, const int* iparams,
                  const double* box, char* err, int errlen) {
    Handle* h = new Handle();
    try {
        for (int i = 0; i < n; i++) h->system.addParticle(masses[i]);
        for (int i = 0; i < ncons; i++) h->system.addConstraint(cons[2 * i], cons[2 * i + 1], 0.1);
        if (ndrude > 0) {
            DrudeForce* f = new DrudeForce();
            for (int i = 0; i < ndrude; i++) f->addParticle(drude[2 * i], drude[2 * i + 1], -1, -1, -1, -1.0, 1e-3, 1.0, 1.0);
            h->system.addForce(f);
        }
        if (iparams[4]) h->system.addForce(new CMMotionRemover());
        h->integrator = new VVIntegrator(params[0], params[1], params[2], params[3], params[4], iparams[0], iparams[1]);
        VVIntegrator& it = *h->integrator;
        it.setMaxDrudeDistance(params[5]);
        if (params[6] >= 0) it.setFriction(params[6]);
        if (params[7] >= 0) it.setDrudeFriction(params[7]);      // (the reference's setter also switches the friction auto-set off: VVIntegrator.h:229-232)
        it.setMirrorLocation(params[8]);
        it.setElectricField(params[9]);
        it.setCosAcceleration(params[10]);
        it.setUseMiddleScheme(iparams[2] != 0);
        if (iparams[3] >= 0) it.setUseCOMTempGroup(iparams[3] != 0);
        for (int i = 0; i < nld; i++) it.addParticleLangevin(ld[i]);
        for (int i = 0; i < nimg; i++) it.addImagePair(img[2 * i], img[2 * i + 1]);
        for (int i = 0; i < nel; i++) it.addParticleElectrolyte(el[i]);
        h->cu = new CudaContext(n, iparams[5] != 0, iparams[6] != 0, h->pd);
        h->cu->setPeriodicBoxSize(box[0], box[1], box[2]);
        h->pd.contexts.push_back(h->cu);
        h->context = new Context(h->system, it, *the_platform());
        std::vector<std::vector<int> > molecules((size_t) nmol);
        for (int i = 0; i < n; i++) molecules.at((size_t) mol_id[i]).push_back(i);
        h->context->getImpl().setMolecules(molecules);
        h->context->getImpl().setPlatformData(&h->pd);
        h->context->initialize();                 // VVIntegrator::initialize -> Cuda*Kernel::initialize (the reference's, all of it)
        return h;
    } catch (const std::exception& e) {
        if (err && errlen > 0) std::snprintf(err, (size_t) errlen, "%s", e.what());
        h->context = nullptr;
        return nullptr;
    }
}

// Parameter changes between steps, through the reference's own setters: which = 0 cosAcceleration, 1 stepSize, 2 temperature; 3 = the periodic box (cubic edge)
int vvrh_set(void* handle, int which, double value) {
    Handle* h = (Handle*) handle;
    VVIntegrator& it = *h->integrator;
    switch (which) {
        case 0: it.setCosAcceleration(value); return 0;
        case 1: it.setStepSize(value); return 0;
        case 2: it.setTemperature(value); return 0;
        case 3: h->cu->setPeriodicBoxSize(value, value, value); return 0;
        default: return -1;
    }
}

// which: 0 velm, 1 posq, 2 posqCorrection, 3 force, 4 random (float4[count]: allocates the buffer on first upload)
int vvrh_upload(void* handle, int which, const void* src, long long count) {
    Handle* h = (Handle*) handle;
    CudaContext& cu = *h->cu;
    if (which == 4) {
        CudaArray& r = cu.getIntegrationUtilities().getRandom();
        if (r.getSize() != (size_t) count) r.initialize((size_t) count, 16, "random");
        r.upload(src);
        return 0;
    }
    CudaArray* a[4] = {&cu.getVelm(), &cu.getPosq(), &cu.getPosqCorrection(), &cu.getForce()};
    a[which]->upload(src);
    return 0;
}
int vvrh_download(void* handle, int which, void* dst) {
    Handle* h = (Handle*) handle;
    CudaContext& cu = *h->cu;
    CudaArray* a[4] = {&cu.getVelm(), &cu.getPosq(), &cu.getPosqCorrection(), &cu.getForce()};
    a[which]->download(dst);
    return 0;
}

int vvrh_step(void* handle, int steps, char* err, int errlen) {
    Handle* h = (Handle*) handle;
    try {
        h->integrator->step(steps);
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen > 0) std::snprintf(err, (size_t) errlen, "%s", e.what());
        return -1;
    }
}

// The thermostat constants CudaModifyDrudeNoseKernel::initialize derived (CudaVVKernels.cpp:505-594) and its chain state; returns
// the number of temperature groups, or 0 when the configuration has no Nose-Hoover kernel.  eta_mass / eta / eta_dot: [3][16].
int vvrh_thermostat(void* handle, double* dof, double* nkbt, double* eta_mass, double* eta, double* eta_dot, double* ke2, double* vscale,
                    int* counts) {
    Handle* h = (Handle*) handle;
    if (h->integrator->particlesNH.empty()) return 0;
    CudaModifyDrudeNoseKernel& k = h->integrator->nhKernel.getAs<CudaModifyDrudeNoseKernel>();
    for (int g = 0; g < 3; g++) dof[g] = k.tempGroupDof[g];
    for (int g = 0; g < k.numTempGroup; g++) {
        nkbt[g] = k.tempGroupNkbT[g];
        for (size_t i = 0; i < k.etaMass[g].size() && i < 16; i++) { eta_mass[16 * g + i] = k.etaMass[g][i]; eta[16 * g + i] = k.eta[g][i]; }
        for (size_t i = 0; i < k.etaDot[g].size() && i < 16; i++) eta_dot[16 * g + i] = k.etaDot[g][i];
        if ((int) k.kineticEnergiesNHVec.size() > g) ke2[g] = k.kineticEnergiesNHVec[g];
        if ((int) k.vscaleFactorsNHVec.size() > g) vscale[g] = k.vscaleFactorsNHVec[g];
    }
    counts[0] = (int) k.particlesNHVec.size(); counts[1] = (int) k.moleculesNHVec.size();
    counts[2] = (int) k.normalParticlesNHVec.size(); counts[3] = (int) k.pairParticlesNHVec.size();
    return k.numTempGroup;
}

// names of the kernels launched so far, joined by ',' (the reference's launch order); clears the record
int vvrh_launches(void* handle, char* out, int outlen) {
    Handle* h = (Handle*) handle;
    std::ostringstream s;
    for (size_t i = 0; i < h->cu->launches.size(); i++) s << (i ? "," : "") << h->cu->launches[i];
    const int n = (int) h->cu->launches.size();
    h->cu->launches.clear();
    const std::string r = s.str();
    if ((int) r.size() + 1 > outlen) return -2;
    std::memcpy(out, r.c_str(), r.size() + 1);
    return n;
}

int vvrh_viscosity(void* handle, double* vmax, double* inv_vis) {
    Handle* h = (Handle*) handle;
    std::vector<double> v = h->integrator->getViscosity();
    *vmax = v[0]; *inv_vis = v[1];
    return 0;
}

double vvrh_time(void* handle) { return ((Handle*) handle)->cu->getTime(); }
void vvrh_destroy(void* handle) { delete (Handle*) handle; }

}  // extern "C"
