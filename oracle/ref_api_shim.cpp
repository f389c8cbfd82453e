// oracle/ref_api_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C entry points around the reference's OWN host class, compiled from where it lies:
//     /root/reference/openmmapi/src/VVIntegrator.cpp  +  openmmapi/include/openmm/{VVIntegrator,VVKernels}.h
// (oracle/Makefile target `refapi`, output oracle/_ref/libvvref_api.so; nothing of the reference is copied).  OpenMM itself is
// not in this image, so the translation unit is compiled against the stand-in headers in compat/ -- which is why, by this tier's
// rule, the build does NOT count as "the reference as shipped" and pins nothing formally (DESIGN.md section 2 says so).  What it
// does give: the reference's own statements of
//   * propagateNHChain                     (VVIntegrator.cpp:340-376)   -> vvref_api_propagate
//   * initialize: COM-group / friction auto rules, particle -> molecule map, molecule masses, the NH / Langevin / image
//     partition and its conflict exceptions  (VVIntegrator.cpp:92-188)   -> vvref_api_create / vvref_api_tables
//   * stepMiddle / stepVV call order incl. the forcesAreValid caching (VVIntegrator.cpp:232-338) -> vvref_api_trace
// executed here, against which the oracle's restatement (vvo_propagate_nh_chain, oracle.build_tables) and the product's
// vv::analyze / openmmapi::VVIntegrator are compared (tests/test_ref_api.py) and from which tests/golden/refapi_*.npz come.
// The seven kernels are mocks that only record that they were called.
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "openmm/VVIntegrator.h"      // the reference's (include path: $(REF)/openmmapi/include before compat/)
#include "openmm/VVKernels.h"
#include "openmm/internal/ContextImpl.h"

using namespace OpenMM;

namespace {

struct Trace { std::vector<std::string> calls; };

#define MOCK_STEP(NAME) void NAME(ContextImpl&, const VVIntegrator&) override { trace->calls.push_back(#NAME); }

class MockMiddle : public IntegrateMiddleStepKernel {
public:
    MockMiddle(std::string n, const Platform& p, Trace* t) : IntegrateMiddleStepKernel(n, p), trace(t) {}
    void initialize(const System&, const VVIntegrator&, const DrudeForce*) override { trace->calls.push_back("init:IntegrateMiddleStep"); }
    MOCK_STEP(firstIntegrate) MOCK_STEP(resetExtraForce) MOCK_STEP(secondIntegrate)
    double computeKineticEnergy(ContextImpl&, const VVIntegrator&) override { trace->calls.push_back("computeKineticEnergy"); return 0.0; }
    Trace* trace;
};
class MockVV : public IntegrateVVStepKernel {
public:
    MockVV(std::string n, const Platform& p, Trace* t) : IntegrateVVStepKernel(n, p), trace(t) {}
    void initialize(const System&, const VVIntegrator&, const DrudeForce*) override { trace->calls.push_back("init:IntegrateVVStep"); }
    MOCK_STEP(firstIntegrate) MOCK_STEP(resetExtraForce) MOCK_STEP(secondIntegrate)
    double computeKineticEnergy(ContextImpl&, const VVIntegrator&) override { trace->calls.push_back("computeKineticEnergy"); return 0.0; }
    Trace* trace;
};
class MockNose : public ModifyDrudeNoseKernel {
public:
    MockNose(std::string n, const Platform& p, Trace* t) : ModifyDrudeNoseKernel(n, p), trace(t) {}
    void initialize(const System&, const VVIntegrator&, const DrudeForce*) override { trace->calls.push_back("init:ModifyDrudeNose"); }
    MOCK_STEP(scaleVelocity)
    Trace* trace;
};
class MockLangevin : public ModifyDrudeLangevinKernel {
public:
    MockLangevin(std::string n, const Platform& p, Trace* t) : ModifyDrudeLangevinKernel(n, p), trace(t) {}
    void initialize(const System&, const VVIntegrator&, const DrudeForce*, Kernel&) override { trace->calls.push_back("init:ModifyDrudeLangevin"); }
    MOCK_STEP(applyLangevinForce)
    Trace* trace;
};
class MockImage : public ModifyImageChargeKernel {
public:
    MockImage(std::string n, const Platform& p, Trace* t) : ModifyImageChargeKernel(n, p), trace(t) {}
    void initialize(const System&, const VVIntegrator&) override { trace->calls.push_back("init:ModifyImageCharge"); }
    MOCK_STEP(updateImagePositions)
    Trace* trace;
};
class MockField : public ModifyElectricFieldKernel {
public:
    MockField(std::string n, const Platform& p, Trace* t) : ModifyElectricFieldKernel(n, p), trace(t) {}
    void initialize(const System&, const VVIntegrator&, Kernel&) override { trace->calls.push_back("init:ModifyElectricField"); }
    MOCK_STEP(applyElectricForce)
    Trace* trace;
};
class MockCos : public ModifyCosineAccelerateKernel {
public:
    MockCos(std::string n, const Platform& p, Trace* t) : ModifyCosineAccelerateKernel(n, p), trace(t) {}
    void initialize(const System&, const VVIntegrator&, Kernel&) override { trace->calls.push_back("init:ModifyCosineAccelerate"); }
    MOCK_STEP(applyCosineForce) MOCK_STEP(calcVelocityBias) MOCK_STEP(removeVelocityBias) MOCK_STEP(restoreVelocityBias)
    void calcViscosity(ContextImpl&, const VVIntegrator&, double& vMax, double& invVis) override { trace->calls.push_back("calcViscosity"); vMax = 0; invVis = 0; }
    Trace* trace;
};

class MockFactory : public KernelFactory {
public:
    explicit MockFactory(Trace* t) : trace(t) {}
    KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl&) const override {
        if (name == IntegrateMiddleStepKernel::Name()) return new MockMiddle(name, platform, trace);
        if (name == IntegrateVVStepKernel::Name()) return new MockVV(name, platform, trace);
        if (name == ModifyDrudeNoseKernel::Name()) return new MockNose(name, platform, trace);
        if (name == ModifyDrudeLangevinKernel::Name()) return new MockLangevin(name, platform, trace);
        if (name == ModifyImageChargeKernel::Name()) return new MockImage(name, platform, trace);
        if (name == ModifyElectricFieldKernel::Name()) return new MockField(name, platform, trace);
        if (name == ModifyCosineAccelerateKernel::Name()) return new MockCos(name, platform, trace);
        throw OpenMMException("unknown kernel " + name);
    }
    Trace* trace;
};
class MockPlatform : public Platform {
public:
    const std::string& getName() const override { static const std::string n = "Mock"; return n; }
};

struct Handle {
    Trace trace;
    System system;
    MockPlatform platform;
    MockFactory factory{&trace};
    VVIntegrator* integrator = nullptr;
    Context* context = nullptr;
    ~Handle() { delete context; delete integrator; }
};

void force_callback(ContextImpl&, void* user) { ((Trace*) user)->calls.push_back("calcForcesAndEnergy"); }

}  // namespace

extern "C" {

// VVIntegrator::propagateNHChain on caller-owned arrays: eta[nchains], eta_dot[nchains + 1], eta_dotdot[nchains], eta_mass[nchains]
int vvref_api_propagate(double step_size, int loops_per_step, int nchains, double* eta, double* eta_dot, double* eta_dotdot,
                        const double* eta_mass, double ke2, double ke2_target, double t_target, double* factor) {
    try {
        VVIntegrator it(300.0, 10.0, 1.0, 40.0, step_size, nchains, loops_per_step);
        std::vector<double> e(eta, eta + nchains), ed(eta_dot, eta_dot + nchains + 1), edd(eta_dotdot, eta_dotdot + nchains), em(eta_mass, eta_mass + nchains);
        double f = 0;
        it.propagateNHChain(e, ed, edd, em, ke2, ke2_target, t_target, f);
        std::memcpy(eta, e.data(), sizeof(double) * nchains);
        std::memcpy(eta_dot, ed.data(), sizeof(double) * (nchains + 1));
        std::memcpy(eta_dotdot, edd.data(), sizeof(double) * nchains);
        *factor = f;
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

// Builds a System (+ DrudeForce, + CMMotionRemover), a reference VVIntegrator and a Context on the mock platform, and runs the
// reference's VVIntegrator::initialize.  use_com: -1 = leave the auto rule, 0 / 1 = setUseCOMTempGroup; friction < 0 = auto rule.
// Returns NULL and the exception text in err when the reference throws.
void* vvref_api_create(int n, const double* masses, int nmol, const int* mol_id, int ndrude, const int* drude, int ncons, const int* cons,
                       int cmm, int nld, const int* ld, int nimg, const int* img, int nel, const int* el, double cos_acceleration,
                       int middle, int use_com, double friction, char* err, int errlen) {
    Handle* h = new Handle();
    try {
        for (int i = 0; i < n; i++) h->system.addParticle(masses[i]);
        for (int i = 0; i < ncons; i++) h->system.addConstraint(cons[2 * i], cons[2 * i + 1], 0.1);
        if (ndrude > 0) {
            DrudeForce* f = new DrudeForce();
            for (int i = 0; i < ndrude; i++) f->addParticle(drude[2 * i], drude[2 * i + 1], -1, -1, -1, -1.0, 1e-3, 1.0, 1.0);
            h->system.addForce(f);
        }
        if (cmm) h->system.addForce(new CMMotionRemover());
        h->integrator = new VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001);
        h->integrator->setUseMiddleScheme(middle != 0);
        h->integrator->setCosAcceleration(cos_acceleration);
        if (use_com >= 0) h->integrator->setUseCOMTempGroup(use_com != 0);
        if (friction >= 0) h->integrator->setFriction(friction);
        for (int i = 0; i < nld; i++) h->integrator->addParticleLangevin(ld[i]);
        for (int i = 0; i < nimg; i++) h->integrator->addImagePair(img[2 * i], img[2 * i + 1]);
        for (int i = 0; i < nel; i++) h->integrator->addParticleElectrolyte(el[i]);
        for (const std::string& name : {IntegrateMiddleStepKernel::Name(), IntegrateVVStepKernel::Name(), ModifyDrudeNoseKernel::Name(),
                                        ModifyDrudeLangevinKernel::Name(), ModifyImageChargeKernel::Name(), ModifyElectricFieldKernel::Name(),
                                        ModifyCosineAccelerateKernel::Name()})
            h->platform.registerKernelFactory(name, &h->factory);
        h->context = new Context(h->system, *h->integrator, h->platform);
        std::vector<std::vector<int> > molecules((size_t) nmol);
        for (int i = 0; i < n; i++) molecules.at((size_t) mol_id[i]).push_back(i);
        h->context->getImpl().setMolecules(molecules);
        h->context->getImpl().setForceCallback(force_callback, &h->trace);
        h->context->initialize();                     // -> the reference's VVIntegrator::initialize (VVIntegrator.cpp:92-188)
        return h;
    } catch (const std::exception& e) {
        if (err && errlen > 0) std::snprintf(err, (size_t) errlen, "%s", e.what());
        h->context = nullptr;                          // Context's destructor would call cleanup() on a half-built integrator: leak it
        return nullptr;
    }
}

// What initialize() decided.  Arrays must hold num_particles entries (mol_inv_mass: num_molecules).
int vvref_api_tables(void* handle, int* particles_nh, int* n_nh, int* molecules_nh, int* n_mol_nh, int* particle_mol_id,
                     double* mol_inv_mass, int* use_com, double* friction, int* is_ld, int* is_image) {
    Handle* h = (Handle*) handle;
    const VVIntegrator& it = *h->integrator;
    const std::vector<int>& pnh = it.getParticlesNH();
    const std::vector<int>& mnh = it.getMoleculesNH();
    *n_nh = (int) pnh.size();
    *n_mol_nh = (int) mnh.size();
    for (size_t i = 0; i < pnh.size(); i++) particles_nh[i] = pnh[i];
    for (size_t i = 0; i < mnh.size(); i++) molecules_nh[i] = mnh[i];
    const int n = h->system.getNumParticles();
    for (int i = 0; i < n; i++) { particle_mol_id[i] = it.getParticleMolId(i); is_ld[i] = it.isParticleLD(i) ? 1 : 0; is_image[i] = it.isParticleImage(i) ? 1 : 0; }
    for (int m = 0; m < it.getNumMolecules(); m++) mol_inv_mass[m] = it.getMoleculeInvMass(m);
    *use_com = it.getUseCOMTempGroup() ? 1 : 0;
    *friction = it.getFriction();
    return it.getNumMolecules();
}

// Runs the reference's step(steps) on the mock kernels and returns the recorded call sequence (names joined by ',') -- the
// init:* entries of initialize() included, so that the kernels created for a configuration are visible too.
int vvref_api_trace(void* handle, int steps, int query_energy_after, char* out, int outlen) {
    Handle* h = (Handle*) handle;
    try {
        h->integrator->step(steps);
        if (query_energy_after) {                      // State::getKineticEnergy() -> Integrator::computeKineticEnergy: invalidates the forces (VVIntegrator.cpp:212-223)
            struct Peek : VVIntegrator { using VVIntegrator::computeKineticEnergy; };
            (h->integrator->*(&Peek::computeKineticEnergy))();
            h->integrator->step(1);
        }
    } catch (const std::exception&) {
        return -1;
    }
    std::ostringstream s;
    for (size_t i = 0; i < h->trace.calls.size(); i++) s << (i ? "," : "") << h->trace.calls[i];
    const std::string r = s.str();
    if ((int) r.size() + 1 > outlen) return -2;
    std::memcpy(out, r.c_str(), r.size() + 1);
    return (int) h->trace.calls.size();
}

void vvref_api_destroy(void* handle) { delete (Handle*) handle; }

}  // extern "C"
