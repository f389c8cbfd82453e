// VVIntegrator -- platform-neutral half of the plugin.  Behavioural contract = the reference's
// openmmapi/src/VVIntegrator.cpp ("REF" below); the code is new (hash sets instead of O(N) std::find scans,
// one thermostat()/extraForces() helper shared by both schemes, an optional fused path).
#include "openmm/VVIntegrator.h"

#include <cmath>
#include <cstdio>

#include "openmm/Context.h"
#include "openmm/DrudeForce.h"
#include "openmm/OpenMMException.h"
#include "openmm/VVKernels.h"
#include "openmm/internal/ContextImpl.h"

using namespace OpenMM;

namespace {
const double kBoltz = (1.380649e-23 * 6.02214076e23) / 1000.0;   // kJ/mol/K (SimTKOpenMMRealType.h: RGAS/KILO)
}

VVIntegrator::VVIntegrator(double T, double freq, double drudeT, double drudeFreq, double dt, int chains, int loops)   // REF:46-70
    : temperature(T), frequency(freq), drudeTemperature(drudeT), drudeFrequency(drudeFreq), maxDrudeDistance(0), friction(5.0),
      drudeFriction(20.0), mirrorLocation(0), electricField(0), cosAcceleration(0), loopsPerStep(loops), numNHChains(chains),
      randomNumberSeed(0), useCOMTempGroup(false), autoSetCOMTempGroup(true), autoSetFriction(true), useMiddleScheme(true),
      debugEnabled(false), forcesAreValid(false) {
    setStepSize(dt);
    setConstraintTolerance(1e-5);
}

VVIntegrator::~VVIntegrator() {}

int VVIntegrator::addImagePair(int image, int parent) {          // REF:76-80
    particlesImage.push_back(image);
    imageSet.insert(image);
    imagePairs.emplace_back(image, parent);
    return (int) imagePairs.size();
}

double VVIntegrator::getMoleculeInvMass(int molid) const {
    if (molid < 0 || molid >= (int) moleculeInvMasses.size()) throw OpenMMException("getMoleculeInvMass: index out of range");
    return moleculeInvMasses[molid];
}

int VVIntegrator::getParticleMolId(int particle) const {
    if (particle < 0 || particle >= (int) particleMolId.size()) throw OpenMMException("getParticleMolId: index out of range");
    return particleMolId[particle];
}

void VVIntegrator::initialize(ContextImpl& ctx) {                // REF:92-188
    if (owner != NULL && &ctx.getOwner() != owner) throw OpenMMException("This Integrator is already bound to a context");
    const System& system = ctx.getSystem();
    const DrudeForce* drude = NULL;
    for (int i = 0; i < system.getNumForces(); i++) {
        const DrudeForce* f = dynamic_cast<const DrudeForce*>(&system.getForce(i));
        if (f == NULL) continue;
        if (drude != NULL) throw OpenMMException("The System contains multiple DrudeForces");
        drude = f;
    }
    const bool polarizable = drude != NULL && drude->getNumParticles() > 0;   // REF:106-121
    if (autoSetCOMTempGroup) useCOMTempGroup = polarizable;
    else if (useCOMTempGroup != polarizable)
        std::printf(polarizable ? "WARNING: You are not using COM temperature group for Drude model\n"
                                : "WARNING: You are using COM temperature group for non-Drude model\n");
    if (autoSetFriction) friction = polarizable ? 5.0 : 1.0;

    const int n = system.getNumParticles();
    const std::vector<std::vector<int> >& molecules = ctx.getMolecules();
    particleMolId.assign(n, -1);
    for (size_t m = 0; m < molecules.size(); m++)
        for (int p : molecules[m]) particleMolId[p] = (int) m;
    moleculeMasses.assign(molecules.size(), 0.0);
    for (int i = 0; i < n; i++) moleculeMasses[particleMolId[i]] += system.getParticleMass(i);
    moleculeInvMasses.clear();
    for (double m : moleculeMasses) moleculeInvMasses.push_back(1.0 / m);

    // thermostat partition: everything that is neither Langevin nor an image is Nose-Hoover (REF:138-145)
    particlesNH.clear(); moleculesNH.clear(); nhSet.clear();
    std::unordered_set<int> nhMolecules;
    for (int i = 0; i < n; i++) {
        if (isParticleLD(i) || isParticleImage(i)) continue;
        particlesNH.push_back(i);
        nhSet.insert(i);
        if (nhMolecules.insert(particleMolId[i]).second) moleculesNH.push_back(particleMolId[i]);
    }
    for (int p : particlesLD)                                   // REF:146-151
        if (nhMolecules.count(particleMolId[p])) throw OpenMMException("NH and Langevin thermostat cannot be applied on the same molecule");
    if (!particlesLD.empty() && cosAcceleration != 0)           // REF:154-155
        throw OpenMMException("Langevin thermostat and periodic perturbation shouldn't be used together");

    context = &ctx;
    owner = &ctx.getOwner();
    Platform& platform = ctx.getPlatform();                     // REF:160-187: step kernel first, the others may borrow from it
    if (useMiddleScheme) {
        vvKernel = platform.createKernel(IntegrateMiddleStepKernel::Name(), ctx);
        vvKernel.getAs<IntegrateMiddleStepKernel>().initialize(system, *this, drude);
    } else {
        vvKernel = platform.createKernel(IntegrateVVStepKernel::Name(), ctx);
        vvKernel.getAs<IntegrateVVStepKernel>().initialize(system, *this, drude);
    }
    if (!particlesNH.empty()) {
        nhKernel = platform.createKernel(ModifyDrudeNoseKernel::Name(), ctx);
        nhKernel.getAs<ModifyDrudeNoseKernel>().initialize(system, *this, drude);
    }
    if (!particlesLD.empty()) {
        ldKernel = platform.createKernel(ModifyDrudeLangevinKernel::Name(), ctx);
        ldKernel.getAs<ModifyDrudeLangevinKernel>().initialize(system, *this, drude, vvKernel);
    }
    if (!particlesImage.empty()) {
        imgKernel = platform.createKernel(ModifyImageChargeKernel::Name(), ctx);
        imgKernel.getAs<ModifyImageChargeKernel>().initialize(system, *this);
    }
    if (!particlesElectrolyte.empty()) {
        efKernel = platform.createKernel(ModifyElectricFieldKernel::Name(), ctx);
        efKernel.getAs<ModifyElectricFieldKernel>().initialize(system, *this, vvKernel);
    }
    if (cosAcceleration != 0) {
        ppKernel = platform.createKernel(ModifyCosineAccelerateKernel::Name(), ctx);
        ppKernel.getAs<ModifyCosineAccelerateKernel>().initialize(system, *this, vvKernel);
    }
}

void VVIntegrator::cleanup() {                                   // REF:190-197
    vvKernel = nhKernel = ldKernel = imgKernel = efKernel = ppKernel = Kernel();
}

std::vector<std::string> VVIntegrator::getKernelNames() {        // REF:199-209
    return {IntegrateVVStepKernel::Name(), IntegrateMiddleStepKernel::Name(), ModifyDrudeNoseKernel::Name(),
            ModifyDrudeLangevinKernel::Name(), ModifyImageChargeKernel::Name(), ModifyElectricFieldKernel::Name(),
            ModifyCosineAccelerateKernel::Name()};
}

double VVIntegrator::computeKineticEnergy() {                    // REF:211-221: an energy query may clobber the forces
    forcesAreValid = false;
    return vvKernel.getAs<VVStepKernelBase>().computeKineticEnergy(*context, *this);
}

void VVIntegrator::step(int steps) {                             // REF:223-230
    if (context == NULL) throw OpenMMException("This Integrator is not bound to a context!");
    if (useMiddleScheme) stepMiddle(steps);
    else stepVV(steps);
}

void VVIntegrator::extraForces() {                               // REF:238-245 == 316-323
    if (particlesLD.empty() && particlesElectrolyte.empty() && cosAcceleration == 0) return;
    vvKernel.getAs<VVStepKernelBase>().resetExtraForce(*context, *this);
    if (!particlesLD.empty()) ldKernel.getAs<ModifyDrudeLangevinKernel>().applyLangevinForce(*context, *this);
    if (!particlesElectrolyte.empty()) efKernel.getAs<ModifyElectricFieldKernel>().applyElectricForce(*context, *this);
    if (cosAcceleration != 0) ppKernel.getAs<ModifyCosineAccelerateKernel>().applyCosineForce(*context, *this);
}

void VVIntegrator::thermostat() {                                // REF:251-260 == 295-304 == 327-336
    if (particlesNH.empty()) return;
    const bool biased = cosAcceleration != 0;
    if (biased) {
        ppKernel.getAs<ModifyCosineAccelerateKernel>().calcVelocityBias(*context, *this);
        ppKernel.getAs<ModifyCosineAccelerateKernel>().removeVelocityBias(*context, *this);
    }
    nhKernel.getAs<ModifyDrudeNoseKernel>().scaleVelocity(*context, *this);
    if (biased) ppKernel.getAs<ModifyCosineAccelerateKernel>().restoreVelocityBias(*context, *this);
}

void VVIntegrator::stepMiddle(int steps) {                       // REF:232-270
    VVStepKernelBase& vv = vvKernel.getAs<VVStepKernelBase>();
    FusedVVStepKernel* fused = dynamic_cast<FusedVVStepKernel*>(&vvKernel.getImpl());
    const bool fuse = fused != NULL && fused->canFuse(*context, *this);
    for (int i = 0; i < steps; ++i) {
        context->updateContextState();
        context->calcForcesAndEnergy(true, false);
        if (fuse) { fused->fusedMiddleStep(*context, *this); continue; }
        extraForces();
        vv.firstIntegrate(*context, *this);      // full kick (+ velocity constraints) + first half drift
        thermostat();
        vv.secondIntegrate(*context, *this);     // second half drift (+ constraints) + hard wall
        if (!particlesImage.empty()) imgKernel.getAs<ModifyImageChargeKernel>().updateImagePositions(*context, *this);
    }
}

void VVIntegrator::stepVV(int steps) {                           // REF:272-338
    VVStepKernelBase& vv = vvKernel.getAs<VVStepKernelBase>();
    FusedVVStepKernel* fused = dynamic_cast<FusedVVStepKernel*>(&vvKernel.getImpl());
    const bool fuse = fused != NULL && fused->canFuse(*context, *this);
    for (int i = 0; i < steps; ++i) {
        // the extra (Langevin / field / cos) forces live in their own buffer, so invalidating the force-field
        // forces (barostat move, energy query) does not lose them: REF:275-292
        if (context->updateContextState()) forcesAreValid = false;
        if (!forcesAreValid) { context->calcForcesAndEnergy(true, false); forcesAreValid = true; }
        if (fuse) fused->fusedVVFirstHalf(*context, *this);
        else {
            thermostat();
            vv.firstIntegrate(*context, *this);  // half kick + drift (+ constraints) + hard wall
            if (!particlesImage.empty()) imgKernel.getAs<ModifyImageChargeKernel>().updateImagePositions(*context, *this);
        }
        context->calcForcesAndEnergy(true, false);
        forcesAreValid = true;
        if (fuse) fused->fusedVVSecondHalf(*context, *this);
        else {
            extraForces();
            vv.secondIntegrate(*context, *this); // half kick (+ velocity constraints)
            thermostat();
        }
    }
}

// Host statement of the chain half-step, kept for API parity (REF:340-376); the HIP backend evaluates the same
// recurrence on the device (csrc/vv_kernels.hip: propagate_preloaded).  NB dt2 = stepSize/loops/2: one call advances
// the thermostat by half a step, the middle scheme calls it once per step (quirk Q1 in SURVEY.md).
void VVIntegrator::propagateNHChain(std::vector<double>& eta, std::vector<double>& etaDot, std::vector<double>& etaDotDot,
                                    const std::vector<double>& etaMass, const double& ke2, const double& ke2Target,
                                    const double& tTarget, double& scale) const {
    const int nc = numNHChains;
    const double h2 = getStepSize() / loopsPerStep / 2, h4 = h2 / 2, h8 = h4 / 2, kT = kBoltz * tTarget;
    double damp = 1.0;
    scale = 1.0;
    etaDotDot[0] = (ke2 - ke2Target) / etaMass[0];
    for (int loop = 0; loop < loopsPerStep; loop++) {
        for (int k = nc - 1; k >= 0; k--) {          // top of the chain down to the particle thermostat
            damp = std::exp(-h8 * etaDot[k + 1]);
            etaDot[k] = (etaDot[k] * damp + etaDotDot[k] * h4) * damp;
        }
        scale *= std::exp(-h2 * etaDot[0]);
        for (int k = 0; k < nc; k++) eta[k] += h2 * etaDot[k];
        etaDotDot[0] = (ke2 * scale * scale - ke2Target) / etaMass[0];
        etaDot[0] = (etaDot[0] * damp + etaDotDot[0] * h4) * damp;     // same damping factor as the last sweep step
        for (int k = 1; k < nc; k++) {               // and back up
            damp = std::exp(-h8 * etaDot[k + 1]);
            etaDotDot[k] = (etaMass[k - 1] * etaDot[k - 1] * etaDot[k - 1] - kT) / etaMass[k];
            etaDot[k] = (etaDot[k] * damp + etaDotDot[k] * h4) * damp;
        }
    }
}

std::vector<double> VVIntegrator::getViscosity() {               // REF:378-383
    double vMax = 0, invVis = 0;
    if (cosAcceleration != 0) ppKernel.getAs<ModifyCosineAccelerateKernel>().calcViscosity(*context, *this, vMax, invVis);
    return {vMax, invVis};
}
