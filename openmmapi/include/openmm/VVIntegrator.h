#ifndef VVHIP_OPENMM_VV_INTEGRATOR_H_
#define VVHIP_OPENMM_VV_INTEGRATOR_H_
// OpenMM::VVIntegrator -- public surface of the reference class (openmmapi/include/openmm/VVIntegrator.h:49-507):
// same constructor, same getters/setters, same protected Integrator overrides, so user code and the SWIG interface
// (python/velocityverletplugin.i) compile against either.  Implementation in ../../src/VVIntegrator.cpp is new.
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "openmm/Integrator.h"
#include "openmm/Kernel.h"
#include "openmm/State.h"
#include "openmm/internal/windowsExportDrude.h"

namespace OpenMM {

class OPENMM_EXPORT_DRUDE VVIntegrator : public Integrator {
public:
    // temperature [K], frequency [1/ps], drudeTemperature [K], drudeFrequency [1/ps], stepSize [ps]
    VVIntegrator(double temperature, double frequency, double drudeTemperature, double drudeFrequency, double stepSize,
                 int numNHChains = 3, int loopsPerStep = 1);
    virtual ~VVIntegrator();

    // ---- thermostat
    double getTemperature() const { return temperature; }
    void setTemperature(double v) { temperature = v; }
    double getFrequency() const { return frequency; }
    void setFrequency(double v) { frequency = v; }
    double getDrudeTemperature() const { return drudeTemperature; }
    void setDrudeTemperature(double v) { drudeTemperature = v; }
    double getDrudeFrequency() const { return drudeFrequency; }
    void setDrudeFrequency(double v) { drudeFrequency = v; }
    int getNumNHChains() const { return numNHChains; }
    void setNumNHChains(int n) { numNHChains = n; }
    int getLoopsPerStep() const { return loopsPerStep; }
    void setLoopsPerStep(int n) { loopsPerStep = n; }
    bool getUseCOMTempGroup() const { return useCOMTempGroup; }
    void setUseCOMTempGroup(bool use) { useCOMTempGroup = use; autoSetCOMTempGroup = false; }
    double getMaxDrudeDistance() const { return maxDrudeDistance; }
    void setMaxDrudeDistance(double d) { maxDrudeDistance = d; }
    const bool& getUseMiddleScheme() const { return useMiddleScheme; }
    void setUseMiddleScheme(bool use) { useMiddleScheme = use; }

    // ---- Langevin subset
    int addParticleLangevin(int particle) { particlesLD.push_back(particle); ldSet.insert(particle); return (int) particlesLD.size(); }
    double getFriction() const { return friction; }
    void setFriction(double f) { friction = f; autoSetFriction = false; }
    double getDrudeFriction() const { return drudeFriction; }
    void setDrudeFriction(double f) { drudeFriction = f; autoSetFriction = false; }
    int getRandomNumberSeed() const { return randomNumberSeed; }
    void setRandomNumberSeed(int seed) { randomNumberSeed = seed; }

    // ---- image charges / external field (kJ/(nm e) per particle)
    int addImagePair(int image, int parent);
    const std::vector<std::pair<int, int> >& getImagePairs() const { return imagePairs; }
    double getMirrorLocation() const { return mirrorLocation; }
    void setMirrorLocation(double z) { mirrorLocation = z; }
    double getElectricField() const { return electricField; }
    void setElectricField(double e) { electricField = e; }
    int addParticleElectrolyte(int particle) { particlesElectrolyte.push_back(particle); return (int) particlesElectrolyte.size(); }
    const std::vector<int>& getParticlesElectrolyte() const { return particlesElectrolyte; }

    // ---- periodic perturbation
    double getCosAcceleration() const { return cosAcceleration; }
    void setCosAcceleration(double a) { cosAcceleration = a; }
    std::vector<double> getViscosity();          // {vMax [nm/ps], 1/viscosity}

    // ---- what initialize() derives (read by the platform kernels)
    const std::vector<int>& getParticlesNH() const { return particlesNH; }
    const std::vector<int>& getParticlesLD() const { return particlesLD; }
    const std::vector<int>& getMoleculesNH() const { return moleculesNH; }
    bool isParticleNH(int i) const { return nhSet.count(i) != 0; }
    bool isParticleLD(int i) const { return ldSet.count(i) != 0; }
    bool isParticleImage(int i) const { return imageSet.count(i) != 0; }
    int getNumMolecules() const { return (int) moleculeMasses.size(); }
    double getMoleculeInvMass(int molid) const;
    int getParticleMolId(int particle) const;

    // Nose-Hoover chain half-step for one temperature group (host reference implementation; the HIP backend runs
    // the same recurrence on the device)
    void propagateNHChain(std::vector<double>& eta, std::vector<double>& etaDot, std::vector<double>& etaDotDot,
                          const std::vector<double>& etaMass, const double& ke2, const double& ke2Target, const double& tTarget,
                          double& scale) const;

    void step(int steps);
    const bool& getDebugEnabled() const { return debugEnabled; }
    void setDebugEnabled(bool e) { debugEnabled = e; }

protected:
    void initialize(ContextImpl& context);
    void cleanup();
    void stateChanged(State::DataType) { forcesAreValid = false; }
    std::vector<std::string> getKernelNames();
    double computeKineticEnergy();
    bool kineticEnergyRequiresForce() const { return false; }
    void stepVV(int steps);
    void stepMiddle(int steps);

private:
    void extraForces();          // reset + Langevin + field + cos, in the reference's order
    void thermostat();           // bias remove -> NH scaling -> bias restore
    double temperature, frequency, drudeTemperature, drudeFrequency, maxDrudeDistance;
    double friction, drudeFriction, mirrorLocation, electricField, cosAcceleration;
    int loopsPerStep, numNHChains, randomNumberSeed;
    bool useCOMTempGroup, autoSetCOMTempGroup, autoSetFriction, useMiddleScheme, debugEnabled, forcesAreValid;
    std::vector<int> particlesNH, moleculesNH, particleMolId, particlesLD, particlesImage, particlesElectrolyte;
    std::unordered_set<int> nhSet, ldSet, imageSet;
    std::vector<double> moleculeMasses, moleculeInvMasses;
    std::vector<std::pair<int, int> > imagePairs;
    Kernel vvKernel, nhKernel, ldKernel, imgKernel, efKernel, ppKernel;
};

}  // namespace OpenMM
#endif
