#ifndef VVHIP_KERNELS_H_
#define VVHIP_KERNELS_H_
// OpenMM KernelImpl adapters of the HIP backend.  Each class implements one of the seven interfaces of
// openmm/VVKernels.h by calling the C ABI of libvvhip (include/vvhip.h); all seven share ONE vvhip_plan per
// HipContext (the reference shares only forceExtra, through getForceExtra(): CudaVVKernels.h:86-88).
// Counterpart of platforms/cuda/include/CudaVVKernels.h.
#include <functional>
#include <memory>
#include <vector>

#include "HipCompat.h"          // with a real OpenMM: "HipContext.h", "HipArray.h", "HipIntegrationUtilities.h"
#include "openmm/VVIntegrator.h"
#include "openmm/VVKernels.h"
#include "openmm/FusedVVStepKernel.h"
#include "vvhip.h"

namespace OpenMM {

// One plan + the parameters it was last told about.
class HipVVPlan {
public:
    HipVVPlan(HipContext& cu, const System& system, const VVIntegrator& integrator, const DrudeForce* force);
    ~HipVVPlan();
    vvhip_plan* get() const { return plan; }
    void syncParameters(const VVIntegrator& integrator);      // the reference reads the getters at every call
    double stepSizeOf(const VVIntegrator& integrator) const;   // integrator.getStepSize(), or the recorded one while recorded stages are replayed
    void check(int rc) const;                                  // vvhip error -> OpenMMException
    int numLangevinRandoms() const { return ldRandoms; }
    bool constraintFree() const { return noConstraints; }
    bool placesVirtualSites() const { return sitesInKernel; }      // the fused steps place the System's virtual sites (vvhip_plan_info.num_virtual_sites)
    static std::shared_ptr<HipVVPlan> find(HipContext& cu);    // the plan the step kernel created for this context
    static std::shared_ptr<HipVVPlan> create(HipContext& cu, const System&, const VVIntegrator&, const DrudeForce*);

    // ---- deferred fusion.  A host class that does not know FusedVVStepKernel -- the reference's own VVIntegrator -- calls the seven
    // kernels stage by stage (VVIntegrator.cpp:232-338), with nothing read back in between.  The adapters therefore only RECORD a stage
    // while the calls follow the sequence the reference's stepMiddle / stepVV produce for the integrator's configuration, and the stage
    // that completes the sequence launches the fused step (2 launches instead of 8).  Any other call order runs what was recorded through
    // the stage-by-stage entry points first, in order, and then the new stage: same results either way.  A recorded stage belongs to the
    // parameters and the box of the moment it was CALLED (the reference reads the getters at every call): the first recorded stage takes
    // a snapshot, a later call that finds the integrator or the box changed runs the recorded stages with the snapshot before anything
    // else, and a replay never reads the integrator again.  Off with constraints OpenMM's solver must interleave, or VVHIP_PLUGIN_DEFER=0.
    enum Stage { ST_RESET, ST_LD, ST_EF, ST_COS, ST_FIRST, ST_CALCBIAS, ST_RMBIAS, ST_SCALE, ST_RESTORE, ST_SECOND };
    // true: recorded, or the sequence is complete and the fused step has been launched -- the caller is done; false: the caller runs its stage now
    bool defer(Stage stage, const VVIntegrator& integrator, std::function<void()> stageByStage);
    void flush();                                              // run what is recorded, stage by stage
    std::function<void(const VVIntegrator&, uint32_t)> fusedMiddle, fusedFirst, fusedSecond;    // set by the step kernel of the context
    uint32_t pendingRandomIndex = 0;                           // slice of the random buffer applyLangevinForce reserved for the recorded step
    bool imagesFresh = false;                                  // the fused step has already mirrored the image particles
    long fusedSteps = 0, stagedCalls = 0;                      // diagnostics: fused launches through deferral / stages run one by one
private:
    std::vector<Stage> pattern;
    std::vector<std::function<void()> > pending;
    vvhip_params pendingParams;                                // parameters and box when the first pending stage was recorded
    double pendingBox[3] = {0, 0, 0};
    bool replaying = false;                                    // flush() in progress: syncParameters / stepSizeOf answer from the snapshot
    int classicHalf = 0;
    bool deferEnabled = true;
    HipContext& cu;
    vvhip_plan* plan;
    vvhip_params last;
    double lastBox[3] = {0, 0, 0};
    int ldRandoms;
    bool noConstraints;
    bool sitesInKernel;
    bool debug = false;      // last VVIntegrator::getDebugEnabled() handed to vvhip_set_trace
};

class HipVVStepCommon {          // code shared by the two step kernels
protected:
    explicit HipVVStepCommon(HipContext& cu) : cu(cu) {}
    void create(const System& system, const VVIntegrator& integrator, const DrudeForce* force);
    void advanceClock(const VVIntegrator& integrator);
    void announceStepSize(const VVIntegrator& integrator, bool classic);     // setNextStepSize / (0, dt) upload on a change (HOST:136-141, 307-319)
    uint32_t nextRandomIndex();
    double prevStepSize = -1.0;
    HipContext& cu;
    std::shared_ptr<HipVVPlan> plan;
};

class HipIntegrateMiddleStepKernel : public IntegrateMiddleStepKernel, public FusedVVStepKernel, private HipVVStepCommon {
public:
    HipIntegrateMiddleStepKernel(std::string name, const Platform& platform, HipContext& cu)
        : IntegrateMiddleStepKernel(name, platform), HipVVStepCommon(cu) {}
    void initialize(const System& system, const VVIntegrator& integrator, const DrudeForce* force);
    void firstIntegrate(ContextImpl& context, const VVIntegrator& integrator);
    void resetExtraForce(ContextImpl& context, const VVIntegrator& integrator);
    void secondIntegrate(ContextImpl& context, const VVIntegrator& integrator);
    double computeKineticEnergy(ContextImpl& context, const VVIntegrator& integrator);
    bool canFuse(ContextImpl& context, const VVIntegrator& integrator) const;
    void fusedMiddleStep(ContextImpl& context, const VVIntegrator& integrator);
    void fusedVVFirstHalf(ContextImpl& context, const VVIntegrator& integrator);
    void fusedVVSecondHalf(ContextImpl& context, const VVIntegrator& integrator);
private:
    void firstIntegrateNow(const VVIntegrator& integrator);
    void secondIntegrateNow(const VVIntegrator& integrator);
    void fusedMiddleWith(const VVIntegrator& integrator, uint32_t randomIndex);
};

class HipIntegrateVVStepKernel : public IntegrateVVStepKernel, public FusedVVStepKernel, private HipVVStepCommon {
public:
    HipIntegrateVVStepKernel(std::string name, const Platform& platform, HipContext& cu)
        : IntegrateVVStepKernel(name, platform), HipVVStepCommon(cu) {}
    void initialize(const System& system, const VVIntegrator& integrator, const DrudeForce* force);
    void firstIntegrate(ContextImpl& context, const VVIntegrator& integrator);
    void resetExtraForce(ContextImpl& context, const VVIntegrator& integrator);
    void secondIntegrate(ContextImpl& context, const VVIntegrator& integrator);
    double computeKineticEnergy(ContextImpl& context, const VVIntegrator& integrator);
    bool canFuse(ContextImpl& context, const VVIntegrator& integrator) const;
    void fusedMiddleStep(ContextImpl& context, const VVIntegrator& integrator);
    void fusedVVFirstHalf(ContextImpl& context, const VVIntegrator& integrator);
    void fusedVVSecondHalf(ContextImpl& context, const VVIntegrator& integrator);
private:
    void firstIntegrateNow(const VVIntegrator& integrator);
    void secondIntegrateNow(const VVIntegrator& integrator);
    void fusedFirstNow(const VVIntegrator& integrator);
    void fusedSecondWith(const VVIntegrator& integrator, uint32_t randomIndex);
};

class HipModifyDrudeNoseKernel : public ModifyDrudeNoseKernel {
public:
    HipModifyDrudeNoseKernel(std::string name, const Platform& platform, HipContext& cu) : ModifyDrudeNoseKernel(name, platform), cu(cu) {}
    void initialize(const System& system, const VVIntegrator& integrator, const DrudeForce* force);
    void scaleVelocity(ContextImpl& context, const VVIntegrator& integrator);
private:
    void scaleVelocityNow(const VVIntegrator& integrator);
    HipContext& cu;
    std::shared_ptr<HipVVPlan> plan;
};

class HipModifyDrudeLangevinKernel : public ModifyDrudeLangevinKernel {
public:
    HipModifyDrudeLangevinKernel(std::string name, const Platform& platform, HipContext& cu) : ModifyDrudeLangevinKernel(name, platform), cu(cu) {}
    void initialize(const System& system, const VVIntegrator& integrator, const DrudeForce* force, Kernel& vvKernel);
    void applyLangevinForce(ContextImpl& context, const VVIntegrator& integrator);
private:
    HipContext& cu;
    std::shared_ptr<HipVVPlan> plan;
};

class HipModifyImageChargeKernel : public ModifyImageChargeKernel {
public:
    HipModifyImageChargeKernel(std::string name, const Platform& platform, HipContext& cu) : ModifyImageChargeKernel(name, platform), cu(cu) {}
    void initialize(const System& system, const VVIntegrator& integrator);
    void updateImagePositions(ContextImpl& context, const VVIntegrator& integrator);
private:
    HipContext& cu;
    std::shared_ptr<HipVVPlan> plan;
};

class HipModifyElectricFieldKernel : public ModifyElectricFieldKernel {
public:
    HipModifyElectricFieldKernel(std::string name, const Platform& platform, HipContext& cu) : ModifyElectricFieldKernel(name, platform), cu(cu) {}
    void initialize(const System& system, const VVIntegrator& integrator, Kernel& vvKernel);
    void applyElectricForce(ContextImpl& context, const VVIntegrator& integrator);
private:
    HipContext& cu;
    std::shared_ptr<HipVVPlan> plan;
};

class HipModifyCosineAccelerateKernel : public ModifyCosineAccelerateKernel {
public:
    HipModifyCosineAccelerateKernel(std::string name, const Platform& platform, HipContext& cu) : ModifyCosineAccelerateKernel(name, platform), cu(cu) {}
    void initialize(const System& system, const VVIntegrator& integrator, Kernel& vvKernel);
    void applyCosineForce(ContextImpl& context, const VVIntegrator& integrator);
    void calcVelocityBias(ContextImpl& context, const VVIntegrator& integrator);
    void removeVelocityBias(ContextImpl& context, const VVIntegrator& integrator);
    void restoreVelocityBias(ContextImpl& context, const VVIntegrator& integrator);
    void calcViscosity(ContextImpl& context, const VVIntegrator& integrator, double& vMax, double& invVis);
private:
    HipContext& cu;
    std::shared_ptr<HipVVPlan> plan;
};

}  // namespace OpenMM
#endif
