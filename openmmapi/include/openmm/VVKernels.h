#ifndef VVHIP_OPENMM_VV_KERNELS_H_
#define VVHIP_OPENMM_VV_KERNELS_H_
// The seven platform-kernel interfaces VVIntegrator drives, with the registry names and virtual signatures of
// the reference (openmmapi/include/openmm/VVKernels.h:48-270), so a platform plugin written against either header
// is interchangeable.  FusedVVStepKernel is this build's one addition: an optional capability a step kernel may
// implement to run a whole step in two launches when no constraint solver has to run in between.
#include <string>
#include <vector>

#include "openmm/KernelImpl.h"
#include "openmm/Platform.h"
#include "openmm/System.h"

#include "openmm/FusedVVStepKernel.h"

namespace OpenMM {

class VVIntegrator;
class DrudeForce;
class ContextImpl;
class Kernel;

#define VV_KERNEL_HEAD(CLASS, NAME)                                                               \
    static std::string Name() { return NAME; }                                                    \
    CLASS(std::string name, const Platform& platform) : KernelImpl(name, platform) {}

// shared shape of the two step kernels (reference VVKernels.h:50-86 and :94-130)
class VVStepKernelBase : public KernelImpl {
public:
    VVStepKernelBase(std::string name, const Platform& platform) : KernelImpl(name, platform) {}
    virtual void initialize(const System& system, const VVIntegrator& integrator, const DrudeForce* force) = 0;
    virtual void firstIntegrate(ContextImpl& context, const VVIntegrator& integrator) = 0;
    virtual void resetExtraForce(ContextImpl& context, const VVIntegrator& integrator) = 0;
    virtual void secondIntegrate(ContextImpl& context, const VVIntegrator& integrator) = 0;
    virtual double computeKineticEnergy(ContextImpl& context, const VVIntegrator& integrator) = 0;
};

class IntegrateMiddleStepKernel : public VVStepKernelBase {
public:
    static std::string Name() { return "IntegrateMiddleStep"; }
    IntegrateMiddleStepKernel(std::string name, const Platform& platform) : VVStepKernelBase(name, platform) {}
};

class IntegrateVVStepKernel : public VVStepKernelBase {
public:
    static std::string Name() { return "IntegrateVVStep"; }
    IntegrateVVStepKernel(std::string name, const Platform& platform) : VVStepKernelBase(name, platform) {}
};

class ModifyDrudeNoseKernel : public KernelImpl {              // reference :138-157
public:
    VV_KERNEL_HEAD(ModifyDrudeNoseKernel, "ModifyDrudeNose")
    virtual void initialize(const System& system, const VVIntegrator& integrator, const DrudeForce* force) = 0;
    virtual void scaleVelocity(ContextImpl& context, const VVIntegrator& integrator) = 0;
};

class ModifyDrudeLangevinKernel : public KernelImpl {          // reference :165-184 (registry name is "ModifyLangevin")
public:
    VV_KERNEL_HEAD(ModifyDrudeLangevinKernel, "ModifyLangevin")
    virtual void initialize(const System& system, const VVIntegrator& integrator, const DrudeForce* force, Kernel& vvKernel) = 0;
    virtual void applyLangevinForce(ContextImpl& context, const VVIntegrator& integrator) = 0;
};

class ModifyImageChargeKernel : public KernelImpl {            // reference :192-211
public:
    VV_KERNEL_HEAD(ModifyImageChargeKernel, "ModifyImageCharge")
    virtual void initialize(const System& system, const VVIntegrator& integrator) = 0;
    virtual void updateImagePositions(ContextImpl& context, const VVIntegrator& integrator) = 0;
};

class ModifyElectricFieldKernel : public KernelImpl {          // reference :219-238
public:
    VV_KERNEL_HEAD(ModifyElectricFieldKernel, "ModifyElectricField")
    virtual void initialize(const System& system, const VVIntegrator& integrator, Kernel& vvKernel) = 0;
    virtual void applyElectricForce(ContextImpl& context, const VVIntegrator& integrator) = 0;
};

class ModifyCosineAccelerateKernel : public KernelImpl {       // reference :246-269
public:
    VV_KERNEL_HEAD(ModifyCosineAccelerateKernel, "ModifyCosineAccelerate")
    virtual void initialize(const System& system, const VVIntegrator& integrator, Kernel& vvKernel) = 0;
    virtual void applyCosineForce(ContextImpl& context, const VVIntegrator& integrator) = 0;
    virtual void calcVelocityBias(ContextImpl& context, const VVIntegrator& integrator) = 0;
    virtual void removeVelocityBias(ContextImpl& context, const VVIntegrator& integrator) = 0;
    virtual void restoreVelocityBias(ContextImpl& context, const VVIntegrator& integrator) = 0;
    virtual void calcViscosity(ContextImpl& context, const VVIntegrator& integrator, double& vMax, double& invVis) = 0;
};

// (the optional fused-step capability of a step kernel -- not in the reference -- lives in openmm/FusedVVStepKernel.h)
#undef VV_KERNEL_HEAD
}  // namespace OpenMM
#endif
