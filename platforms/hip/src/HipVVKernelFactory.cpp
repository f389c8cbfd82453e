// Plugin registration for the HIP platform; counterpart of platforms/cuda/src/CudaVVKernelFactory.cpp.
#include "HipVVKernelFactory.h"

#include "HipVVKernels.h"
#include "openmm/OpenMMException.h"
#include "openmm/internal/ContextImpl.h"

using namespace OpenMM;

extern "C" void registerPlatforms() {}

extern "C" void registerKernelFactories() {
    try {
        Platform& platform = Platform::getPlatformByName("HIP");
        HipVVKernelFactory* factory = new HipVVKernelFactory();   // lives as long as the platform, like the reference's
        for (const std::string& name : {IntegrateMiddleStepKernel::Name(), IntegrateVVStepKernel::Name(), ModifyDrudeNoseKernel::Name(),
                                        ModifyDrudeLangevinKernel::Name(), ModifyImageChargeKernel::Name(),
                                        ModifyElectricFieldKernel::Name(), ModifyCosineAccelerateKernel::Name()})
            platform.registerKernelFactory(name, factory);
    } catch (const std::exception&) {
        // no HIP platform in this OpenMM: nothing to attach to (the reference swallows the same case, :52-54)
    }
}

extern "C" void registerHipVVKernelFactories() {
    try {
        Platform::getPlatformByName("HIP");
    } catch (...) {
        Platform::registerPlatform(new HipPlatform());
    }
    registerKernelFactories();
}

KernelImpl* HipVVKernelFactory::createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const {
    HipContext& cu = *static_cast<HipPlatform::PlatformData*>(context.getPlatformData())->contexts[0];
    if (name == IntegrateMiddleStepKernel::Name()) return new HipIntegrateMiddleStepKernel(name, platform, cu);
    if (name == IntegrateVVStepKernel::Name()) return new HipIntegrateVVStepKernel(name, platform, cu);
    if (name == ModifyDrudeNoseKernel::Name()) return new HipModifyDrudeNoseKernel(name, platform, cu);
    if (name == ModifyDrudeLangevinKernel::Name()) return new HipModifyDrudeLangevinKernel(name, platform, cu);
    if (name == ModifyImageChargeKernel::Name()) return new HipModifyImageChargeKernel(name, platform, cu);
    if (name == ModifyElectricFieldKernel::Name()) return new HipModifyElectricFieldKernel(name, platform, cu);
    if (name == ModifyCosineAccelerateKernel::Name()) return new HipModifyCosineAccelerateKernel(name, platform, cu);
    throw OpenMMException((std::string("Tried to create kernel with illegal kernel name '") + name + "'").c_str());
}
