#ifndef VVHIP_KERNEL_FACTORY_H_
#define VVHIP_KERNEL_FACTORY_H_
// KernelFactory of the HIP backend; counterpart of platforms/cuda/include/CudaVVKernelFactory.h in the reference.
#include "openmm/KernelFactory.h"

namespace OpenMM {
class HipVVKernelFactory : public KernelFactory {
public:
    KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const;
};
}  // namespace OpenMM

// plugin entry points looked up by OpenMM's plugin loader (reference: CudaVVKernelFactory.cpp:37,40,57)
extern "C" void registerPlatforms();
extern "C" void registerKernelFactories();
extern "C" void registerHipVVKernelFactories();
#endif
