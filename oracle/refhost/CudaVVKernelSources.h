// oracle/refhost/CudaVVKernelSources.h -- TEST INFRASTRUCTURE.  In the reference this header is GENERATED at build time: CMake embeds
// the text of platforms/cuda/src/kernels/*.cu as strings (CudaVVKernelSources.h.in / EncodeCUDAFiles.cmake) for NVRTC.  Here the
// kernels are compiled ahead of time for the CPU from the same files, so each "source" is just the name of its translation unit; the
// host concatenates vectorOps + <unit> (CudaVVKernels.cpp:102, 281, 649, 817, 896, 963, 1016) and the launcher looks the unit up in
// that string.
#pragma once
#include <string>
namespace OpenMM {
class CudaVVKernelSources {
public:
    static const std::string vectorOps, middle, velocityVerlet, drudeNoseHoover, drudeLangevin, imageCharge, electricField, cosineAccelerate;
};
}  // namespace OpenMM
