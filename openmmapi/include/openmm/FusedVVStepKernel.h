// openmm/FusedVVStepKernel.h -- optional capability of a step kernel, NOT part of the reference's interface (its VVKernels.h has no such
// class): a kernel implementation may additionally derive from it, and a VVIntegrator that knows about it (openmmapi/src/VVIntegrator.cpp of
// this repository) asks for it with dynamic_cast and takes the fused path.  Kept in its own header so that the HIP plugin also builds against
// the REFERENCE's openmmapi headers unchanged -- the reference's VVIntegrator then simply drives the seven kernels through their own virtuals
// (oracle/Makefile target `refplugin`, tests/test_cpp_plugin.py).
#ifndef OPENMM_FUSEDVVSTEPKERNEL_H_
#define OPENMM_FUSEDVVSTEPKERNEL_H_

namespace OpenMM {
class ContextImpl;
class VVIntegrator;

// Optional capability of a step kernel (not in the reference): everything between two force evaluations in as few
// launches as the data dependencies allow.  VVIntegrator uses it only when canFuse() says no solver must interleave.
class FusedVVStepKernel {
public:
    virtual ~FusedVVStepKernel() {}
    virtual bool canFuse(ContextImpl& context, const VVIntegrator& integrator) const = 0;
    virtual void fusedMiddleStep(ContextImpl& context, const VVIntegrator& integrator) = 0;          // VVIntegrator.cpp:237-268 of the reference
    virtual void fusedVVFirstHalf(ContextImpl& context, const VVIntegrator& integrator) = 0;         // :294-310
    virtual void fusedVVSecondHalf(ContextImpl& context, const VVIntegrator& integrator) = 0;        // :315-336
};

}  // namespace OpenMM
#endif
