// compat/HipCompat.h -- stand-in for the pieces of OpenMM's HIP platform (HipPlatform, HipContext, HipArray,
// HipIntegrationUtilities, ContextSelector; OpenMM >= 8.2) that the kernel adapters in platforms/hip use.
//
// Every member below mirrors the shape in which the REFERENCE uses the CUDA twin of that service (CudaContext, CudaArray,
// CudaIntegrationUtilities; OpenMM's HIP platform keeps the same member names with hip types) -- cited per member as
// HOST:line = /root/reference/platforms/cuda/src/CudaVVKernels.cpp -- so that tests/test_cpp_plugin.py fails where an adapter
// assumes a different signature:
//   * getDevicePointer() is an LVALUE device-pointer handle (the reference takes its address: HOST:144-147, 179-184);
//   * getPeriodicBoxSize() returns a double4 BY VALUE (HOST:1129-1130), getInvPeriodicBoxSizePointer() a host pointer (HOST:1057);
//   * ContextSelector guards every initialize() (HOST:60, 246, 466, ...), setAsCurrent() every call (HOST:123, 133, 165, ...);
//   * integration.setNextStepSize() / getStepSize() carry the step size to OpenMM's own kernels (HOST:138-141, 309-319);
//   * constraint / virtual-site / reorder / kinetic-energy hooks are OpenMM's: here they only count calls.
// Nothing here is OpenMM source; INTEGRATION.md section 2 lists the same signatures.
#pragma once
#include "OpenMMCompat.h"

namespace OpenMM {

typedef void* hipDeviceptr_compat;     // OpenMM's HipArray hands out hipDeviceptr_t (= void*)

class HipArray {
public:
    HipArray() : ptr(nullptr), n(0), elem(0) {}
    ~HipArray() { if (ptr) (void) hipFree(ptr); }
    void initialize(size_t count, size_t elementSize) {
        n = count; elem = elementSize;
        const size_t bytes = count * elementSize;
        if (hipMalloc(&ptr, bytes > 0 ? bytes : 16) != hipSuccess) throw OpenMMException("hipMalloc failed");
        (void) hipMemset(ptr, 0, count * elementSize);
    }
    hipDeviceptr_compat& getDevicePointer() { return ptr; }        // lvalue, as CudaArray::getDevicePointer() (HOST:144)
    size_t getSize() const { return n; }
    size_t getElementSize() const { return elem; }
    void upload(const void* src) { if (hipMemcpy(ptr, src, n * elem, hipMemcpyHostToDevice) != hipSuccess) throw OpenMMException("upload failed"); }   // HOST:312
    void download(void* dst) const { if (hipMemcpy(dst, ptr, n * elem, hipMemcpyDeviceToHost) != hipSuccess) throw OpenMMException("download failed"); }
private:
    hipDeviceptr_compat ptr; size_t n, elem;
};

class HipContext;
class HipIntegrationUtilities {
public:
    explicit HipIntegrationUtilities(HipContext& cu) : cu(cu), randomPos(0), lastStepSize(0) {}
    HipArray& getPosDelta() { return posDelta; }                   // HOST:155
    HipArray& getRandom() { return random; }                       // HOST:869
    HipArray& getStepSize() { return stepSize; }                   // mixed2 (previous, next) step size, HOST:147, 312
    void initRandomNumberGenerator(unsigned int) { calls.initRandom++; }                  // HOST:63
    int prepareRandomNumbers(int numValues) {                      // HOST:863: hand out slices of the buffer, rewind when exhausted
        if (randomPos + numValues <= (int) random.getSize()) { int old = randomPos; randomPos += numValues; return old; }
        randomPos = numValues;
        return 0;
    }
    void setNextStepSize(double size);                             // HOST:139 (defined below: needs HipContext)
    double getLastStepSize() const { return lastStepSize; }
    void applyConstraints(double) { calls.applyConstraints++; }    // HOST:176, 351
    void applyVelocityConstraints(double) { calls.applyVelocityConstraints++; }   // HOST:151, 427
    void computeVirtualSites() { calls.computeVirtualSites++; }    // HOST:214, 374
    double computeKineticEnergy(double) { calls.computeKineticEnergy++; return 0.0; }   // HOST:234
    struct Calls { int initRandom = 0, setNextStepSize = 0, applyConstraints = 0, applyVelocityConstraints = 0, computeVirtualSites = 0, computeKineticEnergy = 0; } calls;
    HipArray posDelta, random, stepSize;
private:
    HipContext& cu;
    int randomPos;
    double lastStepSize;
};

class HipPlatform : public Platform {
public:
    struct PlatformData {
        std::vector<HipContext*> contexts;
        int initializeContextsCalls = 0;
        void initializeContexts(const System&) { initializeContextsCalls++; }           // HOST:61
    };
    const std::string& getName() const override { static const std::string n = "HIP"; return n; }
};

class HipContext {
public:
    HipContext(int numAtoms, bool useDouble, bool useMixed)
        : numAtoms(numAtoms), paddedNumAtoms((numAtoms + 31) / 32 * 32), useDouble(useDouble), useMixed(useMixed), integration(*this),
          time(0), stepCount(0), stream(nullptr), platformData(nullptr) {
        const size_t rs = useDouble ? 8 : 4, ms = (useDouble || useMixed) ? 8 : 4;
        velm.initialize(numAtoms, 4 * ms);
        posq.initialize(numAtoms, 4 * rs);
        posqCorrection.initialize(numAtoms, 4 * rs);
        force.initialize((size_t) 3 * paddedNumAtoms, 8);
        integration.posDelta.initialize(numAtoms, 4 * ms);
        integration.stepSize.initialize(1, 2 * ms);
        box = make_double4(1, 1, 1, 0);
        invBox = make_double4(1, 1, 1, 0);
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) throw OpenMMException("hipStreamCreate failed");
    }
    ~HipContext() { if (stream) (void) hipStreamDestroy(stream); }
    int getNumAtoms() const { return numAtoms; }                                   // HOST:65
    int getPaddedNumAtoms() const { return paddedNumAtoms; }                       // HOST:80
    bool getUseDoublePrecision() const { return useDouble; }                       // HOST:91
    bool getUseMixedPrecision() const { return useMixed; }                         // HOST:100
    HipArray& getVelm() { return velm; }                                           // HOST:144
    HipArray& getPosq() { return posq; }                                           // HOST:179
    HipArray& getPosqCorrection() { return posqCorrection; }                       // HOST:180
    HipArray& getForce() { return force; }                                         // HOST:145
    HipIntegrationUtilities& getIntegrationUtilities() { return integration; }     // HOST:134
    HipPlatform::PlatformData& getPlatformData() { if (!platformData) throw OpenMMException("no platform data"); return *platformData; }   // HOST:61
    void setPlatformData(HipPlatform::PlatformData* pd) { platformData = pd; }
    hipStream_t getCurrentStream() const { return stream; }
    void setAsCurrent() { setAsCurrentCalls++; }                                   // HOST:123, 133, 165, ...
    void pushAsCurrent() { selectorDepth++; selectorUses++; }                      // what ContextSelector does
    void popAsCurrent() { selectorDepth--; }
    void reorderAtoms() { reorderCalls++; }                                        // HOST:216, 381
    double getTime() const { return time; }
    void setTime(double t) { time = t; }                                           // HOST:219
    long long getStepCount() const { return stepCount; }
    void setStepCount(long long s) { stepCount = s; }                              // HOST:220
    void setPeriodicBoxSize(double x, double y, double z) { box = make_double4(x, y, z, 0); invBox = make_double4(1 / x, 1 / y, 1 / z, 0); }
    double4 getPeriodicBoxSize() const { return box; }                             // by value: HOST:1129
    void* getInvPeriodicBoxSizePointer() { return &invBox; }                       // HOST:1057 (kernel-argument pointer)
    int setAsCurrentCalls = 0, selectorUses = 0, selectorDepth = 0, reorderCalls = 0;
private:
    int numAtoms, paddedNumAtoms;
    bool useDouble, useMixed;
    HipArray velm, posq, posqCorrection, force;
    HipIntegrationUtilities integration;
    double time; long long stepCount; double4 box, invBox;
    hipStream_t stream;
    HipPlatform::PlatformData* platformData;
};

inline void HipIntegrationUtilities::setNextStepSize(double size) {                // OpenMM uploads (lastStepSize, size) to the stepSize array
    calls.setNextStepSize++;
    if (cu.getUseDoublePrecision() || cu.getUseMixedPrecision()) { double ss[2] = {lastStepSize, size}; stepSize.upload(ss); }
    else { float ss[2] = {(float) lastStepSize, (float) size}; stepSize.upload(ss); }
    lastStepSize = size;
}

class ContextSelector {                  // RAII guard of the reference's initialize() methods (HOST:60, 246, 466, 765, 882, 944, 1002)
public:
    explicit ContextSelector(HipContext& cu) : cu(cu) { cu.pushAsCurrent(); }
    ~ContextSelector() { cu.popAsCurrent(); }
private:
    HipContext& cu;
};

}  // namespace OpenMM
