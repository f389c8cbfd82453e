// compat/HipCompat.h -- stand-in for the pieces of OpenMM's HIP platform (HipPlatform, HipContext, HipArray,
// HipIntegrationUtilities; OpenMM >= 8.2) that the kernel adapters in platforms/hip use: device arrays in OpenMM's
// layouts, step size / time bookkeeping, a random buffer, and no-op constraint / virtual-site / reorder hooks
// (those solvers are OpenMM's: SURVEY.md §8f-1).
#pragma once
#include "OpenMMCompat.h"

namespace OpenMM {

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
    void* getDevicePointer() const { return ptr; }
    size_t getSize() const { return n; }
    size_t getElementSize() const { return elem; }
    void upload(const void* src) { if (hipMemcpy(ptr, src, n * elem, hipMemcpyHostToDevice) != hipSuccess) throw OpenMMException("upload failed"); }
    void download(void* dst) const { if (hipMemcpy(dst, ptr, n * elem, hipMemcpyDeviceToHost) != hipSuccess) throw OpenMMException("download failed"); }
private:
    void* ptr; size_t n, elem;
};

class HipContext;
class HipIntegrationUtilities {
public:
    explicit HipIntegrationUtilities(HipContext&) : randomPos(0) {}
    HipArray& getPosDelta() { return posDelta; }
    HipArray& getRandom() { return random; }
    void initRandomNumberGenerator(unsigned int) {}
    int prepareRandomNumbers(int numValues) {                       // hand out slices of the buffer, rewind when exhausted
        if (randomPos + numValues <= (int) random.getSize()) { int old = randomPos; randomPos += numValues; return old; }
        randomPos = numValues;
        return 0;
    }
    void setNextStepSize(double) {}
    void applyConstraints(double) {}
    void applyVelocityConstraints(double) {}
    void computeVirtualSites() {}
    double computeKineticEnergy(double) { return 0.0; }
    HipArray posDelta, random;
private:
    int randomPos;
};

class HipContext {
public:
    HipContext(int numAtoms, bool useDouble, bool useMixed)
        : numAtoms(numAtoms), paddedNumAtoms((numAtoms + 31) / 32 * 32), useDouble(useDouble), useMixed(useMixed), integration(*this),
          time(0), stepCount(0), stream(nullptr) {
        const size_t rs = useDouble ? 8 : 4, ms = (useDouble || useMixed) ? 8 : 4;
        velm.initialize(numAtoms, 4 * ms);
        posq.initialize(numAtoms, 4 * rs);
        posqCorrection.initialize(numAtoms, 4 * rs);
        force.initialize((size_t) 3 * paddedNumAtoms, 8);
        integration.posDelta.initialize(numAtoms, 4 * ms);
        box[0] = box[1] = box[2] = 1.0;
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) throw OpenMMException("hipStreamCreate failed");
    }
    ~HipContext() { if (stream) (void) hipStreamDestroy(stream); }
    int getNumAtoms() const { return numAtoms; }
    int getPaddedNumAtoms() const { return paddedNumAtoms; }
    bool getUseDoublePrecision() const { return useDouble; }
    bool getUseMixedPrecision() const { return useMixed; }
    HipArray& getVelm() { return velm; }
    HipArray& getPosq() { return posq; }
    HipArray& getPosqCorrection() { return posqCorrection; }
    HipArray& getForce() { return force; }
    HipIntegrationUtilities& getIntegrationUtilities() { return integration; }
    hipStream_t getCurrentStream() const { return stream; }
    void setAsCurrent() {}
    void reorderAtoms() {}
    double getTime() const { return time; }
    void setTime(double t) { time = t; }
    long long getStepCount() const { return stepCount; }
    void setStepCount(long long s) { stepCount = s; }
    void setPeriodicBoxSize(double x, double y, double z) { box[0] = x; box[1] = y; box[2] = z; }
    const double* getPeriodicBoxSize() const { return box; }
private:
    int numAtoms, paddedNumAtoms;
    bool useDouble, useMixed;
    HipArray velm, posq, posqCorrection, force;
    HipIntegrationUtilities integration;
    double time; long long stepCount; double box[3];
    hipStream_t stream;
};

class HipPlatform : public Platform {
public:
    struct PlatformData { std::vector<HipContext*> contexts; };
    const std::string& getName() const override { static const std::string n = "HIP"; return n; }
};

}  // namespace OpenMM
