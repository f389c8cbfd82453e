// oracle/refhost/CudaContext.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Stand-in for the slice of OpenMM's CUDA platform (CudaContext, CudaArray, CudaIntegrationUtilities, CudaPlatform::PlatformData,
// ContextSelector, the CUDA driver handle types) that the REFERENCE's host code uses --
//     /root/reference/platforms/cuda/src/CudaVVKernels.cpp, CudaVVKernelFactory.cpp, include/CudaVVKernels.h
// -- so that those files compile IN PLACE (oracle/Makefile target `refhost`) and run on the CPU: "device" arrays are host memory,
// cu.createModule / getKernel / executeKernel resolve to the reference's own kernels, compiled for the CPU from K/*.cu by the same
// Makefile (ref_prelude.h, one block of one thread).  Together with the reference's VVIntegrator.cpp this executes the reference's
// WHOLE step -- its sequencing, its host constants (DOF, thermostat masses, random and field factors), its launch arguments and
// its kernel arithmetic -- with only OpenMM's services replaced (arrays, launches, the random buffer, no constraints / virtual
// sites / reordering).  Needs stand-ins for OpenMM headers and for the generated CudaVVKernelSources, so by this tier's rule it is
// a cross-check, not "the reference as shipped" (DESIGN.md section 2).  Nothing here is OpenMM or reference source.
#pragma once
#include <cmath>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "openmm/OpenMMException.h"
#include "openmm/Platform.h"
#include "openmm/System.h"

// ---- CUDA vector types (int2, float4, double4, make_*): the HIP host headers compat/OpenMMCompat.h already includes define them with
//      the same layout as the kernels' own (oracle/ref_prelude.h)

// ---- driver handles
typedef unsigned long long CUdeviceptr;                  // holds a host address here
struct VVRefHostModule { std::string source; std::map<std::string, std::string> defines; };
struct VVRefHostFunction { VVRefHostModule* module; std::string name; };
typedef VVRefHostModule* CUmodule;
typedef VVRefHostFunction* CUfunction;

// the launcher (oracle/ref_host_kernels.cpp, compiled with the kernels' prelude): sets the size macros from the module's defines and
// calls the reference kernel `name` of translation unit `source` with the unpacked arguments
extern "C" void vvrh_launch(const char* source, const char* name, const std::map<std::string, std::string>* defines, void** args);

namespace OpenMM {

class CudaContext;

class CudaArray {
public:
    template <class T> static CudaArray* create(CudaContext& cu, size_t size, const std::string& name) {
        CudaArray* a = new CudaArray();
        a->initialize(size, sizeof(T), name);
        return a;
    }
    CudaArray() : ptr(0), n(0), elem(0) {}
    ~CudaArray() { delete[] (char*) (size_t) ptr; }
    void initialize(size_t size, size_t elementSize, const std::string& nm) {
        n = size; elem = elementSize; name = nm;
        char* p = new char[size * elementSize + 64];
        std::memset(p, 0, size * elementSize + 64);      // (cudaMalloc does not zero; the reference relies on zeros for its KE buffer)
        ptr = (CUdeviceptr) (size_t) p;
    }
    CUdeviceptr& getDevicePointer() { return ptr; }
    size_t getSize() const { return n; }
    int getElementSize() const { return (int) elem; }
    template <class T> void upload(const std::vector<T>& v) {
        if (v.size() * sizeof(T) != n * elem) throw OpenMMException("CudaArray::upload: size mismatch for " + name);
        std::memcpy((void*) (size_t) ptr, v.data(), n * elem);
    }
    void upload(const void* src) { std::memcpy((void*) (size_t) ptr, src, n * elem); }
    template <class T> void download(std::vector<T>& v) const {
        if (v.size() * sizeof(T) != n * elem) v.resize(n * elem / sizeof(T));
        std::memcpy(v.data(), (const void*) (size_t) ptr, n * elem);
    }
    void download(void* dst) const { std::memcpy(dst, (const void*) (size_t) ptr, n * elem); }
private:
    CUdeviceptr ptr; size_t n, elem; std::string name;
};

class CudaIntegrationUtilities {
public:
    explicit CudaIntegrationUtilities(CudaContext& cu) : cu(cu), randomPos(0), lastStepSize(0) {}
    CudaArray& getPosDelta() { return posDelta; }
    CudaArray& getStepSize() { return stepSize; }
    CudaArray& getRandom() { return random; }
    void setNextStepSize(double size);
    void initRandomNumberGenerator(unsigned int) {}
    int prepareRandomNumbers(int numValues) {            // slices of the injected buffer, rewound when exhausted (as the oracle does)
        if (randomPos + numValues <= (int) random.getSize()) { int old = randomPos; randomPos += numValues; return old; }
        randomPos = numValues;
        return 0;
    }
    void applyConstraints(double) {}                     // no constraints in the systems driven through this build
    void applyVelocityConstraints(double) {}
    void computeVirtualSites() {}
    double computeKineticEnergy(double) { return 0.0; }
    CudaArray posDelta, stepSize, random;
private:
    CudaContext& cu;
    int randomPos;
    double lastStepSize;
};

class CudaPlatform : public Platform {
public:
    struct PlatformData {
        std::vector<CudaContext*> contexts;
        void initializeContexts(const System&) {}
    };
    const std::string& getName() const override { static const std::string n = "CUDA"; return n; }
};

class CudaContext {
public:
    static const int ThreadBlockSize = 64;
    CudaContext(int numAtoms, bool useDouble, bool useMixed, CudaPlatform::PlatformData& pd)
        : numAtoms(numAtoms), paddedNumAtoms((numAtoms + 31) / 32 * 32), useDouble(useDouble), useMixed(useMixed), integration(*this),
          time(0), stepCount(0), platformData(pd) {
        const size_t rs = useDouble ? 8 : 4, ms = (useDouble || useMixed) ? 8 : 4;
        velm.initialize(numAtoms, 4 * ms, "velm");
        posq.initialize(numAtoms, 4 * rs, "posq");
        posqCorrection.initialize(numAtoms, 4 * rs, "posqCorrection");
        force.initialize((size_t) 3 * paddedNumAtoms, 8, "force");
        integration.posDelta.initialize(numAtoms, 4 * ms, "posDelta");
        integration.stepSize.initialize(1, 2 * ms, "stepSize");
        setPeriodicBoxSize(1, 1, 1);
    }
    ~CudaContext() { for (auto* m : modules) delete m; for (auto* f : functions) delete f; }
    int getNumAtoms() const { return numAtoms; }
    int getPaddedNumAtoms() const { return paddedNumAtoms; }
    bool getUseDoublePrecision() const { return useDouble; }
    bool getUseMixedPrecision() const { return useMixed; }
    int getNumThreadBlocks() const { return 1; }
    CudaArray& getVelm() { return velm; }
    CudaArray& getPosq() { return posq; }
    CudaArray& getPosqCorrection() { return posqCorrection; }
    CudaArray& getForce() { return force; }
    CudaIntegrationUtilities& getIntegrationUtilities() { return integration; }
    CudaPlatform::PlatformData& getPlatformData() { return platformData; }
    void setAsCurrent() {}
    void pushAsCurrent() {}
    void popAsCurrent() {}
    void reorderAtoms() {}
    double getTime() const { return time; }
    void setTime(double t) { time = t; }
    long long getStepCount() const { return stepCount; }
    void setStepCount(long long s) { stepCount = s; }
    void setPeriodicBoxSize(double x, double y, double z) {
        box = make_double4(x, y, z, 0);
        invBoxD = make_double4(1 / x, 1 / y, 1 / z, 0);
        invBoxF = make_float4((float) (1 / x), (float) (1 / y), (float) (1 / z), 0);
    }
    double4 getPeriodicBoxSize() const { return box; }
    void* getInvPeriodicBoxSizePointer() { return useDouble ? (void*) &invBoxD : (void*) &invBoxF; }     // real4, passed by value to the kernels
    std::string intToString(int v) const { std::ostringstream s; s << v; return s.str(); }
    std::string doubleToString(double v) const { std::ostringstream s; s.precision(17); s << v; return s.str(); }
    CUmodule createModule(const std::string& source, const std::map<std::string, std::string>& defines, const char* = "") {
        modules.push_back(new VVRefHostModule{source, defines});
        return modules.back();
    }
    CUfunction getKernel(CUmodule module, const std::string& name) {
        functions.push_back(new VVRefHostFunction{module, name});
        return functions.back();
    }
    void executeKernel(CUfunction kernel, void** arguments, int /*threads*/, int /*blockSize*/ = -1, unsigned int /*sharedSize*/ = 0) {
        launches.push_back(kernel->name);
        vvrh_launch(kernel->module->source.c_str(), kernel->name.c_str(), &kernel->module->defines, arguments);
    }
    std::vector<std::string> launches;                   // names of the kernels launched so far (the reference's launch order)
private:
    int numAtoms, paddedNumAtoms;
    bool useDouble, useMixed;
    CudaArray velm, posq, posqCorrection, force;
    CudaIntegrationUtilities integration;
    double time; long long stepCount;
    double4 box, invBoxD; float4 invBoxF;
    CudaPlatform::PlatformData& platformData;
    std::vector<VVRefHostModule*> modules;
    std::vector<VVRefHostFunction*> functions;
};

inline void CudaIntegrationUtilities::setNextStepSize(double size) {     // (previous, next) step size, as OpenMM's integration utilities keep it
    if (cu.getUseDoublePrecision() || cu.getUseMixedPrecision()) { double ss[2] = {lastStepSize, size}; stepSize.upload(ss); }
    else { float ss[2] = {(float) lastStepSize, (float) size}; stepSize.upload(ss); }
    lastStepSize = size;
}

class ContextSelector {
public:
    explicit ContextSelector(CudaContext& cu) : cu(cu) { cu.pushAsCurrent(); }
    ~ContextSelector() { cu.popAsCurrent(); }
private:
    CudaContext& cu;
};

}  // namespace OpenMM
