// compat/OpenMMCompat.h -- the slice of OpenMM's public + plugin API that openmmapi/ and platforms/hip/ touch,
// so that both build and run where OpenMM itself is not installed (this image).
// With a real OpenMM >= 8.2 (the first release with an in-tree HIP platform) put $OPENMM_DIR/include in front of
// compat/ on the include path and these stand-ins are never seen.  Written from the call sites in the reference
// (openmmapi/src/VVIntegrator.cpp, platforms/cuda/src/CudaVVKernel*.cpp); nothing here is OpenMM source.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <utility>
#include <vector>

namespace OpenMM {

class OpenMMException : public std::runtime_error {
public:
    explicit OpenMMException(const std::string& m) : std::runtime_error(m) {}
};

struct Vec3 {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }      // (OpenMM's Vec3 is indexed, it has no named members)
};
class State { public: enum DataType { Positions = 1, Velocities = 2, Forces = 4, Energy = 8, Parameters = 16 }; };

class Force { public: virtual ~Force() {} };
class CMMotionRemover : public Force {};
class DrudeForce : public Force {
public:
    int addParticle(int p, int p1, int p2, int p3, int p4, double charge, double polarizability, double aniso12, double aniso34) {
        rows.push_back({p, p1, p2, p3, p4, charge, polarizability, aniso12, aniso34});
        return (int) rows.size() - 1;
    }
    int getNumParticles() const { return (int) rows.size(); }
    void getParticleParameters(int i, int& p, int& p1, int& p2, int& p3, int& p4, double& charge, double& polarizability,
                               double& aniso12, double& aniso34) const {
        const Row& r = rows.at(i);
        p = r.p; p1 = r.p1; p2 = r.p2; p3 = r.p3; p4 = r.p4; charge = r.q; polarizability = r.pol; aniso12 = r.a12; aniso34 = r.a34;
    }
private:
    struct Row { int p, p1, p2, p3, p4; double q, pol, a12, a34; };
    std::vector<Row> rows;
};

// OpenMM's virtual-site classes (openmm/VirtualSite.h), as far as the adapters read them
class VirtualSite {
public:
    virtual ~VirtualSite() {}
    int getNumParticles() const { return (int) particles.size(); }
    int getParticle(int i) const { return particles.at(i); }
protected:
    void setParticles(const std::vector<int>& p) { particles = p; }
private:
    std::vector<int> particles;
};
class TwoParticleAverageSite : public VirtualSite {
public:
    TwoParticleAverageSite(int p1, int p2, double w1, double w2) : w{w1, w2} { setParticles({p1, p2}); }
    double getWeight(int i) const { return w[i]; }
private:
    double w[2];
};
class ThreeParticleAverageSite : public VirtualSite {
public:
    ThreeParticleAverageSite(int p1, int p2, int p3, double w1, double w2, double w3) : w{w1, w2, w3} { setParticles({p1, p2, p3}); }
    double getWeight(int i) const { return w[i]; }
private:
    double w[3];
};
class OutOfPlaneSite : public VirtualSite {
public:
    OutOfPlaneSite(int p1, int p2, int p3, double w12, double w13, double wc) : w12(w12), w13(w13), wc(wc) { setParticles({p1, p2, p3}); }
    double getWeight12() const { return w12; }
    double getWeight13() const { return w13; }
    double getWeightCross() const { return wc; }
private:
    double w12, w13, wc;
};
class LocalCoordinatesSite : public VirtualSite {
public:
    LocalCoordinatesSite(int p1, int p2, int p3, const Vec3& ow, const Vec3& xw, const Vec3& yw, const Vec3& local)
        : ow{ow.x, ow.y, ow.z}, xw{xw.x, xw.y, xw.z}, yw{yw.x, yw.y, yw.z}, local(local) { setParticles({p1, p2, p3}); }
    void getOriginWeights(std::vector<double>& w) const { w = ow; }
    void getXWeights(std::vector<double>& w) const { w = xw; }
    void getYWeights(std::vector<double>& w) const { w = yw; }
    const Vec3& getLocalPosition() const { return local; }
private:
    std::vector<double> ow, xw, yw;
    Vec3 local;
};

class System {
public:
    ~System() { for (Force* f : forces) delete f; for (auto& s : sites) delete s.second; }
    void setVirtualSite(int index, VirtualSite* site) { delete sites[index]; sites[index] = site; }      // takes ownership, as OpenMM does
    bool isVirtualSite(int index) const { return sites.count(index) != 0; }
    const VirtualSite& getVirtualSite(int index) const { return *sites.at(index); }
    int addParticle(double mass) { masses.push_back(mass); return (int) masses.size() - 1; }
    int getNumParticles() const { return (int) masses.size(); }
    double getParticleMass(int i) const { return masses.at(i); }
    int addConstraint(int a, int b, double d) { cons.push_back({a, b, d}); return (int) cons.size() - 1; }
    int getNumConstraints() const { return (int) cons.size(); }
    void getConstraintParameters(int i, int& a, int& b, double& d) const { a = cons.at(i).a; b = cons.at(i).b; d = cons.at(i).d; }
    int addForce(Force* f) { forces.push_back(f); return (int) forces.size() - 1; }   // takes ownership, as OpenMM does
    int getNumForces() const { return (int) forces.size(); }
    const Force& getForce(int i) const { return *forces.at(i); }
private:
    struct Con { int a, b; double d; };
    std::vector<double> masses;
    std::vector<Con> cons;
    std::vector<Force*> forces;
    std::map<int, VirtualSite*> sites;
};

class Platform;
class ContextImpl;
class Context;

class KernelImpl {
public:
    KernelImpl(std::string name, const Platform& platform) : name(std::move(name)), platform(&platform) {}
    virtual ~KernelImpl() {}
    const std::string& getName() const { return name; }
    const Platform& getPlatform() const { return *platform; }
private:
    friend class Kernel;
    std::string name;
    const Platform* platform;
    int refs = 0;
};

class Kernel {   // ref-counted handle, owns the impl (SURVEY.md §8b "ownership")
public:
    Kernel() : impl(nullptr) {}
    explicit Kernel(KernelImpl* i) : impl(i) { if (impl) impl->refs++; }
    Kernel(const Kernel& o) : impl(o.impl) { if (impl) impl->refs++; }
    Kernel& operator=(const Kernel& o) {
        if (o.impl) o.impl->refs++;
        release();
        impl = o.impl;
        return *this;
    }
    ~Kernel() { release(); }
    template <class T> T& getAs() {
        T* t = dynamic_cast<T*>(impl);
        if (!t) throw OpenMMException("Kernel::getAs: wrong kernel type");
        return *t;
    }
    KernelImpl& getImpl() { return *impl; }
private:
    void release() { if (impl && --impl->refs == 0) delete impl; impl = nullptr; }
    KernelImpl* impl;
};

class KernelFactory {
public:
    virtual ~KernelFactory() {}
    virtual KernelImpl* createKernelImpl(std::string name, const Platform& platform, ContextImpl& context) const = 0;
};

class Platform {
public:
    virtual ~Platform() {}
    virtual const std::string& getName() const = 0;
    void registerKernelFactory(const std::string& name, KernelFactory* f) { factories[name] = f; }
    bool hasKernelFactory(const std::string& name) const { return factories.count(name) != 0; }
    Kernel createKernel(const std::string& name, ContextImpl& context) const {
        auto it = factories.find(name);
        if (it == factories.end()) throw OpenMMException("Called createKernel() on a Platform which does not support the requested kernel");
        return Kernel(it->second->createKernelImpl(name, *this, context));
    }
    static void registerPlatform(Platform* p) { registry().push_back(p); }
    static Platform& getPlatformByName(const std::string& name) {
        for (Platform* p : registry()) if (p->getName() == name) return *p;
        throw OpenMMException("There is no registered Platform called \"" + name + "\"");
    }
private:
    static std::vector<Platform*>& registry() { static std::vector<Platform*> r; return r; }
    std::map<std::string, KernelFactory*> factories;
};

class Integrator;
class Context {
public:
    Context(const System& system, Integrator& integrator, Platform& platform);   // defined in CompatContext.h (needs ContextImpl)
    ~Context();
    ContextImpl& getImpl() { return *impl; }
    Integrator& getIntegrator() { return *integrator; }
    void initialize();            // what OpenMM's ContextImpl constructor does last: integrator.initialize(impl)
private:
    ContextImpl* impl;
    Integrator* integrator;
};

class ContextImpl {
public:
    typedef void (*ForceCallback)(ContextImpl&, void* user);
    ContextImpl(Context& owner, const System& system, Integrator& integrator, Platform& platform)
        : owner(owner), system(system), integrator(integrator), platform(platform) {}
    Context& getOwner() { return owner; }
    const System& getSystem() const { return system; }
    Integrator& getIntegrator() { return integrator; }
    Platform& getPlatform() { return platform; }
    void* getPlatformData() { return platformData; }
    void setPlatformData(void* d) { platformData = d; }
    // bond-connected components; the stand-in lets the host say it directly
    const std::vector<std::vector<int> >& getMolecules() const { return molecules; }
    void setMolecules(std::vector<std::vector<int> > m) { molecules = std::move(m); }
    bool updateContextState() { return false; }                       // barostat / CMMotionRemover: OpenMM's
    double calcForcesAndEnergy(bool, bool) { if (forceCallback) forceCallback(*this, forceUser); return 0.0; }   // force kernels: OpenMM's
    void setForceCallback(ForceCallback cb, void* user) { forceCallback = cb; forceUser = user; }
private:
    Context& owner;
    const System& system;
    Integrator& integrator;
    Platform& platform;
    void* platformData = nullptr;
    std::vector<std::vector<int> > molecules;
    ForceCallback forceCallback = nullptr;
    void* forceUser = nullptr;
};

class Integrator {
public:
    Integrator() : owner(nullptr), context(nullptr), stepSize(0), constraintTol(1e-5) {}
    virtual ~Integrator() {}
    double getStepSize() const { return stepSize; }
    void setStepSize(double s) { stepSize = s; }
    double getConstraintTolerance() const { return constraintTol; }
    void setConstraintTolerance(double t) { constraintTol = t; }
    virtual void step(int steps) = 0;
protected:
    friend class Context;
    Context* owner;
    ContextImpl* context;
    virtual void initialize(ContextImpl& context) = 0;
    virtual void cleanup() {}
    virtual std::vector<std::string> getKernelNames() = 0;
    virtual void stateChanged(State::DataType) {}
    virtual double computeKineticEnergy() = 0;
    virtual bool kineticEnergyRequiresForce() const { return true; }
private:
    double stepSize, constraintTol;
};

inline Context::Context(const System& system, Integrator& integ, Platform& platform)
    : impl(new ContextImpl(*this, system, integ, platform)), integrator(&integ) {}
inline Context::~Context() { integrator->cleanup(); delete impl; }
inline void Context::initialize() { integrator->initialize(*impl); }

}  // namespace OpenMM
