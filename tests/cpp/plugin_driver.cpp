// tests/cpp/plugin_driver.cpp -- test code.  Drives the C++ layers the way OpenMM does:
//   registerHipVVKernelFactories() -> Platform "HIP" -> Context -> VVIntegrator::initialize -> createKernel(...) ->
//   Hip*Kernel adapters -> libvvhip C ABI -> HIP kernels.
//   vv_plugin_driver registry                 (no GPU needed) checks registration, names and error behaviour
//   vv_plugin_driver chain                    (no GPU needed) prints VVIntegrator::propagateNHChain on fixed inputs
//   vv_plugin_driver run OUT middle cons cos N   (GPU) runs N steps on a small Drude system and dumps system + result
//   vv_plugin_driver fuzz OUT middle cons cos NOPS SEED HAND   (GPU) a seeded random sequence of steps, parameter / box changes and queries
//                                                               (HAND = 1: stages driven by hand, changes and queries BETWEEN the stages)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "HipVVKernelFactory.h"
#include "HipVVKernels.h"
#include "openmm/CMMotionRemover.h"
#include "openmm/Context.h"
#include "openmm/DrudeForce.h"
#include "openmm/VVIntegrator.h"
#include "openmm/VVKernels.h"

using namespace OpenMM;

static unsigned long long lcg_state = 88172645463325252ull;
static double uniform() { lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull; return (double) (lcg_state >> 11) / 9007199254740992.0; }
static double gauss() { double s = 0; for (int i = 0; i < 12; i++) s += uniform(); return s - 6.0; }

struct ProbeIntegrator : VVIntegrator {       // exposes the protected kernel-name list and the kinetic-energy query
    using VVIntegrator::VVIntegrator;
    std::vector<std::string> names() { return getKernelNames(); }
    double kineticEnergy() { return computeKineticEnergy(); }
};
// second random stream: the operations of the fuzz run (the first one lays out the system, the same for every seed)
static unsigned long long op_state = 1;
static unsigned op_next(unsigned n) { op_state = op_state * 6364136223846793005ull + 1442695040888963407ull; return (unsigned) ((op_state >> 33) % n); }

static int registry() {
    registerPlatforms();
    registerHipVVKernelFactories();
    registerHipVVKernelFactories();            // idempotent: must not add a second platform
    Platform& hip = Platform::getPlatformByName("HIP");
    ProbeIntegrator probe(300, 10, 1, 40, 0.001);
    int ok = 1;
    for (const std::string& n : probe.names()) {
        std::printf("kernel %s: %s\n", n.c_str(), hip.hasKernelFactory(n) ? "registered" : "MISSING");
        ok &= hip.hasKernelFactory(n);
    }
    HipVVKernelFactory f;
    System sys; sys.addParticle(1.0);
    Context ctx(sys, probe, hip);
    HipPlatform::PlatformData pd; pd.contexts.push_back(NULL);
    ctx.getImpl().setPlatformData(&pd);
    try { hip.createKernel("NoSuchKernel", ctx.getImpl()); ok = 0; } catch (const OpenMMException& e) { std::printf("unknown via platform: %s\n", e.what()); }
    try { probe.step(1); ok = 0; } catch (const OpenMMException& e) { std::printf("unbound step: %s\n", e.what()); }
    std::printf(ok ? "REGISTRY OK\n" : "REGISTRY FAILED\n");
    return ok ? 0 : 1;
}

static int chain() {
    ProbeIntegrator it(333.0, 10.0, 1.0, 40.0, 0.001, 3, 2);
    std::vector<double> eta = {0.01, -0.02, 0.03}, ed = {0.5, -1.5, 2.5, 0.0}, edd = {0.1, 0.2, 0.3}, mass = {4500.0, 0.0277, 0.0277};
    double scale = 0;
    it.propagateNHChain(eta, ed, edd, mass, 460000.0, 457000.0, 333.0, scale);
    std::printf("%.17g", scale);
    for (double v : eta) std::printf(" %.17g", v);
    for (double v : ed) std::printf(" %.17g", v);
    for (double v : edd) std::printf(" %.17g", v);
    std::printf("\n");
    return 0;
}

struct ForceUser { HipContext* cu; HipArray* site; };
static void tether(ContextImpl&, void* user) {
    ForceUser* u = (ForceUser*) user;
    std::shared_ptr<HipVVPlan> plan = HipVVPlan::find(*u->cu);
    plan->check(vvhip_synth_tether_force(plan->get(), u->site->getDevicePointer(), 1000.0, 209200.0));
}

template <class T> static void put(std::ofstream& f, const std::vector<T>& v) { long long n = (long long) v.size(); f.write((const char*) &n, 8); f.write((const char*) v.data(), n * sizeof(T)); }

static int run(const char* out, bool middle, int consMode, double cosacc, int nsteps, int hostMode = 0) {
    registerHipVVKernelFactories();
    Platform& hip = Platform::getPlatformByName("HIP");
    const int nmol = 40, per = 8;              // [heavy, drude, heavy, drude, heavy, drude, H, H] per molecule
    const int nLiq = nmol * per;
    System system;
    DrudeForce* drude = new DrudeForce();
    std::vector<double> masses, charges;
    std::vector<int> molId;
    std::vector<std::vector<int> > molecules(nmol);
    std::vector<int> pairs;
    for (int m = 0; m < nmol; m++)
        for (int k = 0; k < per; k++) {
            const int i = m * per + k;
            const bool isDrude = k < 6 && (k & 1);
            const double mass = k >= 6 ? 1.008 : (isDrude ? 0.4 : 11.611 + (k == 0 ? 1.996 : 0.0));
            system.addParticle(mass);
            masses.push_back(mass); charges.push_back(isDrude ? -2.0 : (k < 6 ? 2.1 : 0.1)); molId.push_back(m);
            molecules[m].push_back(i);
            if (isDrude) { drude->addParticle(i, i - 1, -1, -1, -1, -2.0, 0.001, 1, 1); pairs.push_back(i); pairs.push_back(i - 1); }
        }
    // consMode 3 ("electrode"): 16 Langevin-thermostatted wall atoms (each its own molecule), a massless image of every liquid particle in its
    // parent's molecule, the liquid in the electrolyte set -- the machinery of examples/run-edl.py (Langevin force, field force, image mirror)
    // consMode 5: one virtual site per molecule behind the liquid (a lone pair in local coordinates on even molecules, a three-particle average
    // on odd ones, hanging on the three heavy particles), in its molecule: kernel B places them, OpenMM's computeVirtualSites is not called
    const int nSites = consMode == 5 ? nmol : 0;
    const int nLiquid = nLiq, nWall = consMode == 3 ? 16 : 0, nImages = consMode == 3 ? nLiquid : 0, nAll = nLiquid + nWall + nImages + nSites;
    for (int w = 0; w < nWall; w++) {
        system.addParticle(32.06);
        masses.push_back(32.06); charges.push_back(0.0); molId.push_back(nmol + w);
        molecules.push_back(std::vector<int>(1, nLiquid + w));
    }
    for (int k = 0; k < nImages; k++) {
        system.addParticle(0.0);
        masses.push_back(0.0); charges.push_back(-charges[k]); molId.push_back(molId[k]);
        molecules[molId[k]].push_back(nLiquid + nWall + k);
    }
    std::vector<int> siteList;
    std::vector<double> siteParams;
    for (int m = 0; m < nSites; m++) {
        const int i = nLiquid + m, p1 = m * per, p2 = m * per + 2, p3 = m * per + 4;
        system.addParticle(0.0);
        masses.push_back(0.0); charges.push_back(-0.3); molId.push_back(m);
        molecules[m].push_back(i);
        double w[12] = {0};
        if (m % 2 == 0) {
            const double lone[12] = {1.0, 0.0, 0.0, 1.0, -1.0, 0.0, 0.0, -1.0, 1.0, 0.03, 0.02, -0.01};
            std::copy(lone, lone + 12, w);
            system.setVirtualSite(i, new LocalCoordinatesSite(p1, p2, p3, Vec3{w[0], w[1], w[2]}, Vec3{w[3], w[4], w[5]}, Vec3{w[6], w[7], w[8]}, Vec3{w[9], w[10], w[11]}));
        } else {
            w[0] = 0.5; w[1] = 0.3; w[2] = 0.2;
            system.setVirtualSite(i, new ThreeParticleAverageSite(p1, p2, p3, w[0], w[1], w[2]));
        }
        siteList.push_back(i); siteList.push_back(m % 2 == 0 ? 3 : 1); siteList.push_back(p1); siteList.push_back(p2); siteList.push_back(p3);
        siteParams.insert(siteParams.end(), w, w + 12);
    }
    system.addForce(drude);
    if (consMode != 3) system.addForce(new CMMotionRemover());
    // consMode 1: particle 4 of molecule 0 tied to a hydrogen of 17 other molecules -- more constraints on one particle than a wave's colouring
    //             takes, so the plan leaves constraints to OpenMM's solver and VVIntegrator takes the un-fused path (the stand-in solver is a
    //             no-op; only the path and the DOF matter);
    // consMode 4: a chain H-heavy-H-H in molecule 0 -- neither a hydrogen-type cluster nor a rigid triangle: a general cluster, solved inside
    //             the fused kernels by coloured sweeps;
    // consMode 2: both hydrogens of every molecule constrained to the heavy particle 4 -- solved inside the fused kernels.
    std::vector<int> cons;
    std::vector<double> consDist;
    auto addCons = [&](int a, int b, double d) { system.addConstraint(a, b, d); cons.push_back(a); cons.push_back(b); consDist.push_back(d); };
    if (consMode == 1)
        for (int m = 1; m <= 17; m++) addCons(m * per + 6, 4, 0.5);
    if (consMode == 4) { addCons(6, 4, 0.1); addCons(7, 6, 0.16); }
    if (consMode == 2)
        for (int m = 0; m < nmol; m++) { addCons(m * per + 6, m * per + 4, 0.1); addCons(m * per + 7, m * per + 4, 0.1); }
    const double box[3] = {3.0, 3.0, 3.0}, kB = (1.380649e-23 * 6.02214076e23) / 1000.0;
    std::vector<double> pos(3 * nAll, 0.0), vel(3 * nAll, 0.0);
    const double mirror = 3.5;
    for (int m = 0; m < nmol; m++) {
        double c[3] = {uniform() * box[0], uniform() * box[1], uniform() * box[2]};
        for (int k = 0; k < per; k++) {
            const int i = m * per + k;
            const bool isDrude = k < 6 && (k & 1);
            for (int d = 0; d < 3; d++) {
                pos[3 * i + d] = isDrude ? pos[3 * (i - 1) + d] + 2e-4 * gauss() : c[d] + 0.15 * (2 * uniform() - 1);
                vel[3 * i + d] = gauss() * std::sqrt(kB * (isDrude ? 30.0 : 333.0) / masses[i]);
            }
        }
    }
    if (consMode == 2 || consMode == 4)         // start on the constraint manifold: |r| = d, no relative velocity along the bond
        for (size_t c = 0; c < consDist.size(); c++) {
            const int h = cons[2 * c], o = cons[2 * c + 1];
            double u[3], len = 0, along = 0;
            for (int d = 0; d < 3; d++) { u[d] = pos[3 * h + d] - pos[3 * o + d]; len += u[d] * u[d]; }
            len = std::sqrt(len);
            for (int d = 0; d < 3; d++) { u[d] /= len; pos[3 * h + d] = pos[3 * o + d] + consDist[c] * u[d]; along += (vel[3 * h + d] - vel[3 * o + d]) * u[d]; }
            for (int d = 0; d < 3; d++) vel[3 * h + d] -= along * u[d];
        }
    for (int w = 0; w < nWall; w++)
        for (int d = 0; d < 3; d++) {
            pos[3 * (nLiquid + w) + d] = d < 2 ? uniform() * box[d] : 0.05 + 0.1 * uniform();
            vel[3 * (nLiquid + w) + d] = gauss() * std::sqrt(kB * 333.0 / 32.06);
        }
    for (int k = 0; k < nImages; k++) {
        const int i = nLiquid + nWall + k;
        pos[3 * i] = pos[3 * k]; pos[3 * i + 1] = pos[3 * k + 1]; pos[3 * i + 2] = 2 * mirror - pos[3 * k + 2];
    }
    ProbeIntegrator it(333.0, 10.0, 1.0, 40.0, 0.001);
    it.setMaxDrudeDistance(0.02);
    it.setUseMiddleScheme(middle);
    it.setCosAcceleration(cosacc);
    std::vector<int> ldList, imgList, elList;
    if (consMode == 3) {
        it.setMirrorLocation(mirror);
        it.setElectricField(2.0 / box[2] * 1.602176634e-22);
        for (int w = 0; w < nWall; w++) { it.addParticleLangevin(nLiquid + w); ldList.push_back(nLiquid + w); }
        for (int k = 0; k < nImages; k++) { it.addImagePair(nLiquid + nWall + k, k); imgList.push_back(nLiquid + nWall + k); imgList.push_back(k); }
        for (int k = 0; k < nLiquid; k++) { it.addParticleElectrolyte(k); elList.push_back(k); }
    }
    const int n = nAll;                        // (shadows the liquid's particle count from here on)
    Context ctx(system, it, hip);
    HipContext cu(n, false, true);             // HipPrecision = mixed
    std::vector<float> normals(4 * 2048);
    for (float& v : normals) v = (float) gauss();
    if (consMode == 3) { cu.getIntegrationUtilities().getRandom().initialize(2048, 16); cu.getIntegrationUtilities().getRandom().upload(normals.data()); }
    cu.setPeriodicBoxSize(box[0], box[1], box[2]);
    std::vector<double> velm(4 * n);
    std::vector<float> posq(4 * n), corr(4 * n, 0.f);
    for (int i = 0; i < n; i++) {
        for (int d = 0; d < 3; d++) {
            velm[4 * i + d] = vel[3 * i + d];
            posq[4 * i + d] = (float) pos[3 * i + d];
            corr[4 * i + d] = (float) (pos[3 * i + d] - (double) posq[4 * i + d]);
        }
        velm[4 * i + 3] = masses[i] != 0 ? 1.0 / masses[i] : 0.0;      // OpenMM: inverse mass 0 for massless particles
        posq[4 * i + 3] = (float) charges[i];
    }
    cu.getVelm().upload(velm.data()); cu.getPosq().upload(posq.data()); cu.getPosqCorrection().upload(corr.data());
    HipArray site; site.initialize(n, 16); site.upload(posq.data());
    HipPlatform::PlatformData pd; pd.contexts.push_back(&cu);
    cu.setPlatformData(&pd);
    ctx.getImpl().setPlatformData(&pd);
    ctx.getImpl().setMolecules(molecules);
    ForceUser fu = {&cu, &site};
    ctx.getImpl().setForceCallback(tether, &fu);
    ctx.initialize();
    (void) hipDeviceSynchronize();
    Kernel vv, nh;                             // hostMode 1 / 2: the hand-driven kernel objects (alive until the counters are printed)
    const auto t0 = std::chrono::steady_clock::now();
    long stepsDone = 0, stepsInterrupted = 0;
    if (hostMode == 10 || hostMode == 11) {
        // Fuzz: `nsteps` random operations.  Between steps (both modes): another step size / temperature / cos acceleration (also 0: the
        // cos stages then drop out of the sequence) / box, a kinetic-energy or viscosity query.  hostMode 11 drives the middle scheme's
        // stages by hand (plain thermostat sequence) and puts the same changes and queries at a random place INSIDE a step as well:
        // what has been recorded then belongs to the old parameters and must run before the change shows.
        const double dts[3] = {0.0005, 0.001, 0.00125}, temps[3] = {300.0, 333.0, 350.0}, coss[3] = {cosacc, 0.5 * cosacc, 0.0}, boxes[2] = {3.0, 3.1};
        const unsigned mask = std::getenv("VV_FUZZ_MASK") ? (unsigned) std::atoi(std::getenv("VV_FUZZ_MASK")) : 63u;      // bisecting: allowed kinds
        auto change = [&](unsigned what) {
            if (!((mask >> what) & 1u)) return;
            switch (what) {
                case 0: it.setStepSize(dts[op_next(3)]); break;
                case 1: it.setTemperature(temps[op_next(3)]); break;
                case 2: if (cosacc != 0 && hostMode == 10) it.setCosAcceleration(coss[op_next(3)]); break;
                case 3: { const double b = boxes[op_next(2)]; cu.setPeriodicBoxSize(b, b, b); break; }
                case 4: (void) it.kineticEnergy(); break;
                default: if (cosacc != 0) (void) it.getViscosity(); break;
            }
        };
        if (hostMode == 11) {
            vv = hip.createKernel(IntegrateMiddleStepKernel::Name(), ctx.getImpl());
            nh = hip.createKernel(ModifyDrudeNoseKernel::Name(), ctx.getImpl());
            vv.getAs<IntegrateMiddleStepKernel>().initialize(system, it, drude);
            nh.getAs<ModifyDrudeNoseKernel>().initialize(system, it, drude);
        }
        // VV_FUZZ_TRACE=file: one line per operation with a checksum of the velocities after it (bisecting a divergence between two runs)
        std::FILE* trace = std::getenv("VV_FUZZ_TRACE") ? std::fopen(std::getenv("VV_FUZZ_TRACE"), "w") : nullptr;
        auto trace_op = [&](int op, const char* what) {
            if (!trace) return;
            (void) hipDeviceSynchronize();
            cu.getVelm().download(velm.data());
            double sum = 0, sq = 0;
            const double* v = reinterpret_cast<const double*>(velm.data());
            const size_t n = velm.size() * sizeof(velm[0]) / sizeof(double);
            for (size_t i = 0; i < n; i++) { sum += v[i]; sq += v[i] * v[i]; }
            std::fprintf(trace, "%d %s dt=%g cos=%g sum=%.17g sq=%.17g\n", op, what, it.getStepSize(), it.getCosAcceleration(), sum, sq);
            std::fflush(trace);
        };
        for (int op = 0; op < nsteps; op++) {
            trace_op(op, "before");
            if (op_next(3) != 0) {                       // two operations in three are steps
                if (hostMode == 10) { const int k = 1 + (int) op_next(3); it.step(k); stepsDone += k; continue; }
                const unsigned where = op_next(6);       // 0..2: a change / query behind that stage; 3..5: none inside this step
                ctx.getImpl().calcForcesAndEnergy(true, false);
                vv.getAs<IntegrateMiddleStepKernel>().firstIntegrate(ctx.getImpl(), it);
                if (where == 0) change(op_next(5));
                nh.getAs<ModifyDrudeNoseKernel>().scaleVelocity(ctx.getImpl(), it);
                if (where == 1) change(op_next(5));
                vv.getAs<IntegrateMiddleStepKernel>().secondIntegrate(ctx.getImpl(), it);
                stepsDone++;
                if (where < 2) stepsInterrupted++;
            } else {
                change(op_next(6));
            }
        }
    } else if (hostMode == 0) {
        it.step(nsteps);
    } else {
        // A host that is neither VVIntegrator: its own kernel objects, driven by hand through the KernelImpl virtuals (middle scheme, no
        // modifiers).  hostMode 1: the reference's order (VVIntegrator.cpp:238-262) -- the adapters' deferred fusion must recognise it;
        // hostMode 2: a kinetic-energy query between firstIntegrate and scaleVelocity -- not the reference's order: the recorded stage must
        // run before the query is answered, the remaining stages one by one, and the trajectory must equal that of hostMode 1.
        vv = hip.createKernel(IntegrateMiddleStepKernel::Name(), ctx.getImpl());
        nh = hip.createKernel(ModifyDrudeNoseKernel::Name(), ctx.getImpl());
        vv.getAs<IntegrateMiddleStepKernel>().initialize(system, it, drude);
        nh.getAs<ModifyDrudeNoseKernel>().initialize(system, it, drude);
        for (int s = 0; s < nsteps; s++) {
            ctx.getImpl().calcForcesAndEnergy(true, false);
            vv.getAs<IntegrateMiddleStepKernel>().firstIntegrate(ctx.getImpl(), it);
            if (hostMode == 2) (void) vv.getAs<IntegrateMiddleStepKernel>().computeKineticEnergy(ctx.getImpl(), it);
            nh.getAs<ModifyDrudeNoseKernel>().scaleVelocity(ctx.getImpl(), it);
            vv.getAs<IntegrateMiddleStepKernel>().secondIntegrate(ctx.getImpl(), it);
        }
    }
    (void) hipDeviceSynchronize();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("TIMING steps=%d wall_s=%.6f steps_per_s=%.1f (host-launched, no graph)\n", nsteps, wall, nsteps / wall);
    std::vector<double> vis = it.getViscosity();
    (void) hipDeviceSynchronize();
    cu.getVelm().download(velm.data()); cu.getPosq().download(posq.data()); cu.getPosqCorrection().download(corr.data());
    std::ofstream f(out, std::ios::binary);
    put(f, masses); put(f, charges); put(f, molId); put(f, pairs); put(f, cons); put(f, pos); put(f, vel);
    put(f, velm); put(f, posq); put(f, corr); put(f, vis); put(f, consDist);
    put(f, ldList); put(f, imgList); put(f, elList); put(f, normals);
    put(f, siteList); put(f, siteParams);
    std::printf("RUN OK steps=%d time=%.6f stepCount=%lld vMax=%.9g\n", nsteps, cu.getTime(), cu.getStepCount(), vis[0]);
    if (hostMode >= 10) std::printf("FUZZ steps=%ld interrupted=%ld\n", stepsDone, stepsInterrupted);
    // how the adapters used the context services (the reference's pattern: HOST:60-63, 136-141, 214-216, 307-319)
    double ss[2] = {-1, -1};
    cu.getIntegrationUtilities().getStepSize().download(ss);
    const HipIntegrationUtilities::Calls& c = cu.getIntegrationUtilities().calls;
    std::printf("SERVICES selector=%d depth=%d initializeContexts=%d initRandom=%d setNextStepSize=%d stepSize=(%.6g,%.6g) virtualSites=%d reorder=%d "
                "applyConstraints=%d applyVelocityConstraints=%d setAsCurrent=%d\n", cu.selectorUses, cu.selectorDepth, pd.initializeContextsCalls,
                c.initRandom, c.setNextStepSize, ss[0], ss[1], c.computeVirtualSites, cu.reorderCalls, c.applyConstraints, c.applyVelocityConstraints,
                cu.setAsCurrentCalls);
    {   // deferred fusion in the adapters (HipVVKernels.h): fused steps launched on a completed stage sequence / stages run one by one
        std::shared_ptr<HipVVPlan> plan = HipVVPlan::find(cu);
        std::printf("DEFER fused=%ld staged=%ld\n", plan->fusedSteps, plan->stagedCalls);
    }
    return 0;
}

int main(int argc, char** argv) {
    try {
        if (argc >= 2 && !std::strcmp(argv[1], "registry")) return registry();
        if (argc >= 2 && !std::strcmp(argv[1], "chain")) return chain();
        if (argc >= 7 && !std::strcmp(argv[1], "run")) return run(argv[2], std::atoi(argv[3]) != 0, std::atoi(argv[4]), std::atof(argv[5]), std::atoi(argv[6]), argc >= 8 ? std::atoi(argv[7]) : 0);
        if (argc >= 9 && !std::strcmp(argv[1], "fuzz")) {
            op_state = 0x9E3779B97F4A7C15ull ^ (unsigned long long) std::atoll(argv[7]);
            return run(argv[2], std::atoi(argv[3]) != 0, std::atoi(argv[4]), std::atof(argv[5]), std::atoi(argv[6]), std::atoi(argv[8]) ? 11 : 10);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: vv_plugin_driver registry | chain | run OUT middle cons cos nsteps [hostMode]\n");
    return 64;
}
