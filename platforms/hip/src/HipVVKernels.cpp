// KernelImpl adapters: OpenMM objects in, C ABI calls out.  Counterpart of platforms/cuda/src/CudaVVKernels.cpp
// ("HOST"), minus everything that moved into libvvhip (table building, launches, the thermostat host round trip).
#include "HipVVKernels.h"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <map>
#include <mutex>
#include <typeinfo>

#include "openmm/CMMotionRemover.h"
#include "openmm/DrudeForce.h"
#include "openmm/OpenMMException.h"
#include "openmm/internal/ContextImpl.h"

using namespace OpenMM;

namespace {
std::mutex registryLock;
std::map<HipContext*, std::weak_ptr<HipVVPlan> > registry;

vvhip_params paramsOf(const VVIntegrator& it) {
    vvhip_params p;
    p.temperature = it.getTemperature(); p.frequency = it.getFrequency();
    p.drude_temperature = it.getDrudeTemperature(); p.drude_frequency = it.getDrudeFrequency();
    p.step_size = it.getStepSize();
    p.num_nh_chains = it.getNumNHChains(); p.loops_per_step = it.getLoopsPerStep();
    p.max_drude_distance = it.getMaxDrudeDistance();
    p.friction = it.getFriction(); p.drude_friction = it.getDrudeFriction();
    p.mirror_location = it.getMirrorLocation(); p.electric_field = it.getElectricField(); p.cos_acceleration = it.getCosAcceleration();
    p.use_com_temp_group = it.getUseCOMTempGroup(); p.use_middle_scheme = it.getUseMiddleScheme();
    p.auto_set_com_temp_group = 0;      // VVIntegrator::initialize has already applied the auto rules
    p.auto_set_friction = 0;
    p.constraint_tolerance = it.getConstraintTolerance();
    return p;
}
bool sameParams(const vvhip_params& a, const vvhip_params& b) {
    return a.temperature == b.temperature && a.frequency == b.frequency && a.drude_temperature == b.drude_temperature &&
           a.drude_frequency == b.drude_frequency && a.step_size == b.step_size && a.loops_per_step == b.loops_per_step &&
           a.max_drude_distance == b.max_drude_distance && a.friction == b.friction && a.drude_friction == b.drude_friction &&
           a.mirror_location == b.mirror_location && a.electric_field == b.electric_field && a.cos_acceleration == b.cos_acceleration &&
           a.constraint_tolerance == b.constraint_tolerance;      // the reference hands getConstraintTolerance() to the solver at every call (HOST:151,176)
}
void boxOf(HipContext& cu, double out[3]) {      // cu.getPeriodicBoxSize() is a double4 by value (HOST:1129-1130)
    const double4 box = cu.getPeriodicBoxSize();
    out[0] = box.x; out[1] = box.y; out[2] = box.z;
}
}  // namespace

// ------------------------------------------------------------------------------------------ shared plan
HipVVPlan::HipVVPlan(HipContext& cu, const System& system, const VVIntegrator& it, const DrudeForce* force) : cu(cu), plan(NULL) {
    const int n = system.getNumParticles();
    std::vector<double> masses(n);
    std::vector<double> consDist;
    std::vector<int32_t> molId(n), pairs, cons, ld(it.getParticlesLD().begin(), it.getParticlesLD().end()), img,
        el(it.getParticlesElectrolyte().begin(), it.getParticlesElectrolyte().end());
    for (int i = 0; i < n; i++) { masses[i] = system.getParticleMass(i); molId[i] = it.getParticleMolId(i); }
    if (force != NULL)
        for (int i = 0; i < force->getNumParticles(); i++) {            // HOST:68-73
            int p, p1, p2, p3, p4; double q, pol, a12, a34;
            force->getParticleParameters(i, p, p1, p2, p3, p4, q, pol, a12, a34);
            pairs.push_back(p); pairs.push_back(p1);
        }
    for (int i = 0; i < system.getNumConstraints(); i++) {
        int a, b; double d;
        system.getConstraintParameters(i, a, b, d);
        cons.push_back(a); cons.push_back(b);
        consDist.push_back(d);
    }
    // Virtual sites of the four classes the fused step can place itself (include/vvhip.h: virtual_sites); any other kind, and the whole list
    // stays with OpenMM's computeVirtualSites
    std::vector<int32_t> sites;
    std::vector<double> siteParams;
    bool sitesDescribed = true;
    for (int i = 0; i < n && sitesDescribed; i++) {
        if (!system.isVirtualSite(i)) continue;
        const VirtualSite& vs = system.getVirtualSite(i);
        double w[12] = {0};
        int kind = -1;
        if (const TwoParticleAverageSite* s2 = dynamic_cast<const TwoParticleAverageSite*>(&vs)) {
            kind = VVHIP_VSITE_AVERAGE2; w[0] = s2->getWeight(0); w[1] = s2->getWeight(1);
        } else if (const ThreeParticleAverageSite* s3 = dynamic_cast<const ThreeParticleAverageSite*>(&vs)) {
            kind = VVHIP_VSITE_AVERAGE3; w[0] = s3->getWeight(0); w[1] = s3->getWeight(1); w[2] = s3->getWeight(2);
        } else if (const OutOfPlaneSite* so = dynamic_cast<const OutOfPlaneSite*>(&vs)) {
            kind = VVHIP_VSITE_OUT_OF_PLANE; w[0] = so->getWeight12(); w[1] = so->getWeight13(); w[2] = so->getWeightCross();
        } else if (const LocalCoordinatesSite* sl = dynamic_cast<const LocalCoordinatesSite*>(&vs)) {
            std::vector<double> ow, xw, yw;
            sl->getOriginWeights(ow); sl->getXWeights(xw); sl->getYWeights(yw);
            if (vs.getNumParticles() == 3 && ow.size() == 3 && xw.size() == 3 && yw.size() == 3) {
                kind = VVHIP_VSITE_LOCAL_COORDS;
                for (int k = 0; k < 3; k++) { w[k] = ow[k]; w[3 + k] = xw[k]; w[6 + k] = yw[k]; }
                const Vec3 lp = sl->getLocalPosition();
                w[9] = lp[0]; w[10] = lp[1]; w[11] = lp[2];
            }
        }
        if (kind < 0) { sitesDescribed = false; break; }
        sites.push_back(i); sites.push_back(kind);
        for (int k = 0; k < 3; k++) sites.push_back(k < vs.getNumParticles() ? vs.getParticle(k) : -1);
        siteParams.insert(siteParams.end(), w, w + 12);
    }
    bool cmm = false;
    for (int i = 0; i < system.getNumForces(); i++)                     // HOST:550-558
        if (dynamic_cast<const CMMotionRemover*>(&system.getForce(i)) != NULL) cmm = true;
    for (const auto& pr : it.getImagePairs()) { img.push_back(pr.first); img.push_back(pr.second); }
    vvhip_system_desc d = {};
    d.num_atoms = n; d.padded_num_atoms = cu.getPaddedNumAtoms();
    d.masses = masses.data(); d.mol_id = molId.data(); d.num_molecules = it.getNumMolecules();
    d.num_drude_pairs = (int) pairs.size() / 2; d.drude_pairs = pairs.data();
    d.num_constraints = (int) cons.size() / 2; d.constraints = cons.data(); d.constraint_distances = consDist.empty() ? NULL : consDist.data();
    d.has_cm_motion_remover = cmm;
    d.num_particles_ld = (int) ld.size(); d.particles_ld = ld.data();
    d.num_image_pairs = (int) img.size() / 2; d.image_pairs = img.data();
    d.num_electrolyte = (int) el.size(); d.particles_electrolyte = el.data();
    if (sitesDescribed && !sites.empty()) { d.num_virtual_sites = (int) sites.size() / 5; d.virtual_sites = sites.data(); d.virtual_site_params = siteParams.data(); }
    last = paramsOf(it);
    const int precision = cu.getUseDoublePrecision() ? VVHIP_DOUBLE : (cu.getUseMixedPrecision() ? VVHIP_MIXED : VVHIP_SINGLE);
    char err[512] = "";
    if (vvhip_plan_create(&d, &last, precision, &plan, err, sizeof(err)) != VVHIP_OK) throw OpenMMException(err);
    vvhip_plan_info info;
    vvhip_plan_get_info(plan, &info);
    ldRandoms = std::max(info.num_normal_ld, 1) + 2 * std::max(info.num_pairs_ld, 1);   // HOST:806-807,863: array sizes are max(n,1)
    noConstraints = info.constraints_fused != 0;      // no constraints at all, or all of them solved inside the kernels
    sitesInKernel = info.num_virtual_sites > 0;       // the fused steps place the virtual sites themselves
    if (const char* e = std::getenv("VVHIP_PLUGIN_DEFER")) deferEnabled = std::atoi(e) != 0;
    HipIntegrationUtilities& integration = cu.getIntegrationUtilities();
    vvhip_buffers b = {};
    // getDevicePointer() is an lvalue device-pointer handle in OpenMM (the reference passes its address as a kernel argument,
    // HOST:144-147); its value is what the C ABI takes
    b.velm = (void*) cu.getVelm().getDevicePointer(); b.posq = (void*) cu.getPosq().getDevicePointer();
    b.posq_correction = cu.getUseMixedPrecision() ? (void*) cu.getPosqCorrection().getDevicePointer() : NULL;
    b.force = (void*) cu.getForce().getDevicePointer(); b.pos_delta = (void*) integration.getPosDelta().getDevicePointer();
    b.random = (const void*) integration.getRandom().getDevicePointer(); b.random_size = (uint32_t) integration.getRandom().getSize();
    b.stream = (void*) cu.getCurrentStream();
    check(vvhip_bind(plan, &b));
    double box[3];
    boxOf(cu, box);
    check(vvhip_set_box(plan, box));
    std::cerr << "HIP velocity-Verlet plan: " << n << " particles in " << info.num_waves << " waves (" << info.num_slots_used
              << " lanes used), " << info.num_temp_groups << " temperature group(s), NH pairs " << info.num_pairs_nh
              << ", Langevin particles " << ld.size() << ", image pairs " << info.num_images << "\n";
    for (int g = 0; g < info.num_temp_groups; g++)
        std::cerr << "    DOF[" << g << "]: " << info.dof[g] << ", NkbT[" << g << "]: " << info.nkbt[g] << ", etaMass[" << g
                  << "]: " << info.eta_mass[g][0] << "\n";
}
HipVVPlan::~HipVVPlan() { vvhip_plan_destroy(plan); }
void HipVVPlan::check(int rc) const { if (rc != VVHIP_OK) throw OpenMMException(vvhip_last_error(plan)); }
void HipVVPlan::syncParameters(const VVIntegrator& it) {
    if (it.getDebugEnabled() != debug) {              // VVIntegrator.h:417-419: one line per kernel call on stdout, here plus roctx ranges
        debug = it.getDebugEnabled();
        check(vvhip_set_trace(plan, debug ? 1 : 0));
    }
    vvhip_params now = replaying ? pendingParams : paramsOf(it);
    if (!sameParams(now, last)) { check(vvhip_set_params(plan, &now)); last = now; }
    double box[3];
    if (replaying) { box[0] = pendingBox[0]; box[1] = pendingBox[1]; box[2] = pendingBox[2]; }
    else boxOf(cu, box);
    if (box[0] != lastBox[0] || box[1] != lastBox[1] || box[2] != lastBox[2]) {      // a barostat move: the box is read live (HOST:1057, 1129)
        check(vvhip_set_box(plan, box));
        lastBox[0] = box[0]; lastBox[1] = box[1]; lastBox[2] = box[2];
    }
}
std::shared_ptr<HipVVPlan> HipVVPlan::create(HipContext& cu, const System& s, const VVIntegrator& it, const DrudeForce* f) {
    std::shared_ptr<HipVVPlan> p(new HipVVPlan(cu, s, it, f));
    std::lock_guard<std::mutex> g(registryLock);
    registry[&cu] = p;
    return p;
}
std::shared_ptr<HipVVPlan> HipVVPlan::find(HipContext& cu) {
    std::lock_guard<std::mutex> g(registryLock);
    std::shared_ptr<HipVVPlan> p = registry[&cu].lock();
    if (!p) throw OpenMMException("the velocity-Verlet step kernel must be initialized before its modifier kernels");
    return p;
}

// ------------------------------------------------------------------------------------------ deferred fusion
double HipVVPlan::stepSizeOf(const VVIntegrator& it) const { return replaying ? pendingParams.step_size : it.getStepSize(); }
void HipVVPlan::flush() {
    std::vector<std::function<void()> > run;
    run.swap(pending);
    pattern.clear();
    replaying = true;            // the recorded stages run with the parameters and the box of the moment they were called
    try { for (auto& f : run) { f(); stagedCalls++; } } catch (...) { replaying = false; throw; }
    replaying = false;
}
bool HipVVPlan::defer(Stage stage, const VVIntegrator& it, std::function<void()> stageByStage) {
    if (!deferEnabled || !noConstraints) return false;
    if (it.getUseMiddleScheme() ? !fusedMiddle : !(fusedFirst && fusedSecond)) return false;      // the context's step kernel is not the scheme's
    if (pending.empty()) {
        // the sequence the reference's VVIntegrator produces for this configuration (VVIntegrator.cpp:238-267; 295-310 and 316-336)
        vvhip_plan_info info;
        vvhip_plan_get_info(plan, &info);
        const bool nh = info.num_particles_nh > 0, cos = it.getCosAcceleration() != 0, ld = !it.getParticlesLD().empty(), ef = !it.getParticlesElectrolyte().empty();
        std::vector<Stage> extra, thermo;
        if (ld || ef || cos) extra.push_back(ST_RESET);
        if (ld) extra.push_back(ST_LD);
        if (ef) extra.push_back(ST_EF);
        if (cos) extra.push_back(ST_COS);
        if (nh) { if (cos) { thermo.push_back(ST_CALCBIAS); thermo.push_back(ST_RMBIAS); } thermo.push_back(ST_SCALE); if (cos) thermo.push_back(ST_RESTORE); }
        pattern.clear();
        if (it.getUseMiddleScheme()) {
            pattern = extra; pattern.push_back(ST_FIRST); pattern.insert(pattern.end(), thermo.begin(), thermo.end()); pattern.push_back(ST_SECOND);
        } else {
            std::vector<Stage> first = thermo, second = extra;
            first.push_back(ST_FIRST);
            second.push_back(ST_SECOND); second.insert(second.end(), thermo.begin(), thermo.end());
            if (stage == first[0] && stage != second[0]) classicHalf = 0;
            else if (stage == second[0] && stage != first[0]) classicHalf = 1;
            pattern = classicHalf == 0 ? first : second;
        }
    }
    if (!pending.empty()) {      // parameters or box changed since the first recorded stage: that stage belongs to the old ones
        double box[3];
        boxOf(cu, box);
        if (!sameParams(paramsOf(it), pendingParams) || box[0] != pendingBox[0] || box[1] != pendingBox[1] || box[2] != pendingBox[2]) {
            flush();
            imagesFresh = false;
            stagedCalls++;
            return false;
        }
    }
    if (pending.size() < pattern.size() && pattern[pending.size()] == stage) {
        if (pending.empty()) { imagesFresh = false; pendingParams = paramsOf(it); boxOf(cu, pendingBox); }
        pending.push_back(std::move(stageByStage));
        if (pending.size() == pattern.size()) {            // complete: one fused step instead of the recorded stages
            pending.clear();
            pattern.clear();
            const uint32_t ri = pendingRandomIndex;
            if (it.getUseMiddleScheme()) { fusedMiddle(it, ri); imagesFresh = true; }
            else if (classicHalf == 0) { fusedFirst(it, ri); classicHalf = 1; imagesFresh = true; }      // (both mirror the images with their position update)
            else { fusedSecond(it, ri); classicHalf = 0; }
            fusedSteps++;
        }
        return true;
    }
    flush();                                               // not the reference's order: what was recorded runs now, then the caller's stage
    imagesFresh = false;
    stagedCalls++;
    return false;
}

// ------------------------------------------------------------------------------------------ step kernels
void HipVVStepCommon::create(const System& system, const VVIntegrator& it, const DrudeForce* force) {
    ContextSelector selector(cu);                                                                       // HOST:60, 246
    cu.getPlatformData().initializeContexts(system);                                                    // HOST:61, 247
    cu.getIntegrationUtilities().initRandomNumberGenerator((unsigned int) it.getRandomNumberSeed());   // HOST:63
    plan = HipVVPlan::create(cu, system, it, force);
    prevStepSize = -1.0;                                                                                // HOST:115
}
// OpenMM's own kernels (constraint solvers, virtual sites) read the step size from integration.getStepSize(): the middle kernel
// announces a change with setNextStepSize (HOST:136-141), the classic kernel uploads (0, dt) itself (HOST:307-319).
void HipVVStepCommon::announceStepSize(const VVIntegrator& it, bool classic) {
    const double stepSize = plan->stepSizeOf(it);
    if (stepSize == prevStepSize) return;
    HipIntegrationUtilities& integration = cu.getIntegrationUtilities();
    if (!classic) integration.setNextStepSize(stepSize);
    else if (cu.getUseDoublePrecision() || cu.getUseMixedPrecision()) { double ss[2] = {0.0, stepSize}; integration.getStepSize().upload(ss); }
    else { float ss[2] = {0.0f, (float) stepSize}; integration.getStepSize().upload(ss); }
    prevStepSize = stepSize;
}
void HipVVStepCommon::advanceClock(const VVIntegrator& it) {      // HOST:219-220, 430-431
    cu.setTime(cu.getTime() + plan->stepSizeOf(it));
    cu.setStepCount(cu.getStepCount() + 1);
}
uint32_t HipVVStepCommon::nextRandomIndex() { return (uint32_t) cu.getIntegrationUtilities().prepareRandomNumbers(plan->numLangevinRandoms()); }

void HipIntegrateMiddleStepKernel::initialize(const System& s, const VVIntegrator& it, const DrudeForce* f) {
    create(s, it, f);
    plan->fusedMiddle = [this](const VVIntegrator& integ, uint32_t randomIndex) { fusedMiddleWith(integ, randomIndex); };
}
void HipIntegrateMiddleStepKernel::resetExtraForce(ContextImpl&, const VVIntegrator& it) {
    if (plan->defer(HipVVPlan::ST_RESET, it, [this] { cu.setAsCurrent(); plan->check(vvhip_reset_extra_force(plan->get())); })) return;
    cu.setAsCurrent();
    plan->check(vvhip_reset_extra_force(plan->get()));
}
void HipIntegrateMiddleStepKernel::firstIntegrate(ContextImpl& context, const VVIntegrator& it) {      // HOST:129-159
    if (plan->defer(HipVVPlan::ST_FIRST, it, [this, &it] { firstIntegrateNow(it); })) return;
    firstIntegrateNow(it);
}
void HipIntegrateMiddleStepKernel::firstIntegrateNow(const VVIntegrator& it) {
    cu.setAsCurrent();
    plan->syncParameters(it);
    announceStepSize(it, false);
    if (it.getDebugEnabled()) std::cout << "HipIntegrateMiddleStepKernel firstIntegrate" << std::endl;
    plan->check(vvhip_middle_kick(plan->get()));
    cu.getIntegrationUtilities().applyVelocityConstraints(it.getConstraintTolerance());
    plan->check(vvhip_middle_half_drift1(plan->get()));
}
void HipIntegrateMiddleStepKernel::secondIntegrate(ContextImpl&, const VVIntegrator& it) {             // HOST:161-231
    if (plan->defer(HipVVPlan::ST_SECOND, it, [this, &it] { secondIntegrateNow(it); })) return;
    secondIntegrateNow(it);
}
void HipIntegrateMiddleStepKernel::secondIntegrateNow(const VVIntegrator& it) {
    cu.setAsCurrent();
    if (it.getDebugEnabled()) std::cout << "HipIntegrateMiddleStepKernel secondIntegrate" << std::endl;
    plan->check(vvhip_middle_half_drift2(plan->get()));
    cu.getIntegrationUtilities().applyConstraints(it.getConstraintTolerance());
    plan->check(vvhip_middle_finish(plan->get()));
    cu.getIntegrationUtilities().computeVirtualSites();
    cu.reorderAtoms();
    advanceClock(it);
}
double HipIntegrateMiddleStepKernel::computeKineticEnergy(ContextImpl&, const VVIntegrator&) {
    plan->flush();
    return cu.getIntegrationUtilities().computeKineticEnergy(0);                                        // HOST:233-235 (OpenMM's)
}
bool HipIntegrateMiddleStepKernel::canFuse(ContextImpl&, const VVIntegrator&) const { return plan->constraintFree(); }
void HipIntegrateMiddleStepKernel::fusedMiddleStep(ContextImpl&, const VVIntegrator& it) {
    plan->flush();
    fusedMiddleWith(it, it.getParticlesLD().empty() ? 0 : nextRandomIndex());
}
void HipIntegrateMiddleStepKernel::fusedMiddleWith(const VVIntegrator& it, uint32_t randomIndex) {
    cu.setAsCurrent();
    plan->syncParameters(it);
    announceStepSize(it, false);
    if (it.getDebugEnabled()) std::cout << "HipIntegrateMiddleStepKernel fusedMiddleStep" << std::endl;
    plan->check(vvhip_step_middle(plan->get(), randomIndex));
    // as the un-fused path and the reference after every position update (HOST:214), unless kernel B has placed the sites already
    if (!plan->placesVirtualSites()) cu.getIntegrationUtilities().computeVirtualSites();
    cu.reorderAtoms();
    advanceClock(it);
}
void HipIntegrateMiddleStepKernel::fusedVVFirstHalf(ContextImpl&, const VVIntegrator&) { throw OpenMMException("middle-scheme kernel asked for a classic step"); }
void HipIntegrateMiddleStepKernel::fusedVVSecondHalf(ContextImpl&, const VVIntegrator&) { throw OpenMMException("middle-scheme kernel asked for a classic step"); }

void HipIntegrateVVStepKernel::initialize(const System& s, const VVIntegrator& it, const DrudeForce* f) {
    create(s, it, f);
    plan->fusedFirst = [this](const VVIntegrator& integ, uint32_t) { fusedFirstNow(integ); };
    plan->fusedSecond = [this](const VVIntegrator& integ, uint32_t randomIndex) { fusedSecondWith(integ, randomIndex); };
}
void HipIntegrateVVStepKernel::resetExtraForce(ContextImpl&, const VVIntegrator& it) {
    if (plan->defer(HipVVPlan::ST_RESET, it, [this] { cu.setAsCurrent(); plan->check(vvhip_reset_extra_force(plan->get())); })) return;
    cu.setAsCurrent();
    plan->check(vvhip_reset_extra_force(plan->get()));
}
void HipIntegrateVVStepKernel::firstIntegrate(ContextImpl&, const VVIntegrator& it) {                   // HOST:296-382
    if (plan->defer(HipVVPlan::ST_FIRST, it, [this, &it] { firstIntegrateNow(it); })) return;
    firstIntegrateNow(it);
}
void HipIntegrateVVStepKernel::firstIntegrateNow(const VVIntegrator& it) {
    cu.setAsCurrent();
    plan->syncParameters(it);
    announceStepSize(it, true);
    plan->check(vvhip_vv_half_kick(plan->get(), 1));
    cu.getIntegrationUtilities().applyConstraints(it.getConstraintTolerance());
    plan->check(vvhip_vv_positions(plan->get()));
    cu.getIntegrationUtilities().computeVirtualSites();
    cu.reorderAtoms();                                   // after the first half, so Langevin indices stay valid (HOST:376-381)
}
void HipIntegrateVVStepKernel::secondIntegrate(ContextImpl&, const VVIntegrator& it) {                  // HOST:395-442
    if (plan->defer(HipVVPlan::ST_SECOND, it, [this, &it] { secondIntegrateNow(it); })) return;
    secondIntegrateNow(it);
}
void HipIntegrateVVStepKernel::secondIntegrateNow(const VVIntegrator& it) {
    cu.setAsCurrent();
    plan->syncParameters(it);
    plan->check(vvhip_vv_half_kick(plan->get(), 0));
    cu.getIntegrationUtilities().applyVelocityConstraints(it.getConstraintTolerance());
    advanceClock(it);
}
double HipIntegrateVVStepKernel::computeKineticEnergy(ContextImpl&, const VVIntegrator&) { plan->flush(); return cu.getIntegrationUtilities().computeKineticEnergy(0); }
bool HipIntegrateVVStepKernel::canFuse(ContextImpl&, const VVIntegrator&) const { return plan->constraintFree(); }
void HipIntegrateVVStepKernel::fusedMiddleStep(ContextImpl&, const VVIntegrator&) { throw OpenMMException("classic kernel asked for a middle-scheme step"); }
void HipIntegrateVVStepKernel::fusedVVFirstHalf(ContextImpl&, const VVIntegrator& it) { plan->flush(); fusedFirstNow(it); }
void HipIntegrateVVStepKernel::fusedFirstNow(const VVIntegrator& it) {
    cu.setAsCurrent();
    plan->syncParameters(it);
    announceStepSize(it, true);
    plan->check(vvhip_step_vv_first(plan->get()));
    if (!plan->placesVirtualSites()) cu.getIntegrationUtilities().computeVirtualSites();      // HOST:374
    cu.reorderAtoms();
}
void HipIntegrateVVStepKernel::fusedVVSecondHalf(ContextImpl&, const VVIntegrator& it) {
    plan->flush();
    fusedSecondWith(it, it.getParticlesLD().empty() ? 0 : nextRandomIndex());
}
void HipIntegrateVVStepKernel::fusedSecondWith(const VVIntegrator& it, uint32_t randomIndex) {
    cu.setAsCurrent();
    plan->syncParameters(it);
    plan->check(vvhip_step_vv_second(plan->get(), randomIndex));
    advanceClock(it);
}

// ------------------------------------------------------------------------------------------ modifier kernels
void HipModifyDrudeNoseKernel::initialize(const System&, const VVIntegrator&, const DrudeForce*) { ContextSelector selector(cu); plan = HipVVPlan::find(cu); }
void HipModifyDrudeNoseKernel::scaleVelocity(ContextImpl&, const VVIntegrator& it) {                    // HOST:670-754
    if (plan->defer(HipVVPlan::ST_SCALE, it, [this, &it] { scaleVelocityNow(it); })) return;
    scaleVelocityNow(it);
}
void HipModifyDrudeNoseKernel::scaleVelocityNow(const VVIntegrator& it) {
    cu.setAsCurrent();
    plan->syncParameters(it);
    plan->check(vvhip_scale_velocity(plan->get()));
}

void HipModifyDrudeLangevinKernel::initialize(const System&, const VVIntegrator&, const DrudeForce*, Kernel&) { ContextSelector selector(cu); plan = HipVVPlan::find(cu); }
void HipModifyDrudeLangevinKernel::applyLangevinForce(ContextImpl&, const VVIntegrator& it) {            // HOST:826-872
    cu.setAsCurrent();
    // the slice of the random buffer is reserved NOW, as the reference does (HOST:863), whether the stage runs now or inside the fused step
    const uint32_t randomIndex = (uint32_t) cu.getIntegrationUtilities().prepareRandomNumbers(plan->numLangevinRandoms());
    auto now = [this, &it, randomIndex] { cu.setAsCurrent(); plan->syncParameters(it); plan->check(vvhip_apply_langevin_force(plan->get(), randomIndex)); };
    plan->pendingRandomIndex = randomIndex;
    if (plan->defer(HipVVPlan::ST_LD, it, now)) return;
    now();
}

void HipModifyImageChargeKernel::initialize(const System&, const VVIntegrator&) { ContextSelector selector(cu); plan = HipVVPlan::find(cu); }
void HipModifyImageChargeKernel::updateImagePositions(ContextImpl&, const VVIntegrator& it) {            // HOST:904-934
    plan->flush();
    if (plan->imagesFresh) { plan->imagesFresh = false; return; }      // the fused step that has just run mirrored them with its position update
    cu.setAsCurrent();
    plan->syncParameters(it);
    plan->check(vvhip_update_image_positions(plan->get()));
}

void HipModifyElectricFieldKernel::initialize(const System&, const VVIntegrator&, Kernel&) { ContextSelector selector(cu); plan = HipVVPlan::find(cu); }
void HipModifyElectricFieldKernel::applyElectricForce(ContextImpl&, const VVIntegrator& it) {            // HOST:971-992
    auto now = [this, &it] { cu.setAsCurrent(); plan->syncParameters(it); plan->check(vvhip_apply_electric_force(plan->get())); };
    if (plan->defer(HipVVPlan::ST_EF, it, now)) return;
    now();
}

void HipModifyCosineAccelerateKernel::initialize(const System&, const VVIntegrator&, Kernel&) { ContextSelector selector(cu); plan = HipVVPlan::find(cu); }
void HipModifyCosineAccelerateKernel::applyCosineForce(ContextImpl&, const VVIntegrator& it) {
    auto now = [this, &it] { cu.setAsCurrent(); plan->syncParameters(it); plan->check(vvhip_apply_cosine_force(plan->get())); };
    if (plan->defer(HipVVPlan::ST_COS, it, now)) return;
    now();
}
void HipModifyCosineAccelerateKernel::calcVelocityBias(ContextImpl&, const VVIntegrator& it) {
    // (the box is read live here: HOST:1057 -- in the classic scheme this is the first call of a step, right behind a barostat move)
    auto now = [this, &it] { cu.setAsCurrent(); plan->syncParameters(it); plan->check(vvhip_calc_velocity_bias(plan->get())); };
    if (plan->defer(HipVVPlan::ST_CALCBIAS, it, now)) return;
    now();
}
void HipModifyCosineAccelerateKernel::removeVelocityBias(ContextImpl&, const VVIntegrator& it) {
    auto now = [this, &it] { cu.setAsCurrent(); plan->syncParameters(it); plan->check(vvhip_remove_velocity_bias(plan->get())); };
    if (plan->defer(HipVVPlan::ST_RMBIAS, it, now)) return;
    now();
}
void HipModifyCosineAccelerateKernel::restoreVelocityBias(ContextImpl&, const VVIntegrator& it) {
    auto now = [this, &it] { cu.setAsCurrent(); plan->syncParameters(it); plan->check(vvhip_restore_velocity_bias(plan->get())); };
    if (plan->defer(HipVVPlan::ST_RESTORE, it, now)) return;
    now();
}
void HipModifyCosineAccelerateKernel::calcViscosity(ContextImpl&, const VVIntegrator& it, double& vMax, double& invVis) {   // HOST:1112-1134
    plan->flush();
    cu.setAsCurrent();
    plan->syncParameters(it);
    plan->check(vvhip_calc_viscosity(plan->get(), &vMax, &invVis));
}
