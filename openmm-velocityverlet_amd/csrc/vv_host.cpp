#include <cstdio>
// vv_host.cpp -- see vv_host.hpp.  "API" = openmmapi/src/VVIntegrator.cpp, "HOST" =
// platforms/cuda/src/CudaVVKernels.cpp of the reference.
#include "vv_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace vv {

namespace {
// SimTKOpenMMRealType.h of OpenMM 8.1.2 (not vendored by the reference): BOLTZ = RGAS / KILO
constexpr double kAvogadro = 6.02214076e23;
constexpr double kBoltz = (1.380649e-23 * kAvogadro) / 1000.0;
enum { TG_ATOM = 0, TG_COM = 1, TG_DRUDE = 2 };

struct Cluster {            // particles that must share one wave
    int32_t first;          // smallest particle index (ordering key)
    std::vector<int32_t> members;
    bool com_segment;       // members form one molecular COM segment
    int32_t big = -1;       // >= 0: chunk of big molecule `big` (a molecule with more than 64 lanes, split over waves)
    bool big_first = false; // first chunk of that molecule
};
}  // namespace

HostPlan analyze(const vvhip_system_desc& sys, const vvhip_params& params_in, int precision) {
    if (precision != VVHIP_SINGLE && precision != VVHIP_MIXED && precision != VVHIP_DOUBLE)
        throw Error(VVHIP_ERR_INVALID, "unknown precision mode");
    const int n = sys.num_atoms;
    if (n <= 0 || !sys.masses || !sys.mol_id || sys.num_molecules <= 0)
        throw Error(VVHIP_ERR_INVALID, "system description is empty");
    if (sys.padded_num_atoms < n)
        throw Error(VVHIP_ERR_INVALID, "padded_num_atoms < num_atoms");
    if (params_in.num_nh_chains < 1 || params_in.num_nh_chains > VVHIP_MAX_CHAINS)
        throw Error(VVHIP_ERR_UNSUPPORTED, "numNHChains must be between 1 and " + std::to_string(VVHIP_MAX_CHAINS));
    if (params_in.loops_per_step < 1)
        throw Error(VVHIP_ERR_INVALID, "loopsPerStep must be >= 1");

    HostPlan hp;
    hp.precision = precision;
    hp.params = params_in;
    hp.num_atoms = n;
    hp.padded_num_atoms = sys.padded_num_atoms;
    vvhip_params& p = hp.params;
    const int nmol = sys.num_molecules;
    const int npairs_all = sys.num_drude_pairs;

    // ---- API:106-121: thermostat defaults depend on whether the System has Drude particles
    if (npairs_all == 0) {
        if (p.auto_set_com_temp_group) p.use_com_temp_group = 0;
        if (p.auto_set_friction) p.friction = 1.0;
    } else {
        if (p.auto_set_com_temp_group) p.use_com_temp_group = 1;
        if (p.auto_set_friction) p.friction = 5.0;
    }

    // ---- API:123-135: molecule masses
    for (int i = 0; i < n; i++)
        if (sys.mol_id[i] < 0 || sys.mol_id[i] >= nmol)
            throw Error(VVHIP_ERR_INVALID, "mol_id out of range");
    std::vector<double> mol_mass(nmol, 0.0), mol_inv_mass(nmol);
    for (int i = 0; i < n; i++) mol_mass[sys.mol_id[i]] += sys.masses[i];
    for (int m = 0; m < nmol; m++) mol_inv_mass[m] = 1.0 / mol_mass[m];

    // ---- API:138-151: NH / Langevin / image partition
    std::vector<char> is_ld(n, 0), is_img(n, 0), is_el(n, 0);
    auto check_index = [&](int i, const char* what) {
        if (i < 0 || i >= n) throw Error(VVHIP_ERR_INVALID, std::string(what) + " index out of range");
    };
    for (int k = 0; k < sys.num_particles_ld; k++) { check_index(sys.particles_ld[k], "Langevin particle"); is_ld[sys.particles_ld[k]] = 1; }
    std::vector<int32_t> image_of(n, -1);
    for (int k = 0; k < sys.num_image_pairs; k++) {
        int img = sys.image_pairs[2 * k], par = sys.image_pairs[2 * k + 1];
        check_index(img, "image particle"); check_index(par, "image parent");
        is_img[img] = 1;
        if (image_of[par] >= 0)
            throw Error(VVHIP_ERR_UNSUPPORTED, "a particle with more than one image particle is not supported");
        image_of[par] = img;
    }
    for (int k = 0; k < sys.num_electrolyte; k++) { check_index(sys.particles_electrolyte[k], "electrolyte particle"); is_el[sys.particles_electrolyte[k]] = 1; }
    std::vector<char> is_nh(n, 0), mol_is_nh(nmol, 0);
    for (int i = 0; i < n; i++) {
        if (!is_ld[i] && !is_img[i]) {
            is_nh[i] = 1;
            hp.particles_nh.push_back(i);
            if (!mol_is_nh[sys.mol_id[i]]) { mol_is_nh[sys.mol_id[i]] = 1; hp.molecules_nh.push_back(sys.mol_id[i]); }
        }
    }
    for (int i = 0; i < n; i++)
        if (is_ld[i] && mol_is_nh[sys.mol_id[i]])
            throw Error(VVHIP_ERR_TOPOLOGY, "NH and Langevin thermostat cannot be applied on the same molecule");
    if (sys.num_particles_ld > 0 && p.cos_acceleration != 0)   // API:154-155
        throw Error(VVHIP_ERR_TOPOLOGY, "Langevin thermostat and periodic perturbation shouldn't be used together");

    // ---- HOST:496-529: DOF of the atomic group, NH pairs
    double dof[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double mass = sys.masses[i];
        if (is_nh[i] && mass != 0.0) {
            dof[TG_ATOM] += 3;
            if (p.use_com_temp_group) dof[TG_ATOM] -= 3 * mass * mol_inv_mass[sys.mol_id[i]];
        }
    }
    std::vector<int32_t> partner(n, -1);
    std::vector<char> is_drude(n, 0), in_pair(n, 0);
    std::vector<char> nh_left(is_nh.begin(), is_nh.end()), ld_left(is_ld.begin(), is_ld.end());
    for (int k = 0; k < npairs_all; k++) {
        int d = sys.drude_pairs[2 * k], par = sys.drude_pairs[2 * k + 1];
        check_index(d, "Drude particle"); check_index(par, "Drude parent");
        if (in_pair[d] || in_pair[par] || d == par)
            throw Error(VVHIP_ERR_UNSUPPORTED, "a particle that belongs to two Drude pairs is not supported");
        in_pair[d] = in_pair[par] = 1;
        is_drude[d] = 1;
        partner[d] = par; partner[par] = d;
        if (is_nh[d] != is_nh[par] || is_ld[d] != is_ld[par])   // HOST:518-519, 786-787
            throw Error(VVHIP_ERR_TOPOLOGY, "Drude particle and its parent atom should be in the same thermostat");
        if (is_nh[d]) {
            nh_left[d] = nh_left[par] = 0;
            hp.pairs_nh.push_back(d); hp.pairs_nh.push_back(par);
            dof[TG_ATOM] -= 3;
            dof[TG_DRUDE] += 3;
        }
        if (is_ld[d]) {                                          // HOST:788-792
            ld_left[d] = ld_left[par] = 0;
            hp.pairs_ld.push_back(d); hp.pairs_ld.push_back(par);
        }
    }
    for (int i = 0; i < n; i++) {
        if (nh_left[i]) hp.normal_nh.push_back(i);               // std::set order = ascending
        if (ld_left[i]) hp.normal_ld.push_back(i);
    }
    // ---- HOST:531-541, 796-802: constraints
    for (int k = 0; k < sys.num_constraints; k++) {
        int a = sys.constraints[2 * k], b = sys.constraints[2 * k + 1];
        check_index(a, "constraint"); check_index(b, "constraint");
        if (is_nh[a] != is_nh[b] || is_ld[a] != is_ld[b])
            throw Error(VVHIP_ERR_TOPOLOGY, "Constrained particle pair should be in the same thermostat");
        if (is_nh[a]) dof[TG_ATOM] -= 1;
    }
    // ---- HOST:547-573
    if (p.use_com_temp_group) dof[TG_COM] = 3.0 * (double) hp.molecules_nh.size();
    if (sys.has_cm_motion_remover) {
        if (p.use_com_temp_group) dof[TG_COM] -= 3;
        else dof[TG_ATOM] -= 3;
    }
    for (double& d : dof) d = std::max(d, 0.0);
    int num_tg = 3;
    if (dof[TG_DRUDE] == 0) {
        num_tg = 2;
        if (dof[TG_COM] == 0) num_tg = 1;
    }
    // ---- HOST:577-594: chain masses
    vvhip_plan_info& info = hp.info;
    std::memset(&info, 0, sizeof(info));
    const double real_kbt = kBoltz * p.temperature, drude_kbt = kBoltz * p.drude_temperature;
    for (int i = 0; i < num_tg; i++) {
        const double kbt = i == TG_DRUDE ? drude_kbt : real_kbt;
        const double tg_mass = i == TG_DRUDE ? drude_kbt / std::pow(p.drude_frequency, 2) : real_kbt / std::pow(p.frequency, 2);
        info.nkbt[i] = dof[i] * kbt;
        info.eta_mass[i][0] = dof[i] * tg_mass;
        for (int c = 1; c < p.num_nh_chains; c++) info.eta_mass[i][c] = tg_mass;
    }
    for (int i = 0; i < 3; i++) info.dof[i] = dof[i];
    double mass_total = 0;                                      // HOST:1028-1031
    for (int i = 0; i < n; i++) mass_total += sys.masses[i];
    info.inv_mass_total = 1.0 / mass_total;
    info.num_particles_nh = (int) hp.particles_nh.size();
    info.num_molecules_nh = (int) hp.molecules_nh.size();
    info.num_normal_nh = (int) hp.normal_nh.size();
    info.num_pairs_nh = (int) hp.pairs_nh.size() / 2;
    info.num_normal_ld = (int) hp.normal_ld.size();
    info.num_pairs_ld = (int) hp.pairs_ld.size() / 2;
    info.num_images = sys.num_image_pairs;
    info.num_electrolyte = sys.num_electrolyte;
    info.num_temp_groups = num_tg;
    info.use_com_temp_group = p.use_com_temp_group;
    info.friction = p.friction;
    hp.has_nh = !hp.particles_nh.empty();
    hp.has_ld = sys.num_particles_ld > 0;
    hp.has_ef = sys.num_electrolyte > 0;
    hp.has_images = sys.num_image_pairs > 0;
    hp.has_pairs = npairs_all > 0;

    // ---- shard
    int sb = sys.shard_begin, se = sys.shard_end;
    if (sb == 0 && se == 0) se = n;
    if (sb < 0 || se > n || sb >= se) throw Error(VVHIP_ERR_INVALID, "bad particle shard range");
    hp.shard_begin = sb; hp.shard_end = se;
    auto in_shard = [&](int i) { return i >= sb && i < se; };

    // ---- Langevin random offsets (K/drudeLangevin.cu:20,30,51-52): normal i -> i, pair i -> Nn + 2 i
    std::vector<int32_t> rand_of(n, -1);
    for (size_t i = 0; i < hp.normal_ld.size(); i++) rand_of[hp.normal_ld[i]] = (int32_t) i;
    for (size_t i = 0; i < hp.pairs_ld.size() / 2; i++)
        rand_of[hp.pairs_ld[2 * i]] = rand_of[hp.pairs_ld[2 * i + 1]] = (int32_t) (hp.normal_ld.size() + 2 * i);

    // ---- constraint clusters for the in-kernel SHAKE (same admission rule as OpenMM's SHAKE kernel: a central particle with
    // one to three peripheral particles of identical mass and distance, every peripheral in exactly one constraint)
    struct Shake { int32_t center; std::vector<int32_t> periph; double d; bool settle = false; double d_pp = 0; };
    std::vector<Shake> shakes;
    std::vector<int32_t> shake_of(n, -1);
    bool constraints_fused = sys.num_constraints == 0;
    if (sys.num_constraints > 0 && sys.constraint_distances) {
        std::vector<int> deg(n, 0);
        for (int k = 0; k < sys.num_constraints; k++) { deg[sys.constraints[2 * k]]++; deg[sys.constraints[2 * k + 1]]++; }
        bool ok = true;
        // rigid three-site molecules first (what OpenMM hands to SETTLE): three particles, three mutual constraints and nothing
        // else; the apex is the particle whose two distances are equal and whose two partners have equal masses
        std::vector<char> used(sys.num_constraints, 0);
        {
            std::vector<std::vector<std::pair<int, int> > > adj(n);      // (other particle, constraint index)
            for (int k = 0; k < sys.num_constraints; k++) {
                const int a = sys.constraints[2 * k], b = sys.constraints[2 * k + 1];
                if (deg[a] == 2 && deg[b] == 2) { adj[a].push_back({b, k}); adj[b].push_back({a, k}); }
            }
            for (int a = 0; a < n && ok; a++) {
                if (adj[a].size() != 2 || shake_of[a] >= 0) continue;
                const int b = adj[a][0].first, c = adj[a][1].first;
                if (b == c || adj[b].size() != 2 || adj[c].size() != 2) continue;
                int kbc = -1;
                for (auto& e : adj[b]) if (e.first == c) kbc = e.second;
                if (kbc < 0) continue;                                    // a chain, not a triangle: left to the rule below (which refuses it)
                const int tri[3] = {a, b, c};
                const double dist[3] = {sys.constraint_distances[adj[a][0].second], sys.constraint_distances[adj[a][1].second], sys.constraint_distances[kbc]};   // ab, ac, bc
                // apex candidates: a (ab == ac, m_b == m_c), b (ab == bc, m_a == m_c), c (ac == bc, m_a == m_b)
                int apex = -1;
                if (dist[0] == dist[1] && sys.masses[b] == sys.masses[c]) apex = 0;
                else if (dist[0] == dist[2] && sys.masses[a] == sys.masses[c]) apex = 1;
                else if (dist[1] == dist[2] && sys.masses[a] == sys.masses[b]) apex = 2;
                if (apex < 0 || sys.masses[a] == 0 || sys.masses[b] == 0 || sys.masses[c] == 0) { ok = false; break; }
                Shake st;
                st.settle = true;
                st.center = tri[apex];
                for (int t = 0; t < 3; t++) if (t != apex) st.periph.push_back(tri[t]);
                std::sort(st.periph.begin(), st.periph.end());
                st.d = apex == 0 ? dist[0] : (apex == 1 ? dist[0] : dist[1]);
                st.d_pp = apex == 0 ? dist[2] : (apex == 1 ? dist[1] : dist[0]);
                for (int t = 0; t < 3; t++) shake_of[tri[t]] = (int32_t) shakes.size();
                shakes.push_back(st);
                used[adj[a][0].second] = used[adj[a][1].second] = used[kbc] = 1;
            }
        }
        // everything else must be hydrogen-type clusters (same admission rule as OpenMM's SHAKE kernel: a central particle with
        // one to three peripheral particles of identical mass and distance, every peripheral in exactly one constraint)
        for (int k = 0; k < sys.num_constraints && ok; k++) {
            if (used[k]) continue;
            int a = sys.constraints[2 * k], b = sys.constraints[2 * k + 1];
            const double d = sys.constraint_distances[k];
            // the central particle is the one with several constraints; for an isolated pair, the heavier one
            int ctr = deg[a] > 1 ? a : (deg[b] > 1 ? b : (sys.masses[a] >= sys.masses[b] ? a : b));
            int per = ctr == a ? b : a;
            if (deg[per] != 1 || sys.masses[ctr] == 0 || sys.masses[per] == 0 ) { ok = false; break; }
            if (shake_of[ctr] < 0) { shake_of[ctr] = (int32_t) shakes.size(); shakes.push_back(Shake{ctr, {}, d}); }
            Shake& s = shakes[shake_of[ctr]];
            if (s.settle || s.center != ctr || s.periph.size() >= 3 || s.d != d || (!s.periph.empty() && sys.masses[s.periph[0]] != sys.masses[per])) { ok = false; break; }
            if (shake_of[per] >= 0) { ok = false; break; }
            s.periph.push_back(per);
            shake_of[per] = shake_of[ctr];
        }
        if (ok) constraints_fused = true;
        else { shakes.clear(); std::fill(shake_of.begin(), shake_of.end(), -1); }
    }
    // ---- anything else (AllBonds, HAngles: chains, rings, triangles of constraints; examples/ommhelper/oplspsffile.py:948-951) becomes GENERAL
    // clusters: every connected component of the constraint graph must sit in one wave with its particles, and the wave relaxes its constraints
    // by coloured Gauss-Seidel sweeps (vv_device.inc: general_positions / general_velocities).  gc_adj = constraint partners of a particle.
    std::vector<std::vector<int32_t> > gc_adj;
    bool general = false;
    if (sys.num_constraints > 0 && sys.constraint_distances && !constraints_fused) {
        general = true;
        for (int k = 0; k < sys.num_constraints && general; k++) {
            const int a = sys.constraints[2 * k], b = sys.constraints[2 * k + 1];
            if (a == b || sys.masses[a] == 0 || sys.masses[b] == 0 || !(sys.constraint_distances[k] > 0)) general = false;      // (OpenMM refuses massless ones too)
        }
        if (general) {
            gc_adj.assign(n, {});
            for (int k = 0; k < sys.num_constraints; k++) {
                gc_adj[sys.constraints[2 * k]].push_back(sys.constraints[2 * k + 1]);
                gc_adj[sys.constraints[2 * k + 1]].push_back(sys.constraints[2 * k]);
            }
            constraints_fused = true;          // withdrawn below if a component does not fit a wave
        }
    }

    // ---- virtual sites (vvhip_system_desc.virtual_sites): kernel B places a site from the lane of one of its parents -- the first parent
    // that places no other site -- so that sites cost no lanes and the wave layout is the one the System has without them (6 000 sites on the
    // headline box as lanes of their own: 2 100 waves instead of 1 752, past what one block per CU holds, -12 % steps/s for the layout alone).
    // A site that has an image particle or sits in a Drude pair, or whose parents all place another site already, gets a lane of its own.
    // vs_host[k] = the particle whose lane places site k; vs_adj = the lane-bearing particles a particle must share a wave with for that.
    std::vector<int32_t> vs_of(n, -1), vs_host;
    std::vector<std::vector<int32_t> > vs_adj;
    bool vsites = sys.num_virtual_sites > 0 && sys.virtual_sites && sys.virtual_site_params;
    if (vsites) {
        for (int k = 0; k < sys.num_virtual_sites; k++) {
            const int32_t* rec = sys.virtual_sites + 5 * (size_t) k;
            const int site = rec[0], kind = rec[1], np = kind == VVHIP_VSITE_AVERAGE2 ? 2 : 3;
            check_index(site, "virtual site");
            if (kind < VVHIP_VSITE_AVERAGE2 || kind > VVHIP_VSITE_LOCAL_COORDS) throw Error(VVHIP_ERR_INVALID, "unknown kind of virtual site");
            if (sys.masses[site] != 0.0) throw Error(VVHIP_ERR_INVALID, "a virtual site must have mass 0");
            if (vs_of[site] >= 0) throw Error(VVHIP_ERR_INVALID, "a particle is described as a virtual site twice");
            vs_of[site] = k;
            for (int q = 0; q < np; q++) check_index(rec[2 + q], "virtual site parent");
        }
        for (int k = 0; k < sys.num_virtual_sites && vsites; k++) {     // a site that hangs on another site: left to the caller's own kernel
            const int32_t* rec = sys.virtual_sites + 5 * (size_t) k;
            for (int q = 0; q < (rec[1] == VVHIP_VSITE_AVERAGE2 ? 2 : 3); q++)
                if (vs_of[rec[2 + q]] >= 0) vsites = false;
        }
    }
    if (vsites) {
        vs_adj.assign(n, {});
        vs_host.assign((size_t) sys.num_virtual_sites, -1);
        std::vector<char> hosting(n, 0);
        auto has_lane_anyway = [&](int i) { return sys.masses[i] != 0.0 || in_pair[i] || image_of[i] >= 0; };
        for (int k = 0; k < sys.num_virtual_sites; k++) {
            const int32_t* rec = sys.virtual_sites + 5 * (size_t) k;
            const int site = rec[0], np = rec[1] == VVHIP_VSITE_AVERAGE2 ? 2 : 3;
            int host = -1;
            if (image_of[site] < 0 && !in_pair[site])
                for (int q = 0; q < np && host < 0; q++)
                    if (!hosting[rec[2 + q]] && has_lane_anyway(rec[2 + q])) host = rec[2 + q];
            if (host < 0) host = site; else hosting[host] = 1;
            vs_host[(size_t) k] = host;
            std::vector<int32_t> bearers;
            if (host == site) bearers.push_back(site);
            for (int q = 0; q < np; q++) if (has_lane_anyway(rec[2 + q])) bearers.push_back(rec[2 + q]);
            for (size_t b = 1; b < bearers.size(); b++)
                if (bearers[b] != bearers[0]) { vs_adj[bearers[0]].push_back(bearers[b]); vs_adj[bearers[b]].push_back(bearers[0]); }
        }
    }

    // ---- which particles need a lane, and which must share a wave
    auto needs_lane = [&](int i) {
        if (sys.masses[i] != 0.0) return true;      // anything massive is integrated
        if (vsites && vs_of[i] >= 0 && vs_host[(size_t) vs_of[i]] == i) return true;   // a virtual site that is placed by a lane of its own
        if (in_pair[i]) return true;                // massless Drude parent (hard-wall branch K/middle.cu:151-173)
        if (image_of[i] >= 0) return true;          // massless image parent: only the mirror copy
        return false;
    };
    std::vector<Cluster> clusters;
    std::vector<int32_t> cluster_of_mol(nmol, -1);
    std::vector<char> done(n, 0);
    for (int i = sb; i < se; i++) {
        if (done[i] || !needs_lane(i)) continue;
        const bool com = p.use_com_temp_group && is_nh[i];
        if (com) {
            int m = sys.mol_id[i];
            if (cluster_of_mol[m] < 0) {
                cluster_of_mol[m] = (int32_t) clusters.size();
                clusters.push_back(Cluster{i, {}, true});
            }
            clusters[cluster_of_mol[m]].members.push_back(i);
            done[i] = 1;
        } else {
            // closure of {Drude partner, SHAKE cluster mates}: a handful of particles at most
            Cluster c{i, {}, false};
            std::vector<int32_t> todo{i};
            done[i] = 1;
            while (!todo.empty()) {
                const int j = todo.back();
                todo.pop_back();
                c.members.push_back(j);
                auto visit = [&](int q) {
                    if (q < 0 || done[q]) return;
                    if (!in_shard(q)) throw Error(VVHIP_ERR_INVALID, "particle shard cuts a Drude pair or a constraint cluster");
                    done[q] = 1;
                    todo.push_back(q);
                };
                if (in_pair[j]) visit(partner[j]);
                if (shake_of[j] >= 0) {
                    const Shake& s = shakes[shake_of[j]];
                    visit(s.center);
                    for (int q : s.periph) visit(q);
                }
                if (general) for (int q : gc_adj[j]) visit(q);
                if (vsites) for (int q : vs_adj[j]) if (needs_lane(q)) visit(q);
            }
            std::sort(c.members.begin(), c.members.end());
            c.first = c.members[0];
            clusters.push_back(c);
        }
    }
    // a COM cluster must contain the whole molecule's thermostatted particles: check the shard did not cut it
    if (p.use_com_temp_group)
        for (int i = 0; i < n; i++)
            if (is_nh[i] && needs_lane(i) && !in_shard(i) && cluster_of_mol[sys.mol_id[i]] >= 0)
                throw Error(VVHIP_ERR_INVALID, "particle shard cuts a molecule");
    // A COM cluster larger than a wave is cut into chunks of <= 64 lanes (Drude pairs stay together); the chunks are
    // ordinary segments for the wave-level scans, the molecule-level sum goes through a small global accumulator.
    {
        std::vector<Cluster> out;
        double max_big_mass = 0;
        for (Cluster& cl : clusters) {
            if (!cl.com_segment || cl.members.size() <= 64) { out.push_back(std::move(cl)); continue; }
            std::sort(cl.members.begin(), cl.members.end());
            const int big = hp.num_big++;
            double mtot = 0;
            for (int i : cl.members) mtot += sys.masses[i];
            max_big_mass = std::max(max_big_mass, mtot);
            std::vector<char> taken(cl.members.size(), 0);
            std::vector<std::vector<int32_t> > units;          // particles that must stay in one wave (Drude pairs, constraint clusters), in index order
            auto index_in = [&](int q) {
                auto it = std::lower_bound(cl.members.begin(), cl.members.end(), q);
                if (it == cl.members.end() || *it != q) throw Error(VVHIP_ERR_UNSUPPORTED, "a Drude pair or a constraint cluster spans two molecules");
                return (size_t) (it - cl.members.begin());
            };
            for (size_t k = 0; k < cl.members.size(); k++) {
                if (taken[k]) continue;
                std::vector<int32_t> u, todo{cl.members[k]};
                taken[k] = 1;
                while (!todo.empty()) {
                    const int j = todo.back();
                    todo.pop_back();
                    u.push_back(j);
                    auto visit = [&](int q) { const size_t t = index_in(q); if (!taken[t]) { taken[t] = 1; todo.push_back(q); } };
                    if (in_pair[j]) visit(partner[j]);
                    if (shake_of[j] >= 0) { visit(shakes[shake_of[j]].center); for (int q : shakes[shake_of[j]].periph) visit(q); }
                    if (general) for (int q : gc_adj[j]) visit(q);
                    if (vsites) for (int q : vs_adj[j]) if (needs_lane(q)) visit(q);
                }
                std::sort(u.begin(), u.end());
                units.push_back(u);
            }
            Cluster chunk{units[0][0], {}, true, big, true};
            for (const auto& u : units) {
                if (chunk.members.size() + u.size() > 64) {
                    std::sort(chunk.members.begin(), chunk.members.end());
                    chunk.first = chunk.members[0];
                    out.push_back(chunk);
                    chunk = Cluster{u[0], {}, true, big, false};
                }
                chunk.members.insert(chunk.members.end(), u.begin(), u.end());
            }
            std::sort(chunk.members.begin(), chunk.members.end());
            chunk.first = chunk.members[0];
            out.push_back(chunk);
        }
        clusters.swap(out);
        if (hp.num_big > 0) {       // |sum m v| <= M |v|max with |v|max ~ 50 nm/ps; keep 4x headroom below 2^62
            const double top = std::ldexp(1.0, 62) / (std::max(max_big_mass, 1.0) * 50.0 * 4.0);
            hp.big_scale = std::ldexp(1.0, std::max(0, std::min(40, (int) std::floor(std::log2(top)))));
        }
    }
    std::stable_sort(clusters.begin(), clusters.end(), [](const Cluster& a, const Cluster& b) { return a.first < b.first; });
    int max_cluster = 0;
    for (auto& c : clusters) {
        std::sort(c.members.begin(), c.members.end());
        max_cluster = std::max(max_cluster, (int) c.members.size());
    }
    info.max_cluster = max_cluster;

    // ---- periodic layout: runs of identical repeat units, possibly repeated cell after cell (vv_host.hpp: PeriodicLayout)
    struct Region { uint64_t sig; int p, reps, atoms, segs; };      // unit of p clusters repeated reps times
    std::vector<Region> regions;
    PeriodicLayout per;
    int per_R = 0, per_clusters_per_cell = 0;
    std::vector<int> reg_cluster_start;                              // first cluster (cell-local) of region r
    auto try_periodic = [&]() -> bool {
        // Measured on MI355X (same box, alternating runs): with 8.9 M particles kernel B 305 -> 277 us.  At 111 k particles the block's copy of
        // the pattern rows and its barrier cost more than the slot load did (kernel B 6.0 -> 6.4 us) and best-fit packing needs 13 % fewer
        // waves.  In between (round 4, tools/probes/layout_crossover.py, each layout with its own best launch shape, steps/s best-fit |
        // arithmetic): 222 k particles 66.0 | 57.7 k, 444 k 38.5 | 35.5 k, 666 k 26.8 | 25.0 k, 888 k 18.75 | 18.91 k, 1.33 M 13.6 | 14.1 k,
        // 2.7 M 7.1 | 7.5 k, 4.4 M 4.12 | 4.63 k.  (Round 2 had put the switch at 0.2 M lanes, +3 .. +7 % then: two blocks of 6-7 tile waves
        // per CU, vv_api.cpp: pick_launch_shape, took the best-fit layout past it.)  With the chain kept inside kernel B up to 2.6 M particles
        // (vv_api.cpp: split_chain_waves) the two meet a little higher: 888 k particles 20.8 | 20.6 k, 1.33 M 14.3 | 14.5 k, 1.78 M 11.2 | 11.4 k.
        // So: from 1.1 M lanes, unless VVHIP_PERIODIC=1 / 0 says always / never.
        size_t lanes = 0;
        for (const Cluster& c : clusters) lanes += c.members.size();
        bool want = lanes >= 1100000;
        if (const char* e = std::getenv("VVHIP_PERIODIC")) want = std::atoi(e) != 0;
        if (!want) return false;
        if (hp.has_ld || hp.has_images || hp.num_big > 0 || clusters.empty() || general || vsites) return false;
        const size_t K = clusters.size();
        std::vector<uint64_t> sig(K);
        int expect = sb;
        for (size_t k = 0; k < K; k++) {
            const Cluster& c = clusters[k];
            if (c.first != expect || c.members.size() > 64) return false;                 // particles must be consecutive, cluster after cluster
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](uint64_t v) { for (int b = 0; b < 8; b++) { h ^= (v >> (8 * b)) & 0xff; h *= 1099511628211ull; } };
            mix(c.members.size()); mix(c.com_segment ? 1 : 0);
            for (size_t j = 0; j < c.members.size(); j++) {
                const int i = c.members[j];
                if (i != c.first + (int) j) return false;
                uint64_t mb; std::memcpy(&mb, &sys.masses[i], 8);
                mix(mb);
                mix((uint64_t) (is_nh[i] ? 1 : 0) | (in_pair[i] ? 2 : 0) | (is_drude[i] ? 4 : 0) | (is_el[i] ? 8 : 0));
                mix(in_pair[i] ? (uint64_t) (int64_t) (partner[i] - c.first) : 0);
                if (shake_of[i] >= 0) {                  // in-kernel constraint cluster: same shape, distances and position inside the unit
                    const Shake& sh = shakes[shake_of[i]];
                    uint64_t db, pb; std::memcpy(&db, &sh.d, 8); std::memcpy(&pb, &sh.d_pp, 8);
                    int role = 0;
                    for (size_t q = 0; q < sh.periph.size(); q++) if (sh.periph[q] == i) role = 1 + (int) q;
                    mix(1 + (uint64_t) role); mix((uint64_t) (int64_t) (sh.center - c.first)); mix(sh.periph.size()); mix(db); mix(pb); mix(sh.settle ? 1 : 0);
                } else {
                    mix(0);
                }
            }
            sig[k] = h;
            expect += (int) c.members.size();
        }
        if (expect != se) return false;
        // greedy decomposition into regions: at each position the unit length (clusters adding up to <= 64 lanes) that covers the most clusters
        regions.clear();
        for (size_t i = 0; i < K;) {
            int best_p = 1, best_reps = 1;
            size_t best_cover = 0;
            for (int pp = 1; pp <= 64 && i + pp <= K; pp++) {
                int atoms = 0;
                for (int j = 0; j < pp; j++) atoms += (int) clusters[i + j].members.size();
                if (atoms > 64) break;
                size_t j = i + pp;
                while (j < K && sig[j] == sig[j - pp]) j++;
                const int reps = (int) ((j - i) / pp);
                if (reps < 2 && pp > 1) continue;         // a unit must repeat: 13 anions followed by 4 waters are not one 17-cluster unit
                if ((size_t) reps * pp > best_cover) { best_cover = (size_t) reps * pp; best_p = pp; best_reps = reps; }
            }
            Region r{0, best_p, best_reps, 0, 0};
            uint64_t h = 1469598103934665603ull;
            for (int j = 0; j < best_p; j++) {
                h = (h ^ sig[i + j]) * 1099511628211ull;
                r.atoms += (int) clusters[i + j].members.size();
                r.segs += clusters[i + j].com_segment ? 1 : 0;
            }
            r.sig = h;
            regions.push_back(r);
            i += (size_t) best_p * best_reps;
            if (regions.size() > ((size_t) 1 << 22)) return false;
        }
        if (std::getenv("VVHIP_PERIODIC_DEBUG"))
            for (size_t r = 0; r < regions.size() && r < 16; r++) std::fprintf(stderr, "  region %zu: unit of %d cluster(s), %d atoms, x%d\n", r, regions[r].p, regions[r].atoms, regions[r].reps);
        // period of the region list
        const int L = (int) regions.size();
        int R = 0;
        for (int cand = 1; cand <= 4 && cand <= L && !R; cand++) {
            if (L % cand) continue;
            bool ok = true;
            for (int i = cand; i < L && ok; i++)
                ok = regions[i].sig == regions[i - cand].sig && regions[i].p == regions[i - cand].p && regions[i].reps == regions[i - cand].reps;
            if (ok) R = cand;
        }
        if (!R) return false;
        per = PeriodicLayout{};
        per.nreg = R;
        per.ncells = L / R;
        reg_cluster_start.assign(R + 1, 0);
        int w = 0, at = 0, sg = 0, cl = 0;
        for (int r = 0; r < R; r++) {
            const Region& g = regions[r];
            const int m = 64 / g.atoms;
            per.reg_wave_start[r] = w; per.reg_atom_start[r] = at; per.reg_seg_start[r] = sg;
            per.reg_P[r] = m * g.atoms; per.reg_spw[r] = m * g.segs;
            reg_cluster_start[r] = cl;
            w += (g.reps + m - 1) / m; at += g.atoms * g.reps; sg += g.segs * g.reps; cl += g.p * g.reps;
            per.reg_atom_end[r] = at;
        }
        reg_cluster_start[R] = cl;
        per.wpc = w; per.apc = at; per.spc = sg;
        per_R = R; per_clusters_per_cell = cl;
        const long total_waves = (long) per.wpc * per.ncells;
        if (total_waves > (1l << 26)) return false;
        per.magic = per.ncells > 1 ? (uint32_t) ((1ull << 32) / (uint64_t) per.wpc + 1) : 0u;
        return true;
    };
    // (wave, first lane) of cluster k under the periodic layout
    auto periodic_place = [&](size_t k, int& wave_out, int& lane_out) {
        const int cell = (int) (k / (size_t) per_clusters_per_cell), kl = (int) (k % (size_t) per_clusters_per_cell);
        int r = 0;
        while (r + 1 < per_R && kl >= reg_cluster_start[r + 1]) r++;
        const Region& g = regions[r];
        const int ku = kl - reg_cluster_start[r], unit = ku / g.p, j = ku % g.p, m = 64 / g.atoms;
        int off = 0;
        for (int t = 0; t < j; t++) off += (int) clusters[k - j + t].members.size();
        wave_out = cell * per.wpc + per.reg_wave_start[r] + unit / m;
        lane_out = (unit % m) * g.atoms + off;
    };

    // ---- wave packing: clusters are visited in particle order and placed best-fit -- into the open wave whose free lanes
    // they fill most tightly, else into a new wave.  Visiting in order keeps neighbouring molecules in neighbouring waves
    // (coalescing, L2 locality); best-fit lets e.g. a 10-particle anion complete a wave that two 27-particle cations left at 54
    // lanes (C3: 87 % -> 99 % lane use, 13 % fewer waves).  O(64) per cluster via free-lane buckets.  With a periodic layout the
    // position of every cluster is given by the formula instead.
    std::vector<int32_t>& slots = hp.slots;
    std::vector<int32_t> lane_of, wave_of;
    int nwaves = 0;
    auto fill_slots = [&](bool periodic) {
    slots.clear();
    hp.seg_mass.clear();
    int lane = 0, wave = -1, num_waves_alloc = 0;
    std::vector<int32_t> fill;                          // lanes used per wave
    std::vector<std::vector<int32_t> > open_by_free(65);  // waves with exactly f free lanes (stacks)
    auto new_wave = [&]() {
        const int w = num_waves_alloc++;
        fill.push_back(0);
        slots.resize((size_t) (w + 1) * 128);
        for (int l = 0; l < 64; l++) { slots[(size_t) w * 128 + 2 * l] = -1; slots[(size_t) w * 128 + 2 * l + 1] = 0; }
        return w;
    };
    lane_of.assign(n, -1); wave_of.assign(n, -1);
    if (periodic) for (long w = 0; w < (long) per.wpc * per.ncells; w++) new_wave();
    int used = 0;
    // Measured on MI355X: best-fit wins while the working set is cache resident (111 k particles: +3 %, 0.9 M: +4 %) and loses
    // once the kernels are HBM bound (8.9 M: -5 %, segments from distant index ranges cost partial cache lines), so very large
    // systems keep plain in-order filling.
    size_t total_lanes = 0;
    for (const Cluster& c : clusters) total_lanes += c.members.size();
    const bool best_fit = total_lanes < ((size_t) 1 << 20);
    int last_wave = -1;
    size_t cluster_index = 0;
    // General constraint clusters: a wave's constraint list lives one constraint per LANE, so a wave holds 64 of them whatever its lanes
    // hold -- the packer counts them too (round-4 advisor: HAngles on hydrocarbon-rich molecules, ~1.5 constraints per particle, filled the
    // lanes of a wave with molecules whose constraints did not fit its list, and the whole System lost the fused path).
    std::vector<int32_t> cons_fill;                     // general constraints per wave
    auto cons_of = [&](const Cluster& c) {
        if (!general) return 0;
        size_t deg = 0;
        for (int i : c.members) deg += gc_adj[i].size();
        return (int) (deg / 2);
    };
    for (const Cluster& c : clusters) {
        const int sz = (int) c.members.size(), nc = cons_of(c);
        wave = -1;
        if (periodic) {
            periodic_place(cluster_index, wave, lane);
            fill[wave] = lane;
        } else if (best_fit) {
            for (int f = sz; f <= 63 && wave < 0; f++) {
                auto& bucket = open_by_free[f];
                for (size_t t = bucket.size(); t-- > 0 && wave < 0;)
                    if (cons_fill[bucket[t]] + nc <= 64) { wave = bucket[t]; bucket.erase(bucket.begin() + (long) t); }
            }
        } else if (last_wave >= 0 && fill[last_wave] + sz <= 64 && cons_fill[last_wave] + nc <= 64) {
            wave = last_wave;
        }
        cluster_index++;
        if (wave < 0) wave = new_wave();
        if ((size_t) wave >= cons_fill.size()) cons_fill.resize((size_t) wave + 1, 0);
        cons_fill[wave] += nc;
        last_wave = wave;
        lane = fill[wave];
        fill[wave] += sz;
        if (!periodic && best_fit && fill[wave] < 64) open_by_free[64 - fill[wave]].push_back(wave);
        const int first_lane = lane, last_lane = lane + sz - 1;
        for (int k = 0; k < sz; k++) { lane_of[c.members[k]] = lane + k; wave_of[c.members[k]] = wave; }
        if (c.com_segment) {
            double m = 0;
            for (int k = 0; k < sz; k++) { const int i = c.members[k]; if (is_nh[i] && sys.masses[i] != 0.0) m += sys.masses[i]; }
            if (hp.seg_mass.size() < (size_t) (wave + 1) * 128) hp.seg_mass.resize((size_t) (wave + 1) * 128, 0.0);
            hp.seg_mass[((size_t) wave * 64 + first_lane) * 2] = m;
            hp.seg_mass[((size_t) wave * 64 + first_lane) * 2 + 1] = m != 0 ? 1.0 / m : 0.0;
        }
        for (int k = 0; k < sz; k++, lane++) {
            const int i = c.members[k];
            uint32_t role;
            const bool massive = sys.masses[i] != 0.0;
            if (is_nh[i] && (massive || in_pair[i]))
                role = in_pair[i] ? (is_drude[i] ? ROLE_NH_DRUDE : ROLE_NH_PARENT) : ROLE_NH_NORMAL;
            else if (is_ld[i] && (massive || in_pair[i]))
                role = in_pair[i] ? (is_drude[i] ? ROLE_LD_DRUDE : ROLE_LD_PARENT) : ROLE_LD_NORMAL;
            else
                role = massive ? ROLE_PLAIN : ROLE_NONE;
            uint32_t meta = role;
            const int pl = in_pair[i] ? lane_of[partner[i]] >= 0 ? lane_of[partner[i]] : lane : lane;
            meta |= (uint32_t) pl << META_PARTNER_SHIFT;
            const int sf = c.com_segment ? first_lane : lane, sl = c.com_segment ? last_lane : lane;
            meta |= (uint32_t) sf << META_SEGFIRST_SHIFT;
            meta |= (uint32_t) sl << META_SEGLAST_SHIFT;
            if (is_el[i]) meta |= META_EFIELD;
            if (image_of[i] >= 0) meta |= META_HAS_IMAGE;
            if (c.com_segment && k == sz - 1) meta |= META_COM_LEADER;      // the LAST lane: its wave-prefix value already ends the segment (kernel A's KE stage)
            if (in_pair[i]) meta |= META_PAIR;
            if (is_drude[i]) meta |= META_IS_DRUDE;
            if (massive) meta |= META_MASSIVE;
            if (shake_of[i] >= 0 || (general && !gc_adj[i].empty())) meta |= META_SHAKE;
            if (c.big >= 0) meta |= META_BIGMOL;
            if (c.big >= 0 && c.big_first && (meta & META_COM_LEADER)) meta |= META_BIG_FIRST;
            slots[(size_t) wave * 128 + 2 * lane] = i - sb;
            slots[(size_t) wave * 128 + 2 * lane + 1] = (int32_t) meta;
            used++;
        }
    }
    // partner lanes were only known for the earlier member of each pair: fix them up
    nwaves = std::max(num_waves_alloc, 1);
    if (num_waves_alloc == 0) new_wave();
    for (int w = 0; w < nwaves; w++)
        for (int l = 0; l < 64; l++) {
            int32_t a = slots[(size_t) w * 128 + 2 * l];
            if (a < 0) continue;
            const int i = a + sb;
            if (in_pair[i]) {
                if (wave_of[partner[i]] != w)
                    throw Error(VVHIP_ERR_UNSUPPORTED, "a Drude particle and its parent are in different molecules");
                uint32_t meta = (uint32_t) slots[(size_t) w * 128 + 2 * l + 1];
                meta &= ~(0x3Fu << META_PARTNER_SHIFT);
                meta |= (uint32_t) lane_of[partner[i]] << META_PARTNER_SHIFT;
                slots[(size_t) w * 128 + 2 * l + 1] = (int32_t) meta;
            }
        }
    info.num_waves = nwaves;
    info.num_slots_used = used;
    hp.seg_mass.resize((size_t) nwaves * 128, 0.0);
    {   // compact the per-segment table: segments numbered wave by wave, inside a wave by the lane of their leader (what the kernels
        // recompute as seg_base[wave] + popcount of the leader lanes below: one 32-byte COM velocity per segment, stored back to back,
        // instead of one at every segment's first-lane position of a 64-entry page per wave)
        std::vector<double> dense;
        hp.seg_base.assign((size_t) nwaves + 1, 0);
        for (int w = 0; w < nwaves; w++) {
            hp.seg_base[w] = (int32_t) (dense.size() / 2);
            for (int l = 0; l < 64; l++) {
                if (slots[(size_t) w * 128 + 2 * l] < 0) continue;
                const uint32_t meta = (uint32_t) slots[(size_t) w * 128 + 2 * l + 1];
                if (!(meta & META_COM_LEADER)) continue;
                const int first = (meta >> META_SEGFIRST_SHIFT) & 63;
                dense.push_back(hp.seg_mass[((size_t) w * 64 + first) * 2]);
                dense.push_back(hp.seg_mass[((size_t) w * 64 + first) * 2 + 1]);
            }
        }
        hp.seg_base[nwaves] = (int32_t) (dense.size() / 2);
        // one spare entry: the kernels load table[seg_base[wave] + leaders below the lane] unconditionally, and the lanes behind a
        // wave's last leader point one entry past the wave's segments (the device COM tables are sized from this one)
        dense.push_back(0.0); dense.push_back(0.0);
        hp.seg_mass.swap(dense);
    }

    };   // fill_slots

    // Does the formula reproduce the table?  Particle index, role word (against the pattern wave: the region's first wave of cell 0)
    // and segment mass (against the pattern wave's segments) of every lane.
    auto periodic_matches = [&]() -> bool {
        if (nwaves != per.wpc * per.ncells) return false;
        for (int w = 0; w < nwaves; w++) {
            const int cell = per.ncells > 1 ? (int) (((uint64_t) (uint32_t) w * per.magic) >> 32) : 0;
            if (cell != w / per.wpc) return false;
            const int wl = w - cell * per.wpc;
            int r = 0;
            for (int k = 1; k < 4; k++) if (wl >= per.reg_wave_start[k]) r = k;
            const int wr = wl - per.reg_wave_start[r], a0 = per.reg_atom_start[r] + wr * per.reg_P[r];
            const int count = std::min(per.reg_P[r], per.reg_atom_end[r] - a0);
            const int pw = per.reg_wave_start[r];
            int ord = 0;
            for (int l = 0; l < 64; l++) {
                const int32_t at = slots[(size_t) w * 128 + 2 * l];
                const uint32_t meta = (uint32_t) slots[(size_t) w * 128 + 2 * l + 1];
                if (l >= count) { if (at >= 0) return false; continue; }
                if (at != cell * per.apc + a0 + l) return false;
                if (meta != (uint32_t) slots[(size_t) pw * 128 + 2 * l + 1]) return false;
                if (sys.masses[at + sb] != sys.masses[slots[(size_t) pw * 128 + 2 * l] + sb]) return false;      // (the per-lane mass tables are pattern rows too)
                if (meta & META_COM_LEADER) {
                    const size_t here = (size_t) hp.seg_base[w] + ord, pat = (size_t) per.reg_seg_start[r] + ord;
                    if ((int) here != cell * per.spc + per.reg_seg_start[r] + wr * per.reg_spw[r] + ord) return false;
                    if (std::memcmp(&hp.seg_mass[2 * here], &hp.seg_mass[2 * pat], 16) != 0) return false;
                    ord++;
                }
            }
        }
        return true;
    };
    bool periodic = try_periodic();
    if (std::getenv("VVHIP_PERIODIC_DEBUG")) std::fprintf(stderr, "periodic: try=%d regions=%zu R=%d cells=%d wpc=%d apc=%d\n", (int) periodic, regions.size(), per_R, per.ncells, per.wpc, per.apc);
    fill_slots(periodic);
    if (periodic && !periodic_matches()) { if (std::getenv("VVHIP_PERIODIC_DEBUG")) std::fprintf(stderr, "periodic: formula does not reproduce the table\n"); periodic = false; fill_slots(false); }
    if (periodic) { per.enabled = 1; hp.per = per; }
    info.periodic_layout = periodic ? 1 : 0;

    if (hp.has_images) {
        hp.slot_image.assign((size_t) nwaves * 64, -1);
        for (int k = 0; k < sys.num_image_pairs; k++) {
            int img = sys.image_pairs[2 * k], par = sys.image_pairs[2 * k + 1];
            if (!in_shard(par)) continue;
            if (!in_shard(img)) throw Error(VVHIP_ERR_UNSUPPORTED, "image particle outside its parent's shard (image charges are single-GPU)");
            hp.image_pairs.push_back(img - sb);
            hp.image_pairs.push_back(par - sb);
        }
    }
    if (hp.has_ld) hp.slot_rand.assign((size_t) nwaves * 64, -1);
    info.constraints_fused = constraints_fused ? 1 : 0;
    info.num_shake_clusters = 0;
    info.num_settle_clusters = 0;
    if (!shakes.empty()) {
        hp.slot_shake.assign((size_t) nwaves * 64, 0);
        hp.slot_shake_param.assign((size_t) nwaves * 64 * 4, 0.0f);
        for (const Shake& s : shakes) {
            if (!in_shard(s.center)) continue;
            const int w = wave_of[s.center], lc = lane_of[s.center];
            // every member's word lists the whole cluster (vv_host.hpp: SHAKE_WORD_*), every member carries the parameters
            uint32_t common = ((uint32_t) s.periph.size() << 2) | ((uint32_t) lc << vv::SHAKE_WORD_CENTRAL_SHIFT) | (s.settle ? vv::SHAKE_WORD_SETTLE : 0u);
            for (size_t k = 0; k < 3; k++) {
                int lane = lc;
                if (k < s.periph.size()) {
                    const int q = s.periph[k];
                    if (wave_of[q] != w) throw Error(VVHIP_ERR_UNSUPPORTED, "a constraint cluster does not fit into one wave with its molecule");
                    lane = lane_of[q];
                }
                common |= (uint32_t) lane << (4 + 6 * k);
            }
            const double imc = 1.0 / sys.masses[s.center], imp = 1.0 / sys.masses[s.periph[0]];
            float prm[4];
            if (s.settle) { prm[0] = (float) s.d; prm[1] = (float) s.d_pp; prm[2] = 0; prm[3] = 0; info.num_settle_clusters++; }   // apex-partner and partner-partner distance
            else { prm[0] = (float) imc; prm[1] = (float) (0.5 / (imc + imp)); prm[2] = (float) (s.d * s.d); prm[3] = (float) imp; info.num_shake_clusters++; }
            hp.slot_shake[(size_t) w * 64 + lc] = (int32_t) (common | 1u);
            std::memcpy(&hp.slot_shake_param[((size_t) w * 64 + lc) * 4], prm, sizeof(prm));
            for (size_t k = 0; k < s.periph.size(); k++) {
                const int lq = lane_of[s.periph[k]];
                hp.slot_shake[(size_t) w * 64 + lq] = (int32_t) (common | 2u | ((uint32_t) k << vv::SHAKE_WORD_OWN_SHIFT));
                std::memcpy(&hp.slot_shake_param[((size_t) w * 64 + lq) * 4], prm, sizeof(prm));
            }
        }
    }
    // Virtual sites: word of the placing lane = lanes of the site's parents | kind (| hosted: the lane belongs to a parent and stores the
    // site by its particle index, vsite_atom), record = row of vsite_params / vsite_atom.  A site whose parents sit in another wave (another
    // molecule, a molecule cut into chunks) or have no lane at all: nothing is placed in-kernel, the caller keeps running its own
    // computeVirtualSites.
    info.num_virtual_sites = 0;
    if (vsites) {
        std::vector<int32_t> table((size_t) nwaves * 128, 0), atoms;
        std::vector<double> prm;
        bool fits = true;
        int count = 0;
        for (int k = 0; k < sys.num_virtual_sites && fits; k++) {
            const int32_t* rec = sys.virtual_sites + 5 * (size_t) k;
            const int site = rec[0], kind = rec[1], np = kind == VVHIP_VSITE_AVERAGE2 ? 2 : 3, host = vs_host[(size_t) k];
            bool any = in_shard(site), all = in_shard(site);
            for (int q = 0; q < np; q++) { any = any || in_shard(rec[2 + q]); all = all && in_shard(rec[2 + q]); }
            if (!any) continue;
            // (a shard that holds a site but not its parents, or the reverse, could place it nowhere and no other rank would: refuse it)
            if (!all) throw Error(VVHIP_ERR_INVALID, "particle shard cuts a virtual site from its parents");
            const int w = wave_of[host];
            uint32_t word = vv::VS_WORD_VALID | ((uint32_t) kind << 18) | (host != site ? vv::VS_WORD_HOSTED : 0u);
            for (int q = 0; q < 3 && w >= 0; q++) {
                const int par = rec[2 + (q < np ? q : 0)];
                if (!in_shard(par) || wave_of[par] != w) { fits = false; break; }
                word |= (uint32_t) lane_of[par] << (6 * q);
            }
            if (w < 0 || !fits) { fits = false; break; }
            table[((size_t) w * 64 + lane_of[host]) * 2] = (int32_t) word;
            table[((size_t) w * 64 + lane_of[host]) * 2 + 1] = count++;
            atoms.push_back(site - sb);
            prm.insert(prm.end(), sys.virtual_site_params + 12 * (size_t) k, sys.virtual_site_params + 12 * (size_t) k + 12);
        }
        if (fits && count > 0) {
            hp.slot_vsite.swap(table);
            hp.vsite_params.swap(prm);
            hp.vsite_atom.swap(atoms);
            info.num_virtual_sites = count;
        }
    }
    info.num_general_constraints = 0;
    info.general_relaxation = 0.0;
    hp.gc_colors = 0;
    if (general) {
        // relaxation factor of the sweeps (vv_layout.h: GC_OMEGA_*): by whether any three constraints close a triangle
        bool triangles = false;
        for (int k = 0; k < sys.num_constraints && !triangles; k++) {
            const int a = sys.constraints[2 * k], b = sys.constraints[2 * k + 1];
            for (int q : gc_adj[a])
                if (q != b && std::find(gc_adj[b].begin(), gc_adj[b].end(), q) != gc_adj[b].end()) { triangles = true; break; }
        }
        hp.gc_omega = triangles ? vv::GC_OMEGA_TRIANGLES : vv::GC_OMEGA_PLAIN;
        // Per wave: its constraints, coloured greedily in System order (two constraints that share a particle get different colours: a
        // colour's constraints are relaxed side by side, one per lane), sorted by colour and handed to lanes 0, 1, ... of the wave.
        // Word: bit 31 valid | colour << 12 | lane of b << 6 | lane of a; parameters: d^2, 0.5 / (1/m_a + 1/m_b), 1/m_a, 1/m_b (float, as
        // OpenMM keeps its SHAKE parameters).  A wave with more than 64 constraints, more than 16 colours, or a constraint across two waves
        // (a molecule cut into chunks): not fused, the split entry points take over as before.
        struct GC { int la, lb, colour; float prm[4]; };
        std::vector<std::vector<GC> > per_wave((size_t) nwaves);
        std::vector<uint32_t> used_colours((size_t) nwaves * 64, 0u);
        bool fits = true;
        for (int k = 0; k < sys.num_constraints && fits; k++) {
            const int a = sys.constraints[2 * k], b = sys.constraints[2 * k + 1];
            if (!in_shard(a) && !in_shard(b)) continue;
            if (!in_shard(a) || !in_shard(b)) throw Error(VVHIP_ERR_INVALID, "particle shard cuts a Drude pair or a constraint cluster");
            const int w = wave_of[a];
            if (w < 0 || wave_of[b] != w) { fits = false; hp.unfused_reason = "a constraint connects particles of two waves (a molecule larger than a wave, cut into chunks)"; break; }
            const int la = lane_of[a], lb = lane_of[b];
            const uint32_t taken = used_colours[(size_t) w * 64 + la] | used_colours[(size_t) w * 64 + lb];
            int colour = 0;
            while (colour < 16 && ((taken >> colour) & 1u)) colour++;
            if (colour >= 16) { fits = false; hp.unfused_reason = "a particle takes part in more constraints than the 16 colours of a wave's sweeps can separate"; break; }
            used_colours[(size_t) w * 64 + la] |= 1u << colour;
            used_colours[(size_t) w * 64 + lb] |= 1u << colour;
            const double ima = 1.0 / sys.masses[a], imb = 1.0 / sys.masses[b], d = sys.constraint_distances[k];
            GC g{la, lb, colour, {(float) (d * d), (float) (0.5 / (ima + imb)), (float) ima, (float) imb}};
            per_wave[(size_t) w].push_back(g);
            if (per_wave[(size_t) w].size() > 64) { fits = false; hp.unfused_reason = "a connected component of the constraint graph holds more than the 64 constraints of a wave's list"; }
            hp.gc_colors = std::max(hp.gc_colors, colour + 1);
        }
        if (!fits) {
            constraints_fused = false;
            info.constraints_fused = 0;
            hp.gc_colors = 0;
        } else {
            info.general_relaxation = hp.gc_omega;
            hp.slot_shake.assign((size_t) nwaves * 64, 0);
            hp.slot_shake_param.assign((size_t) nwaves * 64 * 4, 0.0f);
            for (int w = 0; w < nwaves; w++) {
                auto& list = per_wave[(size_t) w];
                std::stable_sort(list.begin(), list.end(), [](const GC& x, const GC& y) { return x.colour < y.colour; });
                for (size_t l = 0; l < list.size(); l++) {
                    const GC& g = list[l];
                    hp.slot_shake[(size_t) w * 64 + l] = (int32_t) (vv::GC_WORD_VALID | ((uint32_t) g.colour << 12) | ((uint32_t) g.lb << 6) | (uint32_t) g.la);
                    std::memcpy(&hp.slot_shake_param[((size_t) w * 64 + l) * 4], g.prm, sizeof(g.prm));
                    info.num_general_constraints++;
                }
            }
        }
    }
    if (hp.per.enabled && !shakes.empty()) {
        // the constraint words and parameters of every wave must be those of its region's pattern wave too
        const PeriodicLayout& q = hp.per;
        bool same = true;
        for (int w = 0; w < nwaves && same; w++) {
            const int wl = w % q.wpc;
            int r = 0;
            for (int k = 1; k < 4; k++) if (wl >= q.reg_wave_start[k]) r = k;
            const int pw = q.reg_wave_start[r];
            for (int l = 0; l < 64 && same; l++) {
                if (slots[(size_t) w * 128 + 2 * l] < 0) continue;
                same = hp.slot_shake[(size_t) w * 64 + l] == hp.slot_shake[(size_t) pw * 64 + l] &&
                       std::memcmp(&hp.slot_shake_param[((size_t) w * 64 + l) * 4], &hp.slot_shake_param[((size_t) pw * 64 + l) * 4], 16) == 0;
            }
        }
        if (!same) { hp.per.enabled = 0; info.periodic_layout = 0; hp.info.periodic_layout = 0; }      // the layout stays, the kernels load their slot words
    }
    if (hp.num_big > 0) {
        hp.slot_big.assign((size_t) nwaves * 64, -1);
        std::vector<int32_t> big_of(n, -1);
        for (const Cluster& cl : clusters)
            if (cl.big >= 0) for (int i : cl.members) big_of[i] = cl.big;
        for (int w = 0; w < nwaves; w++)
            for (int l = 0; l < 64; l++) {
                const int32_t at = slots[(size_t) w * 128 + 2 * l];
                if (at >= 0) hp.slot_big[(size_t) w * 64 + l] = big_of[at + sb];
            }
    }
    if (hp.has_images || hp.has_ld)
        for (int w = 0; w < nwaves; w++)
            for (int l = 0; l < 64; l++) {
                int32_t a = slots[(size_t) w * 128 + 2 * l];
                if (a < 0) continue;
                if (hp.has_images && image_of[a + sb] >= 0) hp.slot_image[(size_t) w * 64 + l] = image_of[a + sb] - sb;
                if (hp.has_ld) hp.slot_rand[(size_t) w * 64 + l] = rand_of[a + sb];
            }
    return hp;
}

}  // namespace vv
