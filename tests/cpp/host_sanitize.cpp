// tests/cpp/host_sanitize.cpp -- the host analysis (csrc/vv_host.cpp: tables, wave layout, arithmetic layout) under AddressSanitizer and
// UndefinedBehaviorSanitizer on the CPU (tests/test_host_sanitizers.py builds and runs it with g++ -fsanitize=address,undefined; GPU
// sanitizers are not available on this pool).  Random inventories of repeated molecules -- Drude pairs, hydrogen constraints, rigid water,
// Langevin / image / electrolyte subsets, general constraint clusters, virtual sites, defects, shard cuts on molecule boundaries, molecules
// larger than a wave -- analysed with the
// arithmetic layout forced on and off; a few invariants are checked so that the optimiser cannot drop the work.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../openmm-velocityverlet_amd/csrc/vv_host.hpp"

namespace {
struct Sys {
    std::vector<double> masses, cdist;
    std::vector<int32_t> mol, pairs, cons, ld, img, el, vs;
    std::vector<double> vsp;
    int nmol = 0;
};

Sys build(std::mt19937& rng, int flavour) {
    auto U = [&](int lo, int hi) { return std::uniform_int_distribution<int>(lo, hi)(rng); };
    Sys s;
    struct Tmpl { std::vector<int> units; double heavy; };      // 0 heavy, 1 hydrogen (constrained to the last heavy), 2 Drude pair
    std::vector<Tmpl> tmpl;
    const int ntmpl = U(1, 3);
    for (int t = 0; t < ntmpl; t++) {
        Tmpl m;
        static const int sizes[] = {1, 2, 3, 4, 7, 10, 19, 27, 33, 64, 70, 130};
        int size = sizes[U(0, flavour == 3 ? 11 : 9)], k = 0;
        while (k < size) {
            const int kind = U(0, 2);
            if (kind == 2 && k + 2 <= size) { m.units.push_back(2); k += 2; }
            else if (kind == 1 && k > 0) { m.units.push_back(1); k += 1; }
            else { m.units.push_back(0); k += 1; }
        }
        m.heavy = 12.0 + U(0, 3);
        tmpl.push_back(m);
    }
    if (flavour == 2) { tmpl.clear(); tmpl.push_back(Tmpl{{0, 1, 1}, 15.9994}); }      // water: rigid triangles below
    const int cells = U(1, 4);
    std::vector<int> counts;
    for (size_t t = 0; t < tmpl.size(); t++) counts.push_back(U(1, 20));
    const int defect_cell = U(0, cells + 1), defect_t = U(0, (int) tmpl.size() - 1);
    auto add = [&](const Tmpl& m) {
        int last_heavy = -1, hcount = 0;
        const int first = (int) s.masses.size();
        for (int u : m.units) {
            if (u == 2) {
                s.masses.push_back(m.heavy - 0.4); s.masses.push_back(0.4);
                const int n = (int) s.masses.size();
                s.pairs.push_back(n - 1); s.pairs.push_back(n - 2);
                s.mol.push_back(s.nmol); s.mol.push_back(s.nmol);
                last_heavy = n - 2; hcount = 0;
            } else if (u == 1) {
                s.masses.push_back(1.008); s.mol.push_back(s.nmol);
                const int h = (int) s.masses.size() - 1;
                if (flavour >= 1 && flavour != 6 && last_heavy >= 0 && hcount < 3) { s.cons.push_back(h); s.cons.push_back(last_heavy); s.cdist.push_back(flavour == 2 ? 0.1 : 0.109); hcount++; }
            } else {
                s.masses.push_back(m.heavy); s.mol.push_back(s.nmol);
                const int h = (int) s.masses.size() - 1;
                if (flavour == 5 && last_heavy >= first) { s.cons.push_back(h); s.cons.push_back(last_heavy); s.cdist.push_back(0.15); }      // heavy-heavy bonds: general clusters
                last_heavy = h; hcount = 0;
            }
        }
        if (flavour == 2) { s.cons.push_back(first + 1); s.cons.push_back(first + 2); s.cdist.push_back(0.1633); }      // H-H: a rigid triangle
        s.nmol++;
    };
    for (int c = 0; c < cells; c++)
        for (size_t t = 0; t < tmpl.size(); t++)
            for (int j = 0; j < counts[t]; j++) {
                add(tmpl[t]);
                if (c == defect_cell && (int) t == defect_t && j == counts[t] / 2) add(Tmpl{{0, 0, 2}, 15.5});
            }
    if (flavour == 4) {      // electrode machinery: a few Langevin atoms in front is not possible after the fact, so append them; images of the first molecule
        const int n0 = (int) s.masses.size();
        for (int w = 0; w < 5; w++) { s.masses.push_back(32.06); s.mol.push_back(s.nmol++); s.ld.push_back(n0 + w); }
        int m0 = 0;
        for (int i = 0; i < n0 && s.mol[i] == 0; i++) m0++;
        for (int i = 0; i < m0; i++) { s.masses.push_back(0.0); s.mol.push_back(0); s.img.push_back((int) s.masses.size() - 1); s.img.push_back(i); s.el.push_back(i); }
    }
    if (flavour == 6) {      // virtual sites behind the last particle, each in the molecule of its first parent; now and then a parent from the next molecule
        const int n0 = (int) s.masses.size();
        for (int i = 0; i + 3 < n0; i += U(1, 9)) {
            if (s.masses[i] == 0) continue;
            const int kind = U(0, 3), np = kind == 0 ? 2 : 3;
            int par[3] = {i, -1, -1};
            bool ok = true;
            for (int q = 1; q < np; q++) {
                par[q] = i + q;
                if (s.mol[par[q]] != s.mol[i] && U(0, 3) != 0) ok = false;
            }
            if (!ok) continue;
            s.masses.push_back(0.0); s.mol.push_back(s.mol[i]);
            s.vs.push_back((int) s.masses.size() - 1); s.vs.push_back(kind);
            for (int q = 0; q < 3; q++) s.vs.push_back(par[q]);
            for (int q = 0; q < 12; q++) s.vsp.push_back(0.1 * (q + 1));
        }
    }
    return s;
}

long analyse(const Sys& s, int use_com, int shard_parts, int part) {
    vvhip_system_desc d;
    std::memset(&d, 0, sizeof(d));
    const int n = (int) s.masses.size();
    d.num_atoms = n; d.padded_num_atoms = (n + 31) / 32 * 32;
    d.masses = s.masses.data(); d.mol_id = s.mol.data(); d.num_molecules = s.nmol;
    d.num_drude_pairs = (int) s.pairs.size() / 2; d.drude_pairs = s.pairs.data();
    d.num_constraints = (int) s.cons.size() / 2; d.constraints = s.cons.data(); d.constraint_distances = s.cdist.empty() ? nullptr : s.cdist.data();
    d.has_cm_motion_remover = 1;
    d.num_particles_ld = (int) s.ld.size(); d.particles_ld = s.ld.data();
    d.num_image_pairs = (int) s.img.size() / 2; d.image_pairs = s.img.data();
    d.num_electrolyte = (int) s.el.size(); d.particles_electrolyte = s.el.data();
    d.num_virtual_sites = (int) s.vs.size() / 5; d.virtual_sites = s.vs.data(); d.virtual_site_params = s.vsp.data();
    if (shard_parts > 1) {      // cut on molecule boundaries
        std::vector<int> starts;
        for (int i = 0; i < n; i++) if (i == 0 || s.mol[i] != s.mol[i - 1]) starts.push_back(i);
        const int per = std::max(1, (int) starts.size() / shard_parts);
        const int b = starts[std::min((size_t) (part * per), starts.size() - 1)];
        const int e = part == shard_parts - 1 ? n : starts[std::min((size_t) ((part + 1) * per), starts.size() - 1)];
        if (e <= b) return 0;
        d.shard_begin = b; d.shard_end = e;
    }
    vvhip_params p;
    std::memset(&p, 0, sizeof(p));
    p.temperature = 300; p.frequency = 10; p.drude_temperature = 1; p.drude_frequency = 40; p.step_size = 0.001;
    p.num_nh_chains = 3; p.loops_per_step = 1; p.max_drude_distance = 0.02; p.friction = 5; p.drude_friction = 20;
    p.use_com_temp_group = use_com; p.use_middle_scheme = 1; p.auto_set_com_temp_group = 0; p.auto_set_friction = 1; p.constraint_tolerance = 1e-5;
    long check = 0;
    try {
        const vv::HostPlan hp = vv::analyze(d, p, VVHIP_MIXED);
        // every slot points inside the shard, every table has the size the kernels index it with
        const int nloc = hp.shard_end - hp.shard_begin;
        const size_t nslots = (size_t) hp.info.num_waves * 64;
        if (hp.slots.size() != 2 * nslots) { std::fprintf(stderr, "slot table size\n"); std::exit(2); }
        for (size_t i = 0; i < nslots; i++) {
            const int32_t a = hp.slots[2 * i];
            if (a >= nloc) { std::fprintf(stderr, "slot out of shard\n"); std::exit(2); }
            if (a >= 0) check += a + (hp.slots[2 * i + 1] & 0xF);
        }
        if (hp.seg_base.size() != (size_t) hp.info.num_waves + 1 || hp.seg_mass.size() < 2 * (size_t) hp.seg_base.back()) { std::fprintf(stderr, "segment tables\n"); std::exit(2); }
        if (!hp.slot_shake.empty() && (hp.slot_shake.size() != nslots || hp.slot_shake_param.size() != 4 * nslots)) { std::fprintf(stderr, "shake tables\n"); std::exit(2); }
        if (!hp.slot_vsite.empty()) {
            if (hp.slot_vsite.size() != 2 * nslots || hp.vsite_params.size() != 12 * (size_t) hp.info.num_virtual_sites) { std::fprintf(stderr, "site tables\n"); std::exit(2); }
            for (size_t i = 0; i < nslots; i++)
                if ((uint32_t) hp.slot_vsite[2 * i] >> 31) {
                    if (hp.slots[2 * i] < 0 || hp.slot_vsite[2 * i + 1] < 0 || hp.slot_vsite[2 * i + 1] >= hp.info.num_virtual_sites) { std::fprintf(stderr, "site record\n"); std::exit(2); }
                    for (int q = 0; q < 3; q++)
                        if (hp.slots[2 * ((i & ~(size_t) 63) + (((uint32_t) hp.slot_vsite[2 * i] >> (6 * q)) & 63u))] < 0) { std::fprintf(stderr, "site parent lane is empty\n"); std::exit(2); }
                    check += hp.slot_vsite[2 * i + 1];
                }
        }
        check += hp.info.num_general_constraints + hp.info.periodic_layout * 1000003L + hp.info.num_waves;
    } catch (const vv::Error&) {
        check += 7;            // refused inventories (a Drude pair across molecules, ...) are fine: the point is that nothing reads out of bounds
    }
    return check;
}
}  // namespace

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 300;
    std::mt19937 rng(20241008);
    long total = 0;
    int periodic = 0, plans = 0;
    for (int r = 0; r < rounds; r++) {
        const int flavour = r % 7;                 // 0 plain, 1 hydrogen constraints, 2 rigid water, 3 big molecules, 4 Langevin + images,
                                                   // 5 bonds between heavy particles as well (general clusters), 6 virtual sites
        const Sys s = build(rng, flavour);
        for (int per = 0; per < 2; per++) {
            setenv("VVHIP_PERIODIC", per ? "1" : "0", 1);
            for (int use_com = 0; use_com < 2; use_com++) {
                const long c = analyse(s, use_com, 1, 0);
                total += c; plans++;
                if (c >= 1000003L) periodic++;
                if (r % 3 == 0) { total += analyse(s, use_com, 2, 0); total += analyse(s, use_com, 2, 1); plans += 2; }
            }
        }
    }
    std::printf("HOST SANITIZE OK plans=%d periodic=%d checksum=%ld\n", plans, periodic, total);
    return 0;
}
