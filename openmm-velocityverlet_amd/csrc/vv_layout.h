// vv_layout.h -- the role word of a lane and the cluster word of the in-kernel constraints: what the host analysis (vv_host.cpp) writes
// into the slot tables and the kernels (vv_device.inc) read.  No includes: this header is also compiled at run time (vv_rtc.cpp).
#pragma once

namespace vv {

// ---- role word of one lane ("slot"): what this particle needs from the kernels -----------------
enum Role : uint32_t {
    ROLE_NONE = 0,       // no integration work (idle lane, or a massless particle kept only as image parent)
    ROLE_PLAIN = 1,      // massive, neither NH nor Langevin (a massive image particle): kick + drift only
    ROLE_NH_NORMAL = 2,  // NH thermostat, not in a Drude pair      (normalParticlesNH, HOST:529)
    ROLE_NH_DRUDE = 3,   // NH, Drude particle of a pair (pair.x)   (pairParticlesNH,   HOST:523)
    ROLE_NH_PARENT = 4,  // NH, parent atom of a pair (pair.y)
    ROLE_LD_NORMAL = 5,  // Langevin subset                          (normalParticlesLD, HOST:804)
    ROLE_LD_DRUDE = 6,   //                                          (pairParticlesLD,   HOST:791)
    ROLE_LD_PARENT = 7,
};
constexpr uint32_t META_ROLE_MASK = 0xF;
constexpr int META_PARTNER_SHIFT = 4;    // 6 bits: lane of the Drude partner (own lane if none)
constexpr int META_SEGFIRST_SHIFT = 10;  // 6 bits: first lane of this lane's COM segment (molecule)
constexpr int META_SEGLAST_SHIFT = 16;   // 6 bits: last lane of the segment
constexpr uint32_t META_EFIELD = 1u << 22;     // particle is in particlesElectrolyte
constexpr uint32_t META_HAS_IMAGE = 1u << 23;  // particle is the parent of an image particle
constexpr uint32_t META_COM_LEADER = 1u << 24; // lane that adds its molecule's M*V^2 to TG_COM
constexpr uint32_t META_PAIR = 1u << 25;       // member of a DrudeForce pair (hard wall applies)
constexpr uint32_t META_IS_DRUDE = 1u << 26;   // the Drude (pair.x) of that pair
constexpr uint32_t META_MASSIVE = 1u << 27;    // mass != 0 (velm.w != 0)
constexpr uint32_t META_BIGMOL = 1u << 28;     // lane belongs to a molecule too large for one wave: its COM comes from bigacc
// slot_shake word of a lane that belongs to a constraint cluster (0 otherwise); every member carries the whole cluster, so that each
// lane can gather its mates' data itself: bit 0 central (apex) lane, bit 1 peripheral lane, bits 2-3 number of peripherals np,
// bits 4-9 / 10-15 / 16-21 lanes of peripherals 0 / 1 / 2 (unused ones: the central lane), bits 22-27 lane of the central particle,
// bits 28-29 the lane's own index among the peripherals, bit 30 the cluster is a rigid triangle (SETTLE)
constexpr int SHAKE_WORD_CENTRAL_SHIFT = 22;
constexpr int SHAKE_WORD_OWN_SHIFT = 28;
constexpr uint32_t SHAKE_WORD_SETTLE = 1u << 30;
constexpr uint32_t META_SHAKE = 1u << 30;      // member of an in-kernel constraint cluster: the kernels fetch its cluster word, parameters and
                                               // position in the same round of loads as the velocity, not after reading the cluster word
constexpr uint32_t META_BIG_FIRST = 1u << 29;  // leader of the FIRST chunk of such a molecule (adds M*V^2 once, clears bigacc)

inline uint32_t meta_role(uint32_t m) { return m & META_ROLE_MASK; }

}  // namespace vv
