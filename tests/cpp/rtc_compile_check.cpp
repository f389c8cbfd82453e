// Compiles one stage set of kernel A / B for gfx950 through the library's run-time path (vv_rtc.cpp) WITHOUT a GPU: what hipRTC is
// handed (the embedded sources, the options) is complete and the device code is free of host headers.
// usage: rtc_compile_check <A|B> <precision 0|1|2> <stage bits> <chain links>
#include <cstdio>
#include <cstdlib>

#include "../../openmm-velocityverlet_amd/csrc/vv_rtc.hpp"

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    std::vector<char> code;
    std::string name, log;
    const bool ok = vv::rtc_compile(argv[1][0], std::atoi(argv[2]), (uint32_t) std::strtoul(argv[3], nullptr, 0), std::atoi(argv[4]), "gfx950", code, name, log);
    std::printf("%s kernel=%s bytes=%zu seconds=%.2f\n", ok ? "OK" : "FAILED", name.c_str(), code.size(), vv::vv_rtc_compile_seconds);
    if (!ok) std::printf("%s\n", log.c_str());
    return ok ? 0 : 1;
}
