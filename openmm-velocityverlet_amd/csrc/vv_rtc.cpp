// vv_rtc.cpp -- see vv_rtc.hpp.
#include "vv_rtc.hpp"

#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "vv_args.hpp"
#include "vv_rtc_sources.inc"      // generated (Makefile): vv_src_device, vv_src_args, vv_src_layout, vv_src_probes, vv_src_vvhip

namespace vv {

std::atomic<unsigned long long> vv_rtc_compiled{0}, vv_rtc_failed{0}, vv_rtc_launches[2];
double vv_rtc_compile_seconds = 0;

static int& rtc_mode_value() {
    static int mode = [] {
        const char* e = std::getenv("VVHIP_RTC");
        return e ? std::atoi(e) : 1;
    }();
    return mode;
}
int rtc_mode() { return rtc_mode_value(); }
int set_rtc_mode(int mode) {
    const int old = rtc_mode_value();
    if (mode >= 0) rtc_mode_value() = mode;
    return old;
}

namespace {

// libhiprtc.so, opened on first use: a process that only runs compiled stage sets never maps the compiler
struct Rtc {
    void* lib = nullptr;
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    decltype(&hiprtcAddNameExpression) add_name = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetLoweredName) lowered = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    bool ok = false;
    Rtc() {
        for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return;
#define VV_RTC_SYM(field, sym) field = (decltype(field)) dlsym(lib, #sym); if (!field) return;
        VV_RTC_SYM(create, hiprtcCreateProgram) VV_RTC_SYM(destroy, hiprtcDestroyProgram) VV_RTC_SYM(add_name, hiprtcAddNameExpression)
        VV_RTC_SYM(compile, hiprtcCompileProgram) VV_RTC_SYM(lowered, hiprtcGetLoweredName) VV_RTC_SYM(log_size, hiprtcGetProgramLogSize)
        VV_RTC_SYM(log, hiprtcGetProgramLog) VV_RTC_SYM(code_size, hiprtcGetCodeSize) VV_RTC_SYM(code, hiprtcGetCode)
#undef VV_RTC_SYM
        ok = true;
    }
};
Rtc& rtc() { static Rtc r; return r; }

std::string replaced(std::string s, const std::string& from, const std::string& to) {
    for (size_t i = s.find(from); i != std::string::npos; i = s.find(from, i + to.size())) s.replace(i, from.size(), to);
    return s;
}

}  // namespace

bool rtc_compile(char kind, int precision, uint32_t flags, int num_chains, const std::string& arch, std::vector<char>& code, std::string& lowered_name,
                 std::string& log, uint32_t flags_a) {
    Rtc& r = rtc();
    if (!r.ok) { log = "libhiprtc.so could not be opened"; return false; }
    if (flags == 0 || (kind != 'A' && kind != 'B')) { log = "no stage bits"; return false; }
    const char* real = precision == VVHIP_DOUBLE ? "double" : "float";
    const char* mixed = precision == VVHIP_SINGLE ? "float" : "double";
    char expr[160];
    if (flags_a != 0 && kind != 'B') { log = "the fused step is an instance of kernel B"; return false; }
    if (flags_a != 0) std::snprintf(expr, sizeof expr, "vv::vv_kernel_b<%s, %s, %uu, %uu>", real, mixed, flags, flags_a);
    else std::snprintf(expr, sizeof expr, "vv::vv_kernel_%c<%s, %s, %uu>", kind == 'A' ? 'a' : 'b', real, mixed, flags);
    // the translation unit: the library's own headers and device code, then the one instantiation asked for
    // (hipRTC keeps its fixed-width integer types in a namespace of its own)
    const std::string unit = std::string("using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; using __hip_internal::uint64_t;\n"
                                         "#include \"vv_args.hpp\"\nnamespace vv {\n#include \"vv_device.inc\"\n}\n");
    // (hipRTC has no host headers: the public header's two C includes and the relative path to it are edited out of the embedded copies)
    const std::string h_args = replaced(vv_src_args, "#include \"../../include/vvhip.h\"", "#include \"vvhip.h\"");
    const std::string h_vvhip = replaced(replaced(vv_src_vvhip, "#include <stddef.h>", ""), "#include <stdint.h>", "");
    const char* header_src[] = {h_args.c_str(), vv_src_layout, vv_src_device, vv_src_probes, h_vvhip.c_str()};
    const char* header_name[] = {"vv_args.hpp", "vv_layout.h", "vv_device.inc", "vv_probes.inc", "vvhip.h"};
    hiprtcProgram prog = nullptr;
    if (r.create(&prog, unit.c_str(), "vv_rtc_unit.hip", 5, header_src, header_name) != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram failed"; return false; }
    bool ok = r.add_name(prog, expr) == HIPRTC_SUCCESS;
    // the options of the ahead-of-time build (Makefile): contraction off (element-wise results equal the reference's operation for
    // operation), the leading scalar arguments preloaded into SGPRs
    const std::string arch_opt = "--offload-arch=" + arch;
    char links[40];
    std::snprintf(links, sizeof links, "-DVV_SF_CHAIN_LINKS=%d", num_chains >= 1 && num_chains <= 4 ? num_chains : 3);
    std::vector<const char*> opts = {arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=16", links};
    const auto t0 = std::chrono::steady_clock::now();
    if (ok) ok = r.compile(prog, (int) opts.size(), opts.data()) == HIPRTC_SUCCESS;
    vv_rtc_compile_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t n = 0;
    if (r.log_size(prog, &n) == HIPRTC_SUCCESS && n > 1) { log.resize(n); (void) r.log(prog, &log[0]); }
    if (ok) {
        const char* low = nullptr;
        ok = r.lowered(prog, expr, &low) == HIPRTC_SUCCESS && low;
        if (ok) lowered_name = low;
    }
    if (ok) ok = r.code_size(prog, &n) == HIPRTC_SUCCESS && n > 0;
    if (ok) { code.resize(n); ok = r.code(prog, code.data()) == HIPRTC_SUCCESS; }
    (void) r.destroy(&prog);
    return ok;
}

hipFunction_t rtc_kernel(char kind, int precision, uint32_t flags, int num_chains, uint32_t flags_a) {
    static std::mutex mutex;
    static std::map<std::tuple<int, char, int, uint32_t, int, uint32_t>, hipFunction_t> cache;      // (device, kernel, precision, stage bits, chain length, fused: A's stage bits)
    std::lock_guard<std::mutex> lock(mutex);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (kind == 'A' || !(flags & B_CHAIN)) num_chains = 3;          // only kernel B's thermostat wave (B_CHAIN) depends on it
    const auto key = std::make_tuple(dev, kind, precision, flags, num_chains, flags_a);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    hipFunction_t fn = nullptr;
    if (const char* deny = std::getenv("VVHIP_RTC_DENY")) {      // debugging: "A:0x13,B:0x10004,A:*" -- these stage sets stay on the generic kernel
        char tag[32];
        std::snprintf(tag, sizeof tag, "%c:0x%x", kind, flags);
        char any[8];
        std::snprintf(any, sizeof any, "%c:*", kind);
        // whole comma-separated entries only ("A:0x13" must not match inside "A:0x1300")
        bool denied = false;
        for (const char* q = deny; *q && !denied;) {
            const char* e = std::strchr(q, ',');
            const size_t len = e ? (size_t) (e - q) : std::strlen(q);
            denied = (len == std::strlen(tag) && std::strncmp(q, tag, len) == 0) || (len == std::strlen(any) && std::strncmp(q, any, len) == 0);
            q += len + (e ? 1 : 0);
        }
        if (denied) { cache[key] = nullptr; return nullptr; }
    }
    hipDeviceProp_t prop;
    std::string why;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        std::vector<char> code;
        std::string name, log;
        if (rtc_compile(kind, precision, flags, num_chains, prop.gcnArchName, code, name, log, flags_a)) {
            // a plan may meet its stage set for the first time while its step is being captured into a graph: loading a code object is
            // not a stream operation, but it allocates, which a thread-local capture forbids -- relax the mode around the load
            hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
            (void) hipThreadExchangeStreamCaptureMode(&mode);
            hipModule_t mod = nullptr;
            hipError_t e = hipModuleLoadData(&mod, code.data());
            if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, name.c_str());
            (void) hipThreadExchangeStreamCaptureMode(&mode);
            if (e != hipSuccess) { fn = nullptr; why = std::string("loading the code object: ") + hipGetErrorString(e); }
            else {
                vv_rtc_compiled++;
                static const bool verbose = std::getenv("VVHIP_RTC_VERBOSE") != nullptr;
                if (verbose) std::fprintf(stderr, "vvhip: compiled kernel %c for stage set 0x%x (precision %d, %d chain links) at run time\n", kind, flags, precision, num_chains);
            }
        } else {
            why = log;
        }
    } else {
        why = "hipGetDeviceProperties failed";
    }
    if (!fn) vv_rtc_failed++;
    if (!fn) std::fprintf(stderr, "vvhip: run-time compilation of kernel %c, stage set 0x%x failed (the generic kernel runs instead): %s\n", kind, flags, why.c_str());
    cache[key] = fn;
    return fn;
}

}  // namespace vv
