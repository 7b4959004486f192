// functor_rtc.hip — mi355_spmv_functor_*: a generalized SpMV whose functor is the CALLER'S code.
//
// The reference's SpMV_merge_based_generalized is a template over a functor_t with static initialize / combine /
// reduce (include/spmv/merge_genl/merge_genl.cuh:19-38) and over FIVE independent types (include/spmv.h:29-34).  A C ABI
// cannot take a C++ type — the merge kind's semirings are an enumeration for that reason — but it can take its TEXT:
// the functor's source and the names of the types are compiled here for gfx950 at run time (hiprtc), in front of the
// kernels of functor_kernel.inc.  hiprtc is bound on first use (dlopen), like RCCL: the library keeps the HIP runtime as
// its only link dependency.
//
// compile  = hiprtc only, no device needed (the code object is kept by the handle);
// spmv     = module loaded on the current device on first use, two launches on the caller's stream, no host sync.

#include <dlfcn.h>
#include <hip/hiprtc.h>   // types and prototypes only; no symbol of it is linked

#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "common.hpp"

namespace mi355 {

static const char kFunctorKernelSource[] =
#include "functor_kernel.inc"
    ;

struct RtcApi {
    void* handle = nullptr;
    hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char* const*) = nullptr;
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*DestroyProgram)(hiprtcProgram*) = nullptr;
    const char* (*GetErrorString)(hiprtcResult) = nullptr;
};

static RtcApi g_rtc;
static std::mutex g_rtc_mutex;
static bool g_rtc_tried = false;
static char g_rtc_error[256] = "";
static thread_local std::string g_functor_log;      // the compiler's output of this thread's last compile

static RtcApi* rtc_api() {
    std::lock_guard<std::mutex> lock(g_rtc_mutex);
    if (!g_rtc_tried) {
        g_rtc_tried = true;
        RtcApi api;
        const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char* why = dlerror();
            snprintf(g_rtc_error, sizeof(g_rtc_error), "mi355_spmv_functor: libhiprtc.so not found (%s)", why ? why : "?");
        } else {
            bool ok = true;
            const char* missing = "";
            auto bind = [&](const char* sym) -> void* {
                void* f = dlsym(api.handle, sym);
                if (!f) { ok = false; missing = sym; }
                return f;
            };
            api.CreateProgram = reinterpret_cast<decltype(api.CreateProgram)>(bind("hiprtcCreateProgram"));
            api.CompileProgram = reinterpret_cast<decltype(api.CompileProgram)>(bind("hiprtcCompileProgram"));
            api.GetProgramLogSize = reinterpret_cast<decltype(api.GetProgramLogSize)>(bind("hiprtcGetProgramLogSize"));
            api.GetProgramLog = reinterpret_cast<decltype(api.GetProgramLog)>(bind("hiprtcGetProgramLog"));
            api.GetCodeSize = reinterpret_cast<decltype(api.GetCodeSize)>(bind("hiprtcGetCodeSize"));
            api.GetCode = reinterpret_cast<decltype(api.GetCode)>(bind("hiprtcGetCode"));
            api.DestroyProgram = reinterpret_cast<decltype(api.DestroyProgram)>(bind("hiprtcDestroyProgram"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(bind("hiprtcGetErrorString"));
            if (!ok) {
                snprintf(g_rtc_error, sizeof(g_rtc_error), "mi355_spmv_functor: libhiprtc.so lacks %s", missing);
                dlclose(api.handle);
                api = RtcApi();
            }
            g_rtc = api;
        }
    }
    if (!g_rtc.handle) {
        set_error("%s", g_rtc_error);
        return nullptr;
    }
    return &g_rtc;
}

}  // namespace mi355

using namespace mi355;

namespace {
constexpr int kLaneChoices = 6;                      // T = 2, 4, 8, 16, 32, 64
struct Loaded {                                      // the code object on one device
    int device = -1;
    hipModule_t module = nullptr;
    hipFunction_t rows[kLaneChoices] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipFunction_t long_rows = nullptr;
};
}  // namespace

struct mi355_spmv_functor {
    int off_type = MI355_OFF_I32;
    std::vector<char> code;
    std::mutex mutex;                                // guards `loaded`
    std::deque<Loaded> loaded;                       // (a deque: references handed out stay valid as devices are added)
};

namespace {

// A type or functor name goes into a typedef: one line of identifier-like text (letters, digits, _ : < > , * & space)
bool plain_type_text(const char* s) {
    if (!s || !*s) return false;
    for (const char* p = s; *p; ++p) {
        const char c = *p;
        const bool ok = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == ':' ||
                        c == '<' || c == '>' || c == ',' || c == ' ' || c == '*' || c == '&';
        if (!ok) return false;
    }
    return true;
}

int load_on_current_device(mi355_spmv_functor* f, const Loaded** out) {
    int device = 0;
    MI355_HIP_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(f->mutex);
    for (const Loaded& l : f->loaded)
        if (l.device == device) { *out = &l; return MI355_SPMV_OK; }
    Loaded l;
    l.device = device;
    MI355_HIP_TRY(hipModuleLoadData(&l.module, f->code.data()));
    static const char* const names[kLaneChoices] = {"mi355_functor_rows_2",  "mi355_functor_rows_4",  "mi355_functor_rows_8",
                                                    "mi355_functor_rows_16", "mi355_functor_rows_32", "mi355_functor_rows_64"};
    hipError_t e = hipSuccess;
    for (int i = 0; i < kLaneChoices && e == hipSuccess; ++i) e = hipModuleGetFunction(&l.rows[i], l.module, names[i]);
    if (e == hipSuccess) e = hipModuleGetFunction(&l.long_rows, l.module, "mi355_functor_long_rows");
    if (e != hipSuccess) {
        (void)hipModuleUnload(l.module);
        set_error("functor_spmv: hipModuleGetFunction -> %s", hipGetErrorString(e));
        return MI355_SPMV_EHIP;
    }
    f->loaded.push_back(l);
    *out = &f->loaded.back();
    return MI355_SPMV_OK;
}

}  // namespace

extern "C" {

int mi355_spmv_functor_compile(mi355_spmv_functor** out, const char* source, const char* functor_type, int off_type,
                               const char* mat_type, const char* x_type, const char* y_type) {
    g_functor_log.clear();
    if (!out) { set_error("functor_compile: null pointer"); return MI355_SPMV_EINVAL; }
    *out = nullptr;
    if (!source) { set_error("functor_compile: null source"); return MI355_SPMV_EINVAL; }
    if (off_type != MI355_OFF_I32 && off_type != MI355_OFF_I64) { set_error("functor_compile: unknown offset type %d", off_type); return MI355_SPMV_EINVAL; }
    if (!plain_type_text(functor_type) || !plain_type_text(mat_type) || !plain_type_text(x_type) || !plain_type_text(y_type)) {
        set_error("functor_compile: the functor and the three value types are given as type names (one line each)");
        return MI355_SPMV_EINVAL;
    }
    RtcApi* api = rtc_api();
    if (!api) return MI355_SPMV_ENOTSUP;
    std::string text;
    text.reserve(strlen(source) + sizeof(kFunctorKernelSource) + 512);
    // (hiprtc has no <cmath>: the two macros a functor's identity usually wants)
    text += "#ifndef INFINITY\n#define INFINITY __builtin_huge_valf()\n#endif\n#ifndef NAN\n#define NAN __builtin_nanf(\"\")\n#endif\n";
    text += "#line 1 \"functor_source\"\n";
    text += source;
    text += "\n#line 1 \"mi355_functor_kernels\"\n";
    text += "typedef ";  text += functor_type;  text += " mi355_functor_t;\n";
    text += off_type == MI355_OFF_I64 ? "typedef long long mi355_off_t;\n" : "typedef int mi355_off_t;\n";
    text += "typedef ";  text += mat_type;  text += " mi355_mat_t;\n";
    text += "typedef ";  text += x_type;    text += " mi355_x_t;\n";
    text += "typedef ";  text += y_type;    text += " mi355_y_t;\n";
    text += kFunctorKernelSource;
    hiprtcProgram prog = nullptr;
    hiprtcResult r = api->CreateProgram(&prog, text.c_str(), "mi355_functor.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) { set_error("hiprtcCreateProgram -> %s", api->GetErrorString(r)); return MI355_SPMV_EHIP; }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    r = api->CompileProgram(prog, 3, opts);
    size_t log_size = 0;
    if (api->GetProgramLogSize(prog, &log_size) == HIPRTC_SUCCESS && log_size > 1) {
        g_functor_log.resize(log_size);
        if (api->GetProgramLog(prog, &g_functor_log[0]) != HIPRTC_SUCCESS) g_functor_log.clear();
        while (!g_functor_log.empty() && g_functor_log.back() == 0) g_functor_log.pop_back();
    }
    if (r != HIPRTC_SUCCESS) {
        set_error("functor_compile: %s; first lines of the log: %.300s", api->GetErrorString(r), g_functor_log.c_str());
        (void)api->DestroyProgram(&prog);
        return MI355_SPMV_EINVAL;                    // (the caller's text did not compile)
    }
    mi355_spmv_functor* f = new (std::nothrow) mi355_spmv_functor();
    size_t code_size = 0;
    if (f) r = api->GetCodeSize(prog, &code_size);
    if (f && r == HIPRTC_SUCCESS) {
        f->code.resize(code_size);
        r = api->GetCode(prog, f->code.data());
    }
    (void)api->DestroyProgram(&prog);
    if (!f) { set_error("functor_compile: host allocation failed"); return MI355_SPMV_ENOMEM; }
    if (r != HIPRTC_SUCCESS) { set_error("hiprtcGetCode -> %s", api->GetErrorString(r)); delete f; return MI355_SPMV_EHIP; }
    f->off_type = off_type;
    *out = f;
    return MI355_SPMV_OK;
}

const char* mi355_spmv_functor_compile_log(void) { return g_functor_log.c_str(); }

int mi355_spmv_functor_spmv(mi355_spmv_functor* f, int32_t n_rows, int32_t n_cols, int64_t nnz, const void* Ap,
                            const int32_t* Aj, const void* Ax, const void* x, void* y, void* stream) {
    if (!f) { set_error("functor_spmv: null handle"); return MI355_SPMV_EINVAL; }
    if (n_rows < 0 || n_cols < 0 || nnz < 0) { set_error("functor_spmv: negative size"); return MI355_SPMV_EINVAL; }
    if (n_rows == 0) return MI355_SPMV_OK;
    if (!Ap || !y || (nnz > 0 && (!Aj || !Ax || !x))) { set_error("functor_spmv: null array"); return MI355_SPMV_EINVAL; }
    if (nnz > 0 && n_cols == 0) { set_error("functor_spmv: nonzeros but no columns"); return MI355_SPMV_EINVAL; }
    const Loaded* l = nullptr;
    if (const int st = load_on_current_device(f, &l)) return st;
    // lanes per row from the mean row length (the CSR-vector rule of the reference, cusp.cuh:203-221, on a 64-wide wave)
    // (a lane takes four nonzeros per step: T lanes cover a row of 4 T in one)
    const int64_t mean = nnz / n_rows;
    int choice = 0;                                  // T = 2 << choice
    while (choice < kLaneChoices - 1 && (int64_t(8) << choice) < mean) ++choice;
    const int T = 2 << choice;
    long long long_len = 128ll * T;                  // (32 steps of the row's T lanes)
    if (long_len < 2048) long_len = 2048;
    const int rows_per_wg = kBlock / T;
    const int64_t want = (int64_t(n_rows) + rows_per_wg - 1) / rows_per_wg;
    const unsigned grid = unsigned(want < int64_t(kCus) * 32 ? want : int64_t(kCus) * 32);
    void* args[] = {&n_rows, &Ap, &Aj, &Ax, &x, &y, &long_len};
    hipStream_t s = static_cast<hipStream_t>(stream);
    MI355_HIP_TRY(hipModuleLaunchKernel(l->rows[choice], grid, 1, 1, kBlock, 1, 1, 0, s, args, nullptr));
    // the long rows: a row of that length only exists when nnz allows it
    if (nnz > long_len) {
        const int64_t waves = (int64_t(n_rows) + kWave - 1) / kWave;
        const int64_t wgs = (waves + kBlock / kWave - 1) / (kBlock / kWave);
        const unsigned g2 = unsigned(wgs < int64_t(kCus) * 8 ? wgs : int64_t(kCus) * 8);
        MI355_HIP_TRY(hipModuleLaunchKernel(l->long_rows, g2, 1, 1, kBlock, 1, 1, 0, s, args, nullptr));
    }
    return MI355_SPMV_OK;
}

int mi355_spmv_functor_destroy(mi355_spmv_functor* f) {
    if (!f) return MI355_SPMV_OK;
    int saved = 0;
    const bool have = hipGetDevice(&saved) == hipSuccess;
    for (Loaded& l : f->loaded) {
        if (hipSetDevice(l.device) == hipSuccess && l.module) (void)hipModuleUnload(l.module);
    }
    if (have) (void)hipSetDevice(saved);
    delete f;
    return MI355_SPMV_OK;
}

}  // extern "C"
