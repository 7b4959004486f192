// spmv/mi355.hpp — the MI355X kinds, in the shape the reference asks of a new kind
// (reference README.md:28-45): a function template
//     SpMV_<name>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y)
// with the five type parameters of include/spmv.h:29-34, device pointers owned by
// the caller, y overwritten.  Each kind is a thin shim over the C ABI of
// include/mi355_spmv.h (libmi355spmv.so holds the HIP kernels):
//
//   SpMV_hip_vector   CSR-vector, sub-wave reduction   (cf. SpMV_cusp_warp_reduce,
//                                                       cusp_warp_reduce.cuh:138-147)
//   SpMV_hip_merge    merge-path                       (cf. SpMV_merge_based,
//                                                       merge_based.cuh:22-56)
//   SpMV_hip_light    dynamic row distribution         (cf. SpMV_light_warp,
//                                                       LightSpMV.cuh:400-416)
//
// Life cycle per call = the reference's: scratch is created before and released
// after the launch (LightSpMV.cuh:383-395), Timer::kernel_start/stop bracket launch
// + synchronise (cusp_warp_reduce.cuh:130-132), the null stream is used
// (main.cu:87-88 relies on it).  A device error aborts with file:line like
// checkCudaErr (common.cuh:13-23).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "../../../include/mi355_spmv.h"
#include "../timer.hpp"

namespace mi355_host {

inline void check(int status, const char* what, const char* file, int line) {
    if (status != MI355_SPMV_OK) {
        std::fprintf(stderr, "MI355 SpMV error at %s:%d code=%d(%s) \"%s\" : %s\n", file, line, status,
                     mi355_spmv_status_string(status), what, mi355_spmv_last_error());
        std::abort();
    }
}
#define MI355_CHECK(expr) ::mi355_host::check((expr), #expr, __FILE__, __LINE__)

// Semiring tags for the generalized merge kind.  The reference passes a functor_t with static
// initialize / combine / reduce (include/spmv/merge_genl/merge_genl.cuh:19-38) into device code; the
// kernels here live behind a C ABI, so the functor is a tag naming one of the built-in semirings.
struct PlusTimes { static constexpr int id = MI355_SEMIRING_PLUS_TIMES; };   // the reference's MergeFunctor
struct MinPlus   { static constexpr int id = MI355_SEMIRING_MIN_PLUS; };
struct MaxTimes  { static constexpr int id = MI355_SEMIRING_MAX_TIMES; };
struct MaxPlus   { static constexpr int id = MI355_SEMIRING_MAX_PLUS; };
struct OrAnd     { static constexpr int id = MI355_SEMIRING_OR_AND; };     // booleans as 0.0 / 1.0 (or 0 / 1 on int values)

// (+, *) for any mix of value types, on the functor-text path (defined at the end of this file)
template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
void run_general(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
                 const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y);

template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t,
          typename vec_y_value_t>
void run_kind(int kind, index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
              const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y,
              int semiring = MI355_SEMIRING_PLUS_TIMES) {
    static_assert(std::is_same<index_t, int>::value || std::is_same<index_t, int32_t>::value,
                  "mi355 kinds: index_t must be a 32-bit int (reference main.cu:15)");
    static_assert(sizeof(offset_t) == 4 || sizeof(offset_t) == 8, "mi355 kinds: offset_t must be 32- or 64-bit");
    static_assert(std::is_integral<offset_t>::value && std::is_signed<offset_t>::value,
                  "mi355 kinds: offset_t must be a signed integer");
    // The TUNED kernels are built for: A, x and y of one floating-point type (reference main.cu:17; every kind); an fp32
    // matrix under fp64 vectors and 32-bit integers throughout (the merge kinds).  The reference's template keeps the
    // three value types apart (spmv.h:29-34; merge_genl.cuh:29-31 computes in the y type): every OTHER mix of float /
    // double / int / long long runs the ordinary (+, *) on the general path — the functor-text kernels, compiled at
    // run time for exactly these types (run_general below) — so that every kind accepts what the reference's does.
    constexpr bool is_fp_mat = std::is_same<mat_value_t, float>::value || std::is_same<mat_value_t, double>::value;
    constexpr bool same3 = std::is_same<mat_value_t, vec_x_value_t>::value && std::is_same<vec_x_value_t, vec_y_value_t>::value;
    constexpr bool all_int = same3 && std::is_same<mat_value_t, int>::value;
    constexpr bool f32_under_f64 = std::is_same<mat_value_t, float>::value && std::is_same<vec_x_value_t, double>::value &&
                                   std::is_same<vec_y_value_t, double>::value;
    if (kind == MI355_KIND_AUTO && (all_int || f32_under_f64)) kind = MI355_KIND_MERGE;   // (what the library would pick)
    const bool tuned = (same3 && is_fp_mat) || ((all_int || f32_under_f64) && kind == MI355_KIND_MERGE);
    if (!tuned) {
        if (semiring != MI355_SEMIRING_PLUS_TIMES) {
            std::fprintf(stderr, "mi355 kinds: the enumerated semirings other than (+, *) are built for one value type; "
                                 "pass a functor's text to SpMV_hip_functor for these types\n");
            std::abort();
        }
        run_general(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);
        return;
    }
    const int off_type = sizeof(offset_t) == 8 ? MI355_OFF_I64 : MI355_OFF_I32;
    const int mat_type = all_int ? MI355_VAL_I32 : std::is_same<mat_value_t, double>::value ? MI355_VAL_F64 : MI355_VAL_F32;
    const int vec_type = all_int ? MI355_VAL_I32 : std::is_same<vec_x_value_t, double>::value ? MI355_VAL_F64 : MI355_VAL_F32;

    // The reference's harness calls a kind TEST_TIMES times in a row on one matrix (main.cu:102-113): the plan of the
    // previous call is found again by the structure pointers and sizes (mi355_spmv_plan_acquire; safe if the arrays were
    // rewritten in place, include/mi355_spmv.h).  Mixed value types go through plan_create_typed every time.
    mi355_spmv_plan* plan = nullptr;
    const bool kept = mat_type == vec_type;
    if (kept)
        MI355_CHECK(mi355_spmv_plan_acquire(&plan, kind, off_type, vec_type, (int32_t)n_rows, (int32_t)n_cols, (int64_t)nnz, Ap,
                                            reinterpret_cast<const int32_t*>(Aj)));
    else
        MI355_CHECK(mi355_spmv_plan_create_typed(&plan, kind, off_type, mat_type, vec_type, vec_type, (int32_t)n_rows,
                                                 (int32_t)n_cols, (int64_t)nnz, Ap, reinterpret_cast<const int32_t*>(Aj),
                                                 MI355_PLAN_DEFAULT));
    MI355_CHECK(mi355_spmv_plan_set_semiring(plan, semiring));
    Timer::kernel_start();
    MI355_CHECK(mi355_spmv_plan_execute(plan, Ax, x, y, /*stream=*/nullptr));
    MI355_CHECK(mi355_spmv_stream_synchronize(/*stream=*/nullptr));
    Timer::kernel_stop();
    if (kept) MI355_CHECK(mi355_spmv_plan_release(plan, 1));
    else MI355_CHECK(mi355_spmv_plan_destroy(plan));
}

// ---- multi-GPU kinds (SURVEY §8(b) last row, §8(e)): "a separate kind/entry that owns its per-device
// plans and RCCL communicator".  The reference is single-device (main.cu:53, common.cuh:8); these kinds take
// the same 8 arguments — device arrays on the CURRENT device — and spread the rows over dist_gpus() GPUs of
// the node through mi355_spmv_dist_* (LOCAL mode: this one host thread drives every GPU).  Unlike the
// single-GPU kinds they keep their handle between calls (cutting the matrix and copying the blocks to the
// other GPUs per call would dwarf the SpMV) — and the handle holds COPIES of the structure, so, to keep the
// semantics of SpMV(kind, ...), which reads its arrays on every call:
//   * the handle is rebuilt when the pointers or sizes change, AND when the fingerprint of Ap / Aj differs from
//     the one taken at create (mi355_spmv_dist_structure_changed: one small kernel per call; a matrix rewritten
//     in place, or another one at the same addresses).  MI355_DIST_TRUST_STRUCTURE=1 skips that check;
//   * Ax is scattered to the other GPUs on EVERY call (values rewritten in place: an iterative solver);
//     MI355_DIST_REUSE_VALUES=1 scatters only when the Ax pointer changes;
//   * x is replicated on every call.
inline int& dist_gpus() {
    static int n = [] { const char* e = std::getenv("MI355_NGPU"); const int v = e ? std::atoi(e) : 1; return v > 0 ? v : 1; }();
    return n;
}
inline bool env_flag(const char* name) { const char* e = std::getenv(name); return e && std::atoi(e) != 0; }
inline bool& dist_trust_structure() { static bool v = env_flag("MI355_DIST_TRUST_STRUCTURE"); return v; }
inline bool& dist_reuse_values() { static bool v = env_flag("MI355_DIST_REUSE_VALUES"); return v; }
inline int& dist_sub_blocks() {
    static int n = [] { const char* e = std::getenv("MI355_SUB_BLOCKS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 0; }();
    return n;
}
struct DistCache {
    mi355_spmv_dist* d = nullptr;
    int kind = -1, off_type = -1, val_type = -1, gpus = 0, sub = 0;
    long long n_rows = -1, n_cols = -1, nnz = -1;
    const void *Ap = nullptr, *Aj = nullptr, *Ax = nullptr;
};
inline DistCache& dist_cache() { static DistCache c; return c; }
inline void dist_release() {
    DistCache& c = dist_cache();
    if (c.d) MI355_CHECK(mi355_spmv_dist_destroy(c.d));
    c = DistCache();
}

template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t,
          typename vec_y_value_t>
void run_dist_kind(int kind, index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
                   const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    static_assert(std::is_same<index_t, int>::value || std::is_same<index_t, int32_t>::value,
                  "mi355 kinds: index_t must be a 32-bit int (reference main.cu:15)");
    static_assert(std::is_integral<offset_t>::value && std::is_signed<offset_t>::value && (sizeof(offset_t) == 4 || sizeof(offset_t) == 8),
                  "mi355 kinds: offset_t must be a signed 32- or 64-bit integer");
    static_assert(std::is_same<mat_value_t, vec_x_value_t>::value && std::is_same<mat_value_t, vec_y_value_t>::value,
                  "mi355 multi-GPU kinds: A, x and y share one value type");
    static_assert(std::is_same<mat_value_t, float>::value || std::is_same<mat_value_t, double>::value,
                  "mi355 kinds: value type is float or double");
    const int off_type = sizeof(offset_t) == 8 ? MI355_OFF_I64 : MI355_OFF_I32;
    const int val_type = std::is_same<mat_value_t, double>::value ? MI355_VAL_F64 : MI355_VAL_F32;
    const int gpus = dist_gpus();
    const int sub = dist_sub_blocks() > 0 ? dist_sub_blocks() : (gpus > 1 ? 4 : 1);
    DistCache& c = dist_cache();
    bool same = c.d && c.kind == kind && c.off_type == off_type && c.val_type == val_type && c.gpus == gpus &&
                c.sub == sub && c.n_rows == (long long)n_rows && c.n_cols == (long long)n_cols &&
                c.nnz == (long long)nnz && c.Ap == Ap && c.Aj == Aj;
    if (same && !dist_trust_structure()) {
        int changed = 0;
        MI355_CHECK(mi355_spmv_dist_structure_changed(c.d, Ap, reinterpret_cast<const int32_t*>(Aj), /*stream=*/nullptr, &changed));
        same = !changed;
    }
    if (!same) {
        dist_release();
        MI355_CHECK(mi355_spmv_dist_create_local(&c.d, kind, off_type, val_type, (int32_t)n_rows, (int32_t)n_cols,
                                                 (int64_t)nnz, Ap, reinterpret_cast<const int32_t*>(Aj), gpus,
                                                 /*devices=*/nullptr, sub, MI355_PLAN_DEFAULT));
        c.kind = kind; c.off_type = off_type; c.val_type = val_type; c.gpus = gpus; c.sub = sub;
        c.n_rows = n_rows; c.n_cols = n_cols; c.nnz = nnz; c.Ap = Ap; c.Aj = Aj; c.Ax = nullptr;
    }
    const void* ax = (gpus > 1 && dist_reuse_values() && c.Ax == Ax) ? nullptr : Ax;   // NULL = the other GPUs keep the values they hold
    Timer::kernel_start();
    MI355_CHECK(mi355_spmv_dist_execute(c.d, ax, x, y, /*stream=*/nullptr));
    MI355_CHECK(mi355_spmv_stream_synchronize(/*stream=*/nullptr));
    Timer::kernel_stop();
    c.Ax = Ax;
}

}  // namespace mi355_host

#define MI355_DEFINE_KIND(NAME, KIND)                                                                          \
    template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t,               \
              typename vec_y_value_t>                                                                          \
    void NAME(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,             \
              const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {                               \
        ::mi355_host::run_kind(KIND, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);                                   \
    }

/// CSR-vector SpMV with per-row sub-wave (wave64) reduction
MI355_DEFINE_KIND(SpMV_hip_vector, MI355_KIND_VECTOR)
/// merge-path load-balanced SpMV (search -> tile -> deterministic fix-up)
MI355_DEFINE_KIND(SpMV_hip_merge, MI355_KIND_MERGE)
/// LightSpMV-style dynamic row distribution (sharded atomic row counters)
MI355_DEFINE_KIND(SpMV_hip_light, MI355_KIND_LIGHT)

/// one of the three, picked by the library from the matrix's structure (MI355_KIND_AUTO, include/mi355_spmv.h)
MI355_DEFINE_KIND(SpMV_hip_auto, MI355_KIND_AUTO)

#define MI355_DEFINE_DIST_KIND(NAME, KIND)                                                                     \
    template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t,               \
              typename vec_y_value_t>                                                                          \
    void NAME(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,             \
              const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {                               \
        ::mi355_host::run_dist_kind(KIND, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);                              \
    }
/// the same three kinds over mi355_host::dist_gpus() GPUs: row blocks, x replicated, allgatherv(y) over RCCL
MI355_DEFINE_DIST_KIND(SpMV_hip_dist_vector, MI355_KIND_VECTOR)
MI355_DEFINE_DIST_KIND(SpMV_hip_dist_merge, MI355_KIND_MERGE)
MI355_DEFINE_DIST_KIND(SpMV_hip_dist_light, MI355_KIND_LIGHT)

/// generalized merge-path SpMV (cf. SpMV_merge_based_generalized, merge_genl.cuh:41-79);
/// functor_t is one of mi355_host::PlusTimes (default, the reference's MergeFunctor), MinPlus, MaxTimes, MaxPlus, OrAnd
template <typename functor_t = ::mi355_host::PlusTimes, typename index_t, typename offset_t,
          typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
void SpMV_hip_merge_generalized(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap,
                                const index_t* Aj, const mat_value_t* Ax, const vec_x_value_t* x,
                                vec_y_value_t* y) {
    ::mi355_host::run_kind(MI355_KIND_MERGE, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y, functor_t::id);
}

// ---- a functor of the caller's own (the reference's functor_t, merge_genl.cuh:19-38) -------------------------------
// The reference hands SpMV_merge_based_generalized a C++ type.  Here the kernels are compiled at run time from the
// functor's TEXT (mi355_spmv_functor_*, include/mi355_spmv.h): MI355_FUNCTOR(Name, definition...) keeps the definition
// as C++ for the host — the CPU twin SpMV_genl_cpu_navie<functor_t> (cpu_navie.hpp:20-34) takes the same type — and
// as a string for the device, so the functor is written ONCE, the way the reference writes it:
//
//   MI355_FUNCTOR(SaturatingAdd,
//       template <typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
//       struct SaturatingAdd {
//           __host__ __device__ __forceinline__ static vec_y_value_t initialize() { return vec_y_value_t(0); }
//           __host__ __device__ __forceinline__ static vec_y_value_t combine(const mat_value_t& a, const vec_x_value_t& x) { ... }
//           __host__ __device__ __forceinline__ static vec_y_value_t reduce(const vec_y_value_t& l, const vec_y_value_t& r) { ... }
//       };)
//   SpMV_hip_functor<SaturatingAdd_text>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);          // a template over the three
//   SpMV_hip_functor<SaturatingAdd_text, false>(...)                                      // a plain struct
//
// All five types of include/spmv.h:29-34 are free (float / double / int / long long values, int / long long offsets);
// the code object is compiled once per instantiation and kept for the life of the process.
#ifndef __HIPCC__
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
#endif
#endif
#define MI355_FUNCTOR(NAME, ...)                           \
    __VA_ARGS__                                            \
    struct NAME##_text {                                   \
        static const char* source() { return #__VA_ARGS__; } \
        static const char* name() { return #NAME; }        \
    };

namespace mi355_host {
template <typename T> struct c_type_name;
template <> struct c_type_name<float> { static const char* get() { return "float"; } };
template <> struct c_type_name<double> { static const char* get() { return "double"; } };
template <> struct c_type_name<int> { static const char* get() { return "int"; } };
template <> struct c_type_name<long long> { static const char* get() { return "long long"; } };
template <> struct c_type_name<long> { static const char* get() { return "long long"; } };   // (LP64: the same 64 bits)
}  // namespace mi355_host

/// generalized SpMV with the caller's own functor (cf. SpMV_merge_based_generalized<functor_t>, merge_genl.cuh:41-79)
template <typename functor_text, bool is_template = true, typename index_t, typename offset_t, typename mat_value_t,
          typename vec_x_value_t, typename vec_y_value_t>
void SpMV_hip_functor(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
                      const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    static_assert(sizeof(index_t) == 4, "mi355 kinds: index_t must be a 32-bit int (reference main.cu:15)");
    static_assert(sizeof(offset_t) == 4 || sizeof(offset_t) == 8, "mi355 kinds: offset_t must be 32- or 64-bit");
    static mi355_spmv_functor* compiled = [] {
        std::string type = functor_text::name();
        if (is_template)
            type += std::string("<") + ::mi355_host::c_type_name<mat_value_t>::get() + ", " +
                    ::mi355_host::c_type_name<vec_x_value_t>::get() + ", " + ::mi355_host::c_type_name<vec_y_value_t>::get() + ">";
        mi355_spmv_functor* f = nullptr;
        const int st = mi355_spmv_functor_compile(&f, functor_text::source(), type.c_str(),
                                                  sizeof(offset_t) == 8 ? MI355_OFF_I64 : MI355_OFF_I32,
                                                  ::mi355_host::c_type_name<mat_value_t>::get(),
                                                  ::mi355_host::c_type_name<vec_x_value_t>::get(),
                                                  ::mi355_host::c_type_name<vec_y_value_t>::get());
        if (st != MI355_SPMV_OK) std::fprintf(stderr, "%s\n", mi355_spmv_functor_compile_log());
        MI355_CHECK(st);
        return f;
    }();
    Timer::kernel_start();
    MI355_CHECK(mi355_spmv_functor_spmv(compiled, (int32_t)n_rows, (int32_t)n_cols, (int64_t)nnz, Ap,
                                        reinterpret_cast<const int32_t*>(Aj), Ax, x, y, /*stream=*/nullptr));
    MI355_CHECK(mi355_spmv_stream_synchronize(/*stream=*/nullptr));
    Timer::kernel_stop();
}

// the ordinary (+, *) as a functor's text: what run_kind falls back on for the type mixes the tuned kernels are not
// built for, and the harness's "hip_functor" row (host/spmv.h)
MI355_FUNCTOR(TimesThenPlus,
    template <typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
    struct TimesThenPlus {
        __host__ __device__ __forceinline__ static vec_y_value_t initialize() { return vec_y_value_t(0); }
        __host__ __device__ __forceinline__ static vec_y_value_t combine(const mat_value_t& nonzero, const vec_x_value_t& x) {
            return vec_y_value_t(nonzero * x);     // (the product in the promoted type, then the y type: merge_genl.cuh:29-31)
        }
        __host__ __device__ __forceinline__ static vec_y_value_t reduce(const vec_y_value_t& lhs, const vec_y_value_t& rhs) { return lhs + rhs; }
    };)

namespace mi355_host {
template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
void run_general(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
                 const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    ::SpMV_hip_functor<TimesThenPlus_text>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);
}
}  // namespace mi355_host
