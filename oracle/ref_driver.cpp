// oracle/ref_driver.cpp
//
// TEST INFRASTRUCTURE ONLY.  Thin extern "C" driver around the REFERENCE'S OWN
// headers, compiled where they lie (-I/root/reference/include) by
// oracle/Makefile into oracle/_ref/libspmv_ref.so.  No reference source is
// copied into this repository; this file only instantiates and calls:
//   LoadCoo / ToCsr      include/load.hpp:268-408, :420-474
//   SpMV_cpu_navie       include/spmv/cpu_navie.hpp:5-17
//   SpMV_genl_cpu_navie  include/spmv/cpu_navie.hpp:20-34 (with the functors defined below)
// The two standard headers below come first because load.hpp uses std::cerr and
// std::numeric_limits without including them (load.hpp:279, :302).
//
// LoadCoo calls exit(1) / throws on malformed input; ref_load() is therefore
// only ever given well-formed files (error paths are pinned by reading, see
// tests/test_oracle.py).

#include <iostream>
#include <limits>
#include <cstdint>
#include <cstring>

#include "load.hpp"
#include "spmv/cpu_navie.hpp"

namespace {
// functor_t instances for the reference's SpMV_genl_cpu_navie<functor_t> (cpu_navie.hpp:20-34).
// The reference's only instance, MergeFunctor (merge_genl.cuh:19-38: initialize 0, combine
// nonzero*x, reduce a+b), sits in a CUDA/CUB header and cannot be included; PlusTimes restates
// its three one-liners, the other two are the semirings the C ABI adds.
// (+-infinity of a value type: the extreme integers for int — numeric_limits<int>::infinity() is 0)
template <typename V> struct Lim {
    static V hi() { return std::numeric_limits<V>::infinity(); }
    static V lo() { return -std::numeric_limits<V>::infinity(); }
};
template <> struct Lim<int> {
    static int hi() { return std::numeric_limits<int>::max(); }
    static int lo() { return std::numeric_limits<int>::lowest(); }
};
template <typename V> struct PlusTimes {
    static V initialize() { return V(0); }
    static V combine(V a, V x) { return a * x; }
    static V reduce(V u, V v) { return u + v; }
};
template <typename V> struct MinPlus {
    static V initialize() { return Lim<V>::hi(); }
    static V combine(V a, V x) { return a + x; }
    static V reduce(V u, V v) { return v < u ? v : u; }
};
template <typename V> struct MaxTimes {
    static V initialize() { return Lim<V>::lo(); }
    static V combine(V a, V x) { return a * x; }
    static V reduce(V u, V v) { return u < v ? v : u; }
};
template <typename V> struct MaxPlus {
    static V initialize() { return Lim<V>::lo(); }
    static V combine(V a, V x) { return a + x; }
    static V reduce(V u, V v) { return u < v ? v : u; }
};
template <typename V> struct OrAnd {
    static V initialize() { return V(0); }
    static V combine(V a, V x) { return (a != V(0) && x != V(0)) ? V(1) : V(0); }
    static V reduce(V u, V v) { return (u != V(0) || v != V(0)) ? V(1) : V(0); }
};
template <typename off_t, typename val_t>
struct Held {
    csr_t<int, off_t, val_t> csr;
};
}  // namespace

extern "C" {

#define REF_TYPED(SUF, OFF, VAL)                                                               \
    void* ref_load_##SUF(const char* path) {                                                   \
        try {                                                                                  \
            auto* h = new Held<OFF, VAL>();                                                    \
            h->csr = ToCsr(LoadCoo<int, OFF, VAL>(std::string(path)));                         \
            return h;                                                                          \
        } catch (...) {                                                                        \
            return nullptr;                                                                    \
        }                                                                                      \
    }                                                                                          \
    void ref_dims_##SUF(void* hp, int64_t* n_rows, int64_t* n_cols, int64_t* nnz) {            \
        auto* h = static_cast<Held<OFF, VAL>*>(hp);                                            \
        *n_rows = h->csr.number_of_rows;                                                       \
        *n_cols = h->csr.number_of_columns;                                                    \
        *nnz = h->csr.number_of_nonzeros;                                                      \
    }                                                                                          \
    void ref_copy_##SUF(void* hp, OFF* Ap, int* Aj, VAL* Ax) {                                 \
        auto* h = static_cast<Held<OFF, VAL>*>(hp);                                            \
        memcpy(Ap, h->csr.row_offsets.data(), h->csr.row_offsets.size() * sizeof(OFF));        \
        memcpy(Aj, h->csr.column_indices.data(), h->csr.column_indices.size() * sizeof(int));  \
        memcpy(Ax, h->csr.nonzero_values.data(), h->csr.nonzero_values.size() * sizeof(VAL));  \
    }                                                                                          \
    void ref_free_##SUF(void* hp) { delete static_cast<Held<OFF, VAL>*>(hp); }                 \
    void ref_spmv_genl_cpu_##SUF(int semiring, int n_rows, int n_cols, OFF nnz, const OFF* Ap,  \
                                 const int* Aj, const VAL* Ax, const VAL* x, VAL* y) {         \
        if (semiring == 0)                                                                     \
            SpMV_genl_cpu_navie<PlusTimes<VAL>, int, OFF, VAL, VAL, VAL>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y); \
        else if (semiring == 1)                                                                \
            SpMV_genl_cpu_navie<MinPlus<VAL>, int, OFF, VAL, VAL, VAL>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);   \
        else if (semiring == 2)                                                                \
            SpMV_genl_cpu_navie<MaxTimes<VAL>, int, OFF, VAL, VAL, VAL>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);  \
        else if (semiring == 3)                                                                \
            SpMV_genl_cpu_navie<MaxPlus<VAL>, int, OFF, VAL, VAL, VAL>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);   \
        else                                                                                   \
            SpMV_genl_cpu_navie<OrAnd<VAL>, int, OFF, VAL, VAL, VAL>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);     \
    }                                                                                          \
    void ref_spmv_cpu_##SUF(int n_rows, int n_cols, OFF nnz, const OFF* Ap, const int* Aj,     \
                            const VAL* Ax, const VAL* x, VAL* y) {                             \
        SpMV_cpu_navie<int, OFF, VAL, VAL, VAL>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);        \
    }

// The reference's serial check with an fp32 MATRIX under fp64 vectors (its template keeps the three value types
// apart, include/spmv.h:29-34): pins the oracle's restatement of that case.
#define REF_MIXED(SUF, OFF)                                                                    \
    void ref_spmv_cpu_mixed_##SUF(int n_rows, int n_cols, OFF nnz, const OFF* Ap, const int* Aj, \
                                  const float* Ax, const double* x, double* y) {               \
        SpMV_cpu_navie<int, OFF, float, double, double>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y); \
    }
REF_MIXED(i32, int)
REF_MIXED(i64, long long)

// the reference's generalized serial check on INTEGER values (its template takes any arithmetic type): pins the
// oracle's integer restatement (values kept small by the callers: signed overflow is undefined in the reference's code)
#define REF_GENL_INT(SUF, OFF)                                                                 \
    void ref_spmv_genl_cpu_##SUF(int semiring, int n_rows, int n_cols, OFF nnz, const OFF* Ap, \
                                 const int* Aj, const int* Ax, const int* x, int* y) {         \
        if (semiring == 0)                                                                     \
            SpMV_genl_cpu_navie<PlusTimes<int>, int, OFF, int, int, int>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y); \
        else if (semiring == 1)                                                                \
            SpMV_genl_cpu_navie<MinPlus<int>, int, OFF, int, int, int>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);   \
        else if (semiring == 2)                                                                \
            SpMV_genl_cpu_navie<MaxTimes<int>, int, OFF, int, int, int>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);  \
        else if (semiring == 3)                                                                \
            SpMV_genl_cpu_navie<MaxPlus<int>, int, OFF, int, int, int>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);   \
        else                                                                                   \
            SpMV_genl_cpu_navie<OrAnd<int>, int, OFF, int, int, int>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);     \
    }
REF_GENL_INT(i32_i32, int)
REF_GENL_INT(i64_i32, long long)

REF_TYPED(i32_f32, int, float)
REF_TYPED(i32_f64, int, double)
REF_TYPED(i64_f32, long long, float)
REF_TYPED(i64_f64, long long, double)

}  // extern "C"
