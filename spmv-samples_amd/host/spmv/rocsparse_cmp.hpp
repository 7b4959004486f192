// spmv/rocsparse_cmp.hpp — COMPARISON COLUMN ONLY.
//
// The vendor library's CSR SpMV as one more kind, the MI355X analogue of the reference's
// `cusparse` kind (include/spmv/cusparse.cuh:38-89: handle + descriptors + buffer built inside the
// call, Timer::kernel_* around the one SpMV call, alpha = 1, beta = 0).  It exists so that the
// harness tables and scripts/gpu_vendor_cmp.py can print a vendor column next to hip_vector /
// hip_merge / hip_light.  The engine (libmi355spmv.so) never links or calls rocSPARSE; this header is
// compiled only when MI355_WITH_ROCSPARSE is defined and only into the harness / comparison tool.
//
// rocSPARSE's csrmv takes 32-bit row offsets (rocsparse_int); with offset_t = int64 the kind reports
// that and exits, the way the reference's library kinds exit on an unsupported call
// (cusparse.cuh:13-21).
#pragma once
#ifdef MI355_WITH_ROCSPARSE

#include <hip/hip_runtime_api.h>
#include <rocsparse/rocsparse.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../timer.hpp"

namespace rocsparse_cmp {

inline void check(rocsparse_status s, const char* what, const char* file, int line) {
    if (s != rocsparse_status_success) {
        std::fprintf(stderr, "rocSPARSE error at %s:%d status=%d \"%s\"\n", file, line, int(s), what);
        std::exit(EXIT_FAILURE);
    }
}
inline void check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        std::fprintf(stderr, "HIP error at %s:%d code=%d(%s) \"%s\"\n", file, line, int(e), hipGetErrorString(e), what);
        std::abort();
    }
}
#define ROCSPARSE_CMP_CHECK(expr) ::rocsparse_cmp::check((expr), #expr, __FILE__, __LINE__)

// One matrix bound to one handle.  analyse = true runs rocsparse_Xcsrmv_analysis, which switches
// csrmv to its adaptive algorithm; false leaves the row-split ("stream") algorithm.
template <typename val_t>
class Csrmv {
    static_assert(std::is_same<val_t, float>::value || std::is_same<val_t, double>::value, "float or double");

public:
    Csrmv(int n_rows, int n_cols, int nnz, const int* Ap, const int* Aj, const val_t* Ax, bool analyse)
        : n_rows_(n_rows), n_cols_(n_cols), nnz_(nnz), Ap_(Ap), Aj_(Aj), Ax_(Ax) {
        ROCSPARSE_CMP_CHECK(rocsparse_create_handle(&handle_));
        ROCSPARSE_CMP_CHECK(rocsparse_set_pointer_mode(handle_, rocsparse_pointer_mode_host));
        ROCSPARSE_CMP_CHECK(rocsparse_create_mat_descr(&descr_));
        ROCSPARSE_CMP_CHECK(rocsparse_create_mat_info(&info_));
        if (analyse) {
            if constexpr (std::is_same<val_t, float>::value)
                ROCSPARSE_CMP_CHECK(rocsparse_scsrmv_analysis(handle_, rocsparse_operation_none, n_rows, n_cols, nnz,
                                                              descr_, Ax, Ap, Aj, info_));
            else
                ROCSPARSE_CMP_CHECK(rocsparse_dcsrmv_analysis(handle_, rocsparse_operation_none, n_rows, n_cols, nnz,
                                                              descr_, Ax, Ap, Aj, info_));
            ROCSPARSE_CMP_CHECK(hipDeviceSynchronize());
        }
        analysed_ = analyse;
    }
    Csrmv(const Csrmv&) = delete;
    Csrmv& operator=(const Csrmv&) = delete;
    ~Csrmv() {
        if (analysed_) rocsparse_csrmv_clear(handle_, info_);
        rocsparse_destroy_mat_info(info_);
        rocsparse_destroy_mat_descr(descr_);
        rocsparse_destroy_handle(handle_);
    }

    // y = A x on the null stream (asynchronous, like every launch)
    void run(const val_t* x, val_t* y) {
        const val_t alpha = 1, beta = 0;
        if constexpr (std::is_same<val_t, float>::value)
            ROCSPARSE_CMP_CHECK(rocsparse_scsrmv(handle_, rocsparse_operation_none, n_rows_, n_cols_, nnz_, &alpha,
                                                 descr_, Ax_, Ap_, Aj_, info_, x, &beta, y));
        else
            ROCSPARSE_CMP_CHECK(rocsparse_dcsrmv(handle_, rocsparse_operation_none, n_rows_, n_cols_, nnz_, &alpha,
                                                 descr_, Ax_, Ap_, Aj_, info_, x, &beta, y));
    }

private:
    int n_rows_, n_cols_, nnz_;
    const int* Ap_;
    const int* Aj_;
    const val_t* Ax_;
    rocsparse_handle handle_ = nullptr;
    rocsparse_mat_descr descr_ = nullptr;
    rocsparse_mat_info info_ = nullptr;
    bool analysed_ = false;
};

template <bool ANALYSE, typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t,
          typename vec_y_value_t>
void run_kind(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
              const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    static_assert(std::is_same<mat_value_t, vec_x_value_t>::value && std::is_same<mat_value_t, vec_y_value_t>::value,
                  "rocsparse kind: A, x and y share one value type");
    if constexpr (sizeof(offset_t) != 4 || sizeof(index_t) != 4) {
        std::fprintf(stderr, "rocsparse kind: csrmv takes 32-bit row offsets; offset_t is %zu bytes\n", sizeof(offset_t));
        std::exit(EXIT_FAILURE);
    } else {
        Csrmv<mat_value_t> op(int(n_rows), int(n_cols), int(nnz), reinterpret_cast<const int*>(Ap),
                              reinterpret_cast<const int*>(Aj), Ax, ANALYSE);
        Timer::kernel_start();
        op.run(x, y);
        ROCSPARSE_CMP_CHECK(hipDeviceSynchronize());
        Timer::kernel_stop();
    }
}

}  // namespace rocsparse_cmp

// "rocsparse": csrmv after analysis (adaptive);  "rocsparse_stream": csrmv without analysis (row split)
template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
void SpMV_rocsparse(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
                    const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    rocsparse_cmp::run_kind<true>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);
}
template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
void SpMV_rocsparse_stream(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
                           const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    rocsparse_cmp::run_kind<false>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);
}

#endif  // MI355_WITH_ROCSPARSE
