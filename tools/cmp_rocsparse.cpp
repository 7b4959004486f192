// tools/cmp_rocsparse.cpp — COMPARISON COLUMN ONLY: rocSPARSE csrmv timed on device
// operands the caller already holds, for scripts/gpu_vendor_cmp.py (ctypes).  Host C++ only:
//   g++ -std=c++17 -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -DMI355_WITH_ROCSPARSE -I/opt/rocm/include \
//       tools/cmp_rocsparse.cpp -o tools/bin/libcmp_rocsparse.so -L/opt/rocm/lib -lrocsparse -lamdhip64
// The engine never links this.
#include "../spmv-samples_amd/host/spmv/rocsparse_cmp.hpp"

namespace {
template <typename val_t>
int time_csrmv(int n_rows, int n_cols, int nnz, const int* Ap, const int* Aj, const void* Ax, const void* x, void* y,
               int analyse, int warmup, int iters, double* us_out, double* analysis_ms_out) {
    hipEvent_t a, b;
    ROCSPARSE_CMP_CHECK(hipEventCreate(&a));
    ROCSPARSE_CMP_CHECK(hipEventCreate(&b));
    ROCSPARSE_CMP_CHECK(hipEventRecord(a, nullptr));
    rocsparse_cmp::Csrmv<val_t> op(n_rows, n_cols, nnz, Ap, Aj, static_cast<const val_t*>(Ax), analyse != 0);
    ROCSPARSE_CMP_CHECK(hipEventRecord(b, nullptr));
    ROCSPARSE_CMP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    ROCSPARSE_CMP_CHECK(hipEventElapsedTime(&ms, a, b));
    *analysis_ms_out = ms;
    for (int i = 0; i < warmup; ++i) op.run(static_cast<const val_t*>(x), static_cast<val_t*>(y));
    ROCSPARSE_CMP_CHECK(hipDeviceSynchronize());
    ROCSPARSE_CMP_CHECK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) op.run(static_cast<const val_t*>(x), static_cast<val_t*>(y));
    ROCSPARSE_CMP_CHECK(hipEventRecord(b, nullptr));
    ROCSPARSE_CMP_CHECK(hipEventSynchronize(b));
    ROCSPARSE_CMP_CHECK(hipEventElapsedTime(&ms, a, b));
    *us_out = double(ms) * 1e3 / iters;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return 0;
}
}  // namespace

// val_type: 0 = float, 1 = double.  All pointers are device pointers; y is overwritten.
extern "C" int cmp_rocsparse_csrmv(int val_type, int n_rows, int n_cols, int nnz, const int* Ap, const int* Aj,
                                   const void* Ax, const void* x, void* y, int analyse, int warmup, int iters,
                                   double* us_out, double* analysis_ms_out) {
    if (val_type == 0)
        return time_csrmv<float>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y, analyse, warmup, iters, us_out, analysis_ms_out);
    return time_csrmv<double>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y, analyse, warmup, iters, us_out, analysis_ms_out);
}
