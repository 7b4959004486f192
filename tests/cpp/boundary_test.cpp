// tests/cpp/boundary_test.cpp — exercises the C++ operator boundary
// (spmv-samples_amd/host/spmv.h) the way the reference's harness does
// (main.cu:48-97): device arrays owned by the caller, SpMV(kind, ...) per label,
// copy y back, compare with a serial CSR loop.  TEST CODE: the serial loop below is
// the checker, not a product path.
//
//   boundary_test                 all labels x {i32,i64} x {f32,f64}; exit 0 when all pass
//   boundary_test --bad-label     calls SpMV("no_such_kind", ...): must print the
//                                 reference's message and exit(EXIT_FAILURE) (spmv.h:46-47)
//   boundary_test --functor       functors of the caller's own through SpMV_hip_functor (compiled at run time)
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../spmv-samples_amd/host/spmv.h"

#define HIP_OK(e)                                                                   \
    do {                                                                            \
        hipError_t _e = (e);                                                        \
        if (_e != hipSuccess) {                                                     \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
            std::exit(2);                                                           \
        }                                                                           \
    } while (0)

static unsigned long long lcg(unsigned long long& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return s >> 33;
}

template <typename offset_t, typename value_t>
static int run_combo(const char* combo, int n_rows, int n_cols, int max_len) {
    using index_t = int;
    unsigned long long seed = 12345;
    std::vector<offset_t> Ap(n_rows + 1, 0);
    std::vector<index_t> Aj;
    std::vector<value_t> Ax;
    for (int r = 0; r < n_rows; ++r) {
        int len = (r % 7 == 3) ? 0 : int(lcg(seed) % (unsigned)(max_len + 1));
        if (r == n_rows / 2) len = 40 * max_len;  // one long row
        for (int k = 0; k < len; ++k) {
            Aj.push_back(index_t(lcg(seed) % (unsigned)n_cols));
            Ax.push_back(value_t(double(lcg(seed) % 2001) / 1000.0 - 1.0));
        }
        Ap[r + 1] = offset_t(Aj.size());
    }
    const offset_t nnz = offset_t(Aj.size());
    std::vector<value_t> x(n_cols);
    for (int c = 0; c < n_cols; ++c) x[c] = value_t(double(lcg(seed) % 2001) / 1000.0 - 1.0);

    std::vector<double> ref(n_rows), mag(n_rows);
    for (int r = 0; r < n_rows; ++r) {
        double s = 0, a = 0;
        for (offset_t k = Ap[r]; k < Ap[r + 1]; ++k) {
            double p = double(Ax[k]) * double(x[Aj[k]]);
            s += p;
            a += std::fabs(p);
        }
        ref[r] = s;
        mag[r] = a;
    }

    offset_t* dAp; index_t* dAj; value_t *dAx, *dX, *dY;
    HIP_OK(hipMalloc((void**)&dAp, (n_rows + 1) * sizeof(offset_t)));
    HIP_OK(hipMalloc((void**)&dAj, (size_t(nnz) + 1) * sizeof(index_t)));
    HIP_OK(hipMalloc((void**)&dAx, (size_t(nnz) + 1) * sizeof(value_t)));
    HIP_OK(hipMalloc((void**)&dX, n_cols * sizeof(value_t)));
    HIP_OK(hipMalloc((void**)&dY, n_rows * sizeof(value_t)));
    HIP_OK(hipMemcpy(dAp, Ap.data(), (n_rows + 1) * sizeof(offset_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAj, Aj.data(), size_t(nnz) * sizeof(index_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAx, Ax.data(), size_t(nnz) * sizeof(value_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dX, x.data(), n_cols * sizeof(value_t), hipMemcpyHostToDevice));

    const double eps = std::is_same<value_t, float>::value ? std::ldexp(1.0, -24) : std::ldexp(1.0, -53);
    int failures = 0;
    std::vector<value_t> y(n_rows);
#define X(label, func)                                                                                  \
    {                                                                                                   \
        /* poison y: a kind that skips a row must not inherit the previous kind's value (quirk 5) */    \
        std::vector<value_t> poison(n_rows, std::numeric_limits<value_t>::quiet_NaN());                 \
        HIP_OK(hipMemcpy(dY, poison.data(), n_rows * sizeof(value_t), hipMemcpyHostToDevice));          \
        SpMV<index_t, offset_t, value_t, value_t, value_t>(label, n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY); \
        HIP_OK(hipMemcpy(y.data(), dY, n_rows * sizeof(value_t), hipMemcpyDeviceToHost));               \
        int bad = 0;                                                                                    \
        double worst = 0;                                                                               \
        for (int r = 0; r < n_rows; ++r) {                                                              \
            const double len = double(Ap[r + 1] - Ap[r]);                                               \
            const double err = std::fabs(double(y[r]) - ref[r]);                                        \
            const double bound = (len + 2) * eps * mag[r];                                              \
            if (!(err <= bound)) ++bad;                                                                 \
            if (err > worst) worst = err;                                                               \
        }                                                                                               \
        std::printf("[%-14s] %-8s rows=%d nnz=%lld max_err=%.3e bad_rows=%d total=%lldus kernel=%lldus\n", \
                    label, combo, n_rows, (long long)nnz, worst, bad, (long long)Timer::total_cost(),   \
                    (long long)Timer::kernel_cost());                                                   \
        failures += bad ? 1 : 0;                                                                        \
        if (Timer::kernel_cost() > Timer::total_cost()) { std::printf("timer order violated\n"); ++failures; } \
    }
    SPMV_KINDS
#undef X
    mi355_host::dist_release();   // the multi-GPU kinds keep a handle on dAp / dAj between calls
    HIP_OK(hipFree(dAp)); HIP_OK(hipFree(dAj)); HIP_OK(hipFree(dAx)); HIP_OK(hipFree(dX)); HIP_OK(hipFree(dY));
    return failures;
}

// fp32 matrix under fp64 vectors through the template boundary: SpMV<int, int, float, double, double>("hip_merge", ...)
static int run_mixed() {
    const int n_rows = 4001, n_cols = 3000;
    unsigned long long seed = 777;
    std::vector<int> Ap(n_rows + 1, 0), Aj;
    std::vector<float> Ax;
    for (int r = 0; r < n_rows; ++r) {
        const int len = (r == 1000) ? 9000 : int(lcg(seed) % 20u);
        for (int k = 0; k < len; ++k) {
            Aj.push_back(int(lcg(seed) % (unsigned)n_cols));
            Ax.push_back(float(double(lcg(seed) % 2001) / 1000.0 - 1.0));
        }
        Ap[r + 1] = int(Aj.size());
    }
    const int nnz = int(Aj.size());
    std::vector<double> x(n_cols), ref(n_rows), mag(n_rows), y(n_rows);
    for (int c = 0; c < n_cols; ++c) x[c] = double(lcg(seed) % 200001) / 100000.0 - 1.0;
    for (int r = 0; r < n_rows; ++r) {
        double s = 0, a = 0;
        for (int k = Ap[r]; k < Ap[r + 1]; ++k) { const double p = double(Ax[k]) * x[Aj[k]]; s += p; a += std::fabs(p); }
        ref[r] = s; mag[r] = a;
    }
    int* dAp; int* dAj; float* dAx; double *dX, *dY;
    HIP_OK(hipMalloc((void**)&dAp, (n_rows + 1) * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dAj, (size_t(nnz) + 4) * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dAx, (size_t(nnz) + 4) * sizeof(float)));
    HIP_OK(hipMalloc((void**)&dX, n_cols * sizeof(double)));
    HIP_OK(hipMalloc((void**)&dY, n_rows * sizeof(double)));
    HIP_OK(hipMemcpy(dAp, Ap.data(), (n_rows + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAj, Aj.data(), size_t(nnz) * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAx, Ax.data(), size_t(nnz) * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dX, x.data(), n_cols * sizeof(double), hipMemcpyHostToDevice));
    int failures = 0;
    for (const char* label : {"hip_merge", "hip_merge_genl"}) {
        std::vector<double> poison(n_rows, std::numeric_limits<double>::quiet_NaN());
        HIP_OK(hipMemcpy(dY, poison.data(), n_rows * sizeof(double), hipMemcpyHostToDevice));
        if (std::strcmp(label, "hip_merge") == 0) SpMV_hip_merge<int, int, float, double, double>(n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
        else SpMV_hip_merge_generalized<mi355_host::PlusTimes, int, int, float, double, double>(n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
        HIP_OK(hipMemcpy(y.data(), dY, n_rows * sizeof(double), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int r = 0; r < n_rows; ++r)
            if (!(std::fabs(y[r] - ref[r]) <= (double(Ap[r + 1] - Ap[r]) + 2) * std::ldexp(1.0, -53) * mag[r])) ++bad;
        std::printf("[%-14s] f32mat_f64vec rows=%d nnz=%d bad_rows=%d\n", label, n_rows, nnz, bad);
        failures += bad ? 1 : 0;
    }
    HIP_OK(hipFree(dAp)); HIP_OK(hipFree(dAj)); HIP_OK(hipFree(dAx)); HIP_OK(hipFree(dX)); HIP_OK(hipFree(dY));
    return failures;
}

// integer values through the template boundary: SpMV_hip_merge_generalized<MinPlus, int, long long, int, int, int>
// (the reference's generalized kind is a template over the value types and the functor) — exact against a serial loop
static int run_integer() {
    const int n_rows = 5003, n_cols = 800;
    unsigned long long seed = 4242;
    std::vector<long long> Ap(n_rows + 1, 0);
    std::vector<int> Aj, Ax;
    for (int r = 0; r < n_rows; ++r) {
        const int len = (r == 77) ? 7000 : int(lcg(seed) % 15u);
        for (int k = 0; k < len; ++k) {
            Aj.push_back(int(lcg(seed) % (unsigned)n_cols));
            Ax.push_back(int(lcg(seed) % 41u) - 20);
        }
        Ap[r + 1] = (long long)Aj.size();
    }
    const long long nnz = (long long)Aj.size();
    std::vector<int> x(n_cols), y(n_rows);
    for (int c = 0; c < n_cols; ++c) x[c] = int(lcg(seed) % 61u) - 30;
    long long* dAp; int *dAj, *dAx, *dX, *dY;
    HIP_OK(hipMalloc((void**)&dAp, (n_rows + 1) * sizeof(long long)));
    HIP_OK(hipMalloc((void**)&dAj, (size_t(nnz) + 4) * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dAx, (size_t(nnz) + 4) * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dX, n_cols * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dY, n_rows * sizeof(int)));
    HIP_OK(hipMemcpy(dAp, Ap.data(), (n_rows + 1) * sizeof(long long), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAj, Aj.data(), size_t(nnz) * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAx, Ax.data(), size_t(nnz) * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dX, x.data(), n_cols * sizeof(int), hipMemcpyHostToDevice));
    int failures = 0;
    for (int which = 0; which < 2; ++which) {          // 0: (+, *), 1: (min, +)
        std::vector<int> poison(n_rows, 123456789);
        HIP_OK(hipMemcpy(dY, poison.data(), n_rows * sizeof(int), hipMemcpyHostToDevice));
        if (which == 0) SpMV_hip_merge_generalized<mi355_host::PlusTimes, int, long long, int, int, int>(n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
        else SpMV_hip_merge_generalized<mi355_host::MinPlus, int, long long, int, int, int>(n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
        HIP_OK(hipMemcpy(y.data(), dY, n_rows * sizeof(int), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int r = 0; r < n_rows; ++r) {
            long long s = which == 0 ? 0 : std::numeric_limits<int>::max();
            for (long long k = Ap[r]; k < Ap[r + 1]; ++k) {
                if (which == 0) s += (long long)Ax[k] * x[Aj[k]];
                else s = std::min<long long>(s, (long long)Ax[k] + x[Aj[k]]);
            }
            if (y[r] != int(s)) ++bad;
        }
        std::printf("[genl on int    ] %s rows=%d nnz=%lld bad_rows=%d\n", which == 0 ? "(+,*)" : "(min,+)", n_rows, nnz, bad);
        failures += bad ? 1 : 0;
    }
    HIP_OK(hipFree(dAp)); HIP_OK(hipFree(dAj)); HIP_OK(hipFree(dAx)); HIP_OK(hipFree(dX)); HIP_OK(hipFree(dY));
    return failures;
}

// Functors of the caller's own, written ONCE (MI355_FUNCTOR, host/spmv/mi355.hpp): the host compiles the definition for
// the serial twin below, the device gets its text (hiprtc).  All five types of include/spmv.h:29-34 differ.
MI355_FUNCTOR(ManhattanTerm,
    template <typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
    struct ManhattanTerm {     // y[r] = sum_k |Ax[k] - x[Aj[k]]|
        __host__ __device__ __forceinline__ static vec_y_value_t initialize() { return vec_y_value_t(0); }
        __host__ __device__ __forceinline__ static vec_y_value_t combine(const mat_value_t& nonzero, const vec_x_value_t& x) {
            const vec_y_value_t d = vec_y_value_t(nonzero) - vec_y_value_t(x);
            return d < vec_y_value_t(0) ? -d : d;
        }
        __host__ __device__ __forceinline__ static vec_y_value_t reduce(const vec_y_value_t& lhs, const vec_y_value_t& rhs) { return lhs + rhs; }
    };)
MI355_FUNCTOR(Bottleneck,
    template <typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
    struct Bottleneck {        // y[r] = max_k min(Ax[k], x[Aj[k]]): the widest path through one more edge
        __host__ __device__ __forceinline__ static vec_y_value_t initialize() { return vec_y_value_t(-1.0e30); }
        __host__ __device__ __forceinline__ static vec_y_value_t combine(const mat_value_t& nonzero, const vec_x_value_t& x) {
            return vec_y_value_t(nonzero) < vec_y_value_t(x) ? vec_y_value_t(nonzero) : vec_y_value_t(x);
        }
        __host__ __device__ __forceinline__ static vec_y_value_t reduce(const vec_y_value_t& lhs, const vec_y_value_t& rhs) { return lhs < rhs ? rhs : lhs; }
    };)
MI355_FUNCTOR(CountPositive,
    struct CountPositive {     // a plain struct: y[r] = how many products Ax[k] * x[Aj[k]] are positive
        __host__ __device__ static long long initialize() { return 0; }
        __host__ __device__ static long long combine(const double& nonzero, const float& x) { return nonzero * x > 0 ? 1 : 0; }
        __host__ __device__ static long long reduce(const long long& lhs, const long long& rhs) { return lhs + rhs; }
    };)

// the serial fold every generalized kind is checked against (what cpu_navie.hpp:20-34 computes), in test code
template <typename functor_t, typename offset_t, typename mat_t, typename x_t, typename y_t>
static void fold_rows(int n_rows, const offset_t* Ap, const int* Aj, const mat_t* Ax, const x_t* x, y_t* y) {
    for (int r = 0; r < n_rows; ++r) {
        y_t acc = functor_t::initialize();
        for (offset_t k = Ap[r]; k < Ap[r + 1]; ++k) acc = functor_t::reduce(acc, functor_t::combine(Ax[k], x[Aj[k]]));
        y[r] = acc;
    }
}

template <typename text_t, bool is_template, typename functor_t, typename offset_t, typename mat_t, typename x_t, typename y_t>
static int run_functor_case(const char* what, int n_rows, int n_cols, int max_len, int hub_len) {
    unsigned long long seed = 777 + (unsigned long long)n_rows;
    std::vector<offset_t> Ap(n_rows + 1, 0);
    std::vector<int> Aj;
    std::vector<mat_t> Ax;
    for (int r = 0; r < n_rows; ++r) {
        int len = (r % 11 == 5) ? 0 : int(lcg(seed) % (unsigned)(max_len + 1));
        if (r == n_rows / 3) len = hub_len;               // one row for the whole-wave kernel
        for (int k = 0; k < len; ++k) {
            Aj.push_back(int(lcg(seed) % (unsigned)n_cols));
            Ax.push_back(mat_t(int(lcg(seed) % 201u) - 100));   // integer-valued: sums are exact in any order
        }
        Ap[r + 1] = offset_t(Aj.size());
    }
    const offset_t nnz = offset_t(Aj.size());
    std::vector<x_t> x(n_cols);
    for (int c = 0; c < n_cols; ++c) x[c] = x_t(int(lcg(seed) % 151u) - 75);
    std::vector<y_t> want(n_rows), got(n_rows);
    fold_rows<functor_t>(n_rows, Ap.data(), Aj.data(), Ax.data(), x.data(), want.data());
    offset_t* dAp; int* dAj; mat_t* dAx; x_t* dX; y_t* dY;
    HIP_OK(hipMalloc((void**)&dAp, (n_rows + 1) * sizeof(offset_t)));
    HIP_OK(hipMalloc((void**)&dAj, (size_t(nnz) + 4) * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dAx, (size_t(nnz) + 4) * sizeof(mat_t)));
    HIP_OK(hipMalloc((void**)&dX, n_cols * sizeof(x_t)));
    HIP_OK(hipMalloc((void**)&dY, n_rows * sizeof(y_t)));
    HIP_OK(hipMemcpy(dAp, Ap.data(), (n_rows + 1) * sizeof(offset_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAj, Aj.data(), size_t(nnz) * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAx, Ax.data(), size_t(nnz) * sizeof(mat_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dX, x.data(), n_cols * sizeof(x_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(dY, 0x5a, n_rows * sizeof(y_t)));
    int bad = 0;
    for (int call = 0; call < 2; ++call) {                 // (the second call reuses the compiled code object)
        SpMV_hip_functor<text_t, is_template>(n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
        HIP_OK(hipMemcpy(got.data(), dY, n_rows * sizeof(y_t), hipMemcpyDeviceToHost));
        for (int r = 0; r < n_rows; ++r) bad += got[r] == want[r] ? 0 : 1;
    }
    std::printf("[functor        ] %s rows=%d nnz=%lld bad_rows=%d kernel_us=%lld\n", what, n_rows, (long long)nnz, bad,
                (long long)Timer::kernel_cost());
    HIP_OK(hipFree(dAp)); HIP_OK(hipFree(dAj)); HIP_OK(hipFree(dAx)); HIP_OK(hipFree(dX)); HIP_OK(hipFree(dY));
    return bad ? 1 : 0;
}

// Every kind accepts any mix of float / double / int / long long for the matrix, x and y, as the reference's templates do
// (spmv.h:29-34): what the tuned kernels are not built for runs (+, *) on the general path.  Integer-valued data: exact.
template <typename offset_t, typename mat_t, typename x_t, typename y_t, typename Call>
static int run_untuned_mix(const char* what, Call call) {
    const int n_rows = 5003, n_cols = 1200;
    unsigned long long seed = 31337;
    std::vector<offset_t> Ap(n_rows + 1, 0);
    std::vector<int> Aj;
    std::vector<mat_t> Ax;
    for (int r = 0; r < n_rows; ++r) {
        const int len = (r % 9 == 2) ? 0 : int(lcg(seed) % 33u);
        for (int k = 0; k < len; ++k) {
            Aj.push_back(int(lcg(seed) % (unsigned)n_cols));
            Ax.push_back(mat_t(int(lcg(seed) % 21u) - 10));
        }
        Ap[r + 1] = offset_t(Aj.size());
    }
    const offset_t nnz = offset_t(Aj.size());
    std::vector<x_t> x(n_cols);
    for (int c = 0; c < n_cols; ++c) x[c] = x_t(int(lcg(seed) % 15u) - 7);
    std::vector<y_t> want(n_rows), got(n_rows);
    fold_rows<TimesThenPlus<mat_t, x_t, y_t>>(n_rows, Ap.data(), Aj.data(), Ax.data(), x.data(), want.data());
    offset_t* dAp; int* dAj; mat_t* dAx; x_t* dX; y_t* dY;
    HIP_OK(hipMalloc((void**)&dAp, (n_rows + 1) * sizeof(offset_t)));
    HIP_OK(hipMalloc((void**)&dAj, (size_t(nnz) + 4) * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dAx, (size_t(nnz) + 4) * sizeof(mat_t)));
    HIP_OK(hipMalloc((void**)&dX, n_cols * sizeof(x_t)));
    HIP_OK(hipMalloc((void**)&dY, n_rows * sizeof(y_t)));
    HIP_OK(hipMemcpy(dAp, Ap.data(), (n_rows + 1) * sizeof(offset_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAj, Aj.data(), size_t(nnz) * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dAx, Ax.data(), size_t(nnz) * sizeof(mat_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dX, x.data(), n_cols * sizeof(x_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(dY, 0x5a, n_rows * sizeof(y_t)));
    call(n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
    HIP_OK(hipMemcpy(got.data(), dY, n_rows * sizeof(y_t), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int r = 0; r < n_rows; ++r) bad += got[r] == want[r] ? 0 : 1;
    std::printf("[untuned mix    ] %s rows=%d nnz=%lld bad_rows=%d\n", what, n_rows, (long long)nnz, bad);
    HIP_OK(hipFree(dAp)); HIP_OK(hipFree(dAj)); HIP_OK(hipFree(dAx)); HIP_OK(hipFree(dX)); HIP_OK(hipFree(dY));
    return bad ? 1 : 0;
}

//   boundary_test --functor
static int run_functor() {
    int failures = 0;
    failures += run_untuned_mix<int, double, float, float>("SpMV_hip_vector<int, int, double, float, float>",
        [](int r, int c, int z, const int* p, const int* j, const double* a, const float* x, float* y) { SpMV_hip_vector(r, c, z, p, j, a, x, y); });
    failures += run_untuned_mix<long long, float, float, double>("SpMV_hip_light<int, long long, float, float, double>",
        [](int r, int c, long long z, const long long* p, const int* j, const float* a, const float* x, double* y) { SpMV_hip_light(r, c, z, p, j, a, x, y); });
    failures += run_untuned_mix<int, int, int, long long>("SpMV_hip_merge<int, int, int, int, long long>",
        [](int r, int c, int z, const int* p, const int* j, const int* a, const int* x, long long* y) { SpMV_hip_merge(r, c, z, p, j, a, x, y); });
    failures += run_untuned_mix<int, int, int, int>("SpMV_hip_vector<int, int, int, int, int>  (integers are tuned for merge only)",
        [](int r, int c, int z, const int* p, const int* j, const int* a, const int* x, int* y) { SpMV_hip_vector(r, c, z, p, j, a, x, y); });
    failures += run_functor_case<ManhattanTerm_text, true, ManhattanTerm<float, int, double>, long long, float, int, double>(
        "sum |a - x|  (i64 offsets, float matrix, int x, double y)", 20011, 3000, 40, 9000);
    failures += run_functor_case<Bottleneck_text, true, Bottleneck<float, double, float>, int, float, double, float>(
        "max min(a, x)  (i32 offsets, float matrix, double x, float y)", 7001, 500, 6, 3000);
    failures += run_functor_case<CountPositive_text, false, CountPositive, int, double, float, long long>(
        "count a * x > 0  (plain struct; double matrix, float x, long long y)", 4099, 4099, 150, 100);
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}

// The multi-GPU kinds keep a handle that holds COPIES of the structure; SpMV(kind, ...) reads its arrays on every call
// (the reference has no state between calls).  So: a call, then the matrix rewritten IN PLACE — other row lengths and
// columns, same sizes, same addresses — and the values and x as well, then a second call that must see all of it.
//   boundary_test --dist-rewrite [gpus]      (gpus > 1: under the emulated RCCL, see tests/test_gpu_host.py)
static int run_dist_rewrite(int gpus) {
    using value_t = float;
    const int n_rows = 6000, n_cols = 5000;
    mi355_host::dist_gpus() = gpus;
    mi355_host::dist_sub_blocks() = 3;
    int failures = 0;
    int* dAp; int* dAj; value_t *dAx, *dX, *dY;
    std::vector<int> Ap(n_rows + 1), Aj;
    std::vector<value_t> Ax, x(n_cols), y(n_rows);
    const int nnz = 24 * n_rows;
    HIP_OK(hipMalloc((void**)&dAp, (n_rows + 1) * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dAj, (size_t(nnz) + 4) * sizeof(int)));
    HIP_OK(hipMalloc((void**)&dAx, (size_t(nnz) + 4) * sizeof(value_t)));
    HIP_OK(hipMalloc((void**)&dX, n_cols * sizeof(value_t)));
    HIP_OK(hipMalloc((void**)&dY, n_rows * sizeof(value_t)));
    for (const char* label : {"hip_dist_vector", "hip_dist_merge", "hip_dist_light"}) {
        for (int version = 0; version < 2; ++version) {
            unsigned long long seed = 99 + 17 * version;
            // version 0: 24 per row; version 1: 8 / 40 alternating (same nnz), other columns, other values, other x
            Aj.clear(); Ax.clear();
            Ap[0] = 0;
            for (int r = 0; r < n_rows; ++r) {
                const int len = version == 0 ? 24 : ((r & 1) ? 40 : 8);
                for (int k = 0; k < len; ++k) {
                    Aj.push_back(int(lcg(seed) % (unsigned)n_cols));
                    Ax.push_back(value_t(double(lcg(seed) % 2001) / 1000.0 - 1.0));
                }
                Ap[r + 1] = int(Aj.size());
            }
            for (int c = 0; c < n_cols; ++c) x[c] = value_t(double(lcg(seed) % 2001) / 1000.0 - 1.0);
            HIP_OK(hipMemcpy(dAp, Ap.data(), (n_rows + 1) * sizeof(int), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(dAj, Aj.data(), size_t(nnz) * sizeof(int), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(dAx, Ax.data(), size_t(nnz) * sizeof(value_t), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(dX, x.data(), n_cols * sizeof(value_t), hipMemcpyHostToDevice));
            std::vector<value_t> poison(n_rows, std::numeric_limits<value_t>::quiet_NaN());
            HIP_OK(hipMemcpy(dY, poison.data(), n_rows * sizeof(value_t), hipMemcpyHostToDevice));
            SpMV<int, int, value_t, value_t, value_t>(label, n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
            HIP_OK(hipMemcpy(y.data(), dY, n_rows * sizeof(value_t), hipMemcpyDeviceToHost));
            int bad = 0;
            for (int r = 0; r < n_rows; ++r) {
                double s = 0, a = 0;
                for (int k = Ap[r]; k < Ap[r + 1]; ++k) { const double p = double(Ax[k]) * double(x[Aj[k]]); s += p; a += std::fabs(p); }
                if (!(std::fabs(double(y[r]) - s) <= (double(Ap[r + 1] - Ap[r]) + 2) * std::ldexp(1.0, -24) * a)) ++bad;
            }
            std::printf("[%-15s] gpus=%d %s bad_rows=%d\n", label, gpus, version ? "rewritten in place" : "first matrix", bad);
            failures += bad ? 1 : 0;
        }
        mi355_host::dist_release();
    }
    HIP_OK(hipFree(dAp)); HIP_OK(hipFree(dAj)); HIP_OK(hipFree(dAx)); HIP_OK(hipFree(dX)); HIP_OK(hipFree(dY));
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}

int main(int argc, char** argv) {
    HIP_OK(hipSetDevice(0));  // USED_DEVICE 0 (common.cuh:8)
    if (argc > 1 && std::strcmp(argv[1], "--dist-rewrite") == 0) return run_dist_rewrite(argc > 2 ? std::atoi(argv[2]) : 1);
    if (argc > 1 && std::strcmp(argv[1], "--functor") == 0) return run_functor();
    if (argc > 1 && std::strcmp(argv[1], "--bad-label") == 0) {
        int* d;
        HIP_OK(hipMalloc((void**)&d, 64));
        SpMV<int, int, float, float, float>("no_such_kind", 1, 1, 0, d, d, (float*)d, (float*)d, (float*)d);
        return 0;  // not reached: SpMV exits with EXIT_FAILURE
    }
    int failures = 0;
    mi355_host::dist_sub_blocks() = 3;   // hip_dist_*: one GPU here, its rows in three blocks
    failures += run_combo<int, float>("i32_f32", 3001, 2500, 24);
    failures += run_combo<int, double>("i32_f64", 3001, 2500, 24);
    failures += run_combo<long long, float>("i64_f32", 3001, 2500, 24);
    failures += run_combo<long long, double>("i64_f64", 777, 1, 3);  // n_cols == 1
    failures += run_mixed();
    failures += run_integer();
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
