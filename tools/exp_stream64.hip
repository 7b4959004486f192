// tools/exp_stream64.hip — scratch: does the fp64 value stream pay for its 32-byte lane stride?
// Pattern A (what the kernels do): per lane one 16-byte load of 4 columns + one 32-byte load of 4 doubles
// (two dwordx4, each with a 32-byte stride between lanes).  Pattern B: per lane two 8-byte loads of 2 columns
// + two 16-byte loads of 2 doubles, every instruction contiguous across the wave.  (not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
typedef double double4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

template <int R, int PATTERN>
__global__ __launch_bounds__(256) void stream64(long long n4, int CH, const int* Aj, const double* Ax, double* out) {
    extern __shared__ float lds[];
    const long long base = (long long)blockIdx.x * CH;   // in groups of 4 elements
    double acc = 0;
    for (int i = threadIdx.x; i < CH; i += 256 * R) {
        if (PATTERN == 0) {
            int4v c[R]; double4v a[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                long long k = base + i + r * 256; k = k < n4 ? k : n4 - 1;
                c[r] = __builtin_nontemporal_load(reinterpret_cast<const int4v*>(Aj) + k);
                a[r] = __builtin_nontemporal_load(reinterpret_cast<const double4v*>(Ax) + k);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc += a[r][0] * c[r][0] + a[r][1] * c[r][1] + a[r][2] * c[r][2] + a[r][3] * c[r][3];
        } else {
            int2v c[R][2]; double2v a[R][2];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // the 256 threads of the block cover 1024 consecutive elements: two halves of 512, 2 per lane each
                long long g = base + (i - threadIdx.x) + r * 256; g = g < n4 - 256 ? g : n4 - 256;   // group of the block's first lane
                const long long e0 = g * 4 + 2 * threadIdx.x;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    c[r][h] = __builtin_nontemporal_load(reinterpret_cast<const int2v*>(Aj + e0 + h * 512));
                    a[r][h] = __builtin_nontemporal_load(reinterpret_cast<const double2v*>(Ax + e0 + h * 512));
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc += a[r][h][0] * c[r][h][0] + a[r][h][1] * c[r][h][1];
        }
    }
    if (acc == 123.456) out[0] = acc + lds[0];
}

template <typename F>
static float time_it(F f, int iters = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

int main() {
    const long long nnz = 1ll << 27, n4 = nnz / 4;
    int* Aj; double* Ax; double* out;
    CK(hipMalloc(&Aj, nnz * 4)); CK(hipMalloc(&Ax, nnz * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemset(Aj, 1, nnz * 4)); CK(hipMemset(Ax, 0, nnz * 8));
#define RUN(R, P, CHK, LDSKB) { const int CH = CHK; const unsigned grid = unsigned((n4 + CH - 1) / CH); \
    float ms = time_it([&] { stream64<R, P><<<grid, 256, LDSKB * 1024>>>(n4, CH, Aj, Ax, out); }); \
    printf("pattern=%d R=%d chunk=%6d groups lds=%2d KB grid=%6u : %7.3f ms %7.1f GB/s\n", P, R, CHK, LDSKB, grid, ms, double(nnz) * 12 / ms / 1e6); }
    for (int rep = 0; rep < 2; ++rep) {
        RUN(2, 0, 8192, 50) RUN(2, 1, 8192, 50) RUN(4, 0, 8192, 50) RUN(4, 1, 8192, 50) RUN(2, 0, 8192, 0) RUN(2, 1, 8192, 0)
    }
    return 0;
}
