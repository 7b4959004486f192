// tools/exp_stream.hip — scratch: two-stream read ceiling for the CHUNKED access pattern of the
// SpMV kernels (each workgroup streams its own contiguous piece of Aj and Ax), by occupancy,
// chunk size, block->chunk mapping and load flavour.  (not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcd_id(unsigned bid, unsigned n) {
    const unsigned per = n / 8, rem = n % 8, x = bid % 8, k = bid / 8;
    return x * per + (x < rem ? x : rem) + k;
}

// each block: chunk of CH float4 groups of each array; each wave iteration loads R groups per lane per array
template <int R, int NT, int XCD>
__global__ __launch_bounds__(256) void chunk_stream(long long n4, int CH, const int4v* Aj, const float4v* Ax, float* out, float* y) {
    extern __shared__ float lds[];
    const unsigned chunk = XCD ? xcd_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const long long base = (long long)chunk * CH;
    float acc = 0.f;
    for (int i = threadIdx.x; i < CH; i += 256 * R) {
        int4v c[R]; float4v a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long k = base + i + r * 256;
            if (i + r * 256 < CH && k < n4) {
                if (NT) { c[r] = __builtin_nontemporal_load(Aj + k); a[r] = __builtin_nontemporal_load(Ax + k); }
                else { c[r] = Aj[k]; a[r] = Ax[k]; }
            } else { c[r] = int4v{0,0,0,0}; a[r] = float4v{0,0,0,0}; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc += a[r][0] * float(c[r][0]) + a[r][1] * float(c[r][1]) + a[r][2] * float(c[r][2]) + a[r][3] * float(c[r][3]);
    }
    if (acc == 123.456f) out[0] = acc + lds[0];
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
    int4v* Aj; float4v* Ax; float *out, *y;
    CK(hipMalloc(&Aj, nnz * 4)); CK(hipMalloc(&Ax, nnz * 4)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&y, 1 << 24));
    CK(hipMemset(Aj, 1, nnz * 4)); CK(hipMemset(Ax, 0, nnz * 4));
#define RUN(R, NT, XCD, CHKB, LDSKB) { const int CH = CHKB * 1024 / 16; const unsigned grid = unsigned((n4 + CH - 1) / CH); \
    CK(hipFuncSetAttribute((const void*)chunk_stream<R, NT, XCD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    float ms = time_it([&] { chunk_stream<R, NT, XCD><<<grid, 256, LDSKB * 1024>>>(n4, CH, Aj, Ax, out, y); }); \
    printf("R=%d nt=%d xcdmap=%d chunk=%4d KB/array lds=%3d KB (blocks/CU<=%d) grid=%6u : %7.3f ms %7.1f GB/s\n", R, NT, XCD, CHKB, LDSKB, LDSKB ? 160 / LDSKB : 8, grid, ms, double(nnz) * 8 / ms / 1e6); }
    for (int rep = 0; rep < 2; ++rep) {
        RUN(4, 1, 1, 128, 37) RUN(4, 1, 0, 128, 37) RUN(4, 0, 1, 128, 37)
        RUN(4, 1, 1, 128, 0) RUN(4, 1, 0, 128, 0) RUN(2, 1, 1, 128, 0) RUN(1, 1, 0, 128, 0)
        RUN(4, 1, 1, 32, 37) RUN(4, 1, 0, 32, 37) RUN(4, 1, 0, 16, 0) RUN(1, 1, 0, 4, 0) RUN(4, 1, 1, 512, 37) RUN(8, 1, 1, 128, 37)
        RUN(4, 1, 1, 128, 20) RUN(4, 1, 1, 128, 75)
    }
    return 0;
}
