// tools/exp_gather.hip — scratch: what does the chip sustain for 4-byte gathers x[Aj[k]] when the
// column stream Aj is read once (16 B per lane, nontemporal) and x is a table of N floats?
// The ceiling the CSR SpMV kernels cannot beat on matrices whose columns have no locality
// (S32-rand: uniform columns, x = 16 MB; C5 R-MAT-24: skewed columns, x = 64 MB).
//   table sizes:  32 KB (L1), 2 MB (one XCD's L2), 16 MB (all L2s / Infinity Cache), 64 MB, 256 MB, 1 GB (HBM)
//   column law:   uniform, or R-MAT (each of the 24 column bits is 1 with probability b + d = 0.24)
//   variants:     G = 16-byte index groups in flight per lane; with/without the Ax stream; x load flavour
// (not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned long long z) {   // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return unsigned((z ^ (z >> 31)) >> 16);
}

// law 0: uniform over [0, n); law 1: R-MAT column bits (P(bit = 1) = 0.24) over `bits` bits
__global__ void fill_cols(int* Aj, long long nnz, unsigned n, int bits, int law, unsigned long long seed) {
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        if (law == 0) {
            Aj[k] = int(mix(seed + k) % n);
        } else {
            unsigned c = 0;
            for (int b = 0; b < bits; ++b) {
                const unsigned r = mix(seed + k * 32 + b) & 0xffff;
                c = (c << 1) | (r < unsigned(0.24 * 65536) ? 1u : 0u);
            }
            Aj[k] = int(c % n);
        }
    }
}

// FLAVOUR 0: plain load of x; 1: nontemporal; 2: relaxed agent-scope atomic load (sc1: bypasses L1)
template <int FLAVOUR>
__device__ __forceinline__ float xload(const float* x, int c) {
    if (FLAVOUR == 1) return __builtin_nontemporal_load(x + c);
    if (FLAVOUR == 2) return __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return x[c];
}

// each workgroup streams a contiguous chunk of CH 16-byte groups; G groups per lane in flight
template <int G, bool AX, int FLAVOUR>
__global__ __launch_bounds__(256) void gather_kernel(long long n4, int CH, const int4v* __restrict__ Aj,
                                                     const float4v* __restrict__ Ax, const float* __restrict__ x,
                                                     float* out) {
    const long long base = (long long)blockIdx.x * CH;
    float acc = 0.f;
    for (int i = threadIdx.x; i < CH; i += 256 * G) {
        int4v c[G];
        float4v a[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            long long k = base + i + g * 256;
            k = k < n4 ? k : n4 - 1;
            c[g] = __builtin_nontemporal_load(Aj + k);
            if (AX) a[g] = __builtin_nontemporal_load(Ax + k);
        }
        float xv[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[g][e] = xload<FLAVOUR>(x, c[g][e]);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += AX ? a[g][e] * xv[g][e] : xv[g][e];
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename F>
static float time_it(F f, int iters = 10) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

int main(int argc, char** argv) {
    const long long nnz = 1ll << 28, n4 = nnz / 4;   // C5's nonzero count
    int* Aj;
    float *Ax, *x, *out;
    CK(hipMalloc(&Aj, nnz * 4));
    CK(hipMalloc(&Ax, nnz * 4));
    CK(hipMalloc(&x, 1ll << 30));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(Ax, 0, nnz * 4));
    CK(hipMemset(x, 0, 1ll << 30));
    const int CH = 8192;   // 128 KB of Aj per workgroup, like the kernels' chunks
    const unsigned grid = unsigned((n4 + CH - 1) / CH);
    struct Case { const char* name; unsigned n; int bits; int law; };
    const Case cases[] = {
        {"uniform  32 KB (L1)      ", 1u << 13, 13, 0}, {"uniform   2 MB (one L2)  ", 1u << 19, 19, 0},
        {"uniform  16 MB (S32-rand)", 1u << 22, 22, 0}, {"uniform  64 MB           ", 1u << 24, 24, 0},
        {"uniform 256 MB           ", 1u << 26, 26, 0}, {"uniform   1 GB (HBM)     ", 1u << 28, 28, 0},
        {"R-MAT    64 MB (C5)      ", 1u << 24, 24, 1}, {"R-MAT    16 MB           ", 1u << 22, 22, 1},
    };
    printf("nnz = 2^28 gathers per pass; 'spmv-equivalent' = 8 B per gather (Aj + Ax stream) / time\n");
    for (const Case& cs : cases) {
        fill_cols<<<4096, 256>>>(Aj, nnz, cs.n, cs.bits, cs.law, 12345);
        CK(hipDeviceSynchronize());
#define RUN(G, AX, FL)                                                                                          \
    {                                                                                                           \
        const float ms = time_it([&] {                                                                          \
            gather_kernel<G, AX, FL><<<grid, 256>>>(n4, CH, (const int4v*)Aj, (const float4v*)Ax, x, out);      \
        });                                                                                                     \
        printf("%s G=%d ax=%d flavour=%d : %8.3f ms  %7.1f Ggather/s  spmv-equivalent %7.1f GB/s\n", cs.name, G, \
               int(AX), FL, ms, double(nnz) / ms / 1e6, double(nnz) * 8 / ms / 1e6);                            \
    }
        RUN(2, false, 0)
        RUN(4, false, 0)
        RUN(8, false, 0)
        RUN(2, true, 0)
        RUN(4, true, 0)
        RUN(4, true, 1)
        RUN(4, true, 2)
    }
    return 0;
}
