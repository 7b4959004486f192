// tools/exp_spmv.hip — scratch micro-benchmarks used to choose the kernel structure
// (not part of the library).  Builds a banded 32-nnz/row fp32 CSR on the device and
// times variants of the inner loop with hipEvents.
//   hipcc -O3 --offload-arch=gfx950 tools/exp_spmv.hip -o gpurun_out/exp_spmv && ./gpurun_out/exp_spmv
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a;
}

// 32 nnz per row, column i of row r in stratum i of the window [r-w, r+w)
__global__ void gen_kernel(int n, int w, int* Ap, int* Aj, float* Ax, float* x, int random_cols) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < (long long)n * 32) {
        const int r = int(k >> 5), i = int(k & 31);
        const unsigned h = hash32(unsigned(k) * 2654435761u + 12345u);
        long long c;
        if (random_cols) {
            const long long stratum = (long long)n / 32;
            c = (long long)i * stratum + (h % (unsigned)stratum);
        } else {
            const int stratum = 2 * w / 32;
            c = (long long)r - w + (long long)i * stratum + (h % (unsigned)stratum);
        }
        if (c < 0) c = i;
        if (c >= n) c = n - 32 + i;
        Aj[k] = int(c);
        Ax[k] = float(hash32(h) & 0xffff) / 32768.0f - 1.0f;
    }
    if (k <= n) Ap[k] = int(k * 32);
    if (k < n) x[k] = float(hash32(unsigned(k) + 99u) & 0xffff) / 32768.0f - 1.0f;
}

// ---- V0: pure two-stream read ceiling -------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void stream_kernel(long long nnz4, const int4v* Aj, const float4v* Ax, float* out) {
    float acc = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nnz4; i += U * stride) {
        int4v c[U]; float4v a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { c[u] = __builtin_nontemporal_load(Aj + i + u * stride); a[u] = __builtin_nontemporal_load(Ax + i + u * stride); }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += a[u][0] * float(c[u][0]) + a[u][1] * float(c[u][1]) + a[u][2] * float(c[u][2]) + a[u][3] * float(c[u][3]);
    }
    if (acc == 123.456f) out[0] = acc;
}

// ---- V1: CSR-vector, T lanes per row, R rows per vector in flight ------------------------
template <int T, int R, int GATHER>
__global__ __launch_bounds__(256) void vec_kernel(int n_rows, const int* __restrict__ Ap, const int* __restrict__ Aj,
                                                  const float* __restrict__ Ax, const float* __restrict__ x,
                                                  float* __restrict__ y) {
    const int lane = threadIdx.x & (T - 1);
    const long long vec = ((long long)blockIdx.x * 256 + threadIdx.x) / T;
    const long long row0 = vec * R;
    int s[R], e[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long row = row0 + r;
        s[r] = row < n_rows ? Ap[row] : 0;
        e[r] = row < n_rows ? Ap[row + 1] : 0;
    }
    int4v c[R]; float4v a[R];
    int j[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        j[r] = (s[r] & ~3) + lane * 4;
        if (j[r] < e[r]) {
            c[r] = __builtin_nontemporal_load((const int4v*)(Aj + j[r]));
            a[r] = __builtin_nontemporal_load((const float4v*)(Ax + j[r]));
        } else { c[r] = int4v{0, 0, 0, 0}; a[r] = float4v{0, 0, 0, 0}; }
    }
    float xv[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[r][q] = GATHER ? x[c[r][q]] : x[(c[r][q] & 63) + lane];
    float sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        sum[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = j[r] + q;
            sum[r] = (k >= s[r] && k < e[r]) ? sum[r] + a[r][q] * xv[r][q] : sum[r];
        }
        // long rows (not present in this benchmark) would loop here
#pragma unroll
        for (int o = T / 2; o >= 1; o >>= 1) sum[r] += __shfl_down(sum[r], o, T);
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) if (row0 + r < n_rows) y[row0 + r] = sum[r];
    }
}

// ---- V3: x window staged in LDS, rows chunk per block -------------------------------------
// block handles ROWS rows; window = [row_begin - w, row_end + w) clipped
template <int T, int R, int ROWS, int WMAX>
__global__ __launch_bounds__(256) void lds_kernel(int n_rows, int w, const int* __restrict__ Ap, const int* __restrict__ Aj,
                                                  const float* __restrict__ Ax, const float* __restrict__ x,
                                                  float* __restrict__ y) {
    __shared__ float s_x[ROWS + 2 * WMAX];
    const int rb = blockIdx.x * ROWS;
    const int wlo = max(rb - w, 0);
    const int whi = min(rb + ROWS + w, n_rows);
    for (int i = threadIdx.x * 4; i < whi - wlo; i += 256 * 4) {
        // window start is a multiple of 4 when w and ROWS are
        *(float4v*)&s_x[i] = *(const float4v*)&x[wlo + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & (T - 1);
    const int vec = threadIdx.x / T;
    constexpr int VECS = 256 / T;
    for (int base = 0; base < ROWS; base += VECS * R) {
        const int row0 = rb + base + vec * R;
        int s[R], e[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            s[r] = row < n_rows ? Ap[row] : 0;
            e[r] = row < n_rows ? Ap[row + 1] : 0;
        }
        int4v c[R]; float4v a[R]; int j[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            j[r] = (s[r] & ~3) + lane * 4;
            if (j[r] < e[r]) {
                c[r] = __builtin_nontemporal_load((const int4v*)(Aj + j[r]));
                a[r] = __builtin_nontemporal_load((const float4v*)(Ax + j[r]));
            } else { c[r] = int4v{wlo, wlo, wlo, wlo}; a[r] = float4v{0, 0, 0, 0}; }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = j[r] + q;
                int ci = c[r][q] - wlo;
                ci = min(max(ci, 0), ROWS + 2 * WMAX - 1);
                const float xv = s_x[ci];
                sum = (k >= s[r] && k < e[r]) ? sum + a[r][q] * xv : sum;
            }
#pragma unroll
            for (int o = T / 2; o >= 1; o >>= 1) sum += __shfl_down(sum, o, T);
            if (lane == 0 && row0 + r < n_rows) y[row0 + r] = sum;
        }
    }
}

static double checksum(const float* d, int n) {
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < n; ++i) s += h[i];
    return s;
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

int main(int argc, char** argv) {
    const int n = 1 << 22, w = 4096;
    const long long nnz = (long long)n * 32;
    int *Ap, *Aj; float *Ax, *x, *y;
    CK(hipMalloc(&Ap, (n + 1) * 4)); CK(hipMalloc(&Aj, nnz * 4)); CK(hipMalloc(&Ax, nnz * 4));
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
    const double bytes = double(nnz) * 8 + (n + 1) * 4.0 + n * 4.0 + n * 4.0;
    for (int random_cols = 0; random_cols < 2; ++random_cols) {
        gen_kernel<<<(unsigned)((nnz + 255) / 256), 256>>>(n, w, Ap, Aj, Ax, x, random_cols);
        CK(hipDeviceSynchronize());
        printf("=== %s columns ===\n", random_cols ? "uniform-random" : "banded w=4096");
        if (!random_cols) {
#define STREAM(U, G) { float ms = time_it([&] { stream_kernel<U><<<G, 256>>>(nnz / 4, (const int4v*)Aj, (const float4v*)Ax, y); }); \
            printf("stream U=%d grid=%6d : %7.3f ms  %7.1f GB/s (Aj+Ax only)\n", U, G, ms, double(nnz) * 8 / ms / 1e6); }
            STREAM(1, 2048) STREAM(2, 2048) STREAM(4, 2048) STREAM(4, 4096) STREAM(8, 2048) STREAM(1, 131072) STREAM(2, 65536)
        }
#define VEC(T, R, G) { CK(hipMemset(y, 0, n * 4)); float ms = time_it([&] { vec_kernel<T, R, G><<<(unsigned)(((long long)n / R * T + 255) / 256), 256>>>(n, Ap, Aj, Ax, x, y); }); \
        printf("vec T=%2d R=%d gather=%d : %7.3f ms  %7.1f GB/s  sum=%.6e\n", T, R, G, ms, bytes / ms / 1e6, checksum(y, n)); }
        VEC(8, 1, 1) VEC(8, 2, 1) VEC(8, 4, 1) VEC(8, 8, 1) VEC(8, 1, 0) VEC(8, 4, 0) VEC(8, 8, 0)
        VEC(4, 4, 1) VEC(16, 4, 1)
        if (!random_cols) {
#define LDS(T, R, ROWS) { CK(hipMemset(y, 0, n * 4)); float ms = time_it([&] { lds_kernel<T, R, ROWS, 4096><<<n / ROWS, 256>>>(n, w, Ap, Aj, Ax, x, y); }); \
            printf("lds T=%2d R=%d rows/block=%4d : %7.3f ms  %7.1f GB/s  sum=%.6e\n", T, R, ROWS, ms, bytes / ms / 1e6, checksum(y, n)); }
            LDS(8, 2, 512) LDS(8, 4, 512) LDS(8, 4, 1024) LDS(8, 4, 2048) LDS(8, 8, 2048) LDS(8, 4, 4096)
        }
    }
    return 0;
}
