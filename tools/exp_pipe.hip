// tools/exp_pipe.hip — scratch: what does software-pipelining the row-group loop buy,
// and what do the sampled-window prologue / the fallback check cost?  Uses the
// library's own headers.  (not part of the library)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../spmv-samples_amd/csrc/xwindow.hpp"

namespace mi355 { void set_error(const char*, ...) {} }
using namespace mi355;

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned a) {
    a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a;
}
__global__ void gen_kernel(int n, int w, int* Ap, int* Aj, float* Ax, float* x) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < (long long)n * 32) {
        const int r = int(k >> 5), i = int(k & 31);
        const unsigned h = hash32(unsigned(k) * 2654435761u + 12345u);
        const int stratum = 2 * w / 32;
        long long c = (long long)r - w + (long long)i * stratum + (h % (unsigned)stratum);
        if (c < 0) c = i;
        if (c >= n) c = n - 32 + i;
        Aj[k] = int(c);
        Ax[k] = float(hash32(h) & 0xffff) / 32768.0f - 1.0f;
    }
    if (k <= n) Ap[k] = int(k * 32);
    if (k < n) x[k] = float(hash32(unsigned(k) + 99u) & 0xffff) / 32768.0f - 1.0f;
}

// MODE 0: library body (sampled window, fallback)          = csr_vector_window_kernel
// MODE 1: sampled window, pipelined row groups (loads one group ahead, Ap two ahead)
// MODE 2: like 1, analytic window (no sample)   -> cost of the sample prologue
// MODE 3: like 1, no fallback check             -> cost of the range check
template <int T, int R, int MODE>
__global__ __launch_bounds__(kBlock) void k_var(int n_rows, int n_cols, int nnz, const int* __restrict__ Ap,
                                                const int* __restrict__ Aj, const float* __restrict__ Ax,
                                                const float* __restrict__ x, float* __restrict__ y,
                                                int rows_per_chunk, int w) {
    __shared__ __attribute__((aligned(16))) float s_x[kWindowBytes / 4];
    __shared__ int s_red[8];
    const unsigned chunk = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int64_t rb = int64_t(chunk) * rows_per_chunk;
    const int64_t re = min(rb + rows_per_chunk, int64_t(n_rows));
    XWindow<float> win;
    constexpr int VECS0 = kBlock / T;
    const int lane0 = threadIdx.x & (T - 1);
    const int vec0 = threadIdx.x / T;
    int pre_b[R + 1]; int4v pre_c[R]; float4v pre_a[R]; int pre_j[R];
    if (MODE == 4) {
#pragma unroll
        for (int r = 0; r <= R; ++r) { const int64_t row = rb + int64_t(vec0) * R + r; pre_b[r] = Ap[row < re ? row : re]; }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            pre_j[r] = (pre_b[r] & ~3) + lane0 * 4;
            if (pre_j[r] < pre_b[r + 1] && pre_j[r] + 4 <= nnz) {
                pre_c[r] = stream_load((const int4v*)(Aj + pre_j[r]));
                pre_a[r] = stream_load((const float4v*)(Ax + pre_j[r]));
            } else { pre_c[r] = int4v{0, 0, 0, 0}; pre_a[r] = float4v{0, 0, 0, 0}; }
        }
        (void)VECS0;
    }
    if ((MODE >= 16) && ((MODE - 16) & 2)) {
        win.s_x = s_x; win.lo = max(int(rb) - w, 0) & ~3; win.len = min(min(int(re) + w, n_cols) - win.lo, kWindowBytes / 4);
    } else if (MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || MODE >= 16) {
        win.s_x = s_x;
        win.lo = max(int(rb) - w, 0) & ~3;
        win.len = min(min(int(re) + w, n_cols) - win.lo, kWindowBytes / 4);
        for (int g = threadIdx.x; g < win.len / 4; g += kBlock) *(float4v*)&s_x[4 * g] = *(const float4v*)&x[win.lo + 4 * g];
        __syncthreads();
    } else {
        win = stage_x_window<int, float>(rb, re, n_cols, Ap, Aj, x, s_x, kWindowBytes / 4, s_red);
    }
    if (MODE == 6 || MODE >= 16) {
        constexpr int FL = MODE >= 16 ? MODE - 16 : 0;
        // bounds of the chunk in LDS; branch-free clamped loads; 2x unrolled ping-pong register sets
        __shared__ int s_b[2048 + 8];
        __shared__ __attribute__((aligned(16))) float s_y[2048 + 64];
        const int rows = int(re - rb);
        for (int i = threadIdx.x; i <= rows; i += kBlock) s_b[i] = Ap[rb + i];
        __syncthreads();
        constexpr int VECS = kBlock / T;
        const int lane = threadIdx.x & (T - 1);
        const int vec = threadIdx.x / T;
        const int stride = VECS * R;
        const int ngroups = (rows + stride - 1) / stride;
        const int jmax = (nnz - 4) & ~3;
        auto issue = [&](int g, int4v* c, float4v* a, int* j, int* b) {
#pragma unroll
            for (int r = 0; r <= R; ++r) {
                if (FL & 8) { const int vw = vec & (kWave / T - 1), wv = vec / (kWave / T);   // row = wave base + r*8 + v
                    const int rl = g * stride + wv * (kWave / T) * R + min(r, R - 1) * (kWave / T) + vw + (r == R ? 1 : 0);
                    b[r] = s_b[min(rl, rows)]; }
                else b[r] = s_b[min(g * stride + vec * R + r, rows)];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int bs = b[r], be = (FL & 8) ? bs + 32 : b[r + 1];   // (flag 8: rows are exactly 32 long in this benchmark)
                j[r] = (bs & ~3) + lane * 4;
                int jl = j[r] < be ? j[r] : (bs & ~3);
                jl = min(jl, jmax);
                c[r] = stream_load((const int4v*)(Aj + jl));
                a[r] = stream_load((const float4v*)(Ax + jl));
            }
        };
        auto compute = [&](int g, const int4v* c, const float4v* a, const int* j, const int* b) {
            const int64_t row0 = rb + int64_t(g) * stride + int64_t(vec) * R;
            float sum[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                sum[r] = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = j[r] + e;
                    const bool valid = (k >= b[r]) && (k < ((FL & 8) ? b[r] + 32 : b[r + 1]));
                    const unsigned rel = unsigned(c[r][e] - win.lo);
                    if (FL & 4) {
                        sum[r] = valid ? sum[r] + a[r][e] * float(rel) : sum[r];
                    } else if (rel < unsigned(win.len)) {
                        const float xv = s_x[rel];
                        sum[r] = valid ? sum[r] + a[r][e] * xv : sum[r];
                    } else if (valid) {
                        sum[r] += a[r][e] * x[c[r][e]];
                    }
                }
                sum[r] = vector_reduce<T, float>(sum[r]);
            }
            if (FL & 32) {
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r) s_y[g * stride + vec * R + r] = sum[r];
                }
            } else
            if ((FL & 1) ? (sum[0] == 1.2345f) : (lane == 0)) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int64_t row = (FL & 8) ? rb + int64_t(g) * stride + (vec / (kWave / T)) * (kWave / T) * R + r * (kWave / T) + (vec & (kWave / T - 1)) : row0 + r;
                    if (row < re) y[row] = sum[r];
                }
            }
        };
        int4v c0[R], c1[R]; float4v a0[R], a1[R]; int j0[R], j1[R], b0[R + 1], b1[R + 1];
        issue(0, c0, a0, j0, b0);
        for (int g = 0; g < ngroups; g += 2) {
            issue(g + 1, c1, a1, j1, b1);
            compute(g, c0, a0, j0, b0);
            issue(g + 2, c0, a0, j0, b0);
            compute(g + 1, c1, a1, j1, b1);
        }
        if (FL & 32) {
            __syncthreads();
            for (int i = threadIdx.x * 4; i < rows; i += kBlock * 4) {
                if (FL & 64) __builtin_nontemporal_store(*(const float4v*)&s_y[i], (float4v*)&y[rb + i]);
                else *(float4v*)&y[rb + i] = *(const float4v*)&s_y[i];
            }
        }
        return;
    }
    if (MODE == 5) {
        constexpr int VECS = kBlock / T;
        const int lane = threadIdx.x & (T - 1);
        const int vec = threadIdx.x / T;
        const int stride = VECS * R;
        auto load_bounds = [&](int64_t base, int* b) {
#pragma unroll
            for (int r = 0; r <= R; ++r) { const int64_t row = base + int64_t(vec) * R + r; b[r] = Ap[row < re ? row : re]; }
        };
        auto issue = [&](const int* b, int4v* c, float4v* a, int* j) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                j[r] = (b[r] & ~3) + lane * 4;
                if (j[r] < b[r + 1] && j[r] + 4 <= nnz) {
                    c[r] = stream_load((const int4v*)(Aj + j[r]));
                    a[r] = stream_load((const float4v*)(Ax + j[r]));
                } else { c[r] = int4v{0, 0, 0, 0}; a[r] = float4v{0, 0, 0, 0}; }
            }
        };
        int b0[R + 1], b1[R + 1], b2[R + 1];
        int4v c0[R], c1[R]; float4v a0[R], a1[R]; int j0[R], j1[R];
        load_bounds(rb, b0);
        load_bounds(rb + stride, b1);
        issue(b0, c0, a0, j0);
        for (int64_t base = rb; base < re; base += stride) {
            load_bounds(base + 2 * stride, b2);      // A: bounds two groups ahead
            issue(b1, c1, a1, j1);                   // B: next group's stream (needs b1, issued before c0/a0)
            const int64_t row0 = base + int64_t(vec) * R;
            float sum[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {            // C: compute the current group
                sum[r] = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = j0[r] + e;
                    const bool valid = (k >= b0[r]) && (k < b0[r + 1]);
                    const unsigned rel = unsigned(c0[r][e] - win.lo);
                    if (rel < unsigned(win.len)) {
                        const float xv = s_x[rel];
                        sum[r] = valid ? sum[r] + a0[r][e] * xv : sum[r];
                    } else if (valid) {
                        sum[r] += a0[r][e] * x[c0[r][e]];     // rare: load, wait and use inside the branch
                    }
                }
                sum[r] = vector_reduce<T, float>(sum[r]);
            }
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) if (row0 + r < re) y[row0 + r] = sum[r];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) { c0[r] = c1[r]; a0[r] = a1[r]; j0[r] = j1[r]; }
#pragma unroll
            for (int r = 0; r <= R; ++r) { b0[r] = b1[r]; b1[r] = b2[r]; }
        }
        return;
    }
    if (MODE == 0) {
        // (the library body of that day; chunk_rows has since become the MODE 6 structure)
        return;
    }
    constexpr int VECS = kBlock / T;
    const int lane = threadIdx.x & (T - 1);
    const int vec = threadIdx.x / T;
    const int stride = VECS * R;
    // pipeline registers
    int bnd_n[R + 1];    // bounds of the NEXT group
    int4v c_n[R]; float4v a_n[R]; int j_n[R];
    auto load_bounds = [&](int64_t base, int* b) {
#pragma unroll
        for (int r = 0; r <= R; ++r) { const int64_t row = base + int64_t(vec) * R + r; b[r] = Ap[row < re ? row : re]; }
    };
    auto issue = [&](const int* b, int4v* c, float4v* a, int* j) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            j[r] = (b[r] & ~3) + lane * 4;
            if (j[r] < b[r + 1] && j[r] + 4 <= nnz) {
                c[r] = stream_load((const int4v*)(Aj + j[r]));
                a[r] = stream_load((const float4v*)(Ax + j[r]));
            } else { c[r] = int4v{0, 0, 0, 0}; a[r] = float4v{0, 0, 0, 0}; }
        }
    };
    int bnd_c[R + 1];
    int4v c_c[R]; float4v a_c[R]; int j_c[R];
    if (MODE == 4) {
#pragma unroll
        for (int r = 0; r <= R; ++r) bnd_c[r] = pre_b[r];
#pragma unroll
        for (int r = 0; r < R; ++r) { c_c[r] = pre_c[r]; a_c[r] = pre_a[r]; j_c[r] = pre_j[r]; }
    } else {
        load_bounds(rb, bnd_c);
        issue(bnd_c, c_c, a_c, j_c);
    }
    load_bounds(rb + stride, bnd_n);
    for (int64_t base = rb; base < re; base += stride) {
        // issue next group's stream loads, and the bounds of the group after it
        issue(bnd_n, c_n, a_n, j_n);
        int bnd_nn[R + 1];
        load_bounds(base + 2 * stride, bnd_nn);
        const int64_t row0 = base + int64_t(vec) * R;
        float sum[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            sum[r] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = j_c[r] + e;
                const bool valid = (k >= bnd_c[r]) && (k < bnd_c[r + 1]);
                float xv;
                if (MODE == 3) { const unsigned rel = unsigned(c_c[r][e] - win.lo); xv = s_x[rel < unsigned(kWindowBytes / 4) ? rel : 0]; }
                else xv = window_gather<float>(win, x, c_c[r][e], valid);
                sum[r] = valid ? sum[r] + a_c[r][e] * xv : sum[r];
            }
            // (rows longer than 4T would continue here, unpipelined; none in this benchmark)
            sum[r] = vector_reduce<T, float>(sum[r]);
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) if (row0 + r < re) y[row0 + r] = sum[r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { c_c[r] = c_n[r]; a_c[r] = a_n[r]; j_c[r] = j_n[r]; }
#pragma unroll
        for (int r = 0; r <= R; ++r) { bnd_c[r] = bnd_n[r]; bnd_n[r] = bnd_nn[r]; }
    }
}

static double checksum(const float* d, int n) {
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < n; ++i) s += h[i];
    return s;
}
template <typename F>
static float time_it(F f, int iters = 30) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

int main() {
    const int n = 1 << 22, w = 4096;
    const long long nnz = (long long)n * 32;
    int *Ap, *Aj; float *Ax, *x, *y;
    CK(hipMalloc(&Ap, (n + 1) * 4)); CK(hipMalloc(&Aj, nnz * 4)); CK(hipMalloc(&Ax, nnz * 4));
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
    const double bytes = double(nnz) * 8 + (n + 1) * 4.0 + n * 4.0 + n * 4.0;
    gen_kernel<<<(unsigned)((nnz + 255) / 256), 256>>>(n, w, Ap, Aj, Ax, x);
    CK(hipDeviceSynchronize());
#define RUN(T, R, MODE, RPC) { CK(hipMemset(y, 0, n * 4)); float ms = time_it([&] { k_var<T, R, MODE><<<n / RPC, 256>>>(n, n, (int)nnz, Ap, Aj, Ax, x, y, RPC, w); }); \
    printf("T=%d R=%d mode=%d rows/chunk=%5d : %7.3f ms  %7.1f GB/s  sum=%.6e\n", T, R, MODE, RPC, ms, bytes / ms / 1e6, checksum(y, n)); }
    for (int rep = 0; rep < 2; ++rep) {
        RUN(8, 4, 2, 1024) RUN(8, 4, 6, 1024) RUN(8, 4, 17, 1024) RUN(8, 4, 48, 1024) RUN(8, 4, 112, 1024) RUN(8, 4, 50, 1024)
    }
    return 0;
}
