// oracle/spmv_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's CSR SpMV hot
// path (peakcrosser7/spmv-samples).  Nothing in the product (the HIP library,
// the host headers under spmv-samples_amd/host) may include, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
// and there only as the checker / the timed CPU baseline.
//
// Parity status: PINNED.  Every function here is checked (tests/test_oracle.py)
// against (a) the reference's own headers compiled where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/ (bit-exact Ap/Aj/Ax/y),
// (b) the only known-answer vector the reference holds, the 9x9 lattice of
// include/spmv/merge_based/device_spmv.cuh:95-128, and (c) the golden vectors
// under tests/golden/ that oracle/make_golden.py produced from (a).
//
// Citations are file:line under /root/reference/.
//
// Built by oracle/Makefile with g++ -O2 -ffp-contract=off (no FMA contraction,
// so sums round exactly as the reference's default x86-64 build does).

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// (1) Serial CSR SpMV  — restates include/spmv/cpu_navie.hpp:5-17.
//     Row loop ascending; sum starts at 0 in the y type (cpu_navie.hpp:10);
//     nonzeros accumulated in ascending CSR order (cpu_navie.hpp:12-14);
//     y[row] overwritten, so an empty row yields 0 (cpu_navie.hpp:15).
//     The reference walks the row with an index_t counter even when offset_t
//     is wider (cpu_navie.hpp:12); for nnz < 2^31 that is the
//     same sequence as the offset_t walk used here.
// ---------------------------------------------------------------------------
template <typename off_t, typename val_t>
void spmv_serial(int32_t row_begin, int32_t row_end, const off_t* Ap,
                 const int32_t* Aj, const val_t* Ax, const val_t* x, val_t* y) {
    for (int32_t row = row_begin; row < row_end; ++row) {
        val_t sum = val_t(0);
        for (off_t k = Ap[row]; k < Ap[row + 1]; ++k) {
            sum += Ax[k] * x[Aj[k]];
        }
        y[row] = sum;
    }
}

// Generalized serial SpMV — restates include/spmv/cpu_navie.hpp:20-34:
//   sum = functor::initialize(); sum = functor::reduce(sum, functor::combine(Ax[k], x[Aj[k]])); y[row] = sum
// with the semirings the C ABI enumerates (include/mi355_spmv.h): 0 = (+, *), 1 = (min, +), 2 = (max, *),
// 3 = (max, +), 4 = (or, and) on 0.0 / 1.0.
template <typename off_t, typename val_t>
void spmv_genl_serial(int semiring, int32_t n_rows, const off_t* Ap, const int32_t* Aj, const val_t* Ax,
                      const val_t* x, val_t* y) {
    const val_t inf = std::numeric_limits<val_t>::infinity();
    for (int32_t row = 0; row < n_rows; ++row) {
        val_t sum = (semiring == 0 || semiring == 4) ? val_t(0) : (semiring == 1 ? inf : -inf);
        for (off_t k = Ap[row]; k < Ap[row + 1]; ++k) {
            if (semiring == 0) sum = sum + Ax[k] * x[Aj[k]];
            else if (semiring == 1) { const val_t v = Ax[k] + x[Aj[k]]; sum = v < sum ? v : sum; }
            else if (semiring == 2) { const val_t v = Ax[k] * x[Aj[k]]; sum = sum < v ? v : sum; }
            else if (semiring == 3) { const val_t v = Ax[k] + x[Aj[k]]; sum = sum < v ? v : sum; }
            else { const val_t v = (Ax[k] != val_t(0) && x[Aj[k]] != val_t(0)) ? val_t(1) : val_t(0);
                   sum = (sum != val_t(0) || v != val_t(0)) ? val_t(1) : val_t(0); }
        }
        y[row] = sum;
    }
}

// The same walk on 32-bit INTEGER values (include/mi355_spmv.h MI355_VAL_I32; the reference's generalized kind is a
// template over the value types, merge_genl.cuh:134-150, and its CPU twin cpu_navie.hpp:20-34 runs on integers as
// it stands — oracle/ref_driver.cpp pins this restatement to it).  Identities 0 / INT32_MAX / INT32_MIN, sums and
// products wrap around (two's complement: computed in uint32_t, so there is no undefined overflow here).
template <typename off_t>
void spmv_genl_serial_i32(int semiring, int32_t n_rows, const off_t* Ap, const int32_t* Aj, const int32_t* Ax,
                          const int32_t* x, int32_t* y) {
    auto add = [](int32_t a, int32_t b) { return int32_t(uint32_t(a) + uint32_t(b)); };
    auto mul = [](int32_t a, int32_t b) { return int32_t(uint32_t(a) * uint32_t(b)); };
    for (int32_t row = 0; row < n_rows; ++row) {
        int32_t sum = (semiring == 0 || semiring == 4) ? 0 : (semiring == 1 ? INT32_MAX : INT32_MIN);
        for (off_t k = Ap[row]; k < Ap[row + 1]; ++k) {
            const int32_t a = Ax[k], b = x[Aj[k]];
            if (semiring == 0) sum = add(sum, mul(a, b));
            else if (semiring == 1) { const int32_t v = add(a, b); sum = v < sum ? v : sum; }
            else if (semiring == 2) { const int32_t v = mul(a, b); sum = sum < v ? v : sum; }
            else if (semiring == 3) { const int32_t v = add(a, b); sum = sum < v ? v : sum; }
            else { const int32_t v = (a != 0 && b != 0) ? 1 : 0; sum = (sum != 0 || v != 0) ? 1 : 0; }
        }
        y[row] = sum;
    }
}

// fp64 serial sum and sum of magnitudes per row: the two quantities of the
// parity bound stated in SURVEY.md §8(c):
//   |y_gpu[r] - y64[r]| <= (len_r + 2) * eps * sum_k |Ax[k] * x[Aj[k]]|
template <typename off_t, typename val_t>
void spmv_ref64(int32_t n_rows, const off_t* Ap, const int32_t* Aj,
                const val_t* Ax, const val_t* x, double* y64, double* yabs) {
    for (int32_t row = 0; row < n_rows; ++row) {
        double s = 0.0, a = 0.0;
        for (off_t k = Ap[row]; k < Ap[row + 1]; ++k) {
            double p = double(Ax[k]) * double(x[Aj[k]]);
            s += p;
            a += std::fabs(p);
        }
        y64[row] = s;
        yabs[row] = a;
    }
}

// spmv_ref64 over rows split into nnz-balanced contiguous ranges, one thread each: every row is still
// the serial fp64 sum above, so the two outputs are identical to the 1-thread call (used by the
// BASELINE-sized parity tests, where the serial pass over 2-3 * 10^8 random gathers takes a while).
template <typename off_t, typename val_t>
void spmv_ref64_parallel(int32_t n_rows, const off_t* Ap, const int32_t* Aj, const val_t* Ax, const val_t* x,
                         double* y64, double* yabs, int n_threads) {
    if (n_threads <= 1 || n_rows < n_threads) {
        spmv_ref64<off_t, val_t>(n_rows, Ap, Aj, Ax, x, y64, yabs);
        return;
    }
    const off_t nnz = Ap[n_rows];
    std::vector<int32_t> cut(n_threads + 1);
    cut[0] = 0;
    cut[n_threads] = n_rows;
    for (int t = 1; t < n_threads; ++t) {
        off_t target = off_t((long double)nnz * t / n_threads);
        cut[t] = int32_t(std::lower_bound(Ap, Ap + n_rows, target) - Ap);
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) {
        pool.emplace_back([=] {
            const int32_t r0 = cut[t], r1 = cut[t + 1];
            spmv_ref64<off_t, val_t>(r1 - r0, Ap + r0, Aj, Ax, x, y64 + r0, yabs + r0);
        });
    }
    for (auto& th : pool) th.join();
}

// All-core variant of (1) for the reported CPU baseline (BASELINE.md §2b ii):
// static, nnz-balanced contiguous row chunks, each chunk running the serial
// loop above, so every y[row] is bit-identical to the 1-core result.
template <typename off_t, typename val_t>
void spmv_parallel(int32_t n_rows, const off_t* Ap, const int32_t* Aj,
                   const val_t* Ax, const val_t* x, val_t* y, int n_threads) {
    if (n_threads <= 1 || n_rows < n_threads) {
        spmv_serial<off_t, val_t>(0, n_rows, Ap, Aj, Ax, x, y);
        return;
    }
    const off_t nnz = Ap[n_rows];
    std::vector<int32_t> cut(n_threads + 1);
    cut[0] = 0;
    cut[n_threads] = n_rows;
    for (int t = 1; t < n_threads; ++t) {
        off_t target = off_t((long double)nnz * t / n_threads);
        cut[t] = int32_t(std::lower_bound(Ap, Ap + n_rows, target) - Ap);
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) {
        pool.emplace_back([=] {
            spmv_serial<off_t, val_t>(cut[t], cut[t + 1], Ap, Aj, Ax, x, y);
        });
    }
    for (auto& th : pool) th.join();
}

// ---------------------------------------------------------------------------
// (2) CSR-vector summation order — restates SURVEY Appendix A.1:
//     include/spmv/cusp/cusp_warp_reduce.cuh:26-57 (accumulate) and
//     include/spmv/cusp/utils.cuh:38-47 (shuffle-down tree).
//     T lanes own one row.  Lane l accumulates elements jj = start+l, +T, ...
//     in ascending order; when `aligned` (the reference does this only for
//     T == 32 and rows longer than 32, cusp_warp_reduce.cuh:33-44) the sweep
//     starts at start - (start mod T) and lane l owns jj == l (mod T).
//     Tree: for o = T/2 .. 1: s_l = s_l + s_{l+o}; lanes past the vector end
//     contribute whatever the neighbouring vector holds in hardware, but only
//     s_0 is stored, and s_0 depends only on lanes < T.
// ---------------------------------------------------------------------------
template <typename off_t, typename val_t>
void spmv_vector_order(int32_t n_rows, const off_t* Ap, const int32_t* Aj,
                       const val_t* Ax, const val_t* x, val_t* y, int T,
                       int aligned_when_longer_than) {
    std::vector<val_t> lane(T);
    for (int32_t row = 0; row < n_rows; ++row) {
        const off_t start = Ap[row], end = Ap[row + 1];
        std::fill(lane.begin(), lane.end(), val_t(0));
        const bool aligned =
            aligned_when_longer_than >= 0 && (end - start) > aligned_when_longer_than;
        for (int l = 0; l < T; ++l) {
            off_t jj = aligned ? (start - (start & off_t(T - 1)) + l) : (start + l);
            for (; jj < end; jj += T) {
                if (jj >= start) lane[l] = lane[l] + Ax[jj] * x[Aj[jj]];
            }
        }
        for (int o = T / 2; o >= 1; o >>= 1) {
            for (int l = 0; l + o < T; ++l) {
                // lanes >= o are not read again for s_0, order inside a level
                // is irrelevant because each level reads only pre-level values
                // of lanes l+o > l processed later in this ascending loop.
                lane[l] = lane[l] + lane[l + o];
            }
        }
        y[row] = lane[0];
    }
}

// ---------------------------------------------------------------------------
// (3) Merge-path decomposition — restates SURVEY Appendix A.2:
//     search      include/spmv/merge_based/thread_search.cuh:15-49
//     tile math   include/spmv/merge_based/dispatch_spmv_orig.cuh:613-623
//     tile body   include/spmv/merge_based/agent_spmv_orig.cuh:454-679
//     carry-out   include/spmv/merge_based/agent_spmv_orig.cuh:736-756
//     fix-up      include/spmv/merge_based/agent_segment_fixup.cuh:228-271
//     row_end[i] = Ap[i+1] (include/spmv/merge_based/device_spmv.cuh:153).
//     The list B is the counting sequence 0..nnz-1, so b[j] == b0 + j.
// ---------------------------------------------------------------------------
template <typename off_t>
void merge_search(int64_t diag, const off_t* a, int64_t a_len, int64_t b0,
                  int64_t b_len, int64_t* out_x, int64_t* out_y) {
    int64_t lo = std::max<int64_t>(diag - b_len, 0);
    int64_t hi = std::min<int64_t>(diag, a_len);
    while (lo < hi) {
        int64_t p = (lo + hi) >> 1;
        if (int64_t(a[p]) <= b0 + (diag - p - 1)) lo = p + 1;
        else hi = p;
    }
    *out_x = std::min<int64_t>(lo, a_len);
    *out_y = diag - lo;
}

template <typename off_t>
void merge_tile_coords(int32_t n_rows, const off_t* Ap, int64_t tile_items,
                       int64_t n_tiles, int64_t* out_x, int64_t* out_y) {
    const int64_t nnz = int64_t(Ap[n_rows]);
    const int64_t items = int64_t(n_rows) + nnz;
    for (int64_t t = 0; t <= n_tiles; ++t) {
        int64_t d = std::min<int64_t>(t * tile_items, items);
        merge_search<off_t>(d, Ap + 1, n_rows, 0, nnz, &out_x[t], &out_y[t]);
    }
}

// Full CPU simulation of the three-kernel pipeline with (block_threads, ipt).
template <typename off_t, typename val_t>
void spmv_merge_order(int32_t n_rows, const off_t* Ap, const int32_t* Aj,
                      const val_t* Ax, const val_t* x, val_t* y,
                      int block_threads, int ipt) {
    if (n_rows <= 0) return;
    const off_t* row_end = Ap + 1;
    const int64_t nnz = int64_t(Ap[n_rows]);
    const int64_t items = int64_t(n_rows) + nnz;
    const int64_t TILE = int64_t(block_threads) * ipt;
    const int64_t tiles = (items + TILE - 1) / TILE;
    struct KV { int64_t key; val_t value; };
    std::vector<KV> carry(tiles);
    std::vector<val_t> s_nz;
    std::vector<int64_t> s_re;
    std::vector<val_t> partial;
    std::vector<KV> seg(ipt), scan_in(block_threads), excl(block_threads);
    std::vector<std::vector<KV>> segs(block_threads, std::vector<KV>(ipt));

    for (int64_t t = 0; t < tiles; ++t) {
        int64_t x0, y0, x1, y1;
        merge_search<off_t>(std::min(t * TILE, items), row_end, n_rows, 0, nnz, &x0, &y0);
        merge_search<off_t>(std::min((t + 1) * TILE, items), row_end, n_rows, 0, nnz, &x1, &y1);
        const int64_t tr = x1 - x0, tn = y1 - y0;
        s_nz.assign(tn, val_t(0));
        for (int64_t j = 0; j < tn; ++j) s_nz[j] = Ax[y0 + j] * x[Aj[y0 + j]];
        s_re.assign(tr + ipt, 0);
        for (int64_t i = 0; i < tr + ipt; ++i)
            s_re[i] = int64_t(row_end[std::min<int64_t>(x0 + i, n_rows - 1)]);
        partial.assign(std::max<int64_t>(tr, 1), val_t(0));

        for (int th = 0; th < block_threads; ++th) {
            int64_t cx, cy;
            // per-thread diagonal inside the tile (agent_spmv_orig.cuh:557-563)
            {
                int64_t diag = int64_t(th) * ipt;
                int64_t lo = std::max<int64_t>(diag - tn, 0);
                int64_t hi = std::min<int64_t>(diag, tr);
                while (lo < hi) {
                    int64_t p = (lo + hi) >> 1;
                    if (s_re[p] <= y0 + (diag - p - 1)) lo = p + 1; else hi = p;
                }
                cx = std::min<int64_t>(lo, tr);
                cy = diag - lo;
            }
            val_t run = val_t(0);
            for (int k = 0; k < ipt; ++k) {
                val_t v;
                // clamp mirrors reading a zero product past the tile's nonzeros
                if (y0 + cy < s_re[cx]) {
                    v = (cy < tn) ? s_nz[cy] : val_t(0);
                    run = run + v;
                    ++cy;
                } else {
                    v = val_t(0);
                    run = val_t(0);
                    ++cx;
                }
                segs[th][k] = KV{cx, v};
            }
            scan_in[th] = KV{cx, run};
        }
        // exclusive scan with ReduceByKeyOp<Sum> (agent_spmv_orig.cuh:616-629)
        KV acc = scan_in[0];
        excl[0] = KV{0, val_t(0)};
        for (int th = 1; th < block_threads; ++th) {
            excl[th] = acc;
            KV b = scan_in[th];
            acc = KV{b.key, (acc.key == b.key) ? val_t(acc.value + b.value) : b.value};
        }
        KV tile_carry = acc;
        // thread 0 starts at tile-relative row 0 with no prefix (agent:632-635)
        {
            // recompute thread 0's start coordinate: diagonal 0 -> (0,0)
            excl[0] = KV{0, val_t(0)};
        }
        if (tr > 0) {
            for (int th = 0; th < block_threads; ++th) {
                std::vector<KV>& sg = segs[th];
                // running totals inside the thread: value at step k is the sum
                // of the current row so far.  The reference carries this in
                // scan_segment[] (agent:638-666); rebuild it here.
                if (excl[th].key != sg[0].key) {
                    if (excl[th].key < tr) partial[excl[th].key] = excl[th].value;
                } else {
                    sg[0].value = sg[0].value + excl[th].value;
                }
                for (int k = 1; k < ipt; ++k) {
                    if (sg[k - 1].key != sg[k].key) {
                        if (sg[k - 1].key < tr) partial[sg[k - 1].key] = sg[k - 1].value;
                    } else {
                        sg[k].value = sg[k].value + sg[k - 1].value;
                    }
                }
            }
            for (int64_t i = 0; i < tr; ++i) y[x0 + i] = partial[i];
        }
        KV c{tile_carry.key + x0, tile_carry.value};
        if (c.key >= n_rows) c = KV{int64_t(n_rows) - 1, val_t(0)};
        carry[t] = c;
    }
    if (tiles > 1) {
        for (int64_t t = 0; t < tiles; ++t) y[carry[t].key] = y[carry[t].key] + carry[t].value;
    }
}

// ---------------------------------------------------------------------------
// (4) Matrix Market ingest + CSR build — restates include/load.hpp:
//     banner        load.hpp:163-236   (five tokens, fields 2..5 lower-cased,
//                                        banner compared on its first 14 bytes)
//     size line     load.hpp:238-266   (skip lines starting with '%', then
//                                        "%zu %zu %zu"; blank line -> keep
//                                        scanning the stream for three numbers)
//     entries       load.hpp:317-360   (pattern -> 1.0; real/integer parsed as
//                                        double then cast; 1-based -> 0-based;
//                                        a zero index is an error)
//     symmetric     load.hpp:362-403   (entry then its mirror, interleaved, in
//                                        file order; diagonal kept once;
//                                        skew/hermitian NOT expanded)
//     CSR           load.hpp:420-474   (counting sort on row only: file order
//                                        kept inside a row, duplicates kept)
//     Error returns replace the reference's exit(1)/throw; the code says which.
// ---------------------------------------------------------------------------
enum {
    MTX_OK = 0,
    MTX_E_OPEN = 1,      // load.hpp:278-281  "File could not be opened"
    MTX_E_BANNER = 2,    // load.hpp:283-286  "Could not process Matrix Market banner"
    MTX_E_ARRAY = 3,     // load.hpp:289-292  "File is not a sparse matrix"
    MTX_E_SIZE = 4,      // load.hpp:296-300  "Could not read file info"
    MTX_E_OVERFLOW = 5,  // load.hpp:302-306  vertex_t / edge_t overflow
    MTX_E_ENTRY = 6,     // load.hpp:324-329, 346-351  short read / zero index
    MTX_E_TYPE = 7,      // load.hpp:357-360  "Unrecognized matrix market format type"
};

struct Csr {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    std::vector<int64_t> Ap;
    std::vector<int32_t> Aj;
    std::vector<double> Ax;  // held as the value type's exact image (see load)
    bool f64 = false;
};

struct Cursor {
    const char* p;
    const char* end;
    bool eof() const { return p >= end; }
    void skip_ws() { while (p < end && isspace((unsigned char)*p)) ++p; }
    // one line, like fgets(line, 1025, f): at most 1024 bytes
    std::string getline(bool* ok) {
        if (p >= end) { *ok = false; return {}; }
        const char* s = p;
        size_t n = 0;
        while (p < end && n < 1024) { char c = *p++; ++n; if (c == '\n') break; }
        *ok = true;
        return std::string(s, n);
    }
    // %zu : optional sign, decimal digits (strtoull semantics)
    bool scan_zu(size_t* out) {
        skip_ws();
        if (p >= end) return false;
        const char* q = p;
        bool neg = false;
        if (*q == '+' || *q == '-') { neg = (*q == '-'); ++q; }
        if (q >= end || !isdigit((unsigned char)*q)) return false;
        unsigned long long v = 0;
        while (q < end && isdigit((unsigned char)*q)) { v = v * 10ull + (unsigned)(*q - '0'); ++q; }
        p = q;
        *out = neg ? size_t(0) - size_t(v) : size_t(v);
        return true;
    }
    // %lf : strtod on a bounded copy of the token
    bool scan_lf(double* out) {
        skip_ws();
        if (p >= end) return false;
        char buf[512];
        size_t n = std::min<size_t>(sizeof(buf) - 1, size_t(end - p));
        memcpy(buf, p, n);
        buf[n] = 0;
        char* e = nullptr;
        double v = strtod(buf, &e);
        if (e == buf) return false;
        p += (e - buf);
        *out = v;
        return true;
    }
};

static void lower(std::string& s) { for (auto& c : s) c = char(tolower((unsigned char)c)); }

template <typename val_t>
int load_mtx(const char* path, int off_bits, Csr* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return MTX_E_OPEN;
    std::string buf;
    {
        char tmp[1 << 16];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, n);
    }
    fclose(f);
    Cursor c{buf.data(), buf.data() + buf.size()};

    // --- banner (load.hpp:163-236)
    bool ok;
    std::string line = c.getline(&ok);
    if (!ok) return MTX_E_BANNER;
    char tok[5][1025];
    if (sscanf(line.c_str(), "%1024s %1024s %1024s %1024s %1024s", tok[0], tok[1], tok[2], tok[3], tok[4]) != 5)
        return MTX_E_BANNER;
    std::string banner = tok[0], mtx = tok[1], crd = tok[2], dt = tok[3], st = tok[4];
    lower(mtx); lower(crd); lower(dt); lower(st);
    if (strncmp(banner.c_str(), "%%MatrixMarket", 14) != 0) return MTX_E_BANNER;
    if (mtx != "matrix") return MTX_E_BANNER;
    bool is_array;
    if (crd == "coordinate") is_array = false;
    else if (crd == "array") is_array = true;
    else return MTX_E_BANNER;
    char data;  // R C P I
    if (dt == "real") data = 'R';
    else if (dt == "complex") data = 'C';
    else if (dt == "pattern") data = 'P';
    else if (dt == "integer") data = 'I';
    else return MTX_E_BANNER;
    char scheme;  // G S H K
    if (st == "general") scheme = 'G';
    else if (st == "symmetric") scheme = 'S';
    else if (st == "hermitian") scheme = 'H';
    else if (st == "skew-symmetric") scheme = 'K';
    else return MTX_E_BANNER;
    if (is_array) return MTX_E_ARRAY;

    // --- size line (load.hpp:238-266)
    size_t M = 0, N = 0, NZ = 0;
    do {
        line = c.getline(&ok);
        if (!ok) return MTX_E_SIZE;
    } while (line[0] == '%');
    if (sscanf(line.c_str(), "%zu %zu %zu", &M, &N, &NZ) != 3) {
        // fscanf(f, "%zu %zu %zu") until it assigns three; a non-numeric token
        // makes the reference spin forever, reported here as a size error.
        size_t v[3];
        if (!(c.scan_zu(&v[0]) && c.scan_zu(&v[1]) && c.scan_zu(&v[2]))) return MTX_E_SIZE;
        M = v[0]; N = v[1]; NZ = v[2];
    }
    const uint64_t idx_max = 0x7fffffffull;  // index_t = int
    const uint64_t off_max = off_bits == 64 ? 0x7fffffffffffffffull : 0x7fffffffull;
    if (M >= idx_max || N >= idx_max) return MTX_E_OVERFLOW;
    if (NZ >= off_max) return MTX_E_OVERFLOW;

    // --- entries (load.hpp:317-360)
    std::vector<int32_t> I(NZ), J(NZ);
    std::vector<val_t> V(NZ);
    if (data == 'P') {
        for (size_t i = 0; i < NZ; ++i) {
            size_t r = 0, cc = 0;
            if (!(c.scan_zu(&r) && c.scan_zu(&cc))) return MTX_E_ENTRY;
            if (r == 0 || cc == 0) return MTX_E_ENTRY;
            I[i] = int32_t(r) - 1; J[i] = int32_t(cc) - 1; V[i] = val_t(1.0);
        }
    } else if (data == 'R' || data == 'I') {
        for (size_t i = 0; i < NZ; ++i) {
            size_t r = 0, cc = 0; double w = 0.0;
            if (!(c.scan_zu(&r) && c.scan_zu(&cc) && c.scan_lf(&w))) return MTX_E_ENTRY;
            if (r == 0 || cc == 0) return MTX_E_ENTRY;
            I[i] = int32_t(r) - 1; J[i] = int32_t(cc) - 1; V[i] = val_t(w);
        }
    } else {
        return MTX_E_TYPE;
    }

    // --- symmetric expansion (load.hpp:362-403)
    if (scheme == 'S') {
        std::vector<int32_t> nI, nJ; std::vector<val_t> nV;
        nI.reserve(2 * NZ); nJ.reserve(2 * NZ); nV.reserve(2 * NZ);
        for (size_t i = 0; i < NZ; ++i) {
            nI.push_back(I[i]); nJ.push_back(J[i]); nV.push_back(V[i]);
            if (I[i] != J[i]) { nI.push_back(J[i]); nJ.push_back(I[i]); nV.push_back(V[i]); }
        }
        I.swap(nI); J.swap(nJ); V.swap(nV);
    }
    const size_t nnz = I.size();

    // --- CSR build (load.hpp:420-474): count, exclusive scan, stable scatter
    out->n_rows = int64_t(M); out->n_cols = int64_t(N); out->nnz = int64_t(nnz);
    out->Ap.assign(M + 1, 0);
    out->Aj.resize(nnz);
    out->Ax.resize(nnz);
    out->f64 = sizeof(val_t) == 8;
    for (size_t n = 0; n < nnz; ++n) ++out->Ap[size_t(I[n]) + 1];
    for (size_t i = 0; i < M; ++i) out->Ap[i + 1] += out->Ap[i];
    std::vector<int64_t> next(out->Ap.begin(), out->Ap.end() - 1);
    for (size_t n = 0; n < nnz; ++n) {
        int64_t d = next[size_t(I[n])]++;
        out->Aj[size_t(d)] = J[n];
        out->Ax[size_t(d)] = double(V[n]);  // exact: float -> double is lossless
    }
    return MTX_OK;
}

}  // namespace

// ===========================================================================
// C entry points (ctypes).  Suffix = offset width _ value type.
// ===========================================================================
extern "C" {

#define ORACLE_TYPED(SUF, OFF, VAL)                                                            \
    void oracle_spmv_serial_##SUF(int32_t n_rows, const OFF* Ap, const int32_t* Aj,            \
                                  const VAL* Ax, const VAL* x, VAL* y) {                       \
        spmv_serial<OFF, VAL>(0, n_rows, Ap, Aj, Ax, x, y);                                    \
    }                                                                                          \
    void oracle_spmv_genl_serial_##SUF(int semiring, int32_t n_rows, const OFF* Ap, const int32_t* Aj,  \
                                       const VAL* Ax, const VAL* x, VAL* y) {                  \
        spmv_genl_serial<OFF, VAL>(semiring, n_rows, Ap, Aj, Ax, x, y);                        \
    }                                                                                          \
    void oracle_spmv_parallel_##SUF(int32_t n_rows, const OFF* Ap, const int32_t* Aj,          \
                                    const VAL* Ax, const VAL* x, VAL* y, int n_threads) {      \
        spmv_parallel<OFF, VAL>(n_rows, Ap, Aj, Ax, x, y, n_threads);                          \
    }                                                                                          \
    void oracle_spmv_ref64_##SUF(int32_t n_rows, const OFF* Ap, const int32_t* Aj,             \
                                 const VAL* Ax, const VAL* x, double* y64, double* yabs) {     \
        spmv_ref64<OFF, VAL>(n_rows, Ap, Aj, Ax, x, y64, yabs);                                \
    }                                                                                          \
    void oracle_spmv_ref64_parallel_##SUF(int32_t n_rows, const OFF* Ap, const int32_t* Aj,    \
                                          const VAL* Ax, const VAL* x, double* y64,            \
                                          double* yabs, int n_threads) {                       \
        spmv_ref64_parallel<OFF, VAL>(n_rows, Ap, Aj, Ax, x, y64, yabs, n_threads);            \
    }                                                                                          \
    void oracle_spmv_vector_order_##SUF(int32_t n_rows, const OFF* Ap, const int32_t* Aj,      \
                                        const VAL* Ax, const VAL* x, VAL* y, int T,            \
                                        int aligned_when_longer_than) {                        \
        spmv_vector_order<OFF, VAL>(n_rows, Ap, Aj, Ax, x, y, T, aligned_when_longer_than);    \
    }                                                                                          \
    void oracle_spmv_merge_order_##SUF(int32_t n_rows, const OFF* Ap, const int32_t* Aj,       \
                                       const VAL* Ax, const VAL* x, VAL* y,                    \
                                       int block_threads, int ipt) {                           \
        spmv_merge_order<OFF, VAL>(n_rows, Ap, Aj, Ax, x, y, block_threads, ipt);              \
    }

// fp32 matrix under fp64 vectors (cpu_navie.hpp:5-17 instantiated <float, double, double>): the product is taken in
// the promoted type (double), the sum kept in vec_y_value_t (double), serial order.
#define ORACLE_MIXED(SUF, OFF)                                                                 \
    void oracle_spmv_serial_mixed_##SUF(int32_t n_rows, const OFF* Ap, const int32_t* Aj,      \
                                        const float* Ax, const double* x, double* y) {         \
        for (int32_t row = 0; row < n_rows; ++row) {                                           \
            double sum = 0.0;                                                                  \
            for (OFF k = Ap[row]; k < Ap[row + 1]; ++k) sum += Ax[k] * x[Aj[k]];               \
            y[row] = sum;                                                                      \
        }                                                                                      \
    }
ORACLE_MIXED(i32, int32_t)
ORACLE_MIXED(i64, int64_t)

void oracle_spmv_genl_serial_i32_i32(int semiring, int32_t n_rows, const int32_t* Ap, const int32_t* Aj,
                                     const int32_t* Ax, const int32_t* x, int32_t* y) {
    spmv_genl_serial_i32<int32_t>(semiring, n_rows, Ap, Aj, Ax, x, y);
}
void oracle_spmv_genl_serial_i64_i32(int semiring, int32_t n_rows, const int64_t* Ap, const int32_t* Aj,
                                     const int32_t* Ax, const int32_t* x, int32_t* y) {
    spmv_genl_serial_i32<int64_t>(semiring, n_rows, Ap, Aj, Ax, x, y);
}
ORACLE_TYPED(i32_f32, int32_t, float)
ORACLE_TYPED(i32_f64, int32_t, double)
ORACLE_TYPED(i64_f32, int64_t, float)
ORACLE_TYPED(i64_f64, int64_t, double)

void oracle_merge_tile_coords_i32(int32_t n_rows, const int32_t* Ap, int64_t tile_items,
                                  int64_t n_tiles, int64_t* out_x, int64_t* out_y) {
    merge_tile_coords<int32_t>(n_rows, Ap, tile_items, n_tiles, out_x, out_y);
}
void oracle_merge_tile_coords_i64(int32_t n_rows, const int64_t* Ap, int64_t tile_items,
                                  int64_t n_tiles, int64_t* out_x, int64_t* out_y) {
    merge_tile_coords<int64_t>(n_rows, Ap, tile_items, n_tiles, out_x, out_y);
}

// Loader: two-step (load into a handle, query sizes, copy out, free).
void* oracle_mtx_load(const char* path, int off_bits, int val_is_f64, int* status) {
    Csr* c = new Csr();
    int st = val_is_f64 ? load_mtx<double>(path, off_bits, c) : load_mtx<float>(path, off_bits, c);
    *status = st;
    if (st != MTX_OK) { delete c; return nullptr; }
    return c;
}
void oracle_mtx_dims(void* h, int64_t* n_rows, int64_t* n_cols, int64_t* nnz) {
    Csr* c = static_cast<Csr*>(h);
    *n_rows = c->n_rows; *n_cols = c->n_cols; *nnz = c->nnz;
}
// Ap as int64, Aj as int32, Ax as double (exact image of the value type).
void oracle_mtx_copy(void* h, int64_t* Ap, int32_t* Aj, double* Ax) {
    Csr* c = static_cast<Csr*>(h);
    memcpy(Ap, c->Ap.data(), c->Ap.size() * sizeof(int64_t));
    if (c->nnz) {
        memcpy(Aj, c->Aj.data(), c->Aj.size() * sizeof(int32_t));
        memcpy(Ax, c->Ax.data(), c->Ax.size() * sizeof(double));
    }
}
void oracle_mtx_free(void* h) { delete static_cast<Csr*>(h); }

int oracle_hardware_threads(void) { return int(std::thread::hardware_concurrency()); }

}  // extern "C"
