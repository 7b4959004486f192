// load.hpp — Matrix Market ingest + CSR build for the MI355X harness.
//
// Same public surface and the same results as the reference's include/load.hpp:
//   coo_t / csr_t                      load.hpp:131-161  (same member names)
//   LoadCoo<index_t,offset_t,value_t>  load.hpp:268-408
//   ToCsr                              load.hpp:420-474
// "Same results" is meant entry for entry (tests/test_host_loader.py compares with the
// reference build and with tests/golden/golden.json): 1-based -> 0-based; pattern -> 1.0;
// real/integer parsed as double then cast; `symmetric` expanded entry-then-mirror in file
// order, diagonal once; skew-symmetric / hermitian parsed but NOT expanded; complex
// rejected; CSR = counting sort on the row only, so file order survives inside a row and
// duplicates are kept.  Error behaviour is the reference's too: exit(1) with its message
// for open/banner/array/size/type failures (load.hpp:278-300, :357-360), exception_t for
// overflow and malformed entries (load.hpp:302-306, :324-329, :346-351).
//
// What is different is HOW (the reference reads one entry per fscanf call on one thread
// and counts with index_t, load.hpp:321-355, :448-470 — minutes for nlpkkt160's 119 M
// lines, and an overflow at 2^31 entries whatever offset_t is):
//   * the file is mmap'ed and tokenised in parallel: fscanf is token-, not line-based, so
//     chunk boundaries are aligned to ENTRY boundaries by a token census (pass 1) before
//     the parse (pass 2) — a file whose entries straddle lines parses as the reference does;
//   * numbers: unsigned decimal by hand; doubles by the exact fast path (<= 15 significant
//     digits, |exp10| <= 22: one correctly rounded multiply/divide) with strtod as the
//     fallback, so every value equals strtod's bit for bit;
//   * all counters are 64-bit;
//   * SaveCsrBinary / LoadCsrBinary: a raw cache of the CSR arrays so a large matrix is
//     parsed once.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

struct exception_t : std::exception {
    std::string report;
    explicit exception_t(std::string message = "") : report(std::move(message)) {}
    const char* what() const noexcept override { return report.c_str(); }
};

inline void throw_if_exception(bool is_exception, const std::string& message = "") {
    if (is_exception) throw exception_t(message);
}

template <typename index_t, typename offset_t, typename value_t>
struct coo_t {
    coo_t(index_t n_rows, index_t n_cols, offset_t nnz)
        : number_of_rows(n_rows), number_of_columns(n_cols), number_of_nonzeros(nnz),
          row_indices(size_t(nnz)), column_indices(size_t(nnz)), nonzero_values(size_t(nnz)) {}
    index_t number_of_rows;
    index_t number_of_columns;
    offset_t number_of_nonzeros;
    std::vector<index_t> row_indices;
    std::vector<index_t> column_indices;
    std::vector<value_t> nonzero_values;
};

template <typename index_t, typename offset_t, typename value_t>
struct csr_t {
    using index_type = index_t;
    using offset_type = offset_t;
    using value_type = value_t;
    index_t number_of_rows = 0;
    index_t number_of_columns = 0;
    offset_t number_of_nonzeros = 0;
    std::vector<offset_t> row_offsets;    // Ap
    std::vector<index_t> column_indices;  // Aj
    std::vector<value_t> nonzero_values;  // Ax
};

namespace mm_detail {

inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

struct Mapped {
    const char* data = nullptr;
    size_t size = 0;
    int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        size = size_t(st.st_size);
        if (size == 0) { data = ""; return true; }
        void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (p == MAP_FAILED) return false;
        data = static_cast<const char*>(p);
        madvise(p, size, MADV_SEQUENTIAL);
        return true;
    }
    ~Mapped() {
        if (data && size) munmap(const_cast<char*>(data), size);
        if (fd >= 0) close(fd);
    }
};

// one line as fgets(line, 1025, f) would deliver it: at most 1024 bytes, newline included
inline bool get_line(const char*& p, const char* end, std::string& line) {
    if (p >= end) return false;
    const char* s = p;
    size_t n = 0;
    while (p < end && n < 1024) {
        const char c = *p++;
        ++n;
        if (c == '\n') break;
    }
    line.assign(s, n);
    return true;
}

inline void lower(std::string& s) {
    for (auto& c : s) c = char(std::tolower(static_cast<unsigned char>(c)));
}

// "%zu": optional sign, decimal digits, wraps like strtoull
inline bool parse_zu(const char*& p, const char* end, size_t& out) {
    while (p < end && is_space(*p)) ++p;
    const char* q = p;
    bool neg = false;
    if (q < end && (*q == '+' || *q == '-')) { neg = (*q == '-'); ++q; }
    if (q >= end || *q < '0' || *q > '9') return false;
    unsigned long long v = 0;
    while (q < end && *q >= '0' && *q <= '9') { v = v * 10ull + unsigned(*q - '0'); ++q; }
    p = q;
    out = neg ? size_t(0) - size_t(v) : size_t(v);
    return true;
}

// "%lf": exact fast path, strtod otherwise — always strtod's value
inline bool parse_lf(const char*& p, const char* end, double& out) {
    while (p < end && is_space(*p)) ++p;
    if (p >= end) return false;
    const char* q = p;
    bool neg = false;
    if (*q == '+' || *q == '-') { neg = (*q == '-'); ++q; }
    uint64_t mant = 0;
    int digits = 0, exp10 = 0;
    bool any = false, fast = true;
    while (q < end && *q >= '0' && *q <= '9') {
        any = true;
        if (mant || *q != '0') { if (digits < 19) { mant = mant * 10 + unsigned(*q - '0'); ++digits; } else { fast = false; ++exp10; } }
        ++q;
    }
    if (q < end && *q == '.') {
        ++q;
        while (q < end && *q >= '0' && *q <= '9') {
            any = true;
            if (mant || *q != '0') { if (digits < 19) { mant = mant * 10 + unsigned(*q - '0'); ++digits; --exp10; } else { fast = false; } }
            else --exp10;
            ++q;
        }
    }
    if (!any) fast = false;   // inf / nan / hex / garbage: let strtod decide
    if (any && q < end && (*q == 'e' || *q == 'E')) {
        const char* r = q + 1;
        bool eneg = false;
        if (r < end && (*r == '+' || *r == '-')) { eneg = (*r == '-'); ++r; }
        if (r < end && *r >= '0' && *r <= '9') {
            int ev = 0;
            while (r < end && *r >= '0' && *r <= '9') { if (ev < 100000) ev = ev * 10 + (*r - '0'); ++r; }
            exp10 += eneg ? -ev : ev;
            q = r;
        }
    }
    // a token that runs on into letters (e.g. "1.5abc", "0x10", "inf") is strtod's business
    if (q < end && !is_space(*q)) fast = false;
    static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    if (fast && digits <= 15 && exp10 >= -22 && exp10 <= 22) {
        double v = double(mant);                 // exact: < 2^53
        v = exp10 < 0 ? v / p10[-exp10] : v * p10[exp10];   // one correctly rounded operation
        out = neg ? -v : v;
        p = q;
        return true;
    }
    char buf[512];
    const size_t n = std::min<size_t>(sizeof(buf) - 1, size_t(end - p));
    std::memcpy(buf, p, n);
    buf[n] = 0;
    char* e = nullptr;
    const double v = std::strtod(buf, &e);
    if (e == buf) return false;
    p += (e - buf);
    out = v;
    return true;
}

inline size_t count_tokens(const char* b, const char* e, const char* file_begin) {
    // tokens that START in [b, e)
    size_t n = 0;
    bool prev_space = (b == file_begin) ? true : is_space(b[-1]);
    for (const char* p = b; p < e; ++p) {
        const bool sp = is_space(*p);
        if (!sp && prev_space) ++n;
        prev_space = sp;
    }
    return n;
}

// The reference prints its message and exits the process on an unreadable file, a bad banner, an array file, a bad
// size line or a complex field (load.hpp:278-300, :357-360): the harness keeps that.  A library cannot: built with
// MI355_LOAD_NO_EXIT (host/load_capi.cpp) the same sites throw fatal_t instead.
struct fatal_t : std::runtime_error { using std::runtime_error::runtime_error; };
[[noreturn]] inline void fatal(const std::string& message) {
#ifdef MI355_LOAD_NO_EXIT
    throw fatal_t(message);
#else
    std::cerr << message << std::endl;
    std::exit(1);
#endif
}

inline unsigned parse_threads() {
    if (const char* v = std::getenv("MI355_LOAD_THREADS")) {
        const int n = std::atoi(v);
        if (n > 0) return unsigned(n);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw ? std::min(hw, 64u) : 1u;
}

}  // namespace mm_detail

template <typename index_t, typename offset_t, typename value_t>
coo_t<index_t, offset_t, value_t> LoadCoo(std::string filename) {
    using namespace mm_detail;
    Mapped f;
    if (!f.open(filename)) fatal("File could not be opened: " + filename);
    const char* p = f.data;
    const char* const end = f.data + f.size;

    // ---- banner: five tokens, fields 2..5 case-insensitive
    auto banner_fail = [] { fatal("Could not process Matrix Market banner"); };
    std::string line;
    if (!get_line(p, end, line)) banner_fail();
    char t0[1025], t1[1025], t2[1025], t3[1025], t4[1025];
    if (std::sscanf(line.c_str(), "%1024s %1024s %1024s %1024s %1024s", t0, t1, t2, t3, t4) != 5) banner_fail();
    std::string object = t1, format = t2, field = t3, symmetry = t4;
    lower(object); lower(format); lower(field); lower(symmetry);
    if (std::strncmp(t0, "%%MatrixMarket", 14) != 0) banner_fail();
    if (object != "matrix") banner_fail();
    if (format != "coordinate" && format != "array") banner_fail();
    if (field != "real" && field != "complex" && field != "pattern" && field != "integer") banner_fail();
    if (symmetry != "general" && symmetry != "symmetric" && symmetry != "hermitian" && symmetry != "skew-symmetric")
        banner_fail();
    if (format == "array") fatal("File is not a sparse matrix");

    // ---- size line: first line not starting with '%'; blank -> next three numbers of the stream
    auto size_fail = [] { fatal("Could not read file info (M, N, NNZ)"); };
    size_t n_rows = 0, n_cols = 0, n_entries = 0;
    do {
        if (!get_line(p, end, line)) size_fail();
    } while (line[0] == '%');
    if (std::sscanf(line.c_str(), "%zu %zu %zu", &n_rows, &n_cols, &n_entries) != 3) {
        if (!(parse_zu(p, end, n_rows) && parse_zu(p, end, n_cols) && parse_zu(p, end, n_entries))) size_fail();
    }
    throw_if_exception(n_rows >= size_t(std::numeric_limits<index_t>::max()) ||
                           n_cols >= size_t(std::numeric_limits<index_t>::max()),
                       "vertex_t overflow");
    throw_if_exception(n_entries >= size_t(std::numeric_limits<offset_t>::max()), "edge_t overflow");

    const bool pattern = field == "pattern";
    if (!pattern && field != "real" && field != "integer") fatal("Unrecognized matrix market format type");
    coo_t<index_t, offset_t, value_t> coo{static_cast<index_t>(n_rows), static_cast<index_t>(n_cols),
                                          static_cast<offset_t>(n_entries)};

    // ---- entries, in parallel.  Pass 1: token census per chunk.  Pass 2: each chunk parses the
    //      entries whose FIRST token starts inside it.
    const size_t tokens_per_entry = pattern ? 2 : 3;
    const bool mirror = symmetry == "symmetric";
    const char* const body = p;
    const size_t body_size = size_t(end - body);
    unsigned n_thr = parse_threads();
    size_t chunk = 1u << 20;
    if (const char* v = std::getenv("MI355_LOAD_CHUNK")) chunk = std::max<size_t>(1, size_t(std::atoll(v)));
    const size_t n_chunks = std::max<size_t>(1, (body_size + chunk - 1) / chunk);
    n_thr = unsigned(std::min<size_t>(n_thr, n_chunks));
    std::vector<size_t> tok_begin(n_chunks + 1, 0);
    auto for_chunks = [&](auto&& fn) {
        if (n_thr <= 1) { for (size_t c = 0; c < n_chunks; ++c) fn(c); return; }
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_thr; ++t)
            pool.emplace_back([&, t] { for (size_t c = t; c < n_chunks; c += n_thr) fn(c); });
        for (auto& th : pool) th.join();
    };
    for_chunks([&](size_t c) {
        const char* b = body + c * chunk;
        const char* e = std::min(b + chunk, end);
        tok_begin[c + 1] = count_tokens(b, e, body);
    });
    for (size_t c = 0; c < n_chunks; ++c) tok_begin[c + 1] += tok_begin[c];

    std::vector<int> status(n_chunks, 0);   // 0 ok, 1 short read, 2 zero index, 3 index beyond the header's size
    for_chunks([&](size_t c) {
        const char* q = body + c * chunk;
        const char* e = std::min(q + chunk, end);
        // skip a token that started in the previous chunk
        if (c > 0 && !is_space(q[-1])) while (q < end && !is_space(*q)) ++q;
        size_t tok = tok_begin[c];
        // advance to the first token that opens an entry
        while (tok % tokens_per_entry != 0 && tok < tok_begin[c + 1]) {
            while (q < end && is_space(*q)) ++q;
            while (q < end && !is_space(*q)) ++q;
            ++tok;
        }
        size_t entry = tok / tokens_per_entry;
        while (tok < tok_begin[c + 1] && entry < n_entries) {
            size_t r = 0, cidx = 0;
            double w = 1.0;
            bool ok = parse_zu(q, end, r) && parse_zu(q, end, cidx);
            if (ok && !pattern) ok = parse_lf(q, end, w);
            if (!ok) { status[c] = 1; return; }
            if (r == 0 || cidx == 0) { status[c] = 2; return; }
            // The reference does not look (load.hpp:329-343 there) and then indexes out of bounds in ToCsr and,
            // through x[column], on the device: a file that passes here behaves exactly as it does there.
            if (r > n_rows || cidx > n_cols) { status[c] = 3; return; }
            if (mirror && (cidx > n_rows || r > n_cols)) { status[c] = 3; return; }   // (the expansion stores (c, r) too)
            coo.row_indices[entry] = index_t(r) - 1;
            coo.column_indices[entry] = index_t(cidx) - 1;
            coo.nonzero_values[entry] = pattern ? value_t(1.0) : value_t(w);
            ++entry;
            tok += tokens_per_entry;
        }
        (void)e;
    });
    const char* short_msg = pattern ? "Could not read edge from market file" : "Could not read weighted edge from market file";
    // the reference stops at the FIRST bad entry in file order
    const size_t total_tokens = tok_begin[n_chunks];
    for (size_t c = 0; c < n_chunks; ++c) {
        if (status[c] == 1) throw exception_t(short_msg);
        if (status[c] == 2) throw exception_t("Market file is zero-indexed");
        if (status[c] == 3) throw exception_t("Market file has an index beyond its declared size");
    }
    throw_if_exception(total_tokens / tokens_per_entry < n_entries, short_msg);

    // ---- symmetric: entry, then its mirror, in file order; diagonal once
    if (symmetry == "symmetric") {
        uint64_t off_diag = 0;
        for (size_t i = 0; i < n_entries; ++i) off_diag += coo.row_indices[i] != coo.column_indices[i];
        const uint64_t total = uint64_t(n_entries) + off_diag;
        std::vector<index_t> I(total), J(total);
        std::vector<value_t> V(total);
        uint64_t k = 0;
        for (size_t i = 0; i < n_entries; ++i) {
            const index_t r = coo.row_indices[i], c = coo.column_indices[i];
            const value_t v = coo.nonzero_values[i];
            I[k] = r; J[k] = c; V[k] = v; ++k;
            if (r != c) { I[k] = c; J[k] = r; V[k] = v; ++k; }
        }
        coo.row_indices.swap(I);
        coo.column_indices.swap(J);
        coo.nonzero_values.swap(V);
        coo.number_of_nonzeros = offset_t(total);
    }
    return coo;
}

// COO -> CSR: stable counting sort on the row index (file order kept inside a row,
// duplicates kept) — the result of the reference's ToCsr (load.hpp:420-474).
template <typename index_t, typename offset_t, typename value_t>
csr_t<index_t, offset_t, value_t> ToCsr(const coo_t<index_t, offset_t, value_t>& coo) {
    csr_t<index_t, offset_t, value_t> csr;
    csr.number_of_rows = coo.number_of_rows;
    csr.number_of_columns = coo.number_of_columns;
    csr.number_of_nonzeros = coo.number_of_nonzeros;
    const size_t n = size_t(coo.number_of_rows), nnz = size_t(coo.number_of_nonzeros);
    csr.row_offsets.assign(n + 1, offset_t(0));
    csr.column_indices.resize(nnz);
    csr.nonzero_values.resize(nnz);
    for (size_t k = 0; k < nnz; ++k) ++csr.row_offsets[size_t(coo.row_indices[k]) + 1];
    for (size_t r = 0; r < n; ++r) csr.row_offsets[r + 1] += csr.row_offsets[r];
    std::vector<offset_t> cursor(csr.row_offsets.begin(), csr.row_offsets.end() - 1);
    for (size_t k = 0; k < nnz; ++k) {
        const size_t dst = size_t(cursor[size_t(coo.row_indices[k])]++);
        csr.column_indices[dst] = coo.column_indices[k];
        csr.nonzero_values[dst] = coo.nonzero_values[k];
    }
    return csr;
}

// ---- binary CSR cache (not in the reference) ----------------------------------------------
template <typename index_t, typename offset_t, typename value_t>
bool SaveCsrBinary(const csr_t<index_t, offset_t, value_t>& csr, const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const uint64_t hdr[8] = {0x3535494d52534331ull /* "1CSRMI55" */, sizeof(index_t), sizeof(offset_t), sizeof(value_t),
                             uint64_t(csr.number_of_rows), uint64_t(csr.number_of_columns),
                             uint64_t(csr.number_of_nonzeros), 0};
    bool ok = std::fwrite(hdr, sizeof(hdr), 1, f) == 1;
    auto put = [&](const void* d, size_t bytes) { if (bytes) ok = ok && std::fwrite(d, 1, bytes, f) == bytes; };
    put(csr.row_offsets.data(), csr.row_offsets.size() * sizeof(offset_t));
    put(csr.column_indices.data(), csr.column_indices.size() * sizeof(index_t));
    put(csr.nonzero_values.data(), csr.nonzero_values.size() * sizeof(value_t));
    return std::fclose(f) == 0 && ok;
}

template <typename index_t, typename offset_t, typename value_t>
bool LoadCsrBinary(const std::string& path, csr_t<index_t, offset_t, value_t>& csr) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    uint64_t hdr[8];
    bool ok = std::fread(hdr, sizeof(hdr), 1, f) == 1 && hdr[0] == 0x3535494d52534331ull &&
              hdr[1] == sizeof(index_t) && hdr[2] == sizeof(offset_t) && hdr[3] == sizeof(value_t);
    if (ok) {
        csr.number_of_rows = index_t(hdr[4]);
        csr.number_of_columns = index_t(hdr[5]);
        csr.number_of_nonzeros = offset_t(hdr[6]);
        csr.row_offsets.resize(size_t(hdr[4]) + 1);
        csr.column_indices.resize(size_t(hdr[6]));
        csr.nonzero_values.resize(size_t(hdr[6]));
        auto get = [&](void* d, size_t bytes) { if (bytes) ok = ok && std::fread(d, 1, bytes, f) == bytes; };
        get(csr.row_offsets.data(), csr.row_offsets.size() * sizeof(offset_t));
        get(csr.column_indices.data(), csr.column_indices.size() * sizeof(index_t));
        get(csr.nonzero_values.data(), csr.nonzero_values.size() * sizeof(value_t));
    }
    std::fclose(f);
    return ok;
}
