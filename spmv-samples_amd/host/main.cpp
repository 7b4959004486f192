// main.cpp — the benchmark harness of peakcrosser7/spmv-samples for an MI355X box
// (SURVEY §8(f)-2).  Same command line, same two tables, same format strings as the
// reference's main.cu:21-124, so that the outputs of the two can be diffed:
//
//     ./bin/spmv <filename.mtx> <SpMV_kind_string>...
//
//   Dataset: <file name>                                   main.cu:38-39
//       n_rows: R  n_cols: C  nnz: Z
//   Compute delta:                                          main.cu:83-97
//   [kind        ] sum: %12lf  avg: %12lf
//   Time cost:                                              main.cu:100-113
//   [kind        ] total: %12lf ms  kernel: %12lf ms
//
// x = 1 (main.cu:41), the CPU result comes from the serial host loop (main.cu:78-81),
// every kind is called through SpMV(kind, ...) on device arrays owned by the harness
// (main.cu:48-74), timing is the mean of TEST_TIMES = 2000 calls read from
// Timer::total_cost / kernel_cost (main.cu:102-113).
//
// Differences, all behind options that come AFTER the kinds and start with "--":
//   --iters N        TEST_TIMES (default 2000, main.cu:19)
//   --dtype f64      value_t = double           (default float,  main.cu:17)
//   --offset 64      offset_t = 64-bit          (default int,    main.cu:16; the reference
//                                                harness cannot express this)
//   --no-poison      keep y between kinds.  By default y is filled with NaN before every
//                    kind: the reference leaves the previous kind's result in dY, so a kind
//                    that skips rows looks correct
//   --unit-us        label the time columns "us": the reference prints microseconds under
//                    an "ms" label (timer.hpp:10 vs main.cu:111-112); default keeps its label
//   --ngpu N         GPUs the hip_dist_* kinds spread the rows over (default 1; the reference is
//                    single-device, main.cu:53); --sub-blocks S = row blocks per GPU (default 4 when N > 1)
// and, since no SuiteSparse file is at hand offline (SURVEY §5 "config / flags", §8(d)), a seeded matrix in the file's place:
//     ./bin/spmv --synthetic band:n=4194304,k=32,w=4096 hip_vector hip_merge --seed 1
//     ./bin/spmv synthetic:rand:n=1048576,k=16 hip_merge
//   band: n rows, exactly k nonzeros per row, sorted columns sampled without replacement from [r - w, r + w] (the
//         S32-band family of SURVEY §8(d)); rand: k columns uniform over [0, n); values U(-1, 1); --seed S (default 1).
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

#include "load.hpp"
#include "spmv.h"
#include "spmv/cpu_check.hpp"

#define checkHipErr(val) CheckHipErr((val), #val, __FILE__, __LINE__)
static void CheckHipErr(hipError_t result, const char* func, const char* file, int line) {
    if (result != hipSuccess) {
        std::fprintf(stderr, "HIP error at %s:%d code=%d(%s) \"%s\" \n", file, line, int(result),
                     hipGetErrorName(result), func);
        std::abort();   // common.cuh:13-23
    }
}

struct Options {
    int iters = 2000;
    bool poison = true;
    bool unit_us = false;
    unsigned long long seed = 1;
};

// ---- seeded synthetic matrices (in the place of a Matrix Market file) ----------------------------------------
static unsigned long long mix(unsigned long long& s) {          // splitmix64
    s += 0x9e3779b97f4a7c15ull;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <typename index_t, typename offset_t, typename value_t>
static csr_t<index_t, offset_t, value_t> make_synthetic(const std::string& spec, unsigned long long seed) {
    const size_t colon = spec.find(':');
    const std::string family = spec.substr(0, colon);
    long long n = 1 << 20, k = 32, w = 4096;
    if (colon != std::string::npos) {
        std::string rest = spec.substr(colon + 1);
        size_t pos = 0;
        while (pos < rest.size()) {
            const size_t comma = rest.find(',', pos);
            const std::string item = rest.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            const size_t eq = item.find('=');
            if (eq == std::string::npos) { std::cerr << "synthetic spec: expected name=value, got " << item << std::endl; std::exit(1); }
            const long long v = std::atoll(item.substr(eq + 1).c_str());
            const std::string name = item.substr(0, eq);
            if (name == "n") n = v; else if (name == "k") k = v; else if (name == "w") w = v;
            else { std::cerr << "synthetic spec: unknown parameter " << name << std::endl; std::exit(1); }
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
    }
    const bool band = family == "band";
    if (!band && family != "rand") { std::cerr << "synthetic spec: family is band or rand" << std::endl; std::exit(1); }
    if (n < 1 || k < 1 || k > n || (band && 2 * w + 1 < k) || n >= (1ll << 31) - 1) {
        std::cerr << "synthetic spec: need 1 <= k <= n < 2^31 and, for band, 2 w + 1 >= k" << std::endl;
        std::exit(1);
    }
    if (n * k >= (long long)std::numeric_limits<offset_t>::max()) { std::cerr << "synthetic spec: n * k does not fit offset_t (use --offset 64)" << std::endl; std::exit(1); }
    csr_t<index_t, offset_t, value_t> csr;
    csr.number_of_rows = index_t(n);
    csr.number_of_columns = index_t(n);
    csr.number_of_nonzeros = offset_t(n * k);
    csr.row_offsets.resize(size_t(n) + 1);
    csr.column_indices.resize(size_t(n * k));
    csr.nonzero_values.resize(size_t(n * k));
    std::vector<long long> cols(static_cast<size_t>(k));
    for (long long r = 0; r <= n; ++r) csr.row_offsets[size_t(r)] = offset_t(r * k);
    for (long long r = 0; r < n; ++r) {
        unsigned long long s = seed * 0x2545f4914f6cdd1dull + (unsigned long long)r;
        if (band) {
            // k sorted columns without replacement from [lo, hi]: stratified, one per stratum, then clipped windows shift inwards
            long long lo = r - w, hi = r + w;
            if (lo < 0) { hi = std::min(n - 1, hi - lo); lo = 0; }
            if (hi > n - 1) { lo = std::max(0ll, lo - (hi - (n - 1))); hi = n - 1; }
            const long long span = hi - lo + 1;
            for (long long i = 0; i < k; ++i) {
                const long long a = lo + span * i / k, b = lo + span * (i + 1) / k;      // stratum [a, b), b > a since span >= k
                cols[size_t(i)] = a + (long long)(mix(s) % (unsigned long long)(b - a));
            }
        } else {
            for (long long i = 0; i < k; ++i) cols[size_t(i)] = (long long)(mix(s) % (unsigned long long)n);   // (file order: unsorted, duplicates kept)
        }
        for (long long i = 0; i < k; ++i) {
            csr.column_indices[size_t(r * k + i)] = index_t(cols[size_t(i)]);
            csr.nonzero_values[size_t(r * k + i)] = value_t(double(mix(s) >> 11) * (2.0 / 9007199254740992.0) - 1.0);
        }
    }
    return csr;
}

template <typename index_t, typename offset_t, typename value_t>
static int run(const char* path, const std::vector<std::string>& kinds, const Options& opt) {
    const bool synthetic = std::strncmp(path, "synthetic:", 10) == 0;
    csr_t<index_t, offset_t, value_t> csr = synthetic ? make_synthetic<index_t, offset_t, value_t>(path + 10, opt.seed)
                                                      : ToCsr(LoadCoo<index_t, offset_t, value_t>(path));
    const index_t n_rows = csr.number_of_rows, n_cols = csr.number_of_columns;
    const offset_t nnz = csr.number_of_nonzeros;
    std::cout << "Dataset: " << (synthetic ? std::string(path) + " seed=" + std::to_string(opt.seed) : std::filesystem::path(path).filename().string()) << std::endl
              << "\tn_rows: " << n_rows << "  n_cols: " << n_cols << "  nnz: " << nnz << std::endl;

    std::vector<value_t> vec_x(size_t(n_cols), value_t(1));
    std::vector<value_t> vec_y(size_t(n_rows), value_t(0));

    offset_t* dAp; index_t* dAj; value_t *dAx, *dX, *dY;
    checkHipErr(hipSetDevice(0));   // USED_DEVICE, common.cuh:8
    checkHipErr(hipMalloc((void**)&dAp, (size_t(n_rows) + 1) * sizeof(offset_t)));
    checkHipErr(hipMalloc((void**)&dAj, (size_t(nnz) + 1) * sizeof(index_t)));
    checkHipErr(hipMalloc((void**)&dAx, (size_t(nnz) + 1) * sizeof(value_t)));
    checkHipErr(hipMalloc((void**)&dX, (size_t(n_cols) + 1) * sizeof(value_t)));
    checkHipErr(hipMalloc((void**)&dY, (size_t(n_rows) + 1) * sizeof(value_t)));
    checkHipErr(hipMemcpy(dAp, csr.row_offsets.data(), (size_t(n_rows) + 1) * sizeof(offset_t), hipMemcpyHostToDevice));
    checkHipErr(hipMemcpy(dAj, csr.column_indices.data(), size_t(nnz) * sizeof(index_t), hipMemcpyHostToDevice));
    checkHipErr(hipMemcpy(dAx, csr.nonzero_values.data(), size_t(nnz) * sizeof(value_t), hipMemcpyHostToDevice));
    checkHipErr(hipMemcpy(dX, vec_x.data(), size_t(n_cols) * sizeof(value_t), hipMemcpyHostToDevice));
    checkHipErr(hipMemcpy(dY, vec_y.data(), size_t(n_rows) * sizeof(value_t), hipMemcpyHostToDevice));

    // CPU SpMV baseline (main.cu:76-81)
    std::vector<value_t> correct_y(size_t(n_rows), value_t(0));
    SpMV_cpu_navie(n_rows, n_cols, nnz, csr.row_offsets.data(), csr.column_indices.data(),
                   csr.nonzero_values.data(), vec_x.data(), correct_y.data());

    std::printf("Compute delta:\n");
    const std::vector<value_t> poison(size_t(n_rows), std::numeric_limits<value_t>::quiet_NaN());
    for (const auto& kind : kinds) {
        if (opt.poison && n_rows > 0)
            checkHipErr(hipMemcpy(dY, poison.data(), size_t(n_rows) * sizeof(value_t), hipMemcpyHostToDevice));
        SpMV(kind, n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
        checkHipErr(hipMemcpy(vec_y.data(), dY, size_t(n_rows) * sizeof(value_t), hipMemcpyDeviceToHost));
        double delta = 0.;
        for (index_t i = 0; i < n_rows; ++i) delta += std::abs(correct_y[i] - vec_y[i]);
        std::printf("[%-12s] sum: %12lf  avg: %12lf\n", kind.data(), delta, delta / n_rows);
    }
    std::printf("\n");

    std::printf("Time cost:\n");
    for (const auto& kind : kinds) {
        int64_t total_time = 0, kernel_time = 0;
        for (int i = 0; i < opt.iters; ++i) {
            SpMV(kind, n_rows, n_cols, nnz, dAp, dAj, dAx, dX, dY);
            total_time += Timer::total_cost();
            kernel_time += Timer::kernel_cost();
        }
        if (opt.unit_us)
            std::printf("[%-12s] total: %12lf us  kernel: %12lf us\n", kind.data(), 1. * total_time / opt.iters,
                        1. * kernel_time / opt.iters);
        else
            std::printf("[%-12s] total: %12lf ms  kernel: %12lf ms\n", kind.data(), 1. * total_time / opt.iters,
                        1. * kernel_time / opt.iters);
    }
    mi355_host::dist_release();   // (the multi-GPU kinds keep their handle between calls)
    checkHipErr(hipFree(dAp)); checkHipErr(hipFree(dAj)); checkHipErr(hipFree(dAx));
    checkHipErr(hipFree(dX)); checkHipErr(hipFree(dY));
    return EXIT_SUCCESS;
}

int main(int argc, char** argv) {
    std::vector<std::string> kinds;
    Options opt;
    std::string dtype = "f32", offset = "32", dataset = argc > 1 ? argv[1] : "";
    int first = 2;
    if (dataset == "--synthetic") {                       // `--synthetic <spec>` in the file's place
        if (argc < 3) { std::cerr << "--synthetic needs a spec (band:n=..,k=..,w=.. | rand:n=..,k=..)" << std::endl; std::exit(1); }
        dataset = std::string("synthetic:") + argv[2];
        first = 3;
    }
    for (int i = first; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char* name) -> std::string {
            if (i + 1 >= argc) { std::cerr << name << " needs a value" << std::endl; std::exit(1); }
            return argv[++i];
        };
        if (a == "--iters") opt.iters = std::max(1, std::atoi(value("--iters").c_str()));
        else if (a == "--dtype") dtype = value("--dtype");
        else if (a == "--offset") offset = value("--offset");
        else if (a == "--ngpu") mi355_host::dist_gpus() = std::max(1, std::atoi(value("--ngpu").c_str()));
        else if (a == "--sub-blocks") mi355_host::dist_sub_blocks() = std::max(1, std::atoi(value("--sub-blocks").c_str()));
        else if (a == "--no-poison") opt.poison = false;
        else if (a == "--unit-us") opt.unit_us = true;
        else if (a == "--seed") opt.seed = std::strtoull(value("--seed").c_str(), nullptr, 10);
        else kinds.push_back(a);
    }
    if (argc < 3 || kinds.empty()) {
        std::cerr << "usage: ./bin/<program-name>  <filename.mtx>  <SpMV_kind_string>..." << std::endl;
        std::exit(1);
    }
    const bool f64 = dtype == "f64", o64 = offset == "64";
    if (!f64 && !o64) return run<int, int, float>(dataset.c_str(), kinds, opt);
    if (f64 && !o64) return run<int, int, double>(dataset.c_str(), kinds, opt);
    if (!f64 && o64) return run<int, long long, float>(dataset.c_str(), kinds, opt);
    return run<int, long long, double>(dataset.c_str(), kinds, opt);
}
