// spmv.h — the operator boundary, MI355X build.
//
// Contract of the reference's include/spmv.h:18-48, kept: one SPMV_KINDS X-macro of
// (label, function) rows; SpMV<index_t, offset_t, mat_value_t, vec_x_value_t, vec_y_value_t>(
// kind_str, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y) finds the row whose label equals kind_str
// (first match in macro order), brackets the call with Timer::total_start/stop, and on an
// unknown label prints the reference's message and exits with EXIT_FAILURE (spmv.h:46-47).
// How the macro is expanded differs (a table of captureless lambdas instead of an if-chain);
// what a caller or a new kind sees does not: a kind is still a function template
// `SpMV_<name>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y)` plus one X row (README.md:28-45).
//
// In the reference tree the MI355X kinds register ALONGSIDE the CUDA kinds by adding
// `#include "spmv/mi355.hpp"` and the X rows below to its SPMV_KINDS.  This
// stand-alone copy lists only the kinds that exist on an MI355X box; the CUDA / cuSPARSE / CUB
// kinds of the reference (spmv.h:19-27) have no place on this hardware.
#pragma once

#include <cstdlib>
#include <iostream>
#include <string>

#include "spmv/mi355.hpp"
#include "spmv/rocsparse_cmp.hpp"
#include "timer.hpp"

// "hip_functor": the general path next to the tuned kinds in the harness's tables — the ordinary (+, *) written as a
// functor's TEXT (TimesThenPlus, spmv/mi355.hpp), the way a user of the reference's generalized kind writes one
// (merge_genl.cuh:19-38), compiled for gfx950 on first use (SpMV_hip_functor).
template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t, typename vec_y_value_t>
void SpMV_hip_functor_times_plus(index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap, const index_t* Aj,
                                 const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    SpMV_hip_functor<TimesThenPlus_text>(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);
}

// Vendor comparison columns (the place of "cusparse" in the reference's list, spmv.h:19): present
// only in a build with -DMI355_WITH_ROCSPARSE, never part of the engine.
#ifdef MI355_WITH_ROCSPARSE
#define SPMV_KINDS_VENDOR                             \
    X("rocsparse", SpMV_rocsparse)                    \
    X("rocsparse_stream", SpMV_rocsparse_stream)      \
    X("cusparse", SpMV_rocsparse)
#else
#define SPMV_KINDS_VENDOR
#endif

// The reference's own labels (spmv.h:18-27), so a command line or script written for it runs unchanged: each
// names the MI355X kind that stands in its place (INTEGRATION.md has the table).  The three CUSP variants differ
// in how a warp reads and reduces a row, the two LightSpMV ones in vector/warp granularity of the row counter,
// cub_merge / merge in who wrote the merge-path kernel: on gfx950 each family is ONE kernel set whose lanes per
// row and rows per fetch the analysis picks.  "cusparse" names the vendor column and exists only in a build with
// -DMI355_WITH_ROCSPARSE (above).
#define SPMV_KINDS_REFERENCE_LABELS                   \
    X("cusp", SpMV_hip_vector)                        \
    X("cusp1", SpMV_hip_vector)                       \
    X("cusp2", SpMV_hip_vector)                       \
    X("light_vec", SpMV_hip_light)                    \
    X("light_warp", SpMV_hip_light)                   \
    X("cub_merge", SpMV_hip_merge)                    \
    X("merge", SpMV_hip_merge)                        \
    X("merge_genl", SpMV_hip_merge_generalized)

/// SPMV kind strings and its function
#define SPMV_KINDS                                    \
    X("hip_vector", SpMV_hip_vector)                  \
    X("hip_merge", SpMV_hip_merge)                    \
    X("hip_light", SpMV_hip_light)                    \
    X("hip_auto", SpMV_hip_auto)                      \
    X("hip_merge_genl", SpMV_hip_merge_generalized)   \
    X("hip_functor", SpMV_hip_functor_times_plus)     \
    X("hip_dist_vector", SpMV_hip_dist_vector)        \
    X("hip_dist_merge", SpMV_hip_dist_merge)          \
    X("hip_dist_light", SpMV_hip_dist_light)          \
    SPMV_KINDS_REFERENCE_LABELS                       \
    SPMV_KINDS_VENDOR

template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t,
          typename vec_y_value_t>
void SpMV(const std::string& kind_str, index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap,
          const index_t* Aj, const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    using kind_fn = void (*)(index_t, index_t, offset_t, const offset_t*, const index_t*, const mat_value_t*,
                             const vec_x_value_t*, vec_y_value_t*);
    struct kind_row {
        const char* label;
        kind_fn call;
    };
    static const kind_row rows[] = {
#define X(label, func)                                                                                   \
    {label, [](index_t r, index_t c, offset_t z, const offset_t* p, const index_t* j, const mat_value_t* a, \
               const vec_x_value_t* xv, vec_y_value_t* yv) { func(r, c, z, p, j, a, xv, yv); }},
        SPMV_KINDS
#undef X
    };
    for (const kind_row& row : rows) {
        if (kind_str != row.label) continue;
        Timer::total_start();
        row.call(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y);
        Timer::total_stop();
        return;
    }
    std::cerr << "SpMV kind \"" << kind_str << "\" is NOT SUPPROT\n";
    std::exit(EXIT_FAILURE);
}
