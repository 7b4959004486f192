// spmv.h — the operator boundary, MI355X build.
//
// Same contract as the reference's include/spmv.h:18-48: one SPMV_KINDS X-macro of
// (label, function) rows, and SpMV<index_t, offset_t, mat_value_t, vec_x_value_t,
// vec_y_value_t>(kind_str, n_rows, n_cols, nnz, Ap, Aj, Ax, x, y) that expands the
// macro into a label-compare chain, brackets the call with Timer::total_start/stop
// and exits with the reference's message on an unknown label (spmv.h:46-47).
//
// In the reference tree the MI355X kinds register ALONGSIDE the CUDA kinds by
// adding `#include "spmv/mi355.hpp"` and the three X rows below to its SPMV_KINDS
// (INTEGRATION.md).  This stand-alone copy of the boundary lists only the kinds
// that exist on an MI355X box; the CUDA/cuSPARSE/CUB kinds of the reference
// (spmv.h:19-27) have no place on this hardware.
#pragma once

#include <cstdlib>
#include <iostream>
#include <string>

#include "timer.hpp"
#include "spmv/mi355.hpp"

/// SPMV kind strings and its function
#define SPMV_KINDS                                    \
    X("hip_vector", SpMV_hip_vector)                  \
    X("hip_merge", SpMV_hip_merge)                    \
    X("hip_light", SpMV_hip_light)                    \
    X("hip_merge_genl", SpMV_hip_merge_generalized)

template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t,
          typename vec_y_value_t>
void SpMV(const std::string& kind_str, index_t n_rows, index_t n_cols, offset_t nnz, const offset_t* Ap,
          const index_t* Aj, const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
#define X(label, func)                               \
    if (kind_str == label) {                         \
        Timer::total_start();                        \
        func(n_rows, n_cols, nnz, Ap, Aj, Ax, x, y); \
        Timer::total_stop();                         \
        return;                                      \
    }
    SPMV_KINDS
#undef X

    std::cerr << "SpMV kind \"" << kind_str << "\" is NOT SUPPROT\n";
    std::exit(EXIT_FAILURE);
}
