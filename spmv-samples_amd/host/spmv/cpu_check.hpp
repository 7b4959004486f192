// spmv/cpu_check.hpp — the harness's CPU reference, NOT a kind.
//
// The reference harness prints, for every kind, the distance of the device result to a
// serial host SpMV (main.cu:76-96, function include/spmv/cpu_navie.hpp:5-17).  This is
// that host loop for the MI355X harness (host/main.cpp).  It is deliberately not listed
// in SPMV_KINDS — exactly as in the reference (spmv.h:18-27) — and nothing in the
// library or in the SpMV<> dispatch can reach it: it only produces the "correct_y" the
// "Compute delta" table is measured against.
#pragma once

template <typename index_t, typename offset_t, typename mat_value_t, typename vec_x_value_t,
          typename vec_y_value_t>
void SpMV_cpu_navie(index_t n_rows, index_t /*n_cols*/, offset_t /*nnz*/, const offset_t* Ap, const index_t* Aj,
                    const mat_value_t* Ax, const vec_x_value_t* x, vec_y_value_t* y) {
    for (index_t r = 0; r < n_rows; ++r) {
        vec_y_value_t acc = vec_x_value_t(0);
        for (offset_t k = Ap[r]; k < Ap[r + 1]; ++k) acc += Ax[k] * x[Aj[k]];
        y[r] = acc;   // empty row -> 0
    }
}
