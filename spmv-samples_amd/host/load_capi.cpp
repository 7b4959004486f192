// load_capi.cpp — the extern "C" surface of include/mi355_load.h around host/load.hpp (LoadCoo + ToCsr, the
// reference's include/load.hpp:268-474 restated for speed): what bench.py --mtx and other non-C++ callers bind.
// The reference exits the process on a bad file (load.hpp:278-300); a library must not, so the loader runs with
// its exit-on-error switched to exceptions (MI355_LOAD_NO_EXIT) and everything becomes a status + message.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#define MI355_LOAD_NO_EXIT 1
#include "load.hpp"
#include "../../include/mi355_load.h"

struct mi355_csr_host {
    int off_type = 0, val_type = 0;
    csr_t<int, int, float> a;
    csr_t<int, int, double> b;
    csr_t<int, long long, float> c;
    csr_t<int, long long, double> d;
};

namespace {
thread_local char g_err[512] = "";
void set_err(const char* m) { std::snprintf(g_err, sizeof(g_err), "%s", m); }

template <typename Csr>
int fill(Csr& into, const char* path) {
    using off_t = typename std::remove_reference<decltype(into.row_offsets[0])>::type;
    using val_t = typename std::remove_reference<decltype(into.nonzero_values[0])>::type;
    try {
        into = ToCsr(LoadCoo<int, off_t, val_t>(std::string(path)));
        return MI355_LOAD_OK;
    } catch (const mm_detail::fatal_t& e) {
        set_err(e.what());
        return MI355_LOAD_EFILE;
    } catch (const exception_t& e) {
        set_err(e.what());
        return std::strstr(e.what(), "overflow") ? MI355_LOAD_ERANGE : MI355_LOAD_EPARSE;
    } catch (const std::bad_alloc&) {
        set_err("out of host memory");
        return MI355_LOAD_ERANGE;
    }
}
}  // namespace

extern "C" {

int mi355_load_mtx(const char* path, int off_type, int val_type, mi355_csr_host** out) {
    g_err[0] = 0;
    if (!path || !out || (off_type != 0 && off_type != 1) || (val_type != 0 && val_type != 1)) {
        set_err("mi355_load_mtx: null pointer or unknown type");
        return MI355_LOAD_EINVAL;
    }
    *out = nullptr;
    mi355_csr_host* h = new (std::nothrow) mi355_csr_host();
    if (!h) { set_err("out of host memory"); return MI355_LOAD_ERANGE; }
    h->off_type = off_type;
    h->val_type = val_type;
    const int st = off_type == 0 ? (val_type == 0 ? fill(h->a, path) : fill(h->b, path))
                                 : (val_type == 0 ? fill(h->c, path) : fill(h->d, path));
    if (st != MI355_LOAD_OK) { delete h; return st; }
    *out = h;
    return MI355_LOAD_OK;
}

#define MI355_PICK(h, expr) ((h)->off_type == 0 ? ((h)->val_type == 0 ? (h)->a.expr : (h)->b.expr) \
                                                : ((h)->val_type == 0 ? (h)->c.expr : (h)->d.expr))

int mi355_csr_host_dims(const mi355_csr_host* h, int64_t* n_rows, int64_t* n_cols, int64_t* nnz) {
    if (!h || !n_rows || !n_cols || !nnz) { set_err("mi355_csr_host_dims: null pointer"); return MI355_LOAD_EINVAL; }
    *n_rows = int64_t(MI355_PICK(h, number_of_rows));
    *n_cols = int64_t(MI355_PICK(h, number_of_columns));
    *nnz = int64_t(MI355_PICK(h, number_of_nonzeros));
    return MI355_LOAD_OK;
}
const void* mi355_csr_host_Ap(const mi355_csr_host* h) {
    if (!h) return nullptr;
    return h->off_type == 0 ? (h->val_type == 0 ? static_cast<const void*>(h->a.row_offsets.data()) : h->b.row_offsets.data())
                            : (h->val_type == 0 ? static_cast<const void*>(h->c.row_offsets.data()) : h->d.row_offsets.data());
}
const int32_t* mi355_csr_host_Aj(const mi355_csr_host* h) {
    if (!h) return nullptr;
    return reinterpret_cast<const int32_t*>(MI355_PICK(h, column_indices.data()));
}
const void* mi355_csr_host_Ax(const mi355_csr_host* h) {
    if (!h) return nullptr;
    return h->off_type == 0 ? (h->val_type == 0 ? static_cast<const void*>(h->a.nonzero_values.data()) : h->b.nonzero_values.data())
                            : (h->val_type == 0 ? static_cast<const void*>(h->c.nonzero_values.data()) : h->d.nonzero_values.data());
}
void mi355_csr_host_free(mi355_csr_host* h) { delete h; }
const char* mi355_load_last_error(void) { return g_err; }

}  // extern "C"
