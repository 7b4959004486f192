/* mi355_load.h — C ABI of the Matrix Market ingest (libmi355load.so; host code only, no GPU).
 *
 * The reference's only input path is argv[1] -> LoadCoo -> ToCsr (main.cu:32-39, include/load.hpp:268-474): a
 * Matrix Market coordinate file becomes host CSR arrays that the harness uploads (main.cu:48-74).  The loader behind
 * these entry points is spmv-samples_amd/host/load.hpp — the same results entry for entry (1-based -> 0-based,
 * pattern -> 1.0, `symmetric` expanded entry-then-mirror in file order with the diagonal once, CSR by a stable
 * counting sort on the row, so file order survives inside a row and duplicates are kept), parsed in parallel from an
 * mmap.  The C++ harness (host/main.cpp) includes load.hpp directly; this ABI is for callers that are not C++ —
 * bench.py --mtx binds it with ctypes.
 *
 * Errors: the reference exits or throws (load.hpp:278-306, :324-360); here every failure is a status and a message.
 */
#ifndef MI355_LOAD_H
#define MI355_LOAD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MI355_LOAD_OK = 0,
    MI355_LOAD_EINVAL = 1,   /* null pointer / unknown type enum                                              */
    MI355_LOAD_EFILE = 2,    /* cannot open, not a Matrix Market coordinate file, unsupported field / symmetry */
    MI355_LOAD_EPARSE = 3,   /* short or malformed entry, zero-based index, index beyond the declared size     */
    MI355_LOAD_ERANGE = 4    /* rows / columns / entries do not fit index_t / offset_t                         */
};

typedef struct mi355_csr_host mi355_csr_host;

/* off_type: 0 = int32 offsets, 1 = int64 (MI355_OFF_*); val_type: 0 = float, 1 = double (MI355_VAL_*); index_t is
 * int32 (main.cu:15).  *out owns the arrays until mi355_csr_host_free.                                           */
int mi355_load_mtx(const char* path, int off_type, int val_type, mi355_csr_host** out);
int mi355_csr_host_dims(const mi355_csr_host* csr, int64_t* n_rows, int64_t* n_cols, int64_t* nnz);
/* the arrays themselves (host memory, valid until free): Ap has n_rows + 1 offsets of the loaded width             */
const void* mi355_csr_host_Ap(const mi355_csr_host* csr);
const int32_t* mi355_csr_host_Aj(const mi355_csr_host* csr);
const void* mi355_csr_host_Ax(const mi355_csr_host* csr);
void mi355_csr_host_free(mi355_csr_host* csr);
/* Message of the last failing call on this thread ("" if none).                                                    */
const char* mi355_load_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355_LOAD_H */
