// csr_vector_f64.hip — the fp64 instantiations of the VECTOR kind (see the end of csr_vector.hip).
#define MI355_TU_F64 1
#include "csr_vector.hip"
