// merge_path_f64.hip — the fp64 (and fp32-matrix-under-fp64-vectors) instantiations of the MERGE kind (see the end of merge_path.hip).
#define MI355_TU_F64 1
#include "merge_path.hip"
