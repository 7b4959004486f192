// merge_path_i32.hip — the 32-bit-integer-value instantiations of the MERGE kind (see the end of merge_path.hip).
#define MI355_TU_I32 1
#include "merge_path.hip"
