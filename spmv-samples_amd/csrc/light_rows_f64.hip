// light_rows_f64.hip — the fp64 instantiations of the LIGHT kind (see the end of light_rows.hip).
#define MI355_TU_F64 1
#include "light_rows.hip"
