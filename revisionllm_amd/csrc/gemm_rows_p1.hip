// gemm_rows.hip, part 1 (FP8 weights, 4 / 5 row blocks): the file is split over four translation units so that they compile side by side
#define RS_PART 1
#include "gemm_rows.hip"
