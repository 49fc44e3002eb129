// gemm_rows.hip, part 3 (FP8 weights, 8 / 9 row blocks): the file is split over four translation units so that they compile side by side
#define RS_PART 3
#include "gemm_rows.hip"
