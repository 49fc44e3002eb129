// gemm_rows.hip, part 2 (bf16 weights, 8 / 9 row blocks): the file is split over four translation units so that they compile side by side
#define RS_PART 2
#include "gemm_rows.hip"
