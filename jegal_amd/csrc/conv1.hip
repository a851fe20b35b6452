// conv1 fused kernel lands here (v1); v0 uses stack_frames + implicit GEMM + maxpool.
#include "common.h"
