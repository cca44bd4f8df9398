// Precision "fp16" (host/ops.py; include/dm4d.h "fp16 precision"): the GEMM / convolution kernels of gemm.hip instantiated with
// PAR = 2 -- fp16 MFMA operands, fp32 side inputs and outputs -- as their own translation unit, so that the two compile in parallel and
// the fast kernels' objects are not touched.  Entry points: dm4d_gemm_f16, dm4d_conv3x3_nhwc_f16.
#define DM4D_GEMM_H16_TU 1
#include "gemm.hip"
