// Internal (not part of the C-ABI): what the translation units behind pv_gemm_conv share.
#pragma once
#include "pv_common.h"

// kernel-side parameter block: the C-ABI struct + byte extents of the three buffer descriptors
struct pv_gemm_params_dev : pv_gemm_params {
    uint32_t a0_bytes, a1_bytes, w_bytes;
};

// pv_convbig.hip: 256 x 320 x 64 tile, one 8-wave workgroup per CU, for the 3x3 convs of the 64 x 64 level (stride 1, pad 1, no upsample,
// no split-K, fp16 output).  Returns -1 when the shape is not one it takes (the caller falls back to the 128-row kernel), else a hipError_t.
int pv_conv_big_launch(const pv_gemm_params_dev& p, hipStream_t stream);
