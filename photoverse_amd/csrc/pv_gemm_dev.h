// Internal (not part of the C-ABI): what the translation units behind pv_gemm_conv share.
#pragma once
#include "pv_common.h"

// kernel-side parameter block: the C-ABI struct + byte extents of the three buffer descriptors
struct pv_gemm_params_dev : pv_gemm_params {
    uint32_t a0_bytes, a1_bytes, w_bytes;
};

// pv_convbig.hip: 256 x 320 tile, one 8-wave workgroup per CU, for the stride-1 / pad-1 3x3 convs (optionally x2-upsampling) whose launch has
// >= 256 such tiles (no split-K, fp16 output).  Returns -1 when the shape is not one it takes (the caller falls back to the 128-row kernel), -2 when in addition the launch asks for the GroupNorm fold
// (a_norm: no fallback exists), else a hipError_t.
__attribute__((visibility("hidden"))) int pv_conv_big_launch(const pv_gemm_params_dev& p, hipStream_t stream);            // library-internal


// pv_gemm.hip: the fixed-order reduction of split-K slabs + GEMM epilogue (+ column statistics) as its own launch
__attribute__((visibility("hidden"))) int pv_gemm_splitk_reduce_launch(const pv_gemm_params_dev& p, int splits, hipStream_t stream);


// pv_gemm_conv_kernel_info: while a probe is installed (this thread), launch<> / launch_big<> describe the launch they WOULD make - the kernel symbol as
// rocprofv3 prints it and the workgroup count - and return without touching the device.  The host side tags its recorded launches with it instead of
// restating the dispatch rules.
struct pv_launch_probe {
    char name[160];
    long wgs;
};
extern thread_local pv_launch_probe* pv_gemm_probe;
