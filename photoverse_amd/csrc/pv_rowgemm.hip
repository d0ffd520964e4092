// pv_row_gemm: Linear layers with a SHORT contraction (K = 320: every Linear of the 64x64-level transformer blocks) as a ROW-OWNING kernel
//
//   out[M][N'] = epilogue( LayerNorm(x)[M][320] . W[N][320]^T + bias )        N' = N (plain) or N / 2 (GEGLU: value * gelu(gate))
//
// i.e. BasicTransformerBlock.norm1 -> fused [to_q; to_k; to_v] (attn1, stock AttnProcessor2_0 installed by /root/reference/models/unet.py:20-24)
// and norm3 -> ff.net[0] (GEGLU proj) of diffusers' transformer block [EXT], the two widest K = 320 GEMMs of a UNet forward.
//
// Why not the tiled GEMM (pv_gemm.hip) for these: at K = 320 a 128 x BN tile re-streams its 128 x 320 activation slab through L2 -> LDS once per
// N-tile (16-20 times for the GEGLU projection: 0.84 GB per launch, as much again for the weights) and LayerNorm is a separate pass over the
// tensor.  Here a wave OWNS 32 rows for the whole launch: the rows are loaded once, straight into the MFMA-B register layout (80 VGPRs),
// normalised in registers (LayerNorm's affine part is folded into W / bias by the caller), and only the weights stream - through a 4-slot LDS ring
// filled by LDS-DMA (counted vmcnt, one raw s_barrier per PAIR of 20-KiB stages).  Same building blocks as phases 0 / 1 / 3 of the fused attn2
// kernel (pv_xfused.hip).  L2 -> LDS traffic halves, the LayerNorm launch (84 MB of HBM traffic) disappears, MFMAs per barrier: 80 per wave.
//
// Geometry: workgroup = 128 rows, 4 waves x 32 rows (two 16-query MFMA columns); weight chunk = 160 rows of W (10 fragments) x K, streamed as
// 5 stages of 160 x 64; accumulators 10 x 2 float4 (80 VGPRs).  The chunk loop is a runtime loop over PAIRS of chunks (10 stages: an even number,
// so the stage pairs of the barrier scheme never straddle an iteration); ring slots are addressed through a runtime base.  M tails need no
// code: rows are read and written through buffer descriptors sized to the workgroup's valid rows (out-of-range loads return 0, stores are dropped).
#include "pv_common.h"
#include <stdlib.h>

namespace {

constexpr int RG_K = 320;
constexpr int RG_CHUNK = 160;                    // weight rows per chunk
constexpr int RG_TILE = RG_CHUNK * 128;          // one stage: 160 rows x 64 k = 20 KiB
constexpr int RG_SLOTS = 4;
constexpr int RG_SMEM = RG_SLOTS * RG_TILE;      // 80 KiB -> two workgroups per CU

struct pv_rowgemm_params_dev : pv_row_gemm_params {
    uint32_t w_bytes;
};

template <int N>
__device__ __forceinline__ void rg_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool GEGLU>
__global__ __launch_bounds__(256, 2) void row_gemm_kernel(const pv_rowgemm_params_dev p) {
    constexpr int C = RG_K;
    constexpr int KK = C / 32;        // 32-deep contraction steps: 10
    constexpr int KT = C / 64;        // ring stages per chunk: 5
    constexpr int NFC = RG_CHUNK / 16;   // fragments per chunk: 10
    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = pv_lane_id(), wave = pv_wave_id();
    const int fr = lane & 15, g = lane >> 4;
    const int m0 = (int)blockIdx.x * 128;
    const int rows_here = min(128, p.M - m0);
    const int n_chunks = p.N / RG_CHUNK;          // even (checked by the launcher)
    const int NT = n_chunks * KT;                 // ring stages of the launch

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(reinterpret_cast<const half_t*>(p.x)) + (size_t)m0 * p.ld_x, 0,
                                                                        rows_here * p.ld_x * 2, 0x00020000);
    const int ld_o = p.ld_out;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<half_t*>(p.out) + (size_t)m0 * ld_o, 0, rows_here * ld_o * 2, 0x00020000);

    // ---- LDS-DMA of one ring stage: 160 weight rows x 64 k = 20 pieces of 8 rows x 128 B, five per wave; swizzle on the SOURCE side:
    // LDS position (lane & 7) of row r holds chunk (lane & 7) ^ (r & 7)
    const int lrow = lane >> 3;
    const unsigned w_lane_off = (unsigned)lrow * (unsigned)(C * 2) + (unsigned)(((lane & 7) ^ lrow) << 4);
    const unsigned piece_off = w_lane_off + (unsigned)(wave * 8) * (unsigned)(C * 2);   // piece i adds 32 i rows: scalar part of the address
    auto issue_stage = [&](int t) {                        // t: runtime stage index (chunk t / 5, k-stage t % 5)
        const int nc = t / KT, kt = t - nc * KT;
        char* dst = smem + (t & (RG_SLOTS - 1)) * RG_TILE;
        const int soff = nc * RG_CHUNK * (C * 2) + kt * 128;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, PV_LDS_PTR(dst + (wave + 4 * i) * 1024), 16, (int)piece_off, soff + i * 32 * (C * 2), 0, 0);
    };
    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- rows into registers (the MFMA-B layout: lane = row, 8 consecutive channels per k-group) ---------------------------------------
    half8_t xf[KK][2];
    int mrow[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        mrow[qi] = wave * 32 + qi * 16 + fr;
        const int off = mrow[qi] * p.ld_x * 2 + g * 16;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) xf[kk][qi] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rx, off, kk * 64, 0));
    }
    issue_stage(0);
    issue_stage(1);
    if (p.x_norm) {
        // GroupNorm of the rows in its affine form (pv_groupnorm_scale_shift's table): x * scale[image][c] + shift[image][c], rounded to fp16 as
        // pv_groupnorm_apply rounds what it writes.  The workgroup's 128 rows lie in one image; the 16 lanes of a k-group read the same 8 channels
        // (one request per wave instruction)
        const float* tab = p.x_norm + (size_t)(m0 / p.rows_per_image) * 2 * C + g * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const float4_t s0 = *reinterpret_cast<const float4_t*>(tab + kk * 32), s1 = *reinterpret_cast<const float4_t*>(tab + kk * 32 + 4);
            const float4_t h0 = *reinterpret_cast<const float4_t*>(tab + C + kk * 32), h1 = *reinterpret_cast<const float4_t*>(tab + C + kk * 32 + 4);
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[kk][qi][j] = (half_t)((float)xf[kk][qi][j] * (j < 4 ? s0[j & 3] : s1[j & 3]) + (j < 4 ? h0[j & 3] : h1[j & 3]));
        }
    }
    if (p.ln) {
        // LayerNorm without its affine part (gamma is folded into the columns of W, beta into the bias): statistics by v_dot2_f32_f16, the
        // centred sum of squares from packed fp16 differences with an exact correction for the rounded mean (as in pv_xfused.hip)
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        const h2_t ones = h2_t{(half_t)1.0f, (half_t)1.0f};
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            float sum = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum = __builtin_amdgcn_fdot2(h2_t{xf[kk][qi][2 * j], xf[kk][qi][2 * j + 1]}, ones, sum, false);
            const float mean = pv_quad_sum(sum) * (1.0f / (float)C);
            const half_t mh = (half_t)mean;
            const h2_t nm = h2_t{(half_t)(-mh), (half_t)(-mh)};
            float sq = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const h2_t d = h2_t{xf[kk][qi][2 * j], xf[kk][qi][2 * j + 1]} + nm;
                    sq = __builtin_amdgcn_fdot2(d, d, sq, false);
                }
            const float dm = mean - (float)mh;
            const float var = pv_quad_sum(sq) * (1.0f / (float)C) - dm * dm;
            const float rstd = rsqrtf(fmaxf(var, 0.f) + p.ln_eps);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[kk][qi][j] = (half_t)fmaf((float)xf[kk][qi][j], rstd, nmr);
        }
    }
    rg_wait_vmcnt<0>();               // rows, and the first two weight stages, have landed: the hand-counted bookkeeping starts from zero

    // fragment reads of a stage: row 16 i + fr, chunk (4 ks + g) ^ (fr & 7): two per-lane offsets + compile-time row offsets + the slot base
    const int ring_lane[2] = {fr * 128 + ((g ^ (fr & 7)) << 4), fr * 128 + (((4 + g) ^ (fr & 7)) << 4)};
    const int pcol = (g & 1) ? 16 + (g - 1) * 4 : g * 4;      // this lane's 8 columns inside a fragment pair after the lane-row swap
    constexpr int OUT_PER_CHUNK = GEGLU ? RG_CHUNK / 2 : RG_CHUNK;   // output columns a chunk produces
    constexpr int STORES = GEGLU ? 6 : 10;                            // VMEM stores per wave at a chunk's end (vmcnt bookkeeping)

    float4_t acc[NFC][2];
    for (int it = 0; it < n_chunks / 2; ++it) {
        const int t0 = it * (2 * KT);
#pragma unroll
        for (int j = 0; j < 2 * KT; ++j) {
            const int t = t0 + j;
            const int kt = j % KT;                    // compile time
            const int nc = 2 * it + j / KT;
            if ((j & 1) == 0) {
                // stages t, t+1 were issued two stages ago; only this wave's chunk-end stores issued since may stay in flight (vmcnt retires in order)
                const bool stores_since = (j == 6) || (j == 0 && it > 0);
                if (t >= 2) {
                    if (stores_since) rg_wait_vmcnt<STORES>(); else rg_wait_vmcnt<0>();
                }
                wg_barrier();
                if (t + 2 < NT) issue_stage(t + 2);
                if (t + 3 < NT) issue_stage(t + 3);
            }
            if (kt == 0) {
                // accumulators start from the bias (rows 4g .. 4g+3 of each fragment): no bias registers beside them
#pragma unroll
                for (int i = 0; i < NFC; ++i)
                    acc[i][0] = acc[i][1] = p.bias ? *reinterpret_cast<const float4_t*>(p.bias + nc * RG_CHUNK + i * 16 + g * 4) : float4_t{0.f, 0.f, 0.f, 0.f};
            }
            const char* slot = smem + (t & (RG_SLOTS - 1)) * RG_TILE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* base = slot + ring_lane[ks];
#pragma unroll
                for (int i = 0; i < NFC; ++i) {
                    const half8_t a = *reinterpret_cast<const half8_t*>(base + i * 2048);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi) acc[i][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[2 * kt + ks][qi], acc[i][qi], 0, 0, 0);
                }
            }
            if (kt == KT - 1) {
                // ---- chunk epilogue: bias, (GEGLU gate,) fp16, lane-row swap -> 16-byte stores --------------------------------------
                const int ocol0 = nc * OUT_PER_CHUNK;
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    const int ooff = mrow[qi] * ld_o * 2 + ocol0 * 2;
                    unsigned pk[GEGLU ? NFC / 2 : NFC][2];
                    if (GEGLU) {
#pragma unroll
                        for (int q = 0; q < NFC / 2; ++q) {
                            const float4_t v = acc[2 * q][qi], gt = acc[2 * q + 1][qi];
                            pk[q][0] = __builtin_bit_cast(unsigned, half2_t{(half_t)(v[0] * pv_gelu_erf(gt[0])), (half_t)(v[1] * pv_gelu_erf(gt[1]))});
                            pk[q][1] = __builtin_bit_cast(unsigned, half2_t{(half_t)(v[2] * pv_gelu_erf(gt[2])), (half_t)(v[3] * pv_gelu_erf(gt[3]))});
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NFC; ++i) {
                            const float4_t v = acc[i][qi];
                            pk[i][0] = __builtin_bit_cast(unsigned, half2_t{(half_t)v[0], (half_t)v[1]});
                            pk[i][1] = __builtin_bit_cast(unsigned, half2_t{(half_t)v[2], (half_t)v[3]});
                        }
                    }
                    constexpr int NOUT = GEGLU ? NFC / 2 : NFC;     // output fragments of the chunk: 5 or 10
#pragma unroll
                    for (int q = 0; q < NOUT / 2; ++q) {
                        const auto v0 = __builtin_amdgcn_permlane16_swap(pk[2 * q][0], pk[2 * q + 1][0], false, false);
                        const auto v1 = __builtin_amdgcn_permlane16_swap(pk[2 * q][1], pk[2 * q + 1][1], false, false);
                        const unsigned sa0 = v0[0], sb0 = v0[1], sa1 = v1[0], sb1 = v1[1];
                        // constant part of the address in the voffset expression (folded into the immediate), never in an SGPR soffset: see pv_xfused.hip
                        __builtin_amdgcn_raw_buffer_store_b128(uint4_t{sa0, sa1, sb0, sb1}, ro, ooff + pcol * 2 + q * 64, 0, 0);
                        asm volatile("s_nop 1" ::: "memory");
                    }
                    if (NOUT & 1)
                        __builtin_amdgcn_raw_buffer_store_b64(uint2_t{pk[NOUT - 1][0], pk[NOUT - 1][1]}, ro, ooff + g * 8 + (NOUT - 1) * 32, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

}  // namespace

extern "C" int pv_row_gemm(const pv_row_gemm_params* pp, void* stream) {
    pv_rowgemm_params_dev p;
    static_cast<pv_row_gemm_params&>(p) = *pp;
    if (!p.x || !p.w || !p.out || p.M <= 0 || p.K != RG_K || p.N <= 0 || (p.N % (2 * RG_CHUNK)) || (p.ld_x % 8) || (p.ld_out % 8) || p.ld_x < RG_K ||
        p.ld_out < (p.geglu ? p.N / 2 : p.N) || (p.x_norm && (p.ln || p.rows_per_image <= 0 || (p.rows_per_image % 128) || (p.M % p.rows_per_image))))
        return (int)hipErrorInvalidValue;
    if ((size_t)p.N * RG_K * 2 >= (1ull << 31)) return (int)hipErrorInvalidValue;
    p.w_bytes = (uint32_t)p.N * RG_K * 2;
    static bool attr_set_dev[64][2] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63][p.geglu ? 1 : 0];
    auto kern = p.geglu ? row_gemm_kernel<true> : row_gemm_kernel<false>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RG_SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((p.M + 127) / 128)), dim3(256), RG_SMEM, (hipStream_t)stream, p);
    return PV_CHECK_LAUNCH();
}
