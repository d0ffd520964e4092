// pv_gemm_conv: MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (CDNA4).
//
//   out[M][N] = epilogue( A[M][K] * W[N][K]^T )           fp16 in, fp32 accumulate
//
// Tile: 128 (M) x BN (N) x 64 (K), BN = 160 (NF=5) or 128 (NF=4); 256 threads = 4 waves in a 2x2
// grid, each wave owns 64 x (BN/2) outputs as 4 x NF fragments of v_mfma_f32_16x16x32_f16.
// Operands are passed swapped (W as MFMA-A, activations as MFMA-B), so a lane's 4 accumulator
// registers are 4 CONSECUTIVE output columns of one row -> 8-byte fp16 stores, 32-byte runs.
//
// Both operands reach LDS by LDS-DMA (global_load_lds_dwordx4): one wave-instruction moves
// 8 rows x 128 B.  The LDS image is lane-linear, so the bank-conflict swizzle (16-B chunk ^= row&7)
// is applied to the per-lane SOURCE address and again on the ds_read_b128 side.  For the 3x3 conv
// the per-lane source address IS the im2col gather: each 64-channel K-chunk lies inside one filter
// tap, out-of-image taps (zero padding) and the M tail read a zero page.  Two LDS stages, one
// barrier per K-step, 2 workgroups per CU.
#include "pv_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;  // 128 B per LDS row

template <int NF>
struct TileCfg {
    static constexpr int BN = NF * 32;
    static constexpr int A_BYTES = BM * ROW_BYTES;
    static constexpr int B_BYTES = BN * ROW_BYTES;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int SMEM_BYTES = 2 * STAGE_BYTES;
};

// activations allowed in the GEMM epilogue (erf-GELU only exists in the GEGLU instantiation)
__device__ __forceinline__ float epi_act(float x, int act) {
    if (act == PV_ACT_SILU) return pv_silu(x);
    if (act == PV_ACT_QUICK_GELU) return pv_quick_gelu(x);
    if (act == PV_ACT_LEAKY_RELU) return x > 0.f ? x : 0.01f * x;
    return x;
}

__device__ __forceinline__ half8_t lds_frag(const char* base, int row, int chunk) {
    return *reinterpret_cast<const half8_t*>(base + row * ROW_BYTES + ((chunk ^ (row & 7)) << 4));
}

template <int NF, bool CONV, bool GEGLU>
__global__ __launch_bounds__(256, 2) void gemm_conv_kernel(const pv_gemm_params p, const int tiles_n, const int nblk,
                                                            const int m_fast) {
    using Cfg = TileCfg<NF>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = pv_lane_id();
    const int wave = pv_wave_id();
    const int bid = pv_xcd_remap((int)blockIdx.x, nblk);
    const int tiles_m = nblk / tiles_n;
    const int tile_m = m_fast ? bid % tiles_m : bid / tiles_n;
    const int tile_n = m_fast ? bid / tiles_m : bid % tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * Cfg::BN;

    const int cin = p.c0 + p.c1;
    const int K = p.taps * cin;
    const int nk = K / BK;
    const int cpt = cin / BK;  // K-chunks per tap

    // ---- per-thread staging geometry ------------------------------------------------------
    const int lrow = lane >> 3;                 // row inside the 8-row piece
    const int src_chunk = (lane & 7) ^ lrow;    // swizzled source chunk (row & 7 == lrow)
    // A: 4 pieces per wave; row r = (wave*4+i)*8 + lrow
    int a_b[4], a_y[4], a_x[4];
    bool a_ok[4];
    const int hw_out = p.hout * p.wout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (wave * 4 + i) * 8 + lrow;
        a_ok[i] = m < p.M;
        if (CONV) {
            const int b = m / hw_out;
            const int rem = m - b * hw_out;
            a_b[i] = b;
            a_y[i] = rem / p.wout;
            a_x[i] = rem - a_y[i] * p.wout;
        } else {
            a_b[i] = m;
            a_y[i] = 0;
            a_x[i] = 0;
        }
    }
    const half_t* wrow[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int n = n0 + (wave * NF + i) * 8 + lrow;
        wrow[i] = reinterpret_cast<const half_t*>(p.w) + (size_t)n * K + src_chunk * 8;
    }
    const half_t* zero = reinterpret_cast<const half_t*>(p.zero_page) + src_chunk * 8;
    const int hl = p.upsample ? p.hin * 2 : p.hin;
    const int wl = p.upsample ? p.win * 2 : p.win;

    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * Cfg::STAGE_BYTES;
        char* sb = sa + Cfg::A_BYTES;
        const int tap = CONV ? kt / cpt : 0;
        const int c = (kt - tap * cpt) * BK;          // channel offset inside the (concatenated) input
        const bool first = c < p.c0;
        const half_t* src = reinterpret_cast<const half_t*>(first ? p.a0 : p.a1);
        const int ld = first ? p.lda0 : p.lda1;
        const int cc = (first ? c : c - p.c0) + src_chunk * 8;
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const half_t* g;
            if (CONV) {
                const int iy = a_y[i] * p.stride + ky - 1;
                const int ix = a_x[i] * p.stride + kx - 1;
                const bool ok = a_ok[i] && iy >= 0 && iy < hl && ix >= 0 && ix < wl;
                const int py = p.upsample ? iy >> 1 : iy;
                const int px = p.upsample ? ix >> 1 : ix;
                const size_t pix = (size_t)(a_b[i] * p.hin + py) * p.win + px;
                g = ok ? src + pix * ld + cc : zero;
            } else {
                g = a_ok[i] ? src + (size_t)a_b[i] * ld + cc : zero;
            }
            pv_glds16(g, sa + (wave * 4 + i) * 8 * ROW_BYTES);
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) pv_glds16(wrow[i] + (size_t)kt * BK, sb + (wave * NF + i) * 8 * ROW_BYTES);
    };

    // ---- accumulators: acc[ni][mi], D[i = n][j = m] ----------------------------------------
    float4_t acc[NF][4];
#pragma unroll
    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = float4_t{0.f, 0.f, 0.f, 0.f};

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
        const char* sa = smem + cur * Cfg::STAGE_BYTES;
        const char* sb = sa + Cfg::A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t xa[4], wb[NF];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) xa[mi] = lds_frag(sa, wm * 64 + mi * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) wb[ni] = lds_frag(sb, wn * (NF * 16) + ni * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int ni = 0; ni < NF; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + fr;
        if (m >= p.M) continue;
        const float* radd = nullptr;
        if (p.rowadd) radd = p.rowadd + (size_t)(m / hw_out) * p.rowadd_ld;
        if (GEGLU) {
#pragma unroll
            for (int q = 0; q < NF / 2; ++q) {
                const int npk = n0 + wn * (NF * 16) + (2 * q) * 16 + fq * 4;   // packed column of the value fragment
                const int j = (n0 >> 1) + wn * (NF * 8) + q * 16 + fq * 4;      // logical output column
                float4_t v = acc[2 * q][mi], g = acc[2 * q + 1][mi];
                if (p.bias) {
                    const float4_t bv = *reinterpret_cast<const float4_t*>(p.bias + npk);
                    const float4_t bg = *reinterpret_cast<const float4_t*>(p.bias + npk + 16);
                    v += bv;
                    g += bg;
                }
                half4_t o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)(v[r] * pv_gelu_erf(g[r]));
                *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + j) = o;
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) {
                const int n = n0 + wn * (NF * 16) + ni * 16 + fq * 4;
                float4_t v = acc[ni][mi];
                if (p.bias) v += *reinterpret_cast<const float4_t*>(p.bias + n);
                if (radd) v += *reinterpret_cast<const float4_t*>(radd + n);
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = epi_act(v[r], p.act);
                }
                if (p.residual) {
                    const half4_t rr = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(p.residual) + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                }
                if (p.out_f32) {
                    *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = v;
                } else {
                    half4_t o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
                    *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + n) = o;
                }
            }
        }
    }
}

template <int NF, bool CONV, bool GEGLU>
int launch(const pv_gemm_params& p, hipStream_t stream) {
    using Cfg = TileCfg<NF>;
    static bool attr_set = false;
    auto kern = gemm_conv_kernel<NF, CONV, GEGLU>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = p.N / Cfg::BN;
    const int nblk = tiles_m * tiles_n;
    // XCD footprint heuristic: walk M fastest when the weight panel is too big to sit in every XCD's L2
    const size_t wbytes = (size_t)p.N * p.taps * (p.c0 + p.c1) * 2;
    const int m_fast = (wbytes > (3u << 20)) && tiles_n >= 8 ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), Cfg::SMEM_BYTES, stream, p, tiles_n, nblk, m_fast);
    return PV_CHECK_LAUNCH();
}

}  // namespace

extern "C" int pv_gemm_conv(const pv_gemm_params* pp, void* stream_) {
    const pv_gemm_params& p = *pp;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int cin = p.c0 + p.c1;
    if (p.M <= 0 || p.N <= 0 || cin <= 0 || (cin % 64) || (p.c0 % 64) || (p.taps != 1 && p.taps != 9) || !p.zero_page || p.act == PV_ACT_GELU ||
        !p.a0 || !p.w || !p.out || (p.c1 && !p.a1) || p.hout * p.wout <= 0)
        return (int)hipErrorInvalidValue;
    if (p.geglu) {
        if (p.taps != 1 || (p.N % 128)) return (int)hipErrorInvalidValue;
        return launch<4, false, true>(p, stream);
    }
    const bool conv = p.taps == 9;
    if (p.N % 160 == 0) return conv ? launch<5, true, false>(p, stream) : launch<5, false, false>(p, stream);
    if (p.N % 128 == 0) return conv ? launch<4, true, false>(p, stream) : launch<4, false, false>(p, stream);
    return (int)hipErrorInvalidValue;
}
