// pv_gemm_conv: MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (CDNA4).
//
//   out[M][N] = epilogue( A[M][K] * W[N][K]^T )           fp16 in, fp32 accumulate
//
// Kernel (gemm_conv_kernel<NF, CONV, GEGLU, CS>): tile 128 (M) x BN (N) x 64 (K), BN = 160 (NF=5) or 128
// (NF=4); 256 threads = 4 waves in a 2x2 grid, each wave owns 64 x (BN/2) outputs as 4 x NF fragments of
// v_mfma_f32_16x16x32_f16; two workgroups per CU.  Operands are passed swapped (W as MFMA-A, activations as MFMA-B), so a
// lane's 4 accumulator registers are 4 CONSECUTIVE output columns of one row; fragment pairs are traded between lane rows
// (v_permlane16_swap) so the epilogue stores 16 bytes per lane.
//
// Both operands reach LDS by LDS-DMA through raw buffer descriptors (buffer_load_dwordx4 ... lds): one wave-instruction
// moves 8 rows x 128 B.  The LDS image is lane-linear, so the bank-conflict swizzle (16-B chunk ^= row&7) is applied to the
// per-lane SOURCE offset and again on the ds_read_b128 side.  For the 3x3 conv the per-lane source offset IS the im2col
// gather: each 64-channel K-chunk lies inside one filter tap; out-of-image taps (zero padding) and the M tail use an
// out-of-range offset that the hardware range check turns into zeros.  K order is channel-chunk major / tap minor (the 9
// shifted re-reads of a slab hit L1/L2).  Two LDS stages; one raw s_barrier per K-step placed BETWEEN the two 32-deep
// halves of the step, so the fragment reads of one half overlap the MFMAs of the other and the next stage's DMA is issued
// behind a counted vmcnt.  Small-M layers split K over several workgroups (fp32 slabs reduced in fixed order).
//
// Larger tiles (256-row / one wave per SIMD, 256 x 160 x 32 with three stages) were built and measured in round 1 and lost
// to this kernel on every UNet shape; that code and its timing ablations live in tools/r01_experiments/ (DESIGN.md section 4).
#include "pv_gemm_dev.h"

namespace {

constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;  // 128 B per LDS row

// 128-row tile, wave tile 64 x BN/2, 2 LDS stages, 2 workgroups / CU (two waves per SIMD)
template <int NF, int MIV = 4>
struct TileCfg {
    static constexpr int MI = MIV;                 // 16-row M fragments per wave (4: 128-row tile; 2: 64-row tile for HBM-bound Linear layers)
    static constexpr int BM = 2 * MI * 16;
    static constexpr int BN = NF * 32;
    static constexpr int NWAVES = 4;               // 2 x 2 waves
    static constexpr int THREADS = NWAVES * 64;
    static constexpr int A_PER_WAVE = BM / 8 / NWAVES;   // 8-row LDS-DMA pieces of the activation tile per wave
    static constexpr int STAGES = 2;
    static constexpr int A_BYTES = BM * ROW_BYTES;
    static constexpr int B_BYTES = BN * ROW_BYTES;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int SMEM_BYTES = STAGES * STAGE_BYTES;
    static constexpr int B_PIECES = NF * 4;                                   // 8-row LDS-DMA pieces of the weight tile
    static constexpr int B_PER_WAVE = (B_PIECES + NWAVES - 1) / NWAVES;       // max pieces any wave issues
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// all-reduce over the 16 lanes of a DPP row (row_ror:8,4,2,1)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// CS: instantiation whose epilogue also produces the GroupNorm column statistics (pv_gemm_params.colstats).  Separate from the
// plain kernel because the extra 40 accumulators cost the launches that do not want them 2-3 %.
template <int NF, bool CONV, bool GEGLU, bool CS = false, bool MULTI = false, int MIV = 4>
__global__ __launch_bounds__(256, MIV == 4 ? 2 : 3) void gemm_conv_kernel(const pv_gemm_params_dev p, const int tiles_n,
                                                                              const int nblk, const int order, const int tpw_arg) {
    const int tpw = MULTI ? tpw_arg : 1;        // MULTI = false: the straight one-tile kernel (the tile loop folds away)
    using Cfg = TileCfg<NF, MIV>;
    static_assert(MIV == 4 || (!CS && !MULTI && !CONV), "the 64-row tile exists for plain Linear layers only");
    constexpr int BM = Cfg::BM;
    constexpr int NW = Cfg::NWAVES;
    constexpr int MI = Cfg::MI;
    constexpr int AP = Cfg::A_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = pv_lane_id();
    const int wave = pv_wave_id();
    // tpw (tiles per workgroup): a workgroup walks `tpw` consecutive N-tiles of one M-tile with ONE continuous K pipeline - the
    // LDS-DMA of the next tile's first stages is issued during the last K-steps of the current one and overlaps its epilogue, so
    // short-K layers (K = 320: five K-steps per tile) do not pay a cold prologue per 128 x BN outputs.  tiles_n / nblk count GROUPS.
    const int bid = (order & 2) ? (int)blockIdx.x : pv_xcd_remap((int)blockIdx.x, nblk);
    const int m_fast = order & 1;
    const int tiles_m = nblk / tiles_n;
    const int tile_m = m_fast ? bid % tiles_m : bid / tiles_n;
    const int tile_n = (m_fast ? bid / tiles_m : bid % tiles_n) * tpw;      // first N-tile of the group
    const int m0 = tile_m * BM;
    int n0 = tile_n * Cfg::BN;                                               // advances by BN per tile

    const int cin = p.c0 + p.c1;
    const int K = p.taps * cin;
    // split-K: blockIdx.y owns K-steps [k_begin, k_begin + nk) and writes an fp32 partial slab (reduced by splitk_reduce_kernel)
    const int nk_total = K / BK;
    const int nk_per = (nk_total + (int)gridDim.y - 1) / (int)gridDim.y;
    const int k_begin = (int)blockIdx.y * nk_per;
    const int nk = max(0, min(nk_total, k_begin + nk_per) - k_begin);

    // ---- per-thread staging geometry ------------------------------------------------------
    // All global->LDS traffic is `buffer_load_dwordx4 ... lds` through raw buffer descriptors: a lane whose source is
    // padding (out-of-image filter tap, M tail) gets an out-of-range offset and the hardware range check writes
    // zeros - no branches, no zero page, and the per-K-step address math is one add + one select per piece.
    const int lrow = lane >> 3;                 // row inside the 8-row piece
    const int src_chunk = (lane & 7) ^ lrow;    // swizzled source chunk (row & 7 == lrow)
    const int lane_cc2 = src_chunk * 16;        // byte offset of this lane's 16-B chunk inside the 64-channel slab
    constexpr unsigned OOB = 0x80000000u;       // >= num_records of every descriptor (launcher checks sizes < 2 GiB)
    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a0), 0, (int)p.a0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a1 ? p.a1 : p.a0), 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const int hw_out = p.hout * p.wout;
    const bool fast_conv = CONV && p.stride == 1 && !p.upsample;   // pad == 1 (checked by the launcher)
    const int hl = p.upsample ? p.hin * 2 : p.hin;
    const int wl = p.upsample ? p.win * 2 : p.win;
    // A: AP pieces per wave; piece j = wave + i*NW covers tile rows [8j, 8j+8)
    unsigned a_off0[AP], a_off1[AP];   // byte offset of the row (centre pixel for convs) in source 0 / 1, + lane chunk
    unsigned a_mask[AP];             // conv: bit t set <=> filter tap t reads inside the image
    int a_b[AP], a_y[AP], a_x[AP];   // generic (strided / upsampled) conv path only
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + (wave + i * NW) * 8 + lrow;
        const bool ok = m < p.M;
        if (CONV) {
            const int b = m / hw_out;
            const int rem = m - b * hw_out;
            const int y = rem / p.wout, x = rem - y * p.wout;
            a_b[i] = b; a_y[i] = y * p.stride + 1 - p.pad; a_x[i] = x * p.stride + 1 - p.pad;   // tap (ky,kx) reads (a_y+ky-1, a_x+kx-1)
            unsigned mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = a_y[i] + t / 3 - 1, ix = a_x[i] + t % 3 - 1;
                if (ok && iy >= 0 && iy < hl && ix >= 0 && ix < wl) mask |= 1u << t;
            }
            a_mask[i] = mask;
            const unsigned pix = (unsigned)((b * p.hin + y) * p.win + x);     // centre pixel (fast path: hin==hout)
            a_off0[i] = pix * (unsigned)(p.lda0 * 2) + lane_cc2;
            a_off1[i] = pix * (unsigned)(p.lda1 * 2) + lane_cc2;
        } else {
            a_mask[i] = ok ? 1u : 0u;
            a_off0[i] = ok ? (unsigned)m * (unsigned)(p.lda0 * 2) + lane_cc2 : OOB;
            a_off1[i] = ok ? (unsigned)m * (unsigned)(p.lda1 * 2) + lane_cc2 : OOB;
            a_b[i] = a_y[i] = a_x[i] = 0;
        }
    }
    // B: piece j = wave + i*NW < B_PIECES covers weight-tile rows [8j, 8j+8)
    unsigned w_off[Cfg::B_PER_WAVE];
#pragma unroll
    for (int i = 0; i < Cfg::B_PER_WAVE; ++i) {
        const int n = n0 + min(wave + i * NW, Cfg::B_PIECES - 1) * 8 + lrow;
        w_off[i] = (unsigned)n * (unsigned)(K * 2) + lane_cc2;
    }
    // number of LDS-DMA instructions this wave issues per stage (for the counted vmcnt)
    const bool b_full = (Cfg::B_PIECES % NW == 0) || (wave + (Cfg::B_PER_WAVE - 1) * NW < Cfg::B_PIECES);

    auto stage = [&](int g_local, int buf) {      // g_local: flat step index over (tile of the group, K-step)
        const int tn = g_local / nk;
        const int kt = k_begin + (g_local - tn * nk);
        char* sa = smem + buf * Cfg::STAGE_BYTES;
        char* sb = sa + Cfg::A_BYTES;
        // K order: channel-chunk major, filter-tap minor.  The 9 taps of one 64-channel slab re-read (shifted) the same
        // input rows in consecutive K-steps, so the im2col re-reads hit L1/L2 instead of going back to HBM/MALL.
        const int chunk = CONV ? kt / 9 : kt;
        const int tap = CONV ? kt - chunk * 9 : 0;
        const int c = chunk * BK;                     // channel offset inside the (concatenated) input
        const bool first = c < p.c0;
        const __amdgpu_buffer_rsrc_t ra = first ? ra0 : ra1;
        const int ld2 = (first ? p.lda0 : p.lda1) * 2;
        const int sc2 = (first ? c : c - p.c0) * 2;   // scalar byte offset of the slab inside the source row
        const int ky = tap / 3, kx = tap - ky * 3;
        const int tap_delta = ((ky - 1) * p.win + (kx - 1)) * ld2 + sc2;   // fast path: centre pixel -> tap pixel
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            unsigned off;
            if (!CONV) {
                off = (first ? a_off0[i] : a_off1[i]) + (unsigned)sc2;   // OOB rows stay out of range (sc2 < 2^16)
            } else if (fast_conv) {
                off = ((a_mask[i] >> tap) & 1u) ? (first ? a_off0[i] : a_off1[i]) + (unsigned)tap_delta : OOB;
            } else {
                const int iy = a_y[i] + ky - 1, ix = a_x[i] + kx - 1;
                const int py = p.upsample ? iy >> 1 : iy, px = p.upsample ? ix >> 1 : ix;
                const unsigned pix = (unsigned)((a_b[i] * p.hin + py) * p.win + px);
                off = ((a_mask[i] >> tap) & 1u) ? pix * (unsigned)ld2 + (unsigned)(sc2 + lane_cc2) : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, PV_LDS_PTR(sa + (wave + i * NW) * 8 * ROW_BYTES), 16, (int)off, 0, 0, 0);
        }
        const unsigned wk2 = (unsigned)(tap * cin + c) * 2u + (unsigned)tn * (unsigned)(Cfg::BN * K * 2);
#pragma unroll
        for (int i = 0; i < Cfg::B_PER_WAVE; ++i) {
            if (i < Cfg::B_PER_WAVE - 1 || b_full)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, PV_LDS_PTR(sb + (wave + i * NW) * 8 * ROW_BYTES), 16, (int)(w_off[i] + wk2), 0, 0, 0);
        }
    };

    // ---- accumulators: acc[ni][mi], D[i = n][j = m] ----------------------------------------
    float4_t acc[NF][MI];
#pragma unroll
    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = float4_t{0.f, 0.f, 0.f, 0.f};

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- main loop ----------------------------------------------------------------------------------------------
    // Each 64-deep K-step is two 32-deep halves (fragment sets F0, F1).  The workgroup barrier sits BETWEEN the halves:
    //
    //   iteration kt:   read F1(kt) | MFMA F0(kt)          <- LDS reads of one half overlap the MFMAs of the other
    //                   wait stage kt+1 landed, lgkmcnt(0), s_barrier
    //                   issue LDS-DMA of stage kt+S into buffer kt % S     (its reads were retired before the barrier)
    //                   read F0(kt+1) | MFMA F1(kt)
    //
    // S = LDS stages (2 or 3): S-1 stages of global->LDS traffic are always in flight behind a COUNTED vmcnt (raw
    // s_barrier: __syncthreads() would drain the DMA queue).  hipcc left alone serialises "ds_read, wait, 4 MFMA".
    constexpr int S = Cfg::STAGES;
    auto read_half = [&](half8_t (&xa)[MI], half8_t (&wb)[NF], int kt, int ks) {
        const char* sa = smem + (kt % S) * Cfg::STAGE_BYTES;
        const char* sb = sa + Cfg::A_BYTES;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xa[mi] = lds_frag(sa, wm * (MI * 16) + mi * 16 + fr, ks * 4 + fq);
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) wb[ni] = lds_frag(sb, wn * (NF * 16) + ni * 16 + fr, ks * 4 + fq);
    };
    auto mma_half = [&](const half8_t (&xa)[MI], const half8_t (&wb)[NF]) {
#pragma unroll
        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
    };

    const int T = nk * tpw;                     // steps of the whole group
#pragma unroll
    for (int s = 0; s < S; ++s)
        if (s < T) stage(s, s);
    // stage 0 landed <=> at most the pieces of the younger issued stages are outstanding; drain fully when short
    if (T >= S && S == 3) {
        if (b_full) wait_vmcnt<(AP + Cfg::B_PER_WAVE) * 2>(); else wait_vmcnt<(AP + Cfg::B_PER_WAVE - 1) * 2>();
    } else if (T >= S && S == 2) {
        if (b_full) wait_vmcnt<(AP + Cfg::B_PER_WAVE)>(); else wait_vmcnt<(AP + Cfg::B_PER_WAVE - 1)>();
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    half8_t xa0[MI], wb0[NF], xa1[MI], wb1[NF];
    if (T > 0) read_half(xa0, wb0, 0, 0);
    for (int tn = 0; tn < tpw; ++tn) {
    for (int kt = 0; kt < nk; ++kt) {
        const int g = tn * nk + kt;             // flat step: LDS buffer g % S
        read_half(xa1, wb1, g, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_half(xa0, wb0);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < T) {
            // stage g+1 landed: all younger issued stages (g+2 .. g+S-1) may stay in flight, if they were all issued
            if (S == 3 && g + S - 1 < T) {
                if (b_full) wait_vmcnt<(AP + Cfg::B_PER_WAVE)>(); else wait_vmcnt<(AP + Cfg::B_PER_WAVE - 1)>();
            } else {
                wait_vmcnt<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of buffer g % S are retired
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (g + S < T) stage(g + S, g % S);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) {                                   // at a tile boundary the first half of the next tile is read after the epilogue
                read_half(xa0, wb0, g + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mma_half(xa1, wb1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue ---------------------------------------------------------------------------
    if (gridDim.y > 1) {   // split-K partial: raw fp32 accumulators into this split's slab
        float* slab = p.splitk_ws + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wm * (MI * 16) + mi * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) {
                const int n = n0 + wn * (NF * 16) + ni * 16 + fq * 4;
                *reinterpret_cast<float4_t*>(slab + (size_t)m * p.N + n) = acc[ni][mi];
            }
        }
        return;
    }
    // All epilogue loads (bias, time-embedding row, residual) are issued up front into registers and only then consumed:
    // interleaving "load, wait, store" per fragment serialises ~20 memory round trips per thread (measured: 35 us of a
    // 45 us K=320 GEMM).
    const int nbase = n0 + wn * (NF * 16) + fq * 4;
    if (GEGLU) {
        float4_t bv[NF / 2], bg[NF / 2];
#pragma unroll
        for (int q = 0; q < NF / 2; ++q) {
            bv[q] = p.bias ? *reinterpret_cast<const float4_t*>(p.bias + nbase + (2 * q) * 16) : float4_t{0.f, 0.f, 0.f, 0.f};
            bg[q] = p.bias ? *reinterpret_cast<const float4_t*>(p.bias + nbase + (2 * q + 1) * 16) : float4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wm * (MI * 16) + mi * 16 + fr;
            if (m >= p.M) continue;
            unsigned pk[NF / 2][2];
#pragma unroll
            for (int q = 0; q < NF / 2; ++q) {
                const float4_t v = acc[2 * q][mi] + bv[q], g = acc[2 * q + 1][mi] + bg[q];
                pk[q][0] = __builtin_bit_cast(unsigned, half2_t{(half_t)(v[0] * pv_gelu_erf(g[0])), (half_t)(v[1] * pv_gelu_erf(g[1]))});
                pk[q][1] = __builtin_bit_cast(unsigned, half2_t{(half_t)(v[2] * pv_gelu_erf(g[2])), (half_t)(v[3] * pv_gelu_erf(g[3]))});
            }
            // logical output column of pair q: (n0 >> 1) + wn*(NF*8) + q*16 + fq*4; 16-byte stores via the lane swap (see below)
            half_t* orow = reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + (n0 >> 1) + wn * (NF * 8);
            static_assert(!GEGLU || NF == 4, "GEGLU epilogue pairs the two value|gate fragment pairs of a wave");
            {
                const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                const unsigned a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
                const int col = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
                typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
                *reinterpret_cast<uint4_t*>(orow + col) = uint4_t{a0, a1, b0, b1};
            }
        }
    } else {
        float4_t bias_v[NF];
#pragma unroll
        for (int ni = 0; ni < NF; ++ni)
            bias_v[ni] = p.bias ? *reinterpret_cast<const float4_t*>(p.bias + nbase + ni * 16) : float4_t{0.f, 0.f, 0.f, 0.f};
        // MULTI (tile loop): the K pipeline's registers stay live across the epilogue, so the residual is fetched in two batches of
        // two row-fragments instead of all four at once (40 -> 20 registers)
        constexpr int RB = MULTI ? 2 : MI;          // row-fragments per residual batch
        half4_t res[NF][MI];
        if (p.residual && !MULTI) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = min(m0 + wm * (MI * 16) + mi * 16 + fr, p.M - 1);
#pragma unroll
                for (int ni = 0; ni < NF; ++ni)
                    res[ni][mi] = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(p.residual) + (size_t)m * p.ldr + nbase + ni * 16);
            }
        }
        // optional GroupNorm column statistics of the tile being written (pv_gemm_params.colstats)
        const bool want_cs = CS && p.colstats != nullptr && !p.out_f32;
        float4_t cs[NF], cq[NF];
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) cs[ni] = cq[ni] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (MULTI && p.residual && mi % RB == 0) {
#pragma unroll
                for (int m2 = mi; m2 < mi + RB; ++m2) {
                    const int mm = min(m0 + wm * (MI * 16) + m2 * 16 + fr, p.M - 1);
#pragma unroll
                    for (int ni = 0; ni < NF; ++ni)
                        res[ni][m2] = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(p.residual) + (size_t)mm * p.ldr + nbase + ni * 16);
                }
            }
            const int m = m0 + wm * (MI * 16) + mi * 16 + fr;
            if (m >= p.M) continue;
            const float* radd = p.rowadd ? p.rowadd + (size_t)(m / hw_out) * p.rowadd_ld + nbase : nullptr;
            unsigned pk[NF][2];   // fp16-packed results of this row's NF fragments
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) {
                float4_t v = acc[ni][mi] + bias_v[ni];
                if (radd) v += *reinterpret_cast<const float4_t*>(radd + ni * 16);
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = epi_act(v[r], p.act);
                }
                if (p.residual) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)res[ni][mi][r];
                }
                if (p.out_f32) {
                    *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + nbase + ni * 16) = v;
                } else {
                    const half4_t hv = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    pk[ni][0] = __builtin_bit_cast(unsigned, half2_t{hv[0], hv[1]});
                    pk[ni][1] = __builtin_bit_cast(unsigned, half2_t{hv[2], hv[3]});
                    if (want_cs) {   // statistics of what is stored (the rounded values), as a pass over the tensor would see
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float f = (float)hv[r];
                            cs[ni][r] += f;
                            cq[ni][r] += f * f;
                        }
                    }
                }
            }
            if (!p.out_f32) {
                // Widen the stores to 16 B: a lane holds 4 columns (8 B) of fragment ni; v_permlane16_swap trades the even
                // lane-rows' fragment ni+1 against the odd lane-rows' fragment ni, after which every lane owns 8 CONSECUTIVE
                // columns -> half the store instructions at the same bytes (the m >= M rows skip as whole swap pairs: the
                // partner lane l^16 has the same row).
                half_t* orow = reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + n0 + wn * (NF * 16);
#pragma unroll
                for (int q = 0; q < NF / 2; ++q) {
                    const auto r0 = __builtin_amdgcn_permlane16_swap(pk[2 * q][0], pk[2 * q + 1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(pk[2 * q][1], pk[2 * q + 1][1], false, false);
                    const unsigned a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
                    const int col = (fq & 1) ? (2 * q + 1) * 16 + (fq - 1) * 4 : (2 * q) * 16 + fq * 4;
                    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<uint4_t*>(orow + col) = uint4_t{a0, a1, b0, b1};
                }
                if (NF & 1) {
                    typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<uint2_t*>(orow + (NF - 1) * 16 + fq * 4) = uint2_t{pk[NF - 1][0], pk[NF - 1][1]};
                }
            }
        }
        if (want_cs && m0 + wm * (MI * 16) < p.M) {
            // sum over the 16 lanes (rows) that share fq: DPP row rotations inside the 16-lane row, no LDS, fixed order
#pragma unroll
            for (int ni = 0; ni < NF; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cs[ni][r] = row16_sum(cs[ni][r]);
                    cq[ni][r] = row16_sum(cq[ni][r]);
                }
            if (fr == 0) {
                float* dst = p.colstats + ((size_t)(m0 / 64 + wm) * 2) * p.N + nbase;
#pragma unroll
                for (int ni = 0; ni < NF; ++ni) {
                    *reinterpret_cast<float4_t*>(dst + ni * 16) = cs[ni];
                    *reinterpret_cast<float4_t*>(dst + p.N + ni * 16) = cq[ni];
                }
            }
        }
    }
    // ---- next tile of the group ------------------------------------------------------------------------------------
    if (tn + 1 < tpw) {
#pragma unroll
        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = float4_t{0.f, 0.f, 0.f, 0.f};
        n0 += Cfg::BN;
        read_half(xa0, wb0, (tn + 1) * nk, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    }   // tiles of this workgroup
}

// Sum the split-K slabs in a fixed order (deterministic) and apply the GEMM epilogue.  One workgroup = 64 rows x 64 columns
// (thread = 4 rows x 4 columns), so that it can also leave the GroupNorm column statistics of its block behind
// (pv_gemm_params.colstats: per 64-row block and column, sum and sum of squares of the fp16-rounded outputs) - the split-K layers
// (16x16 / 8x8 levels) then need no statistics pass either.
// An in-kernel reduction by the tile's last-arriving K-slice (arrival ticket + agent-scope release / acquire) was built and
// measured in round 2: correct and deterministic, but the 80-KB-per-slice slab hand-off made the 8x8-level convs 2x slower
// (104 vs 50 us) - the "splitk-seam" result of MI355X_MICROARCH.md; the separate launch stays.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const pv_gemm_params_dev p, const int splits) {
    __shared__ float red[2][16][64];
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;           // column quad, row group
    const int n = (int)blockIdx.x * 64 + cq * 4;
    const int mb = (int)blockIdx.y * 64 + rg * 4;
    const bool ncol = n < p.N;
    float4_t cs = float4_t{0.f, 0.f, 0.f, 0.f}, cq2 = float4_t{0.f, 0.f, 0.f, 0.f};
    if (ncol) {
        const float4_t bias = p.bias ? *reinterpret_cast<const float4_t*>(p.bias + n) : float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mb + r;
            if (m >= p.M) break;
            float4_t v = float4_t{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < splits; ++s) v += *reinterpret_cast<const float4_t*>(p.splitk_ws + ((size_t)s * p.M + m) * p.N + n);
            v += bias;
            if (p.rowadd) v += *reinterpret_cast<const float4_t*>(p.rowadd + (size_t)(m / (p.hout * p.wout)) * p.rowadd_ld + n);
            if (p.act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = epi_act(v[j], p.act);
            }
            if (p.residual) {
                const half4_t rr = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(p.residual) + (size_t)m * p.ldr + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)rr[j];
            }
            if (p.out_f32) {
                *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = v;
            } else {
                half4_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = (half_t)v[j];
                    const float f = (float)o[j];
                    cs[j] += f;
                    cq2[j] += f * f;
                }
                *reinterpret_cast<half4_t*>(reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + n) = o;
            }
        }
    }
    if (p.colstats == nullptr) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][rg][cq * 4 + j] = cs[j];
        red[1][rg][cq * 4 + j] = cq2[j];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) a += red[which][g][col];            // fixed order
        const int nn = (int)blockIdx.x * 64 + col;
        if (nn < p.N) p.colstats[((size_t)blockIdx.y * 2 + which) * p.N + nn] = a;
    }
}

template <int NF, bool CONV, bool GEGLU, bool CS = false, bool MULTI = false, int MIV = 4>
int launch(const pv_gemm_params_dev& p, hipStream_t stream, int tpw = 1) {
    using Cfg = TileCfg<NF, MIV>;
    if (pv_gemm_probe) {                 // pv_gemm_conv_kernel_info: describe, do not launch
        const int splits_ = (!GEGLU && p.splitk > 1 && p.splitk_ws) ? p.splitk : 1;
        snprintf(pv_gemm_probe->name, sizeof(pv_gemm_probe->name), "gemm_conv_kernel<%d, %s, %s, %s, %s, %d>", NF, CONV ? "true" : "false", GEGLU ? "true" : "false",
                 CS ? "true" : "false", MULTI ? "true" : "false", MIV);
        pv_gemm_probe->wgs = (long)(((p.M + Cfg::BM - 1) / Cfg::BM) * (p.N / Cfg::BN) / tpw) * splits_;
        return 0;
    }
    static bool attr_set_dev[64] = {};   // per device: one process may drive several GPUs
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    auto kern = gemm_conv_kernel<NF, CONV, GEGLU, CS, MULTI, MIV>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM;
    const int tiles_n = p.N / Cfg::BN;
    const int nblk = tiles_m * tiles_n;
    // XCD footprint heuristic: walk M fastest when the weight panel is too big to sit in every XCD's L2
    const size_t wbytes = (size_t)p.N * p.taps * (p.c0 + p.c1) * 2;
    static const int order_env = getenv("PV_TILE_ORDER") ? atoi(getenv("PV_TILE_ORDER")) : -1;   // experiments: bit0 M-fastest, bit1 no XCD remap
    const int m_fast = order_env >= 0 ? order_env : ((wbytes > (3u << 20)) && tiles_n >= 8 ? 1 : 0);
    const int splits = (!GEGLU && p.splitk > 1 && p.splitk_ws) ? p.splitk : 1;
    hipLaunchKernelGGL(kern, dim3(nblk / tpw, splits), dim3(Cfg::THREADS), Cfg::SMEM_BYTES, stream, p, tiles_n / tpw, nblk / tpw, m_fast, tpw);
    if (splits > 1) return pv_gemm_splitk_reduce_launch(p, splits, stream);
    return PV_CHECK_LAUNCH();
}

// One tile shape (128 x BN x 64, two workgroups per CU) for every layer but the small Linear ones (64 x BN x 64, below): larger tiles lost
// on every UNet shape (see the header).
// tiles per workgroup: the shortest-K GEGLU layers (K = 320: five K-steps per tile, 10240 tiles at bs = 16) let one workgroup walk
// several N-tiles with a continuous K pipeline, as long as the launch keeps >= 1024 workgroups.  Measured (kbench, MI355X): -4 % at
// K = 320 (181.5 -> 174.6 us), nothing at K = 640, +6 % at K = 1280 - so only nk <= 5 uses it.
template <int NF>
int choose_tpw(const pv_gemm_params_dev& p) {
    static const int tpw_env = getenv("PV_GEMM_TPW") ? atoi(getenv("PV_GEMM_TPW")) : 0;      // experiments: force (1 = off)
    if (p.taps != 1 || (p.splitk > 1 && p.splitk_ws)) return 1;
    const int tiles_n = p.N / (NF * 32), nk = (p.c0 + p.c1) / BK;
    const long nblk = (long)((p.M + 127) / 128) * tiles_n;
    if (tpw_env > 0) return tiles_n % tpw_env == 0 ? tpw_env : 1;
    for (int c = 8; c >= 2; --c)
        if (tiles_n % c == 0 && nk <= 5 && nblk / c >= 1024) return c;
    return 1;
}

// One tile shape (128 x BN x 64, two workgroups per CU) for every layer but the small Linear ones (64 x BN x 64, below): larger tiles lost
// on every UNet shape (see the header).
template <int NF, bool CONV, bool GEGLU>
int dispatch(const pv_gemm_params_dev& p, hipStream_t stream) {
    if constexpr (!CONV && GEGLU) {       // the plain-epilogue instantiations spill with the tile loop (NF = 5: 300+ B of scratch): GEGLU only
        const int tpw = choose_tpw<NF>(p);
        if (tpw > 1) return launch<NF, false, true, false, true>(p, stream, tpw);
    }
    if constexpr (!GEGLU) {
        if (p.colstats && !(p.splitk > 1 && p.splitk_ws)) return launch<NF, CONV, false, true>(p, stream);   // split-K: the reduce launch makes them
    }
    if constexpr (!CONV && !GEGLU) {
        // 64-row tiles for the Linear layers whose 128-row tiling leaves the chip under two workgroups per CU (M = 4096 rows x N = 1280 at
        // the 16x16 level, the text / image-token K,V projections, CLIP / adapter layers): 4096x1280->1280 30.0 -> 25.8 us, 1232x768->640
        // 15.8 -> 10.4 us.  With >= 512 tiles the 128-row tile wins (65536x320->960: 70.8 vs 88.1 us).  PV_GEMM_MI2: 0 = never, N = below N tiles.
        static const int mi2_env = getenv("PV_GEMM_MI2") ? atoi(getenv("PV_GEMM_MI2")) : 512;
        const long tiles128 = (long)((p.M + 127) / 128) * (p.N / (NF * 32));
        if (!(p.splitk > 1 && p.splitk_ws) && tiles128 < mi2_env) return launch<NF, false, false, false, false, 2>(p, stream);
    }
    return launch<NF, CONV, GEGLU>(p, stream);
}

}  // namespace

int pv_gemm_splitk_reduce_launch(const pv_gemm_params_dev& p, int splits, hipStream_t stream) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((p.N + 63) / 64), (unsigned)((p.M + 63) / 64)), dim3(256), 0, stream, p, splits);
    return PV_CHECK_LAUNCH();
}

thread_local pv_launch_probe* pv_gemm_probe = nullptr;

extern "C" int pv_gemm_conv(const pv_gemm_params* pp, void* stream_);

// The kernel pv_gemm_conv would launch for this parameter block (symbol as rocprofv3 prints it, workgroups incl. split-K slices): the same validation and
// the same dispatch code, with the launchers in describe-only mode.  No HIP call is made.
extern "C" int pv_gemm_conv_kernel_info(const pv_gemm_params* pp, char* name, int32_t name_len, int64_t* workgroups) {
    if (!pp || !name || name_len <= 0) return (int)hipErrorInvalidValue;
    pv_launch_probe probe;
    probe.name[0] = 0;
    probe.wgs = 0;
    pv_gemm_probe = &probe;
    const int rc = pv_gemm_conv(pp, nullptr);
    pv_gemm_probe = nullptr;
    if (rc != 0) return rc;
    if ((int)strlen(probe.name) >= name_len) return (int)hipErrorInvalidValue;
    strcpy(name, probe.name);
    if (workgroups) *workgroups = probe.wgs;
    return 0;
}

extern "C" int pv_gemm_conv(const pv_gemm_params* pp, void* stream_) {
    pv_gemm_params_dev p;
    static_cast<pv_gemm_params&>(p) = *pp;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int cin = p.c0 + p.c1;
    if (p.M <= 0 || p.N <= 0 || cin <= 0 || (cin % 64) || (p.c0 % 64) || (p.taps != 1 && p.taps != 9) || p.act == PV_ACT_GELU || p.splitk < 0 || p.splitk > 16 ||
        !p.a0 || !p.w || !p.out || (p.c1 && !p.a1) || p.hout * p.wout <= 0)
        return (int)hipErrorInvalidValue;
    if (p.colstats && (p.geglu || p.out_f32)) return (int)hipErrorInvalidValue;
    if (p.taps == 9 && !(p.pad == 1 || (p.pad == 0 && p.stride == 2 && !p.upsample))) return (int)hipErrorInvalidValue;
    {
        const size_t a_rows = p.taps == 9 ? (size_t)p.batch * p.hin * p.win : (size_t)p.M;
        const size_t b0 = ((a_rows - 1) * p.lda0 + p.c0) * 2, b1 = p.c1 ? ((a_rows - 1) * p.lda1 + p.c1) * 2 : 0;
        const size_t bw = (size_t)p.N * p.taps * cin * 2;
        if (b0 >= (1ull << 31) || b1 >= (1ull << 31) || bw >= (1ull << 31)) return (int)hipErrorInvalidValue;
        p.a0_bytes = (uint32_t)b0;
        p.a1_bytes = (uint32_t)b1;
        p.w_bytes = (uint32_t)bw;
    }
    if (p.geglu && (p.taps != 1 || (p.N % 128))) return (int)hipErrorInvalidValue;
    // the GroupNorm fold exists for 3x3 convs only, with no activation or SiLU behind the affine part: a Linear launch carrying a_norm would
    // run the 256-row tile and silently ignore it (the taps == 1 branch of pv_conv_big_launch returns before any a_norm check)
    if (p.a_norm && (p.taps != 9 || (p.a_norm_act != PV_ACT_NONE && p.a_norm_act != PV_ACT_SILU))) return (int)hipErrorInvalidValue;
    {                                                    // the 256-row tile (pv_convbig.hip) where the launch fills the chip with it
        const int rc = pv_conv_big_launch(p, stream);
        if (rc >= 0) return rc;
        if (rc == -2 || p.a_norm) return (int)hipErrorInvalidValue;     // the GroupNorm fold (a_norm) exists on the LDS-resident-patch path only (header)
    }
    if (p.ln_rowsum) return (int)hipErrorInvalidValue;   // the LayerNorm fold exists on the 256-row-tile Linear path only (header)
    if (p.geglu) return dispatch<4, false, true>(p, stream);
    const bool conv = p.taps == 9;
    if (p.N % 160 == 0) return conv ? dispatch<5, true, false>(p, stream) : dispatch<5, false, false>(p, stream);
    if (p.N % 128 == 0) return conv ? dispatch<4, true, false>(p, stream) : dispatch<4, false, false>(p, stream);
    return (int)hipErrorInvalidValue;
}
