// Self-attention backward at d = 40 / 80 (attn1 of the 64 x 64 / 32 x 32 levels of the UNet, train.py:505-536 of the reference: the gradient crosses
// the frozen stock AttnProcessor2_0 blocks; written out below for d = 40, the d = 80 parameters are in BW<80>) as 8-wave workgroups whose SIMD partners alternate matrix and vector segments - the structure of
// attn8_kernel (pv_attn.hip) applied to the two passes of pv_attention_backward (pv_train.hip):
//
//   pass dKV: a workgroup OWNS 512 keys (wave: 64 = four 16-key MFMA columns, K / V fragments in registers) and WALKS the queries
//   pass dQ : a workgroup OWNS 512 queries (wave: 64, scaled-Q / dO fragments in registers)            and WALKS the keys
//
// Both passes are the same loop over 32-row STEPS of the walked side (two per 64-row LDS tile):
//   matrix segment j:  [the gradient products of step j: dV^T += dO^T P, dK^T += Q^T dS  |  dQ^T += K^T dS^T]   12 / 24 MFMAs
//                      [S, dP of step j + 1: 48-deep contractions (16x16x32 + 16x16x16)]                          32 MFMAs (24 x 16x16x32 worth)
//   vector segment j:  P = exp2(S), dS = P dP, both rounded to fp16 MFMA operands (32 exponentials per lane)
// Waves 4-7 run one barrier interval behind waves 0-3, so on every SIMD one wave is in its matrix segment while its partner exponentiates.
//
// What the older kernels spent their time on, and what replaces it (profiles/r05_attn_bwd_*.txt):
//   * one 16-row fragment pair per wave -> four: every fragment read from LDS feeds four MFMAs instead of two;
//   * d = 40 padded to a 64-deep contraction -> 48 (the tail step is a 16x16x16): a quarter fewer score-MFMA cycles;
//   * the softmax statistics ride in the padding: column 40 / 41 of a scaled-query row hold -lse as an fp16 (hi, lo) pair and the same
//     columns of a dO row hold -delta, the matching K / V columns hold 1.0 - S' = q k - lse and dP - delta come out of the MFMA, no
//     accumulator-initialising moves (128 per step), no statistics tiles in LDS;
//   * global -> registers -> ds_write staging with two __syncthreads per tile -> LDS-DMA into a four-slot ring, raw s_barrier per segment.
// The scaled queries and dO are first copied head-major into 48-column rows (`prep`, one launch: it also computes delta), so a walked
// 64-row tile of the dKV pass is 6 KiB of contiguous memory.
//
// Taken by pv_attention_backward (pv_train.hip) for d = 40 (nq and nk multiples of 512) and d = 80 (multiples of 256), no mask, when the caller passes
// the workspace.
#include "pv_common.h"

// variant taken when PV_ATTN8_BWD is not set (bits: attn8_bwd_kernel)
#ifndef PV_ATTN8_BWD_DEFAULT
#define PV_ATTN8_BWD_DEFAULT 81
#endif
#ifndef PV_ATTN8_BWD_PAD_KV
#define PV_ATTN8_BWD_PAD_KV 1      // words behind a 32-byte alignment in front of the step loop; -1: no directive (1 / 3 / 5: -0.6 % of the launch)
#endif
#ifndef PV_ATTN8_BWD_PAD_Q
#define PV_ATTN8_BWD_PAD_Q 7       // (-0.4 %)
#endif

namespace {

// Per head dimension: row stride RS of the LDS / workspace images in halfs (a multiple of 8 x odd dwords: conflict-free ds_read_b64_tr_b16, ACfg::VS),
// fragments per wave NF (owned rows per workgroup = 8 waves x NF x 16), d-major accumulator fragments DT, contraction = KS32 32-deep steps (+ a
// 16-deep TAIL).  The statistics columns sit right behind the data: d = 40: columns 40 / 41 inside the 16-deep tail (48-deep contraction);
// d = 80: columns 80 / 81 inside a third 32-deep step (96-deep; 64 + 16 + a fourth 16-deep step for the two columns would cost the same cycles).
template <int D> struct BW;
template <> struct BW<40> { static constexpr int RS = 48, NF = 4, DT = 3, KS32 = 1; static constexpr bool TAIL = true; };
template <> struct BW<80> { static constexpr int RS = 112, NF = 2, DT = 5, KS32 = 3; static constexpr bool TAIL = false; };
template <int D> struct BWD : BW<D> {
    using B = BW<D>;
    static constexpr int CPR = B::RS / 8;                 // 16-byte chunks per image row
    static constexpr int DCH = D / 8;                     // data chunks; chunk DCH = {1 | -stat hi, 1 | -stat lo, 0 x 6}
    static constexpr int KCH = (B::KS32 * 32 + (B::TAIL ? 16 : 0)) / 8;      // chunks the contractions read (the ones behind DCH are zero)
    static constexpr int TILE = 64 * B::RS;               // halfs of a 64-row tile image
    static constexpr int PPM = TILE * 2 / 1024;           // 1-KiB LDS-DMA pieces per matrix
    static constexpr int STAGE = 2 * TILE;                // ring slot: matrix 0 (K | scaled Q) then matrix 1 (V | dO)
    static constexpr int OWN = 8 * B::NF * 16;            // owned rows per workgroup
    static constexpr int NIT = (2 * PPM + 7) / 8;         // LDS-DMA instructions per wave and tile
    static_assert(TILE * 2 % 1024 == 0 && D % 8 == 0 && KCH <= CPR && DCH < KCH, "image layout");
};

#ifdef PV_ATTN8_BWD_STAMPS
__device__ unsigned long long pv_attn8_bwd_stamps[2 * 8 * 8];
#define B8_NOW() __builtin_amdgcn_s_memtime()
#endif

__device__ __forceinline__ half8_t bz8() { return half8_t{0, 0, 0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ half8_t bone8() {
    half8_t v = bz8();
    v[0] = (half_t)1.0f;
    v[1] = (half_t)1.0f;
    return v;
}

// fp16 (hi, lo) pair of a float in columns 0 / 1 of a chunk: hi + lo == x to ~2^-22 relative for normal-range x (exact products with the 1.0
// columns, fp32 accumulate)
__device__ __forceinline__ half8_t split_hi_lo(float x) {
    half8_t t = bz8();
    t[0] = (half_t)x;
    t[1] = (half_t)(x - (float)t[0]);
    return t;
}

// qs[b][h][q][RS] = {fp16(q * scale * log2 e) (the forward kernel's rounding), -lse (hi, lo), zeros up to the contraction length};
// dos = {dO, -delta (hi, lo), zeros};  delta[b][h][q] = sum_c dO[q][c] O[q][c].  One thread per (row, head), heads fastest: a wave reads whole rows.
template <int D>
__global__ __launch_bounds__(256) void attn8_bwd_prep_kernel(const pv_attn_bwd_params p, half_t* __restrict__ qs_ws, half_t* __restrict__ do_ws) {
    using C = BWD<D>;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.batch * p.nq * p.heads;
    if (idx >= total) return;
    const int h = (int)(idx % p.heads);
    const long row = idx / p.heads;                                  // b * nq + q
    const int q = (int)(row % p.nq), b = (int)(row / p.nq);
    const half_t* o = reinterpret_cast<const half_t*>(p.out) + (size_t)row * p.ldo + h * D;
    const half_t* g = reinterpret_cast<const half_t*>(p.dout) + (size_t)row * p.lddo + h * D;
    const half_t* qr = reinterpret_cast<const half_t*>(p.q) + (size_t)row * p.ldq + h * D;
    const size_t bhq = ((size_t)b * p.heads + h) * p.nq + q;
    half_t* qs = qs_ws + bhq * C::RS;
    half_t* ds = do_ws + bhq * C::RS;
    const float qscale = rsqrtf((float)D) * 1.4426950408889634f;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 8) {
        const half8_t x = *reinterpret_cast<const half8_t*>(o + c), y = *reinterpret_cast<const half8_t*>(g + c);
        const half8_t qv = *reinterpret_cast<const half8_t*>(qr + c);
        half8_t sv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a += (float)x[j] * (float)y[j];
            sv[j] = (half_t)((float)qv[j] * qscale);
        }
        *reinterpret_cast<half8_t*>(qs + c) = sv;
        *reinterpret_cast<half8_t*>(ds + c) = y;
    }
    p.delta[bhq] = a;
    *reinterpret_cast<half8_t*>(qs + D) = split_hi_lo(-p.lse[bhq]);
    *reinterpret_cast<half8_t*>(ds + D) = split_hi_lo(-a);
#pragma unroll
    for (int ch = C::DCH + 1; ch < C::KCH; ++ch) {
        *reinterpret_cast<half8_t*>(qs + ch * 8) = bz8();
        *reinterpret_cast<half8_t*>(ds + ch * 8) = bz8();
    }
}

// A fragment of the TRANSPOSE of a [rows][RS] image: output rows d = dv0 + fr.., the 8 contraction slots {r0 + 4 fq .. + 3, r0 + 16 + 4 fq .. + 3}
template <int RS>
__device__ __forceinline__ half8_t bt_frag(const half_t* sR, int r0, int dv0, int fr, int fq) {
    const half_t* a = sR + (r0 + fq * 4 + (fr >> 2)) * RS + dv0 + (fr & 3) * 4;
    const fp16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a));
    const fp16x4_t t2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a + 16 * RS));
    const half4_t h1 = __builtin_bit_cast(half4_t, t1), h2 = __builtin_bit_cast(half4_t, t2);
    return __builtin_shufflevector(h1, h2, 0, 1, 2, 3, 4, 5, 6, 7);
}

// KV = true : own keys (K, V from the caller's tensors), walk queries (qs48 / do48 tiles);   accumulate dV^T (A) and dK^T (B)
// KV = false: own queries (qs48 / do48 rows),            walk keys (K, V from the caller's tensors); accumulate dQ^T (B)
// VAR bit 0: stagger; bit 4 (16): s_setprio 1 in the matrix segments; bit 6 (64): the vector segment's three passes are fenced (the source order alone
// already gives that schedule).  Measured (EXPERIMENTS.md round 5, B = 16, N = 4096, sustained, same box): 4-wave passes 2 035 us; 0: 1 580;
// 1: 1 480 -> 1 396 with the three-pass vector segment; 81 (default): 1 276-1 310; static priority for waves 4-7, priority in the VECTOR segment
// and a prefetch of the dQ pass's K^T fragments at the head of the vector segment (attn8_kernel's bits 6 + 8): 1 362 / 1 506 / +2 % - removed.
template <int BD, bool KV, int VAR>
__global__ __launch_bounds__(512, 2) void attn8_bwd_kernel(const pv_attn_bwd_params p, const half_t* __restrict__ qs48, const half_t* __restrict__ do48) {
    using C = BWD<BD>;
    constexpr int BRS = C::RS, BNF = C::NF, BDT = C::DT, BTILE = C::TILE, BSTAGE = C::STAGE, KS32 = C::KS32;
    constexpr bool TAIL = C::TAIL;
    constexpr bool STAGGER = (VAR & 1) != 0, SEGPRIO = (VAR & 16) != 0, PRO_PRIO = SEGPRIO;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sbase = reinterpret_cast<half_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    const bool late = STAGGER && wave >= 4;
    const int n_own = KV ? p.nk : p.nq, n_walk = KV ? p.nq : p.nk;
    const int nwt = n_own / C::OWN;
    const int rid = pv_xcd_remap((int)blockIdx.x, (int)gridDim.x);     // the workgroups of one (sample, head) share an XCD: its L2 holds the walked side once
    const int ot = rid % nwt, h = (rid / nwt) % p.heads, b = rid / (nwt * p.heads);
    const size_t bh = (size_t)b * p.heads + h;
    const half_t* Kg = reinterpret_cast<const half_t*>(p.k) + (size_t)b * p.nk * p.ldk + h * BD;
    const half_t* Vg = reinterpret_cast<const half_t*>(p.v) + (size_t)b * p.nk * p.ldv + h * BD;
    const half_t* Qs = qs48 + bh * p.nq * BRS;
    const half_t* DOs = do48 + bh * p.nq * BRS;

    if (!KV) {
        // columns D / D + 1 of every K and V row in the ring = 1.0 (they meet -lse / -delta of the query rows), the rest of the contraction length = 0;
        // the LDS-DMA leaves these chunks alone
        for (int i = tid; i < 4 * 2 * 64; i += 512) {
            half_t* row = sbase + (i >> 7) * BSTAGE + ((i >> 6) & 1) * BTILE + (i & 63) * BRS;
            *reinterpret_cast<half8_t*>(row + BD) = bone8();
#pragma unroll
            for (int ch = C::DCH + 1; ch < C::KCH; ++ch) *reinterpret_cast<half8_t*>(row + ch * 8) = bz8();
        }
    }

    // ---- the owned side: B operands (column = own row fr of fragment i; k slots: 32 ks + 8 fq .. + 7 of a 32-deep step, 32 KS32 + 4 fq .. + 3 of the tail)
    half8_t b0[BNF][KS32], b1[BNF][KS32];          // matrix 0 (K | scaled Q), matrix 1 (V | dO)
    half4_t t0[TAIL ? BNF : 1], t1[TAIL ? BNF : 1];
    int orow[BNF];
#pragma unroll
    for (int i = 0; i < BNF; ++i) {
        orow[i] = ot * C::OWN + (wave * BNF + i) * 16 + fr;
        const half_t* r0p = KV ? Kg + (size_t)orow[i] * p.ldk : Qs + (size_t)orow[i] * BRS;
        const half_t* r1p = KV ? Vg + (size_t)orow[i] * p.ldv : DOs + (size_t)orow[i] * BRS;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
            const int ch = ks * 4 + fq;                  // K / V rows end at column D: the statistics chunk is 1.0 there, zeros behind it
            if (!KV || ch < C::DCH) {
                b0[i][ks] = *reinterpret_cast<const half8_t*>(r0p + ch * 8);
                b1[i][ks] = *reinterpret_cast<const half8_t*>(r1p + ch * 8);
            } else {
                b0[i][ks] = b1[i][ks] = ch == C::DCH ? bone8() : bz8();
            }
        }
        if constexpr (TAIL) {
            const int col = KS32 * 32 + fq * 4;
            const half4_t one = half4_t{(half_t)1.0f, (half_t)1.0f, 0, 0}, zero = half4_t{0, 0, 0, 0};
            if (!KV || col < BD) {
                t0[i] = *reinterpret_cast<const half4_t*>(r0p + col);
                t1[i] = *reinterpret_cast<const half4_t*>(r1p + col);
            } else {
                t0[i] = t1[i] = col == BD ? one : zero;
            }
        }
    }
    float4_t accA[KV ? BNF : 1][BDT], accB[BNF][BDT];
#pragma unroll
    for (int i = 0; i < BNF; ++i)
#pragma unroll
        for (int f = 0; f < BDT; ++f) {
            accB[i][f] = float4_t{0.f, 0.f, 0.f, 0.f};
            if constexpr (KV) accA[i][f] = float4_t{0.f, 0.f, 0.f, 0.f};
        }
    float4_t s[2][BNF], dp[2][BNF];    // [16-row block of the step][own fragment]
    half8_t pb[KV ? BNF : 1], dsb[BNF];

    // ---- matrix segment, second half: S' = rows(matrix 0) . own 0, dP' = rows(matrix 1) . own 1 for step j
    auto sdp = [&](int j) {
        const half_t* s0 = sbase + ((j >> 1) & 3) * BSTAGE + (j & 1) * 32 * BRS;
        const half_t* s1 = s0 + BTILE;
        half8_t a0[2][KS32], a1[2][KS32];
        half4_t c0[2], c1[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int off = (rb * 16 + fr) * BRS;
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
                a0[rb][ks] = *reinterpret_cast<const half8_t*>(s0 + off + ks * 32 + fq * 8);
                a1[rb][ks] = *reinterpret_cast<const half8_t*>(s1 + off + ks * 32 + fq * 8);
            }
            if constexpr (TAIL) {
                c0[rb] = *reinterpret_cast<const half4_t*>(s0 + off + KS32 * 32 + fq * 4);
                c1[rb] = *reinterpret_cast<const half4_t*>(s1 + off + KS32 * 32 + fq * 4);
            }
        }
        const float4_t z = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < BNF; ++i) {
                    s[rb][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[rb][ks], b0[i][ks], ks == 0 ? z : s[rb][i], 0, 0, 0);
                    dp[rb][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[rb][ks], b1[i][ks], ks == 0 ? z : dp[rb][i], 0, 0, 0);
                }
            // every step of a chain stays a whole group of independent MFMAs behind the one it accumulates onto: a 16x16x16 tail that hipcc put
            // ONE instruction behind its 16x16x32 returned wrong sums (EXPERIMENTS.md, round 5)
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (TAIL) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < BNF; ++i) {
                    s[rb][i] = __builtin_amdgcn_mfma_f32_16x16x16f16(c0[rb], t0[i], s[rb][i], 0, 0, 0);
                    dp[rb][i] = __builtin_amdgcn_mfma_f32_16x16x16f16(c1[rb], t1[i], dp[rb][i], 0, 0, 0);
                }
        }
    };
    // ---- vector segment: P = exp2(S'), dS = P dP' -> fp16 B operands (k slots {4 fq + r, 16 + 4 fq + r} of the step's 32 walked rows: the
    //      order the transposed fragment reads below use)
    auto vec = [&]() {
        // three passes, in this order: the 32 exponentials (quarter rate, ~16 cycles each, independent), then the products, then the roundings.  Left to
        // itself hipcc puts each product two instructions behind its exponential and the wave waits out the transcendental latency 32 times
        // (1 490 cycles for ~110 VALU instructions, profiles/r05_attn_bwd_stamps.txt)
#pragma unroll
        for (int i = 0; i < BNF; ++i)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[rb][i][r] = PV_EXP2(s[rb][i][r]);
        if (VAR & 64) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < BNF; ++i)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) dp[rb][i] *= s[rb][i];
        if (VAR & 64) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < BNF; ++i)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (KV) pb[i][rb * 4 + r] = (half_t)s[rb][i][r];
                    dsb[i][rb * 4 + r] = (half_t)dp[rb][i][r];
                }
    };
    // ---- matrix segment, first half: the gradient products of step j
    auto grad = [&](int j) {
        const half_t* s0 = sbase + ((j >> 1) & 3) * BSTAGE;
        const half_t* s1 = s0 + BTILE;
        const int r0 = (j & 1) * 32;
#pragma unroll
        for (int f = 0; f < BDT; ++f) {
            const half8_t x0 = bt_frag<BRS>(s0, r0, f * 16, fr, fq);
            if constexpr (KV) {
                const half8_t x1 = bt_frag<BRS>(s1, r0, f * 16, fr, fq);
#pragma unroll
                for (int i = 0; i < BNF; ++i) accA[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, pb[i], accA[i][f], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < BNF; ++i) accB[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x0, dsb[i], accB[i][f], 0, 0, 0);
        }
    };

    // ---- LDS-DMA: a tile = 2 PPM pieces of 1 KiB (PPM per matrix); wave w stages pieces w, w + 8, ...
    const __amdgpu_buffer_rsrc_t r0 = KV ? __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Qs), 0, p.nq * BRS * 2, 0x00020000)
                                         : __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Kg), 0, ((p.nk - 1) * p.ldk + BD) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = KV ? __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(DOs), 0, p.nq * BRS * 2, 0x00020000)
                                         : __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Vg), 0, ((p.nk - 1) * p.ldv + BD) * 2, 0x00020000);
    // piece pc of matrix m: lane -> 16-byte chunk 64 pc + lane of the [64][CPR] image; K / V rows bring their data chunks only (the constant chunks
    // were written once), workspace rows everything the contractions read
    unsigned poff[C::NIT];
    int plds[C::NIT];
    bool plive[C::NIT];
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
        const int pc = wave + 8 * it, m = pc >= C::PPM ? 1 : 0, pcl = pc - m * C::PPM;
        const int c = 64 * pcl + lane, row = c / C::CPR, ch = c - row * C::CPR;
        plive[it] = pc < 2 * C::PPM && ch < (KV ? C::KCH : C::DCH);
        poff[it] = KV ? (unsigned)(c * 16) : (unsigned)(row * (m ? p.ldv : p.ldk) * 2 + ch * 16);
        plds[it] = m * BTILE * 2 + pcl * 1024;
    }
    const int step0 = KV ? BTILE * 2 : 64 * p.ldk * 2, step1 = KV ? BTILE * 2 : 64 * p.ldv * 2;     // bytes per tile in the source
    auto issue_tile = [&](int t) {
        char* slot = reinterpret_cast<char*>(sbase + (t & 3) * BSTAGE);
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) {
            const bool m1 = wave + 8 * it >= C::PPM;      // wave-uniform
            if (plive[it]) {
                if (m1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, PV_LDS_PTR(slot + plds[it]), 16, (int)poff[it], t * step1, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, PV_LDS_PTR(slot + plds[it]), 16, (int)poff[it], t * step0, 0, 0);
            }
        }
    };
    auto interval = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // hipcc sinks pure arithmetic across s_barrier into the block of first use: every segment's results are pinned on their side (pv_attn.hip)
    auto pin_sdp = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < BNF; ++i) asm volatile("" : "+v"(s[rb][i]), "+v"(dp[rb][i]));
    };
    auto pin_vec = [&]() {
#pragma unroll
        for (int i = 0; i < BNF; ++i) {
            asm volatile("" : "+v"(dsb[i]));
            if constexpr (KV) asm volatile("" : "+v"(pb[i]));
        }
    };
    auto pin_acc = [&]() {
#pragma unroll
        for (int i = 0; i < BNF; ++i)
#pragma unroll
            for (int f = 0; f < BDT; ++f) {
                asm volatile("" : "+v"(accB[i][f]));
                if constexpr (KV) asm volatile("" : "+v"(accA[i][f]));
            }
    };

    const int ntiles = n_walk / 64, nsteps = 2 * ntiles;
    __syncthreads();
    issue_tile(0);
    if (ntiles > 1) issue_tile(1);
    if (ntiles > 2) issue_tile(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    interval();
    if (late) interval();
    if (PRO_PRIO) __builtin_amdgcn_s_setprio(1);
    sdp(0);
    pin_sdp();
    if (PRO_PRIO) __builtin_amdgcn_s_setprio(0);
#ifdef PV_ATTN8_BWD_STAMPS
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_t0 = 0, st_t1, st_t2, st_t3, st_c0 = 0, st_r0 = 0;
#endif
    // the step loop's offset inside the 32-byte instruction-fetch windows, pinned (pv_attn.hip, attn8_kernel: 2.5 % between the best and the worst offset)
    constexpr int LOOP_PAD = KV ? PV_ATTN8_BWD_PAD_KV : PV_ATTN8_BWD_PAD_Q;
    if constexpr (LOOP_PAD >= 0) asm volatile(".p2align 5\n .rept %0\n s_nop 0\n .endr" ::"n"(LOOP_PAD));
    for (int j = 0; j < nsteps; ++j) {
#ifdef PV_ATTN8_BWD_STAMPS
        st_t3 = B8_NOW();
#endif
        interval();
#ifdef PV_ATTN8_BWD_STAMPS
        st_t1 = B8_NOW();
        if (j == 16) { st_c0 = st_t1; st_r0 = __builtin_amdgcn_s_memrealtime(); }
        if (j > 16 && j < nsteps - 16) { st_acc[2] += st_t3 - st_t0; st_acc[3] += st_t1 - st_t3; }
#endif
        vec();
        pin_vec();
#ifdef PV_ATTN8_BWD_STAMPS
        st_t2 = B8_NOW();
#endif
        interval();
#ifdef PV_ATTN8_BWD_STAMPS
        st_t0 = B8_NOW();
        if (j >= 16 && j < nsteps - 16) { st_acc[0] += st_t2 - st_t1; st_acc[1] += st_t0 - st_t2; }
        if (j == nsteps - 17 && blockIdx.x == 300 && lane == 0) {
            unsigned long long* o8 = pv_attn8_bwd_stamps + ((KV ? 1 : 0) * 8 + wave) * 8;
            o8[0] = st_acc[0]; o8[1] = st_acc[1]; o8[2] = st_acc[2]; o8[3] = st_acc[3];
            o8[4] = st_t0 - st_c0; o8[5] = __builtin_amdgcn_s_memrealtime() - st_r0; o8[6] = (unsigned long long)(nsteps - 32);
        }
#endif
        if (SEGPRIO) __builtin_amdgcn_s_setprio(1);
        // this wave's pieces (issued at the end of an earlier matrix segment) have landed; hipcc puts this wait in front of the first LDS read
        // behind an LDS-DMA issue anyway (it cannot tell the ring slots apart)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grad(j);
        sdp(j + 1);                                   // past the last step: a stale slot that nothing reads (no branch between the products)
        pin_acc();
        pin_sdp();
        // tile t + 3 goes into the slot of tile t - 1: its last readers (the late waves' products of step 2 t - 1) finished two intervals ago; first
        // read in matrix segment 2 t + 5, behind every wave's landed-wait and a barrier
        if ((j & 1) && (j >> 1) + 3 < ntiles) issue_tile((j >> 1) + 3);
        if (SEGPRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (STAGGER && !late) interval();

    if constexpr (KV) {
        half_t* dK = reinterpret_cast<half_t*>(p.dk) + (size_t)b * p.nk * p.lddk + h * BD;
        half_t* dV = reinterpret_cast<half_t*>(p.dv) + (size_t)b * p.nk * p.lddv + h * BD;
#pragma unroll
        for (int i = 0; i < BNF; ++i)
#pragma unroll
            for (int f = 0; f < BDT; ++f) {
                const int dv = f * 16 + fq * 4;
                if (dv < BD) {
                    half4_t ok_, ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ok_[r] = (half_t)(accB[i][f][r] * 0.6931471805599453f);      // the queries carried log2 e
                        ov[r] = (half_t)accA[i][f][r];
                    }
                    *reinterpret_cast<half4_t*>(dK + (size_t)orow[i] * p.lddk + dv) = ok_;
                    *reinterpret_cast<half4_t*>(dV + (size_t)orow[i] * p.lddv + dv) = ov;
                }
            }
    } else {
        const float scale = rsqrtf((float)BD);
        half_t* dQ = reinterpret_cast<half_t*>(p.dq) + (size_t)b * p.nq * p.lddq + h * BD;
#pragma unroll
        for (int i = 0; i < BNF; ++i)
#pragma unroll
            for (int f = 0; f < BDT; ++f) {
                const int dv = f * 16 + fq * 4;
                if (dv < BD) {
                    half4_t o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)(accB[i][f][r] * scale);
                    *reinterpret_cast<half4_t*>(dQ + (size_t)orow[i] * p.lddq + dv) = o;
                }
            }
    }
}

}  // namespace

#ifdef PV_ATTN8_BWD_STAMPS
extern "C" int pv_attn8_bwd_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pv_attn8_bwd_stamps), sizeof(pv_attn8_bwd_stamps));
}
#endif

namespace {
// the forms instantiated below (PV_B8_CASE): anything else named by PV_ATTN8_BWD is not eligible and falls back to the 4-wave kernels
constexpr bool attn8_bwd_variant_built(int var) { return var == 0 || var == 1 || var == 65 || var == 81; }
static_assert(attn8_bwd_variant_built(PV_ATTN8_BWD_DEFAULT), "the default variant must be an instantiated one");

template <int D>
int launch_attn8_bwd(const pv_attn_bwd_params& p, hipStream_t s, int var) {
    using C = BWD<D>;
    void (*kq)(const pv_attn_bwd_params, const half_t*, const half_t*) = nullptr;
    void (*kkv)(const pv_attn_bwd_params, const half_t*, const half_t*) = nullptr;
    switch (var) {
#define PV_B8_CASE(V) case V: kq = attn8_bwd_kernel<D, false, V>; kkv = attn8_bwd_kernel<D, true, V>; break;
        PV_B8_CASE(0) PV_B8_CASE(1) PV_B8_CASE(65) PV_B8_CASE(81)
#undef PV_B8_CASE
        default: return (int)hipErrorInvalidValue;
    }
    constexpr int smem = 4 * C::STAGE * 2;                // d = 40: 48 KiB; d = 80: 112 KiB
    static bool attr_set[64][128] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63][var & 127]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set[dev & 63][var & 127] = true;
    }
    half_t* qs = reinterpret_cast<half_t*>(p.ws);
    half_t* dos = qs + (size_t)p.batch * p.heads * p.nq * C::RS;
    const long rows = (long)p.batch * p.nq * p.heads;
    hipLaunchKernelGGL(attn8_bwd_prep_kernel<D>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, p, qs, dos);
    const unsigned bh = (unsigned)(p.batch * p.heads);
    hipLaunchKernelGGL(kkv, dim3(bh * (unsigned)(p.nk / C::OWN)), dim3(512), smem, s, p, (const half_t*)qs, (const half_t*)dos);
    hipLaunchKernelGGL(kq, dim3(bh * (unsigned)(p.nq / C::OWN)), dim3(512), smem, s, p, (const half_t*)qs, (const half_t*)dos);
    return PV_CHECK_LAUNCH();
}
}  // namespace

// bytes of workspace pv_attention_backward needs to take this path: scaled queries and dO head-major in RS-column rows
__attribute__((visibility("hidden"))) size_t pv_attn8_bwd_ws_bytes(const pv_attn_bwd_params& p) {
    return (size_t)2 * p.batch * p.heads * p.nq * (p.d == 80 ? BWD<80>::RS : BWD<40>::RS) * sizeof(half_t);
}

__attribute__((visibility("hidden"))) bool pv_attn8_bwd_eligible(const pv_attn_bwd_params& p) {
    if ((p.d != 40 && p.d != 80) || p.causal || !p.ws) return false;
    const int own = p.d == 80 ? BWD<80>::OWN : BWD<40>::OWN, rs = p.d == 80 ? BWD<80>::RS : BWD<40>::RS;
    if (p.nq % own || p.nk % own) return false;
    if ((size_t)p.ws_bytes < pv_attn8_bwd_ws_bytes(p)) return false;
    if ((size_t)p.nk * (size_t)(p.ldk > p.ldv ? p.ldk : p.ldv) * 2 >= (1ull << 31) || (size_t)p.nq * rs * 2 >= (1ull << 31)) return false;
    const char* env = getenv("PV_ATTN8_BWD");             // read per call (tests run both forms in one process); -1 = the 4-wave kernels
    if (env && !attn8_bwd_variant_built(atoi(env))) return false;      // negative or not an instantiated form: the 4-wave kernels take the launch
    const char* envmin = getenv("PV_ATTN8_BWD_MIN");      // fewest workgroups a pass must have (default: half the CUs)
    const long wgs = (long)p.batch * p.heads * ((p.nq < p.nk ? p.nq : p.nk) / own);
    return wgs >= (envmin ? atol(envmin) : 128);
}

__attribute__((visibility("hidden"))) int pv_attn8_bwd_launch(const pv_attn_bwd_params& p, hipStream_t s) {
    const char* env = getenv("PV_ATTN8_BWD");
    const int var = env ? atoi(env) : PV_ATTN8_BWD_DEFAULT;
    return p.d == 80 ? launch_attn8_bwd<80>(p, s, var) : launch_attn8_bwd<40>(p, s, var);
}
