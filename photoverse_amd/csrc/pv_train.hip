// Backward kernels of the stock SD-v1.5 / CLIP blocks the training step back-propagates THROUGH (train.py:505-506, :536: the
// UNet and the text encoder are frozen apart from LoRA, but the gradient has to cross them to reach the adapters, to_k_ip / to_v_ip and
// the LoRA factors):
//
//   pv_attention_backward     flash-style backward of softmax(Q K^T / sqrt(d)) V (attn1 of every transformer block, CLIP text layers
//                             with the causal mask): MFMA v_mfma_f32_16x16x32_f16, probabilities recomputed from the forward's
//                             log-sum-exp, no atomics - one kernel owns 64 keys and walks the queries (dK, dV), one owns 64 queries
//                             and walks the keys (dQ); fixed summation order
//   pv_groupnorm_backward     GroupNorm (+ SiLU) backward over NHWC fp16, two-source channel concat like the forward
//   pv_geglu_backward, pv_act_backward, pv_add_rows_f16, pv_dilate2x, pv_pool2x_sum, pv_sign_f32, pv_gather_rows_f32
//                             elementwise / layout pieces (GEGLU gate, quick-GELU, gradient accumulation, the data-gradient of the
//                             stride-2 and the nearest-upsample convolutions, |x|.mean() gradient, concept-row gather)
//
// Data gradients of every Linear / 3x3 convolution are the forward MFMA GEMM (pv_gemm_conv) on transposed / tap-flipped weights.
#include "pv_common.h"

// pv_attnbwd.hip: the 8-wave staggered form of the two passes at d = 40
__attribute__((visibility("hidden"))) bool pv_attn8_bwd_eligible(const pv_attn_bwd_params& p);
__attribute__((visibility("hidden"))) int pv_attn8_bwd_launch(const pv_attn_bwd_params& p, hipStream_t s);

namespace {

__device__ __forceinline__ half8_t tz8() { return half8_t{0, 0, 0, 0, 0, 0, 0, 0}; }

template <int D>
struct BCfg {
    static constexpr int DK = (D + 31) / 32 * 32;   // contraction length over the head dimension (zero padded)
    static constexpr int KSTEPS = DK / 32;
    static constexpr int DT = (D + 15) / 16;        // 16-row fragments of the d-major outputs
    static constexpr int RS = DK + 8;               // row stride (halfs) of the row-major tiles
    static constexpr int CH = D / 8;                // 16-byte chunks per row
    static constexpr int TILE = 64 * RS;            // halfs of one row-major tile
};

// 64 rows of a [rows][ld] fp16 matrix (columns [0, D) of one head) -> registers (gload), registers -> a row-major LDS tile with zero
// padding beyond D / beyond the last row (swrite).  Split so the next tile's global loads are in flight while the current tile computes.
template <int D, int NT = 256>
struct TileRegs {
    static constexpr int CHP = BCfg<D>::DK / 8;
    static constexpr int NI = (64 * CHP + NT - 1) / NT;
    half8_t v[NI];
    // branch-free: always load from a clamped (valid) address, select zero afterwards - per-item exec-mask branches in the tile loop cost
    // more than the wasted loads of the pad chunks
    __device__ __forceinline__ void gload(const half_t* g, int ld, int row0, int nrows) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int r = idx / CHP, c = idx - r * CHP;
            const bool ok = c < BCfg<D>::CH && row0 + r < nrows && (((64 * CHP) % NT == 0) || idx < 64 * CHP);
            const int rr = min(row0 + r, nrows - 1), cc = c < BCfg<D>::CH ? c : 0;
            const half8_t x = *reinterpret_cast<const half8_t*>(g + (size_t)rr * ld + cc * 8);
            v[i] = ok ? x : tz8();
        }
    }
    __device__ __forceinline__ void swrite(half_t* sR, float mul) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int r = idx / CHP, c = idx - r * CHP;
            if (((64 * CHP) % NT == 0) || idx < 64 * CHP) {
                half8_t x = v[i];
                if (mul != 1.0f) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = (half_t)((float)x[j] * mul);
                }
                *reinterpret_cast<half8_t*>(sR + r * BCfg<D>::RS + c * 8) = x;
            }
        }
    }
};

// MFMA-A fragment of the TRANSPOSE of a row-major tile, read with the gfx950 transposing LDS load (ds_read_b64_tr_b16, as the
// forward kernel's V^T operand): rows d = f*16 + fr, the 8 contraction slots {s2*32 + fq*4 + 0..3, s2*32 + 16 + fq*4 + 0..3}
template <int D>
__device__ __forceinline__ half8_t tfrag(const half_t* sR, int f, int s2, int fr, int fq) {
    using C = BCfg<D>;
    const half_t* a = sR + (s2 * 32 + fq * 4 + (fr >> 2)) * C::RS + f * 16 + (fr & 3) * 4;
    const fp16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a));
    const fp16x4_t t2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a + 16 * C::RS));
    const half4_t h1 = __builtin_bit_cast(half4_t, t1), h2 = __builtin_bit_cast(half4_t, t2);      // same bits: no per-element conversion
    return __builtin_shufflevector(h1, h2, 0, 1, 2, 3, 4, 5, 6, 7);
}

// delta[b][h][q] = sum_c dO[q][c] * O[q][c]; also qs = fp16(q * softmax_scale * log2 e) - the forward kernel's rounding of the scaled
// query, written once so that the dK / dV kernel stages it without a per-tile multiply
__global__ void attn_bwd_delta_kernel(const pv_attn_bwd_params p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.batch * p.heads * p.nq;
    if (idx >= total) return;
    const int q = (int)(idx % p.nq), h = (int)((idx / p.nq) % p.heads), b = (int)(idx / ((long)p.nq * p.heads));
    const half_t* o = reinterpret_cast<const half_t*>(p.out) + ((size_t)b * p.nq + q) * p.ldo + h * p.d;
    const half_t* g = reinterpret_cast<const half_t*>(p.dout) + ((size_t)b * p.nq + q) * p.lddo + h * p.d;
    const half_t* qr = reinterpret_cast<const half_t*>(p.q) + ((size_t)b * p.nq + q) * p.ldq + h * p.d;
    half_t* qs = reinterpret_cast<half_t*>(p.qs) + ((size_t)b * p.nq + q) * p.ldqs + h * p.d;
    const float qscale = rsqrtf((float)p.d) * 1.4426950408889634f;
    float a = 0.f;
    for (int c = 0; c < p.d; c += 8) {
        const half8_t x = *reinterpret_cast<const half8_t*>(o + c), y = *reinterpret_cast<const half8_t*>(g + c);
        const half8_t qv = *reinterpret_cast<const half8_t*>(qr + c);
        half8_t sv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a += (float)x[j] * (float)y[j];
            sv[j] = (half_t)((float)qv[j] * qscale);
        }
        *reinterpret_cast<half8_t*>(qs + c) = sv;
    }
    p.delta[idx] = a;
}

// dQ: one workgroup = 64 * NF queries of one (sample, head) (wave w: NF fragments of 16 queries, each the MFMA column of its own
// products), keys walked in tiles of 64.  The K / V / K^T operand fragments read from LDS are shared by the NF query fragments - with one
// fragment per wave the kernels are LDS-read bound (1 MFMA per KB read), two halve that.
//   S'^T = K (q qscale)^T - lse  (log2 units)     P^T = exp2(S'^T)        dP^T = V dO^T        dS^T = P^T (dP^T - delta)
//   dQ^T += K^T dS^T   (contraction over the 64 keys, order permuted identically on both operands)
template <int D, int NF>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const pv_attn_bwd_params p) {
    using C = BCfg<D>;
    constexpr int QW = 64 * NF;                              // queries per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sK = reinterpret_cast<half_t*>(smem);
    half_t* sV = sK + C::TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    const int nqt = (p.nq + QW - 1) / QW;
    // XCD-aware: the query tiles of one (sample, head) get consecutive remapped ids = one XCD, whose L2 then holds that head's K / V once
    // (spread round-robin, every XCD streams every active head: the working set of ~16 heads does not fit a 4 MB L2)
    const int rid = pv_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int qt = rid % nqt, h = (rid / nqt) % p.heads, b = rid / (nqt * p.heads);
    const half_t* Q = reinterpret_cast<const half_t*>(p.q) + (size_t)b * p.nq * p.ldq + h * D;
    const half_t* DO = reinterpret_cast<const half_t*>(p.dout) + (size_t)b * p.nq * p.lddo + h * D;
    const half_t* Kg = reinterpret_cast<const half_t*>(p.k) + (size_t)b * p.nk * p.ldk + h * D;
    const half_t* Vg = reinterpret_cast<const half_t*>(p.v) + (size_t)b * p.nk * p.ldv + h * D;
    const float scale = rsqrtf((float)D), qscale = scale * 1.4426950408889634f;
    const size_t bh = ((size_t)b * p.heads + h) * p.nq;

    int qrow[NF];
    bool qok[NF];
    half8_t qf[NF][C::KSTEPS], dof[NF][C::KSTEPS];
    float nlse[NF], delta[NF];
    float4_t acc[NF][C::DT];
#pragma unroll
    for (int qi = 0; qi < NF; ++qi) {
        qrow[qi] = qt * QW + (wave * NF + qi) * 16 + fr;
        qok[qi] = qrow[qi] < p.nq;
        const int qc = qok[qi] ? qrow[qi] : p.nq - 1;
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ++ks) {
            const int c = ks * 4 + fq;
            qf[qi][ks] = c < C::CH ? *reinterpret_cast<const half8_t*>(Q + (size_t)qc * p.ldq + c * 8) : tz8();
            dof[qi][ks] = (c < C::CH && qok[qi]) ? *reinterpret_cast<const half8_t*>(DO + (size_t)qc * p.lddo + c * 8) : tz8();
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qi][ks][j] = (half_t)((float)qf[qi][ks][j] * qscale);    // same fp16 rounding as the forward kernel
        }
        nlse[qi] = qok[qi] ? -p.lse[bh + qc] : -INFINITY;
        delta[qi] = qok[qi] ? -p.delta[bh + qc] : 0.f;     // -delta: the initial value of the dP accumulators (dS = P (dP - delta) needs no subtraction)
#pragma unroll
        for (int f = 0; f < C::DT; ++f) acc[qi][f] = float4_t{0.f, 0.f, 0.f, 0.f};
    }

    int ntiles = (p.nk + 63) / 64;
    if (p.causal) ntiles = min(ntiles, min(qt * QW + QW - 1, p.nq - 1) / 64 + 1);
    TileRegs<D> rk, rv;
    rk.gload(Kg, p.ldk, 0, p.nk);
    rv.gload(Vg, p.ldv, 0, p.nk);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        rk.swrite(sK, 1.0f);
        rv.swrite(sV, 1.0f);
        __syncthreads();
        if (t + 1 < ntiles) {
            rk.gload(Kg, p.ldk, (t + 1) * 64, p.nk);
            rv.gload(Vg, p.ldv, (t + 1) * 64, p.nk);
        }
        // the tile body exists twice: the unmasked copy (every tile but a ragged last one, nothing causal) is one straight-line block the
        // scheduler can interleave freely; with a run-time flag inside it every score carried its own branch around the mask
        auto tile_body = [&](auto mask_tag) {
        constexpr bool masked = decltype(mask_tag)::value;
        float4_t ds[NF][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            half8_t ka[C::KSTEPS], va[C::KSTEPS];
#pragma unroll
            for (int ks = 0; ks < C::KSTEPS; ++ks) {
                ka[ks] = *reinterpret_cast<const half8_t*>(sK + (kb * 16 + fr) * C::RS + (ks * 4 + fq) * 8);
                va[ks] = *reinterpret_cast<const half8_t*>(sV + (kb * 16 + fr) * C::RS + (ks * 4 + fq) * 8);
            }
#pragma unroll
            for (int qi = 0; qi < NF; ++qi) {
                float4_t s = float4_t{nlse[qi], nlse[qi], nlse[qi], nlse[qi]}, dp = float4_t{delta[qi], delta[qi], delta[qi], delta[qi]};
#pragma unroll
                for (int ks = 0; ks < C::KSTEPS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka[ks], qf[qi][ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(va[ks], dof[qi][ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = PV_EXP2(s[r]);
                    if (masked) {
                        const int key = t * 64 + kb * 16 + fq * 4 + r;
                        if (key >= p.nk || (p.causal && key > qrow[qi])) pr = 0.f;
                    }
                    ds[qi][kb][r] = pr * dp[r];
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            half8_t bsl[NF];
#pragma unroll
            for (int qi = 0; qi < NF; ++qi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bsl[qi][r] = (half_t)ds[qi][2 * s2][r];
                    bsl[qi][r + 4] = (half_t)ds[qi][2 * s2 + 1][r];
                }
#pragma unroll
            for (int f = 0; f < C::DT; ++f) {
                const half8_t tk = tfrag<D>(sK, f, s2, fr, fq);
#pragma unroll
                for (int qi = 0; qi < NF; ++qi) acc[qi][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk, bsl[qi], acc[qi][f], 0, 0, 0);
            }
        }
        };
        if (p.causal || (t + 1) * 64 > p.nk) tile_body(std::true_type{});
        else tile_body(std::false_type{});
    }
#pragma unroll
    for (int qi = 0; qi < NF; ++qi)
        if (qok[qi]) {
            half_t* dQ = reinterpret_cast<half_t*>(p.dq) + ((size_t)b * p.nq + qrow[qi]) * p.lddq + h * D;
#pragma unroll
            for (int f = 0; f < C::DT; ++f) {
                const int dv = f * 16 + fq * 4;
                if (dv < D) {
                    half4_t o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[qi][f][r] * scale);
                    *reinterpret_cast<half4_t*>(dQ + dv) = o;
                }
            }
        }
}

// dK, dV: one workgroup = 64 * NF keys of one (sample, head) (wave w: NF fragments of 16 keys as MFMA columns), queries walked in tiles
// of 64; the Q / dO / Q^T / dO^T fragments read from LDS are shared by the NF key fragments.
//   S' = (q qscale) K^T - lse      P = exp2(S')      dP = dO V^T      dS = P (dP - delta)
//   dV^T += dO^T P      dK^T += (q qscale)^T dS / log2(e)
template <int D, int NF>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const pv_attn_bwd_params p) {
    using C = BCfg<D>;
    constexpr int KW = 64 * NF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sQ = reinterpret_cast<half_t*>(smem);
    half_t* sDO = sQ + C::TILE;
    float* sL = reinterpret_cast<float*>(sDO + C::TILE);       // -lse of the 64 staged queries (-inf beyond nq)
    float* sDl = sL + 64;                                      // -delta of the same queries
    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    const int nkt = (p.nk + KW - 1) / KW;
    const int rid = pv_xcd_remap((int)blockIdx.x, (int)gridDim.x);      // one (sample, head) per XCD at a time: its Q / dO stay in that L2
    const int kt = rid % nkt, h = (rid / nkt) % p.heads, b = rid / (nkt * p.heads);
    const half_t* Q = reinterpret_cast<const half_t*>(p.qs) + (size_t)b * p.nq * p.ldqs + h * D;       // pre-scaled queries
    const half_t* DO = reinterpret_cast<const half_t*>(p.dout) + (size_t)b * p.nq * p.lddo + h * D;
    const half_t* Kg = reinterpret_cast<const half_t*>(p.k) + (size_t)b * p.nk * p.ldk + h * D;
    const half_t* Vg = reinterpret_cast<const half_t*>(p.v) + (size_t)b * p.nk * p.ldv + h * D;
    const size_t bh = ((size_t)b * p.heads + h) * p.nq;
    const bool wg_masked = p.causal || (kt + 1) * KW > p.nk;   // workgroup-uniform: some key of this workgroup needs the mask

    int key[NF];
    bool kok[NF];
    half8_t kf[NF][C::KSTEPS], vf[NF][C::KSTEPS];
    float4_t accK[NF][C::DT], accV[NF][C::DT];
#pragma unroll
    for (int ki = 0; ki < NF; ++ki) {
        key[ki] = kt * KW + (wave * NF + ki) * 16 + fr;
        kok[ki] = key[ki] < p.nk;
        const int kc = kok[ki] ? key[ki] : p.nk - 1;
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ++ks) {
            const int c = ks * 4 + fq;
            kf[ki][ks] = (c < C::CH && kok[ki]) ? *reinterpret_cast<const half8_t*>(Kg + (size_t)kc * p.ldk + c * 8) : tz8();
            vf[ki][ks] = (c < C::CH && kok[ki]) ? *reinterpret_cast<const half8_t*>(Vg + (size_t)kc * p.ldv + c * 8) : tz8();
        }
#pragma unroll
        for (int f = 0; f < C::DT; ++f) accK[ki][f] = accV[ki][f] = float4_t{0.f, 0.f, 0.f, 0.f};
    }

    const int nqt = (p.nq + 63) / 64;
    const int t0 = p.causal ? (kt * KW) / 64 : 0;          // causal: queries before this workgroup's first key never see it
    TileRegs<D> rq, rdo;
    float pl = -INFINITY, pd = 0.f;
    auto gload_stats = [&](int t) {                        // every thread loads (clamped address): no exec branch in the loop
        const int qr = t * 64 + (tid & 63), qc = min(qr, p.nq - 1);
        const float l = p.lse[bh + qc], dd = p.delta[bh + qc];
        pl = qr < p.nq ? -l : -INFINITY;
        pd = qr < p.nq ? -dd : 0.f;                        // staged negated: the dP accumulators start from -delta
    };
    if (t0 < nqt) {
        rq.gload(Q, p.ldqs, t0 * 64, p.nq);
        rdo.gload(DO, p.lddo, t0 * 64, p.nq);
        gload_stats(t0);
    }
    for (int t = t0; t < nqt; ++t) {
        __syncthreads();
        rq.swrite(sQ, 1.0f);
        rdo.swrite(sDO, 1.0f);
        if (tid < 64) {
            sL[tid] = pl;
            sDl[tid] = pd;
        }
        __syncthreads();
        if (t + 1 < nqt) {
            rq.gload(Q, p.ldqs, (t + 1) * 64, p.nq);
            rdo.gload(DO, p.lddo, (t + 1) * 64, p.nq);
            gload_stats(t + 1);
        }
        auto tile_body = [&](auto mask_tag) {                   // two copies, see the dQ kernel
        constexpr bool masked = decltype(mask_tag)::value;
        float4_t pw[NF][4], ds[NF][4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const float4_t l4 = *reinterpret_cast<const float4_t*>(sL + qb * 16 + fq * 4);
            const float4_t dl = *reinterpret_cast<const float4_t*>(sDl + qb * 16 + fq * 4);
            half8_t qa[C::KSTEPS], da[C::KSTEPS];
#pragma unroll
            for (int ks = 0; ks < C::KSTEPS; ++ks) {
                qa[ks] = *reinterpret_cast<const half8_t*>(sQ + (qb * 16 + fr) * C::RS + (ks * 4 + fq) * 8);
                da[ks] = *reinterpret_cast<const half8_t*>(sDO + (qb * 16 + fr) * C::RS + (ks * 4 + fq) * 8);
            }
#pragma unroll
            for (int ki = 0; ki < NF; ++ki) {
                float4_t s = l4, dp = dl;
#pragma unroll
                for (int ks = 0; ks < C::KSTEPS; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[ks], kf[ki][ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(da[ks], vf[ki][ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = PV_EXP2(s[r]);
                    if (masked) {
                        const int qr = t * 64 + qb * 16 + fq * 4 + r;
                        if (!kok[ki] || (p.causal && key[ki] > qr)) pr = 0.f;
                    }
                    pw[ki][qb][r] = pr;
                    ds[ki][qb][r] = pr * dp[r];
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            half8_t bp[NF], bs[NF];
#pragma unroll
            for (int ki = 0; ki < NF; ++ki)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bp[ki][r] = (half_t)pw[ki][2 * s2][r];
                    bp[ki][r + 4] = (half_t)pw[ki][2 * s2 + 1][r];
                    bs[ki][r] = (half_t)ds[ki][2 * s2][r];
                    bs[ki][r + 4] = (half_t)ds[ki][2 * s2 + 1][r];
                }
#pragma unroll
            for (int f = 0; f < C::DT; ++f) {
                const half8_t td = tfrag<D>(sDO, f, s2, fr, fq), tq = tfrag<D>(sQ, f, s2, fr, fq);
#pragma unroll
                for (int ki = 0; ki < NF; ++ki) {
                    accV[ki][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(td, bp[ki], accV[ki][f], 0, 0, 0);
                    accK[ki][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tq, bs[ki], accK[ki][f], 0, 0, 0);
                }
            }
        }
        };
        if (wg_masked) tile_body(std::true_type{});
        else tile_body(std::false_type{});
    }
#pragma unroll
    for (int ki = 0; ki < NF; ++ki)
        if (kok[ki]) {
            half_t* dK = reinterpret_cast<half_t*>(p.dk) + ((size_t)b * p.nk + key[ki]) * p.lddk + h * D;
            half_t* dV = reinterpret_cast<half_t*>(p.dv) + ((size_t)b * p.nk + key[ki]) * p.lddv + h * D;
#pragma unroll
            for (int f = 0; f < C::DT; ++f) {
                const int dv = f * 16 + fq * 4;
                if (dv < D) {
                    half4_t ok_, ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ok_[r] = (half_t)(accK[ki][f][r] * 0.6931471805599453f);
                        ov[r] = (half_t)accV[ki][f][r];
                    }
                    *reinterpret_cast<half4_t*>(dK + dv) = ok_;
                    *reinterpret_cast<half4_t*>(dV + dv) = ov;
                }
            }
        }
}

template <int D>
int launch_attn_bwd(const pv_attn_bwd_params& p, hipStream_t s) {
    using C = BCfg<D>;
    // fragments per wave: 2 where the accumulators fit (d <= 80) and the sequence is long enough to keep >= 2 workgroups per CU busy
    constexpr int NFMAX = D <= 80 ? 2 : 1;
    constexpr int smem_dq = 2 * C::TILE * 2;
    constexpr int smem_dkv = 2 * C::TILE * 2 + 128 * 4;
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    if (!attr_set) {
        const void* fns[4] = {reinterpret_cast<const void*>(attn_bwd_dq_kernel<D, 1>), reinterpret_cast<const void*>(attn_bwd_dq_kernel<D, NFMAX>),
                              reinterpret_cast<const void*>(attn_bwd_dkv_kernel<D, 1>), reinterpret_cast<const void*>(attn_bwd_dkv_kernel<D, NFMAX>)};
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, i < 2 ? smem_dq : smem_dkv);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    const long rows = (long)p.batch * p.heads * p.nq;
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, p);
    const long bh = (long)p.batch * p.heads;
    static const int nf_env = getenv("PV_ATTN_BWD_NF") ? atoi(getenv("PV_ATTN_BWD_NF")) : -1;      // experiments: bit 0 = dK/dV, bit 1 = dQ
    const bool big = NFMAX == 2 && bh * ((p.nk + 127) / 128) >= 1024 && bh * ((p.nq + 127) / 128) >= 1024;
    // d = 80 (N = 1024, B = 16): two fragments in the dQ pass only 343 -> 320 us, in both 327, in the dK/dV pass only 350 (profiles/r05_attn_bwd_d80_nf.txt)
    const bool two_kv = big && (nf_env < 0 ? D == 40 : (nf_env & 1)), two_q = big && (nf_env < 0 ? (D == 40 || D == 80) : (nf_env & 2));
    if (two_kv) hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, NFMAX>), dim3((unsigned)(((p.nk + 127) / 128) * bh)), dim3(256), smem_dkv, s, p);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, 1>), dim3((unsigned)(((p.nk + 63) / 64) * bh)), dim3(256), smem_dkv, s, p);
    if (two_q) hipLaunchKernelGGL((attn_bwd_dq_kernel<D, NFMAX>), dim3((unsigned)(((p.nq + 127) / 128) * bh)), dim3(256), smem_dq, s, p);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<D, 1>), dim3((unsigned)(((p.nq + 63) / 64) * bh)), dim3(256), smem_dq, s, p);
    return PV_CHECK_LAUNCH();
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Dual-branch cross attention backward (PhotoVerseAttnProcessor2_0, attention_processor.py:317-322 text branch, :392-420 image-token
// branch + fusion): O = w_t softmax(S_t) V_t + w_i softmax(S_i) V_i over the 96-row K / V image (text rows [0, nt), image-token rows
// [80, 80 + nip)).  Same MFMA scheme as the self-attention backward; the key set is one tile, so
//   pass 1 (xattn_bwd_dq): 64 queries per workgroup - both softmaxes in registers, dQ, and the per-(query, branch) statistics
//                          (lse, delta = sum_j P_j dP_j) for pass 2
//   pass 2 (xattn_bwd_dkv): 6 waves = the 96 keys, one workgroup per chunk of 512 queries -> fp32 partial dK / dV, summed in chunk
//                          order by xattn_bwd_reduce (deterministic), which also adds the to_v_ip_norm regulariser gradient.
constexpr int XB_KEYS = 96;
constexpr int XB_IP0 = 80;
constexpr int XB_QCH = 512;     // queries per pass-2 workgroup

// stage the 96-row K (or V) image of (b, h) from its two sources into a row-major LDS tile
template <int D, int NT>
__device__ __forceinline__ void xb_stage_kv(const pv_xattn_bwd_params& p, int b, int h, bool want_v, half_t* sR) {
    using C = BCfg<D>;
    constexpr int CHP = C::DK / 8;
    const half_t* t = reinterpret_cast<const half_t*>(want_v ? p.vt : p.kt);
    const half_t* ipp = reinterpret_cast<const half_t*>(want_v ? p.vip : p.kip);
    const int ldt = want_v ? p.ldvt : p.ldkt, ldi = want_v ? p.ldvip : p.ldkip;
    for (int i = threadIdx.x; i < XB_KEYS * CHP; i += NT) {
        const int j = i / CHP, c = i - j * CHP;
        half8_t v = tz8();
        if (c < C::CH) {
            if (j < p.nt) v = *reinterpret_cast<const half8_t*>(t + ((size_t)b * p.nt + j) * ldt + h * D + c * 8);
            else if (j >= XB_IP0 && j < XB_IP0 + p.nip) v = *reinterpret_cast<const half8_t*>(ipp + ((size_t)b * p.nip + j - XB_IP0) * ldi + h * D + c * 8);
        }
        *reinterpret_cast<half8_t*>(sR + j * C::RS + c * 8) = v;
    }
}

template <int D>
__global__ __launch_bounds__(256) void xattn_bwd_dq_kernel(const pv_xattn_bwd_params p) {
    using C = BCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sK = reinterpret_cast<half_t*>(smem);
    half_t* sV = sK + XB_KEYS * C::RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    const int nqt = (p.nq + 63) / 64;
    const int qt = blockIdx.x % nqt, h = (blockIdx.x / nqt) % p.heads, b = blockIdx.x / (nqt * p.heads);
    const float w_text = p.fusion ? p.fusion[0] : p.w_text, w_ip = p.fusion ? p.fusion[1] : p.w_ip;
    const float scale = rsqrtf((float)D), qscale = scale * 1.4426950408889634f;
    xb_stage_kv<D, 256>(p, b, h, false, sK);
    xb_stage_kv<D, 256>(p, b, h, true, sV);

    const half_t* Q = reinterpret_cast<const half_t*>(p.q) + (size_t)b * p.nq * p.ldq + h * D;
    const half_t* DO = reinterpret_cast<const half_t*>(p.dout) + (size_t)b * p.nq * p.lddo + h * D;
    const int qrow = qt * 64 + wave * 16 + fr;
    const bool qok = qrow < p.nq;
    const int qc = qok ? qrow : p.nq - 1;
    half8_t qf[C::KSTEPS], dof[C::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
        const int c = ks * 4 + fq;
        qf[ks] = c < C::CH ? *reinterpret_cast<const half8_t*>(Q + (size_t)qc * p.ldq + c * 8) : tz8();
        dof[ks] = (c < C::CH && qok) ? *reinterpret_cast<const half8_t*>(DO + (size_t)qc * p.lddo + c * 8) : tz8();
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (half_t)((float)qf[ks][j] * qscale);
    }
    __syncthreads();

    float4_t s[6], dp[6];
#pragma unroll
    for (int kb = 0; kb < 6; ++kb) {
        s[kb] = dp[kb] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ++ks) {
            const half8_t ka = *reinterpret_cast<const half8_t*>(sK + (kb * 16 + fr) * C::RS + (ks * 4 + fq) * 8);
            const half8_t va = *reinterpret_cast<const half8_t*>(sV + (kb * 16 + fr) * C::RS + (ks * 4 + fq) * 8);
            s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qf[ks], s[kb], 0, 0, 0);
            dp[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, dof[ks], dp[kb], 0, 0, 0);
        }
    }
    // key (kb, r) = kb*16 + fq*4 + r: subtiles 0..4 are text rows (valid below nt), subtile 5 the image-token rows (valid below nip)
    float mt = -INFINITY, mi = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 6; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kb * 16 + fq * 4 + r;
            if (kb < 5) { if (key < p.nt) mt = fmaxf(mt, s[kb][r]); }
            else if (key - XB_IP0 < p.nip) mi = fmaxf(mi, s[kb][r]);
        }
    mt = pv_quad_max(mt);
    mi = pv_quad_max(mi);
    float lt = 0.f, li = 0.f;
#pragma unroll
    for (int kb = 0; kb < 6; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kb * 16 + fq * 4 + r;
            float e = 0.f;
            if (kb < 5) { if (key < p.nt) { e = PV_EXP2(s[kb][r] - mt); lt += e; } }
            else if (key - XB_IP0 < p.nip) { e = PV_EXP2(s[kb][r] - mi); li += e; }
            s[kb][r] = e;
        }
    lt = pv_quad_sum(lt);
    li = pv_quad_sum(li);
    const float it = lt > 0.f ? 1.0f / lt : 0.f, ii = li > 0.f ? 1.0f / li : 0.f;
    float dt = 0.f, di = 0.f;
#pragma unroll
    for (int kb = 0; kb < 6; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kb][r] *= kb < 5 ? it : ii;                       // P
            if (kb < 5) dt += s[kb][r] * dp[kb][r]; else di += s[kb][r] * dp[kb][r];
        }
    dt = pv_quad_sum(dt);
    di = pv_quad_sum(di);
    if (qok && fq == 0) {
        float4_t st = float4_t{mt + __log2f(lt), mi + __log2f(li), dt, di};
        *reinterpret_cast<float4_t*>(p.stats + (((size_t)b * p.heads + h) * p.nq + qrow) * 4) = st;
    }
    float4_t acc[C::DT];
#pragma unroll
    for (int f = 0; f < C::DT; ++f) acc[f] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
        half8_t bsl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k0 = 2 * s2, k1 = 2 * s2 + 1;
            bsl[r] = (half_t)((k0 < 5 ? w_text : w_ip) * s[k0][r] * (dp[k0][r] - (k0 < 5 ? dt : di)));
            bsl[r + 4] = (half_t)((k1 < 5 ? w_text : w_ip) * s[k1][r] * (dp[k1][r] - (k1 < 5 ? dt : di)));
        }
#pragma unroll
        for (int f = 0; f < C::DT; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tfrag<D>(sK, f, s2, fr, fq), bsl, acc[f], 0, 0, 0);
    }
    if (qok) {
        half_t* dQ = reinterpret_cast<half_t*>(p.dq) + ((size_t)b * p.nq + qrow) * p.lddq + h * D;
        const float os = scale * p.out_scale;
#pragma unroll
        for (int f = 0; f < C::DT; ++f) {
            const int dv = f * 16 + fq * 4;
            if (dv < D) {
                half4_t o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[f][r] * os);
                *reinterpret_cast<half4_t*>(dQ + dv) = o;
            }
        }
    }
}

template <int D>
__global__ __launch_bounds__(384) void xattn_bwd_dkv_kernel(const pv_xattn_bwd_params p, const int nchunk) {
    using C = BCfg<D>;
    constexpr int NT = 384;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sQ = reinterpret_cast<half_t*>(smem);
    half_t* sDO = sQ + C::TILE;
    float* sSt = reinterpret_cast<float*>(sDO + C::TILE);      // [64][4]: lse_t, lse_i, delta_t, delta_i of the staged queries
    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    const int chunk = blockIdx.x % nchunk, h = (blockIdx.x / nchunk) % p.heads, b = blockIdx.x / (nchunk * p.heads);
    const float qscale = rsqrtf((float)D) * 1.4426950408889634f;
    const bool is_ip = wave == 5;                              // wave-uniform branch: waves 0..4 own text rows, wave 5 the image-token rows
    const float wb = p.fusion ? p.fusion[is_ip ? 1 : 0] : (is_ip ? p.w_ip : p.w_text);
    const int key = wave * 16 + fr;
    const bool kok = is_ip ? (key - XB_IP0 < p.nip) : (key < p.nt);
    const half_t* Ksrc = is_ip ? reinterpret_cast<const half_t*>(p.kip) + ((size_t)b * p.nip + (kok ? key - XB_IP0 : 0)) * p.ldkip
                               : reinterpret_cast<const half_t*>(p.kt) + ((size_t)b * p.nt + (kok ? key : 0)) * p.ldkt;
    const half_t* Vsrc = is_ip ? reinterpret_cast<const half_t*>(p.vip) + ((size_t)b * p.nip + (kok ? key - XB_IP0 : 0)) * p.ldvip
                               : reinterpret_cast<const half_t*>(p.vt) + ((size_t)b * p.nt + (kok ? key : 0)) * p.ldvt;
    half8_t kf[C::KSTEPS], vf[C::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
        const int c = ks * 4 + fq;
        kf[ks] = (c < C::CH && kok) ? *reinterpret_cast<const half8_t*>(Ksrc + h * D + c * 8) : tz8();
        vf[ks] = (c < C::CH && kok) ? *reinterpret_cast<const half8_t*>(Vsrc + h * D + c * 8) : tz8();
    }
    const half_t* Q = reinterpret_cast<const half_t*>(p.q) + (size_t)b * p.nq * p.ldq + h * D;
    const half_t* DO = reinterpret_cast<const half_t*>(p.dout) + (size_t)b * p.nq * p.lddo + h * D;
    const float* stats = p.stats + ((size_t)b * p.heads + h) * p.nq * 4;
    float4_t accK[C::DT], accV[C::DT];
#pragma unroll
    for (int f = 0; f < C::DT; ++f) accK[f] = accV[f] = float4_t{0.f, 0.f, 0.f, 0.f};

    const int q0 = chunk * XB_QCH, q1 = min(q0 + XB_QCH, p.nq);
    const int nt_ = (q1 - q0 + 63) / 64;
    TileRegs<D, NT> rq, rdo;
    float4_t pst = float4_t{INFINITY, INFINITY, 0.f, 0.f};
    auto gstats = [&](int t) {
        if (tid < 64) {
            const int qr = q0 + t * 64 + tid;
            pst = qr < q1 ? *reinterpret_cast<const float4_t*>(stats + (size_t)qr * 4) : float4_t{INFINITY, INFINITY, 0.f, 0.f};
        }
    };
    rq.gload(Q, p.ldq, q0, q1);
    rdo.gload(DO, p.lddo, q0, q1);
    gstats(0);
    for (int t = 0; t < nt_; ++t) {
        __syncthreads();
        rq.swrite(sQ, qscale);
        rdo.swrite(sDO, 1.0f);
        if (tid < 64) *reinterpret_cast<float4_t*>(sSt + tid * 4) = pst;
        __syncthreads();
        if (t + 1 < nt_) {
            rq.gload(Q, p.ldq, q0 + (t + 1) * 64, q1);
            rdo.gload(DO, p.lddo, q0 + (t + 1) * 64, q1);
            gstats(t + 1);
        }
        float4_t pw[4], ds[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            float4_t sc, dl, dpv = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* st = sSt + (qb * 16 + fq * 4 + r) * 4;
                sc[r] = -st[is_ip ? 1 : 0];
                dl[r] = st[is_ip ? 3 : 2];
            }
#pragma unroll
            for (int ks = 0; ks < C::KSTEPS; ++ks) {
                const half8_t qa = *reinterpret_cast<const half8_t*>(sQ + (qb * 16 + fr) * C::RS + (ks * 4 + fq) * 8);
                const half8_t da = *reinterpret_cast<const half8_t*>(sDO + (qb * 16 + fr) * C::RS + (ks * 4 + fq) * 8);
                sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, kf[ks], sc, 0, 0, 0);
                dpv = __builtin_amdgcn_mfma_f32_16x16x32_f16(da, vf[ks], dpv, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = kok ? wb * PV_EXP2(sc[r]) : 0.f;
                pw[qb][r] = pr;
                ds[qb][r] = pr * (dpv[r] - dl[r]);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            half8_t bp, bs;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bp[r] = (half_t)pw[2 * s2][r];
                bp[r + 4] = (half_t)pw[2 * s2 + 1][r];
                bs[r] = (half_t)ds[2 * s2][r];
                bs[r + 4] = (half_t)ds[2 * s2 + 1][r];
            }
#pragma unroll
            for (int f = 0; f < C::DT; ++f) {
                accV[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tfrag<D>(sDO, f, s2, fr, fq), bp, accV[f], 0, 0, 0);
                accK[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tfrag<D>(sQ, f, s2, fr, fq), bs, accK[f], 0, 0, 0);
            }
        }
    }
    float* part = p.partial + ((((size_t)b * p.heads + h) * nchunk + chunk) * 2) * XB_KEYS * D + (size_t)key * D;
#pragma unroll
    for (int f = 0; f < C::DT; ++f) {
        const int dv = f * 16 + fq * 4;
        if (dv < D) {
            float4_t k4, v4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                k4[r] = accK[f][r] * 0.6931471805599453f;
                v4[r] = accV[f][r];
            }
            *reinterpret_cast<float4_t*>(part + dv) = k4;
            *reinterpret_cast<float4_t*>(part + XB_KEYS * D + dv) = v4;
        }
    }
}

// sum the per-tile partials in tile order (deterministic) and scatter to the four gradient tensors (fp32 rows [B*nt | B*nip][C]);
// adds the to_v_ip_norm regulariser gradient vnorm_coef * v / ||v|| to dV_ip
template <int D>
__global__ void xattn_bwd_reduce_kernel(const pv_xattn_bwd_params p, const int ntile) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.batch * p.heads * XB_KEYS * D;
    if (idx >= total) return;
    const int c = (int)(idx % D), j = (int)((idx / D) % XB_KEYS), h = (int)((idx / ((long)D * XB_KEYS)) % p.heads), b = (int)(idx / ((long)D * XB_KEYS * p.heads));
    const bool is_t = j < p.nt, is_i = j >= XB_IP0 && j < XB_IP0 + p.nip;
    if (!is_t && !is_i) return;
    const float* src = p.partial + (((size_t)b * p.heads + h) * ntile * 2) * XB_KEYS * D + (size_t)j * D + c;
    float dk = 0.f, dv = 0.f;
    for (int t = 0; t < ntile; ++t) {
        dk += src[(size_t)t * 2 * XB_KEYS * D];
        dv += src[(size_t)t * 2 * XB_KEYS * D + XB_KEYS * D];
    }
    if (is_t) {
        p.dkt[((size_t)b * p.nt + j) * p.ld_dt + h * D + c] = dk * p.out_scale;
        p.dvt[((size_t)b * p.nt + j) * p.ld_dt + h * D + c] = dv * p.out_scale;
    } else {
        const int pi = j - XB_IP0;
        if (p.vnorm_coef != 0.f || p.vnorm_grad) {
            const half_t* v = reinterpret_cast<const half_t*>(p.vip) + ((size_t)b * p.nip + pi) * p.ldvip + h * D;
            float n2 = 0.f;
            for (int cc = 0; cc < D; ++cc) n2 += (float)v[cc] * (float)v[cc];
            const float gnorm = p.vnorm_coef + (p.vnorm_grad ? p.vnorm_grad[((size_t)b * p.heads + h) * p.nip + pi] : 0.f);
            dv += gnorm * (float)v[c] * rsqrtf(fmaxf(n2, 1e-30f));
        }
        p.dkip[((size_t)b * p.nip + pi) * p.ld_di + h * D + c] = dk * p.out_scale;
        p.dvip[((size_t)b * p.nip + pi) * p.ld_di + h * D + c] = dv * p.out_scale;
    }
}

template <int D>
int launch_xattn_bwd(const pv_xattn_bwd_params& p, hipStream_t s) {
    using C = BCfg<D>;
    constexpr int smem_dq = 2 * XB_KEYS * C::RS * 2;
    constexpr int smem_dkv = 2 * C::TILE * 2 + 64 * 4 * 4;
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_bwd_dq_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_dq);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_bwd_dkv_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_dkv);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nqt = (p.nq + 63) / 64, nchunk = (p.nq + XB_QCH - 1) / XB_QCH;
    hipLaunchKernelGGL(xattn_bwd_dq_kernel<D>, dim3((unsigned)(nqt * p.heads * p.batch)), dim3(256), smem_dq, s, p);
    hipLaunchKernelGGL(xattn_bwd_dkv_kernel<D>, dim3((unsigned)(nchunk * p.heads * p.batch)), dim3(384), smem_dkv, s, p, nchunk);
    const long total = (long)p.batch * p.heads * XB_KEYS * D;
    hipLaunchKernelGGL(xattn_bwd_reduce_kernel<D>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, nchunk);
    return PV_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// GroupNorm (+ SiLU) backward.  y = act(z), z = gamma xhat + beta, xhat = (x - mean) rstd, statistics per (image, group):
//   dxhat = dy act'(z) gamma;  dx = rstd (dxhat - mean_g(dxhat) - xhat mean_g(dxhat xhat))
// Thread = one 8-channel chunk x one pixel lane; grid (pixel splits, images, 64-chunk blocks).
constexpr int GB_CPB = 64;    // most channel chunks a workgroup takes

struct GnbCoef {
    float a[8], bsh[8], g[8], rstd[8], m1[8], m2[8];
};

__device__ __forceinline__ half8_t gnb_load(const pv_groupnorm_bwd_params& p, size_t row, int chunk) {
    const int c = chunk * 8;
    if (c < p.c0) return *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.x0) + row * p.ld0 + c);
    return *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.x1) + row * p.ld1 + (c - p.c0));
}

__device__ __forceinline__ void gnb_coef(const pv_groupnorm_bwd_params& p, int b, int chunk, bool with_sums, GnbCoef& k) {
    const int C = p.c0 + p.c1, cpg = C / p.groups;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chunk * 8 + j, g = c / cpg;
        const float* st = p.stats + (size_t)b * p.stats_stride + g * 2;
        const float mean = st[0], rstd = st[1];
        k.rstd[j] = rstd;
        k.g[j] = p.gamma[c];
        k.a[j] = rstd;                    // xhat = x * a + bsh
        k.bsh[j] = -mean * rstd;
        if (with_sums) {
            k.m1[j] = p.sums[((size_t)b * p.groups + g) * 2];
            k.m2[j] = p.sums[((size_t)b * p.groups + g) * 2 + 1];
        }
    }
}

// dxhat for one element
__device__ __forceinline__ float gnb_dxhat(float xhat, float dy, float gamma, float beta, int act) {
    float dz = dy;
    if (act == PV_ACT_SILU) {
        const float z = xhat * gamma + beta;
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
        dz = dy * sg * (1.0f + z * (1.0f - sg));
    }
    return dz * gamma;
}

// cpb chunk lanes x rows pixel lanes = up to 256 threads: cpb = the block's share of the C / 8 chunks (<= 64), rows = 256 / cpb - at C = 320 that is 40 x 6
// (the fixed 64 x 4 layout left 96 of 256 threads idle there and 192 at the VAE decoder's C = 128); two pixels per thread and iteration in flight
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const pv_groupnorm_bwd_params p, const int cpb, const int rows) {
    __shared__ float red[256][16];
    const int C = p.c0 + p.c1, nchunk = C / 8;
    const int tid = threadIdx.x, cl = tid % cpb, r = tid / cpb;
    const int chunk = blockIdx.z * cpb + cl, b = blockIdx.y, split = blockIdx.x;
    const bool ok = chunk < nchunk && r < rows;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    if (ok) {
        GnbCoef k;
        gnb_coef(p, b, chunk, false, k);
        float beta[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) beta[j] = p.beta[chunk * 8 + j];
        const int per = (p.hw + p.splits - 1) / p.splits;
        const int px0 = split * per, px1 = min(px0 + per, p.hw);
        auto one = [&](const half8_t& xv, const half8_t& dv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (float)xv[j] * k.a[j] + k.bsh[j];
                const float dxh = gnb_dxhat(xh, (float)dv[j], k.g[j], beta[j], p.act);
                s1[j] += dxh;
                s2[j] += dxh * xh;
            }
        };
        const half_t* dyp = reinterpret_cast<const half_t*>(p.dy) + chunk * 8;
        int px = px0 + r;
        for (; px + rows < px1; px += 2 * rows) {
            const size_t row = (size_t)b * p.hw + px;
            const half8_t xa = gnb_load(p, row, chunk), xb = gnb_load(p, row + rows, chunk);
            const half8_t da = *reinterpret_cast<const half8_t*>(dyp + row * p.ld_dy), db = *reinterpret_cast<const half8_t*>(dyp + (row + rows) * p.ld_dy);
            one(xa, da);
            one(xb, db);
        }
        if (px < px1) {
            const size_t row = (size_t)b * p.hw + px;
            one(gnb_load(p, row, chunk), *reinterpret_cast<const half8_t*>(dyp + row * p.ld_dy));
        }
    }
    if (tid < cpb * rows) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[tid][j] = s1[j];
            red[tid][8 + j] = s2[j];
        }
    }
    __syncthreads();
    if (r == 0 && chunk < nchunk) {
        float* out = p.partial + (((size_t)b * p.splits + split) * 2) * C + chunk * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = 0.f, q = 0.f;
            for (int rr = 0; rr < rows; ++rr) {           // pixel lanes in order: deterministic
                a += red[rr * cpb + cl][j];
                q += red[rr * cpb + cl][8 + j];
            }
            out[j] = a;
            out[C + j] = q;
        }
    }
}

// one wave per (image, group): fixed-order sum over splits x channels of the group -> sums[b][g] = (mean dxhat, mean dxhat*xhat)
__global__ void gn_bwd_finalize_kernel(const pv_groupnorm_bwd_params p) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= p.batch * p.groups) return;
    const int b = idx / p.groups, g = idx - b * p.groups;
    const int C = p.c0 + p.c1, cpg = C / p.groups;
    float a = 0.f, q = 0.f;
    for (int i = lane; i < p.splits * cpg; i += 64) {
        const int s = i / cpg, c = g * cpg + (i - s * cpg);
        const float* src = p.partial + (((size_t)b * p.splits + s) * 2) * C + c;
        a += src[0];
        q += src[C];
    }
    a = pv_wave_sum(a);
    q = pv_wave_sum(q);
    if (lane == 0) {
        const float n = (float)cpg * (float)p.hw;
        p.sums[((size_t)b * p.groups + g) * 2] = a / n;
        p.sums[((size_t)b * p.groups + g) * 2 + 1] = q / n;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const pv_groupnorm_bwd_params p, const int px_per_block, const int cpb, const int rows) {
    const int C = p.c0 + p.c1, nchunk = C / 8;
    const int tid = threadIdx.x, cl = tid % cpb, r = tid / cpb;
    const int chunk = blockIdx.z * cpb + cl, b = blockIdx.y;
    if (chunk >= nchunk || r >= rows) return;
    GnbCoef k;
    gnb_coef(p, b, chunk, true, k);
    float beta[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) beta[j] = p.beta[chunk * 8 + j];
    const int c = chunk * 8;
    const bool first = c < p.c0;
    half_t* dx = first ? reinterpret_cast<half_t*>(p.dx0) : reinterpret_cast<half_t*>(p.dx1);
    const half_t* add = first ? reinterpret_cast<const half_t*>(p.add0) : reinterpret_cast<const half_t*>(p.add1);
    const int ldx = first ? p.ld_dx0 : p.ld_dx1, lda = first ? p.ld_add0 : p.ld_add1, cc = first ? c : c - p.c0;
    if (dx == nullptr) return;
    const int px0 = blockIdx.x * px_per_block, px1 = min(px0 + px_per_block, p.hw);
    const half_t* dyp = reinterpret_cast<const half_t*>(p.dy) + chunk * 8;
    auto one = [&](size_t row, const half8_t& xv, const half8_t& dv, const half8_t& av) {
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (float)xv[j] * k.a[j] + k.bsh[j];
            const float dxh = gnb_dxhat(xh, (float)dv[j], k.g[j], beta[j], p.act);
            o[j] = (half_t)(k.rstd[j] * (dxh - k.m1[j] - xh * k.m2[j]) + (float)av[j]);
        }
        *reinterpret_cast<half8_t*>(dx + row * ldx + cc) = o;
    };
    int px = px0 + r;
    for (; px + rows < px1; px += 2 * rows) {
        const size_t ra = (size_t)b * p.hw + px, rb = ra + rows;
        const half8_t xa = gnb_load(p, ra, chunk), xb = gnb_load(p, rb, chunk);
        const half8_t da = *reinterpret_cast<const half8_t*>(dyp + ra * p.ld_dy), db = *reinterpret_cast<const half8_t*>(dyp + rb * p.ld_dy);
        half8_t aa = tz8(), ab = tz8();
        if (add) {
            aa = *reinterpret_cast<const half8_t*>(add + ra * lda + cc);
            ab = *reinterpret_cast<const half8_t*>(add + rb * lda + cc);
        }
        one(ra, xa, da, aa);
        one(rb, xb, db, ab);
    }
    if (px < px1) {
        const size_t ra = (size_t)b * p.hw + px;
        one(ra, gnb_load(p, ra, chunk), *reinterpret_cast<const half8_t*>(dyp + ra * p.ld_dy), add ? *reinterpret_cast<const half8_t*>(add + ra * lda + cc) : tz8());
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// elementwise / layout pieces

// GEGLU: y = a * gelu(g), h = [a | g] (each n wide).  dh = [dy gelu(g) | dy a gelu'(g)], gelu'(g) = Phi(g) + g phi(g)
__global__ void geglu_bwd_kernel(const half_t* h, int ldh, const half_t* dy, int lddy, half_t* dh, int lddh, int rows, int n) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = n / 8;
    if (idx >= (long)rows * nch) return;
    const int r = (int)(idx / nch), c = (int)(idx % nch) * 8;
    const half8_t a = *reinterpret_cast<const half8_t*>(h + (size_t)r * ldh + c);
    const half8_t g = *reinterpret_cast<const half8_t*>(h + (size_t)r * ldh + n + c);
    const half8_t d = *reinterpret_cast<const half8_t*>(dy + (size_t)r * lddy + c);
    half8_t da, dg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gv = (float)g[j], dv = (float)d[j];
        const float cdf = 0.5f * (1.0f + pv_erf_fast(gv * 0.70710678118654752440f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * gv * gv);
        da[j] = (half_t)(dv * gv * cdf);
        dg[j] = (half_t)(dv * (float)a[j] * (cdf + gv * pdf));
    }
    *reinterpret_cast<half8_t*>(dh + (size_t)r * lddh + c) = da;
    *reinterpret_cast<half8_t*>(dh + (size_t)r * lddh + n + c) = dg;
}

// dx = dy * act'(x) for the pointwise activations applied to a saved pre-activation x
__global__ void act_bwd_kernel(const half_t* x, int ldx, const half_t* dy, int lddy, half_t* dx, int lddx, int rows, int cols, int act) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = cols / 8;
    if (idx >= (long)rows * nch) return;
    const int r = (int)(idx / nch), c = (int)(idx % nch) * 8;
    const half8_t xv = *reinterpret_cast<const half8_t*>(x + (size_t)r * ldx + c);
    const half8_t dv = *reinterpret_cast<const half8_t*>(dy + (size_t)r * lddy + c);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (float)xv[j];
        float g = 1.0f;
        if (act == PV_ACT_QUICK_GELU) {
            const float sg = 1.0f / (1.0f + __expf(-1.702f * v));
            g = sg * (1.0f + 1.702f * v * (1.0f - sg));
        } else if (act == PV_ACT_SILU) {
            const float sg = 1.0f / (1.0f + __expf(-v));
            g = sg * (1.0f + v * (1.0f - sg));
        } else if (act == PV_ACT_LEAKY_RELU) {
            g = v > 0.f ? 1.0f : 0.01f;
        } else if (act == PV_ACT_GELU) {
            g = 0.5f * (1.0f + pv_erf_fast(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
        }
        o[j] = (half_t)((float)dv[j] * g);
    }
    *reinterpret_cast<half8_t*>(dx + (size_t)r * lddx + c) = o;
}

// y = act(x) as its own pass (training keeps the pre-activation; inference fuses the activation into the GEMM epilogue)
__global__ void act_fwd_kernel(const half_t* x, int ldx, half_t* y, int ldy, int rows, int cols, int act) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = cols / 8;
    if (idx >= (long)rows * nch) return;
    const int r = (int)(idx / nch), c = (int)(idx % nch) * 8;
    const half8_t xv = *reinterpret_cast<const half8_t*>(x + (size_t)r * ldx + c);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)pv_apply_act((float)xv[j], act);
    *reinterpret_cast<half8_t*>(y + (size_t)r * ldy + c) = o;
}

// Philox4x32-10 (Salmon et al. 2011), all four output words
__device__ __forceinline__ void philox4(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Dropout on the input of a LoRA branch (peft: result += B(A(dropout(x))) * scaling; train.py:265 lora_dropout).  Counter-based: the
// keep mask of element (row, col) of copy k is a pure function of (key, site, iteration, k, row, col) - the backward launch recomputes
// it, nothing is stored.  rng = {key lo, key hi, iteration counter} (the block pv_fusion_draw advances once per forward).
//   forward  (bwd = 0): out[r][k * cols + c] = x[r][c] * keep_k / (1 - p)                        k < copies
//   backward (bwd = 1): out[r][c] = sum_k x[r][k * cols + c] * keep_k / (1 - p) (+ add[r][c])
__global__ void dropout_kernel(const half_t* x, int ldx, half_t* out, int ldo, const half_t* add, int ldadd, int rows, int cols, int copies, float p,
                               const uint32_t* rng, uint32_t site, int bwd) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = cols / 8;
    if (idx >= (long)rows * nch) return;
    const int r = (int)(idx / nch), c = (int)(idx % nch);
    const uint32_t thr = (uint32_t)(p * 65536.0f);
    const float inv = 1.0f / (1.0f - p);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (bwd && add) {
        const half8_t a = *reinterpret_cast<const half8_t*>(add + (size_t)r * ldadd + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (float)a[j];
    }
    half8_t xin = tz8();
    if (!bwd) xin = *reinterpret_cast<const half8_t*>(x + (size_t)r * ldx + c * 8);
    for (int k = 0; k < copies; ++k) {
        uint32_t w[4];
        philox4(rng[0], rng[1], (uint32_t)idx, site, rng[2], (uint32_t)k, w);
        if (bwd) xin = *reinterpret_cast<const half8_t*>(x + (size_t)r * ldx + k * cols + c * 8);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool keep = ((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) >= thr;
            const float v = keep ? (float)xin[j] * inv : 0.f;
            if (bwd) acc[j] += v; else o[j] = (half_t)v;
        }
        if (!bwd) *reinterpret_cast<half8_t*>(out + (size_t)r * ldo + k * cols + c * 8) = o;
    }
    if (bwd) {
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
        *reinterpret_cast<half8_t*>(out + (size_t)r * ldo + c * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Pieces of the ArcFace identity loss (models/loss.py:26-78, models/arcface_resnet.py:12-133) and of the VAE-decoder backward it needs

// y[r][c] = x[r][c] * scale[c] + shift[c]: eval-mode BatchNorm that cannot be folded into a convolution (IRBlock.bn0 / bn4 sit in FRONT of a
// zero-padded conv); its backward is the same launch with shift = NULL
__global__ void col_affine_kernel(const half_t* x, int ldx, const float* scale, const float* shift, half_t* y, int ldy, int rows, int cols) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = cols / 8;
    if (idx >= (long)rows * nch) return;
    const int r = (int)(idx / nch), c = (int)(idx % nch) * 8;
    const half8_t v = *reinterpret_cast<const half8_t*>(x + (size_t)r * ldx + c);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)v[j] * scale[c + j] + (shift ? shift[c + j] : 0.f));
    *reinterpret_cast<half8_t*>(y + (size_t)r * ldy + c) = o;
}

// PReLU with ONE learned slope (nn.PReLU(), arcface_resnet.py:21,73): forward y = x > 0 ? x : a x; backward dx = dy (x > 0 ? 1 : a)
__global__ void prelu_kernel(const half_t* x, int ldx, const half_t* dy, int lddy, const float* slope, half_t* out, int ldo, int rows, int cols) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = cols / 8;
    if (idx >= (long)rows * nch) return;
    const int r = (int)(idx / nch), c = (int)(idx % nch) * 8;
    const float a = slope[0];
    const half8_t v = *reinterpret_cast<const half8_t*>(x + (size_t)r * ldx + c);
    half8_t o;
    if (dy) {
        const half8_t g = *reinterpret_cast<const half8_t*>(dy + (size_t)r * lddy + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)g[j] * ((float)v[j] > 0.f ? 1.0f : a));
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)v[j] > 0.f ? (float)v[j] : a * (float)v[j]);
    }
    *reinterpret_cast<half8_t*>(out + (size_t)r * ldo + c) = o;
}

// MaxPool2d(2, 2) over NHWC rows (arcface_resnet.py:74).  dy == NULL: out (B, h/2, w/2, c) = window max; else out (B, h, w, c) = dy routed
// to the FIRST maximal element of each window (row-major), zero elsewhere
__global__ void maxpool2_kernel(const half_t* x, const half_t* dy, half_t* out, int batch, int h, int w, int c) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = c / 8, ho = h / 2, wo = w / 2;
    if (idx >= (long)batch * ho * wo * nch) return;
    const int ch = (int)(idx % nch);
    const long px = idx / nch;
    const int j = (int)(px % wo), i = (int)((px / wo) % ho), b = (int)(px / ((long)ho * wo));
    half8_t v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const half8_t*>(x + (((size_t)b * h + 2 * i + (t >> 1)) * w + 2 * j + (t & 1)) * c + ch * 8);
    if (!dy) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)fmaxf(fmaxf((float)v[0][e], (float)v[1][e]), fmaxf((float)v[2][e], (float)v[3][e]));
        *reinterpret_cast<half8_t*>(out + (size_t)px * c + ch * 8) = o;
        return;
    }
    const half8_t g = *reinterpret_cast<const half8_t*>(dy + (size_t)px * c + ch * 8);
    half8_t o[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int best = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t) if ((float)v[t][e] > (float)v[best][e]) best = t;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t][e] = t == best ? g[e] : (half_t)0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<half8_t*>(out + (((size_t)b * h + 2 * i + (t >> 1)) * w + 2 * j + (t & 1)) * c + ch * 8) = o[t];
}

// FaceLoss.preprocess (loss.py:26-36): RGB -> grayscale (0.2989, 0.5870, 0.1140) and F.interpolate(bilinear, align_corners=False) to
// (S, S), optionally / 127.5 - 1.  x: fp32 NCHW (B, 3, H, W) (channel stride H*W, image stride xs); y: fp32 (B, 1, S, S).
__device__ __forceinline__ void bilin_taps(int o, int in, int out, int& i0, int& i1, float& w1) {
    float src = ((float)o + 0.5f) * ((float)in / (float)out) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 < in - 1 ? i0 + 1 : i0;
    w1 = src - (float)i0;
}
__global__ void gray_resize_kernel(const float* x, long xs, float* y, int batch, int h, int w, int S, float mul, float add) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * S * S) return;
    const int ox = (int)(idx % S), oy = (int)((idx / S) % S), b = (int)(idx / ((long)S * S));
    int y0, y1, x0, x1;
    float wy, wx;
    bilin_taps(oy, h, S, y0, y1, wy);
    bilin_taps(ox, w, S, x0, x1, wx);
    const float* p = x + (size_t)b * xs;
    const long hw = (long)h * w;
    auto gray = [&](int yy, int xx) { const long o = (long)yy * w + xx; return 0.2989f * p[o] + 0.5870f * p[hw + o] + 0.1140f * p[2 * hw + o]; };
    const float v = (1.f - wy) * ((1.f - wx) * gray(y0, x0) + wx * gray(y0, x1)) + wy * ((1.f - wx) * gray(y1, x0) + wx * gray(y1, x1));
    y[idx] = v * mul + add;
}
// backward: one thread per INPUT pixel gathers from the output pixels whose two taps per axis include it (deterministic, no atomics)
__global__ void gray_resize_bwd_kernel(const float* dy, long dys, float* dx, int batch, int h, int w, int S, float mul) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * h * w) return;
    const int ix = (int)(idx % w), iy = (int)((idx / w) % h), b = (int)(idx / ((long)h * w));
    const float sy = (float)S / (float)h, sx = (float)S / (float)w;
    const int oy0 = max(0, (int)floorf(((float)iy - 1.0f + 0.5f) * sy - 0.5f) - 1), oy1 = min(S - 1, (int)ceilf(((float)iy + 1.0f + 0.5f) * sy - 0.5f) + 1);
    const int ox0 = max(0, (int)floorf(((float)ix - 1.0f + 0.5f) * sx - 0.5f) - 1), ox1 = min(S - 1, (int)ceilf(((float)ix + 1.0f + 0.5f) * sx - 0.5f) + 1);
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
        int y0, y1;
        float wy;
        bilin_taps(oy, h, S, y0, y1, wy);
        const float cy = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
        if (cy == 0.f) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            int x0, x1;
            float wx;
            bilin_taps(ox, w, S, x0, x1, wx);
            const float cx = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
            if (cx != 0.f) acc += cy * cx * dy[(size_t)b * dys + (size_t)oy * S + ox];
        }
    }
    acc *= mul;
    const long hw = (long)h * w, o = (long)iy * w + ix;
    float* d = dx + (size_t)b * 3 * hw;
    d[o] = 0.2989f * acc;
    d[hw + o] = 0.5870f * acc;
    d[2 * hw + o] = 0.1140f * acc;
}

// torch.nn.CosineEmbeddingLoss (loss.py:17,78; margin 0, mean): target +1: 1 - cos(e1, e2); target -1: max(0, cos).  One wave per sample;
// per-sample losses to ls[b], de2 = gscale / B * dloss/de2 (fp16).  e1 fp16 (the real image's embedding, no gradient), e2 fp16.
__global__ void cosine_loss_kernel(const half_t* e1, const half_t* e2, int dim, float target, float gscale, int batch, float* ls, half_t* de2) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const half_t* a = e1 + (size_t)b * dim;
    const half_t* c = e2 + (size_t)b * dim;
    float ab = 0.f, aa = 0.f, cc = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float x = (float)a[i], y = (float)c[i];
        ab += x * y; aa += x * x; cc += y * y;
    }
    ab = pv_wave_sum(ab); aa = pv_wave_sum(aa); cc = pv_wave_sum(cc);
    const float eps = 1e-8f;                        // torch: cos = ab / sqrt((aa + eps) * (cc + eps))
    const float den = sqrtf((aa + eps) * (cc + eps));
    const float cs = ab / den;
    float loss, dcos;
    if (target > 0.f) { loss = 1.f - cs; dcos = -1.f; }
    else { loss = cs > 0.f ? cs : 0.f; dcos = cs > 0.f ? 1.f : 0.f; }
    if (lane == 0) ls[b] = loss;
    if (de2) {
        const float k = gscale * dcos / (float)batch;
        for (int i = lane; i < dim; i += 64) {
            const float x = (float)a[i], y = (float)c[i];
            de2[(size_t)b * dim + i] = (half_t)(k * (x / den - cs * y / (cc + eps)));
        }
    }
}

// in-place backward of softmax_rows: dp <- scale * p * (dp - sum_j p_j dp_j) per row (the GEMM-composed attention of the VAE mid block)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const half_t* __restrict__ pm, int ldp, half_t* __restrict__ dp, int lddp, int cols, float scale) {
    __shared__ float red[4];
    const half_t* prow = pm + (size_t)blockIdx.x * ldp;
    half_t* drow = dp + (size_t)blockIdx.x * lddp;
    const int nch = cols >> 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s = 0.f;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const half8_t a = *reinterpret_cast<const half8_t*>(prow + ch * 8), g = *reinterpret_cast<const half8_t*>(drow + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)g[j];
    }
    s = pv_wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float dot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const half8_t a = *reinterpret_cast<const half8_t*>(prow + ch * 8), g = *reinterpret_cast<const half8_t*>(drow + ch * 8);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)(scale * (float)a[j] * ((float)g[j] - dot));
        *reinterpret_cast<half8_t*>(drow + ch * 8) = o;
    }
}

// out = dy where lo < y < hi, else 0 (gradient of .clamp(lo, hi), infer.py:122)
__global__ void clamp_mask_kernel(const float* y, const float* dy, float lo, float hi, float* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (y[i] > lo && y[i] < hi) ? dy[i] : 0.f;
}

__global__ void add_rows_kernel(const half_t* a, int lda, const half_t* b, int ldb, half_t* out, int ldo, int rows, int cols) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = cols / 8;
    if (idx >= (long)rows * nch) return;
    const int r = (int)(idx / nch), c = (int)(idx % nch) * 8;
    const half8_t x = *reinterpret_cast<const half8_t*>(a + (size_t)r * lda + c);
    const half8_t y = *reinterpret_cast<const half8_t*>(b + (size_t)r * ldb + c);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)x[j] + (float)y[j]);
    *reinterpret_cast<half8_t*>(out + (size_t)r * ldo + c) = o;
}

// z[b][2i][2j][:] = x[b][i][j][:], zeros elsewhere (input of the data-gradient convolution of a stride-2 3x3 conv)
__global__ void dilate2x_kernel(const half_t* x, int ldx, half_t* z, int batch, int h, int w, int c) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = c / 8;
    const long total = (long)batch * 4 * h * w * nch;
    if (idx >= total) return;
    const int ch = (int)(idx % nch);
    const long px = idx / nch;
    const int j2 = (int)(px % (2 * w)), i2 = (int)((px / (2 * w)) % (2 * h)), b = (int)(px / ((long)4 * h * w));
    half8_t v = tz8();
    if (!(i2 & 1) && !(j2 & 1)) v = *reinterpret_cast<const half8_t*>(x + (((size_t)b * h + i2 / 2) * w + j2 / 2) * ldx + ch * 8);
    *reinterpret_cast<half8_t*>(z + (size_t)px * c + ch * 8) = v;
}

// out[b][i][j][:] = sum of the 2x2 block of g (+ add): gradient of the x2 nearest upsample
__global__ void pool2x_kernel(const half_t* g, const half_t* add, int ldadd, half_t* out, int batch, int h, int w, int c) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = c / 8;
    const long total = (long)batch * h * w * nch;
    if (idx >= total) return;
    const int ch = (int)(idx % nch);
    const long px = idx / nch;
    const int j = (int)(px % w), i = (int)((px / w) % h), b = (int)(px / ((long)h * w));
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    if (add) {
        const half8_t v = *reinterpret_cast<const half8_t*>(add + (size_t)px * ldadd + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (float)v[e];
    }
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const half8_t v = *reinterpret_cast<const half8_t*>(g + (((size_t)b * 2 * h + 2 * i + di) * 2 * w + 2 * j + dj) * c + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
        }
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)a[e];
    *reinterpret_cast<half8_t*>(out + (size_t)px * c + ch * 8) = o;
}

__global__ void sign_kernel(const float* x, float coef, float* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] > 0.f ? coef : (x[i] < 0.f ? -coef : 0.f);
}

// out[b][e][:] = scale * x[b * seq + idx[b] + e][:]  (fp16 rows -> fp32): gradient of the concept rows written by
// _inject_concept_embeddings (clip.py:17-24)
__global__ void gather_rows_kernel(const half_t* x, int ldx, const int32_t* idx, float* out, int batch, int seq, int n_e, int dim, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * n_e * dim) return;
    const int c = (int)(i % dim), e = (int)((i / dim) % n_e), b = (int)(i / ((long)dim * n_e));
    const int row = idx[b] + e;
    out[i] = (row >= 0 && row < seq) ? scale * (float)x[((size_t)b * seq + row) * ldx + c] : 0.f;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient dW[n][k] = sum_m dY[m][n] X[m][k] (both operands are ROW matrices over the m the sum runs over - the layout the forward
// and the data-gradient chain leave them in).  The contraction index is the slow one of both operands, so the MFMA operands are read from
// row-major LDS tiles with the transposing LDS read (ds_read_b64_tr_b16, as the attention backward does) - no transposed copies of dY / X in
// HBM (the first version ran pv_transpose_f16 on both and then the forward GEMM: 280 extra launches per training step).
// One workgroup = a 128 (n) x 128 (k) tile of dW over a slab of rows [split * rows_per_split, ...): 64 rows per step, next step's global
// loads in flight while the current one multiplies; 4 waves as 2 x 2, 64 x 64 per wave.  Partial slabs fp32 [nsplit][n][k], summed in split
// order by the caller's reduce launch (deterministic).
struct WgradArgs { const half_t* dy; int lddy; const half_t* x; int ldx; int m, n, k; float* out; int rows_per_split; };

constexpr int WG_RS = 136;                                  // LDS row stride (halfs) of the 64 x 128 tiles

__device__ __forceinline__ half8_t wg_tfrag(const half_t* sR, int f, int s2, int fr, int fq) {
    const half_t* a = sR + (s2 * 32 + fq * 4 + (fr >> 2)) * WG_RS + f * 16 + (fr & 3) * 4;
    const fp16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a));
    const fp16x4_t t2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a + 16 * WG_RS));
    const half4_t h1 = __builtin_bit_cast(half4_t, t1), h2 = __builtin_bit_cast(half4_t, t2);
    return __builtin_shufflevector(h1, h2, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256) void wgrad_tn_kernel(const WgradArgs p) {
    __shared__ __attribute__((aligned(16))) half_t sA[64 * WG_RS];
    __shared__ __attribute__((aligned(16))) half_t sB[64 * WG_RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 128, split = blockIdx.z;
    const int m0 = split * p.rows_per_split, m1 = min(p.m, m0 + p.rows_per_split);
    half8_t ra[4], rb[4];
    auto gload = [&](int mb) {                               // 64 rows x 16 chunks per tile: 4 chunks per thread and tile, zero beyond m1 / n / k
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, r = idx >> 4, c = (idx & 15) * 8;
            const int row = mb + r, rc = min(row, p.m - 1);
            const bool okr = row < m1;
            const half8_t va = *reinterpret_cast<const half8_t*>(p.dy + (size_t)rc * p.lddy + min(n0 + c, p.n - 8));
            const half8_t vb = *reinterpret_cast<const half8_t*>(p.x + (size_t)rc * p.ldx + min(k0 + c, p.k - 8));
            ra[i] = (okr && n0 + c < p.n) ? va : tz8();
            rb[i] = (okr && k0 + c < p.k) ? vb : tz8();
        }
    };
    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    if (m0 < m1) gload(m0);
    for (int mb = m0; mb < m1; mb += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, r = idx >> 4, c = (idx & 15) * 8;
            *reinterpret_cast<half8_t*>(sA + r * WG_RS + c) = ra[i];
            *reinterpret_cast<half8_t*>(sB + r * WG_RS + c) = rb[i];
        }
        __syncthreads();
        if (mb + 64 < m1) gload(mb + 64);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            half8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = wg_tfrag(sA, wm * 4 + i, s2, fr, fq);
                fb[i] = wg_tfrag(sB, wn * 4 + i, s2, fr, fq);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    float* out = p.out + (size_t)split * p.n * p.k;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k0 + (wn * 4 + j) * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n0 + (wm * 4 + i) * 16 + fq * 4 + r;
                if (nn < p.n && kk < p.k) out[(size_t)nn * p.k + kk] = acc[i][j][r];
            }
        }
}

}  // namespace

extern "C" int pv_cross_attention_backward(const pv_xattn_bwd_params* p, void* stream) {
    if (!p || !p->q || !p->kt || !p->vt || !p->kip || !p->vip || !p->dout || !p->dq || !p->partial || !p->stats || !p->dkt || !p->dvt || !p->dkip || !p->dvip ||
        p->batch <= 0 || p->heads <= 0 || p->nq <= 0 || p->nt <= 0 || p->nt > XB_IP0 || p->nip <= 0 || p->nip > XB_KEYS - XB_IP0 || p->out_scale == 0.f ||
        p->ld_dt < p->heads * p->d || p->ld_di < p->heads * p->d)
        return (int)hipErrorInvalidValue;
    if ((p->ldq | p->ldkt | p->ldvt | p->ldkip | p->ldvip | p->lddo | p->lddq) % 8) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    switch (p->d) {
        case 40: return launch_xattn_bwd<40>(*p, s);
        case 80: return launch_xattn_bwd<80>(*p, s);
        case 160: return launch_xattn_bwd<160>(*p, s);
        default: return (int)hipErrorInvalidValue;
    }
}

extern "C" int pv_attention_backward(const pv_attn_bwd_params* p, void* stream) {
    if (!p || !p->q || !p->k || !p->v || !p->out || !p->dout || !p->lse || !p->delta || !p->qs || !p->dq || !p->dk || !p->dv) return (int)hipErrorInvalidValue;
    if (p->batch <= 0 || p->heads <= 0 || p->nq <= 0 || p->nk <= 0) return (int)hipErrorInvalidValue;
    if ((p->ldq | p->ldk | p->ldv | p->ldo | p->lddo | p->lddq | p->lddk | p->lddv | p->ldqs) % 8) return (int)hipErrorInvalidValue;
    if (p->ws != nullptr && p->ws_bytes < 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (pv_attn8_bwd_eligible(*p)) return pv_attn8_bwd_launch(*p, s);
    switch (p->d) {
        case 40: return launch_attn_bwd<40>(*p, s);
        case 64: return launch_attn_bwd<64>(*p, s);
        case 80: return launch_attn_bwd<80>(*p, s);
        case 160: return launch_attn_bwd<160>(*p, s);
        default: return (int)hipErrorInvalidValue;
    }
}

extern "C" int pv_groupnorm_backward(const pv_groupnorm_bwd_params* p, void* stream) {
    if (!p || !p->x0 || !p->dy || !p->stats || !p->gamma || !p->beta || !p->partial || !p->sums) return (int)hipErrorInvalidValue;
    const int C = p->c0 + p->c1;
    if (C % 8 || p->c0 % 8 || C % p->groups || p->groups > 64 || p->splits < 1 || p->splits > 64 || (p->c1 > 0 && !p->x1)) return (int)hipErrorInvalidValue;
    if ((p->ld0 | p->ld1 | p->ld_dy | p->ld_dx0 | p->ld_dx1 | p->ld_add0 | p->ld_add1) % 8) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = C / 8, zb = (nchunk + GB_CPB - 1) / GB_CPB;
    const int cpb = (nchunk + zb - 1) / zb, rows = 256 / cpb;          // chunk lanes x pixel lanes of a workgroup (cpb <= 64, rows >= 4)
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3((unsigned)p->splits, (unsigned)p->batch, (unsigned)zb), dim3(256), 0, s, *p, cpb, rows);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((unsigned)((p->batch * p->groups + 3) / 4)), dim3(256), 0, s, *p);
    const int ppb = 64;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((p->hw + ppb - 1) / ppb), (unsigned)p->batch, (unsigned)zb), dim3(256), 0, s, *p, ppb, cpb, rows);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_geglu_backward(const void* h, int32_t ldh, const void* dy, int32_t lddy, void* dh, int32_t lddh, int32_t rows, int32_t n, void* stream) {
    if (!h || !dy || !dh || rows <= 0 || n <= 0 || n % 8 || (ldh | lddy | lddh) % 8) return (int)hipErrorInvalidValue;
    const long total = (long)rows * (n / 8);
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)h, ldh, (const half_t*)dy, lddy,
                       (half_t*)dh, lddh, rows, n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_act_backward(const void* x, int32_t ldx, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t rows, int32_t cols, int32_t act,
                               void* stream) {
    if (!x || !dy || !dx || rows <= 0 || cols <= 0 || cols % 8 || (ldx | lddy | lddx) % 8) return (int)hipErrorInvalidValue;
    const long total = (long)rows * (cols / 8);
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, (const half_t*)dy, lddy,
                       (half_t*)dx, lddx, rows, cols, act);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_act_forward(const void* x, int32_t ldx, void* y, int32_t ldy, int32_t rows, int32_t cols, int32_t act, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || cols % 8 || (ldx | ldy) % 8) return (int)hipErrorInvalidValue;
    const long total = (long)rows * (cols / 8);
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, (half_t*)y, ldy, rows, cols,
                       act);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_dropout_f16(const void* x, int32_t ldx, void* out, int32_t ldo, const void* add, int32_t ldadd, int32_t rows, int32_t cols, int32_t copies,
                              float p, const int32_t* rng, int32_t site, int32_t backward, void* stream) {
    if (!x || !out || !rng || rows <= 0 || cols <= 0 || cols % 8 || copies < 1 || copies > 4 || !(p >= 0.f && p < 1.f) || (ldx | ldo | ldadd) % 8)
        return (int)hipErrorInvalidValue;
    const long total = (long)rows * (cols / 8);
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, (half_t*)out, ldo,
                       (const half_t*)add, ldadd, rows, cols, copies, p, reinterpret_cast<const uint32_t*>(rng), (uint32_t)site, backward);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_col_affine_f16(const void* x, int32_t ldx, const float* scale, const float* shift, void* y, int32_t ldy, int32_t rows, int32_t cols, void* stream) {
    if (!x || !scale || !y || rows <= 0 || cols <= 0 || cols % 8 || (ldx | ldy) % 8) return (int)hipErrorInvalidValue;
    const long total = (long)rows * (cols / 8);
    hipLaunchKernelGGL(col_affine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, scale, shift, (half_t*)y, ldy,
                       rows, cols);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_prelu_f16(const void* x, int32_t ldx, const void* dy, int32_t lddy, const float* slope, void* out, int32_t ldo, int32_t rows, int32_t cols,
                            void* stream) {
    if (!x || !slope || !out || rows <= 0 || cols <= 0 || cols % 8 || (ldx | lddy | ldo) % 8) return (int)hipErrorInvalidValue;
    const long total = (long)rows * (cols / 8);
    hipLaunchKernelGGL(prelu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, (const half_t*)dy, lddy, slope,
                       (half_t*)out, ldo, rows, cols);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_maxpool2x2(const void* x, const void* dy, void* out, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
    if (!x || !out || batch <= 0 || h <= 0 || w <= 0 || (h | w) & 1 || c <= 0 || c % 8) return (int)hipErrorInvalidValue;
    const long total = (long)batch * (h / 2) * (w / 2) * (c / 8);
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)dy, (half_t*)out,
                       batch, h, w, c);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_gray_resize(const float* x, int64_t x_image_stride, float* y, int32_t batch, int32_t h, int32_t w, int32_t size, float mul, float add,
                              void* stream) {
    if (!x || !y || batch <= 0 || h <= 0 || w <= 0 || size <= 0) return (int)hipErrorInvalidValue;
    const long total = (long)batch * size * size;
    hipLaunchKernelGGL(gray_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)x_image_stride, y, batch, h, w, size, mul,
                       add);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_gray_resize_backward(const float* dy, int64_t dy_image_stride, float* dx, int32_t batch, int32_t h, int32_t w, int32_t size, float mul,
                                       void* stream) {
    if (!dy || !dx || batch <= 0 || h <= 0 || w <= 0 || size <= 0) return (int)hipErrorInvalidValue;
    const long total = (long)batch * h * w;
    hipLaunchKernelGGL(gray_resize_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, (long)dy_image_stride, dx, batch, h, w,
                       size, mul);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_cosine_embedding_loss(const void* e1, const void* e2, int32_t batch, int32_t dim, float target, float gscale, float* per_sample, void* de2,
                                        void* stream) {
    if (!e1 || !e2 || !per_sample || batch <= 0 || dim <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cosine_loss_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, (const half_t*)e1, (const half_t*)e2, dim, target, gscale, batch,
                       per_sample, (half_t*)de2);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_wgrad_tn(const void* dy, int32_t lddy, const void* x, int32_t ldx, int32_t m, int32_t n, int32_t k, float* out, int32_t nsplit,
                           int32_t rows_per_split, void* stream) {
    if (!dy || !x || !out || m <= 0 || n < 8 || k < 8 || (n | k | lddy | ldx) % 8 || nsplit <= 0 || rows_per_split <= 0 || rows_per_split % 64 ||
        (long)nsplit * rows_per_split < m)
        return (int)hipErrorInvalidValue;
    const WgradArgs a{reinterpret_cast<const half_t*>(dy), lddy, reinterpret_cast<const half_t*>(x), ldx, m, n, k, out, rows_per_split};
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3((unsigned)((k + 127) / 128), (unsigned)((n + 127) / 128), (unsigned)nsplit), dim3(256), 0, (hipStream_t)stream, a);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_softmax_rows_backward(const void* p, int32_t ldp, void* dp, int32_t lddp, int32_t rows, int32_t cols, float scale, void* stream) {
    if (!p || !dp || rows <= 0 || cols <= 0 || cols % 8 || (ldp | lddp) % 8) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const half_t*)p, ldp, (half_t*)dp, lddp, cols, scale);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_clamp_mask_f32(const float* y, const float* dy, float lo, float hi, float* out, int64_t n, void* stream) {
    if (!y || !dy || !out || n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(clamp_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dy, lo, hi, out, (long)n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_add_rows_f16(const void* a, int32_t lda, const void* b, int32_t ldb, void* out, int32_t ldo, int32_t rows, int32_t cols, void* stream) {
    if (!a || !b || !out || rows <= 0 || cols <= 0 || cols % 8 || (lda | ldb | ldo) % 8) return (int)hipErrorInvalidValue;
    const long total = (long)rows * (cols / 8);
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, lda, (const half_t*)b, ldb,
                       (half_t*)out, ldo, rows, cols);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_dilate2x(const void* x, int32_t ldx, void* z, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
    if (!x || !z || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 || ldx % 8 || ldx < c) return (int)hipErrorInvalidValue;
    const long total = (long)batch * 4 * h * w * (c / 8);
    hipLaunchKernelGGL(dilate2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, (half_t*)z, batch, h, w, c);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_pool2x_sum(const void* g, const void* add, int32_t ldadd, void* out, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
    if (!g || !out || batch <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 || (add && (ldadd % 8 || ldadd < c))) return (int)hipErrorInvalidValue;
    const long total = (long)batch * h * w * (c / 8);
    hipLaunchKernelGGL(pool2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)g, (const half_t*)add, ldadd, (half_t*)out,
                       batch, h, w, c);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_sign_f32(const float* x, float coef, float* out, int64_t n, void* stream) {
    if (!x || !out || n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, coef, out, (long)n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_gather_rows_f32(const void* x, int32_t ldx, const int32_t* idx, float* out, int32_t batch, int32_t seq, int32_t n_e, int32_t dim, float scale,
                                  void* stream) {
    if (!x || !idx || !out || batch <= 0 || seq <= 0 || n_e <= 0 || dim <= 0) return (int)hipErrorInvalidValue;
    const long total = (long)batch * n_e * dim;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, idx, out, batch, seq, n_e,
                       dim, scale);
    return PV_CHECK_LAUNCH();
}
