// Flash-style attention kernels for gfx950:
//   pv_attention        self attention (UNet attn1, CLIP encoders), online softmax over 64-key tiles
//   pv_cross_attention  PhotoVerse dual-branch cross attention: text keys + image-token keys in ONE
//                       96-row K/V image, two independent softmaxes, one P.V pass
//
// Wave-level scheme (both kernels): a wave owns 32 query rows (2 fragments of 16).  Scores are
// computed TRANSPOSED, S^T[key][q] = K . Q^T with v_mfma_f32_16x16x32_f16 (K rows as MFMA-A from LDS,
// Q as MFMA-B from registers), so a lane holds 4 consecutive keys x its own query column: row max /
// row sum are 15 in-register ops + two cross-lane steps.  The exponentiated scores are already laid
// out as the B operand of the second product O^T[dv][q] = V^T . P^T (k order permuted identically on
// both operands), whose A operand V^T is read straight from the row-major V tile with the gfx950
// transposing LDS read ds_read_b64_tr_b16.  No P round trip through LDS.
#include "pv_common.h"
#include <type_traits>

#ifndef PV_ATTN8_DEFAULT
// variant of attn8_kernel (its VAR bits) taken where a d = 40 launch has at least one 512-query workgroup per CU; -1 = the 4-wave kernel
// everywhere.  497 = stagger + exponentiate-first reference check + V-fragment prefetch / C-operand reference + 48-deep score contraction (225:
// 392-404 us against 483 us for attn_kernel<40, 4, true> on the 64 x 64 level's launch, same box, sustained) + the tile's LDS-DMA issued behind
// the prefetch reads + per-segment priority (388-390 us where 225 takes 404; +1.0 % of a step over 225, profiles/r05_attn8_ab7*.txt).
#define PV_ATTN8_DEFAULT 497
#endif
#ifndef PV_ATTN8_FENCE_LOOP
#define PV_ATTN8_FENCE_LOOP 1      // 0: the round-5 build (fence in the prologue's copy only) - timing A/B of the fence, never shipped
#endif
#ifndef PV_ATTN8_MAX3
// 1: the exponentiate-first reference check reduces the 32 packed fp16 P registers of a tile with gfx950's three-input v_pk_maximum3_f16 (16 instructions on the
// shared vector issue port instead of 31 v_pk_max_f16); same boolean, same results.  Round 6, same box, sustained: 378.4 -> 368.5 us (-2.6 %) together with
// the loop re-pinned in the instruction-fetch windows (PV_ATTN8_LOOP_PAD 3 -> 2: the shorter vector segment moved the loop; at pad 3 the launch reads 377.7)
#define PV_ATTN8_MAX3 1
#endif
#ifndef PV_ATTN8_RECOMPUTE
// 1: a wave that finds a score above its reference computes the tile's scores AGAIN in its vector segment instead of keeping the 64 score registers alive across
// the check (attn8_kernel, PCHECK path).  On the common path the scores die at their exponential and the packed P takes their registers: 254 -> 204 VGPRs, and the
// eight 64-bit + sixteen 32-bit register copies hipcc had put on the common path are gone.  Round 6, same box: alone 371 -> 377 us (SLOWER: the common path now ends
// in a taken branch), in the loop 33.00 -> 33.11 steps/s at loop pad 4 (+0.3 %, three rounds; profiles/r06_attn8_recompute.txt): the loop decides.
#define PV_ATTN8_RECOMPUTE 1
#endif
#ifndef PV_ATTN8_REDO_HINT
#define PV_ATTN8_REDO_HINT 0      // 1: the 'a score exceeded the reference' branch carries an unlikely hint (A/B switch)
#endif
#ifndef PV_ATTN8_LOOP_PAD
#define PV_ATTN8_LOOP_PAD 4      // -1: no alignment directive (round 5: 3; round 6: 2 with PV_ATTN8_MAX3, 4 with PV_ATTN8_RECOMPUTE - best of the loop A/B)
#endif
#ifndef PV_ATTN_LAZY_UP
#define PV_ATTN_LAZY_UP 8.f    // attn_kernel: how far (log2 units) a score may exceed its row's softmax reference before the reference moves; 0 = eager
#endif
#ifndef PV_ATTN_ABLATE
#define PV_ATTN_ABLATE 0   // 1 no exp, 2 no QK MFMA, 3 no PV MFMA: timing-only builds (wrong results)
#endif

#ifdef PV_ATTN8_STAMPS
// diagnostic build only (tools/diag/attn8_stamps.py): per-wave shader-cycle sums of the four parts of attn8_kernel's tile loop, for three workgroups
__device__ unsigned long long pv_attn8_stamps[3 * 8 * 8];
extern "C" int pv_attn8_read_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pv_attn8_stamps), sizeof(pv_attn8_stamps)); }
#define A8_NOW() __builtin_amdgcn_s_memtime()
#endif

namespace {

template <int D>
struct ACfg {
    static constexpr int DK = (D + 31) / 32 * 32;  // contraction length of Q.K^T padded to the MFMA K
    static constexpr int KSTEPS = DK / 32;
    static constexpr int DVF = (D + 15) / 16;      // 16-wide output fragments of P.V
    // K rows in LDS: DK == 64 -> 128-B rows with the 16-B chunk XOR-swizzled by (row & 7) (conflict-free ds_read_b128, as
    // in the GEMM); otherwise rows padded by 16 B (2-way conflicts, only the small d = 80 / 160 levels)
    static constexpr bool KSWZ = DK == 64;
    static constexpr int KS = KSWZ ? DK : DK + 8;  // LDS row stride in halfs
    static constexpr int KCH = KS / 8;             // 16-B chunks per K row (incl. zero padding)
    __device__ static __forceinline__ int koff(int row, int chunk) {
        return KSWZ ? row * KS + ((chunk ^ (row & 7)) << 3) : row * KS + (chunk << 3);
    }
    // V row stride in halfs.  The P.V operand is read with ds_read_b64_tr_b16: a 32-lane group reads 8 consecutive key rows x 32 B, so
    // the eight rows must land on eight DIFFERENT 32-B slots of the 256-B bank window: stride = 8 * odd dwords (mod 64).  The smallest
    // such stride that holds the DVF fragments: d = 40 -> 48 halfs (the old 56 put rows 0 and 7 on the same banks: SQ_LDS_BANK_CONFLICT
    // was 37 % of the kernel's LDS cycles, profiles/r03_base_pmc_attn40.txt), d = 64 / 80 -> 80, d = 160 -> 176.
    static constexpr int vs_dwords() {
        int v = DVF * 8;
        while ((v & 63) != 8 && (v & 63) != 24 && (v & 63) != 40 && (v & 63) != 56) v += 4;
        return v;
    }
    static constexpr int VS = vs_dwords() * 2;
    static constexpr int CH = D / 8;               // 16-byte chunks per row
};

__device__ __forceinline__ half8_t zero8() { return half8_t{0, 0, 0, 0, 0, 0, 0, 0}; }

// V^T fragment (MFMA-A) for output rows dv0..dv0+15 and the 8 keys {key0+4fq..+3, key0+16+4fq..+3}
__device__ __forceinline__ half8_t vt_frag(const half_t* sV, int VS, int key0, int dv0, int fr, int fq) {
    const half_t* a = sV + (key0 + fq * 4 + (fr >> 2)) * VS + dv0 + (fr & 3) * 4;
    const fp16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a));
    const fp16x4_t t2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a + 16 * VS));
    half8_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (half_t)t1[j];
        r[j + 4] = (half_t)t2[j];
    }
    return r;
}

// waves per SIMD the register allocator is asked to leave room for (occupancy hides the serial QK -> softmax -> PV chain)
template <int D>
constexpr int attn_min_waves() { return 1; }

// NQ: 16-query fragments per wave (a workgroup owns 64 * NQ queries): every K / V fragment read from LDS and every staged tile (two
// barriers) serves NQ fragments
// DMA (d = 40 only): the K / V tiles reach LDS by LDS-DMA through buffer descriptors (the swizzle and the zero padding are per-lane
// SOURCE offsets, out-of-range offsets read zeros) into a ring of four tile slots with one raw s_barrier per PAIR of tiles, instead
// of global -> registers -> ds_write with two __syncthreads per tile: the per-tile stamps of round 2 put 1.4 k of a wave's 5.5 k cycles
// per tile into that staging (tools/diag/attn_stamps.py).
template <int D, int NQ, bool DMA>
__global__ __launch_bounds__(256, attn_min_waves<D>()) void attn_kernel(const pv_attn_params p) {
    using C = ACfg<D>;
    static_assert(!DMA || (C::KSWZ && C::VS == 48 && D == 40), "the DMA staging is laid out for d = 40 (128-B swizzled K rows, 96-B V rows)");
    constexpr int KB = 64;
    constexpr int NCHUNK = KB * C::CH;
    constexpr int KPT = (NCHUNK + 255) / 256;
    // ONES: the P.V output has a spare padded column (D % 16 != 0).  V's column D is set to 1.0 so that column accumulates
    // the softmax denominator sum_k P[q][k] inside the MFMA (fp16-rounded P, consistent with the numerator) - no VALU adds.
#ifdef PV_ATTN_NO_ONES
    constexpr bool ONES = false;
#else
    constexpr bool ONES = (D % 16) != 0;
#endif
    // two K/V LDS stages -> one barrier per tile: measured slower in round 1 and again with NQ = 4 (666 vs 660 us at d = 40, N = 4096): off
    constexpr bool DBUF = false;
    constexpr int STAGE = KB * (C::KS + C::VS);          // halfs per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sbase = reinterpret_cast<half_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    // 1-D grid, XCD-aware: the q-tiles of one (batch, head) get consecutive remapped ids, i.e. run on ONE XCD, so its K/V
    // (655 KB at N = 4096) are fetched into one L2 instead of all eight (FETCH_SIZE was 8x the K/V bytes)
    constexpr int QW = 64 * NQ;
    const int nqt = (p.nq + QW - 1) / QW;
    const int rid = pv_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int qt = rid % nqt, h = (rid / nqt) % p.heads, b = rid / (nqt * p.heads);
    const half_t* Q = reinterpret_cast<const half_t*>(p.q) + (size_t)b * p.nq * p.ldq + h * D;
    const half_t* Kg = reinterpret_cast<const half_t*>(p.k) + (size_t)b * p.nk * p.ldk + h * D;
    const half_t* Vg = reinterpret_cast<const half_t*>(p.v) + (size_t)b * p.nk * p.ldv + h * D;

    // pad columns, written once (staging only touches columns [0, D)): K pads are zero (the contraction runs over DK >= D);
    // V pads are zero except column D = 1.0 when ONES
    for (int st = 0; st < (DMA ? 4 : DBUF ? 2 : 1); ++st) {
        half_t* sK = sbase + st * STAGE;
        half_t* sV = sK + KB * C::KS;
        constexpr int NPC = C::KCH - C::CH, NPV = (C::VS - D) / 8;
        for (int i = tid; i < KB * NPC; i += 256) {
            const int r = i / NPC, c = i - r * NPC;
            *reinterpret_cast<half8_t*>(sK + C::koff(r, C::CH + c)) = zero8();
        }
        for (int i = tid; i < KB * NPV; i += 256) {
            const int r = i / NPV, c = i - r * NPV;
            half8_t v = zero8();
            if (ONES && c == 0) v[0] = (half_t)1.0f;
            *reinterpret_cast<half8_t*>(sV + r * C::VS + D + c * 8) = v;
        }
    }

    const float qscale = rsqrtf((float)D) * 1.4426950408889634f;
    half8_t qf[NQ][C::KSTEPS];
    int qrow[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        qrow[qi] = qt * QW + (wave * NQ + qi) * 16 + fr;
        const int rc = min(qrow[qi], p.nq - 1);
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ++ks) {
            const int c = ks * 4 + fq;
            qf[qi][ks] = c < C::CH ? *reinterpret_cast<const half8_t*>(Q + (size_t)rc * p.ldq + c * 8) : zero8();
            // fold softmax_scale * log2(e) into Q once (one extra fp16 rounding of q): the scores leave the MFMA in log2
            // units and, with the accumulators initialised to -m_run, already relative to the running row maximum
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qi][ks][j] = (half_t)((float)qf[qi][ks][j] * qscale);
        }
    }

    half8_t kreg[KPT], vreg[KPT];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / C::CH, c = idx - key * C::CH;
            const int gk = t * KB + key;
            if (idx < NCHUNK && gk < p.nk) {
                kreg[i] = *reinterpret_cast<const half8_t*>(Kg + (size_t)gk * p.ldk + c * 8);
                vreg[i] = *reinterpret_cast<const half8_t*>(Vg + (size_t)gk * p.ldv + c * 8);
            } else {
                kreg[i] = zero8();
                vreg[i] = zero8();
            }
        }
    };
    auto swrite = [&](int st) {
        half_t* sK = sbase + st * STAGE;
        half_t* sV = sK + KB * C::KS;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / C::CH, c = idx - key * C::CH;
            if (idx < NCHUNK) {
                *reinterpret_cast<half8_t*>(sK + C::koff(key, c)) = kreg[i];
                *reinterpret_cast<half8_t*>(sV + key * C::VS + c * 8) = vreg[i];
            }
        }
    };

    float4_t o[C::DVF][NQ];
#pragma unroll
    for (int f = 0; f < C::DVF; ++f)
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) o[f][qi] = float4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) m_run[qi] = l_run[qi] = 0.f;     // m_run is meaningful from the first tile on (FIRST path)

    // one 64-key tile: S'^T = K.Q'^T - m_run (log2 units, relative to the running maximum), online softmax, O^T += V^T.P^T
    auto tile = [&](int t, int st, const bool MASKED, const bool FIRST) {
        const half_t* sK = sbase + st * STAGE;
        const half_t* sV = sK + KB * C::KS;
        float4_t s[4][NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            const float init = FIRST ? 0.f : -m_run[qi];      // the row constant rides in the MFMA accumulator
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[kb][qi] = float4_t{init, init, init, init};
        }
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const half8_t a = *reinterpret_cast<const half8_t*>(sK + C::koff(kb * 16 + fr, ks * 4 + fq));
#pragma unroll
#if PV_ATTN_ABLATE == 2
                asm volatile("" ::"v"(a));
#else
                for (int qi = 0; qi < NQ; ++qi) s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[qi][ks], s[kb][qi], 0, 0, 0);
#endif
            }
        half8_t pb[2][NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            if (MASKED) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * KB + kb * 16 + fq * 4 + r;
                        if (key >= p.nk || (p.causal && key > qrow[qi])) s[kb][qi][r] = -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qi][r]);
            mx = pv_quad_max(mx);
            // mx = (tile row max) - m_run.  Fast path (wave-uniform, the common case after the first tiles): no row of the
            // wave exceeded its running maximum -> P = exp2(S') with no per-score subtraction at all.
            // Slow path: shift the reference by d = max(mx, 0), rescale the running output by exp2(-d).
            // (lazy since round 5, as in attn8_kernel: the reference moves only when a score exceeds it by more than 8 log2 units - P <= 256 keeps its
            // relative precision in fp16, the sums are fp32; with the eager form this block ran in ~60 % of the (fragment, tile) pairs of a long row)
            float d = 0.f;
            if (FIRST || __any(mx > PV_ATTN_LAZY_UP)) {
                d = FIRST ? (mx == -INFINITY ? 0.f : mx) : fmaxf(mx, 0.f);
                const float alpha = FIRST ? 0.f : PV_EXP2(-d);
                m_run[qi] += d;
                if (!ONES) l_run[qi] *= alpha;
#pragma unroll
                for (int f = 0; f < C::DVF; ++f) o[f][qi] *= alpha;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kb][qi][r] -= d;
            }
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#if PV_ATTN_ABLATE == 1
                    const float e = s[kb][qi][r];
#else
                    const float e = PV_EXP2(s[kb][qi][r]);
#endif
                    if (!ONES) rs += e;
                    s[kb][qi][r] = e;
                }
            if (!ONES) l_run[qi] += rs;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pb[s2][qi][r] = (half_t)s[2 * s2][qi][r];
                    pb[s2][qi][r + 4] = (half_t)s[2 * s2 + 1][qi][r];
                }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int f = 0; f < C::DVF; ++f) {
                const half8_t a = vt_frag(sV, C::VS, s2 * 32, f * 16, fr, fq);
#if PV_ATTN_ABLATE == 3
                asm volatile("" ::"v"(a), "v"(pb[s2][0]), "v"(pb[s2][1]));
#else
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) o[f][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[s2][qi], o[f][qi], 0, 0, 0);
#endif
            }
    };

    int ntiles = (p.nk + KB - 1) / KB;
    if (p.causal) ntiles = min(ntiles, (min(qt * QW + QW - 1, p.nq - 1)) / KB + 1);
    if constexpr (DMA) {
        // K tile = 64 rows x 128 B = 8 pieces of 8 rows (waves w: pieces w, w + 4); V tile = 64 rows x 96 B = 6 linear pieces (waves 0, 1: two,
        // waves 2, 3: one).  Lane l of K piece j: row 8 j + (l >> 3), LDS position l & 7 holds data chunk (l & 7) ^ (row & 7); chunks >= 5
        // (the zero padding of the 64-deep contraction) and keys >= nk read out of range = zeros.  V: linear 16-B chunk 64 j + l = (row, c) of
        // the [64][6] tile; c == 5 is the pad chunk holding the 1.0 column, written once above: those lanes stay out of the DMA (EXEC mask).
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Kg), 0, ((p.nk - 1) * p.ldk + D) * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Vg), 0, ((p.nk - 1) * p.ldv + D) * 2, 0x00020000);
        constexpr unsigned OOB = 0x80000000u;
        unsigned koff[2], voff[2];
        bool vlive[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 8 * (wave + 4 * i) + (lane >> 3), c = (lane & 7) ^ (lane >> 3);
            koff[i] = c < C::CH ? (unsigned)(r * p.ldk * 2 + c * 16) : OOB;
            const int q = 64 * (wave + 4 * i) + lane, vr = q / 6, vc = q - vr * 6;
            voff[i] = (unsigned)(vr * p.ldv * 2 + vc * 16);
            vlive[i] = vc < C::CH;
        }
        auto issue_tile = [&](int t) {
            char* sK = reinterpret_cast<char*>(sbase + (t & 3) * STAGE);
            char* sV = sK + KB * C::KS * 2;
            const int sk = t * KB * p.ldk * 2, sv = t * KB * p.ldv * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, PV_LDS_PTR(sK + (wave + 4 * i) * 1024), 16, (int)koff[i], sk, 0, 0);
            if (vlive[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, PV_LDS_PTR(sV + wave * 1024), 16, (int)voff[0], sv, 0, 0);
            if (wave < 2) {
                if (vlive[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, PV_LDS_PTR(sV + (wave + 4) * 1024), 16, (int)voff[1], sv, 0, 0);
            }
        };
        __syncthreads();                           // the pad columns written above are visible before any tile is read
        issue_tile(0);
        if (ntiles > 1) issue_tile(1);
        for (int t = 0; t < ntiles; ++t) {
            if ((t & 1) == 0) {
                // ONE barrier per PAIR of tiles (four slots: two being read, two in flight): tiles t and t + 1 were issued two tiles ago and
                // nothing younger is in flight, so "landed" is vmcnt(0); behind the barrier the slots of tiles t - 2 / t - 1 are read out
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (t + 2 < ntiles) issue_tile(t + 2);
                if (t + 3 < ntiles) issue_tile(t + 3);
            }
            const bool need_mask = p.causal || (t + 1) * KB > p.nk;
            tile(t, t & 3, need_mask, t == 0);
        }
    } else {
    gload(0);
    if (DBUF) {
        swrite(0);
        __syncthreads();
    }
    for (int t = 0; t < ntiles; ++t) {
        const int st = DBUF ? (t & 1) : 0;
        if (!DBUF) {
            __syncthreads();  // previous tile fully consumed
            swrite(0);
            __syncthreads();
        }
        if (t + 1 < ntiles) gload(t + 1);
        const bool need_mask = p.causal || (t + 1) * KB > p.nk;
        tile(t, st, need_mask, t == 0);
        if (DBUF) {
            if (t + 1 < ntiles) swrite(st ^ 1);   // the other stage was last read in iteration t-1 (barrier below)
            __syncthreads();
        }
    }
    }

    half_t* O = reinterpret_cast<half_t*>(p.out) + (size_t)b * p.nq * p.ldo + h * D;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float l;
        if (ONES) {
            // denominator lives in output column D: fragment D/16, lanes with fq == (D%16)/4, register (D%4)
            l = __shfl(o[D / 16][qi][D % 4], fr + 16 * ((D % 16) / 4), 64);
        } else {
            l = pv_quad_sum(l_run[qi]);
        }
        const float inv = 1.0f / l;
        // log-sum-exp of the scaled scores in log2 units (P = exp2(S' - lse)): what pv_attention_backward recomputes P from
        if (p.lse != nullptr && fq == 0 && qrow[qi] < p.nq) p.lse[((size_t)b * p.heads + h) * p.nq + qrow[qi]] = m_run[qi] + __log2f(l);
        if (qrow[qi] < p.nq) {
#pragma unroll
            for (int f = 0; f < C::DVF; ++f) {
                const int dv = f * 16 + fq * 4;
                if (dv < D) {
                    half4_t ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[f][qi][r] * inv);
                    *reinterpret_cast<half4_t*>(O + (size_t)qrow[qi] * p.ldo + dv) = ov;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// d = 40 self attention as ONE 8-wave workgroup per CU (512 queries), the two waves of every SIMD STAGGERED by one barrier interval
// (MI355X_MICROARCH.md, "Two waves per SIMD"): a wave alternates a MATRIX segment - P.V of tile t, then K.Q^T of tile t + 1: 56 MFMAs and
// the fragment reads, plus its one or two LDS-DMA pieces of tile t + 3 - with a VECTOR segment - the online softmax of tile t + 1: the
// exponentials, no MFMA.  Waves 4-7 run one interval behind waves 0-3, so on every SIMD one wave holds the matrix pipe while its partner
// issues the exponentials; in the 4-wave kernel above the two co-resident workgroups drift freely and a wave's 32-MFMA clump meets its
// partner's clump as often as its softmax.  K / V tiles: the same LDS images, four-slot ring, each tile staged once per 512 queries (half
// the L2 -> LDS bytes of the 256-query workgroup).
//   VAR bit 0 (1): stagger (waves 4-7 one interval late); bit 1 (2): s_setprio 1 for waves 4-7, once, before the loop; bit 2 (4): ONE
//   wave-uniform rescale decision per tile for the four query fragments; bit 3 (8): lazy reference; bit 4 (16): the wave in its matrix
//   segment runs at s_setprio 1 (its MFMAs are 8 issue cycles in 16: they delay the partner's VALU by at most that, while a delayed MFMA
//   stretches the interval for all eight waves); bit 5 (32): the reference check reads the exponentiated scores (below; implies bits 2, 3);
//   bit 6 (64): the V^T fragments of a tile are requested at the HEAD of the vector segment in front of it (the first three of the six: their LDS latency sits under the
//   softmax instead of at the head of the matrix segment) and the first score MFMA of a chain takes -m_run as its C operand from a
//   loop-carried vector (no accumulator-initialising v_mov); bit 7 (128): 48-deep score contraction (16x16x32 + 16x16x16: a quarter fewer
//   score-MFMA cycles; the d = 40 rows are zero beyond column 40 either way); bit 8 (256, with bit 6): the tile's LDS-DMA pieces are issued at the head of the
//   vector segment BEHIND the prefetch reads instead of at the end of the matrix segment in front of them.
// Variant 9 (stagger + lazy reference, decided per query fragment) computes, per query row, exactly what attn_kernel<40, 4, true> computes, in its
// order: BIT-IDENTICAL results (tests/test_hip_kernels.py); variant 1 is that kernel's round-4 (eager) arithmetic.
// Measured (EXPERIMENTS.md, round 5; same box, sustained): the stagger alone ties the 4-wave kernel (483 us: its free-running workgroups de-phase by
// themselves) and beats the same workgroup without it by 5 %; the lazy reference is what moves the launch (435 us), the exponentiate-first check,
// the prefetch and the 48-deep contraction take it to 392 us (225, the default).  Per-segment priorities (241) remove another 14 % of the CYCLES
// and none of the time: the launch runs at the package power limit (2.17 instead of 2.29 GHz).  What does come back as time is removing the
// stall hipcc put at the head of every vector segment (its vmcnt(0) between the LDS-DMA issue and the prefetch reads: bit 8) TOGETHER with
// the priorities: 497 = 225 + 16 + 256 takes 389 us where 225 and 241 take 404 and 481 (bit 8 alone) 414.
template <int VAR>
__global__ __launch_bounds__(512, 2) void attn8_kernel(const pv_attn_params p) {
    constexpr int D = 40, NQ = 4, KB = 64;
    using C = ACfg<D>;
    constexpr bool STAGGER = (VAR & 1) != 0, PRIO = (VAR & 2) != 0, PCHECK = (VAR & 32) != 0, JOINT = (VAR & 4) != 0 || PCHECK, LAZY = (VAR & 8) != 0 || PCHECK;
    constexpr bool SEGPRIO = (VAR & 16) != 0, PREF = (VAR & 64) != 0, K48 = (VAR & 128) != 0;
    constexpr bool DMA_IN_VEC = PREF && (VAR & 256) != 0;     // bit 8 (256): with PREF, the tile's LDS-DMA is issued behind the prefetch reads (below)
    // LAZY: a row's softmax reference moves only when a score exceeds it by more than 8 log2 units (P <= 256: the same relative precision in
    // fp16, sums in fp32).  With the eager form the rescale block runs in ~60 % of the (fragment, tile) pairs of a 4096-key row of random
    // scores (a new maximum among 16 rows x 64 keys has probability ~ min(1, 16 / t) at tile t); lazily, in the first tiles only.
    // PCHECK: the lazy test needs no row maximum at all: P = exp2(S') is computed (and rounded to fp16) first, the packed fp16 values are
    // reduced with v_pk_max_f16 (31 for the wave's 64 x 64 scores instead of 60 fp32 max / lane-swap operations in four dependent chains), and only
    // a wave that finds a P above 256 goes back to the scores, takes the row maxima and rescales (the scores are still in registers).
    constexpr float UP = LAZY ? 8.f : 0.f;
    constexpr int STAGE = KB * (C::KS + C::VS);          // halfs per ring slot (14 336 B)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sbase = reinterpret_cast<half_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    const bool late = STAGGER && wave >= 4;
    constexpr int QW = 64 * NQ * 2;                      // 512 queries per workgroup
    const int nqt = (p.nq + QW - 1) / QW;
    const int rid = pv_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int qt = rid % nqt, h = (rid / nqt) % p.heads, b = rid / (nqt * p.heads);
    const half_t* Q = reinterpret_cast<const half_t*>(p.q) + (size_t)b * p.nq * p.ldq + h * D;
    const half_t* Kg = reinterpret_cast<const half_t*>(p.k) + (size_t)b * p.nk * p.ldk + h * D;
    const half_t* Vg = reinterpret_cast<const half_t*>(p.v) + (size_t)b * p.nk * p.ldv + h * D;

    // pad columns of the four slots, written once (as in attn_kernel): K pads zero, V pads zero except column D = 1.0 (the softmax
    // denominator accumulates in the P.V product's spare output column)
    for (int st = 0; st < 4; ++st) {
        half_t* sK = sbase + st * STAGE;
        half_t* sV = sK + KB * C::KS;
        constexpr int NPC = C::KCH - C::CH, NPV = (C::VS - D) / 8;
        for (int i = tid; i < KB * NPC; i += 512) {
            const int r = i / NPC, c = i - r * NPC;
            *reinterpret_cast<half8_t*>(sK + C::koff(r, C::CH + c)) = zero8();
        }
        for (int i = tid; i < KB * NPV; i += 512) {
            const int r = i / NPV, c = i - r * NPV;
            half8_t v = zero8();
            if (c == 0) v[0] = (half_t)1.0f;
            *reinterpret_cast<half8_t*>(sV + r * C::VS + D + c * 8) = v;
        }
    }

    const float qscale = rsqrtf((float)D) * 1.4426950408889634f;
    half8_t qf[NQ][K48 ? 1 : C::KSTEPS];
    half4_t qf4[NQ];                                      // K48: the queries' columns 32 + 4 fq .. + 3 (B operand of the 16-deep tail step)
    int qrow[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        qrow[qi] = qt * QW + (wave * NQ + qi) * 16 + fr;
        const int rc = min(qrow[qi], p.nq - 1);
#pragma unroll
        for (int ks = 0; ks < (K48 ? 1 : C::KSTEPS); ++ks) {
            const int c = ks * 4 + fq;
            qf[qi][ks] = c < C::CH ? *reinterpret_cast<const half8_t*>(Q + (size_t)rc * p.ldq + c * 8) : zero8();
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qi][ks][j] = (half_t)((float)qf[qi][ks][j] * qscale);
        }
        if (K48) {
            qf4[qi] = fq < 2 ? *reinterpret_cast<const half4_t*>(Q + (size_t)rc * p.ldq + 32 + fq * 4) : half4_t{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) qf4[qi][j] = (half_t)((float)qf4[qi][j] * qscale);
        }
    }

    float4_t o[C::DVF][NQ];
#pragma unroll
    for (int f = 0; f < C::DVF; ++f)
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) o[f][qi] = float4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[NQ];
    float4_t negm[NQ];                                    // PREF: {-m_run} x 4, the C operand of a score chain's first MFMA
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        m_run[qi] = 0.f;
        negm[qi] = float4_t{0.f, 0.f, 0.f, 0.f};
    }
    float4_t s[4][NQ];
    half8_t pb[2][NQ];
    half8_t vf[C::DVF];                                   // PREF: the first three of the tile's six V^T fragments (keys 0-31)

    // --- MATRIX segment, first half: S'^T = K.Q'^T - m_run for tile t (slot t & 3)
    auto qk = [&](int t, const bool FIRST) {
        const half_t* sK = sbase + (t & 3) * STAGE;
        if (!PREF) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const float init = FIRST ? 0.f : -m_run[qi];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) s[kb][qi] = float4_t{init, init, init, init};
            }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const half8_t a = *reinterpret_cast<const half8_t*>(sK + C::koff(kb * 16 + fr, fq));
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[qi][0], PREF ? negm[qi] : s[kb][qi], 0, 0, 0);
        }
        // Keep the 16-deep tail steps behind ALL the 32-deep ones, in EVERY copy of this block (prologue and tile loop).  hipcc is free to put a tail
        // step one or two slots behind the 16x16x32 whose result it accumulates onto; in the backward passes (pv_attnbwd.hip, same two-shape chain) that
        // schedule returned wrong sums, non-deterministically, once the wave ran at s_setprio 1.  The fence does not depend on where the scheduler
        // happens to put the loop's copy today; tests/test_host_cpu.py scans the emitted ISA for such chains (tools/diag/mfma_chain_scan.py).
        if (K48 && (FIRST || PV_ATTN8_FENCE_LOOP)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (K48) {
                // columns 32 + 4 fq .. + 3 of key row kb * 16 + fr: chunk 4 (data) for fq < 2, chunk 5 (zero padding) above
                const half4_t a = *reinterpret_cast<const half4_t*>(sK + C::koff(kb * 16 + fr, 4 + (fq >> 1)) + (fq & 1) * 4);
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, qf4[qi], s[kb][qi], 0, 0, 0);
            } else {
                const half8_t a = *reinterpret_cast<const half8_t*>(sK + C::koff(kb * 16 + fr, 4 + fq));
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[qi][K48 ? 0 : 1], s[kb][qi], 0, 0, 0);
            }
        }
    };
    // --- VECTOR segment: online softmax of tile t; leaves P (fp16, the second product's B operand) in pb
    auto rescale = [&](int qi, float mxq, const bool FIRST) {       // a row exceeded its reference (rare after the first tiles when LAZY)
        const float d = FIRST ? (mxq == -INFINITY ? 0.f : mxq) : fmaxf(mxq, 0.f);
        const float alpha = FIRST ? 0.f : PV_EXP2(-d);
        m_run[qi] += d;
        negm[qi] = float4_t{-m_run[qi], -m_run[qi], -m_run[qi], -m_run[qi]};
#pragma unroll
        for (int f = 0; f < C::DVF; ++f) o[f][qi] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][qi][r] -= d;
    };
    auto row_max = [&](float (&mx)[NQ]) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            float m = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kb][qi][r]);
            mx[qi] = m;
        }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) mx[qi] = pv_quad_max(mx[qi]);
    };
    auto exp_pack = [&](int qi) {                                   // pb[.][qi] = fp16(exp2(s[.][qi])); the scores stay in s
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pb[s2][qi][r] = (half_t)PV_EXP2(s[2 * s2][qi][r]);
                pb[s2][qi][r + 4] = (half_t)PV_EXP2(s[2 * s2 + 1][qi][r]);
            }
    };
    auto mask_tail = [&](int t) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * KB + kb * 16 + fq * 4 + r >= p.nk) s[kb][qi][r] = -INFINITY;
    };
    auto softmax = [&](int t, const bool MASKED, const bool FIRST) {
        if (MASKED) mask_tail(t);
        if (PCHECK) {
            bool redo = FIRST;
            if (!FIRST) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) exp_pack(qi);
                half2_t m2 = half2_t{(half_t)0.f, (half_t)0.f};
#if PV_ATTN8_MAX3
                // gfx950's three-input packed maximum: two of the 32 packed registers per instruction (16 instead of 31 on the shared vector issue port)
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            const half2_t a = half2_t{pb[s2][qi][2 * j], pb[s2][qi][2 * j + 1]}, b2 = half2_t{pb[s2][qi][2 * j + 2], pb[s2][qi][2 * j + 3]};
                            asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(m2) : "v"(m2), "v"(a), "v"(b2));
                        }
#else
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 4; ++j) m2 = __builtin_elementwise_max(m2, half2_t{pb[s2][qi][2 * j], pb[s2][qi][2 * j + 1]});
#endif
                redo = __any(fmaxf((float)m2[0], (float)m2[1]) > 256.f);         // an fp16 overflow (inf) lands here too
            }
#if PV_ATTN8_REDO_HINT
            if (__builtin_expect(redo, 0)) {
#else
            if (redo) {
#endif
#if PV_ATTN8_RECOMPUTE
                // the rare path computes the tile's scores AGAIN (its K fragments are still in slot t & 3, the reference has not moved yet) instead of
                // keeping 64 score registers alive across the check: on the common path they die at their exponential and the packed P takes their place
                if (PREF && K48) {
                    qk(t, FIRST);
                    if (MASKED) mask_tail(t);
                }
#endif
                float mx[NQ];
                row_max(mx);
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    rescale(qi, mx[qi], FIRST);
                    exp_pack(qi);
                }
            }
            return;
        }
        float mx[NQ];
        row_max(mx);
        if (JOINT) {
            if (FIRST || __any(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])) > UP)) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) rescale(qi, mx[qi], FIRST);
            }
        }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            if (!JOINT) {
                if (FIRST || __any(mx[qi] > UP)) rescale(qi, mx[qi], FIRST);
            }
            exp_pack(qi);
        }
    };
    // --- MATRIX segment, second half: O^T += V^T.P^T for tile t
    auto load_v = [&](int t) {
        const half_t* sV = sbase + (t & 3) * STAGE + KB * C::KS;
#pragma unroll
        for (int f = 0; f < C::DVF; ++f) vf[f] = vt_frag(sV, C::VS, 0, f * 16, fr, fq);
    };
    auto pv = [&](int t) {
        const half_t* sV = sbase + (t & 3) * STAGE + KB * C::KS;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int f = 0; f < C::DVF; ++f) {
                const half8_t a = (PREF && s2 == 0) ? vf[f] : vt_frag(sV, C::VS, s2 * 32, f * 16, fr, fq);
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) o[f][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[s2][qi], o[f][qi], 0, 0, 0);
            }
    };

    // LDS-DMA pieces of a tile (1 KiB each): K = 8 pieces of 8 swizzled 128-B rows -> wave w stages piece w; V = 6 linear pieces ->
    // waves 0-5 (lane -> 16-B chunk 64 w + lane of the [64][6] tile; chunk 5 of a row is the pad chunk and stays out of the DMA)
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Kg), 0, ((p.nk - 1) * p.ldk + D) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Vg), 0, ((p.nk - 1) * p.ldv + D) * 2, 0x00020000);
    unsigned koff, voff;
    bool vlive;
    {
        const int r = 8 * wave + (lane >> 3), c = (lane & 7) ^ (lane >> 3);
        koff = c < C::CH ? (unsigned)(r * p.ldk * 2 + c * 16) : 0x80000000u;
        const int q = 64 * wave + lane, vr = q / 6, vc = q - vr * 6;
        voff = (unsigned)(vr * p.ldv * 2 + vc * 16);
        vlive = vc < C::CH && wave < 6;
    }
    auto issue_tile = [&](int t) {
        char* sK = reinterpret_cast<char*>(sbase + (t & 3) * STAGE);
        char* sV = sK + KB * C::KS * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, PV_LDS_PTR(sK + wave * 1024), 16, (int)koff, t * KB * p.ldk * 2, 0, 0);
        if (vlive) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, PV_LDS_PTR(sV + wave * 1024), 16, (int)voff, t * KB * p.ldv * 2, 0, 0);
    };
    auto interval = [&]() {                      // segment boundary: nothing crosses it in either direction
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // The segments are pinned: hipcc sinks pure arithmetic (the exponentials, even whole MFMA chains) across s_barrier into the block of
    // its first use; an empty asm that "modifies" a segment's results keeps them on their side of the barrier.
    auto pin_s = [&]() {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) asm volatile("" : "+v"(s[kb][qi]));
    };
    auto pin_o = [&]() {
#pragma unroll
        for (int f = 0; f < C::DVF; ++f)
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) asm volatile("" : "+v"(o[f][qi]));
    };
    auto pin_p = [&]() {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) asm volatile("" : "+v"(pb[0][qi]), "+v"(pb[1][qi]), "+v"(m_run[qi]));
    };
    auto pin_v = [&]() {
#pragma unroll
        for (int f = 0; f < C::DVF; ++f) asm volatile("" : "+v"(vf[f]));
    };

    const int ntiles = (p.nk + KB - 1) / KB;
    __syncthreads();                             // the pad columns are visible before any tile is read
    issue_tile(0);
    if (ntiles > 1) issue_tile(1);
    if (ntiles > 2) issue_tile(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    interval();
    if (late) interval();                        // waves 4-7: one interval behind
    if (SEGPRIO) __builtin_amdgcn_s_setprio(1);
    qk(0, true);
    pin_s();
    if (SEGPRIO) __builtin_amdgcn_s_setprio(0);
#ifdef PV_ATTN8_STAMPS
    unsigned long long a8_acc[4] = {0, 0, 0, 0}, a8_t0 = 0, a8_t1, a8_t2, a8_t3, a8_c0 = 0, a8_r0 = 0;
#endif
#if PV_ATTN8_LOOP_PAD >= 0
    // Where the tile loop's code sits relative to the instruction-fetch lines moves the whole launch by 2.5 % (period 32 bytes: 372.5 us at the best
    // offset, 376 next to it, 380-382 at the other six; an edit to the PROLOGUE that moved the loop by 12 bytes cost 382.7 -> 392.5 us on another box;
    // profiles/r05_attn8_loop_alignment.txt).  Pinned: align to 32 bytes here, then PV_ATTN8_LOOP_PAD words - executed once, in front of the loop.
    asm volatile(".p2align 5\n .rept %0\n s_nop 0\n .endr" ::"n"(PV_ATTN8_LOOP_PAD));
#endif
    for (int t = 0; t < ntiles; ++t) {
#ifdef PV_ATTN8_STAMPS
        a8_t3 = A8_NOW();                          // end of the matrix segment
#endif
        interval();
#ifdef PV_ATTN8_STAMPS
        a8_t1 = A8_NOW();
        if (t == 8) { a8_c0 = a8_t1; a8_r0 = __builtin_amdgcn_s_memrealtime(); }
        if (t > 8 && t < ntiles - 8) { a8_acc[2] += a8_t3 - a8_t0; a8_acc[3] += a8_t1 - a8_t3; }
#endif
        if (PREF) {
            load_v(t);                           // tile t landed three segments ago; nothing waits for these reads before the matrix segment
            // ... and THEN this wave's LDS-DMA pieces of tile t + 2: hipcc waits vmcnt(0) in front of the first LDS read behind an LDS-DMA issue (it
            // cannot tell the ring slots apart) - issued at the end of the matrix segment, as without PREF, that wait sat in front of the reads
            // above and exposed the DMA's L2 latency at the head of every vector segment.  Slot (t + 2) & 3 held tile t - 2, last read two
            // matrix segments ago; landed-wait at the head of matrix segment t, first read (K) in matrix segment t + 1
            if (DMA_IN_VEC && t >= 1 && t + 2 < ntiles) issue_tile(t + 2);
        }
        softmax(t, (t + 1) * KB > p.nk, t == 0);
        pin_p();
        pin_o();
        if (PREF) pin_v();
#ifdef PV_ATTN8_STAMPS
        a8_t2 = A8_NOW();                          // end of the vector segment
#endif
        interval();
#ifdef PV_ATTN8_STAMPS
        a8_t0 = A8_NOW();
        if (t >= 8 && t < ntiles - 8) { a8_acc[0] += a8_t2 - a8_t1; a8_acc[1] += a8_t0 - a8_t2; }
        if (t == ntiles - 9) {
            const int slot = blockIdx.x == 0 ? 0 : blockIdx.x == 300 ? 1 : blockIdx.x == 700 ? 2 : -1;
            if (slot >= 0 && lane == 0) {
                unsigned long long* o8 = pv_attn8_stamps + (slot * 8 + wave) * 8;
                o8[0] = a8_acc[0]; o8[1] = a8_acc[1]; o8[2] = a8_acc[2]; o8[3] = a8_acc[3];
                o8[4] = a8_t0 - a8_c0; o8[5] = __builtin_amdgcn_s_memrealtime() - a8_r0; o8[6] = (unsigned long long)(ntiles - 16);
            }
        }
#endif
        if (SEGPRIO) __builtin_amdgcn_s_setprio(1);
        // this wave's pieces of tile t + 2 (issued at the end of its previous matrix segment, a whole vector segment ago) have landed
        // (hipcc would put this wait in front of the first LDS read below anyway: it cannot tell the ring slots apart)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pv(t);
        qk(t + 1, false);                        // past the last tile: scores of a stale slot that nothing reads (no branch between the
                                                 // two products, so the K fragments are requested under the P.V MFMAs)
        pin_o();
        pin_s();
        // tile t + 3 goes into the slot of tile t - 1, whose last reader (the late half's P.V) finished one interval ago; it is first
        // read in matrix segment t + 2, behind every wave's landed-wait at the head of its segment t + 1 and a barrier
        if (!DMA_IN_VEC && t + 3 < ntiles) issue_tile(t + 3);
        if (SEGPRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (STAGGER && !late) interval();            // every wave passes the same number of barriers

    half_t* O = reinterpret_cast<half_t*>(p.out) + (size_t)b * p.nq * p.ldo + h * D;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        // the denominator is output column D: fragment D / 16, lanes with fq == (D % 16) / 4, register D % 4
        const float l = __shfl(o[D / 16][qi][D % 4], fr + 16 * ((D % 16) / 4), 64);
        const float inv = 1.0f / l;
        if (p.lse != nullptr && fq == 0 && qrow[qi] < p.nq) p.lse[((size_t)b * p.heads + h) * p.nq + qrow[qi]] = m_run[qi] + __log2f(l);
        if (qrow[qi] < p.nq) {
#pragma unroll
            for (int f = 0; f < C::DVF; ++f) {
                const int dv = f * 16 + fq * 4;
                if (dv < D) {
                    half4_t ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[f][qi][r] * inv);
                    *reinterpret_cast<half4_t*>(O + (size_t)qrow[qi] * p.ldo + dv) = ov;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dual-branch cross attention.  K/V image rows: [0,nt) text, [IP0, IP0+nip) image tokens, rest zero.
constexpr int XKEYS = 96;
constexpr int IP0 = 80;

// qt_per_wg: 128-row query tiles one workgroup walks with ONE staging of the (b, h) K/V image (the staging and its barrier
// were most of a 4096-workgroup launch's time at one tile each).
template <int D, bool MULTI>
__global__ __launch_bounds__(256) void xattn_kernel(const pv_xattn_params p, const int qt_per_wg_arg) {
    const int qt_per_wg = MULTI ? qt_per_wg_arg : 1;      // MULTI = false is the straight one-tile kernel (no prefetch registers)
    using C = ACfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sK = reinterpret_cast<half_t*>(smem);
    half_t* sV = sK + XKEYS * C::KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    const int nqt = (p.nq + 127) / 128;
    const int nqg = (nqt + qt_per_wg - 1) / qt_per_wg;   // query-tile groups per (b, h)
    const int rid = (int)blockIdx.x;     // K/V are 96 rows: no L2 affinity to gain from an XCD remap (measured slower)
    const int qg = rid % nqg, h = (rid / nqg) % p.heads, b = rid / (nqg * p.heads);
    const half_t* Q = reinterpret_cast<const half_t*>(p.q) + (size_t)b * p.nq * p.ldq + h * D;

    // stage K and V (zero-filled pads)
    {
        constexpr int KC = C::KCH, VC = C::VS / 8;
        for (int i = tid; i < XKEYS * KC; i += 256) {
            const int r = i / KC, c = i - r * KC;
            half8_t v = zero8();
            if (c < C::CH) {
                if (r < p.nt)
                    v = *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.kt) + ((size_t)b * p.nt + r) * p.ldkt + h * D + c * 8);
                else if (r >= IP0 && r < IP0 + p.nip)
                    v = *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.kip) + ((size_t)b * p.nip + (r - IP0)) * p.ldkip + h * D + c * 8);
            }
            *reinterpret_cast<half8_t*>(sK + C::koff(r, c)) = v;
        }
        for (int i = tid; i < XKEYS * VC; i += 256) {
            const int r = i / VC, c = i - r * VC;
            half8_t v = zero8();
            if (c < C::CH) {
                if (r < p.nt)
                    v = *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.vt) + ((size_t)b * p.nt + r) * p.ldvt + h * D + c * 8);
                else if (r >= IP0 && r < IP0 + p.nip)
                    v = *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.vip) + ((size_t)b * p.nip + (r - IP0)) * p.ldvip + h * D + c * 8);
            }
            *reinterpret_cast<half8_t*>(sV + r * C::VS + c * 8) = v;
        }
    }

    half8_t qf[2][C::KSTEPS], qn[MULTI ? 2 : 1][MULTI ? C::KSTEPS : 1];
    int qrow[2];
    auto load_q = [&](int qt, auto& dst) {
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int rc = min(qt * 128 + wave * 32 + qi * 16 + fr, p.nq - 1);
#pragma unroll
            for (int ks = 0; ks < C::KSTEPS; ++ks) {
                const int c = ks * 4 + fq;
                dst[qi][ks] = c < C::CH ? *reinterpret_cast<const half8_t*>(Q + (size_t)rc * p.ldq + c * 8) : zero8();
            }
        }
    };
    const int qt0 = qg * qt_per_wg;
    const int qt1 = min(qt0 + qt_per_wg, nqt);
    if constexpr (MULTI) load_q(qt0, qn); else load_q(qt0, qf);
    __syncthreads();

    // to_v_ip_norm (attention_processor.py:397): ||Vip[b,p,h,:]||_2, once per (b,h)
    if (p.vnorm && qg == 0 && tid < p.nip) {
        float a = 0.f;
        for (int d = 0; d < D; ++d) {
            const float v = (float)sV[(IP0 + tid) * C::VS + d];
            a += v * v;
        }
        p.vnorm[((size_t)b * p.heads + h) * p.nip + tid] = sqrtf(a);
    }

    constexpr int NKB = XKEYS / 16;
#pragma unroll 1
    for (int qt = qt0; qt < (MULTI ? qt1 : qt0 + 1); ++qt) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        qrow[qi] = qt * 128 + wave * 32 + qi * 16 + fr;
        if constexpr (MULTI) {
#pragma unroll
            for (int ks = 0; ks < C::KSTEPS; ++ks) qf[qi][ks] = qn[qi][ks];
        }
    }
    if constexpr (MULTI) {
        if (qt + 1 < qt1) load_q(qt + 1, qn);      // next tile's queries are in flight during this tile's softmax
    }
    float4_t s[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) s[kb][qi] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const half8_t a = *reinterpret_cast<const half8_t*>(sK + C::koff(kb * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[qi][ks], s[kb][qi], 0, 0, 0);
        }

    const float sc = rsqrtf((float)D) * 1.4426950408889634f;
    half8_t pb[NKB / 2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        float mt = -INFINITY, mi = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb * 16 + fq * 4 + r;
                const float v = s[kb][qi][r];
                if (key < p.nt) mt = fmaxf(mt, v);
                if (key >= IP0 && key < IP0 + p.nip) mi = fmaxf(mi, v);
            }
        mt = pv_quad_max(mt);
        mi = pv_quad_max(mi);
        float lt = 0.f, li = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb * 16 + fq * 4 + r;
                const bool is_t = key < p.nt;
                const bool is_i = key >= IP0 && key < IP0 + p.nip;
                float e = 0.f;
                if (is_t) {
                    e = PV_EXP2((s[kb][qi][r] - mt) * sc);
                    lt += e;
                } else if (is_i) {
                    e = PV_EXP2((s[kb][qi][r] - mi) * sc);
                    li += e;
                }
                s[kb][qi][r] = e;
            }
        lt = pv_quad_sum(lt);
        li = pv_quad_sum(li);
        const float ft = (p.fusion ? p.fusion[0] : p.w_text) / lt, fi = (p.fusion ? p.fusion[1] : p.w_ip) / li;
#pragma unroll
        for (int s2 = 0; s2 < NKB / 2; ++s2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = (2 * s2) * 16 + fq * 4 + r, k1 = k0 + 16;
                pb[s2][qi][r] = (half_t)(s[2 * s2][qi][r] * (k0 < IP0 ? ft : fi));
                pb[s2][qi][r + 4] = (half_t)(s[2 * s2 + 1][qi][r] * (k1 < IP0 ? ft : fi));
            }
    }

    float4_t o[C::DVF][2];
#pragma unroll
    for (int f = 0; f < C::DVF; ++f)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) o[f][qi] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < NKB / 2; ++s2)
#pragma unroll
        for (int f = 0; f < C::DVF; ++f) {
            const half8_t a = vt_frag(sV, C::VS, s2 * 32, f * 16, fr, fq);
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) o[f][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[s2][qi], o[f][qi], 0, 0, 0);
        }

    half_t* O = reinterpret_cast<half_t*>(p.out) + (size_t)b * p.nq * p.ldo + h * D;
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        if (qrow[qi] < p.nq) {
#pragma unroll
            for (int f = 0; f < C::DVF; ++f) {
                const int dv = f * 16 + fq * 4;
                if (dv < D) {
                    half4_t ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)o[f][qi][r];
                    *reinterpret_cast<half4_t*>(O + (size_t)qrow[qi] * p.ldo + dv) = ov;
                }
            }
        }
    }
    }   // query tiles of this workgroup
}

// Which kernel a pv_attention launch takes: ONE rule, used by the launcher and by pv_attention_kernel_info (the host side tags its launches with the
// symbol rocprofv3 will show; it asks instead of restating this rule).
enum AttnKind { ATTN8 = 0, ATTN40_4_DMA, ATTN40_2_DMA, ATTN_4, ATTN_2 };
struct AttnChoice {
    AttnKind kind;
    int var8;          // ATTN8: the instantiated variant
    long wgs;          // workgroups of the launch
};
// the forms of EXPERIMENTS.md's round-5 table that are instantiated (the per-segment priority schemes measured there were removed again)
#define PV_A8_VARIANTS(X) X(0) X(1) X(9) X(13) X(33) X(49) X(73) X(201) X(225) X(241) X(481) X(497)
constexpr bool attn8_variant_built(int v) {
#define PV_A8_TEST(V) if (v == V) return true;
    PV_A8_VARIANTS(PV_A8_TEST)
#undef PV_A8_TEST
    return false;
}
static_assert(attn8_variant_built(PV_ATTN8_DEFAULT), "the default variant must be an instantiated one");

template <int D>
AttnChoice choose_attn(const pv_attn_params& p) {
    // four query fragments per wave where the accumulators fit two waves per SIMD (d = 40) and the launch still fills the chip
    static const int nq_env = getenv("PV_ATTN_NQ") ? atoi(getenv("PV_ATTN_NQ")) : 0;
    const long wg256 = (long)((p.nq + 255) / 256) * p.heads * p.batch, wg128 = (long)((p.nq + 127) / 128) * p.heads * p.batch;
    const bool four = D == 40 && (nq_env ? nq_env == 4 : wg256 >= 1024);
    if constexpr (D == 40) {
        static const bool no_dma = getenv("PV_ATTN_NO_DMA") != nullptr;      // A/B switch
        if (!no_dma && (size_t)p.nk * (size_t)(p.ldk > p.ldv ? p.ldk : p.ldv) * 2 < (1ull << 31)) {
            // 8-wave staggered form: one 512-query workgroup per CU; taken when the launch fills the chip with them
            // (read per call, not cached: tests run several forms in one process; launches are recorded once and replayed from graphs)
            const char* env8 = getenv("PV_ATTN8");                  // negative or not an instantiated form: the 4-wave kernels take the launch
            const char* env8min = getenv("PV_ATTN8_MIN");           // fewest 512-query workgroups a launch must have (default: one per CU)
            const int var8 = env8 ? atoi(env8) : PV_ATTN8_DEFAULT;
            const long wg512 = (long)((p.nq + 511) / 512) * p.heads * p.batch;
            if (attn8_variant_built(var8) && !p.causal && wg512 >= (env8min ? atol(env8min) : 256)) return {ATTN8, var8, wg512};
            return four ? AttnChoice{ATTN40_4_DMA, 0, wg256} : AttnChoice{ATTN40_2_DMA, 0, wg128};
        }
    }
    return four ? AttnChoice{ATTN_4, 0, wg256} : AttnChoice{ATTN_2, 0, wg128};
}

template <int D>
int launch_attn(const pv_attn_params& p, hipStream_t s) {
    using C = ACfg<D>;
    constexpr int smem1 = 64 * (C::KS + C::VS) * 2;
    const AttnChoice ch = choose_attn<D>(p);
    if constexpr (D == 40) {
        constexpr int smem3 = 4 * smem1;            // 56 KiB: above the 48-KiB default of dynamic LDS
        if (ch.kind == ATTN8) {
            void (*kern)(const pv_attn_params) = nullptr;
            switch (ch.var8) {
#define PV_A8_CASE(V) case V: kern = attn8_kernel<V>; break;
                PV_A8_VARIANTS(PV_A8_CASE)
#undef PV_A8_CASE
                default: return (int)hipErrorInvalidValue;          // unreachable: choose_attn names instantiated forms only
            }
            static bool attr8_set[64][512] = {};
            int dev8 = 0;
            (void)hipGetDevice(&dev8);
            if (!attr8_set[dev8 & 63][ch.var8 & 511]) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem3);
                if (e != hipSuccess) return (int)e;
                attr8_set[dev8 & 63][ch.var8 & 511] = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)ch.wgs), dim3(512), smem3, s, p);
            return PV_CHECK_LAUNCH();
        }
        if (ch.kind == ATTN40_4_DMA || ch.kind == ATTN40_2_DMA) {
            static bool attr_set_dev[64] = {};
            int dev_id = 0;
            (void)hipGetDevice(&dev_id);
            if (!attr_set_dev[dev_id & 63]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<40, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem3);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<40, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem3);
                if (e != hipSuccess) return (int)e;
                attr_set_dev[dev_id & 63] = true;
            }
            if (ch.kind == ATTN40_4_DMA) hipLaunchKernelGGL((attn_kernel<40, 4, true>), dim3((unsigned)ch.wgs), dim3(256), smem3, s, p);
            else hipLaunchKernelGGL((attn_kernel<40, 2, true>), dim3((unsigned)ch.wgs), dim3(256), smem3, s, p);
            return PV_CHECK_LAUNCH();
        }
    }
    if (ch.kind == ATTN_4) hipLaunchKernelGGL((attn_kernel<D, D == 40 ? 4 : 2, false>), dim3((unsigned)ch.wgs), dim3(256), smem1, s, p);
    else hipLaunchKernelGGL((attn_kernel<D, 2, false>), dim3((unsigned)ch.wgs), dim3(256), smem1, s, p);
    return PV_CHECK_LAUNCH();
}

template <int D>
int attn_info(const pv_attn_params& p, char* name, int name_len, int64_t* workgroups) {
    const AttnChoice ch = choose_attn<D>(p);
    int n;
    switch (ch.kind) {
        case ATTN8: n = snprintf(name, (size_t)name_len, "attn8_kernel<%d>", ch.var8); break;
        case ATTN40_4_DMA: n = snprintf(name, (size_t)name_len, "attn_kernel<40, 4, true>"); break;
        case ATTN40_2_DMA: n = snprintf(name, (size_t)name_len, "attn_kernel<40, 2, true>"); break;
        case ATTN_4: n = snprintf(name, (size_t)name_len, "attn_kernel<%d, 4, false>", D); break;
        default: n = snprintf(name, (size_t)name_len, "attn_kernel<%d, 2, false>", D); break;
    }
    if (n < 0 || n >= name_len) return (int)hipErrorInvalidValue;
    if (workgroups) *workgroups = ch.wgs;
    return 0;
}

template <int D>
int launch_xattn(const pv_xattn_params& p, hipStream_t s) {
    using C = ACfg<D>;
    constexpr int smem = XKEYS * (C::KS + C::VS) * 2;
    static bool attr_set_dev[64] = {};   // per device: one process may drive several GPUs
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    if (smem > 48 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_kernel<D, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_kernel<D, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nqt = (p.nq + 127) / 128;
    const long wgs = (long)nqt * p.heads * p.batch;
    int qt_per_wg = (int)(wgs / 1024);                       // keep >= ~1024 workgroups (4 per CU) in the launch
    qt_per_wg = qt_per_wg < 1 ? 1 : (qt_per_wg > 4 ? 4 : qt_per_wg);
    const int nqg = (nqt + qt_per_wg - 1) / qt_per_wg;
    if (qt_per_wg > 1) hipLaunchKernelGGL((xattn_kernel<D, true>), dim3(nqg * p.heads * p.batch), dim3(256), smem, s, p, qt_per_wg);
    else hipLaunchKernelGGL((xattn_kernel<D, false>), dim3(nqg * p.heads * p.batch), dim3(256), smem, s, p, 1);
    return PV_CHECK_LAUNCH();
}

}  // namespace

extern "C" int pv_attention(const pv_attn_params* p, void* stream) {
    if (!p->q || !p->k || !p->v || !p->out || p->batch <= 0 || p->heads <= 0 || p->nq <= 0 || p->nk <= 0 || (p->ldq % 8) ||
        (p->ldk % 8) || (p->ldv % 8) || (p->ldo % 4))
        return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    switch (p->d) {
        case 40: return launch_attn<40>(*p, s);
        case 64: return launch_attn<64>(*p, s);
        case 80: return launch_attn<80>(*p, s);
        case 160: return launch_attn<160>(*p, s);
        default: return (int)hipErrorInvalidValue;
    }
}

// The kernel pv_attention would launch for this parameter block (symbol as rocprofv3 prints it, workgroup count): the launcher's own rule.  No HIP call.
extern "C" int pv_attention_kernel_info(const pv_attn_params* p, char* name, int32_t name_len, int64_t* workgroups) {
    if (!p || !name || name_len <= 0 || p->batch <= 0 || p->heads <= 0 || p->nq <= 0 || p->nk <= 0) return (int)hipErrorInvalidValue;
    switch (p->d) {
        case 40: return attn_info<40>(*p, name, name_len, workgroups);
        case 64: return attn_info<64>(*p, name, name_len, workgroups);
        case 80: return attn_info<80>(*p, name, name_len, workgroups);
        case 160: return attn_info<160>(*p, name, name_len, workgroups);
        default: return (int)hipErrorInvalidValue;
    }
}

extern "C" int pv_cross_attention(const pv_xattn_params* p, void* stream) {
    if (!p->q || !p->kt || !p->vt || !p->kip || !p->vip || !p->out || p->batch <= 0 || p->heads <= 0 || p->nq <= 0 || p->nt <= 0 ||
        p->nt > IP0 || p->nip <= 0 || p->nip > XKEYS - IP0 || (p->ldq % 8) || (p->ldkt % 8) || (p->ldvt % 8) || (p->ldkip % 8) ||
        (p->ldvip % 8) || (p->ldo % 4))
        return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    switch (p->d) {
        case 40: return launch_xattn<40>(*p, s);
        case 80: return launch_xattn<80>(*p, s);
        case 160: return launch_xattn<160>(*p, s);
        default: return (int)hipErrorInvalidValue;
    }
}
