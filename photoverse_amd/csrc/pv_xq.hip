// pv_cross_attention_lnq: norm2 -> to_q -> dual-branch cross attention of the C = 1280 / d = 160 attn2 layers in ONE launch, head-parallel.
//
//   ctx[b, rows, h*d : (h+1)*d] = w_t * softmax(q Kt^T / sqrt(d)) Vt + w_i * softmax(q Kip^T / sqrt(d)) Vip,   q = to_q(LayerNorm(hs))[:, h*d : (h+1)*d]
//
// (/root/reference/models/attention_processor.py:297 to_q, :307-322 text SDPA, :392-420 image-token SDPA + fusion; norm2 of the [EXT]
// BasicTransformerBlock in front).  to_out + bias + residual (:423) follows as one pv_gemm_conv launch: the branch is TWO launches instead of
// four (LayerNorm, to_q GEMM, pv_cross_attention, to_out GEMM).  The one-launch kernel of pv_xfused.hip keeps a wave's rows in registers
// for the whole chain; at C = 1280 a 16-row slab is 160 registers and 16 x 256 rows make 64 such workgroups.  Here a
// workgroup is (128 query rows, ONE head): 32 x 8 = 256 workgroups on the 16 x 16 level, one per CU.
//
//   GEMM   Q^T[n][row] = Wq'[h*160 + n][:] . X^T, K = C in 64-deep steps: the head's 160 weight rows stream through a four-stage LDS ring
//          (LDS-DMA, 8-row x 128-B pieces, chunk ^= row & 7, counted vmcnt, one raw s_barrier per step); X fragments (MFMA-B: lane = row,
//          8 consecutive channels) are read from the same ring.
//   roles  144 KiB of LDS = one workgroup per CU, so nothing overlaps a wave's stalls but its own workgroup: waves 0-3 COMPUTE (one per
//          SIMD), waves 4-7 only LOAD - they issue every LDS-DMA of the kernel: per step 20 weight pieces and the 16 pieces of the
//          workgroup's rows (full 128-B lines; the first form read the rows as MFMA-shaped fragments straight from global memory - 16
//          64-B segments per wave-instruction - and spent 2 240 - 2 860 cycles per step against 640 of MFMA issue,
//          profiles/r04_xlnq_stamps.txt), and, once the GEMM has released the ring, the K / V images.
//   norm2  folded algebraically, so the GEMM runs on the RAW rows:  to_q(LN(x)) = rstd * (Wq' . x - mean * rowsum(Wq')) + Wq . beta,
//          Wq' = Wq diag(gamma).  sum(x) and sum(x^2) are accumulated from the very fragments the MFMAs consume (v_dot2_f32_f16); the
//          normalised activations are never rounded to fp16 (the four-launch path rounds them once).
//   SDPA   pv_attn.hip's dual-branch scheme: S^T = K . Q^T with the accumulator layout of the GEMM used directly as the B operand (lane =
//          row, 8 slots = 4 registers of fragment 2s + 4 of fragment 2s+1).  The weight ROWS of the head are streamed in a permuted order
//          (PHI below: the per-lane source row of the LDS-DMA) chosen so that those 8 slots are 8 CONSECUTIVE features: K and V then stay in
//          their natural layout and reach LDS by LDS-DMA straight from the projected text / image-token rows (zero rows = out-of-range
//          offsets), in flight under the GEMM.  Two softmaxes in registers, O^T = V^T . P^T by ds_read_b64_tr_b16.
#include "pv_common.h"

namespace {

constexpr int D = 160;                       // head dim
constexpr int NFQ = D / 16;                  // 10 feature fragments of the head
constexpr int KSTEPS = D / 32;               // 5 contraction steps of S^T = K . Q^T
constexpr int XKEYS = 96, IP0 = 80;          // K / V image rows: text keys [0, nt), image-token keys [80, 80 + nip), zeros elsewhere
constexpr int NKB = XKEYS / 16;
constexpr int KS = D;                        // K image rows are unpadded (the image is one contiguous LDS-DMA target); 16-B chunk ^= F[(row >> 2) & 3]
constexpr int VS = 176;                      // V image row stride in halfs (pv_attn.hip ACfg<160>::VS: eight key rows on eight different 32-B slots)
constexpr int NST = 4;                       // ring stages
constexpr int W_BYTES = D * 128;             // weight part of a stage: 160 rows x 64 k x 2 B = 20 KiB
constexpr int X_BYTES = 128 * 128;           // activation part: the workgroup's 128 rows x 64 k x 2 B = 16 KiB
constexpr int W_STAGE = W_BYTES + X_BYTES;   // 36 KiB
constexpr int NLOAD = 4;                     // loader waves (waves 4-7): one per SIMD beside a computing wave
constexpr int LPW = D / 8 / NLOAD, LPX = 128 / 8 / NLOAD;   // 8-row LDS-DMA pieces per loader and stage: 5 weight + 4 activation
constexpr int LP = LPW + LPX;
constexpr int KV_BYTES = XKEYS * KS * 2 + XKEYS * VS * 2;   // 30 + 33 KiB: the K / V images take over the ring once the GEMM is done
constexpr int SMEM_BYTES = NST * W_STAGE;    // 144 KiB: one workgroup per CU
static_assert(KV_BYTES <= SMEM_BYTES, "K / V images alias the ring");
constexpr int K_INSTR = XKEYS * (KS / 8) / 64;      // 30 LDS-DMA wave-instructions fill the K image, 33 the V image (pad chunks read zeros)
constexpr int V_INSTR = XKEYS * (VS / 8) / 64;
static_assert(XKEYS * (KS / 8) % 64 == 0 && XKEYS * (VS / 8) % 64 == 0, "K / V images are whole numbers of 1-KiB LDS-DMA writes");

// chunk swizzle of the 320-B-row K image: F[(row >> 2) & 3], F = {0, 2, 3, 1} on the low two bits of the chunk index (a ds_read_b128 lane group
// touches rows r, r + 4, r + 8, r + 12 of equal r & 3 = equal 64-B quarter of the bank row: F sends them to four different 16-B slots)
__device__ __forceinline__ int kswz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }
// weight row streamed at position n of the head's 160 (n = 16 f + 4 g + r: accumulator register r of lane group g of fragment f):
// PHI(n) = 32 (f >> 1) + 8 g + 4 (f & 1) + r, so that B-operand slot (k-step s, lane group g, j) = accumulators (2s + (j >> 2), g, j & 3) = feature 32 s + 8 g + j
__device__ __forceinline__ int phi(int n) {
    const int f = n >> 4, g = (n >> 2) & 3, r = n & 3;
    return 32 * (f >> 1) + 8 * g + 4 * (f & 1) + r;
}

__device__ __forceinline__ half8_t zero8() { return half8_t{0, 0, 0, 0, 0, 0, 0, 0}; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ half8_t vt_frag(const half_t* sV, int key0, int dv0, int fr, int fq) {
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

__device__ __forceinline__ float dot2(half2_t a, half2_t b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }

struct I0 { static constexpr int value = 0; };
struct I1 { static constexpr int value = 1; };
struct I2 { static constexpr int value = 2; };

template <int HPW>       // heads per 160-feature block: 1 (d = 160, C = 1280) or 2 (d = 80, C = 640)
__global__ __launch_bounds__(512, 2) void xattn_lnq_kernel(const pv_xattn_lnq_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    half_t* sK = reinterpret_cast<half_t*>(smem);            // (after the GEMM)
    half_t* sV = sK + XKEYS * KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();
    const int fr = lane & 15, fq = lane >> 4;
    const int C = p.heads * p.d;
    const int nblk = C / D;                                  // 160-feature blocks = workgroups per row tile (h below is the BLOCK index)
    const int nqt = (p.nq + 127) / 128;
    const int rid = (int)blockIdx.x;
    // heads fastest: the eight workgroups that read the same 128 rows run together (the rows come from L2 once they have been touched)
    const int h = rid % nblk, qt = (rid / nblk) % nqt, b = rid / (nblk * nqt);
    const int nk = C / 64;

    // ---- the head's K / V images (conditioning only) by LDS-DMA: image chunk idx = 64 j + lane of wave-instruction j -> (row, position); the
    // source is the row's chunk (position ^ swizzle) of the text / image-token K (V) rows, an out-of-range offset (zeros) for padding ----
    auto issue_kv = [&](int lw) {                   // lw: loader index 0 .. 3
        constexpr unsigned OOB = 0x80000000u;
        const __amdgpu_buffer_rsrc_t rkt = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.kt), 0, (int)(((size_t)p.batch * p.nt - 1) * p.ldkt * 2 + (size_t)C * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rvt = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.vt), 0, (int)(((size_t)p.batch * p.nt - 1) * p.ldvt * 2 + (size_t)C * 2), 0x00020000);
        for (int j = lw; j < K_INSTR; j += NLOAD) {
            const int idx = j * 64 + lane, r = idx / (KS / 8), c = idx - r * (KS / 8);
            const int sc = (c & ~3) | ((c & 3) ^ kswz(r));
            const bool is_t = r < p.nt, is_i = r >= IP0 && r < IP0 + p.nip;
            const unsigned off_t = (unsigned)(((size_t)b * p.nt + r) * p.ldkt + h * D + sc * 8) * 2u;
            const unsigned off_i = (unsigned)(((size_t)b * p.nip + (r - IP0)) * p.ldkip + h * D + sc * 8) * 2u;
            // the image-token rows ([80, 80 + nip): one or two of the 30 wave-instructions) are patched from registers behind the DMA, which has
            // zero-filled their slots: a second DMA over the same 1 KiB would zero the text rows that share it
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rkt, PV_LDS_PTR(reinterpret_cast<char*>(sK) + j * 1024), 16, (int)(is_t ? off_t : OOB), 0, 0, 0);
            if (__builtin_amdgcn_ballot_w64(is_i)) {
                if (is_i) {
                    const half8_t v = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(p.kip) + off_i);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    *reinterpret_cast<half8_t*>(reinterpret_cast<char*>(sK) + idx * 16) = v;
                }
            }
        }
        for (int j = lw; j < V_INSTR; j += NLOAD) {
            const int idx = j * 64 + lane, r = idx / (VS / 8), c = idx - r * (VS / 8);
            const bool in_row = c < D / 8;
            const bool is_t = in_row && r < p.nt, is_i = in_row && r >= IP0 && r < IP0 + p.nip;
            const unsigned off_t = (unsigned)(((size_t)b * p.nt + r) * p.ldvt + h * D + c * 8) * 2u;
            const unsigned off_i = (unsigned)(((size_t)b * p.nip + (r - IP0)) * p.ldvip + h * D + c * 8) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rvt, PV_LDS_PTR(reinterpret_cast<char*>(sV) + j * 1024), 16, (int)(is_t ? off_t : OOB), 0, 0, 0);
            if (__builtin_amdgcn_ballot_w64(is_i)) {
                if (is_i) {
                    const half8_t v = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(p.vip) + off_i);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    *reinterpret_cast<half8_t*>(reinterpret_cast<char*>(sV) + idx * 16) = v;
                }
            }
        }
    };

    // ---- GEMM: Q^T = Wq'[head rows] . X^T on the raw rows ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wq), 0, C * C * 2, 0x00020000);
    const int lrow = lane >> 3;
    const bool loader = wave >= 4;
    const int lw = max(wave - 4, 0);
    // stage kt: weight pieces j = lw + 4 i (i < 5): LDS rows 8 j + lrow hold weight rows h*D + PHI(8 j + lrow); activation pieces j = lw + 4 i (i < 4):
    // the workgroup's rows 8 j + lrow (rows past the sample's nq: out-of-range offset -> zeros); 16-B chunk (lane & 7) ^ lrow of the 64-deep slab
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.hs), 0, (int)(((size_t)p.batch * p.nq - 1) * p.ld_hs * 2 + (size_t)C * 2), 0x00020000);
    unsigned w_off[LPW], x_off[LPX];
#pragma unroll
    for (int i = 0; i < LPW; ++i)
        w_off[i] = (unsigned)(h * D + phi((lw + NLOAD * i) * 8 + lrow)) * (unsigned)(C * 2) + (unsigned)(((lane & 7) ^ lrow) * 16);
#pragma unroll
    for (int i = 0; i < LPX; ++i) {
        const int r = qt * 128 + (lw + NLOAD * i) * 8 + lrow;
        x_off[i] = r < p.nq ? (unsigned)(((size_t)b * p.nq + r) * p.ld_hs * 2) + (unsigned)(((lane & 7) ^ lrow) * 16) : 0x80000000u;
    }
    auto issue_stage = [&](int kt) {
        char* dst = sW + (kt & (NST - 1)) * W_STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, PV_LDS_PTR(dst + (lw + NLOAD * i) * 8 * 128), 16, (int)(w_off[i] + (unsigned)kt * 128u), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < LPX; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, PV_LDS_PTR(dst + W_BYTES + (lw + NLOAD * i) * 8 * 128), 16,
                                                     (int)(x_off[i] == 0x80000000u ? x_off[i] : x_off[i] + (unsigned)kt * 128u), 0, 0, 0);
    };
    if (loader) {
        // ---- loader waves: stages 0, 1, 2, then per step stage kt+3 into the buffer the computing waves left at the barrier; "stage kt landed" =
        // at most the two younger stages outstanding.  After the GEMM: the K / V images into the (now free) ring. ----
        issue_stage(0);
        if (nk > 1) issue_stage(1);
        if (nk > 2) issue_stage(2);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 2 < nk) wait_vmcnt<2 * LP>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 3 < nk) issue_stage(kt + 3);
        }
        __builtin_amdgcn_s_barrier();                         // every computing wave has read the last stage: the ring is free
        asm volatile("" ::: "memory");
        issue_kv(lw);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                         // K / V images landed
        return;
    }
    int qrow[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) qrow[qi] = qt * 128 + wave * 32 + qi * 16 + fr;
    float4_t qacc[NFQ][2];
#pragma unroll
    for (int f = 0; f < NFQ; ++f)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) qacc[f][qi] = float4_t{0.f, 0.f, 0.f, 0.f};
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const half2_t one2 = half2_t{(half_t)1.0f, (half_t)1.0f};
    // computing waves: no vector-memory operation in the loop; behind the barrier stage kt has landed and stage kt-1 (refilled next) is free
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* sw = sW + (kt & (NST - 1)) * W_STAGE;
        const char* sx = sw + W_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t xb[2];
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                const int row = wave * 32 + qi * 16 + fr;
                xb[qi] = *reinterpret_cast<const half8_t*>(sx + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int f = 0; f < NFQ; ++f) {
                const int row = f * 16 + fr;
                const half8_t a = *reinterpret_cast<const half8_t*>(sw + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) qacc[f][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[qi], qacc[f][qi], 0, 0, 0);
            }
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const half2_t x2 = half2_t{xb[qi][2 * j], xb[qi][2 * j + 1]};
                    s1[qi] = dot2(x2, one2, s1[qi]);
                    s2[qi] = dot2(x2, x2, s2[qi]);
                }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // ring free -> the loaders fill in the K / V images while the queries are finalised below
    asm volatile("" ::: "memory");

    // ---- norm2 fold, query bias, B operand of the score product ----
    half8_t qf[2][KSTEPS];
    {
        const int nbase = h * D + fq * 8;                        // + 32 (f >> 1) + 4 (f & 1): the four features of accumulator fragment f (PHI)
        float mean[2], rstd[2];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const float a = pv_quad_sum(s1[qi]), q2 = pv_quad_sum(s2[qi]);
            mean[qi] = p.ln ? a / (float)C : 0.f;
            const float var = fmaxf(q2 / (float)C - mean[qi] * mean[qi], 0.f);
            rstd[qi] = p.ln ? rsqrtf(var + p.ln_eps) : 1.f;
        }
#pragma unroll
        for (int f = 0; f < NFQ; ++f) {
            const int nf = nbase + 32 * (f >> 1) + 4 * (f & 1);
            const float4_t cs = p.ln ? *reinterpret_cast<const float4_t*>(p.wq_rowsum + nf) : float4_t{0.f, 0.f, 0.f, 0.f};
            const float4_t qb = p.q_bias ? *reinterpret_cast<const float4_t*>(p.q_bias + nf) : float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int r = 0; r < 4; ++r) qf[qi][f >> 1][(f & 1) * 4 + r] = (half_t)(rstd[qi] * (qacc[f][qi][r] - mean[qi] * cs[r]) + qb[r]);
        }
    }

    __builtin_amdgcn_s_barrier();                             // K / V images landed (the loaders waited for their DMAs in front of this barrier)
    asm volatile("" ::: "memory");

    // to_v_ip_norm (attention_processor.py:397): ||Vip[b, p, head, :]||_2, once per (b, head)
    constexpr int DH = D / HPW;                              // head dim
    if (p.vnorm && qt == 0 && tid < p.nip * HPW) {
        const int hh = tid / p.nip, ip = tid - hh * p.nip;
        float a = 0.f;
        for (int d = 0; d < DH; ++d) {
            const float v = (float)sV[(IP0 + ip) * VS + hh * DH + d];
            a += v * v;
        }
        p.vnorm[((size_t)b * p.heads + h * HPW + hh) * p.nip + ip] = sqrtf(a);
    }

    // ---- per head: S^T = K . Q^T, two softmaxes (pv_attn.hip: xattn_kernel); then O^T = V^T . P^T ----
    // HPW = 2: head hh owns features [80 hh, 80 hh + 80) of the block = k-steps {0, 1, lower half of 2} / {upper half of 2, 3, 4}: the shared
    // step is taken with the other head's lane groups zeroed in the B operand (lane group g of step 2 = features 64 + 8 g .. + 7)
    const float sc = rsqrtf((float)DH) * 1.4426950408889634f;
    half8_t pb[HPW][NKB / 2][2];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        float4_t s[NKB][2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) s[kb][qi] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if (HPW == 2 && ((hh == 0 && ks > 2) || (hh == 1 && ks < 2))) continue;
            half8_t qm[2] = {qf[0][ks], qf[1][ks]};
            if (HPW == 2 && ks == 2) {
                const bool mine = hh == 0 ? fq < 2 : fq >= 2;
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) qm[qi] = mine ? qm[qi] : zero8();
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const half8_t a = *reinterpret_cast<const half8_t*>(sK + (kb * 16 + fr) * KS + (ks * 4 + (fq ^ kswz(fr))) * 8);
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qm[qi], s[kb][qi], 0, 0, 0);
            }
        }
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
            const float ft = (p.fusion ? p.fusion[0] : p.w_text) / lt, fi = p.nip ? (p.fusion ? p.fusion[1] : p.w_ip) / li : 0.f;
#pragma unroll
            for (int s2i = 0; s2i < NKB / 2; ++s2i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k0 = (2 * s2i) * 16 + fq * 4 + r, k1 = k0 + 16;
                    pb[hh][s2i][qi][r] = (half_t)(s[2 * s2i][qi][r] * (k0 < IP0 ? ft : fi));
                    pb[hh][s2i][qi][r + 4] = (half_t)(s[2 * s2i + 1][qi][r] * (k1 < IP0 ? ft : fi));
                }
        }
    }
    float4_t o[NFQ][2];
#pragma unroll
    for (int f = 0; f < NFQ; ++f)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) o[f][qi] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2i = 0; s2i < NKB / 2; ++s2i)
#pragma unroll
        for (int f = 0; f < NFQ; ++f) {
            const half8_t a = vt_frag(sV, s2i * 32, f * 16, fr, fq);
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) o[f][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[(f * 16) / DH][s2i][qi], o[f][qi], 0, 0, 0);
        }
    half_t* O = reinterpret_cast<half_t*>(p.out) + (size_t)b * p.nq * p.ldo + h * D;     // the block's 160 features = HPW consecutive heads
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        if (qrow[qi] < p.nq) {
#pragma unroll
            for (int f = 0; f < NFQ; ++f) {
                half4_t ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)o[f][qi][r];
                *reinterpret_cast<half4_t*>(O + (size_t)qrow[qi] * p.ldo + f * 16 + fq * 4) = ov;
            }
        }
    }
}

}  // namespace

extern "C" int pv_cross_attention_lnq(const pv_xattn_lnq_params* p, void* stream) {
    if (!p || !p->hs || !p->wq || !p->kt || !p->vt || !p->out || (p->d != D && p->d != D / 2) || p->heads <= 0 || p->batch <= 0 || p->nq <= 0 || p->nt <= 0 || p->nt > IP0 ||
        p->nip < 0 || p->nip > XKEYS - IP0 || (p->nip > 0 && (!p->kip || !p->vip)) || (p->ln && !p->wq_rowsum) || (p->ld_hs % 8) || (p->ldo % 4) ||
        (p->ldkt % 4) || (p->ldvt % 8) || (p->nip > 0 && ((p->ldkip % 4) || (p->ldvip % 8))))
        return (int)hipErrorInvalidValue;
    const size_t C = (size_t)p->heads * p->d;
    if (C % D) return (int)hipErrorInvalidValue;
    if (C * C * 2 >= (1ull << 31)) return (int)hipErrorInvalidValue;
    // the kernel addresses rows through 32-bit buffer offsets (0x80000000 is its out-of-range sentinel): every extent must stay below 2 GiB
    const size_t rows_m1 = (size_t)p->batch * p->nq - 1;
    if (rows_m1 * (size_t)p->ld_hs * 2 + C * 2 >= (1ull << 31) || rows_m1 * (size_t)p->ldo * 2 + C * 2 >= (1ull << 31) ||
        ((size_t)p->batch * p->nt - 1) * (size_t)(p->ldkt > p->ldvt ? p->ldkt : p->ldvt) * 2 + C * 2 >= (1ull << 31) ||
        (p->nip > 0 && ((size_t)p->batch * p->nip - 1) * (size_t)(p->ldkip > p->ldvip ? p->ldkip : p->ldvip) * 2 + C * 2 >= (1ull << 31)))
        return (int)hipErrorInvalidValue;
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_lnq_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_lnq_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nqt = (p->nq + 127) / 128;
    const dim3 grid((unsigned)(nqt * (C / D) * p->batch));
    if (p->d == D) hipLaunchKernelGGL(xattn_lnq_kernel<1>, grid, dim3(512), SMEM_BYTES, reinterpret_cast<hipStream_t>(stream), *p);
    else hipLaunchKernelGGL(xattn_lnq_kernel<2>, grid, dim3(512), SMEM_BYTES, reinterpret_cast<hipStream_t>(stream), *p);
    return PV_CHECK_LAUNCH();
}
