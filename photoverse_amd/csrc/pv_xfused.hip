// pv_cross_attention_fused: the whole attn2 branch of a BasicTransformerBlock as ONE kernel for gfx950 (CDNA4)
//
//   out = hs + to_out( w_t * softmax(q Kt^T / sqrt(d)) Vt + w_i * softmax(q Kip^T / sqrt(d)) Vip ) + bias,   q = to_q(LayerNorm(hs))
//
// i.e. norm2 -> PhotoVerseAttnProcessor2_0.__call__ (/root/reference/models/attention_processor.py:297 to_q, :307-322 text
// SDPA, :392-420 image-token SDPA + fusion, :423 to_out[0]) -> residual add of the transformer block.  It replaces four
// launches (LayerNorm, to_q GEMM, dual-branch attention, to_out GEMM) and six passes of a (B*N, C) tensor through HBM by one
// read of hs and one write of out.  Built for the C = 320 / d = 40 layers (64x64 and larger levels: 63 % of attn2's launches'
// time at 512x512), where M = B*N is large and the GEMMs are too short-K to run well on their own; the C = 640 / d = 80
// instantiation (one head per 80-feature group) runs 64-row workgroups so that the 16 x 1024 rows of a 32x32 level still fill the chip
// (74 us vs 80 us for the four launches; 128-row workgroups = 128 of 256 CUs: 99 us).  C = 1280 (16 x 256 rows) stays on four launches.
//
// Structure: one workgroup = 128 query rows x ALL heads, 4 waves, 80 KiB of LDS -> two workgroups per CU.  A wave owns 32 query
// rows (two 16-column MFMA operands; NQ = 1: 16 rows) for the whole chain, so no activation ever leaves the register file:
//
//   phase 0  X^T (B operand of v_mfma_f32_16x16x32_f16: lane = query, 8 consecutive channels per k-group) is loaded straight
//            into registers; LayerNorm statistics by v_dot2_f32_f16 + a 4-lane swap reduction; normalised in place (its affine
//            part is folded into Wq / a query bias by the caller).
//   phase 1  Q^T[n][q] = Wq[n][:] . X^T : Wq streams through a 3-stage LDS ring (80 rows x 64 k per stage, LDS-DMA, counted
//            vmcnt, one raw s_barrier per stage); the accumulator layout (lane = query, 4 consecutive features per k-group)
//            IS the next product's B operand once converted to fp16 - with a permuted contraction order, which the K image
//            and the columns of Wo are pre-permuted to match (pv_xattn_pack_kv / pv_xattn_fused_wo_slot).
//   phase 2  per 80-feature group (= 2 heads of 40): S^T = K.Q^T (K image rows from LDS), two independent softmaxes in
//            registers, O^T = V^T.P^T (V^T fragments by ds_read_b64_tr_b16); the K/V images of the (sample, group) are LDS-DMA
//            double-buffered (group 0 is prefetched at kernel start).  The 2.5-fragment head boundary needs no padding:
//            fragment 2 of a group is computed for both heads and merged by lane group.
//   phase 3  out^T[n][q] = Wo'[n][:] . ctx^T through the same kind of ring (prefetched during the last group); the epilogue adds
//            bias + residual (requested at the start of each 80-column chunk) and stores fp16.
//
// Measured (MI355X, B=16, N=4096): 78 us vs 118 us for the four launches; 430 TFLOP/s = 17 % of the dense fp16 MFMA peak.  Where the
// rest goes (tools/diag/xfused_stamps.py): every workgroup of the single wave of 512 runs the same phases at the same time, so the
// HBM phases (X load, output store) and the MFMA phases do not overlap chip-wide; phase 2 issues ~3300 VALU instructions per wave
// (two softmaxes over 96 keys per head) against 336 MFMAs.  An 8-wave / one-workgroup-per-CU variant (features split over two wave
// columns, context exchanged through LDS) measured 82 us: a workgroup barrier costs ~400 cycles when all waves of a CU run in lockstep.
#include "pv_common.h"

namespace {

constexpr int XK = 96;     // rows of a K / V image: text keys [0, nt), image-token keys [XIP0, XIP0 + nip), zeros elsewhere
constexpr int XIP0 = 80;
constexpr int GF = 80;     // features per group = 5 MFMA fragments of 16 (two heads of 40)
constexpr int KROW = 128;  // bytes per K image row: 64 contraction slots (32 + 32, the second half mostly zero)

// contraction slot kappa (0..63) of head parity hh -> head-local feature, or -1 (zero slot).  k-step 0 pairs the two whole
// fragments of the head (h0: group fragments 0,1; h1: 3,4); k-step 1 is fragment 2, shared by the two heads (h0 owns its rows
// 0-7 = lane groups 0,1; h1 rows 8-15 = lane groups 2,3).  Within a k-step, lane group g holds slots 8g..8g+7 = rows 4g..4g+3
// of the first fragment followed by rows 4g..4g+3 of the second.
__host__ __device__ inline int kslot_feature40(int hh, int kappa) {
    const int s = kappa >> 5, g = (kappa >> 3) & 3, jj = kappa & 7;
    if (s == 0) {
        const int base = hh == 0 ? 0 : 8;
        return jj < 4 ? base + 4 * g + jj : base + 16 + 4 * g + (jj - 4);
    }
    if (jj >= 4) return -1;
    const int gf = 32 + 4 * g + jj;   // group feature inside fragment 2
    if (hh == 0) return gf < 40 ? gf : -1;
    return gf >= 40 ? gf - 40 : -1;
}

// d = 80 (C = 640): one head = one 80-feature group = five whole fragments.  Slot kappa (0..127; 128 slots = 256-B K image rows, the fourth
// k-step never read) of k-step s = kappa >> 5: s < 2 pairs the fragments 2s, 2s+1; s == 2 is fragment 4 alone (upper half zero).
__host__ __device__ inline int kslot_feature80(int kappa) {
    const int s = kappa >> 5, g = (kappa >> 3) & 3, jj = kappa & 7;
    if (s < 2) return jj < 4 ? 32 * s + 4 * g + jj : 32 * s + 16 + 4 * g + (jj - 4);
    if (s == 2 && jj < 4) return 64 + 4 * g + jj;
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// pv_xattn_pack_kv: builds, once per conditioning, the exact LDS images the fused kernel DMA-copies:
//   d = 40: kimg [B][heads][96][64] fp16, slot order of kslot_feature40, 16-B chunk c of row r stored at position c ^ (r & 7)
//   d = 80: kimg [B][heads][96][128] fp16, slot order of kslot_feature80, chunk c of row r at position c ^ (r & 15) (256-B rows span all banks)
//   vimg [B][C/80][96][80] fp16 (natural column order)
//   vnorm[B][heads][nip]   = ||Vip[b, p, h, :]||_2   (attention_processor.py:397)
__global__ __launch_bounds__(256) void xattn_pack_kv_kernel(const half_t* kt, const half_t* vt, int ldkt, int ldvt, const half_t* kip,
                                                            const half_t* vip, int ldkip, int ldvip, half_t* kimg, half_t* vimg,
                                                            float* vnorm, int heads, int nt, int nip, int d) {
    const int b = blockIdx.x / XK, key = blockIdx.x % XK;
    const int C = heads * d;
    const half_t *krow = nullptr, *vrow = nullptr;
    if (key < nt) {
        krow = kt + (size_t)(b * nt + key) * ldkt;
        vrow = vt + (size_t)(b * nt + key) * ldvt;
    } else if (key >= XIP0 && key < XIP0 + nip) {
        krow = kip + (size_t)(b * nip + key - XIP0) * ldkip;
        vrow = vip + (size_t)(b * nip + key - XIP0) * ldvip;
    }
    const int tid = threadIdx.x;
    if (d == 40) {
        for (int i = tid; i < heads * 8; i += 256) {        // K: one 16-B chunk (8 slots) per item
            const int h = i >> 3, c = i & 7;
            half8_t v = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (krow) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int f = kslot_feature40(h & 1, c * 8 + jj);
                    if (f >= 0) v[jj] = krow[h * 40 + f];
                }
            }
            *reinterpret_cast<half8_t*>(kimg + ((size_t)(b * heads + h) * XK + key) * 64 + ((c ^ (key & 7)) << 3)) = v;
        }
    } else {
        for (int i = tid; i < heads * 16; i += 256) {       // d = 80: sixteen chunks per 256-B row
            const int h = i >> 4, c = i & 15;
            half8_t v = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (krow) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int f = kslot_feature80(c * 8 + jj);
                    if (f >= 0) v[jj] = krow[h * 80 + f];
                }
            }
            *reinterpret_cast<half8_t*>(kimg + ((size_t)(b * heads + h) * XK + key) * 128 + ((c ^ (key & 15)) << 3)) = v;
        }
    }
    for (int i = tid; i < C / 8; i += 256) {            // V: 16-B chunks, natural order
        const int grp = (i * 8) / GF, j = i * 8 - grp * GF;
        half8_t v = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (vrow) v = *reinterpret_cast<const half8_t*>(vrow + i * 8);
        *reinterpret_cast<half8_t*>(vimg + ((size_t)(b * (C / GF) + grp) * XK + key) * GF + j) = v;
    }
    if (vnorm && key >= XIP0 && key < XIP0 + nip && tid < heads) {
        float a = 0.f;
        for (int j = 0; j < d; ++j) {
            const float v = (float)vrow[tid * d + j];
            a += v * v;
        }
        vnorm[((size_t)b * heads + tid) * nip + (key - XIP0)] = sqrtf(a);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ half8_t ld_frag128(const char* base, int row, int chunk) {   // 128-B rows, chunk ^= row & 7
    return *reinterpret_cast<const half8_t*>(base + row * 128 + ((chunk ^ (row & 7)) << 4));
}
__device__ __forceinline__ half8_t ld_frag256(const char* base, int row, int chunk) {   // 256-B rows (d = 80 K image), chunk ^= row & 15
    return *reinterpret_cast<const half8_t*>(base + row * 256 + ((chunk ^ (row & 15)) << 4));
}

// V^T fragment (MFMA-A): output rows dv0..dv0+15, contraction slots = keys {key0+4g..+3, key0+16+4g..+3}
__device__ __forceinline__ half8_t vt_frag80(const half_t* sV, int key0, int dv0, int fr, int fq) {
    const half_t* a = sV + (key0 + fq * 4 + (fr >> 2)) * GF + dv0 + (fr & 3) * 4;
    const fp16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a));
    const fp16x4_t t2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(a + 16 * GF));
    half8_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (half_t)t1[j];
        r[j + 4] = (half_t)t2[j];
    }
    return r;
}

__device__ __forceinline__ half8_t cat4(half4_t a, half4_t b) { return half8_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// Materialise a converted fragment HERE: without it LLVM sinks the fp32 -> fp16 conversion to the fragment's first use (a phase
// later) and keeps - or spills - the fp32 accumulators in between.
__device__ __forceinline__ void xf_pin(half4_t& v) {
    typedef unsigned xf_u2 __attribute__((ext_vector_type(2)));
    xf_u2 t = __builtin_bit_cast(xf_u2, v);
    asm volatile("" : "+v"(t));
    v = __builtin_bit_cast(half4_t, t);
}

template <int N>
__device__ __forceinline__ void xf_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifndef PV_XF_SPAN
// 1: the C = 320 form runs its weight rings over both 40-KiB buffers (8 slots, three pairs of stages in flight; xattn_fused_kernel, SPAN).  Round 6, same box,
// sustained: 53.9 -> 51.1 us at B = 16 / N = 4096 (P = 5: 54.2 -> 52.5), level in the loop; results unchanged (profiles/r06_xfused_span.txt)
#define PV_XF_SPAN 1
#endif
#ifndef PV_XF_S640
#define PV_XF_S640 8
#endif
#ifndef PV_XF_S640_NQ2
#define PV_XF_S640_NQ2 4           // the 128-row form at C = 640 is one workgroup per CU too (390 VGPRs): 8 = all of the CU's LDS for its two rings (A/B switch)
#endif
constexpr int xf_ring_slots(int C, int NQ) { return C == 640 ? (NQ == 1 ? PV_XF_S640 : PV_XF_S640_NQ2) : 4; }

// wave-uniform count -> immediate
__device__ __forceinline__ void xf_wait_vmcnt_dyn(int n) {
    switch (n) {
#define XF_W(N) case N: xf_wait_vmcnt<N>(); break;
        XF_W(0) XF_W(1) XF_W(2) XF_W(3) XF_W(4) XF_W(5) XF_W(6) XF_W(7) XF_W(8) XF_W(9) XF_W(10) XF_W(11) XF_W(12) XF_W(13) XF_W(14) XF_W(15) XF_W(16)
        XF_W(17) XF_W(18) XF_W(19) XF_W(20) XF_W(21) XF_W(22) XF_W(23) XF_W(24) XF_W(25) XF_W(26) XF_W(27) XF_W(28) XF_W(29) XF_W(30) XF_W(31) XF_W(32)
#undef XF_W
        default: xf_wait_vmcnt<0>(); break;     // a smaller count only waits longer
    }
}

struct pv_xfused_params_dev : pv_xattn_fused_params {
    uint32_t w_bytes, kimg_bytes, vimg_bytes;
};

// IP1: exactly one image token (the reference's inference default, token_index = 0 -> (B, 1, 768), adapters.py:32-37).  A softmax over
// ONE key is 1 whatever the query, so the whole image-token branch is "+ w_ip * v_ip": its score fragment, its max / exp / sum and its
// two 4-lane reductions disappear (a fifth of the attention phase's VALU work); the value row still rides in the P.V contraction with
// the constant probability w_ip.
// C = 640 (d = 80): the same four phases; a group is ONE head of 80 features (five whole fragments), the K image rows are 256 B (three 32-slot
// k-steps used).  The row operands alone are X^ (160 VGPRs) + Q / context (160): one wave per SIMD (512-register budget, one workgroup per CU).
template <int C, bool IP1, int NQ = 2>
__global__ __launch_bounds__(256, (C == 320 || NQ == 1) ? 2 : 1) void xattn_fused_kernel(const pv_xfused_params_dev p) {
    constexpr int ROWS = 64 * NQ;     // query rows per workgroup: each of the 4 waves owns NQ 16-row MFMA column tiles
    static_assert(C % GF == 0 && C % 64 == 0, "C must be a multiple of 80 and of 64");
    constexpr int D = C / 8;          // 8 heads: d = 40 (C = 320) or 80 (C = 640)
    static_assert(D == 40 || D == 80, "instantiated for the C = 320 and C = 640 layers");
    constexpr int NG = C / GF;        // 80-feature groups: 4 pairs of heads (d = 40) or 8 heads (d = 80)
    constexpr int NFR = C / 16;       // 16-feature fragments per row: 20
    constexpr int KK = C / 32;        // 32-deep contraction steps over C: 10
    constexpr int KT = C / 64;        // 64-deep ring stages per 80-row weight chunk: 5
    constexpr int NT = NG * KT;       // ring stages per GEMM phase: 20
    // ring slots: two stages being read + 2 LEAD in flight (one workgroup barrier per PAIR of stages).  The 64-row form has ONE workgroup per CU
    // (256 workgroups) and all of the CU's LDS: a deeper ring keeps three pairs of stages in flight instead of one - with one pair the weight
    // stream of a lone workgroup is bound by the L2 round trip, not by its MFMAs
    // SPAN (C = 320, PV_XF_SPAN): the two-workgroups-per-CU form keeps its 80 KiB but runs each weight ring over BOTH buffers - eight slots, three pairs of
    // stages in flight instead of one.  Round 6 (profiles/r06_xfused_occupancy.txt): a workgroup ALONE on its CU takes 35 us for its 128 rows where its
    // resource floors sum to ~22 us, and a second co-resident workgroup adds only 17 us: the weight phases wait for the L2 round trip of the one pair of
    // stages in flight, as the C = 640 form did before its 8-slot ring.  The K / V group that used to wait in buffer 1 through phase 1 is fetched when the
    // last Wq stages have left that half (t = 16); the Wo ring starts in buffer 1 under the last group and takes buffer 0 over at the phase boundary.
    constexpr bool SPAN = C == 320 && NQ == 2 && PV_XF_SPAN != 0;
    constexpr int S = SPAN ? 8 : xf_ring_slots(C, NQ);
    constexpr int LEAD = S / 2 - 1;   // pairs of stages in flight behind the pair being read
    constexpr int TILE_BYTES = GF * 128;                   // one stage: 80 weight rows x 64 k = 10 KiB
    constexpr int KIMG_BYTES = XK * KROW;                  // one head
    constexpr int VIMG_BYTES = XK * GF * 2;
    constexpr int GROUP_BYTES = 2 * KIMG_BYTES + VIMG_BYTES;   // K images of the group's two heads + V image: 39 KiB
    constexpr int GROUP_PIECES = GROUP_BYTES / 1024;
    constexpr int BUF_BYTES = (SPAN ? 4 : S) * TILE_BYTES;     // 40 KiB
    static_assert(GROUP_BYTES % 1024 == 0 && GROUP_BYTES <= BUF_BYTES, "LDS plan: a K/V group fits inside one ring buffer");
    // LDS: two 40-KiB buffers.  Buffer 0 = Wq ring (phase 1), K/V of groups 1 and 3; buffer 1 = K/V of groups 0 and 2, Wo ring (phase 3).
    // 80 KiB per workgroup -> TWO independent workgroups per CU (each other's barrier / DMA / HBM waits are covered, as in the GEMM).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const buf0 = smem;
    char* const buf1 = smem + BUF_BYTES;

    const int lane = pv_lane_id(), wave = pv_wave_id();
    const int fr = lane & 15, g = lane >> 4;
    const int m0 = (int)blockIdx.x * ROWS;
    const int b = m0 / p.nq;
    const half_t* hs = reinterpret_cast<const half_t*>(p.hs);
    const float w_text = p.fusion ? p.fusion[0] : p.w_text;
    const float w_ip = p.fusion ? p.fusion[1] : p.w_ip;

    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wq), 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wo), 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.kimg), 0, (int)p.kimg_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.vimg), 0, (int)p.vimg_bytes, 0x00020000);
    // the workgroup's 128 rows of hs / out through buffer descriptors: one 32-bit row offset per lane instead of 64-bit pointers
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(hs) + (size_t)m0 * p.ld_hs, 0, ROWS * p.ld_hs * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<half_t*>(p.out) + (size_t)m0 * p.ld_out, 0, ROWS * p.ld_out * 2, 0x00020000);

    // ---- LDS-DMA issue helpers (one piece = one wave instruction = 1 KiB) -----------------------------------------------
    // Ring stage t of a [C][C] weight: rows [80 (t/KT), +80) x k [64 (t%KT), +64); 10 pieces of 8 rows x 128 B: waves 0,1 issue 3,
    // waves 2,3 issue 2.  Swizzle on the source: LDS position (lane & 7) of row r holds chunk (lane & 7) ^ (r & 7).
    const int lrow = lane >> 3;
    const unsigned w_lane_off = (unsigned)lrow * (unsigned)(C * 2) + (unsigned)(((lane & 7) ^ lrow) << 4);
    // per-lane part of the source offset of this wave's (up to) three pieces: stage-invariant VGPRs; the stage part goes into the
    // instruction's scalar offset (otherwise the compiler keeps 60 per-stage offsets alive from phase 1 to phase 3 and spills them)
    unsigned piece_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) piece_off[i] = w_lane_off + (unsigned)((wave + 4 * i) * 8) * (unsigned)(C * 2);
    // ring slot of stage t: phase 1 (Wq) counts from buffer 0, phase 3 (Wo) from buffer 1; SPAN: slots 4-7 lie in the other buffer
    auto slot1 = [&](int t) -> char* { return buf0 + (t % S) * TILE_BYTES; };                        // buffers are adjacent: slot s at smem + s * 10 KiB
    auto slot3 = [&](int t) -> char* { const int sl = t % S; return SPAN && sl >= 4 ? buf0 + (sl - 4) * TILE_BYTES : buf1 + sl * TILE_BYTES; };
    auto issue_stage = [&](const __amdgpu_buffer_rsrc_t& rw, char* dst, int t) {
        const int nc = t / KT, kt = t % KT;
        const int soff = nc * GF * (C * 2) + kt * 128;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 4 * i;
            if (i < 2 || wave < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, PV_LDS_PTR(dst + j * 1024), 16, (int)piece_off[i], soff, 0, 0);
        }
    };
    // K/V images of (sample b, group grp): a linear 39-KiB copy (the images are stored in LDS order): waves 0-2 issue 10 pieces, wave 3: 9
    auto issue_group = [&](int grp, char* dst) {
        const unsigned kbase = (unsigned)((b * (NG * 2) + 2 * grp) * KIMG_BYTES) + (unsigned)lane * 16u;
        const unsigned vbase = (unsigned)((b * NG + grp) * VIMG_BYTES) + (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int j = wave + 4 * i;
            if (j < 2 * KIMG_BYTES / 1024)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, PV_LDS_PTR(dst + j * 1024), 16, (int)(kbase + (unsigned)j * 1024u), 0, 0, 0);
            else if (j < GROUP_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, PV_LDS_PTR(dst + j * 1024), 16, (int)(vbase + (unsigned)(j - 2 * KIMG_BYTES / 1024) * 1024u), 0, 0, 0);
        }
    };
    // vmcnt bookkeeping in units of "this wave's pieces": ring stage = 3 (waves 0,1) or 2; K/V group = 10 (waves 0-2) or 9
    auto wait_except = [&](int stages, int groups) {       // wait for everything but the youngest `stages` ring stages + `groups` K/V groups
        const int n = stages * (wave < 2 ? 3 : 2) + groups * (wave < 3 ? 10 : 9);
        xf_wait_vmcnt_dyn(n);
    };
    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads are retired
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- phase 0: X^T into registers (+ LayerNorm) -----------------------------------------------------------------------
    half8_t xf[KK][NQ];
    int mrow[NQ];
    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        mrow[qi] = wave * (16 * NQ) + qi * 16 + fr;                 // row inside the workgroup's block
        const int off = mrow[qi] * p.ld_hs * 2 + g * 16;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) xf[kk][qi] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rh, off, kk * 64, 0));
    }
    if (!SPAN) issue_group(0, buf1);      // K/V of group 0 waits in buffer 1 while the Wq ring runs in buffer 0 (SPAN: fetched at t = 16)
#pragma unroll
    for (int t = 0; t < 2 * LEAD; ++t) issue_stage(rq, slot1(t), t);
    if (p.ln) {
        // LayerNorm WITHOUT its affine part (the caller folds gamma into the columns of wq and beta into q_bias):
        // x^ = x * rstd - mean * rstd, one mixed-precision FMA per element (fp16 in, fp32 math, fp16 out).  Statistics: the row sum
        // by v_dot2_f32_f16 against (1, 1); the centred sum of squares from d = x - fp16(mean) (packed fp16 subtract, the rounding
        // of d averages out over the 320 terms) by v_dot2_f32_f16(d, d), corrected exactly for the rounding of the mean.
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        const h2_t ones = h2_t{(half_t)1.0f, (half_t)1.0f};
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
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
    // The register loads above made the compiler wait; start the hand-counted DMA bookkeeping from a known state too:
    // everything issued so far (K/V group 0 and the first two Wq stages) has landed after this wait.
    xf_wait_vmcnt<0>();

    // Fragment reads of a ring: row 16 i + fr, chunk (4 ks + g) ^ (fr & 7): two per-lane bases (ks = 0 / 1) + compile-time offsets
    const int ring_lane[2] = {fr * 128 + ((g ^ (fr & 7)) << 4), fr * 128 + (((4 + g) ^ (fr & 7)) << 4)};

    // ---- phase 1: Q^T = Wq . X^T -----------------------------------------------------------------------------------------
    // qf[f][qi]: fp16 of fragment f (features 16 f + 4g + r) for this lane's query, pre-scaled by log2(e)/sqrt(d)
    half4_t qf[NFR][NQ];
    const float qscale = rsqrtf((float)D) * 1.4426950408889634f;
    {
        float4_t acc[5][NQ];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nc = t / KT, kt = t % KT;
            if (kt == 0) {
                // accumulators start from the query bias (beta of the folded LayerNorm pushed through to_q), rows 4g..4g+3 of each fragment
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    { const float4_t v0 = p.q_bias ? *reinterpret_cast<const float4_t*>(p.q_bias + nc * GF + i * 16 + g * 4) : float4_t{0.f, 0.f, 0.f, 0.f}; for (int qi = 0; qi < NQ; ++qi) acc[i][qi] = v0; }
            }
            if ((t & 1) == 0) {
                // ONE barrier per pair of stages: stages t and t+1 were issued two stages ago right behind that barrier and nothing younger
                // is in flight, so "landed" is vmcnt(0); behind the barrier the slots of stages t-2 / t-1 are read out by every wave
                // stages t, t+1 landed; younger pairs may be in flight (SPAN: and, from t = 18 on, K / V group 0, issued behind the last stages)
                if (t >= 2) wait_except(min(2 * (LEAD - 1), max(NT - (t + 2), 0)), SPAN && t >= NT - 2 ? 1 : 0);
                wg_barrier();
                if (t + 2 * LEAD < NT) issue_stage(rq, slot1(t + 2 * LEAD), t + 2 * LEAD);
                if (t + 2 * LEAD + 1 < NT) issue_stage(rq, slot1(t + 2 * LEAD + 1), t + 2 * LEAD + 1);
                // SPAN: behind this barrier every wave has read stages <= 15 out of slots 4-7 (buffer 1), and no later stage goes there
                if (SPAN && t == NT - 4) issue_group(0, buf1);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* base = slot1(t) + ring_lane[ks];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const half8_t a = *reinterpret_cast<const half8_t*>(base + i * 2048);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) acc[i][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[2 * kt + ks][qi], acc[i][qi], 0, 0, 0);
                }
            }
            if (kt == KT - 1) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) qf[nc * 5 + i][qi][r] = (half_t)(acc[i][qi][r] * qscale);
                        xf_pin(qf[nc * 5 + i][qi]);      // convert here: do not carry fp32 accumulators into the next chunk / phase
                    }
            }
        }
    }

    // ---- phase 2: dual-branch attention, one 80-feature group (two heads) at a time; K/V double-buffered ------------------
    // group g lives in buffer (g + 1) & 1: group 0 was prefetched into buffer 1 at kernel start
    wg_barrier();                          // every wave is done with the Wq ring (buffer 0)
    issue_group(1, buf0);
    half4_t cf[NFR][NQ];                    // context fragments, same layout as qf
    // padding keys of fragment 4 (text tail) and fragment 5 (image tokens): their K rows are zero, so the MFMA leaves the
    // accumulator's initial value in place - start those from -inf instead of selecting per score afterwards
    float4_t tinit, iinit;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tinit[r] = 64 + g * 4 + r < p.nt ? 0.f : -INFINITY;
        iinit[r] = g * 4 + r < p.nip ? 0.f : -INFINITY;
    }
    const half4_t ip1p = half4_t{g == 0 ? (half_t)w_ip : (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    float4_t zero4 = float4_t{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(zero4));            // one register quad for the whole phase (a literal would be re-materialised per chain)
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        // group grp landed; in flight behind it: group grp+1 (grp < 3); at grp == 3 additionally the three Wo stages issued after group 2
        if (grp == 0) wait_except(0, 1);
        else if (grp < NG - 1) { wait_except(0, 1); }
        else wait_except(SPAN ? 4 : 2 * LEAD, 0);
        wg_barrier();
        const char* sbuf = ((grp + 1) & 1) ? buf1 : buf0;
        const half_t* sV = reinterpret_cast<const half_t*>(sbuf + 2 * KIMG_BYTES);
        if constexpr (D == 80) {
            // ---- d = 80: the group IS one head; five whole fragments, three 32-slot k-steps (the third half empty) ----------------------
            const char* sK = sbuf;
            const int f0 = grp * 5;
            constexpr int NKB = IP1 ? 5 : 6;
            float4_t s[NKB][NQ];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                half8_t bq[NQ];
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) bq[qi] = ks < 2 ? cat4(qf[f0 + 2 * ks][qi], qf[f0 + 2 * ks + 1][qi]) : cat4(qf[f0 + 4][qi], qf[f0 + 4][qi]);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    const half8_t a = ld_frag256(sK, kb * 16 + fr, ks * 4 + g);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
                        s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq[qi], ks == 0 ? (kb == 4 ? tinit : kb == 5 ? iinit : zero4) : s[kb][qi], 0, 0, 0);
                }
            }
            half8_t pb[3][NQ];
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                float mt = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 5; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mt = fmaxf(mt, s[kb][qi][r]);
                mt = pv_quad_max(mt);
                float lt = 0.f;
#pragma unroll
                for (int kb = 0; kb < 5; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s[kb][qi][r] = PV_EXP2(s[kb][qi][r] - mt);
                        lt += s[kb][qi][r];
                    }
                lt = pv_quad_sum(lt);
                const float ft = w_text * __builtin_amdgcn_rcpf(lt);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pb[s2][qi][r] = (half_t)(s[2 * s2][qi][r] * ft);
                        pb[s2][qi][r + 4] = (half_t)(s[2 * s2 + 1][qi][r] * ft);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[2][qi][r] = (half_t)(s[4][qi][r] * ft);
                if constexpr (IP1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pb[2][qi][r + 4] = ip1p[r];
                } else {
                    float mi = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 4; ++r) mi = fmaxf(mi, s[NKB - 1][qi][r]);
                    mi = pv_quad_max(mi);
                    float li = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s[NKB - 1][qi][r] = PV_EXP2(s[NKB - 1][qi][r] - mi);
                        li += s[NKB - 1][qi][r];
                    }
                    li = pv_quad_sum(li);
                    const float fi = w_ip * __builtin_amdgcn_rcpf(li);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pb[2][qi][r + 4] = (half_t)(s[NKB - 1][qi][r] * fi);
                }
            }
            float4_t o[5][NQ];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                for (int fi = 0; fi < 5; ++fi) {
                    const half8_t a = vt_frag80(sV, s2 * 32, fi * 16, fr, g);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) o[fi][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[s2][qi], s2 == 0 ? zero4 : o[fi][qi], 0, 0, 0);
                }
#pragma unroll
            for (int fi = 0; fi < 5; ++fi)
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cf[f0 + fi][qi][r] = (half_t)o[fi][qi][r];
                    xf_pin(cf[f0 + fi][qi]);
                }
        } else {
            float4_t o2_h0[NQ];                 // head 0's fragment 2 (its rows 0-7 are head 0's features 32..39)
    #pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const char* sK = sbuf + hh * KIMG_BYTES;
                const int fa = grp * 5 + (hh == 0 ? 0 : 3), fb = fa + 1, f2 = grp * 5 + 2;
                constexpr int NKB = IP1 ? 5 : 6;   // score fragments: 5 of text keys (+ 1 of image-token keys)
                float4_t s[NKB][NQ];
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    half8_t bq[NQ];
    #pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) bq[qi] = ks == 0 ? cat4(qf[fa][qi], qf[fb][qi]) : cat4(qf[f2][qi], qf[f2][qi]);
    #pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) {
                        const half8_t a = ld_frag128(sK, kb * 16 + fr, ks * 4 + g);
                        // a chain's first step reads its initial value (zero / the padding masks) as the C operand from loop-invariant registers: no
                        // accumulator-initialising v_mov (a tenth of the kernel's vector instructions were those)
    #pragma unroll
                        for (int qi = 0; qi < NQ; ++qi)
                            s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq[qi], ks == 0 ? (kb == 4 ? tinit : kb == 5 ? iinit : zero4) : s[kb][qi], 0, 0, 0);
                    }
                }
                // two independent softmaxes over the key axis (registers r, fragments kb, and the 4 lane groups).  Text keys are
                // fragments 0..4 (keys 0..79; only the tail of fragment 4 can be padding), image-token keys are fragment 5.
                half8_t pb[3][NQ];
    #pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    float mt = -INFINITY;
    #pragma unroll
                    for (int kb = 0; kb < 5; ++kb)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) mt = fmaxf(mt, s[kb][qi][r]);
                    mt = pv_quad_max(mt);
                    float lt = 0.f;
    #pragma unroll
                    for (int kb = 0; kb < 5; ++kb)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            s[kb][qi][r] = PV_EXP2(s[kb][qi][r] - mt);       // exp2(-inf) = 0 for the padding keys
                            lt += s[kb][qi][r];
                        }
                    lt = pv_quad_sum(lt);
                    const float ft = w_text * __builtin_amdgcn_rcpf(lt);
    #pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pb[s2][qi][r] = (half_t)(s[2 * s2][qi][r] * ft);
                            pb[s2][qi][r + 4] = (half_t)(s[2 * s2 + 1][qi][r] * ft);
                        }
    #pragma unroll
                    for (int r = 0; r < 4; ++r) pb[2][qi][r] = (half_t)(s[4][qi][r] * ft);
                    if constexpr (IP1) {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) pb[2][qi][r + 4] = ip1p[r];      // softmax over one key = 1: P = w_ip at key XIP0, 0 elsewhere
                    } else {
                        float mi = -INFINITY;
    #pragma unroll
                        for (int r = 0; r < 4; ++r) mi = fmaxf(mi, s[NKB - 1][qi][r]);
                        mi = pv_quad_max(mi);
                        float li = 0.f;
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            s[NKB - 1][qi][r] = PV_EXP2(s[NKB - 1][qi][r] - mi);
                            li += s[NKB - 1][qi][r];
                        }
                        li = pv_quad_sum(li);
                        const float fi = w_ip * __builtin_amdgcn_rcpf(li);
    #pragma unroll
                        for (int r = 0; r < 4; ++r) pb[2][qi][r + 4] = (half_t)(s[NKB - 1][qi][r] * fi);
                    }
                }
                // O^T = V^T . P^T for this head's three fragments of the group's 80 value columns
                float4_t o[3][NQ];
    #pragma unroll
                for (int s2 = 0; s2 < 3; ++s2)
    #pragma unroll
                    for (int fi = 0; fi < 3; ++fi) {
                        const half8_t a = vt_frag80(sV, s2 * 32, (hh == 0 ? fi : fi + 2) * 16, fr, g);
    #pragma unroll
                        for (int qi = 0; qi < NQ; ++qi) o[fi][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[s2][qi], s2 == 0 ? zero4 : o[fi][qi], 0, 0, 0);
                    }
    #pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    if (hh == 0) {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            cf[grp * 5 + 0][qi][r] = (half_t)o[0][qi][r];
                            cf[grp * 5 + 1][qi][r] = (half_t)o[1][qi][r];
                        }
                        xf_pin(cf[grp * 5 + 0][qi]);
                        xf_pin(cf[grp * 5 + 1][qi]);
                        o2_h0[qi] = o[2][qi];
                    } else {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            cf[grp * 5 + 2][qi][r] = (half_t)(g < 2 ? o2_h0[qi][r] : o[0][qi][r]);
                            cf[grp * 5 + 3][qi][r] = (half_t)o[1][qi][r];
                            cf[grp * 5 + 4][qi][r] = (half_t)o[2][qi][r];
                        }
                        xf_pin(cf[grp * 5 + 2][qi]);
                        xf_pin(cf[grp * 5 + 3][qi]);
                        xf_pin(cf[grp * 5 + 4][qi]);
                    }
                }
            }
        }
        if (grp + 2 < NG) {
            wg_barrier();                  // every wave is done with this buffer: refill it with group grp + 2
            issue_group(grp + 2, ((grp + 1) & 1) ? buf1 : buf0);
        } else if (grp == NG - 2) {
            wg_barrier();                  // buffer 1 (group 2) is free: the Wo ring starts there while group 3 computes out of buffer 0
#pragma unroll
            for (int t = 0; t < (SPAN ? 4 : 2 * LEAD); ++t) issue_stage(ro, slot3(t), t);     // SPAN: slots 4-7 (buffer 0) follow at the phase boundary
        }
    }

    // ---- phase 3: out^T = Wo' . ctx^T, + bias + residual ----------------------------------------------------------------
    // A lane's accumulator registers are 4 consecutive output columns of fragment i (8 bytes of fp16).  v_permlane16_swap trades the
    // even lane-rows' fragment 2q+1 against the odd lane-rows' fragment 2q, after which every lane owns 8 CONSECUTIVE columns: the
    // residual rows are fetched and the output rows stored with 16 bytes per lane (fragment 4, the odd one out, keeps 8 bytes).
    {
        float4_t acc[5][NQ];
        uint4_t res16[2][NQ];              // residual of fragment pairs (0,1), (2,3) in the swapped (16-byte) layout
        uint2_t res8[NQ];                  // residual of fragment 4
        const int pcol = (g & 1) ? 16 + (g - 1) * 4 : g * 4;      // this lane's 8 columns inside a fragment pair
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nc = t / KT, kt = t % KT;
            const int nb = nc * GF + g * 4;
            if ((t & 1) == 0) {
                // stages t, t+1 (issued two stages ago) landed.  vmcnt retires in order, so only what this wave issued AFTER those pieces may
                // stay in flight: the 6 output stores of a chunk that ended in stage t-2 or t-1 and the 6 residual loads of a chunk that
                // started there (the bias loads are not counted: a smaller count only waits longer).  Counting them exactly keeps the HBM
                // write latency of a chunk's stores out of the next stages' critical path.
                constexpr int RES = 3 * NQ, STO = 3 * NQ;
                // vmcnt retires in order: everything this wave issued AFTER the DMA of stages t, t+1 may stay in flight - the younger ring stages
                // and the residual loads / output stores of the chunks that started / ended in the last 2 LEAD stages
                int younger = 0;
#pragma unroll
                for (int j = t - 2 * LEAD; j < t; ++j) {
                    if (j < 0) continue;
                    if (j % KT == 0) younger += RES;
                    if (j % KT == KT - 1) younger += STO;
                }
                younger += (t >= 2 ? min(2 * (LEAD - 1), max(NT - (t + 2), 0)) : 2 * LEAD - 2) * (wave < 2 ? 3 : 2);
                if (t >= 2) xf_wait_vmcnt_dyn(younger);
                else xf_wait_vmcnt<0>();
                wg_barrier();
                if (SPAN && t == 0) {          // buffer 0 (the last K / V group) is free now: the ring's other half
                    issue_stage(ro, slot3(4), 4);
                    issue_stage(ro, slot3(5), 5);
                }
                if (t + 2 * LEAD < NT) issue_stage(ro, slot3(t + 2 * LEAD), t + 2 * LEAD);
                if (t + 2 * LEAD + 1 < NT) issue_stage(ro, slot3(t + 2 * LEAD + 1), t + 2 * LEAD + 1);
            }
            if (kt == 0) {
                // chunk start: accumulators start from the output bias; the residual rows are requested now and consumed five stages later
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    { const float4_t v0 = p.bias_o ? *reinterpret_cast<const float4_t*>(p.bias_o + nb + i * 16) : float4_t{0.f, 0.f, 0.f, 0.f}; for (int qi = 0; qi < NQ; ++qi) acc[i][qi] = v0; }
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const int roff = mrow[qi] * p.ld_hs * 2;
                    res16[0][qi] = __builtin_amdgcn_raw_buffer_load_b128(rh, roff + pcol * 2 + nc * (GF * 2), 0, 0);
                    res16[1][qi] = __builtin_amdgcn_raw_buffer_load_b128(rh, roff + pcol * 2 + nc * (GF * 2) + 64, 0, 0);
                    res8[qi] = __builtin_amdgcn_raw_buffer_load_b64(rh, roff + g * 8 + nc * (GF * 2) + 128, 0, 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* base = slot3(t) + ring_lane[ks];
                half8_t bc[NQ];
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) bc[qi] = cat4(cf[2 * (2 * kt + ks)][qi], cf[2 * (2 * kt + ks) + 1][qi]);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const half8_t a = *reinterpret_cast<const half8_t*>(base + i * 2048);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) acc[i][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bc[qi], acc[i][qi], 0, 0, 0);
                }
            }
            if (kt == KT - 1) {
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const int ooff = mrow[qi] * p.ld_out * 2;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // residual back to the accumulator layout (the swap is its own inverse), add in fp32, round once, swap, store
                        const auto u0 = __builtin_amdgcn_permlane16_swap(res16[q][qi][0], res16[q][qi][2], false, false);
                        const auto u1 = __builtin_amdgcn_permlane16_swap(res16[q][qi][1], res16[q][qi][3], false, false);
                        const unsigned ra0 = u0[0], rb0 = u0[1], ra1 = u1[0], rb1 = u1[1];
                        const half2_t r00 = __builtin_bit_cast(half2_t, ra0), r01 = __builtin_bit_cast(half2_t, ra1);
                        const half2_t r10 = __builtin_bit_cast(half2_t, rb0), r11 = __builtin_bit_cast(half2_t, rb1);
                        const float4_t a0 = acc[2 * q][qi], a1 = acc[2 * q + 1][qi];
                        const unsigned p00 = __builtin_bit_cast(unsigned, half2_t{(half_t)(a0[0] + (float)r00[0]), (half_t)(a0[1] + (float)r00[1])});
                        const unsigned p01 = __builtin_bit_cast(unsigned, half2_t{(half_t)(a0[2] + (float)r01[0]), (half_t)(a0[3] + (float)r01[1])});
                        const unsigned p10 = __builtin_bit_cast(unsigned, half2_t{(half_t)(a1[0] + (float)r10[0]), (half_t)(a1[1] + (float)r10[1])});
                        const unsigned p11 = __builtin_bit_cast(unsigned, half2_t{(half_t)(a1[2] + (float)r11[0]), (half_t)(a1[3] + (float)r11[1])});
                        const auto v0 = __builtin_amdgcn_permlane16_swap(p00, p10, false, false);
                        const auto v1 = __builtin_amdgcn_permlane16_swap(p01, p11, false, false);
                        const unsigned sa0 = v0[0], sb0 = v0[1], sa1 = v1[0], sb1 = v1[1];
                        // constant part of the address in the IMMEDIATE offset, never in soffset: with an SGPR soffset the compiler does not separate a
                        // > 8-byte store from a following VALU write of its data registers (the store then reads overwritten data in some lanes)
                        __builtin_amdgcn_raw_buffer_store_b128(uint4_t{sa0, sa1, sb0, sb1}, rout, ooff + pcol * 2 + (nc * (GF * 2) + q * 64), 0, 0);
                        asm volatile("s_nop 1" ::: "memory");
                    }
                    const half2_t r0 = __builtin_bit_cast(half2_t, (unsigned)res8[qi][0]), r1 = __builtin_bit_cast(half2_t, (unsigned)res8[qi][1]);
                    const float4_t a4 = acc[4][qi];
                    const unsigned p0 = __builtin_bit_cast(unsigned, half2_t{(half_t)(a4[0] + (float)r0[0]), (half_t)(a4[1] + (float)r0[1])});
                    const unsigned p1 = __builtin_bit_cast(unsigned, half2_t{(half_t)(a4[2] + (float)r1[0]), (half_t)(a4[3] + (float)r1[1])});
                    __builtin_amdgcn_raw_buffer_store_b64(uint2_t{p0, p1}, rout, ooff + g * 8 + (nc * (GF * 2) + 128), 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

}  // namespace

extern "C" int pv_xattn_pack_kv(const void* kt, const void* vt, int32_t ldkt, int32_t ldvt, const void* kip, const void* vip, int32_t ldkip,
                                int32_t ldvip, void* kimg, void* vimg, float* vnorm, int32_t batch, int32_t heads, int32_t d, int32_t nt,
                                int32_t nip, void* stream) {
    if (!kt || !vt || !kip || !vip || !kimg || !vimg || batch <= 0 || heads <= 0 || (heads & 1) || (d != 40 && d != 80) || nt <= 0 || nt > XIP0 || nip <= 0 ||
        nip > XK - XIP0 || (ldkt % 8) || (ldvt % 8) || (ldkip % 8) || (ldvip % 8))
        return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3(batch * XK), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const half_t*>(kt),
                       reinterpret_cast<const half_t*>(vt), ldkt, ldvt, reinterpret_cast<const half_t*>(kip), reinterpret_cast<const half_t*>(vip),
                       ldkip, ldvip, reinterpret_cast<half_t*>(kimg), reinterpret_cast<half_t*>(vimg), vnorm, heads, nt, nip, d);
    return PV_CHECK_LAUNCH();
}

// ctx slot order of the to_out contraction: slot 32 s + 8 g + jj of the permuted matrix holds natural column
// 16 (2s) + 4 g + jj (jj < 4) or 16 (2s + 1) + 4 g + (jj - 4)
extern "C" int pv_xattn_fused_wo_slot(int32_t slot) {
    const int s = slot >> 5, g = (slot >> 3) & 3, jj = slot & 7;
    return jj < 4 ? 32 * s + 4 * g + jj : 32 * s + 16 + 4 * g + (jj - 4);
}

extern "C" int pv_cross_attention_fused(const pv_xattn_fused_params* pp, void* stream) {
    pv_xfused_params_dev p;
    static_cast<pv_xattn_fused_params&>(p) = *pp;
    const int C = p.heads * p.d;
    if (!p.hs || !p.wq || !p.wo || !p.kimg || !p.vimg || !p.out || p.batch <= 0 || p.nq <= 0 || (p.nq % 128) || p.heads != 8 || (C != 320 && C != 640) ||
        p.nt <= 64 || p.nt > XIP0 || p.nip <= 0 || p.nip > XK - XIP0 || (p.ld_hs % 8) || (p.ld_out % 8))
        return (int)hipErrorInvalidValue;
    p.w_bytes = (uint32_t)C * C * 2;
    const size_t kb = (size_t)p.batch * p.heads * XK * (p.d == 40 ? 128 : 256), vb = (size_t)p.batch * (C / GF) * XK * GF * 2;
    if (kb >= (1ull << 31) || vb >= (1ull << 31)) return (int)hipErrorInvalidValue;
    p.kimg_bytes = (uint32_t)kb;
    p.vimg_bytes = (uint32_t)vb;
    // rows per workgroup: 128 (a wave owns two 16-row tiles: every weight fragment read from LDS feeds two MFMAs) or 64.  The C = 640 levels
    // have 16 x 1024 rows = 128 workgroups of 128 - half of the 256 CUs; 64-row workgroups fill the chip.  A caller with two such launches side by
    // side (the CFG branches on two streams) asks for 128 rows through rows_per_workgroup: 58 vs 78 us per layer in that schedule (PV_XF_ROWS=64/128 overrides both).
    static const int rows_env = [] { const char* e = getenv("PV_XF_ROWS"); return e ? atoi(e) : 0; }();
    const size_t M = (size_t)p.batch * p.nq;
    int rows = C == 640 && M / 128 < 384 ? 64 : 128;
    if (C == 640 && (p.rows_per_workgroup == 64 || p.rows_per_workgroup == 128)) rows = p.rows_per_workgroup;
    if (C == 640 && (rows_env == 64 || rows_env == 128)) rows = rows_env;
    static bool attr_set_dev[64][6] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const bool ip1 = p.nip == 1;
    const int variant = C == 320 ? 0 : rows == 128 ? 1 : 2;
    bool& attr_set = attr_set_dev[dev_id & 63][variant * 2 + (ip1 ? 1 : 0)];
    auto kern = variant == 0 ? (ip1 ? xattn_fused_kernel<320, true, 2> : xattn_fused_kernel<320, false, 2>)
              : variant == 1 ? (ip1 ? xattn_fused_kernel<640, true, 2> : xattn_fused_kernel<640, false, 2>)
                             : (ip1 ? xattn_fused_kernel<640, true, 1> : xattn_fused_kernel<640, false, 1>);
    // two buffers of one weight ring (or one 39-KiB K/V group) each: 2 x 40 KiB = two workgroups per CU; the 64-row form 2 x 80 KiB = the whole CU
    // PV_XF_LDS_PAD (diagnostics): extra dynamic LDS bytes per workgroup - 4096 leaves room for ONE 128-row workgroup per CU instead of two (occupancy probe)
    static const int lds_pad = getenv("PV_XF_LDS_PAD") ? atoi(getenv("PV_XF_LDS_PAD")) : 0;
    const int SMEM = 2 * xf_ring_slots(C, rows == 64 ? 1 : 2) * GF * 128 + (rows == 128 ? lds_pad : 0);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(M / rows)), dim3(256), SMEM, (hipStream_t)stream, p);
    return PV_CHECK_LAUNCH();
}
