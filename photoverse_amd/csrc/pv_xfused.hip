// pv_cross_attention_fused: the whole attn2 branch of a BasicTransformerBlock as ONE kernel for gfx950 (CDNA4)
//
//   out = hs + to_out( w_t * softmax(q Kt^T / sqrt(d)) Vt + w_i * softmax(q Kip^T / sqrt(d)) Vip ) + bias,   q = to_q(LayerNorm(hs))
//
// i.e. norm2 -> PhotoVerseAttnProcessor2_0.__call__ (/root/reference/models/attention_processor.py:297 to_q, :307-322 text
// SDPA, :392-420 image-token SDPA + fusion, :423 to_out[0]) -> residual add of the transformer block.  It replaces four
// launches (LayerNorm, to_q GEMM, dual-branch attention, to_out GEMM) and six passes of a (B*N, C) tensor through HBM by one
// read of hs and one write of out.  Built for the C = 320 / d = 40 layers (64x64 and larger levels: 63 % of attn2's launches'
// time at 512x512), where M = B*N is large and the GEMMs are too short-K to run well on their own.
//
// Structure: one workgroup = 128 query rows x ALL heads, 4 waves, each wave owns 32 rows (two 16-row MFMA columns) for the
// whole chain, so no activation ever leaves the register file:
//
//   phase 0  X^T (B operand of v_mfma_f32_16x16x32_f16: lane = query, 8 consecutive channels per k-group) is loaded straight
//            into registers; LayerNorm statistics are 80 in-lane adds + a 4-lane swap reduction; normalised in place.
//   phase 1  Q^T[n][q] = Wq[n][:] . X^T : Wq streams through a 4-stage LDS ring (80 rows x 64 k per stage, LDS-DMA, counted
//            vmcnt, one raw s_barrier per stage); the accumulator layout (lane = query, 4 consecutive features per k-group)
//            IS the next product's B operand once converted to fp16 - with a permuted contraction order, which the K image
//            and the columns of Wo are pre-permuted to match on the host / in pv_xattn_pack_kv.
//   phase 2  per 80-feature group (= 2 heads of 40): S^T = K.Q^T (K image rows from LDS), two independent softmaxes in
//            registers, O^T = V^T.P^T (V^T fragments by ds_read_b64_tr_b16); K/V images of the (sample, group) are LDS-DMA
//            double-buffered.  The 2.5-fragment head boundary needs no padding: fragment 2 of a group is computed for both
//            heads and merged by lane group.
//   phase 3  out^T[n][q] = Wo'[n][:] . ctx^T through the same ring; epilogue adds bias + residual and stores fp16.
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

// ---------------------------------------------------------------------------------------------------------------------
// pv_xattn_pack_kv: builds, once per conditioning, the exact LDS images the fused kernel DMA-copies:
//   kimg [B][heads][96][64] fp16, slot order of kslot_feature40, 16-B chunk c of row r stored at position c ^ (r & 7)
//   vimg [B][C/80][96][80] fp16 (natural column order)
//   vnorm[B][heads][nip]   = ||Vip[b, p, h, :]||_2   (attention_processor.py:397)
__global__ __launch_bounds__(256) void xattn_pack_kv_kernel(const half_t* kt, const half_t* vt, int ldkt, int ldvt, const half_t* kip,
                                                            const half_t* vip, int ldkip, int ldvip, half_t* kimg, half_t* vimg,
                                                            float* vnorm, int heads, int nt, int nip) {
    const int b = blockIdx.x / XK, key = blockIdx.x % XK;
    const int C = heads * 40;
    const half_t *krow = nullptr, *vrow = nullptr;
    if (key < nt) {
        krow = kt + (size_t)(b * nt + key) * ldkt;
        vrow = vt + (size_t)(b * nt + key) * ldvt;
    } else if (key >= XIP0 && key < XIP0 + nip) {
        krow = kip + (size_t)(b * nip + key - XIP0) * ldkip;
        vrow = vip + (size_t)(b * nip + key - XIP0) * ldvip;
    }
    const int tid = threadIdx.x;
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
    for (int i = tid; i < C / 8; i += 256) {            // V: 16-B chunks, natural order
        const int grp = (i * 8) / GF, j = i * 8 - grp * GF;
        half8_t v = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (vrow) v = *reinterpret_cast<const half8_t*>(vrow + i * 8);
        *reinterpret_cast<half8_t*>(vimg + ((size_t)(b * (C / GF) + grp) * XK + key) * GF + j) = v;
    }
    if (vnorm && key >= XIP0 && key < XIP0 + nip && tid < heads) {
        float a = 0.f;
        for (int d = 0; d < 40; ++d) {
            const float v = (float)vrow[tid * 40 + d];
            a += v * v;
        }
        vnorm[((size_t)b * heads + tid) * nip + (key - XIP0)] = sqrtf(a);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ half8_t ld_frag128(const char* base, int row, int chunk) {   // 128-B rows, chunk ^= row & 7
    return *reinterpret_cast<const half8_t*>(base + row * 128 + ((chunk ^ (row & 7)) << 4));
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

template <int N>
__device__ __forceinline__ void xf_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct pv_xfused_params_dev : pv_xattn_fused_params {
    uint32_t w_bytes, kimg_bytes, vimg_bytes;
};

template <int C>
__global__ __launch_bounds__(256, 1) void xattn_fused_kernel(const pv_xfused_params_dev p) {
    static_assert(C % GF == 0 && C % 64 == 0, "C must be a multiple of 80 and of 64");
    constexpr int D = 40;
    constexpr int NG = C / GF;        // feature groups (pairs of heads)
    constexpr int NFR = C / 16;       // 16-feature fragments per row
    constexpr int KK = C / 32;        // 32-deep contraction steps over C
    constexpr int KT = C / 64;        // 64-deep ring stages per 80-row weight chunk
    constexpr int NT = NG * KT;       // ring stages per GEMM phase
    constexpr int S = 4;              // ring depth
    constexpr int TILE_BYTES = GF * 128;                   // 80 weight rows x 64 k
    constexpr int KIMG_BYTES = XK * KROW;                  // one head
    constexpr int VIMG_BYTES = XK * GF * 2;
    constexpr int GROUP_BYTES = 2 * KIMG_BYTES + VIMG_BYTES;
    constexpr int GROUP_PIECES = GROUP_BYTES / 1024;       // 39
    static_assert(GROUP_BYTES % 1024 == 0 && S * TILE_BYTES <= 2 * GROUP_BYTES, "LDS plan");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = pv_lane_id(), wave = pv_wave_id();
    const int fr = lane & 15, g = lane >> 4;
    const int m0 = (int)blockIdx.x * 128;
    const int b = m0 / p.nq;
    const half_t* hs = reinterpret_cast<const half_t*>(p.hs);
    const float w_text = p.fusion ? p.fusion[0] : p.w_text;
    const float w_ip = p.fusion ? p.fusion[1] : p.w_ip;

    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wq), 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wo), 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.kimg), 0, (int)p.kimg_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.vimg), 0, (int)p.vimg_bytes, 0x00020000);

    // ---- LDS-DMA issue helpers ------------------------------------------------------------------------------------------
    // weight ring stage t: rows [80 nc, +80) x k [64 kt, +64) of a [C][C] matrix; piece j = 8 rows x 128 B; waves 0,1 issue 3
    // pieces, waves 2,3 issue 2.  Swizzle on the source: LDS position (lane & 7) of row r holds chunk (lane & 7) ^ (r & 7).
    const int lrow = lane >> 3;
    const unsigned w_lane_off = (unsigned)lrow * (unsigned)(C * 2) + (unsigned)(((lane & 7) ^ lrow) << 4);
    auto issue_tile = [&](const __amdgpu_buffer_rsrc_t& rw, int t) {
        const int nc = t / KT, kt = t - nc * KT;
        char* dst = smem + (t % S) * TILE_BYTES;
        const unsigned base = (unsigned)(nc * GF) * (unsigned)(C * 2) + (unsigned)(kt * 128) + w_lane_off;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 4 * i;
            if (i < 2 || wave < 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, PV_LDS_PTR(dst + j * 1024), 16, (int)(base + (unsigned)(j * 8) * (unsigned)(C * 2)), 0, 0, 0);
        }
    };
    // K/V images of (sample b, group grp): a linear 39-KiB copy (the images are stored in LDS order); piece j = 1 KiB
    auto issue_group = [&](int grp, int buf) {
        char* dst = smem + buf * GROUP_BYTES;
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
    // wait until this wave's pieces of everything but the `y` youngest ring stages have landed
    auto wait_tiles = [&](int y) {
        if (wave < 2) { if (y == 2) xf_wait_vmcnt<6>(); else if (y == 1) xf_wait_vmcnt<3>(); else xf_wait_vmcnt<0>(); }
        else          { if (y == 2) xf_wait_vmcnt<4>(); else if (y == 1) xf_wait_vmcnt<2>(); else xf_wait_vmcnt<0>(); }
    };

    // ---- phase 0: X^T into registers (+ LayerNorm) -----------------------------------------------------------------------
    issue_tile(rq, 0);
    issue_tile(rq, 1);
    issue_tile(rq, 2);
    half8_t xf[KK][2];
    int mrow[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        mrow[qi] = m0 + wave * 32 + qi * 16 + fr;
        const half_t* src = hs + (size_t)mrow[qi] * p.ld_hs + g * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) xf[kk][qi] = *reinterpret_cast<const half8_t*>(src + kk * 32);
    }
    if (p.ln_gamma) {
        float mean[2], rstd[2];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            float sum = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += (float)xf[kk][qi][j];
            mean[qi] = pv_quad_sum(sum) * (1.0f / (float)C);
            float sq = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = (float)xf[kk][qi][j] - mean[qi];
                    sq += d * d;
                }
            rstd[qi] = rsqrtf(pv_quad_sum(sq) * (1.0f / (float)C) + p.ln_eps);
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const float4_t g0 = *reinterpret_cast<const float4_t*>(p.ln_gamma + kk * 32 + g * 8);
            const float4_t g1 = *reinterpret_cast<const float4_t*>(p.ln_gamma + kk * 32 + g * 8 + 4);
            const float4_t b0 = *reinterpret_cast<const float4_t*>(p.ln_beta + kk * 32 + g * 8);
            const float4_t b1 = *reinterpret_cast<const float4_t*>(p.ln_beta + kk * 32 + g * 8 + 4);
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gm = j < 4 ? g0[j & 3] : g1[j & 3], bt = j < 4 ? b0[j & 3] : b1[j & 3];
                    xf[kk][qi][j] = (half_t)(((float)xf[kk][qi][j] - mean[qi]) * rstd[qi] * gm + bt);
                }
        }
    }

    // ---- phase 1: Q^T = Wq . X^T -----------------------------------------------------------------------------------------
    // qf[f][qi]: fp16 of fragment f (features 16f + 4g + r) for this lane's query, pre-scaled by log2(e)/sqrt(d)
    half4_t qf[NFR][2];
    const float qscale = rsqrtf((float)D) * 1.4426950408889634f;
    {
        float4_t acc[5][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nc = t / KT, kt = t % KT;
            if (kt == 0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i][0] = acc[i][1] = float4_t{0.f, 0.f, 0.f, 0.f};
            }
            wait_tiles(NT - 1 - t < 2 ? NT - 1 - t : 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 3 < NT) issue_tile(rq, t + 3);
            const char* sw = smem + (t % S) * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const half8_t a = ld_frag128(sw, i * 16 + fr, ks * 4 + g);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi) acc[i][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[2 * kt + ks][qi], acc[i][qi], 0, 0, 0);
                }
            if (kt == KT - 1) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                        for (int r = 0; r < 4; ++r) qf[nc * 5 + i][qi][r] = (half_t)(acc[i][qi][r] * qscale);
            }
        }
    }

    // ---- phase 2: dual-branch attention, one 80-feature group (two heads) at a time -------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave is done with the weight ring
    asm volatile("" ::: "memory");
    issue_group(0, 0);
    issue_group(1, 1);
    half4_t cf[NFR][2];                    // context fragments, same layout as qf
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        if (grp + 1 < NG) { if (wave < 3) xf_wait_vmcnt<10>(); else xf_wait_vmcnt<9>(); }
        else xf_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* sbuf = smem + (grp & 1) * GROUP_BYTES;
        const half_t* sV = reinterpret_cast<const half_t*>(sbuf + 2 * KIMG_BYTES);
        float4_t o2_h0[2];                 // head 0's fragment 2 (its rows 0-7 are head 0's features 32..39)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const char* sK = sbuf + hh * KIMG_BYTES;
            const int fa = grp * 5 + (hh == 0 ? 0 : 3), fb = fa + 1, f2 = grp * 5 + 2;
            float4_t s[6][2];
#pragma unroll
            for (int kb = 0; kb < 6; ++kb) s[kb][0] = s[kb][1] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8_t bq[2];
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) bq[qi] = ks == 0 ? cat4(qf[fa][qi], qf[fb][qi]) : cat4(qf[f2][qi], qf[f2][qi]);
#pragma unroll
                for (int kb = 0; kb < 6; ++kb) {
                    const half8_t a = ld_frag128(sK, kb * 16 + fr, ks * 4 + g);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi) s[kb][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq[qi], s[kb][qi], 0, 0, 0);
                }
            }
            // two independent softmaxes over the key axis (registers r, fragments kb, and the 4 lane groups)
            half8_t pb[3][2];
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                float mt = -INFINITY, mi = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 6; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb * 16 + g * 4 + r;
                        const float v = s[kb][qi][r];
                        if (key < p.nt) mt = fmaxf(mt, v);
                        if (key >= XIP0 && key < XIP0 + p.nip) mi = fmaxf(mi, v);
                    }
                mt = pv_quad_max(mt);
                mi = pv_quad_max(mi);
                float lt = 0.f, li = 0.f;
#pragma unroll
                for (int kb = 0; kb < 6; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb * 16 + g * 4 + r;
                        float e = 0.f;
                        if (key < p.nt) {
                            e = PV_EXP2(s[kb][qi][r] - mt);
                            lt += e;
                        } else if (key >= XIP0 && key < XIP0 + p.nip) {
                            e = PV_EXP2(s[kb][qi][r] - mi);
                            li += e;
                        }
                        s[kb][qi][r] = e;
                    }
                lt = pv_quad_sum(lt);
                li = pv_quad_sum(li);
                const float ft = w_text / lt, fi = w_ip / li;
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k0 = (2 * s2) * 16 + g * 4 + r, k1 = k0 + 16;
                        pb[s2][qi][r] = (half_t)(s[2 * s2][qi][r] * (k0 < XIP0 ? ft : fi));
                        pb[s2][qi][r + 4] = (half_t)(s[2 * s2 + 1][qi][r] * (k1 < XIP0 ? ft : fi));
                    }
            }
            // O^T = V^T . P^T for this head's three fragments of the group's 80 value columns
            float4_t o[3][2];
#pragma unroll
            for (int fi = 0; fi < 3; ++fi) o[fi][0] = o[fi][1] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
                for (int fi = 0; fi < 3; ++fi) {
                    const half8_t a = vt_frag80(sV, s2 * 32, (hh == 0 ? fi : fi + 2) * 16, fr, g);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi) o[fi][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[s2][qi], o[fi][qi], 0, 0, 0);
                }
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                if (hh == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cf[grp * 5 + 0][qi][r] = (half_t)o[0][qi][r];
                        cf[grp * 5 + 1][qi][r] = (half_t)o[1][qi][r];
                    }
                    o2_h0[qi] = o[2][qi];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cf[grp * 5 + 2][qi][r] = (half_t)(g < 2 ? o2_h0[qi][r] : o[0][qi][r]);
                        cf[grp * 5 + 3][qi][r] = (half_t)o[1][qi][r];
                        cf[grp * 5 + 4][qi][r] = (half_t)o[2][qi][r];
                    }
                }
            }
        }
        if (grp + 2 < NG) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave is done with this buffer
            asm volatile("" ::: "memory");
            issue_group(grp + 2, grp & 1);
        }
    }

    // ---- phase 3: out^T = Wo' . ctx^T, + bias + residual ----------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_tile(ro, 0);
    issue_tile(ro, 1);
    issue_tile(ro, 2);
    {
        half_t* outp = reinterpret_cast<half_t*>(p.out);
        float4_t acc[5][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nc = t / KT, kt = t % KT;
            if (kt == 0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i][0] = acc[i][1] = float4_t{0.f, 0.f, 0.f, 0.f};
            }
            wait_tiles(NT - 1 - t < 2 ? NT - 1 - t : 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 3 < NT) issue_tile(ro, t + 3);
            const char* sw = smem + (t % S) * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8_t bc[2];
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) bc[qi] = cat4(cf[2 * (2 * kt + ks)][qi], cf[2 * (2 * kt + ks) + 1][qi]);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const half8_t a = ld_frag128(sw, i * 16 + fr, ks * 4 + g);
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi) acc[i][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bc[qi], acc[i][qi], 0, 0, 0);
                }
            }
            if (kt == KT - 1) {
                const int nb = nc * GF + g * 4;
                half4_t res[5][2];
                float4_t bias[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    bias[i] = p.bias_o ? *reinterpret_cast<const float4_t*>(p.bias_o + nb + i * 16) : float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi) res[i][qi] = *reinterpret_cast<const half4_t*>(hs + (size_t)mrow[qi] * p.ld_hs + nb + i * 16);
                }
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi) {
                        half4_t ov;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ov[r] = (half_t)(acc[i][qi][r] + bias[i][r] + (float)res[i][qi][r]);
                        *reinterpret_cast<half4_t*>(outp + (size_t)mrow[qi] * p.ld_out + nb + i * 16) = ov;
                    }
            }
        }
    }
}

}  // namespace

extern "C" int pv_xattn_pack_kv(const void* kt, const void* vt, int32_t ldkt, int32_t ldvt, const void* kip, const void* vip, int32_t ldkip,
                                int32_t ldvip, void* kimg, void* vimg, float* vnorm, int32_t batch, int32_t heads, int32_t d, int32_t nt,
                                int32_t nip, void* stream) {
    if (!kt || !vt || !kip || !vip || !kimg || !vimg || batch <= 0 || heads <= 0 || (heads & 1) || d != 40 || nt <= 0 || nt > XIP0 || nip <= 0 ||
        nip > XK - XIP0 || (ldkt % 8) || (ldvt % 8) || (ldkip % 8) || (ldvip % 8))
        return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3(batch * XK), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const half_t*>(kt),
                       reinterpret_cast<const half_t*>(vt), ldkt, ldvt, reinterpret_cast<const half_t*>(kip), reinterpret_cast<const half_t*>(vip),
                       ldkip, ldvip, reinterpret_cast<half_t*>(kimg), reinterpret_cast<half_t*>(vimg), vnorm, heads, nt, nip);
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
    if (!p.hs || !p.wq || !p.wo || !p.kimg || !p.vimg || !p.out || p.batch <= 0 || p.nq <= 0 || (p.nq % 128) || p.d != 40 || C != 320 ||
        p.nt <= 0 || p.nt > XIP0 || p.nip <= 0 || p.nip > XK - XIP0 || (p.ld_hs % 8) || (p.ld_out % 4) || (p.ln_gamma && !p.ln_beta))
        return (int)hipErrorInvalidValue;
    p.w_bytes = (uint32_t)C * C * 2;
    const size_t kb = (size_t)p.batch * p.heads * XK * KROW, vb = (size_t)p.batch * (C / GF) * XK * GF * 2;
    if (kb >= (1ull << 31) || vb >= (1ull << 31)) return (int)hipErrorInvalidValue;
    p.kimg_bytes = (uint32_t)kb;
    p.vimg_bytes = (uint32_t)vb;
    constexpr int SMEM = 2 * (2 * XK * KROW + XK * GF * 2);
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    auto kern = xattn_fused_kernel<320>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)p.batch * p.nq / 128)), dim3(256), SMEM, (hipStream_t)stream, p);
    return PV_CHECK_LAUNCH();
}
