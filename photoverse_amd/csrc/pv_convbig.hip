// pv_convbig: the 3x3 convolutions whose launch fills the chip with 256 x 320 tiles - the 64 x 64 level (M = B * 4096 output pixels,
// N = 320 / 640) and the two x2-upsampling convs (640 -> 640 onto 64 x 64, 1280 -> 1280 onto 32 x 32).
//
// Why a second conv kernel.  The 128 x 160 tile of pv_gemm.hip runs its main loop at the L2 -> LDS gather floor of the tile (DMA-only
// build 94 us of 132 us, EXPERIMENTS.md): what moves it is bytes staged and read per flop.  This tile stages (256 + 320) rows per
// 2 * 256 * 320 flop per k = 142 flop / B (128 x 160: 71) and reads 26 fragments per 80 MFMAs per wave (128 x 160: 18 per 40).
// Price: 144 KiB of LDS -> ONE 8-wave workgroup per CU, so the two waves of a SIMD belong to the same workgroup and run in lock-step -
// nothing hides a wave's DMA issue, LDS latency or epilogue but its own instruction stream, and nothing hides the L2 latency of the
// staging but the depth of the LDS ring.  (Round 4, first form: 64-deep stages in two buffers = ONE stage in flight; in-kernel stamps
// showed every K-step waiting ~1700 of its 4500 cycles for that stage to land, profiles/r04_convbig_v1_stamps.txt.)  Hence:
//
//   * stages are 32 deep (A 16 KiB + W 20 KiB = 36 KiB) in a FOUR-buffer ring: up to three stages (108 KiB) are in flight behind a counted
//     vmcnt while the fourth is read;
//   * the two waves of a SIMD (waves w and w + 4: the two 128-row halves of the tile) run STAGGERED by one barrier interval.  A wave
//     alternates LOAD segments (the 13 fragment reads of a stage + its share of the LDS-DMA issue) with MFMA segments (40 MFMAs: eight
//     row fragments x five column fragments), a barrier between segments; waves 4-7 execute one extra barrier up front, so that while one
//     half of the workgroup holds the matrix pipes the other half reads LDS and issues DMA (MI355X_MICROARCH.md "Two waves per SIMD";
//     the first ring form ran both halves in phase - matrix || matrix then load || load - at 65 % matrix-pipe use);
//   * per stage and wave: LOAD(s) (13 fragment reads, the wave's 4 - 5 LDS-DMA pieces of stage s+3) | barrier | MFMA(s) | barrier;
//     a stage's buffer is free once waves 4-7 finished LOAD(s), i.e. before anybody's LOAD(s+1), where the refill (stage s+4) starts
//     (20-MFMA segments, four barriers per stage, measured 2200 cycles per stage against 1280 of matrix-pipe time: barrier skew per segment);
//   * the K order is pv_gemm.hip's: 64-channel chunk major, filter tap minor, the two 32-deep halves of a (chunk, tap) in sequence - so
//     every accumulator sees the same products in the same order and the results are BIT-IDENTICAL to the 128-row kernel's.
//
// LDS image of a stage: [rows][64 B], one LDS-DMA wave-instruction = 16 rows x 64 B; bank-conflict swizzle on the 16-B chunk index,
// chunk ^= F[(row >> 2) & 3], F = {0, 2, 3, 1}, applied to the per-lane SOURCE offset and again on the ds_read_b128 side (with 64-B rows a
// 256-B bank row holds four rows; F sends the four row-quads a ds_read_b128 lane group touches to four different chunk columns).
// Operand roles, zero padding by out-of-range buffer offsets and the epilogue (bias, time-embedding row, activation, residual, fp16 stores
// widened by v_permlane16_swap, GroupNorm column statistics per 64-row block) are those of pv_gemm.hip.
#include "pv_gemm_dev.h"

#ifndef PV_PATCH_ABLATE
#define PV_PATCH_ABLATE 0   // timing-only builds (wrong results): 1 fixed fragment addresses, 2 no patch pieces, 4 no MFMA / VALU interleave
#endif
#ifdef PV_CONVBIG_STAMPS
// diagnostic build only (tools/diag/convbig_seg_stamps.py): per-wave shader-cycle sums of the four parts of a stage (LOAD segment, wait at the barrier
// behind it, MFMA segment, wait at the barrier behind that) over the whole main loop, for three workgroups
__device__ unsigned long long pv_convbig_stamps[3 * 8 * 8];
extern "C" int pv_convbig_read_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pv_convbig_stamps), sizeof(pv_convbig_stamps)); }
#define CB_DECL unsigned long long cb_acc[4] = {0, 0, 0, 0}, cb_t = __builtin_amdgcn_s_memtime(), cb_u, cb_r0 = __builtin_amdgcn_s_memrealtime(), cb_c0 = cb_t; int cb_n = 0;
#define CB_MARK(i) do { cb_u = __builtin_amdgcn_s_memtime(); cb_acc[i] += cb_u - cb_t; cb_t = cb_u; } while (0)
#define CB_COUNT() (++cb_n)
#define CB_DUMP() do { const int slot_ = blockIdx.x == 0 ? 0 : blockIdx.x == 50 ? 1 : blockIdx.x == 100 ? 2 : -1; \
        if (slot_ >= 0 && lane == 0) { unsigned long long* o_ = pv_convbig_stamps + (slot_ * 8 + wave) * 8; \
            o_[0] = cb_acc[0]; o_[1] = cb_acc[1]; o_[2] = cb_acc[2]; o_[3] = cb_acc[3]; o_[4] = cb_t - cb_c0; o_[5] = __builtin_amdgcn_s_memrealtime() - cb_r0; o_[6] = cb_n; } } while (0)
#else
#define CB_DECL
#define CB_MARK(i)
#define CB_COUNT()
#define CB_DUMP()
#endif

namespace {

constexpr int BK = 32;
constexpr int ROW_BYTES = BK * 2;                   // 64 B per LDS row
constexpr int NW = 8;
constexpr int NBUF = 4;                             // ring: three stages in flight, one being read
// MODE: 0 = 3x3 conv, 1 = Linear / 1x1 (taps = 1), 2 = Linear with the GEGLU gate in the epilogue.  Waves: 2 (M) x 4 (N); a wave owns (16 MI) rows x
// (16 NF) columns; NF = 5 -> 320-column tile, NF = 4 (GEGLU: value | gate fragment pairs, pv_gemm.hip's pack_geglu row order) -> 256-column tile
template <int MI, int NF>
struct BigCfg {
    static constexpr int BM = 2 * MI * 16;              // two wave rows
    static constexpr int BN = 4 * NF * 16;              // four wave columns
    static constexpr int AP = BM / 16 / NW;             // 16-row activation pieces per wave and stage: 2
    static constexpr int B_PIECES = BN / 16;            // 16-row weight pieces per stage: 20 (3 for waves 0-3, 2 for waves 4-7) / 16 (2 each)
    static constexpr int BP = (B_PIECES + NW - 1) / NW;
    static constexpr int BP_LO = B_PIECES % NW ? BP - 1 : BP;    // pieces of the waves 4-7
    static_assert(B_PIECES % NW == 0 || B_PIECES % NW == NW / 2, "the vmcnt bookkeeping assumes waves 0-3 (wm == 0) are the ones with the extra piece");
    static constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;   // 36 KiB / 32 KiB
    static constexpr int SMEM_BYTES = NBUF * STAGE_BYTES;   // 144 KiB / 128 KiB: one workgroup per CU either way
    // PATCH modes (3x3 conv with the LDS-resident input patch): the ring holds weight stages only, the activations of a 32-channel chunk live in one
    // of two patch buffers of 32 LDS-DMA pieces (512 pixel rows of 64 B; (R + 2) x (W + 2) <= 396 of them are the patch)
    static constexpr int PATCH_PIECES = 32;
    static constexpr int PATCH_BYTES = PATCH_PIECES * 16 * ROW_BYTES;                   // 32 KiB
    static constexpr int SMEM_PATCH_BYTES = NBUF * B_BYTES + 2 * PATCH_BYTES;           // 80 + 64 = 144 KiB
    // MODE 5 / 6 (GroupNorm + SiLU of the conv's input folded in): the image's scale / shift table, fp32 [2][cin], behind the patch buffers
    static constexpr int NORM_TABLE_BYTES = 16 * 1024;                                   // cin <= 2048
    static constexpr int SMEM_NORM_BYTES = SMEM_PATCH_BYTES + NORM_TABLE_BYTES;         // 160 KiB: all of a CU's LDS
};

__device__ __forceinline__ float epi_act(float x, int act) {
    if (act == PV_ACT_SILU) return pv_silu(x);
    if (act == PV_ACT_QUICK_GELU) return pv_quick_gelu(x);
    if (act == PV_ACT_LEAKY_RELU) return x > 0.f ? x : 0.01f * x;
    return x;
}

// chunk swizzle of the 64-B-row image: F[(row >> 2) & 3], F = {0, 2, 3, 1}
__device__ __forceinline__ int swz(int rq) { return (0x78 >> (2 * rq)) & 3; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

template <int V>
struct IC { static constexpr int value = V; };

// LN (Linear modes): LayerNorm without its affine part folded into the launch - the GEMM runs on the raw rows, the row statistics are accumulated from
// the A fragments the MFMAs consume (v_dot2_f32_f16: in the shadow of the 40 MFMAs of a segment), the epilogue applies rstd * (acc - mean * rowsum(W)).
template <bool CS, bool UPS, int MI, int MODE, bool LN = false>
__global__ __launch_bounds__(512, 2) void big_tile_kernel(const pv_gemm_params_dev p, const int tiles_n, const int nblk) {
    constexpr int NF = MODE == 2 ? 4 : 5;
    // MODE 3 / 4: the 3x3 conv (stride 1, pad 1, one image row = 64 / 32 pixels) with the LDS-RESIDENT INPUT PATCH: the tile's 256 output pixels are
    // R = 4 / 8 whole image rows; per 32-channel chunk the (R + 2) x (W + 2) input pixels they read are staged ONCE (25 KiB at W = 64) and the nine
    // taps read them at shifted addresses, instead of gathering the shifted 256 rows once per tap (9 x 16 KiB).  L2 -> LDS bytes per chunk:
    // 25 + 9 x 20 = 205 KiB instead of 9 x 36 = 324 KiB, LDS-DMA instructions per stage and wave 2.9 instead of 4.5.  K order: 32-channel chunk
    // major, tap minor (the 128-row kernel: 64-channel chunk major) - NOT bit-identical to it.
    // MODE 5 / 6 = MODE 3 / 4 with the GroupNorm (+ SiLU) of the conv's INPUT folded in (pv_gemm_params.a_norm): the patch is the one place where every
    // input pixel of a tile passes exactly once, so each wave normalises the pieces it staged, IN PLACE in LDS (ds_read 16 B, pv_groupnorm_apply's fp32
    // arithmetic, ds_write), four stages after their LDS-DMA and under the MFMAs of a segment; padding pixels stay zero.  The 64 x 64 level's
    // GroupNorm-apply launches (write + re-read of every conv input) disappear.
    constexpr bool PATCH = MODE >= 3, NORMA = MODE >= 5;
    constexpr bool IS_CONV = MODE == 0 || PATCH;
    constexpr int LOG2W = (MODE == 3 || MODE == 5) ? 6 : 5, PW = 1 << LOG2W, PR = 256 >> LOG2W, PS = PW + 2, NPIX = (PR + 2) * PS;
    constexpr int TAPS = IS_CONV ? 9 : 1;
    using Cfg = BigCfg<MI, NF>;
    static_assert(!PATCH || (MI == 8 && !UPS && !LN), "the patch modes are the plain 256-row conv");
    static_assert(NPIX <= Cfg::PATCH_PIECES * 16, "the patch fits its 32 pieces");
    constexpr int BM = Cfg::BM, BN = Cfg::BN, AP = Cfg::AP, BP = Cfg::BP, B_PIECES = Cfg::B_PIECES, A_BYTES = Cfg::A_BYTES, STAGE_BYTES = Cfg::STAGE_BYTES;
    constexpr int P_HI = AP + BP, P_LO = AP + Cfg::BP_LO;      // LDS-DMA instructions per stage of a wave 0-3 / 4-7
    static_assert(!UPS || MODE == 0, "the upsampling gather belongs to the conv");
    static_assert(!LN || (MODE != 0 && !CS), "the LayerNorm fold exists for the Linear modes (their outputs feed attention / a Linear: no column statistics)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = pv_lane_id();
    const int wave = pv_wave_id();
    const int bid = pv_xcd_remap((int)blockIdx.x, nblk);
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int cin = p.c0 + p.c1;
    const int K = TAPS * cin;
    // split-K (pv_gemm.hip's partition: blockIdx.y owns the 64-deep K-steps [kb, kb + nk_per), fp32 partial slab, fixed-order reduce launch)
    const int nk64 = K / 64;
    const int nk_per = (nk64 + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kb = (int)blockIdx.y * nk_per;
    const int ns = 2 * max(0, min(nk64, kb + nk_per) - kb);   // stages of this workgroup: two per (64-channel chunk, tap); stage index s is relative to kb

    constexpr unsigned OOB = 0x80000000u;
    const int hw_out = p.hout * p.wout;
    const int prow = lane >> 2;                      // row inside a 16-row piece
    float4_t acc[NF][MI];
#pragma unroll
    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = float4_t{0.f, 0.f, 0.f, 0.f};

    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int arow = wm * (MI * 16) + fr;            // + mi * 16
    const int brow = wn * (NF * 16) + fr;            // + ni * 16
    const int frag_off = fr * ROW_BYTES + ((fq ^ swz((fr >> 2) & 3)) << 4);   // fragment row bases are multiples of 16: the swizzle only sees fr

    half8_t wb[NF], xa[MI];
    float ln_s1[LN ? MI : 1], ln_s2[LN ? MI : 1];
#pragma unroll
    for (int mi = 0; mi < (LN ? MI : 1); ++mi) ln_s1[mi] = ln_s2[mi] = 0.f;
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (!PATCH) {
        // ---- staging geometry: stride 1, pad 1; UPS: the logical input is the x2 nearest upsample of (hin, win), i.e. tap (ky, kx) of output
        // pixel (y, x) reads source pixel ((y + ky - 1) >> 1, (x + kx - 1) >> 1) - relative to the centre's source pixel (y >> 1, x >> 1) that is
        // a row step of -1 / 0 (even y) or 0 / +1 (odd y) for ky = 0 / 2, likewise in x: the per-lane offset is the centre's plus two selects ----
        const int lane_cc2 = ((lane & 3) ^ swz((lane >> 4) & 3)) * 16;   // swizzled source chunk: byte offset inside the 32-channel slab
        const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a0), 0, (int)p.a0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a1 ? p.a1 : p.a0), 0, (int)p.a1_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
        unsigned a_off0[AP], a_off1[AP], a_mask[AP];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = m0 + (wave + i * NW) * 16 + prow;
            const bool ok = m < p.M;
            if (MODE != 0) {                               // Linear (single source): the row itself, or out of range past M; no masks, no second source
                a_off0[i] = ok ? (unsigned)m * (unsigned)(p.lda0 * 2) + lane_cc2 : OOB;
                a_off1[i] = a_mask[i] = 0;
                continue;
            }
            const int b = m / hw_out;
            const int rem = m - b * hw_out;
            const int y = rem / p.wout, x = rem - y * p.wout;
            const int hl = UPS ? 2 * p.hin : p.hin, wl = UPS ? 2 * p.win : p.win;       // logical input extent
            unsigned mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
                mask |= (ok && iy >= 0 && iy < hl && ix >= 0 && ix < wl) ? 1u << t : 0u;
            }
            if (UPS) mask |= (unsigned)(y & 1) << 9 | (unsigned)(x & 1) << 10;          // parities select the tap's source step
            a_mask[i] = mask;
            const unsigned pix = UPS ? (unsigned)((b * p.hin + (y >> 1)) * p.win + (x >> 1)) : (unsigned)((b * p.hin + y) * p.win + x);
            a_off0[i] = pix * (unsigned)(p.lda0 * 2) + lane_cc2;
            a_off1[i] = pix * (unsigned)(p.lda1 * 2) + lane_cc2;
        }
        unsigned w_off[BP];
#pragma unroll
        for (int i = 0; i < BP; ++i) w_off[i] = (unsigned)(n0 + min(wave + i * NW, B_PIECES - 1) * 16 + prow) * (unsigned)(K * 2) + lane_cc2;
        const bool b_full = wave + (BP - 1) * NW < B_PIECES;          // 320-column tile: waves 0-3 (wm == 0) issue a third weight piece

        // K position of a stage: s -> (64-channel chunk, filter tap, 32-deep half), advanced incrementally in the loop (the closed form costs
        // two divisions by constants = a dozen dependent scalar multiplies per stage in front of every LDS-DMA issue)
        struct KPos { int s, chunk, tap, ky, kx; };
        auto kpos_of = [&](int s) {
            const int g = kb + (s >> 1), chunk = g / TAPS, tap = g - chunk * TAPS, ky = TAPS == 9 ? tap / 3 : 1;
            return KPos{s, chunk, tap, ky, TAPS == 9 ? tap - ky * 3 : 1};      // Linear: the "centre tap" (ky, kx) = (1, 1): no pixel shift
        };
        auto kpos_next = [&](KPos& k) {
            if (k.s & 1) {                                 // second half done: next tap (kx fastest), then next chunk
                if (TAPS == 9) {
                    ++k.tap; ++k.kx;
                    if (k.kx == 3) { k.kx = 0; ++k.ky; }
                    if (k.tap == 9) { k.tap = 0; k.ky = 0; ++k.chunk; }
                } else {
                    ++k.chunk;
                }
            }
            ++k.s;
        };
        // pieces [J0, J1) of the stage at K position k into buffer k.s & 3; piece j < AP: activation piece j, else weight piece j - AP
        auto issue = [&](const KPos& k, auto j0c, auto j1c) {
            constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
            char* sa = smem + (k.s & (NBUF - 1)) * STAGE_BYTES;
            char* sb = sa + A_BYTES;
            const int c = k.chunk * 64 + (k.s & 1) * BK;
            const bool first = c < p.c0;
            const __amdgpu_buffer_rsrc_t ra = first ? ra0 : ra1;
            const int ld2 = (first ? p.lda0 : p.lda1) * 2;
            const int sc2 = (first ? c : c - p.c0) * 2;
            const int tap_delta = ((k.ky - 1) * p.win + (k.kx - 1)) * ld2 + sc2;
            // UPS: source step of this tap for even / odd output coordinates
            const int oy_even = (k.ky == 0 ? -p.win : 0) * ld2, oy_odd = (k.ky == 2 ? p.win : 0) * ld2;
            const int ox_even = (k.kx == 0 ? -1 : 0) * ld2, ox_odd = (k.kx == 2 ? 1 : 0) * ld2;
            const unsigned wk2 = (unsigned)(k.tap * cin + c) * 2u;
            const unsigned tap_bit = 1u << k.tap;
#pragma unroll
            for (int j = J0; j < J1; ++j) {
                if (j < AP && MODE != 0) {
                    const unsigned off = a_off0[j] == OOB ? OOB : a_off0[j] + (unsigned)(c * 2);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra0, PV_LDS_PTR(sa + (wave + j * NW) * 16 * ROW_BYTES), 16, (int)off, 0, 0, 0);
                } else if (j < AP) {
                    unsigned off = first ? a_off0[j] : a_off1[j];
                    if (UPS) off += (unsigned)(((a_mask[j] >> 9) & 1u ? oy_odd : oy_even) + ((a_mask[j] >> 10) & 1u ? ox_odd : ox_even) + sc2);
                    else off += (unsigned)tap_delta;
                    off = (a_mask[j] & tap_bit) ? off : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, PV_LDS_PTR(sa + (wave + j * NW) * 16 * ROW_BYTES), 16, (int)off, 0, 0, 0);
                } else if (j - AP < BP - 1 || b_full) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, PV_LDS_PTR(sb + (wave + (j - AP) * NW) * 16 * ROW_BYTES), 16, (int)(w_off[j - AP] + wk2), 0, 0, 0);
                }
            }
        };

        auto read_frags = [&](int s) {                   // all 13 fragments of stage s: 5 column (W) + 8 row (A)
            const char* sa = smem + (s & (NBUF - 1)) * STAGE_BYTES + wm * (MI * 16) * ROW_BYTES + frag_off;
            const char* sb = smem + (s & (NBUF - 1)) * STAGE_BYTES + A_BYTES + wn * (NF * 16) * ROW_BYTES + frag_off;
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) wb[ni] = *reinterpret_cast<const half8_t*>(sb + ni * 16 * ROW_BYTES);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xa[mi] = *reinterpret_cast<const half8_t*>(sa + mi * 16 * ROW_BYTES);
        };
        // "stage s+1 landed" <=> at most the pieces of the two younger issued stages (s+2, s+3) are outstanding.  A wave counts its own pieces:
        // P_HI per stage for waves 0-3 (wm == 0), P_LO for waves 4-7 (320-column tile: 5 / 4).  The last stages (nothing younger in flight) drain fully.
        // ---- prologue: stages 0, 1, 2 (the LOAD segment of stage s issues stage s+3) ----
        {
            const int pre = min(ns, 3);
            for (int s = 0; s < pre; ++s) issue(kpos_of(s), IC<0>{}, IC<AP + BP>{});
            if (pre == 3) { if (wm == 0) wait_vmcnt<2 * P_HI>(); else wait_vmcnt<2 * P_LO>(); }   // stage 0 landed
            else wait_vmcnt<0>();
        }
        seg_barrier();
        if (wm == 1) seg_barrier();                      // the stagger: waves 4-7 run one barrier interval behind waves 0-3

        KPos kn = kpos_of(3);                            // the stage the next LOAD segment issues
        CB_DECL
        for (int s = 0; s < ns; ++s) {
            CB_COUNT();
            // ---- LOAD(s): waves 0-3 in interval 2s, waves 4-7 in interval 2s+1 (after which buffer s & 3 is free: stage s+4 goes there) ----
            read_frags(s);
            if (s + 3 < ns) issue(kn, IC<0>{}, IC<AP + BP>{});
            kpos_next(kn);
            if (wm == 1) {                                           // waves 4-7: interval 2s+1 is the one in front of waves 0-3's LOAD(s+1)
                if (s + 3 < ns) wait_vmcnt<2 * P_LO>(); else wait_vmcnt<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragments in registers BEFORE the barrier: the buffer's reads are retired when it is refilled
            CB_MARK(0);
            seg_barrier();
            CB_MARK(1);
            // ---- MFMA(s): 40 MFMAs (LN: + the row sums of the fragments, 8 v_dot2 per fragment) ----
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NF; ++ni)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
                if constexpr (LN) {
                    const half2_t one2 = half2_t{(half_t)1.0f, (half_t)1.0f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const half2_t x2 = half2_t{xa[mi][2 * j], xa[mi][2 * j + 1]};
                        ln_s1[mi] = __builtin_amdgcn_fdot2(x2, one2, ln_s1[mi], false);
                        ln_s2[mi] = __builtin_amdgcn_fdot2(x2, x2, ln_s2[mi], false);
                    }
                }
            }
            if (wm == 0) {                                           // waves 0-3: stage s+1 is read right behind the next barrier
                if (s + 3 < ns) wait_vmcnt<2 * P_HI>(); else wait_vmcnt<0>();
            }
            CB_MARK(2);
            seg_barrier();
            CB_MARK(3);
        }
        CB_DUMP();
        if (wm == 0) seg_barrier();                      // waves 0-3 execute as many barriers as waves 4-7

    } else {
        // ================= 3x3 conv with the LDS-resident input patch (MODE 3: 64-pixel image rows, MODE 4: 32) =================
        constexpr int B_BYTES = Cfg::B_BYTES, PATCH_BYTES = Cfg::PATCH_BYTES;
        constexpr int PB_HI = BP, PB_LO = Cfg::BP_LO;              // weight pieces per stage: waves 0-3 / 4-7
        char* const s_patch = smem + NBUF * B_BYTES;
        const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a0), 0, (int)p.a0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a1 ? p.a1 : p.a0), 0, (int)p.a1_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
        const int nchunk = cin / BK;                                // 32-channel chunks; stage s = chunk * 9 + tap
        const int nst = 9 * nchunk;
        // the tile = PR whole image rows of one image: rows y0 .. y0 + PR - 1; patch pixel (py, px) = image pixel (y0 - 1 + py, px - 1)
        const int img = m0 / hw_out, y0 = (m0 - img * hw_out) >> LOG2W;
        // LDS image of a patch buffer: [pixel][64 B], the 16-B chunk of pixel pp at position chunk ^ (2 * ((pp >> 2) & 1)).  This swizzle (period two
        // pixel quads) keeps a ds_read_b128 fragment of 16 CONSECUTIVE pixels conflict-free at ANY pixel alignment - the taps shift the fragment by
        // kx + ky * (W + 2) pixels; the F swizzle of the gathered image above is conflict-free at multiples of 16 only.
        unsigned ppix[4];                                           // this lane's pixel in patch pieces wave, wave + 8, wave + 16, wave + 24
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pp = 16 * (wave + 8 * i) + prow;
            const int py = pp / PS, px = pp - py * PS;
            const int iy = y0 - 1 + py, ix = px - 1;
            const bool ok = pp < NPIX && iy >= 0 && iy < p.hin && ix >= 0 && ix < PW;
            ppix[i] = ok ? (unsigned)((img * p.hin + iy) * PW + ix) : OOB;
        }
        const unsigned patch_cc = (unsigned)(((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16);     // (pp >> 2) & 1 == (lane >> 4) & 1: pieces start at multiples of 16
        const int lane_cc2 = ((lane & 3) ^ swz((lane >> 4) & 3)) * 16;                        // weight pieces: the F swizzle, as above
        // weight piece wave + 8 j of a stage: ONE per-lane offset (piece `wave`); the piece step (128 weight rows) and the stage's K position ride in the
        // SCALAR offset - per-lane offsets per piece and tap would be loop invariants that hipcc hoists and then spills next to the 160 accumulators
        const unsigned w_off = (unsigned)(n0 + wave * 16 + prow) * (unsigned)(K * 2) + lane_cc2;
        const int w_piece_step = NW * 16 * K * 2;
        const bool b_full = wave + (BP - 1) * NW < B_PIECES;
        auto issue_w = [&](int s, int chunk, int tap) {            // the weight stage of (chunk, tap) into ring buffer s & 3
            char* sb = smem + (s & (NBUF - 1)) * B_BYTES;
            const int wk2 = (tap * cin + chunk * BK) * 2;
#pragma unroll
            for (int j = 0; j < BP; ++j)
                if (j < BP - 1 || b_full)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, PV_LDS_PTR(sb + (wave + j * NW) * 16 * ROW_BYTES), 16, (int)w_off, wk2 + j * w_piece_step, 0, 0);
        };
        // (loop invariants pinned in scalar registers: left to itself hipcc re-loads them from the kernel arguments inside the loop, and the
        // s_waitcnt lgkmcnt(0) in front of their first use also waits for the thirteen fragment reads in flight)
        int ld2_0 = p.lda0 * 2, ld2_1 = p.lda1 * 2, c0s = p.c0;
        asm volatile("" : "+s"(ld2_0), "+s"(ld2_1), "+s"(c0s));
        // piece wave + 8 i of the patch of a chunk.  The pixel of piece i is taken from ppix[0] and the four registers ROTATE (five v_mov): selecting
        // ppix[i] by the wave-uniform tap compiles to a chain of eight scalar branches, ~400 cycles per piece (ablation: profiles/r05_conv_patch_ablate.txt)
        auto issue_patch = [&](int chunk, int i) {
            const int c = chunk * BK;
            const bool first = c < c0s;
            const unsigned ld2 = (unsigned)(first ? ld2_0 : ld2_1), sc2 = (unsigned)(first ? c : c - c0s) * 2u;
            const unsigned px = ppix[0];
            ppix[0] = ppix[1]; ppix[1] = ppix[2]; ppix[2] = ppix[3]; ppix[3] = px;
            const unsigned off = px == OOB ? OOB : px * ld2 + sc2 + patch_cc;
            char* dst = s_patch + (chunk & 1) * PATCH_BYTES + (wave + 8 * i) * 16 * ROW_BYTES;
            const __amdgpu_buffer_rsrc_t ra = first ? ra0 : ra1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, PV_LDS_PTR(dst), 16, (int)off, 0, 0, 0);
        };
        // A fragment mi of this wave = 16 consecutive output pixels of ONE image row: row (wm * 128 + mi * 16) >> LOG2W of the tile, x0 = (mi * 16) & (W - 1);
        // at tap (ky, kx) lane fr reads patch pixel (row + ky) * PS + x0 + fr + kx, 16-B chunk fq
        const int lane_pp = (wm * (128 >> LOG2W)) * PS + fr;
        // the wave's 128 rows = NROW image rows; the fragments of one image row are 16 pixels = 1 KiB apart and share the swizzle bit
        // (a step of 16 pixels keeps (pp >> 2) & 1): ONE address per image row, the rest are ds_read immediates
        constexpr int NROW = (MI * 16) >> LOG2W, FPR = PW / 16;    // image rows per wave, fragments per image row
        unsigned a_row[NROW];                                       // byte addresses of the NEXT stage's A fragments (computed under the MFMAs)
        auto frag_addrs = [&](int chunk, int tapshift) {           // tapshift = ky * PS + kx
            const int base = lane_pp + tapshift;
            const unsigned buf = (unsigned)(NBUF * B_BYTES + (chunk & 1) * PATCH_BYTES);
#pragma unroll
            for (int r = 0; r < NROW; ++r) {
                const int pp = base + r * PS;
                a_row[r] = buf + (unsigned)pp * ROW_BYTES + (unsigned)((fq ^ ((pp >> 1) & 2)) << 4);
            }
        };
        auto read_frags_p = [&](int s) {
            const char* sb = smem + (s & (NBUF - 1)) * B_BYTES + wn * (NF * 16) * ROW_BYTES + frag_off;
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) wb[ni] = *reinterpret_cast<const half8_t*>(sb + ni * 16 * ROW_BYTES);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xa[mi] = *reinterpret_cast<const half8_t*>(smem + a_row[mi / FPR] + (mi % FPR) * 16 * ROW_BYTES);
        };
        // NORMA: normalise, in place, the patch piece this wave staged (its pixel is ppix[0]; the registers rotate as in issue_patch): lane = (pixel,
        // 16-B position); the position holds data chunk pos ^ swizzle = 8 channels c8 .. c8 + 7 of the chunk; y = act(x * scale + shift) with
        // pv_groupnorm_apply's expressions; lanes of padding pixels leave the zeros of the LDS-DMA alone
        float* const s_tab = reinterpret_cast<float*>(smem + Cfg::SMEM_PATCH_BYTES);       // [2][cin]
        const int lane_dc8 = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 8;
        auto norm_piece = [&](int chunk, int i) {
            const unsigned px = ppix[0];
            ppix[0] = ppix[1]; ppix[1] = ppix[2]; ppix[2] = ppix[3]; ppix[3] = px;
            char* at = s_patch + (chunk & 1) * PATCH_BYTES + (wave + 8 * i) * 16 * ROW_BYTES + lane * 16;
            const half8_t v = *reinterpret_cast<const half8_t*>(at);
            const float* sc = s_tab + chunk * BK + lane_dc8;
            const float4_t s0 = *reinterpret_cast<const float4_t*>(sc), s1 = *reinterpret_cast<const float4_t*>(sc + 4);
            const float4_t h0 = *reinterpret_cast<const float4_t*>(sc + cin), h1 = *reinterpret_cast<const float4_t*>(sc + cin + 4);
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (float)v[j] * (j < 4 ? s0[j & 3] : s1[j & 3]) + (j < 4 ? h0[j & 3] : h1[j & 3]);
                if (p.a_norm_act == PV_ACT_SILU) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
                o[j] = (half_t)f;
            }
            if (px != OOB) *reinterpret_cast<half8_t*>(at) = o;
        };
        if constexpr (NORMA) {
            // the image's table into LDS (issued FIRST: the oldest vector-memory operations, covered by every later vmcnt wait)
            const float* tab = p.a_norm + (size_t)img * 2 * cin;
            for (int i = (int)threadIdx.x; i < 2 * cin; i += NW * 64) s_tab[i] = tab[i];
        }
        // ---- prologue: the first chunk's patch (all four pieces), weight stages 0, 1, 2 (taps 0-2 of chunk 0: a conv has >= 9 stages) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_patch(0, i);
        for (int s = 0; s < 3; ++s) issue_w(s, 0, s);
        if (wm == 0) wait_vmcnt<2 * PB_HI>(); else wait_vmcnt<2 * PB_LO>();       // patch 0 and stage 0 landed
        if constexpr (NORMA) {
            __syncthreads();                                        // the table is in LDS
#pragma unroll
            for (int i = 0; i < 4; ++i) norm_piece(0, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // written back before the barrier below releases the readers
        }
        frag_addrs(0, 0);
        seg_barrier();
        if (wm == 1) seg_barrier();                                 // the stagger: waves 4-7 one interval behind

        // vmcnt bookkeeping: LOAD(s) issues the weight stage s + 3 and THEN, in the first four taps of a chunk, one piece of the NEXT chunk's patch (a patch
        // buffer is free once the late half finished LOAD of the previous chunk's last tap, i.e. before anybody's LOAD of this chunk's first tap).  "weight
        // stage s + 1 landed" <=> at most the pieces issued in LOAD(s) and LOAD(s - 1) are outstanding
        // (the wait is for 2 PB outstanding whatever rode along - up to three patch pieces too many have to land: choosing the exact immediate by
        // e(s) + e(s - 1) + e(s - 2) costs a chain of scalar branches per stage, ~150 cycles, more than the patch pieces' early landing)
        int chunk = 0, tap = 0, kx = 0, tapshift = 0;               // K position of stage s: tapshift = ky * PS + kx
        int c3 = 0, t3 = 3;                                         // ... and of stage s + 3
        CB_DECL
        for (int s = 0; s < nst; ++s) {
            CB_COUNT();
            // ---- LOAD(s) ----
            read_frags_p(s);
#if PV_PATCH_ABLATE == 2
            const int e_now = 0;
#else
            const int e_now = (tap < 4 && chunk + 1 < nchunk) ? 1 : 0;
#endif
            if (s + 3 < nst) issue_w(s + 3, c3, t3);
            if (e_now) issue_patch(chunk + 1, tap);
            if (++t3 == 9) { t3 = 0; ++c3; }
            if (wm == 1) {
                if (s + 3 < nst) wait_vmcnt<2 * PB_LO>(); else wait_vmcnt<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            CB_MARK(0);
            seg_barrier();
            CB_MARK(1);
            // ---- MFMA(s): 40 MFMAs; the next stage's fragment addresses ride in their shadow ----
            ++tap; ++kx; ++tapshift;
            if (kx == 3) { kx = 0; tapshift += PS - 3; }
            if (tap == 9) { tap = 0; tapshift = 0; ++chunk; }
#if PV_PATCH_ABLATE != 1
            frag_addrs(chunk, tapshift);
#endif
            if constexpr (NORMA) {
                // taps 4-7 of a chunk (tap was advanced above: 5-8): piece tap - 5 of the NEXT chunk's patch landed at least two stages ago.  In a block of
                // its own in FRONT of the MFMAs: ~600 cycles per piece (LDS round trip + 75 VALU, nothing under them).  Interleaved with the MFMAs (a
                // second copy of the MFMA block, sched_group_barrier) the temporaries do not fit next to 160 accumulators + 52 fragment registers:
                // 536 B of scratch.  This is why the fold is OFF by default (ops.Recorder.GN_FOLD).
                if (tap >= 5 && chunk + 1 < nchunk) norm_piece(chunk + 1, tap - 5);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NF; ++ni)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NROW; ++r) asm volatile("" : "+v"(a_row[r]));      // computed HERE (hipcc sinks them behind the vmcnt branches below)
            // the address arithmetic and the K-position bookkeeping go BETWEEN the MFMAs (an MFMA holds the issue port 8 cycles in 16): left to
            // itself hipcc puts them behind the fortieth MFMA, where they lengthen the segment by what they cost (stamps: +150 cycles)
#if PV_PATCH_ABLATE != 4
#pragma unroll
            for (int g = 0; g < MI; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
            }
#endif
            if (wm == 0) {
                if (s + 3 < nst) wait_vmcnt<2 * PB_HI>(); else wait_vmcnt<0>();
            }
            CB_MARK(2);
            seg_barrier();
            CB_MARK(3);
        }
        CB_DUMP();
        if (wm == 0) seg_barrier();
    }

    const int nbase = n0 + wn * (NF * 16) + fq * 4;
    if (gridDim.y > 1) {   // split-K partial: raw fp32 accumulators into this split's slab
        float* slab = p.splitk_ws + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + arow + mi * 16;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) *reinterpret_cast<float4_t*>(slab + (size_t)m * p.N + nbase + ni * 16) = acc[ni][mi];
        }
        return;
    }
    // LN: mean / rstd of this lane's rows (the four lanes that share fr hold the four 8-channel groups of every 32-deep stage)
    // (in place: ln_s1 becomes the mean, ln_s2 the reciprocal standard deviation)
    if constexpr (LN) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const float a = pv_quad_sum(ln_s1[mi]) / (float)K, q2 = pv_quad_sum(ln_s2[mi]) / (float)K;
            ln_s1[mi] = a;
            ln_s2[mi] = rsqrtf(fmaxf(q2 - a * a, 0.f) + p.ln_eps);
        }
    }
    auto ln_fold = [&](float4_t v, const float4_t rs, int mi) {       // rstd * (acc - mean * rowsum)
        if constexpr (LN) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ln_s2[mi] * (v[r] - ln_s1[mi] * rs[r]);
        }
        return v;
    };
    if constexpr (MODE == 2) {
        // ---- GEGLU epilogue (pv_gemm.hip's): the wave's four fragments are (value, gate) x two 16-column groups in pack_geglu's row order;
        // out[m][(n0 >> 1) + 32 wn + 16 q + ...] = (value + bias) * gelu(gate + bias), 16-byte stores through the lane-row swap ----
        float4_t bv[2], bg[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bv[q] = p.bias ? *reinterpret_cast<const float4_t*>(p.bias + nbase + (2 * q) * 16) : float4_t{0.f, 0.f, 0.f, 0.f};
            bg[q] = p.bias ? *reinterpret_cast<const float4_t*>(p.bias + nbase + (2 * q + 1) * 16) : float4_t{0.f, 0.f, 0.f, 0.f};
        }
        float4_t rv[2], rg[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            rv[q] = LN ? *reinterpret_cast<const float4_t*>(p.ln_rowsum + nbase + (2 * q) * 16) : float4_t{0.f, 0.f, 0.f, 0.f};
            rg[q] = LN ? *reinterpret_cast<const float4_t*>(p.ln_rowsum + nbase + (2 * q + 1) * 16) : float4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + arow + mi * 16;
            if (m >= p.M) continue;
            unsigned pk[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4_t v = ln_fold(acc[2 * q][mi], rv[q], mi) + bv[q], g = ln_fold(acc[2 * q + 1][mi], rg[q], mi) + bg[q];
                pk[q][0] = __builtin_bit_cast(unsigned, half2_t{(half_t)(v[0] * pv_gelu_erf(g[0])), (half_t)(v[1] * pv_gelu_erf(g[1]))});
                pk[q][1] = __builtin_bit_cast(unsigned, half2_t{(half_t)(v[2] * pv_gelu_erf(g[2])), (half_t)(v[3] * pv_gelu_erf(g[3]))});
            }
            half_t* orow = reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + (n0 >> 1) + wn * (NF * 8);
            const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            const unsigned a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
            const int col = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
            typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
            *reinterpret_cast<uint4_t*>(orow + col) = uint4_t{a0, a1, b0, b1};
        }
        return;
    }
    // ---- epilogue (pv_gemm.hip's arithmetic, element for element).  All 256 workgroups of a launch reach it together and 42 MB leave at once: it runs at
    // the chip's store bandwidth (~21 k cycles), PROVIDED the stores start with the first row and each wave writes whole 160-byte row pieces: walking
    // column groups (64 + 64 + 32-byte pieces at three different times) measured 33-37 k cycles, staging the tile through LDS for 640-byte row stores
    // (arithmetic and stores serialised by barriers) 42 k.  So: ONE pass per 64-row block over all five fragments, and what used to spill next to the 160
    // accumulators - bias, LayerNorm row sums, the block's time-embedding row - is staged once in the released ring and re-read per row. ----
    float* s_bias = reinterpret_cast<float*>(smem);           // [BN]
    float* s_rs = s_bias + BN;                                // [BN] (LN)
    float* s_radd = s_rs + BN;                                // [2 * MI / 4][BN]: block (wm, hb) -> its image's row (MODE 0)
    static_assert((2 + 2 * (MI / 4)) * BN * 4 <= BigCfg<MI, NF>::SMEM_BYTES, "the epilogue staging must fit the ring");
    if constexpr (NF == 5) {
        const int tid = (int)threadIdx.x;
        for (int i = tid; i < BN; i += NW * 64) {
            s_bias[i] = p.bias ? p.bias[n0 + i] : 0.f;
            if (LN) s_rs[i] = p.ln_rowsum[n0 + i];
        }
        if (IS_CONV && p.rowadd) {
            for (int i = tid; i < 2 * (MI / 4) * BN; i += NW * 64) {
                const int blk = i / BN, c = i - blk * BN;
                const int img = min(m0 + blk * 64, p.M - 1) / hw_out;
                s_radd[i] = p.rowadd[(size_t)img * p.rowadd_ld + n0 + c];
            }
        }
        __syncthreads();
    }
    const int ncol = wn * (NF * 16) + fq * 4;                  // this lane's first column inside the tile
    const bool want_cs = CS && p.colstats != nullptr;
    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
    auto block = [&](int hb) {
        // time-embedding row: one per IMAGE.  When the block's 64 rows lie inside one image (always, for hw_out % 64 == 0) it is the staged one
        // (added AFTER the bias, per element, as pv_gemm.hip does: the results stay bit-identical)
        const int mb0 = m0 + wm * (MI * 16) + hb * 64;
        const bool one_image = IS_CONV && p.rowadd && (mb0 / hw_out) == (min(mb0 + 63, p.M - 1) / hw_out);   // (the Linear modes keep the per-row form: no UNet Linear has a row term)
        const float* s_ra = s_radd + (wm * (MI / 4) + hb) * BN + ncol;
        constexpr int RBUF = CS ? 1 : 2;               // residual rows: one in use (+ one in flight, unless the 40 statistics registers are live too)
        half4_t res[RBUF][NF];
        auto fetch_res = [&](int q) {
            const int mm = min(m0 + arow + (hb * 4 + q) * 16, p.M - 1);
#pragma unroll
            for (int t = 0; t < NF; ++t)
                res[q % RBUF][t] = *reinterpret_cast<const half4_t*>(reinterpret_cast<const half_t*>(p.residual) + (size_t)mm * p.ldr + nbase + t * 16);
        };
        if (RBUF == 2 && p.residual) fetch_res(0);
        float4_t cs[NF], cq[NF];
#pragma unroll
        for (int t = 0; t < NF; ++t) cs[t] = cq[t] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mi = hb * 4 + q;
            if (p.residual && (RBUF == 1 || q + 1 < 4)) fetch_res(RBUF == 1 ? q : q + 1);
            const int m = m0 + arow + mi * 16;
            if (m >= p.M) continue;
            asm volatile("" ::: "memory");             // the staged operands are RE-READ per row (held across the rows they would spill)
            const float* radd = (p.rowadd && !one_image) ? p.rowadd + (size_t)(m / hw_out) * p.rowadd_ld + nbase : nullptr;
            half_t* orow = reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + n0 + wn * (NF * 16);
            auto value = [&](int t, unsigned (&pk)[2]) {
                float4_t v = acc[t][mi];
                if constexpr (LN) v = ln_fold(v, *reinterpret_cast<const float4_t*>(s_rs + ncol + t * 16), mi);
                v += *reinterpret_cast<const float4_t*>(s_bias + ncol + t * 16);
                if (IS_CONV && one_image) v += *reinterpret_cast<const float4_t*>(s_ra + t * 16);
                else if (radd) v += *reinterpret_cast<const float4_t*>(radd + t * 16);
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = epi_act(v[r], p.act);
                }
                if (p.residual) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)res[q % RBUF][t][r];
                }
                const half4_t hv = half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                pk[0] = __builtin_bit_cast(unsigned, half2_t{hv[0], hv[1]});
                pk[1] = __builtin_bit_cast(unsigned, half2_t{hv[2], hv[3]});
                if (want_cs) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float f = (float)hv[r];
                        cs[t][r] += f;
                        cq[t][r] += f * f;
                    }
                }
            };
#pragma unroll
            for (int t = 0; t + 1 < NF; t += 2) {   // a fragment pair at a time (the row's 160 bytes still leave within a few hundred cycles): 16-byte stores -
                unsigned pa[2], pb[2];              // v_permlane16_swap trades the even lane rows' second fragment against the odd lane rows' first
                value(t, pa);
                value(t + 1, pb);
                const auto r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                const unsigned a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
                const int col = (fq & 1) ? (t + 1) * 16 + (fq - 1) * 4 : t * 16 + fq * 4;
                *reinterpret_cast<uint4_t*>(orow + col) = uint4_t{a0, a1, b0, b1};
                asm volatile("" ::: "memory");
            }
            if constexpr (NF & 1) {                 // the odd fifth fragment: 8-byte stores
                unsigned pa[2];
                value(NF - 1, pa);
                *reinterpret_cast<uint2_t*>(orow + (NF - 1) * 16 + fq * 4) = uint2_t{pa[0], pa[1]};
            }
        }
        if (want_cs && mb0 < p.M) {
#pragma unroll
            for (int t = 0; t < NF; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cs[t][r] = row16_sum(cs[t][r]);
                    cq[t][r] = row16_sum(cq[t][r]);
                }
            if (fr == 0) {
                float* dst = p.colstats + ((size_t)(m0 / 64 + wm * (MI / 4) + hb) * 2) * p.N + nbase;
#pragma unroll
                for (int t = 0; t < NF; ++t) {
                    *reinterpret_cast<float4_t*>(dst + t * 16) = cs[t];
                    *reinterpret_cast<float4_t*>(dst + p.N + t * 16) = cq[t];
                }
            }
        }
    };
    if constexpr (NF == 5) {
#pragma unroll
        for (int hb = 0; hb < MI / 4; ++hb) block(hb);
    }
}

template <bool CS, bool UPS, int MI, int MODE, bool LN = false>
int launch_big(const pv_gemm_params_dev& p, hipStream_t stream) {
    using Cfg = BigCfg<MI, MODE == 2 ? 4 : 5>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, SMEM_BYTES = MODE >= 5 ? Cfg::SMEM_NORM_BYTES : MODE >= 3 ? Cfg::SMEM_PATCH_BYTES : Cfg::SMEM_BYTES;
    if (pv_gemm_probe) {                 // pv_gemm_conv_kernel_info: describe, do not launch
        snprintf(pv_gemm_probe->name, sizeof(pv_gemm_probe->name), "big_tile_kernel<%s, %s, %d, %d, %s>", CS ? "true" : "false", UPS ? "true" : "false", MI, MODE,
                 LN ? "true" : "false");
        pv_gemm_probe->wgs = (long)((p.M + BM - 1) / BM) * (p.N / BN) * ((p.splitk > 1 && p.splitk_ws) ? p.splitk : 1);
        return 0;
    }
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool& attr_set = attr_set_dev[dev_id & 63];
    auto kern = big_tile_kernel<CS, UPS, MI, MODE, LN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
    const int splits = (p.splitk > 1 && p.splitk_ws) ? p.splitk : 1;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, splits), dim3(NW * 64), SMEM_BYTES, stream, p, tiles_n, tiles_m * tiles_n);
    if (splits > 1) return pv_gemm_splitk_reduce_launch(p, splits, stream);
    return PV_CHECK_LAUNCH();
}

}  // namespace

int pv_conv_big_launch(const pv_gemm_params_dev& p, hipStream_t stream) {
    // PV_CONV_BIG: 0 = never (the 128-row kernel everywhere), otherwise the minimum number of 256-row tiles a launch must have
    // (read per call, not cached: the parity test runs both kernels in one process; launches are recorded once and replayed from graphs)
    const char* env = getenv("PV_CONV_BIG");
    const int min_tiles = p.big_tile_min ? p.big_tile_min : (env ? atoi(env) : 256);     // the caller's threshold wins (pv_gemm_params.big_tile_min)
    if (min_tiles <= 0) return p.a_norm ? -2 : -1;              // (-2: a launch with the GroupNorm fold that this path does not take - no fallback exists)
    const int splits = (p.splitk > 1 && p.splitk_ws) ? p.splitk : 1;
    const int cin = p.c0 + p.c1;
    if (p.taps == 1) {
        // Linear / 1x1 layers (round 4): K >= 640 (at K = 320 the ten 32-deep stages do not amortise the ring's prologue and the one-round epilogue;
        // PV_GEMM_BIG=0 keeps them all on pv_gemm.hip), single source, no split-K, fp16 output
        const char* genv = getenv("PV_GEMM_BIG");
        if (genv && atoi(genv) == 0) return -1;
        const int bn = p.geglu ? 256 : 320;
        if (p.c1 || splits > 1 || p.out_f32 || cin < 640 || (p.N % bn) || (p.geglu && p.colstats)) return -1;
        if ((long)((p.M + 255) / 256) * (p.N / bn) < min_tiles) {
            // OFF by default (PV_GEMM_BIG128=1 / =<minimum K> enables): the 128-row form of the tile (MI = 4: 128 x 320, still one 8-wave workgroup per CU) for
            // Linear launches that reach one workgroup per CU only with it - the N = 1280 Linear layers of the 16 x 16 level in the merged plan (M = 8192:
            // 64 x 4 = 256 tiles; the 128 x 160 kernel runs them as 512 tiles).  Round 6, same box: bit-identical and FASTER alone, sustained (K = 1280
            // 38.9 -> 38.0 us, K = 5120 115.5 -> 108 us) - and 0.4 % SLOWER in the loop (33.67 vs 33.81 steps/s, three rounds): behind other launches the
            // one-workgroup-per-CU ring starts cold where the 128 x 160 kernel's second workgroup covers it (profiles/r06_linear128_ab.txt, r06_loop_ab_*.txt)
            static const int b128 = getenv("PV_GEMM_BIG128") ? atoi(getenv("PV_GEMM_BIG128")) : 0;
            if (b128 && !p.geglu && !p.ln_rowsum && cin >= (b128 > 1 ? b128 : 1280) && (long)((p.M + 127) / 128) * (p.N / bn) >= (min_tiles > 256 ? min_tiles : 256))
                // measured at K >= 1280 (PV_GEMM_BIG128=<K> lowers the bound); only where the 128-row tiles fill the whole chip: as a half-chip launch beside
                // the other CFG branch (min_tiles = 128) the form loses to the 128 x 160 kernel (M = 4096: 31.4 vs 24.9 us)
                return p.colstats ? launch_big<true, false, 4, 1>(p, stream) : launch_big<false, false, 4, 1>(p, stream);
            return -1;
        }
        if (p.ln_rowsum) {
            if (p.colstats) return -1;
            return p.geglu ? launch_big<false, false, 8, 2, true>(p, stream) : launch_big<false, false, 8, 1, true>(p, stream);
        }
        if (p.geglu) return launch_big<false, false, 8, 2>(p, stream);
        return p.colstats ? launch_big<true, false, 8, 1>(p, stream) : launch_big<false, false, 8, 1>(p, stream);
    }
    constexpr int BN = 320;
    const int up = p.upsample ? 2 : 1;
    const bool shape_ok = p.taps == 9 && p.stride == 1 && p.pad == 1 && p.hin * up == p.hout && p.win * up == p.wout && (p.N % BN) == 0 &&
                          !p.geglu;
    if (!shape_ok) return -1;
    if (splits == 1 && p.out_f32) return -1;                    // fp32 outputs exist only behind the reduce launch here
    if (splits > 1 && (9 * cin / 64) / splits < 8) return -1;
    // workgroups: split-K launches (16 x 16 level) count their K slices.  (A 128-row form of this tile - MI = 4, 128 x 2 tiles on the 32 x 32 level -
    // was built and measured in round 4: bit-identical, but 142 vs 127 us on conv 640 -> 640 @32 and -2.3 % in the loop: at 91 flop / B it is bound
    // by the L2 -> LDS stream like the 128 x 160 kernel, without that kernel's two independent workgroups per CU.  EXPERIMENTS.md.)
    const long tiles256 = (long)((p.M + 255) / 256) * (p.N / BN) * splits;
    if (tiles256 < min_tiles) return -1;
    const bool cs = p.colstats && splits == 1;                  // with split-K the reduce launch produces the column statistics
    // the LDS-resident input patch (MODE 3 / 4): whole 64- / 32-pixel image rows per tile, no split-K; PV_CONV_PATCH=0 keeps the gathered form
    // PV_CONV_PATCH: 0 never; 64 (default) the 64-pixel-row form only; 1 the 32-pixel-row form too.  Same box, sustained: the 64 x 64 convs 3.7 - 5.3 %
    // faster than the gathered form; the 32 x 32 convs (128 half-chip workgroups) 4 - 6 % slower alone and level in the loop (profiles/r05_conv_patch_*.txt)
    const char* penv = getenv("PV_CONV_PATCH");
    const int pmode = penv ? atoi(penv) : 64;
    if (pmode != 0 && !p.upsample && splits == 1 && (p.wout == 64 || (p.wout == 32 && pmode != 64)) && ((p.hout * p.wout) % 256) == 0 && (p.M % 256) == 0 &&
        (p.c0 % 32) == 0 && (cin % 32) == 0) {
        if (p.a_norm) {
            if (2 * cin * 4 > BigCfg<8, 5>::NORM_TABLE_BYTES) return -2;
            if (p.wout == 64) return cs ? launch_big<true, false, 8, 5>(p, stream) : launch_big<false, false, 8, 5>(p, stream);
            return cs ? launch_big<true, false, 8, 6>(p, stream) : launch_big<false, false, 8, 6>(p, stream);
        }
        if (p.wout == 64) return cs ? launch_big<true, false, 8, 3>(p, stream) : launch_big<false, false, 8, 3>(p, stream);
        return cs ? launch_big<true, false, 8, 4>(p, stream) : launch_big<false, false, 8, 4>(p, stream);
    }
    if (p.a_norm) return -2;                                    // the fold exists on the patch path only: the caller must not fall back
    if (p.upsample) return cs ? launch_big<true, true, 8, 0>(p, stream) : launch_big<false, true, 8, 0>(p, stream);
    return cs ? launch_big<true, false, 8, 0>(p, stream) : launch_big<false, false, 8, 0>(p, stream);
}
