// Backward of PhotoVerse's OWN trainable modules (the part of the training step that is the reference's code, not diffusers'):
//
//   (pv_cross_attention_backward - the dual-branch SDPA backward - lives in pv_train.hip with the other MFMA attention backward kernels)
//   pv_transpose_f16              row-major fp16 transpose with zero padding (operand layout of the weight-gradient GEMMs
//                                 dW = dY^T . X run on pv_gemm_conv)
//   pv_layernorm_backward         LayerNorm (+ LeakyReLU) backward of the adapter MLPs (adapters.py:15-19)
//
//   pv_adamw_step, pv_clip_coef   the optimizer (train.py:372-377, :538-545)
//
// fp32 VALU arithmetic on fp16 operands, fixed-order reductions (deterministic).  The MFMA backward kernels (both attentions) and the
// backward of the stock SD-v1.5 / CLIP blocks are in pv_train.hip.
#include "pv_common.h"

namespace {

// out[c][r] = x[r][c] for r < rows, 0 for rows <= r < rows_pad
__global__ void transpose_f16_kernel(const half_t* x, int ldx, int rows, int cols, half_t* out, int ldo, int rows_pad) {
    __shared__ half_t tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? x[(size_t)r * ldx + c] : (half_t)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows_pad) out[(size_t)c * ldo + r] = tile[tx][i];
    }
}

// LayerNorm (+ LeakyReLU 0.01) backward, one wave per row.  y = act(gamma * xhat + beta):
//   g = dy * act'(z); dxhat = g * gamma; dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
//   dgamma / dbeta: per-row-block partials [blocks][2][cols] (summed by the caller's reduce: pv_reduce_cols)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const pv_layernorm_bwd_params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = p.rows_per_wave > 0 ? p.rows_per_wave : 1;     // rows a wave walks: fewer, larger dgamma / dbeta partial blocks
    extern __shared__ float sacc[];     // [4 waves][2][cols] partial dgamma / dbeta of this block's rows
    for (int i = threadIdx.x; i < 8 * p.cols; i += 256) sacc[i] = 0.f;
    __syncthreads();
    for (int ri = 0; ri < rpw; ++ri) {
        const int row = (blockIdx.x * 4 + wave) * rpw + ri;
        if (row >= p.rows) break;
        const half_t* x = reinterpret_cast<const half_t*>(p.x) + (size_t)row * p.ldx;
        // dy may be shared by groups of rows (the patch-token mean of adapters.py:36: every patch row of a sample receives
        // dy[sample] / count, the leading `dy_skip` rows of a group - the CLS row - receive nothing)
        const int grp = p.dy_group > 1 ? p.dy_group : 1;
        const int drow = row / grp;
        const float dys = (p.dy_group > 1 && (row - drow * grp) < p.dy_skip) ? 0.f : p.dy_scale;
        const half_t* dy = reinterpret_cast<const half_t*>(p.dy) + (size_t)drow * p.lddy;
        float sum = 0.f;
        for (int c = lane; c < p.cols; c += 64) sum += (float)x[c];
        const float mean = pv_wave_sum(sum) / (float)p.cols;
        float sq = 0.f;
        for (int c = lane; c < p.cols; c += 64) { const float d = (float)x[c] - mean; sq += d * d; }
        const float rstd = rsqrtf(pv_wave_sum(sq) / (float)p.cols + p.eps);
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < p.cols; c += 64) {
            const float xh = ((float)x[c] - mean) * rstd;
            const float z = p.gamma[c] * xh + p.beta[c];
            float g = (float)dy[c] * dys;
            if (p.act == PV_ACT_LEAKY_RELU && z < 0.f) g *= 0.01f;
            const float dxh = g * p.gamma[c];
            s1 += dxh;
            s2 += dxh * xh;
            sacc[(wave * 2 + 0) * p.cols + c] += g * xh;      // this row's dgamma / dbeta terms (one row per wave)
            sacc[(wave * 2 + 1) * p.cols + c] += g;
        }
        s1 = pv_wave_sum(s1) / (float)p.cols;
        s2 = pv_wave_sum(s2) / (float)p.cols;
        half_t* dx = reinterpret_cast<half_t*>(p.dx) + (size_t)row * p.lddx;
        const half_t* ad = p.add ? reinterpret_cast<const half_t*>(p.add) + (size_t)row * p.ldadd : nullptr;
        for (int c = lane; c < p.cols; c += 64) {
            const float xh = ((float)x[c] - mean) * rstd;
            const float z = p.gamma[c] * xh + p.beta[c];
            float g = (float)dy[c] * dys;
            if (p.act == PV_ACT_LEAKY_RELU && z < 0.f) g *= 0.01f;
            dx[c] = (half_t)(rstd * (g * p.gamma[c] - s1 - xh * s2) + (ad ? (float)ad[c] : 0.f));
        }
    }
    __syncthreads();
    if (p.dgb_partial) {
        for (int c = threadIdx.x; c < p.cols; c += 256) {
            float dg = 0.f, db = 0.f;
            for (int w = 0; w < 4; ++w) { dg += sacc[(w * 2 + 0) * p.cols + c]; db += sacc[(w * 2 + 1) * p.cols + c]; }
            p.dgb_partial[((size_t)blockIdx.x * 2 + 0) * p.cols + c] = dg;
            p.dgb_partial[((size_t)blockIdx.x * 2 + 1) * p.cols + c] = db;
        }
    }
}

// data-gradient-only LayerNorm backward (the frozen LayerNorms of the UNet / text encoder): LPR lanes per row (64 / LPR rows per wave, so
// that every lane loads: at 320 columns one wave per row left 24 of 64 lanes idle - 58 us at 65536 rows), 16-byte loads, the row and its
// gradient live in registers between the three passes (statistics, the two projections, the result)
template <int LPR>
__device__ __forceinline__ float sub_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int LPR, int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const pv_layernorm_bwd_params p) {
    constexpr int RPW = 64 / LPR;                            // rows per wave
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int row_raw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row_raw < p.rows;                      // dead rows compute on a clamped row (the shuffles need every lane) and store nothing
    const int row = live ? row_raw : p.rows - 1;
    const int nchunk = p.cols >> 3;
    const half_t* x = reinterpret_cast<const half_t*>(p.x) + (size_t)row * p.ldx;
    const int grp = p.dy_group > 1 ? p.dy_group : 1;
    const int drow = row / grp;
    const float dys = (p.dy_group > 1 && (row - drow * grp) < p.dy_skip) ? 0.f : p.dy_scale;
    const half_t* dy = reinterpret_cast<const half_t*>(p.dy) + (size_t)drow * p.lddy;
    float xv[NCH][8], gv[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = sub + i * LPR;
        if (ch < nchunk) {
            const half8_t a = *reinterpret_cast<const half8_t*>(x + ch * 8), g = *reinterpret_cast<const half8_t*>(dy + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { xv[i][j] = (float)a[j]; gv[i][j] = (float)g[j] * dys; sum += xv[i][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[i][j] = gv[i][j] = 0.f;
        }
    }
    const float mean = sub_sum<LPR>(sum) / (float)p.cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (sub + i * LPR < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = xv[i][j] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(sub_sum<LPR>(sq) / (float)p.cols + p.eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = sub + i * LPR;
        if (ch < nchunk) {
            const float4_t g0 = *reinterpret_cast<const float4_t*>(p.gamma + ch * 8), g1 = *reinterpret_cast<const float4_t*>(p.gamma + ch * 8 + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gam = j < 4 ? g0[j] : g1[j - 4];
                const float xh = (xv[i][j] - mean) * rstd;
                float g = gv[i][j];
                if (p.act == PV_ACT_LEAKY_RELU && gam * xh + p.beta[ch * 8 + j] < 0.f) g *= 0.01f;
                const float dxh = g * gam;
                xv[i][j] = xh;
                gv[i][j] = dxh;
                s1 += dxh;
                s2 += dxh * xh;
            }
        }
    }
    s1 = sub_sum<LPR>(s1) / (float)p.cols;
    s2 = sub_sum<LPR>(s2) / (float)p.cols;
    if (!live) return;
    half_t* dx = reinterpret_cast<half_t*>(p.dx) + (size_t)row * p.lddx;
    const half_t* ad = p.add ? reinterpret_cast<const half_t*>(p.add) + (size_t)row * p.ldadd : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = sub + i * LPR;
        if (ch < nchunk) {
            half8_t o, a = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (ad) a = *reinterpret_cast<const half8_t*>(ad + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)(rstd * (gv[i][j] - s1 - xv[i][j] * s2) + (float)a[j]);
            *reinterpret_cast<half8_t*>(dx + ch * 8) = o;
        }
    }
}

template <int LPR, int NCH>
void launch_ln_bwd_vec(const pv_layernorm_bwd_params& p, hipStream_t s) {
    constexpr int rows_per_wg = 4 * (64 / LPR);
    hipLaunchKernelGGL((layernorm_bwd_vec_kernel<LPR, NCH>), dim3((unsigned)((p.rows + rows_per_wg - 1) / rows_per_wg)), dim3(256), 0, s, p);
}

// out[k][c] = scale * sum_i x[i][k][c] in i order: column reductions (dgamma / dbeta partials, bias gradients), deterministic
// 32 columns x 8 block-groups per workgroup: group g sums blocks g, g + 8, ... (four loads in flight), the 8 group sums are added in g order -
// a fixed order, and 8 x more workgroups with 8 x shorter serial chains than one thread per column (514 blocks x 2048 columns: 64 -> 9 us)
__global__ __launch_bounds__(256) void reduce_blocks_kernel(const float* x, int nblk, long inner, float scale, float* out) {
    __shared__ float part[8][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < inner) {
        int b = g;
        for (; b + 24 < nblk; b += 32) {
            a0 += x[(size_t)b * inner + i];
            a1 += x[(size_t)(b + 8) * inner + i];
            a2 += x[(size_t)(b + 16) * inner + i];
            a3 += x[(size_t)(b + 24) * inner + i];
        }
        for (; b < nblk; b += 8) a0 += x[(size_t)b * inner + i];
    }
    part[g][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && i < inner) {
        float a = part[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) a += part[k][c];
        out[i] = a * scale;
    }
}

// column sums of fp16 rows (bias gradient db = sum_m dy[m][:]) in two deterministic stages
__global__ __launch_bounds__(256) void colsum_f16_kernel(const half_t* x, int ldx, int rows, int cols, int rows_per_blk, float* partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += (float)x[(size_t)r * ldx + c];
    partial[(size_t)blockIdx.y * cols + c] = a;
}


struct MtEntry { float* p; const float* g; float* m; float* v; const float* gs; long n; };

__global__ __launch_bounds__(256) void sumsq_multi_kernel(const MtEntry* e, const int* blk_tensor, const int* blk_chunk, int chunk, float* partial) {
    __shared__ float red[4];
    const MtEntry t = e[blk_tensor[blockIdx.x]];
    const long i0 = (long)blk_chunk[blockIdx.x] * chunk, i1 = min(i0 + chunk, t.n);
    float a = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) a += t.g[i] * t.g[i];
    a = pv_wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// group g owns the partial sums [start[g], start[g+1]) (fixed order): out[g] = {base * min(1, max_norm / (norm + 1e-6)), norm}, norm of grad / scale.
// OVERFLOW GUARD (gradients are fp16 with a static loss scale; the fp32 reference cannot overflow there): if ANY group's norm is not finite
// the whole optimizer step is skipped, as torch's GradScaler does - every coefficient becomes -1 (pv_adamw_multi leaves parameters, moments
// and the bias-correction step untouched) and counters[1] counts the skipped step; otherwise counters[0], the number of APPLIED steps, advances.
// out has groups + 1 rows: row [groups] = {base, 0} (or {-1, 0} on overflow) is the scale of tensors that belong to NO clip group, so that they
// are skipped together with the grouped ones.
__global__ __launch_bounds__(256) void clip_coef_groups_kernel(const float* partial, const int* start, int groups, float max_norm, float base, float* out, int* counters) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int g = threadIdx.x; g < groups; g += 256) {
        float a = 0.f;
        for (int i = start[g]; i < start[g + 1]; ++i) a += partial[i];
        const float norm = sqrtf(a) * base;
        out[2 * g] = fminf(1.f, max_norm / (norm + 1e-6f)) * base;
        out[2 * g + 1] = norm;
        if (!isfinite(norm)) atomicOr(&bad, 1);          // LDS flag; the result does not depend on arrival order
    }
    __syncthreads();
    if (bad)
        for (int g = threadIdx.x; g < groups; g += 256) out[2 * g] = -1.f;
    if (threadIdx.x == 0) {
        out[2 * groups] = bad ? -1.f : base;
        out[2 * groups + 1] = 0.f;
        if (counters) counters[bad ? 1 : 0] += 1;
    }
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(const MtEntry* e, const int* blk_tensor, const int* blk_chunk, int chunk, float lr, float b1, float b2,
                                                          float eps, float wd, int step, const int* counters) {
    const MtEntry t = e[blk_tensor[blockIdx.x]];
    const long i0 = (long)blk_chunk[blockIdx.x] * chunk, i1 = min(i0 + chunk, t.n);
    const float gs = t.gs ? t.gs[0] : 1.f;
    if (gs < 0.f) return;                                // overflow in this step's gradients: skipped (clip_coef_groups_kernel)
    const int tstep = counters ? counters[0] : step;     // bias correction counts APPLIED steps
    if (tstep <= 0) return;                              // no applied step yet (1 - b1^0 = 0 would divide by zero)
    const float bc1 = 1.f - powf(b1, (float)tstep), bc2_sqrt = sqrtf(1.f - powf(b2, (float)tstep));
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float gi = t.g[i] * gs;
        float pi = t.p[i] * (1.f - lr * wd);
        const float mi = b1 * t.m[i] + (1.f - b1) * gi;
        const float vi = b2 * t.v[i] + (1.f - b2) * gi * gi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        t.p[i] = pi; t.m[i] = mi; t.v[i] = vi;
    }
}

// ---- multi-tensor weight packing: every trainable weight's fp16 copy (+ transposed twin) re-made from its fp32 master in ONE launch ----
// entry = 9 x int64 {src, ld_src, rows, cols, scale (float bits), dst, ld_dst, dstT (or 0), ld_dstT}; workgroup b = 32 x 32 tile blk_tile[b]
// (row-major over the entry's tiles) of entry blk_entry[b]: dst[r][c] = fp16(scale * src[r][c]), dstT[c][r] = the same value
struct PackEntry { const float* src; long ld_src, rows, cols, scale_bits; half_t* dst; long ld_dst; half_t* dstT; long ld_dstT; };

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackEntry* e, const int* blk_entry, const int* blk_tile) {
    __shared__ float tile[32][33];
    const PackEntry t = e[blk_entry[blockIdx.x]];
    const int tiles_c = (int)((t.cols + 31) / 32);
    const long r0 = (long)(blk_tile[blockIdx.x] / tiles_c) * 32, c0 = (long)(blk_tile[blockIdx.x] % tiles_c) * 32;
    const float sc = __int_as_float((int)t.scale_bits);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long row = r0 + r, col = c0 + tx;
        float v = 0.f;
        if (row < t.rows && col < t.cols) {
            v = sc * t.src[row * t.ld_src + col];
            t.dst[row * t.ld_dst + col] = (half_t)v;
        }
        tile[r][tx] = v;
    }
    if (!t.dstT) return;                                                  // uniform over the workgroup
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long col = c0 + r, row = r0 + tx;
        if (row < t.rows && col < t.cols) t.dstT[col * t.ld_dstT + row] = (half_t)tile[tx][r];
    }
}

}  // namespace

extern "C" int pv_pack_weights(const int64_t* entries, const int32_t* blk_entry, const int32_t* blk_tile, int32_t n_blocks, void* stream) {
    if (!entries || !blk_entry || !blk_tile || n_blocks <= 0) return (int)hipErrorInvalidValue;
    static_assert(sizeof(PackEntry) == 9 * sizeof(int64_t), "PackEntry is the int64 [9] row the header documents");
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackEntry*>(entries), blk_entry,
                       blk_tile);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_sumsq_multi(const int64_t* entries, const int32_t* blk_tensor, const int32_t* blk_chunk, int32_t n_blocks, int32_t chunk, float* partial,
                              void* stream) {
    if (!entries || !blk_tensor || !blk_chunk || !partial || n_blocks <= 0 || chunk <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const MtEntry*>(entries), blk_tensor,
                       blk_chunk, chunk, partial);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_clip_coef_groups(const float* partial, const int32_t* group_start, int32_t groups, float max_norm, float base, float* out, int32_t* counters,
                                   void* stream) {
    if (!partial || !group_start || !out || groups <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(clip_coef_groups_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, group_start, groups, max_norm, base, out, counters);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_adamw_multi(const int64_t* entries, const int32_t* blk_tensor, const int32_t* blk_chunk, int32_t n_blocks, int32_t chunk, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int32_t step, const int32_t* counters, void* stream) {
    if (!entries || !blk_tensor || !blk_chunk || n_blocks <= 0 || chunk <= 0 || (step <= 0 && !counters)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const MtEntry*>(entries), blk_tensor,
                       blk_chunk, chunk, lr, beta1, beta2, eps, weight_decay, step, counters);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_transpose_f16(const void* x, int32_t ldx, int32_t rows, int32_t cols, void* out, int32_t ldo, int32_t rows_pad, void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || rows_pad < rows || ldo < rows_pad || ldx < cols) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(transpose_f16_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows_pad + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(x), ldx, rows, cols, reinterpret_cast<half_t*>(out), ldo, rows_pad);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_layernorm_backward(const pv_layernorm_bwd_params* p, void* stream) {
    if (!p->x || !p->dy || !p->dx || !p->gamma || !p->beta || p->rows <= 0 || p->cols <= 0 || p->cols > 2048) return (int)hipErrorInvalidValue;
    if (p->add && p->ldadd < p->cols) return (int)hipErrorInvalidValue;
    // the vector kernel reads 16 bytes at a time from every operand, the optional accumulated gradient included
    if (!p->dgb_partial && p->cols % 8 == 0 && (p->ldx | p->lddy | p->lddx | (p->add ? p->ldadd : 0)) % 8 == 0) {
        const int nchunk = p->cols / 8;                      // fewest lanes per row whose (lanes x chunks) cover the row
        hipStream_t st = (hipStream_t)stream;
        if (nchunk <= 40) launch_ln_bwd_vec<8, 5>(*p, st);          // 320 columns
        else if (nchunk <= 80) launch_ln_bwd_vec<16, 5>(*p, st);    // 640
        else if (nchunk <= 96) launch_ln_bwd_vec<16, 6>(*p, st);    // 768
        else if (nchunk <= 128) launch_ln_bwd_vec<32, 4>(*p, st);   // 1024
        else if (nchunk <= 160) launch_ln_bwd_vec<32, 5>(*p, st);   // 1280
        else launch_ln_bwd_vec<64, 4>(*p, st);
        return PV_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((p->rows + 4 * (p->rows_per_wave > 0 ? p->rows_per_wave : 1) - 1) / (4 * (p->rows_per_wave > 0 ? p->rows_per_wave : 1)))), dim3(256), 8 * p->cols * sizeof(float), (hipStream_t)stream, *p);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_reduce_blocks(const float* x, int32_t nblk, int64_t inner, float scale, float* out, void* stream) {
    if (!x || !out || nblk <= 0 || inner <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(reduce_blocks_kernel, dim3((unsigned)((inner + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x, nblk, (long)inner, scale, out);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_colsum_f16(const void* x, int32_t ldx, int32_t rows, int32_t cols, float* partial, int32_t nblk, float* out, void* stream) {
    if (!x || !partial || !out || rows <= 0 || cols <= 0 || nblk <= 0) return (int)hipErrorInvalidValue;
    const int rpb = (rows + nblk - 1) / nblk;
    hipLaunchKernelGGL(colsum_f16_kernel, dim3((unsigned)((cols + 255) / 256), (unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(x), ldx, rows, cols, rpb, partial);
    hipLaunchKernelGGL(reduce_blocks_kernel, dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, (hipStream_t)stream, partial, nblk, (long)cols, 1.0f, out);
    return PV_CHECK_LAUNCH();
}
