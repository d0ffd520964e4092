// GroupNorm (NHWC, optional two-source channel concat) and LayerNorm for gfx950.  HBM-bound:
// 16-byte vector loads/stores, fp32 statistics.
#include "pv_common.h"

namespace {

// thread t owns 16-byte channel chunk (t % nchunk) of pixel rows (t / nchunk), (t / nchunk) + R, ...
__device__ __forceinline__ half8_t gn_load(const pv_groupnorm_params& p, size_t row, int chunk) {
    const int c = chunk * 8;
    if (c < p.c0) return *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.x0) + row * p.ld0 + c);
    return *reinterpret_cast<const half8_t*>(reinterpret_cast<const half_t*>(p.x1) + row * p.ld1 + (c - p.c0));
}

__global__ void gn_stats_kernel(const pv_groupnorm_params p, const int nchunk, const int rows_per_pass) {
    extern __shared__ float sh[];  // [rows_per_pass][2][C] per-thread partials, reduced in a fixed order (deterministic)
    const int C = p.c0 + p.c1;
    const int b = blockIdx.y, s = blockIdx.x;
    const int tid = threadIdx.x;
    const int chunk = tid % nchunk, r = tid / nchunk;
    const int pps = p.hw / p.splits;
    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = sq[j] = 0.f;
    const size_t row0 = (size_t)b * p.hw + (size_t)s * pps;
    // 4 independent 16-byte loads in flight per thread (the kernel is a pure HBM stream)
    int px = r;
    for (; px + 3 * rows_per_pass < pps; px += 4 * rows_per_pass) {
        half8_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = gn_load(p, row0 + px + u * rows_per_pass, chunk);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = (float)v[u][j];
                sum[j] += f;
                sq[j] += f * f;
            }
    }
    for (; px < pps; px += rows_per_pass) {
        const half8_t v = gn_load(p, row0 + px, chunk);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j];
            sum[j] += f;
            sq[j] += f * f;
        }
    }
    float* mine = sh + (size_t)r * 2 * C;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mine[chunk * 8 + j] = sum[j];
        mine[C + chunk * 8 + j] = sq[j];
    }
    __syncthreads();
    if (tid < p.groups) {
        const int cpg = C / p.groups;
        float a = 0.f, q = 0.f;
        for (int rr = 0; rr < rows_per_pass; ++rr) {
            const float* src = sh + (size_t)rr * 2 * C;
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                a += src[c];
                q += src[C + c];
            }
        }
        float* out = p.partial + (((size_t)b * p.splits + s) * p.groups + tid) * 2;
        out[0] = a;
        out[1] = q;
    }
}

// partial (sum, sumsq) over the S <= 64 splits -> (mean, rstd) per (image, group), written over partial[b][0][g][*].
// One wave per (image, group): lane s loads split s, fixed shuffle tree (deterministic) - a serial loop of S dependent
// loads per thread took 17 us here.
__global__ void gn_finalize_kernel(const pv_groupnorm_params p) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= p.batch * p.groups) return;
    const int b = idx / p.groups, g = idx - b * p.groups;
    float* part = p.partial + ((size_t)b * p.splits * p.groups + g) * 2;
    float a = 0.f, q = 0.f;
    if (lane < p.splits) {
        a = part[(size_t)lane * p.groups * 2];
        q = part[(size_t)lane * p.groups * 2 + 1];
    }
    a = pv_wave_sum(a);
    q = pv_wave_sum(q);
    if (lane == 0) {
        const float n = (float)((p.c0 + p.c1) / p.groups) * (float)p.hw;
        const float mean = a / n;
        const float var = fmaxf(q / n - mean * mean, 0.f);
        part[0] = mean;
        part[1] = rsqrtf(var + p.eps);
    }
}

// (mean, rstd) per (image, group) from the per-64-row-block column sums the producing GEMM epilogues wrote
// (pv_gemm_params.colstats: [block][2][c] fp32).  One wave per (image, group); lanes stride over (block, channel) pairs,
// fixed shuffle tree: deterministic.  Replaces gn_stats_kernel + gn_finalize_kernel: no pass over the activations.
__global__ __launch_bounds__(256) void gn_colstats_finalize_kernel(const pv_groupnorm_params p, float* table) {
    // one 256-thread workgroup per (image, group): 2-3 independent load pairs per thread instead of 10 dependent trips of one wave
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x;
    const int b = idx / p.groups, g = idx - b * p.groups;
    const int C = p.c0 + p.c1;
    const int cpg = C / p.groups;
    const int R = p.hw >> 6;                       // 64-row blocks per image
    float a = 0.f, q = 0.f;
    for (int i = tid; i < R * cpg; i += 256) {
        const int r = i / cpg, c = g * cpg + (i - r * cpg);
        const size_t blk = (size_t)b * R + r;
        if (c < p.c0) {
            const float* src = p.colstats0 + blk * 2 * p.c0 + c;
            a += src[0];
            q += src[p.c0];
        } else {
            const float* src = p.colstats1 + blk * 2 * p.c1 + (c - p.c0);
            a += src[0];
            q += src[p.c1];
        }
    }
    a = pv_wave_sum(a);
    q = pv_wave_sum(q);
    if (lane == 0) { red[wave] = a; red[4 + wave] = q; }
    __syncthreads();
    if (tid == 0) {
        a = (red[0] + red[1]) + (red[2] + red[3]);          // fixed order: deterministic
        q = (red[4] + red[5]) + (red[6] + red[7]);
        float* part = p.partial + ((size_t)b * p.splits * p.groups + g) * 2;
        const float n = (float)cpg * (float)p.hw;
        const float mean = a / n;
        const float var = fmaxf(q / n - mean * mean, 0.f);
        part[0] = mean;
        part[1] = rsqrtf(var + p.eps);
        red[0] = mean;
        red[1] = part[1];
    }
    if (table == nullptr) return;
    // pv_groupnorm_scale_shift: the affine form of this (image, group)'s channels, with gn_apply_kernel's own expressions (a = gamma * rstd,
    // shift = beta - mean * a) so that a consumer applying x * a + shift reproduces pv_groupnorm_apply bit for bit
    __syncthreads();
    if (tid < cpg) {
        const int c = g * cpg + tid;
        const float a = p.gamma[c] * red[1];
        table[((size_t)b * 2) * C + c] = a;
        table[((size_t)b * 2 + 1) * C + c] = p.beta[c] - red[0] * a;
    }
}

__global__ void gn_apply_kernel(const pv_groupnorm_params p, const int nchunk, const int rows_per_pass, const int px_per_block) {
    __shared__ float s_mean[64], s_rstd[64];
    const int C = p.c0 + p.c1;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < p.groups) {
        const float* part = p.partial + ((size_t)b * p.splits * p.groups + tid) * 2;   // finalized by gn_finalize_kernel
        s_mean[tid] = part[0];
        s_rstd[tid] = part[1];
    }
    __syncthreads();
    const int chunk = tid % nchunk, r = tid / nchunk;
    const int cpg = C / p.groups;
    float sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chunk * 8 + j;
        const int g = c / cpg;
        const float a = p.gamma[c] * s_rstd[g];
        sc[j] = a;
        sf[j] = p.beta[c] - s_mean[g] * a;
    }
    const int px0 = blockIdx.x * px_per_block;
    const int px1 = min(px0 + px_per_block, p.hw);
    half_t* y = reinterpret_cast<half_t*>(p.y);
    auto norm_store = [&](const half8_t& v, size_t row) {
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j] * sc[j] + sf[j];
            if (p.act == PV_ACT_SILU) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
            o[j] = (half_t)f;
        }
        *reinterpret_cast<half8_t*>(y + row * C + chunk * 8) = o;
    };
    int px = px0 + r;
    for (; px + 3 * rows_per_pass < px1; px += 4 * rows_per_pass) {   // 4 loads in flight per thread
        half8_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = gn_load(p, (size_t)b * p.hw + px + u * rows_per_pass, chunk);
#pragma unroll
        for (int u = 0; u < 4; ++u) norm_store(v[u], (size_t)b * p.hw + px + u * rows_per_pass);
    }
    for (; px < px1; px += rows_per_pass) norm_store(gn_load(p, (size_t)b * p.hw + px, chunk), (size_t)b * p.hw + px);
}

// one wave per row; lane owns chunks lane, lane+64, ...
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const pv_layernorm_params p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nchunk = p.cols >> 3;
    const half_t* x = reinterpret_cast<const half_t*>(p.x) + (size_t)row * p.ldx;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk) {
            const half8_t h = *reinterpret_cast<const half8_t*>(x + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = (float)h[j];
                sum += v[i][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = pv_wave_sum(sum) / (float)p.cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(pv_wave_sum(sq) / (float)p.cols + p.eps);
    half_t* y = reinterpret_cast<half_t*>(p.y) + (size_t)row * p.ldy;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk) {
            const float4_t g0 = *reinterpret_cast<const float4_t*>(p.gamma + ch * 8);
            const float4_t g1 = *reinterpret_cast<const float4_t*>(p.gamma + ch * 8 + 4);
            const float4_t b0 = *reinterpret_cast<const float4_t*>(p.beta + ch * 8);
            const float4_t b1 = *reinterpret_cast<const float4_t*>(p.beta + ch * 8 + 4);
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = j < 4 ? g0[j] : g1[j - 4];
                const float bb = j < 4 ? b0[j] : b1[j - 4];
                o[j] = (half_t)pv_apply_act((v[i][j] - mean) * rstd * g + bb, p.act);
            }
            *reinterpret_cast<half8_t*>(y + ch * 8) = o;
        }
    }
}

// LayerNorm for the narrow rows of the 64x64-level transformer blocks (cols = 64 * CPL = 320): 8 rows per wave, lane = (row, sub), a
// lane owns the 16-byte chunks {sub, sub + 8, ...} of its row - every lane loads (one wave per row leaves 24 of 64 lanes
// idle at 320 columns), a row's 8 lanes read 128 contiguous bytes, reductions are three lane-xor steps inside the 8 lanes.
template <int CPL>
__global__ __launch_bounds__(256) void layernorm8_kernel(const pv_layernorm_params p) {
    const int lane = threadIdx.x & 63, sub = lane & 7;
    const int row_raw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    const bool ok = row_raw < p.rows;
    const int row = ok ? row_raw : p.rows - 1;
    const half_t* x = reinterpret_cast<const half_t*>(p.x) + (size_t)row * p.ldx;
    float v[CPL][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const half8_t hv = *reinterpret_cast<const half8_t*>(x + (sub + i * 8) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[i][j] = (float)hv[j];
            sum += v[i][j];
        }
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    const float mean = sum / (float)p.cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[i][j] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    sq += __shfl_xor(sq, 4, 64);
    const float rstd = rsqrtf(sq / (float)p.cols + p.eps);
    if (!ok) return;
    half_t* y = reinterpret_cast<half_t*>(p.y) + (size_t)row * p.ldy;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int ch = sub + i * 8;
        const float4_t g0 = *reinterpret_cast<const float4_t*>(p.gamma + ch * 8);
        const float4_t g1 = *reinterpret_cast<const float4_t*>(p.gamma + ch * 8 + 4);
        const float4_t b0 = *reinterpret_cast<const float4_t*>(p.beta + ch * 8);
        const float4_t b1 = *reinterpret_cast<const float4_t*>(p.beta + ch * 8 + 4);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = j < 4 ? g0[j] : g1[j - 4];
            const float bb = j < 4 ? b0[j] : b1[j - 4];
            o[j] = (half_t)pv_apply_act((v[i][j] - mean) * rstd * g + bb, p.act);
        }
        *reinterpret_cast<half8_t*>(y + ch * 8) = o;
    }
}

bool gn_geometry(const pv_groupnorm_params& p, int& nchunk, int& threads, int& rpp) {
    const int C = p.c0 + p.c1;
    if (C <= 0 || (C % 8) || (p.c0 % 8) || p.groups <= 0 || p.groups > 64 || (C % p.groups) || p.splits <= 0 || p.splits > 64 || (p.hw % p.splits))
        return false;
    nchunk = C / 8;
    if (nchunk > 1024) return false;
    rpp = 256 / nchunk;
    if (rpp < 1) rpp = 1;
    threads = nchunk * rpp;
    return true;
}

}  // namespace

extern "C" int pv_groupnorm_stats(const pv_groupnorm_params* p, void* stream) {
    int nchunk, threads, rpp;
    if (!gn_geometry(*p, nchunk, threads, rpp) || !p->partial || !p->x0) return (int)hipErrorInvalidValue;
    const int C = p->c0 + p->c1;
    if (threads < p->groups) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(p->splits, p->batch), dim3(threads), (size_t)rpp * 2 * C * sizeof(float), (hipStream_t)stream, *p,
                       nchunk, rpp);
    const int ng = p->batch * p->groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((ng + 3) / 4), dim3(256), 0, (hipStream_t)stream, *p);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_groupnorm_stats_from_colstats(const pv_groupnorm_params* p, void* stream) {
    int nchunk, threads, rpp;
    if (!gn_geometry(*p, nchunk, threads, rpp) || !p->partial || !p->colstats0 || (p->c1 > 0 && !p->colstats1) || (p->hw % 64) ||
        p->ld0 != p->c0 || (p->c1 > 0 && p->ld1 != p->c1))
        return (int)hipErrorInvalidValue;
    const int ng = p->batch * p->groups;
    hipLaunchKernelGGL(gn_colstats_finalize_kernel, dim3(ng), dim3(256), 0, (hipStream_t)stream, *p, (float*)nullptr);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_groupnorm_scale_shift(const pv_groupnorm_params* p, float* table, void* stream) {
    int nchunk, threads, rpp;
    if (!table || !gn_geometry(*p, nchunk, threads, rpp) || !p->partial || !p->gamma || !p->beta || !p->colstats0 || (p->c1 > 0 && !p->colstats1) || (p->hw % 64) ||
        p->ld0 != p->c0 || (p->c1 > 0 && p->ld1 != p->c1) || (p->c0 + p->c1) / p->groups > 256)
        return (int)hipErrorInvalidValue;
    const int ng = p->batch * p->groups;
    hipLaunchKernelGGL(gn_colstats_finalize_kernel, dim3(ng), dim3(256), 0, (hipStream_t)stream, *p, table);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_groupnorm_apply(const pv_groupnorm_params* p, void* stream) {
    int nchunk, threads, rpp;
    if (!gn_geometry(*p, nchunk, threads, rpp) || !p->partial || !p->x0 || !p->y || !p->gamma || !p->beta)
        return (int)hipErrorInvalidValue;
    if (threads < p->groups) return (int)hipErrorInvalidValue;
    // enough workgroups to fill the chip (>= ~4096 when the tensor allows), at least 8 pixel rows each
    // PV_GN_WGS (A/B): target workgroup count.  Round 6, sustained, alone: 4096 / 2048 / 1024 / 512 workgroups read 15.7 / 15.6 / 15.0 / 16.0 us on the
    // 64 x 64 level's tensor (5.3 - 5.6 TB/s) and 8.6 / 8.7 / 8.6 / 9.7 us on the 16 x 16 level's: nothing to gain (profiles/r06_gn_apply_ab.txt)
    static const long wg_target = getenv("PV_GN_WGS") ? atol(getenv("PV_GN_WGS")) : 4096;
    long ppb = ((long)p->batch * p->hw) / wg_target;
    const int px_per_block = (int)(ppb < 8 ? 8 : (ppb > 128 ? 128 : ppb));
    const int gx = (p->hw + px_per_block - 1) / px_per_block;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, p->batch), dim3(threads), 0, (hipStream_t)stream, *p, nchunk, rpp, px_per_block);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_layernorm(const pv_layernorm_params* p, void* stream) {
    if (p->rows <= 0 || p->cols <= 0 || (p->cols % 8) || p->cols > 4096 || !p->x || !p->y || !p->gamma || !p->beta)
        return (int)hipErrorInvalidValue;
    const int nch = (p->cols / 8 + 63) / 64;
    const dim3 grid((p->rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (p->cols == 320) {          // 8 rows per wave (same-box A/B: 23.6 -> 19.4 us at 65536 rows; no gain at 640 columns)
        hipLaunchKernelGGL(layernorm8_kernel<5>, dim3((p->rows + 31) / 32), block, 0, s, *p);
        return PV_CHECK_LAUNCH();
    }
    switch (nch) {
        case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, *p); break;
        case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, *p); break;
        case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, *p); break;
        case 4: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, *p); break;
        default: hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, *p); break;
    }
    return PV_CHECK_LAUNCH();
}
