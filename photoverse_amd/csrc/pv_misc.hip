// Small HBM-bound kernels around the UNet: timestep embedding, conv_in / conv_out (layout change
// fused), GEGLU gate, CFG + DPM-Solver++ update, casts, patch-token mean.
#include "pv_common.h"

extern "C" int pv_abi_version(void) { return PV_ABI_VERSION; }

extern "C" int pv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

namespace {

// step index of the loop state block, clamped to the table length state[1] (0 = length unknown): a step() past the end of the
// schedule re-reads the last row instead of reading beyond the tables (the host raises before that, pipeline.DenoiseLoop.step)
__device__ __forceinline__ int pv_step_index(const int32_t* state) {
    const int i = state[0], n = state[1];
    return n > 0 ? min(i, n - 1) : i;
}

// [EXT] diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void timestep_embedding_kernel(const float* timesteps, const int32_t* state, int rows, int dim, half_t* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * dim) return;
    const int row = idx / dim, i = idx - row * dim;
    const int half = dim >> 1;
    const float t = state ? timesteps[pv_step_index(state)] : timesteps[row];
    const int k = i < half ? i : i - half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
    const float a = t * freq;
    out[idx] = (half_t)(i < half ? cosf(a) : sinf(a));
}

// conv_out: NHWC fp16 -> NCHW fp32, 3x3 pad 1, cout <= 8 (UNet 320 -> 4, VAE 128 -> 3).  Far too few output channels for the
// MFMA tile: VALU kernel.  8 threads per output pixel, each owning the 16-byte channel chunks {sub, sub+8, ...} of all 9
// taps (one pixel's 8 threads read 128 contiguous bytes; neighbouring pixels re-read the same rows out of L1); the whole
// filter (cout*9*cin halfs, <= 23 KB) sits in LDS and is read as broadcasts; products by v_dot2_f32_f16; the 8 partial sums
// of a pixel meet through three lane-xor steps.  (The first version - one wave per pixel, weights from global - spent
// 109 us on the UNet's 65536 x 320 -> 4 launch.)
// CPT: 16-byte channel chunks per thread (cin = 64 * CPT) - the chunk loads of a tap are issued together and the next tap's
// loads are in flight while the current tap is multiplied (45 dependent load -> use round trips per thread otherwise).
template <int COUT, int CPT>
__global__ __launch_bounds__(256, 4) void conv_out_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int batch, int cin,
                                                       int h, int wd) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    half_t* sw = reinterpret_cast<half_t*>(smem_raw);
    const int tid = threadIdx.x;
    const int wtot = COUT * 9 * cin;
    for (int i = tid * 8; i < wtot; i += 256 * 8) *reinterpret_cast<half8_t*>(sw + i) = *reinterpret_cast<const half8_t*>(w + i);
    __syncthreads();
    const int sub = tid & 7;
    const long total = (long)batch * h * wd;
    // grid-stride over 32-pixel groups: the LDS filter fill (as many bytes as 36 pixels of input) is paid once per workgroup
    for (long grp = blockIdx.x; grp * 32 < total; grp += gridDim.x) {
    const long pix_raw = grp * 32 + (tid >> 3);
    const bool ok = pix_raw < total;
    const long pix = ok ? pix_raw : total - 1;
    const int b = (int)(pix / (h * wd));
    const int rem = (int)(pix - (long)b * h * wd);
    const int y = rem / wd, xx = rem - y * wd;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
    auto load_tap = [&](int tap, half8_t (&v)[CPT]) {
        const int iy = y + tap / 3 - 1, ix = xx + tap % 3 - 1;
        const bool in = iy >= 0 && iy < h && ix >= 0 && ix < wd;
        const half_t* xr = x + ((long)(b * h + (in ? iy : y)) * wd + (in ? ix : xx)) * cin;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            v[i] = *reinterpret_cast<const half8_t*>(xr + (sub + i * 8) * 8);
            if (!in) v[i] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    auto mac_tap = [&](int tap, const half8_t (&v)[CPT]) {
        const half_t* wr = sw + tap * cin;
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                const half8_t ww = *reinterpret_cast<const half8_t*>(wr + c * 9 * cin + (sub + i * 8) * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[c] = __builtin_amdgcn_fdot2(half2_t{v[i][2 * j], v[i][2 * j + 1]}, half2_t{ww[2 * j], ww[2 * j + 1]}, acc[c], false);
            }
    };
    half8_t va[CPT], vb[CPT];
    load_tap(0, va);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {      // rolled: fully unrolled, hipcc hoists every load / LDS read and spills (916 B scratch)
        load_tap(min(tap + 1, 8), vb);
        mac_tap(tap, va);
#pragma unroll
        for (int i = 0; i < CPT; ++i) va[i] = vb[i];
    }
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        acc[c] += __shfl_xor(acc[c], 1, 64);
        acc[c] += __shfl_xor(acc[c], 2, 64);
        acc[c] += __shfl_xor(acc[c], 4, 64);
    }
    if (ok && sub < COUT) {
        float v = acc[0];
#pragma unroll
        for (int c = 1; c < COUT; ++c) v = (sub == c) ? acc[c] : v;
        out[(((long)b * COUT + sub) * h + y) * wd + xx] = v + (bias ? bias[sub] : 0.f);
    }
    }
}

template <int COUT>
int launch_conv_out(const half_t* x, const half_t* w, const float* bias, float* out, int batch, int cin, int h, int wd, hipStream_t stream) {
    const long pix = (long)batch * h * wd;
    const size_t smem = (size_t)COUT * 9 * cin * sizeof(half_t);
    const long groups = (pix + 31) / 32;
    const long per_cu = smem ? (long)(160 * 1024 / smem) : 8;                     // workgroups per CU that fit (LDS, <= 8 by waves)
    const long cap = 256 * (per_cu < 8 ? per_cu : 8);
    const dim3 grid((unsigned)(groups < cap ? groups : cap)), block(256);
    switch (cin / 64) {
        case 1: hipLaunchKernelGGL((conv_out_kernel<COUT, 1>), grid, block, smem, stream, x, w, bias, out, batch, cin, h, wd); break;
        case 2: hipLaunchKernelGGL((conv_out_kernel<COUT, 2>), grid, block, smem, stream, x, w, bias, out, batch, cin, h, wd); break;
        case 4: hipLaunchKernelGGL((conv_out_kernel<COUT, 4>), grid, block, smem, stream, x, w, bias, out, batch, cin, h, wd); break;
        case 5: hipLaunchKernelGGL((conv_out_kernel<COUT, 5>), grid, block, smem, stream, x, w, bias, out, batch, cin, h, wd); break;
        case 8: hipLaunchKernelGGL((conv_out_kernel<COUT, 8>), grid, block, smem, stream, x, w, bias, out, batch, cin, h, wd); break;
        default: return (int)hipErrorInvalidValue;
    }
    return PV_CHECK_LAUNCH();
}

__global__ void geglu_kernel(const half_t* x, int ldx, half_t* out, int ldo, int rows, int n) {
    const int nchunk = n >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * nchunk) return;
    const int ch = (int)(idx % nchunk);
    const long r = idx / nchunk;
    const half8_t a = *reinterpret_cast<const half8_t*>(x + r * ldx + ch * 8);
    const half8_t g = *reinterpret_cast<const half8_t*>(x + r * ldx + n + ch * 8);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] * pv_gelu_erf((float)g[j]));
    *reinterpret_cast<half8_t*>(out + r * ldo + ch * 8) = o;
}

// coef row (8 floats): {ca, cb, cx, c0, c1, -, -, -}
//   eps = eps_u + g (eps_c - eps_u)              infer.py:116
//   x0  = ca*x + cb*eps                          epsilon -> data prediction
//   x'  = cx*x + c0*x0 + c1*x0_prev              DPM-Solver++ 1st / 2nd order (midpoint)
__global__ void cfg_dpm_step_kernel(const float* eu, const float* ec, float* lat, float* x0p, const float* coef,
                                    const int32_t* state, float g, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float* c = coef + (long)pv_step_index(state) * 8;
    const float ca = c[0], cb = c[1], cx = c[2], c0 = c[3], c1 = c[4];
    const float4_t u = *reinterpret_cast<const float4_t*>(eu + i);
    const float4_t cc = *reinterpret_cast<const float4_t*>(ec + i);
    float4_t x = *reinterpret_cast<const float4_t*>(lat + i);
    float4_t xp = *reinterpret_cast<const float4_t*>(x0p + i);
    float4_t x0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float e = u[j] + g * (cc[j] - u[j]);
        x0[j] = ca * x[j] + cb * e;
        x[j] = cx * x[j] + c0 * x0[j] + c1 * xp[j];
    }
    *reinterpret_cast<float4_t*>(lat + i) = x;
    *reinterpret_cast<float4_t*>(x0p + i) = x0;
}

__global__ void step_advance_kernel(int32_t* state) { state[0] += 1; }

// Philox4x32-10 (Salmon et al. 2011): counter-based, so one launch = one independent draw per layer with no stored stream
__device__ __forceinline__ uint32_t philox_first_word(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1) {
    uint32_t c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

__global__ void fusion_draw_kernel(const int32_t* state, uint32_t* rng, const float* forced, float* out, int n, float r1, float r2,
                                   float scale, int only_last) {
    const int i = threadIdx.x;
    const uint32_t call = rng[2];
    float wt = 1.f, wi = 1.f;
    const bool on = !(only_last && state) || state[0] == state[1] - 1;
    if (i < n && on) {
        float u = (float)(philox_first_word(rng[0], rng[1], call, (uint32_t)i) >> 8) * (1.0f / 16777216.0f);   // [0, 1)
        if (forced && forced[i] >= 0.f) u = forced[i];
        if (u < r1) { wt = scale; wi = 0.f; }
        else if (u > r2) { wt = 0.f; wi = scale; }
    }
    if (i < n) { out[2 * i] = wt; out[2 * i + 1] = wi; }
    __syncthreads();
    if (i == 0) rng[2] = call + 1u;
}

__global__ void cast_f32_f16_kernel(const float* x, half_t* y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (half_t)x[i];
}
__global__ void cast_f16_f32_kernel(const half_t* x, float* y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (float)x[i];
}

// one workgroup = one group x 32 eight-column chunks x 8 row lanes: lane l sums rows l, l + 8, ...; the eight partial sums are added in lane
// order (fixed).  16 groups x 257 rows x 1024 columns: 78 -> ~10 us against one thread walking all rows of its chunk.
__global__ __launch_bounds__(256) void rows_mean_kernel(const half_t* x, int ldx, half_t* y, int ldy, int groups, int count, int group_rows, int cols,
                                                        int accumulate) {
    __shared__ float part[8][32][9];
    const int nchunk = cols >> 3, cpb = (nchunk + 31) / 32;
    const int g = blockIdx.x / cpb, ch = (blockIdx.x % cpb) * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (ch < nchunk)
        for (int r = rl; r < count; r += 8) {
            const half8_t v = *reinterpret_cast<const half8_t*>(x + ((long)g * group_rows + r) * ldx + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[rl][threadIdx.x & 31][j] = acc[j];
    __syncthreads();
    if (rl != 0 || ch >= nchunk) return;
#pragma unroll
    for (int k = 1; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += part[k][threadIdx.x & 31][j];
    half_t* yo = y + (long)g * ldy + ch * 8;
    half8_t o;
    half8_t prev = accumulate ? *reinterpret_cast<const half8_t*>(yo) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(acc[j] / (float)count + (float)prev[j]);
    *reinterpret_cast<half8_t*>(yo) = o;
}

// im2col of a 3x3 / pad 1 convolution over an NCHW fp32 tensor with few channels (the UNet's conv_in, cin = 4):
// rows [B*H*W][kpad] fp16, column k = ci*9 + ky*3 + kx (the flattened Conv2d weight order), zero padded to kpad.
// The conv itself then runs on the MFMA GEMM.
__global__ void im2col3x3_kernel(const float* __restrict__ x, half_t* __restrict__ out, int batch, int cin, int h, int wd, int kpad) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread = 8 consecutive columns of one row
    const int nch = kpad >> 3;
    if (idx >= (long)batch * h * wd * nch) return;
    const int ch = (int)(idx % nch);
    const long pix = idx / nch;
    const int b = (int)(pix / (h * wd));
    const int rem = (int)(pix - (long)b * h * wd);
    const int y = rem / wd, xx = rem - y * wd;
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ch * 8 + j;
        float v = 0.f;
        if (k < cin * 9) {
            const int ci = k / 9, t = k - ci * 9;
            const int iy = y + t / 3 - 1, ix = xx + t % 3 - 1;
            if (iy >= 0 && iy < h && ix >= 0 && ix < wd) v = x[(((long)b * cin + ci) * h + iy) * wd + ix];
        }
        o[j] = (half_t)v;
    }
    *reinterpret_cast<half8_t*>(out + pix * kpad + ch * 8) = o;
}

// in-place row softmax of fp16 scores: x[r][c] = softmax_c(scale * x[r][c]); one 256-thread workgroup per row, fp32 math
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* __restrict__ x, int ld, int cols, float scale) {
    __shared__ float red[4];
    half_t* row = x + (size_t)blockIdx.x * ld;
    const int nch = cols >> 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float sl2 = scale * 1.4426950408889634f;
    float mx = -INFINITY;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const half8_t v = *reinterpret_cast<const half8_t*>(row + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)v[j]);
    }
    mx = pv_wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * sl2;
    __syncthreads();
    float sum = 0.f;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const half8_t v = *reinterpret_cast<const half8_t*>(row + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += __builtin_amdgcn_exp2f(fmaf((float)v[j], sl2, -mx));
    }
    sum = pv_wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
        const half8_t v = *reinterpret_cast<const half8_t*>(row + ch * 8);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)(__builtin_amdgcn_exp2f(fmaf((float)v[j], sl2, -mx)) * inv);
        *reinterpret_cast<half8_t*>(row + ch * 8) = o;
    }
}

// 1x1 convolution over an NCHW fp32 tensor with few channels (VAE post_quant_conv, 4 -> 4)
__global__ void pointwise_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                      float* __restrict__ out, int batch, int cin, int cout, int hw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * hw) return;
    const int b = (int)(idx / hw), px = (int)(idx - (long)b * hw);
    for (int co = 0; co < cout; ++co) {
        float acc = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < cin; ++ci) acc += w[co * cin + ci] * x[((long)b * cin + ci) * hw + px];
        out[((long)b * cout + co) * hw + px] = acc;
    }
}

__global__ void clamp_f32_kernel(float* x, float lo, float hi, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fminf(fmaxf(x[i], lo), hi);
}

// CLIP ViT patchify: NCHW fp32 pixels -> fp16 rows [B*gh*gw][kpad], row = one patch flattened (c, py, px), zero padded
__global__ void patchify_kernel(const float* __restrict__ x, half_t* __restrict__ out, int batch, int ch, int img, int patch, int kpad) {
    const int g = img / patch;
    const long total = (long)batch * g * g * kpad;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % kpad);
    const long row = idx / kpad;
    float v = 0.f;
    if (k < ch * patch * patch) {
        const int c = k / (patch * patch), r = k - c * patch * patch;
        const int py = r / patch, px = r - py * patch;
        const int b = (int)(row / (g * g)), pr = (int)(row - (long)b * g * g);
        const int gy = pr / g, gx = pr - gy * g;
        v = x[(((long)b * ch + c) * img + gy * patch + py) * img + gx * patch + px];
    }
    out[idx] = (half_t)v;
}

// CLIP vision embeddings: out[b][0] = cls + pos[0]; out[b][1+i] = patch[b][i] + pos[1+i]     (fp16 rows, fp32 params)
__global__ void vision_embed_kernel(const half_t* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
                                    half_t* __restrict__ out, int batch, int ntok, int dim) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * ntok * dim) return;
    const int d = (int)(idx % dim);
    const long row = idx / dim;
    const int t = (int)(row % ntok), b = (int)(row / ntok);
    const float v = t == 0 ? cls[d] : (float)patches[((long)b * (ntok - 1) + (t - 1)) * dim + d];
    out[idx] = (half_t)(v + pos[(long)t * dim + d]);
}

// CLIP text embeddings with PhotoVerse concept injection (models/clip.py:17-24,57-63), one thread per element:
//   j <  idx       : tok[ids[b][j]]
//   idx<=j<idx+E   : concept[b][j-idx]
//   j >= idx+E     : tok[ids[b][j-E+1]]          (tail shifted right by E-1, truncated)
// then + pos[j].  E == 0 (no concept) is the stock embedding.
__global__ void text_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                                  const half_t* __restrict__ concept, const int64_t* __restrict__ pidx, int E, half_t* __restrict__ out,
                                  int batch, int seq, int dim) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * seq * dim) return;
    const int d = (int)(i % dim);
    const long row = i / dim;
    const int j = (int)(row % seq), b = (int)(row / seq);
    float v;
    const int idx = E > 0 ? (int)pidx[b] : seq;
    if (j < idx) v = tok[ids[(long)b * seq + j] * dim + d];
    else if (j < idx + E) v = (float)concept[((long)b * E + (j - idx)) * dim + d];
    else v = tok[ids[(long)b * seq + (j - E + 1)] * dim + d];
    out[i] = (half_t)(v + pos[(long)j * dim + d]);
}

}  // namespace

extern "C" int pv_im2col3x3(const float* x, void* out, int32_t batch, int32_t cin, int32_t h, int32_t wd, int32_t kpad, void* stream) {
    if (batch <= 0 || cin <= 0 || h <= 0 || wd <= 0 || (kpad % 8) || kpad < cin * 9 || !x || !out) return (int)hipErrorInvalidValue;
    const long total = (long)batch * h * wd * (kpad / 8);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<half_t*>(out), batch, cin, h, wd, kpad);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_softmax_rows(void* x, int32_t ld, int32_t rows, int32_t cols, float scale, void* stream) {
    if (rows <= 0 || cols <= 0 || (cols % 8) || (ld % 8) || !x) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<half_t*>(x), ld, cols, scale);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int32_t batch, int32_t cin, int32_t cout,
                                 int32_t hw, void* stream) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || hw <= 0 || !x || !w || !out) return (int)hipErrorInvalidValue;
    const long total = (long)batch * hw;
    hipLaunchKernelGGL(pointwise_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias, out,
                       batch, cin, cout, hw);
    return PV_CHECK_LAUNCH();
}

namespace {
// out[b][i] = ca[b] * x[b][i] + cb[b] * y[b][i]   (per-sample coefficients; y / cb may be NULL: out = ca[b] * x)
__global__ void affine_rows_kernel(const float* x, const float* y, const float* ca, const float* cb, float* out, long per, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long b = i / per;
    float v = ca[b] * x[i];
    if (y) v += cb[b] * y[i];
    out[i] = v;
}
// posterior sample of the VAE encoder: out = mean + exp(0.5 * clamp(logvar, -30, 20)) * eps; moments = [B][2c][hw] (mean | logvar)
__global__ void posterior_sample_kernel(const float* moments, const float* eps, float* out, long chw, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long b = i / chw, r = i - b * chw;
    const float mean = moments[b * 2 * chw + r];
    const float logvar = fminf(fmaxf(moments[b * 2 * chw + chw + r], -30.f), 20.f);
    out[i] = mean + __expf(0.5f * logvar) * eps[i];
}
// stage 1 of a deterministic mean: block partial sums of f(a, b); mode 0: a, 1: |a|, 2: (a - b)^2, 3: a^2.  a / b: fp32, or fp16 when f16 != 0
__global__ __launch_bounds__(256) void reduce_partial_kernel(const void* a_, const void* b_, int mode, int f16, long n, float* partial) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float a = f16 ? (float)reinterpret_cast<const half_t*>(a_)[i] : reinterpret_cast<const float*>(a_)[i];
        if (mode == 0) acc += a;
        else if (mode == 1) acc += fabsf(a);
        else if (mode == 3) acc += a * a;
        else {
            const float b = f16 ? (float)reinterpret_cast<const half_t*>(b_)[i] : reinterpret_cast<const float*>(b_)[i];
            acc += (a - b) * (a - b);
        }
    }
    acc = pv_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void reduce_final_kernel(const float* partial, int nb, float scale, float* out) {
    float acc = 0.f;
    for (int i = 0; i < nb; ++i) acc += partial[i];     // fixed order
    out[0] = acc * scale;
}
}  // namespace

extern "C" int pv_affine_rows_f32(const float* x, const float* y, const float* ca, const float* cb, float* out, int64_t per_sample, int32_t batch,
                                  void* stream) {
    if (!x || !ca || !out || per_sample <= 0 || batch <= 0 || (y && !cb)) return (int)hipErrorInvalidValue;
    const long n = (long)per_sample * batch;
    hipLaunchKernelGGL(affine_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, ca, cb, out, (long)per_sample, n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_posterior_sample(const float* moments, const float* eps, float* out, int32_t batch, int64_t chw, void* stream) {
    if (!moments || !eps || !out || batch <= 0 || chw <= 0) return (int)hipErrorInvalidValue;
    const long n = (long)batch * chw;
    hipLaunchKernelGGL(posterior_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, moments, eps, out, (long)chw, n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_reduce_mean(const void* a, const void* b, int32_t mode, int32_t is_f16, int64_t n, float* partial, int32_t n_partial, float* out,
                              void* stream) {
    if (!a || !partial || !out || n <= 0 || n_partial <= 0 || n_partial > 4096 || mode < 0 || mode > 2 || (mode == 2 && !b)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(reduce_partial_kernel, dim3((unsigned)n_partial), dim3(256), 0, (hipStream_t)stream, a, b, mode, is_f16, (long)n, partial);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, partial, n_partial, 1.0f / (float)n, out);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_reduce_sumsq(const float* a, int64_t n, float scale, float* partial, int32_t n_partial, float* out, void* stream) {
    if (!a || !partial || !out || n <= 0 || n_partial <= 0 || n_partial > 4096) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(reduce_partial_kernel, dim3((unsigned)n_partial), dim3(256), 0, (hipStream_t)stream, a, nullptr, 3, 0, (long)n, partial);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, partial, n_partial, scale, out);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_clamp_f32(float* x, float lo, float hi, int64_t n, void* stream) {
    if (n <= 0 || !x) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(clamp_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, lo, hi, (long)n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_patchify(const float* x, void* out, int32_t batch, int32_t ch, int32_t img, int32_t patch, int32_t kpad, void* stream) {
    if (batch <= 0 || ch <= 0 || img <= 0 || patch <= 0 || (img % patch) || kpad < ch * patch * patch || !x || !out)
        return (int)hipErrorInvalidValue;
    const long total = (long)batch * (img / patch) * (img / patch) * kpad;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<half_t*>(out), batch, ch, img, patch, kpad);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_clip_vision_embed(const void* patches, const float* cls, const float* pos, void* out, int32_t batch, int32_t ntok,
                                    int32_t dim, void* stream) {
    if (batch <= 0 || ntok <= 1 || dim <= 0 || !patches || !cls || !pos || !out) return (int)hipErrorInvalidValue;
    const long total = (long)batch * ntok * dim;
    hipLaunchKernelGGL(vision_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(patches), cls, pos, reinterpret_cast<half_t*>(out), batch, ntok, dim);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_clip_text_embed(const int64_t* ids, const float* tok, const float* pos, const void* concept, const int64_t* placeholder_idx,
                                  int32_t n_concept, void* out, int32_t batch, int32_t seq, int32_t dim, void* stream) {
    if (batch <= 0 || seq <= 0 || dim <= 0 || n_concept < 0 || n_concept > seq || !ids || !tok || !pos || !out ||
        (n_concept > 0 && (!concept || !placeholder_idx)))
        return (int)hipErrorInvalidValue;
    const long total = (long)batch * seq * dim;
    hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, tok, pos,
                       reinterpret_cast<const half_t*>(concept), placeholder_idx, n_concept, reinterpret_cast<half_t*>(out), batch, seq, dim);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_timestep_embedding(const float* timesteps, const int32_t* state, int32_t rows, int32_t dim, void* out,
                                     void* stream) {
    if (rows <= 0 || dim <= 0 || (dim & 1) || !timesteps || !out) return (int)hipErrorInvalidValue;
    const int n = rows * dim;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, timesteps, state, rows,
                       dim, reinterpret_cast<half_t*>(out));
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_conv_out(const void* x, const void* w, const float* bias, float* out, int32_t batch, int32_t cin, int32_t h,
                           int32_t wd, int32_t cout, void* stream) {
    if (batch <= 0 || (cin % 64) || (cout != 4 && cout != 3) || !x || !w || !out) return (int)hipErrorInvalidValue;
    // cin in {64, 128, 256, 320, 512}: the filter (<= 37 KB) lives in LDS
    if (cout == 4)
        return launch_conv_out<4>(reinterpret_cast<const half_t*>(x), reinterpret_cast<const half_t*>(w), bias, out, batch, cin, h, wd, (hipStream_t)stream);
    return launch_conv_out<3>(reinterpret_cast<const half_t*>(x), reinterpret_cast<const half_t*>(w), bias, out, batch, cin, h, wd, (hipStream_t)stream);
}

extern "C" int pv_geglu(const void* x, int32_t ldx, void* out, int32_t ldo, int32_t rows, int32_t n, void* stream) {
    if (rows <= 0 || n <= 0 || (n % 8) || !x || !out) return (int)hipErrorInvalidValue;
    const long total = (long)rows * (n / 8);
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(x), ldx, reinterpret_cast<half_t*>(out), ldo, rows, n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_cfg_dpm_step(const float* eps_uncond, const float* eps_cond, float* latents, float* x0_prev, const float* coef,
                               const int32_t* state, float guidance, int64_t n, void* stream) {
    if (n <= 0 || (n % 4) || !eps_uncond || !eps_cond || !latents || !x0_prev || !coef || !state) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cfg_dpm_step_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, eps_uncond,
                       eps_cond, latents, x0_prev, coef, state, guidance, (long)n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_step_advance(int32_t* state, void* stream) {
    if (!state) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_fusion_draw(const int32_t* state, uint32_t* rng, const float* forced, float* out, int32_t n_layers, float rule1, float rule2,
                              float scale, int32_t only_last_step, void* stream) {
    if (!rng || !out || n_layers <= 0 || n_layers > 256) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(fusion_draw_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, state, rng, forced, out, n_layers, rule1, rule2, scale,
                       only_last_step);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream) {
    if (n <= 0 || !x || !y) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<half_t*>(y), (long)n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_cast_f16_to_f32(const void* x, float* y, int64_t n, void* stream) {
    if (n <= 0 || !x || !y) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(cast_f16_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(x), y, (long)n);
    return PV_CHECK_LAUNCH();
}

extern "C" int pv_rows_mean(const void* x, int32_t ldx, void* y, int32_t ldy, int32_t groups, int32_t count, int32_t group_rows,
                            int32_t cols, int32_t accumulate, void* stream) {
    if (groups <= 0 || count <= 0 || group_rows < count || cols <= 0 || (cols % 8) || !x || !y) return (int)hipErrorInvalidValue;
    const int blocks = groups * ((cols / 8 + 31) / 32);
    hipLaunchKernelGGL(rows_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(x), ldx, reinterpret_cast<half_t*>(y), ldy, groups, count, group_rows, cols, accumulate);
    return PV_CHECK_LAUNCH();
}
