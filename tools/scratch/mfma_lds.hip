// What does a ds_read_b128 cost the MFMA pipe?  16 x v_mfma_f32_16x16x32_f16 per loop body + R ds_read_b128
// (burst = all reads first; spread = one read after every (16/R)th MFMA), 1 or 2 waves per SIMD, every CU busy.
// Reports SIMD cycles per MFMA (18 = pipe saturated).   hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define MF(i) "v_mfma_f32_16x16x32_f16 %" #i ", %8, %9, %" #i "\n"
#define RD(o, off) "ds_read_b128 %" #o ", %18 offset:" #off "\n"

template <int R, int SPREAD>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = seed + i;
  __syncthreads();
  half8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(sinf(seed + threadIdx.x * 0.37f + i * 1.7f)); b[i] = (_Float16)(cosf(seed * 0.5f + threadIdx.x * 0.11f - i)); }
  f4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  f4 r0 = c0, r1 = c0, r2 = c0, r3 = c0, r4 = c0, r5 = c0, r6 = c0, r7 = c0;
  const unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (R == 0) {
      asm volatile(MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    } else if (R == 4 && !SPREAD) {
      asm volatile("s_waitcnt lgkmcnt(0)\n" RD(10, 0) RD(11, 4096) RD(12, 8192) RD(13, 12288)
                   MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b),
                     "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(addr));
    } else if (R == 4 && SPREAD) {
      asm volatile("s_waitcnt lgkmcnt(0)\n" MF(0) MF(1) RD(10, 0) MF(2) MF(3) MF(4) MF(5) RD(11, 4096) MF(6) MF(7) MF(0) MF(1) RD(12, 8192) MF(2) MF(3) MF(4) MF(5) RD(13, 12288) MF(6) MF(7)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b),
                     "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(addr));
    } else if (R == 8 && !SPREAD) {
      asm volatile("s_waitcnt lgkmcnt(0)\n" RD(10, 0) RD(11, 4096) RD(12, 8192) RD(13, 12288) RD(14, 0) RD(15, 4096) RD(16, 8192) RD(17, 12288)
                   MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b),
                     "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(addr));
    } else if (R == 8 && SPREAD) {
      asm volatile("s_waitcnt lgkmcnt(0)\n" MF(0) RD(10, 0) MF(1) MF(2) RD(11, 4096) MF(3) MF(4) RD(12, 8192) MF(5) MF(6) RD(13, 12288) MF(7) MF(0) RD(14, 0) MF(1) MF(2) RD(15, 4096) MF(3) MF(4) RD(16, 8192) MF(5) MF(6) RD(17, 12288) MF(7)
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b),
                     "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(addr));
    } else if (R == 16) {   // reads only, no MFMA: LDS issue/return rate reference
      asm volatile("s_waitcnt lgkmcnt(0)\n" RD(10, 0) RD(11, 4096) RD(12, 8192) RD(13, 12288) RD(14, 0) RD(15, 4096) RD(16, 8192) RD(17, 12288)
                   :: "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7), "v"(a), "v"(b),
                     "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(addr));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + r0[0] + r1[0] + r2[0] + r3[0] + r4[0] + r5[0] + r6[0] + r7[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (s == 12345.678f) out[2] = 1;
}
// NOTE: the asm writes r0..r7 although they are declared inputs (the compiler must not know they change, or it would
// insert its own waits); they are never read for anything that matters.
template <int R, int SPREAD> void run(const char* name, unsigned long long* d, int wps) {
  const int iters = 50000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<R, SPREAD><<<256, 256 * wps>>>(d, iters, 1.0f); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<R, SPREAD><<<256, 256 * wps>>>(d, iters, 1.0f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double per_body = (double)h[0] / iters;
  const double mf = (R == 16) ? 0.0 : 16.0 * iters * 4.0 * wps * 256.0;   // MFMAs chip-wide
  printf("%-34s waves/SIMD %d  memtime ticks/body %.1f  wall %.2f ms -> %.0f ns/body, %.2f PFLOP/s, shader cycles/MFMA/SIMD @2.4GHz %.1f\n", name, wps, per_body,
         ms, ms * 1e6 / iters, mf * 16384.0 / (ms * 1e-3) / 1e15, mf > 0 ? (ms * 1e-3 * 2.4e9) / (16.0 * iters * wps) : 0.0);
}
int main() { unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  for (int wps = 1; wps <= 2; ++wps) {
    run<0, 0>("16 MFMA", d, wps);
    run<4, 0>("16 MFMA + 4 ds_read_b128 burst", d, wps);
    run<4, 1>("16 MFMA + 4 ds_read_b128 spread", d, wps);
    run<8, 0>("16 MFMA + 8 ds_read_b128 burst", d, wps);
    run<8, 1>("16 MFMA + 8 ds_read_b128 spread", d, wps);
    run<16, 0>("8 ds_read_b128 only", d, wps);
  }
  return 0; }
