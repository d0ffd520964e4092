// MFMA issue-rate + in-kernel clock probe (one wave per SIMD, every CU busy)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, float seed) {
  half8 a, b; half4 a4, b4;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(sinf(seed + threadIdx.x * 0.37f + i * 1.7f)); b[i] = (_Float16)(cosf(seed * 0.5f + threadIdx.x * 0.11f - i)); }
  for (int i = 0; i < 4; i++) { a4[i] = a[i]; b4[i] = b[i]; }
  f4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0; f16v d0 = {}, d1 = {};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f4 c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n"
                   "v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x32_f16 %7, %8, %9, %7\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
    }
    if (MODE == 1) {
      asm volatile("v_mfma_f32_16x16x16_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x16_f16 %1, %8, %9, %1\n v_mfma_f32_16x16x16_f16 %2, %8, %9, %2\n v_mfma_f32_16x16x16_f16 %3, %8, %9, %3\n"
                   "v_mfma_f32_16x16x16_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x16_f16 %5, %8, %9, %5\n v_mfma_f32_16x16x16_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x16_f16 %7, %8, %9, %7\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a4), "v"(b4));
    }
    if (MODE == 2) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n"
                   "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n"
                   : "+v"(d0), "+v"(d1) : "v"(a), "v"(b));
    }
  }
  c0 += c4 + c5 + c6 + c7;
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (s == 12345.678f) out[2] = 1;
}
template <int MODE> void run(const char* name, unsigned long long* d) {
  const int iters = 200000;
  k<MODE><<<256, 256>>>(d, iters, 1.0f); hipDeviceSynchronize();
  k<MODE><<<256, 256>>>(d, iters, 1.0f); hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double cyc_per = (double)h[0] / (iters * 8.0), ghz = (double)h[0] / ((double)h[1] * 10.0) ;
  printf("%-22s cycles/MFMA %.2f  in-kernel clock %.3f GHz (memtime/memrealtime@100MHz)\n", name, cyc_per, ghz);
}
int main() { unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  run<0>("16x16x32_f16", d); run<1>("16x16x16_f16 (legacy)", d); run<2>("32x32x16_f16", d); return 0; }
