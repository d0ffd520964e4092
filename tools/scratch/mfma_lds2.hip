// Phase-structured probe: each phase issues 9 ds_read_b128 into one fragment set and runs 20 MFMAs on the OTHER set
// (loaded one phase earlier), as the GEMM main loop does.  Variants: reads in a burst before the MFMAs, or one read after
// every second MFMA.  1 block of 4 or 8 waves per CU, or 2 blocks of 4 waves.  Wall-clock PFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define M(c, a, b) "v_mfma_f32_16x16x32_f16 %" #c ", %" #a ", %" #b ", %" #c "\n"
#define R(o, off) "ds_read_b128 %" #o ", %26 offset:" #off "\n"
// operands: 0-7 acc | 8-16 set X (8-11 = W frags, 12-16 = A frags... any) | 17-25 set Y | 26 addr
#define MMA20(s0,s1,s2,s3,s4,s5,s6,s7,s8) M(0,s4,s0) M(1,s4,s1) M(2,s4,s2) M(3,s4,s3) M(4,s5,s0) M(5,s5,s1) M(6,s5,s2) M(7,s5,s3) \
  M(0,s6,s0) M(1,s6,s1) M(2,s6,s2) M(3,s6,s3) M(4,s7,s0) M(5,s7,s1) M(6,s7,s2) M(7,s7,s3) M(0,s8,s0) M(1,s8,s1) M(2,s8,s2) M(3,s8,s3)
#define RD9(s0,s1,s2,s3,s4,s5,s6,s7,s8) R(s0,0) R(s1,2048) R(s2,4096) R(s3,6144) R(s4,16384) R(s5,18432) R(s6,20480) R(s7,22528) R(s8,24576)
#define SPREAD(s0,s1,s2,s3,s4,s5,s6,s7,s8, r0,r1,r2,r3,r4,r5,r6,r7,r8) \
  M(0,s4,s0) M(1,s4,s1) R(r0,0) M(2,s4,s2) M(3,s4,s3) R(r1,2048) M(4,s5,s0) M(5,s5,s1) R(r2,4096) M(6,s5,s2) M(7,s5,s3) R(r3,6144) \
  M(0,s6,s0) M(1,s6,s1) R(r4,16384) M(2,s6,s2) M(3,s6,s3) R(r5,18432) M(4,s7,s0) M(5,s7,s1) R(r6,20480) M(6,s7,s2) M(7,s7,s3) R(r7,22528) \
  M(0,s8,s0) M(1,s8,s1) R(r8,24576) M(2,s8,s2) M(3,s8,s3)

template <int MODE>   // 0 burst, 1 spread, 2 MFMA only, 3 reads only, 4 burst + drain + s_barrier, 5 burst + drain
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 36864 / 2; i += blockDim.x) { unsigned h = (unsigned)i * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const float u = (float)(h & 0xffff) / 65536.0f - 0.5f, v = (float)(h >> 16) / 65536.0f - 0.5f;
    reinterpret_cast<_Float16*>(lds)[i] = seed == 0.f ? (_Float16)0.f : seed == 1.f ? (_Float16)(0.01f * ((i * 37) % 97 - 48)) : (_Float16)(seed * (u + v) * 2.0f); }
  __syncthreads();
  f4 c[8]; for (int i = 0; i < 8; ++i) c[i] = f4{0, 0, 0, 0};
  f4 x[9], y[9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned addr = (unsigned)(size_t)lds + (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4) + (wave & 3) * 512;
  for (int i = 0; i < 9; ++i) { x[i] = *reinterpret_cast<f4*>(lds + (lane * 16 + i * 1024) % 36000); y[i] = x[i]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
#define OPS : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), \
              "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), \
              "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]), "+v"(y[8]) : "v"(addr)
    if (MODE == 0) {
      asm volatile(RD9(8,9,10,11,12,13,14,15,16) "s_waitcnt lgkmcnt(9)\n" MMA20(17,18,19,20,21,22,23,24,25)
                   RD9(17,18,19,20,21,22,23,24,25) "s_waitcnt lgkmcnt(9)\n" MMA20(8,9,10,11,12,13,14,15,16) OPS);
    } else if (MODE == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)\n" SPREAD(17,18,19,20,21,22,23,24,25, 8,9,10,11,12,13,14,15,16)
                   "s_waitcnt lgkmcnt(0)\n" SPREAD(8,9,10,11,12,13,14,15,16, 17,18,19,20,21,22,23,24,25) OPS);
    } else if (MODE == 4) {   // as the real kernel: full drain + workgroup barrier between the two phases of a K-step
      asm volatile(RD9(8,9,10,11,12,13,14,15,16) "s_waitcnt lgkmcnt(9)\n" MMA20(17,18,19,20,21,22,23,24,25)
                   "s_waitcnt lgkmcnt(0)\n s_barrier\n"
                   RD9(17,18,19,20,21,22,23,24,25) "s_waitcnt lgkmcnt(9)\n" MMA20(8,9,10,11,12,13,14,15,16) OPS);
    } else if (MODE == 5) {   // full drain, no barrier
      asm volatile(RD9(8,9,10,11,12,13,14,15,16) "s_waitcnt lgkmcnt(9)\n" MMA20(17,18,19,20,21,22,23,24,25)
                   "s_waitcnt lgkmcnt(0)\n"
                   RD9(17,18,19,20,21,22,23,24,25) "s_waitcnt lgkmcnt(9)\n" MMA20(8,9,10,11,12,13,14,15,16) OPS);
    } else if (MODE == 2) {
      asm volatile(MMA20(17,18,19,20,21,22,23,24,25) MMA20(8,9,10,11,12,13,14,15,16) OPS);
    } else {
      asm volatile(RD9(8,9,10,11,12,13,14,15,16) "s_waitcnt lgkmcnt(9)\n" RD9(17,18,19,20,21,22,23,24,25) "s_waitcnt lgkmcnt(9)\n" OPS);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[2] = (float)(t1 - t0); out[3] = (float)(r1 - r0); }
  float s = 0; for (int i = 0; i < 8; ++i) s += c[i][i & 3]; for (int i = 0; i < 9; ++i) s += x[i][0] + y[i][1];
  if (s == 12345.678f) out[0] = s;
}
template <int MODE> void run(const char* name, float* d, int blocks_per_cu, int waves, float seed = 1.0f) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  k<MODE><<<256 * blocks_per_cu, 64 * waves, 73728>>>(d, iters, seed); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256 * blocks_per_cu, 64 * waves, 73728>>>(d, iters, seed);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  float h[4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double mf = (MODE == 3) ? 0.0 : 40.0 * iters * waves * blocks_per_cu * 256.0;
  const double rd = (MODE == 2) ? 0.0 : 18.0 * iters * waves * blocks_per_cu * 256.0;
  printf("%-12s %d block(s)/CU x %d waves: %7.2f ms  %5.2f PFLOP/s  LDS reads %6.1f TB/s   ns/step %.0f  clock %.2f GHz  -> %.0f cycles/step\n", name, blocks_per_cu, waves, ms,
         mf * 16384.0 / (ms * 1e-3) / 1e15, rd * 1024.0 / (ms * 1e-3) / 1e12, ms * 1e6 / iters, h[2] / h[3] * 0.1, ms * 1e6 / iters * h[2] / h[3] * 0.1);
}
int main() { float* d; hipMalloc(&d, 64);
  const int cfg[3][2] = {{1, 4}, {1, 8}, {2, 4}};
  for (auto& c : cfg) {
    run<2>("MFMA only", d, c[0], c[1]); run<3>("reads only", d, c[0], c[1]); run<0>("burst", d, c[0], c[1]); run<1>("spread", d, c[0], c[1]);
  }
  printf("data dependence (2 blocks/CU x 4 waves, burst): LDS holds zeros / small grid / random ~N(0,0.6) / random ~N(0,2.4)\n");
  run<0>("zeros", d, 2, 4, 0.f); run<0>("grid", d, 2, 4, 1.f); run<0>("rand 1", d, 2, 4, 1.5f); run<0>("rand 4", d, 2, 4, 6.f);
  printf("same, MFMA only (operands loaded once from that LDS image)\n");
  run<2>("zeros", d, 2, 4, 0.f); run<2>("grid", d, 2, 4, 1.f); run<2>("rand 1", d, 2, 4, 1.5f);
  printf("drain / barrier between the phases (2 blocks/CU x 4 waves, random data)\n");
  run<0>("burst", d, 2, 4, 1.5f); run<5>("burst+drain", d, 2, 4, 1.5f); run<4>("burst+drain+barrier", d, 2, 4, 1.5f);
  run<0>("burst, zeros", d, 2, 4, 0.f); run<5>("burst+drain, zeros", d, 2, 4, 0.f); run<4>("burst+drain+barrier, zeros", d, 2, 4, 0.f);
  printf("same, spread\n");
  run<1>("zeros", d, 2, 4, 0.f); run<1>("rand 1", d, 2, 4, 1.5f);
  return 0; }
