#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned v = threadIdx.x;
  auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  out[threadIdx.x] = a[0]; out[64 + threadIdx.x] = a[1];
  auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  out[128 + threadIdx.x] = b[0]; out[192 + threadIdx.x] = b[1];
  unsigned w = 1000 + threadIdx.x;
  auto c = __builtin_amdgcn_permlane16_swap(v, w, false, false);
  out[256 + threadIdx.x] = c[0]; out[320 + threadIdx.x] = c[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4);
  k<<<1, 64>>>(d);
  unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"p16(v,v)[0]", "p16(v,v)[1]", "p32(v,v)[0]", "p32(v,v)[1]", "p16(v,w)[0]", "p16(v,w)[1]"};
  for (int r = 0; r < 6; r++) { printf("%s:", names[r]); for (int i = 0; i < 64; i++) printf(" %u", h[r * 64 + i]); printf("\n"); }
}
