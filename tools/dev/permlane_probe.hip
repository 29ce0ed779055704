// developer probe: what v_permlane16_swap_b32 / v_permlane32_swap_b32 return for two copies of one register (one wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  const unsigned x = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]"};
  for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
