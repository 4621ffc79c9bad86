// Dev probe: lane mapping of gfx950's v_permlane16_swap / v_permlane32_swap as hipcc's builtins expose them.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probe_permlane tools/probe_permlane.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned v = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane16_swap(v, v + 100, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v + 100, false, false);
  o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[] = {"permlane16_swap(v, v+100)[0]", "permlane16_swap(v, v+100)[1]", "permlane32_swap(v, v+100)[0]", "permlane32_swap(v, v+100)[1]"};
  for (int t = 0; t < 4; ++t) { printf("%s:", nm[t]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * t + l]); printf("\n"); }
  return 0;
}
