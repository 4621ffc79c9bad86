// Dev probe (not product): times the dense kernels of csrc/tall.hip on C5's layer shape (204800 x 128 x 128) with parts
// of the work removed (-DHCG_PROBE_NOLOAD / NOSTORE / NOMFMA), to see what bounds them.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize [-DHCG_PROBE_...] -o tools/probe_tall_x tools/probe_tall.hip -L hcatgnet_amd/csrc -lhcatgnet_hip '-Wl,-rpath,$ORIGIN/../hcatgnet_amd/csrc'
#include "../hcatgnet_amd/csrc/tall.hip"
#include <cstdio>
#include <vector>
#define CKH(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  const int N = 204800, F = 128, D = 128;
  float *x, *W, *out, *slabs;
  CKH(hipMalloc(&x, (size_t)N * F * 4)); CKH(hipMalloc(&W, D * F * 4)); CKH(hipMalloc(&out, (size_t)N * D * 4));
  CKH(hipMalloc(&slabs, (size_t)512 * D * F * 4));
  std::vector<float> h((size_t)N * F);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2000) / 1000.f - 1.f;
  CKH(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(W, h.data(), D * F * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(out, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
  const size_t lds = (size_t)3 * 128 * (128 + WPAD) * 2;
  CKH(hipFuncSetAttribute((const void*)k_tall_mm<128, 4, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const size_t ldsw = (size_t)3 * (128 + 128) * (64 + 8) * 2;
  CKH(hipFuncSetAttribute((const void*)k_tall_dw<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));
  for (int which = 0; which < 2; ++which) {
    auto launch = [&]() {
      if (which == 0)
        hipLaunchKernelGGL((k_tall_mm<128, 4, false, false>), dim3(mm_grid(N)), dim3(TT), lds, 0, x, F, W, D, F, out, D, (const float*)nullptr, 0.01f, N);
      else
        hipLaunchKernelGGL((k_tall_dw<4, 4>), dim3(dw_grid(N)), dim3(DWT), ldsw, 0, out, x, F, slabs, N);
    };
    for (int i = 0; i < 5; ++i) launch();
    CKH(hipDeviceSynchronize());
    CKH(hipEventRecord(e0, 0));
    for (int i = 0; i < 50; ++i) launch();
    CKH(hipEventRecord(e1, 0)); CKH(hipEventSynchronize(e1));
    float ms; CKH(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: %.1f us\n", which == 0 ? "k_tall_mm<128,4> fwd" : "k_tall_dw<4,4>", ms * 1000.f / 50);
  }
  return 0;
}
