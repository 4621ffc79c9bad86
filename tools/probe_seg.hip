// Dev probe (not product): phase stamps (s_memtime: shader clock, printed in units of 100 ticks ~ 0.05 us) of k_seg_fwd on a C5-like batch (1024 ring graphs of 200 nodes,
// 424 directed edges, 128-d).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DHCG_SEG_STAMP -o tools/probe_tall_seg tools/probe_seg.hip -L hcatgnet_amd/csrc -lhcatgnet_hip '-Wl,-rpath,$ORIGIN/../hcatgnet_amd/csrc'
#include "../hcatgnet_amd/csrc/tall.hip"
#include <cstdio>
#include <vector>
#define CKH(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  const int B = 1024, n = 200, D = 128, N = B * n;
  std::vector<int> gp(B + 1), ep(B + 1);
  std::vector<long long> src, dst;
  for (int g = 0; g < B; ++g) {
    gp[g] = g * n; ep[g] = (int)src.size();
    auto bond = [&](int i, int j) { src.push_back(g * n + i); dst.push_back(g * n + j); src.push_back(g * n + j); dst.push_back(g * n + i); };
    for (int i = 0; i < n; ++i) bond(i, (i + 1) % n);
    for (int k = 0; k < 12; ++k) bond(3 * k, (3 * k + 50 + 7 * k) % n);
  }
  gp[B] = N; ep[B] = (int)src.size();
  const int E = (int)src.size();
  std::vector<long long> ei(2 * (size_t)E);
  for (int e = 0; e < E; ++e) { ei[e] = src[e]; ei[E + e] = dst[e]; }
  float *H, *out, *emb, *bias; long long* dei; int *dgp, *dep, *dstatus;
  CKH(hipMalloc(&H, (size_t)N * D * 4)); CKH(hipMalloc(&out, (size_t)N * D * 4)); CKH(hipMalloc(&emb, (size_t)B * 2 * D * 4));
  CKH(hipMalloc(&bias, D * 4)); CKH(hipMalloc(&dei, ei.size() * 8)); CKH(hipMalloc(&dgp, (B + 1) * 4)); CKH(hipMalloc(&dep, (B + 1) * 4));
  CKH(hipMalloc(&dstatus, 16));
  std::vector<float> h((size_t)N * D);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2000) / 1000.f - 1.f;
  CKH(hipMemcpy(H, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CKH(hipMemcpy(bias, h.data(), D * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dei, ei.data(), ei.size() * 8, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dgp, gp.data(), (B + 1) * 4, hipMemcpyHostToDevice)); CKH(hipMemcpy(dep, ep.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CKH(hipMemset(dstatus, 0, 16));
  const int npad = seg_npad(n);
  const size_t slds = seg_tile_bytes(npad);
  const size_t wbuf = (size_t)2 * WCH * sizeof(short);
  short* img; CKH(hipMalloc(&img, 3 * 128 * 128 * 2));
  float* W; CKH(hipMalloc(&W, 128 * 128 * 4)); CKH(hipMemcpy(W, h.data(), 128 * 128 * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_split_weight, dim3(64), dim3(256), 0, 0, W, 128, 128, 128, img);
  CKH(hipFuncSetAttribute((const void*)k_seg_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 512 - sizeof(SegLdsT<false>))));
  hipEvent_t e0, e1; CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
  auto launch = [&]() {
    hipLaunchKernelGGL((k_seg_fwd<true>), dim3(seg_grid(B)), dim3(SN), slds + wbuf, 0, H, D, 128, (const short*)img, bias, (const int64_t*)dei, (int64_t)E, dgp, dep, B, npad, 0.01f, 1, out, emb, dstatus);
  };
  for (int i = 0; i < 5; ++i) launch();
  CKH(hipDeviceSynchronize());
  CKH(hipEventRecord(e0, 0));
  for (int i = 0; i < 50; ++i) launch();
  CKH(hipEventRecord(e1, 0)); CKH(hipEventSynchronize(e1));
  float ms; CKH(hipEventElapsedTime(&ms, e0, e1));
  int st; CKH(hipMemcpy(&st, dstatus, 4, hipMemcpyDeviceToHost));
  printf("k_seg_fwd<pool>: %.1f us (status %d, E %d)\n", ms * 1000.f / 50, st, E);
#ifdef HCG_SEG_KSTAMP
  {
    unsigned long long ks[4 * 2 * 8 * 6];
    CKH(hipMemcpyFromSymbol(ks, HIP_SYMBOL(g_seg_kstamp), sizeof(ks)));
    for (int b = 0; b < 2; ++b)
      for (int it = 0; it < 2; ++it) {
        printf("block %d graph %d k-steps (ticks: top -> MFMAs issued -> next A split -> chunk stored -> barrier passed):\n", b, it);
        for (int k = 0; k < 8; ++k) {
          const unsigned long long* q = ks + ((b * 2 + it) * 8 + k) * 6;
          printf("   ks %d: %6llu %6llu %6llu %6llu   (k-step total %llu)\n", k, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[4] - q[0]);
        }
      }
  }
#endif
#ifdef HCG_SEG_STAMP
  unsigned long long stamp[8 * 8 * 8];
  CKH(hipMemcpyFromSymbol(stamp, HIP_SYMBOL(g_seg_stamp), sizeof(stamp)));
  for (int b = 0; b < 3; ++b) {
    const unsigned long long t0 = stamp[(b * 8 + 0) * 8 + 0];
    for (int it = 0; it < 4; ++it) {
      printf("block %d graph %d:", b, it);
      for (int ph = 0; ph < 6; ++ph) printf(" %7.2f", (double)(stamp[(b * 8 + it) * 8 + ph] - t0) / 100.0);
      printf("  x100 ticks (start, csr done, tile written, barrier, sums+stores issued, barrier)\n");
    }
  }
#endif
  return 0;
}
