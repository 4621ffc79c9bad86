// Dev probe (not product): the stacked forward with the readout head in its tail (hcg_fused_forward, training form) on a
// C3-like batch, with s_memtime stamps of the head phases (head_tile.h: H16STAMP) of the first workgroups' waves.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DHCG_HEAD_STAMP -o tools/probe_head_tail tools/probe_head_tail.hip
#include "../hcatgnet_amd/csrc/fused.hip"
#include <cstdio>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const int B = 4096, n = 30, F = 64, D = 64, N = B * n;
  std::mt19937 rng(1);
  std::vector<int> gp(B + 1), ep(B + 1);
  std::vector<long long> src, dst;
  std::vector<float> x((size_t)N * F), W(D * F), bias(D, 0.1f), W0(64 * 128), y(B, 1.f);
  for (auto& v : x) v = (float)(rng() % 2000) / 1000.f - 1.f;
  for (auto& v : W) v = (float)(rng() % 2000) / 8000.f - 0.125f;
  for (auto& v : W0) v = (float)(rng() % 2000) / 8000.f - 0.125f;
  for (int g = 0; g < B; ++g) {
    gp[g] = g * n; ep[g] = (int)src.size();
    auto bond = [&](int i, int j) { src.push_back(g * n + i); dst.push_back(g * n + j); src.push_back(g * n + j); dst.push_back(g * n + i); };
    for (int i = 0; i < n; ++i) bond(i, (i + 1) % n);
    bond(0, 15); bond(7, 22);
  }
  gp[B] = N; ep[B] = (int)src.size();
  const int E = (int)src.size();
  std::vector<long long> ei(2 * (size_t)E);
  for (int e = 0; e < E; ++e) { ei[e] = src[e]; ei[E + e] = dst[e]; }
  float *dx, *dW, *db, *dW2, *dW0, *dy, *dout1, *demb0, *dz, *dout, *ddemb; long long* dei; int *dgp, *dep, *dstatus; uint32_t* bits; void* hws;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dW2, W.size() * 4)); CK(hipMalloc(&db, D * 4));
  CK(hipMalloc(&dW0, W0.size() * 4)); CK(hipMalloc(&dy, B * 4)); CK(hipMalloc(&dout1, (size_t)N * D * 4));
  CK(hipMalloc(&demb0, (size_t)B * 2 * D * 4)); CK(hipMalloc(&ddemb, (size_t)B * 2 * D * 4)); CK(hipMalloc(&dz, (size_t)B * D * 4));
  CK(hipMalloc(&dout, B * 4)); CK(hipMalloc(&dei, ei.size() * 8)); CK(hipMalloc(&dgp, (B + 1) * 4)); CK(hipMalloc(&dep, (B + 1) * 4));
  CK(hipMalloc(&dstatus, 16)); CK(hipMalloc(&bits, hcg_fused_aux_bytes(HCG_FUSED_POOLBITS, B, 1)));
  const size_t hwb = hcg_fused_aux_bytes(HCG_FUSED_HEAD_WS, B, 1); CK(hipMalloc(&hws, hwb));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW2, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bias.data(), D * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW0, W0.data(), W0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dei, ei.data(), ei.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dgp, gp.data(), (B + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dep, ep.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dstatus, 0, 16));
  hcg_fused_fwd_args a{};
  a.x = dx; a.W1 = dW; a.b1 = db; a.W2 = dW2; a.b2 = db; a.edge_index = (const int64_t*)dei; a.E = E; a.graph_ptr = dgp; a.edge_ptr = dep;
  a.N = N; a.B = B; a.F = F; a.D = D; a.graphs_per_tile = 1; a.apply_act = 1; a.slope = 0.01f; a.out1 = dout1; a.emb = demb0;
  a.poolbits = bits; a.status = dstatus;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int head = 0; head < 2; ++head) {
    if (head) {
      a.y = dy; a.head_W0 = dW0; a.head_b0 = db; a.head_W1 = dW0; a.head_b1 = db; a.C = 1; a.z = dz; a.out = dout; a.demb = ddemb;
      a.head_workspace = hws; a.head_workspace_bytes = hwb;
    }
    for (int it = 0; it < 55; ++it) {
      if (it == 5) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
      int rc = hcg_fused_forward(&a, 0);
      if (rc) { printf("rc %d\n", rc); return 1; }
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stacked forward, training form, head %d: %.2f us per launch\n", head, ms * 1000.f / 50);
  }
#ifdef HCG_HEAD_STAMP
  unsigned long long st[4][8][16];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_h16_stamp), sizeof(st)));
  const char* nm[] = {"", "prefetch issue + barrier (wait for the slowest wave)", "rows", "tile entry (+ wait for the fragments)", "fwd MFMA", "B2+z",
                      "B3+out/dz", "B4", "bwd MFMA+demb", "dW0 slab", "small+end"};
  for (int blk = 0; blk < 2; ++blk)
    for (int w : {0, 3, 4, 7}) {
      printf("block %d wave %d (s_memtime: core clocks):", blk, w);
      for (int i = 1; i <= 10; ++i) printf(" %s %llu |", nm[i], st[blk][w][i] - st[blk][w][i - 1]);
      printf(" total %llu\n", st[blk][w][10] - st[blk][w][0]);
    }
#endif
  return 0;
}
