// Dev probe (not product): the one-graph-per-workgroup forward of csrc/mid.hip (k_mid_layer_fwd) on reference-sized
// graphs, with s_memtime stamps of the per-graph phases (mid.hip: MSTAMP under -DHCG_MID_STAMP).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DHCG_MID_STAMP -o tools/probe_mid tools/probe_mid.hip \
//         -L hcatgnet_amd/csrc -lhcatgnet_hip '-Wl,-rpath,$ORIGIN/../hcatgnet_amd/csrc'
#include "../hcatgnet_amd/csrc/mid.hip"
#include <cstdio>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int B = 4096, F = argc > 1 ? atoi(argv[1]) : 25, D = 64;
  std::mt19937 rng(1);
  std::vector<int> gp(B + 1), ep(B + 1);
  std::vector<long long> src, dst;
  int N = 0;
  for (int g = 0; g < B; ++g) {
    const int n = 57 + (int)(rng() % 61);           // 57 .. 117 atoms
    gp[g] = N; ep[g] = (int)src.size();
    auto bond = [&](int i, int j) { src.push_back(N + i); dst.push_back(N + j); src.push_back(N + j); dst.push_back(N + i); };
    for (int i = 1; i < n; ++i) bond(i, (int)(rng() % i) < i - 3 ? i - 1 : (int)(rng() % i));
    for (int c = 0; c < 4; ++c) bond((int)(rng() % (n / 2)), n / 2 + (int)(rng() % (n / 2)));
    N += n;
  }
  gp[B] = N; ep[B] = (int)src.size();
  const int E = (int)src.size();
  std::vector<long long> ei(2 * (size_t)E);
  for (int e = 0; e < E; ++e) { ei[e] = src[e]; ei[E + e] = dst[e]; }
  std::vector<float> x((size_t)N * F), W(D * F), bias(D, 0.1f);
  for (auto& v : x) v = (float)(rng() % 2000) / 1000.f - 1.f;
  for (auto& v : W) v = (float)(rng() % 2000) / 8000.f - 0.125f;
  printf("N %d E %d\n", N, E);
  float *dx, *dW, *db, *dout, *demb; long long* dei; int *dgp, *dep, *dstatus;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&db, D * 4));
  CK(hipMalloc(&dout, (size_t)N * D * 4)); CK(hipMalloc(&demb, (size_t)B * 2 * D * 4));
  CK(hipMalloc(&dei, ei.size() * 8)); CK(hipMalloc(&dgp, (B + 1) * 4)); CK(hipMalloc(&dep, (B + 1) * 4)); CK(hipMalloc(&dstatus, 16));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dei, ei.data(), ei.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dgp, gp.data(), (B + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dep, ep.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dstatus, 0, 16));
  const bool bwd = argc > 2 && argv[2][0] == 'b';
  float *da = nullptr, *ddx = nullptr, *dws = nullptr, *ddemb = nullptr;
  size_t wsb = 0;
  if (bwd) {   // pooled form for F = 64 (layer 2: demb / emb / a_out -> dx), dout form otherwise (layer 1: no dx)
    wsb = hcg_mid_workspace_bytes(B, F, D, 117, 300);
    CK(hipMalloc(&da, (size_t)N * D * 4)); CK(hipMalloc(&ddx, (size_t)N * F * 4)); CK(hipMalloc(&dws, wsb)); CK(hipMalloc(&ddemb, (size_t)B * 2 * D * 4));
    CK(hipMemset(da, 0, (size_t)N * D * 4)); CK(hipMemset(ddemb, 0, (size_t)B * 2 * D * 4)); CK(hipMemset(demb, 0, (size_t)B * 2 * D * 4));
    CK(hipMemset(dout, 0, (size_t)N * D * 4));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 25; ++it) {
    if (it == 5) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
    int rc;
    if (!bwd) rc = hcg_mid_layer_fwd(dx, dW, db, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 117, 300, 0.01f, 1, dout, F == 64 ? demb : nullptr, nullptr, nullptr, nullptr, dstatus, 0);
    else if (F == 64) rc = hcg_mid_layer_bwd(nullptr, ddemb, demb, da, dx, dW, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 117, 300, 0.01f, 3, ddx, dstatus, dws, wsb, 0);
    else rc = hcg_mid_layer_bwd(dout, nullptr, nullptr, nullptr, dx, dW, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 117, 300, 0.01f, 0, nullptr, dstatus, dws, wsb, 0);
    if (rc) { printf("rc %d\n", rc); return 1; }
  }
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("mid %s (F = %d): %.2f us per launch\n", bwd ? "backward" : "forward", F, ms * 1000.f / 20);
#ifdef HCG_MID_STAMP
  static unsigned long long st[MW][4][16];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_mid_stamp), sizeof(st)));
  if (!bwd) {
    const int idx[] = {0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11};
    const char* nm[] = {"", "count + slots + x write", "barrier", "dinv (+ CSR route)", "GEMM", "H' write", "barrier",
                        "prefetch issue", "aggregate + store", "pool", "end barrier"};
    for (int w : {0, 2, 5, 7})
      for (int it = 0; it < 3; ++it) {
        printf("wave %d graph %d:", w, it);
        for (int i = 1; i < 11; ++i) printf(" %s %llu |", nm[i], st[w][it][idx[i]] - st[w][it][idx[i - 1]]);
        printf(" total %llu\n", st[w][it][11] - st[w][it][0]);
      }
  } else {
    const char* nm[] = {"", "row loads issued", "count / scan / fill", "sort", "pool ties", "dY' tile + barrier", "transpose sum", "barrier",
                        "x write + barrier", "dW k-loop", "dx GEMM + stores", "end barrier"};
    for (int w : {0, 2, 5, 7})
      for (int it = 0; it < 3; ++it) {
        printf("wave %d graph %d:", w, it);
        for (int i = 1; i < 12; ++i) printf(" %s %llu |", nm[i], st[w][it][i] - st[w][it][i - 1]);
        printf(" total %llu\n", st[w][it][11] - st[w][it][0]);
      }
  }
#endif
  return 0;
}
