// Dev probe (not product): times the fused forward kernel on a C2-like batch and prints s_memtime
// phase stamps of a few waves.  Build: hipcc -O3 --offload-arch=gfx950 -DHCG_STAMP -o /tmp/probe tools/probe_fused.hip
// pass -DHCG_STAMP for the s_memtime build
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
  std::vector<float> x((size_t)N * F), W(D * F), bias(D, 0.1f);
  for (auto& v : x) v = (float)(rng() % 2000) / 1000.f - 1.f;
  for (auto& v : W) v = (float)(rng() % 2000) / 8000.f - 0.125f;
  // ring + 2 chords per graph: 32 bonds -> 64 directed edges
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
  printf("N %d E %d\n", N, E);
  float *dx, *dW, *db, *dout, *demb; long long* dei; int *dgp, *dep, *dstatus; unsigned long long* dstamp;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&db, D * 4));
  CK(hipMalloc(&dout, (size_t)N * D * 4)); CK(hipMalloc(&demb, (size_t)B * 2 * D * 4));
  CK(hipMalloc(&dei, ei.size() * 8)); CK(hipMalloc(&dgp, (B + 1) * 4)); CK(hipMalloc(&dep, (B + 1) * 4));
  CK(hipMalloc(&dstatus, 16)); CK(hipMalloc(&dstamp, 4 * WAVES * 64 * 8));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), D * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dei, ei.data(), ei.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dgp, gp.data(), (B + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dep, ep.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dstatus, 0, 16)); CK(hipMemset(dstamp, 0, 4 * WAVES * 64 * 8));
#ifdef HCG_STAMP
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dstamp, sizeof(dstamp)));
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int pool = 0; pool < 2; ++pool) {
    for (int it = 0; it < 5; ++it)
      hcg_fused_layer_fwd(dx, dW, db, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 1, 0.01f, 1, dout, pool ? demb : nullptr, dstatus, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int iters = 50;
    for (int it = 0; it < iters; ++it)
      hcg_fused_layer_fwd(dx, dW, db, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 1, 0.01f, 1, dout, pool ? demb : nullptr, dstatus, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fwd pool=%d: %.2f us per launch\n", pool, ms * 1000.f / iters);
  }
  {  // backward kernels (layer 2: pooled gradient + dx; layer 1: dout, no dx)
    float *ddx, *dws; CK(hipMalloc(&ddx, (size_t)N * F * 4));
    size_t wsb = hcg_fused_workspace_bytes(B, F, D, 1); CK(hipMalloc(&dws, wsb));
    CK(hipMemset(demb, 0, (size_t)B * 2 * D * 4));
    for (int variant = 0; variant < 2; ++variant) {
      for (int it = 0; it < 55; ++it) {
        if (it == 5) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
        if (variant == 0) hcg_fused_layer_bwd(nullptr, demb, demb, dout, dx, dW, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 1, 0.01f, 1, ddx, dstatus, dws, wsb, 0);
        else hcg_fused_layer_bwd(dout, nullptr, nullptr, dout, dx, dW, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 1, 0.01f, 1, nullptr, dstatus, dws, wsb, 0);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("bwd %s: %.2f us per launch\n", variant == 0 ? "layer2 (pooled grad + dx)" : "layer1 (dout, no dx)", ms * 1000.f / 50);
    }
  }
  {  // stacked two-layer forward (the product's forward launch); its stamps are the ones printed below
    float *dW2, *dout2;
    CK(hipMalloc(&dW2, W.size() * 4)); CK(hipMalloc(&dout2, (size_t)N * D * 4));
    CK(hipMemcpy(dW2, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    for (int it = 0; it < 55; ++it) {
      if (it == 5) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
      hcg_fused_stack2_fwd_train(dx, dW, db, dW2, db, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 1, 0.01f, 1, dout, demb, (uint32_t*)dout2, dstatus, 0);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stack2 fwd (training form): %.2f us per launch\n", ms * 1000.f / 50);
  }
#ifndef HCG_STAMP
  return 0;
#endif
#ifdef HCG_STAMP_BWD
  {  // the product's layer-2 backward (bit form, premasked dx) LAST: its stamps are the ones in the buffer
    float *ddx, *dws; CK(hipMalloc(&ddx, (size_t)N * F * 4));
    uint32_t* bits; CK(hipMalloc(&bits, hcg_fused_poolbits_bytes(B, 1))); CK(hipMemset(bits, 0x55, hcg_fused_poolbits_bytes(B, 1)));
    size_t wsb = hcg_fused_workspace_bytes(B, F, D, 1); CK(hipMalloc(&dws, wsb));
    for (int it = 0; it < 25; ++it) {
      if (it == 5) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
      hcg_fused_layer_bwd_poolbits(demb, bits, dx, dW, (const int64_t*)dei, E, dgp, dep, N, B, F, D, 1, 0.01f, 3, ddx, dstatus, dws, wsb, 0);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("bwd layer 2, bit form + premask: %.2f us per launch\n", ms * 1000.f / 20);
    std::vector<unsigned long long> sb(4 * WAVES * 64);
    CK(hipMemcpy(sb.data(), dstamp, sb.size() * 8, hipMemcpyDeviceToHost));
    const char* nm[] = {"count build", "dY'", "agg MFMA", "x staged", "dW MFMA", "dx GEMM", "dx stores", "next loads"};
    for (int w : {0, 1, 8, 17}) {
      const unsigned long long* s = sb.data() + (size_t)w * 64;
      printf("wave %2d:", w);
      for (int it = 0; it < 3 && s[1 + 8 * it]; ++it) {
        printf(" tile%d:", it);
        for (int k = 0; k < 8; ++k) printf(" %s %llu", nm[k], s[k + 1 + 8 * it] - (k == 0 ? (it == 0 ? s[0] : s[8 * it]) : s[k + 8 * it]));
        printf(" |");
      }
      printf(" combine+slab %llu | total %llu\n", s[63] - s[60], s[63] - s[0]);
    }
    return 0;
  }
#endif
  std::vector<unsigned long long> st(4 * WAVES * 64);
  CK(hipMemcpy(st.data(), dstamp, st.size() * 8, hipMemcpyDeviceToHost));
  int status[4]; CK(hipMemcpy(status, dstatus, 16, hipMemcpyDeviceToHost)); printf("status %d\n", status[0]);
  const char* names[] = {"", "tile start", "prefetch-issue", "gemm", "scale", "epilogue+store+nextstage", "agg-mfma"};
  for (int w : {0, 1, 8, 9, 17}) {
    unsigned long long* s = &st[(size_t)w * 64];
    printf("wave %2d: first-stage %llu | ", w, s[1] - s[0]);
    for (int it = 0; it < 3; ++it) {
      if (!s[1 + 8 * it]) break;
      printf("tile%d:", it);
      printf(" %s %llu", names[2], s[2 + 8 * it] - s[1 + 8 * it]);
      printf(" %s %llu", names[3], s[3 + 8 * it] - s[2 + 8 * it]);
      printf(" %s %llu", names[4], s[4 + 8 * it] - s[3 + 8 * it]);
      printf(" %s %llu", names[6], s[6 + 8 * it] - s[4 + 8 * it]);
      printf(" epi-LDSwrite %llu readback+stores %llu", s[7 + 8 * it] - s[6 + 8 * it], s[5 + 8 * it] - s[7 + 8 * it]);
      if (s[1 + 8 * (it + 1)]) printf(" stage-next %llu", s[1 + 8 * (it + 1)] - s[5 + 8 * it]);
      printf(" | ");
    }
    printf("\n   tile0 L1-stores %llu | L2: gemm %llu scale %llu agg %llu epiLDS %llu stores %llu | pool %llu\n   ", s[40] - s[7], s[41] - s[40], s[42] - s[41],
           s[43] - s[42], s[44] - s[43], s[45] - s[44], s[5] - s[45]);
    printf("total %llu cyc | last stage-next: wait+xwrite %llu build %llu (of it: before %llu)\n", s[63] - s[0], s[57] - s[56], s[58] - s[57], s[56] - s[5]);
  }
  return 0;
}
