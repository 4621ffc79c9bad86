// Dev probe (not product): phase stamps of the one-launch head kernel.  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DHCG_HEAD_STAMP -o tools/probe_head tools/probe_head.hip
#include "../hcatgnet_amd/csrc/head.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  const int B = 4096, C = 1;
  float *emb, *y, *W0, *b0, *W1, *b1, *z, *out, *loss, *demb; void* ws; int* sync;
  size_t wsb = hcg_head_workspace_bytes(B);
  CK(hipMalloc(&emb, B * 128 * 4)); CK(hipMalloc(&y, B * 4)); CK(hipMalloc(&W0, 64 * 128 * 4)); CK(hipMalloc(&b0, 256)); CK(hipMalloc(&W1, 256)); CK(hipMalloc(&b1, 16));
  CK(hipMalloc(&z, B * 64 * 4)); CK(hipMalloc(&out, B * 4)); CK(hipMalloc(&loss, 16)); CK(hipMalloc(&demb, B * 128 * 4)); CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&sync, HCG_HEAD_SYNC_WORDS * 4));
  CK(hipMemset(emb, 0, B * 128 * 4)); CK(hipMemset(y, 0, B * 4)); CK(hipMemset(W0, 0, 64 * 128 * 4)); CK(hipMemset(b0, 0, 256)); CK(hipMemset(W1, 0, 256)); CK(hipMemset(b1, 0, 16));
  CK(hipMemset(sync, 0, HCG_HEAD_SYNC_WORDS * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 30; ++it) {
    if (it == 10) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
    int rc = hcg_head_fwd_bwd(emb, y, W0, b0, W1, b1, B, 64, C, 0.01f, 1, z, out, loss, demb, ws, wsb, sync, nullptr, 0);
    if (rc) { printf("rc %d\n", rc); return 1; }
  }
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("head: %.2f us per launch\n", ms * 1000.f / 20);
#ifdef HCG_HEAD_STAMP
  unsigned long long st[256];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_head_stamp), sizeof(st)));
  // stamps in program order: 0 start, 1 tile staged, 2 forward MFMA, 3 squared error (partial published right after),
  // 5 dz, 6 backward MFMA, 4 sum collected (the exchange wait ends), 7 sums folded
  const char* names[] = {"loads+W->LDS+emb", "fwd MFMA", "z/out/sse", "dz", "bwd MFMA", "wait for the sum", "scale+stores+fold"};
  const int order[] = {0, 1, 2, 3, 5, 6, 4, 7};
  for (int b : {0, 1, 7, 15}) {
    printf("block %2d:", b);
    for (int i = 0; i < 7; ++i) printf(" %s %llu |", names[i], st[b * 16 + order[i + 1]] - st[b * 16 + order[i]]);
    printf(" total %llu\n", st[b * 16 + 7] - st[b * 16]);
  }
  unsigned long long mn = ~0ull, mx = 0;
  for (int b = 0; b < 16; ++b) { mn = st[b * 16] < mn ? st[b * 16] : mn; mx = st[b * 16 + 7] > mx ? st[b * 16 + 7] : mx; }
  printf("first start -> last end over blocks 0..15: %llu ticks\n", mx - mn);
#endif
  return 0;
}
