// Dev probe (not product): can the exact-f32 MFMA contractions of fused.hip be replaced by split-bf16
// MFMAs (x = x1 + x2 + x3, three bf16 pieces; 6 of the 9 cross products kept) without leaving the 1e-5
// parity budget, and what does a tile cost?  Prints max errors vs an fp64 host reference and timings.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/probe_bf16x tools/probe_bf16x.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// 8 floats -> three bf16x8 fragments (RNE pieces: x ~= p1 + p2 + p3 to 2^-24)
struct Split3 { bf16x8 p1, p2, p3; };
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
  u32x4 a, b, c;
  float r[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u = cvt_pk_bf16(x[2 * j], x[2 * j + 1]);
    a[j] = u;
    r[2 * j] = x[2 * j] - __uint_as_float(u << 16);
    r[2 * j + 1] = x[2 * j + 1] - __uint_as_float(u & 0xffff0000u);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u = cvt_pk_bf16(r[2 * j], r[2 * j + 1]);
    b[j] = u;
    r[2 * j] -= __uint_as_float(u << 16);
    r[2 * j + 1] -= __uint_as_float(u & 0xffff0000u);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = cvt_pk_bf16(r[2 * j], r[2 * j + 1]);
  Split3 s;
  s.p1 = __builtin_bit_cast(bf16x8, a);
  s.p2 = __builtin_bit_cast(bf16x8, b);
  s.p3 = __builtin_bit_cast(bf16x8, c);
  return s;
}

#define MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)

__device__ __forceinline__ constexpr int krow(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// mode 0: f32 MFMA, 1: bf16 x6, 2: bf16 x3 (hi*hi, hi*mid, mid*hi)
// out[32][64] = X[32][64] W^T (W [64][64]); then agg[32][64] = Cnt[32][32] * out  (accumulators as B operand)
__global__ void k_check(const float* X, const float* W, const float* Cnt, float* out, float* agg, int mode) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc[2];
  for (int nb = 0; nb < 2; ++nb) for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  if (mode == 0) {
    for (int k = 0; k < 64; k += 2) {
      const float a = X[r * 64 + k + h];
      for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, W[(nb * 32 + r) * 64 + k + h], acc[nb], 0, 0, 0);
    }
  } else {
    for (int s = 0; s < 4; ++s) {
      float xa[8], wb[8];
      for (int j = 0; j < 8; ++j) xa[j] = X[r * 64 + 16 * s + 8 * h + j];
      const Split3 A = split3(xa);
      for (int nb = 0; nb < 2; ++nb) {
        for (int j = 0; j < 8; ++j) wb[j] = W[(nb * 32 + r) * 64 + 16 * s + 8 * h + j];
        const Split3 Bs = split3(wb);
        if (mode == 1) {
          acc[nb] = MFMA_BF16(A.p3, Bs.p1, acc[nb]);
          acc[nb] = MFMA_BF16(A.p1, Bs.p3, acc[nb]);
          acc[nb] = MFMA_BF16(A.p2, Bs.p2, acc[nb]);
        }
        acc[nb] = MFMA_BF16(A.p2, Bs.p1, acc[nb]);
        acc[nb] = MFMA_BF16(A.p1, Bs.p2, acc[nb]);
        acc[nb] = MFMA_BF16(A.p1, Bs.p1, acc[nb]);
      }
    }
  }
  for (int nb = 0; nb < 2; ++nb)
    for (int i = 0; i < 16; ++i) out[krow(i, h) * 64 + nb * 32 + r] = acc[nb][i];
  // aggregation: Y[m][col] = sum_k Cnt[m][k] H[k][col]; H = acc (k = row index)
  f32x16 y[2];
  for (int nb = 0; nb < 2; ++nb) for (int i = 0; i < 16; ++i) y[nb][i] = 0.f;
  if (mode == 0) {
    for (int i = 0; i < 16; ++i) {
      const float a = Cnt[r * 32 + krow(i, h)];
      for (int nb = 0; nb < 2; ++nb) y[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[nb][i], y[nb], 0, 0, 0);
    }
  } else {
    for (int s = 0; s < 2; ++s) {
      float ca[8];
      for (int j = 0; j < 8; ++j) ca[j] = Cnt[r * 32 + krow(8 * s + j, h)];
      const Split3 A = split3(ca);   // small integers: exact in p1, p2 = p3 = 0
      for (int nb = 0; nb < 2; ++nb) {
        float hb[8];
        for (int j = 0; j < 8; ++j) hb[j] = acc[nb][8 * s + j];
        const Split3 Bs = split3(hb);
        if (mode == 1) y[nb] = MFMA_BF16(A.p1, Bs.p3, y[nb]);
        y[nb] = MFMA_BF16(A.p1, Bs.p2, y[nb]);
        y[nb] = MFMA_BF16(A.p1, Bs.p1, y[nb]);
      }
    }
  }
  for (int nb = 0; nb < 2; ++nb)
    for (int i = 0; i < 16; ++i) agg[krow(i, h) * 64 + nb * 32 + r] = y[nb][i];
}

// timing: every wave runs `iters` tile GEMMs (32x64x64) + aggregations out of LDS, 8 waves per workgroup
constexpr int HS = 68, WS = 72;   // f32 tile row stride (floats); bf16 W row stride (shorts): 144 B
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_time(const float* X, const float* W, float* sink, int iters) {
  __shared__ float buf[8][32 * HS];
  __shared__ short wl[3][64 * WS];
  __shared__ float wf[64 * 65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  for (int idx = lane; idx < 32 * 64; idx += 64) buf[wave][(idx >> 6) * HS + (idx & 63)] = X[idx];
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 512) {
    const float w = W[idx];
    wf[(idx >> 6) * 65 + (idx & 63)] = w;
    const unsigned u1 = cvt_pk_bf16(w, 0.f) & 0xffff;
    const float r1 = w - __uint_as_float(u1 << 16);
    const unsigned u2 = cvt_pk_bf16(r1, 0.f) & 0xffff;
    const float r2 = r1 - __uint_as_float(u2 << 16);
    const unsigned u3 = cvt_pk_bf16(r2, 0.f) & 0xffff;
    wl[0][(idx >> 6) * WS + (idx & 63)] = (short)u1;
    wl[1][(idx >> 6) * WS + (idx & 63)] = (short)u2;
    wl[2][(idx >> 6) * WS + (idx & 63)] = (short)u3;
  }
  __syncthreads();
  float tot = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc[2], y[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[nb][i] = 0.f; y[nb][i] = 0.f; }
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(&buf[wave][r * HS + 8 * t + 4 * h]);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wf[(nb * 32 + r) * 65 + 8 * t + 4 * h + u], acc[nb], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) y[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)(i == r), acc[nb][i], y[nb], 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 a0 = *reinterpret_cast<const float4*>(&buf[wave][r * HS + 16 * s + 8 * h]);
        const float4 a1 = *reinterpret_cast<const float4*>(&buf[wave][r * HS + 16 * s + 8 * h + 4]);
        const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const Split3 A = split3(xa);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&wl[0][(nb * 32 + r) * WS + 16 * s + 8 * h]);
          const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(&wl[1][(nb * 32 + r) * WS + 16 * s + 8 * h]);
          const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(&wl[2][(nb * 32 + r) * WS + 16 * s + 8 * h]);
          acc[nb] = MFMA_BF16(A.p3, b1, acc[nb]);
          acc[nb] = MFMA_BF16(A.p1, b3, acc[nb]);
          acc[nb] = MFMA_BF16(A.p2, b2, acc[nb]);
          acc[nb] = MFMA_BF16(A.p2, b1, acc[nb]);
          acc[nb] = MFMA_BF16(A.p1, b2, acc[nb]);
          acc[nb] = MFMA_BF16(A.p1, b1, acc[nb]);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float ca[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ca[j] = (float)(krow(8 * s + j, h) == r);
        u32x4 cu;
#pragma unroll
        for (int j = 0; j < 4; ++j) cu[j] = cvt_pk_bf16(ca[2 * j], ca[2 * j + 1]);
        const bf16x8 A1 = __builtin_bit_cast(bf16x8, cu);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          float hb[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) hb[j] = acc[nb][8 * s + j];
          const Split3 Bs = split3(hb);
          y[nb] = MFMA_BF16(A1, Bs.p3, y[nb]);
          y[nb] = MFMA_BF16(A1, Bs.p2, y[nb]);
          y[nb] = MFMA_BF16(A1, Bs.p1, y[nb]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += y[0][i] + y[1][i];
    buf[wave][r * HS + (it & 63)] += 1e-9f * tot;   // loop-carried dependence through LDS: nothing hoists
  }
  if (tot == 12345.f) sink[0] = tot;
}

int main() {
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (int scale_case = 0; scale_case < 3; ++scale_case) {
    std::vector<float> X(32 * 64), W(64 * 64), C(32 * 32, 0.f);
    for (auto& v : X) { v = nd(rng); if (scale_case == 1) v *= std::exp(4.f * nd(rng)); if (scale_case == 2) v = (float)(rng() % 2); }
    for (auto& v : W) v = 0.2f * nd(rng);
    for (int i = 0; i < 32; ++i) { C[i * 32 + i] = 1.f; for (int k = 0; k < 3; ++k) C[i * 32 + (int)(rng() % 32)] += 1.f; }
    std::vector<double> ref(32 * 64), refa(32 * 64), mag(32 * 64), maga(32 * 64);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 64; ++n) {
      double s = 0, a = 0;
      for (int k = 0; k < 64; ++k) { s += (double)X[m * 64 + k] * W[n * 64 + k]; a += std::fabs((double)X[m * 64 + k] * W[n * 64 + k]); }
      ref[m * 64 + n] = s; mag[m * 64 + n] = a;
    }
    float *dX, *dW, *dC, *dout, *dagg;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMalloc(&dout, 32 * 64 * 4)); CK(hipMalloc(&dagg, 32 * 64 * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; ++mode) {
      hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dX, dW, dC, dout, dagg, mode);
      CK(hipDeviceSynchronize());
      std::vector<float> out(32 * 64), agg(32 * 64);
      CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(agg.data(), dagg, agg.size() * 4, hipMemcpyDeviceToHost));
      double e_inf = 0, r_inf = 0, e_rel_mag = 0, ea_inf = 0, ra_inf = 0;
      for (int i = 0; i < 32 * 64; ++i) {
        e_inf = std::fmax(e_inf, std::fabs(out[i] - ref[i])); r_inf = std::fmax(r_inf, std::fabs(ref[i]));
        e_rel_mag = std::fmax(e_rel_mag, std::fabs(out[i] - ref[i]) / mag[i]);
      }
      for (int m = 0; m < 32; ++m) for (int n = 0; n < 64; ++n) {   // aggregation reference from the DEVICE's own H (isolates step 2)
        double s = 0;
        for (int k = 0; k < 32; ++k) s += (double)C[m * 32 + k] * out[k * 64 + n];
        ea_inf = std::fmax(ea_inf, std::fabs(agg[m * 64 + n] - s)); ra_inf = std::fmax(ra_inf, std::fabs(s));
      }
      printf("case %d mode %d: gemm |err|inf/|ref|inf %.3e  max err/sum|ab| %.3e   agg |err|inf/|ref|inf %.3e\n", scale_case, mode,
             e_inf / r_inf, e_rel_mag, ea_inf / ra_inf);
    }
  }
  // timing
  float *dX, *dW, *dsink;
  CK(hipMalloc(&dX, 32 * 64 * 4)); CK(hipMalloc(&dW, 64 * 64 * 4)); CK(hipMalloc(&dsink, 16));
  CK(hipMemset(dX, 0, 32 * 64 * 4)); CK(hipMemset(dW, 0, 64 * 64 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 64;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(256), dim3(512), 0, 0, dX, dW, dsink, iters);
      else hipLaunchKernelGGL(k_time<1>, dim3(256), dim3(512), 0, 0, dX, dW, dsink, iters);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("time mode %d: %.1f us for %d tiles/wave, 2 waves/SIMD -> %.2f us per tile-layer per wave pair\n", mode, ms * 1e3, iters, ms * 1e3 / iters);
    }
  }
  return 0;
}
