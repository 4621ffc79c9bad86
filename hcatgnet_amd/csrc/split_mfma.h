// Split-bf16 MFMA building blocks shared by the fused small-graph kernels (fused.hip) and the mid-size graph
// kernels (mid.hip); gfx950 only.  Numerics and measurements: see the header comment of fused.hip.
#pragma once
#include "common.h"

namespace {

constexpr int DD = 64;          // layer width of the fused kernels (embedding_dim = 64)
constexpr int HS = DD + 4;      // LDS row stride (floats) of an fp32 [rows][DD] tile
constexpr int WPAD = 8;         // bf16 padding of a pre-split weight row: rows of (K + 8) * 2 bytes are 16-byte
                                // aligned and 16 consecutive rows cover all 64 LDS banks (ds_read_b128, conflict-free)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));   // one MFMA A/B fragment: 8 bf16 in 4 VGPRs
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// rows of a 32x32 MFMA accumulator: register i of lane-half h holds row krow(i, h)
__device__ __forceinline__ constexpr int krow(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- split-bf16 arithmetic ------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32 (round to nearest even)
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

struct Split3 { bf16x8 p1, p2, p3; };
// 8 f32 -> three bf16 fragments with x = p1 + p2 + p3 (each residual is exact in f32)
// (the residuals are PAIR subtractions, v_pk_add_f32: same IEEE results, one instruction per two values -- the split is most
//  of the VALU work of every MFMA kernel here, and those kernels are bound by VALU issue.  HCG_SPLIT_SCALAR: the one-value
//  form, for A/B measurements with tools/build_variants.sh)
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
  u32x4 a, b, c;
#ifndef HCG_SPLIT_SCALAR
  f32x2 r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u = pk_bf16(x[2 * j], x[2 * j + 1]);
    a[j] = u;
    r[j] = f32x2{x[2 * j], x[2 * j + 1]} - f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u = pk_bf16(r[j].x, r[j].y);
    b[j] = u;
    r[j] -= f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = pk_bf16(r[j].x, r[j].y);
#else
  float r[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u = pk_bf16(x[2 * j], x[2 * j + 1]);
    a[j] = u;
    r[2 * j] = x[2 * j] - __uint_as_float(u << 16);
    r[2 * j + 1] = x[2 * j + 1] - __uint_as_float(u & 0xffff0000u);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u = pk_bf16(r[2 * j], r[2 * j + 1]);
    b[j] = u;
    r[2 * j] -= __uint_as_float(u << 16);
    r[2 * j + 1] -= __uint_as_float(u & 0xffff0000u);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = pk_bf16(r[2 * j], r[2 * j + 1]);
#endif
  Split3 s;
  s.p1 = __builtin_bit_cast(bf16x8, a);
  s.p2 = __builtin_bit_cast(bf16x8, b);
  s.p3 = __builtin_bit_cast(bf16x8, c);
  return s;
}

#define HCG_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)

// Distance between the LAST MFMA of a chain and the first VALU read of its result registers.
//
// Symptom (MI355X, tools/dbg_determinism.py: 80 launches of the stacked forward on 4096 tiles, two waves per SIMD): 11..31
// of 80 launches came back with ONE wrong 1x16 block -- lanes 48-63 of one accumulator register of the second layer,
// holding the value from before the chain's last MFMA landed.  Silent, timing dependent, a different tile every time.
//
// The instruction pair (ISA of the pre-fix build: this file without the s_nops below, no -fno-slp-vectorize;
// `hipcc -O3 -S`, k_fused_layer_fwd<64,true,true,true,true>):
//     v_mfma_f32_32x32x16_bf16 v[0:15], v[32:35], v[204:207], v[0:15]    ; last MFMA of the H = X W^T chain on v[0:15]
//     v_mfma_f32_32x32x16_bf16 v[16:31], v[32:35], v[232:235], v[16:31]
//     s_nop 10
//     v_pk_mul_f32 v[32:33], v[0:1], v[56:57]                            ; H' = dinv . H: first VALU read -- 12 wait states
// and, in the epilogue of the aggregation chain,
//     v_mfma_f32_32x32x16_bf16 v[0:15], v[52:55], v[20:23], v[0:15]      ; last MFMA of Y = (C + I) H'
//     s_nop 7 ; v_fma_f32 v16, v32, .. ; v_mul_f32 ; v_max_f32 ; s_nop 0
//     v_fma_f32 v0, v0, v56, v126                                        ; first read of v0 -- 12 wait states
// Every such site of the pre-fix object sits at EXACTLY 12 wait states (tools/isa_lint.py --dis: 173 sites closer than 19,
// none closer than 12): hipcc pads to the gfx950 hazard table, "XDL write VGPR -> VALU read / write" of an 8-pass MFMA
// (32 issue cycles: v_mfma_f32_32x32x16_bf16) = passes + 2 + 1 = 11 wait states (LLVM GCNHazardRecognizer,
// GFX940_XDL_N_PassWriteVgprVALURawWaitStates with the gfx950 +1).  So the object met the documented requirement and
// still failed.  What the table assumes is that the MFMA enters the matrix pipe when it issues; with TWO MFMA-dense waves
// on one SIMD the pipe is shared and fully paced (MI355X_MICROARCH.md, "Two waves per SIMD", item 1: a partner's MFMAs
// come straight out of your stream), so a partner's 8-pass MFMA can be ahead of ours in the pipe and our last pass --
// the 1x16 block that was stale -- lands up to 8 passes later than issue + 8.  Nothing interlocks a VALU read of an MFMA
// destination (that is what the software wait states are for; dependent MFMAs through srcC are interlocked, which is why
// the chains themselves never failed).  Safe distance at two waves per SIMD: 11 + 8 = 19 wait states.
//
// The model was then tested (round 2, tools/build_variants.sh + tools/stress_determinism.py): a fence of only 8 wait states
// (`s_nop 7`: 19-20 with the compiler's own padding, the modelled minimum) and one of 20 both returned 0 bad launches of 600
// (400 x C3, 100 x reference-sized, 100 x ragged batches), where 12 had failed 11-31 of 80.  They buy 0.8 % of the step
// (0.1106 vs 0.1115 ms: the partner wave fills most of the idle slots), so the product keeps the long fence as margin.
//
// Fix kept here: 64 idle quad-cycles (4 x s_nop 15) between a chain's last MFMA and the first non-MFMA touch of its
// accumulators (0 of 80 afterwards), and -fno-slp-vectorize for the MFMA files (v_pk_*_f32 beside MFMAs is slower on CDNA4
// anyway; it is not the cause: plain v_fma_f32 sites sat at the same 12).  Guards: tools/isa_lint.py (tests/
// test_isa_lint.py, CPU, every build: no non-MFMA instruction may touch an MFMA destination within 19 wait states in
// fused.o / mid.o / head.o) and tests/test_gpu_train_step.py::test_forty_launches_are_bitwise_identical (behaviour).
#ifndef HCG_FENCE_ASM      // (tools/build_variants.sh builds shorter fences for A/B measurements; the product build uses this one)
#define HCG_FENCE_ASM "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
#endif
__device__ __forceinline__ void mfma_results_fence(f32x16& a, f32x16& b) {
  asm volatile(HCG_FENCE_ASM : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void mfma_results_fence(f32x16& a) {
  asm volatile(HCG_FENCE_ASM : "+v"(a));
}

// acc += a * b with both operands split: the six cross terms >= 2^-24, smallest first
__device__ __forceinline__ void mfma_split(f32x16& acc, const Split3& a, const bf16x8& b1, const bf16x8& b2, const bf16x8& b3) {
  acc = HCG_MFMA(a.p3, b1, acc);
  acc = HCG_MFMA(a.p1, b3, acc);
  acc = HCG_MFMA(a.p2, b2, acc);
  acc = HCG_MFMA(a.p2, b1, acc);
  acc = HCG_MFMA(a.p1, b2, acc);
  acc = HCG_MFMA(a.p1, b1, acc);
}
// acc += a * b with an EXACT one-piece a (small integers) and a split b
__device__ __forceinline__ void mfma_exact_a(f32x16& acc, const bf16x8& a, const Split3& b) {
  acc = HCG_MFMA(a, b.p3, acc);
  acc = HCG_MFMA(a, b.p2, acc);
  acc = HCG_MFMA(a, b.p1, acc);
}

// Pre-split weight image in LDS: three bf16 planes [rows][K + WPAD]; plane p at wl + p * rows * (K + WPAD).
// Block-cooperative (every thread of the workgroup takes part), coalesced global reads.
//   TRANS = false: image row n, column k  <-  g[n * cols + k]   (B[k][n] = W[n][k]: the H = X W^T operand)
//   TRANS = true : image row f, column d  <-  g[d * cols + f]   (B[d][f] = W[d][f]: the dX = dH W operand)
// `ROWS` x `K` is the image extent (zero padded past the matrix), `g` is [grows][cols] row-major, `col0` the first
// column of `g` the image's column window (TRANS: row window) starts at (K-chunked operands).
// NT = threads of the workgroup, a compile-time constant so that the element loop unrolls completely and EVERY global
// load is issued before the first LDS write: as a run-time loop hipcc emits load -> s_waitcnt vmcnt(0) -> 3 x
// ds_write_b16 per element, i.e. 8 (two weights: 16) serialised memory round trips in the prologue of every workgroup.
template <bool TRANS, int NT, int ROWS, int K>
__device__ __forceinline__ void stage_weight_split(short* wl, const float* __restrict__ g, int grows, int cols, int col0 = 0) {
  constexpr int ld = K + WPAD, plane = ROWS * ld;
  constexpr int total = ROWS * K;            // TRANS: idx = d * ROWS + f (d < K = DD rows of g); else idx = n * K + k
  static_assert(total % NT == 0, "image elements per thread");
  constexpr int PER = total / NT;
  float v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (TRANS) {
      const int d = idx / ROWS, f = idx - d * ROWS;
      v[j] = g[(d < grows ? d : grows - 1) * cols + (col0 + f < cols ? col0 + f : cols - 1)];   // (a weight: 32-bit offsets)
    } else {
      const int n = idx / K, k = idx - n * K;
      v[j] = g[(n < grows ? n : grows - 1) * cols + (col0 + k < cols ? col0 + k : cols - 1)];
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = threadIdx.x + j * NT;
    int ir, ic;       // image row / column
    float x = v[j];
    if (TRANS) {
      const int d = idx / ROWS, f = idx - d * ROWS;
      if (col0 + f >= cols || d >= grows) x = 0.f;
      ir = f;
      ic = d;
    } else {
      const int n = idx / K, k = idx - n * K;
      if (col0 + k >= cols || n >= grows) x = 0.f;
      ir = n;
      ic = k;
    }
    const unsigned u1 = pk_bf16(x, 0.f) & 0xffffu;
    const float r1 = x - __uint_as_float(u1 << 16);
    const unsigned u2 = pk_bf16(r1, 0.f) & 0xffffu;
    const float r2 = r1 - __uint_as_float(u2 << 16);
    const unsigned u3 = pk_bf16(r2, 0.f) & 0xffffu;
    wl[ir * ld + ic] = (short)u1;
    wl[plane + ir * ld + ic] = (short)u2;
    wl[2 * plane + ir * ld + ic] = (short)u3;
  }
}

// acc{0,1}[TM x 64] += buf[TM x K] * (weight image: 64 rows n, K columns k)
template <int K>
__device__ __forceinline__ void tile_gemm_split(const float* buf, const short* wl, f32x16& acc0, f32x16& acc1, int lane) {
  const int r = lane & 31, h = lane >> 5;
  constexpr int ld = K + WPAD, plane = DD * ld;
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    const float4 a0 = *reinterpret_cast<const float4*>(buf + r * HS + 16 * s + 8 * h);
    const float4 a1 = *reinterpret_cast<const float4*>(buf + r * HS + 16 * s + 8 * h + 4);
    const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const Split3 A = split3(xa);
    const short* w0 = wl + r * ld + 16 * s + 8 * h;
    const short* w1 = w0 + 32 * ld;
    mfma_split(acc0, A, *reinterpret_cast<const bf16x8*>(w0), *reinterpret_cast<const bf16x8*>(w0 + plane),
               *reinterpret_cast<const bf16x8*>(w0 + 2 * plane));
    mfma_split(acc1, A, *reinterpret_cast<const bf16x8*>(w1), *reinterpret_cast<const bf16x8*>(w1 + plane),
               *reinterpret_cast<const bf16x8*>(w1 + 2 * plane));
  }
}

// The same product TRANSPOSED: the weight image is the MFMA's A operand, the tile rows its B operand, so that
// acc{0,1} = H^T blocks -- lane (r, h) holds ROW r of the tile, registers 4g .. 4g+3 its columns 8g + 4h + 0..3 (+ 32 for
// acc1): four consecutive columns of one row per register quad.  A row-major write-back is then 4 x ds_write_b128 per
// accumulator (lane-contiguous, conflict free with the HS row stride) instead of 16 x ds_write_b32, and a per-row scale is
// ONE value per lane.  Same six cross terms in the same order as tile_gemm_split.
template <int K>
__device__ __forceinline__ void tile_gemm_split_t(const float* buf, const short* wl, f32x16& acc0, f32x16& acc1, int lane) {
  const int r = lane & 31, h = lane >> 5;
  constexpr int ld = K + WPAD, plane = DD * ld;
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    const float4 a0 = *reinterpret_cast<const float4*>(buf + r * HS + 16 * s + 8 * h);
    const float4 a1 = *reinterpret_cast<const float4*>(buf + r * HS + 16 * s + 8 * h + 4);
    const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const Split3 X = split3(xa);
    const short* w0 = wl + r * ld + 16 * s + 8 * h;
    const short* w1 = w0 + 32 * ld;
#define HCG_SPLIT_T(ACC, W)                                                        \
    do {                                                                           \
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(W);                       \
      const bf16x8 b2 = *reinterpret_cast<const bf16x8*>((W) + plane);             \
      const bf16x8 b3 = *reinterpret_cast<const bf16x8*>((W) + 2 * plane);         \
      ACC = HCG_MFMA(b1, X.p3, ACC);                                               \
      ACC = HCG_MFMA(b3, X.p1, ACC);                                               \
      ACC = HCG_MFMA(b2, X.p2, ACC);                                               \
      ACC = HCG_MFMA(b1, X.p2, ACC);                                               \
      ACC = HCG_MFMA(b2, X.p1, ACC);                                               \
      ACC = HCG_MFMA(b1, X.p1, ACC);                                               \
    } while (0)
    HCG_SPLIT_T(acc0, w0);
    HCG_SPLIT_T(acc1, w1);
#undef HCG_SPLIT_T
  }
}

}  // namespace
