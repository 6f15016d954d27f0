// softmax_fused.h -- SoftmaxLayer::forward for up to 96 classes in ONE kernel
// (forward_softmax, clstm_compute.cc:324-345: z = limexp(W1[:,1:] . x + W1[:,0]); z /= colsum(z), no
// max-subtraction).
//
// The generic route was a 64x64-tile GEMM whose second column tile held 19 of 83 classes, followed by a
// normalisation pass over Z (17.6 + 8.2 us at the bench shape; this kernel: 22 us).  Here a wave owns 16
// frames x ALL classes (1 x 6 v_mfma_f32_16x16x4_f32 tiles), so a whole row of logits lives in one 16-lane
// group and the row sum is four DPP row rotations -- no second pass, no LDS exchange, H is read once.
// Operand staging, register prefetch and the unconditional-issue / exact-vmcnt scheme are those of
// gemm_mfma.h.
#pragma once
#include "gemm_mfma.h"

namespace clstm {

constexpr int SMX_PF = 3;   // k-tiles in flight in registers
constexpr int SMX_COLS = 96;   // classes per workgroup (6 MFMA column tiles)
constexpr int SMX_LDB = 112;   // LDS row stride of the weight tile (112 mod 32 = 16, see gemm_mfma.h)

DEVFN float smx_limexp(float x) {  // tensor.h:78-82
  if (x < -30.0f) return (float)0x1.a56e0c2b7ab97p-44;  // (Float)exp(-30.0)
  if (x > 30.0f) return (float)0x1.37047090c0b53p+43;   // (Float)exp(30.0)
  return expf(x);
}

// A: frames x K (k contiguous); W1: [nc][1+K] column-major (bias in column 0); Z: [N][nc]
// 4 waves x 16 frames per workgroup.  (A one-wave-per-workgroup variant -- 800 instead of 200 workgroups, no
// barriers -- was measured slower, 31 vs 22 us: every wave then stages the whole 16 x 96 weight tile itself.)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(GemmOperand A, const float* W1, long long w1_elems,
                                                         float* Z, int N, int nc, int K, int* nanflag, int step_no) {
  __shared__ __attribute__((aligned(16))) float As[GEMM_BK * GEMM_LD];
  __shared__ __attribute__((aligned(16))) float Bs[GEMM_BK * SMX_LDB];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int r0 = blockIdx.x * GEMM_BT;

  const int a_mn = tid >> 2, a_k = (tid & 3) * 4;            // A: 4 consecutive k of frame r0 + a_mn
  const BufF32 abuf = make_buf(A.p, (size_t)A.elems * 4);
  const BufF32 bbuf = make_buf(W1, (size_t)w1_elems * 4);
  const unsigned a_base = (unsigned)(r0 + a_mn) * A.ld + a_k;
  // B(k, c) = W1[nc * (1 + k) + c]: 24 float4 per k row, 384 per tile -> threads 0..255 and 0..127 again
  const int b_k0 = tid / 24, b_c0 = (tid % 24) * 4;
  const int b_k1 = (tid + 256) / 24, b_c1 = ((tid + 256) % 24) * 4;
  const bool b_second = tid + 256 < GEMM_BK * 24;

  f32x4 ra[SMX_PF], rb0[SMX_PF], rb1[SMX_PF];
  auto load_tile = [&](int k0, f32x4& a, f32x4& b0, f32x4& b1) {
    const bool live = k0 < K;
    a = buf_load4(abuf, live ? (a_base + (unsigned)k0) * 4u : BUF_OOB);
    b0 = buf_load4(bbuf, live ? ((unsigned)nc * (1 + k0 + b_k0) + b_c0) * 4u : BUF_OOB);
    b1 = buf_load4(bbuf, live && b_second ? ((unsigned)nc * (1 + k0 + b_k1) + b_c1) * 4u : BUF_OOB);
  };
  f32x4 acc[6];
  float bias[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    bias[j] = buf_load(bbuf, j * 16 + (lane & 15) < nc ? (unsigned)(j * 16 + (lane & 15)) * 4u : BUF_OOB);   // W1[:, 0]
#pragma unroll
    for (int q = 0; q < 4; q++) acc[j][q] = 0.0f;
  }
#pragma unroll
  for (int p = 0; p < SMX_PF; p++) {
    load_tile(p * GEMM_BK, ra[p], rb0[p], rb1[p]);
    SCHED_FENCE();
  }
  const int fk = lane >> 4, fi = lane & 15;
  for (int kb = 0; kb < K; kb += SMX_PF * GEMM_BK) {
#pragma unroll
    for (int p = 0; p < SMX_PF; p++) {
      const int k0 = kb + p * GEMM_BK;   // phases past K multiply zeros
#pragma unroll
      for (int i = 0; i < 4; i++) As[(a_k + i) * GEMM_LD + a_mn] = (k0 + a_k + i < K) ? ra[p][i] : 0.0f;
      {
        const bool l0 = k0 + b_k0 < K, l1 = k0 + b_k1 < K;
        f32x4 v0 = rb0[p], v1 = rb1[p];
#pragma unroll
        for (int i = 0; i < 4; i++) { v0[i] = l0 ? v0[i] : 0.0f; v1[i] = l1 ? v1[i] : 0.0f; }
        *reinterpret_cast<f32x4*>(&Bs[b_k0 * SMX_LDB + b_c0]) = v0;
        if (b_second) *reinterpret_cast<f32x4*>(&Bs[b_k1 * SMX_LDB + b_c1]) = v1;
      }
      __syncthreads();
      load_tile(k0 + SMX_PF * GEMM_BK, ra[p], rb0[p], rb1[p]);
      SCHED_FENCE();
#pragma unroll
      for (int kk = 0; kk < GEMM_BK; kk += 4) {
        const float af = As[(kk + fk) * GEMM_LD + wave * 16 + fi];
#pragma unroll
        for (int j = 0; j < 6; j++) acc[j] = mfma16x16x4(af, Bs[(kk + fk) * SMX_LDB + j * 16 + fi], acc[j]);
      }
      __syncthreads();
    }
  }
  // epilogue: lane holds rows (lane>>4)*4 + q and columns j*16 + (lane&15); a row's 96 columns sit in the
  // 16 lanes of one row group.  (The six biases of a lane were requested before the k loop: read inside this loop
  // they were 24 dependent L1 round trips -- the epilogue took 6.8 of the kernel's 21 us.)
  // Branch-free: clamp + selects for limexp's two cut-offs, out-of-range buffer offsets for the masked stores (with
  // `if`s hipcc built an exec-masked block per element: 96 s_and_saveexec in this epilogue).
  const BufF32 zbuf = make_buf(Z, (size_t)N * nc * 4);
  bool cok[6];
#pragma unroll
  for (int j = 0; j < 6; j++) cok[j] = j * 16 + (lane & 15) < nc;
  bool nonfinite = false;   // limexp's clamp would swallow a NaN logit: looked at before it (k_update, ops.h)
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int r = r0 + wave * 16 + (lane >> 4) * 4 + q;
    float e[6];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const float x = acc[j][q] + bias[j];
      nonfinite |= r < N && cok[j] && !f32_finite(x);
      float v = expf(fminf(fmaxf(x, -30.0f), 30.0f));
      v = x < -30.0f ? (float)0x1.a56e0c2b7ab97p-44 : v;   // (Float)exp(-30.0), tensor.h:78-82
      v = x > 30.0f ? (float)0x1.37047090c0b53p+43 : v;    // (Float)exp(30.0)
      e[j] = cok[j] ? v : 0.0f;
      s += e[j];
    }
    s += row_ror<8>(s);
    s += row_ror<4>(s);
    s += row_ror<2>(s);
    s += row_ror<1>(s);
    const unsigned rowoff = (unsigned)r * (unsigned)nc;
#pragma unroll
    for (int j = 0; j < 6; j++)
      buf_store(zbuf, r < N && cok[j] ? (rowoff + (unsigned)(j * 16 + (lane & 15))) * 4u : BUF_OOB, e[j] / s);
  }
  if (nanflag && nonfinite) raise_nonfinite(nanflag, step_no);
}

inline void softmax_fwd(hipStream_t stream, GemmOperand A, const float* W1, long long w1_elems, float* Z, int N, int nc,
                        int K, int* nanflag = nullptr, int step_no = 0) {
  if (N <= 0) return;
  CLSTM_LAUNCH(softmax_fwd_kernel, dim3((N + GEMM_BT - 1) / GEMM_BT), dim3(256), 0, stream, A, W1, w1_elems, Z, N, nc, K, nanflag, step_no);
}

}  // namespace clstm
