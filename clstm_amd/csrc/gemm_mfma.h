// gemm_mfma.h -- fp32 MFMA GEMM for the hoisted (non-recurrent) gate / softmax / dW products.
//
//   out(r, c) = sum_{k in split} A(r, k) * B(k, c)      r in [0,R), c in [0,Cn), k in [0,K)
//
// 64x64 output tile per 256-thread workgroup (4 waves as 2x2, each 32x32 = 2x2 MFMA tiles of
// v_mfma_f32_16x16x4_f32), BK = 16 staged through LDS as [k][mn] with leading dimension 81
// (odd, so both the k-contiguous and the mn-contiguous staging patterns spread over banks and
// the fragment reads of two adjacent k rows overlap on one bank only).  Operands are described
// by functors so that the virtual operands of the reference's ops -- [1 ; x_t ; h_{t-1}]
// (forward_stack_delay, clstm_compute.cc:377-397), time-shifted h, gate-interleaved row
// scatter into the flat Params layout -- need no materialised copies.
// The f32 MFMA is bit-for-bit an fmaf chain in k order (guide: cdna_hip_programming.md §3), so
// the result is the reference's `contract()` up to summation order.
//
// Staging mode per operand says which index is contiguous in memory:
//   KC = contiguous along the contraction index, MC = contiguous along the output index.
// blockIdx.z = split-K slice; the epilogue functor receives it.
#pragma once
#include "devintrin.h"

namespace clstm {

enum { GEMM_KC = 0, GEMM_MC = 1 };
constexpr int GEMM_BT = 64;   // tile rows / cols
constexpr int GEMM_BK = 16;
constexpr int GEMM_LD = 81;

template <int AMODE, int BMODE, class FA, class FB, class FE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(FA fa, FB fb, FE fe, int R, int Cn, int K,
                                                       int ksplit) {
  __shared__ float As[GEMM_BK * GEMM_LD];
  __shared__ float Bs[GEMM_BK * GEMM_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r0 = blockIdx.y * GEMM_BT, c0 = blockIdx.x * GEMM_BT;
  const int z = blockIdx.z;
  const int kbeg = z * ksplit;
  const int kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;

  // staging coordinates: 4 elements per thread per operand per k-tile
  int a_k[4], a_m[4], b_k[4], b_n[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (AMODE == GEMM_KC) { a_k[i] = tid & 15; a_m[i] = (tid >> 4) + 16 * i; }
    else                  { a_m[i] = tid & 63; a_k[i] = (tid >> 6) + 4 * i; }
    if (BMODE == GEMM_KC) { b_k[i] = tid & 15; b_n[i] = (tid >> 4) + 16 * i; }
    else                  { b_n[i] = tid & 63; b_k[i] = (tid >> 6) + 4 * i; }
  }
  float ra[4], rb[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + a_m[i], ka = k0 + a_k[i];
      ra[i] = (r < R && ka < kend) ? fa(r, ka) : 0.0f;
      const int c = c0 + b_n[i], kb = k0 + b_k[i];
      rb[i] = (c < Cn && kb < kend) ? fb(kb, c) : 0.0f;
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;

  const int fk = lane >> 4, fi = lane & 15;
  if (kbeg < kend) load_tile(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      As[a_k[i] * GEMM_LD + a_m[i]] = ra[i];
      Bs[b_k[i] * GEMM_LD + b_n[i]] = rb[i];
    }
    __syncthreads();
    if (k0 + GEMM_BK < kend) load_tile(k0 + GEMM_BK);  // global loads fly under the MFMAs
#pragma unroll
    for (int kk = 0; kk < GEMM_BK; kk += 4) {
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        af[i] = As[(kk + fk) * GEMM_LD + wm * 32 + i * 16 + fi];
        bf[i] = Bs[(kk + fk) * GEMM_LD + wn * 32 + i * 16 + fi];
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma16x16x4(af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = r0 + wm * 32 + i * 16 + (lane >> 4) * 4 + q;
        const int c = c0 + wn * 32 + j * 16 + (lane & 15);
        if (r < R && c < Cn) fe(r, c, acc[i][j][q], z);
      }
}

template <int AMODE, int BMODE, class FA, class FB, class FE>
inline void gemm_f32(hipStream_t stream, FA fa, FB fb, FE fe, int R, int Cn, int K, int nsplit = 1) {
  if (R <= 0 || Cn <= 0) return;
  if (nsplit < 1) nsplit = 1;
  int ksplit = (K + nsplit - 1) / nsplit;
  ksplit = ((ksplit + GEMM_BK - 1) / GEMM_BK) * GEMM_BK;
  if (ksplit < GEMM_BK) ksplit = GEMM_BK;
  dim3 grid((Cn + GEMM_BT - 1) / GEMM_BT, (R + GEMM_BT - 1) / GEMM_BT, nsplit);
  CLSTM_LAUNCH((gemm_f32_kernel<AMODE, BMODE, FA, FB, FE>), grid, dim3(256), 0, stream, fa, fb, fe,
               R, Cn, K, ksplit);
}

// ---------------------------------------------------------------------------------------------------
// Weight-gradient form: out(r, c) = sum_{k in split} A[k*lda + r] * B[k*ldb + c] with a LONG
// contraction (k = every frame of the minibatch) and a small output (a Params matrix).  Both operands
// are frame-major arrays, contiguous along their output index -- exactly the f32 MFMA fragment order
// (16 consecutive rows / columns for 4 consecutive k) -- so every lane loads its fragments straight
// from global memory (L1/L2-served 64-byte segments): no LDS, no barriers.  Loads are unconditional
// buffer loads: the descriptor ends at the slab's last frame, so k >= kend reads 0; rows / columns
// past R / Cn read finite neighbouring data whose products land in outputs that are never stored.
// The next k-step's fragments are fetched into a second register set while the current 20 MFMAs
// issue (loop unrolled by two, no register moves).
// Workgroup = 2x2 waves, wave tile = 5x4 MFMA tiles (80x64), workgroup tile 160 x 128.
// blockIdx.z = split-K slab; slabs are reduced deterministically by k_reduce_scatter.
constexpr int GEMM_TN_TR = 5, GEMM_TN_TC = 4;
constexpr int GEMM_TN_ROWS = 2 * GEMM_TN_TR * 16, GEMM_TN_COLS = 2 * GEMM_TN_TC * 16;

template <class FE>
__global__ __launch_bounds__(256) void gemm_tn_direct_kernel(const float* A, int lda, const float* B, int ldb,
                                                             FE fe, int R, int Cn, int K, int ksplit) {
  constexpr int TR = GEMM_TN_TR, TC = GEMM_TN_TC;
  const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int r0 = blockIdx.y * GEMM_TN_ROWS + wr * TR * 16;
  const int c0 = blockIdx.x * GEMM_TN_COLS + wc * TC * 16;
  const int z = blockIdx.z;
  const int kbeg = z * ksplit;
  const int kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;
  const int fi = lane & 15, fk = lane >> 4;
  // descriptors start at the slab's first frame and end at its last one
  const BufF32 abuf = make_buf(A + (size_t)kbeg * lda, kend > kbeg ? (size_t)(kend - kbeg) * lda * 4 : 0);
  const BufF32 bbuf = make_buf(B + (size_t)kbeg * ldb, kend > kbeg ? (size_t)(kend - kbeg) * ldb * 4 : 0);
  const unsigned aoff = ((unsigned)fk * lda + r0 + fi) * 4u, boff = ((unsigned)fk * ldb + c0 + fi) * 4u;
  const unsigned astep = 4u * lda * 4u, bstep = 4u * ldb * 4u;  // 4 frames per k-step

  f32x4 acc[TR][TC];
#pragma unroll
  for (int i = 0; i < TR; i++)
#pragma unroll
    for (int j = 0; j < TC; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;

  float aA[TR], bA[TC], aB[TR], bB[TC];
  auto load = [&](unsigned step, float (&av)[TR], float (&bv)[TC]) {
#pragma unroll
    for (int i = 0; i < TR; i++) av[i] = buf_load(abuf, aoff + step * astep + 64u * i);
#pragma unroll
    for (int j = 0; j < TC; j++) bv[j] = buf_load(bbuf, boff + step * bstep + 64u * j);
  };
  auto mma = [&](const float (&av)[TR], const float (&bv)[TC]) {
#pragma unroll
    for (int i = 0; i < TR; i++)
#pragma unroll
      for (int j = 0; j < TC; j++) acc[i][j] = mfma16x16x4(av[i], bv[j], acc[i][j]);
  };
  const unsigned nsteps = kend > kbeg ? (unsigned)(kend - kbeg + 3) / 4 : 0;
  load(0, aA, bA);
  unsigned st = 0;
  for (; st + 1 < nsteps; st += 2) {
    load(st + 1, aB, bB);
    mma(aA, bA);
    load(st + 2, aA, bA);  // past the slab: reads 0
    mma(aB, bB);
  }
  if (st < nsteps) mma(aA, bA);
#pragma unroll
  for (int i = 0; i < TR; i++)
#pragma unroll
    for (int j = 0; j < TC; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = r0 + i * 16 + (lane >> 4) * 4 + q;
        const int c = c0 + j * 16 + (lane & 15);
        if (r < R && c < Cn) fe(r, c, acc[i][j][q], z);
      }
}

// A: [K][lda] (rows r < R used), B: [K][ldb] (columns c < Cn used).  The arrays must extend at least
// GEMM_TN_ROWS / GEMM_TN_COLS floats past the last used row start only in the sense of "readable or
// past the end of the slab": reads beyond the slab return 0, reads inside it hit finite data.
template <class FE>
inline void gemm_tn_direct(hipStream_t stream, const float* A, int lda, const float* B, int ldb, FE fe, int R,
                           int Cn, int K, int nsplit) {
  if (R <= 0 || Cn <= 0) return;
  if (nsplit < 1) nsplit = 1;
  int ksplit = (K + nsplit - 1) / nsplit;
  ksplit = ((ksplit + 7) / 8) * 8;
  dim3 grid((Cn + GEMM_TN_COLS - 1) / GEMM_TN_COLS, (R + GEMM_TN_ROWS - 1) / GEMM_TN_ROWS, nsplit);
  CLSTM_LAUNCH((gemm_tn_direct_kernel<FE>), grid, dim3(256), 0, stream, A, lda, B, ldb, fe, R, Cn, K, ksplit);
}

}  // namespace clstm
