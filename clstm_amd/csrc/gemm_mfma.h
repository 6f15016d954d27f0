// gemm_mfma.h -- fp32 MFMA GEMM for the hoisted (non-recurrent) gate / softmax / dW products.
//
//   out(r, c) = sum_{k in slab} A(r, k) * B(k, c)      r in [0,R), c in [0,Cn), k in [0,K)
//
// 64x64 output tile per 256-thread workgroup (4 waves as 2x2, each 32x32 = 2x2 MFMA tiles of
// v_mfma_f32_16x16x4_f32), BK = 16 staged through LDS as [k][mn] with leading dimension 80
// (80 mod 32 = 16: the two k rows a half-wave reads for one fragment fall on disjoint banks, and
// rows stay 16-byte aligned for ds_write_b128).  The f32 MFMA is bit-for-bit an fmaf chain in k
// order (guide: cdna_hip_programming.md §3), so the result is the reference's `contract()` up to
// summation order.
//
// Every operand is an affine array and says which index is contiguous in memory:
//   KC: A(r,k) = A[r*ld + k]   (frame-major activations as the left operand, transposed weights)
//   MC: A(r,k) = A[k*ld + r]   (frame-major arrays contracted over frames: weight gradients)
// Staging is one 16-byte buffer load per thread per operand per k-tile (dword alignment is enough
// for global dwordx4), unconditional -- no exec-mask branches: rows / columns past R / Cn read
// neighbouring finite data whose products land in outputs that are never stored, accesses past the
// array end are dropped by the descriptor's bounds check, and the k tail is masked by four selects.
// MC tiles go to LDS with one ds_write_b128, KC tiles are transposed by four ds_write_b32.  The
// global loads of the next GEMM_PF tiles fly under the MFMAs.
// blockIdx.z = batch * nsplit + split-K slab; operand pointers advance by `bstride` floats per batch
// entry (the two directions of a BiLSTM layer share one launch); the epilogue functor receives
// blockIdx.z (deterministic slab reduction in ops.h:k_reduce_scatter).
#pragma once
#include "devintrin.h"

namespace clstm {

enum { GEMM_KC = 0, GEMM_MC = 1 };
constexpr int GEMM_BT = 64;   // tile rows / cols
constexpr int GEMM_BK = 16;
constexpr int GEMM_LD = 80;
constexpr int GEMM_PF = 3;   // k-tiles prefetched in registers
constexpr int GEMM_LDO = 68;  // LDS row stride of the output tile in the epilogue

struct GemmOperand {
  const float* p;
  int ld;
  long long elems;  // floats readable from p (array extent; may include up to 3 floats of slack)
  long long bstride;  // floats between batch entries (0: shared by all batch entries)
};

// one workgroup of the GEMM: `lin` is its linear index in a (gx, gy, gz) grid
// BK: contraction indices staged per barrier pair (16 or 32).  32 halves the barriers and LDS hand-overs per flop --
// it pays where the k loop is long (the split-K weight-gradient GEMMs: K = frames of a slab) and costs registers
// (two staging quadruples per operand and tile in flight) and LDS (2 x 32 x 80 floats).
template <int AMODE, int BMODE, class FE, int BK = GEMM_BK, int PF = GEMM_PF>
DEVFN void gemm_f32_body(float* smem, GemmOperand A, GemmOperand B, FE fe, int R, int Cn, int K, int ksplit, int nsplit,
                         const unsigned lin, const unsigned gx, const unsigned gy, const unsigned gz) {
  constexpr int NP = BK / 16;   // staging passes of 16 contraction indices
  // smem: operand tiles during the k loop; the 64 x 64 output tile (row stride 68) during the epilogue
  float* As = smem;
  float* Bs = smem + BK * GEMM_LD;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (guide T1): the dispatcher places workgroup b on XCD b % 8, each XCD has its
  // own L2.  Remapped so that an XCD gets a CONTIGUOUS run of tiles in (x fastest, y, z) order: the
  // tiles sharing one split-K slab / one row panel then pull their operands through one L2 instead of
  // eight (measured on the weight-gradient GEMM: 250 MB -> see profiles/ of HBM reads per launch).
  // Bijective for any grid size: XCD x owns q + (x < r) tiles.
  int bx, by, z;
  {
    const unsigned total = gx * gy * gz;
    const unsigned xcd = lin & 7u, idx = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = (int)(v % gx);
    by = (int)((v / gx) % gy);
    z = (int)(v / (gx * gy));
  }
  const int r0 = by * GEMM_BT, c0 = bx * GEMM_BT;
  const int batch = z / nsplit;
  const int kbeg = (z - batch * nsplit) * ksplit;
  const int kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;

  // staging coordinates of this thread's float4:
  //   KC: 4 consecutive k of row  (tid>>2)      -> transposed into LDS
  //   MC: 4 consecutive mn of k-row (tid>>4)    -> one ds_write_b128
  const int a_mn = AMODE == GEMM_KC ? (tid >> 2) : (tid & 15) * 4;
  const int a_k = AMODE == GEMM_KC ? (tid & 3) * 4 : (tid >> 4);
  const int b_mn = BMODE == GEMM_KC ? (tid >> 2) : (tid & 15) * 4;
  const int b_k = BMODE == GEMM_KC ? (tid & 3) * 4 : (tid >> 4);
  const BufF32 abuf = make_buf(A.p + batch * A.bstride, (size_t)(A.elems - batch * A.bstride) * 4);
  const BufF32 bbuf = make_buf(B.p + batch * B.bstride, (size_t)(B.elems - batch * B.bstride) * 4);
  const unsigned a_base = AMODE == GEMM_KC ? (unsigned)(r0 + a_mn) * A.ld + a_k : (unsigned)a_k * A.ld + r0 + a_mn;
  const unsigned b_base = BMODE == GEMM_KC ? (unsigned)(c0 + b_mn) * B.ld + b_k : (unsigned)b_k * B.ld + c0 + b_mn;
  const unsigned a_kstep = AMODE == GEMM_KC ? 1u : (unsigned)A.ld, b_kstep = BMODE == GEMM_KC ? 1u : (unsigned)B.ld;

  // Loads are issued unconditionally (tiles past the slab get an out-of-range offset, which the
  // descriptor drops without touching memory): with a fixed number of loads per phase the compiler
  // can count vmcnt exactly instead of draining the queue at every control-flow join.
  auto load_tile = [&](int k0, f32x4 (&ra)[NP], f32x4 (&rb)[NP]) {
#pragma unroll
    for (int h = 0; h < NP; h++) {
      const bool live = k0 + 16 * h < kend;
      ra[h] = buf_load4(abuf, live ? (a_base + (unsigned)(k0 + 16 * h) * a_kstep) * 4u : BUF_OOB);
      rb[h] = buf_load4(bbuf, live ? (b_base + (unsigned)(k0 + 16 * h) * b_kstep) * 4u : BUF_OOB);
    }
  };
  // frames / contraction indices past the slab hold real data (next slab, next row): zero them.  Done
  // when the tile is staged, not when it is loaded -- touching the registers earlier would put a
  // vmcnt wait right behind the load and serialise the prefetch.
  auto mask_tile = [&](int k0, f32x4 (&ra)[NP], f32x4 (&rb)[NP]) {
#pragma unroll
    for (int h = 0; h < NP; h++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        ra[h][i] = (k0 + 16 * h + a_k + (AMODE == GEMM_KC ? i : 0) < kend) ? ra[h][i] : 0.0f;
        rb[h][i] = (k0 + 16 * h + b_k + (BMODE == GEMM_KC ? i : 0) < kend) ? rb[h][i] : 0.0f;
      }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;

  // GEMM_PF k-tiles are in flight in registers: a tile's global loads are issued GEMM_PF iterations
  // before it is staged, which covers the ~2000-cycle HBM latency with MFMA work of the same workgroup
  // (the grids here are only 1-3 workgroups per CU, so there is little inter-workgroup overlap to lean on)
  f32x4 ra[PF][NP], rb[PF][NP];
#pragma unroll
  for (int p = 0; p < PF; p++) {
    load_tile(kbeg + p * BK, ra[p], rb[p]);
    SCHED_FENCE();   // same issue order as inside the loop, so the vmcnt at the loop head stays exact
  }
  const int fk = lane >> 4, fi = lane & 15;
  for (int kb = kbeg; kb < kend; kb += PF * BK) {
#pragma unroll
    for (int p = 0; p < PF; p++) {
      const int k0 = kb + p * BK;   // phases past the slab multiply zeros (no early exit: the
                                    // straight-line body keeps the accumulators and vmcnt exact)
      mask_tile(k0, ra[p], rb[p]);
#pragma unroll
      for (int h = 0; h < NP; h++) {
        if (AMODE == GEMM_KC) {
#pragma unroll
          for (int i = 0; i < 4; i++) As[(16 * h + a_k + i) * GEMM_LD + a_mn] = ra[p][h][i];
        } else {
          *reinterpret_cast<f32x4*>(&As[(16 * h + a_k) * GEMM_LD + a_mn]) = ra[p][h];
        }
        if (BMODE == GEMM_KC) {
#pragma unroll
          for (int i = 0; i < 4; i++) Bs[(16 * h + b_k + i) * GEMM_LD + b_mn] = rb[p][h][i];
        } else {
          *reinterpret_cast<f32x4*>(&Bs[(16 * h + b_k) * GEMM_LD + b_mn]) = rb[p][h];
        }
      }
      __syncthreads();
      if (PF > 1 || K > BK) load_tile(k0 + PF * BK, ra[p], rb[p]);
      SCHED_FENCE();
#pragma unroll
      for (int kk = 0; kk < BK; kk += 4) {
        if (PF == 1 && kk > 0 && wave_uniform(k0 + kk >= kend ? 1 : 0)) break;   // single-pass form: stop at the last k-step
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
  }
  if (fe.vec4()) {
    // Epilogue through LDS: the MFMA layout holds 16-float pieces of a row per lane group; transposed
    // through the tile every lane writes one 16-byte piece and a wave covers four whole 256-byte row
    // segments per store instruction (16 partial-line stores per lane otherwise).
    // (the last k phase ended with a barrier: the operand tiles are dead)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          smem[(wm * 32 + i * 16 + (lane >> 4) * 4 + q) * GEMM_LDO + wn * 32 + j * 16 + (lane & 15)] = acc[i][j][q];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int rl = it * 16 + (tid >> 4), cl = (tid & 15) * 4;
      const int r = r0 + rl, c = c0 + cl;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&smem[rl * GEMM_LDO + cl]);
      if (r < R) {
        if (c + 3 < Cn) fe.row4(r, c, v, z);
        else
          for (int e = 0; e < 4; e++)
            if (c + e < Cn) fe(r, c + e, v[e], z);
      }
    }
    return;
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

constexpr int gemm_smem_floats(int bk) { return 2 * bk * GEMM_LD > GEMM_BT * GEMM_LDO ? 2 * bk * GEMM_LD : GEMM_BT * GEMM_LDO; }
template <int AMODE, int BMODE, class FE, int BK = GEMM_BK, int PF = GEMM_PF>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmOperand A, GemmOperand B, FE fe, int R, int Cn,
                                                       int K, int ksplit, int nsplit) {
  __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats(BK)];
  gemm_f32_body<AMODE, BMODE, FE, BK, PF>(smem, A, B, fe, R, Cn, K, ksplit, nsplit,
                                  blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y,
                                  gridDim.z);
}
// Two INDEPENDENT small GEMMs in one launch (1-D grid: the first nb1 workgroups run the first problem): each is
// far from filling the chip and mostly prologue / epilogue latency, so their workgroups interleave on the CUs
// instead of running back to back (the softmax layer's x.d and W.d products, both reading z.d).
struct GemmProblem {
  GemmOperand A, B;
  int R, Cn, K, ksplit, nsplit;
  unsigned gx, gy, gz;
};
template <int A1, int B1, class FE1, int A2, int B2, class FE2>
__global__ __launch_bounds__(256) void gemm_f32_pair_kernel(GemmProblem p1, FE1 fe1, GemmProblem p2, FE2 fe2, unsigned nb1) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_BT * GEMM_LDO];
  if (blockIdx.x < nb1)
    gemm_f32_body<A1, B1, FE1>(smem, p1.A, p1.B, fe1, p1.R, p1.Cn, p1.K, p1.ksplit, p1.nsplit, blockIdx.x, p1.gx, p1.gy, p1.gz);
  else
    gemm_f32_body<A2, B2, FE2>(smem, p2.A, p2.B, fe2, p2.R, p2.Cn, p2.K, p2.ksplit, p2.nsplit, blockIdx.x - nb1, p2.gx, p2.gy,
                               p2.gz);
}

// Operand constructors.  `slack` floats after the array may be read (and are multiplied into outputs
// that are never stored): library-owned buffers are over-allocated, so 3 is always safe for them.
inline GemmOperand gemm_kc(const float* p, int ld, long long rows, int slack = 3) {
  return GemmOperand{p, ld, rows * (long long)ld + slack, 0};
}
inline GemmOperand gemm_mc(const float* p, int ld, long long frames, int slack = 3) {
  return GemmOperand{p, ld, frames * (long long)ld + slack, 0};
}
// batch entry b reads from p + b*bstride; `elems` must cover the last entry
inline GemmOperand gemm_batched(GemmOperand o, long long bstride, int nbatch) {
  o.elems += bstride * (nbatch - 1);
  o.bstride = bstride;
  return o;
}

inline GemmProblem gemm_problem(GemmOperand A, GemmOperand B, int R, int Cn, int K, int nsplit = 1, int nbatch = 1) {
  if (nsplit < 1) nsplit = 1;
  int ksplit = (K + nsplit - 1) / nsplit;
  const int kq = nsplit > 1 ? GEMM_PF * GEMM_BK : GEMM_BK;
  ksplit = ((ksplit + kq - 1) / kq) * kq;
  if (ksplit < kq) ksplit = kq;
  return GemmProblem{A, B, R, Cn, K, ksplit, nsplit, (unsigned)((Cn + GEMM_BT - 1) / GEMM_BT),
                     (unsigned)((R + GEMM_BT - 1) / GEMM_BT), (unsigned)(nsplit * nbatch)};
}
template <int A1, int B1, class FE1, int A2, int B2, class FE2>
inline void gemm_f32_pair(hipStream_t stream, GemmProblem p1, FE1 fe1, GemmProblem p2, FE2 fe2) {
  const unsigned nb1 = p1.gx * p1.gy * p1.gz, nb2 = p2.gx * p2.gy * p2.gz;
  CLSTM_LAUNCH((gemm_f32_pair_kernel<A1, B1, FE1, A2, B2, FE2>), dim3(nb1 + nb2), dim3(256), 0, stream, p1, fe1, p2, fe2, nb1);
}
template <int AMODE, int BMODE, class FE, int BK = GEMM_BK, int PF = GEMM_PF>
inline void gemm_f32(hipStream_t stream, GemmOperand A, GemmOperand B, FE fe, int R, int Cn, int K, int nsplit = 1,
                     int nbatch = 1) {
  if (R <= 0 || Cn <= 0) return;
  if (nsplit < 1) nsplit = 1;
  int ksplit = (K + nsplit - 1) / nsplit;
  const int kq = nsplit > 1 ? PF * BK : BK;   // whole pipeline rounds per slab
  ksplit = ((ksplit + kq - 1) / kq) * kq;
  if (ksplit < kq) ksplit = kq;
  dim3 grid((Cn + GEMM_BT - 1) / GEMM_BT, (R + GEMM_BT - 1) / GEMM_BT, nsplit * nbatch);
  CLSTM_LAUNCH((gemm_f32_kernel<AMODE, BMODE, FE, BK, PF>), grid, dim3(256), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit);
}

}  // namespace clstm
