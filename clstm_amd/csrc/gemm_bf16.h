// gemm_bf16.h -- the hoisted gate GEMMs with bf16 inputs and f32 accumulation
// (v_mfma_f32_16x16x32_bf16; BASELINE config "2 x BiLSTM(512), bf16 MFMA").  Opt-in
// (clstm_net_set_gemm_precision): the default path is the exact-f32 kernel of gemm_mfma.h.
//
// Same interface and tile decomposition as gemm_f32_kernel (affine KC / MC operands in f32 memory,
// 64x64 output tile, 2x2 waves x 2x2 MFMA tiles, split-K / batch in blockIdx.z, XCD-aware tile order,
// register prefetch with exact vmcnt), different inner product:
//   * BK = 32; operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) WHILE they are staged, and live in
//     LDS as [mn][k] with k contiguous (row stride 40 halfs = 80 B: a fragment is one conflict-free
//     ds_read_b128 of 8 bf16);
//   * KC operands (k contiguous in memory): a thread converts 8 consecutive k of one row -> one
//     ds_write_b128;  MC operands (mn contiguous): a thread loads 4 mn at k and at k+1 and writes four
//     packed (k, k+1) pairs -> ds_write_b32, which is the transpose;
//   * lane (i, kb) feeds the MFMA with k = 8 kb .. 8 kb + 7 of row i for BOTH operands; the instruction's
//     internal k order is irrelevant because A and B use the same slot assignment.
// One 32-k block costs a wave 4 ds_read_b128 + 4 MFMAs (~70 cycles) instead of 16 ds_read_b32 + 32 f32
// MFMAs (1024 cycles); the kernel is then bound by the global->LDS staging of its 64x64 tiles.
#pragma once
#include <stdexcept>
#include "dbgopt.h"
#include "gemm_mfma.h"

namespace clstm {

constexpr int GB_BK = 32;
constexpr int GB_LDH = 40;   // halfs per LDS row
constexpr int GB_PF = 3;     // k-tiles prefetched in registers

template <int AMODE, int BMODE, class FE>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmOperand A, GemmOperand B, FE fe, int R, int Cn,
                                                        int K, int ksplit, int nsplit) {
  __shared__ __attribute__((aligned(16))) unsigned short As[GEMM_BT * GB_LDH];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[GEMM_BT * GB_LDH];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by, z;   // XCD-aware tile order, see gemm_mfma.h
  {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
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

  // staging unit of this thread (two float4 per operand per k-tile):
  //   KC: row tid>>2, k = (tid&3)*8 .. +7          (second float4 = +4 floats)
  //   MC: mn = (tid&15)*4 .. +3, k = (tid>>4)*2, +1 (second float4 = +ld floats)
  const int a_mn = AMODE == GEMM_KC ? (tid >> 2) : (tid & 15) * 4;
  const int a_k = AMODE == GEMM_KC ? (tid & 3) * 8 : (tid >> 4) * 2;
  const int b_mn = BMODE == GEMM_KC ? (tid >> 2) : (tid & 15) * 4;
  const int b_k = BMODE == GEMM_KC ? (tid & 3) * 8 : (tid >> 4) * 2;
  const BufF32 abuf = make_buf(A.p + batch * A.bstride, (size_t)(A.elems - batch * A.bstride) * 4);
  const BufF32 bbuf = make_buf(B.p + batch * B.bstride, (size_t)(B.elems - batch * B.bstride) * 4);
  const unsigned a_base = AMODE == GEMM_KC ? (unsigned)(r0 + a_mn) * A.ld + a_k : (unsigned)a_k * A.ld + r0 + a_mn;
  const unsigned b_base = BMODE == GEMM_KC ? (unsigned)(c0 + b_mn) * B.ld + b_k : (unsigned)b_k * B.ld + c0 + b_mn;
  const unsigned a_kstep = AMODE == GEMM_KC ? 1u : (unsigned)A.ld, b_kstep = BMODE == GEMM_KC ? 1u : (unsigned)B.ld;
  const unsigned a_second = AMODE == GEMM_KC ? 4u : (unsigned)A.ld, b_second = BMODE == GEMM_KC ? 4u : (unsigned)B.ld;

  f32x4 ra[GB_PF][2], rb[GB_PF][2];
  auto load_tile = [&](int k0, f32x4 (&a)[2], f32x4 (&b)[2]) {   // unconditional issue, see gemm_mfma.h
    const bool live = k0 < kend;
    const unsigned ao = (a_base + (unsigned)k0 * a_kstep) * 4u, bo = (b_base + (unsigned)k0 * b_kstep) * 4u;
    a[0] = buf_load4(abuf, live ? ao : BUF_OOB);
    a[1] = buf_load4(abuf, live ? ao + a_second * 4u : BUF_OOB);
    b[0] = buf_load4(bbuf, live ? bo : BUF_OOB);
    b[1] = buf_load4(bbuf, live ? bo + b_second * 4u : BUF_OOB);
  };
  // round to bf16 and store k-contiguous; contraction indices past the slab are zeroed here
  auto stage = [&](const int MODE, unsigned short* S, const int mn, const int kk, const int k0, const f32x4 (&r)[2]) {
    if (MODE == GEMM_KC) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = (k0 + kk + i < kend) ? r[i >> 2][i & 3] : 0.0f;
      *reinterpret_cast<u16x8*>(&S[mn * GB_LDH + kk]) = bf16_pack8(x);
    } else {
      const bool l0 = k0 + kk < kend, l1 = k0 + kk + 1 < kend;
#pragma unroll
      for (int i = 0; i < 4; i++)
        *reinterpret_cast<unsigned*>(&S[(mn + i) * GB_LDH + kk]) = bf16_pack2(l0 ? r[0][i] : 0.0f, l1 ? r[1][i] : 0.0f);
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;

#pragma unroll
  for (int p = 0; p < GB_PF; p++) {
    load_tile(kbeg + p * GB_BK, ra[p], rb[p]);
    SCHED_FENCE();
  }
  const int fk = lane >> 4, fi = lane & 15;
  for (int kb = kbeg; kb < kend; kb += GB_PF * GB_BK) {
#pragma unroll
    for (int p = 0; p < GB_PF; p++) {
      const int k0 = kb + p * GB_BK;   // phases past the slab multiply zeros
      stage(AMODE, As, a_mn, a_k, k0, ra[p]);
      stage(BMODE, Bs, b_mn, b_k, k0, rb[p]);
      __syncthreads();
      load_tile(k0 + GB_PF * GB_BK, ra[p], rb[p]);
      SCHED_FENCE();
      u16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        af[i] = *reinterpret_cast<const u16x8*>(&As[(wm * 32 + i * 16 + fi) * GB_LDH + fk * 8]);
        bf[i] = *reinterpret_cast<const u16x8*>(&Bs[(wn * 32 + i * 16 + fi) * GB_LDH + fk * 8]);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma16x16x32_bf16(af[i], bf[j], acc[i][j]);
      __syncthreads();
    }
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

// ---- 128 x 128 tile ------------------------------------------------------------------------------------------------
// The 64 x 64 kernel above moves 2 x 64 x 32 f32 through a CU's vector cache per 4 MFMAs of a wave: at the configs[4]
// shapes it ran at 5-7 % of the bf16 MFMA rate (weight-gradient product of 2 x BiLSTM(512): 430 GFLOP in 3.0 ms).  Here a
// workgroup owns 128 x 128 (its four waves 64 x 64 each = 4 x 4 MFMA tiles, 16 MFMAs per 8 fragment reads), which halves
// the operand bytes per flop; the operand tiles are double-buffered in LDS (ONE barrier per 32-k block: block t+1 is
// converted and written into the other buffer while the MFMAs of block t run) behind a ring of three register-staged
// blocks (2 KB of loads per thread in flight, two workgroups per CU).  Operands stay f32 in memory and are rounded to
// bf16 while staged, as above, so with 0.0625 B per MAC the kernel is bound by the vector-cache fill rate (64 B/clk per
// CU = half the bf16 MFMA rate) -- the next step would be bf16 copies of the activations written by their producers.
//   KC staging: thread t -> row t>>1, 16 consecutive k at (t&1)*16: four float4 -> two ds_write_b128
//   MC staging: lane l of wave w -> mn = 32 w + 4 (l&7) .. +3, k = 4 (l>>3) .. +3: four float4 (one per k), transposed
//               in registers -> four ds_write_b64 (row mn+i, 4 consecutive k)
constexpr int GB2_BT = 128;
constexpr int GB2_PF = 3;
// LDS image of an operand block: [mn][32 k] bf16, 64-byte rows WITHOUT padding; the four 16-byte k-chunks of row r sit
// at chunk position c ^ gb2_sw(r).  Found by enumerating layouts against the instruction lane groups of the LDS
// (MI355X_MICROARCH.md, LDS): fragment reads (ds_read_b128, 4 x 16 lanes) and the KC staging writes (ds_write_b128,
// 8 x 8 lanes) are conflict-free, the transposing MC staging writes (ds_write_b64, 4 x 16 lanes) 2-way.  The padded
// [mn][40] image of the 64 x 64 kernel costs 2x / 2x / 4x on the same three -- and the LDS array, not the MFMA pipe, is
// what this kernel saturates first (measured by leaving parts out: staging alone was 35 % of the weight-gradient GEMM).
constexpr int GB2_LDH = 32;
constexpr int GB2_TILE = GB2_BT * GB2_LDH;   // halfs per operand buffer
DEVFN int gb2_sw(int row) { return ((row >> 1) ^ (row >> 2)) & 3; }

// Epilogue of the 128 x 128 kernels.  Their MFMAs are issued with the operands exchanged (B fragment first), so an
// accumulator holds the TRANSPOSED 16 x 16 tile: lane l has four consecutive columns (4 (l >> 4) .. + 3) of output row
// l & 15 -- one 16-byte store (row4) instead of four 4-byte stores of a column piece; a wave's store instruction covers
// sixteen 64-byte row segments.
template <class FE, int WI = 4>
DEVFN void gb2_store(const FE& fe, const f32x4 (&acc)[WI][4], const int rw, const int cw, const int lane, const int R, const int Cn, const int z) {
  const bool v4 = fe.vec4();
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int r = rw + i * 16 + (lane & 15);
      const int c = cw + j * 16 + (lane >> 4) * 4;
      if (r < R) {
        if (v4 && c + 3 < Cn) fe.row4(r, c, acc[i][j], z);
        else {
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (c + e < Cn) fe(r, c + e, acc[i][j][e], z);
        }
      }
    }
}

template <int AMODE, int BMODE, class FE>
__global__ __launch_bounds__(256, 2) void gemm_bf16_128_kernel(GemmOperand A, GemmOperand B, FE fe, int R, int Cn,
                                                               int K, int ksplit, int nsplit) {
  __shared__ __attribute__((aligned(16))) unsigned short As[2 * GB2_TILE];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * GB2_TILE];
  constexpr bool COAL = false;   // (the row-per-load mapping of gemm_x3_128_kernel is not built for this kernel)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by, z;   // XCD-aware tile order, see gemm_mfma.h
  {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = lin & 7u, idx = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = (int)(v % gx);
    by = (int)((v / gx) % gy);
    z = (int)(v / (gx * gy));
  }
  const int r0 = by * GB2_BT, c0 = bx * GB2_BT;
  const int batch = z / nsplit;
  const int kbeg = (z - batch * nsplit) * ksplit;
  const int kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;

  const int a_mn = AMODE == GEMM_KC ? (COAL ? (tid >> 3) : (tid >> 1)) : wave * 32 + (lane & 7) * 4;
  const int a_k = AMODE == GEMM_KC ? (COAL ? (tid & 7) * 4 : (tid & 1) * 16) : (lane >> 3) * 4;
  const int b_mn = BMODE == GEMM_KC ? (COAL ? (tid >> 3) : (tid >> 1)) : wave * 32 + (lane & 7) * 4;
  const int b_k = BMODE == GEMM_KC ? (COAL ? (tid & 7) * 4 : (tid & 1) * 16) : (lane >> 3) * 4;
  const BufF32 abuf = make_buf(A.p + batch * A.bstride, (size_t)(A.elems - batch * A.bstride) * 4);
  const BufF32 bbuf = make_buf(B.p + batch * B.bstride, (size_t)(B.elems - batch * B.bstride) * 4);
  const unsigned a_base = AMODE == GEMM_KC ? (unsigned)(r0 + a_mn) * A.ld + a_k : (unsigned)a_k * A.ld + r0 + a_mn;
  const unsigned b_base = BMODE == GEMM_KC ? (unsigned)(c0 + b_mn) * B.ld + b_k : (unsigned)b_k * B.ld + c0 + b_mn;
  const unsigned a_kstep = AMODE == GEMM_KC ? 1u : (unsigned)A.ld, b_kstep = BMODE == GEMM_KC ? 1u : (unsigned)B.ld;
  const unsigned a_next = AMODE == GEMM_KC ? (COAL ? 32u * (unsigned)A.ld : 4u) : (unsigned)A.ld;   // load j: 32 rows on / 4 k on / 1 k row on
  const unsigned b_next = BMODE == GEMM_KC ? (COAL ? 32u * (unsigned)B.ld : 4u) : (unsigned)B.ld;

  // Block addresses: per-lane byte offsets of the four float4 (fixed) + the block's offset (one add per load, the sum
  // stays inside the descriptor's bounds check).  Blocks past the slab re-read its last block (their products are
  // zeroed when staged): no select per load.
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    aoff[j] = (a_base + (unsigned)j * a_next) * 4u;
    boff[j] = (b_base + (unsigned)j * b_next) * 4u;
  }
  const int klast = kbeg + ((kend - kbeg - 1) / GB_BK) * GB_BK;   // first k of the slab's last block (kend > kbeg)
  f32x4 ra[GB2_PF][4], rb[GB2_PF][4];
  auto load_tile = [&](int k0, f32x4 (&a)[4], f32x4 (&b)[4]) {
    const unsigned kc = (unsigned)wave_uniform(k0 < klast ? k0 : klast);
    const unsigned ao = kc * a_kstep * 4u, bo = kc * b_kstep * 4u;
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = buf_load4(abuf, aoff[j] + ao);
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = buf_load4(bbuf, boff[j] + bo);
  };
  // round to bf16 and store k-contiguous.  Whole blocks take the plain path; the slab's last block (and the
  // zero blocks behind it) mask contraction indices >= kend per element (one wave-uniform branch per block).
  auto stage = [&](const int MODE, unsigned short* S, const int mn, const int kk, const int k0, const f32x4 (&r)[4]) {
    const bool whole = wave_uniform(k0 + GB_BK <= kend ? 1 : 0) != 0;
    if (MODE == GEMM_KC) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float x[8];
        if (whole) {
#pragma unroll
          for (int i = 0; i < 8; i++) x[i] = r[2 * h + (i >> 2)][i & 3];
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) x[i] = (k0 + kk + 8 * h + i < kend) ? r[2 * h + (i >> 2)][i & 3] : 0.0f;
        }
        *reinterpret_cast<u16x8*>(&S[mn * GB2_LDH + ((((kk >> 3) + h) ^ gb2_sw(mn)) << 3)]) = bf16_pack8(x);
      }
    } else {
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = r[j];
      if (!whole) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) v[j][i] = (k0 + kk + j < kend) ? v[j][i] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u32x2 w2;
        w2[0] = bf16_pack2(v[0][i], v[1][i]);
        w2[1] = bf16_pack2(v[2][i], v[3][i]);
        *reinterpret_cast<u32x2*>(&S[(mn + i) * GB2_LDH + (((kk >> 3) ^ gb2_sw(mn + i)) << 3) + (kk & 4)]) = w2;
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;

#pragma unroll
  for (int p = 0; p < GB2_PF; p++) {
    load_tile(kbeg + p * GB_BK, ra[p], rb[p]);
    SCHED_FENCE();
  }
  // block 0 -> buffer 0
  stage(AMODE, As, a_mn, a_k, kbeg, ra[0]);
  stage(BMODE, Bs, b_mn, b_k, kbeg, rb[0]);
  load_tile(kbeg + GB2_PF * GB_BK, ra[0], rb[0]);
  SCHED_FENCE();
  __syncthreads();
  const int fk = lane >> 4, fi = lane & 15;
  const int fsw = (fk ^ gb2_sw(fi)) << 3;   // this lane's chunk position (rows 16 apart share the swizzle)
  int cur = 0;   // LDS buffer (in halfs) that holds the block the MFMAs are about to consume
  for (int kb = kbeg; kb < kend; kb += GB2_PF * GB_BK) {
#pragma unroll
    for (int p = 0; p < GB2_PF; p++) {
      const int k0 = kb + p * GB_BK;   // block in LDS buffer `cur` (phases past the slab multiply zeros)
      constexpr int pn_of[3] = {1, 2, 0};
      const int pn = pn_of[p];         // register set of block k0 + 32: convert it into the other buffer ...
      {
      stage(AMODE, As + (cur ^ GB2_TILE), a_mn, a_k, k0 + GB_BK, ra[pn]);
      stage(BMODE, Bs + (cur ^ GB2_TILE), b_mn, b_k, k0 + GB_BK, rb[pn]);
      }
      load_tile(k0 + GB_BK + GB2_PF * GB_BK, ra[pn], rb[pn]);   // ... and re-use the set for the block three ahead
      SCHED_FENCE();
      u16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        af[i] = *reinterpret_cast<const u16x8*>(&As[cur + (wm * 64 + i * 16 + fi) * GB2_LDH + fsw]);
        bf[i] = *reinterpret_cast<const u16x8*>(&Bs[cur + (wn * 64 + i * 16 + fi) * GB2_LDH + fsw]);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = mfma16x16x32_bf16(bf[j], af[i], acc[i][j]);   // transposed: see gb2_store
      __syncthreads();
      cur ^= GB2_TILE;
    }
  }
  gb2_store(fe, acc, r0 + wm * 64, c0 + wn * 64, lane, R, Cn, z);
}

// ---- the same 128 x 128 tile with f32-GRADE products: hi + lo split operands, three MFMAs per product ------------------
// (gemm_x3_body's arithmetic -- x = hi + lo, hi = bf16(x), lo = bf16(x - hi), product = hi.hi + hi.lo + lo.hi, < 2^-16 |x||y|
// dropped per product, f32 accumulation -- on the big tile: the backward products of WIDE layers in the exact-f32 mode.
// Their f32-MFMA form runs near its own roof (96-114 TFLOP/s of 157) and was 6.5 of the 15.2 ms of a configs[4] step; three
// bf16 MFMAs of 16 cycles replace eight f32 MFMAs of 32.)  Each operand block lives in LDS as a hi and a lo image
// (4 x 8 KB per buffer, two buffers = 64 KB: one workgroup per CU by LDS... two by registers).
// COAL (k-contiguous operands): a thread's four 16-byte loads of a block are four ROWS, eight neighbouring lanes covering 128
// contiguous bytes of one row -- the address pipeline takes a 16-byte access of a lane that has no neighbour as a request of its
// own, and the older form (a thread = 64 contiguous bytes of one row, neighbouring lanes 64 bytes apart) issued 64 per instruction.
// ILV: the conversion of block t + 1 (VALU + LDS writes into the other buffer) and the loads of block t + 4 are issued BETWEEN the 48
// MFMAs of block t (sched_group_barrier pattern) instead of in front of them: a workgroup is four waves, one per SIMD, and a wave
// issues in order -- staged first, its ~130 conversion instructions (4 cycles each) and its MFMAs (16 cycles each) ran one after
// the other.
#ifdef CLSTM_HIP_EMU
#define GX3_SGB(mask, n) do {} while (0)
#else
#define GX3_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif
template <int AMODE, int BMODE, class FE, bool COAL = true, bool ILV = true>
__global__ __launch_bounds__(256, 2) void gemm_x3_128_kernel(GemmOperand A, GemmOperand B, FE fe, int R, int Cn,
                                                             int K, int ksplit, int nsplit) {
  unsigned short* const x3_smem = dyn_smem<unsigned short>();   // [A hi | A lo | B hi | B lo] x 2 buffers
  unsigned short* const As = x3_smem;                 // hi images: As[cur], lo images: As[cur + 2 * GB2_TILE]
  unsigned short* const Bs = x3_smem + 4 * GB2_TILE;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by, z;   // XCD-aware tile order, see gemm_mfma.h
  {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = lin & 7u, idx = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = (int)(v % gx);
    by = (int)((v / gx) % gy);
    z = (int)(v / (gx * gy));
  }
  const int r0 = by * GB2_BT, c0 = bx * GB2_BT;
  const int batch = z / nsplit;
  const int kbeg = (z - batch * nsplit) * ksplit;
  const int kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;

  const int a_mn = AMODE == GEMM_KC ? (COAL ? (tid >> 3) : (tid >> 1)) : wave * 32 + (lane & 7) * 4;
  const int a_k = AMODE == GEMM_KC ? (COAL ? (tid & 7) * 4 : (tid & 1) * 16) : (lane >> 3) * 4;
  const int b_mn = BMODE == GEMM_KC ? (COAL ? (tid >> 3) : (tid >> 1)) : wave * 32 + (lane & 7) * 4;
  const int b_k = BMODE == GEMM_KC ? (COAL ? (tid & 7) * 4 : (tid & 1) * 16) : (lane >> 3) * 4;
  const BufF32 abuf = make_buf(A.p + batch * A.bstride, (size_t)(A.elems - batch * A.bstride) * 4);
  const BufF32 bbuf = make_buf(B.p + batch * B.bstride, (size_t)(B.elems - batch * B.bstride) * 4);
  const unsigned a_base = AMODE == GEMM_KC ? (unsigned)(r0 + a_mn) * A.ld + a_k : (unsigned)a_k * A.ld + r0 + a_mn;
  const unsigned b_base = BMODE == GEMM_KC ? (unsigned)(c0 + b_mn) * B.ld + b_k : (unsigned)b_k * B.ld + c0 + b_mn;
  const unsigned a_kstep = AMODE == GEMM_KC ? 1u : (unsigned)A.ld, b_kstep = BMODE == GEMM_KC ? 1u : (unsigned)B.ld;
  const unsigned a_next = AMODE == GEMM_KC ? (COAL ? 32u * (unsigned)A.ld : 4u) : (unsigned)A.ld;   // load j: 32 rows on / 4 k on / 1 k row on
  const unsigned b_next = BMODE == GEMM_KC ? (COAL ? 32u * (unsigned)B.ld : 4u) : (unsigned)B.ld;

  // Block addresses: per-lane byte offsets of the four float4 (fixed) + the block's offset (one add per load, the sum
  // stays inside the descriptor's bounds check).  Blocks past the slab re-read its last block (their products are
  // zeroed when staged): no select per load.
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    aoff[j] = (a_base + (unsigned)j * a_next) * 4u;
    boff[j] = (b_base + (unsigned)j * b_next) * 4u;
  }
  const int klast = kbeg + ((kend - kbeg - 1) / GB_BK) * GB_BK;   // first k of the slab's last block (kend > kbeg)
  f32x4 ra[GB2_PF][4], rb[GB2_PF][4];
  auto load_tile = [&](int k0, f32x4 (&a)[4], f32x4 (&b)[4]) {
    const unsigned kc = (unsigned)wave_uniform(k0 < klast ? k0 : klast);
    const unsigned ao = kc * a_kstep * 4u, bo = kc * b_kstep * 4u;
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = buf_load4(abuf, aoff[j] + ao);
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = buf_load4(bbuf, boff[j] + bo);
  };
  // round to bf16 and store k-contiguous.  Whole blocks take the plain path; the slab's last block (and the
  // zero blocks behind it) mask contraction indices >= kend per element (one wave-uniform branch per block).
  auto stage = [&](const int MODE, unsigned short* S, const int mn, const int kk, const int k0, const f32x4 (&r)[4]) {
    const bool whole = wave_uniform(k0 + GB_BK <= kend ? 1 : 0) != 0;
    if (MODE == GEMM_KC && COAL) {
#pragma unroll
      for (int j = 0; j < 4; j++) {   // row mn + 32 j, contraction indices kk .. kk + 3
        f32x4 v = r[j];
        if (!whole) {
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = (k0 + kk + i < kend) ? v[i] : 0.0f;
        }
        u32x2 w2, l2;
        w2[0] = bf16_pack2(v[0], v[1]);
        w2[1] = bf16_pack2(v[2], v[3]);
        l2[0] = bf16_pack2(v[0] - __builtin_bit_cast(float, w2[0] << 16), v[1] - __builtin_bit_cast(float, w2[0] & 0xFFFF0000u));
        l2[1] = bf16_pack2(v[2] - __builtin_bit_cast(float, w2[1] << 16), v[3] - __builtin_bit_cast(float, w2[1] & 0xFFFF0000u));
        const int row = mn + 32 * j;
        const int at = row * GB2_LDH + (((kk >> 3) ^ gb2_sw(row)) << 3) + (kk & 4);
        *reinterpret_cast<u32x2*>(&S[at]) = w2;
        *reinterpret_cast<u32x2*>(&S[at + 2 * GB2_TILE]) = l2;
      }
    } else if (MODE == GEMM_KC) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float x[8];
        if (whole) {
#pragma unroll
          for (int i = 0; i < 8; i++) x[i] = r[2 * h + (i >> 2)][i & 3];
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) x[i] = (k0 + kk + 8 * h + i < kend) ? r[2 * h + (i >> 2)][i & 3] : 0.0f;
        }
        const u16x8 hi = bf16_pack8(x);
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; i++) y[i] = x[i] - __builtin_bit_cast(float, (unsigned)hi[i] << 16);   // exact
        const int at = mn * GB2_LDH + ((((kk >> 3) + h) ^ gb2_sw(mn)) << 3);
        *reinterpret_cast<u16x8*>(&S[at]) = hi;
        *reinterpret_cast<u16x8*>(&S[at + 2 * GB2_TILE]) = bf16_pack8(y);
      }
    } else {
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = r[j];
      if (!whole) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) v[j][i] = (k0 + kk + j < kend) ? v[j][i] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u32x2 w2, l2;
        w2[0] = bf16_pack2(v[0][i], v[1][i]);
        w2[1] = bf16_pack2(v[2][i], v[3][i]);
        l2[0] = bf16_pack2(v[0][i] - __builtin_bit_cast(float, w2[0] << 16), v[1][i] - __builtin_bit_cast(float, w2[0] & 0xFFFF0000u));
        l2[1] = bf16_pack2(v[2][i] - __builtin_bit_cast(float, w2[1] << 16), v[3][i] - __builtin_bit_cast(float, w2[1] & 0xFFFF0000u));
        const int at = (mn + i) * GB2_LDH + (((kk >> 3) ^ gb2_sw(mn + i)) << 3) + (kk & 4);
        *reinterpret_cast<u32x2*>(&S[at]) = w2;
        *reinterpret_cast<u32x2*>(&S[at + 2 * GB2_TILE]) = l2;
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;

#pragma unroll
  for (int p = 0; p < GB2_PF; p++) {
    load_tile(kbeg + p * GB_BK, ra[p], rb[p]);
    SCHED_FENCE();
  }
  // block 0 -> buffer 0
  stage(AMODE, As, a_mn, a_k, kbeg, ra[0]);
  stage(BMODE, Bs, b_mn, b_k, kbeg, rb[0]);
  load_tile(kbeg + GB2_PF * GB_BK, ra[0], rb[0]);
  SCHED_FENCE();
  __syncthreads();
  const int fk = lane >> 4, fi = lane & 15;
  const int fsw = (fk ^ gb2_sw(fi)) << 3;   // this lane's chunk position (rows 16 apart share the swizzle)
  int cur = 0;   // LDS buffer (in halfs) that holds the block the MFMAs are about to consume
  for (int kb = kbeg; kb < kend; kb += GB2_PF * GB_BK) {
#pragma unroll
    for (int p = 0; p < GB2_PF; p++) {
      const int k0 = kb + p * GB_BK;   // block in LDS buffer `cur` (phases past the slab multiply zeros)
      constexpr int pn_of[3] = {1, 2, 0};
      const int pn = pn_of[p];         // register set of block k0 + 32: convert it into the other buffer ...
      if (!ILV) {
      stage(AMODE, As + (cur ^ GB2_TILE), a_mn, a_k, k0 + GB_BK, ra[pn]);
      stage(BMODE, Bs + (cur ^ GB2_TILE), b_mn, b_k, k0 + GB_BK, rb[pn]);
      load_tile(k0 + GB_BK + GB2_PF * GB_BK, ra[pn], rb[pn]);   // ... and re-use the set for the block three ahead
      SCHED_FENCE();
      }
      if (!ILV) {
        u16x8 af[4], bf[4], al[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          af[i] = *reinterpret_cast<const u16x8*>(&As[cur + (wm * 64 + i * 16 + fi) * GB2_LDH + fsw]);
          bf[i] = *reinterpret_cast<const u16x8*>(&Bs[cur + (wn * 64 + i * 16 + fi) * GB2_LDH + fsw]);
          al[i] = *reinterpret_cast<const u16x8*>(&As[cur + 2 * GB2_TILE + (wm * 64 + i * 16 + fi) * GB2_LDH + fsw]);
          bl[i] = *reinterpret_cast<const u16x8*>(&Bs[cur + 2 * GB2_TILE + (wn * 64 + i * 16 + fi) * GB2_LDH + fsw]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) {   // small terms first (transposed: see gb2_store)
            acc[i][j] = mfma16x16x32_bf16(bl[j], af[i], acc[i][j]);
            acc[i][j] = mfma16x16x32_bf16(bf[j], al[i], acc[i][j]);
            acc[i][j] = mfma16x16x32_bf16(bf[j], af[i], acc[i][j]);
          }
      } else {
        // two halves of 24 MFMAs (A row tiles 0-1, then 2-3: 48 instead of 64 fragment registers live), the A conversion between the
        // MFMAs of the first, the B conversion and the loads between those of the second
        u16x8 bf[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          bf[j] = *reinterpret_cast<const u16x8*>(&Bs[cur + (wn * 64 + j * 16 + fi) * GB2_LDH + fsw]);
          bl[j] = *reinterpret_cast<const u16x8*>(&Bs[cur + 2 * GB2_TILE + (wn * 64 + j * 16 + fi) * GB2_LDH + fsw]);
        }
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          u16x8 af[2], al[2];
#pragma unroll
          for (int i = 0; i < 2; i++) {
            af[i] = *reinterpret_cast<const u16x8*>(&As[cur + (wm * 64 + (2 * hf + i) * 16 + fi) * GB2_LDH + fsw]);
            al[i] = *reinterpret_cast<const u16x8*>(&As[cur + 2 * GB2_TILE + (wm * 64 + (2 * hf + i) * 16 + fi) * GB2_LDH + fsw]);
          }
          SCHED_FENCE();
          if (hf == 0) stage(AMODE, As + (cur ^ GB2_TILE), a_mn, a_k, k0 + GB_BK, ra[pn]);
          else {
            stage(BMODE, Bs + (cur ^ GB2_TILE), b_mn, b_k, k0 + GB_BK, rb[pn]);
            load_tile(k0 + GB_BK + GB2_PF * GB_BK, ra[pn], rb[pn]);
          }
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
              acc[2 * hf + i][j] = mfma16x16x32_bf16(bl[j], af[i], acc[2 * hf + i][j]);
              acc[2 * hf + i][j] = mfma16x16x32_bf16(bf[j], al[i], acc[2 * hf + i][j]);
              acc[2 * hf + i][j] = mfma16x16x32_bf16(bf[j], af[i], acc[2 * hf + i][j]);
            }
#pragma unroll
          for (int q = 0; q < 24; q++) {   // one MFMA, three conversion instructions; an LDS write every third, a load every third of the second half
            GX3_SGB(0x008, 1);
            GX3_SGB(0x002, 3);
            if (q % 3 == 0) GX3_SGB(0x200, 1);
            if (hf == 1 && q % 3 == 1) GX3_SGB(0x020, 1);
          }
          SCHED_FENCE();
        }
      }
      __syncthreads();
      cur ^= GB2_TILE;
    }
  }
  gb2_store(fe, acc, r0 + wm * 64, c0 + wn * 64, lane, R, Cn, z);
}

// ---- 128 x 128 tile, BOTH operands already bf16 and k-contiguous in memory ------------------------------------------
//   out(r, c) = sum_k A16[r * lda + k] * B16[c * ldb + k]
// What the f32-source kernel spends most of its time on -- 15 GB of f32 operands through the L2s per configs[4] step, the
// conversion, the transposing LDS writes -- does not exist here: a thread moves 2 x 16 bytes (16 k) per operand and
// block straight from memory into the swizzled LDS image (ds_write_b128, conflict-free).  Used where a producer can leave
// a bf16 k-contiguous copy of its output at no extra cost (the persistent recurrence's delta ring IS that array) and
// for the weight operand (packed once per update).  No split-K, no batch: the products it serves have K <= 4096.
struct GemmOperand16 { const unsigned short* p; int ld; long long elems; };   // ld, elems in halfs (ld even)
// (WI as in gemm_b16mc_kernel below: 4 -> 128 x 128 tile / four waves, 8 -> 256 x 256 / eight waves of 128 x 64)
// STAG (WI = 8 only): the two waves of a SIMD run HALF A BLOCK APART.  With one barrier per block all eight waves stage and read
// their fragments together (~800 cycles of LDS traffic with the matrix pipe idle) and then issue their MFMAs together (~1,000
// cycles per SIMD with the LDS idle): 46 % MFMA busy at best (profiles/r04_b2_mfma_utilisation.txt).  Here waves 0-3 (one per
// SIMD, the upper 128 rows of the tile) and waves 4-7 (the lower 128) alternate: while one group stages its half of block t+1
// and reads its fragments of block t, the other group's 32 MFMAs of the previous half-step run -- two barriers per block, group
// B one barrier behind (it enters the loop through an extra barrier, group A leaves through one).  Every staged element, every
// fragment and every MFMA is the one-barrier loop's, in the same order per accumulator: the results are bit-identical.
//   A: [stage t+1 | frags t] B [MFMA t]            B [stage t+2 | frags t+1] B [MFMA t+1] ...
//   B:                       B [stage t+1 | frags t] B [MFMA t]             B ...
// Block t+1's buffer is the one block t-1 lived in: its last reader was group B between the two barriers before group A's
// stage of t+1 (reads are complete at a barrier: __syncthreads waits for lgkmcnt).
#ifdef CLSTM_GEMM_PROF   // diagnostics build (make variant VARIANT=gprof EXTRA=-DCLSTM_GEMM_PROF; scripts/gpu_gemmprof_r5.py)
// per-segment shader-clock sums of waves 0 and 4 of workgroup 0: [wave group][segment]; LO: parts of the loop left out
__device__ long long clstm_gemm_prof[2 * 8];
#define GPROF_DECL long long gp_t = 0, gp_s[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const bool gp_on = blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 255) == 0
#define GPROF_TOP() do { if (gp_on) gp_t = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define GPROF(i) do { if (gp_on) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); gp_s[i] += t_ - gp_t; gp_t = t_; } } while (0)
#define GPROF_WRITE() do { if (gp_on) for (int i_ = 0; i_ < 8; i_++) clstm_gemm_prof[(threadIdx.x >> 8) * 8 + i_] = gp_s[i_]; } while (0)
#else
#define GPROF_DECL
#define GPROF_TOP()
#define GPROF(i)
#define GPROF_WRITE()
#endif
template <class FE, int WI, bool STAG = false, int LO = 0>
__global__ __launch_bounds__(64 * WI, 2) void gemm_b16kk_kernel(GemmOperand16 A, GemmOperand16 B, FE fe, int R, int Cn, int K) {
  static_assert(!STAG || WI == 8, "the staggered loop pairs the two waves of each SIMD of an eight-wave workgroup");
  constexpr int NWN = WI / 2, BT = 32 * WI, TILE = BT * GB2_LDH;
  __shared__ __attribute__((aligned(16))) unsigned short As[2 * TILE];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  int bx, by;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy;
    const unsigned lin = blockIdx.x + gx * blockIdx.y;
    const unsigned xcd = lin & 7u, idx = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = (int)(v % gx);
    by = (int)(v / gx);
  }
  const int r0 = by * BT, c0 = bx * BT;
  const int s_mn = tid >> 1, s_k = (tid & 1) * 16;   // row of the tile, first of its 16 k
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(A.p), (size_t)A.elems * 2);
  const BufF32 bbuf = make_buf(reinterpret_cast<const float*>(B.p), (size_t)B.elems * 2);
  const unsigned aoff = ((unsigned)(r0 + s_mn) * (unsigned)A.ld + (unsigned)s_k) * 2u;
  const unsigned boff = ((unsigned)(c0 + s_mn) * (unsigned)B.ld + (unsigned)s_k) * 2u;
  const int klast = ((K - 1) / GB_BK) * GB_BK;
  f32x4 ra[GB2_PF][2], rb[GB2_PF][2];
  auto load_tile = [&](int k0, f32x4 (&a)[2], f32x4 (&b)[2]) {
    const unsigned kc = (unsigned)wave_uniform(k0 < klast ? k0 : klast) * 2u;
    a[0] = buf_load4(abuf, aoff + kc);
    a[1] = buf_load4(abuf, aoff + kc + 16u);
    b[0] = buf_load4(bbuf, boff + kc);
    b[1] = buf_load4(bbuf, boff + kc + 16u);
  };
  auto stage = [&](unsigned short* S, const int k0, const f32x4 (&r)[2]) {
    const bool whole = wave_uniform(k0 + GB_BK <= K ? 1 : 0) != 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      u16x8 v = __builtin_bit_cast(u16x8, r[h]);
      if (!whole) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (k0 + s_k + 8 * h + i < K) ? v[i] : (unsigned short)0;
      }
      *reinterpret_cast<u16x8*>(&S[s_mn * GB2_LDH + ((((s_k >> 3) + h) ^ gb2_sw(s_mn)) << 3)]) = v;
    }
  };
  f32x4 acc[WI][4];
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;
#pragma unroll
  for (int p = 0; p < GB2_PF; p++) {
    load_tile(p * GB_BK, ra[p], rb[p]);
    SCHED_FENCE();
  }
  stage(As, 0, ra[0]);
  stage(Bs, 0, rb[0]);
  load_tile(GB2_PF * GB_BK, ra[0], rb[0]);
  SCHED_FENCE();
  __syncthreads();
  const int fk = lane >> 4, fi = lane & 15;
  const int fsw = (fk ^ gb2_sw(fi)) << 3;
  int cur = 0;
  GPROF_DECL;
  u16x8 af[WI], bf[4];
  if (LO & 8) {   // (diagnostics: fragments read once)
#pragma unroll
    for (int i = 0; i < WI; i++) af[i] = *reinterpret_cast<const u16x8*>(&As[(wm * (16 * WI) + i * 16 + fi) * GB2_LDH + fsw]);
#pragma unroll
    for (int j = 0; j < 4; j++) bf[j] = *reinterpret_cast<const u16x8*>(&Bs[(wn * 64 + j * 16 + fi) * GB2_LDH + fsw]);
  }
  if (STAG && wm == 1) __syncthreads();   // group B runs one barrier behind group A (wave-uniform)
  for (int kb = 0; kb < K; kb += GB2_PF * GB_BK) {
#pragma unroll
    for (int p = 0; p < GB2_PF; p++) {
      const int k0 = kb + p * GB_BK;
      const int pn = p == GB2_PF - 1 ? 0 : p + 1;
      GPROF_TOP();
      if (!(LO & 4)) {
        stage(As + (cur ^ TILE), k0 + GB_BK, ra[pn]);
        stage(Bs + (cur ^ TILE), k0 + GB_BK, rb[pn]);
      }
      GPROF(0);   // staged (incl. the wait for the block's loads)
      if (!(LO & 2)) load_tile(k0 + GB_BK + GB2_PF * GB_BK, ra[pn], rb[pn]);
      SCHED_FENCE();
      if (!(LO & 8)) {
#pragma unroll
        for (int i = 0; i < WI; i++) af[i] = *reinterpret_cast<const u16x8*>(&As[cur + (wm * (16 * WI) + i * 16 + fi) * GB2_LDH + fsw]);
#pragma unroll
        for (int j = 0; j < 4; j++) bf[j] = *reinterpret_cast<const u16x8*>(&Bs[cur + (wn * 64 + j * 16 + fi) * GB2_LDH + fsw]);
      }
      GPROF(1);   // loads + fragment reads issued
      if (STAG) {                          // ... the other group's MFMAs ran beside the staging and the reads above
        // (the fences pin the MFMAs BETWEEN the two barriers: they touch no memory, so nothing else keeps hipcc from issuing
        //  them in front of the first one as the fragments arrive -- which is the one-barrier loop again)
        SCHED_FENCE();
        __syncthreads();
        SCHED_FENCE();
      }
      GPROF(2);   // fragments arrived + barrier
      if (LO & 1) {
#pragma unroll
        for (int i = 0; i < WI; i++) asm volatile("" ::"v"(af[i]));
#pragma unroll
        for (int j = 0; j < 4; j++) asm volatile("" ::"v"(bf[j]));
      } else {
#pragma unroll
      for (int i = 0; i < WI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16x16x32_bf16(bf[j], af[i], acc[i][j]);   // transposed: see gb2_store
      }
      if (STAG) SCHED_FENCE();
      GPROF(3);   // MFMAs issued
      __syncthreads();
      if (STAG) SCHED_FENCE();
      GPROF(4);   // barrier
      cur ^= TILE;
    }
  }
  GPROF_WRITE();
  if (STAG && wm == 0) __syncthreads();   // (group A's epilogue stores run beside group B's last MFMAs)
  gb2_store<FE, WI>(fe, acc, r0 + wm * (16 * WI), c0 + wn * 64, lane, R, Cn, 0);
}
// ---- the same product with the operand tiles brought in by LDS-DMA (round 5) ----------------------------------------------
// Phase stamps and leave-out builds of the staggered loop above (scripts/gpu_gemmprof_r5.py, profiles/r05_gemm_phases.txt): of the
// ~1,950 cycles a 32-k block takes, ~500 are the four ds_write_b128 per thread that stage the next block (the VGPR -> LDS
// transfer path: ~125 cycles per wave instruction beside the other group's MFMAs) and with them gone the kernel ran 37 % faster
// (4096^3: 163 -> 103 us) -- more than without global loads (-13 %) or fragment reads (-18 %).  Here nothing is staged through
// registers: each wave brings 32 rows of both tiles per block with four buffer_load_dwordx4 ... lds (lds_dma16: 1 KB = 16 rows
// x 64 B per instruction, lane l -> row l >> 2, chunk POSITION l & 3, which holds source chunk (l & 3) ^ gb2_sw(row): the
// swizzle of the register-staged image, applied to the source address), THREE buffers of A | B (96 KB), block t + 2 requested
// while block t is consumed.  Fragment reads, MFMA order and epilogue are the staggered loop's: bit-identical results.
//   group A (waves 0-3):  W b0 [dma t+2 | frags t] b [MFMA t | W] b [dma t+3 | frags t+1] b ...
//   group B (waves 4-7):  W b0       b [dma t+2 | frags t | W] b [MFMA t] b ...               (W = wait for MY part of block t+1)
// A block's buffer is reused by the request of block t + 3 -- issued behind a barrier that every wave passes only after its
// fragment reads of block t have returned.  Every wave waits for its own requests of block t + 1 in front of the barrier that
// precedes group A's reads of it (vmcnt(4): the four requests of block t + 2 may still be in flight).  K % 32 == 0 (a DMA
// cannot mask a contraction tail); rows past the operand read zeros (the descriptor ends at the last row).
// Epilogue of the LDS-DMA kernels (eight waves of 128 x 64, accumulators transposed as in gb2_store) THROUGH LDS: a lane's
// accumulator registers are four columns of sixteen different rows, so a direct store instruction writes sixteen 64-byte
// pieces -- half cache lines; 4,096 of them per 256 x 256 tile, and the 419 MB of pre-activations of a configs[4] layer left
// the chip in half lines (halving the BYTES with bf16 outputs bought 13 us of 281: it is the pieces that cost,
// profiles/r05_gemm_variants.txt).  Here a wave lays two 16-row strips (32 rows x 64 columns) into its own 8.5 KB of the
// now idle operand buffers (row stride 68 floats: conflict-free both ways) and reads them back row-wise: a store instruction
// then writes four rows x 256 contiguous bytes -- whole lines.  No barrier: the region is the wave's own.
constexpr int GKD_EPI_LD = 68, GKD_EPI_FLOATS = 32 * GKD_EPI_LD;   // per wave
template <class FE, int WI>
DEVFN void gb2_store_lds(float* const epi, const FE& fe, const f32x4 (&acc)[WI][4], const int rw, const int cw, const int lane, const int R, const int Cn, const int z) {
  static_assert(WI % 2 == 0, "two strips per pass");
#pragma unroll
  for (int p = 0; p < WI / 2; p++) {
#pragma unroll
    for (int ii = 0; ii < 2; ii++)
#pragma unroll
      for (int j = 0; j < 4; j++) *reinterpret_cast<f32x4*>(&epi[(ii * 16 + (lane & 15)) * GKD_EPI_LD + j * 16 + (lane >> 4) * 4]) = acc[2 * p + ii][j];
    wave_lds_fence();   // (the rows this lane reads were written by other lanes of its wave)
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int rl = 4 * q + (lane >> 4), cl = (lane & 15) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&epi[rl * GKD_EPI_LD + cl]);
      const int r = rw + p * 32 + rl, c = cw + cl;
      if (r < R) {
        if (c + 3 < Cn) fe.row4(r, c, v, z);
        else {
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (c + e < Cn) fe(r, c + e, v[e], z);
        }
      }
    }
    wave_lds_fence();   // (... and the next pass overwrites them)
  }
}
constexpr int GKD_SMEM_HALFS = 3 * 2 * 256 * GB2_LDH;   // three buffers of A | B: 96 KB
// S: ONE LDS array [buffer][A rows | B rows][32 k]; lin, gx, gy: this workgroup's index in the product's gx x gy tile grid
template <class FE>
DEVFN void gemm_b16kk_dma_body(unsigned short* const S, GemmOperand16 A, GemmOperand16 B, FE fe, int R, int Cn, int K,
                               const unsigned lin, const unsigned gx, const unsigned gy) {
  constexpr int WI = 8, TILE = 256 * GB2_LDH, BUF = 2 * TILE, NBUF = 3;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int bx, by;
  {
    const unsigned total = gx * gy;
    const unsigned xcd = lin & 7u, idx = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = (int)(v % gx);
    by = (int)(v / gx);
  }
  const int r0 = by * 256, c0 = bx * 256;
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(A.p), (size_t)A.elems * 2);
  const BufF32 bbuf = make_buf(reinterpret_cast<const float*>(B.p), (size_t)B.elems * 2);
  // DMA role: rows 32 wave + 16 h + (lane >> 2) of both tiles, position lane & 3 of the row's four 16-byte chunks
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int row = 32 * wave + 16 * h + (lane >> 2);
    const unsigned c = (unsigned)((lane & 3) ^ gb2_sw(row));
    aoff[h] = ((unsigned)(r0 + row) * (unsigned)A.ld + 8u * c) * 2u;
    boff[h] = ((unsigned)(c0 + row) * (unsigned)B.ld + 8u * c) * 2u;
  }
  const int nblk = K / GB_BK;
  auto dma = [&](const int blk, const int buf) {   // block `blk` (clamped: the requests past the end re-read the last block, nobody reads them)
    const unsigned kc = (unsigned)wave_uniform(blk < nblk ? blk : nblk - 1) * (GB_BK * 2u);
    unsigned short* const base = S + buf * BUF + (32 * wave) * GB2_LDH;
#pragma unroll
    for (int h = 0; h < 2; h++) lds_dma16(abuf, aoff[h] + kc, base + (16 * h) * GB2_LDH);
#pragma unroll
    for (int h = 0; h < 2; h++) lds_dma16(bbuf, boff[h] + kc, base + TILE + (16 * h) * GB2_LDH);
  };
  f32x4 acc[WI][4];
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;
  const int fk = lane >> 4, fi = lane & 15;
  const int fsw = (fk ^ gb2_sw(fi)) << 3;
  dma(0, 0);
  dma(1, 1);
  wait_vmcnt<4>();
  wg_barrier();                    // block 0 is in LDS, everybody's part
  if (wm == 1) wg_barrier();       // group B runs one barrier behind
  int cur = 0, nxt2 = 2;           // buffers of block t and of block t + 2
  for (int t = 0; t < nblk; t++) {
    dma(t + 2, nxt2);
    SCHED_FENCE();
    const unsigned short* const As = S + cur * BUF;
    const unsigned short* const Bs = As + TILE;
    u16x8 af[WI], bf[4];
#pragma unroll
    for (int i = 0; i < WI; i++) af[i] = *reinterpret_cast<const u16x8*>(&As[(wm * 128 + i * 16 + fi) * GB2_LDH + fsw]);
#pragma unroll
    for (int j = 0; j < 4; j++) bf[j] = *reinterpret_cast<const u16x8*>(&Bs[(wn * 64 + j * 16 + fi) * GB2_LDH + fsw]);
    SCHED_FENCE();
    if (wm == 1) wait_vmcnt<4>();  // my part of block t + 1 (group B: in front of the barrier group A reads it behind)
    wait_lgkmcnt0();
    SCHED_FENCE();
    wg_barrier();
    SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < WI; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16x16x32_bf16(bf[j], af[i], acc[i][j]);   // transposed: see gb2_store
    SCHED_FENCE();
    if (wm == 0) wait_vmcnt<4>();  // (group A: the same wait behind its MFMAs)
    wg_barrier();
    SCHED_FENCE();
    cur = cur == NBUF - 1 ? 0 : cur + 1;
    nxt2 = nxt2 == NBUF - 1 ? 0 : nxt2 + 1;
  }
  if (wm == 0) wg_barrier();
  wait_vmcnt<0>();                 // (the two requests past the end)
  if (fe.vec4()) {
    wg_barrier();                  // ... everybody's: the operand buffers are free
    static_assert(8 * GKD_EPI_FLOATS * 2 <= GKD_SMEM_HALFS, "epilogue regions fit the operand buffers");
    gb2_store_lds<FE, WI>(reinterpret_cast<float*>(S) + wave * GKD_EPI_FLOATS, fe, acc, r0 + wm * 128, c0 + wn * 64, lane, R, Cn, 0);
  } else gb2_store<FE, WI>(fe, acc, r0 + wm * 128, c0 + wn * 64, lane, R, Cn, 0);
}
template <class FE>
__global__ __launch_bounds__(512, 2) void gemm_b16kk_dma_kernel(GemmOperand16 A, GemmOperand16 B, FE fe, int R, int Cn, int K) {
  __shared__ __attribute__((aligned(1024))) unsigned short S[GKD_SMEM_HALFS];
  gemm_b16kk_dma_body<FE>(S, A, B, fe, R, Cn, K, blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y);
}
inline bool gemm_tile256(int R, int Cn);
inline int gemm_stag_default() {   // experiment switch (dbgopt.h): 2 LDS-DMA tiles + staggered wave groups, 1 register-staged + staggered, 0 the one-barrier loop
  return dbg_opt("gemm_stag", 2);
}
template <class FE>
inline void gemm_b16kk(hipStream_t stream, GemmOperand16 A, GemmOperand16 B, FE fe, int R, int Cn, int K, int stag = -1) {
  if (R <= 0 || Cn <= 0 || K <= 0) return;
  if (stag < 0) stag = gemm_stag_default();
  if (gemm_tile256(R, Cn)) {
    dim3 grid((Cn + 255) / 256, (R + 255) / 256, 1);
    if (stag == 2 && K % GB_BK == 0 && K >= 2 * GB_BK && (A.ld & 7) == 0 && (B.ld & 7) == 0) {   // operand tiles by LDS-DMA (16-byte aligned chunks)
      CLSTM_LAUNCH((gemm_b16kk_dma_kernel<FE>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K);
      return;
    }
#ifdef CLSTM_GEMM_PROF
    switch (stag >> 4) {   // (diagnostics: stag = 1 | leave-out bits << 4)
      case 1: CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 8, true, 1>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K); return;
      case 2: CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 8, true, 2>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K); return;
      case 4: CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 8, true, 4>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K); return;
      case 8: CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 8, true, 8>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K); return;
      case 12: CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 8, true, 12>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K); return;
      case 14: CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 8, true, 14>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K); return;
      default: break;
    }
    stag &= 1;
#endif
    if (stag) CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 8, true>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K);
    else CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 8, false>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K);
    return;
  }
  dim3 grid((Cn + GB2_BT - 1) / GB2_BT, (R + GB2_BT - 1) / GB2_BT, 1);
  CLSTM_LAUNCH((gemm_b16kk_kernel<FE, 4>), grid, dim3(256), 0, stream, A, B, fe, R, Cn, K);
}

// ---- 128 x 128 tile, both operands bf16 and CONTRACTION-major in memory (the weight-gradient product) -----------------
//   out(r, c) = sum_n A16[n * lda + r] * B16[n * ldb + c]          (n = frame-line index, 25600 of them at configs[4])
// The f32-source kernel above has to transpose such operands while it stages them (four ds_write_b64 per four float4,
// 2-way conflicts) after pulling twice the bytes through the vector cache.  Here the rows go from memory into LDS as
// they are -- thread t moves 16 bytes = 8 consecutive r of contraction row t >> 4 (and of row 16 + (t >> 4)), one
// ds_write_b128 each -- and the transposition is done by the LDS itself: ds_read_b64_tr_b16 hands lane i of a 16-lane
// group column i of a [4 n][16 r] block, which is exactly an MFMA fragment (4 of a lane's 8 contraction slots; a second
// read 16 rows further fills the other 4; A and B use the same slot -> n assignment, which is all the MFMA asks for).
// LDS image of an operand block: 8 strips of 16 columns, each [32 n][16] bf16 = 1 KB contiguous (a wave's fragment read
// covers 512 contiguous bytes: conflict-free by construction) + 32 bytes so that the 8 lanes of a ds_write_b128 group
// (4 strips x 2 halves of one row) fall into 8 distinct 16-byte bank slots.
struct GemmOperand16B { const unsigned short* p; int ld; long long elems; long long bstride; };   // halfs; ld % 8 == 0, bstride even
constexpr int GT_PF = 3;   // blocks in flight in registers (NB = 1; two with NB = 2: the same 128 rows ahead)
// WI = 16-row strips of a wave's tile: 4 -> 128 x 128 per workgroup (four waves 2 x 2, 64 x 64 each), 8 -> 256 x 256 (eight
// waves 2 x 4, 128 x 64 each: half the LDS and vector-cache bytes per MFMA; one workgroup per CU).
// NB = 32-row sub-blocks per barrier (contraction rows per LDS buffer = 32 NB).
// Tile: 32 WI rows x 64 NWN columns, waves 2 x NWN (a wave owns 16 WI x 64): WI = 4 -> 128 x 128, four waves; WI = 8 -> 256 x 256,
// WI = 6 -> 192 x 256 (round 5: 576 = 3 x 192 rows of the first configs[4] layer's weight gradient instead of three 256-row
// panels for 2.25), eight waves.
// A2 / a2_rows (optional): output rows r < a2_rows (a multiple of the tile height) take their A columns from a SECOND array
// shared by all batches -- the weight gradient's x-part rows straight from the bf16 outputs of the layer below instead of a
// copy of them inside the source rows (k_source_x_bf16: 30 us + 105 MB of traffic per configs[4] step for layer 2).
// As, Bs: the two double-buffered LDS images (GmcSmem); lin, gx, gy, gz: this workgroup's index in the gx x gy x gz grid.
template <class FE, int WI, int NB, bool STAG = false>   // (STAG: the staggered loop of gemm_b16kk_kernel, eight waves)
DEVFN void gemm_b16mc_body(unsigned short* const As, unsigned short* const Bs, GemmOperand16B A, GemmOperand16B B, FE fe, int R, int Cn, int K,
                           int ksplit, int nsplit, GemmOperand16B A2, int a2_rows, const unsigned lin, const unsigned gx, const unsigned gy, const unsigned gz) {
  constexpr int NWN = WI == 4 ? 2 : 4, NT = 128 * NWN, BTM = 32 * WI, BTN = 64 * NWN;
  constexpr int STRIP = NB * 512 + 16, TILE_A = (BTM / 16) * STRIP, TILE_B = (BTN / 16) * STRIP;
  constexpr int CH_A = BTM / 8, CH_B = BTN / 8;          // 16-byte chunks of a contraction row of the tile
  constexpr int BKB = 32 * NB, NL = 2 * NB, PF = NB == 1 ? GT_PF : 2;
  static_assert(16 * CH_A <= NT && 16 * CH_B == NT, "one thread per (chunk, contraction row mod 16)");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  int bx, by, z;   // XCD-aware tile order, see gemm_mfma.h
  {
    const unsigned total = gx * gy * gz;
    const unsigned xcd = lin & 7u, idx = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = (int)(v % gx);
    by = (int)((v / gx) % gy);
    z = (int)(v / (gx * gy));
  }
  const int r0 = by * BTM, c0 = bx * BTN;
  const int batch = z / nsplit;
  const int kbeg = (z - batch * nsplit) * ksplit;
  const int kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;
  // staging role: 16-byte chunk s?_c of contraction rows s?_k + 16 h of the block.  A tile 192 rows high has 24 chunks per row:
  // threads 384.. (waves 6, 7: wave-uniform) stage no A chunk.
  const int sa_c = tid % CH_A, sa_k = tid / CH_A, sb_c = tid % CH_B, sb_k = tid / CH_B;
  const bool a_on = wave_uniform(tid < 16 * CH_A ? 1 : 0) != 0;
  const bool use2 = wave_uniform(A2.p != nullptr && r0 < a2_rows ? 1 : 0) != 0;   // (a whole tile: a2_rows is a multiple of BTM)
  const unsigned short* const ap = use2 ? A2.p : A.p + batch * A.bstride;
  const long long ael = use2 ? A2.elems : A.elems - batch * A.bstride;
  const unsigned ald = (unsigned)(use2 ? A2.ld : A.ld);
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(ap), (size_t)ael * 2);
  const BufF32 bbuf = make_buf(reinterpret_cast<const float*>(B.p + batch * B.bstride), (size_t)(B.elems - batch * B.bstride) * 2);
  // columns past R / Cn read whatever follows in the row (or the next row): they only reach outputs that are not stored
  const unsigned aoff = ((unsigned)sa_k * ald + (unsigned)(r0 + sa_c * 8)) * 2u, a16 = 32u * ald;
  const unsigned boff = ((unsigned)sb_k * (unsigned)B.ld + (unsigned)(c0 + sb_c * 8)) * 2u, b16 = 32u * (unsigned)B.ld;
  const unsigned a_kstep = 2u * ald, b_kstep = 2u * (unsigned)B.ld;
  f32x4 ra[PF][NL], rb[PF][NL];
  // contraction rows past the slab load zeros: their offset is pushed out of the descriptor's range (one select per
  // load instead of one per staged element; nothing to mask when the block is staged)
  auto load_tile = [&](int k0, f32x4 (&a)[NL], f32x4 (&b)[NL]) {
    const unsigned kc = (unsigned)wave_uniform(k0);
#pragma unroll
    for (int h = 0; h < NL; h++) {
      const bool lv = a_on && k0 + sa_k + 16 * h < kend;
      a[h] = buf_load4(abuf, lv ? aoff + kc * a_kstep + (unsigned)h * a16 : BUF_OOB);
    }
#pragma unroll
    for (int h = 0; h < NL; h++) {
      const bool lv = k0 + sb_k + 16 * h < kend;
      b[h] = buf_load4(bbuf, lv ? boff + kc * b_kstep + (unsigned)h * b16 : BUF_OOB);
    }
  };
  const int sa_at = (sa_c >> 1) * STRIP + sa_k * 16 + (sa_c & 1) * 8, sb_at = (sb_c >> 1) * STRIP + sb_k * 16 + (sb_c & 1) * 8;
  auto stage = [&](unsigned short* Sa, unsigned short* Sb, const f32x4 (&a)[NL], const f32x4 (&b)[NL]) {
    if (a_on) {
#pragma unroll
      for (int h = 0; h < NL; h++) *reinterpret_cast<f32x4*>(&Sa[sa_at + h * 256]) = a[h];
    }
#pragma unroll
    for (int h = 0; h < NL; h++) *reinterpret_cast<f32x4*>(&Sb[sb_at + h * 256]) = b[h];
  };
  f32x4 acc[WI][4];
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;
  const int f_at = lane * 4;   // lane l of a group points at chunk l of the group's [4 n][16] block: rows 4 (l >> 4) .. + 3
  auto mfma_sub = [&](const int abuf_at, const int bbuf_at) {   // one 32-row sub-block: fragments by transpose reads, WI x 4 MFMAs
    u16x8 bf[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned short* bp = &Bs[bbuf_at + (wn * 4 + j) * STRIP + f_at];
      bf[j] = join_u16x8(lds_read_tr16(bp), lds_read_tr16(bp + 256));
    }
    if constexpr (STAG) {   // every fragment of the block first, a barrier, then the MFMAs alone (the other group's LDS phase runs beside them)
      u16x8 af[WI];
#pragma unroll
      for (int i = 0; i < WI; i++) {
        const unsigned short* ap = &As[abuf_at + (wm * WI + i) * STRIP + f_at];
        af[i] = join_u16x8(lds_read_tr16(ap), lds_read_tr16(ap + 256));
      }
      SCHED_FENCE();   // (pin the MFMAs between the two barriers: see gemm_b16kk_kernel)
      __syncthreads();
      SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < WI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = mfma16x16x32_bf16(bf[j], af[i], acc[i][j]);
      SCHED_FENCE();
      return;
    }
    constexpr int AG = WI % 4 == 0 ? 4 : 3;   // A strips at a time: 8 AG fragment registers live, not 16 + 4 WI
#pragma unroll
    for (int i0 = 0; i0 < WI; i0 += AG) {
      u16x8 af[AG];
#pragma unroll
      for (int i = 0; i < AG; i++) {
        const unsigned short* ap = &As[abuf_at + (wm * WI + i0 + i) * STRIP + f_at];
        af[i] = join_u16x8(lds_read_tr16(ap), lds_read_tr16(ap + 256));
      }
#pragma unroll
      for (int i = 0; i < AG; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i0 + i][j] = mfma16x16x32_bf16(bf[j], af[i], acc[i0 + i][j]);   // transposed: see gb2_store
      if (NB > 1) SCHED_FENCE();
    }
  };
#pragma unroll
  for (int p = 0; p < PF; p++) {
    load_tile(kbeg + p * BKB, ra[p], rb[p]);
    SCHED_FENCE();
  }
  stage(As, Bs, ra[0], rb[0]);
  load_tile(kbeg + PF * BKB, ra[0], rb[0]);
  SCHED_FENCE();
  __syncthreads();
  static_assert(!STAG || (WI >= 6 && NB == 1), "the staggered loop pairs the two waves of each SIMD of an eight-wave workgroup");
  int cur = 0;   // 0 / 1: the LDS buffer pair that holds the block the MFMAs are about to consume
  if (STAG && wm == 1) __syncthreads();   // group B (waves 4-7) runs one barrier behind group A: see gemm_b16kk_kernel
  for (int kb = kbeg; kb < kend; kb += PF * BKB) {
#pragma unroll
    for (int p = 0; p < PF; p++) {
      const int k0 = kb + p * BKB;
      const int pn = p == PF - 1 ? 0 : p + 1;
      stage(As + (cur ^ 1) * TILE_A, Bs + (cur ^ 1) * TILE_B, ra[pn], rb[pn]);
      load_tile(k0 + BKB + PF * BKB, ra[pn], rb[pn]);
      SCHED_FENCE();
#pragma unroll
      for (int sb = 0; sb < NB; sb++) mfma_sub(cur * TILE_A + sb * 512, cur * TILE_B + sb * 512);
      __syncthreads();
      if (STAG) SCHED_FENCE();
      cur ^= 1;
    }
  }
  if (STAG && wm == 0) __syncthreads();
  gb2_store<FE, WI>(fe, acc, r0 + wm * (16 * WI), c0 + wn * 64, lane, R, Cn, z);
}
template <int WI, int NB> struct GmcSmem {   // halfs of the two double-buffered operand images of gemm_b16mc_body
  static constexpr int STRIP = NB * 512 + 16, A = 2 * (2 * WI) * STRIP, B = 2 * ((WI == 4 ? 128 : 256) / 16) * STRIP;
};
template <class FE, int WI, int NB, bool STAG = false>
__global__ __launch_bounds__(WI == 4 ? 256 : 512, 2) void gemm_b16mc_kernel(GemmOperand16B A, GemmOperand16B B, FE fe, int R, int Cn, int K,
                                                                            int ksplit, int nsplit, GemmOperand16B A2, int a2_rows) {
  __shared__ __attribute__((aligned(16))) unsigned short As[GmcSmem<WI, NB>::A];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[GmcSmem<WI, NB>::B];
  gemm_b16mc_body<FE, WI, NB, STAG>(As, Bs, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows,
                                    blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y, gridDim.z);
}
// ---- the contraction-major product with its tiles brought in by LDS-DMA (round 5, third image; see the note below) --------------
// LDS image of an operand block [32 n][256 columns]: sixteen UNITS of 1 KB, unit (ch, q) = contraction rows 4 q .. 4 q + 3 x columns
// 128 ch .. 128 ch + 127 as eight [4 n][16 columns] blocks of 128 bytes -- the block a 16-lane group of ds_read_b64_tr_b16 reads
// (contiguous: one LDS pass per group, like the strips of gemm_b16mc_kernel) -- with block b at position (b + (q & 1)) & 7, so
// that the two lane groups a read services together (n-quads q, q + 1) fall into different halves of the 256-byte bank window.  A
// DMA instruction fills one unit: lane l -> position l >> 3, row (l >> 1) & 3, column half l & 1, i.e. four contraction rows x 256
// contiguous bytes of memory each (the k-contiguous kernel's requests are 64-byte runs).  Two units per wave, operand and block.
template <class FE, int WI>
DEVFN void gemm_b16mc_dma_body(unsigned short* const S, GemmOperand16B A, GemmOperand16B B, FE fe, int R, int Cn, int K, int ksplit, int nsplit,
                               GemmOperand16B A2, int a2_rows, const unsigned lin, const unsigned gx, const unsigned gy, const unsigned gz) {
  static_assert(WI == 8 || WI == 6, "256- or 192-row tiles, eight waves");
  constexpr int BTM = 32 * WI, UNIT = 512, IMG = 16 * UNIT, BUF = 2 * IMG, NBUF = 3;   // halfs: a unit, an operand image, A | B
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
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
  const int r0 = by * BTM, c0 = bx * 256;
  const int batch = z / nsplit;
  const int kbeg = (z - batch * nsplit) * ksplit;
  const int kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;
  const bool use2 = wave_uniform(A2.p != nullptr && r0 < a2_rows ? 1 : 0) != 0;
  const unsigned short* const ap = use2 ? A2.p : A.p + batch * A.bstride;
  const long long ael = use2 ? A2.elems : A.elems - batch * A.bstride;
  const unsigned ald = (unsigned)(use2 ? A2.ld : A.ld);
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(ap), (size_t)ael * 2);
  const BufF32 bbuf = make_buf(reinterpret_cast<const float*>(B.p + batch * B.bstride), (size_t)(B.elems - batch * B.bstride) * 2);
  // DMA role: units u = 2 wave + s (s = 0, 1) of each operand: ch = u >> 3, q = u & 7
  int dn[2];
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int sl = 0; sl < 2; sl++) {
    const int u = 2 * wave + sl, ch = u >> 3, q = u & 7;
    const int blk = ((lane >> 3) - (q & 1)) & 7;              // the block this lane's position holds
    dn[sl] = 4 * q + ((lane >> 1) & 3);
    const unsigned col = (unsigned)(128 * ch + 16 * blk + 8 * (lane & 1));
    aoff[sl] = ((unsigned)dn[sl] * ald + (unsigned)r0 + col) * 2u;
    boff[sl] = ((unsigned)dn[sl] * (unsigned)B.ld + (unsigned)c0 + col) * 2u;
  }
  const unsigned a_blk = 64u * ald, b_blk = 64u * (unsigned)B.ld;   // bytes per 32 contraction rows
  const int nblk = (kend - kbeg + GB_BK - 1) / GB_BK;
  auto dma = [&](const int blk, const int buf) {   // (requests past the slab read rows >= kend: zeros, nobody reads them)
    const int k0 = kbeg + wave_uniform(blk) * GB_BK;
    const unsigned kb = (unsigned)wave_uniform(kbeg / GB_BK + blk);   // kbeg is a multiple of 32 (ksplit is)
    unsigned short* const ia = S + buf * BUF, * const ib = ia + IMG;
#pragma unroll
    for (int sl = 0; sl < 2; sl++) lds_dma16(abuf, k0 + dn[sl] < kend ? aoff[sl] + kb * a_blk : BUF_OOB, ia + (2 * wave + sl) * UNIT);
#pragma unroll
    for (int sl = 0; sl < 2; sl++) lds_dma16(bbuf, k0 + dn[sl] < kend ? boff[sl] + kb * b_blk : BUF_OOB, ib + (2 * wave + sl) * UNIT);
  };
  f32x4 acc[WI][4];
#pragma unroll
  for (int i = 0; i < WI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;
  // fragment role: lane group g = lane >> 4 reads n-quad g (first read) / 4 + g (second) of strip j: unit (j >> 3, quad),
  // position ((j & 7) + (g & 1)) & 7, piece lane & 15
  // (ds_read_b64_tr_b16 as inline asm: beside pending DMA the builtin costs an s_waitcnt vmcnt(0) per block, devintrin.h)
  const int fg = lane >> 4;
  const int f_base = fg * UNIT + (lane & 15) * 4;            // halfs; + (j >> 3) * 8 UNIT + position * 64; second read + 4 UNIT
  auto frag = [&](const unsigned short* img, const int j) -> u16x8 {
    const LdsAddr p = lds_addr(img + f_base + (j >> 3) * (8 * UNIT) + ((((j & 7) + (fg & 1)) & 7) << 6));
    return join_u16x8(lds_read_tr16_raw<0>(p), lds_read_tr16_raw<4 * UNIT * 2>(p));
  };
  dma(0, 0);
  dma(1, 1);
  wait_vmcnt<4>();
  wg_barrier();
  if (wm == 1) wg_barrier();
  int cur = 0, nxt2 = 2;
  for (int t = 0; t < nblk; t++) {
    dma(t + 2, nxt2);
    SCHED_FENCE();
    const unsigned short* const ia = S + cur * BUF;
    const unsigned short* const ib = ia + IMG;
    u16x8 af[WI], bf[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bf[j] = frag(ib, wn * 4 + j);
#pragma unroll
    for (int i = 0; i < WI; i++) af[i] = frag(ia, wm * WI + i);
    SCHED_FENCE();
    if (wm == 1) wait_vmcnt<4>();
    wait_lgkmcnt0();
    SCHED_FENCE();
    wg_barrier();
    SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < WI; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = mfma16x16x32_bf16(bf[j], af[i], acc[i][j]);
    SCHED_FENCE();
    if (wm == 0) wait_vmcnt<4>();
    wg_barrier();
    SCHED_FENCE();
    cur = cur == NBUF - 1 ? 0 : cur + 1;
    nxt2 = nxt2 == NBUF - 1 ? 0 : nxt2 + 1;
  }
  if (wm == 0) wg_barrier();
  wait_vmcnt<0>();
  if (fe.vec4()) {   // (whole-line stores through LDS: gb2_store_lds)
    wg_barrier();
    gb2_store_lds<FE, WI>(reinterpret_cast<float*>(S) + wave * GKD_EPI_FLOATS, fe, acc, r0 + wm * (16 * WI), c0 + wn * 64, lane, R, Cn, z);
  } else gb2_store<FE, WI>(fe, acc, r0 + wm * (16 * WI), c0 + wn * 64, lane, R, Cn, z);
}
constexpr int GMD_SMEM_HALFS = 3 * 2 * 16 * 512;   // 96 KB
template <class FE, int WI>
__global__ __launch_bounds__(512, 2) void gemm_b16mc_dma_kernel(GemmOperand16B A, GemmOperand16B B, FE fe, int R, int Cn, int K,
                                                                int ksplit, int nsplit, GemmOperand16B A2, int a2_rows) {
  __shared__ __attribute__((aligned(1024))) unsigned short S[GMD_SMEM_HALFS];
  gemm_b16mc_dma_body<FE, WI>(S, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                              gridDim.x, gridDim.y, gridDim.z);
}
// ---- the weight gradient and the input deltas of a wide layer as ONE launch (round 5) --------------------------------------------
// Both depend on the layer's backward recurrence and on nothing else; apart, each leaves a quarter of the chip idle: the
// weight-gradient product of the upper configs[4] layer is 96 tiles x 2 slabs = 192 workgroups of 400 blocks on 256 CUs, x.d
// 400 tiles of 128 blocks = two rounds, the second 56 % full (656 block-times end to end).  One grid -- the long weight-gradient
// workgroups first, the x.d tiles behind them, taken by whichever CU comes free -- needs ~500 + the tail.  Roles by block index
// (wave-uniform); ONE LDS array serves both bodies.
template <class FEW, int WI, class FEX, bool WDMA>   // WDMA: the weight-gradient role's tiles by LDS-DMA too (gemm_b16mc_dma_body)
__global__ __launch_bounds__(512, 2) void gemm_dw_dx_kernel(GemmOperand16B A, GemmOperand16B B, FEW few, int R, int Cn, int K, int ksplit, int nsplit,
                                                            GemmOperand16B A2, int a2_rows, unsigned gxw, unsigned gyw, unsigned gzw,
                                                            GemmOperand16 XA, GemmOperand16 XB, FEX fex, int XR, int XCn, int XK, unsigned gxx, unsigned gyx) {
  constexpr int MC_HALFS = GmcSmem<WI, 1>::A + GmcSmem<WI, 1>::B;
  static_assert(GMD_SMEM_HALFS == GKD_SMEM_HALFS, "one 96 KB array serves both DMA bodies");
  __shared__ __attribute__((aligned(1024))) unsigned short S[MC_HALFS > GKD_SMEM_HALFS ? MC_HALFS : GKD_SMEM_HALFS];
  const unsigned nw = gxw * gyw * gzw;
  if (blockIdx.x < nw) {
    if constexpr (WDMA) gemm_b16mc_dma_body<FEW, WI>(S, A, B, few, R, Cn, K, ksplit, nsplit, A2, a2_rows, blockIdx.x, gxw, gyw, gzw);
    else gemm_b16mc_body<FEW, WI, 1, true>(S, S + GmcSmem<WI, 1>::A, A, B, few, R, Cn, K, ksplit, nsplit, A2, a2_rows, blockIdx.x, gxw, gyw, gzw);
  } else gemm_b16kk_dma_body<FEX>(S, XA, XB, fex, XR, XCn, XK, blockIdx.x - nw, gxx, gyx);
}
// (Round 5, on the way to gemm_b16mc_dma_kernel: its first three forms -- the strips of gemm_b16mc_kernel filled one DMA instruction
// per strip, a row-major [32 n][256] image with XOR-permuted chunks, and the unit image above -- all ran 1.9x SLOWER than the
// register-staged loop (522 / 481 / 533 vs 281 us at 1544 x 2048 x 25600, two slabs), whatever the image: hipcc had put
// s_waitcnt vmcnt(0) in front of the first ds_read_b64_tr_b16 of every block -- it cannot see which LDS bytes the BUILTIN reads
// and takes every pending DMA for its producer -- so the request pipeline drained at every block.  With the reads as inline asm
// (devintrin.h:lds_read_tr16_raw) the unit image runs 240 us.  profiles/r05_gemm_variants.txt.)
// 256 x 256 tiles where the problem is large enough and their padding costs at most 25 % more work than 128 x 128 tiles do:
// the big tile runs ~1.4x faster per flop (1537 rows: 7 x 256 vs 13 x 128, + 8 %; 561 rows: 3 x 256 vs 5 x 128, + 20 %:
// 98 vs 116 us for one direction of the first layer's weight gradient)
inline bool gemm_tile256(int R, int Cn) {
  if (R < 192 || Cn < 192) return false;
  const long long w256 = (long long)((R + 255) / 256) * ((Cn + 255) / 256) * 4, w128 = (long long)((R + 127) / 128) * ((Cn + 127) / 128);
  return 4 * w256 <= 5 * w128;
}
// tile height of the big-tile contraction-major product: 256 rows, or 192 where that pads R less (576 = 3 x 192)
inline int gemm_mc_tile_rows(int R) {
  const int p256 = (R + 255) / 256 * 256, p192 = (R + 191) / 192 * 192;
  return p192 < p256 ? 192 : 256;
}
// rows of one output tile of gemm_b16mc / gemm_dw_dx for an R x Cn product: an EXTERNAL x block (operand A2: rows that are not
// copied into the source array but read from the layer below's bf16 outputs) must fill whole tiles.  The forward pass decides
// "external" with this same function for every R the backward pass may launch with (R, and R - 1 when the bias row is left out),
// and the launchers REFUSE an A2 that does not fit instead of dropping it (the x rows would then be read from columns nobody wrote).
inline int gemm_mc_rows_per_tile(int R, int Cn) { return gemm_tile256(R, Cn) ? gemm_mc_tile_rows(R) : 128; }
inline void gemm_mc_check_a2(const void* a2, int a2_rows, int th) {
  if (a2 && a2_rows % th != 0) throw std::runtime_error("internal: external x rows of the weight-gradient product do not fill whole row tiles");
}
template <class FE>
inline void gemm_b16mc(hipStream_t stream, GemmOperand16B A, GemmOperand16B B, FE fe, int R, int Cn, int K, int nsplit = 1, int nbatch = 1,
                       GemmOperand16B A2 = GemmOperand16B{nullptr, 0, 0, 0}, int a2_rows = 0, int stag = -1) {
  if (R <= 0 || Cn <= 0 || K <= 0) return;
  if (nsplit < 1) nsplit = 1;
  if (stag < 0) stag = gemm_stag_default();
  // (64 contraction rows per barrier on the 256 x 256 tile -- 231 VGPRs, 130 KB LDS -- measured equal to the 32-row
  // loop, 227 vs 226 us at 1544 x 2048 x 25600, and is gone)
  const bool big = gemm_tile256(R, Cn);
  int ksplit = (K + nsplit - 1) / nsplit;
  const int kq = nsplit > 1 ? GT_PF * GB_BK : GB_BK;   // whole ring rounds per slab
  ksplit = ((ksplit + kq - 1) / kq) * kq;
  if (big) {
    const int th = gemm_mc_tile_rows(R);
    dim3 grid((Cn + 255) / 256, (R + th - 1) / th, nsplit * nbatch);
    gemm_mc_check_a2(A2.p, a2_rows, th);
    const bool dma_ok = stag >= 2 && (A.ld & 7) == 0 && (B.ld & 7) == 0 && (A.bstride & 7) == 0 && (B.bstride & 7) == 0 && (!A2.p || (A2.ld & 7) == 0) &&
                        ksplit >= 2 * GB_BK && ksplit % GB_BK == 0 && (((size_t)A.p | (size_t)B.p | (size_t)A2.p) & 15) == 0;
    if (dma_ok) {   // operand tiles by LDS-DMA (16-byte aligned chunks, slabs of whole blocks)
      if (th == 192) CLSTM_LAUNCH((gemm_b16mc_dma_kernel<FE, 6>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows);
      else CLSTM_LAUNCH((gemm_b16mc_dma_kernel<FE, 8>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows);
      return;
    }
    if (th == 192) {
      if (stag) CLSTM_LAUNCH((gemm_b16mc_kernel<FE, 6, 1, true>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows);
      else CLSTM_LAUNCH((gemm_b16mc_kernel<FE, 6, 1, false>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows);
    } else if (stag) CLSTM_LAUNCH((gemm_b16mc_kernel<FE, 8, 1, true>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows);
    else CLSTM_LAUNCH((gemm_b16mc_kernel<FE, 8, 1, false>), grid, dim3(512), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows);
    return;
  }
  dim3 grid((Cn + GB2_BT - 1) / GB2_BT, (R + GB2_BT - 1) / GB2_BT, nsplit * nbatch);
  gemm_mc_check_a2(A2.p, a2_rows, GB2_BT);
  CLSTM_LAUNCH((gemm_b16mc_kernel<FE, 4, 1>), grid, dim3(256), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit, A2, a2_rows);
}

// weight gradient (contraction-major, big tiles) + input deltas (k-contiguous, LDS-DMA) as one launch; false: not eligible, nothing launched
template <class FEW, class FEX>
inline bool gemm_dw_dx(hipStream_t stream, GemmOperand16B A, GemmOperand16B B, FEW few, int R, int Cn, int K, int nsplit, int nbatch, GemmOperand16B A2, int a2_rows,
                       GemmOperand16 XA, GemmOperand16 XB, FEX fex, int XR, int XCn, int XK) {
  if (gemm_stag_default() != 2 || !gemm_tile256(R, Cn) || !gemm_tile256(XR, XCn)) return false;
  if (XK % GB_BK != 0 || XK < 2 * GB_BK || (XA.ld & 7) != 0 || (XB.ld & 7) != 0 || R <= 0 || Cn <= 0 || K <= 0) return false;
  if ((((size_t)XA.p | (size_t)XB.p) & 15) != 0) return false;   // the x.d role stages by LDS-DMA: 16-byte aligned operands only
  if (nsplit < 1) nsplit = 1;
  int ksplit = (K + nsplit - 1) / nsplit;
  const int kq = nsplit > 1 ? GT_PF * GB_BK : GB_BK;
  ksplit = ((ksplit + kq - 1) / kq) * kq;
  const int th = gemm_mc_tile_rows(R);
  gemm_mc_check_a2(A2.p, a2_rows, th);
  const unsigned gxw = (Cn + 255) / 256, gyw = (R + th - 1) / th, gzw = nsplit * nbatch, gxx = (XCn + 255) / 256, gyx = (XR + 255) / 256;
  const dim3 grid(gxw * gyw * gzw + gxx * gyx);
  const bool wdma = (A.ld & 7) == 0 && (B.ld & 7) == 0 && (A.bstride & 7) == 0 && (B.bstride & 7) == 0 && (!A2.p || (A2.ld & 7) == 0) &&
                    ksplit >= 2 * GB_BK && ksplit % GB_BK == 0 && (((size_t)A.p | (size_t)B.p | (size_t)A2.p) & 15) == 0;
#define CLSTM_DWDX(WI_, WD_) CLSTM_LAUNCH((gemm_dw_dx_kernel<FEW, WI_, FEX, WD_>), grid, dim3(512), 0, stream, A, B, few, R, Cn, K, ksplit, nsplit, A2, a2_rows, \
                                          gxw, gyw, gzw, XA, XB, fex, XR, XCn, XK, gxx, gyx)
  if (th == 192) { if (wdma) CLSTM_DWDX(6, true); else CLSTM_DWDX(6, false); }
  else { if (wdma) CLSTM_DWDX(8, true); else CLSTM_DWDX(8, false); }
#undef CLSTM_DWDX
  return true;
}

// ---- f32-grade products on the bf16 MFMA: 64 x 64 tile, operands split hi + lo ------------------------------------------
// Same interface as gemm_f32_body (gemm_mfma.h) and -- to ~1e-6 of a product, see gemm_dw.h -- the same result:
// every f32 operand element is staged as hi = bf16(x) and lo = bf16(x - hi), a product is hi.hi + hi.lo + lo.hi on
// v_mfma_f32_16x16x32_bf16 with f32 accumulation.  Leaving parts out of the f32 kernel showed the MFMA pipe as the
// largest single item of the short products of a training step (softmax W.d / x.d pair 9 of 24 us, W_x 5 of 21, fused
// softmax 6 of 21): a 32-k block costs a wave 12 MFMAs of 16 cycles here instead of 32 of 32.
// Four swizzled [mn][32 k] images (A hi, A lo, B hi, B lo; gb2_sw) per buffer, two buffers, one barrier per block;
// every thread stages 8 elements of each operand (KC: 8 consecutive k of a row -> one ds_write_b128 per image;
// MC: 4 columns x (k, k+1) -> four ds_write_b32 per image); ring of three register-staged blocks.
// NT = 3: three bf16 terms per operand (x1 + x2 + x3 represents an f32 exactly), six products -- the operand-exact form of
// gemm_dw.h, the default of the narrow layers' backward products; images [A terms | B terms] per buffer.
constexpr int GX3_IMG = 64 * 32;                 // halfs per image
constexpr int gx3_smem_floats(int nt) { return 2 * 2 * nt * GX3_IMG / 2; }   // 32 KB (NT = 2) / 48 KB (NT = 3)
static_assert(gx3_smem_floats(2) >= GEMM_BT * GEMM_LDO, "epilogue tile");

template <int AMODE, int BMODE, class FE, int NT = 2>
DEVFN void gemm_x3_body(float* smem, GemmOperand A, GemmOperand B, FE fe, int R, int Cn, int K, int ksplit, int nsplit,
                        const unsigned lin, const unsigned gx, const unsigned gy, const unsigned gz) {
  unsigned short* img = reinterpret_cast<unsigned short*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int bx, by, z;   // XCD-aware tile order, see gemm_mfma.h
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

  // staging unit of this thread (two float4 per operand and block):
  //   KC: row tid>>2, k = (tid&3)*8 .. +7      MC: mn = (tid&15)*4 .. +3, k = (tid>>4)*2, +1
  const int a_mn = AMODE == GEMM_KC ? (tid >> 2) : (tid & 15) * 4, a_k = AMODE == GEMM_KC ? (tid & 3) * 8 : (tid >> 4) * 2;
  const int b_mn = BMODE == GEMM_KC ? (tid >> 2) : (tid & 15) * 4, b_k = BMODE == GEMM_KC ? (tid & 3) * 8 : (tid >> 4) * 2;
  const BufF32 abuf = make_buf(A.p + batch * A.bstride, (size_t)(A.elems - batch * A.bstride) * 4);
  const BufF32 bbuf = make_buf(B.p + batch * B.bstride, (size_t)(B.elems - batch * B.bstride) * 4);
  const unsigned a_base = AMODE == GEMM_KC ? (unsigned)(r0 + a_mn) * A.ld + a_k : (unsigned)a_k * A.ld + r0 + a_mn;
  const unsigned b_base = BMODE == GEMM_KC ? (unsigned)(c0 + b_mn) * B.ld + b_k : (unsigned)b_k * B.ld + c0 + b_mn;
  const unsigned a_kstep = AMODE == GEMM_KC ? 1u : (unsigned)A.ld, b_kstep = BMODE == GEMM_KC ? 1u : (unsigned)B.ld;
  const unsigned a_second = AMODE == GEMM_KC ? 4u : (unsigned)A.ld, b_second = BMODE == GEMM_KC ? 4u : (unsigned)B.ld;
  const unsigned aoff0 = a_base * 4u, aoff1 = (a_base + a_second) * 4u, boff0 = b_base * 4u, boff1 = (b_base + b_second) * 4u;
  const int klast = kbeg + ((kend - kbeg - 1) / GB_BK) * GB_BK;   // blocks past the slab re-read its last block (zeroed when staged)

  f32x4 ra[GB_PF][2], rb[GB_PF][2];
  auto load_tile = [&](int k0, f32x4 (&a)[2], f32x4 (&b)[2]) {
    const unsigned kc = (unsigned)wave_uniform(k0 < klast ? k0 : klast);
    const unsigned ao = kc * a_kstep * 4u, bo = kc * b_kstep * 4u;
    a[0] = buf_load4(abuf, aoff0 + ao);
    a[1] = buf_load4(abuf, aoff1 + ao);
    b[0] = buf_load4(bbuf, boff0 + bo);
    b[1] = buf_load4(bbuf, boff1 + bo);
  };
  // split + store; contraction indices >= kend are zeroed (only the slab's last block and the zero blocks behind it)
  constexpr int BUFH = 2 * NT * GX3_IMG;   // halfs per buffer
  auto stage = [&](const int MODE, unsigned short* hi, const int mn, const int kk, const int k0, const f32x4 (&r)[2]) {
    const bool whole = wave_uniform(k0 + GB_BK <= kend ? 1 : 0) != 0;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = r[i >> 2][i & 3];
    if (!whole) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int k = MODE == GEMM_KC ? kk + i : kk + (i >> 2);
        x[i] = k0 + k < kend ? x[i] : 0.0f;
      }
    }
    if (MODE == GEMM_KC) {
      const int o = mn * 32 + (((kk >> 3) ^ gb2_sw(mn)) << 3);
#pragma unroll
      for (int t = 0; t < NT; t++) {   // term t, then the exact remainder
        const u16x8 h = bf16_pack8(x);
        *reinterpret_cast<u16x8*>(hi + t * GX3_IMG + o) = h;
        if (t + 1 < NT) {
#pragma unroll
          for (int i = 0; i < 8; i++) x[i] -= __builtin_bit_cast(float, (unsigned)h[i] << 16);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {   // column mn + i: the pair (k, k+1)
        float e0 = x[i], e1 = x[4 + i];
        const int o = (mn + i) * 32 + ((((kk >> 3) ^ gb2_sw(mn + i)) << 3) | (kk & 7));
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const unsigned h = bf16_pack2(e0, e1);
          *reinterpret_cast<unsigned*>(hi + t * GX3_IMG + o) = h;
          if (t + 1 < NT) { e0 -= __builtin_bit_cast(float, h << 16); e1 -= __builtin_bit_cast(float, h & 0xffff0000u); }
        }
      }
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;

#pragma unroll
  for (int p = 0; p < GB_PF; p++) {
    load_tile(kbeg + p * GB_BK, ra[p], rb[p]);
    SCHED_FENCE();
  }
  stage(AMODE, img, a_mn, a_k, kbeg, ra[0]);
  stage(BMODE, img + NT * GX3_IMG, b_mn, b_k, kbeg, rb[0]);
  load_tile(kbeg + GB_PF * GB_BK, ra[0], rb[0]);
  SCHED_FENCE();
  __syncthreads();
  const int fk = lane >> 4, fi = lane & 15;
  const int fofs = fi * 32 + ((fk ^ gb2_sw(fi)) << 3);
  int cur = 0;
  for (int kb = kbeg; kb < kend; kb += GB_PF * GB_BK) {
#pragma unroll
    for (int p = 0; p < GB_PF; p++) {
      static_assert(GB_PF == 3, "register ring of three blocks");
      const int k0 = kb + p * GB_BK;
      const int pn = p == 2 ? 0 : p + 1;
      unsigned short* nb = img + (cur ^ 1) * BUFH;
      stage(AMODE, nb, a_mn, a_k, k0 + GB_BK, ra[pn]);
      stage(BMODE, nb + NT * GX3_IMG, b_mn, b_k, k0 + GB_BK, rb[pn]);
      load_tile(k0 + GB_BK + GB_PF * GB_BK, ra[pn], rb[pn]);
      SCHED_FENCE();
      const unsigned short* b0 = img + cur * BUFH;
      u16x8 at[NT][2], bt[NT][2];
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
          at[t][i] = *reinterpret_cast<const u16x8*>(b0 + t * GX3_IMG + (wm * 32 + i * 16) * 32 + fofs);
          bt[t][i] = *reinterpret_cast<const u16x8*>(b0 + (NT + t) * GX3_IMG + (wn * 32 + i * 16) * 32 + fofs);
        }
      // smallest terms first: ta + tb = NT - 1 .. 0 (NT = 2: lo.hi, hi.lo, hi.hi)
#pragma unroll
      for (int w = NT - 1; w >= 0; w--)
#pragma unroll
        for (int ta = w; ta >= 0; ta--)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = mfma16x16x32_bf16(at[ta][i], bt[w - ta][j], acc[i][j]);
      __syncthreads();
      cur ^= 1;
    }
  }
  if (fe.vec4()) {   // epilogue through LDS, as gemm_f32_body
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

template <int AMODE, int BMODE, class FE, int NT = 2>
__global__ __launch_bounds__(256) void gemm_x3_kernel(GemmOperand A, GemmOperand B, FE fe, int R, int Cn, int K, int ksplit,
                                                      int nsplit) {
  __shared__ __attribute__((aligned(16))) float smem[gx3_smem_floats(NT)];
  gemm_x3_body<AMODE, BMODE, FE, NT>(smem, A, B, fe, R, Cn, K, ksplit, nsplit,
                                     blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y, gridDim.z);
}
template <int A1, int B1, class FE1, int A2, int B2, class FE2, int NT = 2>
__global__ __launch_bounds__(256) void gemm_x3_pair_kernel(GemmProblem p1, FE1 fe1, GemmProblem p2, FE2 fe2, unsigned nb1) {
  __shared__ __attribute__((aligned(16))) float smem[gx3_smem_floats(NT)];
  if (blockIdx.x < nb1)
    gemm_x3_body<A1, B1, FE1, NT>(smem, p1.A, p1.B, fe1, p1.R, p1.Cn, p1.K, p1.ksplit, p1.nsplit, blockIdx.x, p1.gx, p1.gy, p1.gz);
  else
    gemm_x3_body<A2, B2, FE2, NT>(smem, p2.A, p2.B, fe2, p2.R, p2.Cn, p2.K, p2.ksplit, p2.nsplit, blockIdx.x - nb1, p2.gx, p2.gy, p2.gz);
}
// launchers with the signatures of gemm_f32 / gemm_f32_pair (KC operands are read 8 floats at a time: same slack rule
// as gemm_bf16)
// `terms`: 2 (hi + lo, three products) or 3 (operand-exact, six products)
template <int AMODE, int BMODE, class FE>
inline void gemm_x3(hipStream_t stream, GemmOperand A, GemmOperand B, FE fe, int R, int Cn, int K, int nsplit = 1, int nbatch = 1,
                    int terms = 2) {
  if (R <= 0 || Cn <= 0) return;
  const GemmProblem p = gemm_problem(A, B, R, Cn, K, nsplit, nbatch);
  if (terms >= 3)
    CLSTM_LAUNCH((gemm_x3_kernel<AMODE, BMODE, FE, 3>), dim3(p.gx, p.gy, p.gz), dim3(256), 0, stream, A, B, fe, R, Cn, K, p.ksplit, p.nsplit);
  else
    CLSTM_LAUNCH((gemm_x3_kernel<AMODE, BMODE, FE, 2>), dim3(p.gx, p.gy, p.gz), dim3(256), 0, stream, A, B, fe, R, Cn, K, p.ksplit, p.nsplit);
}
template <int A1, int B1, class FE1, int A2, int B2, class FE2>
inline void gemm_x3_pair(hipStream_t stream, GemmProblem p1, FE1 fe1, GemmProblem p2, FE2 fe2, int terms = 2) {
  const unsigned nb1 = p1.gx * p1.gy * p1.gz, nb2 = p2.gx * p2.gy * p2.gz;
  if (terms >= 3)
    CLSTM_LAUNCH((gemm_x3_pair_kernel<A1, B1, FE1, A2, B2, FE2, 3>), dim3(nb1 + nb2), dim3(256), 0, stream, p1, fe1, p2, fe2, nb1);
  else
    CLSTM_LAUNCH((gemm_x3_pair_kernel<A1, B1, FE1, A2, B2, FE2, 2>), dim3(nb1 + nb2), dim3(256), 0, stream, p1, fe1, p2, fe2, nb1);
}

// shapes that fill 128 x 128 tiles reasonably
inline bool gemm_bf16_big(int R, int Cn) { return R >= 96 && Cn >= 96; }
// f32-grade product on 128 x 128 tiles (gemm_x3_128_kernel); same signature as gemm_f32 / gemm_bf16
template <int AMODE, int BMODE, class FE>
inline void gemm_x3_big(hipStream_t stream, GemmOperand A, GemmOperand B, FE fe, int R, int Cn, int K, int nsplit = 1, int nbatch = 1) {
  if (R <= 0 || Cn <= 0 || K <= 0) return;
  if (nsplit < 1) nsplit = 1;
  int ksplit = (K + nsplit - 1) / nsplit;
  const int kq = nsplit > 1 ? GB2_PF * GB_BK : GB_BK;   // whole ring rounds per slab
  ksplit = ((ksplit + kq - 1) / kq) * kq;
  const size_t smem = (size_t)8 * GB2_TILE * sizeof(unsigned short);   // 64 KB
#ifndef CLSTM_HIP_EMU
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_x3_128_kernel<AMODE, BMODE, FE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
#endif
  dim3 grid((Cn + GB2_BT - 1) / GB2_BT, (R + GB2_BT - 1) / GB2_BT, nsplit * nbatch);
  static const bool coal = dbg_opt("x3_coal", 1) != 0;   // (the older k-contiguous load mapping: 0)
  if ((AMODE == GEMM_KC || BMODE == GEMM_KC) && !coal) {
#ifndef CLSTM_HIP_EMU
    static bool attr_set0 = false;
    if (!attr_set0) {
      (void)hipFuncSetAttribute((const void*)gemm_x3_128_kernel<AMODE, BMODE, FE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set0 = true;
    }
#endif
    CLSTM_LAUNCH((gemm_x3_128_kernel<AMODE, BMODE, FE, false>), grid, dim3(256), smem, stream, A, B, fe, R, Cn, K, ksplit, nsplit);
    return;
  }
  static const bool ilv = dbg_opt("x3_ilv", 1) != 0;     // (conversion in front of the MFMAs instead of between them: 0)
  if (!ilv) {
#ifndef CLSTM_HIP_EMU
    static bool attr_set1 = false;
    if (!attr_set1) {
      (void)hipFuncSetAttribute((const void*)gemm_x3_128_kernel<AMODE, BMODE, FE, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set1 = true;
    }
#endif
    CLSTM_LAUNCH((gemm_x3_128_kernel<AMODE, BMODE, FE, true, false>), grid, dim3(256), smem, stream, A, B, fe, R, Cn, K, ksplit, nsplit);
    return;
  }
  CLSTM_LAUNCH((gemm_x3_128_kernel<AMODE, BMODE, FE>), grid, dim3(256), smem, stream, A, B, fe, R, Cn, K, ksplit, nsplit);
}

// Operand slack: the second float4 of a KC row may run 7 floats past the row end (library buffers carry
// >= 64 floats of slack; exact-size user arrays get a descriptor that ends at the last element).
template <int AMODE, int BMODE, class FE>
inline void gemm_bf16(hipStream_t stream, GemmOperand A, GemmOperand B, FE fe, int R, int Cn, int K, int nsplit = 1,
                      int nbatch = 1) {
  if (R <= 0 || Cn <= 0) return;
  if (nsplit < 1) nsplit = 1;
  int ksplit = (K + nsplit - 1) / nsplit;
  const int kq = nsplit > 1 ? GB_PF * GB_BK : GB_BK;
  ksplit = ((ksplit + kq - 1) / kq) * kq;
  if (ksplit < kq) ksplit = kq;
  if (gemm_bf16_big(R, Cn)) {
    // (KC rows are read 16 floats at a time here: up to 15 floats past the row end, same slack rule)
    dim3 grid((Cn + GB2_BT - 1) / GB2_BT, (R + GB2_BT - 1) / GB2_BT, nsplit * nbatch);
    CLSTM_LAUNCH((gemm_bf16_128_kernel<AMODE, BMODE, FE>), grid, dim3(256), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit);
    return;
  }
  dim3 grid((Cn + GEMM_BT - 1) / GEMM_BT, (R + GEMM_BT - 1) / GEMM_BT, nsplit * nbatch);
  CLSTM_LAUNCH((gemm_bf16_kernel<AMODE, BMODE, FE>), grid, dim3(256), 0, stream, A, B, fe, R, Cn, K, ksplit, nsplit);
}

}  // namespace clstm
