// wave_tiles.h -- a small product computed by the waves of ONE workgroup without LDS or barriers:
//
//   out[frame][col0 + c] = sum_k X[frame][k] * W[c][k] (+ bias[c])      16-frame tiles x 16-column tiles, f32 MFMA
//
// the shape of the helper roles of the fused launches (lstm_fwd_fused.h: G = W_x.x + b in front of the forward recurrence;
// lstm_bwd_dw.h: the softmax layer's x.d = W^T z.d in front of the backward recurrence).  Both operands are k-contiguous
// rows; a lane (i, kq) loads four consecutive k of row i and MFMA e of a 16-k group uses element e of both operands'
// float4 (the same k permutation on both sides, as lstm_wide.h).  The MFMAs are issued with the operands exchanged, so
// that a lane ends up with four consecutive COLUMNS of one frame: one 16-byte store, written through (the consumer is
// another workgroup of the same launch).  Every operand of a pass (up to WT_KG 16-k groups) is requested before the first
// MFMA.  v_mfma_f32_16x16x4_f32 is an exact fmaf chain: parity-grade.
#pragma once
#include "devintrin.h"

namespace clstm {

constexpr int WT_JW = 5;    // column tiles per wave at most
constexpr int WT_KG = 4;    // 16-k groups held in registers at once (longer contractions loop)

struct WaveTileProblem {
  BufF32 xbuf; int ldx;            // X rows (frames), k contiguous
  BufF32 wbuf; int ldw;            // W rows (output columns), k contiguous; rows past the descriptor read zeros
  int ng;                          // 16-k groups of the contraction
  int kvalid;                      // X elements with k >= kvalid are taken as zero (W rows are not zero padded); < 0: W is padded
  BufF32 bbuf;                     // bias per column (an empty descriptor reads zeros)
  BufF32 obuf; int ldo, ocol0;     // out[frame * ldo + ocol0 + c]
  int ncols, ntiles;               // valid columns, 16-column tiles
};

// wave `wave` of `pw` computes column tiles wave, wave + pw, ... (at most WT_JW) for FT tiles of 16 frames whose global
// first frames are fbase[t] (rows f with fok_lo[t] <= f < fok_hi[t] of the tile are real, the others masked)
template <int FT>
DEVFN void wave_tiles(const WaveTileProblem& p, const long long (&fbase)[FT], const int (&flo)[FT], const int (&fhi)[FT],
                      const int wave, const int pw) {
  constexpr int JW = WT_JW;
  const int lane = threadIdx.x & 63;
  const int fi = lane & 15, kq = lane >> 4;
  bool fok[FT];
  unsigned xrow[FT];
#pragma unroll
  for (int t = 0; t < FT; t++) {
    fok[t] = fi >= flo[t] && fi < fhi[t];
    xrow[t] = fok[t] ? (unsigned)((fbase[t] + fi) * p.ldx + 4 * kq) * 4u : BUF_OOB_BASE;
  }
  unsigned wrow[JW];   // row fi of this wave's i-th tile
  f32x4 bv[JW];        // (requested with the operands: behind the MFMAs it was one more round trip)
  f32x4 acc[FT][JW];
#pragma unroll
  for (int i = 0; i < JW; i++) {
    const int tile = wave + i * pw, col = 16 * tile + 4 * kq;
    wrow[i] = tile < p.ntiles ? (unsigned)((tile * 16 + fi) * p.ldw + 4 * kq) * 4u : BUF_OOB_BASE;
    bv[i] = buf_load4(p.bbuf, tile < p.ntiles && col < p.ncols ? (unsigned)col * 4u : BUF_OOB);
#pragma unroll
    for (int t = 0; t < FT; t++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[t][i][q] = 0.0f;
  }
  for (int g0 = 0; g0 < p.ng; g0 += WT_KG) {
    f32x4 xv[WT_KG][FT], wv[WT_KG][JW];
#pragma unroll
    for (int g = 0; g < WT_KG; g++) {     // groups past the contraction read zeros (out-of-range offsets)
      const bool live = g0 + g < p.ng;
#pragma unroll
      for (int t = 0; t < FT; t++) xv[g][t] = buf_load4(p.xbuf, live ? xrow[t] + (unsigned)(g0 + g) * 64u : BUF_OOB);
#pragma unroll
      for (int i = 0; i < JW; i++) wv[g][i] = buf_load4(p.wbuf, live ? wrow[i] + (unsigned)(g0 + g) * 64u : BUF_OOB);
    }
    if (p.kvalid >= 0) {   // X rows run on into the next frame behind their last real element
#pragma unroll
      for (int g = 0; g < WT_KG; g++)
#pragma unroll
        for (int t = 0; t < FT; t++)
#pragma unroll
          for (int e = 0; e < 4; e++) xv[g][t][e] = 16 * (g0 + g) + 4 * kq + e < p.kvalid ? xv[g][t][e] : 0.0f;
    }
#pragma unroll
    for (int g = 0; g < WT_KG; g++)
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int i = 0; i < JW; i++)
#pragma unroll
          for (int t = 0; t < FT; t++) acc[t][i] = mfma16x16x4(wv[g][i][e], xv[g][t][e], acc[t][i]);   // transposed: lane = (frame fi, column quad kq)
  }
#pragma unroll
  for (int t = 0; t < FT; t++)
#pragma unroll
    for (int i = 0; i < JW; i++) {
      const int tile = wave + i * pw, col = 16 * tile + 4 * kq;
      const bool cok = tile < p.ntiles && col < p.ncols;
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; q++) o[q] = acc[t][i][q] + bv[i][q];
      const unsigned ooff = (unsigned)((fbase[t] + fi) * p.ldo + p.ocol0 + col) * 4u;
      if (col + 3 < p.ncols && ((p.ldo | p.ocol0) & 3) == 0) buf_store4_wt(p.obuf, fok[t] && cok ? ooff : BUF_OOB, o);
      else {   // the last columns of a row whose length is not a multiple of four
#pragma unroll
        for (int q = 0; q < 4; q++) buf_store_wt(p.obuf, fok[t] && col + q < p.ncols ? ooff + 4u * q : BUF_OOB, o[q]);
      }
    }
}

}  // namespace clstm
