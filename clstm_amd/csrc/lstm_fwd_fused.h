// lstm_fwd_fused.h -- the forward half of a training step as ONE launch with three workgroup roles.
//
// The forward recurrence of a narrow layer (lstm_seq.h) occupies lines x directions workgroups -- 128 of the 256 CUs at
// the bench minibatch -- for ~98 us, and on one stream it sat between two products that have nothing to do with its
// dependent chain: the batched gate GEMM G = W_x.x + b in front (forward_full1's input half for every frame,
// clstm_compute.cc:275-293; 20-22 us) and the softmax layer behind it (forward_softmax, clstm_compute.cc:324-345;
// 15-18 us).  What must stay serial is the step-to-step dependence of clstm.cc:612-620; G is needed chunk by chunk and
// a frame's softmax needs only that frame's two h vectors.  Here all three live in one grid:
//
//   blocks [0, nrec)            the recurrence, lstm_fwd_body<.., FUSED>: dispatched first, one workgroup per CU;
//   blocks [nrec, nrec + npb)   PRODUCERS: every wave computes one item = the pre-activations of 16 iterations of one
//                               (line, direction), 16 frames x 4 no columns, in time order (all lines' chunk 0 first),
//                               stores them write-through and raises the item's flag;
//   the rest                    CONSUMERS: every wave computes one item = the softmax outputs of 16 frames of one line
//                               as soon as both directions report those frames complete (progress words, as gemm_dw.h),
//                               items sorted by the iteration at which they become ready (the middle of a line first).
//
// Items are wave-sized on purpose: no LDS, no barrier, no cross-wave reduction -- a producer / consumer workgroup is
// seven independent waves, so a helper never waits for a sibling and an item's latency is one wave's ~300 MFMAs
// (v_mfma_f32_16x16x4_f32, f32 accumulate: exact fmaf chains, the parity path).  Operands come k-contiguous: the frames
// as they are (X rows / H rows), the weights from per-update packs (ops.h: Wk, W1k).  Fragment convention as
// lstm_wide.h: lane (i, kq) loads four consecutive k of row i and MFMA e of a 16-k group uses element e of both
// operands' float4 -- the same k permutation on both sides.
//
// Hand-off (guide: in-launch producer / consumer, write-through variant): payload stores are sc0 sc1 (16-byte for G),
// the storing wave drains its VMEM queue, ONE lane stores the flag / progress word at system scope; readers poll
// relaxed and load the payload with sc0 sc1 loads -- no fences, correct for any placement of the roles on the XCDs.
// Deadlock freedom: producers never wait; the recurrence workgroups wait only for producers, which sit in front of
// the consumers in dispatch order; the host launches this form only when lines x directions leaves CUs free.
#pragma once
#include "lstm_seq.h"

namespace clstm {

struct FwdFusedArgs {
  // producers
  const float* X; int ldx; long long x_elems;    // layer input rows [N][ldx] (k contiguous), readable floats
  const float* Wk; int kp; int njp;              // [ndir][njp * 16][kp] gate rows k-contiguous, zero padded (ops.h:pack_wk)
  const float* bias;                             // [ndir * 4 no]
  const int* pitems; int npitems;                // producer items in time order, 4 ints each: (line << 13 | dir << 12 | chunk), line offset, T, 0
  int* gflag;                                    // = LstmSeqArgs::gflag
  // consumers
  const float* W1k; int kps; int sm_k;           // [96][kps] softmax rows k-contiguous, zero padded; sm_k = ndir * no
  const float* b1;                               // softmax bias = W1[:, 0]
  float* Z; int nc;
  const int* citems; int ncitems;                // consumer items by readiness: (line << 12 | 16-frame block)
  const int* prog;                               // progress words (see LstmSeqArgs::prog_off)
  int nrec, npb;                                 // recurrence workgroups, producer workgroups
  long long* trace;                              // diagnostics (CLSTM_FW_TRACE): [nrec + npitems + ncitems][4] wall-clock stamps
  int* nanflag; int step_no;                      // non-finite logits (ops.h:raise_nonfinite), or null
};

// ---- producer item (one WORKGROUP): G[frames f0 .. f0+15 of line b][dir][:] = W_x . x + b  -------------------------
// The item's 16-column tiles are dealt round-robin to FOUR waves (25 tiles at 100 cells: 7 or 6 each); the other waves
// of the workgroup retire at once, so that three such workgroups fit a CU (the launch's register count admits 12 waves).
// Every operand of the item is requested before the first MFMA -- the contraction is short (K = 48 for a text-line
// image: three 16-k groups), so an item is one round of loads, ~80 MFMAs per wave, one round of write-through stores.
// Measured history (profiles/r03_fwd_timeline.txt): a whole item per WAVE behind twelve dependent rounds of loads took
// 25 us and the recurrence started 28 us into the launch; one item per seven-wave workgroup with one group prefetched
// took 6.2 us with one workgroup per CU -- barely faster than the recurrence consumes (a chunk per 8 us).
constexpr int FWD_JW = 5;    // column tiles per wave at most: ceil(28 / 6) (workgroups of >= 6 waves)
constexpr int FWD_FT = 2;    // 16-frame tiles (chunks) per producer item: the weight fragments are loaded once for both
constexpr int FWD_KG = 4;    // 16-k groups held in registers at once (K <= 64 per pass; longer contractions loop)
// wave `wave` of `pw` computes column tiles wave, wave + pw, ... (at most FWD_JW) for FT tiles of 16 frames starting at
// f0[0..FT) of one line (a tile with f0 = INT_MIN/2 is skipped: every access masked)
template <int FT>
DEVFN void fwd_gx_tiles(const LstmSeqArgs& a, const FwdFusedArgs& h, const int dir, const int off, const int T, const int (&f0)[FT],
                        const int wave, const int pw) {
  constexpr int JW = FWD_JW;
  const int lane = threadIdx.x & 63;
  const int fi = lane & 15, kq = lane >> 4;
  const int no4 = 4 * a.no, nd = a.ndir;
  const BufF32 xbuf = make_buf(h.X, (size_t)h.x_elems * 4);
  const BufF32 wbuf = make_buf(h.Wk + (size_t)dir * h.njp * 16 * h.kp, (size_t)h.njp * 16 * h.kp * 4);
  const BufF32 bbuf = make_buf(h.bias + (size_t)dir * no4, (size_t)no4 * 4);
  bool fok[FT];
  unsigned xrow[FT];
#pragma unroll
  for (int t = 0; t < FT; t++) {
    const int f = f0[t] + fi;
    fok[t] = f >= 0 && f < T;
    xrow[t] = fok[t] ? (unsigned)((long long)(off + f) * h.ldx + 4 * kq) * 4u : BUF_OOB_BASE;
  }
  unsigned wrow[JW];   // row fi of this wave's i-th tile
  f32x4 bv[JW];        // (requested with the operands: behind the MFMAs it was one more round trip)
  f32x4 acc[FT][JW];
#pragma unroll
  for (int i = 0; i < JW; i++) {
    const int tile = wave + i * pw, col = 16 * tile + 4 * kq;
    wrow[i] = tile < h.njp ? (unsigned)((tile * 16 + fi) * h.kp + 4 * kq) * 4u : BUF_OOB_BASE;
    bv[i] = buf_load4(bbuf, tile < h.njp && col < no4 ? (unsigned)col * 4u : BUF_OOB);
#pragma unroll
    for (int t = 0; t < FT; t++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[t][i][q] = 0.0f;
  }
  const int ng = h.kp >> 4;
  for (int g0 = 0; g0 < ng; g0 += FWD_KG) {
    f32x4 xv[FWD_KG][FT], wv[FWD_KG][JW];
#pragma unroll
    for (int g = 0; g < FWD_KG; g++) {     // groups past the contraction read zeros (out-of-range offsets)
      const bool live = g0 + g < ng;
#pragma unroll
      for (int t = 0; t < FT; t++) xv[g][t] = buf_load4(xbuf, live ? xrow[t] + (unsigned)(g0 + g) * 64u : BUF_OOB);
#pragma unroll
      for (int i = 0; i < JW; i++) wv[g][i] = buf_load4(wbuf, live ? wrow[i] + (unsigned)(g0 + g) * 64u : BUF_OOB);
    }
#pragma unroll
    for (int g = 0; g < FWD_KG; g++)
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int i = 0; i < JW; i++)
#pragma unroll
          for (int t = 0; t < FT; t++) acc[t][i] = mfma16x16x4(wv[g][i][e], xv[g][t][e], acc[t][i]);   // transposed: lane = (frame fi, column quad kq)
  }
  // lane holds columns 16 j + 4 kq + (0..3) of frame fi = the four gates of cell 4 j + kq: one 16-byte store
  const BufF32 gbuf = make_buf(a.G, (size_t)(a.line_off[a.bs]) * nd * no4 * 4);
#pragma unroll
  for (int t = 0; t < FT; t++)
#pragma unroll
    for (int i = 0; i < JW; i++) {
      const int tile = wave + i * pw, col = 16 * tile + 4 * kq;
      const bool cok = tile < h.njp && col < no4;
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; q++) o[q] = acc[t][i][q] + bv[i][q];
      buf_store4_wt(gbuf, fok[t] && cok ? (unsigned)(((long long)(off + f0[t] + fi) * nd + dir) * no4 + col) * 4u : BUF_OOB, o);
    }
}
// the recurrence workgroup's own chunk 0 (all its waves; the caller drains and synchronises)
DEVFN void fwd_self_produce(const LstmSeqArgs& a, const FwdFusedArgs& h, const int b, const int dir, const int off, const int T) {
  const int wave = wave_uniform((int)threadIdx.x >> 6), nw = (int)blockDim.x >> 6;   // nw >= FWD_CW = 6: at most 5 tiles per wave
  const int f0[1] = {dir == 0 ? 0 : T - 16};
  fwd_gx_tiles<1>(a, h, dir, off, T, f0, wave, nw);
}
// producer item (one workgroup, all its waves): chunks c0 .. c0 + FWD_FT - 1 of one (line, direction)
DEVFN void fwd_gx_item(const LstmSeqArgs& a, const FwdFusedArgs& h, const int it) {
  const int wave = wave_uniform((int)threadIdx.x >> 6), nw = (int)blockDim.x >> 6;
  const long long t_start = h.trace ? wall_clock() : 0;
  // the item record carries the line's extent: one round trip in front of the operand loads instead of two
  struct alignas(16) ItemRec { int code, off, T, pad; };
  const ItemRec rec = reinterpret_cast<const ItemRec*>(h.pitems)[it];
  const int b = rec.code >> 13, dir = (rec.code >> 12) & 1, c0 = rec.code & 4095, off = rec.off, T = rec.T;
  const int nchunk = (T + 15) >> 4;
  // iterations [16 c, 16 c + 16): direction 0 visits frame it, direction 1 frame T - 1 - it (a negative first frame in a
  // line's last chunk: masked row by row)
  int f0[FWD_FT];
#pragma unroll
  for (int t = 0; t < FWD_FT; t++) f0[t] = c0 + t < nchunk ? (dir == 0 ? 16 * (c0 + t) : T - 16 * (c0 + t) - 16) : -(1 << 30);
  fwd_gx_tiles<FWD_FT>(a, h, dir, off, T, f0, wave, nw);
  drain_vmem();      // every storing wave: its rows are in memory ...
  __syncthreads();
  if (threadIdx.x == 0) {   // ... before the flags are
#pragma unroll
    for (int t = 0; t < FWD_FT; t++)
      if (c0 + t < nchunk) store_i32_wt(h.gflag + ((size_t)dir * a.bs + b) * a.gchunks + c0 + t, a.gepoch);
    if (h.trace) { long long* tr = h.trace + (size_t)(h.nrec + it) * 4; tr[0] = t_start; tr[2] = wall_clock(); tr[3] = c0; }
  }
}

// ---- consumer item (one WORKGROUP): Z[frames f0 .. f0+15 of line b] = softmax(W1 . [h_fwd ; h_rev] + b1) ----------------
// The six 16-class tiles go to six waves (any further wave retires: two such workgroups fit a CU); a wave requests its
// whole operand -- the 16 H rows (system-scope loads: written through by the recurrence) and its 16 weight rows, 13 + 13
// 16-byte loads at 200 inputs -- before the first MFMA, and the row sums meet in LDS (the launch's dynamic LDS: the
// recurrence role's h buffer, unused here).  What matters is the latency of the LAST items, which become ready when the
// recurrence ends: one round of loads, 52 MFMAs, one barrier.  (A whole item per wave: 14-18 us behind the recurrence.)
constexpr int FWD_CW = 6;     // consumer waves (class tiles) per workgroup
constexpr int FWD_CG = 14;    // 16-k groups in registers: ndir * no <= 224
DEVFN void fwd_softmax_item(const LstmSeqArgs& a, const FwdFusedArgs& h, const int b, const int blk, const int it) {
  const int lane = threadIdx.x & 63, wave = wave_uniform((int)threadIdx.x >> 6);
  if (wave >= FWD_CW) return;
  float* part = dyn_smem<float>();     // [FWD_CW][16] partial row sums
  const long long t_start = h.trace ? wall_clock() : 0;
  const int fi = lane & 15, kq = lane >> 4;
  const int off = a.line_off[b], T = a.line_off[b + 1] - off;
  const int f0 = 16 * blk;
  const int fhi = f0 + 16 < T ? f0 + 16 : T;
  // weights and bias do not depend on the recurrence: requested before the wait
  const BufF32 wbuf = make_buf(h.W1k, (size_t)96 * h.kps * 4);
  const unsigned wrow = (unsigned)((wave * 16 + fi) * h.kps + 4 * kq) * 4u;
  const int ng = h.kps >> 4;
  f32x4 wv[FWD_CG];
#pragma unroll
  for (int g = 0; g < FWD_CG; g++) wv[g] = buf_load4(wbuf, g < ng ? wrow + (unsigned)g * 64u : BUF_OOB);
  const BufF32 bbuf = make_buf(h.b1, (size_t)h.nc * 4);
  const bool cok = wave * 16 + fi < h.nc;
  const float bias = buf_load(bbuf, cok ? (unsigned)(wave * 16 + fi) * 4u : BUF_OOB);
  // both directions must have stored these frames: direction 0 after fhi iterations, direction 1 after T - f0
  {
    const int need0 = fhi, need1 = T - f0;
    const int* p0 = h.prog + (size_t)b * PROG_STRIDE;
    const int* p1 = h.prog + ((size_t)(a.ndir - 1) * a.bs + b) * PROG_STRIDE;
    int polls = 0;
    for (;;) {
      const int have0 = wave_uniform(load_i32_wt(p0)) - a.prog_base;
      const int have1 = a.ndir > 1 ? wave_uniform(load_i32_wt(p1)) - a.prog_base : 0x3fffffff;
      const int d0 = need0 - have0, d1 = a.ndir > 1 ? need1 - have1 : 0;
      const int deficit = d0 > d1 ? d0 : d1;
      if (deficit <= 0) break;
      if (++polls > (1 << 22)) { if (lane == 0) atomic_add_i32(a.timeouts, 1); break; }   // never hang the device
      sleep_iterations(deficit > 12 ? deficit - 6 : 1);   // ~0.35 us per missing iteration (at most ~14 us per nap)
    }
  }
  const long long t_ready = h.trace ? wall_clock() : 0;
  const bool fok = f0 + fi < T;
  const BufF32 hbuf = make_buf(a.H, (size_t)((long long)a.line_off[a.bs] * a.ldh + 16) * 4);
  const unsigned hrow = fok ? (unsigned)((long long)(off + f0 + fi) * a.ldh + a.hofs + 4 * kq) * 4u : BUF_OOB_BASE;
  f32x4 hv[FWD_CG];
#pragma unroll
  for (int g = 0; g < FWD_CG; g++) hv[g] = buf_load4_wt(hbuf, g < ng ? hrow + (unsigned)g * 64u : BUF_OOB);
  f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int g = 0; g < FWD_CG; g++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc = mfma16x16x4(hv[g][e], wv[g][e], acc);   // rows = frames 4 kq + q, column = class fi of the tile
  // limexp and the row sums: a frame's 16 classes of this tile sit in the 16 lanes of one row group
  float ev[4];
  bool nonfinite = false;   // limexp's clamp would swallow a NaN logit: looked at before it (k_update, ops.h)
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float x = acc[q] + bias;
    nonfinite |= cok && f0 + 4 * kq + q < T && !f32_finite(x);
    float v = expf(fminf(fmaxf(x, -30.0f), 30.0f));
    v = x < -30.0f ? (float)0x1.a56e0c2b7ab97p-44 : v;   // (Float)exp(-30.0), tensor.h:78-82
    v = x > 30.0f ? (float)0x1.37047090c0b53p+43 : v;    // (Float)exp(30.0)
    ev[q] = cok ? v : 0.0f;
    float s = ev[q];
    s += row_ror<8>(s);
    s += row_ror<4>(s);
    s += row_ror<2>(s);
    s += row_ror<1>(s);
    if (fi == 0) part[wave * 16 + 4 * kq + q] = s;
  }
  __syncthreads();   // (six live waves)
  const BufF32 zbuf = make_buf(h.Z, (size_t)a.line_off[a.bs] * h.nc * 4);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int fr = f0 + 4 * kq + q;
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < FWD_CW; w++) s += part[w * 16 + 4 * kq + q];   // fixed order: deterministic
    buf_store(zbuf, fr < T && cok ? ((unsigned)(off + fr) * (unsigned)h.nc + (unsigned)(wave * 16 + fi)) * 4u : BUF_OOB, ev[q] / s);
  }
  if (h.nanflag && nonfinite) raise_nonfinite(h.nanflag, h.step_no);
  if (h.trace && threadIdx.x == 0) {
    long long* tr = h.trace + (size_t)(h.nrec + h.npitems + it) * 4;
    tr[0] = t_start; tr[1] = t_ready; tr[2] = wall_clock(); tr[3] = (fhi > T - f0 ? fhi : T - f0);
  }
}

struct FwdFusedKernelArgs { LstmSeqArgs a; FwdFusedArgs h; };

// helper roles: one item per workgroup -- block (hblock) of the producer list, then of the consumer list
DEVFN void fwd_fused_helper(const LstmSeqArgs& a, const FwdFusedArgs& h, const int hblock) {
  if (hblock < h.npb) {
    fwd_gx_item(a, h, hblock);
  } else {
    const int code = h.citems[hblock - h.npb];
    fwd_softmax_item(a, h, code >> 12, code & 4095, hblock - h.npb);
  }
}

template <int NK4, int KU>
__global__ __launch_bounds__(64 * NK4) CLSTM_TWO_WAVES_PER_SIMD void lstm_fwd_fused_kernel(FwdFusedKernelArgs k) {
  if ((int)blockIdx.x < k.h.nrec) {
#ifndef CLSTM_HIP_EMU
    __builtin_amdgcn_s_setprio(3);
#endif
    const int bl = (int)blockIdx.x % k.a.bs;
    const long long t0 = k.h.trace ? wall_clock() : 0;
    lstm_fwd_body<NK4, KU, true>(k.a, k.a.order ? k.a.order[bl] : bl, (int)blockIdx.x / k.a.bs, &k.h);
    if (k.h.trace && threadIdx.x == 0) { k.h.trace[blockIdx.x * 4] = t0; k.h.trace[blockIdx.x * 4 + 2] = wall_clock(); }
  } else {
    fwd_fused_helper(k.a, k.h, (int)blockIdx.x - k.h.nrec);
  }
}

}  // namespace clstm
