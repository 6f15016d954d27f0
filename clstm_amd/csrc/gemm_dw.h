// gemm_dw.h -- the weight-gradient GEMM of a BiLSTM layer as a CONSUMER that runs beside the backward recurrence.
//
//   W.d[dir] = sum_frames  [1 ; x_t ; h_{t-1}] (x) delta_t          (backward_full1 / backward_lin1, clstm_compute.cc:294-304)
//
// In the reference this rank-1 update happens inside every time step.  Hoisted to one split-K GEMM over all
// frames it cannot start before the recurrence kernel has finished (gemm_mfma.h) -- 52 us on an otherwise idle half
// of the chip (128 recurrence workgroups on 256 CUs).  Here the contraction is cut along TIME: a "chunk" is a range
// of the recurrence's iterations, the same for every line; the slabs of chunk c contract the frames those
// iterations visit (dir 0 walks a line backwards, dir 1 forwards), and a slab's workgroups start as soon as every
// line reports those iterations complete (lstm_seq.h: progress words written through with the per-step delta
// store).  The launch runs on a stream whose CU mask is the complement of the recurrence stream's, so neither
// kernel takes issue slots from the other.
//
// Frames of a chunk are not contiguous in the packed line batch: a host-built table lists the k-tiles
// (first frame, count <= 16) in slab order.  Tile / staging / MFMA structure as gemm_mfma.h (MC x MC operands).
#pragma once
#include <type_traits>
#include "gemm_bf16.h"

namespace clstm {

struct DwSlab {
  int tile_begin, ntiles;   // range in the direction's k-tile table
  int need_it;              // iterations of every line that must be complete (the chunk's end)
  int dir;
  int out_z;                // slab index in the partial-sum array: dir * slabs_per_dir + index within the direction
  int pad[3];
};

struct GemmDwArgs {
  const float* S; long long sdir; int lds; long long s_elems;   // A: S[dir][frame][col]
  const float* D; int M; int no4; long long d_elems;            // B: D[frame][dir*no4 + c]  (written concurrently: system-scope loads)
  const int* ktab;          // [ndir][ntiles_max][2] (first frame, count)
  int ntiles_max;
  const DwSlab* slabs;      // in readiness order
  int nslabs;
  const int* prog;          // [ndir][bs] progress words (iterations complete, biased by prog_base)
  const int* line_off;
  int bs, prog_base, ndir;
  float* partial;           // [ndir * slabs_per_dir][R][Cn]
  int R, Cn;
  unsigned gx, gy;          // output tiles along Cn, R
  int* timeouts;            // incremented when a slab gave up waiting (diagnostics; results are then wrong)
  int* minprog;             // [ndir] (PROG_STRIDE apart): prog_base + iterations EVERY line of that direction has completed,
                            //   published by the monitor workgroup (gemm_dw_monitor); what the items poll
  int tcap;                 // value published once every line is complete (longest line + 32)
  int* done;                // += 1 by every recurrence workgroup when its line is complete and its stores have landed
  int done_target;          //   value of *done when ALL of this launch's lines are complete (the counter is never reset)
  long long* trace;         // diagnostics (CLSTM_DW_TRACE): [workgroup][4] wall-clock stamps -- start, ready, done
  int trace_base;           // first trace row of the GEMM role's workgroups
  // a second product of the same form that depends on NOTHING in this launch -- the softmax layer's W.d = sum_t z.d_t [1;h_t]^T
  // (SoftmaxLayer::backward, clstm.cc:411-417): its items come first in dispatch order and run on the idle half of the
  // chip while the first chunk of the recurrence is still being produced (x3 items only)
  const float* xS; int xlds; long long xs_elems;   // A': [frame][xlds], row r of the product = column r
  const float* xD; int xM; long long xd_elems;     // B': [frame][xM]
  const int* xtab; const DwSlab* xslabs; int xnslabs;   // its k-tile table (contiguous frames) and slabs (need_it = 0)
  float* xpartial; int xR, xCn; unsigned xgx, xgy;
  int x3;                   // 1: products on the bf16 MFMA with both operands split into bf16 terms (gemm_dw_item_x3), 0: f32 MFMA
  int terms;                // x3: 3 = hi + mid + lo, six products, operand-exact (default); 2 = hi + lo, three products
};

// blocks of the x3 item in flight in registers (6 and 8 were measured: the fused launch then needs > 168 registers,
// only one GEMM workgroup fits a CU, 142 / 148 us against 116)
constexpr int DW_PF = 3;
#ifdef CLSTM_HIP_EMU
#define DW_SGB(mask, n) do {} while (0)
#else
#define DW_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif
// LDS of the GEMM role: two buffers of 2 NT 64 x 32 bf16 images (NT terms per operand; the f32 item: 8192 floats; the epilogue
// tile fits too) + the slab's k-tile table (DW_STAB_MAX entries of 2 ints)
#ifndef CLSTM_DW_STAB_MAX
#define CLSTM_DW_STAB_MAX 256   // 2 KB: with three terms the role's LDS is 50 KB -- three workgroups per CU (1024 entries: 56 KB, two)
#endif
constexpr int DW_STAB_MAX = CLSTM_DW_STAB_MAX;
constexpr int dw_img_floats(int nt) { return nt <= 2 ? 8192 : 2 * 2 * nt * 64 * 32 / 2; }
constexpr int dw_smem_floats(int nt) { return dw_img_floats(nt) + 2 * DW_STAB_MAX; }
static_assert(8192 >= GEMM_BT * GEMM_LDO, "epilogue tile");

constexpr int DW_WATCHDOG_POLLS = 1 << 16;

// ---- progress: one monitor, many waiters ---------------------------------------------------------------------------
// A workgroup that waited by looking at the lines' progress words itself paid a system-scope load of up to 64 cache
// lines per look, so hundreds of waiters could only look rarely (every look starves the recurrence's write-through
// stores: 92 -> 630 us when they polled freely) and the last chunk's items noticed the end of the recurrence ~13 us
// late (measured with the contraction left out: 109 us against 96).  Now ONE wave of ONE extra workgroup (the first
// behind the recurrence's in dispatch order) looks at the lines every ~0.5 us and publishes, per direction, the number
// of iterations every line has completed; the items poll that single word.
DEVFN void gemm_dw_monitor(const GemmDwArgs& a) {
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x & 63;
  int polls = 0;
  for (;;) {
    int all = 0x3fffffff;
    int pv[2][4];                      // all loads of a look are requested before the first is used (up to 256 lines per direction)
#pragma unroll
    for (int dir = 0; dir < 2; dir++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int b = lane + 64 * j;
        pv[dir][j] = dir < a.ndir && b < a.bs ? load_i32_wt(a.prog + ((size_t)dir * a.bs + b) * PROG_STRIDE) : 0;
      }
    for (int dir = 0; dir < a.ndir; dir++) {
      int m = 0x3fffffff;
      for (int b = lane, j = 0; b < a.bs; b += 64, j++) {
        const int T = a.line_off[b + 1] - a.line_off[b];
        const int p = (j < 4 ? pv[dir][j] : load_i32_wt(a.prog + ((size_t)dir * a.bs + b) * PROG_STRIDE)) - a.prog_base;   // < 0: words of an older launch
        const int v = p >= T ? 0x3fffffff : (p < 0 ? 0 : p);   // a finished line constrains nothing
        m = v < m ? v : m;
      }
      m = -wave_max_i(-m);
      if (lane == 0) store_i32_wt(a.minprog + dir * PROG_STRIDE, a.prog_base + (m > a.tcap ? a.tcap : m));
      all = m < all ? m : all;
    }
    if (all >= 0x3fffffff) break;            // every line of every direction is complete (and tcap is published)
    if (++polls > (DW_WATCHDOG_POLLS << 4)) {   // never hang the device: release the waiters, flag it, produce garbage
      if (lane == 0) {
        atomic_add_i32(a.timeouts, 1);
        for (int dir = 0; dir < a.ndir; dir++) store_i32_wt(a.minprog + dir * PROG_STRIDE, a.prog_base + a.tcap);
      }
      break;
    }
  }
}

// wait until every line has completed `need_it` iterations of direction `dir` (called by all 256 threads): wave 0
// polls the monitor's word, sleeping for about the time the missing iterations take (at most ~8 of them)
DEVFN void gemm_dw_wait(const GemmDwArgs& a, const int dir, const int need_it) {
  const int wave = wave_uniform(threadIdx.x >> 6);
  if (wave == 0) {
    const int need = need_it < a.tcap ? need_it : a.tcap;
    // the last chunk (every iteration of every line) is released by the recurrence workgroups themselves: one counter,
    // no monitor hop (its look + publish + our poll cost ~2.5 us exactly where the launch's tail is)
    const bool all = a.done && need_it >= a.tcap - 32;
    int polls = 0;
    for (;;) {
      int have;
      if (all) have = (int)((unsigned)wave_uniform(load_i32_wt(a.done)) - (unsigned)a.done_target) >= 0 ? need : need - 1;
      else have = wave_uniform(load_i32_wt(a.minprog + dir * PROG_STRIDE) - a.prog_base);
      if (have >= need) break;
      if (++polls > (DW_WATCHDOG_POLLS << 4)) {   // the monitor never showed up
        if ((threadIdx.x & 63) == 0) atomic_add_i32(a.timeouts, 1);
        break;
      }
      const int deficit = need - (have < 0 ? 0 : have);
      sleep_iterations(deficit > 12 ? 6 : 1);   // close to ready: look every ~0.35 us (a look is one word)
    }
  }
  __syncthreads();
}

// one work item: output tile `tile` of slab `si` (256 threads; `smem` = GEMM_BT * GEMM_LDO floats)
DEVFN void gemm_dw_item(const GemmDwArgs& a, float* smem, const unsigned si, const unsigned tile) {
  float* As = smem;
  float* Bs = smem + GEMM_BK * GEMM_LD;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const DwSlab sl = a.slabs[si];
  const int r0 = (int)(tile / a.gx) * GEMM_BT, c0 = (int)(tile % a.gx) * GEMM_BT;
  const int dir = sl.dir;
  gemm_dw_wait(a, dir, sl.need_it);

  const int a_mn = (tid & 15) * 4, a_k = tid >> 4;   // MC staging: 4 consecutive columns of frame row (tid >> 4)
  const BufF32 abuf = make_buf(a.S + (size_t)dir * a.sdir, (size_t)(a.s_elems - (long long)dir * a.sdir) * 4);
  const BufF32 bbuf = make_buf(a.D, (size_t)a.d_elems * 4);
  const unsigned a_col = (unsigned)(r0 + a_mn), b_col = (unsigned)(dir * a.no4 + c0 + a_mn);
  const int* tab = a.ktab + (size_t)dir * a.ntiles_max * 2;
  const int tend = sl.tile_begin + sl.ntiles;

  // unconditional loads (a tile past the slab / a frame row past the tile's count gets an out-of-range offset) so
  // that the VMEM queue is counted exactly; zeroing happens when the tile is staged
  auto load_tile = [&](int t, f32x4& ra, f32x4& rb, int& cnt) {
    const bool live = t < tend;
    const int tt = live ? t : sl.tile_begin;
    const int f0 = tab[2 * tt];
    cnt = live ? tab[2 * tt + 1] : 0;
    const bool row = a_k < cnt;
    const unsigned f = (unsigned)(f0 + a_k);
    ra = buf_load4(abuf, row ? (f * (unsigned)a.lds + a_col) * 4u : BUF_OOB);
    rb = buf_load4_wt(bbuf, row ? (f * (unsigned)a.M + b_col) * 4u : BUF_OOB);
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;

  f32x4 ra[GEMM_PF], rb[GEMM_PF];
  int cnt[GEMM_PF];
#pragma unroll
  for (int p = 0; p < GEMM_PF; p++) {
    load_tile(sl.tile_begin + p, ra[p], rb[p], cnt[p]);
    SCHED_FENCE();
  }
  const int fk = lane >> 4, fi = lane & 15;
  for (int tb = sl.tile_begin; tb < tend; tb += GEMM_PF) {
#pragma unroll
    for (int p = 0; p < GEMM_PF; p++) {
      const bool row = a_k < cnt[p];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        ra[p][i] = row ? ra[p][i] : 0.0f;
        rb[p][i] = row ? rb[p][i] : 0.0f;
      }
      *reinterpret_cast<f32x4*>(&As[a_k * GEMM_LD + a_mn]) = ra[p];
      *reinterpret_cast<f32x4*>(&Bs[a_k * GEMM_LD + a_mn]) = rb[p];
      __syncthreads();
      load_tile(tb + p + GEMM_PF, ra[p], rb[p], cnt[p]);
      SCHED_FENCE();
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
  }
  // epilogue through LDS: whole 256-byte row segments per store instruction (as gemm_mfma.h)
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        smem[(wm * 32 + i * 16 + (lane >> 4) * 4 + q) * GEMM_LDO + wn * 32 + j * 16 + (lane & 15)] = acc[i][j][q];
  __syncthreads();
  float* out = a.partial + (size_t)sl.out_z * a.R * a.Cn;
  const bool v4 = (a.Cn & 3) == 0 && ((size_t)a.partial & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int rl = it * 16 + (tid >> 4), cl = (tid & 15) * 4;
    const int r = r0 + rl, c = c0 + cl;
    const f32x4 v = *reinterpret_cast<const f32x4*>(&smem[rl * GEMM_LDO + cl]);
    if (r < a.R) {
      if (v4 && c + 3 < a.Cn) *reinterpret_cast<f32x4*>(out + (size_t)r * a.Cn + c) = v;
      else
        for (int e = 0; e < 4; e++)
          if (c + e < a.Cn) out[(size_t)r * a.Cn + c + e] = v[e];
    }
  }
}
// ---- the same work item on the bf16 MFMA with f32 operands split into bf16 TERMS (the differences are exact in f32):
//   NT = 3 (default): x = x1 + x2 + x3, x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) -- 3 x 8 significant bits:
//     the split represents every normal f32 EXACTLY.  A product is the six terms of weight >= 2^-16,
//     x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1), each exact in the f32 accumulator; what is dropped (x2y3, x3y2, x3y3) is
//     < 2^-23 |x||y| -- the size of the rounding an f32 multiply makes itself.  "Operand-exact": nothing narrower than the
//     reference's float enters the gradient.  24 MFMAs of 16 cycles per 32-frame block and wave
//     against 32 of 32 on the f32 MFMA.
//   NT = 2 (experiment option split_terms=2; rounds 3-5's default): hi + lo, hi.hi + hi.lo + lo.hi, 12 MFMAs, < 2^-16 |x||y|
//     per product -- the gradient stayed within ~1e-5 of its largest entry of the float64 oracle's either way
//     (tests/test_gpu_e2e.py::test_full_shape_gradient_error_vs_float64).
// The product that needed the whole chip for ~45 us on the f32 MFMA (and held the fused backward launch 39 us past the
// recurrence's end) keeps pace with the recurrence on the half of the chip the recurrence leaves idle.
// Staging: waves 0-1 convert the S block, waves 2-3 the D block; lane (m8, kg) loads frames 4 kg .. 4 kg + 3 of the
// 32-frame block (two 16-frame table entries), 4 columns each, transposes in registers and writes 4 k of one row per
// ds_write_b64 into swizzled [mn][32 k] images (gemm_bf16.h: conflict-free fragment reads); images are double-buffered,
// one barrier per block.
// ILV: the conversion of block t + 1 issued BETWEEN the MFMAs of block t (sched_group_barrier; as gemm_bf16.h: gemm_x3_128_kernel)
template <int NT, bool ILV = false>
DEVFN void gemm_dw_item_x3(const GemmDwArgs& a, float* smem, const unsigned si, const unsigned tile, const bool extra = false) {
  constexpr int IMG = 64 * 32;                 // halfs per image
  constexpr int BUFH = 2 * NT * IMG;           // halfs per buffer
  unsigned short* img = reinterpret_cast<unsigned short*>(smem);   // [buffer][A terms | B terms][64][32]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const DwSlab sl = extra ? a.xslabs[si] : a.slabs[si];
  const unsigned pgx = extra ? a.xgx : a.gx;
  const int pR = extra ? a.xR : a.R, pCn = extra ? a.xCn : a.Cn;
  const int r0 = (int)(tile / pgx) * GEMM_BT, c0 = (int)(tile % pgx) * GEMM_BT;
  const int dir = sl.dir;
  const long long t_start = a.trace ? wall_clock() : 0;
  // The slab's k-tile table goes to LDS while the item waits.  Read from global memory inside the loop, an entry was a
  // DEPENDENT load in front of every block's operand loads, and VMEM returns in order: waiting for it meant waiting for
  // every operand load still in flight -- one full memory latency per block (measured: 15 us per 16-block item).
  const int* tab = extra ? a.xtab : a.ktab + (size_t)dir * a.ntiles_max * 2;
  int* stab = reinterpret_cast<int*>(smem + dw_img_floats(NT));
  for (int i = tid; i < 2 * sl.ntiles && i < 2 * DW_STAB_MAX; i += 256) stab[i] = tab[2 * sl.tile_begin + i];   // (the host keeps slabs <= DW_STAB_MAX entries)
  if (!(a.x3 & 2) && sl.need_it > 0) gemm_dw_wait(a, dir, sl.need_it);
  else __syncthreads();
  const long long t_ready = a.trace ? wall_clock() : 0;   // (bits 2, 4 of x3: perf experiments -- no wait / no contraction)

  const bool isB = wave >= 2;                                      // wave-uniform staging role
  const int s_mn = (wave & 1) * 32 + (lane & 7) * 4, s_kg = lane >> 3;   // 4 columns x frames 4 kg .. 4 kg + 3
  const BufF32 abuf = extra ? make_buf(a.xS, (size_t)a.xs_elems * 4)
                            : make_buf(a.S + (size_t)dir * a.sdir, (size_t)(a.s_elems - (long long)dir * a.sdir) * 4);
  const BufF32 bbuf = extra ? make_buf(a.xD, (size_t)a.xd_elems * 4) : make_buf(a.D, (size_t)a.d_elems * 4);
  const unsigned col = isB ? (unsigned)((extra ? 0 : dir * a.no4) + c0 + s_mn) : (unsigned)(r0 + s_mn);
  const unsigned ldrow = isB ? (unsigned)(extra ? a.xM : a.M) : (unsigned)(extra ? a.xlds : a.lds);
  const int tend = (a.x3 & 4) ? sl.tile_begin : sl.tile_begin + sl.ntiles;
  const int e_half = s_kg >> 2, kk0 = (s_kg & 3) * 4;              // table entry of the pair, first frame row within it

  // block b = table entries (tile_begin + 2b, + 2b + 1); unconditional loads, rows past an entry's count read zeros.
  // ROLE is a compile-time constant inside each copy of the loop below: a wave-uniform `isB ? load_wt : load` compiles to
  // branches around the VMEM instructions, and hipcc then waits with vmcnt(0) at every join -- the ring of blocks in
  // flight collapsed to one (measured: ~0.9 us per block, 15 us per 16-block item).
  auto load_block = [&](auto role, int t, f32x4 (&r)[4]) {
    constexpr bool ROLE_B = decltype(role)::value;
    const int te = t + e_half;
    const bool live = te < tend;
    const int tt = live ? te : sl.tile_begin;
    const int f0 = stab[2 * (tt - sl.tile_begin)];
    const int cn = stab[2 * (tt - sl.tile_begin) + 1];
    const int cnt = live ? cn : 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned off = kk0 + j < cnt ? ((unsigned)(f0 + kk0 + j) * ldrow + col) * 4u : BUF_OOB;
      if constexpr (ROLE_B) r[j] = buf_load4_wt(bbuf, off);
      else r[j] = buf_load4(abuf, off);
    }
  };
  unsigned short* const my_img = img + (isB ? NT * IMG : 0);
  int wofs[4];                                                     // halfs: row, swizzled chunk, half chunk
#pragma unroll
  for (int i = 0; i < 4; i++) wofs[i] = (s_mn + i) * 32 + ((((s_kg >> 1) ^ gb2_sw(s_mn + i)) << 3) | ((s_kg & 1) << 2));
  auto stage = [&](int buf, const f32x4 (&r)[4]) {
    unsigned short* d = my_img + buf * BUFH;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float e[4] = {r[0][i], r[1][i], r[2][i], r[3][i]};
#pragma unroll
      for (int t = 0; t < NT; t++) {   // term t of the four frames, then the exact remainders
        u32x2 h;
        h[0] = bf16_pack2(e[0], e[1]);
        h[1] = bf16_pack2(e[2], e[3]);
        *reinterpret_cast<u32x2*>(d + t * IMG + wofs[i]) = h;
        if (t + 1 < NT) {
          e[0] -= __builtin_bit_cast(float, h[0] << 16); e[1] -= __builtin_bit_cast(float, h[0] & 0xffff0000u);
          e[2] -= __builtin_bit_cast(float, h[1] << 16); e[3] -= __builtin_bit_cast(float, h[1] & 0xffff0000u);
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

  const int fk = lane >> 4, fi = lane & 15;
  const int fofs = fi * 32 + ((fk ^ gb2_sw(fi)) << 3);
  auto run = [&](auto role) {
  f32x4 rr[DW_PF][4];
#pragma unroll
  for (int p = 0; p < DW_PF; p++) {
    load_block(role, sl.tile_begin + 2 * p, rr[p]);
    SCHED_FENCE();
  }
  stage(0, rr[0]);
  load_block(role, sl.tile_begin + 2 * DW_PF, rr[0]);
  SCHED_FENCE();
  __syncthreads();
  int cur = 0;
  for (int tb = sl.tile_begin; tb < tend; tb += 2 * DW_PF) {
#pragma unroll
    for (int p = 0; p < DW_PF; p++) {
      const int pn = p + 1 == DW_PF ? 0 : p + 1;
      if (!ILV) {
      stage(cur ^ 1, rr[pn]);                                   // block tb/2 + p + 1 into the other buffer
      load_block(role, tb + 2 * (p + 1 + DW_PF), rr[pn]);
      SCHED_FENCE();
      }
      const unsigned short* b0 = img + cur * BUFH;
      u16x8 at[NT][2], bt[NT][2];
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
          at[t][i] = *reinterpret_cast<const u16x8*>(b0 + t * IMG + (wm * 32 + i * 16) * 32 + fofs);
          bt[t][i] = *reinterpret_cast<const u16x8*>(b0 + (NT + t) * IMG + (wn * 32 + i * 16) * 32 + fofs);
        }
      if (ILV) {
        SCHED_FENCE();
        stage(cur ^ 1, rr[pn]);
        load_block(role, tb + 2 * (p + 1 + DW_PF), rr[pn]);
      }
      // smallest terms first (term weights 2^-8 apart): ta + tb = 2, then 1, then 0
#pragma unroll
      for (int w = NT - 1; w >= 0; w--)
#pragma unroll
        for (int ta = 0; ta <= w; ta++)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = mfma16x16x32_bf16(at[ta][i], bt[w - ta][j], acc[i][j]);
      if (ILV) {
        constexpr int NM = 2 * NT * (NT + 1);                    // MFMAs of a block: 12 (two terms) / 24 (three)
#pragma unroll
        for (int q = 0; q < NM; q++) {   // one MFMA, four conversion instructions; an LDS write every second, a load every sixth
          DW_SGB(0x008, 1);
          DW_SGB(0x002, 4);
          if (q % 2 == 0) DW_SGB(0x200, 1);
          if (q % 6 == 0) DW_SGB(0x020, 1);
        }
        SCHED_FENCE();
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  };
  if (isB) run(std::true_type{}); else run(std::false_type{});
  // epilogue through LDS: whole 256-byte row segments per store instruction (as gemm_mfma.h)
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        smem[(wm * 32 + i * 16 + (lane >> 4) * 4 + q) * GEMM_LDO + wn * 32 + j * 16 + (lane & 15)] = acc[i][j][q];
  __syncthreads();
  float* out = (extra ? a.xpartial : a.partial) + (size_t)sl.out_z * pR * pCn;
  const bool v4 = (pCn & 3) == 0 && ((size_t)out & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int rl = it * 16 + (tid >> 4), cl = (tid & 15) * 4;
    const int r = r0 + rl, c = c0 + cl;
    const f32x4 v = *reinterpret_cast<const f32x4*>(&smem[rl * GEMM_LDO + cl]);
    if (r < pR) {
      if (v4 && c + 3 < pCn) *reinterpret_cast<f32x4*>(out + (size_t)r * pCn + c) = v;
      else
        for (int e = 0; e < 4; e++)
          if (c + e < pCn) out[(size_t)r * pCn + c + e] = v[e];
    }
  }
  if (a.trace && tid == 0 && !extra) {
    long long* tr = a.trace + ((size_t)a.trace_base + si * (a.gx * a.gy) + tile) * 4;
    tr[0] = t_start; tr[1] = t_ready; tr[2] = wall_clock(); tr[3] = sl.need_it;
  }
}

// grid mode: workgroup `block` computes one item.  Workgroup b runs on XCD b % 8: XCD x takes slabs x, x+8, ... (all
// output tiles of a slab pull its frames through ONE L2), and because slabs are listed in readiness order every XCD
// gets early and late ones alike
// NT: the item form is a compile-time choice (and so part of the kernel's NAME: a profile of a process that runs several forms --
// bench.py's strict_f32 leg -- keeps their launch statistics apart): 0 the f32 MFMA (a.x3 == 0), 2 / 3 = a.terms of the split
template <int NT, bool ILV = false>
DEVFN void gemm_dw_body(const GemmDwArgs& a, float* smem, unsigned block) {
  constexpr bool X3 = NT != 0;
  if (block == 0) { gemm_dw_monitor(a); return; }   // the first workgroup behind the recurrence's watches the lines
  block -= 1;
  const unsigned nextra = X3 ? (unsigned)a.xnslabs * a.xgx * a.xgy : 0u;   // independent items first (see GemmDwArgs)
  if constexpr (X3) {
    if (block < nextra) { gemm_dw_item_x3<NT, ILV>(a, smem, block / (a.xgx * a.xgy), block % (a.xgx * a.xgy), true); return; }
  }
  block -= nextra;
  const unsigned tiles = a.gx * a.gy;
  const unsigned xcd = block & 7u, idx = block >> 3;
  const unsigned si = (idx / tiles) * 8u + xcd;
  if (si >= (unsigned)a.nslabs) return;
  if constexpr (X3) gemm_dw_item_x3<NT, ILV>(a, smem, si, idx % tiles);
  else gemm_dw_item(a, smem, si, idx % tiles);
}
template <int NT, bool ILV = false>
__global__ __launch_bounds__(256) void gemm_dw_kernel(GemmDwArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[dw_smem_floats(NT)];
  gemm_dw_body<NT, ILV>(a, smem, blockIdx.x);
}

}  // namespace clstm
