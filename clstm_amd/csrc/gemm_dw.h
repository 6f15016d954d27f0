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
#include "gemm_mfma.h"

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
  // persistent workers (lstm_bwd_dw.h): per-XCD work queues and the CUs the recurrence occupies
  int* qhead;               // [8] next item of XCD x (zeroed between launches by k_reduce_scatter)
  int* cu_busy;             // [8 * 256] = prog_base of this launch where a recurrence workgroup runs (hw_cu_slot())
};

constexpr int DW_WATCHDOG_POLLS = 1 << 16;

// ---- wait until every line has completed `need_it` iterations of direction `dir` (called by all 256 threads) -------
// Only wave 0 looks (one progress word per lane), and rarely: a poll is a system-scope load of up to 64 cache
// lines, and hundreds of workgroups polling every microsecond starve the recurrence's write-through stores
// (measured: 92 -> 630 us).  After each look the wave sleeps for most of the time the slowest line still needs
// (~0.45 us per iteration), so a workgroup polls a handful of times in all.
DEVFN void gemm_dw_wait(const GemmDwArgs& a, const int dir, const int need_it) {
  const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
  if (wave == 0) {
    int polls = 0;
    for (;;) {
      int deficit = 0;
      for (int b = lane; b < a.bs; b += 64) {
        const int T = a.line_off[b + 1] - a.line_off[b];
        const int need = a.prog_base + (need_it < T ? need_it : T);
        const int d = need - load_i32_wt(a.prog + ((size_t)dir * a.bs + b) * PROG_STRIDE);
        deficit = d > deficit ? d : deficit;
      }
      deficit = wave_max_i(deficit);
      if (deficit <= 0) break;
      if (++polls > DW_WATCHDOG_POLLS) {   // never hang the device: give up, flag it, produce garbage
        if (lane == 0) atomic_add_i32(a.timeouts, 1);
        break;
      }
      sleep_iterations(deficit);
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
// grid mode: workgroup `block` computes one item.  Workgroup b runs on XCD b % 8: XCD x takes slabs x, x+8, ... (all
// output tiles of a slab pull its frames through ONE L2), and because slabs are listed in readiness order every XCD
// gets early and late ones alike
DEVFN void gemm_dw_body(const GemmDwArgs& a, float* smem, const unsigned block) {
  const unsigned tiles = a.gx * a.gy;
  const unsigned xcd = block & 7u, idx = block >> 3;
  const unsigned si = (idx / tiles) * 8u + xcd;
  if (si >= (unsigned)a.nslabs) return;
  gemm_dw_item(a, smem, si, idx % tiles);
}
__global__ __launch_bounds__(256) void gemm_dw_kernel(GemmDwArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_BT * GEMM_LDO];
  gemm_dw_body(a, smem, blockIdx.x);
}

// persistent worker (a role of lstm_bwd_dw_kernel): pulls items of its XCD's queue until it is empty.  A worker that
// finds itself on a CU where a recurrence workgroup runs would only get the issue slots that workgroup leaves and
// hold its slab back; it waits until every line is complete and joins for the remainder.
DEVFN void gemm_dw_worker(const GemmDwArgs& a, float* smem, int* lds_item) {
  const int tid = threadIdx.x;
  const unsigned tiles = a.gx * a.gy;
  const int xcd = hw_xcc_id() & 7;
  const int nslabs_x = (a.nslabs - xcd + 7) / 8;
  const int nitems = nslabs_x * (int)tiles;
  sleep_iterations(8);    // ~3 us: the recurrence workgroups (dispatched first) have marked their CUs by now
  if (tid == 0) *lds_item = load_i32_wt(a.cu_busy + hw_cu_slot()) == a.prog_base ? 1 : 0;
  __syncthreads();
  const bool shared_cu = *lds_item != 0;
  __syncthreads();
  if (shared_cu) {
    for (int d = 0; d < a.ndir; d++) gemm_dw_wait(a, d, 0x3fffffff);
  }
  for (;;) {
    if (tid == 0) *lds_item = atomic_fetch_add_i32(a.qhead + xcd, 1);
    __syncthreads();
    const int item = *lds_item;
    __syncthreads();
    if (item >= nitems) break;
    gemm_dw_item(a, smem, (unsigned)(item / (int)tiles) * 8u + (unsigned)xcd, (unsigned)item % tiles);
    __syncthreads();
  }
}

}  // namespace clstm
