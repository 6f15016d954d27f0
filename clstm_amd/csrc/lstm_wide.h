// lstm_wide.h -- NPLSTM recurrence for hidden sizes whose recurrent weights do not fit the
// register file of one workgroup (no > 128; BASELINE config "2 x BiLSTM(512)").
//
// Same arithmetic as lstm_seq.h (GenericNPLSTM::forward / ::backward, clstm.cc:600-653), different
// parallelisation: the minibatch's lines advance in lock-step, one launch per time step, and the
// recurrent product of a step is a (lines x no) . (no x 4no) MFMA GEMM spread over the whole chip
// instead of a per-line mat-vec:
//
//   forward  step s : pre[m][cell,g] = G_x[frame_m(s)][cell,g] + sum_k h[frame_m(s-1)][k] R_g[cell][k]
//   backward step s : dh_rec[m][k]   = sum_{g,j} delta_g[frame_m(s+1)][j] R_g[j][k]
//
// A workgroup owns a 16-column tile of the product (forward: 4 cells x 4 gates, so the gate
// nonlinearities and the c/h update of those cells fuse into its epilogue; backward: 16 cells) for
// 16*MT lines.  Its four waves split the contraction range (split-K, v_mfma_f32_16x16x4_f32 fed
// straight from 16-byte buffer loads: lane (i, kq) loads four consecutive k of row i, and MFMA e of
// a group uses element e of both operands' float4 -- the k permutation is the same on both sides),
// partial tiles are summed through LDS, and thread (line, cell) finishes the step.
//
// Lines of different length simply drop out (their rows read as zeros through an out-of-range
// buffer offset and their epilogue is skipped).  `Reversed` is index arithmetic as in lstm_seq.h.
// Weights are repacked k-contiguous and zero-padded (ops.h:k_pack_wide), so operand rows may run
// past `no` into neighbouring finite data without masking.
#pragma once
#include "devintrin.h"

namespace clstm {

struct LstmWideArgs {
  const float* Rw;      // fwd: [dir][ceil(no/4)][16 = cell_local*4+gate][kp]   R_g[cell][k]
                        // bwd: [dir][ceil(no/16)][16 = cell k][kp]             R_g[j][k] at kk = 4*j+g
  long long rw_elems;
  float* G;             // [N][nd][no][4]  pre-activations in, activations out (forward)
  float* C;             // [N][nd][no]
  float* H;             // [N][ldh]        [.. 1 | h_dir0 | h_dir1], h at column hofs
  const float* dH;      // [N][nd*no]      (backward)
  float* D;             // [N][nd][no][4]  gate pre-activation deltas (backward)
  float* dC;            // [bs][nd][no]    carried state delta dc_{s+1} * gf_{s+1} (backward)
  const int* line_off;  // [bs+1]
  float* S;             // [nd][N][lds] source rows (see lstm_seq.h)
  long long sdir;
  long long N;          // frames in the batch (array extents)
  int lds, sofs, ldh, hofs;
  int no, ndir, bs;
  int kp;               // padded contraction length, multiple of 64
  int step;             // lock-step index: forward own step s = step, backward own step s = T-1-step
  int tmax;             // cooperative kernels: number of lock-steps (longest line)
  int zb0, zbn;         // persistent per-XCD kernels: this launch walks the 16-line blocks [zb0, zb0 + zbn), zbn * ndir <= 8
  int* sync;            // cooperative kernels: [0] barrier ticket counter (zeroed per launch), [1] watchdog flag
  // bf16 MFMA operands (per-step kernels lstm_wide_*_step_bf16; BASELINE config "2 x BiLSTM(512), bf16 MFMA"):
  const unsigned short* Rw16;   // the same weight rows as Rw, bf16, row length kp16
  unsigned short* Hb;           // [N][nd][kp16]  bf16 copy of h (forward A operand), pad columns stay zero
  unsigned short* Db;           // [N][nd][kp16]  bf16 copy of the gate deltas at column 4*cell+gate (backward A operand)
  unsigned short* Hbf;          // persistent forward kernel: per-frame [N][hbf_ld] bf16 copy of h (dir d at column d*no): the next
  int hbf_ld;                   //   layer's W_x product reads it as its k-contiguous A operand; or null
  unsigned epoch;               // persistent forward kernel with the tagged ring: launch number (tags = epoch << 12 | step)
  int skip_d;                   // persistent backward kernel: the f32 deltas D are not stored (every consumer reads Dbf; the host expands Dbf if one does not)
  unsigned short* Sbf;          // persistent forward kernel: bf16 source rows [x | h_{t-1} | 1] of THIS layer, [dir][N][sbf_ld] (h-part written here), or null
  int sbf_ld, sbf_ofs; long long sbf_dir;
  unsigned short* Dbf;          // persistent backward kernel: per-frame [N][nd][kp16] bf16 deltas (operand of the x.d GEMM), or null
  int kp16;                     // padded contraction length of the bf16 rows, multiple of 32 * WIDE_NW
};

constexpr int WIDE_LDW = 20;  // LDS row stride of a partial tile (16 columns + pad, float4-aligned)
constexpr int WIDE_PF = 4;    // 16-k groups in flight per wave, per-step kernels (8 measured slower: 200 VGPRs)
constexpr int WIDE_PF_COOP = 8;  // cooperative kernels: all groups of a 512-cell layer in flight at once
constexpr int WIDE_NW = 4;    // waves per workgroup = split-K factor (8 was measured on MI355X at 512 cells: 9.9 / 8.5 ms
                              // against 9.7 / 8.3 ms per pass -- the step is not bound by the MFMA/load rounds of a wave)
constexpr int WIDE_THREADS = 64 * WIDE_NW;
constexpr int WIDE_WPAD = 4;  // LDS weight rows are kp + 4 floats: 16 rows x b128 reads cover all banks once

// acc[i] += A_i(16 rows x kslice) . B(kslice x 16 cols) for this wave's quarter of the contraction,
// then the four waves' partial tiles are left in red[wave][row][col] (caller syncs).
// COOP = false: B rows come from global memory (bbuf/brow), A rows by plain loads (per-step launches).
// COOP = true : B rows are resident in LDS (wl, row stride kp + WIDE_WPAD), A rows were written by OTHER
//               workgroups of the same launch -> device-scope (sc1) loads.
template <int MT, int NT, bool COOP>
DEVFN void wide_tile(const BufF32 abuf, const unsigned (&arow)[MT], const BufF32 bbuf, const unsigned brow,
                     const float* wl, const int kp, float* red) {
  constexpr int PF = COOP ? WIDE_PF_COOP : WIDE_PF;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int kw = kp / WIDE_NW;            // contraction range of one wave (multiple of 16)
  const int ngroups = kw >> 4;
  const unsigned klane = (unsigned)(wave * kw + 4 * (lane >> 4)) * 4u;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;
  f32x4 ra[PF][MT], rb[PF][NT];
  // unconditional issue (groups past the end get out-of-range offsets): exact vmcnt, see gemm_mfma.h
  auto load_group = [&](int g, f32x4 (&a)[MT], f32x4 (&b)[NT]) {
    const bool live = g < ngroups;
    const unsigned ko = klane + (unsigned)g * 64u;
#pragma unroll
    for (int i = 0; i < MT; i++)
      a[i] = COOP ? buf_load4_dev(abuf, live ? arow[i] + ko : BUF_OOB) : buf_load4(abuf, live ? arow[i] + ko : BUF_OOB);
    if (!COOP) {
#pragma unroll
      for (int j = 0; j < NT; j++) b[j] = buf_load4(bbuf, live ? brow + (unsigned)j * 16u * (unsigned)kp * 4u + ko : BUF_OOB);
    }
  };
  const int ldw = kp + WIDE_WPAD;
  const float* wrow = COOP ? wl + (lane & 15) * ldw + (klane >> 2) : nullptr;
#pragma unroll
  for (int p = 0; p < PF; p++) {
    load_group(p, ra[p], rb[p]);
    SCHED_FENCE();
  }
  for (int g0 = 0; g0 < ngroups; g0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; p++) {
      f32x4 av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; i++) av[i] = ra[p][i];
      if (COOP) {  // groups past the end multiply zero A rows: any in-range weight group will do
        const int g = g0 + p < ngroups ? g0 + p : ngroups - 1;
#pragma unroll
        for (int j = 0; j < NT; j++) bv[j] = *reinterpret_cast<const f32x4*>(wrow + j * 16 * ldw + g * 16);
      } else {
#pragma unroll
        for (int j = 0; j < NT; j++) bv[j] = rb[p][j];
      }
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
          for (int j = 0; j < NT; j++) acc[i][j] = mfma16x16x4(av[i][e], bv[j][e], acc[i][j]);
      load_group(g0 + p + PF, ra[p], rb[p]);
      SCHED_FENCE();
    }
  }
  // D layout: lane holds rows (lane>>4)*4 + q, column lane&15; red row stride NT*16 + 4
  constexpr int LDR = NT * 16 + 4;
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        red[((wave * MT + i) * 16 + (lane >> 4) * 4 + q) * LDR + j * 16 + (lane & 15)] = acc[i][j][q];
}

// The same tile with bf16 operands on v_mfma_f32_16x16x32_bf16 (f32 accumulation): a lane's fragment of a k32 group is
// ONE 16-byte load -- 8 consecutive k of its row -- for A (bf16 copies of h / the deltas) and B (bf16 weight rows)
// alike; a wave owns a quarter of the contraction range, partial tiles meet in `red` exactly as above.  One eighth of
// the f32 kernel's MFMA instructions and half its operand bytes per step.
// `after_prologue` runs once the first PF groups are requested: the place for loads whose latency is longer than the
// tile's own (epilogue operands from HBM) -- VMEM returns in order, so anything issued BEFORE the tile's loads would be
// waited for by the first MFMA.
template <int MT, int PF, class AP>
DEVFN void wide_tile_bf16(const BufF32 abuf, const unsigned (&arow)[MT], const BufF32 bbuf, const unsigned brow, const int kp,
                          float* red, AP after_prologue) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int kw = kp / WIDE_NW;            // contraction range of one wave (multiple of 32)
  const int ngroups = kw >> 5;
  const unsigned klane = (unsigned)(wave * kw + 8 * (lane >> 4)) * 2u;   // bytes
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[i][q] = 0.0f;
  f32x4 ra[PF][MT], rb[PF];
  auto load_group = [&](int g, f32x4 (&a)[MT], f32x4& b) {   // unconditional issue: exact vmcnt
    const bool live = g < ngroups;
    const unsigned ko = klane + (unsigned)g * 64u;
#pragma unroll
    for (int i = 0; i < MT; i++) a[i] = buf_load4(abuf, live ? arow[i] + ko : BUF_OOB);
    b = buf_load4(bbuf, live ? brow + ko : BUF_OOB);
  };
#pragma unroll
  for (int p = 0; p < PF; p++) {
    load_group(p, ra[p], rb[p]);
    SCHED_FENCE();
  }
  after_prologue();
  SCHED_FENCE();
  for (int g0 = 0; g0 < ngroups; g0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; p++) {
      const u16x8 bv = __builtin_bit_cast(u16x8, rb[p]);
#pragma unroll
      for (int i = 0; i < MT; i++) acc[i] = mfma16x16x32_bf16(__builtin_bit_cast(u16x8, ra[p][i]), bv, acc[i]);
      load_group(g0 + p + PF, ra[p], rb[p]);
      SCHED_FENCE();
    }
  }
  constexpr int LDR = 16 + 4;
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int q = 0; q < 4; q++) red[((wave * MT + i) * 16 + (lane >> 4) * 4 + q) * LDR + (lane & 15)] = acc[i][q];
}

#ifndef CLSTM_WEXP   // perf experiments only: bit mask of work to leave out of the forward step (results are then wrong)
#define CLSTM_WEXP 0
#endif
// 16 lines x 64 columns (four 16-row weight tiles, `btile` bytes apart) on the same instruction: one A fragment serves
// four MFMAs, so a workgroup pulls a quarter of the FRESH bytes (h_{t-1}: written one launch ago by other XCDs, an L2
// miss) per product column; the weight rows it reads instead are L2-resident for the whole sequence.
template <int PF, class AP>
DEVFN void wide_tile_bf16_n4(const BufF32 abuf, const unsigned arow, const BufF32 bbuf, const unsigned brow,
                             const unsigned btile, const int kp, float* red, AP after_prologue) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int kw = kp / WIDE_NW;
  const int ngroups = kw >> 5;
  const unsigned klane = (unsigned)(wave * kw + 8 * (lane >> 4)) * 2u;   // bytes
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[j][q] = 0.0f;
  f32x4 ra[PF], rb[PF][4];
  auto load_group = [&](int g, f32x4& a, f32x4 (&b)[4]) {   // unconditional issue: exact vmcnt
    const bool live = g < ngroups;
    const unsigned ko = klane + (unsigned)g * 64u;
    a = buf_load4(abuf, live ? arow + ko : BUF_OOB);
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = buf_load4(bbuf, live ? brow + (unsigned)j * btile + ko : BUF_OOB);
  };
#pragma unroll
  for (int p = 0; p < PF; p++) {
    load_group(p, ra[p], rb[p]);
    SCHED_FENCE();
  }
  after_prologue();
  SCHED_FENCE();
  for (int g0 = 0; g0 < ngroups; g0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; p++) {
      const u16x8 av = __builtin_bit_cast(u16x8, ra[p]);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = mfma16x16x32_bf16(av, __builtin_bit_cast(u16x8, rb[p][j]), acc[j]);
      load_group(g0 + p + PF, ra[p], rb[p]);
      SCHED_FENCE();
    }
  }
  constexpr int LDR = 64 + 4;
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) red[(wave * 16 + (lane >> 4) * 4 + q) * LDR + j * 16 + (lane & 15)] = acc[j][q];
}

// XCD-aware order of a launch's work items (same bijection as gemm_mfma.h): the dispatcher places workgroup `lin` on
// XCD lin % 8; XCD x gets the CONTIGUOUS run of items [x*q ...), so with the cell tile as the fastest item index all
// tiles of one (line block, direction) -- which read the same fresh h rows and write neighbouring cells of the same
// cache lines -- sit behind one L2.
DEVFN unsigned xcd_contiguous_item(const unsigned lin, const unsigned total) {
  const unsigned xcd = lin & 7u, idx = lin >> 3;
  const unsigned q = total >> 3, r = total & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- forward, bf16 operands: one time step of 16 lines x (16 cells x 4 gates) ----------------------------------------
// What bounds a per-step launch at 512 cells is not the MFMA work (16 instructions per wave) but the bytes that are
// new since the previous launch and the shape of the stores (measured by leaving parts out, 64 lines x 400 frames:
// 9.45 us per step, of which stores 3.2, h loads 2.2, other epilogue loads 0.9, epilogue math 0.8, empty launch 2.6):
//  * the bf16 copy of h lives in a LOCK-STEP ring Hb[step parity][dir][line][kp16] -- its address does not depend on
//    the line offsets, so the loads are the first instructions of the kernel, not behind a dependent load;
//  * 16 cells per workgroup and the XCD-contiguous item order: every store instruction of a wave writes 16 x 64 B
//    (C, H, S) or 16 x 256 B (G) runs, and the cache lines of a frame's state are completed inside ONE L2 (with 4 cells
//    per workgroup and round-robin placement eight XCDs each wrote 16 B of every 128-B line).
DEVFN void wide_fwd_tile16_bf16(const LstmWideArgs& a, const int sg, const int ct, const int dir, const int zb, float* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int no = a.no, nd = a.ndir;
  const int ncg = (no + 3) >> 2;
  const int m = zb * 16 + (lane & 15);
  const unsigned arow = (sg >= 1 && m < a.bs && !(CLSTM_WEXP & 1)) ? (unsigned)(((((sg - 1) & 1) * nd + dir) * a.bs + m) * a.kp16) * 2u : BUF_OOB_BASE;
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Hb), (size_t)2 * nd * a.bs * a.kp16 * 2);
  const BufF32 bbuf = make_buf(reinterpret_cast<const float*>(a.Rw16), (size_t)a.rw_elems * 2);
  const unsigned brow = (CLSTM_WEXP & 2) ? BUF_OOB_BASE : (unsigned)(((long long)(dir * ncg + ct * 4) * 16 + (lane & 15)) * a.kp16) * 2u;   // rows past the
  const unsigned btile = (unsigned)(16 * a.kp16) * 2u;                   // last cell group: dropped by the descriptor

  const int ml = tid >> 4, c16 = tid & 15;
  const int line = zb * 16 + ml, cell = ct * 16 + c16;
  // the line's offsets are REQUESTED first and USED behind the tile's loads (no branch, no early wait)
  const BufF32 lbuf = make_buf(reinterpret_cast<const float*>(a.line_off), (size_t)(a.bs + 1) * 4);
  const float lo0 = buf_load(lbuf, line < a.bs ? (unsigned)line * 4u : BUF_OOB);
  const float lo1 = buf_load(lbuf, line < a.bs ? (unsigned)line * 4u + 4u : BUF_OOB);
  SCHED_FENCE();
  bool live;
  int off, T;
  long long n;
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  f32x4 gx;
  float c_prev;
  wide_tile_bf16_n4<4>(abuf, arow, bbuf, brow, btile, a.kp16, red, [&]() {
    off = __builtin_bit_cast(int, lo0);
    T = __builtin_bit_cast(int, lo1) - off;
    live = line < a.bs && cell < no && sg < T && !(CLSTM_WEXP & 32);
    n = off + (dir == 0 ? sg : T - 1 - sg);
    gx = buf_load4(gbuf, live ? (unsigned)(((n * nd + dir) * no + cell) * 16) : BUF_OOB);
    c_prev = buf_load(cbuf, live && sg >= 1
        ? (unsigned)((((long long)(off + (dir == 0 ? sg - 1 : T - sg)) * nd + dir) * no + cell) * 4) : BUF_OOB);
  });
  __syncthreads();

  float h = 0.0f;
  if (CLSTM_WEXP & 32) live = line < a.bs && cell < no && sg < T;
  if (live && !(CLSTM_WEXP & 16)) {
    f32x4 k;
#pragma unroll
    for (int q = 0; q < 4; q++) k[q] = 0.0f;
#pragma unroll
    for (int w = 0; w < WIDE_NW; w++) {   // columns of cell c16: cell group c16>>2, slot (c16&3)*4 + gate
      const f32x4 p = *reinterpret_cast<const f32x4*>(&red[(w * 16 + ml) * 68 + c16 * 4]);
#pragma unroll
      for (int q = 0; q < 4; q++) k[q] += p[q];
    }
    const float gi = gate_act(k[0] + gx[0], false), gf = gate_act(k[1] + gx[1], false),
                go = gate_act(k[2] + gx[2], false), ci = gate_act(k[3] + gx[3], true);
    const float c = ci * gi + gf * c_prev;      // c_prev reads 0 at the first step
    h = gate_act(c, true) * go;
    f32x4 act;
    act[0] = gi; act[1] = gf; act[2] = go; act[3] = ci;
    if ((CLSTM_WEXP & 8) && h != 12345.678f) return;
    *reinterpret_cast<f32x4*>(a.G + ((n * nd + dir) * no + cell) * 4) = act;
    a.C[(n * nd + dir) * no + cell] = c;
    a.H[n * a.ldh + a.hofs + dir * no + cell] = h;
    float* srow = a.S + (size_t)dir * a.sdir;
    if (sg == 0) srow[n * a.lds + a.sofs + cell] = 0.0f;                      // h_{-1} = 0
    if (sg + 1 < T) srow[(long long)(off + (dir == 0 ? sg + 1 : T - 2 - sg)) * a.lds + a.sofs + cell] = h;
  }
  // next step's A operand: two cells per 4-byte store (the odd lane's h comes over by DPP; every lane takes part)
  const float hn = quad_xor1(h);
  if (live && !(c16 & 1))
    *reinterpret_cast<unsigned*>(a.Hb + ((size_t)((sg & 1) * nd + dir) * a.bs + line) * a.kp16 + cell) =
        bf16_pack2(h, cell + 1 < no ? hn : 0.0f);
}

// grid: ceil(no/16) * ndir * ceil(bs/16) workgroups (1-D), 256 threads
__global__ __launch_bounds__(WIDE_THREADS) void lstm_wide_fwd_step16_bf16(LstmWideArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WIDE_NW * 16 * 68];
  const int ntile = (a.no + 15) >> 4;
  const unsigned v = xcd_contiguous_item(blockIdx.x, gridDim.x);
  const int ct = (int)(v % (unsigned)ntile), dir = (int)((v / (unsigned)ntile) % (unsigned)a.ndir),
            zb = (int)(v / (unsigned)(ntile * a.ndir));
  wide_fwd_tile16_bf16(a, a.step, ct, dir, zb, red);
}

// ---- forward: one time step of 16*MT lines for one (cell group, direction) ------------------------
// loff: line offsets (global for the per-step launch, an LDS copy in the cooperative kernel)
template <int MT, bool COOP, bool BF16 = false>
DEVFN void wide_fwd_tile(const LstmWideArgs& a, const int sg, const int cg, const int dir, const int zb,
                         const int* loff, const float* wl, float* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int no = a.no, nd = a.ndir;
  const int ncg = (no + 3) >> 2;

  unsigned arow[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) {
    const int m = (zb * MT + i) * 16 + (lane & 15);
    arow[i] = BUF_OOB_BASE;
    if (m < a.bs) {
      const int off = loff[m], T = loff[m + 1] - off;
      if (sg >= 1 && sg < T) {
        const int fprev = dir == 0 ? sg - 1 : T - sg;   // frame of own step s-1
        arow[i] = BF16 ? (unsigned)((((long long)(off + fprev) * nd + dir) * a.kp16) * 2)
                       : (unsigned)((long long)(off + fprev) * a.ldh + a.hofs + dir * no) * 4u;
      }
    }
    if (CLSTM_WEXP & 1) arow[i] = BUF_OOB_BASE;
  }
  const BufF32 abuf = BF16 ? make_buf(reinterpret_cast<const float*>(a.Hb), (size_t)a.N * nd * a.kp16 * 2)
                            : make_buf(a.H, (size_t)a.N * a.ldh * 4);
  const BufF32 bbuf = BF16 ? make_buf(reinterpret_cast<const float*>(a.Rw16), (size_t)a.rw_elems * 2)
                            : make_buf(a.Rw, (size_t)a.rw_elems * 4);
  const unsigned brow = (CLSTM_WEXP & 2) ? BUF_OOB_BASE
                        : BF16 ? (unsigned)(((long long)(dir * ncg + cg) * 16 + (lane & 15)) * a.kp16) * 2u
                               : (unsigned)(((long long)(dir * ncg + cg) * 16 + (lane & 15)) * a.kp) * 4u;

  // epilogue role of this thread: (line, cell); its operands are requested before the MFMA loop so
  // that their HBM latency hides under it (masked threads read nothing: out-of-range offsets)
  const int ml = tid >> 2, cl = tid & 3;
  const int line = zb * MT * 16 + ml, cell = cg * 4 + cl;
  bool live = ml < MT * 16 && line < a.bs && cell < no;
  int off = 0, T = 0;
  if (live) {
    off = loff[line];
    T = loff[line + 1] - off;
    live = sg < T;
  }
  const long long n = off + (dir == 0 ? sg : T - 1 - sg);
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  const unsigned goff = live && !(CLSTM_WEXP & 32) ? (unsigned)(((n * nd + dir) * no + cell) * 16) : BUF_OOB;
  const f32x4 gx = buf_load4(gbuf, goff);
  // c_{s-1} was written by this very thread one step ago (same (line, cell) role), so a plain load is
  // coherent in the cooperative kernel as well
  const float c_prev = buf_load(cbuf, live && sg >= 1 && !(CLSTM_WEXP & 32)
      ? (unsigned)((((long long)(off + (dir == 0 ? sg - 1 : T - sg)) * nd + dir) * no + cell) * 4) : BUF_OOB);

  if (BF16) wide_tile_bf16<MT, 4>(abuf, arow, bbuf, brow, a.kp16, red, []() {});
  else wide_tile<MT, 1, COOP>(abuf, arow, bbuf, brow, wl, a.kp, red);
  __syncthreads();

  // fused forward_full1 x4 + forward_statemem + forward_nonlingate for (line, cell)
  if (live && !(CLSTM_WEXP & 16)) {
    f32x4 k;
#pragma unroll
    for (int q = 0; q < 4; q++) k[q] = 0.0f;
#pragma unroll
    for (int w = 0; w < WIDE_NW; w++) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(&red[((w * MT + (ml >> 4)) * 16 + (ml & 15)) * WIDE_LDW + cl * 4]);
#pragma unroll
      for (int q = 0; q < 4; q++) k[q] += p[q];
    }
    const float gi = gate_act(k[0] + gx[0], false), gf = gate_act(k[1] + gx[1], false),
                go = gate_act(k[2] + gx[2], false), ci = gate_act(k[3] + gx[3], true);
    const float c = ci * gi + gf * c_prev;      // c_prev reads 0 at the first step
    const float h = gate_act(c, true) * go;
    f32x4 act;
    act[0] = gi; act[1] = gf; act[2] = go; act[3] = ci;
    if ((CLSTM_WEXP & 8) && h != 12345.678f) return;
    *reinterpret_cast<f32x4*>(a.G + ((n * nd + dir) * no + cell) * 4) = act;
    a.C[(n * nd + dir) * no + cell] = c;
    // h_t is next step's A operand of every workgroup of this direction
    if (COOP) buf_store_dev(abuf, (unsigned)(n * a.ldh + a.hofs + dir * no + cell) * 4u, h);
    else a.H[n * a.ldh + a.hofs + dir * no + cell] = h;
    if (BF16) a.Hb[(n * nd + dir) * a.kp16 + cell] = (unsigned short)(bf16_pack2(h, 0.0f) & 0xFFFFu);   // next step's A operand
    float* srow = a.S + (size_t)dir * a.sdir;
    if (sg == 0) srow[n * a.lds + a.sofs + cell] = 0.0f;                      // h_{-1} = 0
    if (sg + 1 < T) srow[(long long)(off + (dir == 0 ? sg + 1 : T - 2 - sg)) * a.lds + a.sofs + cell] = h;
  }
}

// per-step launch: grid (ceil(no/4), ndir, ceil(bs / 16MT)), 256 threads
template <int MT>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_wide_fwd_step(LstmWideArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WIDE_NW * MT * 16 * WIDE_LDW];
  wide_fwd_tile<MT, false>(a, a.step, blockIdx.x, blockIdx.y, blockIdx.z, a.line_off, nullptr, red);
}

template <int MT>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_wide_fwd_step_bf16(LstmWideArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WIDE_NW * MT * 16 * WIDE_LDW];
  wide_fwd_tile<MT, false, true>(a, a.step, blockIdx.x, blockIdx.y, blockIdx.z, a.line_off, nullptr, red);
}

// ---- backward: one time step of 16 lines for one (16-cell tile, direction) ------------------------
#ifndef CLSTM_WEXPB   // perf experiments only: bit mask of work to leave out of the backward step (results are then wrong)
#define CLSTM_WEXPB 0
#endif
template <bool COOP, bool BF16 = false>
DEVFN void wide_bwd_tile(const LstmWideArgs& a, const int sg, const int ct, const int dir, const int zb,
                         const int* loff, const float* wl, float* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int no = a.no, nd = a.ndir;
  const int nct = (no + 15) >> 4;

  unsigned arow[1];
  {
    const int m = zb * 16 + (lane & 15);
    arow[0] = BUF_OOB_BASE;
    if (m < a.bs) {
      const int off = loff[m], T = loff[m + 1] - off;
      if (sg >= 1 && sg < T) {
        const int fnext = dir == 0 ? T - sg : sg - 1;   // frame of own step s+1, s = T-1-sg
        arow[0] = BF16 ? (unsigned)((((long long)(off + fnext) * nd + dir) * a.kp16) * 2)
                       : (unsigned)(((long long)(off + fnext) * nd + dir) * 4 * no) * 4u;
      }
    }
    if (CLSTM_WEXPB & 1) arow[0] = BUF_OOB_BASE;
  }
  const BufF32 abuf = BF16 ? make_buf(reinterpret_cast<const float*>(a.Db), (size_t)a.N * nd * a.kp16 * 2)
                            : make_buf(a.D, (size_t)a.N * nd * 4 * no * 4);
  const BufF32 bbuf = BF16 ? make_buf(reinterpret_cast<const float*>(a.Rw16), (size_t)a.rw_elems * 2)
                            : make_buf(a.Rw, (size_t)a.rw_elems * 4);
  const unsigned brow = (CLSTM_WEXPB & 2) ? BUF_OOB_BASE
                        : BF16 ? (unsigned)(((long long)(dir * nct + ct) * 16 + (lane & 15)) * a.kp16) * 2u
                               : (unsigned)(((long long)(dir * nct + ct) * 16 + (lane & 15)) * a.kp) * 4u;

  // epilogue operands of thread (line, cell), requested ahead of the MFMA loop
  const int ml = tid >> 4, c16 = tid & 15;
  const int line = zb * 16 + ml, cell = ct * 16 + c16;
  bool live = ml < 16 && line < a.bs && cell < no;
  int off = 0, T = 0;
  if (live) {
    off = loff[line];
    T = loff[line + 1] - off;
    live = sg < T;
  }
  const bool lv = live;
  if (CLSTM_WEXPB & 32) live = false;
  const int s = T - 1 - sg;
  const long long n = off + (dir == 0 ? s : sg);
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  const BufF32 hbuf = make_buf(a.dH, (size_t)a.N * nd * no * 4);
  const BufF32 dcbuf = make_buf(a.dC, (size_t)a.bs * nd * no * 4);
  const unsigned coff = (unsigned)(((n * nd + dir) * no + cell) * 4);
  const f32x4 act = buf_load4(gbuf, live ? coff * 4u : BUF_OOB);
  const float dh_in = buf_load(hbuf, live ? (unsigned)((n * (nd * no) + dir * no + cell) * 4) : BUF_OOB);
  const float c_s = buf_load(cbuf, live ? coff : BUF_OOB);
  const float c_m1 = buf_load(cbuf, live && s >= 1       // c_{s-1}; 0 at s = 0 ("gf.d untouched when last < 0")
      ? (unsigned)((((long long)(off + (dir == 0 ? s - 1 : sg + 1)) * nd + dir) * no + cell) * 4) : BUF_OOB);
  const unsigned dcoff = (unsigned)((((long long)line * nd + dir) * no + cell) * 4);
  const float dc_carry = buf_load(dcbuf, live && sg >= 1 ? dcoff : BUF_OOB);   // own write of the previous step

  if (BF16) wide_tile_bf16<1, 8>(abuf, arow, bbuf, brow, a.kp16, red, []() {});
  else wide_tile<1, 1, COOP>(abuf, arow, bbuf, brow, wl, a.kp, red);
  __syncthreads();
  live = lv;

  if (live && !(CLSTM_WEXPB & 16)) {
    float dh_rec = 0.0f;
#pragma unroll
    for (int w = 0; w < WIDE_NW; w++) dh_rec += red[(w * 16 + ml) * WIDE_LDW + c16];
    const float gi = act[0], gf = act[1], go = act[2], ci = act[3];
    const float dh = dh_in + dh_rec;           // out[s].d, clstm.cc:626-628 + :646
    const float th = gate_act(c_s, true);      // backward_nonlingate recomputes tanh(state)
    const float d_go = th * dh;
    const float dc = dc_carry + (-th * th + 1.0f) * (go * dh);
    a.dC[dcoff / 4] = dc * gf;                 // backward_statemem (clstm_compute.cc:509-515)
    const float d_gf = dc * c_m1;
    const float d_gi = dc * ci, d_ci = dc * gi;
    f32x4 dl;                                  // backward_nonlin0: y(1-y) for SIG, 1-y^2 for TANH
    dl[0] = (gi * (-gi + 1.0f)) * d_gi;
    dl[1] = (gf * (-gf + 1.0f)) * d_gf;
    dl[2] = (go * (-go + 1.0f)) * d_go;
    dl[3] = (-ci * ci + 1.0f) * d_ci;
    if ((CLSTM_WEXPB & 8) && dl[3] != 12345.678f) return;
    // the deltas are next step's A operand of every workgroup of this direction
    if (COOP) buf_store4_dev(abuf, coff * 4u, dl);
    else *reinterpret_cast<f32x4*>(a.D + ((n * nd + dir) * no + cell) * 4) = dl;
    if (BF16) {   // next step's A operand
      unsigned* db = reinterpret_cast<unsigned*>(a.Db + (n * nd + dir) * a.kp16 + 4 * cell);
      db[0] = bf16_pack2(dl[0], dl[1]);
      db[1] = bf16_pack2(dl[2], dl[3]);
    }
  }
}
// ---- backward, bf16 operands: the same three measures as wide_fwd_tile16_bf16 (measured by leaving parts out: 7.75 us
// per step = delta loads 1.4 + weight loads 1.3 + epilogue operand loads 1.3 + epilogue 1.0 + empty launch 2.8) --
// the bf16 deltas in a lock-step ring Db[step parity][dir][line][kp16], XCD-contiguous item order (one XCD: the 16 lines'
// 64 KB of fresh deltas once, one direction's 2 MB of weight rows resident in its L2), and the epilogue's operands --
// forward-pass arrays that come from HBM -- requested BEHIND the first round of tile loads.
DEVFN void wide_bwd_tile16_bf16(const LstmWideArgs& a, const int sg, const int ct, const int dir, const int zb, float* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int no = a.no, nd = a.ndir;
  const int nct = (no + 15) >> 4;
  unsigned arow[1];
  {
    const int m = zb * 16 + (lane & 15);
    arow[0] = (sg >= 1 && m < a.bs) ? (unsigned)(((((sg - 1) & 1) * nd + dir) * a.bs + m) * a.kp16) * 2u : BUF_OOB_BASE;
  }
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Db), (size_t)2 * nd * a.bs * a.kp16 * 2);
  const BufF32 bbuf = make_buf(reinterpret_cast<const float*>(a.Rw16), (size_t)a.rw_elems * 2);
  const unsigned brow = (unsigned)(((long long)(dir * nct + ct) * 16 + (lane & 15)) * a.kp16) * 2u;

  const int ml = tid >> 4, c16 = tid & 15;
  const int line = zb * 16 + ml, cell = ct * 16 + c16;
  const BufF32 lbuf = make_buf(reinterpret_cast<const float*>(a.line_off), (size_t)(a.bs + 1) * 4);
  const float lo0 = buf_load(lbuf, line < a.bs ? (unsigned)line * 4u : BUF_OOB);   // requested first, used behind the
  const float lo1 = buf_load(lbuf, line < a.bs ? (unsigned)line * 4u + 4u : BUF_OOB);   // tile's loads
  SCHED_FENCE();
  bool live;
  int off, T, s;
  long long n;
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  const BufF32 hbuf = make_buf(a.dH, (size_t)a.N * nd * no * 4);
  const BufF32 dcbuf = make_buf(a.dC, (size_t)a.bs * nd * no * 4);
  unsigned coff;
  const unsigned dcoff = (unsigned)((((long long)line * nd + dir) * no + cell) * 4);
  f32x4 act;
  float dh_in, c_s, c_m1, dc_carry;
  wide_tile_bf16<1, 8>(abuf, arow, bbuf, brow, a.kp16, red, [&]() {
    off = __builtin_bit_cast(int, lo0);
    T = __builtin_bit_cast(int, lo1) - off;
    live = line < a.bs && cell < no && sg < T;
    s = T - 1 - sg;
    n = off + (dir == 0 ? s : sg);
    coff = (unsigned)(((n * nd + dir) * no + cell) * 4);
    act = buf_load4(gbuf, live ? coff * 4u : BUF_OOB);
    dh_in = buf_load(hbuf, live ? (unsigned)((n * (nd * no) + dir * no + cell) * 4) : BUF_OOB);
    c_s = buf_load(cbuf, live ? coff : BUF_OOB);
    c_m1 = buf_load(cbuf, live && s >= 1       // c_{s-1}; 0 at s = 0 ("gf.d untouched when last < 0")
        ? (unsigned)((((long long)(off + (dir == 0 ? s - 1 : sg + 1)) * nd + dir) * no + cell) * 4) : BUF_OOB);
    dc_carry = buf_load(dcbuf, live && sg >= 1 ? dcoff : BUF_OOB);   // own write of the previous step
  });
  __syncthreads();

  if (live) {
    float dh_rec = 0.0f;
#pragma unroll
    for (int w = 0; w < WIDE_NW; w++) dh_rec += red[(w * 16 + ml) * WIDE_LDW + c16];
    const float gi = act[0], gf = act[1], go = act[2], ci = act[3];
    const float dh = dh_in + dh_rec;           // out[s].d, clstm.cc:626-628 + :646
    const float th = gate_act(c_s, true);      // backward_nonlingate recomputes tanh(state)
    const float d_go = th * dh;
    const float dc = dc_carry + (-th * th + 1.0f) * (go * dh);
    a.dC[dcoff / 4] = dc * gf;                 // backward_statemem (clstm_compute.cc:509-515)
    const float d_gf = dc * c_m1;
    const float d_gi = dc * ci, d_ci = dc * gi;
    f32x4 dl;                                  // backward_nonlin0: y(1-y) for SIG, 1-y^2 for TANH
    dl[0] = (gi * (-gi + 1.0f)) * d_gi;
    dl[1] = (gf * (-gf + 1.0f)) * d_gf;
    dl[2] = (go * (-go + 1.0f)) * d_go;
    dl[3] = (-ci * ci + 1.0f) * d_ci;
    *reinterpret_cast<f32x4*>(a.D + ((n * nd + dir) * no + cell) * 4) = dl;
    unsigned* db = reinterpret_cast<unsigned*>(a.Db + ((size_t)((sg & 1) * nd + dir) * a.bs + line) * a.kp16 + 4 * cell);
    db[0] = bf16_pack2(dl[0], dl[1]);          // next step's A operand
    db[1] = bf16_pack2(dl[2], dl[3]);
  }
}
// grid: ceil(no/16) * ndir * ceil(bs/16) workgroups (1-D), 256 threads
__global__ __launch_bounds__(WIDE_THREADS) void lstm_wide_bwd_step16_bf16(LstmWideArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WIDE_NW * 16 * WIDE_LDW];
  const int ntile = (a.no + 15) >> 4;
  const unsigned v = xcd_contiguous_item(blockIdx.x, gridDim.x);
  const int ct = (int)(v % (unsigned)ntile), dir = (int)((v / (unsigned)ntile) % (unsigned)a.ndir),
            zb = (int)(v / (unsigned)(ntile * a.ndir));
  wide_bwd_tile16_bf16(a, a.step, ct, dir, zb, red);
}

__global__ __launch_bounds__(WIDE_THREADS) void lstm_wide_bwd_step_bf16(LstmWideArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WIDE_NW * 16 * WIDE_LDW];
  wide_bwd_tile<false, true>(a, a.step, blockIdx.x, blockIdx.y, blockIdx.z, a.line_off, nullptr, red);
}

// per-step launch: grid (ceil(no/16), ndir, ceil(bs/16)), 256 threads
__global__ __launch_bounds__(WIDE_THREADS) void lstm_wide_bwd_step(LstmWideArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WIDE_NW * 16 * WIDE_LDW];
  wide_bwd_tile<false>(a, a.step, blockIdx.x, blockIdx.y, blockIdx.z, a.line_off, nullptr, red);
}

// ---- cooperative persistent variants -----------------------------------------------------------------
// ONE launch walks all time steps: every workgroup keeps its 16 weight rows in LDS for the whole
// sequence, and the steps are separated by a grid-wide barrier instead of a kernel boundary.  All
// workgroups must be co-resident (launched with hipLaunchCooperativeKernel, grid <= CU count).
// Cross-workgroup data (h_t forward, the gate deltas backward) travels by device-scope (sc1) stores and
// loads, so the barrier itself needs no cache maintenance: drain the stores, one relaxed agent-scope
// ticket, relaxed polling (guide: "in-launch hand-off", sc1 variant).  The poll loop carries a watchdog:
// a workgroup that waits longer than ~seconds raises sync[1] and every workgroup leaves, so a scheduling
// accident surfaces as an error instead of a hung GPU.
struct CoopLds {
  int weights, red, loff, flag, words;
};
// nrows weight rows resident per workgroup, ncols = columns of its partial tile
inline __host__ __device__ CoopLds coop_lds_layout(int kp, int nrows, int ncols, int bs) {
  CoopLds l;
  int o = 0;
  l.weights = o; o += nrows * (kp + WIDE_WPAD);
  l.red = o;     o += WIDE_NW * 16 * (ncols + 4);
  l.loff = o;    o += ((bs + 1 + 3) / 4) * 4;
  l.flag = o;    o += 4;
  l.words = o;
  return l;
}
// weight rows [row0, row0 + nrows) of the packed array (rows >= rows_total read as zero) and the line
// offsets, once per launch
DEVFN void coop_stage(const LstmWideArgs& a, const float* wbase, long long row0, long long rows_total, int nrows,
                      float* wl, int* loff) {
  const int tid = threadIdx.x;
  const int k4 = a.kp >> 2;
  for (int i = tid; i < nrows * k4; i += WIDE_THREADS) {
    const int row = i / k4, c4 = i - row * k4;
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = 0.0f;
    if (row0 + row < rows_total) v = *reinterpret_cast<const f32x4*>(wbase + (size_t)(row0 + row) * a.kp + c4 * 4);
    *reinterpret_cast<f32x4*>(wl + row * (a.kp + WIDE_WPAD) + c4 * 4) = v;
  }
  for (int i = tid; i <= a.bs; i += WIDE_THREADS) loff[i] = a.line_off[i];
  __syncthreads();
}

// Cooperative forward tile: 16 lines x (16 cells x 4 gates).  Compared with the per-step kernel's
// 64 lines x 4 cells it quarters the h_{t-1} bytes a workgroup pulls per step (32 KB at no = 512) --
// those come at the cross-XCD per-workgroup rate -- and instead keeps 64 weight rows (132 KB) in LDS.
DEVFN void coop_fwd_tile(const LstmWideArgs& a, const int sg, const int ct, const int dir, const int zb,
                         const int* loff, const float* wl, float* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int no = a.no, nd = a.ndir;
  unsigned arow[1];
  {
    const int m = zb * 16 + (lane & 15);
    arow[0] = BUF_OOB_BASE;
    if (m < a.bs) {
      const int off = loff[m], T = loff[m + 1] - off;
      if (sg >= 1 && sg < T) {
        const int fprev = dir == 0 ? sg - 1 : T - sg;   // frame of own step s-1
        arow[0] = (unsigned)((long long)(off + fprev) * a.ldh + a.hofs + dir * no) * 4u;
      }
    }
  }
  const BufF32 abuf = make_buf(a.H, (size_t)a.N * a.ldh * 4);
  const int ml = tid >> 4, c16 = tid & 15;
  const int line = zb * 16 + ml, cell = ct * 16 + c16;
  bool live = ml < 16 && line < a.bs && cell < no;
  int off = 0, T = 0;
  if (live) {
    off = loff[line];
    T = loff[line + 1] - off;
    live = sg < T;
  }
  const long long n = off + (dir == 0 ? sg : T - 1 - sg);
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  const f32x4 gx = buf_load4(gbuf, live ? (unsigned)(((n * nd + dir) * no + cell) * 16) : BUF_OOB);
  const float c_prev = buf_load(cbuf, live && sg >= 1    // own write of the previous step
      ? (unsigned)((((long long)(off + (dir == 0 ? sg - 1 : T - sg)) * nd + dir) * no + cell) * 4) : BUF_OOB);

  wide_tile<1, 4, true>(abuf, arow, abuf, 0u, wl, a.kp, red);
  __syncthreads();

  if (live) {
    f32x4 k;
#pragma unroll
    for (int q = 0; q < 4; q++) k[q] = 0.0f;
#pragma unroll
    for (int w = 0; w < WIDE_NW; w++) {   // columns of cell c16: cell group c16>>2, slot (c16&3)*4 + gate
      const f32x4 p = *reinterpret_cast<const f32x4*>(&red[(w * 16 + ml) * 68 + c16 * 4]);
#pragma unroll
      for (int q = 0; q < 4; q++) k[q] += p[q];
    }
    const float gi = gate_act(k[0] + gx[0], false), gf = gate_act(k[1] + gx[1], false),
                go = gate_act(k[2] + gx[2], false), ci = gate_act(k[3] + gx[3], true);
    const float c = ci * gi + gf * c_prev;
    const float h = gate_act(c, true) * go;
    f32x4 act;
    act[0] = gi; act[1] = gf; act[2] = go; act[3] = ci;
    *reinterpret_cast<f32x4*>(a.G + ((n * nd + dir) * no + cell) * 4) = act;
    a.C[(n * nd + dir) * no + cell] = c;
    buf_store_dev(abuf, (unsigned)(n * a.ldh + a.hofs + dir * no + cell) * 4u, h);   // next step's A operand
    float* srow = a.S + (size_t)dir * a.sdir;
    if (sg == 0) srow[n * a.lds + a.sofs + cell] = 0.0f;                      // h_{-1} = 0
    if (sg + 1 < T) srow[(long long)(off + (dir == 0 ? sg + 1 : T - 2 - sg)) * a.lds + a.sofs + cell] = h;
  }
}

// grid (ceil(no/16), ndir, zsplit), 256 threads; workgroup z walks line blocks z, z + zsplit, ...
__global__ __launch_bounds__(WIDE_THREADS) void lstm_coop_fwd(LstmWideArgs a) {
  float* smem = dyn_smem<float>();
  const CoopLds L = coop_lds_layout(a.kp, 64, 64, a.bs);
  float* wl = smem + L.weights;
  float* red = smem + L.red;
  int* loff = reinterpret_cast<int*>(smem + L.loff);
  const int ct = blockIdx.x, dir = blockIdx.y;
  const int ncg = (a.no + 3) >> 2;
  // packed rows of cell groups 4ct .. 4ct+3 of this direction (16 rows each: cell_local*4 + gate)
  coop_stage(a, a.Rw + (size_t)dir * ncg * 16 * a.kp, (long long)ct * 64, (long long)ncg * 16, 64, wl, loff);
  const int nzb = (a.bs + 15) / 16;
  const int nwg = gridDim.x * gridDim.y * gridDim.z;
  for (int sg = 0; sg < a.tmax; sg++) {
    for (int zb = blockIdx.z; zb < nzb; zb += gridDim.z) {
      coop_fwd_tile(a, sg, ct, dir, zb, loff, wl, red);
      if (zb + (int)gridDim.z < nzb) __syncthreads();   // partial tiles are reused by the next line block
    }
    if (sg + 1 < a.tmax && !grid_barrier(a.sync, (sg + 1) * nwg, reinterpret_cast<int*>(smem + L.flag))) return;
  }
}

// grid (ceil(no/16), ndir, zsplit), 256 threads; workgroup z walks line blocks z, z + zsplit, ...
__global__ __launch_bounds__(WIDE_THREADS) void lstm_coop_bwd(LstmWideArgs a) {
  float* smem = dyn_smem<float>();
  const CoopLds L = coop_lds_layout(a.kp, 16, 16, a.bs);
  float* wl = smem + L.weights;
  float* red = smem + L.red;
  int* loff = reinterpret_cast<int*>(smem + L.loff);
  const int ct = blockIdx.x, dir = blockIdx.y;
  const int nct = (a.no + 15) >> 4;
  coop_stage(a, a.Rw + (size_t)dir * nct * 16 * a.kp, (long long)ct * 16, (long long)nct * 16, 16, wl, loff);
  const int nzb = (a.bs + 15) / 16;
  const int nwg = gridDim.x * gridDim.y * gridDim.z;
  for (int sg = 0; sg < a.tmax; sg++) {
    for (int zb = blockIdx.z; zb < nzb; zb += gridDim.z) {
      wide_bwd_tile<true>(a, sg, ct, dir, zb, loff, wl, red);
      if (zb + (int)gridDim.z < nzb) __syncthreads();
    }
    if (sg + 1 < a.tmax && !grid_barrier(a.sync, (sg + 1) * nwg, reinterpret_cast<int*>(smem + L.flag))) return;
  }
}

// ---- persistent forward recurrence, bf16 operands, one workgroup GROUP per XCD -------------------------------------------
// The per-step launches of lstm_wide_fwd_step16_bf16 cannot go below ~5 us per step: 3.2 us for an empty dependent launch
// plus an L2-bound round of operand loads, of which the 64 KB of weight rows per workgroup and step (1.3 us) never change.
// Here ONE launch walks all time steps.  The unit of synchronisation is the XCD, not the chip: the 32 cell tiles of one
// (line block, direction) form a group, a group lives on ONE XCD (a workgroup reads its XCD from the hardware id register
// and claims the next tile of that XCD's group), so
//   * a group's h ring rows are written and read through ONE L2: plain stores (they stay in that L2) + s_waitcnt vmcnt(0),
//     L1-bypassing (sc1) loads -- no write-through to memory, no cross-XCD hop;
//   * a step is separated from the next by a GROUP barrier (32 arrivals on the group's own counter), not a grid barrier;
//   * the 64 weight rows of a tile (64 KB bf16) are staged into LDS once and stay there;
//   * the epilogue operands of step s (gate pre-activations from HBM, c_{s-1}) are requested BEFORE the wait for step
//     s-1's h, so their latency is off the dependent chain.
// Grid = 8 x ceil(no/16) workgroups of 256 threads with > 80 KB of LDS each (one per CU, all co-resident: at most 256);
// groups = ndir x ceil(bs/16) <= 8.  Placement is CHECKED, not assumed: after claiming, every workgroup waits until all
// have claimed and verifies that every group got its ceil(no/16) tiles; otherwise it raises sync[1] and leaves BEFORE
// anything is written (the host then runs the per-step path).  Every poll loop carries a watchdog.
struct XcdSyncLayout { enum { ARRIVED = 0, ERROR = 1, SLOT0 = 8, GROUP0 = 32, GROUP_STRIDE = 32, WORDS = 32 + 8 * 32 }; };
constexpr int XCD_LDW = 512 + 8;        // halfs per resident weight row (conflict-free ds_read_b128 fragments), kp16 <= 512
inline __host__ __device__ int xcd_fwd_lds_bytes() { return 64 * XCD_LDW * 2 + WIDE_NW * 16 * 68 * 4 + 64; }

// thread 0 polls, everybody learns the outcome; `code` is what a time-out writes into the error word (1: before anything
// was written -- the host may fall back to the per-step path; 2: in the middle of the sequence -- fatal)
DEVFN bool xcd_poll(int* word, int target, int* err, int* lds_flag, int code) {
  if (threadIdx.x == 0) {
    int spins = 0, bad = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      poll_pause();
      if ((++spins & 63) == 0) {
        bad = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!bad && spins > GRID_WATCHDOG_SPINS) { bad = code; __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (bad) break;
      }
    }
    *lds_flag = bad;
  }
  __syncthreads();
  const bool ok = *lds_flag == 0;
  __syncthreads();
  return ok;
}

// The per-step group barrier as STAMPS instead of a counter (CLSTM_XCD_STAMPS, default): an atomic arrival leaves the L2
// (its line is dropped: the pollers' next look goes to memory, ~1 us); a plain store of "my tile has finished step s" into
// the group's own 128-byte line stays in the XCD's L2, and the poller's lanes read all tiles' stamps with one L1-bypassing
// load each.
#ifndef CLSTM_XCD_STAMPS
#define CLSTM_XCD_STAMPS 1
#endif
DEVFN void xcd_arrive(int* gwords, const int tile, const int step_done) {   // called by thread 0 behind drain + barrier
  if (CLSTM_XCD_STAMPS) __hip_atomic_store(gwords + tile, step_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_add(gwords, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every tile of the group has finished `steps_done` steps
DEVFN bool xcd_wait_group(int* gwords, const int ntile, const int steps_done, int* err, int* lds_flag) {
  if (!CLSTM_XCD_STAMPS) return xcd_poll(gwords, ntile * steps_done, err, lds_flag, 2);
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int spins = 0, bad = 0;
    for (;;) {
      const int v = lane < ntile ? __hip_atomic_load(gwords + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : steps_done;
      if (wave_ballot(v < steps_done) == 0ull) break;
      poll_pause();
      if ((++spins & 63) == 0) {
        bad = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!bad && spins > GRID_WATCHDOG_SPINS) { bad = 2; if (lane == 0) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        bad = wave_uniform(bad);
        if (bad) break;
      }
    }
    if (lane == 0) *lds_flag = bad;
  }
  __syncthreads();
  const bool ok = *lds_flag == 0;
  __syncthreads();
  return ok;
}

// LL = 1 (EXPERIMENT, CLSTM_XCD_LL=1; measured slower than the stamp barrier, see profiles/README.md) -- "flag in data",
// the low-latency protocol of collective libraries: every 8-byte unit of the h ring carries
// two bf16 cells AND a 32-bit tag (launch epoch << 12 | step); a consumer needs no barrier -- it loads the ring rows it is
// going to multiply and checks the tags of what arrived, re-loading until every unit carries the tag of the step it waits
// for.  An aligned 8-byte store is one L2 write, so a unit is never seen half-written.  Compared with the stamp barrier
// (store, wait for the L2's acknowledgement, workgroup barrier, stamp store, poll, THEN load) the dependent chain between
// two steps is one store and one load.  The ring doubles (8 bytes per cell pair); a slot is rewritten two steps later,
// which a producer can only reach after it has consumed everybody's data of the step in between (so every consumer of
// the old contents is done), and the epoch keeps a previous launch's units from matching.
#ifndef CLSTM_LL_POLICY   // (diagnostics) cache policy of the tagged ring: 1 system-scope loads, 2 + system-scope stores
#define CLSTM_LL_POLICY 0
#endif
template <bool LL>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_fwd_bf16(LstmWideArgs a) {
  unsigned short* wl = dyn_smem<unsigned short>();                         // [64][XCD_LDW]
  float* red = reinterpret_cast<float*>(wl + 64 * XCD_LDW);                // [4][16][68]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * 16 * 68);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nzb = a.zbn, ngroups = nd * nzb, ncg = (no + 3) >> 2;
  int* const sync = a.sync;
  // ---- claim a tile of this XCD's group, then check the placement of the whole grid ----
  const int xcd = hw_xcc_id() & 7;
  if (tid == 0) {
    flag[1] = __hip_atomic_fetch_add(sync + XcdSyncLayout::SLOT0 + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sync + XcdSyncLayout::ARRIVED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int ct = flag[1];
  if (!xcd_poll(sync + XcdSyncLayout::ARRIVED, (int)gridDim.x, sync + XcdSyncLayout::ERROR, flag, 1)) return;
  if (tid == 0) {
    int bad = 0;
    for (int g = 0; g < ngroups; g++)
      bad |= __hip_atomic_load(sync + XcdSyncLayout::SLOT0 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ntile;
    if (bad) __hip_atomic_store(sync + XcdSyncLayout::ERROR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[0] = bad;
  }
  __syncthreads();
  if (flag[0] != 0) return;                       // uneven placement: nothing has been written yet
  if (xcd >= ngroups || ct >= ntile) return;      // spare workgroup
  const int dir = xcd % nd, zb = a.zb0 + xcd / nd;
  int* const gcount = sync + XcdSyncLayout::GROUP0 + xcd * XcdSyncLayout::GROUP_STRIDE;

  // ---- the tile's 64 weight rows: cell groups 4ct .. 4ct+3 of this direction, 16 rows (cell_local*4 + gate) each ----
  {
    const int c8 = a.kp16 >> 3;   // 16-byte chunks per row
    for (int i = tid; i < 64 * c8; i += WIDE_THREADS) {
      const int row = i / c8, c = i - row * c8;
      const long long grow = (long long)(dir * ncg + ct * 4) * 16 + row;
      u16x8 v;
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = 0;
      if ((ct * 4) * 16 + row < ncg * 16) v = *reinterpret_cast<const u16x8*>(a.Rw16 + grow * a.kp16 + c * 8);
      *reinterpret_cast<u16x8*>(wl + row * XCD_LDW + c * 8) = v;
    }
  }
  // epilogue role: (line, cell); the line's extent is fixed for the whole launch
  const int ml = tid >> 4, c16 = tid & 15;
  const int line = zb * 16 + ml, cell = ct * 16 + c16;
  int off = 0, T = 0;
  if (line < a.bs) { off = a.line_off[line]; T = a.line_off[line + 1] - off; }
  const bool mine = line < a.bs && cell < no;
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Hb), (size_t)2 * nd * a.bs * a.kp16 * (LL ? 4 : 2));
  // A fragment of this lane: line zb*16 + (lane&15), 8 k at wave*kw + 32 g + 8 (lane>>4)
  const int kw = a.kp16 / WIDE_NW, ngrp = kw >> 5;   // <= 4 groups of 32 per wave
  const int am = zb * 16 + (lane & 15);
  const unsigned akl = (unsigned)(wave * kw + 8 * (lane >> 4)) * (LL ? 4u : 2u);   // (tagged ring: 4 bytes per cell)
  const unsigned short* wfrag = wl + (lane & 15) * XCD_LDW + wave * kw + 8 * (lane >> 4);
  const int kprod = ntile * 16;   // cells that have a producer (the rest of kp16 is padding nobody writes)
  __syncthreads();

  // The gate pre-activations of step s come from HBM (~2 us) and VMEM returns in order: requested at the top of step s
  // they would hold back the h rows requested behind them.  They are requested one step AHEAD, behind that step's h
  // loads; c_{s-1} is this thread's own result of the previous step and stays in a register.
  auto gx_load = [&](int sg) -> f32x4 {
    const bool lv = mine && sg < T;
    const long long nn = off + (dir == 0 ? sg : T - 1 - sg);
    return buf_load4(gbuf, lv ? (unsigned)(((nn * nd + dir) * no + cell) * 16) : BUF_OOB);
  };
  f32x4 gx = gx_load(0);
  float c_prev = 0.0f;
  // the per-frame outputs of one step (nobody inside the pass reads them)
  auto store_frame = [&](const f32x4 act, const float c_new, const float h, const unsigned hp, const long long n, const int sg, const bool live) {
    if (live && !(CLSTM_WEXP & 256)) {
      *reinterpret_cast<f32x4*>(a.G + ((n * nd + dir) * no + cell) * 4) = act;
    }
    if (live && !(CLSTM_WEXP & 256) && !((CLSTM_WEXP & 512) && (c16 & 3))) {
      a.C[(n * nd + dir) * no + cell] = c_new;
      a.H[n * a.ldh + a.hofs + dir * no + cell] = h;
      if (!(CLSTM_WEXP & 64)) {
      float* srow = a.S + (size_t)dir * a.sdir;
      if (sg == 0) srow[n * a.lds + a.sofs + cell] = 0.0f;
      if (sg + 1 < T) srow[(long long)(off + (dir == 0 ? sg + 1 : T - 2 - sg)) * a.lds + a.sofs + cell] = h;
      }
    }
    if (live && !(c16 & 1)) {
      if constexpr (!LL) *reinterpret_cast<unsigned*>(a.Hb + ((size_t)((sg & 1) * nd + dir) * a.bs + line) * a.kp16 + cell) = hp;
      if (a.Hbf && !(CLSTM_WEXP & 256)) *reinterpret_cast<unsigned*>(a.Hbf + (size_t)n * a.hbf_ld + dir * no + cell) = hp;
      if (a.Sbf && !(CLSTM_WEXP & 256)) {   // h_{t-1} column block of the next frame's bf16 source row (weight-gradient operand, gemm_b16mc)
        unsigned short* sb = a.Sbf + (size_t)dir * a.sbf_dir + a.sbf_ofs + cell;
        if (sg == 0) *reinterpret_cast<unsigned*>(sb + (size_t)n * a.sbf_ld) = 0u;
        if (sg + 1 < T) *reinterpret_cast<unsigned*>(sb + (size_t)(off + (dir == 0 ? sg + 1 : T - 2 - sg)) * a.sbf_ld) = hp;
      }
    }
  };
  f32x4 fr_act = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float fr_c = 0.0f, fr_h = 0.0f;
  unsigned fr_hp = 0u;
  long long fr_n = 0;
  int fr_sg = 0;
  bool fr_live = false;
  for (int sg = 0; sg < a.tmax; sg++) {
    const bool live = mine && sg < T;
    const long long n = off + (dir == 0 ? sg : T - 1 - sg);
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[j][q] = 0.0f;
    f32x4 gx_next;
    if constexpr (LL) {
      // ---- tagged ring: load, check, re-load until every unit this lane multiplies carries this step's tag ----
      const unsigned want = (a.epoch << 12) | (unsigned)sg;
      const unsigned arow = (sg >= 1 && am < a.bs) ? (unsigned)(((((sg - 1) & 1) * nd + dir) * a.bs + am) * a.kp16) * 4u + akl : BUF_OOB_BASE;
      const int kl = wave * kw + 8 * (lane >> 4);
      U32x4 ra[4][2];
      int spins = 0;
      bool first = true;
      for (;;) {
        COMPILER_MEMORY_BARRIER();   // the loads below must be re-issued every round (nothing in the loop writes that memory as far as the compiler can see)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int hh = 0; hh < 2; hh++) ra[g][hh] = buf_load4u_dev(abuf, g < ngrp ? arow + (unsigned)g * 128u + (unsigned)hh * 16u : BUF_OOB);
        SCHED_FENCE();
        if (first) { gx_next = gx_load(sg + 1); store_frame(fr_act, fr_c, fr_h, fr_hp, fr_n, fr_sg, fr_live); first = false; }
        SCHED_FENCE();
        unsigned miss = 0u;
        if (sg >= 1 && am < a.bs) {
#pragma unroll
          for (int g = 0; g < 4; g++)
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
#pragma unroll
              for (int e = 0; e < 2; e++)
                if (g < ngrp && kl + 32 * g + 4 * hh + 2 * e < kprod) miss |= ra[g][hh].v[2 * e + 1] ^ want;
        }
        if (wave_ballot(miss != 0u) == 0ull) break;
        // Not all there yet.  Re-loading the whole operand (8 KB per wave and round, every wave of every workgroup) floods
        // the L2 -- measured 3.0 us per step against 2.0 with the stamp barrier -- so wait on SAMPLES instead: one unit of
        // every producer store instruction this wave depends on (its 8 tiles x the 4 producer waves = 32 lanes, 16 bytes
        // each), and re-load the operand when they all carry the tag.
        {
          const int tl = ((wave * kw) >> 4) + (lane >> 2), sl = zb * 16 + 4 * (lane & 3);
          const bool sv = lane < 32 && (lane >> 2) < (kw >> 4) && sl < a.bs && tl * 16 < kprod;
          const unsigned soff = sv ? (unsigned)(((((sg - 1) & 1) * nd + dir) * a.bs + sl) * a.kp16) * 4u + (unsigned)tl * 64u : BUF_OOB;
          bool abandon = false;
          for (;;) {
            COMPILER_MEMORY_BARRIER();
            const U32x4 sm = buf_load4u_dev(abuf, soff);
            if (wave_ballot(sv && sm.v[1] != want) == 0ull) break;
            poll_pause();
            if ((++spins & 63) == 0) {
              int bad = __hip_atomic_load(sync + XcdSyncLayout::ERROR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (!bad && spins > GRID_WATCHDOG_SPINS) {
                bad = 2;
                if (lane == 0) {   // diagnostics for the host's error message (sync words 2..6)
                  sync[2] = sg; sync[3] = (int)sm.v[1]; sync[4] = (int)want; sync[5] = (wave << 16) | lane; sync[6] = (xcd << 8) | ct;
                  __hip_atomic_store(sync + XcdSyncLayout::ERROR, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
              }
              if (wave_uniform(bad)) { abandon = true; break; }
            }
          }
          if (abandon) break;   // abandoned launch: run to the end on whatever is there, the host reports the error
        }
      }
#pragma unroll
      for (int g = 0; g < 4; g++) {
        if (g < ngrp) {
          U32x4 d;
          d.v[0] = ra[g][0].v[0]; d.v[1] = ra[g][0].v[2]; d.v[2] = ra[g][1].v[0]; d.v[3] = ra[g][1].v[2];
          const u16x8 av = __builtin_bit_cast(u16x8, d);
#pragma unroll
          for (int j = 0; j < 4; j++)
            acc[j] = mfma16x16x32_bf16(av, *reinterpret_cast<const u16x8*>(wfrag + j * 16 * XCD_LDW + g * 32), acc[j]);
        }
      }
    } else {
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, sg, sync + XcdSyncLayout::ERROR, flag)) return;   // h_{s-1} of the whole group is in the L2
    // ---- 16 lines x 64 columns, split-K over the four waves ----
    const unsigned arow = (sg >= 1 && am < a.bs) ? (unsigned)(((((sg - 1) & 1) * nd + dir) * a.bs + am) * a.kp16) * 2u + akl : BUF_OOB_BASE;
    f32x4 ra[4];
#pragma unroll
    for (int g = 0; g < 4; g++) ra[g] = buf_load4_dev(abuf, g < ngrp ? arow + (unsigned)g * 64u : BUF_OOB);
    SCHED_FENCE();
    gx_next = gx_load(sg + 1);
    SCHED_FENCE();
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g < ngrp) {
        const u16x8 av = __builtin_bit_cast(u16x8, ra[g]);
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[j] = mfma16x16x32_bf16(av, *reinterpret_cast<const u16x8*>(wfrag + j * 16 * XCD_LDW + g * 32), acc[j]);
      }
    }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) red[(wave * 16 + (lane >> 4) * 4 + q) * 68 + j * 16 + (lane & 15)] = acc[j][q];
    __syncthreads();
    float h = 0.0f, c_new = 0.0f;
    f32x4 act;
    if (live) {
      f32x4 k;
#pragma unroll
      for (int q = 0; q < 4; q++) k[q] = 0.0f;
#pragma unroll
      for (int w = 0; w < WIDE_NW; w++) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&red[(w * 16 + ml) * 68 + c16 * 4]);
#pragma unroll
        for (int q = 0; q < 4; q++) k[q] += p[q];
      }
      const float gi = gate_act(k[0] + gx[0], false), gf = gate_act(k[1] + gx[1], false),
                  go = gate_act(k[2] + gx[2], false), ci = gate_act(k[3] + gx[3], true);
      c_new = ci * gi + gf * c_prev;
      h = gate_act(c_new, true) * go;
      act[0] = gi; act[1] = gf; act[2] = go; act[3] = ci;
    }
    const float hn = quad_xor1(h);
    const unsigned hp = bf16_pack2(h, cell + 1 < no ? hn : 0.0f);   // (h = 0 for a line that has ended)
    if constexpr (LL) {   // every line of the block publishes every step (a finished line: zeros), or its consumers would wait
      if (line < a.bs && !(c16 & 1)) {
        // (workgroup-scope store, sc0, like the stamps of the barrier variant)
        const unsigned long long unit = (unsigned long long)hp | ((unsigned long long)((a.epoch << 12) | (unsigned)(sg + 1)) << 32);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.Hb) + ((((size_t)((sg & 1) * nd + dir) * a.bs + line) * (a.kp16 >> 1)) + (cell >> 1)),
                           unit, __ATOMIC_RELAXED, CLSTM_LL_POLICY >= 2 ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      // the per-frame outputs wait in registers until the next step's operand loads have been issued (store_frame in the
      // wait loop above): the VMEM counter is in order, so a load issued behind them would also wait for THEIR acknowledgements
      fr_act = act; fr_c = c_new; fr_h = h; fr_hp = hp; fr_n = n; fr_sg = sg; fr_live = live;
    } else {
      store_frame(act, c_new, h, hp, n, sg, live);
    }
    // publish: every store of this workgroup acknowledged by the L2, then one arrival on the group's counter.  (Storing
    // the bf16 h first and the other arrays behind the arrival was measured SLOWER, 3.5 vs 3.2 us per step: VMEM
    // completes in order, so the next step's operand loads then wait behind those stores.)
    c_prev = c_new;
    gx = gx_next;
    if constexpr (LL) {
      __syncthreads();   // (the reduction buffer is rewritten by the next step)
    } else {
    drain_vmem();
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, ct, sg + 1);
    }
  }
  if constexpr (LL) store_frame(fr_act, fr_c, fr_h, fr_hp, fr_n, fr_sg, fr_live);   // the last step's outputs
}

// ---- persistent backward recurrence, same scheme: 16 lines x 16 cells per workgroup, its 16 weight rows (R^T, 2048 k)
// resident in LDS, the group's bf16 delta ring exchanged through the XCD's L2, the carried state delta in a register ----
constexpr int XCD_LDWB = 2048 + 8;      // halfs per resident weight row of the backward tile, kp16 <= 2048
// NT: 16-cell tiles per workgroup.  Every workgroup of a group reads the group's WHOLE delta ring row block (16 lines x
// 2048 k x 2 B = 64 KB) each step, so the step is bound by that L2's bandwidth (32 workgroups: 2 MB per step; the
// forward kernel moves 0.5 MB and runs 2.9 us per step against 4.3): with two tiles per workgroup a group has 16
// workgroups, half the traffic, and twice the MFMA work per wave (still < 0.3 us).
template <int NT>
inline __host__ __device__ int xcd_bwd_lds_bytes() {
  const int need = NT * 16 * XCD_LDWB * 2 + WIDE_NW * 16 * (NT * 16 + 4) * 4 + 64;
  return need > 84 * 1024 ? need : 84 * 1024;   // > 80 KB: one workgroup per CU, whatever the tile needs
}

template <int NT>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_bwd_bf16(LstmWideArgs a) {
  constexpr int LDR = NT * 16 + 4;
  unsigned short* wl = dyn_smem<unsigned short>();                         // [NT*16][XCD_LDWB]
  float* red = reinterpret_cast<float*>(wl + NT * 16 * XCD_LDWB);          // [4][16][LDR]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * 16 * LDR);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, ntile2 = (ntile + NT - 1) / NT, nzb = a.zbn, ngroups = nd * nzb;
  int* const sync = a.sync;
  const int xcd = hw_xcc_id() & 7;
  if (tid == 0) {
    flag[1] = __hip_atomic_fetch_add(sync + XcdSyncLayout::SLOT0 + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sync + XcdSyncLayout::ARRIVED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int slot = flag[1];
  if (!xcd_poll(sync + XcdSyncLayout::ARRIVED, (int)gridDim.x, sync + XcdSyncLayout::ERROR, flag, 1)) return;
  if (tid == 0) {
    int bad = 0;
    for (int g = 0; g < ngroups; g++)
      bad |= __hip_atomic_load(sync + XcdSyncLayout::SLOT0 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ntile2;
    if (bad) __hip_atomic_store(sync + XcdSyncLayout::ERROR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[0] = bad;
  }
  __syncthreads();
  if (flag[0] != 0) return;
  if (xcd >= ngroups || slot >= ntile2) return;
  const int dir = xcd % nd, zb = a.zb0 + xcd / nd;
  int* const gcount = sync + XcdSyncLayout::GROUP0 + xcd * XcdSyncLayout::GROUP_STRIDE;
  {
    const int c8 = a.kp16 >> 3;
    for (int i = tid; i < NT * 16 * c8; i += WIDE_THREADS) {
      const int row = i / c8, c = i - row * c8;
      const int ct = slot * NT + (row >> 4);
      u16x8 v;
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = 0;
      if (ct < ntile) v = *reinterpret_cast<const u16x8*>(a.Rw16 + ((long long)(dir * ntile + ct) * 16 + (row & 15)) * a.kp16 + c * 8);
      *reinterpret_cast<u16x8*>(wl + row * XCD_LDWB + c * 8) = v;
    }
  }
  const int ml = tid >> 4, c16 = tid & 15;
  const int line = zb * 16 + ml;
  int off = 0, T = 0;
  if (line < a.bs) { off = a.line_off[line]; T = a.line_off[line + 1] - off; }
  int cellj[NT];
  bool minej[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) { cellj[j] = (slot * NT + j) * 16 + c16; minej[j] = line < a.bs && cellj[j] < no; }
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  const BufF32 hbuf = make_buf(a.dH, (size_t)a.N * nd * no * 4);
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Db), (size_t)2 * nd * a.bs * a.kp16 * 2);
  const int kw = a.kp16 / WIDE_NW, ngrp = kw >> 5;   // <= 16 groups of 32 per wave
  const int am = zb * 16 + (lane & 15);
  const unsigned akl = (unsigned)(wave * kw + 8 * (lane >> 4)) * 2u;
  const unsigned short* wfrag = wl + (lane & 15) * XCD_LDWB + wave * kw + 8 * (lane >> 4);
  float dc_carry[NT];      // dc_{s+1} * gf_{s+1} of this thread's (line, cell)s: carried in registers, not through memory
#pragma unroll
  for (int j = 0; j < NT; j++) dc_carry[j] = 0.0f;
  __syncthreads();

  // Epilogue operands (forward-pass arrays, from HBM) are requested one step AHEAD and behind that step's delta loads, so
  // that they never sit in front of them in the in-order VMEM queue; c_s of a step is the c_{s-1} the previous step loaded.
  struct Ops { f32x4 act; float dh_in, c_m1; };
  auto ops_load = [&](int sg, int j) -> Ops {
    const bool lv = minej[j] && sg < T;
    const int ss = T - 1 - sg;
    const long long nn = off + (dir == 0 ? ss : sg);
    Ops o;
    o.act = buf_load4(gbuf, lv ? (unsigned)(((nn * nd + dir) * no + cellj[j]) * 16) : BUF_OOB);
    o.dh_in = buf_load(hbuf, lv ? (unsigned)((nn * (nd * no) + dir * no + cellj[j]) * 4) : BUF_OOB);
    o.c_m1 = buf_load(cbuf, lv && ss >= 1
        ? (unsigned)((((long long)(off + (dir == 0 ? ss - 1 : sg + 1)) * nd + dir) * no + cellj[j]) * 4) : BUF_OOB);
    return o;
  };
  Ops cur[NT];
  float c_s[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) {
    cur[j] = ops_load(0, j);
    const bool lv = minej[j] && 0 < T;
    c_s[j] = buf_load(cbuf, lv ? (unsigned)((((long long)(off + (dir == 0 ? T - 1 : 0)) * nd + dir) * no + cellj[j]) * 4) : BUF_OOB);
  }
  for (int sg = 0; sg < a.tmax; sg++) {
    const int s = T - 1 - sg;
    const long long n = off + (dir == 0 ? s : sg);
    bool live[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) live[j] = minej[j] && sg < T;
    if (sg >= 1 && !xcd_wait_group(gcount, ntile2, sg, sync + XcdSyncLayout::ERROR, flag)) return;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[j][q] = 0.0f;
    const unsigned arow = (sg >= 1 && am < a.bs) ? (unsigned)(((((sg - 1) & 1) * nd + dir) * a.bs + am) * a.kp16) * 2u + akl : BUF_OOB_BASE;
    Ops nxt[NT];
    {   // all sixteen 32-k groups of the wave's quarter requested at once: ONE L2 round trip per step (two rounds of
        // eight cost a second one: 4.3 vs 3.x us per step)
      f32x4 ra[16];
#pragma unroll
      for (int g = 0; g < 16; g++) ra[g] = buf_load4_dev(abuf, g < ngrp ? arow + (unsigned)g * 64u : BUF_OOB);
      SCHED_FENCE();
#pragma unroll
      for (int j = 0; j < NT; j++) nxt[j] = ops_load(sg + 1, j);
      SCHED_FENCE();
#pragma unroll
      for (int g = 0; g < 16; g++)
        if (g < ngrp) {
          const u16x8 av = __builtin_bit_cast(u16x8, ra[g]);
#pragma unroll
          for (int j = 0; j < NT; j++)
            acc[j] = mfma16x16x32_bf16(av, *reinterpret_cast<const u16x8*>(wfrag + j * 16 * XCD_LDWB + g * 32), acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) red[(wave * 16 + (lane >> 4) * 4 + q) * LDR + j * 16 + (lane & 15)] = acc[j][q];
    __syncthreads();
    f32x4 dl[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) {
      if (live[j]) {
        float dh_rec = 0.0f;
#pragma unroll
        for (int w = 0; w < WIDE_NW; w++) dh_rec += red[(w * 16 + ml) * LDR + j * 16 + c16];
        const float gi = cur[j].act[0], gf = cur[j].act[1], go = cur[j].act[2], ci = cur[j].act[3];
        const float dh = cur[j].dh_in + dh_rec;
        const float th = gate_act(c_s[j], true);
        const float d_go = th * dh;
        const float dc = (sg >= 1 ? dc_carry[j] : 0.0f) + (-th * th + 1.0f) * (go * dh);
        dc_carry[j] = dc * gf;
        const float d_gf = dc * cur[j].c_m1;
        const float d_gi = dc * ci, d_ci = dc * gi;
        dl[j][0] = (gi * (-gi + 1.0f)) * d_gi;
        dl[j][1] = (gf * (-gf + 1.0f)) * d_gf;
        dl[j][2] = (go * (-go + 1.0f)) * d_go;
        dl[j][3] = (-ci * ci + 1.0f) * d_ci;
        unsigned* db = reinterpret_cast<unsigned*>(a.Db + ((size_t)((sg & 1) * nd + dir) * a.bs + line) * a.kp16 + 4 * cellj[j]);
        db[0] = bf16_pack2(dl[j][0], dl[j][1]);   // what the group waits for goes first
        db[1] = bf16_pack2(dl[j][2], dl[j][3]);
      }
    }
    drain_vmem();
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, slot, sg + 1);
#pragma unroll
    for (int j = 0; j < NT; j++) {
      if (live[j]) {
        if (!a.skip_d) *reinterpret_cast<f32x4*>(a.D + ((n * nd + dir) * no + cellj[j]) * 4) = dl[j];
        if (a.Dbf) {   // k-contiguous bf16 copy per frame: the ready-made A operand of the x.d product (gemm_b16kk)
          unsigned* df = reinterpret_cast<unsigned*>(a.Dbf + (size_t)(n * nd + dir) * a.kp16 + 4 * cellj[j]);
          df[0] = bf16_pack2(dl[j][0], dl[j][1]);
          df[1] = bf16_pack2(dl[j][2], dl[j][3]);
        }
      }
      c_s[j] = cur[j].c_m1;
      cur[j] = nxt[j];
    }
  }
}

// ---- the same persistent per-XCD scheme with f32 operands (the parity-grade path of wide layers) -----------------------
// v_mfma_f32_16x16x4_f32, weight rows f32 in LDS (64 x (kp + 4) forward = 132 KB at 512 cells, 16 x (kp + 4) backward),
// h_{t-1} / delta_{t+1} rows read straight from the per-frame arrays H / D (plain stores by the group's workgroups,
// L1-bypassing loads -- one L2), fragments as in wide_tile: lane (i, kq) loads four consecutive k of row i, MFMA e of a
// 16-k group uses element e of both operands' float4.  Arithmetic identical to the per-step kernels (same tile, same
// split-K order), so the results are bit-identical to them.
inline __host__ __device__ int xcd_fwd_f32_lds_bytes(int kp) { return (64 * (kp + WIDE_WPAD) + WIDE_NW * 16 * 68 + 16) * 4; }
inline __host__ __device__ int xcd_bwd_f32_lds_bytes(int kp) {
  const int need = (16 * (kp + WIDE_WPAD) + WIDE_NW * 16 * WIDE_LDW + 16) * 4;
  return need > 84 * 1024 ? need : 84 * 1024;
}

// role assignment + placement check shared by the f32 kernels; returns false if this workgroup has nothing to do (or the
// launch is being abandoned)
DEVFN bool xcd_claim(int* sync, int* flag, const int ntile, const int ngroups, int& xcd, int& slot) {
  const int tid = threadIdx.x;
  xcd = hw_xcc_id() & 7;
  if (tid == 0) {
    flag[1] = __hip_atomic_fetch_add(sync + XcdSyncLayout::SLOT0 + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(sync + XcdSyncLayout::ARRIVED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  slot = flag[1];
  if (!xcd_poll(sync + XcdSyncLayout::ARRIVED, (int)gridDim.x, sync + XcdSyncLayout::ERROR, flag, 1)) return false;
  if (tid == 0) {
    int bad = 0;
    for (int g = 0; g < ngroups; g++)
      bad |= __hip_atomic_load(sync + XcdSyncLayout::SLOT0 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ntile;
    if (bad) __hip_atomic_store(sync + XcdSyncLayout::ERROR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[0] = bad;
  }
  __syncthreads();
  const bool ok = flag[0] == 0;
  __syncthreads();
  return ok && xcd < ngroups && slot < ntile;
}

__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_fwd_f32(LstmWideArgs a) {
  float* wl = dyn_smem<float>();                                   // [64][kp + 4]
  const int ldw = a.kp + WIDE_WPAD;
  float* red = wl + 64 * ldw;                                      // [4][16][68]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * 16 * 68);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nzb = a.zbn, ncg = (no + 3) >> 2;
  int xcd, ct;
  if (!xcd_claim(a.sync, flag, ntile, nd * nzb, xcd, ct)) return;
  const int dir = xcd % nd, zb = a.zb0 + xcd / nd;
  int* const gcount = a.sync + XcdSyncLayout::GROUP0 + xcd * XcdSyncLayout::GROUP_STRIDE;
  {
    const int k4 = a.kp >> 2;
    const float* wbase = a.Rw + (size_t)dir * ncg * 16 * a.kp;
    for (int i = tid; i < 64 * k4; i += WIDE_THREADS) {
      const int row = i / k4, c4 = i - row * k4;
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = 0.0f;
      if ((long long)ct * 64 + row < (long long)ncg * 16) v = *reinterpret_cast<const f32x4*>(wbase + ((size_t)ct * 64 + row) * a.kp + c4 * 4);
      *reinterpret_cast<f32x4*>(wl + row * ldw + c4 * 4) = v;
    }
  }
  const int ml = tid >> 4, c16 = tid & 15;
  const int line = zb * 16 + ml, cell = ct * 16 + c16;
  int off = 0, T = 0;
  if (line < a.bs) { off = a.line_off[line]; T = a.line_off[line + 1] - off; }
  const bool mine = line < a.bs && cell < no;
  // A-fragment role of this lane: line zb*16 + (lane&15)
  const int am = zb * 16 + (lane & 15);
  int aoff = 0, aT = 0;
  if (am < a.bs) { aoff = a.line_off[am]; aT = a.line_off[am + 1] - aoff; }
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 hbuf = make_buf(a.H, (size_t)a.N * a.ldh * 4);
  const int kw = a.kp / WIDE_NW, ngrp = kw >> 4;     // 16-k groups per wave (<= 8 at 512 cells)
  const unsigned klane = (unsigned)(wave * kw + 4 * (lane >> 4)) * 4u;
  const float* wrow = wl + (lane & 15) * ldw + wave * kw + 4 * (lane >> 4);
  auto gx_load = [&](int sg) -> f32x4 {
    const bool lv = mine && sg < T;
    const long long nn = off + (dir == 0 ? sg : T - 1 - sg);
    return buf_load4(gbuf, lv ? (unsigned)(((nn * nd + dir) * no + cell) * 16) : BUF_OOB);
  };
  f32x4 gx = gx_load(0);
  float c_prev = 0.0f;
  __syncthreads();

  for (int sg = 0; sg < a.tmax; sg++) {
    const bool live = mine && sg < T;
    const long long n = off + (dir == 0 ? sg : T - 1 - sg);
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, sg, a.sync + XcdSyncLayout::ERROR, flag)) return;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[j][q] = 0.0f;
    const unsigned arow = (sg >= 1 && am < a.bs && sg < aT)
        ? (unsigned)((long long)(aoff + (dir == 0 ? sg - 1 : aT - sg)) * a.ldh + a.hofs + dir * no) * 4u + klane : BUF_OOB_BASE;
    f32x4 ra[8];
#pragma unroll
    for (int g = 0; g < 8; g++) ra[g] = buf_load4_dev(hbuf, g < ngrp ? arow + (unsigned)g * 64u : BUF_OOB);
    SCHED_FENCE();
    const f32x4 gx_next = gx_load(sg + 1);
    SCHED_FENCE();
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g < ngrp) {
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bv[j] = *reinterpret_cast<const f32x4*>(wrow + j * 16 * ldw + g * 16);
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j] = mfma16x16x4(ra[g][e], bv[j][e], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) red[(wave * 16 + (lane >> 4) * 4 + q) * 68 + j * 16 + (lane & 15)] = acc[j][q];
    __syncthreads();
    float c_new = 0.0f;
    if (live) {
      f32x4 k;
#pragma unroll
      for (int q = 0; q < 4; q++) k[q] = 0.0f;
#pragma unroll
      for (int w = 0; w < WIDE_NW; w++) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&red[(w * 16 + ml) * 68 + c16 * 4]);
#pragma unroll
        for (int q = 0; q < 4; q++) k[q] += p[q];
      }
      const float gi = gate_act(k[0] + gx[0], false), gf = gate_act(k[1] + gx[1], false),
                  go = gate_act(k[2] + gx[2], false), ci = gate_act(k[3] + gx[3], true);
      c_new = ci * gi + gf * c_prev;
      const float h = gate_act(c_new, true) * go;
      f32x4 act;
      act[0] = gi; act[1] = gf; act[2] = go; act[3] = ci;
      a.H[n * a.ldh + a.hofs + dir * no + cell] = h;     // next step's A operand of the whole group
      *reinterpret_cast<f32x4*>(a.G + ((n * nd + dir) * no + cell) * 4) = act;
      a.C[(n * nd + dir) * no + cell] = c_new;
      float* srow = a.S + (size_t)dir * a.sdir;
      if (sg == 0) srow[n * a.lds + a.sofs + cell] = 0.0f;
      if (sg + 1 < T) srow[(long long)(off + (dir == 0 ? sg + 1 : T - 2 - sg)) * a.lds + a.sofs + cell] = h;
    }
    c_prev = c_new;
    gx = gx_next;
    drain_vmem();
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, ct, sg + 1);
  }
}

__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_bwd_f32(LstmWideArgs a) {
  float* wl = dyn_smem<float>();                                   // [16][kp + 4]
  const int ldw = a.kp + WIDE_WPAD;
  float* red = wl + 16 * ldw;                                      // [4][16][WIDE_LDW]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * 16 * WIDE_LDW);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nzb = a.zbn;
  int xcd, ct;
  if (!xcd_claim(a.sync, flag, ntile, nd * nzb, xcd, ct)) return;
  const int dir = xcd % nd, zb = a.zb0 + xcd / nd;
  int* const gcount = a.sync + XcdSyncLayout::GROUP0 + xcd * XcdSyncLayout::GROUP_STRIDE;
  {
    const int k4 = a.kp >> 2;
    const float* wbase = a.Rw + ((size_t)dir * ntile + ct) * 16 * a.kp;
    for (int i = tid; i < 16 * k4; i += WIDE_THREADS) {
      const int row = i / k4, c4 = i - row * k4;
      *reinterpret_cast<f32x4*>(wl + row * ldw + c4 * 4) = *reinterpret_cast<const f32x4*>(wbase + (size_t)row * a.kp + c4 * 4);
    }
  }
  const int ml = tid >> 4, c16 = tid & 15;
  const int line = zb * 16 + ml, cell = ct * 16 + c16;
  int off = 0, T = 0;
  if (line < a.bs) { off = a.line_off[line]; T = a.line_off[line + 1] - off; }
  const bool mine = line < a.bs && cell < no;
  const int am = zb * 16 + (lane & 15);
  int aoff = 0, aT = 0;
  if (am < a.bs) { aoff = a.line_off[am]; aT = a.line_off[am + 1] - aoff; }
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  const BufF32 hbuf = make_buf(a.dH, (size_t)a.N * nd * no * 4);
  const BufF32 dbuf = make_buf(a.D, (size_t)a.N * nd * 4 * no * 4);
  const int kw = a.kp / WIDE_NW, ngrp = kw >> 4;     // 16-k groups per wave (32 at 512 cells), walked in rounds of 8
  const unsigned klane = (unsigned)(wave * kw + 4 * (lane >> 4)) * 4u;
  const float* wrow = wl + (lane & 15) * ldw + wave * kw + 4 * (lane >> 4);
  struct Ops { f32x4 act; float dh_in, c_m1; };
  auto ops_load = [&](int sg) -> Ops {
    const bool lv = mine && sg < T;
    const int ss = T - 1 - sg;
    const long long nn = off + (dir == 0 ? ss : sg);
    Ops o;
    o.act = buf_load4(gbuf, lv ? (unsigned)(((nn * nd + dir) * no + cell) * 16) : BUF_OOB);
    o.dh_in = buf_load(hbuf, lv ? (unsigned)((nn * (nd * no) + dir * no + cell) * 4) : BUF_OOB);
    o.c_m1 = buf_load(cbuf, lv && ss >= 1
        ? (unsigned)((((long long)(off + (dir == 0 ? ss - 1 : sg + 1)) * nd + dir) * no + cell) * 4) : BUF_OOB);
    return o;
  };
  Ops cur = ops_load(0);
  float c_s = buf_load(cbuf, mine && 0 < T ? (unsigned)((((long long)(off + (dir == 0 ? T - 1 : 0)) * nd + dir) * no + cell) * 4) : BUF_OOB);
  float dc_carry = 0.0f;
  __syncthreads();

  for (int sg = 0; sg < a.tmax; sg++) {
    const bool live = mine && sg < T;
    const int s = T - 1 - sg;
    const long long n = off + (dir == 0 ? s : sg);
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, sg, a.sync + XcdSyncLayout::ERROR, flag)) return;
    f32x4 acc;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = 0.0f;
    const unsigned arow = (sg >= 1 && am < a.bs && sg < aT)
        ? (unsigned)(((long long)(aoff + (dir == 0 ? aT - sg : sg - 1)) * nd + dir) * 4 * no) * 4u + klane : BUF_OOB_BASE;
    // rounds of 8 groups, the next round's loads requested before this round's MFMAs
    f32x4 ra[2][8];
#pragma unroll
    for (int g = 0; g < 8; g++) ra[0][g] = buf_load4_dev(dbuf, g < ngrp ? arow + (unsigned)g * 64u : BUF_OOB);
    SCHED_FENCE();
    Ops nxt = ops_load(sg + 1);
    SCHED_FENCE();
    for (int r0 = 0; r0 < ngrp; r0 += 16) {
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int gb = r0 + half * 8;
#pragma unroll
        for (int g = 0; g < 8; g++) ra[half ^ 1][g] = buf_load4_dev(dbuf, gb + 8 + g < ngrp ? arow + (unsigned)(gb + 8 + g) * 64u : BUF_OOB);
        SCHED_FENCE();
#pragma unroll
        for (int g = 0; g < 8; g++) {
          const int gg = gb + g < ngrp ? gb + g : ngrp - 1;   // groups past the end multiply zero A rows
          const f32x4 bv = *reinterpret_cast<const f32x4*>(wrow + gg * 16);
#pragma unroll
          for (int e = 0; e < 4; e++) acc = mfma16x16x4(ra[half][g][e], bv[e], acc);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) red[(wave * 16 + (lane >> 4) * 4 + q) * WIDE_LDW + (lane & 15)] = acc[q];
    __syncthreads();
    f32x4 dl;
    if (live) {
      float dh_rec = 0.0f;
#pragma unroll
      for (int w = 0; w < WIDE_NW; w++) dh_rec += red[(w * 16 + ml) * WIDE_LDW + c16];
      const float gi = cur.act[0], gf = cur.act[1], go = cur.act[2], ci = cur.act[3];
      const float dh = cur.dh_in + dh_rec;
      const float th = gate_act(c_s, true);
      const float d_go = th * dh;
      const float dc = (sg >= 1 ? dc_carry : 0.0f) + (-th * th + 1.0f) * (go * dh);
      dc_carry = dc * gf;
      const float d_gf = dc * cur.c_m1;
      const float d_gi = dc * ci, d_ci = dc * gi;
      dl[0] = (gi * (-gi + 1.0f)) * d_gi;
      dl[1] = (gf * (-gf + 1.0f)) * d_gf;
      dl[2] = (go * (-go + 1.0f)) * d_go;
      dl[3] = (-ci * ci + 1.0f) * d_ci;
      *reinterpret_cast<f32x4*>(a.D + ((n * nd + dir) * no + cell) * 4) = dl;   // next step's A operand of the whole group
    }
    c_s = cur.c_m1;
    cur = nxt;
    drain_vmem();
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, ct, sg + 1);
  }
}

// contraction padding of the packed weights
inline int wide_kp_fwd(int no) { return ((no + 16 * WIDE_NW - 1) / (16 * WIDE_NW)) * 16 * WIDE_NW; }
inline int wide_kp_bwd(int no) { return ((4 * no + 16 * WIDE_NW - 1) / (16 * WIDE_NW)) * 16 * WIDE_NW; }
// bf16 rows: every wave's quarter of the contraction is whole k32 groups
inline int wide_kp16_fwd(int no) { return ((no + 32 * WIDE_NW - 1) / (32 * WIDE_NW)) * 32 * WIDE_NW; }
inline int wide_kp16_bwd(int no) { return ((4 * no + 32 * WIDE_NW - 1) / (32 * WIDE_NW)) * 32 * WIDE_NW; }

}  // namespace clstm
