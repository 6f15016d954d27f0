// lstm_wide.h -- NPLSTM recurrence for hidden sizes whose recurrent weights do not fit the
// register file of one workgroup (no > 128; BASELINE config "2 x BiLSTM(512)").
//
// Same arithmetic as lstm_seq.h (GenericNPLSTM::forward / ::backward, clstm.cc:600-653), different
// parallelisation: the minibatch's lines advance in lock-step -- ONE persistent launch per pass with a workgroup
// group per XCD (lstm_xcd_*, the default), or one launch per time step (the fallback when the placement
// check fails, and what CLSTM_XCD_REC=0 selects) -- and the recurrent product of a step is a
// (lines x no) . (no x 4no) MFMA GEMM spread over the whole chip instead of a per-line mat-vec:
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
  int tmax;             // persistent kernels: number of lock-steps (longest line)
  int zb0, zbn;         // persistent per-XCD kernels: this launch walks the 16-line blocks [zb0, zb0 + zbn), zbn * ndir <= 8
                        //   (lstm_xcd_bwd_bf16_c32: groups of 8 EPT lines, zbn * ndir <= 16)
  int* sync;            // persistent kernels: XcdSyncLayout words (zeroed per launch)
  // bf16 MFMA operands (per-step kernels lstm_wide_*_step_bf16; BASELINE config "2 x BiLSTM(512), bf16 MFMA"):
  const unsigned short* Rw16;   // the same weight rows as Rw, bf16, row length kp16
  unsigned short* Hb;           // [N][nd][kp16]  bf16 copy of h (forward A operand), pad columns stay zero
  unsigned short* Db;           // [N][nd][kp16]  bf16 copy of the gate deltas at column 4*cell+gate (backward A operand)
  unsigned short* Hbf;          // persistent forward kernel: per-frame [N][hbf_ld] bf16 copy of h (dir d at column d*no): the next
  int hbf_ld;                   //   layer's W_x product reads it as its k-contiguous A operand; or null
  int skip_d;                   // persistent backward kernel: the f32 deltas D are not stored (every consumer reads Dbf; the host expands Dbf if one does not)
  int skip_h, skip_s;           // persistent bf16 forward kernel: the f32 outputs H / the h_{t-1} columns of the f32 source rows S are not
                                //   stored -- every consumer of this pass reads Hbf / Sbf; the host rebuilds them exactly (h = tanh(c) go
                                //   from C and G) if something still asks (ops.h:k_h_from_state, k_source_h)
  unsigned short* Sbf;          // persistent forward kernel: bf16 source rows [x | h_{t-1} | 1] of THIS layer, [dir][N][sbf_ld] (h-part written here), or null
  int sbf_ld, sbf_ofs; long long sbf_dir;
  unsigned short* Dbf;          // persistent backward kernel: per-frame [N][nd][kp16] bf16 deltas (operand of the x.d GEMM), or null
  float* dbias;                 // persistent bf16 backward kernels: [bs][nd][no][4] -- per line, the sum over its frames of the gate deltas AS STORED
                                //   IN Dbf (bf16-rounded, summed in f32): the bias row of the weight gradient, W.d[:,0] += sum_b y.d
                                //   (clstm_compute.cc:301), which then needs no 1537th row in the weight-gradient product (gemm_b16mc); or null
  int kp16;                     // padded contraction length of the bf16 rows, multiple of 32 * WIDE_NW
  float* Rf;                    // persistent f32 kernels: tiled lock-step ring of h (forward) / the gate deltas (backward), ring32_* below
  // persistent bf16 forward kernel with the input projection folded in (lstm_xcd_fwd_bf16_fx): the layer's input frames as
  // bf16 rows [N][x_ld] (k contiguous), W_x as bf16 rows [ndir * 4 no][ni] in gate-column order (Layer::WtbT) and the bias
  const unsigned short* Xb; int x_ld, x_ni;
  const unsigned short* Wxb;
  const float* bias;
  int debug_fail_claim;         // tests: the placement check of the persistent kernels reports failure (clstm_debug_set_device_error 4)
  int stamp_base;               // persistent kernels: the group barrier's stamps of this launch are stamp_base + steps finished (see xcd_finish)
  int* out_sticky;              // persistent kernels: the process's sticky device error word (takes a non-zero outcome), or null
  int* out_host;                // ... and a pinned host word that takes the outcome (0 = fine) when the launch has ended, or null
  long long rw_plane, ring_plane; // f32-grade backward recurrence on the bf16 MFMA (lstm_xcd_bwd_x3): halfs between the hi and lo planes of Rw16 and of the ring Db
  long long* prof;              // diagnostics build (CLSTM_LSTM_PROF) only: per-phase cycle sums, [2 workgroups][4 waves][12]; else null
};

constexpr int WIDE_LDW = 20;  // LDS row stride of a partial tile (16 columns + pad, float4-aligned)
constexpr int WIDE_PF = 4;    // 16-k groups in flight per wave, per-step kernels (8 measured slower: 200 VGPRs)
constexpr int WIDE_NW = 4;    // waves per workgroup = split-K factor (8 was measured on MI355X at 512 cells: 9.9 / 8.5 ms
                              // against 9.7 / 8.3 ms per pass -- the step is not bound by the MFMA/load rounds of a wave)
constexpr int WIDE_THREADS = 64 * WIDE_NW;
constexpr int WIDE_WPAD = 4;  // LDS weight rows are kp + 4 floats: 16 rows x b128 reads cover all banks once

// acc[i] += A_i(16 rows x kslice) . B(kslice x 16 cols) for this wave's quarter of the contraction,
// then the four waves' partial tiles are left in red[wave][row][col] (caller syncs).
// B rows come from global memory (bbuf/brow), A rows by plain loads (per-step launches).
template <int MT, int NT>
DEVFN void wide_tile(const BufF32 abuf, const unsigned (&arow)[MT], const BufF32 bbuf, const unsigned brow,
                     const int kp, float* red) {
  constexpr int PF = WIDE_PF;
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
      a[i] = buf_load4(abuf, live ? arow[i] + ko : BUF_OOB);
#pragma unroll
    for (int j = 0; j < NT; j++) b[j] = buf_load4(bbuf, live ? brow + (unsigned)j * 16u * (unsigned)kp * 4u + ko : BUF_OOB);
  };
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
#pragma unroll
      for (int j = 0; j < NT; j++) bv[j] = rb[p][j];
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
  const unsigned arow = (sg >= 1 && m < a.bs) ? (unsigned)(((((sg - 1) & 1) * nd + dir) * a.bs + m) * a.kp16) * 2u : BUF_OOB_BASE;
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Hb), (size_t)2 * nd * a.bs * a.kp16 * 2);
  const BufF32 bbuf = make_buf(reinterpret_cast<const float*>(a.Rw16), (size_t)a.rw_elems * 2);
  const unsigned brow = (unsigned)(((long long)(dir * ncg + ct * 4) * 16 + (lane & 15)) * a.kp16) * 2u;   // rows past the
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
    live = line < a.bs && cell < no && sg < T;
    n = off + (dir == 0 ? sg : T - 1 - sg);
    gx = buf_load4(gbuf, live ? (unsigned)(((n * nd + dir) * no + cell) * 16) : BUF_OOB);
    c_prev = buf_load(cbuf, live && sg >= 1
        ? (unsigned)((((long long)(off + (dir == 0 ? sg - 1 : T - sg)) * nd + dir) * no + cell) * 4) : BUF_OOB);
  });
  __syncthreads();

  float h = 0.0f;
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
    // (bf16 mode: the five-operation affine forms, devintrin.h:act_affine -- every bf16-mode kernel of a wide layer uses them, so the
    // persistent pass, the per-step launches and the on-demand rebuild of h agree bit for bit)
    const float gi = act_affine(k[0] + gx[0], ACT_SIG_SCALE, 1.0f, 0.0f), gf = act_affine(k[1] + gx[1], ACT_SIG_SCALE, 1.0f, 0.0f),
                go = act_affine(k[2] + gx[2], ACT_SIG_SCALE, 1.0f, 0.0f), ci = tanh_fast(k[3] + gx[3]);
    const float c = ci * gi + gf * c_prev;      // c_prev reads 0 at the first step
    h = tanh_fast(c) * go;
    f32x4 act;
    act[0] = gi; act[1] = gf; act[2] = go; act[3] = ci;
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
template <int MT>
DEVFN void wide_fwd_tile(const LstmWideArgs& a, const int sg, const int cg, const int dir, const int zb,
                         const int* loff, float* red) {
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
        arow[i] = (unsigned)((long long)(off + fprev) * a.ldh + a.hofs + dir * no) * 4u;
      }
    }
  }
  const BufF32 abuf = make_buf(a.H, (size_t)a.N * a.ldh * 4);
  const BufF32 bbuf = make_buf(a.Rw, (size_t)a.rw_elems * 4);
  const unsigned brow = (unsigned)(((long long)(dir * ncg + cg) * 16 + (lane & 15)) * a.kp) * 4u;

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
  const unsigned goff = live ? (unsigned)(((n * nd + dir) * no + cell) * 16) : BUF_OOB;
  const f32x4 gx = buf_load4(gbuf, goff);
  const float c_prev = buf_load(cbuf, live && sg >= 1
      ? (unsigned)((((long long)(off + (dir == 0 ? sg - 1 : T - sg)) * nd + dir) * no + cell) * 4) : BUF_OOB);

  wide_tile<MT, 1>(abuf, arow, bbuf, brow, a.kp, red);
  __syncthreads();

  // fused forward_full1 x4 + forward_statemem + forward_nonlingate for (line, cell)
  if (live) {
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
    *reinterpret_cast<f32x4*>(a.G + ((n * nd + dir) * no + cell) * 4) = act;
    a.C[(n * nd + dir) * no + cell] = c;
    a.H[n * a.ldh + a.hofs + dir * no + cell] = h;   // next step's A operand of every workgroup of this direction
    float* srow = a.S + (size_t)dir * a.sdir;
    if (sg == 0) srow[n * a.lds + a.sofs + cell] = 0.0f;                      // h_{-1} = 0
    if (sg + 1 < T) srow[(long long)(off + (dir == 0 ? sg + 1 : T - 2 - sg)) * a.lds + a.sofs + cell] = h;
  }
}

// per-step launch: grid (ceil(no/4), ndir, ceil(bs / 16MT)), 256 threads
template <int MT>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_wide_fwd_step(LstmWideArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WIDE_NW * MT * 16 * WIDE_LDW];
  wide_fwd_tile<MT>(a, a.step, blockIdx.x, blockIdx.y, blockIdx.z, a.line_off, red);
}

// ---- backward: one time step of 16 lines for one (16-cell tile, direction) ------------------------
DEVFN void wide_bwd_tile(const LstmWideArgs& a, const int sg, const int ct, const int dir, const int zb,
                         const int* loff, float* red) {
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
        arow[0] = (unsigned)(((long long)(off + fnext) * nd + dir) * 4 * no) * 4u;
      }
    }
  }
  const BufF32 abuf = make_buf(a.D, (size_t)a.N * nd * 4 * no * 4);
  const BufF32 bbuf = make_buf(a.Rw, (size_t)a.rw_elems * 4);
  const unsigned brow = (unsigned)(((long long)(dir * nct + ct) * 16 + (lane & 15)) * a.kp) * 4u;

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

  wide_tile<1, 1>(abuf, arow, bbuf, brow, a.kp, red);
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
    // the deltas are next step's A operand of every workgroup of this direction
    *reinterpret_cast<f32x4*>(a.D + ((n * nd + dir) * no + cell) * 4) = dl;
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
    const float th = tanh_fast(c_s);           // backward_nonlingate recomputes tanh(state)
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

// per-step launch: grid (ceil(no/16), ndir, ceil(bs/16)), 256 threads
__global__ __launch_bounds__(WIDE_THREADS) void lstm_wide_bwd_step(LstmWideArgs a) {
  __shared__ __attribute__((aligned(16))) float red[WIDE_NW * 16 * WIDE_LDW];
  wide_bwd_tile(a, a.step, blockIdx.x, blockIdx.y, blockIdx.z, a.line_off, red);
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
// The lock-step rings Hb / Db of the PERSISTENT bf16 kernels are tiled: [step parity][dir][16-line block][32-k block][line][32 k].
// The sixteen lines' 64-byte pieces of one 32-k block lie side by side (1 KB), so one A-fragment load instruction (lane = line +
// 16 x 16-byte chunk) asks the L2 for eight WHOLE 128-byte lines; with row-major rows it asked for sixteen half lines, and the
// backward step -- every workgroup of a group reads the group's whole 64 KB delta block -- spent 3,000 of its 6,100 cycles
// issuing those loads (scripts/gpu_xcdprof.py).  Offsets in bf16 elements; nkb = kp16 / 32, nblk = ceil(bs / 16).
DEVFN unsigned ring_block(const int parity, const int nd, const int dir, const int nblk, const int blk, const int nkb) {
  return (unsigned)(((parity * nd + dir) * nblk + blk) * nkb) * 512u;
}
DEVFN unsigned ring_elem(const int kb, const int line16, const int k32) { return (unsigned)(kb * 512 + line16 * 32 + k32); }

// The f32 kernels' ring, same idea with 16-k blocks of floats: [step parity][dir][16-line block][16-k block][line][16 k]; a lane
// group's load (lane = line + 16 x four consecutive k) again covers 1 KB of whole lines.  Offsets in floats; nkb = kp / 16.
DEVFN unsigned ring32_block(const int parity, const int nd, const int dir, const int nblk, const int blk, const int nkb) {
  return (unsigned)(((parity * nd + dir) * nblk + blk) * nkb) * 256u;
}
DEVFN unsigned ring32_elem(const int kb, const int line16, const int k16) { return (unsigned)(kb * 256 + line16 * 16 + k16); }
inline __host__ __device__ size_t ring32_floats(int nd, int bs, int kp) { return (size_t)2 * nd * ((bs + 15) / 16) * 16 * kp; }

// diagnostics build only: per-phase cycle stamps of the persistent bf16 kernels (scripts/gpu_xcdprof.py); a stamp costs
// ~60 cycles and drains lgkmcnt
#ifdef CLSTM_LSTM_PROF
#define XCD_PROF_DECL long long xacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long xpt; \
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(xpt) :: "memory")
#define XCD_STAMP(k) do { long long now_; SCHED_FENCE(); \
                          asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
                          SCHED_FENCE(); xacc[k] += now_ - xpt; xpt = now_; } while (0)
#define XCD_PROF_WRITE(xcd, slot, ntile) do { if (a.prof && (xcd) == 0 && ((slot) == 0 || (slot) == (ntile) - 1) && (threadIdx.x & 63) == 0) \
    for (int k_ = 0; k_ < 12; k_++) a.prof[(((slot) == 0 ? 0 : 1) * 4 + (threadIdx.x >> 6)) * 12 + k_] = xacc[k_]; } while (0)
#else
#define XCD_PROF_DECL do {} while (0)
#define XCD_STAMP(k) do {} while (0)
#define XCD_PROF_WRITE(xcd, slot, ntile) do {} while (0)
#endif
// EXITED: workgroups that have left the kernel; the last one publishes the launch's outcome (LAST_ERROR, the process's sticky
// error word, a pinned host word) and returns the COUNTER words (ARRIVED, ERROR, SLOT0.., EXITED) to zero for the next launch:
// no memset, no outcome kernel and no event record around a pass (they cost ~15 us of stream time per pass, four passes per
// configs[4] step).  Those words are only ever touched by agent-scope atomics, which are coherent across the XCDs.  The
// GROUP lines are NOT: a group's tiles write their stamps with plain stores that live (dirty) in their XCD's L2, and a zero
// written by a workgroup of ANOTHER XCD would race with that line's write-back at the end of the kernel -- the next launch
// could then find last launch's stamps and walk through its group barriers (seen once as a 2e-5 error in one of 30 runs of a
// ragged f32 case).  So the stamps are never reset: every launch counts from its own base (LstmWideArgs::stamp_base, the host
// adds tmax + 2 per launch), and a line has one writing XCD for ever.

struct XcdSyncLayout { enum { ARRIVED = 0, ERROR = 1, SLOT0 = 8, EXITED = 16, LAST_ERROR = 17, GROUP0 = 32, GROUP_STRIDE = 32, WORDS = 32 + 16 * 32 }; };   // (up to 16 groups: lstm_xcd_bwd_bf16_c32)
constexpr int XCD_LDW = 512 + 8;        // halfs per resident weight row (conflict-free ds_read_b128 fragments), kp16 <= 512
inline __host__ __device__ int xcd_fwd_lds_bytes(int mt = 1) { return 64 * XCD_LDW * 2 + WIDE_NW * mt * 16 * 68 * 4 + 64; }

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

// The per-step group barrier is made of STAMPS, not a counter: an atomic arrival leaves the L2 (its line is dropped: the
// pollers' next look goes to memory, ~1 us; measured 8.35 vs 7.6 ms per configs[4] minibatch); a plain store of "my tile
// has finished step s" into the group's own 128-byte line stays in the XCD's L2, and the poller's lanes read all tiles'
// stamps with one L1-bypassing load each.
DEVFN void xcd_arrive(int* gwords, const int tile, const int step_done) {   // called by thread 0 behind drain + barrier
  __hip_atomic_store(gwords + tile, step_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// every tile of the group has finished `steps_done` steps
DEVFN bool xcd_wait_group(int* gwords, const int ntile, const int steps_done, int* err, int* lds_flag) {
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

// role assignment + placement check shared by the persistent kernels; returns false if this workgroup has nothing to do (or the
// launch is being abandoned)
DEVFN bool xcd_claim(int* sync, int* flag, const int ntile, const int ngroups, int& xcd, int& slot, const int debug_fail = 0) {
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
    int bad = debug_fail;
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

// last act of every workgroup of a persistent launch (also of those that found nothing to do or bailed out): see EXITED
DEVFN void xcd_finish(const LstmWideArgs& a) {
  __syncthreads();
  if (threadIdx.x != 0) return;
  int* const sync = a.sync;
  if (__hip_atomic_fetch_add(sync + XcdSyncLayout::EXITED, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (int)gridDim.x - 1) return;
  const int e = __hip_atomic_load(sync + XcdSyncLayout::ERROR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int w = 0; w < XcdSyncLayout::GROUP0; w++)   // (the counter words; the groups' stamp lines keep counting)
    __hip_atomic_store(sync + w, w == XcdSyncLayout::LAST_ERROR ? e : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (e && a.out_sticky) __hip_atomic_store(a.out_sticky, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.out_host) store_i32_wt(a.out_host, e);
}

// ---- per-frame outputs of the persistent bf16 forward kernels --------------------------------------------------------------
// One (line, cell) per thread and line tile; a step stores the activations (G), c (C), the bf16 h into the lock-step ring, into
// Hbf and into the next frame's bf16 source row (Sbf) -- five stores whose addresses used to be recomputed from the frame
// number each step (64-bit multiply-adds, ~60 VALU operations and ten branches on a wave that has its SIMD to itself: a third
// of the 1,330-cycle epilogue, scripts/gpu_xcdprof.py).  Here every array is a buffer resource, a thread keeps its byte offset
// in each and advances it by one frame per step, and a lane without a live element stores out of range (dropped by the
// bounds check) instead of branching.  32-bit offsets: the host keeps every array of this path below 2 GiB.
struct FwdOutBufs { BufF32 g, c, ring, hbf, sbf, h, s; unsigned s4, shb, ssb, sh, ssf, ring_parity; bool ring_here; };
struct FwdOutOfs { unsigned o4, ohb, osb, rofs, oh, osf; };
// ring_here: the kernel's lock-step ring takes the one packed bf16 word `hp` (the bf16 kernels); false: the caller stores its ring itself
DEVFN FwdOutBufs fwd_out_bufs(const LstmWideArgs& a, const int dir, const BufF32 ring, const int nblk, const int nkb, const bool ring_here = true) {
  const int nd = a.ndir, fstep = dir == 0 ? 1 : -1;   // the forward walk: frames 0 .. T-1 of direction 0, T-1 .. 0 of the reversed one
  const size_t gbytes = (size_t)a.N * nd * a.no * 16;
  FwdOutBufs b;
  b.g = make_buf(a.G, gbytes);
  b.c = make_buf(a.C, gbytes / 4);
  b.ring = ring;
  b.ring_here = ring_here;
  b.hbf = make_buf(reinterpret_cast<const float*>(a.Hbf), a.Hbf ? (size_t)a.N * a.hbf_ld * 2 : 0);
  b.sbf = make_buf(reinterpret_cast<const float*>(a.Sbf), a.Sbf ? ((size_t)(nd - 1) * a.sbf_dir + (size_t)a.N * a.sbf_ld) * 2 : 0);
  b.h = make_buf(a.skip_h ? nullptr : a.H, a.skip_h ? 0 : (size_t)a.N * a.ldh * 4);
  b.s = make_buf(a.skip_s ? nullptr : a.S, a.skip_s ? 0 : ((size_t)(nd - 1) * a.sdir + (size_t)a.N * a.lds) * 4);
  b.s4 = (unsigned)(fstep * nd * a.no * 4);
  b.shb = (unsigned)(fstep * a.hbf_ld * 2);
  b.ssb = (unsigned)(fstep * a.sbf_ld * 2);
  b.sh = (unsigned)(fstep * a.ldh * 4);
  b.ssf = (unsigned)(fstep * a.lds * 4);
  b.ring_parity = ring_block(1, nd, 0, nblk, 0, nkb) * 2u;
  return b;
}
DEVFN FwdOutOfs fwd_out_ofs(const LstmWideArgs& a, const int dir, const int cell, const int line, const int off, const int T, const int nblk, const int nkb) {
  const int nd = a.ndir;
  const long long n0 = off + (dir == 0 ? 0 : T - 1);
  FwdOutOfs o;
  o.o4 = (unsigned)(((n0 * nd + dir) * a.no + cell) * 4);
  o.ohb = (unsigned)((n0 * a.hbf_ld + dir * a.no + cell) * 2);
  o.osb = (unsigned)(((long long)dir * a.sbf_dir + n0 * a.sbf_ld + a.sbf_ofs + cell) * 2);
  o.rofs = (ring_block(0, nd, dir, nblk, line >> 4, nkb) + ring_elem(cell >> 5, line & 15, cell & 31)) * 2u;
  o.oh = (unsigned)((n0 * a.ldh + a.hofs + dir * a.no + cell) * 4);
  o.osf = (unsigned)(((long long)dir * a.sdir + n0 * a.lds + a.sofs + cell) * 4);
  return o;
}
// live: this thread's (line, cell) has a frame at step sg; pair: ... and it is the even cell of a pair (hp = its h and the next cell's)
DEVFN void fwd_out_store(const LstmWideArgs& a, const FwdOutBufs& b, FwdOutOfs& o, const f32x4 act, const float c_new, const float h, const unsigned hp,
                         const int sg, const bool live, const bool pair, const int T) {
  buf_store4(b.g, live ? o.o4 << 2 : BUF_OOB, act);
  buf_store(b.c, live ? o.o4 : BUF_OOB, c_new);
  if (b.ring_here) buf_store_u32_s(b.ring, pair ? o.rofs : BUF_OOB_BASE, (sg & 1) ? b.ring_parity : 0u, hp);
  if (a.Hbf) buf_store_u32(b.hbf, pair ? o.ohb : BUF_OOB, hp);
  if (a.Sbf) {   // h_{t-1} column block of the NEXT frame's bf16 source row (weight-gradient operand, gemm_b16mc); the first frame's is zero
    if (sg == 0) buf_store_u32(b.sbf, pair ? o.osb : BUF_OOB, 0u);
    buf_store_u32(b.sbf, pair && sg + 1 < T ? o.osb + b.ssb : BUF_OOB, hp);
  }
  if (!a.skip_h) buf_store(b.h, live ? o.oh : BUF_OOB, h);
  if (!a.skip_s) {   // the same column block of the f32 source rows
    if (sg == 0) buf_store(b.s, live ? o.osf : BUF_OOB, 0.0f);
    buf_store(b.s, live && sg + 1 < T ? o.osf + b.ssf : BUF_OOB, h);
  }
  o.o4 += b.s4; o.ohb += b.shb; o.osb += b.ssb; o.oh += b.sh; o.osf += b.ssf;
}

// MT: 16-line tiles per group (1: a group = 16 lines; 2: 32 lines -- minibatches of more than 8 / ndir blocks of 16 lines walk
// half as many sequential launches, every weight fragment read from LDS serves two MFMAs, and a step's barrier and ring round
// trip -- what a step costs -- are paid once for twice the lines)
template <int MT>
DEVFN void lstm_xcd_fwd_bf16_body(const LstmWideArgs& a) {
  unsigned short* wl = dyn_smem<unsigned short>();                         // [64][XCD_LDW]
  float* red = reinterpret_cast<float*>(wl + 64 * XCD_LDW);                // [4][MT * 16][68]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * MT * 16 * 68);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nzb = a.zbn, ngroups = nd * nzb, ncg = (no + 3) >> 2;
  int* const sync = a.sync;
  // ---- claim a tile of this XCD's group, then check the placement of the whole grid ----
  int xcd, ct;
  if (!xcd_claim(sync, flag, ntile, ngroups, xcd, ct, a.debug_fail_claim)) return;   // (uneven placement: nothing has been written yet)
  const int dir = xcd % nd, zb = a.zb0 + xcd / nd;               // zb: block of 16 MT lines
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
  // epilogue role: (line, cell) for each of the MT line tiles; a line's extent is fixed for the whole launch
  const int ml = tid >> 4, c16 = tid & 15;
  const int cell = ct * 16 + c16;
  int line[MT], off[MT], T[MT];
  bool mine[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) {
    line[i] = (zb * MT + i) * 16 + ml;
    off[i] = 0; T[i] = 0;
    if (line[i] < a.bs) { off[i] = a.line_off[line[i]]; T[i] = a.line_off[line[i] + 1] - off[i]; }
    mine[i] = line[i] < a.bs && cell < no;
  }
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const int nblk = (a.bs + 15) >> 4, nkb = a.kp16 >> 5;
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Hb), (size_t)2 * nd * nblk * 16 * a.kp16 * 2);
  // A fragment of this lane: line (zb MT + i) 16 + (lane&15), 8 k at wave*kw + 32 g + 8 (lane>>4)
  const int kw = a.kp16 / WIDE_NW, ngrp = kw >> 5;   // <= 4 groups of 32 per wave
  const unsigned akl = ring_elem(wave * ngrp, lane & 15, 8 * (lane >> 4)) * 2u;
  const unsigned short* wfrag = wl + (lane & 15) * XCD_LDW + wave * kw + 8 * (lane >> 4);
  __syncthreads();
  // this wave's B fragments -- 64 columns x its quarter of the contraction = 16 x 16 bytes per lane -- stay in REGISTERS for the
  // whole sequence (a wave has a SIMD's register file to itself): no LDS read on a step's path
  u16x8 wreg[4][4];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int j = 0; j < 4; j++) wreg[g][j] = *reinterpret_cast<const u16x8*>(wfrag + j * 16 * XCD_LDW + (g < ngrp ? g : 0) * 32);

  // The gate pre-activations of step s come from HBM (~2 us) and VMEM returns in order: requested at the top of step s
  // they would hold back the h rows requested behind them.  They are requested one step AHEAD, behind that step's h
  // loads; c_{s-1} is this thread's own result of the previous step and stays in a register.
  auto gx_load = [&](int sg, int i) -> f32x4 {
    const bool lv = mine[i] && sg < T[i];
    const long long nn = off[i] + (dir == 0 ? sg : T[i] - 1 - sg);
    return buf_load4(gbuf, lv ? (unsigned)(((nn * nd + dir) * no + cell) * 16) : BUF_OOB);
  };
  f32x4 gx[MT];
  float c_prev[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) { gx[i] = gx_load(0, i); c_prev[i] = 0.0f; }
  // the per-frame outputs of one step (nobody inside the pass reads them): fwd_out_store
  const FwdOutBufs ob = fwd_out_bufs(a, dir, abuf, nblk, nkb);
  FwdOutOfs oo[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) oo[i] = fwd_out_ofs(a, dir, cell, line[i], off[i], T[i], nblk, nkb);
  XCD_PROF_DECL;
  for (int sg = 0; sg < a.tmax; sg++) {
    f32x4 acc[4][MT];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[j][i][q] = 0.0f;
    f32x4 gx_next[MT];
    XCD_STAMP(0);   // loop top
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, a.stamp_base + sg, sync + XcdSyncLayout::ERROR, flag)) return;   // h_{s-1} of the whole group is in the L2
    XCD_STAMP(1);   // group wait
    // ---- 16 MT lines x 64 columns, split-K over the four waves ----
    f32x4 ra[4][MT];
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int am = (zb * MT + i) * 16 + (lane & 15);
      const unsigned arow = (sg >= 1 && am < a.bs) ? ring_block((sg - 1) & 1, nd, dir, nblk, zb * MT + i, nkb) * 2u + akl : BUF_OOB_BASE;
#pragma unroll
      for (int g = 0; g < 4; g++) ra[g][i] = buf_load4_dev(abuf, g < ngrp ? arow + (unsigned)g * 1024u : BUF_OOB);
    }
    SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < MT; i++) gx_next[i] = gx_load(sg + 1, i);
    SCHED_FENCE();
    XCD_STAMP(2);   // loads issued
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g < ngrp) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int i = 0; i < MT; i++) acc[j][i] = mfma16x16x32_bf16(__builtin_bit_cast(u16x8, ra[g][i]), wreg[g][j], acc[j][i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) red[((wave * MT + i) * 16 + (lane >> 4) * 4 + q) * 68 + j * 16 + (lane & 15)] = acc[j][i][q];
    XCD_STAMP(3);   // ring loads returned + MFMAs + partial tile to LDS
    __syncthreads();
    XCD_STAMP(4);   // barrier
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const bool live = mine[i] && sg < T[i];
      float h = 0.0f, c_new = 0.0f;
      f32x4 act = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (live) {
        f32x4 k;
#pragma unroll
        for (int q = 0; q < 4; q++) k[q] = 0.0f;
#pragma unroll
        for (int w = 0; w < WIDE_NW; w++) {
          const f32x4 p = *reinterpret_cast<const f32x4*>(&red[((w * MT + i) * 16 + ml) * 68 + c16 * 4]);
#pragma unroll
          for (int q = 0; q < 4; q++) k[q] += p[q];
        }
        // (sigmoid / tanh as r = 1 / (1 + 2^(x scale)), act = r A + B: five VALU operations each, absolute error <= 3e-7 -- far
        // inside this mode's bf16 operands -- where gate_act spends ~25 on a wave that has its SIMD to itself)
        const float gi = act_affine(k[0] + gx[i][0], ACT_SIG_SCALE, 1.0f, 0.0f), gf = act_affine(k[1] + gx[i][1], ACT_SIG_SCALE, 1.0f, 0.0f),
                    go = act_affine(k[2] + gx[i][2], ACT_SIG_SCALE, 1.0f, 0.0f), ci = tanh_fast(k[3] + gx[i][3]);
        c_new = ci * gi + gf * c_prev[i];
        h = tanh_fast(c_new) * go;
        act[0] = gi; act[1] = gf; act[2] = go; act[3] = ci;
      }
      const float hn = quad_xor1(h);
      const unsigned hp = bf16_pack2(h, cell + 1 < no ? hn : 0.0f);   // (h = 0 for a line that has ended)
      fwd_out_store(a, ob, oo[i], act, c_new, h, hp, sg, live, live && !(c16 & 1), T[i]);
      c_prev[i] = c_new;
      gx[i] = gx_next[i];
    }
    // publish: every store of this workgroup acknowledged by the L2, then one arrival on the group's counter.  (Storing
    // the bf16 h first and the other arrays behind the arrival was measured SLOWER in round 2, 3.5 vs 3.2 us per step: VMEM
    // completes in order, so the next step's operand loads then wait behind those stores; measured again in round 5 on
    // today's 1.65 us step -- both forward kernels, CLSTM_FWD_LATE -- it is a wash: 1.3701 vs 1.3705 ms for the two forward
    // passes of configs[4], profiles/README.md: the wait behind the arrival grows by what the epilogue in front of it shrinks.)
    XCD_STAMP(5);   // epilogue + stores issued
    drain_vmem();
    XCD_STAMP(6);   // stores acknowledged
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, ct, a.stamp_base + sg + 1);
    XCD_STAMP(7);   // barrier + arrival
  }
  XCD_PROF_WRITE(xcd, ct, ntile);
}

// ---- the same kernel with the layer's INPUT PROJECTION folded in (MT = 1) -------------------------------------------------
// The hoisted product G = W_x x + b writes 4 no x ndir floats per frame (419 MB per layer at configs[4]) that the recurrence
// reads back once: 116 us (layer 1, K = 64: purely the store stream) + 321 us (layer 2, 215 GFLOP at 670 TFLOP/s) of a 4.83 ms
// step.  Neither the product nor its operands depend on the recurrence, so here the workgroup that owns a (16 lines x 64 gate
// columns) tile computes its x-part itself, in the SHADOW of the group hand-off: a step waits ~1,000 cycles for the other
// tiles' h (scripts/gpu_xcdprof.py) -- time in which the waves were idle.
//   * W_x fragments of the tile (64 columns x the wave's share of ni) stay in registers for the whole sequence, like R's;
//   * the 16 lines' input rows of step s+1 are requested as WHOLE rows one step ahead (lane = 16 bytes, a wave instruction
//     = 1 KB of one row: full 128-byte lines), parked in registers, written to LDS behind the arrival of step s, and read back
//     as A fragments (conflict-free ds_read_b128: row stride ni + 8 halfs) at the top of step s+1 BEFORE the group wait;
//   * their MFMAs start the step's accumulators; the recurrent MFMAs continue them behind the wait; the bias is added in the
//     epilogue.  G never holds pre-activations -- the epilogue writes the activations as before.
// Same products, f32 accumulation, one summation order per tile: the results differ from the hoisted form only by the order of
// the f32 additions.  NGX = 32-k groups of the input contraction per wave: wave w takes groups [w NGX, (w+1) NGX) of ni / 32.
template <int NGX>
DEVFN void lstm_xcd_fwd_bf16_fx_body(const LstmWideArgs& a) {
  constexpr int NX = NGX;                                     // 16-byte row chunks a thread stages per step: 16 lines x ni / 8 chunks over 256 threads (ni <= 128 NGX)
  unsigned short* wl = dyn_smem<unsigned short>();            // [64][XCD_LDW] while the weights are staged, then the x rows [16][ni + 8]
  float* red = reinterpret_cast<float*>(wl + 64 * XCD_LDW);   // [4][16][68]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * 16 * 68);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir, ni = a.x_ni;
  const int ntile = (no + 15) >> 4, nzb = a.zbn, ngroups = nd * nzb, ncg = (no + 3) >> 2;
  int* const sync = a.sync;
  int xcd, ct;
  if (!xcd_claim(sync, flag, ntile, ngroups, xcd, ct, a.debug_fail_claim)) return;
  const int dir = xcd % nd, zb = a.zb0 + xcd / nd;
  int* const gcount = sync + XcdSyncLayout::GROUP0 + xcd * XcdSyncLayout::GROUP_STRIDE;
  {
    const int c8 = a.kp16 >> 3;
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
  const int ml = tid >> 4, c16 = tid & 15;
  const int cell = ct * 16 + c16;
  const int line = zb * 16 + ml;
  int off = 0, T = 0;
  if (line < a.bs) { off = a.line_off[line]; T = a.line_off[line + 1] - off; }
  const bool mine = line < a.bs && cell < no;
  const int nblk = (a.bs + 15) >> 4, nkb = a.kp16 >> 5;
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Hb), (size_t)2 * nd * nblk * 16 * a.kp16 * 2);
  const int kw = a.kp16 / WIDE_NW, ngrp = kw >> 5;
  const unsigned akl = ring_elem(wave * ngrp, lane & 15, 8 * (lane >> 4)) * 2u;
  const unsigned short* wfrag = wl + (lane & 15) * XCD_LDW + wave * kw + 8 * (lane >> 4);
  __syncthreads();
  u16x8 wreg[4][4];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int j = 0; j < 4; j++) wreg[g][j] = *reinterpret_cast<const u16x8*>(wfrag + j * 16 * XCD_LDW + (g < ngrp ? g : 0) * 32);
  // ---- the input projection's operands ----
  const int ngx = ni >> 5;                                   // 32-k groups of the input contraction
  const int gx0 = wave * NGX;                                // this wave's first group
  u16x8 wxreg[NGX][4];
  {
    const BufF32 wxbuf = make_buf(reinterpret_cast<const float*>(a.Wxb), (size_t)nd * 4 * no * ni * 2);
#pragma unroll
    for (int g = 0; g < NGX; g++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int col = ct * 64 + j * 16 + (lane & 15);      // gate column of this direction: 4 cell + gate
        const bool lv = col < 4 * no && gx0 + g < ngx;
        wxreg[g][j] = __builtin_bit_cast(u16x8, buf_load4(wxbuf, lv ? (unsigned)(((dir * 4 * no + col) * ni + (gx0 + g) * 32 + 8 * (lane >> 4)) * 2) : BUF_OOB));
      }
  }
  f32x4 bias4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if (mine) bias4 = *reinterpret_cast<const f32x4*>(a.bias + dir * 4 * no + 4 * cell);
  // staging role: chunk c = tid + 256 i of the [16 lines][ni / 8] chunk grid
  const int XLD = ni + 8;                                    // halfs per LDS row
  unsigned short* xs = wl;
  const BufF32 xbuf = make_buf(reinterpret_cast<const float*>(a.Xb), (size_t)a.N * a.x_ld * 2);
  const int cpr = ni >> 3;
  int xoff[NX], xT[NX], xcol[NX], xlds[NX];
#pragma unroll
  for (int i = 0; i < NX; i++) {
    const int c = tid + WIDE_THREADS * i;
    const int row = c / cpr, cc = c - row * cpr;
    const int ln = zb * 16 + row;
    xoff[i] = 0; xT[i] = 0;
    if (row < 16 && ln < a.bs) { xoff[i] = a.line_off[ln]; xT[i] = a.line_off[ln + 1] - xoff[i]; }
    xcol[i] = cc * 8;
    xlds[i] = row < 16 ? row * XLD + cc * 8 : -1;
  }
  auto x_load = [&](const int sg, f32x4 (&r)[NX]) {          // the rows of step sg (zeros for lines that have ended)
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const bool lv = sg < xT[i];
      const int fr = xoff[i] + (dir == 0 ? sg : xT[i] - 1 - sg);
      r[i] = buf_load4(xbuf, lv ? (unsigned)(fr * a.x_ld + xcol[i]) * 2u : BUF_OOB);
    }
  };
  auto x_stage = [&](const f32x4 (&r)[NX]) {
#pragma unroll
    for (int i = 0; i < NX; i++)
      if (xlds[i] >= 0) *reinterpret_cast<f32x4*>(xs + xlds[i]) = r[i];
  };
  const unsigned short* xfrag = xs + (lane & 15) * XLD + gx0 * 32 + 8 * (lane >> 4);
  f32x4 xr[NX];
  x_load(0, xr);
  __syncthreads();                                           // every wave has its weight fragments: the staging area is free
  x_stage(xr);
  x_load(1, xr);
  float c_prev = 0.0f;
  const FwdOutBufs ob = fwd_out_bufs(a, dir, abuf, nblk, nkb);
  FwdOutOfs oo = fwd_out_ofs(a, dir, cell, line, off, T, nblk, nkb);
  XCD_PROF_DECL;
  for (int sg = 0; sg < a.tmax; sg++) {
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[j][q] = 0.0f;
    XCD_STAMP(0);   // loop top
    __syncthreads();                                         // the rows of this step are in LDS (written behind the previous arrival)
    {
      u16x8 xa[NGX];
#pragma unroll
      for (int g = 0; g < NGX; g++) xa[g] = *reinterpret_cast<const u16x8*>(xfrag + (gx0 + g < ngx ? g : 0) * 32);
#pragma unroll
      for (int g = 0; g < NGX; g++)
        if (gx0 + g < ngx) {
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j] = mfma16x16x32_bf16(xa[g], wxreg[g][j], acc[j]);
        }
    }
    XCD_STAMP(8);   // x-part: fragments + MFMAs (in the shadow of the hand-off)
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, a.stamp_base + sg, sync + XcdSyncLayout::ERROR, flag)) return;
    XCD_STAMP(1);   // group wait
    f32x4 ra[4];
    {
      const int am = zb * 16 + (lane & 15);
      const unsigned arow = (sg >= 1 && am < a.bs) ? ring_block((sg - 1) & 1, nd, dir, nblk, zb, nkb) * 2u + akl : BUF_OOB_BASE;
#pragma unroll
      for (int g = 0; g < 4; g++) ra[g] = buf_load4_dev(abuf, g < ngrp ? arow + (unsigned)g * 1024u : BUF_OOB);
    }
    SCHED_FENCE();
    XCD_STAMP(2);   // loads issued
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g < ngrp) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = mfma16x16x32_bf16(__builtin_bit_cast(u16x8, ra[g]), wreg[g][j], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) red[(wave * 16 + (lane >> 4) * 4 + q) * 68 + j * 16 + (lane & 15)] = acc[j][q];
    XCD_STAMP(3);
    __syncthreads();
    XCD_STAMP(4);
    {
      const bool live = mine && sg < T;
      float h = 0.0f, c_new = 0.0f;
      f32x4 act = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (live) {
        f32x4 k = bias4;
#pragma unroll
        for (int w = 0; w < WIDE_NW; w++) {
          const f32x4 p = *reinterpret_cast<const f32x4*>(&red[(w * 16 + ml) * 68 + c16 * 4]);
#pragma unroll
          for (int q = 0; q < 4; q++) k[q] += p[q];
        }
        const float gi = act_affine(k[0], ACT_SIG_SCALE, 1.0f, 0.0f), gf = act_affine(k[1], ACT_SIG_SCALE, 1.0f, 0.0f),
                    go = act_affine(k[2], ACT_SIG_SCALE, 1.0f, 0.0f), ci = tanh_fast(k[3]);   // (see lstm_xcd_fwd_bf16_body)
        c_new = ci * gi + gf * c_prev;
        h = tanh_fast(c_new) * go;
        act[0] = gi; act[1] = gf; act[2] = go; act[3] = ci;
      }
      const float hn = quad_xor1(h);
      const unsigned hp = bf16_pack2(h, cell + 1 < no ? hn : 0.0f);
      fwd_out_store(a, ob, oo, act, c_new, h, hp, sg, live, live && !(c16 & 1), T);
      c_prev = c_new;
    }
    XCD_STAMP(5);
    drain_vmem();                                            // (also: the rows of step sg + 1, requested a step ago, are in registers)
    XCD_STAMP(6);
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, ct, a.stamp_base + sg + 1);
    XCD_STAMP(7);
    // in the shadow of the hand-off: next step's rows into LDS (every wave read this step's fragments two barriers ago), the
    // rows of the step after that requested
    x_stage(xr);
    x_load(sg + 2, xr);
    XCD_STAMP(9);
  }
  XCD_PROF_WRITE(xcd, ct, ntile);
}
// (LDS: xcd_fwd_lds_bytes(1) -- the 16 x (ni + 8) halfs of input rows re-use the weight staging area: ni <= 2048)

// (Round 4, measured and not kept: the x-part of a 1024-input layer on FOUR EXTRA WAVES of the same workgroup -- recurrence and
// x-role meeting only at the workgroup's barriers, x rows staged through LDS a step ahead -- ran the configs[4] forward passes
// at 1.93 ms against 1.67 ms with the hoisted product: 256 VGPRs + spills for the x-role's 128 registers of W_x fragments,
// and five barriers per step that couple the two roles.  profiles/r04_fused_wx_variants.txt; the kernel is in the history.)

// ---- persistent backward recurrence, same scheme: 16 lines x 16 cells per workgroup, its 16 weight rows (R^T, 2048 k)
// resident in LDS, the group's bf16 delta ring exchanged through the XCD's L2, the carried state delta in a register ----
constexpr int XCD_LDWB = 2048 + 8;      // halfs per resident weight row of the backward tile, kp16 <= 2048
// Every workgroup of a group reads the group's WHOLE delta ring row block (16 lines x 2048 k x 2 B = 64 KB) each step
// (two 16-cell tiles per workgroup -- half that traffic per L2 -- was measured no faster: 7.51 vs 6.99 ms per minibatch).
inline __host__ __device__ int xcd_bwd_lds_bytes(int mt = 1) {
  const int need = 16 * XCD_LDWB * 2 + WIDE_NW * mt * 16 * (16 + 4) * 4 + 64;
  return need > 84 * 1024 ? need : 84 * 1024;   // > 80 KB: one workgroup per CU, whatever the tile needs
}

// MT: 16-line tiles per group, as in the forward kernel (the delta ring block a workgroup reads per step doubles with it:
// 128 KB at MT = 2 -- sixteen 16-byte loads per lane and line tile, all in flight at once)
// bias-gradient accumulation of the persistent bf16 backward kernels: the four gate deltas of a frame exactly as Dbf holds them
DEVFN void bias_acc(f32x4& s, const u32x2 pk) {
  s[0] += __builtin_bit_cast(float, pk[0] << 16);
  s[1] += __builtin_bit_cast(float, pk[0] & 0xffff0000u);
  s[2] += __builtin_bit_cast(float, pk[1] << 16);
  s[3] += __builtin_bit_cast(float, pk[1] & 0xffff0000u);
}
template <int MT>
DEVFN void lstm_xcd_bwd_bf16_body(const LstmWideArgs& a) {
  constexpr int LDR = 16 + 4;
  unsigned short* wl = dyn_smem<unsigned short>();                         // [16][XCD_LDWB]
  float* red = reinterpret_cast<float*>(wl + 16 * XCD_LDWB);               // [4][MT * 16][LDR]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * MT * 16 * LDR);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nzb = a.zbn, ngroups = nd * nzb;
  int* const sync = a.sync;
  int xcd, slot;
  if (!xcd_claim(sync, flag, ntile, ngroups, xcd, slot, a.debug_fail_claim)) return;
  const int dir = xcd % nd, zb = a.zb0 + xcd / nd;                         // zb: block of 16 MT lines
  int* const gcount = sync + XcdSyncLayout::GROUP0 + xcd * XcdSyncLayout::GROUP_STRIDE;
  {
    const int c8 = a.kp16 >> 3;
    for (int i = tid; i < 16 * c8; i += WIDE_THREADS) {
      const int row = i / c8, c = i - row * c8;
      *reinterpret_cast<u16x8*>(wl + row * XCD_LDWB + c * 8) =
          *reinterpret_cast<const u16x8*>(a.Rw16 + ((long long)(dir * ntile + slot) * 16 + row) * a.kp16 + c * 8);
    }
  }
  const int ml = tid >> 4, c16 = tid & 15;
  const int cell = slot * 16 + c16;
  int line[MT], off[MT], T[MT];
  bool mine[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) {
    line[i] = (zb * MT + i) * 16 + ml;
    off[i] = 0; T[i] = 0;
    if (line[i] < a.bs) { off[i] = a.line_off[line[i]]; T[i] = a.line_off[line[i] + 1] - off[i]; }
    mine[i] = line[i] < a.bs && cell < no;
  }
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * nd * no * 16);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * nd * no * 4);
  const BufF32 hbuf = make_buf(a.dH, (size_t)a.N * nd * no * 4);
  const int nblk = (a.bs + 15) >> 4, nkb = a.kp16 >> 5;
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Db), (size_t)2 * nd * nblk * 16 * a.kp16 * 2);
  const int kw = a.kp16 / WIDE_NW, ngrp = kw >> 5;   // <= 16 groups of 32 per wave
  const unsigned akl = ring_elem(wave * ngrp, lane & 15, 8 * (lane >> 4)) * 2u;
  const unsigned short* wfrag = wl + (lane & 15) * XCD_LDWB + wave * kw + 8 * (lane >> 4);
  float dc_carry[MT];      // dc_{s+1} * gf_{s+1} of this thread's (line, cell)s: carried in registers, not through memory
  f32x4 bsum[MT];          // sum over the line's frames of the four gate deltas as stored in Dbf (LstmWideArgs::dbias)
#pragma unroll
  for (int i = 0; i < MT; i++) { dc_carry[i] = 0.0f; bsum[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
  __syncthreads();
  u16x8 wreg[16];   // this wave's B fragments (16 rows x its quarter of the contraction) in registers for the whole sequence
#pragma unroll
  for (int g = 0; g < 16; g++) wreg[g] = *reinterpret_cast<const u16x8*>(wfrag + (g < ngrp ? g : 0) * 32);

  // Epilogue operands (forward-pass arrays, from HBM) are requested one step AHEAD and behind that step's delta loads, so
  // that they never sit in front of them in the in-order VMEM queue; c_s of a step is the c_{s-1} the previous step loaded.
  struct Ops { f32x4 act; float dh_in, c_m1; };
  auto ops_load = [&](int sg, int i) -> Ops {
    const bool lv = mine[i] && sg < T[i];
    const int ss = T[i] - 1 - sg;
    const long long nn = off[i] + (dir == 0 ? ss : sg);
    Ops o;
    o.act = buf_load4(gbuf, lv ? (unsigned)(((nn * nd + dir) * no + cell) * 16) : BUF_OOB);
    o.dh_in = buf_load(hbuf, lv ? (unsigned)((nn * (nd * no) + dir * no + cell) * 4) : BUF_OOB);
    o.c_m1 = buf_load(cbuf, lv && ss >= 1
        ? (unsigned)((((long long)(off[i] + (dir == 0 ? ss - 1 : sg + 1)) * nd + dir) * no + cell) * 4) : BUF_OOB);
    return o;
  };
  Ops cur[MT];
  float c_s[MT];
#pragma unroll
  for (int i = 0; i < MT; i++) {
    cur[i] = ops_load(0, i);
    const bool lv = mine[i] && 0 < T[i];
    c_s[i] = buf_load(cbuf, lv ? (unsigned)((((long long)(off[i] + (dir == 0 ? T[i] - 1 : 0)) * nd + dir) * no + cell) * 4) : BUF_OOB);
  }
  XCD_PROF_DECL;
  for (int sg = 0; sg < a.tmax; sg++) {
    XCD_STAMP(0);   // loop top
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, a.stamp_base + sg, sync + XcdSyncLayout::ERROR, flag)) return;
    XCD_STAMP(1);   // group wait
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[i][q] = 0.0f;
    Ops nxt[MT];
    {   // all sixteen 32-k groups of the wave's quarter requested at once: ONE L2 round trip per step (two rounds of
        // eight cost a second one: 4.3 vs 3.x us per step)
      f32x4 ra[16][MT];
#pragma unroll
      for (int i = 0; i < MT; i++) {
        const int am = (zb * MT + i) * 16 + (lane & 15);
        const unsigned arow = (sg >= 1 && am < a.bs) ? ring_block((sg - 1) & 1, nd, dir, nblk, zb * MT + i, nkb) * 2u + akl : BUF_OOB_BASE;
#pragma unroll
        for (int g = 0; g < 16; g++) ra[g][i] = buf_load4_dev(abuf, g < ngrp ? arow + (unsigned)g * 1024u : BUF_OOB);
      }
      SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < MT; i++) nxt[i] = ops_load(sg + 1, i);
      SCHED_FENCE();
      XCD_STAMP(2);   // loads issued
#pragma unroll
      for (int g = 0; g < 16; g++)
        if (g < ngrp) {
#pragma unroll
          for (int i = 0; i < MT; i++) acc[i] = mfma16x16x32_bf16(__builtin_bit_cast(u16x8, ra[g][i]), wreg[g], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int q = 0; q < 4; q++) red[((wave * MT + i) * 16 + (lane >> 4) * 4 + q) * LDR + (lane & 15)] = acc[i][q];
    XCD_STAMP(3);   // ring loads returned + MFMAs + partial tile to LDS
    __syncthreads();
    XCD_STAMP(4);   // barrier
    f32x4 dl[MT];
    bool live[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) {
      live[i] = mine[i] && sg < T[i];
      if (live[i]) {
        float dh_rec = 0.0f;
#pragma unroll
        for (int w = 0; w < WIDE_NW; w++) dh_rec += red[((w * MT + i) * 16 + ml) * LDR + c16];
        const float gi = cur[i].act[0], gf = cur[i].act[1], go = cur[i].act[2], ci = cur[i].act[3];
        const float dh = cur[i].dh_in + dh_rec;
        const float th = tanh_fast(c_s[i]);   // (2 sigmoid(2x) - 1: <= 3e-7 absolute, five VALU operations; see lstm_xcd_bwd_bf16_c32)
        const float d_go = th * dh;
        const float dc = (sg >= 1 ? dc_carry[i] : 0.0f) + (-th * th + 1.0f) * (go * dh);
        dc_carry[i] = dc * gf;
        const float d_gf = dc * cur[i].c_m1;
        const float d_gi = dc * ci, d_ci = dc * gi;
        dl[i][0] = (gi * (-gi + 1.0f)) * d_gi;
        dl[i][1] = (gf * (-gf + 1.0f)) * d_gf;
        dl[i][2] = (go * (-go + 1.0f)) * d_go;
        dl[i][3] = (-ci * ci + 1.0f) * d_ci;
        // (one 8-byte store: what the group waits for goes first)
        *reinterpret_cast<u32x2*>(a.Db + ring_block(sg & 1, nd, dir, nblk, zb * MT + i, nkb) + ring_elem(cell >> 3, ml, (4 * cell) & 31)) =
            u32x2{bf16_pack2(dl[i][0], dl[i][1]), bf16_pack2(dl[i][2], dl[i][3])};
      }
    }
    XCD_STAMP(5);   // epilogue + ring store issued
    drain_vmem();
    XCD_STAMP(6);   // stores acknowledged
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, slot, a.stamp_base + sg + 1);
    XCD_STAMP(7);   // barrier + arrival
#pragma unroll
    for (int i = 0; i < MT; i++) {
      if (live[i]) {
        const long long n = off[i] + (dir == 0 ? T[i] - 1 - sg : sg);
        if (!a.skip_d) *reinterpret_cast<f32x4*>(a.D + ((n * nd + dir) * no + cell) * 4) = dl[i];
        if (a.Dbf) {   // k-contiguous bf16 copy per frame: the ready-made A operand of the x.d product (gemm_b16kk)
          const u32x2 pk{bf16_pack2(dl[i][0], dl[i][1]), bf16_pack2(dl[i][2], dl[i][3])};
          *reinterpret_cast<u32x2*>(a.Dbf + (size_t)(n * nd + dir) * a.kp16 + 4 * cell) = pk;
          bias_acc(bsum[i], pk);
        }
      }
      c_s[i] = cur[i].c_m1;
      cur[i] = nxt[i];
    }
    XCD_STAMP(8);   // per-frame stores issued
  }
  if (a.dbias) {
#pragma unroll
    for (int i = 0; i < MT; i++)
      if (mine[i]) *reinterpret_cast<f32x4*>(a.dbias + (((long long)line[i] * nd + dir) * no + cell) * 4) = bsum[i];
  }
  XCD_PROF_WRITE(xcd, slot, ntile);
}

// ---- backward, 32 cells per workgroup: TWO groups per XCD (round 4) ----------------------------------------------------------
// A step of lstm_xcd_bwd_bf16 moves the group's whole delta block -- (lines of the group) x 4 no bf16 -- into EVERY workgroup of
// the group: 64 KB per step and CU at 16 lines x 512 cells = 1,024 cycles of a CU's 64 B/clk fill path, 2 MB per step out of
// one XCD's L2 (scripts/gpu_xcdprof.py: 1,350 cycles of load issue and 1,150 more until the last MFMA, of a 5,180-cycle step).
// That block scales with the LINES of a group, not with the cells of a workgroup.  Here a workgroup owns 32 cells (two MFMA
// column tiles; B fragments 2 x 64 VGPRs per lane, still resident for the whole sequence), a group is half as many workgroups,
// and the chip holds twice as many groups of half as many lines: 8 EPT lines per group (EPT = epilogue elements per thread:
// 256 threads = 8 lines x 32 cells), two groups per XCD, both in its L2.  EPT = 1 (up to 16 / ndir x 8 = 64 lines at two
// directions): eight lines -- the MFMA's rows 8..15 are requested out of range (zeros, no traffic) and a step fills 32 KB;
// EPT = 2 / 4: 16 / 32 lines per group where the 16-cell kernel needs 32 / 64.  The k split over the four waves, the order of
// a wave's MFMAs and the cross-wave sum are those of lstm_xcd_bwd_bf16: bit-identical results.  Needs no % 32 == 0.
inline __host__ __device__ int xcd_bwd_c32_lds_bytes(int ept) {
  const int lt = ept >= 2 ? ept / 2 : 1;
  const int need = 16 * XCD_LDWB * 2 + WIDE_NW * lt * 16 * (32 + 4) * 4 + 64;
  return need > 84 * 1024 ? need : 84 * 1024;   // > 80 KB: one workgroup per CU
}
// FULLK: the wave's quarter of the contraction is all sixteen 32-k groups (4 no = 2048): no per-group branches in the step
template <int EPT, bool FULLK>
DEVFN void lstm_xcd_bwd_bf16_c32_body(const LstmWideArgs& a) {
  constexpr int LT = EPT >= 2 ? EPT / 2 : 1;   // 16-row MFMA line tiles per group
  constexpr int LG = 8 * EPT;                  // lines per group
  constexpr int LDR = 32 + 4;
  unsigned short* wl = dyn_smem<unsigned short>();                         // [16][XCD_LDWB]: one 16-cell tile's rows at a time
  float* red = reinterpret_cast<float*>(wl + 16 * XCD_LDWB);               // [4][LT * 16][LDR]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * LT * 16 * LDR);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nt32 = ntile >> 1;   // workgroups per XCD | per group
  const int ngroups = nd * a.zbn;                        // groups of this launch: <= 16, two per XCD
  int* const sync = a.sync;
  int xcd, slot;
  if (!xcd_claim(sync, flag, ntile, (ngroups + 1) >> 1, xcd, slot, a.debug_fail_claim)) return;
  const int sub = slot >= nt32 ? 1 : 0, ct = slot - sub * nt32;
  const int grp = xcd * 2 + sub;
  if (grp >= ngroups) return;
  const int dir = grp % nd, lg = a.zb0 + grp / nd;                         // lg: group of LG lines
  int* const gcount = sync + XcdSyncLayout::GROUP0 + grp * XcdSyncLayout::GROUP_STRIDE;
  const int kw = a.kp16 / WIDE_NW, ngrp = kw >> 5;   // <= 16 groups of 32 per wave
  // this wave's B fragments -- 2 x (16 cells x its quarter of the contraction) -- in registers for the whole sequence
  u16x8 wreg[2][16];
  {
    const int c8 = a.kp16 >> 3;
    const unsigned short* wfrag = wl + (lane & 15) * XCD_LDWB + wave * kw + 8 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      for (int i = tid; i < 16 * c8; i += WIDE_THREADS) {
        const int row = i / c8, c = i - row * c8;
        *reinterpret_cast<u16x8*>(wl + row * XCD_LDWB + c * 8) =
            *reinterpret_cast<const u16x8*>(a.Rw16 + ((long long)(dir * ntile + 2 * ct + j) * 16 + row) * a.kp16 + c * 8);
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < 16; g++) wreg[j][g] = *reinterpret_cast<const u16x8*>(wfrag + (g < ngrp ? g : 0) * 32);
      __syncthreads();
    }
  }
  // epilogue role: cell c32 of the tile, lines lrow + 8 e of the group
  const int lrow = tid >> 5, c32 = tid & 31;
  const int cell = ct * 32 + c32;
  int line[EPT], off[EPT], T[EPT];
  bool mine[EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    line[e] = lg * LG + lrow + 8 * e;
    off[e] = 0; T[e] = 0;
    if (line[e] < a.bs) { off[e] = a.line_off[line[e]]; T[e] = a.line_off[line[e] + 1] - off[e]; }
    mine[e] = line[e] < a.bs && cell < no;
  }
  const size_t gbytes = (size_t)a.N * nd * no * 16;
  const BufF32 gbuf = make_buf(a.G, gbytes);
  const BufF32 cbuf = make_buf(a.C, gbytes / 4);
  const BufF32 hbuf = make_buf(a.dH, gbytes / 4);
  const BufF32 dbuf = make_buf(a.skip_d ? nullptr : a.D, a.skip_d ? 0 : gbytes);                                  // (no records: every store dropped)
  const BufF32 dbfbuf = make_buf(reinterpret_cast<const float*>(a.Dbf), a.Dbf ? (size_t)a.N * nd * a.kp16 * 2 : 0);
  const int nblk = (a.bs + 15) >> 4, nkb = a.kp16 >> 5;
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Db), (size_t)2 * nd * nblk * 16 * a.kp16 * 2);
  // A fragment of this lane in line tile i: line lg LG + 16 i + (lane & 15) -- a row of the 16-line ring block it lies in
  unsigned aoff[LT];
  int ablk[LT];
  bool aval[LT];
#pragma unroll
  for (int i = 0; i < LT; i++) {
    const int rl = 16 * i + (lane & 15), am = lg * LG + rl;
    aval[i] = rl < LG && am < a.bs;
    ablk[i] = am >> 4;
    aoff[i] = ring_elem(wave * ngrp, am & 15, 8 * (lane >> 4)) * 2u;
  }
  // EPT = 1, dense form: lane (r, kc) holds row r & 7 of the eight lines, 32-k group 2p + (r >> 3) in its p-th load
  const int dhalf = (lane & 15) >> 3;
  const int dline = lg * LG + (lane & 7);
  const bool dval = dline < a.bs;
  const int dblk = dline >> 4;
  const unsigned doff = ring_elem(wave * ngrp + dhalf, dline & 15, 8 * (lane >> 4)) * 2u;
  float dc_carry[EPT];
  f32x4 bsum[EPT];         // sum over the line's frames of the four gate deltas as stored in Dbf (LstmWideArgs::dbias)
#pragma unroll
  for (int e = 0; e < EPT; e++) { dc_carry[e] = 0.0f; bsum[e] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }

  // Byte offsets of a thread's element in the per-frame arrays, advanced by one frame per step instead of recomputed (a 64-bit
  // multiply-add chain per access; the arrays stay below 2 GiB -- host check -- so 32 bits carry them): o4 = ((n nd + dir) no + cell) 4
  // addresses C and dH, o4 << 2 addresses G and D; ol runs one step ahead (the operand loads), os / ob with the step (D / Dbf).
  const int fstep = dir == 0 ? -1 : 1;   // the backward walk: frames T-1 .. 0 of direction 0, 0 .. T-1 of the reversed one
  const unsigned s4 = (unsigned)(fstep * nd * no * 4), sB = (unsigned)(fstep * nd * a.kp16 * 2);
  unsigned ol[EPT], os[EPT], ob[EPT], rofs[EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    const long long n0 = off[e] + (dir == 0 ? T[e] - 1 : 0);
    ol[e] = os[e] = (unsigned)(((n0 * nd + dir) * no + cell) * 4);
    ob[e] = (unsigned)(((n0 * nd + dir) * a.kp16 + 4 * cell) * 2);
    rofs[e] = (ring_block(0, nd, dir, nblk, line[e] >> 4, nkb) + ring_elem(cell >> 3, line[e] & 15, (4 * cell) & 31)) * 2u;
  }
  const unsigned ring_parity = ring_block(1, nd, 0, nblk, 0, nkb) * 2u;   // bytes between the two halves of the ring

  // Epilogue operands (forward-pass arrays, from HBM) are requested one step AHEAD and behind that step's delta loads, so
  // that they never sit in front of them in the in-order VMEM queue; c_s of a step is the c_{s-1} the previous step loaded.
  struct Ops { f32x4 act; float dh_in, c_m1; };
  auto ops_load = [&](int sg, int e) -> Ops {   // (called once per step, in order: advances ol)
    const bool lv = mine[e] && sg < T[e];
    Ops o;
    o.act = buf_load4(gbuf, lv ? ol[e] << 2 : BUF_OOB);
    o.dh_in = buf_load(hbuf, lv ? ol[e] : BUF_OOB);
    ol[e] += s4;
    o.c_m1 = buf_load(cbuf, lv && sg + 1 < T[e] ? ol[e] : BUF_OOB);
    return o;
  };
  Ops cur[EPT];
  float c_s[EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    c_s[e] = buf_load(cbuf, mine[e] && 0 < T[e] ? ol[e] : BUF_OOB);
    cur[e] = ops_load(0, e);
  }
  XCD_PROF_DECL;
  for (int sg = 0; sg < a.tmax; sg++) {
    XCD_STAMP(0);   // loop top
    if (sg >= 1 && !xcd_wait_group(gcount, nt32, a.stamp_base + sg, sync + XcdSyncLayout::ERROR, flag)) return;
    XCD_STAMP(1);   // group wait
    f32x4 acc[LT][2];
#pragma unroll
    for (int i = 0; i < LT; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[i][j][q] = 0.0f;
    Ops nxt[EPT];
    if constexpr (EPT == 1) {
      // Eight lines: a load instruction costs the CU's address path its 64 lanes whether or not they hit memory, so the lanes
      // of MFMA rows 8..15 are not parked out of range but fetch the NEXT 32-k group of rows 0..7 -- eight dense loads instead
      // of sixteen half-empty ones.  Group 2p multiplies the registers as they are (its rows 8..15 hold the other group's
      // deltas: they only reach output rows 8..15, which nobody reads), group 2p+1 after a row rotation by eight lanes (DPP).
      // Same groups in the same order: the sums are those of the sixteen-load form.
      // (All eight loads first: issuing them a few ahead with the arrived pair's MFMAs in between was measured SLOWER -- 1.64 /
      // 1.60 / 1.53 ms for 3 / 4 / 6 ahead against 1.46 for all at once, two backward passes of configs[4]: the wait in front of
      // each MFMA group is for the data's round trip, and only all loads in flight together hide it.  Leave-out builds: the loads
      // cost 630 cycles of a step, the MFMAs + rotations another 630, additively.)
      f32x4 ra[8];
      const unsigned arow = (sg >= 1 && dval) ? ring_block((sg - 1) & 1, nd, dir, nblk, dblk, nkb) * 2u + doff : BUF_OOB_BASE;
#pragma unroll
      for (int p = 0; p < 8; p++) ra[p] = buf_load4_dev(abuf, FULLK || 2 * p + dhalf < ngrp ? arow + (unsigned)p * 2048u : BUF_OOB);
      SCHED_FENCE();
      nxt[0] = ops_load(sg + 1, 0);
      SCHED_FENCE();
      XCD_STAMP(2);   // loads issued
#pragma unroll
      for (int p = 0; p < 8; p++) {
        if (FULLK || 2 * p < ngrp) {
#pragma unroll
          for (int j = 0; j < 2; j++) acc[0][j] = mfma16x16x32_bf16(__builtin_bit_cast(u16x8, ra[p]), wreg[j][2 * p], acc[0][j]);
        }
        if (FULLK || 2 * p + 1 < ngrp) {
          f32x4 t;
#pragma unroll
          for (int q = 0; q < 4; q++) t[q] = row_ror<8>(ra[p][q]);
#pragma unroll
          for (int j = 0; j < 2; j++) acc[0][j] = mfma16x16x32_bf16(__builtin_bit_cast(u16x8, t), wreg[j][2 * p + 1], acc[0][j]);
        }
      }
    } else
    {   // all sixteen 32-k groups of the wave's quarter requested at once: one L2 round trip per step
      f32x4 ra[16][LT];
#pragma unroll
      for (int i = 0; i < LT; i++) {
        const unsigned arow = (sg >= 1 && aval[i]) ? ring_block((sg - 1) & 1, nd, dir, nblk, ablk[i], nkb) * 2u + aoff[i] : BUF_OOB_BASE;
#pragma unroll
        for (int g = 0; g < 16; g++) ra[g][i] = buf_load4_dev(abuf, FULLK || g < ngrp ? arow + (unsigned)g * 1024u : BUF_OOB);
      }
      SCHED_FENCE();
#pragma unroll
      for (int e = 0; e < EPT; e++) nxt[e] = ops_load(sg + 1, e);
      SCHED_FENCE();
      XCD_STAMP(2);   // loads issued
#pragma unroll
      for (int g = 0; g < 16; g++)
        if (FULLK || g < ngrp) {
#pragma unroll
          for (int i = 0; i < LT; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = mfma16x16x32_bf16(__builtin_bit_cast(u16x8, ra[g][i]), wreg[j][g], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < LT; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) red[((wave * LT + i) * 16 + (lane >> 4) * 4 + q) * LDR + j * 16 + (lane & 15)] = acc[i][j][q];
    XCD_STAMP(3);   // ring loads returned + MFMAs + partial tile to LDS
    __syncthreads();
    XCD_STAMP(4);   // barrier
    f32x4 dl[EPT];
    u32x2 pk[EPT];
    bool live[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      live[e] = mine[e] && sg < T[e];
      dl[e] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (live[e]) {
        const int ll = lrow + 8 * e;
        float dh_rec = 0.0f;
#pragma unroll
        for (int w = 0; w < WIDE_NW; w++) dh_rec += red[((w * LT + (ll >> 4)) * 16 + (ll & 15)) * LDR + c32];
        const float gi = cur[e].act[0], gf = cur[e].act[1], go = cur[e].act[2], ci = cur[e].act[3];
        const float dh = cur[e].dh_in + dh_rec;
        // (tanh as 2 sigmoid(2x) - 1: absolute error <= 3e-7, far inside this mode's bf16 operands; five VALU operations
        // instead of gate_act's ~25 on a wave that has the SIMD to itself)
        const float th = tanh_fast(c_s[e]);
        const float d_go = th * dh;
        const float dc = (sg >= 1 ? dc_carry[e] : 0.0f) + (-th * th + 1.0f) * (go * dh);
        dc_carry[e] = dc * gf;
        const float d_gf = dc * cur[e].c_m1;
        const float d_gi = dc * ci, d_ci = dc * gi;
        dl[e][0] = (gi * (-gi + 1.0f)) * d_gi;
        dl[e][1] = (gf * (-gf + 1.0f)) * d_gf;
        dl[e][2] = (go * (-go + 1.0f)) * d_go;
        dl[e][3] = (-ci * ci + 1.0f) * d_ci;
      }
      // (one 8-byte store: what the group waits for goes first; a lane without a live element stores out of range)
      pk[e] = u32x2{bf16_pack2(dl[e][0], dl[e][1]), bf16_pack2(dl[e][2], dl[e][3])};
      buf_store_u32x2_s(abuf, live[e] ? rofs[e] : BUF_OOB_BASE, (sg & 1) ? ring_parity : 0u, pk[e]);
    }
    XCD_STAMP(5);   // epilogue + ring store issued
    drain_vmem();
    XCD_STAMP(6);   // stores acknowledged
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, ct, a.stamp_base + sg + 1);
    XCD_STAMP(7);   // barrier + arrival
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      if (!a.skip_d) buf_store4(dbuf, live[e] ? os[e] << 2 : BUF_OOB, dl[e]);
      // k-contiguous bf16 copy per frame: the ready-made A operand of the x.d product (gemm_b16kk)
      buf_store_u32x2_s(dbfbuf, live[e] ? ob[e] : BUF_OOB, 0u, pk[e]);
      bias_acc(bsum[e], pk[e]);   // (a dead element's deltas are zeros)
      os[e] += s4; ob[e] += sB;
      c_s[e] = cur[e].c_m1;
      cur[e] = nxt[e];
    }
    XCD_STAMP(8);   // per-frame stores issued
  }
  if (a.dbias) {
#pragma unroll
    for (int e = 0; e < EPT; e++)
      if (mine[e]) *reinterpret_cast<f32x4*>(a.dbias + (((long long)line[e] * nd + dir) * no + cell) * 4) = bsum[e];
  }
  XCD_PROF_WRITE(xcd, slot, ntile);
}

// ---- the same persistent per-XCD scheme with f32 operands (the parity-grade path of wide layers) -----------------------
// v_mfma_f32_16x16x4_f32, weight rows f32 in LDS (64 x (kp + 4) forward = 132 KB at 512 cells, 16 x (kp + 4) backward),
// h_{t-1} / delta_{t+1} exchanged through the tiled f32 ring Rf (ring32_*: written beside the per-frame store into H / D; plain
// stores by the group's workgroups, L1-bypassing loads -- one L2; until round 3 the rows were read from H / D themselves: sixteen
// 64-byte pieces of sixteen frames per load instruction), fragments as in wide_tile: lane (i, kq) loads four consecutive k of
// row i, MFMA e of a 16-k group uses element e of both operands' float4.  Arithmetic identical to the per-step kernels (same tile, same
// split-K order), so the results are bit-identical to them.
inline __host__ __device__ int xcd_fwd_f32_lds_bytes(int kp) { return (64 * (kp + WIDE_WPAD) + WIDE_NW * 16 * 68 + 16) * 4; }
inline __host__ __device__ int xcd_bwd_f32_lds_bytes(int kp) {
  const int need = (16 * (kp + WIDE_WPAD) + WIDE_NW * 16 * WIDE_LDW + 16) * 4;
  return need > 84 * 1024 ? need : 84 * 1024;
}

DEVFN void lstm_xcd_fwd_f32_body(const LstmWideArgs& a) {
  float* wl = dyn_smem<float>();                                   // [64][kp + 4]
  const int ldw = a.kp + WIDE_WPAD;
  float* red = wl + 64 * ldw;                                      // [4][16][68]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * 16 * 68);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nzb = a.zbn, ncg = (no + 3) >> 2;
  int xcd, ct;
  if (!xcd_claim(a.sync, flag, ntile, nd * nzb, xcd, ct, a.debug_fail_claim)) return;
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
  const int nblk = (a.bs + 15) >> 4, nkb = a.kp >> 4;
  const BufF32 hbuf = make_buf(a.Rf, ring32_floats(nd, a.bs, a.kp) * 4);
  const int kw = a.kp / WIDE_NW, ngrp = kw >> 4;     // 16-k groups per wave (<= 8 at 512 cells)
  const unsigned klane = ring32_elem(wave * ngrp, lane & 15, 4 * (lane >> 4)) * 4u;
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
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, a.stamp_base + sg, a.sync + XcdSyncLayout::ERROR, flag)) return;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[j][q] = 0.0f;
    const unsigned arow = (sg >= 1 && am < a.bs && sg < aT) ? ring32_block((sg - 1) & 1, nd, dir, nblk, zb, nkb) * 4u + klane : BUF_OOB_BASE;
    f32x4 ra[8];
#pragma unroll
    for (int g = 0; g < 8; g++) ra[g] = buf_load4_dev(hbuf, g < ngrp ? arow + (unsigned)g * 1024u : BUF_OOB);
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
      a.Rf[ring32_block(sg & 1, nd, dir, nblk, zb, nkb) + ring32_elem(cell >> 4, ml, cell & 15)] = h;   // next step's A operand of the whole group
      a.H[n * a.ldh + a.hofs + dir * no + cell] = h;
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
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, ct, a.stamp_base + sg + 1);
  }
}

DEVFN void lstm_xcd_bwd_f32_body(const LstmWideArgs& a) {
  float* wl = dyn_smem<float>();                                   // [16][kp + 4]
  const int ldw = a.kp + WIDE_WPAD;
  float* red = wl + 16 * ldw;                                      // [4][16][WIDE_LDW]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * 16 * WIDE_LDW);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nzb = a.zbn;
  int xcd, ct;
  if (!xcd_claim(a.sync, flag, ntile, nd * nzb, xcd, ct, a.debug_fail_claim)) return;
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
  const int nblk = (a.bs + 15) >> 4, nkb = a.kp >> 4;
  const BufF32 dbuf = make_buf(a.Rf, ring32_floats(nd, a.bs, a.kp) * 4);
  const int kw = a.kp / WIDE_NW, ngrp = kw >> 4;     // 16-k groups per wave (32 at 512 cells), walked in rounds of 8
  const unsigned klane = ring32_elem(wave * ngrp, lane & 15, 4 * (lane >> 4)) * 4u;
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
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, a.stamp_base + sg, a.sync + XcdSyncLayout::ERROR, flag)) return;
    f32x4 acc;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = 0.0f;
    const unsigned arow = (sg >= 1 && am < a.bs && sg < aT) ? ring32_block((sg - 1) & 1, nd, dir, nblk, zb, nkb) * 4u + klane : BUF_OOB_BASE;
    // rounds of 8 groups, the next round's loads requested before this round's MFMAs
    f32x4 ra[2][8];
#pragma unroll
    for (int g = 0; g < 8; g++) ra[0][g] = buf_load4_dev(dbuf, g < ngrp ? arow + (unsigned)g * 1024u : BUF_OOB);
    SCHED_FENCE();
    Ops nxt = ops_load(sg + 1);
    SCHED_FENCE();
    for (int r0 = 0; r0 < ngrp; r0 += 16) {
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int gb = r0 + half * 8;
#pragma unroll
        for (int g = 0; g < 8; g++) ra[half ^ 1][g] = buf_load4_dev(dbuf, gb + 8 + g < ngrp ? arow + (unsigned)(gb + 8 + g) * 1024u : BUF_OOB);
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
      *reinterpret_cast<f32x4*>(a.Rf + ring32_block(sg & 1, nd, dir, nblk, zb, nkb) + ring32_elem(cell >> 2, ml, 4 * (cell & 3))) = dl;   // next step's A operand of the whole group
      *reinterpret_cast<f32x4*>(a.D + ((n * nd + dir) * no + cell) * 4) = dl;
    }
    c_s = cur.c_m1;
    cur = nxt;
    drain_vmem();
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, ct, a.stamp_base + sg + 1);
  }
}

// contraction padding of the packed weights
inline int wide_kp_fwd(int no) { return ((no + 16 * WIDE_NW - 1) / (16 * WIDE_NW)) * 16 * WIDE_NW; }
inline int wide_kp_bwd(int no) { return ((4 * no + 16 * WIDE_NW - 1) / (16 * WIDE_NW)) * 16 * WIDE_NW; }
// bf16 rows: every wave's quarter of the contraction is whole k32 groups
inline int wide_kp16_fwd(int no) { return ((no + 32 * WIDE_NW - 1) / (32 * WIDE_NW)) * 32 * WIDE_NW; }
inline int wide_kp16_bwd(int no) { return ((4 * no + 32 * WIDE_NW - 1) / (32 * WIDE_NW)) * 32 * WIDE_NW; }

// ---- exact-f32 mode: the BACKWARD persistent recurrence as an f32-grade "x3" product on the bf16 MFMA (round 4) ---------------
// lstm_xcd_bwd_f32 spends 128 v_mfma_f32_16x16x4_f32 per wave and step -- 4,096 of its ~9,700 cycles -- because the f32 MFMA has
// a sixteenth of the bf16 MFMA's rate.  The backward products of this mode (weight gradients, input deltas) already run as
// hi + lo split bf16 products (hi.hi + hi.lo + lo.hi, < 2^-16 per product: gemm_bf16.h); here the recurrent product R^T delta
// joins them: the gate deltas travel as TWO planes of the bf16 kernels' tiled ring (hi | lo: the same 4 bytes per value as the f32
// ring), the weights are two register-resident fragment sets (ops.h:k_pack_wide_split), a 32-k group costs three 16-cycle MFMAs
// instead of eight 32-cycle ones.  Everything else -- operands from the forward pass, gate_act's exact tanh, the f32 delta array D
// -- is the f32 kernel's.  At configs[4]'s full size the minibatch gradient sits 1.45e-5 of its largest entry from the float64
// oracle, the f32 MFMA kernel 1.44e-5, the f32 oracle itself 3.04e-5 (tests/test_gpu_e2e.py::test_configs4_full_shape_f32_vs_oracle).
// The FORWARD recurrence stays on the f32 MFMA: its three-term form (six or all nine products; 2.58 -> 2.10 / 2.20 ms) kept every
// saved activation inside the 1e-4 bar but moved the softmax outputs of one line of that deliberately ill-conditioned case to
// 1.3e-5 / 1.1e-5 from float64 (f32 MFMA: 3.4e-6) -- enough to break the test's 1e-3 bar on the CTC posteriors; measured, removed
// (profiles/r04_f32_grade_recurrence.txt).  CLSTM_GEMM_X3=0 / clstm_net_set_strict_f32 / CLSTM_REC_X3=0 keep the f32 MFMA kernel.
DEVFN void split2_bf16(const float x, unsigned& hi, unsigned& lo) {   // 16-bit patterns in the low halves; x = hi + lo to 2^-17
  hi = bf16_pack2(x, 0.0f) & 0xFFFFu;
  lo = bf16_pack2(x - __builtin_bit_cast(float, hi << 16), 0.0f) & 0xFFFFu;
}

DEVFN void lstm_xcd_bwd_x3_body(const LstmWideArgs& a) {
  constexpr int LDR = 16 + 4;
  constexpr int NA = 2, NB = 2, NT = 2;   // planes of the deltas / of the weights; a product of planes i, j is kept when i + j < NT
  unsigned short* wl = dyn_smem<unsigned short>();                         // [16][XCD_LDWB], one plane at a time
  float* red = reinterpret_cast<float*>(wl + 16 * XCD_LDWB);               // [4][16][LDR]
  int* flag = reinterpret_cast<int*>(red + WIDE_NW * 16 * LDR);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int no = a.no, nd = a.ndir;
  const int ntile = (no + 15) >> 4, nzb = a.zbn, ngroups = nd * nzb;
  int* const sync = a.sync;
  int xcd, slot;
  if (!xcd_claim(sync, flag, ntile, ngroups, xcd, slot, a.debug_fail_claim)) return;
  const int dir = xcd % nd, zb = a.zb0 + xcd / nd;
  int* const gcount = sync + XcdSyncLayout::GROUP0 + xcd * XcdSyncLayout::GROUP_STRIDE;
  const int kw = a.kp16 / WIDE_NW, ngrp = kw >> 5;   // <= 16 groups of 32 per wave
  u16x8 wreg[NB][16];
  {
    const int c8 = a.kp16 >> 3;
    const unsigned short* wfrag = wl + (lane & 15) * XCD_LDWB + wave * kw + 8 * (lane >> 4);
#pragma unroll
    for (int p = 0; p < NB; p++) {
      for (int i = tid; i < 16 * c8; i += WIDE_THREADS) {
        const int row = i / c8, c = i - row * c8;
        *reinterpret_cast<u16x8*>(wl + row * XCD_LDWB + c * 8) =
            *reinterpret_cast<const u16x8*>(a.Rw16 + p * a.rw_plane + ((long long)(dir * ntile + slot) * 16 + row) * a.kp16 + c * 8);
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < 16; g++) wreg[p][g] = *reinterpret_cast<const u16x8*>(wfrag + (g < ngrp ? g : 0) * 32);
      __syncthreads();
    }
  }
  const int ml = tid >> 4, c16 = tid & 15;
  const int cell = slot * 16 + c16, line = zb * 16 + ml;
  int off = 0, T = 0;
  if (line < a.bs) { off = a.line_off[line]; T = a.line_off[line + 1] - off; }
  const bool mine = line < a.bs && cell < no;
  const size_t gbytes = (size_t)a.N * nd * no * 16;
  const BufF32 gbuf = make_buf(a.G, gbytes), cbuf = make_buf(a.C, gbytes / 4), hbuf = make_buf(a.dH, gbytes / 4), dbuf = make_buf(a.D, gbytes);
  const int nblk = (a.bs + 15) >> 4, nkb = a.kp16 >> 5;
  const unsigned plane_b = (unsigned)a.ring_plane * 2u;
  const BufF32 abuf = make_buf(reinterpret_cast<const float*>(a.Db), (size_t)NA * a.ring_plane * 2);
  const unsigned akl = ring_elem(wave * ngrp, lane & 15, 8 * (lane >> 4)) * 2u;
  const int am = zb * 16 + (lane & 15);
  // running byte offsets (see lstm_xcd_bwd_bf16_c32): ol one step ahead for the operand loads, os with the step for D
  const int fstep = dir == 0 ? -1 : 1;
  const unsigned s4 = (unsigned)(fstep * nd * no * 4);
  unsigned ol, os;
  {
    const long long n0 = off + (dir == 0 ? T - 1 : 0);
    ol = os = (unsigned)(((n0 * nd + dir) * no + cell) * 4);
  }
  const unsigned rofs = (ring_block(0, nd, dir, nblk, zb, nkb) + ring_elem(cell >> 3, ml, (4 * cell) & 31)) * 2u;
  const unsigned ring_parity = ring_block(1, nd, 0, nblk, 0, nkb) * 2u;
  struct Ops { f32x4 act; float dh_in, c_m1; };
  auto ops_load = [&](int sg) -> Ops {   // (called once per step, in order: advances ol)
    const bool lv = mine && sg < T;
    Ops o;
    o.act = buf_load4(gbuf, lv ? ol << 2 : BUF_OOB);
    o.dh_in = buf_load(hbuf, lv ? ol : BUF_OOB);
    ol += s4;
    o.c_m1 = buf_load(cbuf, lv && sg + 1 < T ? ol : BUF_OOB);
    return o;
  };
  float c_s = buf_load(cbuf, mine && 0 < T ? ol : BUF_OOB);
  Ops cur = ops_load(0);
  float dc_carry = 0.0f;
  for (int sg = 0; sg < a.tmax; sg++) {
    if (sg >= 1 && !xcd_wait_group(gcount, ntile, a.stamp_base + sg, sync + XcdSyncLayout::ERROR, flag)) return;
    f32x4 acc;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = 0.0f;
    Ops nxt;
    {
      f32x4 ra[NA][16];
      const unsigned arow = (sg >= 1 && am < a.bs) ? ring_block((sg - 1) & 1, nd, dir, nblk, zb, nkb) * 2u + akl : BUF_OOB_BASE;
#pragma unroll
      for (int g = 0; g < 16; g++)
#pragma unroll
        for (int p = 0; p < NA; p++) ra[p][g] = buf_load4_dev(abuf, g < ngrp ? arow + (unsigned)p * plane_b + (unsigned)g * 1024u : BUF_OOB);
      SCHED_FENCE();
      nxt = ops_load(sg + 1);
      SCHED_FENCE();
#pragma unroll
      for (int g = 0; g < 16; g++)
        if (g < ngrp) {
#pragma unroll
          for (int t = NT - 1; t >= 0; t--)     // small terms first: planes (i, j) with i + j = t
#pragma unroll
            for (int i = 0; i < NA; i++) {
              const int j = t - i;
              if (j >= 0 && j < NB) acc = mfma16x16x32_bf16(__builtin_bit_cast(u16x8, ra[i][g]), wreg[j][g], acc);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) red[(wave * 16 + (lane >> 4) * 4 + q) * LDR + (lane & 15)] = acc[q];
    __syncthreads();
    const bool live = mine && sg < T;
    f32x4 dl = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
      float dh_rec = 0.0f;
#pragma unroll
      for (int w = 0; w < WIDE_NW; w++) dh_rec += red[(w * 16 + ml) * LDR + c16];
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
    }
    {   // the two planes of the four deltas: 8 bytes per plane (what the group waits for goes first)
      unsigned pl[4][2];
#pragma unroll
      for (int q = 0; q < 4; q++) split2_bf16(dl[q], pl[q][0], pl[q][1]);
      const unsigned par = (sg & 1) ? ring_parity : 0u;
#pragma unroll
      for (int p = 0; p < NA; p++)
        buf_store_u32x2_s(abuf, live ? rofs + (unsigned)p * plane_b : BUF_OOB_BASE, par, u32x2{pl[0][p] | (pl[1][p] << 16), pl[2][p] | (pl[3][p] << 16)});
    }
    drain_vmem();
    __syncthreads();
    if (tid == 0 && sg + 1 < a.tmax) xcd_arrive(gcount, slot, a.stamp_base + sg + 1);
    buf_store4(dbuf, live ? os << 2 : BUF_OOB, dl);
    os += s4;
    c_s = cur.c_m1;
    cur = nxt;
  }
}

// ---- the persistent kernels: body + xcd_finish ----
template <int MT>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_fwd_bf16(LstmWideArgs a) { lstm_xcd_fwd_bf16_body<MT>(a); xcd_finish(a); }
template <int NGX>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_fwd_bf16_fx(LstmWideArgs a) { lstm_xcd_fwd_bf16_fx_body<NGX>(a); xcd_finish(a); }
template <int MT>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_bwd_bf16(LstmWideArgs a) { lstm_xcd_bwd_bf16_body<MT>(a); xcd_finish(a); }
template <int EPT, bool FULLK>
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_bwd_bf16_c32(LstmWideArgs a) { lstm_xcd_bwd_bf16_c32_body<EPT, FULLK>(a); xcd_finish(a); }
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_fwd_f32(LstmWideArgs a) { lstm_xcd_fwd_f32_body(a); xcd_finish(a); }
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_bwd_x3(LstmWideArgs a) { lstm_xcd_bwd_x3_body(a); xcd_finish(a); }
__global__ __launch_bounds__(WIDE_THREADS) void lstm_xcd_bwd_f32(LstmWideArgs a) { lstm_xcd_bwd_f32_body(a); xcd_finish(a); }

}  // namespace clstm
