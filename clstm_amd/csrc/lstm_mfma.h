// lstm_mfma.h -- the NARROW layer's forward recurrence batched over lines on the matrix cores (minibatches that fill the chip).
//
// lstm_seq.h walks one line per workgroup: R.h_{t-1} is 100 x 100 x 4 packed f32 FMAs per line and step, the MFMA pipes idle,
// and once every CU holds two such workgroups (256 lines) the kernel is VALU-bound (27 % of the f32 vector peak).  Here ONE
// workgroup owns 16 lines x one direction and the WHOLE gate product of GenericNPLSTM::forward (clstm.cc:612-620:
// forward_stack_delay + four forward_full1 = forward_lin1 + forward_nonlin0, clstm_compute.cc:275-314, 377-397) is a real GEMM
//     pre[4 no x 16 lines] = [R | W_x | b][4 no x (no + ni + 1)] . [h_{t-1} ; x_t ; 1][(no + ni + 1) x 16 lines]
// on v_mfma_f32_16x16x32_f16 with f32 accumulation: no hoisted W_x GEMM in front of this kernel, no pre-activation array.
// (Round 6, v1 read the hoisted product's pre-activations back row by row: 70 KB of global traffic per step on ONE CU, 28 B/clk,
// 2,500 cycles per step for the memory pipeline alone -- profiles/r06_mfma_v1_phase_cycles_256.txt.  With x_t as 48 more k the
// input side of a step is 3 KB.)
//
// Arithmetic.  Both operands are f32 values split into TWO f16 terms, x = hi + lo with hi = f16(x), lo = f16(x - hi) (the
// difference is exact), after a power-of-two scaling that keeps lo out of the f16 subnormal range ([R | W_x | b] by 2^e with
// max |.| 2^e in [2^13, 2^14); h in [-1, 1], x and the constant 1 by 2^8: inputs must stay below 255 in magnitude, normalised
// text lines are in [0, 1]; the inverse scale rides the multiply in front of the gate nonlinearity).  f16 carries 11 significant
// bits, so hi + lo represents x to 2^-22 |x| or better (f32 itself: 2^-24), each f16 x f16 product is exact in the f32
// accumulator, and a product is hi.hi + hi.lo + lo.hi: what is dropped (lo.lo) is < 2^-22 |x y|.  That is the f32 MFMA's
// accuracy class at 5x its rate, NOT the 2^-16 of a bf16 hi + lo split.  Parity: every saved activation within 1e-4 of the
// oracle (tests/test_mfma_recurrence.py).
//
// Geometry (NO cells, NO % 4 == 0, NO <= 128; NI inputs, NI % 4 == 0, NI <= 64).  M = gate rows in tiles of 16 = 4 cells x 4
// gates (row m = 4 cs + q, q = 0 gi, 1 gf, 2 go, 3 ci), N = 16 lines, K = [cells | inputs | 1] in blocks of 32.  In the MFMA's
// result layout lane l = 16 cs + n then holds the FOUR gates of ONE cell for ONE line (rows 4 (l >> 4) + i, column l & 15):
// forward_statemem / forward_nonlingate (clstm_compute.cc:504-537) are lane-local.  Tiles come in pairs (2p, 2p + 1) holding
// cells 8p + 2cs and 8p + 2cs + 1, so that a lane packs the two h values it produces into one dword of the next step's B
// operand, and the two tiles of a pair are two INDEPENDENT accumulator chains that alternate on the matrix pipe (an MFMA behind
// its own predecessor is forwarded only when nothing sits between them; the epilogue's VALU work is meant to sit there).  Four
// waves, one per SIMD; wave w keeps the hi and lo A fragments of its pairs (and, for an odd tile count, one wave the last tile)
// in registers for the whole sequence: 25 tiles x 5 k-blocks x 2 x 4 registers = 250 KB of the CU's 512 KB file at NO = 100.
//
// Memory.  The result layout spreads a row of G / C / H over 16-byte pieces of 16 different frames per instruction -- hopeless
// for the memory pipeline.  Every global access is therefore row-contiguous (wave w moves lines 4w .. 4w + 3) and LDS does the
// transposition: activations / c / h go through an LDS row image [line][cell ^ line] (16-byte slots, XOR-swizzled:
// conflict-free for the epilogue's per-tile writes and for the row reads going out) that is double-buffered, and leave -- spread
// over the NEXT step's MFMA stream, so that the memory pipeline drains beside the matrix pipe instead of in front of it.  One
// barrier per step.
#pragma once
#include "devintrin.h"
#ifndef CLSTM_HIP_EMU

namespace clstm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int MF_HS = 8;   // h, x and the constant 1 are scaled by 2^MF_HS before the f16 split

struct LstmMfmaArgs {
  const unsigned short* W;   // A fragments [dir][tile][k-block][hi | lo][lane][8 halfs]   (k_pack_mfma)
  const float* inv_scale;    // [dir] 2^-(e + MF_HS)
  const float* X; int ldx;   // the layer's input frames [N][ldx]
  float *G, *C, *H, *S;      // as LstmSeqArgs (G: activations out only)
  const float* dH; float* D;
  const int* line_off;
  const int* order;          // [bs] lines, longest first (a group of 16 consecutive entries shares a workgroup), or null
  int bs, ndir, ldh, hofs, lds, sofs;
  int dbg;                   // experiments (mfma_dbg): 1 no activation-row stores, 2 no c / h / source stores, 4 no input loads
  int store_s;               // 1: deposit h_t in the next frame's source row (else k_source_h rebuilds those columns from H)
  long long sdir;
  long long N;               // frames in the batch
  long long* prof;           // diagnostics build (-DCLSTM_LSTM_PROF): [4 waves][8] summed phase cycles of workgroup (0, 0)
};

template <int NO, int NI>
struct MfmaGeom {
  static_assert(NO % 4 == 0 && NO >= 32 && NO <= 128 && NI % 4 == 0 && NI >= 4 && NI <= 64, "cells / inputs");
  static constexpr int NW = 8;                               // waves: two per SIMD -- one wave's epilogue beside the other's MFMAs
  static constexpr int NT = NO / 4, TW = NT / NW;            // tiles of 4 cells; per wave TW of them in registers ...
  static constexpr int EXTRA = NT - NW * TW;                 // ... and the last wave EXTRA more, fragments in LDS (0 or 1)
  static_assert(TW >= 2 && EXTRA <= 1, "geometry not instantiable");
  static constexpr int NPW = TW / 2, LONE = TW & 1;          // a wave's tiles: NPW pairs, then a lone one
  static constexpr int KT = NO + NI + 1, KB = (KT + 31) / 32, NCH = 4 * KB;   // k = [cells | inputs | 1 | zero pad]
  static constexpr int SLOTS = (NO + 15) / 16 * 16;   // 16-byte slots (one per cell) in a staged row of gate values
  static constexpr int RS = SLOTS * 16, RH = (SLOTS + 63) / 64;
  static constexpr int PART = NCH * 256, HBUF = 2 * PART;   // B image: [buffer][hi | lo][chunk of 8 k][16 lines][8 halfs]
  static constexpr int OUT_OFF = 2 * HBUF, CS_REL = 16 * RS, HS_REL = CS_REL + 16 * NO * 4, OUTSZ = HS_REL + 16 * NO * 4;
  static constexpr int WS_OFF = OUT_OFF + 2 * OUTSZ;        // the extra tile's A fragments [k-block][hi | lo][lane][16 bytes]
  static constexpr int DUMP_OFF = WS_OFF + EXTRA * KB * 2048;   // where lanes without a datum write
  static constexpr int SMEM = DUMP_OFF + 64;
  static constexpr long long W_HALFS_PER_DIR = (long long)NT * KB * 2 * 64 * 8;
};
// cell of (tile, cell slot cs): wave v = tile / TW owns cells 4 TW v .. 4 TW (v + 1) - 1; inside it pairs of tiles interleave
// (tiles 2p, 2p + 1 of the wave hold cells 8p + 2cs, 8p + 2cs + 1), a lone tile and the extra tile hold 4 consecutive cells
__host__ __device__ inline int mfma_cell(int nt, int tile, int cs) {
  const int tw = nt / 8;
  if (tile >= 8 * tw) return 4 * tile + cs;
  const int v = tile / tw, i = tile % tw, npw = tw / 2;
  return 4 * tw * v + (i < 2 * npw ? 8 * (i >> 1) + 2 * cs + (i & 1) : 8 * npw + cs);
}

// ---- packing: find max |[R | W_x | b]| of a direction, pick the scale, write the fragments --------------------------------
// PackDesc (ops.h) as the other packs: W_q(cell, col) = v[p_off[dir][q] + cell + no col], col 0 bias, 1 + j input j,
// 1 + ni + k cell k (tensor.h:263-264); k of the product: [cells | inputs | bias]
struct MfmaPackArgs { const float* v; long long p_off[2][4]; int ni, no, nt, kb; unsigned short* W; float* inv_scale; };
DEVFN int mfma_pack_col(int k, int no, int ni) { return k < no ? 1 + ni + k : (k < no + ni ? 1 + (k - no) : (k == no + ni ? 0 : -1)); }
// (grid (blocks, directions): EVERY block finds the direction's max itself -- 60 K floats out of L2 -- and packs its share of the
//  fragments; as ONE workgroup per direction the kernel took 128 us of every training step, profiles/r06_mfma_busy_mb2048.txt)
__global__ __launch_bounds__(1024) void k_pack_mfma(MfmaPackArgs p) {
  __shared__ float red[16];
  __shared__ int e_sh;
  const int dir = blockIdx.y, tid = threadIdx.x;
  const int no = p.no, ncol = 1 + p.ni + no;
  float mx = 0.0f;
  for (int q = 0; q < 4; q++) {
    const float* vq = p.v + p.p_off[dir][q];
    for (int r = tid; r < no * ncol; r += 1024) {
      const float x = fabsf(vq[r]);
      mx = x > mx ? x : mx;   // (NaN compares false: a non-finite parameter leaves the scale alone and surfaces in the outputs)
    }
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = 0.0f;
    for (int i = 0; i < 16; i++) m = red[i] > m ? red[i] : m;
    int e = 0;
    if (m > 0.0f && f32_finite(m)) { e = 13 - ilogbf(m); e = e > 40 ? 40 : (e < -40 ? -40 : e); }
    e_sh = e;
    if (blockIdx.x == 0) p.inv_scale[dir] = ldexpf(1.0f, -(e + MF_HS));
  }
  __syncthreads();
  const float sc = ldexpf(1.0f, e_sh);
  const long long per_dir = (long long)p.nt * p.kb * 2 * 64 * 8;
  for (long long i = (long long)blockIdx.x * 1024 + tid; i < (long long)p.nt * p.kb * 64 * 8; i += (long long)gridDim.x * 1024) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int kb = (int)((i >> 9) % p.kb), tile = (int)((i >> 9) / p.kb);
    const int m = lane & 15, cs = m >> 2, q = m & 3;
    const int cell = mfma_cell(p.nt, tile, cs);
    const int col = mfma_pack_col(kb * 32 + 8 * (lane >> 4) + j, no, p.ni);
    const float x = (cell < no && col >= 0) ? p.v[p.p_off[dir][q] + cell + (long long)no * col] * sc : 0.0f;
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    unsigned short* dst = p.W + dir * per_dir + ((long long)(tile * p.kb + kb) * 2) * 512 + lane * 8 + j;
    dst[0] = __builtin_bit_cast(unsigned short, hi);
    dst[512] = __builtin_bit_cast(unsigned short, lo);
  }
}

DEVFN f32x4 mfma16x16x32_f16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// LDS rendezvous that leaves global loads / stores in flight (__syncthreads() would wait vmcnt(0)); one asm statement with a
// memory clobber, so that neither LDS stores sink below it nor LDS loads rise above it
DEVFN void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
DEVFN f32x4 buf_load4_s(BufF32 b, unsigned lane_off, unsigned uniform_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, lane_off, uniform_off, 0));
}
// (No store takes an SGPR soffset here.  The ISA manual lists no wait state between a 128-bit buffer store in that form and a
//  write of its data registers, and LLVM inserts none; with two waves per SIMD the overwrite DID reach memory on MI355X --
//  activation rows came back holding the next store's offset register.  scripts/dbg/scan_store_hazard.py scans the assembly.)
// instruction-group hints for the scheduler (IGroupLP): the next `n` instructions of class `mask` in program order
#define MF_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_VALU = 0x2, SG_MFMA = 0x8, SG_VMEM_W = 0x40, SG_DS_R = 0x100, SG_DS_W = 0x200;

template <int NO, int NI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_fwd_mfma_kernel(LstmMfmaArgs a) {
  using Gm = MfmaGeom<NO, NI>;
  constexpr int TW = Gm::TW, NPW = Gm::NPW, LONE = Gm::LONE, EXTRA = Gm::EXTRA, KB = Gm::KB, RH = Gm::RH, RS = Gm::RS, NW = Gm::NW;
  char* const smem = dyn_smem<char>();
  const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
  const int n = lane & 15, cl = lane >> 4;
  const int dir = blockIdx.y, grp = blockIdx.x;
  const int nd = a.ndir;

  // this lane's line (column n of the product) and, as scalars, the two lines whose rows this wave moves
  const int gl = grp * 16 + n;
  const int b = gl < a.bs ? (a.order ? a.order[gl] : gl) : -1;
  const int off = b >= 0 ? a.line_off[b] : 0;
  const int T = b >= 0 ? a.line_off[b + 1] - off : 0;
  int tmx = T;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) { const int o = __shfl_xor(tmx, m, 64); tmx = o > tmx ? o : tmx; }
  const int Tmax = wave_uniform(tmx);
  int offj[2], Tj[2];
#pragma unroll
  for (int j = 0; j < 2; j++) { offj[j] = __builtin_amdgcn_readlane(off, 2 * w + j); Tj[j] = __builtin_amdgcn_readlane(T, 2 * w + j); }
  if (Tmax <= 0) return;

  // A fragments (hi, lo) of this wave's TW tiles, resident for the whole sequence.  (The extra tile's stay in LDS and are read
  // every step by the last wave: with them it would need (TW + 1) x KB x 8 operand registers.)
  f16x8 Wh[TW][KB], Wl[TW][KB];
#pragma unroll
  for (int i = 0; i < TW; i++)
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      const unsigned short* wp = a.W + dir * Gm::W_HALFS_PER_DIR + ((long long)((w * TW + i) * KB + kb) * 2) * 512 + lane * 8;
      Wh[i][kb] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wp));
      Wl[i][kb] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wp + 512));
    }
  // (consumed here once, so that the waits for these loads sit in front of the loop and not -- as s_waitcnt vmcnt(N) with a small
  //  N -- at the top of every step: lstm_mfma_bwd.h)
#pragma unroll
  for (int i = 0; i < TW; i++)
#pragma unroll
    for (int kb = 0; kb < KB; kb++) { asm volatile("" : "+a"(Wh[i][kb])); asm volatile("" : "+a"(Wl[i][kb])); }
  const float inv = a.inv_scale[dir];
  const float sig_k = inv * ACT_SIG_SCALE, tanh_k = inv * ACT_TANH_SCALE;   // (powers of two times a constant: exact products)
  // B image: h_{-1} = 0, zero padding, and the constant 1 behind the inputs (both buffers)
  for (int i = tid * 16; i < 2 * Gm::HBUF; i += 512 * 16) *reinterpret_cast<u32x4*>(smem + i) = (u32x4){0u, 0u, 0u, 0u};
  if (EXTRA)
    for (int i = tid; i < KB * 128; i += 512)
      *reinterpret_cast<u32x4*>(smem + Gm::WS_OFF + i * 16) =
          *reinterpret_cast<const u32x4*>(a.W + dir * Gm::W_HALFS_PER_DIR + (long long)(NW * TW) * KB * 1024 + i * 8);
  __syncthreads();
  if (tid < 32) {
    constexpr int kone = NO + NI;
    const _Float16 one = (_Float16)(float)(1 << MF_HS);
    *reinterpret_cast<_Float16*>(smem + (tid >> 4) * Gm::HBUF + ((kone >> 3) * 16 + (tid & 15)) * 16 + (kone & 7) * 2) = one;
  }

  // ---- global side: whole rows, wave w owns lines 2w, 2w + 1 ----
  const unsigned gstr = (unsigned)nd * 4 * NO * 4, cstr = (unsigned)nd * NO * 4, hstr = (unsigned)a.ldh * 4, sstr = (unsigned)a.lds * 4;
  const unsigned xstr = (unsigned)a.ldx * 4;
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * gstr);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * cstr);
  const BufF32 hbuf = make_buf(a.H, (size_t)a.N * hstr);
  const BufF32 sbuf = make_buf(a.S + (size_t)dir * a.sdir, (size_t)a.N * sstr);
  const BufF32 xbuf = make_buf(a.X, (size_t)a.N * xstr);
  const unsigned dbg_g = (a.dbg & 1) ? 0x80000000u : 0u, dbg_ch = (a.dbg & 2) ? 0x80000000u : 0u, dbg_x = (a.dbg & 4) ? 0x80000000u : 0u;
  unsigned gvo[RH];   // gate rows: lane = cell (16 bytes) within the half row
#pragma unroll
  for (int hh = 0; hh < RH; hh++) gvo[hh] = 64 * hh + lane < NO ? (unsigned)(64 * hh + lane) * 16u : BUF_OOB_BASE;
  // scalar row state: frame of step t of line j (Reversed = index arithmetic, clstm.cc:458-478)
  auto row_valid = [&](int j, int t) { return (unsigned)t < (unsigned)Tj[j]; };   // (t = -1: no)
  auto row_tok = [&](int j, int t) { return offj[j] + (dir == 0 ? t : Tj[j] - 1 - t); };
  auto oob_if = [&](bool ok) -> unsigned { return ok ? 0u : 0x80000000u; };
  // (validity as arithmetic, never as control flow: a branch inside a step ends the scheduling region the MFMA stream and the
  //  work beside it are interleaved in)
  auto oob_lane = [&](int t, int Tl) -> unsigned { return ~(unsigned)((t - Tl) >> 31) & 0x80000000u; };   // 0 while t < Tl
  // c / h rows leave as one instruction for both lines (lanes 0-31 / 32-63, 16 bytes = 4 cells per lane)
  const int l5 = lane & 31;
  const int rrow = 2 * w + (lane >> 5);
  const int offr = __shfl(off, rrow, 64), Tr = __shfl(T, rrow, 64);
  const unsigned chl = l5 < NO / 4 ? (unsigned)l5 * 16u : BUF_OOB_BASE;
  const unsigned s_off_mask = a.store_s ? 0u : 0x80000000u;
  // input frames: lane = (line 2w + (lane >> 4), float4 lane & 15 of its frame), lanes 0-31
  const int xq = lane & 15, xrow = 2 * w + ((lane >> 4) & 1);
  const bool xon = lane < 32 && xq < NI / 4;
  const int offx = __shfl(off, xrow, 64), Tx = __shfl(T, xrow, 64);
  const unsigned xvl = xon ? (unsigned)xq * 16u : BUF_OOB_BASE;
  auto load_x = [&](int t) -> f32x4 {
    const unsigned tok = (unsigned)(offx + (dir == 0 ? t : Tx - 1 - t));
    return buf_load4(xbuf, (tok * xstr + xvl) | oob_lane(t, Tx) | dbg_x);
  };
  // k = NO + 4 xq .. + 3 of line xrow (lanes without an input: the dump slot)
  const unsigned xwo_h = (unsigned)((((NO + 4 * xq) >> 3) * 16 + xrow) * 16 + ((NO + 4 * xq) & 7) * 2);
  auto put_x = [&](char* img, f32x4 x) {   // two f16 terms of 2^8 x into the B image
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float xs = x[e] * (float)(1 << MF_HS);
      hi[e] = (_Float16)xs;
      lo[e] = (_Float16)(xs - (float)hi[e]);
    }
    char* const dst = xon ? img + xwo_h : smem + Gm::DUMP_OFF;
    *reinterpret_cast<f16x4*>(dst) = hi;
    *reinterpret_cast<f16x4*>(dst + (xon ? Gm::PART : 8)) = lo;
  };

  // ---- LDS side ----
  // epilogue, lane (cl, n), tile i of the wave (TW: the extra tile): cell c -> slot (c ^ n) of row n in the activation image,
  // k = c in the B image (chunk c >> 3, element c & 7)
  unsigned gxo[TW + 1], hbo[TW + 1];
  int cellv[TW + 1];
#pragma unroll
  for (int i = 0; i <= TW; i++) {
    const int c = mfma_cell(Gm::NT, i < TW ? w * TW + i : NW * TW, cl);
    cellv[i] = c;
    gxo[i] = (unsigned)(n * RS + ((c ^ n) << 4));
    hbo[i] = (unsigned)(((c >> 3) * 16 + n) * 16 + (c & 7) * 2);
  }
  const unsigned cho = (unsigned)(n * NO * 4);             // + 4 c
  const unsigned bfo = (unsigned)(cl * 256 + n * 16);      // B fragments: + 1024 kb (+ PART for lo)

  // outputs of a step leave as rows during the next one: activations (G), c, h (H and, shifted by one frame, the source rows S);
  // each portion reads its rows from the LDS image right in front of its stores
  auto store_g = [&](const char* out, const int tp, const int j) {   // the activation row of line 2w + j
    const int row = 2 * w + j;
    f32x4 av[RH];
#pragma unroll
    for (int hh = 0; hh < RH; hh++)
      av[hh] = *reinterpret_cast<const f32x4*>(out + row * RS + (((64 * hh + lane < Gm::SLOTS ? 64 * hh + lane : 0) ^ row) << 4));
    const bool ok = row_valid(j, tp);
    const unsigned so = (unsigned)row_tok(j, tp) * gstr + (unsigned)dir * NO * 16u;   // (an invalid row: every lane out of range)
#pragma unroll
    for (int hh = 0; hh < RH; hh++) buf_store4(gbuf, (gvo[hh] | oob_if(ok) | dbg_g) + so, av[hh]);
  };
  auto store_ch = [&](const char* out, const int tp) {  // c, h of lines 2w, 2w + 1
    const int lo = l5 < NO / 4 ? l5 : 0;
    const f32x4 cv = *reinterpret_cast<const f32x4*>(out + Gm::CS_REL + rrow * NO * 4 + lo * 16);
    const f32x4 hv = *reinterpret_cast<const f32x4*>(out + Gm::HS_REL + rrow * NO * 4 + lo * 16);
    const unsigned bad = oob_lane(tp, Tr) | (unsigned)(tp >> 31 & 0x80000000) | dbg_ch;
    const unsigned tok = (unsigned)(offr + (dir == 0 ? tp : Tr - 1 - tp));
    buf_store4(cbuf, (tok * cstr + (unsigned)dir * NO * 4u + chl) | bad, cv);
    buf_store4(hbuf, (tok * hstr + (unsigned)(a.hofs + dir * NO) * 4u + chl) | bad, hv);
    // h_t is the recurrent part of the NEXT step's source row (forward_stack_delay, clstm_compute.cc:377-397)
    const unsigned bads = oob_lane(tp + 1, Tr) | (unsigned)(tp >> 31 & 0x80000000) | s_off_mask | dbg_ch;
    const unsigned toks = (unsigned)(offr + (dir == 0 ? tp + 1 : Tr - 2 - tp));
    buf_store4(sbuf, (toks * sstr + (unsigned)a.sofs * 4u + chl) | bads, hv);
  };
  // h_{-1} = 0 in the first source row of every line (forward_stack_delay with last < 0)
  {
    const unsigned tok = (unsigned)(offr + (dir == 0 ? 0 : Tr - 1));
    buf_store4(sbuf, a.store_s && Tr > 0 ? tok * sstr + (unsigned)a.sofs * 4u + chl : BUF_OOB, (f32x4){0.f, 0.f, 0.f, 0.f});
  }
  // x_0 straight into buffer 0; x_1, x_2 in flight
  f32x4 xr[2];
  put_x(smem, load_x(0));
  xr[1] = load_x(1);
  xr[0] = load_x(2);
  float cprev[TW + 1];
#pragma unroll
  for (int i = 0; i <= TW; i++) cprev[i] = 0.0f;

#ifdef CLSTM_LSTM_PROF
  long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long pt = 0;
#define MF_STAMP(k) do { long long now_; SCHED_FENCE(); \
                         asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
                         SCHED_FENCE(); pacc[k] += now_ - pt; pt = now_; } while (0)
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(pt) :: "memory");
#else
#define MF_STAMP(k) do {} while (0)
#endif
  auto step = [&](const int t, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    const char* const hr = smem + PAR * Gm::HBUF;                       // [h_{t-1} ; x_t ; 1]
    char* const hw = smem + (PAR ^ 1) * Gm::HBUF;                       // [h_t ; x_{t+1} ; 1]
    char* const outw = smem + Gm::OUT_OFF + PAR * Gm::OUTSZ;            // this step's output rows
    const char* const outr = smem + Gm::OUT_OFF + (PAR ^ 1) * Gm::OUTSZ;   // the previous step's
    lds_barrier();
    MF_STAMP(0);
    // pin the fragments in the accumulation half of the register file (MFMA operands may live there; left alone, hipcc keeps
    // them in VGPRs, runs out, and shuttles them through v_accvgpr_read in front of every MFMA)
#pragma unroll
    for (int i = 0; i < TW; i++)
#pragma unroll
      for (int kb = 0; kb < KB; kb++) { asm volatile("" : "+a"(Wh[i][kb])); asm volatile("" : "+a"(Wl[i][kb])); }
    f16x8 Bh[KB], Bl[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      Bh[kb] = *reinterpret_cast<const f16x8*>(hr + bfo + 1024 * kb);
      Bl[kb] = *reinterpret_cast<const f16x8*>(hr + Gm::PART + bfo + 1024 * kb);
    }
    put_x(hw, xr[PAR ^ 1]);                // x_{t+1}, requested two steps ago
    xr[PAR ^ 1] = load_x(t + 3);
    // Two INDEPENDENT accumulator chains alternate on the matrix pipe (a pair of tiles, or the even / odd k-blocks of one): an
    // MFMA behind its own predecessor is forwarded only back to back, and the epilogue's VALU work is meant to sit between them
    auto mm2 = [&](const int i0, f32x4& a0, f32x4& a1) {
      a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0;
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        a0 = mfma16x16x32_f16(Wl[i0][kb], Bh[kb], a0); a1 = mfma16x16x32_f16(Wl[i0 + 1][kb], Bh[kb], a1);
        a0 = mfma16x16x32_f16(Wh[i0][kb], Bl[kb], a0); a1 = mfma16x16x32_f16(Wh[i0 + 1][kb], Bl[kb], a1);
        a0 = mfma16x16x32_f16(Wh[i0][kb], Bh[kb], a0); a1 = mfma16x16x32_f16(Wh[i0 + 1][kb], Bh[kb], a1);
      }
    };
    auto mm1 = [&](const f16x8 (&wh)[KB], const f16x8 (&wl)[KB]) -> f32x4 {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int kb = 0; kb < KB; kb += 2) {
        a0 = mfma16x16x32_f16(wl[kb], Bh[kb], a0); if (kb + 1 < KB) a1 = mfma16x16x32_f16(wl[kb + 1], Bh[kb + 1], a1);
        a0 = mfma16x16x32_f16(wh[kb], Bl[kb], a0); if (kb + 1 < KB) a1 = mfma16x16x32_f16(wh[kb + 1], Bl[kb + 1], a1);
        a0 = mfma16x16x32_f16(wh[kb], Bh[kb], a0); if (kb + 1 < KB) a1 = mfma16x16x32_f16(wh[kb + 1], Bh[kb + 1], a1);
      }
      return a0 + a1;
    };
    // forward_nonlin0 x 4 (clstm_compute.cc:195-229), forward_statemem (:504-508), forward_nonlingate (:530-537) of one tile
    auto epi = [&](const int i, const f32x4 acc) -> float {
      const float gi = fast_rcp(1.0f + fast_exp2(acc[0] * sig_k));
      const float gf = fast_rcp(1.0f + fast_exp2(acc[1] * sig_k));
      const float go = fast_rcp(1.0f + fast_exp2(acc[2] * sig_k));
      const float ci = fmaf(fast_rcp(1.0f + fast_exp2(acc[3] * tanh_k)), 2.0f, -1.0f);
      const float c = fmaf(gf, cprev[i], ci * gi);
      const float h = go * tanh_fast(c);
      cprev[i] = c;
      *reinterpret_cast<f32x4*>(outw + gxo[i]) = (f32x4){gi, gf, go, ci};
      *reinterpret_cast<float*>(outw + Gm::CS_REL + cho + 4 * cellv[i]) = c;
      *reinterpret_cast<float*>(outw + Gm::HS_REL + cho + 4 * cellv[i]) = h;
      return h * (float)(1 << MF_HS);   // the next step's B operand: two f16 terms of 2^8 h
    };
    auto put_h2 = [&](const int i0, const float he, const float ho) {   // cells c, c + 1 of a pair: one dword per term
      f16x2 hi2, lo2;
      hi2[0] = (_Float16)he; hi2[1] = (_Float16)ho;
      lo2[0] = (_Float16)(he - (float)hi2[0]); lo2[1] = (_Float16)(ho - (float)hi2[1]);
      *reinterpret_cast<f16x2*>(hw + hbo[i0]) = hi2;
      *reinterpret_cast<f16x2*>(hw + Gm::PART + hbo[i0]) = lo2;
    };
    auto put_h1 = [&](const int i, const float hs) {
      const _Float16 hi1 = (_Float16)hs;
      *reinterpret_cast<_Float16*>(hw + hbo[i]) = hi1;
      *reinterpret_cast<_Float16*>(hw + Gm::PART + hbo[i]) = (_Float16)(hs - (float)hi1);
    };
    // the previous step's rows go out in two portions beside the later MFMA phases
    f32x4 A0, A1, N0 = {0.f, 0.f, 0.f, 0.f}, N1 = N0;
    mm2(0, A0, A1);
    MF_STAMP(1);
#pragma unroll
    for (int u = 0; u < NPW; u++) {
      if (u + 1 < NPW) mm2(2 * u + 2, N0, N1);
      else if (LONE) N0 = mm1(Wh[TW - 1], Wl[TW - 1]);
      const float he = epi(2 * u, A0), ho = epi(2 * u + 1, A1);
      put_h2(2 * u, he, ho);
      if (u == 0) { store_g(outr, t - 1, 0); if (NPW > 1 || !LONE) store_ch(outr, t - 1); }
      if (u == NPW - 1 && !LONE) store_g(outr, t - 1, 1);
      if (u + 1 < NPW) {
#pragma unroll
        for (int k = 0; k < 6 * KB; k++) { MF_SGB(SG_MFMA, 1); MF_SGB(SG_VALU, 3); }
      } else if (LONE) {
#pragma unroll
        for (int k = 0; k < 3 * KB; k++) { MF_SGB(SG_MFMA, 1); MF_SGB(SG_VALU, 5); }
      }
      A0 = N0; A1 = N1;
      MF_STAMP(2 + (u < 2 ? u : 2));
    }
    if (LONE) {
      if (NPW == 1) store_ch(outr, t - 1);
      store_g(outr, t - 1, 1);
      if (EXTRA && w == NW - 1) {   // the last wave: the extra tile, fragments from LDS
        f16x8 sh[KB], sl[KB];
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
          sh[kb] = *reinterpret_cast<const f16x8*>(smem + Gm::WS_OFF + kb * 2048 + lane * 16);
          sl[kb] = *reinterpret_cast<const f16x8*>(smem + Gm::WS_OFF + kb * 2048 + 1024 + lane * 16);
        }
        const f32x4 X0 = mm1(sh, sl);
        put_h1(TW - 1, epi(TW - 1, A0));
#pragma unroll
        for (int k = 0; k < 3 * KB; k++) { MF_SGB(SG_MFMA, 1); MF_SGB(SG_VALU, 3); }
        put_h1(TW, epi(TW, X0));
      } else put_h1(TW - 1, epi(TW - 1, A0));
    } else if (EXTRA && w == NW - 1) {
      f16x8 sh[KB], sl[KB];
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        sh[kb] = *reinterpret_cast<const f16x8*>(smem + Gm::WS_OFF + kb * 2048 + lane * 16);
        sl[kb] = *reinterpret_cast<const f16x8*>(smem + Gm::WS_OFF + kb * 2048 + 1024 + lane * 16);
      }
      put_h1(TW, epi(TW, mm1(sh, sl)));
    }
    MF_STAMP(5);
  };
  int t = 0;
  for (; t + 1 < Tmax; t += 2) {
    step(t, std::integral_constant<int, 0>{});
    step(t + 1, std::integral_constant<int, 1>{});
  }
  if (t < Tmax) { step(t, std::integral_constant<int, 0>{}); t++; }
  lds_barrier();
  {
    const char* const out = smem + Gm::OUT_OFF + ((t - 1) & 1) * Gm::OUTSZ;
    store_g(out, t - 1, 0);
    store_g(out, t - 1, 1);
    store_ch(out, t - 1);
  }
#ifdef CLSTM_LSTM_PROF
  if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
    for (int k = 0; k < 8; k++) a.prof[w * 8 + k] = pacc[k];
#endif
}

}  // namespace clstm
#endif  // CLSTM_HIP_EMU
