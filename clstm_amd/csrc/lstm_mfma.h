// lstm_mfma.h -- the NARROW layer's recurrence batched over lines on the matrix cores (minibatches that fill the chip).
//
// lstm_seq.h walks one line per workgroup: R.h_{t-1} is 100 x 100 x 4 packed f32 FMAs per line and step, the MFMA pipes idle,
// and once every CU holds two such workgroups (256 lines) the kernel is VALU-bound (27 % of the f32 vector peak).  Here ONE
// workgroup owns 16 lines x one direction and the per-step product of GenericNPLSTM::forward (clstm.cc:612-620: four
// forward_full1 = forward_lin1 + forward_nonlin0, clstm_compute.cc:275-314) is a real GEMM
//     pre[4 no x 16 lines] = R[4 no x no] . h_{t-1}[no x 16 lines]  (+ the hoisted W_x.x_t + b, gemm_mfma.h)
// on v_mfma_f32_16x16x32_f16 with f32 accumulation.
//
// Arithmetic.  Both operands are f32 values split into TWO f16 terms, x = hi + lo with hi = f16(x), lo = f16(x - hi) (the
// difference is exact), after a power-of-two scaling that keeps lo out of the f16 subnormal range (R by 2^e with max |R| 2^e in
// [2^13, 2^14), h in [-1, 1] by 2^8; the inverse scale rides the fma that adds the input part).  f16 carries 11 significant
// bits, so hi + lo represents x to 2^-22 |x| or better (f32 itself: 2^-24), each f16 x f16 product is exact in the f32
// accumulator, and a product is hi.hi + hi.lo + lo.hi: what is dropped (lo.lo) is < 2^-22 |x y|.  That is the f32 MFMA's
// accuracy class at 5x its rate, NOT the 2^-16 of a bf16 hi + lo split.  Parity: every saved activation within 1e-4 of the
// oracle (tests/test_mfma_recurrence.py).
//
// Geometry (NO cells, NO % 4 == 0, NO <= 128).  M = gate rows in tiles of 16 = 4 cells x 4 gates (row m = 4 cs + q, q = 0 gi,
// 1 gf, 2 go, 3 ci), N = 16 lines, K = cells in blocks of 32.  In the MFMA's result layout lane l = 16 cs + n then holds the
// FOUR gates of ONE cell for ONE line (rows 4 (l >> 4) + i, column l & 15): forward_statemem / forward_nonlingate
// (clstm_compute.cc:504-537) are lane-local.  Tiles come in pairs (2p, 2p + 1) holding cells 8p + 2cs and 8p + 2cs + 1, so that
// a lane packs the two h values it produces into one dword of the next step's B operand.  Four waves, one per SIMD; wave w keeps
// the hi and lo A fragments of its pairs (and, for an odd tile count, wave 0 the last tile) in registers for the whole
// sequence: 25 tiles x 4 k-blocks x 2 x 4 registers = 200 KB of the CU's 512 KB file at NO = 100.
//
// Memory.  The result layout spreads a row of G / C / H over 16-byte pieces of 16 different frames per instruction -- hopeless
// for the memory pipeline.  Every global access is therefore row-contiguous (wave w moves lines 4w .. 4w + 3) and LDS does the
// transposition: pre-activations are fetched two steps ahead into registers, written to an LDS row image [line][cell ^ line]
// (16-byte slots, XOR-swizzled: conflict-free for the row writes, for the per-tile reads of the epilogue and for both sides of
// the activation image going out), activations / c / h go through the same kind of image and leave at the top of the NEXT step.
// Two barriers per step: B1 (h_t, the output images and the free input image), B2 (the input image, the free output images).
#pragma once
#include "devintrin.h"
#ifndef CLSTM_HIP_EMU

namespace clstm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int MF_HS = 8;   // h is scaled by 2^MF_HS before the f16 split

struct LstmMfmaArgs {
  const unsigned short* W;   // A fragments [dir][tile][k-block][hi | lo][lane][8 halfs]   (k_pack_mfma)
  const float* inv_scale;    // [dir] 2^-(e + MF_HS)
  float *G, *C, *H, *S;      // as LstmSeqArgs
  const float* dH; float* D;
  const int* line_off;
  const int* order;          // [bs] lines, longest first (a group of 16 consecutive entries shares a workgroup), or null
  int bs, ndir, ldh, hofs, lds, sofs;
  long long sdir;
  long long N;               // frames in the batch
  long long* prof;           // diagnostics build (-DCLSTM_LSTM_PROF): [4 waves][8] summed phase cycles of workgroup (0, 0)
};

template <int NO>
struct MfmaGeom {
  static_assert(NO % 4 == 0 && NO >= 16 && NO <= 128, "cells");
  static constexpr int NT = NO / 4, FP = NT / 2, SINGLE = NT & 1, PPW = (FP + 3) / 4, KB = (NO + 31) / 32, NCH = 4 * KB;
  static constexpr int TPW = 2 * PPW + SINGLE;
  static constexpr int SW = (FP % PPW == 0 && FP / PPW == 4) ? 0 : 3;   // the wave that takes the unpaired tile
  static constexpr int SLOTS = (NO + 15) / 16 * 16;   // 16-byte slots (one per cell) in a staged row of gate values
  static constexpr int RS = SLOTS * 16, RH = (SLOTS + 63) / 64;
  static constexpr int PART = NCH * 256, HBUF = 2 * PART;   // h image: [buffer][hi | lo][chunk of 8 cells][16 lines][8 halfs]
  static constexpr int GXS_OFF = 2 * HBUF, ACT_OFF = GXS_OFF + 16 * RS, CS_OFF = ACT_OFF + 16 * RS, HS_OFF = CS_OFF + 16 * NO * 4;
  static constexpr int SMEM = HS_OFF + 16 * NO * 4;
  static constexpr long long W_HALFS_PER_DIR = (long long)NT * KB * 2 * 64 * 8;
};

// ---- packing: one workgroup per direction finds max |R|, picks the scale and writes the fragments -------------------------
// PackDesc (ops.h) as the other packs: R_q(cell, k) = v[p_off[dir][q] + cell + no (1 + ni + k)]   (tensor.h:263-264)
struct MfmaPackArgs { const float* v; long long p_off[2][4]; int ni, no, nt, kb; unsigned short* W; float* inv_scale; };
__global__ __launch_bounds__(1024) void k_pack_mfma(MfmaPackArgs p) {
  __shared__ float red[16];
  __shared__ int e_sh;
  const int dir = blockIdx.x, tid = threadIdx.x;
  const int no = p.no;
  float mx = 0.0f;
  for (int i = tid; i < 4 * no * no; i += 1024) {
    const int q = i / (no * no), r = i % (no * no);
    const float x = fabsf(p.v[p.p_off[dir][q] + (r % no) + (long long)no * (1 + p.ni + r / no)]);
    mx = x > mx ? x : mx;   // (NaN compares false: a non-finite parameter leaves the scale alone and surfaces in the outputs)
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
    p.inv_scale[dir] = ldexpf(1.0f, -(e + MF_HS));
  }
  __syncthreads();
  const float sc = ldexpf(1.0f, e_sh);
  const int FP = p.nt / 2;
  const long long per_dir = (long long)p.nt * p.kb * 2 * 64 * 8;
  for (long long i = tid; i < (long long)p.nt * p.kb * 64 * 8; i += 1024) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int kb = (int)((i >> 9) % p.kb), tile = (int)((i >> 9) / p.kb);
    const int m = lane & 15, cs = m >> 2, q = m & 3;
    const int pr = tile >> 1, r = tile & 1;
    const int cell = pr < FP ? 8 * pr + 2 * cs + r : 8 * FP + cs;
    const int k = kb * 32 + 8 * (lane >> 4) + j;
    const float x = (cell < no && k < no) ? p.v[p.p_off[dir][q] + cell + (long long)no * (1 + p.ni + k)] * sc : 0.0f;
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
DEVFN void buf_store4_s(BufF32 b, unsigned lane_off, unsigned uniform_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), b.r, lane_off, uniform_off, 0);
}

template <int NO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_mfma_kernel(LstmMfmaArgs a) {
  using Gm = MfmaGeom<NO>;
  constexpr int PPW = Gm::PPW, KB = Gm::KB, TPW = Gm::TPW, RH = Gm::RH, RS = Gm::RS;
  char* const smem = dyn_smem<char>();
  const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
  const int n = lane & 15, cl = lane >> 4;
  const int dir = blockIdx.y, grp = blockIdx.x;
  const int nd = a.ndir;

  // this lane's line (column n of the product) and, as scalars, the four lines whose rows this wave moves
  const int gl = grp * 16 + n;
  const int b = gl < a.bs ? (a.order ? a.order[gl] : gl) : -1;
  const int off = b >= 0 ? a.line_off[b] : 0;
  const int T = b >= 0 ? a.line_off[b + 1] - off : 0;
  int tmx = T;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) { const int o = __shfl_xor(tmx, m, 64); tmx = o > tmx ? o : tmx; }
  const int Tmax = wave_uniform(tmx);
  int offj[4], Tj[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { offj[j] = __builtin_amdgcn_readlane(off, 4 * w + j); Tj[j] = __builtin_amdgcn_readlane(T, 4 * w + j); }
  if (Tmax <= 0) return;

  // A fragments (hi, lo) of this wave's tiles, resident for the whole sequence
  f16x8 Wh[TPW][KB], Wl[TPW][KB];
#pragma unroll
  for (int i = 0; i < TPW; i++) {
    const int pr = w * PPW + (i >> 1);
    const bool act = i < 2 * PPW ? pr < Gm::FP : w == Gm::SW;
    const int tile = i < 2 * PPW ? 2 * pr + (i & 1) : 2 * Gm::FP;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      const unsigned short* wp = a.W + dir * Gm::W_HALFS_PER_DIR + ((long long)(tile * KB + kb) * 2) * 512 + lane * 8;
      const u32x4 z = {0u, 0u, 0u, 0u};
      Wh[i][kb] = __builtin_bit_cast(f16x8, act ? *reinterpret_cast<const u32x4*>(wp) : z);
      Wl[i][kb] = __builtin_bit_cast(f16x8, act ? *reinterpret_cast<const u32x4*>(wp + 512) : z);
    }
  }
  const float inv = a.inv_scale[dir];
  for (int i = tid * 16; i < 2 * Gm::HBUF; i += 256 * 16) *reinterpret_cast<u32x4*>(smem + i) = (u32x4){0u, 0u, 0u, 0u};   // h_{-1} = 0

  // ---- global side: whole rows, wave w owns lines 4w .. 4w+3 ----
  const unsigned gstr = (unsigned)nd * 4 * NO * 4, cstr = (unsigned)nd * NO * 4, hstr = (unsigned)a.ldh * 4, sstr = (unsigned)a.lds * 4;
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * gstr);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * cstr);
  const BufF32 hbuf = make_buf(a.H, (size_t)a.N * hstr);
  const BufF32 sbuf = make_buf(a.S + (size_t)dir * a.sdir, (size_t)a.N * sstr);
  unsigned gvo[RH];   // gate rows: lane = cell (16 bytes) within the half row
#pragma unroll
  for (int hh = 0; hh < RH; hh++) gvo[hh] = 64 * hh + lane < NO ? (unsigned)(64 * hh + lane) * 16u : BUF_OOB_BASE;
  // scalar row state: frame of step t of line j (Reversed = index arithmetic, clstm.cc:458-478)
  auto row_valid = [&](int j, int t) { return t < Tj[j]; };
  auto row_tok = [&](int j, int t) { return offj[j] + (dir == 0 ? t : Tj[j] - 1 - t); };
  auto oob_if = [&](bool ok) -> unsigned { return ok ? 0u : 0x80000000u; };
  // c / h rows leave two lines per instruction (lanes 0-31 / 32-63, 16 bytes = 4 cells per lane)
  const int l5 = lane & 31;
  int offr[2], Tr[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = 4 * w + 2 * i + (lane >> 5);
    offr[i] = __shfl(off, row, 64);
    Tr[i] = __shfl(T, row, 64);
  }
  const unsigned chl = l5 < NO / 4 ? (unsigned)l5 * 16u : BUF_OOB_BASE;

  // ---- LDS side ----
  // epilogue, lane (cl, n), tile i: cell c -> slot (c ^ n) of row n in the gate images
  unsigned gxo[TPW];
  int cellv[TPW];
#pragma unroll
  for (int i = 0; i < TPW; i++) {
    const int pr = w * PPW + (i >> 1);
    const int c = i < 2 * PPW ? 8 * pr + 2 * cl + (i & 1) : 8 * Gm::FP + cl;
    cellv[i] = c;
    gxo[i] = (unsigned)(n * RS + ((c ^ n) << 4));
  }
  const unsigned cho = (unsigned)(n * NO * 4);             // + 4 c
  const unsigned hwo = (unsigned)(n * 16 + cl * 4);        // pairs: + 256 p (+ PART for lo); the unpaired tile: n * 16 + cl * 2
  const unsigned bfo = (unsigned)(cl * 256 + n * 16);      // B fragments: + 1024 kb (+ PART for lo)

  // input pre-activations: rows of step t + 2 are requested at the top of step t
  f32x4 gset[2][4 * RH];
  auto load_gx = [&](int t, f32x4 (&gs)[4 * RH]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool ok = row_valid(j, t);
      const unsigned so = ok ? (unsigned)row_tok(j, t) * gstr + (unsigned)dir * NO * 16u : 0u;
#pragma unroll
      for (int hh = 0; hh < RH; hh++) gs[j * RH + hh] = buf_load4_s(gbuf, gvo[hh] | oob_if(ok), so);
    }
  };
  auto stage_gx = [&](const f32x4 (&gs)[4 * RH]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int row = 4 * w + j;
#pragma unroll
      for (int hh = 0; hh < RH; hh++)
        if (64 * hh + lane < Gm::SLOTS)
          *reinterpret_cast<f32x4*>(smem + Gm::GXS_OFF + row * RS + (((64 * hh + lane) ^ row) << 4)) = gs[j * RH + hh];
    }
  };
  // outputs of step tp leave as rows: activations (G), c, h (H and, shifted by one frame, the source rows S)
  auto store_rows = [&](int tp) {
    f32x4 av[4 * RH], cv[2], hv[2];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int row = 4 * w + j;
#pragma unroll
      for (int hh = 0; hh < RH; hh++)
        av[j * RH + hh] = *reinterpret_cast<const f32x4*>(smem + Gm::ACT_OFF + row * RS + (((64 * hh + lane < Gm::SLOTS ? 64 * hh + lane : 0) ^ row) << 4));
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = 4 * w + 2 * i + (lane >> 5);
      const int lo = l5 < NO / 4 ? l5 : 0;
      cv[i] = *reinterpret_cast<const f32x4*>(smem + Gm::CS_OFF + row * NO * 4 + lo * 16);
      hv[i] = *reinterpret_cast<const f32x4*>(smem + Gm::HS_OFF + row * NO * 4 + lo * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool ok = row_valid(j, tp);
      const unsigned so = ok ? (unsigned)row_tok(j, tp) * gstr + (unsigned)dir * NO * 16u : 0u;
#pragma unroll
      for (int hh = 0; hh < RH; hh++) buf_store4_s(gbuf, gvo[hh] | oob_if(ok), so, av[j * RH + hh]);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bool ok = tp < Tr[i];
      const unsigned tok = (unsigned)(offr[i] + (dir == 0 ? tp : Tr[i] - 1 - tp));
      buf_store4(cbuf, ok ? tok * cstr + (unsigned)dir * NO * 4u + chl : BUF_OOB, cv[i]);
      buf_store4(hbuf, ok ? tok * hstr + (unsigned)(a.hofs + dir * NO) * 4u + chl : BUF_OOB, hv[i]);
      // h_t is the recurrent part of the NEXT step's source row (forward_stack_delay, clstm_compute.cc:377-397)
      const bool oks = tp + 1 < Tr[i];
      const unsigned toks = (unsigned)(offr[i] + (dir == 0 ? tp + 1 : Tr[i] - 2 - tp));
      buf_store4(sbuf, oks ? toks * sstr + (unsigned)a.sofs * 4u + chl : BUF_OOB, hv[i]);
    }
  };
  // h_{-1} = 0 in the first source row of every line (forward_stack_delay with last < 0)
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const unsigned tok = (unsigned)(offr[i] + (dir == 0 ? 0 : Tr[i] - 1));
    buf_store4(sbuf, Tr[i] > 0 ? tok * sstr + (unsigned)a.sofs * 4u + chl : BUF_OOB, (f32x4){0.f, 0.f, 0.f, 0.f});
  }
  load_gx(0, gset[0]);
  load_gx(1, gset[1]);
  float cprev[TPW];
#pragma unroll
  for (int i = 0; i < TPW; i++) cprev[i] = 0.0f;

  // (wave-uniform; constant for the pairs of the instantiated sizes: 4 PPW == FP)
  auto tile_on = [&](const int i) -> bool {
    if (i >= 2 * PPW) return w == Gm::SW;
    if constexpr (Gm::FP == 4 * PPW) return true;
    return w * PPW + (i >> 1) < Gm::FP;
  };
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
    const char* const hr = smem + PAR * Gm::HBUF;          // h_{t-1}
    char* const hw = smem + (PAR ^ 1) * Gm::HBUF;          // h_t
    lds_barrier();                                         // B1
    MF_STAMP(0);
    f16x8 Bh[KB], Bl[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      Bh[kb] = *reinterpret_cast<const f16x8*>(hr + bfo + 1024 * kb);
      Bl[kb] = *reinterpret_cast<const f16x8*>(hr + Gm::PART + bfo + 1024 * kb);
    }
    // Two tiles = two INDEPENDENT accumulator chains, alternated: an MFMA that follows its own predecessor on the same
    // accumulator back to back is forwarded, one that follows it behind anything else (the epilogue's VALU work is meant to sit
    // between the MFMAs) waits ~43 cycles for the write-back (MI355X_MICROARCH.md, per-instruction constants)
    auto mm2 = [&](const int i0, f32x4& a0, f32x4& a1) {
      a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0;
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        a0 = mfma16x16x32_f16(Wl[i0][kb], Bh[kb], a0); a1 = mfma16x16x32_f16(Wl[i0 + 1][kb], Bh[kb], a1);
        a0 = mfma16x16x32_f16(Wh[i0][kb], Bl[kb], a0); a1 = mfma16x16x32_f16(Wh[i0 + 1][kb], Bl[kb], a1);
        a0 = mfma16x16x32_f16(Wh[i0][kb], Bh[kb], a0); a1 = mfma16x16x32_f16(Wh[i0 + 1][kb], Bh[kb], a1);
      }
    };
    // the unpaired tile: its k-blocks alternate between the two chains, which are summed
    auto mm1 = [&](const int i, f32x4& a0, f32x4& a1) {
      a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0;
#pragma unroll
      for (int kb = 0; kb < KB; kb += 2) {
        a0 = mfma16x16x32_f16(Wl[i][kb], Bh[kb], a0); if (kb + 1 < KB) a1 = mfma16x16x32_f16(Wl[i][kb + 1], Bh[kb + 1], a1);
        a0 = mfma16x16x32_f16(Wh[i][kb], Bl[kb], a0); if (kb + 1 < KB) a1 = mfma16x16x32_f16(Wh[i][kb + 1], Bl[kb + 1], a1);
        a0 = mfma16x16x32_f16(Wh[i][kb], Bh[kb], a0); if (kb + 1 < KB) a1 = mfma16x16x32_f16(Wh[i][kb + 1], Bh[kb + 1], a1);
      }
    };
    // forward_full1 x 4 (clstm_compute.cc:308-314), forward_statemem (:504-508), forward_nonlingate (:530-537) of one tile
    auto epi = [&](const int i, const f32x4 acc) -> float {
      const f32x4 gx = *reinterpret_cast<const f32x4*>(smem + Gm::GXS_OFF + gxo[i]);
      const float gi = act_affine(fmaf(acc[0], inv, gx[0]), ACT_SIG_SCALE, 1.0f, 0.0f);
      const float gf = act_affine(fmaf(acc[1], inv, gx[1]), ACT_SIG_SCALE, 1.0f, 0.0f);
      const float go = act_affine(fmaf(acc[2], inv, gx[2]), ACT_SIG_SCALE, 1.0f, 0.0f);
      const float ci = act_affine(fmaf(acc[3], inv, gx[3]), ACT_TANH_SCALE, 2.0f, -1.0f);
      const float c = fmaf(gf, cprev[i], ci * gi);
      const float h = go * tanh_fast(c);
      cprev[i] = c;
      *reinterpret_cast<f32x4*>(smem + Gm::ACT_OFF + gxo[i]) = (f32x4){gi, gf, go, ci};
      *reinterpret_cast<float*>(smem + Gm::CS_OFF + cho + 4 * cellv[i]) = c;
      *reinterpret_cast<float*>(smem + Gm::HS_OFF + cho + 4 * cellv[i]) = h;
      return h * (float)(1 << MF_HS);   // the next step's B operand: two f16 terms of 2^8 h
    };
    f32x4 A0, A1, N0, N1;
    mm2(0, A0, A1);
    MF_STAMP(1);
    stage_gx(gset[PAR]);
    MF_STAMP(2);
    load_gx(t + 2, gset[PAR]);
    if (t > 0) store_rows(t - 1);
    MF_STAMP(3);
    lds_barrier();                                         // B2
    MF_STAMP(4);
#pragma unroll
    for (int u = 0; u < PPW; u++) {
      N0 = A0; N1 = A1;
      if (u + 1 < PPW) { if (tile_on(2 * u + 2)) mm2(2 * u + 2, N0, N1); }
      else if (Gm::SINGLE && tile_on(2 * PPW)) mm1(2 * PPW, N0, N1);
      if (tile_on(2 * u)) {
        const float he = epi(2 * u, A0), ho = epi(2 * u + 1, A1);
        const int pr = w * PPW + u;
        f16x2 hi2, lo2;
        hi2[0] = (_Float16)he; hi2[1] = (_Float16)ho;
        lo2[0] = (_Float16)(he - (float)hi2[0]); lo2[1] = (_Float16)(ho - (float)hi2[1]);
        *reinterpret_cast<f16x2*>(hw + hwo + 256 * pr) = hi2;
        *reinterpret_cast<f16x2*>(hw + Gm::PART + hwo + 256 * pr) = lo2;
      }
      A0 = N0; A1 = N1;
    }
    if (Gm::SINGLE && tile_on(2 * PPW)) {
      const float hs = epi(2 * PPW, A0 + A1);
      const _Float16 hi1 = (_Float16)hs;
      const _Float16 lo1 = (_Float16)(hs - (float)hi1);
      *reinterpret_cast<_Float16*>(hw + n * 16 + cl * 2 + 256 * Gm::FP) = hi1;
      *reinterpret_cast<_Float16*>(hw + Gm::PART + n * 16 + cl * 2 + 256 * Gm::FP) = lo1;
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
  store_rows(t - 1);
#ifdef CLSTM_LSTM_PROF
  if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
    for (int k = 0; k < 8; k++) a.prof[w * 8 + k] = pacc[k];
#endif
}

}  // namespace clstm
#endif  // CLSTM_HIP_EMU
