// lstm_mfma_bwd.h -- the NARROW layer's BACKWARD recurrence batched over 16 lines per workgroup on the matrix cores: the twin of
// lstm_mfma.h for minibatches that fill the chip (>= 640 lines per GPU; below that lstm_seq.h's one-workgroup-per-line kernel).
//
// GenericNPLSTM::backward (clstm.cc:622-653) per own step s = T-1 .. 0 of a line: backward_nonlingate, backward_statemem, four
// backward_full1 (= backward_nonlin0 + backward_lin1) and backward_stack_delay (clstm_compute.cc:231-267, 294-304, 398-410,
// 509-547).  What depends on the previous step is the recurrent delta
//     dh_rec[k] = sum_{gate q, cell j} R_q[j][k] . delta_q[j](s + 1)
// -- here ONE GEMM per step for 16 lines:  dh_rec[no x 16 lines] = R^T[no x 4 no] . delta[4 no x 16 lines]  on
// v_mfma_f32_16x16x32_bf16 with f32 accumulation.  The weight gradient (delta . [1; x; h]^T) and the input deltas stay the hoisted
// GEMMs behind this kernel (gemm_dw.h in its separate-launch form), fed by the same D array the per-line kernel writes.
//
// Arithmetic.  Deltas span ten orders of magnitude inside one minibatch, so the f16 split of the forward kernel (which needs one
// power-of-two scale for the whole operand) does not apply; both operands are split into bf16 TERMS instead (bf16 has the f32
// exponent range; the differences are exact in f32): NT = 2: x = hi + lo, product = hi.hi + hi.lo + lo.hi, what is dropped is
// < 2^-16 |R||delta| per product -- the class of the split weight-gradient products of rounds 3-5, here inside a contraction of
// 400 terms whose result is added to an O(1)-accurate dH; parity: every gate delta of every line within 1e-4 of the line's
// largest (tests/test_mfma_recurrence.py).  (NT = 3, six products, operand-exact, is the same code; its fragments need 156
// registers per wave and leave too few for two waves per SIMD -- not instantiated.)
//
// Geometry (NO cells, NO % 4 == 0, NO <= 128).  M = cells k of dh_rec in tiles of 16 (wave w owns tile w: 7 of the 8 waves at NO =
// 100), N = 16 lines, K = the 4 NO gate rows m = 4 j + q IN THE ORDER OF A D ROW ([cell][gate], lstm_seq.h), blocks of 32.  In the
// MFMA's result layout lane l = 16 cs + n holds result rows 4 cs + i (i = 0..3) of the tile for line n; the tile's rows are packed
// so that row 4 cs + i is cell 16 w + 4 i + cs, and that lane does the element-wise work of those four cells: it fetches their
// four 16-byte pieces of the line's activation row (instruction i: the 64 contiguous bytes of cells 4 i .. 4 i + 3 across the
// lanes cs = 0..3 of a line), reads their c and dH values, keeps their dc carry, stores their four pieces of the delta row, and
// deposits the sixteen deltas as bf16 terms in the LDS image [term][chunk of 8 m][16 lines][8 halfs] that every wave's B fragments of the NEXT step come
// from.  R^T stays in registers (accumulation half of the file) for the whole sequence.  One barrier per step; the image is
// double-buffered; the operands of own step s - 2 (cell states: s - 4) are requested at the end of step s.
#pragma once
#include "lstm_mfma.h"
#ifndef CLSTM_HIP_EMU

namespace clstm {

struct LstmMfmaBwdArgs {
  const unsigned short* W;   // R^T fragments [dir][tile][k-block][term][lane][8 halfs]   (k_pack_mfma_bwd)
  const float *G, *C, *dH;   // as LstmSeqArgs: saved activations [N][ndir][4 no], cell states [N][ndir][no], delta on H [N][ndir][no]
  float* D;                  // gate pre-activation deltas [N][ndir][4 no]
  const int* line_off;
  const int* order;          // [bs] lines, longest first (16 consecutive entries share a workgroup), or null
  int bs, ndir;
  long long N;
  long long prog_off;        // progress words of the weight-gradient items behind D[0] (gemm_dw.h), or -1: the kernel marks its
  int prog_base;             //   lines complete when it ends (the items run as a launch of their own behind this one)
  int dbg;                   // experiments (mfma_bwd_dbg; results are then wrong): 1 no B reads / MFMAs, 2 no image writes,
                             //   4 no delta stores, 8 no operand requests
};

template <int NO, int NT>
struct MfmaBwdGeom {
  static_assert(NO % 4 == 0 && NO >= 16 && NO <= 128 && (NT == 2 || NT == 3), "cells / terms");
  static constexpr int NTL = (NO + 15) / 16;                 // tiles of 16 output cells = waves that compute
  static constexpr int KM = 4 * NO, KB = (KM + 31) / 32, NCH = 4 * KB;   // k = gate rows m = 4 j + q, zero-padded to blocks of 32
  static constexpr int PART = NCH * 256;                     // bytes of one term's image: [chunk of 8 m][16 lines][8 halfs]
  static constexpr int BUF = NT * PART;                      // one buffer: [term][...]
  // operand staging by LDS-DMA, per computing wave: two slots (own step parity) of [g0 g1 g2 g3 dh] x 1 KB (lane l's 16 bytes at
  // 16 l) and a ring of four 1 KB cell-state rows (own step mod 4)
  static constexpr int IN_OFF = 2 * BUF, IN_WAVE = 5 * 1024, IN_SLOT = NTL * IN_WAVE;
  static constexpr int C_OFF = IN_OFF + 2 * IN_SLOT, C_SLOT = NTL * 1024;
  static constexpr int DUMP_OFF = C_OFF + 4 * C_SLOT;        // where lanes without cells write
  static constexpr int SMEM = DUMP_OFF + 64;
  static_assert(SMEM <= 160 * 1024, "LDS");
  static constexpr long long W_HALFS_PER_DIR = (long long)NTL * KB * NT * 64 * 8;
};

// ---- packing: R_q[j][k] = W_q(row j, column 1 + ni + k) (tensor.h:263-264) as A fragments of R^T, split into bf16 terms ----
struct MfmaBwdPackArgs { const float* v; long long p_off[2][4]; int ni, no, ntl, kb, nt; unsigned short* W; };
__global__ __launch_bounds__(256) void k_pack_mfma_bwd(MfmaBwdPackArgs p) {
  const int dir = blockIdx.y;
  const long long per_dir = (long long)p.ntl * p.kb * p.nt * 512;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)p.ntl * p.kb * 512; i += (long long)gridDim.x * 256) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int kb = (int)((i >> 9) % p.kb), tile = (int)((i >> 9) / p.kb);
    const int r = lane & 15;
    const int kout = 16 * tile + 4 * (r & 3) + (r >> 2);   // tile row 4 cs + i holds cell 4 i + cs (see the kernel's memory mapping)
    const int m = kb * 32 + 8 * (lane >> 4) + j, jj = m >> 2, q = m & 3;
    float x = (kout < p.no && jj < p.no) ? p.v[p.p_off[dir][q] + jj + (long long)p.no * (1 + p.ni + kout)] : 0.0f;
    unsigned short* dst = p.W + dir * per_dir + ((long long)(tile * p.kb + kb) * p.nt) * 512 + lane * 8 + j;
    for (int t = 0; t < p.nt; t++) {
      const __bf16 h = (__bf16)x;
      dst[t * 512] = __builtin_bit_cast(unsigned short, h);
      x -= (float)h;
    }
  }
}

template <int NO, int NT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_bwd_mfma_kernel(LstmMfmaBwdArgs a) {
  using Gm = MfmaBwdGeom<NO, NT>;
  constexpr int KB = Gm::KB, NTL = Gm::NTL, PART = Gm::PART, BUF = Gm::BUF;
  char* const smem = dyn_smem<char>();
  const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
  const int n = lane & 15, cs = lane >> 4;
  const int dir = blockIdx.y, grp = blockIdx.x;
  const int nd = a.ndir;

  // this lane's line (column n of the product)
  const int gl = grp * 16 + n;
  const int b = gl < a.bs ? (a.order ? a.order[gl] : gl) : -1;
  const int off = b >= 0 ? a.line_off[b] : 0;
  const int T = b >= 0 ? a.line_off[b + 1] - off : 0;
  int tmx = T;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) { const int o = __shfl_xor(tmx, m, 64); tmx = o > tmx ? o : tmx; }
  const int Tmax = wave_uniform(tmx);
  int* const progw = a.prog_off >= 0 && b >= 0 && tid < 16 ? reinterpret_cast<int*>(a.D + a.prog_off) + ((size_t)dir * a.bs + b) * PROG_STRIDE : nullptr;
  if (Tmax <= 0) {
    if (progw) store_i32_wt(progw, a.prog_base);
    return;
  }

  const bool act = w < NTL;                                 // wave-uniform: this wave owns a tile of output cells
  // this lane's four cells: 16 w + 4 i + cs, i = 0..3 (result register i of the tile whose row 4 cs + i was packed with that
  // cell).  NOT 4 cs + i: instruction i of a group of four then touches, per line, the 64 contiguous bytes of cells 4 i .. 4 i + 3
  // (lanes cs = 0..3) -- 16 requests of 64 bytes -- where the natural assignment issued 64 separate 16-byte requests per
  // instruction and the step spent 4,500 cycles in the address pipeline (measured by leaving the stores / requests out)
  const int c0 = 16 * w + cs;                               // + 4 i
  const bool vrow = act && 16 * w + 4 * cs < NO;            // cells 16 w + 4 cs .. + 3: what this lane FETCHES of the c / dH rows
  // R^T fragments of the wave's tile, resident for the whole sequence
  u16x8 A[KB][NT];
#pragma unroll
  for (int kb = 0; kb < KB; kb++)
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const unsigned short* wp = a.W + dir * Gm::W_HALFS_PER_DIR + ((long long)(((act ? w : 0) * KB + kb) * NT + t)) * 512 + lane * 8;
      A[kb][t] = __builtin_bit_cast(u16x8, *reinterpret_cast<const u32x4*>(wp));
    }
  // (consumed HERE once: the waits for these loads belong in front of the loop.  Pinned only inside it, hipcc's wait-count pass
  //  sees them "possibly pending" at the loop header for ever and puts s_waitcnt vmcnt(39) .. vmcnt(14) at the top of every step --
  //  which in the steady state means "all but the newest 14 memory operations have returned": the operand prefetch of two
  //  steps and the delta stores could never stay in flight)
#pragma unroll
  for (int kb = 0; kb < KB; kb++)
#pragma unroll
    for (int t = 0; t < NT; t++) asm volatile("" : "+a"(A[kb][t]));
  // delta image: delta(T) = 0 in both buffers, and the zero padding of k for good
  for (int i = tid * 16; i < 2 * BUF; i += 512 * 16) *reinterpret_cast<u32x4*>(smem + i) = (u32x4){0u, 0u, 0u, 0u};

  const unsigned gstr = (unsigned)nd * 4 * NO * 4, cstr = (unsigned)nd * NO * 4;
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * gstr);
  const BufF32 dbuf = make_buf(a.D, (size_t)a.N * gstr);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * cstr);
  const BufF32 hbuf = make_buf(a.dH, (size_t)a.N * cstr);
  unsigned glo[4];                                          // activation / delta row: 16 bytes of cell c0 + 4 i
#pragma unroll
  for (int i = 0; i < 4; i++) glo[i] = act && c0 + 4 * i < NO ? ((unsigned)dir * 4 * NO + 4 * (c0 + 4 * i)) * 4u : BUF_OOB_BASE;
  const unsigned clo = vrow ? ((unsigned)dir * NO + 16 * w + 4 * cs) * 4u : BUF_OOB_BASE;
  // frame of own step s (Reversed = index arithmetic, clstm.cc:458-478); validity as arithmetic, never as control flow
  auto tok = [&](int s) -> unsigned { return (unsigned)(off + (dir == 0 ? s : T - 1 - s)); };
  auto bad = [&](int s) -> unsigned { return (unsigned)s >= (unsigned)T ? 0x80000000u : 0u; };   // s < 0 or s >= T
  // Operand prefetch by LDS-DMA (buffer_load ... lds: no register ever holds a prefetched operand).  At the end of own step s a
  // wave requests activations + dH of step s - 2 into slot (s & 1) of its staging area and the cell states of step s - 4 into
  // ring entry (s & 3) -- c_s is needed at step s AND, as c_{s-1}, at step s + 1 -- behind its four delta stores: ten memory
  // operations per step in a fixed order, so ONE explicit s_waitcnt vmcnt(10) in front of a step's operand reads says "everything
  // up to the g / dH requests of two steps ago has arrived" and leaves the newest batch and the stores in front of it in flight.
  // (Register prefetch, first versions: hipcc either waited for the newest batch behind the step's own stores -- 4 us per step --
  //  or, with a longer ring, copied the newest loads between registers at the loop's back-edge behind s_waitcnt vmcnt(0).)
  char* const in_w = smem + Gm::IN_OFF + (act ? w : 0) * Gm::IN_WAVE;     // + slot * IN_SLOT
  char* const c_w = smem + Gm::C_OFF + (act ? w : 0) * 1024;              // + entry * C_SLOT
  const unsigned dbg_st = (a.dbg & 4) ? 0x80000000u : 0u, dbg_rq = (a.dbg & 8) ? 0x80000000u : 0u;
  auto request = [&](int s, int slot) {
    const unsigned o = bad(s) | dbg_rq, tk = tok(s);
#pragma unroll
    for (int i = 0; i < 4; i++) lds_dma16(gbuf, (tk * gstr + glo[i]) | o, in_w + slot * Gm::IN_SLOT + i * 1024);
    lds_dma16(hbuf, (tk * cstr + clo) | o, in_w + slot * Gm::IN_SLOT + 4 * 1024);
  };
  auto request_c = [&](int s, int entry) { lds_dma16(cbuf, (tok(s) * cstr + clo) | bad(s) | dbg_rq, c_w + entry * Gm::C_SLOT); };
  // LDS: B fragments (lane (cs, n): chunk 4 kb + cs, line n) and this lane's two chunks of the image it writes
  const unsigned bfo = (unsigned)(cs * 256 + n * 16);
  // image: cell c0 + 4 i = chunk (c0 + 4 i) >> 1, half (c0 & 1) of its 16 bytes for line n
  const unsigned iwo = (unsigned)(((c0 >> 1)) * 256 + n * 16 + (c0 & 1) * 8);   // + 512 i  (c0 + 4 i) >> 1 = (c0 >> 1) + 2 i
  // c / dH of cell c0 + 4 i: element cs of the 16 bytes lane (i, n) fetched
  const unsigned cro = (unsigned)(n * 16 + cs * 4);                               // + 256 i

  if (act) {
#pragma unroll
    for (int k = 0; k < 4; k++) request_c(Tmax - 1 - k, k);
    request(Tmax - 1, 0);
    request(Tmax - 2, 1);
  }
  float dcc[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // (waits vmcnt(0): the prologue's requests have landed; from here on the count per step is exact)

  // iteration it = Tmax - 1 - s, K = it mod 4: operands in slot K & 1, c_s = ring entry K, c_{s-1} = entry K + 1; image buffer K & 1
  auto step = [&](const int s, auto k_tag) __attribute__((always_inline)) {
    constexpr int K = decltype(k_tag)::value;
    constexpr int PAR = K & 1;
    const char* const ir = smem + PAR * BUF;          // delta(s + 1)
    char* const iw = smem + (PAR ^ 1) * BUF;          // delta(s)
    lds_barrier();
#pragma unroll
    for (int kb = 0; kb < KB; kb++)
#pragma unroll
      for (int t = 0; t < NT; t++) asm volatile("" : "+a"(A[kb][t]));   // fragments live in the accumulation half (see lstm_mfma.h)
    // this step's operands: requested two (cell states: four / three) steps ago.  Everything that does not depend on this step's
    // product -- tanh(c), the derivative factors, the second factor of every gate delta -- is computed HERE, in front of the
    // MFMA stream, so that the scheduler can issue it between the (dependent) MFMAs (the per-line kernel's rule, lstm_seq.h)
    wait_vmcnt<10>();   // (10, not 11: whatever order the ten operations of a step were issued in, the batch of two steps ago is complete)
    f32x4 f_gi, f_gf, f_go, f_ci, gthv, gfv, dhv;
    {
      const char* const src = in_w + (K & 1) * Gm::IN_SLOT + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(src + i * 1024);
        const float cci = *reinterpret_cast<const float*>(c_w + K * Gm::C_SLOT + cro + 256 * i);
        const float cm1 = *reinterpret_cast<const float*>(c_w + ((K + 1) & 3) * Gm::C_SLOT + cro + 256 * i);
        dhv[i] = *reinterpret_cast<const float*>(in_w + (K & 1) * Gm::IN_SLOT + 4 * 1024 + cro + 256 * i);
        const float gi = g[0], gf = g[1], go = g[2], ci = g[3];
        const float th = tanh_fast(cci);                          // backward_nonlingate recomputes tanh(state)
        gthv[i] = go * fmaf(-th, th, 1.0f);                        // state.d += (1 - t^2) go out.d
        gfv[i] = gf;
        f_gi[i] = ci * fmaf(-gi, gi, gi);                          // gi.d = c.d ci, through sigma'
        f_gf[i] = cm1 * fmaf(-gf, gf, gf);                         // gf.d = c.d c_{s-1}  (c_{-1} = 0: "untouched when last < 0")
        f_go[i] = th * fmaf(-go, go, go);                          // go.d = tanh(c) out.d
        f_ci[i] = gi * fmaf(-ci, ci, 1.0f);                        // ci.d = c.d gi, through tanh'
      }
    }
    // dh_rec = R^T . delta(s + 1): two accumulator chains alternate over the k-blocks (an MFMA behind its own predecessor is
    // forwarded only back to back); smallest terms first
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (!(a.dbg & 1))
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      u16x8 B[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) B[t] = *reinterpret_cast<const u16x8*>(ir + t * PART + bfo + 1024 * kb);
#pragma unroll
      for (int wt = NT - 1; wt >= 0; wt--)
#pragma unroll
        for (int ta = 0; ta <= wt; ta++) {
          if (kb & 1) acc1 = mfma16x16x32_bf16(A[kb][ta], B[wt - ta], acc1);
          else acc0 = mfma16x16x32_bf16(A[kb][ta], B[wt - ta], acc0);
        }
    }
    const f32x4 acc = acc0 + acc1;
    // the dependent tail (lstm_seq.h:lstm_bwd_body, the same expressions)
    f32x4 dl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float dh = dhv[i] + acc[i];                           // delta from above + recurrent delta (clstm.cc:626-628, :646)
      const float dc = fmaf(gthv[i], dh, dcc[i]);
      dcc[i] = dc * gfv[i];                                       // c_{s-1}.d += c.d gf       (backward_statemem, :509-515)
      dl[i][0] = dc * f_gi[i];
      dl[i][1] = dc * f_gf[i];
      dl[i][2] = dh * f_go[i];
      dl[i][3] = dc * f_ci[i];
    }
    {
      const unsigned o = bad(s), tk = tok(s);
#pragma unroll
      for (int i = 0; i < 4; i++) buf_store4(dbuf, (tk * gstr + glo[i]) | o | dbg_st, dl[i]);
    }
    // the sixteen deltas as bf16 terms into the next step's B image: per cell 8 bytes (m = 4 cell .. + 3) per term
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float e[4] = {dl[i][0], dl[i][1], dl[i][2], dl[i][3]};
      const bool ok = act && c0 + 4 * i < NO && !(a.dbg & 2);
      char* const dst = ok ? iw + iwo + 512 * i : smem + Gm::DUMP_OFF;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        u32x2 h;
        h[0] = bf16_pack2(e[0], e[1]);
        h[1] = bf16_pack2(e[2], e[3]);
        *reinterpret_cast<u32x2*>(dst + (ok ? t * PART : 0)) = h;
        if (t + 1 < NT) {
          e[0] -= __builtin_bit_cast(float, h[0] << 16); e[1] -= __builtin_bit_cast(float, h[0] & 0xffff0000u);
          e[2] -= __builtin_bit_cast(float, h[1] << 16); e[3] -= __builtin_bit_cast(float, h[1] & 0xffff0000u);
        }
      }
    }
    // (the reads above have returned: the data was used -- a request may overwrite the slot)
    request(s - 2, K & 1);
    request_c(s - 4, K);
  };
  if (act) {
    int s = Tmax - 1;
    auto four = [&](const int s0) __attribute__((always_inline)) {
      step(s0, std::integral_constant<int, 0>{});
      step(s0 - 1, std::integral_constant<int, 1>{});
      step(s0 - 2, std::integral_constant<int, 2>{});
      step(s0 - 3, std::integral_constant<int, 3>{});
    };
    // the first four steps are peeled: the wait counts of the loop's first step are computed for BOTH ways into it, and on the
    // way in from the prologue the operands sit a few loads from the end of the queue (s_waitcnt vmcnt(3) in the steady state =
    // wait for the newest batch and the delta stores in front of it); behind four steps both ways look alike
    if (s >= 3) { four(s); s -= 4; }
    for (; s >= 3; s -= 4) four(s);
    if (s >= 0) step(s, std::integral_constant<int, 0>{});
    if (s >= 1) step(s - 1, std::integral_constant<int, 1>{});
    if (s >= 2) step(s - 2, std::integral_constant<int, 2>{});
  } else {
    for (int s = 0; s < Tmax; s++) lds_barrier();   // (a wave without a tile of cells: NO = 100 has seven tiles) keeps the count
  }
  if (a.prog_off >= 0) {   // the lines are complete: every store of every wave acknowledged, then the words the items look at
    drain_vmem();
    __syncthreads();
    if (progw) store_i32_wt(progw, a.prog_base + T);
  }
}


// ---- the same recurrence with every global access a whole ROW (the default; option bwd_mfma_rows=0 selects the kernel above) ----
// lstm_bwd_mfma_kernel above is bound by the per-CU address pipeline below 2048 lines (64-byte requests: ~1.4 us of a 2.95 us
// step, profiles/r06_mfma_bwd_leaveout.txt); the forward kernel, whose rows go through LDS and leave 1 KB per instruction, pays
// almost nothing for 38 KB per step.  Here the operands arrive the same way: wave v requests the rows of lines 2 v, 2 v + 1 by
// LDS-DMA -- the activation row (1600 B at NO = 100: two 16-byte-per-lane instructions), the c and dH rows (400 B: two dword
// instructions each) -- into row images padded by 16 bytes per line (conflict-free for the compute lanes' reads), the compute
// lanes write their deltas into a row image, and wave v stores the delta rows of its two lines during the next step.  LDS: ONE
// delta image for the MFMA (a second barrier per step separates its readers from its writers) + two slots of activation / dH
// rows + a ring of three c rows + the delta-row image = 157 KB at NO = 100.  Measured (scripts/dbg/bwd_rows.sh): 256 lines 0.603 ->
// 0.488 ms, 1024 lines 0.676 -> 0.549, 2048 lines 0.794 -> 0.691 (= 4.75 TB/s, 0.59 of the HBM peak).
template <int NO, int NT>
struct MfmaBwdRowsGeom {
  using G0 = MfmaBwdGeom<NO, NT>;
  static constexpr int GROW = (4 * NO * 4 + 1023) / 1024 * 1024 + 16;   // activation row: whole 1 KB instructions + pad
  static constexpr int CROW = (NO * 4 + 255) / 256 * 256 + 16;          // c / dH row: whole 256-byte instructions + pad
  static constexpr int DROW = 4 * NO * 4 + 16;                          // delta row image
  static constexpr int NG = GROW / 1024, NC = CROW / 256;               // instructions per row
  static constexpr int BIMG = G0::BUF;
  static constexpr int GS_OFF = BIMG, GS_SLOT = 16 * GROW;
  static constexpr int CR_OFF = GS_OFF + 2 * GS_SLOT, CR_ENT = 16 * CROW;
  static constexpr int DH_OFF = CR_OFF + 3 * CR_ENT;
  static constexpr int DI_OFF = DH_OFF + 2 * CR_ENT;
  static constexpr int DUMP_OFF = DI_OFF + 16 * DROW;
  static constexpr int SMEM = DUMP_OFF + 64;
  static_assert(SMEM <= 160 * 1024, "LDS");
};

// REPORT: the kernel is one ROLE of a launch whose other workgroups consume the deltas while it runs (lstm_bwd_mfma_dw_kernel
// below): delta rows are stored write-through, and every wave publishes, per step, how many iterations of its two lines are
// complete in memory (the progress words gemm_dw.h's monitor reads)
template <int NO, int NT, bool REPORT>
DEVFN void lstm_bwd_mfma_rows_body(const LstmMfmaBwdArgs& a, const int grp, const int dir) {
  using Gm = MfmaBwdGeom<NO, NT>;
  using Gr = MfmaBwdRowsGeom<NO, NT>;
  constexpr int KB = Gm::KB, NTL = Gm::NTL, PART = Gm::PART;
  char* const smem = dyn_smem<char>();
  const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
  const int n = lane & 15, cs = lane >> 4;
  const int nd = a.ndir;
  const int gl = grp * 16 + n;
  const int b = gl < a.bs ? (a.order ? a.order[gl] : gl) : -1;
  const int off = b >= 0 ? a.line_off[b] : 0;
  const int T = b >= 0 ? a.line_off[b + 1] - off : 0;
  int tmx = T;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) { const int o = __shfl_xor(tmx, m, 64); tmx = o > tmx ? o : tmx; }
  const int Tmax = wave_uniform(tmx);
  int* const progw = a.prog_off >= 0 && b >= 0 && tid < 16 ? reinterpret_cast<int*>(a.D + a.prog_off) + ((size_t)dir * a.bs + b) * PROG_STRIDE : nullptr;
  if (Tmax <= 0) {
    if (progw) store_i32_wt(progw, a.prog_base);
    return;
  }
  // the two lines whose rows this wave moves (scalars)
  int offj[2], Tj[2];
#pragma unroll
  for (int j = 0; j < 2; j++) { offj[j] = __builtin_amdgcn_readlane(off, 2 * w + j); Tj[j] = __builtin_amdgcn_readlane(T, 2 * w + j); }
  // REPORT: lane j < 2 of the wave owns the progress word of line 2 w + j
  const int rb = __shfl(b, 2 * w + (lane & 1), 64), rT = __shfl(T, 2 * w + (lane & 1), 64);
  int* const rword = REPORT && a.prog_off >= 0 && lane < 2 && rb >= 0 ? reinterpret_cast<int*>(a.D + a.prog_off) + ((size_t)dir * a.bs + rb) * PROG_STRIDE : nullptr;

  const bool act = w < NTL;
  const int c0 = 16 * w + cs;                               // this lane's cells c0 + 4 i (see lstm_bwd_mfma_kernel)
  u16x8 A[KB][NT];
#pragma unroll
  for (int kb = 0; kb < KB; kb++)
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const unsigned short* wp = a.W + dir * Gm::W_HALFS_PER_DIR + ((long long)(((act ? w : 0) * KB + kb) * NT + t)) * 512 + lane * 8;
      A[kb][t] = __builtin_bit_cast(u16x8, *reinterpret_cast<const u32x4*>(wp));
    }
#pragma unroll
  for (int kb = 0; kb < KB; kb++)
#pragma unroll
    for (int t = 0; t < NT; t++) asm volatile("" : "+a"(A[kb][t]));
  for (int i = tid * 16; i < Gr::BIMG; i += 512 * 16) *reinterpret_cast<u32x4*>(smem + i) = (u32x4){0u, 0u, 0u, 0u};

  const unsigned gstr = (unsigned)nd * 4 * NO * 4, cstr = (unsigned)nd * NO * 4;
  const BufF32 gbuf = make_buf(a.G, (size_t)a.N * gstr);
  const BufF32 dbuf = make_buf(a.D, (size_t)a.N * gstr);
  const BufF32 cbuf = make_buf(a.C, (size_t)a.N * cstr);
  const BufF32 hbuf = make_buf(a.dH, (size_t)a.N * cstr);
  const unsigned dbg_st = (a.dbg & 4) ? 0x80000000u : 0u, dbg_rq = (a.dbg & 8) ? 0x80000000u : 0u;
  // row requests / row stores of line 2 w + j at own step s: lane l = bytes 16 l (activations, deltas) / 4 l (c, dH) of a piece
  auto rtok = [&](int j, int s) -> unsigned { return (unsigned)(offj[j] + (dir == 0 ? s : Tj[j] - 1 - s)); };
  auto rbad = [&](int j, int s) -> unsigned { return (unsigned)s >= (unsigned)Tj[j] ? 0x80000000u : 0u; };
  // request_gd: activations + dH of own step sg into slot `slot`; request_c: cell states of own step sc into ring entry `cent`.  c_s is needed at step s
  // AND, as c_{s-1}, one step EARLIER (step s + 1): it is requested three steps ahead, the rest two -- whatever a step reads was
  // requested at least two steps before it and waited for (wait_vmcnt below) one barrier before it
  auto request_gd = [&](int sg, int slot) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int row = 2 * w + j;
      const unsigned o = rbad(j, sg) | dbg_rq, tk = rtok(j, sg);
#pragma unroll
      for (int q = 0; q < Gr::NG; q++) {
        const unsigned bo = 1024u * q + 16u * lane;
        lds_dma16(gbuf, bo < 4u * NO * 4 ? (tk * gstr + (unsigned)dir * 4 * NO * 4 + bo) | o : BUF_OOB,
                  smem + Gr::GS_OFF + slot * Gr::GS_SLOT + row * Gr::GROW + 1024 * q);
      }
#pragma unroll
      for (int q = 0; q < Gr::NC; q++) {
        const unsigned bo = 256u * q + 4u * lane;
        lds_dma4(hbuf, bo < 4u * NO ? (tk * cstr + (unsigned)dir * NO * 4 + bo) | o : BUF_OOB,
                 smem + Gr::DH_OFF + slot * Gr::CR_ENT + row * Gr::CROW + 256 * q);
      }
    }
  };
  auto request_c = [&](int sc, int cent) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int row = 2 * w + j;
      const unsigned oc = rbad(j, sc) | dbg_rq, tkc = rtok(j, sc);
#pragma unroll
      for (int q = 0; q < Gr::NC; q++) {
        const unsigned bo = 256u * q + 4u * lane;
        lds_dma4(cbuf, bo < 4u * NO ? (tkc * cstr + (unsigned)dir * NO * 4 + bo) | oc : BUF_OOB,
                 smem + Gr::CR_OFF + cent * Gr::CR_ENT + row * Gr::CROW + 256 * q);
      }
    }
  };
  constexpr int NREQ = 2 * (Gr::NG + 2 * Gr::NC);          // memory operations of one request() per wave
  constexpr int NST = 2 * Gr::NG;                          // ... of one store_rows()
  auto store_rows = [&](int s) {                           // the delta rows of own step s (complete in the image)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int row = 2 * w + j;
      const unsigned o = rbad(j, s) | dbg_st, tk = rtok(j, s);
#pragma unroll
      for (int q = 0; q < Gr::NG; q++) {
        const unsigned bo = 1024u * q + 16u * lane;
        const f32x4 v = *reinterpret_cast<const f32x4*>(smem + Gr::DI_OFF + row * Gr::DROW + (bo < 4u * NO * 4 ? bo : 0u));
        const unsigned so = bo < 4u * NO * 4 ? (tk * gstr + (unsigned)dir * 4 * NO * 4 + bo) | o : BUF_OOB;
        if constexpr (REPORT) buf_store4_wt(dbuf, so, v); else buf_store4(dbuf, so, v);
      }
    }
  };
  const unsigned bfo = (unsigned)(cs * 256 + n * 16);
  const unsigned iwo = (unsigned)(((c0 >> 1)) * 256 + n * 16 + (c0 & 1) * 8);   // + 512 i
  const unsigned gro = (unsigned)(n * Gr::GROW + c0 * 16), cro = (unsigned)(n * Gr::CROW + c0 * 4), dwo = (unsigned)(n * Gr::DROW + c0 * 16);   // + 64 i / 16 i / 64 i

  // prologue: activations / dH of the first two steps, c of the first three
  request_gd(Tmax - 1, 0);
  request_gd(Tmax - 2, 1);
#pragma unroll
  for (int k = 0; k < 3; k++) request_c(Tmax - 1 - k, k);
  float dcc[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // (waits vmcnt(0))

  // iteration it = Tmax - 1 - s, K = it mod 6: rows in slot K & 1, c_s = ring entry K % 3, c_{s-1} = entry (K + 1) % 3
  auto step = [&](const int s, auto k_tag) __attribute__((always_inline)) {
    constexpr int K = decltype(k_tag)::value;
    lds_barrier();                                          // rows of this step (and its c_{s-1}) have landed; delta(s+1) is in the images
    if (s + 1 < Tmax) store_rows(s + 1);                    // (wave-uniform; the first step has no predecessor)
    f32x4 f_gi, f_gf, f_go, f_ci, gthv, gfv, dhv;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (act) {
#pragma unroll
      for (int kb = 0; kb < KB; kb++)
#pragma unroll
        for (int t = 0; t < NT; t++) asm volatile("" : "+a"(A[kb][t]));
      const char* const gs = smem + Gr::GS_OFF + (K & 1) * Gr::GS_SLOT + gro;
      const char* const hs = smem + Gr::DH_OFF + (K & 1) * Gr::CR_ENT + cro;
      const char* const c0s = smem + Gr::CR_OFF + (K % 3) * Gr::CR_ENT + cro;
      const char* const c1s = smem + Gr::CR_OFF + ((K + 1) % 3) * Gr::CR_ENT + cro;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const bool ok = c0 + 4 * i < NO;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gs + (ok ? 64 * i : 0));
        const float cci = *reinterpret_cast<const float*>(c0s + (ok ? 16 * i : 0));
        const float cm1 = *reinterpret_cast<const float*>(c1s + (ok ? 16 * i : 0));
        dhv[i] = *reinterpret_cast<const float*>(hs + (ok ? 16 * i : 0));
        const float gi = g[0], gf = g[1], go = g[2], ci = g[3];
        const float th = tanh_fast(cci);
        gthv[i] = go * fmaf(-th, th, 1.0f);
        gfv[i] = gf;
        f_gi[i] = ci * fmaf(-gi, gi, gi);
        f_gf[i] = cm1 * fmaf(-gf, gf, gf);
        f_go[i] = th * fmaf(-go, go, go);
        f_ci[i] = gi * fmaf(-ci, ci, 1.0f);
      }
      if (!(a.dbg & 1))
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        u16x8 B[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) B[t] = *reinterpret_cast<const u16x8*>(smem + t * PART + bfo + 1024 * kb);
#pragma unroll
        for (int wt = NT - 1; wt >= 0; wt--)
#pragma unroll
          for (int ta = 0; ta <= wt; ta++) {
            if (kb & 1) acc1 = mfma16x16x32_bf16(A[kb][ta], B[wt - ta], acc1);
            else acc0 = mfma16x16x32_bf16(A[kb][ta], B[wt - ta], acc0);
          }
      }
    }
    const f32x4 acc = acc0 + acc1;
    lds_barrier();                                          // every wave has read delta(s+1), its rows and its operands: all may be overwritten
    if (act) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const bool ok = c0 + 4 * i < NO;
        const float dh = dhv[i] + acc[i];
        const float dc = fmaf(gthv[i], dh, dcc[i]);
        dcc[i] = dc * gfv[i];
        float e[4] = {dc * f_gi[i], dc * f_gf[i], dh * f_go[i], dc * f_ci[i]};
        *reinterpret_cast<f32x4*>(ok ? smem + Gr::DI_OFF + dwo + 64 * i : smem + Gr::DUMP_OFF) = (f32x4){e[0], e[1], e[2], e[3]};
        const bool oki = ok && !(a.dbg & 2);
        char* const dst = oki ? smem + iwo + 512 * i : smem + Gr::DUMP_OFF + 16;
#pragma unroll
        for (int t = 0; t < NT; t++) {
          u32x2 h;
          h[0] = bf16_pack2(e[0], e[1]);
          h[1] = bf16_pack2(e[2], e[3]);
          *reinterpret_cast<u32x2*>(dst + (oki ? t * PART : 0)) = h;
          if (t + 1 < NT) {
            e[0] -= __builtin_bit_cast(float, h[0] << 16); e[1] -= __builtin_bit_cast(float, h[0] & 0xffff0000u);
            e[2] -= __builtin_bit_cast(float, h[1] << 16); e[3] -= __builtin_bit_cast(float, h[1] & 0xffff0000u);
          }
        }
      }
    }
    // the requests of the previous step (operands of step s - 1) must be in LDS before the next barrier; only this step's row
    // stores were issued behind them
    wait_vmcnt<NST>();
    // REPORT: everything this wave issued before this step's row stores has completed -- the delta rows of own steps >= s + 2
    // of its two lines are in memory: T - (s + 2) iterations (a line not yet started: 0)
    if constexpr (REPORT) {
      if (rword) { const int it = rT - (s + 2); store_i32_wt(rword, a.prog_base + (it < 0 ? 0 : it)); }
    }
    request_gd(s - 2, K & 1);
    request_c(s - 3, K % 3);
  };
  {
    int s = Tmax - 1;
    auto six = [&](const int s0) __attribute__((always_inline)) {
      step(s0, std::integral_constant<int, 0>{});
      step(s0 - 1, std::integral_constant<int, 1>{});
      step(s0 - 2, std::integral_constant<int, 2>{});
      step(s0 - 3, std::integral_constant<int, 3>{});
      step(s0 - 4, std::integral_constant<int, 4>{});
      step(s0 - 5, std::integral_constant<int, 5>{});
    };
    for (; s >= 5; s -= 6) six(s);
    if (s >= 0) step(s, std::integral_constant<int, 0>{});
    if (s >= 1) step(s - 1, std::integral_constant<int, 1>{});
    if (s >= 2) step(s - 2, std::integral_constant<int, 2>{});
    if (s >= 3) step(s - 3, std::integral_constant<int, 3>{});
    if (s >= 4) step(s - 4, std::integral_constant<int, 4>{});
  }
  lds_barrier();
  store_rows(0);
  (void)NREQ;
  if (a.prog_off >= 0) {
    drain_vmem();
    __syncthreads();
    if (progw) store_i32_wt(progw, a.prog_base + T);
  }
}
template <int NO, int NT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_bwd_mfma_rows_kernel(LstmMfmaBwdArgs a) {
  lstm_bwd_mfma_rows_body<NO, NT, false>(a, (int)blockIdx.x, (int)blockIdx.y);
}

}  // namespace clstm
#endif  // CLSTM_HIP_EMU
