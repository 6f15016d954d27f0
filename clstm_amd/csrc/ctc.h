// ctc.h -- clstm's CTC alignment (ctc.cc:24-134) and trivial_decode (ctc.cc:159-190) on device.
//
// One workgroup per text line.  clstm's variant is NOT the textbook CTC: transitions are
// stay / advance-by-one only, every state and frame carries a skip = -5 boundary term,
// log_add cuts off at |x-y| > 10 (tensor.h:86-89), posteriors are normalised per STATE over
// time and then per frame.  All of that is replicated here:
//   A. lmatch[t][s] = log(out_t[class_s]), out_t = max(1e-5, p_t) / sum      (ctc.cc:66-77)
//   B. waves 0-3 run the forward recursion, waves 4-7 the same recursion on the (t,s)-reversed
//      lattice (= forwardbackward, ctc.cc:42-55).  The label axis is spread over 256 lanes
//      (ceil(S/256) states each); the j-1 neighbour of a lane's first state is handed over
//      through a double-buffered LDS vector, so only t is serial (one barrier per frame).
//   C. epath = limexp(both - max)                                            (ctc.cc:82)
//   D. per-state normalisation over t, floor 1e-9, double accumulator        (ctc.cc:83-88)
//   E. aligned[t][c] = sum_s epath[t][s] [class_s == c]; per-frame normalise (ctc.cc:91-109)
//      and the fused delta  d = aligned - p                                  (clstmhl.h:211-212)
// Lines whose lattice fits in LDS (T <= tile, T*S <= 13312: the OCR benchmark shape) take ctc_short_line():
// the same arithmetic, organised for one CU -- match scores once per DISTINCT class (a blank-interleaved
// target has L+1 equal blank columns) and, where the carve has room, read by the recursion straight from LDS
// (ctc_lattice: LDS_SRC); up to 64 states and 256 frames: phases C / D with lane = state and the cells in registers;
// the lattice tile resident in LDS from C to E; guards branch-free (clamped index + select, masked stores to a
// dump word) in the tails and absent in whole batches, so that independent elements interleave instead
// of paying the LDS / double-precision latency one element at a time.
// Targets are given as one class per state (the Classes overload, ctc.cc:136-146; mktargets'
// blank-interleaved list for OCR lines, ctc.cc:148-157).
// Numerical note: in step E the blank class keeps the reference's double accumulator; a label that
// occurs k >= 3 times in one transcript is accumulated in float (<= 1 ulp from the reference's
// double accumulator narrowed to Float).
#pragma once
#include <type_traits>
#include "devintrin.h"
#define CR_FN DEVFN
#include "cr_math.h"

namespace clstm {

constexpr int CTC_THREADS = 512;   // waves 0-3: forward recursion, waves 4-7: reversed-lattice recursion
constexpr int CTC_GROUP = 256;     // lanes per recursion when S > 64
constexpr int CTC_RMAX = 8;        // register-resident recursion: up to 2048 target states per line (transcripts of up to
                                   // 1023 labels; <= 512 states run the 2-states-per-lane instantiation).  Longer
                                   // transcripts -- the reference has no limit (ctc.cc:57-112) -- take ctc_lattice_huge:
                                   // states in rounds of 256 per frame, the previous row read back from the lattice
constexpr int CTC_SMAX_LDS = CTC_GROUP * CTC_RMAX;   // the per-state LDS arrays are carved for at most this many states
constexpr int CTC_MLP = 8;         // independent global loads a thread keeps in flight in the streaming phases
constexpr int CTC_MLPT = 20;       // ... in the tile staging / write-out loops of the tiled path (a tile's frames of one wave at once)
constexpr int CTC_MAX_TILE = 256;  // frames per LDS tile (phases A and E)
#ifndef CLSTM_CTC_PD
#define CLSTM_CTC_PD 8
#endif
constexpr int CTC_PD = CLSTM_CTC_PD;   // frames the one-wave lattice recursions request their match scores ahead

// what a workgroup needs to know about its line, in ONE 32-byte record indexed by the workgroup: the kernel used to walk
// order[] -> line_off[] / state_off[] / lat_off[] -> states / posteriors, three dependent trips to memory before its first
// useful load; with the record it is two
struct CtcLine { long long lat_off; int b, off, T, soff, S, pad; };
struct CtcArgs {
  const CtcLine* lines;  // [bs] per workgroup (largest lattice first)
  const float* P;        // [N][nc] softmax outputs
  float* Dz;             // [N][nc] out: aligned - P
  float* aligned;        // [N][nc] out (optional, may be null): alignment posteriors
  const int* states;     // packed state classes (a line's at lines[].soff)
  float* lat;            // lattice workspace: per line 3*T*S floats at lines[].lat_off
  const double* tables;  // device copy of ctc_tables.h: exp2_32[32] | invc[64] | logc[64] | softplus[929][4]
  int nc;
  int ncp;               // LDS row stride of the class tile (odd)
  int tile;              // frames per LDS tile
  int smax;              // max states of any line in the batch (sizes the LDS carve)
  long long* prof;       // optional [16] phase timestamps of block 0 (diagnostics)
  int float_logadd;      // experiment option ctc_float: log_add on the float transcendentals (ctc_align_kernel<true>; host side only)
};

// LDS carve shared by host (size) and kernel (offsets); all offsets in 4-byte words
struct CtcLds {
  int tables, part, tot, rowbuf, etile, asum, states, lists, ucol, ucls, ccol, vx, red, dump, words;
};
inline __host__ __device__ CtcLds ctc_lds_layout(int tile, int ncp, int smax) {
  CtcLds l;
  int o = 0;
  l.tables = o; o += CTC_TABLE_WORDS;  // doubles first: 8-byte aligned
  l.part = o;   o += 2 * 512;
  l.tot = o;    o += 2 * smax;
  o = (o + 3) & ~3;                    // 16-byte aligned: cleared with ds_write_b128
  l.rowbuf = o; o += tile * ncp;
  l.etile = o;  o += tile * (smax | 1);
  l.asum = o;   o += tile;
  l.states = o; o += smax;
  l.lists = o;  o += smax;   // short-line path: [blank states | first label states | repeats]
  l.ucol = o;   o += smax;   //   column of a state in the table of distinct classes
  l.ucls = o;   o += smax;   //   class of a column
  l.ccol = o;   o += ncp <= CTC_THREADS + 1 ? ncp : 0;   //   column of a class (-1: none); short-line path only (<= 512 classes)
  l.vx = o;     o += 2 * 2 * (CTC_GROUP + 2);
  l.red = o;    o += 64;
  l.dump = o;   o += 2;      // target of masked-off LDS stores (branch-free guards)
  l.words = o;
  return l;
}

// Float-only form of log(exp(d) + 1) (experiment option ctc_float=1; NOT the default): exp as in ctc_limexp
// (<= 2 ulp), the reference's float add of 1, log2 by v_log_f32 (1 ulp) times ln 2 -- 9 float operations instead of ~25
// half-rate f64 operations + a table read.  These are not the reference's roundings: measured against the oracle at T = 200,
// S = 51 (profiles/r06_ctc_float_logadd.txt) the posteriors are within 5.6e-5 absolute / 3.2e-4 relative (exact form: 7.3e-6 /
// 1.1e-4), the reference's known answer within 4.6e-6 either way; the CTC launch of the bench step takes 33.6 instead of 45.2 us.
DEVFN float ctc_softplus_float(float d) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
  const float t = d * L2E_HI;
  const float r = fmaf(d, L2E_LO, fmaf(d, L2E_HI, -t));
  const float e2 = fast_exp2(t);
  const float ef = fmaf(e2 * 0.693147182464599609375f, r, e2);
  const float sf = ef + 1.0f;
  return fast_log2(sf) * 0.693147182464599609375f;
}
template <bool FLA = false>
DEVFN float ctc_log_add(float x, float y, const CrTables tb) {  // tensor.h:86-89
  const float d = x - y;
  float lg;
  if constexpr (FLA) lg = ctc_softplus_float(fminf(fmaxf(d, -10.0f), 10.0f)) + y;
  else lg = cr_softplusf(d, tb) + y;          // log(exp(x-y)+1)+y, every float rounding reproduced (cr_math.h)
  return fabsf(d) > 10.0f ? fmaxf(x, y) : lg; // a select (an asm v_max here turns it into an exec-masked branch)
}
// limexp (tensor.h:78-82) = exp of the argument clamped to [-30, 30], for phase C.  Phase C is OUTSIDE the recursion: an
// error of its exp is not amplified by anything (epath -> per-state totals -> projection), so the correctly rounded double
// evaluation the recursion needs (cr_math.h: ~13 f64 operations at half rate + a table read, 20 calls per thread = 7.7k of the
// kernel's 128k cycles) buys nothing here.  Float evaluation, <= 2 ulp: x log2 e as t + r with the product's rounding error
// and the constant's low part in r, 2^t by v_exp_f32 (1 ulp), first-order correction 2^t (1 + r ln 2) (|r| < 3e-6).
DEVFN float ctc_limexp(float x) {
  const float xc = fminf(fmaxf(x, -30.0f), 30.0f);
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;   // log2 e = HI + LO
  const float t = xc * L2E_HI;
  const float r = fmaf(xc, L2E_LO, fmaf(xc, L2E_HI, -t));
  const float e = fast_exp2(t);
  return fmaf(e * 0.693147182464599609375f, r, e);
}

// Phase B of both paths: alpha into `al`, the reversed-lattice alpha into `be` ([T][S] each).  The match
// scores are prefetched from HBM CTC_PD frames ahead: inside a training step a row this workgroup itself wrote a few
// microseconds earlier takes ~1 us to come back (two frames ahead -- enough for the kernel on its own -- cost the step
// 3.5 us; profiles/r05_ctc_tuning.txt).
// LDS_OUT (S <= 64 only): `al` / `be` are LDS arrays -- the stores of a step then leave the vmcnt queue, whose
// in-order count otherwise makes the wait for a prefetched match row also a wait for the previous stores'
// write acknowledgements.
// LDS_SRC (short-line path, S <= 64, where the carve has the room): the match scores never leave LDS -- lml[t][nup] holds
// them per DISTINCT class, lane j reads column ucol[state]; no rows to HBM in phase A, no store drain in front of the
// recursion, and the first frames arrive in an LDS latency.  (With the scores requested two frames ahead an LDS source had
// measured 100 cycles per frame slower than HBM -- its reads share lgkmcnt with the table reads of log_add; CTC_PD frames
// ahead a score has long arrived when the step that uses it waits for its table entry.)
template <bool LDS_OUT, bool LDS_SRC = false, bool FLA = false>
DEVFN void ctc_lattice(const float* lm, float* al, float* be, float* vx, float* dump, const CrTables tb, const int T,
                       const int S, const float* lml = nullptr, const int nup = 0, const int* ucol = nullptr) {
  // (= forwardbackward(), ctc.cc:42-55); serial in t, parallel over the label axis
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const size_t latbytes = (size_t)T * S * 4;
  const BufF32 lmb = make_buf(lm, latbytes);
  if (S <= 64) {
    // one wave per direction: the j-1 neighbour arrives by a DPP wave shift, no barrier per frame
    __syncthreads();  // lattice rows of phase A visible
    if (wave < 2) {
      const bool rev = wave == 1;
      const BufF32 outb = make_buf(rev ? be : al, LDS_OUT ? 0 : latbytes);
      const int j = lane;
      // lattice cell of (step i, state j): byte offset = lane part + wave-uniform frame part; masked lanes (j >= S) sit at
      // BUF_OOB_BASE, prefetches past the end re-read the end frame.  The frame parts are carried along (one add and a clamp per
      // step; the output pointer of a lane one add): a single wave issues an instruction every four cycles or so whatever its
      // kind, and deriving both from the step number cost a dozen scalar operations and an integer multiply per frame.
      const unsigned lanepart = j < S ? (unsigned)(rev ? S - 1 - j : j) * 4u : BUF_OOB_BASE;
      const int rowbytes = S * 4, lastb = (T - 1) * rowbytes, stepb = rev ? -rowbytes : rowbytes;
      auto clampf = [&](int x) -> int { return x < 0 ? 0 : (x > lastb ? lastb : x); };
      int cf = rev ? lastb : 0;                 // frame of step i
      const int ostep = j < S ? (rev ? -S : S) : 0;
      float* op = j < S ? (rev ? be : al) + (rev ? S - 1 - j : j) + (rev ? (T - 1) * S : 0) : dump;   // LDS_OUT: this lane's cell of step i
      float v = -5.0f * (float)j;            // skip * j, exact in float
      float skipi = 0.0f;                    // skip * i, accumulated: exact while 5 T < 2^24
      float lmq[CTC_PD], kaq[CTC_PD];
      int pf = cf;                              // frame of the next prefetch (CTC_PD steps ahead once the ring is primed)
      // LDS_SRC: this lane's column of the score table, at the frame of the next prefetch (never past the line's ends)
      const float* pp = LDS_SRC ? lml + ucol[j < S ? (rev ? S - 1 - j : j) : 0] + (rev ? (T - 1) * nup : 0) : nullptr;
      const int pstep = rev ? -nup : nup;
#pragma unroll
      for (int q = 0; q < CTC_PD; q++) {
        if (LDS_SRC) {
          lmq[q] = 0.0f;
          if (q < T) { lmq[q] = *pp; pp += pstep; }   // wave-uniform
        } else {
          lmq[q] = buf_load_s(lmb, lanepart, (unsigned)pf); pf = clampf(pf + stepb);
        }
        kaq[q] = 0.0f;
      }
      auto step = [&](float& lmr, float& ka, const bool more) {   // more (LDS_SRC; wave-uniform): the frame CTC_PD ahead exists
        KEEP_ALIVE(ka);
        const float lmv = lmr;
        const float same = v + lmv;
        // next = w + lmatch with w = v[j-1] (lane 0: skip * i), the lane shift folded into the add
        const float next = add_wave_shr1(skipi + lmv, v, lmv);
        skipi -= 5.0f;
        if (LDS_SRC) {
          if (more) { lmr = *pp; pp += pstep; }
        } else {
          lmr = buf_load_s(lmb, lanepart, (unsigned)pf);  // CTC_PD frames ahead
          pf = clampf(pf + stepb);
        }
        v = ctc_log_add<FLA>(same, next, tb);
        if (LDS_OUT) { *op = v; op += ostep; }
        else { buf_store_s(outb, lanepart, (unsigned)cf, v); cf += stepb; }
        ka = v;
      };
      int i = 0;
      for (; i + 2 * CTC_PD <= T; i += CTC_PD) {   // rounds whose every step has a frame CTC_PD ahead
#pragma unroll
        for (int q = 0; q < CTC_PD; q++) step(lmq[q], kaq[q], true);
      }
#pragma unroll
      for (int q = 0; q < 2 * CTC_PD - 1; q++)
        if (i + q < T) step(lmq[q % CTC_PD], kaq[q % CTC_PD], i + q + CTC_PD < T);
    }
  } else if (S <= 128) {
    // 65..128 states (transcripts of 33..63 labels: configs[4]'s 50 labels = 101 states): still ONE wave per direction and no
    // barrier per frame -- lane u holds states 2u and 2u + 1; state 2u + 1 takes its j-1 neighbour from the lane's own other
    // register, state 2u from lane u - 1's odd state by the DPP wave shift.  The two log_adds of a lane are independent, so a
    // frame costs little more than the one-state chain; the 256-lane form below pays a workgroup barrier per frame
    // (profiles/r04_ctc_phase_cycles.txt: phase B at T = 400, S = 101).
    __syncthreads();  // lattice rows of phase A visible
    if (wave < 2) {
      const bool rev = wave == 1;
      const BufF32 outb = make_buf(rev ? be : al, latbytes);
      const int j0 = 2 * lane, j1 = 2 * lane + 1;
      const unsigned lp0 = j0 < S ? (unsigned)(rev ? S - 1 - j0 : j0) * 4u : BUF_OOB_BASE;
      const unsigned lp1 = j1 < S ? (unsigned)(rev ? S - 1 - j1 : j1) * 4u : BUF_OOB_BASE;
      const int rowbytes = S * 4, lastb = (T - 1) * rowbytes, stepb = rev ? -rowbytes : rowbytes;   // (frame parts carried along: see above)
      auto clampf = [&](int x) -> int { return x < 0 ? 0 : (x > lastb ? lastb : x); };
      int cf = rev ? lastb : 0;
      float v0 = -5.0f * (float)j0, v1 = -5.0f * (float)j1;   // skip * j, exact in float
      float skipi = 0.0f;
      float l0q[CTC_PD], l1q[CTC_PD], k0q[CTC_PD], k1q[CTC_PD];
      int pf = cf;
#pragma unroll
      for (int q = 0; q < CTC_PD; q++) {
        l0q[q] = buf_load_s(lmb, lp0, (unsigned)pf); l1q[q] = buf_load_s(lmb, lp1, (unsigned)pf);
        pf = clampf(pf + stepb); k0q[q] = k1q[q] = 0.0f;
      }
      auto step = [&](float& l0, float& l1, float& k0, float& k1) {
        KEEP_ALIVE(k0); KEEP_ALIVE(k1);
        const float m0 = l0, m1 = l1;
        const float same0 = v0 + m0, same1 = v1 + m1;
        const float next1 = v0 + m1;                               // w = v_old[j - 1], the lane's own even state
        const float next0 = add_wave_shr1(skipi + m0, v1, m0);     // ... lane u - 1's odd state (lane 0: skip * i)
        skipi -= 5.0f;
        l0 = buf_load_s(lmb, lp0, (unsigned)pf);                    // CTC_PD frames ahead
        l1 = buf_load_s(lmb, lp1, (unsigned)pf);
        pf = clampf(pf + stepb);
        v0 = ctc_log_add<FLA>(same0, next0, tb);
        v1 = ctc_log_add<FLA>(same1, next1, tb);
        buf_store_s(outb, lp0, (unsigned)cf, v0);
        buf_store_s(outb, lp1, (unsigned)cf, v1);
        cf += stepb;
        k0 = v0; k1 = v1;
      };
      int i = 0;
      for (; i + CTC_PD <= T; i += CTC_PD) {
#pragma unroll
        for (int q = 0; q < CTC_PD; q++) step(l0q[q], l1q[q], k0q[q], k1q[q]);
      }
#pragma unroll
      for (int q = 0; q < CTC_PD - 1; q++)
        if (i + q < T) step(l0q[q], l1q[q], k0q[q], k1q[q]);
    }
  } else {
    const int R = (S + CTC_GROUP - 1) / CTC_GROUP;
    // RM: states a lane can hold (2 for S <= 512, else CTC_RMAX): every frame issues RM prefetches and RM guarded updates
    auto run = [&](auto rm_tag) {
    constexpr int RM = decltype(rm_tag)::value;
    const int grp = wave >> 2, u = tid & (CTC_GROUP - 1);
    float* vxg = vx + grp * 2 * (CTC_GROUP + 2);
    vxg[u] = (float)(-5.0 * (u * R + R - 1));
    __syncthreads();
    const bool rev = grp == 1;
    const BufF32 outb = make_buf(rev ? be : al, latbytes);
    auto loff = [&](int i, int j) -> unsigned {
      if (i >= T || j >= S) return BUF_OOB;
      return (unsigned)(rev ? (size_t)(T - 1 - i) * S + (S - 1 - j) : (size_t)i * S + j) * 4u;
    };
    float v[RM], lmA[RM], lmB[RM], kaA[RM], kaB[RM];
#pragma unroll
    for (int r = 0; r < RM; r++) {
      const int j = u * R + r;
      v[r] = (float)(-5.0 * j);
      lmA[r] = buf_load(lmb, r < R ? loff(0, j) : BUF_OOB);
      lmB[r] = buf_load(lmb, r < R ? loff(1, j) : BUF_OOB);
      kaA[r] = kaB[r] = 0.0f;
    }
    auto step = [&](const int i, float (&lmr)[RM], float (&ka)[RM]) {
#pragma unroll
      for (int r = 0; r < RM; r++) KEEP_ALIVE(ka[r]);
      const float from_prev = vxg[(i & 1) * (CTC_GROUP + 2) + (u > 0 ? u - 1 : 0)];
      float lmv[RM];
#pragma unroll
      for (int r = 0; r < RM; r++) {
        lmv[r] = lmr[r];
        lmr[r] = buf_load(lmb, r < R ? loff(i + 2, u * R + r) : BUF_OOB);  // two frames ahead
      }
#pragma unroll
      for (int r = RM - 1; r >= 0; r--) {
        if (r < R) {
          const int j = u * R + r;
          float w = (r == 0) ? from_prev : v[r - 1];
          if (j == 0) w = (float)(-5.0 * i);
          const float same = v[r] + lmv[r];
          const float next = w + lmv[r];
          v[r] = ctc_log_add<FLA>(same, next, tb);
          buf_store(outb, loff(i, j), v[r]);
          ka[r] = v[r];
        }
      }
      float last = v[0];
#pragma unroll
      for (int r = 1; r < RM; r++)
        if (r == R - 1) last = v[r];
      vxg[((i + 1) & 1) * (CTC_GROUP + 2) + u] = last;
      __syncthreads();
    };
    int i = 0;
    for (; i + 1 < T; i += 2) {
      step(i, lmA, kaA);
      step(i + 1, lmB, kaB);
    }
    if (i < T) step(i, lmA, kaA);
    };
    if (R <= 2) run(std::integral_constant<int, 2>{}); else run(std::integral_constant<int, CTC_RMAX>{});
  }
}

// Phase B for lines of more than CTC_SMAX_LDS states: the same recursion (forward_algorithm, ctc.cc:24-40, on the
// lattice and on its (t,s)-reversed image), every group of 256 lanes walking the label axis in rounds; the previous
// frame's row is read back from the output lattice (system-scope accesses + one barrier per frame: the rows are written
// and read by different waves).  Correctness path, not a tuned one: such transcripts exceed 1023 labels.
template <bool FLA = false>
DEVFN void ctc_lattice_huge(const float* lm, float* al, float* be, const CrTables tb, const int T, const int S) {
  const int tid = threadIdx.x;
  const int grp = tid >> 8, u = tid & (CTC_GROUP - 1);
  const bool rev = grp == 1;
  const size_t latbytes = (size_t)T * S * 4;
  const BufF32 lmb = make_buf(lm, latbytes), outb = make_buf(rev ? be : al, latbytes);
  auto loff = [&](int i, int j) -> unsigned {
    return (unsigned)(rev ? (size_t)(T - 1 - i) * S + (S - 1 - j) : (size_t)i * S + j) * 4u;
  };
  __syncthreads();
  for (int i = 0; i < T; i++) {
    for (int j = u; j < S; j += CTC_GROUP) {
      const float lmv = buf_load(lmb, loff(i, j));
      const float vj = i == 0 ? (float)(-5.0 * j) : buf_load_wt(outb, loff(i - 1, j));
      const float w = j == 0 ? (float)(-5.0 * i) : (i == 0 ? (float)(-5.0 * (j - 1)) : buf_load_wt(outb, loff(i - 1, j - 1)));
      buf_store_wt(outb, loff(i, j), ctc_log_add<FLA>(vj + lmv, w + lmv, tb));
    }
    drain_vmem();
    __syncthreads();
  }
}

#define CTC_STAMP(k) do { if (a.prof && b == 0 && threadIdx.x == 0) a.prof[k] = dev_clock(); } while (0)
constexpr int CTC_TREG = (CTC_TABLE_DOUBLES + CTC_THREADS - 1) / CTC_THREADS;
constexpr int CTC_PREG = 34;    // posteriors per thread held in registers across the state classification
constexpr int CTC_NB = 12;      // items a thread of the short-line path takes per round in the match-score and write-out loops
constexpr int CTC_CLANE = 32;   // frames per wave of the lane = state form of phases C / D (lines of up to 64 states and 256 frames)
constexpr int CTC_CCACHE = 26;  // lattice cells per thread kept in registers between the two passes of phase C (26 x 512 = 13312
                                // cells: lines of up to 261 frames x 51 states -- a ragged OCR minibatch, T ~ U{150..250} -- stay on the short-line path;
                                // at 24 the 10 % of such lines beyond 240 frames took the tiled path and the launch 107 us instead of 75)

// ---- lines whose lattice fits in LDS: phases A..E on one CU, see the header ------------------------------
// Guards are branch-free throughout: reads use a clamped index and a select, masked-off stores go to `dump`,
// global accesses go through buffer descriptors (out-of-range = no-op).  Loops are written as a batch of
// independent reads followed by the arithmetic, so the scheduler can interleave the elements of a batch.
template <bool FLA>
DEVFN void ctc_short_line(const CtcArgs& a, float* lds, const CtcLds& L, const CrTables tb, const int b, float* lm,
                          const int off, const int T, const int S, const double (&treg)[CTC_TREG],
                          const float (&preg)[CTC_PREG], const bool flat) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int nc = a.nc, ncp = a.ncp;
  double* part = reinterpret_cast<double*>(lds + L.part);   // 512 doubles, reused phase by phase
  double* tot = reinterpret_cast<double*>(lds + L.tot);
  // One region (the carve's row buffer + lattice tile + row sums, `cap` words) is re-used phase by phase, so that a line
  // needs max(T (ncp + nup), 2 T S, T (sp + nup)) words, not T (ncp + sp):
  //   A: posteriors [T][ncp] | match scores per distinct class [T][nup]      B: alpha [T][S] | reversed alpha [T][S]
  //   C..E: lattice tile [T][sp] | class columns [T][nup]
  float* rowbuf = lds + L.rowbuf;
  float* etile = rowbuf;
  const int cap = L.asum + a.tile - L.rowbuf;
  int* stl = reinterpret_cast<int*>(lds + L.states);
  int* lists = reinterpret_cast<int*>(lds + L.lists);
  int* ucol = reinterpret_cast<int*>(lds + L.ucol);
  int* ucls = reinterpret_cast<int*>(lds + L.ucls);
  int* ccol = reinterpret_cast<int*>(lds + L.ccol);
  float* red = lds + L.red;
  int* wcnt = reinterpret_cast<int*>(red + 16);
  float* dump = lds + L.dump;
  float* al = lm + (size_t)T * S;
  float* be = al + (size_t)T * S;
  const int TS = T * S, sp = S | 1;
  const size_t latbytes = (size_t)TS * 4;
  const BufF32 pb = make_buf(a.P + (size_t)off * nc, (size_t)T * nc * 4);

  const int npost = T * nc;   // (flat: the posteriors were requested by the caller and land in LDS after the classification)

  // ---- classify the target states: blank / first state of its class / repeat; distinct classes -> columns
  int* firstof = reinterpret_cast<int*>(rowbuf);  // first state of a class; rowbuf holds >= ncp words and is free until
                                                  // the posteriors land (etile can be smaller than nc for a short line)
  for (int c = tid; c < nc; c += CTC_THREADS) firstof[c] = 0x7fffffff;
  __syncthreads();
  const bool live = tid < S;
  const int sc = stl[live ? tid : 0];
  if (live) lds_atomic_min(&firstof[sc], tid);
  __syncthreads();
  const int fo = firstof[sc];
  const bool isblank = live && sc == 0;
  const bool islab = live && sc != 0 && fo == tid, isrep = live && sc != 0 && fo != tid;
  // the distinct classes of the line get the columns 0..nu-1 in CLASS order: a frame's total over the columns
  // is then the reference's sum over all classes (absent classes contribute exact zeros)
  const bool pres = tid < nc && firstof[tid < nc ? tid : 0] != 0x7fffffff;
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long mP = wave_ballot(pres), mB = wave_ballot(isblank), mL = wave_ballot(islab),
                           mR = wave_ballot(isrep);
  if (lane == 0) {
    wcnt[wave * 4 + 0] = __builtin_popcountll(mP); wcnt[wave * 4 + 1] = __builtin_popcountll(mB);
    wcnt[wave * 4 + 2] = __builtin_popcountll(mL); wcnt[wave * 4 + 3] = __builtin_popcountll(mR);
  }
  __syncthreads();
  int bP = 0, bB = 0, bL = 0, bR = 0, nu = 0, nb = 0, nf = 0;
#pragma unroll
  for (int w = 0; w < CTC_THREADS / 64; w++) {
    const int cP = wcnt[w * 4], cB = wcnt[w * 4 + 1], cL = wcnt[w * 4 + 2], cR = wcnt[w * 4 + 3];
    const bool pre = w < wave;
    bP += pre ? cP : 0; bB += pre ? cB : 0; bL += pre ? cL : 0; bR += pre ? cR : 0;
    nu += cP; nb += cB; nf += cL;
  }
  const int nr = S - nb - nf, nup = nu | 1;
  if (tid < nc) {
    const int r = bP + __builtin_popcountll(mP & below);
    ccol[tid] = pres ? r : -1;
    if (pres) ucls[r] = tid;
  }
  if (isblank) lists[bB + __builtin_popcountll(mB & below)] = tid;
  if (islab) lists[nb + bL + __builtin_popcountll(mL & below)] = tid;
  if (isrep) lists[nb + nf + bR + __builtin_popcountll(mR & below)] = tid;
  __syncthreads();
  if (live) ucol[tid] = ccol[sc];
  CTC_STAMP(12);
  // the match scores per distinct class, [T][nup]: behind the posteriors -- and, where the region has the room, behind the two
  // LDS lattices of phase B as well, which then reads them where they are (ctc_lattice: LDS_SRC)
  const bool lds_lat = S <= 64 && 2 * TS <= cap;
  const int lmu_hi = 2 * TS > T * ncp ? 2 * TS : T * ncp;
  const bool lm_lds = lds_lat && lmu_hi + T * nup <= cap;   // uniform per workgroup
  float* lmu = rowbuf + (lm_lds ? lmu_hi : T * ncp);
  // the posteriors the write-out at the very end subtracts (first round of its loop: (frame, present class) items tid,
  // tid + 512, ..): requested HERE, a whole kernel ahead -- inside a training step such a load is a microsecond on the tail
  float pw[CTC_NB];
  {
    const int n = T * nu;
    const int dq = CTC_THREADS / nu, dr = CTC_THREADS - dq * nu;
    int tq = tid / nu, uq = tid - tq * nu;
#pragma unroll
    for (int u = 0; u < CTC_NB; u++) {
      const bool in = tid + u * CTC_THREADS < n;
      pw[u] = buf_load(pb, in ? (unsigned)(tq * nc + ucls[in ? uq : 0]) * 4u : BUF_OOB);
      tq += dq; uq += dr;
      if (uq >= nu) { uq -= nu; tq++; }
    }
  }
  {
    double* tabs = reinterpret_cast<double*>(lds + L.tables);
#pragma unroll
    for (int k = 0; k < CTC_TREG; k++)
      if (tid + k * CTC_THREADS < CTC_TABLE_DOUBLES) tabs[tid + k * CTC_THREADS] = treg[k];
  }

  // ---- A: match scores, once per distinct class:  lmu[t][u] = log(max(1e-5, p_t[c_u]) / sum_c max(1e-5, p_t[c]))
  //         x/sum is (float)((double)x * (1/sum)): equal to the reference's float division except for ~1e-8
  //         of the values (1 ulp of double before the rounding to float)
  if (flat) {
#pragma unroll
    for (int k = 0; k < CTC_PREG; k++) {
      const int i = tid + k * CTC_THREADS;
      float* w = i < npost ? &rowbuf[i] : dump;
      *w = fmaxf(1e-5f, preg[k]);
    }
  } else {
    for (int cb = 0; cb < nc; cb += 64) {   // one wave per frame, lanes over classes, eight frames in flight
      const int c = cb + lane;
      const bool cok = c < nc;
      for (int t0 = wave; t0 < T; t0 += 8 * (CTC_THREADS / 64)) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int t = t0 + u * (CTC_THREADS / 64);
          x[u] = buf_load(pb, (cok && t < T) ? (unsigned)(t * nc + c) * 4u : BUF_OOB);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int t = t0 + u * (CTC_THREADS / 64);
          float* w = (cok && t < T) ? &rowbuf[t * ncp + c] : dump;
          *w = fmaxf(1e-5f, x[u]);
        }
      }
    }
  }
  __syncthreads();
  CTC_STAMP(13);
  for (int t = tid; t < T; t += CTC_THREADS) {  // sequential float sum in class order, as asum1() (tensor.h:337-342)
    // (whole batches carry no guards: with a clamped index and a select per element the loop was 190 instructions per
    //  sixteen classes, issue-bound on the one wave per SIMD that has frames -- 4.7k cycles at nc = 83)
    const float* r = rowbuf + t * ncp;
    float acc = 0.0f;
    int c0 = 0;
    for (; c0 + 16 <= nc; c0 += 16) {
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; u++) x[u] = r[c0 + u];
#pragma unroll
      for (int u = 0; u < 16; u++) acc += x[u];
    }
    if (c0 < nc) {
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; u++) x[u] = r[c0 + u < nc ? c0 + u : 0];
#pragma unroll
      for (int u = 0; u < 16; u++) acc += (c0 + u < nc) ? x[u] : 0.0f;   // + 0.0f is exact
    }
    part[t] = 1.0 / (double)acc;
  }
  __syncthreads();
  CTC_STAMP(14);
  {
    const int n = T * nu;
    const int dq = CTC_THREADS / nu, dr = CTC_THREADS - dq * nu;   // (t, u) of item i, followed incrementally
    int tq = tid / nu, uq = tid - tq * nu;
    for (int i0 = tid; i0 < n; i0 += CTC_NB * CTC_THREADS) {   // (a line of the bench shape: one round)
      float q[CTC_NB];
      float* w[CTC_NB];
#pragma unroll
      for (int u = 0; u < CTC_NB; u++) {
        const bool in = i0 + u * CTC_THREADS < n;
        const int tc = in ? tq : 0, uc = in ? uq : 0;
        q[u] = (float)((double)rowbuf[tc * ncp + ucls[uc]] * part[tc]);
        w[u] = in ? &lmu[tc * nup + uc] : dump;
        tq += dq; uq += dr;
        if (uq >= nu) { uq -= nu; tq++; }
      }
#pragma unroll
      for (int u = 0; u < CTC_NB; u++) *w[u] = cr_logf(q[u], tb);
    }
  }
  __syncthreads();
  CTC_STAMP(15);
  if (!lm_lds) {  // lmatch rows for the recursion (it prefetches them from HBM: vmcnt, not lgkmcnt)
    const BufF32 lmw = make_buf(lm, latbytes);
    for (int s0 = 0; s0 < S; s0 += 64) {   // one wave per frame, lanes over states
      const int st = s0 + lane;
      const bool sok = st < S;
      const float* col = lmu + ucol[sok ? st : 0];
      for (int t0 = wave; t0 < T; t0 += 8 * (CTC_THREADS / 64)) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int t = t0 + u * (CTC_THREADS / 64); x[u] = col[(t < T ? t : T - 1) * nup]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int t = t0 + u * (CTC_THREADS / 64);
          buf_store(lmw, (sok && t < T) ? (unsigned)(t * S + st) * 4u : BUF_OOB, x[u]);
        }
      }
    }
  }
  CTC_STAMP(1);

  // ---- B (S <= 64: waves 0 and 1).  The other waves meanwhile emit the part of the result that does not
  //         depend on the lattice: a class without a target state has aligned = 0, delta = -p.
  float* all = rowbuf;            // alpha / reversed alpha in LDS (row buffer + lattice tile are both free here)
  float* bel = rowbuf + TS;
  if (lm_lds) ctc_lattice<true, true, FLA>(lm, all, bel, lds + L.vx, dump, tb, T, S, lmu, nup, ucol);
  else if (lds_lat) ctc_lattice<true, false, FLA>(lm, all, bel, lds + L.vx, dump, tb, T, S);
  else ctc_lattice<false, false, FLA>(lm, al, be, lds + L.vx, dump, tb, T, S);
  const BufF32 dzb = make_buf(a.Dz + (size_t)off * nc, (size_t)T * nc * 4);
  const BufF32 agb = make_buf(a.aligned ? a.aligned + (size_t)off * nc : a.Dz, a.aligned ? (size_t)T * nc * 4 : 0);
  if (S > 64 || (wave & 2)) {   // waves 2, 3, 6, 7: not on the SIMDs of the two lattice waves (w mod 4 = 0, 1)
    const int w0 = S > 64 ? wave : (wave & 1) + ((wave >> 2) << 1), nw = S > 64 ? CTC_THREADS / 64 : 4;
    for (int cb = 0; cb < nc; cb += 64) {   // one wave per frame, lanes over classes
      const int c = cb + lane;
      const bool absent = c < nc && ccol[c < nc ? c : 0] < 0;
      for (int t0 = w0; t0 < T; t0 += 8 * nw) {
        float x[8];
        unsigned ofs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int t = t0 + u * nw;
          ofs[u] = (absent && t < T) ? (unsigned)(t * nc + c) * 4u : BUF_OOB;
          x[u] = buf_load(pb, ofs[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          buf_store(agb, ofs[u], 0.0f);
          buf_store(dzb, ofs[u], 0.0f - x[u]);   // (float)0 - p, as aligned - p with aligned = 0
        }
      }
    }
  }
  __syncthreads();
  CTC_STAMP(2);

  // Lines of up to 64 states and 8 x CTC_CLANE frames walk phases C and D with lane = state, wave w on the frames w, w + 8, ..:
  // no index arithmetic, no per-element guards (frames in wave-uniform groups of four), the cells stay in registers from
  // the load to the normalised store, and the per-state sums over a wave's frames ARE the time chunks of the generic phase D
  // (thread q of a state took the frames q, q + 8, ..; Q = 8 for S <= 64), so every sum has the operands and the order it had.
  // The tile then holds the NORMALISED cells (float)((double)e * 1/total), which phase E otherwise forms on the fly.
  const bool lanemap = S <= 64 && T <= (CTC_THREADS / 64) * CTC_CLANE;   // uniform per workgroup
  if (lanemap) {
    const BufF32 alb = make_buf(al, latbytes), beb = make_buf(be, latbytes);
    const bool sok = lane < S;
    const int sl = sok ? lane : 0;
    float bo[CTC_CLANE];
    float mx = -3.0e38f;
    // (the source is chosen by a BRANCH around the whole loop: as a select per element, `lds_lat ? LDS : buffer load`, the
    //  compiler issued both and every group of cells waited for global loads it did not need -- 7.0k cycles against 1.6k)
    auto load_cells = [&](auto lds_tag) {
      constexpr bool LL = decltype(lds_tag)::value;
#pragma unroll
      for (int g = 0; g < CTC_CLANE / 4; g++) {
        if (wave + 32 * g < T) {   // wave-uniform
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int k = 4 * g + u, t = wave + 8 * k;
            const bool in = sok && t < T;
            const int ic = (t < T ? t : 0) * S + sl;
            float x;
            if constexpr (LL) x = all[ic] + bel[ic];
            else {
              const unsigned io = in ? (unsigned)ic * 4u : BUF_OOB;
              x = buf_load(alb, io) + buf_load(beb, io);
            }
            bo[k] = in ? x : -3.0e38f;
            mx = fmaxf(mx, bo[k]);
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4; u++) bo[4 * g + u] = -3.0e38f;
        }
      }
    };
    if (lds_lat) load_cells(std::true_type{}); else load_cells(std::false_type{});
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();   // (also: every alpha / beta cell is in registers -- the tile may overwrite them)
    CTC_STAMP(6);
    mx = red[0];
#pragma unroll
    for (int i = 1; i < CTC_THREADS / 64; i++) mx = fmaxf(mx, red[i]);
    CTC_STAMP(7);
    double acc = 0.0;
#pragma unroll
    for (int g = 0; g < CTC_CLANE / 4; g++) {
      if (wave + 32 * g < T) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = 4 * g + u, t = wave + 8 * k;
          const float e = ctc_limexp(bo[k] - mx);
          bo[k] = e;
          acc += (t < T) ? (double)e : 0.0;   // increasing t; + 0.0 is exact
        }
      }
    }
    CTC_STAMP(3);
    if (sok) part[wave * S + lane] = acc;
    __syncthreads();
    CTC_STAMP(8);
    if (tid < S) {
      double sum = 0.0;
      for (int k = 0; k < CTC_THREADS / 64; k++) sum += part[k * S + tid];
      tot[tid] = 1.0 / fmax(1e-9, sum);
    }
    __syncthreads();
    CTC_STAMP(9);
    const double it = tot[sl];
#pragma unroll
    for (int g = 0; g < CTC_CLANE / 4; g++) {
      if (wave + 32 * g < T) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int k = 4 * g + u, t = wave + 8 * k;
          float* w = (sok && t < T) ? &etile[t * sp + lane] : dump;
          *w = (float)((double)bo[k] * it);
        }
      }
    }
    __syncthreads();
    CTC_STAMP(4);
  } else {
  // ---- C: epath = limexp(both - amax2(both)) -> etile[t][s]   (ctc.cc:82)
  {
    const BufF32 alb = make_buf(al, latbytes), beb = make_buf(be, latbytes);
    float bo[CTC_CCACHE];
    float mx = -3.0e38f;
    if (lds_lat) {   // (a branch around the loop, not a select per cell: see load_cells above)
#pragma unroll
      for (int k = 0; k < CTC_CCACHE; k++) {
        const int i = tid + k * CTC_THREADS;
        const int ic = i < TS ? i : 0;
        bo[k] = i < TS ? all[ic] + bel[ic] : -3.0e38f;
        mx = fmaxf(mx, bo[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < CTC_CCACHE; k++) {
        const int i = tid + k * CTC_THREADS;
        const float x = buf_load(alb, (unsigned)i * 4u) + buf_load(beb, (unsigned)i * 4u);
        bo[k] = i < TS ? x : -3.0e38f;
        mx = fmaxf(mx, bo[k]);
      }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    CTC_STAMP(6);
    mx = red[0];
#pragma unroll
    for (int i = 1; i < CTC_THREADS / 64; i++) mx = fmaxf(mx, red[i]);
    CTC_STAMP(7);
    const int dq = CTC_THREADS / S, dr = CTC_THREADS - dq * S;   // (t, s) of cell i, followed incrementally
    int tq = tid / S, sq = tid - tq * S;
#pragma unroll
    for (int k = 0; k < CTC_CCACHE; k++) {
      // limexp (tensor.h:78-82): the clamp to [-30, 30] followed by the exp gives its three cases
      float* w = (tid + k * CTC_THREADS < TS) ? &etile[tq * sp + sq] : dump;
      *w = ctc_limexp(bo[k] - mx);
      tq += dq; sq += dr;
      if (sq >= S) { sq -= S; tq++; }
    }
  }
  __syncthreads();
  CTC_STAMP(3);

  // ---- D: per-state totals over time (double accumulator, floor 1e-9) -> tot[s] = 1/total;
  //         x/total is (float)((double)x * (1/total)), see the note at phase D of the tiled path
  {
    const int Q = CTC_THREADS / S > 8 ? 8 : CTC_THREADS / S;   // time chunks per state
    const int q = tid / S, st = tid - q * S;
    const bool act = q < Q;
    double acc = 0.0;
    int t0 = act ? q : T;
    for (; t0 + 7 * Q < T; t0 += 8 * Q) {   // whole batches: no guards
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = etile[(t0 + u * Q) * sp + st];
#pragma unroll
      for (int u = 0; u < 8; u++) acc += (double)x[u];
    }
    if (t0 < T) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int t = t0 + u * Q; x[u] = etile[(t < T ? t : T - 1) * sp + st]; }
#pragma unroll
      for (int u = 0; u < 8; u++) acc += (t0 + u * Q < T) ? (double)x[u] : 0.0;
    }
    if (act) part[q * S + st] = acc;
    __syncthreads();
    CTC_STAMP(8);
    if (tid < S) {
      double sum = 0.0;
      for (int k = 0; k < Q; k++) sum += part[k * S + tid];
      tot[tid] = 1.0 / fmax(1e-9, sum);   // reciprocal: the divisions become double multiplies
    }
    __syncthreads();
    CTC_STAMP(9);
    // (the normalisation itself is applied where phase E consumes the cells: one LDS pass less)
  }
  CTC_STAMP(4);
  }

  // ---- E: project states onto classes (ctc.cc:91-109), compact: one column per class that has a state.
  //         Waves 4-7: lane = one first-occurrence label state, looping over frames (plain stores).
  //         Waves 0-3: lane = one frame, the blank states summed in state order in double (class 0 collects
  //         L+1 states: the reference's double accumulator).
  float* rowc = rowbuf + T * sp;   // [T][nup], behind the lattice tile
  auto project = [&](auto pre_tag) {   // PRE: the tile holds the normalised cells (lane = state form of phases C / D)
    constexpr bool PRE = decltype(pre_tag)::value;
    if (wave >= 4) {
      for (int f0 = 0; f0 < nf; f0 += 64) {
        const bool fok = f0 + lane < nf;
        const int st = lists[nb + (fok ? f0 + lane : 0)];
        const int col = ucol[st];
        const double it = PRE ? 1.0 : tot[st];   // per-state normalisation (phase D), applied on the fly unless the tile holds it
        int t0 = wave - 4;
        for (; t0 + 28 < T; t0 += 4 * 8) {   // whole batches: only the lane guard
          float x[8];
#pragma unroll
          for (int u = 0; u < 8; u++) x[u] = etile[(t0 + 4 * u) * sp + st];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            float* w = fok ? &rowc[(t0 + 4 * u) * nup + col] : dump;
            *w = PRE ? x[u] : (float)((double)x[u] * it);
          }
        }
        if (t0 < T) {
          float x[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { const int t = t0 + 4 * u; x[u] = etile[(t < T ? t : T - 1) * sp + st]; }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int t = t0 + 4 * u;
            float* w = (fok && t < T) ? &rowc[t * nup + col] : dump;
            *w = PRE ? x[u] : (float)((double)x[u] * it);
          }
        }
      }
    } else {
      for (int t = tid; t < T; t += 256) {
        const float* e = etile + t * sp;
        double blank = 0.0;
        int i0 = 0;
        for (; i0 + 8 <= nb; i0 += 8) {   // whole batches: no guards
          float x[8];
          double it[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int st = lists[i0 + u];
            x[u] = e[st];
            it[u] = PRE ? 1.0 : tot[st];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) blank += PRE ? (double)x[u] : (double)(float)((double)x[u] * it[u]);
        }
        if (i0 < nb) {
          float x[8];
          double it[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int st = lists[i0 + u < nb ? i0 + u : 0];
            x[u] = e[st];
            it[u] = PRE ? 1.0 : tot[st];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) blank += (i0 + u < nb) ? (PRE ? (double)x[u] : (double)(float)((double)x[u] * it[u])) : 0.0;
        }
        part[t] = blank;
      }
    }
  };
  if (lanemap) project(std::true_type{}); else project(std::false_type{});
  __syncthreads();
  CTC_STAMP(10);
  for (int t = tid; t < T; t += CTC_THREADS) {
    float* row = rowc + t * nup;
    const float* e = etile + t * sp;
    for (int i = 0; i < nr; i++) {   // few; a class may repeat more than once: read-modify-write in state order
      const int st = lists[nb + nf + i];
      row[ucol[st]] += lanemap ? e[st] : (float)((double)e[st] * tot[st]);
    }
    if (nb > 0) row[ccol[0]] = (float)part[t];
    double total = 0.0;   // columns are in class order: the reference's double sum, minus its exact zeros
    int u0 = 0;
    for (; u0 + 8 <= nu; u0 += 8) {   // whole batches: no guards
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = row[u0 + u];
#pragma unroll
      for (int u = 0; u < 8; u++) total += (double)x[u];
    }
    if (u0 < nu) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = row[u0 + u < nu ? u0 + u : 0];
#pragma unroll
      for (int u = 0; u < 8; u++) total += (u0 + u < nu) ? (double)x[u] : 0.0;
    }
    part[t] = 1.0 / fmax(total, 1e-9);
  }
  __syncthreads();
  CTC_STAMP(11);
  {  // per-frame normalisation and the fused delta  d = aligned - p  (clstmhl.h:211-212) for the present classes
    const int n = T * nu;
    const int dq = CTC_THREADS / nu, dr = CTC_THREADS - dq * nu;   // (t, u) of item i, followed incrementally
    int tq = tid / nu, uq = tid - tq * nu;
    for (int i0 = tid; i0 < n; i0 += CTC_NB * CTC_THREADS) {   // (every posterior of the thread requested before the first is used)
      float p[CTC_NB], av[CTC_NB];
      unsigned ofs[CTC_NB];
#pragma unroll
      for (int u = 0; u < CTC_NB; u++) {
        const bool in = i0 + u * CTC_THREADS < n;
        const int tc = in ? tq : 0, uc = in ? uq : 0;
        ofs[u] = in ? (unsigned)(tc * nc + ucls[uc]) * 4u : BUF_OOB;
        av[u] = (float)((double)rowc[tc * nup + uc] * part[tc]);
        tq += dq; uq += dr;
        if (uq >= nu) { uq -= nu; tq++; }
      }
      if (i0 == tid) {   // (uniform: every thread is in its first round together)
#pragma unroll
        for (int u = 0; u < CTC_NB; u++) p[u] = pw[u];
      } else {
#pragma unroll
        for (int u = 0; u < CTC_NB; u++) p[u] = buf_load(pb, ofs[u]);
      }
#pragma unroll
      for (int u = 0; u < CTC_NB; u++) {
        buf_store(agb, ofs[u], av[u]);
        buf_store(dzb, ofs[u], av[u] - p[u]);
      }
    }
  }
  CTC_STAMP(5);
}

// FLA: the experiment option ctc_float as its own instantiation -- the default kernel is the code it was without the option
template <bool FLA>
__global__ __launch_bounds__(CTC_THREADS) void ctc_align_kernel(CtcArgs a) {
  float* lds = dyn_smem<float>();
  const CtcLds L = ctc_lds_layout(a.tile, a.ncp, a.smax);
  double* tabs = reinterpret_cast<double*>(lds + L.tables);
  double* part = reinterpret_cast<double*>(lds + L.part);
  double* tot = reinterpret_cast<double*>(lds + L.tot);
  float* rowbuf = lds + L.rowbuf;
  float* etile = lds + L.etile;
  float* asum = lds + L.asum;
  int* stl = reinterpret_cast<int*>(lds + L.states);
  float* vx = lds + L.vx;
  float* red = lds + L.red;
  const CrTables tb{tabs, tabs + 32, tabs + 96, tabs + 160};
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const CtcLine ln = a.lines[blockIdx.x];
  const int b = ln.b;
  const int nc = a.nc, ncp = a.ncp, TT = a.tile;
  const int off = ln.off, T = ln.T;
  const int soff = ln.soff, S = ln.S;
  if (T <= 0 || S <= 0) return;
  const float* P = a.P + (size_t)off * nc;
  float* Dz = a.Dz + (size_t)off * nc;
  float* lm = a.lat + ln.lat_off;
  float* al = lm + (size_t)T * S;
  float* be = al + (size_t)T * S;
  CTC_STAMP(0);
  // Request order = completion order (vmcnt counts in order): the target states first (needed at once), then the
  // tables and -- short lines -- the posteriors, whose latency overlaps the state classification.
  // (the short-line path classifies ONE target state per thread: transcripts of more than 255 labels take the tiled path)
  // (the region the short-line path re-uses phase by phase must hold the line: always true for T <= tile)
  const int sl_cap = L.asum + TT - L.rowbuf, sl_nub = ((S + 1) / 2 < nc ? (S + 1) / 2 : nc) | 1;   // nub: bound of its class columns
  const bool short_line = T <= CTC_THREADS && (long)T * (ncp + sl_nub) <= sl_cap && (long)T * ((S | 1) + sl_nub) <= sl_cap &&
                          S <= CTC_THREADS && T * S <= CTC_THREADS * CTC_CCACHE && nc <= CTC_THREADS;   // uniform per workgroup
  const int st0 = a.states[soff + (tid < S ? tid : 0)];
  double treg[CTC_TREG];
#pragma unroll
  for (int k = 0; k < CTC_TREG; k++) {
    const int i = tid + k * CTC_THREADS;
    treg[k] = a.tables[i < CTC_TABLE_DOUBLES ? i : CTC_TABLE_DOUBLES - 1];
  }
  if (short_line) {
    const BufF32 pb = make_buf(P, (size_t)T * nc * 4);
    const bool flat = ncp == nc && T * nc <= CTC_PREG * CTC_THREADS;   // rows back to back in LDS too: flat copy
    float preg[CTC_PREG];
    if (flat) {
#pragma unroll
      for (int k = 0; k < CTC_PREG; k++) preg[k] = buf_load(pb, (unsigned)(tid + k * CTC_THREADS) * 4u);   // past the end: 0
    }
    if (tid < S) stl[tid] = st0;
    for (int s = tid + CTC_THREADS; s < S; s += CTC_THREADS) stl[s] = a.states[soff + s];
    __syncthreads();
    ctc_short_line<FLA>(a, lds, L, tb, b, lm, off, T, S, treg, preg, flat);
    return;
  }
  // lines of more than CTC_SMAX_LDS states: the per-state arrays do not fit the LDS carve -- target states are read from
  // memory, the per-state totals live behind the line's lattice, the projection reads the lattice directly
  const bool huge = S > CTC_SMAX_LDS;
  const int* stg = a.states + soff;
  double* totg = reinterpret_cast<double*>(lm + (((size_t)3 * T * S + 1) & ~(size_t)1));   // (8-byte aligned: lat_off is even)
  if (!huge) {
    if (tid < S) stl[tid] = st0;
    for (int s = tid + CTC_THREADS; s < S; s += CTC_THREADS) stl[s] = a.states[soff + s];
  }
  auto state_at = [&](int s) -> int { return huge ? stg[s] : stl[s]; };
  // blank states | label states, each in state order (phase E walks them on two threads per frame, branch-free)
  int* lists = reinterpret_cast<int*>(lds + L.lists);
  int nblank = 0;
  if (!huge) {
    __syncthreads();
    int* cnt = reinterpret_cast<int*>(red + 32);
    if (wave == 0) {   // one wave, rounds of 64 states: ballot + popcount ranks
      int nb = 0;
      for (int s0 = 0; s0 < S; s0 += 64) nb += __builtin_popcountll(wave_ballot(s0 + lane < S && stl[s0 + lane < S ? s0 + lane : 0] == 0));
      int ib = 0, il = nb;
      const unsigned long long below = (1ull << lane) - 1ull;
      for (int s0 = 0; s0 < S; s0 += 64) {
        const bool in = s0 + lane < S;
        const bool isb = in && stl[in ? s0 + lane : 0] == 0;
        const unsigned long long mb = wave_ballot(isb), ml = wave_ballot(in && !isb);
        if (isb) lists[ib + __builtin_popcountll(mb & below)] = s0 + lane;
        if (in && !isb) lists[il + __builtin_popcountll(ml & below)] = s0 + lane;
        ib += __builtin_popcountll(mb); il += __builtin_popcountll(ml);
      }
      if (lane == 0) cnt[0] = nb;
    }
    __syncthreads();
    nblank = cnt[0];
  }
#pragma unroll
  for (int k = 0; k < CTC_TREG; k++)
    if (tid + k * CTC_THREADS < CTC_TABLE_DOUBLES) tabs[tid + k * CTC_THREADS] = treg[k];

  // ---- A: match scores  lmatch[t][s] = log(max(1e-5,p_t[class_s]) / sum_c max(1e-5,p_t[c])) ------
  for (int t0 = 0; t0 < T; t0 += TT) {
    const int nt = (T - t0) < TT ? (T - t0) : TT;
    __syncthreads();
    if (ncp == nc) {  // rows are back to back in LDS too: flat coalesced copy, CTC_MLP loads in flight
      const float* src = P + (size_t)t0 * nc;
      const int n = nt * nc;
      for (int i0 = tid; i0 < n; i0 += CTC_MLP * CTC_THREADS) {
        float x[CTC_MLP];
#pragma unroll
        for (int u = 0; u < CTC_MLP; u++) x[u] = (i0 + u * CTC_THREADS < n) ? src[i0 + u * CTC_THREADS] : 0.0f;
#pragma unroll
        for (int u = 0; u < CTC_MLP; u++)
          if (i0 + u * CTC_THREADS < n) rowbuf[i0 + u * CTC_THREADS] = fmaxf(1e-5f, x[u]);
      }
    } else {
      for (int i = tid; i < nt * nc; i += CTC_THREADS) {
        const int t = i / nc, c = i - t * nc;
        rowbuf[t * ncp + c] = fmaxf(1e-5f, P[(size_t)t0 * nc + i]);
      }
    }
    __syncthreads();
    if (tid < nt) {  // sequential float sum in class order, as asum1() (tensor.h:337-342)
      float acc = 0.0f;
      const float* r = rowbuf + tid * ncp;
      for (int c = 0; c < nc; c++) acc += r[c];
      asum[tid] = acc;
    }
    __syncthreads();
    // (frame, state) pairs flattened over all threads (states need not be a multiple of the wave); the
    // pair of the next element follows incrementally (no integer division in the loop).  (Measured and not kept, round 4: one
    // log per frame for all blank states + one per label state -- half the logs -- left the phase at 78k -> 81k cycles at the
    // configs[4] lattice: the logs are not what bounds it.)
    {
      const int dq = CTC_THREADS / S, dr = CTC_THREADS - dq * S;
      int tq = tid / S, sq = tid - tq * S;
      for (int i0 = tid; i0 < nt * S; i0 += 4 * CTC_THREADS) {
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool in = i0 + u * CTC_THREADS < nt * S;
          o[u] = in ? rowbuf[tq * ncp + state_at(sq)] / asum[tq] : 1.0f;
          tq += dq; sq += dr;
          if (sq >= S) { sq -= S; tq++; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + u * CTC_THREADS;
          const float l = cr_logf(o[u], tb);
          if (i < nt * S) lm[(size_t)t0 * S + i] = l;
        }
      }
    }
  }
  CTC_STAMP(1);

  // ---- B: forward recursion and the same recursion on the (t,s)-reversed lattice
  if (huge) ctc_lattice_huge<FLA>(lm, al, be, tb, T, S);
  else ctc_lattice<false, false, FLA>(lm, al, be, vx, nullptr, tb, T, S);
  __syncthreads();
  CTC_STAMP(2);

  // ---- C: epath = limexp(both - amax2(both)) ---------------------------------------------
  const int TS = T * S;
  constexpr int CCACHE = CTC_CCACHE;
  if (TS <= CTC_THREADS * CCACHE) {
    float bo[CCACHE];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < CCACHE; k++) {
      const int i = tid + k * CTC_THREADS;
      bo[k] = i < TS ? al[i] + be[i] : -3.0e38f;
      mx = fmaxf(mx, bo[k]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int i = 1; i < CTC_THREADS / 64; i++) mx = fmaxf(mx, red[i]);
#pragma unroll
    for (int k = 0; k < CCACHE; k++) {
      const int i = tid + k * CTC_THREADS;
      if (i < TS) al[i] = ctc_limexp(bo[k] - mx);
    }
  } else {
    // (CTC_MLP cells per thread in flight: as a plain loop each cell was a dependent round trip to the L2 -- 70k of the
    // kernel's 590k cycles at the configs[4] lattice, 400 x 101)
    const BufF32 alb = make_buf(al, (size_t)TS * 4), beb = make_buf(be, (size_t)TS * 4);
    float mx = -3.0e38f;
    for (int i0 = tid; i0 < TS; i0 += CTC_MLP * CTC_THREADS) {
      float x[CTC_MLP], y[CTC_MLP];
#pragma unroll
      for (int u = 0; u < CTC_MLP; u++) {
        const int i = i0 + u * CTC_THREADS;
        x[u] = buf_load(alb, i < TS ? (unsigned)i * 4u : BUF_OOB);
        y[u] = buf_load(beb, i < TS ? (unsigned)i * 4u : BUF_OOB);
      }
#pragma unroll
      for (int u = 0; u < CTC_MLP; u++) mx = fmaxf(mx, i0 + u * CTC_THREADS < TS ? x[u] + y[u] : -3.0e38f);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int i = 1; i < CTC_THREADS / 64; i++) mx = fmaxf(mx, red[i]);
    for (int i0 = tid; i0 < TS; i0 += CTC_MLP * CTC_THREADS) {
      float x[CTC_MLP], y[CTC_MLP];
#pragma unroll
      for (int u = 0; u < CTC_MLP; u++) {
        const int i = i0 + u * CTC_THREADS;
        x[u] = buf_load(alb, i < TS ? (unsigned)i * 4u : BUF_OOB);
        y[u] = buf_load(beb, i < TS ? (unsigned)i * 4u : BUF_OOB);
      }
#pragma unroll
      for (int u = 0; u < CTC_MLP; u++) {
        const int i = i0 + u * CTC_THREADS;
        buf_store(alb, i < TS ? (unsigned)i * 4u : BUF_OOB, ctc_limexp((x[u] + y[u]) - mx));
      }
    }
  }
  __syncthreads();
  CTC_STAMP(3);

  // ---- D: per-state totals over time (double accumulator, floor 1e-9); x/total is evaluated as
  //         (float)((double)x * (1/total)): equal to the reference's (Float)(x/total) except for
  //         ~1e-8 of the values (1 ulp of double before the rounding to float) -----------------------
  if (S > CTC_THREADS) {   // long transcripts (more states than threads): one thread per state, states in rounds
    double* td = huge ? totg : tot;
    for (int s = tid; s < S; s += CTC_THREADS) {
      double acc = 0.0;
      for (int t = 0; t < T; t++) acc += (double)al[(size_t)t * S + s];
      td[s] = 1.0 / fmax(1e-9, acc);
    }
    __syncthreads();
  } else {
    const int Q = CTC_THREADS / S > 8 ? 8 : CTC_THREADS / S;  // time chunks per state (S <= 512)
    if (tid < S * Q) {
      const int s = tid % S, q = tid / S;
      const BufF32 alb = make_buf(al, (size_t)TS * 4);
      double acc = 0.0;
      for (int t0 = q; t0 < T; t0 += CTC_MLP * Q) {   // (same order of additions; the loads of a batch in flight together)
        float x[CTC_MLP];
#pragma unroll
        for (int u = 0; u < CTC_MLP; u++) { const int t = t0 + u * Q; x[u] = buf_load(alb, t < T ? (unsigned)(t * S + s) * 4u : BUF_OOB); }
#pragma unroll
        for (int u = 0; u < CTC_MLP; u++) acc += (t0 + u * Q < T) ? (double)x[u] : 0.0;
      }
      part[q * S + s] = acc;
    }
    __syncthreads();
    if (tid < S) {
      double acc = 0.0;
      for (int q = 0; q < Q; q++) acc += part[q * S + tid];
      tot[tid] = 1.0 / fmax(1e-9, acc);  // reciprocal: the divisions below become double multiplies
    }
    __syncthreads();
  }
  CTC_STAMP(4);

  // ---- E: per-state normalisation applied on the fly; project states onto classes, normalise
  //         per frame, emit deltas ---------------------------------------------------------------
  const int sp = S | 1;
  // (diagnostics: cycles of phase E's parts summed over the tiles, slots 6..9 -- scripts/gpu_ctcprof.py)
#define CTC_ACC(k) do { if (a.prof && b == 0 && threadIdx.x == 0) { const long long now_ = dev_clock(); a.prof[k] += now_ - eprev; eprev = now_; } } while (0)
  long long eprev = 0;
  if (a.prof && b == 0 && threadIdx.x == 0) { eprev = dev_clock(); a.prof[6] = a.prof[7] = a.prof[8] = a.prof[9] = 0; }
  for (int t0 = 0; t0 < T; t0 += TT) {
    const int nt = (T - t0) < TT ? (T - t0) : TT;
    for (int s0 = lane; s0 < S && !huge; s0 += 64) {  // coalesced staging of the lattice tile, CTC_MLP frames in flight
      const double it = tot[s0];
      const BufF32 alb = make_buf(al + (size_t)t0 * S, (size_t)nt * S * 4);
      for (int t = wave; t < nt; t += CTC_MLPT * (CTC_THREADS / 64)) {
        float x[CTC_MLPT];
#pragma unroll
        for (int u = 0; u < CTC_MLPT; u++) {
          const int tu = t + u * (CTC_THREADS / 64);
          x[u] = buf_load(alb, tu < nt ? (unsigned)(tu * S + s0) * 4u : BUF_OOB);
        }
#pragma unroll
        for (int u = 0; u < CTC_MLPT; u++) {
          const int tu = t + u * (CTC_THREADS / 64);
          float* w = tu < nt ? &etile[tu * sp + s0] : lds + L.dump;
          *w = (float)((double)x[u] * it);
        }
      }
    }
    for (int i = tid; i < nt * ncp; i += CTC_THREADS) rowbuf[i] = 0.0f;
    __syncthreads();
    CTC_ACC(6);   // lattice tile staged, class rows cleared
    if (huge) {
      if (tid < nt) {
        float* row = rowbuf + tid * ncp;
        double blank = 0.0;  // class 0 collects L+1 states: keep the reference's double accumulator
        const float* ag = al + (size_t)(t0 + tid) * S;
        for (int s0 = 0; s0 < S; s0++) {
          const int c = state_at(s0);
          const float x = (float)((double)ag[s0] * totg[s0]);
          if (c == 0) blank += (double)x;
          else row[c] += x;   // (ds_add_f32 instead was measured 40 % slower for this phase)
        }
        part[tid] = blank;
      }
    } else {
      // Two threads per frame, on different waves (a tile holds <= 256 frames): thread f sums the blank states in state order
      // in double (no memory dependence between its reads), thread 256 + f walks the label states and adds each into its class
      // column -- a read-modify-write chain through LDS that used to run behind the blank sum on ONE thread with 2/3 of the
      // workgroup idle (158k of the kernel's 590k cycles at the configs[4] lattice).  Classes and cells of a batch of 8 states
      // are read before the batch's additions.
      const int f = tid & 255;
      if (f < nt) {
        const float* e = etile + f * sp;
        if (tid < 256) {
          double blank = 0.0;
          for (int i0 = 0; i0 < nblank; i0 += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = e[lists[i0 + u < nblank ? i0 + u : 0]];
#pragma unroll
            for (int u = 0; u < 8; u++) blank += (i0 + u < nblank) ? (double)x[u] : 0.0;   // (+ 0.0 is exact)
          }
          part[f] = blank;
        } else {
          float* row = rowbuf + f * ncp;
          float* const sink = lds + L.dump;
          for (int i0 = nblank; i0 < S; i0 += 8) {
            float x[8];
            float* w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const bool in = i0 + u < S;
              const int st = lists[in ? i0 + u : nblank];
              x[u] = e[st];
              w[u] = in ? &row[stl[st]] : sink;   // branch-free: masked-off additions go to the dump word
            }
#pragma unroll
            for (int u = 0; u < 8; u++) *w[u] += x[u];   // state order within a class (ds_add_f32 was measured 40 % slower)
          }
        }
      }
    }
    __syncthreads();
    CTC_ACC(7);   // states projected onto classes
    if (tid < nt) {
      float* row = rowbuf + tid * ncp;
      row[0] = (float)part[tid];
      double total = 0.0;
      for (int c0 = 0; c0 < nc; c0 += 8) {   // class order, as the reference's double sum
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = row[c0 + u < nc ? c0 + u : 0];
#pragma unroll
        for (int u = 0; u < 8; u++) total += (c0 + u < nc) ? (double)x[u] : 0.0;
      }
      part[tid] = 1.0 / fmax(total, 1e-9);
    }
    __syncthreads();
    CTC_ACC(8);   // frame totals
    for (int c = lane; c < nc; c += 64) {  // coalesced write-out, one wave per frame, CTC_MLP frames in flight
      const BufF32 pb = make_buf(P + (size_t)t0 * nc, (size_t)nt * nc * 4), db = make_buf(Dz + (size_t)t0 * nc, (size_t)nt * nc * 4);
      const BufF32 ab = make_buf(a.aligned ? a.aligned + ((size_t)off + t0) * nc : Dz, a.aligned ? (size_t)nt * nc * 4 : 0);
      for (int t = wave; t < nt; t += CTC_MLPT * (CTC_THREADS / 64)) {
        float p[CTC_MLPT];
        unsigned ofs[CTC_MLPT];
#pragma unroll
        for (int u = 0; u < CTC_MLPT; u++) {
          const int tu = t + u * (CTC_THREADS / 64);
          ofs[u] = tu < nt ? (unsigned)(tu * nc + c) * 4u : BUF_OOB;
          p[u] = buf_load(pb, ofs[u]);
        }
#pragma unroll
        for (int u = 0; u < CTC_MLPT; u++) {
          const int tu = t + u * (CTC_THREADS / 64);
          const int tc = tu < nt ? tu : 0;
          const float av = (float)((double)rowbuf[tc * ncp + c] * part[tc]);
          buf_store(ab, ofs[u], av);
          buf_store(db, ofs[u], av - p[u]);
        }
      }
    }
    __syncthreads();
    CTC_ACC(9);   // aligned / deltas written
  }
  CTC_STAMP(5);
}

// argmax per frame, ties -> last index (tensor.h:357-366)
__global__ __launch_bounds__(256) void argmax_kernel(const float* P, int* idx, float* val, int N, int nc) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* p = P + (size_t)n * nc;
  int mi = -1;
  float mv = p[0];
  for (int i = 0; i < nc; i++) {
    const float x = p[i];
    if (x < mv) continue;
    mi = i;
    mv = x;
  }
  idx[n] = mi;
  val[n] = mv;
}

// trivial_decode (ctc.cc:159-190): one wave per line; 64 frames are loaded at once and replayed
// through the run-length state machine with wave-uniform lane reads.
__global__ __launch_bounds__(64) void decode_kernel(const int* idx, const float* val, const int* line_off,
                                                     int* out_cls, int* out_loc, int* out_cnt) {
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  const int off = line_off[b], T = line_off[b + 1] - off;
  int n = 0;
  float mv = 0.0f;
  int mc = -1, mt = -1;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    const int my_i = t < T ? idx[off + t] : 0;
    const float my_v = t < T ? val[off + t] : 0.0f;
    const int cnt = (T - t0) < 64 ? (T - t0) : 64;
    for (int k = 0; k < 64; k++) {
      const int index = wave_shfl_i(my_i, k);
      const float v = wave_shfl(my_v, k);
      if (k < cnt) {
        if (index == 0) {
          if (mc != -1 && mc != 0) {
            if (lane == 0) { out_cls[off + n] = mc; out_loc[off + n] = mt; }
            n++;
          }
          mv = 0.0f; mc = -1; mt = -1;
        } else if (v > mv) {
          mv = v; mc = index; mt = t0 + k;
        }
      }
    }
  }
  if (lane == 0) out_cnt[b] = n;
}

}  // namespace clstm
