// ctc.h -- clstm's CTC alignment (ctc.cc:24-134) and trivial_decode (ctc.cc:159-190) on device.
//
// One workgroup per text line.  clstm's variant is NOT the textbook CTC: transitions are
// stay / advance-by-one only, every state and frame carries a skip = -5 boundary term,
// log_add cuts off at |x-y| > 10 (tensor.h:86-89), posteriors are normalised per STATE over
// time and then per frame.  All of that is replicated here:
//   A. lmatch[t][s] = log(out_t[class_s]), out_t = max(1e-5, p_t) / sum      (ctc.cc:66-77)
//   B. wave 0 runs the forward recursion, wave 1 the same recursion on the (t,s)-reversed
//      lattice (= forwardbackward, ctc.cc:42-55).  Lane l owns R = ceil(S/64) consecutive
//      states; the j-1 neighbour of its first state comes from lane l-1 (wave shift), so the
//      label axis is a wave scan and only t is serial.
//   C. epath = limexp(both - max)                                            (ctc.cc:82)
//   D. per-state normalisation over t, floor 1e-9, double accumulator        (ctc.cc:83-88)
//   E. aligned[t][c] = sum_s epath[t][s] [class_s == c]; per-frame normalise (ctc.cc:91-109)
//      and the fused delta  d = aligned - p                                  (clstmhl.h:211-212)
// Targets are given as one class per state (the Classes overload, ctc.cc:136-146; mktargets'
// blank-interleaved list for OCR lines, ctc.cc:148-157).
// Numerical note: step E accumulates in float in state order (the reference uses a double
// accumulator narrowed to Float) -- a <=1e-7 relative difference, inside the 1e-4 parity bar.
#pragma once
#include "devintrin.h"

namespace clstm {

constexpr int CTC_RMAX = 8;      // up to 512 states per line
constexpr int CTC_THREADS = 256;

struct CtcArgs {
  const float* P;        // [N][nc] softmax outputs
  float* Dz;             // [N][nc] out: aligned - P
  float* aligned;        // [N][nc] out (optional, may be null): alignment posteriors
  const int* line_off;   // [bs+1]
  const int* states;     // packed state classes
  const int* state_off;  // [bs+1]
  float* lat;            // lattice workspace: per line 3*T*S floats at lat_off[b]
  const long long* lat_off;
  int nc;
};

DEVFN float ctc_log_add(float x, float y) {  // tensor.h:86-89
  if (fabsf(x - y) > 10.0f) return fmaxf(x, y);
  return logf(expf(x - y) + 1.0f) + y;
}
DEVFN float ctc_limexp(float x) {  // tensor.h:78-82
  if (x < -30.0f) return (float)exp(-30.0);
  if (x > 30.0f) return (float)exp(30.0);
  return expf(x);
}

__global__ __launch_bounds__(CTC_THREADS) void ctc_align_kernel(CtcArgs a) {
  __shared__ float red[CTC_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int nc = a.nc;
  const int off = a.line_off[b], T = a.line_off[b + 1] - off;
  const int soff = a.state_off[b], S = a.state_off[b + 1] - soff;
  if (T <= 0 || S <= 0) return;
  const float* P = a.P + (size_t)off * nc;
  float* Dz = a.Dz + (size_t)off * nc;
  const int* st = a.states + soff;
  float* lm = a.lat + a.lat_off[b];
  float* al = lm + (size_t)T * S;
  float* be = al + (size_t)T * S;

  // ---- A: match scores -------------------------------------------------------------
  for (int t = tid; t < T; t += CTC_THREADS) {
    const float* p = P + (size_t)t * nc;
    float asum = 0.0f;
    for (int c = 0; c < nc; c++) asum += fmaxf(1e-5f, p[c]);
    for (int s = 0; s < S; s++) {
      const float o = fmaxf(1e-5f, p[st[s]]) / asum;
      lm[(size_t)t * S + s] = (float)log((double)o);
    }
  }
  __syncthreads();

  // ---- B: forward (wave 0) and reversed-lattice forward (wave 1) -------------------------
  if (wave < 2) {
    const bool rev = wave == 1;
    const int R = (S + 63) / 64;
    float v[CTC_RMAX], lmv[CTC_RMAX];
#pragma unroll
    for (int r = 0; r < CTC_RMAX; r++) {
      const int j = lane * R + r;
      v[r] = (float)(-5.0 * j);
      lmv[r] = 0.0f;
      if (r < R && j < S) lmv[r] = rev ? lm[(size_t)(T - 1) * S + (S - 1 - j)] : lm[j];
    }
    float* out = rev ? be : al;
    for (int i = 0; i < T; i++) {
      float lmn[CTC_RMAX];
#pragma unroll
      for (int r = 0; r < CTC_RMAX; r++) {  // prefetch next lattice row
        const int j = lane * R + r;
        lmn[r] = 0.0f;
        if (r < R && j < S && i + 1 < T)
          lmn[r] = rev ? lm[(size_t)(T - 2 - i) * S + (S - 1 - j)] : lm[(size_t)(i + 1) * S + j];
      }
      float last = v[0];
#pragma unroll
      for (int r = 1; r < CTC_RMAX; r++)
        if (r == R - 1) last = v[r];
      const float from_prev_lane = wave_shfl_up1(last);
#pragma unroll
      for (int r = CTC_RMAX - 1; r >= 0; r--) {
        if (r < R) {
          const int j = lane * R + r;
          float w = (r == 0) ? from_prev_lane : v[r - 1];
          if (j == 0) w = (float)(-5.0 * i);
          const float same = v[r] + lmv[r];
          const float next = w + lmv[r];
          v[r] = ctc_log_add(same, next);
          if (j < S) {
            if (rev) out[(size_t)(T - 1 - i) * S + (S - 1 - j)] = v[r];
            else out[(size_t)i * S + j] = v[r];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < CTC_RMAX; r++) lmv[r] = lmn[r];
    }
  }
  __syncthreads();

  // ---- C: epath = limexp(both - amax2(both)) ---------------------------------------------
  const int TS = T * S;
  float mx = -3.0e38f;
  for (int i = tid; i < TS; i += CTC_THREADS) mx = fmaxf(mx, al[i] + be[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int i = 1; i < CTC_THREADS / 64; i++) mx = fmaxf(mx, red[i]);
  for (int i = tid; i < TS; i += CTC_THREADS) al[i] = ctc_limexp((al[i] + be[i]) - mx);
  __syncthreads();

  // ---- D: normalise every state column over time -----------------------------------------
  for (int s = tid; s < S; s += CTC_THREADS) {
    double total = 0.0;
    for (int t = 0; t < T; t++) total += al[(size_t)t * S + s];
    total = fmax(1e-9, total);
    for (int t = 0; t < T; t++) al[(size_t)t * S + s] = (float)((double)al[(size_t)t * S + s] / total);
  }
  __syncthreads();

  // ---- E: project states onto classes, normalise per frame, emit deltas --------------------
  for (int t = tid; t < T; t += CTC_THREADS) {
    float* row = Dz + (size_t)t * nc;
    for (int c = 0; c < nc; c++) row[c] = 0.0f;
    for (int s = 0; s < S; s++) row[st[s]] += al[(size_t)t * S + s];
    double total = 0.0;
    for (int c = 0; c < nc; c++) total += row[c];
    total = fmax(total, 1e-9);
    const float* p = P + (size_t)t * nc;
    float* arow = a.aligned ? a.aligned + ((size_t)off + t) * nc : nullptr;
    for (int c = 0; c < nc; c++) {
      const float av = (float)((double)row[c] / total);
      if (arow) arow[c] = av;
      row[c] = av - p[c];
    }
  }
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
