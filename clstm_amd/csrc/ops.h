// ops.h -- element-wise / small kernels: the 1:1 per-op side of the C ABI
// (clstm_compute.h:72-103 operators on the BiLSTM+CTC path), softmax normalisation,
// the fused clip+SGD update and the weight re-packing for the sequence kernels.
#pragma once
#include "devintrin.h"

namespace clstm {

enum { LIN = 0, SIG = 1, TANH = 2, RELU = 3, LOGMAG = 4 };  // clstm_compute.h:10-14

DEVFN float nonlin_fwd(float x, int nl) {  // clstm_compute.cc:113-129
  switch (nl) {
    case SIG: return sigmoid_dev(x);
    case TANH: return tanh_dev(x);
    case RELU: return x > 0.0f ? x : 0.0f;
    case LOGMAG: return logf(fabsf(x) + 1.0f) * ((x < 0.0f ? 1.0f : 0.0f) * -2.0f + 1.0f);
    default: return x;
  }
}
DEVFN float nonlin_deriv(float y, int nl) {  // f'(x) expressed through y=f(x), :152-167
  switch (nl) {
    case SIG: return y * (-y + 1.0f);
    case TANH: return -y * y + 1.0f;
    case RELU: return y > 0.0f ? 1.0f : 0.0f;
    case LOGMAG: return expf(-fabsf(y));
    default: return 1.0f;
  }
}

#define CLSTM_GRID_STRIDE(i, n) \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); i += (size_t)gridDim.x * blockDim.x)

__global__ void k_forward_nonlin0(float* y, size_t len, int nl) {
  CLSTM_GRID_STRIDE(i, len) y[i] = nonlin_fwd(y[i], nl);
}
__global__ void k_backward_nonlin0(const float* yv, float* yd, size_t len, int nl) {
  CLSTM_GRID_STRIDE(i, len) yd[i] = nonlin_deriv(yv[i], nl) * yd[i];
}
__global__ void k_forward_nonlin(float* y, const float* x, size_t len, int nl) {
  CLSTM_GRID_STRIDE(i, len) y[i] = nonlin_fwd(x[i], nl);
}
__global__ void k_backward_nonlin(const float* yv, const float* yd, float* xd, size_t len, int nl) {
  CLSTM_GRID_STRIDE(i, len) xd[i] += nonlin_deriv(yv[i], nl) * yd[i];
}
// y(i,b) = sum_k W(i,1+k) x(k,b) + W(i,0); optional nonlinearity (full1) -- per-op path only
__global__ void k_forward_lin1(float* y, const float* W, const float* x, int n, int m, int bs, int nl) {
  CLSTM_GRID_STRIDE(e, (size_t)n * bs) {
    const int i = e % n, b = e / n;
    const int nx = m - 1;
    float acc = 0.0f;
    for (int k = 0; k < nx; k++) acc += W[i + (size_t)n * (1 + k)] * x[k + (size_t)nx * b];
    acc += W[i];
    y[e] = nl >= 0 ? nonlin_fwd(acc, nl) : acc;
  }
}
// x.d(k,b) (+)= sum_i W(i,1+k) y.d(i,b)
__global__ void k_backward_lin1_dx(const float* yd, const float* W, float* xd, int n, int m, int bs, int assign) {
  const int nx = m - 1;
  CLSTM_GRID_STRIDE(e, (size_t)nx * bs) {
    const int k = e % nx, b = e / nx;
    float acc = 0.0f;
    for (int i = 0; i < n; i++) acc += W[i + (size_t)n * (1 + k)] * yd[i + (size_t)n * b];
    xd[e] = assign ? acc : xd[e] + acc;
  }
}
// W.d(i,j) += sum_b y.d(i,b) * [1 ; x](j,b)
__global__ void k_backward_lin1_dw(const float* yd, float* Wd, const float* x, int n, int m, int bs) {
  const int nx = m - 1;
  CLSTM_GRID_STRIDE(e, (size_t)n * m) {
    const int i = e % n, j = e / n;
    float acc = 0.0f;
    for (int b = 0; b < bs; b++) acc += yd[i + (size_t)n * b] * (j == 0 ? 1.0f : x[(j - 1) + (size_t)nx * b]);
    Wd[e] += acc;
  }
}
// z = limexp(z) / colsum  (no max-subtraction, clstm_compute.cc:333-339). One wave per column.
DEVFN float limexp_dev(float x) {  // tensor.h:78-82
  if (x < -30.0f) return (float)exp(-30.0);
  if (x > 30.0f) return (float)exp(30.0);
  return expf(x);
}
__global__ __launch_bounds__(256) void k_softmax_norm(float* z, int n, size_t cols, int* nanflag = nullptr, int step_no = 0) {
  const int lane = threadIdx.x & 63;
  const size_t col = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const bool ok = col < cols;   // keep whole waves alive for the wave reduction
  float* p = z + (ok ? col : 0) * n;
  float s = 0.0f;
  bool nonfinite = false;   // limexp's clamp would swallow a NaN logit: looked at before it (k_update)
  for (int i = lane; i < n; i += 64) {
    const float x = p[i];
    nonfinite |= ok && !f32_finite(x);
    const float e = limexp_dev(x);
    if (ok) p[i] = e;
    s += e;
  }
  if (nanflag && nonfinite) raise_nonfinite(nanflag, step_no);
  s = wave_sum(s);
  if (ok)
    for (int i = lane; i < n; i += 64) p[i] = p[i] / s;
}
__global__ void k_stack(float* z, const float* x, const float* y, int nx, int ny, int bs) {
  CLSTM_GRID_STRIDE(e, (size_t)(nx + ny) * bs) {
    const int i = e % (nx + ny), b = e / (nx + ny);
    z[e] = i < nx ? x[i + (size_t)nx * b] : (y ? y[(i - nx) + (size_t)ny * b] : 0.0f);
  }
}
__global__ void k_unstack_add(const float* z, float* x, float* y, int nx, int ny, int bs) {
  CLSTM_GRID_STRIDE(e, (size_t)(nx + ny) * bs) {
    const int i = e % (nx + ny), b = e / (nx + ny);
    if (i < nx) x[i + (size_t)nx * b] += z[e];
    else if (y) y[(i - nx) + (size_t)ny * b] += z[e];
  }
}
// Sequence blocks: dims (rows, bs, 2, N) -- batches.h:79-86
__global__ void k_forward_reverse(float* y, const float* x, size_t step, int N) {
  CLSTM_GRID_STRIDE(e, step * N) {
    const size_t t = e / step, r = e % step;
    y[(size_t)(N - 1 - t) * step + r] = x[e];  // step = rows*bs*2: copies v and d planes
  }
}
__global__ void k_backward_reverse(const float* y, float* x, size_t plane, int N) {
  CLSTM_GRID_STRIDE(e, plane * N) {
    const size_t t = e / plane, r = e % plane;
    x[(size_t)(N - 1 - t) * 2 * plane + plane + r] += y[t * 2 * plane + plane + r];
  }
}
// forward_btswitch / backward_btswitch (clstm_compute.cc:425-447): swap the batch and time axes of a Sequence block
// (rows, bs, 2, N) -> (rows, N, 2, bs); forward moves the value plane, backward accumulates the derivative plane
__global__ void k_forward_btswitch(float* y, const float* x, int rows, int bs, int N) {
  CLSTM_GRID_STRIDE(e, (size_t)rows * bs * N) {
    const int i = e % rows, b = (e / rows) % bs, t = e / ((size_t)rows * bs);
    y[i + (size_t)rows * (t + (size_t)N * (0 + 2 * b))] = x[i + (size_t)rows * (b + (size_t)bs * (0 + 2 * t))];
  }
}
__global__ void k_backward_btswitch(const float* y, float* x, int rows, int bs, int N) {
  CLSTM_GRID_STRIDE(e, (size_t)rows * bs * N) {
    const int i = e % rows, b = (e / rows) % bs, t = e / ((size_t)rows * bs);
    x[i + (size_t)rows * (b + (size_t)bs * (1 + 2 * t))] += y[i + (size_t)rows * (t + (size_t)N * (1 + 2 * b))];
  }
}
// forward_batchstack / backward_batchstack (clstm_compute.cc:451-500): y(d*(pre+k) + f, b, ., t) = x(f, b + k, ., t) for
// k = -pre..post where b + k is a valid batch column, zero elsewhere; forward clears BOTH planes of y and fills the
// value plane, backward accumulates the derivative plane back into x
__global__ void k_forward_batchstack(float* y, const float* x, int d, int bs, int N, int pre, int post) {
  const int copies = pre + post + 1;
  CLSTM_GRID_STRIDE(e, (size_t)copies * d * bs * 2 * N) {
    const int r = e % (copies * d);
    const size_t q = e / (copies * d);
    const int b = q % bs, p = (q / bs) % 2, t = q / ((size_t)bs * 2);
    const int k = r / d - pre, f = r % d, sb = b + k;
    y[e] = (p == 0 && sb >= 0 && sb < bs) ? x[f + (size_t)d * (sb + (size_t)bs * (0 + 2 * t))] : 0.0f;
  }
}
__global__ void k_backward_batchstack(const float* y, float* x, int d, int bs, int N, int pre, int post) {
  const int copies = pre + post + 1;
  CLSTM_GRID_STRIDE(e, (size_t)d * bs * N) {   // one thread per x element: no atomics, k summed in the reference's order
    const int f = e % d, sb = (e / d) % bs, t = e / ((size_t)d * bs);
    float acc = x[f + (size_t)d * (sb + (size_t)bs * (1 + 2 * t))];
    for (int k = -pre; k <= post; k++) {
      const int b = sb - k;
      if (b >= 0 && b < bs) acc += y[(size_t)d * (pre + k) + f + (size_t)copies * d * (b + (size_t)bs * (1 + 2 * t))];
    }
    x[f + (size_t)d * (sb + (size_t)bs * (1 + 2 * t))] = acc;
  }
}
__global__ void k_forward_statemem(float* st, const float* ci, const float* gi, const float* last,
                                   const float* gf, size_t len) {
  CLSTM_GRID_STRIDE(i, len) {
    float c = ci[i] * gi[i];
    if (last) c += gf[i] * last[i];
    st[i] = c;
  }
}
__global__ void k_backward_statemem(const float* sd, const float* ci, float* cid, const float* gi, float* gid,
                                    const float* last, float* lastd, const float* gf, float* gfd, size_t len) {
  CLSTM_GRID_STRIDE(i, len) {
    const float d = sd[i];
    if (last) { lastd[i] += d * gf[i]; gfd[i] += d * last[i]; }
    gid[i] += d * ci[i];
    cid[i] += d * gi[i];
  }
}
__global__ void k_forward_nonlingate(float* out, const float* st, const float* go, size_t len, int nl) {
  CLSTM_GRID_STRIDE(i, len) out[i] = nonlin_fwd(st[i], nl) * go[i];
}
__global__ void k_backward_nonlingate(const float* outd, const float* st, float* std_, const float* go,
                                      float* god, size_t len, int nl) {
  CLSTM_GRID_STRIDE(i, len) {
    const float t = nonlin_fwd(st[i], nl);
    god[i] += t * outd[i];
    std_[i] += nonlin_deriv(t, nl) * (go[i] * outd[i]);
  }
}
__global__ void k_clip(float* d, size_t len, float clip) {
  CLSTM_GRID_STRIDE(i, len) d[i] = fmaxf(-clip, fminf(clip, d[i]));
}
__global__ void k_sgd(float* v, float* d, size_t len, float lr, float mom) {
  CLSTM_GRID_STRIDE(i, len) {
    const float di = d[i];
    v[i] += di * lr;
    d[i] = di * mom;
  }
}
// fused: d += g ; clip ; v += lr*d ; d *= mom    (clstm.cc:201-217 on the flat buffers).
// `err` (may be null): the device error words of the library -- [0] outcome of the persistent recurrences of earlier
// launches (workgroups misplaced / a group barrier timed out), [1] weight-gradient items that gave up waiting.  The
// host learns about those asynchronously (it keeps enqueueing ahead of the GPU); the UPDATE must not: a gradient
// computed from unwritten activations is not applied, parameters and momentum stay as they were.
// `step_word` (may be null; pinned host memory): set to step_id -- tells a host that feeds frames from its own memory
// that every kernel of this step, the input ingest first of all, is behind it (clstm_net_train_step_h).
// `nanflag` (may be null: CLSTM_NANCHECK=0): device error word [3].  The reference aborts on a NaN in any backward step
// (clstm.cc:630-649, nine assert scans per time step).  Here (a) the forward pass looks at every softmax logit BEFORE limexp's
// clamp (which would turn a NaN into exp(-30)): a NaN / Inf anywhere in the recurrent state reaches the logits of its frame,
// so a diverged forward pass raises the word before the backward pass starts and the WHOLE update is skipped; (b) every
// gradient entry passes through exactly one thread of the slab reduction / of this kernel anyway: a non-finite entry born
// in the backward pass is never applied (entry by entry when the update is fused into the reduction).  The word takes the
// number of the training step, every later update is skipped (dev_err_set) and the host reports it at its next
// synchronisation point.
__global__ void k_update(float* v, float* d, const float* g, size_t len, float lr, float mom, float clip, const int* err,
                         int* step_word, int step_id, int* nanflag, int step_no) {
  if (step_word && blockIdx.x == 0 && threadIdx.x == 0) store_i32_wt(step_word, step_id);
  if (dev_err_set(err)) return;
  CLSTM_GRID_STRIDE(i, len) {
    const float gi = g[i];
    if (nanflag && !f32_finite(gi)) { raise_nonfinite(nanflag, step_no); continue; }   // (after an all-reduce every rank sees the same g: same decisions)
    float di = d[i] + gi;
    if (clip < 1e6f) di = fmaxf(-clip, fminf(clip, di));
    v[i] += di * lr;
    d[i] = di * mom;
  }
}

// ---- one-shot peer-read all-reduce fused into the update (clstm_hip.hip: PeerExchange) --------------------
// Every rank's fresh minibatch gradient lies in an exchange buffer of its own that the other ranks have mapped (HIP IPC;
// over xGMI between GPUs).  k_peer_barrier: one workgroup; lane r stores this step's sequence number into rank r's flag
// array (my slot of it) and lane r waits until rank r's number has arrived in mine -- after that every rank's buffer of
// this step is complete (the stores of the producing kernels were released at their kernel boundaries) and nobody still
// reads the buffers of two steps ago (the slot this step's reductions wrote).  k_peer_allreduce_update: thread i sums
// element i of all ranks' buffers IN RANK ORDER (every rank forms the same sum: replicas stay bit-identical, like a
// deterministic all-reduce), leaves it in g and applies k_update's arithmetic -- the 0.54 MB ncclAllReduce (tens of us of
// exposed latency between the last reduction and the update) and the separate update launch become one barrier of one
// workgroup plus one pass.  System-scope (sc0 sc1) loads on the reader side: a rank's L2 must not serve the same slot's
// lines of two steps ago.
constexpr int PEER_MAX_RANKS = 16;
struct PeerArgs {
  const float* x[PEER_MAX_RANKS];   // the ranks' exchange buffers (this step's slot), own rank included
  int* f[PEER_MAX_RANKS];           // the ranks' flag arrays (this step's slot): f[r][q] = last sequence number rank q announced to r
  int nranks, rank;
};
// `timeout_ticks`: wall_clock() ticks (100 MHz) a lane waits for its peer before it gives up.  The HOSTS of all ranks have
// announced this sequence number to each other before any of them enqueues this kernel (clstm_hip.hip: Comm::announce), so what
// is waited for here is the peers' queued device work, never a rank whose host is busy elsewhere (rank 0's test / save phases
// in clstmocrtrain ngpu=N): the bound is a hang detector (default two minutes), not a scheduling assumption.  A time-out bumps
// the error word the caller passes -- device error word [6] in training steps: its own message, updates skipped, nothing else
// in the library changes mode.
__global__ void k_peer_barrier(PeerArgs p, int seq, int* err, long long timeout_ticks) {
  const int r = threadIdx.x;
  if (blockIdx.x != 0 || r >= p.nranks) return;
  store_i32_wt(p.f[r] + p.rank, seq);
  const long long t0 = wall_clock();
  int spins = 0;
  while (load_i32_wt(p.f[p.rank] + r) != seq) {
    poll_pause();
    if ((++spins & 1023) == 0 && wall_clock() - t0 > timeout_ticks) { atomic_add_i32(err, 1); break; }   // never hang the device: the update is skipped (dev_err_set)
  }
}
// set-up probe of the mapped buffers (clstm_hip.hip: Comm::peer_ready).  k_peer_fill: a rank writes a pattern that depends on
// (rank, element, round) into the WHOLE slot it owns, with the plain stores the gradient reductions use.  k_peer_probe, behind
// one k_peer_barrier: every element of every rank's slot must show that rank's pattern of THIS round through this rank's
// mapping of it.  The host runs four rounds -- both slots, each twice with different patterns -- so a mapping that opens but
// serves another buffer, drops part of a slot, or hands back the previous round's lines (a cache between the two devices that
// the system-scope loads do not bypass) fails here, at set-up, and every rank falls back to RCCL.
DEVFN float peer_pattern(int rank, size_t i, int round) { return (float)(((unsigned)i * 31u + (unsigned)rank * 4099u + (unsigned)round * 977u) & 0xFFFFFu); }
__global__ void k_peer_fill(float* slot, size_t len, int rank, int round) {
  CLSTM_GRID_STRIDE(i, len) slot[i] = peer_pattern(rank, i, round);
}
__global__ void k_peer_probe(PeerArgs p, size_t len, int round, int* err) {
  int bad = 0;
  for (int r = 0; r < p.nranks; r++) {
    const BufF32 b = make_buf(p.x[r], len * 4);
    CLSTM_GRID_STRIDE(i, len)
      if (buf_load_wt(b, (unsigned)(i * 4)) != peer_pattern(r, i, round)) bad = 1;
  }
  if (bad) atomic_add_i32(err, 1);
}
__global__ void k_peer_allreduce_update(PeerArgs p, float* v, float* d, float* g, size_t len, float lr, float mom, float clip, const int* err,
                                        int* step_word, int step_id, int* nanflag, int step_no) {
  if (step_word && blockIdx.x == 0 && threadIdx.x == 0) store_i32_wt(step_word, step_id);
  const bool apply = v != nullptr && !dev_err_set(err);   // (v null: a plain in-place all-reduce into g)
  BufF32 xb[PEER_MAX_RANKS];
  for (int r = 0; r < p.nranks; r++) xb[r] = make_buf(p.x[r], len * 4);
  const size_t n4 = (len + 3) / 4;
  CLSTM_GRID_STRIDE(q, n4) {
    f32x4 acc = buf_load4_wt(xb[0], (unsigned)(q * 16));          // (past the end: zeros, the descriptor ends at len)
    for (int r = 1; r < p.nranks; r++) {
      const f32x4 t = buf_load4_wt(xb[r], (unsigned)(q * 16));
      acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const size_t i = q * 4 + e;
      if (i >= len) break;
      const float gi = acc[e];
      g[i] = gi;
      if (!apply) continue;
      if (nanflag && !f32_finite(gi)) { raise_nonfinite(nanflag, step_no); continue; }
      float di = d[i] + gi;
      if (clip < 1e6f) di = fmaxf(-clip, fminf(clip, di));
      v[i] += di * lr;
      d[i] = di * mom;
    }
  }
}

// ---- replica consistency check (every rank applies the identical update, so the replicas must stay bit-identical; the
// reference re-synchronises instead, distribute_weights / average_weights, clstm.cc:718-729, 746-760) -------------------------
// Every `check_every` training steps each rank folds its parameter buffer into two 32-bit integer sums (the bit patterns, and
// the bit patterns rotated by the element index: integer adds commute, so the launch geometry does not matter), splits them
// into four 16-bit pieces -- exact as floats, and exact when summed over up to 256 ranks -- and the pieces travel through the
// SAME all-reduce the gradients use.  k_replica_verify then holds sum == nranks * own on every rank: any rank whose pieces differ
// from the mean sees it, raises device error word [7] with the step number, and no update is applied from then on; the host
// reports "replicas diverged" at its next synchronisation point.
__global__ void k_param_checksum(const float* v, size_t len, unsigned* acc2) {
  unsigned a = 0, b = 0;
  CLSTM_GRID_STRIDE(i, len) {
    const unsigned x = __builtin_bit_cast(unsigned, v[i]);
    const unsigned r = (unsigned)i & 31u;
    a += x;
    b += (x << r) | (r ? x >> (32u - r) : 0u);
  }
  atomic_add_i32(reinterpret_cast<int*>(acc2), (int)a);       // (wrapping adds: the same sums for any grid)
  atomic_add_i32(reinterpret_cast<int*>(acc2) + 1, (int)b);
}
// chk[0..3] = this rank's pieces (kept), chk[4..7] = the copy the all-reduce sums in place; acc2 is returned to zero
__global__ void k_checksum_pieces(unsigned* acc2, float* chk) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const unsigned a = acc2[0], b = acc2[1];
  const float p[4] = {(float)(a & 0xFFFFu), (float)(a >> 16), (float)(b & 0xFFFFu), (float)(b >> 16)};
  for (int i = 0; i < 4; i++) { chk[i] = p[i]; chk[4 + i] = p[i]; }
  acc2[0] = 0; acc2[1] = 0;
}
__global__ void k_replica_verify(const float* chk, int nranks, int* errword, int step_no) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  bool same = true;
  for (int i = 0; i < 4; i++) same = same && chk[4 + i] == chk[i] * (float)nranks;
  if (!same && *errword == 0) *errword = step_no > 0 ? step_no : 1;
}

// ---- weight packing for the sequence kernels ---------------------------------------------------
struct PackDesc {
  long long p_off[2][4];  // flat offsets of (dir, slot) blocks; slot 0 gi,1 gf,2 go,3 ci
  int ni, no, ndir, nk4, nthreads;
  int ku;   // k values a forward lane owns = cells per quarter (<= 4*nk4)
};
// One launch repacks a layer's parameters after every update:
//   Wt[k][m] (k < ni, m = dir*4no + 4*cell + slot) and bias[m] for the hoisted input GEMM,
//   Rf[dir][(s*KQP + kk)][tid] = R_{s^q}[cell][q*KU + kk]        forward recurrence registers,
//   Rb[dir][(i*SLP + pp)][tid] = R_g[j][4*kg + (i^Q)], (g,j) = pair js*SL+pp   backward recurrence registers.
// Every packed element is a COPY of one parameter (or a zero pad): `*_src` gives its flat index in v (-1: zero).  The maps
// depend only on the layer's geometry, so the per-step repack of a training step reads them from a table built once
// (k_pack_index; the integer divisions below then leave the step) -- k_pack_layer evaluates them in place.
DEVFN long long pack_wx_src(size_t e, const PackDesc& p) {
  const int M = p.ndir * 4 * p.no;
  const int j = e / M, m = e % M;
  const int dir = m / (4 * p.no), c = (m % (4 * p.no)) >> 2, s = m & 3;
  return p.p_off[dir][s] + c + (long long)p.no * j;
}
DEVFN void pack_wx_store(size_t e, float x, float* Wt, float* bias, const PackDesc& p) {   // row 0 = bias, rows 1.. = Wt
  const size_t M = (size_t)p.ndir * 4 * p.no;
  if (e < M) bias[e] = x; else Wt[e - M] = x;
}
DEVFN void pack_wx(size_t e, const float* v, float* Wt, float* bias, const PackDesc& p) { pack_wx_store(e, v[pack_wx_src(e, p)], Wt, bias, p); }
DEVFN long long pack_rf_src(size_t e, const PackDesc& p) {
  const int KQP = 4 * p.nk4;
  const size_t per_dir = (size_t)4 * KQP * p.nthreads;
  const int dir = e / per_dir;
  const size_t r = e % per_dir;
  const int tid = r % p.nthreads, gk = r / p.nthreads;
  const int g = gk / KQP, kk = gk % KQP;
  const int lane = tid & 63, wave = tid >> 6;
  const int cell = wave * 16 + (lane >> 2), q = lane & 3;
  const int k = stag_on(p.nk4) ? stag_fwd_k(q, kk, p.ku) : q * p.ku + kk;   // (staggered recurrence: [group A cells | group B cells])
  // register slot g of lane q holds gate g^q: the quad reduce-scatter then needs no selects (lstm_seq.h)
  if (cell < p.no && kk < p.ku && k < p.no) return p.p_off[dir][g ^ q] + cell + (long long)p.no * (1 + p.ni + k);
  return -1;
}
DEVFN void pack_rf(size_t e, const float* v, float* Rf, const PackDesc& p) { const long long s = pack_rf_src(e, p); Rf[e] = s >= 0 ? v[s] : 0.0f; }
DEVFN long long pack_rb_src(size_t e, const PackDesc& p) {
  const int SLP = 4 * p.nk4;
  const int SL = (4 * p.no + 15) / 16;
  const size_t per_dir = (size_t)4 * SLP * p.nthreads;
  const int dir = e / per_dir;
  const size_t r = e % per_dir;
  const int tid = r % p.nthreads, ip = r / p.nthreads;
  const int i = ip / SLP, pp = ip % SLP;
  const int lane = tid & 63, wave = tid >> 6;
  const int js = lane & 15;
  const int kcell = 4 * (wave * 4 + (lane >> 4)) + (i ^ (js >> 2));   // slot i of quad Q holds cell i^Q
  const int pr = js * SL + pp;
  if (pp < SL && pr < 4 * p.no && kcell < p.no) {
    const int g = pr / p.no, j = pr % p.no;
    return p.p_off[dir][g] + j + (long long)p.no * (1 + p.ni + kcell);
  }
  return -1;
}
DEVFN void pack_rb(size_t e, const float* v, float* Rb, const PackDesc& p) { const long long s = pack_rb_src(e, p); Rb[e] = s >= 0 ? v[s] : 0.0f; }
// k-contiguous packs for the wave-sized items of the fused forward launch (lstm_fwd_fused.h), zero padded:
//   Wk[dir][m = 4*cell + slot][k]   = W_slot[cell][k] (input columns only), njp*16 rows of kp floats per direction
//   W1k[c][k]                       = W1[c][1 + k], 96 rows of kps floats (softmax layer, sm_off = its flat offset)
struct PackFused {
  float* Wk; int kp, njp;
  float* W1k; int kps, nc, sm_k; long long sm_off;
};
DEVFN size_t pack_fused_count(const PackFused& f, const PackDesc& p) {
  return f.Wk ? (size_t)p.ndir * f.njp * 16 * f.kp + (size_t)96 * f.kps : 0;
}
DEVFN long long pack_fused_src(size_t e, const PackFused& f, const PackDesc& p) {
  const size_t nwk = (size_t)p.ndir * f.njp * 16 * f.kp;
  if (e < nwk) {
    const int k = e % f.kp;
    const size_t r = e / f.kp;
    const int m = r % (f.njp * 16), dir = r / (f.njp * 16);
    const int cell = m >> 2, slot = m & 3;
    return (cell < p.no && k < p.ni) ? p.p_off[dir][slot] + cell + (long long)p.no * (1 + k) : -1;
  }
  const size_t e2 = e - nwk;
  const int k = e2 % f.kps, c = e2 / f.kps;
  return (c < f.nc && k < f.sm_k) ? f.sm_off + c + (long long)f.nc * (1 + k) : -1;
}
DEVFN void pack_fused_store(size_t e, float x, const PackFused& f, const PackDesc& p) {
  const size_t nwk = (size_t)p.ndir * f.njp * 16 * f.kp;
  if (e < nwk) f.Wk[e] = x; else f.W1k[e - nwk] = x;
}
DEVFN void pack_fused(size_t e, const float* v, const PackFused& f, const PackDesc& p) {
  const long long s = pack_fused_src(e, f, p);
  pack_fused_store(e, s >= 0 ? v[s] : 0.0f, f, p);
}
// tab[e] = source index of packed element e of the combined range [W_x rows | Rf | Rb | fused-forward packs] (-1: zero pad)
__global__ void k_pack_index(int* tab, PackDesc p, PackFused pf) {
  const size_t nwx = (size_t)(1 + p.ni) * p.ndir * 4 * p.no;
  const size_t nr = (size_t)p.ndir * 4 * 4 * p.nk4 * p.nthreads;
  const size_t nf = pack_fused_count(pf, p);
  CLSTM_GRID_STRIDE(e, nwx + 2 * nr + nf) {
    long long s;
    if (e < nwx) s = pack_wx_src(e, p);
    else if (e < nwx + nr) s = pack_rf_src(e - nwx, p);
    else if (e < nwx + 2 * nr) s = pack_rb_src(e - nwx - nr, p);
    else s = pack_fused_src(e - nwx - 2 * nr, pf, p);
    tab[e] = (int)s;
  }
}
__global__ void k_pack_layer(const float* v, float* Wt, float* bias, float* Rf, float* Rb, PackDesc p, PackFused pf) {
  const size_t nwx = (size_t)(1 + p.ni) * p.ndir * 4 * p.no;
  const size_t nr = (size_t)p.ndir * 4 * 4 * p.nk4 * p.nthreads;
  const size_t nf = pack_fused_count(pf, p);
  CLSTM_GRID_STRIDE(e, nwx + 2 * nr + nf) {
    if (e < nwx) pack_wx(e, v, Wt, bias, p);
    else if (e < nwx + nr) pack_rf(e - nwx, v, Rf, p);
    else if (e < nwx + 2 * nr) pack_rb(e - nwx - nr, v, Rb, p);
    else pack_fused(e - nwx - 2 * nr, v, pf, p);
  }
}
// k-contiguous, zero-padded recurrent weights of the lock-step recurrence (lstm_wide.h):
//   Rwf[dir][cg][j = cl*4+g][k] = R_g[4cg+cl][k]          (kpf floats per row)
//   Rwb[dir][ct][c][kk = 4j+g]  = R_g[j][16ct+c]          (kpb floats per row)
__global__ void k_pack_wide(const float* v, float* Rwf, float* Rwb, PackDesc p, int kpf, int kpb) {
  const int ncg = (p.no + 3) / 4, nct = (p.no + 15) / 16;
  const size_t nf = (size_t)p.ndir * ncg * 16 * kpf, nb = (size_t)p.ndir * nct * 16 * kpb;
  CLSTM_GRID_STRIDE(e, nf + nb) {
    if (e < nf) {
      const int k = e % kpf;
      const size_t r = e / kpf;
      const int j = r % 16, cg = (r / 16) % ncg, dir = r / ((size_t)16 * ncg);
      const int cell = cg * 4 + (j >> 2), g = j & 3;
      Rwf[e] = (cell < p.no && k < p.no) ? v[p.p_off[dir][g] + cell + (size_t)p.no * (1 + p.ni + k)] : 0.0f;
    } else {
      const size_t eb = e - nf;
      const int kk = eb % kpb;
      const size_t r = eb / kpb;
      const int c = r % 16, ct = (r / 16) % nct, dir = r / ((size_t)16 * nct);
      const int kcell = ct * 16 + c, j = kk >> 2, g = kk & 3;
      Rwb[eb] = (kcell < p.no && j < p.no) ? v[p.p_off[dir][g] + j + (size_t)p.no * (1 + p.ni + kcell)] : 0.0f;
    }
  }
}
// recurrent weights of a wide layer as bf16 for the *_step_bf16 kernels of lstm_wide.h, zero padded:
//   Rbf[dir][rows_f][kp]   row = cell*4+slot : R_slot[cell][k]            (forward: B rows, k contiguous)
//   Rbb[dir][rows_b][kpb]  row = cell k      : R_slot[j][k] at col 4*j+slot (backward: contraction over (j, slot))
__global__ void k_pack_wide_bf16(const float* v, unsigned short* Rbf, unsigned short* Rbb, PackDesc p, int rows_f, int kp,
                                 int rows_b, int kpb) {
  const size_t nf = (size_t)p.ndir * rows_f * kp, nb = (size_t)p.ndir * rows_b * kpb;
  CLSTM_GRID_STRIDE(e, nf + nb) {
    float x = 0.0f;
    if (e < nf) {
      const int k = e % kp;
      const size_t q = e / kp;
      const int row = q % rows_f, dir = q / rows_f;
      const int cell = row >> 2, slot = row & 3;
      if (cell < p.no && k < p.no) x = v[p.p_off[dir][slot] + cell + (size_t)p.no * (1 + p.ni + k)];
      Rbf[e] = (unsigned short)(bf16_pack2(x, 0.0f) & 0xFFFFu);
    } else {
      const size_t e2 = e - nf;
      const int col = e2 % kpb;
      const size_t q = e2 / kpb;
      const int k = q % rows_b, dir = q / rows_b;
      const int j = col >> 2, slot = col & 3;
      if (j < p.no && k < p.no) x = v[p.p_off[dir][slot] + j + (size_t)p.no * (1 + p.ni + k)];
      Rbb[e2] = (unsigned short)(bf16_pack2(x, 0.0f) & 0xFFFFu);
    }
  }
}

// The backward layout with every weight as TWO bf16 terms (hi | lo, `plane` halfs apart): the B operand of the f32-grade backward
// recurrence on the bf16 MFMA (lstm_wide.h: lstm_xcd_bwd_x3)
__global__ void k_pack_wide_split(const float* v, unsigned short* Rb2, PackDesc p, int rows_b, int kpb, long long plane) {
  const size_t nb = (size_t)p.ndir * rows_b * kpb;
  CLSTM_GRID_STRIDE(e, nb) {
    float x = 0.0f;
    const int col = e % kpb;
    const size_t q = e / kpb;
    const int k = q % rows_b, dir = q / rows_b;
    const int j = col >> 2, slot = col & 3;
    if (j < p.no && k < p.no) x = v[p.p_off[dir][slot] + j + (size_t)p.no * (1 + p.ni + k)];
    const unsigned hi = bf16_pack2(x, 0.0f) & 0xFFFFu;
    Rb2[e] = (unsigned short)hi;
    Rb2[e + plane] = (unsigned short)(bf16_pack2(x - __builtin_bit_cast(float, hi << 16), 0.0f) & 0xFFFFu);
  }
}

// Every packed copy a WIDE layer needs in bf16 mode, in ONE pass over its parameters (no % 128 == 0, ni % 32 == 0: no padding
// anywhere): Wt / bias (f32, the hoisted product's fallback forms), Wtb [ni][M] and WtbT [M][ni] (bf16 W_x in both
// orientations), Rbf (forward recurrence: rows (cell, gate), k contiguous) and Rbb (backward: rows k, columns (cell, gate)).
// The five single-purpose kernels it replaces (k_pack_wide_bf16, k_pack_layer, k_to_bf16 x 2, k_transpose_to_bf16) read the
// column-major parameter blocks with a 2 KB stride between neighbouring threads wherever their output was row-major: 102 us
// per configs[4] step for 105 MB of traffic.  Here a workgroup moves a tile of 32 cells x 32 columns of all four gates
// through LDS: reads are 128-byte runs along the cells, both kinds of output are written in whole 32- / 256-byte runs.
__global__ __launch_bounds__(256) void k_pack_wide_tiles(const float* v, PackDesc p, float* Wt, float* bias, unsigned short* Wtb,
                                                        unsigned short* WtbT, unsigned short* Rbf, unsigned short* Rbb, int kf, int kb,
                                                        int rows_f, int rows_b) {
  __shared__ __attribute__((aligned(16))) float tile[32 * 132];   // [column][cell * 4 + gate], row stride 132
  const int no = p.no, ni = p.ni, M = p.ndir * 4 * no;
  const int nct = no >> 5, nkx = ni >> 5, nk = nkx + (no >> 5);
  int b = blockIdx.x;
  const int ct = b % nct; b /= nct;
  const int kt = b % nk, dir = b / nk;
  const bool isx = kt < nkx;
  const int k0 = (isx ? kt : kt - nkx) << 5, c0 = ct << 5;
  const int tid = threadIdx.x;
  {
    const int cell = tid & 31, q = tid >> 5;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int g = i & 3, cc = q + 8 * (i >> 2);
      x[i] = v[p.p_off[dir][g] + (c0 + cell) + (size_t)no * (1 + (isx ? 0 : ni) + k0 + cc)];
    }
#pragma unroll
    for (int i = 0; i < 16; i++) tile[(q + 8 * (i >> 2)) * 132 + cell * 4 + (i & 3)] = x[i];
  }
  if (isx && kt == 0 && tid < 128) bias[dir * 4 * no + 4 * c0 + tid] = v[p.p_off[dir][tid & 3] + c0 + (tid >> 2)];
  __syncthreads();
  {   // outputs whose rows are parameter COLUMNS: 16 consecutive (cell, gate) values of one column per thread
    const int cc = tid >> 3, ch = tid & 7;
    const int k = k0 + cc;
    float x[16];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(&tile[cc * 132 + ch * 16 + j * 4]);
      x[4 * j] = t[0]; x[4 * j + 1] = t[1]; x[4 * j + 2] = t[2]; x[4 * j + 3] = t[3];
    }
    float lo8[8], hi8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { lo8[j] = x[j]; hi8[j] = x[8 + j]; }
    const int col0 = 4 * (c0 + 4 * ch);          // 4 cell + gate of the first value
    if (isx) {
      float* wt = Wt + (size_t)k * M + dir * 4 * no + col0;
#pragma unroll
      for (int j = 0; j < 4; j++) *reinterpret_cast<f32x4*>(wt + 4 * j) = f32x4{x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
      unsigned short* wb = Wtb + (size_t)k * M + dir * 4 * no + col0;
      *reinterpret_cast<u16x8*>(wb) = bf16_pack8(lo8);
      *reinterpret_cast<u16x8*>(wb + 8) = bf16_pack8(hi8);
    } else {
      unsigned short* rb = Rbb + ((size_t)dir * rows_b + k) * kb + col0;
      *reinterpret_cast<u16x8*>(rb) = bf16_pack8(lo8);
      *reinterpret_cast<u16x8*>(rb + 8) = bf16_pack8(hi8);
    }
  }
  {   // outputs whose rows are (cell, gate): 16 consecutive columns of one row per thread
    const int r = tid >> 1, hf = tid & 1;
    float lo8[8], hi8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { lo8[j] = tile[(hf * 16 + j) * 132 + r]; hi8[j] = tile[(hf * 16 + 8 + j) * 132 + r]; }
    const size_t row = (size_t)4 * c0 + r;        // 4 cell + gate within the direction
    unsigned short* dst = isx ? WtbT + ((size_t)dir * 4 * no + row) * ni + k0 + hf * 16
                              : Rbf + ((size_t)dir * rows_f + row) * kf + k0 + hf * 16;
    *reinterpret_cast<u16x8*>(dst) = bf16_pack8(lo8);
    *reinterpret_cast<u16x8*>(dst + 8) = bf16_pack8(hi8);
  }
}

// S[dir][n][0] = 1, S[dir][n][1..ni] = x_n for every direction: the non-recurrent part of the source
// rows [1 | x_t | h_{t-1}] (forward_stack_delay + the bias column of Params, tensor.h:263-264)
__global__ void k_build_source(float* S, const float* x, size_t N, int ni, int ldx, int lds, int ndir, long long sdir) {
  CLSTM_GRID_STRIDE(e, N * (size_t)(1 + ni)) {
    const size_t n = e / (1 + ni);
    const int j = e % (1 + ni);
    const float v = j == 0 ? 1.0f : x[n * ldx + (j - 1)];
    for (int d = 0; d < ndir; d++) S[(size_t)d * sdir + n * lds + j] = v;
  }
}
// On-demand rebuilds of what a persistent bf16 forward pass of a wide layer does not store (lstm_wide.h: skip_h / skip_s),
// exact: the kernel's own h = tanh(c) * go from the stored state and output gate; the recurrent columns of the f32 source
// rows are the outputs of the line's previous own step (forward_stack_delay, clstm_compute.cc:398-410), 0 at its first.
__global__ void k_h_from_state(float* H, const float* G, const float* C, size_t N, int no, int ndir, int ldh, int hofs) {
  CLSTM_GRID_STRIDE(e, N * (size_t)ndir * no) {
    const size_t n = e / ((size_t)ndir * no);
    const int r = e % ((size_t)ndir * no);
    H[n * ldh + hofs + r] = tanh_fast(C[e]) * G[e * 4 + 2];   // (the bf16-mode kernels' own form of tanh)
  }
}
__global__ void k_source_h(float* S, const float* H, const int* line_off, int bs, size_t N, int no, int ndir, int ldh, int hofs,
                           int lds, int sofs, long long sdir) {
  CLSTM_GRID_STRIDE(e, N * (size_t)ndir * no) {
    const size_t n = e / ((size_t)ndir * no);
    const int r = e % ((size_t)ndir * no), dir = r / no, cell = r % no;
    int lo = 0, hi = bs;                       // line of frame n: line_off[lo] <= n < line_off[lo + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((size_t)line_off[mid] <= n) lo = mid; else hi = mid; }
    const long long off = line_off[lo], T = line_off[lo + 1] - off, t = (long long)n - off;
    const long long prev = dir == 0 ? t - 1 : t + 1;   // frame of the own step before
    S[(size_t)dir * sdir + n * lds + sofs + cell] = (prev >= 0 && prev < T) ? H[(off + prev) * ldh + hofs + r] : 0.0f;
  }
}
// g[off(b, c) + rs*r] = sum_z partial[b*nsplit + z][r][c]   (deterministic split-K reduction + row scatter;
// every parameter is produced by exactly one (b, r, c), so this assigns and g needs no clearing)
struct ReduceDesc {
  const float* partial;
  const long long* moff;
  long long base;
  int nsplit, nbatch, R, Cn, rs;
};
// The SGD update folded into the reduction (clstm_net_train_step without a communicator: one launch less per step): the
// thread that produces g[o] also applies k_update's arithmetic to element o -- every parameter is produced exactly once.
// The packed copies of a narrow layer's parameters (k_ingest_pack / k_pack_layer) kept current BY the fused update: inv[o * kd ..]
// lists the packed elements that are copies of parameter o (-1 ends the list), addressed as in the table form of k_ingest_pack
// ([W_x rows + bias | forward recurrence registers | backward ... | fused-launch forms]); the thread that moves v[o] rewrites
// them, and the next step's ingest launch carries no repack blocks (~2 us of every training step).
struct PackDst {
  const int* inv; int kd;   // null inv: the packed copies are not touched
  float *Wt, *bias, *Rf, *Rb;
  PackDesc p; PackFused pf;
  unsigned nwx, nr;
};
DEVFN void pack_dst_store(const PackDst& k, unsigned e, float x) {
  if (e < k.nwx) pack_wx_store(e, x, k.Wt, k.bias, k.p);
  else if (e < k.nwx + k.nr) k.Rf[e - k.nwx] = x;
  else if (e < k.nwx + 2 * k.nr) k.Rb[e - k.nwx - k.nr] = x;
  else pack_fused_store(e - k.nwx - 2 * k.nr, x, k.pf, k.p);
}
struct UpdateFuse {
  float* v; float* d;       // null v: reduce only
  float lr, mom, clip;
  const int* err;           // device error words (see k_update)
  int* step_word; int step_id;
  int* nanflag; int step_no; // non-finite gradient entries: see k_update (checked whether or not the update is fused: v may be null)
  PackDst pk;
};
DEVFN void reduce_scatter_one(const ReduceDesc& d, size_t e, float* g, const UpdateFuse& u, const bool apply) {
  const size_t RC = (size_t)d.R * d.Cn;
  const int b = e / RC;
  const int rc = e - b * RC;
  const int r = rc / d.Cn, c = rc % d.Cn;
  const float* p = d.partial + (size_t)b * d.nsplit * RC + rc;
  float s[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  int z = 0;
  for (; z + 8 <= d.nsplit; z += 8) {   // eight independent loads in flight (the kernel is latency-bound), fixed order
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = p[(size_t)(z + u) * RC];
#pragma unroll
    for (int u = 0; u < 8; u++) s[u] += x[u];
  }
  {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = z + u < d.nsplit ? p[(size_t)(z + u) * RC] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; u++) s[u] += x[u];
  }
  const long long o = (d.moff ? d.moff[(size_t)b * d.Cn + c] : d.base + c) + (long long)d.rs * r;
  const float gv = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  g[o] = gv;
  bool fin = true;
  if (u.nanflag && !f32_finite(gv)) { raise_nonfinite(u.nanflag, u.step_no); fin = false; }
  if (apply && fin) {
    float di = u.d[o] + gv;
    if (u.clip < 1e6f) di = fmaxf(-u.clip, fminf(u.clip, di));
    const float nv = u.v[o] + di * u.lr;
    u.v[o] = nv;
    u.d[o] = di * u.mom;
    if (u.pk.inv) {
      const int* ip = u.pk.inv + (size_t)o * u.pk.kd;
      for (int k = 0; k < u.pk.kd; k++) {
        const int e = ip[k];
        if (e < 0) break;
        pack_dst_store(u.pk, (unsigned)e, nv);
      }
    }
  }
}
// Bias row of a weight gradient whose product left it out (clstm_hip.hip: dw_bias_out): dbias [bs][ndir][Cn] holds, per line, the
// sum over its frames of the gate deltas (lstm_wide.h: LstmWideArgs::dbias); row 0 of direction d's FIRST slab takes their sum
// over the lines (in line order), row 0 of its other slabs zeros -- the slab reduction then finds W.d[:,0] += sum_b y.d
// (clstm_compute.cc:301) where the 1537th row of the product used to put it.
// (eight lines in flight per thread, eight interleaved sums combined pairwise: the loop is a chain of memory latencies -- one
// line at a time it took ~30 us per layer at configs[4], 64 lines x 4096 columns on sixteen workgroups)
__global__ void k_bias_rows(const float* dbias, float* partial, int bs, int ndir, int nsplit, int R, int Cn) {
  CLSTM_GRID_STRIDE(e, (size_t)ndir * Cn) {
    const int dir = e / Cn, c = e % Cn;
    const float* p = dbias + (size_t)dir * Cn + c;
    const size_t ls = (size_t)ndir * Cn;
    float s[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int b = 0; b < bs; b += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = b + u < bs ? p[(size_t)(b + u) * ls] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; u++) s[u] += x[u];
    }
    const float t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    for (int z = 0; z < nsplit; z++) partial[(((size_t)dir * nsplit + z) * R) * Cn + c] = z == 0 ? t : 0.0f;
  }
}
// up to two slab sets per launch (the softmax layer's weight gradient rides with the top LSTM layer's)
DEVFN void reduce_scatter_blocks(const ReduceDesc& d0, const ReduceDesc& d1, float* g, int* zero, const int zero_n, const UpdateFuse& u, const unsigned nb) {
  // (house-keeping that rides this launch: the work-queue heads of the fused backward launch return to zero)
  if (zero && blockIdx.x == 0 && (int)threadIdx.x < zero_n) zero[threadIdx.x] = 0;
  if (u.step_word && blockIdx.x == 0 && threadIdx.x == 0) store_i32_wt(u.step_word, u.step_id);
  const bool apply = u.v && !dev_err_set(u.err);   // (a failed launch / a non-finite gradient earlier on: the gradient is not applied)
  const size_t n0 = (size_t)d0.R * d0.Cn * d0.nbatch, n1 = (size_t)d1.R * d1.Cn * d1.nbatch;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n0 + n1; e += (size_t)nb * blockDim.x) {
    if (e < n0) reduce_scatter_one(d0, e, g, u, apply);
    else reduce_scatter_one(d1, e - n0, g, u, apply);
  }
}
__global__ void k_reduce_scatter(ReduceDesc d0, ReduceDesc d1, float* g, int* zero, int zero_n, UpdateFuse u) {
  reduce_scatter_blocks(d0, d1, g, zero, zero_n, u, gridDim.x);
}
// diagnostics: the cross-lane primitives applied to the lane index (tests/test_intrinsics.py)
__global__ void k_debug_lane_ops(float* out) {
  const int lane = threadIdx.x & 63;
  const float x = (float)lane;
  out[0 * 64 + lane] = quad_xor1(x);
  out[1 * 64 + lane] = quad_xor2(x);
  out[2 * 64 + lane] = quad_bcast<0>(x);
  out[3 * 64 + lane] = quad_bcast<1>(x);
  out[4 * 64 + lane] = quad_bcast<2>(x);
  out[5 * 64 + lane] = quad_bcast<3>(x);
  out[6 * 64 + lane] = row_ror<1>(x);
  out[7 * 64 + lane] = row_ror<4>(x);
  out[8 * 64 + lane] = row_ror<8>(x);
  out[9 * 64 + lane] = row_half_mirror(x);
  out[10 * 64 + lane] = wave_shr1(x);
}
// device-resident input frames: one pass copies them into the net's input block AND lays down the
// first layer's source rows (replaces a D2D memcpy followed by k_build_source)
__global__ void k_ingest(const float* x, float* X, float* S, size_t N, int ni, int lds, int ndir, long long sdir) {
  CLSTM_GRID_STRIDE(e, N * (size_t)(1 + ni)) {
    const size_t n = e / (1 + ni);
    const int j = e % (1 + ni);
    float v = 1.0f;
    if (j > 0) {
      v = x[n * ni + (j - 1)];
      X[n * ni + (j - 1)] = v;
    }
    for (int d = 0; d < ndir; d++) S[(size_t)d * sdir + n * lds + j] = v;
  }
}
// f32 expansion of a bf16 array (exact): the gate deltas for a fallback product when the persistent backward recurrence
// stored only their bf16 form
__global__ void k_bf16_to_f32(const unsigned short* src, float* dst, size_t n) {
  CLSTM_GRID_STRIDE(e, n) dst[e] = __builtin_bit_cast(float, (unsigned)src[e] << 16);
}
// bf16 copy of an f32 array (weight operands of the bf16-source GEMM, gemm_bf16.h)
__global__ void k_to_bf16(const float* src, unsigned short* dst, size_t n) {
  CLSTM_GRID_STRIDE(e, n) dst[e] = (unsigned short)(bf16_pack2(src[e], 0.0f) & 0xFFFFu);
}

// x-part and bias column of the bf16 source rows [x_t | h_{t-1} | 1 | 0 ..] (pitch ldsb, one array per direction; ni and
// one_col = ni + no multiples of 8): the contraction-major operand of the weight-gradient GEMM (gemm_b16mc).  x comes from
// the layer below's bf16 outputs (xb, 16-byte copies) or from f32 input frames (xf); the h-part is stored by the
// persistent forward recurrence itself (lstm_wide.h).
// the constant part of those rows alone: the bias column (1, followed by the row's zero padding) -- written once per batch
// geometry when the x-part is not copied at all (the weight-gradient GEMM then reads it from the layer below's bf16 outputs)
__global__ void k_source_one_bf16(unsigned short* Sbf, size_t N, int one_col, int ldsb, int ndir, long long sbdir) {
  CLSTM_GRID_STRIDE(n, N) {
    u16x8 v;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (unsigned short)(i == 0 ? 0x3F80 : 0);
    for (int d = 0; d < ndir; d++) *reinterpret_cast<u16x8*>(&Sbf[(size_t)d * sbdir + n * ldsb + one_col]) = v;
  }
}
__global__ void k_source_x_bf16(unsigned short* Sbf, const float* xf, const unsigned short* xb, int ldx, size_t N, int ni, int one_col,
                                int ldsb, int ndir, long long sbdir) {
  const int cpr = (ni >> 3) + 1;
  CLSTM_GRID_STRIDE(e, N * (size_t)cpr) {
    const size_t n = e / cpr;
    const int j = (int)(e - n * cpr);
    u16x8 v;
    int col = j * 8;
    if (j < (ni >> 3)) {
      if (xb) v = *reinterpret_cast<const u16x8*>(&xb[n * ldx + col]);
      else {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = xf[n * ldx + col + i];
        v = bf16_pack8(x);
      }
    } else {
      col = one_col;
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = (unsigned short)(i == 0 ? 0x3F80 : 0);
    }
    for (int d = 0; d < ndir; d++) *reinterpret_cast<u16x8*>(&Sbf[(size_t)d * sbdir + n * ldsb + col]) = v;
  }
}

// dst[c][k] = bf16(src[k][c]): the k-contiguous bf16 form of W_x ([ni][M] -> [M][ni]) for the bf16-source W_x.x product
__global__ void k_transpose_to_bf16(const float* src, unsigned short* dst, int rows, int cols) {
  CLSTM_GRID_STRIDE(e, (size_t)rows * cols) {
    const int c = (int)(e / rows), k = (int)(e % rows);
    dst[e] = (unsigned short)(bf16_pack2(src[(size_t)k * cols + c], 0.0f) & 0xFFFFu);
  }
}

// k_ingest and the first layer's k_pack_layer in ONE launch (a single narrow layer whose parameters changed since
// the last pack -- every training step): blocks [0, nbi) ingest, the rest repack.  The two jobs are independent
// and each is far too small to fill the chip, so one launch ramp / tail instead of two.
// the ingest blocks' job (blocks blk of nbi): frames -> the net's input block X and the first layer's source rows [1 | x] per direction
DEVFN void ingest_rows(const float* x, float* X, float* S, const size_t N, const int ni, const int lds, const int ndir, const long long sdir,
                       const unsigned blk, const unsigned nbi) {
  if ((ni & 3) == 0 && (lds & 3) == 0 && ((size_t)x & 15) == 0) {
    // 16 bytes per thread: chunk c of a frame = x[4c .. 4c+3] -> X as it is, and -- shifted by the bias column -- floats
    // 4c .. 4c+3 of the source row [1 | x] = (x[4c-1] or the 1, x[4c], x[4c+1], x[4c+2]); the last chunk holds x[ni-1] alone
    const size_t nch = (size_t)ni / 4 + 1;
    for (size_t e = (size_t)blk * blockDim.x + threadIdx.x; e < N * nch; e += (size_t)nbi * blockDim.x) {
      const size_t n = e / nch;
      const int c = (int)(e - n * nch);
      const float prev = c == 0 ? 1.0f : x[n * ni + 4 * c - 1];
      if (4 * c < ni) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + n * ni + 4 * c);
        *reinterpret_cast<f32x4*>(X + n * ni + 4 * c) = xv;
        const f32x4 sv = f32x4{prev, xv[0], xv[1], xv[2]};
        for (int d = 0; d < ndir; d++) *reinterpret_cast<f32x4*>(S + (size_t)d * sdir + n * lds + 4 * c) = sv;
      } else {
        for (int d = 0; d < ndir; d++) S[(size_t)d * sdir + n * lds + 4 * c] = prev;
      }
    }
  } else {
    for (size_t e = (size_t)blk * blockDim.x + threadIdx.x; e < N * (size_t)(1 + ni); e += (size_t)nbi * blockDim.x) {
      const size_t n = e / (1 + ni);
      const int j = e % (1 + ni);
      float val = 1.0f;
      if (j > 0) {
        val = x[n * ni + (j - 1)];
        X[n * ni + (j - 1)] = val;
      }
      for (int d = 0; d < ndir; d++) S[(size_t)d * sdir + n * lds + j] = val;
    }
  }
}
// optional trailing blocks (blk counts from the first of them): small host arrays, straight from their pinned slots
// (one element per thread: a read of host memory takes microseconds, so they must all be in flight at once)
DEVFN void ingest_small(int blk, const int* lo_src, int* lo_dst, const int lo_n, const int* aux_src, int* aux_dst, const int aux_n) {
  if (lo_src) {                       //   the line offsets: first trailing block
    if (blk == 0) {
      for (int i = threadIdx.x; i < lo_n; i += blockDim.x) lo_dst[i] = lo_src[i];
      return;
    }
    blk--;
  }
  const int i = blk * (int)blockDim.x + (int)threadIdx.x;   //   the CTC metadata of this training step
  if (i < aux_n) aux_dst[i] = aux_src[i];
}
__global__ void k_ingest_pack(const float* x, float* X, float* S, size_t N, int ni, int lds, int ndir, long long sdir,
                              int nbi, int nbp, const float* v, float* Wt, float* bias, float* Rf, float* Rb, PackDesc p, PackFused pf,
                              const int* lo_src, int* lo_dst, int lo_n, const int* aux_src, int* aux_dst, int aux_n, const int* tab) {
  if ((int)blockIdx.x >= nbi + nbp) { ingest_small((int)blockIdx.x - (nbi + nbp), lo_src, lo_dst, lo_n, aux_src, aux_dst, aux_n); return; }
  if ((int)blockIdx.x < nbi) {
    ingest_rows(x, X, S, N, ni, lds, ndir, sdir, blockIdx.x, (unsigned)nbi);
  } else {
    const size_t nwx = (size_t)(1 + p.ni) * p.ndir * 4 * p.no;
    const size_t nr = (size_t)p.ndir * 4 * 4 * p.nk4 * p.nthreads;
    const size_t nf = pack_fused_count(pf, p);
    for (size_t e = (size_t)(blockIdx.x - nbi) * blockDim.x + threadIdx.x; e < nwx + 2 * nr + nf; e += (size_t)nbp * blockDim.x) {
      if (tab) {   // source indices from the table built once per net (k_pack_index)
        const int si = tab[e];
        const float val = si >= 0 ? v[si] : 0.0f;
        if (e < nwx) pack_wx_store(e, val, Wt, bias, p);
        else if (e < nwx + nr) Rf[e - nwx] = val;
        else if (e < nwx + 2 * nr) Rb[e - nwx - nr] = val;
        else pack_fused_store(e - nwx - 2 * nr, val, pf, p);
      }
      else if (e < nwx) pack_wx(e, v, Wt, bias, p);
      else if (e < nwx + nr) pack_rf(e - nwx, v, Rf, p);
      else if (e < nwx + 2 * nr) pack_rb(e - nwx - nr, v, Rb, p);
      else pack_fused(e - nwx - 2 * nr, v, pf, p);
    }
  }
}
// The NEXT minibatch's ingest riding the last launch of a training step (abi.inc: clstm_net_train_step_next): blocks
// [0, nb_main) reduce and update, the nbi blocks behind them copy the next frames into the input block and lay down the source
// rows -- the forward pass of step k read them for the last time a whole backward pass ago --, the blocks behind those fetch the
// next step's line offsets and CTC metadata from their pinned slots.  One launch less on the step's critical path (k_ingest_pack:
// 5.7 us in front of the forward launch of a 64 x 200 step).
struct IngestTail {
  const float* x; float* X; float* S; unsigned long long N; int ni, lds, ndir; long long sdir;
  int nb_main, nbi;
  const int* lo_src; int* lo_dst; int lo_n; const int* aux_src; int* aux_dst; int aux_n;
};
__global__ void k_reduce_scatter_ingest(ReduceDesc d0, ReduceDesc d1, float* g, int* zero, int zero_n, UpdateFuse u, IngestTail t) {
  if ((int)blockIdx.x < t.nb_main) { reduce_scatter_blocks(d0, d1, g, zero, zero_n, u, (unsigned)t.nb_main); return; }
  const int blk = (int)blockIdx.x - t.nb_main;
  if (blk < t.nbi) { ingest_rows(t.x, t.X, t.S, (size_t)t.N, t.ni, t.lds, t.ndir, t.sdir, (unsigned)blk, (unsigned)t.nbi); return; }
  ingest_small(blk - t.nbi, t.lo_src, t.lo_dst, t.lo_n, t.aux_src, t.aux_dst, t.aux_n);
}
__global__ void k_fill_col0(float* H, size_t rows, int ld, int col) {
  CLSTM_GRID_STRIDE(e, rows) H[e * ld + col] = 1.0f;
}
// gather rows of a strided plane: out[n][c] = src[n*ld + c]
__global__ void k_gather_rows(const float* src, float* out, size_t N, int no, int ld) {
  CLSTM_GRID_STRIDE(e, N * no) {
    const size_t n = e / no;
    out[e] = src[n * ld + (e % no)];
  }
}
// gather one NPLSTM state plane into [N][no] for the parity tests
__global__ void k_gather_state(const float* src, float* out, size_t N, int no, int ndir, int dir, int slot) {
  CLSTM_GRID_STRIDE(e, N * no) {
    const size_t n = e / no;
    const int c = e % no;
    out[e] = slot < 0 ? src[(n * ndir + dir) * no + c] : src[(n * ndir + dir) * 4 * no + c * 4 + slot];
  }
}

}  // namespace clstm
