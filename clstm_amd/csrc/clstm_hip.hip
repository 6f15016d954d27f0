// clstm_hip.hip -- host orchestration + C ABI (include/clstm_abi.h) of the MI355X hot path.
//
// The fused network mirrors Stacked{Parallel{NPLSTM, Reversed{NPLSTM}} x L, SoftmaxLayer}
// (clstm_prefab.cc:52-109, clstm.cc:391-656) on packed minibatches of text lines:
//   forward : per layer  G = W_x.x + b   (one MFMA GEMM over every frame of the minibatch)
//                        lstm_fwd_kernel (persistent recurrence, one workgroup per line x dir)
//             softmax    Z = W1.[h_f;h_r] + b (MFMA GEMM) ; limexp / column normalise
//   CTC     : ctc_align_kernel (one workgroup per line), deltas = aligned - Z
//   backward: softmax dX (MFMA GEMM), dW1 (split-K MFMA GEMM over frames)
//             per layer lstm_bwd_kernel ; dW = delta.[1;x;h_prev]^T (split-K MFMA GEMM) ;
//             dX = W_x^T.delta (MFMA GEMM, only where a consumer exists)
//   update  : k_update on the flat reference-layout buffers, then re-pack kernel weights.
// Stacked/Parallel/Reversed never copy: they are pointer hand-offs and index arithmetic.
#include "../../include/clstm_abi.h"
#include "dbgopt.h"
#include "ctc.h"
#include "devintrin.h"
#include "gemm_mfma.h"
#include "gemm_bf16.h"
#include "gemm_dw.h"
#include "lstm_bwd_dw.h"
#include "lstm_fwd_fused.h"
#include "softmax_fused.h"
#include "lstm_seq.h"
#include "lstm_wide.h"
#include "lstm_mfma.h"
#include "lstm_mfma_bwd.h"
#include "ops.h"

#include <algorithm>
#include <chrono>
#include <sched.h>
#include <cstring>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace clstm {

static thread_local std::string g_err;
static thread_local hipStream_t g_stream = nullptr;

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
#define HIPCHECK(expr)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      throw Error(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" + __FILE__ + \
                  ":" + std::to_string(__LINE__) + ")");                                     \
  } while (0)
#define REQUIRE(cond, msg)                    \
  do {                                        \
    if (!(cond)) throw Error(std::string(msg)); \
  } while (0)
#define ABI_BEGIN try {
#define ABI_END                         \
  return 0;                             \
  }                                     \
  catch (const std::exception& e) {     \
    g_err = e.what();                   \
    return 1;                           \
  }                                     \
  catch (...) {                         \
    g_err = "unknown error";            \
    return 1;                           \
  }

static inline int nblocks(size_t n, int bs = 256) {
  size_t b = (n + bs - 1) / bs;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}
static void check_launch() { HIPCHECK(hipGetLastError()); }

// zero-fill ordered before everything that follows on ANY stream (see DevBuf::reserve)
static void zero_fill(void* p, size_t bytes) {
  HIPCHECK(hipMemsetAsync(p, 0, bytes, g_stream));
  HIPCHECK(hipStreamSynchronize(g_stream));
}
template <class T>
struct DevBuf {  // grow-only device buffer
  T* p = nullptr;
  size_t cap = 0;
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) HIPCHECK(hipFree(p));
    p = nullptr;
    size_t want = n + n / 4 + 64;
    HIPCHECK(hipMalloc((void**)&p, want * sizeof(T)));
    // zero-fill on the library stream and wait for it: whatever initialises parts of the fresh buffer next --
    // kernels on this stream (k_fill_col0 ...) or a blocking null-stream copy (the CTC tables) -- is then ordered
    // behind the fill without relying on how a null-stream hipMemset synchronises with a non-blocking stream.
    // Buffers only grow, so this drain happens a handful of times per process (hipFree drains the device anyway).
    zero_fill(p, want * sizeof(T));
    cap = want;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Pinned host staging ring: small per-minibatch host arrays (line offsets, CTC target states) are
// copied here and DMA'd asynchronously, so declaring a batch never drains the stream and the host
// can run ahead of the GPU.  A slot is reused only after the copy recorded on it has completed.
struct PinnedRing {
  // An event record between two kernels costs ~5 us of stream time on MI355X (a barrier packet the next dispatch waits
  // for; profiles/r02_timeline_one_stream.txt of the first version shows the gaps), so slots are recycled in GROUPS:
  // one event per GROUP commits, recorded behind the group's last copy and waited for when the ring comes round to
  // the group's first slot again.
  static const int SLOTS = 32, GROUP = 8;
  void* h[SLOTS] = {};
  size_t cap[SLOTS] = {};
  hipEvent_t ev[SLOTS / GROUP] = {};
  bool busy[SLOTS / GROUP] = {};
  int cur = SLOTS - 1;
  void* acquire(size_t bytes) {
    cur = (cur + 1) % SLOTS;
    const int g = cur / GROUP;
    if (cur % GROUP == 0 && busy[g]) { HIPCHECK(hipEventSynchronize(ev[g])); busy[g] = false; }
    if (cap[cur] < bytes) {
      if (h[cur]) (void)hipHostFree(h[cur]);
      cap[cur] = bytes * 2 + 256;
      HIPCHECK(hipHostMalloc(&h[cur], cap[cur]));
    }
    return h[cur];
  }
  void commit(hipStream_t s) {   // the copy / kernel reading the current slot has been enqueued on s
    if (cur % GROUP != GROUP - 1) return;
    const int g = cur / GROUP;
    if (!ev[g]) HIPCHECK(hipEventCreateWithFlags(&ev[g], hipEventDisableTiming));
    HIPCHECK(hipEventRecord(ev[g], s));
    busy[g] = true;
  }
  ~PinnedRing() {
    for (int i = 0; i < SLOTS / GROUP; i++)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    for (int i = 0; i < SLOTS; i++)
      if (h[i]) (void)hipHostFree(h[i]);
  }
};

// ---- GEMM operand functors ---------------------------------------------------------------------
// operator(): one element; row4(): four consecutive columns of one row (c % 4 == 0) when vec4() says the
// destination rows are 16-byte aligned -- the GEMM epilogue then writes whole 256-byte row segments
struct StoreBias {  // out[r*ld + c] = val + bias[c]
  float* out; long long ld; const float* bias;
  DEVMFN void operator()(int r, int c, float v, int) const { out[(long long)r * ld + c] = v + bias[c]; }
  DEVMFN bool vec4() const { return (ld & 3) == 0 && (((size_t)out | (size_t)bias) & 15) == 0; }
  DEVMFN void row4(int r, int c, f32x4 v, int) const {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c);
    f32x4 o;
    o[0] = v[0] + b[0]; o[1] = v[1] + b[1]; o[2] = v[2] + b[2]; o[3] = v[3] + b[3];
    *reinterpret_cast<f32x4*>(out + (long long)r * ld + c) = o;
  }
};
static long long g_path_count[24];   // clstm_debug_path_count (diagnostics)
struct StorePlain {
  float* out; long long ld;
  DEVMFN void operator()(int r, int c, float v, int) const { out[(long long)r * ld + c] = v; }
  DEVMFN bool vec4() const { return (ld & 3) == 0 && ((size_t)out & 15) == 0; }
  DEVMFN void row4(int r, int c, f32x4 v, int) const { *reinterpret_cast<f32x4*>(out + (long long)r * ld + c) = v; }
};
struct StorePartial {  // split-K slabs [z][R][Cn]
  float* out; int R, Cn;
  DEVMFN void operator()(int r, int c, float v, int z) const { out[((long long)z * R + r) * Cn + c] = v; }
  DEVMFN bool vec4() const { return (Cn & 3) == 0 && ((size_t)out & 15) == 0; }
  DEVMFN void row4(int r, int c, f32x4 v, int z) const {
    *reinterpret_cast<f32x4*>(out + ((long long)z * R + r) * Cn + c) = v;
  }
};
struct StorePartialShift {  // split-K slabs of a product WITHOUT the bias row: operand rows [x | h], row r lands in row r + 1 of [1 | x | h] (R = rows of the slab)
  float* out; int R, Cn;
  DEVMFN void operator()(int r, int c, float v, int z) const { out[((long long)z * R + r + 1) * Cn + c] = v; }
  DEVMFN bool vec4() const { return (Cn & 3) == 0 && ((size_t)out & 15) == 0; }
  DEVMFN void row4(int r, int c, f32x4 v, int z) const { *reinterpret_cast<f32x4*>(out + ((long long)z * R + r + 1) * Cn + c) = v; }
};
struct StorePartialRot {  // split-K slabs whose operand rows were ordered [x | h | 1]: row r lands in row (r + 1) mod R of [1 | x | h]
  float* out; int R, Cn;
  DEVMFN void operator()(int r, int c, float v, int z) const { out[((long long)z * R + (r + 1 == R ? 0 : r + 1)) * Cn + c] = v; }
  DEVMFN bool vec4() const { return (Cn & 3) == 0 && ((size_t)out & 15) == 0; }
  DEVMFN void row4(int r, int c, f32x4 v, int z) const {
    *reinterpret_cast<f32x4*>(out + ((long long)z * R + (r + 1 == R ? 0 : r + 1)) * Cn + c) = v;
  }
};
#ifndef GEMM_BK_DW
#define GEMM_BK_DW 16   // frames staged per barrier pair in the weight-gradient GEMM (32 measured slower: 61.1 vs 58.3 us)
#endif
static const int kNK4Table[] = {1, 2, 4, 7, 8};
static int pick_nk4(int no) {
  int need = ((no + 3) / 4 + 3) / 4;
  for (int v : kNK4Table)
    if (v >= need) return v;
  return -1;
}
template <int NK4, int KU>
static void launch_fwd(const LstmSeqArgs& a, int bs, int nthreads, hipStream_t s) {
  const size_t smem = (2 * 4 * (size_t)lstm_qstride(NK4) + 4) * sizeof(float);
  CLSTM_LAUNCH((lstm_fwd_kernel<NK4, KU>), dim3(bs, a.ndir), dim3(nthreads), smem, s, a);
}
template <int NK4, int KU>
static void launch_bwd(const LstmSeqArgs& a, int bs, int nthreads, hipStream_t s) {
  const size_t smem = (2 * 16 * (size_t)lstm_qstride(NK4) + 4) * sizeof(float);
  CLSTM_LAUNCH((lstm_bwd_kernel<NK4, KU>), dim3(bs, a.ndir), dim3(nthreads), smem, s, a);
}
template <int NK4, int KU>
static void launch_fwd_fused(const FwdFusedKernelArgs& k, unsigned nblk, int nthreads, hipStream_t s) {
  const size_t smem = (2 * 4 * (size_t)lstm_qstride(NK4) + 4) * sizeof(float);
#ifdef CLSTM_HIP_EMU
  CLSTM_LAUNCH_COOP((lstm_fwd_fused_kernel<NK4, KU>), dim3(nblk), dim3(nthreads), smem, s, k);   // emulator: every workgroup live at once
#else
  CLSTM_LAUNCH((lstm_fwd_fused_kernel<NK4, KU>), dim3(nblk), dim3(nthreads), smem, s, k);
#endif
}
static bool launch_lstm_fwd_fused(int nk4, int ku, const FwdFusedKernelArgs& k, unsigned nblk, int nthreads, hipStream_t s) {
#define CASE_(N, K) if (nk4 == N && ku == K) { launch_fwd_fused<N, K>(k, nblk, nthreads, s); check_launch(); return true; }
  CASE_(7, 25) CASE_(7, 28) CASE_(8, 32)    // (the fused form needs >= 5 waves: a polling wave among the first four, a reporting wave behind them)
#undef CASE_
  return false;
}
// k values per lane actually used: the padded 4*nk4 in general, exact for the 97..100-cell case (uw3 BiLSTM(100))
static int pick_ku(int no, int nk4) { return (nk4 == 7 && (no + 3) / 4 == 25) ? 25 : 4 * nk4; }
static void launch_lstm(bool fwd, int nk4, int ku, LstmSeqArgs a, int bs, int nthreads, hipStream_t s) {
#define CASE_(N, K) if (nk4 == N && ku == K) { if (fwd) launch_fwd<N, K>(a, bs, nthreads, s); else launch_bwd<N, K>(a, bs, nthreads, s); check_launch(); return; }
  CASE_(1, 4) CASE_(2, 8) CASE_(4, 16) CASE_(7, 28) CASE_(7, 25) CASE_(8, 32)
#undef CASE_
  throw Error("unsupported nhidden for the register-resident recurrence");
}

template <int NK4, int KU>
static void launch_bwd_dw(const LstmSeqArgs& a, const GemmDwArgs& g, int nrec, unsigned ngemm, int nthreads, hipStream_t s) {
  const size_t smem = (2 * 16 * (size_t)lstm_qstride(NK4) + 4) * sizeof(float);
  if (g.x3 && g.terms >= 3) CLSTM_LAUNCH((lstm_bwd_dw_kernel<NK4, KU, 3>), dim3(nrec + ngemm), dim3(nthreads), smem, s, a, g, nrec);
  else if (g.x3) CLSTM_LAUNCH((lstm_bwd_dw_kernel<NK4, KU, 2>), dim3(nrec + ngemm), dim3(nthreads), smem, s, a, g, nrec);
  else CLSTM_LAUNCH((lstm_bwd_dw_kernel<NK4, KU, 0>), dim3(nrec + ngemm), dim3(nthreads), smem, s, a, g, nrec);
}
static bool launch_lstm_bwd_dw(int nk4, int ku, const LstmSeqArgs& a, const GemmDwArgs& g, int nrec, unsigned ngemm, int nthreads, hipStream_t s) {
#define CASE_(N, K) if (nk4 == N && ku == K) { launch_bwd_dw<N, K>(a, g, nrec, ngemm, nthreads, s); check_launch(); return true; }
  CASE_(7, 25) CASE_(7, 28) CASE_(4, 16) CASE_(8, 32)    // (thread count must cover the GEMM role's 256)
#undef CASE_
  return false;
}

// lock-step recurrence (lstm_wide.h): one persistent launch for the whole sequence when every workgroup
// can be resident at once (grid <= CU count, weights fit LDS), else one launch per time step
static int device_cu_count() {
#ifndef CLSTM_HIP_EMU
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return v;
  }();
  return n;
#else
  static const int n = getenv("CLSTM_EMU_CUS") ? atoi(getenv("CLSTM_EMU_CUS")) : 16;   // emulator: keeps cooperative test grids small
  return n;
#endif
}
template <class K>
static void coop_set_smem(K kernel, size_t smem) {
#ifndef CLSTM_HIP_EMU
  HIPCHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#endif
}
// The per-step launches of one sequence pass are a launch-bound inner loop of up to a few hundred
// dependent kernels: captured once into a hipGraph and replayed while the batch geometry and the buffers
// stay the same (the per-line lengths live in device memory, so only tmax / bs / pointers key the graph).
struct StepGraphCache {
  bool unsupported = false;   // the stream refused capture once: keep launching directly
#ifndef CLSTM_HIP_EMU
  struct Entry { std::vector<char> key; int kind; int tmax; hipGraphExec_t exec; };
  std::vector<Entry> entries;
  // key: the kernel argument struct with the per-step fields zeroed (pointers, geometry), kind: which pass / kernel family
  hipGraphExec_t find(int kind, const std::vector<char>& key, int tmax) {
    for (auto& e : entries)
      if (e.kind == kind && e.tmax == tmax && e.key == key) return e.exec;
    return nullptr;
  }
  void put(int kind, const std::vector<char>& key, int tmax, hipGraphExec_t exec) {
    if (entries.size() >= 16) { (void)hipGraphExecDestroy(entries.front().exec); entries.erase(entries.begin()); }
    entries.push_back(Entry{key, kind, tmax, exec});
  }
  ~StepGraphCache() { for (auto& e : entries) (void)hipGraphExecDestroy(e.exec); }
#endif
};
template <class A>
static std::vector<char> graph_key(A a) {
  a.step = 0;
  std::vector<char> k(sizeof(A));
  memcpy(k.data(), &a, sizeof(A));
  return k;
}
template <class A, class F>
static void launch_steps(StepGraphCache& cache, int kind, const A& a, int tmax, hipStream_t s, F&& body) {
#ifndef CLSTM_HIP_EMU
  const bool use_graph = dbg_opt("wide_graph", 1) != 0;
  if (use_graph && tmax >= 8) {
    const std::vector<char> key = graph_key(a);
    hipGraphExec_t exec = cache.find(kind, key, tmax);
    if (!exec && !cache.unsupported) {
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();      // e.g. the legacy default stream cannot be captured: plain launches
        cache.unsupported = true;
      } else {
        body();
        HIPCHECK(hipStreamEndCapture(s, &graph));
        HIPCHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIPCHECK(hipGraphDestroy(graph));
        cache.put(kind, key, tmax, exec);
      }
    }
    if (exec) {
      HIPCHECK(hipGraphLaunch(exec, s));
      return;
    }
  }
#endif
  body();
}
// Outcome of the persistent (one launch per pass) recurrence kernels.  The first launches of a process are checked
// synchronously -- a failed placement check (error word 1: nothing written) falls back to the per-step launches for
// good.  Later launches copy their error word into a pinned ring and are checked when the slot comes round again or at
// the next host read-back: the host keeps enqueueing ahead of the GPU (four stream synchronisations per minibatch cost
// ~0.1 ms of idle GPU at the configs[4] shape).
// Device error words shared by every net of the process: [0] sticky outcome of persistent recurrence launches,
// [1] weight-gradient items of the fused backward launch that gave up waiting (gemm_dw.h), [2] waits of the fused FORWARD
// launch that gave up (lstm_fwd_fused.h, lstm_seq.h:wait_chunk), [3] number of the training step whose gradient had a
// non-finite entry (ops.h:k_update; the reference asserts on NaN in every backward step, clstm.cc:630-649), [6] device-side
// peer-barrier waits that timed out (ops.h:k_peer_barrier), [7] the training step at which the replica check found the ranks'
// parameters different (ops.h:k_replica_verify); [4], [5] unused.  The update
// kernels skip the update while any is set; the host throws at its next synchronisation point (clstm_synchronize, any
// read-back).
static int* g_dev_err = nullptr;
static int* dev_err_words() {
  if (!g_dev_err) {
    HIPCHECK(hipMalloc((void**)&g_dev_err, 16 * sizeof(int)));
    zero_fill(g_dev_err, 16 * sizeof(int));
  }
  return g_dev_err;
}
static bool g_xcd_failed = false;   // a persistent launch failed its placement check: wide layers use the per-step launches from now on
struct XcdOutcome {
  // The persistent kernels publish their own outcome (lstm_wide.h:xcd_finish): the last workgroup to leave stores the launch's
  // error word into a pinned host word and -- if non-zero -- into the sticky device word the update kernels look at.  The host
  // marks the slot pending (-1) before the launch; a slot is looked at when the ring comes round to it again (16 launches
  // later: long finished) or at a synchronisation point.  No event record, no extra kernel on the stream.
  static const int SLOTS = 16;
  volatile int* pinned = nullptr;
  bool pending[SLOTS] = {};
  int next = 0, verified = 0;
  void check_slot(int i) {
    if (!pending[i]) return;
    if (pinned[i] == -1) {   // (only when the ring wraps within one un-synchronised burst: wait for that launch)
      const auto t0 = std::chrono::steady_clock::now();
      while (pinned[i] == -1 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20)) sched_yield();
      if (pinned[i] == -1) { HIPCHECK(hipDeviceSynchronize()); }
    }
    pending[i] = false;
    const int e = pinned[i];
    if (e > 0 && g_dev_err) (void)hipMemset(g_dev_err, 0, sizeof(int));   // reported once: updates resume
    if (e == 1) g_xcd_failed = true;   // not all workgroups were resident (another tenant on the device): per-step launches from now on
    if (e > 0)
      throw Error(e == 1 ? "persistent recurrence: the workgroups of a later launch were not all resident / not spread evenly over the XCDs (another process or stream on the device?); "
                           "the minibatches enqueued since then were NOT applied -- every later update was skipped; in a multi-layer net the layers above the failing one may "
                           "have taken their update of that one minibatch -- and the library has switched to the per-step launches (CLSTM_XCD_REC=0) for the rest of the process"
                         : "persistent recurrence: a group barrier timed out in the middle of the sequence; the minibatches enqueued since then were NOT applied (layers above the "
                           "failing one may have taken their update of that one minibatch) -- set CLSTM_XCD_REC=0");
  }
  void check_all() { for (int i = 0; i < SLOTS; i++) check_slot(i); }
  // before a launch: the pinned word the kernel reports to (null while launches are still verified synchronously / on the emulator)
  int* prepare() {
#ifdef CLSTM_HIP_EMU
    return nullptr;
#else
    if (verified < 4) return nullptr;
    if (!pinned) { int* p = nullptr; HIPCHECK(hipHostMalloc((void**)&p, SLOTS * sizeof(int))); pinned = p; }
    const int i = next;
    next = (next + 1) % SLOTS;
    check_slot(i);
    pinned[i] = -1;
    pending[i] = true;
    return (int*)(pinned + i);
#endif
  }
  // after it; returns false if the launch failed its placement check and nothing was written (synchronous phase only).
  // last_err_d: the launch's surviving outcome word (XcdSyncLayout::LAST_ERROR)
  bool after_launch(const int* last_err_d, hipStream_t s) {
#ifdef CLSTM_HIP_EMU
    if (*last_err_d != 0 && g_dev_err) g_dev_err[0] = 0;   // handled right here (emulator: device memory is host memory)
    if (*last_err_d == 2) throw Error("persistent recurrence: group barrier timed out");
    return *last_err_d == 0;
#else
    if (verified < 4) {
      int flag = 0;
      HIPCHECK(hipMemcpyAsync(&flag, last_err_d, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHECK(hipStreamSynchronize(s));
      if (flag == 0) { verified++; return true; }
      if (g_dev_err) (void)hipMemset(g_dev_err, 0, sizeof(int));   // handled right here: nothing was written, the per-step path redoes the pass
      if (flag != 1) throw Error("persistent recurrence: a group barrier timed out in the middle of the sequence; set CLSTM_XCD_REC=0");
      return false;
    }
    return true;
#endif
  }
};
static XcdOutcome g_xcd_outcome;
// after a stream synchronisation: everything enqueued so far has run -- report what the device flagged
static void check_device_errors() {
  g_xcd_outcome.check_all();
  if (!g_dev_err) return;
  int w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHECK(hipMemcpy(w, g_dev_err, sizeof(w), hipMemcpyDeviceToHost));
  if ((w[0] | w[1] | w[2] | w[3] | w[6] | w[7]) == 0) return;
  (void)hipMemset(g_dev_err, 0, sizeof(w));
  if (w[7]) throw Error("data-parallel replicas diverged: the parameter checksums of the ranks differ at training step " + std::to_string(w[7]) + " of a net (replica check, "
                        "CLSTM_REPLICA_CHECK_EVERY); every rank applies the identical update to the identical all-reduced gradient, so this is a fault (a skipped update on "
                        "one rank, memory corruption), not drift -- no update was applied since.  The reference re-broadcasts the weights instead (distribute_weights, clstm.cc:718-729)");
  if (w[6]) throw Error("gradient exchange: " + std::to_string(w[6]) + " wait(s) of the device-side peer barrier timed out (CLSTM_PEER_TIMEOUT_S, default 120 s) although every "
                        "rank's host had announced the exchange: a peer's GPU never reached it; the minibatches enqueued since then were NOT applied -- CLSTM_PEER_ALLREDUCE=0 "
                        "puts the exchange back on RCCL");
  if (w[3]) throw Error("non-finite value (NaN or Inf) in the softmax logits or the gradient of training step " + std::to_string(w[3]) + " of a net (forward passes counted per net "
                        "from 1): no non-finite entry reaches the parameters or the momentum -- a diverged forward pass skips the whole update, a non-finite gradient entry is "
                        "skipped -- and no update enqueued since was applied.  The reference aborts here (clstm.cc:630-649).  Lower the learning rate, or CLSTM_NANCHECK=0 "
                        "to train on regardless");
  if (w[2]) throw Error("fused forward launch: " + std::to_string(w[2]) + " wait(s) for a gate-GEMM chunk / a finished frame block gave up (watchdog): outputs read back since then are "
                        "not valid and the minibatches enqueued since then were NOT applied -- set CLSTM_OVERLAP=0");
  if (w[0] == 1) g_xcd_failed = true;
  if (w[0]) throw Error("persistent recurrence: a launch failed (code " + std::to_string(w[0]) + (w[0] == 1 ? ": workgroups not all resident; per-step launches from now on" : "") +
                        "); the minibatches enqueued since then were NOT applied (layers above the failing one may have taken their update of that one minibatch) -- set CLSTM_XCD_REC=0");
  throw Error("fused backward launch: " + std::to_string(w[1]) + " weight-gradient item(s) gave up waiting for the recurrence (watchdog); the "
              "minibatches since then were NOT applied -- set CLSTM_OVERLAP=0");
}
static bool g_wide_persistent = false;   // the last launch_lstm_wide call ran the persistent per-XCD kernels
static int g_debug_fail_claims = 0;      // tests: this many upcoming persistent launches fail their placement check ...
static int g_debug_fail_skip = 0;        // ... after this many that do not

// fx_ngx > 0 (forward, bf16): try ONLY the persistent kernel with the input projection folded in (lstm_xcd_fwd_bf16_fx<fx_ngx>);
// returns false -- nothing launched or nothing written -- if it does not apply or its placement check failed: the caller then
// runs the hoisted product and calls again with fx_ngx = 0.
static std::map<const void*, int> g_stamp_base;   // per sync buffer: where the group-barrier stamps of its next persistent launch start
static bool launch_lstm_wide(bool fwd, LstmWideArgs a, int tmax, DevBuf<int>& sync, StepGraphCache& graphs, hipStream_t s, bool bf16 = false, int fx_ngx = 0, bool x3 = false) {
  g_wide_persistent = false;
  REQUIRE((double)a.N * a.ndir * 4 * a.no * 4 < 2147483000.0,
          "minibatch too large for the lock-step recurrence (frames x 4 x nhidden x ndir x 4 B must stay below 2 GiB)");
  const int no = a.no, ncu = device_cu_count();
  const int ntile = (no + 15) / 16, nzb16 = (a.bs + 15) / 16;
  // persistent bf16 kernels: 32-line groups (two 16-line MFMA tiles per workgroup) once 16-line groups would need more than
  // one launch of 8 groups -- a step's barrier and ring round trip are then paid once for twice the lines
  const int mt = !bf16 ? 1 : nzb16 * a.ndir > 16 ? 4 : nzb16 * a.ndir > 8 ? 2 : 1;
  const int nzb_default = (a.bs + 16 * mt - 1) / (16 * mt);
  a.tmax = tmax;
  sync.reserve(XcdSyncLayout::WORDS);
  a.sync = sync.p;
  // ONE launch per pass, a workgroup group per XCD with its weight rows resident in LDS (lstm_xcd_*: the default).
  // CLSTM_XCD_REC=0 selects the per-step launches below; they are also the fallback when the placement check of the
  // first launch fails (workgroups not spread evenly over the XCDs: nothing has been written at that point).
  const bool xcd_on = !(getenv("CLSTM_XCD_REC") && atoi(getenv("CLSTM_XCD_REC")) == 0);
  // (minibatches of more than 8 / ndir line blocks: one launch per chunk of line blocks)
  const int zb_per_default = std::max(1, 8 / a.ndir);
  auto persistent = [&](auto kernel, size_t smem, int nthreads = WIDE_THREADS, int nzb_ = -1, int zb_per_ = -1) {
    bool ok = true;
    const int nzb = nzb_ > 0 ? nzb_ : nzb_default, zb_per = zb_per_ > 0 ? zb_per_ : zb_per_default;
    for (int zb0 = 0; zb0 < nzb && ok; zb0 += zb_per) {
      a.zb0 = zb0; a.zbn = std::min(zb_per, nzb - zb0);
      a.debug_fail_claim = g_debug_fail_skip > 0 ? (g_debug_fail_skip--, 0) : g_debug_fail_claims > 0 ? (g_debug_fail_claims--, 1) : 0;
      // (the counter words are zero: DevBuf zero-fills, and every persistent launch returns them to zero as its last act;
      // the stamps of the group barriers keep counting up: lstm_wide.h:xcd_finish)
      {
        int& base = g_stamp_base[(const void*)sync.p];
        if (base > (1 << 30)) {   // (once per ~2 million launches)
          HIPCHECK(hipMemsetAsync(sync.p, 0, XcdSyncLayout::WORDS * sizeof(int), s));
          base = 0;
        }
        a.stamp_base = base;
        base += tmax + 2;
      }
      a.out_sticky = dev_err_words();
      a.out_host = g_xcd_outcome.prepare();
      coop_set_smem(kernel, smem);
      // An ORDINARY launch: one workgroup per CU (LDS), at most as many workgroups as CUs, the stream's previous kernel complete --
      // they are all resident, and if they ever were not, the placement check of xcd_claim times out with error 1 before anything
      // has been written.  The first four launches of a process are checked synchronously and the per-step path redoes such a pass;
      // a later failure is found asynchronously (XcdOutcome): the updates since then are skipped on the device, the host reports it
      // at its next check and uses the per-step launches from then on.  hipLaunchCooperativeKernel cost ~20 us of gaps around every
      // one of the four launches of a configs[4] step.  (The host emulator needs its "all workgroups live" launch.)
#ifdef CLSTM_HIP_EMU
      CLSTM_LAUNCH_COOP(kernel, dim3(8 * ntile), dim3(nthreads), smem, s, a);
#else
      CLSTM_LAUNCH(kernel, dim3(8 * ntile), dim3(nthreads), smem, s, a);
#endif
      check_launch();
      ok = g_xcd_outcome.after_launch(sync.p + XcdSyncLayout::LAST_ERROR, s);
      REQUIRE(ok || zb0 == 0, "persistent recurrence: placement failed after the first chunk had run; set CLSTM_XCD_REC=0");
    }
    if (ok) g_wide_persistent = true; else g_xcd_failed = true;
    return ok;
  };
  // (the persistent bf16 kernels address their per-frame arrays through 32-bit buffer offsets: G / C / D / Dbf are covered by the check
  // above, the bf16 and f32 source rows -- ndir x N x (ni + no + 8) halfs / (1 + ni + no) floats -- and the output rows must stay below 2 GiB
  // too, or the per-step launches run)
  const bool off32 = !bf16 || ((!a.Sbf || (double)a.ndir * a.N * a.sbf_ld * 2 < 2147483000.0) &&
                               (a.skip_s || (double)a.ndir * a.N * a.lds * 4 < 2147483000.0) && (double)a.N * a.ldh * 4 < 2147483000.0);
  const bool fits = xcd_on && !g_xcd_failed && off32 && tmax > 1 && ntile <= 32 && 8 * ntile <= std::max(ncu, 16);
  if (fx_ngx > 0) {
    if (!(fwd && bf16 && fits && mt == 1 && a.kp16 <= 512 && (no & 3) == 0 && a.x_ni <= 128 * fx_ngx && a.x_ni <= 2048)) return false;
    const size_t smem = (size_t)xcd_fwd_lds_bytes(1);
    return fx_ngx == 1 ? persistent(lstm_xcd_fwd_bf16_fx<1>, smem) : fx_ngx == 4 ? persistent(lstm_xcd_fwd_bf16_fx<4>, smem)
         : fx_ngx == 8 ? persistent(lstm_xcd_fwd_bf16_fx<8>, smem) : false;
  }
  if (fwd) {
    if (fits && !bf16 && (size_t)xcd_fwd_f32_lds_bytes(a.kp) <= 160 * 1024 && persistent(lstm_xcd_fwd_f32, (size_t)xcd_fwd_f32_lds_bytes(a.kp))) return true;
    if (fits && bf16 && a.kp16 <= 512 &&
        (mt == 4 ? persistent(lstm_xcd_fwd_bf16<4>, (size_t)xcd_fwd_lds_bytes(4))
         : mt == 2 ? persistent(lstm_xcd_fwd_bf16<2>, (size_t)xcd_fwd_lds_bytes(2)) : persistent(lstm_xcd_fwd_bf16<1>, (size_t)xcd_fwd_lds_bytes(1)))) return true;
    const int mts = a.bs > 32 ? 4 : a.bs > 16 ? 2 : 1;
    const dim3 grid((no + 3) / 4, a.ndir, (a.bs + 16 * mts - 1) / (16 * mts));
    const dim3 grid16(ntile * a.ndir * nzb16);
    launch_steps(graphs, bf16 ? 2 : 0, a, tmax, s, [&]() {
      LstmWideArgs w = a;
      for (int t = 0; t < tmax; t++) {
        w.step = t;
        if (bf16) CLSTM_LAUNCH(lstm_wide_fwd_step16_bf16, grid16, dim3(WIDE_THREADS), 0, s, w);
        else if (mts == 4) CLSTM_LAUNCH(lstm_wide_fwd_step<4>, grid, dim3(WIDE_THREADS), 0, s, w);
        else if (mts == 2) CLSTM_LAUNCH(lstm_wide_fwd_step<2>, grid, dim3(WIDE_THREADS), 0, s, w);
        else CLSTM_LAUNCH(lstm_wide_fwd_step<1>, grid, dim3(WIDE_THREADS), 0, s, w);
      }
    });
  } else {
    if (fits && !bf16 && x3 && a.Rw16 && a.kp16 <= 2048) {
      if (persistent(lstm_xcd_bwd_x3, (size_t)xcd_bwd_lds_bytes(1))) { g_path_count[11]++; return true; }
    } else
    if (fits && !bf16 && (size_t)xcd_bwd_f32_lds_bytes(a.kp) <= 160 * 1024 && persistent(lstm_xcd_bwd_f32, (size_t)xcd_bwd_f32_lds_bytes(a.kp))) return true;
    // 32 cells per workgroup, two groups per XCD, groups of 8 / 16 / 32 lines (lstm_wide.h:lstm_xcd_bwd_bf16_c32): half the
    // delta block per step and CU of the 16-cell kernel below, which stays for hidden sizes that are not multiples of 32
    const bool c32_on = dbg_opt("bwd_c32", 1) != 0;   // (read per pass: tests compare both kernels in one process)
    if (fits && bf16 && a.kp16 <= 2048 && c32_on && no % 32 == 0 && a.ndir <= 2) {
      const int per = 16 / a.ndir;   // line groups per launch
      const int ept = a.bs <= 8 * per ? 1 : a.bs <= 16 * per ? 2 : 4;
      const int ng = (a.bs + 8 * ept - 1) / (8 * ept);
      const size_t smem = (size_t)xcd_bwd_c32_lds_bytes(ept);
      const bool fullk = a.kp16 == 2048;
      auto go = [&](auto k_full, auto k_any) { return fullk ? persistent(k_full, smem, WIDE_THREADS, ng, per) : persistent(k_any, smem, WIDE_THREADS, ng, per); };
      if (ept == 1 ? go(lstm_xcd_bwd_bf16_c32<1, true>, lstm_xcd_bwd_bf16_c32<1, false>)
          : ept == 2 ? go(lstm_xcd_bwd_bf16_c32<2, true>, lstm_xcd_bwd_bf16_c32<2, false>) : go(lstm_xcd_bwd_bf16_c32<4, true>, lstm_xcd_bwd_bf16_c32<4, false>)) {
        g_path_count[9]++;
        return true;
      }
    } else
    if (fits && bf16 && a.kp16 <= 2048 &&
        (mt == 4 ? persistent(lstm_xcd_bwd_bf16<4>, (size_t)xcd_bwd_lds_bytes(4))
         : mt == 2 ? persistent(lstm_xcd_bwd_bf16<2>, (size_t)xcd_bwd_lds_bytes(2)) : persistent(lstm_xcd_bwd_bf16<1>, (size_t)xcd_bwd_lds_bytes(1)))) return true;
    const dim3 grid(ntile, a.ndir, nzb16);
    const dim3 grid16(ntile * a.ndir * nzb16);
    launch_steps(graphs, bf16 ? 3 : 1, a, tmax, s, [&]() {
      LstmWideArgs w = a;
      for (int t = 0; t < tmax; t++) {
        w.step = t;
        if (bf16) CLSTM_LAUNCH(lstm_wide_bwd_step16_bf16, grid16, dim3(WIDE_THREADS), 0, s, w);
        else CLSTM_LAUNCH(lstm_wide_bwd_step, grid, dim3(WIDE_THREADS), 0, s, w);
      }
    });
  }
  check_launch();
  return true;
}

// ---- per-kernel device timing (bench.py roofline) ---------------------------------------------
#ifndef CLSTM_HIP_EMU
struct Timing {
  bool on = false;
  struct Rec { std::string name; std::vector<ClstmLaunchEvents> launches; };
  std::vector<Rec> pending;
  std::map<std::string, std::pair<double, int>> acc;
  // a bracket names the launches issued inside it; each launch brings its own pair of events (devintrin.h, CLSTM_LAUNCH)
  void begin(const char* name, hipStream_t) {
    if (!on) return;
    pending.emplace_back();
    pending.back().name = name;
    clstm_launch_sink = &pending.back().launches;
  }
  void end(hipStream_t) {
    if (!on) return;
    clstm_launch_sink = nullptr;
  }
  void collect(hipStream_t s) {
    clstm_launch_sink = nullptr;
    if (pending.empty()) return;
    HIPCHECK(hipStreamSynchronize(s));
    for (auto& r : pending) {
      double sum = 0;
      for (auto& e : r.launches) {
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, e.a, e.b));
        sum += ms;
        clstm_event_pool.push_back(e);
      }
      if (r.launches.empty()) continue;      // (a bracket around graph replays or copies only: nothing to report)
      auto& a = acc[r.name];
      a.first += sum; a.second += 1;
    }
    pending.clear();
  }
};
#else
struct Timing {
  bool on = false;
  std::map<std::string, std::pair<double, int>> acc;
  void begin(const char*, hipStream_t) {}
  void end(hipStream_t) {}
  void collect(hipStream_t) {}
};
#endif


// ---- roctx ranges (a rocprofv3 --marker-trace of a drop-in run should read ingest / forward / ctc / backward /
// allreduce / update) ----------------------------------------------------------------------------------------------------
// librocprofiler-sdk-roctx is bound at first use like RCCL; ranges are emitted when a profiler tool is attached to the
// process (rocprofv3 exports ROCP_TOOL_LIBRARIES) or CLSTM_ROCTX=1 asks for them; CLSTM_ROCTX=0 switches them off.
#ifndef CLSTM_HIP_EMU
}  // namespace clstm
#include <dlfcn.h>
namespace clstm {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  static Roctx& get() {
    static Roctx r = [] {
      Roctx x;
      const char* e = getenv("CLSTM_ROCTX");
      const bool want = e ? atoi(e) != 0 : getenv("ROCP_TOOL_LIBRARIES") != nullptr;
      if (!want) return x;
      void* h = nullptr;
      for (const char* name : {"librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
      if (!h) return x;
      x.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      x.pop = (int (*)())dlsym(h, "roctxRangePop");
      if (!x.push || !x.pop) x.push = nullptr;
      return x;
    }();
    return r;
  }
};
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) : on(Roctx::get().push != nullptr) { if (on) Roctx::get().push(name); }
  ~RoctxRange() { if (on) Roctx::get().pop(); }
};
#else
struct RoctxRange { explicit RoctxRange(const char*) {} };
#endif

#include "comm.h"   // the gradient exchange: communicator, peer-read all-reduce, replica check buffers

struct Layer {
  int ni, no, nk4, nthreads;
  bool wide = false;          // lock-step recurrence (lstm_wide.h) instead of the register-resident one
  int kpf = 0, kpb = 0;
  long long nwf = 0, nwb = 0;
  PackDesc pd;
  float *Wt = nullptr, *bias = nullptr, *Rf = nullptr, *Rb = nullptr, *Rwf = nullptr, *Rwb = nullptr;
  float* Wk = nullptr;        // k-contiguous W_x rows for the producer items of the fused forward launch (ops.h:PackFused)
  int wk_kp = 0, wk_njp = 0;
  DevBuf<float> dCc;
  // minibatches that fill the chip: the recurrence batched over 16 lines on the f16 MFMA (lstm_mfma.h); fragments repacked when
  // the parameters have moved (Net::params_epoch)
  DevBuf<unsigned short> Wmf;
  DevBuf<float> mf_scale;
  long long mf_epoch = -1;
  DevBuf<unsigned short> Wmfb;   // R^T fragments of the batched backward recurrence (lstm_mfma_bwd.h)
  long long mfb_epoch = -1;
  DevBuf<int> pack_tab;       // source index of every packed element (k_pack_index), narrow layers in training steps
  DevBuf<int> pack_inv;       // ... and its inverse, PACK_KD packed elements per parameter (ops.h: PackDst); pack_inv_state: 0 not built, 1 ready, -1 unusable
  int pack_inv_state = 0;
  // bf16 recurrence of a wide layer (lstm_wide_bf16.h): packed weights and the bf16 copies of h / the deltas
  unsigned short *Rbf = nullptr, *Rbb = nullptr;
  DevBuf<unsigned short> R2b, D2;   // f32-grade backward recurrence on the bf16 MFMA (lstm_xcd_bwd_x3): hi | lo planes of the weights and of the delta ring
  bool r2b_ready = false;           // ... R2b holds the CURRENT weights (set / cleared by every repack)
  DevBuf<unsigned short> Hb, Db;
  DevBuf<float> Rf32;          // tiled lock-step ring of the persistent f32 recurrences (one pass at a time uses it)
  long long* moff = nullptr;  // [ndir*4no] flat offset of packed row m (column 0)
  DevBuf<float> G, C, H, D, dH, S;
  DevBuf<unsigned short> Hbf;  // per-frame bf16 h of both directions written by the persistent forward kernel (A operand of the next layer's W_x product)
  DevBuf<unsigned short> WtbT; // bf16 W_x, k-contiguous ([M][ni]): B operand of that product
  bool fwd_persistent = false; // this forward pass ran the persistent kernel (Hbf is valid)
  DevBuf<unsigned short> Sbf;  // bf16 source rows [x | h_{t-1} | 1] per direction (x: k_source_x_bf16, h: the persistent forward kernel)
  bool sbf_ready = false;      // ... complete for this forward pass
  bool sbf_x_external = false; // ... except their x columns: the weight-gradient GEMM reads those from the layer below's Hbf (gemm_b16mc A2)
  long long sbf_one_key = -1;  // batch geometry (N) the constant bias column of Sbf was written for
  bool d_f32_valid = true;     // D (f32 gate deltas) is current (a persistent bf16 backward pass may leave only Dbf)
  bool sx_valid = true;        // the [1 | x] columns of S (f32) are current (built lazily when the bf16 rows serve the weight gradient)
  bool h_f32_valid = true;     // the f32 outputs H are current (a persistent bf16 forward pass of a lower layer leaves only Hbf)
  bool sh_valid = true;        // the h_{t-1} columns of S (f32) are current (... only Sbf)
  DevBuf<unsigned short> Dbf;  // per-frame bf16 gate deltas written by the persistent backward kernel (A operand of the x.d GEMM)
  DevBuf<float> dbias;         // [bs][ndir][no][4] per-line sums of those deltas (the bias row of the weight gradient), same kernel
  DevBuf<float> partial;       // stacked nets: this layer's own split-K slabs of the weight gradient (reduced behind the LAST recurrence of the pass)
  DevBuf<unsigned short> Wtb;  // bf16 copy of Wt ([ni][M], k = gate column contiguous): B operand of the bf16-source x.d product
  int lds = 0;
  int wt_slack = 32;          // floats past Wt a vectorised staging load may touch
  int ldh = 0, hofs = 4;      // H rows: [pad pad pad 1 | h_dir0 | h_dir1], h at column hofs (16-byte aligned)
  float* hrow() const { return H.p + hofs; }            // h block of frame 0
  float* srow() const { return H.p + hofs - 1; }        // [1 | h] = the next layer's / softmax's source row
};

struct Net {
  clstm_net_desc desc;
  int ndir;
  std::vector<Layer> L;
  int sm_ni = 0;
  long long sm_off = 0;
  int nparams = 0;
  float *v = nullptr, *d = nullptr, *g = nullptr;
  bool own_v = false, own_d = false, own_g = false;
  float lr = 1e-4f, mom = 0.9f, gclip = 100.0f;
  bool packed_dirty = true;
  long long params_epoch = 0;   // bumped whenever v may have changed (set_params, params_changed, every update)
  bool bf16_gemm = false;   // hoisted gate GEMMs: bf16 in, f32 accumulate   (clstm_net_set_gemm_precision)
  bool bf16_rec = false;    // wide layers: bf16 MFMA operands in the recurrence
  bool want_dx0 = false;
  // batch
  int bs = 0, tmax = 0;
  bool src0_ready = false;    // layer 0's source rows [1 | x] already written by set_inputs_d
  long long N = 0;
  std::vector<int> line_off_h;
  DevBuf<int> line_off;            // [bs + 1] first frame of each line | [bs] dispatch order (set_batch)
  std::vector<int> order_h;
  PinnedRing ring;
  const int* lo_stage = nullptr;   // pinned copy of the line offsets not yet on the device
  bool lo_pending = false;
  void flush_line_off() {
    if (!lo_pending) return;
    hipStream_t s = stream();
    HIPCHECK(hipMemcpyAsync(line_off.p, lo_stage, (2 * bs + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    ring.commit(s);
    lo_pending = false;
  }
  DevBuf<float> X, Z, Dz, dX0, partial, partial_sm, aligned, tmp;
  DevBuf<unsigned short> xbf;   // bf16 copy of the input frames (first wide layer's W_x product in precision mode 2)
  ReduceDesc sm_red{};
  DevBuf<long long> lstm_prof;  // diagnostics build only
  StepGraphCache step_graphs;   // captured per-step launch sequences of the lock-step recurrence
  DevBuf<int> coop_sync;      // grid-barrier ticket counter + watchdog flag of the cooperative recurrence
  // ctc / decode
  DevBuf<int> states, state_off, dec_idx, dec_cls, dec_loc, dec_cnt;
  DevBuf<float> dec_val, lat;
  DevBuf<long long> lat_off;
  Timing timing;
  // --- weight-gradient GEMM beside the backward recurrence (gemm_dw.h) ---
  static const int PROG_LINES = 2048;                          // overlap only for minibatches up to this many lines
  static const int PROG_WORDS = 2 * PROG_LINES * PROG_STRIDE;   // progress words at the tail of a narrow layer's D allocation: [ndir][bs], one per 128 B
  // CLSTM_OVERLAP / clstm_net_set_overlap: 0 off (GEMM after the recurrence); 1 (default): recurrence and GEMM as two
  // roles of ONE launch (lstm_bwd_dw.h) for batches large enough -- 0.369 -> 0.356 ms per step at the bench shape;
  // 2: the same always (tests force it onto tiny nets).  (Two streams with complementary CU masks were measured
  // slower than mode 0 -- 0.443 vs 0.377 ms per step, profiles/r02_timeline_overlap.txt -- and are gone.)
  int overlap = getenv("CLSTM_OVERLAP") ? atoi(getenv("CLSTM_OVERLAP")) : 1;
  unsigned dw_done_total = 0;     // recurrence workgroups launched so far through the fused launch (GemmDwArgs::done)
  DevBuf<long long> dw_trace;
  DevBuf<int> dw_ktab, dw_slabs, dw_queue;   // dw_queue: the monitor's published minima and the `done` counter, each on its own 128-byte line
  std::vector<int> dw_key;        // line offsets the tables were built for
  // the softmax layer's W.d as independent items of the top layer's fused backward launch (gemm_dw.h, GemmDwArgs::x*)
  DevBuf<int> dwx_tab;            // [entries][2] contiguous 16-frame entries | slabs
  long long dwx_N = -1;
  int dwx_nslabs = 0, dwx_entries = 0;
  bool dwx_active = false;        // this backward pass: the top layer's launch carries them
  int dw_nslabs = 0, dw_slabs_per_dir = 0, dw_ntiles_max = 0;
  int prog_base = 1024;           // grows with every backward launch: stale progress words never look complete
  long long dw_launches = 0;      // overlapped backward passes so far (tests check the path was taken)
  Comm* comm = nullptr;         // data-parallel ranks: all-reduce of g before the update (not owned)
  // --- host-fed training steps (clstm_net_train_step_h): frames travel on a copy stream into one of two device input
  // buffers while the previous step computes; the compute stream waits for the copy's event; a slot is rewritten only
  // after the step that read it has ended -- that step's update kernel says so in a pinned host word (no event record on
  // the compute stream: one costs ~5 us of stream time)
  struct HostFeed {
    hipStream_t cs = nullptr;
    hipEvent_t copied[2] = {};
    DevBuf<float> xin[2];
    void* pin[2] = {nullptr, nullptr};     // staging for pageable sources
    size_t pin_cap[2] = {0, 0};
    int* step_done = nullptr;              // pinned: number of the last step whose update kernel has run
    long long steps = 0;
    bool ready = false;
  } hf;
  // --- fused forward launch (lstm_fwd_fused.h): W_x GEMM producers + recurrence + softmax consumers ---
  float* W1k = nullptr;         // k-contiguous softmax rows (PackFused)
  int w1k_kps = 0;
  DevBuf<int> fw_items, fw_flags;
  std::vector<int> fw_key;      // line offsets the item lists were built for
  int fw_npitems = 0, fw_ncitems = 0, fw_chunks = 0;
  int fw_epoch = 0, fw_prog_base = 1024;
  long long fw_launches = 0;

  hipStream_t stream() const { return g_stream; }

  static bool l_is_only_layer(const clstm_net_desc& ds) { return ds.nlayers == 1; }
  // source-index table of a narrow layer's per-step repack (ops.h:k_pack_index), built at first use
  int* pack_table(Layer& y) {
    if (!y.pack_tab.p) {
      const PackFused pf = pack_fused_desc(y);
      const size_t n = pack_count(y);
      y.pack_tab.reserve(n);
      CLSTM_LAUNCH(k_pack_index, dim3(nblocks(n)), dim3(256), 0, stream(), y.pack_tab.p, y.pd, pf);
      check_launch();
    }
    return y.pack_tab.p;
  }
  // the inverse of pack_table(): built once per net on the host (one synchronous read-back of the table)
  static constexpr int PACK_KD = 4;
  size_t pack_count(const Layer& y) const {
    const PackFused pf = pack_fused_desc(y);
    return (size_t)(1 + y.ni) * ndir * 4 * y.no + 2 * ((size_t)ndir * 4 * 4 * y.nk4 * y.nthreads) +
           (pf.Wk ? (size_t)ndir * pf.njp * 16 * pf.kp + (size_t)96 * pf.kps : 0);
  }
  bool pack_inverse(Layer& y) {
    if (y.pack_inv_state) return y.pack_inv_state > 0;
    const int* tab = pack_table(y);
    const size_t n = pack_count(y);
    std::vector<int> t(n);
    HIPCHECK(hipMemcpyAsync(t.data(), tab, n * sizeof(int), hipMemcpyDeviceToHost, stream()));
    HIPCHECK(hipStreamSynchronize(stream()));
    std::vector<int> inv((size_t)nparams * PACK_KD, -1), cnt(nparams, 0);
    y.pack_inv_state = 1;
    for (size_t e = 0; e < n && y.pack_inv_state > 0; e++) {
      const int si = t[e];
      if (si < 0) continue;
      if (si >= nparams || cnt[si] >= PACK_KD) { y.pack_inv_state = -1; break; }
      inv[(size_t)si * PACK_KD + cnt[si]++] = (int)e;
    }
    if (y.pack_inv_state < 0) return false;
    y.pack_inv.reserve(inv.size());
    HIPCHECK(hipMemcpyAsync(y.pack_inv.p, inv.data(), inv.size() * sizeof(int), hipMemcpyHostToDevice, stream()));
    HIPCHECK(hipStreamSynchronize(stream()));
    return true;
  }
  PackFused pack_fused_desc(const Layer& y) const {
    PackFused f{};
    if (y.Wk) { f.Wk = y.Wk; f.kp = y.wk_kp; f.njp = y.wk_njp; f.W1k = W1k; f.kps = w1k_kps; f.nc = desc.nclasses; f.sm_k = sm_ni; f.sm_off = sm_off; }
    return f;
  }
  void build(const clstm_net_desc& ds, float* pv, float* pd, float* pg) {
    desc = ds;
    REQUIRE(ds.nlayers >= 1 && ds.nlayers <= CLSTM_MAX_LAYERS, "nlayers out of range");
    REQUIRE(ds.ninput > 0 && ds.nclasses >= 2, "bad ninput/nclasses (Softmax requires no>=2, clstm.cc:399)");
    ndir = ds.unidirectional ? 1 : 2;
    static const int blockidx[4] = {2, 1, 3, 0};  // slot gi,gf,go,ci -> alphabetical WCI,WGF,WGI,WGO
    long long off = 0;
    int ni = ds.ninput;
    L.resize(ds.nlayers);
    for (int l = 0; l < ds.nlayers; l++) {
      Layer& y = L[l];
      y.ni = ni;
      y.no = ds.nhidden[l];
      REQUIRE(y.no > 0, "nhidden must be positive");
      y.nk4 = pick_nk4(y.no);
      const bool force_wide = getenv("CLSTM_FORCE_WIDE") && atoi(getenv("CLSTM_FORCE_WIDE")) != 0;
      y.wide = y.nk4 < 0 || force_wide;
      if (y.wide) y.nk4 = 1;
      y.nthreads = y.wide ? 64 : 64 * ((y.no + 15) / 16);
      y.lds = ((1 + y.ni + y.no + 15) / 16) * 16;   // source rows [1 | x | h_prev], padded to 64-byte rows
      y.ldh = ((y.hofs + ndir * y.no + 15) / 16) * 16;   // 64-byte aligned rows
      const long long blk = (long long)y.no * (1 + y.ni + y.no);
      y.pd.ni = y.ni; y.pd.no = y.no; y.pd.ndir = ndir; y.pd.nk4 = y.nk4; y.pd.nthreads = y.wide ? 0 : y.nthreads; y.pd.ku = pick_ku(y.no, y.nk4);
      for (int dir = 0; dir < ndir; dir++) {
        for (int s = 0; s < 4; s++) y.pd.p_off[dir][s] = off + blockidx[s] * blk;
        off += 4 * blk;
      }
      if (ndir == 1) for (int s = 0; s < 4; s++) y.pd.p_off[1][s] = y.pd.p_off[0][s];
      ni = ndir * y.no;
    }
    sm_ni = ni;
    sm_off = off;
    off += (long long)ds.nclasses * (1 + sm_ni);
    nparams = (int)off;
    auto own = [&](float*& dst, float* given, bool& flag) {
      if (given) { dst = given; flag = false; }
      else {
        HIPCHECK(hipMalloc((void**)&dst, (size_t)nparams * sizeof(float)));
        zero_fill(dst, (size_t)nparams * sizeof(float));
        flag = true;
      }
    };
    own(v, pv, own_v); own(d, pd, own_d); own(g, pg, own_g);
    for (auto& y : L) {
      const int M = ndir * 4 * y.no, KQP = 4 * y.nk4;
      HIPCHECK(hipMalloc((void**)&y.Wt, ((size_t)y.ni * M + y.wt_slack) * sizeof(float)));
      zero_fill(y.Wt, ((size_t)y.ni * M + y.wt_slack) * sizeof(float));
      HIPCHECK(hipMalloc((void**)&y.bias, (size_t)M * sizeof(float)));
      if (y.wide) {
        y.kpf = wide_kp_fwd(y.no); y.kpb = wide_kp_bwd(y.no);
        y.nwf = (long long)ndir * ((y.no + 3) / 4) * 16 * y.kpf;
        y.nwb = (long long)ndir * ((y.no + 15) / 16) * 16 * y.kpb;
        HIPCHECK(hipMalloc((void**)&y.Rwf, (size_t)(y.nwf + 4) * sizeof(float)));
        HIPCHECK(hipMalloc((void**)&y.Rwb, (size_t)(y.nwb + 4) * sizeof(float)));
        HIPCHECK(hipMalloc((void**)&y.Rbf, ((size_t)ndir * ((y.no + 3) / 4) * 16 * wide_kp16_fwd(y.no) + 8) * sizeof(unsigned short)));
        HIPCHECK(hipMalloc((void**)&y.Rbb, ((size_t)ndir * ((y.no + 15) / 16) * 16 * wide_kp16_bwd(y.no) + 8) * sizeof(unsigned short)));
      } else {
        HIPCHECK(hipMalloc((void**)&y.Rf, (size_t)ndir * 4 * KQP * y.nthreads * sizeof(float)));
        HIPCHECK(hipMalloc((void**)&y.Rb, (size_t)ndir * 4 * KQP * y.nthreads * sizeof(float)));
      }
      if (!y.wide && l_is_only_layer(ds) && ds.nclasses <= SMX_COLS) {
        y.wk_kp = (y.ni + 15) / 16 * 16; y.wk_njp = (4 * y.no + 15) / 16;
        HIPCHECK(hipMalloc((void**)&y.Wk, ((size_t)ndir * y.wk_njp * 16 * y.wk_kp + 64) * sizeof(float)));
        w1k_kps = (ndir * y.no + 15) / 16 * 16;
        HIPCHECK(hipMalloc((void**)&W1k, ((size_t)96 * w1k_kps + 64) * sizeof(float)));
      }
      HIPCHECK(hipMalloc((void**)&y.moff, (size_t)M * sizeof(long long)));
      std::vector<long long> mo(M);
      for (int m = 0; m < M; m++) {
        const int dir = m / (4 * y.no), c = (m % (4 * y.no)) >> 2, s = m & 3;
        mo[m] = y.pd.p_off[dir][s] + c;
      }
      HIPCHECK(hipMemcpy(y.moff, mo.data(), (size_t)M * sizeof(long long), hipMemcpyHostToDevice));
    }
    packed_dirty = true;
  }
  ~Net() {
    for (auto& y : L) {
      (void)hipFree(y.Wt); (void)hipFree(y.bias); (void)hipFree(y.Rf); (void)hipFree(y.Rb); (void)hipFree(y.moff); (void)hipFree(y.Wk);
      (void)hipFree(y.Rwf); (void)hipFree(y.Rwb); (void)hipFree(y.Rbf); (void)hipFree(y.Rbb); y.dCc.release(); y.Hb.release(); y.Db.release(); y.Rf32.release(); y.R2b.release(); y.D2.release();
      y.G.release(); y.C.release(); y.H.release(); y.D.release(); y.dH.release(); y.S.release(); y.Sbf.release(); y.sbf_ready = false; y.pack_tab.release(); y.pack_inv.release(); y.Wmf.release(); y.mf_scale.release(); y.Wmfb.release(); y.pack_inv_state = 0; y.partial.release(); y.dbias.release();
    }
    (void)hipFree(W1k); fw_items.release(); fw_flags.release();
    for (int i = 0; i < 2; i++) { hf.xin[i].release(); if (hf.pin[i]) (void)hipHostFree(hf.pin[i]); if (hf.copied[i]) (void)hipEventDestroy(hf.copied[i]); }
    if (hf.step_done) (void)hipHostFree(hf.step_done);
    if (hf.cs) (void)hipStreamDestroy(hf.cs);
    if (own_v) (void)hipFree(v);
    if (own_d) (void)hipFree(d);
    if (own_g) (void)hipFree(g);
    line_off.release(); X.release(); Z.release(); Dz.release();
    dX0.release(); partial.release(); partial_sm.release(); coop_sync.release(); lstm_prof.release(); aligned.release(); tmp.release(); states.release();
    state_off.release(); dec_idx.release(); dec_cls.release(); dec_loc.release(); dec_cnt.release();
    dec_val.release(); lat.release(); lat_off.release();
  }

  void repack() {
    if (!packed_dirty) return;
    hipStream_t s = stream();
    for (auto& y : L) {
      const int M = ndir * 4 * y.no, KQP = 4 * y.nk4;
      const size_t nr = y.wide ? 0 : (size_t)ndir * 4 * KQP * y.nthreads;
      if (y.wide && bf16_rec && y.no % 128 == 0 && y.ni % 32 == 0 && wide_kp16_fwd(y.no) == y.no && wide_kp16_bwd(y.no) == 4 * y.no &&
          dbg_opt("pack_tiles", 1) != 0) {
        // every bf16-mode copy of the layer in one tiled pass (ops.h:k_pack_wide_tiles)
        const int rf = (y.no + 3) / 4 * 16, kf = wide_kp16_fwd(y.no), rb = (y.no + 15) / 16 * 16, kb = wide_kp16_bwd(y.no);
        y.Wtb.reserve((size_t)y.ni * M + 64);
        y.WtbT.reserve((size_t)y.ni * M + 64);
        const unsigned nblk = (unsigned)ndir * (unsigned)((y.ni + y.no) / 32) * (unsigned)(y.no / 32);
        CLSTM_LAUNCH(k_pack_wide_tiles, dim3(nblk), dim3(256), 0, s, (const float*)v, y.pd, y.Wt, y.bias, y.Wtb.p, y.WtbT.p, y.Rbf, y.Rbb, kf, kb, rf, rb);
        continue;
      }
      if (y.wide && bf16_rec) {
        const int rf = (y.no + 3) / 4 * 16, kf = wide_kp16_fwd(y.no), rb = (y.no + 15) / 16 * 16, kb = wide_kp16_bwd(y.no);
        CLSTM_LAUNCH(k_pack_wide_bf16, dim3(nblocks((size_t)ndir * ((size_t)rf * kf + (size_t)rb * kb))), dim3(256), 0, s,
                     (const float*)v, y.Rbf, y.Rbb, y.pd, rf, kf, rb, kb);
      }
      else if (y.wide) {
        CLSTM_LAUNCH(k_pack_wide, dim3(nblocks((size_t)(y.nwf + y.nwb))), dim3(256), 0, s, (const float*)v, y.Rwf, y.Rwb,
                     y.pd, y.kpf, y.kpb);
        y.r2b_ready = false;
        if (rec_x3()) {   // hi | lo planes of the backward recurrence's weights
          y.r2b_ready = true;
          const int rb = (y.no + 15) / 16 * 16, kb = wide_kp16_bwd(y.no);
          const long long pb = (long long)ndir * rb * kb;
          y.R2b.reserve((size_t)2 * pb + 64);
          CLSTM_LAUNCH(k_pack_wide_split, dim3(nblocks((size_t)pb)), dim3(256), 0, s, (const float*)v, y.R2b.p, y.pd, rb, kb, pb);
        }
      }
      CLSTM_LAUNCH(k_pack_layer, dim3(nblocks((size_t)(1 + y.ni) * M + 2 * nr)), dim3(256), 0, s, (const float*)v, y.Wt,
                   y.bias, y.Rf, y.Rb, y.pd, pack_fused_desc(y));
      if (y.wide && bf16_rec) {   // bf16 copy of W_x^T for the bf16-source x.d product
        y.Wtb.reserve((size_t)y.ni * M + 64);
        CLSTM_LAUNCH(k_to_bf16, dim3(nblocks((size_t)y.ni * M)), dim3(256), 0, s, (const float*)y.Wt, y.Wtb.p, (size_t)y.ni * M);
        y.WtbT.reserve((size_t)y.ni * M + 64);
        CLSTM_LAUNCH(k_transpose_to_bf16, dim3(nblocks((size_t)y.ni * M)), dim3(256), 0, s, (const float*)y.Wt, y.WtbT.p, y.ni, M);
      }
    }
    check_launch();
    packed_dirty = false;
  }

  void set_batch(const int* T_h, int nb) {
    REQUIRE(nb > 0, "empty batch");
    bs = nb;
    line_off_h.assign(nb + 1, 0);
    for (int b = 0; b < nb; b++) {
      REQUIRE(T_h[b] >= 0, "negative line length");
      line_off_h[b + 1] = line_off_h[b] + T_h[b];
    }
    N = line_off_h[nb];
    REQUIRE(N > 0, "batch has no frames");
    tmax = 0;
    for (int b = 0; b < nb; b++) tmax = std::max(tmax, T_h[b]);
    src0_ready = false;
    // behind the offsets: the order in which the per-line workgroups take the lines -- longest first, so that when
    // there are more lines than CUs the long ones are not the last to start (stable: equal lengths keep their order)
    order_h.resize(nb);
    for (int b = 0; b < nb; b++) order_h[b] = b;
    std::stable_sort(order_h.begin(), order_h.end(), [&](int x, int y) { return T_h[x] > T_h[y]; });
    line_off.reserve(2 * nb + 1);
    hipStream_t s = stream();
    int* stage = (int*)ring.acquire((2 * nb + 1) * sizeof(int));
    memcpy(stage, line_off_h.data(), (nb + 1) * sizeof(int));
    memcpy(stage + nb + 1, order_h.data(), nb * sizeof(int));
    // The device copy is made by the next input-ingest launch (which reads the pinned slot directly: one DMA
    // launch of ~4 us less per step) or, if something else needs it first, by flush_line_off().
    lo_stage = stage; lo_pending = true;
    X.reserve((size_t)N * desc.ninput);
    for (auto& y : L) {
      y.G.reserve((size_t)N * ndir * 4 * y.no);
      y.C.reserve((size_t)N * ndir * y.no);
      {
        const size_t cap0 = y.H.cap;
        y.H.reserve((size_t)N * y.ldh + 64 + (y.Wk ? PROG_WORDS : 0));   // (fused forward: progress words behind the rows)
        if (y.H.cap != cap0) {
          const size_t rows = (y.H.cap - (y.Wk ? PROG_WORDS + 64 : 0)) / y.ldh;
          CLSTM_LAUNCH(k_fill_col0, dim3(nblocks(rows)), dim3(256), 0, s, y.H.p, rows, y.ldh, y.hofs - 1);
        }
      }
      y.D.reserve((size_t)N * ndir * 4 * y.no + (y.wide ? 0 : PROG_WORDS + 64));
      y.dH.reserve((size_t)N * ndir * y.no);
      if (y.wide && bf16_rec) {
        // lock-step rings [step parity][dir][line][k] (lstm_wide.h); never smaller than the per-frame layout of the first version
        y.Hb.reserve((size_t)std::max<long long>(N, 4LL * bs + 32) * ndir * wide_kp16_fwd(y.no) + 64);   // (2 x whole 16-line blocks: the tiled ring of the persistent kernels)
        y.Db.reserve((size_t)std::max<long long>(N, 2LL * bs + 32) * ndir * wide_kp16_bwd(y.no) + 64);
      }
      y.S.reserve((size_t)N * ndir * y.lds + 64);
    }
    Z.reserve((size_t)N * desc.nclasses);
    Dz.reserve((size_t)N * desc.nclasses);
  }

  const float* layer_input(int l) const { return l == 0 ? X.p : L[l - 1].hrow(); }
  int layer_input_ld(int l) const { return l == 0 ? desc.ninput : L[l - 1].ldh; }
  int layer_input_slack(int l) const { return 32; }   // X and H are over-allocated by >= 64 floats

  LstmWideArgs wide_args(Layer& y, bool fwd) {
    LstmWideArgs w{};
    w.Rw = fwd ? y.Rwf : y.Rwb; w.rw_elems = fwd ? y.nwf : y.nwb;
    w.G = y.G.p; w.C = y.C.p; w.H = y.H.p; w.dH = y.dH.p; w.D = y.D.p;
    y.dCc.reserve((size_t)bs * ndir * y.no);
    w.dC = y.dCc.p; w.line_off = line_off.p; w.S = y.S.p; w.sdir = (long long)N * y.lds; w.N = N;
    w.lds = y.lds; w.sofs = 1 + y.ni; w.ldh = y.ldh; w.hofs = y.hofs; w.no = y.no; w.ndir = ndir; w.bs = bs;
    w.kp = fwd ? y.kpf : y.kpb;
    if (!bf16_rec) { y.Rf32.reserve(ring32_floats(ndir, bs, std::max(y.kpf, y.kpb)) + 64); w.Rf = y.Rf32.p; }
    if (!bf16_rec && !fwd && rec_x3() && y.r2b_ready) {   // (tried first by launch_lstm_wide; the f32 kernel stays as the fallback)
      const int nblk = (bs + 15) / 16;
      w.kp16 = wide_kp16_bwd(y.no);
      w.ring_plane = 2LL * ndir * nblk * 16 * w.kp16;
      y.D2.reserve((size_t)2 * w.ring_plane + 64);
      w.Db = y.D2.p; w.Rw16 = y.R2b.p; w.rw_plane = (long long)ndir * ((y.no + 15) / 16 * 16) * w.kp16;
    }
    if (bf16_rec) {
      w.Rw16 = fwd ? y.Rbf : y.Rbb;
      w.kp16 = fwd ? wide_kp16_fwd(y.no) : wide_kp16_bwd(y.no);
      w.rw_elems = (long long)ndir * (fwd ? (y.no + 3) / 4 : (y.no + 15) / 16) * 16 * w.kp16;
      w.Hb = y.Hb.p; w.Db = y.Db.p;
      if (!fwd && wide_kp16_bwd(y.no) == 4 * y.no) {
        y.Dbf.reserve((size_t)N * ndir * w.kp16 + 64); w.Dbf = y.Dbf.p;
        // with bf16 GEMMs behind it, nobody reads the f32 deltas of a persistent pass (16 bytes per lane and step, 0.17 ms
        // per configs[4] minibatch); ensure_delta_f32() expands Dbf for a fallback product
        w.skip_d = bf16_gemm;
        y.dbias.reserve((size_t)bs * ndir * 4 * y.no + 64); w.dbias = y.dbias.p;   // per-line bias-gradient sums (lstm_wide.h: LstmWideArgs::dbias)
      }
      const bool b16mc_on = dbg_opt("gemm_b16mc", 1) != 0;
      if (fwd && b16mc_on && bf16_gemm && (y.ni & 7) == 0 && (y.no & 7) == 0 && wide_kp16_bwd(y.no) == 4 * y.no) {
        const int ldsb = y.ni + y.no + 8;
        { const size_t cap0 = y.Sbf.cap; y.Sbf.reserve((size_t)N * ndir * ldsb + 64); if (y.Sbf.cap != cap0) y.sbf_one_key = -1; }
        w.Sbf = y.Sbf.p; w.sbf_ld = ldsb; w.sbf_ofs = y.ni; w.sbf_dir = (long long)N * ldsb;
      }
      if (fwd && (y.no & 1) == 0 && &y != &L.back()) { y.Hbf.reserve((size_t)N * ndir * y.no + 64); w.Hbf = y.Hbf.p; w.hbf_ld = ndir * y.no; }
      // Stores nobody reads in this mode (the per-frame stores of the persistent kernel cost per INSTRUCTION, seven per wave
      // and step): the f32 outputs of a layer whose consumer takes Hbf (the next layer's W_x product and source rows), the
      // f32 h_{t-1} source columns when the weight gradient takes Sbf.  ensure_h_f32 / ensure_source rebuild them exactly.
      if (fwd) {   // (measured at configs[4]: forward passes 1.647 -> 1.580 ms per minibatch)
        const int l = (int)(&y - L.data());
        const bool next_takes_hbf = w.Hbf && l + 1 < (int)L.size() && bf16_gemm && L[l + 1].WtbT.p &&
                                    L[l + 1].ni == ndir * y.no && (L[l + 1].ni & 1) == 0;
        w.skip_h = next_takes_hbf;
        w.skip_s = w.Sbf != nullptr;
      }
    }
#ifdef CLSTM_LSTM_PROF
    lstm_prof.reserve(128); w.prof = lstm_prof.p;    // (the last pass launched before clstm_debug_lstm_cycles is what it reads)
#endif
    return w;
  }

  void ensure_delta_f32(int l) {
    Layer& y = L[l];
    if (y.d_f32_valid) return;
    CLSTM_LAUNCH(k_bf16_to_f32, dim3(nblocks((size_t)N * ndir * 4 * y.no)), dim3(256), 0, stream(), y.Dbf.p, y.D.p, (size_t)N * ndir * 4 * y.no);
    y.d_f32_valid = true;
  }
  void ensure_h_f32(int l) {   // exact: the kernel's own h = tanh(c) * go (ops.h:k_h_from_state)
    Layer& y = L[l];
    if (y.h_f32_valid) return;
    CLSTM_LAUNCH(k_h_from_state, dim3(nblocks((size_t)N * ndir * y.no)), dim3(256), 0, stream(), y.H.p, (const float*)y.G.p, (const float*)y.C.p,
                 (size_t)N, y.no, ndir, y.ldh, y.hofs);
    y.h_f32_valid = true;
  }
  void ensure_source_h(int l) {
    Layer& y = L[l];
    if (y.sh_valid) return;
    ensure_h_f32(l);
    flush_line_off();
    CLSTM_LAUNCH(k_source_h, dim3(nblocks((size_t)N * ndir * y.no)), dim3(256), 0, stream(), y.S.p, (const float*)y.H.p, (const int*)line_off.p, bs,
                 (size_t)N, y.no, ndir, y.ldh, y.hofs, y.lds, 1 + y.ni, (long long)N * y.lds);
    y.sh_valid = true;
  }
  void ensure_source(int l) { ensure_source_x(l); ensure_source_h(l); }   // whole f32 source rows [1 | x | h_prev]
  void ensure_source_x(int l) {
    Layer& y = L[l];
    if (y.sx_valid) return;
    if (l > 0) ensure_h_f32(l - 1);   // (reads the layer below's f32 outputs)
    hipStream_t s = stream();
    timing.begin("build_source", s);
    CLSTM_LAUNCH(k_build_source, dim3(nblocks((size_t)N * (1 + y.ni))), dim3(256), 0, s, y.S.p, layer_input(l),
                 (size_t)N, y.ni, layer_input_ld(l), y.lds, ndir, (long long)N * y.lds);
    timing.end(s);
    y.sx_valid = true;
  }

  // ---- narrow layers, chip-filling minibatches: the recurrence batched over 16 lines per workgroup on the MFMA (lstm_mfma.h) ----
  // fwd_mfma: 0 never, 1 (default) from 640 lines per GPU on (measured crossover, profiles/r06_mfma_v3_vs_perline.txt: below
  // that the launch has fewer 16-line workgroups than the chip has CUs and the per-line kernel wins), 2 always (tests)
  bool mfma_eligible(const Layer& y) const {
#ifdef CLSTM_HIP_EMU
    return false;
#else
    const int mode = dbg_opt("fwd_mfma", 1);
    if (!mode || y.wide || bf16_gemm) return false;
    if (!((y.no == 64 || y.no == 100 || y.no == 128) && y.ni == 48)) return false;   // the instantiated (cells, inputs) geometries
    const double lim = 2147483000.0;   // 32-bit byte offsets inside one descriptor
    if ((double)N * ndir * 4 * y.no * 4 >= lim || (double)N * y.ldh * 4 >= lim || (double)N * y.lds * 4 >= lim) return false;
    if ((double)N * layer_input_ld((int)(&y - L.data())) * 4 >= lim) return false;
    return mode >= 2 || bs >= 640;
#endif
  }
#ifndef CLSTM_HIP_EMU
  template <int NO, int NI>
  void launch_mfma_no(Layer& y, bool fwd, hipStream_t s) {
    using Gm = MfmaGeom<NO, NI>;
    const int l = (int)(&y - L.data());
    if (y.mf_epoch != params_epoch) {
      y.Wmf.reserve((size_t)ndir * Gm::W_HALFS_PER_DIR + 64);
      y.mf_scale.reserve(8);
      MfmaPackArgs p{};
      p.v = v; p.ni = y.ni; p.no = y.no; p.nt = Gm::NT; p.kb = Gm::KB; p.W = y.Wmf.p; p.inv_scale = y.mf_scale.p;
      for (int d = 0; d < 2; d++) for (int q = 0; q < 4; q++) p.p_off[d][q] = y.pd.p_off[d][q];
      CLSTM_LAUNCH(k_pack_mfma, dim3(ndir), dim3(1024), 0, s, p);
      y.mf_epoch = params_epoch;
    }
    LstmMfmaArgs a{};
    a.W = y.Wmf.p; a.inv_scale = y.mf_scale.p; a.G = y.G.p; a.C = y.C.p; a.H = y.H.p; a.S = y.S.p; a.dH = y.dH.p; a.D = y.D.p;
    a.line_off = line_off.p; a.order = line_off.p + bs + 1; a.bs = bs; a.ndir = ndir; a.ldh = y.ldh; a.hofs = y.hofs;
    a.lds = y.lds; a.sofs = 1 + y.ni; a.sdir = (long long)N * y.lds; a.N = N;
    a.X = layer_input(l); a.ldx = layer_input_ld(l); a.store_s = 1; a.dbg = dbg_opt("mfma_dbg", 0);
#ifdef CLSTM_LSTM_PROF
    lstm_prof.reserve(128); a.prof = lstm_prof.p;
#endif
    static const bool smem_set = (coop_set_smem(lstm_fwd_mfma_kernel<NO, NI>, (size_t)Gm::SMEM), true);
    (void)smem_set; (void)fwd;
    CLSTM_LAUNCH((lstm_fwd_mfma_kernel<NO, NI>), dim3((unsigned)((bs + 15) / 16), (unsigned)ndir), dim3(512), (size_t)Gm::SMEM, s, a);
    g_path_count[16]++;
  }
#endif
  // bwd_mfma: the backward twin (lstm_mfma_bwd.h): 0 never, 1 (default) from 1024 lines per GPU on, 2 always (tests).  It gives up
  // the fused launch (the weight-gradient items then run behind it as a launch of their own, 0.54 ms at 1024 lines): 0.69 + 0.54
  // against 1.35 ms at 1024 lines, 0.95 + 1.09 against 2.7 at 2048 (profiles/r06_mfma_bwd_leaveout.txt)
  bool mfma_bwd_eligible(const Layer& y) const {
#ifdef CLSTM_HIP_EMU
    return false;
#else
    const int mode = dbg_opt("bwd_mfma", 1);
    if (!mode || y.wide || bf16_gemm) return false;
    if (!(y.no == 64 || y.no == 100)) return false;   // (128 cells: image + operand staging would need 176 KB of LDS)
    const double lim = 2147483000.0;   // 32-bit byte offsets inside one descriptor
    if ((double)N * ndir * 4 * y.no * 4 >= lim) return false;
    return mode >= 2 || bs >= 1024;
#endif
  }
#ifndef CLSTM_HIP_EMU
  template <int NO>
  void launch_mfma_bwd_no(Layer& y, const LstmSeqArgs& sa, hipStream_t s) {
    constexpr int NT = 2;
    using Gm = MfmaBwdGeom<NO, NT>;
    if (y.mfb_epoch != params_epoch) {
      y.Wmfb.reserve((size_t)ndir * Gm::W_HALFS_PER_DIR + 64);
      MfmaBwdPackArgs p{};
      p.v = v; p.ni = y.ni; p.no = y.no; p.ntl = Gm::NTL; p.kb = Gm::KB; p.nt = NT; p.W = y.Wmfb.p;
      for (int d = 0; d < 2; d++) for (int q = 0; q < 4; q++) p.p_off[d][q] = y.pd.p_off[d][q];
      CLSTM_LAUNCH(k_pack_mfma_bwd, dim3(16, (unsigned)ndir), dim3(256), 0, s, p);
      y.mfb_epoch = params_epoch;
    }
    LstmMfmaBwdArgs a{};
    a.W = y.Wmfb.p; a.G = sa.G; a.C = sa.C; a.dH = sa.dH; a.D = sa.D; a.line_off = sa.line_off; a.order = sa.order;
    a.bs = bs; a.ndir = ndir; a.N = N; a.prog_off = sa.prog_off; a.prog_base = sa.prog_base; a.dbg = dbg_opt("mfma_bwd_dbg", 0);
    static const bool smem_set = (coop_set_smem(lstm_bwd_mfma_kernel<NO, NT>, (size_t)Gm::SMEM), true);
    (void)smem_set;
    CLSTM_LAUNCH((lstm_bwd_mfma_kernel<NO, NT>), dim3((unsigned)((bs + 15) / 16), (unsigned)ndir), dim3(512), (size_t)Gm::SMEM, s, a);
    g_path_count[17]++;
  }
#endif
  // the narrow layer's backward recurrence: batched over lines on the MFMA where that pays, else one workgroup per line
  void launch_bwd_narrow(Layer& y, const LstmSeqArgs& a, hipStream_t s) {
#ifndef CLSTM_HIP_EMU
    if (mfma_bwd_eligible(y)) {
      if (y.no == 64) launch_mfma_bwd_no<64>(y, a, s);
      else launch_mfma_bwd_no<100>(y, a, s);
      check_launch();
      return;
    }
#endif
    launch_lstm(false, y.nk4, y.pd.ku, a, bs, y.nthreads, s);
  }
  void launch_mfma(Layer& y, bool fwd, hipStream_t s) {
#ifndef CLSTM_HIP_EMU
    if (y.no == 64) launch_mfma_no<64, 48>(y, fwd, s);
    else if (y.no == 100) launch_mfma_no<100, 48>(y, fwd, s);
    else launch_mfma_no<128, 48>(y, fwd, s);
    check_launch();
#else
    (void)y; (void)fwd; (void)s;
#endif
  }

  void forward() {
    REQUIRE(N > 0, "set_batch first");
    RoctxRange range_("clstm:forward");
    flush_line_off();
    repack();
    hipStream_t s = stream();
    if (forward_fused_eligible() && !(L.size() == 1 && mfma_eligible(L[0]))) { forward_fused(); return; }
    for (int l = 0; l < (int)L.size(); l++) {
      Layer& y = L[l];
      const int M = ndir * 4 * y.no;
      const bool x_from_hbf = bf16_gemm && bf16_rec && l > 0 && L[l - 1].fwd_persistent && L[l - 1].Hbf.p && y.WtbT.p && y.ni == ndir * L[l - 1].no && (y.ni & 1) == 0;
      // the lock-step recurrence of a wide layer + what follows it (bf16 source rows for the weight gradient); fx_ngx > 0: the
      // persistent kernel with the input projection folded in (lstm_wide.h:lstm_xcd_fwd_bf16_fx) -- false if it did not run
      auto run_wide = [&](LstmWideArgs w, int fx_ngx) {
        timing.begin("lstm_fwd", s);
        const bool ran = launch_lstm_wide(true, w, tmax, coop_sync, step_graphs, s, bf16_rec, fx_ngx);
        timing.end(s);
        if (!ran) return false;
        y.fwd_persistent = g_wide_persistent && bf16_rec;
        y.h_f32_valid = !(y.fwd_persistent && w.skip_h);   // (the per-step kernels store everything)
        y.sh_valid = !(y.fwd_persistent && w.skip_s);
        if (g_wide_persistent) g_path_count[0]++;
        if (fx_ngx) g_path_count[6]++;
        y.sbf_ready = y.fwd_persistent && w.Sbf;
        if (y.sbf_ready) {   // the non-recurrent columns of the bf16 source rows (the recurrence stored the h columns)
          const bool from16 = l > 0 && L[l - 1].fwd_persistent && L[l - 1].Hbf.p && y.ni == ndir * L[l - 1].no;
          if (l > 0 && !from16) ensure_h_f32(l - 1);
          // x columns that fill whole tiles of the weight-gradient GEMM are not copied: the GEMM reads them from Hbf itself
          const int R_ = 1 + y.ni + y.no, Cn_ = 4 * y.no;
          // (decided for the row count the backward pass WILL launch with -- R_ - 1 when the bias row is left out, the predicate of
          //  `dw_bias_out` there -- with the same tile-height function; should the backward pass still come out with another tile
          //  height, gemm_mc_check_a2 refuses the launch loudly instead of reading x rows nobody wrote.  Requiring BOTH R_ and
          //  R_ - 1 to fit, the first form of this fix, switched the path off at configs[4]: 1537 rows pick 192-row tiles.)
          const int R_bwd = wide_kp16_bwd(y.no) == 4 * y.no && gemm_bf16_big(R_ - 1, Cn_) ? R_ - 1 : R_;   // (= the backward's dbias condition)
          y.sbf_x_external = from16 && y.ni % gemm_mc_rows_per_tile(R_bwd, Cn_) == 0;
          if (y.sbf_x_external) {
            if (y.sbf_one_key != (long long)N) {
              CLSTM_LAUNCH(k_source_one_bf16, dim3(nblocks((size_t)N)), dim3(256), 0, s, y.Sbf.p, (size_t)N, y.ni + y.no, w.sbf_ld, ndir, w.sbf_dir);
              y.sbf_one_key = (long long)N;
            }
          } else {
            y.sbf_one_key = -1;
            CLSTM_LAUNCH(k_source_x_bf16, dim3(nblocks((size_t)N * ((y.ni >> 3) + 1))), dim3(256), 0, s, y.Sbf.p, from16 ? nullptr : layer_input(l),
                         from16 ? L[l - 1].Hbf.p : nullptr, from16 ? y.ni : layer_input_ld(l), (size_t)N, y.ni, y.ni + y.no, w.sbf_ld, ndir, w.sbf_dir);
          }
        }
        return true;
      };
      // bf16 mode, wide layer: no hoisted product at all when the persistent recurrence can take the input projection with it
      // (its operands are the bf16 rows the layer below left, or a bf16 copy of the input frames)
      bool fx_done = false;
      // experiment option fuse_wx (CLSTM_DEBUG): 0 never, 1 (default) layers of up to 128 inputs, 2 every eligible layer.  Measured at configs[4]
      // (profiles/r04_xcd_phase_cycles_fused_wx.txt): there is no idle shadow to hide the x-part in -- a step's "group wait" is
      // one L2 round trip of the poll, not waiting for late tiles -- so the fused work lands on the chain: +400 cycles per step
      // for 64 inputs (67 us per pass against the 121 us of product + bf16 copy it replaces: kept), +2,950 for 1024 inputs
      // (490 us against 321: not kept).  (read per pass: tests switch it inside one process)
      const int fx_mode = dbg_opt("fuse_wx", 1);
      if (fx_mode > 0 && (fx_mode > 1 || y.ni <= 128) && y.wide && bf16_gemm && bf16_rec && y.WtbT.p && (y.ni & 31) == 0 && (l == 0 || x_from_hbf)) {
        const int ngx = y.ni <= 128 ? 1 : y.ni <= 512 ? 4 : y.ni <= 1024 ? 8 : 0;
        if (ngx) {
          const unsigned short* xb = l > 0 ? L[l - 1].Hbf.p : nullptr;
          if (l == 0) {
            xbf.reserve((size_t)N * y.ni + 64);
            CLSTM_LAUNCH(k_to_bf16, dim3(nblocks((size_t)N * y.ni)), dim3(256), 0, s, layer_input(0), xbf.p, (size_t)N * y.ni);
            xb = xbf.p;
          }
          LstmWideArgs w = wide_args(y, true);
          w.Xb = xb; w.x_ld = y.ni; w.x_ni = y.ni; w.Wxb = y.WtbT.p; w.bias = y.bias;
          y.sx_valid = l == 0 && src0_ready;
          fx_done = run_wide(w, ngx);
        }
      }
      if (fx_done) { if (!y.sbf_ready) ensure_source_x(l); continue; }
      if (l > 0 && !x_from_hbf) ensure_h_f32(l - 1);   // the products below read the f32 outputs of the layer underneath
      if (mfma_eligible(y)) {
        // chip-filling minibatch of a narrow layer: the input product is part of the batched recurrence (lstm_mfma.h) -- no
        // hoisted W_x GEMM, no pre-activation array; the [1 | x] columns of the source rows are still the weight gradient's
        y.sx_valid = l == 0 && src0_ready;
        ensure_source_x(l);
        timing.begin("lstm_fwd", s);
        launch_mfma(y, true, s);
        y.h_f32_valid = y.sh_valid = true;
        timing.end(s);
        continue;
      }
      timing.begin("gemm_gates_x", s);
      if (x_from_hbf)
      {
        // the layer below left its outputs as a k-contiguous bf16 array: both operands go to LDS as they are
        g_path_count[2]++;
        gemm_b16kk(s, GemmOperand16{L[l - 1].Hbf.p, y.ni, (long long)N * y.ni}, GemmOperand16{y.WtbT.p, y.ni, (long long)M * y.ni},
                   StoreBias{y.G.p, M, y.bias}, (int)N, M, y.ni);
      } else if (bf16_gemm && bf16_rec && l == 0 && y.wide && y.WtbT.p && (y.ni & 7) == 0 && gemm_tile256((int)N, M)) {
        // first layer: a bf16 copy of the input frames (N x ni, a few MB) buys the bf16-source kernel with its 256 x 256
        // tiles and 16-byte stores for the product whose 4 M floats of pre-activations per frame-line are its whole cost
        xbf.reserve((size_t)N * y.ni + 64);
        CLSTM_LAUNCH(k_to_bf16, dim3(nblocks((size_t)N * y.ni)), dim3(256), 0, s, layer_input(0), xbf.p, (size_t)N * y.ni);
        g_path_count[2]++;
        gemm_b16kk(s, GemmOperand16{xbf.p, y.ni, (long long)N * y.ni}, GemmOperand16{y.WtbT.p, y.ni, (long long)M * y.ni},
                   StoreBias{y.G.p, M, y.bias}, (int)N, M, y.ni);
      } else if (bf16_gemm)
        gemm_bf16<GEMM_KC, GEMM_MC>(s, gemm_kc(layer_input(l), layer_input_ld(l), N, layer_input_slack(l)),
                                    gemm_mc(y.Wt, M, y.ni, y.wt_slack), StoreBias{y.G.p, M, y.bias}, (int)N, M, y.ni);
      else
        gemm_f32<GEMM_KC, GEMM_MC>(s, gemm_kc(layer_input(l), layer_input_ld(l), N), gemm_mc(y.Wt, M, y.ni, 0),
                                   StoreBias{y.G.p, M, y.bias}, (int)N, M, y.ni);
      timing.end(s);
      check_launch();
      // the [1 | x] columns of the f32 source rows: written now -- unless this is an upper layer whose weight-gradient
      // product will read the bf16 rows instead (decided after the recurrence below; ensure_source_x() then builds them
      // only if something still asks for them: a fallback path or the state API)
      y.sx_valid = l == 0 && src0_ready;
      if (l == 0 || !y.wide) ensure_source_x(l);   // (the register-resident recurrence of narrow layers reads whole source rows)
      LstmSeqArgs a{};
      a.Rpk = y.Rf; a.G = y.G.p; a.C = y.C.p; a.H = y.H.p; a.dH = nullptr; a.D = nullptr;
      a.line_off = line_off.p; a.order = line_off.p + bs + 1; a.no = y.no; a.ndir = ndir; a.ldh = y.ldh; a.hofs = y.hofs;
      a.S = y.S.p; a.lds = y.lds; a.sofs = 1 + y.ni; a.sdir = (long long)N * y.lds;
#ifdef CLSTM_LSTM_PROF
      lstm_prof.reserve(128); a.prof = lstm_prof.p;
#endif
      if (y.wide) run_wide(wide_args(y, true), 0);
      else {
        timing.begin("lstm_fwd", s);
        launch_lstm(true, y.nk4, y.pd.ku, a, bs, y.nthreads, s); y.h_f32_valid = y.sh_valid = true;
        timing.end(s);
      }
      if (!y.sbf_ready) ensure_source_x(l);
    }
    const int nc = desc.nclasses;
    const float* W1 = v + sm_off;
    // the top layer's output rows are [1 | h]: they ARE the softmax layer's source rows
    if (nc <= SMX_COLS) {   // logits, limexp and normalisation in one kernel (softmax_fused.h)
      timing.begin("gemm_softmax", s);
      softmax_fwd(s, gemm_kc(L.back().hrow(), L.back().ldh, N), W1, (long long)nc * (1 + sm_ni), Z.p, (int)N, nc, sm_ni, fwd_nanflag(), step_no() + 1);
      timing.end(s);
      check_launch();
    } else {
      timing.begin("gemm_softmax", s);
      if (bf16_gemm && gemm_x3_on && gemm_bf16_big((int)N, nc))   // bf16 modes: f32-grade bf16 x 3 on 128 x 128 tiles (88 -> ~30 us at configs[4])
        gemm_x3_big<GEMM_KC, GEMM_MC>(s, gemm_kc(L.back().hrow(), L.back().ldh, N, 32), gemm_mc(W1 + nc, nc, sm_ni, 0),
                                      StoreBias{Z.p, nc, W1}, (int)N, nc, sm_ni);
      else
      gemm_f32<GEMM_KC, GEMM_MC>(s, gemm_kc(L.back().hrow(), L.back().ldh, N), gemm_mc(W1 + nc, nc, sm_ni, 0),
                                 StoreBias{Z.p, nc, W1}, (int)N, nc, sm_ni);
      timing.end(s);
      check_launch();
      timing.begin("softmax_norm", s);
      CLSTM_LAUNCH(k_softmax_norm, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, Z.p, nc, (size_t)N, fwd_nanflag(), step_no() + 1);
      timing.end(s);
      check_launch();
    }
  }

  // ---- the forward half as one launch: W_x GEMM producers + recurrence + softmax consumers (lstm_fwd_fused.h) ----
  bool forward_fused_eligible() {
    if (!overlap || L.size() != 1 || bf16_gemm) return false;
    const Layer& y = L[0];
    if (!y.Wk || y.wide || y.nthreads < 64 * FWD_CW || y.no % 16 == 0 || y.wk_njp > FWD_JW * FWD_CW || w1k_kps > 16 * FWD_CG) return false;     // >= 5 waves, the last lane owns no cell
    if (desc.nclasses > SMX_COLS || bs > PROG_LINES || bs >= (1 << 18) || tmax >= (1 << 16)) return false;
    if ((overlap == 1) && (tmax < 64 || N < 2048)) return false;                 // too small to profit
    if ((double)y.H.cap * 4.0 >= 2147483000.0) return false;                     // 32-bit byte offsets inside one descriptor
    // the recurrence workgroups (one per CU, dispatched first) wait for producers: CUs must be left for those
    return bs * ndir <= device_cu_count() - std::max(8, device_cu_count() / 8);
  }
  void build_fwd_items() {
    if (fw_key == line_off_h && fw_ncitems > 0) return;
    fw_chunks = (tmax + 15) / 16;
    std::vector<int> items;
    // producers: time order from chunk 1 on (chunk 0 is the recurrence workgroup's own), the longest lines of a chunk
    // first; records of 4 ints: code, the line's first frame, its length, 0
    for (int c = 1; c < fw_chunks; c += FWD_FT)
      for (int i = 0; i < bs; i++) {
        const int b = order_h[i], T = line_off_h[b + 1] - line_off_h[b];
        if (T > 16 * c)
          for (int dir = 0; dir < ndir; dir++) { items.push_back(b << 13 | dir << 12 | c); items.push_back(line_off_h[b]); items.push_back(T); items.push_back(0); }
      }
    fw_npitems = (int)items.size() / 4;
    std::vector<std::pair<int, int>> cons;                    // (iteration at which the block is complete, code)
    for (int b = 0; b < bs; b++) {
      const int T = line_off_h[b + 1] - line_off_h[b];
      for (int blk = 0; 16 * blk < T; blk++)
        cons.push_back({std::max(std::min(16 * blk + 16, T), ndir > 1 ? T - 16 * blk : 0), b << 12 | blk});
    }
    std::stable_sort(cons.begin(), cons.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; });
    fw_ncitems = (int)cons.size();
    for (auto& c : cons) items.push_back(c.second);
    fw_items.reserve(items.size() + 8);
    fw_flags.reserve((size_t)ndir * bs * fw_chunks + 8);      // (zero-filled when it grows; flags carry the launch epoch)
    hipStream_t s = stream();
    int* stage = (int*)ring.acquire(items.size() * sizeof(int));
    memcpy(stage, items.data(), items.size() * sizeof(int));
    HIPCHECK(hipMemcpyAsync(fw_items.p, stage, items.size() * sizeof(int), hipMemcpyHostToDevice, s));
    ring.commit(s);
    fw_key = line_off_h;
  }
  void forward_fused() {
    hipStream_t s = stream();
    Layer& y = L[0];
    build_fwd_items();
    if (++fw_epoch > (1 << 30)) fw_epoch = 1;
    fw_prog_base += tmax + 64;
    if (fw_prog_base > (1 << 30)) fw_prog_base = 1024;
    fw_launches++;
    g_path_count[5]++;
    FwdFusedKernelArgs k{};
    LstmSeqArgs& a = k.a;
    a.Rpk = y.Rf; a.G = y.G.p; a.C = y.C.p; a.H = y.H.p; a.dH = nullptr; a.D = nullptr;
    a.line_off = line_off.p; a.order = line_off.p + bs + 1; a.no = y.no; a.ndir = ndir; a.ldh = y.ldh; a.hofs = y.hofs;
    a.S = y.S.p; a.lds = y.lds; a.sofs = 1 + y.ni; a.sdir = (long long)N * y.lds; a.bs = bs;
    a.prog_off = (long long)y.H.cap - 64 - PROG_WORDS;
    REQUIRE(a.prog_off >= (long long)N * y.ldh + 16, "internal: progress words overlap the output rows");
    a.prog_base = fw_prog_base;
    a.gflag = fw_flags.p; a.gchunks = fw_chunks; a.gepoch = fw_epoch; a.timeouts = dev_err_words() + 2;   // (the forward launch's own error word)
    FwdFusedArgs& h = k.h;
    h.X = layer_input(0); h.ldx = layer_input_ld(0); h.x_elems = (long long)N * h.ldx + 32;
    h.Wk = y.Wk; h.kp = y.wk_kp; h.njp = y.wk_njp; h.bias = y.bias;
    h.pitems = fw_items.p; h.npitems = fw_npitems; h.gflag = fw_flags.p;
    h.W1k = W1k; h.kps = w1k_kps; h.sm_k = sm_ni; h.b1 = v + sm_off; h.Z = Z.p; h.nc = desc.nclasses;
    h.citems = fw_items.p + 4 * fw_npitems; h.ncitems = fw_ncitems;
    h.nanflag = fwd_nanflag(); h.step_no = step_no() + 1;
    h.prog = (const int*)(y.H.p + a.prog_off);
    h.nrec = bs * ndir; h.npb = fw_npitems;
    const unsigned nblk = (unsigned)(h.nrec + h.npb + fw_ncitems);   // one item per helper workgroup
    y.sx_valid = src0_ready;
    ensure_source_x(0);
    static const char* trace_path = getenv("CLSTM_FW_TRACE");   // diagnostics: wall-clock stamps of every workgroup / item of the launch
    const size_t trace_rows = (size_t)h.nrec + fw_npitems + fw_ncitems;
    if (trace_path) { dw_trace.reserve(trace_rows * 4); HIPCHECK(hipMemsetAsync(dw_trace.p, 0, trace_rows * 4 * sizeof(long long), s)); h.trace = dw_trace.p; }
    timing.begin("lstm_fwd", s);
    REQUIRE(launch_lstm_fwd_fused(y.nk4, y.pd.ku, k, nblk, y.nthreads, s), "internal: no fused forward instantiation");
    timing.end(s);
    if (trace_path) {
      HIPCHECK(hipStreamSynchronize(s));
      std::vector<long long> t(trace_rows * 4);
      HIPCHECK(hipMemcpy(t.data(), dw_trace.p, t.size() * sizeof(long long), hipMemcpyDeviceToHost));
      if (FILE* f = fopen(trace_path, "w")) {
        fprintf(f, "# %d recurrence rows (start - end -), %d producer items (start - done chunk), %d consumer items (start ready done ready_iteration); 100 MHz ticks\n",
                h.nrec, fw_npitems, fw_ncitems);
        for (size_t i = 0; i < trace_rows; i++) fprintf(f, "%lld %lld %lld %lld\n", t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
        fclose(f);
      }
    }
  }

  // overlapped weight-gradient GEMM: 1 = bf16 MFMA on f32 operands split into bf16 terms (gemm_dw.h), 0 = f32 MFMA
  // (experiment option dw_x3=0)
  int dw_x3 = dbg_opt("dw_x3", 1);
  // terms per operand of those split products and of the softmax layer's (gemm_x3): 3 = operand-exact (x1 + x2 + x3 is the f32
  // itself, six products), 2 = hi + lo, three products (< 2^-16 per product; rounds 3-5)
  int split_terms = dbg_opt("split_terms", 3);
  // the softmax layer's backward products W.d / x.d the same way (gemm_x3, gemm_bf16.h); experiment option gemm_x3=0 (CLSTM_DEBUG): f32 MFMA.
  // NOT the forward product W_x.x: its ~2^-17 relative error per product shows up in gate pre-activations that cancel to
  // ~0 (a tanh gate at -0.0021 came out 5.6e-6 off where the parity bar allows 2.2e-6), and with K = 49 the split costs
  // more staging than it saves MFMA time (28.5 vs 20.9 us).
  bool gemm_x3_on = dbg_opt("gemm_x3", 1) != 0;
  // exact-f32 mode, wide layers: the persistent BACKWARD recurrence as an f32-grade x3 product on the bf16 MFMA (lstm_wide.h:
  // lstm_xcd_bwd_x3) like the backward GEMMs of this mode; off with them (gemm_x3=0 / strict f32) or alone (rec_x3=0, CLSTM_DEBUG options;
  // read per pass: tests compare both kernels in one process)
  bool rec_x3() const {
    if (bf16_gemm || bf16_rec || !gemm_x3_on) return false;
    return dbg_opt("rec_x3", 1) != 0;
  }
  // split-K slabs for the weight-gradient GEMMs: enough workgroups to cover the 256 CUs
  // big tiles of the contraction-major bf16 product (gemm_b16mc: 256 or 192 rows x 256 columns): one workgroup per CU, never a second round
  int pick_split_mc(int R, int Cn, int nbatch) const {
    const int th = gemm_mc_tile_rows(R);
    const long long tiles = (long long)((R + th - 1) / th) * ((Cn + 255) / 256) * nbatch;
    long long want = device_cu_count() / tiles;
    const long long maxs = (N + 63) / 64;
    if (want > maxs) want = maxs;
    if (want > 64) want = 64;
    if (want < 1) want = 1;
    return (int)want;
  }
  int pick_split(int R, int Cn, int nbatch = 1, int tile = GEMM_BT) const {
    const long long tiles = (long long)((R + tile - 1) / tile) * ((Cn + tile - 1) / tile) * nbatch;
    const long long target = tile == GEMM_BT ? 640 : 480;   // 64 x 64 tiles: 640 measured best (272: slower); 128 x 128 tiles: two workgroups per CU
    long long want = (target + tiles - 1) / tiles;
    if (tile == 256) want = device_cu_count() / tiles;          // 256 x 256 tiles: one workgroup per CU, never a second round
    const long long maxs = (N + 63) / 64;   // at least 64 frames per slab
    if (want > maxs) want = maxs;
    if (want > 64) want = 64;
    if (want < 1) want = 1;
    return (int)want;
  }

  // ---- overlap machinery ---------------------------------------------------------------------------------------
  bool overlap_eligible(const Layer& y) {
    if (!overlap || y.wide || y.no % 16 == 0) return false;          // the reporting lane must own no cell
    if (bs > PROG_LINES) return false;
    if ((overlap == 1) && (tmax < 64 || N < 2048)) return false;     // too small to profit
    if ((double)y.D.cap * 4.0 >= 2147483000.0) return false;         // 32-bit byte offsets inside one descriptor
    return overlap == 2 || y.nthreads >= 256;                        // one launch, two workgroup roles (lstm_bwd_dw.h)
  }
  // k-tile tables and slabs of the chunked weight-gradient GEMM for the current batch geometry (rebuilt only when the
  // line lengths change).  Chunks are ranges of recurrence iterations, longest first: the work left when the
  // recurrence ends is what the last (short) chunk holds.
  void build_dw_tables() {
    if (dw_key == line_off_h && dw_nslabs > 0) return;
    std::vector<int> cb;   // chunk ends (iterations), multiples of 8 except the last
    // equal chunks of 16 iterations measured best at the bench shape (8: 0.407 ms per step, 16: 0.356, 32: 0.359,
    // 48: 0.360, a decreasing plan 64..16: 0.3615)
    for (int done = 16; done < tmax; done += 16) cb.push_back(done);
    cb.push_back(tmax);
    const int tiles_per_slab = std::min(DW_STAB_MAX, std::max(8, (int)((N / 16 + 15) / 16)));   // ~16 slabs per direction (a slab's table must fit the items' LDS copy)
    std::vector<std::vector<int>> tab(ndir);                            // (first frame, count) pairs
    struct Sl { int tb, nt, need, dir, chunk, part; };
    std::vector<std::vector<Sl>> sl(ndir);
    for (int dir = 0; dir < ndir; dir++) {
      int cs = 0;
      for (size_t c = 0; c < cb.size(); c++) {
        const int ce = cb[c];
        const int t0 = (int)tab[dir].size() / 2;
        for (int b = 0; b < bs; b++) {
          const int off = line_off_h[b], T = line_off_h[b + 1] - off;
          if (T <= cs) continue;
          const int e = std::min(ce, T);
          // iterations [cs, e): dir 0 of the backward pass visits frame T-1-it, dir 1 frame it
          const int f_lo = dir == 0 ? T - e : cs, f_hi = dir == 0 ? T - cs : e;
          for (int f = f_lo; f < f_hi; f += 16) { tab[dir].push_back(off + f); tab[dir].push_back(std::min(16, f_hi - f)); }
        }
        const int nt = (int)tab[dir].size() / 2 - t0;
        int parts = std::max(1, (nt + tiles_per_slab - 1) / tiles_per_slab);
        // The items of the LAST chunks cannot start before the recurrence ends, so their latency -- a serial walk over a
        // slab's frames, ~1 us per 32 -- is the launch's tail (profiles/r02_dw_timeline.txt): cut those chunks into
        // more, shorter slabs (>= 2 table entries each).
        const int tail_parts = 4, tail_chunks = 3;   // (parts 1: 118 us, 4: 112, 8: 115 -- more items than free CUs at the end)
        if (c + tail_chunks >= cb.size()) parts = std::max(parts, std::min(tail_parts, std::max(1, nt / 2)));   // (those whose items still run when the recurrence ends)
        for (int p = 0; p < parts; p++) {
          const int a0 = t0 + (int)((long long)nt * p / parts), a1 = t0 + (int)((long long)nt * (p + 1) / parts);
          sl[dir].push_back(Sl{a0, a1 - a0, ce, dir, (int)c, p});
        }
        cs = ce;
      }
    }
    dw_slabs_per_dir = (int)sl[0].size();
    for (int dir = 1; dir < ndir; dir++) REQUIRE((int)sl[dir].size() == dw_slabs_per_dir, "internal: asymmetric slab lists");
    dw_ntiles_max = 0;
    for (int dir = 0; dir < ndir; dir++) dw_ntiles_max = std::max(dw_ntiles_max, (int)tab[dir].size() / 2);
    // readiness order: chunk, then part, then direction
    std::vector<DwSlab> slabs;
    for (int i = 0; i < dw_slabs_per_dir; i++)
      for (int dir = 0; dir < ndir; dir++) {
        const Sl& x = sl[dir][i];
        slabs.push_back(DwSlab{x.tb, x.nt, x.need, dir, dir * dw_slabs_per_dir + i, {0, 0, 0}});
      }
    dw_nslabs = (int)slabs.size();
    const size_t nk = (size_t)ndir * dw_ntiles_max * 2, nsw = slabs.size() * sizeof(DwSlab) / sizeof(int);
    dw_ktab.reserve(nk + 8);
    dw_slabs.reserve(nsw + 8);
    hipStream_t s = stream();
    int* stage = (int*)ring.acquire((nk + nsw) * sizeof(int));
    for (int dir = 0; dir < ndir; dir++) {
      std::fill(stage + (size_t)dir * dw_ntiles_max * 2, stage + (size_t)(dir + 1) * dw_ntiles_max * 2, 0);
      std::copy(tab[dir].begin(), tab[dir].end(), stage + (size_t)dir * dw_ntiles_max * 2);
    }
    memcpy(stage + nk, slabs.data(), nsw * sizeof(int));
    HIPCHECK(hipMemcpyAsync(dw_ktab.p, stage, nk * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(dw_slabs.p, stage + nk, nsw * sizeof(int), hipMemcpyHostToDevice, s));
    ring.commit(s);
    dw_key = line_off_h;
  }
  // backward recurrence + the chunked weight-gradient GEMM as two workgroup roles of one launch
  void backward_layer_overlapped(Layer& y, LstmSeqArgs a, int R, int Cn) {
    hipStream_t s = stream();
    build_dw_tables();
    const int M = ndir * 4 * y.no;
    const long long prog_off = (long long)y.D.cap - 64 - PROG_WORDS;
    REQUIRE(prog_off >= (long long)N * M, "internal: progress words overlap the deltas");
    prog_base += tmax + 64;
    dw_launches++;
    if (prog_base > (1 << 30)) prog_base = 1024;   // (words left from ~5 million launches ago could look complete: harmless in practice, D is rewritten)
    a.prog_off = prog_off;
    a.prog_base = prog_base;
    DevBuf<float>& partial = layer_partial(y);
    partial.reserve((size_t)ndir * dw_slabs_per_dir * R * Cn);
    GemmDwArgs g{};
    g.S = y.S.p; g.sdir = (long long)N * y.lds; g.lds = y.lds; g.s_elems = (long long)N * ndir * y.lds + 3;
    g.D = y.D.p; g.M = M; g.no4 = 4 * y.no; g.d_elems = (long long)N * M + 3;
    g.ktab = dw_ktab.p; g.ntiles_max = dw_ntiles_max; g.slabs = (const DwSlab*)dw_slabs.p; g.nslabs = dw_nslabs;
    g.prog = (const int*)(y.D.p + prog_off); g.line_off = line_off.p; g.bs = bs; g.prog_base = prog_base;
    g.partial = partial.p; g.R = R; g.Cn = Cn;
    g.gx = (unsigned)((Cn + GEMM_BT - 1) / GEMM_BT); g.gy = (unsigned)((R + GEMM_BT - 1) / GEMM_BT);
    g.timeouts = dev_err_words() + 1;
    if (!dw_queue.p) dw_queue.reserve(5 * PROG_STRIDE);
    g.ndir = ndir;
    g.minprog = dw_queue.p + PROG_STRIDE;   // own 128-byte lines
    g.tcap = tmax + 32;
    g.done = nullptr; g.done_target = 0;
    g.x3 = dw_x3;
    g.terms = split_terms;
    unsigned nextra = 0;
    if (dwx_active && &y == &L.back()) {
      const int xR = 1 + sm_ni, xCn = desc.nclasses;
      g.xS = y.srow(); g.xlds = y.ldh; g.xs_elems = (long long)N * y.ldh + 3;
      g.xD = Dz.p; g.xM = xCn; g.xd_elems = (long long)N * xCn + 3;
      g.xtab = dwx_tab.p; g.xslabs = (const DwSlab*)(dwx_tab.p + 2 * dwx_entries); g.xnslabs = dwx_nslabs;
      g.xpartial = partial_sm.p; g.xR = xR; g.xCn = xCn;
      g.xgx = (unsigned)((xCn + GEMM_BT - 1) / GEMM_BT); g.xgy = (unsigned)((xR + GEMM_BT - 1) / GEMM_BT);
      nextra = (unsigned)dwx_nslabs * g.xgx * g.xgy;
    }
    static const char* trace_path = getenv("CLSTM_DW_TRACE");   // diagnostics: wall-clock stamps of every workgroup of the fused launch
    const size_t trace_rows = (size_t)bs * ndir + (size_t)((dw_nslabs + 7) / 8) * 8 * g.gx * g.gy + 8;
    if (trace_path) { dw_trace.reserve(trace_rows * 4); g.trace = dw_trace.p; g.trace_base = bs * ndir; }
    const unsigned nblk = 1u + nextra + (unsigned)((dw_nslabs + 7) / 8) * 8u * g.gx * g.gy;   // the monitor + the independent items + one per item
#ifndef CLSTM_HIP_EMU
    if (y.nthreads >= 256 && !mfma_bwd_eligible(y)) {   // ONE launch: the recurrence's workgroups first, the GEMM's (one (slab, tile) item each) behind them
      timing.begin("lstm_bwd", s);
      g.done = g.minprog + 2 * PROG_STRIDE;   // own 128-byte line behind the monitor's words; zero-filled once, then only added to
      dw_done_total += (unsigned)(bs * ndir);
      g.done_target = (int)dw_done_total;
      REQUIRE(launch_lstm_bwd_dw(y.nk4, y.pd.ku, a, g, bs * ndir, nblk, y.nthreads, s), "internal: no fused instantiation");
      timing.end(s);
      if (trace_path) {
        HIPCHECK(hipStreamSynchronize(s));
        std::vector<long long> h(trace_rows * 4);
        HIPCHECK(hipMemcpy(h.data(), dw_trace.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(trace_path, "w")) {
          fprintf(f, "# rows 0..%d: recurrence workgroups (start, -, end); then one row per (slab, tile) item: start ready done need_it; 100 MHz ticks\n", bs * ndir - 1);
          for (size_t i = 0; i < trace_rows; i++) fprintf(f, "%lld %lld %lld %lld\n", h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
          fclose(f);
        }
      }
      return;
    }
#endif
    // two launches one after the other (host emulator; layers too narrow for the GEMM role's 256 threads when the
    // tests force the path): the items find every progress word complete
    // (... and minibatches whose backward recurrence runs batched on the MFMA, lstm_mfma_bwd.h: it marks its lines complete)
    timing.begin("lstm_bwd", s);
    launch_bwd_narrow(y, a, s);
    timing.end(s);
    timing.begin("gemm_gates_dw", s);
    if (g.x3 && g.terms >= 3) CLSTM_LAUNCH(gemm_dw_kernel<3>, dim3(nblk), dim3(256), 0, s, g);
    else if (g.x3) CLSTM_LAUNCH(gemm_dw_kernel<2>, dim3(nblk), dim3(256), 0, s, g);
    else CLSTM_LAUNCH(gemm_dw_kernel<0>, dim3(nblk), dim3(256), 0, s, g);
    timing.end(s);
    check_launch();
  }

  void backward() {
    REQUIRE(N > 0, "set_batch first");
    const bool fuse = fuse_update;   // consumed here: an exception below must not leave it set for a later pass
    fuse_update = false;
    const bool peer = peer_step && comm;
    peer_step = false; peer_pending = false;
    float* const g_local = g;
    float* const gdst = peer ? comm->peer.slot_ptr(comm->peer.seq + 1) : g_local;   // where the reductions leave the fresh gradient
    nbackward++;
    RoctxRange range_("clstm:backward");
    update_applied = false;
    pending_red.clear();   // (entries a throwing pass left behind must not be reduced -- or, with the update fused in, APPLIED -- by this one)
    flush_line_off();
    repack();
    hipStream_t s = stream();
    const int nc = desc.nclasses;
    const float* W1 = v + sm_off;
    // every entry of g is assigned by exactly one reduce below: no clearing pass
    // SoftmaxLayer::backward (clstm.cc:411-417): x.d = W^T z.d ; W.d += z.d [1;x]^T
    Layer& top = L.back();
    {  // (a side stream for this GEMM was measured on MI355X: no gain -- the recurrence workgroups it would
       // overlap with slow down by as much -- so everything stays on one stream)
      const int R = 1 + sm_ni, Cn = nc;
      // (bf16 modes, shapes that fill 128 x 128 tiles: the two products as launches of the big-tile f32-grade kernel)
      const bool sm_big = bf16_gemm && gemm_x3_on && gemm_bf16_big(R, Cn) && gemm_bf16_big((int)N, sm_ni);
      int ns = sm_big ? pick_split(R, Cn, 1, GB2_BT) : pick_split(R, Cn);
      // W.d depends on nothing the backward recurrence produces: when the top layer's backward runs as the fused launch
      // (lstm_bwd_dw.h) its slabs are items of THAT launch -- they execute on the idle half of the chip during the ~14 us
      // before the recurrence's first chunk is released -- and only x.d stays in front of the recurrence.
      // (only while the recurrence leaves CUs idle: with 256 lines the same items cost the fused launch +42 us for 19 saved)
      dwx_active = !bf16_gemm && (dw_x3 & 1) && overlap_eligible(top) &&
                   (long long)bs * ndir * 4 <= 3LL * device_cu_count();
      if (dwx_active) {
        if (dwx_N != N) {   // contiguous frames: entries of 16, slabs of 32 entries (512 frames: one short item each)
          dwx_entries = (int)((N + 15) / 16);
          const int eps = std::min(DW_STAB_MAX, std::max(32, (dwx_entries + 63) / 64));   // entries per slab: at most ~64 slabs to reduce
          dwx_nslabs = (dwx_entries + eps - 1) / eps;
          const size_t nsw = (size_t)dwx_nslabs * sizeof(DwSlab) / sizeof(int);
          dwx_tab.reserve((size_t)2 * dwx_entries + nsw + 8);
          int* stage = (int*)ring.acquire(((size_t)2 * dwx_entries + nsw) * sizeof(int));
          for (int e = 0; e < dwx_entries; e++) { stage[2 * e] = 16 * e; stage[2 * e + 1] = (int)std::min<long long>(16, N - 16LL * e); }
          DwSlab* sl = (DwSlab*)(stage + 2 * dwx_entries);
          for (int i = 0; i < dwx_nslabs; i++) sl[i] = DwSlab{eps * i, std::min(eps, dwx_entries - eps * i), 0, 0, i, {0, 0, 0}};
          HIPCHECK(hipMemcpyAsync(dwx_tab.p, stage, ((size_t)2 * dwx_entries + nsw) * sizeof(int), hipMemcpyHostToDevice, s));
          ring.commit(s);
          dwx_N = N;
        }
        ns = dwx_nslabs;
      }
      partial_sm.reserve((size_t)ns * R * Cn);
      // W.d (split-K slabs) and x.d in ONE launch: two small independent products, each mostly prologue and
      // epilogue latency on its own (13.9 + 12.4 us back to back)
      timing.begin("gemm_softmax_dw_dx", s);
      if (dwx_active)
        gemm_x3<GEMM_KC, GEMM_KC>(s, gemm_kc(Dz.p, nc, N), gemm_kc(W1 + nc, nc, sm_ni, 0), StorePlain{top.dH.p, sm_ni}, (int)N, sm_ni, nc,
                                  1, 1, split_terms);
      else if (sm_big) {
        gemm_x3_big<GEMM_MC, GEMM_MC>(s, gemm_mc(top.srow(), top.ldh, N, 32), gemm_mc(Dz.p, nc, N, 32), StorePartial{partial_sm.p, R, Cn}, R, Cn, (int)N, ns);
        gemm_x3_big<GEMM_KC, GEMM_KC>(s, gemm_kc(Dz.p, nc, N, 32), gemm_kc(W1 + nc, nc, sm_ni, 0), StorePlain{top.dH.p, sm_ni}, (int)N, sm_ni, nc);
      } else if (gemm_x3_on)
        gemm_x3_pair<GEMM_MC, GEMM_MC, StorePartial, GEMM_KC, GEMM_KC, StorePlain>(
            s, gemm_problem(gemm_mc(top.srow(), top.ldh, N), gemm_mc(Dz.p, nc, N), R, Cn, (int)N, ns),
            StorePartial{partial_sm.p, R, Cn},
            gemm_problem(gemm_kc(Dz.p, nc, N), gemm_kc(W1 + nc, nc, sm_ni, 0), (int)N, sm_ni, nc), StorePlain{top.dH.p, sm_ni},
            split_terms);
      else
      gemm_f32_pair<GEMM_MC, GEMM_MC, StorePartial, GEMM_KC, GEMM_KC, StorePlain>(
          s, gemm_problem(gemm_mc(top.srow(), top.ldh, N), gemm_mc(Dz.p, nc, N), R, Cn, (int)N, ns),
          StorePartial{partial_sm.p, R, Cn},
          gemm_problem(gemm_kc(Dz.p, nc, N), gemm_kc(W1 + nc, nc, sm_ni, 0), (int)N, sm_ni, nc), StorePlain{top.dH.p, sm_ni});
      timing.end(s);
      // the slabs are reduced together with the top layer's weight-gradient slabs below
      sm_red = ReduceDesc{partial_sm.p, nullptr, (long long)sm_off, ns, 1, R, Cn, nc};
    }
    check_launch();
    for (int l = (int)L.size() - 1; l >= 0; l--) {
      Layer& y = L[l];
      const int M = ndir * 4 * y.no;
      LstmSeqArgs a{};
      a.Rpk = y.Rb; a.G = y.G.p; a.C = y.C.p; a.H = y.H.p; a.dH = y.dH.p; a.D = y.D.p;
      a.line_off = line_off.p; a.order = line_off.p + bs + 1; a.no = y.no; a.ndir = ndir; a.bs = bs;
      // W.d += delta [1; x_t; h_{t-1}]^T for the four gates of each direction
      // (both directions in one batched launch: half the slabs per direction fill the chip)
      const int R = 1 + y.ni + y.no, Cn = 4 * y.no;
      int ns;
      bool bwd_persistent = false;
      a.prog_off = -1; a.prog_base = 0;
      if (!bf16_gemm && overlap_eligible(y)) {
        // the recurrence and the weight-gradient GEMM run side by side (gemm_dw.h)
        backward_layer_overlapped(y, a, R, Cn);
        ns = dw_slabs_per_dir;
        timing.begin("reduce_scatter", s);
      } else {
      timing.begin("lstm_bwd", s);
      int skipped_d = 0;
      if (y.wide) {
        const LstmWideArgs w = wide_args(y, false);
        launch_lstm_wide(false, w, tmax, coop_sync, step_graphs, s, bf16_rec, 0, rec_x3());
        skipped_d = w.skip_d;
      } else launch_bwd_narrow(y, a, s);
      timing.end(s);
      bwd_persistent = y.wide && g_wide_persistent;
      y.d_f32_valid = !(bwd_persistent && bf16_rec && skipped_d);
      if (bwd_persistent) g_path_count[1]++;
      }
      const bool dw_from_bf16 = bf16_gemm && bf16_rec && bwd_persistent && y.sbf_ready && y.Dbf.p && gemm_bf16_big(R, Cn);
      const bool dw_bias_out = dw_from_bf16 && y.dbias.p && gemm_bf16_big(R - 1, Cn);
      // exact-f32 mode, wide layer: the backward products as f32-grade bf16 x 3 on 128 x 128 tiles (gemm_x3_128_kernel) -- what
      // narrow layers already do inside their fused backward launch; gemm_x3=0 (CLSTM_DEBUG) / clstm_net_set_strict_f32: the f32 MFMA
      const bool x3_big = !bf16_gemm && y.wide && gemm_x3_on && gemm_bf16_big(R, Cn);
      if (bf16_gemm || !overlap_eligible(y))
        ns = dw_bias_out && gemm_tile256(R - 1, Cn) ? pick_split_mc(R - 1, Cn, ndir)
             : dw_from_bf16 && gemm_tile256(R, Cn) ? pick_split_mc(R, Cn, ndir)
             : (bf16_gemm || x3_big) && gemm_bf16_big(R, Cn) ? pick_split(R, Cn, ndir, GB2_BT) : pick_split(R, Cn, ndir);
      if (!dw_from_bf16) { ensure_source(l); ensure_delta_f32(l); }   // the f32-source products below read S and D
      DevBuf<float>& pbuf = layer_partial(y);
      bool dx_done = false;   // the input deltas rode the weight-gradient launch (gemm_dw_dx)
      auto do_dw = [&](hipStream_t q) {
        if (bf16_gemm || !overlap_eligible(y)) {
          pbuf.reserve((size_t)ndir * ns * R * Cn);
          timing.begin("gemm_gates_dw", q);
          if (dw_from_bf16) {
            // both operands bf16 as their producers left them (deltas: the persistent backward recurrence; sources: the
            // forward pass), transposed by the LDS on the way into the MFMA
            const int ldsb = y.ni + y.no + 8;
            g_path_count[4]++;
            const GemmOperand16B a2 = y.sbf_x_external ? GemmOperand16B{L[l - 1].Hbf.p, y.ni, (long long)N * y.ni, 0} : GemmOperand16B{nullptr, 0, 0, 0};
            // ... and, where the layer also owes input deltas from the same bf16 delta array, BOTH products as one launch
            // (gemm_bf16.h:gemm_dw_dx_kernel: apart, each leaves a quarter of the chip idle)
            float* const dxp = l > 0 ? L[l - 1].dH.p : nullptr;
            if (dw_bias_out && dxp && wide_kp16_bwd(y.no) == 4 * y.no && y.Wtb.p &&
                gemm_dw_dx(q, GemmOperand16B{y.Sbf.p, ldsb, (long long)N * ndir * ldsb, (long long)N * ldsb}, GemmOperand16B{y.Dbf.p, M, (long long)N * M, 4LL * y.no},
                           StorePartialShift{pbuf.p, R, Cn}, R - 1, Cn, (int)N, ns, ndir, a2, y.sbf_x_external ? y.ni : 0,
                           GemmOperand16{y.Dbf.p, M, (long long)N * M}, GemmOperand16{y.Wtb.p, M, (long long)y.ni * M}, StorePlain{dxp, y.ni}, (int)N, y.ni, M)) {
              CLSTM_LAUNCH(k_bias_rows, dim3((unsigned)(((size_t)ndir * Cn + 63) / 64)), dim3(64), 0, q, (const float*)y.dbias.p, pbuf.p, bs, ndir, ns, R, Cn);
              g_path_count[13]++; g_path_count[3]++; g_path_count[14]++;
              dx_done = true;
            } else if (dw_bias_out) {
              // The bias row W.d[:,0] += sum_b y.d (clstm_compute.cc:301) is not a row of this product: 1 + ni + no rows are one
              // more than a whole number of row panels at both configs[4] layers (1537 = 6 x 256 + 1: a seventh panel, 14 % of
              // the product, for one row; 577 = 3 x 192 + 1) -- the persistent backward recurrence sums the deltas of a line while
              // it produces them (LstmWideArgs::dbias) and k_bias_rows lays the sum over lines into row 0 of the first slab.
              gemm_b16mc(q, GemmOperand16B{y.Sbf.p, ldsb, (long long)N * ndir * ldsb, (long long)N * ldsb},
                         GemmOperand16B{y.Dbf.p, M, (long long)N * M, 4LL * y.no}, StorePartialShift{pbuf.p, R, Cn}, R - 1, Cn, (int)N, ns, ndir, a2,
                         y.sbf_x_external ? y.ni : 0);
              CLSTM_LAUNCH(k_bias_rows, dim3((unsigned)(((size_t)ndir * Cn + 63) / 64)), dim3(64), 0, q, (const float*)y.dbias.p, pbuf.p, bs, ndir, ns, R, Cn);
              g_path_count[13]++;
            } else
            gemm_b16mc(q, GemmOperand16B{y.Sbf.p, ldsb, (long long)N * ndir * ldsb, (long long)N * ldsb},
                       GemmOperand16B{y.Dbf.p, M, (long long)N * M, 4LL * y.no}, StorePartialRot{pbuf.p, R, Cn}, R, Cn, (int)N, ns, ndir, a2,
                       y.sbf_x_external ? y.ni : 0);
            if (y.sbf_x_external) g_path_count[8]++;
          } else if (bf16_gemm)
            gemm_bf16<GEMM_MC, GEMM_MC>(q, gemm_batched(gemm_mc(y.S.p, y.lds, N), (long long)N * y.lds, ndir),
                                        gemm_batched(gemm_mc(y.D.p, M, N, 0), 4LL * y.no, 1),
                                        StorePartial{pbuf.p, R, Cn}, R, Cn, (int)N, ns, ndir);
          else if (x3_big)
            gemm_x3_big<GEMM_MC, GEMM_MC>(q, gemm_batched(gemm_mc(y.S.p, y.lds, N), (long long)N * y.lds, ndir),
                                          gemm_batched(gemm_mc(y.D.p, M, N, 0), 4LL * y.no, 1), StorePartial{pbuf.p, R, Cn}, R, Cn, (int)N, ns, ndir);
          else
            gemm_f32<GEMM_MC, GEMM_MC, StorePartial, GEMM_BK_DW>(q, gemm_batched(gemm_mc(y.S.p, y.lds, N), (long long)N * y.lds, ndir),
                                                                 gemm_batched(gemm_mc(y.D.p, M, N, 0), 4LL * y.no, 1),
                                                                 StorePartial{pbuf.p, R, Cn}, R, Cn, (int)N, ns, ndir);
        }
        const ReduceDesc gates{pbuf.p, y.moff, 0LL, ns, ndir, R, Cn, y.no};
        ReduceDesc extra{};   // empty unless this is the top layer
        if (l == (int)L.size() - 1) extra = sm_red;
        if (L.size() > 1) {   // stacked net: every reduction behind the last recurrence of the pass (see reduce_layer)
          pending_red.push_back(PendingReduce{gates, extra, l});
          timing.end(q);
          check_launch();
          return;
        }
        reduce_layer(gates, extra, l, true, fuse, peer, gdst, q);
        timing.end(q);
        check_launch();
      };
      auto do_dx = [&]() {
        // input deltas: x.d = sum_dir W_x^T delta (Parallel::backward sums both subs, clstm.cc:538-541)
        float* dx = nullptr;
        if (l > 0) dx = L[l - 1].dH.p;
        else if (want_dx0) { dX0.reserve((size_t)N * y.ni); dx = dX0.p; }
        if (!dx || dx_done) return;
        timing.begin("gemm_gates_dx", s);
        if (bf16_gemm && bf16_rec && bwd_persistent && wide_kp16_bwd(y.no) == 4 * y.no && y.Wtb.p && y.Dbf.p)
        {
          // the persistent recurrence left the deltas as a k-contiguous bf16 array: both operands go to LDS as they are
          g_path_count[3]++;
          gemm_b16kk(s, GemmOperand16{y.Dbf.p, M, (long long)N * M}, GemmOperand16{y.Wtb.p, M, (long long)y.ni * M},
                     StorePlain{dx, y.ni}, (int)N, y.ni, M);
        } else if (ensure_delta_f32(l), bf16_gemm)
          gemm_bf16<GEMM_KC, GEMM_KC>(s, gemm_kc(y.D.p, M, N, 32), gemm_kc(y.Wt, M, y.ni, y.wt_slack), StorePlain{dx, y.ni},
                                      (int)N, y.ni, M);
        else if (y.wide && gemm_x3_on && gemm_bf16_big((int)N, y.ni))
          gemm_x3_big<GEMM_KC, GEMM_KC>(s, gemm_kc(y.D.p, M, N, 32), gemm_kc(y.Wt, M, y.ni, y.wt_slack), StorePlain{dx, y.ni}, (int)N, y.ni, M);
        else
          gemm_f32<GEMM_KC, GEMM_KC>(s, gemm_kc(y.D.p, M, N), gemm_kc(y.Wt, M, y.ni, 0), StorePlain{dx, y.ni}, (int)N,
                                     y.ni, M);
        timing.end(s);
        check_launch();
      };
      do_dw(s);
      do_dx();
    }
    if (!pending_red.empty()) {
      timing.begin("reduce_scatter", s);
      for (size_t i = 0; i < pending_red.size(); i++)
        reduce_layer(pending_red[i].gates, pending_red[i].extra, pending_red[i].l, i + 1 == pending_red.size(), fuse, peer, gdst, s);
      pending_red.clear();
      timing.end(s);
      check_launch();
    }
  }
  // The slab reduction of one layer (+ the softmax layer's, riding the top layer's).  `last`: the last reduction launch of the
  // backward pass.  With the update riding the reductions (fuse: train_step without an exchange), ALL of them must see the final
  // error state of the pass -- a sticky device error raised by a lower layer's recurrence after an upper layer had taken its update
  // would leave half a step applied -- so a stacked net keeps a slab buffer per layer (Layer::partial) and reduces every layer here,
  // behind the LAST recurrence; a single layer is reduced where it always was.  (What a reduction itself can still find is a
  // non-finite gradient ENTRY: skipped entry by entry, ops.h:k_update.)
  struct PendingReduce { ReduceDesc gates, extra; int l; };
  std::vector<PendingReduce> pending_red;
  DevBuf<float>& layer_partial(Layer& y) { return L.size() > 1 ? y.partial : partial; }
  void reduce_layer(const ReduceDesc& gates, const ReduceDesc& extra, int l, bool last, bool fuse, bool peer, float* gdst, hipStream_t q) {
    const size_t work = (size_t)gates.nbatch * gates.R * gates.Cn + (size_t)extra.R * extra.Cn * extra.nbatch;
    UpdateFuse uf{};
    uf.nanflag = nanflag(); uf.step_no = step_no();
    if (fuse) {   // (train_step without a communicator) this layer's parameters are updated by the reduction itself
      uf = UpdateFuse{v, d, lr, mom, gclip, (const int*)dev_err_words(), last ? update_step_word : nullptr, update_step_id, nanflag(), step_no(), PackDst{}};
      if (last) { update_step_word = nullptr; update_applied = true; }
      // a single narrow layer whose packed copies are current: the update keeps them current (ops.h: PackDst) and the next
      // step's ingest launch has nothing to repack
      packs_follow_update = false;
      if (L.size() == 1 && !L[0].wide && !packed_dirty && dbg_opt("update_repack", 1) && pack_inverse(L[0])) {
        Layer& y = L[0];
        const size_t nr = (size_t)ndir * 4 * 4 * y.nk4 * y.nthreads;
        uf.pk = PackDst{y.pack_inv.p, PACK_KD, y.Wt, y.bias, y.Rf, y.Rb, y.pd, pack_fused_desc(y), (unsigned)((size_t)(1 + y.ni) * ndir * 4 * y.no), (unsigned)nr};
        packs_follow_update = true;
        g_path_count[10]++;
      }
    }
    CLSTM_LAUNCH(k_reduce_scatter, dim3(nblocks(work)), dim3(256), 0, q, gates, extra, gdst, (int*)nullptr, 0, uf);
    if (peer && last) peer_pending = true;
    (void)l;
  }

  long long nbackward = 0;           // backward passes of this net so far = the number of the current training step (from 1)
  int step_no() const { return (int)std::min<long long>(nbackward, 2147483647LL); }
  // The reference asserts on non-finite values in BACKWARD only (clstm.cc:630-649): a forward pass that belongs to no
  // training step (predict, the test-set pass of clstmocrtrain: clstm_net_set_training(net, 0)) must not arm the process-wide
  // word -- one NaN logit in an inference pass would block the updates of every net of the process.
  bool training = true;
  int* fwd_nanflag() const { return training ? nanflag() : nullptr; }
  static int* nanflag() {            // device error word [3], or null when CLSTM_NANCHECK=0
    static const bool on = !(getenv("CLSTM_NANCHECK") && atoi(getenv("CLSTM_NANCHECK")) == 0);
    return on ? dev_err_words() + 3 : nullptr;
  }
  int* update_step_word = nullptr;   // (host-fed steps) pinned word the update kernel writes update_step_id into
  int update_step_id = 0;
  bool peer_step = false;     // the NEXT backward pass writes its gradient into the communicator's exchange slot and update() runs the
                              //   peer-read all-reduce fused with the update (set by train_step when a communicator of > 1 ranks is attached)
  bool peer_pending = false;  // ... this backward pass did so
  bool fuse_update = false;   // the NEXT backward pass applies the update inside its reductions (set by train_step)
  // The update rides the slab reductions (train_step without an exchange).  All or nothing: every reduction of a pass runs behind
  // its last recurrence (reduce_layer) -- in a stacked net the layers' reductions used to sit between the recurrences, top down, and
  // an error word raised by a lower layer found the upper layers' parameters already updated: half a step.
  bool fuse_eligible() const { return !comm || comm->nranks == 1; }
  bool update_applied = false; // ... and has done so: update() has nothing left to launch
  bool packs_follow_update = false;   // ... and rewrote the packed copies of the parameters it moved
  void update() {
    hipStream_t s = stream();
    RoctxRange range_("clstm:update");
    params_epoch++;
    if (update_applied) {   // done by the reductions of the backward pass just enqueued
      update_applied = false;
      packed_dirty = !packs_follow_update;
      packs_follow_update = false;
      return;
    }
    if (peer_pending && comm) {   // one-shot peer-read all-reduce fused into the update (ops.h: k_peer_barrier / k_peer_allreduce_update)
      peer_pending = false;
      RoctxRange range2_("clstm:allreduce+update");
      const int sq = ++comm->peer.seq;
      const PeerArgs pa = comm->peer.args(sq, comm->rank, comm->nranks);
      timing.begin("allreduce_grads", s);
      comm->peer_barrier(sq, s);   // (the hosts announce the exchange to each other first: Comm::peer_barrier)
      timing.end(s);
      timing.begin("sgd_update", s);
      CLSTM_LAUNCH(k_peer_allreduce_update, dim3(nblocks((size_t)(nparams + 3) / 4)), dim3(256), 0, s, pa, v, d, g, (size_t)nparams, lr, mom, gclip,
                   (const int*)dev_err_words(), update_step_word, update_step_id, nanflag(), step_no());
      update_step_word = nullptr;
      timing.end(s);
      check_launch();
      g_path_count[7]++;
      packed_dirty = true;
      maybe_replica_check(s);
      return;
    }
    if (comm && comm->nranks > 1) {   // sum of the ranks' fresh minibatch gradients, in place, on this stream (share_deltas, clstm.cc:731-744)
      RoctxRange range2_("clstm:allreduce");
      timing.begin("allreduce_grads", s);
      comm->allreduce(g, nparams, s);
      timing.end(s);
    }
    timing.begin("sgd_update", s);
    CLSTM_LAUNCH(k_update, dim3(nblocks(nparams)), dim3(256), 0, s, v, d, (const float*)g, (size_t)nparams, lr, mom, gclip, (const int*)dev_err_words(), update_step_word, update_step_id,
                 nanflag(), step_no());
    update_step_word = nullptr;
    timing.end(s);
    check_launch();
    packed_dirty = true;
    if (comm && comm->nranks > 1) maybe_replica_check(s);
  }
  // Replica consistency (ops.h:k_param_checksum ...): every replica_every-th update of a net whose communicator has several
  // ranks, and on request (clstm_net_replica_check).  Everything stays on the stream; a mismatch lands in device error word [7].
  long long nupdates_dp = 0;
  static int replica_every() {
    static const int n = getenv("CLSTM_REPLICA_CHECK_EVERY") ? atoi(getenv("CLSTM_REPLICA_CHECK_EVERY")) : 256;
    return n;
  }
  void maybe_replica_check(hipStream_t s) {
    nupdates_dp++;
    const int every = replica_every();
    if (every > 0 && nupdates_dp % every == 0) replica_check(s);
  }
  void replica_check(hipStream_t s) {
    if (!comm || comm->nranks < 2) return;
    RoctxRange range_("clstm:replica_check");
    comm->chk.reserve(8);
    comm->chk_acc.reserve(2);
    CLSTM_LAUNCH(k_param_checksum, dim3(std::min<unsigned>(nblocks(nparams), 1024u)), dim3(256), 0, s, (const float*)v, (size_t)nparams, comm->chk_acc.p);
    CLSTM_LAUNCH(k_checksum_pieces, dim3(1), dim3(64), 0, s, comm->chk_acc.p, comm->chk.p);
    comm->allreduce(comm->chk.p + 4, 4, s);
    CLSTM_LAUNCH(k_replica_verify, dim3(1), dim3(64), 0, s, (const float*)comm->chk.p, comm->nranks, dev_err_words() + 7, step_no());
    check_launch();
    g_path_count[12]++;
  }
};

static thread_local long long* g_last_ctc_prof = nullptr;
// CTC on an arbitrary packed batch (used by the net and by the stand-alone ABI entry)
struct CtcWorkspace {
  CtcArgs pending{};            // kernel arguments of a prepared-but-not-launched alignment (train step)
  size_t pending_smem = 0;
  int pending_bs = 0;
  PinnedRing ring;
  DevBuf<long long> prof;
  DevBuf<double> tables;
  DevBuf<char> meta;
  DevBuf<float> lat;
};
// The per-minibatch metadata block [line records | states] is staged in a pinned slot;
// `defer` (non-null): do not enqueue its copy -- the caller folds it into a kernel it launches anyway before the CTC
// kernel (the input-ingest launch of a training step) and receives source, destination and size here.
struct CtcMetaCopy { const int* src = nullptr; int* dst = nullptr; int nwords = 0; };
static void run_ctc(CtcWorkspace& w, const float* probs, float* deltas, float* aligned, int nc,
                    const int* line_off_h, const int* states_h, const int* state_off_h, int bs,
                    hipStream_t s, CtcMetaCopy* defer = nullptr, bool launch = true) {
  REQUIRE(bs > 0, "empty batch");
  std::vector<long long> lo(bs + 1, 0);
  for (int b = 0; b < bs; b++) {
    const long long T = line_off_h[b + 1] - line_off_h[b], S = state_off_h[b + 1] - state_off_h[b];
    REQUIRE(T >= 0 && S >= 0, "bad offsets");
    // three lattices; lines of more than CTC_SMAX_LDS states keep their per-state totals (S doubles) behind them
    lo[b + 1] = lo[b] + 3 * T * S + (S > CTC_SMAX_LDS ? 2 * S + 2 : 0);
    lo[b + 1] += lo[b + 1] & 1;   // (8-byte alignment of those doubles)
  }
  const int ns = state_off_h[bs];
  for (int i = 0; i < ns; i++) REQUIRE(states_h[i] >= 0 && states_h[i] < nc, "target class out of range");
  w.lat.reserve((size_t)(lo[bs] > 0 ? lo[bs] : 1));
  // one pinned slot, one device block, one copy: [line records (bs x 32 bytes, in workgroup order) | states]
  const size_t nln = (size_t)bs * sizeof(CtcLine), nst = (size_t)(ns > 0 ? ns : 1) * sizeof(int);
  w.meta.reserve(nln + nst);
  {
    char* stage = (char*)w.ring.acquire(nln + nst);
    // workgroups take the lines largest lattice first (one workgroup per line; more lines than CUs run in rounds)
    std::vector<int> order(bs);
    for (int b = 0; b < bs; b++) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return lo[x + 1] - lo[x] > lo[y + 1] - lo[y]; });
    CtcLine* ln = (CtcLine*)stage;
    for (int i = 0; i < bs; i++) {
      const int b = order[i];
      ln[i] = CtcLine{lo[b], b, line_off_h[b], line_off_h[b + 1] - line_off_h[b], state_off_h[b], state_off_h[b + 1] - state_off_h[b], 0};
    }
    if (ns > 0) memcpy(stage + nln, states_h, (size_t)ns * sizeof(int));
    if (defer) {
      defer->src = (const int*)stage; defer->dst = (int*)w.meta.p; defer->nwords = (int)((nln + nst) / sizeof(int));
    } else {
      HIPCHECK(hipMemcpyAsync(w.meta.p, stage, nln + nst, hipMemcpyHostToDevice, s));
      w.ring.commit(s);
    }
  }
  CtcArgs a{};
  a.lines = (const CtcLine*)w.meta.p;
  a.P = probs; a.Dz = deltas; a.aligned = aligned;
  a.states = (const int*)(w.meta.p + nln);
  a.lat = w.lat.p; a.nc = nc;
  w.prof.reserve(16); a.prof = w.prof.p; g_last_ctc_prof = w.prof.p;
  a.float_logadd = dbg_opt("ctc_float", 0) != 0;   // experiment option (ctc.h: ctc_softplus_float); read per alignment
  if (!w.tables.p) {
    w.tables.reserve(CTC_TABLE_DOUBLES);
    std::vector<double> tb(CTC_TABLE_DOUBLES);
    for (int i = 0; i < 32; i++) tb[i] = CTC_EXP2_32[i];
    for (int i = 0; i < 64; i++) { tb[32 + i] = CTC_LOG_INVC[i]; tb[96 + i] = CTC_LOG_LOGC[i]; }
    for (int k = 0; k < 2 * CTC_SP_KMAX + 1; k++)
      for (int c = 0; c < 4; c++) tb[160 + 4 * k + c] = CTC_SOFTPLUS[k][c];
    HIPCHECK(hipMemcpy(w.tables.p, tb.data(), CTC_TABLE_DOUBLES * sizeof(double), hipMemcpyHostToDevice));
  }
  a.tables = w.tables.p;
  int smax = 1, tmax = 1;
  for (int b = 0; b < bs; b++) {
    smax = std::max(smax, state_off_h[b + 1] - state_off_h[b]);
    tmax = std::max(tmax, line_off_h[b + 1] - line_off_h[b]);
  }
  if (smax > CTC_SMAX_LDS) smax = CTC_SMAX_LDS;     // (longer lines do not use the per-state LDS arrays)
  a.smax = smax;
  a.ncp = nc | 1;                                   // odd row stride: conflict-free row-per-lane access
  // frames per LDS tile: what the 160 KiB carve leaves after the tables and the per-state vectors
  const long fixed = (long)ctc_lds_layout(0, a.ncp, smax).words * (long)sizeof(float);
  int tile = (int)((160 * 1024 - fixed) / (long)((a.ncp + (smax | 1) + 1) * sizeof(float)));
  if (tile > CTC_MAX_TILE) tile = CTC_MAX_TILE;
  if (tile > tmax) tile = tmax;
  REQUIRE(tile >= 1, "too many classes / target states for the CTC row tile");
  a.tile = tile;
  const size_t smem = (size_t)ctc_lds_layout(a.tile, a.ncp, a.smax).words * sizeof(float);
  REQUIRE(smem <= 160 * 1024, "CTC LDS carve exceeds 160 KiB");
#ifndef CLSTM_HIP_EMU
  static size_t smem_set = 0;
  if (smem > smem_set) {
    HIPCHECK(hipFuncSetAttribute((const void*)ctc_align_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    HIPCHECK(hipFuncSetAttribute((const void*)ctc_align_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    smem_set = smem;
  }
#endif
  w.pending = a; w.pending_smem = smem; w.pending_bs = bs;
  if (launch) {
    if (a.float_logadd) CLSTM_LAUNCH(ctc_align_kernel<true>, dim3(bs), dim3(CTC_THREADS), smem, s, a);
    else CLSTM_LAUNCH(ctc_align_kernel<false>, dim3(bs), dim3(CTC_THREADS), smem, s, a);
    check_launch();
  }
}
struct DecodeWorkspace {
  DevBuf<int> line_off, idx, cls, loc, cnt;
  DevBuf<float> val;
};
static void run_decode(DecodeWorkspace& w, const float* probs, int nc, const int* line_off_h, int bs,
                       int* classes_h, int* locs_h, int* counts_h, hipStream_t s) {
  const int N = line_off_h[bs];
  REQUIRE(bs > 0 && N > 0, "empty batch");
  w.line_off.reserve(bs + 1); w.idx.reserve(N); w.val.reserve(N); w.cls.reserve(N); w.loc.reserve(N); w.cnt.reserve(bs);
  HIPCHECK(hipMemcpyAsync(w.line_off.p, line_off_h, (bs + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  CLSTM_LAUNCH(argmax_kernel, dim3((N + 255) / 256), dim3(256), 0, s, probs, w.idx.p, w.val.p, N, nc);
  CLSTM_LAUNCH(decode_kernel, dim3(bs), dim3(64), 0, s, (const int*)w.idx.p, (const float*)w.val.p,
               (const int*)w.line_off.p, w.cls.p, w.loc.p, w.cnt.p);
  check_launch();
  HIPCHECK(hipMemcpyAsync(counts_h, w.cnt.p, bs * sizeof(int), hipMemcpyDeviceToHost, s));
  if (classes_h) HIPCHECK(hipMemcpyAsync(classes_h, w.cls.p, N * sizeof(int), hipMemcpyDeviceToHost, s));
  if (locs_h) HIPCHECK(hipMemcpyAsync(locs_h, w.loc.p, N * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
}

static thread_local CtcWorkspace* g_ctc_ws = nullptr;
static thread_local DecodeWorkspace* g_dec_ws = nullptr;

}  // namespace clstm

using namespace clstm;
struct clstm_net {
  Net net;
  CtcWorkspace ctc;
  DecodeWorkspace dec;
};

#define EW(kernel, len, ...) \
  CLSTM_LAUNCH(kernel, dim3(nblocks(len)), dim3(256), 0, g_stream, __VA_ARGS__); check_launch();

extern "C" {

const char* clstm_last_error(void) { return g_err.c_str(); }
int clstm_abi_version(void) { return 1; }
int clstm_set_stream(void* s) { g_stream = (hipStream_t)s; return 0; }
int clstm_synchronize(void) { ABI_BEGIN HIPCHECK(hipStreamSynchronize(g_stream)); check_device_errors(); ABI_END }

// ---- per-op entry points -------------------------------------------------------------------------
int clstm_forward_nonlin0(float* y, int len, int nl) { ABI_BEGIN EW(k_forward_nonlin0, len, y, (size_t)len, nl) ABI_END }
int clstm_backward_nonlin0(const float* yv, float* yd, int len, int nl) { ABI_BEGIN EW(k_backward_nonlin0, len, yv, yd, (size_t)len, nl) ABI_END }
int clstm_forward_nonlin(float* y, const float* x, int len, int nl) { ABI_BEGIN EW(k_forward_nonlin, len, y, x, (size_t)len, nl) ABI_END }
int clstm_backward_nonlin(const float* yv, const float* yd, float* xd, int len, int nl) { ABI_BEGIN EW(k_backward_nonlin, len, yv, yd, xd, (size_t)len, nl) ABI_END }
int clstm_forward_lin1(float* y, const float* W, const float* x, int n, int m, int bs) {
  ABI_BEGIN EW(k_forward_lin1, (size_t)n * bs, y, W, x, n, m, bs, -1) ABI_END
}
int clstm_backward_lin1(const float* yd, const float* W, float* Wd, const float* x, float* xd, int n, int m, int bs) {
  ABI_BEGIN
  EW(k_backward_lin1_dx, (size_t)(m - 1) * bs, yd, W, xd, n, m, bs, 0)
  EW(k_backward_lin1_dw, (size_t)n * m, yd, Wd, x, n, m, bs)
  ABI_END
}
int clstm_forward_full1(float* y, const float* W, const float* x, int n, int m, int bs, int nl) {
  ABI_BEGIN
  REQUIRE(nl >= 0 && nl <= 4, "bad nonlinearity code (clstm_compute.cc:147 aborts)");
  EW(k_forward_lin1, (size_t)n * bs, y, W, x, n, m, bs, nl)
  ABI_END
}
int clstm_backward_full1(const float* yv, float* yd, const float* W, float* Wd, const float* x, float* xd,
                         int n, int m, int bs, int nl) {
  ABI_BEGIN
  REQUIRE(nl >= 0 && nl <= 4, "bad nonlinearity code");
  EW(k_backward_nonlin0, (size_t)n * bs, yv, yd, (size_t)n * bs, nl)
  EW(k_backward_lin1_dx, (size_t)(m - 1) * bs, (const float*)yd, W, xd, n, m, bs, 0)
  EW(k_backward_lin1_dw, (size_t)n * m, (const float*)yd, Wd, x, n, m, bs)
  ABI_END
}
int clstm_forward_softmax(float* z, const float* W, const float* x, int n, int m, int bs) {
  ABI_BEGIN
  REQUIRE(n >= 2, "Softmax requires n>=2 (clstm_compute.cc:328)");
  EW(k_forward_lin1, (size_t)n * bs, z, W, x, n, m, bs, -1)
  CLSTM_LAUNCH(k_softmax_norm, dim3((bs + 3) / 4), dim3(256), 0, g_stream, z, n, (size_t)bs, (int*)nullptr, 0);
  check_launch();
  ABI_END
}
int clstm_backward_softmax(const float* zd, const float* W, float* Wd, const float* x, float* xd, int n, int m, int bs) {
  ABI_BEGIN
  EW(k_backward_lin1_dx, (size_t)(m - 1) * bs, zd, W, xd, n, m, bs, 1)
  EW(k_backward_lin1_dw, (size_t)n * m, zd, Wd, x, n, m, bs)
  ABI_END
}
int clstm_forward_stack(float* z, const float* x, const float* y, int nx, int ny, int bs) {
  ABI_BEGIN REQUIRE(y != nullptr, "null operand"); EW(k_stack, (size_t)(nx + ny) * bs, z, x, y, nx, ny, bs) ABI_END
}
int clstm_backward_stack(const float* zd, float* xd, float* yd, int nx, int ny, int bs) {
  ABI_BEGIN REQUIRE(yd != nullptr, "null operand"); EW(k_unstack_add, (size_t)(nx + ny) * bs, zd, xd, yd, nx, ny, bs) ABI_END
}
int clstm_forward_stack_delay(float* z, const float* x, const float* ylast, int nx, int ny, int bs) {
  ABI_BEGIN EW(k_stack, (size_t)(nx + ny) * bs, z, x, ylast, nx, ny, bs) ABI_END
}
int clstm_backward_stack_delay(const float* zd, float* xd, float* ylastd, int nx, int ny, int bs) {
  ABI_BEGIN EW(k_unstack_add, (size_t)(nx + ny) * bs, zd, xd, ylastd, nx, ny, bs) ABI_END
}
int clstm_forward_reverse(float* y, const float* x, int rows, int bs, int N) {
  ABI_BEGIN EW(k_forward_reverse, (size_t)rows * bs * 2 * N, y, x, (size_t)rows * bs * 2, N) ABI_END
}
int clstm_backward_reverse(const float* y, float* x, int rows, int bs, int N) {
  ABI_BEGIN EW(k_backward_reverse, (size_t)rows * bs * N, y, x, (size_t)rows * bs, N) ABI_END
}
int clstm_forward_btswitch(float* y, const float* x, int rows, int bs, int N) {
  ABI_BEGIN EW(k_forward_btswitch, (size_t)rows * bs * N, y, x, rows, bs, N) ABI_END
}
int clstm_backward_btswitch(const float* y, float* x, int rows, int bs, int N) {
  ABI_BEGIN EW(k_backward_btswitch, (size_t)rows * bs * N, y, x, rows, bs, N) ABI_END
}
int clstm_forward_batchstack(float* y, const float* x, int d, int bs, int N, int pre, int post) {
  ABI_BEGIN
  REQUIRE(pre >= 0 && post >= 0, "batchstack: negative pre/post");
  EW(k_forward_batchstack, (size_t)(pre + post + 1) * d * bs * 2 * N, y, x, d, bs, N, pre, post)
  ABI_END
}
int clstm_backward_batchstack(const float* y, float* x, int d, int bs, int N, int pre, int post) {
  ABI_BEGIN
  REQUIRE(pre >= 0 && post >= 0, "batchstack: negative pre/post");
  EW(k_backward_batchstack, (size_t)d * bs * N, y, x, d, bs, N, pre, post)
  ABI_END
}
int clstm_forward_statemem(float* st, const float* ci, const float* gi, const float* last, const float* gf, int len) {
  ABI_BEGIN EW(k_forward_statemem, len, st, ci, gi, last, gf, (size_t)len) ABI_END
}
int clstm_backward_statemem(const float* sd, const float* ci, float* cid, const float* gi, float* gid,
                            const float* last, float* lastd, const float* gf, float* gfd, int len) {
  ABI_BEGIN EW(k_backward_statemem, len, sd, ci, cid, gi, gid, last, lastd, gf, gfd, (size_t)len) ABI_END
}
int clstm_forward_nonlingate(float* out, const float* st, const float* go, int len, int nl) {
  ABI_BEGIN EW(k_forward_nonlingate, len, out, st, go, (size_t)len, nl) ABI_END
}
int clstm_backward_nonlingate(const float* outd, const float* st, float* std_, const float* go, float* god, int len, int nl) {
  ABI_BEGIN EW(k_backward_nonlingate, len, outd, st, std_, go, god, (size_t)len, nl) ABI_END
}
int clstm_clip_gradient(float* d, int len, float clip) {
  ABI_BEGIN
  if (clip >= 1e6f) return 0;  // clstm_compute.cc:554
  REQUIRE(clip > 0, "clip must be positive (clstm_compute.cc:555)");
  EW(k_clip, len, d, (size_t)len, clip)
  ABI_END
}
int clstm_sgd_update(float* v, float* d, int len, float lr, float mom) { ABI_BEGIN EW(k_sgd, len, v, d, (size_t)len, lr, mom) ABI_END }

// ---- CTC ------------------------------------------------------------------------------------------
int clstm_mktargets(int* states_h, const int* transcript_h, int L) {
  for (int t = 0; t < 2 * L + 1; t++) states_h[t] = (t % 2 == 1) ? transcript_h[(t - 1) / 2] : 0;
  return 0;
}
int clstm_ctc_align_batch(const float* probs, float* deltas, float* aligned, int nc, const int* line_off_h,
                          const int* states_h, const int* state_off_h, int bs) {
  ABI_BEGIN
  if (!g_ctc_ws) g_ctc_ws = new CtcWorkspace();
  run_ctc(*g_ctc_ws, probs, deltas, aligned, nc, line_off_h, states_h, state_off_h, bs, g_stream);
  ABI_END
}
int clstm_trivial_decode_batch(const float* probs, int nc, const int* line_off_h, int bs, int* classes_h,
                               int* locs_h, int* counts_h) {
  ABI_BEGIN
  if (!g_dec_ws) g_dec_ws = new DecodeWorkspace();
  run_decode(*g_dec_ws, probs, nc, line_off_h, bs, classes_h, locs_h, counts_h, g_stream);
  ABI_END
}

// ---- fused network ----------------------------------------------------------------------------------
int clstm_net_nparams_for(const clstm_net_desc* ds) {
  long long total = 0;
  int ni = ds->ninput;
  const int nd = ds->unidirectional ? 1 : 2;
  for (int l = 0; l < ds->nlayers; l++) {
    const int no = ds->nhidden[l];
    total += (long long)nd * 4 * no * (ni + no + 1);
    ni = nd * no;
  }
  total += (long long)ds->nclasses * (ni + 1);
  return (int)total;
}
int clstm_net_create(clstm_net** out, const clstm_net_desc* ds, float* pv, float* pd, float* pg) {
  ABI_BEGIN
  REQUIRE(out && ds, "null argument");
  clstm_net* h = new clstm_net();
  try { h->net.build(*ds, pv, pd, pg); } catch (...) { delete h; throw; }
  *out = h;
  ABI_END
}
int clstm_net_destroy(clstm_net* h) { ABI_BEGIN delete h; ABI_END }
int clstm_net_nparams(clstm_net* h) { return h->net.nparams; }
int clstm_net_buffers(clstm_net* h, float** v, float** d, float** g) {
  if (v) *v = h->net.v;
  if (d) *d = h->net.d;
  if (g) *g = h->net.g;
  return 0;
}
static void copy_h2d(float* dst, const float* src, size_t n) {
  HIPCHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyHostToDevice, g_stream));
  HIPCHECK(hipStreamSynchronize(g_stream));
}
static void copy_d2h(float* dst, const float* src, size_t n) {
  HIPCHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToHost, g_stream));
  HIPCHECK(hipStreamSynchronize(g_stream));
  check_device_errors();   // whatever is read back was produced by launches whose outcome is known now
}
int clstm_net_set_params_h(clstm_net* h, const float* p) { ABI_BEGIN copy_h2d(h->net.v, p, h->net.nparams); h->net.packed_dirty = true; h->net.params_epoch++; ABI_END }
int clstm_net_get_params_h(clstm_net* h, float* p) { ABI_BEGIN copy_d2h(p, h->net.v, h->net.nparams); ABI_END }
int clstm_net_set_derivs_h(clstm_net* h, const float* p) { ABI_BEGIN copy_h2d(h->net.d, p, h->net.nparams); ABI_END }
int clstm_net_get_derivs_h(clstm_net* h, float* p) { ABI_BEGIN copy_d2h(p, h->net.d, h->net.nparams); ABI_END }
int clstm_net_get_grads_h(clstm_net* h, float* p) { ABI_BEGIN copy_d2h(p, h->net.g, h->net.nparams); ABI_END }
int clstm_net_params_changed(clstm_net* h) { h->net.packed_dirty = true; h->net.params_epoch++; return 0; }
int clstm_net_set_learning_rate(clstm_net* h, float lr, float mom) { h->net.lr = lr; h->net.mom = mom; return 0; }
int clstm_net_set_gradient_clip(clstm_net* h, float c) {
  ABI_BEGIN REQUIRE(c > 0, "clip must be positive"); h->net.gclip = c; ABI_END
}
int clstm_net_set_batch(clstm_net* h, const int* T_h, int bs) { ABI_BEGIN h->net.set_batch(T_h, bs); ABI_END }
int clstm_net_set_inputs_h(clstm_net* h, const float* x) {
  ABI_BEGIN
  REQUIRE(h->net.N > 0, "set_batch first");
  copy_h2d(h->net.X.p, x, (size_t)h->net.N * h->net.desc.ninput);
  h->net.src0_ready = false;
  ABI_END
}
static void net_set_inputs_d(clstm_net* h, const float* x, const CtcMetaCopy* aux = nullptr) {
  Net& n = h->net;
  REQUIRE(n.N > 0, "set_batch first");
  RoctxRange range_("clstm:ingest");
  Layer& y = n.L[0];
  bool aux_done = false;
  const bool with_pack = n.packed_dirty && n.L.size() == 1 && !y.wide;   // the training step of a narrow net: ingest + weight repack in one launch
  if (with_pack || n.lo_pending || (aux && aux->nwords > 0)) {
    // (any net: the small host arrays of the step ride the ingest launch -- a separate copy of the CTC metadata cost a
    // configs[4] step ~25 us of DMA set-up in front of its first kernel)
    const int M = n.ndir * 4 * y.no, KQP = 4 * y.nk4;
    const size_t nr = (size_t)n.ndir * 4 * KQP * y.nthreads;
    // (the ingest blocks' 16-byte form -- ops.h:k_ingest_pack -- has one item per four input floats: launching a thread per float
    //  there dispatched 1,800 workgroups that found nothing to do)
    const bool vec16 = (y.ni & 3) == 0 && (y.lds & 3) == 0 && ((size_t)x & 15) == 0;
    const int nbi = nblocks(vec16 ? (size_t)n.N * (y.ni / 4 + 1) : (size_t)n.N * (1 + y.ni)), nbp = with_pack ? nblocks((size_t)(1 + y.ni) * M + 2 * nr) : 0;
    const bool lo = n.lo_pending, ax = aux && aux->nwords > 0;
    // optional trailing blocks read small host arrays straight from their pinned slots: the line offsets and -- in a
    // training step -- the CTC metadata (no DMA launches, no event records on the stream's critical path)
    CLSTM_LAUNCH(k_ingest_pack, dim3(nbi + nbp + (lo ? 1 : 0) + (ax ? (aux->nwords + 255) / 256 : 0)), dim3(256), 0, g_stream, x, n.X.p, y.S.p, (size_t)n.N, y.ni, y.lds,
                 n.ndir, (long long)n.N * y.lds, nbi, nbp, (const float*)n.v, y.Wt, y.bias, y.Rf, y.Rb, y.pd, n.pack_fused_desc(y),
                 lo ? n.lo_stage : nullptr, n.line_off.p, 2 * n.bs + 1, ax ? aux->src : nullptr, ax ? aux->dst : nullptr, ax ? aux->nwords : 0,
                 with_pack ? (const int*)n.pack_table(y) : (const int*)nullptr);
    if (lo) { n.ring.commit(g_stream); n.lo_pending = false; }
    if (ax) { h->ctc.ring.commit(g_stream); aux_done = true; }
    if (with_pack) n.packed_dirty = false;
  } else {
    CLSTM_LAUNCH(k_ingest, dim3(nblocks((size_t)n.N * (1 + y.ni))), dim3(256), 0, g_stream, x, n.X.p, y.S.p, (size_t)n.N, y.ni,
                 y.lds, n.ndir, (long long)n.N * y.lds);
  }
  check_launch();
  if (aux && aux->nwords > 0 && !aux_done) {   // not the fused launch: a plain asynchronous copy
    HIPCHECK(hipMemcpyAsync(aux->dst, aux->src, (size_t)aux->nwords * sizeof(int), hipMemcpyHostToDevice, g_stream));
    h->ctc.ring.commit(g_stream);
  }
  n.src0_ready = true;
}
int clstm_net_set_inputs_d(clstm_net* h, const float* x) {
  ABI_BEGIN
  net_set_inputs_d(h, x);
  ABI_END
}
int clstm_net_forward(clstm_net* h) { ABI_BEGIN h->net.forward(); ABI_END }
int clstm_net_outputs(clstm_net* h, float** p, float** d) {
  if (p) *p = h->net.Z.p;
  if (d) *d = h->net.Dz.p;
  return 0;
}
int clstm_net_get_outputs_h(clstm_net* h, float* p) { ABI_BEGIN copy_d2h(p, h->net.Z.p, (size_t)h->net.N * h->net.desc.nclasses); ABI_END }
int clstm_net_set_output_deltas_h(clstm_net* h, const float* p) { ABI_BEGIN copy_h2d(h->net.Dz.p, p, (size_t)h->net.N * h->net.desc.nclasses); ABI_END }
static void net_ctc(clstm_net* h, const int* labels_h, const int* L_h, float* aligned_h, CtcMetaCopy* defer = nullptr,
                    bool launch = true) {
  Net& n = h->net;
  REQUIRE(n.N > 0, "set_batch first");
  std::vector<int> soff(n.bs + 1, 0), states;
  int lpos = 0;
  for (int b = 0; b < n.bs; b++) {
    const int L = L_h[b];
    REQUIRE(L >= 0, "negative transcript length");
    states.resize(soff[b] + 2 * L + 1);
    clstm_mktargets(states.data() + soff[b], labels_h + lpos, L);
    for (int i = 0; i < L; i++) REQUIRE(labels_h[lpos + i] != 0, "transcript contains the blank class (Codec::encode asserts c != 0, clstm.cc:232)");
    lpos += L;
    soff[b + 1] = soff[b] + 2 * L + 1;
  }
  float* al = nullptr;
  if (aligned_h) { n.aligned.reserve((size_t)n.N * n.desc.nclasses); al = n.aligned.p; }
  RoctxRange range_(launch ? "clstm:ctc" : "clstm:ctc_prepare");
  if (launch) n.timing.begin("ctc_align", g_stream);
  run_ctc(h->ctc, n.Z.p, n.Dz.p, al, n.desc.nclasses, n.line_off_h.data(), states.data(), soff.data(), n.bs, g_stream, defer, launch);
  if (launch) n.timing.end(g_stream);
  if (aligned_h && launch) copy_d2h(aligned_h, al, (size_t)n.N * n.desc.nclasses);
}
// the alignment prepared by net_ctc(..., launch = false)
static void net_ctc_launch(clstm_net* h) {
  Net& n = h->net;
  RoctxRange range_("clstm:ctc");
  n.timing.begin("ctc_align", g_stream);
  if (h->ctc.pending.float_logadd) CLSTM_LAUNCH(ctc_align_kernel<true>, dim3(h->ctc.pending_bs), dim3(CTC_THREADS), h->ctc.pending_smem, g_stream, h->ctc.pending);
  else CLSTM_LAUNCH(ctc_align_kernel<false>, dim3(h->ctc.pending_bs), dim3(CTC_THREADS), h->ctc.pending_smem, g_stream, h->ctc.pending);
  check_launch();
  n.timing.end(g_stream);
}
int clstm_net_ctc(clstm_net* h, const int* labels_h, const int* L_h, float* aligned_h) {
  ABI_BEGIN
  net_ctc(h, labels_h, L_h, aligned_h);
  ABI_END
}
int clstm_net_backward(clstm_net* h) { ABI_BEGIN h->net.fuse_update = false; h->net.peer_step = false; h->net.backward(); ABI_END }
int clstm_net_enable_input_deltas(clstm_net* h, int on) { h->net.want_dx0 = on != 0; return 0; }
int clstm_net_get_input_deltas_h(clstm_net* h, float* dx) {
  ABI_BEGIN
  REQUIRE(h->net.want_dx0 && h->net.dX0.p, "input deltas not enabled / no backward yet");
  copy_d2h(dx, h->net.dX0.p, (size_t)h->net.N * h->net.desc.ninput);
  ABI_END
}
int clstm_net_update(clstm_net* h) { ABI_BEGIN h->net.update(); ABI_END }
int clstm_net_decode(clstm_net* h, int* cls, int* locs, int* cnt) {
  ABI_BEGIN
  Net& n = h->net;
  REQUIRE(n.N > 0, "set_batch first");
  run_decode(h->dec, n.Z.p, n.desc.nclasses, n.line_off_h.data(), n.bs, cls, locs, cnt, g_stream);
  ABI_END
}
int clstm_net_get_state_h(clstm_net* h, int layer, int dir, int which, float* out) {
  ABI_BEGIN
  Net& n = h->net;
  REQUIRE(layer >= 0 && layer < (int)n.L.size() && dir >= 0 && dir < n.ndir && which >= 0 && which <= 9, "bad state selector");
  Layer& y = n.L[layer];
  n.tmp.reserve((size_t)n.N * y.no);
  if (which >= 6) n.ensure_delta_f32(layer);   // (a persistent bf16 backward pass leaves the deltas as bf16 only)
  if (which == 5) n.ensure_h_f32(layer);       // (... forward pass of a lower layer: the outputs as bf16 only)
  const float* src = which < 4 ? y.G.p : which == 4 ? y.C.p : y.D.p;
  const int slot = which < 4 ? which : which >= 6 ? which - 6 : -1;
  if (which == 5)
    CLSTM_LAUNCH(k_gather_rows, dim3(nblocks((size_t)n.N * y.no)), dim3(256), 0, g_stream, (const float*)(y.hrow() + dir * y.no),
                 n.tmp.p, (size_t)n.N, y.no, y.ldh);
  else
    CLSTM_LAUNCH(k_gather_state, dim3(nblocks((size_t)n.N * y.no)), dim3(256), 0, g_stream, src, n.tmp.p, (size_t)n.N, y.no, n.ndir, dir, slot);
  check_launch();
  copy_d2h(out, n.tmp.p, (size_t)n.N * y.no);
  ABI_END
}
int clstm_net_set_gemm_precision(clstm_net* h, int mode) {
  ABI_BEGIN
  REQUIRE(mode >= 0 && mode <= 2, "precision: 0 = f32 (exact), 1 = bf16 inputs / f32 accumulation in the hoisted GEMMs, 2 = 1 + bf16 MFMA operands in the lock-step recurrence");
  Net& n = h->net;
  const bool rec = mode == 2;
  if (rec != n.bf16_rec) n.packed_dirty = true;   // the other weight packing is needed
  n.bf16_gemm = mode >= 1;
  n.bf16_rec = rec;
  if (n.N > 0 && rec)   // a batch is already declared: make room for the bf16 operand copies
    for (auto& y : n.L)
      if (y.wide) {
        y.Hb.reserve((size_t)std::max<long long>(n.N, 4LL * n.bs + 32) * n.ndir * wide_kp16_fwd(y.no) + 64);
        y.Db.reserve((size_t)std::max<long long>(n.N, 2LL * n.bs + 32) * n.ndir * wide_kp16_bwd(y.no) + 64);
      }
  ABI_END
}
int clstm_net_enable_timing(clstm_net* h, int on) { h->net.timing.on = on != 0; return 0; }
int clstm_net_kernel_time_ms(clstm_net* h, const char* name, double* total_ms, int* launches) {
  ABI_BEGIN
  h->net.timing.collect(g_stream);
  auto it = h->net.timing.acc.find(name);
  if (it == h->net.timing.acc.end()) { *total_ms = 0; *launches = 0; }
  else { *total_ms = it->second.first; *launches = it->second.second; }
  ABI_END
}
int clstm_net_reset_timing(clstm_net* h) { ABI_BEGIN h->net.timing.collect(g_stream); h->net.timing.acc.clear(); ABI_END }

// ---- one-call training step ---------------------------------------------------------------------------
int clstm_net_train_step(clstm_net* h, const int* T_h, int bs, const float* x_d, const int* labels_h, const int* L_h) {
  ABI_BEGIN
  REQUIRE(h && T_h && x_d && labels_h && L_h, "null argument");
  h->net.set_batch(T_h, bs);
  CtcMetaCopy meta;
  net_ctc(h, labels_h, L_h, nullptr, &meta, false);   // host half of the alignment first: its metadata rides the ingest launch
  net_set_inputs_d(h, x_d, &meta);
  h->net.forward();
  net_ctc_launch(h);
  // no exchange in between (no communicator, or one of a single rank): the reductions of the backward pass apply the update
  // themselves; a communicator of several ranks: the peer-read all-reduce fused into the update where the ranks could map each
  // other (else ncclAllReduce + k_update)
  h->net.fuse_update = h->net.fuse_eligible();
  h->net.peer_step = h->net.comm && h->net.comm->nranks > 1 && h->net.comm->peer_ready((size_t)h->net.nparams, g_stream);
  h->net.backward();
  h->net.update();   // all-reduces the fresh gradient first when a communicator is attached
  ABI_END
}

// host frames in, no host synchronisation: see Net::HostFeed
int clstm_net_train_step_h(clstm_net* h, const int* T_h, int bs, const float* x_h, const int* labels_h, const int* L_h) {
  ABI_BEGIN
  REQUIRE(h && T_h && x_h && labels_h && L_h && bs > 0, "null argument");
  Net& n = h->net;
  Net::HostFeed& f = n.hf;
  long long N = 0;
  for (int b = 0; b < bs; b++) { REQUIRE(T_h[b] >= 0, "negative line length"); N += T_h[b]; }
  REQUIRE(N > 0, "batch has no frames");
  const size_t bytes = (size_t)N * n.desc.ninput * sizeof(float);
  if (!f.ready) {
    f.ready = true;
    HIPCHECK(hipStreamCreateWithFlags(&f.cs, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) HIPCHECK(hipEventCreateWithFlags(&f.copied[i], hipEventDisableTiming));
    HIPCHECK(hipHostMalloc((void**)&f.step_done, 64));
    *f.step_done = 0;
  }
  // This step's number (1, 2, ...) is COMMITTED only when its last kernel -- the one that publishes it in the pinned word --
  // has been enqueued: a call that fails on the way (a bad label, a launch error) leaves f.steps where it was, so the next
  // call reuses number and slot and never waits for a step that was not enqueued (two failed calls in a row used to spin
  // here for ever).
  const long long k = f.steps + 1;
  const int slot = (int)(k & 1);
  // the slot was read by step k - 2: its update kernel has run when the pinned word says so (the host is at most a few
  // steps ahead of the GPU, so this rarely waits).  Bounded: after ~2 s of yielding the stream is drained instead --
  // everything enqueued has then run, whatever the word says.
  {
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (k > 2 && (int)((unsigned)(k - 2) - (unsigned)__atomic_load_n(f.step_done, __ATOMIC_ACQUIRE)) > 0) {   // (wrap-safe)
      sched_yield();
      if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
        HIPCHECK(hipStreamSynchronize(g_stream));
        check_device_errors();
        break;
      }
    }
  }
  f.xin[slot].reserve((size_t)N * n.desc.ninput + 64);
  const void* src = x_h;
#ifndef CLSTM_HIP_EMU
  hipPointerAttribute_t attr;
  const bool pinned = hipPointerGetAttributes(&attr, x_h) == hipSuccess && attr.type == hipMemoryTypeHost;
  if (!pinned) {   // pageable memory: one host copy into the slot's pinned staging buffer, then the same DMA
    (void)hipGetLastError();
    if (f.pin_cap[slot] < bytes) {
      if (f.pin[slot]) HIPCHECK(hipHostFree(f.pin[slot]));
      f.pin_cap[slot] = bytes + bytes / 4;
      HIPCHECK(hipHostMalloc(&f.pin[slot], f.pin_cap[slot]));
    }
    memcpy(f.pin[slot], x_h, bytes);
    src = f.pin[slot];
  }
#endif
  HIPCHECK(hipMemcpyAsync(f.xin[slot].p, src, bytes, hipMemcpyHostToDevice, f.cs));
  HIPCHECK(hipEventRecord(f.copied[slot], f.cs));
  HIPCHECK(hipStreamWaitEvent(g_stream, f.copied[slot], 0));
  n.set_batch(T_h, bs);
  CtcMetaCopy meta;
  net_ctc(h, labels_h, L_h, nullptr, &meta, false);
  net_set_inputs_d(h, f.xin[slot].p, &meta);
  n.forward();
  net_ctc_launch(h);
  n.update_step_word = f.step_done; n.update_step_id = (int)(unsigned)k;
  n.fuse_update = n.fuse_eligible();
  n.peer_step = n.comm && n.comm->nranks > 1 && n.comm->peer_ready((size_t)n.nparams, g_stream);
  try {
    n.backward();
    n.update();
  } catch (...) { n.update_step_word = nullptr; throw; }
  REQUIRE(n.update_step_word == nullptr, "internal: no kernel of the step took the step word");
  f.steps = k;
  ABI_END
}
int clstm_host_alloc(void** p, size_t bytes) { ABI_BEGIN REQUIRE(p, "null argument"); HIPCHECK(hipHostMalloc(p, bytes ? bytes : 1)); ABI_END }
int clstm_host_free(void* p) { ABI_BEGIN if (p) HIPCHECK(hipHostFree(p)); ABI_END }

// ---- state externalisation (clstm.cc:762-811) -----------------------------------------------------------
namespace clstm {
// One entry per Sequence that walk_states(net, f, "", true) visits (clstm.cc:63-70): every node's inputs and
// outputs, then its registered states in std::map order (NPLSTM: ci, gf, gi, go, source, state --
// ENROLL(gi, gf, go, ci, state, source), clstm.cc:560), then its subs in order.
struct StateEntry {
  enum Kind { X, Z, LIN, HBOTH, HDIR, GATE, CELL, SOURCE } kind;
  int layer, dir, slot, rows;
  bool rev;   // stored in the time order of the NPLSTM inside Reversed (frame T-1-t at step t)
};
static std::vector<StateEntry> state_walk(const Net& n) {
  std::vector<StateEntry> w;
  const int nl = (int)n.L.size();
  auto lin_rows = [&](int l) { return l == 0 ? n.desc.ninput : n.ndir * n.L[l - 1].no; };
  auto nplstm = [&](int l, int dir, bool rev) {
    const int no = n.L[l].no;
    w.push_back({StateEntry::LIN, l, dir, 0, lin_rows(l), rev});            // inputs
    w.push_back({StateEntry::HDIR, l, dir, 0, no, rev});                    // outputs
    w.push_back({StateEntry::GATE, l, dir, 3, no, rev});                    // ci
    w.push_back({StateEntry::GATE, l, dir, 1, no, rev});                    // gf
    w.push_back({StateEntry::GATE, l, dir, 0, no, rev});                    // gi
    w.push_back({StateEntry::GATE, l, dir, 2, no, rev});                    // go
    w.push_back({StateEntry::SOURCE, l, dir, 0, lin_rows(l) + no, rev});    // source = [x_t ; h_{t-1}]
    w.push_back({StateEntry::CELL, l, dir, 0, no, rev});                    // state
  };
  w.push_back({StateEntry::X, 0, 0, 0, n.desc.ninput, false});              // Stacked.inputs
  w.push_back({StateEntry::Z, 0, 0, 0, n.desc.nclasses, false});            // Stacked.outputs
  for (int l = 0; l < nl; l++) {
    if (n.ndir == 2) {
      w.push_back({StateEntry::LIN, l, 0, 0, lin_rows(l), false});          // Parallel.inputs
      w.push_back({StateEntry::HBOTH, l, 0, 0, 2 * n.L[l].no, false});      // Parallel.outputs
      nplstm(l, 0, false);
      w.push_back({StateEntry::LIN, l, 1, 0, lin_rows(l), false});          // Reversed.inputs
      w.push_back({StateEntry::HDIR, l, 1, 0, n.L[l].no, false});           // Reversed.outputs
      nplstm(l, 1, true);
    } else {
      nplstm(l, 0, false);
    }
  }
  w.push_back({StateEntry::LIN, nl, 0, 0, lin_rows(nl), false});            // SoftmaxLayer.inputs
  w.push_back({StateEntry::Z, 0, 0, 0, n.desc.nclasses, false});            // SoftmaxLayer.outputs
  return w;
}
// a reference Sequence is rectangular (size x rows x cols): every line of the batch must have the same length
static int states_T(const Net& n) {
  REQUIRE(n.N > 0, "no batch: run forward() (or set_states) first");
  const int T = n.line_off_h[1] - n.line_off_h[0];
  for (int b = 0; b < n.bs; b++)
    REQUIRE(n.line_off_h[b + 1] - n.line_off_h[b] == T, "state externalisation needs equal-length lines (a Sequence is size x rows x cols)");
  return T;
}
struct HostArrays {   // host mirrors of the device arrays the walk touches
  std::vector<float> X, Z;
  std::vector<std::vector<float>> G, C, H, S;
};
// element (t, i, b) of entry e <-> (array, flat index)
static float* state_elem(const Net& n, HostArrays& a, const StateEntry& e, int T, int t, int i, int b) {
  const int tf = e.rev ? T - 1 - t : t;
  const size_t f = (size_t)n.line_off_h[b] + tf;
  switch (e.kind) {
    case StateEntry::X: return &a.X[f * n.desc.ninput + i];
    case StateEntry::Z: return &a.Z[f * n.desc.nclasses + i];
    case StateEntry::LIN:
      if (e.layer == 0) return &a.X[f * n.desc.ninput + i];
      return &a.H[e.layer - 1][f * n.L[e.layer - 1].ldh + n.L[e.layer - 1].hofs + i];
    case StateEntry::HBOTH: return &a.H[e.layer][f * n.L[e.layer].ldh + n.L[e.layer].hofs + i];
    case StateEntry::HDIR: return &a.H[e.layer][f * n.L[e.layer].ldh + n.L[e.layer].hofs + e.dir * n.L[e.layer].no + i];
    case StateEntry::GATE: return &a.G[e.layer][((f * n.ndir + e.dir) * n.L[e.layer].no + i) * 4 + e.slot];
    case StateEntry::CELL: return &a.C[e.layer][(f * n.ndir + e.dir) * n.L[e.layer].no + i];
    case StateEntry::SOURCE: return &a.S[e.layer][(size_t)e.dir * n.N * n.L[e.layer].lds + f * n.L[e.layer].lds + 1 + i];
  }
  return nullptr;
}
static long long states_total(const Net& n, int T) {
  long long total = 0;
  for (auto& e : state_walk(n)) total += (long long)T * e.rows * n.bs + 4;   // n_states, clstm.cc:762-769
  return total;
}
}  // namespace clstm
int clstm_net_n_states(clstm_net* h, long long* out) {
  ABI_BEGIN
  REQUIRE(out, "null argument");
  *out = states_total(h->net, states_T(h->net));
  ABI_END
}
static void states_transfer(clstm_net* h, float* data, long long total, bool get) {
  Net& n = h->net;
  const int T = states_T(n);
  REQUIRE(total == states_total(n, T), get ? "size mismatch in get_states" : "size mismatch in set_states");
  HostArrays a;
  const size_t N = (size_t)n.N;
  a.X.resize(N * n.desc.ninput); a.Z.resize(N * n.desc.nclasses);
  a.G.resize(n.L.size()); a.C.resize(n.L.size()); a.H.resize(n.L.size()); a.S.resize(n.L.size());
  n.flush_line_off();
  // both directions start from the device contents: set_states only overwrites what the walk covers
  copy_d2h(a.X.data(), n.X.p, a.X.size());
  copy_d2h(a.Z.data(), n.Z.p, a.Z.size());
  for (size_t l = 0; l < n.L.size(); l++) {
    Layer& y = n.L[l];
    a.G[l].resize(N * n.ndir * 4 * y.no); a.C[l].resize(N * n.ndir * y.no);
    a.H[l].resize(N * y.ldh); a.S[l].resize(N * n.ndir * y.lds);
    n.ensure_source((int)l); n.ensure_h_f32((int)l);
    copy_d2h(a.G[l].data(), y.G.p, a.G[l].size()); copy_d2h(a.C[l].data(), y.C.p, a.C[l].size());
    copy_d2h(a.H[l].data(), y.H.p, a.H[l].size()); copy_d2h(a.S[l].data(), y.S.p, a.S[l].size());
  }
  long long index = 0;
  for (auto& e : state_walk(n)) {
    if (get) {
      data[index++] = 999999.0f; data[index++] = (float)T; data[index++] = (float)e.rows; data[index++] = (float)n.bs;
    } else {   // set_states: magic, size, rows, cols must describe this net and batch (clstm.cc:795-803)
      REQUIRE((int)data[index] == 999999 && (int)data[index + 1] == T && (int)data[index + 2] == e.rows &&
                  (int)data[index + 3] == n.bs, "size mismatch in set_states");
      index += 4;
    }
    for (int t = 0; t < T; t++)
      for (int i = 0; i < e.rows; i++)
        for (int b = 0; b < n.bs; b++) {
          float* p = state_elem(n, a, e, T, t, i, b);
          if (get) data[index++] = *p; else *p = data[index++];
        }
  }
  REQUIRE(index == total, "size mismatch in states walk");
  if (!get) {
    copy_h2d(n.X.p, a.X.data(), a.X.size());
    copy_h2d(n.Z.p, a.Z.data(), a.Z.size());
    for (size_t l = 0; l < n.L.size(); l++) {
      Layer& y = n.L[l];
      for (size_t f = 0; f < N; f++) {   // the bias inputs of the packed rows are constants, not states
        a.H[l][f * y.ldh + y.hofs - 1] = 1.0f;
        for (int d = 0; d < n.ndir; d++) a.S[l][(size_t)d * N * y.lds + f * y.lds] = 1.0f;
      }
      copy_h2d(y.G.p, a.G[l].data(), a.G[l].size()); copy_h2d(y.C.p, a.C[l].data(), a.C[l].size());
      copy_h2d(y.H.p, a.H[l].data(), a.H[l].size()); copy_h2d(y.S.p, a.S[l].data(), a.S[l].size());
      y.sbf_ready = false;   // the bf16 copies made by the forward pass no longer match these states
      y.h_f32_valid = y.sh_valid = y.sx_valid = true;   // (the f32 arrays were just written whole)
      y.sx_valid = true;
    }
    n.src0_ready = true;
  }
}
int clstm_net_get_states_h(clstm_net* h, float* data, long long total) {
  ABI_BEGIN
  REQUIRE(data, "null argument");
  states_transfer(h, data, total, true);
  ABI_END
}
int clstm_net_set_states_h(clstm_net* h, const float* data, long long total) {
  ABI_BEGIN
  REQUIRE(data && total >= 4, "null argument");
  Net& n = h->net;
  // the first header carries the batch geometry (set_states resizes every Sequence from its header, clstm.cc:804)
  REQUIRE((int)data[0] == 999999, "bad magic in set_states");
  const int T = (int)data[1], bs = (int)data[3];
  REQUIRE(T > 0 && bs > 0 && (int)data[2] == n.desc.ninput, "size mismatch in set_states");
  std::vector<int> Ts(bs, T);
  n.set_batch(Ts.data(), bs);
  states_transfer(h, const_cast<float*>(data), total, false);
  ABI_END
}

// ---- data-parallel exchange ---------------------------------------------------------------------------
struct clstm_comm { Comm c; };
int clstm_comm_unique_id(char* id_h) {
  ABI_BEGIN
  REQUIRE(id_h, "null argument");
#ifndef CLSTM_HIP_EMU
  if (getenv("CLSTM_COMM_NO_RCCL") && atoi(getenv("CLSTM_COMM_NO_RCCL")) != 0) {   // (tests: ranks that share one device)
    FILE* f = fopen("/dev/urandom", "rb");
    REQUIRE(f && fread(id_h, 1, CLSTM_COMM_ID_BYTES, f) == (size_t)CLSTM_COMM_ID_BYTES, "cannot read /dev/urandom");
    fclose(f);
    return 0;
  }
  ncclUniqueId id;
  static_assert(sizeof(id) == CLSTM_COMM_ID_BYTES, "ncclUniqueId size");
  RCCLCHECK(RcclApi::get().GetUniqueId(&id));
  memcpy(id_h, &id, sizeof(id));
#else
  static int counter = 0;
  memset(id_h, 0, CLSTM_COMM_ID_BYTES);
  snprintf(id_h, CLSTM_COMM_ID_BYTES, "/clstm_emu_%d_%d", (int)getpid(), counter++);   // name of the shared segment
#endif
  ABI_END
}
int clstm_comm_create(clstm_comm** out, const char* id_h, int rank, int nranks) {
  ABI_BEGIN
  REQUIRE(out && id_h && nranks >= 1 && rank >= 0 && rank < nranks, "bad communicator arguments");
  clstm_comm* c = new clstm_comm();
  c->c.rank = rank; c->c.nranks = nranks;
#ifndef CLSTM_HIP_EMU
  try {
    memcpy(c->c.id, id_h, CLSTM_COMM_ID_BYTES);
    // CLSTM_COMM_NO_RCCL=1 (tests): no RCCL communicator -- the exchange is the peer-read path alone, which also works
    // between rank processes that share ONE device (RCCL refuses that: "duplicate GPU")
    if (!(getenv("CLSTM_COMM_NO_RCCL") && atoi(getenv("CLSTM_COMM_NO_RCCL")) != 0)) {
      ncclUniqueId id;
      memcpy(&id, id_h, sizeof(id));
      RCCLCHECK(RcclApi::get().CommInitRank(&c->c.comm, nranks, id, rank));
    }
  } catch (...) { delete c; throw; }
#else
  try { c->c.open(id_h, rank, nranks); } catch (...) { delete c; throw; }
#endif
  *out = c;
  ABI_END
}
int clstm_comm_destroy(clstm_comm* c) { ABI_BEGIN delete c; ABI_END }
int clstm_comm_rank(clstm_comm* c) { return c ? c->c.rank : 0; }
int clstm_comm_size(clstm_comm* c) { return c ? c->c.nranks : 1; }
int clstm_comm_peer_active(clstm_comm* c) { return c && c->c.peer.ok ? 1 : 0; }
int clstm_allreduce_flat(clstm_comm* c, float* buf_d, long long n) {
  ABI_BEGIN
  REQUIRE(c && buf_d && n >= 0, "bad all-reduce arguments");
  if (n > 0) c->c.allreduce(buf_d, n, g_stream);
  ABI_END
}
int clstm_net_set_overlap(clstm_net* h, int mode) {
  ABI_BEGIN
  REQUIRE(mode >= 0 && mode <= 2, "overlap mode: 0 off, 1 one launch with two roles where it pays, 2 the same always (tests)");
  h->net.overlap = mode;
  ABI_END
}
int clstm_net_set_strict_f32(clstm_net* h, int on) {
  ABI_BEGIN
  Net& n = h->net;
  if (on) { n.dw_x3 = 0; n.gemm_x3_on = false; }
  else {
    n.dw_x3 = dbg_opt("dw_x3", 1);
    n.gemm_x3_on = dbg_opt("gemm_x3", 1) != 0;
    n.split_terms = dbg_opt("split_terms", 3);
  }
  n.packed_dirty = true;   // (the hi | lo weights of the f32-grade backward recurrence are only packed while that mode is on)
  ABI_END
}
int clstm_net_overlap_stats(clstm_net* h, long long* launches, int* timeouts) {
  ABI_BEGIN
  Net& n = h->net;
  if (launches) *launches = n.dw_launches;
  if (timeouts) {
    *timeouts = 0;
    if (g_dev_err) {   // (process-wide count; also reported -- and cleared -- by the next synchronisation point)
      HIPCHECK(hipStreamSynchronize(g_stream));
      HIPCHECK(hipMemcpy(timeouts, g_dev_err + 1, sizeof(int), hipMemcpyDeviceToHost));
    }
  }
  ABI_END
}
int clstm_net_set_comm(clstm_net* h, clstm_comm* c) { h->net.comm = c ? &c->c : nullptr; return 0; }
int clstm_net_replica_check(clstm_net* h) {
  ABI_BEGIN
  REQUIRE(h, "null argument");
  h->net.replica_check(g_stream);   // (no-op without a communicator of several ranks; the verdict arrives with the next synchronisation)
  ABI_END
}
int clstm_net_set_training(clstm_net* h, int on) {
  ABI_BEGIN
  REQUIRE(h, "null argument");
  h->net.training = on != 0;
  ABI_END
}

// ---- diagnostics ----------------------------------------------------------------------------------
int clstm_debug_ctc_cycles(long long* out_h) {
  ABI_BEGIN
  REQUIRE(g_last_ctc_prof, "no CTC launch yet");
  HIPCHECK(hipStreamSynchronize(g_stream));
  HIPCHECK(hipMemcpy(out_h, g_last_ctc_prof, 16 * sizeof(long long), hipMemcpyDeviceToHost));
  ABI_END
}
#ifdef CLSTM_LSTM_PROF
int clstm_debug_lstm_cycles(clstm_net* h, long long* out_h) {   // diagnostics build only (not in the product ABI)
  ABI_BEGIN
  HIPCHECK(hipStreamSynchronize(g_stream));
  HIPCHECK(hipMemcpy(out_h, h->net.lstm_prof.p, 96 * sizeof(long long), hipMemcpyDeviceToHost));
  ABI_END
}
#endif
int clstm_debug_set_device_error(int which, int value) {   // tests: what a failed persistent launch / a timed-out item leaves behind
  ABI_BEGIN
  REQUIRE(which >= 0 && which <= 7, "bad error word");
  HIPCHECK(hipStreamSynchronize(g_stream));
  if (which == 4) { g_debug_fail_claims = value & 255; g_debug_fail_skip = value >> 8; return 0; }   // after (value >> 8) persistent launches the next (value & 255) fail their placement check
  if (which == 5) {                                                  // forget a placement failure: persistent launches again, `value` of them verified synchronously
    g_xcd_failed = false;
    g_xcd_outcome.check_all();
    g_xcd_outcome.verified = value ? 4 : 0;
    return 0;
  }
  HIPCHECK(hipMemcpy(dev_err_words() + which, &value, sizeof(int), hipMemcpyHostToDevice));
  ABI_END
}
int clstm_debug_set_option(const char* name, int value) {   // experiment switches (dbgopt.h); name NULL: forget every option set so far
  ABI_BEGIN
  HIPCHECK(hipStreamSynchronize(g_stream));
  if (!name) dbg_opts().clear(); else dbg_opts()[name] = value;
  ABI_END
}
int clstm_debug_path_count(int which, long long* out_h) {
  ABI_BEGIN
  REQUIRE(which >= 0 && which < 24 && out_h, "bad path index");
  *out_h = g_path_count[which];
  ABI_END
}
#ifdef CLSTM_GEMM_PROF
int clstm_debug_gemm_prof(long long* out_h) {   // diagnostics build only (not in the product ABI)
  ABI_BEGIN
  HIPCHECK(hipStreamSynchronize(g_stream));
  HIPCHECK(hipMemcpyFromSymbol(out_h, HIP_SYMBOL(clstm_gemm_prof), 16 * sizeof(long long)));
  ABI_END
}
#endif
int clstm_debug_lane_ops(float* out) {
  ABI_BEGIN
  CLSTM_LAUNCH(k_debug_lane_ops, dim3(1), dim3(64), 0, g_stream, out);
  check_launch();
  ABI_END
}
int clstm_debug_gemm(int mode, const float* A, const float* B, float* Cm, int R, int Cn, int K, int nsplit) {
  ABI_BEGIN
  static thread_local DevBuf<float>* part = nullptr;
  // user arrays are exact-size: no slack, the descriptor ends at the last element
  if (mode == 0) gemm_f32<GEMM_KC, GEMM_MC>(g_stream, gemm_kc(A, K, R, 0), gemm_mc(B, Cn, K, 0), StorePlain{Cm, Cn}, R, Cn, K);
  else if (mode == 1) gemm_f32<GEMM_KC, GEMM_KC>(g_stream, gemm_kc(A, K, R, 0), gemm_kc(B, K, Cn, 0), StorePlain{Cm, Cn}, R, Cn, K);
  else if (mode == 2) {
    if (!part) part = new DevBuf<float>();
    if (nsplit < 1) nsplit = 1;
    part->reserve((size_t)nsplit * R * Cn);
    gemm_f32<GEMM_MC, GEMM_MC>(g_stream, gemm_mc(A, R, K, 0), gemm_mc(B, Cn, K, 0), StorePartial{part->p, R, Cn}, R, Cn, K, nsplit);
    CLSTM_LAUNCH(k_reduce_scatter, dim3(nblocks((size_t)R * Cn)), dim3(256), 0, g_stream,
                 (ReduceDesc{part->p, nullptr, 0LL, nsplit, 1, R, Cn, Cn}), (ReduceDesc{}), Cm, (int*)nullptr, 0, (UpdateFuse{}));
  } else if (mode == 10) gemm_bf16<GEMM_KC, GEMM_MC>(g_stream, gemm_kc(A, K, R, 0), gemm_mc(B, Cn, K, 0), StorePlain{Cm, Cn}, R, Cn, K);
  else if (mode == 11) gemm_bf16<GEMM_KC, GEMM_KC>(g_stream, gemm_kc(A, K, R, 0), gemm_kc(B, K, Cn, 0), StorePlain{Cm, Cn}, R, Cn, K);
  else if (mode == 12) {
    if (!part) part = new DevBuf<float>();
    if (nsplit < 1) nsplit = 1;
    part->reserve((size_t)nsplit * R * Cn);
    gemm_bf16<GEMM_MC, GEMM_MC>(g_stream, gemm_mc(A, R, K, 0), gemm_mc(B, Cn, K, 0), StorePartial{part->p, R, Cn}, R, Cn, K, nsplit);
    CLSTM_LAUNCH(k_reduce_scatter, dim3(nblocks((size_t)R * Cn)), dim3(256), 0, g_stream,
                 (ReduceDesc{part->p, nullptr, 0LL, nsplit, 1, R, Cn, Cn}), (ReduceDesc{}), Cm, (int*)nullptr, 0, (UpdateFuse{}));
  } else if (mode == 34) {   // the kk product with its tiles brought in by LDS-DMA (gemm_b16kk_dma_kernel)
    gemm_b16kk(g_stream, GemmOperand16{(const unsigned short*)A, K, (long long)R * K}, GemmOperand16{(const unsigned short*)B, K, (long long)Cn * K},
               StorePlain{Cm, Cn}, R, Cn, K, 2);
  } else if (mode == 30 || mode == 31) {   // A: [R][K] bf16, B: [Cn][K] bf16 (the caller passes halfs in float-typed pointers); 31: the one-barrier loop
    gemm_b16kk(g_stream, GemmOperand16{(const unsigned short*)A, K, (long long)R * K}, GemmOperand16{(const unsigned short*)B, K, (long long)Cn * K},
               StorePlain{Cm, Cn}, R, Cn, K, mode == 31 ? 0 : 1 | (nsplit > 1 ? nsplit << 4 : 0));   // (diagnostics build: nsplit = leave-out bits)
  } else if (mode == 32 || mode == 33 || mode == 35) {   // A: [K][R] bf16, B: [K][Cn] bf16 (R, Cn multiples of 8), split-K slabs reduced afterwards; 33: the one-barrier loop
    if (!part) part = new DevBuf<float>();
    if (nsplit < 1) nsplit = 1;
    part->reserve((size_t)nsplit * R * Cn);
    gemm_b16mc(g_stream, GemmOperand16B{(const unsigned short*)A, R, (long long)K * R, 0}, GemmOperand16B{(const unsigned short*)B, Cn, (long long)K * Cn, 0},
               StorePartial{part->p, R, Cn}, R, Cn, K, nsplit, 1, GemmOperand16B{nullptr, 0, 0, 0}, 0, mode == 33 ? 0 : mode == 35 ? 3 : 1);   // 35: tiles by LDS-DMA
    CLSTM_LAUNCH(k_reduce_scatter, dim3(nblocks((size_t)R * Cn)), dim3(256), 0, g_stream,
                 (ReduceDesc{part->p, nullptr, 0LL, nsplit, 1, R, Cn, Cn}), (ReduceDesc{}), Cm, (int*)nullptr, 0, (UpdateFuse{}));
  } else if (mode == 20) gemm_x3<GEMM_KC, GEMM_MC>(g_stream, gemm_kc(A, K, R, 0), gemm_mc(B, Cn, K, 0), StorePlain{Cm, Cn}, R, Cn, K);
  else if (mode == 21) gemm_x3<GEMM_KC, GEMM_KC>(g_stream, gemm_kc(A, K, R, 0), gemm_kc(B, K, Cn, 0), StorePlain{Cm, Cn}, R, Cn, K);
  else if (mode == 22) {
    if (!part) part = new DevBuf<float>();
    if (nsplit < 1) nsplit = 1;
    part->reserve((size_t)nsplit * R * Cn);
    gemm_x3<GEMM_MC, GEMM_MC>(g_stream, gemm_mc(A, R, K, 0), gemm_mc(B, Cn, K, 0), StorePartial{part->p, R, Cn}, R, Cn, K, nsplit);
    CLSTM_LAUNCH(k_reduce_scatter, dim3(nblocks((size_t)R * Cn)), dim3(256), 0, g_stream,
                 (ReduceDesc{part->p, nullptr, 0LL, nsplit, 1, R, Cn, Cn}), (ReduceDesc{}), Cm, (int*)nullptr, 0, (UpdateFuse{}));
  } else if (mode == 23) gemm_x3_big<GEMM_KC, GEMM_MC>(g_stream, gemm_kc(A, K, R, 0), gemm_mc(B, Cn, K, 0), StorePlain{Cm, Cn}, R, Cn, K);
  else if (mode == 24) gemm_x3_big<GEMM_KC, GEMM_KC>(g_stream, gemm_kc(A, K, R, 0), gemm_kc(B, K, Cn, 0), StorePlain{Cm, Cn}, R, Cn, K);
  else if (mode == 25) {
    if (!part) part = new DevBuf<float>();
    if (nsplit < 1) nsplit = 1;
    part->reserve((size_t)nsplit * R * Cn);
    gemm_x3_big<GEMM_MC, GEMM_MC>(g_stream, gemm_mc(A, R, K, 0), gemm_mc(B, Cn, K, 0), StorePartial{part->p, R, Cn}, R, Cn, K, nsplit);
    CLSTM_LAUNCH(k_reduce_scatter, dim3(nblocks((size_t)R * Cn)), dim3(256), 0, g_stream,
                 (ReduceDesc{part->p, nullptr, 0LL, nsplit, 1, R, Cn, Cn}), (ReduceDesc{}), Cm, (int*)nullptr, 0, (UpdateFuse{}));
  } else throw Error("bad mode");
  check_launch();
  ABI_END
}

}  // extern "C"
