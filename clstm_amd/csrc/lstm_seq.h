// lstm_seq.h -- persistent per-line NPLSTM recurrence kernels (forward and backward).
//
// Replaces the per-timestep op chain of GenericNPLSTM::forward / ::backward
// (clstm.cc:600-653: forward_stack_delay, 4x forward_full1, forward_statemem,
// forward_nonlingate and their backward counterparts, clstm_compute.cc:275-547) by ONE
// launch per layer and pass: workgroup (line b, direction d) walks its own T_b frames without
// returning to the host.  `Reversed` (clstm.cc:458-478) is index arithmetic: direction 1
// visits frame T-1-s at its own step s.
//
// Only the recurrent half R.h_{t-1} of the gate products is inside the loop; the input half
// W_x.x_t + b for every frame of the minibatch is one hoisted MFMA GEMM (gemm_mfma.h), whose
// result G[token][dir][cell][gate] this kernel overwrites in place with the activations.
//
// Thread layout ("wavefront per cell tile"): wave w owns cells 16w..16w+15.
//   forward : lane = 4*cell_local + q.  Lane (cell,q) keeps the 4 gates x KQP recurrent weights
//             R_g[cell][k], k in quarter q, in VGPRs for the whole sequence; per step it reads
//             its quarter of h_{t-1} from LDS (float4, conflict-free: quarter stride/4 is odd),
//             does 4*KQP FMAs, sums the four quarters with two DPP quad_perm adds, then lane q
//             applies gate q's nonlinearity, the quad exchanges the four activations by DPP
//             broadcast and every lane forms c_t and h_t (fused statemem + nonlingate).
//   backward: matvec phase lane = 16*r + js : k-group (4 output cells) x 16 slices of the 4*no
//             (gate,j) delta pairs, weights R_g[j][k] in VGPRs, deltas from LDS; a 4-step DPP
//             row_ror all-reduce leaves dh_rec for the 4 cells in every lane of the row; the
//             element-wise phase re-uses the forward's (cell, gate) = (lane>>2, lane&3) mapping.
// One __syncthreads per step (h / delta vector double-buffered in LDS).
//
// What a step costs (measured per wave, scripts/gpu_lstmprof.py): FMA phase of the first
// wave of a SIMD + FMA phase of the second + the dependent tail of the second.  Hence: everything that does
// not depend on the step's mat-vec runs BEFORE it (backward), everything fire-and-forget runs AFTER the
// barrier (forward: the global stores of step t issue at the top of step t+1), prefetches are issued at the
// top of a step into a register set that is dead (three rotating sets, backward), and no s_waitcnt sized for
// the first iteration may sit inside the loop (values loaded in the prologue are touched before it).
#pragma once
#include <type_traits>
#include "devintrin.h"

namespace clstm {

struct LstmSeqArgs {
  const float* Rpk;     // packed recurrent weights for this pass: [dir][16*NK4][nthreads]
  float* G;             // [N][2][4*no]  fwd: in pre-activation (x part + bias), out activation
  float* C;             // [N][2][no]    cell state
  float* H;             // [N][ldh]      outputs: column hofs-1 holds the constant 1 (bias input of the
                        //               next layer / softmax), dir d at column hofs + d*no = Parallel's stacked
                        //               output; hofs = 4 keeps the h block of a row 16-byte aligned
  int ldh, hofs;
  const float* dH;      // [N][2*no]     bwd: delta on H            (backward only)
  float* D;             // [N][2][4*no]  bwd: gate pre-activation deltas (backward only)
  const int* line_off;  // [bs+1] first token of each line
  const int* order;     // [bs] line that the b-th workgroup of a direction walks (longest first), or null
  int no;
  int ndir;             // 2 (bidirectional) or 1 (forward only, "lstm1")
  float* S;             // [ndir][N][lds] source rows [1 | x_t | h_{t-1}] for the weight-gradient GEMM;
  int lds, sofs;        //   the forward pass deposits h_{t-1} at column sofs = 1 + ni of the NEXT step's row
  long long sdir;       //   floats between the two directions' S arrays
  long long* prof;      // diagnostics build (-DCLSTM_LSTM_PROF) only: [8 waves][12] summed phase cycles of workgroup 0
  // Progress words for consumers that run CONCURRENTLY with the recurrence (backward: the weight-gradient items of
  // gemm_dw.h reading D; forward, fused launch of lstm_fwd_fused.h: the softmax items reading H): word (dir, line) lives
  // prog_off floats behind D[0] / H[0] (the tail of that array's allocation, so that the per-step store reaches it
  // through the same descriptor) and counts the iterations of that line whose stores are complete in memory.  -1: none.
  long long prog_off;
  int prog_base;        // value that means "0 iterations complete" for this launch (monotonic across launches)
  int bs;               // lines in the batch (index stride of the progress words)
  // forward, fused launch only -- the gate pre-activations G are PRODUCED inside the launch (16 iterations of one line
  // and direction per item): flag (dir, line, chunk) == gepoch once that chunk's rows are in memory
  const int* gflag;     // [ndir][bs][gchunks]
  int gchunks, gepoch;
  int* timeouts;        // watchdog count (device error word) for the waits above
};

// One workgroup per CU, at most two of its waves per SIMD: tell hipcc, or its scheduler trades the up-front issue of
// a step's LDS reads for a register count that would admit a third wave nobody launches (seen: 166 -> 148 VGPRs,
// every ds_read_b128 followed by lgkmcnt(0), backward 92 -> 104 us).
#ifdef CLSTM_HIP_EMU
#define CLSTM_TWO_WAVES_PER_SIMD
#else
#define CLSTM_TWO_WAVES_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 2)))
#endif
constexpr int lstm_qstride(int nk4) { return 4 * nk4 + ((nk4 & 1) ? 0 : 4); }

// NK4: float4 groups of k per lane (register / LDS capacity 4*NK4); KU <= 4*NK4: k values a lane really
// owns = cells per quarter.  (7, 25) is the 100-cell instantiation: 50 instead of 56 packed FMAs per step.
// FUSED (lstm_fwd_fused.h): the same recurrence as one role of a launch whose other workgroups produce G ahead of it
// and consume H behind it.  Three things change, none of them on the step's dependent chain:
//  * G is read with system-scope loads (written through by the producers, per-XCD L2s are not coherent).  The workgroup
//    computes the pre-activations of its OWN first 16 iterations before it starts (seven waves, ~3 us: waiting for a
//    producer's chunk 0 cost 8-10 us); from then on ONE wave -- wave 3, the first-dispatched wave that has a SIMD to
//    itself and idles ~350 cycles at every barrier -- makes sure the chunk of the loads about to be issued is there:
//    it requests the chunk flag two steps before it needs the answer (the same software pipeline as the
//    pre-activations) and, if the chunk is missing, spins in front of its barrier arrival, which holds the workgroup;
//  * H is stored write-through, and the LAST lane of the workgroup, which owns no cell (4 no < threads), rides that
//    store with its own address and data: at the flush of step tp it writes that iterations < tp - 2 are complete
//    (every wave has passed the barrier of step tp, i.e. has waited for a load it issued behind its stores of step
//    tp - 3; VMEM operations of a wave complete in order), exactly as the backward kernel reports its deltas.
struct FwdFusedArgs;
// (lstm_fwd_fused.h) the workgroup computes the gate pre-activations of its own first 16 iterations
DEVFN void fwd_self_produce(const LstmSeqArgs& a, const FwdFusedArgs& h, int b, int dir, int off, int T);
template <int NK4, int KU, bool FUSED>
DEVFN void lstm_fwd_body(const LstmSeqArgs& a, const int b, const int dir, const FwdFusedArgs* fh = nullptr) {
  constexpr int KQP = 4 * NK4;
  constexpr int QS = KQP + ((NK4 & 1) ? 0 : 4);
  constexpr int HB = 4 * QS;
  float* lds = dyn_smem<float>();  // hbuf[2][HB] + dump word
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x;          // 64 * ceil(no / 16) <= 64 * NK4
  const int no = a.no;
  const int q = lane & 3, cell = wave * 16 + (lane >> 2);
  const bool valid = cell < no;
  const bool lead = valid && q == 0;
  const int nd = a.ndir;

  // recurrent weights as (gi,gf) and (go,ci) pairs: one v_pk_fma_f32 serves two gates
  f32x2 w01[KQP], w23[KQP];
  {
    const float* rp = a.Rpk + (size_t)dir * 4 * KQP * nthreads + tid;
#pragma unroll
    for (int kk = 0; kk < KQP; kk++) {
      w01[kk] = (f32x2){rp[(size_t)(0 * KQP + kk) * nthreads], rp[(size_t)(1 * KQP + kk) * nthreads]};
      w23[kk] = (f32x2){rp[(size_t)(2 * KQP + kk) * nthreads], rp[(size_t)(3 * KQP + kk) * nthreads]};
    }
  }
  for (int i = tid; i < 2 * HB + 4; i += nthreads) lds[i] = 0.0f;

  const int off = a.line_off[b];
  const int T = a.line_off[b + 1] - off;
  // per-line windows; frame fr(t) within the line (Reversed = index arithmetic); byte offsets are
  // lane part + wave-uniform frame part, masked lanes sit at BUF_OOB_BASE
  const unsigned gstride4 = (unsigned)nd * 4 * no * 4, cstride4 = (unsigned)nd * no * 4;
  const BufF32 gbuf = make_buf(a.G + (size_t)off * (gstride4 / 4), (size_t)T * gstride4);
  const BufF32 cbuf = make_buf(a.C + (size_t)off * (cstride4 / 4), (size_t)T * cstride4);
  const unsigned hstride4 = (unsigned)a.ldh * 4;
  // (fused: the descriptor reaches this line's progress word behind the array)
  const long long prog_rel = FUSED ? a.prog_off + ((long long)dir * a.bs + b) * PROG_STRIDE - (long long)off * a.ldh : 0;
  const BufF32 hbuf = make_buf(a.H + (size_t)off * a.ldh, FUSED ? (size_t)(prog_rel + 1) * 4 : (size_t)T * hstride4);
  const bool tagl = FUSED && tid == nthreads - 1;          // requires cell(tid) >= no (checked by the host)
  const unsigned ptag = (unsigned)prog_rel * 4u;
  const unsigned sstride4 = (unsigned)a.lds * 4;
  const BufF32 sbuf = make_buf(a.S + (size_t)dir * a.sdir + (size_t)off * a.lds, (size_t)T * sstride4);
  const unsigned gl = valid ? ((unsigned)dir * 4 * no + cell * 4 + q) * 4u : BUF_OOB_BASE;
  const unsigned cl = lead ? ((unsigned)dir * no + cell) * 4u : BUF_OOB_BASE;
  const unsigned sl = lead ? ((unsigned)a.sofs + cell) * 4u : BUF_OOB_BASE;
  const unsigned hl = lead ? ((unsigned)a.hofs + (unsigned)dir * no + cell) * 4u : BUF_OOB_BASE;
  auto fr = [&](int t) -> unsigned {  // clamped: prefetches past the end re-read the last frame
    const int tc = t < T ? t : T - 1;
    return (unsigned)(dir == 0 ? tc : T - 1 - tc);
  };
  // Workgroups of more than four waves run STAGGERED: group A = waves 0..3 (cells 0..63, the first wave of every
  // SIMD), group B = the rest (each shares a SIMD with an A wave); a lane's contraction slots are ordered
  // [A cells | B cells] (devintrin.h:stag_*).  With ONE barrier per step the two waves of a SIMD issue their FMAs in the
  // same window and then run their dependent tails side by side -- measured (scripts/gpu_width_sweep.py): 64 cells / 4
  // waves 317 ns per step, 100 cells / 7 waves 505 ns, although a lane has only 50 instead of 32 packed FMAs.  Two
  // barriers per step let the groups run half a step apart: X_{t+1} = "A's half of h_t is in LDS", Y_{t+1} = "B's half".
  //   A, step t:  reads own half | stores of t-1 | FMAs own half | Y_t | reads B's half | FMAs | tail | write | X_{t+1}
  //   B, step t:  reads all | stores of t-1 | FMAs | X_{t+1} | tail | write | Y_{t+1}
  // so A's tail runs under B's FMAs and B's tail under A's first FMAs: 505 -> 465 ns per step at 100 cells (pure kernel
  // 100.9 -> 92.9 us, the fused forward launch 118.5 -> 114.5).  What did NOT help (all measured, gpurun_out of round 3):
  // moving the split (8 / 16 / 24 of A's 32 own FMAs in front of Y; X behind B's reduction or nonlinearity): +-1 us;
  // raising the priority of a wave in its tail (s_setprio): +2..4 us; B fetching A's half right behind X into registers:
  // +14 us; B's stores behind X instead of in front of its FMAs: +20 us; B's FMAs at raised priority: +17 us.  A lone wave's
  // chain (LDS read -> 50 FMAs -> reduce, two transcendental rounds, LDS write -> barrier) is ~850 cycles; this runs 1120.
  constexpr bool STAG = stag_on(NK4);
  constexpr int NJA = STAG ? STAG_KA / 4 : NK4;   // float4 groups of a lane's slice that hold group A's cells
  const int hslot = STAG ? stag_fwd_slot(valid ? cell : 0, KU, QS) : (cell / KU) * QS + (cell % KU);
  // lane q finishes gate q: q = 0 gi, 1 gf, 2 go (sigmoid), 3 ci (tanh) -- one affine form for both (act_affine)
  const float a_scale = q == 3 ? ACT_TANH_SCALE : ACT_SIG_SCALE, a_mul = q == 3 ? 2.0f : 1.0f, a_add = q == 3 ? -1.0f : 0.0f;
  const float* rdA = lds + q * QS;            // even steps read buffer 0, write buffer 1
  const float* rdB = lds + HB + q * QS;
  float* wrA = lead ? lds + HB + hslot : lds + 2 * HB;
  float* wrB = lead ? lds + hslot : lds + 2 * HB;
  float c_prev = 0.0f;
  if (T <= 0) {   // an empty line is complete at once
    if (FUSED && tid == 0) store_i32_wt(reinterpret_cast<int*>(a.H + a.prog_off) + ((size_t)dir * a.bs + b) * PROG_STRIDE, a.prog_base);
    return;
  }
  const int nchunk = (T + 15) >> 4;
  const int* gfl = FUSED ? a.gflag + ((size_t)dir * a.bs + b) * a.gchunks : nullptr;
  auto gload = [&](unsigned o) -> float { return FUSED ? buf_load_wt(gbuf, o) : buf_load(gbuf, o); };
  // fused: wait (every wave by itself, wave-uniform) until the chunk is there
  auto wait_chunk = [&](int c) {
    int spins = 0;
    while (wave_uniform(load_i32_wt(gfl + c)) != a.gepoch) {
      sleep_iterations(1);
      if (++spins > (1 << 20)) { if (lane == 0) atomic_add_i32(a.timeouts, 1); break; }   // never hang the device
    }
  };
  if constexpr (FUSED) {   // chunk 0 of this line and direction: computed here, written through, then visible to every wave
    fwd_self_produce(a, *fh, b, dir, off, T);
    drain_vmem();
    __syncthreads();
  }
  // input pre-activations are fetched two steps ahead into two alternating registers (the loop is
  // unrolled by two so that no register rotation forces an early wait on an in-flight load)
  float gxA = gload(gl + fr(0) * gstride4);
  float gxB = gload(gl + fr(1) * gstride4);
  float kaA0 = 0.f, kaA1 = 0.f, kaA2 = 0.f, kaB0 = 0.f, kaB1 = 0.f, kaB2 = 0.f;  // store-data pins
  buf_store(sbuf, sl + fr(0) * sstride4, 0.0f);  // h_{-1} = 0 (forward_stack_delay, last < 0)
  // Touch every value loaded so far HERE.  hipcc otherwise places the wait for the weight loads at their
  // first use inside the time loop -- a static s_waitcnt vmcnt(3) at the top of every step, sized for the
  // first iteration, which in steady state also waits for the previous step's stores (measured: ~80
  // stalled cycles per wave and step); the same goes for the first two pre-activation loads.
#pragma unroll
  for (int kk = 0; kk < KQP; kk++) { KEEP_ALIVE2(w01[kk]); KEEP_ALIVE2(w23[kk]); }
  KEEP_ALIVE(gxA); KEEP_ALIVE(gxB);
  __syncthreads();
  // diagnostics build only: per-phase cycle stamps (scripts/gpu_lstmprof.py).  Each stamp costs ~60 cycles
  // and drains lgkmcnt, so the instrumented step is ~25 % longer than the real one.
#ifdef CLSTM_LSTM_PROF
  long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long pt = 0;
#define LSTM_STAMP(k) do { long long now_; SCHED_FENCE(); \
                           asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
                           SCHED_FENCE(); pacc[k] += now_ - pt; pt = now_; } while (0)
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(pt) :: "memory");
#else
#define LSTM_STAMP(k) do {} while (0)
#endif
  // The global stores of a step are DEFERRED to the top of the next one: they are fire-and-forget, but
  // issuing four of them cost ~200 cycles at the end of the dependent tail of every step; behind the
  // barrier they issue while the wave would wait for its LDS reads / its SIMD's other wave anyway.
  // (pa*, pb*: data and offsets of the previous step = the other register set.)
  // (pa*: the previous step's activation, c and h = the other register set; tp: that step, -1 before the first.
  //  The offset arithmetic happens here too, off the tail.)
  auto flush = [&](const int tp, float pa0, float pa1, float pa2, auto report_tag) {
    constexpr bool REPORT = decltype(report_tag)::value;
    const bool any = tp >= 0;
    const unsigned f = fr(tp < 0 ? 0 : tp);
    buf_store(gbuf, any ? gl + f * gstride4 : BUF_OOB, pa0);
    buf_store(cbuf, any ? cl + f * cstride4 : BUF_OOB, pa1);
    if constexpr (REPORT) {   // (the wave with the reporting lane: address and data by select)
      const float sdat = tagl ? __builtin_bit_cast(float, a.prog_base + (tp - 2 > 0 ? tp - 2 : 0)) : pa2;
      buf_store_wt(hbuf, tagl ? (any ? ptag : BUF_OOB) : (any ? hl + f * hstride4 : BUF_OOB), sdat);
    } else if constexpr (FUSED) buf_store_wt(hbuf, any ? hl + f * hstride4 : BUF_OOB, pa2);
    else buf_store(hbuf, any ? hl + f * hstride4 : BUF_OOB, pa2);
    // h_t is the recurrent part of the NEXT step's source row (dropped after the last step)
    buf_store(sbuf, any && tp + 1 < T ? sl + fr(tp + 1) * sstride4 : BUF_OOB, pa2);
  };
  // WHICH side of the barrier a wave stores on depends on its role (measured: the four stores of all seven waves
  // issued together right behind the barrier queue at the texture addresser -- 28 wave-instructions at ~10-16
  // cycles each -- and the waves that define the step then start their LDS reads / FMAs that much later; with no
  // stores at all the kernel runs 91.6 us instead of 115.4).  The first-dispatched waves (one per SIMD, they win the
  // VALU arbitration and then sit ~400 cycles at the barrier) store at the END of their step; the later waves, whose
  // FMAs wait for the older wave of their SIMD anyway, store right behind the barrier.
  constexpr int EARLY_WAVES = 4;
  // The two roles are two copies of the whole time loop (one wave-uniform branch in front): inside a copy the
  // stores sit at a fixed place in straight-line code, so hipcc still counts the VMEM queue exactly (a branch
  // inside the step made it wait with vmcnt(1), i.e. for the previous step's stores).
  const bool early = wave_uniform(wave) < EARLY_WAVES;
  // fused roles: POLL = wave 3 (early), REPORT = the last wave (late); the host launches the fused form only for
  // workgroups of at least five waves whose last lane owns no cell
  auto run = [&](auto early_tag, auto poll_tag, auto report_tag) {
  constexpr bool EARLY = decltype(early_tag)::value, POLL = decltype(poll_tag)::value;
  int rdyA = a.gepoch, rdyB = a.gepoch;   // (POLL) chunk flags requested two steps ago
  auto step = [&](const int t, float& gxr, int& rdy, const float* hq, float* hw, float& ka0, float& ka1, float& ka2,
                  float pa0, float pa1, float pa2) {
    if constexpr (!EARLY && !STAG) {
      KEEP_ALIVE(ka0); KEEP_ALIVE(ka1); KEEP_ALIVE(ka2);  // stores of step t-2 have long completed
      flush(t - 1, pa0, pa1, pa2, report_tag);
    }
    if constexpr (STAG) { KEEP_ALIVE(ka0); KEEP_ALIVE(ka1); KEEP_ALIVE(ka2); }
    f32x2 a01 = splat2(0.0f), a23 = splat2(0.0f);
    LSTM_STAMP(0);   // loop overhead since the barrier
    // (hipcc copies element w of every ds_read_b128 into a fresh pair -- 6 v_mov_b32 per step; spelling the
    //  FMAs as inline asm with op_sel removes them but serialises the LDS reads behind single waits: not kept)
    // All LDS reads of the step first, pinned in front of the FMAs (see the backward kernel).
    float4 hv[NK4];
    auto fma_groups = [&](const int j0, const int j1) {
#pragma unroll
      for (int j = j0; j < j1; j++) {
        if (4 * j < KU) { a01 = fma2(w01[4 * j], splat2(hv[j].x), a01); a23 = fma2(w23[4 * j], splat2(hv[j].x), a23); }
        if (4 * j + 1 < KU) { a01 = fma2(w01[4 * j + 1], splat2(hv[j].y), a01); a23 = fma2(w23[4 * j + 1], splat2(hv[j].y), a23); }
        if (4 * j + 2 < KU) { a01 = fma2(w01[4 * j + 2], splat2(hv[j].z), a01); a23 = fma2(w23[4 * j + 2], splat2(hv[j].z), a23); }
        if (4 * j + 3 < KU) { a01 = fma2(w01[4 * j + 3], splat2(hv[j].w), a01); a23 = fma2(w23[4 * j + 3], splat2(hv[j].w), a23); }
      }
    };
    if constexpr (STAG && EARLY) {
      // group A: its own half of h_{t-1} has been visible since barrier X_t (the end of its previous step); group B's
      // half becomes visible at barrier Y_t, which B reaches at the end of ITS step t-1 -- half a step behind.
      // The previous step's global stores issue under the latency of the first LDS reads (both groups).
#pragma unroll
      for (int j = 0; j < NJA; j++) hv[j] = *reinterpret_cast<const float4*>(hq + 4 * j);
      SCHED_FENCE();
      flush(t - 1, pa0, pa1, pa2, report_tag);
      SCHED_FENCE();
      fma_groups(0, NJA);
      LSTM_STAMP(7);     // first LDS reads, deferred stores, the FMAs in front of Y
      __syncthreads();   // Y_t
      LSTM_STAMP(8);     // wait at Y
#pragma unroll
      for (int j = NJA; j < NK4; j++) hv[j] = *reinterpret_cast<const float4*>(hq + 4 * j);
      SCHED_FENCE();
      fma_groups(NJA, NK4);
    } else if constexpr (STAG) {
#pragma unroll
      for (int j = 0; j < NK4; j++) hv[j] = *reinterpret_cast<const float4*>(hq + 4 * j);
      SCHED_FENCE();
      flush(t - 1, pa0, pa1, pa2, report_tag);
      SCHED_FENCE();
      fma_groups(0, NK4);
    } else {
#pragma unroll
      for (int j = 0; j < NK4; j++) hv[j] = *reinterpret_cast<const float4*>(hq + 4 * j);
      SCHED_FENCE();
      fma_groups(0, NK4);
    }
    // (early role: the pins end only here, so that the LDS reads above cannot land in the store-data registers --
    //  hipcc guards an LDS return into such a register with s_waitcnt vmcnt(0), i.e. it would wait for the stores
    //  just issued)
    if constexpr (EARLY && !STAG) { KEEP_ALIVE(ka0); KEEP_ALIVE(ka1); KEEP_ALIVE(ka2); }
    LSTM_STAMP(1);   // LDS reads + FMAs
    if constexpr (STAG && !EARLY) {
      __syncthreads();   // X_{t+1}: group A has written its half of h_t
      LSTM_STAMP(9);     // wait at X
    }
    // reduce-scatter over the quad: lane q ends with gate q's sum over the four k-quarters
    // (register slot s of lane q holds gate s^q -- pack_rf -- so what a lane keeps and what it sends sit
    // in fixed registers: three v_add_f32_dpp, no selects)
    const float k0 = a01[0] + quad_xor2(a23[0]);
    const float k1 = a01[1] + quad_xor2(a23[1]);
    const float k = k0 + quad_xor1(k1);
    // lane q finishes gate q: q=0 gi, 1 gf, 2 go (sigmoid); 3 ci (tanh)   [forward_full1]
    const float pre = k + gxr;
    LSTM_STAMP(2);   // quad reduce + wait for the prefetched pre-activation
    // re-issue into the SAME register only now that its old value is dead (no back-edge copy, so
    // the load really stays in flight for two steps)
    gxr = gload(gl + fr(t + 2) * gstride4);
    const float act = act_affine(pre, a_scale, a_mul, a_add);
    // forward_statemem (clstm_compute.cc:504-508): c = ci*gi + gf*c_prev with the quad broadcasts folded into
    // the arithmetic (v_mul_f32_dpp, v_fmac_f32_dpp); c_prev = 0 at t = 0 makes the second term an exact +0, so
    // no first-step special case (and no loop peeling) is needed
    LSTM_STAMP(3);   // gate nonlinearity
    const float cig = mul_quad_bcast<0>(act, quad_bcast<3>(act));   // gi * ci
    const float c = fmac_quad_bcast<1>(cig, act, c_prev);           // + gf * c_{t-1}
    const float h = mul_quad_bcast<2>(act, tanh_fast(c));           // forward_nonlingate (clstm_compute.cc:530-537): go * tanh(c)
    c_prev = c;
    LSTM_STAMP(4);   // state update + tanh(c)
    *hw = h;
    ka0 = act; ka1 = c; ka2 = h;                      // stored by the next step's flush (later waves)
    if constexpr (EARLY && !STAG) {
      // store FROM the pinned registers: a VMEM store reads its data late, and a copy of the value in a register
      // that the next step's LDS reads overwrite would put s_waitcnt vmcnt(0) at the top of every step
      OPAQUE(ka0); OPAQUE(ka1); OPAQUE(ka2);
      flush(t, ka0, ka1, ka2, report_tag);
    }
    LSTM_STAMP(5);   // LDS write issued
    if constexpr (POLL) {
      // behind this barrier the workgroup issues, in step t + 1, the loads for iteration t + 3: chunk (t + 3) >> 4 must
      // be there.  Its flag was requested two steps ago; if it is not up yet, hold the barrier.  (Chunk 0 is the
      // workgroup's own: no flag.)
      // (the chunk index is clamped to the line's last chunk FIRST: a line of 13..16 frames has only chunk 0, whose flag nobody
      //  raises -- waiting for "chunk 1 clamped to 0" ran into the watchdog, scripts/gpu_stress_overlap.py)
      const int c = (t + 3) >> 4 < nchunk ? (t + 3) >> 4 : nchunk - 1;
      if (c > 0 && __builtin_expect(wave_uniform(rdy) != a.gepoch, 0)) wait_chunk(c);
      const int cn = (t + 5) >> 4;
      rdy = load_i32_wt(gfl + (cn < nchunk ? cn : nchunk - 1));
    }
    __syncthreads();
    LSTM_STAMP(6);   // barrier
  };
  // staggered: barriers alternate Y_0 X_1 Y_1 X_2 ... ; group A runs [.. Y_t .. X_{t+1}] per step, group B [.. X_{t+1} .. Y_{t+1}]
  // behind one leading Y_0, and A meets B's last Y_T after its loop -- every wave executes the same 2 T + 1 barriers
  if constexpr (STAG && !EARLY) __syncthreads();   // Y_0
  int t = 0;
  for (; t + 1 < T; t += 2) {
    step(t, gxA, rdyA, rdA, wrA, kaA0, kaA1, kaA2, kaB0, kaB1, kaB2);
    step(t + 1, gxB, rdyB, rdB, wrB, kaB0, kaB1, kaB2, kaA0, kaA1, kaA2);
  }
  if (t < T) {
    step(t, gxA, rdyA, rdA, wrA, kaA0, kaA1, kaA2, kaB0, kaB1, kaB2);
    if constexpr (!EARLY || STAG) flush(t, kaA0, kaA1, kaA2, report_tag);
  } else {
    if constexpr (!EARLY || STAG) flush(t - 1, kaB0, kaB1, kaB2, report_tag);
  }
  if constexpr (STAG && EARLY) __syncthreads();    // Y_T
  };
  if constexpr (FUSED) {
    if (wave_uniform(wave) == 3) run(std::true_type{}, std::true_type{}, std::false_type{});
    else if (early) run(std::true_type{}, std::false_type{}, std::false_type{});
    else if (wave_uniform(wave) == (nthreads >> 6) - 1) run(std::false_type{}, std::false_type{}, std::true_type{});
    else run(std::false_type{}, std::false_type{}, std::false_type{});
    // the line is complete: every store of every wave acknowledged, then the final word
    drain_vmem();
    __syncthreads();
    if (tid == 0) store_i32_wt(reinterpret_cast<int*>(a.H + a.prog_off) + ((size_t)dir * a.bs + b) * PROG_STRIDE, a.prog_base + T);
  } else {
    if (early) run(std::true_type{}, std::false_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}, std::false_type{});
  }
#ifdef CLSTM_LSTM_PROF
  if (a.prof && b == 0 && dir == 0 && lane == 0)
    for (int k = 0; k < 12; k++) a.prof[wave * 12 + k] = pacc[k];
#endif
}
template <int NK4, int KU>
__global__ __launch_bounds__(64 * NK4) CLSTM_TWO_WAVES_PER_SIMD void lstm_fwd_kernel(LstmSeqArgs a) {
  lstm_fwd_body<NK4, KU, false>(a, a.order ? a.order[blockIdx.x] : (int)blockIdx.x, blockIdx.y, nullptr);
}

template <int NK4, int KU>
DEVFN void lstm_bwd_body(const LstmSeqArgs& a, const int b, const int dir) {
  constexpr int SLP = 4 * NK4;
  constexpr int QS = SLP + ((NK4 & 1) ? 0 : 4);
  constexpr int DB = 16 * QS;
  float* lds = dyn_smem<float>();  // dbuf[2][DB] + dump word
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x;
  const int no = a.no, nd = a.ndir;
  const int SL = (4 * no + 15) / 16;  // (gate,j) pairs per slice
  const int js = lane & 15;
  const int g = lane & 3, cell = wave * 16 + (lane >> 2);   // quad Q = (lane >> 2) & 3 finishes cell 4 kg + Q
  const bool valid = cell < no;

  // R_g[j][k] for this lane's 4 output cells as (cell 0,1) and (cell 2,3) pairs
  f32x2 wb01[SLP], wb23[SLP];
  {
    const float* rp = a.Rpk + (size_t)dir * 4 * SLP * nthreads + tid;
#pragma unroll
    for (int pp = 0; pp < SLP; pp++) {
      wb01[pp] = (f32x2){rp[(size_t)(0 * SLP + pp) * nthreads], rp[(size_t)(1 * SLP + pp) * nthreads]};
      wb23[pp] = (f32x2){rp[(size_t)(2 * SLP + pp) * nthreads], rp[(size_t)(3 * SLP + pp) * nthreads]};
    }
  }
  for (int i = tid; i < 2 * DB + 4; i += nthreads) lds[i] = 0.0f;

  const int off = a.line_off[b];
  const int T = a.line_off[b + 1] - off;
  if (T <= 0) {   // an empty line is complete at once
    if (a.prog_off >= 0 && tid == 0) {
      store_i32_wt(reinterpret_cast<int*>(a.D + a.prog_off) + ((size_t)dir * a.bs + b) * PROG_STRIDE, a.prog_base);
    }
    return;
  }
  const int pidx = g * no + cell;  // this lane's delta goes to pair (gate g, j = cell)
  const int dslot = (pidx / SL) * QS + (pidx % SL);
  const unsigned gstride4 = (unsigned)nd * 4 * no * 4, cstride4 = (unsigned)nd * no * 4;
  const BufF32 gbuf = make_buf(a.G + (size_t)off * (gstride4 / 4), (size_t)T * gstride4);
  // Progress reporting (a.prog_off >= 0): the LAST lane of the workgroup owns no cell when 4*no is not a multiple of
  // the wave size; it rides the per-step delta store -- same instruction, its own address (lane stride 0) and data --
  // and writes, at iteration `it`, that iterations < it - 3 of this line are complete: every wave has passed the
  // barrier of iteration it-1, i.e. has consumed operands whose loads were issued behind its delta store of
  // iteration it-4, and VMEM operations of a wave complete in order (vmcnt).  Stores are written through.
  const bool report = a.prog_off >= 0;
  const bool tagl = report && tid == nthreads - 1;            // requires cell(tid) >= no (checked by the host)
  const long long prog_rel = a.prog_off + ((long long)dir * a.bs + b) * PROG_STRIDE - (long long)off * (gstride4 / 4);
  const BufF32 dbuf = make_buf(a.D + (size_t)off * (gstride4 / 4), report ? (size_t)(prog_rel + 1) * 4 : (size_t)T * gstride4);
  const BufF32 cbuf = make_buf(a.C + (size_t)off * (cstride4 / 4), (size_t)T * cstride4);
  const BufF32 hbuf = make_buf(a.dH + (size_t)off * (cstride4 / 4), (size_t)T * cstride4);
  const unsigned gl = valid ? ((unsigned)dir * 4 * no + cell * 4 + g) * 4u : BUF_OOB_BASE;
  const unsigned ptag = (unsigned)(prog_rel * 4);                    // the reporting lane's store offset (every step the same)
  const unsigned cl = valid ? ((unsigned)dir * no + cell) * 4u : BUF_OOB_BASE;
  auto fr = [&](int s) -> unsigned {  // own step s -> frame, clamped at the first step
    const int sc = s > 0 ? s : 0;
    return (unsigned)(dir == 0 ? sc : T - 1 - sc);
  };
  const float* rdA = lds + js * QS;           // first step reads buffer 0, writes buffer 1
  const float* rdB = lds + DB + js * QS;
  float* wrA = valid ? lds + DB + dslot : lds + 2 * DB;
  float* wrB = valid ? lds + dslot : lds + 2 * DB;
  const bool gb1 = (g & 2) != 0, gb0 = (g & 1) != 0;

  // Operands (gate activations, dH, c) are fetched two steps ahead into THREE rotating register sets: the
  // loads for step s-2 are issued at the very top of step s into the set step s+1 finished with.  With two
  // sets the reload had to wait for the old value's last use inside the step, hipcc loaded into fresh
  // registers anyway and copied them back at the loop back-edge behind an s_waitcnt vmcnt(1) -- a stall on
  // prefetches issued a few hundred cycles earlier, every second step.  Now whatever the compiler copies at
  // the back-edge was requested at least a full step before.  c_{s-1} is the NEXT set's c (no extra load).
  struct Ops { float act, dh, cc; };
  Ops X0, X1, X2;
  X0.act = buf_load(gbuf, gl + fr(T - 1) * gstride4); X1.act = buf_load(gbuf, gl + fr(T - 2) * gstride4);
  X0.dh = buf_load(hbuf, cl + fr(T - 1) * cstride4);  X1.dh = buf_load(hbuf, cl + fr(T - 2) * cstride4);
  X0.cc = buf_load(cbuf, cl + fr(T - 1) * cstride4);  X1.cc = buf_load(cbuf, T >= 2 ? cl + fr(T - 2) * cstride4 : BUF_OOB);
  X2.act = X2.dh = X2.cc = 0.0f;
  float dc_carry = 0.0f;
  float ka0 = 0.f, ka1 = 0.f, ka2 = 0.f;  // store-data pins (see KEEP_ALIVE)
  __syncthreads();
  // cur: operands of step s; nxt: operands of step s-1 (its c is c_{s-1}); ld: set to refill for step s-2
  auto step = [&](const int s, Ops& cur, const Ops& nxt, Ops& ld, const float* dq, float* dw, float& ka, float& kprev) {
    KEEP_ALIVE(ka);
    ld.act = buf_load(gbuf, gl + fr(s - 2) * gstride4);
    ld.dh = buf_load(hbuf, cl + fr(s - 2) * cstride4);
    ld.cc = buf_load(cbuf, s >= 2 ? cl + fr(s - 2) * cstride4 : BUF_OOB);   // before the first step: 0 = c_{-1}
    // Everything that does not depend on this step's mat-vec is computed BEFORE it (its operands were
    // requested two steps ago): tanh(c), the gate broadcasts, the derivative factor and the second factor
    // of this lane's gate delta.  The dependent tail behind the reduction is then five VALU operations.
    const float actr = cur.act;
    // backward_nonlin0 in place (clstm_compute.cc:231-267): y(1-y) = y - y^2 for SIG, 1 - y^2 for TANH
    const float deriv = fmaf(-actr, actr, g == 3 ? 1.0f : actr);
    const float th = tanh_fast(cur.cc);        // backward_nonlingate recomputes tanh(state) (same form as the forward)
    const float gth = mul_quad_bcast<2>(actr, fmaf(-th, th, 1.0f));   // state.d += (1-t^2) * (go*out.d): (1-t^2)*go
    // backward_statemem (clstm_compute.cc:509-515); c_{-1} = 0 reproduces "gf.d untouched when last < 0"
    float c_m1 = nxt.cc;   // own step s-1's c; the set loaded for "step -1" read out of range = 0
    OPAQUE(c_m1);   // otherwise hipcc branches around the wait for this load for the lanes that do not use it
    // this lane's gate delta is ONE product of two selected factors (selects, no exec-masked branches):
    //   gi.d = dc*ci   gf.d = dc*c_{s-1}   go.d = tanh(c)*out.d   ci.d = dc*gi
    // lanes 0 and 3 of a quad need each other's activation (ci / gi): one mirrored quad permutation
    const float mir = quad_mirror(actr);
    const float mid = gb0 ? c_m1 : th;                    // lane 1: c_{s-1}, lane 2: tanh(c)
    float fb = deriv * (gb0 == gb1 ? mir : mid);
    OPAQUE(fb);     // really before the mat-vec
    // dh_rec[k] = sum_{g,j} R_g[j][k] * delta_g[j](s+1)      [backward_lin1 recurrent half +
    //                                                         backward_stack_delay, :294-304,:398-410]
    f32x2 a01 = splat2(0.0f), a23 = splat2(0.0f);
    // every LDS read of the step is issued before the first FMA, and pinned there: left to itself hipcc sometimes
    // (depending on unrelated code elsewhere in the step) re-uses one register quadruple for all of them and waits
    // for each read in turn -- seven exposed LDS latencies per step (backward 92 -> 104 us)
    float4 dv[NK4];
#pragma unroll
    for (int j = 0; j < NK4; j++) dv[j] = *reinterpret_cast<const float4*>(dq + 4 * j);
    SCHED_FENCE();
    // the PREVIOUS step's delta store (and progress word), under the latency of the LDS reads -- the forward kernel's
    // placement.  (Deferred to the TOP of the next step, in front of the reads, it was slower in round 1: 126 -> 137 us; here
    // the pure kernel is equal, 98 us, and the fused launch, whose recurrence workgroups compete with the polling item
    // workgroups for the memory pipeline, 120.4 -> 118.3 us: 0.2953 -> 0.2913 ms per step in three alternating runs.)
    buf_store_wt(dbuf, s + 1 < T ? (tagl ? ptag : gl + fr(s + 1) * gstride4) : BUF_OOB, kprev);
    SCHED_FENCE();
#pragma unroll
    for (int j = 0; j < NK4; j++) {
      // pairs KU .. 4*NK4-1 of a slice are zero padding (SL = ceil(4 no / 16) <= KU)
      if (4 * j < KU) { a01 = fma2(wb01[4 * j], splat2(dv[j].x), a01); a23 = fma2(wb23[4 * j], splat2(dv[j].x), a23); }
      if (4 * j + 1 < KU) { a01 = fma2(wb01[4 * j + 1], splat2(dv[j].y), a01); a23 = fma2(wb23[4 * j + 1], splat2(dv[j].y), a23); }
      if (4 * j + 2 < KU) { a01 = fma2(wb01[4 * j + 2], splat2(dv[j].z), a01); a23 = fma2(wb23[4 * j + 2], splat2(dv[j].z), a23); }
      if (4 * j + 3 < KU) { a01 = fma2(wb01[4 * j + 3], splat2(dv[j].w), a01); a23 = fma2(wb23[4 * j + 3], splat2(dv[j].w), a23); }
    }
    // reduce-scatter over the row of 16 slices: the quad of cell Q ends with dh_rec of that cell.
    // ror:8 pairs quad Q with Q^2, half_mirror pairs Q with Q^1 (slice j with 3-j, which the
    // quad sum below makes irrelevant).
    // Register slot i of quad Q holds cell i^Q (pack_rb): keep / send sit in fixed registers, no selects.
    const float k0 = a01[0] + row_ror<8>(a23[0]);
    const float k1 = a01[1] + row_ror<8>(a23[1]);
    float k = k0 + row_half_mirror(k1);
    k += quad_xor1(k);
    k += quad_xor2(k);
    const float dh = cur.dh + k;               // out[s].d = delta from above + recurrent delta, clstm.cc:626-628 + :646
    const float dc = dc_carry + gth * dh;
    dc_carry = mul_quad_bcast_old<1>(actr, dc);   // c_{s-1}.d += c.d * gf (gf broadcast folded into the multiply)
    const float delta = (g == 2 ? dh : dc) * fb;
    // (the reporting lane stores the progress word instead: iterations < it - 3 are complete -- stored one step later with
    //  the deltas, which only makes the claim older)
    // (address and data by select: a per-lane stride through v_mad_u32_u24 made hipcc serialise the step's LDS reads)
    const float sdat = tagl ? __builtin_bit_cast(float, a.prog_base + (T - 1 - s) - 3) : delta;
    *dw = delta;
    ka = sdat;
    __syncthreads();
  };
  // 3 operand sets x 2 LDS phases: the pattern repeats every 6 steps
  int s = T - 1;
  for (; s >= 5; s -= 6) {
    step(s, X0, X1, X2, rdA, wrA, ka0, ka2);
    step(s - 1, X1, X2, X0, rdB, wrB, ka1, ka0);
    step(s - 2, X2, X0, X1, rdA, wrA, ka2, ka1);
    step(s - 3, X0, X1, X2, rdB, wrB, ka0, ka2);
    step(s - 4, X1, X2, X0, rdA, wrA, ka1, ka0);
    step(s - 5, X2, X0, X1, rdB, wrB, ka2, ka1);
  }
  if (s >= 0) step(s, X0, X1, X2, rdA, wrA, ka0, ka2);
  if (s >= 1) step(s - 1, X1, X2, X0, rdB, wrB, ka1, ka0);
  if (s >= 2) step(s - 2, X2, X0, X1, rdA, wrA, ka2, ka1);
  if (s >= 3) step(s - 3, X0, X1, X2, rdB, wrB, ka0, ka2);
  if (s >= 4) step(s - 4, X1, X2, X0, rdA, wrA, ka1, ka0);
  {   // the last step's deltas (own step 0): the store data sits in slot (T - 1) mod 3
    const int r = (T - 1) % 3;
    const float last = r == 0 ? ka0 : r == 1 ? ka1 : ka2;
    buf_store_wt(dbuf, tagl ? ptag : gl + fr(0) * gstride4, last);
  }
  if (report) {   // the line is complete: drain this wave's stores, meet, publish "all T iterations"
    drain_vmem();
    __syncthreads();
    if (tagl) buf_store_wt(dbuf, ptag, __builtin_bit_cast(float, a.prog_base + T));
  }
}
template <int NK4, int KU>
__global__ __launch_bounds__(64 * NK4) CLSTM_TWO_WAVES_PER_SIMD void lstm_bwd_kernel(LstmSeqArgs a) {
  lstm_bwd_body<NK4, KU>(a, a.order ? a.order[blockIdx.x] : (int)blockIdx.x, blockIdx.y);
}

}  // namespace clstm
