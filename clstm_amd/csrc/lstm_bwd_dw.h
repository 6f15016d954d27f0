// lstm_bwd_dw.h -- the backward recurrence and its weight-gradient GEMM as ONE launch with two workgroup roles.
//
// The recurrence occupies 2 x lines workgroups (128 at the bench minibatch) of a 256-CU chip for ~95 us while the
// split-K weight-gradient GEMM behind it (gemm_dw.h: its slabs follow the recurrence's iterations) needs the whole
// chip for ~52 us.  Running the GEMM on a second stream costs more than it saves (fork and join through events
// ~37 us, profiles/r02_timeline_overlap.txt).  Here both live in one grid: blocks [0, nrec) walk their line exactly as
// lstm_bwd_kernel does -- they are dispatched first, never wait for anything and run at raised priority -- the block
// behind them is the monitor, and every further block computes ONE (slab, output tile) item of the GEMM as soon as every
// line reports the slab's iterations complete (gemm_dw_body).  (Persistent workers pulling items from per-XCD queues were
// measured slower -- 0.362 vs 0.356 ms per step: a 448-thread slot per worker leaves room for two on an idle CU.)
#pragma once
#include "gemm_dw.h"
#include "lstm_seq.h"

namespace clstm {

// NT: the weight-gradient items' arithmetic (gemm_dw_body): 0 f32 MFMA, 2 / 3 bf16 terms per operand
template <int NK4, int KU, int NT>
__global__ __launch_bounds__(64 * NK4) CLSTM_TWO_WAVES_PER_SIMD void lstm_bwd_dw_kernel(LstmSeqArgs a, GemmDwArgs g, int nrec) {
  // (8 KB more than the items need since their k-tile table shrank to DW_STAB_MAX = 256 entries: with 50 KB a third item workgroup
  //  fits a CU beside the recurrence's, and the 64-line step measured 0.3 us slower for it -- three interleaved runs, 0.2866 vs 0.2861 ms)
  __shared__ __attribute__((aligned(16))) float gsm[dw_img_floats(NT) + 2 * 1024];
  if ((int)blockIdx.x < nrec) {
#ifndef CLSTM_HIP_EMU
    __builtin_amdgcn_s_setprio(3);
#endif
    const long long t0 = g.trace ? wall_clock() : 0;
    const int bl = (int)blockIdx.x % a.bs;
    lstm_bwd_body<NK4, KU>(a, a.order ? a.order[bl] : bl, (int)blockIdx.x / a.bs);
    if (g.done && threadIdx.x == 0) atomic_add_i32(g.done, 1);   // (the body ended with drain + barrier: this line's deltas are in memory)
    if (g.trace && threadIdx.x == 0) { g.trace[blockIdx.x * 4] = t0; g.trace[blockIdx.x * 4 + 2] = wall_clock(); }
  } else {
    if (threadIdx.x >= 256) return;   // the GEMM role is four waves; the others retire (a barrier counts live waves only)
    gemm_dw_body<NT>(g, gsm, blockIdx.x - (unsigned)nrec);   // the monitor, then one item per workgroup in dispatch order
  }
}

}  // namespace clstm
