// lstm_bwd_dw.h -- the backward recurrence and its weight-gradient GEMM as ONE launch with two workgroup roles.
//
// The recurrence occupies 2 x lines workgroups (128 at the bench minibatch) of a 256-CU chip for ~95 us while the
// split-K weight-gradient GEMM behind it (gemm_dw.h: its slabs follow the recurrence's iterations) needs the whole
// chip for ~52 us.  Running the GEMM on a second stream costs more than it saves (fork and join through events
// ~37 us, profiles/r02_timeline_overlap.txt).  Here both live in one grid: blocks [0, nrec) walk their line exactly as
// lstm_bwd_kernel does -- they are dispatched first and never wait for anything -- and the blocks behind them are
// persistent GEMM workers: each pulls (slab, output tile) items from its XCD's queue and starts on an item as soon as
// every line reports the slab's iterations complete.  Workers that land on a CU the recurrence occupies (it marks its
// CUs) stay out of its way until all lines are done, then help with the remainder; the recurrence waves also run at
// raised priority.
#pragma once
#include "gemm_dw.h"
#include "lstm_seq.h"

namespace clstm {

template <int NK4, int KU>
__global__ __launch_bounds__(64 * NK4) CLSTM_TWO_WAVES_PER_SIMD void lstm_bwd_dw_kernel(LstmSeqArgs a, GemmDwArgs g, int nrec, int workers) {
  __shared__ __attribute__((aligned(16))) float gsm[DW_SMEM_FLOATS];
  __shared__ int item;
  if ((int)blockIdx.x < nrec) {
    if (threadIdx.x == 0) store_i32_wt(g.cu_busy + hw_cu_slot(), g.prog_base);   // this CU belongs to the recurrence
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
    if (workers && blockIdx.x == (unsigned)nrec) { gemm_dw_monitor(g); return; }
    if (workers) gemm_dw_worker(g, gsm, &item);
    else gemm_dw_body(g, gsm, blockIdx.x - (unsigned)nrec);   // one item per workgroup, in dispatch order
  }
}

}  // namespace clstm
