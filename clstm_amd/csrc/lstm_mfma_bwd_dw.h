// lstm_mfma_bwd_dw.h -- the batched-MFMA backward recurrence and the weight-gradient items as ONE launch with two workgroup roles:
// lstm_bwd_dw.h's arrangement for minibatches of 640 .. 2047 lines, where the recurrence (one 16-line workgroup per CU, lines / 8
// workgroups) leaves CUs idle -- half of the chip at 1024 lines.  Blocks [0, nrec) walk their 16 lines (lstm_mfma_bwd.h, whole-row
// form, REPORT mode: write-through delta rows, per-step progress words), the block behind them is the monitor, every further block
// computes one (time slab, output tile) item of W.d = sum_t [1; x; h]^T delta_t as soon as every line reports the slab's
// iterations complete (gemm_dw.h) -- on the idle CUs while the recurrence runs, on all of them afterwards.  Both roles take their
// LDS from the launch's dynamic allocation (157 KB: the recurrence's; an item uses the first 56 KB of it).
#pragma once
#include "gemm_dw.h"
#include "lstm_mfma_bwd.h"
#ifndef CLSTM_HIP_EMU

namespace clstm {

template <int NO, int NT, int DWT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_bwd_mfma_dw_kernel(LstmMfmaBwdArgs a, GemmDwArgs g, int nrec, int ngroups) {
  static_assert((int)MfmaBwdRowsGeom<NO, NT>::SMEM >= dw_smem_floats(DWT) * 4, "the items' LDS must fit the recurrence's allocation");
  if ((int)blockIdx.x < nrec) {
    __builtin_amdgcn_s_setprio(3);
    lstm_bwd_mfma_rows_body<NO, NT, true>(a, (int)blockIdx.x % ngroups, (int)blockIdx.x / ngroups);
  } else {
    if (threadIdx.x >= 256) return;   // the GEMM role is four waves; the others retire (a barrier counts live waves only)
    gemm_dw_body<DWT>(g, dyn_smem<float>(), blockIdx.x - (unsigned)nrec);
  }
}

}  // namespace clstm
#endif
