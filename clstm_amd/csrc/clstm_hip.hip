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
#include "lstm_mfma_bwd_dw.h"
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
#include "runtime.inc"   // errors, buffers, launch helpers, timing
#include "net.inc"       // Layer / Net: the step scheduler
#include "ctc_run.inc"   // CTC / decode launches, host side
}  // namespace clstm

#include "abi.inc"       // extern "C": include/clstm_abi.h
