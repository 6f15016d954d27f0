// clstm_compute_hip.h -- declarations of the HIP-backed operators (what clstm_compute.h's DEFGENERIC list,
// clstm_compute.h:72-103, resolves to when the device is an MI355X).  In a clstm tree the DEFGENERIC macro's GPU branch
// forwards `NAME(args...)` to `NAME(HipDevice*, args...)`; here the generic names are plain inline forwarders.
#pragma once
#ifdef CLSTM_REFERENCE_TREE
#include "clstm_compute.h"      // the reference's own types and nonlinearity codes
#else
#include "mock/clstm_types.h"
namespace ocropus {
constexpr int LIN = 0, SIG = 1, TANH = 2, RELU = 3, LOGMAG = 4;   // clstm_compute.h:10-14
}
#endif

namespace ocropus {
struct HipDevice {};   // tag standing in for Eigen::GpuDevice in the overload set
HipDevice* hip_device();

#define CLSTM_HIP_OPS(X)                                                                             \
  X(forward_nonlin, (Batch & y, Batch & x, int nl), (y, x, nl))                                      \
  X(backward_nonlin, (Batch & y, Batch & x, int nl), (y, x, nl))                                     \
  X(forward_nonlin0, (Batch & y, int nl), (y, nl))                                                   \
  X(backward_nonlin0, (Batch & y, int nl), (y, nl))                                                  \
  X(forward_lin1, (Batch & y, Params & W1, Batch & x), (y, W1, x))                                   \
  X(backward_lin1, (Batch & y, Params & W1, Batch & x), (y, W1, x))                                  \
  X(forward_full1, (Batch & y, Params & W1, Batch & x, int nl), (y, W1, x, nl))                      \
  X(backward_full1, (Batch & y, Params & W1, Batch & x, int nl), (y, W1, x, nl))                     \
  X(forward_stack, (Batch & z, Batch & x, Batch & y), (z, x, y))                                     \
  X(backward_stack, (Batch & z, Batch & x, Batch & y), (z, x, y))                                    \
  X(forward_stack_delay, (Batch & z, Batch & x, Sequence & y, int last), (z, x, y, last))            \
  X(backward_stack_delay, (Batch & z, Batch & x, Sequence & y, int last), (z, x, y, last))           \
  X(forward_reverse, (Sequence & y, Sequence & x), (y, x))                                           \
  X(backward_reverse, (Sequence & y, Sequence & x), (y, x))                                          \
  X(forward_btswitch, (Sequence & y, Sequence & x), (y, x))                                          \
  X(backward_btswitch, (Sequence & y, Sequence & x), (y, x))                                         \
  X(forward_batchstack, (Sequence & y, Sequence & x, int pre, int post), (y, x, pre, post))          \
  X(backward_batchstack, (Sequence & y, Sequence & x, int pre, int post), (y, x, pre, post))         \
  X(forward_softmax, (Batch & z, Params & W1, Batch & x), (z, W1, x))                                \
  X(backward_softmax, (Batch & z, Params & W1, Batch & x), (z, W1, x))                               \
  X(forward_statemem, (Batch & state, Batch & ci, Batch & gi, Sequence & states, int last, Batch & gf), \
    (state, ci, gi, states, last, gf))                                                               \
  X(backward_statemem, (Batch & state, Batch & ci, Batch & gi, Sequence & states, int last, Batch & gf), \
    (state, ci, gi, states, last, gf))                                                               \
  X(forward_nonlingate, (Batch & out, Batch & state, Batch & go, int nl), (out, state, go, nl))      \
  X(backward_nonlingate, (Batch & out, Batch & state, Batch & go, int nl), (out, state, go, nl))     \
  X(clip_gradient, (Batch & x, Float c), (x, c))                                                     \
  X(sgd_update, (Params & p, Float lr, Float mom), (p, lr, mom))

// device overloads (defined in clstm_compute_hip.cc)
#define X(NAME, PARAMS, ARGS) void NAME(HipDevice*, CLSTM_STRIP PARAMS);
#define CLSTM_STRIP(...) __VA_ARGS__
CLSTM_HIP_OPS(X)
#undef X

#ifndef CLSTM_REFERENCE_TREE
// the generic names (DEFGENERIC's job in the reference, clstm_compute.h:50-70)
#define X(NAME, PARAMS, ARGS) inline void NAME PARAMS { NAME(hip_device(), CLSTM_STRIP ARGS); }
CLSTM_HIP_OPS(X)
#undef X
#endif
}  // namespace ocropus
