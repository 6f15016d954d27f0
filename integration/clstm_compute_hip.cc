// clstm_compute_hip.cc -- the translation unit a clstm maintainer adds next to clstm_compute.cc to run the operators
// on an MI355X: every DEFGENERIC operator (clstm_compute.h:72-103) forwarded, argument for argument, to the C ABI of
// libclstm_hip.so (include/clstm_abi.h).  It replaces clstm_compute_cuda.cc (SConstruct:117-122) -- which re-compiles
// the Eigen expressions of clstm_compute.cc with nvcc -- and needs no Eigen itself.
//
// In a clstm tree: compile with -DCLSTM_REFERENCE_TREE -I<clstm> -I<this repo>/include, link -lclstm_hip, and let
// DEFGENERIC's GPU branch hand out `hip_device()`.  Here (no Eigen in the image) it is compiled against
// integration/mock/clstm_types.h and exercised by integration/test_cderiv_hip.cc, a re-hosting of test-cderiv.cc.
//
// All pointers must be device-accessible (Tensor2::ptr with gpu >= 0 in the reference).  Errors of the C ABI become
// `throw const char*` as everywhere in the reference (SConstruct:43).  This is the LITERAL drop-in level (one launch
// per operator, as the reference executes them); the fast path is the fused clstm_net_* interface.
#include "clstm_compute_hip.h"
#include "clstm_abi.h"

namespace ocropus {
static HipDevice the_device;
HipDevice* hip_device() { return &the_device; }

static void chk(int rc) {
  if (rc) THROW(clstm_last_error());
  // the per-op level is used like the CPU operators (results read on the host right away): finish the launch
  if (clstm_synchronize()) THROW(clstm_last_error());
}
static int len(Batch& b) { return b.rows() * b.cols(); }
static Float* vlast(Sequence& s, int last) { return last >= 0 ? s[last].v.ptr : nullptr; }
static Float* dlast(Sequence& s, int last) { return last >= 0 ? s[last].d.ptr : nullptr; }

// clstm_compute.cc:130-150 / :168-188
void forward_nonlin(HipDevice*, Batch& y, Batch& x, int nl) { chk(clstm_forward_nonlin(y.v.ptr, x.v.ptr, len(y), nl)); }
void backward_nonlin(HipDevice*, Batch& y, Batch& x, int nl) { chk(clstm_backward_nonlin(y.v.ptr, y.d.ptr, x.d.ptr, len(y), nl)); }
// :209-229 / :247-267
void forward_nonlin0(HipDevice*, Batch& y, int nl) { chk(clstm_forward_nonlin0(y.v.ptr, len(y), nl)); }
void backward_nonlin0(HipDevice*, Batch& y, int nl) { chk(clstm_backward_nonlin0(y.v.ptr, y.d.ptr, len(y), nl)); }
// :275-293 / :294-304
void forward_lin1(HipDevice*, Batch& y, Params& W1, Batch& x) {
  chk(clstm_forward_lin1(y.v.ptr, W1.v.ptr, x.v.ptr, W1.v.dimension(0), W1.v.dimension(1), x.cols()));
}
void backward_lin1(HipDevice*, Batch& y, Params& W1, Batch& x) {
  chk(clstm_backward_lin1(y.d.ptr, W1.v.ptr, W1.d.ptr, x.v.ptr, x.d.ptr, W1.v.dimension(0), W1.v.dimension(1), x.cols()));
}
// :308-314 / :316-320
void forward_full1(HipDevice*, Batch& y, Params& W1, Batch& x, int nl) {
  chk(clstm_forward_full1(y.v.ptr, W1.v.ptr, x.v.ptr, W1.v.dimension(0), W1.v.dimension(1), x.cols(), nl));
}
void backward_full1(HipDevice*, Batch& y, Params& W1, Batch& x, int nl) {
  chk(clstm_backward_full1(y.v.ptr, y.d.ptr, W1.v.ptr, W1.d.ptr, x.v.ptr, x.d.ptr, W1.v.dimension(0), W1.v.dimension(1),
                           x.cols(), nl));
}
// :324-345 / :346-356
void forward_softmax(HipDevice*, Batch& z, Params& W1, Batch& x) {
  chk(clstm_forward_softmax(z.v.ptr, W1.v.ptr, x.v.ptr, W1.v.dimension(0), W1.v.dimension(1), x.cols()));
}
void backward_softmax(HipDevice*, Batch& z, Params& W1, Batch& x) {
  chk(clstm_backward_softmax(z.d.ptr, W1.v.ptr, W1.d.ptr, x.v.ptr, x.d.ptr, W1.v.dimension(0), W1.v.dimension(1), x.cols()));
}
// :360-367 / :368-373
void forward_stack(HipDevice*, Batch& z, Batch& x, Batch& y) {
  chk(clstm_forward_stack(z.v.ptr, x.v.ptr, y.v.ptr, x.rows(), y.rows(), x.cols()));
}
void backward_stack(HipDevice*, Batch& z, Batch& x, Batch& y) {
  chk(clstm_backward_stack(z.d.ptr, x.d.ptr, y.d.ptr, x.rows(), y.rows(), x.cols()));
}
// :377-397 / :398-410
void forward_stack_delay(HipDevice*, Batch& z, Batch& x, Sequence& y, int last) {
  chk(clstm_forward_stack_delay(z.v.ptr, x.v.ptr, vlast(y, last), x.rows(), y.rows(), x.cols()));
}
void backward_stack_delay(HipDevice*, Batch& z, Batch& x, Sequence& y, int last) {
  chk(clstm_backward_stack_delay(z.d.ptr, x.d.ptr, dlast(y, last), x.rows(), y.rows(), x.cols()));
}
// :414-417 / :418-421   (raw Sequence blocks, dims (rows, cols, 2, N), batches.h:79-86)
void forward_reverse(HipDevice*, Sequence& y, Sequence& x) { chk(clstm_forward_reverse(y.data, x.data, x.rows(), x.cols(), x.size())); }
void backward_reverse(HipDevice*, Sequence& y, Sequence& x) { chk(clstm_backward_reverse(y.data, x.data, x.rows(), x.cols(), x.size())); }
// :425-436 / :437-447
void forward_btswitch(HipDevice*, Sequence& y, Sequence& x) {
  assert(y.rows() == x.rows() && y.cols() == x.size() && y.size() == x.cols());
  chk(clstm_forward_btswitch(y.data, x.data, x.rows(), x.cols(), x.size()));
}
void backward_btswitch(HipDevice*, Sequence& y, Sequence& x) {
  assert(y.rows() == x.rows() && y.cols() == x.size() && y.size() == x.cols());
  chk(clstm_backward_btswitch(y.data, x.data, x.rows(), x.cols(), x.size()));
}
// :451-475 / :476-500
void forward_batchstack(HipDevice*, Sequence& y, Sequence& x, int pre, int post) {
  assert(y.rows() == (pre + post + 1) * x.rows() && y.cols() == x.cols() && y.size() == x.size());
  chk(clstm_forward_batchstack(y.data, x.data, x.rows(), x.cols(), x.size(), pre, post));
}
void backward_batchstack(HipDevice*, Sequence& y, Sequence& x, int pre, int post) {
  assert(y.rows() == (pre + post + 1) * x.rows() && y.cols() == x.cols() && y.size() == x.size());
  chk(clstm_backward_batchstack(y.data, x.data, x.rows(), x.cols(), x.size(), pre, post));
}
// :504-508 / :509-515
void forward_statemem(HipDevice*, Batch& state, Batch& ci, Batch& gi, Sequence& states, int last, Batch& gf) {
  chk(clstm_forward_statemem(state.v.ptr, ci.v.ptr, gi.v.ptr, vlast(states, last), gf.v.ptr, len(state)));
}
void backward_statemem(HipDevice*, Batch& state, Batch& ci, Batch& gi, Sequence& states, int last, Batch& gf) {
  chk(clstm_backward_statemem(state.d.ptr, ci.v.ptr, ci.d.ptr, gi.v.ptr, gi.d.ptr, vlast(states, last), dlast(states, last),
                              gf.v.ptr, gf.d.ptr, len(state)));
}
// :530-537 / :539-547
void forward_nonlingate(HipDevice*, Batch& out, Batch& state, Batch& go, int nl) {
  chk(clstm_forward_nonlingate(out.v.ptr, state.v.ptr, go.v.ptr, len(out), nl));
}
void backward_nonlingate(HipDevice*, Batch& out, Batch& state, Batch& go, int nl) {
  chk(clstm_backward_nonlingate(out.d.ptr, state.v.ptr, state.d.ptr, go.v.ptr, go.d.ptr, len(out), nl));
}
// :553-558 / :560-563
void clip_gradient(HipDevice*, Batch& x, Float c) { chk(clstm_clip_gradient(x.d.ptr, len(x), c)); }
void sgd_update(HipDevice*, Params& p, Float lr, Float mom) { chk(clstm_sgd_update(p.v.ptr, p.d.ptr, p.rows() * p.cols(), lr, mom)); }
}  // namespace ocropus
