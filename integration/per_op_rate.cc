// per_op_rate.cc -- what the LITERAL per-operator drop-in costs: the reference's layer loops (NPLSTM::forward / backward,
// clstm.cc:600-653; Parallel / Reversed / Stacked, clstm.cc:461-560; SoftmaxLayer, clstm.cc:405-417) written over the HIP-backed
// operators of clstm_compute_hip.cc -- one launch per operator and time step, exactly the call sequence a clstm tree makes when
// only clstm_compute_cuda.cc is replaced.  Times one training pass (forward + backward + sgd_update) of a
// BiLSTM(nh) + softmax net over ONE line of T frames and prints ms per line.  No numerics are checked here: every operator is
// pinned on its own (tests/test_ops_parity.py, test_cderiv_hip.cc).
// usage: per_op_rate [T ni nh nc reps]
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include "clstm_compute_hip.h"
#include "../include/clstm_abi.h"

using namespace ocropus;

struct Lstm {   // NPLSTM, clstm.cc:565-598
  Params WGI, WGF, WGO, WCI;
  Sequence source, gi, gf, go, ci, state, out;
  int ni, no;
  void init(int ni_, int no_) {
    ni = ni_; no = no_;
    for (Params* w : {&WGI, &WGF, &WGO, &WCI}) {
      w->setZero(no, 1 + ni + no);
      for (int i = 0; i < no; i++)
        for (int j = 0; j < 1 + ni + no; j++) w->v(i, j) = 0.01f * (float)((i * 7 + j * 3) % 11 - 5);
    }
  }
  void forward(Sequence& in) {   // clstm.cc:600-621
    const int N = in.size(), bs = in.cols();
    source.resize(N, ni + no, bs); gi.resize(N, no, bs); gf.resize(N, no, bs); go.resize(N, no, bs); ci.resize(N, no, bs);
    state.resize(N, no, bs); out.resize(N, no, bs);
    for (int t = 0; t < N; t++) {
      forward_stack_delay(source[t], in[t], out, t - 1);
      forward_full1(gi[t], WGI, source[t], SIG);
      forward_full1(gf[t], WGF, source[t], SIG);
      forward_full1(go[t], WGO, source[t], SIG);
      forward_full1(ci[t], WCI, source[t], TANH);
      forward_statemem(state[t], ci[t], gi[t], state, t - 1, gf[t]);
      forward_nonlingate(out[t], state[t], go[t], TANH);
    }
  }
  void backward(Sequence& in) {  // clstm.cc:622-653
    const int N = in.size();
    for (int t = N - 1; t >= 0; t--) {
      backward_nonlingate(out[t], state[t], go[t], TANH);
      backward_statemem(state[t], ci[t], gi[t], state, t - 1, gf[t]);
      backward_full1(gi[t], WGI, source[t], SIG);
      backward_full1(gf[t], WGF, source[t], SIG);
      backward_full1(go[t], WGO, source[t], SIG);
      backward_full1(ci[t], WCI, source[t], TANH);
      backward_stack_delay(source[t], in[t], out, t - 1);
    }
  }
  void update(Float lr, Float mom) { for (Params* w : {&WGI, &WGF, &WGO, &WCI}) sgd_update(*w, lr, mom); }
};

// The mock tensors live in managed memory so that host code can index them; without XNACK that is host memory the GPU reaches over
// PCIe -- a clstm tree's GPU tensors are device memory (tensor.h:122-170, alloc_gpu).  After the set-up pass every tensor of the net is
// moved into device memory (same layout, the steps re-pointed), so the timed passes measure launches, not PCIe.
#ifdef CLSTM_INTEGRATION_HIP
static Float* device_copy(const Float* src, size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(Float)) != hipSuccess || hipMemcpy(p, src, n * sizeof(Float), hipMemcpyDefault) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(2); }
  return (Float*)p;
}
static void to_device(Sequence& s) {
  if (!s.data) return;
  const int N = s.size(), n = s.rows(), m = s.cols();
  Float* d = device_copy(s.data, (size_t)s.total_size());
  device_free(s.data);
  s.data = d;
  for (int t = 0; t < N; t++) {
    s.steps[t].v.displaceTo(d + (size_t)(n * m) * (2 * t), n, m);
    s.steps[t].d.displaceTo(d + (size_t)(n * m) * (2 * t + 1), n, m);
  }
}
static void to_device(Params& w) {   // (the two device copies are not freed: the process ends right after the measurement)
  const int n = w.rows(), m = w.cols();
  Float* v = device_copy(w.v.ptr, (size_t)n * m);
  Float* d = device_copy(w.d.ptr, (size_t)n * m);
  w.v.displaceTo(v, n, m);
  w.d.displaceTo(d, n, m);
}
#else
static void to_device(Sequence&) {}
static void to_device(Params&) {}
#endif

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 447, ni = argc > 2 ? atoi(argv[2]) : 48, nh = argc > 3 ? atoi(argv[3]) : 50,
            nc = argc > 4 ? atoi(argv[4]) : 83, reps = argc > 5 ? atoi(argv[5]) : 5;
  Lstm fw, bw;
  fw.init(ni, nh); bw.init(ni, nh);
  Params W1;
  W1.setZero(nc, 1 + 2 * nh);
  for (int i = 0; i < nc; i++)
    for (int j = 0; j < 1 + 2 * nh; j++) W1.v(i, j) = 0.01f * (float)((i * 5 + j) % 9 - 4);
  Sequence in, rin, rout, both, z;
  in.resize(T, ni, 1);
  for (int t = 0; t < T; t++)
    for (int i = 0; i < ni; i++) in[t].v(i, 0) = (float)((t * 13 + i * 7) % 17) / 17.0f;
  auto pass = [&]() {
    // Parallel(LSTM, Reversed(LSTM)) -> Stacked softmax (clstm_prefab.cc:52-68)
    fw.forward(in);
    rin.like(in); forward_reverse(rin, in);
    bw.forward(rin);
    rout.like(bw.out); forward_reverse(rout, bw.out);
    both.resize(T, 2 * nh, 1); z.resize(T, nc, 1);
    for (int t = 0; t < T; t++) forward_stack(both[t], fw.out[t], rout[t]);
    for (int t = 0; t < T; t++) forward_softmax(z[t], W1, both[t]);
    // (the CTC alignment between the passes is one fused call in either drop-in; a stand-in delta keeps the backward ops busy)
    for (int t = 0; t < T; t++) backward_softmax(z[t], W1, both[t]);
    for (int t = 0; t < T; t++) backward_stack(both[t], fw.out[t], rout[t]);
    backward_reverse(rout, bw.out);
    bw.backward(rin);
    backward_reverse(rin, in);
    fw.backward(in);
    fw.update(1e-4f, 0.9f); bw.update(1e-4f, 0.9f); sgd_update(W1, 1e-4f, 0.9f);
    clstm_synchronize();
  };
  pass();
  for (Lstm* l : {&fw, &bw}) {
    for (Sequence* q : {&l->source, &l->gi, &l->gf, &l->go, &l->ci, &l->state, &l->out}) to_device(*q);
    for (Params* w : {&l->WGI, &l->WGF, &l->WGO, &l->WCI}) to_device(*w);
  }
  for (Sequence* q : {&in, &rin, &rout, &both, &z}) to_device(*q);
  to_device(W1);
  pass();
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) pass();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  const long launches = (long)T * (7 + 7) * 2 + 4L * T + 4 + 9;
  printf("per-op drop-in: %.2f ms per line (T = %d frames, BiLSTM(%d), %d classes; ~%ld operator launches per line = %.2f us each)\n", ms, T, nh, nc, launches,
         1e3 * ms / launches);
  return 0;
}
