/*
 * clstm_abi.h -- C ABI of libclstm_hip.so: the MI355X (gfx950) backing of clstm's hot path.
 *
 * This is the drop-in boundary for tmbdev/clstm's per-timestep LSTM forward/backward compute
 * and CTC alignment.  Everything is `extern "C"`, plain pointers and sizes; no C++ or torch
 * types.  All `float*` arguments are DEVICE pointers (hipMalloc / torch.cuda memory) unless the
 * parameter name ends in `_h` (HOST pointer).  All matrices are column-major exactly as the
 * reference's Tensor2 (tensor.h:252,288): element (i, b) of a (rows x cols) batch is
 * ptr[i + rows*b]; a Params matrix W is (no x (1+ni)) with the bias in column 0
 * (tensor.h:263-264).  Functions return 0 on success, non-zero on error;
 * clstm_last_error() returns the message (the reference throws `const char*`, SConstruct:43).
 * Kernels are enqueued on the stream set with clstm_set_stream() (default: the null stream);
 * calls are asynchronous unless stated otherwise.  Handles are not thread-safe (the reference
 * is single-threaded and not re-entrant, batches.cc:11, clstm_compute.cc:47).
 *
 * Two granularities, as SURVEY.md §8(b) requires:
 *  (1) per-op entry points, 1:1 with the DEFGENERIC operator list of clstm_compute.h:72-103
 *      that lies on the 1-D BiLSTM+CTC path (what `clstm_compute.cc` would call per timestep);
 *  (2) fused sequence-level entry points (`clstm_net_*`): what the INetwork layers
 *      (clstm.cc NPLSTM/Parallel/Reversed/Stacked/SoftmaxLayer), ctc.cc and sgd_update actually
 *      use for a whole minibatch of text lines.
 */
#ifndef CLSTM_ABI_H_
#define CLSTM_ABI_H_

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* nonlinearity codes: clstm_compute.h:10-14 */
enum { CLSTM_LIN = 0, CLSTM_SIG = 1, CLSTM_TANH = 2, CLSTM_RELU = 3, CLSTM_LOGMAG = 4 };

const char* clstm_last_error(void);
int clstm_abi_version(void);
/* stream for every subsequent call on this host thread (a hipStream_t passed as void*) */
int clstm_set_stream(void* hip_stream);
int clstm_synchronize(void);

/* ------------------------------------------------------------------------------------------
 * (1) per-op entry points.  `n`/`m` are the Params dims (rows, cols incl. bias column),
 *     `bs` the batch columns, `len` = rows*bs element counts.
 * ---------------------------------------------------------------------------------------- */
/* forward_nonlin0 / backward_nonlin0   clstm_compute.cc:209-229 / :247-267 */
int clstm_forward_nonlin0(float* y_v, int len, int nl);
int clstm_backward_nonlin0(const float* y_v, float* y_d, int len, int nl);
/* forward_nonlin / backward_nonlin     clstm_compute.cc:130-150 / :168-188 */
int clstm_forward_nonlin(float* y_v, const float* x_v, int len, int nl);
int clstm_backward_nonlin(const float* y_v, const float* y_d, float* x_d, int len, int nl);
/* forward_lin1 / backward_lin1         clstm_compute.cc:275-293 / :294-304 */
int clstm_forward_lin1(float* y_v, const float* W_v, const float* x_v, int n, int m, int bs);
int clstm_backward_lin1(const float* y_d, const float* W_v, float* W_d, const float* x_v, float* x_d,
                        int n, int m, int bs);
/* forward_full1 / backward_full1       clstm_compute.cc:308-314 / :316-320 */
int clstm_forward_full1(float* y_v, const float* W_v, const float* x_v, int n, int m, int bs, int nl);
int clstm_backward_full1(const float* y_v, float* y_d, const float* W_v, float* W_d, const float* x_v,
                         float* x_d, int n, int m, int bs, int nl);
/* forward_softmax / backward_softmax   clstm_compute.cc:324-345 / :346-356 */
int clstm_forward_softmax(float* z_v, const float* W_v, const float* x_v, int n, int m, int bs);
int clstm_backward_softmax(const float* z_d, const float* W_v, float* W_d, const float* x_v,
                           float* x_d, int n, int m, int bs);
/* forward_stack / backward_stack       clstm_compute.cc:360-367 / :368-373 */
int clstm_forward_stack(float* z_v, const float* x_v, const float* y_v, int nx, int ny, int bs);
int clstm_backward_stack(const float* z_d, float* x_d, float* y_d, int nx, int ny, int bs);
/* forward_stack_delay / backward_stack_delay  clstm_compute.cc:377-397 / :398-410
 * ylast_* = y[last].v / y[last].d, or NULL when last < 0 */
int clstm_forward_stack_delay(float* z_v, const float* x_v, const float* ylast_v, int nx, int ny, int bs);
int clstm_backward_stack_delay(const float* z_d, float* x_d, float* ylast_d, int nx, int ny, int bs);
/* forward_reverse / backward_reverse   clstm_compute.cc:414-417 / :418-421
 * on raw Sequence blocks (batches.h:79-86): dims (rows, bs, 2, N); forward copies v AND d */
int clstm_forward_reverse(float* y_seq, const float* x_seq, int rows, int bs, int N);
int clstm_backward_reverse(const float* y_seq, float* x_seq, int rows, int bs, int N);
/* forward_btswitch / backward_btswitch clstm_compute.cc:425-436 / :437-447 (2-D LSTM plumbing, off the 1-D OCR path):
 * x_seq dims (rows, bs, 2, N) -> y_seq dims (rows, N, 2, bs); forward moves .v, backward accumulates .d into x */
int clstm_forward_btswitch(float* y_seq, const float* x_seq, int rows, int bs, int N);
int clstm_backward_btswitch(const float* y_seq, float* x_seq, int rows, int bs, int N);
/* forward_batchstack / backward_batchstack clstm_compute.cc:451-475 / :476-500: x_seq dims (d, bs, 2, N),
 * y_seq dims ((pre+post+1)*d, bs, 2, N); row block pre+k of batch column b holds x's column b+k (zero outside) */
int clstm_forward_batchstack(float* y_seq, const float* x_seq, int d, int bs, int N, int pre, int post);
int clstm_backward_batchstack(const float* y_seq, float* x_seq, int d, int bs, int N, int pre, int post);
/* forward_statemem / backward_statemem clstm_compute.cc:504-508 / :509-515 (last_* NULL if last<0) */
int clstm_forward_statemem(float* state_v, const float* ci_v, const float* gi_v, const float* last_v,
                           const float* gf_v, int len);
int clstm_backward_statemem(const float* state_d, const float* ci_v, float* ci_d, const float* gi_v,
                            float* gi_d, const float* last_v, float* last_d, const float* gf_v,
                            float* gf_d, int len);
/* forward_nonlingate / backward_nonlingate  clstm_compute.cc:530-537 / :539-547 (no heap temp) */
int clstm_forward_nonlingate(float* out_v, const float* state_v, const float* go_v, int len, int nl);
int clstm_backward_nonlingate(const float* out_d, const float* state_v, float* state_d,
                              const float* go_v, float* go_d, int len, int nl);
/* clip_gradient / sgd_update           clstm_compute.cc:553-558 / :560-563 */
int clstm_clip_gradient(float* d, int len, float clip);
int clstm_sgd_update(float* v, float* d, int len, float lr, float mom);

/* ------------------------------------------------------------------------------------------
 * CTC on packed line batches.  A "line batch" is bs text lines packed frame-major:
 * line b owns frames [line_off[b], line_off[b+1]) of a [N][nc] array (nc contiguous).
 * ---------------------------------------------------------------------------------------- */
/* ctc_align_targets(Sequence&, Sequence&, Classes&)  ctc.cc:136-146 (and :57-134), batched.
 * probs/deltas/aligned: DEVICE [N][nc]; aligned may be NULL.  deltas = aligned - probs
 * (clstmhl.h:211-212).  line_off_h[bs+1], states_h (one class per target state, packed),
 * state_off_h[bs+1]: HOST.  Synchronous with respect to the host arrays (they are copied). */
int clstm_ctc_align_batch(const float* probs, float* deltas, float* aligned, int nc,
                          const int* line_off_h, const int* states_h, const int* state_off_h, int bs);
/* mktargets  ctc.cc:148-157: 2L+1 state classes, blank (0) on even positions.  HOST arrays. */
int clstm_mktargets(int* states_h, const int* transcript_h, int L);
/* trivial_decode  ctc.cc:159-190 (+ argmax tensor.h:357-366), batched.  probs DEVICE [N][nc];
 * classes_h/locs_h HOST [N] (line b's result starts at line_off[b]); counts_h HOST [bs].
 * Blocking. */
int clstm_trivial_decode_batch(const float* probs, int nc, const int* line_off_h, int bs,
                               int* classes_h, int* locs_h, int* counts_h);

/* ------------------------------------------------------------------------------------------
 * (2) fused network: Stacked{ Parallel{NPLSTM, Reversed{NPLSTM}} x nlayers, SoftmaxLayer }
 *     = prefab "bidi" / "bidi2" (clstm_prefab.cc:52-68, 86-109); unidirectional = "lstm1".
 * ---------------------------------------------------------------------------------------- */
typedef struct clstm_net clstm_net;
#define CLSTM_MAX_LAYERS 4
typedef struct {
  int nlayers;                    /* BiLSTM layers (1 = bidi, 2 = bidi2) */
  int unidirectional;             /* 1: Stacked{NPLSTM, Softmax} (lstm1) */
  int ninput;                     /* features per frame (48 for OCR lines) */
  int nhidden[CLSTM_MAX_LAYERS];  /* NPLSTM units per direction */
  int nclasses;                   /* softmax outputs */
} clstm_net_desc;

/* Number of floats of the flat parameter buffer: exactly n_params() (clstm.cc:852-857) in
 * walk_params order (clstm.cc:59-62): per NPLSTM WCI,WGF,WGI,WGO (std::map order), forward
 * before reversed, layers in order, softmax W1 last; each Params block column-major. */
int clstm_net_nparams_for(const clstm_net_desc* desc);
/* params_d / derivs_d / grads_d: caller-owned DEVICE buffers of nparams floats (the analogue of
 * share_params, clstm.cc:859-870), or NULL to let the library allocate.
 *   params = Params.v ; derivs = Params.d (gradient + carried momentum, clstm_compute.cc:560-563);
 *   grads  = this step's fresh minibatch gradient sum (what data-parallel ranks all-reduce). */
int clstm_net_create(clstm_net** out, const clstm_net_desc* desc, float* params_d, float* derivs_d,
                     float* grads_d);
int clstm_net_destroy(clstm_net* net);
/* Precision of the hoisted gate GEMMs (W_x.x for all frames, the weight-gradient and input-delta GEMMs):
 *   0 (default) exact f32 MFMA -- the parity path (1e-4 on activations);
 *   1           bf16 inputs, f32 accumulation (v_mfma_f32_16x16x32_bf16) in those GEMMs; the recurrence, softmax
 *               and CTC stay f32;
 *   2           1 + bf16 MFMA operands (recurrent weights, h, gate deltas) inside the lock-step recurrence of
 *               layers wider than 128 cells (csrc/lstm_wide.h: lstm_xcd_*_bf16) -- what BASELINE config "2 x BiLSTM(512),
 *               bf16 MFMA" names; accumulation, cell state, softmax and CTC stay f32.  Not parity modes.
 * Layers wider than 128 cells run their recurrence as ONE persistent launch per pass in every mode (a workgroup group
 * per XCD, csrc/lstm_wide.h; environment CLSTM_XCD_REC=0: one launch per time step); in modes 0 and 1 with f32 operands
 * and results bit-identical to the per-step kernels. */
int clstm_net_set_gemm_precision(clstm_net* net, int mode);
int clstm_net_nparams(clstm_net* net);
int clstm_net_buffers(clstm_net* net, float** params_d, float** derivs_d, float** grads_d);
/* get_params/set_params/get_derivs/set_derivs (clstm.cc:872-917): HOST buffers, blocking. */
int clstm_net_set_params_h(clstm_net* net, const float* params_h);
int clstm_net_get_params_h(clstm_net* net, float* params_h);
int clstm_net_set_derivs_h(clstm_net* net, const float* derivs_h);
int clstm_net_get_derivs_h(clstm_net* net, float* derivs_h);
int clstm_net_get_grads_h(clstm_net* net, float* grads_h);
/* must be called after writing the params buffer directly on the device */
int clstm_net_params_changed(clstm_net* net);
/* INetwork::setLearningRate (clstm.cc:158-161) + gradient_clip attr (clstm.cc:204) */
int clstm_net_set_learning_rate(clstm_net* net, float lr, float momentum);
int clstm_net_set_gradient_clip(clstm_net* net, float clip);

/* Declare the next minibatch: bs lines of T_h[b] frames each (HOST).  N = sum T. */
int clstm_net_set_batch(clstm_net* net, const int* T_h, int bs);
/* set_inputs (clstm.cc:684-690): x [N][ninput], frame-major, feature contiguous. */
int clstm_net_set_inputs_h(clstm_net* net, const float* x_h);
int clstm_net_set_inputs_d(clstm_net* net, const float* x_d);
/* net->forward() */
int clstm_net_forward(clstm_net* net);
/* net->outputs: DEVICE pointer to [N][nclasses] softmax outputs / their deltas (.d plane) */
int clstm_net_outputs(clstm_net* net, float** probs_d, float** deltas_d);
int clstm_net_get_outputs_h(clstm_net* net, float* probs_h);
int clstm_net_set_output_deltas_h(clstm_net* net, const float* deltas_h);
/* CLSTMOCR::fwdbwd CTC leg (clstmhl.h:207-212): mktargets + ctc_align_targets + deltas for
 * every line of the batch.  labels_h packed transcripts, L_h[bs] lengths (HOST).
 * aligned_h (HOST, [N][nc]) may be NULL. */
int clstm_net_ctc(clstm_net* net, const int* labels_h, const int* L_h, float* aligned_h);
/* net->backward(): accumulates this minibatch's gradient sum into grads (zeroed first). */
int clstm_net_backward(clstm_net* net);
/* input deltas of the first layer are only computed if enabled (nothing on the OCR path reads
 * them; the gradient tests do). */
int clstm_net_enable_input_deltas(clstm_net* net, int on);
int clstm_net_get_input_deltas_h(clstm_net* net, float* dx_h);
/* sgd_update(Network) (clstm.cc:201-217) on the flat buffers:
 *   derivs += grads ; derivs = clip(derivs, +-gradient_clip) ; params += lr*derivs ;
 *   derivs *= momentum.   (lr is NOT normalised, see SURVEY.md §7 "lr normalisation".) */
int clstm_net_update(clstm_net* net);
/* trivial_decode of every line; HOST outputs as clstm_trivial_decode_batch.  Blocking. */
int clstm_net_decode(clstm_net* net, int* classes_h, int* locs_h, int* counts_h);
/* internal NPLSTM state for parity tests: which = 0 gi,1 gf,2 go,3 ci,4 state,5 output h,
 * 6 gate delta gi,7 gf,8 go,9 ci (pre-activation deltas after backward_nonlin0).
 * dir 0 = forward NPLSTM, 1 = the NPLSTM inside Reversed.  out_h: HOST [N][nhidden] in FRAME
 * order (i.e. already un-reversed).  Blocking. */
int clstm_net_get_state_h(clstm_net* net, int layer, int dir, int which, float* out_h);
/* name and device time (ms) of the forward/backward kernels since the last reset -- used by bench.py for the roofline
 * object.  Enable with clstm_net_enable_timing(net, 1).  While it is on, every launch carries a start / stop event pair
 * bound to its own dispatch packet (hipExtLaunchKernel): the time is the kernel's duration by the packet's own time
 * stamps, the figure rocprofv3 --kernel-trace reports; nothing is inserted into the stream.  total_ms sums the launches
 * of every phase of that name, launches counts the phases. */
/* The weight-gradient GEMM of a narrow BiLSTM layer runs beside the backward recurrence (csrc/gemm_dw.h):
 * mode 0: off (GEMM after the recurrence); 1 (default): both as two workgroup roles of ONE launch
 * (csrc/lstm_bwd_dw.h) for batches large enough to profit; 2: the same always (tests); 3: two launches on streams
 * with complementary CU masks (measured slower than mode 0, kept for the record).  Results are the same sums in a
 * different slab order; in modes 1-3 the products run on the bf16 MFMA with both f32 operands split EXACTLY into three
 * bf16 terms (x1 + x2 + x3 = x; the six products of weight >= 2^-16 summed in f32: what is dropped is < 2^-23 |x||y| per
 * product, the size of an f32 multiply's own rounding; experiment options CLSTM_DEBUG="split_terms=2" -- two terms, three
 * products, < 2^-16 -- and "dw_x3=0" -- the f32 MFMA), and the top layer's launch also computes the softmax layer's W.d.  stats: overlapped backward passes so far; slabs that gave up waiting for the recurrence
 * (must stay 0). */
int clstm_net_set_overlap(clstm_net* net, int mode);
/* on != 0: every product of the training step on the f32 MFMA -- the backward products that default to operand-exact split
 * products on the bf16 MFMA (weight gradients, the softmax layer's W.d / x.d) included.  Same as
 * CLSTM_DEBUG="dw_x3=0,gemm_x3=0", per net (bench.py's `strict_f32` leg and --strict-f32). */
int clstm_net_set_strict_f32(clstm_net* net, int on);
int clstm_net_overlap_stats(clstm_net* net, long long* launches, int* timeouts);
int clstm_net_enable_timing(clstm_net* net, int on);
int clstm_net_kernel_time_ms(clstm_net* net, const char* kernel_name, double* total_ms, int* launches);
int clstm_net_reset_timing(clstm_net* net);

/* CLSTMOCR::train (clstmhl.h:201-223) for a whole minibatch in ONE call: set_batch + set_inputs_d + forward +
 * ctc + backward + [all-reduce of grads when a communicator is attached] + update, enqueued on the library
 * stream without any host synchronisation.  T_h[bs], labels_h (packed transcripts), L_h[bs]: HOST;
 * x_d: DEVICE [sum T][ninput].  Decode with clstm_net_decode() afterwards if the caller wants the output. */
int clstm_net_train_step(clstm_net* net, const int* T_h, int bs, const float* x_d, const int* labels_h,
                         const int* L_h);
/* clstm_net_train_step for a loop that knows its NEXT minibatch (a data loader one minibatch ahead: what clstmocrtrain's
 * sample loop, clstmocrtrain.cc:167-172, is once its lines are batched).  Tn_h / bsn / xn_d / labels_n_h / Ln_h describe the
 * minibatch of the next call in the same form (all NULL / 0: exactly clstm_net_train_step).  The front half of that next step --
 * batch geometry, the host half of its alignment, the copy of its frames into the net's input block, of its line offsets and CTC
 * metadata -- is done by THIS call: by extra workgroups of this step's last launch (slab reduction + update), so the next
 * call starts with its forward launch (one launch less per step: 5.7 us of a 64 x 200 step).  The next call must pass the same
 * x pointer and the same T / transcripts (compared by content) to profit; anything else -- another minibatch after all, a
 * communicator of several ranks, clstm_net_set_batch in between -- silently takes the ordinary path.  xn_d must stay unchanged
 * from this call on.  Until the next step the net has no current minibatch: clstm_net_get_outputs_h / decode / ctc / backward /
 * get_state_h refuse.  Results are bit-identical to clstm_net_train_step's. */
int clstm_net_train_step_next(clstm_net* net, const int* T_h, int bs, const float* x_d, const int* labels_h, const int* L_h,
                              const int* Tn_h, int bsn, const float* xn_d, const int* labels_n_h, const int* Ln_h);
/* The same step fed from HOST memory (what clstmocrtrain has after read_png + CenterNormalizer, clstmocrtrain.cc:167-172,
 * extras.cc:227-285): x_h: [sum T][ninput].  Asynchronous: the frames go to the device on a copy stream (a DMA from
 * x_h itself if it is pinned -- clstm_host_alloc -- else through a pinned staging buffer of the library) into one of two
 * device input buffers, overlapping the previous step's kernels; the call returns once everything is enqueued.  x_h may
 * be reused when the call returns if it is pageable, after the NEXT call returns if it is pinned (two buffers in
 * flight). */
int clstm_net_train_step_h(clstm_net* net, const int* T_h, int bs, const float* x_h, const int* labels_h,
                           const int* L_h);
/* pinned host memory for the above */
int clstm_host_alloc(void** p, size_t bytes);
int clstm_host_free(void* p);

/* State externalisation: n_states / get_states / set_states (clstm.cc:762-811; upstream test
 * test-lstm2.cc:79-142).  The reference walks every `Sequence` state of every layer (walk_states,
 * clstm.cc:64-67: per NPLSTM the std::map order ci, gf, gi, go, source, state, then the layer's
 * inputs/outputs are NOT states) and copies the .v planes.  Here: the same order and contents for the
 * current minibatch, per layer: forward NPLSTM then the NPLSTM inside Reversed (in ITS time order, i.e.
 * frame T-1-t at its step t), each state as [T][rows][bs] flattened exactly as Sequence::v would be for a
 * bs=1 line batch; for bs>1 lines are concatenated line after line.  `source` = [x_t ; h_{t-1}] rows
 * (clstm_compute.cc:377-397).  HOST buffers, blocking. */
int clstm_net_n_states(clstm_net* net, long long* n);
int clstm_net_get_states_h(clstm_net* net, float* states_h, long long n);
int clstm_net_set_states_h(clstm_net* net, const float* states_h, long long n);

/* ------------------------------------------------------------------------------------------
 * Data-parallel exchange: every GPU owns an independent shard of the minibatch's lines; the
 * ONE exchange is an all-reduce (sum, fp32) of the flat fresh-gradient buffer before the identical update on
 * every rank.  Reference precedent: share_deltas (clstm.cc:731-744) -- which sums Params.d, momentum
 * included; here only the fresh gradient crosses GPUs (clstm_net_create: grads_d).
 * RCCL (librccl.so.1) is bound at first use; one communicator per process = per GPU (hipSetDevice first).
 * ---------------------------------------------------------------------------------------- */
typedef struct clstm_comm clstm_comm;
#define CLSTM_COMM_ID_BYTES 128
/* rank 0: ncclGetUniqueId into id_h[CLSTM_COMM_ID_BYTES]; ship the bytes to the other ranks by any means */
int clstm_comm_unique_id(char* id_h);
/* collective over all ranks (ncclCommInitRank on the current device) */
int clstm_comm_create(clstm_comm** out, const char* id_h, int rank, int nranks);
int clstm_comm_destroy(clstm_comm* comm);
int clstm_comm_rank(clstm_comm* comm);
int clstm_comm_size(clstm_comm* comm);
/* 1 once the ranks have mapped each other's exchange buffers (HIP IPC; decided collectively at the first clstm_net_train_step with
 * the communicator attached): clstm_net_train_step then runs the one-shot peer-read all-reduce fused into the update kernel
 * (gradient buffers up to 4 MB; environment CLSTM_PEER_ALLREDUCE=0: always ncclAllReduce + update).  0: ncclAllReduce. */
int clstm_comm_peer_active(clstm_comm* comm);
/* in-place sum over ranks of buf_d[0..n) (DEVICE, f32), enqueued on the library stream: no cross-stream
 * event, no host synchronisation. */
int clstm_allreduce_flat(clstm_comm* comm, float* buf_d, long long n);
/* attach (or detach with NULL) a communicator: clstm_net_update() / clstm_net_train_step() then all-reduce
 * the fresh gradient buffer `grads` before derivs += grads. */
int clstm_net_set_comm(clstm_net* net, clstm_comm* comm);
/* Replica consistency (the reference RE-SYNCHRONISES its replicas, distribute_weights / average_weights, clstm.cc:718-729,
 * 746-760; here every rank applies the identical update to the identical all-reduced gradient, so the replicas must stay
 * bit-identical and that is CHECKED): enqueue a parameter checksum, its all-reduce over the attached communicator and the
 * comparison sum == nranks * own.  A mismatch raises a sticky device error (no later update is applied) that the next
 * synchronisation point reports as "replicas diverged ... at training step N".  Called by the library itself every
 * CLSTM_REPLICA_CHECK_EVERY (default 256, 0: never) updates of a net whose communicator has several ranks; collective: every
 * rank must call it at the same point.  No-op without such a communicator. */
int clstm_net_replica_check(clstm_net* net);
/* on = 0: forward passes of this net belong to no training step (CLSTMOCR::predict, the test-set pass of clstmocrtrain,
 * clstmhl.h:225-253): a non-finite logit there does not arm the device NaN / Inf flag that blocks updates (the reference only
 * asserts in backward, clstm.cc:630-649).  Default 1. */
int clstm_net_set_training(clstm_net* net, int on);

/* ------------------------------------------------------------------------------------------
 * diagnostics (used by tests/ to pin the hardware lane layouts the kernels rely on)
 * ---------------------------------------------------------------------------------------- */
/* out: DEVICE [11][64] floats; row k = op k applied to the lane index:
 * 0 quad_xor1, 1 quad_xor2, 2..5 quad_bcast<0..3>, 6 row_ror<1>, 7 row_ror<4>, 8 row_ror<8>,
 * 9 row_half_mirror, 10 wave_shr1 */
int clstm_debug_lane_ops(float* out);
/* C = A.B through the MFMA GEMM used by the hoisted products.  mode 0 "NN": A [R][K] row-major,
 * B [K][Cn] row-major; mode 1 "NT": A [R][K], B given as [Cn][K]; mode 2 "TN": A given as [K][R],
 * B [K][Cn] (split over K into nsplit slabs, reduced deterministically).  C [R][Cn] row-major.
 * Modes 10/11/12: the same three layouts through the bf16-input kernel (gemm_bf16.h). */
/* shader-clock timestamps of the last CTC launch's block 0 (HOST [16]): [0] start, [1..5] after phases A..E,
 * [6..15] sub-phase stamps of the short-line path (see scripts/gpu_ctcprof.py) */
int clstm_debug_ctc_cycles(long long* out_h);
int clstm_debug_gemm(int mode, const float* A, const float* B, float* C, int R, int Cn, int K, int nsplit);
/* how often this process has taken an optional fast path (HOST out): which = 0 persistent per-XCD forward recurrence,
 * 1 persistent backward recurrence, 2 W_x.x from the lower layer's bf16 outputs, 3 x.d from the bf16 delta array,
 * 4 weight-gradient product from contraction-major bf16 operands (LDS transpose reads), 5 the forward half as one
 * launch (W_x producers + recurrence + softmax consumers, lstm_fwd_fused.h).  Tests use it to make sure the
 * path they mean to cover is the one that ran. */
int clstm_debug_path_count(int which, long long* out_h);
/* Experiment switches of the library (clstm_amd/csrc/dbgopt.h: kept kernels against measured losers, e.g. "gemm_stag", "bwd_c32",
 * "rec_x3", "pack_tiles"): tests compare the two sides bit for bit within one process.  name NULL: forget every option.  Whole
 * programs: environment CLSTM_DEBUG="name=value,...".  Not product settings. */
int clstm_debug_set_option(const char* name, int value);
/* (tests) set a device error word: which = 0 the outcome word of the persistent recurrences, 1 the count of weight-gradient
 * items that gave up waiting.  While either is non-zero clstm_net_update() applies nothing; the next synchronisation
 * point (clstm_synchronize, any *_h read-back) reports the error and clears the words. */
int clstm_debug_set_device_error(int which, int value);

#ifdef __cplusplus
}
#endif
#endif /* CLSTM_ABI_H_ */
