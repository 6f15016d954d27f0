/*
 * clstm_oracle.c -- CPU restatement of the clstm hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the MI355X path.  It restates, op for op, the
 * reference's Eigen/CPU algorithm for: the on-path clstm_compute.cc operators, the
 * NPLSTM / Softmax / Stacked / Parallel / Reversed layers of clstm.cc, sgd_update,
 * the rinit LCG of batches.cc and ctc.cc (alignment, mktargets, trivial_decode).
 * Every function cites the reference file:line it follows (paths are relative to
 * the upstream tmbdev/clstm tree).
 *
 * It is NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product path (clstm_amd/) never links it.
 *
 * PINNING STATUS
 *   - CTC alignment: pinned against the reference's known-answer tests
 *     (test-ctc.cc:47-74 and :76-109) in tests/test_oracle_ctc.py.
 *   - Per-op and whole-net gradients: pinned by the reference's own finite-difference
 *     methodology (test-cderiv.cc:172-252, test-deriv.cc:100-177), re-hosted in
 *     tests/test_oracle_deriv.py on the double build of this file.
 *   - LSTM / softmax activations: the reference holds no golden activation vectors and
 *     cannot be built here (Eigen, an un-vendored header dependency pinned only by
 *     docker/16.04/Dockerfile:15-16 to RLovelett/eigen tag 3.3-rc1, is absent) ->
 *     "activation parity unpinned by reference outputs"; defined against this
 *     restatement.  Eigen's contraction summation order and its vectorised tanh/exp
 *     polynomials are not reproduced bit-for-bit; the 1e-4 relative bar absorbs that.
 *
 * Build: see oracle/Makefile.  Compiled twice: Float=float (liboracle_f32.so) and
 * Float=double with -DORA_DOUBLE (liboracle_f64.so), mirroring tensor.h:62-66.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* sched_setaffinity (ora_bench_lines_pinned) */
#endif
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORA_DOUBLE
typedef double Float;
#define F_EXP exp
#define F_LOG log
#define F_TANH tanh
#define F_FABS fabs
#define F_FMAX fmax
#else
typedef float Float;
/* In the reference (C++, Float=float) exp/log/tanh/fabs/fmax resolve to the float
 * overloads; here we call the f-suffixed libm entry points explicitly. */
#define F_EXP expf
#define F_LOG logf
#define F_TANH tanhf
#define F_FABS fabsf
#define F_FMAX fmaxf
#endif

#define MAXEXP 30 /* tensor.h:74-76 */

int ora_sizeof_float(void) { return (int)sizeof(Float); }

/* ------------------------------------------------------------------------- */
/* scalar helpers: tensor.h:78-91                                             */
/* ------------------------------------------------------------------------- */
static inline Float limexp(Float x) { /* tensor.h:78-82 */
  /* exp(-MAXEXP) takes an int argument in the reference -> double exp, then narrowed */
  if (x < -MAXEXP) return (Float)exp((double)-MAXEXP);
  if (x > MAXEXP) return (Float)exp((double)MAXEXP);
  return F_EXP(x);
}
static inline Float log_add(Float x, Float y) { /* tensor.h:86-89 */
  if (F_FABS(x - y) > 10) return F_FMAX(x, y);
  return F_LOG(F_EXP(x - y) + 1) + y;
}
static inline Float log_mul(Float x, Float y) { return x + y; } /* tensor.h:91 */
Float ora_limexp(Float x) { return limexp(x); }
Float ora_log_add(Float x, Float y) { return log_add(x, y); }

/* ------------------------------------------------------------------------- */
/* LCG + rinit: batches.cc:11-52                                              */
/* ------------------------------------------------------------------------- */
static double lcg_state = 0.1; /* batches.cc:11 (seed env var, default 0.1) */
void ora_seed(double s) { lcg_state = s; }
double ora_get_seed(void) { return lcg_state; }
static inline double randu(void) { /* batches.cc:13-17 */
  /* volatile temporaries forbid fused multiply-add contraction so the LCG bits
   * are those of the reference's plain multiply followed by add. */
  volatile double prod = 189843.9384938 * lcg_state;
  volatile double sum = prod + 0.328340981343;
  lcg_state = sum;
  lcg_state -= floor(lcg_state);
  return lcg_state;
}
double ora_randu(void) { return randu(); }
static inline double randn(void) { /* batches.cc:19-26 */
  double u1 = randu();
  double u2 = randu();
  double r = -2 * log(u1);
  double theta = 2 * M_PI * u2;
  return r * cos(theta);
}
/* a is column-major n x m: a(i,j) = a[i + n*j]; fill order i outer, j inner. */
void ora_rinit(Float *a, int n, int m, Float s, const char *mode, Float offset) {
  /* batches.cc:32-52 */
  if (!strcmp(mode, "unif")) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++) a[i + (size_t)n * j] = 2 * s * randu() - s + offset;
  } else if (!strcmp(mode, "negbiased")) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++) a[i + (size_t)n * j] = 3 * s * randu() - 2 * s + offset;
  } else if (!strcmp(mode, "pos")) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++) a[i + (size_t)n * j] = s * randu() + offset;
  } else if (!strcmp(mode, "neg")) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++) a[i + (size_t)n * j] = -s * randu() + offset;
  } else if (!strcmp(mode, "normal")) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++) a[i + (size_t)n * j] = s * randn() + offset;
  }
}

/* ------------------------------------------------------------------------- */
/* Sequence: batches.h:45-148.  One block, dims (n=rows, m=batch, 2, N=time);  */
/* step t value plane at data + n*m*(2t), derivative plane at data+n*m*(2t+1). */
/* ------------------------------------------------------------------------- */
typedef struct {
  Float *data;
  int n, m, N;
} Seq;
static inline size_t seq_total(const Seq *s) { return (size_t)s->n * s->m * 2 * s->N; }
static inline Float *seq_v(Seq *s, int t) { return s->data + (size_t)s->n * s->m * (2 * t); }
static inline Float *seq_d(Seq *s, int t) { return s->data + (size_t)s->n * s->m * (2 * t + 1); }
static void seq_free(Seq *s) {
  free(s->data);
  s->data = NULL;
  s->n = s->m = s->N = 0;
}
static void seq_resize(Seq *s, int N, int n, int m) { /* batches.h:115-128 */
  if (N != s->N || n != s->n || m != s->m) {
    free(s->data);
    s->n = n;
    s->m = m;
    s->N = N;
    s->data = (Float *)malloc(seq_total(s) * sizeof(Float) + 1);
  }
  memset(s->data, 0, seq_total(s) * sizeof(Float)); /* reset data, new or reused */
}
static void seq_copy(Seq *dst, const Seq *src) { /* batches.h:133-140 */
  seq_resize(dst, src->N, src->n, src->m);
  memcpy(dst->data, src->data, seq_total(src) * sizeof(Float));
}
static void seq_zero_grad(Seq *s) { /* batches.h:145-147 */
  for (int t = 0; t < s->N; t++) memset(seq_d(s, t), 0, (size_t)s->n * s->m * sizeof(Float));
}

/* ------------------------------------------------------------------------- */
/* clstm_compute.cc operators (CLSTM_ALL_TENSOR branch, SConstruct:44).        */
/* All batches are column-major (rows x bs).  W is n x m column-major with     */
/* bias in column 0 (tensor.h:263-264).                                        */
/* ------------------------------------------------------------------------- */
enum { LIN = 0, SIG = 1, TANH = 2, RELU = 3, LOGMAG = 4 }; /* clstm_compute.h:10-14 */

void ora_forward_nonlin0(Float *y, int len, int nl) { /* clstm_compute.cc:192-229 */
  switch (nl) {
    case LIN: break;
    case SIG:
      for (int i = 0; i < len; i++) y[i] = (Float)1 / ((Float)1 + F_EXP(-y[i]));
      break;
    case TANH:
      for (int i = 0; i < len; i++) y[i] = F_TANH(y[i]);
      break;
    case RELU:
      for (int i = 0; i < len; i++) y[i] = y[i] > 0 ? y[i] : 0;
      break;
    case LOGMAG:
      for (int i = 0; i < len; i++)
        y[i] = F_LOG(F_FABS(y[i]) + (Float)1) * ((y[i] < 0 ? (Float)1 : (Float)0) * (Float)-2 + (Float)1);
      break;
    default: abort();
  }
}
void ora_backward_nonlin0(const Float *yv, Float *yd, int len, int nl) { /* :231-267 */
  switch (nl) {
    case LIN: break;
    case SIG:
      for (int i = 0; i < len; i++) yd[i] = yv[i] * (-yv[i] + (Float)1) * yd[i];
      break;
    case TANH:
      for (int i = 0; i < len; i++) yd[i] = (-yv[i] * yv[i] + (Float)1) * yd[i];
      break;
    case RELU:
      for (int i = 0; i < len; i++) yd[i] = yd[i] * (yv[i] > 0 ? (Float)1 : (Float)0);
      break;
    case LOGMAG:
      for (int i = 0; i < len; i++) yd[i] = yd[i] * F_EXP(-F_FABS(yv[i]));
      break;
    default: abort();
  }
}
/* y = f(x) into a separate buffer; x.d += f'(y) * y.d  (clstm_compute.cc:113-188) */
void ora_forward_nonlin(Float *yv, const Float *xv, int len, int nl) {
  memcpy(yv, xv, len * sizeof(Float));
  ora_forward_nonlin0(yv, len, nl);
}
void ora_backward_nonlin(const Float *yv, const Float *yd, Float *xd, int len, int nl) {
  switch (nl) {
    case LIN:
      for (int i = 0; i < len; i++) xd[i] += yd[i];
      break;
    case SIG:
      for (int i = 0; i < len; i++) xd[i] += yv[i] * (-yv[i] + (Float)1) * yd[i];
      break;
    case TANH:
      for (int i = 0; i < len; i++) xd[i] += (-yv[i] * yv[i] + (Float)1) * yd[i];
      break;
    case RELU:
      for (int i = 0; i < len; i++) xd[i] += yd[i] * (yv[i] > 0 ? (Float)1 : (Float)0);
      break;
    case LOGMAG:
      for (int i = 0; i < len; i++) xd[i] += yd[i] * F_EXP(-F_FABS(yv[i]));
      break;
    default: abort();
  }
}

/* y.v = W[:,1:] . x.v  then  += W[:,0] broadcast  (clstm_compute.cc:275-293) */
void ora_forward_lin1(Float *yv, const Float *W, const Float *xv, int n, int m, int bs) {
  /* Loop order is i-inner (contiguous columns of W) so the compiler vectorises it like Eigen's
   * contraction kernels do; every y(i,b) still accumulates its products in k order. */
  int nx = m - 1;
  for (int b = 0; b < bs; b++) {
    Float *y = yv + (size_t)n * b;
    for (int i = 0; i < n; i++) y[i] = 0;
    for (int k = 0; k < nx; k++) {
      const Float xk = xv[k + (size_t)nx * b];
      const Float *w = W + (size_t)n * (1 + k);
      for (int i = 0; i < n; i++) y[i] += w[i] * xk;
    }
  }
  for (int b = 0; b < bs; b++)
    for (int i = 0; i < n; i++) yv[i + (size_t)n * b] += W[i];
}
/* x.d += W[:,1:]^T y.d ; W.d[:,1:] += y.d x.v^T ; W.d[:,0] += sum_b y.d   (:294-304) */
void ora_backward_lin1(const Float *yd, const Float *W, Float *Wd, const Float *xv, Float *xd,
                       int n, int m, int bs) {
  int nx = m - 1;
  for (int b = 0; b < bs; b++)
    for (int k = 0; k < nx; k++) {
      const Float *w = W + (size_t)n * (1 + k);
      const Float *y = yd + (size_t)n * b;
      Float acc = 0;
#pragma omp simd reduction(+ : acc)
      for (int i = 0; i < n; i++) acc += w[i] * y[i];
      xd[k + (size_t)nx * b] += acc;
    }
  if (bs == 1) { /* rank-1 update, the online-SGD case */
    for (int k = 0; k < nx; k++) {
      const Float xk = xv[k];
      Float *wd = Wd + (size_t)n * (1 + k);
      for (int i = 0; i < n; i++) wd[i] += yd[i] * xk;
    }
  } else {
    for (int k = 0; k < nx; k++)
      for (int i = 0; i < n; i++) {
        Float acc = 0;
        for (int b = 0; b < bs; b++) acc += yd[i + (size_t)n * b] * xv[k + (size_t)nx * b];
        Wd[i + (size_t)n * (1 + k)] += acc;
      }
  }
  for (int i = 0; i < n; i++) {
    Float acc = 0;
    for (int b = 0; b < bs; b++) acc += yd[i + (size_t)n * b];
    Wd[i] += acc;
  }
}
void ora_forward_full1(Float *yv, const Float *W, const Float *xv, int n, int m, int bs, int nl) {
  /* clstm_compute.cc:308-314 */
  ora_forward_lin1(yv, W, xv, n, m, bs);
  ora_forward_nonlin0(yv, n * bs, nl);
}
void ora_backward_full1(const Float *yv, Float *yd, const Float *W, Float *Wd, const Float *xv,
                        Float *xd, int n, int m, int bs, int nl) {
  /* clstm_compute.cc:316-320 */
  ora_backward_nonlin0(yv, yd, n * bs, nl);
  ora_backward_lin1(yd, W, Wd, xv, xd, n, m, bs);
}
/* z.v = colnorm(limexp(W[:,1:].x + W[:,0]))  -- no max subtraction (:324-345) */
void ora_forward_softmax(Float *zv, const Float *W, const Float *xv, int n, int m, int bs) {
  ora_forward_lin1(zv, W, xv, n, m, bs);
  for (int i = 0; i < n * bs; i++) zv[i] = limexp(zv[i]);
  for (int b = 0; b < bs; b++) {
    Float sum = 0;
    for (int i = 0; i < n; i++) sum += zv[i + (size_t)n * b];
    for (int i = 0; i < n; i++) zv[i + (size_t)n * b] = zv[i + (size_t)n * b] / sum;
  }
}
/* x.d = W[:,1:]^T z.d (ASSIGN) ; W.d accumulates as lin1 (:346-356) */
void ora_backward_softmax(const Float *zd, const Float *W, Float *Wd, const Float *xv, Float *xd,
                          int n, int m, int bs) {
  int nx = m - 1;
  memset(xd, 0, (size_t)nx * bs * sizeof(Float));
  ora_backward_lin1(zd, W, Wd, xv, xd, n, m, bs);
}
/* z.v = [x.v ; y.v]   (:360-367) */
void ora_forward_stack(Float *zv, const Float *xv, const Float *yv, int nx, int ny, int bs) {
  for (int b = 0; b < bs; b++) {
    memcpy(zv + (size_t)(nx + ny) * b, xv + (size_t)nx * b, nx * sizeof(Float));
    memcpy(zv + (size_t)(nx + ny) * b + nx, yv + (size_t)ny * b, ny * sizeof(Float));
  }
}
/* x.d += z.d[:nx] ; y.d += z.d[nx:]   (:368-373) */
void ora_backward_stack(const Float *zd, Float *xd, Float *yd, int nx, int ny, int bs) {
  for (int b = 0; b < bs; b++) {
    for (int i = 0; i < nx; i++) xd[i + (size_t)nx * b] += zd[i + (size_t)(nx + ny) * b];
    for (int i = 0; i < ny; i++) yd[i + (size_t)ny * b] += zd[nx + i + (size_t)(nx + ny) * b];
  }
}
/* z.v = [x.v ; ylast.v] or zeros when ylast==NULL   (:377-397) */
void ora_forward_stack_delay(Float *zv, const Float *xv, const Float *ylast_v, int nx, int ny,
                             int bs) {
  for (int b = 0; b < bs; b++) {
    memcpy(zv + (size_t)(nx + ny) * b, xv + (size_t)nx * b, nx * sizeof(Float));
    if (ylast_v)
      memcpy(zv + (size_t)(nx + ny) * b + nx, ylast_v + (size_t)ny * b, ny * sizeof(Float));
    else
      memset(zv + (size_t)(nx + ny) * b + nx, 0, ny * sizeof(Float));
  }
}
/* x.d += z.d[:nx] ; ylast.d += z.d[nx:] if last>=0   (:398-410) */
void ora_backward_stack_delay(const Float *zd, Float *xd, Float *ylast_d, int nx, int ny, int bs) {
  for (int b = 0; b < bs; b++) {
    for (int i = 0; i < nx; i++) xd[i + (size_t)nx * b] += zd[i + (size_t)(nx + ny) * b];
    if (ylast_d)
      for (int i = 0; i < ny; i++) ylast_d[i + (size_t)ny * b] += zd[nx + i + (size_t)(nx + ny) * b];
  }
}
/* state.v = ci.v*gi.v (+ gf.v*last.v)   (:504-508) */
void ora_forward_statemem(Float *state_v, const Float *ci_v, const Float *gi_v,
                          const Float *last_v, const Float *gf_v, int len) {
  for (int i = 0; i < len; i++) state_v[i] = ci_v[i] * gi_v[i];
  if (last_v)
    for (int i = 0; i < len; i++) state_v[i] += gf_v[i] * last_v[i];
}
/* (:509-515) */
void ora_backward_statemem(const Float *state_d, const Float *ci_v, Float *ci_d, const Float *gi_v,
                           Float *gi_d, const Float *last_v, Float *last_d, const Float *gf_v,
                           Float *gf_d, int len) {
  if (last_v)
    for (int i = 0; i < len; i++) last_d[i] += state_d[i] * gf_v[i];
  if (last_v)
    for (int i = 0; i < len; i++) gf_d[i] += state_d[i] * last_v[i];
  for (int i = 0; i < len; i++) gi_d[i] += state_d[i] * ci_v[i];
  for (int i = 0; i < len; i++) ci_d[i] += state_d[i] * gi_v[i];
}
/* out.v = f(state.v) * go.v  (heap temp in the reference)   (:519-537) */
void ora_forward_nonlingate(Float *out_v, const Float *state_v, const Float *go_v, int len, int nl) {
  Float *temp = (Float *)calloc(len ? len : 1, sizeof(Float));
  ora_forward_nonlin(temp, state_v, len, nl);
  for (int i = 0; i < len; i++) out_v[i] = temp[i] * go_v[i];
  free(temp);
}
/* t=f(state.v); go.d += t*out.d; temp.d = go.v*out.d; state.d += f'(t)*temp.d (:539-547) */
void ora_backward_nonlingate(const Float *out_d, const Float *state_v, Float *state_d,
                             const Float *go_v, Float *go_d, int len, int nl) {
  Float *temp_v = (Float *)calloc(len ? len : 1, sizeof(Float));
  Float *temp_d = (Float *)calloc(len ? len : 1, sizeof(Float));
  ora_forward_nonlin(temp_v, state_v, len, nl);
  for (int i = 0; i < len; i++) go_d[i] += temp_v[i] * out_d[i];   /* backward_gate :524 */
  for (int i = 0; i < len; i++) temp_d[i] += go_v[i] * out_d[i];   /* backward_gate :525 */
  ora_backward_nonlin(temp_v, temp_d, state_d, len, nl);
  free(temp_v);
  free(temp_d);
}
void ora_clip_gradient(Float *d, int len, Float clip) { /* :553-558 */
  if (clip >= 1e6) return;
  for (int i = 0; i < len; i++) d[i] = d[i] < clip ? d[i] : clip;   /* cwiseMin */
  for (int i = 0; i < len; i++) d[i] = d[i] > -clip ? d[i] : -clip; /* cwiseMax */
}
void ora_sgd_update(Float *v, Float *d, int len, Float lr, Float mom) { /* :560-563 */
  for (int i = 0; i < len; i++) v[i] += d[i] * lr;
  for (int i = 0; i < len; i++) d[i] = d[i] * mom;
}

/* ------------------------------------------------------------------------- */
/* argmax (ties -> last): tensor.h:357-366                                     */
/* ------------------------------------------------------------------------- */
int ora_argmax(const Float *m, int n) {
  int mi = -1;
  Float mv = m[0];
  for (int i = 0; i < n; i++) {
    if (m[i] < mv) continue;
    mi = i;
    mv = m[i];
  }
  return mi;
}

/* ------------------------------------------------------------------------- */
/* CTC: ctc.cc:24-157.  EigenTensor2 is column-major in the reference; the      */
/* element order is irrelevant to the arithmetic, we use row-major (t, s).     */
/* ------------------------------------------------------------------------- */
static void forward_algorithm(Float *lr, const Float *lmatch, int n, int m, double skip) {
  /* ctc.cc:24-40 ; lr,lmatch are n x m row-major */
  Float *v = (Float *)malloc(sizeof(Float) * (m > 0 ? m : 1));
  Float *w = (Float *)malloc(sizeof(Float) * (m > 0 ? m : 1));
  for (int j = 0; j < m; j++) v[j] = skip * j;
  for (int i = 0; i < n; i++) {
    w[0] = skip * i;
    for (int j = 1; j < m; j++) w[j] = v[j - 1];
    for (int j = 0; j < m; j++) {
      Float same = log_mul(v[j], lmatch[(size_t)i * m + j]);
      Float next = log_mul(w[j], lmatch[(size_t)i * m + j]);
      v[j] = log_add(same, next);
    }
    for (int j = 0; j < m; j++) lr[(size_t)i * m + j] = v[j];
  }
  free(v);
  free(w);
}
static void forwardbackward(Float *both, const Float *lmatch, int n, int m) { /* ctc.cc:42-55 */
  size_t sz = (size_t)n * m;
  Float *lr = (Float *)malloc(sizeof(Float) * (sz ? sz : 1));
  Float *rlmatch = (Float *)malloc(sizeof(Float) * (sz ? sz : 1));
  Float *rrl = (Float *)malloc(sizeof(Float) * (sz ? sz : 1));
  forward_algorithm(lr, lmatch, n, m, -5);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++) rlmatch[(size_t)i * m + j] = lmatch[(size_t)(n - i - 1) * m + (m - j - 1)];
  forward_algorithm(rrl, rlmatch, n, m, -5);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++)
      both[(size_t)i * m + j] = lr[(size_t)i * m + j] + rrl[(size_t)(n - i - 1) * m + (m - j - 1)];
  free(lr);
  free(rlmatch);
  free(rrl);
}
/* posteriors (n1 x nc) from outputs (n1 x nc) and targets (n2 x nc); all row-major.
 * ctc.cc:57-112 verbatim, including the double accumulators and Float stores. */
void ora_ctc_align_targets(Float *posteriors, const Float *outputs, const Float *targets, int n1,
                           int n2, int nc) {
  double lo = 1e-5;
  size_t lsz = (size_t)n1 * n2;
  Float *lmatch = (Float *)malloc(sizeof(Float) * (lsz ? lsz : 1));
  Float *out = (Float *)malloc(sizeof(Float) * nc);
  for (int t1 = 0; t1 < n1; t1++) {
    for (int i = 0; i < nc; i++) out[i] = fmax(lo, outputs[(size_t)t1 * nc + i]);
    Float asum = 0; /* asum1(): tensor.h Float accumulator */
    for (int i = 0; i < nc; i++) asum += out[i];
    for (int i = 0; i < nc; i++) out[i] = out[i] / asum;
    for (int t2 = 0; t2 < n2; t2++) {
      double total = 0.0;
      for (int k = 0; k < nc; k++) total += out[k] * targets[(size_t)t2 * nc + k];
      lmatch[(size_t)t1 * n2 + t2] = log(total);
    }
  }
  Float *both = (Float *)malloc(sizeof(Float) * (lsz ? lsz : 1));
  forwardbackward(both, lmatch, n1, n2);
  /* epath = limexp(both - amax2(both)) */
  Float mx = both[0];
  for (size_t i = 0; i < lsz; i++) mx = F_FMAX(mx, both[i]);
  Float *epath = (Float *)malloc(sizeof(Float) * (lsz ? lsz : 1));
  for (size_t i = 0; i < lsz; i++) epath[i] = limexp(both[i] - mx);
  for (int j = 0; j < n2; j++) {
    double total = 0.0;
    for (int i = 0; i < n1; i++) total += epath[(size_t)i * n2 + j];
    total = fmax(1e-9, total);
    for (int i = 0; i < n1; i++) epath[(size_t)i * n2 + j] /= total;
  }
  for (int i = 0; i < n1; i++) {
    for (int j = 0; j < nc; j++) {
      double total = 0.0;
      for (int k = 0; k < n2; k++) {
        double value = epath[(size_t)i * n2 + k] * targets[(size_t)k * nc + j];
        total += value;
      }
      posteriors[(size_t)i * nc + j] = total;
    }
  }
  for (int i = 0; i < n1; i++) {
    double total = 0.0;
    for (int j = 0; j < nc; j++) total += posteriors[(size_t)i * nc + j];
    total = fmax(total, 1e-9);
    for (int j = 0; j < nc; j++) posteriors[(size_t)i * nc + j] /= total;
  }
  free(lmatch);
  free(out);
  free(both);
  free(epath);
}
/* ctc.cc:136-146: Classes overload = one class per target state, one-hot rows */
void ora_ctc_align_classes(Float *posteriors, const Float *outputs, const int *classes, int n1,
                           int n2, int nc) {
  Float *targets = (Float *)calloc((size_t)n2 * nc + 1, sizeof(Float));
  for (int t = 0; t < n2; t++) targets[(size_t)t * nc + classes[t]] = 1.0;
  ora_ctc_align_targets(posteriors, outputs, targets, n1, n2, nc);
  free(targets);
}
/* mktargets: ctc.cc:148-157.  Writes the 2L+1 state classes (blank=0 at even s). */
int ora_mktargets_classes(int *states, const int *transcript, int L) {
  for (int t = 0; t < 2 * L + 1; t++) states[t] = (t % 2 == 1) ? transcript[(t - 1) / 2] : 0;
  return 2 * L + 1;
}
/* trivial_decode: ctc.cc:159-190.  outputs is T x nc row-major (one batch column).
 * Returns the number of classes written to cs (and frame locations to locs if non-NULL). */
int ora_trivial_decode(int *cs, int *locs, const Float *outputs, int N, int nc) {
  int ncs = 0;
  int t = 0;
  float mv = 0;
  int mc = -1;
  int mt = -1;
  while (t < N) {
    int index = ora_argmax(outputs + (size_t)t * nc, nc);
    float v = outputs[(size_t)t * nc + index];
    if (index == 0) {
      if (mc != -1 && mc != 0) {
        cs[ncs] = mc;
        if (locs) locs[ncs] = mt;
        ncs++;
      }
      mv = 0;
      mc = -1;
      mt = -1;
      t++;
      continue;
    }
    if (v > mv) {
      mv = v;
      mc = index;
      mt = t;
    }
    t++;
  }
  return ncs;
}

/* ------------------------------------------------------------------------- */
/* Layers: clstm.cc.  NPLSTM (:546-656), SoftmaxLayer (:391-419),              */
/* Stacked (:421-455), Reversed (:458-478), Parallel (:506-543).               */
/* ------------------------------------------------------------------------- */
enum { WGI = 0, WGF = 1, WGO = 2, WCI = 3 };
typedef struct {
  int ni, no, nf;
  Float *W[4];  /* WGI, WGF, WGO, WCI : no x (nf+1), views into the flat buffer */
  Float *dW[4];
  Seq inputs, outputs;
  Seq source, gi, gf, go, ci, state;
  Seq out; /* backward's copy of outputs, clstm.cc:626-628 */
} Lstm;

static void lstm_forward(Lstm *L) { /* clstm.cc:600-621 */
  int N = L->inputs.N, bs = L->inputs.m;
  int ni = L->ni, no = L->no, nf = L->nf;
  seq_resize(&L->source, N, nf, bs);
  seq_resize(&L->state, N, no, bs);
  seq_resize(&L->gi, N, no, bs);
  seq_resize(&L->go, N, no, bs);
  seq_resize(&L->gf, N, no, bs);
  seq_resize(&L->ci, N, no, bs);
  seq_resize(&L->outputs, N, no, bs);
  for (int t = 0; t < N; t++) {
    ora_forward_stack_delay(seq_v(&L->source, t), seq_v(&L->inputs, t),
                            t > 0 ? seq_v(&L->outputs, t - 1) : NULL, ni, no, bs);
    ora_forward_full1(seq_v(&L->gi, t), L->W[WGI], seq_v(&L->source, t), no, nf + 1, bs, SIG);
    ora_forward_full1(seq_v(&L->gf, t), L->W[WGF], seq_v(&L->source, t), no, nf + 1, bs, SIG);
    ora_forward_full1(seq_v(&L->go, t), L->W[WGO], seq_v(&L->source, t), no, nf + 1, bs, SIG);
    ora_forward_full1(seq_v(&L->ci, t), L->W[WCI], seq_v(&L->source, t), no, nf + 1, bs, TANH);
    ora_forward_statemem(seq_v(&L->state, t), seq_v(&L->ci, t), seq_v(&L->gi, t),
                         t > 0 ? seq_v(&L->state, t - 1) : NULL, seq_v(&L->gf, t), no * bs);
    ora_forward_nonlingate(seq_v(&L->outputs, t), seq_v(&L->state, t), seq_v(&L->go, t), no * bs,
                           TANH);
  }
}

static int g_nan_asserts = 0; /* the reference's live per-step anynan scans, clstm.cc:630-649 */
void ora_set_nan_asserts(int on) { g_nan_asserts = on; }
static int seq_anynan_step(Seq *s, int t) { /* batches.cc:75-93 on one Batch (v and d) */
  size_t len = (size_t)s->n * s->m;
  Float *v = seq_v(s, t), *d = seq_d(s, t);
  for (size_t i = 0; i < len; i++)
    if (isnan((float)v[i])) return 1;
  for (size_t i = 0; i < len; i++)
    if (isnan((float)d[i])) return 1;
  return 0;
}
static int seq_anynan(Seq *s) {
  for (int t = 0; t < s->N; t++)
    if (seq_anynan_step(s, t)) return 1;
  return 0;
}
static void nan_check(int bad) {
  if (bad) {
    fprintf(stderr, "oracle: NaN assert failed\n");
    abort();
  }
}

static void lstm_backward(Lstm *L) { /* clstm.cc:622-653 */
  int N = L->inputs.N, bs = L->inputs.m;
  int ni = L->ni, no = L->no, nf = L->nf;
  /* clearStateDerivs (clstm.cc:188-195): inputs and every enrolled state */
  seq_zero_grad(&L->inputs);
  seq_zero_grad(&L->gi);
  seq_zero_grad(&L->gf);
  seq_zero_grad(&L->go);
  seq_zero_grad(&L->ci);
  seq_zero_grad(&L->state);
  seq_zero_grad(&L->source);
  seq_copy(&L->out, &L->outputs);
  for (int t = N - 1; t >= 0; t--) {
    if (g_nan_asserts) nan_check(seq_anynan_step(&L->source, t));
    ora_backward_nonlingate(seq_d(&L->out, t), seq_v(&L->state, t), seq_d(&L->state, t),
                            seq_v(&L->go, t), seq_d(&L->go, t), no * bs, TANH);
    if (g_nan_asserts) nan_check(seq_anynan_step(&L->source, t));
    ora_backward_statemem(seq_d(&L->state, t), seq_v(&L->ci, t), seq_d(&L->ci, t),
                          seq_v(&L->gi, t), seq_d(&L->gi, t),
                          t > 0 ? seq_v(&L->state, t - 1) : NULL,
                          t > 0 ? seq_d(&L->state, t - 1) : NULL, seq_v(&L->gf, t),
                          seq_d(&L->gf, t), no * bs);
    if (g_nan_asserts) nan_check(seq_anynan_step(&L->source, t));
    ora_backward_full1(seq_v(&L->ci, t), seq_d(&L->ci, t), L->W[WCI], L->dW[WCI],
                       seq_v(&L->source, t), seq_d(&L->source, t), no, nf + 1, bs, TANH);
    if (g_nan_asserts) nan_check(seq_anynan_step(&L->source, t));
    ora_backward_full1(seq_v(&L->go, t), seq_d(&L->go, t), L->W[WGO], L->dW[WGO],
                       seq_v(&L->source, t), seq_d(&L->source, t), no, nf + 1, bs, SIG);
    if (g_nan_asserts) nan_check(seq_anynan_step(&L->source, t));
    ora_backward_full1(seq_v(&L->gf, t), seq_d(&L->gf, t), L->W[WGF], L->dW[WGF],
                       seq_v(&L->source, t), seq_d(&L->source, t), no, nf + 1, bs, SIG);
    if (g_nan_asserts) nan_check(seq_anynan_step(&L->source, t));
    ora_backward_full1(seq_v(&L->gi, t), seq_d(&L->gi, t), L->W[WGI], L->dW[WGI],
                       seq_v(&L->source, t), seq_d(&L->source, t), no, nf + 1, bs, SIG);
    if (g_nan_asserts) {
      nan_check(seq_anynan(&L->out)); /* whole-sequence scan: the O(T^2) term */
      nan_check(seq_anynan_step(&L->source, t));
    }
    ora_backward_stack_delay(seq_d(&L->source, t), seq_d(&L->inputs, t),
                             t > 0 ? seq_d(&L->out, t - 1) : NULL, ni, no, bs);
    if (g_nan_asserts) {
      nan_check(seq_anynan_step(&L->source, t));
      nan_check(seq_anynan_step(&L->inputs, t));
    }
  }
}

typedef struct {
  int ni, no;
  Float *W1, *dW1;
  Seq inputs, outputs;
} Softmax;
static void softmax_forward(Softmax *S) { /* clstm.cc:405-410 */
  seq_resize(&S->outputs, S->inputs.N, S->no, S->inputs.m);
  for (int t = 0; t < S->inputs.N; t++)
    ora_forward_softmax(seq_v(&S->outputs, t), S->W1, seq_v(&S->inputs, t), S->no, S->ni + 1,
                        S->inputs.m);
}
static void softmax_backward(Softmax *S) { /* clstm.cc:411-417 */
  for (int t = S->outputs.N - 1; t >= 0; t--)
    ora_backward_softmax(seq_d(&S->outputs, t), S->W1, S->dW1, seq_v(&S->inputs, t),
                         seq_d(&S->inputs, t), S->no, S->ni + 1, S->inputs.m);
}

/* forward_reverse: y[N-1-i] = x[i], copies v AND d (clstm_compute.cc:414-417) */
static void forward_reverse(Seq *y, Seq *x) {
  size_t len = (size_t)x->n * x->m;
  for (int i = 0; i < x->N; i++) {
    memcpy(seq_v(y, x->N - i - 1), seq_v(x, i), len * sizeof(Float));
    memcpy(seq_d(y, x->N - i - 1), seq_d(x, i), len * sizeof(Float));
  }
}
/* backward_reverse: x[N-1-i].d += y[i].d (clstm_compute.cc:418-421) */
static void backward_reverse(Seq *y, Seq *x) {
  size_t len = (size_t)x->n * x->m;
  for (int i = 0; i < x->N; i++) {
    Float *xd = seq_d(x, x->N - i - 1), *yd = seq_d(y, i);
    for (size_t k = 0; k < len; k++) xd[k] += yd[k];
  }
}

/* ------------------------------------------------------------------------- */
/* 2-D plumbing operators on raw Sequence blocks (feature, batch, {v,d}, time), */
/* batches.h:79-86: element (i, b, p, t) of a (n, m, 2, N) block at               */
/* i + n*(b + m*(p + 2*t)).                                                       */
/* ------------------------------------------------------------------------- */
#define SEQ_AT(ptr, n, m, i, b, p, t) ((ptr)[(size_t)(i) + (size_t)(n) * ((size_t)(b) + (size_t)(m) * ((size_t)(p) + 2 * (size_t)(t)))])
/* forward_btswitch (clstm_compute.cc:425-436): y4.chip(0,2) = x4.chip(0,2).shuffle({0,2,1}).
 * x dims (rows, bs, 2, N) -> y dims (rows, N, 2, bs); chip(0,2) is the VALUE plane, shuffle
 * {0,2,1} makes output axis 1 the input's axis 2 (time) and output axis 2 the input's axis 1
 * (batch): y.v(i, t, b) = x.v(i, b, t).  The derivative plane of y is not touched. */
void ora_forward_btswitch(Float *y, const Float *x, int rows, int bs, int N) {
  for (int t = 0; t < N; t++)
    for (int b = 0; b < bs; b++)
      for (int i = 0; i < rows; i++) SEQ_AT(y, rows, N, i, t, 0, b) = SEQ_AT(x, rows, bs, i, b, 0, t);
}
/* backward_btswitch (:437-447): x4.chip(1,2) += y4.chip(1,2).shuffle({0,2,1}) -- derivative planes */
void ora_backward_btswitch(const Float *y, Float *x, int rows, int bs, int N) {
  for (int t = 0; t < N; t++)
    for (int b = 0; b < bs; b++)
      for (int i = 0; i < rows; i++) SEQ_AT(x, rows, bs, i, b, 1, t) += SEQ_AT(y, rows, N, i, t, 1, b);
}
/* forward_batchstack (:451-475): y (copies*d, bs, 2, N) = 0 (BOTH planes, :464), then for
 * k = -pre..post: source = max(k,0), dest = max(-k,0), crimp = |k|;
 * y[d*(pre+k) .. +d, dest .. dest+bs-crimp, v, :] = x[0..d, source .. source+bs-crimp, v, :]. */
void ora_forward_batchstack(Float *y, const Float *x, int d, int bs, int N, int pre, int post) {
  int copies = pre + post + 1;
  memset(y, 0, sizeof(Float) * (size_t)copies * d * bs * 2 * N);
  for (int k = -pre; k <= post; k++) {
    int source = k > 0 ? k : 0, dest = -k > 0 ? -k : 0, crimp = k < 0 ? -k : k;
    for (int t = 0; t < N; t++)
      for (int j = 0; j < bs - crimp; j++)
        for (int f = 0; f < d; f++)
          SEQ_AT(y, copies * d, bs, d * (pre + k) + f, dest + j, 0, t) = SEQ_AT(x, d, bs, f, source + j, 0, t);
  }
}
/* backward_batchstack (:476-500): x.d slices += y.d slices, same offsets, k ascending; x.d is
 * NOT cleared first (the clearing line :489 is commented out in the reference). */
void ora_backward_batchstack(const Float *y, Float *x, int d, int bs, int N, int pre, int post) {
  int copies = pre + post + 1;
  for (int k = -pre; k <= post; k++) {
    int source = k > 0 ? k : 0, dest = -k > 0 ? -k : 0, crimp = k < 0 ? -k : k;
    for (int t = 0; t < N; t++)
      for (int j = 0; j < bs - crimp; j++)
        for (int f = 0; f < d; f++)
          SEQ_AT(x, d, bs, f, source + j, 1, t) += SEQ_AT(y, copies * d, bs, d * (pre + k) + f, dest + j, 1, t);
  }
}

/* ------------------------------------------------------------------------- */
/* Full<NONLIN> (clstm.cc:354-389: LinearLayer / SigmoidLayer / TanhLayer /    */
/* ReluLayer) and Stacked{Full<NONLIN>, SoftmaxLayer}: a per-frame MLP.        */
/* forward: outputs[t] = forward_full1(W1, inputs[t]) for t ascending (:369-373) */
/* backward: backward_full1 for t DESCENDING (:375-378), W1.d accumulates.      */
/* Sequences as [T][bs][n] arrays of value / derivative planes.                 */
/* ------------------------------------------------------------------------- */
void ora_full_forward(Float *out_v, const Float *W1, const Float *in_v, int T, int no, int ni, int bs, int nl) {
  for (int t = 0; t < T; t++)
    ora_forward_full1(out_v + (size_t)t * no * bs, W1, in_v + (size_t)t * ni * bs, no, ni + 1, bs, nl);
}
void ora_full_backward(const Float *out_v, Float *out_d, const Float *W1, Float *W1d, const Float *in_v,
                       Float *in_d, int T, int no, int ni, int bs, int nl) {
  for (int t = T - 1; t >= 0; t--)
    ora_backward_full1(out_v + (size_t)t * no * bs, out_d + (size_t)t * no * bs, W1, W1d,
                       in_v + (size_t)t * ni * bs, in_d + (size_t)t * ni * bs, no, ni + 1, bs, nl);
}
/* SoftmaxLayer over a sequence (clstm.cc:405-417), same array conventions */
void ora_softmax_seq_forward(Float *out_v, const Float *W1, const Float *in_v, int T, int no, int ni, int bs) {
  for (int t = 0; t < T; t++)
    ora_forward_softmax(out_v + (size_t)t * no * bs, W1, in_v + (size_t)t * ni * bs, no, ni + 1, bs);
}
void ora_softmax_seq_backward(const Float *out_d, const Float *W1, Float *W1d, const Float *in_v, Float *in_d,
                              int T, int no, int ni, int bs) {
  for (int t = T - 1; t >= 0; t--)
    ora_backward_softmax(out_d + (size_t)t * no * bs, W1, W1d, in_v + (size_t)t * ni * bs,
                         in_d + (size_t)t * ni * bs, no, ni + 1, bs);
}

/* One Parallel{ NPLSTM, Reversed{NPLSTM} } block (clstm_prefab.cc:52-68). */
typedef struct {
  Seq inputs, outputs;         /* Parallel ports */
  Lstm fwd;                    /* sub[0] */
  Seq rev_inputs, rev_outputs; /* Reversed ports (sub[1]) */
  Lstm rev;                    /* Reversed.sub[0] */
} BiLayer;

static void bilayer_forward(BiLayer *B) { /* Parallel::forward clstm.cc:513-528 */
  int N = B->inputs.N;
  seq_copy(&B->fwd.inputs, &B->inputs);
  lstm_forward(&B->fwd);
  seq_copy(&B->rev_inputs, &B->inputs);
  /* Reversed::forward clstm.cc:461-469 */
  seq_resize(&B->rev.inputs, B->rev_inputs.N, B->rev_inputs.n, B->rev_inputs.m);
  forward_reverse(&B->rev.inputs, &B->rev_inputs);
  lstm_forward(&B->rev);
  seq_resize(&B->rev_outputs, B->rev.outputs.N, B->rev.outputs.n, B->rev.outputs.m);
  forward_reverse(&B->rev_outputs, &B->rev.outputs);
  seq_resize(&B->outputs, N, B->fwd.no + B->rev.no, B->inputs.m);
  for (int t = 0; t < N; t++)
    ora_forward_stack(seq_v(&B->outputs, t), seq_v(&B->fwd.outputs, t), seq_v(&B->rev_outputs, t),
                      B->fwd.no, B->rev.no, B->inputs.m);
}
static void bilayer_backward(BiLayer *B) { /* Parallel::backward clstm.cc:529-542 */
  int N = B->outputs.N;
  seq_zero_grad(&B->fwd.outputs);
  seq_zero_grad(&B->rev_outputs);
  for (int t = N - 1; t >= 0; t--)
    ora_backward_stack(seq_d(&B->outputs, t), seq_d(&B->fwd.outputs, t), seq_d(&B->rev_outputs, t),
                       B->fwd.no, B->rev.no, B->inputs.m);
  lstm_backward(&B->fwd);
  /* Reversed::backward clstm.cc:470-477 */
  seq_zero_grad(&B->rev.outputs);
  backward_reverse(&B->rev_outputs, &B->rev.outputs);
  lstm_backward(&B->rev);
  seq_zero_grad(&B->rev_outputs);
  backward_reverse(&B->rev.inputs, &B->rev_inputs);
  size_t len = (size_t)B->inputs.n * B->inputs.m;
  for (int t = 0; t < N; t++) {
    Float *d = seq_d(&B->inputs, t), *a = seq_d(&B->fwd.inputs, t), *b = seq_d(&B->rev_inputs, t);
    for (size_t k = 0; k < len; k++) d[k] = a[k];
    for (size_t k = 0; k < len; k++) d[k] += b[k];
  }
}

/* ------------------------------------------------------------------------- */
/* Network: Stacked{ BiLayer x nlayers, SoftmaxLayer }  = prefab "bidi" (1) or  */
/* "bidi2" (2), clstm_prefab.cc:52-68, 86-109.  unidirectional=1 gives "lstm1"  */
/* (Stacked{NPLSTM, Softmax}, clstm_prefab.cc:23-33) for the test-lstm task.    */
/* Flat parameter order = walk_params (clstm.cc:59-62): per NPLSTM alphabetical */
/* WCI,WGF,WGI,WGO; fwd before rev; layers in order; softmax W1 last.           */
/* ------------------------------------------------------------------------- */
#define ORA_MAXLAYERS 4
typedef struct {
  int nlayers, ninput, nclasses, unidirectional;
  int nhidden[ORA_MAXLAYERS];
  BiLayer layer[ORA_MAXLAYERS];
  Softmax sm;
  Seq inputs, outputs; /* Stacked ports */
  Float *params, *derivs;
  int nparams;
  Float lr, momentum, gclip;
  Seq aligned;
} OraNet;

static const int flat_order[4] = {WCI, WGF, WGI, WGO}; /* alphabetical std::map order */

static Float *bind_lstm(Lstm *L, Float *p, Float *d, int ni, int no) {
  L->ni = ni;
  L->no = no;
  L->nf = ni + no;
  size_t sz = (size_t)no * (L->nf + 1);
  for (int k = 0; k < 4; k++) {
    L->W[flat_order[k]] = p + k * sz;
    L->dW[flat_order[k]] = d + k * sz;
  }
  return p + 4 * sz;
}
static void init_lstm(Lstm *L, Float scale, const char *mode, Float offset) {
  /* GenericNPLSTM::initialize clstm.cc:587-590: WGI, WGF, WGO, WCI in that order */
  ora_rinit(L->W[WGI], L->no, L->nf + 1, scale, mode, offset);
  ora_rinit(L->W[WGF], L->no, L->nf + 1, scale, mode, offset);
  ora_rinit(L->W[WGO], L->no, L->nf + 1, scale, mode, offset);
  ora_rinit(L->W[WCI], L->no, L->nf + 1, scale, mode, offset);
}

int ora_net_nparams_for(int nlayers, int unidirectional, int ninput, const int *nhidden,
                        int nclasses) {
  int total = 0, ni = ninput;
  for (int l = 0; l < nlayers; l++) {
    int no = nhidden[l];
    int dirs = unidirectional ? 1 : 2;
    total += dirs * 4 * no * (ni + no + 1);
    ni = dirs * no;
  }
  total += nclasses * (ni + 1);
  return total;
}

/* make_net("bidi"/"bidi2"/"lstm1") + initialize (clstm.cc:88-115, 30-36) using the
 * current LCG state; init_scale=0.01, init_mode="negbiased", init_offset=0. */
OraNet *ora_net_create(int nlayers, int unidirectional, int ninput, const int *nhidden,
                       int nclasses, int do_init) {
  OraNet *net = (OraNet *)calloc(1, sizeof(OraNet));
  net->nlayers = nlayers;
  net->unidirectional = unidirectional;
  net->ninput = ninput;
  net->nclasses = nclasses;
  for (int l = 0; l < nlayers; l++) net->nhidden[l] = nhidden[l];
  net->nparams = ora_net_nparams_for(nlayers, unidirectional, ninput, nhidden, nclasses);
  net->params = (Float *)calloc(net->nparams, sizeof(Float));
  net->derivs = (Float *)calloc(net->nparams, sizeof(Float));
  net->lr = 1e-4;
  net->momentum = 0.9;
  net->gclip = 100.0;
  Float *p = net->params;
  int ni = ninput;
  for (int l = 0; l < nlayers; l++) {
    int no = nhidden[l];
    Float *q = bind_lstm(&net->layer[l].fwd, p, net->derivs + (p - net->params), ni, no);
    if (do_init) init_lstm(&net->layer[l].fwd, 0.01, "negbiased", 0.0);
    p = q;
    if (!unidirectional) {
      q = bind_lstm(&net->layer[l].rev, p, net->derivs + (p - net->params), ni, no);
      if (do_init) init_lstm(&net->layer[l].rev, 0.01, "negbiased", 0.0);
      p = q;
    }
    ni = (unidirectional ? 1 : 2) * no;
  }
  net->sm.ni = ni;
  net->sm.no = nclasses;
  net->sm.W1 = p;
  net->sm.dW1 = net->derivs + (p - net->params);
  if (do_init) ora_rinit(net->sm.W1, nclasses, ni + 1, 0.01, "negbiased", 0.0);
  return net;
}
static void lstm_free(Lstm *L) {
  seq_free(&L->inputs); seq_free(&L->outputs); seq_free(&L->source); seq_free(&L->gi);
  seq_free(&L->gf); seq_free(&L->go); seq_free(&L->ci); seq_free(&L->state); seq_free(&L->out);
}
void ora_net_free(OraNet *net) {
  for (int l = 0; l < net->nlayers; l++) {
    BiLayer *B = &net->layer[l];
    seq_free(&B->inputs); seq_free(&B->outputs); seq_free(&B->rev_inputs); seq_free(&B->rev_outputs);
    lstm_free(&B->fwd); lstm_free(&B->rev);
  }
  seq_free(&net->sm.inputs); seq_free(&net->sm.outputs);
  seq_free(&net->inputs); seq_free(&net->outputs); seq_free(&net->aligned);
  free(net->params); free(net->derivs); free(net);
}
int ora_net_nparams(OraNet *net) { return net->nparams; }
void ora_net_get_params(OraNet *net, Float *out) { memcpy(out, net->params, net->nparams * sizeof(Float)); }
void ora_net_set_params(OraNet *net, const Float *in) { memcpy(net->params, in, net->nparams * sizeof(Float)); }
void ora_net_get_derivs(OraNet *net, Float *out) { memcpy(out, net->derivs, net->nparams * sizeof(Float)); }
void ora_net_set_derivs(OraNet *net, const Float *in) { memcpy(net->derivs, in, net->nparams * sizeof(Float)); }
void ora_net_set_lr(OraNet *net, Float lr, Float mom) { net->lr = lr; net->momentum = mom; }

/* set_inputs(Network, TensorMap2): clstm.cc:684-690 generalised to a batch:
 * x is [T][bs][ni] (feature contiguous) -> inputs[t].v(i,b); resize memsets v and d. */
void ora_net_set_inputs(OraNet *net, const Float *x, int T, int bs) {
  seq_resize(&net->inputs, T, net->ninput, bs);
  for (int t = 0; t < T; t++)
    memcpy(seq_v(&net->inputs, t), x + (size_t)t * bs * net->ninput,
           (size_t)bs * net->ninput * sizeof(Float));
}

void ora_net_forward(OraNet *net) { /* Stacked::forward clstm.cc:424-439 */
  Seq *cur = &net->inputs;
  for (int l = 0; l < net->nlayers; l++) {
    BiLayer *B = &net->layer[l];
    if (net->unidirectional) {
      seq_copy(&B->fwd.inputs, cur);
      lstm_forward(&B->fwd);
      cur = &B->fwd.outputs;
    } else {
      seq_copy(&B->inputs, cur);
      bilayer_forward(B);
      cur = &B->outputs;
    }
  }
  seq_copy(&net->sm.inputs, cur);
  softmax_forward(&net->sm);
  seq_copy(&net->outputs, &net->sm.outputs);
}

static void copy_d(Seq *dst, Seq *src) { /* per-step .d copies, clstm.cc:445-453 */
  size_t len = (size_t)dst->n * dst->m;
  for (int t = 0; t < dst->N; t++) memcpy(seq_d(dst, t), seq_d(src, t), len * sizeof(Float));
}
void ora_net_backward(OraNet *net) { /* Stacked::backward clstm.cc:440-454 */
  copy_d(&net->sm.outputs, &net->outputs);
  softmax_backward(&net->sm);
  Seq *upstream = &net->sm.inputs;
  for (int l = net->nlayers - 1; l >= 0; l--) {
    BiLayer *B = &net->layer[l];
    if (net->unidirectional) {
      copy_d(&B->fwd.outputs, upstream);
      lstm_backward(&B->fwd);
      upstream = &B->fwd.inputs;
    } else {
      copy_d(&B->outputs, upstream);
      bilayer_backward(B);
      upstream = &B->inputs;
    }
  }
  copy_d(&net->inputs, upstream);
}

/* outputs as [T][bs][nc] */
void ora_net_get_outputs(OraNet *net, Float *out) {
  size_t len = (size_t)net->outputs.n * net->outputs.m;
  for (int t = 0; t < net->outputs.N; t++) memcpy(out + t * len, seq_v(&net->outputs, t), len * sizeof(Float));
}
/* outputs[t].d = d[t]  ([T][bs][nc]) */
void ora_net_set_output_deltas(OraNet *net, const Float *d) {
  size_t len = (size_t)net->outputs.n * net->outputs.m;
  for (int t = 0; t < net->outputs.N; t++) memcpy(seq_d(&net->outputs, t), d + t * len, len * sizeof(Float));
}
void ora_net_get_output_deltas(OraNet *net, Float *d) {
  size_t len = (size_t)net->outputs.n * net->outputs.m;
  for (int t = 0; t < net->outputs.N; t++) memcpy(d + t * len, seq_d(&net->outputs, t), len * sizeof(Float));
}
void ora_net_get_input_deltas(OraNet *net, Float *d) {
  size_t len = (size_t)net->inputs.n * net->inputs.m;
  for (int t = 0; t < net->inputs.N; t++) memcpy(d + t * len, seq_d(&net->inputs, t), len * sizeof(Float));
}
/* set_targets(net, Sequence): outputs[t].d = targets[t] - outputs[t].v  (clstm.cc:142-150) */
void ora_net_set_targets(OraNet *net, const Float *targets) {
  size_t len = (size_t)net->outputs.n * net->outputs.m;
  for (int t = 0; t < net->outputs.N; t++) {
    Float *d = seq_d(&net->outputs, t), *v = seq_v(&net->outputs, t);
    for (size_t k = 0; k < len; k++) d[k] = targets[t * len + k] - v[k];
  }
}

/* CLSTMOCR::fwdbwd's CTC leg for bs==1 (clstmhl.h:207-212): mktargets, align,
 * outputs[t].d = aligned[t] - outputs[t].v.  aligned_out (T x nc) optional. */
void ora_net_ctc_deltas(OraNet *net, const int *transcript, int L, Float *aligned_out) {
  int T = net->outputs.N, nc = net->nclasses;
  if (net->outputs.m != 1) { fprintf(stderr, "oracle: CTC requires bs==1 (ctc.cc:116)\n"); abort(); }
  int S = 2 * L + 1;
  int *states = (int *)malloc(sizeof(int) * S);
  ora_mktargets_classes(states, transcript, L);
  Float *outs = (Float *)malloc(sizeof(Float) * (size_t)T * nc);
  Float *aligned = (Float *)malloc(sizeof(Float) * (size_t)T * nc);
  for (int t = 0; t < T; t++) memcpy(outs + (size_t)t * nc, seq_v(&net->outputs, t), nc * sizeof(Float));
  ora_ctc_align_classes(aligned, outs, states, T, S, nc);
  for (int t = 0; t < T; t++) {
    Float *d = seq_d(&net->outputs, t), *v = seq_v(&net->outputs, t);
    for (int j = 0; j < nc; j++) d[j] = aligned[(size_t)t * nc + j] - v[j];
  }
  if (aligned_out) memcpy(aligned_out, aligned, sizeof(Float) * (size_t)T * nc);
  free(states); free(outs); free(aligned);
}
int ora_net_decode(OraNet *net, int *cs, int *locs, int batch) { /* trivial_decode on outputs */
  int T = net->outputs.N, nc = net->nclasses, bs = net->outputs.m;
  Float *outs = (Float *)malloc(sizeof(Float) * (size_t)T * nc);
  for (int t = 0; t < T; t++)
    memcpy(outs + (size_t)t * nc, seq_v(&net->outputs, t) + (size_t)nc * batch, nc * sizeof(Float));
  (void)bs;
  int n = ora_trivial_decode(cs, locs, outs, T, nc);
  free(outs);
  return n;
}

/* sgd_update(Network): clstm.cc:201-217.  effective_lr() never divides because the
 * layers shadow nseq/nsteps (clstm.cc:357-358,393-394,553-554 vs clstm.h:118-119), so
 * lr is un-normalised.  Parameter clip +-gclip; the state-derivative clips (:210-213)
 * have no observable effect (state derivs are cleared at the next backward). */
void ora_net_update(OraNet *net) {
  ora_clip_gradient(net->derivs, net->nparams, net->gclip);
  ora_sgd_update(net->params, net->derivs, net->nparams, net->lr, net->momentum);
}
void ora_net_clear_derivs(OraNet *net) { memset(net->derivs, 0, net->nparams * sizeof(Float)); }

/* Internal state access for parity tests.  which: 0=gi 1=gf 2=go 3=ci 4=state 5=outputs(h)
 * 6=source; dir: 0 = forward NPLSTM, 1 = the NPLSTM inside Reversed (its OWN time order,
 * i.e. step t of that LSTM saw input frame T-1-t).  plane 0 = v, 1 = d. Layout [T][bs][n]. */
int ora_net_get_state(OraNet *net, int layer, int dir, int which, int plane, Float *out) {
  Lstm *L = dir ? &net->layer[layer].rev : &net->layer[layer].fwd;
  Seq *s = NULL;
  switch (which) {
    case 0: s = &L->gi; break;
    case 1: s = &L->gf; break;
    case 2: s = &L->go; break;
    case 3: s = &L->ci; break;
    case 4: s = &L->state; break;
    case 5: s = &L->outputs; break;
    case 6: s = &L->source; break;
    default: return -1;
  }
  size_t len = (size_t)s->n * s->m;
  if (out)
    for (int t = 0; t < s->N; t++)
      memcpy(out + t * len, plane ? seq_d(s, t) : seq_v(s, t), len * sizeof(Float));
  return (int)(len * s->N);
}

/* One CLSTMOCR::train step for bs==1 (clstmhl.h:201-223): forward, CTC deltas, backward,
 * decode, update.  x: [T][ni].  Returns decode length; cs receives the decoded classes. */
int ora_net_train_line(OraNet *net, const Float *x, int T, const int *transcript, int L, int *cs,
                       int do_update) {
  ora_net_set_inputs(net, x, T, 1);
  ora_net_forward(net);
  ora_net_ctc_deltas(net, transcript, L, NULL);
  ora_net_backward(net);
  int n = cs ? ora_net_decode(net, cs, NULL, 0) : 0;
  if (do_update) ora_net_update(net);
  return n;
}

/* CPU-baseline helper: fwd+CTC+bwd over nlines independent lines with nthreads host
 * threads (each thread owns a clone of the net; gradients are left in the clones).
 * x: packed [sum T][ni]; offs[nlines+1]; labels packed with loffs[nlines+1].
 * Returns wall seconds.  (OpenMP over lines = the "Eigen/OpenMP" figure of BASELINE.md C2.) */
/* `cpus` (nthreads entries, or NULL): the logical CPU thread i pins itself to -- bench.py passes one CPU per physical core, or
 * all of them.  Every thread creates ITS OWN net (allocated and first touched where it runs), trains one untimed line on it --
 * which sizes every Sequence of the net (seq_resize allocates only when a shape changes; the memset per forward pass is the
 * reference's, batches.h:127, and stays) -- and only then, behind a barrier, the clock starts: the timed loop is the
 * reference's per-line work and nothing else. */
#ifdef _OPENMP
#include <sched.h>
#endif
#ifdef __GLIBC__
#include <malloc.h>
#endif
double ora_bench_lines_pinned(OraNet *net, const Float *x, const int *offs, const int *labels,
                              const int *loffs, int nlines, int nthreads, int reps, const int *cpus) {
  if (nthreads < 1) nthreads = 1;
  double t0 = 0, t1 = 0;
#ifdef __GLIBC__
  /* The per-call temporaries of the restated operators (forward/backward_nonlingate's zeroed temps, the CTC matrices: the
   * reference allocates them per call too) make every thread's malloc arena grow and shrink all the time; glibc gives a
   * shrinking arena's pages back to the kernel above 128 KB of free top and faults them in again at the next call, under the
   * PROCESS-wide mmap lock -- measured on the 128-core host of the GPU box: 9 % parallel efficiency.  Keep freed memory in the
   * arenas for the duration of the baseline (the generous figure the north_star asks for). */
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_MMAP_THRESHOLD, 32 << 20);
  mallopt(M_TOP_PAD, 16 << 20);
#endif
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
  {
    const int tid = omp_get_thread_num();
#ifdef __linux__
    if (cpus) {
      cpu_set_t set;
      CPU_ZERO(&set);
      CPU_SET(cpus[tid], &set);
      sched_setaffinity(0, sizeof set, &set);
    }
#endif
    OraNet *c = ora_net_create(net->nlayers, net->unidirectional, net->ninput, net->nhidden, net->nclasses, 0);
    ora_net_set_params(c, net->params);
    {
      const int i = tid % nlines;
      ora_net_train_line(c, x + (size_t)offs[i] * net->ninput, offs[i + 1] - offs[i], labels + loffs[i],
                         loffs[i + 1] - loffs[i], NULL, 0);
    }
#pragma omp barrier
#pragma omp master
    t0 = omp_get_wtime();
#pragma omp barrier
#pragma omp for schedule(dynamic, 1)
    for (int k = 0; k < nlines * reps; k++) {
      const int i = k % nlines;
      ora_net_train_line(c, x + (size_t)offs[i] * net->ninput, offs[i + 1] - offs[i], labels + loffs[i],
                         loffs[i + 1] - loffs[i], NULL, 0);
    }
#pragma omp master
    t1 = omp_get_wtime();
    ora_net_free(c);
  }
#else
  (void)cpus;
  OraNet *c = ora_net_create(net->nlayers, net->unidirectional, net->ninput, net->nhidden, net->nclasses, 0);
  ora_net_set_params(c, net->params);
  struct timespec a, b;
  ora_net_train_line(c, x, offs[1] - offs[0], labels, loffs[1] - loffs[0], NULL, 0);
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int k = 0; k < nlines * reps; k++) {
    const int i = k % nlines;
    ora_net_train_line(c, x + (size_t)offs[i] * net->ninput, offs[i + 1] - offs[i], labels + loffs[i],
                       loffs[i + 1] - loffs[i], NULL, 0);
  }
  clock_gettime(CLOCK_MONOTONIC, &b);
  t0 = a.tv_sec + 1e-9 * a.tv_nsec; t1 = b.tv_sec + 1e-9 * b.tv_nsec;
  ora_net_free(c);
#endif
  return t1 - t0;
}
double ora_bench_lines(OraNet *net, const Float *x, const int *offs, const int *labels,
                       const int *loffs, int nlines, int nthreads, int reps) {
  return ora_bench_lines_pinned(net, x, offs, labels, loffs, nlines, nthreads, reps, NULL);
}

/* Parity helper for full-size minibatches: the reference's semantics of a minibatch are nlines
 * independent CLSTMOCR::train passes without the update in between (clstmhl.h:201-217), every
 * line's fwd / CTC / bwd accumulating into Params.d.  Here every line runs on its own clone of
 * the net (OpenMP over lines, as ora_bench_lines), its outputs / aligned posteriors / decode are
 * written to the packed result arrays, and the per-line gradients are summed INTO net->derivs in
 * line order afterwards (one float add per line and parameter -- the sequential accumulation of
 * the reference interleaves those adds with the per-step ones; the difference is float rounding
 * of a different summation order).  Lines listed in keep[] leave their clone alive in kept[] so
 * that the caller can read every internal state with ora_net_get_state (free with ora_net_free).
 * outputs / aligned: [sum T][nc] packed; dec_cls: [sum T] packed at offs[i], dec_n[i] its length. */
void ora_minibatch_lines(OraNet *net, const Float *x, const int *offs, const int *labels, const int *loffs,
                         int nlines, int nthreads, Float *outputs, Float *aligned, int *dec_cls, int *dec_n,
                         const int *keep, int nkeep, OraNet **kept) {
  if (nthreads < 1) nthreads = 1;
  OraNet **clones = (OraNet **)calloc(nlines, sizeof(OraNet *));
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
  for (int i = 0; i < nlines; i++) {
    OraNet *c = ora_net_create(net->nlayers, net->unidirectional, net->ninput, net->nhidden, net->nclasses, 0);
    ora_net_set_params(c, net->params);
    const int T = offs[i + 1] - offs[i], nc = net->nclasses;
    ora_net_set_inputs(c, x + (size_t)offs[i] * net->ninput, T, 1);
    ora_net_forward(c);
    if (outputs) ora_net_get_outputs(c, outputs + (size_t)offs[i] * nc);
    if (dec_cls) {
      int *cs = (int *)malloc(sizeof(int) * (T + 1));
      int n = ora_net_decode(c, cs, NULL, 0);
      memcpy(dec_cls + offs[i], cs, sizeof(int) * n);
      dec_n[i] = n;
      free(cs);
    }
    ora_net_ctc_deltas(c, labels + loffs[i], loffs[i + 1] - loffs[i], aligned ? aligned + (size_t)offs[i] * nc : NULL);
    ora_net_backward(c);
    clones[i] = c;
  }
  for (int i = 0; i < nlines; i++)
    for (int k = 0; k < net->nparams; k++) net->derivs[k] += clones[i]->derivs[k];
  for (int j = 0; j < nkeep; j++) {
    kept[j] = clones[keep[j]];
    clones[keep[j]] = NULL;
  }
  for (int i = 0; i < nlines; i++)
    if (clones[i]) ora_net_free(clones[i]);
  free(clones);
}
