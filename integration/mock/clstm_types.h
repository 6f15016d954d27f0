// clstm_types.h -- Eigen-free stand-ins for the reference's Tensor2 / Batch / Params / Sequence
// (tensor.h:176-330, batches.h:12-148) with the member names and memory layout the operators of
// clstm_compute.{h,cc} see: column-major (rows x cols) matrices, a Sequence as ONE block of dims
// (rows, cols, 2, N) whose steps are displaced views (v plane at 2t, d plane at 2t+1).
//
// They exist so that integration/clstm_compute_hip.cc -- the translation unit a clstm maintainer would add --
// compiles and RUNS without Eigen (absent in this image): integration/test_cderiv_hip.cc drives it exactly as the
// reference's test-cderiv.cc drives clstm_compute.cc.  In a real clstm tree this header is not needed: the shim
// includes the reference's own clstm_compute.h.
//
// Memory: device_alloc()/device_free() below.  Against the host emulator (CPU CI) that is calloc; against the GPU
// library it is hipMallocManaged (-DCLSTM_INTEGRATION_HIP), so the test's host-side element accessors keep working.
// The fused-level adapter (inetwork/hip_tensor.h) defines CLSTM_TENSOR_HOST_MEMORY: its Sequences never reach a kernel
// (forward() / backward() stage them through the *_h entry points), so they are plain host memory there.
#pragma once
#include <assert.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#ifdef CLSTM_INTEGRATION_HIP
#include <hip/hip_runtime.h>
#endif

namespace ocropus {
typedef float Float;
#ifndef THROW
#define THROW(X) throw(X)
#endif

inline Float* device_alloc(size_t n) {
#if defined(CLSTM_INTEGRATION_HIP) && !defined(CLSTM_TENSOR_HOST_MEMORY)
  void* p = nullptr;
  if (hipMallocManaged(&p, n * sizeof(Float)) != hipSuccess) THROW("hipMallocManaged failed");
  memset(p, 0, n * sizeof(Float));
  return (Float*)p;
#else
  return (Float*)calloc(n ? n : 1, sizeof(Float));
#endif
}
inline void device_free(Float* p) {
#if defined(CLSTM_INTEGRATION_HIP) && !defined(CLSTM_TENSOR_HOST_MEMORY)
  if (p) (void)hipFree(p);
#else
  free(p);
#endif
}

// ---- TensorMap2 stand-in (tensor.h:60-68: Eigen::TensorMap<Eigen::Tensor<Float, 2>>) -----------------------------------
// A non-owning column-major view with exactly the expression forms the reference's high-level code writes on it
// (clstmhl.h:129,211, clstmocrtrain.cc:73: `d() = a.v() - b.v()`, `raw() = -raw() + Float(1)`), element access,
// dimension(), data() and setZero().  Expressions are evaluated element by element on assignment, like Eigen's.
template <class E>
struct TExpr {
  const E& self() const { return static_cast<const E&>(*this); }
};
template <class A>
struct TNeg : TExpr<TNeg<A>> {
  A a;
  explicit TNeg(const A& a_) : a(a_) {}
  int dimension(int i) const { return a.dimension(i); }
  Float at(int i, int j) const { return -a.at(i, j); }
};
template <class A, class B>
struct TSub : TExpr<TSub<A, B>> {
  A a; B b;
  TSub(const A& a_, const B& b_) : a(a_), b(b_) {}
  int dimension(int i) const { return a.dimension(i); }
  Float at(int i, int j) const { return a.at(i, j) - b.at(i, j); }
};
template <class A>
struct TAddScalar : TExpr<TAddScalar<A>> {
  A a; Float s;
  TAddScalar(const A& a_, Float s_) : a(a_), s(s_) {}
  int dimension(int i) const { return a.dimension(i); }
  Float at(int i, int j) const { return a.at(i, j) + s; }
};
struct TensorMap2 : TExpr<TensorMap2> {
  Float* p = nullptr;
  int n = 0, m = 0;
  TensorMap2() {}
  TensorMap2(Float* p_, int n_, int m_) : p(p_), n(n_), m(m_) {}
  int dimension(int i) const { return i == 0 ? n : m; }
  Float* data() const { return p; }
  Float at(int i, int j) const { return p[i + (size_t)n * j]; }
  Float& operator()(int i, int j) const { return p[i + (size_t)n * j]; }
  void setZero() const { if (p) memset(p, 0, sizeof(Float) * n * m); }
  template <class E>
  const TensorMap2& operator=(const TExpr<E>& e) const {
    const E& x = e.self();
    assert(x.dimension(0) == n && x.dimension(1) == m);
    for (int j = 0; j < m; j++)
      for (int i = 0; i < n; i++) p[i + (size_t)n * j] = x.at(i, j);
    return *this;
  }
  const TensorMap2& operator=(const TensorMap2& o) const {   // element-wise copy like Eigen (NOT a rebind)
    if (o.p != p) { assert(o.n == n && o.m == m); memcpy(p, o.p, sizeof(Float) * n * m); }
    return *this;
  }
  TensorMap2(const TensorMap2& o) : TExpr<TensorMap2>(), p(o.p), n(o.n), m(o.m) {}
};
template <class A> TNeg<A> operator-(const TExpr<A>& a) { return TNeg<A>(a.self()); }
template <class A, class B> TSub<A, B> operator-(const TExpr<A>& a, const TExpr<B>& b) { return TSub<A, B>(a.self(), b.self()); }
template <class A> TAddScalar<A> operator+(const TExpr<A>& a, Float s) { return TAddScalar<A>(a.self(), s); }

struct Tensor2 {   // tensor.h:176-330
  int dims[2] = {0, 0};
  Float* ptr = nullptr;
  bool displaced = false;
  Tensor2() {}
  Tensor2(const Tensor2& o) { *this = o; }
  ~Tensor2() { reset(); }
  void operator=(const Tensor2& o) {
    resize(o.dims[0], o.dims[1]);
    if (ptr) memcpy(ptr, o.ptr, sizeof(Float) * dims[0] * dims[1]);
  }
  void displaceTo(Float* p, int n, int m) { reset(); displaced = true; ptr = p; dims[0] = n; dims[1] = m; }
  void reset() {
    if (ptr && !displaced) device_free(ptr);
    displaced = false; ptr = nullptr; dims[0] = dims[1] = 0;
  }
  void resize(int n, int m) {
    if (dims[0] == n && dims[1] == m) return;
    assert(!displaced);
    reset();
    if (n == 0 || m == 0) return;
    dims[0] = n; dims[1] = m;
    ptr = device_alloc((size_t)n * m);
  }
  void setZero() { if (ptr) memset(ptr, 0, sizeof(Float) * dims[0] * dims[1]); }
  void setZero(int n, int m) { resize(n, m); setZero(); }
  void like(const TensorMap2& o) { resize(o.dimension(0), o.dimension(1)); }                 // tensor.h:231-233
  void operator=(const TensorMap2& o) { resize(o.n, o.m); if (ptr) memcpy(ptr, o.p, sizeof(Float) * o.n * o.m); }   // :311-315
  TensorMap2 operator*() const { return TensorMap2(ptr, dims[0], dims[1]); }                // :249-251: *x, x(), x.map()
  TensorMap2 operator()() { return **this; }
  TensorMap2 map() { return **this; }
  Float* data() { return ptr; }
  int total_size() const { return dims[0] * dims[1]; }
  int getGpu() const { return 0; }
  int dimension(int i) const { return dims[i]; }
  int rows() const { return dims[0]; }
  int cols() const { return dims[1]; }
  Float& operator()(int i, int j) { return ptr[i + (size_t)dims[0] * j]; }          // column-major, tensor.h:252,288
  const Float& operator()(int i, int j) const { return ptr[i + (size_t)dims[0] * j]; }
};

struct Batch {     // batches.h:12-24
  Tensor2 v, d;
  virtual ~Batch() {}
  int rows() const { return v.dimension(0); }
  int cols() const { return v.dimension(1); }
  void clear() { v.setZero(); d.setZero(); }            // batches.h:19-22
  void zeroGrad() { d.setZero(rows(), cols()); }
};
struct BatchStorage : Batch {   // batches.h:26-41
  void setZero(int n, int m) { v.setZero(n, m); d.setZero(n, m); }
  void resize(int n, int m) { setZero(n, m); }
};
typedef BatchStorage Params;

struct Sequence {  // batches.h:45-148
  std::vector<BatchStorage> steps;
  Float* data = nullptr;
  int dims[4] = {0, 0, 0, 0};
  Sequence() {}
  Sequence(const Sequence& o) { copy(o); }
  ~Sequence() { clear(); }
  void clear() {
    steps.clear();
    device_free(data);
    data = nullptr;
    dims[0] = dims[1] = dims[2] = dims[3] = 0;
  }
  int size() const { return dims[3]; }
  int rows() const { return dims[0]; }
  int cols() const { return dims[1]; }
  int total_size() const { return dims[0] * dims[1] * dims[2] * dims[3]; }
  void resize(int N, int n, int m) {
    if (N != size() || n != rows() || m != cols()) {
      clear();
      dims[0] = n; dims[1] = m; dims[2] = 2; dims[3] = N;
      data = device_alloc((size_t)total_size());
      steps.resize(N);
      for (int t = 0; t < N; t++) {
        steps[t].v.displaceTo(data + (size_t)(n * m) * (2 * t), n, m);
        steps[t].d.displaceTo(data + (size_t)(n * m) * (2 * t + 1), n, m);
      }
    }
    memset(data, 0, sizeof(Float) * total_size());   // batches.h:127: resize always clears
  }
  void like(const Sequence& o) { resize(o.size(), o.rows(), o.cols()); }
  void copy(const Sequence& o) {
    resize(o.size(), o.rows(), o.cols());
    if (data) memcpy(data, o.data, sizeof(Float) * total_size());
  }
  void operator=(const Sequence& o) { copy(o); }
  Batch& operator[](int i) { return steps[i]; }
  const Batch& operator[](int i) const { return steps[i]; }
  void check() const { assert(dims[3] == 0 ? !data : data != nullptr); }   // batches.h:91-113 (the layout holds by construction here)
  void zero() { for (auto& s : steps) s.clear(); }
  void zeroGrad() { for (auto& s : steps) s.zeroGrad(); }
};
}  // namespace ocropus
