// hip_emu.h -- TEST-ONLY host emulator of the handful of HIP/gfx950 primitives that
// clstm_amd/csrc uses (see devintrin.h).  It lets the CPU test-suite (-m "not gpu") execute
// the real kernel sources -- one FIBER per GPU thread (a workgroup is a set of fibers switched
// cooperatively on one OS thread, so a barrier costs a few context switches instead of a futex
// storm), workgroups spread over the host cores -- to validate indexing, LDS hand-offs and
// barrier placement where no GPU is available.
// MFMA / DPP lane layouts follow /opt/skills/guides/cdna_hip_programming.md §3 and the LLVM
// DppCtrl table; they are assumptions of the emulator, re-checked on hardware by
// tests/test_gpu_intrinsics.py.  Never built into, or loaded by, the product library.
#pragma once
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
extern thread_local uint3_emu threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return 0; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
typedef int hipEvent_t;
#define hipEventDisableTiming 0
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, int) { *e = 1; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
#define hipStreamNonBlocking 0
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, int) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, int) { return 0; }
inline hipError_t hipHostMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return 0; }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local   /* one workgroup at a time per OS thread */
#define __restrict__
#define hipFuncSetAttribute(...) 0
#define DEVFN static inline
#define DEVMFN inline

struct f32x4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
struct float4 { float x, y, z, w; };
typedef float f32x2 __attribute__((vector_size(8)));
typedef unsigned u32x2 __attribute__((vector_size(8)));
inline f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return a * b + c; }
inline f32x2 splat2(float x) { return (f32x2){x, x}; }
inline f32x2 pair_lo(const f32x4& v) { return (f32x2){v.v[0], v.v[1]}; }
inline f32x2 pair_hi(const f32x4& v) { return (f32x2){v.v[2], v.v[3]}; }
inline f32x2 fma2_lo(f32x2 w, f32x2 hp, f32x2 acc) { return (f32x2){fmaf(w[0], hp[0], acc[0]), fmaf(w[1], hp[0], acc[1])}; }
inline f32x2 fma2_hi(f32x2 w, f32x2 hp, f32x2 acc) { return (f32x2){fmaf(w[0], hp[1], acc[0]), fmaf(w[1], hp[1], acc[1])}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

// ---- block / wave runtime -----------------------------------------------------------
struct EmuBar { unsigned n = 0, count = 0, gen = 0; };
struct EmuFiber { void* sp; uint3_emu tid; unsigned wave; bool done; };
struct EmuBlock {
  EmuBar block_bar;
  std::vector<EmuBar> wave_bar;
  std::vector<float> xf;   // [wave][64][16] exchange slots
  char* smem;
  std::vector<EmuFiber> fib;
  unsigned cur = 0, live = 0;
  void* main_sp = nullptr;
  void (*body)(void*) = nullptr;
  void* body_arg = nullptr;
};
extern thread_local EmuBlock* emu_blk;
void emu_yield();                                   // run the next live fiber of this workgroup
void emu_run_block(EmuBlock& blk, dim3 block, void (*body)(void*), void* arg);
void emu_parallel_for(unsigned n, void (*fn)(unsigned, void*), void* arg);
inline void emu_bar_wait(EmuBar& b) {
  const unsigned g = b.gen;
  if (++b.count >= b.n) { b.count = 0; b.gen++; return; }
  while (b.gen == g) emu_yield();
}
inline void __syncthreads() { emu_bar_wait(emu_blk->block_bar); }
inline int emu_lane() { return threadIdx.x & 63; }
inline int emu_wave() { return threadIdx.x >> 6; }
inline void emu_wave_sync() { emu_bar_wait(emu_blk->wave_bar[emu_wave()]); }
inline float* emu_slot(int lane, int k = 0) { return &emu_blk->xf[((size_t)emu_wave() * 64 + lane) * 16 + k]; }

inline float wave_shfl(float x, int src) {
  *emu_slot(emu_lane()) = x;
  emu_wave_sync();
  float r = *emu_slot(src & 63);
  emu_wave_sync();
  return r;
}
inline int wave_shfl_i(int x, int src) {
  float f; memcpy(&f, &x, 4);
  f = wave_shfl(f, src);
  int r; memcpy(&r, &f, 4);
  return r;
}
inline int wave_shfl_xor_i(int x, int m) { return wave_shfl_i(x, emu_lane() ^ m); }
inline unsigned long long wave_ballot(bool p) {
  *emu_slot(emu_lane()) = p ? 1.0f : 0.0f;
  emu_wave_sync();
  unsigned long long m = 0;
  for (int l = 0; l < 64; l++) if (*emu_slot(l) != 0.0f) m |= 1ull << l;
  emu_wave_sync();
  return m;
}
inline int lds_atomic_min(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
inline float wave_shfl_up1(float x) { int l = emu_lane(); return wave_shfl(x, l == 0 ? 0 : l - 1); }
inline float wave_shr1(float x) { return wave_shfl_up1(x); }
inline float add_wave_shr1(float old, float a, float b) { const float s = wave_shfl_up1(a) + b; return emu_lane() == 0 ? old : s; }
inline float quad_xor1(float x) { return wave_shfl(x, emu_lane() ^ 1); }
inline float quad_xor2(float x) { return wave_shfl(x, emu_lane() ^ 2); }
template <int I> inline float quad_bcast(float x) { return wave_shfl(x, (emu_lane() & ~3) | I); }
template <int I> inline float mul_quad_bcast(float x, float y) { return quad_bcast<I>(x) * y; }
template <int I> inline float fmac_quad_bcast(float acc, float x, float y) { return fmaf(quad_bcast<I>(x), y, acc); }
template <int I> inline float mul_quad_bcast_old(float x, float y) { return quad_bcast<I>(x) * y; }
inline float quad_mirror(float x) { return wave_shfl(x, emu_lane() ^ 3); }
template <int N> inline float row_ror(float x) {
  // DPP row_ror:N -- lane i of a 16-lane row receives the value of lane (i+N)%16 of that row
  int l = emu_lane();
  return wave_shfl(x, (l & ~15) | ((l + N) & 15));
}
inline float row_half_mirror(float x) { return wave_shfl(x, emu_lane() ^ 7); }
inline float wave_max(float x) { for (int m = 32; m >= 1; m >>= 1) x = fmaxf(x, wave_shfl(x, emu_lane() ^ m)); return x; }
inline float wave_sum(float x) { for (int m = 32; m >= 1; m >>= 1) x += wave_shfl(x, emu_lane() ^ m); return x; }

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+r][col=l&15]
inline f32x4 mfma16x16x4(float a, float b, f32x4 c) {
  int l = emu_lane();
  *emu_slot(l, 0) = a;
  *emu_slot(l, 1) = b;
  emu_wave_sync();
  f32x4 d = c;
  for (int r = 0; r < 4; r++) {
    int row = (l >> 4) * 4 + r, col = l & 15;
    float acc = c[r];
    for (int k = 0; k < 4; k++) acc = fmaf(*emu_slot(k * 16 + row, 0), *emu_slot(k * 16 + col, 1), acc);
    d[r] = acc;
  }
  emu_wave_sync();
  return d;
}
// bf16 helpers + v_mfma_f32_16x16x32_bf16 (slot j of lane group kb pairs A with B; see devintrin.h)
struct alignas(16) u16x8 {
  unsigned short v[8];
  unsigned short& operator[](int i) { return v[i]; }
  const unsigned short& operator[](int i) const { return v[i]; }
};
inline unsigned short emu_f2bf(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40);   // quiet NaN
  u += 0x7FFFu + ((u >> 16) & 1u);                                                   // round to nearest even
  return (unsigned short)(u >> 16);
}
inline float emu_bf2f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
inline u16x8 bf16_pack8(const float (&x)[8]) { u16x8 r; for (int i = 0; i < 8; i++) r[i] = emu_f2bf(x[i]); return r; }
inline unsigned bf16_pack2(float lo, float hi) { return (unsigned)emu_f2bf(lo) | ((unsigned)emu_f2bf(hi) << 16); }
inline f32x4 mfma16x16x32_bf16(u16x8 a, u16x8 b, f32x4 c) {
  int l = emu_lane();
  for (int j = 0; j < 8; j++) { *emu_slot(l, j) = emu_bf2f(a[j]); *emu_slot(l, 8 + j) = emu_bf2f(b[j]); }
  emu_wave_sync();
  f32x4 d = c;
  for (int r = 0; r < 4; r++) {
    int row = (l >> 4) * 4 + r, col = l & 15;
    float acc = c[r];
    for (int kb = 0; kb < 4; kb++)
      for (int j = 0; j < 8; j++) acc = fmaf(*emu_slot(kb * 16 + row, j), *emu_slot(kb * 16 + col, 8 + j), acc);
    d[r] = acc;
  }
  emu_wave_sync();
  return d;
}
// ds_read_b64_tr_b16: see devintrin.h (lane i of a 16-lane group gets element i % 4 of the chunks of lanes i / 4 + 4 j)
inline u32x2 lds_read_tr16(const unsigned short* p) {
  const int l = emu_lane();
  memcpy(emu_slot(l, 0), p, 8);
  emu_wave_sync();
  unsigned short r[4];
  const int g = l & ~15, i = l & 15;
  for (int j = 0; j < 4; j++) r[j] = reinterpret_cast<const unsigned short*>(emu_slot(g + 4 * j + (i >> 2), 0))[i & 3];
  emu_wave_sync();
  u32x2 out;
  memcpy(&out, r, 8);
  return out;
}
typedef const char* LdsAddr;
inline LdsAddr lds_addr(const void* p) { return static_cast<const char*>(p); }
inline LdsAddr lds_addr_add(LdsAddr a, int bytes) { return a + bytes; }
template <int OFF> inline u32x2 lds_read_tr16_raw(LdsAddr a) { return lds_read_tr16(reinterpret_cast<const unsigned short*>(a + OFF)); }
inline u16x8 join_u16x8(u32x2 a, u32x2 b) {
  u16x8 r;
  memcpy(&r.v[0], &a, 8);
  memcpy(&r.v[4], &b, 8);
  return r;
}
inline int wave_uniform(int x) { return x; }
inline long long dev_clock() { return 0; }
inline long long wall_clock() {   // 100 MHz ticks, like s_memrealtime
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 100000000LL + ts.tv_nsec / 10;
}
inline float fast_exp(float x) { return expf(x); }
inline float fast_exp2(float x) { return exp2f(x); }
inline float fast_rcp(float x) { return 1.0f / x; }
inline float fast_log2(float x) { return log2f(x); }
#define KEEP_ALIVE(x) (void)(x)
#define SCHED_FENCE() ((void)0)
#define OPAQUE(x) ((void)0)
struct BufF32 { float* base; size_t bytes; };
constexpr unsigned BUF_OOB = 0xFFFFFFF0u;
constexpr unsigned BUF_OOB_BASE = 0x80000000u;
inline BufF32 make_buf(const float* base, size_t bytes) { return BufF32{const_cast<float*>(base), bytes}; }
inline float buf_load(BufF32 b, unsigned off) { return ((size_t)off + 4 <= b.bytes) ? b.base[off / 4] : 0.0f; }
inline f32x4 buf_load4(BufF32 b, unsigned off) { f32x4 r; for (int i = 0; i < 4; i++) r[i] = buf_load(b, off + 4 * i); return r; }
inline void buf_store(BufF32 b, unsigned off, float v) { if ((size_t)off + 4 <= b.bytes) b.base[off / 4] = v; }
// LDS-DMA: lane l of the wave deposits its 16 bytes (zeros when out of range) at the wave-uniform base + 16 l; synchronous here
inline void lds_dma16(BufF32 b, unsigned off, void* lds_wave_base) {
  const f32x4 v = buf_load4(b, off);
  *reinterpret_cast<f32x4*>(static_cast<char*>(lds_wave_base) + 16 * (threadIdx.x & 63)) = v;
}
template <int N> inline void wait_vmcnt() {}
inline void wait_lgkmcnt0() {}
inline void wg_barrier() { __syncthreads(); }
inline void wave_lds_fence() { emu_wave_sync(); }
inline float buf_load_s(BufF32 b, unsigned lane_off, unsigned uni) { return ((size_t)lane_off + 4 <= b.bytes) ? b.base[(lane_off + uni) / 4] : 0.0f; }
inline void buf_store_s(BufF32 b, unsigned lane_off, unsigned uni, float v) { if ((size_t)lane_off + 4 <= b.bytes) b.base[(lane_off + uni) / 4] = v; }
inline void buf_store4(BufF32 b, unsigned off, f32x4 v) { for (int i = 0; i < 4; i++) buf_store(b, off + 4 * i, v[i]); }
inline void buf_store_u32(BufF32 b, unsigned off, unsigned v) { if ((size_t)off + 4 <= b.bytes) reinterpret_cast<unsigned*>(b.base)[off / 4] = v; }
inline void buf_store_u32_s(BufF32 b, unsigned lane_off, unsigned uni, unsigned v) {
  if ((size_t)lane_off + 4 <= b.bytes) reinterpret_cast<unsigned*>(b.base)[(lane_off + uni) / 4] = v;
}
inline void buf_store_u32x2_s(BufF32 b, unsigned lane_off, unsigned uni, u32x2 v) {
  if ((size_t)lane_off + 8 <= b.bytes) { unsigned* p = reinterpret_cast<unsigned*>(b.base) + (lane_off + uni) / 4; p[0] = v[0]; p[1] = v[1]; }
}
inline float max_f32(float x, float y) { return fmaxf(x, y); }
#define KEEP_ALIVE2(x) (void)(x)
inline f32x4 buf_load4_dev(BufF32 b, unsigned off) { return buf_load4(b, off); }
#define COMPILER_MEMORY_BARRIER() asm volatile("" ::: "memory")
inline void buf_store_wt(BufF32 b, unsigned off, float v) { buf_store(b, off, v); }
inline f32x4 buf_load4_wt(BufF32 b, unsigned off) { return buf_load4(b, off); }
inline float buf_load_wt(BufF32 b, unsigned off) { return buf_load(b, off); }
inline void buf_store4_wt(BufF32 b, unsigned off, f32x4 v) { for (int i = 0; i < 4; i++) buf_store(b, off + 4 * i, v[i]); }
inline int load_i32_wt(const int* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
inline void store_i32_wt(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline void atomic_add_i32(int* p, int v) { __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomic_fetch_add_i32(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int hw_xcc_id() { return (int)(blockIdx.x & 7u); }   // the dispatcher's round-robin placement
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 1
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST)
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, __ATOMIC_SEQ_CST)
constexpr int GRID_WATCHDOG_SPINS = 1 << 24;
inline void sleep_some() { emu_yield(); sched_yield(); }
inline void poll_pause() { emu_yield(); sched_yield(); }
inline void sleep_iterations(int) { emu_yield(); sched_yield(); }
inline int wave_max_i(int x) { for (int m = 32; m >= 1; m >>= 1) { const int y = wave_shfl_i(x, emu_lane() ^ m); x = y > x ? y : x; } return x; }
inline void drain_vmem() { emu_wave_sync(); }   // hardware: covers every lane of the wave -- here the lanes are fibers that run apart
inline unsigned mad_u24(unsigned a, unsigned b, unsigned c) { return a * b + c; }
template <typename T> inline T* dyn_smem() { return reinterpret_cast<T*>(emu_blk->smem); }

inline void emu_init_block(EmuBlock& blk, unsigned nthreads, std::vector<char>& sm, size_t smem) {
  if (nthreads % 64 != 0) { fprintf(stderr, "emu: block size must be a multiple of 64\n"); abort(); }
  blk.xf.assign((size_t)nthreads * 16, 0.f);
  sm.assign(smem + 16, 0);
  blk.smem = sm.data();
}

// cooperative launch: ALL blocks live at once (needed by grid_barrier) -- one OS thread per workgroup; only
// for small test grids
template <typename K, typename A>
void emu_launch_coop(K kernel, dim3 grid, dim3 block, size_t smem, A arg) {
  const unsigned nthreads = block.x * block.y * block.z;
  const unsigned nblocks = grid.x * grid.y * grid.z;
  if (nblocks > 512) { fprintf(stderr, "emu: cooperative grid too large for the emulator\n"); abort(); }
  struct Ctx { K kernel; A* arg; } ctx{kernel, &arg};
  std::vector<std::thread> th;
  th.reserve(nblocks);
  for (unsigned b = 0; b < nblocks; b++)
    th.emplace_back([&, b]() {
      EmuBlock blk;
      std::vector<char> sm;
      emu_init_block(blk, nthreads, sm, smem);
      blockIdx = {b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y)};
      blockDim = block;
      gridDim = grid;
      emu_run_block(blk, block, [](void* p) { Ctx* c = (Ctx*)p; c->kernel(*c->arg); }, &ctx);
    });
  for (auto& t : th) t.join();
}
#define CLSTM_LAUNCH_COOP(kernel, grid, block, smem, stream, argstruct) \
  emu_launch_coop(kernel, dim3(grid), dim3(block), smem, argstruct)

template <typename K, typename... Args>
void emu_launch(K kernel, dim3 grid, dim3 block, size_t smem, Args... args) {
  const unsigned nthreads = block.x * block.y * block.z;
  auto call = [&]() { kernel(args...); };
  typedef decltype(call) Call;
  struct Ctx { Call* fn; dim3 grid, block; unsigned nthreads; size_t smem; } ctx{&call, grid, block, nthreads, smem};
  emu_parallel_for(grid.x * grid.y * grid.z, [](unsigned b, void* p) {
    Ctx* c = (Ctx*)p;
    EmuBlock blk;
    std::vector<char> sm;
    emu_init_block(blk, c->nthreads, sm, c->smem);
    blockIdx = {b % c->grid.x, (b / c->grid.x) % c->grid.y, b / (c->grid.x * c->grid.y)};
    blockDim = c->block;
    gridDim = c->grid;
    emu_run_block(blk, c->block, [](void* q) { (*((Ctx*)q)->fn)(); }, c);
  }, &ctx);
}
#define CLSTM_LAUNCH(kernel, grid, block, smem, stream, ...) \
  emu_launch(kernel, dim3(grid), dim3(block), smem, __VA_ARGS__)
