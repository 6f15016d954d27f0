// devintrin.h -- the few gfx950 primitives the kernels are written against.
//
// Product build (hipcc --offload-arch=gfx950): DPP cross-lane moves, the f32 MFMA,
// native exp/rcp.  wave = 64 lanes everywhere (CDNA4).
//
// When CLSTM_HIP_EMU is defined (ONLY by tests/hipemu/, a host-thread emulator used by the
// CPU test-suite to exercise kernel indexing/synchronisation logic without a GPU) the same
// names are provided by tests/hipemu/hip_emu.h.  Nothing under clstm_amd/ ever builds or
// loads that variant.
#pragma once

#ifdef CLSTM_HIP_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define DEVFN __device__ __forceinline__
#define DEVMFN __device__ __forceinline__  // member functions

// ---- DPP lane moves (LLVM DppCtrl encodings: quad_perm 0x00-0xFF, row_ror:n 0x120+n) ----
template <int CTRL>
DEVFN float dpp_mov(float x) {
  const int xi = __builtin_bit_cast(int, x);  // old = src: no zero-init mov, lanes without a source keep x
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, CTRL, 0xF, 0xF, false));
}
// Permutations in which EVERY lane has a source lane (quad_perm, row_ror, row_half_mirror): old = 0 with
// bound_ctrl and full row/bank masks makes `old` dead, so hipcc needs no copy to seed the destination and can
// fold the move into the consuming VALU instruction (v_add_f32_dpp, v_mul_f32_dpp) -- each saved instruction
// sits on the dependent tail of a recurrence step.
template <int CTRL>
DEVFN float dpp_perm(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
#ifndef CLSTM_USE_SHFL
DEVFN float quad_xor1(float x) { return dpp_perm<0xB1>(x); }  // quad_perm [1,0,3,2]
DEVFN float quad_xor2(float x) { return dpp_perm<0x4E>(x); }  // quad_perm [2,3,0,1]
template <int I>
DEVFN float quad_bcast(float x) { return dpp_perm<I * 0x55>(x); }  // quad_perm [I,I,I,I]
template <int N>
DEVFN float row_ror(float x) { return dpp_perm<0x120 + N>(x); }  // rotate within a row of 16
DEVFN float row_half_mirror(float x) { return dpp_perm<0x141>(x); }  // lane i <-> i^7 within 8 lanes
DEVFN float quad_mirror(float x) { return dpp_perm<0x1B>(x); }        // quad_perm [3,2,1,0]: lane i <-> i^3
#else
// Fallback through ds_bpermute (definitionally correct; used to cross-check the DPP codes).
DEVFN float quad_xor1(float x) { return __shfl_xor(x, 1, 64); }
DEVFN float quad_xor2(float x) { return __shfl_xor(x, 2, 64); }
template <int I>
DEVFN float quad_bcast(float x) { return __shfl(x, (int)((threadIdx.x & 63u) & ~3u) | I, 64); }
template <int N>
DEVFN float row_ror(float x) {
  const int lane = threadIdx.x & 63;
  return __shfl(x, (lane & ~15) | ((lane + N) & 15), 64);
}
DEVFN float row_half_mirror(float x) { return __shfl(x, (int)(threadIdx.x & 63u) ^ 7, 64); }
DEVFN float quad_mirror(float x) { return __shfl_xor(x, 3, 64); }
#endif
DEVFN float wave_shfl(float x, int src) { return __shfl(x, src, 64); }
DEVFN float wave_shfl_up1(float x) { return __shfl_up(x, 1, 64); }
#ifdef CLSTM_USE_SHFL
template <int I> DEVFN float mul_quad_bcast(float x, float y) { return quad_bcast<I>(x) * y; }
template <int I> DEVFN float fmac_quad_bcast(float acc, float x, float y) { return fmaf(quad_bcast<I>(x), y, acc); }
template <int I> DEVFN float mul_quad_bcast_old(float x, float y) { return quad_bcast<I>(x) * y; }
#endif
#ifndef CLSTM_USE_SHFL
DEVFN float wave_shr1(float x) { return dpp_mov<0x138>(x); }  // DPP wave_shr:1 (lane 0 keeps its value)
// lanes >= 1: a[lane-1] + b[lane]; lane 0 (no source lane, bound_ctrl off) keeps `old`.  One VALU operation;
// hipcc does not fold a DPP move with a live old operand into its consumer, hence the asm (s_nop: the
// VALU-write -> DPP-read hazard is invisible to the compiler inside asm).
DEVFN float add_wave_shr1(float old, float a, float b) {
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(a), "v"(b));
  return old;
}
#else
DEVFN float wave_shr1(float x) { return __shfl_up(x, 1, 64); }
DEVFN float add_wave_shr1(float old, float a, float b) {
  const float s = __shfl_up(a, 1, 64) + b;
  return (threadIdx.x & 63) == 0 ? old : s;
}
#endif
DEVFN unsigned long long wave_ballot(bool p) { return __ballot(p); }   // bit l = predicate of lane l
DEVFN int lds_atomic_min(int* p, int v) { return atomicMin(p, v); }
DEVFN int wave_shfl_i(int x, int src) { return __shfl(x, src, 64); }
DEVFN int wave_shfl_xor_i(int x, int m) { return __shfl_xor(x, m, 64); }
DEVFN float wave_max(float x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x = fmaxf(x, __shfl_xor(x, m, 64));
  return x;
}
DEVFN float wave_sum(float x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
  return x;
}

// packed f32 FMA (v_pk_fma_f32): two lanes of work per VALU issue slot
DEVFN f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
DEVFN f32x2 splat2(float x) { return (f32x2){x, x}; }
// acc += w * (one element of an aligned register pair, broadcast to both halves) as ONE v_pk_fma_f32: op_sel picks
// the element.  hipcc finds this form for elements x, y, z of a ds_read_b128 result but copies element w into a
// fresh pair first (a v_mov_b32 per group of four k on the recurrence's issue-bound mat-vec); spelled out here.
DEVFN f32x2 pair_lo(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
DEVFN f32x2 pair_hi(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }
DEVFN f32x2 fma2_lo(f32x2 w, f32x2 hp, f32x2 acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(hp));
  return acc;
}
DEVFN f32x2 fma2_hi(f32x2 w, f32x2 hp, f32x2 acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(w), "v"(hp));
  return acc;
}
// Quad broadcasts folded into their consumer (one VALU operation instead of v_mov_b32_dpp + the arithmetic; hipcc
// does not fold these).  s_nop 1: the VALU-write -> DPP-read hazard on x is invisible to the compiler inside asm.
#define CLSTM_QUAD_OPS(I)                                                                                              \
  template <> DEVFN float mul_quad_bcast<I>(float x, float y) { /* x[quad lane I] * y */                               \
    float r;                                                                                                           \
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 quad_perm:[" #I "," #I "," #I "," #I "] row_mask:0xf bank_mask:0xf bound_ctrl:1" \
        : "=v"(r) : "v"(x), "v"(y));                                                                                  \
    return r;                                                                                                          \
  }                                                                                                                    \
  template <> DEVFN float fmac_quad_bcast<I>(float acc, float x, float y) { /* acc + x[quad lane I] * y, fused */      \
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[" #I "," #I "," #I "," #I "] row_mask:0xf bank_mask:0xf bound_ctrl:1" \
        : "+v"(acc) : "v"(x), "v"(y));                                                                                \
    return acc;                                                                                                        \
  }
// ..._old: x is known to have been written long before (no hazard, no s_nop on the step's dependent tail)
#define CLSTM_QUAD_OPS_OLD(I)                                                                                          \
  template <> DEVFN float mul_quad_bcast_old<I>(float x, float y) {                                                    \
    float r;                                                                                                           \
    asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[" #I "," #I "," #I "," #I "] row_mask:0xf bank_mask:0xf bound_ctrl:1"     \
        : "=v"(r) : "v"(x), "v"(y));                                                                                  \
    return r;                                                                                                          \
  }
#ifndef CLSTM_USE_SHFL
template <int I> DEVFN float mul_quad_bcast(float x, float y);
template <int I> DEVFN float mul_quad_bcast_old(float x, float y);
template <int I> DEVFN float fmac_quad_bcast(float acc, float x, float y);
CLSTM_QUAD_OPS(0) CLSTM_QUAD_OPS(1) CLSTM_QUAD_OPS(2) CLSTM_QUAD_OPS(3)
CLSTM_QUAD_OPS_OLD(0) CLSTM_QUAD_OPS_OLD(1) CLSTM_QUAD_OPS_OLD(2) CLSTM_QUAD_OPS_OLD(3)
#endif

// ---- f32 MFMA: D(16x16) += A(16x4) * B(4x16); exact f32 fma chain (guide: cdna §3) ----
// lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; holds D[row=(l>>4)*4+r][col=l&15].
DEVFN f32x4 mfma16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- bf16 MFMA (gemm_bf16.h): D(16x16) += A(16x32) * B(32x16), f32 accumulate ----
// lane l supplies 8 bf16 of row/column l&15; the k slots of lane group l>>4 are paired A-to-B by the
// hardware, so any slot->k assignment is valid as long as both operands use the same one.
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
DEVFN u16x8 bf16_pack8(const float (&x)[8]) {   // round to nearest even (v_cvt_pk_bf16_f32)
  bf16x8_t v;
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = (__bf16)x[i];
  return __builtin_bit_cast(u16x8, v);
}
DEVFN unsigned bf16_pack2(float lo, float hi) {
  bf16x2_t v;
  v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
DEVFN f32x4 mfma16x16x32_bf16(u16x8 a, u16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ds_read_b64_tr_b16 (gfx950): the 16 lanes of a lane group each pass the address of 4 consecutive bf16 (8-byte aligned);
// with lane i pointing at chunk i of a [4][16] row-major block, lane i receives COLUMN i (elements i, 16 + i, 32 + i,
// 48 + i), i.e. result(i, j) = chunk (4 j + i / 4), element i % 4 -- measured on the part by scripts/probe/tr16.hip.
// This is the MFMA fragment of an operand that lies contraction-major in memory (gemm_bf16.h, gemm_b16mc).
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
DEVFN u32x2 lds_read_tr16(const unsigned short* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)p));
}
// The same read as inline asm, byte address in LDS + immediate offset: beside LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in front
// of the BUILTIN (it cannot tell which LDS bytes the read touches and takes every pending DMA for a producer): the whole
// request pipeline drained at every block (gemm_b16mc_dma_kernel: 533 vs 284 us, found in the ISA).  An asm statement is
// invisible to that pass -- the caller orders the read against the DMA itself (wait_vmcnt + wg_barrier) and waits for the result
// with wait_lgkmcnt0() before it is used.
typedef unsigned LdsAddr;   // byte address in LDS (the host emulator: a pointer)
DEVFN LdsAddr lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }
DEVFN LdsAddr lds_addr_add(LdsAddr a, int bytes) { return a + (unsigned)bytes; }
template <int OFF>
DEVFN u32x2 lds_read_tr16_raw(LdsAddr addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
DEVFN u16x8 join_u16x8(u32x2 a, u32x2 b) {
  u32x4 v;
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
  return __builtin_bit_cast(u16x8, v);
}

// value known to be wave-uniform (e.g. threadIdx.x >> 6): tell the compiler so that buffer
// descriptors derived from it live in SGPRs instead of waterfall loops (guide T20)
DEVFN int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
DEVFN long long dev_clock() { return (long long)__builtin_readcyclecounter(); }
// chip-wide constant-rate clock (100 MHz): comparable across CUs, for launch-internal timelines (diagnostics)
DEVFN long long wall_clock() { return (long long)__builtin_amdgcn_s_memrealtime(); }
DEVFN float fast_exp(float x) { return __expf(x); }
DEVFN float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
DEVFN float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEVFN float fast_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32

// ---- buffer (SRD) loads/stores: out-of-range offsets are dropped by the bounds check, so masked
// lanes need no exec-mask branch and hipcc can count the VMEM queue exactly (vmcnt(N), N>0: the
// fire-and-forget stores of a recurrence step never sit on the next step's critical path).
struct BufF32 { __amdgpu_buffer_rsrc_t r; };
constexpr unsigned BUF_OOB = 0xFFFFFFF0u;
// lane base offset for masked lanes when a (wave-uniform, < 2^31) step offset is ADDED to it: the sum
// stays >= 2^31 > num_records, hence out of range, and cannot wrap into the buffer
constexpr unsigned BUF_OOB_BASE = 0x80000000u;
DEVFN BufF32 make_buf(const float* base, size_t bytes) {
  BufF32 b;
  b.r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0x7FFFFFF0ull ? 0x7FFFFFF0ull : bytes), 0x00020000);
  return b;
}
DEVFN float buf_load(BufF32 b, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, byte_off, 0, 0));
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
DEVFN f32x4 buf_load4(BufF32 b, unsigned byte_off) {  // 16 bytes, dword alignment suffices
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, byte_off, 0, 0));
}
DEVFN void buf_store(BufF32 b, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), b.r, byte_off, 0, 0);
}
// LDS-DMA (buffer_load_dwordx4 ... lds): 16 bytes per lane straight from memory into LDS -- no VGPR, no ds_write.  The
// destination is WAVE-UNIFORM base + 16 x lane (a wave instruction fills 1 KB of LDS in lane order; `lds_wave_base` must be
// the same in every lane), the source offset is per lane: a swizzled LDS image is obtained by permuting the SOURCE chunks.
// Out-of-range lanes deposit zeros.  Counts in vmcnt like a load; the data is in LDS for the issuing wave once its vmcnt says
// so, for the other waves behind a barrier after that (wait_vmcnt<N>() + wg_barrier(): __syncthreads() would wait vmcnt(0)).
DEVFN void lds_dma16(BufF32 b, unsigned byte_off, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
// dword form: 4 bytes per lane, a wave instruction fills 256 bytes of LDS in lane order (rows that are not a multiple of 16 lanes x 16 B)
DEVFN void lds_dma4(BufF32 b, unsigned byte_off, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, byte_off, 0, 0, 0);
}
template <int N> DEVFN void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
DEVFN void wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
DEVFN void wg_barrier() { __builtin_amdgcn_s_barrier(); }   // bare s_barrier: no implied waits
// lanes of ONE wave exchanging data through LDS: the wave executes in lock-step and its LDS operations complete in order, so the
// hardware needs nothing but the compiler's own lgkmcnt wait (the host emulator runs lanes as fibers: a real rendezvous there)
DEVFN void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
// lane offset (VGPR, bounds-checked) + wave-uniform offset (SGPR, NOT part of the bounds check of a raw
// buffer): no VALU add per access, and a lane parked at BUF_OOB_BASE stays out of range
DEVFN float buf_load_s(BufF32 b, unsigned lane_off, unsigned uniform_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, lane_off, uniform_off, 0));
}
DEVFN void buf_store_s(BufF32 b, unsigned lane_off, unsigned uniform_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), b.r, lane_off, uniform_off, 0);
}
// wider / integer forms of the same (byte offsets; a lane at BUF_OOB drops its store)
DEVFN void buf_store4(BufF32 b, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), b.r, byte_off, 0, 0);
}
DEVFN void buf_store_u32(BufF32 b, unsigned byte_off, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32((int)v, b.r, byte_off, 0, 0); }
DEVFN void buf_store_u32x2_s(BufF32 b, unsigned lane_off, unsigned uniform_off, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, v), b.r, lane_off, uniform_off, 0);
}
DEVFN void buf_store_u32_s(BufF32 b, unsigned lane_off, unsigned uniform_off, unsigned v) {
  __builtin_amdgcn_raw_buffer_store_b32((int)v, b.r, lane_off, uniform_off, 0);
}
DEVFN float max_f32(float x, float y) {   // v_max_f32 without the canonicalising self-max of fmaxf (operands are never sNaN)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// ---- device-scope (sc1) accesses and the grid barrier of the cooperative kernels (lstm_wide.h) ----
// Data produced and consumed by DIFFERENT workgroups of one launch: per-XCD L2s are not coherent with
// each other, so such stores are written through (sc1) and such loads ask at device scope (sc1); the
// barrier then needs no cache maintenance (guide: in-launch hand-off, sc1 variant).
DEVFN f32x4 buf_load4_dev(BufF32 b, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, byte_off, 0, 16));
}
// polling loops over such loads: the compiler may hoist a loop-invariant load out of the loop (no write to that memory is
// visible to it); a memory clobber at the top of the loop keeps it inside without changing the cache policy
#define COMPILER_MEMORY_BARRIER() asm volatile("" ::: "memory")
// Producer / consumer pairs in DIFFERENT launches that run concurrently (lstm_bwd -> gemm_dw.h): system-scope
// write-through stores and system-scope loads on both sides (guide: "sc0 sc1 stores and loads both sides" needs no
// fences; per-XCD L2s are not coherent with each other)
DEVFN void buf_store_wt(BufF32 b, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), b.r, byte_off, 0, 17);
}
DEVFN f32x4 buf_load4_wt(BufF32 b, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, byte_off, 0, 17));
}
DEVFN float buf_load_wt(BufF32 b, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, byte_off, 0, 17));
}
DEVFN void buf_store4_wt(BufF32 b, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), b.r, byte_off, 0, 17);
}
DEVFN int load_i32_wt(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
DEVFN void store_i32_wt(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
DEVFN void atomic_add_i32(int* p, int v) { atomicAdd(p, v); }
DEVFN int atomic_fetch_add_i32(int* p, int v) { return atomicAdd(p, v); }
// where this wave runs: XCC (XCD) id 0..7
DEVFN int hw_xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return (int)(x & 0xF); }
DEVFN void sleep_some() { __builtin_amdgcn_s_sleep(16); }
DEVFN void poll_pause() { __builtin_amdgcn_s_sleep(1); }   // ~64 cycles between two looks at a word another workgroup will change
// park the wave for roughly 0.35 us per recurrence iteration still missing (s_sleep 13 ~ 832 cycles), at most ~14 us
DEVFN void sleep_iterations(int n) {
  n = n > 40 ? 40 : n;
  for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(13);
}
DEVFN int wave_max_i(int x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const int y = __shfl_xor(x, m, 64); x = y > x ? y : x; }
  return x;
}
DEVFN void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // this wave's loads returned, stores acknowledged
DEVFN unsigned mad_u24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }   // v_mad_u32_u24 (operands < 2^24)
constexpr int GRID_WATCHDOG_SPINS = 1 << 21;   // ~seconds of polling before a stuck barrier is reported
// bench.py's per-kernel timing.  While a Timing bracket of the host code is open (clstm_hip.hip), every launch carries a start /
// stop event pair bound to its own dispatch packet (hipExtLaunchKernel): stop - start is the kernel's duration by the packet's own
// time stamps -- the figure rocprofv3 --kernel-trace reports -- and nothing is inserted into the stream.  (An event RECORDED in
// front of a launch makes the command processor finish the record before it looks at the launch, which exposes the launch's
// set-up time, 5-7 us for the big fused launches, that normally hides behind the preceding kernel: brackets of recorded events
// read 11 % above rocprofv3.)  Cooperative launches have no such variant and keep recorded events around them.
struct ClstmLaunchEvents { hipEvent_t a, b; };
inline thread_local std::vector<ClstmLaunchEvents>* clstm_launch_sink = nullptr;
inline thread_local std::vector<ClstmLaunchEvents> clstm_event_pool;   // events of collected launches, re-used: creating a pair
inline ClstmLaunchEvents clstm_launch_events() {                       // per launch makes the timed steps host-bound
  ClstmLaunchEvents e{};
  if (!clstm_event_pool.empty()) { e = clstm_event_pool.back(); clstm_event_pool.pop_back(); }
  else { (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); }
  clstm_launch_sink->push_back(e);
  return e;
}
#define CLSTM_LAUNCH_COOP(kernel, grid, block, smem, stream, argstruct)                                  \
  do {                                                                                                   \
    void* coop_args_[] = {(void*)&(argstruct)};                                                          \
    ClstmLaunchEvents ev_{};                                                                             \
    if (clstm_launch_sink) { ev_ = clstm_launch_events(); HIPCHECK(hipEventRecord(ev_.a, (hipStream_t)(stream))); } \
    HIPCHECK(hipLaunchCooperativeKernel((const void*)(kernel), grid, block, coop_args_, smem, (hipStream_t)(stream))); \
    if (ev_.b) HIPCHECK(hipEventRecord(ev_.b, (hipStream_t)(stream)));                                   \
  } while (0)

// A VMEM store reads its data VGPR when the memory pipeline executes it, so hipcc inserts a vmcnt
// wait before that register may be overwritten.  KEEP_ALIVE(x) placed two recurrence steps later
// pins the register until then, which moves that wait off the per-step critical path.
#define KEEP_ALIVE(x) asm volatile("" ::"v"(x))
// instruction-scheduling fence: nothing is moved across it (keeps load issue order = source order)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// value the optimiser must treat as computed here (no sinking into a conditional, no re-materialisation)
#define OPAQUE(x) asm volatile("" : "+v"(x))
#define KEEP_ALIVE2(x) asm volatile("" ::"v"(x))

template <typename T>
DEVFN T* dyn_smem() {
  extern __shared__ __attribute__((aligned(16))) char clstm_dyn_smem_[];
  return reinterpret_cast<T*>(clstm_dyn_smem_);
}

#define CLSTM_LAUNCH(kernel, grid, block, smem, stream, ...)                                                       \
  do {                                                                                                             \
    if (__builtin_expect(clstm_launch_sink != nullptr, 0)) {                                                       \
      const ClstmLaunchEvents ev_ = clstm_launch_events();                                                         \
      hipExtLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), ev_.a, ev_.b, 0, __VA_ARGS__);       \
    } else hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__);                      \
  } while (0)
#endif  // CLSTM_HIP_EMU

// ---- common to both builds ----
// device error words (clstm_hip.hip:dev_err_words; [4], [5] unused): any set -> the update kernels skip the update
DEVFN bool dev_err_set(const int* err) { return err && (err[0] | err[1] | err[2] | err[3] | err[6] | err[7]) != 0; }
DEVFN bool f32_finite(float x) { return (__builtin_bit_cast(unsigned, x) & 0x7f800000u) != 0x7f800000u; }
DEVFN void raise_nonfinite(int* nanflag, int step_no) { if (*nanflag == 0) *nanflag = step_no; }   // (every writer of a launch stores the same value)


// progress words (lstm_seq.h -> gemm_dw.h) sit one per 128-byte line: a workgroup rewrites its word every step
constexpr int PROG_STRIDE = 32;

// ---- staggered recurrence (lstm_seq.h, workgroups of more than four waves): contraction order of a lane ----------
// Waves 0..3 (cells 0..63, the first-dispatched wave of each SIMD) form group A, the rest group B.  A lane's slice of the
// recurrent contraction is ordered [A part | B part] so that a step can start on the half of h_{t-1} that is ready:
//   forward, lane quarter q, slot kk < ku:  kk < 16: cell k = 16 q + kk ;  else cell k = 64 + (ku - 16) q + (kk - 16)
//   LDS position of cell c in the h vector (quarter stride qs): the same map inverted.
constexpr int STAG_NA = 64, STAG_KA = 16;
__host__ __device__ inline constexpr bool stag_on(int nk4) { return nk4 >= 5; }
__host__ __device__ inline int stag_fwd_k(int q, int kk, int ku) { return kk < STAG_KA ? STAG_KA * q + kk : STAG_NA + (ku - STAG_KA) * q + (kk - STAG_KA); }
__host__ __device__ inline int stag_fwd_slot(int cell, int ku, int qs) {
  return cell < STAG_NA ? (cell / STAG_KA) * qs + (cell % STAG_KA)
                        : ((cell - STAG_NA) / (ku - STAG_KA)) * qs + STAG_KA + (cell - STAG_NA) % (ku - STAG_KA);
}

// ---------------------------------------------------------------------------------------
// activation functions shared by every kernel (and replicated in numpy by the CPU tests)
// ---------------------------------------------------------------------------------------
// tanh(x) = (e-1)/(e+1), e = exp(2x) (native v_exp_f32 / v_rcp_f32), x clamped to +-15 so e stays
// finite; below |x| = 0.01 the cancellation in e-1 would cost more than ~6e-6 relative, there
// x - x^3/3 is exact to 1e-9.  Worst-case relative error ~6e-6 (at |x| = 0.01), 3e-7 for |x| > 0.2:
// well inside the 1e-4 activation bar, and less than half the instructions of a polynomial/exp
// hybrid on the per-timestep critical path.
DEVFN float tanh_dev(float x) {
  const float xc = fminf(fmaxf(x, -15.0f), 15.0f);
  const float e = fast_exp(2.0f * xc);
  const float big = (e - 1.0f) * fast_rcp(e + 1.0f);
  const float small = xc - xc * xc * xc * (1.0f / 3.0f);
  // (hipcc turns this select into an exec-masked branch around the exp/rcp chain; forcing it branch-free
  //  with an opaque value was measured SLOWER in the recurrence: 123 -> 134 us)
  return fabsf(xc) < 0.01f ? small : big;
}

// sigmoid(x) = 1/(1+exp(-x))  (Eigen scalar_sigmoid_op form used by clstm_compute.cc:117,196)
DEVFN float sigmoid_dev(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// gate nonlinearity without divergence: sigmoid for the three gates, tanh for the cell input;
// one exp + one rcp either way.
DEVFN float gate_act(float x, bool is_tanh) {
  // One exp and one rcp on the dependent chain, no clamps: with e = exp(-2|x|) in (0, 1] the tanh form
  // (1 - e) / (1 + e) cannot overflow, and sigmoid's exp(-x) = inf gives rcp(inf) = 0, the correct limit.
  // The exponent argument is (x with the sign bit cleared for tanh) * (-2 log2 e or -log2 e): the mask and
  // the scale depend only on the lane, so the chain in front of v_exp_f32 is v_and + v_mul.
  const float ax = fabsf(x);
  const unsigned mask = is_tanh ? 0x7FFFFFFFu : 0xFFFFFFFFu;
  const float scale = is_tanh ? -2.0f * 1.44269504088896341f : -1.44269504088896341f;
  const float e = fast_exp2(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & mask) * scale);
  const float r = fast_rcp(1.0f + e);
  // |x| < 0.35: 1 - e cancels (at |x| = 0.01 the quotient form is 50 float ulps off, and in a 400-step recurrence of
  // 512 cells those relative errors are what the next steps amplify) -- the odd Taylor series to x^11 instead, whose
  // truncation error at 0.35 is 1.2e-8 relative; above it 1 - e >= 0.5 and the quotient is good to ~2 ulp.
  const float s = ax * ax;
  const float p = fmaf(s, fmaf(s, fmaf(s, fmaf(s, fmaf(s, -1382.0f / 155925.0f, 62.0f / 2835.0f), -17.0f / 315.0f), 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
  const float th = copysignf(ax < 0.35f ? ax * p : (1.0f - e) * r, x);
  return is_tanh ? th : r;
}

// The recurrence kernels' nonlinearity: FIVE dependent VALU operations for either kind, no select, no branch:
//   r = 1 / (1 + 2^(x * scale)) ;  act = r * A + B
//   sigmoid: scale = -log2 e, A = 1, B = 0 (exactly gate_act's sigmoid);  tanh x = 2 sigmoid(2x) - 1: scale = -2 log2 e,
//   A = 2, B = -1.  Limits are exact (2^+big = inf -> r = 0; 2^-big = 0 -> r = 1).  The tanh form cancels for small
//   |x|: its ABSOLUTE error stays ~2 ulp of 1 (<= 3e-7: v_exp_f32 and v_rcp_f32 are 1-ulp operations), i.e. inside
//   the parity tests' absolute floor (2e-6) where the relative bar (1e-4) no longer binds.  gate_act's
//   branchy small-|x| series + select + copysign cost ~12 VALU operations and two exec-mask round trips through
//   the scalar unit per evaluation, twice per time step on the step's dependent tail (lstm_seq.h).
constexpr float ACT_SIG_SCALE = -1.44269504088896341f, ACT_TANH_SCALE = -2.0f * 1.44269504088896341f;
DEVFN float act_affine(float x, float scale, float A, float B) {
  const float r = fast_rcp(1.0f + fast_exp2(x * scale));
  return fmaf(r, A, B);
}
DEVFN float tanh_fast(float x) { return act_affine(x, ACT_TANH_SCALE, 2.0f, -1.0f); }
