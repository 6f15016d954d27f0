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

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DEVFN __device__ __forceinline__
#define DEVMFN __device__ __forceinline__  // member functions

// ---- DPP lane moves (LLVM DppCtrl encodings: quad_perm 0x00-0xFF, row_ror:n 0x120+n) ----
template <int CTRL>
DEVFN float dpp_mov(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
#ifndef CLSTM_USE_SHFL
DEVFN float quad_xor1(float x) { return dpp_mov<0xB1>(x); }  // quad_perm [1,0,3,2]
DEVFN float quad_xor2(float x) { return dpp_mov<0x4E>(x); }  // quad_perm [2,3,0,1]
template <int I>
DEVFN float quad_bcast(float x) { return dpp_mov<I * 0x55>(x); }  // quad_perm [I,I,I,I]
template <int N>
DEVFN float row_ror(float x) { return dpp_mov<0x120 + N>(x); }  // rotate within a row of 16
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
#endif
DEVFN float wave_shfl(float x, int src) { return __shfl(x, src, 64); }
DEVFN float wave_shfl_up1(float x) { return __shfl_up(x, 1, 64); }
DEVFN int wave_shfl_i(int x, int src) { return __shfl(x, src, 64); }
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

// ---- f32 MFMA: D(16x16) += A(16x4) * B(4x16); exact f32 fma chain (guide: cdna §3) ----
// lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; holds D[row=(l>>4)*4+r][col=l&15].
DEVFN f32x4 mfma16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

DEVFN float fast_exp(float x) { return __expf(x); }
DEVFN float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <typename T>
DEVFN T* dyn_smem() {
  extern __shared__ __attribute__((aligned(16))) char clstm_dyn_smem_[];
  return reinterpret_cast<T*>(clstm_dyn_smem_);
}

#define CLSTM_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#endif  // CLSTM_HIP_EMU

// ---------------------------------------------------------------------------------------
// activation functions shared by every kernel (and replicated in numpy by the CPU tests)
// ---------------------------------------------------------------------------------------
// sigmoid(x) = 1/(1+exp(-x))  (Eigen scalar_sigmoid_op form used by clstm_compute.cc:117,196)
DEVFN float sigmoid_dev(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// tanh: odd Taylor polynomial (through x^11) below 0.4, 1 - 2/(1+exp(2|x|)) above; ~5e-7 relative.
DEVFN float tanh_dev(float x) {
  const float ax = fabsf(x);
  const float x2 = x * x;
  float p = -1382.0f / 155925.0f;
  p = p * x2 + 62.0f / 2835.0f;
  p = p * x2 - 17.0f / 315.0f;
  p = p * x2 + 2.0f / 15.0f;
  p = p * x2 - 1.0f / 3.0f;
  p = p * x2 + 1.0f;
  const float small = x * p;
  const float r = fast_rcp(1.0f + fast_exp(2.0f * ax));
  const float big = copysignf(1.0f - 2.0f * r, x);
  return ax < 0.4f ? small : big;
}
