// micro-benchmark 2: v_pk_fma_f32 with 56 distinct weight register pairs, as in the recurrence step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, long long* cyc, int iters) {
  f32x2 w01[28], w23[28];
  for (int i = 0; i < 28; i++) { w01[i] = (f32x2){in[threadIdx.x + i], in[threadIdx.x + 64 + i]}; w23[i] = (f32x2){in[threadIdx.x + 2 * i], in[threadIdx.x + 3 * i]}; }
  float hv[28];
  for (int i = 0; i < 28; i++) hv[i] = in[(threadIdx.x & 3) * 28 + i];
  f32x2 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = (f32x2){0.f, 0.f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 2) {   // h changes every iteration: the {h,h} pairs cannot be hoisted -> op_sel broadcast form
#pragma unroll
      for (int i = 0; i < 28; i++) asm volatile("" : "+v"(hv[i]));
    }
#pragma unroll
    for (int j = 0; j < 7; j++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float h = MODE == 0 ? hv[0] : hv[4 * j + e];
        const f32x2 hb = (f32x2){h, h};
        acc[2 * e] = __builtin_elementwise_fma(w01[4 * j + e], hb, acc[2 * e]);
        acc[2 * e + 1] = __builtin_elementwise_fma(w23[4 * j + e], hb, acc[2 * e + 1]);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float *in, *out; long long* cyc; hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
  const int iters = 20000;
  for (int threads : {64, 256, 448}) {
    for (int mode = 0; mode < 3; mode++) {
      long long c = 0;
      for (int rep = 0; rep < 2; rep++) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, in, out, cyc, iters);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, in, out, cyc, iters);
        else hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, in, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      }
      printf("threads %3d %s: %6.2f cycles per v_pk_fma_f32 (wave 0), %7.1f per 56\n", threads,
             mode == 0 ? "distinct weights, one h   " : mode == 1 ? "distinct weights, 28 h    " : "28 h, op_sel broadcast   ", (double)c / iters / 56, (double)c / iters);
    }
  }
  return 0;
}
