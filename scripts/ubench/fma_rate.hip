// micro-benchmark: issue cost of v_pk_fma_f32 vs v_fma_f32 per wave64 on gfx950, 1 / 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x2 a[8]; float s[16];
  for (int i = 0; i < 8; i++) a[i] = (f32x2){0.f + i, 1.f + threadIdx.x};
  for (int i = 0; i < 16; i++) s[i] = i + threadIdx.x * 0.5f;
  f32x2 w = (f32x2){1.0001f, 0.9999f}; float ws = 1.0001f;
  f32x2 h = (f32x2){0.5f, 0.25f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], w, h);      // 32 pk_fma
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = __builtin_fmaf(s[i], ws, 0.5f);            // 64 v_fma
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float acc = 0;
  for (int i = 0; i < 8; i++) acc += a[i][0] + a[i][1];
  for (int i = 0; i < 16; i++) acc += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
  const int iters = 200000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {64, 256, 512, 768, 1024}) {
    for (int mode = 0; mode < 2; mode++) {
      long long c = 0; float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      }
      const int n = mode == 0 ? 32 : 64;
      const double instr = (double)iters * n;
      printf("threads %4d %-12s: %6.2f ticks/instr (wave 0), %6.3f ns/instr per wave, tick = %.3f ns, CU rate %.1f MAC/ns\n", threads,
             mode == 0 ? "v_pk_fma_f32" : "v_fma_f32", (double)c / instr, ms * 1e6 / instr, ms * 1e6 / (double)c,
             instr * (threads / 64) * (mode == 0 ? 128 : 64) / (ms * 1e6));
    }
  }
  return 0;
}
