// micro-benchmark 3: latency of DEPENDENT operations in a single wave (what bounds the serial chains of the
// recurrence tails and of the CTC lattice step).  Each chain is 64 dependent instructions per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE>
__global__ __launch_bounds__(64) void k(const float* in, float* out, long long* cyc, int iters) {
  __shared__ double tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = 0.0;   // zeros: a loaded value doubles as the next byte offset
  __syncthreads();
  float a = in[threadIdx.x], b = in[threadIdx.x + 64] + 1.0f;
  double da = a, db = b;
  unsigned addr = (unsigned)(size_t)tab + (threadIdx.x & 63) * 8;   // LDS byte address (low 32 bits)
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (MODE == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
    if (MODE == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(da) : "v"(db));) }
    if (MODE == 3) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(da) : "v"(db));) }
    if (MODE == 4) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(da) : "v"(db));) }
    if (MODE == 5) { REP8(REP8(asm volatile("v_cvt_f64_f32 %1, %0\n v_cvt_f32_f64 %0, %1" : "+v"(a), "+v"(da));)) }   // 128 ops
    if (MODE == 6) { REP64(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));) }
    if (MODE == 7) { REP64(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));) }
    if (MODE == 8) { REP64(asm volatile("v_exp_f32 %0, %0" : "+v"(a));) }
    if (MODE == 9) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
    if (MODE == 10) { unsigned lo, hi; REP64(asm volatile("ds_read_b64 %0, %2\n s_waitcnt lgkmcnt(0)\n v_add_u32 %2, %2, %1" : "=&v"(da), "=&v"(lo), "+v"(addr)); (void)hi; lo = 0;) }
    if (MODE == 11) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(da) : "v"(db));) }
    if (MODE == 12) { REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(a));) }
    if (MODE == 13) { REP64(asm volatile("v_cvt_f32_f64 %0, %1\n" : "+v"(a) : "v"(da)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(a));) }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a + (float)da + (float)addr;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> double run(const float* in, float* out, long long* cyc, int iters) {
  long long c = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  }
  return (double)c / iters / 64;
}
int main() {
  float *in, *out; long long* cyc; hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
  const int it = 2000;
  printf("dependent v_add_f32            %6.2f cycles\n", run<0>(in, out, cyc, it));
  printf("dependent v_fma_f32            %6.2f\n", run<1>(in, out, cyc, it));
  printf("dependent v_pk_fma_f32         %6.2f\n", run<11>(in, out, cyc, it));
  printf("dependent v_fma_f64            %6.2f\n", run<2>(in, out, cyc, it));
  printf("dependent v_add_f64            %6.2f\n", run<3>(in, out, cyc, it));
  printf("dependent v_mul_f64            %6.2f\n", run<4>(in, out, cyc, it));
  printf("cvt_f64_f32 + cvt_f32_f64 pair %6.2f (per pair)\n", run<5>(in, out, cyc, it));
  printf("cvt_f32_f64 + v_add_f32 pair   %6.2f (per pair)\n", run<13>(in, out, cyc, it));
  printf("dependent v_mov_b32_dpp shr    %6.2f\n", run<6>(in, out, cyc, it));
  printf("dependent v_add_f32_dpp quad   %6.2f\n", run<7>(in, out, cyc, it));
  printf("dependent v_exp_f32            %6.2f\n", run<8>(in, out, cyc, it));
  printf("dependent v_rcp_f32            %6.2f\n", run<12>(in, out, cyc, it));
  printf("dependent v_cndmask_b32        %6.2f\n", run<9>(in, out, cyc, it));
  printf("ds_read_b64 -> address chain   %6.2f (incl. v_add_u32)\n", run<10>(in, out, cyc, it));
  return 0;
}
