// micro-benchmark 4: issue rate of INDEPENDENT operations (8 chains) with 1, 4 and 8 waves on one CU
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(1024) void k(const float* in, float* out, long long* cyc, int iters) {
  __shared__ float tab[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = 0.0f;
  __syncthreads();
  float a[8]; double d[8];
  for (int i = 0; i < 8; i++) { a[i] = in[threadIdx.x + i]; d[i] = a[i] + 1.0; }
  float b = in[threadIdx.x + 64] + 1.0f; double db = b;
  const float sb = __builtin_amdgcn_readfirstlane(in[blockIdx.x + 3]) + 1.0f; const double sdb = sb;
  const unsigned long long msk = __builtin_amdgcn_ballot_w64(in[threadIdx.x] == 0.0f && (threadIdx.x & 1));
  unsigned long long mo[2] = {0, 0};
  unsigned addr = (unsigned)(size_t)tab + (threadIdx.x & 63) * 4;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#define ALL8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define F32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define F64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(db));
#define M64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
#define A64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
#define CVTA(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define CVTB(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define LDS(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*256" : "=v"(a[i]) : "v"(addr));
#define LDSW(i) asm volatile("ds_write_b32 %1, %0 offset:" #i "*256" :: "v"(a[i]), "v"(addr));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
#define FSG(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "s"(sb));
#define FSG64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "s"(sdb));
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
#define CMPS(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mo[i & 1]) : "v"(a[i]), "v"(b));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define DPPA(i) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define LIT(i) asm volatile("v_add_f32 %0, 0x40490fdb, %0" : "+v"(a[i]));
    if (MODE == 0) { REP8(ALL8(F32)) }
    if (MODE == 1) { REP8(ALL8(F64)) }
    if (MODE == 2) { REP8(ALL8(M64)) }
    if (MODE == 3) { REP8(ALL8(A64)) }
    if (MODE == 4) { REP8(ALL8(CVTA)) }
    if (MODE == 5) { REP8(ALL8(CVTB)) }
    if (MODE == 6) { REP8(ALL8(LDS) asm volatile("s_waitcnt lgkmcnt(0)");) }
    if (MODE == 7) { REP8(ALL8(LDSW) asm volatile("s_waitcnt lgkmcnt(0)");) }
    if (MODE == 8) { REP8(ALL8(CND)) }
    if (MODE == 9) { REP8(ALL8(CND64)) }
    if (MODE == 10) { REP8(ALL8(FSG)) }
    if (MODE == 11) { REP8(ALL8(FSG64)) }
    if (MODE == 12) { REP8(ALL8(CMP)) }
    if (MODE == 13) { REP8(ALL8(CMPS)) }
    if (MODE == 14) { REP8(ALL8(MAXF)) }
    if (MODE == 15) { REP8(ALL8(DPPA)) }
    if (MODE == 16) { REP8(ALL8(LIT)) }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float)d[i];
  out[threadIdx.x] = s + (float)(mo[0] + mo[1]);
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE> void run(const char* name, const float* in, float* out, long long* cyc) {
  printf("%-22s", name);
  for (int threads : {64, 512, 1024}) {
    long long c = 0, cs[16];
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, in, out, cyc, 2000);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(cs, cyc, 8 * 16, hipMemcpyDeviceToHost);
      c = 0; for (int w = 0; w < threads / 64; w++) if (cs[w] > c) c = cs[w];
    }
    printf("  %d waves: %5.2f", threads / 64, (double)c / 2000 / 64);
  }
  printf("   cycles per instruction (slowest wave)\n");
}
int main() {
  float *in, *out; long long* cyc;
  (void)hipMalloc(&in, 1 << 20); (void)hipMemset(in, 0, 1 << 20); (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 256);
  run<0>("v_fma_f32", in, out, cyc); run<1>("v_fma_f64", in, out, cyc); run<2>("v_mul_f64", in, out, cyc);
  run<3>("v_add_f64", in, out, cyc); run<4>("v_cvt_f64_f32", in, out, cyc); run<5>("v_cvt_f32_f64", in, out, cyc);
  run<8>("v_cndmask_b32 vcc", in, out, cyc); run<9>("v_cndmask_b32 s[..]", in, out, cyc); run<10>("v_fma_f32 sgpr src", in, out, cyc);
  run<11>("v_fma_f64 sgpr src", in, out, cyc); run<12>("v_cmp_lt_f32 vcc", in, out, cyc); run<13>("v_cmp_lt_f32 s[..]", in, out, cyc);
  run<14>("v_max_f32", in, out, cyc); run<15>("v_add_f32_dpp", in, out, cyc); run<16>("v_fma_f32 literal", in, out, cyc); run<6>("ds_read_b32 (x8, wait)", in, out, cyc); run<7>("ds_write_b32 (x8, wait)", in, out, cyc);
  return 0;
}
