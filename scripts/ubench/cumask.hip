// cumask.hip -- which CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) really get on MI355X?
// Every workgroup records (XCC_ID, SE, SH, CU) from the hardware id registers; the host prints the census
// per XCD for a few mask patterns.  A mask that leaves an XCD without CUs would strand that XCD's share of
// the grid, so every pattern tried here keeps CUs in every XCD under any of the plausible bit->CU mappings.
//   build: hipcc -O2 --offload-arch=gfx950 -o cumask cumask.hip      run: timeout 60 ./cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void census(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char* name, const std::vector<uint32_t>* mask, int nblocks) {
  hipStream_t s;
  if (mask) CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask->size(), mask->data()));
  else CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* d;
  CK(hipMalloc(&d, nblocks * 8));
  CK(hipMemset(d, 0, nblocks * 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, s));
  hipLaunchKernelGGL(census, dim3(nblocks), dim3(448), 0, s, d, 200000);
  CK(hipEventRecord(b, s));
  CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned> h(2 * nblocks);
  CK(hipMemcpy(h.data(), d, nblocks * 8, hipMemcpyDeviceToHost));
  std::set<unsigned> cus[8];
  int perxcc[8] = {0};
  for (int i = 0; i < nblocks; i++) {
    const unsigned hw = h[2 * i], x = h[2 * i + 1] & 0xF;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    if (x < 8) { cus[x].insert(se * 32 + sh * 16 + cu); perxcc[x]++; }
  }
  printf("%-28s blocks %4d  %.3f ms | distinct CUs per XCD:", name, nblocks, ms);
  int tot = 0;
  for (int x = 0; x < 8; x++) { printf(" %2zu", cus[x].size()); tot += (int)cus[x].size(); }
  printf(" (total %d) | blocks per XCD:", tot);
  for (int x = 0; x < 8; x++) printf(" %d", perxcc[x]);
  printf("\n   XCD0 CU ids (se*32+sh*16+cu):");
  for (unsigned c : cus[0]) printf(" %u", c);
  printf("\n");
  CK(hipFree(d)); CK(hipStreamDestroy(s));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("%s  CUs %d\n", p.name, p.multiProcessorCount);
  auto mk = [](auto pred) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i++) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
  run("no mask", nullptr, 128);
  run("no mask", nullptr, 256);
  run("no mask", nullptr, 512);
  auto A = mk([](int i) { return (((i >> 3) ^ i) & 1) == 0; });
  auto B = mk([](int i) { return (((i >> 3) ^ i) & 1) == 1; });
  run("mask bit3^bit0 == 0", &A, 128);
  run("mask bit3^bit0 == 1", &B, 128);
  run("mask bit3^bit0 == 0", &A, 256);
  // two streams with complementary masks running concurrently: do they overlap in time?
  {
    hipStream_t s1, s2;
    CK(hipExtStreamCreateWithCUMask(&s1, 8, A.data()));
    CK(hipExtStreamCreateWithCUMask(&s2, 8, B.data()));
    unsigned *d1, *d2; CK(hipMalloc(&d1, 128 * 8)); CK(hipMalloc(&d2, 128 * 8));
    hipEvent_t a, b, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, s1));
    hipLaunchKernelGGL(census, dim3(128), dim3(448), 0, s1, d1, 2000000);
    hipLaunchKernelGGL(census, dim3(128), dim3(448), 0, s2, d2, 2000000);
    CK(hipEventRecord(b, s1)); CK(hipEventRecord(c, s2));
    CK(hipDeviceSynchronize());
    float m1, m2; CK(hipEventElapsedTime(&m1, a, b)); CK(hipEventElapsedTime(&m2, a, c));
    printf("two complementary masked streams, 128 blocks each, ~0.83 ms of spin per block: s1 done %.3f ms, s2 done %.3f ms\n", m1, m2);
  }
  return 0;
}
