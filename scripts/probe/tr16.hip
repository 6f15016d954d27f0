// Probe (hipcc --offload-arch=gfx950 -O2 tr16.hip -o tr16; run on the GPU box): which LDS element does lane l, element j of ds_read_b64_tr_b16 return when lane l passes base + l*8 bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = (short)i;
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = r[j];
}
int main() {
  short* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) printf("lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
