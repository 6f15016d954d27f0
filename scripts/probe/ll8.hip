// Probe (hipcc --offload-arch=gfx950 -O2 ll8.hip -o ll8; run on the GPU box): 8-byte "data + tag" units written by one
// workgroup (workgroup-scope atomic store) and polled by another workgroup of the same XCD with 16-byte sc1 buffer loads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long* ring, unsigned* out, int* go) {
  const int lane = threadIdx.x;
  if (blockIdx.x == 0) {   // producer: 64 units, data = 0xD000 + lane, tag = 0x1001
    while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
    const unsigned long long unit = (unsigned long long)(0xD000u + lane) | (0x1001ull << 32);
    __hip_atomic_store(ring + lane, unit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else if (blockIdx.x == 8) {   // consumer on the same XCD: lane l polls units 2l, 2l+1 (l < 32)
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, 512, 0x00020000);
    if (lane == 0) __hip_atomic_store(go, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    i32x4 v;
    for (;;) {
      asm volatile("" ::: "memory");
      v = __builtin_amdgcn_raw_buffer_load_b128(r, lane < 32 ? lane * 16 : 0xFFFFFFF0u, 0, 16);
      const bool miss = lane < 32 && (v[1] != 0x1001 || v[3] != 0x1001);
      if (__ballot(miss) == 0ull || ++spins > (1 << 20)) break;
      __builtin_amdgcn_s_sleep(1);
    }
    for (int q = 0; q < 4; q++) out[lane * 4 + q] = (unsigned)v[q];
    if (lane == 0) out[256] = (unsigned)spins;
  }
}
int main() {
  unsigned long long* ring; unsigned* out; int* go;
  hipMalloc(&ring, 512); hipMemset(ring, 0, 512); hipMalloc(&out, 4096); hipMemset(out, 0, 4096); hipMalloc(&go, 4); hipMemset(go, 0, 4);
  hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, 0, ring, out, go);
  unsigned h[257]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  printf("spins %u\n", h[256]);
  for (int l = 0; l < 4; l++) printf("lane %d: %x %x %x %x\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
