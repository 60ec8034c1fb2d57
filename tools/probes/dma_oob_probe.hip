// GPU probe: does `buffer_load_dwordx4 ... offen lds` (gfx950 LDS-DMA through a buffer resource) write ZEROS for lanes whose
// offset is out of the resource's range, and skip EXEC-masked lanes?   hipcc --offload-arch=gfx950 dma_oob_probe.hip -o dma_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int n_floats) {
  __shared__ float s[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) s[i] = -7.f;
  __syncthreads();
  i32x4 rsrc;
  unsigned long long b = (unsigned long long)(size_t)x;
  rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));
  rsrc.z = __builtin_amdgcn_readfirstlane(n_floats * 4);
  rsrc.w = 0x00020000;
  // lanes 0-15 in range (permuted), 16-31 out of range (0x80000000 + small), 32-47 in range, 48-63 just past the end
  const int l = threadIdx.x;
  unsigned voff = l < 16 ? (15 - l) * 16 : l < 32 ? 0x80000000u + l * 16 : l < 48 ? l * 16 : (unsigned)(n_floats * 4) + (l - 48) * 16;
  unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)s);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff), "s"(rsrc) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) y[i] = s[i];
}
int main() {
  float *x, *y, hx[64 * 4], hy[256];
  for (int i = 0; i < 256; ++i) hx[i] = 100.f + i;
  hipMalloc(&x, sizeof(hx));
  hipMalloc(&y, sizeof(hy));
  hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, 48 * 4);  // the resource covers 48 lanes' worth
  hipMemcpy(hy, y, sizeof(hy), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %6.1f %6.1f %6.1f %6.1f%s", l, hy[4 * l], hy[4 * l + 1], hy[4 * l + 2], hy[4 * l + 3], (l & 3) == 3 ? "\n" : "   ");
  return 0;
}
