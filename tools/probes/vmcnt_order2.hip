// Probe 2: the wait pattern of conv3x3_m16.hip in isolation.  Per wave, in program order:
//   D_a (2 LDS-DMA pieces, cold source) ; L1 (buffer_load, hot) ; D_b (2 LDS-DMA pieces, cold) ; L2 (buffer_load, hot) ;
//   s_waitcnt vmcnt(4)   -- "all but the 4 youngest": D_a must have landed ;  read D_a's LDS words.
// (No instruction offsets on the DMAs: the offset field is added to the LDS address as well as to the global one.)
// The LDS words are preset to a sentinel; a sentinel read back = D_a had NOT landed although 4 or fewer ops were outstanding.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) probe(const float* cold, const float* hot, int* bad, long long cold_elems, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 1024];  // per wave 4 KB: 2 pieces x 1 KB for D_a, 2 for D_b
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long hb = (unsigned long long)(size_t)hot;
  i32x4 rh = {(int)(unsigned)hb, (int)((hb >> 32) & 0xffff), (int)0xfffffff0u, 0x00020000};
  rh.x = __builtin_amdgcn_readfirstlane(rh.x); rh.y = __builtin_amdgcn_readfirstlane(rh.y);
  float* my = lds + wave * 1024;
  int nbad = 0, na = 0, nb = 0, nr = 0;
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < 4; ++k) reinterpret_cast<f32x4*>(my)[lane + 64 * k] = f32x4{-12345.f, -12345.f, -12345.f, -12345.f};
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long g = ((unsigned long long)blockIdx.x * 977 + it * 131071ull + wave * 8191ull) * 4096ull + lane * 16;
    const char* c0 = reinterpret_cast<const char*>(cold) + (g % (unsigned long long)(cold_elems * 4 - 65536));
    const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)my);
    f32x4 r1, r2;
    const unsigned hoff = (lane & 3) * 16;
    asm volatile(
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
        "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, off\n\t"
        "buffer_load_dwordx4 %0, %3, %5, 0 offen\n\t"
        "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, off\n\t"
        "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, off\n\t"
        "buffer_load_dwordx4 %1, %3, %5, 0 offen\n\t"
        "s_waitcnt vmcnt(4)"
        : "=&v"(r1), "=&v"(r2) : "v"(c0), "v"(hoff), "s"(l0), "s"(rh), "v"(c0 + 2048), "v"(c0 + 8192), "v"(c0 + 12288) : "memory");
    f32x4 a = reinterpret_cast<f32x4*>(my)[lane], b = reinterpret_cast<f32x4*>(my)[lane + 64];
    na += (a.x == -12345.f); nb += (b.x == -12345.f);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r1), "+v"(r2)::"memory");
    nr += (r1.x != 0.f) + (r2.x != 0.f);
  }
  if (na) atomicAdd(bad, na);
  if (nb) atomicAdd(bad + 1, nb);
  if (nr) atomicAdd(bad + 2, nr);
}
int main() {
  const long long cold_elems = 1ll << 30;
  float *cold, *hot; int* bad;
  (void)hipMalloc(&cold, cold_elems * 4); (void)hipMalloc(&hot, 4096); (void)hipMalloc(&bad, 16);
  (void)hipMemset(cold, 0x3f, cold_elems * 4); (void)hipMemset(hot, 0, 4096); (void)hipMemset(bad, 0, 16);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(256 * 8), dim3(512), 0, 0, cold, hot, bad, cold_elems, 200);
  int h[4] = {-1, -1, -1, -1}; (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
  printf("pattern D D L D D L ; vmcnt(4): piece0 not landed %d, piece1 not landed %d, register loads wrong %d  (of %lld lane-iterations)\n", h[0], h[1], h[2], 3ll * 256 * 8 * 8 * 64 * 200);
  return 0;
}
