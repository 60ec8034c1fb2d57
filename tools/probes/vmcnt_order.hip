// Probe: do vector-memory loads of one wave retire in issue order on gfx950?
// Each wave issues load A (cold, far apart: HBM miss), then load B (hot: the same 64 B for everyone), waits
// s_waitcnt vmcnt(1) -- "all but the youngest are done" -- and copies A's destination register out.  A's register
// is preset to a sentinel; a sentinel in the output means B retired before A (out of order).
// variant 0: A = buffer_load, B = buffer_load;  1: A = buffer_load, B = global_load_lds (LDS-DMA);
// variant 2: A = global_load, B = global_load_lds; 3: A = global_load, B = global_load
// variant 4: A = global_load_lds (cold source -> LDS, preset to the sentinel), B = buffer_load (hot): is the LDS word there at vmcnt(1)?
// variant 5: same with B = global_load
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ void probe(const float* cold, const float* hot, float* out, long long cold_elems) {
  __shared__ float lds[64 * 4];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long cb = (unsigned long long)(size_t)cold, hb = (unsigned long long)(size_t)hot;
  i32x4 rc = {(int)(unsigned)cb, (int)((cb >> 32) & 0xffff), (int)0xfffffff0u, 0x00020000};
  i32x4 rh = {(int)(unsigned)hb, (int)((hb >> 32) & 0xffff), (int)0xfffffff0u, 0x00020000};
  rc.x = __builtin_amdgcn_readfirstlane(rc.x); rc.y = __builtin_amdgcn_readfirstlane(rc.y);
  rh.x = __builtin_amdgcn_readfirstlane(rh.x); rh.y = __builtin_amdgcn_readfirstlane(rh.y);
  const unsigned coff = (unsigned)(((unsigned long long)gid * 4099ull * 64ull) % (unsigned long long)(cold_elems - 64)) * 4u & ~15u;
  const unsigned hoff = (threadIdx.x & 3) * 16;
  float a = -12345.0f, b = 0.f;
  const float* cptr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(cold) + coff);
  const float* hptr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(hot) + hoff);
  const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
  if (V == 0) {
    asm volatile("buffer_load_dword %0, %2, %3, 0 offen\n\tbuffer_load_dword %1, %4, %5, 0 offen\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0"
                 : "+v"(a), "+v"(b) : "v"(coff), "s"(rc), "v"(hoff), "s"(rh) : "memory");
  } else if (V == 1) {
    asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dword %0, %2, %3, 0 offen\n\tglobal_load_lds_dword %4, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0"
                 : "+v"(a), "+v"(b) : "v"(coff), "s"(rc), "v"(hptr), "s"(ldsaddr) : "memory");
  } else if (V == 2) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_dword %0, %2, off\n\tglobal_load_lds_dword %3, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0"
                 : "+v"(a), "+v"(b) : "v"(cptr), "v"(hptr), "s"(ldsaddr) : "memory");
  } else if (V == 4 || V == 5) {
    lds[threadIdx.x & 63] = -12345.0f;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned la = ldsaddr;  // wave-uniform base: lane i lands at base + 4 i
    if (V == 4)
      asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\tbuffer_load_dword %1, %3, %5, 0 offen\n\ts_waitcnt vmcnt(1)\n\tds_read_b32 %0, %6\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(a), "+v"(b) : "v"(cptr), "v"(hoff), "s"(la), "s"(rh), "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 4) : "memory");
    else
      asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\tglobal_load_dword %1, %3, off\n\ts_waitcnt vmcnt(1)\n\tds_read_b32 %0, %5\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(a), "+v"(b) : "v"(cptr), "v"(hptr), "s"(la), "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 4) : "memory");
    b = a;
  } else {
    float t = 0.f;
    asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %2, %4, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                 : "+v"(a), "+v"(b), "+v"(t) : "v"(cptr), "v"(hptr) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[gid] = b;
}
int main() {
  const long long cold_elems = 1ll << 30;  // 4 GB
  float *cold, *hot, *out;
  const int blocks = 256 * 64, threads = 256;
  hipMalloc(&cold, cold_elems * 4); hipMalloc(&hot, 4096); hipMalloc(&out, (size_t)blocks * threads * 4);
  hipMemset(cold, 0x3f, cold_elems * 4); hipMemset(hot, 0, 4096);
  float* h = (float*)malloc((size_t)blocks * threads * 4);
  for (int v = 0; v < 6; ++v) {
    long long bad = 0;
    for (int rep = 0; rep < 5; ++rep) {
      if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, cold, hot, out, cold_elems);
      if (v == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, cold, hot, out, cold_elems);
      if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(threads), 0, 0, cold, hot, out, cold_elems);
      if (v == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(threads), 0, 0, cold, hot, out, cold_elems);
      if (v == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(64), 0, 0, cold, hot, out, cold_elems);
      if (v == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(64), 0, 0, cold, hot, out, cold_elems);
      hipMemcpy(h, out, (size_t)blocks * threads * 4, hipMemcpyDeviceToHost);
      for (long long i = 0; i < (long long)blocks * threads; ++i) bad += h[i] == -12345.0f;
    }
    printf("variant %d: %lld of %lld lanes saw the sentinel (older load not back at vmcnt(1))\n", v, bad, 5ll * blocks * threads);
  }
  return 0;
}
