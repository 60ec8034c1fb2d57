// 3x3 / stride 1 / pad 1 conv, up to 128 output channels per workgroup, FOUR waves and TWO workgroups per CU (round 3).
// (included by conv3x3_m16.hip; the fp32-input sibling of conv3x3_w4.h)   OPT-IN (PRV2_Q4=1 / 2): faster than the 8-wave kernels on dense random
// operands, slower on the frame's real activations -- see conv3x3_q4_usable() in conv3x3_m16.hip and profiles/r03_experiments.txt.
//
// The 8-wave kernels of conv3x3_m16.hip hold one 157 KB workgroup per CU: on the decoder layers behind a x2 upsample (256 -> 128 at
// 384 x 512, 98 -> 98, 194 -> 194, ...: 51 ms of the frame at 0.45 of nominal) a tile is 28 ... 72 steps and its prologue, epilogue
// and stalls run with the MFMA pipe idle -- these layers are CYCLE-bound (same time on all-zero operands, tools/probes/
// power_or_cycles.py), unlike the 256-column kernels.  Here a workgroup is 4 waves on an 8 x 16 pixel tile with 70.5 KB of LDS, two per
// CU in different phases:
//   * wave w = image rows 2w, 2w + 1 (2 pixel runs of 16) x ALL 128 columns: 2 x 8 accumulators of v_mfma_f32_16x16x32_bf16 computed
//     transposed (A = weights, B = pixels): a lane holds 32 channels of one pixel, four lanes a pixel -- bias, LayerNorm, activation,
//     residual and the stores are wave-local (no C tile in LDS, no epilogue barrier); a lane's accumulators 2s, 2s + 1 are 8
//     consecutive channels (output channel order of conv3x3_w4.h): 32-byte stores, 128 B per pixel over the four lanes;
//   * a step = (32-channel slab, tap): weight tile 128 rows x 128 B by LDS-DMA, three rotating buffers, two steps ahead;
//   * ONE halo buffer (10 x 18 pixels x 128 B, slots swizzled by the pixel index): the next slab is loaded into registers as six half
//     items per thread (4 channels of a pixel: one 16-byte load, or four for the taps of the fused x2 upsample), one per step, each
//     converted to bf16 hi / lo a step later (4 registers), and written behind the barrier of the slab's last step -- the last tap's
//     fragments are read one step early;
//   * the im2col tail tile of the Cin = 32 k + 2 layers is one more step, its B fragments gathered from the staged last slab.
// Arithmetic: products and their order per accumulator, the loader's interpolation (upsample_bilinear_kernel's operation order) and
// the epilogue's formulas are those of conv3x3_m16.hip; LayerNorm statistics as ln_row_stats (four partial sums per pixel).
#pragma once

namespace prv2 {

namespace q4 {
constexpr int BN = 128, TH = 8, TW = 16, HWP = TW + 2;
constexpr int HALO = (TH + 2) * HWP;  // 180 halo pixels
constexpr int A_BYTES = HALO * 128, B_BYTES = BN * 128, NBUF = 3;
constexpr int SMEM_BYTES = NBUF * B_BYTES + A_BYTES;  // 70.5 KB
constexpr int META_BYTES = 3 * 3 * 256 * 4;           // fused-upsample variant: tap offset / weights of a thread's three items (9 KB)
constexpr int NWD = 4;                                // weight DMA instructions per wave and tile
constexpr int NIT = 3;                                // halo items (pixel, 8 channels) per thread and slab: 720 over 256 threads
static_assert((SMEM_BYTES + META_BYTES) * 2 <= 160 * 1024, "two workgroups per CU");
}  // namespace q4

// same partial-sum tree as ln_row_stats (igemm.h): lane g of a pixel sums its channels in ascending order, then (p0 + p1) + (p2 + p3)
template <bool UPS, bool TAIL>
__global__ void __launch_bounds__(256, 2) conv3x3_q4_kernel(const IgemmParams p) {
  using namespace q4;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES + (UPS ? META_BYTES : 0)];
  char* const Bs_b = smem;
  constexpr int A_OFF = NBUF * B_BYTES;

  // ---- XCD-aware block -> (pixel tile, column tile): the column tiles of a pixel tile are neighbours (they share the halo in L2) ----
  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
  int t = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  const int tile_n = t % p.tiles_n;
  t /= p.tiles_n;
  const int tx = t % tiles_x;
  const int ty = (t / tiles_x) % tiles_y;
  const int n_img = t / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, g = lane >> 4;

  // ---- halo items ------------------------------------------------------------------------------------------------------
  constexpr unsigned OOB = 0x80000000u;
  const int cin4 = (p.Cin + 3) & ~3;
  const int ups_slabs = UPS ? p.ups_c / 32 : 0;
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x + (long long)n_img * p.x_bstride), 0, (int)((((long long)p.H * p.W - 1) * p.ldx + p.Cin) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(UPS ? p.xu + (long long)n_img * p.xu_bstride : p.x), 0, UPS ? (int)((((long long)p.uH * p.uW - 1) * p.ldxu + p.ups_c) * 4) : 0, 0x00020000);
  // Item it of a thread = (halo pixel hp, 8-channel group q) with 4 hp + q = tid + 256 it; it is fetched as two HALVES of 4 channels
  // (sub-item k = 2 it + half: one 16-byte load, or four for the taps of an interpolated slab).  Everything an item needs -- source
  // offsets, tap geometry, interpolation weights -- is recomputed from the thread id when the half is requested (a few dozen VALU
  // per slab against ~500 MFMAs): kept across the slab loop it cost 15+ registers and the interpolating variant spilled.
  const unsigned u_dx = UPS ? (unsigned)(p.ldxu * 4) : 0u, u_dy = UPS ? (unsigned)(p.uW * p.ldxu * 4) : 0u;
  const int relu_floor = p.relu_in ? 0 : (int)0x80000000;
  typedef int i32x4v __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
  f32x4 raw[UPS ? 4 : 1];        // the half in flight
  float w1x = 0.f, w1y = 0.f;    // its interpolation weights
  bf16x4v cvh[2 * NIT], cvl[2 * NIT];  // converted halves of the next slab
  int tidv = tid;                // (made opaque per slab)
  // interpolated slabs: tap (y0, x0)'s byte offset in the low-resolution image (or OOB) with "x1 = x0 + 1" / "y1 = y0 + 1" in bits 0 / 1
  // (align_corners clamps them at the far edge), and the two fractional weights -- ac_tap's values, once per item, in LDS
  unsigned* const meta = reinterpret_cast<unsigned*>(smem + SMEM_BYTES) + tid;  // [item][3][256]
  if constexpr (UPS) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it, hp = idx >> 2;
      const int hy = hp / HWP, hx = hp - hy * HWP;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = hp < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const AxisTap ay = ac_tap(ok ? iy : 0, p.usy, p.uH), ax = ac_tap(ok ? ix : 0, p.usx, p.uW);
      meta[(it * 3 + 0) * 256] = (ok ? (unsigned)(((ay.i0 * p.uW + ax.i0) * p.ldxu + (idx & 3) * 8) * 4) : OOB) | (ax.i1 != ax.i0 ? 1u : 0u) | (ay.i1 != ay.i0 ? 2u : 0u);
      meta[(it * 3 + 1) * 256] = __builtin_bit_cast(unsigned, ax.w1);
      meta[(it * 3 + 2) * 256] = __builtin_bit_cast(unsigned, ay.w1);
    }
  }
  auto item_load = [&](int cc, int k, bool ups) {  // (ups: slab-uniform)
    if (UPS && ups) {
      const unsigned m0 = meta[((k >> 1) * 3 + 0) * 256];
      w1x = __builtin_bit_cast(float, meta[((k >> 1) * 3 + 1) * 256]);
      w1y = __builtin_bit_cast(float, meta[((k >> 1) * 3 + 2) * 256]);
      const unsigned o0 = (m0 & ~3u) + (unsigned)(cc * 128 + (k & 1) * 16);  // (an OOB base stays out of range with the tap offsets added)
      const unsigned dx = (m0 & 1u) ? u_dx : 0u, dy = (m0 & 2u) ? u_dy : 0u;
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
        raw[UPS ? tp : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rs, o0 + ((tp & 1) ? dx : 0u) + ((tp & 2) ? dy : 0u), 0, 0));
    } else {
      int tv = tidv;
      asm volatile("" : "+v"(tv));  // (opaque HERE: the geometry below is formed when the half is requested, not at the top of the slab)
      const int idx = tv + 256 * (k >> 1), hp = idx >> 2, c8 = (idx & 3) * 8 + (k & 1) * 4;
      const int hy = hp / HWP, hx = hp - hy * HWP;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = hp < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned o = ok && cc * 32 + c8 < cin4 ? (unsigned)(((iy * p.W + ix) * p.ldx + cc * 32 + c8) * 4) : OOB;
      raw[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, o, 0, 0));
    }
  };
  auto item_convert = [&](int k, bool ups) {
    f32x4 v = raw[0];
    if (UPS && ups) {  // upsample_bilinear_kernel's operation order (gather.hip), element by element
      const float w0x = 1.0f - w1x, w0y = 1.0f - w1y;
      f32x4 top = {0.f, 0.f, 0.f, 0.f}, bot = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        top[e] += w0x * raw[0][e];
        top[e] += w1x * raw[UPS ? 1 : 0][e];
        bot[e] += w0x * raw[UPS ? 2 : 0][e];
        bot[e] += w1x * raw[UPS ? 3 : 0][e];
        r[e] += w0y * top[e];
        r[e] += w1y * bot[e];
      }
      v = r;
    }
    i32x4v vi = __builtin_bit_cast(i32x4v, v);
    vi.x = max(vi.x, relu_floor);  // ReLU on the bits: negative floats are negative ints; floor INT_MIN = identity
    vi.y = max(vi.y, relu_floor);
    vi.z = max(vi.z, relu_floor);
    vi.w = max(vi.w, relu_floor);
    bf16x4 hh, ll;
    split_bf16(__builtin_bit_cast(f32x4, vi), hh, ll);
    cvh[k] = hh;
    cvl[k] = ll;
  };
  auto items_store = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it, hp = idx >> 2, q = idx & 3;
      if (idx < HALO * 4) {
        const unsigned a = (unsigned)(A_OFF + hp * 128 + ((q ^ (hp & 7)) << 4));  // bf16 hi of the item's 8 channels; lo: ^ 64
        *reinterpret_cast<bf16x8*>(smem + a) = __builtin_shufflevector(cvh[2 * it], cvh[2 * it + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8*>(smem + (a ^ 64u)) = __builtin_shufflevector(cvl[2 * it], cvl[2 * it + 1], 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
  };

  // ---- weight DMA (conv3x3_w4.h): LDS row R = 16 jj + m of the tile = output channel perm(jj, m) of this column tile ---------
  const long long w_row_stride = 9LL * p.Cin_pad;  // floats per packed row
  unsigned wsrc[NWD];
  auto tile_row = [&](int i, int& c, int& sw, int ln) {
    const int R = (4 * wave + i) * 8 + (ln >> 3), slot = ln & 7;
    const int jj = R >> 4, m = R & 15;
    c = tile_n * BN + 32 * (jj >> 1) + 8 * (m >> 2) + 4 * (jj & 1) + (m & 3);
    sw = (slot ^ ((m >> 1) & 7) ^ ((c >> 1) & 7)) << 4;
  };
#pragma unroll
  for (int i = 0; i < NWD; ++i) {
    int c, sw;
    tile_row(i, c, sw, lane);
    wsrc[i] = (unsigned)(((long long)c * w_row_stride) * 4 + sw);
  }
  auto dma_issue = [&](const char* base, unsigned voff, int bbuf, int i) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(Bs_b + bbuf * B_BYTES + (4 * wave + i) * 1024));
    const unsigned long long b = (unsigned long long)(size_t)base;
    const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sb) : "memory");
  };
  const int cslabs = p.Cin_pad / 32;            // slabs staged through the halo buffer
  const int cchunks = cslabs - (TAIL ? 1 : 0);  // slabs walked by the 9-tap loop
  // tile of global step s = 9 cc + tap (s == 9 cchunks: the tail tile; beyond the last step: the last tile again, nobody reads it)
  auto dma_step = [&](int s, int bbuf) {
    const int last = 9 * cchunks + (TAIL ? 1 : 0) - 1;
    s = s < last ? s : last;
    if (TAIL && s == 9 * cchunks) {
      int ln = lane;
      asm volatile("" : "+v"(ln));  // (opaque: the row offsets are formed HERE, twice per workgroup, not hoisted and kept across the loop)
#pragma unroll
      for (int i = 0; i < NWD; ++i) {
        int c, sw;
        tile_row(i, c, sw, ln);
        dma_issue(reinterpret_cast<const char*>(p.w_tail), (unsigned)(c * 128 + sw), bbuf, i);
      }
    } else {
      const int cc = s / 9, tap = s - cc * 9;
      const char* base = reinterpret_cast<const char*>(p.w) + ((long long)tap * p.Cin_pad + cc * 32) * 4;
#pragma unroll
      for (int i = 0; i < NWD; ++i) dma_issue(base, wsrc[i], bbuf, i);
    }
  };

  // ---- fragments ---------------------------------------------------------------------------------------------------------
  const int w_off = m16 * 128 + ((g ^ ((m16 >> 1) & 7)) << 4);  // bf16 hi of the lane's weight fragment; + 2048 jj; lo: ^ 64
  const int hp0 = 2 * wave * HWP + m16;
  int hp0v = hp0;  // (opaque per slab: see conv3x3_w4.h)
  bf16x8 xh[2], xl[2], wh[3], wl[3];
  auto read_x = [&](int tap) {
    const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int hp = hp0v + (f + dy) * HWP + dx;
      const int a = A_OFF + hp * 128 + ((g ^ (hp & 7)) << 4);
      xh[f] = *reinterpret_cast<const bf16x8*>(smem + a);
      xl[f] = *reinterpret_cast<const bf16x8*>(smem + (a ^ 64));
    }
  };
  auto read_w = [&](int slot, int bbuf, int jj) {
    wh[slot] = *reinterpret_cast<const bf16x8*>(smem + w_off + bbuf * B_BYTES + jj * 2048);
    wl[slot] = *reinterpret_cast<const bf16x8*>(smem + (w_off ^ 64) + bbuf * B_BYTES + jj * 2048);
  };
  auto mma6 = [&](f32x4& c0, f32x4& c1, const bf16x8& wh_, const bf16x8& wl_) {  // x_lo w_hi, x_hi w_lo, x_hi w_hi (conv3x3_m16.hip's order)
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_, xl[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_, xl[1], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl_, xh[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl_, xh[1], c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_, xh[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_, xh[1], c1, 0, 0, 0);
  };
  f32x4 acc[2][8];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma_tile = [&](int bbuf) {
    read_w(0, bbuf, 0);
    read_w(1, bbuf, 1);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      if (jj + 2 < 8) read_w((jj + 2) % 3, bbuf, jj + 2);
#ifdef PRV2_Q4_SCHED_BARRIERS
      __builtin_amdgcn_sched_barrier(0);
#endif
      mma6(acc[0][jj], acc[1][jj], wh[jj % 3], wl[jj % 3]);
#ifdef PRV2_Q4_SCHED_BARRIERS
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };

  // ---- prologue: the first two weight tiles, slab 0 -----------------------------------------------------------------------------
  dma_step(0, 0);
  dma_step(1, 1);
#pragma unroll
  for (int k = 0; k < 2 * NIT; ++k) {
    item_load(0, k, 0 < ups_slabs);
    item_convert(k, 0 < ups_slabs);
  }
  items_store();

  // ---- main loop: step u of slab cc = tap u, weight tile in buffer (9 cc + u) % 3 = u % 3 --------------------------------------------
  for (int cc = 0; cc < cchunks; ++cc) {
    const int ccn = cc + 1 < cslabs ? cc + 1 : cc;  // behind the last slab: the same slab again, nobody reads it (uniform VMEM counts)
    hp0v = hp0;
    tidv = tid;
    asm volatile("" : "+v"(hp0v), "+v"(tidv));
    auto slab = [&](auto nu_c) {
      constexpr bool NU = decltype(nu_c)::value;  // the NEXT slab's channels are interpolated: 4 loads per half item instead of 1
      constexpr int L = NU ? 4 : 1;
      auto step = [&](auto u_c) {
        constexpr int u = decltype(u_c)::value;
        // Program order of a step's VMEM instructions: the DMA of step u + 2, then (steps 0 .. 5) the loads of half item u.  Issued
        // after this step's weight tile (two steps ago), i.e. allowed to be in flight here: step u - 2's half item, step u - 1's DMA
        // and half item
        constexpr int newer = NWD + (u >= 2 && u <= 7 ? L : 0) + (u >= 1 && u <= 6 ? L : 0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(newer) : "memory");
        if constexpr (u < 8) read_x(u);  // (tap 8: read at the end of step 7 -- the halo buffer is rewritten during step 8)
        else items_store();
        // the half requested a step ago, BEFORE this step's DMA is issued: the compiler's wait for its loads (it does not know the
        // DMAs) is a vmcnt(0), which then only covers instructions that are at least a step old
        if constexpr (u >= 1 && u <= 6) item_convert(u - 1, NU);
        dma_step(9 * cc + u + 2, (u + 2) % 3);
        if constexpr (u <= 5) item_load(ccn, u, NU);
        mma_tile(u % 3);
        if constexpr (u == 7) read_x(8);
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
      step(std::integral_constant<int, 8>{});
    };
    if (UPS && ccn < ups_slabs) slab(std::true_type{});  // block-uniform
    else slab(std::false_type{});
  }
  if constexpr (TAIL) {
    // ---- tail step: k = 2 tap + c over the last slab's two channels (the first 4 bytes of a pixel's hi / lo plane); lane group g
    // covers taps 4g .. 4g + 3.  Its weight tile (step 9 cchunks) was issued at step 7 of the last slab; nothing was issued behind it
    // that is still needed, the staged slab was published by that slab's step-8 store + this barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    int hpt = hp0;
    asm volatile("" : "+v"(hpt));  // (opaque: 16 gather addresses would otherwise be formed in front of the slab loop and spilled)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      u32x4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tp = 4 * g + i, tpc = tp < 9 ? tp : 0;
        const int dy = tpc / 3, dx = tpc - dy * 3;
        const int hp = hpt + (f + dy) * HWP + dx;
        const unsigned hv = *reinterpret_cast<const unsigned*>(smem + A_OFF + hp * 128 + ((0 ^ (hp & 7)) << 4));
        const unsigned lv = *reinterpret_cast<const unsigned*>(smem + A_OFF + hp * 128 + ((4 ^ (hp & 7)) << 4));
        h[i] = tp < 9 ? hv : 0u;
        l[i] = tp < 9 ? lv : 0u;
      }
      xh[f] = __builtin_bit_cast(bf16x8, h);
      xl[f] = __builtin_bit_cast(bf16x8, l);
    }
    mma_tile((9 * cchunks) % 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the clamped DMAs of the last steps: nothing of this workgroup may land later)

  // ---- epilogue, wave-local ----------------------------------------------------------------------------------------------------
  // lane's channels of accumulator j: col(j) + e, col(j) = 128 tile_n + 32 (j >> 1) + 8 g + 4 (j & 1)
  const int colb = tile_n * BN + 8 * g;
  auto col_of = [&](int j) { return colb + 32 * (j >> 1) + 4 * (j & 1); };
  auto load4 = [&](const float* q, int col) {  // q[col .. col + 3], zeros behind Ncols
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col + 3 < p.Ncols) v = *reinterpret_cast<const f32x4*>(q + col);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < p.Ncols) v[e] = q[col + e];
    }
    return v;
  };
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 b = load4(p.bias, col_of(j));
      acc[0][j] += b;
      acc[1][j] += b;
    }
  }
  float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
  const bool has_ln = p.ln_w != nullptr;  // block-uniform (host: one column tile)
  if (has_ln) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col_of(j) + e < p.Cout) s += acc[f][j][e];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      mean[f] = s / (float)p.Cout;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col_of(j) + e < p.Cout) {
            const float d = acc[f][j][e] - mean[f];
            q += d * d;
          }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      rstd[f] = 1.0f / sqrtf(q / (float)p.Cout + p.ln_eps);
    }
  }
  const long long img_m = (long long)n_img * p.H * p.W;
  int pix[2];
  bool inside[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int oy = y0 + 2 * wave + f, ox = x0 + m16;
    inside[f] = oy < p.H && ox < p.W;
    pix[f] = min(oy, p.H - 1) * p.W + min(ox, p.W - 1);
  }
  dispatch_act(p.act, [&](auto act_c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = col_of(j);
      if (col >= p.Ncols) continue;
      f32x4 lw = {1.f, 1.f, 1.f, 1.f}, lb = {0.f, 0.f, 0.f, 0.f};
      if (has_ln) {
        lw = load4(p.ln_w, col);
        lb = load4(p.ln_b, col);
      }
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        f32x4 rv = {0.f, 0.f, 0.f, 0.f};
        if (p.res) rv = load4(p.res + (img_m + pix[f]) * p.ld_res, col);
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float tv = acc[f][j][e];
          if (has_ln) tv = (tv - mean[f]) * rstd[f] * lw[e] + lb[e];
          tv = act_apply_bf(tv, decltype(act_c)::value);
          if (p.res) tv += rv[e];
          ov[e] = col + e < p.Ncols ? tv : 0.f;  // pad channels behind cout stay zero
        }
        if (inside[f]) {
          float* dst = p.y + (long long)n_img * p.y_bstride + (long long)pix[f] * p.ldy + col;
          *reinterpret_cast<f32x4*>(dst) = ov;
        }
      }
    }
  });
}

}  // namespace prv2
