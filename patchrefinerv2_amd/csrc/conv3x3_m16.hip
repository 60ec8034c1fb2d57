// 3x3 / stride 1 / pad 1 convolution, LDS halo tile, bf16 modes on v_mfma_f32_16x16x32_bf16.
//
// Same decomposition as conv3x3.hip (8 x 32 output pixels x BN channels per 512-thread workgroup, a
// 32-channel slab of the (8+2) x (32+2) halo staged once and reused by the nine taps, weight tile of each
// tap by LDS-DMA), but the MFMA is the 16x16x32 shape: one instruction spans the whole 32-channel slab of a
// tap, and on MI355X the chip holds a higher clock on this shape than on 32x32x16 for the same FLOPs
// (MI355X_MICROARCH.md, DVFS give-back (7)); the 3x3 convs are MFMA-dense enough to be clock-limited
// (profiles/: 1.66 GHz effective, MFMA pipe 72 % busy, on the 32x32x16 kernel).
//
// Fragment layout (lane = 16*g + m): A operand = halo pixel m of a 16-pixel run, channels 8g..8g+7;
// B operand = output channel m of a 16-channel column, input channels 8g..8g+7; accumulator = pixels
// 4g..4g+3 x channel m.  A wave owns NI image rows x 32 pixels (NA = 2*NI pixel runs) x NJ columns.
//
// LDS: halo rows are 160 bytes (32 bf16 hi | 32 bf16 lo | 32 B pad): with a 10-slot row stride the 16 lanes
// of every ds_read_b128 group (pixels p..p+15 at slot g, mixed g) hit 16 distinct 16-byte bank slots for
// every tap shift (exhaustive check in tests/test_host_logic.py).  Weight rows are the same unpadded,
// XOR-swizzled 128-byte rows as in conv3x3.hip (slot q of row r at q ^ ((r >> 1) & 7)) -- also conflict
// free for this access pattern.  2 halo buffers (106 KB) + 3 weight buffers (48 KB).
//
// Pipeline per (slab, tap) step, NJ column phases of NA*3 MFMAs each:
//   phase j < NJ-1 : read column j+1's weights, then the MFMAs of column j
//   barrier        : before the last phase (everything in LDS that the rest needs is now a step old)
//   last phase     : read column 0 of the NEXT step's weights; MFMAs of the last column, and as each pixel
//                    run retires, its fragment registers are refilled with the NEXT tap's run.
// Global traffic is two steps ahead of its first use: the weight DMA of step s+3 goes out right after the
// barrier of step s (all waves have read step s's tile by then) and has to have landed at the barrier of
// step s+2; the halo item loaded at tap t is stored at tap t+2.  All waits are counted (s_waitcnt vmcnt(N)).
//
// PRV2_ABL_NOA / _NOB / _NOBAR compile out the halo stream / the weight DMA / the barrier for timing ablations
// (tools/ab_conv.sh; results are wrong with any of them, and stale LDS data raises the clock by itself --
// compare cycles via tools/pmc_conv.sh, not only wall time).
#include <cstdlib>
#include <type_traits>

#include "igemm.h"

namespace prv2 {

namespace m16 {
constexpr int HALO = 10 * 34;                   // halo pixels of an 8 x 32 tile -- and of a 32 x 8 one
constexpr int A_IT = (HALO * 8 + 511) / 512;    // float4 loads per thread per slab (6)
constexpr int AROW = 160;                       // bytes per halo pixel in LDS
typedef float f32x4v __attribute__((ext_vector_type(4)));
}  // namespace m16

template <int BN, bool PERSIST = false, bool SINGLE = false>
constexpr int halo16_smem_floats() {
  constexpr int main_ = ((SINGLE ? 1 : 2) * m16::HALO * m16::AROW + 3 * BN * 128) / 4, epi = 256 * (BN + 4) + 2 * 256;
  return PERSIST ? main_ + epi : (main_ > epi ? main_ : epi);  // PERSIST: the C tile has LDS of its own
}

// one workgroup; `bid` of `nwg` = its index among the workgroups of its tile shape.
// PERSIST (BN = 32 without tail tile: the 64->32 / 128->32 layers at full resolution): the workgroup walks the tiles
// bid, bid + nwg, ... as ONE continuous slab sequence -- the halo of the next tile's first slab is loaded during the last
// slab of the current one, the weight DMAs wrap around, and the C tile has its own 38 KB of LDS, so the epilogue of a tile
// runs while the next tile's operands are in flight.  These tiles are 18-36 steps of 0.2 us: their fixed cost (workgroup
// turnaround + halo latency + store drain, ~11 us) was 75 % of the tile time.
// SINGLE (BN = 32): ONE halo buffer -- the next slab waits in registers (loaded during taps 0-5 as before) and is converted and
// stored behind the barrier of tap 8, when nobody reads the current slab any more; one extra barrier per slab.  66 KB of LDS and
// ~110 registers: TWO workgroups per CU.  The narrow layers run 18-36 steps of 384 MFMA cycles per tile against ~1.4 k cycles
// per step (barrier, DMA issue, fragment latency, and the in-order vmcnt that queues a tile's stores in front of the next
// halo): a second resident workgroup fills those gaps.
// GATE (BN = Cout = 32 or 128, a GatedConvUnit's fusion_conv: bi_directional_fusion_model.py:44-51,70-80): the epilogue goes on
// with the 1x1 gate on the tile -- normalise + activate + bf16 split of the C tile in place, gate GEMM from LDS with the (small)
// weight fragments in registers, sigmoid * mul (+ res) in the store loop -- see conv3x3_gate.hip, which does the same at 256
// channels on tiles of its own.
// UPS (BN = 128): the first p.ups_c input channels are bilinear(align_corners=True) samples of the low-resolution tensor p.xu,
// formed while the halo is staged: an item = 4 tap loads (hardware zero fill outside the image) combined with
// upsample_bilinear_kernel's exact arithmetic, so the layer equals "upsample into the concat buffer, then conv" bit for bit
// without the upsampled tensor ever being written (fusion_model.py:15-24, bi_directional_fusion_model.py:139-142,201).  Slabs behind
// ups_c keep the plain loader; their three extra tap slots are out-of-range loads (zeros, no memory traffic), so that every
// item is four VMEM instructions and the counted waits stay compile-time constants.
template <int BN, int PREC, bool TAIL, bool TALL, bool PERSIST = false, bool SINGLE = false, bool GATE = false, bool UPS = false>
__device__ __forceinline__ void halo16_body(const IgemmParams& p, float* smem, int bid, const int nwg) {
  using namespace m16;
  static_assert(!UPS || (!PERSIST && !SINGLE && !GATE && BN == 128), "UPS: the 128-column kernel");
  constexpr int LPI = UPS ? 4 : 1;  // buffer loads per halo item
  // Tile = 8 rows x 32 pixels, or (TALL) 32 rows x 8 pixels for the remainder strip of images whose width is
  // 32k + (1..8): same pixel count, same halo size, a pixel run of 16 is then 2 rows x 8 pixels.
  constexpr int TH = TALL ? 32 : 8, TW = TALL ? 8 : 32, HW_ = TW + 2;
  static_assert(PREC == PRV2_PREC_BF16X3 || PREC == PRV2_PREC_BF16, "bf16 modes only");
  // (SINGLE with BN = 64: a wave takes ONE image row x all 64 columns -- 2 runs x 4 columns, 16 fewer fragment registers than
  //  2 rows x 32 columns: the kernel has to fit 128 registers for two workgroups per CU)
  constexpr int WN = (BN >= 64 && !(SINGLE && BN == 64)) ? 2 : 1;
  constexpr int NI = (BN >= 64 && !(SINGLE && BN == 64)) ? 2 : 1;  // image rows per wave
  constexpr int NA = 2 * NI;                       // 16-pixel runs per wave
  constexpr int NJ = BN / (16 * WN);               // 16-channel columns per wave
  constexpr int ND = BN >= 64 ? BN / 64 : 1;       // LDS-DMA pieces (8 rows x 128 B) per wave per weight tile
  constexpr int A_BYTES = HALO * AROW;
  constexpr int B_BYTES = BN * 128;
  constexpr int CLD = BN + 4;
  constexpr int NBUF = 3;
  constexpr int NHB = SINGLE ? 1 : 2;  // halo buffers
  static_assert(NBUF == 3 && (NHB * A_BYTES + NBUF * B_BYTES) / 4 <= halo16_smem_floats<BN, PERSIST, SINGLE>() &&
                TH * TW * CLD + 2 * TH * TW <= halo16_smem_floats<BN, PERSIST, SINGLE>(), "LDS budget");  // main loop / C tile + LN statistics
  static_assert(!PERSIST || (!TAIL && !TALL && BN == 32), "PERSIST: plain 8 x 32 tiles of the BN = 32 kernel");
  static_assert(!SINGLE || (!PERSIST && BN <= 64), "SINGLE: the narrow kernels, two workgroups per CU");
  static_assert(!GATE || (!PERSIST && !TAIL && (BN == 32 || BN == 128)), "GATE: Cout = BN = 32 or 128");
  char* const As_b = reinterpret_cast<char*>(smem);
  char* const Bs_b = As_b + NHB * A_BYTES;
  float* const csm = PERSIST ? smem + (2 * A_BYTES + NBUF * B_BYTES) / 4 : smem;  // C tile (+ LN statistics)

  // ---- XCD-aware block -> (pixel tile, channel tile) ----------------------------------------
  const int tiles_x = TALL ? 1 : p.tiles_x, tiles_y = (p.H + TH - 1) / TH;  // (tiles_x excludes the strip columns [rx0, W))
  const int ntiles = PERSIST ? p.N * tiles_y * tiles_x * p.tiles_n : nwg;   // tiles of this shape in the launch
  struct Tile {
    int tile_n, n_img, y0, x0;
  };
  auto decode = [&](int t) {
    int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    Tile c;
    c.tile_n = t % p.tiles_n;
    int tm = t / p.tiles_n;
    const int tx = tm % tiles_x;
    tm /= tiles_x;
    const int ty = tm % tiles_y;
    c.n_img = tm / tiles_y;
    c.y0 = ty * TH;
    c.x0 = TALL ? p.rx0 : tx * TW;
    return c;
  };
  Tile tl = decode(bid);
  const int tile_n = tl.tile_n;  // (PERSIST: tiles_n == 1)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // Wave -> (pixel rows wm, channel half wn).  A workgroup's waves go to the SIMDs in cyclic order, so waves w and w + 4 share a
  // SIMD: with wn = wave >> 2 every SIMD hosts one wave of EACH channel half -- when the upper half of a tile is (partly)
  // padding (Cout = 98, 194, 322, 642, 770 ...) and its all-pad 16-column blocks are skipped (jv below), the saved MFMA
  // cycles are the SIMD's, not an idle half of the chip's.  (p.wave_map = 0: the old wave & 1 mapping, for A/B.)
  const int wm = WN == 2 ? (p.wave_map ? wave & 3 : wave >> 1) : wave, wn = WN == 2 ? (p.wave_map ? wave >> 2 : wave & 1) : 0;
  // 16-column blocks of this wave that hold at least one real output channel (block-uniform per wave)
  const int jv = p.wave_map ? __builtin_amdgcn_readfirstlane(min(NJ, max(0, (p.Ncols - tile_n * BN - wn * (BN / WN) + 15) >> 4))) : NJ;
  const int m16 = lane & 15, g = lane >> 4;
  // loader role: float4 `chunk` of halo pixels prow + 64*i.  A 16-lane ds_write_b64 group covers two pixels: with
  // 160-byte rows neighbours p, p+1 share 8 of the 32 store banks, p and p+2 none -> swap bits 0/1 of the pixel index
  const int chunk = tid & 7, prow_lin = tid >> 3;
  const int prow = (prow_lin & ~3) | ((prow_lin & 1) << 1) | ((prow_lin >> 1) & 1);

  // ---- halo loader (constant over the K loop) -----------------------------------------------------
  // The image is read through a buffer resource (base = this image, num_records = its extent): a lane whose
  // halo pixel is zero padding, or whose 4 channels lie beyond Cin, gets the out-of-range offset 2^31 and the
  // hardware returns zeros -- no select on the loaded data.  (conv3x3_halo16_usable: extent < 2^31 bytes.)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned OOB = 0x80000000u;
  struct Halo {
    i32x4 rsrc;
    unsigned off[A_IT];  // byte offset of (halo pixel prow + 64*it, channel chunk*4), or OOB
  };
  auto halo_of = [&](const Tile& c) {
    Halo hl;
    const unsigned long long img_base = (unsigned long long)(size_t)(p.x + (long long)c.n_img * p.x_bstride);
    hl.rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)img_base);
    hl.rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((img_base >> 32) & 0xffffu));  // stride 0: raw buffer
    hl.rsrc.z = __builtin_amdgcn_readfirstlane(
        (int)(unsigned)((((long long)p.H * p.W - 1) * p.ldx + ((p.Cin + 3) & ~3)) * 4));  // bytes up to the last channel read
    hl.rsrc.w = 0x00020000;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int hp = prow + 64 * it;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = c.y0 - 1 + hy, ix = c.x0 - 1 + hx;
      const bool ok = hp < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      hl.off[it] = ok ? (unsigned)(((iy * p.W + ix) * p.ldx + chunk * 4) * 4) : OOB;
    }
    return hl;
  };
  Halo hcur = halo_of(tl);
  Halo hnxt = hcur;  // PERSIST: the next tile of this workgroup (loaded from during the last slab)
  // UPS: per item the byte offset of tap (y0, x0) in the low-resolution image (or OOB), whether x1 / y1 are the next pixel /
  // row (align_corners clamps them at the far edge), and the two fractional weights -- ac_tap's values
  struct HaloU {
    i32x4 rsrc;
    unsigned off[A_IT];
    float wy[A_IT], wx[A_IT];
    unsigned edge;  // bit 2 it: x1 = x0 + 1, bit 2 it + 1: y1 = y0 + 1
  };
  HaloU hu;
  const int ups_slabs = UPS ? p.ups_c / BK : 0;
  unsigned u_dx = 0, u_dy = 0;
  if constexpr (UPS) {
    const unsigned long long ub = (unsigned long long)(size_t)(p.xu + (long long)tl.n_img * p.xu_bstride);
    hu.rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ub);
    hu.rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((ub >> 32) & 0xffffu));
    hu.rsrc.z = __builtin_amdgcn_readfirstlane((int)(unsigned)((((long long)p.uH * p.uW - 1) * p.ldxu + p.ups_c) * 4));
    hu.rsrc.w = 0x00020000;
    hu.edge = 0;
    u_dx = (unsigned)(p.ldxu * 4);
    u_dy = (unsigned)(p.uW * p.ldxu * 4);
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int hp = prow + 64 * it;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = tl.y0 - 1 + hy, ix = tl.x0 - 1 + hx;
      const bool ok = hp < HALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const AxisTap ty = ac_tap(ok ? iy : 0, p.usy, p.uH), tx = ac_tap(ok ? ix : 0, p.usx, p.uW);
      hu.off[it] = ok ? (unsigned)(((ty.i0 * p.uW + tx.i0) * p.ldxu + chunk * 4) * 4) : OOB;
      hu.wy[it] = ty.w1;
      hu.wx[it] = tx.w1;
      hu.edge |= (tx.i1 != tx.i0 ? 1u : 0u) << (2 * it);
      hu.edge |= (ty.i1 != ty.i0 ? 2u : 0u) << (2 * it);
    }
  }
  const long long w_row_stride = 9LL * p.Cin_pad;
  // With a tail tile (igemm.h: has_tail_tile) the last slab -- 2 real channels -- is not walked tap by tap: its
  // 9 taps x 2 channels are ONE extra step (k = 2*tap + c) after the full slabs.
  constexpr bool tail = TAIL;  // a template flag: the regular layers pay nothing for it
  const int cslabs = p.Cin_pad / BK;            // slabs staged through the halo buffers
  const int cchunks = cslabs - (tail ? 1 : 0);  // slabs walked by the 9-tap loop
  const int nsteps = 9 * cchunks + (tail ? 1 : 0);
  const int cin4 = (p.Cin + 3) & ~3;
  const int relu_floor = p.relu_in ? 0 : (int)0x80000000;  // fused input ReLU as an integer max on the float bits

  f32x4 ra[A_IT][LPI];
  auto a_voff = [&](const Halo& hl, int cc, int it) {  // (2^31 + cc*128 stays out of range: no wrap)
    return cc * BK + chunk * 4 < cin4 ? hl.off[it] + (unsigned)(cc * BK * 4) : OOB;
  };
  // Inline asm: (a) hipcc has no counted wait once LDS-DMAs are in flight -- it treats vmcnt as unordered and
  // waits vmcnt(0) in front of the first use; the value is handed back by the counted wait (its "+v" operand);
  // (b) the buffer form with hardware range checking.
  auto load_a_async = [&](const Halo& hl, int cc, int it) {
    if constexpr (!UPS) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ra[it][0]) : "v"(a_voff(hl, cc, it)), "s"(hl.rsrc) : "memory");
    } else {
      const bool ups = cc < ups_slabs;  // block-uniform
      i32x4 rs;
      rs.x = ups ? hu.rsrc.x : hl.rsrc.x;
      rs.y = ups ? hu.rsrc.y : hl.rsrc.y;
      rs.z = ups ? hu.rsrc.z : hl.rsrc.z;
      rs.w = hl.rsrc.w;
      // (an OOB base stays out of range with the tap offsets added: image extents are < 2^29 floats)
      const unsigned o0 = ups ? hu.off[it] + (unsigned)(cc * BK * 4) : a_voff(hl, cc, it);
      unsigned edge = hu.edge;
      asm volatile("" : "+v"(edge));  // opaque: the tap offsets are formed here, per load -- hoisted out of the slab loop they cost 18 registers
      const unsigned dx = ((edge >> (2 * it)) & 1u) ? u_dx : 0u, dy = ((edge >> (2 * it + 1)) & 1u) ? u_dy : 0u;
      const unsigned o1 = ups ? o0 + dx : OOB, o2 = ups ? o0 + dy : OOB, o3 = ups ? o0 + dx + dy : OOB;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ra[it][0]) : "v"(o0), "s"(rs) : "memory");
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ra[it][LPI - 3]) : "v"(o1), "s"(rs) : "memory");
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ra[it][LPI - 2]) : "v"(o2), "s"(rs) : "memory");
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ra[it][LPI - 1]) : "v"(o3), "s"(rs) : "memory");
    }
  };
  // the counted wait that hands item `it` back (its registers are the asm's in/out operands: nothing may be scheduled across)
#define PRV2_WAIT_ITEM(newer, it)                                                                                                     \
  do {                                                                                                                              \
    if constexpr (!UPS) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ra[it][0]) : "n"(newer) : "memory");                                \
    else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[it][0]), "+v"(ra[it][LPI - 3]), "+v"(ra[it][LPI - 2]), "+v"(ra[it][LPI - 1]) : "n"(newer) : "memory"); \
  } while (0)
  // cc_item: the slab the item was loaded for (UPS: decides whether its four taps are interpolated)
  auto store_a = [&](int abuf, int it, int cc_item) {
    const int hp = prow + 64 * it;
    if (hp >= HALO) return;
    f32x4 v_in = ra[it][0];
    if constexpr (UPS) {
      if (cc_item < ups_slabs) {  // block-uniform; upsample_bilinear_kernel's operation order (gather.hip), element by element
        float w1x = hu.wx[it], w1y = hu.wy[it];
        asm volatile("" : "+v"(w1x), "+v"(w1y));  // opaque: 1 - w is recomputed per item instead of living in 12 more registers
        const float w0x = 1.0f - w1x, w0y = 1.0f - w1y;
        f32x4 top = {0.f, 0.f, 0.f, 0.f}, bot = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          top[e] += w0x * ra[it][0][e];
          top[e] += w1x * ra[it][LPI - 3][e];
          bot[e] += w0x * ra[it][LPI - 2][e];
          bot[e] += w1x * ra[it][LPI - 1][e];
          r[e] += w0y * top[e];
          r[e] += w1y * bot[e];
        }
        v_in = r;
      }
    }
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    i32x4v vi = __builtin_bit_cast(i32x4v, v_in);
    vi.x = max(vi.x, relu_floor);  // ReLU on the bits: negative floats are negative ints; floor INT_MIN = identity
    vi.y = max(vi.y, relu_floor);
    vi.z = max(vi.z, relu_floor);
    vi.w = max(vi.w, relu_floor);
    const f32x4 v = __builtin_bit_cast(f32x4, vi);
    // hi = RNE bf16 of v (2 x v_cvt_pk), back to fp32 by shift / mask of the packed pairs, lo = RNE bf16 of v - hi
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
    const u32x2 hw = __builtin_bit_cast(u32x2, hi);
    f32x4 hf;
    hf.x = __builtin_bit_cast(float, hw.x << 16);
    hf.y = __builtin_bit_cast(float, hw.x & 0xffff0000u);
    hf.z = __builtin_bit_cast(float, hw.y << 16);
    hf.w = __builtin_bit_cast(float, hw.y & 0xffff0000u);
    const bf16x4 lo = __builtin_convertvector(v - hf, bf16x4);
    const unsigned addr = (unsigned)(size_t)(As_b + abuf * A_BYTES + hp * AROW) + chunk * 8;
    const unsigned long long h = __builtin_bit_cast(unsigned long long, hi), l = __builtin_bit_cast(unsigned long long, lo);
    // asm for the same reason: a compiler-visible ds_write waits for ALL in-flight LDS-DMAs first
    if constexpr (PREC == PRV2_PREC_BF16X3) asm volatile("ds_write2_b64 %0, %1, %2 offset1:8" ::"v"(addr), "v"(h), "v"(l) : "memory");
    else asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(h) : "memory");
  };
  // weight tile of step s: 64 lanes x 16 B = 8 rows x 128 B per DMA, linear in LDS (pre-swizzled in HBM)
  const int dma_row = lane >> 3, dma_slot = lane & 7;
  const float* wdma = reinterpret_cast<const float*>(p.w) + ((long long)tile_n * BN + dma_row) * w_row_stride + dma_slot * 4;
  const float* wdma_tail = reinterpret_cast<const float*>(p.w_tail) + ((long long)tile_n * BN + dma_row) * 32 + dma_slot * 4;
  auto dma_src = [&](int s, int i) {
    const int cc = s / 9, tap = s - cc * 9;
    const int piece = (wave * ND + i) % (BN / 8);  // BN = 32: waves 4-7 repeat pieces 0-3 (uniform vmcnt bookkeeping)
    const float* main = wdma + (long long)tap * p.Cin_pad + cc * BK + (long long)(piece * 8) * w_row_stride;
    return (tail && s == nsteps - 1) ? wdma_tail + piece * 8 * 32 : main;
  };
  auto dma_dst = [&](int bbuf, int i) { return Bs_b + bbuf * B_BYTES + ((wave * ND + i) % (BN / 8)) * 1024; };
  auto dma_b = [&](int s, int bbuf, int i) {  // prologue: compiler-visible
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)dma_src(s, i),
                                     (__attribute__((address_space(3))) void*)dma_dst(bbuf, i), 16, 0, 0);
  };
  // Main loop: the same instruction as inline asm.  While hipcc sees an LDS-DMA in flight it degrades every
  // s_waitcnt lgkmcnt(N) in front of an MFMA to lgkmcnt(0) (a flat-LDS access makes the counter "unordered"
  // in its model), which drains the fragment reads issued a few instructions earlier for the NEXT phase.
  auto dma_b_async = [&](int s, int bbuf, int i) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dma_dst(bbuf, i));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(dma_src(s, i)) : "memory");  // (m0 is not live anywhere else in the loop)
  };

  f32x4 acc[NA][NJ];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing ---------------------------------------------------------------------------
  // pixel run a = (image row a / 2 of the wave, half a % 2): halo pixel of lane m at tap (ky, kx)
  // halo pixel of lane m at tap (0,0) of run 0, and the pixel offset of (run a, tap ky,kx) from it
  const char* const a_lane = As_b + (TALL ? (wm * NA * 2 + (m16 >> 3)) * HW_ + (m16 & 7) : (NI * wm) * HW_ + m16) * AROW + g * 16;
  auto run_off = [](int a, int ky, int kx) { return TALL ? (2 * a + ky) * HW_ + kx : (a / 2 + ky) * HW_ + (a % 2) * 16 + kx; };
  const int b_key = (m16 >> 1) & 7;  // rows wn*(BN/WN) + 16 j + m: the key only depends on m
  const char* const b_lane_hi = Bs_b + (wn * (BN / WN) + m16) * 128 + ((g ^ b_key) << 4);
  const char* const b_lane_lo = Bs_b + (wn * (BN / WN) + m16) * 128 + (((4 + g) ^ b_key) << 4);
  bf16x8 ah[NA], al[NA], bh[2], bl[2];
  auto read_a = [&](int a, int abuf, int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const char* q = a_lane + abuf * A_BYTES + run_off(a, ky, kx) * AROW;
    ah[a] = *reinterpret_cast<const bf16x8*>(q);
    if constexpr (PREC == PRV2_PREC_BF16X3) al[a] = *reinterpret_cast<const bf16x8*>(q + 64);
  };
  auto read_b = [&](int slot, int bbuf, int j) {
    bh[slot] = *reinterpret_cast<const bf16x8*>(b_lane_hi + bbuf * B_BYTES + j * 16 * 128);
    if constexpr (PREC == PRV2_PREC_BF16X3) bl[slot] = *reinterpret_cast<const bf16x8*>(b_lane_lo + bbuf * B_BYTES + j * 16 * 128);
  };
  // product pr of the split (bf16x3: lo*hi, hi*lo, hi*hi -- smallest terms first; bf16: the single product)
  constexpr int NP = PREC == PRV2_PREC_BF16X3 ? 3 : 1;
  auto mma = [&](int a, int j, int slot, int pr) {
    if constexpr (PREC == PRV2_PREC_BF16X3) {
      if (pr == 0) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[a], bh[slot], acc[a][j], 0, 0, 0);
      if (pr == 1) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bl[slot], acc[a][j], 0, 0, 0);
      if (pr == 2) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh[slot], acc[a][j], 0, 0, 0);
    } else {
      acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh[slot], acc[a][j], 0, 0, 0);
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------
#pragma unroll
  for (int it = 0; it < A_IT; ++it) load_a_async(hcur, 0, it);
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    dma_b(0, 0, i);
    dma_b(1, 1, i);
    dma_b(2, 2, i);
  }
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    PRV2_WAIT_ITEM(0, it);
    store_a(0, it, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // full fence: lgkmcnt for the asm stores, vmcnt(0) for the DMAs
#pragma unroll
  for (int a = 0; a < NA; ++a) read_a(a, 0, 0);
  read_b(0, 0, 0);

  int gs = 0;  // slabs walked so far by this workgroup: slab gs sits in halo buffer gs & 1
  for (int t = bid;;) {
  const int t_next = t + nwg;
  const bool has_next = PERSIST && t_next < ntiles;  // block-uniform
  Tile tl_next = tl;
  if (has_next) {
    tl_next = decode(t_next);
    hnxt = halo_of(tl_next);
  }
  for (int cc = 0; cc < cchunks; ++cc) {
    // the slab staged during this one: the next slab of the tile; behind the last one the first slab of this workgroup's
    // next tile (PERSIST) -- or, when there is none, clamped data nobody reads (no branches around the loads)
    const bool wrap = PERSIST && has_next && cc + 1 == cslabs;
    const int ccn = cc + 1 < cslabs ? cc + 1 : (wrap ? 0 : cc);
    const Halo& hl = wrap ? hnxt : hcur;
    const int ab = SINGLE ? 0 : (gs + cc) & 1;  // halo buffer of this slab
    const int ab_next = SINGLE ? 0 : ab ^ 1;
    auto step = [&](auto tap_c) {
      constexpr int tap = decltype(tap_c)::value;
      constexpr int L0 = tap < A_IT ? 1 : 0, Lm1 = (tap >= 1 && tap - 1 < A_IT) ? 1 : 0;
      const int s = cc * 9 + tap;
      const int s3 = s + 3 < nsteps ? s + 3 : (PERSIST ? s + 3 - nsteps : nsteps - 1);  // PERSIST: wraps into the next tile
      constexpr int bb = tap % 3;  // 9 taps per slab: step mod 3 == tap mod 3
#pragma unroll
      for (int j = 0; j < NJ - 1; ++j) {
        read_b((j + 1) & 1, bb, j + 1);
        __builtin_amdgcn_sched_barrier(0);
        // two pixel runs at a time, products outer: consecutive MFMAs never share an accumulator, and the runs
        // refilled last (end of the previous step) are needed last
#pragma unroll
        for (int a = 0; a < NA; a += 2) {
          if (j < jv) {
#pragma unroll
          for (int pr = 0; pr < NP; ++pr) {
            mma(a, j, j & 1, pr);
            mma(a + 1, j, j & 1, pr);
          }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (j == 0 && a == 0) {
#ifndef PRV2_ABL_NOA
            if constexpr (tap < A_IT) load_a_async(hl, ccn, tap);
#endif
          }
          if (j == (NJ > 2 ? 1 : 0) && a == NA - 2) {
#ifndef PRV2_ABL_NOA
            if constexpr (!SINGLE && tap >= 2 && tap - 2 < A_IT) {
              // VMEM instructions issued since load_a_async(tap - 2): DMAs of steps tap-2 and tap-1, loads tap-1, tap
              constexpr int newer = 2 * ND + LPI * (Lm1 + L0);
              PRV2_WAIT_ITEM(newer, tap - 2);
              store_a(ab ^ 1, tap - 2, ccn);  // other halo buffer: last read in the previous slab
            }
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // Barrier: every wave has issued (and, lgkmcnt(0), received) its reads of this step's weight tile and, at
      // tap 8, of this slab's halo; the DMA of step s+1 (issued two barriers ago) has landed: newer than it are
      // the halo loads of steps s-1 and s and the DMA issued at the previous barrier.
#ifdef PRV2_ABL_NOBAR
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ND + LPI * (Lm1 + L0)) : "memory");
#else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(ND + LPI * (Lm1 + L0)) : "memory");
#endif
      read_b(NJ & 1, (tap + 1) % 3, 0);  // column 0 of the next step (the last column sits in slot (NJ-1)&1)
      __builtin_amdgcn_sched_barrier(0);
      // last column, two pixel runs at a time; as a pair retires its registers take the NEXT tap's runs
#pragma unroll
      for (int a = 0; a < NA; a += 2) {
        if (NJ - 1 < jv) {
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
          mma(a, NJ - 1, (NJ - 1) & 1, pr);
          mma(a + 1, NJ - 1, (NJ - 1) & 1, pr);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SINGLE && tap == 8) {
          // behind this tap's barrier nobody reads the slab any more: the next one (in registers since taps 0-5) takes its place
          if (a == 0 && cc + 1 < cslabs) {  // (first pass only; block-uniform)
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
              // VMEM instructions issued since the last halo load (tap A_IT - 1): the DMAs of taps A_IT - 1 .. 7
              PRV2_WAIT_ITEM((8 - (A_IT - 1)) * ND, it);
              store_a(0, it, ccn);
            }
          }
        } else {
          read_a(a, tap == 8 ? ab_next : ab, (tap + 1) % 9);
          read_a(a + 1, tap == 8 ? ab_next : ab, (tap + 1) % 9);
        }
#ifndef PRV2_ABL_NOB
        if (a / 2 < ND) dma_b_async(s3, tap % 3, a / 2);
#endif  // this step's tile buffer is free since the barrier
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SINGLE && tap == 8) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the new slab is in place
#pragma unroll
        for (int a = 0; a < NA; ++a) read_a(a, 0, 0);
      }
      if constexpr ((NJ & 1) != 0) {  // (not instantiated: NJ is 2 or 4, so next step's column 0 sits in slot 0)
        bh[0] = bh[1];
        bl[0] = bl[1];
      }
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
  }
  if (tail) {
    // ---- tail step: A operand gathered from the staged last slab, k = 2*tap + c: lane group g covers taps
    // 4g..4g+3 (channels 0,1 of the slab = the first 4 bytes of the pixel's hi / lo plane).  The weight tile sits
    // in buffer 0 (step index 9*cchunks), column 0 already in slot 0; the fragments refilled by the last
    // regular step are replaced.
    const int abuf = SINGLE ? 0 : cchunks & 1;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      u32x4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tp = 4 * g + i;
        const int tpc = tp < 9 ? tp : 0;
        const int ky = tpc / 3, kx = tpc - ky * 3;
        const char* q = a_lane - g * 16 + abuf * A_BYTES + run_off(a, ky, kx) * AROW;
        const unsigned hv = *reinterpret_cast<const unsigned*>(q), lv = *reinterpret_cast<const unsigned*>(q + 64);
        h[i] = tp < 9 ? hv : 0u;
        l[i] = tp < 9 ? lv : 0u;
      }
      ah[a] = __builtin_bit_cast(bf16x8, h);
      al[a] = __builtin_bit_cast(bf16x8, l);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) read_b((j + 1) & 1, 0, j + 1);
      if (j < jv) {
#pragma unroll
      for (int pr = 0; pr < NP; ++pr)
#pragma unroll
        for (int a = 0; a < NA; ++a) mma(a, j, j & 1, pr);
      }
    }
  }
  if constexpr (!PERSIST) {
  // The clamped DMAs of the last two steps are still in flight and hipcc does not know it (they are inline asm, so
  // __syncthreads() alone emits NO vmcnt wait): drain them by hand before the C tile overwrites the buffers.  Without
  // this a late DMA (HBM contention from kernels on other streams) lands on top of the C tile.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // (PERSIST: the C tile has LDS of its own, the DMAs in flight belong to the next tile's first steps -- no drain.  Its
  // previous contents were last read in the store loop of the previous tile, >= 9 barriers ago.)
  auto epilogue = [&](const Tile& c) {
    // ---- epilogue through LDS (see igemm.hip) ------------------------------------------------------
  #pragma unroll
    for (int a = 0; a < NA; ++a)
  #pragma unroll
      for (int j = 0; j < NJ; ++j)
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = TALL ? ((wm * NA + a) * 2 + ((4 * g + e) >> 3)) * TW + ((4 * g + e) & 7)
                               : (NI * wm + a / 2) * TW + (a % 2) * 16 + 4 * g + e;
          csm[row * CLD + wn * (BN / WN) + j * 16 + m16] = acc[a][j][e];
        }
    __syncthreads();
    if (p.pre) {  // block-uniform: the conv's coarse half (prv2_conv2d_pre) joins the C tile in front of bias / LayerNorm / activation
      constexpr int PC4 = BN / 4, PRPP = 512 / PC4, PNR = TH * TW / PRPP;
      const int pc = c.tile_n * BN + (tid % PC4) * 4;
      if (pc < p.Cout) {  // (host: Cout % 4 == 0)
        const float* const pbase = p.pre + (long long)c.n_img * p.H * p.W * p.ld_pre + pc;
        // four rows of the thread requested at a time (the narrow kernels live within 128 registers; rows outside the image: clamped
        // address, added to rows nobody stores)
        constexpr int CH = PNR < 4 ? PNR : 4;
#pragma unroll 1
        for (int i0 = 0; i0 < PNR; i0 += CH) {
          f32x4 pv[CH];
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            const int rr = tid / PC4 + (i0 + i) * PRPP, py = rr / TW, px = rr - py * TW;
            pv[i] = *reinterpret_cast<const f32x4*>(pbase + (long long)(min(c.y0 + py, p.H - 1) * p.W + min(c.x0 + px, p.W - 1)) * p.ld_pre);
          }
#pragma unroll
          for (int i = 0; i < CH; ++i) {
            f32x4* q = reinterpret_cast<f32x4*>(&csm[(tid / PC4 + (i0 + i) * PRPP) * CLD + (tid % PC4) * 4]);
            *q = *q + pv[i];
          }
        }
      }
      __syncthreads();
    }
    float* const ln_stats = csm + TH * TW * CLD;
    if constexpr (GATE) {
      // ---- gate stage (host: p.ln_w != null, Cout == BN, act none / ReLU, 16-byte rows) ----------------------------------
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      constexpr int ROWS = TH * TW, KS2 = BN / 32;
      // wave -> (pixel runs, 16-column blocks) of the gate GEMM: BN = 128: every run x block `wave`; BN = 32: runs 2w, 2w + 1 x both
      constexpr int G_RUNS = BN == 128 ? 16 : 2, G_COLS = BN == 128 ? 1 : 2;
      const int run0 = BN == 128 ? 0 : 2 * wave, cb0 = BN == 128 ? wave : 0;
      // the wave's weight fragments (BN = 128: 32 registers), requested before the statistics pass
      u32x4 wf[KS2][G_COLS][2];
      {
        const u32x4* const gw = reinterpret_cast<const u32x4*>(p.gate_w) + lane;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
          for (int j = 0; j < G_COLS; ++j) {
            wf[ks][j][0] = gw[gate_frag_index(BN, cb0 + j, ks, 0)];
            if constexpr (PREC == PRV2_PREC_BF16X3) wf[ks][j][1] = gw[gate_frag_index(BN, cb0 + j, ks, 1)];
          }
      }
      ln_row_stats(p, csm, CLD, ROWS, tid, ln_stats);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (bare barriers: __syncthreads() also drains vmcnt)
      {  // normalise + activate + split in place: the 32 bytes of 8 fp32 channels become [8 bf16 hi | 8 bf16 lo] = one A fragment
        const int r = tid & (ROWS - 1);
        const float mean = ln_stats[r], rstd = ln_stats[ROWS + r];
        const float act_floor = p.act == PRV2_ACT_RELU ? 0.f : -__builtin_inff();
#pragma unroll
        for (int i = 0; i < BN / 16; ++i) {
          const int c8 = (tid >> 8) + 2 * i;
          float* q = csm + r * CLD + c8 * 8;
          f32x4 v0 = *reinterpret_cast<const f32x4*>(q), v1 = *reinterpret_cast<const f32x4*>(q + 4);
          const float* bp = p.bias ? p.bias + c8 * 8 : nullptr;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = fmaxf((v0[e] + (bp ? bp[e] : 0.f) - mean) * rstd * p.ln_w[c8 * 8 + e] + p.ln_b[c8 * 8 + e], act_floor);
            v1[e] = fmaxf((v1[e] + (bp ? bp[4 + e] : 0.f) - mean) * rstd * p.ln_w[c8 * 8 + 4 + e] + p.ln_b[c8 * 8 + 4 + e], act_floor);
          }
          bf16x4 h0, l0, h1, l1;
          split_bf16(v0, h0, l0);
          split_bf16(v1, h1, l1);
          *reinterpret_cast<bf16x8*>(q) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          *reinterpret_cast<bf16x8*>(q + 4) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      f32x4 acc2[G_RUNS][G_COLS];
#pragma unroll
      for (int a = 0; a < G_RUNS; ++a)
#pragma unroll
        for (int j = 0; j < G_COLS; ++j) acc2[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
        for (int a0 = 0; a0 < G_RUNS; a0 += (G_RUNS < 4 ? G_RUNS : 4)) {  // up to four pixel runs at a time
          constexpr int NR_ = G_RUNS < 4 ? G_RUNS : 4;
          bf16x8 xh[NR_], xl[NR_];
#pragma unroll
          for (int a = 0; a < NR_; ++a) {
            const float* q = csm + ((run0 + a0 + a) * 16 + m16) * CLD + ks * 32 + 8 * g;
            xh[a] = *reinterpret_cast<const bf16x8*>(q);
            xl[a] = *reinterpret_cast<const bf16x8*>(q + 4);
          }
#pragma unroll
          for (int j = 0; j < G_COLS; ++j) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wf[ks][j][0]);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, wf[ks][j][1]);
#pragma unroll
            for (int a = 0; a < NR_; ++a) {  // same product order as the stand-alone GEMM: lo*hi, hi*lo, hi*hi per slab
              if constexpr (PREC == PRV2_PREC_BF16X3) {
                acc2[a0 + a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[a], wh, acc2[a0 + a][j], 0, 0, 0);
                acc2[a0 + a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[a], wl, acc2[a0 + a][j], 0, 0, 0);
              }
              acc2[a0 + a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[a], wh, acc2[a0 + a][j], 0, 0, 0);
            }
          }
        }
      // final stage: y = mul * sigmoid(gate + gate_bias) + res; all mul / res rows of the thread requested up front
      constexpr int C4g = BN / 4, RPPg = 512 / C4g, NRg = ROWS / RPPg;
      const int col4g = tid % C4g;
      auto pix_of = [&](int i) {  // (pixels outside the image: clamped address, never stored)
        const int rr = tid / C4g + i * RPPg;
        const int py = rr / TW, px = rr - py * TW;
        return min(c.y0 + py, p.H - 1) * p.W + min(c.x0 + px, p.W - 1);
      };
      const long long img_mg = (long long)c.n_img * p.H * p.W;
      const unsigned img_px = (unsigned)(p.H * p.W - 1);
      const __amdgpu_buffer_rsrc_t mul_rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.mul ? p.mul + img_mg * p.ld_mul : p.x), 0, p.mul ? (int)((img_px * p.ld_mul + BN) * 4u) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.res ? p.res + img_mg * p.ld_res : p.x), 0, p.res ? (int)((img_px * p.ld_res + BN) * 4u) : 0, 0x00020000);
      const bool has_mul = p.mul != nullptr;  // block-uniform
      f32x4 mv[NRg], rv[NRg];
#pragma unroll
      for (int i = 0; i < NRg; ++i) {
        const int pix = pix_of(i);
        mv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mul_rs, (pix * p.ld_mul + col4g * 4) * 4, 0, 0));
        rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, (pix * p.ld_res + col4g * 4) * 4, 0, 0));
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read its rows of the C tile
#pragma unroll
      for (int a = 0; a < G_RUNS; ++a)
#pragma unroll
        for (int j = 0; j < G_COLS; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) csm[((run0 + a) * 16 + 4 * g + e) * CLD + (cb0 + j) * 16 + m16] = acc2[a][j][e];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      f32x4 gb = {0.f, 0.f, 0.f, 0.f};
      if (p.gate_bias) gb = *reinterpret_cast<const f32x4*>(p.gate_bias + col4g * 4);
      float* const ybase = p.y + (long long)c.n_img * p.y_bstride + col4g * 4;
#pragma unroll
      for (int i = 0; i < NRg; ++i) {
        const int rr = tid / C4g + i * RPPg;
        const int py = rr / TW, px = rr - py * TW;
        const f32x4 cv = *reinterpret_cast<const f32x4*>(&csm[rr * CLD + col4g * 4]);
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          ov[e] = (has_mul ? mv[i][e] : 1.0f) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((cv[e] + gb[e]) * -1.44269504088896340736f)) + rv[i][e];
        if (c.y0 + py < p.H && c.x0 + px < p.W) {
          float* dst = ybase + (long long)pix_of(i) * p.ldy;
          asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
        }
      }
      return;
    }
    if (p.ln_w) {  // block-uniform
      ln_row_stats(p, csm, CLD, TH * TW, tid, ln_stats);
      __syncthreads();
    }

    constexpr int C4 = BN / 4;
    constexpr int RPP = 512 / C4;
    // Store-loop roles: thread = (channel quad col4, rows rr0 + k RPP).  For BN = 128 a wave is two whole rows (any order is conflict free).
    // For BN = 32 / 64 the natural tid % C4 / tid / C4 made every float4 read of the C tile 2-way: ds_read_b128 serves the lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS; tools/lds_bank_model.py) -- so each group is given whole
    // rows: BN = 64 one row (16 quads = 64 banks), BN = 32 rows r and r + 8 (pitch 36: the other 32 banks).  SQ_LDS_BANK_CONFLICT of the
    // narrow kernels was 10-28 % of their LDS cycles (profiles/r04_bf16x3_pmc_lds_frame.txt).
    int col4 = tid % C4, rr0 = tid / C4;
    if constexpr (BN == 32 || BN == 64) {
      const int seg4 = (lane >> 2) & 7, k = (0x96 >> seg4) & 1;  // lane group of the quad-lane: [0, 1, 1, 0, 1, 0, 0, 1]
      if constexpr (BN == 32) {
        col4 = ((seg4 & 2) << 1) + (lane & 3);
        rr0 = 16 * (wave >> 1) + 4 * (wave & 1) + 2 * (lane >> 5) + k + 8 * (seg4 >> 2);
      } else {
        col4 = 4 * (seg4 >> 1) + (lane & 3);
        rr0 = 4 * wave + 2 * (lane >> 5) + k;
      }
    }
    EpiCols ec;
    if (!epi_cols(p, c.tile_n * BN + col4 * 4, ec)) return;
    const long long img_m = (long long)c.n_img * p.H * p.W, img_o = (long long)c.n_img * p.y_bstride + ec.co;
    // The store loop is VALU-bound (16 rows per thread, 2 waves per SIMD): the general epilogue costs ~100 instructions a
    // row -- every optional stage as a select -- which was 14 k of the 280 k cycles of a 512->256 tile.
    // Lean paths for the three common shapes: bias + act; bias + LN + act (fusion encoders); bias + act + one residual
    // (GatedConvUnit.conv).  Everything else (gates, gamma, two residuals, ragged channels) takes the general loop below.
    const bool simple = ec.vec && !p.gamma && !p.mul && !p.res2 && !(p.ln_w && p.res);  // block-uniform
    if (simple) {
      float* const ybase = p.y + img_o;
      const float* const rbase = p.res ? p.res + img_m * p.ld_res + ec.co : nullptr;
      auto lean = [&](auto act_c, auto ln_c, auto res_c) {
        constexpr bool LN = decltype(ln_c)::value, RES = decltype(res_c)::value;
        for (int rr = rr0; rr < TH * TW; rr += RPP) {
          const int py = rr / TW, px = rr - py * TW;
          const int oy = c.y0 + py, ox = c.x0 + px;
          if (oy >= p.H || ox >= p.W) continue;
          const f32x4 cv = *reinterpret_cast<const f32x4*>(&csm[rr * CLD + col4 * 4]);
          const int pix = oy * p.W + ox;
          f32x4 rv = {0.f, 0.f, 0.f, 0.f};
          if constexpr (RES) rv = *reinterpret_cast<const f32x4*>(rbase + (unsigned)(pix * p.ld_res));
          f32x4 ov;
  #pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = cv[e] + ec.bias[e];
            if constexpr (LN) t = (t - ln_stats[rr]) * ln_stats[TH * TW + rr] * ec.lnw[e] + ec.lnb[e];
            t = act_apply_bf(t, decltype(act_c)::value);
            if constexpr (RES) t += rv[e];
            ov[e] = e < ec.nvalid ? t : 0.f;  // pad channels behind cout stay zero
          }
          float* dst = ybase + (unsigned)(pix * p.ldy);
          asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
        }
      };
      if (p.tail1 && col4 == 0 && c.tile_n == p.tiles_n - 1) {
        // [pred1 | pred2 | 0 | 0] behind the Cout channels of this thread's rows: depth_pair_fill_kernel's arithmetic (gather.hip),
        // one 16-byte store per pixel that completes the row the loop below writes (fusion_model.py:91-118)
        float* const tbase = p.y + (long long)c.n_img * p.y_bstride + p.Cout;
        const float* const t1 = p.tail1 + (long long)c.n_img * p.tH * p.tW;
        const float* const t2 = p.tail2 + (long long)c.n_img * p.tH * p.tW;
        for (int rr = rr0; rr < TH * TW; rr += RPP) {
          const int py = rr / TW, px = rr - py * TW;
          const int oy = c.y0 + py, ox = c.x0 + px;
          if (oy >= p.H || ox >= p.W) continue;
          const AxisTap ty = ac_tap(oy, p.tsy, p.tH), tx = ac_tap(ox, p.tsx, p.tW);
          auto interp = [&](const float* q) {
            const float v00 = q[ty.i0 * p.tW + tx.i0], v01 = q[ty.i0 * p.tW + tx.i1];
            const float v10 = q[ty.i1 * p.tW + tx.i0], v11 = q[ty.i1 * p.tW + tx.i1];
            float top = 0.f, bot = 0.f, r = 0.f;
            top += tx.w0 * v00;
            top += tx.w1 * v01;
            bot += tx.w0 * v10;
            bot += tx.w1 * v11;
            r += ty.w0 * top;
            r += ty.w1 * bot;
            return r;
          };
          const f32x4 ov = {interp(t1), interp(t2), 0.f, 0.f};
          float* dst = tbase + (unsigned)((oy * p.W + ox) * p.ldy);
          asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
        }
      }
      using T_ = std::true_type;
      using F_ = std::false_type;
      dispatch_act(p.act, [&](auto act_c) {
        if (p.ln_w) lean(act_c, T_{}, F_{});
        else if (p.res) lean(act_c, F_{}, T_{});
        else lean(act_c, F_{}, F_{});
      });
      return;
    }
    dispatch_act(p.act, [&](auto act_c) {
      for (int rr = rr0; rr < TH * TW; rr += RPP) {
        const int py = rr / TW, px = rr - py * TW;
        const int oy = c.y0 + py, ox = c.x0 + px;
        if (oy >= p.H || ox >= p.W) continue;
        const f32x4 cv = *reinterpret_cast<const f32x4*>(&csm[rr * CLD + col4 * 4]);
        const int pix = oy * p.W + ox;
        epi_store<decltype(act_c)::value>(p, ec, cv, img_m + pix, img_o + (long long)pix * p.ldy, ln_stats[rr],
                                          ln_stats[TH * TW + rr]);
      }
    });
  };
#ifndef PRV2_ABL_NOEPI
  epilogue(tl);
#endif
  if (!has_next) break;
  // next tile of this workgroup: its first slab is staged, its first fragments and weight column are in registers
  t = t_next;
  tl = tl_next;
  hcur = hnxt;
  gs += cchunks;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }  // tiles of this workgroup
  if constexpr (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land in LDS after the workgroup is gone
}

// One launch = the 32-pixel tile columns [0, tiles_x) as 8 x 32 tiles + (strip_blocks > 0) the remainder strip
// [rx0, W) as 32 x 8 tiles.  The strip workgroups come first in the grid: they are ordinary workgroups of the same
// duration, so the strip costs its share of workgroups (3-5 %) instead of an under-filled launch of its own.
template <int BN, int PREC, bool TAIL>
__global__ void __launch_bounds__(512, 2) conv3x3_halo16_kernel(const IgemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[halo16_smem_floats<BN>()];
  const int strip = p.strip_blocks;  // block-uniform
  if ((int)blockIdx.x < strip) halo16_body<BN, PREC, TAIL, true>(p, smem, blockIdx.x, strip);
  else halo16_body<BN, PREC, TAIL, false>(p, smem, blockIdx.x - strip, gridDim.x - strip);
}

// the same with the fused-upsample loader (p.xu: the first p.ups_c input channels are bilinear samples of a low-resolution tensor)
template <int BN, int PREC, bool TAIL>
__global__ void __launch_bounds__(512, 2) conv3x3_halo16_ups_kernel(const IgemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[halo16_smem_floats<BN>()];
  const int strip = p.strip_blocks;  // block-uniform
  if ((int)blockIdx.x < strip) halo16_body<BN, PREC, TAIL, true, false, false, false, true>(p, smem, blockIdx.x, strip);
  else halo16_body<BN, PREC, TAIL, false, false, false, false, true>(p, smem, blockIdx.x - strip, gridDim.x - strip);
}

bool conv3x3_halo16_ups_usable(const IgemmParams& p, int prec) {
  return p.xu && conv3x3_halo16_usable(p, prec) && p.Ncols > 64 && p.ups_c > 0 && p.ups_c % 32 == 0 && p.ups_c <= p.Cin && !p.relu_in && p.ldxu % 4 == 0 &&
         p.ldxu >= p.ups_c && (reinterpret_cast<uintptr_t>(p.xu) & 15) == 0 && p.xu_bstride % 4 == 0 && (long long)p.uH * p.uW * p.ldxu < (1LL << 29) &&
         p.uH >= 1 && p.uW >= 1;
}

// BN = 32 without tail tile: the 8 x 32 tiles are walked by persistent workgroups (one per CU: 160 KB of LDS), the strip
// tiles stay ordinary leading workgroups.
template <int PREC>
__global__ void __launch_bounds__(512, 2) conv3x3_halo16_persist_kernel(const IgemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[halo16_smem_floats<32, true>()];
  const int strip = p.strip_blocks;  // block-uniform
  if ((int)blockIdx.x < strip) halo16_body<32, PREC, false, true>(p, smem, blockIdx.x, strip);
  else halo16_body<32, PREC, false, false, true>(p, smem, blockIdx.x - strip, gridDim.x - strip);
}

// BN = 32 / 64, single halo buffer: two workgroups per CU (66 / 79 KB of LDS, <= 128 registers)
template <int BN, int PREC, bool TAIL>
__global__ void __launch_bounds__(512, 4) conv3x3_halo16_narrow_kernel(const IgemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[halo16_smem_floats<BN, false, true>()];
  const int strip = p.strip_blocks;  // block-uniform
  if ((int)blockIdx.x < strip) halo16_body<BN, PREC, TAIL, true, false, true>(p, smem, blockIdx.x, strip);
  else halo16_body<BN, PREC, TAIL, false, false, true>(p, smem, blockIdx.x - strip, gridDim.x - strip);
}

// GatedConvUnit tail at 128 / 32 channels (one workgroup per CU / the two-workgroup narrow scheme)
template <int BN, int PREC>
__global__ void __launch_bounds__(512, BN == 32 ? 4 : 2) conv3x3_halo16_gate_kernel(const IgemmParams p) {
  constexpr bool SGL = BN == 32;
  __shared__ __attribute__((aligned(16))) float smem[halo16_smem_floats<BN, false, SGL>()];
  const int strip = p.strip_blocks;  // block-uniform
  if ((int)blockIdx.x < strip) halo16_body<BN, PREC, false, true, false, SGL, true>(p, smem, blockIdx.x, strip);
  else halo16_body<BN, PREC, false, false, false, SGL, true>(p, smem, blockIdx.x - strip, gridDim.x - strip);
}

bool conv3x3_halo16_gate_usable(const IgemmParams& p, int prec) {
  return conv3x3_halo16_usable(p, prec) && (p.Cout == 32 || p.Cout == 128) && p.Ncols == p.Cout && p.Cin % 32 == 0 && p.ln_w && p.vec_epi &&
         !p.gamma && !p.res2 && (p.act == PRV2_ACT_NONE || p.act == PRV2_ACT_RELU) && (long long)p.H * p.W * (p.ld_mul > p.ld_res ? p.ld_mul : p.ld_res) < (1LL << 29);
}

void launch_conv3x3_halo16_gate(IgemmParams& p, int prec, hipStream_t s) {
  p.wave_map = 1;
  p.tiles_n = 1;
  p.strip_blocks = p.rw > 0 ? p.N * (int)cdiv(p.H, 32) : 0;
  const int blocks = p.N * (int)cdiv(p.H, 8) * p.tiles_x + p.strip_blocks;
  set_kernel("conv3x3_halo16_gate_kernel", p.Cout, prec);
  if (p.Cout == 128) {
    if (prec == PRV2_PREC_BF16X3) hipLaunchKernelGGL((conv3x3_halo16_gate_kernel<128, PRV2_PREC_BF16X3>), dim3(blocks), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_halo16_gate_kernel<128, PRV2_PREC_BF16>), dim3(blocks), dim3(512), 0, s, p);
  } else {
    if (prec == PRV2_PREC_BF16X3) hipLaunchKernelGGL((conv3x3_halo16_gate_kernel<32, PRV2_PREC_BF16X3>), dim3(blocks), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_halo16_gate_kernel<32, PRV2_PREC_BF16>), dim3(blocks), dim3(512), 0, s, p);
  }
}

static int persist_workgroups() {  // one persistent workgroup per CU of the CURRENT device (cached per device ordinal)
  static int n[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (n[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    n[dev] = cus;
  }
  return n[dev];
}


void launch_conv3x3_halo16(IgemmParams& p, int prec, hipStream_t s) {
  using namespace m16;
  static const int wave_map = getenv("PRV2_HALO_WAVE_MAP") ? atoi(getenv("PRV2_HALO_WAVE_MAP")) : 1;  // A/B switch
  p.wave_map = wave_map;
  p.tiles_n = p.Ncols > 64 ? (int)cdiv(p.Ncols, 128) : 1;
  p.strip_blocks = p.rw > 0 ? p.N * (int)cdiv(p.H, 32) * p.tiles_n : 0;
  const int blocks = p.N * (int)cdiv(p.H, 8) * p.tiles_x * p.tiles_n + p.strip_blocks;
  if (p.xu) {  // fused-upsample loader: the 128-column kernel (conv2d_impl checked conv3x3_halo16_ups_usable)
    set_kernel("conv3x3_halo16_ups_kernel", 128, prec);
#define PRV2_LAUNCH_UPS(PREC_)                                                                                             \
  do {                                                                                                                   \
    if (p.w_tail) hipLaunchKernelGGL((conv3x3_halo16_ups_kernel<128, PREC_, true>), dim3(blocks), dim3(512), 0, s, p);   \
    else hipLaunchKernelGGL((conv3x3_halo16_ups_kernel<128, PREC_, false>), dim3(blocks), dim3(512), 0, s, p);           \
  } while (0)
    if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_UPS(PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_UPS(PRV2_PREC_BF16);
#undef PRV2_LAUNCH_UPS
    return;
  }
  static const bool no_persist = getenv("PRV2_HALO_NO_PERSIST") != nullptr;  // A/B switches
  static const int n32_mode = getenv("PRV2_HALO_N32") ? atoi(getenv("PRV2_HALO_N32")) : 1;  // 1: two single-halo workgroups per CU; 0: round-1 paths
  set_kernel("conv3x3_halo16_kernel", p.Ncols > 64 ? 128 : (p.Ncols > 32 ? 64 : 32), prec);
#define PRV2_LAUNCH_NARROW(BN_, PREC_)                                                                                       \
  do {                                                                                                                       \
    if (p.w_tail) hipLaunchKernelGGL((conv3x3_halo16_narrow_kernel<BN_, PREC_, true>), dim3(blocks), dim3(512), 0, s, p);    \
    else hipLaunchKernelGGL((conv3x3_halo16_narrow_kernel<BN_, PREC_, false>), dim3(blocks), dim3(512), 0, s, p);            \
  } while (0)
  if (p.Ncols <= 32 && n32_mode == 1) {
    if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_NARROW(32, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_NARROW(32, PRV2_PREC_BF16);
    return;
  }
  static const int n64_mode = getenv("PRV2_HALO_N64") ? atoi(getenv("PRV2_HALO_N64")) : 1;
  if (p.Ncols > 32 && p.Ncols <= 64 && n64_mode == 1) {
    if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_NARROW(64, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_NARROW(64, PRV2_PREC_BF16);
    return;
  }
  // (wider layers as N-tiles of 64 on this kernel -- two workgroups per CU, the halo staged once per N-tile -- are no faster than one
  //  workgroup with 128 columns: 98->98 272 vs 283 TF, 194->194 336 vs 335, 322->322 376 vs 397)
#undef PRV2_LAUNCH_NARROW
  if (p.Ncols <= 32 && !p.w_tail && !no_persist) {
    const int tiles = blocks - p.strip_blocks, wgs = tiles < persist_workgroups() ? tiles : persist_workgroups();
    if (prec == PRV2_PREC_BF16X3) hipLaunchKernelGGL((conv3x3_halo16_persist_kernel<PRV2_PREC_BF16X3>), dim3(wgs + p.strip_blocks), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_halo16_persist_kernel<PRV2_PREC_BF16>), dim3(wgs + p.strip_blocks), dim3(512), 0, s, p);
    return;
  }
#define PRV2_LAUNCH_HALO16(BN_, PREC_)                                                                                 \
  do {                                                                                                                 \
    if (p.w_tail) hipLaunchKernelGGL((conv3x3_halo16_kernel<BN_, PREC_, true>), dim3(blocks), dim3(512), 0, s, p);     \
    else hipLaunchKernelGGL((conv3x3_halo16_kernel<BN_, PREC_, false>), dim3(blocks), dim3(512), 0, s, p);             \
  } while (0)
  if (p.Ncols > 64) {
    if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_HALO16(128, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_HALO16(128, PRV2_PREC_BF16);
  } else if (p.Ncols > 32) {
    if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_HALO16(64, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_HALO16(64, PRV2_PREC_BF16);
  } else {
    if (prec == PRV2_PREC_BF16X3) PRV2_LAUNCH_HALO16(32, PRV2_PREC_BF16X3);
    else PRV2_LAUNCH_HALO16(32, PRV2_PREC_BF16);
  }
#undef PRV2_LAUNCH_HALO16
}

}  // namespace prv2
