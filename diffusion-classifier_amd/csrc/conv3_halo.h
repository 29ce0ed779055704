// conv3_halo.h — geometry, LDS layout and epilogue functors shared by the halo-tile 3x3 convolution kernels
// (conv3_halo.hip: every wave loads, transforms and multiplies; conv3_ws.hip: wave-specialised — loader / transform waves + MFMA waves).
#pragma once
#include <stdlib.h>
#include "igemm_epilogue.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct HaloGeom {
  int ltw, lth, lni;        // log2 of tile width / height / images per tile (tw*th*ni == NW*64)
  int tiles_x, tiles_y;     // tiles per image
  int hw, hp, HR, nxl;      // halo width, halo pixels per image patch, halo rows per tile, X loads per lane
  int n_img, H, W;
  float inv_hp, inv_hw;     // 1/hp, 1/hw: the loaders' piece -> (image, row, column) split without integer divisions
  // mosaic (images smaller than 8x8, one whole image per 16-pixel MFMA fragment): the 2^lni images of a patch are laid out as a
  // 2^lmc-column grid that SHARES its zero borders — cell pitch (th+1) x (tw+1), one separator row / column between and around
  // the images — so the patch is ONE halo of (rows*(th+1)+1) x (cols*(tw+1)+1) pixels: 32 images of 4x4 = 21 x 41 = 861 rows
  // (7 LDS-DMA pieces per lane) instead of 32 separate 6x6 halos (1152 rows, 9 pieces).  A tap is still one row offset.
  int mos, lmc;
  float inv_ch, inv_cw;     // 1/(th+1), 1/(tw+1)
  // xbuf: one image per patch and every source image below 2 GiB -> the halo pieces are fetched through per-image buffer
  // descriptors (buffer_load ... lds: 32-bit per-lane offset, hardware range check = zero padding) instead of 64-bit
  // per-lane addresses + zero page: ~6 VALU per chunk instead of ~150.
  int xbuf;
  // sws: chunk swizzle of the halo image.  The image keeps one 64-byte row per halo pixel, but the 16-byte slot c of a row holds the
  // row's logical chunk c ^ 2 * ((X >> sws) & 1), X = the row's halo column.  A ds_read_b128 of a pixel fragment is served in four
  // 16-lane groups that mix two values of lane >> 4 (MI355X_MICROARCH.md, LDS table); un-swizzled, two of every group's lanes met on
  // each bank (2-way conflicts on all eight activation reads of a tap: 640 of a 1024-cycle tap's LDS cycles for eight waves).  With
  // sws = 2 for fragments of 16 pixels in a row, 1 for 2 x 8, 0 for 4 x 4 (mosaic) every group covers the 64 banks exactly once for
  // every tap offset (exhaustive check over alignments: tests/test_halo_swizzle.py; SQ_LDS_BANK_CONFLICT 0.36-0.40 of SQ_LDS_IDX_ACTIVE
  // -> 0 on every kernel that has it, profiles/r03_lds_bank_conflicts.log) — no pitch change, the loaders fetch the other chunk, the
  // readers keep one per-lane offset per column offset of the tap.  Used by the lock-step loops of conv3_halo.hip and by conv3_ws_kernel;
  // the staggered loop and the opt-in persistent kernels keep the plain image (conv3_halo.hip says why).
  int sws;
  int lpt;                  // log2 of the tiles per image (producer-side GroupNorm: the workgroups that share a sample's statistics)
};


constexpr int HALO_WST = 128 * 64;                      // bytes per W tap tile

template <int NW> struct HaloCfg {
  static constexpr int NT = NW * 64;                    // threads
  static constexpr int PIX = NW * 64;                   // output pixels per workgroup
  static constexpr int NXL = NW == 4 ? 6 : 7;           // max LDS-DMA instructions per lane per halo (8x8 images: 8 x 100 rows)
  static constexpr int XBUF = NXL * NT * 16;            // bytes per X halo buffer (NXL instructions x NT lanes x 16 B)
  static constexpr int XROWS = NXL * NT / 4;
  static constexpr int WLD = 512 / NT;                  // W LDS-DMA instructions per lane per tap (8 KiB tile)
  static constexpr int WR = NW == 4 ? 3 : 4;            // W ring stages (prefetch distance WR-1 taps); 3 keeps NW=4 at 72 KiB -> 2 per CU
  static constexpr int GNOFF = 2 * XBUF + WR * HALO_WST; // fused-GroupNorm affine of the workgroup's sample: scale[C], shift[C]
  static constexpr int GNMAXC = 512;
  static constexpr int TBLOFF = GNOFF + 2 * GNMAXC * 4;  // per-image sample bases of the sources: int [ni <= 8][4] (src0, src1, src2, -)
  static constexpr int LDS_MAIN = TBLOFF + 128;
  static constexpr int LDS = LDS_MAIN;
};

// W tile: 64-byte rows, 4 rows per 256-B bank row.  ds_read_b128 is served in 16-lane groups that MIX two values
// of lane>>4, so the chunk swizzle is g(q) = [0,2,3,1][(row>>2)&3] (conflict-free for every service group).
__device__ __forceinline__ int swz64(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }
__device__ __forceinline__ int lds64_off(int row, int chunk) { return row * 64 + ((chunk ^ swz64(row)) << 4); }


// quad statistics (IgemmArgs::qstats): a wave's 128 pixels are one part of one image, or (8x8 images) two whole images
struct HaloQs {
  static constexpr bool on = true;
  int nbase, ltp, n_img, tile_in_img, wm, np, padd;
  __device__ __forceinline__ bool whole() const { return ltp >= 7; }
  __device__ __forceinline__ int parts() const { return np; }
  __device__ __forceinline__ bool operator()(int half, int& n, int& part) const {
    const int p0 = wm * 128 + half * 64;
    n = nbase + (p0 >> ltp);
    part = padd + (ltp >= 7 ? (tile_in_img << (ltp - 7)) + ((p0 & ((1 << ltp) - 1)) >> 7) : tile_in_img);
    return n < n_img;
  }
};

// bias (+ per-sample row vector) of the workgroup's N tile staged in LDS at kernel start (igemm_epilogue.h BiasFn): run k of the
// lane starts 32 k channels after p
struct HaloLdsBias {
  static constexpr bool on = true, has_rowvec = true;
  const float* p;
  __device__ __forceinline__ void operator()(int k, float (&bs)[8], float (&)[8]) const {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p + 32 * k), hi = *reinterpret_cast<const f32x4*>(p + 32 * k + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bs[e] = lo[e]; bs[4 + e] = hi[e]; }
  }
};

template <int N> __device__ __forceinline__ void hwait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int V> struct IC { static constexpr int value = V; };

