// conv3_halo.hip — 3x3 / stride-1 / pad-1 convolution as a halo-tile direct convolution on MFMA.
//
// Why: the tap-refetch implicit GEMM (igemm_pipe.hip) re-reads every input pixel 9 times from
// L2 — rocprofv3 showed its waves parked on vmcnt/barrier 51 % of the time at ~20-30 GB/s of
// LDS-DMA per CU.  Here LDS is the reuse level for the taps: a workgroup owns an output patch
// (ni images x th x tw pixels) x 128 couts and, per 64-byte channel chunk (32 bf16/f16 or 16 f32
// channels), brings the (th+2)x(tw+2) input halo ONCE into LDS; the 9 taps are just 9 different
// row offsets of the MFMA B-operand fragment reads.
//
// Two geometries of the same kernel (template NW = waves per workgroup):
//   NW = 4: 256-pixel patch, 4 waves as 2(M) x 2(N), ~76 KiB of LDS -> TWO workgroups per CU.  One
//           workgroup's HBM-bound prologue / epilogue overlaps the other's MFMA loop, and the two
//           waves that share a SIMD belong to different workgroups, so they are not barrier-locked
//           into reading fragments and issuing MFMAs at the same moments (the 8-wave form measured
//           ~2,000 cycles per tap against 1,024 of MFMA work).  [default]
//   NW = 8: 512-pixel patch, 8 waves as 4(M) x 2(N), 147 KiB, one workgroup per CU: half the weight
//           traffic per pixel (images of 8x8 and below).
//   Wave tile 128 pixels x 64 couts (32 MFMA 16x16x32 per tap) in both.
//
//   X halo: double-buffered per channel chunk (the next chunk's halo is fetched during the current
//           chunk's 9 taps), stored un-swizzled so a fragment address is "per-lane constant + tap offset";
//   W[tap] tiles (128 couts x 64 B): 4-stage ring, prefetch distance 3 taps, XOR-swizzled.
//   All transfers are LDS-DMA (global_load_lds_dwordx4), counted vmcnt, one s_barrier per tap.
//   Padding / out-of-range images read a zero page; the epilogue runs straight from the accumulators
//   (igemm_epilogue.h): weight rows enter LDS permuted so that a lane ends up with runs of 8 consecutive couts.
#include <stdlib.h>
#include "common.h"
#ifdef DC_STAMPS
// diagnostic build only: per-block s_memtime stamps (never compiled into the shipped library)
static __device__ unsigned long long* g_stamps;
extern "C" void dc_debug_set_stamps(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }
#define DC_STAMP(k) do { if (threadIdx.x == 0 && g_stamps) g_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
// timing-only ablations of the staggered loop (results wrong on purpose): 1 no MFMAs, 2 no fragment reads, 4 no LDS-DMA instructions,
// 8 no halo LDS-DMA instructions (the W tiles are still fetched)
static __device__ int g_halo_abl;
extern "C" void dc_debug_set_halo_abl(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_halo_abl), &v, sizeof(v)); }
#define DC_HALO_ABL() __builtin_amdgcn_readfirstlane(g_halo_abl)
#else
#define DC_STAMP(k) do {} while (0)
#define DC_HALO_ABL() 0
#endif
#include "igemm_epilogue.h"
#include "epi_pn.h"

static __device__ unsigned g_pn_timeouts;     // producer-side GroupNorm: waves whose wait for their sample's statistics ran out (epi_pn.h)
static __device__ chunk16 g_zero_page[16];   // per translation unit (no device-side linking)
#include "conv3_halo.h"
DC_CLOCK_DECL(conv3_halo)

// NTAP: 9 = the 3x3 conv.  4 = one PHASE of "nearest-2x upsample, then 3x3 conv" (dc_igemm_params.up4): output pixel
// (2y+pa, 2x+pb) only sees the 2x2 source pixels (y+pa-1+dy, x+pb-1+dx), with the 3x3 taps that fall on the same source
// pixel summed when the weights are packed — 4 taps instead of 9, the same halo of the LOW-resolution image; the four
// phases are four times the N tiles of the grid (g.* then describes the low-resolution image).
// MODE: 0 = images of at least 8x8, per-lane 64-bit addresses; 1 (XB) = buffer-descriptor loaders (HaloGeom::xbuf: one image per
// patch); 2 (MOS) = mosaic patches of images below 8x8 (HaloGeom::mos; a wave's pixels then span eight samples, so the epilogue
// fetches the per-sample row vector per pixel fragment).  Separate instantiations, so that no form carries another's registers
// (the kernel sits at the SGPR / VGPR limits of two waves per SIMD).
// STG (8 waves): the tap loop in HALF-steps with the two wave groups (waves 0-3 / 4-7 = the two waves of every
// SIMD) running ONE BARRIER APART — the schedule of igemm_wide.hip applied to the halo loop.  A step is
//   b ; R_a { issue W(s+PD) ; read the 4 W fragments and pixel fragments 0-3 } ; b ; M_a { 16 MFMAs } ;
//   b ; R_b { tap 0: issue X(cc+1) ; read pixel fragments 4-7 ; wait for my piece of W(s+1) } ; b ; M_b { 16 MFMAs }
// and group B executes one barrier more in front of the loop (A one behind it): while one wave of a SIMD issues its 16 MFMAs the
// other reads fragments and issues LDS-DMA.  In the lock-step loop both waves of a SIMD read at the same moment and then share the
// matrix pipe at the same moment (asymptote 1.35 PF, fixed cost 26 tap-times, 2.33 GHz held: stall-bound — DESIGN 6c).
// Hazards, barrier instance n (A: R_a(s) in (4s+1, 4s+2), M_a (4s+2, 4s+3), R_b (4s+3, 4s+4), M_b (4s+4, 4s+5); B one later):
//   RAW  W(s+1) / X(cc+1) are first read by A after instance 4s+5 (R_a(s+1)); every wave waits for its own pieces before its last
//        barrier of step s (A: instance 4s+4, B: 4s+5).
//   WAR  W(s+PD) goes into the slot of W(s-1) (WR = PD+1): its last readers (B, R_a(s-1)) retire their reads before they arrive at
//        instance 4s; the earliest re-stage (A, R_a(s)) comes after instance 4s+1.  X(cc+1) goes into X(cc-1)'s buffer: last read by
//        B in R_b of the previous chunk's last step, retired before B arrives at the instance A's R_b of tap 0 waits behind.
// One accumulator gets one MFMA per step in both loops: results are bit-identical.
// PN (one image per patch, 4 waves, 3x3): producer-side GroupNorm — the output is stored normalised for the GroupNorm that consumes it
// (epi_pn.h).  The workgroups of one (sample, N tile) then sit on consecutive block indices (see there).
template <typename T, int NW, int NTAP = 9, int MODE = 0, bool STG = false, bool PN = false>
__global__ __launch_bounds__(NW * 64, 2) void conv3_halo_kernel(const IgemmArgs a, const HaloGeom g) {
  // PN with NW = 4: images that span workgroups (statistics exchanged through memory, epi_pn.h).  PN with NW = 8 (8x8 images, staggered loop):
  // every image lies inside one wave, the statistics never leave it (igemm_epilogue.h, EpiPnLocal8x8).
  static_assert(!PN || (MODE == 1 && NW == 4 && !STG) || ((MODE == 0 || MODE == 2) && NW == 8 && STG && NTAP == 9), "producer-side GroupNorm: one-image-per-patch forms, or 8x8 / 4x4 images");
  constexpr bool PNX = PN && NW == 4, PNL = PN && NW == 8;
  static_assert(!STG || NW == 8, "staggered loop: the 8-wave kernel");
  constexpr bool XB = MODE == 1, MOS = MODE == 2;
  using Cfg = HaloCfg<NW>;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 4 * EPC;                  // channels per 64-byte chunk row
  constexpr int TM = 8, TN = 4;
  constexpr int NT = Cfg::NT, WLD = Cfg::WLD, WR = Cfg::WR, PD = Cfg::WR - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Wring = smem + 2 * Cfg::XBUF;

  DC_STAMP(0);
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // NW/2 waves along pixels, 2 along couts
  const int lr = lane & 15, lq = lane >> 4;
  int tile_m, tile_n;
  if constexpr (PNX) {
    // The workgroups that share a sample's statistics — a group: (sample, N tile) x the 2^lpt tiles of the image — sit on CONSECUTIVE
    // positions of ONE XCD's dispatch queue: blocks are dealt round-robin to the 8 XCDs (b and b + 8 share one), so position k = b >> 3
    // of queue b & 7 is tile k & (2^lpt - 1) of that queue's group k >> lpt, and the groups are dealt round-robin to the queues.  The
    // grid is padded to a multiple of 8 groups; the surplus workgroups leave at once.  (With the groups on consecutive BLOCK indices
    // instead, their members sat in different queues, the queues drifted apart, and a 16-tile group waited 30 k cycles of a 60 k-cycle
    // tile for its slowest member — stamps, tools/stamp_pn.py; correctness never depends on the placement, see epi_pn.h.)
    // Four-phase upsample conv (NTAP = 4): the output sample is written by the four phases of every low-resolution tile, so a group is
    // 4 x 2^lpt workgroups — position (phase, tile) inside it — and a.tiles_n counts phase-major N tiles (4 x the real ones).
    constexpr int LPH = NTAP == 4 ? 2 : 0;
    const int tnr = a.tiles_n >> LPH;                         // real N tiles
    const int kq = blockIdx.x >> 3;
    const int grp = ((kq >> (g.lpt + LPH)) << 3) + (blockIdx.x & 7);
    if (grp >= g.n_img * tnr) return;
    const int smp = grp / tnr;
    const int within = kq & ((1 << (g.lpt + LPH)) - 1);
    tile_n = __builtin_amdgcn_readfirstlane((within >> g.lpt) * tnr + grp - smp * tnr);
    tile_m = __builtin_amdgcn_readfirstlane((smp << g.lpt) + (within & ((1 << g.lpt) - 1)));
    asm volatile("" : "+s"(tile_m), "+s"(tile_n));
  } else if constexpr (XB) tile_of_block_scalar(a, tile_m, tile_n);     // buffer-descriptor loaders: scalar offsets must be SGPRs (no waterfall loops)
  else tile_of_block(a, tile_m, tile_n);
  constexpr bool UP4 = NTAP == 4;
  int phase = 0;
  if (UP4) { const int tn = a.tiles_n >> 2; phase = tile_n / tn; tile_n -= phase * tn; }
  const int pa = phase >> 1, pb = phase & 1;
  const int tx = tile_m % g.tiles_x;
  const int ty = (tile_m / g.tiles_x) % g.tiles_y;
  const int ng = tile_m / (g.tiles_x * g.tiles_y);
  const int tw = 1 << g.ltw, th = 1 << g.lth;
  const int HW = g.H * g.W;                             // g.H x g.W: the conv's (= output) extent; the source is half of it when upsampling
  const int HWs = a.upsample ? (g.H >> 1) * (g.W >> 1) : HW;
  const int Ctot = a.C0 + a.C1;
  const int c0chunks = a.C0 / BKE, nchunks = Ctot / BKE;

  // ---- X loader: lane fetches LDS position p = i*NT + t  -> halo row p>>2, chunk p&3 (image is not swizzled) ----
  constexpr int NXL = Cfg::NXL;
  // per piece: (image of the patch << 20) | pixel offset inside the sample, -1 = padding (zero page); the sample bases
  // (map lookups) live in a small LDS table, so the loader keeps ONE register per piece whatever the source count
  int pp[NXL];
  const int xlx = t & 3;
  int* const tbl = reinterpret_cast<int*>(smem + Cfg::TBLOFF);
  if (t < (1 << g.lni)) {
    const int n = (ng << g.lni) + t;
    const bool vn = n < g.n_img;
    tbl[4 * t] = vn ? (a.map0 ? a.map0[n] : n) * HWs : -1;
    tbl[4 * t + 1] = (vn && a.src1) ? (a.map1 ? a.map1[n] : n) * HWs : -1;
    tbl[4 * t + 2] = (vn && a.src2) ? (a.map2 ? a.map2[n] : n) * HWs : -1;
  }
  // piece i of this lane: (image of the patch << 20) | pixel offset inside the sample, or -1 (padding)
  auto piece_code = [&](int i, int* col = nullptr) -> int {
    const int hr = (i * NT + t) >> 2;
    int code = -1;
    if (col) *col = 0;
    if (i < g.nxl && hr < g.HR) {
      // hr < 2^12 and (hr + 0.5) / hp is at least 0.5 / hp away from an integer: the fp32 product floors exactly
      if constexpr (MOS) {
        // mosaic: halo position (hy, hx) -> grid cell (cy, cx) and position inside the cell; row / column 0 of a cell is a shared
        // zero separator (all quotients are < 2^10 and at least half a step away from an integer: the fp32 products floor exactly)
        const int hy = (int)(((float)hr + 0.5f) * g.inv_hw), hx = hr - hy * g.hw;
        const int cy = (int)(((float)hy + 0.5f) * g.inv_ch), ry = hy - cy * (th + 1);
        const int cx = (int)(((float)hx + 0.5f) * g.inv_cw), rx = hx - cx * (tw + 1);
        if (col) *col = hx;
        if (ry > 0 && rx > 0) code = (((cy << g.lmc) + cx) << 20) | ((ry - 1) * g.W + rx - 1);
      } else {
        const int img = (int)(((float)hr + 0.5f) * g.inv_hp), r = hr - img * g.hp;
        const int hy = (int)(((float)r + 0.5f) * g.inv_hw), hx = r - hy * g.hw;
        const int iy = ty * th + hy - 1, ix = tx * tw + hx - 1;
        if (col) *col = hx;
        // nearest-2x upsample folded into the gather: halo pixel (iy, ix) of the upsampled image reads source (iy>>1, ix>>1)
        if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
          code = (img << 20) | (a.upsample ? (iy >> 1) * (g.W >> 1) + (ix >> 1) : iy * g.W + ix);
      }
    }
    return code;
  };
  // chunk swizzle of the halo image (conv3_halo.h, HaloGeom::sws): the 16-byte slot p & 3 of LDS row p >> 2 holds the row's LOGICAL chunk
  // (p & 3) ^ 2 * bit(sws) of the row's halo column.  LDS-DMA fixes the slot a lane writes, so the lane fetches the other chunk: one bit
  // per piece, carried in bit 30 of the piece's code (these kernels have no register to spare).
  // The staggered loop keeps the UN-swizzled image: its eight reads of a read phase then share one address add per tap row (the column
  // offset is the instruction's immediate), with the swizzle every read needs its own — measured 13.3 -> 13.8 ms per cfg2 step for
  // conv3_halo<8w>, while the conflicts it removes (40 % of the LDS cycles by SQ_LDS_BANK_CONFLICT) cost that loop no time.
  constexpr bool SWP = !STG;
#pragma unroll
  for (int i = 0; i < NXL; ++i) {
    int col;
    pp[i] = piece_code(i, &col);
    if constexpr (SWP) { if (pp[i] >= 0) pp[i] |= ((col >> g.sws) & 1) << 30; }
  }
  auto sw_of = [&](int pk) -> int { if constexpr (SWP) return (pk >> 30) & 1; else return 0; };
  constexpr int PKMASK = SWP ? 0x3FFFFFFF : -1;
  // XB (one image per patch = one sample per workgroup): bias + the sample's row vector of this N tile, summed once into LDS (the
  // GroupNorm slot, unused here) while the piece offsets are being formed — the epilogue then reads them at LDS latency instead of
  // paying a global-load round trip (~2.5 k of a ~19 k-cycle epilogue by s_memtime stamps) in front of its first use.
  float* const brv = reinterpret_cast<float*>(smem + Cfg::GNOFF);
  constexpr bool STAGE_BRV = XB;
  if (STAGE_BRV && t < 128) {
    const int c = tile_n * 128 + t;
    float v = 0.f;
    if (c < a.Cout) {
      if (a.bias) v = a.bias[c];
      if (a.rowvec) v += a.rowvec[(size_t)(a.rowvec_map ? a.rowvec_map[ng] : ng) * a.rowvec_ld + c];
    }
    brv[t] = v;
  }
  if constexpr (PNX) {     // gamma / beta of the consumer's GroupNorm for this N tile, behind the bias table
    if (t >= 128 && t < 256) {
      const int c = tile_n * 128 + t - 128;
      brv[t] = a.pn_gamma[c];
      brv[t + 128] = a.pn_beta[c];
    }
  }
  __syncthreads();
  const char* zero = reinterpret_cast<const char*>(g_zero_page);      // wave-uniform (any 16 zero bytes do)
  // STG without the buffer-descriptor halo loaders (several images per patch): pp[i] holds the piece's ROW in the tensor of the source
  // being fetched (sample base + pixel, -1 = padding / no such sample), rebuilt when the chunk stream moves to the next source (at
  // most twice per tile) — a piece then costs one 64-bit multiply-add instead of an LDS table lookup in front of ~15 VALU (stamps:
  // 720 cycles per piece in the MFMA block that carries it, tools/stamp_halo.py)
  constexpr bool XROW = STG && !XB;
  auto retarget = [&](int which) {
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int code = piece_code(i);
      int row = -1;
      if (code >= 0) {
        const int b = tbl[4 * (code >> 20) + which];
        if (b >= 0) row = b + (code & 0xFFFFF);
      }
      pp[i] = row;
    }
  };
  if constexpr (XROW) retarget(0);

  // ---- XB: buffer descriptors (wave-uniform by construction: kernel arguments and blockIdx only).  All four share the range
  // (2 GiB - 1: every real offset is below it by the host's checks) and the format word, so only the bases differ.
  // W: per-lane offset = the lane's (row, chunk) inside an N tile — ONE register for the whole kernel — and the (N tile, tap,
  // channel chunk) part in the scalar offset: no vector arithmetic per tap.  X: based at the patch's image of each source; a
  // padding piece gets offset 0xffffffff (out of range: the hardware returns zeros).
  const T* xb0 = nullptr; const T* xb1 = nullptr; const T* xb2 = nullptr;
  int wvoff = 0, ldb0 = 0, ldb1 = 0, ldb2 = 0;
  if (XB) {
    ldb0 = a.ld0 * (int)sizeof(T); ldb1 = a.ld1 * (int)sizeof(T); ldb2 = a.ld2 * (int)sizeof(T);
    asm volatile("" : "+s"(ldb0), "+s"(ldb1), "+s"(ldb2));
    const int n = ng;
    const int s0 = __builtin_amdgcn_readfirstlane(a.map0 ? a.map0[n] : n);
    xb0 = reinterpret_cast<const T*>(a.src0) + (size_t)s0 * HWs * a.ld0;
    if (a.src1) { const int s1 = __builtin_amdgcn_readfirstlane(a.map1 ? a.map1[n] : n); xb1 = reinterpret_cast<const T*>(a.src1) + (size_t)s1 * HWs * a.ld1; }
    if (a.src2) { const int s2 = __builtin_amdgcn_readfirstlane(a.map2 ? a.map2[n] : n); xb2 = reinterpret_cast<const T*>(a.src2) + (size_t)s2 * HWs * a.ld2; }
  }
  if (XB || STG) {                                   // (STG: the W tiles go through the descriptor in every loader mode — no address VALU per issue)
    const int wrow0 = t >> 2;                        // LDS row of the lane's first piece (piece i is 64 rows further when NT == 256)
    wvoff = (epi_wrow(wrow0, false) * a.Ktot + ((t & 3) ^ swz64(wrow0)) * EPC) * (int)sizeof(T);
  }
  auto rsrc_of = [](const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000); };

  auto issue_x = [&](int cc, int i0 = 0, int i1 = 64) {    // cc >= nchunks: chunk cc - nchunks of the 1x1 side source; pieces [i0, i1)
    const int which = cc >= nchunks ? 2 : (cc >= c0chunks ? 1 : 0);
    if constexpr (XB) {
      // (selected from LOCAL copies: a select between fields of the by-value argument struct becomes a select of their addresses,
      //  which moves the whole struct into scratch)
      int ldb = ldb0, cb = cc;
      const T* xb = xb0;
      if (cc >= nchunks) { ldb = ldb2; cb = cc - nchunks; xb = xb2; }
      else if (cc >= c0chunks) { ldb = ldb1; cb = cc - c0chunks; xb = xb1; }
      const int cofs = cb * 64 + xlx * 16;
      const __amdgpu_buffer_rsrc_t rs = rsrc_of(xb);
      char* xs = smem + (cc & 1) * Cfg::XBUF + wave * 1024;
#pragma unroll
      for (int i = 0; i < NXL; ++i) {
        if (i < i0 || i >= i1) continue;
        int pk = pp[i];
        asm volatile("" : "+v"(pk));      // form the offset HERE, once per chunk (hoisted it would cost a register per piece)
        const int voff = pk < 0 ? -1 : (pk & PKMASK) * ldb + (cofs ^ (sw_of(pk) << 5));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(xs + i * (NT * 16)), 16, voff, 0, 0, 0);
      }
      return;
    } else {
    const T* src = reinterpret_cast<const T*>(which == 2 ? a.src2 : (which == 1 ? a.src1 : a.src0));
    const int ld = which == 2 ? a.ld2 : (which == 1 ? a.ld1 : a.ld0);
    const int coff = (which == 2 ? cc - nchunks : (which == 1 ? cc - c0chunks : cc)) * BKE + xlx * EPC;
    char* xs = smem + (cc & 1) * Cfg::XBUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      // always NXL instructions (pieces past the halo fetch the zero page into the unused tail of the buffer):
      // the counted vmcnt waits of the tap loop are then compile-time constants
      if (i < i0 || i >= i1) continue;
      int pk = pp[i];
      asm volatile("" : "+v"(pk));        // decode HERE, once per chunk: hoisted out of the tap loop the fields cost 2 VGPRs per piece
      int rowi;
      if constexpr (XROW) rowi = pk;      // already the row of the source in flight
      else {
        const int base = pk < 0 ? -1 : tbl[4 * ((pk & PKMASK) >> 20) + which];
        rowi = base < 0 ? -1 : base + (pk & 0xFFFFF);
      }
      const size_t e = (size_t)(rowi < 0 ? 0 : rowi) * ld + (coff ^ (sw_of(pk) << 1) * EPC);
      const char* gp = rowi < 0 ? zero : reinterpret_cast<const char*>(src + e);
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(xs + i * (NT * 16)), 16, 0, 0);
    }
    }
  };
  // ---- W loader: 128 couts x 64 B per (chunk, tap): position i*NT + t -> row >>2, phys chunk &3 (swizzled) ----
  // element offset of the lane's W row chunk from the tile's first row (32-bit: Cout_pad*Ktot < 2^31), recomputed at every
  // issue from an opaque copy of t (a handful of VALU per tap) instead of living in VGPRs through the tap loop
  const int wtile0 = ((UP4 ? phase * (a.tiles_n >> 2) : 0) + tile_n) * 128 * a.Ktot;    // element offset of the N tile (up4: [phase][Cout_pad][4 taps * C]); < 2^30 (host check)
  auto issue_w = [&](int cc, int tap, int slot) {
    if constexpr (XB || STG) {
      const int so = (wtile0 + tap * Ctot + cc * BKE) * (int)sizeof(T);                  // wave-uniform: scalar offset
      const __amdgpu_buffer_rsrc_t wrs = rsrc_of(a.W);
#pragma unroll
      for (int i = 0; i < WLD; ++i)    // piece i: LDS rows 64 i + (t >> 2) = packed rows 64 further (epi_wrow and the swizzle keep the low part)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr_t)(Wring + slot * HALO_WST + i * (NT * 16) + wave * 1024), 16, wvoff,
                                                 __builtin_amdgcn_readfirstlane(so + i * 64 * a.Ktot * (int)sizeof(T)), 0, 0);
    } else {
      const T* wb = reinterpret_cast<const T*>(a.W) + (size_t)wtile0 + (size_t)tap * Ctot + (size_t)cc * BKE;   // wave-uniform
      int tt = t;
      asm volatile("" : "+v"(tt));     // recomputed at every issue from an opaque copy of t instead of living in VGPRs through the tap loop
#pragma unroll
      for (int i = 0; i < WLD; ++i) {
        const int row = (i * NT + tt) >> 2;
        const int wro = epi_wrow(row, false) * a.Ktot + ((tt & 3) ^ swz64(row)) * EPC;
        __builtin_amdgcn_global_load_lds((gptr_t) reinterpret_cast<const char*>(wb + wro),
                                         (lptr_t)(Wring + slot * HALO_WST + i * (NT * 16) + wave * 1024), 16, 0, 0);
      }
    }
  };

  // side source weights W2 [Cout_pad][C2]: chunk e of 32 channels, same LDS tile image; the row pointers are rebuilt here
  // (a handful of VALU, nx times per tile) rather than kept in registers through the tap loop
  auto issue_w2 = [&](int e, int slot) {
    int tt = t;
    asm volatile("" : "+v"(tt));          // keep the address arithmetic HERE (hoisted, it would sit in VGPRs through the whole tap loop)
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int row = (i * NT + tt) >> 2;
      const T* wp = reinterpret_cast<const T*>(a.W2) + (size_t)(tile_n * 128 + epi_wrow(row, false)) * a.C2 + ((tt & 3) ^ swz64(row)) * EPC + e * BKE;
      __builtin_amdgcn_global_load_lds((gptr_t) reinterpret_cast<const char*>(wp),
                                       (lptr_t)(Wring + slot * HALO_WST + i * (NT * 16) + wave * 1024), 16, 0, 0);
    }
  };
  const int nx = a.src2 ? a.C2 / BKE : 0;            // 32-channel chunks of the side source (>= 2 when present)

  // ---- fragment read addresses: per-lane part (one register each) + wave-uniform part per fragment (SGPRs) ----
  // pixel p = wm*128 + j*16 + lr: the lr bits never carry into the bit fields set by j (tile widths are >= 8 and a
  // power of two), so the halo offset splits into f(lr, lq) + f(j); cout fragment i is 16 rows = 1024 B further.
  // per-lane part, one per swizzle variant: the chunk slot of a pixel fragment's row depends on bit sws of its halo column
  // X = (lane's column inside the fragment) + kx (+ the fragment's cell column in a mosaic: odd cells for odd j, + the phase's pb in
  // the four-tap form) — variant k = what is added to the lane's column (mosaic: only its parity counts)
  constexpr int NV = 3;
  int xlv[NV];
  {
    const int xx = lr & ((tw < 16 ? tw : 16) - 1);
    const int rowpart = ((lr >> g.ltw) * g.hw + (lr & (tw - 1))) * 64;
#pragma unroll
    for (int k = 0; k < NV; ++k) xlv[k] = rowpart + ((lq ^ (SWP ? (((xx + k + (UP4 ? pb : 0)) >> g.sws) & 1) << 1 : 0)) << 4);
  }
  int joff[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int p = wm * 128 + j * 16;
    const int img = p >> (g.ltw + g.lth), py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
    const int o = MOS ? ((img >> g.lmc) * (th + 1) + py) * g.hw + (img & ((1 << g.lmc) - 1)) * (tw + 1) + px : img * g.hp + py * g.hw + px;
    joff[j] = __builtin_amdgcn_readfirstlane(o * 64);
  }
  const int woff0 = lds64_off(wn * 64 + lr, lq);

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one tap of one channel chunk: W fragments, then the X fragments in two halves that share registers (all eight at once
  // need 16 more VGPRs than this kernel has: a spill inside the tap loop is reloaded behind vmcnt(0), which drains the
  // LDS-DMA ring); MFMAs in j-major order.  The sched_barrier keeps hipcc from hoisting the second half's reads.
  auto mma_tap = [&](const char* Wst, const char* Xb, int tapoff, auto kxc) {
    constexpr int kx = decltype(kxc)::value;      // swizzle variant of the tap's column offset
    chunk16 wf[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const chunk16*>(Wst + woff0 + i * 1024);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      chunk16 xf[TM / 2];
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) xf[j] = *reinterpret_cast<const chunk16*>(Xb + (tapoff + joff[h * (TM / 2) + j]) + xlv[MOS ? ((kx + j) & 1) : kx]);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][h * (TM / 2) + j] = Mma<T>::run(wf[i], xf[j], acc[i][h * (TM / 2) + j]);
      if (h == 0) __builtin_amdgcn_sched_barrier(0);
    }
  };

  DC_STAMP(1);
  if constexpr (STG) {
    const bool grpB = wave >= 4;                       // wave-uniform
    const int abl = DC_HALO_ABL();                     // 0 outside diagnostic builds
    int cur_which = 0;
    issue_x(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) issue_w(0, i, i);
    hwait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (grpB) __builtin_amdgcn_s_barrier();
    DC_CLOCK(0);
    chunk16 wf[TN], xf[TM / 2];
    // fragment reads as opaque instructions with explicit waits tied to their registers (common.h), MFMA blocks fenced with
    // sched_barrier: hipcc otherwise moves the MFMAs across the barriers (two s_barrier back to back in the ISA) and re-serialises
    // the reads.  Addresses: per-lane LDS address of pixel fragment j (8 registers) + a wave-uniform (buffer, tap row) part.
    const uint32_t lds0 = lds_addr_of(smem);
    uint32_t xaddr[TM];                   // per-lane LDS address of pixel fragment j (no swizzle here: xlv[0] = xlv[1] = xlv[2])
#pragma unroll
    for (int j = 0; j < TM; ++j) xaddr[j] = lds0 + (uint32_t)(joff[j] + xlv[0]);
    const uint32_t waddr = lds_addr_of(Wring) + (uint32_t)woff0;
    for (int cc = 0; cc < nchunks; ++cc) {
      const bool side_next = cc + 1 == nchunks && nx > 0;
      const bool has_next = cc + 1 < nchunks || side_next;
      const int s0 = cc * NTAP;
      const uint32_t xbo = (uint32_t)(cc & 1) * Cfg::XBUF;
      if constexpr (XROW) {               // the halo issued during this chunk is chunk cc + 1's
        const int wnext = cc + 1 >= nchunks ? 2 : (cc + 1 >= c0chunks ? 1 : 0);
        if (has_next && wnext != cur_which) { retarget(wnext); cur_which = wnext; }
      }
      auto step = [&](auto tapc) {
        constexpr int tap = decltype(tapc)::value;
        constexpr int ky = UP4 ? (tap >> 1) : tap / 3, kx = UP4 ? (tap & 1) : tap - ky * 3;
        const uint32_t rowo = xbo + (uint32_t)(((ky + pa) * g.hw + pb) * 64);      // wave-uniform; kx * 64 is the instruction's immediate
        const uint32_t wso = (uint32_t)((s0 + tap) % WR) * HALO_WST;
        // ---- half a: reads, then 16 MFMAs with this step's W LDS-DMA after the fourth (an LDS-DMA issued among MFMAs costs the
        // wave ~60 cycles; issued in front of the fragment reads it held the whole read phase up: 1880 -> 1266 cycles per step
        // without the instruction, tools/stamp_halo.py ablations) ----
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 2)) {
        wf[0] = ds_read16_async_off<0>(waddr + wso);
        wf[1] = ds_read16_async_off<1024>(waddr + wso);
        wf[2] = ds_read16_async_off<2048>(waddr + wso);
        wf[3] = ds_read16_async_off<3072>(waddr + wso);
#pragma unroll
        for (int j = 0; j < TM / 2; ++j) xf[j] = ds_read16_async_off<kx * 64>(xaddr[j] + rowo);
        }
        constexpr int t2 = tap + PD;
#ifndef DC_STG_W_IN_M     // the W tile's LDS-DMA goes out in the read phase, behind the fragment reads (inside the MFMA block: 13.6 vs 13.2 ms)
        if (!(abl & 4)) {
        if (t2 < NTAP) issue_w(cc, t2, (s0 + t2) % WR);
        else if (side_next) { if (t2 - NTAP < nx) issue_w2(t2 - NTAP, (s0 + t2) % WR); }
        else if (has_next) issue_w(cc + 1, t2 - NTAP, (s0 + t2) % WR);
        }
#endif
#ifdef DC_STG_WAIT_FIRST  // experiment: retire the fragment reads in front of the barrier instead of behind it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]));
        __builtin_amdgcn_s_barrier();
#else
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]));
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 1)) {
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][0] = Mma<T>::run(wf[i], xf[0], acc[i][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef DC_STG_W_IN_M
        if (!(abl & 4)) {
        if (t2 < NTAP) issue_w(cc, t2, (s0 + t2) % WR);
        else if (side_next) { if (t2 - NTAP < nx) issue_w2(t2 - NTAP, (s0 + t2) % WR); }
        else if (has_next) issue_w(cc + 1, t2 - NTAP, (s0 + t2) % WR);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 1)) {
#pragma unroll
        for (int j = 1; j < TM / 2; ++j)
#pragma unroll
          for (int i = 0; i < TN; ++i) acc[i][j] = Mma<T>::run(wf[i], xf[j], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- half b: reads of pixel fragments 4-7; the NEXT chunk's halo pieces ride inside the MFMA block (3x3: two at tap 0, one at
        // taps 1-5; four-tap form: all at tap 0) ----
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 2)) {
#pragma unroll
        for (int j = 0; j < TM / 2; ++j) xf[j] = ds_read16_async_off<kx * 64>(xaddr[TM / 2 + j] + rowo);
        }
        // my piece(s) of W(s+1) must be in before my last barrier of this step (they are first read right behind the first barrier of
        // the next step, by the other group first).  Issue order per wave: ... W(s+1) | X pieces of tap s-2 | W(s+2) | X pieces of tap
        // s-1 | W(s+3) | this wait: the younger groups that may stay in flight are the W groups that exist and the X pieces of the two
        // previous taps.  At the chunk's last tap no X piece is younger: the whole next halo is in.
        constexpr int XP_A = UP4 ? (tap == 1 ? NXL : 0) : (tap == 1 ? 2 : (tap >= 2 && tap <= 6 ? 1 : 0));     // pieces issued at tap - 1
#ifndef DC_STG_X_IN_M     // the halo pieces go out in the read phase, in front of this wait (inside the MFMA block — DC_STG_X_IN_M — they idle the
                          // matrix pipe for the ~470 cycles an HBM-bound piece takes to issue: conv3_halo<8w> 15.3 vs 13.9 ms per cfg2 step)
        constexpr int XP_C = UP4 ? (tap == 0 ? NXL : 0) : (tap == 0 ? 2 : (tap <= 5 ? 1 : 0));               // pieces issued at this tap
        if (has_next && !(abl & 12)) {
          if constexpr (UP4) { if (tap == 0) issue_x(cc + 1); }
          else if constexpr (tap == 0) issue_x(cc + 1, 0, 2);
          else if constexpr (tap <= 5) issue_x(cc + 1, tap + 1, tap + 2);
        }
        constexpr int XP_B = XP_C + (UP4 ? (tap == 2 ? NXL : 0) : (tap == 2 ? 2 : (tap >= 3 && tap <= 7 ? 1 : 0)));
#else
        constexpr int XP_B = UP4 ? (tap == 2 ? NXL : 0) : (tap == 2 ? 2 : (tap >= 3 && tap <= 7 ? 1 : 0));     // pieces issued at tap - 2
#endif
        if (has_next) {
          if (tap == NTAP - 1 && side_next && nx < PD) hwait_vmcnt<0>();      // fewer side-source W tiles than the prefetch distance
          else hwait_vmcnt<(PD - 1) * WLD + XP_A + XP_B>();
        } else {
          constexpr int left = NTAP - 1 - tap;          // W groups behind this step's
          if constexpr (left > 0) hwait_vmcnt<((left - 1) < (PD - 1) ? (left - 1) : (PD - 1)) * WLD>();
        }
#ifdef DC_STG_WAIT_FIRST
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]));
        __builtin_amdgcn_s_barrier();
#else
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]));
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 1)) {
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][TM / 2] = Mma<T>::run(wf[i], xf[0], acc[i][TM / 2]);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef DC_STG_X_IN_M
        if (has_next && !(abl & 12)) {
          if constexpr (UP4) { if (tap == 0) issue_x(cc + 1); }
          else if constexpr (tap == 0) issue_x(cc + 1, 0, 2);
          else if constexpr (tap <= 5) issue_x(cc + 1, tap + 1, tap + 2);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 1)) {
#pragma unroll
        for (int j = 1; j < TM / 2; ++j)
#pragma unroll
          for (int i = 0; i < TN; ++i) acc[i][TM / 2 + j] = Mma<T>::run(wf[i], xf[j], acc[i][TM / 2 + j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{});
      if constexpr (NTAP == 9) { step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{}); step(IC<8>{}); }
    }
    if (!grpB) __builtin_amdgcn_s_barrier();           // the groups are level again (the side-source steps below run in lock-step)
    DC_CLOCK(1);
  } else {
  // ---- tap loop: the 9 taps of a channel chunk are unrolled, so tap offsets, ring slots and every counted vmcnt
  // are compile-time constants (the rolled loop spent ~300 cycles of scalar control per 512-cycle MFMA block).
  // W(s) sits in ring slot s % WR, prefetch distance PD; X(cc+1) is issued at tap 0 behind W(s+PD). ----
  issue_x(0);
#pragma unroll
  for (int i = 0; i < PD; ++i) issue_w(0, i, i);     // a chunk has NTAP taps >= PD
  constexpr int FLYL = (PD - 1) * WLD;               // W(s+1) .. W(s+PD-1)
  DC_CLOCK(0);
  for (int cc = 0; cc < nchunks; ++cc) {
    // "has_next": another X chunk and more W groups follow this chunk — the next 3x3 chunk, or the first chunk of the
    // 1x1 side source, whose W2 tiles simply continue the W stream (s >= NS)
    const bool side_next = cc + 1 == nchunks && nx > 0;
    const bool has_next = cc + 1 < nchunks || side_next;
    const int s0 = cc * NTAP;
    const char* Xb = smem + (cc & 1) * Cfg::XBUF;
    auto step = [&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
      // W(s) (and X(cc) when tap == 0) must have landed.  Younger LDS-DMA groups that may stay in flight: W(s+1) ..
      // W(s+PD-1) and, for tap in 1..PD, the NXL loads of X(cc+1) issued at tap 0; the last chunk has fewer W groups left.
      if (has_next) {
        if (tap >= 1 && tap <= PD) hwait_vmcnt<FLYL + NXL>();
        else hwait_vmcnt<FLYL>();
      } else {
        constexpr int left = NTAP - 1 - tap;        // W groups behind this one
        hwait_vmcnt<(left < PD - 1 ? left : PD - 1) * WLD>();
      }
      __builtin_amdgcn_s_barrier();
      constexpr int t2 = tap + PD;                    // the W group to issue now: s + PD
      if (t2 < NTAP) issue_w(cc, t2, (s0 + t2) % WR);
      else if (side_next) { if (t2 - NTAP < nx) issue_w2(t2 - NTAP, (s0 + t2) % WR); }
      else if (has_next) issue_w(cc + 1, t2 - NTAP, (s0 + t2) % WR);
      if (tap == 0 && has_next) issue_x(cc + 1);

      const char* Wst = Wring + ((s0 + tap) % WR) * HALO_WST;
      constexpr int ky = UP4 ? (tap >> 1) : tap / 3, kx = UP4 ? (tap & 1) : tap - ky * 3;
      const int tapoff = ((ky + pa) * g.hw + kx + pb) * 64;       // pa = pb = 0 for the 3x3 conv
      mma_tap(Wst, Xb, tapoff, IC<kx>{});
    };
    step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{});
    if constexpr (NTAP == 9) { step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{}); step(IC<8>{}); }
  }

  DC_CLOCK(1);
  }
  constexpr int FLY = (PD - 1) * WLD;
  // ---- 1x1 side source (a ResNet's conv_shortcut folded into its conv2): nx steps of ONE tap (the centre) each.  X2(0)
  // and W2(0 .. PD-1) were issued inside the last 3x3 chunk; a step needs a fresh halo chunk per 32 MFMAs, so from the second
  // step on the loads of the previous step are simply drained (vmcnt 0): ~2 k exposed cycles per step, against the whole
  // shortcut GEMM launch and the residual round trip this replaces. ----
  {
    const int NSm = nchunks * NTAP;
    for (int e = 0; e < nx; ++e) {
      if (e == 0 && nx >= PD) hwait_vmcnt<FLY>();      // W2(1 .. PD-1) may stay in flight; X2(0) landed long ago
      else hwait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (e + PD < nx) issue_w2(e + PD, (NSm + e + PD) % WR);
      if (e + 1 < nx) issue_x(nchunks + e + 1);
      const char* Xb = smem + ((nchunks + e) & 1) * Cfg::XBUF;
      const char* Wst = Wring + ((NSm + e) % WR) * HALO_WST;
      const int tapoff = (g.hw + 1) * 64;              // centre tap
      mma_tap(Wst, Xb, tapoff, IC<1>{});
    }
  }

  DC_STAMP(2);
  // ---- epilogue: straight from the accumulators (igemm_epilogue.h: the weight rows were loaded permuted) ----
  const int nw0 = min((ng << g.lni) + ((wm * 128) >> (g.ltw + g.lth)), g.n_img - 1);
  const int nw1 = min((ng << g.lni) + ((wm * 128 + 127) >> (g.ltw + g.lth)), g.n_img - 1);
  HaloQs qsfn;
  qsfn.nbase = ng << g.lni; qsfn.ltp = g.ltw + g.lth; qsfn.n_img = g.n_img; qsfn.tile_in_img = ty * g.tiles_x + tx; qsfn.wm = wm;
  qsfn.np = HW >= 128 ? HW >> 7 : 1;
  qsfn.padd = 0;
  if (UP4) { qsfn.padd = phase * qsfn.np; qsfn.np *= 4; }       // every phase contributes its own parts of the output sample
  auto rowfn = [&](int j, EpiRow& r) {
    const int p = wm * 128 + j * 16 + lr;
    int n = (ng << g.lni) + (p >> (g.ltw + g.lth));
    r.ok = n < g.n_img;
    n = r.ok ? n : g.n_img - 1;
    const int py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
    const int rem = (ty * th + py) * g.W + tx * tw + px;
    r.samp = n;
    r.o = UP4 ? n * (4 * HW) + (2 * (ty * th + py) + pa) * (2 * g.W) + 2 * (tx * tw + px) + pb : n * HW + rem;
    r.r = (a.residual && a.res_map ? a.res_map[n] : n) * HW + rem;
  };
  if constexpr (PNL && MOS) {
    epi_direct_act<T, TM, DC_ACT_NONE, false, false>(a, acc, tile_n, wn, lq, nw0, nw1, rowfn, EpiNoPre(), qsfn, EpiNoBias(), EpiPnLocal4x4());
  } else if constexpr (PNL) {
    epi_direct_act<T, TM, DC_ACT_NONE, false, true>(a, acc, tile_n, wn, lq, nw0, nw1, rowfn, EpiNoPre(), qsfn, EpiNoBias(), EpiPnLocal8x8());
  } else if constexpr (PNX) {
    PnCtx pc;
    // (four-phase upsample: every phase contributes its own parts of the output sample, and all four wait for each other)
    pc.sample = ng; pc.part = (UP4 ? phase * (HW >> 7) : 0) + ((ty * g.tiles_x + tx) << 1) + wm;
    pc.parts = (HW >> 7) << (UP4 ? 2 : 0); pc.tiles = 1 << (g.lpt + (UP4 ? 2 : 0));
    pc.qpg = (a.Cout / a.pn_groups) >> 2;
    pc.cnt = a.pn_cnt + (size_t)ng * (UP4 ? a.tiles_n >> 2 : a.tiles_n) + tile_n; pc.timeouts = &g_pn_timeouts;
    pc.gam = brv + 128 + wn * 64 + lq * 8; pc.bet = brv + 256 + wn * 64 + lq * 8;
    // records of the wave's 64 channels: a sample of several tiles -> per wave (<= 32 parts x 128 B); one tile -> the two parts meet in an
    // area shared by the two waves of the N half
    pc.scr = reinterpret_cast<float2*>(pc.tiles > 1 ? smem + wave * 4096 : smem + wn * 256);
    pc.flag = reinterpret_cast<int*>(brv + 384);
    pc.eps = a.pn_eps; pc.silu = a.pn_silu;
    epi_halo_pn<T>(a, acc, tile_n, wn, lq, rowfn, brv + wn * 64 + lq * 8, pc);
  } else if constexpr (STAGE_BRV) {
    epi_direct_act<T, TM, DC_ACT_NONE, false, true>(a, acc, tile_n, wn, lq, nw0, nw1, rowfn, EpiNoPre(), qsfn, HaloLdsBias{brv + wn * 64 + lq * 8});
  } else {
    epi_direct_act<T, TM, DC_ACT_NONE, false, !MOS>(a, acc, tile_n, wn, lq, nw0, nw1, rowfn, EpiNoPre(), qsfn);
  }
  DC_STAMP(7);
}

// ------------------------------------------------------------------------------------------------------------------
// Thin-output variant (Cout <= 16: the UNets' conv_out, 3-12 channels): same 256-pixel halo patch and X loader, but ONE
// 16-cout MFMA fragment; the four waves split the patch (64 pixels each).  The layer reads ~1 GB of activations for 3
// output channels: it is bound by that read, and the tap-gather kernel it replaces (igemm_kernel<128x32>) re-read every
// pixel nine times from L2 (0.93 ms against ~0.3 ms of HBM time).  All nine W[tap] tiles of a chunk (16 couts x 64 B each)
// are fetched with the chunk's halo; one barrier per CHUNK.  Epilogue: bias only, element stores (fp32 or 16-bit output).
struct ThinCfg {
  static constexpr int NT = 256, NXL = 6;
  static constexpr int XBUF = NXL * NT * 16;                 // 24 KiB per halo buffer
  static constexpr int WBUF = 3 * NT * 16;                   // nine 1-KiB tap tiles inside the 12 KiB that 3 x 256 lanes x 16 B write
  static constexpr int TBLOFF = 2 * XBUF + 2 * WBUF;
  static constexpr int LDS = TBLOFF + 128;
  static constexpr int GNOFF = LDS, GNMAXC = 512;            // fused GroupNorm: scale[C], shift[C] of the workgroup's sample
  static constexpr int LDS_GN = GNOFF + 2 * GNMAXC * 4;      // 77.9 KiB: still two workgroups per CU
};

// GN: GroupNorm(+SiLU) applied in place on the landed halo chunk (y = act(x * scale[n][c] + shift[n][c]), rounded to T exactly where
// the GroupNorm kernel would round) — this layer reads 1 GB for 3 output channels and its VALU is idle, so the normalised tensor of
// conv_norm_out need not exist: no GroupNorm launch, 2 of the 3.3 passes over the tensor gone.  Every lane transforms the pieces it
// fetched itself (padding pieces stay zero: the reference pads the NORMALISED tensor); two barriers per chunk instead of one, so
// that the next chunk's fetch is in flight while this one is transformed.
template <typename T, bool GN>
__global__ __launch_bounds__(256, 2) void conv3_thin_kernel(const IgemmArgs a, const HaloGeom g) {
  using Cfg = ThinCfg;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 4 * EPC;
  constexpr int TM = 4, NT = Cfg::NT, NXL = Cfg::NXL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Wl = smem + 2 * Cfg::XBUF;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int tile_m = blockIdx.x;
  const int tx = tile_m % g.tiles_x;
  const int ty = (tile_m / g.tiles_x) % g.tiles_y;
  const int ng = tile_m / (g.tiles_x * g.tiles_y);
  const int tw = 1 << g.ltw, th = 1 << g.lth;
  const int HW = g.H * g.W;
  const int Ctot = a.C0 + a.C1;
  const int c0chunks = a.C0 / BKE, nchunks = Ctot / BKE;

  int pp[NXL];
  const int xlx = t & 3;
  int* const tbl = reinterpret_cast<int*>(smem + Cfg::TBLOFF);
  if (t < (1 << g.lni)) {
    const int n = (ng << g.lni) + t;
    const bool vn = n < g.n_img;
    tbl[4 * t] = vn ? (a.map0 ? a.map0[n] : n) * HW : -1;
    tbl[4 * t + 1] = (vn && a.src1) ? (a.map1 ? a.map1[n] : n) * HW : -1;
  }
#pragma unroll
  for (int i = 0; i < NXL; ++i) {
    const int hr = (i * NT + t) >> 2;
    pp[i] = -1;
    if (i < g.nxl && hr < g.HR) {
      // hr < 2^12 and (hr + 0.5) / hp is at least 0.5 / hp away from an integer: the fp32 product floors exactly
      const int img = (int)(((float)hr + 0.5f) * g.inv_hp), r = hr - img * g.hp;
      const int hy = (int)(((float)r + 0.5f) * g.inv_hw), hx = r - hy * g.hw;
      const int iy = ty * th + hy - 1, ix = tx * tw + hx - 1;
      if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) pp[i] = (img << 20) | (iy * g.W + ix);
    }
  }
  __syncthreads();
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  // W loader: position p = i*256 + t -> tap p >> 6, row (p >> 2) & 15, physical chunk p & 3 (row-swizzled like the wide W tile)
  auto issue = [&](int cc) {
    const int which = cc >= c0chunks ? 1 : 0;
    const T* src = reinterpret_cast<const T*>(which ? a.src1 : a.src0);
    const int ld = which ? a.ld1 : a.ld0;
    const int coff = (which ? cc - c0chunks : cc) * BKE + xlx * EPC;
    char* xs = smem + (cc & 1) * Cfg::XBUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      int pk = pp[i];
      asm volatile("" : "+v"(pk));
      const int base = pk < 0 ? -1 : tbl[4 * (pk >> 20) + which];
      const size_t e = (size_t)(base < 0 ? 0 : base + (pk & 0xFFFFF)) * ld + coff;
      const char* gp = base < 0 ? zero : reinterpret_cast<const char*>(src + e);
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(xs + i * (NT * 16)), 16, 0, 0);
    }
    char* ws = Wl + (cc & 1) * Cfg::WBUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p = i * NT + t;
      const int tap = p >> 6, row = (p >> 2) & 15;
      const bool ok = tap < 9;
      const T* wp = reinterpret_cast<const T*>(a.W) + (size_t)row * a.Ktot + (size_t)(ok ? tap : 0) * Ctot + cc * BKE + ((p & 3) ^ swz64(row)) * EPC;
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? reinterpret_cast<const char*>(wp) : zero), (lptr_t)(ws + i * (NT * 16)), 16, 0, 0);
    }
  };
  // fragment addresses: this wave's 64 pixels = pixel fragments 4*wave .. 4*wave+3 of the patch
  const int xl = ((lr >> g.ltw) * g.hw + (lr & (tw - 1))) * 64 + lq * 16;
  int joff[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int p = wave * 64 + j * 16;
    const int img = p >> (g.ltw + g.lth), py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
    joff[j] = __builtin_amdgcn_readfirstlane((img * g.hp + py * g.hw + px) * 64);
  }
  const int woff0 = lds64_off(lr, lq);
  f32x4 acc[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float* const gnp = reinterpret_cast<float*>(smem + Cfg::GNOFF);
  if constexpr (GN) {
    if (ng < g.n_img)
      for (int c = t; c < Ctot; c += NT) {
        gnp[c] = a.gn_scale[(size_t)ng * Ctot + c];
        gnp[Ctot + c] = a.gn_shift[(size_t)ng * Ctot + c];
      }
    // the table is read by OTHER waves after the chunk loop's first raw s_barrier, which waits for no counter (and hwait_vmcnt
    // covers vmcnt only): drain this wave's LDS writes here
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0)
  }
  issue(0);
  for (int cc = 0; cc < nchunks; ++cc) {
    hwait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                    // chunk cc is in for everyone; everyone is done with chunk cc - 1
    if (cc + 1 < nchunks) issue(cc + 1);
    if constexpr (GN) {
      const float* sc = gnp + cc * BKE + xlx * EPC;
      const float* sh = sc + Ctot;
      float scr[EPC], shr[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) { scr[e] = sc[e]; shr[e] = sh[e]; }
#pragma unroll
      for (int i = 0; i < NXL; ++i) {
        if (pp[i] >= 0) {
          chunk16* q = reinterpret_cast<chunk16*>(smem + (cc & 1) * Cfg::XBUF + (i * NT + t) * 16);
          float f[EPC];
          chunk_to_f<T>(*q, f);
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            float v = f[e] * scr[e] + shr[e];
            if (a.gn_silu) v = silu_t<T>(v);
            f[e] = v;
          }
          *q = f_to_chunk<T>(f);
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): my in-place writes are in LDS
      __builtin_amdgcn_s_barrier();                  // everyone's pieces of chunk cc are normalised
    }
    const char* Xb = smem + (cc & 1) * Cfg::XBUF;
    const char* Wb = Wl + (cc & 1) * Cfg::WBUF;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int tapoff = (ky * g.hw + kx) * 64;
      const chunk16 wf = *reinterpret_cast<const chunk16*>(Wb + tap * 1024 + woff0);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const chunk16 xf = *reinterpret_cast<const chunk16*>(Xb + (tapoff + joff[j]) + xl);
        acc[j] = Mma<T>::run(wf, xf, acc[j]);
      }
    }
  }
  // epilogue: lane holds couts 4 lq + r of pixel lr of fragment j
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int p = wave * 64 + j * 16 + lr;
    const int n = (ng << g.lni) + (p >> (g.ltw + g.lth));
    if (n >= g.n_img) continue;
    const int py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
    const size_t o = ((size_t)n * HW + (ty * th + py) * g.W + tx * tw + px) * a.out_ld;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = lq * 4 + r;
      if (c < a.Cout) store_as(a.out, o + c, a.out_dtype, acc[j][r] + (a.bias ? a.bias[c] : 0.f));
    }
  }
}

// plain bias-only 3x3 stride-1 conv with at most 16 output channels on a power-of-two image of at least 16x16
bool dc_conv3_thin_applicable(const IgemmArgs& a, int dtype) {
  static const bool off = getenv("DCAMD_NO_THIN") != nullptr;
  if (off || a.taps != 9 || a.stride != 1 || a.upsample || a.act != DC_ACT_NONE || a.gate || a.rowvec || a.residual || a.src2) return false;
  if (a.gn_scale && a.C0 + a.C1 > ThinCfg::GNMAXC) return false;       // fused GroupNorm: the sample's affine table must fit its LDS slot
  if (a.Cout > 16) return false;
  const int H = a.Hin, W = a.Win;
  if (H < 16 || W < 16 || (H & (H - 1)) || (W & (W - 1))) return false;
  if ((long long)a.M >= (1LL << 31)) return false;
  const int bke = 64 / dc_dtype_size(dtype);
  return a.C0 % bke == 0 && a.C1 % bke == 0;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// true when the halo kernel can take this problem (3x3 stride 1, pow-2 extents >= 8, no activation / gate)
bool dc_conv3_halo_applicable(const IgemmArgs& a, int dtype) {
  // plain convolutions only (bias / per-sample row vector / residual): the 128-accumulator wave tile leaves room for ONE
  // epilogue variant; an activation or a gate goes to igemm_pipe.hip (the UNets apply SiLU in the GroupNorm pass)
  if (a.taps != 9 || a.stride != 1 || a.act != DC_ACT_NONE || a.gate) return false;
  const int H = a.Hin, W = a.Win;
  static const bool no_mosaic = getenv("DCAMD_NO_MOSAIC") != nullptr;
  const bool mosaic = !no_mosaic && H == 4 && W == 4 && !a.upsample;                      // 4x4 images: 32 per patch, shared zero borders (HaloGeom::mos)
  if (!mosaic && (H < 8 || W < 8 || (H & (H - 1)) || (W & (W - 1)))) return false;     // anything else below 8x8 stays on igemm_pipe
  if ((long long)a.M >= (1LL << 31)) return false;
  const int bke = 64 / dc_dtype_size(dtype);
  if (a.C0 % bke || a.C1 % bke) return false;
  return true;
}

template <typename T, int NW>
static int launch_halo(const IgemmArgs& a0, int n_img, hipStream_t s, bool up4 = false) {
  using Cfg = HaloCfg<NW>;
  static bool attr_done_v[2] = {false, false};
  bool& attr_done = attr_done_v[up4 ? 1 : 0];
  void (*kern)(const IgemmArgs, const HaloGeom) = up4 ? conv3_halo_kernel<T, NW, 4> : conv3_halo_kernel<T, NW>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_halo_kernel<T, NW, 9, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_halo_kernel<T, NW, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    if constexpr (NW == 8) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_halo_kernel<T, NW, 9, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_halo_kernel<T, NW, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    }
    attr_done = true;
  }
  IgemmArgs a = a0;
  HaloGeom g;
  g.H = a.Hin; g.W = a.Win; g.n_img = n_img;
  const int tw = g.W < 32 ? g.W : 32;
  int th = Cfg::PIX / tw; if (th > g.H) th = g.H;
  const int ni = Cfg::PIX / (tw * th);
  g.ltw = ilog2(tw); g.lth = ilog2(th); g.lni = ilog2(ni);
  g.tiles_x = g.W / tw; g.tiles_y = g.H / th;
  g.hw = tw + 2; g.hp = (th + 2) * g.hw; g.HR = ni * g.hp;
  g.mos = 0; g.lmc = 0; g.inv_ch = g.inv_cw = 0.f;
  if (g.H < 8 || g.W < 8) {            // whole small images: mosaic with shared zero borders (only the 512-pixel patch takes them)
    if (NW != 8 || tw != g.W || th != g.H || ni < 2) { dc_set_error("conv3_halo: %dx%d images need the 8-wave patch", g.H, g.W); return DC_ERR_SHAPE; }
    g.mos = 1;
    g.lmc = (g.lni + 1) / 2;           // columns >= rows: 32 images -> 8 x 4
    const int cols = 1 << g.lmc, rows = ni >> g.lmc;
    g.hw = cols * (tw + 1) + 1;
    g.hp = 0;
    g.HR = (rows * (th + 1) + 1) * g.hw;
    g.inv_ch = 1.0f / (float)(th + 1); g.inv_cw = 1.0f / (float)(tw + 1);
  }
  g.inv_hp = g.hp ? 1.0f / (float)g.hp : 0.f; g.inv_hw = 1.0f / (float)g.hw;
  g.sws = (g.ltw < 4 ? g.ltw : 4) - 2;     // tw >= 16: 2, 8: 1, 4 (mosaic): 0
  g.lpt = ilog2(g.tiles_x * g.tiles_y);
  {
    const long long hws = (long long)(a.upsample ? (g.H >> 1) * (g.W >> 1) : g.H * g.W);
    const long long ldmax = a.ld0 > a.ld1 ? (a.ld0 > a.ld2 ? a.ld0 : a.ld2) : (a.ld1 > a.ld2 ? a.ld1 : a.ld2);
    g.xbuf = (ni == 1 && !g.mos && hws * ldmax * (long long)sizeof(T) < (1LL << 31) &&
              (long long)a.tiles_n * 128 * a.Ktot * (long long)sizeof(T) < (1LL << 31)) ? 1 : 0;
  }
  if ((long long)a.tiles_n * 128 * a.Ktot >= (1LL << 31)) { dc_set_error("conv3_halo: weight matrix of %d x %d too large", a.tiles_n * 128, a.Ktot); return DC_ERR_SHAPE; }
  g.nxl = (g.HR * 4 + Cfg::NT - 1) / Cfg::NT;
  if (g.HR > Cfg::XROWS || g.nxl > Cfg::NXL || g.nxl < 3) { dc_set_error("conv3_halo: halo of %d rows does not fit", g.HR); return DC_ERR_SHAPE; }
  a.tiles_m = ((n_img + ni - 1) / ni) * g.tiles_x * g.tiles_y;
  const long long nblk = (long long)a.tiles_m * a.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { dc_set_error("conv3_halo: bad grid %lld", nblk); return DC_ERR_SHAPE; }
  if (g.xbuf) kern = up4 ? conv3_halo_kernel<T, NW, 4, 1> : conv3_halo_kernel<T, NW, 9, 1>;
  bool pn_local = false;
  if (a.pn_out) {
    // producer-side GroupNorm: images that span workgroups -> the one-image-per-patch form with the exchange of epi_pn.h; 8x8 images ->
    // the staggered 8-wave kernel, every image inside one wave; dc_conv3_halo_pn_ok said so, this is the launch-time proof
    if constexpr (NW == 4) {
      if (!g.xbuf || ni != 1) { dc_set_error("conv3_halo: producer-side GroupNorm needs the one-image-per-patch form"); return DC_ERR_SHAPE; }
      static bool pn_attr[2] = {false, false};
      kern = up4 ? conv3_halo_kernel<T, 4, 4, 1, false, true> : conv3_halo_kernel<T, 4, 9, 1, false, true>;
      if (!pn_attr[up4 ? 1 : 0]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS); pn_attr[up4 ? 1 : 0] = true; }
    } else {
      const bool i8 = !g.mos && !g.xbuf && g.H == 8 && g.W == 8 && ni == 8, i4 = g.mos && g.H == 4 && g.W == 4 && ni == 32;
      if (up4 || !(i8 || i4)) { dc_set_error("conv3_halo: producer-side GroupNorm on the 8-wave patch needs 8x8 or 4x4 images"); return DC_ERR_SHAPE; }
      pn_local = true;
    }
  }
  if constexpr (NW == 8) {
    if (g.mos) kern = up4 ? conv3_halo_kernel<T, NW, 4, 2> : conv3_halo_kernel<T, NW, 9, 2>;
    // staggered wave groups (STG) unless DCAMD_HALO_NO_STAG (read per call: A/B runs in one process)
    // (the four-tap upsample form stays on the lock-step loop: its whole next halo would ride in one MFMA block — measured slower)
    if (!up4 && !getenv("DCAMD_HALO_NO_STAG") && (long long)a.tiles_n * 128 * a.Ktot * (long long)sizeof(T) < (1LL << 31)) {
      const int mode = pn_local ? (g.mos ? 4 : 3) : (g.mos ? 2 : (g.xbuf ? 1 : 0));
      static bool stg_attr[5] = {false, false, false, false, false};
      switch (mode) {
        case 4: kern = conv3_halo_kernel<T, NW, 9, 2, true, true>; break;
        case 3: kern = conv3_halo_kernel<T, NW, 9, 0, true, true>; break;
        case 0: kern = conv3_halo_kernel<T, NW, 9, 0, true>; break;
        case 1: kern = conv3_halo_kernel<T, NW, 9, 1, true>; break;
        default: kern = conv3_halo_kernel<T, NW, 9, 2, true>; break;
      }
      if (!stg_attr[mode]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
        stg_attr[mode] = true;
      }
    }
  }
  if constexpr (NW == 8) {
    if (pn_local && kern != conv3_halo_kernel<T, 8, 9, 0, true, true> && kern != conv3_halo_kernel<T, 8, 9, 2, true, true>) { dc_set_error("conv3_halo: producer-side GroupNorm on 8x8 images needs the staggered loop"); return DC_ERR_SHAPE; }
  }
  long long grid = nblk;
  if (a.pn_out && !pn_local) {          // whole groups of 2^lpt workgroups, a multiple of 8 of them (PN block order, see the kernel)
    const long long groups = (long long)n_img * (up4 ? a.tiles_n >> 2 : a.tiles_n);
    grid = ((groups + 7) / 8 * 8) << (g.lpt + (up4 ? 2 : 0));
    if (grid > 0x7fffffffLL) { dc_set_error("conv3_halo: bad grid %lld", grid); return DC_ERR_SHAPE; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::NT), Cfg::LDS, s, a, g);
  return dc_check_launch("dc_igemm(conv3_halo)");
}

// producer-side GroupNorm (epi_pn.h): the conv must be the 4-wave one-image-per-patch form (power-of-two images of 16x16 ... 64x64: at
// most 16 workgroups and 32 quad-record parts per sample), Cout a multiple of 128 whose GroupNorm groups are 4 ... 32 channels wide
// (a group never leaves a wave's 64 channels); output (raw and normalised) in the compute type
constexpr int PN_MAX_TILES = 16;
bool dc_conv3_halo_pn_ok(const IgemmArgs& a, int dtype, bool up4) {
  static const bool off = getenv("DCAMD_NO_PN") != nullptr;
  if (off || a.src1) return false;
  if (!up4 && ((a.Hin == 8 && a.Win == 8) || (a.Hin == 4 && a.Win == 4 && !getenv("DCAMD_NO_MOSAIC")))) {
    // 8x8 / 4x4 images: the staggered 8-wave kernel, every image inside one wave, the statistics never leave it (EpiPnLocal8x8 / 4x4)
    if (a.upsample || !dc_conv3_halo_applicable(a, dtype) || getenv("DCAMD_HALO_NO_STAG")) return false;
    if (a.Cout % 128 || a.pn_groups <= 0 || a.Cout % a.pn_groups) return false;
    const int cpg8 = a.Cout / a.pn_groups;
    if (cpg8 != 4 && cpg8 != 8 && cpg8 != 16 && cpg8 != 32) return false;
    return (long long)a.tiles_n * 128 * a.Ktot * dc_dtype_size(dtype) < (1LL << 31);
  }
  // four-phase upsample conv: the kernel walks the low-resolution image, and the four phases of every tile share the output sample
  if (up4 ? !(a.upsample && dc_conv3_up4_applicable(a, dtype)) : (a.upsample || !dc_conv3_halo_applicable(a, dtype))) return false;
  const int H = up4 ? a.Hin >> 1 : a.Hin, W = up4 ? a.Win >> 1 : a.Win;
  if (H < 16 || W < 16 || H * W < 256 || ((H * W) / 256) * (up4 ? 4 : 1) > PN_MAX_TILES) return false;
  if (a.Cout % 128 || a.pn_groups <= 0 || a.Cout % a.pn_groups) return false;
  const int cpg = a.Cout / a.pn_groups;
  if (cpg != 4 && cpg != 8 && cpg != 16 && cpg != 32) return false;
  const long long es = dc_dtype_size(dtype);
  const long long ldmax = a.ld0 > a.ld2 ? a.ld0 : a.ld2;
  if ((long long)H * W * ldmax * es >= (1LL << 31)) return false;                       // buffer-descriptor loaders (HaloGeom::xbuf)
  if ((long long)a.tiles_n * 128 * a.Ktot * es >= (1LL << 31)) return false;
  return true;
}

unsigned dc_conv3_halo_pn_timeouts() {
  unsigned v = 0, z = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_pn_timeouts), sizeof(v)) != hipSuccess) return ~0u;
  if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pn_timeouts), &z, sizeof(z));
  return v;
}

// "nearest-2x upsample, then 3x3 conv" as four 2x2-tap phases on the low-resolution image (a.W: the phase-summed weights
// [4][Cout_pad][4 * C], see dc_igemm_params.up4): 4/9 of the MACs
bool dc_conv3_up4_applicable(const IgemmArgs& a, int dtype) {
  if (a.taps != 9 || a.stride != 1 || !a.upsample || a.act != DC_ACT_NONE || a.gate || a.gn_scale || a.src2 || a.residual) return false;
  IgemmArgs lo = a;
  lo.upsample = 0; lo.Hin = a.Hin >> 1; lo.Win = a.Win >> 1;
  return dc_conv3_halo_applicable(lo, dtype);
}

int dc_conv3_up4_launch(const IgemmArgs& a0, int dtype, int n_img, hipStream_t s) {
  IgemmArgs a = a0;
  a.upsample = 0; a.Hin = a0.Hin >> 1; a.Win = a0.Win >> 1;      // the kernel walks the LOW-resolution image
  a.Ktot = 4 * (a.C0 + a.C1);
  a.tiles_n = 4 * a0.tiles_n;                                    // phase-major N tiles
  a.n_fast = 0;
  const int nw = (a.Hin <= 8 || a.Win <= 8) ? 8 : 4;
  if (nw == 8) {
    if (dtype == DC_BF16) return launch_halo<__bf16, 8>(a, n_img, s, true);
    if (dtype == DC_F16) return launch_halo<_Float16, 8>(a, n_img, s, true);
    return launch_halo<float, 8>(a, n_img, s, true);
  }
  if (dtype == DC_BF16) return launch_halo<__bf16, 4>(a, n_img, s, true);
  if (dtype == DC_F16) return launch_halo<_Float16, 4>(a, n_img, s, true);
  return launch_halo<float, 4>(a, n_img, s, true);
}

int dc_conv3_halo_launch(const IgemmArgs& a, int dtype, int n_img, hipStream_t s) {
  // 8x8 images: deep layers (Cout >= 256) are weight-traffic bound, so they take the 512-pixel patch (half the
  // weight bytes per pixel); the 256-pixel patch would also need 4 x 100 halo rows = 7 loads per lane
  const int nw = (a.Hin <= 8 || a.Win <= 8) ? 8 : 4;
  if (nw == 8) {
    if (dtype == DC_BF16) return launch_halo<__bf16, 8>(a, n_img, s);
    if (dtype == DC_F16) return launch_halo<_Float16, 8>(a, n_img, s);
    return launch_halo<float, 8>(a, n_img, s);
  }
  if (dtype == DC_BF16) return launch_halo<__bf16, 4>(a, n_img, s);
  if (dtype == DC_F16) return launch_halo<_Float16, 4>(a, n_img, s);
  return launch_halo<float, 4>(a, n_img, s);
}

template <typename T>
static int launch_thin(const IgemmArgs& a0, int n_img, hipStream_t s) {
  static bool attr_done = false;
  const bool gn = a0.gn_scale != nullptr;
  void (*kern)(const IgemmArgs, const HaloGeom) = gn ? conv3_thin_kernel<T, true> : conv3_thin_kernel<T, false>;
  const int lds = gn ? ThinCfg::LDS_GN : ThinCfg::LDS;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_thin_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ThinCfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_thin_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ThinCfg::LDS_GN);
    attr_done = true;
  }
  IgemmArgs a = a0;
  HaloGeom g;
  g.H = a.Hin; g.W = a.Win; g.n_img = n_img;
  const int tw = g.W < 32 ? g.W : 32;
  int th = 256 / tw; if (th > g.H) th = g.H;
  const int ni = 256 / (tw * th);
  g.ltw = ilog2(tw); g.lth = ilog2(th); g.lni = ilog2(ni);
  g.tiles_x = g.W / tw; g.tiles_y = g.H / th;
  g.hw = tw + 2; g.hp = (th + 2) * g.hw; g.HR = ni * g.hp;
  g.mos = 0; g.lmc = 0; g.inv_ch = g.inv_cw = 0.f; g.xbuf = 0; g.sws = 0;     // (the thin kernel keeps its own un-swizzled image)
  g.inv_hp = 1.0f / (float)g.hp; g.inv_hw = 1.0f / (float)g.hw;
  g.nxl = (g.HR * 4 + ThinCfg::NT - 1) / ThinCfg::NT;
  if (g.nxl > ThinCfg::NXL) { dc_set_error("conv3_thin: halo of %d rows does not fit", g.HR); return DC_ERR_SHAPE; }
  const long long nblk = (long long)((n_img + ni - 1) / ni) * g.tiles_x * g.tiles_y;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { dc_set_error("conv3_thin: bad grid %lld", nblk); return DC_ERR_SHAPE; }
  if (gn && ni != 1) { dc_set_error("conv3_thin: the fused GroupNorm needs one sample per patch"); return DC_ERR_SHAPE; }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(ThinCfg::NT), lds, s, a, g);
  return dc_check_launch("dc_igemm(conv3_thin)");
}

int dc_conv3_thin_launch(const IgemmArgs& a, int dtype, int n_img, hipStream_t s) {
  if (dtype == DC_BF16) return launch_thin<__bf16>(a, n_img, s);
  if (dtype == DC_F16) return launch_thin<_Float16>(a, n_img, s);
  return launch_thin<float>(a, n_img, s);
}
