// conv3_ws.hip — wave-specialised halo-tile 3x3 convolution, with the GroupNorm(+SiLU) of its input applied on the fly.
//
// What it replaces: the pair "dc_groupnorm (normalise + SiLU, a pure read + write pass over the tensor) -> conv3_halo" at the
// GroupNorm -> Conv2d 3x3 sites of diffusers' ResnetBlock2D (behind reference nets/unet.py:186-195): 13-16 % of a UNet scoring
// step was that extra pass.  The affine a[n][c] = rstd * gamma, b[n][c] = beta - mean * a comes from the PRODUCER's quad records
// (dc_groupnorm statistics-only mode, gn_qaffine_kernel: no pass over the tensor at all); this kernel reads the RAW tensor and the
// normalised tensor never exists in HBM.
//
// Why wave-specialised: putting y = silu(x a + b) into the MFMA waves' own instruction stream was built twice (conv3_halo's GN
// variant) and lost — the in-order wave stalls its MFMAs behind the LDS round trip and ~300 issue cycles of VALU per 16-byte
// piece (conv +42 %).  Here a 512-thread workgroup is two teams of four waves, one wave of each per SIMD:
//   waves 4-7  LOADERS: issue every LDS-DMA of the tile (halo chunks through buffer descriptors, W[tap] tiles), wait for them with
//              counted vmcnt, and transform each landed halo chunk IN PLACE (one 16-byte piece per lane and tap, the chunk after
//              the one being multiplied), skipping padding pieces (the reference pads the NORMALISED tensor with zeros);
//   waves 0-3  MFMA waves: per tap 12 fragment reads + 32 MFMAs (wave tile 128 pixels x 64 couts), nothing else; then the
//              lane-resident epilogue (bias + row vector from LDS, residual, quad statistics of the output).
// The teams meet at ONE s_barrier per tap (the W ring's hand-over), exactly the barrier the 4-wave kernel has.  Separate code paths
// per team: the loaders never hold an accumulator, the MFMA waves never hold a piece offset (the kernel's register count is the
// larger of the two, not their sum).
//
// Same tile, LDS image, tap order, accumulation order and epilogue as conv3_halo_kernel<T, 4, ., 9, 1>: results are bit-identical
// to "dc_groupnorm + conv3_halo" (the transform is the GroupNorm kernels' own expression, rounded to T where they round).
// One workgroup per CU (8 waves at <= 256 registers); 78 KiB of LDS.
#include "common.h"
#ifdef DC_STAMPS
static __device__ unsigned long long* g_ws_stamps;
extern "C" void dc_debug_set_ws_stamps(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stamps), &p, sizeof(p)); }
#define DC_STAMP(k) do { if ((threadIdx.x & 255) == 0 && g_ws_stamps) g_ws_stamps[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
// cycles spent inside a wait (barrier / vmcnt), summed per team over one tile into slot k: WAIT_T0 before, WAIT_ADD(sum) after
#define DC_WAIT_T0() const unsigned long long wt0_ = __builtin_amdgcn_s_memtime()
#define DC_WAIT_ADD(sum) do { sum += __builtin_amdgcn_s_memtime() - wt0_; } while (0)
#define DC_STAMP_VAL(k, v) do { if ((threadIdx.x & 255) == 0 && g_ws_stamps) g_ws_stamps[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 8 + (k)] = (v); } while (0)
// timing-only ablations of the one-tile kernel (results are wrong on purpose): 1 no MFMAs, 2 W LDS-DMA through a zero-record
// descriptor (the range check drops the fetch, the instruction stays), 4 the same for the halo chunks, 8 no transform, 16 no LDS-DMA
// instructions at all, 32 no fragment reads (and no MFMAs); conv3_wr_kernel: 8, and 64 = W fragment loads shaped as a fragment-major
// weight image would make them (1 KiB contiguous per instruction)
static __device__ int g_ws_abl;
extern "C" void dc_debug_set_ws_abl(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ws_abl), &v, sizeof(v)); }
#define DC_WS_ABL() __builtin_amdgcn_readfirstlane(g_ws_abl)
#else
#define DC_WS_ABL() 0
#define DC_STAMP(k) do {} while (0)
#define DC_WAIT_T0() do {} while (0)
#define DC_WAIT_ADD(sum) do {} while (0)
#define DC_STAMP_VAL(k, v) do {} while (0)
#endif
#include "conv3_halo.h"
DC_CLOCK_DECL(conv3_ws)

static __device__ chunk16 g_ws_zero_page[4];      // source of the table pieces that have nothing to fetch

struct WsCfg {
  static constexpr int NT = 512, NTL = 256;                 // threads; threads per team
  static constexpr int NXL = 6;                             // LDS-DMA pieces per loader lane per halo chunk (<= 384 halo rows)
  static constexpr int XBUF = NXL * NTL * 16;               // 24 KiB per halo buffer
  static constexpr int WLD = 512 / NTL;                     // W LDS-DMA pieces per loader lane per tap (8 KiB tile)
#ifdef DC_WS_WR
  static constexpr int WR = DC_WS_WR;                       // diagnostic builds: deeper W ring (plain variant only: the transform schedule assumes 3)
#else
  static constexpr int WR = 3;                              // W ring slots (prefetch distance 2 taps)
#endif
  static constexpr int GNOFF = 2 * XBUF + WR * HALO_WST;    // GroupNorm affine of the workgroup's sample: scale[C], shift[C]
  static constexpr int GNMAXC = 512;
  static constexpr int BRVOFF = GNOFF + 2 * GNMAXC * 4;     // bias + row vector of the N tile (128 floats)
  static constexpr int LDS = BRVOFF + 128 * 4;
};

template <typename T, bool GN>
__global__ __launch_bounds__(512, 2) void conv3_ws_kernel(const IgemmArgs a, const HaloGeom g) {
  using Cfg = WsCfg;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 4 * EPC;                  // channels per 64-byte chunk row
  constexpr int TM = 8, TN = 4, NTAP = 9;
  constexpr int NTL = Cfg::NTL, NXL = Cfg::NXL, WLD = Cfg::WLD, WR = Cfg::WR, PD = WR - 1;
  constexpr int FLY = (PD - 1) * WLD;           // W(s+1) .. W(s+PD-1)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Wring = smem + 2 * Cfg::XBUF;
  float* const gnp = reinterpret_cast<float*>(smem + Cfg::GNOFF);
  float* const brv = reinterpret_cast<float*>(smem + Cfg::BRVOFF);

  DC_STAMP(0);
#ifdef DC_STAMPS   // workgroup turnover on a CU (tools/dev/ws_turnover.py): wall clock (100 MHz) at start / end and where the workgroup ran
  if (threadIdx.x == 256) {
    DC_STAMP_VAL(3, __builtin_amdgcn_s_memrealtime());
    DC_STAMP_VAL(4, (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32));   // HW_ID, XCC_ID
  }
#endif
  const int abl = DC_WS_ABL();                  // 0 outside diagnostic builds
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave >= 4;                // wave-uniform (scalar branch)
  const int cw = wave & 3, tl = t & 255;        // wave / thread inside the team
  int tile_m, tile_n;
  tile_of_block_scalar(a, tile_m, tile_n);
  const int tx = tile_m % g.tiles_x;
  const int ty = (tile_m / g.tiles_x) % g.tiles_y;
  const int ng = tile_m / (g.tiles_x * g.tiles_y);      // one image per patch: the workgroup's sample
  const int tw = 1 << g.ltw, th = 1 << g.lth;
  const int HW = g.H * g.W;
  const int Ctot = a.C0 + a.C1;
  const int c0chunks = a.C0 / BKE, nchunks = Ctot / BKE;
  const int nx = a.src2 ? a.C2 / BKE : 0;      // 32-channel chunks of the 1x1 side source (>= 2 when present)

  // ---- tables of the workgroup's sample, by everybody: bias + row vector of the N tile, GroupNorm affine ----
  if (t < 128) {
    const int c = tile_n * 128 + t;
    float v = 0.f;
    if (c < a.Cout) {
      if (a.bias) v = a.bias[c];
      if (a.rowvec) v += a.rowvec[(size_t)(a.rowvec_map ? a.rowvec_map[ng] : ng) * a.rowvec_ld + c];
    }
    brv[t] = v;
  }
  if (GN) {
    for (int c = t; c < Ctot; c += Cfg::NT) {
      gnp[c] = a.gn_scale[(size_t)ng * Ctot + c];
      gnp[Ctot + c] = a.gn_shift[(size_t)ng * Ctot + c];
    }
  }
  __syncthreads();

  if (loader) {
    // =============================================== LOADER TEAM ===============================================
    // piece i of lane tl: LDS position i * 256 + tl -> halo row >> 2, 16-byte chunk & 3 (the image is not swizzled)
    int pp[NXL];                                // pixel offset inside the sample, -1 = padding
    int hsw = 0;                                // chunk swizzle (HaloGeom::sws): bit i = this lane's slot of piece i holds logical chunk xlx ^ 2
    const int xlx = tl & 3;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int hr = (i * NTL + tl) >> 2;
      pp[i] = -1;
      if (i < g.nxl && hr < g.HR) {
        // hr < 2^12 and (hr + 0.5) / hw is at least 0.5 / hw away from an integer: the fp32 product floors exactly
        const int hy = (int)(((float)hr + 0.5f) * g.inv_hw), hx = hr - hy * g.hw;
        const int iy = ty * th + hy - 1, ix = tx * tw + hx - 1;
        hsw |= ((hx >> g.sws) & 1) << i;
        if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) pp[i] = iy * g.W + ix;
      }
    }
    // buffer descriptors (wave-uniform by construction: kernel arguments and blockIdx only); a padding piece gets offset
    // 0xffffffff (out of range: the hardware returns zeros)
    int ldb0 = a.ld0 * (int)sizeof(T), ldb1 = a.ld1 * (int)sizeof(T), ldb2 = a.ld2 * (int)sizeof(T);
    asm volatile("" : "+s"(ldb0), "+s"(ldb1), "+s"(ldb2));
    const int s0 = __builtin_amdgcn_readfirstlane(a.map0 ? a.map0[ng] : ng);
    const T* xb0 = reinterpret_cast<const T*>(a.src0) + (size_t)s0 * HW * a.ld0;
    const T* xb1 = nullptr; const T* xb2 = nullptr;
    if (a.src1) { const int s1 = __builtin_amdgcn_readfirstlane(a.map1 ? a.map1[ng] : ng); xb1 = reinterpret_cast<const T*>(a.src1) + (size_t)s1 * HW * a.ld1; }
    if (a.src2) { const int s2 = __builtin_amdgcn_readfirstlane(a.map2 ? a.map2[ng] : ng); xb2 = reinterpret_cast<const T*>(a.src2) + (size_t)s2 * HW * a.ld2; }
    const int wrow0 = tl >> 2;                  // LDS row of the lane's first W piece (piece i is 64 rows further)
    const int wvoff = (epi_wrow(wrow0, false) * a.Ktot + ((tl & 3) ^ swz64(wrow0)) * EPC) * (int)sizeof(T);
    const int wtile0 = tile_n * 128 * a.Ktot;   // element offset of the N tile; < 2^30 (host check)
    auto rsrc_of = [](const void* base, int rec = 0x7fffffff) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, rec, 0x00020000); };

    auto issue_x = [&](int cc) {                // cc >= nchunks: chunk cc - nchunks of the 1x1 side source
      int ldb = ldb0, cb = cc;
      const T* xb = xb0;
      if (cc >= nchunks) { ldb = ldb2; cb = cc - nchunks; xb = xb2; }
      else if (cc >= c0chunks) { ldb = ldb1; cb = cc - c0chunks; xb = xb1; }
      const int cofs = cb * 64 + xlx * 16;
      const __amdgpu_buffer_rsrc_t rs = rsrc_of(xb, (abl & 4) ? 0 : 0x7fffffff);
      if (abl & 16) return;
      char* xs = smem + (cc & 1) * Cfg::XBUF + cw * 1024;
#pragma unroll
      for (int i = 0; i < NXL; ++i) {
        // always NXL instructions (pieces past the halo are out of range: zeros into the unused tail): the counted waits stay constants
        const int pk = pp[i];
        const int voff = pk < 0 ? -1 : pk * ldb + (cofs ^ (((hsw >> i) & 1) << 5));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(xs + i * (NTL * 16)), 16, voff, 0, 0, 0);
      }
    };
    auto issue_w = [&](int cc, int tap, int slot) {
      const int so = (wtile0 + tap * Ctot + cc * BKE) * (int)sizeof(T);         // wave-uniform: scalar offset
      const __amdgpu_buffer_rsrc_t wrs = rsrc_of(a.W, (abl & 2) ? 0 : 0x7fffffff);
      if (abl & 16) return;
#pragma unroll
      for (int i = 0; i < WLD; ++i)             // piece i: LDS rows 64 i + (tl >> 2) = packed rows 64 further
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr_t)(Wring + slot * HALO_WST + i * (NTL * 16) + cw * 1024), 16, wvoff,
                                                 __builtin_amdgcn_readfirstlane(so + i * 64 * a.Ktot * (int)sizeof(T)), 0, 0);
    };
    auto issue_w2 = [&](int e, int slot) {      // side source weights W2 [Cout_pad][C2]: chunk e, same LDS tile image
#pragma unroll
      for (int i = 0; i < WLD; ++i) {
        const int row = (i * NTL + tl) >> 2;
        const T* wp = reinterpret_cast<const T*>(a.W2) + (size_t)(tile_n * 128 + epi_wrow(row, false)) * a.C2 + ((tl & 3) ^ swz64(row)) * EPC + e * BKE;
        __builtin_amdgcn_global_load_lds((gptr_t) reinterpret_cast<const char*>(wp),
                                         (lptr_t)(Wring + slot * HALO_WST + i * (NTL * 16) + cw * 1024), 16, 0, 0);
      }
    };
    // y = act(x * scale[c] + shift[c]) in place on the lane's own piece i of chunk ccx (the GroupNorm kernels' expression,
    // norms.hip, rounded to T as they round); padding pieces stay zero
    float scr[EPC], shr[EPC];
    auto load_affine = [&](int ccx) {
      const float* sc = gnp + ccx * BKE + xlx * EPC;
#pragma unroll
      for (int e = 0; e < EPC; ++e) { scr[e] = sc[e]; shr[e] = sc[Ctot + e]; }
    };
    auto xform = [&](int ccx, int i) {
      if (pp[i] >= 0 && !(abl & 8)) {
        // the lane transforms the slot that holds logical chunk xlx of its row (its own, or its neighbour's two lanes over — same wave,
        // same row, so the same vmcnt wait and the same padding test cover it): the affine registers stay one chunk's
        chunk16* q = reinterpret_cast<chunk16*>(smem + (ccx & 1) * Cfg::XBUF + (((i * NTL + tl) * 16) ^ (((hsw >> i) & 1) << 5)));
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
    };

    DC_STAMP(1);
    issue_x(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) issue_w(0, i, i);
    if (GN) {                                   // chunk 0: transformed before the first tap
      hwait_vmcnt<PD * WLD>();                  // own X(0) pieces have landed (the W groups may stay in flight)
      load_affine(0);
#pragma unroll
      for (int i = 0; i < NXL; ++i) xform(0, i);
    }
    for (int cc = 0; cc < nchunks; ++cc) {
      // "has_next": another X chunk and more W groups follow — the next 3x3 chunk, or the first chunk of the 1x1 side source
      const bool side_next = cc + 1 == nchunks && nx > 0;
      const bool has_next = cc + 1 < nchunks || side_next;
      const bool gn_next = GN && cc + 1 < nchunks;         // the side source is multiplied raw
      const int s0c = cc * NTAP;
      auto step = [&](auto tapc) {
        constexpr int tap = decltype(tapc)::value;
        // W(s) (and X(cc) when tap == 0) must have landed.  Younger groups that may stay in flight: W(s+1) .. W(s+PD-1) and, for tap
        // in 1..PD, the NXL pieces of X(cc+1) issued at tap 0; the last chunk has fewer W groups left.
        if (has_next) {
          if (tap >= 1 && tap <= PD) hwait_vmcnt<FLY + NXL>();
          else hwait_vmcnt<FLY>();
        } else {
          constexpr int left = NTAP - 1 - tap;
          hwait_vmcnt<(left < PD - 1 ? left : PD - 1) * WLD>();
        }
        if (GN && tap == 0) __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): my in-place writes of this chunk are in LDS
#ifdef DC_WS_ROWBAR_ABL   // diagnostic builds, timing only (the W ring is then racy): one barrier per tap ROW instead of one per tap
        if (tap % 3 == 0)
#endif
        __builtin_amdgcn_s_barrier();
        constexpr int t2 = tap + PD;              // the W group to issue now: s + PD
        if (t2 < NTAP) issue_w(cc, t2, (s0c + t2) % WR);
        else if (side_next) { if (t2 - NTAP < nx) issue_w2(t2 - NTAP, (s0c + t2) % WR); }
        else if (has_next) issue_w(cc + 1, t2 - NTAP, (s0c + t2) % WR);
        if (tap == 0 && has_next) issue_x(cc + 1);
        // X(cc+1) was issued at tap 0 and waited for at tap PD+1: from then on one piece per tap, while the other team multiplies
        // this chunk (they read the other buffer; they read this one only after the next chunk's first barrier)
        if (tap == PD + 1 && gn_next) load_affine(cc + 1);
        if (tap > PD && gn_next) xform(cc + 1, tap > PD ? tap - PD - 1 : 0);
      };
      step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{}); step(IC<8>{});
    }
    // 1x1 side source (a ResNet's conv_shortcut folded into its conv2): nx steps of the centre tap
    {
      const int NSm = nchunks * NTAP;
      for (int e = 0; e < nx; ++e) {
        if (e == 0 && nx >= PD) hwait_vmcnt<FLY>();      // W2(1 .. PD-1) may stay in flight; X2(0) landed long ago
        else hwait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (e + PD < nx) issue_w2(e + PD, (NSm + e + PD) % WR);
        if (e + 1 < nx) issue_x(nchunks + e + 1);
      }
    }
    DC_STAMP(2);
    return;
  }

  // ================================================= MFMA TEAM =================================================
  const int wm = cw >> 1, wn = cw & 1;          // 2 waves along pixels, 2 along couts
  const int lr = lane & 15, lq = lane >> 4;
  // fragment read addresses: per-lane part + wave-uniform part per fragment (SGPRs); pixel p = wm*128 + j*16 + lr
  int xlv[3];                                   // per-lane part per column offset of the tap (chunk swizzle: conv3_halo.h, HaloGeom::sws)
  {
    const int xx = lr & ((tw < 16 ? tw : 16) - 1);
    const int rowpart = ((lr >> g.ltw) * g.hw + (lr & (tw - 1))) * 64;
#pragma unroll
    for (int k = 0; k < 3; ++k) xlv[k] = rowpart + ((lq ^ ((((xx + k) >> g.sws) & 1) << 1)) << 4);
  }
  int joff[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int p = wm * 128 + j * 16;
    const int py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
    joff[j] = __builtin_amdgcn_readfirstlane((py * g.hw + px) * 64);
  }
  const int woff0 = lds64_off(wn * 64 + lr, lq);
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // one tap of one channel chunk: W fragments, then the X fragments in two halves; MFMAs in j-major order (the 4-wave kernel's order)
  auto mma_tap = [&](const char* Wst, const char* Xb, int tapoff, auto kxc) {
    constexpr int kx = decltype(kxc)::value;
    if (abl & 32) return;
    chunk16 wf[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const chunk16*>(Wst + woff0 + i * 1024);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      chunk16 xf[TM / 2];
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) xf[j] = *reinterpret_cast<const chunk16*>(Xb + (tapoff + joff[h * (TM / 2) + j]) + xlv[kx]);
#ifdef DC_STAMPS
      if (abl & 1) {                              // timing only: the fragments are read, nothing is multiplied
#pragma unroll
        for (int j = 0; j < TM / 2; ++j) asm volatile("" ::"v"(xf[j]));
#pragma unroll
        for (int i = 0; i < TN; ++i) asm volatile("" ::"v"(wf[i]));
        continue;
      }
#endif
#pragma unroll
      for (int j = 0; j < TM / 2; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][h * (TM / 2) + j] = Mma<T>::run(wf[i], xf[j], acc[i][h * (TM / 2) + j]);
      if (h == 0) __builtin_amdgcn_sched_barrier(0);
    }
  };
  DC_STAMP(1);
  DC_CLOCK(0);
  for (int cc = 0; cc < nchunks; ++cc) {
    const int s0c = cc * NTAP;
    const char* Xb = smem + (cc & 1) * Cfg::XBUF;
    auto step = [&](auto tapc) {
      constexpr int tap = decltype(tapc)::value;
#ifdef DC_WS_ROWBAR_ABL
      if (tap % 3 == 0)
#endif
      __builtin_amdgcn_s_barrier();
      const char* Wst = Wring + ((s0c + tap) % WR) * HALO_WST;
      constexpr int ky = tap / 3, kx = tap - ky * 3;
      mma_tap(Wst, Xb, (ky * g.hw + kx) * 64, IC<kx>{});
    };
    step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{}); step(IC<8>{});
  }
  DC_CLOCK(1);
  {
    const int NSm = nchunks * NTAP;
    for (int e = 0; e < nx; ++e) {
      __builtin_amdgcn_s_barrier();
      mma_tap(Wring + ((NSm + e) % WR) * HALO_WST, smem + ((nchunks + e) & 1) * Cfg::XBUF, (g.hw + 1) * 64, IC<1>{});     // centre tap
    }
  }
  DC_STAMP(2);
  // ---- epilogue: straight from the accumulators (igemm_epilogue.h: the weight rows were loaded permuted) ----
  HaloQs qsfn;
  qsfn.nbase = ng; qsfn.ltp = g.ltw + g.lth; qsfn.n_img = g.n_img; qsfn.tile_in_img = ty * g.tiles_x + tx; qsfn.wm = wm;
  qsfn.np = HW >= 128 ? HW >> 7 : 1;
  qsfn.padd = 0;
  auto rowfn = [&](int j, EpiRow& r) {
    const int p = wm * 128 + j * 16 + lr;
    const int py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
    const int rem = (ty * th + py) * g.W + tx * tw + px;
    r.ok = true;
    r.samp = ng;
    r.o = ng * HW + rem;
    r.r = (a.residual && a.res_map ? a.res_map[ng] : ng) * HW + rem;
  };
  epi_direct_act<T, TM, DC_ACT_NONE, false, true>(a, acc, tile_n, wn, lq, ng, ng, rowfn, EpiNoPre(), qsfn, HaloLdsBias{brv + wn * 64 + lq * 8});
  DC_STAMP(7);
#ifdef DC_STAMPS
  if (threadIdx.x == 0) DC_STAMP_VAL(6, __builtin_amdgcn_s_memrealtime());
#endif
}

// ====================================================================================================================
// conv3_wsp_kernel — the same two teams as PERSISTENT workgroups with a software-pipelined MFMA team.
//
// What the one-tile kernel above loses (s_memtime stamps, tools/stamp_ws.py): with one workgroup per CU nothing fills (a) the
// tile's prologue — the loaders' first halo chunk at HBM latency plus its transform, during which the MFMA team idles —, (b) the
// fragment-read latency and the barrier at the head of every tap (in the 4-wave kernel the SIMD's other wave, from the CU's other
// workgroup, fills them).  Here:
//   * a workgroup walks a list of tiles (one workgroup per CU; each XCD label takes a contiguous range of the tile list, its
//     workgroups stride through it).  Right after a tile's last barrier the loaders fetch, transform and publish the NEXT tile's
//     first chunk and W tiles while the MFMA team runs the epilogue: the next tile's prologue disappears behind the epilogue.  The
//     per-tile tables (GroupNorm affine of the sample, bias and row vector of the N tile) arrive by LDS-DMA one tile ahead, into
//     slots alternating with the tile parity.  W ring slots and halo buffers continue to rotate across tiles.
//   * the MFMA team's step is { second-half X reads ; 16 MFMAs ; barrier of the NEXT step ; next step's W and first-half X reads ;
//     16 MFMAs }: every fragment read and the barrier sit under 16 MFMAs of the same wave (W fragments ping-pong between two
//     register sets by step parity; a chunk has 9 steps, so the chunk body exists for both start parities).
// Same LDS image, tap order and accumulation order: bit-identical to the one-tile kernel.
struct WspCfg {
#ifdef DC_WS_WR
  static constexpr int WR = DC_WS_WR;
#else
  static constexpr int WR = 3;
#endif
  static constexpr int NT = 512, NTL = 256, NXL = 6, WLD = 2, TBLN = 2;
  static constexpr int XBUF = NXL * NTL * 16;
  static constexpr int GNOFF = 2 * XBUF + WR * HALO_WST;    // [2 tile parities][scale[C] | shift[C]] (4 KiB each)
  static constexpr int GNMAXC = 512;
  static constexpr int BRVOFF = GNOFF + 2 * 2 * GNMAXC * 4; // [2 tile parities][bias[128] | rowvec[128]] (1 KiB each)
  static constexpr int DUMPOFF = BRVOFF + 2 * 256 * 4;      // where placeholder table pieces land (4 + 4 KiB)
  static constexpr int LDS = DUMPOFF + 8 * 1024;
};

struct WsLdsBias2 {                                         // bias + row vector of the N tile from their LDS tables (run k: 32 k channels on)
  static constexpr bool on = true, has_rowvec = true;
  const float* p;                                           // bias entries; the row vector's sit 128 floats further
  __device__ __forceinline__ void operator()(int k, float (&bs)[8], float (&)[8]) const {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p + 32 * k), hi = *reinterpret_cast<const f32x4*>(p + 32 * k + 4);
    const f32x4 rl = *reinterpret_cast<const f32x4*>(p + 128 + 32 * k), rh = *reinterpret_cast<const f32x4*>(p + 128 + 32 * k + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bs[e] = lo[e] + rl[e]; bs[4 + e] = hi[e] + rh[e]; }
  }
};

template <typename T, bool GN>
__global__ __launch_bounds__(512, 2) void conv3_wsp_kernel(const IgemmArgs a, const HaloGeom g, const int total_tiles) {
  using Cfg = WspCfg;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 4 * EPC;
  constexpr int TM = 8, TN = 4, NTAP = 9;
  constexpr int NTL = Cfg::NTL, NXL = Cfg::NXL, WLD = Cfg::WLD, WR = Cfg::WR, PD = WR - 1, TBLN = Cfg::TBLN;
  constexpr int FLY = (PD - 1) * WLD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Wring = smem + 2 * Cfg::XBUF;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave >= 4;
  const int cw = wave & 3, tl = t & 255;
  const int tw = 1 << g.ltw, th = 1 << g.lth;
  const int HW = g.H * g.W;
  const int Ctot = a.C0 + a.C1;
  const int c0chunks = a.C0 / BKE, nchunks = Ctot / BKE;
  const int nx = a.src2 ? a.C2 / BKE : 0;
  const int tiles_img = g.tiles_x * g.tiles_y;
  // the workgroup's tiles: XCD label (blockIdx & 7) -> a contiguous range of the tile list, strided over the label's workgroups
  const int xcd = blockIdx.x & 7, q8 = total_tiles >> 3, r8 = total_tiles & 7;
  const int lid0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
  const int lid_end = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + q8 + (xcd < r8 ? 1 : 0);
  const int lid_step = ((int)gridDim.x - xcd + 7) >> 3;
  auto tile_of = [&](int lid, int& tile_m, int& tile_n) __attribute__((always_inline)) {
    if (a.n_fast) { tile_m = lid / a.tiles_n; tile_n = lid - tile_m * a.tiles_n; }
    else { tile_n = lid / a.tiles_m; tile_m = lid - tile_n * a.tiles_m; }
    tile_m = __builtin_amdgcn_readfirstlane(tile_m);     // (the divisions run on the vector unit: see tile_of_block_scalar)
    tile_n = __builtin_amdgcn_readfirstlane(tile_n);
  };

  if (loader) {
    // =============================================== LOADER TEAM ===============================================
    const int xlx = tl & 3;
    const int wrow0 = tl >> 2;
    const int wvoff = (epi_wrow(wrow0, false) * a.Ktot + ((tl & 3) ^ swz64(wrow0)) * EPC) * (int)sizeof(T);
    int ldb0 = a.ld0 * (int)sizeof(T), ldb1 = a.ld1 * (int)sizeof(T), ldb2 = a.ld2 * (int)sizeof(T);
    asm volatile("" : "+s"(ldb0), "+s"(ldb1), "+s"(ldb2));
    auto rsrc_of = [](const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000); };
    // per-tile tables by LDS-DMA, TBLN = 2 pieces per lane (absent rows / lanes past the table: offset 0xffffffff -> zeros):
    //   piece 0: floats [256 cw, 256 cw + 256) of [scale[C] | shift[C]] of the tile's sample;  piece 1 (wave 0): bias | row vector
    auto issue_tables = [&](int lid, int par, bool real) __attribute__((always_inline)) {
      // real = false: placeholders (zeros into a dump slot) that only keep the counted waits of every chunk the same
      int tile_m, tile_n;
      tile_of(real ? lid : lid0, tile_m, tile_n);
      const int ng = tile_m / tiles_img;
      const char* zero = reinterpret_cast<const char*>(g_ws_zero_page);
      {   // 64-bit per-lane source addresses (the two halves of the table come from two tensors): global_load_lds, not a descriptor
        const int idx = cw * 256 + 4 * lane;
        const char* src = zero;
        if (GN && real && idx < Ctot) src = reinterpret_cast<const char*>(a.gn_scale + (size_t)ng * Ctot + idx);
        else if (GN && real && idx < 2 * Ctot) src = reinterpret_cast<const char*>(a.gn_shift + (size_t)ng * Ctot + (idx - Ctot));
        char* dst = real ? smem + Cfg::GNOFF + par * (2 * Cfg::GNMAXC * 4) + cw * 1024 : smem + Cfg::DUMPOFF + cw * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      }
      {   // wave 0: bias | row vector of the N tile; the other waves' piece is always a placeholder
        const int c = tile_n * 128 + 4 * (lane & 31);
        const int rrow = (real && a.rowvec) ? __builtin_amdgcn_readfirstlane(a.rowvec_map ? a.rowvec_map[ng] : ng) : 0;
        const char* src = zero;
        if (real && cw == 0 && c < a.Cout) {
          if (lane < 32) { if (a.bias) src = reinterpret_cast<const char*>(a.bias + c); }
          else if (a.rowvec) src = reinterpret_cast<const char*>(a.rowvec + (size_t)rrow * a.rowvec_ld + c);
        }
        char* dst = (real && cw == 0) ? smem + Cfg::BRVOFF + par * 1024 : smem + Cfg::DUMPOFF + 4096 + cw * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      }
    };
    // first tile's tables: fetched, landed and published before anything reads them (the one extra barrier of the kernel)
    issue_tables(lid0, 0, lid0 < lid_end);
    hwait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    int sb = 0, xpar = 0, tc = 0;                    // W ring phase, halo buffer parity, tile counter (all continue across tiles)
    for (int lid = lid0; lid < lid_end; lid += lid_step, ++tc) {
      int tile_m, tile_n;
      tile_of(lid, tile_m, tile_n);
      const int tx = tile_m % g.tiles_x, ty = (tile_m / g.tiles_x) % g.tiles_y, ng = tile_m / tiles_img;
      const int par = tc & 1;
      const float* gnp = reinterpret_cast<const float*>(smem + Cfg::GNOFF + par * (2 * Cfg::GNMAXC * 4));
      int pp[NXL];
#pragma unroll
      for (int i = 0; i < NXL; ++i) {
        const int hr = (i * NTL + tl) >> 2;
        pp[i] = -1;
        if (i < g.nxl && hr < g.HR) {
          const int hy = (int)(((float)hr + 0.5f) * g.inv_hw), hx = hr - hy * g.hw;
          const int iy = ty * th + hy - 1, ix = tx * tw + hx - 1;
          if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) pp[i] = iy * g.W + ix;
        }
      }
      const int s0 = __builtin_amdgcn_readfirstlane(a.map0 ? a.map0[ng] : ng);
      const T* xb0 = reinterpret_cast<const T*>(a.src0) + (size_t)s0 * HW * a.ld0;
      const T* xb1 = nullptr; const T* xb2 = nullptr;
      if (a.src1) { const int s1 = __builtin_amdgcn_readfirstlane(a.map1 ? a.map1[ng] : ng); xb1 = reinterpret_cast<const T*>(a.src1) + (size_t)s1 * HW * a.ld1; }
      if (a.src2) { const int s2 = __builtin_amdgcn_readfirstlane(a.map2 ? a.map2[ng] : ng); xb2 = reinterpret_cast<const T*>(a.src2) + (size_t)s2 * HW * a.ld2; }
      const int wtile0 = tile_n * 128 * a.Ktot;

      auto issue_x = [&](int cc) __attribute__((always_inline)) {                // cc >= nchunks: chunk cc - nchunks of the 1x1 side source
        int ldb = ldb0, cb = cc;
        const T* xb = xb0;
        if (cc >= nchunks) { ldb = ldb2; cb = cc - nchunks; xb = xb2; }
        else if (cc >= c0chunks) { ldb = ldb1; cb = cc - c0chunks; xb = xb1; }
        const int cofs = cb * 64 + xlx * 16;
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(xb);
        char* xs = smem + ((xpar + cc) & 1) * Cfg::XBUF + cw * 1024;
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
          const int pk = pp[i];
          const int voff = pk < 0 ? -1 : pk * ldb + cofs;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(xs + i * (NTL * 16)), 16, voff, 0, 0, 0);
        }
      };
      auto issue_w = [&](int cc, int tap, int slot) __attribute__((always_inline)) {
        const int so = (wtile0 + tap * Ctot + cc * BKE) * (int)sizeof(T);
        const __amdgpu_buffer_rsrc_t wrs = rsrc_of(a.W);
#pragma unroll
        for (int i = 0; i < WLD; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr_t)(Wring + slot * HALO_WST + i * (NTL * 16) + cw * 1024), 16, wvoff,
                                                   __builtin_amdgcn_readfirstlane(so + i * 64 * a.Ktot * (int)sizeof(T)), 0, 0);
      };
      auto issue_w2 = [&](int e, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
          const int row = (i * NTL + tl) >> 2;
          const T* wp = reinterpret_cast<const T*>(a.W2) + (size_t)(tile_n * 128 + epi_wrow(row, false)) * a.C2 + ((tl & 3) ^ swz64(row)) * EPC + e * BKE;
          __builtin_amdgcn_global_load_lds((gptr_t) reinterpret_cast<const char*>(wp),
                                           (lptr_t)(Wring + slot * HALO_WST + i * (NTL * 16) + cw * 1024), 16, 0, 0);
        }
      };
      float scr[EPC], shr[EPC];
      auto load_affine = [&](int ccx) __attribute__((always_inline)) {
        const float* sc = gnp + ccx * BKE + xlx * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) { scr[e] = sc[e]; shr[e] = sc[Ctot + e]; }
      };
      auto xform = [&](int ccx, int i) __attribute__((always_inline)) {
        if (pp[i] >= 0) {
          chunk16* q = reinterpret_cast<chunk16*>(smem + ((xpar + ccx) & 1) * Cfg::XBUF + (i * NTL + tl) * 16);
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
      };

      [[maybe_unused]] unsigned long long lvm = 0, lbar = 0;   // diagnostic builds: cycles in the counted vmcnt waits / at the barriers
      if (tc == 1) DC_STAMP(1);
      // the tile's first chunk and W tiles: issued right after the PREVIOUS tile's last barrier — that tile's last step reads the
      // other halo buffer and ring slot (sb + 2) % 3 = the slot before sb; slots sb, sb + 1 and this buffer were released earlier
      issue_x(0);
#pragma unroll
      for (int i = 0; i < PD; ++i) issue_w(0, i, (sb + i) % WR);
      if (GN) {
        hwait_vmcnt<PD * WLD>();
        load_affine(0);
#pragma unroll
        for (int i = 0; i < NXL; ++i) xform(0, i);
      }
      for (int cc = 0; cc < nchunks; ++cc) {
        const bool side_next = cc + 1 == nchunks && nx > 0;
        const bool has_next = cc + 1 < nchunks || side_next;
        const bool gn_next = GN && cc + 1 < nchunks;
        const int s0c = cc * NTAP;
        auto step = [&](auto tapc) __attribute__((always_inline)) {
          constexpr int tap = decltype(tapc)::value;
          // as in the one-tile kernel, plus the TBLN table pieces issued at tap 0 behind X(cc+1)
          {
            DC_WAIT_T0();
            if (has_next) {
              if (tap >= 1 && tap <= PD) hwait_vmcnt<FLY + NXL + TBLN>();
              else hwait_vmcnt<FLY>();
            } else {
              constexpr int left = NTAP - 1 - tap;
              hwait_vmcnt<(left < PD - 1 ? left : PD - 1) * WLD + ((tap >= 1 && tap <= PD) ? TBLN : 0)>();
            }
            if (GN && tap == 0) __builtin_amdgcn_s_waitcnt(0xC07F);
            DC_WAIT_ADD(lvm);
          }
          {
            DC_WAIT_T0();
            __builtin_amdgcn_s_barrier();
            DC_WAIT_ADD(lbar);
          }
          constexpr int t2 = tap + PD;                     // the ring slot of the tile's step s is (sb + s) % WR
          if (t2 < NTAP) issue_w(cc, t2, (sb + s0c + t2) % WR);
          else if (side_next) { if (t2 - NTAP < nx) issue_w2(t2 - NTAP, (sb + s0c + t2) % WR); }
          else if (has_next) issue_w(cc + 1, t2 - NTAP, (sb + s0c + t2) % WR);
          if (tap == 0 && has_next) issue_x(cc + 1);
          if (tap == 0) issue_tables(lid + lid_step, par ^ 1, cc == 0 && lid + lid_step < lid_end);   // chunk 0: the next tile's tables; else placeholders
          if (tap == PD + 1 && gn_next) load_affine(cc + 1);
          if (tap > PD && gn_next) xform(cc + 1, tap > PD ? tap - PD - 1 : 0);
        };
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{}); step(IC<8>{});
      }
      for (int e = 0; e < nx; ++e) {
        if (e == 0 && nx >= PD) hwait_vmcnt<FLY>();
        else hwait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (e + PD < nx) issue_w2(e + PD, (sb + nchunks * NTAP + e + PD) % WR);
        if (e + 1 < nx) issue_x(nchunks + e + 1);
      }
      if (tc == 1) { DC_STAMP(2); DC_STAMP_VAL(3, lbar); DC_STAMP_VAL(4, lvm); }
      sb = (sb + nchunks * NTAP + nx) % WR;         // (WR = 3: the nchunks * 9 steps leave the ring phase where it was)
      xpar = (xpar + nchunks + nx) & 1;
    }
    return;
  }

  // ================================================= MFMA TEAM =================================================
  const int wm = cw >> 1, wn = cw & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int xl = ((lr >> g.ltw) * g.hw + (lr & (tw - 1))) * 64 + lq * 16;
  int joff[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int p = wm * 128 + j * 16;
    const int py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
    joff[j] = __builtin_amdgcn_readfirstlane((py * g.hw + px) * 64);
  }
  const int woff0 = lds64_off(wn * 64 + lr, lq);
  __builtin_amdgcn_s_barrier();                      // the first tile's tables are in

  int sb = 0, xpar = 0, tc = 0;
  for (int lid = lid0; lid < lid_end; lid += lid_step, ++tc) {
    int tile_m, tile_n;
    tile_of(lid, tile_m, tile_n);
    const int tx = tile_m % g.tiles_x, ty = (tile_m / g.tiles_x) % g.tiles_y, ng = tile_m / tiles_img;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // One step = 32 MFMAs in four quarters of 8: (W01 x XA) (W01 x XB) (W23 x XA) (W23 x XB) — W01 / W23 = cout fragments 0-1 / 2-3,
    // XA / XB = pixel fragments 0-3 / 4-7.  A register set is re-loaded for the NEXT step as soon as its last quarter has been
    // issued: W01 after the second quarter (behind the next step's barrier), XA after the third, XB and W23 after the fourth; every
    // read then has at least 8 MFMAs (128 matrix cycles) between its issue and its first use, with ONE register set per operand
    // (no ping-pong, no step parity).  Each accumulator still takes exactly one product per step, in step order: same sums.
    chunk16 w01[2], w23[2], xa[TM / 2], xb[TM / 2];
    [[maybe_unused]] unsigned long long wbar = 0;      // diagnostic builds: cycles this wave waited at the steps' barriers
    // (the per-lane address parts enter every read through an opaque copy: hoisted out of the tap / chunk / tile loops the
    //  fragment addresses of a chunk would live in registers next to the accumulators, and spill)
    auto rd_w01 = [&](const char* Wst) __attribute__((always_inline)) {
      int wo = woff0;
      asm volatile("" : "+v"(wo));
      w01[0] = *reinterpret_cast<const chunk16*>(Wst + wo);
      w01[1] = *reinterpret_cast<const chunk16*>(Wst + wo + 1024);
    };
    auto rd_w23 = [&](const char* Wst) __attribute__((always_inline)) {
      int wo = woff0;
      asm volatile("" : "+v"(wo));
      w23[0] = *reinterpret_cast<const chunk16*>(Wst + wo + 2048);
      w23[1] = *reinterpret_cast<const chunk16*>(Wst + wo + 3072);
    };
    auto rd_xa = [&](const char* Xb, int tapoff) __attribute__((always_inline)) {
      int xo = xl;
      asm volatile("" : "+v"(xo));
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) xa[j] = *reinterpret_cast<const chunk16*>(Xb + (tapoff + joff[j]) + xo);
    };
    auto rd_xb = [&](const char* Xb, int tapoff) __attribute__((always_inline)) {
      int xo = xl;
      asm volatile("" : "+v"(xo));
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) xb[j] = *reinterpret_cast<const chunk16*>(Xb + (tapoff + joff[TM / 2 + j]) + xo);
    };
    // `more`: another step follows (its W tile at Wn, its halo buffer Xn, tap offset tapoff_n)
    auto step = [&](bool more, const char* Wn, const char* Xn, int tapoff_n) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) { acc[0][j] = Mma<T>::run(w01[0], xa[j], acc[0][j]); acc[1][j] = Mma<T>::run(w01[1], xa[j], acc[1][j]); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) { acc[0][TM / 2 + j] = Mma<T>::run(w01[0], xb[j], acc[0][TM / 2 + j]); acc[1][TM / 2 + j] = Mma<T>::run(w01[1], xb[j], acc[1][TM / 2 + j]); }
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        // every read of this step is back (the last, W23, was issued a step ago) before the next step's barrier lets the loaders
        // refill what it read
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w23[0]), "+v"(w23[1]));
        {
          DC_WAIT_T0();
          __builtin_amdgcn_s_barrier();
          DC_WAIT_ADD(wbar);
        }
        rd_w01(Wn);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) { acc[2][j] = Mma<T>::run(w23[0], xa[j], acc[2][j]); acc[3][j] = Mma<T>::run(w23[1], xa[j], acc[3][j]); }
      __builtin_amdgcn_sched_barrier(0);
      if (more) rd_xa(Xn, tapoff_n);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) { acc[2][TM / 2 + j] = Mma<T>::run(w23[0], xb[j], acc[2][TM / 2 + j]); acc[3][TM / 2 + j] = Mma<T>::run(w23[1], xb[j], acc[3][TM / 2 + j]); }
      __builtin_amdgcn_sched_barrier(0);
      if (more) { rd_xb(Xn, tapoff_n); rd_w23(Wn); }
    };
    const int S = nchunks * NTAP + nx;               // steps of the tile
    const int hw64 = g.hw * 64;
    auto slot_of = [&](int s) __attribute__((always_inline)) { return Wring + ((sb + s) % WR) * HALO_WST; };
    auto xbuf_of = [&](int c) __attribute__((always_inline)) { return smem + ((xpar + c) & 1) * Cfg::XBUF; };
    // step 0's operands
    if (tc == 1) DC_STAMP(1);
    __builtin_amdgcn_s_barrier();
    rd_w01(slot_of(0)); rd_xa(xbuf_of(0), 0); rd_xb(xbuf_of(0), 0); rd_w23(slot_of(0));
    for (int cc = 0; cc < nchunks; ++cc) {
      const char* Xb = xbuf_of(cc);
      const int s0c = cc * NTAP;
      auto tap = [&](auto tapc) __attribute__((always_inline)) {
        constexpr int tp = decltype(tapc)::value;
        if constexpr (tp < NTAP - 1) {
          constexpr int kyn = (tp + 1) / 3, kxn = (tp + 1) - kyn * 3;
          step(true, slot_of(s0c + tp + 1), Xb, kyn * hw64 + kxn * 64);
        } else {
          // tap 8: the next step is tap 0 of the next chunk, or the first side step (centre tap), or nothing
          step(s0c + NTAP < S, slot_of(s0c + NTAP), xbuf_of(cc + 1), cc + 1 < nchunks ? 0 : hw64 + 64);
        }
      };
      tap(IC<0>{}); tap(IC<1>{}); tap(IC<2>{}); tap(IC<3>{}); tap(IC<4>{}); tap(IC<5>{}); tap(IC<6>{}); tap(IC<7>{}); tap(IC<8>{});
    }
    for (int e = 0; e < nx; ++e)                     // 1x1 side source: centre tap of chunk nchunks + e
      step(e + 1 < nx, slot_of(nchunks * NTAP + e + 1), xbuf_of(nchunks + e + 1), hw64 + 64);
    if (tc == 1) { DC_STAMP(2); DC_STAMP_VAL(3, wbar); }
    sb = (sb + nchunks * NTAP + nx) % WR;
    xpar = (xpar + nchunks + nx) & 1;
    // ---- epilogue: straight from the accumulators; bias / row vector from this tile's LDS tables ----
    HaloQs qsfn;
    qsfn.nbase = ng; qsfn.ltp = g.ltw + g.lth; qsfn.n_img = g.n_img; qsfn.tile_in_img = ty * g.tiles_x + tx; qsfn.wm = wm;
    qsfn.np = HW >= 128 ? HW >> 7 : 1;
    qsfn.padd = 0;
    auto rowfn = [&](int j, EpiRow& r) __attribute__((always_inline)) {
      const int p = wm * 128 + j * 16 + lr;
      const int py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
      const int rem = (ty * th + py) * g.W + tx * tw + px;
      r.ok = true;
      r.samp = ng;
      r.o = ng * HW + rem;
      r.r = (a.residual && a.res_map ? a.res_map[ng] : ng) * HW + rem;
    };
    const float* brv = reinterpret_cast<const float*>(smem + Cfg::BRVOFF + (tc & 1) * 1024);
    epi_direct_act<T, TM, DC_ACT_NONE, false, true>(a, acc, tile_n, wn, lq, ng, ng, rowfn, EpiNoPre(), qsfn, WsLdsBias2{brv + wn * 64 + lq * 8});
    if (tc == 1) DC_STAMP(7);
    if (tc == 0) DC_STAMP(0);        // (diagnostic builds: the workgroup's SECOND tile is the stamped one — steady state)
  }
}

// ====================================================================================================================
// conv3_wr_kernel — persistent, wave-specialised, WEIGHTS THROUGH REGISTERS.
//
// What the two kernels above showed (s_memtime accounting, profiles/r03_stamp_ws_*.log): with the W[tap] tiles in the loaders' LDS-DMA
// stream the LOADER team is the pole — 104 LDS-DMA pieces per lane and tile at ~150-200 cycles of its own time each, 24 transformed
// pieces at ~500-800, all of it between one barrier per tap — and the MFMA team, however well its reads are pipelined, waits at every
// tap's barrier (loop 34-48 k cycles against 18.4 k of MFMA issue).  Three quarters of those pieces are W tiles that LDS does not
// help: a tap's W tile is read ONCE per wave and tap (4 fragments), its only reuse is across the two waves that share a cout half.
// So here:
//   * the MFMA waves fetch their W fragments THEMSELVES, straight from L2 into registers (4 x 16-byte buffer loads per wave and step,
//     two steps ahead of their use: three fragment sets rotating with the step, 9 steps per chunk = 0 mod 3);
//   * the loader team only moves and transforms the X halo chunks (6 pieces per lane and chunk), two chunks ahead through a ring of
//     THREE halo buffers: in iteration k it waits at the barrier of chunk k-2, issues chunk k, waits for chunk k-1 to land and
//     transforms it;
//   * ONE barrier per CHUNK (9 taps) instead of one per tap: B(g) = "chunk g is in LDS, normalised; chunk g-1's buffer is free";
//   * persistent workgroups as in conv3_wsp_kernel: the chunk stream runs across tiles, so the next tile's first chunks are ready
//     when the epilogue ends; per-tile tables by LDS-DMA one tile ahead.
// The MFMA team's step is the quarter schedule of conv3_wsp_kernel.  Same LDS image, tap order and accumulation order: bit-identical
// to the one-tile kernel.  Needs at least three chunks per tile (C0 + C1 + C2 >= 96 16-bit channels): the table slots alternate.
struct WrCfg {
  static constexpr int NT = 512, NTL = 256, NXL = 6, TBLN = 2, XR = 3;
  static constexpr int XBUF = NXL * NTL * 16;
  static constexpr int GNOFF = XR * XBUF;                   // [2 tile parities][scale[C] | shift[C]] (4 KiB each)
  static constexpr int GNMAXC = 512;
  static constexpr int BRVOFF = GNOFF + 2 * 2 * GNMAXC * 4; // [2 tile parities][bias[128] | rowvec[128]]
  static constexpr int DUMPOFF = BRVOFF + 2 * 256 * 4;      // placeholder pieces (4 + 4 KiB)
  static constexpr int LDS = DUMPOFF + 8 * 1024;
};

template <typename T, bool GN>
__global__ __launch_bounds__(512, 2) void conv3_wr_kernel(const IgemmArgs a, const HaloGeom g, const int total_tiles) {
  using Cfg = WrCfg;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 4 * EPC;
  constexpr int TM = 8, TN = 4, NTAP = 9;
  constexpr int NTL = Cfg::NTL, NXL = Cfg::NXL, TBLN = Cfg::TBLN, XR = Cfg::XR;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave >= 4;
  const int cw = wave & 3, tl = t & 255;
  const int tw = 1 << g.ltw, th = 1 << g.lth;
  const int HW = g.H * g.W;
  const int Ctot = a.C0 + a.C1;
  const int c0chunks = a.C0 / BKE, nchunks = Ctot / BKE;
  const int nx = a.src2 ? a.C2 / BKE : 0;
  const int NC = nchunks + nx;                              // chunks per tile (the side source's are single-step chunks)
  const int abl = DC_WS_ABL();
  const int tiles_img = g.tiles_x * g.tiles_y;
  const int xcd = blockIdx.x & 7, q8 = total_tiles >> 3, r8 = total_tiles & 7;
  const int lid0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
  const int lid_end = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + q8 + (xcd < r8 ? 1 : 0);
  const int lid_step = ((int)gridDim.x - xcd + 7) >> 3;
  const int ntile = lid0 < lid_end ? (lid_end - lid0 + lid_step - 1) / lid_step : 0;       // this workgroup's tiles
  auto tile_of = [&](int lid, int& tile_m, int& tile_n) __attribute__((always_inline)) {
    if (a.n_fast) { tile_m = lid / a.tiles_n; tile_n = lid - tile_m * a.tiles_n; }
    else { tile_n = lid / a.tiles_m; tile_m = lid - tile_n * a.tiles_m; }
    tile_m = __builtin_amdgcn_readfirstlane(tile_m);
    tile_n = __builtin_amdgcn_readfirstlane(tile_n);
  };
  auto rsrc_of = [](const void* base) __attribute__((always_inline)) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000); };

  if (loader) {
    // =============================================== LOADER TEAM ===============================================
    const int xlx = tl & 3;
    int ldb0 = a.ld0 * (int)sizeof(T), ldb1 = a.ld1 * (int)sizeof(T), ldb2 = a.ld2 * (int)sizeof(T);
    asm volatile("" : "+s"(ldb0), "+s"(ldb1), "+s"(ldb2));
    auto issue_tables = [&](int lid, int par, bool real) __attribute__((always_inline)) {
      int tile_m, tile_n;
      tile_of(real ? lid : lid0, tile_m, tile_n);
      const int ng = tile_m / tiles_img;
      const char* zero = reinterpret_cast<const char*>(g_ws_zero_page);
      {
        const int idx = cw * 256 + 4 * lane;
        const char* src = zero;
        if (GN && real && idx < Ctot) src = reinterpret_cast<const char*>(a.gn_scale + (size_t)ng * Ctot + idx);
        else if (GN && real && idx < 2 * Ctot) src = reinterpret_cast<const char*>(a.gn_shift + (size_t)ng * Ctot + (idx - Ctot));
        char* dst = real ? smem + Cfg::GNOFF + par * (2 * Cfg::GNMAXC * 4) + cw * 1024 : smem + Cfg::DUMPOFF + cw * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      }
      {
        const int c = tile_n * 128 + 4 * (lane & 31);
        const int rrow = (real && a.rowvec) ? __builtin_amdgcn_readfirstlane(a.rowvec_map ? a.rowvec_map[ng] : ng) : 0;
        const char* src = zero;
        if (real && cw == 0 && c < a.Cout) {
          if (lane < 32) { if (a.bias) src = reinterpret_cast<const char*>(a.bias + c); }
          else if (a.rowvec) src = reinterpret_cast<const char*>(a.rowvec + (size_t)rrow * a.rowvec_ld + c);
        }
        char* dst = (real && cw == 0) ? smem + Cfg::BRVOFF + par * 1024 : smem + Cfg::DUMPOFF + 4096 + cw * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      }
    };
    issue_tables(lid0, 0, ntile > 0);
    hwait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    // piece offsets of a tile's halo (pixel offset inside the sample, -1 = padding)
    auto pieces_of = [&](int tile_m, int (&pp)[NXL]) __attribute__((always_inline)) {
      const int tx = tile_m % g.tiles_x, ty = (tile_m / g.tiles_x) % g.tiles_y;
#pragma unroll
      for (int i = 0; i < NXL; ++i) {
        const int hr = (i * NTL + tl) >> 2;
        pp[i] = -1;
        if (i < g.nxl && hr < g.HR) {
          const int hy = (int)(((float)hr + 0.5f) * g.inv_hw), hx = hr - hy * g.hw;
          const int iy = ty * th + hy - 1, ix = tx * tw + hx - 1;
          if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) pp[i] = iy * g.W + ix;
        }
      }
    };
    const int N = ntile * NC;                      // chunks of this workgroup's stream
    int pp[NXL], ppv[NXL];                         // pieces of the ISSUE cursor's tile / of the tile before it
    int ti = 0, ci = 0, lid_i = lid0;              // issue cursor: tile ordinal, chunk in tile, tile id
    int ng_i = 0;
    const T* xb0 = nullptr; const T* xb1 = nullptr; const T* xb2 = nullptr;
#pragma unroll
    for (int i = 0; i < NXL; ++i) { pp[i] = -1; ppv[i] = -1; }
    for (int k = 0; k < N + 2; ++k) {
      if (k >= 2) {
        __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): my in-place writes of chunk k-2 (transformed last iteration) are in LDS
        __builtin_amdgcn_s_barrier();              // B(k-2): chunk k-2 is ready; chunk k-3's buffer (= chunk k's) is free
      }
      if (k < N) {
        if (ci == 0) {                             // the issue cursor enters a new tile
#pragma unroll
          for (int i = 0; i < NXL; ++i) ppv[i] = pp[i];
          int tile_m, tile_n;
          tile_of(lid_i, tile_m, tile_n);
          pieces_of(tile_m, pp);
          ng_i = tile_m / tiles_img;
          const int s0 = __builtin_amdgcn_readfirstlane(a.map0 ? a.map0[ng_i] : ng_i);
          xb0 = reinterpret_cast<const T*>(a.src0) + (size_t)s0 * HW * a.ld0;
          if (a.src1) { const int s1 = __builtin_amdgcn_readfirstlane(a.map1 ? a.map1[ng_i] : ng_i); xb1 = reinterpret_cast<const T*>(a.src1) + (size_t)s1 * HW * a.ld1; }
          if (a.src2) { const int s2 = __builtin_amdgcn_readfirstlane(a.map2 ? a.map2[ng_i] : ng_i); xb2 = reinterpret_cast<const T*>(a.src2) + (size_t)s2 * HW * a.ld2; }
        }
        // chunk k = chunk ci of tile ti -> halo buffer k % 3
        {
          int ldb = ldb0, cb = ci;
          const T* xb = xb0;
          if (ci >= nchunks) { ldb = ldb2; cb = ci - nchunks; xb = xb2; }
          else if (ci >= c0chunks) { ldb = ldb1; cb = ci - c0chunks; xb = xb1; }
          const int cofs = cb * 64 + xlx * 16;
          const __amdgpu_buffer_rsrc_t rs = rsrc_of(xb);
          char* xs = smem + (k % XR) * Cfg::XBUF + cw * 1024;
#pragma unroll
          for (int i = 0; i < NXL; ++i) {
            const int pk = pp[i];
            const int voff = pk < 0 ? -1 : pk * ldb + cofs;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(xs + i * (NTL * 16)), 16, voff, 0, 0, 0);
          }
        }
        // the NEXT tile's tables ride behind the last chunk of this one (placeholders elsewhere: the counted wait below is a constant)
        issue_tables(lid_i + lid_step, (ti + 1) & 1, ci == NC - 1 && ti + 1 < ntile);
      }
      if (k >= 1 && k - 1 < N) {
        // chunk k-1 (issued last iteration) has landed: everything but what this iteration issued
        if (k < N) hwait_vmcnt<NXL + TBLN>();
        else hwait_vmcnt<0>();
        // chunk k-1 = chunk (ci - 1) of this tile, or the last chunk of the tile before (then ci == 0 ... or k == N)
        const bool prev_tile = (k < N) ? ci == 0 : true;
        const int ct = (k < N && ci > 0) ? ci - 1 : NC - 1;
        const int tpar = (k < N) ? ((prev_tile ? ti - 1 : ti) & 1) : ((ntile - 1) & 1);
        if (GN && ct < nchunks && !(abl & 8)) {
          const float* gnp = reinterpret_cast<const float*>(smem + Cfg::GNOFF + tpar * (2 * Cfg::GNMAXC * 4));
          const float* sc = gnp + ct * BKE + xlx * EPC;
          float scr[EPC], shr[EPC];
#pragma unroll
          for (int e = 0; e < EPC; ++e) { scr[e] = sc[e]; shr[e] = sc[Ctot + e]; }
          char* xbuf = smem + ((k - 1) % XR) * Cfg::XBUF;
          chunk16 pc[NXL];
#pragma unroll
          for (int i = 0; i < NXL; ++i) pc[i] = *reinterpret_cast<const chunk16*>(xbuf + (i * NTL + tl) * 16);
#pragma unroll
          for (int i = 0; i < NXL; ++i) {
            const int pk = (k < N && !prev_tile) ? pp[i] : ((k < N) ? ppv[i] : pp[i]);
            if (pk >= 0) {
              float f[EPC];
              chunk_to_f<T>(pc[i], f);
#pragma unroll
              for (int e = 0; e < EPC; ++e) {
                float v = f[e] * scr[e] + shr[e];
                if (a.gn_silu) v = silu_t<T>(v);
                f[e] = v;
              }
              *reinterpret_cast<chunk16*>(xbuf + (i * NTL + tl) * 16) = f_to_chunk<T>(f);
            }
          }
        }
      }
      if (k < N) {                                  // advance the issue cursor
        if (++ci == NC) { ci = 0; ++ti; lid_i += lid_step; }
      }
    }
    return;
  }

  // ================================================= MFMA TEAM =================================================
  const int wm = cw >> 1, wn = cw & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int xl = ((lr >> g.ltw) * g.hw + (lr & (tw - 1))) * 64 + lq * 16;
  int joff[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int p = wm * 128 + j * 16;
    const int py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
    joff[j] = __builtin_amdgcn_readfirstlane((py * g.hw + px) * 64);
  }
  // W fragment i of this lane: packed weight row epi_wrow(wn * 64 + 16 i + lr) (the row the LDS-DMA kernels put at that LDS row),
  // k elements 8 lq .. +7 of the step's 32-wide slice: per-lane byte offsets, the (N tile, tap, chunk) part is the scalar offset
  // (fragment i's row is fragment 0's + (i >> 1) * 32 + (i & 1) * 4: that part goes into the scalar offset too — one offset register)
  const int wrow0 = epi_wrow(wn * 64 + lr, false);
  const int wvo = (wrow0 * a.Ktot + lq * EPC) * (int)sizeof(T), wvo2 = (wrow0 * a.C2 + lq * EPC) * (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t wrs = rsrc_of(a.W), wrs2 = rsrc_of(a.src2 ? a.W2 : a.W);
  __builtin_amdgcn_s_barrier();                      // the first tile's tables are in

  int gc = 0;                                        // chunks consumed so far: chunk g lives in halo buffer g % 3
  const int hw64 = g.hw * 64;
  for (int it = 0; it < ntile; ++it) {
    int tile_m, tile_n;
    tile_of(lid0 + it * lid_step, tile_m, tile_n);
    const int tx = tile_m % g.tiles_x, ty = (tile_m / g.tiles_x) % g.tiles_y, ng = tile_m / tiles_img;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    chunk16 ws[3][TN], xa[TM / 2], xb[TM / 2];     // W fragment sets (step s uses set s % 3), X fragments of the current step
    const int wt0 = tile_n * 128 * a.Ktot * (int)sizeof(T), wt2 = tile_n * 128 * a.C2 * (int)sizeof(T);
    // W fragments of step (chunk cc, tap tp) of the 3x3 part / of side step e, into set P
    auto ld_w = [&](auto pc, int cc, int tp) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
#ifdef DC_WR_CONTIG     // timing only (diagnostic builds): the fragment loads a fragment-major weight image would need (1 KiB contiguous each)
      const int soc = __builtin_amdgcn_readfirstlane(wt0 + ((cc * NTAP + tp) * 8 + wn * 4) * 1024);
#pragma unroll
      for (int i = 0; i < TN; ++i)
        ws[P][i] = __builtin_bit_cast(chunk16, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, soc + i * 1024, 0));
      return;
#endif
      const int so = __builtin_amdgcn_readfirstlane(wt0 + (tp * Ctot + cc * BKE) * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < TN; ++i)
        ws[P][i] = __builtin_bit_cast(chunk16, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so + ((i >> 1) * 32 + (i & 1) * 4) * a.Ktot * (int)sizeof(T), 0));
    };
    auto ld_w2 = [&](auto pc, int e) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
      const int so = __builtin_amdgcn_readfirstlane(wt2 + e * BKE * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < TN; ++i)
        ws[P][i] = __builtin_bit_cast(chunk16, __builtin_amdgcn_raw_buffer_load_b128(wrs2, wvo2, so + ((i >> 1) * 32 + (i & 1) * 4) * a.C2 * (int)sizeof(T), 0));
    };
    auto rd_xa = [&](const char* Xb, int tapoff) __attribute__((always_inline)) {
      int xo = xl;
      asm volatile("" : "+v"(xo));
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) xa[j] = *reinterpret_cast<const chunk16*>(Xb + (tapoff + joff[j]) + xo);
    };
    auto rd_xb = [&](const char* Xb, int tapoff) __attribute__((always_inline)) {
      int xo = xl;
      asm volatile("" : "+v"(xo));
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) xb[j] = *reinterpret_cast<const chunk16*>(Xb + (tapoff + joff[TM / 2 + j]) + xo);
    };
    auto xbuf_of = [&](int c) __attribute__((always_inline)) { return smem + ((gc + c) % XR) * Cfg::XBUF; };
    // one step on W set P: four quarters of 8 MFMAs; `last`: last step of its chunk — the next chunk's barrier after the second
    // quarter (every LDS read of this chunk is back by then); `more`: another step follows in this tile (halo buffer Xn, offset tn)
    auto step = [&](auto pc, bool last, bool more, const char* Xn, int tn) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) { acc[0][j] = Mma<T>::run(ws[P][0], xa[j], acc[0][j]); acc[1][j] = Mma<T>::run(ws[P][1], xa[j], acc[1][j]); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) { acc[0][TM / 2 + j] = Mma<T>::run(ws[P][0], xb[j], acc[0][TM / 2 + j]); acc[1][TM / 2 + j] = Mma<T>::run(ws[P][1], xb[j], acc[1][TM / 2 + j]); }
      __builtin_amdgcn_sched_barrier(0);
      if (last && more) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) { acc[2][j] = Mma<T>::run(ws[P][2], xa[j], acc[2][j]); acc[3][j] = Mma<T>::run(ws[P][3], xa[j], acc[3][j]); }
      __builtin_amdgcn_sched_barrier(0);
      if (more) rd_xa(Xn, tn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TM / 2; ++j) { acc[2][TM / 2 + j] = Mma<T>::run(ws[P][2], xb[j], acc[2][TM / 2 + j]); acc[3][TM / 2 + j] = Mma<T>::run(ws[P][3], xb[j], acc[3][TM / 2 + j]); }
      __builtin_amdgcn_sched_barrier(0);
      if (more) rd_xb(Xn, tn);
    };
    // the tile's first two W fragment sets, its first chunk's barrier, step 0's X fragments
    if (it == 1) DC_STAMP(1);
    ld_w(IC<0>{}, 0, 0);
    ld_w(IC<1>{}, 0, 1);
    __builtin_amdgcn_s_barrier();
    rd_xa(xbuf_of(0), 0); rd_xb(xbuf_of(0), 0);
    for (int cc = 0; cc < nchunks; ++cc) {
      const char* Xb = xbuf_of(cc);
      auto tap = [&](auto tapc) __attribute__((always_inline)) {
        constexpr int tp = decltype(tapc)::value;
        constexpr int P = tp % 3, PN = (tp + 2) % 3;
        // W fragments two steps ahead (set PN: the set of the step before this one, all of whose MFMAs have been issued)
        if constexpr (tp + 2 < NTAP) ld_w(IC<PN>{}, cc, tp + 2);
        else {
          if (cc + 1 < nchunks) ld_w(IC<PN>{}, cc + 1, tp + 2 - NTAP);
          else if (tp + 2 - NTAP < nx) ld_w2(IC<PN>{}, tp + 2 - NTAP);
        }
        if constexpr (tp < NTAP - 1) {
          constexpr int kyn = (tp + 1) / 3, kxn = (tp + 1) - kyn * 3;
          step(IC<P>{}, false, true, Xb, kyn * hw64 + kxn * 64);
        } else {
          step(IC<P>{}, true, cc + 1 < NC, xbuf_of(cc + 1), cc + 1 < nchunks ? 0 : hw64 + 64);
        }
      };
      tap(IC<0>{}); tap(IC<1>{}); tap(IC<2>{}); tap(IC<3>{}); tap(IC<4>{}); tap(IC<5>{}); tap(IC<6>{}); tap(IC<7>{}); tap(IC<8>{});
    }
    // 1x1 side source: step e = the centre tap of chunk nchunks + e, W2 fragments in set e % 3 (9 nchunks = 0 mod 3)
    for (int e0 = 0; e0 < nx; e0 += 3) {
      auto side = [&](auto pc, int e) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value;
        if (e + 2 < nx) ld_w2(IC<(P + 2) % 3>{}, e + 2);
        step(IC<P>{}, true, e + 1 < nx, xbuf_of(nchunks + e + 1), hw64 + 64);
      };
      side(IC<0>{}, e0);
      if (e0 + 1 < nx) side(IC<1>{}, e0 + 1);
      if (e0 + 2 < nx) side(IC<2>{}, e0 + 2);
    }
    gc += NC;
    if (it == 1) DC_STAMP(2);
    // ---- epilogue: straight from the accumulators; bias / row vector from this tile's LDS tables ----
    HaloQs qsfn;
    qsfn.nbase = ng; qsfn.ltp = g.ltw + g.lth; qsfn.n_img = g.n_img; qsfn.tile_in_img = ty * g.tiles_x + tx; qsfn.wm = wm;
    qsfn.np = HW >= 128 ? HW >> 7 : 1;
    qsfn.padd = 0;
    auto rowfn = [&](int j, EpiRow& r) {
      const int p = wm * 128 + j * 16 + lr;
      const int py = (p >> g.ltw) & (th - 1), px = p & (tw - 1);
      const int rem = (ty * th + py) * g.W + tx * tw + px;
      r.ok = true;
      r.samp = ng;
      r.o = ng * HW + rem;
      r.r = (a.residual && a.res_map ? a.res_map[ng] : ng) * HW + rem;
    };
    const float* brv = reinterpret_cast<const float*>(smem + Cfg::BRVOFF + (it & 1) * 1024);
    epi_direct_act<T, TM, DC_ACT_NONE, false, true>(a, acc, tile_n, wn, lq, ng, ng, rowfn, EpiNoPre(), qsfn, WsLdsBias2{brv + wn * 64 + lq * 8});
    if (it == 1) DC_STAMP(7);
    if (it == 0) DC_STAMP(0);
  }
}

static int ws_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// true when the wave-specialised kernel can take this problem: what conv3_halo's one-image-per-patch / buffer-descriptor form takes
// (3x3 stride 1, power-of-two images of at least 256 pixels and 16 columns, every source sample below 2 GiB), affine table within its slot
bool dc_conv3_ws_ok(const IgemmArgs& a, int dtype) {
  static const bool off = getenv("DCAMD_NO_WS") != nullptr;
  if (off || !dc_conv3_halo_applicable(a, dtype) || a.upsample) return false;
  const int H = a.Hin, W = a.Win;
  if (H < 8 || W < 16 || H * W < 256) return false;
  const int tw = W < 32 ? W : 32;
  const int th = 256 / tw;
  if (th > H) return false;
  const int hr = (th + 2) * (tw + 2);
  if ((hr * 4 + WsCfg::NTL - 1) / WsCfg::NTL > WsCfg::NXL) return false;
  if (a.gn_scale && a.C0 + a.C1 > WsCfg::GNMAXC) return false;
  const long long es = dc_dtype_size(dtype);
  const long long ldmax = a.ld0 > a.ld1 ? (a.ld0 > a.ld2 ? a.ld0 : a.ld2) : (a.ld1 > a.ld2 ? a.ld1 : a.ld2);
  if ((long long)H * W * ldmax * es >= (1LL << 31)) return false;
  if ((long long)a.tiles_n * 128 * a.Ktot * es >= (1LL << 31)) return false;
  return true;
}

template <typename T>
static int launch_ws(const IgemmArgs& a0, int n_img, hipStream_t s) {
  using Cfg = WsCfg;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_ws_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_ws_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    attr_done = true;
  }
  IgemmArgs a = a0;
  HaloGeom g;
  g.H = a.Hin; g.W = a.Win; g.n_img = n_img;
  const int tw = g.W < 32 ? g.W : 32, th = 256 / tw;
  g.ltw = ws_ilog2(tw); g.lth = ws_ilog2(th); g.lni = 0;
  g.tiles_x = g.W / tw; g.tiles_y = g.H / th;
  g.hw = tw + 2; g.hp = (th + 2) * g.hw; g.HR = g.hp;
  g.mos = 0; g.lmc = 0; g.inv_ch = g.inv_cw = 0.f; g.xbuf = 1;
  g.sws = 2;          // fragments of 16 pixels in a row (tw >= 16); the opt-in persistent kernels below keep the un-swizzled image
  g.inv_hp = 1.0f / (float)g.hp; g.inv_hw = 1.0f / (float)g.hw;
  g.nxl = (g.HR * 4 + Cfg::NTL - 1) / Cfg::NTL;
  if (g.nxl > Cfg::NXL || g.nxl < 3) { dc_set_error("conv3_ws: halo of %d rows does not fit", g.HR); return DC_ERR_SHAPE; }
  a.tiles_m = n_img * g.tiles_x * g.tiles_y;
  const long long nblk = (long long)a.tiles_m * a.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { dc_set_error("conv3_ws: bad grid %lld", nblk); return DC_ERR_SHAPE; }
  // Which kernel: the one-tile kernel (conv3_ws_kernel).  The two persistent forms are OPT-IN experiments, both measured slower on
  // cfg2 (20 fused convs per step; profiles/r03_stamp_ws_*.log):
  //   DCAMD_WS_PERSIST  conv3_wsp_kernel (W tiles by LDS-DMA): 27.3 ms against 26.9 — the loader team is the pole (~150-200 cycles of its
  //                     own time per LDS-DMA piece, 104 pieces per tile, ~500-800 per transformed piece, all between lock-step barriers);
  //   DCAMD_WS_WR       conv3_wr_kernel (W fragments by the MFMA waves' own buffer loads, one barrier per chunk): 43.0 ms — a fragment
  //                     load takes 16 bytes from each of 16 weight rows per 16-lane group (half of every 128-byte line unused, 16 lines
  //                     per instruction): 1770 cycles per step at the vector-memory path against 512 of MFMA issue.  A fragment-major
  //                     weight layout would make those loads whole lines; not built.
  static const bool env_wsp = getenv("DCAMD_WS_PERSIST") != nullptr, env_wr = getenv("DCAMD_WS_WR") != nullptr;
  const int bke64 = 64 / (int)sizeof(T);
  const int nchunks_all = (a.C0 + a.C1) / bke64 + (a.src2 ? a.C2 / bke64 : 0);
  const bool use_wr = env_wr && !env_wsp && nchunks_all >= 3;
  if (!use_wr && !env_wsp) {
    void (*kern)(const IgemmArgs, const HaloGeom) = a.gn_scale ? conv3_ws_kernel<T, true> : conv3_ws_kernel<T, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(Cfg::NT), Cfg::LDS, s, a, g);
    return dc_check_launch("dc_igemm(conv3_ws)");
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { dc_set_error("conv3_ws: no device properties"); return DC_ERR_LAUNCH; }
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  static bool attr_p = false;
  if (!attr_p) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wsp_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WspCfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wsp_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, WspCfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wr_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WrCfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wr_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, WrCfg::LDS);
    attr_p = true;
  }
  // one workgroup per CU (8 waves at <= 256 registers, 83 KiB of LDS); every XCD label needs at least one workgroup per 8 tiles
  const int grid = (int)(nblk < n_cu ? nblk : n_cu);
  if (use_wr) {
    void (*kr)(const IgemmArgs, const HaloGeom, const int) = a.gn_scale ? conv3_wr_kernel<T, true> : conv3_wr_kernel<T, false>;
    hipLaunchKernelGGL(kr, dim3((unsigned)grid), dim3(WrCfg::NT), WrCfg::LDS, s, a, g, (int)nblk);
    return dc_check_launch("dc_igemm(conv3_wr)");
  }
  void (*kp)(const IgemmArgs, const HaloGeom, const int) = a.gn_scale ? conv3_wsp_kernel<T, true> : conv3_wsp_kernel<T, false>;
  hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(WspCfg::NT), WspCfg::LDS, s, a, g, (int)nblk);
  return dc_check_launch("dc_igemm(conv3_wsp)");
}

int dc_conv3_ws_launch(const IgemmArgs& a, int dtype, int n_img, hipStream_t s) {
  if (dtype == DC_BF16) return launch_ws<__bf16>(a, n_img, s);
  if (dtype == DC_F16) return launch_ws<_Float16>(a, n_img, s);
  return launch_ws<float>(a, n_img, s);
}
