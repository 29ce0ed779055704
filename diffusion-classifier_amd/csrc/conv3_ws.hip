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
// One workgroup per CU (8 waves at <= 256 registers); 78 KiB of LDS.  Two further forms (persistent workgroups with a pipelined MFMA team;
// W fragments through registers) were built in round 3, measured slower and removed: DESIGN.md 6d / 8 and the git history keep them.
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
// instructions at all, 32 no fragment reads (and no MFMAs)
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
  g.sws = 2;          // fragments of 16 pixels in a row (tw >= 16)
  g.inv_hp = 1.0f / (float)g.hp; g.inv_hw = 1.0f / (float)g.hw;
  g.nxl = (g.HR * 4 + Cfg::NTL - 1) / Cfg::NTL;
  if (g.nxl > Cfg::NXL || g.nxl < 3) { dc_set_error("conv3_ws: halo of %d rows does not fit", g.HR); return DC_ERR_SHAPE; }
  a.tiles_m = n_img * g.tiles_x * g.tiles_y;
  const long long nblk = (long long)a.tiles_m * a.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { dc_set_error("conv3_ws: bad grid %lld", nblk); return DC_ERR_SHAPE; }
  void (*kern)(const IgemmArgs, const HaloGeom) = a.gn_scale ? conv3_ws_kernel<T, true> : conv3_ws_kernel<T, false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(Cfg::NT), Cfg::LDS, s, a, g);
  return dc_check_launch("dc_igemm(conv3_ws)");
}

int dc_conv3_ws_launch(const IgemmArgs& a, int dtype, int n_img, hipStream_t s) {
  if (dtype == DC_BF16) return launch_ws<__bf16>(a, n_img, s);
  if (dtype == DC_F16) return launch_ws<_Float16>(a, n_img, s);
  return launch_ws<float>(a, n_img, s);
}
