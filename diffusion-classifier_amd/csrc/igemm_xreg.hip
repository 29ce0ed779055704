// igemm_xreg.hip — activation-stationary GEMM for the short-K linear layers (K <= 256, 16-bit): the 1x1
// shortcuts and every projection of the 16x16 transformer blocks (q/k/v, attention out, proj_in/out, GEGLU).
//
// Why: with K = 256 the pipelined tile kernel (igemm_pipe.hip) spends a tile's life in prologue / drain — four
// K-steps cannot hide the HBM latency of the activation rows, and every N tile re-fetches them (measured
// 310-370 TFLOP/s, 13 us per 128x128 tile against 1 us of MFMA work).  Here a workgroup owns 96 pixel rows for
// the WHOLE output width:
//   * the activation rows are read from HBM exactly once, straight into registers as MFMA B-operand fragments
//     (3 pixel fragments x K/32 chunks per wave: 96 VGPRs at K = 256) and stay there;
//   * the weight matrix (<= 1 MiB, L2 resident) streams through a 4-slot LDS ring of 128 couts x 64 k slices by
//     LDS-DMA, prefetch distance 3 slices, one s_barrier per slice; the stream never drains between N tiles;
//   * per N tile a wave (48 rows x 64 couts) reads 4 weight fragments per 12 MFMAs — a third of the LDS traffic of
//     the tile kernel — and finishes with the lane-resident epilogue (igemm_epilogue.h) while the next tile's
//     weight slices are already landing.  Two workgroups per CU, so one's epilogue overlaps the other's MFMAs.
//
// vmcnt bookkeeping: LDS-DMA groups are waited for with counted vmcnt; the epilogue's stores share that counter,
// so the wave drains its own DMA (vmcnt 0) BEFORE an epilogue and remembers how many slices are known to have
// landed — the next PD slices need no wait, and by the time one is needed the stores are three slices old.
#include <stdlib.h>
#include "igemm_epilogue.h"
DC_CLOCK_DECL(igemm_xreg)
#ifdef DC_XR_STAMPS
// diagnostic build only (tools/stamp_xreg.py): cycles wave 0 of every workgroup spends in its K loops / in its epilogues (s_memtime)
static __device__ unsigned long long* g_xr_stamps;
extern "C" void dc_debug_set_xr_stamps(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xr_stamps), &p, sizeof(p)); }
#define XR_NOW() __builtin_amdgcn_s_memtime()
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace {
constexpr int XR_TN = 4;             // cout fragments per wave (64 couts); pixel fragments per wave TM and K/32 chunks KC
                                     // are template parameters: (3, 8) for K <= 256, (2, 16) for K <= 512
constexpr int XR_R = 4, XR_PD = 3;   // weight ring slots, prefetch distance (slices of 64 k)
constexpr int XR_SLICE = 128 * 128;  // bytes per slice: 128 couts x 64 k x 2 B
constexpr int XR_WL = 4;             // LDS-DMA instructions per lane per slice (256 lanes x 16 B x 4 = 16 KiB)
}

// bias values of an N tile held in registers since before its K loop (igemm_epilogue.h BiasFn)
struct XrPreBias {
  static constexpr bool on = true, has_rowvec = false;
  const float (&v)[8]; const float (&g)[8];
  __device__ __forceinline__ void operator()(int, float (&bs)[8], float (&bgt)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs[e] = v[e]; bgt[e] = g[e]; }
  }
};

template <int N> static __device__ __forceinline__ void xr_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// GEGLU: one kernel per epilogue variant — the activations are live across the epilogue, and two variants inside one
// kernel pushed hipcc into spilling them
template <typename T, bool GEGLU, int TM, int KC>
__global__ __launch_bounds__(256, 2) void igemm_xreg_kernel(const IgemmArgs a) {
  static_assert(sizeof(T) == 2, "16-bit operands only");
  constexpr int TN = XR_TN, R = XR_R, PD = XR_PD, WL = XR_WL;
  constexpr int WROWS = TM * 16, XR_BM = 2 * WROWS;     // pixel rows per wave / per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int tile_m = blockIdx.x;
  const int HWo = a.Hout * a.Wout;
  const int kcn = a.Ktot >> 5;                 // 32-element chunks of K
  const int c0c = a.C0 >> 5;
  const int spn = a.Ktot >> 6;                 // slices per N tile
  const int Q = a.tiles_n * spn;               // slices of the whole weight matrix

  // ---- weight loader (LDS image of a slice = the W stage of igemm_pipe.hip: 128-B rows, lds_off swizzle) ----
  const int lrow = t >> 3;
  const int lchunk = (t & 7) ^ ((t >> 4) & 7);
  constexpr bool geglu = GEGLU;
  int wro[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) wro[i] = epi_wrow(lrow + 32 * i, geglu) * a.Ktot + lchunk * 8;
  const T* const Wg = reinterpret_cast<const T*>(a.W);
  int iq = 0, i_nt = 0, i_ks = 0;              // next slice to issue: index, N tile, slice inside the tile
  auto issue = [&]() {
    const T* wp = Wg + (size_t)i_nt * 128 * a.Ktot + i_ks * 64;
    char* dst = smem + (iq % R) * XR_SLICE + wave * 1024;
#pragma unroll
    for (int i = 0; i < WL; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t) reinterpret_cast<const char*>(wp + wro[i]), (lptr_t)(dst + i * 4096), 16, 0, 0);
    ++iq;
    if (++i_ks == spn) { i_ks = 0; ++i_nt; }
  };
#pragma unroll
  for (int p = 0; p < PD; ++p)
    if (iq < Q) issue();

  // ---- the activation rows of this wave, once, as B-operand fragments: lane (lr, lq) holds k = 32 kc + 8 lq .. +7 of row lr ----
  chunk16 xr[TM][KC];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = tile_m * XR_BM + wm * WROWS + j * 16 + lr;
    const int mm = m < a.M ? m : a.M - 1;      // rows past M compute garbage-free duplicates, never stored
    const int n = mm / HWo, pix = mm - n * HWo;
    const T* b0 = reinterpret_cast<const T*>(a.src0) + ((size_t)(a.map0 ? a.map0[n] : n) * HWo + pix) * a.ld0 + lq * 8;
    const T* b1 = a.src1 ? reinterpret_cast<const T*>(a.src1) + ((size_t)(a.map1 ? a.map1[n] : n) * HWo + pix) * a.ld1 + lq * 8 : b0;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      xr[j][kc] = chunk16{0u, 0u, 0u, 0u};
      if (kc < kcn) xr[j][kc] = *reinterpret_cast<const chunk16*>(kc < c0c ? b0 + kc * 32 : b1 + (kc - c0c) * 32);
    }
  }
  // consume the loads HERE: a load still pending at the loop head would make hipcc wait vmcnt(0) in every iteration
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) asm volatile("" ::"v"(xr[j][kc]));
  // fused row LayerNorm (no affine: gamma / beta are folded into W / bias by the packer): a row's K channels sit in the four
  // lanes lq = 0..3 of its column lr, 8 per 32-channel chunk — two-pass mean / variance in fp32, two xor-shuffles each,
  // and the normalised row goes back into the fragment registers (rounded to T exactly where the LayerNorm kernel would)
  if (a.ln_eps > 0.f) {
    const float invK = 1.0f / (float)a.Ktot;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float sum = 0.f;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        if (kc < kcn) {
          float f[8];
          chunk_to_f<T>(xr[j][kc], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += f[e];
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * invK;
      float var = 0.f;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        if (kc < kcn) {
          float f[8];
          chunk_to_f<T>(xr[j][kc], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) var += (f[e] - mean) * (f[e] - mean);
        }
      var += __shfl_xor(var, 16, 64);
      var += __shfl_xor(var, 32, 64);
      const float rstd = rsqrtf(var * invK + a.ln_eps);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        if (kc < kcn) {
          float f[8];
          chunk_to_f<T>(xr[j][kc], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd;
          xr[j][kc] = f_to_chunk<T>(f);
        }
    }
  }
  int landed = iq;                             // that wait also covered the prologue's weight slices

  const uint32_t sbase = lds_addr_of(smem);
  uint32_t woff[2];                            // fragment read offsets of sub-chunk 0 / 1 (cout fragment i adds i * 2048)
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) woff[sub] = sbase + lds_off(wn * 64 + lr, sub * 4 + lq);

  int q = 0;
#ifdef DC_XR_STAMPS
  unsigned long long xs_k = 0, xs_e = 0, xs_t0 = XR_NOW(), xs_w = 0;
#endif
  DC_CLOCK(0);       // around the whole N-tile loop: K loops and the (VALU-heavy) epilogues between them
  for (int nt = 0; nt < a.tiles_n; ++nt) {
    // GEGLU: this N tile's bias (8 value-row and 8 gate-row entries per lane) is fetched HERE, ahead of the K loop: the epilogue
    // of the short-K projection ran once per 96 MFMAs and paid a global-load latency in front of its VALU work every time.
    // (These loads are older than every weight slice issued below, so the counted vmcnt waits of the loop still mean what
    //  they say: vmcnt is in order.)
    float pbv[8], pbg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { pbv[e] = 0.f; pbg[e] = 0.f; }
    if (GEGLU && a.bias) {
      const int c = nt * 64 + wn * 32 + lq * 8;
      if (c < (a.Cout >> 1)) {
        const int p = (c >> 4) * 32 + (c & 15);
        ld8(a.bias + p, pbv);
        ld8(a.bias + p + 16, pbg);
      }
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef DC_XR_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long xs_a = XR_NOW();
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int ks = 0; ks < KC / 2; ++ks) {
      if (ks < spn) {
#ifdef DC_XR_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long xs_w0 = XR_NOW();
#endif
        if (q >= landed) {                     // my pieces of slice q; younger groups that may stay in flight: min(PD-1, Q-1-q)
          const int young = Q - 1 - q;
          if (young >= 2) xr_wait_vmcnt<2 * WL>();
          else if (young == 1) xr_wait_vmcnt<WL>();
          else xr_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();          // everyone's pieces are in; everyone is done with slice q-1 (its slot is refilled next)
#ifdef DC_XR_STAMPS
        xs_w += XR_NOW() - xs_w0;
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (iq < Q) issue();
        const uint32_t sl = (q % R) * XR_SLICE;
        chunk16 wf[2][TN];
        lgkm_fence0();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int i = 0; i < TN; ++i) wf[sub][i] = ds_read16_async(woff[sub] + sl + i * 2048);
#define XR_MMA(SUB, I, NLEFT)                                                                        \
        lgkm_wait<NLEFT>(wf[SUB][I]);                                                                \
        _Pragma("unroll") for (int j = 0; j < TM; ++j) acc[I][j] = Mma<T>::run(wf[SUB][I], xr[j][2 * ks + SUB], acc[I][j]); \
        __builtin_amdgcn_sched_barrier(0);
        XR_MMA(0, 0, 7) XR_MMA(0, 1, 6) XR_MMA(0, 2, 5) XR_MMA(0, 3, 4)
        XR_MMA(1, 0, 3) XR_MMA(1, 1, 2) XR_MMA(1, 2, 1) XR_MMA(1, 3, 0)
#undef XR_MMA
        ++q;
      }
    }
#ifdef DC_XR_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long xs_b = XR_NOW();
    xs_k += xs_b - xs_a;
    __builtin_amdgcn_sched_barrier(0);
#endif
    auto rowfn = [&](int j, EpiRow& r) {
      const int m = tile_m * XR_BM + wm * WROWS + j * 16 + lr;
      r.ok = m < a.M;
      const int mm = r.ok ? m : a.M - 1;
      const int n = mm / HWo;
      r.samp = n;
      r.o = mm;
      r.r = (a.residual && a.res_map) ? a.res_map[n] * HWo + (mm - n * HWo) : mm;
    };
    const int mw0 = tile_m * XR_BM + wm * WROWS;
    const int sf = min(mw0, a.M - 1) / HWo, sl_ = min(mw0 + WROWS - 1, a.M - 1) / HWo;
    // SiLU / tanh-GELU / gated outputs stay on igemm_pipe.hip
    // the epilogue drains my own LDS-DMA (vmcnt 0) right after issuing its bias / row-vector loads and before its stores
    // join the counter: the next iq - q slices are then known to be in
    auto drain = [&]() {
      xr_wait_vmcnt<0>();
      landed = iq;
    };
    if constexpr (GEGLU) {
      epi_direct_act<T, TM, DC_ACT_GEGLU, false, true>(a, acc, nt, wn, lq, sf, sl_, rowfn, drain, EpiNoQs(), XrPreBias{pbv, pbg});
    } else {
      epi_direct_act<T, TM, DC_ACT_NONE, false, true>(a, acc, nt, wn, lq, sf, sl_, rowfn, drain);
    }
#ifdef DC_XR_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    xs_e += XR_NOW() - xs_b;
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
#ifdef DC_XR_STAMPS
  if (threadIdx.x == 0 && g_xr_stamps) {
    g_xr_stamps[blockIdx.x * 4] = xs_k; g_xr_stamps[blockIdx.x * 4 + 1] = xs_e; g_xr_stamps[blockIdx.x * 4 + 2] = XR_NOW() - xs_t0; g_xr_stamps[blockIdx.x * 4 + 3] = xs_w;
  }
#endif
  DC_CLOCK(1);
}

bool dc_igemm_xreg_applicable(const IgemmArgs& a, int dtype) {
  static const bool off = getenv("DCAMD_NO_XREG") != nullptr;
  if (off || dtype == DC_F32 || a.taps != 1 || a.gate) return false;
  if (a.act != DC_ACT_NONE && a.act != DC_ACT_GEGLU) return false;
  if (a.rowvec && a.Hout * a.Wout < 48) return false;     // a wave's 48 (32) rows must not span more than two samples
  // pays when the activation rows are reused across N tiles (measured: N = 2048 605 vs 317 TFLOP/s, N = 768 549 vs 333;
  // N = 256 equal, N = 128 slower than the tile kernel, whose two short tiles per CU overlap better)
  constexpr int min_tiles = 3;
  if (a.tiles_n < min_tiles) return false;
  if (a.Ktot > 512 || (a.Ktot & 63) || (a.C0 & 31) || (a.C1 & 31)) return false;
  return true;
}

template <typename T, int TM, int KC>
static int xreg_launch_t(const IgemmArgs& a0, hipStream_t s) {
  constexpr int lds = XR_R * XR_SLICE;          // 64 KiB: two workgroups per CU
  IgemmArgs a = a0;
  a.tiles_m = (a.M + 2 * TM * 16 - 1) / (2 * TM * 16);
  static bool attr_done[2] = {false, false};
  const bool gg = a.act == DC_ACT_GEGLU;
  void (*kern)(const IgemmArgs) = gg ? igemm_xreg_kernel<T, true, TM, KC> : igemm_xreg_kernel<T, false, TM, KC>;
  if (!attr_done[gg ? 1 : 0]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done[gg ? 1 : 0] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles_m), dim3(256), lds, s, a);
  return dc_check_launch("dc_igemm(xreg)");
}

int dc_igemm_xreg_launch(const IgemmArgs& a, int dtype, hipStream_t s) {
  // K <= 256: 96 rows per workgroup (3 fragments x 8 chunks = 96 activation registers per wave);
  // K <= 512: 64 rows (2 x 16 = 128 registers)
  if (a.Ktot <= 256) return dtype == DC_BF16 ? xreg_launch_t<__bf16, 3, 8>(a, s) : xreg_launch_t<_Float16, 3, 8>(a, s);
  return dtype == DC_BF16 ? xreg_launch_t<__bf16, 2, 16>(a, s) : xreg_launch_t<_Float16, 2, 16>(a, s);
}
