// igemm_pipe.hip — the throughput implicit-GEMM kernel: 256 pixels x 128 couts per 512-thread
// workgroup (8 waves as 4(M) x 2(N), 64x64 per wave), K-step = 128 B per row, operands brought
// HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging) into a 3-stage ring with a
// prefetch distance of two K-steps, ONE raw s_barrier per K-step and counted vmcnt waits (never 0
// inside the loop), so two K-steps of loads are always in flight behind the MFMAs.
//
// LDS-DMA writes base + lane*16, i.e. the LDS image is lane-linear; the XOR swizzle of lds_off()
// is therefore applied on the per-lane SOURCE address (which 16-B chunk of the row a lane fetches)
// and again on the fragment reads.  Out-of-image taps (3x3 padding) and rows past M fetch from a
// 256-B zero page, so the gather needs no branches and no LDS zero-fill.
//
// Pipeline (S = 3 stages; group g = the loads of K-step g, 6 LDS-DMA instructions per lane):
//   prologue: issue g0, g1
//   iteration ks:  s_waitcnt vmcnt(6|0)   own loads of g[ks] have landed (g[ks+1] may be in flight)
//                  s_barrier              everybody's have, and everybody finished compute(ks-1)
//                  issue g[ks+2]          into the stage compute(ks-1) has just released
//                  compute(ks)            16 ds_read_b128 + 32 MFMA 16x16x32 per wave
#include <stdlib.h>
#include "igemm_epilogue.h"

__device__ chunk16 g_zero_page[16];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BM = 256, S = 3: 8 waves, 144 KiB, one workgroup per CU  — long K loops (3x3 convs, big-K GEMMs)
// BM = 128, S = 2: 4 waves,  66 KiB, two workgroups per CU — short K loops, where the prologue/epilogue of
//                  one workgroup must overlap the K loop of the other (K <= 512 spends most of a tile there)
// NH: 64-cout halves per wave — always 1 here (the 256 x 256 tile, NH = 2, is igemm_wide.hip's 8-phase kernel)
// EV: epilogue variant compiled in: -1 = all five behind a wave-uniform switch; 0 = plain, 2 = GEGLU, 3 = tanh-GELU, 4 = plain + gate.
//     The 256x256 kernel holds 128 accumulator registers and is instantiated per variant (several variants in one
//     kernel pushed hipcc into spilling accumulators inside the K loop).
// SLIM: 1-tap GEMM — the loader keeps one row pointer per source instead of the tap geometry (gating this at run time
//       kept the tap state live and cost the GEMMs ~20 %: it is a separate instantiation).
template <typename T, int BM, int S, int NH, int EV, bool SLIM>
__global__ __launch_bounds__(BM * 2, 2) void igemm_pipe_kernel(const IgemmArgs a) {
  static_assert(NH == 1, "the 256 x 256 tile (NH = 2) moved to igemm_wide.hip in round 3; its 2-stage loop here was removed in round 4");
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 8 * EPC;
  constexpr int BN = 128 * NH;
  constexpr int NT = BM * 2;                 // threads
  constexpr int RPI = NT / 8;                // tile rows covered by one LDS-DMA instruction of the workgroup
  constexpr int WL = BN / RPI;               // W loads per lane per K-step (X loads: BM / RPI == 4)
  constexpr int LPS = 4 + WL;                // LDS-DMA instructions per lane per K-step
  constexpr int XST = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int TM = 4, TN = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  int tile_m, tile_n;
  tile_of_block(a, tile_m, tile_n);
  // EV == 5: one phase of "nearest-2x upsample, then 3x3 conv" on the LOW-resolution image (dc_igemm_params.up4; sources too
  // small for the halo kernel): 4 taps (dy, dx) at offsets (pa + dy, pb + dx) of the padded source, the four phases are four
  // times the N tiles (phase-major stacked weights [4 * Cout_pad][4 * C]), output pixel (2y + pa, 2x + pb)
  constexpr bool UP4 = EV == 5;
  const int tile_nw = tile_n;                               // N tile of the stacked weight matrix
  int pa = 0, pb = 0;
  if (UP4) { const int tn = a.tiles_n >> 2, ph = tile_n / tn; tile_n -= ph * tn; pa = ph >> 1; pb = ph & 1; }

  const int HWo = a.Hout * a.Wout;
  const int pad = (a.taps == 9) ? 1 : 0;
  const int Hs = a.upsample ? (a.Hin >> 1) : a.Hin;
  const int Ws = a.upsample ? (a.Win >> 1) : a.Win;
  const int lrow = t >> 3;                                   // loader rows are lrow + RPI*i
  const int lchunk = (t & 7) ^ ((t >> 4) & 7);               // logical chunk this lane fetches (same for every i)

  // per loader row: 64-bit sample base of each source (map lookups happen HERE, never in the loop),
  // top-left tap coordinates; in-sample offsets stay 32-bit.
  const T* base0[4]; const T* base1[4]; int iy0[4], ix0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = tile_m * BM + lrow + RPI * i;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    const int n = mm / HWo;
    const int rem = mm - n * HWo;
    const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
    const int n0 = a.map0 ? a.map0[n] : n;
    const int n1 = a.src1 ? (a.map1 ? a.map1[n] : n) : 0;
    base0[i] = reinterpret_cast<const T*>(a.src0) + (size_t)n0 * Hs * Ws * a.ld0;
    base1[i] = reinterpret_cast<const T*>(a.src1) + (size_t)n1 * Hs * Ws * a.ld1;
    iy0[i] = vm ? oy * a.stride - pad : -(1 << 20);
    ix0[i] = ox * a.stride - pad;
    // consume the map loads now: an ordinary load still pending inside the loop would make hipcc
    // drain the LDS-DMA queue (vmcnt(0)) at its first use there
    asm volatile("" ::"v"(base0[i]), "v"(base1[i]));
  }
  // weight rows enter LDS permuted (epi_wrow) so that the epilogue finds 8 consecutive couts per lane
  const T* wbase = reinterpret_cast<const T*>(a.W) + (size_t)(tile_nw * BN) * a.Ktot + lchunk * EPC;
  int wro[WL];                                                // element offset of the W row behind LDS row lrow + RPI*i
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    const int R = lrow + RPI * i;                             // LDS row: 128-row sub-tile R >> 7, permuted inside it
    wro[i] = ((R >> 7) * 128 + epi_wrow(R & 127, a.act == DC_ACT_GEGLU)) * a.Ktot;
  }
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + (t & 7) * 16;
  const int Hm1 = a.Hin - 1, Wm1 = a.Win - 1;

  const T* xrow0[4]; const T* xrow1[4];
  constexpr bool slim = SLIM;
  if (slim) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool vm = iy0[i] >= 0;
      xrow0[i] = vm ? base0[i] + (iy0[i] * Ws + ix0[i]) * a.ld0 + lchunk * EPC : nullptr;
      xrow1[i] = (vm && a.src1) ? base1[i] + (iy0[i] * Ws + ix0[i]) * a.ld1 + lchunk * EPC : nullptr;
    }
  }
  int itap = 0, icc = 0;   // (tap, channel chunk) of the next K-step to issue
  // tap geometry of the loader rows, refreshed when the issue stream moves to the next tap: source pixel index inside
  // the sample, -1 = padding (zero page).  (Recomputing it in every K-step cost ~20 VALU per row and step.)
  int pixo[4] = {-1, -1, -1, -1};
  auto set_tap = [&](int tap) {
    int ky = 0, kx = 0;
    if (UP4) { ky = (tap >> 1) + pa; kx = (tap & 1) + pb; }
    else if (a.taps == 9) { ky = tap / 3; kx = tap - ky * 3; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = iy0[i] + ky, ix = ix0[i] + kx;
      const bool ok = (unsigned)iy <= (unsigned)Hm1 && (unsigned)ix <= (unsigned)Wm1;
      const int sy = a.upsample ? (iy >> 1) : iy, sx = a.upsample ? (ix >> 1) : ix;
      pixo[i] = ok ? sy * Ws + sx : -1;
    }
  };
  if (!slim) set_tap(0);
  auto issue = [&](int ks) {
    const int st = ks % S;
    if (slim) {
      const bool s1 = icc >= a.c0chunks;
      const int coff = (s1 ? icc - a.c0chunks : icc) * BKE;
      char* xs = smem + st * STAGE + wave * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const T* rp = s1 ? xrow1[i] : xrow0[i];
        const char* gp = rp ? reinterpret_cast<const char*>(rp + coff) : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(xs + i * (NT * 16)), 16, 0, 0);
      }
      char* ws = smem + st * STAGE + XST + wave * 1024;
#pragma unroll
      for (int i = 0; i < WL; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t) reinterpret_cast<const char*>(wbase + wro[i] + ks * BKE), (lptr_t)(ws + i * (NT * 16)), 16, 0, 0);
      ++icc;
      return;
    }
    const bool s1 = icc >= a.c0chunks;
    const int ld = s1 ? a.ld1 : a.ld0;
    const int coff = (s1 ? icc - a.c0chunks : icc) * BKE + lchunk * EPC;
    char* xs = smem + st * STAGE + wave * 1024;              // wave-uniform LDS base; HW adds lane*16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* gp = reinterpret_cast<const char*>((s1 ? base1[i] : base0[i]) + (pixo[i] < 0 ? 0 : pixo[i]) * ld + coff);
      gp = pixo[i] < 0 ? zero : gp;
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(xs + i * (NT * 16)), 16, 0, 0);
    }
    char* ws = smem + st * STAGE + XST + wave * 1024;
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const char* gp = reinterpret_cast<const char*>(wbase + wro[i] + ks * BKE);
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(ws + i * (NT * 16)), 16, 0, 0);
    }
    if (++icc == a.cpt) { icc = 0; ++itap; set_tap(itap); }   // wave-uniform: once per tap, not per K-step
  };

  // number of 16-row pixel tiles of this wave that contain real rows
  const int rows_left = a.M - (tile_m * BM + wm * 64);
  const int jmax = __builtin_amdgcn_readfirstlane(rows_left <= 0 ? 0 : (rows_left >= 64 ? 4 : (rows_left + 15) >> 4));
  f32x4 acc[NH][TN][TM];                     // [64-cout half of the wave][cout fragment][pixel fragment]
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wrow0 = wn * 64 * NH;            // first LDS weight row of the wave
  int wfo[2], xfo[2];                        // stage-relative byte offsets of the wave's first W / X fragment, sub-steps 0 / 1
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    wfo[sub] = XST + lds_off(wrow0 + lr, sub * 4 + lq);
    xfo[sub] = lds_off(wm * 64 + lr, sub * 4 + lq);
  }

  issue(0);
  if (S == 3 && a.nk > 1) issue(1);
  for (int ks = 0; ks < a.nk; ++ks) {
    if (S == 3 && ks + 1 < a.nk) wait_vmcnt<LPS>();   // group ks+1 may stay in flight
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (ks + S - 1 < a.nk) issue(ks + S - 1);
    const char* Xs = smem + (ks % S) * STAGE;
    const char* Wsm = Xs + XST;
    if (jmax == TM && NH == 1) {  // the hot path: every pixel tile of the wave is real
      // all 16 fragment reads of the K-step go out first (asynchronous, common.h); the MFMAs follow in groups of
      // four behind counted lgkmcnt waits, so the later fragments arrive under the running matrix pipe
      // (fragment i / j is 16 rows = 2048 B further and keeps the row's swizzle: an immediate offset, one address register
      //  per operand and sub-step)
      const uint32_t stb = lds_addr_of(Xs);
      chunk16 xf[2][TM], wf[2][TN];
      lgkm_fence0();
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const uint32_t wa = stb + wfo[sub], xa = stb + xfo[sub];
        wf[sub][0] = ds_read16_async_off<0>(wa); wf[sub][1] = ds_read16_async_off<2048>(wa);
        wf[sub][2] = ds_read16_async_off<4096>(wa); wf[sub][3] = ds_read16_async_off<6144>(wa);
        xf[sub][0] = ds_read16_async_off<0>(xa); xf[sub][1] = ds_read16_async_off<2048>(xa);
        xf[sub][2] = ds_read16_async_off<4096>(xa); xf[sub][3] = ds_read16_async_off<6144>(xa);
      }
#define PIPE_MMA_GROUP(SUB, J, NLEFT)                                                     \
      lgkm_wait<NLEFT>(xf[SUB][J]);                                                         \
      _Pragma("unroll") for (int i = 0; i < TN; ++i) acc[0][i][J] = Mma<T>::run(wf[SUB][i], xf[SUB][J], acc[0][i][J]);  \
      __builtin_amdgcn_sched_barrier(0);
      lgkm_wait<11>(wf[0][0], wf[0][1], wf[0][2], wf[0][3], xf[0][0]);
      PIPE_MMA_GROUP(0, 0, 11) PIPE_MMA_GROUP(0, 1, 10) PIPE_MMA_GROUP(0, 2, 9) PIPE_MMA_GROUP(0, 3, 8)
      lgkm_wait<3>(wf[1][0], wf[1][1], wf[1][2], wf[1][3], xf[1][0]);
      PIPE_MMA_GROUP(1, 0, 3) PIPE_MMA_GROUP(1, 1, 2) PIPE_MMA_GROUP(1, 2, 1) PIPE_MMA_GROUP(1, 3, 0)
#undef PIPE_MMA_GROUP
    } else {                      // small-M side-path GEMMs: pixel tiles past M cost nothing (wave-uniform)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int c = sub * 4 + lq;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          chunk16 wf[TN];
#pragma unroll
          for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const chunk16*>(Wsm + lds_off(wrow0 + h * 64 + i * 16 + lr, c));
#pragma unroll
          for (int j = 0; j < TM; ++j)
            if (j < jmax) {
              const chunk16 x1 = *reinterpret_cast<const chunk16*>(Xs + lds_off(wm * 64 + j * 16 + lr, c));
#pragma unroll
              for (int i = 0; i < TN; ++i) acc[h][i][j] = Mma<T>::run(wf[i], x1, acc[h][i][j]);
            }
        }
      }
    }
  }
  // ---- epilogue: straight from the accumulators (igemm_epilogue.h), no LDS, no barrier ----
  const int mw0 = tile_m * BM + wm * 64;
  auto rowfn = [&](int j, EpiRow& r) {
    const int m = tile_m * BM + wm * 64 + j * 16 + lr;
    r.ok = m < a.M;
    const int mm = r.ok ? m : a.M - 1;
    const int n = mm / HWo;
    r.samp = n;
    r.o = mm;
    if (UP4) {
      const int rem = mm - n * HWo, y = rem / a.Wout, x = rem - y * a.Wout;
      r.o = n * (4 * HWo) + (2 * y + pa) * (2 * a.Wout) + 2 * x + pb;
    }
    r.r = (a.residual && a.res_map) ? a.res_map[n] * HWo + (mm - n * HWo) : mm;
  };
  const int sf = min(mw0, a.M - 1) / HWo, sl = min(mw0 + 63, a.M - 1) / HWo;
  // one 64-cout half of the wave at a time (written out, not looped: a loop index would demote acc[] to scratch):
  // 128-cout tile index and half inside it
  auto epi = [&](f32x4 (&ac)[TN][TM], int tile128, int half) {
    if (EV < 0) epi_direct<T, TM>(a, ac, tile128, half, lq, sf, sl, rowfn);
    else epi_direct_act<T, TM, (EV == 3 ? DC_ACT_GELU_TANH : (EV == 2 ? DC_ACT_GEGLU : DC_ACT_NONE)), EV == 4, false>(a, ac, tile128, half, lq, sf, sl, rowfn);   // EV 0 / 5: plain
  };
  epi(acc[0], tile_n, wn);
}

template <typename T, int BM, int S, int NH, int EV, bool SLIM>
static int launch_pipe(const IgemmArgs& a0, hipStream_t s) {
  constexpr int lds = S * (BM + 128 * NH) * 128;                   // 144 KiB (256x128, S 3) / 64 KiB (128x128, S 2) / 128 KiB (256x256, S 2)
  static bool attr_done = false;
  auto kern = igemm_pipe_kernel<T, BM, S, NH, EV, SLIM>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  IgemmArgs a = a0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = a0.tiles_n / NH;                                     // a0.tiles_n counts 128-cout tiles
  const long long nblk = (long long)a.tiles_m * a.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { dc_set_error("dc_igemm: bad grid %lld", nblk); return DC_ERR_SHAPE; }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(BM * 2), lds, s, a);
  return dc_check_launch("dc_igemm(pipe)");
}

// 0: 128x128 / 2 stages, 1: 256x128 / 3 stages, 2: 256x256 (igemm_wide.hip)
int dc_igemm_pipe_shape(const IgemmArgs& a) {
  constexpr int light_nk = 8;
  static const bool no_wide = getenv("DCAMD_PIPE_NO_WIDE") != nullptr;
  constexpr long long wide_min_tiles = 400;
  constexpr int wide_min_nk = 4;
  const bool wide_act = a.act == DC_ACT_NONE || ((a.act == DC_ACT_GELU_TANH || a.act == DC_ACT_GEGLU) && !a.gate);   // the variants igemm_wide.hip has
  // 1-tap GEMMs on wide layers: the 256x256 tile whenever it fills the chip (>= wide_min_tiles tiles), whatever K
  if (!no_wide && wide_act && a.taps == 1 && (a.tiles_n & 1) == 0 && a.nk >= wide_min_nk &&
      (long long)((a.M + 255) / 256) * (a.tiles_n / 2) >= wide_min_tiles) return 2;
  if (a.nk <= light_nk) return 0;
  // a grid of 256-row tiles that does not give every CU a workgroup (the fp32 side-path GEMMs: M = work units, e.g. DiT-B/4's adaLN
  // modulation, 2 x 36 tiles): the 128-row tile, two workgroups per CU — 5.75 -> 3.41 ms per cfg5 step for those 50 launches
  const char* ct = getenv("DCAMD_PIPE_CHIP_TILES");          // read per call (~0.1 us): the tests pin the 256-row tile on small shapes with 0
  const long long chip_tiles = ct ? atoll(ct) : 256;
  if ((long long)((a.M + 255) / 256) * a.tiles_n < chip_tiles) return 0;
  return 1;
}

// four-phase upsample conv on the tap-gather kernel (sources smaller than 8x8): a0 as dc_igemm received it (upsampled extents)
int dc_igemm_launch_pipe_up4(const IgemmArgs& a0, int dtype, hipStream_t s) {
  IgemmArgs a = a0;
  a.upsample = 0; a.Hin = a0.Hin >> 1; a.Win = a0.Win >> 1; a.Hout = a.Hin; a.Wout = a.Win;
  a.M = (int)((long long)a0.M >> 2);
  a.Ktot = 4 * (a.C0 + a.C1);
  a.nk = 4 * a.cpt;
  a.tiles_n = 4 * a0.tiles_n;
  a.n_fast = 0;
  if (dtype == DC_BF16) return launch_pipe<__bf16, 256, 3, 1, 5, false>(a, s);
  if (dtype == DC_F16) return launch_pipe<_Float16, 256, 3, 1, 5, false>(a, s);
  return launch_pipe<float, 256, 3, 1, 5, false>(a, s);
}

int dc_igemm_launch_pipe(const IgemmArgs& a, int dtype, hipStream_t s) {
  const int shape = dc_igemm_pipe_shape(a);
  const bool slim = a.taps == 1;
  if (shape == 0) {
    if (slim) {
      if (dtype == DC_BF16) return launch_pipe<__bf16, 128, 2, 1, -1, true>(a, s);
      if (dtype == DC_F16) return launch_pipe<_Float16, 128, 2, 1, -1, true>(a, s);
      return launch_pipe<float, 128, 2, 1, -1, true>(a, s);
    }
    if (dtype == DC_BF16) return launch_pipe<__bf16, 128, 2, 1, -1, false>(a, s);
    if (dtype == DC_F16) return launch_pipe<_Float16, 128, 2, 1, -1, false>(a, s);
    return launch_pipe<float, 128, 2, 1, -1, false>(a, s);
  }
  if (shape == 2) return dc_igemm_launch_wide8(a, dtype, s);     // the 256 x 256 tile lives in igemm_wide.hip (8-phase loop)
  if (slim) {
    if (dtype == DC_BF16) return launch_pipe<__bf16, 256, 3, 1, -1, true>(a, s);
    if (dtype == DC_F16) return launch_pipe<_Float16, 256, 3, 1, -1, true>(a, s);
    return launch_pipe<float, 256, 3, 1, -1, true>(a, s);
  }
  if (dtype == DC_BF16) return launch_pipe<__bf16, 256, 3, 1, -1, false>(a, s);
  if (dtype == DC_F16) return launch_pipe<_Float16, 256, 3, 1, -1, false>(a, s);
  return launch_pipe<float, 256, 3, 1, -1, false>(a, s);
}
