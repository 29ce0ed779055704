// igemm_wide.hip — the 256 x 256 tile of the 1-tap GEMMs (Linear / Conv2d 1x1 on wide layers: every DiT-B/4 linear behind
// reference nets/dit.py:49-51, the q/k/v, attn_out, ff_out, proj_in/out GEMMs of the UNets' transformer blocks behind
// nets/unet.py:186-195) on an 8-phase main loop: 8 waves (4 along pixels x 2 along couts, wave tile 64 pixels x 128 couts =
// 128 accumulator registers), K-tile = 128 bytes per row (BK = 64 for 16-bit types), the whole 160 KiB-class LDS budget as
// 2 K-tiles x 4 HALF-TILES of 16 KiB, operands brought in by LDS-DMA only, hand-counted vmcnt (never 0 inside the loop).
//
// Why not the 2-stage loop this replaces (igemm_pipe_kernel<.., 256, 2, 2, ..>): that loop waited vmcnt(0) + one barrier per
// K-tile, with both waves of a SIMD reading fragments at the same moment and then sharing the matrix pipe at the same moment:
// 0.72 PF at K = 768 and 0.90 PF at K = 3072 on DiT-B/4, `SQ_WAIT_ANY` 0.47.  Here (cdna_hip_programming.md section 5, "The
// 256^2 8-phase template"; the example file itself is not on this machine, the schedule below is derived from its rules):
//
//  * a K-tile is consumed in FOUR phases, one accumulator quadrant (32 pixels x 64 couts x K 64 = 16 MFMAs per wave) each:
//      q0: W half 0 x X half 0   reads X0 (4 x ds_read_b128), then W0 (8)
//      q1: W half 0 x X half 1   reads X1 (4)
//      q2: W half 1 x X half 1   reads W1 (8) into W0's registers
//      q3: W half 1 x X half 0   reads nothing (X0 is still in registers)
//    so 64 fragment registers serve 128 accumulators, and every half-tile (X0 / W0 / X1 / W1: the 32-pixel or 64-cout half of
//    EVERY wave's tile, 128 rows x 128 B) is read in exactly one phase;
//  * every phase is  { fragment reads ; ONE half-tile of LDS-DMA (2 per lane) ; s_barrier ; lgkmcnt(0) ; 16 MFMAs ; s_barrier };
//  * the two wave groups (waves 0-3 / 4-7: the two waves of every SIMD belong to different groups) run ONE BARRIER APART, so
//    that while one wave of a SIMD issues its 16 MFMAs the other reads fragments and issues its LDS-DMA: the matrix pipe never
//    waits for an LDS read of its own SIMD's other wave;
//  * the LDS-DMA stream runs 7 half-tiles ahead of the phase that issues it: stream item (K-tile t, half h), order X0 W0 X1 W1,
//    is issued in phase 4 t + h - 7.  One counted wait per K-tile, in q3: vmcnt(6) — the three youngest half-tiles (2 LDS-DMA
//    per lane each) stay in flight across it and across every barrier; what it retires (all of K-tile t + 1) is first read one
//    phase later.
//
// Hazards, with the groups one barrier apart (barrier instance n: group A's n-th = group B's n-th arrival; A runs phase p's
// MFMAs between instances 2p-1 and 2p, B between 2p and 2p+1; reads of phase p: A between 2p-2 and 2p-1, B between 2p-1 and 2p):
//   RAW (LDS-DMA -> ds_read): every wave's vmcnt(6) of phase p sits before its first barrier of that phase (instance <= 2p);
//     the retired data is first read in phase p + 1, i.e. after instance 2p.
//   WAR (ds_read -> LDS-DMA into the same slot): W0 is re-staged 2 phases after its reads, X1 2, W1 2 — the readers' lgkmcnt(0)
//     sits before instance 2p+1, the re-stage comes after instance 2p+2.  X0 is re-staged ONE phase after its reads (q0 -> q1):
//     its four reads are issued FIRST in q0 and retired by `s_waitcnt lgkmcnt(8)` before the wave's first barrier of q0
//     (instance <= 2p), and the re-stage in q1 comes after instance 2p.
// Group A executes one balancing barrier after the loop (B executed one more in front of it).
//
// LDS image of a half-tile: 128 rows of 128 B, 16-byte chunks XOR-swizzled by (row >> 1) & 7 (igemm_common.h lds_off) on the
// LDS-DMA SOURCE address and again on the fragment reads: conflict-free ds_read_b128.  Rows past M fetch a zero page.
// Epilogue: igemm_epilogue.h, straight from the accumulators (the weight rows enter LDS permuted).
#include <stdlib.h>
#include "igemm_epilogue.h"

static __device__ chunk16 g_zero_page_w[16];
DC_CLOCK_DECL(igemm_wide)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// EV: epilogue variant compiled in (one per instantiation: several in one kernel pushed hipcc into spilling accumulators):
// 0 = plain (bias / row vector / residual), 2 = GEGLU, 3 = tanh-GELU, 4 = plain + per-sample gate (DiT adaLN-Zero).
template <typename T, int EV>
__global__ __launch_bounds__(512, 2) void igemm_wide8_kernel(const IgemmArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 8 * EPC;                 // elements per 128-byte K-tile row
  constexpr int HALF = 128 * 128;              // bytes per half-tile
  constexpr int BUF = 4 * HALF;                // one K-tile: X0 | W0 | X1 | W1
  constexpr int SX0 = 0, SW0 = HALF, SX1 = 2 * HALF, SW1 = 3 * HALF;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  int tile_m, tile_n;
  tile_of_block(a, tile_m, tile_n);
  const int HWo = a.Hout * a.Wout;
  const int nk = a.nk, c0chunks = a.c0chunks;

  // ---- loader: lane t owns pieces i = 0, 1 of every half-tile: half-tile row hr = 64 i + (t >> 3), physical chunk t & 7 ----
  const int lchunk = (t & 7) ^ ((t >> 4) & 7);             // logical 16-byte chunk of the row this lane fetches (source-side swizzle)
  const T* xr0[2][2]; const T* xr1[2][2];                  // [X half][piece]: row pointers into source 0 / 1 (nullptr: row past M)
  int wro[2][2];                                           // [W half][piece]: element offset of the packed weight row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int hr = 64 * i + (t >> 3);
#pragma unroll
    for (int xh = 0; xh < 2; ++xh) {
      const int m = tile_m * 256 + (hr >> 5) * 64 + xh * 32 + (hr & 31);      // half-tile row hr = wave row wm = hr >> 5, pixel hr & 31 of half xh
      const bool vm = m < a.M;
      const int mm = vm ? m : 0;
      const int n = mm / HWo;
      const int rem = mm - n * HWo;
      const int n0 = a.map0 ? a.map0[n] : n;
      const int n1 = a.src1 ? (a.map1 ? a.map1[n] : n) : 0;
      const T* b0 = reinterpret_cast<const T*>(a.src0) + ((size_t)n0 * HWo + rem) * a.ld0 + lchunk * EPC;
      const T* b1 = reinterpret_cast<const T*>(a.src1) + ((size_t)n1 * HWo + rem) * a.ld1 + lchunk * EPC;
      xr0[xh][i] = vm ? b0 : nullptr;
      xr1[xh][i] = (vm && a.src1) ? b1 : nullptr;
      // consume the map loads now: an ordinary load still pending inside the loop would make hipcc drain the LDS-DMA queue there
      asm volatile("" ::"v"(xr0[xh][i]), "v"(xr1[xh][i]));
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int R = (hr >> 6) * 128 + h * 64 + (hr & 63);  // row of the 256-cout tile: wave column wn = hr >> 6, half h
      wro[h][i] = ((R >> 7) * 128 + epi_wrow(R & 127, a.act == DC_ACT_GEGLU)) * a.Ktot;
    }
  }
  const T* wbase = reinterpret_cast<const T*>(a.W) + (size_t)(tile_n * 256) * a.Ktot + lchunk * EPC;
  const char* zero = reinterpret_cast<const char*>(g_zero_page_w) + (t & 7) * 16;
  char* const ldst = smem + wave * 1024;                   // wave-uniform LDS-DMA base (the hardware adds lane * 16)

  // stream item (K-tile kt, half-tile slot): 2 LDS-DMA per lane; nothing past the last K-tile (the counted waits know)
  auto stage_x = [&](int kt, int xh, int slot) {
    if (kt >= nk) return;
    const bool s1 = kt >= c0chunks;
    const int coff = (s1 ? kt - c0chunks : kt) * BKE;
    char* dst = ldst + (kt & 1) * BUF + slot;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const T* rp = s1 ? xr1[xh][i] : xr0[xh][i];
      const char* gp = rp ? reinterpret_cast<const char*>(rp + coff) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(dst + i * 8192), 16, 0, 0);
    }
  };
  auto stage_w = [&](int kt, int h, int slot) {
    if (kt >= nk) return;
    char* dst = ldst + (kt & 1) * BUF + slot;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t) reinterpret_cast<const char*>(wbase + wro[h][i] + kt * BKE), (lptr_t)(dst + i * 8192), 16, 0, 0);
  };

  f32x4 acc[2][4][4];                          // [W half][cout fragment][pixel fragment]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment read addresses inside a half-tile, k sub-steps 0 / 1 (fragment i / j is 16 rows = 2048 B further: an immediate)
  const uint32_t lds0 = lds_addr_of(smem);
  uint32_t xfo[2], wfo[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    xfo[sub] = lds0 + lds_off(wm * 32 + lr, sub * 4 + lq);
    wfo[sub] = lds0 + lds_off(wn * 64 + lr, sub * 4 + lq);
  }

  // ---- prologue: stream items 0 .. 6 (K-tile 0 whole, X0 W0 X1 of K-tile 1), K-tile 0 landed, one barrier; group B one more ----
  stage_x(0, 0, SX0); stage_w(0, 0, SW0); stage_x(0, 1, SX1); stage_w(0, 1, SW1);
  stage_x(1, 0, SX0); stage_w(1, 0, SW0); stage_x(1, 1, SX1);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wave >= 4) __builtin_amdgcn_s_barrier();

  chunk16 wf[2][4], xa[2][2], xb[2][2];        // W half in flight, X half 0, X half 1  ([k sub-step][fragment])
#define W8_MMA(H, XF, J0)                                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                              \
  _Pragma("unroll") for (int sub = 0; sub < 2; ++sub)                                                         \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
        acc[H][i][J0 + jj] = Mma<T>::run(wf[sub][i], XF[sub][jj], acc[H][i][J0 + jj]);                        \
  __builtin_amdgcn_s_setprio(0);                                                                              \
  __builtin_amdgcn_sched_barrier(0);
#define W8_READ_X(DST, SLOT)                                                                                \
  _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                       \
    DST[sub][0] = ds_read16_async_off<SLOT>(xfo[sub] + bufo);                                                 \
    DST[sub][1] = ds_read16_async_off<SLOT + 2048>(xfo[sub] + bufo);                                          \
  }
#define W8_READ_W(SLOT)                                                                                     \
  _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                       \
    wf[sub][0] = ds_read16_async_off<SLOT>(wfo[sub] + bufo);                                                  \
    wf[sub][1] = ds_read16_async_off<SLOT + 2048>(wfo[sub] + bufo);                                           \
    wf[sub][2] = ds_read16_async_off<SLOT + 4096>(wfo[sub] + bufo);                                           \
    wf[sub][3] = ds_read16_async_off<SLOT + 6144>(wfo[sub] + bufo);                                           \
  }
  // all outstanding LDS reads are back (tied to the registers they fill, so no MFMA is scheduled above the wait)
#define W8_WAIT_W() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(wf[0][3]), \
                                 "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]), "+v"(wf[1][3]))
#define W8_WAIT_X(XF) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XF[0][0]), "+v"(XF[0][1]), "+v"(XF[1][0]), "+v"(XF[1][1]))

  DC_CLOCK(0);
  for (int kt = 0; kt < nk; ++kt) {
    const uint32_t bufo = (uint32_t)(kt & 1) * BUF;
    // ---- q0: W0 x X0.  X0's four reads go first and are retired before the barrier: X0's slot is re-staged in q1 ----
    lgkm_fence0();
    W8_READ_X(xa, SX0)
    __builtin_amdgcn_sched_barrier(0);
    W8_READ_W(SW0)
    stage_w(kt + 1, 1, SW1);
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[1][0]), "+v"(xa[1][1]));
    __builtin_amdgcn_s_barrier();
    W8_WAIT_W();
    W8_MMA(0, xa, 0)
    __builtin_amdgcn_s_barrier();
    // ---- q1: W0 x X1 ----
    W8_READ_X(xb, SX1)
    stage_x(kt + 2, 0, SX0);
    __builtin_amdgcn_s_barrier();
    W8_WAIT_X(xb);
    W8_MMA(0, xb, 2)
    __builtin_amdgcn_s_barrier();
    // ---- q2: W1 x X1 ----
    W8_READ_W(SW1)
    stage_w(kt + 2, 0, SW0);
    __builtin_amdgcn_s_barrier();
    W8_WAIT_W();
    W8_MMA(1, xb, 2)
    __builtin_amdgcn_s_barrier();
    // ---- q3: W1 x X0 (registers only).  The K-tile's one counted wait: everything but the three youngest half-tiles
    //      (K-tile kt + 2's X0 W0 X1) has landed, i.e. all of K-tile kt + 1, which is first read in the next phase ----
    stage_x(kt + 2, 1, SX1);
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    W8_MMA(1, xa, 0)
    __builtin_amdgcn_s_barrier();
  }
#undef W8_MMA
#undef W8_READ_X
#undef W8_READ_W
#undef W8_WAIT_W
#undef W8_WAIT_X
  DC_CLOCK(1);
  if (wave < 4) __builtin_amdgcn_s_barrier();          // group B arrived once more in front of the loop

  // ---- epilogue: straight from the accumulators (igemm_epilogue.h), no LDS, no barrier ----
  const int mw0 = tile_m * 256 + wm * 64;
  auto rowfn = [&](int j, EpiRow& r) {
    const int m = mw0 + j * 16 + lr;
    r.ok = m < a.M;
    const int mm = r.ok ? m : a.M - 1;
    const int n = mm / HWo;
    r.samp = n;
    r.o = mm;
    r.r = (a.residual && a.res_map) ? a.res_map[n] * HWo + (mm - n * HWo) : mm;
  };
  const int sf = min(mw0, a.M - 1) / HWo, sl = min(mw0 + 63, a.M - 1) / HWo;
  constexpr int ACT = EV == 3 ? DC_ACT_GELU_TANH : (EV == 2 ? DC_ACT_GEGLU : DC_ACT_NONE);
  epi_direct_act<T, 4, ACT, EV == 4, false>(a, acc[0], tile_n * 2 + wn, 0, lq, sf, sl, rowfn);
  __builtin_amdgcn_sched_barrier(0);         // one epilogue's registers at a time
  epi_direct_act<T, 4, ACT, EV == 4, false>(a, acc[1], tile_n * 2 + wn, 1, lq, sf, sl, rowfn);
}

template <typename T, int EV>
static int launch_wide8(const IgemmArgs& a0, hipStream_t s) {
  constexpr int lds = 2 * 4 * 128 * 128;                           // 128 KiB: one workgroup per CU
  static bool attr_done = false;
  auto kern = igemm_wide8_kernel<T, EV>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  IgemmArgs a = a0;
  a.tiles_m = (a.M + 255) / 256;
  a.tiles_n = a0.tiles_n / 2;                                      // a0.tiles_n counts 128-cout tiles
  const long long nblk = (long long)a.tiles_m * a.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { dc_set_error("dc_igemm: bad grid %lld", nblk); return DC_ERR_SHAPE; }
  if ((long long)a0.tiles_n * 128 * a.Ktot >= (1LL << 31)) { dc_set_error("dc_igemm(wide8): weight matrix of %d x %d too large", a0.tiles_n * 128, a.Ktot); return DC_ERR_SHAPE; }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), lds, s, a);
  return dc_check_launch("dc_igemm(wide8)");
}

template <typename T>
static int launch_wide8_t(const IgemmArgs& a, hipStream_t s) {
  if (a.act == DC_ACT_GELU_TANH) return launch_wide8<T, 3>(a, s);
  if (a.act == DC_ACT_GEGLU) return launch_wide8<T, 2>(a, s);
  if (a.gate) return launch_wide8<T, 4>(a, s);
  return launch_wide8<T, 0>(a, s);
}

// 1-tap GEMM on the 256 x 256 tile (dc_igemm_pipe_shape(a) == 2 decides; same epilogue variants as the 2-stage loop it replaces)
int dc_igemm_launch_wide8(const IgemmArgs& a, int dtype, hipStream_t s) {
  if (dtype == DC_BF16) return launch_wide8_t<__bf16>(a, s);
  if (dtype == DC_F16) return launch_wide8_t<_Float16>(a, s);
  return launch_wide8_t<float>(a, s);
}
