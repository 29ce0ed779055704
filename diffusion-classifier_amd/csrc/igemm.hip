// igemm.hip — implicit-GEMM convolution / GEMM on MFMA for gfx950.
//
// Replaces every Conv2d(3x3|1x1) and Linear of the diffusers backbone the reference calls
// through nets/unet.py:186-195 and nets/dit.py:49-51 (cuDNN/cuBLAS there).
//
// Data layout: activations NHWC (rows = sample*H*W pixels, channels contiguous); weights
// packed [Cout_pad][K], K contiguous, k = tap*(C0+C1) + c.  Both MFMA operands are therefore
// K-contiguous and every global / LDS transfer is a 16-byte chunk.
//
// Tile: BM pixels x BN output channels per 256-thread workgroup (4 waves), K-step = 128 bytes
// per row (64 bf16/f16 or 32 f32 elements), LDS double-buffered, register-staged prefetch of
// the next K-step while the current one is on the matrix cores.  LDS rows are 128 B with the
// 16-B chunk index XOR-swizzled by (row>>1)&7, which makes the ds_read_b128 fragment reads of
// a 16-row MFMA operand conflict-free (MI355X LDS: 64 banks x 4 B, b128 served per 16 lanes).
//
// MFMA orientation: A-operand = weight rows (cout), B-operand = pixel rows, so the f32
// accumulator of a lane is 4 CONSECUTIVE output channels of ONE pixel -> 8/16-byte NHWC stores.
//   bf16/f16: v_mfma_f32_16x16x32_{bf16,f16};  f32: v_mfma_f32_16x16x4_f32 (exact f32 fma chain).
//
// Block->tile map is XCD-aware: the 8 XCDs (private 4 MiB L2 each) receive contiguous ranges of
// the tile list, ordered M-fastest inside one N panel, so the workgroups sharing an L2 stream the
// same weight panel.
#include <stdio.h>
#include <stdlib.h>
#include "igemm_common.h"

template <typename T, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 8 * EPC;            // elements per K-step
  constexpr int XL = BM / 32;             // 16-B X loads per thread per K-step
  constexpr int WL = (BN + 31) / 32;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  static_assert(WGM * WGN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUF = (BM + BN) * 128;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = lane & 15, lq = lane >> 4;

  int tile_m, tile_n;
  tile_of_block(a, tile_m, tile_n);

  const int HWo = a.Hout * a.Wout;
  const int pad = (a.taps == 9) ? 1 : 0;
  const int Hs = a.upsample ? (a.Hin >> 1) : a.Hin;
  const int Ws = a.upsample ? (a.Win >> 1) : a.Win;
  const int chunk = t & 7, lrow = t >> 3;

  int ns0[XL], ns1[XL], iy0[XL], ix0[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int m = tile_m * BM + lrow + 32 * i;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    const int n = mm / HWo;
    const int rem = mm - n * HWo;
    const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
    ns0[i] = a.map0 ? a.map0[n] : n;
    ns1[i] = a.src1 ? (a.map1 ? a.map1[n] : n) : 0;
    iy0[i] = vm ? oy * a.stride - pad : -(1 << 20);
    ix0[i] = ox * a.stride - pad;
  }
  const T* wbase = reinterpret_cast<const T*>(a.W) + (size_t)(tile_n * BN + lrow) * a.Ktot + chunk * EPC;

  chunk16 xr[XL], wr[WL];
  const chunk16 zero = {0u, 0u, 0u, 0u};

  auto load_regs = [&](int ks, int tap, int cc) {
    int ky = 0, kx = 0;
    if (a.taps == 9) { ky = tap / 3; kx = tap - ky * 3; }
    const bool s1 = cc >= a.c0chunks;
    const T* src = reinterpret_cast<const T*>(s1 ? a.src1 : a.src0);
    const int ld = s1 ? a.ld1 : a.ld0;
    const int coff = (s1 ? cc - a.c0chunks : cc) * BKE + chunk * EPC;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int iy = iy0[i] + ky, ix = ix0[i] + kx;
      const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const int sy = a.upsample ? (iy >> 1) : iy, sx = a.upsample ? (ix >> 1) : ix;
      const size_t pix = ((size_t)(s1 ? ns1[i] : ns0[i]) * Hs + sy) * Ws + sx;
      xr[i] = zero;
      if (ok) xr[i] = *reinterpret_cast<const chunk16*>(src + pix * ld + coff);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      if (BN >= 32 * (i + 1) || lrow + 32 * i < BN)
        wr[i] = *reinterpret_cast<const chunk16*>(wbase + (size_t)(32 * i) * a.Ktot + (size_t)ks * BKE);
    }
  };
  auto write_lds = [&](int buf) {
    char* Xs = smem + buf * BUF;
    char* Wsm = Xs + BM * 128;
#pragma unroll
    for (int i = 0; i < XL; ++i) *reinterpret_cast<chunk16*>(Xs + lds_off(lrow + 32 * i, chunk)) = xr[i];
#pragma unroll
    for (int i = 0; i < WL; ++i)
      if (BN >= 32 * (i + 1) || lrow + 32 * i < BN)
        *reinterpret_cast<chunk16*>(Wsm + lds_off(lrow + 32 * i, chunk)) = wr[i];
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tap = 0, cc = 0;
  load_regs(0, 0, 0);
  write_lds(0);
  __syncthreads();
  for (int ks = 0; ks < a.nk; ++ks) {
    const bool more = ks + 1 < a.nk;
    if (more) {
      if (++cc == a.cpt) { cc = 0; ++tap; }
      load_regs(ks + 1, tap, cc);
    }
    const char* Xs = smem + (ks & 1) * BUF;
    const char* Wsm = Xs + BM * 128;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int c = sub * 4 + lq;
      chunk16 xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const chunk16*>(Xs + lds_off(wm * WTM + j * 16 + lr, c));
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const chunk16*>(Wsm + lds_off(wn * WTN + i * 16 + lr, c));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::run(wf[i], xf[j], acc[i][j]);
    }
    if (more) write_lds((ks + 1) & 1);
    __syncthreads();
  }

  igemm_epilogue<TM, TN>(a, acc, tile_m * BM + wm * WTM, tile_n * BN + wn * WTN, lr, lq);
}

template <typename T, int BM, int BN, int WGM, int WGN>
static int launch(const IgemmArgs& a, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * 128;
  static bool attr_done = false;
  auto kern = igemm_kernel<T, BM, BN, WGM, WGN>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const long long nblk = (long long)a.tiles_m * a.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { dc_set_error("dc_igemm: bad grid %lld", nblk); return DC_ERR_SHAPE; }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, s, a);
  return dc_check_launch("dc_igemm");
}

extern "C" int32_t dc_igemm_cout_pad(int32_t cout, int32_t tile_n) {
  const int tn = tile_n == 32 ? 32 : 128;
  return (cout + tn - 1) / tn * tn;
}

static int igemm_run(const dc_igemm_params* p, dc_stream stream, const char** variant);

extern "C" int dc_igemm(const dc_igemm_params* p, dc_stream stream) { return igemm_run(p, stream, nullptr); }

extern "C" int32_t dc_igemm_gn_fusable(const dc_igemm_params* p) {
  if (!p) return 0;
  dc_igemm_params q = *p;
  static const float dummy = 0.f;
  q.gn_scale = &dummy; q.gn_shift = &dummy;
  const char* v = nullptr;
  return igemm_run(&q, nullptr, &v) == DC_OK ? 1 : 0;
}

extern "C" int32_t dc_igemm_ln_ok(const dc_igemm_params* p) {
  if (!p) return 0;
  dc_igemm_params q = *p;
  if (!(q.ln_eps > 0.f)) q.ln_eps = 1e-5f;
  const char* v = nullptr;
  return igemm_run(&q, nullptr, &v) == DC_OK ? 1 : 0;
}

extern "C" int32_t dc_igemm_qstats_parts(const dc_igemm_params* p) {
  if (!p) return 0;
  dc_igemm_params q = *p;
  alignas(16) static float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  if (!q.qstats) q.qstats = dummy;
  const char* v = nullptr;
  if (igemm_run(&q, nullptr, &v) != DC_OK) return 0;
  const int hw = p->Hout * p->Wout;
  if (p->up4) { const int lo = hw >> 2; return 4 * (lo >= 128 ? lo / 128 : 1); }     // per phase, on the low-resolution image
  return hw >= 128 ? hw / 128 : 1;          // one part per wave-sized run of 128 pixels (64-pixel images: one)
}

extern "C" int32_t dc_igemm_pn_ok(const dc_igemm_params* p) {
  if (!p || p->pn_groups <= 0) return 0;
  dc_igemm_params q = *p;
  alignas(16) static float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  if (!q.pn_out) q.pn_out = dummy;
  if (!q.qstats && !(q.Hin == 4 && q.Win == 4)) q.qstats = dummy;      // (4x4 mosaic patches form no quad records: their GroupNorm needs none)
  if (!q.pn_gamma) q.pn_gamma = dummy;
  if (!q.pn_beta) q.pn_beta = dummy;
  if (!q.pn_cnt) q.pn_cnt = reinterpret_cast<uint32_t*>(dummy);
  if (!(q.pn_eps > 0.f)) q.pn_eps = 1e-5f;
  const char* v = nullptr;
  return igemm_run(&q, nullptr, &v) == DC_OK ? 1 : 0;
}

extern "C" int32_t dc_pn_timeouts(void) { return (int32_t)dc_conv3_halo_pn_timeouts(); }

extern "C" int32_t dc_igemm_up4_ok(const dc_igemm_params* p) {
  if (!p) return 0;
  dc_igemm_params q = *p;
  q.up4 = 1;
  const char* v = nullptr;
  return igemm_run(&q, nullptr, &v) == DC_OK ? 1 : 0;
}

extern "C" int32_t dc_igemm_side_ok(const dc_igemm_params* p) {
  if (!p) return 0;
  dc_igemm_params q = *p;
  alignas(16) static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  if (!q.src2) { q.src2 = dummy; q.W2 = dummy; }
  const char* v = nullptr;
  return igemm_run(&q, nullptr, &v) == DC_OK ? 1 : 0;
}

extern "C" const char* dc_igemm_variant(const dc_igemm_params* p) {
  const char* v = "invalid";
  (void)igemm_run(p, nullptr, &v);
  return v;
}

// variant != nullptr: dry run — validate, pick the kernel, report its name, launch nothing.
static int igemm_run(const dc_igemm_params* p, dc_stream stream, const char** variant) {
  DC_REQUIRE(p, DC_ERR_ARG, "dc_igemm: null params");
  DC_REQUIRE(p->dtype == DC_F32 || p->dtype == DC_BF16 || p->dtype == DC_F16, DC_ERR_DTYPE, "dc_igemm: dtype %d", p->dtype);
  DC_REQUIRE(p->taps == 1 || p->taps == 9, DC_ERR_ARG, "dc_igemm: taps must be 1 or 9 (got %d)", p->taps);
  DC_REQUIRE(p->stride == 1 || p->stride == 2, DC_ERR_ARG, "dc_igemm: stride %d", p->stride);
  DC_REQUIRE(p->tile_n == 128 || p->tile_n == 32, DC_ERR_ARG, "dc_igemm: tile_n %d", p->tile_n);
  DC_REQUIRE(p->src0 && p->W && (p->out || p->pn_out), DC_ERR_ARG, "dc_igemm: null src0/W/out");
  const int bke = 128 / dc_dtype_size(p->dtype);
  DC_REQUIRE(p->C0 > 0 && p->C0 % bke == 0, DC_ERR_SHAPE, "dc_igemm: C0=%d not a multiple of %d", p->C0, bke);
  DC_REQUIRE(p->C1 >= 0 && p->C1 % bke == 0 && ((p->C1 > 0) == (p->src1 != nullptr)), DC_ERR_SHAPE,
             "dc_igemm: C1=%d / src1 mismatch or not a multiple of %d", p->C1, bke);
  DC_REQUIRE(p->n_img > 0 && p->Hout > 0 && p->Wout > 0 && p->Hin > 0 && p->Win > 0 && p->Cout > 0, DC_ERR_SHAPE,
             "dc_igemm: non-positive extent");
  if (p->taps == 9) {
    DC_REQUIRE(p->Hout == (p->Hin + 2 - 3) / p->stride + 1 && p->Wout == (p->Win + 2 - 3) / p->stride + 1, DC_ERR_SHAPE,
               "dc_igemm: 3x3 output %dx%d does not match input %dx%d stride %d", p->Hout, p->Wout, p->Hin, p->Win, p->stride);
  } else {
    DC_REQUIRE(p->stride == 1 && !p->upsample && p->Hout == p->Hin && p->Wout == p->Win, DC_ERR_SHAPE,
               "dc_igemm: taps=1 needs stride 1, no upsample, equal in/out extents");
  }
  if (p->upsample) DC_REQUIRE((p->Hin % 2 == 0) && (p->Win % 2 == 0), DC_ERR_SHAPE, "dc_igemm: upsample needs even Hin/Win");
  const long long M = (long long)p->n_img * p->Hout * p->Wout;
  DC_REQUIRE(M < (1LL << 31), DC_ERR_SHAPE, "dc_igemm: M=%lld too large", M);
  DC_REQUIRE(p->act >= 0 && p->act <= 3, DC_ERR_ARG, "dc_igemm: act %d", p->act);
  if (p->act == DC_ACT_GEGLU) DC_REQUIRE(p->Cout % 32 == 0 && p->tile_n == 128, DC_ERR_SHAPE, "dc_igemm: GEGLU needs Cout%%32==0, tile_n 128");
  DC_REQUIRE(((uintptr_t)p->src0 & 15) == 0 && ((uintptr_t)p->W & 15) == 0 && ((uintptr_t)p->src1 & 15) == 0, DC_ERR_ALIGN,
             "dc_igemm: src/W must be 16-byte aligned");
  const int epc = 16 / dc_dtype_size(p->dtype);
  DC_REQUIRE(p->ld0 % epc == 0 && p->ld1 % epc == 0 && (p->ld0 == 0 || p->ld0 >= p->C0) && (p->ld1 == 0 || p->ld1 >= p->C1),
             DC_ERR_SHAPE, "dc_igemm: ld0=%d ld1=%d", p->ld0, p->ld1);
  const int cout_out = p->act == DC_ACT_GEGLU ? p->Cout / 2 : p->Cout;
  DC_REQUIRE(p->out_ld >= cout_out, DC_ERR_SHAPE, "dc_igemm: out_ld %d < %d", p->out_ld, cout_out);
  if (p->residual) DC_REQUIRE(p->res_ld >= cout_out, DC_ERR_SHAPE, "dc_igemm: res_ld");
  if (p->rowvec) DC_REQUIRE(p->rowvec_ld >= cout_out, DC_ERR_SHAPE, "dc_igemm: rowvec_ld");
  if (p->gate) DC_REQUIRE(p->gate_ld >= cout_out, DC_ERR_SHAPE, "dc_igemm: gate_ld");

  IgemmArgs a;
  a.src0 = p->src0; a.map0 = p->map0; a.src1 = p->src1; a.map1 = p->map1; a.W = p->W; a.bias = p->bias;
  a.rowvec = p->rowvec; a.rowvec_map = p->rowvec_map; a.gate = p->gate; a.gate_map = p->gate_map;
  a.residual = p->residual; a.res_map = p->res_map; a.out = p->out;
  a.gn_scale = p->gn_scale; a.gn_shift = p->gn_shift; a.gn_silu = p->gn_silu;
  a.src2 = p->src2; a.map2 = p->map2; a.W2 = p->W2; a.C2 = p->C2; a.ld2 = p->ld2 ? p->ld2 : p->C2;
  a.ln_eps = p->ln_eps; a.qstats = p->qstats;
  a.pn_out = p->pn_out; a.pn_gamma = p->pn_gamma; a.pn_beta = p->pn_beta; a.pn_cnt = reinterpret_cast<unsigned*>(p->pn_cnt);
  a.pn_ld = p->pn_ld ? p->pn_ld : p->Cout; a.pn_groups = p->pn_groups; a.pn_silu = p->pn_silu; a.pn_eps = p->pn_eps;
  DC_REQUIRE((p->gn_scale == nullptr) == (p->gn_shift == nullptr), DC_ERR_ARG, "dc_igemm: gn_scale/gn_shift must both be set or null");
  a.C0 = p->C0; a.C1 = p->C1; a.ld0 = p->ld0 ? p->ld0 : p->C0; a.ld1 = p->ld1 ? p->ld1 : p->C1;
  a.rowvec_ld = p->rowvec_ld; a.gate_ld = p->gate_ld; a.res_dtype = p->res_dtype;
  a.res_ld = p->res_ld; a.out_dtype = p->out_dtype; a.out_ld = p->out_ld; a.act = p->act;
  a.taps = p->taps; a.stride = p->stride; a.upsample = p->upsample; a.Hin = p->Hin; a.Win = p->Win;
  a.Hout = p->Hout; a.Wout = p->Wout; a.Cout = p->Cout;
  a.M = (int)M; a.Ktot = p->taps * (p->C0 + p->C1);
  a.c0chunks = p->C0 / bke; a.cpt = (p->C0 + p->C1) / bke; a.nk = p->taps * a.cpt;
  const int bn = p->tile_n;
  a.tiles_m = (a.M + 127) / 128;
  a.tiles_n = dc_igemm_cout_pad(p->Cout, bn) / bn;
  {
    // N fastest when the whole weight matrix can stay in an XCD's L2 next to the activation stream (<= 2 MiB), and for 1-tap GEMMs
    // up to DCAMD_NFAST_GEMM_BYTES (default 16 MiB): there the activation panel of an M tile (rows x K) is the big operand — with
    // M fastest it is re-fetched from HBM once per N panel (DiT-B/4 qkv: 9 x 786 MB per launch, the GEMM ran HBM-bound), with N
    // fastest the N tiles of an M tile run side by side on one XCD and share it in L2, while the weights come back from L2 / MALL
    constexpr long long gemm_cap = 16LL << 20;
    const long long wbytes = (long long)dc_igemm_cout_pad(p->Cout, bn) * a.Ktot * dc_dtype_size(p->dtype);
    a.n_fast = (a.tiles_n > 1 && (wbytes <= (2 << 20) || (p->taps == 1 && wbytes <= gemm_cap))) ? 1 : 0;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // tile_n 128: the LDS-DMA kernels (igemm_pipe.hip, igemm_wide.hip, igemm_xreg.hip, conv3_*.hip).
  // the LDS-DMA kernels finish with the lane-resident epilogue (igemm_epilogue.h): whole 16-byte runs of 8 channels in
  // and out.  Anything else (channel counts / leading dimensions that are not multiples of 8, unaligned side
  // operands) takes the register-staged kernel, whose element-wise epilogue handles every case.
  const bool lane_epi_ok = cout_out % 8 == 0 && p->out_ld % 8 == 0 && ((uintptr_t)p->out & 15) == 0 &&      // (a null `out`: producer-side GroupNorm only)
                           (!p->residual || (p->res_ld % 8 == 0 && ((uintptr_t)p->residual & 15) == 0)) &&
                           (!p->bias || ((uintptr_t)p->bias & 15) == 0) &&
                           (!p->rowvec || (((uintptr_t)p->rowvec & 15) == 0 && p->rowvec_ld % 4 == 0)) &&
                           (!p->gate || (((uintptr_t)p->gate & 15) == 0 && p->gate_ld % 4 == 0 && p->act == DC_ACT_NONE)) &&
                           (!p->residual || p->res_dtype == p->dtype) && (p->out_dtype == p->dtype || p->out_dtype == DC_F32);
  const bool use_v1 = !lane_epi_ok;
  static const bool no_halo = getenv("DCAMD_NO_HALO") != nullptr;
  const char* dn = p->dtype == DC_BF16 ? "bf16" : (p->dtype == DC_F16 ? "f16" : "f32");
  const bool halo_ok = bn == 128 && !use_v1 && !no_halo && dc_conv3_halo_applicable(a, p->dtype);
  // four-phase upsample conv (W in the phase-summed form): the caller opted in, so anything else is an error
  if (p->up4) {
    const bool up4_halo = bn == 128 && !use_v1 && !no_halo && !a.src1 && dc_conv3_up4_applicable(a, p->dtype) &&
                          (!a.qstats || (p->out_dtype == p->dtype && p->Cout % 8 == 0 && ((uintptr_t)a.qstats & 15) == 0 &&
                                         (a.Hin >> 1) >= 8 && (a.Win >> 1) >= 8));      // no quad statistics from mosaic (< 8x8) patches
    // sources smaller than 8x8 (4x4 -> 8x8): the same four phases on the tap-gather kernel (no quad statistics there)
    const bool up4_pipe = !up4_halo && bn == 128 && !use_v1 && !a.src1 && !a.qstats && p->taps == 9 && p->stride == 1 &&
                          p->upsample && p->act == DC_ACT_NONE && !p->gate && !p->residual && !a.gn_scale && !a.src2 &&
                          p->Hin >= 4 && p->Win >= 4 && p->Hin % 2 == 0 && p->Win % 2 == 0 && (p->Hin < 16 || p->Win < 16);
    if (p->pn_out) {      // producer-side GroupNorm on the four-phase form: the 4-wave one-image-per-patch kernel only (sources of 16x16 and more)
      const bool pn_ok = up4_halo && a.Hin > 16 && a.Win > 16 && dc_conv3_halo_pn_ok(a, p->dtype, true) && a.qstats && a.pn_gamma && a.pn_beta && a.pn_cnt &&
                         p->out_dtype == p->dtype && a.pn_ld % 8 == 0 && a.pn_ld >= p->Cout && (((uintptr_t)a.pn_out | (uintptr_t)a.qstats) & 15) == 0 &&
                         ((uintptr_t)a.pn_cnt & 3) == 0 && a.pn_eps > 0.f;
      if (!pn_ok) {
        if (variant) { *variant = "producer-groupnorm-unsupported"; return DC_ERR_UNSUPPORTED; }
        dc_set_error("dc_igemm: pn_out given but this upsample conv cannot normalise its own output (see dc_igemm_pn_ok)");
        return DC_ERR_UNSUPPORTED;
      }
    }
    if (!up4_halo && !up4_pipe) {
      if (variant) { *variant = "up4-unsupported"; return DC_ERR_UNSUPPORTED; }
      dc_set_error("dc_igemm: up4 given but this problem cannot take the four-phase upsample conv (see dc_igemm_up4_ok)");
      return DC_ERR_UNSUPPORTED;
    }
    if (variant) {
      static thread_local char name4[64];
      if (up4_halo) snprintf(name4, sizeof(name4), p->pn_out ? "conv3_up4<%s,%dw,pn>" : "conv3_up4<%s,%dw>", dn, (a.Hin <= 16 || a.Win <= 16) ? 8 : 4);
      else snprintf(name4, sizeof(name4), "igemm_pipe_up4<%s,256x128,3st>", dn);
      *variant = name4;
      return DC_OK;
    }
    if (up4_halo) return dc_conv3_up4_launch(a, p->dtype, p->n_img, s);
    return dc_igemm_launch_pipe_up4(a, p->dtype, s);
  }
  // wave-specialised halo conv: takes the fused GroupNorm prologue (also together with the 1x1 side source / a residual); DCAMD_WS_PLAIN
  // routes the plain one-image-per-patch convs there too (A/B of the structure alone)
  static const bool ws_plain = getenv("DCAMD_WS_PLAIN") != nullptr;
  const bool ws_ok = halo_ok && dc_conv3_ws_ok(a, p->dtype) && !dc_conv3_thin_applicable(a, p->dtype);
  const bool use_ws = ws_ok && (a.gn_scale || ws_plain);
  if (a.src2) {
    const int bke64 = 64 / dc_dtype_size(p->dtype);
    const bool side_ok = halo_ok && (!a.gn_scale || ws_ok) && !a.upsample && a.W2 && a.C2 >= 2 * bke64 && a.C2 % bke64 == 0 && a.ld2 % epc == 0 &&
                         (((uintptr_t)a.src2 | (uintptr_t)a.W2) & 15) == 0 && lane_epi_ok;
    if (!side_ok) {
      if (variant) { *variant = "side-source-unsupported"; return DC_ERR_UNSUPPORTED; }
      dc_set_error("dc_igemm: src2/W2 given but this problem cannot take the 1x1 side source (see dc_igemm_side_ok)");
      return DC_ERR_UNSUPPORTED;
    }
  }
  const bool thin_gn = a.gn_scale && dc_conv3_thin_applicable(a, p->dtype) && ((uintptr_t)p->out & 3) == 0;   // conv_out: normalised in the halo
  if (a.gn_scale && !thin_gn && !ws_ok) {
    if (variant) { *variant = "gn-not-fusable"; return DC_ERR_UNSUPPORTED; }
    dc_set_error("dc_igemm: gn_scale/gn_shift given but this problem cannot take the fused GroupNorm prologue (see dc_igemm_gn_fusable)");
    return DC_ERR_UNSUPPORTED;
  }
  if (a.qstats) {
    const bool thin_q = dc_conv3_thin_applicable(a, p->dtype);
    const bool qs_ok = halo_ok && !thin_q && p->out_dtype == p->dtype && p->Cout % 8 == 0 && ((uintptr_t)a.qstats & 15) == 0 &&
                       a.Hin >= 8 && a.Win >= 8;          // mosaic patches (images below 8x8) emit none: a wave's half holds four images
    if (!qs_ok) {
      if (variant) { *variant = "qstats-unsupported"; return DC_ERR_UNSUPPORTED; }
      dc_set_error("dc_igemm: qstats given but this problem cannot emit quad statistics (see dc_igemm_qstats_parts)");
      return DC_ERR_UNSUPPORTED;
    }
  }
  // producer-side GroupNorm: the caller opted in (dc_igemm_pn_ok), anything that cannot take it is an error
  const bool use_pn = a.pn_out != nullptr;
  if (use_pn) {
    const bool pn_ok = halo_ok && !use_ws && !a.gn_scale && !dc_conv3_thin_applicable(a, p->dtype) && dc_conv3_halo_pn_ok(a, p->dtype, false) &&
                       (a.qstats || (a.Hin == 4 && a.Win == 4)) && a.pn_gamma && a.pn_beta && a.pn_cnt && p->out_dtype == p->dtype && a.pn_ld % 8 == 0 &&
                       a.pn_ld >= p->Cout && (((uintptr_t)a.pn_out | (uintptr_t)a.qstats) & 15) == 0 && ((uintptr_t)a.pn_cnt & 3) == 0 && a.pn_eps > 0.f;
    if (!pn_ok) {
      if (variant) { *variant = "producer-groupnorm-unsupported"; return DC_ERR_UNSUPPORTED; }
      dc_set_error("dc_igemm: pn_out given but this problem cannot normalise its own output (see dc_igemm_pn_ok)");
      return DC_ERR_UNSUPPORTED;
    }
  } else {
    DC_REQUIRE(p->out, DC_ERR_ARG, "dc_igemm: null out");
  }
  // short-K GEMMs: the activation-stationary kernel wins for GEGLU (448 vs 376 TFLOP/s at K = 256, 584 vs 544 at K = 512);
  // for plain epilogues the 256x256 tile is faster where it applies (q/k/v 505-709 vs 478-556), xreg elsewhere
  const bool ln_ok = bn == 128 && !use_v1 && !a.src1 && dc_igemm_xreg_applicable(a, p->dtype);
  if (a.ln_eps > 0.f && !ln_ok) {
    if (variant) { *variant = "row-layernorm-unsupported"; return DC_ERR_UNSUPPORTED; }
    dc_set_error("dc_igemm: ln_eps given but this problem cannot take the fused row LayerNorm (see dc_igemm_ln_ok)");
    return DC_ERR_UNSUPPORTED;
  }
  const bool use_xreg = bn == 128 && dc_igemm_xreg_applicable(a, p->dtype) && (p->act == DC_ACT_GEGLU || a.ln_eps > 0.f || dc_igemm_pipe_shape(a) != 2);
  const bool thin = dc_conv3_thin_applicable(a, p->dtype) && ((uintptr_t)p->out & 3) == 0;
  if (variant) {
    static thread_local char name[64];
    if (thin) { snprintf(name, sizeof(name), "conv3_thin<%s>", dn); *variant = name; return DC_OK; }
    if (use_ws) snprintf(name, sizeof(name), a.gn_scale ? "conv3_ws<%s,gn>" : "conv3_ws<%s>", dn);
    else if (use_pn) snprintf(name, sizeof(name), "conv3_halo<%s,%dw,pn>", dn, (a.Hin <= 8 || a.Win <= 8) ? 8 : 4);
    else if (bn == 128 && !use_v1 && !no_halo && dc_conv3_halo_applicable(a, p->dtype)) snprintf(name, sizeof(name), "conv3_halo<%s,%dw>", dn, (a.Hin <= 8 || a.Win <= 8) ? 8 : 4);
    else if (bn == 128 && !use_v1 && use_xreg) snprintf(name, sizeof(name), "igemm_xreg<%s,96xN>", dn);
    else if (bn == 128 && !use_v1) {
      static const char* const shapes[4] = {"igemm_pipe<%s,128x128,2st>", "igemm_pipe<%s,256x128,3st>", "igemm_pipe<%s,256x256,2st>",
                                            "igemm_wide8<%s,256x256>"};
      const int shp = dc_igemm_pipe_shape(a);
      snprintf(name, sizeof(name), shapes[shp == 2 ? 3 : shp], dn);
    }
    else snprintf(name, sizeof(name), "igemm<%s,128x%d>", dn, bn);
    *variant = name;
    return DC_OK;
  }
  if (thin) return dc_conv3_thin_launch(a, p->dtype, p->n_img, s);
  if (use_ws) return dc_conv3_ws_launch(a, p->dtype, p->n_img, s);
  if (bn == 128 && !use_v1 && !no_halo && dc_conv3_halo_applicable(a, p->dtype)) return dc_conv3_halo_launch(a, p->dtype, p->n_img, s);
  if (bn == 128 && !use_v1 && use_xreg) return dc_igemm_xreg_launch(a, p->dtype, s);
  if (bn == 128 && !use_v1) return dc_igemm_launch_pipe(a, p->dtype, s);
  if (bn == 128) {
    if (p->dtype == DC_BF16) return launch<__bf16, 128, 128, 2, 2>(a, s);
    if (p->dtype == DC_F16) return launch<_Float16, 128, 128, 2, 2>(a, s);
    return launch<float, 128, 128, 2, 2>(a, s);
  }
  if (p->dtype == DC_BF16) return launch<__bf16, 128, 32, 4, 1>(a, s);
  if (p->dtype == DC_F16) return launch<_Float16, 128, 32, 4, 1>(a, s);
  return launch<float, 128, 32, 4, 1>(a, s);
}
