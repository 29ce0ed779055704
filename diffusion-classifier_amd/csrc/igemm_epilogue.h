// igemm_epilogue.h — lane-resident epilogue shared by igemm_pipe.hip and conv3_halo.hip.
//
// s_memtime stamps of the earlier LDS-staged epilogue: 34 k (no residual) to 49 k (residual) cycles of a
// 114-156 k-cycle conv3_halo tile — four workgroup barriers, each behind the full latency of the residual
// loads / output stores of only two waves per SIMD.  This version needs no LDS and no barrier:
//
//  * the weight rows of an N tile are PERMUTED on their way into LDS (epi_wrow, applied to the LDS-DMA source
//    address), so that after the MFMAs a lane owns, for each of its pixels, runs of 8 CONSECUTIVE output
//    channels: D row (lq*4 + reg) of cout fragment i is channel  wave_base + (i>>1)*32 + lq*8 + (i&1)*4 + reg.
//    One 16-byte access per run (16-bit data), and the four lanes lq = 0..3 of a pixel cover 64 contiguous bytes;
//  * GEGLU: fragments (2h, 2h+1) carry the value / gate rows of the same channels wave_base + lq*8 + h*4 + reg,
//    so the product is formed in registers and a lane stores 8 consecutive outputs;
//  * every load of the wave (residual chunks, bias, per-sample row vector / gate) is issued AND consumed before the
//    first store (see epi_direct_act): no load ever queues behind a store acknowledgement.
#pragma once
#include "igemm_common.h"
#ifndef DC_STAMP
#define DC_STAMP(k) do {} while (0)   // conv3_halo.hip defines the diagnostic version (-DDC_STAMPS builds only)
#endif
#ifndef DC_HALO_ABL
#define DC_HALO_ABL() 0               // timing-only ablation switches of diagnostic builds (conv3_halo.hip)
#endif

// LDS row R (0..127) of the weight tile -> packed weight row of the N tile that must be loaded there
__device__ __forceinline__ int epi_wrow(int R, bool geglu) {
  const int wn = R >> 6, i = (R >> 4) & 3, lq = (R >> 2) & 3, reg = R & 3;
  if (geglu) {
    const int o = wn * 32 + lq * 8 + (i >> 1) * 4 + reg;           // output channel inside the tile (64 per tile)
    return (o >> 4) * 32 + ((i & 1) << 4) + (o & 15);              // host packing: 16 value rows, 16 gate rows, ...
  }
  return wn * 64 + (i >> 1) * 32 + lq * 8 + (i & 1) * 4 + reg;
}

struct EpiRow { int o, r, samp; bool ok; };   // output row, residual row (units of one channel row; both < 2^31), sample

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}

// acc[i][j]: cout fragment i (4 per wave), pixel fragment j; rowfn(j, EpiRow&) describes the lane's pixel of fragment j;
// samp_first / samp_last: sample of the wave's first / last pixel (wave-uniform).  TWO_SAMP: the caller guarantees that
// every pixel of the wave belongs to one of those two samples.
// T = compute type of the kernel.  Preconditions (checked by the dispatcher, igemm.hip `lane_epi_ok`): output channels,
// out_ld, res_ld multiples of 8; out / residual / bias / rowvec / gate 16-byte aligned; rowvec_ld, gate_ld multiples
// of 4; the residual is stored as T and the output as T or fp32 (so no data-type branch sits inside the unrolled
// loops: taken scalar branches cost ~20 cycles each on this core, and the first version had ten per store).
//
// vmcnt is ONE in-order counter for loads and stores on gfx9: a load issued after a store cannot be waited for
// without also waiting for that store's acknowledgement (a first version that alternated "loads, math, stores" per
// group of pixels spent 25-38 k cycles per conv tile exactly there).  Hence three phases: (A) per batch of (up to) four pixel
// fragments: all loads, then the final fp32 values are formed IN PLACE in the accumulators; (B) only after the last
// batch: conversions and stores, nothing left to wait for.
struct EpiNoPre { __device__ __forceinline__ void operator()() const {} };
// bias values fetched by the caller ahead of time (igemm_xreg: the loads of an N tile's bias are issued before its K loop, so
// their latency sits under the MFMAs instead of in front of the epilogue).  BiasFn::on selects the code; biasfn(k, bs, bgt)
// fills run k's 8 bias values (and, GEGLU, the 8 gate-row values).  has_rowvec: the values already include the per-sample row
// vector (conv3_halo stages bias + row vector of its sample in LDS at kernel start), so the epilogue fetches neither.
struct EpiNoBias {
  static constexpr bool on = false, has_rowvec = false;
  __device__ __forceinline__ void operator()(int, float (&)[8], float (&)[8]) const {}
};

// quad statistics for a following GroupNorm (IgemmArgs::qstats).  QsFn::on selects the code; qsfn(half, n, part):
// the output sample and part index of pixel fragments [4 half, 4 half + 4) of the wave (wave-uniform), false when that
// sample does not exist; qsfn.whole(): both halves belong to the same (sample, part) and are written as one record.
template <int V> struct EpiIC { static constexpr int value = V; };
// wave-local producer-side GroupNorm (conv3_halo's 8x8-image form: every image of the patch lies inside one wave's 128 pixels, two per
// wave): PnL::on selects the code — the epilogue then ALSO stores act(GroupNorm(v)) of its output v into IgemmArgs::pn_out, from the
// quad records it forms anyway; nothing crosses a wave (csrc/epi_pn.h is the form for images that span several workgroups)
struct EpiNoPnLocal { static constexpr bool on = false; static constexpr int IMG_FR = 4; };
struct EpiPnLocal8x8 { static constexpr bool on = true; static constexpr int IMG_FR = 4; };     // an image = 4 pixel fragments of the wave (records from `emit`)
struct EpiPnLocal4x4 { static constexpr bool on = true; static constexpr int IMG_FR = 1; };     // mosaic patches: an image = ONE fragment (16 pixels); no quad records exist
struct EpiNoQs {
  static constexpr bool on = false;
  __device__ __forceinline__ bool operator()(int, int&, int&) const { return false; }
  __device__ __forceinline__ bool whole() const { return true; }
  __device__ __forceinline__ int parts() const { return 1; }
};

template <typename T, int TM, int ACT, bool GATE, bool TWO_SAMP, typename RowFn, typename PreFn = EpiNoPre, typename QsFn = EpiNoQs,
          typename BiasFn = EpiNoBias, typename PnL = EpiNoPnLocal>
__device__ __forceinline__ void epi_direct_act(const IgemmArgs& a, f32x4 (&acc)[4][TM], int tile_n, int wn, int lq,
                                               int samp_first, int samp_last, RowFn rowfn, PreFn prefn = PreFn(), QsFn qsfn = QsFn(),
                                               BiasFn biasfn = BiasFn(), PnL = PnL()) {
  static_assert(!PnL::on || (TM == 8 && ACT == DC_ACT_NONE && !GATE && (QsFn::on || PnL::IMG_FR == 1)), "wave-local GroupNorm: the plain 128-pixel halo epilogue");
  constexpr bool geglu = ACT == DC_ACT_GEGLU;
  constexpr int NK = geglu ? 1 : 2;                  // 8-channel runs per pixel
  constexpr int JB = TM;                             // pixel fragments per load batch: all of them — one exposed residual latency per tile
  const int cout_out = geglu ? (a.Cout >> 1) : a.Cout;
  const int c0 = (geglu ? tile_n * 64 + wn * 32 : tile_n * 128 + wn * 64) + lq * 8;     // run k starts at c0 + 32 k
  constexpr bool res16 = sizeof(T) == 2;
  bool con[NK];                                      // run k holds real channels
#pragma unroll
  for (int k = 0; k < NK; ++k) con[k] = c0 + 32 * k < cout_out;

  // ---- phase A1: bias + per-sample row vector, added into the accumulators (their registers are free again before
  // the residual chunks are fetched).  Packed bias index: GEGLU value rows at (c>>4)*32 + (c&15), gate rows 16 further.
  {
    float bs[NK][8], bgt[8];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int c = c0 + 32 * k;
#pragma unroll
      for (int e = 0; e < 8; ++e) { bs[k][e] = 0.f; if (k == 0) bgt[e] = 0.f; }
      if constexpr (BiasFn::on) {
        biasfn(k, bs[k], bgt);
      } else if (a.bias && con[k]) {
        const int p = geglu ? (c >> 4) * 32 + (c & 15) : c;
        ld8(a.bias + p, bs[k]);
        if (geglu) ld8(a.bias + p + 16, bgt);
      }
    }
    // one row-vector fetch for the wave when its pixels share a sample (samp_first == samp_last, both wave-uniform:
    // a wave's pixels normally lie inside one image)
    const bool uni = samp_first == samp_last;
    prefn();      // the caller's own waits (igemm_xreg drains its LDS-DMA here) overlap the bias loads issued above
    // The row vector is added AFTER the bias in every branch — (acc + bias) + rowvec — also when the whole wave shares one sample and
    // one fetch serves it.  Rounds 1-3 pre-summed bias + rowvec in that case: acc + (bias + rowvec) differs from the per-pixel
    // branches in the last fp32 bit, and which branch a sample meets depends on where it sits in the launch (the LAST sample of a
    // launch is alone in its wave where it otherwise shares one): the quad statistics of an 8x8-level conv then depended on the
    // micro-batch size (round 4: tests/test_gpu_dist.py, world size 3 against 1).
    const bool rv_uni = !BiasFn::has_rowvec && !geglu && a.rowvec && uni;
    float rvu[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) rvu[k][e] = 0.f;
    if (rv_uni) {
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (con[k]) ld8(a.rowvec + (size_t)(a.rowvec_map ? a.rowvec_map[samp_first] : samp_first) * a.rowvec_ld + c0 + 32 * k, rvu[k]);
    }
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (geglu) { acc[(e >> 2) * 2][j][e & 3] += bs[0][e]; acc[(e >> 2) * 2 + 1][j][e & 3] += bgt[e]; }
        else {
#pragma unroll
          for (int k = 0; k < NK; ++k) acc[2 * k + (e >> 2)][j][e & 3] += bs[k][e];
        }
      }
    if (rv_uni) {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int k = 0; k < NK; ++k) acc[2 * k + (e >> 2)][j][e & 3] += rvu[k][e];
    }
    if (!BiasFn::has_rowvec && !geglu && a.rowvec && !uni) {
      if (TWO_SAMP) {
        // the wave's pixels belong to samp_first or samp_last only (images at least as large as the wave's pixel range:
        // the dispatcher guarantees it): fetch both vectors, select per lane — no per-pixel address arithmetic
#pragma unroll
        for (int k = 0; k < NK; ++k)
          if (con[k]) {
            float ra[8], rb[8];
            ld8(a.rowvec + (size_t)(a.rowvec_map ? a.rowvec_map[samp_first] : samp_first) * a.rowvec_ld + c0 + 32 * k, ra);
            ld8(a.rowvec + (size_t)(a.rowvec_map ? a.rowvec_map[samp_last] : samp_last) * a.rowvec_ld + c0 + 32 * k, rb);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
              EpiRow rj;
              rowfn(j, rj);
              const bool second = rj.samp != samp_first;
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[2 * k + (e >> 2)][j][e & 3] += second ? rb[e] : ra[e];
            }
          }
      } else {                                 // any number of samples per wave (tiny images): one fetch per fragment
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          EpiRow rj;
          rowfn(j, rj);
#pragma unroll
          for (int k = 0; k < NK; ++k)
            if (con[k]) {
              float rv[8];
              ld8(a.rowvec + (size_t)(a.rowvec_map ? a.rowvec_map[rj.samp] : rj.samp) * a.rowvec_ld + c0 + 32 * k, rv);
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[2 * k + (e >> 2)][j][e & 3] += rv[e];
            }
        }
      }
    }
  }

  DC_STAMP(3);
  int orow[TM];                                      // output row per pixel fragment, -1 = nothing to store
  // ---- phase A2: per batch, residual chunks (and gates), then activation / gate / residual in place ----
#pragma unroll
  for (int j0 = 0; j0 < TM; j0 += JB) {
    EpiRow row[JB];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) { rowfn(j0 + jj, row[jj]); orow[j0 + jj] = row[jj].ok ? row[jj].o : -1; }
    // 16-bit residual: one 16-byte chunk per run; an fp32 residual (fp32 parity path only) is fetched at the point of
    // use, still ahead of every store
    chunk16 rc[JB][NK];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        rc[jj][k] = chunk16{0u, 0u, 0u, 0u};
        if (res16 && a.residual && row[jj].ok && con[k])
          rc[jj][k] = *reinterpret_cast<const chunk16*>(reinterpret_cast<const char*>(a.residual) + ((size_t)row[jj].r * a.res_ld + c0 + 32 * k) * 2);
      }
    if (j0 == 0) DC_STAMP(4);
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = j0 + jj;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        if (!(row[jj].ok && con[k])) continue;
        float rf[8], gt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { rf[e] = 0.f; gt[e] = 1.f; }
        if (GATE) ld8(a.gate + (size_t)(a.gate_map ? a.gate_map[row[jj].samp] : row[jj].samp) * a.gate_ld + c0 + 32 * k, gt);
        if (a.residual) {
          if (!res16) ld8(reinterpret_cast<const float*>(a.residual) + (size_t)row[jj].r * a.res_ld + c0 + 32 * k, rf);
          else chunk_to_f<T>(rc[jj][k], rf);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (geglu) {       // value fragments 0 / 2 receive the product; the gate fragments are dead afterwards
            acc[(e >> 2) * 2][j][e & 3] = acc[(e >> 2) * 2][j][e & 3] * gelu_erf_t<T>(acc[(e >> 2) * 2 + 1][j][e & 3]) + rf[e];
          } else {
            float x = acc[2 * k + (e >> 2)][j][e & 3];
            if (ACT == DC_ACT_SILU) x = silu_t<T>(x);
            if (ACT == DC_ACT_GELU_TANH) x = gelu_tanh_t<T>(x);
            if (GATE) x *= gt[e];
            acc[2 * k + (e >> 2)][j][e & 3] = x + rf[e];
          }
        }
      }
    }
  }
  DC_STAMP(5);
  // ---- quad statistics of the values about to be stored, for the GroupNorm that consumes the tensor: per (sample, part, quad
  // of 4 consecutive channels) the MEAN and the centred second moment M2 = sum (v - mean)^2 over the quad's 4 x npx values —
  // not (sum, sum of squares), whose difference cancels when |mean| >> std.  Shifted sums: every 16-lane row of the wave (16
  // pixels x 8 channels of a fragment row) takes, per quad, the first pixel's first channel as the pivot (DPP row_share:0),
  // accumulates S' = sum (v - pivot) and Q' = sum (v - pivot)^2 over the lane's pixels (two adjacent accumulator registers per
  // packed instruction), reduces them over the row's 16 lanes with four DPP adds, and the row's first lane forms
  // mean = pivot + S'/n, M2 = Q' - S'^2/n and writes one 16-byte record pair {mean0, M2_0, mean1, M2_1} per run.  What is left
  // inside S' and Q' is the spread BETWEEN the channels of a quad, which is part of the group's variance anyway.
  // Fixed order: the result depends on the tile geometry only.  The fp32 values are taken BEFORE the rounding to T (the
  // re-conversion cost as much as the sums): they differ from the stored tensor's statistics by ~2^-9 / sqrt(count) relative.
  float pnr[(PnL::on && PnL::IMG_FR == 4) ? 2 : 1][NK][4];                    // wave-local GroupNorm: the quad records of the wave's two images (every lane of a row holds them)
  float pgm[PnL::on ? NK : 1][8], pbt[PnL::on ? NK : 1][8];
  if constexpr (PnL::on) {                              // gamma / beta of the lane's runs, fetched in front of every store of the wave
#pragma unroll
    for (int k = 0; k < NK; ++k) { ld8(a.pn_gamma + c0 + 32 * k, pgm[k]); ld8(a.pn_beta + c0 + 32 * k, pbt[k]); }
  }
  if constexpr (QsFn::on) {
    static_assert(!QsFn::on || (TM == 8 && ACT != DC_ACT_GEGLU), "quad statistics: 128-pixel wave tiles, plain epilogue");
    if (a.qstats) {
      typedef __attribute__((ext_vector_type(2))) float f32x2;    // two adjacent accumulator registers: v_pk_add_f32 / v_pk_fma_f32
      auto share0 = [](float v) {                                 // the value lane 0 of my 16-lane row holds
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150, 0xF, 0xF, true));
      };
      // sum over the 16 pixel lanes of a row of the wave: four DPP adds (xor 1, xor 2, half-row mirror, row mirror)
      auto row_sum = [](float v) {
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
        return v;
      };
      // one (sample, part) record set from pixel fragments [J0, J1) of the wave
      auto emit = [&](auto j0c, auto j1c, int h) {
        // the half-tile instances (8x8 images: two per wave tile) and the whole-tile one must do the SAME arithmetic, or an image's
        // statistics would depend on which half of a patch — i.e. on how many units share a launch — it sits in: no implicit
        // contraction in here, fused multiply-adds only where written
#pragma clang fp contract(off)
        constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
        float piv[NK][2];
        f32x2 sm[NK][2], sq[NK][2];                               // [run][quad]: (even, odd) register pairs, folded at the end
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
          for (int qd = 0; qd < 2; ++qd) {
            piv[k][qd] = share0(acc[2 * k + qd][J0][0]);
            sm[k][qd] = f32x2{0.f, 0.f}; sq[k][qd] = f32x2{0.f, 0.f};
          }
#pragma unroll
        for (int j = J0; j < J1; ++j)
#pragma unroll
          for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int qd = 0; qd < 2; ++qd)
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                const f32x2 d = f32x2{acc[2 * k + qd][j][e], acc[2 * k + qd][j][e + 1]} - f32x2{piv[k][qd], piv[k][qd]};
                sm[k][qd] += d;
                sq[k][qd] = __builtin_elementwise_fma(d, d, sq[k][qd]);
              }
        const float inv_n = 1.0f / (float)((J1 - J0) * 16 * 4);    // values per quad record
        float r[NK][4];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
          for (int qd = 0; qd < 2; ++qd) {
            const float S = row_sum(sm[k][qd][0] + sm[k][qd][1]), Q = row_sum(sq[k][qd][0] + sq[k][qd][1]);
            r[k][2 * qd] = __builtin_fmaf(S, inv_n, piv[k][qd]);
            r[k][2 * qd + 1] = fmaxf(__builtin_fmaf(-S * S, inv_n, Q), 0.f);
          }
        if constexpr (PnL::on && PnL::IMG_FR == 4) {
#pragma unroll
          for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) pnr[h][k][e] = r[k][e];
        }
        int n = 0, part = 0;
        if ((threadIdx.x & 15) == 0 && qsfn(h, n, part)) {
#pragma unroll
          for (int k = 0; k < NK; ++k)
            if (con[k])
              *reinterpret_cast<f32x4*>(a.qstats + (((size_t)n * qsfn.parts() + part) * (a.Cout >> 2) + ((c0 + 32 * k) >> 2)) * 2) =
                  f32x4{r[k][0], r[k][1], r[k][2], r[k][3]};
        }
      };
      // no per-pixel / per-run masks: a pixel is only ever invalid together with its whole sample, whose record is not
      // written (qsfn returns false), and runs past Cout are not written either
      if (qsfn.whole()) emit(EpiIC<0>{}, EpiIC<TM>{}, 0);
      else { emit(EpiIC<0>{}, EpiIC<TM / 2>{}, 0); emit(EpiIC<TM / 2>{}, EpiIC<TM>{}, 1); }
    }
  }
  // ---- phase B: conversions and stores only (one straight-line sequence per output type) ----
  auto store_all = [&](void* outp, int ld, int odt) {
    if (odt == DC_F32) {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int k = 0; k < NK; ++k)
          if (orow[j] >= 0 && con[k]) {
            float* op = reinterpret_cast<float*>(outp) + (size_t)orow[j] * ld + c0 + 32 * k;
            if (geglu) { *reinterpret_cast<f32x4*>(op) = acc[0][j]; *reinterpret_cast<f32x4*>(op + 4) = acc[2][j]; }
            else { *reinterpret_cast<f32x4*>(op) = acc[2 * k][j]; *reinterpret_cast<f32x4*>(op + 4) = acc[2 * k + 1][j]; }
          }
    } else {
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int k = 0; k < NK; ++k)
          if (orow[j] >= 0 && con[k]) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = geglu ? acc[(e >> 2) * 2][j][e & 3] : acc[2 * k + (e >> 2)][j][e & 3];
            *reinterpret_cast<chunk16*>(reinterpret_cast<T*>(outp) + (size_t)orow[j] * ld + c0 + 32 * k) = f_to_chunk<T>(v);
          }
    }
  };
  if constexpr (!PnL::on) {
    store_all(a.out, a.out_ld, a.out_dtype);
  } else {
    // ---- wave-local producer-side GroupNorm (8x8 images: pixel fragments 0-3 are one image, 4-7 the next): raw store (if the raw tensor
    // has a reader), then the group statistics from the quad records of each image — merged over the group's quads in channel order,
    // with the lanes 16 / 32 apart that hold the group's other quads when a group is wider than a run — and y = act(v a + b) in place
    if (a.out) store_all(a.out, a.out_ld, a.out_dtype);
    constexpr int IFR = PnL::IMG_FR, NIMG = TM / IFR;       // pixel fragments per image, images per wave
    const int qpg = (a.Cout / a.pn_groups) >> 2;             // quads per group: 1, 2, 4 or 8
    const float nq = (float)(IFR * 64);                      // values per quad record: 16 pixels x 4 channels per fragment
#pragma unroll
    for (int h = 0; h < NIMG; ++h) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        float rq[4];                                         // (mean, M2) of the run's two quads over image h
        if constexpr (IFR == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) rq[e] = pnr[h][k][e];
        } else {
          // one fragment per image: the shifted sums of `emit`, over the row's 16 pixels (no quad records are written for mosaic patches)
#pragma clang fp contract(off)
          auto share0l = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150, 0xF, 0xF, true)); };
          auto row_suml = [](float v) {
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
            return v;
          };
#pragma unroll
          for (int qd = 0; qd < 2; ++qd) {
            const float pv = share0l(acc[2 * k + qd][h][0]);
            float sm_ = 0.f, sq_ = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc[2 * k + qd][h][e] - pv; sm_ += d; sq_ = __builtin_fmaf(d, d, sq_); }
            const float S = row_suml(sm_), Q = row_suml(sq_);
            rq[2 * qd] = __builtin_fmaf(S, 1.0f / 64.0f, pv);
            rq[2 * qd + 1] = fmaxf(__builtin_fmaf(-S * S, 1.0f / 64.0f, Q), 0.f);
          }
        }
        // the (mean, variance) of the group(s) of this run's two quads
        float gm[2], gv[2];
        if (qpg == 1) {
#pragma unroll
          for (int qd = 0; qd < 2; ++qd) { gm[qd] = rq[2 * qd]; gv[qd] = rq[2 * qd + 1] / nq; }
        } else {
#pragma clang fp contract(off)
          // sets in channel order; a lane pair / quadruple orders them alike, so every lane of the group ends with the same bits
          float ms[8], m2[8];
          ms[0] = rq[0]; m2[0] = rq[1]; ms[1] = rq[2]; m2[1] = rq[3];
          int nset = 2;
          if (qpg >= 4) {
            const bool hi = lq & 1;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __shfl_xor(rq[e], 16);
            const float a0 = ms[0], a1 = m2[0], a2 = ms[1], a3 = m2[1];
            ms[0] = hi ? o[0] : a0; m2[0] = hi ? o[1] : a1; ms[1] = hi ? o[2] : a2; m2[1] = hi ? o[3] : a3;
            ms[2] = hi ? a0 : o[0]; m2[2] = hi ? a1 : o[1]; ms[3] = hi ? a2 : o[2]; m2[3] = hi ? a3 : o[3];
            nset = 4;
            if (qpg == 8) {
              const bool hi2 = lq & 2;
              float p[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) { p[2 * e] = __shfl_xor(ms[e], 32); p[2 * e + 1] = __shfl_xor(m2[e], 32); }
              float cm[4], c2[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) { cm[e] = ms[e]; c2[e] = m2[e]; }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                ms[e] = hi2 ? p[2 * e] : cm[e]; m2[e] = hi2 ? p[2 * e + 1] : c2[e];
                ms[4 + e] = hi2 ? cm[e] : p[2 * e]; m2[4 + e] = hi2 ? c2[e] : p[2 * e + 1];
              }
              nset = 8;
            }
          }
          float sm = 0.f, s2 = 0.f;
          for (int i = 0; i < nset; ++i) { sm += ms[i]; s2 += m2[i]; }
          const float mean = sm / (float)nset;
          float sd = 0.f;
          for (int i = 0; i < nset; ++i) { const float d = ms[i] - mean; sd = __builtin_fmaf(d, d, sd); }
          gm[0] = gm[1] = mean;
          gv[0] = gv[1] = fmaxf(__builtin_fmaf(nq, sd, s2) / (nq * (float)nset), 0.f);
        }
        float ga[8], gb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = rsqrtf(gv[e >> 2] + a.pn_eps) * pgm[k][e];
          ga[e] = sc; gb[e] = pbt[k][e] - gm[e >> 2] * sc;
        }
        if (a.pn_silu) {
#pragma unroll
          for (int jj = 0; jj < IFR; ++jj)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = acc[2 * k + (e >> 2)][h * IFR + jj][e & 3];
              acc[2 * k + (e >> 2)][h * IFR + jj][e & 3] = silu_t<T>(x * ga[e] + gb[e]);
            }
        } else {
#pragma unroll
          for (int jj = 0; jj < IFR; ++jj)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = acc[2 * k + (e >> 2)][h * IFR + jj][e & 3];
              acc[2 * k + (e >> 2)][h * IFR + jj][e & 3] = x * ga[e] + gb[e];
            }
        }
      }
    }
    store_all(a.pn_out, a.pn_ld, sizeof(T) == 4 ? DC_F32 : (int)a.out_dtype);
  }
}

// A per-sample gate (DiT adaLN) only comes with DC_ACT_NONE (dispatcher: igemm.hip `lane_epi_ok`).
template <typename T, int TM, typename RowFn>
__device__ __forceinline__ void epi_direct(const IgemmArgs& a, f32x4 (&acc)[4][TM], int tile_n, int wn, int lq,
                                           int samp_first, int samp_last, RowFn rowfn) {
  switch (a.act) {   // wave-uniform
    case DC_ACT_SILU: epi_direct_act<T, TM, DC_ACT_SILU, false, false>(a, acc, tile_n, wn, lq, samp_first, samp_last, rowfn); break;
    case DC_ACT_GEGLU: epi_direct_act<T, TM, DC_ACT_GEGLU, false, false>(a, acc, tile_n, wn, lq, samp_first, samp_last, rowfn); break;
    case DC_ACT_GELU_TANH: epi_direct_act<T, TM, DC_ACT_GELU_TANH, false, false>(a, acc, tile_n, wn, lq, samp_first, samp_last, rowfn); break;
    default:
      if (a.gate) epi_direct_act<T, TM, DC_ACT_NONE, true, false>(a, acc, tile_n, wn, lq, samp_first, samp_last, rowfn);
      else epi_direct_act<T, TM, DC_ACT_NONE, false, false>(a, acc, tile_n, wn, lq, samp_first, samp_last, rowfn);
      break;
  }
}
