// igemm_epilogue.h — LDS-staged epilogue shared by igemm_pipe.hip and conv3_halo.hip.
//
// phase 1 (epi_stage): a wave writes its accumulators (+bias, +per-sample row vector, activation,
//   *gate) into an fp32 LDS tile.  Bias / row-vector / gate values are fetched as 16-byte vectors and
//   ALL loads of a pixel row are issued before the first use (the first version fetched them one
//   float at a time behind branches: s_memtime stamps showed 59 k cycles per 256-row pass there).
// phase 2 (epi_store): 16 consecutive lanes cover one output row; residual rows are read and the
//   result is written as one 16-byte access per chunk, residual loads issued four chunks ahead.
#pragma once
#include "igemm_common.h"

// 4 consecutive floats p[0..3]; elements at index >= n_valid read as `fill` (channel tails)
__device__ __forceinline__ f32x4 ld4(const float* p, bool vec, int n_valid, float fill) {
  if (n_valid >= 4) {
    if (vec) return *reinterpret_cast<const f32x4*>(p);
    return f32x4{p[0], p[1], p[2], p[3]};
  }
  f32x4 v = {fill, fill, fill, fill};
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r < n_valid) v[r] = p[r];
  return v;
}

// samp[j]: sample index of the lane's pixel in tile j (clamped to a valid sample for padded rows)
// rloc0: first staging row of the wave; lc0: first staging column of the wave; pc0: first PACKED
// global cout of the wave (bias/rowvec/gate index; == output channel unless GEGLU)
// ACT is a template parameter on purpose: with a runtime `a.act` hipcc if-converted the activation
// choice and evaluated expf + tanhf + erff for every element (43 k cycles per 256-row pass, stamped).
template <int TM, int TN, int ACT>
__device__ __forceinline__ void epi_stage_act(const IgemmArgs& a, f32x4 (&acc)[TN][TM], float* otile, int old_, int rloc0,
                                              int lc0, int pc0, const int (&samp)[TM], int lr, int lq) {
  constexpr bool geglu = ACT == DC_ACT_GEGLU;
  const int cout_lim = a.Cout;                       // packed channel limit (GEGLU: value+gate rows)
  const bool vb = a.bias && (((uintptr_t)a.bias & 15) == 0);
  const bool vr = a.rowvec && (((uintptr_t)a.rowvec & 15) == 0) && ((a.rowvec_ld & 3) == 0);
  const bool vg = a.gate && (((uintptr_t)a.gate & 15) == 0) && ((a.gate_ld & 3) == 0);
  f32x4 bv[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int pc = pc0 + i * 16 + lq * 4;
    bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv[i] = ld4(a.bias + pc, vb, cout_lim - pc, 0.f);
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int rloc = rloc0 + j * 16 + lr;
    f32x4 rvv[TN], gtv[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int pc = pc0 + i * 16 + lq * 4;
      rvv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      gtv[i] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (a.rowvec)
        rvv[i] = ld4(a.rowvec + (size_t)(a.rowvec_map ? a.rowvec_map[samp[j]] : samp[j]) * a.rowvec_ld + pc, vr, cout_lim - pc, 0.f);
      if (a.gate)
        gtv[i] = ld4(a.gate + (size_t)(a.gate_map ? a.gate_map[samp[j]] : samp[j]) * a.gate_ld + pc, vg, cout_lim - pc, 1.f);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      f32x4 v;
      int lc;
      if (geglu) {
        if (i & 1) continue;
        lc = ((lc0 + i * 16) >> 1) + lq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (acc[i][j][r] + bv[i][r]) * gelu_erf_f(acc[(i + 1) % TN][j][r] + bv[(i + 1) % TN][r]);
      } else {
        lc = lc0 + i * 16 + lq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[i][j][r] + bv[i][r] + rvv[i][r];
          if (ACT == DC_ACT_SILU) x = silu_f(x);
          if (ACT == DC_ACT_GELU_TANH) x = gelu_tanh_f(x);
          v[r] = x * gtv[i][r];
        }
      }
      *reinterpret_cast<f32x4*>(otile + rloc * old_ + lc) = v;
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void epi_stage(const IgemmArgs& a, f32x4 (&acc)[TN][TM], float* otile, int old_, int rloc0,
                                          int lc0, int pc0, const int (&samp)[TM], int lr, int lq) {
  switch (a.act) {   // wave-uniform
    case DC_ACT_SILU: epi_stage_act<TM, TN, DC_ACT_SILU>(a, acc, otile, old_, rloc0, lc0, pc0, samp, lr, lq); break;
    case DC_ACT_GEGLU: epi_stage_act<TM, TN, DC_ACT_GEGLU>(a, acc, otile, old_, rloc0, lc0, pc0, samp, lr, lq); break;
    case DC_ACT_GELU_TANH: epi_stage_act<TM, TN, DC_ACT_GELU_TANH>(a, acc, otile, old_, rloc0, lc0, pc0, samp, lr, lq); break;
    default: epi_stage_act<TM, TN, DC_ACT_NONE>(a, acc, otile, old_, rloc0, lc0, pc0, samp, lr, lq); break;
  }
}

// RowFn: (int rloc, size_t& out_row, size_t& res_row) -> bool valid   (rows in elements of one channel row)
template <int ES, int UNR, typename RowFn>
__device__ __forceinline__ void epi_store_es(const IgemmArgs& a, const float* otile, int old_, int rows, int tcols, int col0,
                                             int cout_out, RowFn rowfn) {
  const int t = threadIdx.x, nt = blockDim.x;
  const int cpr = tcols / ES;
  const int total = rows * cpr;
  const bool res16 = a.residual && a.res_dtype != DC_F32;
  // vector path: whole 16-byte chunks in and out (a 16-bit residual under an f32 output would be an
  // 8-byte read: left to the element path)
  const bool vec_ok = (cout_out % ES == 0) && (a.out_ld % ES == 0) && (!a.residual || ((a.res_ld % ES == 0) && !(res16 && ES == 4)));
  for (int base = t; base < total; base += UNR * nt) {
    bool ok[UNR]; size_t o[UNR], rr[UNR]; int ch[UNR], rl[UNR];
    chunk16 rc[UNR][2];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int idx = base + u * nt;
      ok[u] = false; o[u] = 0; rr[u] = 0; ch[u] = 0; rl[u] = 0;
      rc[u][0] = chunk16{0u, 0u, 0u, 0u}; rc[u][1] = rc[u][0];
      if (idx < total) {
        rl[u] = idx / cpr; ch[u] = idx - rl[u] * cpr;
        const int c = col0 + ch[u] * ES;
        ok[u] = rowfn(rl[u], o[u], rr[u]) && c < cout_out;
        o[u] = o[u] * a.out_ld + c; rr[u] = rr[u] * a.res_ld + c;
        if (ok[u] && a.residual && vec_ok) {
          const char* rp = reinterpret_cast<const char*>(a.residual) + rr[u] * (res16 ? 2 : 4);
          rc[u][0] = *reinterpret_cast<const chunk16*>(rp);                       // ES elems 16-bit = 16 B (ES 8) / 8 B (ES 4)
          if (!res16 && ES == 8) rc[u][1] = *reinterpret_cast<const chunk16*>(rp + 16);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (!ok[u]) continue;
      const float* src = otile + rl[u] * old_ + ch[u] * ES;
      float v[8];
      const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
      f32x4 hi = {0.f, 0.f, 0.f, 0.f};
      if (ES == 8) hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
      if (vec_ok) {
        if (a.residual) {
          float rf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (a.res_dtype == DC_F32) {
            const f32x4 r0 = __builtin_bit_cast(f32x4, rc[u][0]), r1 = __builtin_bit_cast(f32x4, rc[u][1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { rf[e] = r0[e]; rf[4 + e] = r1[e]; }
          } else if (a.res_dtype == DC_BF16) chunk_to_f<__bf16>(rc[u][0], rf);
          else chunk_to_f<_Float16>(rc[u][0], rf);
#pragma unroll
          for (int e = 0; e < ES; ++e) v[e] += rf[e];
        }
        if (a.out_dtype == DC_F32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + o[u]) = f32x4{v[0], v[1], v[2], v[3]};
          if (ES == 8) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + o[u] + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else if (ES == 8) {
          if (a.out_dtype == DC_BF16) *reinterpret_cast<chunk16*>(reinterpret_cast<__bf16*>(a.out) + o[u]) = f_to_chunk<__bf16>(v);
          else *reinterpret_cast<chunk16*>(reinterpret_cast<_Float16*>(a.out) + o[u]) = f_to_chunk<_Float16>(v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) store_as(a.out, o[u] + e, a.out_dtype, v[e]);
        }
      } else {
        const int c = col0 + ch[u] * ES;
#pragma unroll
        for (int e = 0; e < ES; ++e)
          if (c + e < cout_out) {
            float x = v[e];
            if (a.residual) x += load_as(a.residual, rr[u] + e, a.res_dtype);
            store_as(a.out, o[u] + e, a.out_dtype, x);
          }
      }
    }
  }
}

template <int UNR = 4, typename RowFn>
__device__ __forceinline__ void epi_store(const IgemmArgs& a, const float* otile, int old_, int rows, int tcols, int col0,
                                          int cout_out, RowFn rowfn) {
  // 16-bit outputs: 8 couts per 16-byte chunk; f32 outputs: 4
  if (a.out_dtype == DC_F32) epi_store_es<4, UNR>(a, otile, old_, rows, tcols, col0, cout_out, rowfn);
  else epi_store_es<8, UNR>(a, otile, old_, rows, tcols, col0, cout_out, rowfn);
}
