// norms.hip — GroupNorm(+SiLU) and LayerNorm(+adaLN modulate) for NHWC activations, gfx950.
//
// Replaces nn.GroupNorm / F.silu / nn.LayerNorm launches inside the diffusers backbone
// (ResnetBlock2D norm1/norm2, Transformer2DModel.norm, conv_norm_out, BasicTransformerBlock
// norm1/norm3, AdaLayerNormZero) reached through reference nets/unet.py:186 / nets/dit.py:49.
// HBM-bound: every transfer is a 16-byte chunk per lane, statistics in fp32, deterministic
// (fixed-order) reductions — no float atomics.
#include <stdlib.h>
#include "common.h"
#include "gn_fold.h"

// ------------------------------------------------------------------ GroupNorm -----
// Statistics are carried as (mean, M2 = sum (v - mean)^2) of value sets — per thread, per split, per producer record — and
// merged with Chan's update in a fixed order; the per-thread sums are SHIFTED by the thread's first value.  The textbook
// sum / sum-of-squares form cancels when |mean| >> std (fp32: var off by ~1e-7 * mean^2 / var relative), which real
// checkpoints with large activation offsets would have turned into parity drift; this form is as robust as torch's own.
// The input may be a channel-concat of two sources (skip connections are never materialised).
struct GnArgs {
  const void* x0; const int32_t* map0; const void* x1; const int32_t* map1;
  void* y; const float* gamma; const float* beta; float* ws;
  int C0, C1, HW, groups, silu, splits, out_dtype; float eps;
  const float* qstats; int qparts;      // statistics formed by the producer (dc_igemm_params.qstats): no statistics sweep
  int wsplits;                          // partial records per sample in ws (= splits, or 1 after gn_qfold_kernel)
};

// merge of many (count, mean, M2) sets in ONE pass without a division per set: shifted by the first set's mean,
// N = sum n, S1 = sum n d, S3 = sum n d^2 (d = mean - pivot), S2 = sum M2  ->  mean = pivot + S1/N, M2 = S2 + S3 - S1^2/N
struct GnMerge {
  float piv = 0.f, N = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f;
  bool have = false;
  __device__ __forceinline__ void add(float n, float mean, float m2) {
    if (n <= 0.f) return;
    if (!have) { piv = mean; have = true; }
    const float d = mean - piv;
    N += n; S1 += n * d; S3 += n * d * d; S2 += m2;
  }
  __device__ __forceinline__ GnAcc result() const {
    GnAcc A;
    if (N > 0.f) { const float iN = 1.0f / N; A.n = N; A.mean = piv + S1 * iN; A.m2 = S2 + fmaxf(S3 - S1 * S1 * iN, 0.f); }
    return A;
  }
};
// pixels [p0, p1) of split k of `splits` over HW pixels (the same formula on the writing and on the reading side)
__device__ __forceinline__ int gn_split_lo(int HW, int k, int splits) { return (int)((long long)HW * k / splits); }

template <typename T>
__device__ __forceinline__ const chunk16* gn_src(const GnArgs& a, int n, int col, int CP0, size_t pix_in_img, int& dummy) {
  // chunk column `col` of the concatenated row -> pointer into the right source
  if (col < CP0) {
    const int ns = a.map0 ? a.map0[n] : n;
    return reinterpret_cast<const chunk16*>(reinterpret_cast<const T*>(a.x0) + ((size_t)ns * a.HW + pix_in_img) * a.C0) + col;
  }
  const int ns = a.map1 ? a.map1[n] : n;
  return reinterpret_cast<const chunk16*>(reinterpret_cast<const T*>(a.x1) + ((size_t)ns * a.HW + pix_in_img) * a.C1) + (col - CP0);
}

// fold the `wsplits` (mean, M2) records of (sample n, group g) in ws, in split order
__device__ __forceinline__ GnAcc gn_fold_ws(const GnArgs& a, int n, int g, int cpg) {
  GnAcc A;
  for (int k = 0; k < a.wsplits; ++k) {
    const float2 v = reinterpret_cast<const float2*>(a.ws)[((size_t)n * a.wsplits + k) * a.groups + g];
    A.add((float)cpg * (float)(gn_split_lo(a.HW, k + 1, a.wsplits) - gn_split_lo(a.HW, k, a.wsplits)), v.x, v.y);
  }
  return A;
}
// Two launches for samples too large for one workgroup.  stats: grid (splits, n); each block reduces its pixel slice of one
// sample to per-group (mean, M2) partials.  apply: same grid; each block folds the `splits` partials of its sample in fixed
// order, then normalises (+affine, +SiLU) its pixel slice.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][PL][C]: per-thread mean, M2
  const int C = a.C0 + a.C1;
  const int CP = C / EPC, CP0 = a.C0 / EPC;
  int TPR = 1; while (TPR < CP && TPR < 256) TPR <<= 1;   // threads per pixel row (pow2 <= 256)
  const int PL = 256 / TPR;                                // pixel lanes
  const int npass = (CP + 255) / 256;                      // column passes when CP > 256
  const int t = threadIdx.x, n = blockIdx.x / a.splits, s = blockIdx.x % a.splits;
  const int p0 = gn_split_lo(a.HW, s, a.splits), p1 = gn_split_lo(a.HW, s + 1, a.splits);
  const int tc = t % TPR, pl = t / TPR;
  float* rs = red; float* rq = red + PL * C;
  int dummy = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int col = pass * 256 + tc;
    if (col >= CP) continue;
    float pv[EPC], sm[EPC], sq[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { pv[e] = 0.f; sm[e] = 0.f; sq[e] = 0.f; }
    int cnt = 0;
    for (int p = p0 + pl; p < p1; p += PL) {
      float f[EPC];
      chunk_to_f<T>(*gn_src<T>(a, n, col, CP0, (size_t)p, dummy), f);
      if (cnt == 0) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) pv[e] = f[e];
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) { const float d = f[e] - pv[e]; sm[e] += d; sq[e] += d * d; }
      ++cnt;
    }
    const float ic = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      rs[pl * C + col * EPC + e] = pv[e] + sm[e] * ic;
      rq[pl * C + col * EPC + e] = fmaxf(sq[e] - sm[e] * sm[e] * ic, 0.f);
    }
  }
  __syncthreads();
  const int cpg = C / a.groups;
  for (int g = t; g < a.groups; g += 256) {
    GnMerge M;
    for (int l = 0; l < PL; ++l) {
      const int left = p1 - p0 - l;
      const float cl = left > 0 ? (float)((left + PL - 1) / PL) : 0.f;      // pixels pixel-lane l walked
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) M.add(cl, rs[l * C + c], rq[l * C + c]);
    }
    const GnAcc A = M.result();
    reinterpret_cast<float2*>(a.ws)[((size_t)n * a.splits + s) * a.groups + g] = float2{A.mean, A.m2};
  }
}

// Large samples with producer statistics: fold the sample's quad records (parts x C/4 of them; 512 parts for a 256x256
// image) into ONE (mean, M2) record per group in ws, in a fixed order, so that the normalise sweep's workgroups do not each
// walk the whole record set.  One workgroup per sample; replaces gn_stats_kernel's sweep of the tensor.  A lane walks its share
// of a quad's parts once, with sums shifted by the first part's mean (equal counts: no division per record).
__global__ __launch_bounds__(256) void gn_qfold_kernel(const GnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [rows][CQ] x (n, mean, M2)
  const int C = a.C0, CQ = C >> 2, n = blockIdx.x, t = threadIdx.x;
  const int cols = CQ < 256 ? CQ : 256, rows = 256 / cols;
  const int ns = a.map0 ? a.map0[n] : n;
  const float2* w = reinterpret_cast<const float2*>(a.qstats) + (size_t)ns * a.qparts * CQ;
  const float nq = 4.0f * (float)a.HW / (float)a.qparts;
  const int r = t / cols, c = t - r * cols;
  if (r < rows)
    for (int q = c; q < CQ; q += cols) {
      float cnt = 0.f, piv = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int p = r; p < a.qparts; p += rows) {
        const float2 v = w[(size_t)p * CQ + q];
        if (cnt == 0.f) piv = v.x;
        const float d = v.x - piv;
        s1 += d; s3 += d * d; s2 += v.y; cnt += 1.f;
      }
      const float ic = cnt > 0.f ? 1.0f / cnt : 0.f;
      red[(r * CQ + q) * 3] = cnt * nq;
      red[(r * CQ + q) * 3 + 1] = piv + s1 * ic;
      red[(r * CQ + q) * 3 + 2] = s2 + nq * fmaxf(s3 - s1 * s1 * ic, 0.f);
    }
  __syncthreads();
  const int qpg = (C / a.groups) >> 2;
  for (int g = t; g < a.groups; g += 256) {
    GnMerge M;
    for (int rr = 0; rr < rows; ++rr)
      for (int q = g * qpg; q < (g + 1) * qpg; ++q) M.add(red[(rr * CQ + q) * 3], red[(rr * CQ + q) * 3 + 1], red[(rr * CQ + q) * 3 + 2]);
    const GnAcc A = M.result();
    reinterpret_cast<float2*>(a.ws)[(size_t)n * a.groups + g] = float2{A.mean, A.m2};
  }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs a) {
  constexpr int EPC = Elem<T>::EPC;     // input chunk elements
  extern __shared__ __attribute__((aligned(16))) float st[];  // mean[groups], rstd[groups]
  const int C = a.C0 + a.C1;
  const int CP = C / EPC, CP0 = a.C0 / EPC;
  int TPR = 1; while (TPR < CP && TPR < 256) TPR <<= 1;
  const int PL = 256 / TPR;
  const int npass = (CP + 255) / 256;
  const int t = threadIdx.x, n = blockIdx.x / a.splits, s = blockIdx.x % a.splits;
  const int cpg = C / a.groups;
  for (int g = t; g < a.groups; g += 256) {
    const GnAcc A = gn_fold_ws(a, n, g, cpg);
    st[g] = A.mean; st[a.groups + g] = rsqrtf(A.var() + a.eps);
  }
  __syncthreads();
  const int p0 = gn_split_lo(a.HW, s, a.splits), p1 = gn_split_lo(a.HW, s + 1, a.splits);
  const int tc = t % TPR, pl = t / TPR;
  int dummy = 0;
  for (int pass = 0; pass < npass; ++pass) {
    const int col = pass * 256 + tc;
    if (col >= CP) continue;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int ch = col * EPC + e, g = ch / cpg;
      const float r = st[a.groups + g] * a.gamma[ch];
      sc[e] = r; sh[e] = a.beta[ch] - st[g] * r;
    }
    for (int p = p0 + pl; p < p1; p += PL) {
      const chunk16 c = *gn_src<T>(a, n, col, CP0, (size_t)p, dummy);
      float f[EPC];
      chunk_to_f<T>(c, f);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float v = f[e] * sc[e] + sh[e];
        if (a.silu) v = silu_t<T>(v);
        f[e] = v;
      }
      TO* o = reinterpret_cast<TO*>(a.y) + ((size_t)n * a.HW + p) * C + (size_t)col * EPC;
      if constexpr (sizeof(TO) * EPC == 16) {
        *reinterpret_cast<chunk16*>(o) = f_to_chunk<TO>(f);
      } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = Elem<TO>::from_f(f[e]);
      }
    }
  }
}

// Normalise sweep when the statistics already exist (the producer's quad records, or one folded record per group in ws):
// short workgroups, each streaming ONE contiguous 16 KiB span of a sample — 256 lanes x 4 chunks, every load issued before
// the first store, non-temporal both ways.  Measured with tools/gn_stream_bench.hip on 8000 x 32x32x128 bf16 (2.1 GB in,
// 2.1 GB out): one 512-lane workgroup looping over a whole sample (gn_image_kernel's sweep) 5.4-5.6 TB/s, a flat grid-stride
// stream 4.6-4.9, these spans 6.4-6.5 — SiLU's VALU work and the statistics fold in front cost nothing at any of them.
// The group statistics are folded in gn_image_kernel's order (parts outer, quads inner), so the two kernels agree bit for bit.
template <typename T, bool NT>
__global__ __launch_bounds__(256) void gn_span_kernel(const GnArgs a, const int spans) {
  constexpr int EPC = Elem<T>::EPC, U = 4, RMAX = 8;             // RMAX x 256 records per sample at most (host check)
  extern __shared__ __attribute__((aligned(16))) float st[];    // mean[groups], rstd[groups], then the sample's records
  const int t = threadIdx.x, n = blockIdx.x / spans, sp = blockIdx.x - n * spans;
  const int C = a.C0, CP = C / EPC;
  const int ns = a.map0 ? a.map0[n] : n;
  const int cpg = C / a.groups;
  float2* const rec = reinterpret_cast<float2*>(st + 2 * a.groups);
  // the statistics records first, every lane ONE independent load per 256 records (a lane per group walking its records one
  // after the other made the workgroup's life a chain of L2 latencies: 16 KiB of payload behind 8-16 dependent loads)
  const int R = a.qstats ? a.qparts * (C >> 2) : a.wsplits * a.groups;
  const float2* const w = a.qstats ? reinterpret_cast<const float2*>(a.qstats) + (size_t)ns * R
                                   : reinterpret_cast<const float2*>(a.ws) + (size_t)n * R;
  float2 rv[RMAX];
#pragma unroll
  for (int k = 0; k < RMAX; ++k) rv[k] = (t + k * 256 < R) ? w[t + k * 256] : float2{0.f, 0.f};
  const chunk16* src = reinterpret_cast<const chunk16*>(reinterpret_cast<const T*>(a.x0) + (size_t)ns * a.HW * C) + (size_t)sp * (256 * U) + t;
  chunk16 c[U];
#pragma unroll
  for (int u = 0; u < U; ++u) c[u] = NT ? __builtin_nontemporal_load(src + u * 256) : src[u * 256];
  const int tc = t & (CP - 1);                                   // CP is a power of two <= 256: the lane's column is the same for its U chunks
  float gm[EPC], bt[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { gm[e] = a.gamma[tc * EPC + e]; bt[e] = a.beta[tc * EPC + e]; }
#pragma unroll
  for (int k = 0; k < RMAX; ++k) if (t + k * 256 < R) rec[t + k * 256] = rv[k];
  __syncthreads();
  for (int g = t; g < a.groups; g += 256) {
    GnAcc A;
    if (a.qstats) A = gn_fold_rec(rec, a.qparts, C >> 2, g, cpg >> 2, 4.0f * (float)a.HW / (float)a.qparts);
    else
      for (int k = 0; k < a.wsplits; ++k) {
        const float2 v = rec[k * a.groups + g];
        A.add((float)cpg * (float)(gn_split_lo(a.HW, k + 1, a.wsplits) - gn_split_lo(a.HW, k, a.wsplits)), v.x, v.y);
      }
    st[g] = A.mean;
    st[a.groups + g] = rsqrtf(A.var() + a.eps);
  }
  __syncthreads();
  float sc[EPC], sh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int g = (tc * EPC + e) / cpg;
    const float r = st[a.groups + g] * gm[e];
    sc[e] = r; sh[e] = bt[e] - st[g] * r;
  }
  chunk16* dst = reinterpret_cast<chunk16*>(reinterpret_cast<T*>(a.y) + (size_t)n * a.HW * C) + (size_t)sp * (256 * U) + t;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float f[EPC];
    chunk_to_f<T>(c[u], f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float v = f[e] * sc[e] + sh[e];
      if (a.silu) v = silu_t<T>(v);
      f[e] = v;
    }
    if (NT) __builtin_nontemporal_store(f_to_chunk<T>(f), dst + u * 256);
    else dst[u * 256] = f_to_chunk<T>(f);
  }
}

// One workgroup per sample: statistics sweep, in-block fold, normalise sweep.  The second sweep re-reads the sample
// while it is still on chip (L2 / Infinity Cache: all resident workgroups together hold ~256 MiB at most), so HBM sees
// the tensor twice (read, write) instead of three times, and the summation order depends on nothing but (HW, C).
// Used for samples of up to 4 MiB (cfg3's 128x128x128 tensors: 17.4 -> 15.9 ms of GroupNorm per step against the split scheme);
// larger ones keep the split two-launch scheme above, which still has parallelism when only a few samples are in flight.
template <typename T>
__global__ __launch_bounds__(512) void gn_image_kernel(const GnArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int NT = 512, UNR = 4;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [2][PL][C] per-thread mean / M2, then mean[groups] / rstd[groups]
  const int C = a.C0 + a.C1;
  const int CP = C / EPC, CP0 = a.C0 / EPC;
  int TPR = 1; while (TPR < CP) TPR <<= 1;                      // CP <= 512 (host check)
  const int PL = NT / TPR;
  const int t = threadIdx.x, n = blockIdx.x;
  const int tc = t % TPR, pl = t / TPR;
  const bool on = tc < CP;
  int dummy = 0;
  const chunk16* src = on ? gn_src<T>(a, n, tc, CP0, 0, dummy) : nullptr;
  const size_t pstride = (size_t)(tc < CP0 ? a.C0 : a.C1) * sizeof(T) / 16;      // chunks per pixel row of the lane's source
  float* rs = red; float* rq = red + PL * C;
  if (on && !a.qstats) {
    float pv[EPC], sm[EPC], sq[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { pv[e] = 0.f; sm[e] = 0.f; sq[e] = 0.f; }
    bool have = false;                                                   // the lane's pivot: its first pixel, taken from the first fetch
    int p = pl;
    for (; p + (UNR - 1) * PL < a.HW; p += UNR * PL) {
      chunk16 c[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) c[u] = src[(size_t)(p + u * PL) * pstride];
      if (!have) { chunk_to_f<T>(c[0], pv); have = true; }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        float f[EPC];
        chunk_to_f<T>(c[u], f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { const float d = f[e] - pv[e]; sm[e] += d; sq[e] += d * d; }
      }
    }
    for (; p < a.HW; p += PL) {
      float f[EPC];
      chunk_to_f<T>(src[(size_t)p * pstride], f);
      if (!have) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) pv[e] = f[e];
        have = true;
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) { const float d = f[e] - pv[e]; sm[e] += d; sq[e] += d * d; }
    }
    const int left = a.HW - pl;
    const float ic = left > 0 ? 1.0f / (float)((left + PL - 1) / PL) : 0.f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      rs[pl * C + tc * EPC + e] = pv[e] + sm[e] * ic;
      rq[pl * C + tc * EPC + e] = fmaxf(sq[e] - sm[e] * sm[e] * ic, 0.f);
    }
  }
  __syncthreads();
  const int cpg = C / a.groups;
  float mean_g = 0.f, rstd_g = 0.f;
  if (t < a.groups) {                                            // groups <= 512 (host check)
    GnAcc A;
    if (a.qstats) {                                              // the producer's quad records, parts in fixed order
      const int ns = a.map0 ? a.map0[n] : n;
      A = gn_fold_rec(reinterpret_cast<const float2*>(a.qstats) + (size_t)ns * a.qparts * (C >> 2), a.qparts, C >> 2, t, cpg >> 2, 4.0f * (float)a.HW / (float)a.qparts);
    } else {
      GnMerge M;
      for (int l = 0; l < PL; ++l) {
        const int left = a.HW - l;
        const float cl = left > 0 ? (float)((left + PL - 1) / PL) : 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) M.add(cl, rs[l * C + c], rq[l * C + c]);
      }
      A = M.result();
    }
    mean_g = A.mean;
    rstd_g = rsqrtf(A.var() + a.eps);
  }
  __syncthreads();
  if (t < a.groups) { red[t] = mean_g; red[a.groups + t] = rstd_g; }
  __syncthreads();
  if (!on) return;
  float sc[EPC], sh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int ch = tc * EPC + e, g = ch / cpg;
    const float r = red[a.groups + g] * a.gamma[ch];
    sc[e] = r; sh[e] = a.beta[ch] - red[g] * r;
  }
  chunk16* dst = reinterpret_cast<chunk16*>(reinterpret_cast<T*>(a.y) + (size_t)n * a.HW * C) + tc;
  const size_t ostride = (size_t)C * sizeof(T) / 16;
  auto norm = [&](const chunk16 cin) {
    float f[EPC];
    chunk_to_f<T>(cin, f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float v = f[e] * sc[e] + sh[e];
      if (a.silu) v = silu_t<T>(v);
      f[e] = v;
    }
    return f_to_chunk<T>(f);
  };
  int p = pl;
  for (; p + (UNR - 1) * PL < a.HW; p += UNR * PL) {
    chunk16 c[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) c[u] = src[(size_t)(p + u * PL) * pstride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) dst[(size_t)(p + u * PL) * ostride] = norm(c[u]);
  }
  for (; p < a.HW; p += PL) dst[(size_t)p * ostride] = norm(src[(size_t)p * pstride]);
}

// Tiny samples (4x4 / 8x8 levels: <= 32 chunks per lane): one WAVE per sample, four samples per workgroup.  The sample
// is read once into registers, the statistics (shifted sums per channel, pivot = the channel's value at pixel 0) are folded
// with xor-shuffles and a wave-private LDS strip, and the normalised chunks are written straight from the registers: no
// workgroup barrier, no second read.
// (The one-workgroup-per-sample kernel ran these 16-32 KiB samples at 1.9 TB/s: three barrier-separated phases of 512
// threads per 16 KiB.)  Summation order depends on (HW, C) only.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void gn_wave_kernel(const GnArgs a, const int n_total) {
  constexpr int EPC = Elem<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) float red[];   // per wave: channel mean[C], channel M2[C], mean[groups], rstd[groups]
  const int C = a.C0 + a.C1;
  const int CP = C / EPC, CP0 = a.C0 / EPC;
  int TPR = 1; while (TPR < CP) TPR <<= 1;                      // CP <= 64 (host check)
  const int PL = 64 / TPR;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wv;
  if (n >= n_total) return;                                     // whole waves leave; no workgroup barrier below
  float* wred = red + wv * (2 * C + 2 * a.groups);
  const int tc = lane % TPR, pl = lane / TPR;
  const bool on = tc < CP;
  int dummy = 0;
  const chunk16* src = on ? gn_src<T>(a, n, tc, CP0, 0, dummy) : nullptr;
  const size_t pstride = (size_t)(tc < CP0 ? a.C0 : a.C1) * sizeof(T) / 16;
  chunk16 c[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int p = pl + i * PL;
    c[i] = (on && p < a.HW) ? src[(size_t)p * pstride] : chunk16{0u, 0u, 0u, 0u};
  }
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const int cpg = C / a.groups;
  // one pass over the registers with SHIFTED sums: the pivot of a channel is its value at pixel 0, which lane tc (pixel lane 0 of
  // the column) holds in its first chunk — one shuffle per channel of the chunk; padding chunks are skipped, so every lane of a
  // column sums (v - pivot) and (v - pivot)^2 of real pixels only and the pixel lanes add up linearly
  float pv[EPC], sm[EPC], sq[EPC];
  {
    float f0[EPC];
    chunk_to_f<T>(c[0], f0);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { pv[e] = __shfl(f0[e], tc, 64); sm[e] = 0.f; sq[e] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (pl + i * PL >= a.HW) continue;
    float f[EPC];
    chunk_to_f<T>(c[i], f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { const float d = f[e] - pv[e]; sm[e] += d; sq[e] += d * d; }
  }
  for (int o = TPR; o < 64; o <<= 1) {                          // over the pixel lanes of the wave
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sm[e] += __shfl_xor(sm[e], o, 64); sq[e] += __shfl_xor(sq[e], o, 64); }
  }
  if (on && pl == 0) {                                          // per channel: mean and M2 over the sample's HW pixels
    const float ihw = 1.0f / (float)a.HW;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      wred[tc * EPC + e] = pv[e] + sm[e] * ihw;
      wred[C + tc * EPC + e] = fmaxf(sq[e] - sm[e] * sm[e] * ihw, 0.f);
    }
  }
  wave_sync();
  for (int g = lane; g < a.groups; g += 64) {                   // equal counts: average of the channel means, M2 merged around it
    float s1 = 0.f, s2 = 0.f;
    for (int ch = g * cpg; ch < (g + 1) * cpg; ++ch) { s1 += wred[ch]; s2 += wred[C + ch]; }
    const float mean = s1 / (float)cpg;
    float s3 = 0.f;
    for (int ch = g * cpg; ch < (g + 1) * cpg; ++ch) { const float d = wred[ch] - mean; s3 += d * d; }
    const float var = (s2 + (float)a.HW * s3) / ((float)cpg * (float)a.HW);
    wred[2 * C + g] = mean;
    wred[2 * C + a.groups + g] = rsqrtf(fmaxf(var, 0.f) + a.eps);
  }
  wave_sync();
  if (!on) return;
  float sc[EPC], sh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int ch = tc * EPC + e, g = ch / cpg;
    const float r = wred[2 * C + a.groups + g] * a.gamma[ch];
    sc[e] = r; sh[e] = a.beta[ch] - wred[2 * C + g] * r;
  }
  chunk16* dst = reinterpret_cast<chunk16*>(reinterpret_cast<T*>(a.y) + (size_t)n * a.HW * C) + tc;
  const size_t ostride = (size_t)C * sizeof(T) / 16;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int p = pl + i * PL;
    if (p >= a.HW) continue;
    float f[EPC];
    chunk16 ci = c[i];
    asm volatile("" : "+v"(ci));          // convert again from the 16-byte chunk: keeping the fp32 copies of the first sweep
    chunk_to_f<T>(ci, f);                 // alive costs 4x the registers (184 instead of ~100 at 16 chunks per lane)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float v = f[e] * sc[e] + sh[e];
      if (a.silu) v = silu_t<T>(v);
      f[e] = v;
    }
    dst[(size_t)p * ostride] = f_to_chunk<T>(f);
  }
}

template <typename T>
static void launch_gn_wave(const GnArgs& a, int n, int nch, size_t lds, hipStream_t s) {
  dim3 g((unsigned)((n + 3) / 4)), b(256);
  if (nch <= 4) hipLaunchKernelGGL((gn_wave_kernel<T, 4>), g, b, lds, s, a, n);
  else if (nch <= 8) hipLaunchKernelGGL((gn_wave_kernel<T, 8>), g, b, lds, s, a, n);
  else if (nch <= 16) hipLaunchKernelGGL((gn_wave_kernel<T, 16>), g, b, lds, s, a, n);
  else hipLaunchKernelGGL((gn_wave_kernel<T, 32>), g, b, lds, s, a, n);
}

// statistics-only mode: fold the split partials and emit the per-(sample, channel) affine for dc_igemm's fused prologue
__global__ __launch_bounds__(256) void gn_finalize_kernel(const GnArgs a, float* out_scale, float* out_shift) {
  __shared__ float st[2 * 64];
  const int C = a.C0 + a.C1, n = blockIdx.x, t = threadIdx.x;
  const int cpg = C / a.groups;
  for (int g = t; g < a.groups; g += 256) {
    const GnAcc A = gn_fold_ws(a, n, g, cpg);
    st[g] = A.mean; st[64 + g] = rsqrtf(A.var() + a.eps);
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    const int g = c / cpg;
    const float r = st[64 + g] * a.gamma[c];
    out_scale[(size_t)n * C + c] = r;
    out_shift[(size_t)n * C + c] = a.beta[c] - st[g] * r;
  }
}

// statistics-only mode with the producer's quad records: no sweep of the tensor at all — fold the records (gn_image_kernel's
// order: parts outer, the group's quads inner, so the affine is bit-identical to what that kernel applies) and emit the
// per-(sample, channel) scale / shift for a consumer that normalises on the fly (dc_igemm gn_scale / gn_shift)
__global__ __launch_bounds__(256) void gn_qaffine_kernel(const GnArgs a, float* out_scale, float* out_shift) {
  extern __shared__ __attribute__((aligned(16))) float st[];    // mean[groups], rstd[groups]
  const int C = a.C0, n = blockIdx.x, t = threadIdx.x;
  const int ns = a.map0 ? a.map0[n] : n;
  const int cpg = C / a.groups;
  for (int g = t; g < a.groups; g += 256) {
    const GnAcc A = gn_fold_rec(reinterpret_cast<const float2*>(a.qstats) + (size_t)ns * a.qparts * (C >> 2), a.qparts, C >> 2, g, cpg >> 2, 4.0f * (float)a.HW / (float)a.qparts);
    st[g] = A.mean;
    st[a.groups + g] = rsqrtf(A.var() + a.eps);
  }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    const int g = c / cpg;
    const float r = st[a.groups + g] * a.gamma[c];
    out_scale[(size_t)n * C + c] = r;
    out_shift[(size_t)n * C + c] = a.beta[c] - st[g] * r;
  }
}

extern "C" int64_t dc_groupnorm_ws_floats(int32_t n, int32_t groups, int32_t splits) {
  return (int64_t)n * groups * splits * 2;
}
extern "C" int32_t dc_groupnorm_splits(int32_t n, int32_t HW, int32_t C) {
  // A function of the spatial size ONLY: the split count fixes the summation order of the
  // statistics, and scores must not depend on how many samples share a launch (micro-batch size,
  // world size).  256 pixels per workgroup, at most 64 splits.
  (void)n; (void)C;
  int s = HW / 256;
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}

extern "C" int dc_groupnorm(const dc_groupnorm_params* p, dc_stream stream) {
  DC_REQUIRE(p && p->x && p->gamma && p->beta && p->ws, DC_ERR_ARG, "dc_groupnorm: null pointer");
  const bool stats_only = p->y == nullptr;
  if (stats_only) DC_REQUIRE(p->out_scale && p->out_shift && (p->groups <= 64 || p->qstats), DC_ERR_ARG, "dc_groupnorm: statistics-only mode needs out_scale/out_shift and groups <= 64");
  const int C1 = p->C1;
  const int C = p->C + C1;
  const int epc = 16 / dc_dtype_size(p->dtype);
  DC_REQUIRE(p->groups > 0 && C % p->groups == 0, DC_ERR_SHAPE, "dc_groupnorm: C=%d groups=%d", C, p->groups);
  DC_REQUIRE(p->C > 0 && p->C % epc == 0 && C1 >= 0 && C1 % epc == 0, DC_ERR_SHAPE,
             "dc_groupnorm: C0=%d C1=%d must be multiples of %d", p->C, C1, epc);
  DC_REQUIRE(p->n > 0 && p->HW > 0 && p->splits > 0 && p->splits <= p->HW, DC_ERR_SHAPE, "dc_groupnorm: n/HW/splits");
  DC_REQUIRE((C1 > 0) == (p->x1 != nullptr), DC_ERR_ARG, "dc_groupnorm: x1/C1 mismatch");
  DC_REQUIRE(p->dtype == p->out_dtype, DC_ERR_DTYPE, "dc_groupnorm: in/out dtype must match");
  GnArgs a;
  a.x0 = p->x; a.map0 = p->map0; a.x1 = p->x1; a.map1 = p->map1; a.y = p->y; a.gamma = p->gamma; a.beta = p->beta; a.ws = p->ws;
  a.C0 = p->C; a.C1 = C1; a.HW = p->HW; a.groups = p->groups; a.silu = p->silu; a.splits = p->splits;
  a.out_dtype = p->out_dtype; a.eps = p->eps;
  a.qstats = nullptr; a.qparts = 0; a.wsplits = p->splits;
  if (p->qstats) {
    DC_REQUIRE(C1 == 0 && p->qparts > 0 && p->HW % p->qparts == 0 && (C / p->groups) % 4 == 0 && ((uintptr_t)p->qstats & 7) == 0, DC_ERR_ARG,
               "dc_groupnorm: qstats needs one source, qparts > 0 dividing HW and (C/groups) %% 4 == 0 (C=%d groups=%d C1=%d HW=%d qparts=%d)", C, p->groups, C1, p->HW, p->qparts);
  }
  const int CP = C / epc;
  int TPR = 1; while (TPR < CP && TPR < 256) TPR <<= 1;
  const int PL = 256 / TPR;
  const size_t lds_stats = (size_t)2 * PL * C * sizeof(float);
  DC_REQUIRE(lds_stats <= 64 * 1024, DC_ERR_SHAPE, "dc_groupnorm: C=%d too large", C);
  const size_t lds_apply = (size_t)2 * p->groups * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long nb = (long long)p->splits * p->n;
  DC_REQUIRE(nb < (1LL << 31), DC_ERR_SHAPE, "dc_groupnorm: grid too large");
  dim3 grid((unsigned)nb), blk(256);
  if (stats_only && p->qstats) {
    DC_REQUIRE(p->groups <= 4096, DC_ERR_SHAPE, "dc_groupnorm: groups=%d", p->groups);
    a.qstats = p->qstats; a.qparts = p->qparts;
    hipLaunchKernelGGL(gn_qaffine_kernel, dim3((unsigned)p->n), blk, lds_apply, s, a, p->out_scale, p->out_shift);
    return dc_check_launch("dc_groupnorm(qaffine)");
  }
  if (stats_only) {
    if (p->dtype == DC_F32) hipLaunchKernelGGL((gn_stats_kernel<float>), grid, blk, lds_stats, s, a);
    else if (p->dtype == DC_BF16) hipLaunchKernelGGL((gn_stats_kernel<__bf16>), grid, blk, lds_stats, s, a);
    else if (p->dtype == DC_F16) hipLaunchKernelGGL((gn_stats_kernel<_Float16>), grid, blk, lds_stats, s, a);
    else { dc_set_error("dc_groupnorm: dtype %d", p->dtype); return DC_ERR_DTYPE; }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(p->n), blk, 0, s, a, p->out_scale, p->out_shift);
    return dc_check_launch("dc_groupnorm(stats)");
  }
  // samples of up to 4 MiB: one workgroup per sample (gn_image_kernel).  The choice depends on (HW, C) only, never
  // on n, so a score does not depend on how many samples share a launch.
  const size_t img_bytes = (size_t)p->HW * C * dc_dtype_size(p->dtype);
  constexpr size_t img_cap = 4u << 20;
  // tiny samples without producer statistics: one wave per sample, register resident (gn_wave_kernel)
  static const bool no_wave = getenv("DCAMD_GN_NO_WAVE") != nullptr;
  if (!no_wave && !p->qstats && CP <= 64 && p->n < (1 << 30)) {
    int tpr = 1; while (tpr < CP) tpr <<= 1;
    const int plw = 64 / tpr, nch = (p->HW + plw - 1) / plw;
    const size_t lds_w = (size_t)4 * (2 * C + 2 * p->groups) * sizeof(float);
    if (nch <= 32 && lds_w <= 64 * 1024) {
      if (p->dtype == DC_F32) launch_gn_wave<float>(a, p->n, nch, lds_w, s);
      else if (p->dtype == DC_BF16) launch_gn_wave<__bf16>(a, p->n, nch, lds_w, s);
      else if (p->dtype == DC_F16) launch_gn_wave<_Float16>(a, p->n, nch, lds_w, s);
      else { dc_set_error("dc_groupnorm: dtype %d", p->dtype); return DC_ERR_DTYPE; }
      return dc_check_launch("dc_groupnorm(wave)");
    }
  }
  // With the producer's statistics the normalise sweep needs no reduction across a sample, so a big sample is better spread
  // over many workgroups (gn_qfold_kernel + gn_apply_kernel, HW/256 splits) than streamed by ONE: the CheXpert / IPMSA plans
  // put only a few hundred 1-4 MiB samples into a launch (cfg3: 4.3 -> 5.x TB/s).  The threshold is a function of (HW, C) only.
  constexpr size_t qsplit_min = 1u << 20;
  const bool qsplit = p->qstats != nullptr && img_bytes >= qsplit_min;
  // producer statistics + a sample that divides into whole 16 KiB spans of one column set: gn_span_kernel (a function of (HW, C) only).
  // Default for samples of 1 MiB and more (the CheXpert / IPMSA plans, where it replaces gn_apply_kernel's split sweep: cfg3 +0.8 %,
  // cfg4 +1.7 % per step); DCAMD_GN_SPAN forces it for every size (tests/test_gpu_ops.py).  For the small samples of cfg2 it
  // is NOT the default: alone it streams 5.9-6.4 TB/s against gn_image_kernel's 5.2-5.6, and inside the scoring step GroupNorm drops
  // from 11.1 to 10.1 ms — but the convolutions that follow slow down by the same amount (77.7 -> 77.9 ms per step): every kernel of
  // that step runs at the socket's power cap (~1.37 kW, 2.08-2.18 GHz by rocm-smi), so a phase that moves the same bytes in less
  // time only hands a hotter chip to the next phase.  DESIGN.md §6c.
  static const bool span_all = getenv("DCAMD_GN_SPAN") != nullptr;
  const bool no_span = !(span_all || qsplit);      // default: only the samples of 1 MiB and more, which had the split apply sweep
  const long long chunks = (long long)p->HW * CP;
  if (!no_span && p->qstats && C1 == 0 && CP <= 256 && (CP & (CP - 1)) == 0 && chunks % 1024 == 0 && p->groups <= 2048 &&
      (long long)p->n * (chunks / 1024) < (1LL << 31)) {
    a.qstats = p->qstats; a.qparts = p->qparts;
    const bool fold = qsplit || (long long)p->qparts * (C >> 2) > 8 * 256;
    if (fold) {                                // many quad records per sample: fold them once per sample, not once per span
      const int CQ = C >> 2, cols = CQ < 256 ? CQ : 256;
      hipLaunchKernelGGL(gn_qfold_kernel, dim3((unsigned)p->n), blk, (size_t)(256 / cols) * CQ * 3 * sizeof(float), s, a);
      a.qstats = nullptr; a.wsplits = 1;
    }
    const int spans = (int)(chunks / 1024);
    const size_t lds_span = lds_apply + (size_t)(fold ? p->groups : p->qparts * (C >> 2)) * sizeof(float2);
    dim3 gs((unsigned)((long long)p->n * spans));
    if (p->dtype == DC_F32) hipLaunchKernelGGL((gn_span_kernel<float, false>), gs, blk, lds_span, s, a, spans);
    else if (p->dtype == DC_BF16) hipLaunchKernelGGL((gn_span_kernel<__bf16, true>), gs, blk, lds_span, s, a, spans);
    else if (p->dtype == DC_F16) hipLaunchKernelGGL((gn_span_kernel<_Float16, false>), gs, blk, lds_span, s, a, spans);
    else { dc_set_error("dc_groupnorm: dtype %d", p->dtype); return DC_ERR_DTYPE; }
    return dc_check_launch("dc_groupnorm(span)");
  }
  if (!qsplit && img_bytes <= img_cap && CP <= 512 && p->groups <= 512 && p->n < (1 << 30)) {
    a.qstats = p->qstats; a.qparts = p->qparts;
    int tpr = 1; while (tpr < CP) tpr <<= 1;
    const size_t lds_img = (size_t)2 * (512 / tpr) * C * sizeof(float);
    dim3 g1((unsigned)p->n), b1(512);
    if (p->dtype == DC_F32) hipLaunchKernelGGL((gn_image_kernel<float>), g1, b1, lds_img, s, a);
    else if (p->dtype == DC_BF16) hipLaunchKernelGGL((gn_image_kernel<__bf16>), g1, b1, lds_img, s, a);
    else if (p->dtype == DC_F16) hipLaunchKernelGGL((gn_image_kernel<_Float16>), g1, b1, lds_img, s, a);
    else { dc_set_error("dc_groupnorm: dtype %d", p->dtype); return DC_ERR_DTYPE; }
    return dc_check_launch("dc_groupnorm(image)");
  }
  // large samples: split scheme; with the producer's quad statistics a small fold launch replaces the statistics sweep
  // (the tensor is read once, not twice)
  const bool sweep = p->qstats == nullptr;
  if (!sweep) {
    a.qstats = p->qstats; a.qparts = p->qparts;
    const int CQ = C >> 2, cols = CQ < 256 ? CQ : 256;
    hipLaunchKernelGGL(gn_qfold_kernel, dim3((unsigned)p->n), blk, (size_t)(256 / cols) * CQ * 3 * sizeof(float), s, a);
    a.qstats = nullptr; a.wsplits = 1;
  }
  if (p->dtype == DC_F32) {
    if (sweep) hipLaunchKernelGGL((gn_stats_kernel<float>), grid, blk, lds_stats, s, a);
    hipLaunchKernelGGL((gn_apply_kernel<float, float>), grid, blk, lds_apply, s, a);
  } else if (p->dtype == DC_BF16) {
    if (sweep) hipLaunchKernelGGL((gn_stats_kernel<__bf16>), grid, blk, lds_stats, s, a);
    hipLaunchKernelGGL((gn_apply_kernel<__bf16, __bf16>), grid, blk, lds_apply, s, a);
  } else if (p->dtype == DC_F16) {
    if (sweep) hipLaunchKernelGGL((gn_stats_kernel<_Float16>), grid, blk, lds_stats, s, a);
    hipLaunchKernelGGL((gn_apply_kernel<_Float16, _Float16>), grid, blk, lds_apply, s, a);
  } else {
    dc_set_error("dc_groupnorm: dtype %d", p->dtype);
    return DC_ERR_DTYPE;
  }
  return dc_check_launch("dc_groupnorm");
}

// ------------------------------------------------------------------ LayerNorm -----
// One wave (64 lanes) per row; three sweeps over the row (mean, centred variance, normalise) —
// the row (<= 8 KB) stays in L1/L2 after the first sweep, so HBM sees it once.
struct LnArgs {
  const void* x; void* y; const float* gamma; const float* beta; const float* scale; const float* shift;
  const int32_t* mod_map; int rows, C, rows_per_sample, mod_ld; float eps;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void ln_kernel(const LnArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int CP = a.C / EPC;
  const chunk16* xr = reinterpret_cast<const chunk16*>(reinterpret_cast<const T*>(a.x) + (size_t)row * a.C);
  float s = 0.f;
  for (int c = lane; c < CP; c += 64) {
    float f[EPC]; chunk_to_f<T>(xr[c], f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) s += f[e];
  }
  const float mean = wave_sum(s) / (float)a.C;
  float q = 0.f;
  for (int c = lane; c < CP; c += 64) {
    float f[EPC]; chunk_to_f<T>(xr[c], f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { const float d = f[e] - mean; q += d * d; }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)a.C + a.eps);
  const float* sc = nullptr; const float* sh = nullptr;
  if (a.scale) {
    const int n = row / a.rows_per_sample;
    const size_t o = (size_t)(a.mod_map ? a.mod_map[n] : n) * a.mod_ld;
    sc = a.scale + o; sh = a.shift + o;
  }
  chunk16* yr = reinterpret_cast<chunk16*>(reinterpret_cast<T*>(a.y) + (size_t)row * a.C);
  for (int c = lane; c < CP; c += 64) {
    float f[EPC]; chunk_to_f<T>(xr[c], f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int ch = c * EPC + e;
      float v = (f[e] - mean) * rstd;
      if (a.gamma) v = v * a.gamma[ch] + a.beta[ch];
      if (sc) v = v * (1.0f + sc[ch]) + sh[ch];
      f[e] = v;
    }
    yr[c] = f_to_chunk<T>(f);
  }
}

// Rows of up to 128 chunks (C <= 1024 16-bit / 512 f32): 16 lanes per row, 4 rows per wave, the
// row lives in registers (read once), statistics by xor-shuffles inside the 16-lane group, affine /
// adaLN parameters fetched as 16-byte vectors.  (The one-wave-per-row kernel above measured 0.9 TB/s.)
// adaLN modulation (DiT): when the workgroup's 16 rows belong to one sample (rows_per_sample % 16 == 0) its scale / shift vectors are
// staged ONCE in LDS — fetched per row from L1 they were 4 x the bytes of the row itself through the vector memory path (768 channels:
// 6 KiB of fp32 parameters against 1.5 KiB of f16 data), and the kernel sat at 4.0 TB/s of read + write.
template <typename T>
__global__ __launch_bounds__(256) void ln16_kernel(const LnArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  __shared__ __attribute__((aligned(16))) float smod[2][1024];
  const int lane = threadIdx.x & 63, l16 = lane & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4);
  const bool live = row < a.rows;
  const int CP = a.C / EPC;
  const bool mod_lds = a.scale != nullptr && a.rows_per_sample % 16 == 0;      // workgroup-uniform
  if (mod_lds) {
    const int n = (blockIdx.x * 16) / a.rows_per_sample;
    const size_t o = (size_t)(a.mod_map ? a.mod_map[n] : n) * a.mod_ld;
    for (int i = threadIdx.x; i < a.C / 4; i += 256) {
      *reinterpret_cast<f32x4*>(&smod[0][4 * i]) = *reinterpret_cast<const f32x4*>(a.scale + o + 4 * i);
      *reinterpret_cast<f32x4*>(&smod[1][4 * i]) = *reinterpret_cast<const f32x4*>(a.shift + o + 4 * i);
    }
    __syncthreads();
  }
  const chunk16* xr = reinterpret_cast<const chunk16*>(reinterpret_cast<const T*>(a.x) + (size_t)(live ? row : 0) * a.C);
  float v[8][EPC];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = l16 + 16 * k;
    if (c < CP) {
      chunk_to_f<T>(xr[c], v[k]);
#pragma unroll
      for (int e = 0; e < EPC; ++e) s += v[k][e];
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (l16 + 16 * k < CP) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { const float d = v[k][e] - mean; q += d * d; }
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)a.C + a.eps);
  if (!live) return;
  const float* sc = nullptr; const float* sh = nullptr;
  if (a.scale) {
    const int n = row / a.rows_per_sample;
    const size_t o = (size_t)(a.mod_map ? a.mod_map[n] : n) * a.mod_ld;
    sc = a.scale + o; sh = a.shift + o;
  }
  chunk16* yr = reinterpret_cast<chunk16*>(reinterpret_cast<T*>(a.y) + (size_t)row * a.C);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = l16 + 16 * k;
    if (c < CP) {
#pragma unroll
      for (int e4 = 0; e4 < EPC; e4 += 4) {
        const int ch = c * EPC + e4;
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f}, ms = {0.f, 0.f, 0.f, 0.f}, mh = {0.f, 0.f, 0.f, 0.f};
        if (a.gamma) { g = *reinterpret_cast<const f32x4*>(a.gamma + ch); b = *reinterpret_cast<const f32x4*>(a.beta + ch); }
        if (mod_lds) { ms = *reinterpret_cast<const f32x4*>(&smod[0][ch]); mh = *reinterpret_cast<const f32x4*>(&smod[1][ch]); }
        else if (sc) { ms = *reinterpret_cast<const f32x4*>(sc + ch); mh = *reinterpret_cast<const f32x4*>(sh + ch); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = (v[k][e4 + e] - mean) * rstd;
          x = x * g[e] + b[e];
          x = x * (1.0f + ms[e]) + mh[e];
          v[k][e4 + e] = x;
        }
      }
      yr[c] = f_to_chunk<T>(v[k]);
    }
  }
}

// Two rows per 16-lane group (round 4): the one-row kernel above keeps 2-6 loads per lane in flight (C = 256 ... 768 in 16-bit) and reduces with
// ds_bpermute shuffles; it moved 3.6-4.3 TB/s.  Here every lane holds its chunks of TWO rows (twice the loads in flight) and the row reductions are
// DPP butterflies (xor 1, xor 2, half-row mirror, row mirror: no LDS crossbar).  Chosen by (dtype, C) only — never by the row count — so a row's
// result does not depend on how many rows share the launch.
template <typename T>
__global__ __launch_bounds__(256) void ln16x2_kernel(const LnArgs a) {
  constexpr int EPC = Elem<T>::EPC, RPG = 2, KMAX = 6;         // C <= 96 chunks of 16 bytes
  __shared__ __attribute__((aligned(16))) float smod[2][1024];
  const int lane = threadIdx.x & 63, l16 = lane & 15;
  const int row0 = (blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (lane >> 4)) * RPG;
  const int CP = a.C / EPC;
  const bool mod_lds = a.scale != nullptr && a.rows_per_sample % (16 * RPG) == 0;      // workgroup-uniform: the workgroup's 32 rows lie in one sample
  if (mod_lds) {
    const int n = (blockIdx.x * 16 * RPG) / a.rows_per_sample;
    const size_t o = (size_t)(a.mod_map ? a.mod_map[n] : n) * a.mod_ld;
    for (int i = threadIdx.x; i < a.C / 4; i += 256) {
      *reinterpret_cast<f32x4*>(&smod[0][4 * i]) = *reinterpret_cast<const f32x4*>(a.scale + o + 4 * i);
      *reinterpret_cast<f32x4*>(&smod[1][4 * i]) = *reinterpret_cast<const f32x4*>(a.shift + o + 4 * i);
    }
    __syncthreads();
  }
  auto row_sum = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
  };
  chunk16 raw[RPG][KMAX];
#pragma unroll
  for (int r = 0; r < RPG; ++r) {
    const int row = row0 + r;
    const chunk16* xr = reinterpret_cast<const chunk16*>(reinterpret_cast<const T*>(a.x) + (size_t)(row < a.rows ? row : 0) * a.C);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      raw[r][k] = chunk16{0u, 0u, 0u, 0u};
      if (l16 + 16 * k < CP) raw[r][k] = xr[l16 + 16 * k];
    }
  }
#pragma unroll
  for (int r = 0; r < RPG; ++r) {
    const int row = row0 + r;
    float v[KMAX][EPC];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (l16 + 16 * k < CP) {
        chunk_to_f<T>(raw[r][k], v[k]);
#pragma unroll
        for (int e = 0; e < EPC; ++e) s += v[k][e];
      }
    const float mean = row_sum(s) / (float)a.C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (l16 + 16 * k < CP) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { const float d = v[k][e] - mean; q += d * d; }
      }
    const float rstd = rsqrtf(row_sum(q) / (float)a.C + a.eps);
    if (row >= a.rows) continue;
    const float* sc = nullptr; const float* sh = nullptr;
    if (a.scale && !mod_lds) {
      const int n = row / a.rows_per_sample;
      const size_t o = (size_t)(a.mod_map ? a.mod_map[n] : n) * a.mod_ld;
      sc = a.scale + o; sh = a.shift + o;
    }
    chunk16* yr = reinterpret_cast<chunk16*>(reinterpret_cast<T*>(a.y) + (size_t)row * a.C);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int c = l16 + 16 * k;
      if (c < CP) {
#pragma unroll
        for (int e4 = 0; e4 < EPC; e4 += 4) {
          const int ch = c * EPC + e4;
          f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f}, ms = {0.f, 0.f, 0.f, 0.f}, mh = {0.f, 0.f, 0.f, 0.f};
          if (a.gamma) { g = *reinterpret_cast<const f32x4*>(a.gamma + ch); b = *reinterpret_cast<const f32x4*>(a.beta + ch); }
          if (mod_lds) { ms = *reinterpret_cast<const f32x4*>(&smod[0][ch]); mh = *reinterpret_cast<const f32x4*>(&smod[1][ch]); }
          else if (sc) { ms = *reinterpret_cast<const f32x4*>(sc + ch); mh = *reinterpret_cast<const f32x4*>(sh + ch); }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = (v[k][e4 + e] - mean) * rstd;
            x = x * g[e] + b[e];
            x = x * (1.0f + ms[e]) + mh[e];
            v[k][e4 + e] = x;
          }
        }
        yr[c] = f_to_chunk<T>(v[k]);
      }
    }
  }
}

extern "C" int dc_layernorm(const dc_layernorm_params* p, dc_stream stream) {
  DC_REQUIRE(p && p->x && p->y, DC_ERR_ARG, "dc_layernorm: null pointer");
  DC_REQUIRE(p->dtype == p->out_dtype, DC_ERR_DTYPE, "dc_layernorm: in/out dtype must match");
  const int epc = 16 / dc_dtype_size(p->dtype);
  DC_REQUIRE(p->rows > 0 && p->C > 0 && p->C % epc == 0, DC_ERR_SHAPE, "dc_layernorm: rows=%d C=%d", p->rows, p->C);
  DC_REQUIRE((p->gamma == nullptr) == (p->beta == nullptr), DC_ERR_ARG, "dc_layernorm: gamma/beta must both be set or null");
  DC_REQUIRE((p->scale == nullptr) == (p->shift == nullptr), DC_ERR_ARG, "dc_layernorm: scale/shift must both be set or null");
  if (p->scale) DC_REQUIRE(p->rows_per_sample > 0 && p->mod_ld >= p->C, DC_ERR_SHAPE, "dc_layernorm: rows_per_sample/mod_ld");
  LnArgs a{p->x, p->y, p->gamma, p->beta, p->scale, p->shift, p->mod_map, p->rows, p->C, p->rows_per_sample, p->mod_ld, p->eps};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool vec16 = p->C / epc <= 128 && (((uintptr_t)p->gamma | (uintptr_t)p->beta | (uintptr_t)p->scale | (uintptr_t)p->shift) & 15) == 0 &&
                     (p->mod_ld % 4 == 0);
  // 16-bit rows of up to 96 chunks (C <= 768): two rows per lane group (a function of dtype and C only)
  if (vec16 && p->dtype != DC_F32 && p->C / epc <= 96) {
    dim3 g32((p->rows + 31) / 32), b32(256);
    if (p->dtype == DC_BF16) hipLaunchKernelGGL((ln16x2_kernel<__bf16>), g32, b32, 0, s, a);
    else hipLaunchKernelGGL((ln16x2_kernel<_Float16>), g32, b32, 0, s, a);
    return dc_check_launch("dc_layernorm");
  }
  if (vec16) {
    dim3 g16((p->rows + 15) / 16), b16(256);
    if (p->dtype == DC_F32) hipLaunchKernelGGL((ln16_kernel<float>), g16, b16, 0, s, a);
    else if (p->dtype == DC_BF16) hipLaunchKernelGGL((ln16_kernel<__bf16>), g16, b16, 0, s, a);
    else if (p->dtype == DC_F16) hipLaunchKernelGGL((ln16_kernel<_Float16>), g16, b16, 0, s, a);
    else { dc_set_error("dc_layernorm: dtype %d", p->dtype); return DC_ERR_DTYPE; }
    return dc_check_launch("dc_layernorm");
  }
  dim3 grid((p->rows + 3) / 4), blk(256);
  if (p->dtype == DC_F32) hipLaunchKernelGGL((ln_kernel<float>), grid, blk, 0, s, a);
  else if (p->dtype == DC_BF16) hipLaunchKernelGGL((ln_kernel<__bf16>), grid, blk, 0, s, a);
  else if (p->dtype == DC_F16) hipLaunchKernelGGL((ln_kernel<_Float16>), grid, blk, 0, s, a);
  else { dc_set_error("dc_layernorm: dtype %d", p->dtype); return DC_ERR_DTYPE; }
  return dc_check_launch("dc_layernorm");
}
