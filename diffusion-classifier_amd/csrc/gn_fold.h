// gn_fold.h — the (count, mean, M2) arithmetic of the GroupNorm kernels (norms.hip): Chan's update and the one-pass fold of a producer's
// quad records.  (The producer-side GroupNorm of csrc/epi_pn.h folds the same records with the same shifted one-pass form, its parts dealt
// to the 16 lanes of a row: pn_fold_row there.)
#pragma once
#include "common.h"

// running (count, mean, M2); add() merges another set (Chan et al.), in the order the caller walks — fixed everywhere below
struct GnAcc {
  float n = 0.f, mean = 0.f, m2 = 0.f;
  __device__ __forceinline__ void add(float n2, float mean2, float m22) {
    if (n2 <= 0.f) return;
    const float nt = n + n2, d = mean2 - mean, w = n2 / nt;
    mean += d * w;
    m2 += m22 + d * d * (n * w);
    n = nt;
  }
  __device__ __forceinline__ float var() const { return n > 0.f ? fmaxf(m2 / n, 0.f) : 0.f; }     // a division: never contracted
};
// fold the producer's quad records of group g (rec: [part][C/4] float2 (mean, M2) of the sample, each over nq = 4 * HW / qparts
// values): equal counts, so mean = average of the means and M2 = sum M2_r + nq * sum (mean_r - mean)^2.  One pass, parts outer,
// the group's quads inner, the means shifted by the first record's (what is left inside the shifted sums is the spread of the
// record means, which is part of the group's variance): as many loads as the sum / sum-of-squares form had, no division per record
__device__ __forceinline__ GnAcc gn_fold_rec(const float2* rec, int qparts, int CQ, int g, int qpg, float nq) {
  // inlined into four kernels (image / span / affine sweeps) that must agree bit for bit: no implicit contraction
#pragma clang fp contract(off)
  const float piv = rec[g * qpg].x;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int part = 0; part < qparts; ++part)
    for (int q = g * qpg; q < (g + 1) * qpg; ++q) {
      const float2 v = rec[(size_t)part * CQ + q];
      const float d = v.x - piv;
      s1 += d; s3 = __builtin_fmaf(d, d, s3); s2 += v.y;
    }
  const float R = (float)(qparts * qpg), iR = 1.0f / R;
  GnAcc A;
  A.n = R * nq;
  A.mean = __builtin_fmaf(s1, iR, piv);
  A.m2 = __builtin_fmaf(nq, fmaxf(__builtin_fmaf(-s1 * s1, iR, s3), 0.f), s2);
  return A;
}

