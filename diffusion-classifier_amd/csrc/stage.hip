// stage.hip — the stage end of the scoring loop on the device (reference diffusion/diffusion_classifier.py:718-721 and the
// ragged class lists it produces, :671-677 / :695-698): mean over the trials evaluated so far, the k smallest classes per
// image, and the next stage's (pair, class) -> work-unit maps, so a multi-stage / fast classify has no device-to-host copy
// and no host index rebuild between stages.  Tiny kernels (BS x classes x T floats): one wave per image, fixed order.
#include "common.h"

// One wave per image.  Lane l owns classes l, l+64, ...; the mean of a class is the fp32 sum over j = 0 .. t_end-1 in
// ascending order divided by t_end (inf for a class with an unevaluated cell, which is then never kept) — the order
// depends on nothing but (t_end), so every rank and every world size selects the same classes.  Selection: k rounds of a
// wave-wide arg-min on (order key of the mean, class id), ties to the lower class id; output ascending by mean like
// torch.topk(largest=False), which the host path this replaces used (reference :720): a NaN mean (an overflowed f16 forward:
// inf - inf) sorts AFTER +inf there, so it does here — the key maps every NaN to the largest value.  Taken classes are tracked
// in a per-lane bit mask, so each of the k <= C rounds yields a valid class id whatever the values are.
__device__ __forceinline__ uint32_t stage_order_key(float v) {
  if (v != v) return 0xFFFFFFFFu;                          // NaN: last
  const uint32_t u = __builtin_bit_cast(uint32_t, v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // monotone in v (the caller folds -0 into +0 first)
}

template <typename OutT>
__global__ __launch_bounds__(64) void stage_topk_kernel(const float* __restrict__ errors, int C, int T, int t_end, int k,
                                                        OutT* __restrict__ keep, float* __restrict__ means) {
  constexpr int MAXPL = 16;                     // classes per lane: C <= 1024
  const int b = blockIdx.x, lane = threadIdx.x;
  uint32_t m[MAXPL];
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + 64 * i;
    float s = __builtin_inff();
    if (c < C) {
      const float* e = errors + ((size_t)b * C + c) * T;
      s = 0.f;
      for (int j = 0; j < t_end; ++j) s += e[j];
      s = s / (float)t_end;
      if (means) means[(size_t)b * C + c] = s;
    }
    s = s + 0.f;                                // -0 -> +0 (they compare equal in torch: one key)
    m[i] = stage_order_key(s);
  }
  uint32_t taken = 0;                           // bit i: class lane + 64 i was selected in an earlier round
  for (int r = 0; r < k; ++r) {
    uint32_t bv = 0xFFFFFFFFu;
    int bc = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
      const int c = lane + 64 * i;
      if (c < C && !((taken >> i) & 1u) && (m[i] < bv || (m[i] == bv && c < bc))) { bv = m[i]; bc = c; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t ov = (uint32_t)__shfl_xor((int)bv, off, 64);
      const int oc = __shfl_xor(bc, off, 64);
      if (ov < bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
    }
    // k <= C and every untaken class of a lane is a candidate (a NaN key equals the initial bv and wins on c < bc): bc is a class id
    if (lane == 0) keep[(size_t)b * k + r] = (OutT)bc;
    if ((bc & 63) == lane) taken |= 1u << (bc >> 6);
  }
}

extern "C" int dc_stage_topk(const float* errors, int32_t BS, int32_t C, int32_t T, int32_t t_end, int32_t k, int32_t* keep,
                             float* means, dc_stream s) {
  DC_REQUIRE(errors && keep, DC_ERR_ARG, "dc_stage_topk: null errors/keep");
  DC_REQUIRE(BS > 0 && C > 0 && C <= 1024 && T > 0 && t_end > 0 && t_end <= T && k > 0 && k <= C, DC_ERR_SHAPE,
             "dc_stage_topk: BS=%d C=%d (<= 1024) T=%d t_end=%d k=%d", BS, C, T, t_end, k);
  hipLaunchKernelGGL(stage_topk_kernel<int32_t>, dim3(BS), dim3(64), 0, reinterpret_cast<hipStream_t>(s), errors, C, T, t_end, k, keep, means);
  return dc_check_launch("dc_stage_topk");
}

// The last stage keeps one class: labels[b] = arg-min_c mean_j errors[b, c, 0 .. t_end)   (reference :718-725), int64 like the
// reference's LongTensor.
extern "C" int dc_reduce_argmin(const float* errors, int32_t BS, int32_t C, int32_t T, int32_t t_end, int64_t* labels, float* means,
                                dc_stream s) {
  DC_REQUIRE(errors && labels, DC_ERR_ARG, "dc_reduce_argmin: null errors/labels");
  DC_REQUIRE(BS > 0 && C > 0 && C <= 1024 && T > 0 && t_end > 0 && t_end <= T, DC_ERR_SHAPE,
             "dc_reduce_argmin: BS=%d C=%d (<= 1024) T=%d t_end=%d", BS, C, T, t_end);
  hipLaunchKernelGGL(stage_topk_kernel<int64_t>, dim3(BS), dim3(64), 0, reinterpret_cast<hipStream_t>(s), errors, C, T, t_end, 1, labels, means);
  return dc_check_launch("dc_reduce_argmin");
}

// Work-unit maps of the next stage's micro-batches, from the surviving classes keep[BS, k].  This rank's r-th pair of the
// stage is global pair g = rank + r * world: trial j = t0 + g / BS, image b = g % BS (dist.py's round-robin deal).  Micro-batch
// m holds local pairs [m * n_bj, (m+1) * n_bj); a slot past the last pair repeats the micro-batch's first pair and scores into
// the dump cell.  maps[m] = | ctx_of_unit[n_bj * k] | out_index[n_bj * k] |: class id of the unit, flat index of errors[b, class, j].
__global__ __launch_bounds__(256) void stage_maps_kernel(const int32_t* __restrict__ keep, int BS, int C, int T, int k, int t0,
                                                         int n_pairs, int rank, int world, int n_bj, int n_mb, int dump,
                                                         int32_t* __restrict__ maps) {
  const int U = n_bj * k;
  const long long total = (long long)n_mb * U;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / U), u = (int)(i - (long long)m * U);
    const int slot = u / k, c = u - slot * k;
    int r = m * n_bj + slot;
    const bool pad = r >= n_pairs;
    if (pad) r = m * n_bj;
    const long long g = rank + (long long)r * world;
    const int j = t0 + (int)(g / BS), b = (int)(g % BS);
    int cls = keep[(size_t)b * k + c];
    cls = cls < 0 ? 0 : (cls >= C ? C - 1 : cls);       // keep[] comes from dc_stage_topk (always a class id); a foreign list must not index out of errors[]
    int32_t* row = maps + (size_t)m * 2 * U;
    row[u] = cls;
    row[U + u] = pad ? dump : (b * C + cls) * T + j;
  }
}

extern "C" int dc_stage_maps(const int32_t* keep, int32_t BS, int32_t C, int32_t T, int32_t k, int32_t t0, int32_t n_pairs,
                             int32_t rank, int32_t world, int32_t n_bj, int32_t n_mb, int32_t dump, int32_t* maps, dc_stream s) {
  DC_REQUIRE(keep && maps, DC_ERR_ARG, "dc_stage_maps: null keep/maps");
  DC_REQUIRE(BS > 0 && C > 0 && T > 0 && k > 0 && k <= C && t0 >= 0 && t0 < T && n_pairs > 0 && world > 0 && rank >= 0 && rank < world &&
             n_bj > 0 && n_mb > 0 && (long long)(n_mb - 1) * n_bj < n_pairs && (long long)n_mb * n_bj >= n_pairs, DC_ERR_SHAPE,
             "dc_stage_maps: inconsistent extents (BS=%d C=%d T=%d k=%d t0=%d pairs=%d rank=%d/%d n_bj=%d n_mb=%d)", BS, C, T, k, t0,
             n_pairs, rank, world, n_bj, n_mb);
  DC_REQUIRE((long long)BS * C * T < (1LL << 31), DC_ERR_SHAPE, "dc_stage_maps: errors tensor too large for int32 indices");
  const long long last_g = rank + (long long)(n_pairs - 1) * world;
  DC_REQUIRE(t0 + last_g / BS < T, DC_ERR_SHAPE, "dc_stage_maps: the stage's last pair lies beyond trial T-1");
  const long long total = (long long)n_mb * n_bj * k;
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(stage_maps_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(s), keep, BS, C, T, k, t0, n_pairs, rank,
                     world, n_bj, n_mb, dump, maps);
  return dc_check_launch("dc_stage_maps");
}
