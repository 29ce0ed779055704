// elementwise.hip — the HBM-bound ends of the scoring step, gfx950:
//   q_sample (+ im2col of conv_in), Philox normal noise, sinusoidal log-SNR embedding,
//   eps-MSE reduction, Haar DWT / inverse.
// Reference lines: diffusion/diffusion_classifier.py:100-117, :690-692 (q_sample),
// :706-711 (v->eps, squared L2 error); utils/wavelet.py:4-68 (Haar); diffusers Timesteps.
#include "common.h"

// ------------------------------------------------------------------ q_sample -------
struct QsArgs {
  const float* x; const float* eps; const float* alpha; const float* sigma; const int32_t* img_of_bj;
  void* out; int out_dtype, n_bj, C, H, W, ld, im2col, patch;
};

template <typename TO>
__global__ __launch_bounds__(256) void qsample_kernel(const QsArgs a) {
  constexpr int EPC = Elem<TO>::EPC;
  const int cpr = a.ld / EPC;                       // chunks per output row
  const int HW = a.H * a.W;
  const int pp = a.im2col == 2 ? a.patch : 1;
  const int gw = a.W / pp, rows_per = (a.H / pp) * gw;   // output rows per sample
  const long long total = (long long)a.n_bj * rows_per * cpr;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int ch = (int)(idx % cpr);
    const long long row = idx / cpr;
    const int p = (int)(row % rows_per);
    const int bj = (int)(row / rows_per);
    const int img = a.img_of_bj ? a.img_of_bj[bj] : bj;
    const float al = a.alpha[bj], sg = a.sigma[bj];
    const int y = p / gw, xw = p - y * gw;
    float f[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int k = ch * EPC + e;
      float v = 0.f;
      if (!a.im2col) {
        if (k < a.C) {
          const size_t o = (size_t)k * HW + p;
          v = al * a.x[(size_t)img * a.C * HW + o] + sg * a.eps[(size_t)bj * a.C * HW + o];
        }
      } else if (a.im2col == 2) {
        if (k < a.C * pp * pp) {
          const int c = k / (pp * pp), r = k - c * pp * pp;
          const size_t o = (size_t)c * HW + (size_t)(y * pp + r / pp) * a.W + (xw * pp + r % pp);
          v = al * a.x[(size_t)img * a.C * HW + o] + sg * a.eps[(size_t)bj * a.C * HW + o];
        }
      } else if (k < 9 * a.C) {
        const int tap = k / a.C, c = k - tap * a.C;
        const int iy = y + tap / 3 - 1, ix = xw + tap % 3 - 1;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          const size_t o = (size_t)c * HW + (size_t)iy * a.W + ix;
          v = al * a.x[(size_t)img * a.C * HW + o] + sg * a.eps[(size_t)bj * a.C * HW + o];
        }
      }
      f[e] = v;
    }
    reinterpret_cast<chunk16*>(a.out)[idx] = f_to_chunk<TO>(f);
  }
}

extern "C" int dc_qsample(const dc_qsample_params* p, dc_stream stream) {
  DC_REQUIRE(p && p->x && p->eps && p->alpha && p->sigma && p->out, DC_ERR_ARG, "dc_qsample: null pointer");
  const int epc = 16 / dc_dtype_size(p->out_dtype);
  DC_REQUIRE(p->n_bj > 0 && p->C > 0 && p->H > 0 && p->W > 0, DC_ERR_SHAPE, "dc_qsample: extents");
  DC_REQUIRE(p->im2col >= 0 && p->im2col <= 2, DC_ERR_ARG, "dc_qsample: im2col=%d", p->im2col);
  int need = p->C, pp = 1;
  if (p->im2col == 1) need = 9 * p->C;
  if (p->im2col == 2) {
    pp = p->patch;
    DC_REQUIRE(pp > 0 && p->H % pp == 0 && p->W % pp == 0, DC_ERR_SHAPE, "dc_qsample: patch=%d does not tile %dx%d", pp, p->H, p->W);
    need = p->C * pp * pp;
  }
  DC_REQUIRE(p->ld % epc == 0 && p->ld >= need, DC_ERR_SHAPE, "dc_qsample: ld=%d (need >= %d, multiple of %d)", p->ld, need, epc);
  QsArgs a{p->x, p->eps, p->alpha, p->sigma, p->img_of_bj, p->out, p->out_dtype, p->n_bj, p->C, p->H, p->W, p->ld, p->im2col, pp};
  const long long total = (long long)p->n_bj * (p->H / pp) * (p->W / pp) * (p->ld / epc);
  const unsigned grid = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p->out_dtype == DC_F32) hipLaunchKernelGGL((qsample_kernel<float>), dim3(grid), dim3(256), 0, s, a);
  else if (p->out_dtype == DC_BF16) hipLaunchKernelGGL((qsample_kernel<__bf16>), dim3(grid), dim3(256), 0, s, a);
  else if (p->out_dtype == DC_F16) hipLaunchKernelGGL((qsample_kernel<_Float16>), dim3(grid), dim3(256), 0, s, a);
  else { dc_set_error("dc_qsample: out_dtype %d", p->out_dtype); return DC_ERR_DTYPE; }
  return dc_check_launch("dc_qsample");
}

// ------------------------------------------------------------------ Philox normal --
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ __launch_bounds__(256) void philox_normal_kernel(float* out, long long rows, long long len4,
                                                            const int64_t* row_ids, uint64_t seed) {
  const long long n4 = rows * len4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long r = i / len4, k = i - r * len4;
    const uint64_t ctr = (uint64_t)(row_ids ? row_ids[r] : r) * (uint64_t)len4 + (uint64_t)k;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float u[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = ((float)(c[e] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float r0 = sqrtf(-2.0f * logf(u[0])), r1 = sqrtf(-2.0f * logf(u[2]));
    float s0, c0, s1, c1;
    sincosf(6.28318530717958647692f * u[1], &s0, &c0);
    sincosf(6.28318530717958647692f * u[3], &s1, &c1);
    reinterpret_cast<float4*>(out)[i] = make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
  }
}

extern "C" int dc_philox_normal(float* out, int64_t rows, int64_t row_len, const int64_t* row_ids, uint64_t seed, dc_stream stream) {
  DC_REQUIRE(out && rows > 0 && row_len > 0 && row_len % 4 == 0, DC_ERR_ARG,
             "dc_philox_normal: rows=%lld row_len=%lld (row_len must be a positive multiple of 4)", (long long)rows, (long long)row_len);
  DC_REQUIRE(((uintptr_t)out & 15) == 0, DC_ERR_ALIGN, "dc_philox_normal: out must be 16-byte aligned");
  const long long n4 = rows * (row_len / 4);
  const unsigned grid = (unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256);
  hipLaunchKernelGGL(philox_normal_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out,
                     (long long)rows, (long long)(row_len / 4), row_ids, seed);
  return dc_check_launch("dc_philox_normal");
}

// ------------------------------------------------------------------ sinusoid -------
__global__ __launch_bounds__(256) void sinusoid_kernel(const float* lam, float* out, int n, int dim, int flip, float shift) {
  const int half = dim / 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * half) return;
  const int r = i / half, k = i - r * half;
  const float ex = (-9.210340371976184f * (float)k) / ((float)half - shift);
  const float arg = lam[r] * expf(ex);
  const float sv = sinf(arg), cv = cosf(arg);
  float* o = out + (size_t)r * dim;
  if (flip) { o[k] = cv; o[half + k] = sv; } else { o[k] = sv; o[half + k] = cv; }
}

extern "C" int dc_sinusoid(const dc_sinusoid_params* p, dc_stream stream) {
  DC_REQUIRE(p && p->lam && p->out && p->n > 0 && p->dim > 0 && p->dim % 2 == 0, DC_ERR_ARG, "dc_sinusoid: bad args");
  const int total = p->n * (p->dim / 2);
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     p->lam, p->out, p->n, p->dim, p->flip_sin_to_cos, p->freq_shift);
  return dc_check_launch("dc_sinusoid");
}

// ------------------------------------------------------------------ eps-MSE --------
// One workgroup per unit; fixed-order accumulation (strided per lane -> xor tree per wave ->
// 4 wave partials added in order) so the result is bit-reproducible run to run.
struct MseArgs {
  const float* pred; const float* eps; const float* x; const float* alpha; const float* sigma;
  const int32_t* bj_of_unit; const int32_t* img_of_bj; const int32_t* out_index; float* out;
  int C, HW, ld, v_param, W, patch;
};

__global__ __launch_bounds__(256) void eps_mse_kernel(const MseArgs a) {
  __shared__ float part[4];
  const int u = blockIdx.x, t = threadIdx.x;
  const int bj = a.bj_of_unit ? a.bj_of_unit[u] : u;
  const int img = a.img_of_bj ? a.img_of_bj[bj] : bj;
  const float al = a.alpha ? a.alpha[bj] : 1.f, sg = a.sigma ? a.sigma[bj] : 0.f;
  const size_t CHW = (size_t)a.C * a.HW;
  const float* pr = a.pred + (size_t)u * (a.patch > 1 ? a.HW / (a.patch * a.patch) : a.HW) * a.ld;
  const float* ep = a.eps + (size_t)bj * CHW;
  const float* xx = a.x ? a.x + (size_t)img * CHW : nullptr;
  float s = 0.f;
  if (a.patch <= 1 && a.C <= 16) {
    // image-shaped prediction with a few channels (UNets: C = 3..12): a lane owns whole pixels, so the C reads of an NHWC
    // row are back to back (cache hits after the first) and the C planes of eps / x are read coalesced across the wave.
    // (The element-indexed loop below walks plane by plane: every pass re-fetched the prediction, 4 bytes per row, and
    // paid a 64-bit division per element — 0.6 TB/s on the 256x256 workloads.)
    for (int p = t; p < a.HW; p += 256) {
      const float* row = pr + (size_t)p * a.ld;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < a.C) {
          const size_t i = (size_t)c * a.HW + p;
          const float e = ep[i];
          float v = row[c];
          if (a.v_param) {
            const float z = al * xx[i] + sg * e;
            v = sg * z + al * v;
          }
          const float d = v - e;
          s += d * d;
        }
    }
  } else
  for (size_t i = t; i < CHW; i += 256) {
    const int c = (int)(i / a.HW), p = (int)(i - (size_t)c * a.HW);
    size_t pi;
    if (a.patch > 1) {
      const int y = p / a.W, xw = p - y * a.W, pp = a.patch;
      pi = ((size_t)(y / pp) * (a.W / pp) + xw / pp) * a.ld + (size_t)((y % pp) * pp + xw % pp) * a.C + c;
    } else {
      pi = (size_t)p * a.ld + c;
    }
    float pv = pr[pi];
    const float e = ep[i];
    if (a.v_param) {
      const float z = al * xx[i] + sg * e;
      pv = sg * z + al * pv;
    }
    const float d = pv - e;
    s += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((t & 63) == 0) part[t >> 6] = s;
  __syncthreads();
  if (t == 0) {
    const float tot = ((part[0] + part[1]) + part[2]) + part[3];
    const float r = sqrtf(tot);   // reference takes torch.norm(...)**2: sqrt, then square
    a.out[a.out_index ? a.out_index[u] : u] = r * r;
  }
}

extern "C" int dc_eps_mse(const dc_eps_mse_params* p, dc_stream stream) {
  DC_REQUIRE(p && p->pred && p->eps && p->out, DC_ERR_ARG, "dc_eps_mse: null pointer");
  const int pp = p->patch > 1 ? p->patch : 1;
  DC_REQUIRE(p->n_units > 0 && p->C > 0 && p->H > 0 && p->W > 0 && p->ld >= p->C * pp * pp, DC_ERR_SHAPE, "dc_eps_mse: extents");
  DC_REQUIRE(p->H % pp == 0 && p->W % pp == 0, DC_ERR_SHAPE, "dc_eps_mse: patch=%d does not tile %dx%d", pp, p->H, p->W);
  if (p->v_param) DC_REQUIRE(p->x && p->alpha && p->sigma, DC_ERR_ARG, "dc_eps_mse: v-param needs x/alpha/sigma");
  MseArgs a{p->pred, p->eps, p->x, p->alpha, p->sigma, p->bj_of_unit, p->img_of_bj, p->out_index, p->out,
            p->C, p->H * p->W, p->ld, p->v_param, p->W, pp};
  hipLaunchKernelGGL(eps_mse_kernel, dim3(p->n_units), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return dc_check_launch("dc_eps_mse");
}

// ------------------------------------------------------------------ sampler step ---
// One ancestral DDPM step with classifier-free guidance (reference diffusion_classifier.py:175-208 ddpm_sampler_step and the
// update at :262-266), fused: the guidance mix, the x-prediction, the clip, the posterior mean and the noise add are one pass over
// the image instead of ~14 elementwise torch launches.  Same operation ORDER as the reference's torch expressions and no
// contraction, so every element equals the torch evaluation of the same fp32 scalars bit for bit — (1 + w) included: the caller
// passes one_plus_w = float(1.0 + w), the Python double rounded ONCE as torch does (1.f + (float)w can differ by an ulp).
struct DdpmArgs {
  const float* z; const float* pred; const float* noise; float* out;
  int C, HW, W, ld, patch, v_param, n;
  float w, one_plus_w, alpha_t, sigma_t, alpha_s, c, sd;
};

__global__ __launch_bounds__(256) void ddpm_step_kernel(const DdpmArgs a) {
#pragma clang fp contract(off)
  const size_t CHW = (size_t)a.C * a.HW, total = CHW * a.n;
  const size_t rows = a.patch > 1 ? (size_t)a.HW / (a.patch * a.patch) : (size_t)a.HW;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / CHW);
    const size_t r = i - (size_t)b * CHW;
    const int c = (int)(r / a.HW), p = (int)(r - (size_t)c * a.HW);
    size_t pi;
    if (a.patch > 1) {
      const int y = p / a.W, xw = p - y * a.W, pp = a.patch;
      pi = ((size_t)(y / pp) * (a.W / pp) + xw / pp) * a.ld + (size_t)((y % pp) * pp + xw % pp) * a.C + c;
    } else {
      pi = (size_t)p * a.ld + c;
    }
    const float pc = a.pred[(size_t)(2 * b) * rows * a.ld + pi], pu = a.pred[(size_t)(2 * b + 1) * rows * a.ld + pi];
    const float zt = a.z[i];
    const float pr = a.one_plus_w * pc - a.w * pu;                              // pred = (1 + w) * pred - w * u_pred
    float xp = a.v_param ? a.alpha_t * zt - a.sigma_t * pr : (zt - a.sigma_t * pr) / a.alpha_t;
    xp = fminf(fmaxf(xp, -1.f), 1.f);                                            // clip
    const float mu = a.alpha_s * (zt * (1.f - a.c) / a.alpha_t + a.c * xp);
    a.out[i] = a.noise ? mu + a.noise[i] * a.sd : fminf(fmaxf(mu, -1.f), 1.f);   // last pass: the clipped mean
  }
}

extern "C" int dc_ddpm_step(const dc_ddpm_step_params* p, dc_stream stream) {
  DC_REQUIRE(p && p->z && p->pred && p->out, DC_ERR_ARG, "dc_ddpm_step: null pointer");
  const int pp = p->patch > 1 ? p->patch : 1;
  DC_REQUIRE(p->n > 0 && p->C > 0 && p->H > 0 && p->W > 0 && p->ld >= p->C * pp * pp, DC_ERR_SHAPE, "dc_ddpm_step: extents");
  DC_REQUIRE(p->H % pp == 0 && p->W % pp == 0, DC_ERR_SHAPE, "dc_ddpm_step: patch=%d does not tile %dx%d", pp, p->H, p->W);
  DdpmArgs a{p->z, p->pred, p->noise, p->out, p->C, p->H * p->W, p->W, p->ld, pp, p->v_param, p->n,
             p->w, p->one_plus_w, p->alpha_t, p->sigma_t, p->alpha_s, p->c, p->sd};
  const size_t total = (size_t)p->n * p->C * p->H * p->W;
  const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return dc_check_launch("dc_ddpm_step");
}

// ------------------------------------------------------------------ Haar -----------
// One lane per 2x2 input block: reads two float2 (rows 2y, 2y+1), writes the four sub-bands.
__global__ __launch_bounds__(256) void haar_dwt2_kernel(const float* in, float* out, long long total, int C, int H, int W, float scale) {
  const int h = H / 2, w = W / 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int xw = (int)(i % w); long long r = i / w;
    const int y = (int)(r % h); r /= h;
    const int c = (int)(r % C); const long long n = r / C;
    const float* src = in + ((n * C + c) * H + 2 * y) * (long long)W + 2 * xw;
    const float2 r0 = *reinterpret_cast<const float2*>(src), r1 = *reinterpret_cast<const float2*>(src + W);
    const float A = r0.x, B = r0.y, Cc = r1.x, D = r1.y;
    float* dst = out + ((n * 4 * C + 4 * c) * h + y) * (long long)w + xw;
    const long long hw = (long long)h * w;
    const float k = 0.5f * scale;
    dst[0] = (A + B + Cc + D) * k;
    dst[hw] = (A + B - Cc - D) * k;
    dst[2 * hw] = (A - B + Cc - D) * k;
    dst[3 * hw] = (A - B - Cc + D) * k;
  }
}
__global__ __launch_bounds__(256) void haar_idwt2_kernel(const float* in, float* out, long long total, int C, int h, int w, float scale) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int xw = (int)(i % w); long long r = i / w;
    const int y = (int)(r % h); r /= h;
    const int c = (int)(r % C); const long long n = r / C;
    const long long hw = (long long)h * w;
    const float* src = in + ((n * 4 * C + 4 * c) * h + y) * (long long)w + xw;
    const float cA = src[0], cH = src[hw], cV = src[2 * hw], cD = src[3 * hw];
    const float k = 0.5f * scale;
    float* dst = out + ((n * C + c) * 2 * h + 2 * y) * (long long)(2 * w) + 2 * xw;
    *reinterpret_cast<float2*>(dst) = make_float2((cA + cH + cV + cD) * k, (cA + cH - cV - cD) * k);
    *reinterpret_cast<float2*>(dst + 2 * w) = make_float2((cA - cH + cV - cD) * k, (cA - cH - cV + cD) * k);
  }
}

typedef float hv4 __attribute__((ext_vector_type(4)));     // native vector: what the nontemporal builtins take
__device__ __forceinline__ hv4 hv4_make(float a, float b, float c, float d) { hv4 v = {a, b, c, d}; return v; }
// Wide forms (W a multiple of 8, 16-byte aligned tensors): a lane owns 4 adjacent output pixels of one subband row — two
// 16-byte loads from each of the two source rows (a wave reads 2 x 2 KiB contiguous), one 16-byte store per subband
// (a wave writes 4 x 1 KiB contiguous).  Pure streaming: 8 bytes moved per input value, HBM-bound.
__global__ __launch_bounds__(256) void haar_dwt2_v4_kernel(const float* __restrict__ in, float* __restrict__ out, long long total4,
                                                           int C, int H, int W, float scale) {
  const int h = H / 2, w4 = W / 8;
  const long long hw = (long long)h * (W / 2);
  const float k = 0.5f * scale;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int xq = (int)(i % w4); long long r = i / w4;
    const int y = (int)(r % h); r /= h;                         // r = n * C + c
    const float* src = in + (r * H + 2 * y) * (long long)W + 8 * xq;
    const hv4 a0 = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(src));
    const hv4 a1 = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(src) + 1);
    const hv4 b0 = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(src + W));
    const hv4 b1 = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(src + W) + 1);
    const float A[4] = {a0.x, a0.z, a1.x, a1.z}, B[4] = {a0.y, a0.w, a1.y, a1.w};
    const float Cc[4] = {b0.x, b0.z, b1.x, b1.z}, D[4] = {b0.y, b0.w, b1.y, b1.w};
    float o[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[0][j] = (A[j] + B[j] + Cc[j] + D[j]) * k;
      o[1][j] = (A[j] + B[j] - Cc[j] - D[j]) * k;
      o[2][j] = (A[j] - B[j] + Cc[j] - D[j]) * k;
      o[3][j] = (A[j] - B[j] - Cc[j] + D[j]) * k;
    }
    const long long c = r % C, n = r / C;
    float* dst = out + ((n * 4 * C + 4 * c) * h + y) * (long long)(W / 2) + 4 * xq;
#pragma unroll
    for (int sb = 0; sb < 4; ++sb)
      __builtin_nontemporal_store(hv4_make(o[sb][0], o[sb][1], o[sb][2], o[sb][3]), reinterpret_cast<hv4*>(dst + sb * hw));
  }
}
__global__ __launch_bounds__(256) void haar_idwt2_v4_kernel(const float* __restrict__ in, float* __restrict__ out, long long total4,
                                                            int C, int h, int w, float scale) {
  const int w4 = w / 4;
  const long long hw = (long long)h * w;
  const float k = 0.5f * scale;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int xq = (int)(i % w4); long long r = i / w4;
    const int y = (int)(r % h); r /= h;
    const long long c = r % C, n = r / C;
    const float* src = in + ((n * 4 * C + 4 * c) * h + y) * (long long)w + 4 * xq;
    const hv4 cA = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(src));
    const hv4 cH = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(src + hw));
    const hv4 cV = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(src + 2 * hw));
    const hv4 cD = __builtin_nontemporal_load(reinterpret_cast<const hv4*>(src + 3 * hw));
    const float a[4] = {cA.x, cA.y, cA.z, cA.w}, hh[4] = {cH.x, cH.y, cH.z, cH.w};
    const float v[4] = {cV.x, cV.y, cV.z, cV.w}, d[4] = {cD.x, cD.y, cD.z, cD.w};
    float t[8], b[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[2 * j] = (a[j] + hh[j] + v[j] + d[j]) * k;  t[2 * j + 1] = (a[j] + hh[j] - v[j] - d[j]) * k;
      b[2 * j] = (a[j] - hh[j] + v[j] - d[j]) * k;  b[2 * j + 1] = (a[j] - hh[j] - v[j] + d[j]) * k;
    }
    float* dst = out + (r * 2 * h + 2 * y) * (long long)(2 * w) + 8 * xq;
    __builtin_nontemporal_store(hv4_make(t[0], t[1], t[2], t[3]), reinterpret_cast<hv4*>(dst));
    __builtin_nontemporal_store(hv4_make(t[4], t[5], t[6], t[7]), reinterpret_cast<hv4*>(dst) + 1);
    __builtin_nontemporal_store(hv4_make(b[0], b[1], b[2], b[3]), reinterpret_cast<hv4*>(dst + 2 * w));
    __builtin_nontemporal_store(hv4_make(b[4], b[5], b[6], b[7]), reinterpret_cast<hv4*>(dst + 2 * w) + 1);
  }
}

static unsigned haar_grid(long long items) {
  const long long b = (items + 255) / 256;
  return (unsigned)(b > 65536 ? 65536 : b);
}

extern "C" int dc_haar_dwt2(const float* in, float* out, int32_t n, int32_t C, int32_t H, int32_t W, float scale, dc_stream stream) {
  DC_REQUIRE(in && out && n > 0 && C > 0 && H > 0 && W > 0, DC_ERR_ARG, "dc_haar_dwt2: bad args");
  DC_REQUIRE(H % 2 == 0 && W % 2 == 0, DC_ERR_SHAPE, "dc_haar_dwt2: H=%d W=%d must be even", H, W);
  DC_REQUIRE(((uintptr_t)in & 7) == 0, DC_ERR_ALIGN, "dc_haar_dwt2: input must be 8-byte aligned");
  const long long total = (long long)n * C * (H / 2) * (W / 2);
  if (W % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
    hipLaunchKernelGGL(haar_dwt2_v4_kernel, dim3(haar_grid(total / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out,
                       total / 4, C, H, W, scale);
    return dc_check_launch("dc_haar_dwt2");
  }
  const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(haar_dwt2_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, total, C, H, W, scale);
  return dc_check_launch("dc_haar_dwt2");
}
extern "C" int dc_haar_idwt2(const float* in, float* out, int32_t n, int32_t C, int32_t h, int32_t w, float scale, dc_stream stream) {
  DC_REQUIRE(in && out && n > 0 && C > 0 && h > 0 && w > 0, DC_ERR_ARG, "dc_haar_idwt2: bad args");
  DC_REQUIRE(((uintptr_t)out & 7) == 0, DC_ERR_ALIGN, "dc_haar_idwt2: output must be 8-byte aligned");
  const long long total = (long long)n * C * h * w;
  if (w % 4 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
    hipLaunchKernelGGL(haar_idwt2_v4_kernel, dim3(haar_grid(total / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out,
                       total / 4, C, h, w, scale);
    return dc_check_launch("dc_haar_idwt2");
  }
  const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(haar_idwt2_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, total, C, h, w, scale);
  return dc_check_launch("dc_haar_idwt2");
}
