/* dcamd.h — C-ABI of the MI355X (gfx950) diffusion-classifier scoring path.
 *
 * One shared library, libdcamd.so, plain C entry points, plain pointers and sizes, no
 * torch types.  Every entry point enqueues HIP kernels on the caller's stream and
 * returns; nothing allocates, frees or synchronises.  The caller owns every buffer.
 *
 * What each entry point replaces in the reference (faverogian/diffusion-classifier):
 *   dc_qsample          diffusion/diffusion_classifier.py:100-117 (diffuse) + :690-692
 *   dc_philox_normal    torch.randn_like at diffusion_classifier.py:113 (throughput mode)
 *   dc_sinusoid         diffusers Timesteps behind nets/unet.py:187 / nets/dit.py:50
 *   dc_igemm            every Conv2d 3x3/1x1 and Linear of the diffusers backbone behind
 *                       nets/unet.py:186-195 and nets/dit.py:49-51 (implicit GEMM on MFMA)
 *   dc_groupnorm        GroupNorm(+SiLU) sites of ResnetBlock2D / Transformer2DModel
 *   dc_layernorm        LayerNorm sites of BasicTransformerBlock; adaLN-Zero modulate (DiT)
 *   dc_attention        F.scaled_dot_product_attention self-attention (attn1)
 *   dc_eps_mse          diffusion_classifier.py:706-711 (v->eps, torch.norm(...)**2)
 *   dc_haar_dwt2/idwt2  utils/wavelet.py:4-35 / :37-68
 *   dc_ddpm_step        diffusion_classifier.py:175-208 (ddpm_sampler_step) + :262-266, one fused pass per sampling step
 *   dc_stage_topk / dc_reduce_argmin / dc_stage_maps
 *                       the stage end, diffusion_classifier.py:718-725 (mean over trials, k smallest classes) and the
 *                       per-image surviving-class lists of the next stage (:695-698, ragged after pruning / fast mode
 *                       :671-677) as device-side work-unit maps
 *   dc_run_plan         the Python double loop body, diffusion_classifier.py:695-714, as
 *                       one native launch sequence (graph-capturable)
 *
 * Errors: 0 = ok; negative dc_status otherwise; text via dc_last_error() (thread-local).
 * No exception crosses the ABI.  HIP launch errors are returned as DC_ERR_LAUNCH.
 */
#ifndef DCAMD_H
#define DCAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: qstats records are (mean, M2) sets (version 1: sum, sum of squares) and qparts must divide HW
 * 3: dc_ddpm_step_params.one_plus_w; dc_attention requires scale > 0; dc_igemm_params.pn_* (producer-side GroupNorm) */
#define DC_ABI_VERSION 4

typedef void* dc_stream; /* hipStream_t */

typedef enum { DC_OK = 0, DC_ERR_ARG = -1, DC_ERR_SHAPE = -2, DC_ERR_DTYPE = -3,
               DC_ERR_ALIGN = -4, DC_ERR_LAUNCH = -5, DC_ERR_UNSUPPORTED = -6 } dc_status;

typedef enum { DC_F32 = 0, DC_BF16 = 1, DC_F16 = 2 } dc_dtype;

typedef enum { DC_ACT_NONE = 0, DC_ACT_SILU = 1, DC_ACT_GEGLU = 2, DC_ACT_GELU_TANH = 3 } dc_act;

int dc_abi_version(void);
const char* dc_last_error(void);
/* "gfx950" — the only code object in the library. */
const char* dc_arch(void);

/* ---------------------------------------------------------------- q_sample ------- */
/* z[bj] = alpha[bj]*x[img[bj]] + sigma[bj]*eps[bj]          (reference :115)
 * x   [n_img, C, H, W] f32 NCHW;  eps [n_bj, C, H, W] f32 NCHW.
 * out: im2col==0: z as NHWC [n_bj, H, W, ld] (channels >= C are written as zero)
 *      im2col==1: 3x3 zero-padded patches [n_bj, H, W, ld], k = tap*C + c for
 *                 tap = ky*3+kx, k in [9C, ld) zero — the A operand of conv_in as a GEMM.
 * out_dtype: DC_F32 / DC_BF16 / DC_F16. */
typedef struct {
  const float* x; const float* eps; const float* alpha; const float* sigma;
  const int32_t* img_of_bj; /* [n_bj] or NULL (identity) */
  void* out; int32_t out_dtype;
  int32_t n_bj, C, H, W, ld, im2col;
  int32_t patch;            /* im2col==2: non-overlapping patch x patch tokens [n_bj, H/p, W/p, ld],
                               k = c*p*p + py*p + px (the A operand of DiT's patch embedding) */
} dc_qsample_params;
int dc_qsample(const dc_qsample_params* p, dc_stream s);

/* out[r, i] ~ N(0,1) for r < rows, i < row_len (multiple of 4): Philox4x32-10 with key = seed and
 * counter = row_ids[r]*(row_len/4) + i/4 (row_ids NULL: r), Box-Muller.  The value of a row
 * depends only on (seed, row id), so sharding rows across GPUs never changes the noise. */
int dc_philox_normal(float* out, int64_t rows, int64_t row_len, const int64_t* row_ids, uint64_t seed, dc_stream s);

/* ---------------------------------------------------------------- sinusoid ------- */
/* emb[i, :] = [cos(lam_i*w_k), sin(lam_i*w_k)] (flip) or [sin, cos]; w_k = exp(-ln(1e4)*k/(half-shift)) */
typedef struct { const float* lam; float* out; int32_t n, dim, flip_sin_to_cos; float freq_shift; } dc_sinusoid_params;
int dc_sinusoid(const dc_sinusoid_params* p, dc_stream s);

/* ---------------------------------------------------------------- igemm ---------- */
/* Implicit GEMM on MFMA: conv 3x3 (pad 1, stride 1|2, optional nearest-2x upsample folded
 * into the gather, optional channel-concat of two sources) or a plain GEMM / 1x1 conv
 * (taps = 1).  Activations are NHWC: rows = n_img*Hout*Wout, K = taps*(C0+C1).
 *   out[row, co] = epilogue( sum_k A[row,k] * Wp[co,k] )
 * epilogue: (+bias[co]) (+rowvec[vmap[n]][co]) -> act -> (*gate[gmap[n]][co]) (+residual[row,co])
 * A/W dtype = dtype (f32 uses v_mfma_f32_16x16x4_f32; bf16/f16 use 16x16x32).
 * W is packed [Cout_pad][K] (K contiguous, k = tap*(C0+C1) + c), Cout_pad = multiple of the
 * kernel's N tile (dc_igemm_cout_pad), zero filled.  For DC_ACT_GEGLU the packed rows
 * interleave 16-row blocks of the value half and the gate half (see dc_igemm docs in
 * DESIGN.md) and the output has Cout/2 channels.
 * Constraints: C0, C1 multiples of 128/sizeof(dtype) elements (32 f32 / 64 bf16,f16). */
typedef struct {
  int32_t dtype, taps, stride, upsample;
  int32_t n_img, Hin, Win, Hout, Wout;      /* Hin/Win: conv input size AFTER upsample */
  const void* src0; const int32_t* map0; int32_t C0, ld0;  /* ld: pixel stride in elements (0: = C) */
  const void* src1; const int32_t* map1; int32_t C1, ld1;
  const void* W; int32_t Cout, tile_n;      /* tile_n: 128 or 32 (packing granularity) */
  const float* bias;
  const float* rowvec; const int32_t* rowvec_map; int32_t rowvec_ld, act;
  const float* gate; const int32_t* gate_map; int32_t gate_ld, pad2_;
  const void* residual; const int32_t* res_map; int32_t res_dtype, res_ld;
  void* out; int32_t out_dtype, out_ld;
  /* fused GroupNorm(+SiLU) prologue on the A operand: a[n,y,x,c] := act(a*gn_scale[n][c] + gn_shift[n][c]) for
   * real pixels (conv padding stays 0).  Only where dc_igemm_gn_fusable() says so (3x3 halo kernel, one sample
   * per workgroup, C0+C1 <= 512; or the thin-output 3x3 conv, Cout <= 16 on images >= 16x16: conv_norm_out + conv_out
   * as one launch, rounded exactly as the GroupNorm kernel rounds); NULL otherwise. */
  const float* gn_scale; const float* gn_shift; int32_t gn_silu, pad3_;
  /* optional 1x1 side source summed into the same output (a ResNet's conv_shortcut folded into its conv2):
   * out += sum_c A2[row, c] * W2p[co, c], A2 = src2 [*, Hout, Wout, C2] read through map2, W2p packed [Cout_pad][C2].
   * Only where dc_igemm_side_ok() says so (3x3 stride-1 halo kernel, no upsample); NULL / 0 otherwise. */
  const void* src2; const int32_t* map2; const void* W2; int32_t C2, ld2;
  /* ln_eps > 0: each A row is LayerNorm-ed on the fly, a := (a - mean(a)) * rsqrt(var(a) + ln_eps) over its K channels, no
   * affine (fold gamma into W's columns and beta into the bias when packing).  Only where dc_igemm_ln_ok() says so (the
   * activation-stationary GEMM: 1 tap, one source, K <= 512, 16-bit). */
  float ln_eps;
  /* up4 = 1 (only with upsample = 1, where dc_igemm_up4_ok() says so): W holds the four-phase form of the 3x3 weights,
   * [phase = 2a+b][Cout_pad][(dy*2+dx) * C + c], in which output pixel (2y+a, 2x+b) = sum over the 2x2 source pixels
   * (y+a-1+dy, x+b-1+dx) — the 3x3 taps that read the same source pixel of the nearest-2x upsampled image are summed
   * when packing (row taps: a=0 -> (k0 | k1+k2), a=1 -> (k0+k1 | k2); columns alike).  4/9 of the MACs, same result up
   * to the rounding of the summed weights. */
  int32_t up4;
  /* quad statistics of the OUTPUT for a following GroupNorm (dc_groupnorm_params.qstats): per output sample n, per part (a run
   * of HW / qparts pixels) and per quad q of 4 consecutive output channels the mean and the centred second moment
   * M2 = sum (v - mean)^2 of the quad's 4 * HW / qparts output values (fp32, before the rounding to out_dtype; shifted sums, no
   * cancellation for |mean| >> std): qstats[((n * qparts + part) * (Cout/4) + q) * 2 + (0 = mean | 1 = M2)],
   * qparts = dc_igemm_qstats_parts().  The GroupNorm then streams the tensor once instead of reading it twice.  NULL otherwise. */
  float* qstats;
  /* producer-side GroupNorm (only where dc_igemm_pn_ok() says so: 3x3 stride-1 conv on power-of-two images of 16x16 ... 64x64, one
   * source, Cout a multiple of 128): besides — or, with out == NULL, instead of — the raw output v the conv stores
   *   pn_out[row, co] = act( (v - mean_g) * rstd_g * pn_gamma[co] + pn_beta[co] ),   act = SiLU if pn_silu else identity,
   * the GroupNorm over pn_groups groups of Cout / pn_groups consecutive channels and all Hout*Wout pixels of the sample (fp32
   * statistics from the fp32 accumulators, the same (mean, M2) quad records and the same fold dc_groupnorm uses; eps = pn_eps), in
   * the compute dtype with row stride pn_ld (0: = Cout).  The consumer of that GroupNorm then reads pn_out with a plain conv / GEMM
   * and the GroupNorm pass over the tensor never runs.  qstats must be given (dc_igemm_qstats_parts() parts per sample: the workgroups
   * of a sample exchange their records through it); pn_cnt: n_img * ceil(Cout / 128) uint32 arrival counters that belong to THIS
   * call site: zeroed once by the caller, then left alone — they only ever grow (every launch adds Hout*Wout/256 to each), which is
   * what makes a stale read of one harmless.  A wait that never completes cannot hang the device: it times out and is counted (dc_pn_timeouts). */
  void* pn_out; const float* pn_gamma; const float* pn_beta; uint32_t* pn_cnt;
  int32_t pn_ld, pn_groups, pn_silu; float pn_eps;
} dc_igemm_params;
int dc_igemm(const dc_igemm_params* p, dc_stream s);
int32_t dc_igemm_cout_pad(int32_t cout, int32_t tile_n);
/* Name of the kernel dc_igemm would launch for these parameters, e.g. "conv3_halo<bf16,4w>",
 * "igemm_pipe<bf16,256x128,3st>" (measurement / profiling only; static string). */
const char* dc_igemm_variant(const dc_igemm_params* p);
/* 1 when dc_igemm can take gn_scale/gn_shift for this problem (the other fields as for dc_igemm). */
int32_t dc_igemm_gn_fusable(const dc_igemm_params* p);
/* 1 when dc_igemm can take the 1x1 side source src2 / W2 for this problem. */
int32_t dc_igemm_side_ok(const dc_igemm_params* p);
/* 1 when dc_igemm can take ln_eps (row LayerNorm of the A operand) for this problem. */
int32_t dc_igemm_ln_ok(const dc_igemm_params* p);
/* > 0: dc_igemm can emit qstats for this problem, with that many parts per sample (3x3 halo kernel, output stored in the
 * compute type, Cout a multiple of 8); 0: it cannot. */
int32_t dc_igemm_qstats_parts(const dc_igemm_params* p);
/* 1 when dc_igemm can take pn_out / pn_groups (producer-side GroupNorm) for this problem. */
int32_t dc_igemm_pn_ok(const dc_igemm_params* p);
/* Number of waves whose wait inside a producer-side-GroupNorm launch timed out since the last call (0 = every launch was sound);
 * reads and clears a device-side counter: SYNCHRONISES with the device.  Non-zero means results of those launches are invalid. */
int32_t dc_pn_timeouts(void);
/* 1 when dc_igemm can take up4 = 1 (four-phase upsample conv) for this problem. */
int32_t dc_igemm_up4_ok(const dc_igemm_params* p);

/* ---------------------------------------------------------------- weight packing - */
/* dc_igemm consumes weights as [Cout_pad][K] in the compute dtype, K contiguous, rows zero-padded to the N tile
 * (Cout_pad = dc_igemm_cout_pad(cout, tile_n)).  These entry points build that form on the device from the framework's fp32
 * parameter tensors (device pointers, row-major, diffusers layouts); run once per (weights, dtype).  All buffers caller-owned;
 * dc_packed_bytes gives the size of a packed [Cout_pad][K] buffer.
 *   matrix   Linear / Conv2d 1x1 weight [cout, K]: out[r][k] = w[row_perm ? row_perm[r] : r][k] * (col_scale ? col_scale[k] : 1),
 *            columns K..kpad-1 zero (kpad >= K: conv_in's K = 9*Cin is padded to the 128-byte K granule).  col_scale folds a
 *            LayerNorm gamma into the consumer GEMM (dc_igemm ln_eps); dc_fold_layernorm_bias gives the matching bias W beta + c.
 *   conv3x3  Conv2d weight [cout, cin, 3, 3], input-channel slice [c_lo, c_hi): out[r][tap*C + c], tap = ky*3 + kx, C = c_hi - c_lo
 *            (a slice: the two halves of a skip-connection conv, conv(cat(a, b)) = conv_a(a) + conv_b(b)).
 *   up4      the four-phase form of "nearest-2x upsample, then 3x3 conv" (dc_igemm_params.up4): out[2a+b][r][(dy*2+dx)*cin + c] =
 *            fp32 sum of the 3x3 taps that read source pixel (y+a-1+dy, x+b-1+dx) — rows a=0: {k0} | {k1,k2}, a=1: {k0,k1} | {k2},
 *            columns alike; buffer = 4 x dc_packed_bytes(cout, 4*cin).
 *   geglu    GEGLU projection [2*n_half, K] (+ bias [2*n_half]): packed row 32*blk + i = value row 16*blk + i (i < 16), gate row
 *            n_half + 16*blk + i - 16 otherwise, so value and gate of a channel meet in one lane of the epilogue; optional LayerNorm
 *            fold (gamma into the columns, out_bias = permuted (bias + W beta)); perm_ws: int32[2*n_half] scratch. */
int64_t dc_packed_bytes(int32_t cout, int32_t K, int32_t dtype, int32_t tile_n);
int dc_pack_weights_matrix(const float* w, int32_t cout, int32_t K, int32_t kpad, const int32_t* row_perm, const float* col_scale,
                           void* out, int32_t dtype, int32_t tile_n, dc_stream s);
int dc_pack_weights_conv3x3(const float* w, int32_t cout, int32_t cin, int32_t c_lo, int32_t c_hi, int32_t kpad, void* out,
                            int32_t dtype, int32_t tile_n, dc_stream s);
int dc_pack_weights_up4(const float* w, int32_t cout, int32_t cin, void* out, int32_t dtype, int32_t tile_n, dc_stream s);
int dc_pack_weights_geglu(const float* w, const float* bias, int32_t n_half, int32_t K, const float* ln_gamma, const float* ln_beta,
                          void* out_w, float* out_bias, int32_t* perm_ws, int32_t dtype, dc_stream s);
/* out_bias[r] = (bias ? bias[r] : 0) + sum_k w[r][k] * ln_beta[k]   (fp32, k ascending) */
int dc_fold_layernorm_bias(const float* w, const float* bias, const float* ln_beta, int32_t cout, int32_t K, float* out_bias, dc_stream s);

/* ---------------------------------------------------------------- norms ---------- */
/* GroupNorm over (C/groups)*HW per (sample, group), NHWC, optional SiLU.  fp32 statistics carried as (mean, M2) sets
 * merged with Chan's update in a fixed order (shifted per-thread sums): no sum / sum-of-squares cancellation.
 * ws: float workspace >= dc_groupnorm_ws_floats(n, groups, splits). */
typedef struct {
  const void* x; const int32_t* map0;   /* source 0: [*, HW, C] ; map: sample -> source sample or NULL */
  const void* x1; const int32_t* map1;  /* optional source 1 (channel concat after source 0), C1 channels */
  void* y; int32_t dtype, out_dtype;    /* y: [n, HW, C+C1] */
  int32_t n, HW, C, C1, groups, silu, splits; float eps;
  const float* gamma; const float* beta; float* ws;
  /* statistics-only mode (y == NULL): instead of normalising, write the per-(sample, channel) affine
   * out_scale[n][C+C1] = rstd*gamma and out_shift = beta - mean*rstd*gamma, which dc_igemm applies on the
   * fly (gn_scale / gn_shift) — the normalised tensor is then never written to HBM. */
  float* out_scale; float* out_shift;
  /* statistics already formed by the producer of x (dc_igemm_params.qstats; qparts parts per sample, qparts divides HW):
   * single source (C1 == 0), (C/groups) a multiple of 4.  The statistics sweep is skipped; in statistics-only mode the tensor
   * is not read at all (x is then only the sample count's witness).  NULL / 0 otherwise. */
  const float* qstats; int32_t qparts, pad_;
} dc_groupnorm_params;
int dc_groupnorm(const dc_groupnorm_params* p, dc_stream s);
int64_t dc_groupnorm_ws_floats(int32_t n, int32_t groups, int32_t splits);
int32_t dc_groupnorm_splits(int32_t n, int32_t HW, int32_t C);
/* Workspace bytes per op (only GroupNorm needs one; the others keep everything in registers / LDS and return 0). */
int64_t dc_workspace_bytes_groupnorm(const dc_groupnorm_params* p);
int64_t dc_workspace_bytes_igemm(const dc_igemm_params* p);

/* LayerNorm over C per row. gamma/beta may be NULL.  If scale/shift given (adaLN):
 * y = ln(x)*(1+scale[m[n]][c]) + shift[m[n]][c], n = row / rows_per_sample. */
typedef struct {
  const void* x; void* y; int32_t dtype, out_dtype;
  int32_t rows, C, rows_per_sample, mod_ld; float eps;
  const float* gamma; const float* beta;
  const float* scale; const float* shift; const int32_t* mod_map;
} dc_layernorm_params;
int dc_layernorm(const dc_layernorm_params* p, dc_stream s);
int64_t dc_workspace_bytes_layernorm(const dc_layernorm_params* p);

/* ---------------------------------------------------------------- attention ------ */
/* softmax(q k^T * scale) v per (sample, head).  q/k/v: [n, L, heads, d] with row stride
 * ld (elements) so a fused QKV GEMM output can be passed as three offset pointers.
 * scale must be > 0 (DC_ERR_ARG otherwise): the kernels take the running max on the raw scores. */
typedef struct {
  const void* q; const void* k; const void* v; void* out;
  int32_t dtype, n, L, heads, d, ld_qkv, ld_out; float scale;
} dc_attention_params;
int dc_attention(const dc_attention_params* p, dc_stream s);
int64_t dc_workspace_bytes_attention(const dc_attention_params* p);

/* ---------------------------------------------------------------- transformer block, attention half --- */
/* One launch for the self-attention half of a UNet transformer block (the backbone behind /root/reference/nets/unet.py:186-195:
 * Transformer2DModel.proj_in -> BasicTransformerBlock.norm1 -> attn1 (to_q/k/v, softmax, to_out) -> + attn2's class vector -> residual):
 *   h = x Wp^T + bp;  hn = LayerNorm(h; eps) * ln_g + ln_b;  q | k | v = hn Wqkv^T (rows [0,C) q, [C,2C) k, [2C,3C) v; head i = channels
 *   [i d, (i+1) d), d = C / heads);  o = softmax(q k^T * scale) v per (sample, head);  out = ((o Wo^T + bo) + rowvec[rowvec_map[n]]) + h.
 * x [n, L, ldx], out [n, L, ld_out] in `dtype` (16-bit); Wp / Wo [C][C], Wqkv [3C][C] packed as dc_igemm takes them (dc_pack_weights_matrix,
 * tile_n 128); biases, LayerNorm affine and rowvec fp32.  h, hn, q, k, v, p, o are rounded to `dtype` where the separate launches
 * (dc_igemm, dc_layernorm, dc_attention) store them, so results agree with that chain to accumulation order.
 * Shapes: dc_tblock_front_ok() (L = 64, C = 256, heads = 4 or 8 today); anything else returns DC_ERR_SHAPE. */
typedef struct {
  const void* x; const void* Wp; const float* bp;
  const float* ln_g; const float* ln_b;
  const void* Wqkv; const void* Wo; const float* bo;
  const float* rowvec; const int32_t* rowvec_map;
  void* out;
  int32_t dtype, n, L, C, heads, ldx, ld_out, rowvec_ld;
  float ln_eps, scale;
} dc_tblock_front_params;
int dc_tblock_front(const dc_tblock_front_params* p, dc_stream s);
int32_t dc_tblock_front_ok(const dc_tblock_front_params* p);     /* 1: the shape / dtype is served (pointers are not looked at) */

/* ---------------------------------------------------------------- eps-MSE -------- */
/* err[u] = (|| eps_hat_u - eps_{bj(u)} ||_2)^2 over C*H*W   (reference :706-711)
 * pred [n_units, H, W, ld] f32 NHWC; eps [n_bj,C,H,W], x [n_img,C,H,W] f32 NCHW.
 * v_param: eps_hat = sigma*z + alpha*pred with z = alpha*x + sigma*eps (fp32), else pred.
 * out_index: err is stored at out[out_index[u]] (NULL: out[u]). Deterministic reduction. */
typedef struct {
  const float* pred; const float* eps; const float* x; const float* alpha; const float* sigma;
  const int32_t* bj_of_unit; const int32_t* img_of_bj; const int32_t* out_index;
  float* out; int32_t n_units, C, H, W, ld, v_param;
  int32_t patch;            /* >1: pred is DiT's un-patchified projection [n_units, H/p, W/p, ld] with
                               k = (py*p+px)*C + c (nets/dit.py un-patchify folded into the read) */
} dc_eps_mse_params;
int dc_eps_mse(const dc_eps_mse_params* p, dc_stream s);

/* ---------------------------------------------------------------- sampler step --- */
/* One ancestral DDPM step with classifier-free guidance (reference :175-208 ddpm_sampler_step + the update :262-266):
 *   pred = (1 + w) * pred_c - w * pred_u;  x = v_param ? alpha_t z - sigma_t pred : (z - sigma_t pred) / alpha_t;  x = clip(x, -1, 1)
 *   mu = alpha_s * (z * (1 - c) / alpha_t + c * x);   out = noise ? mu + noise * sd : clip(mu, -1, 1)      (sd = sqrt(sigma_s^2 c))
 * z / noise / out [n, C, H, W] f32 NCHW; pred [2n, H, W, ld] f32 NHWC, rows 2b (class token) and 2b+1 (null token) of image b —
 * the output of ONE batch-2 backbone plan (patch > 1: DiT's un-patchified projection as in dc_eps_mse, whose feature stride is C:
 * the backbone's out_channels must equal C).  Same operation order as the reference's torch expressions, no contraction: equal to
 * them bit for bit for the same fp32 scalars (w, one_plus_w, alpha_*, sigma_t, c, sd — formed by the caller). */
typedef struct {
  const float* z; const float* pred; const float* noise; float* out;
  int32_t n, C, H, W, ld, patch, v_param;
  float w, alpha_t, sigma_t, alpha_s, c, sd;
  float one_plus_w;        /* (float)(1.0 + (double)w): the reference forms 1 + w as a Python double and torch rounds it once */
} dc_ddpm_step_params;
int dc_ddpm_step(const dc_ddpm_step_params* p, dc_stream s);

/* ---------------------------------------------------------------- Haar ----------- */
/* in [n, C, H, W] f32 -> out [n, 4C, H/2, W/2], channel 4i+{0,1,2,3} = cA,cH,cV,cD; out*=scale */
int dc_haar_dwt2(const float* in, float* out, int32_t n, int32_t C, int32_t H, int32_t W, float scale, dc_stream s);
/* in [n, 4C, h, w] -> out [n, C, 2h, 2w] */
int dc_haar_idwt2(const float* in, float* out, int32_t n, int32_t C, int32_t h, int32_t w, float scale, dc_stream s);

/* ---------------------------------------------------------------- stage end ------ */
/* errors [BS, C, T] f32 (+inf = cell not evaluated).  mean[b, c] = (sum_{j < t_end} errors[b, c, j]) / t_end, fp32, j ascending
 * (a fixed order: identical on every rank and for every world size).  keep[b, 0..k) = the k classes of smallest mean, ascending,
 * ties to the lower class id (reference: torch.topk(mean, k, largest=False), :720-721).  means [BS, C] optional (NULL). C <= 1024. */
int dc_stage_topk(const float* errors, int32_t BS, int32_t C, int32_t T, int32_t t_end, int32_t k, int32_t* keep, float* means, dc_stream s);
/* The last stage (k = 1): labels[b] = arg-min class as int64 (the LongTensor classify returns, :725). */
int dc_reduce_argmin(const float* errors, int32_t BS, int32_t C, int32_t T, int32_t t_end, int64_t* labels, float* means, dc_stream s);
/* Next stage's work-unit maps from keep[BS, k]: this rank's r-th pair is global pair g = rank + r*world of the stage, trial
 * j = t0 + g / BS, image b = g % BS; micro-batch m = pairs [m*n_bj, (m+1)*n_bj) (n_mb = ceil(n_pairs / n_bj)); slots past n_pairs
 * repeat the micro-batch's first pair and score into cell `dump`.  maps[m] = | ctx_of_unit[n_bj*k] | out_index[n_bj*k] | (int32):
 * class id of each unit and the flat index of errors[b, class, j] it writes (dc_eps_mse out_index). */
int dc_stage_maps(const int32_t* keep, int32_t BS, int32_t C, int32_t T, int32_t k, int32_t t0, int32_t n_pairs, int32_t rank,
                  int32_t world, int32_t n_bj, int32_t n_mb, int32_t dump, int32_t* maps, dc_stream s);

/* ---------------------------------------------------------------- plan ----------- */
typedef enum { DC_OP_QSAMPLE = 1, DC_OP_SINUSOID = 2, DC_OP_IGEMM = 3, DC_OP_GROUPNORM = 4,
               DC_OP_LAYERNORM = 5, DC_OP_ATTENTION = 6, DC_OP_EPS_MSE = 7, DC_OP_TBLOCK_FRONT = 8 } dc_op_kind;
typedef struct { int32_t kind; int32_t pad_; const void* params; } dc_op;
/* Launch ops[0..n) in order on the stream; stops at the first failure and returns its
 * status (failed index via dc_last_error text). */
int dc_run_plan(const dc_op* ops, int32_t n, dc_stream s);
/* Measurement variant (bench.py only): same launches bracketed by HIP events ON THE SAME STREAM;
 * synchronises the stream at the end and writes the elapsed milliseconds of op i to ms[i]. */
int dc_run_plan_timed(const dc_op* ops, int32_t n, dc_stream s, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* DCAMD_H */
