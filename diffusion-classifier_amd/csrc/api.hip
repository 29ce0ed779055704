// api.hip — C-ABI glue of libdcamd: version / error reporting and the native plan runner.
// The plan runner replaces the reference's Python double loop body
// (diffusion/diffusion_classifier.py:695-714, one eager op per line) by ONE native call that
// enqueues the whole scoring step on a stream; it performs no allocation or synchronisation,
// so the caller may capture it into a hipGraph.
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int dc_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dc_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return DC_ERR_LAUNCH;
  }
  return DC_OK;
}

extern "C" int dc_abi_version(void) { return DC_ABI_VERSION; }
extern "C" const char* dc_last_error(void) { return g_err; }
extern "C" const char* dc_arch(void) { return "gfx950"; }

static int run_one(const dc_op* ops, int i, dc_stream s);

extern "C" int dc_run_plan(const dc_op* ops, int32_t n, dc_stream s) {
  DC_REQUIRE(ops || n == 0, DC_ERR_ARG, "dc_run_plan: null ops");
  for (int i = 0; i < n; ++i) {
    const int rc = run_one(ops, i, s);
    if (rc != DC_OK) return rc;
  }
  return DC_OK;
}

extern "C" int dc_run_plan_timed(const dc_op* ops, int32_t n, dc_stream s, float* ms) {
  DC_REQUIRE((ops && ms) || n == 0, DC_ERR_ARG, "dc_run_plan_timed: null ops/ms");
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  hipEvent_t* ev = new hipEvent_t[n + 1];
  for (int i = 0; i <= n; ++i) (void)hipEventCreate(&ev[i]);
  int rc = DC_OK;
  (void)hipEventRecord(ev[0], st);
  int done = 0;
  for (; done < n; ++done) {
    rc = run_one(ops, done, s);
    if (rc != DC_OK) break;
    (void)hipEventRecord(ev[done + 1], st);
  }
  (void)hipStreamSynchronize(st);
  for (int i = 0; i < done; ++i) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
  for (int i = 0; i <= n; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

static int run_one(const dc_op* ops, int i, dc_stream s) {
  {
    int rc;
    switch (ops[i].kind) {
      case DC_OP_QSAMPLE: rc = dc_qsample(static_cast<const dc_qsample_params*>(ops[i].params), s); break;
      case DC_OP_SINUSOID: rc = dc_sinusoid(static_cast<const dc_sinusoid_params*>(ops[i].params), s); break;
      case DC_OP_IGEMM: rc = dc_igemm(static_cast<const dc_igemm_params*>(ops[i].params), s); break;
      case DC_OP_GROUPNORM: rc = dc_groupnorm(static_cast<const dc_groupnorm_params*>(ops[i].params), s); break;
      case DC_OP_LAYERNORM: rc = dc_layernorm(static_cast<const dc_layernorm_params*>(ops[i].params), s); break;
      case DC_OP_ATTENTION: rc = dc_attention(static_cast<const dc_attention_params*>(ops[i].params), s); break;
      case DC_OP_TBLOCK_FRONT: rc = dc_tblock_front(static_cast<const dc_tblock_front_params*>(ops[i].params), s); break;
      case DC_OP_EPS_MSE: rc = dc_eps_mse(static_cast<const dc_eps_mse_params*>(ops[i].params), s); break;
      default: dc_set_error("dc_run_plan: op %d has unknown kind %d", i, ops[i].kind); return DC_ERR_ARG;
    }
    if (rc != DC_OK) {
      char inner[400];
      snprintf(inner, sizeof(inner), "%s", g_err);
      dc_set_error("dc_run_plan: op %d (kind %d) failed: %s", i, ops[i].kind, inner);
      return rc;
    }
  }
  return DC_OK;
}
