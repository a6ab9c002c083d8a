/*
 * TEST INFRASTRUCTURE — a stand-in for liblbm_hip.so with the ABI of include/lbm.h and NO numerics, so that the thin C
 * host (host/d2q9-bgk.c: mmap scanner, parallel obstacle parse, threaded final_state formatter) can run on a CPU-only
 * box under AddressSanitizer / UBSan (`make asan`, tests/test_host_cli.py).  Never linked into the product.
 *   default            lbm_create fails like the real library without a GPU (LBM_ERR_HIP)
 *   LBM_STUB_RUN=1     lbm_create succeeds; the "state" is a deterministic pattern of the cell coordinates, av_vels[t] =
 *                      (t+1)*1e-6: enough for the host to walk its whole output path
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lbm.h"

struct lbm_ctx {
  lbm_params p;
  int steps;
  int32_t *obstacles;
};

static const char *g_err = "";

const char *lbm_last_error(void) { return g_err; }
const char *lbm_version(void) { return "lbm-stub (no device code)"; }

int lbm_create(lbm_ctx **out, const lbm_params *params, const int32_t *obstacles, int ndev, const int *dev_ids)
{
  (void)ndev; (void)dev_ids;
  if (!out || !params || !obstacles) { g_err = "NULL argument"; return LBM_ERR_ARG; }
  *out = NULL;
  if (!getenv("LBM_STUB_RUN")) { g_err = "HIP error during 'hipGetDeviceCount': no ROCm-capable device is detected (stub)"; return LBM_ERR_HIP; }
  lbm_ctx *c = (lbm_ctx *)calloc(1, sizeof *c);
  if (!c) return LBM_ERR_ARG;
  c->p = *params;
  const size_t n = (size_t)params->nx * params->ny;
  c->obstacles = (int32_t *)malloc(n * sizeof(int32_t));
  if (!c->obstacles) { free(c); return LBM_ERR_ARG; }
  memcpy(c->obstacles, obstacles, n * sizeof(int32_t));
  if (getenv("LBM_STUB_PRINT")) fprintf(stderr, "stub: free_cells_inv %.9e\n", (double)params->free_cells_inv);
  *out = c;
  return LBM_OK;
}

int lbm_create_rank(lbm_ctx **out, const lbm_params *p, const int32_t *o, int rank, int nranks, int device, const void *id)
{
  (void)rank; (void)nranks; (void)device; (void)id;
  return lbm_create(out, p, o, 1, NULL);
}
size_t lbm_comm_id_size(void) { return 128; }
int lbm_comm_get_id(void *o) { memset(o, 0, 128); return LBM_OK; }
size_t lbm_peer_info_size(void) { return 8; }
int lbm_peer_info(lbm_ctx *c, void *o) { (void)c; memset(o, 0, 8); return LBM_OK; }
int lbm_connect_peers(lbm_ctx *c, const void *a, const void *b) { (void)c; (void)a; (void)b; return LBM_OK; }
int lbm_set_default(const char *k, long v) { (void)k; (void)v; return LBM_OK; }
int lbm_upload(lbm_ctx *c, const float *cells) { (void)cells; c->steps = 0; return LBM_OK; }
int lbm_disconnect_peers(lbm_ctx *c) { (void)c; return LBM_OK; }
int lbm_upload_obstacles(lbm_ctx *c, const int32_t *obstacles) { (void)c; return obstacles ? LBM_OK : LBM_ERR_ARG; }
int lbm_run(lbm_ctx *c, int n)
{
  if (n < 0 || c->steps + n > c->p.max_iters) { g_err = "max_iters exceeded"; return LBM_ERR_STATE; }
  c->steps += n;
  return LBM_OK;
}
int lbm_run_timed(lbm_ctx *c, int n, double *ms) { if (ms) *ms = 1.0; return lbm_run(c, n); }
int lbm_run_profiled(lbm_ctx *c, int n, double *st) { memset(st, 0, 8 * sizeof(double)); return lbm_run(c, n); }
int lbm_sync(lbm_ctx *c) { (void)c; return LBM_OK; }
int lbm_host_alloc(void **p, size_t bytes) { if (!p || !bytes) return LBM_ERR_ARG; *p = malloc(bytes); return *p ? LBM_OK : LBM_ERR_HIP; }
int lbm_host_free(void *p) { free(p); return LBM_OK; }
int lbm_steps_done(const lbm_ctx *c) { return c ? c->steps : -1; }
int lbm_row_range(const lbm_ctx *c, int *y0, int *y1) { if (y0) *y0 = 0; if (y1) *y1 = c->p.ny; return LBM_OK; }
int lbm_download(lbm_ctx *c, float *cells, float *av)
{
  const size_t n = (size_t)c->p.nx * c->p.ny;
  if (cells) for (size_t i = 0; i < 9 * n; i++) cells[i] = 0.01f;
  if (av) for (int t = 0; t < c->steps; t++) av[t] = (float)(t + 1) * 1e-6f;
  return LBM_OK;
}
int lbm_final_state(lbm_ctx *c, float *ux, float *uy, float *u, float *pr)
{
  const int nx = c->p.nx, ny = c->p.ny;
  for (int y = 0; y < ny; y++)
    for (int x = 0; x < nx; x++) {
      const size_t i = (size_t)y * nx + x;
      const int ob = c->obstacles[i] != 0;
      if (ux) ux[i] = ob ? 0.0f : (float)x * 1e-3f - 0.0625f;   /* negative values too: the longest output lines */
      if (uy) uy[i] = ob ? 0.0f : -(float)y * 1e-4f;
      if (u) u[i] = ob ? 0.0f : (float)(x + y) * 1e-5f;
      if (pr) pr[i] = c->p.density / 3.0f;
    }
  return LBM_OK;
}
int lbm_reynolds(lbm_ctx *c, float *re) { (void)c; *re = 1.0f; return LBM_OK; }
int lbm_set_option(lbm_ctx *c, const char *k, long v) { (void)c; (void)k; (void)v; return LBM_OK; }
int lbm_get_option(const lbm_ctx *c, const char *k, long *v) { (void)c; (void)k; *v = 0; return LBM_OK; }
int lbm_copy_bandwidth(size_t b, int it, double *g) { (void)b; (void)it; *g = 0.0; return LBM_OK; }
int lbm_valu_rate(int n, double *g) { (void)n; *g = 0.0; return LBM_OK; }
void lbm_destroy(lbm_ctx *c)
{
  if (!c) return;
  free(c->obstacles);
  free(c);
}
