/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see d2q9_oracle.h).  Serial CPU restatement of the
 * reference's D2Q9-BGK timestep; every function cites the reference lines it follows.
 * Optional -fopenmp parallelises over rows; the result does not depend on the thread count
 * (per-row velocity sums are combined serially in row order).
 */
#include "d2q9_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORACLE_PAIRWISE
#define ORACLE_PAIRWISE 0
#endif

#define NSPEEDS 9
#define SQRT_REAL(x) ((sizeof(REAL) == sizeof(float)) ? (REAL)sqrtf((float)(x)) : (REAL)sqrt((double)(x)))
/* SoA index, d2q9-bgk.c:73 / kernels.cl:7: plane sp, row ii (y), column jj (x) */
#define IDX(jj, ii, sp, nx, ny) ((size_t)(sp) * (size_t)(nx) * (size_t)(ny) + (size_t)(ii) * (size_t)(nx) + (size_t)(jj))

int oracle_real_size(void) { return (int)sizeof(REAL); }
int oracle_pairwise_momentum(void) { return ORACLE_PAIRWISE; }

/* d2q9-bgk.c:457-497 */
int oracle_load_params(const char *paramfile, oracle_params *p, char *err)
{
  FILE *fp = fopen(paramfile, "r");
  if (fp == NULL) {
    snprintf(err, 256, "could not open input parameter file: %s", paramfile);
    return -1;
  }
  float density, accel, omega;
  const char *bad = NULL;
  if (fscanf(fp, "%d\n", &p->nx) != 1) bad = "nx";
  else if (fscanf(fp, "%d\n", &p->ny) != 1) bad = "ny";
  else if (fscanf(fp, "%d\n", &p->max_iters) != 1) bad = "maxIters";
  else if (fscanf(fp, "%d\n", &p->reynolds_dim) != 1) bad = "reynolds_dim";
  else if (fscanf(fp, "%f\n", &density) != 1) bad = "density";
  else if (fscanf(fp, "%f\n", &accel) != 1) bad = "accel";
  else if (fscanf(fp, "%f\n", &omega) != 1) bad = "omega";
  fclose(fp);
  if (bad) {
    snprintf(err, 256, "could not read param file: %s", bad);
    return -1;
  }
  /* the reference reads the three reals as fp32 (d2q9-bgk.c:83-85,482-490); the golden
   * files come from the fp64 ancestor that read them as double.  Re-parse as double for
   * the fp64 build so 0.1/0.005/1.85 are the fp64 literals. */
  if (sizeof(REAL) == sizeof(double)) {
    fp = fopen(paramfile, "r");
    int d0, d1, d2, d3;
    double a, b, c;
    if (fp && fscanf(fp, "%d %d %d %d %lf %lf %lf", &d0, &d1, &d2, &d3, &a, &b, &c) == 7) {
      p->density = (REAL)a; p->accel = (REAL)b; p->omega = (REAL)c;
    } else {
      p->density = (REAL)density; p->accel = (REAL)accel; p->omega = (REAL)omega;
    }
    if (fp) fclose(fp);
  } else {
    p->density = (REAL)density; p->accel = (REAL)accel; p->omega = (REAL)omega;
  }
  p->free_cells_inv = (REAL)1 / (REAL)(p->nx * p->ny);
  return 0;
}

/* d2q9-bgk.c:553-591 */
int oracle_load_obstacles(const char *obstaclefile, oracle_params *p, int *obstacles, char *err)
{
  const int nx = p->nx, ny = p->ny;
  int free_cells = nx * ny;
  memset(obstacles, 0, sizeof(int) * (size_t)nx * (size_t)ny);
  FILE *fp = fopen(obstaclefile, "r");
  if (fp == NULL) {
    snprintf(err, 256, "could not open input obstacles file: %s", obstaclefile);
    return -1;
  }
  int xx, yy, blocked, retval;
  while ((retval = fscanf(fp, "%d %d %d\n", &xx, &yy, &blocked)) != EOF) {
    const char *bad = NULL;
    if (retval != 3) bad = "expected 3 values per line in obstacle file";
    else if (xx < 0 || xx > nx - 1) bad = "obstacle x-coord out of range";
    else if (yy < 0 || yy > ny - 1) bad = "obstacle y-coord out of range";
    else if (blocked != 1) bad = "obstacle blocked value should be 1";
    if (bad) {
      snprintf(err, 256, "%s", bad);
      fclose(fp);
      return -1;
    }
    /* duplicates (every shipped file lists the corners twice) count once, d2q9-bgk.c:583-585 */
    if (!obstacles[yy * nx + xx]) free_cells--;
    obstacles[yy * nx + xx] = blocked;
  }
  fclose(fp);
  p->free_cells_inv = (REAL)1 / (REAL)free_cells;
  return 0;
}

/* d2q9-bgk.c:529-550: uniform density at rest */
void oracle_init_cells(const oracle_params *p, REAL *cells)
{
  const int nx = p->nx, ny = p->ny;
  const REAL w0 = p->density * (REAL)4 / (REAL)9;
  const REAL w1 = p->density / (REAL)9;
  const REAL w2 = p->density / (REAL)36;
  const size_t n = (size_t)nx * (size_t)ny;
  for (size_t i = 0; i < n; i++) {
    cells[0 * n + i] = w0;
    for (int k = 1; k <= 4; k++) cells[k * n + i] = w1;
    for (int k = 5; k <= 8; k++) cells[k * n + i] = w2;
  }
}

/* kernels.cl:9-53 */
void oracle_accelerate_flow(const oracle_params *p, REAL *cells, const int *obstacles)
{
  oracle_accelerate_row(p, cells, obstacles, p->ny - 2); /* kernels.cl:18 */
}

void oracle_accelerate_row(const oracle_params *p, REAL *cells, const int *obstacles, int row)
{
  const int nx = p->nx, ny = p->ny;
  const REAL w1 = p->density * p->accel / (REAL)9;
  const REAL w2 = p->density * p->accel / (REAL)36;
  const int ii = row;
  for (int jj = 0; jj < nx; jj++) {
    REAL r3 = cells[IDX(jj, ii, 3, nx, ny)];
    REAL r6 = cells[IDX(jj, ii, 6, nx, ny)];
    REAL r7 = cells[IDX(jj, ii, 7, nx, ny)];
    /* kernels.cl:29-33: not an obstacle and no density would go negative */
    if (!obstacles[ii * nx + jj] && (r3 - w1) > (REAL)0 && (r6 - w2) > (REAL)0 && (r7 - w2) > (REAL)0) {
      cells[IDX(jj, ii, 1, nx, ny)] += w1;
      cells[IDX(jj, ii, 5, nx, ny)] += w2;
      cells[IDX(jj, ii, 8, nx, ny)] += w2;
      cells[IDX(jj, ii, 3, nx, ny)] = r3 - w1;
      cells[IDX(jj, ii, 6, nx, ny)] = r6 - w2;
      cells[IDX(jj, ii, 7, nx, ny)] = r7 - w2;
    }
  }
}

/* one row of kernels.cl:56-231; returns sum over fluid cells of |j|/rho for this row */
static double timestep_row(const oracle_params *p, const REAL *src, REAL *dst, const int *obstacles, int ii)
{
  const int nx = p->nx, ny = p->ny;
  const REAL omega = p->omega;
  const REAL ic_sq = (REAL)3;            /* kernels.cl:63 */
  const REAL w0 = (REAL)4 / (REAL)9;    /* kernels.cl:65-67 */
  const REAL w1 = (REAL)1 / (REAL)9;
  const REAL w2 = (REAL)1 / (REAL)36;
  static const int opp[NSPEEDS] = {0, 3, 4, 1, 2, 7, 8, 5, 6}; /* kernels.cl:69 lookup[k][0] */
  const int y_n = (ii + 1 == ny) ? 0 : ii + 1; /* kernels.cl:91-93 */
  const int y_s = (ii == 0) ? ny - 1 : ii - 1;
  double row_u = 0.0;

  for (int jj = 0; jj < nx; jj++) {
    const int x_e = (jj + 1 >= nx) ? jj + 1 - nx : jj + 1; /* kernels.cl:99-102 */
    const int x_w = (jj == 0) ? nx - 1 : jj - 1;
    REAL g[NSPEEDS];
    /* pull-stream gather, kernels.cl:104-112 */
    g[0] = src[IDX(jj, ii, 0, nx, ny)];
    g[1] = src[IDX(x_w, ii, 1, nx, ny)];
    g[2] = src[IDX(jj, y_s, 2, nx, ny)];
    g[3] = src[IDX(x_e, ii, 3, nx, ny)];
    g[4] = src[IDX(jj, y_n, 4, nx, ny)];
    g[5] = src[IDX(x_w, y_s, 5, nx, ny)];
    g[6] = src[IDX(x_e, y_s, 6, nx, ny)];
    g[7] = src[IDX(x_e, y_n, 7, nx, ny)];
    g[8] = src[IDX(x_w, y_n, 8, nx, ny)];

    if (obstacles[ii * nx + jj]) {
      /* rebound: un-relaxed value goes to the opposite plane, kernels.cl:187-197 with lmask=0 */
      for (int k = 0; k < NSPEEDS; k++) dst[IDX(jj, ii, opp[k], nx, ny)] = g[k];
      continue;
    }

    /* kernels.cl:119-129 */
    REAL dens = g[0];
    for (int k = 1; k < NSPEEDS; k++) dens += g[k];
    const REAL densinv = (REAL)1 / dens;
    /* momenta (not divided by density), kernels.cl:131-141 */
#if ORACLE_PAIRWISE
    /* pairwise differences make a cell at rest give exactly zero momentum in fp32 */
    const REAL diag_a = g[5] - g[7], diag_b = g[8] - g[6];
    const REAL u_x = (g[1] - g[3]) + (diag_a + diag_b);
    const REAL u_y = (g[2] - g[4]) + (diag_a - diag_b);
#else
    REAL u_x = g[1] + g[5]; u_x += g[8]; u_x -= g[3]; u_x -= g[6]; u_x -= g[7];
    REAL u_y = g[2] + g[5]; u_y += g[6]; u_y -= g[4]; u_y -= g[7]; u_y -= g[8];
#endif
    const REAL u_sq = u_x * u_x + u_y * u_y;
    /* kernels.cl:146-154 */
    REAL uvec[NSPEEDS];
    uvec[0] = 0;
    uvec[1] = u_x;        uvec[2] = u_y;
    uvec[3] = -u_x;       uvec[4] = -u_y;
    uvec[5] = u_x + u_y;  uvec[6] = -u_x + u_y;
    uvec[7] = -u_x - u_y; uvec[8] = u_x - u_y;
    /* equilibria in the division-free form of kernels.cl:156-185 */
    const REAL half_densinv_icsq = (REAL)0.5 * densinv * ic_sq;
    REAL d_equ[NSPEEDS];
    d_equ[0] = w0 * (dens - half_densinv_icsq * u_sq);
    for (int k = 1; k < NSPEEDS; k++) {
      const REAL t = uvec[k] * ic_sq;
      const REAL tsq = t * uvec[k];
      d_equ[k] = ((k < 5) ? w1 : w2) * (dens + t + half_densinv_icsq * (tsq - u_sq));
    }
    /* relaxation, kernels.cl:189-197 with lmask=1 */
    for (int k = 0; k < NSPEEDS; k++) dst[IDX(jj, ii, k, nx, ny)] = g[k] + omega * (d_equ[k] - g[k]);
    /* kernels.cl:198 */
    row_u += (double)(SQRT_REAL(u_sq) * densinv);
  }
  return row_u;
}

/* kernels.cl:56-231 + the reduction of kernels.cl:202-229,234-290 */
REAL oracle_timestep(const oracle_params *p, const REAL *src, REAL *dst, const int *obstacles)
{
  const int ny = p->ny;
  double *row_sums = (double *)malloc(sizeof(double) * (size_t)ny);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int ii = 0; ii < ny; ii++) row_sums[ii] = timestep_row(p, src, dst, obstacles, ii);
  double tot_u = 0.0;
  for (int ii = 0; ii < ny; ii++) tot_u += row_sums[ii];
  free(row_sums);
  return (REAL)(tot_u * (double)p->free_cells_inv);
}

double oracle_timestep_rows(const oracle_params *p, const REAL *src, REAL *dst, const int *obstacles, int y0, int y1)
{
  if (y1 <= y0) return 0.0;
  /* per-row sums added up in row order, as in oracle_timestep: the same value with and without OpenMP */
  double *row_sums = (double *)malloc(sizeof(double) * (size_t)(y1 - y0));
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int ii = y0; ii < y1; ii++) row_sums[ii - y0] = timestep_row(p, src, dst, obstacles, ii);
  double tot_u = 0.0;
  for (int ii = y0; ii < y1; ii++) tot_u += row_sums[ii - y0];
  free(row_sums);
  return tot_u;
}

/* d2q9-bgk.c:221-238 */
void oracle_run(const oracle_params *p, REAL *cells, REAL *tmp_cells, const int *obstacles,
                REAL *av_vels, int nsteps)
{
  REAL *bufs[2] = {cells, tmp_cells};
  int rd = 0;
  for (int tt = 0; tt < nsteps; tt++) {
    oracle_accelerate_flow(p, bufs[rd], obstacles);
    REAL av = oracle_timestep(p, bufs[rd], bufs[rd ^ 1], obstacles);
    if (av_vels) av_vels[tt] = av;
    rd ^= 1;
  }
  /* the reference reads back ocl.cells unconditionally (d2q9-bgk.c:251-253), correct only for
   * even maxIters; here the final state is always returned in `cells` */
  if (rd == 1) memcpy(cells, tmp_cells, sizeof(REAL) * NSPEEDS * (size_t)p->nx * (size_t)p->ny);
}

/* per-cell density and velocity as in d2q9-bgk.c:404-437 / 803-831 */
static inline void cell_moments(const oracle_params *p, const REAL *cells, int jj, int ii,
                                REAL *dens, REAL *u_x, REAL *u_y)
{
  const int nx = p->nx, ny = p->ny;
  REAL f[NSPEEDS];
  REAL local_density = 0;
  for (int kk = 0; kk < NSPEEDS; kk++) {
    f[kk] = cells[IDX(jj, ii, kk, nx, ny)];
    local_density += f[kk];
  }
  *dens = local_density;
  *u_x = (f[1] + f[5] + f[8] - f[3] - f[6] - f[7]) / local_density;
  *u_y = (f[2] + f[5] + f[6] - f[4] - f[7] - f[8]) / local_density;
}

/* d2q9-bgk.c:396-442 */
REAL oracle_av_velocity(const oracle_params *p, const REAL *cells, const int *obstacles)
{
  REAL tot_u = 0;
  for (int ii = 0; ii < p->ny; ii++)
    for (int jj = 0; jj < p->nx; jj++)
      if (!obstacles[ii * p->nx + jj]) {
        REAL d, u_x, u_y;
        cell_moments(p, cells, jj, ii, &d, &u_x, &u_y);
        tot_u += (REAL)sqrt((double)((u_x * u_x) + (u_y * u_y)));
      }
  return tot_u * p->free_cells_inv;
}

/* d2q9-bgk.c:747-752 */
REAL oracle_calc_reynolds(const oracle_params *p, const REAL *cells, const int *obstacles)
{
  const REAL viscosity = (REAL)1 / (REAL)6 * ((REAL)2 / p->omega - (REAL)1);
  return oracle_av_velocity(p, cells, obstacles) * (REAL)p->reynolds_dim / viscosity;
}

/* d2q9-bgk.c:754-770 */
REAL oracle_total_density(const oracle_params *p, const REAL *cells)
{
  REAL total = 0;
  const size_t n = (size_t)NSPEEDS * (size_t)p->nx * (size_t)p->ny;
  for (size_t i = 0; i < n; i++) total += cells[i];
  return total;
}

/* the arithmetic of d2q9-bgk.c:787-832 */
void oracle_final_fields(const oracle_params *p, const REAL *cells, const int *obstacles,
                         REAL *u_x, REAL *u_y, REAL *u, REAL *pressure)
{
  const REAL c_sq = (REAL)1 / (REAL)3;
  for (int ii = 0; ii < p->ny; ii++)
    for (int jj = 0; jj < p->nx; jj++) {
      const size_t c = (size_t)ii * (size_t)p->nx + (size_t)jj;
      if (obstacles[c]) {
        u_x[c] = u_y[c] = u[c] = 0;
        pressure[c] = p->density * c_sq;
      } else {
        REAL d;
        cell_moments(p, cells, jj, ii, &d, &u_x[c], &u_y[c]);
        u[c] = (REAL)sqrt((double)((u_x[c] * u_x[c]) + (u_y[c] * u_y[c])));
        pressure[c] = d * c_sq;
      }
    }
}

/* d2q9-bgk.c:772-856 */
int oracle_write_values(const oracle_params *p, const REAL *cells, const int *obstacles,
                        const REAL *av_vels, const char *final_state_path, const char *av_vels_path)
{
  const size_t n = (size_t)p->nx * (size_t)p->ny;
  REAL *fields = (REAL *)malloc(sizeof(REAL) * 4 * n);
  if (!fields) return -1;
  oracle_final_fields(p, cells, obstacles, fields, fields + n, fields + 2 * n, fields + 3 * n);
  FILE *fp = fopen(final_state_path, "w");
  if (fp == NULL) { free(fields); return -1; }
  for (int ii = 0; ii < p->ny; ii++)
    for (int jj = 0; jj < p->nx; jj++) {
      const size_t c = (size_t)ii * (size_t)p->nx + (size_t)jj;
      fprintf(fp, "%d %d %.12E %.12E %.12E %.12E %d\n", jj, ii, (double)fields[c], (double)fields[n + c],
              (double)fields[2 * n + c], (double)fields[3 * n + c], obstacles[c]);
    }
  fclose(fp);
  free(fields);
  fp = fopen(av_vels_path, "w");
  if (fp == NULL) return -1;
  for (int ii = 0; ii < p->max_iters; ii++) fprintf(fp, "%d:\t%.12E\n", ii, (double)av_vels[ii]);
  fclose(fp);
  return 0;
}
