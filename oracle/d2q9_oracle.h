/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link,
 * import or execute it, and only as the checker / the reported CPU baseline.
 *
 * Serial CPU restatement of the reference's D2Q9-BGK timestep
 * (ag14774/OpenCL-Lattice-Boltzmann: kernels.cl + d2q9-bgk.c).  The reference ships no
 * CPU implementation of the timestep (its d2q9-bgk.c is an OpenCL host program), so this
 * file restates the device kernels' algorithm in plain C.
 *
 * Parity pin: the fp64 build reproduces every shipped golden file of the reference
 * (check/{128x128,128x256}.{av_vels,final_state}.dat, check/{256x256,1024x1024}.av_vels.dat),
 * the Reynolds numbers printed in README.md:78,88,98 and the 256x256 pressure values leaked
 * by the 256x256 check.txt transcripts under profiles/ (stages 1,3,4,7) — see tests/test_oracle_golden.py.
 *
 * One source, two builds: -DREAL=double (golden-file parity) and -DREAL=float
 * (like-for-like partner of the fp32 GPU kernel).
 */
#ifndef D2Q9_ORACLE_H
#define D2Q9_ORACLE_H

#ifndef REAL
#define REAL double
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* run constants; mirrors t_param, d2q9-bgk.c:81-92 */
typedef struct {
  int nx, ny, max_iters, reynolds_dim;
  REAL density, accel, omega, free_cells_inv;
} oracle_params;

/* sizeof(REAL) of this build (4 or 8) so a ctypes caller can check which library it loaded */
int oracle_real_size(void);
/* 1 when the momentum sums use pairwise differences, 0 = kernels.cl:131-141 order */
int oracle_pairwise_momentum(void);

/* d2q9-bgk.c:466-492; returns 0 or -1 with a message in err (>=256 bytes) */
int oracle_load_params(const char *paramfile, oracle_params *p, char *err);
/* d2q9-bgk.c:553-591; obstacles[ny*nx] is zeroed then filled; sets p->free_cells_inv */
int oracle_load_obstacles(const char *obstaclefile, oracle_params *p, int *obstacles, char *err);
/* d2q9-bgk.c:529-550 */
void oracle_init_cells(const oracle_params *p, REAL *cells);

/* kernels.cl:9-53 — in place on row ny-2 */
void oracle_accelerate_flow(const oracle_params *p, REAL *cells, const int *obstacles);
/* kernels.cl:56-231 — pull-stream + rebound + collide src->dst; returns av_vels[t] */
REAL oracle_timestep(const oracle_params *p, const REAL *src, REAL *dst, const int *obstacles);
/* The same two kernels restricted to chosen rows (used by the row-partition tests): accelerate a given
 * row instead of ny-2; compute only rows [y0, y1) of dst and return the RAW sum of |j|/rho over them */
void oracle_accelerate_row(const oracle_params *p, REAL *cells, const int *obstacles, int row);
double oracle_timestep_rows(const oracle_params *p, const REAL *src, REAL *dst, const int *obstacles, int y0, int y1);
/* d2q9-bgk.c:221-238 loop: nsteps of accelerate+timestep with ping-pong; the final state is
 * always left in `cells` (copied back when nsteps is odd).  av_vels may be NULL. */
void oracle_run(const oracle_params *p, REAL *cells, REAL *tmp_cells, const int *obstacles,
                REAL *av_vels, int nsteps);

/* d2q9-bgk.c:396-442, 747-752, 754-770 */
REAL oracle_av_velocity(const oracle_params *p, const REAL *cells, const int *obstacles);
REAL oracle_calc_reynolds(const oracle_params *p, const REAL *cells, const int *obstacles);
REAL oracle_total_density(const oracle_params *p, const REAL *cells);
/* d2q9-bgk.c:772-856 with real u_x/u_y in columns 3-4 (the golden files hold real values) */
int oracle_write_values(const oracle_params *p, const REAL *cells, const int *obstacles,
                        const REAL *av_vels, const char *final_state_path, const char *av_vels_path);
/* per-cell output fields of write_values without the text formatting (row-major [ny][nx]) */
void oracle_final_fields(const oracle_params *p, const REAL *cells, const int *obstacles,
                         REAL *u_x, REAL *u_y, REAL *u, REAL *pressure);

#ifdef __cplusplus
}
#endif
#endif
