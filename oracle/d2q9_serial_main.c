/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see d2q9_oracle.h).
 * Serial CPU driver with the reference's command line and outputs (d2q9-bgk.c:165-280):
 *   d2q9-bgk-serial <paramfile> <obstaclefile>  ->  final_state.dat, av_vels.dat in the CWD.
 * Extras (environment, not part of the reference contract):
 *   ORACLE_MAX_ITERS=n   run only n steps (rate-only CPU baseline on large grids)
 *   ORACLE_NO_OUTPUT=1   skip writing the two .dat files
 */
#include "d2q9_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <sys/resource.h>
#include <sys/time.h>

static void die(const char *message, const int line, const char *file)
{
  /* d2q9-bgk.c:868-874 */
  fprintf(stderr, "Error at line %d of file %s:\n", line, file);
  fprintf(stderr, "%s\n", message);
  fflush(stderr);
  exit(EXIT_FAILURE);
}

static double wall(void)
{
  struct timeval t;
  gettimeofday(&t, NULL);
  return t.tv_sec + t.tv_usec / 1000000.0;
}

int main(int argc, char *argv[])
{
  if (argc != 3) { /* d2q9-bgk.c:183-186,876-880 */
    fprintf(stderr, "Usage: %s <paramfile> <obstaclefile>\n", argv[0]);
    exit(EXIT_FAILURE);
  }
  oracle_params params;
  char err[256];
  if (oracle_load_params(argv[1], &params, err)) die(err, __LINE__, __FILE__);
  if (getenv("ORACLE_MAX_ITERS")) params.max_iters = atoi(getenv("ORACLE_MAX_ITERS"));
  const size_t n = (size_t)params.nx * (size_t)params.ny;
  REAL *cells = (REAL *)malloc(sizeof(REAL) * 9 * n);
  REAL *tmp_cells = (REAL *)malloc(sizeof(REAL) * 9 * n);
  int *obstacles = (int *)malloc(sizeof(int) * n);
  REAL *av_vels = (REAL *)malloc(sizeof(REAL) * (size_t)(params.max_iters > 0 ? params.max_iters : 1));
  if (!cells || !tmp_cells) die("cannot allocate memory for cells", __LINE__, __FILE__);
  if (!obstacles) die("cannot allocate column memory for obstacles", __LINE__, __FILE__);
  if (oracle_load_obstacles(argv[2], &params, obstacles, err)) die(err, __LINE__, __FILE__);
  oracle_init_cells(&params, cells);

  const double tic = wall();
  oracle_run(&params, cells, tmp_cells, obstacles, av_vels, params.max_iters);
  const double toc = wall();
  struct rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  const double usrtim = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1000000.0;
  const double systim = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1000000.0;

  /* d2q9-bgk.c:271-275 */
  printf("==done==\n");
  printf("Reynolds number:\t\t%.12E\n", (double)oracle_calc_reynolds(&params, cells, obstacles));
  printf("Elapsed time:\t\t\t%.6lf (s)\n", toc - tic);
  printf("Elapsed user CPU time:\t\t%.6lf (s)\n", usrtim);
  printf("Elapsed system CPU time:\t%.6lf (s)\n", systim);
  printf("MLUPS:\t\t\t\t%.3f (serial oracle, %d-byte reals)\n",
         (double)n * params.max_iters / (toc - tic) / 1e6, oracle_real_size());
  if (!getenv("ORACLE_NO_OUTPUT"))
    if (oracle_write_values(&params, cells, obstacles, av_vels, "final_state.dat", "av_vels.dat"))
      die("could not open file output file", __LINE__, __FILE__);
  free(cells); free(tmp_cells); free(obstacles); free(av_vels);
  return EXIT_SUCCESS;
}
