/*
 * d2q9-bgk — thin C host for the MI355X-native D2Q9-BGK lattice-Boltzmann timestep.
 *
 * Keeps the process contract of the reference's d2q9-bgk.c (ag14774/OpenCL-Lattice-Boltzmann):
 *   usage          d2q9-bgk <paramfile> <obstaclefile>                 (d2q9-bgk.c:183-191,876-880)
 *   inputs         7-token parameter file, "x y 1" obstacle lines      (d2q9-bgk.c:466-492,571-586)
 *   errors         "Error at line N of file F:\n<message>\n", exit 1   (d2q9-bgk.c:868-874)
 *   outputs        final_state.dat and av_vels.dat in the CWD          (d2q9-bgk.c:69-70,772-856)
 *   stdout         ==done== / Reynolds number / three elapsed times    (d2q9-bgk.c:271-275)
 *   timed region   initial-state transfer + step loop + read-back      (d2q9-bgk.c:196-263)
 * All device work goes through the C ABI of liblbm_hip.so (include/lbm.h); this file contains no
 * GPU code and no numerics of the timestep.  The two input files are mmap'ed and scanned in place.
 *
 * Environment (extensions; none is needed for a reference-style run):
 *   LBM_NGPUS=n | LBM_DEVICES=a,b,..  row-partition the grid over several GPUs (one process)
 *   LBM_MAX_ITERS=n                   override maxIters (benchmark runs on large grids)
 *   LBM_NO_OUTPUT=1                   skip writing the two .dat files
 *   LBM_HOST_INIT=1                   build the initial state on the host and upload it, as the
 *                                     reference does (default: initialise on the device)
 *   LBM_PARSER_THREADS=n, LBM_WRITER_THREADS=n   threads of the obstacle-file scan / of the final_state formatter
 *                                     (default: up to 16, at least 64 KB of obstacle text per thread)
 */
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "lbm.h"

#define NSPEEDS 9
#define FINALSTATEFILE "final_state.dat"
#define AVVELSFILE "av_vels.dat"

static void die(const char *message, const int line, const char *file)
{
  fprintf(stderr, "Error at line %d of file %s:\n", line, file);
  fprintf(stderr, "%s\n", message);
  fflush(stderr);
  exit(EXIT_FAILURE);
}

static void usage(const char *exe)
{
  fprintf(stderr, "Usage: %s <paramfile> <obstaclefile>\n", exe);
  exit(EXIT_FAILURE);
}

/* like the reference's checkError (d2q9-bgk.c:858-866), for the C ABI */
static void check_lbm(int rc, const char *op, const int line)
{
  if (rc != LBM_OK) {
    fprintf(stderr, "LBM error during '%s' on line %d: %d (%s)\n", op, line, rc, lbm_last_error());
    fflush(stderr);
    exit(EXIT_FAILURE);
  }
}

/* ---- mmap'ed text scanning ------------------------------------------------------------------ */

typedef struct { const char *base, *cur, *end; size_t len; } text_t;

static int map_file(const char *path, text_t *t)
{
  int fd = open(path, O_RDONLY);
  if (fd < 0) return -1;
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return -1; }
  t->len = (size_t)st.st_size;
  t->base = NULL;
  if (t->len > 0) {
    void *p = mmap(NULL, t->len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) { close(fd); return -1; }
    t->base = (const char *)p;
  }
  close(fd);
  t->cur = t->base;
  t->end = t->base + t->len;
  return 0;
}

static void unmap_file(text_t *t)
{
  if (t->base) munmap((void *)t->base, t->len);
}

static void skip_space(text_t *t)
{
  while (t->cur < t->end && (*t->cur == ' ' || (*t->cur >= '\t' && *t->cur <= '\r'))) t->cur++;
}

/* scanf("%d") on the mapped text: 1 converted, 0 matching failure, EOF at end of input */
static int scan_int(text_t *t, int *out)
{
  skip_space(t);
  if (t->cur >= t->end) return EOF;
  const char *p = t->cur;
  int neg = 0;
  if (*p == '+' || *p == '-') { neg = (*p == '-'); p++; }
  if (p >= t->end || *p < '0' || *p > '9') return 0;
  long v = 0;
  while (p < t->end && *p >= '0' && *p <= '9') {
    if (v < (1L << 40)) v = v * 10 + (*p - '0');
    p++;
  }
  *out = (int)(neg ? -v : v);
  t->cur = p;
  return 1;
}

/* scanf("%f") on the mapped text */
static int scan_float(text_t *t, float *out)
{
  skip_space(t);
  if (t->cur >= t->end) return EOF;
  char buf[64];
  size_t n = 0;
  const char *p = t->cur;
  while (p < t->end && n < sizeof(buf) - 1 && !(*p == ' ' || (*p >= '\t' && *p <= '\r'))) buf[n++] = *p++;
  buf[n] = '\0';
  char *endp = NULL;
  errno = 0;
  float v = strtof(buf, &endp);
  if (endp == buf) return 0;
  *out = v;
  t->cur += (endp - buf);
  return 1;
}

/* ---- input stage: d2q9-bgk.c:444-597 --------------------------------------------------------- */

static void load_params(const char *paramfile, lbm_params *params)
{
  char message[1024];
  text_t t;
  if (map_file(paramfile, &t) != 0) {
    snprintf(message, sizeof message, "could not open input parameter file: %s", paramfile);
    die(message, __LINE__, __FILE__);
  }
  if (scan_int(&t, &params->nx) != 1) die("could not read param file: nx", __LINE__, __FILE__);
  if (scan_int(&t, &params->ny) != 1) die("could not read param file: ny", __LINE__, __FILE__);
  if (scan_int(&t, &params->max_iters) != 1) die("could not read param file: maxIters", __LINE__, __FILE__);
  if (scan_int(&t, &params->reynolds_dim) != 1) die("could not read param file: reynolds_dim", __LINE__, __FILE__);
  if (scan_float(&t, &params->density) != 1) die("could not read param file: density", __LINE__, __FILE__);
  if (scan_float(&t, &params->accel) != 1) die("could not read param file: accel", __LINE__, __FILE__);
  if (scan_float(&t, &params->omega) != 1) die("could not read param file: omega", __LINE__, __FILE__);
  unmap_file(&t);
}

/* The obstacle records, serially: the statement of the reference's loop (d2q9-bgk.c:553-591) on the mapped text.
 * Used for small files and whenever the parallel scan below meets anything that is not a plain integer token, so that
 * malformed files fail exactly where and how the reference's fscanf("%d %d %d\n") loop fails. */
static long scan_obstacles_serial(text_t *t, const lbm_params *params, int32_t *obstacles)
{
  long blocked_cells = 0;
  for (;;) {
    int xx = 0, yy = 0, blocked = 0;
    /* fscanf("%d %d %d\n"): EOF only when the input ends before the first conversion */
    int r0 = scan_int(t, &xx);
    if (r0 == EOF) break;
    int retval = r0;
    if (retval == 1) {
      int r1 = scan_int(t, &yy);
      if (r1 == 1) {
        retval = 2;
        if (scan_int(t, &blocked) == 1) retval = 3;
      }
    }
    /* some checks, d2q9-bgk.c:573-580 */
    if (retval != 3) die("expected 3 values per line in obstacle file", __LINE__, __FILE__);
    if (xx < 0 || xx > params->nx - 1) die("obstacle x-coord out of range", __LINE__, __FILE__);
    if (yy < 0 || yy > params->ny - 1) die("obstacle y-coord out of range", __LINE__, __FILE__);
    if (blocked != 1) die("obstacle blocked value should be 1", __LINE__, __FILE__);
    /* a cell listed twice is one blocked cell, d2q9-bgk.c:583-585 */
    if (!obstacles[(size_t)yy * params->nx + xx]) blocked_cells++;
    obstacles[(size_t)yy * params->nx + xx] = blocked;
  }
  return blocked_cells;
}

/* Parallel scan of a large obstacle file (the tiled-up 8192x8192 geometry is 4 MB of text, ~330 000 records): the
 * mapped text is cut at whitespace into one byte range per thread; pass 1 counts the integer tokens of every range
 * (records are token triples, wherever the line breaks are — exactly fscanf's view), pass 2 lets every thread convert
 * the records whose first token lies in its range.  A cell is claimed with an atomic exchange, so duplicates are
 * counted once without a second pass.  The first offending record in FILE order decides the error, as in the
 * serial loop. */
enum { OB_OK = 0, OB_FIELDS = 1, OB_XRANGE = 2, OB_YRANGE = 3, OB_VALUE = 4 };

typedef struct {
  const char *beg, *end, *file_end;
  const lbm_params *params;
  int32_t *obstacles;
  long ntokens;        /* pass 1: integer tokens in [beg, end) */
  int malformed;       /* pass 1: a token that is not a plain integer */
  long first_token;    /* pass 2: global index of the range's first token */
  long blocked_cells;  /* pass 2: cells this thread claimed first */
  long err_record;     /* pass 2: index of the first bad record seen by this thread, or -1 */
  int err_kind;
} ob_job;

static int is_space_ch(char ch) { return ch == ' ' || (ch >= '\t' && ch <= '\r'); }

static void *ob_count_tokens(void *arg)
{
  ob_job *j = (ob_job *)arg;
  const char *p = j->beg;
  long n = 0;
  while (p < j->end) {
    while (p < j->end && is_space_ch(*p)) p++;
    if (p >= j->end) break;
    const char *q = p;
    if (*q == '+' || *q == '-') q++;
    const char *digits = q;
    while (q < j->file_end && *q >= '0' && *q <= '9') q++;
    if (q == digits || (q < j->file_end && !is_space_ch(*q))) { j->malformed = 1; break; }
    n++;
    p = q;
  }
  j->ntokens = n;
  return NULL;
}

/* next integer token at or after *pp (tokens are known to be well formed); 0 at the end of the file */
static int ob_next_int(const char **pp, const char *file_end, int *out)
{
  const char *p = *pp;
  while (p < file_end && is_space_ch(*p)) p++;
  if (p >= file_end) return 0;
  int neg = 0;
  if (*p == '+' || *p == '-') { neg = (*p == '-'); p++; }
  long v = 0;
  while (p < file_end && *p >= '0' && *p <= '9') {
    if (v < (1L << 40)) v = v * 10 + (*p - '0');
    p++;
  }
  *out = (int)(neg ? -v : v);
  *pp = p;
  return 1;
}

static void *ob_convert_records(void *arg)
{
  ob_job *j = (ob_job *)arg;
  const char *p = j->beg;
  long tok = j->first_token;
  int skip = (int)((3 - tok % 3) % 3), dummy;   /* tokens that finish a record begun in an earlier range */
  long left = j->ntokens;
  for (; skip > 0 && left > 0; skip--, left--, tok++) ob_next_int(&p, j->file_end, &dummy);
  j->err_record = -1;
  while (left > 0) {
    /* a record whose first token is in this range; its other two may lie beyond `end` */
    int v[3], got = 0;
    while (got < 3 && ob_next_int(&p, j->file_end, &v[got])) got++;
    const long record = tok / 3;
    int kind = OB_OK;
    if (got != 3) kind = OB_FIELDS;
    else if (v[0] < 0 || v[0] > j->params->nx - 1) kind = OB_XRANGE;
    else if (v[1] < 0 || v[1] > j->params->ny - 1) kind = OB_YRANGE;
    else if (v[2] != 1) kind = OB_VALUE;
    if (kind != OB_OK) { j->err_record = record; j->err_kind = kind; return NULL; }
    if (__atomic_exchange_n(&j->obstacles[(size_t)v[1] * j->params->nx + v[0]], 1, __ATOMIC_RELAXED) == 0) j->blocked_cells++;
    tok += 3;
    left -= 3;
  }
  return NULL;
}

static void run_jobs(ob_job *jobs, int n, void *(*fn)(void *))
{
  pthread_t tids[64];
  for (int t = 0; t < n; t++)
    if (pthread_create(&tids[t], NULL, fn, &jobs[t]) != 0) { fn(&jobs[t]); tids[t] = 0; }
  for (int t = 0; t < n; t++)
    if (tids[t]) pthread_join(tids[t], NULL);
}

static void load_obstacles(const char *obstaclefile, lbm_params *params, int32_t *obstacles)
{
  char message[1024];
  text_t t;
  if (map_file(obstaclefile, &t) != 0) {
    snprintf(message, sizeof message, "could not open input obstacles file: %s", obstaclefile);
    die(message, __LINE__, __FILE__);
  }
  long blocked_cells = -1;
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  int nthreads = (int)(ncpu < 1 ? 1 : (ncpu > 16 ? 16 : ncpu));
  if (getenv("LBM_PARSER_THREADS")) nthreads = atoi(getenv("LBM_PARSER_THREADS"));
  if (nthreads > 64) nthreads = 64;
  if ((size_t)nthreads > t.len / 65536) nthreads = (int)(t.len / 65536);  /* at least 64 KB of text per thread */
  if (getenv("LBM_PARSER_THREADS") && atoi(getenv("LBM_PARSER_THREADS")) > 1 && t.len > 0)  /* tests: force the parallel path */
    nthreads = atoi(getenv("LBM_PARSER_THREADS")) > 64 ? 64 : atoi(getenv("LBM_PARSER_THREADS"));
  if (nthreads > 1) {
    ob_job jobs[64];
    memset(jobs, 0, sizeof jobs);
    const char *cut = t.base;
    for (int i = 0; i < nthreads; i++) {
      const char *stop = (i == nthreads - 1) ? t.end : t.base + (size_t)((double)t.len * (i + 1) / nthreads);
      if (stop < cut) stop = cut;
      while (stop < t.end && !is_space_ch(*stop)) stop++;   /* never inside a token */
      jobs[i].beg = cut; jobs[i].end = stop; jobs[i].file_end = t.end;
      jobs[i].params = params; jobs[i].obstacles = obstacles;
      cut = stop;
    }
    run_jobs(jobs, nthreads, ob_count_tokens);
    int malformed = 0;
    long total = 0;
    for (int i = 0; i < nthreads; i++) {
      malformed |= jobs[i].malformed;
      jobs[i].first_token = total;
      total += jobs[i].ntokens;
    }
    if (!malformed) {
      run_jobs(jobs, nthreads, ob_convert_records);
      long bad = -1;
      int kind = OB_OK;
      blocked_cells = 0;
      for (int i = 0; i < nthreads; i++) {
        blocked_cells += jobs[i].blocked_cells;
        if (jobs[i].err_record >= 0 && (bad < 0 || jobs[i].err_record < bad)) { bad = jobs[i].err_record; kind = jobs[i].err_kind; }
      }
      /* same checks, same messages, first offending record first (d2q9-bgk.c:573-580) */
      if (kind == OB_FIELDS) die("expected 3 values per line in obstacle file", __LINE__, __FILE__);
      if (kind == OB_XRANGE) die("obstacle x-coord out of range", __LINE__, __FILE__);
      if (kind == OB_YRANGE) die("obstacle y-coord out of range", __LINE__, __FILE__);
      if (kind == OB_VALUE) die("obstacle blocked value should be 1", __LINE__, __FILE__);
    }
  }
  if (blocked_cells < 0) blocked_cells = scan_obstacles_serial(&t, params, obstacles);
  unmap_file(&t);
  params->free_cells_inv = 1.0f / ((long)params->nx * params->ny - blocked_cells);
}

/* ---- output stage: d2q9-bgk.c:772-856 ---------------------------------------------------------- */
/* Same bytes as the reference's fprintf loop (d2q9-bgk.c:835: "%d %d %.12E %.12E %.12E %.12E %d\n"), but the
 * rows of a block are formatted by several threads into memory and written with one fwrite per thread
 * buffer: the 1024x1024 file (1 M lines, 90 MB) takes tens of milliseconds instead of about a second. */

#define LINE_MAX_BYTES 128 /* 2 ints (<= 11 chars) + 4 x 19 chars + flag + separators */

typedef struct {
  const lbm_params *params;
  const float *u_x, *u_y, *u, *pressure;
  const int32_t *obstacles;
  int row_begin, row_end;
  char *buf;
  size_t len;
} format_job;

static void *format_rows(void *arg)
{
  format_job *j = (format_job *)arg;
  const int nx = j->params->nx;
  char *p = j->buf;
  for (int ii = j->row_begin; ii < j->row_end; ii++) {
    for (int jj = 0; jj < nx; jj++) {
      const size_t c = (size_t)ii * nx + jj;
      p += sprintf(p, "%d %d %.12E %.12E %.12E %.12E %d\n", jj, ii, j->u_x[c], j->u_y[c], j->u[c], j->pressure[c],
                   j->obstacles[c]);
    }
  }
  j->len = (size_t)(p - j->buf);
  return NULL;
}

static int write_values(const lbm_params *params, const float *u_x, const float *u_y, const float *u,
                        const float *pressure, const int32_t *obstacles, const float *av_vels)
{
  FILE *fp = fopen(FINALSTATEFILE, "w");
  if (fp == NULL) die("could not open file output file", __LINE__, __FILE__);
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  int nthreads = (int)(ncpu < 1 ? 1 : (ncpu > 16 ? 16 : ncpu));
  if (getenv("LBM_WRITER_THREADS")) nthreads = atoi(getenv("LBM_WRITER_THREADS"));
  if (nthreads < 1) nthreads = 1;
  /* rows per thread and block: at most ~32 MB of text per thread in flight */
  int rows_per_job = (int)((32u << 20) / ((size_t)params->nx * LINE_MAX_BYTES));
  if (rows_per_job < 1) rows_per_job = 1;
  format_job *jobs = (format_job *)calloc((size_t)nthreads, sizeof(format_job));
  pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  if (!jobs || !tids) die("cannot allocate memory for the output writer", __LINE__, __FILE__);
  for (int t = 0; t < nthreads; t++) {
    jobs[t].buf = (char *)malloc((size_t)rows_per_job * params->nx * LINE_MAX_BYTES + 1);
    if (!jobs[t].buf) die("cannot allocate memory for the output writer", __LINE__, __FILE__);
  }
  for (int row = 0; row < params->ny;) {
    int used = 0;
    for (int t = 0; t < nthreads && row < params->ny; t++, used++) {
      format_job *j = &jobs[t];
      j->params = params; j->u_x = u_x; j->u_y = u_y; j->u = u; j->pressure = pressure; j->obstacles = obstacles;
      j->row_begin = row;
      j->row_end = row + rows_per_job < params->ny ? row + rows_per_job : params->ny;
      row = j->row_end;
      if (pthread_create(&tids[t], NULL, format_rows, j) != 0) {  /* no threads available: format inline */
        format_rows(j);
        tids[t] = 0;
      }
    }
    for (int t = 0; t < used; t++) {
      if (tids[t]) pthread_join(tids[t], NULL);
      if (fwrite(jobs[t].buf, 1, jobs[t].len, fp) != jobs[t].len) die("could not write output file", __LINE__, __FILE__);
    }
  }
  for (int t = 0; t < nthreads; t++) free(jobs[t].buf);
  free(jobs);
  free(tids);
  fclose(fp);

  fp = fopen(AVVELSFILE, "w");
  if (fp == NULL) die("could not open file output file", __LINE__, __FILE__);
  for (int ii = 0; ii < params->max_iters; ii++) fprintf(fp, "%d:\t%.12E\n", ii, av_vels[ii]);
  fclose(fp);
  return EXIT_SUCCESS;
}

static int parse_devices(int *devs, int max)
{
  const char *list = getenv("LBM_DEVICES");
  int n = 0;
  if (list && *list) {
    char *copy = strdup(list);
    for (char *tok = strtok(copy, ","); tok && n < max; tok = strtok(NULL, ",")) devs[n++] = atoi(tok);
    free(copy);
    return n;
  }
  const char *ng = getenv("LBM_NGPUS");
  if (ng && atoi(ng) > 1) {
    n = atoi(ng) < max ? atoi(ng) : max;
    for (int i = 0; i < n; i++) devs[i] = i;
  }
  return n;
}

int main(int argc, char *argv[])
{
  if (argc != 3) usage(argv[0]);
  const char *paramfile = argv[1];
  const char *obstaclefile = argv[2];

  /* initialise: parameters, obstacle map, context (d2q9-bgk.c:194) */
  lbm_params params;
  memset(&params, 0, sizeof params);
  load_params(paramfile, &params);
  if (getenv("LBM_MAX_ITERS")) params.max_iters = atoi(getenv("LBM_MAX_ITERS"));
  const size_t ncells = (size_t)params.nx * params.ny;
  int32_t *obstacles = (int32_t *)calloc(ncells, sizeof(int32_t));
  if (obstacles == NULL) die("cannot allocate column memory for obstacles", __LINE__, __FILE__);
  load_obstacles(obstaclefile, &params, obstacles);

  int devs[64];
  const int ndev = parse_devices(devs, 64);
  lbm_ctx *ctx = NULL;
  check_lbm(lbm_create(&ctx, &params, obstacles, ndev > 0 ? ndev : 1, ndev > 0 ? devs : NULL), "creating context", __LINE__);

  /* the read-back targets (the reference mallocs them in initialise(), d2q9-bgk.c:519-526,597 — before its timed region):
   * page-locked, so that the device -> host copies inside the timed region run at the PCIe rate instead of faulting in
   * every page of a fresh malloc */
  float *av_vels = NULL, *fields = NULL;
  check_lbm(lbm_host_alloc((void **)&av_vels, sizeof(float) * (size_t)(params.max_iters > 0 ? params.max_iters : 1)),
            "allocating memory for av_vels", __LINE__);
  check_lbm(lbm_host_alloc((void **)&fields, sizeof(float) * 4 * ncells), "allocating memory for output fields", __LINE__);

  float *cells = NULL;
  if (getenv("LBM_HOST_INIT")) {
    /* d2q9-bgk.c:519-550 */
    cells = (float *)malloc(sizeof(float) * NSPEEDS * ncells);
    if (cells == NULL) die("cannot allocate memory for cells", __LINE__, __FILE__);
    const float w0 = params.density * 4.0f / 9.0f, w1 = params.density / 9.0f, w2 = params.density / 36.0f;
    for (size_t i = 0; i < ncells; i++) {
      cells[i] = w0;
      for (int k = 1; k <= 4; k++) cells[k * ncells + i] = w1;
      for (int k = 5; k <= 8; k++) cells[k * ncells + i] = w2;
    }
  }

  struct timeval timstr;
  struct rusage ru;
  gettimeofday(&timstr, NULL);
  const double tic = timstr.tv_sec + (timstr.tv_usec / 1000000.0);

  check_lbm(lbm_upload(ctx, cells), "writing cells data", __LINE__);
  /* the reference times its obstacle transfer too (d2q9-bgk.c:205-209) */
  check_lbm(lbm_upload_obstacles(ctx, obstacles), "writing obstacles data", __LINE__);
  double loop_ms = 0.0;
  check_lbm(lbm_run_timed(ctx, params.max_iters, &loop_ms), "running timesteps", __LINE__);
  check_lbm(lbm_sync(ctx), "waiting for queue", __LINE__);
  check_lbm(lbm_download(ctx, NULL, av_vels), "reading av_vels data", __LINE__);
  /* the reference reads the whole 9-plane state back and derives the output columns on the host
   * (d2q9-bgk.c:251-253,787-832); here the device computes the four columns and only those move */
  check_lbm(lbm_final_state(ctx, fields, fields + ncells, fields + 2 * ncells, fields + 3 * ncells),
            "reading cells data", __LINE__);
  float reynolds = 0.0f;
  check_lbm(lbm_reynolds(ctx, &reynolds), "computing reynolds number", __LINE__);

  gettimeofday(&timstr, NULL);
  const double toc = timstr.tv_sec + (timstr.tv_usec / 1000000.0);
  getrusage(RUSAGE_SELF, &ru);
  const double usrtim = ru.ru_utime.tv_sec + (ru.ru_utime.tv_usec / 1000000.0);
  const double systim = ru.ru_stime.tv_sec + (ru.ru_stime.tv_usec / 1000000.0);

  /* d2q9-bgk.c:271-275 */
  printf("==done==\n");
  printf("Reynolds number:\t\t%.12E\n", reynolds);
  printf("Elapsed time:\t\t\t%.6lf (s)\n", toc - tic);
  printf("Elapsed user CPU time:\t\t%.6lf (s)\n", usrtim);
  printf("Elapsed system CPU time:\t%.6lf (s)\n", systim);
  /* extra labelled lines (not in the reference) */
  const double lu = (double)ncells * params.max_iters;
  if (params.max_iters > 0 && loop_ms > 0.0) {
    printf("Step loop time:\t\t\t%.6lf (s)\n", loop_ms * 1e-3);
    printf("MLUPS (step loop):\t\t%.1f\n", lu / (loop_ms * 1e-3) / 1e6);
    printf("MLUPS (elapsed time):\t\t%.1f\n", lu / (toc - tic) / 1e6);
    /* 72 B per lattice update x updates per second: what a one-step-per-launch kernel would have to move; the
     * multi-step kernels keep the intermediate states on the chip, so this is NOT a memory bandwidth */
    printf("Algorithmic GB/s (72 B/LU):\t%.1f\n", 72.0 * lu / (loop_ms * 1e-3) / 1e9);
  }
  if (!getenv("LBM_NO_OUTPUT"))
    write_values(&params, fields, fields + ncells, fields + 2 * ncells, fields + 3 * ncells, obstacles, av_vels);

  /* finalise (d2q9-bgk.c:715-744) */
  lbm_destroy(ctx);
  lbm_host_free(fields);
  lbm_host_free(av_vels);
  free(cells);
  free(obstacles);
  return EXIT_SUCCESS;
}
