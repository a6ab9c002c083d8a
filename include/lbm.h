/*
 * lbm.h — C ABI of liblbm_hip.so: the MI355X-native D2Q9-BGK timestep.
 *
 * This is the drop-in boundary for the reference's host<->device seam.  The reference
 * (ag14774/OpenCL-Lattice-Boltzmann) has no plugin API; its hot path sits behind the OpenCL
 * calls made by d2q9-bgk.c.  Each entry point below replaces the cited call site(s); a host
 * written against this header needs no OpenCL, no kernels.cl and no JIT.
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no C++/torch types;
 *   - every function returns LBM_OK (0) or a non-zero LBM_ERR_* code; lbm_last_error() returns
 *     the message of the calling thread's most recent failure (the reference's checkError()
 *     prints such a message and exits, d2q9-bgk.c:858-866 — the host does the same);
 *   - the caller owns all host arrays (d2q9-bgk.c:519-526,597,720-727); the library owns all
 *     device memory (d2q9-bgk.c:687-710,729-733);
 *   - one host thread drives a context; calls are asynchronous on the context's HIP streams and
 *     ordered like the reference's single in-order queue (d2q9-bgk.c:612-616);
 *   - cells layout = the reference's SoA: float[9][ny][nx], speed k at k*nx*ny + y*nx + x
 *     (d2q9-bgk.c:73, kernels.cl:7), speeds numbered 0 rest, 1 E, 2 N, 3 W, 4 S, 5 NE, 6 NW,
 *     7 SW, 8 SE (d2q9-bgk.c:7-13); obstacles = int32[ny][nx], 0 fluid / non-zero blocked.
 */
#ifndef LBM_H
#define LBM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_OK 0
#define LBM_ERR_ARG 1    /* bad argument */
#define LBM_ERR_HIP 2    /* HIP runtime error (message has the hipError string) */
#define LBM_ERR_STATE 3  /* call not valid in the context's current state */
#define LBM_ERR_COMM 4   /* halo transport error (RCCL, peer mapping, a neighbour that never arrived) */

/* Run constants: the reference's t_param (d2q9-bgk.c:81-92), same fields. */
typedef struct lbm_params {
  int nx;               /* cells in x */
  int ny;               /* cells in y */
  int max_iters;        /* capacity of the av_vels record (steps that can be run) */
  int reynolds_dim;     /* dimension for the Reynolds number */
  float density;        /* density per link */
  float accel;          /* density redistribution */
  float omega;          /* relaxation parameter */
  float free_cells_inv; /* 1 / number of non-blocked cells (d2q9-bgk.c:591) */
} lbm_params;

typedef struct lbm_ctx lbm_ctx; /* opaque; replaces t_ocl (d2q9-bgk.c:97-119) */

/*
 * Create a context: select device(s), allocate the two cell grids, the obstacle mask and the
 * reduction buffers.  Replaces selectOpenCLDevice + clCreateContext/Queue/Program/Kernel/Buffer
 * (d2q9-bgk.c:600-710, 885-944) and the obstacle upload (d2q9-bgk.c:205-209).
 *   obstacles  borrowed for the duration of the call.
 *   ndev       number of row slabs; the grid is row-partitioned over dev_ids[0..ndev).  ndev <= 1
 *              with dev_ids == NULL uses the current HIP device.  A device may be listed more than
 *              once (several slabs on one GPU).  Every slab needs at least 4 rows.  Halo rows move by
 *              the PEER transport (a kernel stores them straight into the neighbour slab's halo rows,
 *              peer access between distinct devices) unless lbm_set_default("transport") asks for
 *              RCCL send/recv (distinct devices only) or device-to-device copies.
 */
int lbm_create(lbm_ctx **out, const lbm_params *params, const int32_t *obstacles, int ndev, const int *dev_ids);

/*
 * One-process-per-GPU form (torchrun / MPI style launch): this process owns slab `rank` of
 * `nranks` on HIP device `device`; halos are exchanged with ranks (rank±1) mod nranks by RCCL
 * send/recv and the velocity sums are combined by an RCCL all-reduce.  `comm_id` is the
 * lbm_comm_id_size()-byte blob produced by lbm_comm_get_id() on one rank and distributed by the
 * caller (e.g. torch.distributed broadcast).  params/obstacles describe the GLOBAL grid.
 * No counterpart in the reference (single device); mandated by the multi-GPU configs.
 * comm_id == NULL creates the rank without RCCL communicator: it must then be connected with
 * lbm_connect_peers before the first lbm_run, and lbm_download returns the velocity sums over THIS rank's
 * rows only (times free_cells_inv) — the caller adds the ranks' records.
 */
int lbm_create_rank(lbm_ctx **out, const lbm_params *params, const int32_t *obstacles,
                    int rank, int nranks, int device, const void *comm_id);
size_t lbm_comm_id_size(void);
int lbm_comm_get_id(void *comm_id_out);

/*
 * PEER halo transport between ranks (the low-latency alternative to RCCL send/recv for the halo rows; the
 * all-reduce stays with RCCL).  Each rank describes its two grids and its flag words in an
 * lbm_peer_info_size()-byte blob (HIP IPC handles + the raw pointers for neighbours inside the same process);
 * the caller distributes the blobs (e.g. torch.distributed all_gather) and hands every rank the blobs of its ring
 * neighbours (rank-1) mod nranks and (rank+1) mod nranks.  From then on a launch set's edge rows are stored
 * straight into the neighbours' halo rows by a push kernel (xGMI peer writes) which then raises a sequence
 * number in the neighbour's flag word; the consumer waits for it with a bounded spin (30 s; reported by lbm_sync
 * as LBM_ERR_COMM) or hipStreamWaitValue32 (option "halo_sync").  lbm_connect_peers is a collective decision:
 * every rank of the ring must call it (or none).  A context WITHOUT communicator runs on peer stores from then on; a
 * context that also has an RCCL communicator stays on RCCL send/recv until lbm_set_option("transport", 3) (or default
 * "transport" = 3 at creation) — peer stores between devices are the caller's explicit choice, to be made once they
 * have been checked against the RCCL ring on the machine at hand (bench.py: transport_check); "transport" 1 | 3
 * switches back and forth.  A blob is recognised as "my own process" by a per-process random nonce + boot id, not by
 * the pid.  A failed call leaves nothing mapped; a repeated call unmaps the previous neighbours first.
 */
size_t lbm_peer_info_size(void);
int lbm_peer_info(lbm_ctx *ctx, void *info_out);
int lbm_connect_peers(lbm_ctx *ctx, const void *south_info, const void *north_info);
/*
 * Unmap the ring neighbours' grids and flag words (waits for this context's queued work first).  ORDER OF TEAR-DOWN
 * between processes: every rank calls lbm_disconnect_peers, the caller synchronises the ranks (a barrier), and only then
 * does any rank call lbm_destroy — freeing device memory that another process still has mapped through HIP IPC is
 * undefined behaviour (as with CUDA IPC).  lbm_destroy alone is enough inside one process and for a rank whose
 * neighbours have already gone.  Afterwards the context is back on RCCL send/recv if it has a communicator, else
 * without transport until lbm_connect_peers is called again.
 */
int lbm_disconnect_peers(lbm_ctx *ctx);

/*
 * Process-wide defaults for contexts created afterwards (validated; replace the environment hooks of round 1):
 *   "force_halo"  0 | 1   a single slab also carries halo rows and exchanges them with itself (a ring of one): the
 *                         whole multi-GPU machinery of a rank on one GPU (tests, tools/scaling_projection.py)
 *   "halo_depth"  0 = by slab size (8 small slabs and slabs from 3M cells / 4 from 2M cells / 3 / 2 thin slabs), else 2..8
 *   "transport"   0 = auto, 1 = RCCL send/recv, 2 = device-to-device copies, 3 = peer stores
 *   "lanes_out"   0 = auto (60), else 4..62: output lanes per 64-lane strip of the window kernels
 */
int lbm_set_default(const char *key, long value);

/*
 * Host -> device copy of the initial state.  Replaces clEnqueueWriteBuffer(cells)
 * (d2q9-bgk.c:200-203).  cells == NULL initialises the uniform rest state on the device
 * (the values of d2q9-bgk.c:529-550) without any host transfer.  Resets the step counter.
 * In rank mode `cells` is the GLOBAL array; only this rank's rows are read.
 */
int lbm_upload(lbm_ctx *ctx, const float *cells);

/*
 * Host -> device copy of the obstacle map: replaces clEnqueueWriteBuffer(obstacles) (d2q9-bgk.c:205-209).  lbm_create
 * has already taken the map it was given (it needs it to lay out the slabs); this entry point exists so that a host
 * that follows the reference's timing rule — everything from the first host->device transfer to the last read-back
 * inside the timed region, d2q9-bgk.c:196-263 — can have the obstacle transfer inside it, and so that a caller may
 * change the map between runs (same nx, ny; params.free_cells_inv is the caller's to keep consistent: it was fixed
 * at lbm_create).  obstacles = int32[ny][nx] of the GLOBAL grid, borrowed.  Synchronises.  Also rebuilds what the library
 * derives from the map (the per-strip bits of rows with blocked cells that option "free_sweeps" consults: one kernel, ~75 us
 * at 8192 x 8192).
 */
int lbm_upload_obstacles(lbm_ctx *ctx, const int32_t *obstacles);

/*
 * Advance nsteps timesteps (accelerate_flow + timestep + av_vels reduction each); asynchronous.
 * Replaces the loop body d2q9-bgk.c:221-238 (accelerate_flow(), timestep(), reduce() wrappers,
 * d2q9-bgk.c:282-393).  May be called repeatedly; after n calls' worth of steps exactly that many
 * steps have been applied and av_vels[t] is defined for every one of them.
 */
int lbm_run(lbm_ctx *ctx, int nsteps);

/* lbm_run + device-side timing: *ms = elapsed time of the nsteps step loop measured with HIP
 * events recorded on the stream the kernels run on (max over slabs).  Synchronises. */
int lbm_run_timed(lbm_ctx *ctx, int nsteps, double *ms);

/*
 * lbm_run with timing events around every launch of the context's first slab (at most 64 launch sets are
 * recorded; run few steps).  Synchronises.  stats[8] (microseconds are means over the recorded launch sets):
 *   [0] launch sets recorded   [1] timesteps per launch set (mean)
 *   [2] edge launch us         [3] halo exchange us (from the end of the edge launch to the end of the push
 *                                  kernel / RCCL send+recv on the edge stream; the last set of a run has none)
 *   [4] interior launch us (the only launch of a context without halo rows)
 *   [5] launch-set period us (start of an interior launch to the start of the next)
 *   [6] start of the interior launch after the start of the edge launch, us   [7] transport in use
 * Diagnostic only (the events cost a few microseconds per set): bench.py --gpus N prints it per rank so that a
 * multi-GPU record explains where a launch set's time went.
 */
int lbm_run_profiled(lbm_ctx *ctx, int nsteps, double *stats);

/* Wait for all queued work.  Replaces clFinish (d2q9-bgk.c:239). */
int lbm_sync(lbm_ctx *ctx);

/*
 * Device -> host.  Replaces the two clEnqueueReadBuffer calls (d2q9-bgk.c:251-260).  Either
 * pointer may be NULL.  cells_out receives the CURRENT state whatever the step parity (the
 * reference reads a fixed buffer, correct only for even step counts).  av_vels_out receives
 * lbm_steps_done() floats.  Synchronises.  In rank mode cells_out is the GLOBAL array and only
 * this rank's rows are written; av_vels_out is the all-reduced global record — asking for it is a
 * collective operation: every rank must make the same call.  lbm_reynolds is collective in rank mode too.
 */
int lbm_download(lbm_ctx *ctx, float *cells_out, float *av_vels_out);

/* Steps applied since the last lbm_upload. */
int lbm_steps_done(const lbm_ctx *ctx);

/* Rows [*y0, *y1) of the global grid held by this context (whole grid unless rank mode). */
int lbm_row_range(const lbm_ctx *ctx, int *y0, int *y1);

/*
 * Output stage on the device: the per-cell columns of final_state.dat (d2q9-bgk.c:787-832:
 * u_x, u_y, speed u, pressure; obstacle cells give 0,0,0,density/3) and the Reynolds number of
 * the current state (av_velocity + calc_reynolds, d2q9-bgk.c:396-442,747-752).  Each output is
 * float[ny][nx] (rank mode: global array, own rows written) and may be NULL.  Synchronises.
 */
int lbm_final_state(lbm_ctx *ctx, float *u_x, float *u_y, float *u, float *pressure);
int lbm_reynolds(lbm_ctx *ctx, float *reynolds_out);

/*
 * Tuning knobs (all optional; defaults are the measured-best on MI355X):
 *   "variant"      how a thread gets its x-1/x+1 neighbours: 0 = auto, 1 = scalar L1 loads,
 *                  2 = unaligned 16-byte loads, 3 = wave64 DPP shifts, 4 = LDS-staged row with halo
 *   "fuse"         1 (or 2) = advance two timesteps per launch (intermediate state kept in registers, half
 *                  the HBM traffic), 3 = three timesteps per launch (two windows of intermediate rows, a third
 *                  of the traffic), 4 = four timesteps per launch (with row slabs only where the halo rows are 4
 *                  deep — slabs of 2M cells and more — else 3), 6..8 = the deep window kernel (lanes of two cells) with
 *                  at most that many timesteps per launch: a run is cut into the fewest launches, of equal depth
 *                  (20 steps = 7 + 7 + 6); with row slabs capped by the halo depth (8 for slabs of 3M cells and more);
 *                  falls back to 4 on grids under 32 rows per slab.  0 = one launch per step, -1 = auto (by size: 8
 *                  from 3M cells per slab; one slab without halo rows: above 300K cells, always as chunk pairs).  5 = the
 *                  chunk pairs at five steps per launch set on row slabs that carry five halo rows (slabs of 240K to 3M cells)
 *                  in compact launch sets — what auto chooses there; the four-step kernel anywhere else.
 *   "twin_steps"   chunk-pair form of the deep window kernel ("pair"): most timesteps per launch, 2..8, 0 = auto (5 below
 *                  3M cells, 8 from there on)
 *   "steady"       deep window kernel: -1/1 = a launch of exactly 5 (chunk pairs below 3M cells), 6, 7 or 8 timesteps runs the
 *                  kernel instantiated for that depth (general form of the row loop while the levels start up, then a steady
 *                  form whose level chain is straight-line code), 0 = the any-depth kernel always.  Same results bit for bit.
 *   "edge_aware"   deep window kernel with row slabs: -1/1 = one-round interior schedule whose last units take over the
 *                  wave slots of the edge launch, 0 = slots reserved for the whole launch set
 *   "obst_paths"   deep window kernel: 1 (and -1, auto) = waves that hold no blocked cell take a collision path without
 *                  the bounce-back selects, 0 = one path
 *   "balance"      deep window kernel, launches that are one round of work units: 1 (and -1, auto) = the strips that hold blocked
 *                  cells in most rows (at most four: a cavity's wall strips) are worked on by twice the waves, each on half of
 *                  every chunk of rows, so that the launch does not end with them; 0 = every strip the same.  Reads back as
 *                  the number of strips doubled.  Same results bit for bit.
 *   "free_sweeps"  deep window kernel at 6, 7, 8 timesteps per launch: 1 (and -1, auto) = a wave whose chunk of rows holds no
 *                  blocked cell inside its strip (looked up in a map built from the obstacle map) sweeps it without any
 *                  obstacle handling, 0 = every wave looks level by level ("obst_paths").  Same results bit for bit.
 *   "windows", "load_bufs"   kept for callers that set them: the three-step kernel exists in ONE form since round 4 — its two
 *                  windows in LDS (two waves per SIMD), one row-set of source loads in flight; "windows" accepts -1 / 1,
 *                  "load_bufs" 0 / 1 (the register-window form and the two-row-set form, 253-256 registers at one wave per
 *                  SIMD, were measured slower in rounds 1-2 and no policy selected them)
 *   "sched_waves"  waves per SIMD the three- / four-step kernels' chunk schedule plans for: 1 or 2, 0 = auto
 *   "pair"         chunk-pair form of the three- / four-step kernels and of the deep window kernel (two chunks that start
 *                  at a common boundary run as one workgroup and hand each other their first rows instead of computing
 *                  them twice): 1 = always, 0 = never, -1 = auto (where all units of a launch are resident at once; the
 *                  deep window kernel: one slab without halo rows; with row slabs the interior of a compact launch set —
 *                  peer stores — from about 650 rows per slab at 8192 cells a row)
 *   "multistep"    T = 1..8: advance T timesteps per launch on LDS-resident tiles (small, launch-bound
 *                  grids), 0 = off, -1 = auto.  Single-slab grids only; takes precedence over "fuse".
 *   "resident"     1 = run all timesteps of an lbm_run (up to 256 per launch: the ring of per-step sums) in ONE launch with the
 *                  grid held in registers (d2q9_resident: bands of 2, 4 or 6 full-width rows, one to eight waves across; neighbouring
 *                  bands trade their edge rows through memory behind step words, bounded by "halo_timeout_ms") where the grid
 *                  allows it — one slab without halo rows, nx a multiple of 4 from 128 to 1024 (a band's last wave may be partly filled), ny a
 *                  multiple of the band height, all
 *                  bands resident on the device at once (up to 1.5M cells on 256 CUs) —, 0 = never, -1 = auto: from 200K cells
 *                  while "fuse" and "multistep" are on auto.  Reads back as the rows per band in use (0 = not in use).
 *                  Bit-identical to single steps.  A band that waits in vain ends the run with LBM_ERR_COMM.  Such a launch needs
 *                  the whole device: the library orders the resident launches of all contexts of a process one behind the other;
 *                  whatever else fills the device meanwhile (another process, a long kernel of the caller's) delays it.
 *   "chunk_rows"   most rows swept by one wave of the two-step kernel (0 = auto)
 *   "chunk_min"    fewest rows per wave at the tapered end of the schedule (0 = auto)
 *   "grid_blocks"  cap on workgroups per launch (0 = auto)
 *   "nt_stores"    1 = non-temporal stores for the destination grid, 0 = plain, -1 = auto
 *   "nt_loads"     source loads of the two-step kernel: 0 = plain, 1 = non-temporal, 2 = non-temporal except for the rows shared
 *                  with the neighbouring chunk, -1 = auto (2).  The three- and four-step kernels always load plain (with two waves
 *                  per SIMD plain loads won at every size; the four-step kernel's non-temporal forms spilled and were removed).
 *   "transport"    halo transport of a context that carries halo rows: 1 = RCCL send/recv (needs a communicator),
 *                  3 = peer stores (needs connected peers); read-only values 2 = device-to-device copies, 0 = none yet
 *   "halo_sync"    peer transport, consumer side: 0 = wait kernel with a bounded spin (default), 1 = hipStreamWaitValue32
 *                  (UNBOUNDED: the stream waits for a neighbour that died for ever; diagnostic use), 2 = the edge tiles /
 *                  edge chunks of the consuming launch poll the flag words themselves (compact launch sets only,
 *                  elsewhere like 0; refused with LBM_ERR_STATE while a ring neighbour lives on another device or in
 *                  another process on another device — there the halo rows are read behind the wait kernel's
 *                  kernel-start acquire)
 *   "halo_timeout_ms"  bound of that spin, 1..600000 (default 30000).  ONE timeout per run: the first wait that gives up
 *                  raises the slab's error word and every later wait falls through at once, so the launch sets already
 *                  queued drain at kernel speed; lbm_sync / lbm_download then return LBM_ERR_COMM and the context
 *                  accepts only lbm_destroy
 *   "push_release" peer transport, producer side: how a pushing wave orders its halo rows before the flag words go up.  1 = release
 *                  fences at system scope around the ticket (the HSA memory model's guarantee, whatever path the stores took),
 *                  0 = write-through stores drained with s_waitcnt (measured and soak-tested between slabs and processes on ONE
 *                  device only), -1 = auto: 1 whenever a ring neighbour lives on another device.  Same results bit for bit.
 *   "debug_stale_exchange"  TEST HOOK, n >= 1: the n-th halo exchange from now announces itself (flag words, events) but
 *                  delivers no rows — on every rank of the ring alike —, 0 = off; clears itself when it fires.  Exists so that the
 *                  checks above the library (bench.py: transport_check) can be shown to catch a transport that loses halo rows.
 *   "compact"      row slabs: -1/1 = one launch per launch set on one stream, its first workgroups — the edge tiles /
 *                  edge chunks — store the halo rows into the neighbours themselves (peer transport: LDS-tile kernel, three- /
 *                  four-step kernels, deep window kernel incl. its five-step chunk pairs on slabs of 300K to 3M cells) or, under
 *                  the RCCL transport on slabs that run the deep window kernel (from 3M cells) or its five-step chunk pairs
 *                  (540K to 3M cells), into staging blocks which the edge stream sends once the flag word of the launch's last
 *                  edge wave is up (hipStreamWaitValue32: "staged" launch sets);
 *                  0 = edge launch / interior launch / push kernel (or RCCL exchange) on two streams
 * Read-only through lbm_get_option: "nslabs", "fuse_units", "halo_depth", "launch_steps" (most timesteps one launch of
 * the context's main kernel advances).
 */
int lbm_set_option(lbm_ctx *ctx, const char *key, long value);
int lbm_get_option(const lbm_ctx *ctx, const char *key, long *value);

/* Roofline denominator: float4 device-to-device copy kernel (read bytes + written bytes per
 * second, in GB/s) on the current device; bytes is the size of each of the two buffers. */
int lbm_copy_bandwidth(size_t bytes, int iters, double *gbps);

/* The other roofline denominator, for the kernels that are bound by instruction issue (d2q9_deep): the rate at which the
 * device issues packed fp32 fused multiply-adds (eight independent chains per thread, eight waves per SIMD), in 1e12
 * lane-instructions per second, over `launches` launches of ~2 ms.  (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3.) */
int lbm_valu_rate(int launches, double *tera_lane_instr_per_s);

/*
 * Page-locked host memory for the read-back targets (the reference mallocs them in initialise(), d2q9-bgk.c:519-526,
 * outside its timed region, and frees them in finalise(), :720-727): device -> host copies of lbm_download and
 * lbm_final_state into such a buffer run at the PCIe rate (~55 GB/s) instead of the ~20 GB/s a freshly malloc'ed,
 * never-touched pageable buffer gets (every page faults inside the copy) — SURVEY 8 (f2): "so the reference-rule wall
 * time is not dominated by I/O".  Optional: every entry point takes any host pointer.  bytes > 0; lbm_host_free(NULL) is
 * a no-op.  The memory belongs to the caller until lbm_host_free; free it before the process's last lbm_destroy or after,
 * either order is valid.
 */
int lbm_host_alloc(void **ptr_out, size_t bytes);
int lbm_host_free(void *ptr);

/* Release everything.  Replaces the clRelease* block of finalise (d2q9-bgk.c:729-741). */
void lbm_destroy(lbm_ctx *ctx);

const char *lbm_last_error(void);
const char *lbm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LBM_H */
