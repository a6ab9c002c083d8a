// liblbm_hip.so — host side of the C ABI declared in include/lbm.h (compiled with hipcc for gfx950).
//
// Replaces the reference's host<->device seam: context/queue/program/buffer setup
// (d2q9-bgk.c:600-710), the kernel-launch wrappers accelerate_flow()/timestep()/reduce()
// (d2q9-bgk.c:282-393), the ping-pong step loop (d2q9-bgk.c:214-239), the transfers
// (d2q9-bgk.c:200-209,251-260) and the releases (d2q9-bgk.c:729-741).
//
// Structure: a context owns one or more row SLABS.  A slab is a contiguous range of grid rows on
// one GPU with its own pair of grids, mask, streams and reduction buffers.  With one slab the y
// wrap-around is resolved inside the kernels.  With several slabs every slab stores H HALO rows (H = 2, 3,
// 4 or 8: as many as a launch set advances timesteps) below and above its own rows inside the same
// row-interleaved arrays; after each launch set a slab's H bottom and H top rows — each one contiguous
// block — go to its ring neighbours' halo rows.  Three transports: PEER (default where it can be set up)
// = a push kernel stores the rows straight into the neighbour's memory (same process, peer device, or
// another process through HIP IPC) and raises a sequence number in the neighbour's flag word; RCCL =
// grouped ncclSend/ncclRecv over xGMI; COPY = device-to-device hipMemcpy between slabs of one process.
// The chunks next to the slab edges are computed first on an edge stream, so the exchange overlaps the
// interior work on the main stream; events join the two streams once per launch set.
#include "../../include/lbm.h"
#include "d2q9_kernels.h"
#include "deep_instances.h"
#include "halo_exchange.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(LBM_ERR_HIP, "HIP error during '%s' (%s:%d): %s", #expr, __FILE__, __LINE__, \
                  hipGetErrorString(e_));                                                      \
  } while (0)

// ---- RCCL, loaded on first multi-GPU use (a single-GPU run never touches it) ------------------
struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return LBM_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(LBM_ERR_COMM, "cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                  \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (!g_rccl.field) return fail(LBM_ERR_COMM, "librccl lacks %s", name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.handle = h;
  return LBM_OK;
}

#define NCCL_TRY(expr)                                                                          \
  do {                                                                                          \
    ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess)                                                                      \
      return fail(LBM_ERR_COMM, "RCCL error during '%s' (%s:%d): %s", #expr, __FILE__, __LINE__, \
                  g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?");                      \
  } while (0)

constexpr bool kStep3LdsDefault = true;   // d2q9_step3 windows in LDS unless option "windows" says otherwise
constexpr int kDeepSteps = 8;    // most timesteps per launch of d2q9_deep (option fuse = 6..8: the limit of a context)
constexpr int kDeepMin = 6;
constexpr int kDeepTwinSteps = 8; // most timesteps per launch of d2q9_deep_twin (chunk pairs; from 6 on through its mailbox)
// ... and what it uses unless option twin_steps says otherwise (tools/ab.py, GLUPS at 5 / 6 / 8 steps per launch: 1024x1024
// 139 / 135 / 131, 2048x2048 234 / 237 / 236, 4096x4096 295 / 302 / 306 — where the lone kernel at 8 steps reaches 314)
constexpr int kDeepTwinDefault = 5;
constexpr int kRingMax = 256;  // most steps of per-workgroup partial sums buffered between reductions
constexpr int kProfSets = 64, kProfEvents = 5;  // lbm_run_profiled: {edge start, edge end, exchange end, interior start, interior end}
enum { TRANSPORT_AUTO = 0, TRANSPORT_RCCL = 1, TRANSPORT_COPY = 2, TRANSPORT_PEER = 3 };

// Process-wide defaults for contexts created afterwards (lbm_set_default): what used to be environment hooks.
struct Defaults {
  int force_halo = 0;               // 1: even a single slab carries halo rows and exchanges them with itself (ring of one)
  int halo_depth = 0;               // 0 = by slab size, else 2..8
  int transport = TRANSPORT_AUTO;   // halo transport of contexts that may choose
  int lanes_out = 0;                // output lanes per strip of the window kernels: 0 = auto (60), else 4..62
};
Defaults g_defaults;

// one ring neighbour as seen from a slab: its two grids and its flag words, mapped into this process
struct PeerLink {
  float *cells[2] = {nullptr, nullptr};
  uint32_t *flags = nullptr;
  int rows = 0;         // rows the neighbour owns (its top halo starts behind them)
  bool remote = false;  // another device or another process: stores and flags travel over xGMI / through another L2
  bool ipc = false;     // opened with hipIpcOpenMemHandle: closed in free_slab
  bool connected = false;
};

// what lbm_peer_info() hands to the caller for distribution to the ring neighbours
struct PeerInfoBlob {
  uint64_t magic;
  int32_t pid, device;
  int32_t rows, row0;
  uint64_t row_stride;
  int32_t pci[4];                     // domain, bus, device of the GPU (+ pad): "is the neighbour on MY device?" across processes,
                                      // whose device indices may be remapped (HIP_VISIBLE_DEVICES)
  uint64_t nonce[2];                  // identifies the PROCESS that made the blob (see process_nonce): a pid alone repeats
                                      // across PID namespaces (containers) and nodes
  uint64_t cells_ptr[2], flags_ptr;   // valid inside the process with that nonce only
  hipIpcMemHandle_t cells[2], flags;  // for everybody else
};
constexpr uint64_t kPeerMagic = 0x4c424d5045455232ull;  // "LBMPEER2"
constexpr unsigned long long kHaloWaitMsDefault = 30000;  // consumer-side wait for a neighbour's halo rows: 30 s
constexpr unsigned long long kTicksPerMs = 100000ull;     // s_memrealtime runs at 100 MHz

// Two random words drawn once per process (/dev/urandom; clock, pid and an address as a fallback) mixed with the host's
// boot id: "is this blob from my own process, so that its raw pointers are valid here?" must not be answered by the
// pid, which two containers or two nodes can share.
const uint64_t *process_nonce() {
  static uint64_t n[2] = {0, 0};
  static bool done = false;
  if (!done) {
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) {
      if (fread(n, sizeof n, 1, f) != 1) n[0] = n[1] = 0;
      fclose(f);
    }
    if (n[0] == 0 && n[1] == 0) {
      struct timespec ts;
      clock_gettime(CLOCK_REALTIME, &ts);
      n[0] = (uint64_t)ts.tv_nsec * 0x9e3779b97f4a7c15ull ^ (uint64_t)ts.tv_sec;
      n[1] = (uint64_t)getpid() << 32 ^ (uint64_t)(uintptr_t)&n;
    }
    uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a of the boot id
    if (FILE *b = fopen("/proc/sys/kernel/random/boot_id", "r")) {
      int ch;
      while ((ch = fgetc(b)) != EOF) h = (h ^ (uint64_t)(unsigned char)ch) * 0x100000001b3ull;
      fclose(b);
    }
    n[1] ^= h;
    n[0] |= 1;  // never all zero: a zeroed blob matches nobody
    done = true;
  }
  return n;
}

// unit schedule of one d2q9_step2 launch (see fuse_schedule)
struct FuseGeom {
  int *chunk_start = nullptr;  // device array [nchunks+1]
  int nchunks = 0, nbands = 1, units_per_band = 0, units = 0;
  int skip = -1;               // chunk whose units do nothing (the interior, in an edge schedule)
  bool single_round = false;   // all units of the launch are resident at once (equal chunks)
  bool paired = false;         // launched with the chunk-pair kernel (d2q9_step3p / d2q9_step4p)
  // d2q9_deep, one-round schedules: strips that hold blocked cells in most rows appear twice, each copy on half of every
  // chunk (Step2Args::vmap / vtab; balance_heavy_strips)
  std::vector<int> starts;     // host copy of chunk_start
  int vstrips = 0;             // strips of the interior decode (real strips + copies), 0 = not balanced
  int *vmap = nullptr, *vtab = nullptr;
};

struct Slab {
  int dev = 0;
  int cus = 256;          // compute units of the device (hipDeviceProp_t::multiProcessorCount): the schedules plan wave slots from it
  int index = 0;          // position in the global ring of slabs
  int y0 = 0, rows = 0;   // global first row, rows owned
  int row0 = 0;           // halo rows stored below (and above) the owned rows: 0 (one slab), else the halo depth (2 or 8)
  int ext_rows = 0;       // rows stored = rows + 2*row0
  int accel_own = -1;     // stored-row index of global row ny-2 if this slab owns it, else -1
  int accel_ext = -1;     // same, also when the row is one of the stored halo rows
  int accel_ext_b = -1;   // a second stored copy (only a slab that is its own ring neighbour has one)
  size_t plane_stride = 0;  // floats between the 9 plane-rows of a grid row (padded row length)
  size_t row_stride = 0;    // floats between grid rows = 9*plane_stride (+ pad)
  float *cells[2] = {nullptr, nullptr};  // [ext_rows][9][plane_stride]
  uint8_t *mask = nullptr;               // [ext_rows][nx]
  float *partials = nullptr;  // [ring][nb_total]
  int nb_main = 0, nb_edge = 0;  // workgroups of the single-step interior / edge launch
  int nb_total = 0;              // slot stride of the ring
  int strips = 0, lanes_out = 0;  // x decomposition of d2q9_step2
  FuseGeom f_main, f_edge;        // whole slab (one slab) or interior chunks; the two edge chunks
  FuseGeom f3_main;               // schedule of d2q9_step3: its own chunk lengths
  FuseGeom f4_main;               // schedule of d2q9_step4 (one slab only): long chunks
  FuseGeom f6_main;               // schedule of d2q9_deep: its own strips of two-cell lanes (whole slab, or the interior)
  FuseGeom f6_edge;               // slab mode: its edge schedule {bottom edge rows, (interior), top edge rows}
  FuseGeom f6_twin;               // one slab: pair schedule of d2q9_deep_twin (used where it is one round of units)
  int strips2 = 0, lanes2 = 0;    // x decomposition of d2q9_deep: strips per row, output lanes (of two cells) per strip
  int strips_tw = 0, lanes_tw = 0;  // ... of d2q9_deep_twin: at up to five steps per launch two halo lanes per side are enough
  unsigned long long *clean_bits = nullptr;  // d2q9_deep's strips x stored rows: which hold a blocked cell (lbm::strip_row_bits)
  int clean_words = 0;            // 64-row words per strip of that map
  bool all_clean = false;         // no blocked cell anywhere in the slab's stored rows
  std::vector<int> heavy;         // d2q9_deep's strips with a blocked cell in most stored rows (a cavity's wall strips), if few
  int edge_rows = 0;              // rows at each slab edge that the edge launch computes (= halo depth)
  int m_tiles_x = 0, m_tiles_y = 0;  // tiles of d2q9_multi
  int m_tx = 32, m_ty = 16;          // its tile size (chosen by how many tiles the slab gives)
  double *av_sum = nullptr;   // [capacity] per-step sum of |j|/rho over this slab's fluid cells
  hipStream_t s_main = nullptr, s_edge = nullptr;
  hipEvent_t ev_main[2] = {nullptr, nullptr};   // interior launch of a launch set done
  hipEvent_t ev_edgek[2] = {nullptr, nullptr};  // edge launch of a launch set done (edge rows final)
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // timing of lbm_run_timed
  hipEvent_t ev_aux = nullptr;                  // cross-stream ordering inside a run
  ncclComm_t comm = nullptr;
  // peer-halo transport: [0] exchanges completed by the south neighbour (its pushes into my bottom halo rows),
  // [1] by the north neighbour, [2] error bits of halo_wait, [3] ticket of halo_push
  uint32_t *halo_flags = nullptr;
  PeerLink south, north;
  lbm::HaloPeer *d_peer = nullptr;   // device copy of what the fused push / wait of d2q9_multi needs (filled when the ring is connected)
  // staged launch sets (RCCL transport + the deep window kernels, staged_sets): the edge units of the ONE launch store their rows a second
  // time into these two blocks of halo-depth rows (write-through, as into a peer's halo rows) and the last edge wave raises halo_flags[4]
  // and [5]; the edge stream waits on those words and sends from the blocks
  float *stage[2] = {nullptr, nullptr};
  lbm::HaloPeer *d_stage_peer = nullptr;
  int can_wait_value = 0;            // hipDeviceAttributeCanUseStreamWaitValue
  // d2q9_resident (one slab, the grid resident in registers over all steps of a launch): bands of res_bh rows x res_w waves
  int res_bh = 0, res_w = 0, res_bands = 0;   // 0 = the grid has no such decomposition
  unsigned *res_words = nullptr;              // [bands][32]: a band's step words, one 128-byte line
  float *res_xrows = nullptr;                 // [2][bands][2][3][nx] exchange rows
  unsigned *res_err = nullptr;
  unsigned res_seq = 0;                       // what the step words have reached
  double *av_tmp = nullptr;   // all-reduce target of the velocity record (rank mode), allocated on first use
  // output-stage scratch
  float *fin_partials = nullptr;
  int fin_blocks = 0;
  float *own(int buf) const { return cells[buf] + (size_t)row0 * row_stride; }
  const uint8_t *mask_own(int nx) const { return mask + (size_t)row0 * nx; }
};

}  // namespace

struct lbm_ctx {
  lbm_params p{};
  std::vector<Slab> slabs;  // local slabs
  int nslabs_global = 1;    // slabs in the ring (== slabs.size() unless rank mode)
  bool rank_mode = false;
  bool halo_mode = false;   // slabs carry halo rows and exchange them (more than one slab, or forced for tests)
  int rows_min = 0;         // smallest slab of the partition (ny / nslabs): every rank takes the size-dependent
                            // decisions (kernel, halo depth, schedule) from it, so that all ranks take the same ones
  int halo_depth = 3;       // rows exchanged per side and launch set: 4 / 3 (four- / three-step kernel; 2 for very thin slabs) or 8
                            // (LDS multi-step kernel, small slabs)
  int rank = 0;
  int cur = 0;              // index of the grid holding the current state
  int steps_done = 0;
  int ring_fill = 0;        // buffered steps not yet reduced
  int ring = kRingMax;      // slots in the partial-sum ring (smaller for grids with many workgroups)
  int transport_eff = TRANSPORT_COPY;
  uint32_t halo_seq = 0;    // exchanges issued so far (PEER transport: the value the neighbours' flags reach)
  int halo_sync = 0;        // PEER transport, consumer side: 0 = halo_wait kernel (bounded spin), 1 = hipStreamWaitValue32
  unsigned long long halo_timeout_ms = kHaloWaitMsDefault;  // bound of that spin (option "halo_timeout_ms")
  int compact = -1;         // PEER transport + d2q9_multi: one launch per launch set on ONE stream, the edge tiles push the
                            // halo rows themselves (-1 auto = on, 0 off = edge stream / interior stream / push kernel)
  int push_release = -1;    // PEER transport, producer side (release_pushed in d2q9_kernels.h): 1 = release fences around the ticket,
                            // 0 = drained write-through stores, -1 = auto: fences whenever a ring neighbour lives on another device
  int stale_exchange = 0;   // TEST HOOK (option "debug_stale_exchange"): the n-th halo exchange from now announces itself (flags, events)
                            // but delivers no rows — what a transport that loses or misplaces halo rows looks like to the checks
                            // above the library (bench.py transport_check); 0 = off, cleared when it fires
  bool failed = false;      // a run ended in an error after launches had begun: only lbm_destroy is valid
  // lbm_run_profiled: timing events of the first local slab, kProfEvents per launch set, for up to kProfSets sets
  std::vector<hipEvent_t> prof_ev;
  int prof_sets = -1;       // -1: not profiling; else launch sets recorded so far in this run
  // options
  int variant = 0;
  int grid_blocks = 0;
  int nt_stores = -1;
  int nt_loads = -1;        // non-temporal source loads in d2q9_step2: -1 auto (with nt stores), 0 off, 1 on
  int fuse = -1;            // two timesteps per launch (d2q9_step2): -1 auto, 0 off, 1 on
  int tile_shape = -1;      // d2q9_multi tile: -1 auto, 0 = 32x16, 1 = 16x16, 2 = 16x8
  int windows = -1;         // d2q9_step3 register windows: 0 = in registers (1 wave/SIMD), 1 = in LDS (2 waves/SIMD), -1 auto
  int load_bufs = 0;        // d2q9_step3 row-sets of loads in flight: 1, 2, 0 = auto
  int sched_waves = 0;      // waves per SIMD the d2q9_step3 schedule plans for: 1, 2, 0 = auto
  int pair = -1;            // d2q9_step3p (chunk pairs share their start-up rows): 1 on, 0 off, -1 auto
  int twin_steps = 0;       // d2q9_deep_twin: most timesteps per launch (2..8), 0 = auto (5)
  int edge_aware = -1;      // d2q9_deep with row slabs: one-round interior schedule whose last units take over the edge launch's slots (-1/1 on, 0 off)
  int steady = -1;          // d2q9_deep: 1 (and -1, auto) = launches of 6, 7 or 8 timesteps run the kernel instantiated for that depth (steady
                            // form of the row loop), 0 = the any-depth kernel always
  int obst_paths = -1;      // d2q9_deep: 1 (and -1, auto) = a second collision path without bounce-back selects for waves without blocked cells
  int balance = -1;         // d2q9_deep, one-round schedules: 1 (and -1, auto) = strips that hold blocked cells in most rows get twice the waves
                            // (balance_heavy_strips), 0 = every strip the same chunks
  int free_sweeps = -1;     // d2q9_deep at the depths with a kernel of their own: 1 (and -1, auto) = a wave whose rows hold no blocked cell in its
                            // strip runs the sweep without obstacle handling (the map of Slab::clean_bits), 0 = every wave looks level by level
  int multistep = -1;       // T timesteps per launch on LDS tiles (d2q9_multi, small grids): -1 auto, 0 off, 1..8 = T
  int resident = -1;        // d2q9_resident (all steps of a launch with the grid in registers): -1 auto, 0 off, 1 on where the grid allows
  int chunk_rows = 0;       // longest chunk (rows per work unit) of d2q9_step2 (0 = auto)
  int chunk_min = 0;        // shortest chunk at the tapered end of a band (0 = auto)
  bool vec4 = true;
  double *av_host = nullptr;  // staging for downloads
};

namespace {

using namespace lbm;

int div_up(long a, long b) { return (int)((a + b - 1) / b); }

int set_dev(const Slab &s) {
  HIP_TRY(hipSetDevice(s.dev));
  return LBM_OK;
}

template <typename T>
int dev_alloc(T **p, size_t count) {
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
  return LBM_OK;
}

// launch geometry of the single-step kernel for `work_rows` rows
int step_blocks(const lbm_ctx *c, int work_rows) {
  const int vec = c->vec4 ? 4 : 1;
  const long threads = (long)(c->p.nx / vec) * work_rows;
  // default: one 256-thread tile per workgroup, no grid-stride.  Measured on 8192x8192: 65536 one-tile
  // workgroups 0.795 ms/step vs 0.85 ms with 2048 persistent ones (the dispatcher walks the grid in
  // address order and keeps every CU's queue full; profiles/r01_tune_layout_grid.txt)
  int nb = div_up(threads, kBlock);
  if (c->grid_blocks > 0) nb = std::min(nb, c->grid_blocks);
  return std::max(1, nb);
}

void split_rows(int ny, int P, int idx, int *y0, int *rows) {
  const int base = ny / P, rem = ny % P;
  *y0 = idx * base + std::min(idx, rem);
  *rows = base + (idx < rem ? 1 : 0);
}

// two-steps-per-launch kernel: float4 rows of at least one wave, a few rows per slab
bool windows_in_lds(const lbm_ctx *c) { return c->windows < 0 ? kStep3LdsDefault : c->windows != 0; }
int step3_load_bufs(const lbm_ctx *c) { return c->load_bufs > 0 ? c->load_bufs : (windows_in_lds(c) ? 1 : 2); }

bool step3_can_pair(const lbm_ctx *c) { return windows_in_lds(c) && step3_load_bufs(c) == 1 && c->pair != 0; }

// waves per SIMD the d2q9_step3 schedule plans for: two with the windows in LDS.  (The unpaired kernel did better with
// one round of 1024 longer units on grids of up to 800K cells — 1024x512 81 / 93 GLUPS planned for 2 / 1, 768x768 90 /
// 102; with chunk pairs, d2q9_step3p, two waves per SIMD win everywhere: 98 and 107, tools/ab.py.)
int step3_sched_waves(const lbm_ctx *c) {
  if (c->sched_waves > 0) return c->sched_waves;
  if (!windows_in_lds(c)) return 1;
  if (step3_can_pair(c)) return 2;
  return (long)c->p.nx * c->rows_min <= 800L * 1024 ? 1 : 2;
}
int step4_sched_waves(const lbm_ctx *c) { return c->sched_waves > 0 ? c->sched_waves : 2; }

bool fuse_possible(const lbm_ctx *c) {
  if (!c->vec4 || c->p.nx < 256) return false;
  return c->rows_min >= 8;
}
// d2q9_deep: rows for its 2*(kDeepSteps-1) start-up iterations; with halo rows, at least kDeepMin of them (a launch set
// advances at most as many steps as the halos are deep)
bool deep_possible(const lbm_ctx *c) {
  return fuse_possible(c) && c->rows_min >= 4 * kDeepSteps && (!c->halo_mode || c->halo_depth >= kDeepMin);
}
// d2q9_deep as chunk pairs (d2q9_deep_twin): one slab without halo rows whose pair schedule is one round of units.
// Same-box A/B (tools/ab.py, GLUPS, best other kernel / twins at five steps per launch): 768x512 88 / 89, 1024x512 97 / 96,
// 768x768 107 / 122, 1024x768 127 / 138, 1024x1024 140 / 149-152, 1536x1024 168 / 190, 2048x1024 187 / 218, 2048x2048 219-226 /
// 238-240, 3072x2048 250 / 264; against the lone kernel at eight steps: 8192x1024 265 / 261, 4096x4096 304 / 291, 8192x2048
// 305 / 290 -> twins below 8M cells.  (Twins at three / four / five steps per launch: 1024x1024 130 / 146 / 149.)
// Round 3, with the steady forms of both kernels (profiles/r03_twin_policy.txt; GLUPS lone at 8 / twins at 5 / twins at 8
// steps per launch): 1024x1024 - / 177 / 133, 1536x1024 - / 206 / 170, 2048x1024 - / 227 / 194, 2048x2048 206 (four-step) /
// 251 / 289, 3072x2048 292 / 274 / 320, 4096x2048 311 / 268 / 331, 3072x3072 319 / 257 / 337, 8192x1024 309 / 269 / 323,
// 4096x4096 356 / 266 / 373, 6144x4096 369 / 265 / 364, 8192x4096 383 / 296 / 405, 8192x6144 410 / 304 / 415, 8192x8192
// 422 / 279 / 431, 12288x8192 435 / 323 / 440 -> one slab without halo rows always runs chunk pairs: at up to five steps
// per launch below 3M cells (the D = 5 instantiation: 60-lane strips, no register windows), at up to eight from 3M cells
// (multi-round schedules included: the steady form made a pair's iterations cheap enough that holding the LDS of both
// chunks no longer costs more than the halved start-up saves).
constexpr long kTwinDeepCells = 3L << 20;
// Row slabs run the deep window kernel (8 halo rows, launch sets of up to eight steps) from this many cells per slab, the
// four-step kernel (4 halo rows) below.  Round 3, ring of one over peer stores, GLUPS four-step / deep (profiles/
// r03_scaling_projection.txt): 2048x1024 and 4096x512 (2M cells) 176 / 161 and 172 / 167; 2048x2048, 4096x1024, 8192x512
// (4M) 209 / 244, 210 / 245, 201 / 238 -> from 3M cells (round 2, before the per-depth kernels: 5M).
constexpr long kSlabDeepCells = 3L << 20;
// Row slabs between these sizes carry FIVE halo rows and, in compact launch sets (peer stores), run the chunk pairs of the
// mid-size grids — d2q9_deep_twin<5, ..., PUSH>, five steps per launch set — instead of the LDS tiles (up to 540K cells) and
// the three- / four-step kernels (up to 3M): one GPU ran the 1024x1024 input at 5.7 us/step with the pairs while a 1024x512
// slab of it took 7.3 (two GPUs slower than one).  Launch sets on two streams (RCCL, copies) keep those kernels,
// with the halo rows they always had up to 540K cells (eight: the LDS tiles, eight steps per exchange — the pairs then use the five
// nearest of the eight stored rows) and with five above (the three- / four-step kernels exchange five rows per set instead of 3 / 4).
// Ring of one over peer stores, us/step, previous choice -> pairs (profiles/r04_slab_twin5.txt): 1024x512 7.34 -> 4.87, 2048x512
// 9.03 -> 6.18, 1024x1024 8.75 -> 6.16, 2048x1024 11.89 -> 9.72, 4096x512 12.18 -> 9.62, 8192x256 12.08 -> 9.92, 2048x1400 14.57 ->
// 12.37; NOT 1024x256 (262K cells): LDS tiles 4.10, pairs 5.00 -> from 300K cells, as on one slab without halo rows.
constexpr long kSlabTwinCells = 300L * 1024;
constexpr int kOneRoundRows = 160;   // d2q9_deep: longest chunk of a one-round schedule (see deep_geometry)
constexpr int kMaxHeavyStrips = 4;  // see build_clean_bits / balance_heavy_strips
bool compact_transport(const lbm_ctx *c) { return c->halo_mode && c->transport_eff == TRANSPORT_PEER && c->compact != 0; }
// RCCL contexts whose launch sets can be STAGED (see staged_sets): the edge stream must be able to wait on a flag word
bool stageable_transport(const lbm_ctx *c) {
  if (!c->halo_mode || c->transport_eff != TRANSPORT_RCCL || c->compact == 0 || c->slabs.empty()) return false;
  for (const Slab &s : c->slabs)
    if (!s.can_wait_value) return false;
  return true;
}
// the slab form of the five-step chunk pairs (see kSlabTwinCells): five halo rows, compact launch sets, kernel choice left to the library
bool slab_twin5(const lbm_ctx *c) {
  if ((long)c->p.nx * c->rows_min <= kSlabTwinCells || (long)c->p.nx * c->rows_min >= kSlabDeepCells) return false;
  // (over RCCL — staged launch sets — only where the five rows exchanged are the rows stored, i.e. above the LDS tiles' range: those
  // slabs ran the three- / four-step kernels on two streams: 2048x1024 17.25 -> 13.08 us/step, 4096x512 18.09 -> 13.10, profiles/r04_staged_rccl.txt)
  const bool transport_ok = compact_transport(c) || (stageable_transport(c) && c->halo_depth == kDeepTwinDefault);
  return transport_ok && (c->halo_depth == kDeepTwinDefault || c->halo_depth == kMultiMaxT) && fuse_possible(c) && c->rows_min >= 4 * kDeepTwinDefault &&
         c->pair != 0 && c->multistep <= 0 && (c->fuse < 0 || c->fuse == kDeepTwinDefault) && c->twin_steps <= 0;
}
int twin_cap(const lbm_ctx *c) {
  if (slab_twin5(c)) return kDeepTwinDefault;
  if (c->twin_steps > 0) return c->twin_steps;
  return (long)c->p.nx * c->rows_min >= kTwinDeepCells ? kDeepTwinSteps : kDeepTwinDefault;
}
bool deep_twin_effective(const lbm_ctx *c) {
  if (c->halo_mode || c->slabs.empty() || !c->slabs[0].f6_twin.paired) return false;
  return c->pair != 0;
}
// 0 = one launch per step, 2 = d2q9_step2, 3 = d2q9_step3 (falls back to 2 for the last steps of a run and
// where the halo rows are fewer than 3)
int fuse_level(const lbm_ctx *c) {
  if (!fuse_possible(c)) return 0;
  if (slab_twin5(c)) return kDeepTwinDefault;
  int lvl;
  if (c->fuse >= 0) {
    lvl = c->fuse == 0 ? 0 : (c->fuse >= 3 ? c->fuse : 2);
  } else {
    // auto (same-box A/B, tools/ab.py + tools/ab.py, GLUPS two-step / three-step): the smallest grids go
    // to the LDS tile kernel (multistep_effective); 768x512 89 / 86 -> two steps per launch; 1024x512 82 / 93,
    // 768x768 89 / 102, 1024x768 106 / 112, 1024x1024 114 / 119, 1536x1024 127 / 147 -> three steps per launch
    // ... and four steps per launch from 1.25M cells up (tools/ab.py, three / four steps, chunk pairs where the
    // launch is one round: 1024x768 129 / 128, 1024x1024 143 / 136, 1536x1024 162 / 175, 2048x1024 170 / 192,
    // 2048x2048 183 / 223, 4096x4096 217 / 273, 8192x8192 230 / 295)
    const long cells = (long)c->p.nx * c->rows_min;
    lvl = cells >= 1280L * 1024 ? 4 : (cells > 450L * 1024 ? 3 : 2);
    // ... and d2q9_deep (first measured at six steps per launch, tools/ab.py, four / six steps: 4096x4096
    // 272 / 272, 8192x1024 244 / 254, 8192x2048 278 / 285, 8192x4096 292 / 317, 6144x6144 276 / 327, 8192x8192 301 / 354,
    // 16384x8192 299 / 362 GLUPS; below: 2048x2048 222 / 227, 1024x1024 137 / 125)
    // (one launch by the steps it advances, 8192x8192, tools/depth_sweep.py: 2..4 steps 940-965 us — the pass over the grid,
    // as long as the four-step kernel's launch —, 5: 1007, 6: 1113, 7: 1304, 8: 1455 us = 369 GLUPS)
    // (with the band count of a one-round schedule chosen freely, r02: 2048x2048 220 / 222, 3072x2048 232 / 252, 4096x2048
    // 251 / 268, 3072x3072 242 / 272, 4096x4096 270 / 323, 8192x1024 256 / 277, 8192x2048 275 / 317, 8192x8192 293 / 367)
    // ... and, as chunk pairs, from 300K cells — right above the LDS tile kernel's range (see deep_twin_effective; round 3,
    // GLUPS two- / three-step kernel against the pairs with their steady form at five steps per launch,
    // profiles/r03_twin_policy.txt: 512x512 67.6 (LDS tiles) / 66.7, 640x512 75.6 / 83.4, 768x512 87.6 / 94.2, 768x640 93.1 /
    // 109.1, 1024x512 97.4 / 101.9; round 2's threshold was 560K cells)
    if (deep_possible(c) && (cells >= kSlabDeepCells || (cells > 300L * 1024 && deep_twin_effective(c)))) lvl = kDeepSteps;
  }
  if (lvl > 4 && !(lvl >= kDeepMin && lvl <= kDeepSteps && deep_possible(c))) lvl = 4;
  if (lvl >= kDeepMin && c->halo_mode) lvl = std::min(lvl, c->halo_depth);
  if (lvl == 4 && ((c->halo_mode && c->halo_depth < 4) || !windows_in_lds(c))) lvl = 3;  // needs 4 halo rows, LDS windows
  if (lvl == 3 && c->halo_mode && c->halo_depth < 3) lvl = 2;
  return lvl;
}
bool fuse_effective(const lbm_ctx *c) { return fuse_level(c) != 0; }

// d2q9_resident: one slab without halo rows whose grid decomposes into bands of BH full-width rows x W = nx/128 <= 8 waves that are
// all resident at once (two waves per SIMD: 8 / W workgroups per CU); res_* are set by resident_geometry.
// Auto: from 200K cells (same-box A/B, us/step, the library's other choice / resident, profiles/r04_resident.txt: 256x256 1.79 / 2.13,
// 128x2048 4.25 / 2.5, 512x512 3.87 / 2.54, 1024x512 5.17 / 3.08, 512x2048 6.41 / 3.74, 1024x1024 5.80 / 3.81); the kernel's limit is
// what fits the registers of the chip at two waves per SIMD: 1.5M cells on 256 CUs (bands of six rows).
bool resident_effective(const lbm_ctx *c) {
  if (c->halo_mode || c->slabs.size() != 1 || c->slabs[0].res_bands <= 0) return false;
  if (c->resident >= 0) return c->resident > 0;
  return c->fuse < 0 && c->multistep < 0 && (long)c->p.nx * c->p.ny >= 200L * 1024;
}
const void *resident_kernel(int bh) {
  return bh == 2 ? (const void *)d2q9_resident<2> : bh == 4 ? (const void *)d2q9_resident<4> : (const void *)d2q9_resident<6>;
}
int resident_geometry(const lbm_ctx *c, Slab &s) {
  s.res_bh = s.res_w = s.res_bands = 0;
  // (nx: a multiple of 4 — the mask is read a dword at a time — up to 1024; a band's last wave may be partly filled)
  if (c->halo_mode || c->resident == 0 || c->p.nx % 4 != 0 || c->p.nx < 128 || c->p.nx > 1024) return LBM_OK;
  const int W = div_up(c->p.nx, 128);
  if (set_dev(s)) return LBM_ERR_HIP;
  int bh = 0;
  for (int cand : {2, 4, 6}) {  // the shortest bands that still fit the chip at once: most waves, least arithmetic per hand-shake
                                // (6 rows: 108 registers of state, 239 in all; bands of 8 rows — 144 — spill: 1.5M cells on 256 CUs is
                                // the limit of this kernel)
    if (c->p.ny % cand != 0) continue;
    // every band must be resident at once (they wait for each other): what the runtime says a CU holds of this instantiation, capped
    // at the two waves per SIMD the schedules of this library plan with
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, resident_kernel(cand), 64 * W, 0));
    const int capacity = s.cus * std::min(per_cu, 8 / W);
    if (c->p.ny / cand <= capacity) { bh = cand; break; }
  }
  if (bh == 0) return LBM_OK;
  s.res_bh = bh;
  s.res_w = W;
  s.res_bands = c->p.ny / bh;
  if (set_dev(s)) return LBM_ERR_HIP;
  if (s.res_words) HIP_TRY(hipFree(s.res_words));
  if (s.res_xrows) HIP_TRY(hipFree(s.res_xrows));
  s.res_words = nullptr;
  s.res_xrows = nullptr;
  if (dev_alloc(&s.res_words, (size_t)s.res_bands * 32)) return LBM_ERR_HIP;
  HIP_TRY(hipMemset(s.res_words, 0, (size_t)s.res_bands * 32 * sizeof(unsigned)));
  if (dev_alloc(&s.res_xrows, (size_t)2 * s.res_bands * 2 * 3 * c->p.nx)) return LBM_ERR_HIP;
  if (!s.res_err) {
    if (dev_alloc(&s.res_err, 1)) return LBM_ERR_HIP;
    HIP_TRY(hipMemset(s.res_err, 0, sizeof(unsigned)));
  }
  s.res_seq = 0;
  s.nb_total = std::max(s.nb_total, s.res_bands * W);
  return LBM_OK;
}

// LDS multi-step kernel: one slab holding the whole periodic grid; worth it only while the grid is launch-bound
int multistep_effective(const lbm_ctx *c) {
  // launch grids stay small (tile count from the smallest tile and the largest slab: identical on every rank)
  if ((long)div_up(c->p.nx, 16) * div_up(c->rows_min + 1, 8) > 65536) return 0;
  if (slab_twin5(c)) return 0;
  // with halo rows a launch can advance at most as many steps as the halos are deep
  const int cap = c->halo_mode ? std::min(kMultiMaxT, c->halo_depth) : kMultiMaxT;
  // big slabs: the register/LDS-window kernels
  if (c->halo_mode && c->halo_depth < kMultiMaxT && c->multistep < 0) return 0;
  if (c->multistep >= 0) return std::min(c->multistep, cap);
  // auto (profiles/r01_kernel_choice.txt, us/step LDS tiles / two-step kernel): 128x128 1.4 / 5, 384x384 3.4 / 4.2,
  // 512x512 3.9 / 4.4, 768x512 5.6 / 4.5, 1024x512 7.3 / 6.4 -> up to 300K cells on one slab.  Slabs that exchange
  // halos keep it up to 1024x512 cells: 8 steps per exchange instead of 2 (1024x512 ring of one: 9 against 25 us/step)
  const long limit = c->halo_mode ? 540L * 1024 : 300L * 1024;
  return ((long)c->p.nx * c->rows_min <= limit) ? cap : 0;
}

// Compact launch sets (peer transport): ONE launch per launch set on one stream — its first workgroups are the edge
// tiles / edge chunks, which store the halo rows into the ring neighbours themselves and raise their flag words — instead
// of edge launch + push kernel on an edge stream beside the interior launch.  For the LDS-tile kernel and for the
// three- / four-step kernels in their default form (LDS windows, one row-set of loads in flight, plain loads).
// Staged launch sets: the compact launch form under the RCCL transport, for slabs that run the deep window kernel (8 halo rows) or its
// five-step chunk pairs (5 halo rows: slab_twin5).  ONE launch per
// set on the main stream — edge units first, chunk pairs, balanced: the kernel and schedule of the peer transport — whose edge units
// push their rows into a local staging block instead of a neighbour; the edge stream waits for the flag word the last edge wave raises
// (hipStreamWaitValue32), sends the blocks and receives into the halo rows while the interior is still running, and the next launch waits
// for that exchange's event.  Two-stream sets (edge launch + interior launch of the lone kernel) ran the 8192x1024 slab at 335 GLUPS per
// rank against 404 over peer stores; option "compact" 0 brings them back.
bool staged_sets(const lbm_ctx *c) {
  if (!stageable_transport(c)) return false;
  return multistep_effective(c) == 0 && (fuse_level(c) >= kDeepMin || slab_twin5(c));
}
bool compact_sets(const lbm_ctx *c) {
  if (staged_sets(c)) return true;
  if (!compact_transport(c)) return false;
  if (slab_twin5(c)) return true;    // d2q9_deep_twin<5, .., PUSH>
  if (multistep_effective(c) > 0) return true;
  const int lvl = fuse_level(c);
  if (lvl >= kDeepMin) return true;  // d2q9_deep<.., PUSH>
  return (lvl == 3 || lvl == 4) && windows_in_lds(c) && step3_load_bufs(c) == 1;
}

// Work decomposition of d2q9_step2 over stored rows [r0, r1): strips x chunks.  A unit's cost is
// proportional to its rows + 2 and all units of a launch finish at about the same time, so equal chunks
// leave the chip partly idle during the last round of units (17 % of the launch with 32-row chunks on
// 8192x8192).  The schedule therefore tapers: every band (the share of one XCD) starts with chunks of
// `cmax` rows and ends with ever shorter ones (guided self-scheduling), down to `cmin`.  (R full rounds of equal
// chunks instead of the taper: within +-2 % on 8192x1024 ... 8192x8192, no consistent sign — not adopted.)
int fuse_schedule(const Slab &s, int r0, int r1, int cmax, int cmin, bool allow_bands, FuseGeom &g, int waves_per_simd = 2,
                  int reserve = 0, bool pairs = false, int strips_of_kernel = 0, int cmax_one = 0) {
  if (cmax_one < cmax) cmax_one = cmax;  // longest chunk of a ONE-round schedule (d2q9_deep: longer than the tapered schedules' first chunks)
  const int rows = r1 - r0;
  const int strips = strips_of_kernel > 0 ? strips_of_kernel : s.strips;
  g.nbands = (allow_bands && rows >= 8 * 4 * cmin) ? 8 : 1;
  // CUs x SIMDs x waves per SIMD the kernel's registers / LDS allow, minus the wave slots a concurrent launch needs
  // (slab mode: the edge launch, which must find its slots at once — see slab_geometry)
  const int waves_resident = std::max(s.cus, s.cus * 4 * waves_per_simd - reserve);
  // The chunk-pair kernels (d2q9_step3p / d2q9_step4p) hold the LDS of BOTH chunks of a pair until the longer one is
  // done, so a one-round schedule needs an EVEN number of chunks per band that still fits the wave slots: 7.3 slots
  // per band means 6 chunks with 8 bands (82 % of the slots) but 14 with 4 bands (96 %) — take the band count that
  // keeps most waves busy, preferring more bands (neighbouring strips then share an XCD's L2).
  g.single_round = false;
  // (the same choice for d2q9_deep, unpaired: 4096x4096 has 6.9 slots per band and strip with 8 bands — 6 chunks, 87 % of
  // the slots, 85 rows + 14 start-up iterations each — but 55 per strip with one band: 75 rows + 14)
  const bool flex_bands = strips_of_kernel > 0;
  if ((pairs || flex_bands) && g.nbands == 8) {
    int best_nb = 8;
    double best = -1.0;
    for (int nb = 8; nb >= 1; nb /= 2) {
      const int fl = pairs ? std::max(2, (int)std::floor((double)waves_resident / nb / strips) & ~1)
                           : std::max(1, (int)std::floor((double)waves_resident / nb / strips));
      const int nrows = div_up(rows, nb);
      if ((int)std::ceil((double)nrows / fl) > cmax_one) continue;  // not a one-round schedule with this band count
      const double busy = (double)std::min(fl, nrows) * nb * (1.0 + 0.01 * nb);
      if (busy > best) { best = busy; best_nb = nb; }
    }
    g.nbands = best_nb;
  }
  double slots = std::max(1.0, (double)waves_resident / g.nbands / strips);  // concurrent chunks per band
  if (pairs) slots = std::max(2.0, (double)((int)std::floor(slots) & ~1));
  std::vector<int> starts;
  int chunks_per_band = 0;
  for (int b = 0; b < g.nbands; b++) {
    int y0, n;
    split_rows(rows, g.nbands, b, &y0, &n);
    std::vector<int> sizes;
    int rem = n;
    // a grid small enough to be done in ONE round of units (all of them resident at once) gets equal chunks
    // that just fill the wave slots: every extra unit costs two redundant rows, and a second, partly filled
    // round costs more than it balances (1024x1024: 3-row chunks = 1720 units: 10.2 us/step; 2-row chunks =
    // 2560 units: 12.1; 4-row chunks = 1280 units: 11.6 — tools/ab.py)
    const int one_round = (int)std::ceil(n / std::max(1.0, std::floor(slots)));
    // (one round only if the units really are resident at once: a pair schedule needs two slots per band and strip —
    // with fewer, "one round" of 64-row chunks was 2192 units on 1500 free slots and the last workgroups started when the
    // first had finished: compact 8192x1024 slab 212 instead of 220 GLUPS)
    const bool fits = (double)waves_resident / g.nbands / strips >= (pairs ? 2.0 : 1.0);
    const bool single_round = one_round <= cmax_one && fits;
    if (b == 0) g.single_round = single_round;
    while (rem > 0) {
      int sz = single_round ? std::max(2, one_round) : (int)std::ceil(rem / (2.0 * slots));
      if (!single_round) sz = std::max(cmin, std::min(cmax, sz));
      sz = std::min(sz, rem);
      sizes.push_back(sz);
      rem -= sz;
    }
    // all bands need the same number of chunks (unit arithmetic in the kernel): band 0 is never
    // shorter than the others (split_rows); pad with empty chunks / merge surplus into the last one
    // (an even count for the chunk-pair kernels, which pair chunks 2p and 2p+1)
    if (b == 0) chunks_per_band = pairs ? ((int)sizes.size() + 1) / 2 * 2 : (int)sizes.size();
    while ((int)sizes.size() < chunks_per_band) sizes.push_back(0);
    int extra = 0;
    while ((int)sizes.size() > chunks_per_band) { extra += sizes.back(); sizes.pop_back(); }
    if (extra) sizes.back() += extra;
    int y = r0 + y0;
    for (int sz : sizes) { starts.push_back(y); y += sz; }
  }
  starts.push_back(r1);
  g.nchunks = chunks_per_band * g.nbands;
  g.units_per_band = chunks_per_band * strips;
  g.units = g.units_per_band * g.nbands;
  g.starts = starts;
  if (set_dev(s)) return LBM_ERR_HIP;
  if (g.chunk_start) HIP_TRY(hipFree(g.chunk_start));
  g.chunk_start = nullptr;
  if (dev_alloc(&g.chunk_start, starts.size())) return LBM_ERR_HIP;
  HIP_TRY(hipMemcpy(g.chunk_start, starts.data(), starts.size() * sizeof(int), hipMemcpyHostToDevice));
  return LBM_OK;
}

// Schedule for a kernel that has a chunk-pair form.  Pairs pay off where chunks are short, i.e. in one-round schedules
// (tools/ab.py, unpaired / paired GLUPS: 1024x1024 117 / 143, 2048x1024 165 / 192, 2048x2048 211 / 223,
// 4096x4096 268 / 273); with the long chunks of multi-round schedules the two waves of a workgroup only hold each
// other's LDS (8192x8192 295 / 289).  c->pair: -1 = that rule, 1 = always, 0 = never.
int fuse_schedule_pairs(const lbm_ctx *c, const Slab &s, int r0, int r1, int cmax, int cmin, FuseGeom &g, int waves_per_simd,
                        int reserve, bool kernel_can_pair) {
  g.paired = false;
  // (not with row slabs: next to the edge launch's 20-KB workgroups and the RCCL kernel the 40-KB pairs of the interior
  // launch no longer all fit at once — 8192x1024 ring of one: 168 GLUPS paired, 232 unpaired, tools/ab.py)
  if (kernel_can_pair && c->pair != 0 && !(c->halo_mode && c->pair < 0 && !compact_sets(c))) {
    if (int rc = fuse_schedule(s, r0, r1, cmax, cmin, true, g, waves_per_simd, 2 * reserve, true)) return rc;
    if (g.single_round || c->pair > 0) {
      g.paired = true;
      return LBM_OK;
    }
  }
  return fuse_schedule(s, r0, r1, cmax, cmin, true, g, waves_per_simd, reserve, false);
}

// Which stored rows hold a blocked cell within which strip of d2q9_deep (Step2Args::clean_bits): rebuilt whenever the strips
// or the obstacle map change.  One pass over the byte mask on the device.
int build_clean_bits(const lbm_ctx *c, Slab &s) {
  if (s.strips2 <= 0 || !s.mask) return LBM_OK;
  if (set_dev(s)) return LBM_ERR_HIP;
  const int words = div_up(s.ext_rows, 64);
  if (s.clean_bits) HIP_TRY(hipFree(s.clean_bits));
  s.clean_bits = nullptr;
  s.clean_words = words;
  if (dev_alloc(&s.clean_bits, (size_t)s.strips2 * words)) return LBM_ERR_HIP;
  hipLaunchKernelGGL(lbm::strip_row_bits, dim3(s.strips2 * words), dim3(64), 0, s.s_main, s.mask, c->p.nx, s.ext_rows, s.strips2, s.lanes2,
                     lbm::deep_halo_lanes(kDeepSteps), s.clean_bits, words);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s.s_main));
  std::vector<unsigned long long> host((size_t)s.strips2 * words);
  HIP_TRY(hipMemcpy(host.data(), s.clean_bits, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long any = 0;
  for (unsigned long long w : host) any |= w;
  s.all_clean = any == 0;
  // strips whose waves look at blocked cells in most of their rows run ~1.3 x as long per row as the others: candidates for
  // balance_heavy_strips (a handful at most — a grid full of obstacles has nothing to balance against)
  s.heavy.clear();
  for (int st = 0; st < s.strips2; st++) {
    long dirty = 0;
    for (int w = 0; w < words; w++) dirty += __builtin_popcountll(host[(size_t)st * words + w]);
    if (2 * dirty > s.ext_rows) s.heavy.push_back(st);
  }
  if ((int)s.heavy.size() > kMaxHeavyStrips) s.heavy.clear();
  return LBM_OK;
}

void free_balance(FuseGeom &g) {
  if (g.vmap) hipFree(g.vmap);
  if (g.vtab) hipFree(g.vtab);
  g.vmap = g.vtab = nullptr;
  g.vstrips = 0;
}

// One-round schedule g of d2q9_deep (pairs: d2q9_deep_twin) over `strips` real strips: every heavy strip gets a second
// (virtual) strip, and its two copies work on the two halves of every chunk — of every chunk PAIR for the twins, whose
// chunks 2p / 2p+1 must stay neighbours: copy j takes chunk 2p+j of the plain table and splits it in the middle.  The
// caller has planned the schedule's wave slots for strips + heavy.size() strips (g.units_per_band / g.units say so).
int balance_heavy_strips(const Slab &s, FuseGeom &g, bool pairs, int strips, int nh) {
  free_balance(g);
  const int n = g.nchunks;
  if (nh == 0) return LBM_OK;  // (option balance = 0, or nothing to balance: the schedule was planned for `strips` strips)
  if (nh != (int)s.heavy.size() || (int)g.starts.size() != n + 1 || (pairs && (n & 1)))
    return fail(LBM_ERR_STATE, "internal: schedule planned for %d heavy strips cannot be balanced", nh);
  std::vector<int> vtab((size_t)3 * n * 2), vmap;
  for (int ch = 0; ch < n; ch++) {
    const int y0 = g.starts[ch], y1 = g.starts[ch + 1];
    vtab[2 * ch] = y0;
    vtab[2 * ch + 1] = y1;
    if (!pairs) {
      const int mid = y0 + (y1 - y0) / 2;
      vtab[2 * (n + ch)] = y0;      vtab[2 * (n + ch) + 1] = mid;
      vtab[2 * (2 * n + ch)] = mid; vtab[2 * (2 * n + ch) + 1] = y1;
    }
  }
  if (pairs)
    for (int p2 = 0; p2 < n; p2 += 2)
      for (int j = 0; j < 2; j++) {
        const int y0 = g.starts[p2 + j], y1 = g.starts[p2 + j + 1], mid = y0 + (y1 - y0) / 2;
        int *t = &vtab[2 * ((size_t)(1 + j) * n + p2)];
        t[0] = y0; t[1] = mid; t[2] = mid; t[3] = y1;
      }
  for (int st = 0; st < strips; st++) {
    const bool heavy = std::find(s.heavy.begin(), s.heavy.end(), st) != s.heavy.end();
    vmap.push_back(st);
    vmap.push_back(heavy ? 1 : 0);
    if (heavy) { vmap.push_back(st); vmap.push_back(2); }
  }
  if (set_dev(s)) return LBM_ERR_HIP;
  if (dev_alloc(&g.vmap, vmap.size()) || dev_alloc(&g.vtab, vtab.size())) return LBM_ERR_HIP;
  HIP_TRY(hipMemcpy(g.vmap, vmap.data(), vmap.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g.vtab, vtab.data(), vtab.size() * sizeof(int), hipMemcpyHostToDevice));
  g.vstrips = strips + nh;
  return LBM_OK;
}

// Free sweeps by default (same-box A/B on an MI355X, GLUPS with / without, profiles/r03_free_sweeps.txt): where a launch is
// several rounds of units — 8192x8192 cavity 463 / 446 — or nothing in the slab is blocked — no obstacles at all: 8192x8192
// 476 / 456, 4096x4096 410 / 397, the 8192x1024 slab of an 8-GPU run 344 / 327.  A ONE-round launch that has blocked cells ends
// with its slowest waves, the ones that look (the two wall strips of a cavity): the free waves' saving buys nothing there and
// the second set of loops costs instruction-cache room — 4096x4096 cavity 376 / 375, 8192x1024 slab 319 / 325, 8192x2048 361 / 368 —
// unless the schedule takes those strips out of the critical path (balance_heavy_strips, g.vstrips > 0): then the free waves
// are what the launch ends with, and the free sweeps pay again (8192x1024 slab with side walls 380-391 / 373,
// profiles/r03_balance.txt).
const unsigned long long *clean_bits_for(const lbm_ctx *c, const Slab &s, const FuseGeom &g) {
  if (c->free_sweeps == 0 || !s.clean_bits) return nullptr;
  if (c->free_sweeps < 0 && g.single_round && !s.all_clean && g.vstrips == 0) return nullptr;
  return s.clean_bits;
}

// Schedules of d2q9_deep for a slab: lanes of two cells, deep_halo_lanes() of them idle at either end of a strip; strips
// start on 64-byte boundaries (8 lanes).  2*(D-1) redundant start-up iterations per chunk -> long chunks.
int twin5_slab_geometry(const lbm_ctx *c, Slab &s);
int deep_geometry(const lbm_ctx *c, Slab &s) {
  s.f6_main.units = s.f6_edge.units = s.f6_twin.units = 0;
  s.f6_twin.paired = false;
  if (slab_twin5(c)) return twin5_slab_geometry(c, s);
  if (!deep_possible(c)) return LBM_OK;
  const int q2 = c->p.nx / 2, lmax = 64 - 2 * lbm::deep_halo_lanes(kDeepSteps);
  s.strips2 = div_up(q2, lmax / 8 * 8);
  s.lanes2 = std::min(lmax, (div_up(q2, s.strips2) + 7) / 8 * 8);
  const int c6max = std::max(8, std::min(c->chunk_rows > 0 ? c->chunk_rows : 96, s.rows));
  const int c6min = std::max(4, std::min(c->chunk_min > 0 ? c->chunk_min : 24, c6max));
  // A launch of ONE round of units may have longer chunks than the first chunks of the tapered multi-round schedule: a grid
  // that needs 1.2 or 1.5 rounds of 96-row chunks ends with a long, half-empty second round.  Same box, cavity, tapered / one
  // round (profiles/r03_balance.txt): 8192x3072 397 / 444 GLUPS (118-row chunks), 8192x4096 426 / 451 (158); from about two
  // rounds on nothing is left to gain (8192x5120 441 / 450, 8192x6144 453 / 462, 8192x8192 470 / 471) or lost (6144x6144, 181-row
  // chunks: 435 / 424) -> one round up to 160 rows per chunk.
  const int c6one = c->chunk_rows > 0 ? c6max : std::max(c6max, std::min(kOneRoundRows, s.rows));
  free_balance(s.f6_main);
  free_balance(s.f6_twin);
  if (int rc = build_clean_bits(c, s)) return rc;
  const int nh = c->balance != 0 ? (int)s.heavy.size() : 0;  // strips that get a second (virtual) strip in one-round schedules
  // a one-round schedule planned for strips2 + nh strips, balanced; any other schedule as it is
  auto one_slab_schedule = [&](FuseGeom &g, bool pairs) -> int {
    if (int rc = fuse_schedule(s, 0, s.rows, c6max, c6min, true, g, 2, 0, pairs, s.strips2, c6one)) return rc;
    if (nh == 0 || !g.single_round) return LBM_OK;
    if (int rc = fuse_schedule(s, 0, s.rows, c6max, c6min, true, g, 2, 0, pairs, s.strips2 + nh, c6one)) return rc;
    if (g.single_round) return balance_heavy_strips(s, g, pairs, s.strips2, nh);
    return fuse_schedule(s, 0, s.rows, c6max, c6min, true, g, 2, 0, pairs, s.strips2, c6one);
  };
  if (!c->halo_mode) {
    if (int rc = one_slab_schedule(s.f6_main, false)) return rc;
    s.nb_total = std::max(s.nb_total, s.f6_main.units);
    // chunk pairs (d2q9_deep_twin, at most kDeepTwinSteps per launch): where the launch is one round of units
    s.f6_twin.units = 0;
    s.f6_twin.paired = false;
    if (c->pair != 0) {
      // Twins of up to five steps per launch run the D = 5 instantiation: 2 halo lanes per side instead of 4, strips of up
      // to 60 lanes (starts on 32-byte boundaries: these grids live in the caches).  1024x1024: 9 strips instead of 10 (the
      // tenth held 8 useful lanes), 226 chunks of 4.5 rows instead of 172 of 6.
      const int tw_cap = twin_cap(c);
      if (tw_cap <= kDeepTwinDefault) {
        const int lmax5 = 64 - 2 * lbm::deep_halo_lanes(kDeepTwinDefault);
        s.strips_tw = div_up(q2, lmax5 / 4 * 4);
        s.lanes_tw = std::min(lmax5, (div_up(q2, s.strips_tw) + 3) / 4 * 4);
      } else {
        s.strips_tw = s.strips2;
        s.lanes_tw = s.lanes2;
      }
      if (tw_cap <= kDeepTwinDefault) {
        if (int rc = fuse_schedule(s, 0, s.rows, c6max, c6min, true, s.f6_twin, 2, 0, true, s.strips_tw)) return rc;
      } else if (int rc = one_slab_schedule(s.f6_twin, true)) {  // (the eight-step twins run d2q9_deep's strips)
        return rc;
      }
      s.f6_twin.paired = s.f6_twin.single_round || c->pair > 0 || (long)c->p.nx * c->rows_min >= kTwinDeepCells;
      if (s.f6_twin.paired) s.nb_total = std::max(s.nb_total, s.f6_twin.units);
    }
    return LBM_OK;
  }
  // slab mode: the edge launch computes the edge_rows rows at either end of the slab (one chunk each), the interior
  // launch the rest and leaves the edge launch its wave slots (see slab_geometry)
  FuseGeom &e = s.f6_edge;
  const int tab[4] = {s.row0, s.row0 + s.edge_rows, s.row0 + s.rows - s.edge_rows, s.row0 + s.rows};
  if (set_dev(s)) return LBM_ERR_HIP;
  if (e.chunk_start) HIP_TRY(hipFree(e.chunk_start));
  e.chunk_start = nullptr;
  if (dev_alloc(&e.chunk_start, 4)) return LBM_ERR_HIP;
  HIP_TRY(hipMemcpy(e.chunk_start, tab, sizeof(tab), hipMemcpyHostToDevice));
  e.nchunks = 3;
  e.skip = 1;
  e.nbands = 1;
  e.units_per_band = e.nchunks * s.strips2;
  e.units = e.units_per_band;
  const int i0 = tab[1], i1 = tab[2];
  if (i1 > i0) {
    // Edge-aware one-round schedule.  The edge launch goes first and holds 2*strips2 wave slots — but only for
    // edge_rows + 2(D-1) iterations, a fraction of an interior unit's sweep.  Reserving those slots for the whole launch
    // set (what the schedules of the other kernels do) made the interior launch of 8192x1024 9 % longer than the
    // undivided slab's (23 instead of 27 chunks per strip: 290 against 266 us, lbm_run_profiled).  Here every strip gets
    // n_full chunks of R rows for the slots that are free at once and, as the LAST units of the launch, two shorter
    // chunks of R - (edge iterations) rows: they are dispatched when the edge units retire and finish with the others.
    const int slots = s.cus * 4 * 2, edge_work = 2 * s.strips2, late_per_strip = 2;
    const int rows = i1 - i0, delay = s.edge_rows + 2 * (kDeepSteps - 1);
    const int vs = s.strips2 + nh;  // the one-round schedules below plan their wave slots for the heavy strips' copies too
    // Chunk PAIRS for the interior (d2q9_deep_twin<..., PUSH>, compact launch sets only): a strip's edge rows are one
    // workgroup (wave 0: bottom edge, wave 1: top edge, both running alone for edge_rows + 2(D-1) iterations), the interior
    // n_pairs workgroups of two chunks of R rows that start at their common boundary (R + D-1 iterations) and, as the last
    // workgroups of the launch, one late pair of R - delay rows that takes over the edge workgroup's slot.
    s.f6_main.paired = false;
    // (two-stream launch sets — RCCL, copies — keep the lone kernel: an interior launch of pairs next to the edge launch and the
    // exchange kernel was measured at 200 against 292 GLUPS on the 8192x1024 ring of one: the 40-KB pair workgroups crowd them out)
    if (c->pair != 0 && c->edge_aware != 0 && compact_sets(c)) {
      // (staged launch sets: NO late pair — the slots the edge workgroups free are where the exchange kernel of the edge stream runs,
      // on CUs whose other wave slots hold interior waves at 244 registers each)
      const bool late = !staged_sets(c);
      const int n_pairs = (slots - edge_work) / (2 * vs);
      const int Rp = n_pairs > 0 ? (late ? div_up(rows + 2 * delay, 2 * n_pairs + 2) : div_up(rows, 2 * n_pairs)) : 0;
      const int rp_late = late ? Rp - delay : 4;
      if (n_pairs >= 1 && Rp <= c6one && rp_late >= 4) {
        std::vector<int> starts;
        int y = i0, left = rows;
        const int nch = late ? 2 * n_pairs + 2 : 2 * n_pairs;
        for (int k = 0; k < nch; k++) {
          const int remaining_full = std::max(0, 2 * n_pairs - k);
          int sz = k < 2 * n_pairs ? div_up(std::max(0, left - (late ? 2 * rp_late : 0)), std::max(1, remaining_full)) : std::min(rp_late, left);
          if (k == nch - 1) sz = left;
          sz = std::max(0, std::min(sz, left));
          starts.push_back(y);
          y += sz;
          left -= sz;
        }
        starts.push_back(i1);
        FuseGeom &g = s.f6_main;
        g.nbands = 1;
        g.nchunks = nch;
        g.units_per_band = g.nchunks * vs;
        g.units = g.units_per_band;
        g.single_round = true;
        g.paired = true;
        g.starts = starts;
        if (g.chunk_start) HIP_TRY(hipFree(g.chunk_start));
        g.chunk_start = nullptr;
        if (dev_alloc(&g.chunk_start, starts.size())) return LBM_ERR_HIP;
        HIP_TRY(hipMemcpy(g.chunk_start, starts.data(), starts.size() * sizeof(int), hipMemcpyHostToDevice));
        if (int rc = balance_heavy_strips(s, g, true, s.strips2, nh)) return rc;
        s.nb_total = std::max(s.nb_total, s.f6_main.units + e.units);
        return LBM_OK;
      }
    }
    const int n_full = (slots - edge_work) / vs;
    const int R = n_full > 0 ? div_up(rows + late_per_strip * delay, n_full + late_per_strip) : 0;
    const int r_late = R - delay;
    if (c->edge_aware != 0 && n_full >= 2 && R <= c6max && r_late >= 4) {  // (the lone kernel's two-stream sets: 8192x4096 over RCCL 433 GLUPS
                                                                            // tapered, 389 as one round of 153-row chunks — c6max, not c6one)
      std::vector<int> starts;
      int y = i0, left = rows;
      for (int k = 0; k < n_full + late_per_strip; k++) {
        // the late chunks take exactly r_late rows (as far as rows are left); the full ones share the rest evenly
        const int remaining_full = std::max(0, n_full - k);
        int sz = k < n_full ? div_up(std::max(0, left - late_per_strip * r_late), std::max(1, remaining_full)) : std::min(r_late, left);
        if (k == n_full + late_per_strip - 1) sz = left;
        sz = std::max(0, std::min(sz, left));
        starts.push_back(y);
        y += sz;
        left -= sz;
      }
      starts.push_back(i1);
      FuseGeom &g = s.f6_main;
      g.nbands = 1;
      g.nchunks = n_full + late_per_strip;
      g.units_per_band = g.nchunks * vs;
      g.units = g.units_per_band;
      g.single_round = true;
      g.paired = false;
      g.starts = starts;
      if (g.chunk_start) HIP_TRY(hipFree(g.chunk_start));
      g.chunk_start = nullptr;
      if (dev_alloc(&g.chunk_start, starts.size())) return LBM_ERR_HIP;
      HIP_TRY(hipMemcpy(g.chunk_start, starts.data(), starts.size() * sizeof(int), hipMemcpyHostToDevice));
      if (int rc = balance_heavy_strips(s, g, false, s.strips2, nh)) return rc;
    } else {
      // more rows than one round of units takes (the slabs of a 2-GPU run, of the weak-scaling leg): the tapered multi-round
      // schedule, as chunk pairs where the launch set is compact (measured on one slab without halo rows: 8192x4096 383 -> 405)
      const bool pairs = c->pair != 0 && compact_sets(c) && rows > (long)c6max * n_full;
      if (int rc = fuse_schedule(s, i0, i1, c6max, c6min, true, s.f6_main, 2, 2 * edge_work, pairs, s.strips2)) return rc;
      s.f6_main.paired = pairs;
    }
  }
  s.nb_total = std::max(s.nb_total, s.f6_main.units + e.units);
  return LBM_OK;
}

// Compact launch sets of d2q9_deep_twin<5, ..., PUSH> (slab_twin5): strips of up to 60 lanes of two cells as on one slab; the
// five rows at either end of the slab are a chunk pair each (edge table {b0, b1, b2 | interior | t0, t0, t1, t2}: chunks 0/1 and
// 4/5, chunk 2 = the interior, chunk 3 empty — see the kernel), the interior one round of chunk pairs on the wave slots
// the 2 x strips edge workgroups leave.
int twin5_slab_geometry(const lbm_ctx *c, Slab &s) {
  const int q2 = c->p.nx / 2, lmax5 = 64 - 2 * lbm::deep_halo_lanes(kDeepTwinDefault);
  s.strips_tw = div_up(q2, lmax5 / 4 * 4);
  s.lanes_tw = std::min(lmax5, (div_up(q2, s.strips_tw) + 3) / 4 * 4);
  free_balance(s.f6_main);
  free_balance(s.f6_twin);
  FuseGeom &e = s.f6_edge;
  const int H = kDeepTwinDefault, b0 = s.row0, t0 = s.row0 + s.rows - H;  // (the five nearest of the stored halo rows: 5 or 8 of them)
  const int tab[7] = {b0, b0 + H / 2, b0 + H, t0, t0, t0 + H - H / 2, t0 + H};
  if (set_dev(s)) return LBM_ERR_HIP;
  if (e.chunk_start) HIP_TRY(hipFree(e.chunk_start));
  e.chunk_start = nullptr;
  if (dev_alloc(&e.chunk_start, 7)) return LBM_ERR_HIP;
  HIP_TRY(hipMemcpy(e.chunk_start, tab, sizeof(tab), hipMemcpyHostToDevice));
  e.nchunks = 6;
  e.skip = 2;
  e.nbands = 1;
  e.units_per_band = e.units = 4 * s.strips_tw;  // edge WAVES: one slot of the velocity sums each
  FuseGeom &g = s.f6_main;
  g.paired = true;
  if (tab[3] > tab[2]) {
    const int c6max = std::max(8, std::min(c->chunk_rows > 0 ? c->chunk_rows : 96, s.rows));
    const int c6min = std::max(4, std::min(c->chunk_min > 0 ? c->chunk_min : 24, c6max));
    if (int rc = fuse_schedule(s, tab[2], tab[3], c6max, c6min, true, g, 2, e.units, true, s.strips_tw)) return rc;
  }
  s.nb_total = std::max(s.nb_total, g.units + e.units);
  return LBM_OK;
}

// All launch geometry of a slab (single-step workgroup counts, fused schedules, ring slot stride).
int slab_geometry(const lbm_ctx *c, Slab &s) {
  const bool multi = c->halo_mode;
  // slab mode: the edge launch computes the `halo_depth` rows at each slab edge (what the neighbours receive),
  // the interior launch the rest
  s.edge_rows = multi ? std::min(c->halo_depth, s.rows / 2) : 0;
  if (multi) {
    s.nb_edge = step_blocks(c, 2 * s.edge_rows);
    s.nb_main = step_blocks(c, std::max(1, s.rows - 2 * s.edge_rows));
  } else {
    s.nb_main = step_blocks(c, s.rows);
    s.nb_edge = 0;
  }
  s.nb_total = s.nb_main + s.nb_edge;
  // tile size of d2q9_multi: the largest of 32x16, 16x16, 16x8 that still gives ~one tile per two CUs
  {
    static const int shapes[3][2] = {{32, 16}, {16, 16}, {16, 8}};
    // A tile's cost is its cell updates over the T = 8 shrinking sub-steps (32x16: 7344, 16x16: 4400, 16x8: 2928) and
    // a CU works through ceil(tiles / 256) of them, so take the shape with the least work per CU (ties: the larger
    // tile, less redundant halo).  Matches every measurement (tools/ab.py --opts tile_shape=.., us/step for 32x16 /
    // 16x16 / 16x8): 128x128 1.85 / 1.55 / 1.38, 128x256 1.94 / 1.63 / 1.49, 256x256 1.97 / 1.85 / 2.07, 384x384 3.86 /
    // 3.46 / 4.25, 512x512 3.89 / 4.12 / 6.18 and the slabs of a ring (r02): 1024x128 2.51 / 3.41 / 5.25, 1024x256
    // 3.97 / 5.95 / 9.66, 1024x512 7.13 / 11.15 / 18.65 (round 1 chose by tile count alone and gave those slabs 16x16)
    static const long updates[3] = {7344, 4400, 2928};
    int pick = 0;
    long best = -1;
    for (int k = 0; k < 3; k++) {
      const long tiles = (long)div_up(c->p.nx, shapes[k][0]) * div_up(s.rows, shapes[k][1]);
      const long work = (long)div_up(tiles, s.cus) * updates[k];
      if (best < 0 || work < best) { best = work; pick = k; }
    }
    if (c->tile_shape >= 0) pick = std::min(2, c->tile_shape);
    // slab mode: the edge tile rows must cover the halo depth
    if (multi && shapes[pick][1] < c->halo_depth) pick = 1;
    s.m_tx = shapes[pick][0];
    s.m_ty = shapes[pick][1];
  }
  s.m_tiles_x = div_up(c->p.nx, s.m_tx);
  s.m_tiles_y = div_up(s.rows, s.m_ty);
  if ((long)s.m_tiles_x * s.m_tiles_y <= 65536 * 4) s.nb_total = std::max(s.nb_total, s.m_tiles_x * s.m_tiles_y);
  if (fuse_possible(c)) {
    const int q4 = c->p.nx / 4;
    // x decomposition: lanes 0 and 63 of a wave are halo lanes, so a strip has at most 62 output lanes — but
    // strips must start on 64-byte boundaries (multiples of 4 lanes): with 61-lane strips (976 B) the stores of
    // neighbouring strips split 64-B DRAM bursts and the kernel ran 9 % slower (147.8 vs 162.1 GLUPS on 8192x8192,
    // tools/ab.py) although 60-lane strips leave more lanes idle
    s.strips = div_up(q4, 60);
    s.lanes_out = std::min(60, (div_up(q4, s.strips) + 3) / 4 * 4);
    if (g_defaults.lanes_out > 0) {  // lbm_set_default("lanes_out"): output lanes per strip (validated 4..62)
      s.lanes_out = g_defaults.lanes_out;
      s.strips = div_up(q4, s.lanes_out);
    }
    // measured optimum (profiles/r01_fused_sweep.txt): short chunks — the rows concurrently in flight on an
    // XCD then fit the caches, which absorbs the re-read boundary rows; 6/2 from 4096x4096 up, 8/4 below
    const bool big = (long)c->p.nx * c->rows_min >= 8L << 20;
    int cmax = c->chunk_rows > 0 ? c->chunk_rows : (big ? 6 : 8);
    cmax = std::max(2, std::min(cmax, s.rows));
    const int cmin = std::max(2, std::min(c->chunk_min > 0 ? c->chunk_min : (big ? 2 : 4), cmax));
    // d2q9_step3: six redundant intermediate rows per chunk.  With the windows in LDS (two waves per SIMD) the
    // schedule is flat between 8/4 and 32/6 (tools/ab.py, 8192x8192: 32/6 226, 16/6 229, 8/4 230, 6/2 224,
    // 4/2 211 GLUPS; 4096x4096: 32/6 215, 16/6 213, 8/4 212; 8192x1024: 32/6 195, 16/6 201, 8/2 190); with register
    // windows (one wave per SIMD) long chunks: 8192x8192 16/6 178, 32/6 186, 64/8 188
    const bool w_lds = windows_in_lds(c);
    int c3max = c->chunk_rows > 0 ? c->chunk_rows : (w_lds ? 16 : (c->rows_min >= 2048 ? 64 : 32));
    c3max = std::max(4, std::min(c3max, s.rows));
    const int c3min = std::max(2, std::min(c->chunk_min > 0 ? c->chunk_min : (!w_lds && c->rows_min >= 2048 ? 8 : 6), c3max));
    // d2q9_step4: twelve redundant intermediate rows per chunk -> longer chunks (tools/ab.py, chunks 16/6,
    // 32/8, 64/16, 128/32: 8192x8192 288 / 299 / 298 / 295 GLUPS, 4096x4096 247 / 252 / 275 / 275, 8192x1024
    // 223 / 246 / 246 / 246)
    const int c4max = std::max(4, std::min(c->chunk_rows > 0 ? c->chunk_rows : 64, s.rows));
    const int c4min = std::max(2, std::min(c->chunk_min > 0 ? c->chunk_min : 16, c4max));
    if (multi) {
      // the two edge chunks hold the rows the neighbours need (2 each); the interior is everything else
      // (edge chunks are as short as the exchange allows — 2 rows when the halo depth is 2: an edge unit is one
      // wave's serial sweep and, with the exchange, the critical path of a launch set, profiles/r01_overlap_trace.txt)
      FuseGeom &e = s.f_edge;
      // edge schedule: chunk table {bottom edge rows, interior, top edge rows}; the launch skips the interior chunk.
      // One chunk per edge.  Splitting an edge into single-row chunks (a round-1 experiment) shortens the edge
      // kernel of d2q9_step3 from 47 to 32 us (5 instead of 7 iterations per wave) but did not shorten the launch set:
      // 8192x1024 ring of one 45.5 -> 47.0 us/step, 4096x512 18.3 -> 17.0, 2048x256 9.1 -> 9.3 (tools/ab.py)
      const int ec = s.edge_rows;
      std::vector<int> tab;
      for (int y = 0; y < s.edge_rows; y += ec) tab.push_back(s.row0 + y);
      const int n_edge_chunks = (int)tab.size();
      tab.push_back(s.row0 + s.edge_rows);
      for (int y = 0; y < s.edge_rows; y += ec) tab.push_back(s.row0 + s.rows - s.edge_rows + y);
      tab.push_back(s.row0 + s.rows);
      if (tab.size() % 2 == 0) tab.push_back(s.row0 + s.rows);  // an empty last chunk: even chunk count for d2q9_step3p
      if (set_dev(s)) return LBM_ERR_HIP;
      if (e.chunk_start) HIP_TRY(hipFree(e.chunk_start));
      e.chunk_start = nullptr;
      if (dev_alloc(&e.chunk_start, tab.size())) return LBM_ERR_HIP;
      HIP_TRY(hipMemcpy(e.chunk_start, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
      e.nchunks = (int)tab.size() - 1;    // the middle chunk [row0+edge, row0+rows-edge) is skipped by the kernel
      e.skip = n_edge_chunks;
      e.nbands = 1;
      e.units_per_band = e.nchunks * s.strips;
      e.units = e.units_per_band;
      const int i0 = s.row0 + s.edge_rows, i1 = s.row0 + s.rows - s.edge_rows;
      if (i1 > i0) {
        // The interior launch leaves the edge launch its wave slots: an interior schedule that fills all 2048 slots
        // in one round of equal units starves the edge workgroups that were not dispatched first until other EDGE
        // workgroups retire (kernel trace, 8192x1024 slab, d2q9_step4: edge kernel 140 us instead of 50, and with
        // the exchange behind it the critical path of the launch set)
        // (compact launch sets: the edge units are the first workgroups of the same launch and hold their slots — a pair
        // workgroup both of its, also where one chunk of the pair is the skipped interior — until they are done)
        const int rsv = compact_sets(c) ? 4 * n_edge_chunks * s.strips : 2 * n_edge_chunks * s.strips;  // the edge units that do work (skipped and empty chunks exit at once)
        if (int rc = fuse_schedule(s, i0, i1, cmax, cmin, true, s.f_main, 2, rsv)) return rc;
        if (int rc = fuse_schedule_pairs(c, s, i0, i1, c3max, c3min, s.f3_main, step3_sched_waves(c), rsv, step3_can_pair(c))) return rc;
        if (int rc = fuse_schedule_pairs(c, s, i0, i1, c4max, c4min, s.f4_main, step4_sched_waves(c), rsv, windows_in_lds(c))) return rc;
      } else {
        // no interior: the edge chunks are the whole slab (the launch form — pair kernel or not — is still taken from
        // these schedules)
        s.f_main.units = s.f3_main.units = s.f4_main.units = 0;
        s.f_main.paired = s.f3_main.paired = s.f4_main.paired = false;
      }
      s.nb_total = std::max(s.nb_total, std::max(s.f_main.units, std::max(s.f3_main.units, s.f4_main.units)) + e.units);
    } else {
      if (int rc = fuse_schedule(s, 0, s.rows, cmax, cmin, true, s.f_main)) return rc;
      s.nb_total = std::max(s.nb_total, s.f_main.units);
      if (int rc = fuse_schedule_pairs(c, s, 0, s.rows, c3max, c3min, s.f3_main, step3_sched_waves(c), 0, step3_can_pair(c))) return rc;
      s.nb_total = std::max(s.nb_total, s.f3_main.units);
      if (int rc = fuse_schedule_pairs(c, s, 0, s.rows, c4max, c4min, s.f4_main, step4_sched_waves(c), 0, windows_in_lds(c))) return rc;
      s.nb_total = std::max(s.nb_total, s.f4_main.units);
    }
    if (int rc = deep_geometry(c, s)) return rc;
  }
  if (int rc = resident_geometry(c, s)) return rc;
  return LBM_OK;
}

bool nt_effective(const lbm_ctx *c) {
  if (c->nt_stores >= 0) return c->nt_stores != 0;
  // both grids of a slab fit the 256 MiB Infinity Cache -> keep the written lines cacheable
  const size_t grid_bytes = c->slabs[0].row_stride * c->slabs[0].ext_rows * sizeof(float);
  return 2 * grid_bytes > ((size_t)192 << 20);
}

// which LoadMode a context uses: option "variant" 1..4 = LM_SCALAR, LM_UNALIGNED, LM_DPP, LM_LDS;
// 0 = auto.  The wave-level modes need whole waves inside one row (nx % 256 == 0).
int effective_mode(const lbm_ctx *c) {
  if (!c->vec4) return LM_SCALAR;
  const bool wave_rows = (c->p.nx % 256 == 0);
  int m = c->variant > 0 ? c->variant - 1 : (wave_rows ? LM_DPP : LM_SCALAR);
  if ((m == LM_DPP || m == LM_LDS) && !wave_rows) m = LM_SCALAR;
  return m;
}

template <int LM>
void launch_step_lm(bool nt, const StepArgs &a, int nblocks, hipStream_t st) {
  if (nt) hipLaunchKernelGGL((d2q9_step<4, true, LM>), dim3(nblocks), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL((d2q9_step<4, false, LM>), dim3(nblocks), dim3(kBlock), 0, st, a);
}

void launch_step(const lbm_ctx *c, const StepArgs &a, int nblocks, hipStream_t st) {
  const bool nt = nt_effective(c);
  if (!c->vec4) {
    if (nt) hipLaunchKernelGGL((d2q9_step<1, true, LM_SCALAR>), dim3(nblocks), dim3(kBlock), 0, st, a);
    else hipLaunchKernelGGL((d2q9_step<1, false, LM_SCALAR>), dim3(nblocks), dim3(kBlock), 0, st, a);
    return;
  }
  switch (effective_mode(c)) {
    case LM_UNALIGNED: launch_step_lm<LM_UNALIGNED>(nt, a, nblocks, st); break;
    case LM_DPP: launch_step_lm<LM_DPP>(nt, a, nblocks, st); break;
    case LM_LDS: launch_step_lm<LM_LDS>(nt, a, nblocks, st); break;
    default: launch_step_lm<LM_SCALAR>(nt, a, nblocks, st); break;
  }
}

// single-step kernel arguments common to all launches of a slab; rows are STORED-row indices
StepArgs base_args(const lbm_ctx *c, const Slab &s, int src, bool apply_accel) {
  StepArgs a{};
  a.src = s.cells[src];
  a.dst = s.cells[src ^ 1];
  a.mask = s.mask;
  a.plane_stride = s.plane_stride;
  a.row_stride = s.row_stride;
  a.nx = c->p.nx;
  a.rows = s.ext_rows;
  a.accel_row = apply_accel ? s.accel_own : -1;
  a.omega = c->p.omega;
  a.aw1 = c->p.density * c->p.accel / 9.0f;
  a.aw2 = c->p.density * c->p.accel / 36.0f;
  for (int k = 0; k < 3; k++) {
    static const int sp[3] = {2, 5, 6}, np[3] = {4, 7, 8};
    // only dereferenced for stored row 0 / ext_rows-1, i.e. with one slab: the y wrap (kernels.cl:91-93)
    a.south_src[k] = s.cells[src] + sp[k] * s.plane_stride + (size_t)(s.ext_rows - 1) * s.row_stride;
    a.north_src[k] = s.cells[src] + np[k] * s.plane_stride;
  }
  return a;
}

Step2Args base_args2(const lbm_ctx *c, const Slab &s, int src, bool accel_next, const FuseGeom &g) {
  Step2Args a{};
  a.src = s.cells[src];
  a.dst = s.cells[src ^ 1];
  a.mask = s.mask;
  a.plane_stride = s.plane_stride;
  a.row_stride = s.row_stride;
  a.nx = c->p.nx;
  a.ny = s.ext_rows;  // rows wrap only with one slab; with halos the chunks never reach row 0 / ext_rows-1
  a.strips = s.strips;
  a.lanes_out = s.lanes_out;
  a.chunk_start = g.chunk_start;
  a.vmap = g.vmap;
  a.vtab = g.vtab;
  a.nchunks = g.nchunks;
  a.nbands = g.nbands;
  a.units_per_band = g.units_per_band;
  a.skip_chunk = -1;
  a.accel_row = s.accel_ext;
  a.accel_row_b = s.accel_ext_b;
  a.accel_next = accel_next ? 1 : 0;
  a.omega = c->p.omega;
  a.aw1 = c->p.density * c->p.accel / 9.0f;
  a.aw2 = c->p.density * c->p.accel / 36.0f;
  return a;
}

// d2q9_step3 in the ONE form that is left of it: windows in LDS (two waves per SIMD), one row-set of loads in flight, plain loads;
// d2q9_step3p = its chunk pairs.  (Round 4 removed the register-window form, the form with two row-sets of loads in flight and the
// non-temporal-load variants — nine instantiations at 253-256 registers that no policy selected: LDS windows 230 against 188 GLUPS on
// 8192x8192, plain loads 227.6 against 221.4 hybrid / 202.3 non-temporal, tools/ab.py, rounds 1-2.)
void launch_step3(const lbm_ctx *c, const Step2Args &a0, float *partials3, int units, hipStream_t st, bool paired = false) {
  (void)c;
  if (paired) {
    // one workgroup of two waves per pair of chunks: a.units_per_band counts pairs x strips
    Step2Args a = a0;
    a.units_per_band = a0.units_per_band / 2;
    hipLaunchKernelGGL((d2q9_step3p<true, 0>), dim3(units / 2), dim3(128), 0, st, a, partials3);
    return;
  }
  hipLaunchKernelGGL((d2q9_step3<true, 0, true, 1>), dim3(units), dim3(64), 0, st, a0, partials3);
}

void launch_step4(const lbm_ctx *c, const Step2Args &a0, float *partials3, float *partials4, int units, hipStream_t st,
                  bool paired = false) {
  // source loads are always plain here: the kernel sits at the 256-VGPR limit and its non-temporal forms spill (6 and
  // 8 registers to scratch) — option "nt_loads" applies to the two- and three-step kernels only (lbm_set_option)
  (void)c;
  if (paired) {
    Step2Args a = a0;
    a.units_per_band = a0.units_per_band / 2;  // chunk pairs x strips
    hipLaunchKernelGGL((d2q9_step4p<true, 0>), dim3(units / 2), dim3(128), 0, st, a, partials3, partials4);
    return;
  }
  hipLaunchKernelGGL((d2q9_step4<true, 0>), dim3(units), dim3(64), 0, st, a0, partials3, partials4);
}

// compact launch set of the three- / four-step kernels: edge units first, then the interior units, one launch
void launch_deep(const lbm_ctx *c, const Slab &s, const FuseGeom &g, const Step2Args &a0, int units, float *partials, int nlev, hipStream_t st) {
  Step2Args a = a0;
  a.strips = g.vstrips > 0 ? g.vstrips : s.strips2;
  a.strips_edge = s.strips2;
  a.lanes_out = s.lanes2;
  a.clean_bits = clean_bits_for(c, s, s.f6_main);  // (an edge launch beside the interior launch follows the interior's policy)
  a.clean_words = s.clean_words;
  const dim3 grid(units), block(64);
  const bool nt = nt_effective(c), paths = c->obst_paths != 0;  // (-1 auto = on)
  // the depths whole runs are cut into have a kernel of their own (steady form of the row loop, see deep_sweep); the
  // default configuration only — every other combination of options runs the any-depth kernel
  if (nt && paths && c->steady != 0 && nlev == 8) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, true, false, 8>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths && c->steady != 0 && nlev == 7) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, true, false, 7>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths && c->steady != 0 && nlev == 6) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, true, false, 6>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, false>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (paths) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, false, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else hipLaunchKernelGGL((d2q9_deep<kDeepSteps, false, false>), grid, block, 0, st, a, partials, s.nb_total, nlev);
}

// compact launch set of d2q9_deep: the edge units first, then the interior units, ONE launch
void launch_deep_compact(const lbm_ctx *c, const Slab &s, const Step2Args &a0, float *partials, int nlev, hipStream_t st) {
  Step2Args a = a0;
  a.strips = s.f6_main.vstrips > 0 ? s.f6_main.vstrips : s.strips2;
  a.strips_edge = s.strips2;
  a.lanes_out = s.lanes2;
  a.clean_bits = clean_bits_for(c, s, s.f6_main);
  a.clean_words = s.clean_words;
  const bool nt = nt_effective(c), paths = c->obst_paths != 0;
  if (slab_twin5(c)) {
    // five halo rows: interior chunk pairs + two edge chunk pairs per strip, d2q9_deep_twin<5, ..., PUSH> (twin5_slab_geometry)
    a.strips = a.strips_edge = s.strips_tw;
    a.lanes_out = s.lanes_tw;
    a.clean_bits = nullptr;
    a.units_per_band = a0.units_per_band / 2;  // chunk pairs x strips
    a.edge_units = 4 * s.strips_tw;            // edge WAVES
    const dim3 pgrid(2 * s.strips_tw + s.f6_main.units / 2), pblock(128);
    constexpr int D5 = kDeepTwinDefault;
    // (plain stores only: both grids of a slab of this size fit the Infinity Cache up to 2.8M cells; no non-temporal instantiations)
    if (paths && c->steady != 0 && nlev == D5) hipLaunchKernelGGL((d2q9_deep_twin<D5, false, true, D5, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    else if (paths) hipLaunchKernelGGL((d2q9_deep_twin<D5, false, true, 0, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    else hipLaunchKernelGGL((d2q9_deep_twin<D5, false, false, 0, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    return;
  }
  if (s.f6_main.paired) {
    // interior chunk pairs + one edge workgroup per strip (bottom and top edge rows on its two waves): d2q9_deep_twin<..., PUSH>
    a.units_per_band = a0.units_per_band / 2;     // chunk pairs x strips
    a.edge_units = 2 * s.strips2;                 // edge WAVES (what the last of them counts up to); edge workgroups = half
    const dim3 pgrid(s.strips2 + s.f6_main.units / 2), pblock(128);
    if (nt && paths && c->steady != 0 && nlev == 8) hipLaunchKernelGGL((d2q9_deep_twin<kDeepSteps, true, true, 8, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    else if (nt && paths && c->steady != 0 && nlev == 7) hipLaunchKernelGGL((d2q9_deep_twin<kDeepSteps, true, true, 7, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    else if (nt && paths && c->steady != 0 && nlev == 6) hipLaunchKernelGGL((d2q9_deep_twin<kDeepSteps, true, true, 6, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    else if (nt && paths) hipLaunchKernelGGL((d2q9_deep_twin<kDeepSteps, true, true, 0, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    else if (nt) hipLaunchKernelGGL((d2q9_deep_twin<kDeepSteps, true, false, 0, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    else if (paths) hipLaunchKernelGGL((d2q9_deep_twin<kDeepSteps, false, true, 0, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    else hipLaunchKernelGGL((d2q9_deep_twin<kDeepSteps, false, false, 0, true>), pgrid, pblock, 0, st, a, partials, s.nb_total, nlev);
    return;
  }
  const dim3 grid(a0.edge_units + s.f6_main.units), block(64);
  if (nt && paths && c->steady != 0 && nlev == 8) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, true, true, 8>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths && c->steady != 0 && nlev == 7) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, true, true, 7>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths && c->steady != 0 && nlev == 6) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, true, true, 6>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, true, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, true, false, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (paths) hipLaunchKernelGGL((d2q9_deep<kDeepSteps, false, true, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else hipLaunchKernelGGL((d2q9_deep<kDeepSteps, false, false, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
}

void launch_deep_twin(const lbm_ctx *c, const Slab &s, const Step2Args &a0, float *partials, int nlev, hipStream_t st) {
  Step2Args a = a0;
  a.strips = s.f6_twin.vstrips > 0 ? s.f6_twin.vstrips : s.strips_tw;
  a.strips_edge = s.strips_tw;
  a.lanes_out = s.lanes_tw;
  a.units_per_band = a0.units_per_band / 2;  // chunk pairs x strips
  const dim3 grid(s.f6_twin.units / 2), block(128);
  const bool nt = nt_effective(c), paths = c->obst_paths != 0;
  if (twin_cap(c) <= kDeepTwinDefault) {
    // all windows in LDS, no mailbox, 2 halo lanes per side
    if (nt && paths && c->steady != 0 && nlev == kDeepTwinDefault) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinDefault, true, true, kDeepTwinDefault>), grid, block, 0, st, a, partials, s.nb_total, nlev);
    else if (!nt && paths && c->steady != 0 && nlev == kDeepTwinDefault) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinDefault, false, true, kDeepTwinDefault>), grid, block, 0, st, a, partials, s.nb_total, nlev);
    else if (nt && paths) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinDefault, true, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
    else if (nt) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinDefault, true, false>), grid, block, 0, st, a, partials, s.nb_total, nlev);
    else if (paths) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinDefault, false, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
    else hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinDefault, false, false>), grid, block, 0, st, a, partials, s.nb_total, nlev);
    return;
  }
  a.clean_bits = clean_bits_for(c, s, s.f6_twin);  // (eight-step twins run d2q9_deep's strips: strips_tw == strips2)
  a.clean_words = s.clean_words;
  if (nt && paths && c->steady != 0 && nlev == kDeepTwinSteps) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinSteps, true, true, kDeepTwinSteps>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths && c->steady != 0 && nlev == 7) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinSteps, true, true, 7>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths && c->steady != 0 && nlev == 6) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinSteps, true, true, 6>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt && paths) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinSteps, true, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (nt) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinSteps, true, false>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else if (paths) hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinSteps, false, true>), grid, block, 0, st, a, partials, s.nb_total, nlev);
  else hipLaunchKernelGGL((d2q9_deep_twin<kDeepTwinSteps, false, false>), grid, block, 0, st, a, partials, s.nb_total, nlev);
}

void launch_compact(int level, bool paired, const Step2Args &a0, float *partials3, float *partials4, int main_units, hipStream_t st) {
  Step2Args a = a0;
  if (paired) {
    a.units_per_band = a0.units_per_band / 2;  // chunk pairs x strips
    const dim3 grid((a0.edge_units + main_units) / 2), block(128);
    if (level == 4) hipLaunchKernelGGL((d2q9_step4p<true, 0, true>), grid, block, 0, st, a, partials3, partials4);
    else hipLaunchKernelGGL((d2q9_step3p<true, 0, true>), grid, block, 0, st, a, partials3);
    return;
  }
  const dim3 grid(a0.edge_units + main_units), block(64);
  if (level == 4) hipLaunchKernelGGL((d2q9_step4<true, 0, true>), grid, block, 0, st, a, partials3, partials4);
  else hipLaunchKernelGGL((d2q9_step3<true, 0, true, 1, true>), grid, block, 0, st, a, partials3);
}

MultiArgs base_args_multi(const lbm_ctx *c, const Slab &s, int src, int T, bool accel_next) {
  MultiArgs a{};
  a.src = s.cells[src];
  a.dst = s.cells[src ^ 1];
  a.mask = s.mask;
  a.plane_stride = s.plane_stride;
  a.row_stride = s.row_stride;
  a.partials_stride = (unsigned long long)s.nb_total;
  a.nx = c->p.nx;
  a.rows = s.rows;
  a.ext_rows = s.ext_rows;
  a.row_off = s.row0;
  a.tiles_x = s.m_tiles_x;
  a.T = T;
  a.gy_off = s.y0 - s.row0;
  a.ny_global = c->p.ny;
  a.accel_next = accel_next ? 1 : 0;
  a.omega = c->p.omega;
  a.aw1 = c->p.density * c->p.accel / 9.0f;
  a.aw2 = c->p.density * c->p.accel / 36.0f;
  return a;
}

void launch_multi(const Slab &s, const MultiArgs &a, int tile_rows, hipStream_t st, bool peer_form = false) {
  const dim3 grid(s.m_tiles_x * tile_rows), block(kMultiThreads);
  if (peer_form) {
    if (s.m_tx == 32) hipLaunchKernelGGL((d2q9_multi<32, 16, true>), grid, block, 0, st, a);
    else if (s.m_ty == 16) hipLaunchKernelGGL((d2q9_multi<16, 16, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((d2q9_multi<16, 8, true>), grid, block, 0, st, a);
    return;
  }
  if (s.m_tx == 32) hipLaunchKernelGGL((d2q9_multi<32, 16>), grid, block, 0, st, a);
  else if (s.m_ty == 16) hipLaunchKernelGGL((d2q9_multi<16, 16>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((d2q9_multi<16, 8>), grid, block, 0, st, a);
}

// `nsteps` timesteps in ONE launch (the caller keeps them within the ring of partial sums)
void launch_resident(lbm_ctx *c, Slab &s, int src, int nsteps, bool accel_next, float *partials, hipStream_t st) {
  ResidentArgs a{};
  a.src = s.cells[src];
  a.dst = s.cells[src ^ 1];
  a.mask = s.mask;
  a.partials = partials;
  a.plane_stride = s.plane_stride;
  a.row_stride = s.row_stride;
  a.pstride = (unsigned long long)s.nb_total;
  a.nx = c->p.nx;
  a.ny = c->p.ny;
  a.nsteps = nsteps;
  a.accel_row = s.accel_ext;
  a.accel_next = accel_next ? 1 : 0;
  a.omega = c->p.omega;
  a.aw1 = c->p.density * c->p.accel / 9.0f;
  a.aw2 = c->p.density * c->p.accel / 36.0f;
  a.words = s.res_words;
  a.xrows = s.res_xrows;
  a.seq_base = s.res_seq;
  a.err = s.res_err;
  a.wait_ticks = c->halo_timeout_ms * kTicksPerMs;
  s.res_seq += (unsigned)nsteps + 1u;
  // A resident launch needs every band on the device at once: two of them running side by side (two contexts of this process on
  // their own streams) would each hold CUs the other is waiting for, until the timeout.  One event per device orders them.
  // (Another PROCESS on the same device is the caller's business, as is any long kernel of its own that fills the device.)
  static hipEvent_t res_done[64] = {};
  hipEvent_t &ev = res_done[s.dev & 63];
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
  if (ev) (void)hipStreamWaitEvent(st, ev, 0);
  const dim3 grid(s.res_bands), block(64 * s.res_w);   // (the waves across a band are a launch parameter of the one kernel per band height)
  if (s.res_bh == 2) hipLaunchKernelGGL((d2q9_resident<2>), grid, block, 0, st, a);
  else if (s.res_bh == 4) hipLaunchKernelGGL((d2q9_resident<4>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((d2q9_resident<6>), grid, block, 0, st, a);
  if (ev) (void)hipEventRecord(ev, st);
}

void launch_step2(const lbm_ctx *c, const Step2Args &a, int units, hipStream_t st) {
  // Non-temporal stores always (a result is not read again before the next launch).  Source loads: HYBRID —
  // non-temporal (lower latency, no cache pollution) for the rows only this chunk reads, plain for the two
  // intermediate rows at either end of the chunk, whose source rows the neighbouring chunk reads at the same
  // time and should find in L2.  All-nt loads lose that reuse (PMC on 8192x8192: 5.90 GB per launch instead
  // of 5.13 GB).  Same-box A/B (tools/ab.py), plain / all-nt / hybrid in GLUPS: 8192x8192 136 / 140 / 153,
  // 4096x4096 126 / 133 / 147, 2048x2048 110 / 117 / 128, 1024x1024 82 / 102 / 113.
  const bool nts = c->nt_stores >= 0 ? c->nt_stores != 0 : true;
  const int ntl = c->nt_loads >= 0 ? c->nt_loads : (nts ? 2 : 0);
  if (ntl == 2) hipLaunchKernelGGL((d2q9_step2<true, 2>), dim3(units), dim3(64), 0, st, a);
  else if (ntl == 1) hipLaunchKernelGGL((d2q9_step2<true, 1>), dim3(units), dim3(64), 0, st, a);
  else if (nts) hipLaunchKernelGGL((d2q9_step2<true>), dim3(units), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((d2q9_step2<false>), dim3(units), dim3(64), 0, st, a);
}

// RCCL behind the transport concept of halo_exchange.h
struct RcclTransport {
  int group_start() { return (int)g_rccl.GroupStart(); }
  int group_end() { return (int)g_rccl.GroupEnd(); }
  int send(const void *p, size_t n, int peer, void *comm, void *stream) {
    return (int)g_rccl.Send(p, n, ncclFloat, peer, (ncclComm_t)comm, (hipStream_t)stream);
  }
  int recv(void *p, size_t n, int peer, void *comm, void *stream) {
    return (int)g_rccl.Recv(p, n, ncclFloat, peer, (ncclComm_t)comm, (hipStream_t)stream);
  }
};

// how a slab's pushing waves order their rows before the ticket (release_pushed): fences between devices unless told otherwise
int push_release_effective(const lbm_ctx *c, const Slab &s) {
  if (c->push_release >= 0) return c->push_release;
  return (s.south.connected && s.south.remote) || (s.north.connected && s.north.remote) ? 1 : 0;
}

// TEST HOOK "debug_stale_exchange": is the exchange about to be issued the one that delivers nothing?
bool stale_exchange_now(lbm_ctx *c) { return c->stale_exchange > 0 && --c->stale_exchange == 0; }

// PEER transport: raise the neighbours' flag words to `seq` without moving a row (the stale-exchange test hook of a compact launch set)
int publish_flags_only(lbm_ctx *c, Slab &s, uint32_t seq, hipStream_t st) {
  PushArgs a{};
  a.n4 = 0;
  a.flag_lo = s.south.flags + 1;
  a.flag_hi = s.north.flags + 0;
  a.seq = seq;
  a.ticket = s.halo_flags + 3;
  a.release = push_release_effective(c, s);
  hipLaunchKernelGGL(halo_push, dim3(2), dim3(kBlock), 0, st, a);
  HIP_TRY(hipGetLastError());
  return LBM_OK;
}

// PEER transport, consumer side: the launch that follows on `st` reads halo rows that exchange number `seq` fills
int wait_halos(lbm_ctx *c, Slab &s, hipStream_t st, uint32_t seq) {
  // (halo_sync 2 — the wait inside the consuming kernel — exists for compact launch sets; every other launch is
  // ordered behind the wait kernel)
  if (c->halo_sync == 1) {
    HIP_TRY(hipStreamWaitValue32(st, s.halo_flags, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
    HIP_TRY(hipStreamWaitValue32(st, s.halo_flags + 1, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
  } else {
    hipLaunchKernelGGL(halo_wait, dim3(1), dim3(64), 0, st, s.halo_flags, seq, s.halo_flags + 2, c->halo_timeout_ms * kTicksPerMs);
    HIP_TRY(hipGetLastError());
  }
  return LBM_OK;
}

// Slab mode: move the `halo_depth` bottom and top owned rows of grid `buf` into the ring neighbours' halo
// rows of their grid `buf`.  Each group of rows is one contiguous block of halo_depth*row_stride floats.
int exchange_halos(lbm_ctx *c, int buf, int evq, bool on_main = false, bool staged = false) {
  const int P = c->nslabs_global;
  const bool stale = !staged && stale_exchange_now(c);  // (a staged set has asked the hook before its launch)
  if (c->transport_eff == TRANSPORT_PEER) {
    const uint32_t seq = c->halo_seq + 1;
    for (Slab &s : c->slabs) {
      if (!s.south.connected || !s.north.connected)
        return fail(LBM_ERR_STATE, "peer transport: the ring neighbours are not connected (lbm_connect_peers)");
      if (set_dev(s)) return LBM_ERR_HIP;
      const size_t count = (size_t)s.row0 * s.row_stride;
      float *g = s.cells[buf];
      PushArgs a{};
      a.src_lo = g + (size_t)s.row0 * s.row_stride;                              // my bottom rows -> south's top halo
      a.dst_lo = s.south.cells[buf] + (size_t)(s.row0 + s.south.rows) * s.row_stride;
      a.src_hi = g + (size_t)s.rows * s.row_stride;                              // my top rows -> north's bottom halo
      a.dst_hi = s.north.cells[buf];
      a.n4 = count / 4;
      a.flag_lo = s.south.flags + 1;  // I am the south neighbour's NORTH neighbour
      a.flag_hi = s.north.flags + 0;
      a.seq = seq;
      a.ticket = s.halo_flags + 3;
      a.release = push_release_effective(c, s);
      if (stale) a.n4 = 0;  // (test hook: the flags go up, the rows stay what they were)
      // enough workgroups to move the rows in a few microseconds without taking the chip from the interior launch
      const int per_side = (int)std::max<size_t>(1, std::min<size_t>(64, (a.n4 + 4 * kBlock - 1) / (4 * kBlock)));
      hipLaunchKernelGGL(halo_push, dim3(2 * per_side), dim3(kBlock), 0, on_main ? s.s_main : s.s_edge, a);
      HIP_TRY(hipGetLastError());
    }
    c->halo_seq = seq;
  } else if (c->transport_eff == TRANSPORT_RCCL) {
    if (stale) return LBM_OK;  // (test hook; every rank of the ring carries the same setting, so nobody waits for a send)
    std::vector<HaloBlock> blocks;
    for (Slab &s : c->slabs) {
      float *g = s.cells[buf];
      HaloBlock b{};
      b.count = (size_t)s.row0 * s.row_stride;
      b.north = (s.index + 1) % P;
      b.south = (s.index + P - 1) % P;
      b.send_north = staged ? s.stage[1] : g + (size_t)s.rows * s.row_stride;   // (staged sets: the blocks the edge units have filled)
      b.send_south = staged ? s.stage[0] : g + (size_t)s.row0 * s.row_stride;
      b.recv_south = g;
      b.recv_north = g + (size_t)(s.row0 + s.rows) * s.row_stride;
      b.comm = s.comm;
      b.stream = s.s_edge;
      blocks.push_back(b);
    }
    RcclTransport t;
    const char *op = nullptr;
    if (int rc = ring_exchange(t, blocks.data(), (int)blocks.size(), &op))
      return fail(LBM_ERR_COMM, "RCCL error during halo exchange (%s): %s", op ? op : "?",
                  g_rccl.GetErrorString ? g_rccl.GetErrorString((ncclResult_t)rc) : "?");
  } else {
    // one process: each slab pulls from its neighbours once their edge launches are done
    for (Slab &s : c->slabs) {
      if (set_dev(s)) return LBM_ERR_HIP;
      const size_t bytes = (size_t)s.row0 * s.row_stride * sizeof(float);
      Slab &north = c->slabs[(s.index + 1) % P];
      Slab &south = c->slabs[(s.index + P - 1) % P];
      HIP_TRY(hipStreamWaitEvent(s.s_edge, south.ev_edgek[evq], 0));
      HIP_TRY(hipStreamWaitEvent(s.s_edge, north.ev_edgek[evq], 0));
      if (stale) continue;
      HIP_TRY(hipMemcpyAsync(s.cells[buf], south.cells[buf] + (size_t)south.rows * south.row_stride, bytes,
                             hipMemcpyDeviceToDevice, s.s_edge));
      HIP_TRY(hipMemcpyAsync(s.cells[buf] + (size_t)(s.row0 + s.rows) * s.row_stride,
                             north.cells[buf] + (size_t)north.row0 * north.row_stride, bytes, hipMemcpyDeviceToDevice, s.s_edge));
    }
  }
  return LBM_OK;
}

int run_steps_impl(lbm_ctx *c, int nsteps, bool timed, double *ms, bool *launched) {
  if (nsteps < 0) return fail(LBM_ERR_ARG, "nsteps must be >= 0");
  if (c->failed) return fail(LBM_ERR_STATE, "an earlier run failed after its launches had begun; destroy the context");
  if (c->halo_mode && c->transport_eff == TRANSPORT_AUTO)
    return fail(LBM_ERR_STATE, "rank context without a transport: pass a comm_id to lbm_create_rank or call lbm_connect_peers");
  if (c->steps_done + nsteps > c->p.max_iters)
    return fail(LBM_ERR_STATE, "av_vels record holds max_iters=%d steps; %d done, %d more requested", c->p.max_iters,
                c->steps_done, nsteps);
  if (timed && ms) *ms = 0.0;
  if (nsteps == 0) return LBM_OK;
  const bool multi = c->halo_mode;
  const int fuse_lvl = fuse_level(c);
  const bool fuse = fuse_lvl != 0;
  const float aw1 = c->p.density * c->p.accel / 9.0f, aw2 = c->p.density * c->p.accel / 36.0f;
  const int nx = c->p.nx;

  // prologue: accelerate_flow of the first step on the current grid (kernels.cl:9-53); later
  // steps get theirs fused into the previous launch's write of row ny-2
  *launched = true;
  for (Slab &s : c->slabs) {
    if (set_dev(s)) return LBM_ERR_HIP;
    if (multi) HIP_TRY(hipStreamSynchronize(s.s_edge));
    if (s.accel_own >= 0) {
      hipLaunchKernelGGL(accelerate_row, dim3(div_up(nx, 128)), dim3(128), 0, s.s_main, s.cells[c->cur], s.plane_stride,
                         s.row_stride, s.mask, nx, s.accel_own, aw1, aw2);
      HIP_TRY(hipGetLastError());
    }
  }
  // compact launch sets (small slabs, peer transport): everything on the main stream
  const bool compact = compact_sets(c);
  const bool staged = staged_sets(c);  // (a compact form: the pushes go to local staging blocks, the exchange is RCCL's on the edge stream)
  if (multi) {
    // halos of the initial state ("launch set -1", event parity 1)
    for (Slab &s : c->slabs) {
      if (set_dev(s)) return LBM_ERR_HIP;
      HIP_TRY(hipEventRecord(s.ev_main[1], s.s_main));
      HIP_TRY(hipEventRecord(s.ev_edgek[1], s.s_main));
      HIP_TRY(hipStreamWaitEvent(s.s_edge, s.ev_main[1], 0));
    }
    if (int rc = exchange_halos(c, c->cur, 1, compact)) return rc;
    if (staged)  // the first launch is ordered behind this exchange as every later one behind its predecessor's
      for (Slab &s : c->slabs) {
        if (set_dev(s)) return LBM_ERR_HIP;
        HIP_TRY(hipEventRecord(s.ev_edgek[1], s.s_edge));
      }
  }
  if (timed)
    for (Slab &s : c->slabs) {
      if (set_dev(s)) return LBM_ERR_HIP;
      if (multi && (!compact || staged)) {
        // start the clock on the main stream once the initial halos have landed
        HIP_TRY(hipEventRecord(s.ev_aux, s.s_edge));
        HIP_TRY(hipStreamWaitEvent(s.s_main, s.ev_aux, 0));
      }
      HIP_TRY(hipEventRecord(s.ev_t0, s.s_main));
    }

  // d2q9_deep as chunk pairs: one slab without halo rows whose pair schedule is one round of units (all slabs alike)
  const bool deep_lvl = fuse_lvl >= kDeepMin || slab_twin5(c);  // the deep window kernels (slab_twin5: their five-step pairs in compact launch sets)
  const bool deep_twin = fuse_lvl >= kDeepMin && deep_twin_effective(c);
  int batch_first = c->steps_done;
  enum { KIND_NONE = 0, KIND_SINGLE = 1, KIND_FUSED2 = 2, KIND_MULTI = 3, KIND_FUSED3 = 4, KIND_FUSED4 = 5, KIND_DEEP = 6, KIND_RESIDENT = 7 };
  const bool resident = resident_effective(c);
  int batch_kind = KIND_NONE;  // launch kind of the steps buffered in the ring (their slot occupancy differs)
  int last_q = 1;
  const int multi_T = multistep_effective(c);
  // second reduction stage over the buffered steps of all slabs (kernels.cl:234-290 counterpart)
  auto flush = [&]() -> int {
    const int fill = c->ring_fill;
    if (fill == 0) return LBM_OK;
    for (Slab &s : c->slabs) {
      if (set_dev(s)) return LBM_ERR_HIP;
      if (multi && !compact) HIP_TRY(hipStreamWaitEvent(s.s_main, s.ev_edgek[last_q], 0));
      int used = s.nb_main + s.nb_edge;
      if (batch_kind == KIND_FUSED2) used = s.f_main.units + (multi ? s.f_edge.units : 0);
      if (batch_kind == KIND_FUSED3) used = s.f3_main.units + (multi ? s.f_edge.units : 0);
      if (batch_kind == KIND_FUSED4) used = s.f4_main.units + (multi ? s.f_edge.units : 0);
      if (batch_kind == KIND_DEEP) used = deep_twin ? s.f6_twin.units : s.f6_main.units + (multi ? s.f6_edge.units : 0);
      if (batch_kind == KIND_MULTI) used = s.m_tiles_x * s.m_tiles_y;
      if (batch_kind == KIND_RESIDENT) used = s.res_bands * s.res_w;
      hipLaunchKernelGGL(reduce_partials, dim3(fill), dim3(kBlock), 0, s.s_main, s.partials, s.nb_total, used,
                         s.av_sum + batch_first);
      HIP_TRY(hipGetLastError());
      if (multi && !compact) {
        // the next batch's edge launches overwrite ring slots: order them after this reduction
        HIP_TRY(hipEventRecord(s.ev_aux, s.s_main));
        HIP_TRY(hipStreamWaitEvent(s.s_edge, s.ev_aux, 0));
      }
    }
    batch_first += fill;
    c->ring_fill = 0;
    return LBM_OK;
  };

  // lbm_run_profiled: timing events around the launches of the first local slab
  auto mark = [&](const Slab &s, int which, hipStream_t st) -> int {
    if (c->prof_sets < 0 || c->prof_sets >= kProfSets || &s != &c->slabs[0]) return LBM_OK;
    HIP_TRY(hipEventRecord(c->prof_ev[(size_t)c->prof_sets * kProfEvents + which], st));
    return LBM_OK;
  };
  int i = 0, set = 0;
  while (i < nsteps) {
    const int src = c->cur;
    // timesteps advanced by this launch set, and with which kernel
    int kind = KIND_SINGLE, adv = 1;
    if (resident) {
      // all remaining steps in one launch, as far as the ring of per-step partial sums reaches
      kind = KIND_RESIDENT;
      adv = std::min(nsteps - i, c->ring);
    } else if (multi_T > 0) {
      kind = KIND_MULTI;
      adv = std::min(multi_T, nsteps - i);
    } else if (deep_lvl && nsteps - i >= 2) {
      // the remaining steps in as few launches as possible, of equal depth (every launch moves the whole grid once:
      // 20 steps = 7+7+6, not 8+8+4)
      kind = KIND_DEEP;
      const int cap = deep_twin ? std::min(fuse_lvl, twin_cap(c)) : fuse_lvl;
      adv = div_up(nsteps - i, div_up(nsteps - i, cap));
    } else if (fuse_lvl == 4 && nsteps - i >= 4) {
      kind = KIND_FUSED4;
      adv = 4;
    } else if (fuse_lvl >= 3 && nsteps - i >= 3) {
      kind = KIND_FUSED3;
      adv = 3;
    } else if (fuse && nsteps - i >= 2) {
      kind = KIND_FUSED2;
      adv = 2;
    }
    const bool last = (i + adv == nsteps);
    const int q = set & 1, qp = q ^ 1;  // event parity of this launch set / of the previous one
    // (test hook: a compact launch set whose exchange is the stale one runs without the fused push; a flags-only launch follows it)
    const bool compact_set = multi && compact && (kind == KIND_MULTI || kind == KIND_FUSED3 || kind == KIND_FUSED4 || kind == KIND_DEEP);
    const bool stale_set = compact_set && !last && stale_exchange_now(c);
    if (batch_kind != kind || c->ring_fill + adv > c->ring)
      if (int rc = flush()) return rc;
    batch_kind = kind;
    for (Slab &s : c->slabs) {
      if (multi && set_dev(s)) return LBM_ERR_HIP;
      float *slot1 = s.partials + (size_t)c->ring_fill * s.nb_total;
      float *slot2 = slot1 + s.nb_total;
      if (!multi) {
        if (int rc = mark(s, 3, s.s_main)) return rc;
        if (kind == KIND_RESIDENT) {
          launch_resident(c, s, src, adv, !last, slot1, s.s_main);
        } else if (kind == KIND_MULTI) {
          MultiArgs a = base_args_multi(c, s, src, adv, !last);
          a.partials = slot1;
          a.ty_begin = 0; a.ty_split = s.m_tiles_y; a.ty_begin2 = 0;
          launch_multi(s, a, s.m_tiles_y, s.s_main);
        } else if (kind == KIND_DEEP) {
          if (deep_twin) launch_deep_twin(c, s, base_args2(c, s, src, !last, s.f6_twin), slot1, adv, s.s_main);
          else launch_deep(c, s, s.f6_main, base_args2(c, s, src, !last, s.f6_main), s.f6_main.units, slot1, adv, s.s_main);
        } else if (kind == KIND_FUSED4) {
          Step2Args a = base_args2(c, s, src, !last, s.f4_main);
          a.partials1 = slot1;
          a.partials2 = slot2;
          float *slot3 = slot2 + s.nb_total;
          launch_step4(c, a, slot3, slot3 + s.nb_total, s.f4_main.units, s.s_main, s.f4_main.paired);
        } else if (kind == KIND_FUSED3) {
          Step2Args a = base_args2(c, s, src, !last, s.f3_main);
          a.partials1 = slot1;
          a.partials2 = slot2;
          launch_step3(c, a, slot2 + s.nb_total, s.f3_main.units, s.s_main, s.f3_main.paired);
        } else if (kind == KIND_FUSED2) {
          Step2Args a = base_args2(c, s, src, !last, s.f_main);
          a.partials1 = slot1;
          a.partials2 = slot2;
          launch_step2(c, a, s.f_main.units, s.s_main);
        } else {
          StepArgs a = base_args(c, s, src, !last);
          a.y_begin = 0; a.y_count = s.rows; a.y_split = s.rows; a.y_begin2 = 0;
          a.partials = slot1;
          launch_step(c, a, s.nb_main, s.s_main);
        }
        HIP_TRY(hipGetLastError());
        if (int rc = mark(s, 4, s.s_main)) return rc;
        continue;
      }
      if (compact && kind == KIND_MULTI) {
        // ---- compact launch set: wait for the neighbours' rows of the latest exchange, then ONE launch of all tile
        // rows, the edge tile rows (tile row 0 and the tile rows from t_top up) first; they push this set's halo rows
        if (c->halo_sync != 2)
          if (int rc = wait_halos(c, s, s.s_main, c->halo_seq)) return rc;
        const int m = s.m_tiles_y;
        const int t_top = std::max(1, std::min(m, (s.rows - s.edge_rows) / s.m_ty));
        const int edge_trows = 1 + (m - t_top);
        MultiArgs a = base_args_multi(c, s, src, adv, !last);
        a.partials = slot1;
        a.ty_begin = 0; a.ty_split = 1; a.ty_begin2 = t_top;     // workgroup rows 1 .. edge_trows-1: tile rows t_top ..
        a.ty_split2 = edge_trows; a.ty_begin3 = 1;               // then the interior tile rows 1 .. t_top-1
        a.edge_blocks = edge_trows * s.m_tiles_x;
        a.peer = s.d_peer;
        if (c->halo_sync == 2) {
          a.peer_mode |= 2;
          a.wait_seq = c->halo_seq;
        }
        if (!last && !stale_set) {
          a.peer_mode |= 1;
          a.peer_buf = src ^ 1;
          a.seq = c->halo_seq + 1;
        }
        if (int rc = mark(s, 3, s.s_main)) return rc;
        launch_multi(s, a, m, s.s_main, true);
        HIP_TRY(hipGetLastError());
        if (stale_set)
          if (int rc = publish_flags_only(c, s, c->halo_seq + 1, s.s_main)) return rc;
        if (int rc = mark(s, 4, s.s_main)) return rc;
        continue;
      }
      if (compact && kind == KIND_DEEP) {
        // ---- compact launch set of d2q9_deep: as for the three- / four-step kernels below
        // (staged: behind the previous set's exchange on the edge stream — its received rows are this launch's halo rows, its
        // sent blocks the ones this launch's edge units overwrite)
        if (staged) HIP_TRY(hipStreamWaitEvent(s.s_main, s.ev_edgek[qp], 0));
        else if (c->halo_sync != 2)
          if (int rc = wait_halos(c, s, s.s_main, c->halo_seq)) return rc;
        Step2Args a = base_args2(c, s, src, !last, s.f6_main);
        a.edge_chunk_start = s.f6_edge.chunk_start;
        a.edge_nchunks = s.f6_edge.nchunks;
        a.edge_units = s.f6_edge.units;
        a.edge_skip = s.f6_edge.skip;
        a.edge_partial_off = s.f6_main.units;
        a.peer = staged ? s.d_stage_peer : s.d_peer;
        if (c->halo_sync == 2 && !staged) {
          a.peer_mode |= 2;
          a.wait_seq = c->halo_seq;
        }
        if (!last && !stale_set) {
          a.peer_mode |= 1;
          a.peer_buf = src ^ 1;
          a.seq = c->halo_seq + 1;
        }
        if (int rc = mark(s, 3, s.s_main)) return rc;
        launch_deep_compact(c, s, a, slot1, adv, s.s_main);
        HIP_TRY(hipGetLastError());
        if (stale_set && !staged)
          if (int rc = publish_flags_only(c, s, c->halo_seq + 1, s.s_main)) return rc;
        if (int rc = mark(s, 4, s.s_main)) return rc;
        continue;
      }
      if (compact && (kind == KIND_FUSED4 || kind == KIND_FUSED3)) {
        // ---- compact launch set of the window kernels: wait for the neighbours' rows of the latest exchange, then ONE
        // launch — the edge chunks first (they push this set's halo rows and raise the flags), then the interior chunks
        if (c->halo_sync != 2)
          if (int rc = wait_halos(c, s, s.s_main, c->halo_seq)) return rc;
        const int level = kind == KIND_FUSED4 ? 4 : 3;
        const FuseGeom &g = level == 4 ? s.f4_main : s.f3_main;
        Step2Args a = base_args2(c, s, src, !last, g);
        a.partials1 = slot1;
        a.partials2 = slot2;
        float *slot3 = slot2 + s.nb_total, *slot4 = slot3 + s.nb_total;
        a.edge_chunk_start = s.f_edge.chunk_start;
        a.edge_nchunks = s.f_edge.nchunks;
        a.edge_units = s.f_edge.units;
        a.edge_skip = s.f_edge.skip;
        a.edge_partial_off = g.units;
        a.peer = s.d_peer;
        if (c->halo_sync == 2) {
          a.peer_mode |= 2;
          a.wait_seq = c->halo_seq;
        }
        if (!last && !stale_set) {
          a.peer_mode |= 1;
          a.peer_buf = src ^ 1;
          a.seq = c->halo_seq + 1;
        }
        if (int rc = mark(s, 3, s.s_main)) return rc;
        launch_compact(level, g.paired, a, slot3, slot4, g.units, s.s_main);
        HIP_TRY(hipGetLastError());
        if (stale_set)
          if (int rc = publish_flags_only(c, s, c->halo_seq + 1, s.s_main)) return rc;
        if (int rc = mark(s, 4, s.s_main)) return rc;
        continue;
      }
      // ---- slab mode: edge rows first (they feed the neighbours), interior meanwhile ----
      // (a compact run gets here with the leftover steps at its end only: no exchange follows, both launches go to
      // the main stream one after the other)
      const hipStream_t s_edge = compact ? s.s_main : s.s_edge;
      // edge launch: needs the previous set's halos (edge stream order; with the peer transport the neighbours'
      // pushes of the latest exchange, announced in this slab's flag words) and interior (event)
      if (c->transport_eff == TRANSPORT_PEER)
        if (int rc = wait_halos(c, s, s_edge, c->halo_seq)) return rc;
      if (staged) HIP_TRY(hipStreamWaitEvent(s.s_main, s.ev_edgek[qp], 0));  // (the previous set's exchange ran on the edge stream)
      if (!compact) {
        HIP_TRY(hipStreamWaitEvent(s.s_edge, s.ev_main[qp], 0));
        // interior launch: needs the previous set's edge rows
        HIP_TRY(hipStreamWaitEvent(s.s_main, s.ev_edgek[qp], 0));
      }
      if (int rc = mark(s, 0, s_edge)) return rc;
      if (!compact)
        if (int rc = mark(s, 3, s.s_main)) return rc;
      if (kind == KIND_MULTI) {
        // edge = the tile rows that hold the halo_depth bottom and top rows (what the neighbours receive):
        // tile row 0 and the tile rows from t_top up; interior = tile rows 1 .. t_top-1
        const int m = s.m_tiles_y;
        const int t_top = std::max(1, std::min(m, (s.rows - s.edge_rows) / s.m_ty));
        const int edge_trows = 1 + (m - t_top), int_trows = t_top - 1;
        MultiArgs e = base_args_multi(c, s, src, adv, !last);
        e.partials = slot1 + (size_t)int_trows * s.m_tiles_x;
        e.ty_begin = 0; e.ty_split = 1; e.ty_begin2 = t_top;
        launch_multi(s, e, edge_trows, s_edge);
        HIP_TRY(hipGetLastError());
        if (int_trows > 0) {
          MultiArgs mm = base_args_multi(c, s, src, adv, !last);
          mm.partials = slot1;
          mm.ty_begin = 1; mm.ty_split = int_trows; mm.ty_begin2 = 0;
          launch_multi(s, mm, int_trows, s.s_main);
          HIP_TRY(hipGetLastError());
        }
      } else if (kind == KIND_DEEP) {
        Step2Args e = base_args2(c, s, src, !last, s.f6_edge);
        e.skip_chunk = s.f6_edge.skip;  // chunk table {bottom edge rows, (interior), top edge rows}
        launch_deep(c, s, s.f6_edge, e, s.f6_edge.units, slot1 + s.f6_main.units, adv, s_edge);
        HIP_TRY(hipGetLastError());
        if (s.f6_main.units > 0) {
          launch_deep(c, s, s.f6_main, base_args2(c, s, src, !last, s.f6_main), s.f6_main.units, slot1, adv, s.s_main);
          HIP_TRY(hipGetLastError());
        }
      } else if (kind == KIND_FUSED4) {
        float *slot3 = slot2 + s.nb_total, *slot4 = slot3 + s.nb_total;
        Step2Args e = base_args2(c, s, src, !last, s.f_edge);
        e.skip_chunk = s.f_edge.skip;  // chunk table {bottom edge rows, (interior), top edge rows}
        e.partials1 = slot1 + s.f4_main.units;
        e.partials2 = slot2 + s.f4_main.units;
        launch_step4(c, e, slot3 + s.f4_main.units, slot4 + s.f4_main.units, s.f_edge.units, s_edge, s.f4_main.paired);
        HIP_TRY(hipGetLastError());
        if (s.f4_main.units > 0) {
          Step2Args m = base_args2(c, s, src, !last, s.f4_main);
          m.partials1 = slot1;
          m.partials2 = slot2;
          launch_step4(c, m, slot3, slot4, s.f4_main.units, s.s_main, s.f4_main.paired);
          HIP_TRY(hipGetLastError());
        }
      } else if (kind == KIND_FUSED3) {
        float *slot3 = slot2 + s.nb_total;
        Step2Args e = base_args2(c, s, src, !last, s.f_edge);
        e.skip_chunk = s.f_edge.skip;  // chunk table {bottom edge rows, (interior), top edge rows}
        e.partials1 = slot1 + s.f3_main.units;
        e.partials2 = slot2 + s.f3_main.units;
        launch_step3(c, e, slot3 + s.f3_main.units, s.f_edge.units, s_edge, s.f3_main.paired);
        HIP_TRY(hipGetLastError());
        if (s.f3_main.units > 0) {
          Step2Args m = base_args2(c, s, src, !last, s.f3_main);
          m.partials1 = slot1;
          m.partials2 = slot2;
          launch_step3(c, m, slot3, s.f3_main.units, s.s_main, s.f3_main.paired);
          HIP_TRY(hipGetLastError());
        }
      } else if (kind == KIND_FUSED2) {
        Step2Args e = base_args2(c, s, src, !last, s.f_edge);
        e.skip_chunk = s.f_edge.skip;  // chunk table {bottom edge rows, (interior), top edge rows}
        e.partials1 = slot1 + s.f_main.units;
        e.partials2 = slot2 + s.f_main.units;
        launch_step2(c, e, s.f_edge.units, s_edge);
        HIP_TRY(hipGetLastError());
        if (s.f_main.units > 0) {
          Step2Args m = base_args2(c, s, src, !last, s.f_main);
          m.partials1 = slot1;
          m.partials2 = slot2;
          launch_step2(c, m, s.f_main.units, s.s_main);
          HIP_TRY(hipGetLastError());
        }
      } else {
        StepArgs e = base_args(c, s, src, !last);
        e.y_begin = s.row0; e.y_count = 2 * s.edge_rows; e.y_split = s.edge_rows; e.y_begin2 = s.row0 + s.rows - s.edge_rows;
        e.partials = slot1 + s.nb_main;
        launch_step(c, e, s.nb_edge, s_edge);
        HIP_TRY(hipGetLastError());
        if (s.rows > 2 * s.edge_rows) {
          StepArgs m = base_args(c, s, src, !last);
          m.y_begin = s.row0 + s.edge_rows; m.y_count = s.rows - 2 * s.edge_rows; m.y_split = m.y_count; m.y_begin2 = 0;
          m.partials = slot1;
          launch_step(c, m, s.nb_main, s.s_main);
          HIP_TRY(hipGetLastError());
        } else {
          HIP_TRY(hipMemsetAsync(slot1, 0, sizeof(float) * s.nb_main, s.s_main));
        }
      }
      if (!compact) {
        HIP_TRY(hipEventRecord(s.ev_edgek[q], s.s_edge));
        HIP_TRY(hipEventRecord(s.ev_main[q], s.s_main));
      }
      if (int rc = mark(s, 1, s_edge)) return rc;
      if (int rc = mark(s, 4, s.s_main)) return rc;
    }
    if (compact_set && staged) {
      // the exchange of a staged set: the edge stream waits for the words the set's last edge wave raises, then RCCL sends the
      // staging blocks and receives into the halo rows of the grid the set has written — beside the set's interior units
      if (!last) {
        c->halo_seq++;
        if (!stale_set) {  // (test hook: no push, no exchange — the next set reads what its halo rows held before)
          for (Slab &s : c->slabs) {
            if (set_dev(s)) return LBM_ERR_HIP;
            HIP_TRY(hipStreamWaitValue32(s.s_edge, s.halo_flags + 4, c->halo_seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
            HIP_TRY(hipStreamWaitValue32(s.s_edge, s.halo_flags + 5, c->halo_seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
          }
          if (int rc = exchange_halos(c, src ^ 1, q, false, true)) return rc;
        }
        for (Slab &s : c->slabs) {
          if (set_dev(s)) return LBM_ERR_HIP;
          HIP_TRY(hipEventRecord(s.ev_edgek[q], s.s_edge));
        }
      }
      if (int rc = mark(c->slabs[0], 2, c->slabs[0].s_edge)) return rc;
    } else if (compact_set) {
      if (!last) c->halo_seq++;     // the edge units of this set's launches have pushed exchange number halo_seq
    } else if (multi) {
      if (!last)
        if (int rc = exchange_halos(c, src ^ 1, q, compact)) return rc;
      if (int rc = mark(c->slabs[0], 2, compact ? c->slabs[0].s_main : c->slabs[0].s_edge)) return rc;
    }
    if (c->prof_sets >= 0) c->prof_sets++;
    c->cur ^= 1;
    c->ring_fill += adv;
    i += adv;
    set++;
    last_q = q;
  }
  if (int rc = flush()) return rc;
  c->steps_done += nsteps;

  if (timed) {
    float worst = 0.0f;
    for (Slab &s : c->slabs) {
      if (set_dev(s)) return LBM_ERR_HIP;
      HIP_TRY(hipEventRecord(s.ev_t1, s.s_main));
    }
    for (Slab &s : c->slabs) {
      if (set_dev(s)) return LBM_ERR_HIP;
      HIP_TRY(hipEventSynchronize(s.ev_t1));
      float t = 0.0f;
      HIP_TRY(hipEventElapsedTime(&t, s.ev_t0, s.ev_t1));
      worst = std::max(worst, t);
    }
    if (ms) *ms = worst;
  }
  return LBM_OK;
}

int sync_all(lbm_ctx *c) {
  for (Slab &s : c->slabs) {
    if (set_dev(s)) return LBM_ERR_HIP;
    if (s.s_edge) HIP_TRY(hipStreamSynchronize(s.s_edge));
    HIP_TRY(hipStreamSynchronize(s.s_main));
    if (s.res_err && s.res_seq != 0) {
      unsigned err = 0;
      HIP_TRY(hipMemcpy(&err, s.res_err, sizeof err, hipMemcpyDeviceToHost));
      if (err) {
        c->failed = true;
        return fail(LBM_ERR_COMM, "d2q9_resident: a band waited %.3g s for its neighbour's rows (the workgroups were not all resident?)",
                    (double)c->halo_timeout_ms * 1e-3);
      }
    }
    if (c->transport_eff == TRANSPORT_PEER && s.halo_flags) {
      uint32_t err = 0;
      HIP_TRY(hipMemcpy(&err, s.halo_flags + 2, sizeof err, hipMemcpyDeviceToHost));
      if (err) {
        c->failed = true;
        if (err & 2u) return fail(LBM_ERR_COMM, "peer transport: slab %d computed a halo row outside its edge rows (internal error)", s.index);
        return fail(LBM_ERR_COMM, "peer transport: slab %d waited %.3g s for a neighbour's halo rows that never came", s.index,
                    (double)c->halo_timeout_ms * 1e-3);
      }
    }
  }
  return LBM_OK;
}

// A failure after launches have begun leaves kernels, events and possibly a half-built exchange behind: let what was
// enqueued finish (so that lbm_destroy's stream synchronisation returns) and refuse further work on the context.
int run_steps(lbm_ctx *c, int nsteps, bool timed, double *ms) {
  bool launched = false;
  const int rc = run_steps_impl(c, nsteps, timed, ms, &launched);
  if (rc != LBM_OK && launched) {
    const std::string keep = g_err;
    for (Slab &s : c->slabs) {
      hipSetDevice(s.dev);
      if (s.s_edge) (void)hipStreamSynchronize(s.s_edge);
      if (s.s_main) (void)hipStreamSynchronize(s.s_main);
    }
    (void)hipGetLastError();
    c->failed = true;
    g_err = keep;
  }
  return rc;
}

void free_slab(Slab &s) {
  hipSetDevice(s.dev);
  for (int i = 0; i < 2; i++) {
    if (s.cells[i]) hipFree(s.cells[i]);
    if (s.ev_main[i]) hipEventDestroy(s.ev_main[i]);
    if (s.ev_edgek[i]) hipEventDestroy(s.ev_edgek[i]);
  }
  if (s.mask) hipFree(s.mask);
  if (s.clean_bits) hipFree(s.clean_bits);
  if (s.partials) hipFree(s.partials);
  if (s.av_sum) hipFree(s.av_sum);
  if (s.fin_partials) hipFree(s.fin_partials);
  if (s.f_main.chunk_start) hipFree(s.f_main.chunk_start);
  if (s.f_edge.chunk_start) hipFree(s.f_edge.chunk_start);
  if (s.f3_main.chunk_start) hipFree(s.f3_main.chunk_start);
  if (s.f4_main.chunk_start) hipFree(s.f4_main.chunk_start);
  if (s.f6_main.chunk_start) hipFree(s.f6_main.chunk_start);
  if (s.f6_edge.chunk_start) hipFree(s.f6_edge.chunk_start);
  if (s.f6_twin.chunk_start) hipFree(s.f6_twin.chunk_start);
  free_balance(s.f6_main);
  free_balance(s.f6_twin);
  if (s.ev_t0) hipEventDestroy(s.ev_t0);
  if (s.ev_t1) hipEventDestroy(s.ev_t1);
  if (s.ev_aux) hipEventDestroy(s.ev_aux);
  for (PeerLink *l : {&s.south, &s.north}) {
    if (!l->ipc) continue;
    for (int i = 0; i < 2; i++)
      if (l->cells[i]) hipIpcCloseMemHandle(l->cells[i]);
    if (l->flags) hipIpcCloseMemHandle(l->flags);
  }
  if (s.halo_flags) hipFree(s.halo_flags);
  if (s.res_words) hipFree(s.res_words);
  if (s.res_xrows) hipFree(s.res_xrows);
  if (s.res_err) hipFree(s.res_err);
  s.res_words = nullptr;
  s.res_xrows = nullptr;
  s.res_err = nullptr;
  for (int k = 0; k < 2; k++)
    if (s.stage[k]) hipFree(s.stage[k]);
  if (s.d_stage_peer) hipFree(s.d_stage_peer);
  s.stage[0] = s.stage[1] = nullptr;
  s.d_stage_peer = nullptr;
  if (s.d_peer) hipFree(s.d_peer);
  if (s.av_tmp) hipFree(s.av_tmp);
  if (s.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(s.comm);
  if (s.s_edge) hipStreamDestroy(s.s_edge);
  if (s.s_main) hipStreamDestroy(s.s_main);
  s = Slab{};
}

int check_params(const lbm_params *p) {
  if (!p) return fail(LBM_ERR_ARG, "params is NULL");
  if (p->nx < 3 || p->ny < 3) return fail(LBM_ERR_ARG, "grid must be at least 3x3 (got %dx%d)", p->nx, p->ny);
  if ((long)p->nx * p->ny > (1L << 31) - 1) return fail(LBM_ERR_ARG, "grid too large: %dx%d", p->nx, p->ny);
  if (p->max_iters < 0) return fail(LBM_ERR_ARG, "max_iters must be >= 0");
  return LBM_OK;
}

// Allocate and fill one slab (rows [y0, y0+rows) of the global grid, plus halo rows in slab mode).
// the slab's byte mask from the caller's int32[ny][nx] (d2q9-bgk.c:205-209: the obstacle transfer)
int upload_mask(const lbm_ctx *c, Slab &s, const int32_t *obstacles) {
  const int nx = c->p.nx, ny = c->p.ny;
  const size_t n_ext = (size_t)nx * s.ext_rows;
  // stored row e holds global row (y0 - row0 + e) mod ny: halo rows carry the neighbours' obstacle flags
  std::vector<uint8_t> m(n_ext);
  for (int e = 0; e < s.ext_rows; e++) {
    const int gy = ((s.y0 - s.row0 + e) % ny + ny) % ny;
    const int32_t *src = obstacles + (size_t)gy * nx;
    for (int x = 0; x < nx; x++) m[(size_t)e * nx + x] = src[x] != 0;
  }
  if (set_dev(s)) return LBM_ERR_HIP;
  HIP_TRY(hipMemcpy(s.mask, m.data(), n_ext, hipMemcpyHostToDevice));
  return LBM_OK;
}

int build_slab(lbm_ctx *c, Slab &s, const int32_t *obstacles) {
  const int nx = c->p.nx, ny = c->p.ny;
  const bool multi = c->halo_mode;
  if (set_dev(s)) return LBM_ERR_HIP;
  {
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s.dev));
    if (cus > 0) s.cus = cus;
  }
  HIP_TRY(hipStreamCreateWithFlags(&s.s_main, hipStreamNonBlocking));
  if (multi) {
    // the edge launch and the halo exchange are the critical path of a launch set (the neighbours wait for
    // them) while the interior launch fills the machine: give the edge stream the highest priority
    int prio_low = 0, prio_high = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    HIP_TRY(hipStreamCreateWithPriority(&s.s_edge, hipStreamNonBlocking, prio_high));
  }
  for (int i = 0; i < 2; i++) {
    HIP_TRY(hipEventCreateWithFlags(&s.ev_main[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s.ev_edgek[i], hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreate(&s.ev_t0));
  HIP_TRY(hipEventCreate(&s.ev_t1));
  HIP_TRY(hipEventCreateWithFlags(&s.ev_aux, hipEventDisableTiming));
  s.row0 = multi ? c->halo_depth : 0;
  s.ext_rows = s.rows + 2 * s.row0;
  // row-interleaved SoA (see d2q9_kernels.h): plane-rows padded to whole 256-B lines
  s.plane_stride = ((size_t)(nx + 63) / 64) * 64;
  s.row_stride = 9 * s.plane_stride;
  for (int i = 0; i < 2; i++) {
    if (dev_alloc(&s.cells[i], s.row_stride * s.ext_rows + 64)) return LBM_ERR_HIP;
    HIP_TRY(hipMemset(s.cells[i], 0, (s.row_stride * s.ext_rows + 64) * sizeof(float)));
  }
  const size_t n_ext = (size_t)nx * s.ext_rows;
  if (dev_alloc(&s.mask, n_ext + 64)) return LBM_ERR_HIP;
  if (int rc = upload_mask(c, s, obstacles)) return rc;
  // the accelerated row ny-2 (kernels.cl:18) in stored-row coordinates
  s.accel_own = s.accel_ext = s.accel_ext_b = -1;
  for (int e = 0; e < s.ext_rows; e++) {
    const int gy = ((s.y0 - s.row0 + e) % ny + ny) % ny;
    if (gy != ny - 2) continue;
    if (e >= s.row0 && e < s.row0 + s.rows) s.accel_own = e;
    if (s.accel_ext < 0) s.accel_ext = e;
    else s.accel_ext_b = e;
  }
  if (dev_alloc(&s.av_sum, (size_t)std::max(1, c->p.max_iters))) return LBM_ERR_HIP;
  if (multi) {
    if (dev_alloc(&s.halo_flags, 64)) return LBM_ERR_HIP;
    HIP_TRY(hipMemset(s.halo_flags, 0, 64 * sizeof(uint32_t)));
    HIP_TRY(hipDeviceGetAttribute(&s.can_wait_value, hipDeviceAttributeCanUseStreamWaitValue, s.dev));
  }
  s.fin_blocks = std::max(1, std::min(div_up((long)nx * s.rows, kBlock), 2048));
  if (dev_alloc(&s.fin_partials, (size_t)s.fin_blocks)) return LBM_ERR_HIP;
  return LBM_OK;
}

// (Re)allocate the ring of per-workgroup partial sums: [ring][nb_total] floats per slab, at most 16 MiB.
int alloc_partials(lbm_ctx *c) {
  int max_nb = 1;
  for (Slab &s : c->slabs) max_nb = std::max(max_nb, s.nb_total);
  c->ring = (int)std::max<long>(8, std::min<long>(kRingMax, (4L << 20) / max_nb));
  for (Slab &s : c->slabs) {
    if (set_dev(s)) return LBM_ERR_HIP;
    if (s.partials) HIP_TRY(hipFree(s.partials));
    s.partials = nullptr;
    if (dev_alloc(&s.partials, (size_t)c->ring * s.nb_total)) return LBM_ERR_HIP;
  }
  c->ring_fill = 0;
  return LBM_OK;
}

int upload_multi_peer(const lbm_ctx *c, Slab &s);
// staged launch sets: the slab's staging blocks (allocated on first use) and the device-side description that makes the edge units of
// a PUSH kernel store into them and raise halo_flags[4], [5]
int upload_stage_peer(const lbm_ctx *c, Slab &s) {
  if (set_dev(s)) return LBM_ERR_HIP;
  const size_t count = (size_t)s.row0 * s.row_stride;
  for (int k = 0; k < 2; k++)
    if (!s.stage[k]) {
      if (dev_alloc(&s.stage[k], count)) return LBM_ERR_HIP;
      HIP_TRY(hipMemset(s.stage[k], 0, count * sizeof(float)));  // (the padding behind nx is never written: sent as zeros)
    }
  HaloPeer h{};
  for (int b = 0; b < 2; b++) {
    h.push[0][b] = s.stage[0];
    h.push[1][b] = s.stage[1];
  }
  h.flag_lo = s.halo_flags + 4;
  h.flag_hi = s.halo_flags + 5;
  h.ticket = s.halo_flags + 3;
  h.wait_flags = s.halo_flags;
  h.wait_err = s.halo_flags + 2;
  h.wait_ticks = c->halo_timeout_ms * kTicksPerMs;
  h.push_rows = s.row0;
  h.row_lo0 = s.row0;
  h.row_hi0 = s.rows;
  h.release = 0;  // write-through stores into this device's own memory, drained before the ticket; the exchange kernel starts behind the flag
  if (!s.d_stage_peer && dev_alloc(&s.d_stage_peer, 1)) return LBM_ERR_HIP;
  HIP_TRY(hipMemcpy(s.d_stage_peer, &h, sizeof h, hipMemcpyHostToDevice));
  return LBM_OK;
}
int rebuild_geometry(lbm_ctx *c) {
  // grid_blocks / chunk changes alter the number of partial sums per step
  for (Slab &s : c->slabs) {
    if (set_dev(s)) return LBM_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(s.s_main));
    if (int rc = slab_geometry(c, s)) return rc;
    // (which rows the pushing kernels store where depends on the kernel the options select: slab_twin5)
    if (s.d_peer && s.south.connected && s.north.connected)
      if (int rc = upload_multi_peer(c, s)) return rc;
  }
  if (staged_sets(c))
    for (Slab &s : c->slabs)
      if (int rc = upload_stage_peer(c, s)) return rc;
  return alloc_partials(c);
}

// ---- peer-halo transport: connecting the ring -------------------------------------------------------
void device_pci(int dev, int32_t (&pci)[4]);
void fill_peer_info(const Slab &s, PeerInfoBlob &b, bool with_ipc, hipError_t *ipc_err) {
  memset(&b, 0, sizeof b);
  b.magic = kPeerMagic;
  b.pid = (int32_t)getpid();
  b.nonce[0] = process_nonce()[0];
  b.nonce[1] = process_nonce()[1];
  device_pci(s.dev, b.pci);
  b.device = s.dev;
  b.rows = s.rows;
  b.row0 = s.row0;
  b.row_stride = s.row_stride;
  b.cells_ptr[0] = (uint64_t)(uintptr_t)s.cells[0];
  b.cells_ptr[1] = (uint64_t)(uintptr_t)s.cells[1];
  b.flags_ptr = (uint64_t)(uintptr_t)s.halo_flags;
  if (with_ipc) {
    hipError_t e = hipIpcGetMemHandle(&b.cells[0], s.cells[0]);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&b.cells[1], s.cells[1]);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&b.flags, s.halo_flags);
    if (ipc_err) *ipc_err = e;
  }
}

void device_pci(int dev, int32_t (&pci)[4]) {
  int v[3] = {-1, -1, -1};
  if (hipDeviceGetAttribute(&v[0], hipDeviceAttributePciDomainID, dev) != hipSuccess) v[0] = -1;
  if (hipDeviceGetAttribute(&v[1], hipDeviceAttributePciBusId, dev) != hipSuccess) v[1] = -1;
  if (hipDeviceGetAttribute(&v[2], hipDeviceAttributePciDeviceId, dev) != hipSuccess) v[2] = -1;
  (void)hipGetLastError();
  pci[0] = v[0]; pci[1] = v[1]; pci[2] = v[2]; pci[3] = 0;
}

bool same_process(const PeerInfoBlob &info) {
  return info.nonce[0] == process_nonce()[0] && info.nonce[1] == process_nonce()[1] && info.pid == (int32_t)getpid();
}

// unmap what connect_link opened through HIP IPC (raw-pointer links own nothing)
void close_link(PeerLink &l) {
  if (l.ipc) {
    if (l.cells[0]) hipIpcCloseMemHandle(l.cells[0]);
    if (l.cells[1]) hipIpcCloseMemHandle(l.cells[1]);
    if (l.flags) hipIpcCloseMemHandle(l.flags);
    (void)hipGetLastError();
  }
  l = PeerLink{};
}

int connect_link(Slab &s, PeerLink &l, const PeerInfoBlob &info, const char *which) {
  if (info.magic != kPeerMagic) return fail(LBM_ERR_ARG, "%s neighbour: not a peer descriptor of this library version", which);
  if (info.row_stride != s.row_stride || info.row0 != s.row0)
    return fail(LBM_ERR_ARG, "%s neighbour stores its rows differently (row stride %llu / %llu floats, halo depth %d / %d): all ranks must "
                "be created with the same params and defaults", which, (unsigned long long)info.row_stride,
                (unsigned long long)s.row_stride, info.row0, s.row0);
  if (set_dev(s)) return LBM_ERR_HIP;
  l = PeerLink{};
  l.rows = info.rows;
  if (same_process(info)) {
    // same process (nonce, not just the pid): the pointers are valid here; another device needs peer access
    if (info.device != s.dev) {
      int can = 0;
      HIP_TRY(hipDeviceCanAccessPeer(&can, s.dev, info.device));
      if (!can) return fail(LBM_ERR_COMM, "device %d cannot access device %d as a peer", s.dev, info.device);
      hipError_t e = hipDeviceEnablePeerAccess(info.device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return fail(LBM_ERR_HIP, "hipDeviceEnablePeerAccess(%d) from device %d: %s", info.device, s.dev, hipGetErrorString(e));
      (void)hipGetLastError();
    }
    l.cells[0] = (float *)(uintptr_t)info.cells_ptr[0];
    l.cells[1] = (float *)(uintptr_t)info.cells_ptr[1];
    l.flags = (uint32_t *)(uintptr_t)info.flags_ptr;
  } else {
    void *p0 = nullptr, *p1 = nullptr, *pf = nullptr;
    // One attempt each, and the first error is the one reported.  (Round 3 retried a failed open four times on a guess; the
    // failures it was meant to ride out came from a rank freeing memory its neighbour still had mapped, which the tear-down
    // order of lbm_disconnect_peers -> barrier -> lbm_destroy removed.  A retry would now only hide a real failure.)
    hipError_t e = hipIpcOpenMemHandle(&p0, info.cells[0], hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) e = hipIpcOpenMemHandle(&p1, info.cells[1], hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) e = hipIpcOpenMemHandle(&pf, info.flags, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      if (p0) hipIpcCloseMemHandle(p0);
      if (p1) hipIpcCloseMemHandle(p1);
      (void)hipGetLastError();
      return fail(LBM_ERR_COMM, "hipIpcOpenMemHandle of the %s neighbour's grids (process %d, device %d): %s", which, info.pid,
                  info.device, hipGetErrorString(e));
    }
    l.cells[0] = (float *)p0;
    l.cells[1] = (float *)p1;
    l.flags = (uint32_t *)pf;
    l.ipc = true;
  }
  {
    // the neighbour's grids live on another GPU unless its PCI address is this slab's (same node: the boot id is in the nonce
    // only for processes, so compare addresses only; a same-process neighbour is also recognised by its device index)
    int32_t mine[4];
    device_pci(s.dev, mine);
    const bool same_gpu = same_process(info) ? info.device == s.dev
                                             : (mine[1] >= 0 && mine[0] == info.pci[0] && mine[1] == info.pci[1] && mine[2] == info.pci[2]);
    l.remote = !same_gpu;
  }
  l.connected = true;
  return LBM_OK;
}

// device-side description of a connected slab's neighbours for the fused push / wait of d2q9_multi
int upload_multi_peer(const lbm_ctx *c, Slab &s) {
  HaloPeer h{};
  h.release = push_release_effective(c, s);
  for (int b = 0; b < 2; b++) {
    h.push[0][b] = s.south.cells[b] + (size_t)(s.row0 + s.south.rows) * s.row_stride;
    h.push[1][b] = s.north.cells[b];
  }
  h.flag_lo = s.south.flags + 1;  // this slab is the south neighbour's NORTH neighbour
  h.flag_hi = s.north.flags + 0;
  h.ticket = s.halo_flags + 3;
  h.wait_flags = s.halo_flags;
  h.wait_err = s.halo_flags + 2;
  h.wait_ticks = c->halo_timeout_ms * kTicksPerMs;
  h.push_rows = s.row0;
  h.row_lo0 = s.row0;      // bottom edge rows [row0, 2 row0) -> the south neighbour's top halo rows
  h.row_hi0 = s.rows;      // top edge rows [rows, rows + row0) -> the north neighbour's bottom halo rows
  if (slab_twin5(c) && s.row0 > kDeepTwinDefault) {
    // the five-step chunk pairs on slabs that store more halo rows than they exchange: the five rows next to the owned ones
    h.push_rows = kDeepTwinDefault;
    h.row_hi0 = s.row0 + s.rows - kDeepTwinDefault;
    for (int b = 0; b < 2; b++) h.push[1][b] = s.north.cells[b] + (size_t)(s.row0 - kDeepTwinDefault) * s.row_stride;
  }
  if (set_dev(s)) return LBM_ERR_HIP;
  if (!s.d_peer && dev_alloc(&s.d_peer, 1)) return LBM_ERR_HIP;
  HIP_TRY(hipMemcpy(s.d_peer, &h, sizeof h, hipMemcpyHostToDevice));
  return LBM_OK;
}

// one process: every slab's neighbours are local slabs
int connect_local_ring(lbm_ctx *c) {
  const int P = (int)c->slabs.size();
  for (Slab &s : c->slabs) {
    PeerInfoBlob so, no;
    fill_peer_info(c->slabs[(s.index + P - 1) % P], so, false, nullptr);
    fill_peer_info(c->slabs[(s.index + 1) % P], no, false, nullptr);
    if (int rc = connect_link(s, s.south, so, "south")) return rc;
    if (int rc = connect_link(s, s.north, no, "north")) return rc;
    if (int rc = upload_multi_peer(c, s)) return rc;
  }
  return LBM_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int lbm_set_default(const char *key, long value) {
  if (!key) return fail(LBM_ERR_ARG, "NULL argument");
  if (!strcmp(key, "force_halo")) {
    if (value != 0 && value != 1) return fail(LBM_ERR_ARG, "force_halo must be 0 or 1");
    g_defaults.force_halo = (int)value;
  } else if (!strcmp(key, "halo_depth")) {
    if (value != 0 && (value < 2 || value > kMultiMaxT)) return fail(LBM_ERR_ARG, "halo_depth must be 0 (auto) or 2..%d", kMultiMaxT);
    g_defaults.halo_depth = (int)value;
  } else if (!strcmp(key, "transport")) {
    if (value < TRANSPORT_AUTO || value > TRANSPORT_PEER) return fail(LBM_ERR_ARG, "transport must be 0 (auto), 1 (rccl), 2 (copy) or 3 (peer)");
    g_defaults.transport = (int)value;
  } else if (!strcmp(key, "lanes_out")) {
    if (value != 0 && (value < 4 || value > 62)) return fail(LBM_ERR_ARG, "lanes_out must be 0 (auto) or 4..62");
    g_defaults.lanes_out = (int)value;
  } else {
    return fail(LBM_ERR_ARG, "unknown default '%s'", key);
  }
  return LBM_OK;
}

size_t lbm_peer_info_size(void) { return sizeof(PeerInfoBlob); }

int lbm_peer_info(lbm_ctx *c, void *info_out) {
  if (!c || !info_out) return fail(LBM_ERR_ARG, "NULL argument");
  if (!c->halo_mode || c->slabs.size() != 1)
    return fail(LBM_ERR_STATE, "peer descriptors exist for rank contexts (one local slab that exchanges halo rows)");
  Slab &s = c->slabs[0];
  if (set_dev(s)) return LBM_ERR_HIP;
  PeerInfoBlob b;
  hipError_t e = hipSuccess;
  fill_peer_info(s, b, true, &e);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(LBM_ERR_HIP, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
  }
  memcpy(info_out, &b, sizeof b);
  return LBM_OK;
}

int lbm_connect_peers(lbm_ctx *c, const void *south_info, const void *north_info) {
  if (!c || !south_info || !north_info) return fail(LBM_ERR_ARG, "NULL argument");
  if (!c->halo_mode || c->slabs.size() != 1) return fail(LBM_ERR_STATE, "lbm_connect_peers is for rank contexts");
  if (int rc = sync_all(c)) return rc;
  Slab &s = c->slabs[0];
  PeerInfoBlob so, no;
  memcpy(&so, south_info, sizeof so);
  memcpy(&no, north_info, sizeof no);
  // a second call replaces the links: unmap the old ones first (north may alias south's mappings: ipc is false there)
  close_link(s.north);
  close_link(s.south);
  PeerLink south, north;
  if (int rc = connect_link(s, south, so, "south")) return rc;
  if (so.nonce[0] == no.nonce[0] && so.nonce[1] == no.nonce[1] && so.pid == no.pid && so.cells_ptr[0] == no.cells_ptr[0]) {
    north = south;      // a ring of one or two: the same neighbour on both sides, mapped once
    north.ipc = false;
  } else if (int rc = connect_link(s, north, no, "north")) {
    const std::string keep = g_err;
    close_link(south);  // do not leave the south neighbour's grids mapped behind a failed call
    g_err = keep;
    return rc;
  }
  s.south = south;
  s.north = north;
  if (int rc = upload_multi_peer(c, s)) return rc;
  // Which transport runs after a successful connect: peer stores when they are the only one there is (no communicator)
  // or when the caller asked for them (default "transport" = 3); a context that also holds an RCCL communicator stays on
  // RCCL send/recv until lbm_set_option("transport", 3) — the caller's explicit decision, e.g. after it has checked the
  // peer ring against the RCCL ring on its machine (bench.py does: transport_check).
  if (!s.comm || g_defaults.transport == TRANSPORT_PEER) c->transport_eff = TRANSPORT_PEER;
  if (c->halo_sync == 2 && (s.south.remote || s.north.remote)) c->halo_sync = 0;  // see lbm_set_option("halo_sync")
  return rebuild_geometry(c);
}

int lbm_disconnect_peers(lbm_ctx *c) {
  if (!c) return fail(LBM_ERR_ARG, "ctx is NULL");
  if (!c->halo_mode || c->slabs.size() != 1) return fail(LBM_ERR_STATE, "lbm_disconnect_peers is for rank contexts");
  // let the work that is queued finish (an error word raised by a dead neighbour must not keep the mappings open)
  for (Slab &s : c->slabs) {
    hipSetDevice(s.dev);
    if (s.s_edge) (void)hipStreamSynchronize(s.s_edge);
    if (s.s_main) (void)hipStreamSynchronize(s.s_main);
  }
  (void)hipGetLastError();
  Slab &s = c->slabs[0];
  close_link(s.north);
  close_link(s.south);
  if (c->transport_eff == TRANSPORT_PEER) c->transport_eff = s.comm ? TRANSPORT_RCCL : TRANSPORT_AUTO;
  return LBM_OK;
}

const char *lbm_last_error(void) { return g_err.c_str(); }
#ifndef LBM_SRC_ID
#define LBM_SRC_ID "unknown"
#endif
const char *lbm_version(void) { return "lbm-hip 0.3 (gfx950) src " LBM_SRC_ID; }

size_t lbm_comm_id_size(void) { return sizeof(ncclUniqueId); }

int lbm_comm_get_id(void *comm_id_out) {
  if (!comm_id_out) return fail(LBM_ERR_ARG, "comm_id_out is NULL");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(comm_id_out, &id, sizeof id);
  return LBM_OK;
}

static int create_common(lbm_ctx **out, const lbm_params *params, const int32_t *obstacles, int nslabs_global,
                         const std::vector<int> &slab_indices, const std::vector<int> &devs, bool rank_mode, int rank,
                         const void *comm_id) {
  if (!out) return fail(LBM_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (int rc = check_params(params)) return rc;
  if (!obstacles) return fail(LBM_ERR_ARG, "obstacles is NULL");
  if (nslabs_global < 1) return fail(LBM_ERR_ARG, "need at least one slab");
  if ((nslabs_global > 1 || g_defaults.force_halo) && params->ny / nslabs_global < 4)
    return fail(LBM_ERR_ARG, "ny=%d gives fewer than 4 rows per slab over %d slabs", params->ny, nslabs_global);
  int ndev_visible = 0;
  HIP_TRY(hipGetDeviceCount(&ndev_visible));
  if (ndev_visible < 1) return fail(LBM_ERR_HIP, "no HIP device visible");
  for (int d : devs)
    if (d < 0 || d >= ndev_visible) return fail(LBM_ERR_ARG, "device index %d out of range (%d visible)", d, ndev_visible);

  lbm_ctx *c = new lbm_ctx();
  c->p = *params;
  c->nslabs_global = nslabs_global;
  // default "force_halo" runs even a single slab through the halo-exchange machinery (a ring of one: the slab
  // is its own north and south neighbour) — lets a 1-GPU box exercise every transport end to end
  c->halo_mode = nslabs_global > 1 || g_defaults.force_halo != 0;
  {
    // halo depth: small slabs are launch-bound and use the LDS multi-step kernel with 8 steps per exchange
    const int rows_min = params->ny / nslabs_global;
    c->rows_min = rows_min;
    const bool small = (long)params->nx * rows_min <= 540L * 1024;
    // ... slabs of 2M cells and more depth 4 (four-steps-per-launch kernel), the others depth 3 (three-step kernel)
    const bool big = (long)params->nx * rows_min >= (2L << 20);
    c->halo_depth = (small && rows_min >= 2 * kMultiMaxT) ? kMultiMaxT : (big ? 4 : (rows_min >= 6 ? 3 : 2));
    // ... and slabs of 5M cells and more depth 8 again: d2q9_deep, up to eight steps per launch set
    if ((long)params->nx * rows_min >= kSlabDeepCells && rows_min >= 4 * kDeepSteps && params->nx % 4 == 0 && params->nx >= 256)
      c->halo_depth = kDeepSteps;
    // ... and slabs between 540K and 3M cells depth 5: five-step chunk pairs in compact launch sets (kSlabTwinCells)
    // (slabs in the LDS tiles' range keep their eight: the pairs use five of them)
    else if (!small && rows_min >= 4 * kDeepTwinDefault && params->nx % 4 == 0 && params->nx >= 256)
      c->halo_depth = kDeepTwinDefault;
    if (g_defaults.halo_depth > 0) c->halo_depth = g_defaults.halo_depth;
    if (rows_min < 2 * c->halo_depth) c->halo_depth = 2;
  }
  c->rank_mode = rank_mode;
  c->rank = rank;
  c->vec4 = (params->nx % 4 == 0);
  c->slabs.resize(slab_indices.size());
  int rc = LBM_OK;
  for (size_t i = 0; i < slab_indices.size() && rc == LBM_OK; i++) {
    Slab &s = c->slabs[i];
    s.dev = devs[i];
    s.index = slab_indices[i];
    split_rows(params->ny, nslabs_global, s.index, &s.y0, &s.rows);
    rc = build_slab(c, s, obstacles);
  }
  for (size_t i = 0; i < c->slabs.size() && rc == LBM_OK; i++) rc = slab_geometry(c, c->slabs[i]);
  if (rc == LBM_OK) rc = alloc_partials(c);
  // transport for halo exchange
  if (rc == LBM_OK && c->halo_mode) {
    bool dup = false;
    for (size_t i = 0; i < devs.size(); i++)
      for (size_t j = i + 1; j < devs.size(); j++) dup |= devs[i] == devs[j];
    const int want = g_defaults.transport;
    const bool whole_ring_local = (int)c->slabs.size() == nslabs_global;
    if (rank_mode && comm_id) {
      // a communicator: RCCL send/recv until (unless) the caller connects the peers; the all-reduce of the velocity
      // record uses it either way
      rc = load_rccl();
      if (rc == LBM_OK) {
        ncclUniqueId id;
        memcpy(&id, comm_id, sizeof id);
        hipSetDevice(c->slabs[0].dev);
        ncclResult_t r = g_rccl.CommInitRank(&c->slabs[0].comm, nslabs_global, id, rank);
        if (r != ncclSuccess) rc = fail(LBM_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
      }
      c->transport_eff = TRANSPORT_RCCL;
      if (rc == LBM_OK && whole_ring_local && want == TRANSPORT_PEER) {
        rc = connect_local_ring(c);
        if (rc == LBM_OK) c->transport_eff = TRANSPORT_PEER;
      }
    } else if (!whole_ring_local) {
      c->transport_eff = TRANSPORT_AUTO;  // rank of a larger ring without communicator: lbm_connect_peers decides
    } else if (want == TRANSPORT_RCCL) {
      if (dup) {
        rc = fail(LBM_ERR_ARG, "RCCL transport needs distinct devices per slab");
      } else {
        rc = load_rccl();
        if (rc == LBM_OK) {
          std::vector<ncclComm_t> comms(devs.size());
          ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
          if (r != ncclSuccess) rc = fail(LBM_ERR_COMM, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
          else
            for (size_t i = 0; i < devs.size(); i++) c->slabs[i].comm = comms[i];
        }
        c->transport_eff = TRANSPORT_RCCL;
      }
    } else if (want == TRANSPORT_COPY) {
      // device-to-device copies: enable peer access between distinct devices
      for (size_t i = 0; i < devs.size(); i++)
        for (size_t j = 0; j < devs.size(); j++)
          if (devs[i] != devs[j]) {
            hipSetDevice(devs[i]);
            hipError_t e = hipDeviceEnablePeerAccess(devs[j], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
          }
      c->transport_eff = TRANSPORT_COPY;
    } else {
      // all slabs in this process: direct stores into the neighbours' halo rows
      rc = connect_local_ring(c);
      c->transport_eff = TRANSPORT_PEER;
    }
  }
  // the chunk schedules depend on the transport (compact launch sets reserve nothing for an edge launch and may pair)
  if (rc == LBM_OK && c->halo_mode) rc = rebuild_geometry(c);
  if (rc == LBM_OK) {
    c->av_host = (double *)malloc(sizeof(double) * (size_t)std::max(1, params->max_iters));
    if (!c->av_host) rc = fail(LBM_ERR_ARG, "out of host memory");
  }
  if (rc != LBM_OK) {
    std::string keep = g_err;
    lbm_destroy(c);
    g_err = keep;
    return rc;
  }
  *out = c;
  // default initial state: uniform rest state on the device
  return lbm_upload(c, nullptr);
}

int lbm_create(lbm_ctx **out, const lbm_params *params, const int32_t *obstacles, int ndev, const int *dev_ids) {
  std::vector<int> idx, devs;
  if (ndev <= 1 && dev_ids == nullptr) {
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    idx.push_back(0);
    devs.push_back(cur);
    ndev = 1;
  } else {
    if (ndev < 1 || !dev_ids) return fail(LBM_ERR_ARG, "ndev=%d needs dev_ids", ndev);
    for (int i = 0; i < ndev; i++) {
      idx.push_back(i);
      devs.push_back(dev_ids[i]);
    }
  }
  return create_common(out, params, obstacles, ndev, idx, devs, false, 0, nullptr);
}

int lbm_create_rank(lbm_ctx **out, const lbm_params *params, const int32_t *obstacles, int rank, int nranks, int device,
                    const void *comm_id) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(LBM_ERR_ARG, "bad rank %d of %d", rank, nranks);
  std::vector<int> idx{rank}, devs{device};
  // with a comm_id the rank gets an RCCL communicator (a single rank too: self-ring tests); without one it has to
  // be connected to its ring neighbours with lbm_connect_peers and returns per-rank velocity sums
  return create_common(out, params, obstacles, nranks, idx, devs, nranks > 1 || comm_id != nullptr, rank, comm_id);
}

int lbm_upload(lbm_ctx *c, const float *cells) {
  if (!c) return fail(LBM_ERR_ARG, "ctx is NULL");
  if (int rc = sync_all(c)) return rc;
  const int nx = c->p.nx, ny = c->p.ny;
  const size_t n_global = (size_t)nx * ny;
  for (Slab &s : c->slabs) {
    if (set_dev(s)) return LBM_ERR_HIP;
    const size_t n = (size_t)nx * s.rows;
    if (cells) {
      // reference SoA plane k (d2q9-bgk.c:73) -> plane-row k of every grid row
      for (int k = 0; k < 9; k++)
        HIP_TRY(hipMemcpy2DAsync(s.own(0) + k * s.plane_stride, s.row_stride * sizeof(float),
                                 cells + k * n_global + (size_t)s.y0 * nx, (size_t)nx * sizeof(float),
                                 (size_t)nx * sizeof(float), s.rows, hipMemcpyHostToDevice, s.s_main));
    } else {
      // d2q9-bgk.c:529-531
      const float w0 = c->p.density * 4.0f / 9.0f, w1 = c->p.density / 9.0f, w2 = c->p.density / 36.0f;
      hipLaunchKernelGGL(init_cells, dim3(std::min(div_up((long)n, 256), 4096)), dim3(256), 0, s.s_main, s.own(0),
                         s.plane_stride, s.row_stride, nx, n, w0, w1, w2);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s.s_main));
  }
  c->cur = 0;
  c->steps_done = 0;
  c->ring_fill = 0;
  return LBM_OK;
}

int lbm_upload_obstacles(lbm_ctx *c, const int32_t *obstacles) {
  if (!c || !obstacles) return fail(LBM_ERR_ARG, "NULL argument");
  if (int rc = sync_all(c)) return rc;
  for (Slab &s : c->slabs)
    if (int rc = upload_mask(c, s, obstacles)) return rc;
  // what the library derives from the map — the deep window kernel's bits of rows with blocked cells per strip, and with them
  // which strips its one-round schedules balance — follows the new map
  return rebuild_geometry(c);
}

int lbm_run(lbm_ctx *c, int nsteps) {
  if (!c) return fail(LBM_ERR_ARG, "ctx is NULL");
  return run_steps(c, nsteps, false, nullptr);
}

int lbm_run_timed(lbm_ctx *c, int nsteps, double *ms) {
  if (!c) return fail(LBM_ERR_ARG, "ctx is NULL");
  return run_steps(c, nsteps, true, ms);
}

int lbm_run_profiled(lbm_ctx *c, int nsteps, double *stats) {
  if (!c || !stats) return fail(LBM_ERR_ARG, "NULL argument");
  for (int i = 0; i < 8; i++) stats[i] = 0.0;
  if (c->slabs.empty()) return fail(LBM_ERR_STATE, "no slab");
  if (set_dev(c->slabs[0])) return LBM_ERR_HIP;
  while (c->prof_ev.size() < (size_t)kProfSets * kProfEvents) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    c->prof_ev.push_back(e);
  }
  if (int rc = sync_all(c)) return rc;
  c->prof_sets = 0;
  const int before = c->steps_done;
  int rc = run_steps(c, nsteps, false, nullptr);
  const int sets_run = c->prof_sets;
  const int sets = std::min(sets_run, kProfSets);
  c->prof_sets = -1;
  if (rc == LBM_OK) rc = sync_all(c);
  if (rc != LBM_OK) return rc;
  if (sets < 1) return LBM_OK;
  auto ev = [&](int set, int which) { return c->prof_ev[(size_t)set * kProfEvents + which]; };
  auto span = [&](hipEvent_t a, hipEvent_t b, double *acc) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess) *acc += ms * 1e3;
    else (void)hipGetLastError();
  };
  double edge = 0, xchg = 0, inner = 0, lag = 0, period = 0;
  for (int k = 0; k < sets; k++) {
    span(ev(k, 3), ev(k, 4), &inner);
    if (c->halo_mode) {
      span(ev(k, 0), ev(k, 1), &edge);
      span(ev(k, 1), ev(k, 2), &xchg);
      span(ev(k, 0), ev(k, 3), &lag);
    }
  }
  if (sets > 1) span(ev(0, 3), ev(sets - 1, 3), &period);
  stats[0] = sets;
  stats[1] = (double)(c->steps_done - before) / std::max(1, sets_run);
  stats[2] = edge / sets;
  stats[3] = xchg / sets;
  stats[4] = inner / sets;
  stats[5] = sets > 1 ? period / (sets - 1) : 0.0;
  stats[6] = lag / sets;
  stats[7] = c->transport_eff;
  return LBM_OK;
}

int lbm_sync(lbm_ctx *c) {
  if (!c) return fail(LBM_ERR_ARG, "ctx is NULL");
  return sync_all(c);
}

int lbm_steps_done(const lbm_ctx *c) { return c ? c->steps_done : -1; }

int lbm_row_range(const lbm_ctx *c, int *y0, int *y1) {
  if (!c) return fail(LBM_ERR_ARG, "ctx is NULL");
  int lo = c->p.ny, hi = 0;
  for (const Slab &s : c->slabs) {
    lo = std::min(lo, s.y0);
    hi = std::max(hi, s.y0 + s.rows);
  }
  if (y0) *y0 = lo;
  if (y1) *y1 = hi;
  return LBM_OK;
}

int lbm_download(lbm_ctx *c, float *cells_out, float *av_vels_out) {
  if (!c) return fail(LBM_ERR_ARG, "ctx is NULL");
  if (int rc = sync_all(c)) return rc;
  const int nx = c->p.nx, ny = c->p.ny;
  const size_t n_global = (size_t)nx * ny;
  if (cells_out) {
    for (Slab &s : c->slabs) {
      if (set_dev(s)) return LBM_ERR_HIP;
      // the grid that is not current is scratch between runs (every row of it is rewritten before it is read):
      // repack into the reference's plane-major layout there, then nine contiguous copies (strided 2-D copies to
      // pageable host memory ran at 0.4 GB/s on 1024x1024, 16 GB/s on 8192x8192)
      const size_t n = (size_t)nx * s.rows;
      float *stage = s.cells[c->cur ^ 1];
      hipLaunchKernelGGL(pack_planes, dim3(std::min(div_up((long)n, 256), 8192)), dim3(256), 0, s.s_main, s.own(c->cur),
                         s.plane_stride, s.row_stride, nx, n, stage);
      HIP_TRY(hipGetLastError());
      for (int k = 0; k < 9; k++)
        HIP_TRY(hipMemcpyAsync(cells_out + k * n_global + (size_t)s.y0 * nx, stage + k * n, n * sizeof(float),
                               hipMemcpyDeviceToHost, s.s_main));
      HIP_TRY(hipStreamSynchronize(s.s_main));
    }
  }
  if (av_vels_out && c->steps_done > 0) {
    const int T = c->steps_done;
    std::vector<double> total(T, 0.0);
    if (c->rank_mode && c->slabs[0].comm) {
      // combine the per-rank velocity sums: one all-reduce over the whole record (into a buffer kept for later calls)
      Slab &s = c->slabs[0];
      if (set_dev(s)) return LBM_ERR_HIP;
      if (!s.av_tmp && dev_alloc(&s.av_tmp, (size_t)std::max(1, c->p.max_iters))) return LBM_ERR_HIP;
      ncclResult_t r = g_rccl.AllReduce(s.av_sum, s.av_tmp, (size_t)T, ncclDouble, ncclSum, s.comm, s.s_main);
      if (r != ncclSuccess) return fail(LBM_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
      hipError_t e = hipMemcpyAsync(total.data(), s.av_tmp, sizeof(double) * T, hipMemcpyDeviceToHost, s.s_main);
      if (e == hipSuccess) e = hipStreamSynchronize(s.s_main);
      if (e != hipSuccess) return fail(LBM_ERR_HIP, "HIP error reading av_vels: %s", hipGetErrorString(e));
    } else {
      // (a rank without communicator returns the sums over ITS rows: the caller adds the ranks' records)
      for (Slab &s : c->slabs) {
        if (set_dev(s)) return LBM_ERR_HIP;
        HIP_TRY(hipMemcpy(c->av_host, s.av_sum, sizeof(double) * T, hipMemcpyDeviceToHost));
        for (int t = 0; t < T; t++) total[t] += c->av_host[t];
      }
    }
    // kernels.cl:202: sum * FREE_CELLS_INV
    for (int t = 0; t < T; t++) av_vels_out[t] = (float)(total[t] * (double)c->p.free_cells_inv);
  }
  return LBM_OK;
}

int lbm_final_state(lbm_ctx *c, float *u_x, float *u_y, float *u, float *pressure) {
  if (!c) return fail(LBM_ERR_ARG, "ctx is NULL");
  if (int rc = sync_all(c)) return rc;
  const int nx = c->p.nx;
  float *outs[4] = {u_x, u_y, u, pressure};
  for (Slab &s : c->slabs) {
    if (set_dev(s)) return LBM_ERR_HIP;
    const size_t n = (size_t)nx * s.rows;
    // the four columns go to the grid that is not current — scratch between runs, 9 n floats of room (as in
    // lbm_download): no allocation inside the reference-rule timed region (d2q9-bgk.c:196-263)
    float *d[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; i++)
      if (outs[i]) d[i] = s.cells[c->cur ^ 1] + (size_t)i * n;
    hipLaunchKernelGGL(final_fields, dim3(s.fin_blocks), dim3(kBlock), 0, s.s_main, s.own(c->cur), s.plane_stride,
                       s.row_stride, nx, s.mask_own(nx), n, c->p.density, d[0], d[1], d[2], d[3], s.fin_partials);
    hipError_t e = hipGetLastError();
    for (int i = 0; i < 4 && e == hipSuccess; i++)
      if (outs[i]) e = hipMemcpyAsync(outs[i] + (size_t)s.y0 * nx, d[i], n * sizeof(float), hipMemcpyDeviceToHost, s.s_main);
    if (e == hipSuccess) e = hipStreamSynchronize(s.s_main);
    if (e != hipSuccess) return fail(LBM_ERR_HIP, "HIP error in output stage: %s", hipGetErrorString(e));
  }
  return LBM_OK;
}

int lbm_reynolds(lbm_ctx *c, float *reynolds_out) {
  if (!c || !reynolds_out) return fail(LBM_ERR_ARG, "NULL argument");
  if (int rc = sync_all(c)) return rc;
  double tot = 0.0;
  for (Slab &s : c->slabs) {
    if (set_dev(s)) return LBM_ERR_HIP;
    const size_t n = (size_t)c->p.nx * s.rows;
    hipLaunchKernelGGL(final_fields, dim3(s.fin_blocks), dim3(kBlock), 0, s.s_main, s.own(c->cur), s.plane_stride,
                       s.row_stride, c->p.nx, s.mask_own(c->p.nx), n, c->p.density, (float *)nullptr, (float *)nullptr, (float *)nullptr,
                       (float *)nullptr, s.fin_partials);
    HIP_TRY(hipGetLastError());
    std::vector<float> part(s.fin_blocks);
    HIP_TRY(hipMemcpyAsync(part.data(), s.fin_partials, sizeof(float) * s.fin_blocks, hipMemcpyDeviceToHost, s.s_main));
    HIP_TRY(hipStreamSynchronize(s.s_main));
    for (float v : part) tot += v;
  }
  if (c->rank_mode && c->slabs[0].comm) {
    Slab &s = c->slabs[0];
    double *tmp = nullptr;
    if (dev_alloc(&tmp, 2)) return LBM_ERR_HIP;
    hipError_t e = hipMemcpy(tmp, &tot, sizeof(double), hipMemcpyHostToDevice);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = g_rccl.AllReduce(tmp, tmp + 1, 1, ncclDouble, ncclSum, s.comm, s.s_main);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(s.s_main);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpy(&tot, tmp + 1, sizeof(double), hipMemcpyDeviceToHost);
    hipFree(tmp);
    if (r != ncclSuccess) return fail(LBM_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
    if (e != hipSuccess) return fail(LBM_ERR_HIP, "HIP error in reynolds: %s", hipGetErrorString(e));
  }
  // d2q9-bgk.c:747-752
  const float viscosity = 1.0f / 6.0f * (2.0f / c->p.omega - 1.0f);
  const float av = (float)(tot * (double)c->p.free_cells_inv);
  *reynolds_out = av * c->p.reynolds_dim / viscosity;
  return LBM_OK;
}

int lbm_set_option(lbm_ctx *c, const char *key, long value) {
  if (!c || !key) return fail(LBM_ERR_ARG, "NULL argument");
  if (!strcmp(key, "variant")) {
    if (value < 0 || value > 4) return fail(LBM_ERR_ARG, "variant must be 0..4");
    c->variant = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "grid_blocks")) {
    if (value < 0) return fail(LBM_ERR_ARG, "grid_blocks must be >= 0");
    if (int rc = sync_all(c)) return rc;
    c->grid_blocks = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "nt_stores")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "nt_stores must be -1 (auto), 0 or 1");
    c->nt_stores = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "nt_loads")) {
    if (value < -1 || value > 2) return fail(LBM_ERR_ARG, "nt_loads must be -1 (auto), 0, 1 or 2");
    c->nt_loads = (int)value;
    return c->halo_mode ? rebuild_geometry(c) : LBM_OK;   // (whether launch sets are compact depends on it)
  }
  if (!strcmp(key, "fuse")) {
    // (5: the five-step chunk pairs of row slabs with five halo rows in compact launch sets, slab_twin5; the four-step kernel elsewhere)
    if (value < -1 || value > kDeepSteps) return fail(LBM_ERR_ARG, "fuse must be -1 (auto), 0, 1 (or 2), 3 .. 8");
    c->fuse = (int)value;
    return c->halo_mode ? rebuild_geometry(c) : LBM_OK;
  }
  if (!strcmp(key, "twin_steps")) {
    if (value != 0 && (value < 2 || value > kDeepTwinSteps)) return fail(LBM_ERR_ARG, "twin_steps must be 0 (auto) or 2..%d", kDeepTwinSteps);
    if (int rc = sync_all(c)) return rc;
    c->twin_steps = (int)value;
    return rebuild_geometry(c);  // (the strips of the twins depend on it)
  }
  if (!strcmp(key, "edge_aware")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "edge_aware must be -1 (auto), 0 or 1");
    if (int rc = sync_all(c)) return rc;
    c->edge_aware = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "steady")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "steady must be -1 (auto), 0 or 1");
    c->steady = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "obst_paths")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "obst_paths must be -1 (auto), 0 or 1");
    c->obst_paths = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "balance")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "balance must be -1 (auto), 0 or 1");
    if (int rc = sync_all(c)) return rc;
    c->balance = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "free_sweeps")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "free_sweeps must be -1 (auto), 0 or 1");
    c->free_sweeps = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "tile_shape")) {
    if (value < -1 || value > 2) return fail(LBM_ERR_ARG, "tile_shape must be -1..2");
    if (int rc = sync_all(c)) return rc;
    c->tile_shape = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "pair")) {
    if (int rc = sync_all(c)) return rc;
    c->pair = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "windows") || !strcmp(key, "load_bufs") || !strcmp(key, "sched_waves")) {
    if (value < -1 || value > 2) return fail(LBM_ERR_ARG, "%s out of range", key);
    // (round 4: the three-step kernel exists with its windows in LDS and one row-set of loads in flight only)
    if (key[0] == 'w' && value == 0) return fail(LBM_ERR_ARG, "windows 0 (register windows of d2q9_step3) was removed: -1 or 1");
    if (key[0] == 'l' && value == 2) return fail(LBM_ERR_ARG, "load_bufs 2 (two row-sets of loads in flight in d2q9_step3) was removed: 0 or 1");
    if (int rc = sync_all(c)) return rc;
    (key[0] == 'w' ? c->windows : (key[0] == 'l' ? c->load_bufs : c->sched_waves)) = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "transport")) {
    // halo transport of a context that has both: every rank of the ring must make the same call at the same point
    if (!c->halo_mode) return fail(LBM_ERR_STATE, "a context without halo rows has no transport");
    if (int rc = sync_all(c)) return rc;
    if (value == TRANSPORT_RCCL) {
      for (const Slab &s : c->slabs)
        if (!s.comm) return fail(LBM_ERR_STATE, "no RCCL communicator: create the context with a comm_id (or default transport 1)");
    } else if (value == TRANSPORT_PEER) {
      for (const Slab &s : c->slabs)
        if (!s.south.connected || !s.north.connected) return fail(LBM_ERR_STATE, "peers are not connected (lbm_connect_peers)");
    } else {
      return fail(LBM_ERR_ARG, "transport must be 1 (RCCL send/recv) or 3 (peer stores)");
    }
    c->transport_eff = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "compact")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "compact must be -1 (auto), 0 or 1");
    if (int rc = sync_all(c)) return rc;
    c->compact = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "halo_sync")) {
    if (value < 0 || value > 2)
      return fail(LBM_ERR_ARG, "halo_sync must be 0 (wait kernel), 1 (hipStreamWaitValue32) or 2 (inside the consuming kernel where it can)");
    if (value == 1) {
      int can = 0;
      for (const Slab &s : c->slabs) {
        HIP_TRY(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, s.dev));
        if (!can) return fail(LBM_ERR_STATE, "device %d cannot wait on memory values", s.dev);
      }
    }
    if (value == 2)
      for (const Slab &s : c->slabs)
        if ((s.south.connected && s.south.remote) || (s.north.connected && s.north.remote))
          return fail(LBM_ERR_STATE, "halo_sync 2 (the consuming kernel polls the flag words and reads the halo rows without a cache "
                      "invalidate in between) is accepted only while every ring neighbour lives on this slab's own device: between "
                      "devices the wait kernel (0) orders the reads behind a kernel-start acquire");
    if (int rc = sync_all(c)) return rc;
    c->halo_sync = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "push_release")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "push_release must be -1 (auto), 0 or 1");
    if (int rc = sync_all(c)) return rc;
    c->push_release = (int)value;
    for (Slab &s : c->slabs)
      if (s.d_peer && s.south.connected && s.north.connected)
        if (int rc = upload_multi_peer(c, s)) return rc;
    return LBM_OK;
  }
  if (!strcmp(key, "debug_stale_exchange")) {
    if (value < 0) return fail(LBM_ERR_ARG, "debug_stale_exchange must be >= 0");
    if (!c->halo_mode) return fail(LBM_ERR_STATE, "a context without halo rows exchanges nothing");
    c->stale_exchange = (int)value;
    return LBM_OK;
  }
  if (!strcmp(key, "halo_timeout_ms")) {
    if (value < 1 || value > 600000) return fail(LBM_ERR_ARG, "halo_timeout_ms must be 1..600000");
    if (int rc = sync_all(c)) return rc;
    c->halo_timeout_ms = (unsigned long long)value;
    for (Slab &s : c->slabs)
      if (s.d_peer && s.south.connected && s.north.connected)
        if (int rc = upload_multi_peer(c, s)) return rc;
    return LBM_OK;
  }
  if (!strcmp(key, "resident")) {
    if (value < -1 || value > 1) return fail(LBM_ERR_ARG, "resident must be -1 (auto), 0 or 1");
    if (int rc = sync_all(c)) return rc;
    c->resident = (int)value;
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "multistep")) {
    if (value < -1 || value > kMultiMaxT) return fail(LBM_ERR_ARG, "multistep must be -1..%d", kMultiMaxT);
    c->multistep = (int)value;
    return c->halo_mode ? rebuild_geometry(c) : LBM_OK;
  }
  if (!strcmp(key, "vec")) {
    // cells per thread of the single-step kernel: 4 (float4 rows) or 1
    if (value != 1 && value != 4) return fail(LBM_ERR_ARG, "vec must be 1 or 4");
    if (value == 4 && c->p.nx % 4 != 0) return fail(LBM_ERR_ARG, "vec=4 needs nx %% 4 == 0");
    if (int rc = sync_all(c)) return rc;
    c->vec4 = (value == 4);
    return rebuild_geometry(c);
  }
  if (!strcmp(key, "chunk_rows") || !strcmp(key, "chunk_min")) {
    if (value < 0) return fail(LBM_ERR_ARG, "%s must be >= 0", key);
    if (int rc = sync_all(c)) return rc;
    (key[6] == 'r' ? c->chunk_rows : c->chunk_min) = (int)value;
    return rebuild_geometry(c);
  }
  return fail(LBM_ERR_ARG, "unknown option '%s'", key);
}

int lbm_get_option(const lbm_ctx *c, const char *key, long *value) {
  if (!c || !key || !value) return fail(LBM_ERR_ARG, "NULL argument");
  if (!strcmp(key, "variant")) *value = effective_mode(c) + 1;
  else if (!strcmp(key, "grid_blocks")) *value = c->slabs.empty() ? 0 : c->slabs[0].nb_main;
  else if (!strcmp(key, "nt_stores")) *value = nt_effective(c);
  else if (!strcmp(key, "fuse")) *value = fuse_level(c) >= 3 ? fuse_level(c) : (fuse_level(c) ? 1 : 0);
  else if (!strcmp(key, "multistep")) *value = resident_effective(c) ? 0 : multistep_effective(c);
  else if (!strcmp(key, "resident")) *value = resident_effective(c) ? c->slabs[0].res_bh : 0;  // rows per band, 0 = not in use
  else if (!strcmp(key, "chunk_rows")) *value = c->chunk_rows;
  else if (!strcmp(key, "windows")) *value = windows_in_lds(c);
  else if (!strcmp(key, "pair")) *value = c->slabs.empty() ? 0 : slab_twin5(c) ? 1 : (fuse_level(c) >= kDeepMin ? (c->halo_mode ? (compact_sets(c) && c->slabs[0].f6_main.paired) : deep_twin_effective(c)) : fuse_level(c) == 4 ? c->slabs[0].f4_main.paired : c->slabs[0].f3_main.paired);
  else if (!strcmp(key, "load_bufs")) *value = step3_load_bufs(c);
  else if (!strcmp(key, "launch_steps")) {
    // most timesteps one launch (launch set) of the context's main kernel advances
    const int ms = multistep_effective(c), lvl = fuse_level(c);
    *value = resident_effective(c) ? c->ring : ms > 0 ? ms : slab_twin5(c) ? kDeepTwinDefault : (lvl >= kDeepMin ? (deep_twin_effective(c) ? std::min(lvl, twin_cap(c)) : lvl) : (lvl >= 3 ? lvl : (lvl ? 2 : 1)));
  }
  else if (!strcmp(key, "fuse_units")) *value = c->slabs.empty() ? 0 : ((fuse_level(c) >= kDeepMin || slab_twin5(c)) ? c->slabs[0].f6_main.units + c->slabs[0].f6_edge.units - c->slabs[0].f_edge.units : fuse_level(c) == 4 ? c->slabs[0].f4_main.units : (fuse_level(c) == 3 ? c->slabs[0].f3_main.units : c->slabs[0].f_main.units)) + c->slabs[0].f_edge.units;
  else if (!strcmp(key, "transport")) *value = c->transport_eff;
  else if (!strcmp(key, "steady")) *value = c->steady != 0;
  else if (!strcmp(key, "balance")) {
    // strips of the context's deep window kernel that got a second (virtual) strip
    *value = 0;
    if (!c->slabs.empty() && fuse_level(c) >= kDeepMin) {
      const Slab &s0 = c->slabs[0];
      const FuseGeom &g = (!c->halo_mode && deep_twin_effective(c)) ? s0.f6_twin : s0.f6_main;
      *value = g.vstrips > 0 ? g.vstrips - s0.strips2 : 0;
    }
  }
  else if (!strcmp(key, "free_sweeps")) {
    // are the launches of the context's deep window kernel given the map?
    *value = 0;
    if (!c->slabs.empty() && fuse_level(c) >= kDeepMin) {
      const Slab &s0 = c->slabs[0];
      *value = clean_bits_for(c, s0, (!c->halo_mode && deep_twin_effective(c)) ? s0.f6_twin : s0.f6_main) != nullptr;
    }
  }
  else if (!strcmp(key, "halo_sync")) *value = c->halo_sync;
  else if (!strcmp(key, "halo_timeout_ms")) *value = (long)c->halo_timeout_ms;
  else if (!strcmp(key, "push_release")) *value = c->slabs.empty() ? 0 : push_release_effective(c, c->slabs[0]);
  else if (!strcmp(key, "debug_stale_exchange")) *value = c->stale_exchange;
  else if (!strcmp(key, "compact")) *value = compact_sets(c);
  else if (!strcmp(key, "halo_depth")) *value = c->halo_mode ? c->halo_depth : 0;
  else if (!strcmp(key, "nslabs")) *value = c->nslabs_global;
  else return fail(LBM_ERR_ARG, "unknown option '%s'", key);
  return LBM_OK;
}

int lbm_valu_rate(int launches, double *tera_lane_instr_per_s) {
  if (!tera_lane_instr_per_s || launches < 1) return fail(LBM_ERR_ARG, "bad argument");
  // 8 workgroups of 256 threads per CU: eight waves per SIMD, each with eight independent chains
  const int nb = 256 * 8, iters = 4096;  // 4 x 8 packed FMAs per loop iteration: 131072 per thread and launch (~1.9 ms)
  float *out = nullptr;
  HIP_TRY(hipMalloc((void **)&out, (size_t)nb * kBlock * sizeof(float)));
  hipEvent_t t0, t1;
  hipEventCreate(&t0);
  hipEventCreate(&t1);
  hipLaunchKernelGGL(lbm::valu_spin, dim3(nb), dim3(kBlock), 0, 0, out, iters, 0.5f);  // warm-up
  hipEventRecord(t0, 0);
  for (int i = 0; i < launches; i++) hipLaunchKernelGGL(lbm::valu_spin, dim3(nb), dim3(kBlock), 0, 0, out, iters, 0.5f);
  hipEventRecord(t1, 0);
  const hipError_t e = hipEventSynchronize(t1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, t0, t1);
  hipEventDestroy(t0);
  hipEventDestroy(t1);
  hipFree(out);
  if (e != hipSuccess || ms <= 0.f) return fail(LBM_ERR_HIP, "valu_spin: %s", hipGetErrorString(e));
  *tera_lane_instr_per_s = (double)nb * kBlock * (double)iters * 32.0 * launches / (ms * 1e-3) / 1e12;
  return LBM_OK;
}

int lbm_copy_bandwidth(size_t bytes, int iters, double *gbps) {
  if (!gbps || iters < 1 || bytes < 16) return fail(LBM_ERR_ARG, "bad argument");
  const size_t n = bytes / 16;
  float4 *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc((void **)&a, n * 16));
  hipError_t e = hipMalloc((void **)&b, n * 16);
  if (e != hipSuccess) {
    hipFree(a);
    return fail(LBM_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
  }
  hipEvent_t t0, t1;
  hipEventCreate(&t0);
  hipEventCreate(&t1);
  hipMemset(a, 1, n * 16);
  // best of two launch shapes: 4096 grid-striding workgroups with plain accesses, one tile per workgroup with
  // non-temporal accesses
  double best = 0.0;
  for (int shape = 0; shape < 2 && e == hipSuccess; shape++) {
    const int nb = shape == 0 ? (int)std::min<size_t>((n + kBlock - 1) / kBlock, 256 * 16) : (int)((n + kBlock - 1) / kBlock);
    auto launch = [&](float4 *src, float4 *dst) {
      if (shape == 0) hipLaunchKernelGGL(lbm::copy_f4, dim3(nb), dim3(kBlock), 0, 0, src, dst, n);
      else hipLaunchKernelGGL(lbm::copy_f4_nt, dim3(nb), dim3(kBlock), 0, 0, (const float *)src, (float *)dst, n);
    };
    launch(a, b);  // warm-up
    hipEventRecord(t0, 0);
    for (int i = 0; i < iters; i++) {
      if (i & 1) launch(b, a);
      else launch(a, b);
    }
    hipEventRecord(t1, 0);
    e = hipEventSynchronize(t1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, t0, t1);
    if (e == hipSuccess && ms > 0.f) best = std::max(best, 2.0 * (double)(n * 16) * iters / (ms * 1e-3) / 1e9);
  }
  hipEventDestroy(t0);
  hipEventDestroy(t1);
  hipFree(a);
  hipFree(b);
  if (e != hipSuccess) return fail(LBM_ERR_HIP, "copy kernel: %s", hipGetErrorString(e));
  *gbps = best;
  return LBM_OK;
}

int lbm_host_alloc(void **ptr_out, size_t bytes) {
  if (!ptr_out || bytes == 0) return fail(LBM_ERR_ARG, "lbm_host_alloc: bad argument");
  *ptr_out = nullptr;
  HIP_TRY(hipHostMalloc(ptr_out, bytes, hipHostMallocDefault));
  return LBM_OK;
}

int lbm_host_free(void *ptr) {
  if (!ptr) return LBM_OK;
  HIP_TRY(hipHostFree(ptr));
  return LBM_OK;
}

void lbm_destroy(lbm_ctx *c) {
  if (!c) return;
  for (Slab &s : c->slabs) {
    hipSetDevice(s.dev);
    if (s.s_main) hipStreamSynchronize(s.s_main);
    if (s.s_edge) hipStreamSynchronize(s.s_edge);
  }
  for (hipEvent_t e : c->prof_ev) hipEventDestroy(e);
  for (Slab &s : c->slabs) free_slab(s);
  free(c->av_host);
  delete c;
}

}  // extern "C"
