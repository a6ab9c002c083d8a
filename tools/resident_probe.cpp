// tools/resident_probe.cpp — what a RESIDENT-tile kernel family could reach on 1024x1024.
//   hipcc -O3 --offload-arch=gfx950 tools/resident_probe.cpp -o tools/resident_probe && ./tools/resident_probe
//
// The design it prices: the grid never leaves the chip.  1024x1024 cells x 36 B = 37.7 MB of state against 128 MB of vector
// registers (256 CUs x 512 KB) and 40 MB of LDS: it fits the registers, not the LDS.  With two waves per SIMD (2048 waves, what
// ~180 registers per lane of state + collision allow) a wave holds 512 cells — a block of 128 x 4 (64 lanes of two cells, as in
// d2q9_deep; 9 planes x 4 rows x 2 = 72 registers) — and ONE persistent launch runs all timesteps: per step a wave
//   (1) stores the three planes of its top row that move up and of its bottom row that move down (64 lanes x 8 B x 3, each way)
//       into exchange rows in memory, double-buffered by step parity, write-through where the reader sits on another XCD,
//   (2) raises its step word, (3) waits for the step words of the block above and the block below (x neighbours: one element
//       per row and plane — left out here, it only adds), (4) loads their rows, (5) collides its 4 rows (~4 x 88 packed
//       instructions, the count of d2q9_deep's steady loop per row and level).
// There is no launch boundary, no start-up row and no redundant halo lane — and one neighbour hand-shake PER STEP on the
// critical path: a block's boundary row of step t+1 needs its neighbour's of step t, which needed this block's of step t-1.
// The probe runs exactly that communication pattern with a stand-in for the arithmetic (a chain of packed FMAs of the same
// length, half of it independent of the received rows = the two inner rows) and reports the step period
//   - for the blockIdx -> block mappings "column" (block = blockIdx: vertical neighbours are 8 workgroups apart and land on the
//     same XCD under round-robin dispatch) and "banded" (32 consecutive block rows per XCD: 64 of 2048 neighbour pairs cross
//     XCDs) — the census of XCC_ID confirms both; all flag and row accesses at agent scope (sc0-only loads, which would stay
//     inside an XCD, do not see another CU's stores: tried, the waits timed out),
//   - with and without the arithmetic, so the hand-shake alone is visible,
//   - for T = 1 and T = 2 timesteps per hand-shake: T rows travel each way, a block then computes 4T + T(T-1) row-steps per
//     hand-shake (the redundant rows of a T-deep halo included), of which only the two inner rows of the first step do not
//     depend on what arrives (T = 2: 8 rows of state = 144 registers; T = 3 would need 180 + the collision's: it does not fit).
// All spins are bounded, one timeout per run.  Nothing here is part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u; }

struct Args {
  unsigned *step;        // [nblocks * 32] a block's step word (one per 128 B)
  float *rows;           // [2 parity][nblocks][2 dir][3 planes][128 floats]
  unsigned *err;
  unsigned *xcd_of;      // [nblocks] which XCD a block ran on (census)
  unsigned long long *ticks;
  int nbx, nby;          // blocks across x (8 at nx = 1024) and along y (256)
  int iters, mapping, T, work;     // mapping 0 column / 1 banded; T timesteps per hand-shake; work: 6 packed instructions per unit and row
};

__device__ __forceinline__ unsigned ld_flag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ v2f ld_row(const float *p) {
  const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v2f v = {__uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32))};
  return v;
}
__device__ __forceinline__ void st_row(float *p, v2f v) {
  const unsigned long long b = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int T>
__global__ __launch_bounds__(64, 2) void resident(const Args a) {
  // (20 KB of LDS per wave: eight waves per CU, two per SIMD — the occupancy the real kernel's registers would give; without
  // it the dispatcher packs these small waves eight to a SIMD and leaves CUs empty)
  __shared__ float pad[5000];
  if (a.iters < 0) pad[threadIdx.x] = 1.f;
  const int lane = threadIdx.x;
  const int nb = a.nbx * a.nby;
  // which block am I?  naive: blockIdx; banded: workgroups are dispatched round-robin over the 8 XCDs, so blockIdx % 8 is
  // (normally) the XCD and blockIdx / 8 the slot within it: XCD x holds block rows [x * nby/8, (x+1) * nby/8)
  int me;
  if (a.mapping == 0) me = blockIdx.x;
  else {
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;  // slot 0 .. nb/8-1
    const int by = x * (a.nby / 8) + slot / a.nbx, bx = slot % a.nbx;
    me = by * a.nbx + bx;
  }
  const int by = me / a.nbx, bx = me % a.nbx;
  const int up = ((by + 1) % a.nby) * a.nbx + bx, dn = ((by + a.nby - 1) % a.nby) * a.nbx + bx;
  if (lane == 0) a.xcd_of[me] = xcc_id();
  v2f st[4][3];  // stand-in state: 4 rows x 3 planes (the planes that matter for the dependence structure)
  for (int r = 0; r < 4; r++)
    for (int k = 0; k < 3; k++) st[r][k] = v2f{1.0f + 0.001f * lane, 1.0f + 0.002f * r + 0.01f * k};
  const size_t blk = (size_t)2 * T * 3 * 128;  // [dir][T rows][3 planes][128 floats]
  constexpr int dep_rows = 4 * T + T * (T - 1) - 2;  // row-steps per hand-shake that need the received rows
  auto row_work = [&](v2f (&row)[3]) {
    for (int w = 0; w < a.work; w++)
      for (int k = 0; k < 3; k++) row[k] = __builtin_elementwise_fma(row[k], v2f{0.999f, 1.001f}, row[(k + 1) % 3] * 1e-3f);
  };
  unsigned long long t0 = 0;
  for (int it = 1; it <= a.iters; it++) {
    if (it == 65) t0 = __builtin_amdgcn_s_memrealtime();
    float *mine = a.rows + ((size_t)(it & 1) * nb + me) * blk;
    // (1) my T boundary rows each way: the top rows' planes for the block above (dir 0), the bottom rows' for the block below (dir 1)
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        st_row(mine + ((0 * T + t) * 3 + k) * 128 + 2 * lane, st[3 - (t & 1)][k]);
        st_row(mine + ((1 * T + t) * 3 + k) * 128 + 2 * lane, st[t & 1][k]);
      }
    // (2) publish: my write-through stores first (acknowledged = they have left the XCD), then the step word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&a.step[me * 32], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the two inner rows of the first step do not need the neighbours: their arithmetic runs while the words travel
    row_work(st[1]);
    row_work(st[2]);
    // (3) wait for both neighbours' step words (lanes 0 and 1 poll, bounded; ONE timeout per run), (4) load their rows
    if (lane < 2) {
      const unsigned *f = &a.step[(lane == 0 ? up : dn) * 32];
      const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
      // (the error word is looked at only by a wait that has already lasted 100 us: 4096 lanes reading ONE word on every
      // hand-shake made that word's memory channel the bottleneck — 4.8 instead of 2.05 us per hand-shake)
      while ((int)(ld_flag(f) - (unsigned)it) < 0) {
        __builtin_amdgcn_s_sleep(1);
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - w0;
        if (dt > 20000000ull) { atomicAdd(a.err, 1u); break; }  // 0.2 s
        if (dt > 10000ull && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const float *fu = a.rows + ((size_t)(it & 1) * nb + up) * blk, *fd = a.rows + ((size_t)(it & 1) * nb + dn) * blk;
    v2f acc_up[3] = {v2f{0, 0}, v2f{0, 0}, v2f{0, 0}}, acc_dn[3] = {v2f{0, 0}, v2f{0, 0}, v2f{0, 0}};
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        acc_up[k] += ld_row(fu + ((1 * T + t) * 3 + k) * 128 + 2 * lane);  // the block above sent its bottom rows down
        acc_dn[k] += ld_row(fd + ((0 * T + t) * 3 + k) * 128 + 2 * lane);  // the block below its top rows up
      }
    // (5) everything else consumes what arrived
    for (int k = 0; k < 3; k++) { st[3][k] += acc_up[k] * 1e-6f; st[0][k] += acc_dn[k] * 1e-6f; }
#pragma unroll
    for (int r = 0; r < dep_rows; r++) row_work(st[r & 1 ? 3 : 0]);
    for (int k = 0; k < 3; k++) { st[1][k] += st[0][k] * 1e-6f; st[2][k] += st[3][k] * 1e-6f; }
  }
  if (lane == 0 && me == 0) *a.ticks = __builtin_amdgcn_s_memrealtime() - t0;
  float s = 0.f;
  for (int r = 0; r < 4; r++)
    for (int k = 0; k < 3; k++) s += st[r][k].x + st[r][k].y;
  if (s == 123.456f) a.step[0] = (unsigned)pad[lane];  // keep the arithmetic (and the LDS allocation)
}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  hipStream_t stq;
  CK(hipStreamCreateWithFlags(&stq, hipStreamNonBlocking));
  int nx = 1024, ny = 1024;
  if (argc > 2) { nx = atoi(argv[1]); ny = atoi(argv[2]); }
  const int nbx = nx / 128, nby = ny / 4, nb = nbx * nby;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%dx%d cells = %d blocks of 128 x 4 (one wave each); the chip holds %d waves at two per SIMD\n", nx, ny, nb,
         prop.multiProcessorCount * 8);
  if (nb > prop.multiProcessorCount * 8 || nby % 8) { printf("does not fit as one resident round\n"); return 1; }
  Args a{};
  const int Tmax = 2;
  const size_t row_floats = (size_t)2 * nb * 2 * Tmax * 3 * 128;
  CK(hipMalloc(&a.step, (size_t)nb * 32 * 4));
  CK(hipMalloc(&a.rows, row_floats * 4));
  CK(hipMalloc(&a.err, 4));
  CK(hipMalloc(&a.xcd_of, (size_t)nb * 4));
  CK(hipMalloc(&a.ticks, 8));
  a.nbx = nbx; a.nby = nby;
  const int iters = 4064;
  // per row-step: 88 packed instructions per level in d2q9_deep's steady loop; the stand-in issues 3 x 2 (mul + fma) per `work` unit
  const int works[2] = {0, 15};
  for (int T = 1; T <= Tmax; T++)
    for (int mapping = 0; mapping < 2; mapping++)
      for (int work : works) {
        CK(hipMemset(a.step, 0, (size_t)nb * 32 * 4));
        CK(hipMemset(a.rows, 0, row_floats * 4));
        CK(hipMemset(a.err, 0, 4));
        a.iters = iters; a.mapping = mapping; a.T = T; a.work = work;
        if (T == 1) hipLaunchKernelGGL(resident<1>, dim3(nb), dim3(64), 0, stq, a);
        else hipLaunchKernelGGL(resident<2>, dim3(nb), dim3(64), 0, stq, a);
        CK(hipStreamSynchronize(stq));
        unsigned long long t = 0;
        unsigned err = 0;
        std::vector<unsigned> xo(nb);
        CK(hipMemcpy(&t, a.ticks, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(xo.data(), a.xcd_of, (size_t)nb * 4, hipMemcpyDeviceToHost));
        int cross = 0;  // block pairs (me, up) that really ran on different XCDs
        for (int by = 0; by < nby; by++)
          for (int bx = 0; bx < nbx; bx++) cross += xo[by * nbx + bx] != xo[((by + 1) % nby) * nbx + bx];
        const double per_shake = t * 0.01 / (iters - 64);
        printf("T %d  mapping %-6s  arithmetic %3d packed instr per row-step: %6.2f us per hand-shake = %5.2f us per timestep   (%d of %d vertical neighbour pairs on different XCDs)%s\n",
               T, mapping ? "banded" : "column", work * 6, per_shake, per_shake / T, cross, nb, err ? "  TIMED OUT" : "");
      }
  return 0;
}
