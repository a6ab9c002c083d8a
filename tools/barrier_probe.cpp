// tools/barrier_probe.cpp — what a persistent kernel would pay per launch set on MI355X, against what a kernel boundary costs.
//   hipcc -O3 --offload-arch=gfx950 tools/barrier_probe.cpp -o tools/barrier_probe && ./tools/barrier_probe
// Question: the latency-bound sizes (128x128 ... 1024x1024, the 1024x128 slabs of an
// 8-GPU run) pay one kernel boundary per launch set of up to 8 timesteps.  Would ONE persistent launch with a grid-wide
// barrier between the sets be cheaper?  Measured here, nothing else in the loop:
//   (a) the period of back-to-back launches of an empty kernel of the same shape (256 workgroups x 1024 threads, and
//       2048 x 64: the shapes of d2q9_multi and of the window kernels) on one stream;
//   (b) a grid barrier on ONE counter (agent-scope atomic add, every workgroup polls it);
//   (c) an XCD-hierarchical barrier: workgroups of one XCD meet on that XCD's counter, the last of each XCD on a global
//       counter, the last of all publishes the generation in eight per-XCD words which the others poll;
//   (d) a neighbour-only hand-shake: every workgroup raises its own generation word and polls those of its two ring
//       neighbours (what tiles of a stencil need: no global meeting point).
// All spins are bounded (a barrier that never completes ends the kernel with an error count instead of hanging the GPU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void empty_kernel(unsigned *sink) { if (sink && threadIdx.x == 0 && blockIdx.x == 0xffffffffu) *sink = 1; }

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wait_ge(const unsigned *p, unsigned v, unsigned *err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((int)(ld_agent(p) - v) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { atomicAdd(err, 1u); return false; }  // 2 s
  }
  return true;
}
__device__ __forceinline__ unsigned xcc_id() {  // XCC_ID register (HW_REG 20), bits 3:0
  return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;
}

struct Bar { unsigned *glob; unsigned *xcd_cnt; unsigned *xcd_gen; unsigned *own; unsigned *err; unsigned *xcd_size; };

// mode 1: flat; mode 2: hierarchical; mode 3: ring neighbours
__global__ void barrier_kernel(Bar b, int iters, int mode, unsigned long long *ticks) {
  const unsigned nwg = gridDim.x, me = blockIdx.x;
  __shared__ unsigned x_sh, xsz_sh;
  if (threadIdx.x == 0) { x_sh = xcc_id(); }
  __syncthreads();
  const unsigned x = x_sh;
  if (mode == 2) {  // census: how many workgroups does each XCD hold? (once, flat barrier behind it)
    if (threadIdx.x == 0) {
      atomicAdd(&b.xcd_size[x], 1u);
      __hip_atomic_fetch_add(b.glob, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_ge(b.glob, nwg, b.err);
      xsz_sh = ld_agent(&b.xcd_size[x]);
    }
    __syncthreads();
  }
  const unsigned xsz = xsz_sh;
  const unsigned base = (mode == 2) ? nwg : 0u;  // the census used the global counter once
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 1; it <= iters; it++) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (mode == 1) {
        __hip_atomic_fetch_add(b.glob, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        wait_ge(b.glob, nwg * (unsigned)it, b.err);
      } else if (mode == 2) {
        const unsigned t = __hip_atomic_fetch_add(&b.xcd_cnt[x * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (t == xsz * (unsigned)it - 1u) {   // last of this XCD
          const unsigned g = __hip_atomic_fetch_add(b.glob, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          unsigned nx = 0;
          for (int k = 0; k < 8; k++) nx += ld_agent(&b.xcd_size[k]) != 0u;
          if (g == base + nx * (unsigned)it - 1u)   // last of all: publish the generation to every XCD
            for (int k = 0; k < 8; k++) __hip_atomic_store(&b.xcd_gen[k * 32], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        wait_ge(&b.xcd_gen[x * 32], (unsigned)it, b.err);
      } else {
        __hip_atomic_store(&b.own[me * 32], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        wait_ge(&b.own[((me + 1) % nwg) * 32], (unsigned)it, b.err);
        wait_ge(&b.own[((me + nwg - 1) % nwg) * 32], (unsigned)it, b.err);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && me == 0) *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned *dev;
  CK(hipMalloc(&dev, 1 << 20));
  unsigned long long *ticks;
  CK(hipMalloc(&ticks, 8));
  const int shapes[3][2] = {{256, 1024}, {2048, 64}, {1024, 128}};
  for (auto &sh : shapes) {
    for (int r = 0; r < 200; r++) hipLaunchKernelGGL(empty_kernel, dim3(sh[0]), dim3(sh[1]), 0, st, dev);
    CK(hipStreamSynchronize(st));
    const int n = 2000;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < n; r++) hipLaunchKernelGGL(empty_kernel, dim3(sh[0]), dim3(sh[1]), 0, st, dev);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("(a) empty kernel %4d x %4d: %.2f us per launch, back to back on one stream\n", sh[0], sh[1], ms * 1e3 / n);
  }
  const char *names[4] = {"", "(b) flat barrier, one counter      ", "(c) XCD-hierarchical barrier       ", "(d) ring neighbours only (2 flags) "};
  const int bshapes[3][2] = {{256, 1024}, {256, 64}, {512, 64}};
  for (auto &sh : bshapes)
    for (int mode = 1; mode <= 3; mode++) {
      CK(hipMemset(dev, 0, 1 << 20));
      Bar b{dev, dev + 1024, dev + 2048, dev + 4096, dev + 3072, dev + 3104};
      const int iters = 2000;
      hipLaunchKernelGGL(barrier_kernel, dim3(sh[0]), dim3(sh[1]), 0, st, b, iters, mode, ticks);
      CK(hipStreamSynchronize(st));
      unsigned long long t = 0;
      unsigned err = 0, xs[8];
      CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&err, dev + 3072, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(xs, dev + 3104, 32, hipMemcpyDeviceToHost));
      printf("%s %4d x %4d: %.2f us per barrier (%d in one launch)%s", names[mode], sh[0], sh[1], t * 0.01 / iters, iters, err ? "  TIMED OUT" : "");
      if (mode == 2) { printf("  workgroups per XCD:"); for (int k = 0; k < 8; k++) printf(" %u", xs[k]); }
      printf("\n");
    }
  return 0;
}
