// Probe of the HIP features the peer-halo transport relies on (run on the GPU box, prints findings):
//   1. hipStreamWaitValue32 / hipStreamWriteValue32 on signal memory and on plain hipMalloc memory
//   2. the same inside a stream capture / as explicit hipGraphAddBatchMemOpNode
//   3. kernel-side flag write + kernel-side flag wait inside a captured graph
//   4. two PROCESSES on one GPU: hipIpcGetMemHandle / hipIpcOpenMemHandle of a grid buffer and of a flag word,
//      peer kernel writes payload + flag, owner waits (stream wait, kernel wait) and checks every word
//   5. latency of one flag hand-over between two streams (write value -> wait value -> tiny kernel)
// hipcc --offload-arch=gfx950 -O2 tools/probe_memops.cpp -o tools/probe_memops
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("  FAIL %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);               \
      fflush(stdout);                                                                          \
      (void)hipGetLastError();                                                                 \
      ok = false;                                                                              \
    }                                                                                          \
  } while (0)

__global__ void fill(unsigned *p, unsigned v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i;
}
__global__ void check(const unsigned *p, unsigned v, size_t n, unsigned *bad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != v + (unsigned)i) atomicAdd(bad, 1u);
}
__global__ void set_flag(unsigned *flag, unsigned v) {
  __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bounded spin: gives up after ~2 s so that a broken hand-over can never hang the box
__global__ void wait_flag(const unsigned *flag, unsigned v, unsigned *timed_out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
    __builtin_amdgcn_s_sleep(8);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { *timed_out = 1; break; }
  }
}
__global__ void tiny(unsigned *p) { if (threadIdx.x == 0) p[0] += 1; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// wait for a stream with a wall-clock bound (a wait value that never comes true must not hang the probe)
static bool stream_done_within(hipStream_t s, double sec) {
  const double t0 = now();
  while (now() - t0 < sec) {
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) { printf("  stream query: %s\n", hipGetErrorString(e)); return false; }
    usleep(200);
  }
  return false;
}

static void single_process() {
  bool ok = true;
  int can = -1;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned *sig = nullptr, *plain = nullptr, *scratch = nullptr;
  CK(hipExtMallocWithFlags((void **)&sig, 64, hipMallocSignalMemory));
  CK(hipMalloc((void **)&plain, 256));
  CK(hipMalloc((void **)&scratch, 256));
  CK(hipMemset(plain, 0, 256));
  CK(hipMemset(scratch, 0, 256));
  if (sig) CK(hipMemset(sig, 0, 8));
  for (int kind = 0; kind < 2; kind++) {
    unsigned *f = kind == 0 ? sig : plain;
    if (!f) continue;
    ok = true;
    printf("[1] wait/write value on %s memory\n", kind == 0 ? "SIGNAL" : "PLAIN hipMalloc");
    CK(hipStreamWaitValue32(sa, f, 5, hipStreamWaitValueGte, 0xFFFFFFFFu));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sa, scratch);
    usleep(20000);
    const bool early = hipStreamQuery(sa) == hipSuccess;
    CK(hipStreamWriteValue32(sb, f, 7, 0));
    const bool done = stream_done_within(sa, 3.0);
    printf("  waiter finished before the write: %d (want 0); finished after the write: %d (want 1)  api ok: %d\n", early, done, ok);
    if (!done) {  // release the waiter by a host-side memset so that teardown cannot hang
      unsigned v = 99;
      (void)hipMemcpy(f, &v, 4, hipMemcpyHostToDevice);
      stream_done_within(sa, 3.0);
    }
    // kernel-written flag releasing a stream wait
    CK(hipMemset(f, 0, 4));
    CK(hipStreamWaitValue32(sa, f, 3, hipStreamWaitValueGte, 0xFFFFFFFFu));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sa, scratch);
    hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, sb, f, 3u);
    printf("  kernel-written flag releases a stream wait: %d\n", stream_done_within(sa, 3.0));
    if (hipStreamQuery(sa) != hipSuccess) { unsigned v = 99; (void)hipMemcpy(f, &v, 4, hipMemcpyHostToDevice); stream_done_within(sa, 3.0); }
  }

  // [2] capture
  {
    ok = true;
    printf("[2] stream capture of write/wait value (plain memory)\n");
    CK(hipMemset(plain, 0, 8));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    hipError_t e1 = hipStreamWriteValue32(sa, plain, 1, 0);
    hipError_t e2 = hipStreamWaitValue32(sa, plain, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sa, scratch);
    hipError_t e3 = hipStreamEndCapture(sa, &g);
    printf("  in capture: write -> %s, wait -> %s, end -> %s\n", hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
    (void)hipGetLastError();
    if (e3 == hipSuccess && g) {
      size_t nn = 0;
      CK(hipGraphGetNodes(g, nullptr, &nn));
      printf("  captured graph has %zu nodes\n", nn);
      hipError_t e4 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      printf("  instantiate -> %s\n", hipGetErrorString(e4));
      if (e4 == hipSuccess) {
        CK(hipGraphLaunch(ge, sa));
        printf("  replay finished: %d\n", stream_done_within(sa, 3.0));
        hipGraphExecDestroy(ge);
      }
      hipGraphDestroy(g);
    }
    // explicit batch mem-op node
    printf("[2b] hipGraphAddBatchMemOpNode\n");
    ok = true;
    CK(hipGraphCreate(&g, 0));
    hipStreamBatchMemOpParams ops[2];
    memset(ops, 0, sizeof ops);
    ops[0].operation = hipStreamMemOpWriteValue32;
    ops[0].writeValue.address = (hipDeviceptr_t)plain;
    ops[0].writeValue.value = 11;
    ops[1].operation = hipStreamMemOpWaitValue32;
    ops[1].waitValue.address = (hipDeviceptr_t)plain;
    ops[1].waitValue.value = 11;
    ops[1].waitValue.flags = hipStreamWaitValueGte;
    hipBatchMemOpNodeParams np;
    memset(&np, 0, sizeof np);
    hipCtx_t ctx = nullptr;
    (void)hipCtxGetCurrent(&ctx);
    np.ctx = ctx;
    np.count = 2;
    np.paramArray = ops;
    hipGraphNode_t node = nullptr;
    hipError_t e5 = hipGraphAddBatchMemOpNode(&node, g, nullptr, 0, &np);
    printf("  add node -> %s\n", hipGetErrorString(e5));
    (void)hipGetLastError();
    if (e5 == hipSuccess) {
      hipError_t e6 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      printf("  instantiate -> %s\n", hipGetErrorString(e6));
      if (e6 == hipSuccess) {
        CK(hipGraphLaunch(ge, sa));
        printf("  replay finished: %d\n", stream_done_within(sa, 3.0));
        unsigned v = 0;
        CK(hipMemcpy(&v, plain, 4, hipMemcpyDeviceToHost));
        printf("  flag after replay = %u (want 11)\n", v);
        hipGraphExecDestroy(ge);
      }
    }
    hipGraphDestroy(g);
  }

  // [3] kernel-side flags inside a captured two-stream graph, replayed
  {
    ok = true;
    printf("[3] kernel-side set/wait flags in a captured fork/join graph\n");
    unsigned *to = nullptr;
    CK(hipMalloc((void **)&to, 4));
    CK(hipMemset(to, 0, 4));
    CK(hipMemset(plain, 0, 8));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, sa));
    CK(hipStreamWaitEvent(sb, fork, 0));
    hipLaunchKernelGGL(wait_flag, dim3(1), dim3(1), 0, sa, plain, 1u, to);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sa, scratch);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, scratch + 16);
    hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, sb, plain, 1u);
    CK(hipEventRecord(join, sb));
    CK(hipStreamWaitEvent(sa, join, 0));
    hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, sa, plain, 0u);  // reset for the next replay
    CK(hipStreamEndCapture(sa, &g));
    if (g) {
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      if (ge) {
        for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ge, sa));
        const bool fin = stream_done_within(sa, 10.0);
        unsigned t = 0;
        CK(hipMemcpy(&t, to, 4, hipMemcpyDeviceToHost));
        printf("  20 replays finished: %d  spin timed out: %u (want 0)\n", fin, t);
        const double t0 = now();
        for (int i = 0; i < 200; i++) CK(hipGraphLaunch(ge, sa));
        CK(hipStreamSynchronize(sa));
        printf("  replay of {wait-kernel, 2 tiny, set-kernel, reset} = %.2f us each\n", (now() - t0) / 200 * 1e6);
        hipGraphExecDestroy(ge);
      }
      hipGraphDestroy(g);
    }
    hipFree(to);
  }

  // [5] latency: stream A kernel -> write flag -> stream B wait -> kernel -> write flag -> A wait ... (ping-pong)
  for (int kind = 0; kind < 2; kind++) {
    unsigned *f = kind == 0 ? sig : plain;
    if (!f) continue;
    ok = true;
    CK(hipMemset(f, 0, 8));
    CK(hipDeviceSynchronize());
    const int N = 200;
    const double t0 = now();
    for (int i = 1; i <= N; i++) {
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sa, scratch);
      CK(hipStreamWriteValue32(sa, f, (unsigned)i, 0));
      CK(hipStreamWaitValue32(sb, f, (unsigned)i, hipStreamWaitValueGte, 0xFFFFFFFFu));
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, scratch + 16);
      CK(hipStreamWriteValue32(sb, f + 1, (unsigned)i, 0));
      CK(hipStreamWaitValue32(sa, f + 1, (unsigned)i, hipStreamWaitValueGte, 0xFFFFFFFFu));
    }
    const double t_issue = now() - t0;
    const bool fin = stream_done_within(sa, 10.0) && stream_done_within(sb, 10.0);
    printf("[5] ping-pong over %s flags: finished %d, %.2f us per round trip (2 kernels + 2 writes + 2 waits), host issue %.2f us\n",
           kind == 0 ? "SIGNAL" : "PLAIN", fin, (now() - t0) / N * 1e6, t_issue / N * 1e6);
    if (!fin) { unsigned v[2] = {1u << 30, 1u << 30}; (void)hipMemcpy(f, v, 8, hipMemcpyHostToDevice); stream_done_within(sa, 5.0); stream_done_within(sb, 5.0); }
  }
  // the same ping-pong with events (what the library does today between its two streams)
  {
    ok = true;
    hipEvent_t ea, eb;
    CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    CK(hipDeviceSynchronize());
    const int N = 200;
    const double t0 = now();
    for (int i = 1; i <= N; i++) {
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sa, scratch);
      CK(hipEventRecord(ea, sa));
      CK(hipStreamWaitEvent(sb, ea, 0));
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, scratch + 16);
      CK(hipEventRecord(eb, sb));
      CK(hipStreamWaitEvent(sa, eb, 0));
    }
    const double t_issue = now() - t0;
    CK(hipDeviceSynchronize());
    printf("[5b] ping-pong over events: %.2f us per round trip, host issue %.2f us\n", (now() - t0) / N * 1e6, t_issue / N * 1e6);
  }
  // kernel-side flags, eager
  {
    ok = true;
    unsigned *to = nullptr;
    CK(hipMalloc((void **)&to, 4));
    CK(hipMemset(to, 0, 4));
    CK(hipMemset(plain, 0, 8));
    CK(hipDeviceSynchronize());
    const int N = 200;
    const double t0 = now();
    for (int i = 1; i <= N; i++) {
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sa, scratch);
      hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, sa, plain, (unsigned)i);
      hipLaunchKernelGGL(wait_flag, dim3(1), dim3(1), 0, sb, plain, (unsigned)i, to);
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, sb, scratch + 16);
      hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, sb, plain + 1, (unsigned)i);
      hipLaunchKernelGGL(wait_flag, dim3(1), dim3(1), 0, sa, plain + 1, (unsigned)i, to);
    }
    const double t_issue = now() - t0;
    CK(hipDeviceSynchronize());
    unsigned t = 0;
    CK(hipMemcpy(&t, to, 4, hipMemcpyDeviceToHost));
    printf("[5c] ping-pong over kernel-side flags: %.2f us per round trip, host issue %.2f us, timed out %u\n", (now() - t0) / N * 1e6,
           t_issue / N * 1e6, t);
    hipFree(to);
  }
  hipFree(plain);
  hipFree(scratch);
  if (sig) hipFree(sig);
  hipStreamDestroy(sa);
  hipStreamDestroy(sb);
}

struct Wire {
  hipIpcMemHandle_t buf, flag, sigflag;
  int have_sig;
};

// [4] owner = parent, writer = child; both on device 0
static int ipc_owner(int rfd, int wfd) {
  bool ok = true;
  const size_t n = 1 << 20;
  unsigned *buf = nullptr, *flag = nullptr, *sig = nullptr, *bad = nullptr, *to = nullptr;
  CK(hipMalloc((void **)&buf, n * 4));
  CK(hipMalloc((void **)&flag, 256));
  CK(hipMalloc((void **)&bad, 4));
  CK(hipMalloc((void **)&to, 4));
  hipError_t es = hipExtMallocWithFlags((void **)&sig, 64, hipMallocSignalMemory);
  CK(hipMemset(buf, 0, n * 4));
  CK(hipMemset(flag, 0, 256));
  CK(hipMemset(bad, 0, 4));
  CK(hipMemset(to, 0, 4));
  if (es == hipSuccess) CK(hipMemset(sig, 0, 8));
  Wire w;
  memset(&w, 0, sizeof w);
  CK(hipIpcGetMemHandle(&w.buf, buf));
  CK(hipIpcGetMemHandle(&w.flag, flag));
  w.have_sig = 0;
  if (es == hipSuccess) {
    hipError_t e = hipIpcGetMemHandle(&w.sigflag, sig);
    printf("[4] owner: hipIpcGetMemHandle(signal memory) -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    w.have_sig = e == hipSuccess;
  }
  CK(hipDeviceSynchronize());
  if (write(wfd, &w, sizeof w) != (ssize_t)sizeof w) return 2;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // round 1: stream wait on the PLAIN flag written by the peer's hipStreamWriteValue32 (value 1)
  // round 2: kernel-side wait on the flag written by the peer's set_flag kernel (value 2)
  // round 3: stream wait on the SIGNAL flag (value 3), if it could be exported
  for (int round = 1; round <= 3; round++) {
    if (round == 3 && !w.have_sig) break;
    CK(hipMemsetAsync(bad, 0, 4, s));
    if (round == 1) CK(hipStreamWaitValue32(s, flag, 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
    if (round == 2) hipLaunchKernelGGL(wait_flag, dim3(1), dim3(1), 0, s, flag, 2u, to);
    if (round == 3) CK(hipStreamWaitValue32(s, sig, 3, hipStreamWaitValueGte, 0xFFFFFFFFu));
    hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, s, buf, 1000u * round, n, bad);
    char go = (char)round;
    if (write(wfd, &go, 1) != 1) return 2;
    const bool fin = stream_done_within(s, 8.0);
    unsigned nb = 0, t = 0;
    if (fin) {
      CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&t, to, 4, hipMemcpyDeviceToHost));
    } else {
      unsigned v = 1u << 30;
      (void)hipMemcpy(flag, &v, 4, hipMemcpyHostToDevice);
      if (sig) (void)hipMemcpy(sig, &v, 4, hipMemcpyHostToDevice);
      stream_done_within(s, 5.0);
    }
    printf("[4] owner round %d (%s): released %d, stale/missing words %u of %zu, kernel spin timed out %u\n", round,
           round == 1 ? "stream wait on plain flag <- peer hipStreamWriteValue32" : round == 2 ? "kernel wait <- peer set_flag kernel"
                                                                                             : "stream wait on signal flag <- peer write value",
           fin, nb, n, t);
    char ack = 0;
    if (read(rfd, &ack, 1) != 1) return 2;
  }
  hipStreamDestroy(s);
  hipFree(buf); hipFree(flag); hipFree(bad); hipFree(to);
  if (sig) hipFree(sig);
  return ok ? 0 : 1;
}

static int ipc_writer(int rfd, int wfd) {
  bool ok = true;
  Wire w;
  if (read(rfd, &w, sizeof w) != (ssize_t)sizeof w) return 2;
  const size_t n = 1 << 20;
  unsigned *buf = nullptr, *flag = nullptr, *sig = nullptr;
  CK(hipIpcOpenMemHandle((void **)&buf, w.buf, hipIpcMemLazyEnablePeerAccess));
  CK(hipIpcOpenMemHandle((void **)&flag, w.flag, hipIpcMemLazyEnablePeerAccess));
  if (w.have_sig) {
    hipError_t e = hipIpcOpenMemHandle((void **)&sig, w.sigflag, hipIpcMemLazyEnablePeerAccess);
    printf("[4] writer: hipIpcOpenMemHandle(signal memory) -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    if (e != hipSuccess) sig = nullptr;
  }
  printf("[4] writer: opened buf=%p flag=%p ok=%d\n", (void *)buf, (void *)flag, ok);
  fflush(stdout);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int round = 1; round <= 3; round++) {
    if (round == 3 && !w.have_sig) break;
    char go = 0;
    if (read(rfd, &go, 1) != 1) return 2;
    usleep(2000);  // let the owner's wait reach the GPU first
    if (buf) hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, s, buf, 1000u * round, n);
    if (round == 1 && flag) {
      hipError_t e = hipStreamWriteValue32(s, flag, 1, 0);
      if (e != hipSuccess) {
        printf("[4] writer: hipStreamWriteValue32(ipc flag) -> %s; falling back to a kernel write\n", hipGetErrorString(e));
        (void)hipGetLastError();
        hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, s, flag, 1u);
      }
    }
    if (round == 2 && flag) hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, s, flag, 2u);
    if (round == 3) {
      if (sig) {
        hipError_t e = hipStreamWriteValue32(s, sig, 3, 0);
        if (e != hipSuccess) { printf("[4] writer: write value to ipc signal -> %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
      }
    }
    CK(hipStreamSynchronize(s));
    char ack = 1;
    if (write(wfd, &ack, 1) != 1) return 2;
  }
  if (buf) CK(hipIpcCloseMemHandle(buf));
  if (flag) CK(hipIpcCloseMemHandle(flag));
  if (sig) CK(hipIpcCloseMemHandle(sig));
  hipStreamDestroy(s);
  return ok ? 0 : 1;
}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  if (argc > 1 && !strcmp(argv[1], "ipc")) {
    // fork BEFORE any HIP call: each process initialises the GPU by itself
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 2;
    pid_t pid = fork();
    if (pid == 0) {
      close(p2c[1]); close(c2p[0]);
      _exit(ipc_writer(p2c[0], c2p[1]));
    }
    close(p2c[0]); close(c2p[1]);
    int rc = ipc_owner(c2p[0], p2c[1]);
    int st = 0;
    waitpid(pid, &st, 0);
    printf("[4] owner rc %d, writer rc %d\n", rc, WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    return rc;
  }
  single_process();
  return 0;
}
