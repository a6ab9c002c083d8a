// CPU-side test of csrc/halo_exchange.h with a mock transport that fails on demand (tests/test_halo_exchange.py
// compiles and runs it with g++; no HIP, no RCCL).  Checks the contract of ring_exchange: the group that was opened
// is always closed, nothing is issued after the first failure, the first failure is what the caller sees, and on
// success every slab's four transfers are issued with the ring neighbours of its position.
#include "../../opencl-lattice-boltzmann_amd/csrc/halo_exchange.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Op {
  char kind;  // 's' / 'r'
  const void *ptr;
  size_t n;
  int peer;
};

struct Mock {
  int fail_at = -1;        // index of the send/recv call that fails (-1: none)
  int fail_start = 0, fail_end = 0;
  int depth = 0;           // open groups
  int starts = 0, ends = 0, calls = 0;
  std::vector<Op> ops;
  int group_start() { starts++; if (fail_start) return fail_start; depth++; return 0; }
  int group_end() { ends++; depth--; return fail_end; }
  int op(char k, const void *p, size_t n, int peer) {
    const int i = calls++;
    if (i == fail_at) return 7;
    ops.push_back({k, p, n, peer});
    return 0;
  }
  int send(const void *p, size_t n, int peer, void *, void *) { return op('s', p, n, peer); }
  int recv(void *p, size_t n, int peer, void *, void *) { return op('r', p, n, peer); }
};

static int failures = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) { printf("FAILED line %d: %s\n", __LINE__, #cond); failures++; } \
  } while (0)

int main() {
  const int P = 4;
  std::vector<std::vector<float>> grids(P, std::vector<float>(100));
  std::vector<lbm::HaloBlock> blocks(P);
  for (int i = 0; i < P; i++) {
    float *g = grids[i].data();
    blocks[i] = lbm::HaloBlock{g + 80, g + 10, g, g + 90, 10, (i + 1) % P, (i + P - 1) % P, nullptr, nullptr};
  }
  {  // success: 4 transfers per slab, ring neighbours, one balanced group
    Mock t;
    const char *op = "x";
    EXPECT(lbm::ring_exchange(t, blocks.data(), P, &op) == 0);
    EXPECT(op == nullptr && t.starts == 1 && t.ends == 1 && t.depth == 0 && (int)t.ops.size() == 4 * P);
    for (int i = 0; i < P; i++) {
      const Op *o = &t.ops[4 * i];
      EXPECT(o[0].kind == 's' && o[0].peer == (i + 1) % P && o[0].ptr == grids[i].data() + 80);
      EXPECT(o[1].kind == 's' && o[1].peer == (i + P - 1) % P && o[1].ptr == grids[i].data() + 10);
      EXPECT(o[2].kind == 'r' && o[2].peer == (i + P - 1) % P && o[2].ptr == grids[i].data());
      EXPECT(o[3].kind == 'r' && o[3].peer == (i + 1) % P && o[3].ptr == grids[i].data() + 90 && o[3].n == 10);
    }
  }
  for (int fail_at = 0; fail_at < 4 * P; fail_at++) {  // every possible first failure
    Mock t;
    t.fail_at = fail_at;
    const char *op = nullptr;
    const int rc = lbm::ring_exchange(t, blocks.data(), P, &op);
    EXPECT(rc == 7);                                   // the transport's code comes back
    EXPECT(t.depth == 0 && t.starts == 1 && t.ends == 1);  // the group is CLOSED (round 1 returned with it open)
    EXPECT(t.calls == fail_at + 1);                    // nothing issued after the failure
    static const char *const names[4] = {"send north", "send south", "recv south", "recv north"};
    EXPECT(op != nullptr && std::string(op) == names[fail_at % 4]);
  }
  {  // the close itself fails
    Mock t;
    t.fail_end = 9;
    const char *op = nullptr;
    EXPECT(lbm::ring_exchange(t, blocks.data(), P, &op) == 9 && std::string(op) == "group_end" && t.depth == 0);
  }
  {  // a failing transfer wins over a failing close
    Mock t;
    t.fail_at = 5;
    t.fail_end = 9;
    const char *op = nullptr;
    EXPECT(lbm::ring_exchange(t, blocks.data(), P, &op) == 7 && std::string(op) == "send south" && t.ends == 1);
  }
  {  // the open fails: nothing else happens, nothing to close
    Mock t;
    t.fail_start = 3;
    const char *op = nullptr;
    EXPECT(lbm::ring_exchange(t, blocks.data(), P, &op) == 3 && std::string(op) == "group_start" && t.calls == 0 && t.ends == 0);
  }
  printf(failures ? "halo_exchange_test: %d FAILED\n" : "halo_exchange_test: ok\n", failures);
  return failures ? 1 : 0;
}
