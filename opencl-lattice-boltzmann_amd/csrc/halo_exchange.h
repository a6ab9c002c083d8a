// Grouped ring exchange of halo rows over a send/recv transport (RCCL in the product) — free of HIP and RCCL types
// so that its error handling is testable on a CPU-only box with a mock transport (tests/cpu/halo_exchange_test.cpp).
//
// One call moves, for every local slab, its top rows to the north neighbour and its bottom rows to the south
// neighbour and receives the matching rows from both (the ring is periodic, kernels.cl:91-93), all
// inside ONE transport group.  The contract this file exists for: whatever fails in the middle, the group that was
// opened is CLOSED before the function returns, nothing is issued after the first failure, and the first failure
// is what the caller gets to see.
#pragma once

#include <cstddef>

namespace lbm {

struct HaloBlock {
  const void *send_north;  // this slab's top `count` elements of owned rows   -> rank `north`
  const void *send_south;  // this slab's bottom rows                          -> rank `south`
  void *recv_south;        // halo rows below the owned rows                   <- rank `south`
  void *recv_north;        // halo rows above the owned rows                   <- rank `north`
  size_t count;            // elements per block
  int north, south;        // ring neighbours
  void *comm;              // the slab's communicator
  void *stream;            // the stream the transfers are ordered on
};

// Transport concept: int group_start(); int group_end(); int send(const void*, size_t, int peer, void *comm, void *stream);
// int recv(void*, size_t, int peer, void *comm, void *stream); all return 0 on success, a transport error code otherwise.
template <class Transport>
int ring_exchange(Transport &t, const HaloBlock *blocks, int nblocks, const char **failed_op) {
  if (failed_op) *failed_op = nullptr;
  if (int rc = t.group_start()) {
    if (failed_op) *failed_op = "group_start";
    return rc;
  }
  int first = 0;
  auto note = [&](int rc, const char *what) {
    if (rc && !first) {
      first = rc;
      if (failed_op) *failed_op = what;
    }
  };
  for (int i = 0; i < nblocks && !first; i++) {
    const HaloBlock &b = blocks[i];
    note(t.send(b.send_north, b.count, b.north, b.comm, b.stream), "send north");
    if (!first) note(t.send(b.send_south, b.count, b.south, b.comm, b.stream), "send south");
    if (!first) note(t.recv(b.recv_south, b.count, b.south, b.comm, b.stream), "recv south");
    if (!first) note(t.recv(b.recv_north, b.count, b.north, b.comm, b.stream), "recv north");
  }
  // always close the group: a group left open poisons every later collective of the process
  const int rc_end = t.group_end();
  if (!first) note(rc_end, "group_end");
  return first;
}

}  // namespace lbm
