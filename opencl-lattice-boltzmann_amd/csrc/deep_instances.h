// The instantiations of the deep window kernels that liblbm_hip.so launches (lbm_hip.cpp: launch_deep, launch_deep_compact,
// launch_deep_twin), as one list.  They are compiled in a translation unit of their own (lbm_deep.cpp) with the compiler's
// max-ILP scheduling strategy — it spaces dependent packed operations instead of padding them with s_nop: same-box A/B
// (profiles/r03_sched_strategy.txt) 8192x8192 468 -> 476 GLUPS, 4096x4096 377 -> 384, 1024x1024 177 -> 182, bit-identical —
// which the rest of the library must not get: d2q9_step4p, at 254 VGPRs, spills 7 registers under it.  lbm_hip.cpp sees the
// list as `extern template` declarations, lbm_deep.cpp as explicit instantiation definitions.
#pragma once
#include "d2q9_kernels.h"

// X(kernel-id with its template arguments): all share the signature (Step2Args, float *partials, int pstride, int nlev)
#define LBM_DEEP_INSTANCES(X)                                                                                        \
  /* d2q9_deep<D, NT, OBST_PATHS, PUSH, LT>: one slab without pairs / the two-stream launch sets of row slabs */     \
  X(d2q9_deep<8, true, true, false, 8>) X(d2q9_deep<8, true, true, false, 7>) X(d2q9_deep<8, true, true, false, 6>)  \
  X(d2q9_deep<8, true, true, false, 0>) X(d2q9_deep<8, true, false, false, 0>) X(d2q9_deep<8, false, true, false, 0>) \
  X(d2q9_deep<8, false, false, false, 0>)                                                                            \
  /* ... compact launch sets of row slabs, lone interior */                                                          \
  X(d2q9_deep<8, true, true, true, 8>) X(d2q9_deep<8, true, true, true, 7>) X(d2q9_deep<8, true, true, true, 6>)     \
  X(d2q9_deep<8, true, true, true, 0>) X(d2q9_deep<8, true, false, true, 0>) X(d2q9_deep<8, false, true, true, 0>)   \
  X(d2q9_deep<8, false, false, true, 0>)                                                                             \
  /* d2q9_deep_twin<D, NT, OBST_PATHS, LT, PUSH>: compact launch sets, interior chunk pairs */                       \
  X(d2q9_deep_twin<8, true, true, 8, true>) X(d2q9_deep_twin<8, true, true, 7, true>)                                \
  X(d2q9_deep_twin<8, true, true, 6, true>) X(d2q9_deep_twin<8, true, true, 0, true>)                                \
  X(d2q9_deep_twin<8, true, false, 0, true>) X(d2q9_deep_twin<8, false, true, 0, true>)                              \
  X(d2q9_deep_twin<8, false, false, 0, true>)                                                                        \
  /* ... one slab, up to five timesteps per launch (300K to 3M cells) */                                             \
  X(d2q9_deep_twin<5, true, true, 5, false>) X(d2q9_deep_twin<5, false, true, 5, false>)                             \
  X(d2q9_deep_twin<5, true, true, 0, false>) X(d2q9_deep_twin<5, true, false, 0, false>)                             \
  X(d2q9_deep_twin<5, false, true, 0, false>) X(d2q9_deep_twin<5, false, false, 0, false>)                           \
  /* ... compact launch sets of row slabs of 240K to 3M cells (five halo rows): interior chunk pairs + edge chunk pairs */ \
  X(d2q9_deep_twin<5, false, true, 5, true>) X(d2q9_deep_twin<5, false, true, 0, true>)                              \
  X(d2q9_deep_twin<5, false, false, 0, true>)                                                                        \
  /* ... one slab, up to eight (from 3M cells) */                                                                    \
  X(d2q9_deep_twin<8, true, true, 8, false>) X(d2q9_deep_twin<8, true, true, 7, false>)                              \
  X(d2q9_deep_twin<8, true, true, 6, false>) X(d2q9_deep_twin<8, true, true, 0, false>)                              \
  X(d2q9_deep_twin<8, true, false, 0, false>) X(d2q9_deep_twin<8, false, true, 0, false>)                            \
  X(d2q9_deep_twin<8, false, false, 0, false>)

namespace lbm {
#ifndef LBM_DEEP_DEFINE
#define LBM_DEEP_DECLARE(...) extern template __global__ void __VA_ARGS__(const Step2Args, float *, int, int);
LBM_DEEP_INSTANCES(LBM_DEEP_DECLARE)
#undef LBM_DEEP_DECLARE
#endif
}  // namespace lbm
