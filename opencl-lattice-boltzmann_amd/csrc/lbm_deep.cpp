// Second translation unit of liblbm_hip.so: the deep window kernels (d2q9_deep, d2q9_deep_twin), compiled with
// -mllvm -amdgpu-sched-strategy=max-ilp (see deep_instances.h and the Makefile).  Nothing but the instantiations.
#define LBM_DEEP_DEFINE
#include "deep_instances.h"

namespace lbm {
#define LBM_DEEP_INSTANTIATE(...) template __global__ void __VA_ARGS__(const Step2Args, float *, int, int);
LBM_DEEP_INSTANCES(LBM_DEEP_INSTANTIATE)
#undef LBM_DEEP_INSTANTIATE
}  // namespace lbm
