// One instantiation unit of the library: the template kernels of shard DG_SHARD (kernel_instances.inc), nothing else.  Built by
// deepgemm_amd/build.py as `hipcc -c -DDG_SHARD=<n> dg_shard.hip`, in parallel with its siblings and with dg_api.hip.
#ifndef DG_SHARD
#error "dg_shard.hip is compiled once per shard: -DDG_SHARD=<n>"
#endif
#define DG_SHARD_TU 1
#include <hip/hip_runtime.h>

#include "fp8_gemm_kernels.hpp"
#include "fp8_gemm_quad.hpp"
#include "fp8_gemm_moe.hpp"
#define DG_HAVE_MOE_HPP 1

namespace dg {
#define DG_KERNEL_INSTANCE(...) template __global__ __VA_ARGS__;
#include "kernel_instances.inc"
#undef DG_KERNEL_INSTANCE
}  // namespace dg
