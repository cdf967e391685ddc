// csrc/rt_launch.h — interface between the host library (rt_host.cpp) and the kernels (rt_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "rt_ir.h"

namespace rt {
// LDS the kernel needs besides the BVH stacks: four per-wave camera-path regeneration queues
static const size_t RT_REGEN_LDS_BYTES = 4u * (7u * 64u * 8u + 6u * 64u * 4u);
// Counter block the kernel reports into: RT_STATS_ROWS copies (row = block index mod rows) of RT_STATS_SLOTS 64-bit counters
static const uint32_t RT_STATS_SLOTS = 16u, RT_STATS_ROWS = 32u;
static const size_t RT_STATS_BYTES = (size_t)RT_STATS_SLOTS * RT_STATS_ROWS * sizeof(unsigned long long);
// Launch the persistent path-tracing kernel: n_blocks blocks of 256 threads, `shmem` bytes of LDS for BVH stacks.
template <typename T> hipError_t launch_pathtrace(const KParams<T>& P, uint32_t scene_feats, uint32_t n_blocks, size_t shmem, hipStream_t stream);
// Resident blocks per CU for the instantiation that serves `scene_feats`.
template <typename T> int pathtrace_blocks_per_cu(uint32_t scene_feats, uint32_t flags, size_t shmem);
}
