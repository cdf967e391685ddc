// csrc/rt_launch.h — interface between the host library (rt_host.cpp) and the kernels (rt_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "rt_ir.h"

namespace rt {
// Shape of the instantiation that serves a scene: workgroup size, entries of each wave's camera-path queue (80 B each), and whether
// it runs as ONE workgroup per CU (BVH kernels: the CU's LDS then holds one copy of the top of the BVH, KParams::n_cached nodes).
struct LaunchShape { uint32_t threads, queue_entries; bool one_per_cu; };
LaunchShape pathtrace_shape(uint32_t scene_feats, uint32_t flags);
uint32_t pathtrace_feats(uint32_t scene_feats, uint32_t flags);      // the instantiation's FEATS template argument (its name: rt::pathtrace_kernel<T, FEATSu>)
// Dynamic LDS of one workgroup: [n_cached nodes][waves x queue][waves x stack_depth x 64 dwords]
inline size_t pathtrace_lds_bytes(const LaunchShape& g, uint32_t stack_depth, uint32_t n_cached, size_t node_bytes) {
    const size_t waves = g.threads / 64u;
    return (size_t)n_cached * node_bytes + waves * ((size_t)g.queue_entries * 80u + (size_t)stack_depth * 64u * sizeof(uint32_t));
}
// The kernels walk a BVH in the reference's order through the f32 filter nodes (rt_ir.h DFNode, rt_kernel.hip bvh_hit_filt): those are
// then what the LDS node cache holds.  The near-first order (opt-in) keeps the plain nodes.
inline bool filtered_walk(uint32_t effective_flags) { return !(effective_flags & 8u /* RT_NEAR_FIRST_BVH */); }
// Counter block the kernel reports into: RT_STATS_ROWS copies (row = block index mod rows) of RT_STATS_SLOTS 64-bit counters
static const uint32_t RT_STATS_SLOTS = 16u, RT_STATS_ROWS = 32u;
static const size_t RT_STATS_BYTES = (size_t)RT_STATS_SLOTS * RT_STATS_ROWS * sizeof(unsigned long long);
// Launch the persistent path-tracing kernel: n_blocks workgroups of pathtrace_shape().threads threads, `shmem` bytes of dynamic LDS (above).
template <typename T> hipError_t launch_pathtrace(const KParams<T>& P, uint32_t scene_feats, uint32_t n_blocks, size_t shmem, hipStream_t stream);
// Known-answer access to the list-scene kernels' closest-hit search (rt_debug_list_hit; the kernel lives in the lean translation unit)
hipError_t launch_list_hit_kat(const KParams<double>& P, uint32_t n, const double* d_rays, const double* d_tlim, double* d_out, hipStream_t stream);
// Resident blocks per CU for the instantiation that serves `scene_feats`.
template <typename T> int pathtrace_blocks_per_cu(uint32_t scene_feats, uint32_t flags, size_t shmem);
}
