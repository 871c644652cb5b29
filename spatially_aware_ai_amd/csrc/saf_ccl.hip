// the connected-component core of `flood_fill_3d` (handy_utils.py:295-480).
//
// The reference walks the label grid in raster order (x outer, z inner; handy_utils.py:368-370), skips
// the null class and empty voxels (:383), flood-fills the 26-connected voxels of the same class
// (:312-344), rejects objects of fewer than `min_voxels` voxels (:390) and gives the k-th accepted object
// the index -2 - k in `voxel_obj_ids` (:352-353, :447-450; without a trained in-situ model).  Here:
// lock-free union-find with index-minimum hooking (the root of a component is its first voxel in raster
// order), a size count per root, an exclusive scan over the accepted roots for the raster-order numbering.
//
// HBM-bound integer work: every voxel reads its 13 forward neighbours' labels (L2 hits) and touches
// parent[] a few times; nothing here is MFMA-shaped.
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "saf_host.h"

namespace saf {
namespace {

__device__ __forceinline__ int ld_parent(const int* p, int i) {
  return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int find_root(const int* parent, int i) {
  for (;;) {
    const int p = ld_parent(parent, i);
    if (p == i) return i;
    i = p;
  }
}
// The same with path halving: every other node of the walked chain is re-pointed at its grandparent.
// A parent is always an ancestor with a smaller index, so a concurrent hook or another walker's store can
// only replace it by another ancestor of the same (eventual) component -- the forest stays a forest.
__device__ __forceinline__ int find_halving(int* parent, int i) {
  for (;;) {
    const int p = ld_parent(parent, i);
    if (p == i) return i;
    const int gp = ld_parent(parent, p);
    if (gp != p) __hip_atomic_store(parent + i, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    i = gp;
  }
}
// Roots only ever decrease, so the loop ends: hook the larger root under the smaller one.
__device__ __forceinline__ void unite(int* parent, int a, int b) {
  for (;;) {
    a = find_halving(parent, a);
    b = find_halving(parent, b);
    if (a == b) return;
    if (a > b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(&parent[b], a);
    if (old == b) return;
    b = old;
  }
}

__device__ __forceinline__ bool is_object(int label, int null_class) { return label != null_class && label != -1; }

__global__ __launch_bounds__(256) void ccl_init_kernel(const int* __restrict__ labels, int n, int null_class,
                                                        int* __restrict__ parent, int* __restrict__ count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  parent[i] = is_object(labels[i], null_class) ? i : -1;
  count[i] = 0;
}

// 13 forward neighbours (the other 13 are somebody else's forward neighbours).
__global__ __launch_bounds__(256) void ccl_union_kernel(const int* __restrict__ labels, int nx, int ny, int nz,
                                                         int null_class, int* __restrict__ parent) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = nx * ny * nz;
  if (i >= n) return;
  const int lab = labels[i];
  if (!is_object(lab, null_class)) return;
  const int z = i % nz, y = (i / nz) % ny, x = i / (nz * ny);
  int ri = find_halving(parent, i);  // kept across the neighbours: most of them are in i's set already
#pragma unroll
  for (int k = 14; k < 27; ++k) {  // (dx,dy,dz) after (0,0,0) in raster order
    const int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;
    const int xx = x + dx, yy = y + dy, zz = z + dz;
    if (xx < 0 || xx >= nx || yy < 0 || yy >= ny || zz < 0 || zz >= nz) continue;
    const int j = (xx * ny + yy) * nz + zz;
    if (labels[j] != lab) continue;
    const int rj = find_halving(parent, j);
    if (rj != ri) {
      unite(parent, ri, rj);
      ri = ri < rj ? ri : rj;  // an ancestor of both after the union (the true root may be smaller still)
    }
  }
}

// One round of pointer jumping: every chain gets half as deep, all voxels at once (coalesced).  A few
// rounds before the flatten kernel replace its long sequential walks by short ones.
__global__ __launch_bounds__(256) void ccl_jump_kernel(int n, int* __restrict__ parent) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int p = ld_parent(parent, i);
  if (p < 0 || p == i) return;
  const int gp = ld_parent(parent, p);
  if (gp != p) __hip_atomic_store(parent + i, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kCclSlots = 64;
__global__ __launch_bounds__(1024) void ccl_flatten_kernel(int n, int* __restrict__ parent, int* __restrict__ count) {
  const int i = blockIdx.x * 1024 + threadIdx.x;
  int r = -1;
  if (i < n && ld_parent(parent, i) >= 0) {
    r = find_halving(parent, i);
    // every voxel points at its root after this kernel (a root is the minimum index of its component,
    // so writing it cannot turn another voxel's chain into a cycle)
    __hip_atomic_store(parent + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // size count, aggregated per workgroup: neighbouring voxels mostly share their root, and a large object
  // would otherwise serialise hundreds of thousands of atomics on one address.  Per wave the lanes of one
  // root elect a leader (ballots); the leaders then merge through a small LDS table (linear probing) and
  // each distinct root of the workgroup costs one global atomic.
  __shared__ int s_key[kCclSlots], s_val[kCclSlots];
  for (int k = threadIdx.x; k < kCclSlots; k += 1024) {
    s_key[k] = -1;
    s_val[k] = 0;
  }
  __syncthreads();
  unsigned long long todo = __ballot(r >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int r0 = __builtin_amdgcn_readlane(r, leader);
    const unsigned long long same = __ballot(r == r0);
    if ((int)(threadIdx.x & 63) == leader) {
      const int add = __popcll(same);
      unsigned slot = ((unsigned)r0 * 2654435761u) % kCclSlots;
      bool placed = false;
      for (int probe = 0; probe < kCclSlots; ++probe) {
        const int old = atomicCAS(&s_key[slot], -1, r0);
        if (old == -1 || old == r0) {
          atomicAdd(&s_val[slot], add);
          placed = true;
          break;
        }
        slot = (slot + 1) % kCclSlots;
      }
      if (!placed) atomicAdd(&count[r0], add);  // table full (more distinct roots than slots): go direct
    }
    todo &= ~same;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kCclSlots; k += 1024)
    if (s_key[k] >= 0) atomicAdd(&count[s_key[k]], s_val[k]);
}

__global__ __launch_bounds__(256) void ccl_flag_kernel(int n, const int* __restrict__ parent, const int* __restrict__ count,
                                                        int min_voxels, int* __restrict__ flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  flag[i] = (parent[i] == i && count[i] >= min_voxels) ? 1 : 0;
}

__global__ __launch_bounds__(256) void ccl_emit_kernel(const int* __restrict__ labels, int n, const int* __restrict__ parent,
                                                        const int* __restrict__ count, const int* __restrict__ rank,
                                                        int min_voxels, int max_objects, int* __restrict__ obj_ids,
                                                        int* __restrict__ n_objects, int* __restrict__ obj_first,
                                                        int* __restrict__ obj_class, int* __restrict__ obj_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = parent[i];
  int id = -1;
  if (r >= 0 && count[r] >= min_voxels) {
    id = -2 - rank[r];
    if (r == i) {
      const int k = rank[r];
      if (k < max_objects) {
        if (obj_first) obj_first[k] = i;
        if (obj_class) obj_class[k] = labels[i];
        if (obj_count) obj_count[k] = count[i];
      }
      atomicMax(n_objects, k + 1);
    }
  }
  obj_ids[i] = id;
}

size_t scan_bytes(int n) {
  size_t b = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const int*)nullptr, (int*)nullptr, n);
  return (b + 255) & ~(size_t)255;
}

}  // namespace
}  // namespace saf

using namespace saf;

extern "C" {

size_t saf_label_components_workspace_bytes(int64_t n_voxels) {
  if (n_voxels <= 0 || n_voxels >= (1ll << 31)) return 0;
  const size_t a = ((size_t)n_voxels * sizeof(int) + 255) & ~(size_t)255;
  return 4 * a + scan_bytes((int)n_voxels);  // parent, count, flag, rank, scan temporaries
}

int saf_label_components(const int32_t* labels, int32_t nx, int32_t ny, int32_t nz, int32_t null_class,
                         int32_t min_voxels, int32_t* obj_ids, int32_t* n_objects, int32_t max_objects,
                         int32_t* obj_first, int32_t* obj_class, int32_t* obj_count, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!labels || !obj_ids || !n_objects || !workspace || nx <= 0 || ny <= 0 || nz <= 0 || min_voxels < 1 || max_objects < 0)
    return fail(SAF_E_INVALID, "label_components: bad arguments");
  const int64_t n64 = (int64_t)nx * ny * nz;
  if (n64 >= (1ll << 31)) return fail(SAF_E_UNSUPPORTED, "label_components: 2^31 voxels or more");
  const int n = (int)n64;
  if (workspace_bytes < saf_label_components_workspace_bytes(n64))
    return fail(SAF_E_INVALID, "label_components: workspace of %zu bytes, %zu needed", workspace_bytes,
                saf_label_components_workspace_bytes(n64));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t a = ((size_t)n * sizeof(int) + 255) & ~(size_t)255;
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  int* parent = reinterpret_cast<int*>(ws);
  int* count = reinterpret_cast<int*>(ws + a);
  int* flag = reinterpret_cast<int*>(ws + 2 * a);
  int* rank = reinterpret_cast<int*>(ws + 3 * a);
  void* tmp = ws + 4 * a;
  size_t tmp_bytes = scan_bytes(n);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (hipMemsetAsync(n_objects, 0, sizeof(int), s) != hipSuccess) return fail(SAF_E_HIP, "hipMemsetAsync(n_objects)");
  hipLaunchKernelGGL(ccl_init_kernel, dim3(blocks), dim3(256), 0, s, labels, n, null_class, parent, count);
  hipLaunchKernelGGL(ccl_union_kernel, dim3(blocks), dim3(256), 0, s, labels, nx, ny, nz, null_class, parent);
  for (int round = 0; round < 4; ++round) hipLaunchKernelGGL(ccl_jump_kernel, dim3(blocks), dim3(256), 0, s, n, parent);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, n, parent, count);
  hipLaunchKernelGGL(ccl_flag_kernel, dim3(blocks), dim3(256), 0, s, n, parent, count, min_voxels, flag);
  int rc = check_launch("ccl kernels");
  if (rc) return rc;
  if (hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flag, rank, n, s) != hipSuccess)
    return fail(SAF_E_HIP, "label_components: scan failed");
  hipLaunchKernelGGL(ccl_emit_kernel, dim3(blocks), dim3(256), 0, s, labels, n, parent, count, rank, min_voxels, max_objects,
                     obj_ids, n_objects, obj_first, obj_class, obj_count);
  return check_launch("ccl_emit_kernel");
}

}  // extern "C"
