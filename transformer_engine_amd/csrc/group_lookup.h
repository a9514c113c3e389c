// Device-side member lookup shared by the grouped kernels (grouped.hip, activation.hip): member shapes
// live in device memory (first_dims / last_dims / tensor_offsets), nothing here is read on the host.
#ifndef TE_AMD_GROUP_LOOKUP_H_
#define TE_AMD_GROUP_LOOKUP_H_

#include "common.h"
#include "device_utils.h"

namespace te {
namespace {

constexpr unsigned kMaxMembers = 1024;  // 256 threads x 4 members in the tile-count scan

// ---------------------------------------------------------------- member lookup (device)
struct Member {
  int64_t off, rows, cols;
};
__device__ __forceinline__ Member member_of(const GroupShapes &g, unsigned t) {
  Member m;
  m.rows = g.first_dims ? g.first_dims[t] : g.uniform_first;
  m.cols = g.last_dims ? g.last_dims[t] : g.uniform_last;
  m.off = g.offsets ? g.offsets[t] : static_cast<int64_t>(t) * g.uniform_first * g.uniform_last;
  return m;
}
// first element past the last member
__device__ __forceinline__ int64_t active_numel(const GroupShapes &g) {
  return g.offsets ? g.offsets[g.num_tensors]
                   : static_cast<int64_t>(g.num_tensors) * g.uniform_first * g.uniform_last;
}
// member that holds flat element i (0 <= i < active_numel): the LAST t with start(t) <= i, which
// steps over empty members
__device__ __forceinline__ unsigned member_holding(const GroupShapes &g, int64_t i) {
  if (!g.offsets) {
    const int64_t per = g.uniform_first * g.uniform_last;
    return per > 0 ? static_cast<unsigned>(i / per) : 0u;
  }
  unsigned lo = 0, hi = g.num_tensors - 1;
  while (lo < hi) {
    const unsigned mid = (lo + hi + 1) >> 1;
    if (g.offsets[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

constexpr int kGroupTile = 128;

// Which member and which of its 128x128 tiles does workgroup `b` own?  256 threads x 4 members:
// tile counts -> block-wide exclusive scan -> the owner announces itself through LDS.
// Returns false when b is past the last tile (the host sizes the grid from an upper bound).
// Up to 64 members (every MoE layer in practice): each wave finds the owner by itself - lane t reads member
// t's shape, a wave scan gives the tile prefix, a ballot the owner, and the owner's shape comes back through
// a lane read instead of a second, dependent round of loads.  No LDS, no barrier in front of the first data load.
__device__ __forceinline__ bool locate_tile_wave(const GroupShapes &g, unsigned b, unsigned &member, unsigned &local,
                                                 Member &m) {
  const unsigned lane = threadIdx.x & 63u;
  Member mine{0, 0, 0};
  unsigned c = 0;
  if (lane < g.num_tensors) {
    mine = member_of(g, lane);
    c = static_cast<unsigned>(((mine.rows + kGroupTile - 1) / kGroupTile) * ((mine.cols + kGroupTile - 1) / kGroupTile));
  }
  unsigned incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(incl, off, 64);
    if (lane >= static_cast<unsigned>(off)) incl += up;
  }
  const unsigned excl = incl - c;
  const unsigned long long owners = __ballot(c > 0 && b >= excl && b < incl);
  if (owners == 0ull) return false;
  const int src = __ffsll(static_cast<long long>(owners)) - 1;
  member = static_cast<unsigned>(src);
  local = b - __shfl(excl, src, 64);
  m.off = __shfl(mine.off, src, 64);
  m.rows = __shfl(mine.rows, src, 64);
  m.cols = __shfl(mine.cols, src, 64);
  return true;
}

__device__ __forceinline__ bool locate_tile(const GroupShapes &g, unsigned b, unsigned &member, unsigned &local) {
  __shared__ unsigned wave_sum[4];
  __shared__ unsigned found[2];
  const unsigned tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  unsigned cnt[4], mine = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned t = tid * 4 + j;
    unsigned c = 0;
    if (t < g.num_tensors) {
      const Member m = member_of(g, t);
      c = static_cast<unsigned>(((m.rows + kGroupTile - 1) / kGroupTile) * ((m.cols + kGroupTile - 1) / kGroupTile));
    }
    cnt[j] = c;
    mine += c;
  }
  unsigned incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(incl, off, 64);
    if (lane >= static_cast<unsigned>(off)) incl += up;
  }
  if (lane == 63) wave_sum[w] = incl;
  if (tid == 0) found[0] = 0xFFFFFFFFu;
  __syncthreads();
  unsigned run = incl - mine;
  for (unsigned k = 0; k < w; ++k) run += wave_sum[k];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (b >= run && b < run + cnt[j]) {
      found[0] = tid * 4 + j;
      found[1] = b - run;
    }
    run += cnt[j];
  }
  __syncthreads();
  member = found[0];
  local = found[1];
  return member != 0xFFFFFFFFu;
}

// Member and tile of workgroup b, with the member's shape: the wave form up to 64 members, the block scan beyond
__device__ __forceinline__ bool locate_tile(const GroupShapes &g, unsigned b, unsigned &member, unsigned &local,
                                            Member &m) {
  if (g.num_tensors <= 64u) return locate_tile_wave(g, b, member, local, m);
  if (!locate_tile(g, b, member, local)) return false;
  m = member_of(g, member);
  return true;
}

}  // namespace
}  // namespace te
#endif
