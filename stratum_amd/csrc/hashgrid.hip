// hashgrid.hip — the spatial hash grid of the reservoir reuse, BUILT on the device in the order hashgrid.h defines.
//
// Upstream builds the grid with atomics inside the path kernels and two small dispatches afterwards
// (src/Shaders/common/hashgrid.hlsli:43-88 find_or_insert / append / compute_indices / swizzle; host side
// src/Node/BDPT.cpp:723-754), so which bucket a cell gets when cells compete, and the order of a bucket's records, are left
// to the scheduler. The order DEFINED for this library (hashgrid.h): the appends take effect in (path index, diffuse vertex)
// order, each running find_or_insert's 32-step linear probe against the table as the earlier appends left it. Round 2 ran
// that probe sequence on the host (two synchronisations per seed). Here it runs on the device, in parallel, and gives the
// same table:
//
//   * In the serial run an append k claims the first slot of its window [home, home + 32) that is empty or already holds
//     its checksum, and nothing ever moves afterwards. Only the FIRST append of a cell — of a (home, checksum) pair — can
//     claim a slot; the later ones find it (or, if it was dropped, are dropped like it: a full window stays full). So the
//     table is that of linear probing with the cells inserted in the order of their first appends, and that table is the
//     one in which every cell sits at the first slot of its window that no EARLIER cell holds. It is reached in parallel
//     by priority insertion (Shun and Blelloch's phase-concurrent linear probing): the first appends (found with a stable
//     sort by key) insert themselves with atomicMin on the slot's owner index; a cell that loses a slot to an earlier one
//     moves on to the next slot of its own window, and is dropped when the window ends, as find_or_insert drops it. A
//     slot's owner index only ever decreases, so a slot that blocked a cell blocks it for good, dropped cells leave no
//     hole, and the order the atomics resolve in does not matter.
//   * Matching is by checksum alone upstream, so two DIFFERENT cells with the same 32-bit checksum can merge when their
//     windows overlap (homes less than 32 apart). A million cells make a hundred such checksum pairs per frame, but only one
//     frame in twenty-five has a pair that close. The sort brings equal checksums together, so these cells are found there;
//     they stay out of the parallel insertion and one thread inserts them afterwards, in the order of their first appends,
//     with the serial rule itself (stop at an earlier slot holding the checksum) — treating slots of LATER cells as free and
//     re-inserting whoever it displaces, which is what the table would look like had they come in their turn. (More than
//     HG_SPECIAL_MAX such cells: the whole table is rebuilt by the one-thread serial probe sequence. Exact always.)
//   * With the table final, every append looks its slot up (first slot of its window holding its checksum: exactly the
//     slot the serial run found for it, or none), the per-bucket counters are plain counts, their exclusive scan gives the
//     bucket ranges (compute_indices), and an append's place inside its bucket's range is its rank among that bucket's
//     appends in append order: a stable radix sort of (bucket, append index) pairs by bucket — the position of a pair IS
//     the destination (swizzle).
// No host hop and no synchronisation: the append count stays on the device, every kernel bounds itself by it.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

namespace sthip {
namespace {

#define HG_EMPTY 0xFFFFFFFFu
#define HG_WINDOW 32u
#define HG_SPECIAL_MAX 1024u  // cells that share a checksum with a cell less than a window away, per grid (expected: a handful)

__global__ void __launch_bounds__(256) k_hg_clear(uint32_t* owner, uint32_t* counters, uint32_t buckets) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < buckets) {
    owner[b] = HG_EMPTY;
    counters[b] = 0u;
  }
}
// key of an append for the first sort: checksum in the high word, home bucket in the low one (padding sorts last)
__global__ void __launch_bounds__(256) k_hg_sort_keys(const uint2* keys, const uint32_t* count, uint32_t slots, unsigned long long* key64, uint32_t* append_index, uint32_t* ctl, uint32_t force_serial) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) {
    ctl[0] = force_serial;
    ctl[1] = 0u;
  }
  if (k >= slots) return;
  append_index[k] = k;
  key64[k] = k < *count ? ((unsigned long long)keys[k].y << 32) | keys[k].x : ~0ull;
}
// priority insertion of the first append of every cell: keys[k] = (home bucket, checksum) of append k, owner[b] = index of
// the append that owns slot b. The sort is stable, so the first entry of a run of equal keys is the cell's earliest append.
// Cells with a same-checksum neighbour less than a window away go to the `special` list instead (k_hg_special).
// ctl: [0] = 1: rebuild serially (forced, or the list overflowed), [1] = entries in the list
__global__ void __launch_bounds__(256) k_hg_insert(const uint2* keys, const uint32_t* count, uint32_t slots, const unsigned long long* sorted_key, const uint32_t* sorted_append, uint32_t* owner,
                                                   uint32_t* ctl, uint32_t* special) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= slots) return;
  uint32_t v = sorted_append[j];
  if (v >= *count) return;
  const unsigned long long mine = sorted_key[j];
  if (j > 0 && sorted_key[j - 1] == mine) return;  // a later append of the same cell
  // the neighbouring cells in (checksum, home) order: the previous entry, and the first entry with another key
  bool near = false;
  if (j > 0) {
    const unsigned long long prev = sorted_key[j - 1];
    near = (uint32_t)(prev >> 32) == (uint32_t)(mine >> 32) && (uint32_t)mine - (uint32_t)prev < HG_WINDOW;
  }
  for (uint32_t q = j + 1; q < slots && !near; q++) {
    const unsigned long long next = sorted_key[q];
    if (next == mine) continue;
    near = (uint32_t)(next >> 32) == (uint32_t)(mine >> 32) && (uint32_t)next - (uint32_t)mine < HG_WINDOW;
    break;
  }
  if (near) {
    const uint32_t e = atomicAdd(&ctl[1], 1u);
    if (e < HG_SPECIAL_MAX)
      special[e] = v;
    else
      ctl[0] = 1u;
    return;
  }
  uint2 kv = keys[v];
  uint32_t i = kv.x;
  while (i < kv.x + HG_WINDOW) {
    const uint32_t old = atomicMin(&owner[i], v);
    if (old == HG_EMPTY || old == v) return;  // claimed an empty slot
    if (old > v) {  // the slot was a later cell's: that one is displaced and goes on probing from the next slot of ITS window
      v = old;
      kv = keys[v];
    }
    i++;
  }
  // the window is exhausted: dropped (find_or_insert returns -1, hashgrid.hlsli:56-58)
}
// One thread: the cells of the special list, in the order of their first appends, with the serial rule; or (ctl[0]) the whole
// serial probe sequence.
__global__ void k_hg_special(const uint2* keys, const uint32_t* count, uint32_t buckets, uint32_t* owner, const uint32_t* ctl, uint32_t* special) {
  if (ctl[0]) {
    for (uint32_t b = 0; b < buckets; b++) owner[b] = HG_EMPTY;
    for (uint32_t k = 0, n = *count; k < n; k++) {
      const uint2 kv = keys[k];
      for (uint32_t i = 0; i < HG_WINDOW; i++) {
        const uint32_t o = owner[kv.x + i];
        if (o == HG_EMPTY) {
          owner[kv.x + i] = k;
          break;
        }
        if (keys[o].y == kv.y) break;
      }
    }
    return;
  }
  const uint32_t n = ctl[1];
  for (uint32_t a = 1; a < n; a++) {  // by first append (a handful of entries)
    const uint32_t x = special[a];
    uint32_t b = a;
    for (; b > 0 && special[b - 1] > x; b--) special[b] = special[b - 1];
    special[b] = x;
  }
  for (uint32_t a = 0; a < n; a++) {
    uint32_t v = special[a];
    uint2 kv = keys[v];
    uint32_t i = kv.x;
    while (i < kv.x + HG_WINDOW) {
      const uint32_t o = owner[i];
      if (o == HG_EMPTY) {
        owner[i] = v;
        break;
      }
      if (o < v) {  // an earlier cell's slot: its checksum is a match (the later cell merges: done), or the probe goes on
        if (keys[o].y == kv.y) break;
      } else {  // a later cell's: as good as free when this cell's turn came; the later one is inserted again behind it
        owner[i] = v;
        v = o;
        kv = keys[v];
      }
      i++;
    }
  }
}
// the table as the lookups of the next seed read it (checksum per slot, 0 = empty), every append's bucket, the bucket counters
__global__ void __launch_bounds__(256) k_hg_checksums(const uint2* keys, const uint32_t* owner, uint32_t buckets, uint32_t* checksums) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < buckets) checksums[b] = owner[b] == HG_EMPTY ? 0u : keys[owner[b]].y;
}
__global__ void __launch_bounds__(256) k_hg_lookup(const uint2* keys, const uint32_t* count, const uint32_t* checksums, uint32_t slots, uint32_t* bucket_of, uint32_t* append_index,
                                                   uint32_t* counters) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= slots) return;
  append_index[k] = k;
  if (k >= *count) {
    bucket_of[k] = HG_EMPTY;  // (padding of the sort: behind every real bucket)
    return;
  }
  const uint2 kv = keys[k];
  uint32_t found = HG_EMPTY;
  for (uint32_t i = 0; i < HG_WINDOW; i++)
    if (checksums[kv.x + i] == kv.y) {
      found = kv.x + i;
      break;
    }
  bucket_of[k] = found;
  if (found != HG_EMPTY) atomicAdd(&counters[found], 1u);
}
// dest[append] = its position in the stable sort by bucket = indices[bucket] + rank inside the bucket; dropped appends: none
__global__ void __launch_bounds__(256) k_hg_dest(const uint32_t* sorted_bucket, const uint32_t* sorted_append, const uint32_t* count, uint32_t slots, uint32_t* dest) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= slots) return;
  const uint32_t k = sorted_append[j];
  if (k < *count) dest[k] = sorted_bucket[j] == HG_EMPTY ? HG_EMPTY : j;
}

}  // namespace

// keys: (home bucket, checksum) per compacted append, `count` of them (device), at most `slots`. buckets = gHashGridBucketCount
// + 32 (probing does not wrap). Outputs (device): checksums / counters / indices per bucket (the table the next seed reads),
// dest per append. scratch: owner[buckets + 2 + HG_SPECIAL_MAX] (behind the table: control words and the special list), four arrays of `slots` uint32
// (bucket_of, append_index and their sorted copies), two of `slots` 64-bit words (the first sort's keys), hipCUB's temporary
// storage (query with tmp == nullptr: returns the bytes needed in tmp_bytes). force_serial: take the one-thread rebuild
// whatever the keys are (tests). Enqueued on `stream`.
hipError_t hashgrid_build_device(const uint2* keys, const uint32_t* count, uint32_t slots, uint32_t buckets, uint32_t* checksums, uint32_t* counters, uint32_t* indices, uint32_t* dest, uint32_t* owner,
                                 uint32_t* bucket_of, uint32_t* append_index, uint32_t* sorted_bucket, uint32_t* sorted_append, unsigned long long* key64, unsigned long long* sorted_key64, void* tmp,
                                 size_t& tmp_bytes, hipStream_t stream, bool force_serial) {
  if (!tmp) {
    size_t a = 0, b = 0, c = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, a, counters, indices, (int)buckets, stream);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, b, bucket_of, sorted_bucket, append_index, sorted_append, (int)slots, 0, 32, stream);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, c, key64, sorted_key64, append_index, sorted_append, (int)slots, 0, 64, stream);
    if (e != hipSuccess) return e;
    tmp_bytes = a > b ? a : b;
    tmp_bytes = tmp_bytes > c ? tmp_bytes : c;
    return hipSuccess;
  }
  if (!slots || !buckets) return hipSuccess;
  const unsigned bgrid = (buckets + 255) / 256, sgrid = (slots + 255) / 256;
  uint32_t* ctl = owner + buckets;  // [0] serial rebuild, [1] special cells, [2 ...] their list
  uint32_t* special = ctl + 2;
  hipLaunchKernelGGL(k_hg_clear, dim3(bgrid), dim3(256), 0, stream, owner, counters, buckets);
  hipLaunchKernelGGL(k_hg_sort_keys, dim3(sgrid), dim3(256), 0, stream, keys, count, slots, key64, append_index, ctl, force_serial ? 1u : 0u);
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key64, sorted_key64, append_index, sorted_append, (int)slots, 0, 64, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_hg_insert, dim3(sgrid), dim3(256), 0, stream, keys, count, slots, sorted_key64, sorted_append, owner, ctl, special);
  hipLaunchKernelGGL(k_hg_special, dim3(1), dim3(1), 0, stream, keys, count, buckets, owner, ctl, special);
  hipLaunchKernelGGL(k_hg_checksums, dim3(bgrid), dim3(256), 0, stream, keys, owner, buckets, checksums);
  hipLaunchKernelGGL(k_hg_lookup, dim3(sgrid), dim3(256), 0, stream, keys, count, checksums, slots, bucket_of, append_index, counters);
  e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, counters, indices, (int)buckets, stream);
  if (e != hipSuccess) return e;
  e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, bucket_of, sorted_bucket, append_index, sorted_append, (int)slots, 0, 32, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_hg_dest, dim3(sgrid), dim3(256), 0, stream, sorted_bucket, sorted_append, count, slots, dest);
  return hipGetLastError();
}

}  // namespace sthip
