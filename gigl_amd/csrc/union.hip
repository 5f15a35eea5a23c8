// union.hip — batch union graph ("collate") on the GPU: node dedup + level-ordered relabel + edge
// dedup + CSR-by-destination build.  No host synchronisation; all counts stay in HBM.
//
// Replaces (paths relative to the reference root):
//   GraphBuilder.__remap_node / add_graph_data / add_edge(skip_if_exists)
//       python/gigl/src/common/graph_builder/abstract_graph_builder.py:16-24,49-100,102-150
//   RootedNodeNeighborhoodBatch.collate_pyg_rooted_node_neighborhood_minibatch + coalesce()
//       python/gigl/src/training/v1/lib/data_loaders/rooted_node_neighborhood_data_loader.py:78-158
//   SupervisedNodeClassificationBatch.collate_pyg_node_classification_minibatch
//       python/gigl/src/training/v1/lib/data_loaders/supervised_node_classification_data_loader.py:74-117
//
// Pipeline (T = b + sum slots[k] stream positions; roots first, then hop slots in order), 9 launches,
// no library sort/scan:
//   1 insert_roots   open-addressing table keyed by node id (atomicCAS); first stream position by
//                    atomicMin; level 0
//   2 insert_slots   same for every sampled slot.  hops <= 2: the level is final here (a hop-1 slot
//                    is level 1; a hop-2 slot is level 1 iff its parent is a root, which step 1 made
//                    visible).  hops > 2: levels are relaxed by `hops` extra rounds.
//   3 count          per 1024-position tile: number of first occurrences per level
//   4 assign         local id = level base + tile prefix + rank inside the tile (ballot/popcount);
//                    deterministic: (level, first stream position) order
//   5 edge_count     rowcnt[dst_local]++ for every sampled edge occurrence (duplicates included)
//   6 row_scan       exclusive scan of rowcnt -> rowptr (one workgroup; only rows that can have edges)
//   7 edge_fill      col[rowptr[dst] + cursor++] = src_local
//   8 row_sort       every row sorted ascending + deduplicated IN PLACE: one wave per row (<= 64
//                    entries, rank-by-counting with v_readlane), rows > 64 queued for
//   9 row_sort_big   one workgroup per queued row, bitonic sort in LDS (<= 16384 entries)
// Rows keep their pre-dedup capacity: row i is col[rowptr[i] .. rowend[i]).
#include "common.h"

#include <mutex>

#include <cstdlib>

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: a process that drives several GPUs (lane
// engines, in-process worlds on different devices) must opt in on each of them — done once per (device, site), and
// UNDER a lock (ctxs of different host threads build union graphs concurrently: the second one must not launch before
// the first one's opt-in has happened)
static hipError_t lds_opt_in(int device, int site, const void* fn, int bytes, const void* fn2 = nullptr, int bytes2 = 0) {
  static std::mutex mu;
  static uint32_t done[256] = {0};
  std::lock_guard<std::mutex> lk(mu);
  const uint32_t bit = 1u << site;
  uint32_t& d = done[device & 255];
  if (d & bit) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && fn2) e = hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, bytes2);
  if (e == hipSuccess) d |= bit;
  return e;
}

// The plans' long-row pass keeps the 128-KB shape (16 K distinct values per row in LDS).  GIGL_LG_BIG_CAP=4096 selects a
// 32-KB shape that finds a CU at once — measured (round 5, same box, 2 runs each): 9.47 / 9.50 G edges/s against 9.79 with
// the wide shape: the pass that waits for an emptied CU throttles its own stream and the projection next to it runs at
// 14.8 instead of 21.9 us/step overlapped.  Kept as a knob, not the default.
static bool lg_big_cap_wide() {
  static const bool wide = [] {
    const char* e = getenv("GIGL_LG_BIG_CAP");
    return !(e && atoi(e) > 0 && atoi(e) <= 4096);
  }();
  return wide;
}

namespace {

constexpr int MAXL = GIGL_MAX_HOPS + 1;
constexpr int TILE = 1024;          // stream positions per count/assign workgroup
constexpr int BIG_ROW_CAP = 16384;  // LDS bitonic capacity (64 KiB of int32)
constexpr int LG_BIG_CAP = 4096;    // ... of the long-row pass under GIGL_LG_BIG_CAP=4096 (32 KB of dynamic LDS; A/B knob)

// one open-addressing slot: everything the passes need about a node sits in ONE 16-byte entry, so a probe
// costs one random memory access instead of one per attribute array
struct __attribute__((aligned(16))) Slot {
  unsigned long long kf;  // (node id << 32) | smallest stream position holding the node; ~0 = empty
  int32_t level;          // BFS level in the batch's union graph (starts at hops)
  int32_t lid;            // local id (assign) | LID_MULTI
};
// set in Slot::lid when the node occurs at more than one stream position: only then can a row of it (or an edge out
// of it) repeat, so only then does edge dedup need the hash set — a node that occurs once has f DISTINCT sampled
// in-edges by construction (local ids are < 2^30: T < 2^31 positions)
constexpr int32_t LID_MULTI = 1 << 30;
__device__ __forceinline__ uint32_t slot_key(unsigned long long kf) { return (uint32_t)(kf >> 32); }

struct UnionArgs {
  const uint32_t* roots;
  int32_t b;
  int32_t hops;
  const uint32_t* nbr[GIGL_MAX_HOPS];
  int32_t fan[GIGL_MAX_HOPS];
  int64_t off[GIGL_MAX_HOPS + 1];  // stream offset of hop k slots; off[hops] = T
  int64_t T;
  // node hash table: one sub-table of (mask+1) slots per batch, sub-table g at slots[g * (mask+1)]
  Slot* slots;
  uint32_t mask;
  // per stream position
  int32_t* slot_of;  // table slot (global index) or -1
  // independent batches in one build (gigl_union_build_groups)
  int32_t grouped;
  uint32_t group_roots;
  uint32_t gdiv[GIGL_MAX_HOPS];  // stream slots of ONE batch at hop k
  // leaf-global mode (hops <= 2, the one-call plan): a last-hop slot whose parent is not a root is a pure LEAF — it
  // is never computed, only read as a feature row — so it gets no table slot and no local id (slot_of = LEAF) and
  // stays a GLOBAL id in its parent's row: rows of level hops-1 hold global ids, rows of lower levels local ids
  int32_t leaf_global;
  // hops >= 2: [slots of hop 0] 1 iff that slot's node is a root — written by the hop-0 insert launch, read by the
  // hop-1 launch instead of a random probe of the parent's table slot (scratch shared with the winner flags, which
  // are written later)
  uint8_t* root_parent;
  int32_t* overflow;  // meta[GIGL_META_OVERFLOW]
  // row aliasing (leaf-global, hops == 2, the last hop's nbr array laid out right behind the col buffer): a level-1
  // node that occurs ONCE in the batch has exactly the in-edges its one parent occurrence sampled — ascending,
  // duplicate-free global ids sitting in nbr[1][slot*f1 ..) already — so its row is that tree segment itself
  // (rowptr = alias_base + slot*f1, rowend = rowptr + cnt): no dedup, no fill, no sort for ~90 % of the edges
  int32_t alias_base;       // index of nbr[1][0] in the col array, or -1
  const int32_t* cnt_last;  // tree cnt of the last hop
};
constexpr int32_t LEAF = -2;  // slot_of value of a leaf occurrence in leaf-global mode

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// stream position -> (hop k or -1 for roots, slot index j within the hop)
__device__ __forceinline__ void locate(const UnionArgs& a, int64_t t, int& k, int64_t& j) {
  if (t < a.b) {
    k = -1;
    j = t;
    return;
  }
  int kk = 0;
  int64_t o = a.off[0];
#pragma unroll
  for (int i = 1; i < GIGL_MAX_HOPS; ++i)
    if (i < a.hops && t >= a.off[i]) {
      kk = i;
      o = a.off[i];
    }
  k = kk;
  j = t - o;
}

// arr[k] for a kernel-argument array WITHOUT a dynamic index: indexing the by-value argument struct with a
// run-time k makes the compiler spill the whole struct to scratch memory in every thread (168 B per thread in
// insert_slots: measured 3x on the kernel); a chain of selects over constant indices stays in registers
template <typename T, int N>
__device__ __forceinline__ T pick(const T (&arr)[N], int k) {
  T v = arr[0];
#pragma unroll
  for (int i = 1; i < N; ++i)
    if (k == i) v = arr[i];
  return v;
}

// stream position -> (node id, hop k or -1 for roots, slot index j within the hop)
__device__ __forceinline__ uint32_t stream_at(const UnionArgs& a, int64_t t, int& k, int64_t& j) {
  locate(a, t, k, j);
  return k < 0 ? a.roots[j] : pick(a.nbr, k)[j];
}

// first slot of the sub-table of the batch that stream slot (k, j) belongs to (0 when there is one batch)
__device__ __forceinline__ uint32_t group_base(const UnionArgs& a, int k, int64_t j) {
  if (!a.grouped) return 0u;
  const uint32_t g = (uint32_t)j / (k < 0 ? a.group_roots : pick(a.gdiv, k));
  return g * (a.mask + 1u);
}

// stream position of the destination (parent) of the occurrence at hop k, slot j (positions fit 31 bits)
__device__ __forceinline__ int64_t parent_pos(const UnionArgs& a, int k, int64_t j) {
  const int64_t p = (int64_t)((uint32_t)j / (uint32_t)pick(a.fan, k));
  return k == 0 ? p : pick(a.off, k - 1) + p;
}

// Claim / find the slot of `id` in the sub-table starting at `base` and fold stream position t into the
// node's first position — in ONE 64-bit atomic for a first occurrence: the slot word is (key << 32 |
// first position), so the claiming CAS deposits the position with the key, and a later occurrence of the same
// key lowers the word with a 64-bit atomicMin (equal high halves: the min picks the smaller position).
// Scattered atomics are what bounds the insert kernels; this makes it ~1.1 instead of 2 per occurrence.
// Returns the global slot index.
__device__ __forceinline__ uint32_t table_insert(const UnionArgs& a, uint32_t base, uint32_t id, uint32_t t) {
  const unsigned long long mine = ((unsigned long long)id << 32) | t;
  uint32_t s = hash_u32(id) & a.mask;
  for (uint32_t probes = 0; probes <= a.mask; ++probes) {
    unsigned long long* w = &a.slots[base + s].kf;
    // look before the read-modify-write: a hub node sits in thousands of positions of one batch, and same-address
    // atomics are serialised by the memory system, while plain reads of a hot word are not.  Once the hub is in
    // the table with a smaller position (almost always: blocks run in stream order) its occurrences cost a load.
    // (an ordinary cached load: a stale view is either "empty" -> the CAS decides, or the same key with an older,
    // larger position -> at worst a redundant atomicMin)
    unsigned long long prev = *(const volatile unsigned long long*)w;
    if (prev == ~0ULL) {
      prev = atomicCAS(w, ~0ULL, mine);
      if (prev == ~0ULL) return base + s;
    }
    if ((uint32_t)(prev >> 32) == id) {
      if ((uint32_t)prev > t) atomicMin(w, mine);  // (positions only go down: prev <= t needs no update)
      if (a.slots[base + s].lid != LID_MULTI) a.slots[base + s].lid = LID_MULTI;  // (benign race: same value)
      return base + s;
    }
    s = (s + 1) & a.mask;
  }
  return GIGL_INVALID;  // the sub-table is full (leaf-global mode sizes it for the batches the plan can hold)
}

// all per-batch table initialisation in one dispatch: empty edge keys (0xFF..), empty node slots
// {INVALID key | +inf position, level = hops, -}, zeros, meta
__global__ __launch_bounds__(256) void init_scratch_kernel(uint4* ff, int64_t ff_vec, uint4* slots, int64_t n_slots,
                                                           uint32_t hops, int32_t* zeros, int64_t zero_words,
                                                           int32_t* meta) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4 f4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  const uint4 s4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, hops, 0u);  // {kf lo = firstpos, kf hi = key, level, lid}
  for (int64_t i = t0; i < ff_vec; i += stride) ff[i] = f4;
  for (int64_t i = t0; i < n_slots; i += stride) slots[i] = s4;
  for (int64_t i = t0; i < zero_words; i += stride) zeros[i] = 0;
  if (t0 < GIGL_META_LEN) meta[t0] = 0;
}

__global__ void insert_roots_kernel(UnionArgs a) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.b) return;
  uint32_t id = a.roots[t];
  if (id == GIGL_INVALID) {
    a.slot_of[t] = -1;
    return;
  }
  uint32_t s = table_insert(a, group_base(a, -1, t), id, (uint32_t)t);  // (never full: every table holds its roots)
  a.slots[s].level = 0;
  a.slot_of[t] = (int32_t)s;
}

// stream positions [lo, hi): launched once for the hop-0 slots and once for all later hops, so that a hop-1
// slot finds its parent's table slot in slot_of (written by the first launch) instead of probing for it.
// A slot's level starts at `hops` (the deepest possible), so only shallower occurrences write it.
__global__ void insert_slots_kernel(UnionArgs a, int64_t lo, int64_t hi) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + lo;
  if (t >= hi) return;
  int k;
  int64_t j;
  uint32_t id = stream_at(a, t, k, j);
  if (id == GIGL_INVALID) {
    a.slot_of[t] = -1;
    return;
  }
  int32_t lvl;
  if (k == 0) {
    lvl = 1;
  } else if (k == 1) {
    // parent = a hop-0 slot node, whose level is final since the previous launch: 0 iff it is a root
    lvl = a.root_parent[(uint32_t)j / (uint32_t)a.fan[1]] ? 1 : 2;
  } else {
    lvl = k + 1;  // upper bound; relaxed below
  }
  if (a.leaf_global && lvl == a.hops) {  // (hops <= 2: lvl is exact here) nothing will ever be computed for it
    a.slot_of[t] = LEAF;
    return;
  }
  uint32_t s = table_insert(a, group_base(a, k, j), id, (uint32_t)t);
  if (s == GIGL_INVALID) {  // table full: more inner-level nodes than the plan's workspace holds — batch reported failed
    atomicAdd(a.overflow, 1);
    if (k == 0 && a.hops >= 2) a.root_parent[j] = 0;
    a.slot_of[t] = LEAF;
    return;
  }
  const int32_t cur = *(const volatile int32_t*)&a.slots[s].level;  // roots were inserted by the previous launch
  if (k == 0 && a.hops >= 2) a.root_parent[j] = cur == 0 ? 1 : 0;
  if (lvl < a.hops && cur > lvl)  // (look-before-atomic: levels only go down, a stale read costs a redundant atomic)
    atomicMin(&a.slots[s].level, lvl);
  a.slot_of[t] = (int32_t)s;
}

// hops > 2 only: level(src) = min(level(src), level(dst)+1)
__global__ void relax_kernel(UnionArgs a) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + a.b;
  if (t >= a.T) return;
  int32_t s = a.slot_of[t];
  if (s < 0) return;
  int k;
  int64_t j;
  locate(a, t, k, j);
  int32_t ds = a.slot_of[parent_pos(a, k, j)];
  int32_t dl = __hip_atomic_load(&a.slots[ds].level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  atomicMin(&a.slots[s].level, dl + 1);
}

// first-occurrence level of stream position t, or -1 (one 8-byte read of the slot)
__device__ __forceinline__ int first_level(const UnionArgs& a, int64_t t) {
  if (t >= a.T) return -1;
  int32_t s = a.slot_of[t];
  if (s < 0) return -1;
  const uint4 q = *reinterpret_cast<const uint4*>(&a.slots[s]);  // {firstpos, key, level, lid}
  return q.x == (uint32_t)t ? (int)q.z : -1;
}
// the same, also handing out the slot's lid word (the LID_MULTI flag at this point)
__device__ __forceinline__ int first_level_flags(const UnionArgs& a, int64_t t, int32_t& lidw) {
  lidw = 0;
  if (t >= a.T) return -1;
  int32_t s = a.slot_of[t];
  if (s < 0) return -1;
  const uint4 q = *reinterpret_cast<const uint4*>(&a.slots[s]);
  lidw = (int32_t)q.w;
  return q.x == (uint32_t)t ? (int)q.z : -1;
}

// tile_counts[tile][l] = number of first occurrences of level l in the tile
__global__ __launch_bounds__(256) void count_kernel(UnionArgs a, int32_t* tile_counts) {
  __shared__ int32_t s_cnt[MAXL];
  if (threadIdx.x < MAXL) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
  int32_t c[MAXL];
#pragma unroll
  for (int l = 0; l < MAXL; ++l) c[l] = 0;
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    int lv = first_level(a, base + r * 256 + threadIdx.x);
#pragma unroll
    for (int l = 0; l < MAXL; ++l) c[l] += (lv == l);
  }
#pragma unroll
  for (int l = 0; l < MAXL; ++l) {
    int v = c[l];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt[l], v);
  }
  __syncthreads();
  if (threadIdx.x < MAXL) tile_counts[blockIdx.x * MAXL + threadIdx.x] = s_cnt[threadIdx.x];
}

// tile_counts[tile][l] -> exclusive prefix over the tiles, in place; row n_tiles receives the totals.  One
// workgroup: thread i owns a contiguous run of tiles (every assign workgroup used to re-add all the tiles before
// it: O(tiles^2) loads per build).
__global__ __launch_bounds__(1024) void tile_scan_kernel(int32_t* tile_counts, int32_t n_tiles) {
  __shared__ int32_t s_w[16][MAXL];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per = (n_tiles + 1023) / 1024;
  const int lo = min(tid * per, n_tiles), hi = min(lo + per, n_tiles);
  int32_t sum[MAXL];
#pragma unroll
  for (int l = 0; l < MAXL; ++l) sum[l] = 0;
  for (int i = lo; i < hi; ++i)
#pragma unroll
    for (int l = 0; l < MAXL; ++l) sum[l] += tile_counts[i * MAXL + l];
  int32_t incl[MAXL];
#pragma unroll
  for (int l = 0; l < MAXL; ++l) {
    int32_t v = sum[l];
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t o = __shfl_up(v, off, 64);
      if (lane >= off) v += o;
    }
    incl[l] = v;
    if (lane == 63) s_w[w][l] = v;
  }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < MAXL; ++l) {
    int32_t run = incl[l] - sum[l];
    for (int q = 0; q < w; ++q) run += s_w[q][l];
    for (int i = lo; i < hi; ++i) {
      const int32_t c = tile_counts[i * MAXL + l];
      tile_counts[i * MAXL + l] = run;
      run += c;
    }
    if (tid == 1023) tile_counts[n_tiles * MAXL + l] = run;  // (the last thread's run ends at the total)
  }
}

// local id = base[level] + (first occurrences of that level at smaller stream positions)
__global__ __launch_bounds__(256) void assign_kernel(UnionArgs a, const int32_t* tile_counts, int32_t n_tiles,
                                                     uint32_t* nodes, int32_t* meta, int32_t* rowptr, int32_t* rowend,
                                                     int32_t* rowcnt, int32_t* alias_edges) {
  __shared__ int32_t s_before[MAXL];  // this level's firsts in earlier tiles
  __shared__ int32_t s_total[MAXL];   // totals per level
  __shared__ int32_t s_wave[TILE / 64][MAXL];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < MAXL) {
    s_before[tid] = tile_counts[blockIdx.x * MAXL + tid];
    s_total[tid] = tile_counts[n_tiles * MAXL + tid];
  }
  __syncthreads();
  // ranks inside the tile: sub-tile r = 256 consecutive positions = 4 waves
  const int64_t base = (int64_t)blockIdx.x * TILE;
  int lv[TILE / 256];
  int32_t lidw[TILE / 256];
  int rank_in_wave[TILE / 256];
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    lv[r] = first_level_flags(a, base + r * 256 + tid, lidw[r]);
    rank_in_wave[r] = 0;
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
      unsigned long long m = __ballot(lv[r] == l);
      if (lv[r] == l) rank_in_wave[r] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[r * 4 + w][l] = __popcll(m);
    }
  }
  __syncthreads();
  int32_t alias_c = 0;
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    if (lv[r] < 0) continue;
    const int l = lv[r];
    int32_t id = s_before[l] + rank_in_wave[r];
    for (int q = 0; q < r * 4 + w; ++q) id += s_wave[q][l];
    for (int ll = 0; ll < l; ++ll) id += s_total[ll];
    const int64_t t = base + r * 256 + tid;
    const int32_t s = a.slot_of[t];
    a.slots[s].lid = id | (lidw[r] & LID_MULTI);
    nodes[id] = slot_key(a.slots[s].kf);
    // row aliasing: a node whose ONE occurrence is this hop-0 slot owns the tree segment of its children as its
    // row; rowcnt = -1 tells row_scan to leave the row alone (no winner is ever counted for it)
    if (a.alias_base >= 0 && !(lidw[r] & LID_MULTI) && t >= a.b && t < a.off[1]) {
      const int32_t e = (int32_t)(t - a.b), c = a.cnt_last[e];
      rowptr[id] = a.alias_base + e * a.fan[1];
      rowend[id] = a.alias_base + e * a.fan[1] + c;
      rowcnt[id] = -1;
      alias_c += c;
    }
  }
  if (a.alias_base >= 0) {  // one counter bump per wave, spread over 32 addresses (same-address atomics ~13 ns each)
    for (int off = 32; off > 0; off >>= 1) alias_c += __shfl_xor(alias_c, off, 64);
    if (lane == 0 && alias_c) atomicAdd(&alias_edges[blockIdx.x & 31], alias_c);
  }
  if (blockIdx.x == 0 && tid == 0) {
    int32_t cum = 0;
    for (int l = 0; l < MAXL; ++l) {
      cum += s_total[l];
      if (l <= a.hops) meta[GIGL_META_LEVEL0 + l] = cum;
    }
    meta[GIGL_META_N_NODES] = cum;
  }
}

// Lanes holding equal `key` in a contiguous run form a segment (every lane of the wave must call this).
// total = flagged lanes of my segment, rank = flagged lanes before me in it, first = its first flagged lane
// (or my own lane if it has none).
__device__ __forceinline__ void seg_rank(uint32_t key, bool flag, int lane, int& total, int& rank, int& first) {
  const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x138 /* wave_shr:1 */, 0xF, 0xF,
                                                              false);  // lane 0 keeps ~key: always a leader
  const unsigned long long leaders = __ballot(prev != key);
  const unsigned long long flags = __ballot(flag);
  const unsigned long long upto = (2ull << lane) - 1ull;  // lanes <= me (lane 63: all ones)
  const int start = 63 - __clzll((long long)(leaders & upto));
  const unsigned long long above = leaders & ~upto;
  const unsigned long long below_end = above ? ((1ull << (__ffsll((long long)above) - 1)) - 1ull) : ~0ull;
  const unsigned long long seg = below_end & ~((1ull << start) - 1ull);
  const unsigned long long fs = flags & seg;
  total = __popcll(fs);
  rank = __popcll(fs & ((1ull << lane) - 1ull));
  first = fs ? __ffsll((long long)fs) - 1 : lane;
}

// edge dedup: the first occurrence to claim (dst_local, src_local) in the edge hash set is the edge's
// "winner"; winners are counted per destination row.  Which occurrence wins is irrelevant (rows are
// sorted afterwards).  A winner leaves its (dst_local, src_local) pair for edge_fill.  Threads t < b also
// publish root_local.  The edge set has one sub-table of (emask+1) keys per batch, like the node table.
__global__ void edge_dedup_count_kernel(UnionArgs a, unsigned long long* ekeys, uint32_t emask,
                                        uint8_t* winner, int2* pairs, int32_t* rowcnt, int32_t* root_local) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = t < a.T;
  const int32_t s = in ? a.slot_of[t] : -1;
  if (in && t < a.b) root_local[t] = s >= 0 ? (a.slots[s].lid & ~LID_MULTI) : -1;
  bool win = false;
  int32_t dl = 0, sl = 0;
  if (in && t >= a.b && (s >= 0 || s == LEAF)) {
    int k;
    int64_t j;
    locate(a, t, k, j);
    const int32_t dlf = a.slots[a.slot_of[parent_pos(a, k, j)]].lid;
    dl = dlf & ~LID_MULTI;
    sl = s >= 0 ? (a.slots[s].lid & ~LID_MULTI) : (int32_t)pick(a.nbr, k)[j];  // a leaf keeps its global id
    if (!(dlf & LID_MULTI) && a.alias_base >= 0 && k == a.hops - 1) {
      // the destination's row is the tree segment itself (patched in by edge_fill): nothing to do for this edge
    } else if (!(dlf & LID_MULTI)) {
      win = true;  // the destination occurs once in the batch: its f sampled in-edges are distinct already
    } else {
      const unsigned long long key = ((unsigned long long)(uint32_t)dl << 32) | (uint32_t)sl;
      unsigned long long* sub = ekeys;
      if (a.grouped) sub += (uint64_t)((uint32_t)j / pick(a.gdiv, k)) * (emask + 1u);
      uint32_t h = hash_u32((uint32_t)sl * 0x9E3779B1u ^ (uint32_t)dl) & emask;
      while (true) {
        unsigned long long prev = atomicCAS(&sub[h], ~0ULL, key);
        if (prev == ~0ULL) {
          win = true;
          break;
        }
        if (prev == key) break;
        h = (h + 1) & emask;
      }
    }
    if (win) pairs[t - a.b] = make_int2(dl, sl);
  }
  // the children of one parent sit in adjacent lanes: one atomicAdd per run of equal destinations
  int total, rank, first;
  seg_rank(win ? (uint32_t)dl : (0x80000000u | (uint32_t)lane), win, lane, total, rank, first);
  if (win && rank == 0) atomicAdd(&rowcnt[dl], total);
  if (in && t >= a.b) winner[t - a.b] = win ? 1 : 0;
}

// exclusive scan of rowcnt[0..n) -> rowptr[0..n], rowend[i] = rowptr[i] (fill cursor); rows >= n get
// rowptr = rowend = total.  n = nodes that can have in-edges (levels < hops).
// SCAN_BLOCKS workgroups, each owning one contiguous chunk of the rows: chunk sums are published, ONE grid
// barrier (arrival counter; 16 workgroups are always co-resident, also with 16 streams doing the same), then
// every workgroup scans its chunk starting from the sum of the chunks before it.  `sync` = {partials[16],
// arrival counter}, zeroed per batch by init_scratch.
constexpr int SCAN_BLOCKS = 16;
__global__ __launch_bounds__(1024) void row_scan_kernel(const int32_t* rowcnt, int32_t* meta, int hops,
                                                        int64_t cap_nodes, int32_t* rowptr, int32_t* rowend,
                                                        int tail_here, int32_t* sync) {
  __shared__ int32_t s_w[16];
  __shared__ int32_t s_carry;
  __shared__ int32_t s_total;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t n = meta[GIGL_META_LEVEL0 + hops - 1];
  const int32_t n_nodes = meta[GIGL_META_N_NODES];
  constexpr int PER = 8;  // consecutive rows per thread: 8192 rows per pass of the workgroup
  const int32_t chunk = ((n + SCAN_BLOCKS - 1) / SCAN_BLOCKS + 1024 * PER - 1) / (1024 * PER) * (1024 * PER);
  const int32_t c_lo = min(n, (int32_t)blockIdx.x * chunk), c_hi = min(n, c_lo + chunk);
  {  // chunk sum -> partials[blockIdx], arrive, wait for everybody, carry = sum of the chunks before mine
    int32_t v = 0;
    for (int32_t i = c_lo + tid; i < c_hi; i += 1024) v += max(rowcnt[i], 0);  // (-1 = aliased row: no entries)
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) s_w[w] = v;
    __syncthreads();
    if (tid == 0) {
      int32_t s = 0;
      for (int q = 0; q < 16; ++q) s += s_w[q];
      __hip_atomic_store(&sync[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&sync[SCAN_BLOCKS], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int64_t spins = 0;
      while (__hip_atomic_load(&sync[SCAN_BLOCKS], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < SCAN_BLOCKS) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1ll << 26)) {  // (seconds) never expected: report instead of hanging the device
          atomicAdd(&meta[GIGL_META_OVERFLOW], 1);
          break;
        }
      }
      int32_t before = 0, total = 0;
      for (int q = 0; q < SCAN_BLOCKS; ++q) {
        const int32_t pq = __hip_atomic_load(&sync[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q < (int)blockIdx.x) before += pq;
        total += pq;
      }
      s_carry = before;
      s_total = total;
    }
    __syncthreads();
  }
  for (int32_t base = c_lo; base < c_hi; base += 1024 * PER) {
    const int32_t i0 = base + tid * PER;
    int32_t c[PER];
    bool aliased[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      c[q] = (i0 + q) < c_hi ? rowcnt[i0 + q] : 0;
      aliased[q] = c[q] < 0;  // the row is a tree segment (assign_kernel set its rowptr / rowend): left alone
      c[q] = max(c[q], 0);
    }
    int32_t v = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) v += c[q];
    int32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      int32_t o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int32_t wave_off = 0;
    for (int q = 0; q < w; ++q) wave_off += s_w[q];
    const int32_t carry = s_carry;
    int32_t ex = carry + wave_off + incl - v;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (i0 + q < c_hi && !aliased[q]) {
        rowptr[i0 + q] = ex;
        rowend[i0 + q] = ex;
      }
      ex += c[q];
    }
    __syncthreads();
    if (tid == 1023) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
  if (blockIdx.x != 0) return;
  const int32_t total = s_total;
  if (tail_here)  // no edge_fill launch follows (a batch without slots): close the row arrays here
    for (int64_t i = (int64_t)n + tid; i <= n_nodes && i <= cap_nodes; i += 1024) {
      rowptr[i] = total;
      rowend[i] = total;
    }
  if (tid == 0) meta[GIGL_META_N_EDGES] = total;  // winners only: the unique edge count
}

// scatter the winners into their rows; the same wide launch also closes the row arrays of the nodes
// that have no in-edges (rows n .. n_nodes get rowptr = rowend = n_edges): n_nodes - n <= E always
__global__ void edge_fill_kernel(UnionArgs a, const uint8_t* winner, const int2* pairs, const int32_t* meta,
                                 int32_t* rowptr, int32_t* rowend, int32_t* col) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = e + a.b < a.T;
  if (in) {
    const int32_t n = meta[GIGL_META_LEVEL0 + a.hops - 1], n_nodes = meta[GIGL_META_N_NODES];
    const int32_t total = meta[GIGL_META_N_EDGES];
    const int64_t i = (int64_t)n + e;
    if (i < n_nodes) rowptr[i] = rowend[i] = total;
    if (e == 0) rowptr[n_nodes] = rowend[n_nodes] = total;
  }
  const bool w = in && winner[e];
  int2 p = make_int2(0, 0);
  if (w) p = pairs[e];  // (dst_local, src_local) left by edge_dedup_count
  // one cursor bump per run of equal destinations (the children of one parent are adjacent lanes)
  int total, rank, first;
  seg_rank(w ? (uint32_t)p.x : (0x80000000u | (uint32_t)lane), w, lane, total, rank, first);
  int32_t base = 0;
  if (w && rank == 0) base = atomicAdd(&rowend[p.x], total);
  base = __shfl(base, first, 64);
  if (w) col[base + rank] = p.y;
}

// one wave per row: sort ascending in place (rows <= 64, values are unique); longer rows are queued
__global__ __launch_bounds__(256) void row_sort_kernel(const int32_t* meta, int hops, const int32_t* rowptr,
                                                       const int32_t* rowend, int32_t* col, int32_t* big_rows,
                                                       int32_t* big_count, int32_t alias_base, int32_t* meta_rw) {
  const int lane = threadIdx.x & 63;
  const int32_t n = meta[GIGL_META_LEVEL0 + hops - 1];
  const int32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int32_t waves_total = (gridDim.x * blockDim.x) >> 6;
  if (alias_base >= 0 && blockIdx.x == 0 && threadIdx.x == 0)
  {
    int32_t extra = 0;  // the edges of the aliased rows (counted by edge_fill)
    for (int q = 0; q < 32; ++q) extra += big_count[32 + q];
    meta_rw[GIGL_META_N_EDGES] += extra;
  }
  for (int32_t i = wave; i < n; i += waves_total) {
    const int32_t s = rowptr[i], m = rowend[i] - s;
    if (m <= 1 || (alias_base >= 0 && s >= alias_base)) continue;  // (an aliased row is a sorted tree segment)
    if (m > 64) {
      if (lane == 0) big_rows[atomicAdd(big_count, 1)] = i;
      continue;
    }
    const int32_t v = lane < m ? col[s + lane] : 0x7FFFFFFF;
    int32_t pos = 0;
    for (int j = 0; j < m; ++j) pos += __builtin_amdgcn_readlane(v, j) < v ? 1 : 0;
    if (lane < m) col[s + pos] = v;
  }
}

// one 1024-thread workgroup per queued row (65 .. BIG_ROW_CAP unique values): bucket sort in LDS.
// Values are unique local ids, so 1024 range buckets over [min, max] hold a handful each; a bucket is
// sorted by one wave (rank-by-counting in registers when <= 64, from LDS otherwise).
constexpr int NBUCKET = 1024;

// LDS hand-off between the lanes of ONE wave: drain this wave's LDS traffic, then keep the compiler and the
// lanes from running ahead
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Rows beyond the LDS sort (more than BIG_ROW_CAP entries: a hub that many parents of one batch sampled under different
// path sums).  The reference has no such bound (its union is a Spark array_distinct / the collate's dict), so neither
// does the library: the row is sorted IN PLACE in global memory by the whole workgroup — a bitonic network in its
// "flip" form, every compare-exchange ascending, so positions past the row's end act as +inf and any length works —
// then made distinct chunk by chunk through LDS (a chunk is read completely before its survivors are written at or
// before its start).  ~0.3 ms for a row of 65,536: rare by construction.  Returns the number of distinct values.
__device__ int32_t huge_row_sort_distinct(int32_t* __restrict__ row, int32_t m, int32_t* lds /* >= lds_ints */,
                                          int32_t* s_tmp /* >= 20 ints of LDS */, int32_t lds_ints = 2 * BIG_ROW_CAP) {
  const int tid = threadIdx.x;
  int64_t pow2 = 1;
  while (pow2 < m) pow2 <<= 1;
  const int64_t half = pow2 >> 1;
  auto cmpx = [&](int64_t lo, int64_t hi) {
    if (hi < m) {
      const int32_t a = row[lo], b = row[hi];
      if (a > b) {
        row[lo] = b;
        row[hi] = a;
      }
    }
  };
  for (int64_t k = 2; k <= pow2; k <<= 1) {
    __threadfence_block();
    __syncthreads();
    for (int64_t t = tid; t < half; t += blockDim.x) {  // flip: block [s, s + k): position i against s + k - 1 - (i - s)
      const int64_t blk = t / (k >> 1), off = t % (k >> 1);
      const int64_t lo = blk * k + off;
      cmpx(lo, blk * k + (k - 1 - off));
    }
    for (int64_t j = k >> 2; j > 0; j >>= 1) {
      __threadfence_block();
      __syncthreads();
      for (int64_t t = tid; t < half; t += blockDim.x) {
        const int64_t lo = (t / j) * (j << 1) + (t % j);
        cmpx(lo, lo + j);
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // distinct, in order: chunks of lds_ints values through LDS
  const int32_t CH = lds_ints;
  int32_t out = 0;
  for (int32_t c0 = 0; c0 < m; c0 += CH) {
    const int32_t n = min(CH, m - c0);
    const int32_t before = c0 > 0 ? row[c0 - 1] : 0;  // (read by everybody before anything of this chunk is written)
    for (int q = tid; q < n; q += blockDim.x) lds[q] = row[c0 + q];
    __syncthreads();
    // per-thread run of consecutive values: count heads, prefix over the 1024 threads, write
    const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
    const int q0 = tid * per, q1 = min(n, q0 + per);
    int32_t cnt = 0;
    for (int q = q0; q < q1; ++q) cnt += (q == 0 ? (c0 == 0 || lds[0] != before) : lds[q] != lds[q - 1]) ? 1 : 0;
    int32_t incl = cnt;
    const int lane = tid & 63, w = tid >> 6;
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) s_tmp[w] = incl;
    __syncthreads();
    int32_t base = out, total = 0;
    for (int q = 0; q < 16; ++q) {
      if (q < w) base += s_tmp[q];
      total += s_tmp[q];
    }
    int32_t pos = base + incl - cnt;
    for (int q = q0; q < q1; ++q)
      if (q == 0 ? (c0 == 0 || lds[0] != before) : lds[q] != lds[q - 1]) row[pos++] = lds[q];
    out += total;
    __threadfence_block();
    __syncthreads();
  }
  return out;
}

__global__ __launch_bounds__(1024) void row_sort_big_kernel(const int32_t* rowptr, const int32_t* rowend,
                                                            int32_t* col, const int32_t* big_rows,
                                                            const int32_t* big_count, int32_t* overflow) {
  extern __shared__ int32_t lds[];  // A = lds[0..CAP), B = lds[CAP..2CAP)
  int32_t* A = lds;
  int32_t* B = lds + BIG_ROW_CAP;
  __shared__ int32_t s_cnt[NBUCKET];
  __shared__ int32_t s_off[NBUCKET + 1];
  __shared__ int32_t s_w[16];
  __shared__ int32_t s_sub[16][64];  // per-wave sub-bucket counters
  __shared__ int32_t s_min, s_max;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t nb = *big_count;
  for (int32_t r = blockIdx.x; r < nb; r += gridDim.x) {
    const int32_t i = big_rows[r];
    const int32_t s = rowptr[i], m = rowend[i] - s;
    if (m > BIG_ROW_CAP) {  // does not fit the LDS sort: sorted in place in global memory (values are unique here)
      __syncthreads();
      huge_row_sort_distinct(col + s, m, lds, s_w);
      __syncthreads();
      continue;
    }
    if (tid == 0) {
      s_min = 0x7FFFFFFF;
      s_max = 0;
    }
    s_cnt[tid] = 0;
    __syncthreads();
    int32_t mn = 0x7FFFFFFF, mx = 0;
    for (int q = tid; q < m; q += 1024) {
      const int32_t v = col[s + q];
      A[q] = v;
      mn = min(mn, v);
      mx = max(mx, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = min(mn, __shfl_xor(mn, off, 64));
      mx = max(mx, __shfl_xor(mx, off, 64));
    }
    if (lane == 0) {
      atomicMin(&s_min, mn);
      atomicMax(&s_max, mx);
    }
    __syncthreads();
    const int64_t vmin = s_min, span = (int64_t)s_max - s_min + 1;
    for (int q = tid; q < m; q += 1024) atomicAdd(&s_cnt[(int)(((int64_t)(A[q] - vmin) * NBUCKET) / span)], 1);
    __syncthreads();
    {  // exclusive scan of the 1024 bucket counts (one per thread)
      const int32_t v = s_cnt[tid];
      int32_t incl = v;
      for (int off = 1; off < 64; off <<= 1) {
        int32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      if (lane == 63) s_w[w] = incl;
      __syncthreads();
      int32_t wave_off = 0;
      for (int q = 0; q < w; ++q) wave_off += s_w[q];
      s_off[tid] = wave_off + incl - v;
      if (tid == 1023) s_off[NBUCKET] = wave_off + incl;
      s_cnt[tid] = wave_off + incl - v;  // reuse as the scatter cursor
    }
    __syncthreads();
    for (int q = tid; q < m; q += 1024) {
      const int32_t v = A[q];
      B[atomicAdd(&s_cnt[(int)(((int64_t)(v - vmin) * NBUCKET) / span)], 1)] = v;
    }
    __syncthreads();
    for (int bk = w; bk < NBUCKET; bk += 16) {
      const int32_t lo = s_off[bk], sz = s_off[bk + 1] - lo;
      if (sz == 0) continue;
      if (sz <= 64) {
        const int32_t v = lane < sz ? B[lo + lane] : 0x7FFFFFFF;
        int32_t pos = 0;
        for (int j = 0; j < sz; ++j) pos += __builtin_amdgcn_readlane(v, j) < v ? 1 : 0;
        if (lane < sz) col[s + lo + pos] = v;
      } else {
        // crowded bucket (clustered ids, e.g. a row of a grouped build whose sources sit in one narrow id range
        // per level): a second, wave-local split into 64 range sub-buckets over the bucket's own [min, max],
        // staged through A[lo .. lo+sz) (free after the scatter above); sub-buckets are ranked in registers
        int32_t bmn = 0x7FFFFFFF, bmx = 0;
        for (int q = lane; q < sz; q += 64) {
          const int32_t v = B[lo + q];
          bmn = min(bmn, v);
          bmx = max(bmx, v);
        }
        for (int off = 32; off > 0; off >>= 1) {
          bmn = min(bmn, __shfl_xor(bmn, off, 64));
          bmx = max(bmx, __shfl_xor(bmx, off, 64));
        }
        const int64_t bspan = (int64_t)bmx - bmn + 1;
        s_sub[w][lane] = 0;
        wave_lds_sync();
        for (int q = lane; q < sz; q += 64)
          atomicAdd(&s_sub[w][(int)(((int64_t)(B[lo + q] - bmn) * 64) / bspan)], 1);
        wave_lds_sync();
        const int32_t c = s_sub[w][lane];
        int32_t incl = c;
        for (int off = 1; off < 64; off <<= 1) {
          int32_t o = __shfl_up(incl, off, 64);
          if (lane >= off) incl += o;
        }
        const int32_t sub_lo = incl - c;
        wave_lds_sync();
        s_sub[w][lane] = sub_lo;  // scatter cursor
        wave_lds_sync();
        for (int q = lane; q < sz; q += 64) {
          const int32_t v = B[lo + q];
          A[lo + atomicAdd(&s_sub[w][(int)(((int64_t)(v - bmn) * 64) / bspan)], 1)] = v;
        }
        wave_lds_sync();
        for (int sb = 0; sb < 64; ++sb) {
          const int32_t slo = __builtin_amdgcn_readlane(sub_lo, sb), ssz = __builtin_amdgcn_readlane(c, sb);
          if (ssz == 0) continue;
          if (ssz <= 64) {
            const int32_t v = lane < ssz ? A[lo + slo + lane] : 0x7FFFFFFF;
            int32_t pos = 0;
            for (int j = 0; j < ssz; ++j) pos += __builtin_amdgcn_readlane(v, j) < v ? 1 : 0;
            if (lane < ssz) col[s + lo + slo + pos] = v;
          } else {  // still crowded: rank against the sub-bucket from LDS
            for (int q = lane; q < ssz; q += 64) {
              const int32_t v = A[lo + slo + q];
              int32_t pos = 0;
              for (int j = 0; j < ssz; ++j) pos += A[lo + slo + j] < v ? 1 : 0;
              col[s + lo + slo + pos] = v;
            }
          }
        }
      }
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------------------
// Leaf-global two-hop build ("LG2"): the one-call plan's union graph for hops == 2, in 8 launches.
//
// In leaf-global mode only the INNER stream — the b roots and the b*f0 hop-0 slots — can hold nodes that need a local
// id (plus, rarely, the children of a hop-0 slot whose node is itself a root: "extras"); the b*f0*f1 last-hop slots
// are pure leaves that stay global ids in their parents' rows.  The generic build above still walks all T = b +
// b*f0 + b*f0*f1 stream positions in four of its passes and keeps a 64-bit edge hash set sized for every edge; here
//   * every pass walks the inner stream only (10x fewer positions at [25,10]); the leaves are touched once, by the
//     threads of the hop-0 slots whose node occurs more than once in the batch (a node that occurs once owns its
//     tree segment as its row: row aliasing, as above),
//   * a node's level is read off its first stream position (a root iff that position is < b): roots and hop-0 slots
//     are inserted by ONE launch, the rare extras by a second one,
//   * rows are sized per destination TABLE SLOT while first occurrences are counted (one pass), and get their
//     storage from one atomic cursor bump per workgroup in the assign pass — no row scan, no grid barrier (where a
//     row lands in `col` does not matter: rows are addressed through rowptr / rowend),
//   * duplicate edges are removed where the rows are sorted (registers for rows <= 64 entries, an LDS hash set for
//     longer ones) instead of by a global hash set that had to be cleared for every batch,
//   * the tile prefix of the numbering is computed by the last workgroup of the counting pass to finish.
// Numbering: level 0 = distinct roots by first stream position (== the documented numbering, so root_local is the
// oracle's); level 1 = the other inner nodes by the position of the hop-0 slot (or extra) that holds their first
// occurrence — the plan's own order inside the level, deterministic.  Everything else is as documented for
// gigl_union_build_impl(leaf_global = 1).
struct Lg2Args {
  const uint32_t* roots;
  const uint32_t* nbr0;
  const uint32_t* nbr1;
  const int32_t* cnt1;
  int32_t b, f0, f1;
  int64_t S0;  // b * f0
  int32_t grouped;
  uint32_t group_roots, gdiv0;  // stream slots of ONE batch: roots, hop-0
  int32_t* slot_of;             // [b + S0]
  int32_t alias_base;
  int32_t whole_rows;           // a slot with fewer than f1 children holds its node's WHOLE in-neighbourhood (simple graphs)
};

__device__ __forceinline__ uint32_t lg2_base(const Lg2Args& g, const UnionArgs& a, int64_t t) {
  if (!g.grouped) return 0u;
  const uint32_t grp = t < g.b ? (uint32_t)t / g.group_roots : (uint32_t)(t - g.b) / g.gdiv0;
  return grp * (a.mask + 1u);
}

// find the table slot of a key that IS in the sub-table (extras are looked up, not cached per position)
__device__ __forceinline__ int32_t lg2_find(const UnionArgs& a, uint32_t base, uint32_t id) {
  uint32_t s = hash_u32(id) & a.mask;
  for (uint32_t probes = 0; probes <= a.mask; ++probes) {
    const unsigned long long kf = a.slots[base + s].kf;
    if ((uint32_t)(kf >> 32) == id) return (int32_t)(base + s);
    if (kf == ~0ULL) return -1;
    s = (s + 1) & a.mask;
  }
  return -1;
}

// does the hop-0 occurrence at stream position t (node in table slot s, c sampled children) add its children to the
// node's row?  An occurrence with c < f1 children sampled the node's WHOLE in-neighbourhood, and so did every other
// occurrence of that node: the list is taken once, from the node's first occurrence (for a node that is also a root
// the first occurrence is a root position, which has no hop-1 children of its own: every occurrence contributes).
__device__ __forceinline__ bool lg2_contributes(const UnionArgs& a, const Lg2Args& g, int32_t s, int64_t t, int c) {
  if (c >= g.f1 || !g.whole_rows) return true;
  const uint32_t fp = (uint32_t)a.slots[s].kf;
  return fp == (uint32_t)t || fp < (uint32_t)g.b;
}

__global__ __launch_bounds__(256) void lg2_init_kernel(uint4* slots, int64_t n_slots, int32_t* zeros, int64_t zero_words,
                                                       int32_t* meta) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4 s4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);  // {kf lo = firstpos, kf hi = key, rowcnt, lid}
  for (int64_t i = t0; i < n_slots; i += stride) slots[i] = s4;
  for (int64_t i = t0; i < zero_words; i += stride) zeros[i] = 0;
  if (t0 < GIGL_META_LEN) meta[t0] = 0;
}

__global__ __launch_bounds__(256) void lg2_insert_kernel(UnionArgs a, Lg2Args g) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= g.b + g.S0) return;
  const uint32_t id = t < g.b ? g.roots[t] : g.nbr0[t - g.b];
  if (id == GIGL_INVALID) {
    g.slot_of[t] = -1;
    return;
  }
  const uint32_t s = table_insert(a, lg2_base(g, a, t), id, (uint32_t)t);
  if (s == GIGL_INVALID) {  // sub-table full: more inner nodes than the plan's workspace holds — batch reported failed
    atomicAdd(a.overflow, 1);
    g.slot_of[t] = -1;
    return;
  }
  g.slot_of[t] = (int32_t)s;
}

// hop-0 slots whose node is a root: the children of that occurrence are in-neighbours of a root, i.e. inner nodes
__global__ __launch_bounds__(256) void lg2_extras_kernel(UnionArgs a, Lg2Args g) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.S0) return;
  const int32_t s = g.slot_of[g.b + p];
  if (s < 0) return;
  if ((uint32_t)a.slots[s].kf >= (uint32_t)g.b) return;  // first position is not a root position
  const int c = g.cnt1[p];
  const uint32_t base = lg2_base(g, a, g.b + p);
  for (int j = 0; j < c; ++j) {
    const uint32_t id = g.nbr1[p * g.f1 + j];
    if (table_insert(a, base, id, (uint32_t)(g.b + g.S0 + p * g.f1 + j)) == GIGL_INVALID) atomicAdd(a.overflow, 1);
  }
}

// first occurrences held by inner stream position t: c0 = level-0 firsts (0/1), c1 = level-1 firsts (the position's
// own and those of its extras).  xmask: bit j set iff extra child j is a first occurrence.
__device__ __forceinline__ void lg2_firsts(const UnionArgs& a, const Lg2Args& g, int64_t t, int32_t& s_out, int& c0,
                                           int& c1, uint32_t& own_first, unsigned long long& xmask, bool& root_parent) {
  c0 = c1 = 0;
  own_first = 0;
  xmask = 0;
  root_parent = false;
  s_out = -1;
  if (t >= g.b + g.S0) return;
  const int32_t s = g.slot_of[t];
  s_out = s;
  if (s < 0) return;
  const uint32_t fp = (uint32_t)a.slots[s].kf;
  if (fp == (uint32_t)t) {
    own_first = 1;
    if (t < g.b) c0 = 1; else c1 = 1;
  }
  if (t >= g.b && fp < (uint32_t)g.b) {
    root_parent = true;
    const int64_t p = t - g.b;
    const int c = g.cnt1[p];
    const uint32_t base = lg2_base(g, a, t);
    for (int j = 0; j < c; ++j) {
      const int32_t sx = lg2_find(a, base, g.nbr1[p * g.f1 + j]);
      if (sx >= 0 && (uint32_t)a.slots[sx].kf == (uint32_t)(g.b + g.S0 + p * g.f1 + j)) {
        xmask |= 1ull << j;
        ++c1;
      }
    }
  }
}

// per 1024-position tile of the inner stream: first-occurrence counts per level; in-edge occurrence counts per
// destination table slot (Slot::level is the counter here); the LAST workgroup to finish turns the tile counts into
// exclusive prefixes (row n_tiles = totals).
__global__ __launch_bounds__(256) void lg2_count_kernel(UnionArgs a, Lg2Args g, int32_t* tile_counts, int32_t n_tiles,
                                                        int32_t* ticket) {
  __shared__ int32_t s_c[2];
  __shared__ int32_t s_last;
  __shared__ int32_t s_w[4][2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 2) s_c[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
  int c0 = 0, c1 = 0;
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    const int64_t t = base + r * 256 + tid;
    int32_t s;
    int k0, k1;
    uint32_t own;
    unsigned long long xm;
    bool rp;
    lg2_firsts(a, g, t, s, k0, k1, own, xm, rp);
    c0 += k0;
    c1 += k1;
    // row sizes: the hop-0 edge (node of slot p -> root p / f0), one bump per run of children of the same root;
    // the children of a node that occurs more than once go to that node's row (a node that occurs once keeps its
    // tree segment as its row)
    const bool edge = s >= 0 && t >= g.b && t < g.b + g.S0;
    int32_t sr = -1;
    if (edge) sr = g.slot_of[(uint32_t)(t - g.b) / (uint32_t)g.f0];
    const bool e_ok = edge && sr >= 0;
    int total, rank, first;
    seg_rank(e_ok ? (uint32_t)sr : (0x80000000u | (uint32_t)lane), e_ok, lane, total, rank, first);
    if (e_ok && rank == 0) atomicAdd(&a.slots[sr].level, total);
    if (edge && ((a.slots[s].lid & LID_MULTI) || g.alias_base < 0)) {
      const int c = g.cnt1[t - g.b];
      if (c > 0 && lg2_contributes(a, g, s, t, c)) atomicAdd(&a.slots[s].level, c);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    c0 += __shfl_xor(c0, off, 64);
    c1 += __shfl_xor(c1, off, 64);
  }
  if (lane == 0) {
    if (c0) atomicAdd(&s_c[0], c0);
    if (c1) atomicAdd(&s_c[1], c1);
  }
  __syncthreads();
  if (tid == 0) {
    tile_counts[blockIdx.x * 2 + 0] = s_c[0];
    tile_counts[blockIdx.x * 2 + 1] = s_c[1];
    __threadfence();  // the counts are visible device-wide before the ticket is taken
    s_last = atomicAdd(ticket, 1) == (int32_t)gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 holds nothing stale of the other workgroups' counts
  // exclusive prefix over the tiles, 256 tiles per round, both levels
  int32_t run0 = 0, run1 = 0;
  for (int32_t t0 = 0; t0 < n_tiles; t0 += 256) {
    const int32_t i = t0 + tid;
    const int32_t v0 = i < n_tiles ? __hip_atomic_load(&tile_counts[i * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int32_t v1 = i < n_tiles ? __hip_atomic_load(&tile_counts[i * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int32_t i0 = v0, i1 = v1;
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t o0 = __shfl_up(i0, off, 64), o1 = __shfl_up(i1, off, 64);
      if (lane >= off) {
        i0 += o0;
        i1 += o1;
      }
    }
    if (lane == 63) {
      s_w[w][0] = i0;
      s_w[w][1] = i1;
    }
    __syncthreads();
    int32_t w0 = 0, w1 = 0, tot0 = 0, tot1 = 0;
    for (int q = 0; q < 4; ++q) {
      if (q < w) {
        w0 += s_w[q][0];
        w1 += s_w[q][1];
      }
      tot0 += s_w[q][0];
      tot1 += s_w[q][1];
    }
    if (i < n_tiles) {
      tile_counts[i * 2 + 0] = run0 + w0 + i0 - v0;
      tile_counts[i * 2 + 1] = run1 + w1 + i1 - v1;
    }
    run0 += tot0;
    run1 += tot1;
    __syncthreads();
  }
  if (tid == 0) {
    tile_counts[n_tiles * 2 + 0] = run0;
    tile_counts[n_tiles * 2 + 1] = run1;
  }
}

// rows of 2..TINY_ROW entries (a node with two or three occurrences): one THREAD per row — a sorting network over the
// row held in registers, duplicates removed on the way out.  A wave per such row spends its time waiting on four
// dependent loads; a thread per row keeps 64 rows in flight per wave.
constexpr int TINY_ROW = 32;

__global__ __launch_bounds__(256) void lg2_row_sort_tiny_kernel(const int32_t* tiny_rows, const int32_t* tiny_count,
                                                                const int32_t* rowptr, int32_t* rowend, int32_t* col,
                                                                int32_t* edge_counters, int ec_stride = 1) {
  // (round 4) the row lives in 32 REGISTERS: padded with +inf, sorted by a fully unrolled bitonic network (240
  // compare-exchanges, no memory access), duplicates dropped on the way out.  The insertion sort in an LDS strip this
  // replaces was a chain of dependent LDS reads — ~40 us of latency per launch whatever the number of rows.
  const int32_t nq = *tiny_count;
  int32_t edges = 0;
  for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
    const int32_t i = tiny_rows[q];
    const int32_t s = rowptr[i], m = rowend[i] - s;
    int32_t v[TINY_ROW];
#pragma unroll
    for (int j = 0; j < TINY_ROW; ++j) v[j] = j < m ? col[s + j] : 0x7FFFFFFF;  // (independent loads: all in flight together)
#pragma unroll
    for (int k = 2; k <= TINY_ROW; k <<= 1)
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
        for (int x = 0; x < TINY_ROW; ++x) {
          const int y = x ^ j;
          if (y > x) {
            const bool up = (x & k) == 0;
            const int32_t lo = min(v[x], v[y]), hi = max(v[x], v[y]);
            v[x] = up ? lo : hi;
            v[y] = up ? hi : lo;
          }
        }
    int32_t u = 0, prev = -1;  // (entries are local or global node ids: non-negative)
#pragma unroll
    for (int j = 0; j < TINY_ROW; ++j) {
      if (j < m && v[j] != prev) {
        col[s + u] = v[j];
        ++u;
      }
      prev = v[j];
    }
    rowend[i] = s + u;
    edges += u;
  }
  for (int off = 32; off > 0; off >>= 1) edges += __shfl_xor(edges, off, 64);
  if ((threadIdx.x & 63) == 0 && edges) atomicAdd(&edge_counters[(blockIdx.x & 31) * ec_stride], edges);
}

// local ids, node list, row storage.  A thread numbers the first occurrences it holds (its own position, then its
// extras in child order); threads in position order.  Row storage: one cursor bump per workgroup.
__global__ __launch_bounds__(256) void lg2_assign_kernel(UnionArgs a, Lg2Args g, const int32_t* tile_counts,
                                                         int32_t n_tiles, uint32_t* nodes, int32_t* meta,
                                                         int32_t* rowptr, int32_t* rowend, int32_t* cursor,
                                                         int32_t* alias_edges, int32_t* sort_rows,
                                                         int32_t* sort_count, int32_t* tiny_rows,
                                                         int32_t* tiny_count) {
  // per (sub-row, wave): level-0 firsts, level-1 firsts, row entries needed, rows queued for sorting (long | tiny)
  __shared__ int32_t s_w[TILE / 64][5];
  __shared__ int32_t s_row_base, s_q_base, s_t_base;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t before0 = tile_counts[blockIdx.x * 2 + 0], before1 = tile_counts[blockIdx.x * 2 + 1];
  const int32_t total0 = tile_counts[n_tiles * 2 + 0], total1 = tile_counts[n_tiles * 2 + 1];
  const int64_t base = (int64_t)blockIdx.x * TILE;
  int32_t sl[TILE / 256];
  int k0[TILE / 256], k1[TILE / 256];
  uint32_t own[TILE / 256];
  unsigned long long xm[TILE / 256];
  bool rp[TILE / 256];
  int32_t own_len[TILE / 256];  // entries of the own first's row, -1: the row is a tree segment (aliased)
  int32_t x0[TILE / 256], x1[TILE / 256], xn[TILE / 256], xq[TILE / 256], xt[TILE / 256];  // exclusive wave prefixes
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    const int64_t t = base + r * 256 + tid;
    lg2_firsts(a, g, t, sl[r], k0[r], k1[r], own[r], xm[r], rp[r]);
    int32_t nd = 0, nq = 0, nt = 0;
    own_len[r] = 0;
    if (own[r]) {
      const int32_t lidw = a.slots[sl[r]].lid;
      const bool alias = g.alias_base >= 0 && t >= g.b && !(lidw & LID_MULTI);
      own_len[r] = alias ? -1 : a.slots[sl[r]].level;
      if (!alias) {
        nd += own_len[r];
        // needs dedup + sort — unless it is the row of a root that occurs once: its entries are the (distinct) nodes
        // of its hop-0 slots, placed by slot number in the fill pass
        const bool sort_it = own_len[r] >= 2 && (lidw & LID_MULTI);
        nq = sort_it && own_len[r] > TINY_ROW ? 1 : 0;
        nt = sort_it && own_len[r] <= TINY_ROW ? 1 : 0;
      }
    }
    // (an extra can be a destination only through a hop-0 occurrence of its own, which then holds its first
    // occurrence: a node first seen as an extra has no in-edges; its counted size is 0)
    if (xm[r]) {
      const int64_t p = t - g.b;
      const uint32_t tb = lg2_base(g, a, t);
      for (int j = 0; j < g.f1; ++j)
        if (xm[r] >> j & 1ull) nd += a.slots[lg2_find(a, tb, g.nbr1[p * g.f1 + j])].level;
    }
    int32_t i0 = k0[r], i1 = k1[r], in = nd, iq = nq, it = nt;
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t o0 = __shfl_up(i0, off, 64), o1 = __shfl_up(i1, off, 64), on = __shfl_up(in, off, 64),
                    oq = __shfl_up(iq, off, 64), ot = __shfl_up(it, off, 64);
      if (lane >= off) {
        i0 += o0;
        i1 += o1;
        in += on;
        iq += oq;
        it += ot;
      }
    }
    x0[r] = i0 - k0[r];
    x1[r] = i1 - k1[r];
    xn[r] = in - nd;
    xq[r] = iq - nq;
    xt[r] = it - nt;
    if (lane == 63) {
      s_w[r * 4 + w][0] = i0;
      s_w[r * 4 + w][1] = i1;
      s_w[r * 4 + w][2] = in;
      s_w[r * 4 + w][3] = iq;
      s_w[r * 4 + w][4] = it;
    }
  }
  __syncthreads();
  if (tid == 0) {  // row storage and queue positions of the whole tile: one bump each
    int32_t tot = 0, totq = 0, tott = 0;
    for (int q = 0; q < TILE / 64; ++q) {
      tot += s_w[q][2];
      totq += s_w[q][3];
      tott += s_w[q][4];
    }
    s_row_base = tot ? atomicAdd(cursor, tot) : 0;
    s_q_base = totq ? atomicAdd(sort_count, totq) : 0;
    s_t_base = tott ? atomicAdd(tiny_count, tott) : 0;
  }
  __syncthreads();
  int32_t alias_c = 0;
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    if (!(k0[r] | k1[r])) continue;
    int32_t p0 = before0 + x0[r], p1 = total0 + before1 + x1[r], pn = s_row_base + xn[r], pq = s_q_base + xq[r],
            pt = s_t_base + xt[r];
    for (int q = 0; q < r * 4 + w; ++q) {
      p0 += s_w[q][0];
      p1 += s_w[q][1];
      pn += s_w[q][2];
      pq += s_w[q][3];
      pt += s_w[q][4];
    }
    const int64_t t = base + r * 256 + tid;
    if (own[r]) {
      const int32_t s = sl[r];
      const int32_t lidw = a.slots[s].lid;
      const int32_t id = k0[r] ? p0 : p1++;
      a.slots[s].lid = id | (lidw & LID_MULTI);
      nodes[id] = slot_key(a.slots[s].kf);
      if (own_len[r] < 0) {  // the row IS the tree segment of its children
        const int64_t p = t - g.b;
        const int32_t c = g.cnt1[p];
        rowptr[id] = g.alias_base + (int32_t)(p * g.f1);
        rowend[id] = g.alias_base + (int32_t)(p * g.f1) + c;
        alias_c += c;
      } else {
        const bool sorted_later = own_len[r] >= 2 && (lidw & LID_MULTI);
        rowptr[id] = pn;
        rowend[id] = sorted_later || own_len[r] < 2 ? pn : pn + own_len[r];  // fill cursor | final end
        pn += own_len[r];
        if (sorted_later && own_len[r] > TINY_ROW) sort_rows[pq] = id;
        else if (sorted_later) tiny_rows[pt] = id;
        else alias_c += own_len[r];  // (the row is final: counted with the aliased edges)
      }
    }
    if (xm[r]) {
      const int64_t p = t - g.b;
      const uint32_t tb = lg2_base(g, a, t);
      for (int j = 0; j < g.f1; ++j) {
        if (!(xm[r] >> j & 1ull)) continue;
        const int32_t s = lg2_find(a, tb, g.nbr1[p * g.f1 + j]);
        const int32_t id = p1++;
        a.slots[s].lid = id | (a.slots[s].lid & LID_MULTI);
        nodes[id] = slot_key(a.slots[s].kf);
        rowptr[id] = pn;
        rowend[id] = pn;
        pn += a.slots[s].level;
      }
    }
  }
  {  // one counter bump per wave, spread over 32 addresses
    for (int off = 32; off > 0; off >>= 1) alias_c += __shfl_xor(alias_c, off, 64);
    if (lane == 0 && alias_c) atomicAdd(&alias_edges[blockIdx.x & 31], alias_c);
  }
  if (blockIdx.x == 0 && tid == 0) {
    meta[GIGL_META_LEVEL0] = total0;
    meta[GIGL_META_LEVEL0 + 1] = total0 + total1;
    meta[GIGL_META_LEVEL0 + 2] = total0 + total1;
    meta[GIGL_META_N_NODES] = total0 + total1;
    rowptr[total0 + total1] = 0;  // (rows past the inner nodes do not exist in leaf-global mode)
    rowend[total0 + total1] = 0;
  }
}

// write the rows: hop-0 edges into the roots' rows, the children of multiply-occurring nodes into those nodes' rows;
// root positions publish root_local
__global__ __launch_bounds__(256) void lg2_fill_kernel(UnionArgs a, Lg2Args g, int32_t* rowend, int32_t* col,
                                                       int32_t* root_local, const int32_t* rowptr) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = t < g.b + g.S0;
  const int32_t s = in ? g.slot_of[t] : -1;
  if (in && t < g.b) root_local[t] = s >= 0 ? (a.slots[s].lid & ~LID_MULTI) : -1;
  const bool edge = in && t >= g.b && s >= 0;
  int32_t sr = -1, lidw = 0;
  if (edge) {
    sr = g.slot_of[(uint32_t)(t - g.b) / (uint32_t)g.f0];
    lidw = a.slots[s].lid;
  }
  bool e_ok = edge && sr >= 0;
  int32_t dlw = e_ok ? a.slots[sr].lid : 0;
  const int32_t dl = dlw & ~LID_MULTI;
  if (e_ok && !(dlw & LID_MULTI) && a.slots[sr].level >= 2) {
    // the root occurs once: its row is its hop-0 slots' nodes in slot order (valid slots are a prefix) — no cursor
    col[rowptr[dl] + (int32_t)((uint32_t)(t - g.b) % (uint32_t)g.f0)] = lidw & ~LID_MULTI;
    e_ok = false;
  }
  int total, rank, first;
  seg_rank(e_ok ? (uint32_t)dl : (0x80000000u | (uint32_t)lane), e_ok, lane, total, rank, first);
  int32_t basep = 0;
  if (e_ok && rank == 0) basep = atomicAdd(&rowend[dl], total);
  basep = __shfl(basep, first, 64);
  if (e_ok) col[basep + rank] = lidw & ~LID_MULTI;
  if (edge && ((lidw & LID_MULTI) || g.alias_base < 0)) {
    const int64_t p = t - g.b;
    const int c = g.cnt1[p];
    if (c > 0 && lg2_contributes(a, g, s, t, c)) {
      const int32_t da = lidw & ~LID_MULTI;
      const int32_t at = atomicAdd(&rowend[da], c);
      const bool local = (uint32_t)a.slots[s].kf < (uint32_t)g.b;  // the node is a root: its row holds local ids
      const uint32_t tb = lg2_base(g, a, t);
      for (int j = 0; j < c; ++j) {
        const uint32_t id = g.nbr1[p * g.f1 + j];
        int32_t v = (int32_t)id;
        if (local) {
          const int32_t sx = lg2_find(a, tb, id);
          v = sx >= 0 ? (a.slots[sx].lid & ~LID_MULTI) : 0;
        }
        col[at + j] = v;
      }
    }
  }
}

// one wave per QUEUED row (rows of >= 2 entries that are not tree segments): <= 64 entries are deduplicated and
// sorted in registers; <= MED_ROW entries through a per-wave LDS hash set + rank by counting; longer ones go to the
// workgroup-per-row kernel.  Edges are counted after dedup (spread counters).
constexpr int MED_ROW = 512;
constexpr int MED_HASH = 1024;

__global__ __launch_bounds__(256) void lg2_row_sort_kernel(const int32_t* sort_rows, const int32_t* sort_count,
                                                           const int32_t* rowptr, int32_t* rowend, int32_t* col,
                                                           int32_t* big_rows, int32_t* big_count,
                                                           int32_t* edge_counters, int ec_stride = 1) {
  __shared__ int32_t s_h[4][MED_HASH];
  __shared__ int32_t s_u[4][MED_ROW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int32_t nq = *sort_count;
  const int32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int32_t waves_total = (gridDim.x * blockDim.x) >> 6;
  int32_t edges = 0;
  for (int32_t q = wave; q < nq; q += waves_total) {
    const int32_t i = sort_rows[q];
    const int32_t s = rowptr[i], m = rowend[i] - s;
    if (m > MED_ROW) {
      if (lane == 0) big_rows[atomicAdd(big_count, 1)] = i;
      continue;
    }
    if (m <= 64) {
      const int32_t v = lane < m ? col[s + lane] : 0x7FFFFFFF;
      bool firstv = lane < m;  // no earlier lane holds the same value
      for (int j = 0; j < m; ++j) {
        const int32_t vj = __builtin_amdgcn_readlane(v, j);
        if (j < lane && vj == v) firstv = false;
      }
      const unsigned long long fm = __ballot(firstv);
      int32_t pos = 0;
      for (int j = 0; j < m; ++j) {
        const int32_t vj = __builtin_amdgcn_readlane(v, j);
        pos += ((fm >> j) & 1ull) && vj < v ? 1 : 0;
      }
      if (firstv) col[s + pos] = v;
      const int32_t len = (int32_t)__popcll(fm);
      if (lane == 0) rowend[i] = s + len;
      edges += len;
      continue;
    }
    // ---- 65 .. MED_ROW entries: distinct values through this wave's LDS hash set
    int32_t* H = s_h[w];
    int32_t* U = s_u[w];
#pragma unroll
    for (int r = 0; r < MED_HASH / 64; ++r) H[lane + 64 * r] = -1;
    wave_lds_sync();
    for (int r = 0; r * 64 < m; ++r) {
      const int idx = lane + 64 * r;
      if (idx < m) {
        const int32_t v = col[s + idx];
        uint32_t h = hash_u32((uint32_t)v) & (MED_HASH - 1);
        for (;;) {
          const int32_t prev = atomicCAS(&H[h], -1, v);
          if (prev == -1 || prev == v) break;
          h = (h + 1) & (MED_HASH - 1);
        }
      }
    }
    wave_lds_sync();
    int32_t u = 0;
#pragma unroll
    for (int r = 0; r < MED_HASH / 64; ++r) {
      const int32_t v = H[lane + 64 * r];
      const unsigned long long vm = __ballot(v != -1);
      if (v != -1) U[u + (int32_t)__popcll(vm & ((1ull << lane) - 1ull))] = v;
      u += (int32_t)__popcll(vm);
    }
    wave_lds_sync();
    // rank by counting: every lane owns the values U[lane + 64 r]; U[j] is read once (a broadcast) for all of them
    int32_t mine[MED_ROW / 64], pos[MED_ROW / 64];
#pragma unroll
    for (int r = 0; r < MED_ROW / 64; ++r) {
      mine[r] = lane + 64 * r < u ? U[lane + 64 * r] : 0x7FFFFFFF;
      pos[r] = 0;
    }
    for (int j = 0; j < u; ++j) {
      const int32_t x = U[j];
#pragma unroll
      for (int r = 0; r < MED_ROW / 64; ++r) pos[r] += x < mine[r] ? 1 : 0;
    }
#pragma unroll
    for (int r = 0; r < MED_ROW / 64; ++r)
      if (lane + 64 * r < u) col[s + pos[r]] = mine[r];
    if (lane == 0) rowend[i] = s + u;
    edges += u;
    wave_lds_sync();
  }
  if (lane == 0 && edges) atomicAdd(&edge_counters[(blockIdx.x & 31) * ec_stride], edges);
}

// queued rows: duplicates removed through an LDS hash set (any number of entries, <= BIG_ROW_CAP distinct), the
// distinct values written back to the head of the row, then the bucket sort of row_sort_big_kernel.  The last
// workgroup to finish publishes meta[N_EDGES].
// CAP: distinct values the LDS set / sort hold (2 * CAP ints of dynamic LDS).  The plans launch CAP = 4096 (32 KB: finds a CU
// next to other streams' kernels; a row of more distinct values — a hub met under > 400 parents of one batch — takes the
// global-memory path), round 5: the 128-KB shape waited ~200 us per call for a CU with that much LDS free.
template <int CAP>
__global__ __launch_bounds__(1024) void lg2_row_sort_big_kernel(const int32_t* rowptr, int32_t* rowend, int32_t* col,
                                                                const int32_t* big_rows, const int32_t* big_count,
                                                                int32_t* overflow, int32_t* edge_counters,
                                                                int32_t* ticket, int32_t* meta, int ec_stride = 1,
                                                                const int32_t* tile_edges = nullptr, int32_t n_tile_edges = 0) {
  // (ec_stride: the 32 spread edge counters sit ec_stride words apart — LG3 keeps each on its own 128-byte line;
  // tile_edges: per-tile edge counts written by plain stores, added to the total here)
  extern __shared__ int32_t lds[];  // hash set of 2*CAP keys, then A = lds[0..CAP), B = lds[CAP..2CAP)
  int32_t* A = lds;
  int32_t* B = lds + CAP;
  __shared__ int32_t s_cnt[NBUCKET];
  __shared__ int32_t s_off[NBUCKET + 1];
  __shared__ int32_t s_w[16];
  __shared__ int32_t s_sub[16][64];  // per-wave sub-bucket counters
  __shared__ int32_t s_min, s_max, s_uniq, s_last;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t nb = *big_count;
  constexpr uint32_t HMASK = 2 * CAP - 1;
  int32_t edges = 0;
  for (int32_t r = blockIdx.x; r < nb; r += gridDim.x) {
    const int32_t i = big_rows[r];
    const int32_t s = rowptr[i], m_raw = rowend[i] - s;
    // ---- distinct values
    for (int q = tid; q < 2 * CAP; q += 1024) lds[q] = -1;
    if (tid == 0) s_uniq = 0;
    __syncthreads();
    for (int q = tid; q < m_raw; q += 1024) {
      // (more distinct values than the sort holds: the row takes the global-memory path below — stop filling the set,
      // whose probe chains grow without bound as it fills)
      if (*(volatile int32_t*)&s_uniq > CAP) break;
      const int32_t v = col[s + q];
      uint32_t h = hash_u32((uint32_t)v) & HMASK;
      for (uint32_t probes = 0; probes <= HMASK; ++probes) {
        const int32_t prev = atomicCAS(&lds[h], -1, v);
        if (prev == -1) {
          atomicAdd(&s_uniq, 1);
          break;
        }
        if (prev == v) break;
        h = (h + 1) & HMASK;
      }
    }
    __syncthreads();
    const int32_t m = s_uniq;
    __syncthreads();  // (everybody has read the count before it is reused as the compaction cursor)
    if (m > CAP) {  // more distinct values than the LDS set / sort hold: sorted + made distinct in global memory
      const int32_t md = huge_row_sort_distinct(col + s, m_raw, lds, s_w, 2 * CAP);
      if (tid == 0) {
        rowend[i] = s + md;
        edges += md;
      }
      __syncthreads();
      continue;
    }
    // compact the set into the head of the row (order irrelevant: sorted next)
    if (tid == 0) s_uniq = 0;
    __syncthreads();
    for (int q = tid; q < 2 * CAP; q += 1024) {
      const int32_t v = lds[q];
      if (v != -1) col[s + atomicAdd(&s_uniq, 1)] = v;
    }
    __syncthreads();
    if (tid == 0) {
      rowend[i] = s + m;
      edges += m;
      s_min = 0x7FFFFFFF;
      s_max = 0;
    }
    s_cnt[tid] = 0;
    __syncthreads();
    if (m <= 1) continue;
    // ---- bucket sort of the m distinct values (as row_sort_big_kernel)
    int32_t mn = 0x7FFFFFFF, mx = 0;
    for (int q = tid; q < m; q += 1024) {
      const int32_t v = col[s + q];
      A[q] = v;
      mn = min(mn, v);
      mx = max(mx, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = min(mn, __shfl_xor(mn, off, 64));
      mx = max(mx, __shfl_xor(mx, off, 64));
    }
    if (lane == 0) {
      atomicMin(&s_min, mn);
      atomicMax(&s_max, mx);
    }
    __syncthreads();
    const int64_t vmin = s_min, span = (int64_t)s_max - s_min + 1;
    for (int q = tid; q < m; q += 1024) atomicAdd(&s_cnt[(int)(((int64_t)(A[q] - vmin) * NBUCKET) / span)], 1);
    __syncthreads();
    {
      const int32_t v = s_cnt[tid];
      int32_t incl = v;
      for (int off = 1; off < 64; off <<= 1) {
        int32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      if (lane == 63) s_w[w] = incl;
      __syncthreads();
      int32_t wave_off = 0;
      for (int q = 0; q < w; ++q) wave_off += s_w[q];
      s_off[tid] = wave_off + incl - v;
      if (tid == 1023) s_off[NBUCKET] = wave_off + incl;
      s_cnt[tid] = wave_off + incl - v;
    }
    __syncthreads();
    for (int q = tid; q < m; q += 1024) {
      const int32_t v = A[q];
      B[atomicAdd(&s_cnt[(int)(((int64_t)(v - vmin) * NBUCKET) / span)], 1)] = v;
    }
    __syncthreads();
    for (int bk = w; bk < NBUCKET; bk += 16) {
      const int32_t lo = s_off[bk], sz = s_off[bk + 1] - lo;
      if (sz == 0) continue;
      if (sz <= 64) {
        const int32_t v = lane < sz ? B[lo + lane] : 0x7FFFFFFF;
        int32_t pos = 0;
        for (int j = 0; j < sz; ++j) pos += __builtin_amdgcn_readlane(v, j) < v ? 1 : 0;
        if (lane < sz) col[s + lo + pos] = v;
      } else {
        // crowded bucket (clustered ids, e.g. a row of a grouped build whose sources sit in one narrow id range
        // per level): a second, wave-local split into 64 range sub-buckets over the bucket's own [min, max],
        // staged through A[lo .. lo+sz) (free after the scatter above); sub-buckets are ranked in registers
        int32_t bmn = 0x7FFFFFFF, bmx = 0;
        for (int q = lane; q < sz; q += 64) {
          const int32_t v = B[lo + q];
          bmn = min(bmn, v);
          bmx = max(bmx, v);
        }
        for (int off = 32; off > 0; off >>= 1) {
          bmn = min(bmn, __shfl_xor(bmn, off, 64));
          bmx = max(bmx, __shfl_xor(bmx, off, 64));
        }
        const int64_t bspan = (int64_t)bmx - bmn + 1;
        s_sub[w][lane] = 0;
        wave_lds_sync();
        for (int q = lane; q < sz; q += 64)
          atomicAdd(&s_sub[w][(int)(((int64_t)(B[lo + q] - bmn) * 64) / bspan)], 1);
        wave_lds_sync();
        const int32_t c = s_sub[w][lane];
        int32_t incl = c;
        for (int off = 1; off < 64; off <<= 1) {
          int32_t o = __shfl_up(incl, off, 64);
          if (lane >= off) incl += o;
        }
        const int32_t sub_lo = incl - c;
        wave_lds_sync();
        s_sub[w][lane] = sub_lo;  // scatter cursor
        wave_lds_sync();
        for (int q = lane; q < sz; q += 64) {
          const int32_t v = B[lo + q];
          A[lo + atomicAdd(&s_sub[w][(int)(((int64_t)(v - bmn) * 64) / bspan)], 1)] = v;
        }
        wave_lds_sync();
        for (int sb = 0; sb < 64; ++sb) {
          const int32_t slo = __builtin_amdgcn_readlane(sub_lo, sb), ssz = __builtin_amdgcn_readlane(c, sb);
          if (ssz == 0) continue;
          if (ssz <= 64) {
            const int32_t v = lane < ssz ? A[lo + slo + lane] : 0x7FFFFFFF;
            int32_t pos = 0;
            for (int j = 0; j < ssz; ++j) pos += __builtin_amdgcn_readlane(v, j) < v ? 1 : 0;
            if (lane < ssz) col[s + lo + slo + pos] = v;
          } else {  // still crowded: rank against the sub-bucket from LDS
            for (int q = lane; q < ssz; q += 64) {
              const int32_t v = A[lo + slo + q];
              int32_t pos = 0;
              for (int j = 0; j < ssz; ++j) pos += A[lo + slo + j] < v ? 1 : 0;
              col[s + lo + slo + pos] = v;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (edges) atomicAdd(&edge_counters[(blockIdx.x & 31) * ec_stride], edges);
    __threadfence();
    s_last = atomicAdd(ticket, 1) == (int32_t)gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // the last workgroup to finish: sorted rows + queued rows + aliased rows
  int32_t total = 0;
  if (tid < 64 && (tile_edges == nullptr || tid < 32))
    total = __hip_atomic_load(&edge_counters[tid * ec_stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int32_t q = tid; q < n_tile_edges; q += 1024) total += tile_edges[q];
  for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
  if (tid == 0) s_uniq = 0;
  __syncthreads();
  if (lane == 0 && total) atomicAdd(&s_uniq, total);
  __syncthreads();
  if (tid == 0) meta[GIGL_META_N_EDGES] = s_uniq;
}

int32_t union_build_lg2(gigl_ctx* ctx, const uint32_t* roots, const gigl_tree* tree, int32_t group_roots,
                        gigl_union* out, bool multiset_rows) {
  const int b = tree->b;
  const int f0 = tree->fanouts[0], f1 = tree->fanouts[1];
  const int64_t S0 = (int64_t)b * f0, S1 = S0 * f1, T_in = b + S0;
  hipStream_t st = ctx->stream;
  int64_t n_groups = 1;
  Lg2Args g{};
  g.roots = roots;
  g.nbr0 = tree->nbr[0];
  g.nbr1 = tree->nbr[1];
  g.cnt1 = tree->cnt[1];
  g.b = b;
  g.f0 = f0;
  g.f1 = f1;
  g.S0 = S0;
  // (rows of a multi-edge graph: fewer ids than the fanout can also mean that several drawn positions held one id)
  g.whole_rows = multiset_rows ? 0 : 1;
  if (group_roots != b) {
    n_groups = b / group_roots;
    g.grouped = 1;
    g.group_roots = (uint32_t)group_roots;
    g.gdiv0 = (uint32_t)group_roots * (uint32_t)f0;
  }
  // node sub-table per batch: the inner stream of one batch at load factor <= 1/2 (a batch that fills it has more
  // inner nodes than the plan's activation workspace: reported through meta[GIGL_META_OVERFLOW])
  const int64_t inner = (int64_t)(b / n_groups) * (1 + f0);
  uint64_t cap = 1024;
  while (cap < (uint64_t)inner * 2) cap <<= 1;
  const int64_t n_slots = (int64_t)cap * n_groups;
  GIGL_REQUIRE(ctx, n_slots < (int64_t)1 << 31 && b + S0 + S1 < (int64_t)1 << 31, "batch too large");
  const int32_t n_tiles = (int32_t)((T_in + TILE - 1) / TILE);
  const int64_t zero_words = 256;  // cursor | tickets | queue counter | edge / alias counters
  int64_t need = 0;
  auto add = [&](int64_t bytes) { need += gigl_align_up(bytes, 256); };
  add(n_slots * (int64_t)sizeof(Slot));
  add(zero_words * 4);
  add(T_in * 4);
  add((int64_t)(n_tiles + 1) * 2 * 4);
  add(T_in * 4);  // queue of long rows
  add(T_in * 4);  // queue of rows to sort
  add(T_in * 4);  // queue of tiny rows to sort
  int32_t rc = gigl_arena_reset(ctx, need + 4096);
  if (rc != GIGL_OK) return rc;
  UnionArgs a{};
  a.slots = (Slot*)gigl_arena_alloc(ctx, n_slots * (int64_t)sizeof(Slot));
  int32_t* zeros = (int32_t*)gigl_arena_alloc(ctx, zero_words * 4);
  g.slot_of = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  int32_t* tile_counts = (int32_t*)gigl_arena_alloc(ctx, (int64_t)(n_tiles + 1) * 2 * 4);
  int32_t* big_rows = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  int32_t* sort_rows = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  int32_t* tiny_rows = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  if (!a.slots || !zeros || !g.slot_of || !tile_counts || !big_rows || !sort_rows || !tiny_rows)
    return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  a.mask = (uint32_t)(cap - 1);
  a.overflow = out->meta + GIGL_META_OVERFLOW;
  g.alias_base = -1;
  if (tree->nbr[1] == (const uint32_t*)(out->col + out->cap_edges) && out->cap_edges + S1 < ((int64_t)1 << 31))
    g.alias_base = (int32_t)out->cap_edges;
  int32_t* cursor = zeros;            // [0]
  int32_t* ticket_count = zeros + 1;  // [1]
  int32_t* ticket_big = zeros + 2;    // [2]
  int32_t* big_count = zeros + 3;     // [3]
  int32_t* sort_count = zeros + 4;    // [4]
  int32_t* tiny_count = zeros + 5;    // [5]
  int32_t* edge_counters = zeros + 64;   // [64..96): sorted + queued rows, [96..128): aliased rows
  const int TB = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); };
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_INSERT);
    hipLaunchKernelGGL(lg2_init_kernel, dim3(512), dim3(256), 0, st, (uint4*)a.slots, n_slots, zeros, zero_words,
                       out->meta);
    hipLaunchKernelGGL(lg2_insert_kernel, grid(T_in), dim3(TB), 0, st, a, g);
    hipLaunchKernelGGL(lg2_extras_kernel, grid(S0), dim3(TB), 0, st, a, g);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_NODES);
    hipLaunchKernelGGL(lg2_count_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, a, g, tile_counts, n_tiles,
                       ticket_count);
    hipLaunchKernelGGL(lg2_assign_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, a, g, tile_counts, n_tiles,
                       out->nodes, out->meta, out->rowptr, out->rowend, cursor, edge_counters + 32, sort_rows,
                       sort_count, tiny_rows, tiny_count);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_EDGE_SORT);
    hipLaunchKernelGGL(lg2_fill_kernel, grid(T_in), dim3(TB), 0, st, a, g, out->rowend, out->col, out->root_local,
                       out->rowptr);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_CSR);
    {
      int64_t tb = (T_in / 16 + 255) / 256;
      if (tb > 1024) tb = 1024;
      if (tb < 16) tb = 16;
      hipLaunchKernelGGL(lg2_row_sort_tiny_kernel, dim3((unsigned)tb), dim3(256), 0, st, tiny_rows, tiny_count,
                         out->rowptr, out->rowend, out->col, edge_counters);
    }
    // (the queue holds the rows of nodes that occur more than once: a fraction of T_in)
    int64_t blocks = (T_in / 8 + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 64) blocks = 64;
    hipLaunchKernelGGL(lg2_row_sort_kernel, dim3((unsigned)blocks), dim3(256), 0, st, sort_rows, sort_count,
                       out->rowptr, out->rowend, out->col, big_rows, big_count, edge_counters);
    // (a handful of workgroups; rows of more than MED_ROW entries are rare)
    if (lg_big_cap_wide()) {
      GIGL_HIP_CHECK(ctx, lds_opt_in(ctx->device, 0, (const void*)lg2_row_sort_big_kernel<BIG_ROW_CAP>, 2 * BIG_ROW_CAP * (int)sizeof(int32_t)));
      hipLaunchKernelGGL(lg2_row_sort_big_kernel<BIG_ROW_CAP>, dim3(8), dim3(1024), 2 * BIG_ROW_CAP * sizeof(int32_t), st,
                       out->rowptr, out->rowend, out->col, big_rows, big_count, out->meta + GIGL_META_OVERFLOW,
                       edge_counters, ticket_big, out->meta);
    } else
    hipLaunchKernelGGL(lg2_row_sort_big_kernel<LG_BIG_CAP>, dim3(8), dim3(1024), 2 * LG_BIG_CAP * sizeof(int32_t), st,
                       out->rowptr, out->rowend, out->col, big_rows, big_count, out->meta + GIGL_META_OVERFLOW,
                       edge_counters, ticket_big, out->meta);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}


// ------------------------------------------------------------------------------------------------------------
// "LG3": the leaf-global two-hop build with the node dedup staged in LDS (round 4).
//
// LG2 keeps one open-addressing table of 16-byte slots per batch in HBM: every inner stream position costs a scattered
// 64-bit atomic to claim / lower its slot, and the counting, numbering and fill passes each pay one or two dependent
// random 16-byte reads per position (plus a probe sequence per extra child) to get back to it.  The table was 2-8 % of
// the HBM roofline by bytes and a quarter of the step's kernel time — bound by those dependent accesses.
//
// Here the table lives in LDS for the one pass that needs a hash table at all:
//   * lg3_dedup_kernel: a batch's keys are split by hash bits over P workgroups; each holds its share in a table of
//     8-byte slots (node id << 32 | 20-bit stream code of the first occurrence << 1 | "occurs once") — claim by
//     ds_cmpst_b64, later occurrences by ds_min_u64 (smaller code wins) + ds_and_b64 (clears "once").  (A 4-byte slot
//     with an 11-bit fingerprint, verified against the stream on a match, was measured first: the verification read
//     of every repeated node sits inside the probe loop, a whole wave waits on L2 per 64 positions — 3.6 us/step.)  Every
//     workgroup streams the batch's whole inner stream (roots, hop-0 slots; L2-resident, just written by the sampler)
//     and keeps its own share; the batch's roots also go into a small exact LDS set, so "this hop-0 node is a root"
//     — whose children are inner nodes too, the extras — is known without the table of another workgroup.
//     What leaves the kernel is indexed by STREAM POSITION, not by hash: fpw[t] = code of the first occurrence of the
//     node at inner position t (| bit 31: the node occurs more than once), fpx[p * f1 + j] likewise for the children
//     of root-valued hop-0 slots, the sizes of the rows that need storage (rowcnt[first position], a few global
//     atomics per batch), and the per-1024-tile counts of first occurrences (the last workgroup to finish turns
//     them into prefixes).  insert + extras + count of LG2 in one launch, no HBM table, nothing to clear but rowcnt.
//   * lg3_assign_kernel / lg3_fill_kernel: LG2's numbering and fill with every `slots[find(id)]` replaced by an array
//     read at a known position.  Numbering order, row order and every output are LG2's, bit for bit
//     (GIGL_UNION_LG2=1 in the environment keeps LG2 for A/B runs; shapes whose per-batch stream does not fit 20-bit
//     codes or 16 partitions fall back to it).
constexpr int LG3_CODE_BITS = 20;
constexpr uint32_t LG3_CODE_MASK = (1u << LG3_CODE_BITS) - 1u;
constexpr uint32_t LG3_EMPTY = 0xFFFFFFFFu;
constexpr int32_t LG3_XTAG = 1 << 30;  // fpx entry that has become (local id | LG3_XTAG): an extra that is a first occurrence
constexpr int LG3_MAX_PARTS = 16;
constexpr int LG3_NT_DEFAULT = 512;      // threads per dedup workgroup (round 5: 512 x 128-KB tables find a CU sooner next to other kernels than 1024 — 7.0 vs 10.7 us/step under three streams; smaller tables lose to their redundant stream scans: profiles/r05d_lg3_shapes.txt)
constexpr int LG3_CAP_DEFAULT = 16384;   // slots of its LDS table (8 bytes each)

struct Lg3Args {
  const uint32_t* roots;
  const uint32_t* nbr0;
  const uint32_t* nbr1;
  const int32_t* cnt0;
  const int32_t* cnt1;
  int32_t b, f0, f1;
  int64_t S0;        // b * f0
  int32_t gr;        // roots per batch (== b: one batch)
  int32_t S0g;       // gr * f0
  int32_t Tg;        // gr * (1 + f0): codes of a batch's inner stream; the extra (pl, j) has code Tg + pl * f1 + j
  int32_t n_groups;
  int32_t P;         // hash partitions (workgroups) per batch
  uint32_t cmask;    // LDS table slots - 1
  uint32_t rmask;    // root set slots - 1
  int32_t rs_shift;  // 32 - log2(root set slots)
  int32_t bl_shift;  // 32 - log2(filter bits): 8 bits per root set slot
  int32_t* fpw;      // [b + S0] by global inner position: code of the node's first occurrence | multi << 31; -1: no node
  int32_t* fpx;      // [S0 * f1] by (global hop-0 slot, child): the same for extras; first occurrences become lid | LG3_XTAG
  int32_t* lid_in;   // [b + S0] local id, by the global inner position of the node's FIRST occurrence
  int32_t* rowcnt;   // [b + S0] entries of the node's row before dedup, by the same index (zeroed per call)
  int32_t alias_base;
  int32_t whole_rows;
  int32_t* overflow;
};

__device__ __forceinline__ uint32_t lg3_part(uint32_t h32, int P) { return (uint32_t)(((uint64_t)h32 * (uint32_t)P) >> 32); }

// global inner position (index into fpw / lid_in / rowcnt) of an INNER code of batch grp
__device__ __forceinline__ int64_t lg3_gpos(const Lg3Args& a, int32_t grp, uint32_t code) {
  return code < (uint32_t)a.gr ? (int64_t)grp * a.gr + code : (int64_t)a.b + (int64_t)grp * a.S0g + (code - a.gr);
}

// batch and group-local code of global inner position t
__device__ __forceinline__ void lg3_locate(const Lg3Args& a, int64_t t, int32_t& grp, uint32_t& code) {
  if (t < a.b) {
    grp = a.n_groups == 1 ? 0 : (int32_t)((uint32_t)t / (uint32_t)a.gr);
    code = (uint32_t)(t - (int64_t)grp * a.gr);
  } else {
    const uint32_t p = (uint32_t)(t - a.b);
    grp = a.n_groups == 1 ? 0 : (int32_t)(p / (uint32_t)a.S0g);
    code = (uint32_t)a.gr + (p - (uint32_t)grp * (uint32_t)a.S0g);
  }
}

// slot word: (node id << 32) | (code of the first occurrence << 1) | "occurs once"; ~0 = empty.  One ds_cmpst_b64 claims
// a slot for a first occurrence; a later occurrence lowers the word with ds_min_u64 (equal high halves: the smaller code
// wins) and clears "once" with ds_and_b64.
__device__ __forceinline__ bool lg3_insert(unsigned long long* table, const Lg3Args& a, uint32_t id, uint32_t h32,
                                           uint32_t code) {
  const unsigned long long mine1 = ((unsigned long long)id << 32) | ((unsigned long long)code << 1) | 1ull;
  uint32_t s = h32 & a.cmask;
  for (uint32_t probes = 0; probes <= a.cmask; ++probes) {
    unsigned long long w = table[s];
    if (w == ~0ull) {
      w = atomicCAS(&table[s], ~0ull, mine1);
      if (w == ~0ull) return true;
    }
    if ((uint32_t)(w >> 32) == id) {
      atomicMin(&table[s], mine1 & ~1ull);
      atomicAnd(&table[s], ~1ull);
      return true;
    }
    s = (s + 1) & a.cmask;
  }
  return false;
}

// the low word of `id`'s slot: (code << 1) | once; LG3_EMPTY: not in the table (only after an overflow)
__device__ __forceinline__ uint32_t lg3_lookup(const unsigned long long* table, const Lg3Args& a, uint32_t id,
                                               uint32_t h32) {
  uint32_t s = h32 & a.cmask;
  for (uint32_t probes = 0; probes <= a.cmask; ++probes) {
    const unsigned long long w = table[s];
    if (w == ~0ull) return LG3_EMPTY;
    if ((uint32_t)(w >> 32) == id) return (uint32_t)w;
    s = (s + 1) & a.cmask;
  }
  return LG3_EMPTY;
}

__device__ __forceinline__ bool lg3_rset_has(const uint32_t* rset, const Lg3Args& a, uint32_t id, uint32_t h32) {
  uint32_t s = (h32 * 0x9E3779B1u) >> a.rs_shift;
  for (;;) {  // (load factor <= 1/2: an empty slot ends every probe sequence)
    const uint32_t k = rset[s];
    if (k == id) return true;
    if (k == GIGL_INVALID) return false;
    s = (s + 1) & a.rmask;
  }
}

// an occurrence with c < f1 children sampled its node's WHOLE in-neighbourhood: the list is taken once, from the node's
// first occurrence — or from every occurrence when the node is a root (lg2_contributes)
__device__ __forceinline__ bool lg3_contributes(const Lg3Args& a, uint32_t first_code, uint32_t my_code, int c) {
  if (c >= a.f1 || !a.whole_rows) return true;
  return first_code == my_code || first_code < (uint32_t)a.gr;
}

// A level-1 node whose occurrences sampled fewer than f1 children each holds its WHOLE in-neighbourhood under every one
// of them (simple graphs: the sampler returns the ascending, duplicate-free list): its row is the tree segment of its
// first occurrence, whether the node occurs once or many times — nothing to count, fill, dedup or sort (round 4; LG2
// copies the first occurrence's list into row storage and queues the row for sorting).  `code` = the node's first code.
__device__ __forceinline__ bool lg3_row_is_segment(const Lg3Args& a, uint32_t code, int c) {
  return a.alias_base >= 0 && a.whole_rows && c < a.f1 && code >= (uint32_t)a.gr;
}

constexpr int LG3_TC = 5;  // per-tile counts: level-0 firsts, level-1 firsts, row entries to store, long / tiny rows to sort

// NT: threads per workgroup (1024, or 512 with smaller tables: more, smaller workgroups that fit next to other kernels)
template <int NT>
__global__ __launch_bounds__(NT) void lg3_dedup_kernel(Lg3Args a, int32_t* tile_counts, int32_t n_tiles,
                                                         int32_t* ticket, int32_t* sort_count, int32_t* tiny_count) {
  extern __shared__ unsigned long long lg3_lds[];
  unsigned long long* table = lg3_lds;                                      // [cmask + 1]
  uint32_t* rset = reinterpret_cast<uint32_t*>(table + (a.cmask + 1u));     // [rmask + 1] exact set of the batch's roots
  uint32_t* bloom = rset + (a.rmask + 1u);                                  // [(rmask + 1) / 4] words: 8 bits per set slot
  int32_t* tcnt = reinterpret_cast<int32_t*>(bloom + ((a.rmask + 1u) >> 2));  // [nt_r + nt_h][LG3_TC]
  __shared__ int32_t s_last;
  __shared__ int32_t s_w[NT / 64][LG3_TC];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // the P workgroups of a batch read the same stream: ids that are consecutive mod 8 share an XCD's L2
  uint32_t vid = blockIdx.x;
  if ((gridDim.x & 7u) == 0) vid = (blockIdx.x >> 3) + (blockIdx.x & 7u) * (gridDim.x >> 3);
  const int32_t grp = (int32_t)(vid / (uint32_t)a.P);
  const int q = (int)(vid % (uint32_t)a.P);
  const int64_t r0 = (int64_t)grp * a.gr, p0 = (int64_t)grp * a.S0g;
  const int32_t tile_lo_r = (int32_t)(r0 >> 10), nt_r = (int32_t)((r0 + a.gr - 1) >> 10) - tile_lo_r + 1;
  const int32_t tile_lo_h = (int32_t)((a.b + p0) >> 10), nt_h = (int32_t)((a.b + p0 + a.S0g - 1) >> 10) - tile_lo_h + 1;
  for (uint32_t i = tid; i <= a.cmask; i += NT) table[i] = ~0ull;
  for (uint32_t i = tid; i <= a.rmask; i += NT) rset[i] = GIGL_INVALID;
  for (uint32_t i = tid; i < ((a.rmask + 1u) >> 2); i += NT) bloom[i] = 0u;
  for (int i = tid; i < LG3_TC * (nt_r + nt_h); i += NT) tcnt[i] = 0;
  __syncthreads();
  int32_t over = 0;
  // ---- roots: all of them into the root set (+ its one-read filter), this partition's into the table
  for (int32_t tl = tid; tl < a.gr; tl += NT) {
    const uint32_t id = a.roots[r0 + tl];
    if (id == GIGL_INVALID) continue;
    const uint32_t h = hash_u32(id);
    uint32_t s = (h * 0x9E3779B1u) >> a.rs_shift;
    for (;;) {
      const uint32_t prev = atomicCAS(&rset[s], GIGL_INVALID, id);
      if (prev == GIGL_INVALID || prev == id) break;
      s = (s + 1) & a.rmask;
    }
    const uint32_t bi = (h * 0x85EBCA6Bu) >> a.bl_shift;
    atomicOr(&bloom[bi >> 5], 1u << (bi & 31));
    if ((int)lg3_part(h, a.P) == q && !lg3_insert(table, a, id, h, (uint32_t)tl)) ++over;
  }
  __syncthreads();
  // ---- hop-0 slots, and the children of those whose node is a root (extras).  A thread takes its slots PF at a time:
  // the ids are requested together, and the probe sequences of the PF keys advance in lock step — PF independent LDS
  // atomics in flight per round instead of one dependent chain per key (the kernel is bound by LDS latency at the four
  // waves per SIMD its table leaves room for).  Whether a slot's node is a root is decided here once (one filter read,
  // the exact set only on a hit) and kept as a bit per slot for the second pass.
  constexpr int PF = 8;
  unsigned long long rootmask = 0;  // bit (slot / NT) of this thread: the slot's node is a root
  for (int32_t base = 0, it0 = 0; base < a.S0g; base += PF * NT, it0 += PF) {
    uint32_t ids[PF], hh[PF], sl[PF], bw[PF];
    uint32_t pend = 0, valid = 0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int32_t pl = base + u * NT + tid;
      ids[u] = pl < a.S0g ? a.nbr0[p0 + pl] : GIGL_INVALID;
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      hh[u] = hash_u32(ids[u]);
      sl[u] = hh[u] & a.cmask;
      const uint32_t bi = (hh[u] * 0x85EBCA6Bu) >> a.bl_shift;
      bw[u] = (bloom[bi >> 5] >> (bi & 31)) & 1u;
      if (ids[u] != GIGL_INVALID) {
        valid |= 1u << u;
        if ((int)lg3_part(hh[u], a.P) == q) pend |= 1u << u;
      }
    }
    for (uint32_t rounds = 0; pend; ++rounds) {
      if (rounds > a.cmask) {  // the table is full
        over += __popc(pend);
        break;
      }
      unsigned long long old[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (pend >> u & 1u)
          old[u] = atomicCAS(&table[sl[u]], ~0ull,
                             ((unsigned long long)ids[u] << 32) | ((unsigned long long)(uint32_t)(a.gr + base + u * NT + tid) << 1) | 1ull);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (!(pend >> u & 1u)) continue;
        if (old[u] == ~0ull) {
          pend &= ~(1u << u);
        } else if ((uint32_t)(old[u] >> 32) == ids[u]) {
          atomicMin(&table[sl[u]], ((unsigned long long)ids[u] << 32) | ((unsigned long long)(uint32_t)(a.gr + base + u * NT + tid) << 1));
          atomicAnd(&table[sl[u]], ~1ull);
          pend &= ~(1u << u);
        } else {
          sl[u] = (sl[u] + 1) & a.cmask;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (!((valid >> u & 1u) && bw[u])) continue;
      if (!lg3_rset_has(rset, a, ids[u], hh[u])) continue;
      rootmask |= 1ull << (it0 + u);
      const int32_t pl = base + u * NT + tid;
      const int c = a.cnt1[p0 + pl];
      const uint32_t* kids = a.nbr1 + (p0 + pl) * a.f1;
      for (int j = 0; j < c; ++j) {
        const uint32_t cid = kids[j];
        const uint32_t ch = hash_u32(cid);
        if ((int)lg3_part(ch, a.P) == q && !lg3_insert(table, a, cid, ch, (uint32_t)(a.Tg + pl * a.f1 + j))) ++over;
      }
    }
  }
  if (over) atomicAdd(a.overflow, over);
  __syncthreads();
  // ---- what the later passes need, by stream position
  for (int32_t tl = tid; tl < a.gr; tl += NT) {
    const uint32_t id = a.roots[r0 + tl];
    if (id == GIGL_INVALID) {
      if (q == 0) a.fpw[r0 + tl] = -1;
      continue;
    }
    const uint32_t h = hash_u32(id);
    if ((int)lg3_part(h, a.P) != q) continue;
    const uint32_t w = lg3_lookup(table, a, id, h);
    if (w == LG3_EMPTY) {
      a.fpw[r0 + tl] = -1;
      continue;
    }
    const uint32_t code = (w >> 1) & LG3_CODE_MASK;
    a.fpw[r0 + tl] = (int32_t)(code | ((w & 1u) ? 0u : 0x80000000u));
    if (code == (uint32_t)tl) atomicAdd(&tcnt[LG3_TC * (int)(((r0 + tl) >> 10) - tile_lo_r) + 0], 1);
    // the valid hop-0 slots of this root position are entries of its node's row
    const int c0 = a.cnt0[r0 + tl];
    if (c0 > 0) atomicAdd(&a.rowcnt[r0 + code], c0);
  }
  for (int32_t base = 0, it0 = 0; base < a.S0g; base += PF * NT, it0 += PF) {
    uint32_t ids[PF], sl[PF], lw[PF];
    int32_t cc[PF];
    uint32_t pend = 0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int32_t pl = base + u * NT + tid;
      ids[u] = pl < a.S0g ? a.nbr0[p0 + pl] : GIGL_INVALID;
      cc[u] = pl < a.S0g ? a.cnt1[p0 + pl] : 0;
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const uint32_t h = hash_u32(ids[u]);
      sl[u] = h & a.cmask;
      lw[u] = LG3_EMPTY;
      if (ids[u] != GIGL_INVALID && (int)lg3_part(h, a.P) == q) pend |= 1u << u;
    }
    const uint32_t mine = pend;
    for (uint32_t rounds = 0; pend && rounds <= a.cmask; ++rounds) {
      unsigned long long wv8[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (pend >> u & 1u) wv8[u] = table[sl[u]];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (!(pend >> u & 1u)) continue;
        if (wv8[u] == ~0ull) {
          pend &= ~(1u << u);  // not in the table (only after an overflow)
        } else if ((uint32_t)(wv8[u] >> 32) == ids[u]) {
          lw[u] = (uint32_t)wv8[u];
          pend &= ~(1u << u);
        } else {
          sl[u] = (sl[u] + 1) & a.cmask;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int32_t pl = base + u * NT + tid;  // (a wave's 64 slots are consecutive: they touch at most two tiles)
      const uint32_t id = ids[u];
      const int c = cc[u];
      const bool in = pl < a.S0g;
      const int32_t my_tile = (int32_t)((a.b + p0 + pl) >> 10) - tile_lo_h;
      bool first = false;
      if (in && id == GIGL_INVALID) {
        if (q == 0) a.fpw[a.b + p0 + pl] = -1;
      } else if (in && (mine >> u & 1u)) {
        const uint32_t mycode = (uint32_t)(a.gr + pl);
        const uint32_t w = lw[u];
        if (w == LG3_EMPTY) {
          a.fpw[a.b + p0 + pl] = -1;
        } else {
          const uint32_t code = (w >> 1) & LG3_CODE_MASK;
          const bool multi = !(w & 1u);
          a.fpw[a.b + p0 + pl] = (int32_t)(code | (multi ? 0x80000000u : 0u));
          first = code == mycode;
          // the children of a node that occurs more than once go to that node's row (a node that occurs once keeps
          // its tree segment as its row)
          if ((multi || a.alias_base < 0) && c > 0 && !lg3_row_is_segment(a, code, c) && lg3_contributes(a, code, mycode, c))
            atomicAdd(&a.rowcnt[lg3_gpos(a, grp, code)], c);
        }
      }
      {  // first occurrences per tile: one LDS atomic per wave and tile
        const int32_t tile0 = __builtin_amdgcn_readfirstlane(my_tile);
        const unsigned long long f0m = __ballot(first && my_tile == tile0), f1m = __ballot(first && my_tile != tile0);
        if (lane == 0) {
          if (f0m) atomicAdd(&tcnt[LG3_TC * (nt_r + tile0) + 1], (int32_t)__popcll(f0m));
          if (f1m) atomicAdd(&tcnt[LG3_TC * (nt_r + tile0 + 1) + 1], (int32_t)__popcll(f1m));
        }
      }
      if (rootmask >> (it0 + u) & 1ull) {
        const uint32_t* kids = a.nbr1 + (p0 + pl) * a.f1;
        for (int j = 0; j < c; ++j) {
          const uint32_t cid = kids[j];
          const uint32_t ch = hash_u32(cid);
          if ((int)lg3_part(ch, a.P) != q) continue;
          const uint32_t xcode = (uint32_t)(a.Tg + pl * a.f1 + j);
          const uint32_t w = lg3_lookup(table, a, cid, ch);
          int32_t out = -1;
          if (w != LG3_EMPTY) {
            const uint32_t code = (w >> 1) & LG3_CODE_MASK;
            out = (int32_t)(code | ((w & 1u) ? 0u : 0x80000000u));
            // (an extra that is a first occurrence is numbered by the thread of its parent slot: counted in its tile)
            if (code == xcode) atomicAdd(&tcnt[LG3_TC * (nt_r + my_tile) + 1], 1);
          }
          a.fpx[(p0 + pl) * a.f1 + j] = out;
        }
      }
    }
  }
  // ---- third pass: the storage and the sort queue entries the first occurrences of each tile need.  A node's rowcnt is
  // only ever added to by the workgroup of the node's partition (root positions and hop-0 occurrences of the node
  // itself), i.e. by THIS workgroup: after the barrier the sizes are final, and the numbering pass finds every offset
  // as a prefix — no cursor, no atomic, rows land where their position says.
  __syncthreads();
  for (int32_t tl = tid; tl < a.gr; tl += NT) {
    const uint32_t id = a.roots[r0 + tl];
    if (id == GIGL_INVALID || (int)lg3_part(hash_u32(id), a.P) != q) continue;
    const int32_t w = a.fpw[r0 + tl];  // (this thread's own store)
    if (w == -1 || ((uint32_t)w & 0x7FFFFFFFu) != (uint32_t)tl) continue;
    const int32_t len = __hip_atomic_load(&a.rowcnt[r0 + tl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int32_t* tc = &tcnt[LG3_TC * (int)(((r0 + tl) >> 10) - tile_lo_r)];
    if (len) atomicAdd(&tc[2], len);
    if (len >= 2 && w < 0) atomicAdd(&tc[len > TINY_ROW ? 3 : 4], 1);
  }
  for (int32_t base = 0; base < a.S0g; base += PF * NT) {
    uint32_t ids[PF];
    int32_t cc[PF], ww[PF], len[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int32_t pl = base + u * NT + tid;
      ids[u] = pl < a.S0g ? a.nbr0[p0 + pl] : GIGL_INVALID;
      cc[u] = pl < a.S0g ? a.cnt1[p0 + pl] : 0;
      ww[u] = pl < a.S0g ? a.fpw[a.b + p0 + pl] : -1;  // (this thread's own store, or another partition's slot)
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int32_t pl = base + u * NT + tid;
      len[u] = -1;  // -1: not a first occurrence of this partition whose row needs storage
      if (ids[u] == GIGL_INVALID || ww[u] == -1 || (int)lg3_part(hash_u32(ids[u]), a.P) != q) continue;
      const uint32_t mycode = (uint32_t)(a.gr + pl);
      if (((uint32_t)ww[u] & 0x7FFFFFFFu) != mycode) continue;
      const bool multi = ww[u] < 0;
      if (a.alias_base >= 0 && (!multi || lg3_row_is_segment(a, mycode, cc[u]))) continue;  // the row is a tree segment
      len[u] = __hip_atomic_load(&a.rowcnt[a.b + p0 + pl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int32_t pl = base + u * NT + tid;
      const int32_t my_tile = (int32_t)((a.b + p0 + pl) >> 10) - tile_lo_h;
      const int32_t tile0 = __builtin_amdgcn_readfirstlane(my_tile);
      const bool have = len[u] > 0, sorted = len[u] >= 2 && ww[u] < 0;
      if (!__ballot(have)) continue;
      const bool t0 = my_tile == tile0;
      const unsigned long long q0 = __ballot(sorted && len[u] > TINY_ROW && t0), q1 = __ballot(sorted && len[u] > TINY_ROW && !t0);
      const unsigned long long s0 = __ballot(sorted && len[u] <= TINY_ROW && t0), s1 = __ballot(sorted && len[u] <= TINY_ROW && !t0);
      const int32_t n0 = __builtin_amdgcn_readlane(gigl_wave_incl_scan(have && t0 ? len[u] : 0), 63);
      const int32_t n1 = __builtin_amdgcn_readlane(gigl_wave_incl_scan(have && !t0 ? len[u] : 0), 63);
      if (lane == 0) {
        int32_t* tc = &tcnt[LG3_TC * (nt_r + tile0)];
        if (n0) atomicAdd(&tc[2], n0);
        if (q0) atomicAdd(&tc[3], (int32_t)__popcll(q0));
        if (s0) atomicAdd(&tc[4], (int32_t)__popcll(s0));
        if (n1) atomicAdd(&tc[LG3_TC + 2], n1);
        if (q1) atomicAdd(&tc[LG3_TC + 3], (int32_t)__popcll(q1));
        if (s1) atomicAdd(&tc[LG3_TC + 4], (int32_t)__popcll(s1));
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < LG3_TC * (nt_r + nt_h); i += NT) {
    const int32_t v = tcnt[i];
    if (v) {
      const int k = i / LG3_TC;
      const int32_t tile = k < nt_r ? tile_lo_r + k : tile_lo_h + (k - nt_r);
      atomicAdd(&tile_counts[tile * LG3_TC + (i - k * LG3_TC)], v);
    }
  }
  __syncthreads();
  if (tid == 0) {
    __threadfence();  // the counts are visible device-wide before the ticket is taken
    s_last = atomicAdd(ticket, 1) == (int32_t)gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  // the last workgroup to finish: exclusive prefixes over the tiles, NT tiles per round (row n_tiles = totals)
  int32_t run[LG3_TC];
#pragma unroll
  for (int c = 0; c < LG3_TC; ++c) run[c] = 0;
  for (int32_t t0 = 0; t0 < n_tiles; t0 += NT) {
    const int32_t i = t0 + tid;
    int32_t v[LG3_TC], inc[LG3_TC];
#pragma unroll
    for (int c = 0; c < LG3_TC; ++c) {
      v[c] = i < n_tiles ? __hip_atomic_load(&tile_counts[i * LG3_TC + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      inc[c] = gigl_wave_incl_scan(v[c]);
    }
    if (lane == 63) {
#pragma unroll
      for (int c = 0; c < LG3_TC; ++c) s_w[wv][c] = inc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < LG3_TC; ++c) {
      int32_t before = 0, tot = 0;
      for (int k = 0; k < NT / 64; ++k) {
        const int32_t x = s_w[k][c];
        if (k < wv) before += x;
        tot += x;
      }
      if (i < n_tiles) tile_counts[i * LG3_TC + c] = run[c] + before + inc[c] - v[c];
      run[c] += tot;
    }
    __syncthreads();
  }
  if (tid == 0) {
#pragma unroll
    for (int c = 0; c < LG3_TC; ++c) tile_counts[n_tiles * LG3_TC + c] = run[c];
    *sort_count = run[3];  // what the sort kernels read: rows queued by the numbering pass
    *tiny_count = run[4];
  }
}

__global__ __launch_bounds__(256) void lg3_init_kernel(int32_t* rowcnt, int64_t n_rowcnt, int32_t* zeros, int64_t zero_words,
                                                       int32_t* meta) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < n_rowcnt; i += stride) rowcnt[i] = 0;
  for (int64_t i = t0; i < zero_words; i += stride) zeros[i] = 0;
  if (t0 < GIGL_META_LEN) meta[t0] = 0;
}

// first occurrences held by inner stream position t (lg2_firsts over the position-indexed arrays)
__device__ __forceinline__ void lg3_firsts(const Lg3Args& a, int64_t t, int32_t& w_out, int32_t& grp, uint32_t& mycode,
                                           int& c0, int& c1, uint32_t& own_first, unsigned long long& xmask,
                                           bool& root_parent) {
  c0 = c1 = 0;
  own_first = 0;
  xmask = 0;
  root_parent = false;
  w_out = -1;
  grp = 0;
  mycode = 0;
  if (t >= a.b + a.S0) return;
  const int32_t w = a.fpw[t];
  w_out = w;
  if (w == -1) return;
  lg3_locate(a, t, grp, mycode);
  const uint32_t code = (uint32_t)w & 0x7FFFFFFFu;
  if (code == mycode) {
    own_first = 1;
    if (t < a.b) c0 = 1; else c1 = 1;
  }
  if (t >= a.b && code < (uint32_t)a.gr) {
    root_parent = true;
    const int64_t p = t - a.b;
    const int c = a.cnt1[p];
    const uint32_t xbase = (uint32_t)a.Tg + (mycode - (uint32_t)a.gr) * (uint32_t)a.f1;
    for (int j = 0; j < c; ++j) {
      const int32_t x = a.fpx[p * a.f1 + j];
      if (x != -1 && ((uint32_t)x & 0x7FFFFFFFu) == xbase + j) {
        xmask |= 1ull << j;
        ++c1;
      }
    }
  }
}

__global__ __launch_bounds__(256) void lg3_assign_kernel(Lg3Args a, const int32_t* tile_counts, int32_t n_tiles,
                                                         uint32_t* nodes, int32_t* meta, int32_t* rowptr, int32_t* rowend,
                                                         int32_t* tile_edges, int32_t* sort_rows, int32_t* tiny_rows) {
  // per (sub-row, wave): level-0 firsts, level-1 firsts, row entries needed, rows queued for sorting (long | tiny); after
  // the barrier: their exclusive prefixes in (sub-row, wave) order
  __shared__ int32_t s_w[TILE / 64 + 1][5];
  __shared__ int32_t s_alias[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t* tcb = tile_counts + (int64_t)blockIdx.x * LG3_TC;
  const int32_t before0 = tcb[0], before1 = tcb[1];
  const int32_t s_row_base = tcb[2], s_q_base = tcb[3], s_t_base = tcb[4];  // (the dedup pass left every prefix: no cursor)
  const int32_t total0 = tile_counts[n_tiles * LG3_TC + 0], total1 = tile_counts[n_tiles * LG3_TC + 1];
  const int64_t base = (int64_t)blockIdx.x * TILE;
  const unsigned long long below = (1ull << lane) - 1ull;
  int32_t fw[TILE / 256];
  int k0[TILE / 256], k1[TILE / 256];
  uint32_t own[TILE / 256];
  unsigned long long xm[TILE / 256];
  int32_t own_len[TILE / 256];  // entries of the own first's row, -1: the row is a tree segment (aliased)
  int32_t x0[TILE / 256], x1[TILE / 256], xn[TILE / 256], xq[TILE / 256], xt[TILE / 256];  // exclusive wave prefixes
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    const int64_t t = base + r * 256 + tid;
    int32_t grp;
    uint32_t mycode;
    bool rp;
    lg3_firsts(a, t, fw[r], grp, mycode, k0[r], k1[r], own[r], xm[r], rp);
    int32_t nd = 0, nq = 0, nt = 0;
    own_len[r] = 0;
    if (own[r]) {
      const bool multi = fw[r] < 0;
      const bool alias = a.alias_base >= 0 && t >= a.b && (!multi || lg3_row_is_segment(a, mycode, a.cnt1[t - a.b]));
      own_len[r] = alias ? -1 : a.rowcnt[t];
      if (!alias) {
        nd += own_len[r];
        const bool sort_it = own_len[r] >= 2 && multi;
        nq = sort_it && own_len[r] > TINY_ROW ? 1 : 0;
        nt = sort_it && own_len[r] <= TINY_ROW ? 1 : 0;
      }
    }
    // (a node first seen as an extra has no hop-0 occurrence, hence no in-edges: its row is empty)
    // 0 / 1 flags are ranked by ballot + popcount, the two counts by a DPP scan: nothing goes through the LDS pipeline
    const unsigned long long m0 = __ballot(k0[r] != 0), mq = __ballot(nq != 0), mt = __ballot(nt != 0);
    const int32_t i1 = gigl_wave_incl_scan(k1[r]), in = gigl_wave_incl_scan(nd);
    x0[r] = (int32_t)__popcll(m0 & below);
    x1[r] = i1 - k1[r];
    xn[r] = in - nd;
    xq[r] = (int32_t)__popcll(mq & below);
    xt[r] = (int32_t)__popcll(mt & below);
    if (lane == 63) {
      s_w[r * 4 + w][0] = (int32_t)__popcll(m0);
      s_w[r * 4 + w][1] = i1;
      s_w[r * 4 + w][2] = in;
      s_w[r * 4 + w][3] = (int32_t)__popcll(mq);
      s_w[r * 4 + w][4] = (int32_t)__popcll(mt);
    }
  }
  __syncthreads();
  if (w == 0) {  // exclusive prefixes over the TILE / 64 (sub-row, wave) entries, the totals in the entry after them
    int32_t v[5], e[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      v[c] = lane < TILE / 64 ? s_w[lane][c] : 0;
      e[c] = gigl_wave_incl_scan(v[c]) - v[c];
    }
#pragma unroll
    for (int c = 0; c < 5; ++c)
      if (lane <= TILE / 64) s_w[lane][c] = e[c];  // (lane TILE / 64 holds the sum of all entries)
  }
  __syncthreads();
  int32_t alias_c = 0;
#pragma unroll
  for (int r = 0; r < TILE / 256; ++r) {
    if (!(k0[r] | k1[r])) continue;
    const int32_t* pre = s_w[r * 4 + w];
    int32_t p0 = before0 + x0[r] + pre[0], p1 = total0 + before1 + x1[r] + pre[1], pn = s_row_base + xn[r] + pre[2],
            pq = s_q_base + xq[r] + pre[3], pt = s_t_base + xt[r] + pre[4];
    const int64_t t = base + r * 256 + tid;
    if (own[r]) {
      const bool multi = fw[r] < 0;
      const int32_t id = k0[r] ? p0 : p1++;
      a.lid_in[t] = id;
      nodes[id] = t < a.b ? a.roots[t] : a.nbr0[t - a.b];
      if (own_len[r] < 0) {  // the row IS the tree segment of its children
        const int64_t p = t - a.b;
        const int32_t c = a.cnt1[p];
        rowptr[id] = a.alias_base + (int32_t)(p * a.f1);
        rowend[id] = a.alias_base + (int32_t)(p * a.f1) + c;
        alias_c += c;
      } else {
        const bool sorted_later = own_len[r] >= 2 && multi;
        rowptr[id] = pn;
        rowend[id] = sorted_later || own_len[r] < 2 ? pn : pn + own_len[r];  // fill cursor | final end
        pn += own_len[r];
        if (sorted_later && own_len[r] > TINY_ROW) sort_rows[pq] = id;
        else if (sorted_later) tiny_rows[pt] = id;
        else alias_c += own_len[r];  // (the row is final: counted with the aliased edges)
      }
    }
    if (xm[r]) {
      const int64_t p = t - a.b;
      for (int j = 0; j < a.f1; ++j) {
        if (!(xm[r] >> j & 1ull)) continue;
        const int32_t id = p1++;
        a.fpx[p * a.f1 + j] = id | LG3_XTAG;
        nodes[id] = a.nbr1[p * a.f1 + j];
        rowptr[id] = pn;
        rowend[id] = pn;
      }
    }
  }
  {  // edges of rows that are final here (tree segments, rows of roots that occur once): one plain store per tile
    for (int off = 32; off > 0; off >>= 1) alias_c += __shfl_xor(alias_c, off, 64);
    if (lane == 0) s_alias[w] = alias_c;
    __syncthreads();
    if (tid == 0) tile_edges[blockIdx.x] = s_alias[0] + s_alias[1] + s_alias[2] + s_alias[3];
  }
  if (blockIdx.x == 0 && tid == 0) {
    meta[GIGL_META_LEVEL0] = total0;
    meta[GIGL_META_LEVEL0 + 1] = total0 + total1;
    meta[GIGL_META_LEVEL0 + 2] = total0 + total1;
    meta[GIGL_META_N_NODES] = total0 + total1;
    rowptr[total0 + total1] = 0;  // (rows past the inner nodes do not exist in leaf-global mode)
    rowend[total0 + total1] = 0;
  }
}

__global__ __launch_bounds__(256) void lg3_fill_kernel(Lg3Args a, int32_t* rowend, int32_t* col, int32_t* root_local,
                                                       const int32_t* rowptr) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = t < a.b + a.S0;
  const int32_t w = in ? a.fpw[t] : -1;
  const bool valid = w != -1;
  int32_t grp = 0;
  uint32_t mycode = 0;
  if (in) lg3_locate(a, t, grp, mycode);
  const uint32_t code = (uint32_t)w & 0x7FFFFFFFu;
  const bool multi = valid && w < 0;
  const int32_t lid = valid ? a.lid_in[lg3_gpos(a, grp, code)] : -1;
  if (in && t < a.b) root_local[t] = lid;
  const bool edge = in && t >= a.b && valid;
  bool e_ok = false;
  int32_t dl = 0;
  if (edge) {
    const int64_t rt = (int64_t)((uint32_t)(t - a.b) / (uint32_t)a.f0);  // the root position this slot hangs under
    const int32_t wr = a.fpw[rt];
    if (wr != -1) {
      e_ok = true;
      const int64_t rpos = (int64_t)grp * a.gr + ((uint32_t)wr & 0x7FFFFFFFu);  // (a root position's node starts at a root position)
      dl = a.lid_in[rpos];
      if (wr >= 0 && a.rowcnt[rpos] >= 2) {
        // the root occurs once: its row is its hop-0 slots' nodes in slot order (valid slots are a prefix) — no cursor
        col[rowptr[dl] + (int32_t)((uint32_t)(t - a.b) % (uint32_t)a.f0)] = lid;
        e_ok = false;
      }
    }
  }
  int total, rank, first;
  seg_rank(e_ok ? (uint32_t)dl : (0x80000000u | (uint32_t)lane), e_ok, lane, total, rank, first);
  int32_t basep = 0;
  if (e_ok && rank == 0) basep = atomicAdd(&rowend[dl], total);
  basep = __shfl(basep, first, 64);
  if (e_ok) col[basep + rank] = lid;
  if (edge && (multi || a.alias_base < 0)) {
    const int64_t p = t - a.b;
    const int c = a.cnt1[p];
    if (c > 0 && !lg3_row_is_segment(a, code, c) && lg3_contributes(a, code, mycode, c)) {
      const int32_t at = atomicAdd(&rowend[lid], c);
      const bool local = code < (uint32_t)a.gr;  // the node is a root: its row holds local ids
      const int64_t xg = (int64_t)grp * a.S0g * a.f1;  // the batch's extras
      for (int j = 0; j < c; ++j) {
        int32_t v = (int32_t)a.nbr1[p * a.f1 + j];
        if (local) {
          int32_t x = a.fpx[p * a.f1 + j];
          v = 0;
          if (x != -1) {
            if (!(x & LG3_XTAG)) {
              const uint32_t c2 = (uint32_t)x & 0x7FFFFFFFu;
              x = c2 < (uint32_t)a.Tg ? a.lid_in[lg3_gpos(a, grp, c2)] | LG3_XTAG : a.fpx[xg + (c2 - a.Tg)];
            }
            v = x & (LG3_XTAG - 1);
          }
        }
        col[at + j] = v;
      }
    }
  }
}

int32_t union_build_lg3(gigl_ctx* ctx, const uint32_t* roots, const gigl_tree* tree, int32_t group_roots,
                        gigl_union* out, bool multiset_rows, bool* taken) {
  *taken = false;
  const int b = tree->b;
  const int f0 = tree->fanouts[0], f1 = tree->fanouts[1];
  const int64_t S0 = (int64_t)b * f0, S1 = S0 * f1, T_in = b + S0;
  const int64_t n_groups = b / group_roots;
  const int64_t gr = group_roots, Tg = gr * (1 + f0), S1g = gr * f0 * f1;
  // (20-bit stream codes; a thread of the dedup pass keeps one bit per 1024 hop-0 slots of the batch in a 64-bit mask)
  // workgroup shape of the dedup pass (round 5): GIGL_LG3_NT = 512 | 1024 threads, GIGL_LG3_CAP slots per table
  static const int lg3_nt_env = [] {
    const char* e = getenv("GIGL_LG3_NT");
    const int v = e ? atoi(e) : 0;
    return v == 512 || v == 1024 ? v : 0;
  }();
  // (a thread keeps one bit per NT hop-0 slots of its batch in a 64-bit mask: batches of more than 64 * 512 slots — B = 4096
  // at fan-out 15 — take the 1024-thread shape)
  int lg3_nt = lg3_nt_env ? lg3_nt_env : (gr * f0 <= 64 * (int64_t)LG3_NT_DEFAULT ? LG3_NT_DEFAULT : 1024);
  if (Tg + S1g >= (int64_t)LG3_CODE_MASK || f1 > 64 || b + S0 + S1 >= ((int64_t)1 << 31) || gr * f0 > 64 * (int64_t)lg3_nt) return GIGL_OK;  // -> LG2
  // LDS table: 8-byte slots at load <= 1/2 for the nodes a batch may hold (more do not fit the plan's workspace)
  static const int64_t cap_max = [] {
    const char* e = getenv("GIGL_LG3_CAP");  // (A/B knob: slots per workgroup, a power of two)
    const int64_t v = e ? atoll(e) : 0;
    return v >= 1024 && v <= 16384 && (v & (v - 1)) == 0 ? v : (int64_t)LG3_CAP_DEFAULT;
  }();
  int64_t cap = 1024;
  while (cap < 2 * Tg && cap < cap_max) cap <<= 1;
  const int64_t P = (2 * Tg + cap - 1) / cap;
  if (P > LG3_MAX_PARTS) return GIGL_OK;
  // (the 512-thread shape pays when the launch fills the GPU — a workgroup per CU and more, other streams' kernels beside
  // it; a launch of fewer workgroups than CUs is over sooner with 1024 threads each: the sharded plan's 16-batch calls,
  // lg3_dedup 3.6 -> 5.0 us per rank-step with 512, profiles/r05e_emulated_world8_kernel_time.txt)
  if (!lg3_nt_env && n_groups * P < 256) lg3_nt = 1024;
  // a call of a few batches (a training step: one) leaves most CUs without a workgroup of the dedup pass, whose duration
  // is that of ONE workgroup walking its batch's whole stream (~80 us at [25,10] x 1024): LG2's position-parallel
  // launches finish such a call sooner (training step 0.42 -> 0.39 ms)
  static const int64_t min_wgs = getenv("GIGL_LG3_MIN_WGS") ? atoll(getenv("GIGL_LG3_MIN_WGS")) : 32;
  if (n_groups * P < min_wgs) return GIGL_OK;
  int64_t rs = 64;
  int rs_log2 = 6;
  while (rs < 2 * gr) {
    rs <<= 1;
    ++rs_log2;
  }
  const int64_t n_tcnt = LG3_TC * ((gr >> 10) + 2 + ((gr * f0) >> 10) + 2);
  const size_t lds_bytes = (size_t)cap * 8 + (size_t)(rs + rs / 4 + n_tcnt) * 4;
  if (lds_bytes > 150 * 1024) return GIGL_OK;
  *taken = true;
  hipStream_t st = ctx->stream;
  const int32_t n_tiles = (int32_t)((T_in + TILE - 1) / TILE);
  constexpr int EC_STRIDE = 32;  // every spread edge counter on its own 128-byte line (same-line atomics serialise in L2)
  const int64_t zero_words = 64 + 32 * EC_STRIDE + (int64_t)(n_tiles + 1) * LG3_TC;  // scalars | edge counters | tile counts
  int64_t need = 0;
  auto add = [&](int64_t bytes) { need += gigl_align_up(bytes, 256); };
  add(zero_words * 4);
  add((int64_t)n_tiles * 4);  // edges of the rows each tile finished
  add(T_in * 4);  // fpw
  add(T_in * 4);  // lid_in
  add(T_in * 4);  // rowcnt
  add(S1 * 4);    // fpx
  add(T_in * 4);  // queue of long rows
  add(T_in * 4);  // queue of rows to sort
  add(T_in * 4);  // queue of tiny rows to sort
  int32_t rc = gigl_arena_reset(ctx, need + 4096);
  if (rc != GIGL_OK) return rc;
  int32_t* zeros = (int32_t*)gigl_arena_alloc(ctx, zero_words * 4);
  int32_t* tile_edges = (int32_t*)gigl_arena_alloc(ctx, (int64_t)n_tiles * 4);
  Lg3Args a{};
  a.fpw = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  a.lid_in = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  a.rowcnt = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  a.fpx = (int32_t*)gigl_arena_alloc(ctx, S1 * 4);
  int32_t* big_rows = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  int32_t* sort_rows = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  int32_t* tiny_rows = (int32_t*)gigl_arena_alloc(ctx, T_in * 4);
  if (!zeros || !tile_edges || !a.fpw || !a.lid_in || !a.rowcnt || !a.fpx || !big_rows || !sort_rows || !tiny_rows)
    return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  a.roots = roots;
  a.nbr0 = tree->nbr[0];
  a.nbr1 = tree->nbr[1];
  a.cnt0 = tree->cnt[0];
  a.cnt1 = tree->cnt[1];
  a.b = b;
  a.f0 = f0;
  a.f1 = f1;
  a.S0 = S0;
  a.gr = (int32_t)gr;
  a.S0g = (int32_t)(gr * f0);
  a.Tg = (int32_t)Tg;
  a.n_groups = (int32_t)n_groups;
  a.P = (int32_t)P;
  a.cmask = (uint32_t)(cap - 1);
  a.rmask = (uint32_t)(rs - 1);
  a.rs_shift = 32 - rs_log2;
  a.bl_shift = 32 - (rs_log2 + 3);
  a.whole_rows = multiset_rows ? 0 : 1;
  a.overflow = out->meta + GIGL_META_OVERFLOW;
  a.alias_base = -1;
  if (tree->nbr[1] == (const uint32_t*)(out->col + out->cap_edges) && out->cap_edges + S1 < ((int64_t)1 << 31))
    a.alias_base = (int32_t)out->cap_edges;
  int32_t* ticket_count = zeros + 1;  // [1]
  int32_t* ticket_big = zeros + 2;    // [2]
  int32_t* big_count = zeros + 3;     // [3]
  int32_t* sort_count = zeros + 4;    // [4]
  int32_t* tiny_count = zeros + 5;    // [5]
  int32_t* edge_counters = zeros + 64;  // 32 counters EC_STRIDE words apart: sorted + queued rows
  int32_t* tile_counts = zeros + 64 + 32 * EC_STRIDE;
  const int TB = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); };
  GIGL_HIP_CHECK(ctx, lds_opt_in(ctx->device, 1, (const void*)lg3_dedup_kernel<1024>, 150 * 1024));
  GIGL_HIP_CHECK(ctx, lds_opt_in(ctx->device, 3, (const void*)lg3_dedup_kernel<512>, 150 * 1024));
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_INSERT);
    hipLaunchKernelGGL(lg3_init_kernel, dim3(256), dim3(256), 0, st, a.rowcnt, T_in, zeros, zero_words, out->meta);
    if (lg3_nt == 512)
      hipLaunchKernelGGL(lg3_dedup_kernel<512>, dim3((unsigned)(n_groups * P)), dim3(512), lds_bytes, st, a, tile_counts,
                         n_tiles, ticket_count, sort_count, tiny_count);
    else
      hipLaunchKernelGGL(lg3_dedup_kernel<1024>, dim3((unsigned)(n_groups * P)), dim3(1024), lds_bytes, st, a, tile_counts,
                         n_tiles, ticket_count, sort_count, tiny_count);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_NODES);
    hipLaunchKernelGGL(lg3_assign_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, a, tile_counts, n_tiles, out->nodes,
                       out->meta, out->rowptr, out->rowend, tile_edges, sort_rows, tiny_rows);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_EDGE_SORT);
    hipLaunchKernelGGL(lg3_fill_kernel, grid(T_in), dim3(TB), 0, st, a, out->rowend, out->col, out->root_local, out->rowptr);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_CSR);
    {
      // (a thread per row: one pass over the queue when up to a quarter of the inner positions own such a row)
      static const int64_t tiny_div = getenv("GIGL_LG3_TINY_DIV") ? atoll(getenv("GIGL_LG3_TINY_DIV")) : 4;
      int64_t tb = (T_in / (tiny_div > 0 ? tiny_div : 4) + 255) / 256;
      if (tb > 4096) tb = 4096;
      if (tb < 16) tb = 16;
      hipLaunchKernelGGL(lg2_row_sort_tiny_kernel, dim3((unsigned)tb), dim3(256), 0, st, tiny_rows, tiny_count,
                         out->rowptr, out->rowend, out->col, edge_counters, EC_STRIDE);
    }
    int64_t blocks = (T_in / 8 + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 64) blocks = 64;
    hipLaunchKernelGGL(lg2_row_sort_kernel, dim3((unsigned)blocks), dim3(256), 0, st, sort_rows, sort_count, out->rowptr,
                       out->rowend, out->col, big_rows, big_count, edge_counters, EC_STRIDE);
    if (lg_big_cap_wide()) {
      GIGL_HIP_CHECK(ctx, lds_opt_in(ctx->device, 4, (const void*)lg2_row_sort_big_kernel<BIG_ROW_CAP>, 2 * BIG_ROW_CAP * (int)sizeof(int32_t)));
      hipLaunchKernelGGL(lg2_row_sort_big_kernel<BIG_ROW_CAP>, dim3(8), dim3(1024), 2 * BIG_ROW_CAP * sizeof(int32_t), st, out->rowptr,
                       out->rowend, out->col, big_rows, big_count, out->meta + GIGL_META_OVERFLOW, edge_counters,
                       ticket_big, out->meta, EC_STRIDE, (const int32_t*)tile_edges, n_tiles);
    } else
    hipLaunchKernelGGL(lg2_row_sort_big_kernel<LG_BIG_CAP>, dim3(8), dim3(1024), 2 * LG_BIG_CAP * sizeof(int32_t), st, out->rowptr,
                       out->rowend, out->col, big_rows, big_count, out->meta + GIGL_META_OVERFLOW, edge_counters,
                       ticket_big, out->meta, EC_STRIDE, (const int32_t*)tile_edges, n_tiles);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // namespace

extern "C" {

int32_t gigl_union_capacity(int32_t b, const int32_t* fanouts, int32_t hops, int64_t* cap_nodes,
                            int64_t* cap_edges) {
  if (!fanouts || hops < 1 || hops > GIGL_MAX_HOPS || b < 0) return GIGL_E_INVALID_ARG;
  int64_t parents = b, edges = 0;
  for (int k = 0; k < hops; ++k) {
    parents *= fanouts[k];
    edges += parents;
  }
  if (cap_nodes) *cap_nodes = edges + b;
  if (cap_edges) *cap_edges = edges;
  return GIGL_OK;
}

int32_t gigl_union_build(gigl_ctx* ctx, const uint32_t* roots, const gigl_tree* tree,
                         gigl_union* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, tree, "null argument");
  return gigl_union_build_groups(ctx, roots, tree, tree->b, out);
}

int32_t gigl_union_build_groups(gigl_ctx* ctx, const uint32_t* roots, const gigl_tree* tree,
                                int32_t group_roots, gigl_union* out) {
  return gigl_union_build_impl(ctx, roots, tree, group_roots, out, 0);
}

}  // extern "C"

// leaf_global != 0 (internal, the one-call plan): see UnionArgs::leaf_global.  Only for hops <= 2 and node ids
// < 2^31 (row values are compared as int32).  Differences of the output: nodes / meta count the nodes of level <
// hops only (meta[LEVEL0 + hops] == meta[LEVEL0 + hops - 1] == n_nodes), and the rows of level hops-1 hold global
// ids; everything else (rows of lower levels, root_local, n_edges, row order) is as documented in gigl_hip.h.
int32_t gigl_union_build_impl(gigl_ctx* ctx, const uint32_t* roots, const gigl_tree* tree, int32_t group_roots,
                              gigl_union* out, int32_t leaf_global) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, tree && out && (roots || tree->b == 0), "null argument");
  GIGL_REQUIRE(ctx, !leaf_global || tree->hops <= 2, "leaf-global union needs hops <= 2");
  GIGL_REQUIRE(ctx, out->meta && out->nodes && out->rowptr && out->rowend && out->col && out->root_local,
               "union output buffers are null");
  const int hops = tree->hops, b = tree->b;
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS && b >= 0, "bad tree");
  int64_t cap_nodes, cap_edges;
  gigl_union_capacity(b, tree->fanouts, hops, &cap_nodes, &cap_edges);
  GIGL_REQUIRE(ctx, out->cap_nodes >= cap_nodes && out->cap_edges >= cap_edges,
               "union buffers too small: need %lld nodes / %lld edges", (long long)cap_nodes,
               (long long)cap_edges);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (b == 0) {
    GIGL_HIP_CHECK(ctx, hipMemsetAsync(out->meta, 0, GIGL_META_LEN * sizeof(int32_t), st));
    GIGL_HIP_CHECK(ctx, hipMemsetAsync(out->rowptr, 0, sizeof(int32_t), st));
    GIGL_HIP_CHECK(ctx, hipMemsetAsync(out->rowend, 0, sizeof(int32_t), st));
    return GIGL_OK;
  }
  if (leaf_global && hops == 2 && !getenv("GIGL_UNION_GENERIC")) {
    GIGL_REQUIRE(ctx, group_roots >= 1 && b % group_roots == 0, "group_roots=%d does not divide b=%d", group_roots, b);
    static const bool keep_lg2 = getenv("GIGL_UNION_LG2") != nullptr;  // (A/B knob)
    if (!keep_lg2) {
      bool taken = false;
      const int32_t rc3 = union_build_lg3(ctx, roots, tree, group_roots, out, (leaf_global & 2) != 0, &taken);
      if (rc3 != GIGL_OK || taken) return rc3;
    }
    return union_build_lg2(ctx, roots, tree, group_roots, out, (leaf_global & 2) != 0);
  }

  UnionArgs a{};
  a.roots = roots;
  a.b = b;
  a.hops = hops;
  a.leaf_global = leaf_global ? 1 : 0;  // (bit 1 of the argument: the sampled graph keeps multi-edges)
  int64_t T = b, parents = b;
  for (int k = 0; k < hops; ++k) {
    a.nbr[k] = tree->nbr[k];
    a.fan[k] = tree->fanouts[k];
    a.off[k] = T;
    parents *= tree->fanouts[k];
    T += parents;
  }
  a.off[hops] = T;
  a.T = T;
  GIGL_REQUIRE(ctx, T < (int64_t)1 << 31, "batch too large: %lld stream positions", (long long)T);
  int64_t n_groups = 1;
  if (group_roots != b) {
    GIGL_REQUIRE(ctx, group_roots >= 1 && b % group_roots == 0, "group_roots=%d does not divide b=%d", group_roots, b);
    n_groups = b / group_roots;
    a.grouped = 1;
    a.group_roots = (uint32_t)group_roots;
    int64_t per_group = group_roots;
    for (int k = 0; k < hops; ++k) {
      per_group *= tree->fanouts[k];
      a.gdiv[k] = (uint32_t)per_group;
    }
  }
  const int64_t E = T - b;              // edge occurrences
  const int64_t max_rows = a.off[hops - 1];  // nodes that are the parent of some slot
  // open-addressing tables at load factor <= 2/3: one node sub-table and one edge sub-table per batch, so the
  // probes of the batches in flight stay inside a few MB instead of scattering over one table of all batches
  uint64_t cap = 1024, ecap = 1024;
  // leaf-global mode: the table holds the nodes of level < hops only.  The plan's activation buffers hold
  // b*(1 + f0 + ...) of them (pipeline.hip guard_levels_kernel fails a batch with more), so the sub-table is sized
  // for twice that — a legitimate batch loads it <= 50 % — and a batch that fills it is reported through
  // meta[GIGL_META_OVERFLOW] by the insert kernels (its surplus occurrences are treated as leaves).
  int64_t inner = b / n_groups, width = b / n_groups;
  for (int k = 0; k + 1 < hops; ++k) {
    width *= tree->fanouts[k];
    inner += width;
  }
  const uint64_t want = a.leaf_global ? (uint64_t)inner * 4 : (uint64_t)(T / n_groups) * 3;
  while (cap * 2 < want) cap <<= 1;
  while (ecap * 2 < (uint64_t)(E / n_groups) * 3) ecap <<= 1;
  a.mask = (uint32_t)(cap - 1);
  const int64_t n_slots = (int64_t)cap * n_groups, n_ekeys = (int64_t)ecap * n_groups;
  GIGL_REQUIRE(ctx, n_slots < (int64_t)1 << 31, "batch too large: %lld table slots", (long long)n_slots);
  const int32_t n_tiles = (int32_t)((T + TILE - 1) / TILE);

  // scratch.  Everything that needs an initial pattern is initialised by ONE kernel:
  // [ekeys] = 0xFF.., [slots] = empty, [rowcnt | big_count] = 0
  int64_t need = 0;
  auto add = [&](int64_t bytes) { need += gigl_align_up(bytes, 256); };
  const int64_t zero_words = cap_nodes + 1 + 64;
  add(n_ekeys * 8);
  add(n_slots * (int64_t)sizeof(Slot));
  add(zero_words * 4);               // rowcnt + big-row counter
  add(T * 4);                        // slot_of
  add((int64_t)(n_tiles + 1) * MAXL * 4);  // tile counts + totals
  add(cap_nodes * 4);                // big-row queue
  add(E + 256);                      // winner flags
  add((E + 1) * 8);                  // winners' (dst, src) pairs
  int32_t rc = gigl_arena_reset(ctx, need + 4096);
  if (rc != GIGL_OK) return rc;
  unsigned long long* ekeys = (unsigned long long*)gigl_arena_alloc(ctx, n_ekeys * 8);
  a.slots = (Slot*)gigl_arena_alloc(ctx, n_slots * (int64_t)sizeof(Slot));
  int32_t* zeros = (int32_t*)gigl_arena_alloc(ctx, zero_words * 4);
  a.slot_of = (int32_t*)gigl_arena_alloc(ctx, T * 4);
  int32_t* tile_counts = (int32_t*)gigl_arena_alloc(ctx, (int64_t)(n_tiles + 1) * MAXL * 4);
  int32_t* big_rows = (int32_t*)gigl_arena_alloc(ctx, cap_nodes * 4);
  uint8_t* winner = (uint8_t*)gigl_arena_alloc(ctx, E + 256);
  int2* pairs = (int2*)gigl_arena_alloc(ctx, (E + 1) * 8);
  if (!ekeys || !a.slots || !zeros || !big_rows || !winner || !a.slot_of || !pairs || !tile_counts)
    return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  a.root_parent = winner;  // (free until edge_dedup_count writes the winner flags)
  a.alias_base = -1;
  a.cnt_last = tree->cnt[hops - 1];
  if (a.leaf_global && hops == 2 && tree->nbr[1] == (const uint32_t*)(out->col + out->cap_edges) &&
      out->cap_edges + (a.off[2] - a.off[1]) < ((int64_t)1 << 31))
    a.alias_base = (int32_t)out->cap_edges;
  a.overflow = out->meta + GIGL_META_OVERFLOW;
  int32_t* rowcnt = zeros;
  int32_t* big_count = zeros + cap_nodes + 1;  // [0] = number of queued rows

  const int TB = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); };
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_INSERT);
    // one launch initialises every table of this batch (instead of six memset dispatches)
    hipLaunchKernelGGL(init_scratch_kernel, dim3(1024), dim3(256), 0, st, (uint4*)ekeys, n_ekeys / 2, (uint4*)a.slots,
                       n_slots, (uint32_t)hops, zeros, zero_words, out->meta);
    hipLaunchKernelGGL(insert_roots_kernel, grid(b), dim3(TB), 0, st, a);
    if (E > 0) hipLaunchKernelGGL(insert_slots_kernel, grid(a.off[1] - b), dim3(TB), 0, st, a, (int64_t)b, a.off[1]);
    if (hops > 1 && T > a.off[1])
      hipLaunchKernelGGL(insert_slots_kernel, grid(T - a.off[1]), dim3(TB), 0, st, a, a.off[1], T);
  }
  if (hops > 2 && E > 0) {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_RELAX);
    for (int r = 0; r < hops; ++r) hipLaunchKernelGGL(relax_kernel, grid(E), dim3(TB), 0, st, a);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_NODES);
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, a, tile_counts);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, tile_counts, n_tiles);
    hipLaunchKernelGGL(assign_kernel, dim3((unsigned)n_tiles), dim3(256), 0, st, a, tile_counts, n_tiles,
                       out->nodes, out->meta, out->rowptr, out->rowend, rowcnt, big_count + 32);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_EDGE_SORT);
    // every node of level < hops may be a row (also leaf-only ones reached under a root parent)
    hipLaunchKernelGGL(edge_dedup_count_kernel, grid(T), dim3(TB), 0, st, a, ekeys, (uint32_t)(ecap - 1), winner,
                       pairs, rowcnt, out->root_local);
    hipLaunchKernelGGL(row_scan_kernel, dim3(SCAN_BLOCKS), dim3(1024), 0, st, rowcnt, out->meta, hops,
                       out->cap_nodes, out->rowptr, out->rowend, E > 0 ? 0 : 1, big_count + 8);
    if (E > 0)
      hipLaunchKernelGGL(edge_fill_kernel, grid(E), dim3(TB), 0, st, a, winner, pairs, out->meta, out->rowptr,
                         out->rowend, out->col);
  }
  if (E > 0) {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_CSR);
    // a row costs a chain of dependent loads and a few instructions: latency-bound, so a wave per (possible) row
    int64_t blocks = (max_rows + 3) / 4;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(row_sort_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out->meta, hops, out->rowptr,
                       out->rowend, out->col, big_rows, big_count, a.alias_base, out->meta);
    GIGL_HIP_CHECK(ctx, lds_opt_in(ctx->device, 2, (const void*)row_sort_big_kernel, 2 * BIG_ROW_CAP * (int)sizeof(int32_t)));
    hipLaunchKernelGGL(row_sort_big_kernel, dim3(256), dim3(1024), 2 * BIG_ROW_CAP * sizeof(int32_t), st,
                       out->rowptr, out->rowend, out->col, big_rows, big_count, out->meta + GIGL_META_OVERFLOW);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}
