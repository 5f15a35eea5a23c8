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

namespace {

constexpr int MAXL = GIGL_MAX_HOPS + 1;
constexpr int TILE = 1024;          // stream positions per count/assign workgroup
constexpr int BIG_ROW_CAP = 16384;  // LDS bitonic capacity (64 KiB of int32)

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
    if (m > BIG_ROW_CAP) {  // does not fit the LDS sort: left unsorted, reported through meta
      if (tid == 0) atomicAdd(overflow, 1);
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

  UnionArgs a{};
  a.roots = roots;
  a.b = b;
  a.hops = hops;
  a.leaf_global = leaf_global ? 1 : 0;
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
    static bool lds_attr_set = false;  // 128 KiB of dynamic LDS needs the opt-in once per process
    if (!lds_attr_set) {
      GIGL_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)row_sort_big_kernel,
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              2 * BIG_ROW_CAP * (int)sizeof(int32_t)));
      lds_attr_set = true;
    }
    hipLaunchKernelGGL(row_sort_big_kernel, dim3(256), dim3(1024), 2 * BIG_ROW_CAP * sizeof(int32_t), st,
                       out->rowptr, out->rowend, out->col, big_rows, big_count, out->meta + GIGL_META_OVERFLOW);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}
