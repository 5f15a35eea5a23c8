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
// Pipeline (T = b + sum slots[k] stream positions; roots first, then hop slots in order):
//   1 insert      open-addressing hash table keyed by node id (atomicCAS), atomicMin of the first
//                 stream position; roots get level 0
//   2 relax xhops level(src) = min(level(src), level(dst)+1) over every sampled edge occurrence
//   3 flag        first occurrences -> per-level one-hot counters; exclusive scan (rocPRIM)
//   4 assign      local id = level base + rank within level; write nodes[], root_local[]
//   5 edges       (dst_local << 32 | src_local) keys; radix sort (rocPRIM) ; unique flags ; scan
//   6 csr         col[] from unique keys; rowptr[] by binary search over the unique keys
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace {

constexpr int MAXL = GIGL_MAX_HOPS + 1;
constexpr int32_t LVL_INF = 1 << 20;

struct LevelCount {
  int32_t c[MAXL];
  LevelCount() = default;
  __host__ __device__ LevelCount(int v) {
#pragma unroll
    for (int i = 0; i < MAXL; ++i) c[i] = v;
  }
  __host__ __device__ LevelCount operator+(const LevelCount& o) const {
    LevelCount r;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) r.c[i] = c[i] + o.c[i];
    return r;
  }
};

struct UnionArgs {
  const uint32_t* roots;
  int32_t b;
  int32_t hops;
  const uint32_t* nbr[GIGL_MAX_HOPS];
  int32_t fan[GIGL_MAX_HOPS];
  int64_t off[GIGL_MAX_HOPS + 1];  // stream offset of hop k slots; off[hops] = T
  int64_t T;
  // hash table
  uint32_t* keys;
  uint32_t* firstpos;
  int32_t* level;
  int32_t* lid;
  uint32_t mask;
  // per stream position
  int32_t* slot_of;  // table slot or -1
};

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// stream position -> (node id, hop k or -1 for roots, slot index j within the hop)
__device__ __forceinline__ uint32_t stream_at(const UnionArgs& a, int64_t t, int& k, int64_t& j) {
  if (t < a.b) {
    k = -1;
    j = t;
    return a.roots[t];
  }
  int kk = 0;
#pragma unroll
  for (int i = 1; i < GIGL_MAX_HOPS; ++i)
    if (i < a.hops && t >= a.off[i]) kk = i;
  k = kk;
  j = t - a.off[kk];
  return a.nbr[kk][j];
}

// stream position of the destination (parent) of the occurrence at hop k, slot j
__device__ __forceinline__ int64_t parent_pos(const UnionArgs& a, int k, int64_t j) {
  int64_t p = j / a.fan[k];
  return k == 0 ? p : a.off[k - 1] + p;
}

__global__ void insert_kernel(UnionArgs a) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.T) return;
  int k;
  int64_t j;
  uint32_t id = stream_at(a, t, k, j);
  if (id == GIGL_INVALID) {
    a.slot_of[t] = -1;
    return;
  }
  uint32_t s = hash_u32(id) & a.mask;
  while (true) {
    uint32_t prev = atomicCAS(&a.keys[s], GIGL_INVALID, id);
    if (prev == GIGL_INVALID || prev == id) break;
    s = (s + 1) & a.mask;
  }
  atomicMin(&a.firstpos[s], (uint32_t)t);
  if (k < 0) atomicMin(&a.level[s], 0);
  a.slot_of[t] = (int32_t)s;
}

__global__ void relax_kernel(UnionArgs a) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + a.b;
  if (t >= a.T) return;
  int32_t s = a.slot_of[t];
  if (s < 0) return;
  int k;
  int64_t j;
  stream_at(a, t, k, j);
  int32_t ds = a.slot_of[parent_pos(a, k, j)];
  // level[] is only ever lowered; a stale read only delays convergence (hops rounds suffice)
  int32_t dl = __hip_atomic_load(&a.level[ds], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (dl + 1 < LVL_INF) atomicMin(&a.level[s], dl + 1);
}

__global__ void flag_kernel(UnionArgs a, LevelCount* flags) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.T) return;
  LevelCount f;
#pragma unroll
  for (int i = 0; i < MAXL; ++i) f.c[i] = 0;
  int32_t s = a.slot_of[t];
  if (s >= 0 && a.firstpos[s] == (uint32_t)t) {
    int32_t l = a.level[s];
    if (l >= 0 && l < MAXL) f.c[l] = 1;
  }
  flags[t] = f;
}

// meta from the scan tail: totals per level -> cumulative
__global__ void meta_kernel(UnionArgs a, const LevelCount* flags, const LevelCount* scan,
                            int32_t* meta, int32_t* level_base) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  LevelCount tot = scan[a.T - 1] + flags[a.T - 1];
  int32_t cum = 0;
  for (int l = 0; l < MAXL; ++l) {
    level_base[l] = cum;
    cum += tot.c[l];
    if (l <= a.hops) meta[GIGL_META_LEVEL0 + l] = cum;
  }
  meta[GIGL_META_N_NODES] = cum;
}

__global__ void assign_kernel(UnionArgs a, const LevelCount* flags, const LevelCount* scan,
                              const int32_t* level_base, uint32_t* nodes) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.T) return;
  int32_t s = a.slot_of[t];
  if (s < 0 || a.firstpos[s] != (uint32_t)t) return;
  int32_t l = a.level[s];
  int32_t id = level_base[l] + scan[t].c[l];
  a.lid[s] = id;
  nodes[id] = a.keys[s];
}

__global__ void root_local_kernel(UnionArgs a, int32_t* root_local) {
  int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.b) return;
  int32_t s = a.slot_of[i];
  root_local[i] = s >= 0 ? a.lid[s] : -1;
}

__global__ void edge_key_kernel(UnionArgs a, uint64_t* ekeys) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t t = e + a.b;
  if (t >= a.T) return;
  int32_t s = a.slot_of[t];
  uint64_t key = ~0ULL;
  if (s >= 0) {
    int k;
    int64_t j;
    stream_at(a, t, k, j);
    int32_t ds = a.slot_of[parent_pos(a, k, j)];
    key = ((uint64_t)(uint32_t)a.lid[ds] << 32) | (uint32_t)a.lid[s];
  }
  ekeys[e] = key;
}

__global__ void edge_flag_kernel(const uint64_t* sorted, int64_t n, int32_t* flags) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  uint64_t k = sorted[e];
  flags[e] = (k != ~0ULL && (e == 0 || sorted[e - 1] != k)) ? 1 : 0;
}

__global__ void edge_write_kernel(const uint64_t* sorted, const int32_t* flags, const int32_t* scan,
                                  int64_t n, int32_t* col, uint64_t* ukeys, int32_t* meta) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (flags[e]) {
    int32_t r = scan[e];
    col[r] = (int32_t)(sorted[e] & 0xFFFFFFFFu);
    ukeys[r] = sorted[e];
  }
  if (e == n - 1) meta[GIGL_META_N_EDGES] = scan[e] + flags[e];
}

// rowptr[i] = number of unique edges with dst < i  (lower_bound of i<<32 in the unique sorted keys)
__global__ void rowptr_kernel(const uint64_t* ukeys, const int32_t* meta, int32_t* rowptr,
                              int64_t cap_nodes) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int32_t nn = meta[GIGL_META_N_NODES];
  if (i > nn || i > cap_nodes) return;
  int32_t ne = meta[GIGL_META_N_EDGES];
  uint64_t target = (uint64_t)i << 32;
  int32_t lo = 0, hi = ne;
  while (lo < hi) {
    int32_t mid = (lo + hi) >> 1;
    if (ukeys[mid] < target) lo = mid + 1;
    else hi = mid;
  }
  rowptr[i] = lo;
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
  GIGL_REQUIRE(ctx, tree && out && (roots || tree->b == 0), "null argument");
  GIGL_REQUIRE(ctx, out->meta && out->nodes && out->rowptr && out->col && out->root_local,
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
  GIGL_HIP_CHECK(ctx, hipMemsetAsync(out->meta, 0, GIGL_META_LEN * sizeof(int32_t), st));
  if (b == 0) {
    GIGL_HIP_CHECK(ctx, hipMemsetAsync(out->rowptr, 0, sizeof(int32_t), st));
    return GIGL_OK;
  }

  UnionArgs a{};
  a.roots = roots;
  a.b = b;
  a.hops = hops;
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
  const int64_t E = T - b;  // edge occurrences
  uint64_t cap = 1024;
  while (cap < (uint64_t)T * 2) cap <<= 1;
  a.mask = (uint32_t)(cap - 1);

  // temp storage sizes for rocPRIM calls
  size_t tmp_scan_lc = 0, tmp_scan_i = 0, tmp_sort = 0;
  hipcub::DeviceScan::ExclusiveSum((void*)nullptr, tmp_scan_lc, (LevelCount*)nullptr,
                                   (LevelCount*)nullptr, (int)T, st);
  hipcub::DeviceScan::ExclusiveSum((void*)nullptr, tmp_scan_i, (int32_t*)nullptr, (int32_t*)nullptr,
                                   (int)(E > 0 ? E : 1), st);
  int key_bits = 32;
  while ((1LL << (key_bits - 32)) < T + 1) ++key_bits;  // dst_local < T
  hipcub::DeviceRadixSort::SortKeys((void*)nullptr, tmp_sort, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                    (int)(E > 0 ? E : 1), 0, key_bits, st);
  size_t tmp_bytes = tmp_scan_lc > tmp_scan_i ? tmp_scan_lc : tmp_scan_i;
  if (tmp_sort > tmp_bytes) tmp_bytes = tmp_sort;

  int64_t need = 0;
  auto add = [&](int64_t bytes) { need += gigl_align_up(bytes, 256); };
  add(cap * 4); add(cap * 4); add(cap * 4); add(cap * 4);      // table
  add(T * 4);                                                  // slot_of
  add(T * sizeof(LevelCount)); add(T * sizeof(LevelCount));    // flags, scan
  add(256);                                                    // level_base
  add(E * 8); add(E * 8); add(E * 8);                          // ekeys, sorted, ukeys
  add(E * 4); add(E * 4);                                      // eflags, escan
  add((int64_t)tmp_bytes);
  int32_t rc = gigl_arena_reset(ctx, need + 4096);
  if (rc != GIGL_OK) return rc;
  a.keys = (uint32_t*)gigl_arena_alloc(ctx, cap * 4);
  a.firstpos = (uint32_t*)gigl_arena_alloc(ctx, cap * 4);
  a.level = (int32_t*)gigl_arena_alloc(ctx, cap * 4);
  a.lid = (int32_t*)gigl_arena_alloc(ctx, cap * 4);
  a.slot_of = (int32_t*)gigl_arena_alloc(ctx, T * 4);
  LevelCount* flags = (LevelCount*)gigl_arena_alloc(ctx, T * sizeof(LevelCount));
  LevelCount* scan = (LevelCount*)gigl_arena_alloc(ctx, T * sizeof(LevelCount));
  int32_t* level_base = (int32_t*)gigl_arena_alloc(ctx, 256);
  uint64_t* ekeys = (uint64_t*)gigl_arena_alloc(ctx, E * 8);
  uint64_t* sorted = (uint64_t*)gigl_arena_alloc(ctx, E * 8);
  uint64_t* ukeys = (uint64_t*)gigl_arena_alloc(ctx, E * 8);
  int32_t* eflags = (int32_t*)gigl_arena_alloc(ctx, E * 4);
  int32_t* escan = (int32_t*)gigl_arena_alloc(ctx, E * 4);
  void* tmp = gigl_arena_alloc(ctx, (int64_t)tmp_bytes);
  if (!tmp || !escan) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");

  const int TB = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + TB - 1) / TB)); };
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_INSERT);
    GIGL_HIP_CHECK(ctx, hipMemsetAsync(a.keys, 0xFF, cap * 4, st));
    GIGL_HIP_CHECK(ctx, hipMemsetAsync(a.firstpos, 0xFF, cap * 4, st));
    // LVL_INF = 0x00100000: bytes are not uniform, so fill via 0x7F (0x7F7F7F7F > any real level)
    GIGL_HIP_CHECK(ctx, hipMemsetAsync(a.level, 0x7F, cap * 4, st));
    hipLaunchKernelGGL(insert_kernel, grid(T), dim3(TB), 0, st, a);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_RELAX);
    for (int r = 0; r < hops; ++r) hipLaunchKernelGGL(relax_kernel, grid(E), dim3(TB), 0, st, a);
  }
  {
    gigl_prof_scope ps(ctx, GIGL_K_UNION_NODES);
    hipLaunchKernelGGL(flag_kernel, grid(T), dim3(TB), 0, st, a, flags);
    GIGL_HIP_CHECK(ctx, hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flags, scan, (int)T, st));
    hipLaunchKernelGGL(meta_kernel, dim3(1), dim3(64), 0, st, a, flags, scan, out->meta, level_base);
    hipLaunchKernelGGL(assign_kernel, grid(T), dim3(TB), 0, st, a, flags, scan, level_base, out->nodes);
    hipLaunchKernelGGL(root_local_kernel, grid(b), dim3(TB), 0, st, a, out->root_local);
  }
  if (E > 0) {
    {
      gigl_prof_scope ps(ctx, GIGL_K_UNION_EDGE_SORT);
      hipLaunchKernelGGL(edge_key_kernel, grid(E), dim3(TB), 0, st, a, ekeys);
      GIGL_HIP_CHECK(ctx, hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, ekeys, sorted, (int)E, 0,
                                                            key_bits, st));
    }
    gigl_prof_scope ps(ctx, GIGL_K_UNION_CSR);
    hipLaunchKernelGGL(edge_flag_kernel, grid(E), dim3(TB), 0, st, sorted, E, eflags);
    GIGL_HIP_CHECK(ctx, hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, eflags, escan, (int)E, st));
    hipLaunchKernelGGL(edge_write_kernel, grid(E), dim3(TB), 0, st, sorted, eflags, escan, E, out->col,
                       ukeys, out->meta);
    hipLaunchKernelGGL(rowptr_kernel, grid(cap_nodes + 1), dim3(TB), 0, st, ukeys, out->meta,
                       out->rowptr, out->cap_nodes);
  } else {
    hipLaunchKernelGGL(rowptr_kernel, grid(cap_nodes + 1), dim3(TB), 0, st, ukeys, out->meta,
                       out->rowptr, out->cap_nodes);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
