// typed_plan.hip — the typed (heterogeneous) batch graph of a SamplingOp DAG in ONE host call.
//
// Replaces, for a batch of roots on a typed graph:
//   GraphDBSampler.getKHopSubgraphForRootNode      scala_spark35/subgraph_sampler/src/main/scala/libs/sampler/
//                                                   GraphDBSampler.scala:40-148 (per root: every op's frontier = the
//                                                   set union of its parents' node sets; an op runs for a root only
//                                                   when all its parents ran and its frontier is not empty)
//   SamplingOpDAG                                   scala_spark35/common/src/main/scala/types/SamplingOpDAG.scala:19-53
//   the trainer-side collate of the typed samples   python/gigl/src/common/graph_builder/abstract_graph_builder.py:49-150,
//                                                   pyg_graph_builder.py:20-69 (per node type the distinct nodes, per
//                                                   edge type the distinct edges, as local ids)
// What gigl_amd/graphdb_sampler.py::batch_graph did with one library call per op and a chain of torch.unique /
// searchsorted calls (each a host synchronisation: their output sizes are data) is one stream of device work here:
//   per op      frontier = parents' results side by side (dag_frontier_kernel) -> gigl_rows_dedup -> ran / mask / path
//               sums (dag_mask_kernel) -> gigl_expand_frontier on the op's (edge type, direction) graph
//   per type    every id the ops produced for the type -> radix sort -> distinct, ascending = the type's local
//               numbering (the numbering of batch_graph: torch.unique)
//   per edge    endpoints -> local ids by binary search -> (src << 32 | dst) keys -> radix sort -> distinct
//   roots       position of every root in its type's node list
// Counts stay on the device (n_nodes[type], n_edges[edge slot]); the caller reads them once per batch.
// Integer work; bound by the sorts (the library's own LSD radix sort over ~b * sum(w * f) keys per type, sortscan.h:
// kernels only, so that a plan can be captured into a hipGraph) and the sampler's dependent loads, not by bandwidth.
#include "common.h"
#include "sortscan.h"

#include <vector>

struct gigl_typed_plan {
  gigl_ctx* ctx = nullptr;
  int32_t n_ops = 0, n_types = 0, n_slots = 0, root_type = 0, b_max = 0;
  std::vector<gigl_dag_op> ops;
  std::vector<int32_t> width;          // frontier slots per root of every op
  std::vector<int64_t> window_end;     // per op: largest hash-window end any of its rows can reach (-1: unbounded)
  // device, per op: frontier [b][w], path sums [b][w], neighbours [b][w][f], counts [b][w], ran [b]
  std::vector<uint32_t*> front, ksum, nbr;
  std::vector<int32_t*> cnt;
  std::vector<uint8_t*> ran;
  // per node type: candidates / sorted (capacity cand_cap[t] per b_max), distinct ids, count
  std::vector<int64_t> cand_per_root;  // candidate ids per root of the type
  std::vector<uint32_t*> cand, sorted, nodes;
  // per edge slot: keys / sorted keys, distinct (src_local << 32 | dst_local)
  std::vector<int64_t> pairs_per_root;
  std::vector<int32_t> slot_src_type, slot_dst_type;
  std::vector<unsigned long long*> ekeys, esorted, edges;
  int32_t* n_nodes = nullptr;   // [n_types]
  int32_t* n_edges = nullptr;   // [n_slots]
  int32_t* root_index = nullptr;  // [b_max]
  int32_t* scratch = nullptr;        // gigl_sort::scratch_words(max items): digit histograms / tile counts
  uint32_t* tmp32 = nullptr;         // [max items] ping-pong buffers of the sorts
  unsigned long long* tmp64 = nullptr;
  int64_t tmp32_words = 0, tmp64_words = 0;
  std::vector<int32_t> id_bits;      // per node type: bits an id of the type can occupy (32: unknown)
  int64_t max_items = 0;
  // merged CSR by destination over all edge slots (gigl_typed_plan_merged_csr; allocated on its first call)
  int64_t m_edges_cap = 0, m_rows_cap = 0;
  unsigned long long *m_keys = nullptr, *m_sorted = nullptr;
  int32_t *m_rowptr = nullptr, *m_col = nullptr, *m_etype = nullptr, *m_counts = nullptr;
  int32_t *m_root_len = nullptr, *m_root_rowptr = nullptr, *m_root_col = nullptr, *m_root_etype = nullptr;
  struct MergedOffsets* m_off = nullptr;
  unsigned long long* m_tmp = nullptr;
  int32_t* m_scratch = nullptr;
  std::vector<void*> owned;
};

namespace {

constexpr int TB = 256;
inline dim3 grid_of(int64_t n) { return dim3((unsigned)((n + TB - 1) / TB > 0 ? (n + TB - 1) / TB : 1)); }

struct FrontierSrc {
  const uint32_t* nbr[GIGL_DAG_MAX_PARENTS];
  int32_t len[GIGL_DAG_MAX_PARENTS];  // w_p * f_p
  int32_t n;
};

// front[r][off_p + j] = parent p's j-th result for root r (parents side by side); root ops: front[r][0] = roots[r]
__global__ __launch_bounds__(TB) void dag_frontier_kernel(FrontierSrc s, const uint32_t* roots, int64_t b, int32_t w,
                                                          uint32_t* front) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * w) return;
  const int64_t r = i / w;
  int32_t j = (int32_t)(i - r * w);
  if (s.n == 0) {
    front[i] = roots[r];
    return;
  }
  for (int p = 0; p < s.n; ++p) {
    if (j < s.len[p]) {
      front[i] = s.nbr[p][r * s.len[p] + j];
      return;
    }
    j -= s.len[p];
  }
}

struct RanSrc {
  const uint8_t* ran[GIGL_DAG_MAX_PARENTS];
  int32_t n;
};

// one wave per root: the op runs for the root when every parent ran and the (deduplicated) frontier is not empty
// (GraphDBSampler.scala:66-86); otherwise its row is emptied.  ksum = frontier id + root (uint32 wrap == the sampler's
// int32 add)
__global__ __launch_bounds__(TB) void dag_mask_kernel(RanSrc s, const uint32_t* roots, int64_t b, int32_t w,
                                                      uint32_t* front, uint32_t* ksum, uint8_t* ran) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  if (r >= b) return;
  bool any = false;
  for (int32_t j = lane; j < w; j += 64) any |= front[r * w + j] != GIGL_INVALID;
  bool ok = __ballot(any) != 0;
  for (int p = 0; p < s.n; ++p) ok = ok && s.ran[p][r] != 0;
  if (lane == 0) ran[r] = ok ? 1 : 0;
  const uint32_t root = roots[r];
  for (int32_t j = lane; j < w; j += 64) {
    const uint32_t v = ok ? front[r * w + j] : GIGL_INVALID;
    front[r * w + j] = v;
    ksum[r * w + j] = v + root;
  }
}

// the ids an op contributes to its two node types: a frontier id where it has at least one sampled neighbour, every
// sampled neighbour
__global__ __launch_bounds__(TB) void cand_frontier_kernel(const uint32_t* front, const int32_t* cnt, int64_t m,
                                                           uint32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) out[i] = cnt[i] > 0 ? front[i] : GIGL_INVALID;
}
__global__ __launch_bounds__(TB) void copy_u32_kernel(const uint32_t* src, int64_t m, uint32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) out[i] = src[i];
}
// position of v in the ascending list a[0 .. n) (v is in it)
__device__ __forceinline__ uint32_t lower_bound(const uint32_t* a, int32_t n, uint32_t v) {
  int32_t lo = 0, hi = n;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return (uint32_t)lo;
}

// an op's sampled edges as (src_local << 32 | dst_local) keys of its edge slot
__global__ __launch_bounds__(TB) void edge_keys_kernel(const uint32_t* front, const uint32_t* nbr, int64_t m, int32_t f,
                                                       int32_t outgoing, const uint32_t* nodes_front,
                                                       const int32_t* n_front, const uint32_t* nodes_got,
                                                       const int32_t* n_got, unsigned long long* keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * f) return;
  const uint32_t fr = front[i / f], nb = nbr[i];
  unsigned long long k = ~0ull;
  if (fr != GIGL_INVALID && nb != GIGL_INVALID) {
    const unsigned long long lf = lower_bound(nodes_front, *n_front, fr), lg = lower_bound(nodes_got, *n_got, nb);
    k = outgoing ? (lf << 32) | lg : (lg << 32) | lf;
  }
  keys[i] = k;
}

__global__ __launch_bounds__(TB) void root_index_kernel(const uint32_t* roots, int64_t b, const uint32_t* nodes,
                                                        const int32_t* n, int32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b) out[i] = (int32_t)lower_bound(nodes, *n, roots[i]);
}

// ---- merged CSR by destination (the operand of the typed attention layers) ------------------------------------------
// What gigl_amd/models_hetero.py::HGTConv built per batch with a chain of torch ops (three concatenations, a stable sort
// by destination, bincount + cumsum, three gathers; for the roots' rows another cumsum / repeat_interleave chain with a
// host read of its size): destinations numbered type after type in the caller's type order, sources slot after slot
// in the caller's slot order (source index = offset of the slot + src_local, as the layer concatenates its per-edge-type
// K / V blocks), edges of one destination in slot order then (src, dst) order = torch.sort(dst, stable=True).
struct MergedArgs {
  int32_t n_types, n_slots, root_type;
  int32_t type_order[16];
  int32_t slot_order[32], slot_src_type[32], slot_dst_type[32], slot_etype[32];
  int64_t cap_off[33];  // capacity prefix of the listed slots (positions of merged_keys_kernel's grid)
  const unsigned long long* edges[32];
  // capacity layout (gigl_typed_plan_merged_csr_ex): destination / source blocks start at CAPACITY prefixes — known when
  // the plan is made — instead of at the batch's counts: every buffer address of a layer over this CSR is then static
  int32_t capacity_layout;
  int32_t type_cap[16];  // by position in type_order
  int32_t src_cap[32];   // by position in slot_order: capacity of the slot's source type
};
}  // namespace
struct MergedOffsets {
  int32_t dst_off[16];   // by node type id (-1: the type is not listed)
  int32_t src_off[32];   // by position in slot_order
  int32_t edge_off[33];  // by position in slot_order; [n_slots] = E
  int32_t n_dst, n_edges;
};
namespace {

__global__ void merged_offsets_kernel(MergedArgs a, const int32_t* n_nodes, const int32_t* n_edges, MergedOffsets* off,
                                      int32_t* counts) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int32_t run = 0;
  for (int t = 0; t < 16; ++t) off->dst_off[t] = -1;
  for (int j = 0; j < a.n_types; ++j) {
    off->dst_off[a.type_order[j]] = run;
    run += a.capacity_layout ? a.type_cap[j] : n_nodes[a.type_order[j]];
  }
  off->n_dst = run;
  int32_t srun = 0, erun = 0;
  for (int j = 0; j < a.n_slots; ++j) {
    off->src_off[j] = srun;
    off->edge_off[j] = erun;
    srun += a.capacity_layout ? a.src_cap[j] : n_nodes[a.slot_src_type[j]];
    erun += n_edges[a.slot_order[j]];
  }
  off->edge_off[a.n_slots] = erun;
  off->n_edges = erun;
  counts[0] = run;
  counts[1] = erun;
}

__global__ __launch_bounds__(TB) void merged_keys_kernel(MergedArgs a, const int32_t* n_edges, const MergedOffsets* off,
                                                         int64_t m, unsigned long long* keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  int j = 0;
  while (j + 1 < a.n_slots && i >= a.cap_off[j + 1]) ++j;
  const int64_t q = i - a.cap_off[j];
  unsigned long long k = ~0ull;
  if (q < n_edges[a.slot_order[j]]) {
    const unsigned long long e = a.edges[j][q];
    const unsigned long long dst = (unsigned long long)(uint32_t)off->dst_off[a.slot_dst_type[j]] + (e & 0xFFFFFFFFull);
    k = (dst << 32) | (uint32_t)(off->edge_off[j] + (int32_t)q);
  }
  keys[i] = k;
}

__global__ __launch_bounds__(TB) void merged_fill_kernel(MergedArgs a, const MergedOffsets* off,
                                                         const unsigned long long* sorted, int64_t m, int32_t* col,
                                                         int32_t* etype) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || i >= off->n_edges) return;
  const int32_t pos = (int32_t)(uint32_t)sorted[i];
  int j = 0;
  while (j + 1 < a.n_slots && pos >= off->edge_off[j + 1]) ++j;
  const unsigned long long e = a.edges[j][pos - off->edge_off[j]];
  col[i] = off->src_off[j] + (int32_t)(e >> 32);
  etype[i] = a.slot_etype[j];
}

__global__ __launch_bounds__(TB) void merged_rowptr_kernel(const MergedOffsets* off, const unsigned long long* sorted,
                                                           int64_t rows_cap, int32_t* rowptr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > rows_cap || r > off->n_dst) return;
  const unsigned long long want = (unsigned long long)r << 32;
  int32_t lo = 0, hi = off->n_edges;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (sorted[mid] < want) lo = mid + 1;
    else hi = mid;
  }
  rowptr[r] = lo;
}

// the roots' rows alone, in root order: lengths -> exclusive sum -> one wave per root copies its slice
__global__ __launch_bounds__(TB) void root_len_kernel(const MergedOffsets* off, int32_t root_type, const int32_t* root_index,
                                                      int32_t b, const int32_t* rowptr, int32_t* len) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > b) return;
  int32_t v = 0;
  if (i < b) {
    const int32_t row = off->dst_off[root_type] + root_index[i];
    v = rowptr[row + 1] - rowptr[row];
  }
  len[i] = v;
}

__global__ __launch_bounds__(TB) void root_rows_kernel(const MergedOffsets* off, int32_t root_type, const int32_t* root_index,
                                                       int32_t b, const int32_t* rowptr, const int32_t* col,
                                                       const int32_t* etype, const int32_t* root_rowptr, int32_t* root_col,
                                                       int32_t* root_etype) {
  const int32_t i = (int32_t)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (i >= b) return;
  const int32_t row = off->dst_off[root_type] + root_index[i];
  const int32_t lo = rowptr[row], n = rowptr[row + 1] - lo, out = root_rowptr[i];
  for (int32_t e = lane; e < n; e += 64) {
    root_col[out + e] = col[lo + e];
    root_etype[out + e] = etype[lo + e];
  }
}

template <typename T>
int32_t dev_alloc(gigl_typed_plan* p, T** out, int64_t count) {
  void* q = nullptr;
  if (hipMalloc(&q, (size_t)(count > 0 ? count : 1) * sizeof(T)) != hipSuccess)
    return gigl_fail(p->ctx, GIGL_E_OOM, "typed plan: hipMalloc of %lld bytes failed", (long long)(count * sizeof(T)));
  p->owned.push_back(q);
  *out = (T*)q;
  return GIGL_OK;
}

int key_bits(int64_t cap) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < cap) ++bits;
  return bits;
}

}  // namespace

extern "C" {

int32_t gigl_typed_plan_destroy(gigl_typed_plan* p) {
  if (!p) return GIGL_OK;
  if (p->ctx) {
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
  }
  for (void* q : p->owned) hipFree(q);
  delete p;
  return GIGL_OK;
}

int32_t gigl_typed_plan_create(gigl_ctx* ctx, const gigl_dag_op* ops, int32_t n_ops, int32_t n_node_types,
                               int32_t root_node_type, int32_t n_edge_slots, int32_t b_max, gigl_typed_plan** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, ops && n_ops >= 1 && n_ops <= GIGL_DAG_MAX_OPS, "between 1 and %d sampling ops", GIGL_DAG_MAX_OPS);
  GIGL_REQUIRE(ctx, n_node_types >= 1 && n_node_types <= 16 && root_node_type >= 0 && root_node_type < n_node_types,
               "node types outside [1,16]");
  GIGL_REQUIRE(ctx, n_edge_slots >= 1 && n_edge_slots <= 32 && b_max >= 1, "bad edge slots / batch size");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_typed_plan* p = new (std::nothrow) gigl_typed_plan();
  if (!p) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  p->ctx = ctx;
  p->n_ops = n_ops;
  p->n_types = n_node_types;
  p->n_slots = n_edge_slots;
  p->root_type = root_node_type;
  p->b_max = b_max;
  p->ops.assign(ops, ops + n_ops);
  p->width.assign(n_ops, 0);
  p->cand_per_root.assign(n_node_types, 0);
  p->pairs_per_root.assign(n_edge_slots, 0);
  p->slot_src_type.assign(n_edge_slots, -1);
  p->slot_dst_type.assign(n_edge_slots, -1);
  p->cand_per_root[root_node_type] = 1;
  p->id_bits.assign(n_node_types, 0);
  int32_t rc = GIGL_OK;
#define PLAN_FAIL(...)                              \
  do {                                              \
    rc = gigl_fail(ctx, GIGL_E_INVALID_ARG, __VA_ARGS__); \
    gigl_typed_plan_destroy(p);                     \
    return rc;                                      \
  } while (0)
  for (int o = 0; o < n_ops; ++o) {
    const gigl_dag_op& op = p->ops[o];
    if (!op.graph || op.fanout < 1 || op.fanout > GIGL_MAX_FANOUT) PLAN_FAIL("op %d: no graph or fanout outside [1,%d]", o, GIGL_MAX_FANOUT);
    if (op.n_parents < 0 || op.n_parents > GIGL_DAG_MAX_PARENTS) PLAN_FAIL("op %d: more than %d parents", o, GIGL_DAG_MAX_PARENTS);
    if (op.frontier_node_type < 0 || op.frontier_node_type >= n_node_types || op.result_node_type < 0 ||
        op.result_node_type >= n_node_types || op.edge_slot < 0 || op.edge_slot >= n_edge_slots)
      PLAN_FAIL("op %d: a type outside the metadata", o);
    int64_t w = op.n_parents == 0 ? 1 : 0;
    for (int k = 0; k < op.n_parents; ++k) {
      const int pp = op.parents[k];
      if (pp < 0 || pp >= o) PLAN_FAIL("op %d: parent %d is not an earlier op", o, pp);
      w += (int64_t)p->width[pp] * p->ops[pp].fanout;
    }
    if (op.n_parents == 0 && op.frontier_node_type != root_node_type) PLAN_FAIL("op %d starts at the roots but not at their type", o);
    if (w > 8192) PLAN_FAIL("op %d: a frontier of %lld ids per root exceeds 8192", o, (long long)w);
    p->width[o] = (int32_t)w;
    p->cand_per_root[op.frontier_node_type] += w;
    p->cand_per_root[op.result_node_type] += w * op.fanout;
    p->pairs_per_root[op.edge_slot] += w * op.fanout;
    const int32_t st = op.outgoing ? op.frontier_node_type : op.result_node_type;
    const int32_t dt = op.outgoing ? op.result_node_type : op.frontier_node_type;
    if (p->slot_src_type[op.edge_slot] >= 0 && (p->slot_src_type[op.edge_slot] != st || p->slot_dst_type[op.edge_slot] != dt))
      PLAN_FAIL("op %d: edge slot %d joins two different node type pairs", o, op.edge_slot);
    p->slot_src_type[op.edge_slot] = st;
    p->slot_dst_type[op.edge_slot] = dt;
    // an op's graph has one row per node of its FRONTIER type: ids of that type lie below its row count
    const int fb = key_bits(op.graph->n + 1);
    p->id_bits[op.frontier_node_type] = fb > p->id_bits[op.frontier_node_type] ? fb : p->id_bits[op.frontier_node_type];
  }
  {  // every op's hash windows end below (frontier id + root id) + hash_add + the longest row: lets the sampler serve
     // all of them from the threshold table (sample.hip) instead of hashing rows whose window it cannot place
    int64_t n_root = 0;
    for (int o = 0; o < n_ops; ++o)
      if (p->ops[o].n_parents == 0 && p->ops[o].graph->n > n_root) n_root = p->ops[o].graph->n;
    p->window_end.assign(n_ops, -1);
    for (int o = 0; o < n_ops; ++o) {
      const gigl_dag_op& op = p->ops[o];
      if (op.hash_add < 0) continue;
      const int64_t end = (op.graph->n > 0 ? op.graph->n - 1 : 0) + (n_root > 0 ? n_root - 1 : 0) + (int64_t)op.hash_add +
                          op.graph->maxdeg;
      if (end < ((int64_t)1 << 32)) p->window_end[o] = end;
    }
  }
  for (int t = 0; t < n_node_types; ++t) {  // a type that is never a frontier: nothing bounds its ids here
    bool frontier = false;
    for (int o = 0; o < n_ops; ++o) frontier |= p->ops[o].frontier_node_type == t;
    if (!frontier) p->id_bits[t] = 32;
  }
#undef PLAN_FAIL
  const int64_t b = b_max;
  p->front.resize(n_ops);
  p->ksum.resize(n_ops);
  p->nbr.resize(n_ops);
  p->cnt.resize(n_ops);
  p->ran.resize(n_ops);
#define PLAN_ALLOC(ptr, count)                       \
  do {                                               \
    rc = dev_alloc(p, &(ptr), (count));              \
    if (rc != GIGL_OK) {                             \
      gigl_typed_plan_destroy(p);                    \
      return rc;                                     \
    }                                                \
  } while (0)
  for (int o = 0; o < n_ops; ++o) {
    const int64_t w = p->width[o], f = p->ops[o].fanout;
    PLAN_ALLOC(p->front[o], b * w);
    PLAN_ALLOC(p->ksum[o], b * w);
    PLAN_ALLOC(p->nbr[o], b * w * f);
    PLAN_ALLOC(p->cnt[o], b * w);
    PLAN_ALLOC(p->ran[o], b);
  }
  p->cand.resize(n_node_types);
  p->sorted.resize(n_node_types);
  p->nodes.resize(n_node_types);
  for (int t = 0; t < n_node_types; ++t) {
    const int64_t m = b * p->cand_per_root[t];
    p->max_items = m > p->max_items ? m : p->max_items;
    PLAN_ALLOC(p->cand[t], m);
    PLAN_ALLOC(p->sorted[t], m);
    PLAN_ALLOC(p->nodes[t], m);
  }
  p->ekeys.resize(n_edge_slots);
  p->esorted.resize(n_edge_slots);
  p->edges.resize(n_edge_slots);
  for (int s = 0; s < n_edge_slots; ++s) {
    const int64_t m = b * p->pairs_per_root[s];
    p->max_items = m > p->max_items ? m : p->max_items;
    PLAN_ALLOC(p->ekeys[s], m);
    PLAN_ALLOC(p->esorted[s], m);
    PLAN_ALLOC(p->edges[s], m);
  }
  PLAN_ALLOC(p->n_nodes, n_node_types);
  PLAN_ALLOC(p->n_edges, n_edge_slots);
  PLAN_ALLOC(p->root_index, b);
  if (p->max_items >= ((int64_t)1 << 31)) {
    rc = gigl_fail(ctx, GIGL_E_UNSUPPORTED, "typed plan: %lld items per batch", (long long)p->max_items);
    gigl_typed_plan_destroy(p);
    return rc;
  }
  {
    const int64_t w1 = gigl_sort::scratch_words(p->max_items), w2 = gigl_sort::batch_scratch_words(gigl_sort::RS_MAX_SEGS);
    PLAN_ALLOC(p->scratch, w1 > w2 ? w1 : w2);
  }
  {
    int64_t s32 = 0, s64 = 0;  // (the batched sorts give every segment its own ping-pong buffer)
    for (int t = 0; t < n_node_types; ++t) s32 += b * p->cand_per_root[t];
    for (int s = 0; s < n_edge_slots; ++s) s64 += b * p->pairs_per_root[s];
    p->tmp32_words = s32 > p->max_items ? s32 : p->max_items;
    p->tmp64_words = s64 > p->max_items ? s64 : p->max_items;
    PLAN_ALLOC(p->tmp32, p->tmp32_words);
    PLAN_ALLOC(p->tmp64, p->tmp64_words);
  }
#undef PLAN_ALLOC
  *out = p;
  return GIGL_OK;
}

int32_t gigl_typed_plan_clone(gigl_typed_plan* src, gigl_ctx* ctx, gigl_typed_plan** out) {
  if (!src || !ctx || !out) return GIGL_E_INVALID_ARG;
  return gigl_typed_plan_create(ctx, src->ops.data(), src->n_ops, src->n_types, src->root_type, src->n_slots, src->b_max, out);
}

int32_t gigl_typed_plan_run(gigl_typed_plan* p, const uint32_t* roots, int32_t b) {
  int32_t rc = gigl_typed_plan_run_nodes(p, roots, b);
  if (rc == GIGL_OK) rc = gigl_typed_plan_run_edges(p, b);
  return rc;
}

int32_t gigl_typed_plan_run_nodes(gigl_typed_plan* p, const uint32_t* roots, int32_t b) {
  if (!p || !p->ctx) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, roots && b >= 1 && b <= p->b_max, "between 1 and %d roots", p->b_max);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // ---- the ops, in order
  for (int o = 0; o < p->n_ops; ++o) {
    const gigl_dag_op& op = p->ops[o];
    const int32_t w = p->width[o];
    FrontierSrc fs{};
    RanSrc rs{};
    fs.n = rs.n = op.n_parents;
    for (int k = 0; k < op.n_parents; ++k) {
      const int pp = op.parents[k];
      fs.nbr[k] = p->nbr[pp];
      fs.len[k] = p->width[pp] * p->ops[pp].fanout;
      rs.ran[k] = p->ran[pp];
    }
    hipLaunchKernelGGL(dag_frontier_kernel, grid_of((int64_t)b * w), dim3(TB), 0, st, fs, roots, (int64_t)b, w, p->front[o]);
    if (op.n_parents > 0) {
      const int32_t rc = gigl_rows_dedup(ctx, p->front[o], b, w);
      if (rc != GIGL_OK) return rc;
    }
    hipLaunchKernelGGL(dag_mask_kernel, dim3((unsigned)((b + TB / 64 - 1) / (TB / 64))), dim3(TB), 0, st, rs, roots,
                       (int64_t)b, w, p->front[o], p->ksum[o], p->ran[o]);
    const int32_t rc = gigl_expand_frontier(ctx, op.graph, p->front[o], p->ksum[o], (int64_t)b * w, op.fanout, op.hash_add, 1,
                                            p->window_end[o], p->nbr[o], p->cnt[o]);
    if (rc != GIGL_OK) return rc;
  }
  // ---- per node type: the distinct ids, ascending
  for (int t = 0; t < p->n_types; ++t) {
    const int64_t m = (int64_t)b * p->cand_per_root[t];
    if (m == 0) {
      gigl_fill_u32(st, (uint32_t*)(p->n_nodes + t), 0u, 1);  // (a kernel, not a memset node: the run may be captured)
      continue;
    }
    int64_t off = 0;
    if (t == p->root_type) {
      hipLaunchKernelGGL(copy_u32_kernel, grid_of(b), dim3(TB), 0, st, roots, (int64_t)b, p->cand[t]);
      off = b;
    }
    for (int o = 0; o < p->n_ops; ++o) {
      const gigl_dag_op& op = p->ops[o];
      const int64_t mw = (int64_t)b * p->width[o];
      if (op.frontier_node_type == t) {
        hipLaunchKernelGGL(cand_frontier_kernel, grid_of(mw), dim3(TB), 0, st, (const uint32_t*)p->front[o],
                           (const int32_t*)p->cnt[o], mw, p->cand[t] + off);
        off += mw;
      }
      if (op.result_node_type == t) {
        hipLaunchKernelGGL(copy_u32_kernel, grid_of(mw * op.fanout), dim3(TB), 0, st, (const uint32_t*)p->nbr[o],
                           mw * op.fanout, p->cand[t] + off);
        off += mw * op.fanout;
      }
    }
  }
  // ids below 2^id_bits - 1, the empty slot (GIGL_INVALID) all ones: the digits of [0, id_bits) order both.  All types in
  // the same launches when they fit the batched sort (sortscan.h), else type by type.
  {
    gigl_sort::SortSegs<uint32_t> sg{};
    gigl_sort::UniqSegs<uint32_t> ug{};
    int bits = 0;
    bool fused = true;
    int64_t tmp_off = 0;
    for (int t = 0; t < p->n_types; ++t) {
      const int64_t m = (int64_t)b * p->cand_per_root[t];
      if (m == 0) continue;
      fused = fused && sg.nseg < gigl_sort::RS_MAX_SEGS && m <= (int64_t)gigl_sort::RS_FUSED_MAX_TILES * gigl_sort::RS_TILE &&
              tmp_off + m <= p->tmp32_words;
      if (!fused) break;
      const int k = sg.nseg++;
      sg.in[k] = p->cand[t];
      sg.out[k] = p->sorted[t];
      sg.tmp[k] = p->tmp32 + tmp_off;
      sg.n[k] = m;
      tmp_off += m;
      ug.sorted[k] = p->sorted[t];
      ug.out[k] = p->nodes[t];
      ug.count[k] = p->n_nodes + t;
      ug.n[k] = m;
      bits = p->id_bits[t] > bits ? p->id_bits[t] : bits;
    }
    ug.nseg = sg.nseg;
    if (fused) {
      gigl_sort::gigl_radix_sort_batch<uint32_t>(st, sg, bits, p->scratch);
      gigl_sort::gigl_unique_compact_batch<uint32_t>(st, ug, (uint32_t)GIGL_INVALID, p->scratch);
    } else {
      for (int t = 0; t < p->n_types; ++t) {
        const int64_t m = (int64_t)b * p->cand_per_root[t];
        if (m == 0) continue;
        int shifts[8];
        const int nd = gigl_sort::add_digits(shifts, 0, 0, p->id_bits[t]);
        gigl_sort::gigl_radix_sort<uint32_t>(st, p->cand[t], p->sorted[t], p->tmp32, m, shifts, nd, p->scratch);
        gigl_sort::gigl_unique_compact<uint32_t>(st, p->sorted[t], m, (uint32_t)GIGL_INVALID, p->nodes[t], p->n_nodes + t,
                                                 p->scratch);
      }
    }
  }
  hipLaunchKernelGGL(root_index_kernel, grid_of(b), dim3(TB), 0, st, roots, (int64_t)b,
                     (const uint32_t*)p->nodes[p->root_type], (const int32_t*)(p->n_nodes + p->root_type), p->root_index);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_typed_plan_run_edges(gigl_typed_plan* p, int32_t b) {
  if (!p || !p->ctx) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, b >= 1 && b <= p->b_max, "between 1 and %d roots", p->b_max);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // ---- per edge slot: the distinct (src, dst) pairs as local ids, ascending by (src, dst)
  for (int s = 0; s < p->n_slots; ++s) {
    const int64_t m = (int64_t)b * p->pairs_per_root[s];
    if (m == 0) {
      gigl_fill_u32(st, (uint32_t*)(p->n_edges + s), 0u, 1);
      continue;
    }
    int64_t off = 0;
    for (int o = 0; o < p->n_ops; ++o) {
      const gigl_dag_op& op = p->ops[o];
      if (op.edge_slot != s) continue;
      const int64_t mw = (int64_t)b * p->width[o];
      hipLaunchKernelGGL(edge_keys_kernel, grid_of(mw * op.fanout), dim3(TB), 0, st, (const uint32_t*)p->front[o],
                         (const uint32_t*)p->nbr[o], mw, op.fanout, op.outgoing ? 1 : 0,
                         (const uint32_t*)p->nodes[op.frontier_node_type], (const int32_t*)(p->n_nodes + op.frontier_node_type),
                         (const uint32_t*)p->nodes[op.result_node_type], (const int32_t*)(p->n_nodes + op.result_node_type),
                         p->ekeys[s] + off);
      off += mw * op.fanout;
    }
  }
  // (src_local << 32 | dst_local): local ids lie below their type's candidate capacity, the empty key is all ones — the
  // two bit fields that can differ, packed side by side, order real keys and keep the empty ones last
  {
    gigl_sort::SortSegs<unsigned long long> sg{};
    gigl_sort::UniqSegs<unsigned long long> ug{};
    int bits = 0;
    bool fused = true;
    int64_t tmp_off = 0;
    for (int s = 0; s < p->n_slots; ++s) {
      const int64_t m = (int64_t)b * p->pairs_per_root[s];
      if (m == 0) continue;
      fused = fused && sg.nseg < gigl_sort::RS_MAX_SEGS && m <= (int64_t)gigl_sort::RS_FUSED_MAX_TILES * gigl_sort::RS_TILE &&
              tmp_off + m <= p->tmp64_words;
      if (!fused) break;
      const int lo = key_bits((int64_t)p->b_max * p->cand_per_root[p->slot_dst_type[s]] + 1);
      const int hi = key_bits((int64_t)p->b_max * p->cand_per_root[p->slot_src_type[s]] + 1);
      const int k = sg.nseg++;
      sg.in[k] = p->ekeys[s];
      sg.out[k] = p->esorted[s];
      sg.tmp[k] = p->tmp64 + tmp_off;
      sg.n[k] = m;
      sg.lo_bits[k] = lo;
      tmp_off += m;
      ug.sorted[k] = p->esorted[s];
      ug.out[k] = p->edges[s];
      ug.count[k] = p->n_edges + s;
      ug.n[k] = m;
      bits = lo + hi > bits ? lo + hi : bits;
    }
    ug.nseg = sg.nseg;
    if (fused) {
      gigl_sort::gigl_radix_sort_batch<unsigned long long>(st, sg, bits, p->scratch);
      gigl_sort::gigl_unique_compact_batch<unsigned long long>(st, ug, ~0ull, p->scratch);
    } else {
      for (int s = 0; s < p->n_slots; ++s) {
        const int64_t m = (int64_t)b * p->pairs_per_root[s];
        if (m == 0) continue;
        int shifts[8];
        int nd = gigl_sort::add_digits(shifts, 0, 0, key_bits((int64_t)p->b_max * p->cand_per_root[p->slot_dst_type[s]] + 1));
        nd = gigl_sort::add_digits(shifts, nd, 32, key_bits((int64_t)p->b_max * p->cand_per_root[p->slot_src_type[s]] + 1));
        gigl_sort::gigl_radix_sort<unsigned long long>(st, p->ekeys[s], p->esorted[s], p->tmp64, m, shifts, nd, p->scratch);
        gigl_sort::gigl_unique_compact<unsigned long long>(st, p->esorted[s], m, ~0ull, p->edges[s], p->n_edges + s, p->scratch);
      }
    }
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_typed_plan_merged_csr(gigl_typed_plan* p, int32_t b, const int32_t* type_order, int32_t n_types_used,
                                   const int32_t* slot_order, const int32_t* slot_etype, int32_t n_slots_used,
                                   gigl_typed_csr_out* out) {
  return gigl_typed_plan_merged_csr_ex(p, b, type_order, n_types_used, slot_order, slot_etype, n_slots_used, 0, out);
}

int32_t gigl_typed_plan_merged_csr_ex(gigl_typed_plan* p, int32_t b, const int32_t* type_order, int32_t n_types_used,
                                      const int32_t* slot_order, const int32_t* slot_etype, int32_t n_slots_used,
                                      int32_t capacity_layout, gigl_typed_csr_out* out) {
  if (!p || !out) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  *out = gigl_typed_csr_out{};
  GIGL_REQUIRE(ctx, type_order && slot_order && n_types_used >= 1 && n_types_used <= p->n_types && n_slots_used >= 1 &&
                        n_slots_used <= p->n_slots && b >= 1 && b <= p->b_max,
               "typed plan merged CSR: bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  MergedArgs a{};
  a.n_types = n_types_used;
  a.n_slots = n_slots_used;
  a.root_type = p->root_type;
  bool root_listed = false;
  uint32_t seen_t = 0, seen_s = 0;
  int64_t rows_cap = 0;
  for (int j = 0; j < n_types_used; ++j) {
    const int32_t t = type_order[j];
    GIGL_REQUIRE(ctx, t >= 0 && t < p->n_types && !(seen_t >> t & 1), "typed plan merged CSR: node type %d", t);
    seen_t |= 1u << t;
    a.type_order[j] = t;
    a.type_cap[j] = (int32_t)((int64_t)p->b_max * p->cand_per_root[t]);
    rows_cap += (int64_t)p->b_max * p->cand_per_root[t];
    root_listed |= t == p->root_type;
  }
  GIGL_REQUIRE(ctx, root_listed, "typed plan merged CSR: the roots' node type is not listed");
  for (int j = 0; j < n_slots_used; ++j) {
    const int32_t sl = slot_order[j];
    GIGL_REQUIRE(ctx, sl >= 0 && sl < p->n_slots && !(seen_s >> sl & 1) && p->slot_src_type[sl] >= 0,
                 "typed plan merged CSR: edge slot %d", sl);
    seen_s |= 1u << sl;
    GIGL_REQUIRE(ctx, (seen_t >> p->slot_src_type[sl] & 1) && (seen_t >> p->slot_dst_type[sl] & 1),
                 "typed plan merged CSR: slot %d joins a node type that is not listed", sl);
    a.slot_order[j] = sl;
    a.slot_src_type[j] = p->slot_src_type[sl];
    a.slot_dst_type[j] = p->slot_dst_type[sl];
    a.slot_etype[j] = slot_etype ? slot_etype[j] : j;
    a.src_cap[j] = (int32_t)((int64_t)p->b_max * p->cand_per_root[p->slot_src_type[sl]]);
    a.cap_off[j + 1] = a.cap_off[j] + (int64_t)p->b_max * p->pairs_per_root[sl];
    a.edges[j] = p->edges[sl];
  }
  a.capacity_layout = capacity_layout ? 1 : 0;
  const int64_t m = a.cap_off[n_slots_used];
  GIGL_REQUIRE(ctx, m < ((int64_t)1 << 31) && rows_cap < ((int64_t)1 << 31) - 1, "typed plan merged CSR: batch too large");
  {
    int64_t src_rows = 0;
    for (int j = 0; j < n_slots_used; ++j) src_rows += a.src_cap[j];
    GIGL_REQUIRE(ctx, src_rows < ((int64_t)1 << 31), "typed plan merged CSR: source blocks too large");
  }
  if (m > p->m_edges_cap || rows_cap > p->m_rows_cap) {  // (first call, or a wider listing than before)
    hipStreamSynchronize(ctx->stream);
    int32_t rc = GIGL_OK;
    const int64_t me = m > 0 ? m : 1;
#define CSR_ALLOC(ptr, count)            \
  do {                                   \
    rc = dev_alloc(p, &(ptr), (count));  \
    if (rc != GIGL_OK) return rc;        \
  } while (0)
    CSR_ALLOC(p->m_keys, me);
    CSR_ALLOC(p->m_sorted, me);
    CSR_ALLOC(p->m_col, me);
    CSR_ALLOC(p->m_etype, me);
    CSR_ALLOC(p->m_root_col, me);
    CSR_ALLOC(p->m_root_etype, me);
    CSR_ALLOC(p->m_rowptr, rows_cap + 2);
    CSR_ALLOC(p->m_root_len, (int64_t)p->b_max + 1);
    CSR_ALLOC(p->m_root_rowptr, (int64_t)p->b_max + 1);
    CSR_ALLOC(p->m_counts, 2);
    if (!p->m_off) CSR_ALLOC(p->m_off, 1);
    CSR_ALLOC(p->m_tmp, me);
    {
      const int64_t w1 = gigl_sort::scratch_words(me), w2 = gigl_sort::batch_scratch_words(1);
      CSR_ALLOC(p->m_scratch, w1 > w2 ? w1 : w2);
    }
#undef CSR_ALLOC
    p->m_edges_cap = m;
    p->m_rows_cap = rows_cap;
  }
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(merged_offsets_kernel, dim3(1), dim3(1), 0, st, a, (const int32_t*)p->n_nodes, (const int32_t*)p->n_edges,
                     p->m_off, p->m_counts);
  if (m > 0) {
    hipLaunchKernelGGL(merged_keys_kernel, grid_of(m), dim3(TB), 0, st, a, (const int32_t*)p->n_edges,
                       (const MergedOffsets*)p->m_off, m, p->m_keys);
    // keys are (destination << 32 | position) with positions ascending in input order: a STABLE sort by the destination
    // field alone is the full order (destinations < rows_cap, the empty key all ones)
    if (m <= (int64_t)gigl_sort::RS_FUSED_MAX_TILES * gigl_sort::RS_TILE) {
      gigl_sort::SortSegs<unsigned long long> sg{};
      sg.nseg = 1;
      sg.in[0] = p->m_keys;
      sg.out[0] = p->m_sorted;
      sg.tmp[0] = p->m_tmp;
      sg.n[0] = m;
      sg.lo_bits[0] = 0;  // the virtual key is the destination field alone
      gigl_sort::gigl_radix_sort_batch<unsigned long long>(st, sg, key_bits(rows_cap + 2), p->m_scratch);
    } else {
      int shifts[8];
      const int nd = gigl_sort::add_digits(shifts, 0, 32, key_bits(rows_cap + 2));
      gigl_sort::gigl_radix_sort<unsigned long long>(st, p->m_keys, p->m_sorted, p->m_tmp, m, shifts, nd, p->m_scratch);
    }
    hipLaunchKernelGGL(merged_fill_kernel, grid_of(m), dim3(TB), 0, st, a, (const MergedOffsets*)p->m_off,
                       (const unsigned long long*)p->m_sorted, m, p->m_col, p->m_etype);
  }
  hipLaunchKernelGGL(merged_rowptr_kernel, grid_of(p->m_rows_cap + 1), dim3(TB), 0, st, (const MergedOffsets*)p->m_off,
                     (const unsigned long long*)p->m_sorted, p->m_rows_cap, p->m_rowptr);
  hipLaunchKernelGGL(root_len_kernel, grid_of((int64_t)b + 1), dim3(TB), 0, st, (const MergedOffsets*)p->m_off, p->root_type,
                     (const int32_t*)p->root_index, b, (const int32_t*)p->m_rowptr, p->m_root_len);
  gigl_sort::gigl_exclusive_scan_small(st, (const int32_t*)p->m_root_len, p->m_root_rowptr, (int64_t)b + 1);
  hipLaunchKernelGGL(root_rows_kernel, grid_of((int64_t)b * 64), dim3(TB), 0, st, (const MergedOffsets*)p->m_off, p->root_type,
                     (const int32_t*)p->root_index, b, (const int32_t*)p->m_rowptr, (const int32_t*)p->m_col,
                     (const int32_t*)p->m_etype, (const int32_t*)p->m_root_rowptr, p->m_root_col, p->m_root_etype);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  out->rowptr = p->m_rowptr;
  out->col = p->m_col;
  out->etype = p->m_etype;
  out->counts = p->m_counts;
  out->root_rowptr = p->m_root_rowptr;
  out->root_col = p->m_root_col;
  out->root_etype = p->m_root_etype;
  out->edges_cap = m;
  out->rows_cap = p->m_rows_cap;
  return GIGL_OK;
}

// The plan's sort + distinct step as a call of its own (tests; callers that number ids of their own): the distinct
// keys of keys[0 .. n) other than `pad`, ascending, count on the device.
int32_t gigl_sort_distinct_u64(gigl_ctx* ctx, const unsigned long long* keys, int64_t n, int32_t low_bits, int32_t high_bits,
                               unsigned long long pad, unsigned long long* out, int32_t* count) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, keys && out && count && n >= 0 && n < ((int64_t)1 << 31), "bad arguments");
  GIGL_REQUIRE(ctx, low_bits >= 0 && low_bits <= 32 && high_bits >= 0 && high_bits <= 32, "bit fields outside [0,32]");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (n == 0) {
    gigl_fill_u32(st, (uint32_t*)count, 0u, 1);
    return GIGL_OK;
  }
  const int64_t sw = gigl_sort::scratch_words(n) > gigl_sort::batch_scratch_words(1) ? gigl_sort::scratch_words(n) : gigl_sort::batch_scratch_words(1);
  int32_t rc = gigl_arena_reset(ctx, n * 16 + sw * 4 + 1024);
  if (rc != GIGL_OK) return rc;
  unsigned long long* sorted = (unsigned long long*)gigl_arena_alloc(ctx, n * 8);
  unsigned long long* tmp = (unsigned long long*)gigl_arena_alloc(ctx, n * 8);
  int32_t* scratch = (int32_t*)gigl_arena_alloc(ctx, sw * 4);
  if (!sorted || !tmp || !scratch) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  if (n <= (int64_t)gigl_sort::RS_FUSED_MAX_TILES * gigl_sort::RS_TILE) {  // the batched form the plans use (one segment)
    gigl_sort::SortSegs<unsigned long long> sg{};
    gigl_sort::UniqSegs<unsigned long long> ug{};
    sg.nseg = ug.nseg = 1;
    sg.in[0] = keys;
    sg.out[0] = sorted;
    sg.tmp[0] = tmp;
    sg.n[0] = ug.n[0] = n;
    sg.lo_bits[0] = low_bits;
    ug.sorted[0] = sorted;
    ug.out[0] = out;
    ug.count[0] = count;
    gigl_sort::gigl_radix_sort_batch<unsigned long long>(st, sg, low_bits + high_bits, scratch);
    gigl_sort::gigl_unique_compact_batch<unsigned long long>(st, ug, pad, scratch);
  } else {
    int shifts[8];
    int nd = gigl_sort::add_digits(shifts, 0, 0, low_bits);
    nd = gigl_sort::add_digits(shifts, nd, 32, high_bits);
    gigl_sort::gigl_radix_sort<unsigned long long>(st, keys, sorted, tmp, n, shifts, nd, scratch);
    gigl_sort::gigl_unique_compact<unsigned long long>(st, sorted, n, pad, out, count, scratch);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_sort_distinct_u32(gigl_ctx* ctx, const uint32_t* keys, int64_t n, int32_t bits, uint32_t pad, uint32_t* out,
                               int32_t* count) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, keys && out && count && n >= 0 && n < ((int64_t)1 << 31) && bits >= 0 && bits <= 32, "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (n == 0) {
    gigl_fill_u32(st, (uint32_t*)count, 0u, 1);
    return GIGL_OK;
  }
  const int64_t sw = gigl_sort::scratch_words(n) > gigl_sort::batch_scratch_words(1) ? gigl_sort::scratch_words(n) : gigl_sort::batch_scratch_words(1);
  int32_t rc = gigl_arena_reset(ctx, n * 8 + sw * 4 + 1024);
  if (rc != GIGL_OK) return rc;
  uint32_t* sorted = (uint32_t*)gigl_arena_alloc(ctx, n * 4);
  uint32_t* tmp = (uint32_t*)gigl_arena_alloc(ctx, n * 4);
  int32_t* scratch = (int32_t*)gigl_arena_alloc(ctx, sw * 4);
  if (!sorted || !tmp || !scratch) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  if (n <= (int64_t)gigl_sort::RS_FUSED_MAX_TILES * gigl_sort::RS_TILE) {
    gigl_sort::SortSegs<uint32_t> sg{};
    gigl_sort::UniqSegs<uint32_t> ug{};
    sg.nseg = ug.nseg = 1;
    sg.in[0] = keys;
    sg.out[0] = sorted;
    sg.tmp[0] = tmp;
    sg.n[0] = ug.n[0] = n;
    ug.sorted[0] = sorted;
    ug.out[0] = out;
    ug.count[0] = count;
    gigl_sort::gigl_radix_sort_batch<uint32_t>(st, sg, bits, scratch);
    gigl_sort::gigl_unique_compact_batch<uint32_t>(st, ug, pad, scratch);
  } else {
    int shifts[8];
    const int nd = gigl_sort::add_digits(shifts, 0, 0, bits);
    gigl_sort::gigl_radix_sort<uint32_t>(st, keys, sorted, tmp, n, shifts, nd, scratch);
    gigl_sort::gigl_unique_compact<uint32_t>(st, sorted, n, pad, out, count, scratch);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_typed_plan_buffers(gigl_typed_plan* p, gigl_typed_plan_out* out) {
  if (!p || !out) return GIGL_E_INVALID_ARG;
  *out = gigl_typed_plan_out{};
  out->n_nodes = p->n_nodes;
  out->n_edges = p->n_edges;
  out->root_index = p->root_index;
  for (int t = 0; t < p->n_types; ++t) {
    out->nodes[t] = p->nodes[t];
    out->nodes_cap[t] = (int64_t)p->b_max * p->cand_per_root[t];
  }
  for (int s = 0; s < p->n_slots; ++s) {
    out->edges[s] = p->edges[s];
    out->edges_cap[s] = (int64_t)p->b_max * p->pairs_per_root[s];
  }
  for (int o = 0; o < p->n_ops; ++o) {
    out->op_frontier[o] = p->front[o];
    out->op_nbr[o] = p->nbr[o];
    out->op_cnt[o] = p->cnt[o];
    out->op_width[o] = p->width[o];
  }
  return GIGL_OK;
}

}  // extern "C"
