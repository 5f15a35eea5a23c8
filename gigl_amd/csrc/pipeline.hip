// pipeline.hip — one-call batch pipeline: k-hop sample -> batch union graph -> GraphSAGE forward ->
// per-root rows.  Everything a batch needs is enqueued by ONE host call on the ctx stream (the host
// side of the hop loop / layer loop lives here, in C++, like the reference's compiled sampler), and
// optionally replayed from hipGraphs so that the ~28 launches of a batch cost one or two host calls.
//
// Replaces, for a RootedNodeNeighborhood batch (paths relative to the reference root):
//   NodeAnchorBasedLinkPredictionModelingTaskSpec.infer_batch
//       python/gigl/src/common/modeling_task_specs/node_anchor_based_link_prediction_modeling_task_spec.py:626-655
//       (`model(data)[root_node_indices]`), and NodeClassificationModelingTaskSpec.infer_batch
//       (.../node_classification_modeling_task_spec.py:176-187),
//   fed by process_raw_pyg_samples_and_collate_fn (rooted_node_neighborhood_data_loader.py:161-241).
#include "common.h"

#include <algorithm>
#include <functional>
#include <new>
#include <vector>

namespace {

// a maximal run of consecutive stages: either replayed from a captured graph or launched eagerly
// (stages that contain launches selected for HIP-event timing stay eager: events recorded inside a
// captured graph cannot be timed on this runtime)
struct Segment {
  int s0 = 0, s1 = 0;  // stages [s0, s1)
  hipGraphExec_t exec = nullptr;
};

// stages [0, GRAPH_STAGES) = sample + union (the GRAPH part: integer work, latency-bound, independent of the weights)
constexpr int GRAPH_STAGES = 2;

}  // namespace

struct gigl_sage_plan {
  gigl_ctx* ctx = nullptr;
  gigl_graph* graph = nullptr;
  gigl_feat* feat = nullptr;
  int32_t b = 0, hops = 0;
  int32_t group_roots = 0;  // roots per independent batch (== b: one batch)
  int32_t fanouts[GIGL_MAX_HOPS] = {0};
  int32_t dims[GIGL_MAX_HOPS + 1] = {0};  // dims[0] = input dim, dims[l+1] = output dim of layer l
  const float* w[GIGL_MAX_HOPS] = {nullptr};     // fused [dims[l+1]][2*dims[l]] (= [W_l | W_r]), device, borrowed
  const float* bias[GIGL_MAX_HOPS] = {nullptr};  // device, borrowed, may be null
  int32_t act_last = 0;
  int32_t aggr = GIGL_AGGR_MEAN;  // the SAGE layers' reduction (gigl_sage_plan_set_aggr)
  // projected input (gigl_sage_plan_set_projected_input): [W_l x | W_r x] of EVERY node, [graph nodes][2*dims[1]]
  // fp32, device, borrowed — the first layer is then one gather (+ self row + bias + activation), no projection
  const float* proj = nullptr;
  // first layer over two fp16 planes per operand (three MFMAs instead of six), each operand brought into the top of the
  // half range by a power of two: hs_sa from the table's largest magnitude (plan_refresh_half_split: at create /
  // set_weights / set_aggr; 0 = the table keeps the layer on the bf16 planes), the weights' scale found on the device
  // at every run from the weights as they are (hs_dev = {s_a, s_w, 1 / (s_a s_w)}, gigl_hs_scale_update)
  bool hs0 = false;
  float hs_sa = 0.f;
  float* hs_dev = nullptr;
  // kind 1: GAT layers instead of SAGE layers (gigl_gat_plan_create): w[l] = lin weight [heads*channels][dims[l]],
  // layer 0 from the input side in one row pass (gigl_gat_input_layer_fused), layers >= 1 projection + attention
  int32_t kind = 0;
  int32_t heads[GIGL_MAX_HOPS] = {0}, channels[GIGL_MAX_HOPS] = {0};
  const float* att_src[GIGL_MAX_HOPS] = {nullptr};
  const float* att_dst[GIGL_MAX_HOPS] = {nullptr};
  float slope = 0.2f;
  float* gat_scratch = nullptr;    // first layer: folded vectors + the per-head tiled operands
  float* alpha_scratch = nullptr;  // layers >= 1: per-node attention dots [2 * act_rows * max heads]
  float* hw = nullptr;             // layers >= 1: projected source rows [act_rows][max heads*channels]
  // leaf-global union (union.hip): pure leaves get no local id and stay global ids in their parents' rows — the
  // plan never computes anything for them, it only gathers their feature rows
  bool leaf_global = false;
  bool wide = false;  // ctx->wide at creation: rows for the worst batch, every node numbered (generic union)
  bool alias_rows = false;  // tree.nbr[hops-1] == un.col + un.cap_edges (rows may alias tree segments)
  int64_t last_slots = 0;
  int64_t act_rows = 0;         // rows of abuf / hbuf
  int32_t* zero_dev = nullptr;  // a device int32 0 (hops == 1: no row of the layer-0 gather holds local ids)
  gigl_tree tree{};
  gigl_union un{};
  float* abuf = nullptr;  // [act_rows][2*max_in], act_rows = b*(1 + f0 + f0*f1 + ...) over hops-1 terms
  // the aggregated matrix in the projection's tiled layout ([row tile of 128][K chunk of 32][128][32]): a tile's chunk
  // is 16 KB contiguous instead of 128 pieces of 128 B at a stride of one row (agg.hip)
  bool tiled = false;
  bool two_source = false;  // the projection takes the self half of [mean | self] from the source rows (no self copy)
  float* hbuf[2] = {nullptr, nullptr};  // ping-pong [act_rows][max_out]
  // fused two-layer projection (agg.hip linear_fused2_kernel; two-layer SAGE, hidden 256, 2 * out <= 96, half-split first
  // layer over an fp32 table): layer 0's projection applies the last layer's [W_l | W_r] to its hidden rows before they
  // leave the workgroup — hbuf[0] then holds the two K-split planes of p = [W_l h | W_r h] ([2][act_rows][96]) instead of
  // hidden rows, and the last layer is one reduction over p rows written straight into the caller's `out`
  // layers >= 1 on the half split too (round 5): their operands are previous-layer outputs, bounded from the previous
  // layer's own scales (gigl_hs_chain_update) — hs_layer[l] = {s_h, s_w, 1 / (s_h s_w)} of layer l, chained from hs_dev
  float* hs_layer[GIGL_MAX_HOPS] = {nullptr};
  bool f2_ok = false;
  float* f2_dev = nullptr;  // {s_h, s_w2, 1 / (s_h s_w2)}: the second product's scales (gigl_fused2_prepare, per run)
  void* w2h = nullptr;      // W2's fp16 planes
  std::vector<void*> owned;
  // hipGraph replay
  bool use_graph = false;
  uint32_t* roots_buf = nullptr;  // static input of the captured graphs
  float* out_buf = nullptr;       // (placeholder `out` of the captures; the stage that writes `out` runs eagerly)
  std::vector<Segment> segs;
  int32_t cap_seed = 0, cap_mode = -1;
  uint32_t cap_prof_mask = 0;
  uint64_t cap_arena_gen = 0;  // the ctx arena the captured launches point into
  bool captured = false;
  // gigl_sage_plan_set_graph_stream: the GRAPH part of every call is issued on this stream (the caller's, typically of a
  // HIGHER priority than the ctx stream) and the layers wait for it on an event — with several plans in flight the
  // latency-bound sampler / union launches of one call are then dispatched ahead of the other calls' bandwidth-bound
  // layers instead of queueing for CUs behind them
  hipStream_t gstream = nullptr;
  bool split = false;
  hipEvent_t ev_in = nullptr, ev_graph = nullptr;
};

namespace {

// The activation buffers hold b*(1 + f0 + ...) rows — the number of nodes of level < hops when no root is another
// root's sampled neighbour.  Roots that ARE neighbours of each other add the children of those occurrences to the
// inner levels (up to the whole tree in a clique of roots): such a batch does not fit the workspace; its level
// counts are zeroed so that no later kernel touches the buffers, and it is reported through
// meta[GIGL_META_OVERFLOW] (the batch's rows are then invalid, like an LDS-sort overflow).
__global__ void guard_levels_kernel(int32_t* meta, int hops, int32_t act_rows) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  bool over = false;
  for (int l = 0; l < hops; ++l) over = over || meta[GIGL_META_LEVEL0 + l] > act_rows;
  if (over) {  // nothing is computed for this batch (rows would index past the workspace)
    for (int l = 0; l < hops; ++l) meta[GIGL_META_LEVEL0 + l] = 0;
    atomicAdd(&meta[GIGL_META_OVERFLOW], 1);
  }
}

__global__ void overflow_add_kernel(const int32_t* __restrict__ meta, int32_t* __restrict__ acc) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && meta[GIGL_META_OVERFLOW] != 0) atomicAdd(acc, 1);
}

// exact work counts of the last batch set, accumulated into acc[GIGL_STATS_LEN] (see gigl_sage_plan_stats)
struct StatsArgs {
  const uint32_t* roots;
  const uint32_t* nbr[GIGL_MAX_HOPS];
  const int32_t* cnt[GIGL_MAX_HOPS];
  int32_t fan[GIGL_MAX_HOPS];
  int64_t parents[GIGL_MAX_HOPS];  // parent slots of hop k
  int32_t hops;
  const int64_t* g_rowptr;  // resident graph (in-degrees of the frontier nodes)
  const int32_t* meta;
  const int32_t* rowptr;
  const int32_t* rowend;
  int64_t rows_cap;
};

__device__ __forceinline__ void stats_add(unsigned long long* acc, int slot, long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(&acc[slot], (unsigned long long)v);
}

__global__ __launch_bounds__(256) void plan_stats_kernel(StatsArgs a, unsigned long long* acc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int L = a.hops;
  long long sampled = 0, bytes = 0;
#pragma unroll
  for (int k = 0; k < GIGL_MAX_HOPS; ++k) {
    if (k >= L) break;
    const uint32_t* par = k == 0 ? a.roots : a.nbr[k - 1];
    for (int64_t p = t0; p < a.parents[k]; p += stride) {
      const uint32_t v = par[p];
      if (v == GIGL_INVALID) continue;
      const long long deg = a.g_rowptr[(int64_t)v + 1] - a.g_rowptr[v];
      // what a POSITION-keyed sampler must move per frontier node (round 5; SURVEY 8(d)'s `16 + 4 deg + 8 min(deg, f)`
      // charges the whole adjacency row, which the reference's rule never needs: the key of entry i is a function of the
      // position i alone, SamplingStrategy.scala:55): the row bounds (16 B); a row of <= f neighbours itself (4 deg);
      // otherwise the <= lambda = f + 4 sqrt(f) + 4 (position, key) pairs of the window's threshold list that can hold
      // the f smallest keys (8 B each) and the f chosen ids (4 f); the sampled (src, dst) pairs out (8 min(deg, f))
      const long long f = a.fan[k];
      const long long lam = f + (long long)(4.0f * sqrtf((float)f)) + 4;
      bytes += 16 + (deg <= f ? 4 * deg : 8 * (deg < lam ? deg : lam) + 4 * f) + 8 * (deg < f ? deg : f);
      sampled += a.cnt[k][p];
    }
  }
  long long agg[GIGL_MAX_HOPS] = {0, 0, 0, 0};
  const int32_t n_rows = a.meta[GIGL_META_LEVEL0 + L - 1];
  for (int64_t i = t0; i < n_rows && i < a.rows_cap; i += stride) {
    const long long len = a.rowend[i] - a.rowptr[i];
#pragma unroll
    for (int l = 0; l < GIGL_MAX_HOPS; ++l)
      if (l < L && i < a.meta[GIGL_META_LEVEL0 + (L - 1 - l)]) agg[l] += len;  // layer l computes rows of level <= L-1-l
  }
  long long agg_all = 0;
#pragma unroll
  for (int l = 0; l < GIGL_MAX_HOPS; ++l) agg_all += agg[l];
  stats_add(acc, GIGL_STATS_SAMPLED, sampled);
  stats_add(acc, GIGL_STATS_AGGREGATED, agg_all);
  stats_add(acc, GIGL_STATS_EXPAND_BYTES, bytes);
#pragma unroll
  for (int l = 0; l < GIGL_MAX_HOPS; ++l) stats_add(acc, GIGL_STATS_AGG_LAYER0 + l, agg[l]);
  if (t0 == 0) {
    atomicAdd(&acc[GIGL_STATS_UNION_EDGES], (unsigned long long)a.meta[GIGL_META_N_EDGES]);
    atomicAdd(&acc[GIGL_STATS_UNION_NODES], (unsigned long long)a.meta[GIGL_META_N_NODES]);
    atomicAdd(&acc[GIGL_STATS_OVERFLOW], (unsigned long long)a.meta[GIGL_META_OVERFLOW]);
    for (int l = 0; l < L; ++l)
      atomicAdd(&acc[GIGL_STATS_ROWS_LAYER0 + l], (unsigned long long)a.meta[GIGL_META_LEVEL0 + (L - 1 - l)]);
  }
}

// stages of one batch: 0 sample, 1 union, 2+2l gather l, 3+2l linear l, 2+2L take_rows
int n_stages(const gigl_sage_plan* p) { return 3 + 2 * p->hops; }

// (A/B knob: GIGL_PLAN_HS_LAYERS=0 keeps layers >= 1 on the three bf16 planes)
bool hs_layers_on() {
  static const bool on = [] {
    const char* e = getenv("GIGL_PLAN_HS_LAYERS");
    return !(e && e[0] == '0');
  }();
  return on;
}

// the fused two-layer projection applies to this plan as it is configured now (every input of the decision drops the
// captured graphs when it changes: weights, reduction, projected input, half split)
bool fused2_on(const gigl_sage_plan* p) {
  return p->f2_ok && p->kind == 0 && p->hops == 2 && p->tiled && p->two_source && p->hs0 && !p->proj &&
         p->feat->dtype == GIGL_DTYPE_F32 && (p->aggr == GIGL_AGGR_MEAN || p->aggr == GIGL_AGGR_SUM) &&
         (((uintptr_t)p->bias[0]) & 15) == 0;
}

uint32_t stage_kernel_mask(const gigl_sage_plan* p, int s) {
  if (s == 0) return (1u << GIGL_K_EXPAND) | (1u << GIGL_K_EXPAND_HEAVY) | (1u << GIGL_K_FIND_HEAVY);
  if (s == 1)
    return (1u << GIGL_K_UNION_INSERT) | (1u << GIGL_K_UNION_RELAX) | (1u << GIGL_K_UNION_NODES) |
           (1u << GIGL_K_UNION_EDGE_SORT) | (1u << GIGL_K_UNION_CSR);
  if (s == n_stages(p) - 1) return 0;
  return ((s - 2) & 1) ? (1u << GIGL_K_LINEAR) : (1u << GIGL_K_GATHER_MEAN);
}

int32_t enqueue_stage(gigl_sage_plan* p, int s, const uint32_t* roots, int32_t sampling_seed, int32_t mode,
                      float* out) {
  gigl_ctx* ctx = p->ctx;
  const int L = p->hops;
  if (s == 0)
    return gigl_sample_khop(ctx, p->graph, roots, p->b, p->fanouts, p->hops, sampling_seed, mode, &p->tree);
  if (s == 1) {
    int32_t rc = gigl_union_build_impl(ctx, roots, &p->tree, p->group_roots, &p->un,
                                       p->leaf_global ? (1 | (p->graph->multi ? 2 : 0)) : 0);
    if (rc != GIGL_OK) return rc;
    hipLaunchKernelGGL(guard_levels_kernel, dim3(1), dim3(64), 0, ctx->stream, p->un.meta, p->hops,
                       (int32_t)(p->act_rows < 0x7FFFFFFF ? p->act_rows : 0x7FFFFFFF));
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    return GIGL_OK;
  }
  if (s == n_stages(p) - 1) {
    const int dout = p->dims[L];
    if (fused2_on(p))  // the last layer over the p rows of the fused projection, one row per root, into `out`
      return gigl_sage_fused_out(ctx, p->hbuf[0], (int64_t)p->act_rows * gigl_fused2_row_floats(), p->un.rowptr, p->un.rowend,
                                 p->un.col, p->un.root_local, p->b, dout, p->bias[L - 1], p->act_last ? 1 : 0, p->aggr,
                                 p->un.meta, out);
    gigl_take_rows(ctx->stream, p->hbuf[(L - 1) & 1], p->un.root_local, p->b, dout, p->un.meta, out);
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    return GIGL_OK;
  }
  const int l = (s - 2) >> 1;
  if (l == 1 && fused2_on(p)) return GIGL_OK;  // (layer 1 runs as the last stage)
  const int32_t* n_rows = p->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
  const int d = p->dims[l];
  // layer l computes the nodes of level <= L-1-l: at most b*(1 + f0 + f0*f1 + ...) of them — the launch is
  // sized for that bound, not for the whole union (the exact count is read on the device)
  const int64_t rows_cap = gigl_level_rows(p->wide, p->b, p->fanouts, L, L - 1 - l);
  int64_t width = p->b;  // slots of hop L-1-l: with rows_cap, the rows of level <= L-l (the sources of layer l)
  for (int i = 0; i <= L - 1 - l; ++i) width *= p->fanouts[i];
  if (p->wide) width = 0;  // (rows_cap is the whole tree already)
  if (p->kind == 1) {
    const int act = (l < L - 1 || p->act_last) ? 1 : 0;
    const bool first = ((s - 2) & 1) == 0;
    if (l == 0) {
      if (!first) return GIGL_OK;  // (the first layer is one stage: aggregation + both heads' projection)
      const int32_t* n_local = p->leaf_global ? (L >= 2 ? p->un.meta + GIGL_META_LEVEL0 + (L - 2) : p->zero_dev) : nullptr;
      if (p->hs0) {
        const int32_t rc = gigl_hs_scale_update(ctx, p->w[0], (int64_t)p->heads[0] * p->channels[0] * d, p->hs_sa, p->hs_dev);
        if (rc != GIGL_OK) return rc;
      }
      return gigl_gat_input_layer_fused_hs(ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, n_local, p->w[0], p->att_src[0],
                                           p->att_dst[0], p->heads[0], p->channels[0], p->slope, p->un.rowptr, p->un.rowend,
                                           p->un.col, n_rows, rows_cap, p->bias[0], act, p->gat_scratch, p->hbuf[0],
                                           p->hs0 ? p->hs_dev : nullptr);
    }
    // sources of layer l: the rows layer l-1 computed (level <= L-l)
    const int32_t* n_src = p->un.meta + GIGL_META_LEVEL0 + (L - l);
    int64_t src_cap = rows_cap + width;
    if (first)
      return gigl_linear(ctx, p->hbuf[(l - 1) & 1], p->w[l], nullptr, n_src, src_cap, d, p->dims[l + 1], 0, p->hw);
    return gigl_gat_aggregate(ctx, p->hw, p->att_src[l], p->att_dst[l], p->heads[l], p->channels[l], p->slope, 1,
                              p->un.rowptr, p->un.rowend, p->un.col, n_src, src_cap, n_rows, rows_cap, p->bias[l], act,
                              p->alpha_scratch, p->hbuf[l & 1]);
  }
  if (l == 0 && p->proj) {
    // lin_l(mean_j x_j) = mean_j lin_l(x_j): the rows arrive projected, the layer is the reduction plus the self row
    if (((s - 2) & 1) != 0) return GIGL_OK;  // (no projection stage)
    const int act = (l < L - 1 || p->act_last) ? 1 : 0;
    return gigl_gather_project_mixed(
        ctx, p->proj, p->proj + p->dims[1], 2 * p->dims[1], p->dims[1], p->un.nodes, p->un.rowptr, p->un.rowend, p->un.col, n_rows, rows_cap,
        p->aggr, p->leaf_global ? (L >= 2 ? p->un.meta + GIGL_META_LEVEL0 + (L - 2) : p->zero_dev) : nullptr, p->bias[0],
        act, p->hbuf[0]);
  }
  // two-source operand: the projection reads the self half from the fp32 source rows themselves (the feature table
  // through union.nodes, or the previous layer's output), the gather writes the reduced half only
  // (an fp16 table's rows serve as the self half too when the layer runs the half split: they are its h1 plane)
  const bool self_half = l == 0 && p->hs0 && p->feat->dtype == GIGL_DTYPE_F16;
  const bool two_src = p->tiled && p->two_source && (l > 0 || p->feat->dtype == GIGL_DTYPE_F32 || self_half);
  const int32_t nkc = p->tiled ? ((two_src ? d : 2 * d) + 31) / 32 : 0;
  if (((s - 2) & 1) == 0) {
    if (p->tiled) {
      if (l == 0)  // (leaf-global: rows of level L-1 — >= the count through level L-2 — hold global source ids)
        return gigl_gather_reduce_mixed(
            ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, p->un.rowptr, p->un.rowend, p->un.col, n_rows, rows_cap,
            p->aggr, p->leaf_global ? (L >= 2 ? p->un.meta + GIGL_META_LEVEL0 + (L - 2) : p->zero_dev) : nullptr,
            p->abuf, nkc, nullptr, nullptr, nullptr, two_src ? 1 : 0);
      return gigl_gather_reduce_mixed(ctx, p->hbuf[(l - 1) & 1], GIGL_DTYPE_F32, d, nullptr, p->un.rowptr, p->un.rowend,
                                      p->un.col, n_rows, rows_cap, p->aggr, nullptr, p->abuf, nkc, nullptr, nullptr, nullptr,
                                      two_src ? 1 : 0);
    }
    if (l == 0 && p->leaf_global)  // rows of level L-1 (>= the count through level L-2) hold global source ids
      return gigl_gather_reduce_mixed(ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, p->un.rowptr, p->un.rowend,
                                      p->un.col, n_rows, rows_cap, p->aggr,
                                      L >= 2 ? p->un.meta + GIGL_META_LEVEL0 + (L - 2) : p->zero_dev, p->abuf);
    if (l == 0)
      return gigl_gather_reduce(ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, p->un.rowptr, p->un.rowend,
                                p->un.col, n_rows, rows_cap, p->aggr, p->abuf);
    return gigl_gather_reduce(ctx, p->hbuf[(l - 1) & 1], GIGL_DTYPE_F32, d, nullptr, p->un.rowptr, p->un.rowend,
                              p->un.col, n_rows, rows_cap, p->aggr, p->abuf);
  }
  const int act = (l < L - 1 || p->act_last) ? 1 : 0;
  const float* hs_scale = nullptr;
  if (l == 0 && p->hs0 && p->tiled) {  // the weights' scale follows the weights as they are now (training rewrites them in place)
    const int32_t rc = gigl_hs_scale_update(ctx, p->w[0], (int64_t)p->dims[1] * 2 * d, p->hs_sa, p->hs_dev);
    if (rc != GIGL_OK) return rc;
    hs_scale = p->hs_dev;
  }
  if (l >= 1 && two_src && p->hs0 && !p->proj && p->hs_layer[l] && hs_layers_on()) {
    // half split for a layer >= 1: three fp16 products instead of six bf16 ones
    const float* prev = l == 1 ? p->hs_dev : p->hs_layer[l - 1];
    float fan = 1.f;
    if (p->aggr == GIGL_AGGR_SUM)
      for (int k = 0; k < p->hops; ++k) fan = (float)p->fanouts[k] > fan ? (float)p->fanouts[k] : fan;
    int32_t rc = gigl_hs_chain_update(ctx, prev, p->bias[l - 1], p->dims[l], 2 * p->dims[l - 1], fan, p->w[l],
                                      (int64_t)p->dims[l + 1] * 2 * d, p->hs_layer[l]);
    if (rc != GIGL_OK) return rc;
    return gigl_linear_tiled(ctx, p->abuf, p->w[l], p->bias[l], n_rows, rows_cap, 2 * d, p->dims[l + 1], act,
                             p->hbuf[l & 1], p->hbuf[(l - 1) & 1], nullptr, d, d, p->hs_layer[l]);
  }
  if (l == 0 && fused2_on(p)) {
    int32_t rc = gigl_fused2_prepare(ctx, p->hs_dev, p->bias[0], p->w[0], p->w[1], p->dims[2], 2 * d, p->f2_dev, p->w2h);
    if (rc != GIGL_OK) return rc;
    return gigl_linear_fused2(ctx, p->abuf, p->w[0], p->bias[0], n_rows, rows_cap, 2 * d, p->hbuf[0],
                              (int64_t)p->act_rows * gigl_fused2_row_floats(), (const float*)p->feat->rows, p->un.nodes, d, d,
                              hs_scale, p->f2_dev, p->w2h,
                              getenv("GIGL_F2_ALL_WR") ? nullptr : p->un.meta + GIGL_META_LEVEL0);
  }
  if (two_src)
    return gigl_linear_tiled(ctx, p->abuf, p->w[l], p->bias[l], n_rows, rows_cap, 2 * d, p->dims[l + 1], act,
                             p->hbuf[l & 1], l == 0 ? (const float*)p->feat->rows : p->hbuf[(l - 1) & 1],
                             l == 0 ? p->un.nodes : nullptr, d, d, hs_scale, self_half);
  if (p->tiled)
    return gigl_linear_tiled(ctx, p->abuf, p->w[l], p->bias[l], n_rows, rows_cap, 2 * d, p->dims[l + 1], act,
                             p->hbuf[l & 1], nullptr, nullptr, 0, 0, hs_scale);
  return gigl_linear(ctx, p->abuf, p->w[l], p->bias[l], n_rows, rows_cap, 2 * d, p->dims[l + 1], act,
                     p->hbuf[l & 1]);
}

int32_t enqueue_range(gigl_sage_plan* p, int s0, int s1, const uint32_t* roots, int32_t sampling_seed, int32_t mode,
                      float* out) {
  for (int s = s0; s < s1; ++s) {
    int32_t rc = enqueue_stage(p, s, roots, sampling_seed, mode, out);
    if (rc != GIGL_OK) return rc;
  }
  return GIGL_OK;
}

void drop_graphs(gigl_sage_plan* p) {
  for (Segment& sg : p->segs)
    if (sg.exec) hipGraphExecDestroy(sg.exec);
  p->segs.clear();
  p->captured = false;
}

// split the stages into timed (eager) and untimed (captured) segments and capture the latter
int32_t capture_segments(gigl_sage_plan* p, int32_t sampling_seed, int32_t mode) {
  gigl_ctx* ctx = p->ctx;
  const int n = n_stages(p);
  const uint32_t mask = ctx->prof_mask;
  int s = 0;
  while (s < n) {
    Segment sg;
    sg.s0 = s;
    // (the last stage — the roots' rows into the caller's `out` — stays out of the graphs: launched eagerly with the
    // pointer of the call, instead of a captured write to a static buffer and a device copy of it per call)
    const bool timed = (stage_kernel_mask(p, s) & mask) != 0 || s == n - 1;
    int e = s + 1;
    while (e < n - 1 && ((stage_kernel_mask(p, e) & mask) != 0) == timed && s != n - 1 && !(p->split && e == GRAPH_STAGES)) ++e;
    sg.s1 = e;
    if (!timed) {
      hipGraph_t graph = nullptr;
      // (a graph-part segment of a split plan is captured from — and later launched on — the plan's graph stream)
      hipStream_t keep = ctx->stream;
      if (p->split && sg.s1 <= GRAPH_STAGES) ctx->stream = p->gstream;
      hipError_t err = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
      int32_t rc = GIGL_OK;
      if (err == hipSuccess) {
        rc = enqueue_range(p, sg.s0, sg.s1, p->roots_buf, sampling_seed, mode, p->out_buf);
        hipError_t e2 = hipStreamEndCapture(ctx->stream, &graph);
        if (rc == GIGL_OK && e2 != hipSuccess) err = e2;
      }
      ctx->stream = keep;
      if (rc == GIGL_OK && err == hipSuccess) err = hipGraphInstantiate(&sg.exec, graph, nullptr, nullptr, 0);
      if (graph) hipGraphDestroy(graph);
      if (rc != GIGL_OK) return rc;
      if (err != hipSuccess) {
        sg.exec = nullptr;
        return gigl_fail(ctx, GIGL_E_HIP, "capturing stages [%d,%d) of the batch pipeline failed: %s", sg.s0, sg.s1,
                         hipGetErrorString(err));
      }
    }
    p->segs.push_back(sg);
    s = e;
  }
  return GIGL_OK;
}


// ---- training step (gigl_sage_train_plan_*)

// cross-entropy on the roots' rows: one wave per root.  loss_rows[i] = lse(out[rl]) - out[rl][label] for the n_valid real
// roots (0 for padding); the gradient (softmax - onehot) / n_valid is ADDED to dout[rl] (zeroed before; two roots of a
// batch that are the same node share a row, hence the atomics — a row with one root gets plain values)
__device__ __forceinline__ void ce_root_row(const float* __restrict__ out, int width, const int32_t* __restrict__ root_local,
                                            const int64_t* __restrict__ labels, const int32_t* __restrict__ n_valid_dev, int b,
                                            const int32_t* __restrict__ meta, float* __restrict__ dout,
                                            float* __restrict__ loss_rows) {
  const int lane = threadIdx.x & 63;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= b) return;
  const int n_valid = *n_valid_dev;
  const int32_t rl = root_local[i];
  if (i >= n_valid || rl < 0 || meta[GIGL_META_OVERFLOW] != 0) {
    if (lane == 0) loss_rows[i] = meta[GIGL_META_OVERFLOW] != 0 ? __builtin_nanf("") : 0.f;
    return;
  }
  const float* row = out + (int64_t)rl * width;
  float mx = -3.402823466e38f;
  for (int c = lane; c < width; c += 64) mx = fmaxf(mx, row[c]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float se = 0.f;
  for (int c = lane; c < width; c += 64) se += __expf(row[c] - mx);
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
  const float lse = mx + __logf(se);
  const int64_t lab = labels[i];
  if (lab < 0 || lab >= width) {
    // (torch's cross_entropy asserts on such a target; here the batch is reported failed — NaN loss, no update — instead
    // of reading past the row: an ignore_index-style -1 or an output narrower than the label space)
    if (lane == 0) {
      loss_rows[i] = __builtin_nanf("");
      atomicAdd(const_cast<int32_t*>(&meta[GIGL_META_OVERFLOW]), 1);
    }
    return;
  }
  const float inv = 1.0f / (float)n_valid;
  for (int c = lane; c < width; c += 64) {
    const float pr = __expf(row[c] - lse);
    atomicAdd(&dout[(int64_t)rl * width + c], (pr - (c == (int)lab ? 1.f : 0.f)) * inv);
  }
  if (lane == 0) loss_rows[i] = lse - row[lab];
}

__global__ __launch_bounds__(256) void ce_roots_kernel(const float* __restrict__ out, int width, const int32_t* __restrict__ root_local,
                                                       const int64_t* __restrict__ labels, const int32_t* __restrict__ n_valid_dev,
                                                       int b, const int32_t* __restrict__ meta, float* __restrict__ dout,
                                                       float* __restrict__ loss_rows) {
  ce_root_row(out, width, root_local, labels, n_valid_dev, b, meta, dout, loss_rows);
}

// ---- round 6: the node-classification step's small launches folded together (each costs ~5 us of a 0.22-ms step) ----
// prep: the cleared block (gw | gb | dh), the transposed weights of layers >= 1 (the backward's da = dh . W needs W^T as the
// projection's weight operand; the weights only change in the previous step's Adam) and the loss ticket, in ONE launch
struct TrainPrep {
  uint32_t* zero;
  int64_t zero_words;
  const float* w[GIGL_MAX_HOPS];
  float* wt[GIGL_MAX_HOPS];
  int32_t rows[GIGL_MAX_HOPS], cols[GIGL_MAX_HOPS];
  int32_t n_t;
  int32_t* ticket;
};
// the step's inputs into the static buffers the captured launches read — labels, the number of real roots, where the
// caller wants the loss — in one launch (was: a device copy, a 32-bit fill and, after the step, another device copy)
__global__ __launch_bounds__(256) void train_stage_kernel(const int64_t* __restrict__ labels, int n_valid, int64_t* __restrict__ labels_buf,
                                                          int32_t* __restrict__ n_valid_buf, float** __restrict__ loss_slot,
                                                          float* loss_out, TrainPrep a) {
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = t0; i < n_valid; i += stride) labels_buf[i] = labels[i];
  if (t0 == 0) {
    n_valid_buf[0] = n_valid;
    *loss_slot = loss_out;
  }
  // ... and the prep work (this launch is eager, ahead of the captured layers: one node less in the replayed graph)
  for (int64_t i = t0; i < a.zero_words; i += stride) a.zero[i] = 0u;
  for (int q = 0; q < a.n_t; ++q) {
    const int64_t n = (int64_t)a.rows[q] * a.cols[q];
    for (int64_t i = t0; i < n; i += stride) {
      const int r = (int)(i / a.cols[q]), c = (int)(i - (int64_t)r * a.cols[q]);
      a.wt[q][(int64_t)c * a.rows[q] + r] = a.w[q][i];
    }
  }
}

// loss = sum_i loss_rows[i] / n_valid in a fixed order (one workgroup); the optimiser's step counter moves on
__global__ __launch_bounds__(1024) void loss_sum_kernel(const float* __restrict__ loss_rows, int b,
                                                        const int32_t* __restrict__ n_valid_dev, float* __restrict__ loss,
                                                        int32_t* __restrict__ step, const int32_t* __restrict__ meta,
                                                        int32_t* __restrict__ halt, float* const* loss_slot = nullptr) {
  __shared__ float s_p[16];
  float v = 0.f;
  for (int i = threadIdx.x; i < b; i += 1024) v += loss_rows[i];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += s_p[k];
    *loss = t / (float)(*n_valid_dev > 0 ? *n_valid_dev : 1);
    // (a failed batch applies no update — adam_kernel returns — so Adam's bias-correction exponent must not move either)
    // `halt` is STICKY: once a batch has failed, every later step of the queue trains nothing either (NaN loss) until the
    // host has seen it (gigl_sage_train_plan_resume) — steps are issued without a host read in between, and the batches
    // behind a failed one must not be applied ahead of its redo
    if (*halt != 0) *loss = __builtin_nanf("");
    else if (meta[GIGL_META_OVERFLOW] == 0) *step += 1;
    else *halt = 1;
    // (round 6: the caller's loss slot, set by train_stage_kernel — no device copy after the step)
    float* extra = loss_slot ? *loss_slot : nullptr;
    if (extra) *extra = *loss;
  }
}

// dy[i][c] = 0 where the layer's activated output is 0 (relu'), rows below *n_rows
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ dy, const float* __restrict__ y,
                                                        const int32_t* __restrict__ n_rows_dev, int width) {
  const int64_t n = (int64_t)(*n_rows_dev) * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (!(y[i] > 0.f)) dy[i] = 0.f;
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ wt) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    wt[(int64_t)c * rows + r] = w[i];
  }
}

// Adam with L2 weight decay (torch.optim.Adam: the decay joins the gradient), every parameter tensor in one launch
struct AdamPack {
  float* p[2 * GIGL_MAX_HOPS];
  const float* g[2 * GIGL_MAX_HOPS];
  float* m[2 * GIGL_MAX_HOPS];
  float* v[2 * GIGL_MAX_HOPS];
  int64_t n[2 * GIGL_MAX_HOPS];
  // round 6: a tensor's gradient may arrive as the weight-gradient kernel's per-chunk PARTIAL sums (part != NULL:
  // [chunks][n] floats, the chunks below ceil(*rows / rc) hold real rows) — summed here in chunk order, one launch less per layer
  const float* part[2 * GIGL_MAX_HOPS];
  const int32_t* rows[2 * GIGL_MAX_HOPS];
  int32_t rc[2 * GIGL_MAX_HOPS];
  int32_t count;
  float lr, beta1, beta2, eps, wd;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamPack a, const int32_t* __restrict__ step_dev,
                                                   const int32_t* __restrict__ meta, const int32_t* __restrict__ halt) {
  if (meta[GIGL_META_OVERFLOW] != 0 || *halt != 0) return;  // a failed batch trains nothing (its loss is NaN)
  const double t = (double)*step_dev;
  const float bc1 = (float)(1.0 - pow((double)a.beta1, t)), bc2s = (float)sqrt(1.0 - pow((double)a.beta2, t));
  const float step_size = a.lr / bc1;
  // gridDim.y > 1: one slice of the grid per tensor (the partial sums are chains of dependent-latency loads: the tensors'
  // chains run side by side instead of one after the other)
  const int k_lo = gridDim.y > 1 ? (int)blockIdx.y : 0, k_hi = gridDim.y > 1 ? (int)blockIdx.y + 1 : a.count;
  for (int k = k_lo; k < k_hi && k < a.count; ++k) {
    float* p = a.p[k];
    const float* g = a.g[k];
    float* m = a.m[k];
    float* v = a.v[k];
    const float* part = a.part[k];
    const int chunks = part ? (*a.rows[k] + a.rc[k] - 1) / a.rc[k] : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n[k]; i += (int64_t)gridDim.x * blockDim.x) {
      const float w = p[i];
      float gsum = 0.f;
      if (part) {  // (sixteen chunks' loads in flight, added in chunk order)
        int c = 0;
        for (; c + 15 < chunks; c += 16) {
          float pv[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) pv[q] = part[(int64_t)(c + q) * a.n[k] + i];
#pragma unroll
          for (int q = 0; q < 16; ++q) gsum += pv[q];
        }
        for (; c < chunks; ++c) gsum += part[(int64_t)c * a.n[k] + i];
      } else {
        gsum = g[i];
      }
      const float gr = gsum + a.wd * w;
      const float mm = m[i] + (gr - m[i]) * (1.f - a.beta1);
      const float vv = v[i] * a.beta2 + (1.f - a.beta2) * gr * gr;
      m[i] = mm;
      v[i] = vv;
      p[i] = w - step_size * (mm / (sqrtf(vv) / bc2s + a.eps));
    }
  }
}

}  // namespace

extern "C" {

int32_t gigl_sage_plan_destroy(gigl_sage_plan* p) {
  if (!p) return GIGL_OK;
  if (p->ctx) {
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
  }
  if (p->gstream) hipStreamSynchronize(p->gstream);
  drop_graphs(p);
  if (p->ev_in) hipEventDestroy(p->ev_in);
  if (p->ev_graph) hipEventDestroy(p->ev_graph);
  for (void* q : p->owned) hipFree(q);
  delete p;
  return GIGL_OK;
}

static int32_t plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b, const int32_t* fanouts,
                           int32_t hops, const int32_t* dims, const float* const* w, const float* const* bias,
                           int32_t act_last, bool with_abuf, gigl_sage_plan** out);
static int32_t plan_refresh_half_split(gigl_sage_plan* p);

int32_t gigl_sage_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b,
                              const int32_t* fanouts, int32_t hops, const int32_t* dims,
                              const float* const* w, const float* const* bias, int32_t act_last,
                              gigl_sage_plan** out) {
  return plan_create(ctx, graph, feat, b, fanouts, hops, dims, w, bias, act_last, true, out);
}

// with_abuf false: no [mean | self] operand buffer (the GAT plan aggregates into its own per-head operands)
static int32_t plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b, const int32_t* fanouts,
                           int32_t hops, const int32_t* dims, const float* const* w, const float* const* bias,
                           int32_t act_last, bool with_abuf, gigl_sage_plan** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, graph && feat && fanouts && dims && w, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS && b >= 1, "bad plan shape");
  GIGL_REQUIRE(ctx, dims[0] == feat->d, "dims[0]=%d != feature dim %d", dims[0], feat->d);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_sage_plan* p = new (std::nothrow) gigl_sage_plan();
  if (!p) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  p->ctx = ctx;
  p->graph = graph;
  p->feat = feat;
  p->b = b;
  p->group_roots = b;
  p->hops = hops;
  p->act_last = act_last;
  int64_t cap_nodes = 0, cap_edges = 0;
  int32_t max_in = 0, max_out = 0;
  for (int k = 0; k < hops; ++k) {
    p->fanouts[k] = fanouts[k];
    p->w[k] = w[k];
    p->bias[k] = bias ? bias[k] : nullptr;
    if (!w[k]) {
      delete p;
      return gigl_fail(ctx, GIGL_E_INVALID_ARG, "weight %d is null", k);
    }
  }
  for (int k = 0; k <= hops; ++k) {
    p->dims[k] = dims[k];
    if (dims[k] < 1) {
      delete p;
      return gigl_fail(ctx, GIGL_E_INVALID_ARG, "dims[%d]=%d", k, dims[k]);
    }
    if (k < hops && dims[k] > max_in) max_in = dims[k];
    if (k > 0 && dims[k] > max_out) max_out = dims[k];
  }
  if (gigl_union_capacity(b, fanouts, hops, &cap_nodes, &cap_edges) != GIGL_OK) {
    delete p;
    return gigl_fail(ctx, GIGL_E_INVALID_ARG, "bad fanouts");
  }
  auto alloc = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
    p->owned.push_back(q);
    return q;
  };
  bool ok = true;
  // (the two-hop leaf-global builds keep a hop-0 slot's children in a 64-bit mask: last-hop fanouts beyond 64 — the
  // workgroup-per-row sampler path — take the generic union build)
  p->wide = ctx->wide;
  p->leaf_global = !p->wide && hops <= 2 && graph->n < ((int64_t)1 << 31) && !(hops == 2 && fanouts[1] > GIGL_FAST_FANOUT);
  // two hops, leaf-global union: the last hop's sampled ids live right behind the union's col array, so that the
  // rows of level-1 nodes that occur once can BE their tree segments (union.hip, row aliasing)
  int64_t last_slots = b;
  for (int k = 0; k < hops; ++k) last_slots *= fanouts[k];
  p->alias_rows = p->leaf_global && hops == 2;
  p->last_slots = last_slots;
  p->un.col = (int32_t*)alloc((size_t)(cap_edges + (p->alias_rows ? last_slots : 0)) * 4);
  ok = p->un.col != nullptr;
  int64_t parents = b;
  for (int k = 0; k < hops && ok; ++k) {
    p->tree.cnt[k] = (int32_t*)alloc((size_t)parents * 4);
    parents *= fanouts[k];
    p->tree.nbr[k] = (p->alias_rows && k == hops - 1) ? (uint32_t*)(p->un.col + cap_edges)
                                                      : (uint32_t*)alloc((size_t)parents * 4);
    ok = p->tree.cnt[k] && p->tree.nbr[k];
  }
  p->un.meta = (int32_t*)alloc(GIGL_META_LEN * 4);
  p->un.nodes = (uint32_t*)alloc((size_t)cap_nodes * 4);
  p->un.rowptr = (int32_t*)alloc((size_t)(cap_nodes + 2) * 4);
  p->un.rowend = (int32_t*)alloc((size_t)(cap_nodes + 2) * 4);
  p->un.root_local = (int32_t*)alloc((size_t)b * 4);
  p->un.cap_nodes = cap_nodes;
  p->un.cap_edges = cap_edges;
  // activations exist only for nodes of level < hops (leaves are read straight from the feature table)
  const int64_t act_rows = gigl_level_rows(p->wide, b, fanouts, hops, hops - 1);
  p->tiled = getenv("GIGL_PLAN_ROW_MAJOR") == nullptr;  // (A/B knob)
  p->two_source = getenv("GIGL_PLAN_SELF_COPY") == nullptr;  // (A/B knob: set = gather writes [mean | self])
  for (int k = 0; k < hops; ++k)
    if ((dims[k] & 3) != 0 || dims[k] > 2048) p->tiled = false;
  if (with_abuf) {
    const size_t row_tiles = ((size_t)act_rows + 127) / 128, nkc = ((size_t)2 * max_in + 31) / 32;
    p->abuf = (float*)alloc(p->tiled ? row_tiles * nkc * 4096 * 4 : (size_t)act_rows * 2 * max_in * 4);
  }
  p->hbuf[0] = (float*)alloc((size_t)act_rows * max_out * 4);
  p->hbuf[1] = hops > 1 ? (float*)alloc((size_t)act_rows * max_out * 4) : p->hbuf[0];
  p->roots_buf = (uint32_t*)alloc((size_t)b * 4);
  p->out_buf = (float*)alloc((size_t)b * dims[hops] * 4);
  p->zero_dev = (int32_t*)alloc(64);  // [0]: an int32 0; [4..7): hs_dev scales, [8..10): its running maximum + ticket (zero)
  if (p->zero_dev && hipMemset(p->zero_dev, 0, 64) != hipSuccess) ok = false;
  p->hs_dev = p->zero_dev ? reinterpret_cast<float*>(p->zero_dev + 4) : nullptr;
  p->act_rows = act_rows;
  for (int k = 1; k < hops && with_abuf; ++k) {
    p->hs_layer[k] = (float*)alloc(64);
    ok = ok && p->hs_layer[k] && hipMemset(p->hs_layer[k], 0, 64) == hipSuccess;
  }
  if (with_abuf && hops == 2 && gigl_fused2_shape_ok(dims[0], dims[1], dims[2]) && getenv("GIGL_PLAN_NO_FUSE2") == nullptr &&
      (int64_t)max_out >= 2 * gigl_fused2_row_floats()) {
    p->f2_dev = (float*)alloc(64);
    p->w2h = alloc((size_t)gigl_fused2_w2h_bytes(2 * dims[0]));
    p->f2_ok = p->f2_dev && p->w2h && hipMemset(p->f2_dev, 0, 64) == hipSuccess;
    ok = ok && p->f2_ok;
  }
  ok = ok && p->zero_dev && p->un.meta && p->un.nodes && p->un.rowptr && p->un.rowend && p->un.col && p->un.root_local &&
       (p->abuf || !with_abuf) && p->hbuf[0] && p->hbuf[1] && p->roots_buf && p->out_buf;
  if (!ok) {
    gigl_sage_plan_destroy(p);
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of the batch workspace failed (cap_nodes=%lld)",
                     (long long)cap_nodes);
  }
  {
    const int32_t rc_hs = plan_refresh_half_split(p);
    if (rc_hs != GIGL_OK) {
      gigl_sage_plan_destroy(p);
      return rc_hs;
    }
  }
  *out = p;
  return GIGL_OK;
}

int32_t gigl_gat_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b, const int32_t* fanouts,
                             int32_t hops, const int32_t* heads, const int32_t* channels, const float* const* w,
                             const float* const* att_src, const float* const* att_dst, const float* const* bias,
                             float negative_slope, int32_t act_last, gigl_sage_plan** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, graph && feat && fanouts && heads && channels && w && att_src && att_dst, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS && b >= 1, "bad plan shape");
  int32_t dims[GIGL_MAX_HOPS + 1];
  dims[0] = feat->d;
  int32_t max_h = 1, max_hc = 1;
  for (int l = 0; l < hops; ++l) {
    GIGL_REQUIRE(ctx, heads[l] >= 1 && channels[l] >= 1 && att_src[l] && att_dst[l], "layer %d: bad heads / channels", l);
    dims[l + 1] = heads[l] * channels[l];
    if (heads[l] > max_h) max_h = heads[l];
    if (dims[l + 1] > max_hc) max_hc = dims[l + 1];
  }
  const int P = (feat->d + 255) / 256;
  if ((feat->d & 3) || (heads[0] != 1 && heads[0] != 2 && heads[0] != 4) || P > 4 ||
      (feat->dtype != GIGL_DTYPE_F32 && feat->dtype != GIGL_DTYPE_F16))
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "GAT plan: feature dim %d / %d heads outside the shapes the first layer is "
                     "built for (gigl_gat_input_layer_fused)", feat->d, heads[0]);
  gigl_sage_plan* p = nullptr;
  int32_t rc = plan_create(ctx, graph, feat, b, fanouts, hops, dims, w, bias, act_last, false, &p);
  if (rc != GIGL_OK) return rc;
  p->kind = 1;
  p->slope = negative_slope;
  for (int l = 0; l < hops; ++l) {
    p->heads[l] = heads[l];
    p->channels[l] = channels[l];
    p->att_src[l] = att_src[l];
    p->att_dst[l] = att_dst[l];
  }
  auto alloc = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
    p->owned.push_back(q);
    return q;
  };
  p->gat_scratch = (float*)alloc((size_t)gigl_gat_input_layer_fused_scratch(feat->d, heads[0], p->act_rows) * 4);
  p->alpha_scratch = (float*)alloc((size_t)2 * p->act_rows * max_h * 4);
  p->hw = (float*)alloc((size_t)p->act_rows * max_hc * 4);
  if (!p->gat_scratch || !p->alpha_scratch || !p->hw) {
    gigl_sage_plan_destroy(p);
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of the GAT plan's workspace failed");
  }
  rc = plan_refresh_half_split(p);
  if (rc != GIGL_OK) {
    gigl_sage_plan_destroy(p);
    return rc;
  }
  *out = p;
  return GIGL_OK;
}

int32_t gigl_gat_plan_set_weights(gigl_sage_plan* p, const float* const* w, const float* const* att_src,
                                  const float* const* att_dst, const float* const* bias) {
  if (!p || !w || !att_src || !att_dst || p->kind != 1) return GIGL_E_INVALID_ARG;
  for (int l = 0; l < p->hops; ++l) {
    GIGL_REQUIRE(p->ctx, w[l] && att_src[l] && att_dst[l], "layer %d: null weight", l);
    p->w[l] = w[l];
    p->att_src[l] = att_src[l];
    p->att_dst[l] = att_dst[l];
    p->bias[l] = bias ? bias[l] : nullptr;
  }
  if (p->captured) {  // weight pointers are baked into the captured kernels
    hipStreamSynchronize(p->ctx->stream);
    drop_graphs(p);
  }
  return plan_refresh_half_split(p);
}

// The first layer's operands are rows of the feature table reduced by mean / max (|.| <= the table's largest magnitude),
// by sum (<= fan-out times it) or by softmax weights (GAT: convex combinations), and the layer's weights.  The layer's
// projection runs over two fp16 planes per operand, each operand multiplied by a power of two that puts its largest
// magnitude into [2^14, 2^15) (undone in the epilogue; all exact): the table's factor is fixed here from the table's
// cached largest magnitude — or the table keeps the layer on the bf16 planes, see gigl_feat_half_split_scale — and the
// weights' factor is found on the device at every run.  Nothing here looks at the weights or synchronises (beyond the
// table's first look), so trainers may call _set_weights every step, also inside a capture.
static int32_t plan_refresh_half_split(gigl_sage_plan* p) {
  const bool before = p->hs0;
  const float sa_before = p->hs_sa;
  p->hs0 = false;
  p->hs_sa = 0.f;
  const bool gat = p->kind == 1 && p->gat_scratch;
  const bool sage = p->kind == 0 && p->tiled && p->abuf;  // (SAGE plans own abuf)
  if ((gat || sage) && p->feat && p->w[0] && p->hs_dev) {
    float fan = 1.f;  // (a sum over up to `fan-out` rows; mean, max and softmax weights stay inside the table's range)
    if (sage && p->aggr == GIGL_AGGR_SUM)
      for (int k = 0; k < p->hops; ++k) fan = (float)p->fanouts[k] > fan ? (float)p->fanouts[k] : fan;
    const int32_t rc = gigl_feat_half_split_scale(p->ctx, p->feat, fan, &p->hs_sa);
    if (rc != GIGL_OK) return rc;
    p->hs0 = p->hs_sa > 0.f;
  }
  if (p->captured && (before != p->hs0 || sa_before != p->hs_sa)) {
    hipStreamSynchronize(p->ctx->stream);
    drop_graphs(p);
  }
  return GIGL_OK;
}

int32_t gigl_sage_plan_set_weights(gigl_sage_plan* p, const float* const* w, const float* const* bias) {
  if (!p || !w) return GIGL_E_INVALID_ARG;
  for (int k = 0; k < p->hops; ++k) {
    if (!w[k]) return gigl_fail(p->ctx, GIGL_E_INVALID_ARG, "weight %d is null", k);
    p->w[k] = w[k];
    p->bias[k] = bias ? bias[k] : nullptr;
  }
  if (p->captured) {  // weight pointers are baked into the captured kernels
    hipStreamSynchronize(p->ctx->stream);
    drop_graphs(p);
  }
  return plan_refresh_half_split(p);
}

int32_t gigl_sage_plan_half_split(gigl_sage_plan* p) { return p && p->hs0 ? 1 : 0; }

int32_t gigl_sage_plan_fused_layers(gigl_sage_plan* p) { return p && fused2_on(p) ? gigl_fused2_planes() : 0; }

int32_t gigl_sage_plan_set_aggr(gigl_sage_plan* p, int32_t aggr) {
  if (!p) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(p->ctx, aggr == GIGL_AGGR_MEAN || aggr == GIGL_AGGR_SUM || aggr == GIGL_AGGR_MAX, "bad aggr %d", aggr);
  GIGL_REQUIRE(p->ctx, p->kind == 0, "the reduction applies to SAGE layers");
  if (p->aggr != aggr && p->captured) {  // (the reduction is baked into the captured launches)
    hipStreamSynchronize(p->ctx->stream);
    drop_graphs(p);
  }
  p->aggr = aggr;
  return plan_refresh_half_split(p);  // (a sum's operand is bounded by fan-out times the table's largest magnitude)
}

int32_t gigl_sage_plan_set_projected_input(gigl_sage_plan* p, const float* proj) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, p->kind == 0, "projected input applies to SAGE plans");
  if (proj) {
    GIGL_REQUIRE(ctx, p->aggr == GIGL_AGGR_MEAN || p->aggr == GIGL_AGGR_SUM,
                 "projected input needs a linear reduction: lin(max_j x_j) != max_j lin(x_j)");
    GIGL_REQUIRE(ctx, (p->dims[1] & 3) == 0 && p->dims[1] <= 2048, "projected input: first-layer width %d", p->dims[1]);
  }
  if (p->captured && p->proj != proj) {  // the pointer is baked into the captured launches
    hipStreamSynchronize(ctx->stream);
    drop_graphs(p);
  }
  p->proj = proj;
  return GIGL_OK;
}

int32_t gigl_sage_plan_set_groups(gigl_sage_plan* p, int32_t group_roots) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, group_roots >= 1 && p->b % group_roots == 0, "group_roots=%d does not divide the plan's b=%d",
               group_roots, p->b);
  if (p->captured) {  // the group split is baked into the captured kernels
    hipStreamSynchronize(ctx->stream);
    drop_graphs(p);
  }
  p->group_roots = group_roots;
  return GIGL_OK;
}

int32_t gigl_sage_plan_buffers(gigl_sage_plan* p, gigl_tree* tree, gigl_union* un) {
  if (!p) return GIGL_E_INVALID_ARG;
  if (tree) *tree = p->tree;
  if (un) {
    *un = p->un;
    if (p->alias_rows) un->cap_edges += p->last_slots;  // rowptr / rowend may point into the aliased tree segments
  }
  return GIGL_OK;
}

int32_t gigl_sage_plan_stats(gigl_sage_plan* p, const uint32_t* roots, int64_t* acc) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, roots && acc, "null argument");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  StatsArgs a{};
  a.roots = roots;
  a.hops = p->hops;
  int64_t parents = p->b, most = p->b;
  for (int k = 0; k < p->hops; ++k) {
    a.nbr[k] = p->tree.nbr[k];
    a.cnt[k] = p->tree.cnt[k];
    a.fan[k] = p->fanouts[k];
    a.parents[k] = parents;
    if (parents > most) most = parents;
    parents *= p->fanouts[k];
  }
  a.g_rowptr = p->graph->rowptr;
  a.meta = p->un.meta;
  a.rowptr = p->un.rowptr;
  a.rowend = p->un.rowend;
  a.rows_cap = p->un.cap_nodes;
  int64_t blocks = (most + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(plan_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, (unsigned long long*)acc);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_sage_plan_overflow_add(gigl_sage_plan* p, int32_t* acc) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, acc, "null argument");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(overflow_add_kernel, dim3(1), dim3(64), 0, ctx->stream, p->un.meta, acc);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_sage_plan_use_graph(gigl_sage_plan* p, int32_t on) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  if (on && ctx->stream == nullptr)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED,
                     "hipGraph replay needs a non-default stream (the legacy default stream cannot be captured): "
                     "bind the ctx to a created stream first (gigl_ctx_set_stream)");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  drop_graphs(p);
  p->use_graph = on != 0;
  return GIGL_OK;
}

int32_t gigl_sage_plan_flush_profile(gigl_sage_plan* p) {
  if (!p) return GIGL_E_INVALID_ARG;
  GIGL_HIP_CHECK(p->ctx, hipStreamSynchronize(p->ctx->stream));
  return GIGL_OK;
}

int32_t gigl_sage_plan_run_part(gigl_sage_plan* p, const uint32_t* roots, int32_t sampling_seed, int32_t mode,
                                float* out, int32_t part) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, roots && out && (part == GIGL_PLAN_PART_GRAPH || part == GIGL_PLAN_PART_LAYERS), "bad argument");
  if (mode == GIGL_MODE_REPLACE)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "the one-call plan needs duplicate-free trees (no with-replacement mode)");
  const int n = n_stages(p);
  return part == GIGL_PLAN_PART_GRAPH ? enqueue_range(p, 0, 2, roots, sampling_seed, mode, out)
                                      : enqueue_range(p, 2, n, roots, sampling_seed, mode, out);
}

int32_t gigl_sage_plan_run(gigl_sage_plan* p, const uint32_t* roots, int32_t sampling_seed, int32_t mode,
                           float* out) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, roots && out, "null argument");
  if (mode == GIGL_MODE_REPLACE)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "the one-call plan needs duplicate-free trees (no with-replacement mode)");
  if (!p->use_graph) return enqueue_range(p, 0, n_stages(p), roots, sampling_seed, mode, out);

  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // (another plan on this ctx may have grown — reallocated — the arena since the capture: the graphs' scratch addresses
  // are then stale)
  if (!p->captured || p->cap_mode != mode || p->cap_seed != sampling_seed || p->cap_prof_mask != ctx->prof_mask ||
      p->cap_arena_gen != ctx->arena_gen) {
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    drop_graphs(p);
    // one eager run first: sizes the arena, builds the hash threshold table, sets kernel attributes — none of
    // which may happen inside a capture.  (This batch is produced by that run.)
    const uint32_t keep_mask = ctx->prof_mask;
    ctx->prof_mask = 0;
    int32_t rc = enqueue_range(p, 0, n_stages(p), roots, sampling_seed, mode, out);
    ctx->prof_mask = keep_mask;
    if (rc != GIGL_OK) return rc;
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    rc = capture_segments(p, sampling_seed, mode);
    if (rc != GIGL_OK) {
      drop_graphs(p);
      return rc;
    }
    p->captured = true;
    p->cap_mode = mode;
    p->cap_seed = sampling_seed;
    p->cap_prof_mask = ctx->prof_mask;
    p->cap_arena_gen = ctx->arena_gen;
    return GIGL_OK;
  }
  if (p->split) {
    // graph part on the plan's graph stream: behind whatever produced `roots` on the ctx stream, and behind the layers of
    // the previous call (they read the union graph this call's graph part rewrites)
    GIGL_HIP_CHECK(ctx, hipEventRecord(p->ev_in, ctx->stream));
    GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(p->gstream, p->ev_in, 0));
    GIGL_HIP_CHECK(ctx, hipMemcpyAsync(p->roots_buf, roots, (size_t)p->b * 4, hipMemcpyDeviceToDevice, p->gstream));
    hipStream_t keep = ctx->stream;
    bool joined = false;
    for (const Segment& sg : p->segs) {
      const bool gpart = sg.s1 <= GRAPH_STAGES;
      if (!gpart && !joined) {  // the layers wait for the graph part
        GIGL_HIP_CHECK(ctx, hipEventRecord(p->ev_graph, p->gstream));
        GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(keep, p->ev_graph, 0));
        joined = true;
      }
      ctx->stream = gpart ? p->gstream : keep;
      int32_t rc = GIGL_OK;
      if (sg.exec) {
        if (hipGraphLaunch(sg.exec, ctx->stream) != hipSuccess) rc = GIGL_E_HIP;
      } else {
        rc = enqueue_range(p, sg.s0, sg.s1, p->roots_buf, sampling_seed, mode, out);
      }
      ctx->stream = keep;
      if (rc != GIGL_OK) return rc == GIGL_E_HIP ? gigl_fail(ctx, rc, "hipGraphLaunch failed") : rc;
    }
    return GIGL_OK;
  }
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(p->roots_buf, roots, (size_t)p->b * 4, hipMemcpyDeviceToDevice, ctx->stream));
  for (const Segment& sg : p->segs) {
    if (sg.exec) {
      GIGL_HIP_CHECK(ctx, hipGraphLaunch(sg.exec, ctx->stream));
    } else {
      int32_t rc = enqueue_range(p, sg.s0, sg.s1, p->roots_buf, sampling_seed, mode, out);
      if (rc != GIGL_OK) return rc;
    }
  }
  return GIGL_OK;
}

int32_t gigl_sage_plan_set_graph_stream(gigl_sage_plan* p, void* hip_stream, int32_t on) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, !on || hip_stream, "the graph part needs a created stream of its own");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (p->gstream) GIGL_HIP_CHECK(ctx, hipStreamSynchronize(p->gstream));
  drop_graphs(p);
  if (on && !p->ev_in) {
    if (hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_graph, hipEventDisableTiming) != hipSuccess)
      return gigl_fail(ctx, GIGL_E_HIP, "hipEventCreate failed");
  }
  p->gstream = on ? (hipStream_t)hip_stream : nullptr;
  p->split = on != 0;
  return GIGL_OK;
}


// ---- gigl_sage_train_plan: one TRAINING step per call, all of it in the library —
//   sample -> batch union graph (the one-call plan's leaf-global build) -> GraphSAGE forward keeping every layer's
//   operand and output -> cross-entropy on the roots -> backward (weight gradients by gigl_linear_weight_grad, the
//   layers' input gradients by one projection over the transposed weights + gigl_gather_mean_backward) -> Adam.
// Replaces the loop body of NodeClassificationModelingTaskSpec._train
// (python/gigl/src/common/modeling_task_specs/node_classification_modeling_task_spec.py:134-173) for batches sampled in
// HBM: ~25 launches and two memsets, no torch kernel between them, replayed from hipGraphs.
// The step has two parts.  The GRAPH part (sample + union) does not depend on the weights: it runs on a ctx and stream of
// the plan's own, into one of two workspaces, and a step that is told the NEXT batch (roots_next) starts that batch's graph part while the
// LAYERS part (forward, loss, backward, Adam: the caller's ctx and stream) of the current batch runs — a training step
// is a chain of small launches (one batch: the weights change between batches), so the two chains share the GPU well.
// graph workspaces of the training plan: the current batch + the next.  (Three — two batches ahead, two graph parts in
// flight on streams of their own — was measured: 0.266 against 0.245 ms/step; the second graph part takes more from the
// layers than it hides.  roots_next2 of _step2 is then not used.  Also measured: the first layer's aggregation — it
// depends on no weight — moved into the graph part, out of the layers' serial chain: 0.257 against 0.245 ms/step; the
// two parts already share the GPU, the step follows the SUM of the launches more than the longer chain.)
constexpr int TRAIN_WS = 3;
// the input gradient of layers >= 1 by gigl_gather_mean_backward_transposed (every row written once, no cleared block, no
// float atomics) when the hidden widths allow float4 rows; GIGL_TRAIN_BWD_ATOMIC=1 keeps the scatter (A/B)
static bool train_bwd_gather(const int32_t* dims, int32_t hops) {
  if (getenv("GIGL_TRAIN_BWD_ATOMIC")) return false;
  for (int l = 1; l < hops; ++l)
    if (dims[l] & 3) return false;
  return true;
}

struct gigl_sage_train_plan {
  gigl_ctx* ctx = nullptr;         // the caller's (its stream carries the layers part; errors are reported on it)
  // two ctxs of the plan's own, for their ARENAS: scratch addresses are baked into the captured launches, and another
  // plan on the caller's ctx (the inference plan of an evaluation pass between epochs) may grow — reallocate — that arena
  gigl_ctx* lctx = nullptr;        // layers part (bound to the caller's stream at every step)
  gigl_ctx* side[TRAIN_WS] = {nullptr};  // graph part: sample + union — a ctx + stream per workspace
  gigl_sage_plan* base[TRAIN_WS] = {nullptr};  // tree / union workspaces (each on its own side ctx / stream: two graph
                                               // parts — of the next batch and of the one after — can be in flight)
  int32_t cur = 0;                 // workspace of the next step
  bool fetched[TRAIN_WS] = {false};  // the workspace holds the graph of a prefetched batch
  const uint32_t* fetched_roots[TRAIN_WS] = {nullptr};  // ... announced as these roots (the step must name the same ones)
  hipEvent_t ev_graph[TRAIN_WS] = {nullptr};   // graph part of the workspace done (recorded on its side stream)
  // layers part that read the workspace done (recorded on the caller's stream); [2]: the caller's stream as it stands
  // when a graph part is issued (the roots it is handed were written there)
  hipEvent_t ev_layers[TRAIN_WS + 1] = {nullptr};
  int32_t L = 0, b = 0, act_last = 0;
  int32_t dims[GIGL_MAX_HOPS + 1] = {0};
  int64_t rows_cap[GIGL_MAX_HOPS] = {0};  // rows layer l may compute
  float* w[GIGL_MAX_HOPS] = {nullptr};    // fused [dims[l+1]][2 dims[l]]: borrowed, UPDATED IN PLACE
  float* bias[GIGL_MAX_HOPS] = {nullptr};
  float* a[GIGL_MAX_HOPS] = {nullptr};    // [rows_cap[l]][2 dims[l]]: the layer's [mean | self] operand
  float* h[GIGL_MAX_HOPS] = {nullptr};    // [rows_cap[l]][dims[l+1]]: its output (activated below the last layer)
  float* dh[GIGL_MAX_HOPS] = {nullptr};   // gradient of h[l]
  bool bwd_gather = false;                // layers >= 1 hand their input gradient down by the transposed gather
  int32_t* tlists[TRAIN_WS][GIGL_MAX_HOPS] = {{nullptr}};  // its transposed lists per workspace and layer >= 1, built by the graph part
  float* da = nullptr;                    // [max rows_cap[l >= 1]][2 max dims]: gradient of a layer's operand
  float* wt = nullptr;                    // a layer's transposed weight
  // round 6 (GIGL_TRAIN_PLAN_UNFUSED=1 keeps the separate launches: A/B): per-layer transposed weights written by the prep
  // kernel, the weight gradients' partial sums (summed inside Adam), the fused loss's ticket and the loss pointer slot
  bool fused_small = false;
  float* wt_l[GIGL_MAX_HOPS] = {nullptr};
  float* part_w[GIGL_MAX_HOPS] = {nullptr};
  float* part_b[GIGL_MAX_HOPS] = {nullptr};
  int32_t part_rc[GIGL_MAX_HOPS] = {0};
  int32_t* ticket = nullptr;
  float** loss_slot = nullptr;
  // the loss rows' sum (+ the step counter / halt flag) on a BRANCH of the captured layers: beside the last layer's weight
  // gradient instead of in front of it, joined before Adam reads the counter
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  float* gw[GIGL_MAX_HOPS] = {nullptr};
  float* gb[GIGL_MAX_HOPS] = {nullptr};
  float* mom[4 * GIGL_MAX_HOPS] = {nullptr};  // m_w, v_w, m_b, v_b per layer
  void* zero_base = nullptr;              // gw | gb | dh: cleared at the start of every step
  size_t zero_bytes = 0;
  int64_t* labels_buf = nullptr;
  int32_t* n_valid_buf = nullptr;  // [0] real roots of the batch, [1] Adam's step counter, [2] sticky "a batch failed" (halt)
  float* loss_rows = nullptr;
  float* loss = nullptr;
  float lr = 0.01f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, wd = 0.f;
  std::vector<void*> owned;
  // hipGraph replay, per workspace and part
  hipGraphExec_t exec_graph[TRAIN_WS] = {nullptr}, exec_layers[TRAIN_WS] = {nullptr};
  bool warm_graph[TRAIN_WS] = {false}, warm_layers = false;  // one eager run of the part has sized arenas / built tables
  int32_t cap_seed = 0, cap_mode = -1;
};

namespace {

int32_t train_enqueue_graph(gigl_sage_train_plan* t, int k, int32_t sampling_seed, int32_t mode) {
  gigl_sage_plan* p = t->base[k];
  int32_t rc = enqueue_range(p, 0, 2, p->roots_buf, sampling_seed, mode, nullptr);  // sample + union (+ the level guard)
  // ... and who reads which source row in the backward of layers >= 1 (a function of the batch graph alone)
  for (int l = 1; l < t->L && rc == GIGL_OK && t->bwd_gather; ++l)
    rc = gigl_transposed_rows_build(p->ctx, p->un.rowptr, p->un.rowend, p->un.col, p->un.meta + GIGL_META_LEVEL0 + (t->L - 1 - l),
                                    t->rows_cap[l], p->un.meta + GIGL_META_LEVEL0 + (t->L - l), t->rows_cap[l - 1],
                                    p->un.cap_edges, t->tlists[k][l]);
  return rc;
}

int32_t train_enqueue_layers(gigl_sage_train_plan* t, int k) {
  gigl_sage_plan* p = t->base[k];
  gigl_ctx* ctx = t->lctx;
  const int L = t->L;
  hipStream_t st = ctx->stream;
  int32_t rc = GIGL_OK;
  const bool fz = t->fused_small;
  if (!fz) gigl_fill_u32(st, t->zero_base, 0u, (int64_t)(t->zero_bytes / 4));  // (fused: train_stage_kernel cleared it)
  const int32_t* n_local = p->leaf_global ? (L >= 2 ? p->un.meta + GIGL_META_LEVEL0 + (L - 2) : p->zero_dev) : nullptr;
  // ---- forward
  for (int l = 0; l < L; ++l) {
    const int32_t* n_rows = p->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    const int d = t->dims[l];
    if (l == 0)
      rc = gigl_gather_reduce_mixed(ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, p->un.rowptr, p->un.rowend, p->un.col,
                                    n_rows, t->rows_cap[0], GIGL_AGGR_MEAN, n_local, t->a[0]);
    else
      rc = gigl_gather_reduce(ctx, t->h[l - 1], GIGL_DTYPE_F32, d, nullptr, p->un.rowptr, p->un.rowend, p->un.col, n_rows,
                              t->rows_cap[l], GIGL_AGGR_MEAN, t->a[l]);
    if (rc != GIGL_OK) return rc;
    rc = gigl_linear(ctx, t->a[l], t->w[l], t->bias[l], n_rows, t->rows_cap[l], 2 * d, t->dims[l + 1],
                     (l < L - 1 || t->act_last) ? 1 : 0, t->h[l]);
    if (rc != GIGL_OK) return rc;
  }
  // ---- loss on the roots, its gradient into dh[L - 1]
  {
    const int width = t->dims[L];
    // (the loss rows and their sum stay two launches: folding the sum into the last workgroup of the first — a ticket behind
    // device-scope fences — took 22.7 us instead of 5.0 + 5.1, and the L2 write-backs of its 256 fences slowed the next batch's
    // graph part on the side stream: measured, round 6)
    hipLaunchKernelGGL(ce_roots_kernel, dim3((unsigned)((t->b + 3) / 4)), dim3(256), 0, st, (const float*)t->h[L - 1], width,
                       (const int32_t*)p->un.root_local, (const int64_t*)t->labels_buf, (const int32_t*)t->n_valid_buf, t->b,
                       (const int32_t*)p->un.meta, t->dh[L - 1], t->loss_rows);
    hipStream_t ls = st;
    if (fz && t->aux) {  // fork
      GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_fork, st));
      GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(t->aux, t->ev_fork, 0));
      ls = t->aux;
    }
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(1024), 0, ls, (const float*)t->loss_rows, t->b,
                       (const int32_t*)t->n_valid_buf, t->loss, t->n_valid_buf + 1, (const int32_t*)p->un.meta,
                       t->n_valid_buf + 2, fz ? (float* const*)t->loss_slot : (float* const*)nullptr);
    if (ls != st) GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_join, ls));
  }
  // ---- backward
  for (int l = L - 1; l >= 0; --l) {
    const int32_t* n_rows = p->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    const int d = t->dims[l], n_out = t->dims[l + 1];
    const bool act = l < L - 1 || t->act_last;
    if (fz)  // (the partial sums only: Adam adds them up)
      rc = gigl_linear_weight_grad_parts(ctx, t->dh[l], t->a[l], act ? t->h[l] : nullptr, n_rows, t->rows_cap[l], n_out, 2 * d,
                                         t->part_w[l], t->bias[l] ? t->part_b[l] : nullptr);
    else
      rc = gigl_linear_weight_grad(ctx, t->dh[l], t->a[l], act ? t->h[l] : nullptr, n_rows, t->rows_cap[l], n_out, 2 * d,
                                   t->gw[l], t->bias[l] ? t->gb[l] : nullptr);
    if (rc != GIGL_OK) return rc;
    if (l == 0) break;  // (the first layer's input is the feature table: no gradient)
    if (act) {
      int64_t blocks = (t->rows_cap[l] * n_out + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t->dh[l], (const float*)t->h[l], n_rows, n_out);
    }
    const float* wt = t->wt;
    if (fz) {
      wt = t->wt_l[l];  // (written by the prep kernel)
    } else {
      int64_t blocks = ((int64_t)n_out * 2 * d + 255) / 256;
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)t->w[l], n_out, 2 * d, t->wt);
    }
    rc = gigl_linear(ctx, t->dh[l], wt, nullptr, n_rows, t->rows_cap[l], n_out, 2 * d, 0, t->da);
    if (rc != GIGL_OK) return rc;
    if (t->bwd_gather)
      rc = gigl_gather_mean_backward_lists(ctx, t->da, d, p->un.rowptr, p->un.rowend, n_rows,
                                           p->un.meta + GIGL_META_LEVEL0 + (L - l), t->rows_cap[l - 1], t->tlists[k][l],
                                           GIGL_AGGR_MEAN, t->dh[l - 1]);
    else
      rc = gigl_gather_mean_backward(ctx, t->da, d, p->un.rowptr, p->un.rowend, p->un.col, n_rows, t->rows_cap[l], t->dh[l - 1]);
    if (rc != GIGL_OK) return rc;
  }
  // ---- Adam
  AdamPack ap{};
  for (int l = 0; l < L; ++l) {
    const int32_t* n_rows = p->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    ap.p[ap.count] = t->w[l];
    ap.g[ap.count] = t->gw[l];
    ap.part[ap.count] = fz ? t->part_w[l] : nullptr;
    ap.rows[ap.count] = n_rows;
    ap.rc[ap.count] = t->part_rc[l];
    ap.m[ap.count] = t->mom[4 * l];
    ap.v[ap.count] = t->mom[4 * l + 1];
    ap.n[ap.count++] = (int64_t)t->dims[l + 1] * 2 * t->dims[l];
    if (t->bias[l]) {
      ap.p[ap.count] = t->bias[l];
      ap.g[ap.count] = t->gb[l];
      ap.part[ap.count] = fz ? t->part_b[l] : nullptr;
      ap.rows[ap.count] = n_rows;
      ap.rc[ap.count] = t->part_rc[l];
      ap.m[ap.count] = t->mom[4 * l + 2];
      ap.v[ap.count] = t->mom[4 * l + 3];
      ap.n[ap.count++] = t->dims[l + 1];
    }
  }
  ap.lr = t->lr;
  ap.beta1 = t->beta1;
  ap.beta2 = t->beta2;
  ap.eps = t->eps;
  ap.wd = t->wd;
  if (fz && t->aux) GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, t->ev_join, 0));  // join: Adam reads the step counter / halt flag
  hipLaunchKernelGGL(adam_kernel, fz ? dim3(208, (unsigned)ap.count) : dim3(256), dim3(256), 0, st, ap,
                     (const int32_t*)(t->n_valid_buf + 1), (const int32_t*)p->un.meta, (const int32_t*)(t->n_valid_buf + 2));
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// run `body` on ctx->stream: eagerly the first time (arena sizing, table builds and kernel attributes cannot happen inside
// a capture), captured the second, replayed from then on
int32_t train_run_part(gigl_ctx* ctx, hipGraphExec_t* exec, bool* warm, const std::function<int32_t()>& body, int part) {
  // (A/B knob: GIGL_TRAIN_PLAN_EAGER=1 both parts eager, =graph / =layers only that part)
  static const char* ev = getenv("GIGL_TRAIN_PLAN_EAGER");
  const bool eager = ev && (ev[0] == '1' || (ev[0] == 'g' && part == 0) || (ev[0] == 'l' && part == 1));
  hipStream_t st = ctx->stream;
  if (eager || st == nullptr) return body();  // (the legacy default stream cannot be captured)
  if (!*exec && !*warm) {
    const int32_t rc = body();
    if (rc != GIGL_OK) return rc;
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(st));
    *warm = true;
    return GIGL_OK;
  }
  if (!*exec) {
    hipGraph_t graph = nullptr;
    int32_t rc = GIGL_OK;
    hipError_t err = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (err == hipSuccess) {
      rc = body();
      const hipError_t e2 = hipStreamEndCapture(st, &graph);
      if (rc == GIGL_OK && e2 != hipSuccess) err = e2;
    }
    if (rc == GIGL_OK && err == hipSuccess) err = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
    if (graph) hipGraphDestroy(graph);
    if (rc != GIGL_OK) return rc;
    if (err != hipSuccess) {
      *exec = nullptr;
      return gigl_fail(ctx, GIGL_E_HIP, "capturing a part of the training step failed: %s", hipGetErrorString(err));
    }
  }
  GIGL_HIP_CHECK(ctx, hipGraphLaunch(*exec, st));
  return GIGL_OK;
}

// the graph part of workspace k for `roots`, on the side stream; ev_graph[k] marks its end
int32_t train_graph_part(gigl_sage_train_plan* t, int k, const uint32_t* roots, int32_t sampling_seed, int32_t mode) {
  gigl_ctx* sc = t->side[k];
  if (t->cap_seed != sampling_seed || t->cap_mode != mode) {  // seed and mode are baked into the captured launches
    GIGL_HIP_CHECK(sc, hipStreamSynchronize(sc->stream));
    for (int i = 0; i < TRAIN_WS; ++i)
      if (t->exec_graph[i]) {
        if (t->side[i]) hipStreamSynchronize(t->side[i]->stream);
        hipGraphExecDestroy(t->exec_graph[i]);
        t->exec_graph[i] = nullptr;
      }
    t->cap_seed = sampling_seed;
    t->cap_mode = mode;
  }
  // the workspace is free once the layers part that last read it is done; `roots` was written on the caller's stream
  if (t->ev_layers[k]) GIGL_HIP_CHECK(sc, hipStreamWaitEvent(sc->stream, t->ev_layers[k], 0));
  GIGL_HIP_CHECK(sc, hipEventRecord(t->ev_layers[TRAIN_WS], t->ctx->stream));
  GIGL_HIP_CHECK(sc, hipStreamWaitEvent(sc->stream, t->ev_layers[TRAIN_WS], 0));
  GIGL_HIP_CHECK(sc, hipMemcpyAsync(t->base[k]->roots_buf, roots, (size_t)t->b * 4, hipMemcpyDeviceToDevice, sc->stream));
  const int32_t rc = train_run_part(sc, &t->exec_graph[k], &t->warm_graph[k],
                                    [&]() { return train_enqueue_graph(t, k, sampling_seed, mode); }, 0);
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(sc, hipEventRecord(t->ev_graph[k], sc->stream));
  return GIGL_OK;
}

}  // namespace

int32_t gigl_sage_train_plan_destroy(gigl_sage_train_plan* t) {
  if (!t) return GIGL_OK;
  if (t->ctx) {
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
  }
  for (int k = 0; k < TRAIN_WS; ++k)
    if (t->side[k]) hipStreamSynchronize(t->side[k]->stream);
  for (int k = 0; k < TRAIN_WS; ++k) {
    if (t->exec_graph[k]) hipGraphExecDestroy(t->exec_graph[k]);
    if (t->exec_layers[k]) hipGraphExecDestroy(t->exec_layers[k]);
    if (t->ev_graph[k]) hipEventDestroy(t->ev_graph[k]);
    if (t->base[k]) gigl_sage_plan_destroy(t->base[k]);
  }
  for (int k = 0; k < TRAIN_WS + 1; ++k)
    if (t->ev_layers[k]) hipEventDestroy(t->ev_layers[k]);
  if (t->aux) {
    hipStreamSynchronize(t->aux);
    hipStreamDestroy(t->aux);
  }
  if (t->ev_fork) hipEventDestroy(t->ev_fork);
  if (t->ev_join) hipEventDestroy(t->ev_join);
  for (void* q : t->owned) hipFree(q);
  for (int k = 0; k < TRAIN_WS; ++k)
    if (t->side[k]) gigl_ctx_destroy(t->side[k]);
  if (t->lctx) {
    gigl_ctx_set_stream(t->lctx, nullptr);  // (the stream is the caller's)
    gigl_ctx_destroy(t->lctx);
  }
  delete t;
  return GIGL_OK;
}

int32_t gigl_sage_train_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b, const int32_t* fanouts,
                                    int32_t hops, const int32_t* dims, float* const* w, float* const* bias,
                                    int32_t act_last, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    gigl_sage_train_plan** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, graph && feat && fanouts && dims && w, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS && b >= 1, "bad plan shape");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_sage_train_plan* t = new (std::nothrow) gigl_sage_train_plan();
  if (!t) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  t->ctx = ctx;
  int32_t rc = gigl_ctx_create(ctx->device, &t->lctx);
  for (int k = 0; k < TRAIN_WS && rc == GIGL_OK; ++k) {
    rc = gigl_ctx_create(ctx->device, &t->side[k]);
    if (rc != GIGL_OK) break;
    t->side[k]->wide = ctx->wide;
    rc = plan_create(t->side[k], graph, feat, b, fanouts, hops, dims, (const float* const*)w, (const float* const*)bias,
                     act_last, false, &t->base[k]);
    if (rc != GIGL_OK) gigl_fail(ctx, rc, "%s", gigl_last_error(t->side[k]));
    if (rc == GIGL_OK && hipEventCreateWithFlags(&t->ev_graph[k], hipEventDisableTiming) != hipSuccess) rc = GIGL_E_HIP;
  }
  for (int k = 0; k < TRAIN_WS + 1 && rc == GIGL_OK; ++k)
    if (hipEventCreateWithFlags(&t->ev_layers[k], hipEventDisableTiming) != hipSuccess) rc = GIGL_E_HIP;
  if (rc != GIGL_OK) {
    gigl_sage_train_plan_destroy(t);
    return rc;
  }
  t->L = hops;
  t->b = b;
  t->act_last = act_last;
  t->lr = lr;
  t->beta1 = beta1;
  t->beta2 = beta2;
  t->eps = eps;
  t->wd = weight_decay;
  auto alloc = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
    t->owned.push_back(q);
    return q;
  };
  bool ok = true;
  size_t zero_floats = 0, da_floats = 16, wt_floats = 16;
  for (int l = 0; l <= hops; ++l) t->dims[l] = dims[l];
  t->bwd_gather = train_bwd_gather(dims, hops);
  for (int l = 0; l < hops; ++l) {
    t->w[l] = w[l];
    t->bias[l] = bias ? bias[l] : nullptr;
    const int64_t rows = gigl_level_rows(ctx->wide, b, fanouts, hops, hops - 1 - l);  // layer l computes the nodes of level <= L-1-l
    t->rows_cap[l] = rows;
    const size_t nw = (size_t)dims[l + 1] * 2 * dims[l];
    zero_floats += nw + dims[l + 1] + (t->bwd_gather && l < hops - 1 ? 0 : (size_t)rows * dims[l + 1]);
    if (l >= 1) {
      da_floats = std::max(da_floats, (size_t)rows * 2 * dims[l]);
      wt_floats = std::max(wt_floats, nw);
    }
    t->a[l] = (float*)alloc((size_t)rows * 2 * dims[l] * 4);
    t->h[l] = (float*)alloc((size_t)rows * dims[l + 1] * 4);
    for (int k = 0; k < 4; ++k) {
      const size_t n = k < 2 ? nw : (size_t)dims[l + 1];
      t->mom[4 * l + k] = (float*)alloc(n * 4);
      if (t->mom[4 * l + k] && hipMemset(t->mom[4 * l + k], 0, n * 4) != hipSuccess) ok = false;
      ok = ok && t->mom[4 * l + k];
    }
    ok = ok && t->a[l] && t->h[l];
  }
  float* z = (float*)alloc(zero_floats * 4);
  t->zero_base = z;
  t->zero_bytes = zero_floats * 4;
  for (int l = 0; l < hops && z; ++l) {
    t->gw[l] = z;
    z += (size_t)dims[l + 1] * 2 * dims[l];
    t->gb[l] = z;
    z += dims[l + 1];
    if (t->bwd_gather && l < hops - 1) {  // (written whole by the transposed gather: not part of the cleared block)
      t->dh[l] = (float*)alloc((size_t)t->rows_cap[l] * dims[l + 1] * 4);
      ok = ok && t->dh[l];
      continue;
    }
    t->dh[l] = z;
    z += (size_t)t->rows_cap[l] * dims[l + 1];
  }
  t->da = (float*)alloc(da_floats * 4);
  t->wt = (float*)alloc(wt_floats * 4);
  for (int k = 0; k < TRAIN_WS && t->bwd_gather; ++k)
    for (int l = 1; l < hops; ++l) {
      t->tlists[k][l] = (int32_t*)alloc((size_t)gigl_transposed_rows_words(t->rows_cap[l - 1], t->base[k]->un.cap_edges) * 4);
      ok = ok && t->tlists[k][l];
    }
  t->labels_buf = (int64_t*)alloc((size_t)b * 8);
  t->n_valid_buf = (int32_t*)alloc(16);
  t->loss_rows = (float*)alloc((size_t)b * 4);
  t->loss = (float*)alloc(16);
  ok = ok && t->zero_base && t->da && t->wt && t->labels_buf && t->n_valid_buf && t->loss_rows && t->loss;
  if (ok && hipMemset(t->n_valid_buf, 0, 16) != hipSuccess) ok = false;
  t->fused_small = getenv("GIGL_TRAIN_PLAN_UNFUSED") == nullptr;
  if (t->fused_small && ok) {
    // (measured, round 6: the sum on a BRANCH of the captured layers — forked after the loss rows, joined before Adam — took
    // the step from 0.201 to 0.263 ms: a graph with a second branch is replayed through two queues with a barrier packet per
    // edge.  GIGL_TRAIN_LOSS_BRANCH=1 turns it on for the A/B; off by default)
    if (getenv("GIGL_TRAIN_LOSS_BRANCH") != nullptr &&
        (hipStreamCreateWithFlags(&t->aux, hipStreamNonBlocking) != hipSuccess ||
         hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming) != hipSuccess))
      t->aux = nullptr;
    t->ticket = (int32_t*)alloc(16);
    t->loss_slot = (float**)alloc(16);
    ok = t->ticket && t->loss_slot && hipMemset(t->ticket, 0, 16) == hipSuccess && hipMemset(t->loss_slot, 0, 16) == hipSuccess;
    for (int l = 0; l < hops && ok; ++l) {
      const int n_out = dims[l + 1], k2 = 2 * dims[l];
      const int64_t chunks = gigl_linear_weight_grad_chunks(t->rows_cap[l], n_out, k2, &t->part_rc[l]);
      t->part_w[l] = (float*)alloc((size_t)chunks * n_out * k2 * 4);
      t->part_b[l] = (float*)alloc((size_t)chunks * n_out * 4);
      if (l >= 1) t->wt_l[l] = (float*)alloc((size_t)n_out * k2 * 4);
      ok = t->part_w[l] && t->part_b[l] && (l == 0 || t->wt_l[l]);
    }
  }
  if (!ok) {
    gigl_sage_train_plan_destroy(t);
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of the training workspace failed");
  }
  *out = t;
  return GIGL_OK;
}

int32_t gigl_sage_train_plan_step(gigl_sage_train_plan* t, const uint32_t* roots, const int64_t* labels, int32_t n_valid,
                                  const uint32_t* roots_next, int32_t sampling_seed, int32_t mode, float* loss_out) {
  return gigl_sage_train_plan_step2(t, roots, labels, n_valid, roots_next, nullptr, sampling_seed, mode, loss_out);
}

int32_t gigl_sage_train_plan_step2(gigl_sage_train_plan* t, const uint32_t* roots, const int64_t* labels, int32_t n_valid,
                                   const uint32_t* roots_next, const uint32_t* roots_next2, int32_t sampling_seed,
                                   int32_t mode, float* loss_out) {
  if (!t) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = t->ctx;
  GIGL_REQUIRE(ctx, roots && labels && n_valid >= 1 && n_valid <= t->b, "bad argument");
  if (mode == GIGL_MODE_REPLACE)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "the training plan needs duplicate-free trees (no with-replacement mode)");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int k = t->cur;
  // (a prefetched workspace belongs to the roots it was announced with: an epoch cut short, a reshuffle or an evaluation
  // between two steps hands in other roots — the graph part is then issued again instead of training these labels
  // against the old batch's graph; as gigl_hgt_infer_run's guard)
  if (!t->fetched[k] || t->fetched_roots[k] != roots) {  // the graph part of THIS batch now (the layers wait for it)
    const int32_t rc = train_graph_part(t, k, roots, sampling_seed, mode);
    if (rc != GIGL_OK) return gigl_fail(ctx, rc, "%s", gigl_last_error(t->side[k]));
  }
  t->fetched[k] = false;
  // the NEXT batch's graph part goes to the side stream BEFORE this batch's layers are enqueued: it waits for what the
  // caller's stream holds now (roots_next was written there; the layers that last read the other workspace), not for
  // the layers about to be enqueued
  // (two batches ahead: the graph parts are chains of small latency-bound launches, two of them in flight on their own
  // streams take little more than one)
  const uint32_t* ahead[2] = {roots_next, roots_next ? roots_next2 : nullptr};
  for (int d = 0; d < 2; ++d) {
    const int kk = (k + 1 + d) % TRAIN_WS;
    if (!ahead[d] || (t->fetched[kk] && t->fetched_roots[kk] == ahead[d]) || kk == k) continue;
    const int32_t rc = train_graph_part(t, kk, ahead[d], sampling_seed, mode);
    if (rc != GIGL_OK) return gigl_fail(ctx, rc, "%s", gigl_last_error(t->side[kk]));
    t->fetched[kk] = true;
    t->fetched_roots[kk] = ahead[d];
  }
  // the step's inputs go into the static buffers the (captured) launches read: labels, the number of real roots (a
  // 32-bit fill: no host memory involved, ordered on the stream)
  if (t->fused_small) {
    TrainPrep pa{};
    pa.zero = (uint32_t*)t->zero_base;
    pa.zero_words = (int64_t)(t->zero_bytes / 4);
    for (int l = 1; l < t->L; ++l) {
      pa.w[pa.n_t] = t->w[l];
      pa.wt[pa.n_t] = t->wt_l[l];
      pa.rows[pa.n_t] = t->dims[l + 1];
      pa.cols[pa.n_t] = 2 * t->dims[l];
      ++pa.n_t;
    }
    pa.ticket = t->ticket;
    int64_t blocks = (pa.zero_words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 64) blocks = 64;
    // (the layers of the previous step read labels_buf / the cleared block: this launch is behind them on the stream)
    hipLaunchKernelGGL(train_stage_kernel, dim3((unsigned)blocks), dim3(256), 0, st, labels, n_valid, t->labels_buf, t->n_valid_buf,
                       t->loss_slot, loss_out, pa);
    GIGL_HIP_CHECK(ctx, hipGetLastError());
  } else {
    GIGL_HIP_CHECK(ctx, hipMemcpyAsync(t->labels_buf, labels, (size_t)n_valid * 8, hipMemcpyDeviceToDevice, st));
    GIGL_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)t->n_valid_buf, n_valid, 1, st));
  }
  GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, t->ev_graph[k], 0));
  if (t->lctx->stream != st || t->lctx->own_stream) {
    const int32_t rs = gigl_ctx_set_stream(t->lctx, st);
    if (rs != GIGL_OK) return gigl_fail(ctx, rs, "%s", gigl_last_error(t->lctx));
  }
  const int32_t rc = train_run_part(t->lctx, &t->exec_layers[k], &t->warm_layers, [&]() { return train_enqueue_layers(t, k); }, 1);
  if (rc != GIGL_OK) return gigl_fail(ctx, rc, "%s", gigl_last_error(t->lctx));
  GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_layers[k], st));
  t->cur = (k + 1) % TRAIN_WS;
  // (fused: the loss kernel wrote loss_out itself, through the pointer slot the stage kernel set)
  if (loss_out && !t->fused_small) GIGL_HIP_CHECK(ctx, hipMemcpyAsync(loss_out, t->loss, 4, hipMemcpyDeviceToDevice, st));
  return GIGL_OK;
}

const float* gigl_sage_train_plan_loss(gigl_sage_train_plan* t) { return t ? t->loss : nullptr; }

int32_t gigl_sage_train_plan_resume(gigl_sage_train_plan* t) {
  if (!t) return GIGL_E_INVALID_ARG;
  GIGL_HIP_CHECK(t->ctx, hipSetDevice(t->ctx->device));
  GIGL_HIP_CHECK(t->ctx, hipMemsetAsync(t->n_valid_buf + 2, 0, 4, t->ctx->stream));
  return GIGL_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// gigl_nablp_train_plan: one LINK-PREDICTION training step per call, all of it in the library (round 5) —
//   NodeAnchorBasedLinkPredictionModelingTaskSpec.train's loop body
//   (python/gigl/src/common/modeling_task_specs/node_anchor_based_link_prediction_modeling_task_spec.py:334-451):
//   infer_task_inputs (utils/infer.py: two encoder forwards — main batch, random-negative batch — query / positive /
//   random-negative embeddings by root index, inner-product scores of the repeated queries against cat(pos, rand_neg):
//   decoder.py:64-70), Retrieval.forward -> RetrievalLoss (loss.py:209-331: temperature, same-query and accidental-hit
//   masks, cross-entropy against the diagonal, summed, / rows), backward, Adam.
// with the reference's default encoder (GraphSAGE, mean, optional L2-normalised output).  Main batch = b anchors x
// (1 + P) rooted trees, anchor-major (anchor, its P positive slots; a missing positive repeats the anchor and is masked
// out by pos_cnt); random negatives = n_rn roots.  Everything is capacity-shaped (Q = b P query rows, C = Q + n_rn
// candidates, validity masks on the device): no host read, no torch kernel, one captured hipGraph per step.
struct gigl_nablp_train_plan {
  gigl_ctx* ctx = nullptr;
  gigl_ctx* lctx = nullptr;  // private ctx (own arena) bound to the caller's stream: every launch of the step
  struct Enc {
    gigl_sage_plan* base = nullptr;  // tree / union workspace of this encode's roots
    int32_t b = 0;
    int64_t rows_cap[GIGL_MAX_HOPS] = {0};
    float* a[GIGL_MAX_HOPS] = {nullptr};
    float* h[GIGL_MAX_HOPS] = {nullptr};
    float* dh[GIGL_MAX_HOPS] = {nullptr};
    float* gw[GIGL_MAX_HOPS] = {nullptr};
    float* gb[GIGL_MAX_HOPS] = {nullptr};
    // (round 6) the weight gradients' per-chunk partial sums: added up inside the Adam kernel, no reduce launch per layer
    float* part_w[GIGL_MAX_HOPS] = {nullptr};
    float* part_b[GIGL_MAX_HOPS] = {nullptr};
    int32_t part_rc[GIGL_MAX_HOPS] = {0};
    float* emb = nullptr;   // [b][d_out]: the roots' embeddings (normalised when the model says so)
    float* inv = nullptr;   // [b]: 1 / max(|h_r|, 1e-12) (1 without normalisation; 0: no such root)
    float* demb = nullptr;  // [b][d_out]
  } enc[2];                 // 0: main batch, 1: random negatives
  int32_t L = 0, b = 0, P = 0, n_rn = 0, normalize = 0, remove_hits = 1, act_last = 0;
  float temperature = 0.07f;
  int32_t dims[GIGL_MAX_HOPS + 1] = {0};
  float* w[GIGL_MAX_HOPS] = {nullptr};  // fused [dims[l+1]][2 dims[l]]: borrowed, UPDATED IN PLACE
  float* bias[GIGL_MAX_HOPS] = {nullptr};
  float* mom[4 * GIGL_MAX_HOPS] = {nullptr};
  float* da = nullptr;
  float* wt = nullptr;
  bool fused_small = false;   // (SAGE encoder, GIGL_TRAIN_PLAN_UNFUSED unset) partial sums inside Adam, W^T once per step
  float* wt_l[GIGL_MAX_HOPS] = {nullptr};
  bool bwd_gather = false;    // (as in gigl_sage_train_plan)
  void* zero_base = nullptr;  // both encodes' gw | gb | dh: cleared at the start of every step
  size_t zero_bytes = 0;
  // the head: repeated queries, candidates, ids, validity, scores and their gradients
  float *rq = nullptr, *cand = nullptr, *cand_t = nullptr, *scores = nullptr, *dscores = nullptr, *d_rq = nullptr, *d_cand = nullptr;
  int64_t *qid = nullptr, *cid = nullptr;
  int32_t* valid = nullptr;    // [C]: candidate column j takes part (rows i < Q use valid[i])
  int32_t* pos_cnt = nullptr;  // [b] static copy of the step's input
  int32_t* consts = nullptr;   // device {Q, C, Adam step, ...}
  float *row_lse = nullptr, *row_loss = nullptr, *loss = nullptr;  // loss[0] = the step's loss, loss[1] = valid query rows
  float lr = 5e-3f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, wd = 0.f;
  // ---- GAT encoder (kind == 1, gigl_gat_nablp_train_plan_create): two layers, the first from the INPUT side
  int kind = 0;
  struct Gat {
    int32_t heads = 1, c0 = 0, c1 = 0, d_in = 0;
    float slope = 0.2f;
    float *w[2] = {nullptr, nullptr}, *att_src[2] = {nullptr, nullptr}, *att_dst[2] = {nullptr, nullptr},
          *bias[2] = {nullptr, nullptr};  // borrowed, UPDATED IN PLACE
    float* g[8] = {nullptr};    // gradients: w0, att_src0, att_dst0, bias0, w1, att_src1, att_dst1, bias1 (both encodes ADD)
    float* mom[16] = {nullptr};  // Adam's m, v per parameter tensor
    int64_t n[8] = {0};
    float *u = nullptr, *du = nullptr;  // folded attention vectors [2H][d] and their gradient
    // forward state per encode: z [H][rows1][d], xw [rows1][c1], out_pre [b][c1]; backward scratch shared by both
    float *z[2] = {nullptr, nullptr}, *xw[2] = {nullptr, nullptr}, *out_pre[2] = {nullptr, nullptr};
    float *dxw = nullptr, *ds = nullptr, *dd = nullptr, *alpha = nullptr, *dh0 = nullptr, *dh0s = nullptr, *dz = nullptr,
          *edge_scratch = nullptr;
    // (round 6, as the SAGE plans: GIGL_TRAIN_PLAN_UNFUSED=1 for the A/B) the projections' weight gradients stay per-chunk partial
    // sums — per encode: W1's, and W0's / b0's per head — added up inside the Adam kernel; W1^T and the heads' W0^T are laid
    // out once per step
    bool fused = false;
    float* part_w1[2] = {nullptr, nullptr};
    int32_t rc_w1[2] = {0, 0};
    float* part_w0[2][4] = {{nullptr}};
    float* part_b0[2][4] = {{nullptr}};
    int32_t rc_w0[2] = {0, 0};
    float* wt1 = nullptr;
    float* wt0[4] = {nullptr};
    bool parts_pending = false;  // the gradient buffers do not hold the partial sums yet (gigl_gat_nablp_train_plan_grads adds them)
    // (fork: the random negatives' encode on a stream of its own) its OWN scratch and accumulators — sized by its rows —
    // so that nothing is shared with the main batch's encode while both run; folded together after the join
    struct Alt {
      float *alpha = nullptr, *dxw = nullptr, *ds = nullptr, *dd = nullptr, *dh0 = nullptr, *dh0s = nullptr, *dz = nullptr,
            *edge_scratch = nullptr, *du = nullptr;
      float* g[8] = {nullptr};
    } x;
  } gat;
  std::vector<void*> owned;
  // Two workspaces of trees + union graphs (as gigl_sage_train_plan): the graph part of the NEXT step's roots (sample +
  // union of both root sets: latency-bound launches) runs on a side stream beside this step's layers
  static constexpr int WS = 2;
  struct Work {
    gigl_ctx* side = nullptr;                    // private ctx with a stream of its own: the graph part's launches
    gigl_sage_plan* base[2] = {nullptr, nullptr};  // 0: main batch, 1: random negatives
    hipGraphExec_t exec_graph = nullptr, exec_layers = nullptr;
    bool warm_graph = false, warm_layers = false;
    hipEvent_t ev_graph = nullptr;   // end of the graph part last enqueued for this workspace
    hipEvent_t ev_layers = nullptr;  // end of the layers part that last read it
    int32_t* tlists[2][GIGL_MAX_HOPS] = {{nullptr}};  // transposed lists of layers >= 1 per encode (SAGE; built by the graph part)
    const uint32_t *fetched_main = nullptr, *fetched_rn = nullptr;  // the caller's buffers its graph part ran for
    bool fetched = false;
  } work[WS];
  hipEvent_t ev_now = nullptr;  // "the caller's stream, now": the roots a graph part copies were written before it
  // (round 6; GIGL_LP_FORK=0 turns it off) the random negatives' encode — ~15 launches of a 512-root batch, all latency — runs
  // on a stream of its own beside the main batch's, forward and backward: forked after the step's shared preparation, joined
  // before the scores, forked again after the loss's backward, joined before Adam.  Its only shared scratch is `da`.
  gigl_ctx* actx = nullptr;
  float* da2 = nullptr;
  hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr};
  bool fork = false;
  // ... and the MAIN batch's weight gradients on a third: nothing in the backward chain waits for them (Adam does), each is a
  // 40-65 us matrix kernel between the chain's latency-bound ones
  gigl_ctx* wctx = nullptr;
  hipEvent_t ev_w = nullptr, ev_wjoin = nullptr;
  int cur = 0;
  int cur_layers_ws = 0;  // the workspace the layers part being enqueued reads
  int32_t cap_seed = 0, cap_mode = -1;
};

namespace {

// e[r] = h[root_local[r]] (x 1 / max(|.|, 1e-12) when normalising: torch.nn.functional.normalize), one wave per root
__global__ __launch_bounds__(256) void lp_take_norm_kernel(const float* __restrict__ h, const int32_t* __restrict__ root_local,
                                                           int b, int d, int normalize, const int32_t* __restrict__ meta,
                                                           float* __restrict__ emb, float* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= b) return;
  const bool failed = meta[GIGL_META_OVERFLOW] != 0;
  const int32_t l = failed ? -1 : root_local[r];
  float* o = emb + (int64_t)r * d;
  if (l < 0) {
    const float v = failed ? __builtin_nanf("") : 0.f;
    for (int c = lane; c < d; c += 64) o[c] = v;
    if (lane == 0) inv[r] = 0.f;
    return;
  }
  const float* row = h + (int64_t)l * d;
  float ss = 0.f;
  for (int c = lane; c < d; c += 64) ss += row[c] * row[c];
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
  const float iv = normalize ? 1.f / fmaxf(sqrtf(ss), 1e-12f) : 1.f;
  for (int c = lane; c < d; c += 64) o[c] = row[c] * iv;
  if (lane == 0) inv[r] = iv;
}

// the head's operands from the two encodes: row q = (anchor i, positive slot j) of rq = the anchor's embedding, row q of
// cand = that positive's; rows Q.. of cand = the random negatives'; ids for the loss masks; validity of every candidate
__global__ __launch_bounds__(256) void lp_pack_kernel(const float* __restrict__ e_main, const float* __restrict__ e_rn,
                                                      const uint32_t* __restrict__ roots_main,
                                                      const uint32_t* __restrict__ roots_rn,
                                                      const int32_t* __restrict__ pos_cnt, const float* __restrict__ inv_main,
                                                      const float* __restrict__ inv_rn, int b, int P, int n_rn, int d,
                                                      float* __restrict__ rq, float* __restrict__ cand,
                                                      int64_t* __restrict__ qid, int64_t* __restrict__ cid,
                                                      int32_t* __restrict__ valid) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int Q = b * P, T = 1 + P;
  if (row >= Q + n_rn) return;
  if (row < Q) {
    const int i = row / P, j = row - i * P;
    const int ra = i * T, rp = ra + 1 + j;
    const uint32_t aid = roots_main[ra], pid = roots_main[rp];
    const bool ok = aid != GIGL_INVALID && pid != GIGL_INVALID && j < pos_cnt[i] && inv_main[ra] != 0.f && inv_main[rp] != 0.f;
    const float* qa = e_main + (int64_t)ra * d;
    const float* pa = e_main + (int64_t)rp * d;
    for (int c = lane; c < d; c += 64) {
      rq[(int64_t)row * d + c] = ok ? qa[c] : 0.f;
      cand[(int64_t)row * d + c] = ok ? pa[c] : 0.f;
    }
    if (lane == 0) {
      qid[row] = (int64_t)aid;
      cid[row] = (int64_t)pid;
      valid[row] = ok ? 1 : 0;
    }
  } else {
    const int r = row - Q;
    const uint32_t id = roots_rn[r];
    const bool ok = id != GIGL_INVALID && inv_rn[r] != 0.f;
    const float* ea = e_rn + (int64_t)r * d;
    for (int c = lane; c < d; c += 64) cand[(int64_t)row * d + c] = ok ? ea[c] : 0.f;
    if (lane == 0) {
      cid[row] = (int64_t)id;
      valid[row] = ok ? 1 : 0;
    }
  }
}

__device__ __forceinline__ bool lp_excluded(const int64_t* qid, const int64_t* cid, const int32_t* valid, int remove_hits,
                                            int Q, int i, int j, int64_t qid_i, int64_t cid_i) {
  if (!valid[j]) return true;
  if (j == i) return false;
  if (j < Q && qid[j] == qid_i) return true;          // another positive of the same query (loss.py:279-305)
  return remove_hits && cid[j] == cid_i;              // the positive itself among the other candidates (:307-331)
}

// retrieval loss rows (loss.hip's retrieval_rows_kernel with a validity mask): one workgroup per query row
__global__ __launch_bounds__(256) void lp_loss_rows_kernel(const float* __restrict__ scores, int Q, int Cn, float temperature,
                                                           const int64_t* __restrict__ qid, const int64_t* __restrict__ cid,
                                                           const int32_t* __restrict__ valid, int remove_hits,
                                                           float* __restrict__ row_lse, float* __restrict__ row_loss) {
  __shared__ float s_m[4], s_s[4];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (!valid[i]) {
    if (tid == 0) {
      row_lse[i] = 0.f;
      row_loss[i] = 0.f;
    }
    return;
  }
  const float* row = scores + (int64_t)i * Cn;
  const int64_t qid_i = qid[i], cid_i = cid[i];
  const float it = temperature > 0.f ? 1.f / temperature : 1.f;
  float m = -INFINITY, s = 0.f;
  for (int j = tid; j < Cn; j += 256) {
    if (lp_excluded(qid, cid, valid, remove_hits, Q, i, j, qid_i, cid_i)) continue;
    const float v = temperature > 0.f ? row[j] / temperature : row[j];
    if (v > m) {
      s = s * expf(m - v) + 1.f;
      m = v;
    } else {
      s += expf(v - m);
    }
  }
  (void)it;
  for (int off = 32; off > 0; off >>= 1) {
    const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
    const float mm = fmaxf(m, m2);
    if (mm != -INFINITY) {
      s = s * expf(m - mm) + s2 * expf(m2 - mm);
      m = mm;
    }
  }
  if (lane == 0) {
    s_m[w] = m;
    s_s[w] = s;
  }
  __syncthreads();
  if (tid == 0) {
    float mm = s_m[0], ss = s_s[0];
    for (int k = 1; k < 4; ++k) {
      const float m2 = s_m[k], s2 = s_s[k], mx = fmaxf(mm, m2);
      if (mx != -INFINITY) {
        ss = ss * expf(mm - mx) + s2 * expf(m2 - mx);
        mm = mx;
      }
    }
    const float lse = mm + logf(ss);
    row_lse[i] = lse;
    row_loss[i] = lse - (temperature > 0.f ? row[i] / temperature : row[i]);
  }
}

// loss[0] = sum of the valid rows' losses / their number (fixed order, in double), loss[1] = that number; Adam's step
// counter moves when the step will be applied
__global__ __launch_bounds__(1024) void lp_loss_sum_kernel(const float* __restrict__ row_loss, const int32_t* __restrict__ valid,
                                                           int Q, float* __restrict__ loss, int32_t* __restrict__ step,
                                                           const int32_t* __restrict__ meta_a, const int32_t* __restrict__ meta_b) {
  __shared__ double s_w[16];
  __shared__ int s_n[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double acc = 0.0;
  int n = 0;
  for (int i = tid; i < Q; i += 1024) {
    acc += (double)row_loss[i];
    n += valid[i] ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_xor(acc, off, 64);
    n += __shfl_xor(n, off, 64);
  }
  if (lane == 0) {
    s_w[w] = acc;
    s_n[w] = n;
  }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    int nn = 0;
    for (int k = 0; k < 16; ++k) {
      t += s_w[k];
      nn += s_n[k];
    }
    const bool failed = meta_a[GIGL_META_OVERFLOW] != 0 || meta_b[GIGL_META_OVERFLOW] != 0;
    loss[0] = failed ? __builtin_nanf("") : (float)(t / (double)(nn > 0 ? nn : 1));
    loss[1] = (float)nn;
    if (!failed) *step += 1;
  }
}

// dscores = d loss / d scores: (softmax - onehot) / temperature / rows for the columns that take part, 0 elsewhere
__global__ __launch_bounds__(256) void lp_loss_backward_kernel(const float* __restrict__ scores, int Q, int Cn, float temperature,
                                                               const int64_t* __restrict__ qid, const int64_t* __restrict__ cid,
                                                               const int32_t* __restrict__ valid, int remove_hits,
                                                               const float* __restrict__ row_lse, const float* __restrict__ loss,
                                                               float* __restrict__ dscores) {
  const int i = blockIdx.x;
  float* out = dscores + (int64_t)i * Cn;
  if (!valid[i]) {
    for (int j = threadIdx.x; j < Cn; j += 256) out[j] = 0.f;
    return;
  }
  const float* row = scores + (int64_t)i * Cn;
  const int64_t qid_i = qid[i], cid_i = cid[i];
  const float lse = row_lse[i];
  const float nrows = loss[1] > 0.f ? loss[1] : 1.f;
  const float scale = (temperature > 0.f ? 1.f / temperature : 1.f) / nrows;
  for (int j = threadIdx.x; j < Cn; j += 256) {
    float d = 0.f;
    if (!lp_excluded(qid, cid, valid, remove_hits, Q, i, j, qid_i, cid_i)) {
      const float v = temperature > 0.f ? row[j] / temperature : row[j];
      d = (expf(v - lse) - (j == i ? 1.f : 0.f)) * scale;
    }
    out[j] = d;
  }
}

// d emb of the two encodes from d rq / d cand: an anchor's row sums its P query rows, a positive's row is its candidate
// row, a random negative's its own; then through the normalisation (dx = (g - y <y, g>) / max(|x|, eps)) and added to the
// last layer's output gradient at the root's local row (roots that are the same node share it: atomics).  One wave per root.
__global__ __launch_bounds__(256) void lp_unpack_scatter_kernel(const float* __restrict__ d_rq, const float* __restrict__ d_cand,
                                                                const int32_t* __restrict__ valid, const float* __restrict__ emb,
                                                                const float* __restrict__ inv,
                                                                const int32_t* __restrict__ root_local, int which, int b, int P,
                                                                int n_rn, int d, int normalize, float* __restrict__ dh_last) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int T = 1 + P, Q = b * P;
  const int n_roots = which == 0 ? b * T : n_rn;
  if (r >= n_roots) return;
  const float iv = inv[r];
  if (iv == 0.f) return;
  const int32_t l = root_local[r];
  if (l < 0) return;
  const float* y = emb + (int64_t)r * d;
  float dot = 0.f;
  // (d <= 64 * 8: a lane keeps up to 8 columns)
  float g[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = lane + 64 * u;
    float v = 0.f;
    if (c < d) {
      if (which == 1) {
        v = valid[Q + r] ? d_cand[(int64_t)(Q + r) * d + c] : 0.f;
      } else {
        const int i = r / T, t = r - i * T;
        if (t == 0) {
          for (int j = 0; j < P; ++j)
            if (valid[i * P + j]) v += d_rq[(int64_t)(i * P + j) * d + c];
        } else if (valid[i * P + t - 1]) {
          v = d_cand[(int64_t)(i * P + t - 1) * d + c];
        }
      }
      dot += v * y[c];
    }
    g[u] = v;
  }
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = lane + 64 * u;
    if (c < d) {
      const float dx = normalize ? (g[u] - y[c] * dot) * iv : g[u];
      if (dx != 0.f) atomicAdd(&dh_last[(int64_t)l * d + c], dx);
    }
  }
}

// Adam over the SUM of the two encodes' gradients (both forwards share the weights)
constexpr int ADAM2_MAX = 16;  // tensors (or slices of tensors: the GAT plan's heads) per launch
struct AdamPack2 {
  float* p[ADAM2_MAX];
  const float* g1[ADAM2_MAX];
  const float* g2[ADAM2_MAX];
  float* m[ADAM2_MAX];
  float* v[ADAM2_MAX];
  int64_t n[ADAM2_MAX];
  // (round 6) part1 / part2 != NULL: the two encodes' gradients as per-chunk partial sums ([chunks][n], the chunks below
  // ceil(*rows / rc) real), each added up in chunk order, then the two totals — the order of reduce + reduce + add; g1 may
  // hold a further term beside them (the GAT plan: the folded attention vectors' share of W0's gradient)
  const float* part1[ADAM2_MAX];
  const float* part2[ADAM2_MAX];
  const int32_t* rows1[ADAM2_MAX];
  const int32_t* rows2[ADAM2_MAX];
  int32_t rc1[ADAM2_MAX], rc2[ADAM2_MAX];
  int32_t count;
  float lr, beta1, beta2, eps, wd;
};

__device__ __forceinline__ float chunk_sum(const float* __restrict__ part, int chunks, int64_t n, int64_t i) {
  float s = 0.f;
  int c = 0;
  for (; c + 15 < chunks; c += 16) {
    float pv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) pv[q] = part[(int64_t)(c + q) * n + i];
#pragma unroll
    for (int q = 0; q < 16; ++q) s += pv[q];
  }
  for (; c < chunks; ++c) s += part[(int64_t)c * n + i];
  return s;
}

__global__ __launch_bounds__(256) void lp_adam_kernel(AdamPack2 a, const int32_t* __restrict__ step_dev,
                                                      const int32_t* __restrict__ meta_a, const int32_t* __restrict__ meta_b) {
  if (meta_a[GIGL_META_OVERFLOW] != 0 || meta_b[GIGL_META_OVERFLOW] != 0) return;  // a failed batch trains nothing
  const double t = (double)*step_dev;
  const float bc1 = (float)(1.0 - pow((double)a.beta1, t)), bc2s = (float)sqrt(1.0 - pow((double)a.beta2, t));
  const float step_size = a.lr / bc1;
  const int k_lo = gridDim.y > 1 ? (int)blockIdx.y : 0, k_hi = gridDim.y > 1 ? (int)blockIdx.y + 1 : a.count;  // (a grid slice per tensor)
  for (int k = k_lo; k < k_hi && k < a.count; ++k) {
    float* p = a.p[k];
    float* m = a.m[k];
    float* v = a.v[k];
    const int ch1 = a.part1[k] ? (*a.rows1[k] + a.rc1[k] - 1) / a.rc1[k] : 0;
    const int ch2 = a.part2[k] ? (*a.rows2[k] + a.rc2[k] - 1) / a.rc2[k] : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n[k]; i += (int64_t)gridDim.x * blockDim.x) {
      const float w = p[i];
      float ga = a.part1[k] ? chunk_sum(a.part1[k], ch1, a.n[k], i) : a.g1[k][i];
      const float gb_ = a.part2[k] ? chunk_sum(a.part2[k], ch2, a.n[k], i) : (a.g2[k] ? a.g2[k][i] : 0.f);
      if (a.part1[k] && a.g1[k]) ga += a.g1[k][i];
      const float gr = (ga + gb_) + a.wd * w;
      const float mm = m[i] + (gr - m[i]) * (1.f - a.beta1);
      const float vv = v[i] * a.beta2 + (1.f - a.beta2) * gr * gr;
      m[i] = mm;
      v[i] = vv;
      p[i] = w - step_size * (mm / (sqrtf(vv) / bc2s + a.eps));
    }
  }
}

int32_t lp_forward(gigl_nablp_train_plan* t, int which) {
  gigl_nablp_train_plan::Enc& e = t->enc[which];
  gigl_sage_plan* p = e.base;
  gigl_ctx* ctx = which == 1 && t->fork ? t->actx : t->lctx;
  const int L = t->L;
  int32_t rc = GIGL_OK;
  const int32_t* n_local = p->leaf_global ? (L >= 2 ? p->un.meta + GIGL_META_LEVEL0 + (L - 2) : p->zero_dev) : nullptr;
  for (int l = 0; l < L; ++l) {
    const int32_t* n_rows = p->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    const int d = t->dims[l];
    if (l == 0)
      rc = gigl_gather_reduce_mixed(ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, p->un.rowptr, p->un.rowend, p->un.col,
                                    n_rows, e.rows_cap[0], GIGL_AGGR_MEAN, n_local, e.a[0]);
    else
      rc = gigl_gather_reduce(ctx, e.h[l - 1], GIGL_DTYPE_F32, d, nullptr, p->un.rowptr, p->un.rowend, p->un.col, n_rows,
                              e.rows_cap[l], GIGL_AGGR_MEAN, e.a[l]);
    if (rc != GIGL_OK) return rc;
    rc = gigl_linear(ctx, e.a[l], t->w[l], t->bias[l], n_rows, e.rows_cap[l], 2 * d, t->dims[l + 1],
                     (l < L - 1 || t->act_last) ? 1 : 0, e.h[l]);
    if (rc != GIGL_OK) return rc;
  }
  const int dout = t->dims[L];
  hipLaunchKernelGGL(lp_take_norm_kernel, dim3((unsigned)((e.b + 3) / 4)), dim3(256), 0, ctx->stream, (const float*)e.h[L - 1],
                     (const int32_t*)p->un.root_local, e.b, dout, t->normalize, (const int32_t*)p->un.meta, e.emb, e.inv);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t lp_backward(gigl_nablp_train_plan* t, int which) {
  gigl_nablp_train_plan::Enc& e = t->enc[which];
  gigl_sage_plan* p = e.base;
  gigl_ctx* ctx = which == 1 && t->fork ? t->actx : t->lctx;
  float* da = which == 1 && t->fork ? t->da2 : t->da;
  hipStream_t st = ctx->stream;
  const int L = t->L;
  int32_t rc = GIGL_OK;
  for (int l = L - 1; l >= 0; --l) {
    const int32_t* n_rows = p->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    const int d = t->dims[l], n_out = t->dims[l + 1];
    const bool act = l < L - 1 || t->act_last;
    const bool side_w = which == 0 && t->fork && t->wctx;  // (this gradient on the third stream: see wctx)
    if (side_w) {
      // the mask first (the chain needs the masked rows anyway), then the gradient of the MASKED rows beside the chain: the
      // same products as the masked read of the unmasked rows
      if (act) {
        int64_t blocks = (e.rows_cap[l] * n_out + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, st, e.dh[l], (const float*)e.h[l], n_rows, n_out);
      }
      GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_w, st));
      GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(t->wctx->stream, t->ev_w, 0));
      rc = gigl_linear_weight_grad_parts(t->wctx, e.dh[l], e.a[l], nullptr, n_rows, e.rows_cap[l], n_out, 2 * d, e.part_w[l],
                                         t->bias[l] ? e.part_b[l] : nullptr);
    } else if (t->fused_small)
      rc = gigl_linear_weight_grad_parts(ctx, e.dh[l], e.a[l], act ? e.h[l] : nullptr, n_rows, e.rows_cap[l], n_out, 2 * d,
                                         e.part_w[l], t->bias[l] ? e.part_b[l] : nullptr);
    else
      rc = gigl_linear_weight_grad(ctx, e.dh[l], e.a[l], act ? e.h[l] : nullptr, n_rows, e.rows_cap[l], n_out, 2 * d, e.gw[l],
                                   t->bias[l] ? e.gb[l] : nullptr);
    if (rc != GIGL_OK) return rc;
    if (l == 0) break;
    if (act && !side_w) {
      int64_t blocks = (e.rows_cap[l] * n_out + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, st, e.dh[l], (const float*)e.h[l], n_rows, n_out);
    }
    const float* wt = t->wt;
    if (t->fused_small) {
      wt = t->wt_l[l];  // (transposed ONCE per step, at the start of the layers: both encodes read it)
    } else {
      int64_t blocks = ((int64_t)n_out * 2 * d + 255) / 256;
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)t->w[l], n_out, 2 * d, t->wt);
    }
    rc = gigl_linear(ctx, e.dh[l], wt, nullptr, n_rows, e.rows_cap[l], n_out, 2 * d, 0, da);
    if (rc != GIGL_OK) return rc;
    if (t->bwd_gather)
      rc = gigl_gather_mean_backward_lists(ctx, da, d, p->un.rowptr, p->un.rowend, n_rows,
                                           p->un.meta + GIGL_META_LEVEL0 + (L - l), e.rows_cap[l - 1],
                                           t->work[t->cur_layers_ws].tlists[which][l], GIGL_AGGR_MEAN, e.dh[l - 1]);
    else
      rc = gigl_gather_mean_backward(ctx, da, d, p->un.rowptr, p->un.rowend, p->un.col, n_rows, e.rows_cap[l], e.dh[l - 1]);
    if (rc != GIGL_OK) return rc;
  }
  return GIGL_OK;
}

// every launch of a step, on lctx's stream (the caller's)
int32_t gat_lp_begin(gigl_nablp_train_plan* t);
int32_t gat_lp_forward(gigl_nablp_train_plan* t, int which);
int32_t gat_lp_backward(gigl_nablp_train_plan* t, int which);
int32_t gat_lp_finish(gigl_nablp_train_plan* t);

// sample + union of both root sets of workspace w, on its side stream
int32_t lp_enqueue_graph(gigl_nablp_train_plan* t, int w, int32_t sampling_seed, int32_t mode) {
  for (int k = 0; k < 2; ++k) {
    gigl_sage_plan* p = t->work[w].base[k];
    int32_t rc = enqueue_range(p, 0, 2, p->roots_buf, sampling_seed, mode, nullptr);
    // (SAGE: who reads which source row in the backward of layers >= 1 — a function of the batch graph alone)
    for (int l = 1; l < t->L && rc == GIGL_OK && t->kind == 0 && t->bwd_gather; ++l)
      rc = gigl_transposed_rows_build(p->ctx, p->un.rowptr, p->un.rowend, p->un.col, p->un.meta + GIGL_META_LEVEL0 + (t->L - 1 - l),
                                      t->enc[k].rows_cap[l], p->un.meta + GIGL_META_LEVEL0 + (t->L - l),
                                      t->enc[k].rows_cap[l - 1], p->un.cap_edges, t->work[w].tlists[k][l]);
    if (rc != GIGL_OK) return rc;
  }
  return GIGL_OK;
}

// everything after the graph part, over workspace w's trees and union graphs, on lctx's stream (the caller's)
int32_t lp_enqueue_layers(gigl_nablp_train_plan* t, int w) {
  gigl_ctx* ctx = t->lctx;
  hipStream_t st = ctx->stream;
  const int L = t->L, d = t->dims[L], Q = t->b * t->P, Cn = Q + t->n_rn;
  int32_t rc = GIGL_OK;
  for (int k = 0; k < 2; ++k) t->enc[k].base = t->work[w].base[k];
  t->cur_layers_ws = w;
  gigl_fill_u32(st, t->zero_base, 0u, (int64_t)(t->zero_bytes / 4));
  if (t->kind == 1) {
    rc = gat_lp_begin(t);
    if (rc != GIGL_OK) return rc;
  }
  for (int l = 1; l < L && t->fused_small; ++l) {  // W_l^T for the input gradients of both encodes
    const int64_t nw = (int64_t)t->dims[l + 1] * 2 * t->dims[l];
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, (const float*)t->w[l],
                       t->dims[l + 1], 2 * t->dims[l], t->wt_l[l]);
  }
  if (t->fork) {
    GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_fork[0], st));
    GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(t->actx->stream, t->ev_fork[0], 0));
  }
  for (int k = 0; k < 2; ++k) {
    rc = t->kind == 1 ? gat_lp_forward(t, k) : lp_forward(t, k);
    if (rc != GIGL_OK) return rc;
  }
  if (t->fork) {
    GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_join[0], t->actx->stream));
    GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, t->ev_join[0], 0));
  }
  const gigl_sage_plan *pm = t->enc[0].base, *pr = t->enc[1].base;
  hipLaunchKernelGGL(lp_pack_kernel, dim3((unsigned)((Cn + 3) / 4)), dim3(256), 0, st, (const float*)t->enc[0].emb,
                     (const float*)t->enc[1].emb, (const uint32_t*)pm->roots_buf, (const uint32_t*)pr->roots_buf,
                     (const int32_t*)t->pos_cnt, (const float*)t->enc[0].inv, (const float*)t->enc[1].inv, t->b, t->P, t->n_rn,
                     d, t->rq, t->cand, t->qid, t->cid, t->valid);
  // scores[q][c] = <rq[q], cand[c]>  (decoder.py:64-70: torch.mm(q, c.T))
  rc = gigl_linear(ctx, t->rq, t->cand, nullptr, t->consts + 0, Q, d, Cn, 0, t->scores);
  if (rc != GIGL_OK) return rc;
  hipLaunchKernelGGL(lp_loss_rows_kernel, dim3((unsigned)Q), dim3(256), 0, st, (const float*)t->scores, Q, Cn, t->temperature,
                     (const int64_t*)t->qid, (const int64_t*)t->cid, (const int32_t*)t->valid, t->remove_hits, t->row_lse,
                     t->row_loss);
  hipLaunchKernelGGL(lp_loss_sum_kernel, dim3(1), dim3(1024), 0, st, (const float*)t->row_loss, (const int32_t*)t->valid, Q,
                     t->loss, t->consts + 2, (const int32_t*)pm->un.meta, (const int32_t*)pr->un.meta);
  hipLaunchKernelGGL(lp_loss_backward_kernel, dim3((unsigned)Q), dim3(256), 0, st, (const float*)t->scores, Q, Cn,
                     t->temperature, (const int64_t*)t->qid, (const int64_t*)t->cid, (const int32_t*)t->valid, t->remove_hits,
                     (const float*)t->row_lse, (const float*)t->loss, t->dscores);
  // d rq = dscores . cand ;  d cand = dscores^T . rq
  {
    int64_t blocks = ((int64_t)Cn * d + 255) / 256;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)t->cand, Cn, d, t->cand_t);
  }
  rc = gigl_linear(ctx, t->dscores, t->cand_t, nullptr, t->consts + 0, Q, Cn, d, 0, t->d_rq);
  if (rc != GIGL_OK) return rc;
  gigl_fill_u32(st, t->d_cand, 0u, (int64_t)Cn * d);  // (gigl_linear_weight_grad ADDS its chunks' sums to dw)
  rc = gigl_linear_weight_grad(ctx, t->dscores, t->rq, nullptr, t->consts + 0, Q, Cn, d, t->d_cand, nullptr);
  if (rc != GIGL_OK) return rc;
  for (int k = 0; k < 2; ++k) {
    const gigl_nablp_train_plan::Enc& e = t->enc[k];
    hipLaunchKernelGGL(lp_unpack_scatter_kernel, dim3((unsigned)((e.b + 3) / 4)), dim3(256), 0, st, (const float*)t->d_rq,
                       (const float*)t->d_cand, (const int32_t*)t->valid, (const float*)e.emb, (const float*)e.inv,
                       (const int32_t*)e.base->un.root_local, k, t->b, t->P, t->n_rn, d, t->normalize, e.dh[L - 1]);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  if (t->fork) {
    GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_fork[1], st));
    GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(t->actx->stream, t->ev_fork[1], 0));
  }
  for (int k = 0; k < 2; ++k) {
    rc = t->kind == 1 ? gat_lp_backward(t, k) : lp_backward(t, k);
    if (rc != GIGL_OK) return rc;
  }
  if (t->fork) {
    GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_join[1], t->actx->stream));
    GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, t->ev_join[1], 0));
    if (t->wctx) {
      GIGL_HIP_CHECK(ctx, hipEventRecord(t->ev_wjoin, t->wctx->stream));
      GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, t->ev_wjoin, 0));
    }
  }
  if (t->kind == 1) return gat_lp_finish(t);
  AdamPack2 ap{};
  for (int l = 0; l < L; ++l) {
    const bool fz = t->fused_small;
    const int32_t* r1 = pm->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    const int32_t* r2 = pr->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    ap.p[ap.count] = t->w[l];
    ap.g1[ap.count] = fz ? nullptr : t->enc[0].gw[l];
    ap.g2[ap.count] = fz ? nullptr : t->enc[1].gw[l];
    ap.part1[ap.count] = fz ? t->enc[0].part_w[l] : nullptr;
    ap.part2[ap.count] = fz ? t->enc[1].part_w[l] : nullptr;
    ap.rows1[ap.count] = r1;
    ap.rows2[ap.count] = r2;
    ap.rc1[ap.count] = t->enc[0].part_rc[l];
    ap.rc2[ap.count] = t->enc[1].part_rc[l];
    ap.m[ap.count] = t->mom[4 * l];
    ap.v[ap.count] = t->mom[4 * l + 1];
    ap.n[ap.count++] = (int64_t)t->dims[l + 1] * 2 * t->dims[l];
    if (t->bias[l]) {
      ap.p[ap.count] = t->bias[l];
      ap.g1[ap.count] = fz ? nullptr : t->enc[0].gb[l];
      ap.g2[ap.count] = fz ? nullptr : t->enc[1].gb[l];
      ap.part1[ap.count] = fz ? t->enc[0].part_b[l] : nullptr;
      ap.part2[ap.count] = fz ? t->enc[1].part_b[l] : nullptr;
      ap.rows1[ap.count] = r1;
      ap.rows2[ap.count] = r2;
      ap.rc1[ap.count] = t->enc[0].part_rc[l];
      ap.rc2[ap.count] = t->enc[1].part_rc[l];
      ap.m[ap.count] = t->mom[4 * l + 2];
      ap.v[ap.count] = t->mom[4 * l + 3];
      ap.n[ap.count++] = t->dims[l + 1];
    }
  }
  ap.lr = t->lr;
  ap.beta1 = t->beta1;
  ap.beta2 = t->beta2;
  ap.eps = t->eps;
  ap.wd = t->wd;
  hipLaunchKernelGGL(lp_adam_kernel, t->fused_small ? dim3(208, (unsigned)ap.count) : dim3(256), dim3(256), 0, st, ap,
                     (const int32_t*)(t->consts + 2),
                     (const int32_t*)pm->un.meta, (const int32_t*)pr->un.meta);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // namespace

int32_t gigl_nablp_train_plan_destroy(gigl_nablp_train_plan* t) {
  if (!t) return GIGL_OK;
  if (t->ctx) {
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
  }
  for (auto& wk : t->work)
    if (wk.side) hipStreamSynchronize(wk.side->stream);
  for (auto& wk : t->work) {
    if (wk.exec_graph) hipGraphExecDestroy(wk.exec_graph);
    if (wk.exec_layers) hipGraphExecDestroy(wk.exec_layers);
    if (wk.ev_graph) hipEventDestroy(wk.ev_graph);
    if (wk.ev_layers) hipEventDestroy(wk.ev_layers);
    for (int k = 0; k < 2; ++k)
      if (wk.base[k]) gigl_sage_plan_destroy(wk.base[k]);
  }
  if (t->ev_now) hipEventDestroy(t->ev_now);
  if (t->actx) hipStreamSynchronize(t->actx->stream);
  if (t->wctx) hipStreamSynchronize(t->wctx->stream);
  if (t->ev_w) hipEventDestroy(t->ev_w);
  if (t->ev_wjoin) hipEventDestroy(t->ev_wjoin);
  for (int i = 0; i < 2; ++i) {
    if (t->ev_fork[i]) hipEventDestroy(t->ev_fork[i]);
    if (t->ev_join[i]) hipEventDestroy(t->ev_join[i]);
  }
  for (void* q : t->owned) hipFree(q);
  for (auto& wk : t->work)
    if (wk.side) gigl_ctx_destroy(wk.side);
  if (t->actx) gigl_ctx_destroy(t->actx);
  if (t->wctx) gigl_ctx_destroy(t->wctx);
  if (t->lctx) {
    gigl_ctx_set_stream(t->lctx, nullptr);  // (the stream is the caller's)
    gigl_ctx_destroy(t->lctx);
  }
  delete t;
  return GIGL_OK;
}

int32_t gigl_nablp_train_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b_anchors,
                                     int32_t num_positives, int32_t n_random_negatives, const int32_t* fanouts, int32_t hops,
                                     const int32_t* dims, float* const* w, float* const* bias, int32_t act_last,
                                     int32_t l2_normalize, float temperature, int32_t remove_accidental_hits, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, gigl_nablp_train_plan** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, graph && feat && fanouts && dims && w, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS && b_anchors >= 1 && num_positives >= 1 && n_random_negatives >= 0,
               "bad plan shape");
  GIGL_REQUIRE(ctx, dims[hops] <= 512, "embedding width %d: the head keeps a row in 8 registers per lane (<= 512)", dims[hops]);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_nablp_train_plan* t = new (std::nothrow) gigl_nablp_train_plan();
  if (!t) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  t->ctx = ctx;
  t->L = hops;
  t->b = b_anchors;
  t->P = num_positives;
  t->n_rn = n_random_negatives;
  t->normalize = l2_normalize ? 1 : 0;
  t->remove_hits = remove_accidental_hits ? 1 : 0;
  t->temperature = temperature;
  t->act_last = act_last;
  t->lr = lr;
  t->beta1 = beta1;
  t->beta2 = beta2;
  t->eps = eps;
  t->wd = weight_decay;
  int32_t rc = gigl_ctx_create(ctx->device, &t->lctx);
  const int32_t nb[2] = {b_anchors * (1 + num_positives), n_random_negatives > 0 ? n_random_negatives : 1};
  for (int wi = 0; wi < gigl_nablp_train_plan::WS && rc == GIGL_OK; ++wi) {
    gigl_nablp_train_plan::Work& wk = t->work[wi];
    rc = gigl_ctx_create(ctx->device, &wk.side);
    if (rc == GIGL_OK) wk.side->wide = ctx->wide;
    for (int k = 0; k < 2 && rc == GIGL_OK; ++k) {
      t->enc[k].b = nb[k];
      rc = plan_create(wk.side, graph, feat, nb[k], fanouts, hops, dims, (const float* const*)w, (const float* const*)bias,
                       act_last, false, &wk.base[k]);
      if (rc != GIGL_OK) gigl_fail(ctx, rc, "%s", gigl_last_error(wk.side));
    }
    if (rc == GIGL_OK && (hipEventCreateWithFlags(&wk.ev_graph, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&wk.ev_layers, hipEventDisableTiming) != hipSuccess))
      rc = GIGL_E_HIP;
  }
  if (rc == GIGL_OK && hipEventCreateWithFlags(&t->ev_now, hipEventDisableTiming) != hipSuccess) rc = GIGL_E_HIP;
  if (rc == GIGL_OK)
    for (int k = 0; k < 2; ++k) t->enc[k].base = t->work[0].base[k];
  if (rc != GIGL_OK) {
    gigl_nablp_train_plan_destroy(t);
    return rc;
  }
  auto alloc = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
    t->owned.push_back(q);
    return q;
  };
  bool ok = true;
  size_t zero_floats = 0, da_floats = 16, wt_floats = 16;
  for (int l = 0; l <= hops; ++l) t->dims[l] = dims[l];
  t->bwd_gather = train_bwd_gather(dims, hops);
  for (int l = 0; l < hops; ++l) {
    t->w[l] = w[l];
    t->bias[l] = bias ? bias[l] : nullptr;
    const size_t nw = (size_t)dims[l + 1] * 2 * dims[l];
    wt_floats = std::max(wt_floats, nw);
    for (int k = 0; k < 4; ++k) {
      const size_t n = k < 2 ? nw : (size_t)dims[l + 1];
      t->mom[4 * l + k] = (float*)alloc(n * 4);
      if (t->mom[4 * l + k] && hipMemset(t->mom[4 * l + k], 0, n * 4) != hipSuccess) ok = false;
      ok = ok && t->mom[4 * l + k];
    }
    for (int k = 0; k < 2; ++k) {
      gigl_nablp_train_plan::Enc& e = t->enc[k];
      const int64_t rows = gigl_level_rows(ctx->wide, e.b, fanouts, hops, hops - 1 - l);
      e.rows_cap[l] = rows;
      zero_floats += nw + dims[l + 1] + (t->bwd_gather && l < hops - 1 ? 0 : (size_t)rows * dims[l + 1]);
      if (l >= 1) da_floats = std::max(da_floats, (size_t)rows * 2 * dims[l]);
      e.a[l] = (float*)alloc((size_t)rows * 2 * dims[l] * 4);
      e.h[l] = (float*)alloc((size_t)rows * dims[l + 1] * 4);
      ok = ok && e.a[l] && e.h[l];
    }
  }
  float* z = (float*)alloc(zero_floats * 4);
  t->zero_base = z;
  t->zero_bytes = zero_floats * 4;
  for (int k = 0; k < 2 && z; ++k)
    for (int l = 0; l < hops; ++l) {
      gigl_nablp_train_plan::Enc& e = t->enc[k];
      e.gw[l] = z;
      z += (size_t)dims[l + 1] * 2 * dims[l];
      e.gb[l] = z;
      z += dims[l + 1];
      if (t->bwd_gather && l < hops - 1) {
        e.dh[l] = (float*)alloc((size_t)e.rows_cap[l] * dims[l + 1] * 4);
        ok = ok && e.dh[l];
        continue;
      }
      e.dh[l] = z;
      z += (size_t)e.rows_cap[l] * dims[l + 1];
    }
  for (int wi = 0; wi < gigl_nablp_train_plan::WS && t->bwd_gather; ++wi)
    for (int k = 0; k < 2; ++k)
      for (int l = 1; l < hops; ++l) {
        t->work[wi].tlists[k][l] = (int32_t*)alloc(
            (size_t)gigl_transposed_rows_words(t->enc[k].rows_cap[l - 1], t->work[wi].base[k]->un.cap_edges) * 4);
        ok = ok && t->work[wi].tlists[k][l];
      }
  const size_t d = (size_t)dims[hops], Q = (size_t)b_anchors * num_positives, Cn = Q + (size_t)t->n_rn;
  for (int k = 0; k < 2; ++k) {
    gigl_nablp_train_plan::Enc& e = t->enc[k];
    e.emb = (float*)alloc((size_t)e.b * d * 4);
    e.inv = (float*)alloc((size_t)e.b * 4);
    ok = ok && e.emb && e.inv;
  }
  t->da = (float*)alloc(da_floats * 4);
  t->wt = (float*)alloc(wt_floats * 4);
  t->rq = (float*)alloc(Q * d * 4);
  t->cand = (float*)alloc(Cn * d * 4);
  t->cand_t = (float*)alloc(Cn * d * 4);
  t->scores = (float*)alloc(Q * Cn * 4);
  t->dscores = (float*)alloc(Q * Cn * 4);
  t->d_rq = (float*)alloc(Q * d * 4);
  t->d_cand = (float*)alloc(Cn * d * 4);
  t->qid = (int64_t*)alloc(Q * 8);
  t->cid = (int64_t*)alloc(Cn * 8);
  t->valid = (int32_t*)alloc(Cn * 4);
  t->pos_cnt = (int32_t*)alloc((size_t)b_anchors * 4);
  t->consts = (int32_t*)alloc(64);
  t->row_lse = (float*)alloc(Q * 4);
  t->row_loss = (float*)alloc(Q * 4);
  t->loss = (float*)alloc(64);
  ok = ok && t->zero_base && t->da && t->wt && t->rq && t->cand && t->cand_t && t->scores && t->dscores && t->d_rq &&
       t->d_cand && t->qid && t->cid && t->valid && t->pos_cnt && t->consts && t->row_lse && t->row_loss && t->loss;
  t->fused_small = getenv("GIGL_TRAIN_PLAN_UNFUSED") == nullptr;
  for (int l = 0; l < hops && ok && t->fused_small; ++l) {
    const int n_out = dims[l + 1], k2 = 2 * dims[l];
    for (int k = 0; k < 2 && ok; ++k) {
      gigl_nablp_train_plan::Enc& e = t->enc[k];
      const int64_t chunks = gigl_linear_weight_grad_chunks(e.rows_cap[l], n_out, k2, &e.part_rc[l]);
      e.part_w[l] = (float*)alloc((size_t)chunks * n_out * k2 * 4);
      e.part_b[l] = (float*)alloc((size_t)chunks * n_out * 4);
      ok = e.part_w[l] && e.part_b[l];
    }
    if (l >= 1) t->wt_l[l] = (float*)alloc((size_t)n_out * k2 * 4);
    ok = ok && (l == 0 || t->wt_l[l]);
  }
  // (default on, GIGL_LP_FORK=0 off: 1.18 -> 1.07 ms per step, bit-identical results.  The forked step's layers part is
  // launched EAGERLY, not replayed: captured, the two-branch graph brought the whole GPU suite down with a segmentation fault
  // inside a step — ~400 tests into the session, three runs of three — and never in six runs once it was launched eagerly)
  const char* fork_env = getenv("GIGL_LP_FORK");
  if (ok && t->fused_small && t->n_rn > 0 && !(fork_env && fork_env[0] == '0')) {
    t->da2 = (float*)alloc(da_floats * 4);
    ok = t->da2 != nullptr && gigl_ctx_create(ctx->device, &t->actx) == GIGL_OK;
    for (int i = 0; i < 2 && ok; ++i)
      ok = hipEventCreateWithFlags(&t->ev_fork[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&t->ev_join[i], hipEventDisableTiming) == hipSuccess;
    const char* w_env = getenv("GIGL_LP_WGRAD_STREAM");  // (=0: the main batch's weight gradients stay in its chain)
    if (ok && t->kind == 0 && !(w_env && w_env[0] == '0'))
      ok = gigl_ctx_create(ctx->device, &t->wctx) == GIGL_OK &&
           hipEventCreateWithFlags(&t->ev_w, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&t->ev_wjoin, hipEventDisableTiming) == hipSuccess;
    t->fork = ok;
  }
  if (ok) {
    const int32_t c[16] = {(int32_t)Q, (int32_t)Cn, 0 /* Adam's step counter */, 0};
    if (hipMemcpy(t->consts, c, sizeof(c), hipMemcpyHostToDevice) != hipSuccess || hipMemset(t->loss, 0, 64) != hipSuccess)
      ok = false;
  }
  if (!ok) {
    gigl_nablp_train_plan_destroy(t);
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of the link-prediction training workspace failed");
  }
  *out = t;
  return GIGL_OK;
}

namespace {
// the graph part of workspace w for these roots, on its side stream; work[w].ev_graph marks its end
int32_t lp_graph_part(gigl_nablp_train_plan* t, int w, const uint32_t* main_roots, const uint32_t* rn_roots,
                      int32_t sampling_seed, int32_t mode) {
  gigl_nablp_train_plan::Work& wk = t->work[w];
  gigl_ctx* sc = wk.side;
  // the workspace is free once the layers part that last read it is done; the roots were written on the caller's stream
  GIGL_HIP_CHECK(sc, hipStreamWaitEvent(sc->stream, wk.ev_layers, 0));
  GIGL_HIP_CHECK(sc, hipEventRecord(t->ev_now, t->ctx->stream));
  GIGL_HIP_CHECK(sc, hipStreamWaitEvent(sc->stream, t->ev_now, 0));
  GIGL_HIP_CHECK(sc, hipMemcpyAsync(wk.base[0]->roots_buf, main_roots, (size_t)t->enc[0].b * 4, hipMemcpyDeviceToDevice, sc->stream));
  if (t->n_rn > 0)
    GIGL_HIP_CHECK(sc, hipMemcpyAsync(wk.base[1]->roots_buf, rn_roots, (size_t)t->n_rn * 4, hipMemcpyDeviceToDevice, sc->stream));
  else
    gigl_fill_u32(sc->stream, wk.base[1]->roots_buf, GIGL_INVALID, 1);
  const int32_t rc = train_run_part(sc, &wk.exec_graph, &wk.warm_graph,
                                    [&]() { return lp_enqueue_graph(t, w, sampling_seed, mode); }, 0);
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(sc, hipEventRecord(wk.ev_graph, sc->stream));
  wk.fetched_main = main_roots;
  wk.fetched_rn = rn_roots;
  wk.fetched = true;
  return GIGL_OK;
}
}  // namespace

int32_t gigl_nablp_train_plan_step2(gigl_nablp_train_plan* t, const uint32_t* main_roots, const int32_t* pos_cnt,
                                    const uint32_t* rn_roots, const uint32_t* next_main_roots, const uint32_t* next_rn_roots,
                                    int32_t sampling_seed, int32_t mode, float* loss_out) {
  if (!t) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = t->ctx;
  GIGL_REQUIRE(ctx, main_roots && pos_cnt && (rn_roots || t->n_rn == 0), "null argument");
  GIGL_REQUIRE(ctx, !next_main_roots || next_rn_roots || t->n_rn == 0, "the next step's random negatives are missing");
  if (mode == GIGL_MODE_REPLACE)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "the training plan needs duplicate-free trees (no with-replacement mode)");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  t->gat.parts_pending = t->kind == 1 && t->gat.fused;  // (this step's gradient buffers will lack the partial sums)
  hipStream_t st = ctx->stream;
  if (t->lctx->stream != st || t->lctx->own_stream) {
    const int32_t rs = gigl_ctx_set_stream(t->lctx, st);
    if (rs != GIGL_OK) return gigl_fail(ctx, rs, "%s", gigl_last_error(t->lctx));
  }
  if (t->cap_seed != sampling_seed || t->cap_mode != mode) {  // seed and mode are baked into the captured graph parts
    for (auto& wk : t->work) {
      GIGL_HIP_CHECK(ctx, hipStreamSynchronize(wk.side->stream));
      if (wk.exec_graph) hipGraphExecDestroy(wk.exec_graph);
      wk.exec_graph = nullptr;
      wk.fetched = false;  // (what was prefetched was sampled under the old seed)
    }
    t->cap_seed = sampling_seed;
    t->cap_mode = mode;
  }
  int32_t rc = GIGL_OK;
  const int w = t->cur;
  gigl_nablp_train_plan::Work& wk = t->work[w];
  // this step's trees: prefetched by the previous call for exactly these buffers, or sampled now
  if (!(wk.fetched && wk.fetched_main == main_roots && wk.fetched_rn == rn_roots)) {
    rc = lp_graph_part(t, w, main_roots, rn_roots, sampling_seed, mode);
    if (rc != GIGL_OK) return gigl_fail(ctx, rc, "%s", gigl_last_error(wk.side));
  }
  wk.fetched = false;  // (consumed)
  const int wn = (w + 1) % gigl_nablp_train_plan::WS;
  if (next_main_roots) {  // the next step's graph part goes to the other workspace, beside this step's layers
    rc = lp_graph_part(t, wn, next_main_roots, next_rn_roots, sampling_seed, mode);
    if (rc != GIGL_OK) return gigl_fail(ctx, rc, "%s", gigl_last_error(t->work[wn].side));
  }
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(t->pos_cnt, pos_cnt, (size_t)t->b * 4, hipMemcpyDeviceToDevice, st));
  GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, wk.ev_graph, 0));
  // (a forked step is launched eagerly: as fast as its replay — 1.07 / 1.71 ms both ways — and no graph with a second branch is
  // ever instantiated; see DESIGN_HISTORY "Round 6" for what those did to a long session)
  if (t->fork) rc = lp_enqueue_layers(t, w);
  else rc = train_run_part(t->lctx, &wk.exec_layers, &wk.warm_layers, [&]() { return lp_enqueue_layers(t, w); }, 1);
  if (rc != GIGL_OK) return gigl_fail(ctx, rc, "%s", gigl_last_error(t->lctx));
  GIGL_HIP_CHECK(ctx, hipEventRecord(wk.ev_layers, st));
  if (loss_out) GIGL_HIP_CHECK(ctx, hipMemcpyAsync(loss_out, t->loss, 8, hipMemcpyDeviceToDevice, st));
  if (next_main_roots) t->cur = wn;
  return GIGL_OK;
}

int32_t gigl_nablp_train_plan_step(gigl_nablp_train_plan* t, const uint32_t* main_roots, const int32_t* pos_cnt,
                                   const uint32_t* rn_roots, int32_t sampling_seed, int32_t mode, float* loss_out) {
  return gigl_nablp_train_plan_step2(t, main_roots, pos_cnt, rn_roots, nullptr, nullptr, sampling_seed, mode, loss_out);
}

const float* gigl_nablp_train_plan_loss(gigl_nablp_train_plan* t) { return t ? t->loss : nullptr; }

namespace {
__global__ __launch_bounds__(256) void lp_acc_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
__global__ __launch_bounds__(256) void lp_add2_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                      float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
}  // namespace

int32_t gigl_sage_train_plan_moments(gigl_sage_train_plan* t, int32_t layer, float* m_w, float* v_w, float* m_b, float* v_b) {
  if (!t) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = t->ctx;
  GIGL_REQUIRE(ctx, layer >= 0 && layer < t->L, "layer %d", layer);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t nw = (size_t)t->dims[layer + 1] * 2 * t->dims[layer], nb = (size_t)t->dims[layer + 1];
  float* dst[4] = {m_w, v_w, m_b, v_b};
  for (int k = 0; k < 4; ++k)
    if (dst[k])
      GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst[k], t->mom[4 * layer + k], (k < 2 ? nw : nb) * 4, hipMemcpyDeviceToDevice, ctx->stream));
  return GIGL_OK;
}

int32_t gigl_nablp_train_plan_moments(gigl_nablp_train_plan* t, int32_t index, float* m, float* v) {
  if (!t) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = t->ctx;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const float *sm = nullptr, *sv = nullptr;
  size_t n = 0;
  if (t->kind == 1) {
    GIGL_REQUIRE(ctx, index >= 0 && index < 8, "tensor %d", index);
    sm = t->gat.mom[2 * index];
    sv = t->gat.mom[2 * index + 1];
    n = (size_t)t->gat.n[index];
  } else {
    GIGL_REQUIRE(ctx, index >= 0 && index < 2 * t->L, "tensor %d", index);
    const int l = index >> 1, bias = index & 1;
    sm = t->mom[4 * l + 2 * bias];
    sv = t->mom[4 * l + 2 * bias + 1];
    n = bias ? (size_t)t->dims[l + 1] : (size_t)t->dims[l + 1] * 2 * t->dims[l];
  }
  if (m && n) GIGL_HIP_CHECK(ctx, hipMemcpyAsync(m, sm, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
  if (v && n) GIGL_HIP_CHECK(ctx, hipMemcpyAsync(v, sv, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
  return GIGL_OK;
}

int32_t gigl_sage_train_plan_adopt(gigl_sage_train_plan* dst, gigl_sage_train_plan* src) {
  if (!dst || !src) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = dst->ctx;
  GIGL_REQUIRE(ctx, dst->L == src->L, "plans of different depth");
  for (int l = 0; l <= dst->L; ++l) GIGL_REQUIRE(ctx, dst->dims[l] == src->dims[l], "plans of different widths");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(src->ctx->stream));
  for (int l = 0; l < dst->L; ++l) {
    const size_t nw = (size_t)dst->dims[l + 1] * 2 * dst->dims[l];
    for (int k = 0; k < 4; ++k)
      GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst->mom[4 * l + k], src->mom[4 * l + k], (k < 2 ? nw : (size_t)dst->dims[l + 1]) * 4,
                                         hipMemcpyDeviceToDevice, ctx->stream));
  }
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst->n_valid_buf + 1, src->n_valid_buf + 1, 4, hipMemcpyDeviceToDevice, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GIGL_OK;
}

int32_t gigl_nablp_train_plan_adopt(gigl_nablp_train_plan* dst, gigl_nablp_train_plan* src) {
  if (!dst || !src) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = dst->ctx;
  GIGL_REQUIRE(ctx, dst->L == src->L && dst->kind == src->kind, "plans of different kinds");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(src->ctx->stream));
  if (dst->kind == 1) {
    for (int i = 0; i < 8; ++i) {
      GIGL_REQUIRE(ctx, dst->gat.n[i] == src->gat.n[i], "plans of different widths");
      for (int j = 0; j < 2; ++j)
        GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst->gat.mom[2 * i + j], src->gat.mom[2 * i + j],
                                           (size_t)(dst->gat.n[i] ? dst->gat.n[i] : 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
  } else {
    for (int l = 0; l <= dst->L; ++l) GIGL_REQUIRE(ctx, dst->dims[l] == src->dims[l], "plans of different widths");
    for (int l = 0; l < dst->L; ++l) {
      const size_t nw = (size_t)dst->dims[l + 1] * 2 * dst->dims[l];
      for (int k = 0; k < 4; ++k)
        GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst->mom[4 * l + k], src->mom[4 * l + k], (k < 2 ? nw : (size_t)dst->dims[l + 1]) * 4,
                                           hipMemcpyDeviceToDevice, ctx->stream));
    }
  }
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst->consts + 2, src->consts + 2, 4, hipMemcpyDeviceToDevice, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GIGL_OK;
}

int32_t gigl_nablp_train_plan_grads(gigl_nablp_train_plan* t, int32_t layer, float* gw, float* gb) {
  if (!t) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = t->ctx;
  GIGL_REQUIRE(ctx, t->kind == 0 && layer >= 0 && layer < t->L && gw, "bad plan / layer / null output");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t nw = (int64_t)t->dims[layer + 1] * 2 * t->dims[layer];
  if (t->fused_small) {  // the step kept partial sums only: add them up now, into the (cleared) gradient buffers
    for (int k = 0; k < 2; ++k) {
      gigl_nablp_train_plan::Enc& e = t->enc[k];
      gigl_fill_u32(ctx->stream, e.gw[layer], 0u, nw);
      if (t->bias[layer]) gigl_fill_u32(ctx->stream, e.gb[layer], 0u, (int64_t)t->dims[layer + 1]);
      const int32_t rc = gigl_linear_weight_grad_sum(ctx, e.part_w[layer], t->bias[layer] ? e.part_b[layer] : nullptr,
                                                     e.base->un.meta + GIGL_META_LEVEL0 + (t->L - 1 - layer), t->dims[layer + 1],
                                                     2 * t->dims[layer], e.part_rc[layer], e.gw[layer],
                                                     t->bias[layer] ? e.gb[layer] : nullptr);
      if (rc != GIGL_OK) return rc;
    }
  }
  hipLaunchKernelGGL(lp_add2_kernel, dim3(256), dim3(256), 0, ctx->stream, (const float*)t->enc[0].gw[layer],
                     (const float*)t->enc[1].gw[layer], nw, gw);
  if (gb && t->bias[layer])
    hipLaunchKernelGGL(lp_add2_kernel, dim3(4), dim3(256), 0, ctx->stream, (const float*)t->enc[0].gb[layer],
                       (const float*)t->enc[1].gb[layer], (int64_t)t->dims[layer + 1], gb);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// ------------------------------------------------------------------------------------------
// The same step with the GAT encoder of configs[4] (GAT.init_conv_layers, python/gigl/src/common/models/pyg/
// homogeneous.py:300-343, under node_anchor_based_link_prediction_modeling_task_spec.py:334-451): two GATConv layers,
// `heads` concatenated heads in the first, one in the second, no edge features.  The first layer runs from the INPUT
// side (as models_attn.GAT does when the stored rows are wider than heads * channels): attention-weighted sums of the
// stored rows under the folded vectors u_h = W_h^T att_h (gigl_gat_input_aggregate), then one projection per head — for
// the nodes of level <= 1 only; the second layer for the roots only.  Backward: gigl_gat_aggregate_backward + epilogue,
// the projections' weight / input gradients, gigl_gat_input_aggregate_backward (d u), the fold's own backward.  Both
// encodes ADD into one set of gradients; head, loss, Adam, workspaces and prefetch are gigl_nablp_train_plan's.
namespace {

// u[h][k] = sum_c att_src[hC + c] W[hC + c][k];  u[H + h][k] = the same with att_dst
__global__ __launch_bounds__(256) void gat_fold_kernel(const float* __restrict__ w, const float* __restrict__ att_src,
                                                       const float* __restrict__ att_dst, int H, int C, int d,
                                                       float* __restrict__ u) {
  // grid (2H, ceil(d / 64)); a workgroup = 64 columns x 4 slices of the C rows
  __shared__ float s_part[4][64];
  const int hh = blockIdx.x;  // 0 .. 2H
  const int h = hh % H;
  const float* att = (hh < H ? att_src : att_dst) + h * C;
  const int k = blockIdx.y * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float acc = 0.f;
  if (k < d)
    for (int c = q; c < C; c += 4) acc += att[c] * w[(int64_t)(h * C + c) * d + k];
  s_part[q][threadIdx.x & 63] = acc;
  __syncthreads();
  if (q == 0 && k < d) u[(int64_t)hh * d + k] = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
}
// the fold's backward, one workgroup per weight row r = hC + c:
//   gw[r][k] += att_src[r] du[h][k] + att_dst[r] du[H + h][k];  g_att_src[r] += <W[r], du[h]>;  g_att_dst[r] += <W[r], du[H + h]>
__global__ __launch_bounds__(256) void gat_fold_backward_kernel(const float* __restrict__ w, const float* __restrict__ att_src,
                                                                const float* __restrict__ att_dst, const float* __restrict__ du,
                                                                int H, int C, int d, float* __restrict__ gw,
                                                                float* __restrict__ g_att_src, float* __restrict__ g_att_dst) {
  __shared__ float s_a[4], s_b[4];
  const int r = blockIdx.x, h = r / C;
  const float as = att_src[r], ad = att_dst[r];
  const float* ds = du + (int64_t)h * d;
  const float* dd = du + (int64_t)(H + h) * d;
  float pa = 0.f, pb = 0.f;
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    const float wv = w[(int64_t)r * d + k], a = ds[k], b = dd[k];
    gw[(int64_t)r * d + k] += as * a + ad * b;
    pa += wv * a;
    pb += wv * b;
  }
  for (int o = 32; o >= 1; o >>= 1) {
    pa += __shfl_xor(pa, o, 64);
    pb += __shfl_xor(pb, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_a[threadIdx.x >> 6] = pa;
    s_b[threadIdx.x >> 6] = pb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    g_att_src[r] += s_a[0] + s_a[1] + s_a[2] + s_a[3];
    g_att_dst[r] += s_b[0] + s_b[1] + s_b[2] + s_b[3];
  }
}
// out[h][i][c] = y[i][hC + c] > 0 ? dy[i][hC + c] : 0 for i < *n_rows: the heads' slices of the relu'd layer's gradient, each
// contiguous (planes rows_cap * C floats apart)
__global__ __launch_bounds__(256) void gat_split_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const int32_t* __restrict__ n_rows_dev, int64_t rows_cap, int H,
                                                             int C, float* __restrict__ out) {
  const int64_t n = (int64_t)*n_rows_dev * H * C;
  const int HC = H * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / HC;
    const int col = (int)(i - row * HC), h = col / C, c = col - h * C;
    out[((int64_t)h * rows_cap + row) * C + c] = y[i] > 0.f ? dy[i] : 0.f;
  }
}
// y[i][c] = x[i][c] + bias[c] for i < *n_rows (x without the bias stays: the attention backward wants it)
__global__ __launch_bounds__(256) void gat_add_bias_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                           const int32_t* __restrict__ n_rows_dev, int C, float* __restrict__ y) {
  const int64_t n = (int64_t)*n_rows_dev * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i] + (bias ? bias[i % C] : 0.f);
}
// gb[c] += sum over the rows i < *n_rows of dy[i][c]: 8 rows per workgroup, one atomic per (workgroup, column)
__global__ __launch_bounds__(256) void gat_bias_grad_kernel(const float* __restrict__ dy, const int32_t* __restrict__ n_rows_dev,
                                                            int C, float* __restrict__ gb) {
  const int n = *n_rows_dev;
  const int r0 = blockIdx.x * 8, r1 = r0 + 8 < n ? r0 + 8 : n;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += dy[(int64_t)r * C + c];
    if (acc != 0.f) atomicAdd(&gb[c], acc);
  }
}

int64_t gat_rows1(const gigl_nablp_train_plan* t, int which) { return t->enc[which].rows_cap[0]; }

// once per step: the folded attention vectors of the first layer (both encodes read them)
int32_t gat_lp_begin(gigl_nablp_train_plan* t) {
  const auto& g = t->gat;
  hipLaunchKernelGGL(gat_fold_kernel, dim3((unsigned)(2 * g.heads), (unsigned)((g.d_in + 63) / 64)), dim3(256), 0,
                     t->lctx->stream, (const float*)g.w[0],
                     (const float*)g.att_src[0], (const float*)g.att_dst[0], g.heads, g.c0, g.d_in, g.u);
  if (g.fused) {  // W1^T and the heads' W0^T for the input gradients of both encodes
    hipStream_t st = t->lctx->stream;
    const int HC = g.heads * g.c0;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(((int64_t)g.c1 * HC + 255) / 256)), dim3(256), 0, st, (const float*)g.w[1],
                       g.c1, HC, g.wt1);
    for (int h = 0; h < g.heads; ++h)
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(((int64_t)g.c0 * g.d_in + 255) / 256)), dim3(256), 0, st,
                         (const float*)(g.w[0] + (int64_t)h * g.c0 * g.d_in), g.c0, g.d_in, g.wt0[h]);
  }
  GIGL_HIP_CHECK(t->lctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gat_lp_forward(gigl_nablp_train_plan* t, int which) {
  gigl_nablp_train_plan::Enc& e = t->enc[which];
  gigl_sage_plan* p = e.base;
  const bool alt = which == 1 && t->fork;
  gigl_ctx* ctx = alt ? t->actx : t->lctx;
  hipStream_t st = ctx->stream;
  auto& g = t->gat;
  float* alpha = alt ? g.x.alpha : g.alpha;
  const int H = g.heads, C0 = g.c0, C1 = g.c1, d = g.d_in;
  const int64_t rows1 = gat_rows1(t, which);
  const int32_t* n0 = p->un.meta + GIGL_META_LEVEL0;
  const int32_t* n1 = p->un.meta + GIGL_META_LEVEL0 + 1;
  int32_t rc = gigl_gat_input_aggregate(ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, g.u, H, g.slope, p->un.rowptr,
                                        p->un.rowend, p->un.col, n1, rows1, g.z[which]);
  if (rc != GIGL_OK) return rc;
  // h0[:, hC:(h+1)C] = relu(z_h W_h^T + b_h): one launch for all heads
  rc = gigl_linear_batched(ctx, g.z[which], g.w[0], g.bias[0], n1, rows1, d, C0, 1, H, rows1 * d, (int64_t)C0 * d, H * C0, e.h[0]);
  if (rc != GIGL_OK) return rc;
  rc = gigl_linear(ctx, e.h[0], g.w[1], nullptr, n1, rows1, H * C0, C1, 0, g.xw[which]);
  if (rc != GIGL_OK) return rc;
  rc = gigl_gat_aggregate(ctx, g.xw[which], g.att_src[1], g.att_dst[1], 1, C1, g.slope, 1, p->un.rowptr, p->un.rowend,
                          p->un.col, n1, rows1, n0, e.rows_cap[1], nullptr, 0, alpha, g.out_pre[which]);
  if (rc != GIGL_OK) return rc;
  {
    int64_t blocks = (e.rows_cap[1] * C1 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gat_add_bias_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)g.out_pre[which],
                       (const float*)g.bias[1], n0, C1, e.h[1]);
  }
  hipLaunchKernelGGL(lp_take_norm_kernel, dim3((unsigned)((e.b + 3) / 4)), dim3(256), 0, st, (const float*)e.h[1],
                     (const int32_t*)p->un.root_local, e.b, C1, t->normalize, (const int32_t*)p->un.meta, e.emb, e.inv);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gat_lp_backward(gigl_nablp_train_plan* t, int which) {
  gigl_nablp_train_plan::Enc& e = t->enc[which];
  gigl_sage_plan* p = e.base;
  const bool alt = which == 1 && t->fork;
  gigl_ctx* ctx = alt ? t->actx : t->lctx;
  hipStream_t st = ctx->stream;
  auto& g = t->gat;
  float *s_alpha = alt ? g.x.alpha : g.alpha, *s_dxw = alt ? g.x.dxw : g.dxw, *s_ds = alt ? g.x.ds : g.ds,
        *s_dd = alt ? g.x.dd : g.dd, *s_dh0 = alt ? g.x.dh0 : g.dh0, *s_dh0s = alt ? g.x.dh0s : g.dh0s,
        *s_dz = alt ? g.x.dz : g.dz, *s_edge = alt ? g.x.edge_scratch : g.edge_scratch, *s_du = alt ? g.x.du : g.du;
  float* const* gg = alt ? g.x.g : g.g;
  const int H = g.heads, C0 = g.c0, C1 = g.c1, d = g.d_in, HC = H * C0;
  const int64_t rows1 = gat_rows1(t, which);
  const int32_t* n0 = p->un.meta + GIGL_META_LEVEL0;
  const int32_t* n1 = p->un.meta + GIGL_META_LEVEL0 + 1;
  float *gw0 = gg[0], *gas0 = gg[1], *gad0 = gg[2], *gb0 = gg[3], *gw1 = gg[4], *gas1 = gg[5], *gad1 = gg[6], *gb1 = gg[7];
  (void)gas0;
  (void)gad0;
  // ---- second layer (the roots' rows): e.dh[1] = d loss / d (out before the bias)
  if (g.bias[1])
    hipLaunchKernelGGL(gat_bias_grad_kernel, dim3((unsigned)((e.rows_cap[1] + 7) / 8)), dim3(256), 0, st, (const float*)e.dh[1],
                       n0, C1, gb1);
  gigl_fill_u32(st, s_dxw, 0u, rows1 * C1);
  gigl_fill_u32(st, s_ds, 0u, s_dd - s_ds + rows1);  // (ds | dd: dd starts where the larger encode's ds ends)
  int32_t rc = gigl_gat_aggregate_backward(ctx, g.xw[which], g.att_src[1], g.att_dst[1], 1, C1, g.slope, p->un.rowptr,
                                           p->un.rowend, p->un.col, n1, rows1, n0, e.rows_cap[1], g.out_pre[which], e.dh[1],
                                           nullptr, 0, p->un.cap_edges, nullptr, s_alpha, s_dxw, s_ds, s_dd, nullptr, nullptr,
                                           nullptr);
  if (rc != GIGL_OK) return rc;
  rc = gigl_gat_backward_epilogue(ctx, s_dxw, s_ds, s_dd, g.xw[which], g.att_src[1], g.att_dst[1], n1, rows1, 1, C1, gas1, gad1);
  if (rc != GIGL_OK) return rc;
  // (measured: the main batch's weight gradients on a third stream, as the GraphSAGE plan runs them — 1.75 against 1.71 ms:
  // these 768-wide ones fill the GPU and slow the chain beside them)
  if (g.fused) rc = gigl_linear_weight_grad_parts(ctx, s_dxw, e.h[0], nullptr, n1, rows1, C1, HC, g.part_w1[which], nullptr);
  else rc = gigl_linear_weight_grad(ctx, s_dxw, e.h[0], nullptr, n1, rows1, C1, HC, gw1, nullptr);
  if (rc != GIGL_OK) return rc;
  if (!g.fused) {
    int64_t blocks = ((int64_t)C1 * HC + 255) / 256;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)g.w[1], C1, HC, t->wt);
  }
  rc = gigl_linear(ctx, s_dxw, g.fused ? g.wt1 : t->wt, nullptr, n1, rows1, C1, HC, 0, s_dh0);
  if (rc != GIGL_OK) return rc;
  // ---- first layer: relu mask, the heads' slices, their projections' backward, the attention-weighted sums' backward
  {
    int64_t blocks = (rows1 * HC + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gat_split_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)s_dh0, (const float*)e.h[0],
                       n1, rows1, H, C0, s_dh0s);
  }
  for (int h = 0; h < H; ++h) {
    const float* dyh = s_dh0s + (int64_t)h * rows1 * C0;
    if (g.fused)
      rc = gigl_linear_weight_grad_parts(ctx, dyh, g.z[which] + (int64_t)h * rows1 * d, nullptr, n1, rows1, C0, d,
                                         g.part_w0[which][h], g.bias[0] ? g.part_b0[which][h] : nullptr);
    else
      rc = gigl_linear_weight_grad(ctx, dyh, g.z[which] + (int64_t)h * rows1 * d, nullptr, n1, rows1, C0, d,
                                   gw0 + (int64_t)h * C0 * d, g.bias[0] ? gb0 + h * C0 : nullptr);
    if (rc != GIGL_OK) return rc;
    if (!g.fused) {
      int64_t blocks = ((int64_t)C0 * d + 255) / 256;
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)(g.w[0] + (int64_t)h * C0 * d),
                         C0, d, t->wt);
    }
    rc = gigl_linear(ctx, dyh, g.fused ? g.wt0[h] : t->wt, nullptr, n1, rows1, C0, d, 0, s_dz + (int64_t)h * rows1 * d);
    if (rc != GIGL_OK) return rc;
  }
  rc = gigl_gat_input_aggregate_backward(ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, g.u, H, g.slope, p->un.rowptr,
                                         p->un.rowend, p->un.col, n1, rows1, s_dz, s_edge, s_du);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return rc;
}

// after both encodes: d u -> the first layer's weights and attention vectors, then Adam over the eight tensors
int32_t gat_lp_finish(gigl_nablp_train_plan* t) {
  gigl_ctx* ctx = t->lctx;
  hipStream_t st = ctx->stream;
  auto& g = t->gat;
  if (t->fork)  // the random negatives' share of d u, accumulated on its own stream
    hipLaunchKernelGGL(lp_acc_kernel, dim3(64), dim3(256), 0, st, g.du, (const float*)g.x.du, (int64_t)2 * g.heads * g.d_in);
  hipLaunchKernelGGL(gat_fold_backward_kernel, dim3((unsigned)(g.heads * g.c0)), dim3(256), 0, st, (const float*)g.w[0],
                     (const float*)g.att_src[0], (const float*)g.att_dst[0], (const float*)g.du, g.heads, g.c0, g.d_in, g.g[0],
                     g.g[1], g.g[2]);
  AdamPack2 ap{};
  float* params[8] = {g.w[0], g.att_src[0], g.att_dst[0], g.bias[0], g.w[1], g.att_src[1], g.att_dst[1], g.bias[1]};
  const int32_t* n1a = t->enc[0].base->un.meta + GIGL_META_LEVEL0 + 1;
  const int32_t* n1b = t->enc[1].base->un.meta + GIGL_META_LEVEL0 + 1;
  auto add = [&](float* prm, const float* grad, float* m, float* v, int64_t n, const float* pa, const float* pb, int32_t rca,
                 int32_t rcb, const float* grad2 = nullptr) {
    ap.p[ap.count] = prm;
    ap.g1[ap.count] = grad;
    ap.g2[ap.count] = grad2;
    ap.part1[ap.count] = pa;
    ap.part2[ap.count] = pb;
    ap.rows1[ap.count] = n1a;
    ap.rows2[ap.count] = n1b;
    ap.rc1[ap.count] = rca;
    ap.rc2[ap.count] = rcb;
    ap.m[ap.count] = m;
    ap.v[ap.count] = v;
    ap.n[ap.count++] = n;
  };
  for (int i = 0; i < 8; ++i) {
    if (!params[i]) continue;
    if (g.fused && (i == 0 || i == 3)) {  // W0 / b0: a slice per head, each with its own partial sums of both encodes
      const int64_t ns = i == 0 ? (int64_t)g.c0 * g.d_in : g.c0;
      for (int h = 0; h < g.heads; ++h)
        add(params[i] + h * ns, g.g[i] + h * ns, g.mom[2 * i] + h * ns, g.mom[2 * i + 1] + h * ns, ns,
            i == 0 ? g.part_w0[0][h] : g.part_b0[0][h], i == 0 ? g.part_w0[1][h] : g.part_b0[1][h], g.rc_w0[0], g.rc_w0[1]);
    } else if (g.fused && i == 4) {
      add(params[i], g.g[i], g.mom[2 * i], g.mom[2 * i + 1], g.n[i], g.part_w1[0], g.part_w1[1], g.rc_w1[0], g.rc_w1[1]);
    } else {  // (fork: the second layer's attention vectors and bias were accumulated per encode)
      add(params[i], g.g[i], g.mom[2 * i], g.mom[2 * i + 1], g.n[i], nullptr, nullptr, 0, 0,
          t->fork && i >= 5 ? g.x.g[i] : nullptr);
    }
  }

  ap.lr = t->lr;
  ap.beta1 = t->beta1;
  ap.beta2 = t->beta2;
  ap.eps = t->eps;
  ap.wd = t->wd;
  hipLaunchKernelGGL(lp_adam_kernel, g.fused ? dim3(104, (unsigned)ap.count) : dim3(256), dim3(256), 0, st, ap,
                     (const int32_t*)(t->consts + 2), (const int32_t*)t->enc[0].base->un.meta,
                     (const int32_t*)t->enc[1].base->un.meta);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // namespace

int32_t gigl_gat_nablp_train_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b_anchors,
                                         int32_t num_positives, int32_t n_random_negatives, const int32_t* fanouts, int32_t hops,
                                         const int32_t* heads, const int32_t* channels, float* const* w,
                                         float* const* att_src, float* const* att_dst, float* const* bias,
                                         float negative_slope, int32_t l2_normalize, float temperature,
                                         int32_t remove_accidental_hits, float lr, float beta1, float beta2, float eps,
                                         float weight_decay, gigl_nablp_train_plan** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, graph && feat && fanouts && heads && channels && w && att_src && att_dst, "null argument");
  GIGL_REQUIRE(ctx, b_anchors >= 1 && num_positives >= 1 && n_random_negatives >= 0, "bad batch shape");
  if (hops != 2 || heads[1] != 1)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "the GAT link-prediction plan runs two layers, one head in the second");
  const int H = heads[0], C0 = channels[0], C1 = channels[1], d = feat->d;
  if ((d & 3) || d > 1024 || (H != 1 && H != 2 && H != 4) || (feat->dtype != GIGL_DTYPE_F32 && feat->dtype != GIGL_DTYPE_F16) ||
      H * C0 > 1024 || (C0 & 3) || (C1 & 3) || C1 > 512)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "GAT link-prediction plan: feature dim %d / %d heads x %d / %d channels outside "
                     "the built shapes", d, H, C0, C1);
  for (int l = 0; l < 2; ++l) GIGL_REQUIRE(ctx, w[l] && att_src[l] && att_dst[l], "layer %d: null parameter", l);
  if (fanouts[1] > GIGL_FAST_FANOUT) return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "second fan-out beyond %d", GIGL_FAST_FANOUT);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_nablp_train_plan* t = new (std::nothrow) gigl_nablp_train_plan();
  if (!t) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  t->ctx = ctx;
  t->kind = 1;
  t->L = 2;
  t->b = b_anchors;
  t->P = num_positives;
  t->n_rn = n_random_negatives;
  t->normalize = l2_normalize ? 1 : 0;
  t->remove_hits = remove_accidental_hits ? 1 : 0;
  t->temperature = temperature;
  t->act_last = 0;
  t->lr = lr;
  t->beta1 = beta1;
  t->beta2 = beta2;
  t->eps = eps;
  t->wd = weight_decay;
  auto& g = t->gat;
  g.heads = H;
  g.c0 = C0;
  g.c1 = C1;
  g.d_in = d;
  g.slope = negative_slope;
  for (int l = 0; l < 2; ++l) {
    g.w[l] = w[l];
    g.att_src[l] = att_src[l];
    g.att_dst[l] = att_dst[l];
    g.bias[l] = bias ? bias[l] : nullptr;
  }
  const int32_t dims[3] = {d, H * C0, C1};
  for (int l = 0; l <= 2; ++l) t->dims[l] = dims[l];
  int32_t rc = gigl_ctx_create(ctx->device, &t->lctx);
  const int32_t nb[2] = {b_anchors * (1 + num_positives), n_random_negatives > 0 ? n_random_negatives : 1};
  for (int wi = 0; wi < gigl_nablp_train_plan::WS && rc == GIGL_OK; ++wi) {
    gigl_nablp_train_plan::Work& wk = t->work[wi];
    rc = gigl_ctx_create(ctx->device, &wk.side);
    if (rc == GIGL_OK) wk.side->wide = ctx->wide;
    for (int k = 0; k < 2 && rc == GIGL_OK; ++k) {
      t->enc[k].b = nb[k];
      // (the base plans are tree + union workspaces here: their own layer buffers stay unused)
      rc = plan_create(wk.side, graph, feat, nb[k], fanouts, hops, dims, (const float* const*)w, (const float* const*)bias, 0,
                       false, &wk.base[k]);
      if (rc != GIGL_OK) gigl_fail(ctx, rc, "%s", gigl_last_error(wk.side));
      // every node of the batch graph is numbered: gigl_gat_input_aggregate reads its sources through un.nodes
      if (rc == GIGL_OK) wk.base[k]->leaf_global = false;
    }
    if (rc == GIGL_OK && (hipEventCreateWithFlags(&wk.ev_graph, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&wk.ev_layers, hipEventDisableTiming) != hipSuccess))
      rc = GIGL_E_HIP;
  }
  if (rc == GIGL_OK && hipEventCreateWithFlags(&t->ev_now, hipEventDisableTiming) != hipSuccess) rc = GIGL_E_HIP;
  if (rc != GIGL_OK) {
    gigl_nablp_train_plan_destroy(t);
    return rc;
  }
  for (int k = 0; k < 2; ++k) t->enc[k].base = t->work[0].base[k];
  auto alloc = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
    t->owned.push_back(q);
    return q;
  };
  bool ok = true;
  const int64_t n_par[8] = {(int64_t)H * C0 * d, H * C0, H * C0, g.bias[0] ? H * C0 : 0, (int64_t)C1 * H * C0, C1, C1, g.bias[1] ? C1 : 0};
  size_t zero_floats = (size_t)2 * H * d;  // du
  for (int i = 0; i < 8; ++i) {
    g.n[i] = n_par[i];
    zero_floats += (size_t)n_par[i];
  }
  // (default on, GIGL_LP_FORK=0 off, as the GraphSAGE plan: the forked step's layers part launched eagerly; needs the partial-sum mode)
  const char* fork_env = getenv("GIGL_LP_FORK");
  const bool want_fork = n_random_negatives > 0 && getenv("GIGL_TRAIN_PLAN_UNFUSED") == nullptr && !(fork_env && fork_env[0] == '0');
  if (want_fork) zero_floats += (size_t)2 * H * d + (size_t)(n_par[5] + n_par[6] + n_par[7]);  // the second encode's du, att / bias sums
  int64_t rows1_max = 0;
  for (int k = 0; k < 2; ++k) {
    gigl_nablp_train_plan::Enc& e = t->enc[k];
    e.rows_cap[0] = gigl_level_rows(ctx->wide, e.b, fanouts, hops, 1);  // nodes of level <= 1
    e.rows_cap[1] = e.b;
    if (e.rows_cap[0] > rows1_max) rows1_max = e.rows_cap[0];
    zero_floats += (size_t)e.b * C1;  // dh[1]
    e.h[0] = (float*)alloc((size_t)e.rows_cap[0] * H * C0 * 4);
    e.h[1] = (float*)alloc((size_t)e.b * C1 * 4);
    g.z[k] = (float*)alloc((size_t)H * e.rows_cap[0] * d * 4);
    g.xw[k] = (float*)alloc((size_t)e.rows_cap[0] * C1 * 4);
    g.out_pre[k] = (float*)alloc((size_t)e.b * C1 * 4);
    e.emb = (float*)alloc((size_t)e.b * C1 * 4);
    e.inv = (float*)alloc((size_t)e.b * 4);
    ok = ok && e.h[0] && e.h[1] && g.z[k] && g.xw[k] && g.out_pre[k] && e.emb && e.inv;
  }
  float* z = (float*)alloc(zero_floats * 4);
  t->zero_base = z;
  t->zero_bytes = zero_floats * 4;
  if (z) {
    for (int i = 0; i < 8; ++i) {
      g.g[i] = z;
      z += n_par[i];
    }
    g.du = z;
    z += (size_t)2 * H * d;
    for (int k = 0; k < 2; ++k) {
      t->enc[k].dh[1] = z;
      z += (size_t)t->enc[k].b * C1;
    }
    if (want_fork) {
      g.x.du = z;
      z += (size_t)2 * H * d;
      for (int i = 5; i < 8; ++i) {
        g.x.g[i] = z;
        z += n_par[i];
      }
    }
  }
  for (int i = 0; i < 8 && ok; ++i)
    for (int j = 0; j < 2; ++j) {
      g.mom[2 * i + j] = (float*)alloc((size_t)(n_par[i] ? n_par[i] : 1) * 4);
      if (!g.mom[2 * i + j] || hipMemset(g.mom[2 * i + j], 0, (size_t)(n_par[i] ? n_par[i] : 1) * 4) != hipSuccess) ok = false;
    }
  const int64_t cap_edges = t->work[0].base[0]->un.cap_edges;
  g.u = (float*)alloc((size_t)2 * H * d * 4);
  g.dxw = (float*)alloc((size_t)rows1_max * C1 * 4);
  g.ds = (float*)alloc((size_t)2 * rows1_max * 4);
  g.dd = g.ds ? g.ds + rows1_max : nullptr;
  g.alpha = (float*)alloc((size_t)(2 * rows1_max + cap_edges) * 4);
  g.dh0 = (float*)alloc((size_t)rows1_max * H * C0 * 4);
  g.dh0s = (float*)alloc((size_t)rows1_max * H * C0 * 4);
  g.dz = (float*)alloc((size_t)H * rows1_max * d * 4);
  g.edge_scratch = (float*)alloc((size_t)2 * H * cap_edges * 4);
  const size_t wt_floats = std::max((size_t)C1 * H * C0, (size_t)C0 * d);
  t->wt = (float*)alloc(wt_floats * 4);
  ok = ok && t->zero_base && g.u && g.dxw && g.ds && g.alpha && g.dh0 && g.dh0s && g.dz && g.edge_scratch && t->wt;
  g.fused = getenv("GIGL_TRAIN_PLAN_UNFUSED") == nullptr;
  if (want_fork && ok) {  // the second encode's own scratch (its rows, its edges), a ctx with a stream of its own, the events
    const int64_t r1 = t->enc[1].rows_cap[0], ce1 = t->work[0].base[1]->un.cap_edges;
    g.x.dxw = (float*)alloc((size_t)r1 * C1 * 4);
    g.x.ds = (float*)alloc((size_t)2 * r1 * 4);
    g.x.dd = g.x.ds ? g.x.ds + r1 : nullptr;
    g.x.alpha = (float*)alloc((size_t)(2 * r1 + ce1) * 4);
    g.x.dh0 = (float*)alloc((size_t)r1 * H * C0 * 4);
    g.x.dh0s = (float*)alloc((size_t)r1 * H * C0 * 4);
    g.x.dz = (float*)alloc((size_t)H * r1 * d * 4);
    g.x.edge_scratch = (float*)alloc((size_t)2 * H * ce1 * 4);
    ok = g.x.dxw && g.x.ds && g.x.alpha && g.x.dh0 && g.x.dh0s && g.x.dz && g.x.edge_scratch && g.x.du &&
         gigl_ctx_create(ctx->device, &t->actx) == GIGL_OK;
    for (int i = 0; i < 2 && ok; ++i)
      ok = hipEventCreateWithFlags(&t->ev_fork[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&t->ev_join[i], hipEventDisableTiming) == hipSuccess;
    const char* w_env = getenv("GIGL_LP_WGRAD_STREAM");  // (=0: the main batch's weight gradients stay in its chain)
    if (ok && t->kind == 0 && !(w_env && w_env[0] == '0'))
      ok = gigl_ctx_create(ctx->device, &t->wctx) == GIGL_OK &&
           hipEventCreateWithFlags(&t->ev_w, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&t->ev_wjoin, hipEventDisableTiming) == hipSuccess;
    t->fork = ok;
  }
  if (g.fused && ok) {
    g.wt1 = (float*)alloc((size_t)C1 * H * C0 * 4);
    ok = g.wt1 != nullptr;
    for (int h = 0; h < H && ok; ++h) {
      g.wt0[h] = (float*)alloc((size_t)C0 * d * 4);
      ok = g.wt0[h] != nullptr;
    }
    for (int k = 0; k < 2 && ok; ++k) {
      const int64_t rows1 = t->enc[k].rows_cap[0];
      const int64_t ch1 = gigl_linear_weight_grad_chunks(rows1, C1, H * C0, &g.rc_w1[k]);
      g.part_w1[k] = (float*)alloc((size_t)ch1 * C1 * H * C0 * 4);
      const int64_t ch0 = gigl_linear_weight_grad_chunks(rows1, C0, d, &g.rc_w0[k]);
      ok = g.part_w1[k] != nullptr;
      for (int h = 0; h < H && ok; ++h) {
        g.part_w0[k][h] = (float*)alloc((size_t)ch0 * C0 * d * 4);
        g.part_b0[k][h] = (float*)alloc((size_t)ch0 * C0 * 4);
        ok = g.part_w0[k][h] && g.part_b0[k][h];
      }
    }
  }
  const size_t de = (size_t)C1, Q = (size_t)b_anchors * num_positives, Cn = Q + (size_t)t->n_rn;
  t->rq = (float*)alloc(Q * de * 4);
  t->cand = (float*)alloc(Cn * de * 4);
  t->cand_t = (float*)alloc(Cn * de * 4);
  t->scores = (float*)alloc(Q * Cn * 4);
  t->dscores = (float*)alloc(Q * Cn * 4);
  t->d_rq = (float*)alloc(Q * de * 4);
  t->d_cand = (float*)alloc(Cn * de * 4);
  t->qid = (int64_t*)alloc(Q * 8);
  t->cid = (int64_t*)alloc(Cn * 8);
  t->valid = (int32_t*)alloc(Cn * 4);
  t->pos_cnt = (int32_t*)alloc((size_t)b_anchors * 4);
  t->consts = (int32_t*)alloc(64);
  t->row_lse = (float*)alloc(Q * 4);
  t->row_loss = (float*)alloc(Q * 4);
  t->loss = (float*)alloc(64);
  ok = ok && t->rq && t->cand && t->cand_t && t->scores && t->dscores && t->d_rq && t->d_cand && t->qid && t->cid && t->valid &&
       t->pos_cnt && t->consts && t->row_lse && t->row_loss && t->loss;
  if (ok) {
    const int32_t c[16] = {(int32_t)Q, (int32_t)Cn, 0 /* Adam's step counter */, 0};
    if (hipMemcpy(t->consts, c, sizeof(c), hipMemcpyHostToDevice) != hipSuccess || hipMemset(t->loss, 0, 64) != hipSuccess)
      ok = false;
  }
  if (!ok) {
    gigl_nablp_train_plan_destroy(t);
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of the GAT link-prediction training workspace failed");
  }
  *out = t;
  return GIGL_OK;
}

int32_t gigl_gat_nablp_train_plan_grads(gigl_nablp_train_plan* t, int32_t layer, float* gw, float* g_att_src,
                                        float* g_att_dst, float* gb) {
  if (!t) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = t->ctx;
  GIGL_REQUIRE(ctx, t->kind == 1 && layer >= 0 && layer < 2 && gw && g_att_src && g_att_dst, "bad plan / layer / null output");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (t->gat.fused && t->gat.parts_pending) {  // the step kept the projections' gradients as partial sums: add them up, once
    auto& g = t->gat;
    for (int k = 0; k < 2; ++k) {
      const int32_t* n1 = t->enc[k].base->un.meta + GIGL_META_LEVEL0 + 1;
      int32_t rc = gigl_linear_weight_grad_sum(ctx, g.part_w1[k], nullptr, n1, g.c1, g.heads * g.c0, g.rc_w1[k], g.g[4], nullptr);
      for (int h = 0; h < g.heads && rc == GIGL_OK; ++h)
        rc = gigl_linear_weight_grad_sum(ctx, g.part_w0[k][h], g.bias[0] ? g.part_b0[k][h] : nullptr, n1, g.c0, g.d_in, g.rc_w0[k],
                                         g.g[0] + (int64_t)h * g.c0 * g.d_in, g.bias[0] ? g.g[3] + h * g.c0 : nullptr);
      if (rc != GIGL_OK) return rc;
    }
    g.parts_pending = false;
  }
  float* dst[4] = {gw, g_att_src, g_att_dst, gb};
  for (int i = 0; i < 4; ++i) {
    const int j = 4 * layer + i;
    if (!dst[i] || !t->gat.n[j]) continue;
    GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst[i], t->gat.g[j], (size_t)t->gat.n[j] * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (t->fork && j >= 5)  // (the second encode's share, accumulated on its own stream)
      hipLaunchKernelGGL(lp_acc_kernel, dim3(4), dim3(256), 0, ctx->stream, dst[i], (const float*)t->gat.x.g[j], t->gat.n[j]);
  }
  return GIGL_OK;
}

}  // extern "C"
