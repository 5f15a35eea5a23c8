// hgt_plan.hip — the typed inference step in ONE library call: SamplingOp DAG -> typed batch graph -> HGT encoder -> the
// roots' rows, captured once as a hipGraph and replayed per batch.
//
// Replaces, per batch of roots on a typed graph (paths relative to the reference root):
//   the HGT encoder's forward           python/gigl/src/common/models/pyg/heterogeneous.py:18-120 (HGT: per-type input
//                                       projections + ReLU, HGTConv layers, output Linear, optional L2 normalise)
//   torch_geometric HGTConv             k / q / v projections per node type, per-edge-type relation transforms, softmax
//                                       over ALL in-edges of a destination, GELU, output projection, gated skip
//   the inference loop around it        python/gigl/src/inference/v1/lib/node_anchor_based_link_prediction_inferencer.py
// What gigl_amd/models_hetero.py::HGT.forward issues from Python — ~150 launches per batch (one projection per node
// type / edge type / layer, torch glue between them), each waiting for the host — is here a fixed sequence of the
// library's own kernels over CAPACITY-sized buffers: row counts stay on the device (gigl_linear takes its row count
// from the typed plan's n_nodes array; the merged CSR is laid out at capacity prefixes: gigl_typed_plan_merged_csr_ex),
// so nothing in a step depends on a number the host would have to read and the whole step replays as one hipGraph.
// Weights are the COMPOSED inference weights (k_rel(K(x)) and v_rel(V(x)) are linear in x: one matrix per edge type;
// the gated skip folded into the output projection), made once per parameter state by the caller.
//
// Per batch (4,096 roots, DBLP-shaped graph): ~40 kernel launches of the plan + 25 of the layers in one replay; bound by
// the typed aggregate (HBM: two source rows per edge) and launch latency of the small sorts, not by the host.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <vector>

struct gigl_hgt_infer {
  gigl_ctx* ctx = nullptr;
  gigl_typed_plan* plan = nullptr;
  gigl_hgt_model m{};
  gigl_typed_plan_out po{};
  int32_t b_max = 0, root_j = -1;
  std::vector<int64_t> cap;        // per used type j: row capacity
  std::vector<int64_t> dst_off;    // per used type j: first row of its block (capacity prefix)
  std::vector<int64_t> src_off;    // per listed slot
  std::vector<int32_t> slot_src_j, slot_dst_j;
  int64_t rows_cap = 0, src_rows = 0;
  float *h[2] = {nullptr, nullptr};  // [rows_cap][hid] activations (ping-pong), blocks at dst_off
  float *xin = nullptr;              // [max cap * max feat dim] gathered input rows of one type at a time
  float *ks = nullptr, *vs = nullptr, *qq = nullptr, *agg = nullptr;
  float *xr = nullptr, *qr = nullptr, *aggr = nullptr, *orow = nullptr;  // the roots' rows of the last layer
  uint32_t* roots = nullptr;         // static copy of the batch's roots (what the captured launches read)
  float* out = nullptr;              // [b_max][out_dim] static result (copied to the caller's buffer)
  int32_t* b_dev = nullptr;          // the batch size as a device count
  // the first layer's projections need the batch's NODES only: they run on a side stream (own ctx: own scratch) while the
  // main stream is still numbering the edges and merging them into the CSR — a fork / join inside the captured graph
  gigl_ctx* side = nullptr;
  hipEvent_t ev_nodes = nullptr, ev_proj = nullptr;
  uint64_t cap_side_gen = 0;
  hipGraphExec_t exec = nullptr;
  bool warm = false, use_graph = true;
  int32_t cap_b = -1;
  uint64_t cap_arena_gen = 0;
  std::vector<void*> owned;
};

namespace {

constexpr int TB = 256;
inline dim3 grid_of(int64_t n) { return dim3((unsigned)((n + TB - 1) / TB > 0 ? (n + TB - 1) / TB : 1)); }

// out[i][:] = table[ids[i]][:] for i < *n (rows of d floats; rows at and beyond *n are left alone)
template <int V>
__global__ __launch_bounds__(TB) void hgt_gather_rows_kernel(const float* __restrict__ table, const uint32_t* __restrict__ ids,
                                                              const int32_t* __restrict__ n, int64_t cap, int d,
                                                              float* __restrict__ out) {
  const int64_t dv = d / V;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap * dv) return;
  const int64_t r = i / dv, c = (i - r * dv) * V;
  if (r >= *n) return;
  const int64_t src = ids ? (int64_t)ids[r] : r;
  if constexpr (V == 4) *(float4*)(out + r * d + c) = *(const float4*)(table + src * d + c);
  else out[r * d + c] = table[src * d + c];
}

// rows of ones (a node type without features: the reference feeds a constant 1-wide input)
__global__ __launch_bounds__(TB) void hgt_ones_kernel(float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = 1.f;
}

// out[i][:] = h[idx[i]][:], i < b (int32 row indices)
__global__ __launch_bounds__(TB) void hgt_take_rows_kernel(const float* __restrict__ h, const int32_t* __restrict__ idx, int b,
                                                           int d, float* __restrict__ out) {
  const int dv = d / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)b * dv) return;
  const int64_t r = i / dv, c = (i - r * dv) * 4;
  *(float4*)(out + r * d + c) = *(const float4*)(h + (int64_t)idx[r] * d + c);
}

// o[r][:] += keep[0] * x[r][:] for r < *n: the gated skip (the gate's other factor is folded into o's projection)
__global__ __launch_bounds__(TB) void hgt_skip_kernel(float* __restrict__ o, const float* __restrict__ x,
                                                      const float* __restrict__ keep, const int32_t* __restrict__ n,
                                                      int64_t cap, int d) {
  const int dv = d / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap * dv) return;
  const int64_t r = i / dv, c = (i - r * dv) * 4;
  if (r >= *n) return;
  const float k = keep[0];
  float4 a = *(float4*)(o + r * d + c);
  const float4 xv = *(const float4*)(x + r * d + c);
  a.x += k * xv.x;
  a.y += k * xv.y;
  a.z += k * xv.z;
  a.w += k * xv.w;
  *(float4*)(o + r * d + c) = a;
}

// rows scaled to unit L2 norm (F.normalize(p=2, dim=1), eps 1e-12): one wave per row
__global__ __launch_bounds__(TB) void hgt_l2_kernel(float* __restrict__ x, int b, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= b) return;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += x[r * d + c] * x[r * d + c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  for (int c = lane; c < d; c += 64) x[r * d + c] *= inv;
}

__global__ __launch_bounds__(TB) void hgt_copy_u32_kernel(const uint32_t* __restrict__ src, int64_t n, uint32_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

template <typename T>
int32_t dev_alloc(gigl_hgt_infer* p, T** out, int64_t count) {
  void* q = nullptr;
  if (hipMalloc(&q, (size_t)(count > 0 ? count : 1) * sizeof(T)) != hipSuccess)
    return gigl_fail(p->ctx, GIGL_E_OOM, "hgt plan: hipMalloc of %lld bytes failed", (long long)(count * sizeof(T)));
  p->owned.push_back(q);
  *out = (T*)q;
  return GIGL_OK;
}

// everything of one batch after the roots are in p->roots: the launches that are captured
int32_t hgt_body(gigl_hgt_infer* p, int32_t b) {
  gigl_ctx* ctx = p->ctx;
  const gigl_hgt_model& m = p->m;
  hipStream_t st = ctx->stream;
  const int Fo = m.hid, H = m.heads, D = Fo / H, L = m.n_layers;
  int32_t rc = gigl_typed_plan_run_nodes(p->plan, p->roots, b);
  if (rc != GIGL_OK) return rc;
  gigl_fill_u32(st, (uint32_t*)p->b_dev, (uint32_t)b, 1);
  float* h = p->h[0];
  float* hn = p->h[1];
  // the projections of layer l (what needs no edge): K / V blocks of the slots the layer reads (the last layer computes the
  // roots' rows: only edges INTO their type), the queries — every type's rows, or the roots' rows of the last layer
  auto project_layer = [&](int l, gigl_ctx* pc, const float* hin) -> int32_t {
    const gigl_hgt_layer_weights& lw = m.layer[l];
    const bool last = l == L - 1;
    int32_t r = GIGL_OK;
    for (int s = 0; s < m.n_slots; ++s) {
      if (last && p->slot_dst_j[s] != p->root_j) continue;
      const int sj = p->slot_src_j[s];
      const int32_t* n_dev = p->po.n_nodes + m.type_order[sj];
      const float* x = hin + p->dst_off[sj] * Fo;
      r = gigl_linear(pc, x, lw.wk[s], lw.bk[s], n_dev, p->cap[sj], Fo, Fo, 0, p->ks + p->src_off[s] * Fo);
      if (r == GIGL_OK) r = gigl_linear(pc, x, lw.wv[s], lw.bv[s], n_dev, p->cap[sj], Fo, Fo, 0, p->vs + p->src_off[s] * Fo);
      if (r != GIGL_OK) return r;
    }
    if (!last) {
      for (int j = 0; j < m.n_types; ++j) {
        r = gigl_linear(pc, hin + p->dst_off[j] * Fo, lw.wq[j], lw.bq[j], p->po.n_nodes + m.type_order[j], p->cap[j], Fo, Fo, 0,
                        p->qq + p->dst_off[j] * Fo);
        if (r != GIGL_OK) return r;
      }
    } else {
      const int rj = p->root_j;
      hipLaunchKernelGGL(hgt_take_rows_kernel, grid_of((int64_t)b * (Fo / 4)), dim3(TB), 0, pc->stream, hin + p->dst_off[rj] * Fo,
                         p->po.root_index, b, Fo, p->xr);
      r = gigl_linear(pc, p->xr, lw.wq[rj], lw.bq[rj], p->b_dev, p->b_max, Fo, Fo, 0, p->qr);
    }
    return r;
  };
  // ---- fork: input projections + ReLU (h0[type block] = relu(W_in x + b_in) over the type's distinct nodes) and the
  // first layer's projections on the side stream ...
  gigl_ctx* pc = (p->side && ctx->prof_mask == 0) ? p->side : ctx;  // (timed runs keep one stream: the timers are the ctx's)
  if (pc != ctx) {
    GIGL_HIP_CHECK(ctx, hipEventRecord(p->ev_nodes, st));
    GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(pc->stream, p->ev_nodes, 0));
  }
  for (int j = 0; j < m.n_types; ++j) {
    const int t = m.type_order[j];
    const int32_t* n_dev = p->po.n_nodes + t;
    const int d = m.feat[j] ? m.feat_dim[j] : 1;
    if (m.feat[j]) {
      if ((d & 3) == 0)
        hipLaunchKernelGGL(hgt_gather_rows_kernel<4>, grid_of(p->cap[j] * (d / 4)), dim3(TB), 0, pc->stream, m.feat[j],
                           p->po.nodes[t], n_dev, p->cap[j], d, p->xin);
      else
        hipLaunchKernelGGL(hgt_gather_rows_kernel<1>, grid_of(p->cap[j] * d), dim3(TB), 0, pc->stream, m.feat[j],
                           p->po.nodes[t], n_dev, p->cap[j], d, p->xin);
    } else {
      hipLaunchKernelGGL(hgt_ones_kernel, grid_of(p->cap[j]), dim3(TB), 0, pc->stream, p->xin, p->cap[j]);
    }
    rc = gigl_linear(pc, p->xin, m.w_in[j], m.b_in[j], n_dev, p->cap[j], d, Fo, 1, h + p->dst_off[j] * Fo);
    if (rc != GIGL_OK) return pc == ctx ? rc : gigl_fail(ctx, rc, "%s", gigl_last_error(pc));
  }
  rc = project_layer(0, pc, h);
  if (rc != GIGL_OK) return pc == ctx ? rc : gigl_fail(ctx, rc, "%s", gigl_last_error(pc));
  if (pc != ctx) GIGL_HIP_CHECK(ctx, hipEventRecord(p->ev_proj, pc->stream));
  // ---- ... while the main stream numbers the edges and merges them into the CSR by destination; join
  rc = gigl_typed_plan_run_edges(p->plan, b);
  if (rc != GIGL_OK) return rc;
  gigl_typed_csr_out csr{};
  rc = gigl_typed_plan_merged_csr_ex(p->plan, b, m.type_order, m.n_types, m.slot_order, m.slot_etype, m.n_slots, 1, &csr);
  if (rc != GIGL_OK) return rc;
  if (pc != ctx) GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, p->ev_proj, 0));
  for (int l = 0; l < L; ++l) {
    const gigl_hgt_layer_weights& lw = m.layer[l];
    const bool last = l == L - 1;
    if (l > 0) {
      rc = project_layer(l, ctx, h);
      if (rc != GIGL_OK) return rc;
    }
    if (!last) {
      rc = gigl_hgt_aggregate_act(ctx, p->qq, p->ks, p->vs, H, D, csr.rowptr, csr.col, csr.etype, lw.p_rel, p->rows_cap, 1,
                                  p->agg);
      if (rc != GIGL_OK) return rc;
      for (int j = 0; j < m.n_types; ++j) {
        const int32_t* n_dev = p->po.n_nodes + m.type_order[j];
        float* o = hn + p->dst_off[j] * Fo;
        rc = gigl_linear(ctx, p->agg + p->dst_off[j] * Fo, lw.wout[j], lw.bout[j], n_dev, p->cap[j], Fo, Fo, 0, o);
        if (rc != GIGL_OK) return rc;
        if (lw.keep[j])
          hipLaunchKernelGGL(hgt_skip_kernel, grid_of(p->cap[j] * (Fo / 4)), dim3(TB), 0, st, o, h + p->dst_off[j] * Fo,
                             lw.keep[j], n_dev, p->cap[j], Fo);
      }
      float* t = h;
      h = hn;
      hn = t;
    } else {
      const int rj = p->root_j;
      // the roots' queries against their slices of the merged CSR (laid out by the plan)
      rc = gigl_hgt_aggregate_act(ctx, p->qr, p->ks, p->vs, H, D, csr.root_rowptr, csr.root_col, csr.root_etype, lw.p_rel, b,
                                    1, p->aggr);
      if (rc == GIGL_OK) rc = gigl_linear(ctx, p->aggr, lw.wout[rj], lw.bout[rj], p->b_dev, p->b_max, Fo, Fo, 0, p->orow);
      if (rc != GIGL_OK) return rc;
      if (lw.keep[rj])
        hipLaunchKernelGGL(hgt_skip_kernel, grid_of((int64_t)b * (Fo / 4)), dim3(TB), 0, st, p->orow, p->xr, lw.keep[rj],
                           p->b_dev, (int64_t)b, Fo);
      rc = gigl_linear(ctx, p->orow, m.w_final, m.b_final, p->b_dev, p->b_max, Fo, m.out_dim, 0, p->out);
      if (rc != GIGL_OK) return rc;
      if (m.l2_normalize)
        hipLaunchKernelGGL(hgt_l2_kernel, grid_of((int64_t)b * 64), dim3(TB), 0, st, p->out, b, m.out_dim);
    }
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // namespace

extern "C" {

int32_t gigl_hgt_infer_destroy(gigl_hgt_infer* p) {
  if (!p) return GIGL_OK;
  if (p->ctx) {
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
  }
  if (p->exec) hipGraphExecDestroy(p->exec);
  if (p->side) gigl_ctx_destroy(p->side);
  if (p->ev_nodes) hipEventDestroy(p->ev_nodes);
  if (p->ev_proj) hipEventDestroy(p->ev_proj);
  for (void* q : p->owned) hipFree(q);
  delete p;
  return GIGL_OK;
}

int32_t gigl_hgt_infer_set_model(gigl_hgt_infer* p, const gigl_hgt_model* model) {
  if (!p || !model) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  const gigl_hgt_model& o = p->m;
  // the shapes are baked into the workspace and the captured launches: only the weight POINTERS may change
  GIGL_REQUIRE(ctx, model->n_types == o.n_types && model->n_slots == o.n_slots && model->n_layers == o.n_layers &&
                        model->heads == o.heads && model->hid == o.hid && model->out_dim == o.out_dim,
               "hgt plan: set_model changes the shape");
  bool same = true;
  same = same && memcmp(model, &p->m, sizeof(gigl_hgt_model)) == 0;
  if (!same) {
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (p->exec) {
      hipGraphExecDestroy(p->exec);
      p->exec = nullptr;
    }
    p->m = *model;
  }
  return GIGL_OK;
}

int32_t gigl_hgt_infer_use_graph(gigl_hgt_infer* p, int32_t enable) {
  if (!p) return GIGL_E_INVALID_ARG;
  p->use_graph = enable != 0;
  return GIGL_OK;
}

int32_t gigl_hgt_infer_create(gigl_ctx* ctx, gigl_typed_plan* plan, int32_t b_max, const gigl_hgt_model* model,
                              const int32_t* slot_src_type, const int32_t* slot_dst_type, gigl_hgt_infer** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, plan && model && slot_src_type && slot_dst_type && b_max >= 1, "null argument");
  const gigl_hgt_model& m = *model;
  GIGL_REQUIRE(ctx, m.n_types >= 1 && m.n_types <= 16 && m.n_slots >= 1 && m.n_slots <= 32 && m.n_layers >= 1 &&
                        m.n_layers <= GIGL_HGT_MAX_LAYERS,
               "hgt plan: between 1 and 16 node types, 32 edge slots, %d layers", GIGL_HGT_MAX_LAYERS);
  GIGL_REQUIRE(ctx, m.heads >= 1 && m.hid >= 4 && m.hid % m.heads == 0 && (m.hid / m.heads) % 4 == 0 && m.out_dim >= 1,
               "hgt plan: hidden width %d over %d heads", m.hid, m.heads);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_hgt_infer* p = new (std::nothrow) gigl_hgt_infer();
  if (!p) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  p->ctx = ctx;
  p->plan = plan;
  p->m = m;
  p->b_max = b_max;
  int32_t rc = gigl_typed_plan_buffers(plan, &p->po);
  if (rc != GIGL_OK) {
    delete p;
    return rc;
  }
#define HGT_FAIL(...)                                     \
  do {                                                    \
    rc = gigl_fail(ctx, GIGL_E_INVALID_ARG, __VA_ARGS__); \
    gigl_hgt_infer_destroy(p);                            \
    return rc;                                            \
  } while (0)
  int64_t max_in = 0;
  int type_pos[16];
  for (int t = 0; t < 16; ++t) type_pos[t] = -1;
  for (int j = 0; j < m.n_types; ++j) {
    const int t = m.type_order[j];
    if (t < 0 || t >= 16 || p->po.nodes_cap[t] <= 0 || type_pos[t] >= 0) HGT_FAIL("hgt plan: node type %d is not one of the plan's", t);
    type_pos[t] = j;
    p->cap.push_back(p->po.nodes_cap[t]);
    p->dst_off.push_back(p->rows_cap);
    p->rows_cap += p->po.nodes_cap[t];
    const int64_t d = m.feat[j] ? m.feat_dim[j] : 1;
    if (d < 1 || !m.w_in[j]) HGT_FAIL("hgt plan: node type %d has no input projection", t);
    max_in = p->po.nodes_cap[t] * d > max_in ? p->po.nodes_cap[t] * d : max_in;
  }
  for (int s = 0; s < m.n_slots; ++s) {
    const int sj = slot_src_type[s] >= 0 && slot_src_type[s] < 16 ? type_pos[slot_src_type[s]] : -1;
    const int dj = slot_dst_type[s] >= 0 && slot_dst_type[s] < 16 ? type_pos[slot_dst_type[s]] : -1;
    if (sj < 0 || dj < 0) HGT_FAIL("hgt plan: edge slot %d joins a node type that is not listed", m.slot_order[s]);
    p->slot_src_j.push_back(sj);
    p->slot_dst_j.push_back(dj);
    p->src_off.push_back(p->src_rows);
    p->src_rows += p->cap[sj];
  }
  // (the roots' type: the one whose block root_index points into — the caller lists it; found through the plan's CSR call)
  p->root_j = -1;
  for (int j = 0; j < m.n_types; ++j)
    if (m.type_order[j] == m.root_type) p->root_j = j;
  if (p->root_j < 0) HGT_FAIL("hgt plan: the roots' node type %d is not listed", m.root_type);
  if (p->rows_cap * m.hid >= ((int64_t)1 << 31) * 4 || p->src_rows >= ((int64_t)1 << 31)) HGT_FAIL("hgt plan: batch too large");
#undef HGT_FAIL
  const int64_t Fo = m.hid;
#define HGT_ALLOC(ptr, count)           \
  do {                                  \
    rc = dev_alloc(p, &(ptr), (count)); \
    if (rc != GIGL_OK) {                \
      gigl_hgt_infer_destroy(p);        \
      return rc;                        \
    }                                   \
  } while (0)
  HGT_ALLOC(p->h[0], p->rows_cap * Fo);
  HGT_ALLOC(p->h[1], p->rows_cap * Fo);
  HGT_ALLOC(p->xin, max_in);
  HGT_ALLOC(p->ks, p->src_rows * Fo);
  HGT_ALLOC(p->vs, p->src_rows * Fo);
  HGT_ALLOC(p->qq, p->rows_cap * Fo);
  HGT_ALLOC(p->agg, p->rows_cap * Fo);
  HGT_ALLOC(p->xr, (int64_t)b_max * Fo);
  HGT_ALLOC(p->qr, (int64_t)b_max * Fo);
  HGT_ALLOC(p->aggr, (int64_t)b_max * Fo);
  HGT_ALLOC(p->orow, (int64_t)b_max * Fo);
  HGT_ALLOC(p->roots, b_max);
  HGT_ALLOC(p->out, (int64_t)b_max * m.out_dim);
  HGT_ALLOC(p->b_dev, 8);
#undef HGT_ALLOC
  {  // the side stream of the first layer's projections: OPT-IN (GIGL_HGT_SIDE_STREAM=1) — measured, the replayed graph
     // gains nothing from the fork (0.7457 against 0.7472 ms/step on typed-dblp: its branches do not run concurrently
     // under this runtime's graph executor), so the default keeps one stream
    const char* e = getenv("GIGL_HGT_SIDE_STREAM");
    if (e && e[0] == '1' && ctx->stream != nullptr) {
      if (gigl_ctx_create(ctx->device, &p->side) != GIGL_OK ||
          hipEventCreateWithFlags(&p->ev_nodes, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&p->ev_proj, hipEventDisableTiming) != hipSuccess) {
        gigl_hgt_infer_destroy(p);
        return gigl_fail(ctx, GIGL_E_HIP, "hgt plan: side stream");
      }
    }
  }
  *out = p;
  return GIGL_OK;
}

int32_t gigl_hgt_infer_run(gigl_hgt_infer* p, const uint32_t* roots, int32_t b, float* out) {
  if (!p || !p->ctx) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, roots && out && b >= 1 && b <= p->b_max, "between 1 and %d roots", p->b_max);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(hgt_copy_u32_kernel, grid_of(b), dim3(TB), 0, st, roots, (int64_t)b, p->roots);
  int32_t rc = GIGL_OK;
  const bool graph_ok = p->use_graph && st != nullptr && ctx->prof_mask == 0;
  if (p->exec && (p->cap_b != b || p->cap_arena_gen != ctx->arena_gen || !graph_ok ||
                  (p->side && p->cap_side_gen != p->side->arena_gen))) {
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(st));
    hipGraphExecDestroy(p->exec);
    p->exec = nullptr;
  }
  if (!graph_ok || !p->warm || p->cap_b != b) {
    // eager: the first batch of a size (workspace growth, table builds and kernel attributes cannot happen in a capture)
    rc = hgt_body(p, b);
    if (rc != GIGL_OK) return rc;
    p->warm = true;
    p->cap_b = b;
  } else {
    if (!p->exec) {
      hipGraph_t graph = nullptr;
      hipError_t err = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      if (err == hipSuccess) {
        rc = hgt_body(p, b);
        const hipError_t e2 = hipStreamEndCapture(st, &graph);
        if (rc == GIGL_OK && e2 != hipSuccess) err = e2;
      }
      if (rc == GIGL_OK && err == hipSuccess) err = hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0);
      if (graph) hipGraphDestroy(graph);
      if (rc != GIGL_OK) return rc;
      if (err != hipSuccess) {
        p->exec = nullptr;
        return gigl_fail(ctx, GIGL_E_HIP, "capturing the typed inference step failed: %s", hipGetErrorString(err));
      }
      p->cap_arena_gen = ctx->arena_gen;
      p->cap_side_gen = p->side ? p->side->arena_gen : 0;
    }
    GIGL_HIP_CHECK(ctx, hipGraphLaunch(p->exec, st));
  }
  hipLaunchKernelGGL(hgt_copy_u32_kernel, grid_of((int64_t)b * p->m.out_dim), dim3(TB), 0, st, (const uint32_t*)p->out,
                     (int64_t)b * p->m.out_dim, (uint32_t*)out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
