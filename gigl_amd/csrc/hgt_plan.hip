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
// A step is TWO captured parts over two plan workspaces: GRAPH (the typed plan: ops, numbering, merged CSR — latency-bound
// small sorts) on a side stream, LAYERS (projections, typed aggregate — bandwidth-bound) on the caller's stream; given
// the NEXT batch's roots, the graph part of batch i + 1 runs under the layers of batch i (as gigl_sage_train_plan does).
// Branches forked INSIDE one captured graph were measured not to overlap under this runtime's executor; two graphs on
// two streams do.
// Per batch (4,096 roots, DBLP-shaped graph): ~45 kernel launches of the plan + 30 of the layers; bound by the typed
// aggregate (HBM: two source rows per edge) and launch latency of the small sorts, not by the host.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <vector>

struct gigl_hgt_infer {
  gigl_ctx* ctx = nullptr;          // the caller's: the LAYERS part runs on its stream
  gigl_ctx* side = nullptr;         // own ctx + stream: the GRAPH part (both plan workspaces live on it)
  gigl_typed_plan* wp[2] = {nullptr, nullptr};  // two workspaces: batch i + 1 is built while batch i is encoded
  gigl_typed_plan_out po[2]{};
  gigl_typed_csr_out csr[2]{};
  gigl_hgt_model m{};
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
  uint32_t* roots[2] = {nullptr, nullptr};  // static copies of the batches' roots (what the captured launches read)
  float* out = nullptr;              // [b_max][out_dim] static result (copied to the caller's buffer)
  int32_t* b_dev = nullptr;          // the batch size as a device count
  // pipeline state
  int cur = 0;
  const uint32_t* fetched[2] = {nullptr, nullptr};  // whose GRAPH part a workspace holds (the caller's roots pointer)
  int32_t fetched_b[2] = {0, 0};
  hipEvent_t ev_graph[2] = {nullptr, nullptr};   // GRAPH part of the workspace done (side stream)
  hipEvent_t ev_layers[2] = {nullptr, nullptr};  // LAYERS part done with the workspace (caller's stream)
  hipGraphExec_t exec_graph[2] = {nullptr, nullptr}, exec_layers[2] = {nullptr, nullptr};
  bool warm_graph[2] = {false, false}, warm_layers[2] = {false, false};
  bool use_graph = true;
  int32_t cap_b = -1;
  uint64_t cap_arena_gen = 0, cap_side_gen = 0;
  // the layers' per-type / per-slot projections as grouped launches (gigl_linear_grouped): device descriptor table, and per
  // workspace and layer where its K|V, Q and output groups sit in it (GIGL_HGT_NO_GROUPED=1: one launch per product, A/B)
  gigl_linear_group* groups_dev = nullptr;
  struct GroupRange {
    int kv_off = 0, kv_n = 0, q_off = 0, q_n = 0, o_off = 0, o_n = 0;
    int64_t kv_cap = 0, q_cap = 0;
  } gr[2][GIGL_HGT_MAX_LAYERS];
  bool grouped = false;
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

// the descriptor table of the layers' grouped projections (after create and after every set_model: weight pointers)
int32_t hgt_build_groups(gigl_hgt_infer* p) {
  const gigl_hgt_model& m = p->m;
  const int64_t Fo = m.hid;
  const int L = m.n_layers;
  std::vector<gigl_linear_group> g;
  for (int k = 0; k < 2; ++k) {
    const gigl_typed_plan_out& po = p->po[k];
    for (int l = 0; l < L; ++l) {
      const gigl_hgt_layer_weights& lw = m.layer[l];
      const bool last = l == L - 1;
      const float* h = p->h[l & 1];
      float* hn = p->h[(l + 1) & 1];
      gigl_hgt_infer::GroupRange& r = p->gr[k][l];
      r = gigl_hgt_infer::GroupRange();
      r.kv_off = (int)g.size();
      for (int s = 0; s < m.n_slots; ++s) {
        if (last && p->slot_dst_j[s] != p->root_j) continue;
        const int sj = p->slot_src_j[s];
        const int32_t* n_dev = po.n_nodes + m.type_order[sj];
        const float* x = h + p->dst_off[sj] * Fo;
        g.push_back(gigl_linear_group{x, lw.wk[s], lw.bk[s], n_dev, p->ks + p->src_off[s] * Fo});
        g.push_back(gigl_linear_group{x, lw.wv[s], lw.bv[s], n_dev, p->vs + p->src_off[s] * Fo});
        if (p->cap[sj] > r.kv_cap) r.kv_cap = p->cap[sj];
      }
      r.kv_n = (int)g.size() - r.kv_off;
      if (last) continue;
      r.q_off = (int)g.size();
      for (int j = 0; j < m.n_types; ++j) {
        g.push_back(gigl_linear_group{h + p->dst_off[j] * Fo, lw.wq[j], lw.bq[j], po.n_nodes + m.type_order[j],
                                      p->qq + p->dst_off[j] * Fo});
        if (p->cap[j] > r.q_cap) r.q_cap = p->cap[j];
      }
      r.q_n = (int)g.size() - r.q_off;
      r.o_off = (int)g.size();
      for (int j = 0; j < m.n_types; ++j)
        g.push_back(gigl_linear_group{p->agg + p->dst_off[j] * Fo, lw.wout[j], lw.bout[j], po.n_nodes + m.type_order[j],
                                      hn + p->dst_off[j] * Fo});
      r.o_n = (int)g.size() - r.o_off;
    }
  }
  GIGL_HIP_CHECK(p->ctx, hipMemcpy(p->groups_dev, g.data(), g.size() * sizeof(gigl_linear_group), hipMemcpyHostToDevice));
  return GIGL_OK;
}

// GRAPH part of workspace k (side stream): the ops, the numbering, the merged CSR at capacity prefixes
int32_t hgt_graph_part(gigl_hgt_infer* p, int k, int32_t b) {
  const gigl_hgt_model& m = p->m;
  int32_t rc = gigl_typed_plan_run(p->wp[k], p->roots[k], b);
  if (rc == GIGL_OK)
    rc = gigl_typed_plan_merged_csr_ex(p->wp[k], b, m.type_order, m.n_types, m.slot_order, m.slot_etype, m.n_slots, 1,
                                       &p->csr[k]);
  if (rc != GIGL_OK) return gigl_fail(p->ctx, rc, "%s", gigl_last_error(p->side));
  return GIGL_OK;
}

// LAYERS part over workspace k (the caller's stream): input projections, the HGT layers, the roots' rows
int32_t hgt_layers_part(gigl_hgt_infer* p, int k, int32_t b) {
  gigl_ctx* ctx = p->ctx;
  const gigl_hgt_model& m = p->m;
  const gigl_typed_plan_out& po = p->po[k];
  const gigl_typed_csr_out& csr = p->csr[k];
  hipStream_t st = ctx->stream;
  const int Fo = m.hid, H = m.heads, D = Fo / H, L = m.n_layers;
  int32_t rc = GIGL_OK;
  gigl_fill_u32(st, (uint32_t*)p->b_dev, (uint32_t)b, 1);
  // ---- input projections + ReLU: h0[type block] = relu(W_in x + b_in) over the type's distinct nodes
  float* h = p->h[0];
  float* hn = p->h[1];
  for (int j = 0; j < m.n_types; ++j) {
    const int t = m.type_order[j];
    const int32_t* n_dev = po.n_nodes + t;
    const int d = m.feat[j] ? m.feat_dim[j] : 1;
    if (m.feat[j]) {
      if ((d & 3) == 0)
        hipLaunchKernelGGL(hgt_gather_rows_kernel<4>, grid_of(p->cap[j] * (d / 4)), dim3(TB), 0, st, m.feat[j], po.nodes[t],
                           n_dev, p->cap[j], d, p->xin);
      else
        hipLaunchKernelGGL(hgt_gather_rows_kernel<1>, grid_of(p->cap[j] * d), dim3(TB), 0, st, m.feat[j], po.nodes[t], n_dev,
                           p->cap[j], d, p->xin);
    } else {
      hipLaunchKernelGGL(hgt_ones_kernel, grid_of(p->cap[j]), dim3(TB), 0, st, p->xin, p->cap[j]);
    }
    rc = gigl_linear(ctx, p->xin, m.w_in[j], m.b_in[j], n_dev, p->cap[j], d, Fo, 1, h + p->dst_off[j] * Fo);
    if (rc != GIGL_OK) return rc;
  }
  for (int l = 0; l < L; ++l) {
    const gigl_hgt_layer_weights& lw = m.layer[l];
    const bool last = l == L - 1;
    const gigl_hgt_infer::GroupRange& gr = p->gr[k][l];
    // K / V blocks of the slots this layer reads (the last layer computes the roots' rows: only edges INTO their type)
    if (p->grouped) {
      rc = gigl_linear_grouped(ctx, p->groups_dev + gr.kv_off, gr.kv_n, gr.kv_cap, Fo, Fo, 0);
      if (rc != GIGL_OK) return rc;
    }
    for (int s = 0; s < m.n_slots && !p->grouped; ++s) {
      if (last && p->slot_dst_j[s] != p->root_j) continue;
      const int sj = p->slot_src_j[s];
      const int32_t* n_dev = po.n_nodes + m.type_order[sj];
      const float* x = h + p->dst_off[sj] * Fo;
      rc = gigl_linear(ctx, x, lw.wk[s], lw.bk[s], n_dev, p->cap[sj], Fo, Fo, 0, p->ks + p->src_off[s] * Fo);
      if (rc == GIGL_OK) rc = gigl_linear(ctx, x, lw.wv[s], lw.bv[s], n_dev, p->cap[sj], Fo, Fo, 0, p->vs + p->src_off[s] * Fo);
      if (rc != GIGL_OK) return rc;
    }
    if (!last) {
      if (p->grouped) {
        rc = gigl_linear_grouped(ctx, p->groups_dev + gr.q_off, gr.q_n, gr.q_cap, Fo, Fo, 0);
        if (rc != GIGL_OK) return rc;
      }
      for (int j = 0; j < m.n_types && !p->grouped; ++j) {
        rc = gigl_linear(ctx, h + p->dst_off[j] * Fo, lw.wq[j], lw.bq[j], po.n_nodes + m.type_order[j], p->cap[j], Fo, Fo, 0,
                         p->qq + p->dst_off[j] * Fo);
        if (rc != GIGL_OK) return rc;
      }
      rc = gigl_hgt_aggregate_act(ctx, p->qq, p->ks, p->vs, H, D, csr.rowptr, csr.col, csr.etype, lw.p_rel, p->rows_cap, 1,
                                  p->agg);
      if (rc != GIGL_OK) return rc;
      if (p->grouped) {
        rc = gigl_linear_grouped(ctx, p->groups_dev + gr.o_off, gr.o_n, gr.q_cap, Fo, Fo, 0);
        if (rc != GIGL_OK) return rc;
      }
      for (int j = 0; j < m.n_types; ++j) {
        const int32_t* n_dev = po.n_nodes + m.type_order[j];
        float* o = hn + p->dst_off[j] * Fo;
        if (!p->grouped) {
          rc = gigl_linear(ctx, p->agg + p->dst_off[j] * Fo, lw.wout[j], lw.bout[j], n_dev, p->cap[j], Fo, Fo, 0, o);
          if (rc != GIGL_OK) return rc;
        }
        if (lw.keep[j])
          hipLaunchKernelGGL(hgt_skip_kernel, grid_of(p->cap[j] * (Fo / 4)), dim3(TB), 0, st, o, h + p->dst_off[j] * Fo,
                             lw.keep[j], n_dev, p->cap[j], Fo);
      }
      float* t = h;
      h = hn;
      hn = t;
    } else {
      const int rj = p->root_j;
      // the roots' rows of the layer's input, their queries, their slices of the merged CSR (laid out by the plan)
      hipLaunchKernelGGL(hgt_take_rows_kernel, grid_of((int64_t)b * (Fo / 4)), dim3(TB), 0, st, h + p->dst_off[rj] * Fo,
                         po.root_index, b, Fo, p->xr);
      rc = gigl_linear(ctx, p->xr, lw.wq[rj], lw.bq[rj], p->b_dev, p->b_max, Fo, Fo, 0, p->qr);
      if (rc == GIGL_OK)
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

// run a part on `c`'s stream: eagerly the first time (workspace growth, table builds and kernel attributes cannot happen
// inside a capture), captured the second, replayed from then on
template <typename F>
int32_t hgt_run_part(gigl_hgt_infer* p, gigl_ctx* c, hipGraphExec_t* exec, bool* warm, bool graph_ok, F body) {
  hipStream_t st = c->stream;
  if (!graph_ok) return body();
  if (!*exec && !*warm) {
    const int32_t rc = body();
    if (rc != GIGL_OK) return rc;
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
      return gigl_fail(p->ctx, GIGL_E_HIP, "capturing a part of the typed inference step failed: %s", hipGetErrorString(err));
    }
  }
  GIGL_HIP_CHECK(p->ctx, hipGraphLaunch(*exec, st));
  return GIGL_OK;
}

void hgt_drop_graphs(gigl_hgt_infer* p, bool graph_parts, bool layer_parts) {
  for (int k = 0; k < 2; ++k) {
    if (graph_parts && p->exec_graph[k]) {
      hipGraphExecDestroy(p->exec_graph[k]);
      p->exec_graph[k] = nullptr;
    }
    if (layer_parts && p->exec_layers[k]) {
      hipGraphExecDestroy(p->exec_layers[k]);
      p->exec_layers[k] = nullptr;
    }
  }
}

// the GRAPH part of `roots` into workspace k, on the side stream; ev_graph[k] marks its end
int32_t hgt_issue_graph(gigl_hgt_infer* p, int k, const uint32_t* roots, int32_t b, bool graph_ok) {
  gigl_ctx* sc = p->side;
  // the workspace is free once the LAYERS part that last read it is done, and the caller's roots are valid at this point
  // of ITS stream: one event recorded there now says both (it sits behind that part, not behind work enqueued later)
  GIGL_HIP_CHECK(p->ctx, hipEventRecord(p->ev_layers[k], p->ctx->stream));
  GIGL_HIP_CHECK(p->ctx, hipStreamWaitEvent(sc->stream, p->ev_layers[k], 0));
  hipLaunchKernelGGL(hgt_copy_u32_kernel, grid_of(b), dim3(TB), 0, sc->stream, roots, (int64_t)b, p->roots[k]);
  const int32_t rc = hgt_run_part(p, sc, &p->exec_graph[k], &p->warm_graph[k], graph_ok, [&] { return hgt_graph_part(p, k, b); });
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(p->ctx, hipEventRecord(p->ev_graph[k], sc->stream));
  p->fetched[k] = roots;
  p->fetched_b[k] = b;
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
  if (p->side) hipStreamSynchronize(p->side->stream);
  hgt_drop_graphs(p, true, true);
  for (int k = 0; k < 2; ++k) {
    if (p->wp[k]) gigl_typed_plan_destroy(p->wp[k]);
    if (p->ev_graph[k]) hipEventDestroy(p->ev_graph[k]);
    if (p->ev_layers[k]) hipEventDestroy(p->ev_layers[k]);
  }
  if (p->side) gigl_ctx_destroy(p->side);
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
  if (memcmp(model, &p->m, sizeof(gigl_hgt_model)) != 0) {
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    hgt_drop_graphs(p, false, true);  // (the GRAPH parts read no weight)
    p->m = *model;
    if (p->grouped) return hgt_build_groups(p);
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
  p->m = m;
  p->b_max = b_max;
  int32_t rc = gigl_ctx_create(ctx->device, &p->side);
  // the caller's plan is the template: two workspaces of the same DAG on the side ctx
  for (int k = 0; k < 2 && rc == GIGL_OK; ++k) {
    rc = gigl_typed_plan_clone(plan, p->side, &p->wp[k]);
    if (rc == GIGL_OK) rc = gigl_typed_plan_buffers(p->wp[k], &p->po[k]);
    if (rc == GIGL_OK && (hipEventCreateWithFlags(&p->ev_graph[k], hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&p->ev_layers[k], hipEventDisableTiming) != hipSuccess))
      rc = GIGL_E_HIP;
  }
  if (rc != GIGL_OK) {
    const int32_t r2 = gigl_fail(ctx, rc, "hgt plan: side workspaces: %s", p->side ? gigl_last_error(p->side) : "ctx");
    gigl_hgt_infer_destroy(p);
    return r2;
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
    if (t < 0 || t >= 16 || p->po[0].nodes_cap[t] <= 0 || type_pos[t] >= 0) HGT_FAIL("hgt plan: node type %d is not one of the plan's", t);
    type_pos[t] = j;
    p->cap.push_back(p->po[0].nodes_cap[t]);
    p->dst_off.push_back(p->rows_cap);
    p->rows_cap += p->po[0].nodes_cap[t];
    const int64_t d = m.feat[j] ? m.feat_dim[j] : 1;
    if (d < 1 || !m.w_in[j]) HGT_FAIL("hgt plan: node type %d has no input projection", t);
    max_in = p->po[0].nodes_cap[t] * d > max_in ? p->po[0].nodes_cap[t] * d : max_in;
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
  HGT_ALLOC(p->roots[0], b_max);
  HGT_ALLOC(p->roots[1], b_max);
  HGT_ALLOC(p->out, (int64_t)b_max * m.out_dim);
  HGT_ALLOC(p->b_dev, 8);
  HGT_ALLOC(p->groups_dev, (int64_t)2 * m.n_layers * (2 * m.n_slots + 2 * m.n_types));
#undef HGT_ALLOC
  p->grouped = (Fo & 3) == 0 && getenv("GIGL_HGT_NO_GROUPED") == nullptr;
  if (p->grouped) {
    rc = hgt_build_groups(p);
    if (rc != GIGL_OK) {
      gigl_hgt_infer_destroy(p);
      return rc;
    }
  }
  // (both workspaces start free)
  for (int k = 0; k < 2; ++k) GIGL_HIP_CHECK(ctx, hipEventRecord(p->ev_layers[k], ctx->stream));
  *out = p;
  return GIGL_OK;
}

int32_t gigl_hgt_infer_run(gigl_hgt_infer* p, const uint32_t* roots, int32_t b, const uint32_t* roots_next, int32_t b_next,
                           float* out) {
  if (!p || !p->ctx) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, roots && out && b >= 1 && b <= p->b_max, "between 1 and %d roots", p->b_max);
  GIGL_REQUIRE(ctx, !roots_next || (b_next >= 1 && b_next <= p->b_max), "between 1 and %d next roots", p->b_max);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // graphs are captured for ONE batch size (the job's; a short last batch runs eagerly), off while the ctx's timers are on
  const bool timers = ctx->prof_mask != 0;
  if (p->cap_b < 0) p->cap_b = b;
  if ((p->exec_graph[0] || p->exec_graph[1]) && p->cap_side_gen != p->side->arena_gen) {
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(p->side->stream));
    hgt_drop_graphs(p, true, false);
  }
  if ((p->exec_layers[0] || p->exec_layers[1]) && p->cap_arena_gen != ctx->arena_gen) {
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(st));
    hgt_drop_graphs(p, false, true);
  }
  auto graph_ok = [&](int32_t bb) { return p->use_graph && !timers && bb == p->cap_b; };
  const int k = p->cur;
  int32_t rc = GIGL_OK;
  if (p->fetched[k] != roots || p->fetched_b[k] != b) {  // not prefetched by the previous call
    rc = hgt_issue_graph(p, k, roots, b, graph_ok(b));
    if (rc != GIGL_OK) return rc;
  }
  if (roots_next) {  // the next batch's GRAPH part starts now, under this batch's layers
    rc = hgt_issue_graph(p, 1 - k, roots_next, b_next, graph_ok(b_next));
    if (rc != GIGL_OK) return rc;
  }
  p->cap_side_gen = p->side->arena_gen;
  GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, p->ev_graph[k], 0));
  // (the legacy default stream cannot be captured: the layers then run eagerly, the graph part still replays)
  rc = hgt_run_part(p, ctx, &p->exec_layers[k], &p->warm_layers[k], graph_ok(b) && st != nullptr,
                    [&] { return hgt_layers_part(p, k, b); });
  if (rc != GIGL_OK) return rc;
  p->cap_arena_gen = ctx->arena_gen;
  GIGL_HIP_CHECK(ctx, hipEventRecord(p->ev_layers[k], st));
  p->fetched[k] = nullptr;
  p->cur = 1 - k;
  hipLaunchKernelGGL(hgt_copy_u32_kernel, grid_of((int64_t)b * p->m.out_dim), dim3(TB), 0, st, (const uint32_t*)p->out,
                     (int64_t)b * p->m.out_dim, (uint32_t*)out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
