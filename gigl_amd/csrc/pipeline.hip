// pipeline.hip — one-call batch pipeline: k-hop sample -> batch union graph -> GraphSAGE forward ->
// per-root rows.  Everything a batch needs is enqueued by ONE host call on the ctx stream (the host
// side of the hop loop / layer loop lives here, in C++, like the reference's compiled sampler).
//
// Replaces, for a RootedNodeNeighborhood batch (paths relative to the reference root):
//   NodeAnchorBasedLinkPredictionModelingTaskSpec.infer_batch
//       python/gigl/src/common/modeling_task_specs/node_anchor_based_link_prediction_modeling_task_spec.py:626-655
//       (`model(data)[root_node_indices]`), and NodeClassificationModelingTaskSpec.infer_batch
//       (.../node_classification_modeling_task_spec.py:176-187),
//   fed by process_raw_pyg_samples_and_collate_fn (rooted_node_neighborhood_data_loader.py:161-241).
#include "common.h"

#include <new>
#include <vector>

struct gigl_sage_plan {
  gigl_ctx* ctx = nullptr;
  gigl_graph* graph = nullptr;
  gigl_feat* feat = nullptr;
  int32_t b = 0, hops = 0;
  int32_t fanouts[GIGL_MAX_HOPS] = {0};
  int32_t dims[GIGL_MAX_HOPS + 1] = {0};  // dims[0] = input dim, dims[l+1] = output dim of layer l
  const float* w[GIGL_MAX_HOPS] = {nullptr};     // fused [dims[l+1]][2*dims[l]] (= [W_l | W_r]), device, borrowed
  const float* bias[GIGL_MAX_HOPS] = {nullptr};  // device, borrowed, may be null
  int32_t act_last = 0;
  gigl_tree tree{};
  gigl_union un{};
  float* abuf = nullptr;  // [cap_nodes][2*max_in]
  float* hbuf[2] = {nullptr, nullptr};  // ping-pong [cap_nodes][max_out]
  std::vector<void*> owned;
};

namespace {

__global__ void take_rows_kernel(const float* __restrict__ h, const int32_t* __restrict__ root_local, int b, int d,
                                 float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)b * d) return;
  int r = (int)(i / d), c = (int)(i % d);
  int32_t l = root_local[r];
  out[i] = l >= 0 ? h[(int64_t)l * d + c] : 0.f;
}

}  // namespace

extern "C" {

int32_t gigl_sage_plan_destroy(gigl_sage_plan* p) {
  if (!p) return GIGL_OK;
  if (p->ctx) {
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
  }
  for (void* q : p->owned) hipFree(q);
  delete p;
  return GIGL_OK;
}

int32_t gigl_sage_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b,
                              const int32_t* fanouts, int32_t hops, const int32_t* dims,
                              const float* const* w, const float* const* bias, int32_t act_last,
                              gigl_sage_plan** out) {
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
  int64_t parents = b;
  for (int k = 0; k < hops && ok; ++k) {
    p->tree.cnt[k] = (int32_t*)alloc((size_t)parents * 4);
    parents *= fanouts[k];
    p->tree.nbr[k] = (uint32_t*)alloc((size_t)parents * 4);
    ok = p->tree.cnt[k] && p->tree.nbr[k];
  }
  p->un.meta = (int32_t*)alloc(GIGL_META_LEN * 4);
  p->un.nodes = (uint32_t*)alloc((size_t)cap_nodes * 4);
  p->un.rowptr = (int32_t*)alloc((size_t)(cap_nodes + 2) * 4);
  p->un.rowend = (int32_t*)alloc((size_t)(cap_nodes + 2) * 4);
  p->un.col = (int32_t*)alloc((size_t)cap_edges * 4);
  p->un.root_local = (int32_t*)alloc((size_t)b * 4);
  p->un.cap_nodes = cap_nodes;
  p->un.cap_edges = cap_edges;
  p->abuf = (float*)alloc((size_t)cap_nodes * 2 * max_in * 4);
  p->hbuf[0] = (float*)alloc((size_t)cap_nodes * max_out * 4);
  p->hbuf[1] = hops > 1 ? (float*)alloc((size_t)cap_nodes * max_out * 4) : p->hbuf[0];
  ok = ok && p->un.meta && p->un.nodes && p->un.rowptr && p->un.rowend && p->un.col && p->un.root_local &&
       p->abuf && p->hbuf[0] && p->hbuf[1];
  if (!ok) {
    gigl_sage_plan_destroy(p);
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of the batch workspace failed (cap_nodes=%lld)",
                     (long long)cap_nodes);
  }
  *out = p;
  return GIGL_OK;
}

int32_t gigl_sage_plan_set_weights(gigl_sage_plan* p, const float* const* w, const float* const* bias) {
  if (!p || !w) return GIGL_E_INVALID_ARG;
  for (int k = 0; k < p->hops; ++k) {
    if (!w[k]) return gigl_fail(p->ctx, GIGL_E_INVALID_ARG, "weight %d is null", k);
    p->w[k] = w[k];
    p->bias[k] = bias ? bias[k] : nullptr;
  }
  return GIGL_OK;
}

int32_t gigl_sage_plan_buffers(gigl_sage_plan* p, gigl_tree* tree, gigl_union* un) {
  if (!p) return GIGL_E_INVALID_ARG;
  if (tree) *tree = p->tree;
  if (un) *un = p->un;
  return GIGL_OK;
}

int32_t gigl_sage_plan_run(gigl_sage_plan* p, const uint32_t* roots, int32_t sampling_seed, int32_t mode,
                           float* out) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, roots && out, "null argument");
  int32_t rc = gigl_sample_khop(ctx, p->graph, roots, p->b, p->fanouts, p->hops, sampling_seed, mode, &p->tree);
  if (rc != GIGL_OK) return rc;
  rc = gigl_union_build(ctx, roots, &p->tree, &p->un);
  if (rc != GIGL_OK) return rc;
  const int L = p->hops;
  const float* h = nullptr;
  for (int l = 0; l < L; ++l) {
    const int32_t* n_rows = p->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    const int d = p->dims[l];
    if (l == 0)
      rc = gigl_gather_mean(ctx, p->feat->rows, p->feat->dtype, d, p->un.nodes, p->un.rowptr, p->un.rowend,
                            p->un.col, n_rows, p->un.cap_nodes, p->abuf);
    else
      rc = gigl_gather_mean(ctx, h, GIGL_DTYPE_F32, d, nullptr, p->un.rowptr, p->un.rowend, p->un.col, n_rows,
                            p->un.cap_nodes, p->abuf);
    if (rc != GIGL_OK) return rc;
    float* y = p->hbuf[l & 1];
    const int act = (l < L - 1 || p->act_last) ? 1 : 0;
    rc = gigl_linear(ctx, p->abuf, p->w[l], p->bias[l], n_rows, p->un.cap_nodes, 2 * d, p->dims[l + 1], act, y);
    if (rc != GIGL_OK) return rc;
    h = y;
  }
  const int dout = p->dims[L];
  const int64_t total = (int64_t)p->b * dout;
  hipLaunchKernelGGL(take_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, h,
                     p->un.root_local, p->b, dout, out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
