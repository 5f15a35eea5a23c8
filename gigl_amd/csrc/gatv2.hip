// Per-edge C-wide convolutions of the homogeneous zoo, forward and backward: GATv2Conv (this header), GINEConv
// (gigl_gine_aggregate, below) and TransformerConv with edge features (gigl_transformer_aggregate_edge, at the end).
//
// GATv2Conv attention + aggregation, forward and backward (PyG 2.5.3 GATv2Conv as configured by GATv2.init_conv_layers,
// python/gigl/src/common/models/pyg/homogeneous.py:346-386; edge features through the optional edge rows):
//   s_ij = xl_j + xr_i,  z_ij = <att_h, leaky_relu(s_ij)> per head,  alpha = softmax_j z_ij over the in-edges of i
//   (self loops removed, one added back),  out_i = sum_j alpha_ij xl_j  (heads concatenated) + bias.
// Unlike GATConv the logit is not a sum of two per-node scalars: every edge needs the C-wide elementwise pass, so the
// logits are formed from the source rows as they are read for the aggregation — one pass, online softmax.
// One wave per destination row; lane l of chunk row v owns the float4 of channels [4q, 4q+4), q = 64 v + l; a head
// is C/4 adjacent lanes (C/4 a power of two <= 64), so all heads advance together and a source row is read once.
#include "common.h"

namespace {

constexpr int U = 4;  // source rows in flight

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
__device__ __forceinline__ float4 leaky4(const float4& s, float slope) {
  return float4{s.x > 0.f ? s.x : slope * s.x, s.y > 0.f ? s.y : slope * s.y, s.z > 0.f ? s.z : slope * s.z,
                s.w > 0.f ? s.w : slope * s.w};
}
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) {
  return float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
}
// sum over the `group` adjacent lanes of a head, returned to each of them
__device__ __forceinline__ float head_sum(float p, int group) { return gigl_group_sum(p, group); }

template <int V>
__global__ __launch_bounds__(256) void gatv2_forward_kernel(
    const float* __restrict__ xl, const float* __restrict__ xr, const float* __restrict__ att,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ col,
    const int32_t* __restrict__ n_rows_dev, int HC, int group, float slope, const float* __restrict__ bias, int act,
    const float* __restrict__ xe, float* __restrict__ out) {
  // xe (optional): lin_edge(edge_attr) rows [edges][HC] in the CSR's edge order, added inside the leaky_relu; the added
  // self loop carries the MEAN of the row's rows (PyG fill_value="mean"; lin_edge has no bias, so mean(lin(e)) =
  // lin(mean e)) — known when the self loop is folded in, last.
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev, chunks = HC >> 2;
  const float4 zero4{0.f, 0.f, 0.f, 0.f};
  bool on[V];
  float4 a4[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    on[v] = v * 64 + lane < chunks;
    a4[v] = on[v] ? ((const float4*)att)[v * 64 + lane] : zero4;
  }
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    float4 r4[V], acc[V], se[V];
    float mx[V], den[V];
    int cnt = 0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      r4[v] = on[v] ? ((const float4*)(xr + (int64_t)i * HC))[v * 64 + lane] : zero4;
      acc[v] = zero4;
      se[v] = zero4;
      mx[v] = -INFINITY;
      den[v] = 0.f;
    }
    // the in-edges, then the self loop as one more "edge" (index m)
    for (int e = 0; e <= m; e += U) {
      int j[U];
      float4 x[U][V], ev[U][V];
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int ee = e + t;
        j[t] = ee < m ? col[e0 + ee] : (ee == m ? i : -1);
        if (ee < m && j[t] == i) j[t] = -1;  // self loops among the edges are dropped
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        if (j[t] < 0) continue;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          x[t][v] = on[v] ? ((const float4*)(xl + (int64_t)j[t] * HC))[v * 64 + lane] : zero4;
          ev[t][v] = zero4;
          if (xe && on[v] && e + t < m) ev[t][v] = ((const float4*)(xe + (int64_t)(e0 + e + t) * HC))[v * 64 + lane];
        }
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        if (j[t] < 0) continue;  // (wave-uniform)
        const bool self = e + t == m;
        if (!self) ++cnt;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if (self && xe && cnt > 0) {
            const float ic = 1.0f / (float)cnt;
            ev[t][v] = float4{se[v].x * ic, se[v].y * ic, se[v].z * ic, se[v].w * ic};
          } else if (!self) {
            se[v] = add4(se[v], ev[t][v]);
          }
          const float z = head_sum(dot4(a4[v], leaky4(add4(add4(x[t][v], r4[v]), ev[t][v]), slope)), group);
          const float nm = fmaxf(mx[v], z);
          const float sc = __expf(mx[v] - nm), pw = __expf(z - nm);
          den[v] = den[v] * sc + pw;
          acc[v].x = acc[v].x * sc + pw * x[t][v].x;
          acc[v].y = acc[v].y * sc + pw * x[t][v].y;
          acc[v].z = acc[v].z * sc + pw * x[t][v].z;
          acc[v].w = acc[v].w * sc + pw * x[t][v].w;
          mx[v] = nm;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (!on[v]) continue;
      const int q = v * 64 + lane;
      const float inv = 1.0f / (den[v] + 1e-16f);
      float4 o{acc[v].x * inv, acc[v].y * inv, acc[v].z * inv, acc[v].w * inv};
      if (bias) o = add4(o, ((const float4*)bias)[q]);
      if (act == 1) o = float4{fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
      ((float4*)(out + (int64_t)i * HC))[q] = o;
    }
  }
}

// backward: with g = d out_i (bias / activation already peeled off by the caller), S = <g, out_i> per head,
//   d alpha_ij = <g, xl_j>,  dz_ij = alpha_ij (d alpha_ij - S),  ds_ij = dz_ij * att * leaky'(s_ij)   (C-wide)
//   d xl_j += alpha_ij g + ds_ij  (atomics: a source has many destinations),  d xr_i = sum_j ds_ij,
//   d att  += sum_ij dz_ij * leaky(s_ij)   (per-wave partial sums, one atomic per lane component at the end).
// Pass 1 recomputes max / denominator, pass 2 the gradients; both read the source rows.
template <int V>
__global__ __launch_bounds__(256) void gatv2_backward_kernel(
    const float* __restrict__ xl, const float* __restrict__ xr, const float* __restrict__ att,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ col,
    const int32_t* __restrict__ n_rows_dev, int HC, int group, float slope, const float* __restrict__ out_pre,
    const float* __restrict__ dout, float* __restrict__ dxl, float* __restrict__ dxr, float* __restrict__ datt,
    const float* __restrict__ xe, float* __restrict__ dxe) {
  // xe / dxe as in the forward: d xe_e = ds_e + ds_self / cnt (the self loop's mean row), so pass 2 starts with the
  // self loop
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev, chunks = HC >> 2;
  const float4 zero4{0.f, 0.f, 0.f, 0.f};
  bool on[V];
  float4 a4[V], da4[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    on[v] = v * 64 + lane < chunks;
    a4[v] = on[v] ? ((const float4*)att)[v * 64 + lane] : zero4;
    da4[v] = zero4;
  }
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    float4 r4[V], g[V], dr[V];
    float mx[V], den[V], S[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int q = v * 64 + lane;
      r4[v] = on[v] ? ((const float4*)(xr + (int64_t)i * HC))[q] : zero4;
      g[v] = on[v] ? ((const float4*)(dout + (int64_t)i * HC))[q] : zero4;
      const float4 o = on[v] ? ((const float4*)(out_pre + (int64_t)i * HC))[q] : zero4;
      S[v] = head_sum(dot4(g[v], o), group);
      dr[v] = zero4;
      mx[v] = -INFINITY;
      den[v] = 0.f;
    }
    float4 se[V];
    int cnt = 0;
#pragma unroll
    for (int v = 0; v < V; ++v) se[v] = zero4;
    if (xe)
      for (int e = 0; e < m; ++e) {
        if (col[e0 + e] == i) continue;
        ++cnt;
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (on[v]) se[v] = add4(se[v], ((const float4*)(xe + (int64_t)(e0 + e) * HC))[v * 64 + lane]);
      }
    if (cnt > 0) {
      const float ic = 1.0f / (float)cnt;
#pragma unroll
      for (int v = 0; v < V; ++v) se[v] = float4{se[v].x * ic, se[v].y * ic, se[v].z * ic, se[v].w * ic};  // the mean row
    }
    auto edge_row = [&](int e, int v) -> float4 {
      if (!xe || !on[v]) return zero4;
      return e < m ? ((const float4*)(xe + (int64_t)(e0 + e) * HC))[v * 64 + lane] : se[v];
    };
    for (int e = 0; e <= m; ++e) {  // pass 1
      const int j = e < m ? col[e0 + e] : i;
      if (e < m && j == i) continue;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float4 x = on[v] ? ((const float4*)(xl + (int64_t)j * HC))[v * 64 + lane] : zero4;
        const float z = head_sum(dot4(a4[v], leaky4(add4(add4(x, r4[v]), edge_row(e, v)), slope)), group);
        const float nm = fmaxf(mx[v], z);
        den[v] = den[v] * __expf(mx[v] - nm) + __expf(z - nm);
        mx[v] = nm;
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) den[v] = 1.0f / (den[v] + 1e-16f);
    float4 ds_self[V];
#pragma unroll
    for (int v = 0; v < V; ++v) ds_self[v] = zero4;
    for (int ee = 0; ee <= m; ++ee) {  // pass 2: the self loop (index m) first, then the edges
      const int e = ee == 0 ? m : ee - 1;
      const int j = e < m ? col[e0 + e] : i;
      if (e < m && j == i) continue;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float4 x = on[v] ? ((const float4*)(xl + (int64_t)j * HC))[v * 64 + lane] : zero4;
        const float4 s = add4(add4(x, r4[v]), edge_row(e, v));
        const float4 l = leaky4(s, slope);
        const float z = head_sum(dot4(a4[v], l), group);
        const float al = __expf(z - mx[v]) * den[v];
        const float dal = head_sum(dot4(g[v], x), group);
        const float dz = al * (dal - S[v]);
        const float4 ds{dz * a4[v].x * (s.x > 0.f ? 1.f : slope), dz * a4[v].y * (s.y > 0.f ? 1.f : slope),
                        dz * a4[v].z * (s.z > 0.f ? 1.f : slope), dz * a4[v].w * (s.w > 0.f ? 1.f : slope)};
        dr[v] = add4(dr[v], ds);
        da4[v].x += dz * l.x;
        da4[v].y += dz * l.y;
        da4[v].z += dz * l.z;
        da4[v].w += dz * l.w;
        if (e == m) ds_self[v] = ds;
        if (on[v]) {
          float* o = dxl + (int64_t)j * HC + 4 * (v * 64 + lane);
          atomicAdd(o + 0, al * g[v].x + ds.x);
          atomicAdd(o + 1, al * g[v].y + ds.y);
          atomicAdd(o + 2, al * g[v].z + ds.z);
          atomicAdd(o + 3, al * g[v].w + ds.w);
          if (dxe && e < m) {
            const float ic = cnt > 0 ? 1.0f / (float)cnt : 0.f;
            ((float4*)(dxe + (int64_t)(e0 + e) * HC))[v * 64 + lane] =
                float4{ds.x + ds_self[v].x * ic, ds.y + ds_self[v].y * ic, ds.z + ds_self[v].z * ic,
                       ds.w + ds_self[v].w * ic};
          }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
      if (on[v]) ((float4*)(dxr + (int64_t)i * HC))[v * 64 + lane] = dr[v];
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    if (!on[v]) continue;
    float* o = datt + 4 * (v * 64 + lane);
    atomicAdd(o + 0, da4[v].x);
    atomicAdd(o + 1, da4[v].y);
    atomicAdd(o + 2, da4[v].z);
    atomicAdd(o + 3, da4[v].w);
  }
}

bool gatv2_shape(gigl_ctx* ctx, int heads, int C, int& V, int& group) {
  const int gl = C / 4, chunks = heads * C / 4;
  V = (chunks + 63) / 64;
  group = gl;
  const bool ok = C % 4 == 0 && gl >= 1 && gl <= 64 && (gl & (gl - 1)) == 0 && (V == 1 || V == 2 || V == 4);
  if (!ok)
    gigl_fail(ctx, GIGL_E_UNSUPPORTED, "heads=%d channels=%d: the GATv2 kernels need channels %% 4 == 0, channels/4 a "
              "power of two <= 64 and heads*channels <= 1024", heads, C);
  return ok;
}

}  // namespace

int32_t gigl_gatv2_aggregate(gigl_ctx* ctx, const float* xl, const float* xr, const float* att, int32_t heads,
                             int32_t channels, float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                             const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, const float* bias,
                             int32_t act, float* out) {
  return gigl_gatv2_aggregate_edge(ctx, xl, xr, att, heads, channels, negative_slope, rowptr, rowend, col, n_rows_dev,
                                   rows_cap, bias, act, nullptr, out);
}

int32_t gigl_gatv2_aggregate_edge(gigl_ctx* ctx, const float* xl, const float* xr, const float* att, int32_t heads,
                                  int32_t channels, float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                                  const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, const float* bias,
                                  int32_t act, const float* edge_rows, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, xl && xr && att && rowptr && rowend && col && n_rows_dev && out, "null argument");
  GIGL_REQUIRE(ctx, heads > 0 && channels > 0 && rows_cap >= 0 && (act == 0 || act == 1), "bad sizes");
  int V, group;
  if (!gatv2_shape(ctx, heads, channels, V, group)) return GIGL_E_UNSUPPORTED;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const int HC = heads * channels;
#define GIGL_V2_FWD(VV)                                                                                              \
  hipLaunchKernelGGL((gatv2_forward_kernel<VV>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, xl, xr, att,    \
                     rowptr, rowend, col, n_rows_dev, HC, group, negative_slope, bias, act, edge_rows, out)
  if (V == 1) GIGL_V2_FWD(1);
  else if (V == 2) GIGL_V2_FWD(2);
  else GIGL_V2_FWD(4);
#undef GIGL_V2_FWD
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gatv2_aggregate_backward(gigl_ctx* ctx, const float* xl, const float* xr, const float* att,
                                      int32_t heads, int32_t channels, float negative_slope, const int32_t* rowptr,
                                      const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                      int64_t rows_cap, const float* out_pre, const float* dout, float* dxl,
                                      float* dxr, float* datt) {
  return gigl_gatv2_aggregate_edge_backward(ctx, xl, xr, att, heads, channels, negative_slope, rowptr, rowend, col,
                                            n_rows_dev, rows_cap, out_pre, dout, nullptr, dxl, dxr, datt, nullptr);
}

int32_t gigl_gatv2_aggregate_edge_backward(gigl_ctx* ctx, const float* xl, const float* xr, const float* att,
                                           int32_t heads, int32_t channels, float negative_slope,
                                           const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                           const int32_t* n_rows_dev, int64_t rows_cap, const float* out_pre,
                                           const float* dout, const float* edge_rows, float* dxl, float* dxr,
                                           float* datt, float* dedge_rows) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, !edge_rows == !dedge_rows, "edge_rows and dedge_rows come together");
  GIGL_REQUIRE(ctx, xl && xr && att && rowptr && rowend && col && n_rows_dev && out_pre && dout && dxl && dxr && datt,
               "null argument");
  GIGL_REQUIRE(ctx, heads > 0 && channels > 0 && rows_cap >= 0, "bad sizes");
  int V, group;
  if (!gatv2_shape(ctx, heads, channels, V, group)) return GIGL_E_UNSUPPORTED;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 4) blocks = 256 * 4;  // (every wave ends with one atomic per att component)
  const int HC = heads * channels;
#define GIGL_V2_BWD(VV)                                                                                              \
  hipLaunchKernelGGL((gatv2_backward_kernel<VV>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, xl, xr, att,   \
                     rowptr, rowend, col, n_rows_dev, HC, group, negative_slope, out_pre, dout, dxl, dxr, datt, edge_rows,  \
                     dedge_rows)
  if (V == 1) GIGL_V2_BWD(1);
  else if (V == 2) GIGL_V2_BWD(2);
  else GIGL_V2_BWD(4);
#undef GIGL_V2_BWD
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// ---- GINEConv aggregation (PyG 2.5.3 GINEConv as configured by GINE.init_conv_layers, homogeneous.py:252-297):
//   out_i = (1 + eps) x_i + sum_{e = (j -> i)} relu(x_j + ee_e),   ee = lin(edge_attr) rows in the CSR's edge order
// (the MLP that follows is gigl_linear).  One wave per destination row, lanes over the d channels.
// Backward: with m_e = [x_j + ee_e > 0]:  dee_e = g_i * m_e (row-owned),  dx_j += g_i * m_e (atomics),
// dx_i += (1 + eps) g_i (atomics: i is also a source of other rows),  deps += <g_i, x_i> (one atomic per row).
namespace {

__global__ __launch_bounds__(256) void gine_forward_kernel(const float* __restrict__ x, const float* __restrict__ ee,
                                                           const float* __restrict__ eps, int d,
                                                           const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ rowend,
                                                           const int32_t* __restrict__ col,
                                                           const int32_t* __restrict__ n_rows_dev,
                                                           float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev;
  const float self_w = 1.0f + *eps;
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    for (int el = lane; el < d; el += 64) {
      float acc = self_w * x[(int64_t)i * d + el];
      for (int e = 0; e < m; ++e) {
        const float v = x[(int64_t)col[e0 + e] * d + el] + ee[(int64_t)(e0 + e) * d + el];
        acc += v > 0.f ? v : 0.f;
      }
      out[(int64_t)i * d + el] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void gine_backward_kernel(const float* __restrict__ x, const float* __restrict__ ee,
                                                            const float* __restrict__ eps, int d,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ rowend,
                                                            const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ n_rows_dev,
                                                            const float* __restrict__ dout, float* __restrict__ dx,
                                                            float* __restrict__ dee, float* __restrict__ deps) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev;
  const float self_w = 1.0f + *eps;
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    float gx = 0.f;
    for (int el = lane; el < d; el += 64) {
      const float g = dout[(int64_t)i * d + el];
      gx += g * x[(int64_t)i * d + el];
      atomicAdd(&dx[(int64_t)i * d + el], self_w * g);
      for (int e = 0; e < m; ++e) {
        const int j = col[e0 + e];
        const float v = x[(int64_t)j * d + el] + ee[(int64_t)(e0 + e) * d + el];
        const float ge = v > 0.f ? g : 0.f;
        dee[(int64_t)(e0 + e) * d + el] = ge;
        if (ge != 0.f) atomicAdd(&dx[(int64_t)j * d + el], ge);
      }
    }
    for (int off = 32; off > 0; off >>= 1) gx += __shfl_xor(gx, off, 64);
    if (lane == 0) atomicAdd(deps, gx);
  }
}

}  // namespace

int32_t gigl_gine_aggregate(gigl_ctx* ctx, const float* x, const float* edge_rows, const float* eps, int32_t d,
                            const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                            const int32_t* n_rows_dev, int64_t rows_cap, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, x && edge_rows && eps && rowptr && rowend && col && n_rows_dev && out && d > 0 && rows_cap >= 0,
               "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gine_forward_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, edge_rows, eps, d, rowptr,
                     rowend, col, n_rows_dev, out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gine_aggregate_backward(gigl_ctx* ctx, const float* x, const float* edge_rows, const float* eps, int32_t d,
                                     const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                     const int32_t* n_rows_dev, int64_t rows_cap, const float* dout, float* dx,
                                     float* dedge_rows, float* deps) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, x && edge_rows && eps && rowptr && rowend && col && n_rows_dev && dout && dx && dedge_rows && deps &&
                        d > 0 && rows_cap >= 0, "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gine_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, edge_rows, eps, d,
                     rowptr, rowend, col, n_rows_dev, dout, dx, dedge_rows, deps);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// ---- TransformerConv attention with edge features (PyG 2.5.3 TransformerConv(edge_dim), Transformer.init_conv_layers,
// homogeneous.py:440-487): with xe = lin_edge(edge_attr) rows in the CSR's edge order,
//   alpha_e = softmax over the in-edges of i of <q_i, k_j + xe_e> / sqrt(C),  out_i = sum_e alpha_e (v_j + xe_e)
// (no self loops are added; an empty row gives 0).  Same lane layout as the GATv2 kernels: one wave per destination row,
// a head = C/4 adjacent lanes.  (Without edge features the layer runs on gigl_hgt_aggregate.)
namespace {

template <int V>
__global__ __launch_bounds__(256) void transformer_edge_forward_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ xe,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ col,
    const int32_t* __restrict__ n_rows_dev, int HC, int group, float scale, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev, chunks = HC >> 2;
  const float4 zero4{0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    float4 q4[V], acc[V];
    float mx[V], den[V];
    bool on[V];
#pragma unroll
    for (int c = 0; c < V; ++c) {
      on[c] = c * 64 + lane < chunks;
      q4[c] = on[c] ? ((const float4*)(q + (int64_t)i * HC))[c * 64 + lane] : zero4;
      acc[c] = zero4;
      mx[c] = -INFINITY;
      den[c] = 0.f;
    }
    for (int e = 0; e < m; ++e) {
      const int j = col[e0 + e];
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const int o = c * 64 + lane;
        const float4 ee = on[c] ? ((const float4*)(xe + (int64_t)(e0 + e) * HC))[o] : zero4;
        const float4 kk = add4(on[c] ? ((const float4*)(k + (int64_t)j * HC))[o] : zero4, ee);
        const float4 vv = add4(on[c] ? ((const float4*)(v + (int64_t)j * HC))[o] : zero4, ee);
        const float z = head_sum(dot4(q4[c], kk), group) * scale;
        const float nm = fmaxf(mx[c], z);
        const float sc = __expf(mx[c] - nm), pw = __expf(z - nm);
        den[c] = den[c] * sc + pw;
        acc[c] = float4{acc[c].x * sc + pw * vv.x, acc[c].y * sc + pw * vv.y, acc[c].z * sc + pw * vv.z,
                        acc[c].w * sc + pw * vv.w};
        mx[c] = nm;
      }
    }
#pragma unroll
    for (int c = 0; c < V; ++c) {
      if (!on[c]) continue;
      const float inv = m > 0 ? 1.0f / (den[c] + 1e-16f) : 0.f;
      ((float4*)(out + (int64_t)i * HC))[c * 64 + lane] =
          float4{acc[c].x * inv, acc[c].y * inv, acc[c].z * inv, acc[c].w * inv};
    }
  }
}

// backward: g = d out_i, S = <g, out_i> per head; per edge kk = k_j + xe_e, vv = v_j + xe_e,
//   d alpha_e = <g, vv>,  dz_e = alpha_e (d alpha_e - S),  dq_i += dz_e kk * scale,  dk_j += dz_e q_i * scale (atomics),
//   dv_j += alpha_e g (atomics),  dxe_e = dz_e q_i * scale + alpha_e g (row-owned).
template <int V>
__global__ __launch_bounds__(256) void transformer_edge_backward_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ xe,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ col,
    const int32_t* __restrict__ n_rows_dev, int HC, int group, float scale, const float* __restrict__ out,
    const float* __restrict__ dout, float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
    float* __restrict__ dxe) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev, chunks = HC >> 2;
  const float4 zero4{0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    float4 q4[V], g[V], dq4[V];
    float mx[V], den[V], S[V];
    bool on[V];
#pragma unroll
    for (int c = 0; c < V; ++c) {
      const int o = c * 64 + lane;
      on[c] = o < chunks;
      q4[c] = on[c] ? ((const float4*)(q + (int64_t)i * HC))[o] : zero4;
      g[c] = on[c] ? ((const float4*)(dout + (int64_t)i * HC))[o] : zero4;
      const float4 oo = on[c] ? ((const float4*)(out + (int64_t)i * HC))[o] : zero4;
      S[c] = head_sum(dot4(g[c], oo), group);
      dq4[c] = zero4;
      mx[c] = -INFINITY;
      den[c] = 0.f;
    }
    for (int e = 0; e < m; ++e) {  // pass 1: max / denominator
      const int j = col[e0 + e];
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const int o = c * 64 + lane;
        const float4 kk = add4(on[c] ? ((const float4*)(k + (int64_t)j * HC))[o] : zero4,
                               on[c] ? ((const float4*)(xe + (int64_t)(e0 + e) * HC))[o] : zero4);
        const float z = head_sum(dot4(q4[c], kk), group) * scale;
        const float nm = fmaxf(mx[c], z);
        den[c] = den[c] * __expf(mx[c] - nm) + __expf(z - nm);
        mx[c] = nm;
      }
    }
#pragma unroll
    for (int c = 0; c < V; ++c) den[c] = 1.0f / (den[c] + 1e-16f);
    for (int e = 0; e < m; ++e) {  // pass 2
      const int j = col[e0 + e];
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const int o = c * 64 + lane;
        const float4 ee = on[c] ? ((const float4*)(xe + (int64_t)(e0 + e) * HC))[o] : zero4;
        const float4 kk = add4(on[c] ? ((const float4*)(k + (int64_t)j * HC))[o] : zero4, ee);
        const float4 vv = add4(on[c] ? ((const float4*)(v + (int64_t)j * HC))[o] : zero4, ee);
        const float z = head_sum(dot4(q4[c], kk), group) * scale;
        const float al = __expf(z - mx[c]) * den[c];
        const float dz = al * (head_sum(dot4(g[c], vv), group) - S[c]) * scale;
        dq4[c] = float4{dq4[c].x + dz * kk.x, dq4[c].y + dz * kk.y, dq4[c].z + dz * kk.z, dq4[c].w + dz * kk.w};
        if (on[c]) {
          const float4 dkk{dz * q4[c].x, dz * q4[c].y, dz * q4[c].z, dz * q4[c].w};
          const float4 dvv{al * g[c].x, al * g[c].y, al * g[c].z, al * g[c].w};
          float* pk = dk + (int64_t)j * HC + 4 * o;
          float* pv = dv + (int64_t)j * HC + 4 * o;
          atomicAdd(pk + 0, dkk.x);
          atomicAdd(pk + 1, dkk.y);
          atomicAdd(pk + 2, dkk.z);
          atomicAdd(pk + 3, dkk.w);
          atomicAdd(pv + 0, dvv.x);
          atomicAdd(pv + 1, dvv.y);
          atomicAdd(pv + 2, dvv.z);
          atomicAdd(pv + 3, dvv.w);
          ((float4*)(dxe + (int64_t)(e0 + e) * HC))[o] = add4(dkk, dvv);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < V; ++c)
      if (on[c]) ((float4*)(dq + (int64_t)i * HC))[c * 64 + lane] = dq4[c];
  }
}

}  // namespace

int32_t gigl_transformer_aggregate_edge(gigl_ctx* ctx, const float* q, const float* k, const float* v,
                                        const float* edge_rows, int32_t heads, int32_t channels, const int32_t* rowptr,
                                        const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                        int64_t rows_cap, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, q && k && v && edge_rows && rowptr && rowend && col && n_rows_dev && out, "null argument");
  GIGL_REQUIRE(ctx, heads > 0 && channels > 0 && rows_cap >= 0, "bad sizes");
  int V, group;
  if (!gatv2_shape(ctx, heads, channels, V, group)) return GIGL_E_UNSUPPORTED;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const int HC = heads * channels;
  const float scale = 1.0f / sqrtf((float)channels);
#define GIGL_TR_FWD(VV)                                                                                                \
  hipLaunchKernelGGL((transformer_edge_forward_kernel<VV>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, q, k, v, \
                     edge_rows, rowptr, rowend, col, n_rows_dev, HC, group, scale, out)
  if (V == 1) GIGL_TR_FWD(1);
  else if (V == 2) GIGL_TR_FWD(2);
  else GIGL_TR_FWD(4);
#undef GIGL_TR_FWD
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_transformer_aggregate_edge_backward(gigl_ctx* ctx, const float* q, const float* k, const float* v,
                                                 const float* edge_rows, int32_t heads, int32_t channels,
                                                 const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                                 const int32_t* n_rows_dev, int64_t rows_cap, const float* out,
                                                 const float* dout, float* dq, float* dk, float* dv,
                                                 float* dedge_rows) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, q && k && v && edge_rows && rowptr && rowend && col && n_rows_dev && out && dout && dq && dk && dv &&
                        dedge_rows, "null argument");
  GIGL_REQUIRE(ctx, heads > 0 && channels > 0 && rows_cap >= 0, "bad sizes");
  int V, group;
  if (!gatv2_shape(ctx, heads, channels, V, group)) return GIGL_E_UNSUPPORTED;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const int HC = heads * channels;
  const float scale = 1.0f / sqrtf((float)channels);
#define GIGL_TR_BWD(VV)                                                                                                \
  hipLaunchKernelGGL((transformer_edge_backward_kernel<VV>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, q, k,  \
                     v, edge_rows, rowptr, rowend, col, n_rows_dev, HC, group, scale, out, dout, dq, dk, dv, dedge_rows)
  if (V == 1) GIGL_TR_BWD(1);
  else if (V == 2) GIGL_TR_BWD(2);
  else GIGL_TR_BWD(4);
#undef GIGL_TR_BWD
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}
