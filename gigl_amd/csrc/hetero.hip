// hetero.hip — attention-weighted segmented reductions of the heterogeneous encoders (HGT, SimpleHGN).
//
// Replaces the message passing of (paths relative to the reference root)
//   HGTConv.forward / message      python/gigl/src/common/models/pyg/nn/conv/hgt_conv.py:161-244
//       alpha_e = <q_i, k'_j> * p_rel[type(e)] / sqrt(D), softmax over ALL in-edges of i (every edge type together:
//       the bipartite graph of :202-206), out_i = sum_e alpha_e v'_j
//   SimpleHGNConv.forward / message python/gigl/src/common/models/pyg/nn/conv/simplehgn_conv.py:113-180
//       alpha_e = softmax over the edges that share the SOURCE node (softmax(alpha, row), row = edge_index[0], :154-155)
//       of leaky_relu(a_l.h_src + a_r.h_dst + a_etype.W_etype(emb[type(e)]) [+ a_efeat.W_efeat e]), out_i = sum_e alpha_e h_src
// The dense parts (per-type k/q/v projections, the per-relation D x D transforms, the output projections) are GEMMs on
// gigl_linear; what is here is the part PyG runs as index_select + scatter: one wave per destination row, every lane
// owning float4 chunks of the H*D-wide row so all heads advance together and each source row is read once, coalesced.
// HBM-bound: per aggregated edge one (SimpleHGN) or two (HGT: k' and v') source rows of 4*H*D bytes.
#include "common.h"

#include <cstdlib>

namespace {

// lanes [g*LPH, (g+1)*LPH) hold the chunks of head g: sum over them (LPH a power of two <= 64)
__device__ __forceinline__ float head_sum(float v, int lph) { return gigl_group_sum(v, lph); }

// HGT: online softmax over the in-edges of row i (all edge types), dot-product logits.  HD = heads * dim floats per
// row, processed in passes of 256 floats (64 lanes x float4) when HD > 256 is not needed: HD <= 256 * PASSES.
template <int PASSES>
__global__ __launch_bounds__(256) void hgt_aggregate_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, int heads, int dim,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ etype,
                                                            const float* __restrict__ p_rel, int64_t n_dst,
                                                            float* __restrict__ out, int gelu) {
  const int lane = threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_dst) return;
  const int hd = heads * dim;
  const int lph = dim / 4;  // lanes per head (power of two, <= 64)
  const float inv_sqrt_d = 1.0f / sqrtf((float)dim);
  float4 qv[PASSES], acc[PASSES];
  float m[PASSES], s[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int c = (p * 64 + lane) * 4;
    qv[p] = c < hd ? *(const float4*)(q + i * hd + c) : make_float4(0, 0, 0, 0);
    acc[p] = make_float4(0, 0, 0, 0);
    m[p] = -INFINITY;
    s[p] = 0.f;
  }
  const int32_t e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int32_t e = e0; e < e1; ++e) {
    const int64_t j = col[e];
    const int t = etype ? etype[e] : 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int c = (p * 64 + lane) * 4;
      const bool in = c < hd;
      const float4 kv = in ? *(const float4*)(k + j * hd + c) : make_float4(0, 0, 0, 0);
      const float4 vv = in ? *(const float4*)(v + j * hd + c) : make_float4(0, 0, 0, 0);
      float d = qv[p].x * kv.x + qv[p].y * kv.y + qv[p].z * kv.z + qv[p].w * kv.w;
      d = head_sum(d, lph);
      const int h = in ? c / dim : 0;
      const float logit = d * (p_rel ? p_rel[t * heads + h] : 1.f) * inv_sqrt_d;
      const float mn = fmaxf(m[p], logit);
      const float r = expf(m[p] - mn), w = expf(logit - mn);  // (m = -inf at the first edge: r = 0)
      s[p] = s[p] * r + w;
      acc[p].x = acc[p].x * r + w * vv.x;
      acc[p].y = acc[p].y * r + w * vv.y;
      acc[p].z = acc[p].z * r + w * vv.z;
      acc[p].w = acc[p].w * r + w * vv.w;
      m[p] = mn;
    }
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int c = (p * 64 + lane) * 4;
    if (c >= hd) continue;
    const float inv = s[p] > 0.f ? 1.f / s[p] : 0.f;  // a row without in-edges aggregates to 0
    float4 o = make_float4(acc[p].x * inv, acc[p].y * inv, acc[p].z * inv, acc[p].w * inv);
    if (gelu) {  // the layer's activation (exact erf form: torch's F.gelu default) in the reduce's epilogue
      o.x = 0.5f * o.x * (1.f + erff(o.x * 0.70710678118654752f));
      o.y = 0.5f * o.y * (1.f + erff(o.y * 0.70710678118654752f));
      o.z = 0.5f * o.z * (1.f + erff(o.z * 0.70710678118654752f));
      o.w = 0.5f * o.w * (1.f + erff(o.w * 0.70710678118654752f));
    }
    *(float4*)(out + i * hd + c) = o;
  }
}

// The same for NARROW rows (heads * dim <= 128 floats): a row's float4 chunks fill only G = heads * dim / 4 lanes, so a
// wave takes 64 / G destination rows at once (one lane group each; the wave-per-row kernel above left three quarters of
// its lanes idle at 64-wide rows); a row's edge list is fetched a chunk at a time by its lane group, four edges' rows are in
// flight per step.  Per row the same operations
// in the same order as hgt_aggregate_kernel: identical bits.
template <int G>
__global__ __launch_bounds__(256) void hgt_aggregate_packed_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, int heads, int dim,
                                                                   const int32_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ col,
                                                                   const int32_t* __restrict__ etype,
                                                                   const float* __restrict__ p_rel, int64_t n_dst,
                                                                   float* __restrict__ out, int gelu) {
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t i = wave * RPW + lane / G;
  const bool row_ok = i < n_dst;
  const int c = (lane % G) * 4;
  const int hd = G * 4;
  const int lph = dim / 4;
  const int h = c / dim;
  const float inv_sqrt_d = 1.0f / sqrtf((float)dim);
  const float4 z4 = make_float4(0, 0, 0, 0);
  const float4 qv = row_ok ? *(const float4*)(q + i * hd + c) : z4;
  float4 acc = z4;
  float m = -INFINITY, s = 0.f;
  int32_t e = row_ok ? rowptr[i] : 0;
  const int32_t e1 = row_ok ? rowptr[i + 1] : 0;
  auto fold = [&](bool on, const float4& kv, const float4& vv, int t) {
    float d = qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w;
    d = head_sum(d, lph);  // (every lane takes part: the sums run on DPP row operations)
    if (!on) return;
    const float logit = d * (p_rel ? p_rel[t * heads + h] : 1.f) * inv_sqrt_d;
    const float mn = fmaxf(m, logit);
    const float r = expf(m - mn), w = expf(logit - mn);
    s = s * r + w;
    acc.x = acc.x * r + w * vv.x;
    acc.y = acc.y * r + w * vv.y;
    acc.z = acc.z * r + w * vv.z;
    acc.w = acc.w * r + w * vv.w;
    m = mn;
  };
  // a chunk of up to G edges of the row at a time: their source rows and types are fetched by the group's lanes in one
  // coalesced load each (not one dependent load per edge), then four edges' k / v rows are in flight per step
  const int gbase = (lane / G) * G, gl = lane % G;
  while (__any(e < e1)) {
    const int32_t ce = e + gl;
    const int32_t cj = ce < e1 ? col[ce] : 0;
    const int32_t ct = (ce < e1 && etype) ? etype[ce] : 0;
    const int32_t left = e1 - e;  // (<= 0 for a finished row)
#pragma unroll 1
    for (int x = 0; x < G; x += 4) {
      if (!__any(x < left)) break;
      int64_t jj[4];
      int tt[4];
      bool on[4];
      float4 kk[4], vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        on[u] = x + u < left;
        jj[u] = __shfl(cj, gbase + x + u);
        tt[u] = __shfl(ct, gbase + x + u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        kk[u] = on[u] ? *(const float4*)(k + jj[u] * hd + c) : z4;
        vv[u] = on[u] ? *(const float4*)(v + jj[u] * hd + c) : z4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) fold(on[u], kk[u], vv[u], tt[u]);
    }
    e += G;
  }
  if (!row_ok) return;
  const float inv = s > 0.f ? 1.f / s : 0.f;
  float4 o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (gelu) {
    o.x = 0.5f * o.x * (1.f + erff(o.x * 0.70710678118654752f));
    o.y = 0.5f * o.y * (1.f + erff(o.y * 0.70710678118654752f));
    o.z = 0.5f * o.z * (1.f + erff(o.z * 0.70710678118654752f));
    o.w = 0.5f * o.w * (1.f + erff(o.w * 0.70710678118654752f));
  }
  *(float4*)(out + i * hd + c) = o;
}

// out_i = sum_e alpha[e][h] * v[col[e]][h*D ..]  (alpha already normalised)
template <int PASSES>
__global__ __launch_bounds__(256) void weighted_aggregate_kernel(const float* __restrict__ alpha,
                                                                 const float* __restrict__ v, int heads, int dim,
                                                                 const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col, int64_t n_dst,
                                                                 float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_dst) return;
  const int hd = heads * dim;
  float4 acc[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) acc[p] = make_float4(0, 0, 0, 0);
  const int32_t e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int32_t e = e0; e < e1; ++e) {
    const int64_t j = col[e];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int c = (p * 64 + lane) * 4;
      if (c >= hd) continue;
      const float w = alpha[(int64_t)e * heads + c / dim];
      const float4 vv = *(const float4*)(v + j * hd + c);
      acc[p].x += w * vv.x;
      acc[p].y += w * vv.y;
      acc[p].z += w * vv.z;
      acc[p].w += w * vv.w;
    }
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int c = (p * 64 + lane) * 4;
    if (c < hd) *(float4*)(out + i * hd + c) = acc[p];
  }
}

// Backward of hgt_aggregate_kernel: one wave per destination row.  The softmax statistics are recomputed (first pass over
// the row's edges: running max and sum per head), then with alpha_e = exp(logit_e - m) / s and D_i = <dout_i, out_i>
// per head (= sum_e alpha_e <dout_i, v_e>):
//     dv[j]  += alpha_e * dout_i                        (atomics: a source row has many destinations)
//     dlogit  = alpha_e * (<dout_i, v_e> - D_i)
//     dq[i]  += dlogit * scale * k_e    dk[j] += dlogit * scale * q_i    dp_rel[t][h] += dlogit * <q_i, k_e> / sqrt(D)
// with scale = p_rel[t][h] / sqrt(D).  dq is accumulated in registers and written once.
template <int PASSES>
__global__ __launch_bounds__(256) void hgt_aggregate_backward_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int heads, int dim,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ etype,
    const float* __restrict__ p_rel, int64_t n_dst, const float* __restrict__ out, const float* __restrict__ dout,
    float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv, float* __restrict__ dp_rel) {
  const int lane = threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_dst) return;
  const int hd = heads * dim;
  const int lph = dim / 4;
  const float inv_sqrt_d = 1.0f / sqrtf((float)dim);
  float4 qv[PASSES], go[PASSES], gq[PASSES];
  float m[PASSES], s[PASSES], di[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int c = (p * 64 + lane) * 4;
    const bool in = c < hd;
    qv[p] = in ? *(const float4*)(q + i * hd + c) : make_float4(0, 0, 0, 0);
    go[p] = in ? *(const float4*)(dout + i * hd + c) : make_float4(0, 0, 0, 0);
    const float4 o = in ? *(const float4*)(out + i * hd + c) : make_float4(0, 0, 0, 0);
    di[p] = head_sum(go[p].x * o.x + go[p].y * o.y + go[p].z * o.z + go[p].w * o.w, lph);
    gq[p] = make_float4(0, 0, 0, 0);
    m[p] = -INFINITY;
    s[p] = 0.f;
  }
  const int32_t e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int32_t e = e0; e < e1; ++e) {  // softmax statistics
    const int64_t j = col[e];
    const int t = etype ? etype[e] : 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int c = (p * 64 + lane) * 4;
      const bool in = c < hd;
      const float4 kv = in ? *(const float4*)(k + j * hd + c) : make_float4(0, 0, 0, 0);
      const float d = head_sum(qv[p].x * kv.x + qv[p].y * kv.y + qv[p].z * kv.z + qv[p].w * kv.w, lph);
      const int h = in ? c / dim : 0;
      const float logit = d * (p_rel ? p_rel[t * heads + h] : 1.f) * inv_sqrt_d;
      const float mn = fmaxf(m[p], logit);
      s[p] = s[p] * expf(m[p] - mn) + expf(logit - mn);
      m[p] = mn;
    }
  }
  for (int32_t e = e0; e < e1; ++e) {
    const int64_t j = col[e];
    const int t = etype ? etype[e] : 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int c = (p * 64 + lane) * 4;
      if (c >= hd) continue;  // (whole head groups are in or out together: head_sum stays inside the active lanes)
      const float4 kv = *(const float4*)(k + j * hd + c);
      const float4 vv = *(const float4*)(v + j * hd + c);
      const float d = head_sum(qv[p].x * kv.x + qv[p].y * kv.y + qv[p].z * kv.z + qv[p].w * kv.w, lph);
      const int h = c / dim;
      const float pr = p_rel ? p_rel[t * heads + h] : 1.f;
      const float alpha = expf(d * pr * inv_sqrt_d - m[p]) / s[p];
      const float da = head_sum(go[p].x * vv.x + go[p].y * vv.y + go[p].z * vv.z + go[p].w * vv.w, lph);
      const float dl = alpha * (da - di[p]);
      const float sc = dl * pr * inv_sqrt_d;
      float* gv = dv + j * hd + c;
      atomicAdd(gv + 0, alpha * go[p].x);
      atomicAdd(gv + 1, alpha * go[p].y);
      atomicAdd(gv + 2, alpha * go[p].z);
      atomicAdd(gv + 3, alpha * go[p].w);
      float* gk = dk + j * hd + c;
      atomicAdd(gk + 0, sc * qv[p].x);
      atomicAdd(gk + 1, sc * qv[p].y);
      atomicAdd(gk + 2, sc * qv[p].z);
      atomicAdd(gk + 3, sc * qv[p].w);
      gq[p].x += sc * kv.x;
      gq[p].y += sc * kv.y;
      gq[p].z += sc * kv.z;
      gq[p].w += sc * kv.w;
      if (dp_rel && (c % dim) == 0) atomicAdd(dp_rel + t * heads + h, dl * d * inv_sqrt_d);
    }
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int c = (p * 64 + lane) * 4;
    if (c < hd) *(float4*)(dq + i * hd + c) = gq[p];
  }
}

// backward of weighted_aggregate_kernel: dalpha[e][h] = <dout_i[h], v_j[h]>, dv[j] += alpha[e][h] * dout_i
template <int PASSES>
__global__ __launch_bounds__(256) void weighted_aggregate_backward_kernel(
    const float* __restrict__ alpha, const float* __restrict__ v, int heads, int dim, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, int64_t n_dst, const float* __restrict__ dout, float* __restrict__ dalpha,
    float* __restrict__ dv) {
  const int lane = threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n_dst) return;
  const int hd = heads * dim;
  const int lph = dim / 4;
  float4 go[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int c = (p * 64 + lane) * 4;
    go[p] = c < hd ? *(const float4*)(dout + i * hd + c) : make_float4(0, 0, 0, 0);
  }
  const int32_t e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int32_t e = e0; e < e1; ++e) {
    const int64_t j = col[e];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int c = (p * 64 + lane) * 4;
      if (c >= hd) continue;
      const int h = c / dim;
      const float4 vv = *(const float4*)(v + j * hd + c);
      const float da = head_sum(go[p].x * vv.x + go[p].y * vv.y + go[p].z * vv.z + go[p].w * vv.w, lph);
      if ((c % dim) == 0) dalpha[(int64_t)e * heads + h] = da;
      const float w = alpha[(int64_t)e * heads + h];
      float* gv = dv + j * hd + c;
      atomicAdd(gv + 0, w * go[p].x);
      atomicAdd(gv + 1, w * go[p].y);
      atomicAdd(gv + 2, w * go[p].z);
      atomicAdd(gv + 3, w * go[p].w);
    }
  }
}

__device__ __forceinline__ void atomic_max_float(float* addr, float val) {
  // floats order like sign-magnitude integers: positive values as int, negative as reversed unsigned
  if (val >= 0.f) atomicMax((int*)addr, __float_as_int(val));
  else atomicMin((unsigned int*)addr, __float_as_uint(val));
}

// SimpleHGN logits per (edge, head) + the running maximum of their SOURCE group
__global__ __launch_bounds__(256) void shgn_logits_kernel(const float* __restrict__ hl, const float* __restrict__ hr,
                                                          const float* __restrict__ het, const float* __restrict__ hef,
                                                          const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                          const int32_t* __restrict__ etype, int64_t n_edges, int heads,
                                                          float slope, float* __restrict__ gmax,
                                                          float* __restrict__ alpha) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_edges * heads) return;
  const int64_t e = t / heads;
  const int h = (int)(t - e * heads);
  const int32_t s = src[e];
  float x = hl[(int64_t)s * heads + h] + hr[(int64_t)dst[e] * heads + h] + het[(int64_t)etype[e] * heads + h];
  if (hef) x += hef[t];
  x = x > 0.f ? x : x * slope;
  alpha[t] = x;
  atomic_max_float(&gmax[(int64_t)s * heads + h], x);
}

__global__ __launch_bounds__(256) void shgn_expsum_kernel(const int32_t* __restrict__ src, int64_t n_edges, int heads,
                                                          const float* __restrict__ gmax, float* __restrict__ gsum,
                                                          float* __restrict__ alpha) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_edges * heads) return;
  const int64_t e = t / heads;
  const int h = (int)(t - e * heads);
  const int64_t g = (int64_t)src[e] * heads + h;
  const float ex = expf(alpha[t] - gmax[g]);
  alpha[t] = ex;
  atomicAdd(&gsum[g], ex);
}

__global__ __launch_bounds__(256) void shgn_normalise_kernel(const int32_t* __restrict__ src, int64_t n_edges, int heads,
                                                             const float* __restrict__ gsum, float* __restrict__ alpha) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_edges * heads) return;
  const int64_t e = t / heads;
  const int h = (int)(t - e * heads);
  alpha[t] = alpha[t] / (gsum[(int64_t)src[e] * heads + h] + 1e-16f);  // torch_geometric.utils.softmax's epsilon
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, int64_t n, float v) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = v;
}

int32_t check_shape(gigl_ctx* ctx, int32_t heads, int32_t dim) {
  const int lph = dim / 4;
  if (heads < 1 || dim < 4 || dim % 4 || (lph & (lph - 1)) || lph > 64 || heads * dim > 1024)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED,
                     "heads=%d dim=%d: the attention kernels need dim %% 4 == 0, dim/4 a power of two <= 64 and "
                     "heads*dim <= 1024", heads, dim);
  return GIGL_OK;
}

}  // namespace

extern "C" {

int32_t gigl_hgt_aggregate(gigl_ctx* ctx, const float* q, const float* k, const float* v, int32_t heads, int32_t dim,
                           const int32_t* rowptr, const int32_t* col, const int32_t* etype, const float* p_rel,
                           int64_t n_dst, float* out) {
  return gigl_hgt_aggregate_act(ctx, q, k, v, heads, dim, rowptr, col, etype, p_rel, n_dst, 0, out);
}

int32_t gigl_hgt_aggregate_act(gigl_ctx* ctx, const float* q, const float* k, const float* v, int32_t heads, int32_t dim,
                               const int32_t* rowptr, const int32_t* col, const int32_t* etype, const float* p_rel,
                               int64_t n_dst, int32_t gelu, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, n_dst >= 0 && (n_dst == 0 || (q && k && v && rowptr && col && out)), "null argument");
  int32_t rc = check_shape(ctx, heads, dim);
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n_dst == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);  // (the typed graphs' segmented reduce: timed with the others)
  {  // narrow rows: several destination rows per wave (identical results)
    const int g = heads * dim / 4;
    static const bool no_pack = getenv("GIGL_HGT_NO_PACK") != nullptr;
    if (!no_pack && (g == 8 || g == 16 || g == 32) && dim / 4 <= 16) {
      const int rpw = 64 / g;
      const dim3 pgrid((unsigned)((n_dst + 4 * rpw - 1) / (4 * rpw))), pblock(256);
#define GIGL_LAUNCH_HGT_PACKED(G)                                                                                     \
  hipLaunchKernelGGL(hgt_aggregate_packed_kernel<G>, pgrid, pblock, 0, ctx->stream, q, k, v, heads, dim, rowptr, col, \
                     etype, p_rel, n_dst, out, gelu ? 1 : 0)
      if (g == 8) GIGL_LAUNCH_HGT_PACKED(8);
      else if (g == 16) GIGL_LAUNCH_HGT_PACKED(16);
      else GIGL_LAUNCH_HGT_PACKED(32);
#undef GIGL_LAUNCH_HGT_PACKED
      GIGL_HIP_CHECK(ctx, hipGetLastError());
      return GIGL_OK;
    }
  }
  const int passes = (heads * dim + 255) / 256;
  const dim3 grid((unsigned)((n_dst + 3) / 4)), block(256);
#define GIGL_LAUNCH_HGT(P)                                                                                          \
  hipLaunchKernelGGL(hgt_aggregate_kernel<P>, grid, block, 0, ctx->stream, q, k, v, heads, dim, rowptr, col, etype, \
                     p_rel, n_dst, out, gelu ? 1 : 0)
  if (passes == 1) GIGL_LAUNCH_HGT(1);
  else if (passes == 2) GIGL_LAUNCH_HGT(2);
  else GIGL_LAUNCH_HGT(4);
#undef GIGL_LAUNCH_HGT
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_weighted_aggregate(gigl_ctx* ctx, const float* alpha, const float* v, int32_t heads, int32_t dim,
                                const int32_t* rowptr, const int32_t* col, int64_t n_dst, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, n_dst >= 0 && (n_dst == 0 || (alpha && v && rowptr && col && out)), "null argument");
  int32_t rc = check_shape(ctx, heads, dim);
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n_dst == 0) return GIGL_OK;
  const int passes = (heads * dim + 255) / 256;
  const dim3 grid((unsigned)((n_dst + 3) / 4)), block(256);
#define GIGL_LAUNCH_WA(P)                                                                                         \
  hipLaunchKernelGGL(weighted_aggregate_kernel<P>, grid, block, 0, ctx->stream, alpha, v, heads, dim, rowptr, col, \
                     n_dst, out)
  if (passes == 1) GIGL_LAUNCH_WA(1);
  else if (passes == 2) GIGL_LAUNCH_WA(2);
  else GIGL_LAUNCH_WA(4);
#undef GIGL_LAUNCH_WA
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_simplehgn_alpha(gigl_ctx* ctx, const float* hl, const float* hr, const float* het, const float* hef,
                             const int32_t* src, const int32_t* dst, const int32_t* etype, int64_t n_edges,
                             int64_t n_nodes, int32_t heads, float negative_slope, float* group_scratch, float* alpha) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, n_edges >= 0 && n_nodes >= 0 && heads >= 1, "bad sizes");
  GIGL_REQUIRE(ctx, n_edges == 0 || (hl && hr && het && src && dst && etype && group_scratch && alpha), "null argument");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n_edges == 0) return GIGL_OK;
  float* gmax = group_scratch;
  float* gsum = group_scratch + n_nodes * heads;
  const int64_t ng = n_nodes * heads, ne = n_edges * heads;
  auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  hipLaunchKernelGGL(fill_kernel, grid(ng), dim3(256), 0, ctx->stream, gmax, ng, -INFINITY);
  GIGL_HIP_CHECK(ctx, hipMemsetAsync(gsum, 0, (size_t)ng * 4, ctx->stream));
  hipLaunchKernelGGL(shgn_logits_kernel, grid(ne), dim3(256), 0, ctx->stream, hl, hr, het, hef, src, dst, etype,
                     n_edges, heads, negative_slope, gmax, alpha);
  hipLaunchKernelGGL(shgn_expsum_kernel, grid(ne), dim3(256), 0, ctx->stream, src, n_edges, heads, gmax, gsum, alpha);
  hipLaunchKernelGGL(shgn_normalise_kernel, grid(ne), dim3(256), 0, ctx->stream, src, n_edges, heads, gsum, alpha);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_hgt_aggregate_backward(gigl_ctx* ctx, const float* q, const float* k, const float* v, int32_t heads,
                                    int32_t dim, const int32_t* rowptr, const int32_t* col, const int32_t* etype,
                                    const float* p_rel, int64_t n_dst, const float* out, const float* dout, float* dq,
                                    float* dk, float* dv, float* dp_rel) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, n_dst >= 0 && (n_dst == 0 || (q && k && v && rowptr && col && out && dout && dq && dk && dv)),
               "null argument");
  int32_t rc = check_shape(ctx, heads, dim);
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n_dst == 0) return GIGL_OK;
  const int passes = (heads * dim + 255) / 256;
  const dim3 grid((unsigned)((n_dst + 3) / 4)), block(256);
#define GIGL_LAUNCH_HGTB(P)                                                                                          \
  hipLaunchKernelGGL(hgt_aggregate_backward_kernel<P>, grid, block, 0, ctx->stream, q, k, v, heads, dim, rowptr, col, \
                     etype, p_rel, n_dst, out, dout, dq, dk, dv, dp_rel)
  if (passes == 1) GIGL_LAUNCH_HGTB(1);
  else if (passes == 2) GIGL_LAUNCH_HGTB(2);
  else GIGL_LAUNCH_HGTB(4);
#undef GIGL_LAUNCH_HGTB
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_weighted_aggregate_backward(gigl_ctx* ctx, const float* alpha, const float* v, int32_t heads, int32_t dim,
                                         const int32_t* rowptr, const int32_t* col, int64_t n_dst, const float* dout,
                                         float* dalpha, float* dv) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, n_dst >= 0 && (n_dst == 0 || (alpha && v && rowptr && col && dout && dalpha && dv)), "null argument");
  int32_t rc = check_shape(ctx, heads, dim);
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n_dst == 0) return GIGL_OK;
  const int passes = (heads * dim + 255) / 256;
  const dim3 grid((unsigned)((n_dst + 3) / 4)), block(256);
#define GIGL_LAUNCH_WAB(P)                                                                                            \
  hipLaunchKernelGGL(weighted_aggregate_backward_kernel<P>, grid, block, 0, ctx->stream, alpha, v, heads, dim, rowptr, \
                     col, n_dst, dout, dalpha, dv)
  if (passes == 1) GIGL_LAUNCH_WAB(1);
  else if (passes == 2) GIGL_LAUNCH_WAB(2);
  else GIGL_LAUNCH_WAB(4);
#undef GIGL_LAUNCH_WAB
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
