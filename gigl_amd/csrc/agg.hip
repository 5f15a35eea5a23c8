// agg.hip — message passing over the union graph: segmented gather + mean reduce (fused with
// feature hydration) and the dense fp32 MFMA projection.
//
// Replaces (paths relative to the reference root):
//   PyG SAGEConv as configured by GraphSAGE.init_conv_layers
//       python/gigl/src/common/models/pyg/homogeneous.py:171-202   (conv call at :122-126)
//       out_i = lin_l(mean_{j->i} x_j) + lin_r(x_i)                 (PyG 2.5.3 defaults: aggr=mean,
//       root_weight=True, bias in lin_l only, normalize=False)
//   hydrateNodes (feature join)  scala/.../pureSpark/SGSPureSparkV1Task.scala:496-547
//
// gather_mean  HBM-bound: per aggregated edge 4 B (col) + d*s B (source row).  One wave per
//              destination row.  The row's source indices are fetched ONCE, 64 at a time, one per
//              lane (col -> gather_ids: two dependent loads for the whole row instead of per edge);
//              then LPR lanes stream each source row with 16-byte loads, G = 64/LPR source rows per
//              wave-instruction and 4 instructions in flight; fp32 accumulate; the destination's own
//              row is copied alongside, producing the [mean | self] operand of the projection.
// linear       exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): every wave owns a 32 x 32*NT output tile and
//              streams A / W rows straight from global (L2-resident W) as float4, register
//              double-buffered one K-step (8) ahead.  K is consumed 8 at a time: lane (r, h) loads
//              k0+4h..k0+4h+3 and the t-th MFMA of the group multiplies the k-pairs {k0+t, k0+4+t} of A
//              and W, so no LDS staging or transposes are needed.
#include "wave_reduce.h"
#include "common.h"

#include <cstring>

#include <cstdlib>
#include <type_traits>

#include <hip/hip_fp16.h>

// power of two s with s * m in [2^14, 2^15) (m > 0, finite), held inside [2^-60, 2^60]
__host__ __device__ static inline float hs_pow2_scale(float m) {
  uint32_t bits;
  memcpy(&bits, &m, 4);
  int e = (int)((bits >> 23) & 0xFF) - 127;  // floor(log2 m) for normal m
  if (((bits >> 23) & 0xFF) == 0) e = -126;
  int se = 14 - e;
  se = se > 60 ? 60 : (se < -60 ? -60 : se);
  const uint32_t sb = (uint32_t)(se + 127) << 23;
  float s;
  memcpy(&s, &sb, 4);
  return s;
}

namespace {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <typename T>
struct RowLoader;

template <>
struct RowLoader<float> {
  static __device__ __forceinline__ float4_t load4(const float* p, int e) {
    return *reinterpret_cast<const float4_t*>(p + e);
  }
};

template <>
struct RowLoader<__half> {
  static __device__ __forceinline__ float4_t load4(const __half* p, int e) {
    const __half2* q = reinterpret_cast<const __half2*>(p + e);
    float2 a = __half22float2(q[0]), b = __half22float2(q[1]);
    float4_t r = {a.x, a.y, b.x, b.y};
    return r;
  }
};

// LPR lanes cooperate on one source row; VPL float4 vectors per lane cover d (d % 4 == 0,
// d <= LPR*VPL*4)
// OP: GIGL_AGGR_MEAN / GIGL_AGGR_SUM / GIGL_AGGR_MAX (PyG aggr="mean" | "sum" | "max"; an empty row reduces to 0)
template <int OP>
__device__ __forceinline__ float4_t aggr_combine(float4_t a, float4_t b) {
  if constexpr (OP == GIGL_AGGR_MAX) {
    float4_t r = {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
    return r;
  } else {
    return a + b;
  }
}

// PROJ (the plan's projected-input first layer, pipeline.hip): `src` holds rows ALREADY multiplied by W_l (fp32, d =
// the layer's output width); the epilogue adds the destination's own W_r row (self_src[self]) and the bias, applies
// the activation and writes the finished layer output out[i][0:d] — no [mean | self] operand, no projection.
// PEER (the sharded plan's peer-mapped route, dist.hip): a non-negative source index is a GLOBAL node id v — its row is row
// v / world of rank (v % world)'s table, peers[v % world] (this rank's own table included; the other ranks' tables are
// mapped into this process: hipIpc handles between processes, plain device pointers inside one) — read where it lives,
// over xGMI for a peer's.  Nothing is claimed, requested, served or received; the sums run over the same rows in the same
// order as the bucketed route's, so the results are bit-identical.  Replicated hot rows stay -1-h (global_map).
template <typename T, int LPR, int VPL, int OP = GIGL_AGGR_MEAN, bool PROJ = false, bool PEER = false>
__global__ __launch_bounds__(256) void gather_mean_kernel(const T* __restrict__ src, int d,
                                                          const uint32_t* __restrict__ gather_ids,
                                                          const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ rowend,
                                                          const int32_t* __restrict__ col,
                                                          const int32_t* __restrict__ n_rows_dev,
                                                          float* __restrict__ out,
                                                          const int32_t* __restrict__ n_local_dev, int tiled_nkc,
                                                          const int32_t* __restrict__ global_map,
                                                          const T* __restrict__ src2, const T* __restrict__ src3,
                                                          const float* __restrict__ self_src = nullptr,
                                                          const float* __restrict__ bias = nullptr, int act = 0,
                                                          int ld = 0, int no_self = 0,
                                                          const int32_t* __restrict__ self_ids = nullptr, int ld3 = 0,
                                                          int own_world = 0, int own_rank = 0,
                                                          const T* const* __restrict__ peers = nullptr) {
  constexpr int G = 64 / LPR;  // source rows per wave-instruction
  const int64_t rs = PROJ ? (int64_t)ld : (int64_t)d;  // elements between source rows
  const int64_t rs3 = PROJ && ld3 ? (int64_t)ld3 : (int64_t)d;  // ... of src3 (the rank's own table) / of the peers' tables
  __shared__ const T* s_peers[PEER ? 64 : 1];
  if constexpr (PEER) {
    if ((int)threadIdx.x < own_world) s_peers[threadIdx.x] = peers[threadIdx.x];
    __syncthreads();
  }
  // a NEGATIVE row index -1-h names a row outside `src` (the sharded plan): h < 2^30 = row h of src2 (replicated hot
  // rows), else row h - 2^30 of src3 (this rank's own feature table)
  auto row_of = [&](int j) -> const T* {
    if (j >= 0) {
      if constexpr (PEER) {
        const uint32_t q = (uint32_t)j / (uint32_t)own_world;
        return s_peers[(uint32_t)j - q * (uint32_t)own_world] + (int64_t)q * rs3;
      } else {
        return src + (int64_t)j * rs;
      }
    }
    const int h = -1 - j;
    return h < (1 << 30) ? src2 + (int64_t)h * d : src3 + (int64_t)(h - (1 << 30)) * rs3;
  };
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR;  // which source row of the instruction
  const int sl = lane % LPR;   // lane within the row
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_rows = *n_rows_dev;
  // rows >= n_local hold GLOBAL source ids already (leaf-global union): no gather_ids translation for them
  const int n_local = n_local_dev ? *n_local_dev : 0x7FFFFFFF;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  constexpr float IDV = OP == GIGL_AGGR_MAX ? -__builtin_inff() : 0.f;  // identity of the reduction
  const float4_t zero4 = {IDV, IDV, IDV, IDV};
  const float4_t none4 = {0.f, 0.f, 0.f, 0.f};

  // source index of column entry `c` of destination row i
  auto translate = [&](int c, int i) -> int {
    if constexpr (PEER) {  // every source is named by its global id; global_map holds the replicated rows' marks only
      const int v = (gather_ids && i < n_local) ? (int)gather_ids[c] : c;
      if (global_map) {
        const int mk = global_map[(uint32_t)v];
        if (mk < 0) return mk;
      }
      return v;
    }
    if (gather_ids && i < n_local) return (int)gather_ids[c];
    if (global_map && i >= n_local) {  // global id -> its row in `src`
      // (own_world: the ids this rank owns — id % world == rank — are row id / world of src3, by arithmetic: they are
      // neither claimed nor looked up, dist.hip)
      if (own_world) {
        const uint32_t q = (uint32_t)c / (uint32_t)own_world;
        if ((uint32_t)c - q * (uint32_t)own_world == (uint32_t)own_rank) return -1 - ((1 << 30) + (int)q);
      }
      return global_map[(uint32_t)c];
    }
    return c;
  };
  // lane `idx` of group `sub`'s LPR lanes (grp), or entry idx + sub of the wave's index vector (wave-wide rows).
  // (v_readlane + select instead of the ds_bpermute behind __shfl was measured for G <= 2 — rmat-shard 12.8 -> 16.2 us /
  // step, products 9.56 -> 9.69: the scalar round trip stalls the loads it feeds; dropped)
  auto pick = [&](int v, int idx, bool grp) -> int {
    return grp ? __shfl(v, (lane & ~(LPR - 1)) + idx, 64) : __shfl(v, (idx + sub) & 63, 64);
  };
  // write the finished row i: `acc` holds the reduction in the lanes sl of group `sub` (grp: every group owns its own
  // row; else the whole wave owns row i and group 0 holds the total)
  auto emit = [&](int i, int m, float4_t (&acc)[VPL], bool grp) {
    const int self = gather_ids ? (int)gather_ids[i] : i;  // (PEER: the destination's global id)
    const bool writer = grp || sub == 0;
    // mean = sum / deg (a true division, like torch's scatter-mean), 0 for an empty row
    const float dv = (OP == GIGL_AGGR_MEAN && m > 0) ? (float)m : 1.f;
    const int c0 = (grp ? sl : lane) * 4, cstep = (grp ? LPR : 64) * 4;  // the self-row copy's lanes
    if constexpr (PROJ) {
      if (writer) {
        // (self_ids: the destination's W_r row sits at row self_ids[i] of self_src — the sharded plan's second receive
        // buffer — instead of at its source index)
        // (a NEGATIVE self_ids[i] = -1 - (2^30 + r): the destination is a node of this rank — its W_r row is the right half
        // of row r of src3, the rank's own [W_l x | W_r x] table, read in place)
        const int sj = self_ids ? self_ids[i] : self;
        const float* pr;
        if constexpr (PEER) pr = reinterpret_cast<const float*>(row_of(self)) + d;  // right half of the owner's [W_l x | W_r x] row
        else pr = sj >= 0 ? self_src + (int64_t)(uint32_t)sj * rs : src3 + (int64_t)((-1 - sj) - (1 << 30)) * rs3 + d;
        float* o = out + (int64_t)i * d;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int el = (v * LPR + sl) * 4;
          if (el < d) {
            float4_t r = ((OP == GIGL_AGGR_MAX && m == 0) ? none4 : acc[v] / dv) +
                         *reinterpret_cast<const float4_t*>(pr + el);
            if (bias) r += *reinterpret_cast<const float4_t*>(bias + el);
            if (act) r = float4_t{fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f)};
            *reinterpret_cast<float4_t*>(o + el) = r;
          }
        }
      }
      return;
    }
    const T* ps = row_of(self);
    if (tiled_nkc) {  // the projection's operand layout: [row tile of 128][K chunk of 32][128 rows][32 floats]
      float* tbase = out + ((int64_t)(i >> 7) * tiled_nkc) * 4096 + (i & 127) * 32;
      if (writer) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int el = (v * LPR + sl) * 4;
          if (el < d)
            *reinterpret_cast<float4_t*>(tbase + (int64_t)(el >> 5) * 4096 + (el & 31)) =
                (OP == GIGL_AGGR_MAX && m == 0) ? none4 : acc[v] / dv;
        }
      }
      // (no_self: the projection reads the self half of its operand straight from the source rows — two-source A
      // tile, linear_split_kernel<.., SELF> — so it is neither read nor written here)
      if (!no_self)
        for (int el = c0; el < d; el += cstep) {
          const int k = d + el;
          *reinterpret_cast<float4_t*>(tbase + (int64_t)(k >> 5) * 4096 + (k & 31)) = RowLoader<T>::load4(ps, el);
        }
      return;
    }
    float* o = out + (int64_t)i * 2 * d;
    if (writer) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int el = (v * LPR + sl) * 4;
        if (el < d) *reinterpret_cast<float4_t*>(o + el) = (OP == GIGL_AGGR_MAX && m == 0) ? none4 : acc[v] / dv;
      }
    }
    for (int el = c0; el < d; el += cstep)  // self row copy
      *reinterpret_cast<float4_t*>(o + d + el) = RowLoader<T>::load4(ps, el);
  };
  // one destination row on the whole wave: G source rows per instruction, four instructions in flight
  auto wave_row = [&](int i) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    float4_t acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = zero4;
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int mm = min(64, m - c0);
      int my = 0;
      if (lane < mm) my = translate(col[e0 + c0 + lane], i);
      for (int e = 0; e < mm; e += 4 * G) {  // wave-uniform trip count (the broadcasts need every lane)
        const int ea = e + sub, eb = ea + G, ec = ea + 2 * G, ed = ea + 3 * G;
        const int ja = pick(my, e, false), jb = pick(my, (e + G) & 63, false), jc = pick(my, (e + 2 * G) & 63, false),
                  jd = pick(my, (e + 3 * G) & 63, false);
        const T* pa = row_of(ja);
        const T* pb = row_of(jb);
        const T* pc = row_of(jc);
        const T* pd = row_of(jd);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int el = (v * LPR + sl) * 4;
          if (el < d) {
            float4_t a = ea < mm ? RowLoader<T>::load4(pa, el) : zero4;
            float4_t b = eb < mm ? RowLoader<T>::load4(pb, el) : zero4;
            float4_t c = ec < mm ? RowLoader<T>::load4(pc, el) : zero4;
            float4_t dd = ed < mm ? RowLoader<T>::load4(pd, el) : zero4;
            acc[v] = aggr_combine<OP>(acc[v], aggr_combine<OP>(aggr_combine<OP>(a, b), aggr_combine<OP>(c, dd)));
          }
        }
      }
    }
    // combine the G partial sums (lanes sl, sl+LPR, ...)
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const float4_t o4 = {__shfl_xor(acc[v].x, off, 64), __shfl_xor(acc[v].y, off, 64),
                             __shfl_xor(acc[v].z, off, 64), __shfl_xor(acc[v].w, off, 64)};
        acc[v] = aggr_combine<OP>(acc[v], o4);
      }
    }
    emit(i, m, acc, false);
  };

  // (GROUP_ROWS: measured on the MI355X — products-shaped 9.56 -> 9.73 us / step, rmat-shard 12.8 -> 14.3: the kernel
  // already moves its bytes at ~80 % of the HBM peak with a wave per row, more rows in flight per wave buy nothing and
  // the longer rows' second pass costs; kept for reference, compiled out)
  constexpr bool GROUP_ROWS = false;
  if constexpr (G == 1 || !GROUP_ROWS) {
    for (int i = wave; i < n_rows; i += waves_total) wave_row(i);
  } else {
    // Rows narrower than a wave (LPR < 64 lanes cover a source row): a wave takes G consecutive destination rows.  The
    // SHORT ones (<= SHORT_MAX in-edges — most rows of a sampled batch: a hop-1 node keeps at most f1 in-edges, on a
    // power-law graph usually a handful) are reduced side by side, one per lane group, four source rows in flight per
    // group: G times the rows in flight of a wave per row, and no cross-group combine.  Longer rows of the set then take
    // the whole wave one after another.  Which path a row takes depends on its own length only, so its bits never
    // depend on its neighbours (groups of batches == single batches).
    constexpr int SHORT_MAX = LPR < 16 ? LPR : 16;
    for (int base = wave * G; base < n_rows; base += waves_total * G) {
      const int ri = base + sub;
      int e0 = 0, m = 0;
      if (ri < n_rows) {
        e0 = rowptr[ri];
        m = rowend[ri] - e0;
      }
      const bool shortrow = ri < n_rows && m <= SHORT_MAX;
      const int ms = shortrow ? m : 0;
      int my = 0;
      if (sl < ms) my = translate(col[e0 + sl], ri);
      int mmax = 0;  // wave-uniform trip count: the longest short row of the set
#pragma unroll
      for (int g = 0; g < G; ++g) mmax = max(mmax, __builtin_amdgcn_readlane(ms, g * LPR));
      float4_t acc[VPL];
#pragma unroll
      for (int v = 0; v < VPL; ++v) acc[v] = zero4;
      for (int e = 0; e < mmax; e += 4) {
        const int ja = pick(my, e, true), jb = pick(my, e + 1, true), jc = pick(my, e + 2, true),
                  jd = pick(my, e + 3, true);
        const T* pa = row_of(ja);
        const T* pb = row_of(jb);
        const T* pc = row_of(jc);
        const T* pd = row_of(jd);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int el = (v * LPR + sl) * 4;
          if (el < d) {
            float4_t a = e < ms ? RowLoader<T>::load4(pa, el) : zero4;
            float4_t b = e + 1 < ms ? RowLoader<T>::load4(pb, el) : zero4;
            float4_t c = e + 2 < ms ? RowLoader<T>::load4(pc, el) : zero4;
            float4_t dd = e + 3 < ms ? RowLoader<T>::load4(pd, el) : zero4;
            acc[v] = aggr_combine<OP>(acc[v], aggr_combine<OP>(aggr_combine<OP>(a, b), aggr_combine<OP>(c, dd)));
          }
        }
      }
      if (shortrow) emit(ri, m, acc, true);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (base + g >= n_rows) break;
        if (__builtin_amdgcn_readlane(m, g * LPR) > SHORT_MAX) wave_row(base + g);
      }
    }
  }
}

// backward of the segmented reduce w.r.t. a dense local source matrix (layers >= 2; raw input features need no
// gradient):  dsrc[i] += dout[i][d:2d];  for every edge e of row i
//   mean: dsrc[col[e]] += dout[i][0:d] / deg_i      sum: += dout[i][0:d]
//   max:  += dout[i][0:d] / ties  for the sources that attain the maximum (torch's amax backward: the gradient
//         is shared evenly among ties); needs src to find them again.
// One wave per destination row; fp32 atomics (several rows can share a source).
__global__ __launch_bounds__(256) void gather_mean_backward_kernel(const float* __restrict__ dout, int d,
                                                                   const int32_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ rowend,
                                                                   const int32_t* __restrict__ col,
                                                                   const int32_t* __restrict__ n_rows_dev,
                                                                   float* __restrict__ dsrc, int op,
                                                                   const float* __restrict__ src, int wpr) {
  // wpr waves share a destination row (a power of two): wave q of the row takes the edges q, q + wpr, ... — a training
  // batch has ~10^3 rows at its last layer, a wave per row leaves most of the GPU idle behind 25 serial edges
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int part = gw & (wpr - 1);
  const int n_rows = *n_rows_dev;
  const int rows_step = ((gridDim.x * blockDim.x) >> 6) / wpr;
  for (int i = gw / wpr; i < n_rows; i += rows_step) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    const float* g = dout + (int64_t)i * 2 * d;
    if (part == 0)
      for (int el = lane; el < d; el += 64) atomicAdd(&dsrc[(int64_t)i * d + el], g[d + el]);
    if (m == 0) continue;
    if (op == GIGL_AGGR_MAX) {
      if (part != 0) continue;  // (ties need the whole row)
      for (int el = lane; el < d; el += 64) {
        float mx = -__builtin_inff();
        for (int e = 0; e < m; ++e) mx = fmaxf(mx, src[(int64_t)col[e0 + e] * d + el]);
        int ties = 0;
        for (int e = 0; e < m; ++e) ties += src[(int64_t)col[e0 + e] * d + el] == mx ? 1 : 0;
        const float share = g[el] / (float)ties;
        for (int e = 0; e < m; ++e) {
          const int j = col[e0 + e];
          if (src[(int64_t)j * d + el] == mx) atomicAdd(&dsrc[(int64_t)j * d + el], share);
        }
      }
      continue;
    }
    const float inv = op == GIGL_AGGR_MEAN ? 1.0f / (float)m : 1.0f;
    if ((d & 3) == 0 && d <= 1024) {  // the row's gradient stays in registers: float4 per lane, 4 atomics per edge
      float4_t gv[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int el = (v * 64 + lane) * 4;
        gv[v] = el < d ? *reinterpret_cast<const float4_t*>(g + el) * inv : float4_t{0.f, 0.f, 0.f, 0.f};
      }
      for (int e = part; e < m; e += wpr) {
        float* t = dsrc + (int64_t)col[e0 + e] * d;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int el = (v * 64 + lane) * 4;
          if (el < d) {
            atomicAdd(t + el, gv[v].x);
            atomicAdd(t + el + 1, gv[v].y);
            atomicAdd(t + el + 2, gv[v].z);
            atomicAdd(t + el + 3, gv[v].w);
          }
        }
      }
      continue;
    }
    for (int e = part; e < m; e += wpr) {
      const int j = col[e0 + e];
      for (int el = lane; el < d; el += 64) atomicAdd(&dsrc[(int64_t)j * d + el], g[el] * inv);
    }
  }
}

// ---- the same backward as a GATHER over the transposed rows (gigl_gather_mean_backward_transposed): who reads source j?
// Built per call from the rows' CSR — count, place, fill — then every source row is written once, without atomics on
// floats: dsrc[j] = dout[j][d:2d] (j < n_rows) + sum over the rows i that list j of dout[i][0:d] / deg_i.  A training
// batch's last layer has ~10^3-10^4 destination rows of ~25 edges each and ~10^5 sources, nearly all of them read by
// one row: the scatter spends its time in 26 M float atomics, the gather is a permuted copy.
__global__ __launch_bounds__(256) void gmt_count_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
                                                        const int32_t* __restrict__ col, const int32_t* __restrict__ n_rows_dev,
                                                        int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev;
  for (int i = wave; i < n_rows; i += waves) {
    const int e0 = rowptr[i], e1 = rowend[i];
    for (int e = e0 + lane; e < e1; e += 64) atomicAdd(&cnt[col[e]], 1);
  }
}
// list segments in arrival order (one atomic per wave on the cursor): ptr[j] = where source j's readers start
__global__ __launch_bounds__(256) void gmt_place_kernel(const int32_t* __restrict__ cnt, const int32_t* __restrict__ n_src_dev,
                                                        int32_t* __restrict__ cursor, int32_t* __restrict__ ptr) {
  const int lane = threadIdx.x & 63;
  const int n_src = *n_src_dev;
  const int stride = gridDim.x * blockDim.x;
  for (int j0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63); j0 < n_src; j0 += stride) {
    const int j = j0 + lane;
    const int c = j < n_src ? cnt[j] : 0;
    const int incl = gigl_wave_incl_scan(c);
    const int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane == 0 && total) base = atomicAdd(cursor, total);
    base = __shfl(base, 0, 64);
    if (j < n_src) ptr[j] = base + incl - c;
  }
}
__global__ __launch_bounds__(256) void gmt_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
                                                       const int32_t* __restrict__ col, const int32_t* __restrict__ n_rows_dev,
                                                       const int32_t* __restrict__ ptr, int32_t* __restrict__ fill,
                                                       int32_t* __restrict__ list) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev;
  for (int i = wave; i < n_rows; i += waves) {
    const int e0 = rowptr[i], e1 = rowend[i];
    for (int e = e0 + lane; e < e1; e += 64) {
      const int j = col[e];
      list[ptr[j] + atomicAdd(&fill[j], 1)] = i;
    }
  }
}
// LPR lanes per source row (a float4 each per pass), 64 / LPR rows per wave
template <int LPR>
__global__ __launch_bounds__(256) void gmt_gather_kernel(const float* __restrict__ dout, int d,
                                                         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
                                                         const int32_t* __restrict__ n_rows_dev,
                                                         const int32_t* __restrict__ n_src_dev, const int32_t* __restrict__ cnt,
                                                         const int32_t* __restrict__ ptr, const int32_t* __restrict__ list,
                                                         float* __restrict__ dsrc, int mean) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, l = lane % LPR;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev, n_src = *n_src_dev;
  for (int j = wave * G + sub; j < n_src; j += waves * G) {
    const int c = cnt[j], p0 = ptr[j];
    for (int el = 4 * l; el < d; el += 4 * LPR) {
      float4_t acc = j < n_rows ? *reinterpret_cast<const float4_t*>(dout + (int64_t)j * 2 * d + d + el)
                                : float4_t{0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < c; ++k) {
        const int i = list[p0 + k];
        const float inv = mean ? 1.0f / (float)(rowend[i] - rowptr[i]) : 1.0f;
        acc += *reinterpret_cast<const float4_t*>(dout + (int64_t)i * 2 * d + el) * inv;
      }
      *reinterpret_cast<float4_t*>(dsrc + (int64_t)j * d + el) = acc;
    }
  }
}

// generic fallback (any d): one wave per row, scalar elements
template <typename T>
__global__ __launch_bounds__(256) void gather_mean_generic_kernel(const T* __restrict__ src, int d,
                                                                  const uint32_t* __restrict__ gather_ids,
                                                                  const int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ rowend,
                                                                  const int32_t* __restrict__ col,
                                                                  const int32_t* __restrict__ n_rows_dev,
                                                                  float* __restrict__ out, int op,
                                                                  const int32_t* __restrict__ n_local_dev) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_rows = *n_rows_dev;
  const int n_local = n_local_dev ? *n_local_dev : 0x7FFFFFFF;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], e1 = rowend[i];
    const int deg = e1 - e0;
    float* o = out + (int64_t)i * 2 * d;
    const int self = gather_ids ? (int)gather_ids[i] : i;
    for (int el = lane; el < d; el += 64) {
      float acc = op == GIGL_AGGR_MAX ? -__builtin_inff() : 0.f;
      for (int e = e0; e < e1; ++e) {
        int j = col[e];
        if (gather_ids && i < n_local) j = (int)gather_ids[j];
        const float v = (float)src[(int64_t)j * d + el];
        acc = op == GIGL_AGGR_MAX ? fmaxf(acc, v) : acc + v;
      }
      o[el] = deg > 0 ? (op == GIGL_AGGR_MEAN ? acc / (float)deg : acc) : 0.f;
      o[d + el] = (float)src[(int64_t)self * d + el];
    }
  }
}

// ------------------------------------------------------------------------------------------
// y[m][n] = act(sum_k a[m][k] * w[n][k] + bias[n])       a: [M][K], w: [N][K] row-major, fp32
// ------------------------------------------------------------------------------------------
// Epilogue of one 32x32 accumulator block: bias / activation, then out through a 32 x 36-float LDS strip owned by the
// wave so that the block leaves as 16-byte stores of full 128-byte row segments (8 lanes per row) instead of 16 dword
// stores per lane — the row-per-lane pattern of the MFMA D layout is store-ISSUE bound (cdna guide T21).
// D layout (32x32, 16 regs): reg v -> row 8*(v/4) + 4*g + (v%4), col r.
__device__ __forceinline__ void store_block_32x32(const float16_t& acc, float* strip, int lane, const float* bias,
                                                  int act, int row0, int col0, int M, int N, float* y,
                                                  int ldy = 0) {
  const int r = lane & 31, g = lane >> 5;
  const float bv = (bias && col0 + r < N) ? bias[col0 + r] : 0.f;
  if (ldy == 0) ldy = N;  // (ldy: row stride of y when the N columns are a slice of wider rows)
  if (((N | ldy) & 3) != 0) {  // rows are not 16-byte aligned: dword stores
    if (col0 + r >= N) return;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = row0 + 8 * (v >> 2) + 4 * g + (v & 3);
      if (row < M) {
        float val = acc[v] + bv;
        if (act == 1) val = val > 0.f ? val : 0.f;
        y[(int64_t)row * ldy + col0 + r] = val;
      }
    }
    return;
  }
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    float val = acc[v] + bv;
    if (act == 1) val = val > 0.f ? val : 0.f;
    strip[(8 * (v >> 2) + 4 * g + (v & 3)) * 36 + r] = val;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const int c4 = (lane & 7) * 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int rr = (lane >> 3) + 8 * t;
    const int row = row0 + rr, col = col0 + c4;
    if (row < M && col < N) {  // (N % 4 == 0: a float4 is inside or outside as a whole)
      const float4_t q = *reinterpret_cast<const float4_t*>(&strip[rr * 36 + c4]);
      *reinterpret_cast<float4_t*>(y + (int64_t)row * ldy + col) = q;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

template <int NT>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          const int32_t* __restrict__ m_dev, int K,
                                                          int N, int act, float* __restrict__ y) {
  const int M = *m_dev;
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_n = (N + 32 * NT - 1) / (32 * NT);
  const int tiles_m = (M + 31) / 32;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= tiles_m * tiles_n) return;
  // consecutive waves share the A row-panel (tn fastest) so it stays in L1/L2
  const int tm = wave / tiles_n, tn = wave % tiles_n;
  const int m0 = tm * 32, n0 = tn * 32 * NT;

  float16_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;

  const bool aok = m0 + r < M;
  const float* ap = a + (int64_t)(aok ? m0 + r : 0) * K;
  const float* wp[NT];
  bool wok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = n0 + j * 32 + r;
    wok[j] = row < N;
    wp[j] = w + (int64_t)(wok[j] ? row : 0) * K;
  }
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool vec_ok = (K & 3) == 0;

  auto load_step = [&](int k0, float4_t& av, float4_t (&wv)[NT]) {
    const int kk = k0 + 4 * h;
    if (vec_ok) {
      const bool kin = kk < K;  // K % 4 == 0 -> whole float4 in range
      av = (aok && kin) ? *reinterpret_cast<const float4_t*>(ap + kk) : zero4;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wv[j] = (wok[j] && kin) ? *reinterpret_cast<const float4_t*>(wp[j] + kk) : zero4;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t] = (aok && kk + t < K) ? ap[kk + t] : 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) wv[j][t] = (wok[j] && kk + t < K) ? wp[j][kk + t] : 0.f;
    }
  };

  // PF K-steps of operands are kept in flight in a register ring (global -> L2/MALL latency is
  // ~1.5k cycles here, one K-step of MFMAs only 256*NT)
  constexpr int PF = 4;
  float4_t av_s[PF], wv_s[PF][NT];
#pragma unroll
  for (int s = 0; s < PF; ++s) load_step(8 * s, av_s[s], wv_s[s]);
  for (int k0 = 0; k0 < K; k0 += 8 * PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      if (k0 + 8 * s >= K) break;
      const float4_t av = av_s[s];
      float4_t wv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) wv[j] = wv_s[s][j];
      if (k0 + 8 * (s + PF) < K) load_step(k0 + 8 * (s + PF), av_s[s], wv_s[s]);  // refill this stage
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], wv[j][t], acc[j], 0, 0, 0);
    }
  }
  // D layout (32x32, 16 regs): reg v -> row 8*(v/4) + 4*h + (v%4), col r
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int coln = n0 + j * 32 + r;
    if (coln >= N) continue;
    const float bv = bias ? bias[coln] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = m0 + 8 * (v >> 2) + 4 * h + (v & 3);
      if (row < M) {
        float val = acc[j][v] + bv;
        if (act == 1) val = val > 0.f ? val : 0.f;
        y[(int64_t)row * N + coln] = val;
      }
    }
  }
}

// LDS-staged variant (K % 4 == 0): a workgroup of 4 waves computes a 128 x 32*NT tile, wave w the rows
// [32w, 32w+32).  Per K-chunk of 32 every wave fetches its own 32x32 A sub-tile and a quarter of the
// shared 32*NT x 32 W tile with fully coalesced 128-byte row segments (8 lanes x 16 B), parks them in
// LDS (row stride 36 floats: conflict-free ds_read_b128 for the MFMA operand layout) and reads them
// back in operand order.  Next chunk's global loads are issued before this chunk's MFMAs.
template <int NT>
__global__ __launch_bounds__(256) void linear_lds_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         const int32_t* __restrict__ m_dev, int K, int N,
                                                         int act, float* __restrict__ y) {
  constexpr int BK = 32, LDK = BK + 4;
  __shared__ float s_a[4][32 * LDK];
  __shared__ float s_w[32 * NT * LDK];
  const int M = *m_dev;
  const int tiles_n = (N + 32 * NT - 1) / (32 * NT);
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0b = tm * 128;
  if (m0b >= M) return;  // whole workgroup
  const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = m0b + wv_id * 32, n0 = tn * 32 * NT;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};

  float16_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;

  // loader mapping: lane -> (row lr + 8*i, k segment lc*4 .. lc*4+3), i = 0..3
  const int lr = lane >> 3, lc = lane & 7;
  // W loader: thread tid -> rows (tid>>3) + 32*i, i < NT
  const int wr = tid >> 3, wc = tid & 7;

  // operands of TWO chunks ahead are kept in flight in registers: a chunk's MFMAs take ~2k cycles, a global load
  // under load more (the kernel sat at 57 % of the fp32-MFMA peak with a one-chunk prefetch distance)
  float4_t ga[2][4], gw[2][NT];
  auto gload = [&](int k0, float4_t (&da)[4], float4_t (&dw)[NT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + lr + 8 * i, kk = k0 + lc * 4;
      da[i] = (row < M && kk < K) ? *reinterpret_cast<const float4_t*>(a + (int64_t)row * K + kk) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int row = n0 + wr + 32 * i, kk = k0 + wc * 4;
      dw[i] = (row < N && kk < K) ? *reinterpret_cast<const float4_t*>(w + (int64_t)row * K + kk) : zero4;
    }
  };
  auto chunk = [&](int k0, float4_t (&ra)[4], float4_t (&rw)[NT]) {
    __syncthreads();  // previous chunk's LDS reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4_t*>(&s_a[wv_id][(lr + 8 * i) * LDK + lc * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < NT; ++i) *reinterpret_cast<float4_t*>(&s_w[(wr + 32 * i) * LDK + wc * 4]) = rw[i];
    __syncthreads();
    if (k0 + 2 * BK < K) gload(k0 + 2 * BK, ra, rw);  // lands during this chunk's and the next chunk's MFMAs
#pragma unroll
    for (int ks = 0; ks < BK; ks += 8) {
      if (k0 + ks >= K) break;
      const float4_t av = *reinterpret_cast<const float4_t*>(&s_a[wv_id][r * LDK + ks + 4 * h]);
      float4_t wv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wv[j] = *reinterpret_cast<const float4_t*>(&s_w[(j * 32 + r) * LDK + ks + 4 * h]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], wv[j][t], acc[j], 0, 0, 0);
    }
  };
  gload(0, ga[0], gw[0]);
  if (BK < K) gload(BK, ga[1], gw[1]);
  for (int k0 = 0; k0 < K; k0 += 2 * BK) {
    chunk(k0, ga[0], gw[0]);
    if (k0 + BK < K) chunk(k0 + BK, ga[1], gw[1]);
  }
  // (s_a[wv_id] is read by this wave only: it is free for the epilogue once the wave's own reads are done)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < NT; ++j) store_block_32x32(acc[j], s_a[wv_id], lane, bias, act, m0, n0 + j * 32, M, N, y);
}

// Weight gradient of the projection: dW[n][k] += sum_{r < m} dy[r][n] * a[r][k]  (and db[n] += sum_r dy[r][n]), the ROWS
// being the inner dimension — a small output over tens of thousands of rows, of which only the first *m_dev are real.
// grid = (row chunks of RC rows) x (64-wide n tiles) x (64-wide k tiles): a workgroup stages 32 rows of its dy and a
// tiles in LDS at a time, every thread keeps a 4 x 4 block of the 64 x 64 partial in registers and adds it to dW with
// fp32 atomics at the end (one pass over the chunk's rows: the chunks are what parallelises the reduction).  With
// relu_y, dy is masked by y > 0 on the way in (the activation's backward), so the masked gradient is never
// materialised for this product.
constexpr int WG_RC = 256;
// (the partial tiles go to scratch — part[chunk][n][k], partb[chunk][n] — and a second kernel adds the chunks in order:
// a fixed summation order, and no atomics piling up on the 256 x 200 addresses of dW)
__global__ __launch_bounds__(256) void linear_weight_grad_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                                 const float* __restrict__ relu_y,
                                                                 const int32_t* __restrict__ m_dev, int N, int K,
                                                                 float* __restrict__ part, float* __restrict__ partb,
                                                                 int rc) {
  __shared__ float s_dy[32][68];
  __shared__ float s_a[32][68];
  const int M = *m_dev;
  const int r0 = blockIdx.x * rc;
  if (r0 >= M) return;
  const int r1 = min(M, r0 + rc);
  const int n0 = blockIdx.y * 64, k0 = blockIdx.z * 64;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  float colsum = 0.f;  // (k tile 0 only) column sums of dy: thread tid < 64 owns column n0 + tid
  // staging: thread -> (row tid / 8 of the 32-row block, 8 columns); the NEXT block's operands are already in registers
  // while the current block is multiplied
  const int lr = tid >> 3, lc = (tid & 7) * 8;
  const bool vec = ((N | K) & 3) == 0;
  float rv[8], ru[8];
  auto fetch = [&](int rb) {
    const int row = rb + lr;
#pragma unroll
    for (int t = 0; t < 8; ++t) rv[t] = ru[t] = 0.f;
    if (row >= r1) return;
    const float* pd = dy + (int64_t)row * N + n0 + lc;
    const float* py = relu_y ? relu_y + (int64_t)row * N + n0 + lc : nullptr;
    const float* pa = a + (int64_t)row * K + k0 + lc;
    if (vec) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (n0 + lc + 4 * h < N) {
          const float4_t v = *reinterpret_cast<const float4_t*>(pd + 4 * h);
          float4_t m = {1.f, 1.f, 1.f, 1.f};
          if (py) m = *reinterpret_cast<const float4_t*>(py + 4 * h);
#pragma unroll
          for (int t = 0; t < 4; ++t) rv[4 * h + t] = (!py || m[t] > 0.f) ? v[t] : 0.f;
        }
        if (k0 + lc + 4 * h < K) {
          const float4_t u = *reinterpret_cast<const float4_t*>(pa + 4 * h);
#pragma unroll
          for (int t = 0; t < 4; ++t) ru[4 * h + t] = u[t];
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (n0 + lc + t < N) rv[t] = (!py || py[t] > 0.f) ? pd[t] : 0.f;
        if (k0 + lc + t < K) ru[t] = pa[t];
      }
    }
  };
  fetch(r0);
  for (int rb = r0; rb < r1; rb += 32) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      s_dy[lr][lc + t] = rv[t];
      s_a[lr][lc + t] = ru[t];
    }
    __syncthreads();
    if (rb + 32 < r1) fetch(rb + 32);
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const float4_t dv = *reinterpret_cast<const float4_t*>(&s_dy[r][ty * 4]);
      const float4_t av = *reinterpret_cast<const float4_t*>(&s_a[r][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += dv[i] * av[j];
    }
    if (partb && blockIdx.z == 0 && tid < 64)
      for (int r = 0; r < 32; ++r) colsum += s_dy[r][tid];
    __syncthreads();
  }
  float* pt = part + (int64_t)blockIdx.x * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty * 4 + i;
    if (n >= N) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + tx * 4 + j;
      if (k < K) pt[(int64_t)n * K + k] = acc[i][j];
    }
  }
  if (partb && blockIdx.z == 0 && tid < 64 && n0 + tid < N) partb[(int64_t)blockIdx.x * N + n0 + tid] = colsum;
}

// dw[i] += sum over the chunks that hold real rows of part[c][i] (in chunk order), likewise db
__global__ __launch_bounds__(256) void linear_weight_grad_reduce_kernel(const float* __restrict__ part,
                                                                        const float* __restrict__ partb,
                                                                        const int32_t* __restrict__ m_dev, int64_t nk, int N,
                                                                        float* __restrict__ dw, float* __restrict__ db,
                                                                        int rc) {
  const int chunks = (*m_dev + rc - 1) / rc;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nk) {
    float s = 0.f;
    int c = 0;
    for (; c + 7 < chunks; c += 8) {  // eight chunks' loads in flight; the additions keep chunk order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(c + u) * nk + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < chunks; ++c) s += part[(int64_t)c * nk + i];
    dw[i] += s;
  }
  if (db && i < N) {
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += partb[(int64_t)c * N + i];
    db[i] += s;
  }
}

// Split-precision variant: every fp32 operand is the exact sum of three bf16 numbers (x = b1 + b2 + b3: b1 = x with its
// low 16 bits cleared, b2 = (x - b1) truncated the same way, b3 = the rest — 8 + 8 + 8 significand bits, the
// subtractions are exact), so a.w = sum over the pairs (i, j) of a_i.w_j; the six pairs with i + j <= 4 carry every
// term down to 2^-24 of the product (the dropped ones are <= 2^-24 relative: fp32's own rounding class), each as ONE
// v_mfma_f32_32x32x16_bf16 — sixteen times the rate of the fp32 MFMA, so 6/16 of its matrix-pipe time — with fp32
// accumulation inside the instruction.  The split happens on the way from the prefetch registers into LDS (and/sub/
// and/sub + packing per element, VALU work that runs beside the matrix pipe); LDS holds the three bf16 planes of the
// A and W tiles in 64-byte rows (32 bf16, no padding) whose four 16-byte slots are XOR-swizzled with the row number
// (slot' = slot ^ ((row >> 2) & 3), split_lds_off).  On gfx950 a ds_read_b128 is served in four 16-lane groups
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) over 64 banks: lane l reads k = 8*(l/32) .. +8 of row l%32, and the
// 16 rows of a group fall on 16 distinct 16-byte slots of the 256-byte bank row (rows equal mod 4 share a quarter of
// it, their row >> 2 differ mod 4 inside every group); a ds_write_b64 is served in contiguous 16-lane groups over 32
// banks: a row's eight lanes write its 64 bytes, a group = an even row (slots 0-3 of the 128-byte bank row) and the
// odd row after it (slots 4-7) — every bank once.  (Round 2 padded the rows to 80 bytes: conflict-free reads, but the
// two rows of a write group overlapped in banks 0-3, one extra LDS cycle per group — SQ_LDS_BANK_CONFLICT was a third
// of the LDS cycles, exactly the writes' share — and the tile pair took 61 KB; 49 KB now: three workgroups per CU.)
// Workgroup tile 128 x 64*NJ, four waves as 2 x 2, each 64 rows x 32*NJ columns.
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
  b1 = __float_as_uint(x) & 0xFFFF0000u;
  const float r1 = x - __uint_as_float(b1);
  b2 = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(b2);
  b3 = __float_as_uint(r2) & 0xFFFF0000u;
}

// offset (in bf16 elements) of element k (0..31) of tile row `row` in a swizzled plane
__device__ __forceinline__ int split_lds_off(int row, int k) {
  return row * 32 + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7);
}

// Weight gradient on the matrix cores: dW[n][k] = sum_r dy[r][n] * a[r][k] is a product whose INNER dimension is the row
// index, so both operands enter transposed.  A thread stages eight consecutive ROWS of one column (eight dword loads,
// each coalesced across the 64 lanes = 64 consecutive columns), splits them into the three bf16 planes (split3: the
// exact-sum split of the projection kernel) and stores eight row-consecutive bf16 per plane as one 16-byte LDS write
// into a [column][32 rows] tile — the same swizzled tile the projection reads its fragments from, with "column" in the
// role of the tile row.  Four waves as 2 x 2 over a 64 (n) x 64 (k) tile, six products per 16 rows, two accumulators per
// wave (one per half of the 32-row block) so that dependent MFMAs alternate.  dy is masked by relu_y > 0 on the way in;
// column sums of dy (the bias gradient) are taken from the staged values.  Partials per chunk of `rc` rows, reduced in
// chunk order by linear_weight_grad_reduce_kernel: a fixed summation order.
__global__ __launch_bounds__(256) void linear_weight_grad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                                      const float* __restrict__ relu_y,
                                                                      const int32_t* __restrict__ m_dev, int N, int K,
                                                                      float* __restrict__ part, float* __restrict__ partb,
                                                                      int rc) {
  __shared__ __attribute__((aligned(16))) short s_d[3][64 * 32];
  __shared__ __attribute__((aligned(16))) short s_w[3][64 * 32];
  __shared__ float s_strip[4][32 * 36];
  __shared__ float s_cs[4][64];
  const int M = *m_dev;
  const int r0 = blockIdx.x * rc;
  if (r0 >= M) return;
  const int r1 = min(M, r0 + rc);
  const int n0 = blockIdx.y * 64, k0 = blockIdx.z * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane, rb = wv;  // staging: column c of the tile, rows rb*8 .. rb*8+7 of the 32-row block
  const bool dcol = n0 + c < N, acol = k0 + c < K;
  float vd[8], va[8];
  auto fetch = [&](int rbase) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = rbase + rb * 8 + t;
      float d = 0.f, x = 0.f;
      if (row < r1) {
        if (dcol) {
          d = dy[(int64_t)row * N + n0 + c];
          if (relu_y && !(relu_y[(int64_t)row * N + n0 + c] > 0.f)) d = 0.f;
        }
        if (acol) x = a[(int64_t)row * K + k0 + c];
      }
      vd[t] = d;
      va[t] = x;
    }
  };
  auto stage = [&](const float* v, short (*pl)[64 * 32]) {
    uint32_t b1[8], b2[8], b3[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) split3(v[t], b1[t], b2[t], b3[t]);
    const int o = split_lds_off(c, rb * 8);
    *reinterpret_cast<uint4*>(&pl[0][o]) = make_uint4((b1[0] >> 16) | b1[1], (b1[2] >> 16) | b1[3], (b1[4] >> 16) | b1[5], (b1[6] >> 16) | b1[7]);
    *reinterpret_cast<uint4*>(&pl[1][o]) = make_uint4((b2[0] >> 16) | b2[1], (b2[2] >> 16) | b2[3], (b2[4] >> 16) | b2[5], (b2[6] >> 16) | b2[7]);
    *reinterpret_cast<uint4*>(&pl[2][o]) = make_uint4((b3[0] >> 16) | b3[1], (b3[2] >> 16) | b3[3], (b3[4] >> 16) | b3[5], (b3[6] >> 16) | b3[7]);
  };
  const int wn = wv >> 1, wk = wv & 1, r = lane & 31, g = lane >> 5;
  float16_t acc0, acc1;
#pragma unroll
  for (int t = 0; t < 16; ++t) acc0[t] = acc1[t] = 0.f;
  float colsum = 0.f;
  fetch(r0);
  for (int rbase = r0; rbase < r1; rbase += 32) {
    stage(vd, s_d);
    stage(va, s_w);
#pragma unroll
    for (int t = 0; t < 8; ++t) colsum += vd[t];
    __syncthreads();
    if (rbase + 32 < r1) fetch(rbase + 32);  // (in flight during this block's MFMAs)
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PW[6] = {0, 2, 1, 0, 1, 0};  // smallest terms first
    bf16x8_t fa0[3], fw0[3], fa1[3], fw1[3];
#pragma unroll
    for (int p_ = 0; p_ < 3; ++p_) {
      fa0[p_] = *reinterpret_cast<const bf16x8_t*>(&s_d[p_][split_lds_off(wn * 32 + r, 8 * g)]);
      fw0[p_] = *reinterpret_cast<const bf16x8_t*>(&s_w[p_][split_lds_off(wk * 32 + r, 8 * g)]);
      fa1[p_] = *reinterpret_cast<const bf16x8_t*>(&s_d[p_][split_lds_off(wn * 32 + r, 16 + 8 * g)]);
      fw1[p_] = *reinterpret_cast<const bf16x8_t*>(&s_w[p_][split_lds_off(wk * 32 + r, 16 + 8 * g)]);
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[PA[t]], fw0[PW[t]], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[PA[t]], fw1[PW[t]], acc1, 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) acc0[t] += acc1[t];
  store_block_32x32(acc0, s_strip[wv], lane, nullptr, 0, n0 + wn * 32, k0 + wk * 32, N, K,
                    part + (int64_t)blockIdx.x * N * K);
  if (partb && blockIdx.z == 0) {
    s_cs[rb][c] = colsum;
    __syncthreads();
    if (tid < 64 && n0 + tid < N)
      partb[(int64_t)blockIdx.x * N + n0 + tid] = (s_cs[0][tid] + s_cs[1][tid]) + (s_cs[2][tid] + s_cs[3][tid]);
  }
}

// four consecutive k of one row -> four bf16 of each plane, stored as 8 bytes per plane
__device__ __forceinline__ void split_store(const float4_t v, short* p1, short* p2, short* p3) {
  uint32_t a1[4], a2[4], a3[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) split3(v[t], a1[t], a2[t], a3[t]);
  *reinterpret_cast<uint2*>(p1) = make_uint2((a1[0] >> 16) | a1[1], (a1[2] >> 16) | a1[3]);
  *reinterpret_cast<uint2*>(p2) = make_uint2((a2[0] >> 16) | a2[1], (a2[2] >> 16) | a2[3]);
  *reinterpret_cast<uint2*>(p3) = make_uint2((a3[0] >> 16) | a3[1], (a3[2] >> 16) | a3[3]);
}

// (values with at most 16 significant bits — widened halves: the third plane would be zero)
__device__ __forceinline__ void split_store2(const float4_t v, short* p1, short* p2) {
  uint32_t a1[4], a2[4], a3[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) split3(v[t], a1[t], a2[t], a3[t]);
  *reinterpret_cast<uint2*>(p1) = make_uint2((a1[0] >> 16) | a1[1], (a1[2] >> 16) | a1[3]);
  *reinterpret_cast<uint2*>(p2) = make_uint2((a2[0] >> 16) | a2[1], (a2[2] >> 16) | a2[3]);
}

// HS (half split): x = h1 + h2 with h1 = fp16(x), h2 = fp16(x - h1) — 11 + 11 significand bits, the subtraction exact,
// subnormal halves honoured by v_mfma_f32_32x32x16_f16 on gfx950 (scripts/micro/mfma_f16_denorm.hip), so the small
// parts keep an absolute precision of 2^-25.  a.w = h1.h1 + h1.h2 + h2.h1 (the dropped h2.h2 is <= 2^-22 relative): THREE
// MFMAs per accumulator instead of six, two planes per operand in LDS instead of three.  fp16 has a narrow range on
// BOTH sides (overflow past 65504; below 2^-14 the halves are subnormal and h2 keeps only an ABSOLUTE 2^-25), so each
// operand is multiplied by a power of two on its way into the planes — hs_scale = {s_a, s_w, 1 / (s_a s_w)} in device
// memory, chosen so that the operand's largest magnitude lands in [2^14, 2^15) (gigl_hs_scale_update: s_a from the
// table's largest magnitude, s_w from the weights as they are on the device) — and the accumulators are multiplied by
// 1 / (s_a s_w) in the epilogue: all three factors are exact, the element error is max(2^-22 |x|, 2^-39 max|x|) whatever
// the operand's scale.  With AHALF the A operand IS its h1 plane (fp16 rows as stored, s_a = 1): two MFMAs.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void hsplit_store(const float4_t v, short* p1, short* p2, float s) {
  _Float16 h1[4], h2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float x = v[t] * s;  // (s is a power of two: exact)
    h1[t] = (_Float16)x;
    h2[t] = (_Float16)(x - (float)h1[t]);
  }
  uint16_t b1[4], b2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    b1[t] = __builtin_bit_cast(uint16_t, h1[t]);
    b2[t] = __builtin_bit_cast(uint16_t, h2[t]);
  }
  *reinterpret_cast<uint2*>(p1) = make_uint2((uint32_t)b1[0] | ((uint32_t)b1[1] << 16), (uint32_t)b1[2] | ((uint32_t)b1[3] << 16));
  *reinterpret_cast<uint2*>(p2) = make_uint2((uint32_t)b2[0] | ((uint32_t)b2[1] << 16), (uint32_t)b2[2] | ((uint32_t)b2[3] << 16));
}
// SELF (the SAGE layer's [mean | self] operand without the self copy): the A operand has TWO sources — columns k <
// d_mean come from the tiled buffer the gather wrote (a_tiled = its chunks per row tile, ceil(d_mean / 32)), columns
// k >= d_mean are element k - d_mean of row self_ids[row] (or `row`) of `self_src` (fp32 rows self_ld apart: the
// resident feature table through union.nodes for the first layer, the previous layer's output after it).  d_mean % 4
// == 0, so a lane's 4-column segment lies wholly in one source; the K order, and with it every bit of the result, is
// that of the single-source operand.
// AHALF: `a` points at fp16 rows (K % 4 == 0, row-major).  A half has 11 significant bits: it is the exact sum of TWO bf16
// numbers, so the A operand keeps two planes and the product with its (zero) third plane is dropped — five MFMAs per
// accumulator instead of six, and the rows are read as stored (2 bytes per element, no widened copy).  The remaining
// products run in the order of the fp32 path: the same accumulators up to the sign of a zero.
// SH (with SELF): the self source's rows are fp16 (the resident table as stored; self_ld in halves) — an fp16 table's first
// layer then needs no fp32 copy of the self half either (the gather writes the reduced half only)
template <int NJ, bool KVEC = true, bool SELF = false, bool AHALF = false, bool HS = false, bool ONE = false, bool SH = false>  // KVEC false: K % 4 != 0 (row-major operands only) — element loads
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((ONE || (HS && AHALF)) ? 3 : 1))) void linear_split_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const int32_t* __restrict__ m_dev, int K, int N, int act,
                                                           float* __restrict__ y, int a_tiled, int ldy,
                                                           int64_t a_bstride, int64_t w_bstride,
                                                           const float* __restrict__ self_src = nullptr,
                                                           const uint32_t* __restrict__ self_ids = nullptr,
                                                           int d_mean = 0, int self_ld = 0,
                                                           const float* __restrict__ hs_scale = nullptr,
                                                           const gigl_linear_group* __restrict__ groups = nullptr) {
  // (grid.y = batch of independent products sharing K/N: operand b of a / w is a_bstride / w_bstride floats on, its
  // bias and its N output columns follow the previous batch's — or, `groups`: every product names its own operands, row
  // count and output rows: gigl_linear_grouped)
  if (groups) {
    const gigl_linear_group g = groups[blockIdx.y];
    a = g.a;
    w = g.w;
    bias = g.bias;
    m_dev = g.m_dev;
    y = g.y;
  } else {
    a += blockIdx.y * a_bstride;
    w += blockIdx.y * w_bstride;
    if (bias) bias += blockIdx.y * N;
    y += blockIdx.y * N;
  }
  constexpr int BK = 32, LDK = 32;          // bf16 elements per LDS row (swizzled slots, no padding: split_lds_off)
  constexpr int BM = 128, BN = 64 * NJ;
  constexpr int NPA = HS ? (AHALF ? 1 : 2) : 3, NPW = HS ? 2 : 3;  // planes per operand
  constexpr int A_EL = NPA * BM * LDK, W_EL = NPW * BN * LDK, STRIP_EL = 4 * 32 * 36 * 2;  // (shorts; the epilogue strips)
  __shared__ __attribute__((aligned(16))) short s_buf[A_EL + W_EL > STRIP_EL ? A_EL + W_EL : STRIP_EL];
  short(*s_a)[BM * LDK] = reinterpret_cast<short(*)[BM * LDK]>(s_buf);
  short(*s_w)[BN * LDK] = reinterpret_cast<short(*)[BN * LDK]>(s_buf + A_EL);
  const int M = *m_dev;
  const int tiles_n = (N + BN - 1) / BN;
  float hs_a = 1.f, hs_w = 1.f, hs_o = 1.f;  // HS: power-of-two operand scales and the epilogue's inverse (uniform)
  if constexpr (HS) {
    if (hs_scale) {
      hs_a = hs_scale[0];
      hs_w = hs_scale[1];
      hs_o = hs_scale[2];
    }
  }
  // Workgroups are dealt to the 8 XCDs round-robin by id and every XCD has its own L2: the column tiles of one row
  // tile are given ids that differ by 8, so they run on the same XCD back to back and the A tile comes from HBM once
  // (PMC: FETCH_SIZE of the kernel was 2.1x the A matrix with adjacent ids).
  int tm, tn;
  {
    const unsigned bid = blockIdx.x, span = 8u * (unsigned)tiles_n;
    const unsigned full = (gridDim.x / span) * span;  // ids beyond the last whole group keep the plain mapping
    if (bid < full) {
      const unsigned grp = bid / span, in = bid % span;
      tm = (int)(grp * 8u + (in & 7u));
      tn = (int)(in >> 3);
    } else {
      tm = (int)(bid / (unsigned)tiles_n);
      tn = (int)(bid % (unsigned)tiles_n);
    }
  }
  const int m0b = tm * BM, n0b = tn * BN;
  if (m0b >= M) return;  // whole workgroup
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const int wm = wv >> 1, wn = wv & 1;  // wave -> rows [64 wm, +64), columns [32 NJ wn, +32 NJ)
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};

  float16_t acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  // loaders: thread -> rows (tid >> 3) + 32 i, k segment (tid & 7) * 4 .. +4 (128-byte row pieces per 8 lanes)
  const int lr = tid >> 3, lc = tid & 7;
  constexpr bool k_vec = KVEC;
  // operands of TWO chunks ahead stay in flight in registers (a chunk's MFMAs are shorter than a global load under load)
  // ONE: one chunk of operands in flight instead of two — half the prefetch registers, which lets the instantiation fit
  // three waves per SIMD (the third wave hides what the second chunk in flight hid).  The plans' tiled launches use
  // it (several streams' kernels share the CUs: +2 % products, +8 % sharded step); a projection running alone —
  // gigl_linear in a training step — is faster with two chunks ahead at two waves (training step 0.533 vs 0.551 ms)
  constexpr bool PF1 = ONE;
  float4_t ga[PF1 ? 1 : 2][4], gw[PF1 ? 1 : 2][2 * NJ];
  const float* self_row[4] = {nullptr, nullptr, nullptr, nullptr};  // SELF: this thread's four rows of the self source
  if constexpr (SELF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0b + lr + 32 * i;
      if (row < M) {
        const int64_t at = (int64_t)(self_ids ? self_ids[row] : (uint32_t)row) * self_ld;
        self_row[i] = SH ? reinterpret_cast<const float*>(reinterpret_cast<const __half*>(self_src) + at) : self_src + at;
      }
    }
  }
  auto gload = [&](int k0, float4_t (&da)[4], float4_t (&dw)[2 * NJ]) {
    const int kk = k0 + lc * 4;
    // (a_tiled: A is stored [row tile of 128][K chunk of 32][128][32] — this tile's chunk is 16 KB contiguous)
    const float* at = a + ((int64_t)tm * a_tiled + (k0 >> 5)) * 4096 + lc * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0b + lr + 32 * i;
      const float* src = a_tiled ? at + (lr + 32 * i) * 32 : a + (int64_t)row * K + kk;
      if constexpr (SELF) {
        if (kk >= d_mean && row < M) src = self_row[i] + (kk - d_mean);
      }
      if constexpr (SELF && SH) {
        if (kk >= d_mean && row < M && kk < K) {  // four halves of the table row, widened (exact)
          const uint2 h4 = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(self_row[i]) + (kk - d_mean));
          const __half2 lo = *reinterpret_cast<const __half2*>(&h4.x), hi = *reinterpret_cast<const __half2*>(&h4.y);
          da[i][0] = __low2float(lo);
          da[i][1] = __high2float(lo);
          da[i][2] = __low2float(hi);
          da[i][3] = __high2float(hi);
          continue;
        }
      }
      if constexpr (AHALF) {
        da[i] = zero4;
        if (row < M && kk < K) {
          const uint2 h4 = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(a) + (int64_t)row * K + kk);
          if constexpr (HS) {  // the halves ARE the operand plane: their bits travel as they are (two registers)
            da[i][0] = __uint_as_float(h4.x);
            da[i][1] = __uint_as_float(h4.y);
          } else {
            const __half2 lo = *reinterpret_cast<const __half2*>(&h4.x), hi = *reinterpret_cast<const __half2*>(&h4.y);
            da[i][0] = __low2float(lo);
            da[i][1] = __high2float(lo);
            da[i][2] = __low2float(hi);
            da[i][3] = __high2float(hi);
          }
        }
      } else if constexpr (k_vec) {
        da[i] = (row < M && kk < K) ? *reinterpret_cast<const float4_t*>(src) : zero4;
      } else {  // K % 4 != 0: rows are not 16-byte aligned and the last segment is partial — element loads
#pragma unroll
        for (int t = 0; t < 4; ++t) da[i][t] = (row < M && kk + t < K) ? src[t] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 2 * NJ; ++i) {
      const int row = n0b + lr + 32 * i;
      const float* src = w + (int64_t)row * K + kk;
      if constexpr (k_vec) {
        dw[i] = (row < N && kk < K) ? *reinterpret_cast<const float4_t*>(src) : zero4;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) dw[i][t] = (row < N && kk + t < K) ? src[t] : 0.f;
      }
    }
  };
  auto chunk = [&](int k0, float4_t (&ra)[4], float4_t (&rw)[2 * NJ]) {
    __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = split_lds_off(lr + 32 * i, lc * 4);
      if constexpr (HS && AHALF)
        *reinterpret_cast<uint2*>(&s_a[0][o]) = make_uint2(__float_as_uint(ra[i][0]), __float_as_uint(ra[i][1]));
      else if constexpr (HS) hsplit_store(ra[i], &s_a[0][o], &s_a[1][o], hs_a);
      else if constexpr (AHALF) split_store2(ra[i], &s_a[0][o], &s_a[1][o]);
      else split_store(ra[i], &s_a[0][o], &s_a[1][o], &s_a[2][o]);
    }
#pragma unroll
    for (int i = 0; i < 2 * NJ; ++i) {
      const int o = split_lds_off(lr + 32 * i, lc * 4);
      if constexpr (HS) hsplit_store(rw[i], &s_w[0][o], &s_w[1][o], hs_w);
      else split_store(rw[i], &s_w[0][o], &s_w[1][o], &s_w[2][o]);
    }
    __syncthreads();
    if constexpr (PF1) {
      if (k0 + BK < K) gload(k0 + BK, ra, rw);  // lands during this chunk's MFMAs
    } else {
      if (k0 + 2 * BK < K) gload(k0 + 2 * BK, ra, rw);  // lands during this chunk's and the next chunk's MFMAs
    }
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      if (k0 + ks >= K) break;
      bf16x8_t fa[2][3], fw[NJ][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < (HS ? NPA : (AHALF ? 2 : 3)); ++p)
          fa[i][p] = *reinterpret_cast<const bf16x8_t*>(&s_a[p][split_lds_off(wm * 64 + i * 32 + r, ks + 8 * g)]);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int p = 0; p < NPW; ++p)
          fw[j][p] = *reinterpret_cast<const bf16x8_t*>(&s_w[p][split_lds_off(wn * 32 * NJ + j * 32 + r, ks + 8 * g)]);
      if constexpr (HS) {  // three products (two with a one-plane A), smallest terms first, accumulators interleaved
        constexpr int HA[3] = {1, 0, 0}, HW[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if (AHALF && HA[t] == 1) continue;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, fa[i][HA[t]]),
                                                                 __builtin_bit_cast(half8_t, fw[j][HW[t]]), acc[i][j], 0, 0, 0);
        }
        continue;
      }
      // six products per accumulator, smallest terms first; the 2*NJ accumulators are interleaved so that two MFMAs
      // on the same accumulator are never back to back (a dependent MFMA waits for the previous one's 16 passes)
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PW[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (AHALF && PA[t] == 2) continue;  // (the third plane of a half is zero)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[t]], fw[j][PW[t]], acc[i][j], 0, 0, 0);
      }
    }
  };
  gload(0, ga[0], gw[0]);
  if constexpr (PF1) {
    for (int k0 = 0; k0 < K; k0 += BK) chunk(k0, ga[0], gw[0]);
  } else {
    if (BK < K) gload(BK, ga[PF1 ? 0 : 1], gw[PF1 ? 0 : 1]);
    for (int k0 = 0; k0 < K; k0 += 2 * BK) {
      chunk(k0, ga[0], gw[0]);
      if (k0 + BK < K) chunk(k0 + BK, ga[PF1 ? 0 : 1], gw[PF1 ? 0 : 1]);
    }
  }
  __syncthreads();  // every wave is done with the operand planes: s_a becomes the waves' epilogue strips
  float* strip = reinterpret_cast<float*>(s_buf) + wv * (32 * 36);
  if constexpr (HS) {  // undo the operands' power-of-two scales (exact)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] *= hs_o;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      store_block_32x32(acc[i][j], strip, lane, bias, act, m0b + wm * 64 + i * 32, n0b + wn * 32 * NJ + j * 32, M, N, y,
                        ldy);
}

// ------------------------------------------------------------------------------------------
// Two SAGE layers' projections in one kernel (round 5): the first layer's [mean | self] projection with the LAST layer's
// [W_l | W_r] applied to its hidden rows before they leave the workgroup.
//   SAGEConv layer 1 (homogeneous.py:122-126 -> PyG SAGEConv): out_r = W_l mean_j h_j + b + W_r h_r, and W_l mean_j h_j =
//   mean_j (W_l h_j): with p_j = [W_l h_j | W_r h_j] (2 x 47 floats instead of the 256-float h_j) the second layer is ONE
//   reduction over p rows (sage_fused_out_kernel) — the 1-KB hidden rows are never written, the second gather reads
//   192-byte pieces, the second projection disappears.
// Shape: hidden width = 256 (two column tiles of 128), 2 * out <= 96.  A workgroup (row tile tm, column tile tn) holds
// hidden columns [128 tn, +128) of 128 rows, so the second product is K-split over the two column tiles: each writes
// its PARTIAL p rows to plane tn of y2 ([2][rows][96] floats) and the reduction kernel adds the planes (a fixed order:
// plane 0 + plane 1 — no atomics).
// First product: as linear_split_kernel<2, KVEC, SELF, false, HS, ONE> (two fp16 planes per operand, three MFMAs, the
// same loaders, LDS planes and swizzle), except that the four waves split the ROWS (wave = 32 rows x 128 hidden) and the
// MFMA operands are swapped, acc = W-fragment x A-fragment = the TRANSPOSED block: lane r holds row r, its 16 registers
// hidden columns 8 (v / 4) + 4 g + v % 4 — which IS the A-fragment layout of the next product (lane = row, registers = k)
// up to a permutation of k inside each 16-block, applied to W2's planes when they are prepared (fused2_split_kernel):
// the hidden tile goes from accumulators to operand registers without touching LDS.
// Second product: h = relu(acc / (s_a s_w) + b1) is bounded by K max|W1| max|a| + max|b1| =: B (found on the device
// from the scales of the first product), so it takes the half split too: h s_h = h1 + h2 with s_h a power of two that
// brings B under 2^14; element error max(2^-22 |h|, 2^-39 B).  W2's planes (fp16, scaled by s_w2) are staged through LDS
// 64 hidden columns at a time (rows padded to 144 bytes: conflict-free 16-byte reads).
typedef _Float16 half8v_t __attribute__((ext_vector_type(8)));
constexpr int F2_N2 = 96;     // padded width of a p row: [W_l h (out, padded to 48) | W_r h (out, padded to 48)]
constexpr int F2_HID = 256;   // hidden width the kernel is built for
constexpr int F2_W2LD = 72;   // halves per staged W2 row (64 + 8 of padding)

// PD = chunks of operands in flight (registers): a chunk's MFMAs last ~0.3 us, a load from HBM under three streams
// 1-2 us — with ONE chunk in flight (round 5) a workgroup's 7 chunks were 7 round trips one after the other.  WPE = waves
// per SIMD the register budget is cut for (3: 170 registers, 2: 256).
template <int PD, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void linear_fused2_kernel(
    const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
    const int32_t* __restrict__ m_dev, int K, float* __restrict__ y2, int64_t plane_stride, int a_tiled,
    const float* __restrict__ self_src, const uint32_t* __restrict__ self_ids, int d_mean, int self_ld,
    const float* __restrict__ hs_scale, const float* __restrict__ f2_scale, const _Float16* __restrict__ w2h) {
  constexpr int BK = 32, LDK = 32, BM = 128, BN = 128, NJ = 2;
  constexpr int A_EL = 2 * BM * LDK, W_EL = 2 * BN * LDK;  // (shorts) 16 KB + 16 KB
  static_assert(2 * F2_N2 * F2_W2LD <= A_EL + W_EL, "the staged W2 planes reuse the operand planes");
  __shared__ __attribute__((aligned(16))) short s_buf[A_EL + W_EL];
  short(*s_a)[BM * LDK] = reinterpret_cast<short(*)[BM * LDK]>(s_buf);
  short(*s_w)[BN * LDK] = reinterpret_cast<short(*)[BN * LDK]>(s_buf + A_EL);
  const int M = *m_dev;
  constexpr int N = F2_HID, tiles_n = 2;
  const float hs_a = hs_scale[0], hs_w = hs_scale[1], hs_o = hs_scale[2];
  int tm, tn;
  {  // the two column tiles of a row tile run back to back on one XCD (ids 8 apart): the A tile comes from HBM once
    const unsigned bid = blockIdx.x, span = 8u * (unsigned)tiles_n;
    const unsigned full = (gridDim.x / span) * span;
    if (bid < full) {
      const unsigned grp = bid / span, in = bid % span;
      tm = (int)(grp * 8u + (in & 7u));
      tn = (int)(in >> 3);
    } else {
      tm = (int)(bid / (unsigned)tiles_n);
      tn = (int)(bid % (unsigned)tiles_n);
    }
  }
  const int m0b = tm * BM, n0b = tn * BN;
  if (m0b >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  float16_t acc[4];  // [hidden block j of 32][.]: lane r = row 32 wv + r, register v = hidden 8 (v / 4) + 4 g + v % 4
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const int lr = tid >> 3, lc = tid & 7;
  float4_t gas[PD][4], gws[PD][2 * NJ];
  const float* self_row[4] = {nullptr, nullptr, nullptr, nullptr};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0b + lr + 32 * i;
    if (row < M) self_row[i] = self_src + (int64_t)(self_ids ? self_ids[row] : (uint32_t)row) * self_ld;
  }
  auto gload = [&](int k0, float4_t (&ga)[4], float4_t (&gw)[2 * NJ]) {
    const int kk = k0 + lc * 4;
    const float* at = a + ((int64_t)tm * a_tiled + (k0 >> 5)) * 4096 + lc * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0b + lr + 32 * i;
      const float* src = at + (lr + 32 * i) * 32;
      if (kk >= d_mean && row < M) src = self_row[i] + (kk - d_mean);
      ga[i] = (row < M && kk < K) ? *reinterpret_cast<const float4_t*>(src) : zero4;
    }
#pragma unroll
    for (int i = 0; i < 2 * NJ; ++i) {
      const int row = n0b + lr + 32 * i;
      gw[i] = (row < N && kk < K) ? *reinterpret_cast<const float4_t*>(w + (int64_t)row * K + kk) : zero4;
    }
  };
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (s * BK < K) gload(s * BK, gas[s], gws[s]);
  auto chunk = [&](int k0, float4_t (&ga)[4], float4_t (&gw)[2 * NJ]) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = split_lds_off(lr + 32 * i, lc * 4);
      hsplit_store(ga[i], &s_a[0][o], &s_a[1][o], hs_a);
    }
#pragma unroll
    for (int i = 0; i < 2 * NJ; ++i) {
      const int o = split_lds_off(lr + 32 * i, lc * 4);
      hsplit_store(gw[i], &s_w[0][o], &s_w[1][o], hs_w);
    }
    __syncthreads();
    if (k0 + PD * BK < K) gload(k0 + PD * BK, ga, gw);  // (the stage just emptied: PD chunks ahead)
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      if (k0 + ks >= K) break;
      half8v_t fa[2], fw[4][2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
        fa[p] = *reinterpret_cast<const half8v_t*>(&s_a[p][split_lds_off(wv * 32 + r, ks + 8 * g)]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          fw[j][p] = *reinterpret_cast<const half8v_t*>(&s_w[p][split_lds_off(j * 32 + r, ks + 8 * g)]);
      // three products, smallest terms first, the four accumulators interleaved; operands swapped: the block comes out
      // transposed (lane = row)
      constexpr int HA[3] = {1, 0, 0}, HW[3] = {0, 1, 0};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j][HW[t]], fa[HA[t]], acc[j], 0, 0, 0);
    }
  };
  for (int k0 = 0; k0 < K; k0 += PD * BK) {
#pragma unroll
    for (int s = 0; s < PD; ++s)
      if (k0 + s * BK < K) chunk(k0 + s * BK, gas[s], gws[s]);
  }
  // ---- second product: p[row][0:96] (partial over this tile's 128 hidden columns) = relu(h) . W2p[:, 128 tn ..]^T
  const float s_h = f2_scale[0], o2 = f2_scale[2];
  _Float16* s_w2 = reinterpret_cast<_Float16*>(s_buf);  // [2 planes][96][F2_W2LD]
  float16_t acc2[3];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc2[n][v] = 0.f;
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    __syncthreads();  // the operand planes (rd = 0) / the previous 64 columns' W2 fragments are done with
    // stage W2p[plane][n2][n0b + 64 rd .. +64) (halves, k already permuted per 16-block): 2 * 96 rows of 128 bytes
    for (int q = tid; q < 2 * F2_N2 * 8; q += 256) {
      const int pl = q / (F2_N2 * 8), rem = q - pl * (F2_N2 * 8), row = rem >> 3, pc = rem & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(w2h + ((int64_t)pl * F2_N2 + row) * F2_HID + n0b + 64 * rd + 8 * pc);
      *reinterpret_cast<uint4*>(s_w2 + (pl * F2_N2 + row) * F2_W2LD + 8 * pc) = v;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * rd + jj;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        // this lane's 8 hidden values of the 16-block: registers v = 8 hf .. 8 hf + 7 = hidden 32 j + 16 hf + {4 g + t, 8 + 4 g + t}
        half8v_t h1, h2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4_t bq = bias ? *reinterpret_cast<const float4_t*>(bias + n0b + 32 * j + 8 * (2 * hf + q) + 4 * g) : zero4;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float x = acc[j][8 * hf + 4 * q + t] * hs_o + bq[t];
            x = (x > 0.f ? x : 0.f) * s_h;
            const _Float16 a1 = (_Float16)x;
            h1[4 * q + t] = a1;
            h2[4 * q + t] = (_Float16)(x - (float)a1);
          }
        }
        const int kb = 2 * jj + hf;  // 16-block of the staged 64 columns
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          const half8v_t w1 = *reinterpret_cast<const half8v_t*>(s_w2 + (n * 32 + r) * F2_W2LD + 16 * kb + 8 * g);
          const half8v_t w2 = *reinterpret_cast<const half8v_t*>(s_w2 + (F2_N2 + n * 32 + r) * F2_W2LD + 16 * kb + 8 * g);
          acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, w1, acc2[n], 0, 0, 0);
          acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, w2, acc2[n], 0, 0, 0);
          acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, w1, acc2[n], 0, 0, 0);
        }
      }
    }
  }
  // D layout of acc2[n]: lane r = column 32 n + r of p, register v = row 8 (v / 4) + 4 g + v % 4 of the wave's 32 rows:
  // one store instruction covers two rows x 32 columns = two whole 128-byte lines (p rows are 384 bytes)
  float* yp = y2 + (int64_t)tn * plane_stride;
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = m0b + wv * 32 + 8 * (v >> 2) + 4 * g + (v & 3);
      if (row < M) yp[(int64_t)row * F2_N2 + n * 32 + r] = acc2[n][v] * o2;
    }
}

// ---- round 6: the same two products with the A operand straight from global memory
// What the round-5 kernel waited on (SQ counters, profiles/r06h_sq_products.json: matrix pipe 30 %, LDS 21-36 %, vector ALU
// ~40 % busy, 2.7 waves per SIMD resident, each waiting ~80 % of its cycles; deeper register prefetch at two waves per SIMD
// was slower, profiles/r06v_*): the chain per 32-k chunk — operands land, EVERY thread splits 16 A + 16 W floats into LDS,
// barrier, fragments back out of LDS, MFMAs, barrier — with three workgroups per CU to overlap it.  But the four waves split
// the ROWS, so a wave's A rows are read by that wave only: they need no LDS.  Here
//   * lane (r, g) of wave wv loads its own row's 16 floats of the chunk (four 16-byte loads: floats 8 q + 4 g .. + 4, so one
//     instruction reads 32 contiguous bytes per row) and splits them in registers into the two MFMA steps' fragments;
//   * W1's fp16 planes are laid out ONCE per run (fused2_images_kernel) as the chunk's 16-KB LDS image — swizzle included, k
//     inside a chunk permuted to the order the lanes hold it: position 16 s + 8 g + e <- k = 16 s + 8 (e / 4) + 4 g + e % 4,
//     the permutation the accumulator layout imposes on the second product anyway — so a chunk of W is four 16-byte copies
//     per thread, no vector ALU work, into one of TWO image buffers: ONE barrier per chunk;
//   * W2's planes likewise, 32 hidden columns (= one accumulator block) per round, ping-ponging between the two buffers.
// Per chunk and workgroup: 16 KB of LDS writes + 64 KB of reads (was 32 + 80), half the conversions, half the barriers.
constexpr int F2_IMG1 = 2 * 128 * 32;  // halves per W1 chunk image: [plane][128 hidden columns][32 k, swizzled]
constexpr int F2_IMG2 = 2 * F2_N2 * 32;  // halves per W2 round image: [plane][96 p columns][32 hidden, swizzled]
__device__ __forceinline__ int f2_perm(int kpos) {  // k (inside a 32-block) held at fragment position kpos
  return 16 * (kpos >> 4) + 8 * ((kpos >> 2) & 1) + 4 * ((kpos >> 3) & 1) + (kpos & 3);
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// DBG: `ablate` switches parts of the kernel off (wrong rows; timing experiments only — GIGL_F2_ABLATE, bits: 1 no A loads,
// 2 no W image copies, 4 one MFMA of three in the first product, 8 no second product, 16 no output stores, 32 no barriers)
template <int AD, int WPE, bool DBG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void linear_fused2w_kernel(
    const float* __restrict__ a, const float* __restrict__ bias, const int32_t* __restrict__ m_dev, int K,
    float* __restrict__ y2, int64_t plane_stride, int a_tiled, const float* __restrict__ self_src,
    const uint32_t* __restrict__ self_ids, int d_mean, int self_ld, const float* __restrict__ hs_scale,
    const float* __restrict__ f2_scale, const _Float16* __restrict__ w1img, const _Float16* __restrict__ w2img, int ablate) {
  const int ab = DBG ? ablate : 0;
  auto barrier = [&]() {
    if (!(ab & 32)) __syncthreads();
  };
  __shared__ __attribute__((aligned(16))) short s_buf[2][F2_IMG1];
  const int M = *m_dev;
  constexpr int tiles_n = 2;
  const float hs_a = hs_scale[0], hs_o = hs_scale[2];
  int tm, tn;
  {  // (as linear_fused2_kernel: the two column tiles of a row tile back to back on one XCD)
    const unsigned bid = blockIdx.x, span = 8u * (unsigned)tiles_n;
    const unsigned full = (gridDim.x / span) * span;
    if (bid < full) {
      const unsigned grp = bid / span, in = bid % span;
      tm = (int)(grp * 8u + (in & 7u));
      tn = (int)(in >> 3);
    } else {
      tm = (int)(bid / (unsigned)tiles_n);
      tn = (int)(bid % (unsigned)tiles_n);
    }
  }
  const int m0b = tm * 128, n0b = tn * 128;
  if (m0b >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const int nc = (K + 31) >> 5;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  float16_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const int row = m0b + wv * 32 + r;
  const bool valid = row < M;
  const float* self_row = valid ? self_src + (int64_t)(self_ids ? self_ids[row] : (uint32_t)row) * self_ld : self_src;
  const float* at = a + (int64_t)tm * a_tiled * 4096 + (wv * 32 + r) * 32 + 4 * g;
  auto aload = [&](int c, float4_t (&ra)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = c * 32 + 8 * q + 4 * g;
      const float* src = kk >= d_mean ? self_row + (kk - d_mean) : at + (int64_t)c * 4096 + 8 * q;
      ra[q] = (valid && kk < K && !(ab & 1)) ? *reinterpret_cast<const float4_t*>(src) : zero4;
    }
  };
  const u32x4_t* w1src = reinterpret_cast<const u32x4_t*>(w1img) + (int64_t)tn * nc * (F2_IMG1 / 8) + tid;
  u32x4_t wr0, wr1, wr2, wr3;  // (named: as an array the compiler kept two of them in scratch memory)
  auto wload = [&](int c) {
    if (ab & 2) return;
    const u32x4_t* src = w1src + (int64_t)c * (F2_IMG1 / 8);
    wr0 = src[0];
    wr1 = src[256];
    wr2 = src[512];
    wr3 = src[768];
  };
  auto wstore = [&](int b) {
    if (ab & 2) return;
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(s_buf[b]) + tid;
    dst[0] = wr0;
    dst[256] = wr1;
    dst[512] = wr2;
    dst[768] = wr3;
  };
  float4_t ras[AD][4];
  wload(0);
#pragma unroll
  for (int d = 0; d < AD; ++d)
    if (d < nc) aload(d, ras[d]);
  wstore(0);
  if (nc > 1) wload(1);
  const u32x4_t* w2src = reinterpret_cast<const u32x4_t*>(w2img) + (int64_t)tn * 4 * (F2_IMG2 / 8) + tid;
  auto w2load = [&](int rd) {
    const u32x4_t* src = w2src + (int64_t)rd * (F2_IMG2 / 8);
    wr0 = src[0];
    wr1 = src[256];
    wr2 = src[512];
  };
  auto w2store = [&](int b) {
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(s_buf[b]) + tid;
    dst[0] = wr0;
    dst[256] = wr1;
    dst[512] = wr2;
  };
  if (nc == 1) w2load(0);
  barrier();
  auto chunk = [&](int c, float4_t (&ra)[4]) {
    half8v_t fa[2][2];  // [MFMA step][plane]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = ra[2 * s + (e >> 2)][e & 3] * hs_a;
        const _Float16 a1 = (_Float16)x;
        fa[s][0][e] = a1;
        fa[s][1][e] = (_Float16)(x - (float)a1);
      }
    if (c + AD < nc) aload(c + AD, ra);
    const short* sw = s_buf[c & 1];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (c * 32 + 16 * s >= K) break;
      half8v_t fw[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          fw[j][p] = *reinterpret_cast<const half8v_t*>(&sw[p * (F2_IMG1 / 2) + split_lds_off(j * 32 + r, 16 * s + 8 * g)]);
      constexpr int HA[3] = {1, 0, 0}, HW[3] = {0, 1, 0};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if ((ab & 4) && t < 2) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j][HW[t]], fa[s][HA[t]], acc[j], 0, 0, 0);
      }
    }
    if (c + 1 < nc) {
      wstore((c + 1) & 1);
      if (c + 2 < nc) wload(c + 2);
      else w2load(0);  // (the image registers are free from here on: the second product's first round)
    }
    barrier();
  };
  for (int c = 0; c < nc; c += AD) {
#pragma unroll
    for (int d = 0; d < AD; ++d)
      if (c + d < nc) chunk(c + d, ras[d]);
  }
  // ---- second product, one accumulator block (32 hidden columns) per round
  const float s_h = f2_scale[0], o2 = f2_scale[2];
  float16_t acc2[3];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc2[n][v] = 0.f;
  w2store(0);
  w2load(1);
  barrier();
#pragma unroll
  for (int rd = 0; rd < 4; ++rd) {
    if (ab & 8) break;
    const _Float16* sw2 = reinterpret_cast<const _Float16*>(s_buf[rd & 1]);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      half8v_t h1, h2;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4_t bq = bias ? *reinterpret_cast<const float4_t*>(bias + n0b + 32 * rd + 8 * (2 * hf + q) + 4 * g) : zero4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float x = acc[rd][8 * hf + 4 * q + t] * hs_o + bq[t];
          x = (x > 0.f ? x : 0.f) * s_h;
          const _Float16 a1 = (_Float16)x;
          h1[4 * q + t] = a1;
          h2[4 * q + t] = (_Float16)(x - (float)a1);
        }
      }
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const half8v_t w1 = *reinterpret_cast<const half8v_t*>(sw2 + split_lds_off(n * 32 + r, 16 * hf + 8 * g));
        const half8v_t w2 = *reinterpret_cast<const half8v_t*>(sw2 + F2_IMG2 / 2 + split_lds_off(n * 32 + r, 16 * hf + 8 * g));
        acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, w1, acc2[n], 0, 0, 0);
        acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, w2, acc2[n], 0, 0, 0);
        acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, w1, acc2[n], 0, 0, 0);
      }
    }
    if (rd + 1 < 4) {
      w2store((rd + 1) & 1);
      if (rd + 2 < 4) w2load(rd + 2);
      barrier();
    }
  }
  float* yp = y2 + (int64_t)tn * plane_stride;
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int orow = m0b + wv * 32 + 8 * (v >> 2) + 4 * g + (v & 3);
      if (orow < M && !(ab & 16)) yp[(int64_t)orow * F2_N2 + n * 32 + r] = acc2[n][v] * o2;
    }
}

// ---- the whole hidden width in one workgroup (round 6, after the ablation runs of linear_fused2w_kernel,
// profiles/r06x_fused2w_ablations.txt: the parts of the kernel ADD UP — A loads 1.5 us/step, output stores 0.7, second product
// 0.9, W copies 0.5, first product's MFMAs 0.7, conversions + bookkeeping 1.3 — nothing hides behind anything else, and two
// waves per SIMD run as fast as three).  So do less of each: a workgroup takes 128 rows x ALL 256 hidden columns (a wave: 32
// rows x 256, 128 accumulator registers, two waves per SIMD): a row's A chunk is loaded and split ONCE (was once per column
// tile: twice), the second product is not K-split any more — one p row of 384 B leaves instead of two partial ones, and
// sage_fused_out_kernel reads one 192-byte piece per edge instead of two.  LDS: two buffers of two W1 chunk images (64 KB).
// GLDS: the images go global -> LDS without passing registers (global_load_lds_dwordx4: a wave instruction lands 1 KB at a
// wave-uniform LDS address + 16 lane — the images are copied in that order anyway): no staging registers, no ds_write pass.
template <bool DBG, bool GLDS, bool GLDS2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void linear_fused2x_kernel(
    const float* __restrict__ a, const float* __restrict__ bias, const int32_t* __restrict__ m_dev, int K,
    float* __restrict__ y2, int a_tiled, const float* __restrict__ self_src, const uint32_t* __restrict__ self_ids,
    int d_mean, int self_ld, const float* __restrict__ hs_scale, const float* __restrict__ f2_scale,
    const _Float16* __restrict__ w1img, const _Float16* __restrict__ w2img, int ablate,
    const int32_t* __restrict__ n_self_rows) {
  __shared__ __attribute__((aligned(16))) short s_buf[2][2 * F2_IMG1];
  const int ab = DBG ? ablate : 0;
  const int M = *m_dev;
  const float hs_a = hs_scale[0], hs_o = hs_scale[2];
  const int tm = blockIdx.x, m0b = tm * 128;
  if (m0b >= M) return;
  // the W_r half of a p row (columns 48 .. 94) is read for ROOTS only (sage_fused_out_kernel: the root's own term), and the
  // rows are numbered level by level — roots first: a tile past them leaves the last 32 columns (all W_r) alone: a third of
  // the second product's MFMAs and W2 fragment reads, 128 of a row's 384 bytes not written
  const bool wr_block = !n_self_rows || m0b < *n_self_rows;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const int nc = (K + 31) >> 5;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  float16_t acc[8];  // [hidden block of 32][.]
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const int row = m0b + wv * 32 + r;
  const bool valid = row < M;
  const float* self_row = valid ? self_src + (int64_t)(self_ids ? self_ids[row] : (uint32_t)row) * self_ld : self_src;
  const float* at = a + (int64_t)tm * a_tiled * 4096 + (wv * 32 + r) * 32 + 4 * g;
  float4_t ra[4];
  auto aload = [&](int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = c * 32 + 8 * q + 4 * g;
      const float* src = kk >= d_mean ? self_row + (kk - d_mean) : at + (int64_t)c * 4096 + 8 * q;
      ra[q] = (valid && kk < K && !(ab & 1)) ? *reinterpret_cast<const float4_t*>(src) : zero4;
    }
  };
  constexpr int IMGU = F2_IMG1 / 8;  // 16-byte units per W1 chunk image
  const u32x4_t* w1src = reinterpret_cast<const u32x4_t*>(w1img) + tid;
  u32x4_t wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto wload = [&](int c, int b) {  // (GLDS: straight into buffer b)
    if (ab & 2) return;
    const u32x4_t* s0 = w1src + (int64_t)c * IMGU;
    const u32x4_t* s1 = w1src + (int64_t)(nc + c) * IMGU;
    if constexpr (GLDS) {
      short* dst = s_buf[b] + wv * 512;  // (this wave's 1-KB piece of each quarter image)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(s0 + 256 * i), (lptr_t)(dst + 2048 * i), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(s1 + 256 * i), (lptr_t)(dst + F2_IMG1 + 2048 * i), 16, 0, 0);
      }
    } else {
      wr0 = s0[0];
      wr1 = s0[256];
      wr2 = s0[512];
      wr3 = s0[768];
      wr4 = s1[0];
      wr5 = s1[256];
      wr6 = s1[512];
      wr7 = s1[768];
    }
  };
  auto wstore = [&](int b) {
    if (GLDS || (ab & 2)) return;
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(s_buf[b]) + tid;
    dst[0] = wr0;
    dst[256] = wr1;
    dst[512] = wr2;
    dst[768] = wr3;
    dst[IMGU] = wr4;
    dst[IMGU + 256] = wr5;
    dst[IMGU + 512] = wr6;
    dst[IMGU + 768] = wr7;
  };
  const u32x4_t* w2src = reinterpret_cast<const u32x4_t*>(w2img) + tid;
  auto w2load = [&](int rd, int b) {
    const u32x4_t* src = w2src + (int64_t)rd * (F2_IMG2 / 8);
    if constexpr (GLDS2) {
      short* dst = s_buf[b] + wv * 512;
#pragma unroll
      for (int i = 0; i < 3; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + 256 * i), (lptr_t)(dst + 2048 * i), 16, 0, 0);
    } else {
      wr0 = src[0];
      wr1 = src[256];
      wr2 = src[512];
    }
  };
  auto w2store = [&](int b) {
    if (GLDS2) return;
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(s_buf[b]) + tid;
    dst[0] = wr0;
    dst[256] = wr1;
    dst[512] = wr2;
  };
  auto barrier = [&]() {
    if (!(ab & 32)) __syncthreads();
  };
  // stage k (W1 chunks 0 .. nc - 1, then W2's rounds) lives in buffer k & 1; its image is fetched while stage k - 1 is worked on
  wload(0, 0);
  aload(0);
  wstore(0);
  if constexpr (!GLDS) {
    if (nc > 1) wload(1, 1);
    else if constexpr (!GLDS2) w2load(0, 1);
  }
  barrier();
  for (int c = 0; c < nc; ++c) {
    half8v_t fa[2][2];  // [MFMA step][plane]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = ra[2 * s + (e >> 2)][e & 3] * hs_a;
        const _Float16 a1 = (_Float16)x;
        fa[s][0][e] = a1;
        fa[s][1][e] = (_Float16)(x - (float)a1);
      }
    if (c + 1 < nc) aload(c + 1);
    if constexpr (GLDS) {
      if (c + 1 < nc) wload(c + 1, (c + 1) & 1);
    }
    if constexpr (GLDS2) {
      if (c + 1 == nc) w2load(0, nc & 1);  // (that buffer's last reader was chunk nc - 2)
    }
    // the first layer's bias for the second product: fetched during the LAST chunk's MFMAs into the tail of the buffer that
    // chunk does not read (a W2 round image takes 12 of its 32 KB) — read per lane from global memory inside the rounds, each
    // of the sixteen (round, half) blocks opened with a wait for its bias load (an L2 round trip behind nine MFMAs)
    // (LDS-direct: no register lives across the MFMAs for it — four more spilled 37)
    if (c + 1 == nc && wv == 0) {
      short* dst = s_buf[nc & 1] + F2_IMG2;
      if (bias) __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const float4_t*>(bias) + lane), (lptr_t)dst, 16, 0, 0);
      else reinterpret_cast<float4_t*>(dst)[lane] = zero4;
    }
    const short* sw = s_buf[c & 1];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (c * 32 + 16 * s >= K) break;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        half8v_t fw[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            fw[j][p] = *reinterpret_cast<const half8v_t*>(
                &sw[tn * F2_IMG1 + p * (F2_IMG1 / 2) + split_lds_off(j * 32 + r, 16 * s + 8 * g)]);
        constexpr int HA[3] = {1, 0, 0}, HW[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if ((ab & 4) && t < 2) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[4 * tn + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j][HW[t]], fa[s][HA[t]], acc[4 * tn + j], 0, 0, 0);
        }
      }
    }
    if constexpr (!GLDS) {
      if (c + 1 < nc) {
        wstore((c + 1) & 1);
        if (c + 2 < nc) wload(c + 2, 0);
        else if constexpr (!GLDS2) w2load(0, 0);  // (the image registers are free from here on: the second product's first round)
      }
    }
    barrier();
  }
  const float4_t* s_bias = reinterpret_cast<const float4_t*>(s_buf[nc & 1] + F2_IMG2);  // [64]: hidden columns 4 i .. 4 i + 3
  // ---- second product, one accumulator block (32 hidden columns) per round, the rounds' W2 images ping-ponging
  const float s_h = f2_scale[0], o2 = f2_scale[2];
  float16_t acc2[3];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc2[n][v] = 0.f;
  if constexpr (!GLDS2) {
    w2store(nc & 1);
    w2load(1, 0);
    barrier();
  }
#pragma unroll
  for (int rd = 0; rd < 8; ++rd) {
    if (ab & 8) break;
    const _Float16* sw2 = reinterpret_cast<const _Float16*>(s_buf[(nc + rd) & 1]);
    if constexpr (GLDS2) {
      if (rd + 1 < 8) w2load(rd + 1, (nc + rd + 1) & 1);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      half8v_t h1, h2;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4_t bq = s_bias[8 * rd + 2 * (2 * hf + q) + g];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float x = acc[rd][8 * hf + 4 * q + t] * hs_o + bq[t];
          x = (x > 0.f ? x : 0.f) * s_h;
          const _Float16 a1 = (_Float16)x;
          h1[4 * q + t] = a1;
          h2[4 * q + t] = (_Float16)(x - (float)a1);
        }
      }
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        if (n == 2 && !wr_block) continue;
        const half8v_t w1 = *reinterpret_cast<const half8v_t*>(sw2 + split_lds_off(n * 32 + r, 16 * hf + 8 * g));
        const half8v_t w2 = *reinterpret_cast<const half8v_t*>(sw2 + F2_IMG2 / 2 + split_lds_off(n * 32 + r, 16 * hf + 8 * g));
        acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, w1, acc2[n], 0, 0, 0);
        acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, w2, acc2[n], 0, 0, 0);
        acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, w1, acc2[n], 0, 0, 0);
      }
    }
    if (rd + 1 < 8) {
      if constexpr (!GLDS2) {
        w2store((nc + rd + 1) & 1);
        if (rd + 2 < 8) w2load(rd + 2, 0);
      }
      barrier();
    }
  }
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    if (n == 2 && !wr_block) continue;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int orow = m0b + wv * 32 + 8 * (v >> 2) + 4 * g + (v & 3);
      if (orow < M && !(ab & 16)) y2[(int64_t)orow * F2_N2 + n * 32 + r] = acc2[n][v] * o2;
    }
  }
}

// the LDS images linear_fused2w_kernel copies: per run, after the scales (hs[1] = s_w of the first product, f2[1] = s_w2).
//   w1img[tn][chunk][plane][split_lds_off(c, kpos)] = plane of s_w W1[128 tn + c][32 chunk + f2_perm(kpos)]  (0 beyond K)
//   w2img[tn][round][plane][split_lds_off(n, kpos)] = plane of s_w2 W2p[n][128 tn + 32 round + f2_perm(kpos)]
// (W2p = [W_l ; W_r] of the last layer padded to 96 rows, as fused2_split_kernel).  One thread per element.
__global__ __launch_bounds__(256) void fused2_images_kernel(const float* __restrict__ hs, const float* __restrict__ f2,
                                                            const float* __restrict__ w1, int K, const float* __restrict__ w2,
                                                            int n_out, _Float16* __restrict__ w1img,
                                                            _Float16* __restrict__ w2img) {
  const int nc = (K + 31) >> 5;
  const int n1 = 2 * nc * 128 * 32, n2 = 2 * 4 * F2_N2 * 32;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n1) {
    const int kpos = i & 31, c = (i >> 5) & 127, blk = i >> 12;  // blk = tn * nc + chunk
    const int tn = blk / nc, ch = blk - tn * nc;
    const int k = 32 * ch + f2_perm(kpos);
    const float x = k < K ? w1[(int64_t)(128 * tn + c) * K + k] * hs[1] : 0.f;
    const _Float16 a1 = (_Float16)x;
    _Float16* img = w1img + (int64_t)blk * F2_IMG1 + split_lds_off(c, kpos);
    img[0] = a1;
    img[F2_IMG1 / 2] = (_Float16)(x - (float)a1);
  } else if (i < n1 + n2) {
    const int e = i - n1;
    const int kpos = e & 31, rest = e >> 5, n = rest % F2_N2, blk = rest / F2_N2;  // blk = tn * 4 + round
    const int k = 32 * blk + f2_perm(kpos);  // (128 tn + 32 round + ...)
    float x = 0.f;
    if (n < n_out) x = w2[(int64_t)n * 2 * F2_HID + k];
    else if (n >= F2_N2 / 2 && n - F2_N2 / 2 < n_out) x = w2[(int64_t)(n - F2_N2 / 2) * 2 * F2_HID + F2_HID + k];
    x *= f2[1];
    const _Float16 a1 = (_Float16)x;
    _Float16* img = w2img + (int64_t)blk * F2_IMG2 + split_lds_off(n, kpos);
    img[0] = a1;
    img[F2_IMG2 / 2] = (_Float16)(x - (float)a1);
  }
}

// f2[0..2] = {s_h, s_w2, 1 / (s_h s_w2)}, per run (the weights are read as they are now): max|b1| and max|W2| over a few
// small workgroups folded with atomicMax (non-negative floats order like their bits), the LAST one to finish — ticket —
// turns them into the scales and clears maxima and ticket for the next run (f2[4], f2[5] = the maxima, f2[6] = ticket:
// zero between runs).  hs = the first product's scales {s_a, s_w, ...} (gigl_hs_scale_update ran before on the same
// stream): max|a| < 2^15 / s_a, max|W1| < 2^15 / s_w.  (As ONE workgroup doing this and the split below the kernel took
// 70 us per call on the stream's critical path next to the other streams' kernels: round 5.)
__global__ __launch_bounds__(256) void fused2_scale_kernel(const float* __restrict__ hs, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, int n_w2, int K1,
                                                           float* __restrict__ f2) {
  __shared__ float s_m[2][4];
  const int tid = threadIdx.x, gt = blockIdx.x * 256 + tid, stride = gridDim.x * 256;
  float mb = 0.f, mw = 0.f;
  if (b1)
    for (int i = gt; i < F2_HID; i += stride) {
      const float v = fabsf(b1[i]);
      mb = fmaxf(mb, v == v ? v : 0.f);
    }
  for (int i = gt; i < n_w2; i += stride) {
    const float v = fabsf(w2[i]);
    mw = fmaxf(mw, v == v ? v : 0.f);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mb = fmaxf(mb, __shfl_xor(mb, o, 64));
    mw = fmaxf(mw, __shfl_xor(mw, o, 64));
  }
  if ((tid & 63) == 0) {
    s_m[0][tid >> 6] = mb;
    s_m[1][tid >> 6] = mw;
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 4; ++i) {
      mb = fmaxf(mb, s_m[0][i]);
      mw = fmaxf(mw, s_m[1][i]);
    }
    uint32_t* acc = reinterpret_cast<uint32_t*>(f2) + 4;
    atomicMax(acc, __float_as_uint(mb));
    atomicMax(acc + 1, __float_as_uint(mw));
    __threadfence();
    if (atomicAdd(acc + 2, 1u) == gridDim.x - 1) {  // the last workgroup: both maxima are in
      const float b = __uint_as_float(atomicExch(acc, 0u)), w = __uint_as_float(atomicExch(acc + 1, 0u));
      acc[2] = 0u;
      // |h| <= K max|W1| max|a| + max|b1| < K 2^30 / (s_a s_w) + max|b1|
      const float inf = __uint_as_float(0x7F800000u);
      float bound = (float)K1 * 1073741824.f * (1.f / hs[0]) * (1.f / hs[1]) + b;
      if (!(bound > 0.f) || !(bound < inf)) bound = 1.f;
      const float s_h = hs_pow2_scale(bound) * 0.5f;  // bound * s_h in [2^13, 2^14)
      const float s_w2 = (w > 0.f && w < inf) ? hs_pow2_scale(w) : 1.f;
      f2[0] = s_h;
      f2[1] = s_w2;
      f2[2] = (1.f / s_h) * (1.f / s_w2);
    }
  }
}

// the fp16 planes of W2p = [W_l ; W_r] of the last layer ([96][256]: rows 0..out-1 = W_l, 48..48+out-1 = W_r, the rest
// zero), scaled by s_w2 = f2[1], each 16-block of k stored in the order the fused kernel's accumulator registers hold it:
// position 8 g + e <- element 8 (e / 4) + 4 g + e % 4.  One thread per element.
__global__ __launch_bounds__(256) void fused2_split_kernel(const float* __restrict__ f2, const float* __restrict__ w2,
                                                           int n_out, _Float16* __restrict__ w2h) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F2_N2 * F2_HID) return;
  const float s_w2 = f2[1];
  const int c = i / F2_HID, pos = i - c * F2_HID;
  const int blk = pos >> 4, in = pos & 15, gg = in >> 3, e = in & 7;
  const int k = blk * 16 + 8 * (e >> 2) + 4 * gg + (e & 3);  // the element that sits at position `pos`
  float x = 0.f;
  if (c < n_out) x = w2[(int64_t)c * 2 * F2_HID + k];
  else if (c >= F2_N2 / 2 && c - F2_N2 / 2 < n_out) x = w2[(int64_t)(c - F2_N2 / 2) * 2 * F2_HID + F2_HID + k];
  x *= s_w2;
  const _Float16 a1 = (_Float16)x;
  w2h[i] = a1;
  w2h[F2_N2 * F2_HID + i] = (_Float16)(x - (float)a1);
}

// Half-split scales of a layer >= 1 (round 5): its operand [reduce(h) | h] is made of the previous layer's outputs h =
// act(a . W^T + b), bounded by K max|W| max|a| + max|b| — and max|a| < 2^15 / s_a, max|W| < 2^15 / s_w are what the
// previous layer's own scales hs_prev = {s_a, s_w, ..} say.  hs_out = {s_h, s_w', 1 / (s_h s_w')}: s_h brings that bound
// under 2^14 (a mean / max of rows stays inside it; a SUM over up to `fan` rows: the bound times fan), s_w' the largest
// |w'| of THIS layer's weights into [2^14, 2^15).  Per run (the weights are read as they are now); hs_out: 8 floats,
// zero-initialised once by the caller.
__global__ __launch_bounds__(256) void hs_chain_kernel(const float* __restrict__ hs_prev, const float* __restrict__ b_prev,
                                                       int n_b, int k_prev, float fan, const float* __restrict__ w,
                                                       int64_t n_w, float* __restrict__ hs_out) {
  // (a few small workgroups + a ticket, as hs_scale_kernel: hs_out[4], [5] = running maxima, [6] = ticket, zero between runs)
  __shared__ float s_m[2][4];
  const int tid = threadIdx.x;
  const int64_t gt = (int64_t)blockIdx.x * 256 + tid, stride = (int64_t)gridDim.x * 256;
  float mb = 0.f, mw = 0.f;
  if (b_prev)
    for (int64_t i = gt; i < n_b; i += stride) {
      const float v = fabsf(b_prev[i]);
      mb = fmaxf(mb, v == v ? v : 0.f);
    }
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(w) & 15) == 0) ? n_w / 4 : 0;
  for (int64_t i = gt; i < n4; i += stride) {
    const float4 v = w4[i];
    const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    mw = fmaxf(mw, a == a ? a : 0.f);
  }
  for (int64_t i = n4 * 4 + gt; i < n_w; i += stride) {
    const float v = fabsf(w[i]);
    mw = fmaxf(mw, v == v ? v : 0.f);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mb = fmaxf(mb, __shfl_xor(mb, o, 64));
    mw = fmaxf(mw, __shfl_xor(mw, o, 64));
  }
  if ((tid & 63) == 0) {
    s_m[0][tid >> 6] = mb;
    s_m[1][tid >> 6] = mw;
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 4; ++i) {
      mb = fmaxf(mb, s_m[0][i]);
      mw = fmaxf(mw, s_m[1][i]);
    }
    uint32_t* acc = reinterpret_cast<uint32_t*>(hs_out) + 4;
    atomicMax(acc, __float_as_uint(mb));
    atomicMax(acc + 1, __float_as_uint(mw));
    __threadfence();
    if (atomicAdd(acc + 2, 1u) == gridDim.x - 1) {
      const float b = __uint_as_float(atomicExch(acc, 0u)), wm = __uint_as_float(atomicExch(acc + 1, 0u));
      acc[2] = 0u;
      const float inf = __uint_as_float(0x7F800000u);
      float bound = ((float)k_prev * 1073741824.f * (1.f / hs_prev[0]) * (1.f / hs_prev[1]) + b) * fan;
      if (!(bound > 0.f) || !(bound < inf)) bound = 1.f;
      const float s_h = hs_pow2_scale(bound) * 0.5f;
      const float s_w = (wm > 0.f && wm < inf) ? hs_pow2_scale(wm) : 1.f;
      hs_out[0] = s_h;
      hs_out[1] = s_w;
      hs_out[2] = (1.f / s_h) * (1.f / s_w);
    }
  }
}

// The last SAGE layer over the p rows of linear_fused2_kernel: for root slot s (local id i = root_local[s]):
//   out[s][c] = act( reduce_{e in row i} (p0 + p1)[col[e]][c] + (p0 + p1)[i][48 + c] + b2[c] ),  c < n_out
// (reduce = mean / sum; p0 / p1 = the two K-split partial planes, added plane 0 first).  One wave per root slot: four
// groups of 16 lanes take every fourth edge, 12 lanes of a group a float4 of the 48-float half row each.  Writes the
// roots' rows straight into the caller's buffer (a failed batch set: NaN rows, as gigl_take_rows).
template <int OP, int PLANES>
__global__ __launch_bounds__(256) void sage_fused_out_kernel(const float* __restrict__ p, int64_t plane_stride,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ rowend,
                                                             const int32_t* __restrict__ col,
                                                             const int32_t* __restrict__ root_local, int b, int n_out,
                                                             const float* __restrict__ bias, int act,
                                                             const int32_t* __restrict__ meta, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15;
  const int s = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (s >= b) return;
  const bool failed = meta[GIGL_META_OVERFLOW] != 0;
  const int i = failed ? -1 : root_local[s];
  float* o = out + (int64_t)s * n_out;
  if (i < 0) {
    const float v = failed ? __builtin_nanf("") : 0.f;
    for (int c = lane; c < n_out; c += 64) o[c] = v;
    return;
  }
  const float* p0 = p;
  const float* p1 = p + plane_stride;
  const int e0 = rowptr[i], m = rowend[i] - e0;
  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  if (sl < 12) {
    for (int e = sub; e < m; e += 8) {  // two edges of this group in flight
      const int ja = col[e0 + e], eb = e + 4;
      const int jb = eb < m ? col[e0 + eb] : ja;
      const float4_t a0 = *reinterpret_cast<const float4_t*>(p0 + (int64_t)ja * F2_N2 + 4 * sl);
      float4_t a1 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (PLANES == 2) a1 = *reinterpret_cast<const float4_t*>(p1 + (int64_t)ja * F2_N2 + 4 * sl);
      float4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (eb < m) {
        b0 = *reinterpret_cast<const float4_t*>(p0 + (int64_t)jb * F2_N2 + 4 * sl);
        if constexpr (PLANES == 2) b1 = *reinterpret_cast<const float4_t*>(p1 + (int64_t)jb * F2_N2 + 4 * sl);
      }
      acc += (a0 + a1) + (b0 + b1);
    }
  }
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    const float4_t o4 = {__shfl_xor(acc.x, off, 64), __shfl_xor(acc.y, off, 64), __shfl_xor(acc.z, off, 64),
                         __shfl_xor(acc.w, off, 64)};
    acc += o4;
  }
  if (sub == 0 && sl < 12) {
    const float dv = (OP == GIGL_AGGR_MEAN && m > 0) ? (float)m : 1.f;
    const float4_t s0 = *reinterpret_cast<const float4_t*>(p0 + (int64_t)i * F2_N2 + F2_N2 / 2 + 4 * sl);
    float4_t s1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PLANES == 2) s1 = *reinterpret_cast<const float4_t*>(p1 + (int64_t)i * F2_N2 + F2_N2 / 2 + 4 * sl);
    float4_t rr = acc / dv + (s0 + s1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = 4 * sl + t;
      if (c < n_out) {
        float v = rr[t] + (bias ? bias[c] : 0.f);
        if (act) v = fmaxf(v, 0.f);
        o[c] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// GCNConv / GATConv aggregation (PyG 2.5.3 semantics as used by homogeneous.py:300-343,488-546)
// ------------------------------------------------------------------------------------------

// hydration only: out[i][0:d] = (float) src[ids[i]][0:d] for i < *n_dev   (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, int d,
                                                          const uint32_t* __restrict__ ids,
                                                          const int32_t* __restrict__ n_dev, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n = *n_dev;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n; i += waves_total) {
    const T* p = src + (int64_t)ids[i] * d;
    for (int el = lane; el < d; el += 64) out[(int64_t)i * d + el] = (float)p[el];
  }
}

// dinv[i] = 1/sqrt(1 + #in-edges of i that are not self loops)   (add_remaining_self_loops + sym. norm)
__global__ __launch_bounds__(256) void gcn_dinv_kernel(const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ rowend,
                                                       const int32_t* __restrict__ col,
                                                       const int32_t* __restrict__ n_nodes_dev, float* __restrict__ dinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_nodes_dev) return;
  const int e0 = rowptr[i], e1 = rowend[i];
  int deg = 1;
  for (int e = e0; e < e1; ++e) deg += col[e] != i;
  dinv[i] = 1.0f / sqrtf((float)deg);
}

// out[i] = act( dinv_i * ( dinv_i * h[idx(i)] + sum_{j in row i, j != i} dinv_j * h[idx(j)] ) + bias )
// one wave per destination row; h rows are fp32 [*, d] gathered through gather_ids (or identity)
template <typename T>
__global__ __launch_bounds__(256) void gcn_gather_kernel(const T* __restrict__ h, int d,
                                                         const uint32_t* __restrict__ gather_ids,
                                                         const float* __restrict__ dinv,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ rowend,
                                                         const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ n_rows_dev,
                                                         const float* __restrict__ bias, int act,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_rows = *n_rows_dev;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    const float di = dinv[i];
    const int64_t self = gather_ids ? (int64_t)gather_ids[i] : i;
    for (int el = lane; el < d; el += 64) {
      float acc = di * (float)h[self * d + el];
      for (int e = 0; e < m; ++e) {
        const int j = col[e0 + e];
        if (j == i) continue;
        const int64_t gj = gather_ids ? (int64_t)gather_ids[j] : j;
        acc += dinv[j] * (float)h[gj * d + el];
      }
      float v = di * acc + (bias ? bias[el] : 0.f);
      if (act == 1) v = v > 0.f ? v : 0.f;
      out[(int64_t)i * d + el] = v;
    }
  }
}

// alpha[i][hd] = sum_c h[i][hd*C + c] * att[hd*C + c]      (one thread per (node, head))
__global__ void gat_alpha_kernel(const float* __restrict__ h, const float* __restrict__ att_src,
                                 const float* __restrict__ att_dst, const int32_t* __restrict__ n_dev, int heads,
                                 int C, float* __restrict__ a_src, float* __restrict__ a_dst) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = *n_dev;
  if (t >= n * heads) return;
  const int64_t i = t / heads;
  const int hd = (int)(t % heads);
  const float* row = h + i * heads * C + hd * C;
  float s = 0.f, dd = 0.f;
  for (int c = 0; c < C; ++c) {
    s += row[c] * att_src[hd * C + c];
    dd += row[c] * att_dst[hd * C + c];
  }
  a_src[t] = s;
  a_dst[t] = dd;
}

// GAT attention + weighted sum for one destination row per wave, all heads:
//   e_ij = leaky_relu(a_src[j] + a_dst[i]) over the in-edges of i (self loops removed) plus ONE self loop,
//   alpha = softmax_j(e_ij), out[i][hd*C + c] = sum_j alpha_ij h[j][hd*C + c]  (+ bias, optional relu);
//   concat == 0: mean over heads -> out[i][c].
__global__ __launch_bounds__(256) void gat_gather_kernel(const float* __restrict__ h, const float* __restrict__ a_src,
                                                         const float* __restrict__ a_dst,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ rowend,
                                                         const int32_t* __restrict__ col,
                                                         const int32_t* __restrict__ n_rows_dev, int heads, int C,
                                                         float slope, int concat, const float* __restrict__ bias,
                                                         int act, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_rows = *n_rows_dev;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int HC = heads * C;
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    for (int hd = 0; hd < heads; ++hd) {
      const float ad = a_dst[(int64_t)i * heads + hd];
      // pass 1: max logit (self loop included)
      float self_e = a_src[(int64_t)i * heads + hd] + ad;
      self_e = self_e > 0.f ? self_e : slope * self_e;
      float mx = self_e;
      for (int e = lane; e < m; e += 64) {
        const int j = col[e0 + e];
        if (j == i) continue;
        float x = a_src[(int64_t)j * heads + hd] + ad;
        x = x > 0.f ? x : slope * x;
        mx = fmaxf(mx, x);
      }
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      // pass 2: denominator
      float den = lane == 0 ? expf(self_e - mx) : 0.f;
      for (int e = lane; e < m; e += 64) {
        const int j = col[e0 + e];
        if (j == i) continue;
        float x = a_src[(int64_t)j * heads + hd] + ad;
        x = x > 0.f ? x : slope * x;
        den += expf(x - mx);
      }
      for (int off = 32; off > 0; off >>= 1) den += __shfl_xor(den, off, 64);
      const float inv = 1.0f / (den + 1e-16f);
      // pass 3: weighted sum; lanes over channels
      for (int c = lane; c < C; c += 64) {
        float acc = expf(self_e - mx) * inv * h[(int64_t)i * HC + hd * C + c];
        for (int e = 0; e < m; ++e) {
          const int j = col[e0 + e];
          if (j == i) continue;
          float x = a_src[(int64_t)j * heads + hd] + ad;
          x = x > 0.f ? x : slope * x;
          acc += expf(x - mx) * inv * h[(int64_t)j * HC + hd * C + c];
        }
        if (concat) {
          float v = acc + (bias ? bias[hd * C + c] : 0.f);
          if (act == 1) v = v > 0.f ? v : 0.f;
          out[(int64_t)i * HC + hd * C + c] = v;
        } else {
          // mean over heads: accumulate in place (heads are processed sequentially by this wave)
          float prev = hd == 0 ? 0.f : out[(int64_t)i * C + c];
          float v = prev + acc / (float)heads;
          if (hd == heads - 1) {
            v += bias ? bias[c] : 0.f;
            if (act == 1) v = v > 0.f ? v : 0.f;
          }
          out[(int64_t)i * C + c] = v;
        }
      }
    }
  }
}

// ---- GATConv with edge features (edge_dim) and EdgeAttrGATConv's edge messages
// a_edge[p][hd] = <edge_attr[p], v_att[hd]>   with v_att[hd][k] = sum_c W_e[hd*C+c][k] * att_edge[hd][c] folded by
// the caller: (W_e e) . att_edge == e . (W_e^T att_edge)     (one thread per (col position, head))
__global__ void gat_edge_alpha_kernel(const float* __restrict__ edge_attr, int De, const float* __restrict__ v_att,
                                      int heads, int64_t cap_edges, float* __restrict__ a_edge) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cap_edges * heads) return;
  const int64_t p = t / heads;
  const int hd = (int)(t % heads);
  const float* e = edge_attr + p * De;
  const float* v = v_att + (int64_t)hd * De;
  float s = 0.f;
  for (int k = 0; k < De; ++k) s += e[k] * v[k];
  a_edge[t] = s;
}

constexpr int GAT_MAX_EDGE_DIM = 256;

// As gat_gather_kernel, with e_ij += a_edge[p] for the edge stored at col position p.  The self loop carries the
// MEAN of the row's (non-self) edge attributes (PyG add_self_loops(fill_value="mean")), hence — by linearity — the
// mean of their a_edge, 0 for a row without in-edges.  w_msg != NULL (EdgeAttrGATConv, edge_attr_gat_conv.py:131-144):
// the message is h_j + W_msg e_ij, so out_i += W_msg (sum_j alpha_ij e_ij) with the self loop's mean attribute.
__global__ __launch_bounds__(256) void gat_edge_gather_kernel(
    const float* __restrict__ h, const float* __restrict__ a_src, const float* __restrict__ a_dst,
    const float* __restrict__ a_edge, const float* __restrict__ edge_attr, int De, const float* __restrict__ w_msg,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ col,
    const int32_t* __restrict__ n_rows_dev, int heads, int C, float slope, int concat,
    const float* __restrict__ bias, int act, float* __restrict__ out) {
  __shared__ float s_z[4][GAT_MAX_EDGE_DIM];
  const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_rows = *n_rows_dev;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int HC = heads * C;
  float* z = s_z[wl];
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    int cnt = 0;
    for (int e = lane; e < m; e += 64) cnt += col[e0 + e] != i;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    const float inv_cnt = cnt > 0 ? 1.0f / (float)cnt : 0.f;
    for (int hd = 0; hd < heads; ++hd) {
      const float ad = a_dst[(int64_t)i * heads + hd];
      float ae_self = 0.f;
      for (int e = lane; e < m; e += 64)
        if (col[e0 + e] != i) ae_self += a_edge[(int64_t)(e0 + e) * heads + hd];
      for (int off = 32; off > 0; off >>= 1) ae_self += __shfl_xor(ae_self, off, 64);
      ae_self *= inv_cnt;
      float self_e = a_src[(int64_t)i * heads + hd] + ad + ae_self;
      self_e = self_e > 0.f ? self_e : slope * self_e;
      float mx = self_e;
      for (int e = lane; e < m; e += 64) {
        const int j = col[e0 + e];
        if (j == i) continue;
        float x = a_src[(int64_t)j * heads + hd] + ad + a_edge[(int64_t)(e0 + e) * heads + hd];
        x = x > 0.f ? x : slope * x;
        mx = fmaxf(mx, x);
      }
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float den = lane == 0 ? expf(self_e - mx) : 0.f;
      for (int e = lane; e < m; e += 64) {
        const int j = col[e0 + e];
        if (j == i) continue;
        float x = a_src[(int64_t)j * heads + hd] + ad + a_edge[(int64_t)(e0 + e) * heads + hd];
        x = x > 0.f ? x : slope * x;
        den += expf(x - mx);
      }
      for (int off = 32; off > 0; off >>= 1) den += __shfl_xor(den, off, 64);
      const float inv = 1.0f / (den + 1e-16f);
      const float al_self = expf(self_e - mx) * inv;
      if (w_msg) {
        // z[k] = sum_j alpha_ij e_ij[k] + alpha_self * mean_j e_ij[k]   (lanes over k)
        for (int k = lane; k < De; k += 64) {
          float zz = 0.f, mean = 0.f;
          for (int e = 0; e < m; ++e) {
            const int j = col[e0 + e];
            if (j == i) continue;
            float x = a_src[(int64_t)j * heads + hd] + ad + a_edge[(int64_t)(e0 + e) * heads + hd];
            x = x > 0.f ? x : slope * x;
            const float ev = edge_attr[(int64_t)(e0 + e) * De + k];
            zz += expf(x - mx) * inv * ev;
            mean += ev;
          }
          z[k] = zz + al_self * mean * inv_cnt;
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (int c = lane; c < C; c += 64) {
        float acc = al_self * h[(int64_t)i * HC + hd * C + c];
        for (int e = 0; e < m; ++e) {
          const int j = col[e0 + e];
          if (j == i) continue;
          float x = a_src[(int64_t)j * heads + hd] + ad + a_edge[(int64_t)(e0 + e) * heads + hd];
          x = x > 0.f ? x : slope * x;
          acc += expf(x - mx) * inv * h[(int64_t)j * HC + hd * C + c];
        }
        if (w_msg) {
          const float* wr = w_msg + (int64_t)(hd * C + c) * De;
          float t = 0.f;
          for (int k = 0; k < De; ++k) t += wr[k] * z[k];
          acc += t;
        }
        if (concat) {
          float v = acc + (bias ? bias[hd * C + c] : 0.f);
          if (act == 1) v = v > 0.f ? v : 0.f;
          out[(int64_t)i * HC + hd * C + c] = v;
        } else {
          float prev = hd == 0 ? 0.f : out[(int64_t)i * C + c];
          float v = prev + acc / (float)heads;
          if (hd == heads - 1) {
            v += bias ? bias[c] : 0.f;
            if (act == 1) v = v > 0.f ? v : 0.f;
          }
          out[(int64_t)i * C + c] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- first GAT layer from the INPUT side (inference).  With x the stored feature rows (d wide, d > H*C is where it
// pays) and W_h the C x d block of head h:  <W_h x_j, att_h> = <x_j, W_h^T att_h>,  so the attention logits need two
// d-vectors per head and never the projected rows; and  sum_j alpha_ij W_h x_j = W_h (sum_j alpha_ij x_j),  so the
// projection runs on the n_dst aggregated rows instead of on all n_src source rows (the union of a [25,10] batch has
// ~4.5x more sources than destinations) and the feature rows are read straight from the resident table in their
// storage type — no fp32 copy of the union's features is made.
//   gat_fold_kernel    u[t][k] = sum_c att_t[h*C+c] * W[h*C+c][k]      (t = h: source side, t = H+h: destination side)
//   gat_score_kernel   s[j][t] = <x[ids[j]], u[t]>                      one wave per source row
//   gat_input_gather   z_h[i] = softmax-weighted mean of x over row i (+ the self loop), per head, written in the
//                      projection's tiled operand layout; the logits are scalars known up front, so the row's maximum
//                      and weights are formed lanes-over-edges before any feature row is touched
// (one workgroup per 64 columns of one folded vector: 4 channel groups x 64 columns, summed through LDS)
__global__ __launch_bounds__(256) void gat_fold_kernel(const float* __restrict__ w, const float* __restrict__ att_src,
                                                       const float* __restrict__ att_dst, int H, int C, int d,
                                                       float* __restrict__ u) {
  __shared__ float part[4][64];
  const int t = blockIdx.y, h = t % H;
  const int kx = threadIdx.x & 63, cg = threadIdx.x >> 6, k = blockIdx.x * 64 + kx;
  const float* att = (t < H ? att_src : att_dst) + h * C;
  float acc = 0.f;
  if (k < d) {
    const float* wp = w + (int64_t)h * C * d + k;
    for (int c = cg; c < C; c += 4) acc += att[c] * wp[(int64_t)c * d];
  }
  part[cg][kx] = acc;
  __syncthreads();
  if (cg == 0 && k < d) u[(int64_t)t * d + k] = part[0][kx] + part[1][kx] + part[2][kx] + part[3][kx];
}

// The two row-reading kernels cut a row into column chunks of 256 elements (one float4 per lane) and give every chunk
// its own waves: the per-wave state stays small and a chunk of a row is still 512 B (fp16) / 1 KB (fp32) of
// consecutive bytes.  Rows of a sampled batch are short (a handful of in-edges), so both kernels are organised around
// keeping feature-row loads in flight rather than around the rows: everything a row load depends on (ids, weights)
// is fetched lanes-over-items ahead of time.
// NV partial sums per lane -> totals over the wave by a halving butterfly: at each step a lane keeps the half of its
// values its own lane bit selects and hands the other half to its partner (NV - 1 exchanges in all, against 6 per
// value for plain xor reductions); the total of value v ends up in val[0] of the lanes [v * 64 / NV, (v + 1) * 64 / NV).
template <int NV, int OFF>
__device__ __forceinline__ void wave_halving_sum(float* val, int lane) {
  if constexpr (NV > 1) {
    const bool hi = (lane & OFF) != 0;
#pragma unroll
    for (int k = 0; k < NV / 2; ++k) {
      const float keep = hi ? val[k + NV / 2] : val[k], send = hi ? val[k] : val[k + NV / 2];
      val[k] = keep + __shfl_xor(send, OFF, 64);
    }
    wave_halving_sum<NV / 2, OFF / 2>(val, lane);
  } else if constexpr (OFF > 0) {
    val[0] += __shfl_xor(val[0], OFF, 64);
    wave_halving_sum<1, OFF / 2>(val, lane);
  }
}

// scores: sp[c][j][t] = <x[ids[j]][chunk c], u[t][chunk c]>  (partial dot products; the reader adds the chunks).
// A wave takes RB consecutive source rows: their ids in one load (lane r holds ids[jb + r]), then the rows R at a time.
template <typename T, int H>
__global__ __launch_bounds__(256) void gat_score_kernel(const T* __restrict__ src, int d,
                                                        const uint32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ n_dev, int64_t cap,
                                                        const float* __restrict__ u, float* __restrict__ sp) {
  constexpr int R = 8, RB = 16;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n = *n_dev, c = blockIdx.y, el = (c * 64 + lane) * 4;
  const bool on = el < d;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  float4_t uu[2 * H];
#pragma unroll
  for (int t = 0; t < 2 * H; ++t) uu[t] = on ? *reinterpret_cast<const float4_t*>(u + (int64_t)t * d + el) : zero4;
  float* out = sp + (int64_t)c * cap * 2 * H;
  for (int jb = wave * RB; jb < n; jb += waves_total * RB) {
    const uint32_t my_id = (lane < RB && jb + lane < n) ? ids[jb + lane] : 0u;
    const int nb = min(RB, n - jb);
    for (int r0 = 0; r0 < nb; r0 += R) {
      float4_t x[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t id = __shfl(my_id, (r0 + r) & 63, 64);  // (outside the `on` branch: every lane takes part)
        x[r] = on ? RowLoader<T>::load4(src + (int64_t)id * d, el) : zero4;
      }
      constexpr int NV = R * 2 * H;
      float val[NV];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < 2 * H; ++t)
          val[r * 2 * H + t] = x[r].x * uu[t].x + x[r].y * uu[t].y + x[r].z * uu[t].z + x[r].w * uu[t].w;
      wave_halving_sum<NV, 32>(val, lane);
      // value v = r * 2H + t of this group is element v of the R rows' contiguous output
      constexpr int LPV = 64 / NV;  // lanes holding the same value
      const int v = lane / LPV;
      if (lane % LPV == 0 && r0 + v / (2 * H) < nb) out[(int64_t)(jb + r0) * 2 * H + v] = val[0];
    }
  }
}

// attention weights of every in-edge, normalised, next to the feature row each edge reads:
//   gid_e[e] = ids[col[e]],  alpha_e[e][h] = softmax weight of edge e in its row (0 for a self loop among the edges),
//   alpha_self[i][h] = weight of the added self loop.   One wave per destination row, lanes over its edges.
template <int H>
__global__ __launch_bounds__(256) void gat_edge_weight_kernel(
    const uint32_t* __restrict__ ids, const float* __restrict__ sp, int n_chunks, int64_t cap,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ col,
    const int32_t* __restrict__ n_rows_dev, float slope, uint32_t* __restrict__ gid_e, float* __restrict__ alpha_e,
    float* __restrict__ alpha_self) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev;
  auto leaky = [&](float v) { return v > 0.f ? v : slope * v; };
  auto score = [&](int j, int t) {  // s[j][t] = sum of the chunk partials
    float v = 0.f;
    for (int c = 0; c < n_chunks; ++c) v += sp[((int64_t)c * cap + j) * 2 * H + t];
    return v;
  };
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    float sd[H], zs[H], mx[H], den[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      sd[h] = score(i, H + h);
      zs[h] = leaky(score(i, h) + sd[h]);  // the self loop's logit
      mx[h] = zs[h];
    }
    // the first 64 edges stay in registers
    const int j0 = lane < m ? col[e0 + lane] : i;
    float lg0[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      lg0[h] = j0 != i ? leaky(score(j0, h) + sd[h]) : -INFINITY;  // (self loops among the edges are dropped)
      mx[h] = fmaxf(mx[h], lg0[h]);
    }
    if (lane < m) gid_e[e0 + lane] = ids[j0];
    for (int c0 = 64; c0 < m; c0 += 64) {
      const int j = c0 + lane < m ? col[e0 + c0 + lane] : i;
      if (c0 + lane < m) gid_e[e0 + c0 + lane] = ids[j];
      if (j != i) {
#pragma unroll
        for (int h = 0; h < H; ++h) mx[h] = fmaxf(mx[h], leaky(score(j, h) + sd[h]));
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int h = 0; h < H; ++h) mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], off, 64));
    float w0[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      w0[h] = __expf(lg0[h] - mx[h]);  // (exp(-inf) = 0 for dropped lanes)
      den[h] = w0[h] + (lane == 0 ? __expf(zs[h] - mx[h]) : 0.f);
    }
    for (int c0 = 64; c0 < m; c0 += 64) {
      const int j = c0 + lane < m ? col[e0 + c0 + lane] : i;
      if (j != i) {
#pragma unroll
        for (int h = 0; h < H; ++h) den[h] += __expf(leaky(score(j, h) + sd[h]) - mx[h]);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int h = 0; h < H; ++h) den[h] += __shfl_xor(den[h], off, 64);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float inv = 1.0f / den[h];
      if (lane < m) alpha_e[(int64_t)(e0 + lane) * H + h] = w0[h] * inv;
      if (lane == 0) alpha_self[(int64_t)i * H + h] = __expf(zs[h] - mx[h]) * inv;
      for (int c0 = 64; c0 < m; c0 += 64) {
        if (c0 + lane >= m) continue;
        const int j = col[e0 + c0 + lane];
        alpha_e[(int64_t)(e0 + c0 + lane) * H + h] = j != i ? __expf(leaky(score(j, h) + sd[h]) - mx[h]) * inv : 0.f;
      }
    }
  }
}

// z_h[i] = alpha_self[i][h] x[ids[i]] + sum_e alpha_e[e][h] x[gid_e[e]]  for one 256-column chunk, written in the
// projection's tiled operand layout.  A wave walks its (row, chunk) items with the NEXT item's row bounds and edge
// list (ids + weights, lanes over edges; lane 63 carries the self loop) already on their way while the current
// item's feature rows are in flight, U at a time.
template <typename T, int H>
__global__ __launch_bounds__(256) void gat_input_gather_kernel(
    const T* __restrict__ src, int d, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ gid_e,
    const float* __restrict__ alpha_e, const float* __restrict__ alpha_self, int n_chunks,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ n_rows_dev,
    int nkc, int64_t head_stride, float* __restrict__ z) {
  constexpr int U = 8;  // feature rows in flight
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t work = (int64_t)n_rows * n_chunks;
  struct Bounds { int e0, m; };
  struct Edges { uint32_t gid; float w[H]; };
  auto bounds = [&](int64_t wk) {
    Bounds b{0, 0};
    if (wk < work) {
      const int i = (int)(wk / n_chunks);
      b.e0 = rowptr[i];
      b.m = rowend[i] - b.e0;
    }
    return b;
  };
  auto edges = [&](int64_t wk, const Bounds& b, int c0) {  // lanes over the edges [c0, c0 + 63) of the item's row
    Edges e;
    e.gid = 0u;
#pragma unroll
    for (int h = 0; h < H; ++h) e.w[h] = 0.f;
    if (wk >= work) return e;
    const int i = (int)(wk / n_chunks);
    if (lane == 63) {  // the self loop rides in the last lane of the first group
      if (c0 == 0) {
        e.gid = ids[i];
#pragma unroll
        for (int h = 0; h < H; ++h) e.w[h] = alpha_self[(int64_t)i * H + h];
      }
    } else if (c0 + lane < b.m) {
      e.gid = gid_e[b.e0 + c0 + lane];
#pragma unroll
      for (int h = 0; h < H; ++h) e.w[h] = alpha_e[(int64_t)(b.e0 + c0 + lane) * H + h];
    }
    return e;
  };
  int64_t wk = wave;
  Bounds b0 = bounds(wk), b1 = bounds(wk + waves_total);
  Edges e0 = edges(wk, b0, 0);
  for (; wk < work; wk += waves_total) {
    const Bounds b2 = bounds(wk + 2 * (int64_t)waves_total);
    const Edges e1 = edges(wk + waves_total, b1, 0);
    const int i = (int)(wk / n_chunks), c = (int)(wk - (int64_t)i * n_chunks);
    const int el = (c * 64 + lane) * 4;
    const bool on = el < d;
    float4_t acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) acc[h] = zero4;
    Edges cur = e0;
    for (int c0 = 0; c0 < b0.m + 1; c0 += 63) {  // (+1: the self loop; 63 edges per group, lane 63 = self / unused)
      if (c0 > 0) cur = edges(wk, b0, c0);
      const int mm = min(63, b0.m - c0);            // edges of this group; the self loop follows them in group 0
      const int cnt = mm + (c0 == 0 ? 1 : 0);
      for (int e = 0; e < cnt; e += U) {
        float4_t x[U];
        float we[U][H];
#pragma unroll
        for (int t = 0; t < U; ++t) {
          const int ee = e + t < mm ? e + t : 63;  // past the edges: the self-loop lane (weight 0 outside group 0)
          const T* row = src + (int64_t)__shfl(cur.gid, ee, 64) * d;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float v = __shfl(cur.w[h], ee, 64);
            we[t][h] = e + t < cnt ? v : 0.f;
          }
          x[t] = (on && e + t < cnt) ? RowLoader<T>::load4(row, el) : zero4;
        }
#pragma unroll
        for (int t = 0; t < U; ++t)
#pragma unroll
          for (int h = 0; h < H; ++h) acc[h] += we[t][h] * x[t];
      }
    }
    if (on) {
      float* tbase = z + ((int64_t)(i >> 7) * nkc) * 4096 + (i & 127) * 32 + (int64_t)(el >> 5) * 4096 + (el & 31);
#pragma unroll
      for (int h = 0; h < H; ++h) *reinterpret_cast<float4_t*>(tbase + h * head_stride) = acc[h];
    }
    b0 = b1;
    b1 = b2;
    e0 = e1;
  }
}

// sum over the 64 lanes of a wave, returned to every lane, on the VALU / SALU only: four DPP steps give every lane its
// row-of-16 total (xor 1, xor 2 inside quads, then the half-row and row mirrors), four v_readlane + scalar adds combine
// the rows.  (__shfl_xor lowers the wide steps to ds_bpermute: the one-pass kernel below spent a quarter of its
// instructions in the LDS pipeline on them.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF,
                                                                   0xF, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}

// ---- the same first layer in ONE row pass (the one-call plan's layer 0): the folded source vectors stay in
// registers and every edge's logit is formed from the feature row that has just been read for the aggregation, online
// softmax over the row.  No per-node score array exists,
// so the sources need no dense numbering: rows >= *n_local_dev hold GLOBAL source ids in `col` (the leaf-global union
// of the plan: pure leaves are never relabelled), rows below it local ids translated through gather_ids.  Every source
// row is read once per edge it appears in; z is written in the projection's tiled layout, one operand per head.
// A workgroup of P waves owns a destination row, wave c its c-th 256-column chunk (one float4 per lane): per-wave
// state is one accumulator per head and the U rows in flight, so many waves fit a SIMD and their loads overlap.  The
// logit needs the whole row's dot product: every wave reduces its chunk's partial, publishes it through LDS, and after
// one barrier per group of U edges all P waves add the partials in the same order — the softmax state is replicated,
// bit-identical, in every wave of the row.
// Four row elements as they are stored (the fp16 table: 8 bytes; widened where they are used, so that the U rows in
// flight cost half the registers and the conversions fold into the multiply-adds)
template <typename T>
struct RawRow4;
template <>
struct RawRow4<float> {
  typedef float4_t type;
  static __device__ __forceinline__ type load(const float* p, int e) { return *reinterpret_cast<const float4_t*>(p + e); }
  static __device__ __forceinline__ float4_t f4(const type& r) { return r; }
  static __device__ __forceinline__ type zero() { return float4_t{0.f, 0.f, 0.f, 0.f}; }
};
template <>
struct RawRow4<__half> {
  typedef uint2 type;
  static __device__ __forceinline__ type load(const __half* p, int e) { return *reinterpret_cast<const uint2*>(p + e); }
  static __device__ __forceinline__ float4_t f4(const type& r) {
    const __half2 lo = *reinterpret_cast<const __half2*>(&r.x), hi = *reinterpret_cast<const __half2*>(&r.y);
    return float4_t{__low2float(lo), __high2float(lo), __low2float(hi), __high2float(hi)};
  }
  static __device__ __forceinline__ type zero() { return make_uint2(0u, 0u); }
};

// ONE WAVE per destination row, lane l owning the V float4 chunks (v*64 + l) of the d-wide row (d <= 256 V).  The kernel
// is bound by VALU issue, not by the rows' bytes (scripts/micro/rowgather.hip: 1.5-KB rows drawn at random from a 47-GB
// table arrive at 6 TB/s; the three-waves-per-row form of round 3 ran at 2.7 — every wave repeated the softmax update,
// every edge cost 2H full-wave butterflies plus an LDS exchange and a barrier per group).  Here, per group of U edges:
// the U*H partial logits are reduced TOGETHER (gigl_wave_reduce16: 35 instructions per 16 totals), the totals are wave-
// uniform scalars (v_readlane), and the running softmax is rescaled once per group — the group's largest logit first,
// then acc = acc*sc + sum_t pw_t x_t.  No LDS, no barrier.
template <typename T, int V, int H>
__global__ __launch_bounds__(256) void gat_input_online_kernel(
    const T* __restrict__ src, int d, const uint32_t* __restrict__ gather_ids, const int32_t* __restrict__ n_local_dev,
    const float* __restrict__ u, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
    const int32_t* __restrict__ col, const int32_t* __restrict__ n_rows_dev, float slope, int nkc, int64_t head_stride,
    float* __restrict__ z) {
  // feature rows in flight per wave: 8 while they, the accumulators and the source vectors fit the registers of two
  // waves per SIMD without spilling (the fp16 table at d = 768, two heads: 204 VGPRs; forcing three waves spilled and
  // ran 30 % slower), 4 for the wide fp32 / four-head shapes
  constexpr int U = (V * H * (sizeof(T) == 2 ? 2 : 4) > 12) ? 4 : 8;
  constexpr int NV = U * H, NC = (NV + 15) / 16;
  typedef RawRow4<T> RR;
  typedef typename RR::type raw_t;
  const int lane = threadIdx.x & 63;
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int n_rows = *n_rows_dev;
  const int n_local = n_local_dev ? *n_local_dev : 0x7FFFFFFF;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  auto leaky = [&](float v) { return v > 0.f ? v : slope * v; };
  auto dot4 = [](const float4_t& a, const float4_t& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; };
  auto lane_value = [](float v, int from) {  // (wave-uniform: lands in a scalar register)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), from));
  };
  int el[V];
  bool on[V];
  float4_t us[H][V];  // (the destination-side vectors are used once per row: re-read from L2 there, not held)
#pragma unroll
  for (int v = 0; v < V; ++v) {
    el[v] = (v * 64 + lane) * 4;
    on[v] = el[v] < d;
#pragma unroll
    for (int h = 0; h < H; ++h) us[h][v] = on[v] ? *reinterpret_cast<const float4_t*>(u + (int64_t)h * d + el[v]) : zero4;
  }
  for (int i = wave; i < n_rows; i += waves) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    const uint32_t self_gid = gather_ids[i];
    const bool local = i < n_local;
    float sd[H], mx[H], den[H];
    float4_t acc[H][V];
    {
      float pv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) pv[k] = 0.f;
      const T* row = src + (int64_t)self_gid * d;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float4_t xs = on[v] ? RR::f4(RR::load(row, el[v])) : zero4;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const float4_t udv = on[v] ? *reinterpret_cast<const float4_t*>(u + (int64_t)(H + h) * d + el[v]) : zero4;
          pv[h] += dot4(xs, us[h][v]);
          pv[H + h] += dot4(xs, udv);
          acc[h][v] = xs;  // the self loop opens the running softmax with weight 1
        }
      }
      const float tot = gigl_wave_reduce16(pv);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float fs = lane_value(tot, h << 2), fd = lane_value(tot, (H + h) << 2);
        sd[h] = fd;
        mx[h] = leaky(fs + fd);
        den[h] = 1.f;
      }
    }
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int mm = min(64, m - c0);
      uint32_t gid = self_gid;  // (lanes past the row and self loops among the edges: skipped below)
      bool take = false;
      if (lane < mm) {
        const int j = col[e0 + c0 + lane];
        if (local) {
          take = j != i;
          gid = gather_ids[j];
        } else {
          gid = (uint32_t)j;
          take = gid != self_gid;
        }
      }
      const unsigned long long keep = __ballot(take);
      for (int e = 0; e < mm; e += U) {
        raw_t x[U][V];
        bool live[U];
#pragma unroll
        for (int t = 0; t < U; ++t) {
          live[t] = e + t < mm && ((keep >> (e + t)) & 1ull);  // (uniform over the wave)
          const T* row = src + (int64_t)__shfl(gid, (e + t) & 63, 64) * d;
#pragma unroll
          for (int v = 0; v < V; ++v) x[t][v] = (live[t] && on[v]) ? RR::load(row, el[v]) : RR::zero();
        }
        // every edge's source logit, all heads: this lane's partials reduced together, one total per quad of lanes
        float fsum[NC * 16];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          float pv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int idx = cc * 16 + k;  // = t * H + h
            float a = 0.f;
            if (idx < NV) {
#pragma unroll
              for (int v = 0; v < V; ++v) a += dot4(RR::f4(x[idx / H][v]), us[idx % H][v]);
            }
            pv[k] = a;
          }
          const float tot = gigl_wave_reduce16(pv);
#pragma unroll
          for (int k = 0; k < 16; ++k) fsum[cc * 16 + k] = lane_value(tot, k << 2);
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float zl[U];
          float gmax = mx[h];
#pragma unroll
          for (int t = 0; t < U; ++t) {
            zl[t] = live[t] ? leaky(fsum[t * H + h] + sd[h]) : -__builtin_inff();
            gmax = fmaxf(gmax, zl[t]);
          }
          const float sc = __expf(mx[h] - gmax);
          float dn = den[h] * sc;
          float pw[U];
#pragma unroll
          for (int t = 0; t < U; ++t) {
            pw[t] = __expf(zl[t] - gmax);  // (0 for an edge that is not there)
            dn += pw[t];
          }
#pragma unroll
          for (int v = 0; v < V; ++v) {
            float4_t a = acc[h][v] * sc;
#pragma unroll
            for (int t = 0; t < U; ++t) a += pw[t] * RR::f4(x[t][v]);
            acc[h][v] = a;
          }
          den[h] = dn;
          mx[h] = gmax;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (!on[v]) continue;
      // (nkc == 0: plain rows, z[h][i][0:d] — what the training path keeps for the projection's weight gradient)
      float* tbase = nkc ? z + ((int64_t)(i >> 7) * nkc) * 4096 + (i & 127) * 32 + (int64_t)(el[v] >> 5) * 4096 + (el[v] & 31)
                         : z + (int64_t)i * d + el[v];
#pragma unroll
      for (int h = 0; h < H; ++h) *reinterpret_cast<float4_t*>(tbase + h * head_stride) = acc[h][v] * (1.0f / den[h]);
    }
  }
}

// ---- backward of that aggregation w.r.t. the folded attention vectors (the rows themselves are inputs: no gradient).
//   z_i^h = sum_e alpha_e^h x_e,  alpha = softmax_e leaky(<x_e, us^h> + <x_i, ud^h>)  (e over the in-edges and the self loop)
//   given dz_i^h:  d alpha_e = <dz_i^h, x_e>,  S = sum_e alpha_e d alpha_e,  dpre_e = alpha_e (d alpha_e - S) leaky'(pre_e),
//                  d us^h += sum_e dpre_e x_e,   d ud^h += (sum_e dpre_e) x_i
// One wave per destination row (the forward's lane layout), two sweeps over the row's edges: the first forms every
// edge's logit and d alpha from the source row as it is read (reduced together, gigl_wave_reduce16) and parks the two
// scalars per head in edge_scr; the second re-reads the rows (L2 / MALL hits) and accumulates d us.  A wave keeps its
// d us / d ud sums in registers over all its rows and adds them to `du` once, at the end (fp32 atomics).
template <typename T, int V, int H>
__global__ __launch_bounds__(256) void gat_input_backward_kernel(
    const T* __restrict__ src, int d, const uint32_t* __restrict__ gather_ids, const float* __restrict__ u,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ col,
    const int32_t* __restrict__ n_rows_dev, float slope, const float* __restrict__ dz, int64_t head_stride,
    float* __restrict__ edge_scr, float* __restrict__ du) {
  constexpr int U = 4;                                  // feature rows in flight per wave (first sweep)
  constexpr int U2 = (V * H * (sizeof(T) == 2 ? 2 : 4) > 12) ? 4 : 8;  // ... second sweep (no logits to hold)
  constexpr int NV = U * 2 * H, NC = (NV + 15) / 16;    // (logit, d alpha) per head and edge of a group
  typedef RawRow4<T> RR;
  typedef typename RR::type raw_t;
  const int lane = threadIdx.x & 63;
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int n_rows = *n_rows_dev;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  auto leaky = [&](float v) { return v > 0.f ? v : slope * v; };
  auto dot4 = [](const float4_t& a, const float4_t& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; };
  auto lane_value = [](float v, int from) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), from));
  };
  int el[V];
  bool on[V];
  float4_t us[H][V], acc_s[H][V], acc_d[H][V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    el[v] = (v * 64 + lane) * 4;
    on[v] = el[v] < d;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      us[h][v] = on[v] ? *reinterpret_cast<const float4_t*>(u + (int64_t)h * d + el[v]) : zero4;
      acc_s[h][v] = zero4;
      acc_d[h][v] = zero4;
    }
  }
  for (int i = wave; i < n_rows; i += waves) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    const uint32_t self_gid = gather_ids[i];
    float4_t xs[V], dzv[H][V];
    float sd[H], das[H], mx[H], den[H], pre_self[H];
    {
      float pv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) pv[k] = 0.f;
      const T* row = src + (int64_t)self_gid * d;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        xs[v] = on[v] ? RR::f4(RR::load(row, el[v])) : zero4;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          dzv[h][v] = on[v] ? *reinterpret_cast<const float4_t*>(dz + h * head_stride + (int64_t)i * d + el[v]) : zero4;
          const float4_t udv = on[v] ? *reinterpret_cast<const float4_t*>(u + (int64_t)(H + h) * d + el[v]) : zero4;
          pv[h] += dot4(xs[v], us[h][v]);
          pv[H + h] += dot4(xs[v], udv);
          pv[2 * H + h] += dot4(xs[v], dzv[h][v]);
        }
      }
      const float tot = gigl_wave_reduce16(pv);  // (3H <= 12 values)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float fs = lane_value(tot, h << 2);
        sd[h] = lane_value(tot, (H + h) << 2);
        das[h] = lane_value(tot, (2 * H + h) << 2);
        pre_self[h] = fs + sd[h];
        mx[h] = leaky(pre_self[h]);
        den[h] = 1.f;
      }
    }
    // a row whose dz is all zero adds nothing (padding rows, rows no root depends on): skipped, wave-uniformly
    {
      bool nz = false;
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int v = 0; v < V; ++v) nz |= dzv[h][v].x != 0.f || dzv[h][v].y != 0.f || dzv[h][v].z != 0.f || dzv[h][v].w != 0.f;
      if (__ballot(nz) == 0ull) continue;
    }
    // ---- sweep 1: logits and d alpha of every edge -> edge_scr[e][h] = (zl, d alpha); running max / denominator
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int mm = min(64, m - c0);
      uint32_t gid = self_gid;
      bool take = false;
      if (lane < mm) {
        const int j = col[e0 + c0 + lane];
        take = j != i;
        gid = gather_ids[j];
      }
      const unsigned long long keep = __ballot(take);
      for (int e = 0; e < mm; e += U) {
        raw_t x[U][V];
        bool live[U];
#pragma unroll
        for (int t = 0; t < U; ++t) {
          live[t] = e + t < mm && ((keep >> (e + t)) & 1ull);
          const T* row = src + (int64_t)__shfl(gid, (e + t) & 63, 64) * d;
#pragma unroll
          for (int v = 0; v < V; ++v) x[t][v] = (live[t] && on[v]) ? RR::load(row, el[v]) : RR::zero();
        }
        float vals[NC * 16];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          float pv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int idx = cc * 16 + k;  // = (t * H + h) * 2 + which
            float a = 0.f;
            if (idx < NV) {
              const int t = idx / (2 * H), h = (idx / 2) % H, which = idx & 1;
#pragma unroll
              for (int v = 0; v < V; ++v) a += dot4(RR::f4(x[t][v]), which ? dzv[h][v] : us[h][v]);
            }
            pv[k] = a;
          }
          const float tot = gigl_wave_reduce16(pv);
#pragma unroll
          for (int k = 0; k < 16; ++k) vals[cc * 16 + k] = lane_value(tot, k << 2);
        }
#pragma unroll
        for (int t = 0; t < U; ++t) {
          if (!live[t]) continue;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float zl = leaky(vals[(t * H + h) * 2] + sd[h]), da = vals[(t * H + h) * 2 + 1];
            const float nm = fmaxf(mx[h], zl);
            den[h] = den[h] * __expf(mx[h] - nm) + __expf(zl - nm);
            mx[h] = nm;
            if (lane == 0) {
              float* sp = edge_scr + ((int64_t)(e0 + c0 + e + t) * H + h) * 2;
              sp[0] = zl;
              sp[1] = da;
            }
          }
        }
      }
    }
    // ---- S = sum_e alpha_e d alpha_e (edges: a lane per edge of a chunk; plus the self loop)
    float inv[H], al_self[H], S[H], dpre_self[H], sum_dpre[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      inv[h] = 1.0f / (den[h] + 1e-16f);
      al_self[h] = __expf(leaky(pre_self[h]) - mx[h]) * inv[h];
      S[h] = al_self[h] * das[h];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (lane 0's parked scalars, read below by their lanes)
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int mm = min(64, m - c0);
      const bool mine = lane < mm && col[e0 + c0 + lane] != i;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float part = 0.f;
        if (mine) {
          const float* sp = edge_scr + ((int64_t)(e0 + c0 + lane) * H + h) * 2;
          part = __expf(sp[0] - mx[h]) * inv[h] * sp[1];
        }
        S[h] += wave_sum_dpp(part);
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      dpre_self[h] = al_self[h] * (das[h] - S[h]) * (pre_self[h] > 0.f ? 1.f : slope);
      sum_dpre[h] = dpre_self[h];
    }
    // ---- sweep 2: d us += dpre_e x_e
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int mm = min(64, m - c0);
      uint32_t gid = self_gid;
      bool take = false;
      if (lane < mm) {
        const int j = col[e0 + c0 + lane];
        take = j != i;
        gid = gather_ids[j];
      }
      float dpre[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        dpre[h] = 0.f;
        if (take) {
          const float* sp = edge_scr + ((int64_t)(e0 + c0 + lane) * H + h) * 2;
          const float zl = sp[0];
          dpre[h] = __expf(zl - mx[h]) * inv[h] * (sp[1] - S[h]) * (zl > 0.f ? 1.f : slope);
        }
        sum_dpre[h] += wave_sum_dpp(dpre[h]);
      }
      const unsigned long long keep = __ballot(take);
      for (int e = 0; e < mm; e += U2) {
        raw_t x[U2][V];
        bool live[U2];
#pragma unroll
        for (int t = 0; t < U2; ++t) {
          live[t] = e + t < mm && ((keep >> (e + t)) & 1ull);
          const T* row = src + (int64_t)__shfl(gid, (e + t) & 63, 64) * d;
#pragma unroll
          for (int v = 0; v < V; ++v) x[t][v] = (live[t] && on[v]) ? RR::load(row, el[v]) : RR::zero();
        }
#pragma unroll
        for (int t = 0; t < U2; ++t) {
          if (!live[t]) continue;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float dp = lane_value(dpre[h], (e + t) & 63);  // (a uniform lane: v_readlane, not an LDS permute)
#pragma unroll
            for (int v = 0; v < V; ++v) acc_s[h][v] += dp * RR::f4(x[t][v]);
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        acc_s[h][v] += dpre_self[h] * xs[v];
        acc_d[h][v] += sum_dpre[h] * xs[v];
      }
  }
  // the workgroup's four waves add their sums in LDS first, in wave order, and one wave adds the total to `du`: every
  // atomic lands on one of only 2 H d addresses, so their number — (workgroups) x 2 H d — is what this tail costs
  __shared__ float4_t s_red[2 * H * V * 64];
  const int wv = threadIdx.x >> 6;
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float4_t& rs = s_red[(h * V + v) * 64 + lane];
          float4_t& rd = s_red[((H + h) * V + v) * 64 + lane];
          rs = w == 0 ? acc_s[h][v] : rs + acc_s[h][v];
          rd = w == 0 ? acc_d[h][v] : rd + acc_d[h][v];
        }
    }
    __syncthreads();
  }
  if (wv != 0) return;
#pragma unroll
  for (int h = 0; h < 2 * H; ++h)
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (!on[v]) continue;
      const float4_t t = s_red[(h * V + v) * 64 + lane];
      float* pu = du + (int64_t)h * d + el[v];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (t[q] != 0.f) atomicAdd(pu + q, t[q]);
    }
}

// ---- the same backward in ONE sweep over the source rows (round 6).  The two-sweep kernel reads every source row twice
// because dpre_e = alpha_e (d alpha_e - S) leaky'(pre_e) needs S = sum_e alpha_e d alpha_e — known only after the row's last
// edge.  But  sum_e dpre_e x_e = ( sum_e w_e l_e d alpha_e x_e  -  S sum_e w_e l_e x_e ) / den   with w_e = exp(zl_e - max),
// l_e = leaky'(pre_e), den = sum_e w_e, S = (sum_e w_e d alpha_e) / den: two vector accumulators A, B (and three scalars) per
// head, kept relative to the RUNNING maximum and rescaled when it grows — the online softmax of the forward applied to its
// own backward.  Each row is read once; no per-edge scalars are parked (edge_scr is not used).  The self loop is the first
// "edge" (x_e = x_i, w = 1).  Same lane layout, same cross-row accumulation and final atomics as the two-sweep kernel.
template <typename T, int V, int H>
__global__ __launch_bounds__(256) void gat_input_backward_onepass_kernel(
    const T* __restrict__ src, int d, const uint32_t* __restrict__ gather_ids, const float* __restrict__ u,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend, const int32_t* __restrict__ col,
    const int32_t* __restrict__ n_rows_dev, float slope, const float* __restrict__ dz, int64_t head_stride,
    float* __restrict__ du) {
  constexpr int U = 4;                                  // feature rows in flight per wave
  constexpr int NV = U * 2 * H, NC = (NV + 15) / 16;    // (logit, d alpha) per head and edge of a group
  typedef RawRow4<T> RR;
  typedef typename RR::type raw_t;
  const int lane = threadIdx.x & 63;
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int n_rows = *n_rows_dev;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  auto leaky = [&](float v) { return v > 0.f ? v : slope * v; };
  auto dot4 = [](const float4_t& a, const float4_t& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; };
  auto lane_value = [](float v, int from) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), from));
  };
  int el[V];
  bool on[V];
  float4_t us[H][V], acc_s[H][V], acc_d[H][V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    el[v] = (v * 64 + lane) * 4;
    on[v] = el[v] < d;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      us[h][v] = on[v] ? *reinterpret_cast<const float4_t*>(u + (int64_t)h * d + el[v]) : zero4;
      acc_s[h][v] = zero4;
      acc_d[h][v] = zero4;
    }
  }
  for (int i = wave; i < n_rows; i += waves) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    const uint32_t self_gid = gather_ids[i];
    float4_t xs[V], dzv[H][V], A[H][V], B[H][V];
    float sd[H], mx[H], den[H], Sw[H], a_s[H], b_s[H];
    {
      float pv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) pv[k] = 0.f;
      const T* row = src + (int64_t)self_gid * d;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        xs[v] = on[v] ? RR::f4(RR::load(row, el[v])) : zero4;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          dzv[h][v] = on[v] ? *reinterpret_cast<const float4_t*>(dz + h * head_stride + (int64_t)i * d + el[v]) : zero4;
          const float4_t udv = on[v] ? *reinterpret_cast<const float4_t*>(u + (int64_t)(H + h) * d + el[v]) : zero4;
          pv[h] += dot4(xs[v], us[h][v]);
          pv[H + h] += dot4(xs[v], udv);
          pv[2 * H + h] += dot4(xs[v], dzv[h][v]);
        }
      }
      const float tot = gigl_wave_reduce16(pv);  // (3H <= 12 values)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float fs = lane_value(tot, h << 2);
        sd[h] = lane_value(tot, (H + h) << 2);
        const float das = lane_value(tot, (2 * H + h) << 2);
        const float pre_self = fs + sd[h];
        const float lk = pre_self > 0.f ? 1.f : slope;
        mx[h] = leaky(pre_self);
        den[h] = 1.f;     // the self loop: w = exp(zl_self - max) = 1
        Sw[h] = das;
        a_s[h] = lk * das;
        b_s[h] = lk;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          A[h][v] = (lk * das) * xs[v];
          B[h][v] = lk * xs[v];
        }
      }
    }
    // a row whose dz is all zero adds nothing (padding rows, rows no root depends on): skipped, wave-uniformly
    {
      bool nz = false;
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int v = 0; v < V; ++v) nz |= dzv[h][v].x != 0.f || dzv[h][v].y != 0.f || dzv[h][v].z != 0.f || dzv[h][v].w != 0.f;
      if (__ballot(nz) == 0ull) continue;
    }
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int mm = min(64, m - c0);
      uint32_t gid = self_gid;
      bool take = false;
      if (lane < mm) {
        const int j = col[e0 + c0 + lane];
        take = j != i;
        gid = gather_ids[j];
      }
      const unsigned long long keep = __ballot(take);
      for (int e = 0; e < mm; e += U) {
        raw_t x[U][V];
        bool live[U];
#pragma unroll
        for (int t = 0; t < U; ++t) {
          live[t] = e + t < mm && ((keep >> (e + t)) & 1ull);
          const T* row = src + (int64_t)__shfl(gid, (e + t) & 63, 64) * d;
#pragma unroll
          for (int v = 0; v < V; ++v) x[t][v] = (live[t] && on[v]) ? RR::load(row, el[v]) : RR::zero();
        }
        float vals[NC * 16];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          float pv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int idx = cc * 16 + k;  // = (t * H + h) * 2 + which
            float a = 0.f;
            if (idx < NV) {
              const int t = idx / (2 * H), h = (idx / 2) % H, which = idx & 1;
#pragma unroll
              for (int v = 0; v < V; ++v) a += dot4(RR::f4(x[t][v]), which ? dzv[h][v] : us[h][v]);
            }
            pv[k] = a;
          }
          const float tot = gigl_wave_reduce16(pv);
#pragma unroll
          for (int k = 0; k < 16; ++k) vals[cc * 16 + k] = lane_value(tot, k << 2);
        }
#pragma unroll
        for (int t = 0; t < U; ++t) {
          if (!live[t]) continue;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float pre = vals[(t * H + h) * 2] + sd[h], da = vals[(t * H + h) * 2 + 1];
            const float zl = leaky(pre), lk = pre > 0.f ? 1.f : slope;
            if (zl > mx[h]) {  // (wave-uniform: the values are broadcasts) the running maximum grows: everything rescales
              const float sc = __expf(mx[h] - zl);
              den[h] *= sc;
              Sw[h] *= sc;
              a_s[h] *= sc;
              b_s[h] *= sc;
#pragma unroll
              for (int v = 0; v < V; ++v) {
                A[h][v] = sc * A[h][v];
                B[h][v] = sc * B[h][v];
              }
              mx[h] = zl;
            }
            const float w = __expf(zl - mx[h]), c = w * lk, cd = c * da;
            den[h] += w;
            Sw[h] += w * da;
            a_s[h] += cd;
            b_s[h] += c;
#pragma unroll
            for (int v = 0; v < V; ++v) {
              const float4_t xv = RR::f4(x[t][v]);
              A[h][v] += cd * xv;
              B[h][v] += c * xv;
            }
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float inv = 1.0f / (den[h] + 1e-16f);
      const float S = Sw[h] * inv;
      const float sum_dpre = (a_s[h] - S * b_s[h]) * inv;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        acc_s[h][v] += inv * (A[h][v] - S * B[h][v]);
        acc_d[h][v] += sum_dpre * xs[v];
      }
    }
  }
  __shared__ float4_t s_red[2 * H * V * 64];
  const int wv = threadIdx.x >> 6;
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float4_t& rs = s_red[(h * V + v) * 64 + lane];
          float4_t& rd = s_red[((H + h) * V + v) * 64 + lane];
          rs = w == 0 ? acc_s[h][v] : rs + acc_s[h][v];
          rd = w == 0 ? acc_d[h][v] : rd + acc_d[h][v];
        }
    }
    __syncthreads();
  }
  if (wv != 0) return;
#pragma unroll
  for (int h = 0; h < 2 * H; ++h)
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (!on[v]) continue;
      const float4_t t = s_red[(h * V + v) * 64 + lane];
      float* pu = du + (int64_t)h * d + el[v];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (t[q] != 0.f) atomicAdd(pu + q, t[q]);
    }
}

// ---- fast GAT path: one wave per row, every lane owns V float4 chunks of the H*C-wide row (chunk q = v*64 + lane
// covers channels [4q, 4q+4), all inside one head because C % 4 == 0), so all heads advance together and each source
// row is read exactly once, coalesced.  Shapes: C/4 a power of two <= 64 (a head = C/4 adjacent lanes of one chunk
// row), or C a multiple of 256 (a head = whole chunk rows).  Anything else takes the generic kernels above.
struct GatShape {
  int V;        // float4 chunks per lane
  int group;    // lanes per head inside a chunk row (64 when a head spans whole chunk rows)
  int rows_per_head;  // chunk rows per head (1 unless C is a multiple of 256)
};
static bool gat_fast_shape(int heads, int C, GatShape& g) {
  const int HC = heads * C;
  if (C % 4 || HC % 4) return false;
  const int chunks = HC / 4;
  g.V = (chunks + 63) / 64;
  if (g.V != 1 && g.V != 2 && g.V != 4) return false;
  if (C % 256 == 0) {
    g.group = 64;
    g.rows_per_head = C / 256;
    return chunks % 64 == 0;
  }
  const int gl = C / 4;
  if (gl > 64 || (gl & (gl - 1))) return false;
  g.group = gl;
  g.rows_per_head = 1;
  return true;
}
__device__ __forceinline__ int gat_head_of(int v, int lane, int group, int rows_per_head) {
  return rows_per_head > 1 ? v / rows_per_head : (v * 64 + lane) / group;
}

// a_src / a_dst for every node: lanes over channels, segmented shuffle reduction per head
template <int V>
__global__ __launch_bounds__(256) void gat_alpha_fast_kernel(const float* __restrict__ h,
                                                             const float* __restrict__ att_src,
                                                             const float* __restrict__ att_dst,
                                                             const int32_t* __restrict__ n_dev, int heads, int C,
                                                             int group, int rows_per_head,
                                                             float* __restrict__ a_src, float* __restrict__ a_dst) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n = *n_dev, HC = heads * C, chunks = HC >> 2;
  float4 ws[V], wd[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int q = v * 64 + lane;
    ws[v] = q < chunks ? ((const float4*)att_src)[q] : float4{0, 0, 0, 0};
    wd[v] = q < chunks ? ((const float4*)att_dst)[q] : float4{0, 0, 0, 0};
  }
  for (int i = wave; i < n; i += waves_total) {
    const float4* row = (const float4*)(h + (int64_t)i * HC);
    float ps[V], pd[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int q = v * 64 + lane;
      const float4 x = q < chunks ? row[q] : float4{0, 0, 0, 0};
      ps[v] = x.x * ws[v].x + x.y * ws[v].y + x.z * ws[v].z + x.w * ws[v].w;
      pd[v] = x.x * wd[v].x + x.y * wd[v].y + x.z * wd[v].z + x.w * wd[v].w;
    }
    if (rows_per_head > 1) {  // a head = rows_per_head whole chunk rows
      for (int hd = 0; hd < heads; ++hd) {
        float s = 0.f, d = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (v / rows_per_head == hd) {
            s += ps[v];
            d += pd[v];
          }
        for (int off = 32; off > 0; off >>= 1) {
          s += __shfl_xor(s, off, 64);
          d += __shfl_xor(d, off, 64);
        }
        if (lane == 0) {
          a_src[(int64_t)i * heads + hd] = s;
          a_dst[(int64_t)i * heads + hd] = d;
        }
      }
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float s = ps[v], d = pd[v];
        for (int off = group >> 1; off > 0; off >>= 1) {
          s += __shfl_xor(s, off, 64);
          d += __shfl_xor(d, off, 64);
        }
        const int q = v * 64 + lane;
        if ((lane & (group - 1)) == 0 && q < chunks) {
          const int hd = q / group;
          a_src[(int64_t)i * heads + hd] = s;
          a_dst[(int64_t)i * heads + hd] = d;
        }
      }
    }
  }
}

// attention + aggregation, single pass (online softmax: running max m, denominator l, weighted sum acc, rescaled when
// the max grows).  The self loop is folded in LAST so that, with edge features, the mean of the row's a_edge is
// known when its logit is formed.  a_edge == NULL: no edge term.
// MSG (EdgeAttrGATConv): messages are h_j + W_msg e_ij, i.e. out_i += W_msg z_i with z_i = sum_j alpha_ij e_ij (per head,
// De values) — accumulated online next to acc: the lanes of a head split the De components (component k lives in lane
// k % group of the head's group, register k / group; at most GAT_ZR registers), the self loop adds alpha_self * mean e;
// the final product reads the TRANSPOSED weight wt[k][H*C] so that a k-row is one coalesced float4 per lane.
constexpr int GAT_ZR = 4;
constexpr int GAT_HEAVY_MIN = 128;   // in-edges from which a row is split over the waves of a workgroup
constexpr int GAT_HEAVY_WAVES = 16;  // waves (of a 1024-thread workgroup) sharing one hub row
template <int V, bool MSG>
__global__ __launch_bounds__(256) void gat_gather_fast_kernel(
    const float* __restrict__ h, const float* __restrict__ a_src, const float* __restrict__ a_dst,
    const float* __restrict__ a_edge, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
    const int32_t* __restrict__ col, const int32_t* __restrict__ n_rows_dev, int heads, int C, int group,
    int rows_per_head, float slope, const float* __restrict__ bias, int act, const float* __restrict__ edge_attr,
    int De, const float* __restrict__ wt, float* __restrict__ out, int32_t* __restrict__ heavy_count,
    int32_t* __restrict__ heavy_list) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const int n_rows = *n_rows_dev, HC = heads * C, chunks = HC >> 2;
  int hd[V];
  bool on[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    on[v] = v * 64 + lane < chunks;
    hd[v] = on[v] ? gat_head_of(v, lane, group, rows_per_head) : 0;
  }
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    if (!MSG && heavy_list && m >= GAT_HEAVY_MIN) {  // hub rows: a whole workgroup each (gat_gather_heavy_kernel)
      if (lane == 0) heavy_list[atomicAdd(heavy_count, 1)] = i;
      continue;
    }
    float ad[V], mx[V], den[V], sum_ae[V];
    float4 acc[V];
    float zz[MSG ? V : 1][GAT_ZR], es[GAT_ZR];  // z per chunk row (rows of one head carry the same values), sum of e
#pragma unroll
    for (int v = 0; v < V; ++v) {
      ad[v] = a_dst[(int64_t)i * heads + hd[v]];
      mx[v] = -INFINITY;
      den[v] = 0.f;
      sum_ae[v] = 0.f;
      acc[v] = float4{0, 0, 0, 0};
      if (MSG) {
#pragma unroll
        for (int r = 0; r < GAT_ZR; ++r) zz[v][r] = 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < GAT_ZR; ++r) es[r] = 0.f;
    const int kl = lane & (group - 1);  // this lane's component slot inside its head group
    int cnt = 0;
    constexpr int U = 4;  // edges in flight
    for (int e = 0; e < m; e += U) {
      int j[U];
      float4 x[U][V];
      float lg[U][V];
      float ev[MSG ? U : 1][GAT_ZR];
#pragma unroll
      for (int t = 0; t < U; ++t) {
        j[t] = e + t < m ? col[e0 + e + t] : -1;
        if (j[t] == i) j[t] = -1;  // self loops are removed, one is added back below
      }
      if (MSG) {
#pragma unroll
        for (int t = 0; t < U; ++t) {
          if (j[t] < 0) continue;
#pragma unroll
          for (int r = 0; r < GAT_ZR; ++r) {
            const int k = r * group + kl;
            ev[t][r] = k < De ? edge_attr[(int64_t)(e0 + e + t) * De + k] : 0.f;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        if (j[t] < 0) continue;
        const float4* row = (const float4*)(h + (int64_t)j[t] * HC);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          x[t][v] = on[v] ? row[v * 64 + lane] : float4{0, 0, 0, 0};
          lg[t][v] = a_src[(int64_t)j[t] * heads + hd[v]] + ad[v];
          if (a_edge) {
            const float ae = a_edge[(int64_t)(e0 + e + t) * heads + hd[v]];
            lg[t][v] += ae;
            sum_ae[v] += ae;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        if (j[t] < 0) continue;
        ++cnt;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float z = lg[t][v];
          z = z > 0.f ? z : slope * z;
          const float nm = fmaxf(mx[v], z);
          const float sc = __expf(mx[v] - nm), pw = __expf(z - nm);
          den[v] = den[v] * sc + pw;
          acc[v].x = acc[v].x * sc + pw * x[t][v].x;
          acc[v].y = acc[v].y * sc + pw * x[t][v].y;
          acc[v].z = acc[v].z * sc + pw * x[t][v].z;
          acc[v].w = acc[v].w * sc + pw * x[t][v].w;
          mx[v] = nm;
          if (MSG) {
#pragma unroll
            for (int r = 0; r < GAT_ZR; ++r) zz[v][r] = zz[v][r] * sc + pw * ev[t][r];
          }
        }
        if (MSG) {
#pragma unroll
          for (int r = 0; r < GAT_ZR; ++r) es[r] += ev[t][r];
        }
      }
    }
    // the self loop (mean edge attribute of the row -> mean a_edge), then normalise
    const float4* self = (const float4*)(h + (int64_t)i * HC);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (!MSG && !on[v]) continue;  // (MSG: whole chunk rows only, see launch_gat_fast — the shuffles need every lane)
      float z = a_src[(int64_t)i * heads + hd[v]] + ad[v] + (a_edge && cnt > 0 ? sum_ae[v] / (float)cnt : 0.f);
      z = z > 0.f ? z : slope * z;
      const float nm = fmaxf(mx[v], z);
      const float sc = __expf(mx[v] - nm), pw = __expf(z - nm);
      const float4 xs = self[v * 64 + lane];
      const float inv = 1.0f / (den[v] * sc + pw + 1e-16f);
      float4 o;
      o.x = (acc[v].x * sc + pw * xs.x) * inv;
      o.y = (acc[v].y * sc + pw * xs.y) * inv;
      o.z = (acc[v].z * sc + pw * xs.z) * inv;
      o.w = (acc[v].w * sc + pw * xs.w) * inv;
      const int q = v * 64 + lane;
      if (MSG) {
        const float mean_w = cnt > 0 ? pw / (float)cnt : 0.f;
        const int base = lane & ~(group - 1);
#pragma unroll
        for (int r = 0; r < GAT_ZR; ++r) {
          const float zr = (zz[v][r] * sc + mean_w * es[r]) * inv;
          for (int kk = 0; kk < group; ++kk) {
            const int k = r * group + kk;
            if (k >= De) break;
            const float zk = __shfl(zr, base + kk, 64);
            const float4 w4 = ((const float4*)(wt + (int64_t)k * HC))[q];
            o.x += w4.x * zk;
            o.y += w4.y * zk;
            o.z += w4.z * zk;
            o.w += w4.w * zk;
          }
        }
      }
      if (bias) {
        const float4 b = ((const float4*)bias)[q];
        o.x += b.x;
        o.y += b.y;
        o.z += b.z;
        o.w += b.w;
      }
      if (act == 1) {
        o.x = fmaxf(o.x, 0.f);
        o.y = fmaxf(o.y, 0.f);
        o.z = fmaxf(o.z, 0.f);
        o.w = fmaxf(o.w, 0.f);
      }
      ((float4*)(out + (int64_t)i * HC))[q] = o;
    }
  }
}

// the dense tail of the GAT layer's backward in ONE pass over the projected rows (what torch ran as ~10 elementwise /
// reduction passes over [nodes][H*C] tensors): per row r and column c of head h
//   dxw[r][c] = dh[r][c] + ds[r][h] att_src[c] + dd[r][h] att_dst[c]        (in place in dh)
//   d att_src[c] += ds[r][h] xw[r][c];   d att_dst[c] += dd[r][h] xw[r][c]  (per-thread partials, one atomic pair per
//                                                                             thread at the end)
// a row whose ds / dd are both zero for the head adds nothing and is not read.  A thread owns one column.
__global__ __launch_bounds__(1024) void gat_backward_epilogue_kernel(float* __restrict__ dh, const float* __restrict__ ds,
                                                                     const float* __restrict__ dd,
                                                                     const float* __restrict__ xw,
                                                                     const float* __restrict__ att_src,
                                                                     const float* __restrict__ att_dst,
                                                                     const int32_t* __restrict__ n_dev, int H, int C,
                                                                     float* __restrict__ d_att_src,
                                                                     float* __restrict__ d_att_dst) {
  const int HC = H * C, c = threadIdx.x;
  if (c >= HC) return;
  const int h = c / C, n = *n_dev;
  const float as = att_src[c], ad = att_dst[c];
  float acc_s = 0.f, acc_d = 0.f;
  for (int r = blockIdx.x; r < n; r += gridDim.x) {
    const float a = ds[(int64_t)r * H + h], b = dd[(int64_t)r * H + h];
    if (a == 0.f && b == 0.f) continue;
    const float x = xw[(int64_t)r * HC + c];
    dh[(int64_t)r * HC + c] += a * as + b * ad;
    acc_s += a * x;
    acc_d += b * x;
  }
  if (acc_s != 0.f) atomicAdd(d_att_src + c, acc_s);
  if (acc_d != 0.f) atomicAdd(d_att_dst + c, acc_d);
}

// ---- backward of the attention aggregation (training): one wave per destination row, same lane layout as the
// forward.  With alpha_e = softmax_e(z_e), out_i = sum_e alpha_e x_e (e over the in-edges and the added self loop):
//   d alpha_e = <g_i, x_e>,  S = sum_e alpha_e d alpha_e = <g_i, out_i>,  dz_e = alpha_e (d alpha_e - S),
//   dpre_e = dz_e * leaky'(pre_e);  dx_e += alpha_e g_i;  d a_src[j] += dpre_e;  d a_dst[i] += dpre_e;
//   d a_edge[e] = dpre_e (+ dpre_self / cnt: the self loop's logit holds the MEAN a_edge of the row).
// Pass 1 recomputes max and denominator from the scalars, pass 2 reads every source row once.
template <int V>
__global__ __launch_bounds__(256) void gat_backward_kernel(
    const float* __restrict__ h, const float* __restrict__ a_src, const float* __restrict__ a_dst,
    const float* __restrict__ a_edge, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
    const int32_t* __restrict__ col, const int32_t* __restrict__ n_rows_dev, int heads, int C, int group,
    int rows_per_head, float slope, const float* __restrict__ out_pre, const float* __restrict__ dout,
    float* __restrict__ dh, float* __restrict__ d_src, float* __restrict__ d_dst, float* __restrict__ d_edge,
    const float* __restrict__ edge_attr, int De, const float* __restrict__ umsg, float* __restrict__ zout, int wpr) {
  // wpr waves share a destination row (a power of two; 1 with the message term): every wave of the row recomputes its
  // scalars (pass 1 reads one float per edge), wave q takes the in-edges q, q + wpr, ... of pass 2 and wave 0 the self
  // loop — a training batch's last layer has ~10^3 rows of ~25 edges, a wave per row leaves most of the GPU idle
  // message term (EdgeAttrGATConv, umsg != NULL): out_i also holds W_msg z_i with z_i = sum_e alpha_e e_e, so
  // d alpha_e gains <u_i, e_e> with u_i = W_msg^T g_i (umsg [rows][heads][De], dense, from the caller) and the kernel
  // returns z_i (zout, same shape) for d W_msg = sum_i g_i (x) z_i.  Component k of a head lives in lane k % group of
  // the head's lanes, register k / group (as in the forward).
  const int lane = threadIdx.x & 63;
  const int gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int part = gwave & (wpr - 1), wave = gwave / wpr;
  const int waves_total = ((gridDim.x * blockDim.x) >> 6) / wpr;
  const int n_rows = *n_rows_dev, HC = heads * C, chunks = HC >> 2;
  const int kl = lane & (group - 1);
  int hd[V];
  bool on[V], writer[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    on[v] = v * 64 + lane < chunks;
    hd[v] = on[v] ? gat_head_of(v, lane, group, rows_per_head) : 0;
    // one lane per head publishes the per-head scalars
    writer[v] = on[v] && (rows_per_head > 1 ? (lane == 0 && v % rows_per_head == 0) : (lane & (group - 1)) == 0);
  }
  // sum of a per-lane partial over the lanes (and chunk rows) of the lane's head, returned to every lane of the head
  auto head_sum = [&](float (&p)[V]) {
    if (rows_per_head > 1) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < V; ++u)
          if (u / rows_per_head == v / rows_per_head) t += p[u];
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        p[v] = t;
      }
      return;
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
      for (int off = group >> 1; off > 0; off >>= 1) p[v] += __shfl_xor(p[v], off, 64);
  };
  auto dot4 = [](const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; };
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    float ad[V], as_i[V], mx[V], den[V], sum_ae[V], S[V], dd[V];
    float4 g[V], xi[V];
    int cnt = 0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      ad[v] = a_dst[(int64_t)i * heads + hd[v]];
      as_i[v] = a_src[(int64_t)i * heads + hd[v]];
      g[v] = on[v] ? ((const float4*)(dout + (int64_t)i * HC))[v * 64 + lane] : float4{0, 0, 0, 0};
      xi[v] = on[v] ? ((const float4*)(h + (int64_t)i * HC))[v * 64 + lane] : float4{0, 0, 0, 0};
      const float4 o = on[v] ? ((const float4*)(out_pre + (int64_t)i * HC))[v * 64 + lane] : float4{0, 0, 0, 0};
      S[v] = dot4(g[v], o);
      mx[v] = -INFINITY;
      den[v] = 0.f;
      sum_ae[v] = 0.f;
      dd[v] = 0.f;
    }
    {
      // a row whose output gradient is all zero (every row but the roots' at the last layer of a whole-batch-graph
      // forward, every row no root depends on below it) contributes zeros only: skipped, wave-uniformly
      bool nz = false;
#pragma unroll
      for (int v = 0; v < V; ++v) nz |= g[v].x != 0.f || g[v].y != 0.f || g[v].z != 0.f || g[v].w != 0.f;
      if (__ballot(nz) == 0ull) continue;
    }
    head_sum(S);
    // pass 1: max / denominator (every lane of a head computes the same scalars)
    for (int e = 0; e < m; ++e) {
      const int j = col[e0 + e];
      if (j == i) continue;
      ++cnt;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float z = a_src[(int64_t)j * heads + hd[v]] + ad[v];
        if (a_edge) {
          const float ae = a_edge[(int64_t)(e0 + e) * heads + hd[v]];
          z += ae;
          sum_ae[v] += ae;
        }
        z = z > 0.f ? z : slope * z;
        const float nm = fmaxf(mx[v], z);
        den[v] = den[v] * __expf(mx[v] - nm) + __expf(z - nm);
        mx[v] = nm;
      }
    }
    // message term: this lane's components of u_i per chunk row (first chunk row of a head only: no double count)
    float ul[V][GAT_ZR], es[GAT_ZR], zl[V][GAT_ZR];
#pragma unroll
    for (int r = 0; r < GAT_ZR; ++r) es[r] = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int r = 0; r < GAT_ZR; ++r) {
        const int k = r * group + kl;
        const bool mine = umsg && on[v] && k < De && (rows_per_head == 1 || v % rows_per_head == 0);
        ul[v][r] = mine ? umsg[((int64_t)i * heads + hd[v]) * De + k] : 0.f;
        zl[v][r] = 0.f;
      }
    if (umsg)
      for (int e = 0; e < m; ++e) {
        if (col[e0 + e] == i) continue;
#pragma unroll
        for (int r = 0; r < GAT_ZR; ++r) {
          const int k = r * group + kl;
          if (k < De) es[r] += edge_attr[(int64_t)(e0 + e) * De + k];
        }
      }
    float pre_self[V], al_self[V], dpre_self[V], dal[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      pre_self[v] = as_i[v] + ad[v] + (a_edge && cnt > 0 ? sum_ae[v] / (float)cnt : 0.f);
      const float z = pre_self[v] > 0.f ? pre_self[v] : slope * pre_self[v];
      const float nm = fmaxf(mx[v], z);
      den[v] = den[v] * __expf(mx[v] - nm) + __expf(z - nm);
      mx[v] = nm;
      den[v] = 1.0f / (den[v] + 1e-16f);
      al_self[v] = __expf(z - nm) * den[v];
      dal[v] = dot4(g[v], xi[v]);
      if (umsg && cnt > 0) {
#pragma unroll
        for (int r = 0; r < GAT_ZR; ++r) {
          dal[v] += ul[v][r] * es[r] / (float)cnt;            // <u_i, mean e>
          zl[v][r] = al_self[v] * es[r] / (float)cnt;          // the self loop's share of z_i
        }
      }
    }
    head_sum(dal);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      dpre_self[v] = al_self[v] * (dal[v] - S[v]) * (pre_self[v] > 0.f ? 1.f : slope);
      if (part != 0) continue;  // (the self loop's contributions: the row's first wave)
      dd[v] += dpre_self[v];
      if (on[v]) {
        float* o = dh + (int64_t)i * HC + 4 * (v * 64 + lane);
        atomicAdd(o + 0, al_self[v] * g[v].x);
        atomicAdd(o + 1, al_self[v] * g[v].y);
        atomicAdd(o + 2, al_self[v] * g[v].z);
        atomicAdd(o + 3, al_self[v] * g[v].w);
      }
      if (writer[v]) atomicAdd(d_src + (int64_t)i * heads + hd[v], dpre_self[v]);
    }
    // pass 2: the in-edges (this wave's share)
    for (int e = part; e < m; e += wpr) {
      const int j = col[e0 + e];
      if (j == i) continue;
      float4 x[V];
      float da[V], pre[V], ev[GAT_ZR];
      if (umsg) {
#pragma unroll
        for (int r = 0; r < GAT_ZR; ++r) {
          const int k = r * group + kl;
          ev[r] = k < De ? edge_attr[(int64_t)(e0 + e) * De + k] : 0.f;
        }
      }
#pragma unroll
      for (int v = 0; v < V; ++v) {
        x[v] = on[v] ? ((const float4*)(h + (int64_t)j * HC))[v * 64 + lane] : float4{0, 0, 0, 0};
        da[v] = dot4(g[v], x[v]);
        if (umsg) {
#pragma unroll
          for (int r = 0; r < GAT_ZR; ++r) da[v] += ul[v][r] * ev[r];
        }
        pre[v] = a_src[(int64_t)j * heads + hd[v]] + ad[v] + (a_edge ? a_edge[(int64_t)(e0 + e) * heads + hd[v]] : 0.f);
      }
      head_sum(da);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float z = pre[v] > 0.f ? pre[v] : slope * pre[v];
        const float al = __expf(z - mx[v]) * den[v];
        const float dpre = al * (da[v] - S[v]) * (pre[v] > 0.f ? 1.f : slope);
        dd[v] += dpre;
        if (umsg) {
#pragma unroll
          for (int r = 0; r < GAT_ZR; ++r) zl[v][r] += al * ev[r];
        }
        if (on[v]) {
          float* o = dh + (int64_t)j * HC + 4 * (v * 64 + lane);
          atomicAdd(o + 0, al * g[v].x);
          atomicAdd(o + 1, al * g[v].y);
          atomicAdd(o + 2, al * g[v].z);
          atomicAdd(o + 3, al * g[v].w);
        }
        if (writer[v]) {
          atomicAdd(d_src + (int64_t)j * heads + hd[v], dpre);
          if (d_edge) d_edge[(int64_t)(e0 + e) * heads + hd[v]] = dpre + (cnt > 0 ? dpre_self[v] / (float)cnt : 0.f);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      if (writer[v]) atomicAdd(d_dst + (int64_t)i * heads + hd[v], dd[v]);
      if (zout && on[v] && (rows_per_head == 1 || v % rows_per_head == 0)) {
#pragma unroll
        for (int r = 0; r < GAT_ZR; ++r) {
          const int k = r * group + kl;
          if (k < De) zout[((int64_t)i * heads + hd[v]) * De + k] = zl[v][r];
        }
      }
    }
  }
}

// Rows with many in-edges (a hub sampled by many parents keeps one edge set per parent: thousands of edges in one
// row) would serialise on one wave: here the 16 waves of a workgroup take contiguous slices of the row, each builds its
// online-softmax state (max, denominator, weighted sum), and wave 0 merges the states — they combine like the edges
// themselves: rescale both to the common max — then folds in the self loop and writes the row.
template <int V>
__global__ __launch_bounds__(1024) void gat_gather_heavy_kernel(
    const float* __restrict__ h, const float* __restrict__ a_src, const float* __restrict__ a_dst,
    const float* __restrict__ a_edge, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowend,
    const int32_t* __restrict__ col, const int32_t* __restrict__ heavy_count, const int32_t* __restrict__ heavy_list,
    int heads, int C, int group, int rows_per_head, float slope, const float* __restrict__ bias, int act,
    float* __restrict__ out) {
  extern __shared__ float s_part[];  // [GAT_HEAVY_WAVES][V][7][64]: acc.x .y .z .w, max, denominator, sum of a_edge
  __shared__ int s_cnt[GAT_HEAVY_WAVES];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int HC = heads * C, chunks = HC >> 2;
  int hd[V];
  bool on[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    on[v] = v * 64 + lane < chunks;
    hd[v] = on[v] ? gat_head_of(v, lane, group, rows_per_head) : 0;
  }
  const int n_heavy = *heavy_count;
  for (int idx = blockIdx.x; idx < n_heavy; idx += gridDim.x) {
    const int i = heavy_list[idx];
    const int e0 = rowptr[i], m = rowend[i] - e0;
    const int per = (((m + GAT_HEAVY_WAVES - 1) / GAT_HEAVY_WAVES) + 3) & ~3;
    const int lo = min(w * per, m), hi = min(lo + per, m);
    float ad[V], mx[V], den[V], sum_ae[V];
    float4 acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      ad[v] = a_dst[(int64_t)i * heads + hd[v]];
      mx[v] = -INFINITY;
      den[v] = 0.f;
      sum_ae[v] = 0.f;
      acc[v] = float4{0, 0, 0, 0};
    }
    int cnt = 0;
    constexpr int U = 4;
    for (int e = lo; e < hi; e += U) {
      int j[U];
      float4 x[U][V];
      float lg[U][V];
#pragma unroll
      for (int t = 0; t < U; ++t) {
        j[t] = e + t < hi ? col[e0 + e + t] : -1;
        if (j[t] == i) j[t] = -1;
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        if (j[t] < 0) continue;
        const float4* row = (const float4*)(h + (int64_t)j[t] * HC);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          x[t][v] = on[v] ? row[v * 64 + lane] : float4{0, 0, 0, 0};
          lg[t][v] = a_src[(int64_t)j[t] * heads + hd[v]] + ad[v];
          if (a_edge) {
            const float ae = a_edge[(int64_t)(e0 + e + t) * heads + hd[v]];
            lg[t][v] += ae;
            sum_ae[v] += ae;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        if (j[t] < 0) continue;
        ++cnt;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float z = lg[t][v];
          z = z > 0.f ? z : slope * z;
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
    if (w > 0) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float* p = s_part + ((w * V + v) * 7) * 64 + lane;
        p[0] = acc[v].x;
        p[64] = acc[v].y;
        p[128] = acc[v].z;
        p[192] = acc[v].w;
        p[256] = mx[v];
        p[320] = den[v];
        p[384] = sum_ae[v];
      }
      if (lane == 0) s_cnt[w] = cnt;
    }
    __syncthreads();
    if (w == 0) {
      for (int pw_ = 1; pw_ < GAT_HEAVY_WAVES; ++pw_) {
        cnt += s_cnt[pw_];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const float* p = s_part + ((pw_ * V + v) * 7) * 64 + lane;
          const float pm = p[256];
          if (pm == -INFINITY) continue;  // an empty slice
          const float nm = fmaxf(mx[v], pm);
          const float s1 = __expf(mx[v] - nm), s2 = __expf(pm - nm);
          den[v] = den[v] * s1 + p[320] * s2;
          acc[v].x = acc[v].x * s1 + p[0] * s2;
          acc[v].y = acc[v].y * s1 + p[64] * s2;
          acc[v].z = acc[v].z * s1 + p[128] * s2;
          acc[v].w = acc[v].w * s1 + p[192] * s2;
          sum_ae[v] += p[384];
          mx[v] = nm;
        }
      }
      const float4* self = (const float4*)(h + (int64_t)i * HC);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (!on[v]) continue;
        float z = a_src[(int64_t)i * heads + hd[v]] + ad[v] + (a_edge && cnt > 0 ? sum_ae[v] / (float)cnt : 0.f);
        z = z > 0.f ? z : slope * z;
        const float nm = fmaxf(mx[v], z);
        const float sc = __expf(mx[v] - nm), pw = __expf(z - nm);
        const float4 xs = self[v * 64 + lane];
        const float inv = 1.0f / (den[v] * sc + pw + 1e-16f);
        float4 o;
        o.x = (acc[v].x * sc + pw * xs.x) * inv;
        o.y = (acc[v].y * sc + pw * xs.y) * inv;
        o.z = (acc[v].z * sc + pw * xs.z) * inv;
        o.w = (acc[v].w * sc + pw * xs.w) * inv;
        const int q = v * 64 + lane;
        if (bias) {
          const float4 b = ((const float4*)bias)[q];
          o.x += b.x;
          o.y += b.y;
          o.z += b.z;
          o.w += b.w;
        }
        if (act == 1) {
          o.x = fmaxf(o.x, 0.f);
          o.y = fmaxf(o.y, 0.f);
          o.z = fmaxf(o.z, 0.f);
          o.w = fmaxf(o.w, 0.f);
        }
        ((float4*)(out + (int64_t)i * HC))[q] = o;
      }
    }
    __syncthreads();
  }
}

__global__ void gat_transpose_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ wt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < rows * cols) wt[(int64_t)(t % cols) * rows + t / cols] = w[t];
}

// launches the fast pair when the shape allows it (concatenated heads, or a single head); false = use the generic path
static bool launch_gat_fast(gigl_ctx* ctx, const float* h, const float* att_src, const float* att_dst, int heads, int C,
                            float slope, int concat, const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                            const int32_t* n_nodes_dev, int64_t nodes_cap, const int32_t* n_rows_dev, int64_t rows_cap,
                            const float* bias, int act, float* a_src, float* a_dst, const float* a_edge, float* out,
                            const float* edge_attr = nullptr, int De = 0, const float* w_msg = nullptr) {
  GatShape g;
  if (!(concat || heads == 1) || !gat_fast_shape(heads, C, g)) return false;
  float* wt = nullptr;
  int32_t *heavy_count = nullptr, *heavy_list = nullptr;
  if (w_msg) {  // the z registers hold at most GAT_ZR * group components; the shuffles need whole chunk rows
    if (De > GAT_ZR * g.group || (heads * C / 4) % 64) return false;
  }
  if (gigl_arena_reset(ctx, (int64_t)heads * C * De * 4 + rows_cap * 4 + 1024) != GIGL_OK) return false;
  if (!w_msg) {
    heavy_count = (int32_t*)gigl_arena_alloc(ctx, 256);
    heavy_list = (int32_t*)gigl_arena_alloc(ctx, rows_cap * 4);
    if (!heavy_count || !heavy_list) return false;
    gigl_fill_u32(ctx->stream, heavy_count, 0u, 1);
  }
  if (w_msg) {
    wt = (float*)gigl_arena_alloc(ctx, (int64_t)heads * C * De * 4);
    if (!wt) return false;
    hipLaunchKernelGGL(gat_transpose_kernel, dim3((unsigned)((heads * C * De + 255) / 256)), dim3(256), 0, ctx->stream,
                       w_msg, heads * C, De, wt);
  }
  if (((uintptr_t)h | (uintptr_t)out | (uintptr_t)att_src | (uintptr_t)att_dst | (uintptr_t)bias) & 15) return false;
  int64_t ablocks = (nodes_cap + 3) / 4, gblocks = (rows_cap + 3) / 4;
  if (ablocks > 256 * 32) ablocks = 256 * 32;
  if (gblocks > 256 * 32) gblocks = 256 * 32;
#define GAT_FAST(VV)                                                                                                  \
  hipLaunchKernelGGL((gat_alpha_fast_kernel<VV>), dim3((unsigned)ablocks), dim3(256), 0, ctx->stream, h, att_src,    \
                     att_dst, n_nodes_dev, heads, C, g.group, g.rows_per_head, a_src, a_dst);                         \
  if (wt)                                                                                                            \
    hipLaunchKernelGGL((gat_gather_fast_kernel<VV, true>), dim3((unsigned)gblocks), dim3(256), 0, ctx->stream, h,    \
                       a_src, a_dst, a_edge, rowptr, rowend, col, n_rows_dev, heads, C, g.group, g.rows_per_head,    \
                       slope, bias, act, edge_attr, De, wt, out, heavy_count, heavy_list);                           \
  else {                                                                                                             \
    hipLaunchKernelGGL((gat_gather_fast_kernel<VV, false>), dim3((unsigned)gblocks), dim3(256), 0, ctx->stream, h,   \
                       a_src, a_dst, a_edge, rowptr, rowend, col, n_rows_dev, heads, C, g.group, g.rows_per_head,    \
                       slope, bias, act, edge_attr, De, wt, out, heavy_count, heavy_list);                           \
    const size_t lds = (size_t)GAT_HEAVY_WAVES * VV * 7 * 64 * 4;                                                                  \
    if (lds > 48 * 1024)                                                                                             \
      hipFuncSetAttribute((const void*)gat_gather_heavy_kernel<VV>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                          (int)lds);                                                                                 \
    hipLaunchKernelGGL((gat_gather_heavy_kernel<VV>), dim3(1024), dim3(1024), lds, ctx->stream, h, a_src, a_dst,      \
                       a_edge, rowptr, rowend, col, heavy_count, heavy_list, heads, C, g.group, g.rows_per_head,     \
                       slope, bias, act, out);                                                                       \
  }
  if (g.V == 1) {
    GAT_FAST(1);
  } else if (g.V == 2) {
    GAT_FAST(2);
  } else {
    GAT_FAST(4);
  }
#undef GAT_FAST
  return true;
}

template <typename T>
int32_t launch_gather(gigl_ctx* ctx, const T* src, int d, const uint32_t* gather_ids,
                      const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                      const int32_t* n_rows_dev, int64_t rows_cap, float* out, int op = GIGL_AGGR_MEAN,
                      const int32_t* n_local_dev = nullptr, int tiled_nkc = 0, const int32_t* global_map = nullptr,
                      const T* src2 = nullptr, const T* src3 = nullptr, int no_self = 0, int own_world = 0,
                      int own_rank = 0, const T* const* peers = nullptr) {
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  dim3 g((unsigned)blocks), b(256);
  hipStream_t st = ctx->stream;
  const int vecs = d / 4;
#define GLO(LPR, VPL, OP)                                                                                              \
  do {                                                                                                                 \
    if (peers)                                                                                                         \
      hipLaunchKernelGGL((gather_mean_kernel<T, LPR, VPL, OP, false, true>), g, b, 0, st, src, d, gather_ids, rowptr,  \
                         rowend, col, n_rows_dev, out, n_local_dev, tiled_nkc, global_map, src2, src3,                 \
                         (const float*)nullptr, (const float*)nullptr, 0, 0, no_self, (const int32_t*)nullptr, 0,      \
                         own_world, own_rank, peers);                                                                  \
    else                                                                                                               \
      hipLaunchKernelGGL((gather_mean_kernel<T, LPR, VPL, OP>), g, b, 0, st, src, d, gather_ids, rowptr, rowend, col,  \
                         n_rows_dev, out, n_local_dev, tiled_nkc, global_map, src2, src3, (const float*)nullptr,       \
                         (const float*)nullptr, 0, 0, no_self, (const int32_t*)nullptr, 0, own_world, own_rank,        \
                         (const T* const*)nullptr);                                                                    \
  } while (0)
#define GL(LPR, VPL)                                        \
  do {                                                      \
    if (op == GIGL_AGGR_MEAN) GLO(LPR, VPL, GIGL_AGGR_MEAN); \
    else if (op == GIGL_AGGR_SUM) GLO(LPR, VPL, GIGL_AGGR_SUM); \
    else GLO(LPR, VPL, GIGL_AGGR_MAX);                      \
  } while (0)
  if (peers && (own_world < 1 || own_world > 64))
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "peer-mapped tables: world %d outside [1, 64]", own_world);
  if ((d & 3) != 0 || vecs > 512) {
    if (tiled_nkc || global_map || peers)
      return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "the tiled operand layout / global row map need d %% 4 == 0 and d <= 2048");
    hipLaunchKernelGGL((gather_mean_generic_kernel<T>), g, b, 0, st, src, d, gather_ids, rowptr, rowend,
                       col, n_rows_dev, out, op, n_local_dev);
  } else if (vecs <= 8) GL(8, 1);
  else if (vecs <= 16) GL(16, 1);
  else if (vecs <= 32) GL(32, 1);
  else if (vecs <= 64) GL(64, 1);
  else if (vecs <= 128) GL(64, 2);
  else if (vecs <= 256) GL(64, 4);
  else GL(64, 8);
#undef GL
#undef GLO
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// the projected-input first layer: gather + mean over fp32 rows of W_l x, + W_r x_self + bias, activation
int32_t launch_gather_projected(gigl_ctx* ctx, const float* src_l, const float* src_r, int ld, int d, const uint32_t* gather_ids,
                                const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                const int32_t* n_rows_dev, int64_t rows_cap, const int32_t* n_local_dev, int op,
                                const float* bias, int act, float* out, const int32_t* global_map, const float* src2,
                                const float* src3, int ld3, const int32_t* self_ids, int own_world, int own_rank,
                                const float* const* peers = nullptr) {
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  dim3 g((unsigned)blocks), b(256);
  const int vecs = d / 4;
#define GLPP(LPR, VPL, OP)                                                                                            \
  hipLaunchKernelGGL((gather_mean_kernel<float, LPR, VPL, OP, true, true>), g, b, 0, ctx->stream, src_l, d, gather_ids, \
                     rowptr, rowend, col, n_rows_dev, out, n_local_dev, 0, global_map, src2, src3, src_r, bias, act, ld, \
                     0, self_ids, ld3, own_world, own_rank, peers)
#define GLP(LPR, VPL)                                                                                                 \
  do {                                                                                                                \
    if (peers && op == GIGL_AGGR_MEAN)                                                                                \
      GLPP(LPR, VPL, GIGL_AGGR_MEAN);                                                                                 \
    else if (peers)                                                                                                   \
      GLPP(LPR, VPL, GIGL_AGGR_SUM);                                                                                  \
    else if (op == GIGL_AGGR_MEAN)                                                                                    \
      hipLaunchKernelGGL((gather_mean_kernel<float, LPR, VPL, GIGL_AGGR_MEAN, true>), g, b, 0, ctx->stream, src_l, d, \
                         gather_ids, rowptr, rowend, col, n_rows_dev, out, n_local_dev, 0, global_map, src2, src3,    \
                         src_r, bias, act, ld, 0, self_ids, ld3, own_world, own_rank, (const float* const*)nullptr);  \
    else                                                                                                              \
      hipLaunchKernelGGL((gather_mean_kernel<float, LPR, VPL, GIGL_AGGR_SUM, true>), g, b, 0, ctx->stream, src_l, d,  \
                         gather_ids, rowptr, rowend, col, n_rows_dev, out, n_local_dev, 0, global_map, src2, src3,    \
                         src_r, bias, act, ld, 0, self_ids, ld3, own_world, own_rank, (const float* const*)nullptr);  \
  } while (0)
  if (peers && (own_world < 1 || own_world > 64))
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "peer-mapped tables: world %d outside [1, 64]", own_world);
  if ((d & 3) != 0 || vecs > 512) return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "projected input: width %d (need d %% 4 == 0, d <= 2048)", d);
  if (vecs <= 8) GLP(8, 1);
  else if (vecs <= 16) GLP(16, 1);
  else if (vecs <= 32) GLP(32, 1);
  else if (vecs <= 64) GLP(64, 1);
  else if (vecs <= 128) GLP(64, 2);
  else if (vecs <= 256) GLP(64, 4);
  else GLP(64, 8);
#undef GLP
#undef GLPP
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

__global__ __launch_bounds__(256) void half_rows_to_f32_kernel(const __half* __restrict__ src, int64_t n,
                                                               float* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const __half2* q = reinterpret_cast<const __half2*>(src + i);
    const float2 a = __half22float2(q[0]), b = __half22float2(q[1]);
    *reinterpret_cast<float4_t*>(out + i) = float4_t{a.x, a.y, b.x, b.y};
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = __half2float(src[j]);
  }
}

}  // namespace

int32_t gigl_gather_project_mixed(gigl_ctx* ctx, const float* src_l, const float* src_r, int32_t ld, int32_t d,
                                  const uint32_t* gather_ids, const int32_t* rowptr, const int32_t* rowend,
                                  const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, int32_t aggr,
                                  const int32_t* n_local_rows_dev, const float* bias, int32_t act, float* out,
                                  const int32_t* global_map, const float* src2, const float* src3, int32_t ld3,
                                  const int32_t* self_ids, int32_t own_world, int32_t own_rank,
                                  const float* const* peers) {
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  if (aggr != GIGL_AGGR_MEAN && aggr != GIGL_AGGR_SUM)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "projected input needs a linear reduction (mean / sum)");
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  return launch_gather_projected(ctx, src_l, src_r, ld, d, gather_ids, rowptr, rowend, col, n_rows_dev, rows_cap,
                                 n_local_rows_dev, aggr, bias, act, out, global_map, src2, src3, ld3, self_ids, own_world,
                                 own_rank, peers);
}

extern "C" {

// [X W_l^T | X W_r^T] over the whole resident feature table (the table of gigl_sage_plan_set_projected_input): ONE
// product per row chunk against the stacked weight [W_l ; W_r] — the feature rows are read (and, for an fp16 table,
// widened) once
int32_t gigl_sage_project_features(gigl_ctx* ctx, gigl_feat* feat, const float* w_fused, int32_t n_out, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, feat && w_fused && out && n_out > 0, "null argument");
  GIGL_REQUIRE(ctx, feat->dtype == GIGL_DTYPE_F32 || feat->dtype == GIGL_DTYPE_F16, "bad feature dtype");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int d = feat->d;
  const int64_t n = feat->n;
  hipStream_t st = ctx->stream;
  const int64_t chunk = (int64_t)1 << 19;
  float *wcat = nullptr, *stage = nullptr;
  int32_t* cnt = nullptr;
  auto cleanup = [&]() {
    hipStreamSynchronize(st);
    hipFree(wcat); hipFree(stage); hipFree(cnt);
  };
  GIGL_HIP_CHECK(ctx, hipMalloc((void**)&wcat, (size_t)2 * n_out * d * 4));
  // two staging buffers for an fp16 table: chunk c + 1 is widened while chunk c is multiplied? (one stream: they run
  // back to back; two buffers only keep the conversion of c + 1 from overwriting the operand of c's product)
  const int64_t cm = n < chunk ? n : chunk;
  // fp16 rows go through the projection as stored when K % 4 == 0 (GIGL_PROJECT_WIDEN=1, or GIGL_LINEAR_EXACT: the
  // widened copy + the fp32 path, for comparison)
  const bool half_direct = feat->dtype == GIGL_DTYPE_F16 && (d & 3) == 0 && !getenv("GIGL_PROJECT_WIDEN") &&
                           !getenv("GIGL_LINEAR_EXACT");
  // (cnt: [0..1] row counts, [2..4] hs scales, [6..7] the scale kernel's running maximum + ticket: zero to begin with)
  if (hipMalloc((void**)&cnt, 64) != hipSuccess || hipMemsetAsync(cnt, 0, 64, st) != hipSuccess ||
      (feat->dtype == GIGL_DTYPE_F16 && !half_direct && hipMalloc((void**)&stage, (size_t)cm * d * 4 + 16) != hipSuccess)) {
    cleanup();
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of the projection workspace failed");
  }
  // [W_l | W_r] rows of 2d floats -> the stacked [2 n_out][d] matrix
  hipMemcpy2DAsync(wcat, (size_t)d * 4, w_fused, (size_t)2 * d * 4, (size_t)d * 4, n_out, hipMemcpyDeviceToDevice, st);
  hipMemcpy2DAsync(wcat + (size_t)n_out * d, (size_t)d * 4, w_fused + d, (size_t)2 * d * 4, (size_t)d * 4, n_out,
                   hipMemcpyDeviceToDevice, st);
  const int32_t counts[2] = {(int32_t)cm, (int32_t)(n % cm)};  // row counts: a whole chunk, the tail
  if (hipMemcpyAsync(cnt, counts, 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    cleanup();
    return gigl_fail(ctx, GIGL_E_HIP, "row-count upload failed");
  }
  int32_t rc = GIGL_OK;
  // fp16 rows ARE the h1 plane of the half split; the weights join as two fp16 planes when they fit the half range:
  // two products per accumulator instead of five
  // (the halves are taken as stored, s_a = 1; the weights are scaled into the top of the half range on the device)
  const bool hs = half_direct && gigl_half_split_enabled();
  float* hs_dev = reinterpret_cast<float*>(cnt + 2);
  if (hs) {
    rc = gigl_hs_scale_update(ctx, wcat, (int64_t)2 * n_out * d, 1.f, hs_dev);
    if (rc != GIGL_OK) {
      cleanup();
      return rc;
    }
  }
  for (int64_t r0 = 0; r0 < n && rc == GIGL_OK; r0 += cm) {
    const int64_t m = (n - r0) < cm ? (n - r0) : cm;
    const float* a;
    if (feat->dtype == GIGL_DTYPE_F16 && half_direct) {  // the rows as stored: two bf16 planes, five products
      gigl_prof_scope ps(ctx, GIGL_K_LINEAR);
      const int nn = 2 * n_out;
      const int64_t bm = (m + 127) / 128;
      const float* ah = reinterpret_cast<const float*>((const __half*)feat->rows + r0 * d);
      if (hs && nn > 64)
        hipLaunchKernelGGL((linear_split_kernel<2, true, false, true, true>), dim3((unsigned)(bm * ((nn + 127) / 128))), dim3(256),
                           0, st, ah, (const float*)wcat, (const float*)nullptr, (const int32_t*)(cnt + (m == cm ? 0 : 1)), d,
                           nn, 0, out + r0 * 2 * n_out, 0, 0, (int64_t)0, (int64_t)0,
                           (const float*)nullptr, (const uint32_t*)nullptr, 0, 0, (const float*)hs_dev);
      else if (hs)
        hipLaunchKernelGGL((linear_split_kernel<1, true, false, true, true>), dim3((unsigned)(bm * ((nn + 63) / 64))), dim3(256),
                           0, st, ah, (const float*)wcat, (const float*)nullptr, (const int32_t*)(cnt + (m == cm ? 0 : 1)), d,
                           nn, 0, out + r0 * 2 * n_out, 0, 0, (int64_t)0, (int64_t)0,
                           (const float*)nullptr, (const uint32_t*)nullptr, 0, 0, (const float*)hs_dev);
      else if (nn > 64)
        hipLaunchKernelGGL((linear_split_kernel<2, true, false, true>), dim3((unsigned)(bm * ((nn + 127) / 128))), dim3(256),
                           0, st, ah, (const float*)wcat, (const float*)nullptr, (const int32_t*)(cnt + (m == cm ? 0 : 1)), d,
                           nn, 0, out + r0 * 2 * n_out, 0, 0, (int64_t)0, (int64_t)0);
      else
        hipLaunchKernelGGL((linear_split_kernel<1, true, false, true>), dim3((unsigned)(bm * ((nn + 63) / 64))), dim3(256),
                           0, st, ah, (const float*)wcat, (const float*)nullptr, (const int32_t*)(cnt + (m == cm ? 0 : 1)), d,
                           nn, 0, out + r0 * 2 * n_out, 0, 0, (int64_t)0, (int64_t)0);
      if (hipGetLastError() != hipSuccess) rc = gigl_fail(ctx, GIGL_E_HIP, "projection launch failed");
      continue;
    }
    if (feat->dtype == GIGL_DTYPE_F16) {
      const int64_t elems = m * d;
      hipLaunchKernelGGL(half_rows_to_f32_kernel, dim3((unsigned)((elems / 4 + 256) / 256)), dim3(256), 0, st,
                         (const __half*)feat->rows + r0 * d, elems, stage);
      a = stage;
    } else {
      a = (const float*)feat->rows + r0 * d;
    }
    rc = gigl_linear(ctx, a, wcat, nullptr, cnt + (m == cm ? 0 : 1), m, d, 2 * n_out, 0, out + r0 * 2 * n_out);
  }
  cleanup();
  return rc;
}

int32_t gigl_gather_mean(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d,
                         const uint32_t* gather_ids, const int32_t* rowptr, const int32_t* rowend,
                         const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, src && rowptr && rowend && col && n_rows_dev && out, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && rows_cap >= 0, "bad sizes");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  if (src_dtype == GIGL_DTYPE_F32)
    return launch_gather<float>(ctx, (const float*)src, d, gather_ids, rowptr, rowend, col, n_rows_dev,
                                rows_cap, out);
  if (src_dtype == GIGL_DTYPE_F16)
    return launch_gather<__half>(ctx, (const __half*)src, d, gather_ids, rowptr, rowend, col, n_rows_dev,
                                 rows_cap, out);
  return gigl_fail(ctx, GIGL_E_INVALID_ARG, "bad dtype %d", src_dtype);
}

int32_t gigl_gather_reduce(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* gather_ids,
                           const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                           const int32_t* n_rows_dev, int64_t rows_cap, int32_t aggr, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, src && rowptr && rowend && col && n_rows_dev && out, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && rows_cap >= 0, "bad sizes");
  GIGL_REQUIRE(ctx, aggr == GIGL_AGGR_MEAN || aggr == GIGL_AGGR_SUM || aggr == GIGL_AGGR_MAX, "bad aggr %d", aggr);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  if (src_dtype == GIGL_DTYPE_F32)
    return launch_gather<float>(ctx, (const float*)src, d, gather_ids, rowptr, rowend, col, n_rows_dev, rows_cap, out,
                                aggr);
  if (src_dtype == GIGL_DTYPE_F16)
    return launch_gather<__half>(ctx, (const __half*)src, d, gather_ids, rowptr, rowend, col, n_rows_dev, rows_cap,
                                 out, aggr);
  return gigl_fail(ctx, GIGL_E_INVALID_ARG, "bad dtype %d", src_dtype);
}

}  // extern "C"

int32_t gigl_gather_reduce_mixed(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* gather_ids,
                                 const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                 const int32_t* n_rows_dev, int64_t rows_cap, int32_t aggr,
                                 const int32_t* n_local_rows_dev, float* out, int32_t tiled_nkc,
                                 const int32_t* global_map, const void* src2, const void* src3, int32_t no_self,
                                 int32_t own_world, int32_t own_rank, const void* const* peers) {
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  if (src_dtype == GIGL_DTYPE_F32)
    return launch_gather<float>(ctx, (const float*)src, d, gather_ids, rowptr, rowend, col, n_rows_dev, rows_cap, out,
                                aggr, n_local_rows_dev, tiled_nkc, global_map, (const float*)src2, (const float*)src3,
                                no_self, own_world, own_rank, (const float* const*)peers);
  return launch_gather<__half>(ctx, (const __half*)src, d, gather_ids, rowptr, rowend, col, n_rows_dev, rows_cap, out,
                               aggr, n_local_rows_dev, tiled_nkc, global_map, (const __half*)src2, (const __half*)src3,
                               no_self, own_world, own_rank, (const __half* const*)peers);
}

extern "C" {

int32_t gigl_gather_rows(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* ids,
                         const int32_t* n_dev, int64_t cap, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, src && ids && n_dev && out && d > 0 && cap >= 0, "bad arguments");
  GIGL_REQUIRE(ctx, src_dtype == GIGL_DTYPE_F32 || src_dtype == GIGL_DTYPE_F16, "bad dtype %d", src_dtype);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (cap == 0) return GIGL_OK;
  int64_t blocks = (cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (src_dtype == GIGL_DTYPE_F32)
    hipLaunchKernelGGL((gather_rows_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)src,
                       d, ids, n_dev, out);
  else
    hipLaunchKernelGGL((gather_rows_kernel<__half>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       (const __half*)src, d, ids, n_dev, out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gcn_aggregate(gigl_ctx* ctx, const void* h, int32_t h_dtype, int32_t d, const uint32_t* gather_ids,
                           const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                           const int32_t* n_nodes_dev, int64_t nodes_cap, const int32_t* n_rows_dev,
                           int64_t rows_cap, const float* bias, int32_t act, float* dinv_scratch, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, h && rowptr && rowend && col && n_nodes_dev && n_rows_dev && dinv_scratch && out, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && rows_cap >= 0 && nodes_cap >= rows_cap, "bad sizes");
  GIGL_REQUIRE(ctx, h_dtype == GIGL_DTYPE_F32 || h_dtype == GIGL_DTYPE_F16, "bad dtype %d", h_dtype);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  hipLaunchKernelGGL(gcn_dinv_kernel, dim3((unsigned)((nodes_cap + 255) / 256)), dim3(256), 0, ctx->stream, rowptr,
                     rowend, col, n_nodes_dev, dinv_scratch);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (h_dtype == GIGL_DTYPE_F32)
    hipLaunchKernelGGL((gcn_gather_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float*)h,
                       d, gather_ids, dinv_scratch, rowptr, rowend, col, n_rows_dev, bias, act, out);
  else
    hipLaunchKernelGGL((gcn_gather_kernel<__half>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       (const __half*)h, d, gather_ids, dinv_scratch, rowptr, rowend, col, n_rows_dev, bias, act, out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gat_aggregate(gigl_ctx* ctx, const float* h, const float* att_src, const float* att_dst, int32_t heads,
                           int32_t channels, float negative_slope, int32_t concat, const int32_t* rowptr,
                           const int32_t* rowend, const int32_t* col, const int32_t* n_nodes_dev, int64_t nodes_cap,
                           const int32_t* n_rows_dev, int64_t rows_cap, const float* bias, int32_t act,
                           float* alpha_scratch, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, h && att_src && att_dst && rowptr && rowend && col && n_nodes_dev && n_rows_dev &&
                        alpha_scratch && out, "null argument");
  GIGL_REQUIRE(ctx, heads > 0 && channels > 0 && rows_cap >= 0 && nodes_cap >= rows_cap, "bad sizes");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  float* a_src = alpha_scratch;
  float* a_dst = alpha_scratch + nodes_cap * heads;
  if (launch_gat_fast(ctx, h, att_src, att_dst, heads, channels, negative_slope, concat, rowptr, rowend, col,
                      n_nodes_dev, nodes_cap, n_rows_dev, rows_cap, bias, act, a_src, a_dst, nullptr, out)) {
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    return GIGL_OK;
  }
  hipLaunchKernelGGL(gat_alpha_kernel, dim3((unsigned)((nodes_cap * heads + 255) / 256)), dim3(256), 0, ctx->stream,
                     h, att_src, att_dst, n_nodes_dev, heads, channels, a_src, a_dst);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gat_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, h, a_src, a_dst, rowptr,
                     rowend, col, n_rows_dev, heads, channels, negative_slope, concat, bias, act, out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gat_aggregate_edge(gigl_ctx* ctx, const float* h, const float* att_src, const float* att_dst,
                                int32_t heads, int32_t channels, float negative_slope, int32_t concat,
                                const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                const int32_t* n_nodes_dev, int64_t nodes_cap, const int32_t* n_rows_dev,
                                int64_t rows_cap, const float* bias, int32_t act, const float* edge_attr,
                                int32_t edge_dim, int64_t cap_edges, const float* att_edge_folded,
                                const float* w_edge_msg, float* alpha_scratch, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, h && att_src && att_dst && rowptr && rowend && col && n_nodes_dev && n_rows_dev &&
                        alpha_scratch && out && edge_attr && att_edge_folded, "null argument");
  GIGL_REQUIRE(ctx, heads > 0 && channels > 0 && rows_cap >= 0 && nodes_cap >= rows_cap && cap_edges >= 0, "bad sizes");
  GIGL_REQUIRE(ctx, edge_dim > 0 && edge_dim <= GAT_MAX_EDGE_DIM, "edge_dim %d outside [1,%d]", edge_dim,
               GAT_MAX_EDGE_DIM);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  float* a_src = alpha_scratch;
  float* a_dst = alpha_scratch + nodes_cap * heads;
  float* a_edge = alpha_scratch + 2 * nodes_cap * heads;
  if (cap_edges > 0)
    hipLaunchKernelGGL(gat_edge_alpha_kernel, dim3((unsigned)((cap_edges * heads + 255) / 256)), dim3(256), 0,
                       ctx->stream, edge_attr, edge_dim, att_edge_folded, heads, cap_edges, a_edge);
  if (cap_edges > 0 &&
      launch_gat_fast(ctx, h, att_src, att_dst, heads, channels, negative_slope, concat, rowptr, rowend, col,
                      n_nodes_dev, nodes_cap, n_rows_dev, rows_cap, bias, act, a_src, a_dst, a_edge, out, edge_attr,
                      edge_dim, w_edge_msg)) {
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    return GIGL_OK;
  }
  hipLaunchKernelGGL(gat_alpha_kernel, dim3((unsigned)((nodes_cap * heads + 255) / 256)), dim3(256), 0, ctx->stream,
                     h, att_src, att_dst, n_nodes_dev, heads, channels, a_src, a_dst);
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gat_edge_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, h, a_src, a_dst,
                     a_edge, edge_attr, edge_dim, w_edge_msg, rowptr, rowend, col, n_rows_dev, heads, channels,
                     negative_slope, concat, bias, act, out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gat_backward_epilogue(gigl_ctx* ctx, float* dh, const float* ds, const float* dd, const float* xw,
                                   const float* att_src, const float* att_dst, const int32_t* n_nodes_dev, int64_t nodes_cap,
                                   int32_t heads, int32_t channels, float* d_att_src, float* d_att_dst) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, dh && ds && dd && xw && att_src && att_dst && n_nodes_dev && d_att_src && d_att_dst, "null argument");
  GIGL_REQUIRE(ctx, heads > 0 && channels > 0 && nodes_cap >= 0, "bad sizes");
  if ((int64_t)heads * channels > 1024)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "gigl_gat_backward_epilogue: heads*channels = %lld > 1024",
                     (long long)heads * channels);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (nodes_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_BWD);
  const int hc = heads * channels, threads = (hc + 63) / 64 * 64;
  int64_t blocks = nodes_cap < 512 ? nodes_cap : 512;  // (every workgroup ends with 2 H C atomics on the same 2 H C addresses)
  hipLaunchKernelGGL(gat_backward_epilogue_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), 0, ctx->stream, dh, ds, dd,
                     xw, att_src, att_dst, n_nodes_dev, heads, channels, d_att_src, d_att_dst);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gat_aggregate_backward(gigl_ctx* ctx, const float* h, const float* att_src, const float* att_dst,
                                    int32_t heads, int32_t channels, float negative_slope, const int32_t* rowptr,
                                    const int32_t* rowend, const int32_t* col, const int32_t* n_nodes_dev,
                                    int64_t nodes_cap, const int32_t* n_rows_dev, int64_t rows_cap,
                                    const float* out_pre, const float* dout, const float* edge_attr, int32_t edge_dim,
                                    int64_t cap_edges, const float* att_edge_folded, float* alpha_scratch, float* dh,
                                    float* d_alpha_src, float* d_alpha_dst, float* d_alpha_edge,
                                    const float* u_msg, float* z_out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, (u_msg == nullptr) == (z_out == nullptr) && (!u_msg || edge_attr),
               "u_msg and z_out go together and need edge_attr");
  GIGL_REQUIRE(ctx, h && att_src && att_dst && rowptr && rowend && col && n_nodes_dev && n_rows_dev && out_pre &&
                        dout && alpha_scratch && dh && d_alpha_src && d_alpha_dst, "null argument");
  GIGL_REQUIRE(ctx, heads > 0 && channels > 0 && rows_cap >= 0 && nodes_cap >= rows_cap && cap_edges >= 0, "bad sizes");
  GIGL_REQUIRE(ctx, (edge_attr == nullptr) == (att_edge_folded == nullptr) && (!edge_attr || d_alpha_edge),
               "edge_attr, att_edge_folded and d_alpha_edge go together");
  GatShape g;
  if (!gat_fast_shape(heads, channels, g) ||
      (((uintptr_t)h | (uintptr_t)dout | (uintptr_t)out_pre | (uintptr_t)dh | (uintptr_t)att_src | (uintptr_t)att_dst) & 15))
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED,
                     "GAT backward needs channels %% 4 == 0 and channels/4 a power of two <= 64 (or channels %% 256 == 0), "
                     "heads*channels <= 1024, 16-byte aligned matrices; got heads=%d channels=%d", heads, channels);
  if (u_msg && (edge_dim > GAT_ZR * g.group || (heads * channels / 4) % 64))
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "message-term backward: edge_dim %d > %d or heads*channels %% 256 != 0",
                     edge_dim, GAT_ZR * g.group);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_BWD);
  float* a_src = alpha_scratch;
  float* a_dst = alpha_scratch + nodes_cap * heads;
  float* a_edge = edge_attr ? alpha_scratch + 2 * nodes_cap * heads : nullptr;
  if (edge_attr && cap_edges > 0)
    hipLaunchKernelGGL(gat_edge_alpha_kernel, dim3((unsigned)((cap_edges * heads + 255) / 256)), dim3(256), 0,
                       ctx->stream, edge_attr, edge_dim, att_edge_folded, heads, cap_edges, a_edge);
  // (few rows — the roots' layer of a training step — are shared by several waves each; not with the message term, whose
  // z_out rows are written whole by one wave)
  static const bool one_wave = getenv("GIGL_GAT_BWD_ONE_WAVE") != nullptr;  // (A/B knob: a wave per row, as before round 5)
  const int wpr = (u_msg || one_wave) ? 1 : (rows_cap >= 32768 ? 1 : (rows_cap >= 8192 ? 2 : (rows_cap >= 2048 ? 4 : 8)));
  int64_t ablocks = (nodes_cap + 3) / 4, gblocks = (rows_cap * wpr + 3) / 4;
  if (ablocks > 256 * 32) ablocks = 256 * 32;
  if (gblocks > 256 * 32) gblocks = 256 * 32;
  gblocks = (gblocks + 1) & ~(int64_t)1;  // (whole rows per launch: 4 waves per workgroup, wpr <= 8)
#define GAT_BWD(VV)                                                                                                  \
  hipLaunchKernelGGL((gat_alpha_fast_kernel<VV>), dim3((unsigned)ablocks), dim3(256), 0, ctx->stream, h, att_src,   \
                     att_dst, n_nodes_dev, heads, channels, g.group, g.rows_per_head, a_src, a_dst);                \
  hipLaunchKernelGGL((gat_backward_kernel<VV>), dim3((unsigned)gblocks), dim3(256), 0, ctx->stream, h, a_src, a_dst, \
                     a_edge, rowptr, rowend, col, n_rows_dev, heads, channels, g.group, g.rows_per_head,            \
                     negative_slope, out_pre, dout, dh, d_alpha_src, d_alpha_dst, d_alpha_edge, edge_attr, edge_dim,  \
                     u_msg, z_out, wpr)
  if (g.V == 1) {
    GAT_BWD(1);
  } else if (g.V == 2) {
    GAT_BWD(2);
  } else {
    GAT_BWD(4);
  }
#undef GAT_BWD
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gather_mean_backward(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                  const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                  int64_t rows_cap, float* dsrc) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, dout && rowptr && rowend && col && n_rows_dev && dsrc, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && rows_cap >= 0, "bad sizes");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_BWD);
  const int wpr = rows_cap >= 32768 ? 1 : (rows_cap >= 8192 ? 2 : (rows_cap >= 2048 ? 4 : 8));
  int64_t blocks = (rows_cap * wpr + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gather_mean_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dout, d,
                     rowptr, rowend, col, n_rows_dev, dsrc, GIGL_AGGR_MEAN, (const float*)nullptr, wpr);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int64_t gigl_transposed_rows_words(int64_t src_cap, int64_t edges_cap) { return 3 * src_cap + edges_cap + 64; }

int32_t gigl_transposed_rows_build(gigl_ctx* ctx, const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                   const int32_t* n_rows_dev, int64_t rows_cap, const int32_t* n_src_dev, int64_t src_cap,
                                   int64_t edges_cap, int32_t* lists) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, rowptr && rowend && col && n_rows_dev && n_src_dev && lists, "null argument");
  GIGL_REQUIRE(ctx, rows_cap >= 0 && src_cap >= rows_cap && edges_cap >= 0 && src_cap < ((int64_t)1 << 31) &&
                        edges_cap < ((int64_t)1 << 31), "bad sizes");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (src_cap == 0) return GIGL_OK;
  // lists: cnt[src_cap] | fill[src_cap] | cursor + pad (cleared together) | ptr[src_cap] | list[edges_cap]
  int32_t* cnt = lists;
  int32_t* fill = cnt + src_cap;
  int32_t* cursor = cnt + 2 * src_cap;
  int32_t* ptr = cnt + 2 * src_cap + 16;
  int32_t* list = ptr + src_cap;
  hipStream_t st = ctx->stream;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_BWD);
  gigl_fill_u32(st, (uint32_t*)cnt, 0u, 2 * src_cap + 1);
  if (rows_cap > 0) {
    int64_t wg = (rows_cap + 3) / 4;
    if (wg > 256 * 8) wg = 256 * 8;
    hipLaunchKernelGGL(gmt_count_kernel, dim3((unsigned)wg), dim3(256), 0, st, rowptr, rowend, col, n_rows_dev, cnt);
    int64_t pg = (src_cap + 255) / 256;
    if (pg > 256 * 8) pg = 256 * 8;
    hipLaunchKernelGGL(gmt_place_kernel, dim3((unsigned)pg), dim3(256), 0, st, (const int32_t*)cnt, n_src_dev, cursor, ptr);
    hipLaunchKernelGGL(gmt_fill_kernel, dim3((unsigned)wg), dim3(256), 0, st, rowptr, rowend, col, n_rows_dev,
                       (const int32_t*)ptr, fill, list);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gather_mean_backward_lists(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                        const int32_t* rowend, const int32_t* n_rows_dev, const int32_t* n_src_dev,
                                        int64_t src_cap, const int32_t* lists, int32_t aggr, float* dsrc) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, dout && rowptr && rowend && n_rows_dev && n_src_dev && lists && dsrc, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && (d & 3) == 0 && src_cap >= 0 && src_cap < ((int64_t)1 << 31), "bad sizes (d must be a multiple of 4)");
  GIGL_REQUIRE(ctx, aggr == GIGL_AGGR_MEAN || aggr == GIGL_AGGR_SUM, "mean or sum");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (src_cap == 0) return GIGL_OK;
  const int32_t* cnt = lists;
  const int32_t* ptr = lists + 2 * src_cap + 16;
  const int32_t* list = ptr + src_cap;
  hipStream_t st = ctx->stream;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_BWD);
  const int mean = aggr == GIGL_AGGR_MEAN ? 1 : 0;
  const int lpr = d >= 256 ? 64 : (d >= 128 ? 32 : (d >= 64 ? 16 : 8));
  int64_t gg = (src_cap * lpr / 64 + 3) / 4;
  if (gg > 256 * 16) gg = 256 * 16;
  if (gg < 1) gg = 1;
#define GMT(L)                                                                                                          \
  hipLaunchKernelGGL(gmt_gather_kernel<L>, dim3((unsigned)gg), dim3(256), 0, st, dout, d, rowptr, rowend, n_rows_dev,   \
                     n_src_dev, cnt, ptr, list, dsrc, mean)
  if (lpr == 64) GMT(64);
  else if (lpr == 32) GMT(32);
  else if (lpr == 16) GMT(16);
  else GMT(8);
#undef GMT
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gather_mean_backward_transposed(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                             const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                             int64_t rows_cap, const int32_t* n_src_dev, int64_t src_cap, int64_t edges_cap,
                                             int32_t aggr, float* dsrc) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, src_cap >= 0 && edges_cap >= 0, "bad sizes");
  if (src_cap == 0) return GIGL_OK;
  const int64_t words = gigl_transposed_rows_words(src_cap, edges_cap);
  int32_t rc = gigl_arena_reset(ctx, words * 4 + 1024);
  if (rc != GIGL_OK) return rc;
  int32_t* lists = (int32_t*)gigl_arena_alloc(ctx, words * 4);
  if (!lists) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  rc = gigl_transposed_rows_build(ctx, rowptr, rowend, col, n_rows_dev, rows_cap, n_src_dev, src_cap, edges_cap, lists);
  if (rc != GIGL_OK) return rc;
  return gigl_gather_mean_backward_lists(ctx, dout, d, rowptr, rowend, n_rows_dev, n_src_dev, src_cap, lists, aggr, dsrc);
}

int32_t gigl_gather_reduce_backward(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                    const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                    int64_t rows_cap, int32_t aggr, const float* src, float* dsrc) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, dout && rowptr && rowend && col && n_rows_dev && dsrc, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && rows_cap >= 0, "bad sizes");
  GIGL_REQUIRE(ctx, aggr == GIGL_AGGR_MEAN || aggr == GIGL_AGGR_SUM || aggr == GIGL_AGGR_MAX, "bad aggr %d", aggr);
  GIGL_REQUIRE(ctx, aggr != GIGL_AGGR_MAX || src, "max needs the forward's source matrix");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_BWD);
  const int wpr = rows_cap >= 32768 ? 1 : (rows_cap >= 8192 ? 2 : (rows_cap >= 2048 ? 4 : 8));
  int64_t blocks = (rows_cap * wpr + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gather_mean_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dout, d,
                     rowptr, rowend, col, n_rows_dev, dsrc, aggr, src, wpr);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_linear(gigl_ctx* ctx, const float* a, const float* w, const float* bias,
                    const int32_t* m_dev, int64_t m_cap, int32_t k, int32_t n, int32_t act,
                    float* y) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, a && w && m_dev && y, "null argument");
  GIGL_REQUIRE(ctx, k > 0 && n > 0 && m_cap >= 0, "bad sizes");
  GIGL_REQUIRE(ctx, act == 0 || act == 1, "bad act %d", act);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m_cap == 0) return GIGL_OK;
  hipStream_t st = ctx->stream;
  gigl_prof_scope ps(ctx, GIGL_K_LINEAR);
  const int64_t tiles_m = (m_cap + 31) / 32;
  static const bool exact_only = getenv("GIGL_LINEAR_EXACT") != nullptr;  // (test / comparison knob)
  // (which kernel runs depends on k alone — never on the row count: the rows of a batch must come out bit-identical
  // whether the batch is computed alone or inside a group of batches)
  if (!exact_only) {  // split-precision bf16 MFMA (fp32-class accuracy); K % 4 != 0: the element-load instantiation
    const int64_t bm = (m_cap + 127) / 128;
    const bool kv = (k & 3) == 0;
    const dim3 g2((unsigned)(bm * ((n + 127) / 128))), g1((unsigned)(bm * ((n + 63) / 64)));
    if (n > 64 && kv)
      hipLaunchKernelGGL((linear_split_kernel<2, true>), g2, dim3(256), 0, st, a, w, bias, m_dev, k, n, act, y, 0, 0,
                         (int64_t)0, (int64_t)0);
    else if (n > 64)
      hipLaunchKernelGGL((linear_split_kernel<2, false>), g2, dim3(256), 0, st, a, w, bias, m_dev, k, n, act, y, 0, 0,
                         (int64_t)0, (int64_t)0);
    else if (kv)
      hipLaunchKernelGGL((linear_split_kernel<1, true>), g1, dim3(256), 0, st, a, w, bias, m_dev, k, n, act, y, 0, 0,
                         (int64_t)0, (int64_t)0);
    else
      hipLaunchKernelGGL((linear_split_kernel<1, false>), g1, dim3(256), 0, st, a, w, bias, m_dev, k, n, act, y, 0, 0,
                         (int64_t)0, (int64_t)0);
  } else if ((k & 3) == 0) {  // LDS-staged, coalesced operand fetch
    const int64_t bm = (m_cap + 127) / 128;
    if (n > 32) {
      hipLaunchKernelGGL((linear_lds_kernel<2>), dim3((unsigned)(bm * ((n + 63) / 64))), dim3(256), 0, st, a, w,
                         bias, m_dev, k, n, act, y);
    } else {
      hipLaunchKernelGGL((linear_lds_kernel<1>), dim3((unsigned)(bm * ((n + 31) / 32))), dim3(256), 0, st, a, w,
                         bias, m_dev, k, n, act, y);
    }
  } else if (n > 32) {
    constexpr int NT = 2;
    int64_t tiles = tiles_m * ((n + 32 * NT - 1) / (32 * NT));
    hipLaunchKernelGGL((linear_mfma_kernel<NT>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a, w,
                       bias, m_dev, k, n, act, y);
  } else {
    constexpr int NT = 1;
    int64_t tiles = tiles_m * ((n + 31) / 32);
    hipLaunchKernelGGL((linear_mfma_kernel<NT>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a, w,
                       bias, m_dev, k, n, act, y);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"

// the projection over an A operand in the tiled layout gigl_gather_reduce_mixed(..., tiled_nkc) writes
// ([row tile of 128][K chunk of 32][128 rows][32 floats], tiled_nkc = ceil(k / 32)); k % 4 == 0
// y rows may be a column slice of wider rows (ldy floats apart): the per-head projections of gigl_gat_input_layer
// rows per chunk of the weight gradient's split over the rows (gigl_linear_weight_grad / _parts)
static int wgrad_rows_per_chunk(int64_t m_cap, int32_t n, int32_t k) {
  // rows per chunk: 256 for the tens of thousands of rows of a first layer; fewer rows (the roots' layer: ~10^3) are cut
  // finer so that the reduction still spreads over a few hundred workgroups
  int rcw = WG_RC;
  while (rcw > 32 && m_cap / rcw < 32) rcw >>= 1;
  // ... and hundreds of thousands of rows (a whole batch graph under a wide table: the GAT layers' backward) coarser, so
  // that the partial sums stay a few tens of MB instead of a copy of dW per 256 rows
  const int64_t tiles = (int64_t)((n + 63) / 64) * ((k + 63) / 64);
  static const int64_t max_wgs = [] {  // (tuning knob: workgroups — chunks x tiles — the split aims to stay under)
    const char* e = getenv("GIGL_WGRAD_MAX_WGS");
    const int64_t v = e ? atoll(e) : 4096;  // (round 6: 8192 before; gat-lp --train 1.75 -> 1.73 ms, the other plans unchanged)
    return v < 256 ? (int64_t)256 : v;
  }();
  while (rcw < 8192 && ((m_cap + rcw - 1) / rcw) * tiles > max_wgs) rcw <<= 1;
  return rcw;
}

// The per-chunk PARTIAL sums only (no reduction launch): part [chunks][n][k], partb [chunks][n] (may be NULL) in the
// caller's buffers, *rows_per_chunk = the split.  chunks = ceil(m_cap / rows_per_chunk); only the chunks below
// ceil(*m_dev / rows_per_chunk) are written.  The training plans sum them inside their Adam kernel (one launch less per layer).
int64_t gigl_linear_weight_grad_chunks(int64_t m_cap, int32_t n, int32_t k, int32_t* rows_per_chunk) {
  const int rcw = wgrad_rows_per_chunk(m_cap, n, k);
  if (rows_per_chunk) *rows_per_chunk = rcw;
  return (m_cap + rcw - 1) / rcw;
}
int32_t gigl_linear_weight_grad_parts(gigl_ctx* ctx, const float* dy, const float* a, const float* relu_y, const int32_t* m_dev,
                                      int64_t m_cap, int32_t n, int32_t k, float* part, float* partb) {
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_LINEAR);
  const int rcw = wgrad_rows_per_chunk(m_cap, n, k);
  const int64_t chunks = (m_cap + rcw - 1) / rcw;
  const dim3 grid((unsigned)chunks, (unsigned)((n + 63) / 64), (unsigned)((k + 63) / 64));
  hipLaunchKernelGGL(linear_weight_grad_mfma_kernel, grid, dim3(256), 0, ctx->stream, dy, a, relu_y, m_dev, n, k, part, partb, rcw);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// dw += the chunks' partial sums (in chunk order), likewise db: the reduction launch of gigl_linear_weight_grad on its own —
// what a plan that keeps partial sums runs when somebody asks for the gradient itself (gigl_nablp_train_plan_grads)
int32_t gigl_linear_weight_grad_sum(gigl_ctx* ctx, const float* part, const float* partb, const int32_t* m_dev, int32_t n, int32_t k,
                                    int32_t rows_per_chunk, float* dw, float* db) {
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int64_t nk = (int64_t)n * k;
  hipLaunchKernelGGL(linear_weight_grad_reduce_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, part, partb,
                     m_dev, nk, n, dw, db, rows_per_chunk);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_linear_weight_grad(gigl_ctx* ctx, const float* dy, const float* a, const float* relu_y, const int32_t* m_dev,
                                int64_t m_cap, int32_t n, int32_t k, float* dw, float* db) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, dy && a && m_dev && dw, "null argument");
  GIGL_REQUIRE(ctx, n > 0 && k > 0 && m_cap >= 0, "bad sizes");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_LINEAR);
  const int rcw = wgrad_rows_per_chunk(m_cap, n, k);
  const int64_t chunks = (m_cap + rcw - 1) / rcw, nk = (int64_t)n * k;
  int32_t rc = gigl_arena_reset(ctx, chunks * (nk + n) * 4 + 1024);
  if (rc != GIGL_OK) return rc;
  float* part = (float*)gigl_arena_alloc(ctx, chunks * nk * 4);
  float* partb = db ? (float*)gigl_arena_alloc(ctx, chunks * n * 4) : nullptr;
  if (!part || (db && !partb)) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  const dim3 grid((unsigned)chunks, (unsigned)((n + 63) / 64), (unsigned)((k + 63) / 64));
  static const bool valu = getenv("GIGL_WGRAD_VALU") != nullptr;  // (A/B knob: the fp32 FMA kernel)
  if (valu)
    hipLaunchKernelGGL(linear_weight_grad_kernel, grid, dim3(256), 0, ctx->stream, dy, a, relu_y, m_dev, n, k, part, partb, rcw);
  else
    hipLaunchKernelGGL(linear_weight_grad_mfma_kernel, grid, dim3(256), 0, ctx->stream, dy, a, relu_y, m_dev, n, k, part, partb,
                       rcw);
  hipLaunchKernelGGL(linear_weight_grad_reduce_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, part,
                     partb, m_dev, nk, n, dw, db, rcw);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// largest |value|, sum of |value| and number of non-zero values of an array (one pass): what decides whether an operand
// may take the half split, and with which power-of-two scale
struct AbsStats {
  uint32_t max_bits;  // (non-negative floats order like their bits); +inf for a NaN anywhere: "unbounded"
  float sum;
  unsigned long long nz;
};

template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ p, int64_t n, AbsStats* __restrict__ out) {
  float m = 0.f, sum = 0.f;
  unsigned long long nz = 0;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = fabsf((float)p[i]);
    bad |= !(v == v);
    m = fmaxf(m, v);
    sum += v;
    nz += v != 0.f;
  }
  if (bad) m = __uint_as_float(0x7F800000u);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    sum += __shfl_xor(sum, o, 64);
    nz += __shfl_xor(nz, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out->max_bits, __float_as_uint(m));
    atomicAdd(&out->sum, sum);
    atomicAdd(&out->nz, nz);
  }
}

template <typename T>
static int32_t dev_absstats(gigl_ctx* ctx, const T* p, int64_t n, float* amax, float* mean_nz) {
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  AbsStats* d = nullptr;
  GIGL_HIP_CHECK(ctx, hipMalloc((void**)&d, sizeof(AbsStats)));
  hipMemsetAsync(d, 0, sizeof(AbsStats), ctx->stream);
  const int64_t wgs = std::min<int64_t>((n + 255) / 256 > 0 ? (n + 255) / 256 : 1, 8192);
  hipLaunchKernelGGL(absmax_kernel<T>, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, p, n, d);
  AbsStats h{};
  const hipError_t e1 = hipMemcpyAsync(&h, d, sizeof(AbsStats), hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipFree(d);
  GIGL_HIP_CHECK(ctx, e1);
  GIGL_HIP_CHECK(ctx, e2);
  memcpy(amax, &h.max_bits, 4);
  if (mean_nz) *mean_nz = h.nz ? h.sum / (float)h.nz : 0.f;
  return GIGL_OK;
}

bool gigl_half_split_enabled() {
  static const bool on = [] {
    const char* e = getenv("GIGL_GEMM_SPLIT");
    return !(e && strcmp(e, "bf16") == 0) && !getenv("GIGL_LINEAR_EXACT");
  }();
  return on;
}

int32_t gigl_dev_absmax_f32(gigl_ctx* ctx, const float* p, int64_t n, float* out) {
  return dev_absstats<float>(ctx, p, n, out, nullptr);
}

int32_t gigl_feat_absmax(gigl_ctx* ctx, gigl_feat* feat, float* out) {
  std::lock_guard<std::mutex> lk(feat->row_crc_mu);
  if (feat->absmax < 0.f) {
    float m = 0.f, mean = 0.f;
    const int64_t n = feat->n * feat->d;
    const int32_t rc = feat->dtype == GIGL_DTYPE_F16 ? dev_absstats<__half>(ctx, (const __half*)feat->rows, n, &m, &mean)
                                                     : dev_absstats<float>(ctx, (const float*)feat->rows, n, &m, &mean);
    if (rc != GIGL_OK) return rc;
    feat->absmean_nz = mean;
    feat->absmax = m;
  }
  *out = feat->absmax;
  return GIGL_OK;
}


int32_t gigl_feat_half_split_scale(gigl_ctx* ctx, gigl_feat* feat, float fan, float* s_a) {
  *s_a = 0.f;
  if (!gigl_half_split_enabled()) return GIGL_OK;
  float fmax = 0.f;
  const int32_t rc = gigl_feat_absmax(ctx, feat, &fmax);
  if (rc != GIGL_OK) return rc;
  const float top = fmax * fan;
  // outside [2^-46, 2^74] the scale would leave [2^-60, 2^60]; an operand whose typical (non-zero) magnitude lies more
  // than 2^10 below its largest (a table dominated by a few outliers) would put most of its elements next to the
  // subnormal halves — both stay on the bf16 planes, as does a table with a NaN / inf, or an all-zero one
  if (!(top >= 0x1p-46f && top <= 0x1p74f)) return GIGL_OK;
  if (!(feat->absmean_nz >= fmax * 0x1p-10f)) return GIGL_OK;
  *s_a = hs_pow2_scale(top);
  return GIGL_OK;
}

// hs[0] = s_a, hs[1] = s_w from the largest |w| as the weights are NOW, hs[2] = 1 / (s_a s_w)      (one workgroup)
// Spread over up to 64 small workgroups: each takes a slice's largest magnitude and folds it into a running maximum
// (non-negative floats order like their bits: atomicMax on the word); the LAST one to finish — ticket — turns the maximum
// into the scales and clears maximum and ticket for the next run.  (As one workgroup walking all the weights the kernel
// took ~100 us next to the other streams' kernels: a serial loop of 50 loads per thread under contention.)
// hs: [0..2] = {s_a, s_w, 1 / (s_a s_w)}, [4] = running maximum (bits), [5] = ticket — both zero between runs.
__global__ __launch_bounds__(256) void hs_scale_kernel(const float* __restrict__ w, int64_t n, float s_a,
                                                       float* __restrict__ hs) {
  __shared__ float s_m[4];
  float m = 0.f;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(w) & 15) == 0) ? n / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = w4[i];
    const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    m = fmaxf(m, a == a ? a : 0.f);  // (a NaN weight makes NaN rows on any path; it does not pick the scale)
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float v = fabsf(w[i]);
    m = fmaxf(m, v == v ? v : 0.f);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) m = fmaxf(m, s_m[i]);
    uint32_t* acc = reinterpret_cast<uint32_t*>(hs) + 4;
    atomicMax(acc, __float_as_uint(m));
    __threadfence();
    if (atomicAdd(acc + 1, 1u) == gridDim.x - 1) {  // the last workgroup: every maximum is in
      const float mm = __uint_as_float(atomicExch(acc, 0u));
      acc[1] = 0u;
      const float s_w = (mm > 0.f && mm < __uint_as_float(0x7F800000u)) ? hs_pow2_scale(mm) : 1.f;
      hs[0] = s_a;
      hs[1] = s_w;
      hs[2] = (1.f / s_a) * (1.f / s_w);
    }
  }
}

int32_t gigl_hs_scale_update(gigl_ctx* ctx, const float* w, int64_t n, float s_a, float* hs_dev) {
  int64_t wgs = (n / 4 + 255) / 256;  // one float4 per thread where the weights allow it
  wgs = wgs < 1 ? 1 : (wgs > 64 ? 64 : wgs);
  hipLaunchKernelGGL(hs_scale_kernel, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, w, n, s_a, hs_dev);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

static int32_t linear_tiled_strided(gigl_ctx* ctx, const float* a_tiled, const float* w, const float* bias,
                                    const int32_t* m_dev, int64_t m_cap, int32_t k, int32_t n, int32_t act, float* y,
                                    int32_t ldy, int32_t batch = 1, int64_t a_bstride = 0, int64_t w_bstride = 0,
                                    const float* self_src = nullptr, const uint32_t* self_ids = nullptr,
                                    int32_t d_mean = 0, int32_t self_ld = 0, const float* hs_scale = nullptr, bool self_half = false) {
  const bool hs = hs_scale != nullptr;  // half split: {s_a, s_w, 1 / (s_a s_w)} on the device (gigl_hs_scale_update)
  GIGL_REQUIRE(ctx, a_tiled && w && m_dev && y && (k & 3) == 0 && n > 0 && ldy >= n * batch && batch >= 1,
               "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m_cap == 0) return GIGL_OK;
  hipStream_t st = ctx->stream;
  gigl_prof_scope ps(ctx, GIGL_K_LINEAR);
  const int nkc = (k + 31) / 32;
  const int64_t bm = (m_cap + 127) / 128;
  if (self_src) {  // two-source operand: the tiled buffer holds the mean chunks only
    GIGL_REQUIRE(ctx, batch == 1 && d_mean > 0 && (d_mean & 3) == 0 && d_mean < k && self_ld >= k - d_mean, "bad two-source operand");
    const int nkc_mean = (d_mean + 31) / 32;
    GIGL_REQUIRE(ctx, !self_half || hs, "fp16 self rows go with the half split");
    if (hs && self_half) {
      if (n > 64)
        hipLaunchKernelGGL((linear_split_kernel<2, true, true, false, true, true, true>), dim3((unsigned)(bm * ((n + 127) / 128)), 1u),
                           dim3(256), 0, st, a_tiled, w, bias, m_dev, k, n, act, y, nkc_mean, ldy, (int64_t)0, (int64_t)0,
                           self_src, self_ids, d_mean, self_ld, hs_scale);
      else
        hipLaunchKernelGGL((linear_split_kernel<1, true, true, false, true, true, true>), dim3((unsigned)(bm * ((n + 63) / 64)), 1u),
                           dim3(256), 0, st, a_tiled, w, bias, m_dev, k, n, act, y, nkc_mean, ldy, (int64_t)0, (int64_t)0,
                           self_src, self_ids, d_mean, self_ld, hs_scale);
      GIGL_HIP_CHECK(ctx, hipGetLastError());
      return GIGL_OK;
    }
    if (hs) {
      if (n > 64)
        hipLaunchKernelGGL((linear_split_kernel<2, true, true, false, true, true>), dim3((unsigned)(bm * ((n + 127) / 128)), 1u),
                           dim3(256), 0, st, a_tiled, w, bias, m_dev, k, n, act, y, nkc_mean, ldy, (int64_t)0, (int64_t)0,
                           self_src, self_ids, d_mean, self_ld, hs_scale);
      else
        hipLaunchKernelGGL((linear_split_kernel<1, true, true, false, true, true>), dim3((unsigned)(bm * ((n + 63) / 64)), 1u),
                           dim3(256), 0, st, a_tiled, w, bias, m_dev, k, n, act, y, nkc_mean, ldy, (int64_t)0, (int64_t)0,
                           self_src, self_ids, d_mean, self_ld, hs_scale);
      GIGL_HIP_CHECK(ctx, hipGetLastError());
      return GIGL_OK;
    }
    if (n > 64)
      hipLaunchKernelGGL((linear_split_kernel<2, true, true>), dim3((unsigned)(bm * ((n + 127) / 128)), 1u), dim3(256), 0, st,
                         a_tiled, w, bias, m_dev, k, n, act, y, nkc_mean, ldy, (int64_t)0, (int64_t)0, self_src, self_ids,
                         d_mean, self_ld);
    else
      hipLaunchKernelGGL((linear_split_kernel<1, true, true, false, false, true>), dim3((unsigned)(bm * ((n + 63) / 64)), 1u), dim3(256), 0, st,
                         a_tiled, w, bias, m_dev, k, n, act, y, nkc_mean, ldy, (int64_t)0, (int64_t)0, self_src, self_ids,
                         d_mean, self_ld);
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    return GIGL_OK;
  }
  if (hs) {
    if (n > 64)
      hipLaunchKernelGGL((linear_split_kernel<2, true, false, false, true, true>), dim3((unsigned)(bm * ((n + 127) / 128)), (unsigned)batch),
                         dim3(256), 0, st, a_tiled, w, bias, m_dev, k, n, act, y, nkc, ldy, a_bstride, w_bstride,
                         (const float*)nullptr, (const uint32_t*)nullptr, 0, 0, hs_scale);
    else
      hipLaunchKernelGGL((linear_split_kernel<1, true, false, false, true, true>), dim3((unsigned)(bm * ((n + 63) / 64)), (unsigned)batch),
                         dim3(256), 0, st, a_tiled, w, bias, m_dev, k, n, act, y, nkc, ldy, a_bstride, w_bstride,
                         (const float*)nullptr, (const uint32_t*)nullptr, 0, 0, hs_scale);
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    return GIGL_OK;
  }
  if (n > 64)
    hipLaunchKernelGGL((linear_split_kernel<2, true, false, false, false, true>), dim3((unsigned)(bm * ((n + 127) / 128)), (unsigned)batch), dim3(256), 0,
                       st, a_tiled, w, bias, m_dev, k, n, act, y, nkc, ldy, a_bstride, w_bstride);
  else
    hipLaunchKernelGGL((linear_split_kernel<1, true, false, false, false, true>), dim3((unsigned)(bm * ((n + 63) / 64)), (unsigned)batch), dim3(256), 0,
                       st, a_tiled, w, bias, m_dev, k, n, act, y, nkc, ldy, a_bstride, w_bstride);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// ---- the fused two-layer projection (linear_fused2_kernel) and its companions
bool gigl_fused2_shape_ok(int32_t d0, int32_t hid, int32_t n_out) {
  return hid == F2_HID && n_out >= 1 && n_out <= F2_N2 / 2 && (d0 & 3) == 0 && d0 >= 4;
}
// [W2's planes as the round-5 kernel reads them | W1 chunk images | W2 round images]  (k1 = the first product's K)
static int64_t fused2_w2h_legacy_bytes() { return (int64_t)2 * F2_N2 * F2_HID * 2; }
static int64_t fused2_img1_bytes(int32_t k1) { return (int64_t)2 * ((k1 + 31) / 32) * F2_IMG1 * 2; }
int64_t gigl_fused2_w2h_bytes(int32_t k1) { return fused2_w2h_legacy_bytes() + fused2_img1_bytes(k1) + (int64_t)2 * 4 * F2_IMG2 * 2; }
static int fused2_variant() {
  static const int variant = [] {
    const char* e = getenv("GIGL_F2_VARIANT");
    return e ? atoi(e) : 7;  // 0: linear_fused2_kernel (round 5), 4: linear_fused2w_kernel, 7: linear_fused2x_kernel, 8: 7 + LDS-direct
  }();
  return variant;
}
int32_t gigl_fused2_row_floats() { return F2_N2; }
int32_t gigl_fused2_planes() { return fused2_variant() >= 7 ? 1 : 2; }

int32_t gigl_fused2_prepare(gigl_ctx* ctx, const float* hs_dev, const float* b1, const float* w1, const float* w2, int32_t n_out,
                            int32_t k1, float* f2, void* w2h) {
  // (f2: 16 floats, zero-initialised by the caller once: [4..6] are the scale kernel's accumulators + ticket)
  hipLaunchKernelGGL(fused2_scale_kernel, dim3(24), dim3(256), 0, ctx->stream, hs_dev, b1, w2, n_out * 2 * F2_HID, k1, f2);
  if (fused2_variant() >= 4) {
    _Float16* img1 = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(w2h) + fused2_w2h_legacy_bytes());
    _Float16* img2 = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(img1) + fused2_img1_bytes(k1));
    const int n = 2 * ((k1 + 31) / 32) * 128 * 32 + 2 * 4 * F2_N2 * 32;
    hipLaunchKernelGGL(fused2_images_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, hs_dev,
                       (const float*)f2, w1, k1, w2, n_out, img1, img2);
  } else
  hipLaunchKernelGGL(fused2_split_kernel, dim3((F2_N2 * F2_HID + 255) / 256), dim3(256), 0, ctx->stream, f2, w2, n_out,
                     (_Float16*)w2h);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_hs_chain_update(gigl_ctx* ctx, const float* hs_prev, const float* b_prev, int32_t n_b, int32_t k_prev, float fan,
                             const float* w, int64_t n_w, float* hs_out) {
  int64_t wgs = (n_w / 4 + 255) / 256;
  wgs = wgs < 1 ? 1 : (wgs > 64 ? 64 : wgs);
  hipLaunchKernelGGL(hs_chain_kernel, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, hs_prev, b_prev, n_b, k_prev, fan, w, n_w,
                     hs_out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_linear_fused2(gigl_ctx* ctx, const float* a_tiled, const float* w, const float* bias, const int32_t* m_dev,
                           int64_t m_cap, int32_t k, float* y2, int64_t plane_stride, const float* self_src,
                           const uint32_t* self_ids, int32_t d_mean, int32_t self_ld, const float* hs_scale,
                           const float* f2, const void* w2h, const int32_t* n_root_rows) {
  GIGL_REQUIRE(ctx, a_tiled && w && m_dev && y2 && self_src && hs_scale && f2 && w2h && (k & 3) == 0 && d_mean > 0 &&
                        (d_mean & 3) == 0 && d_mean < k && self_ld >= k - d_mean && (((uintptr_t)bias) & 15) == 0,
               "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_LINEAR);
  const int64_t bm = (m_cap + 127) / 128;
  const int variant = fused2_variant();
  if (variant >= 4) {
    const _Float16* img1 = reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(w2h) + fused2_w2h_legacy_bytes());
    const _Float16* img2 = reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(img1) + fused2_img1_bytes(k));
    static const int ablate = [] {
      const char* e = getenv("GIGL_F2_ABLATE");
      return e ? atoi(e) : 0;
    }();
#define GIGL_F2W_LAUNCH(AD, WPE, DBG)                                                                                         \
  hipLaunchKernelGGL((linear_fused2w_kernel<AD, WPE, DBG>), dim3((unsigned)(bm * 2)), dim3(256), 0, ctx->stream, a_tiled, bias, \
                     m_dev, k, y2, plane_stride, (d_mean + 31) / 32, self_src, self_ids, d_mean, self_ld, hs_scale, f2, img1,  \
                     img2, ablate)
    if (variant >= 7) {
#define GIGL_F2X_LAUNCH(DBG, GLDS, GLDS2)                                                                                   \
  hipLaunchKernelGGL((linear_fused2x_kernel<DBG, GLDS, GLDS2>), dim3((unsigned)bm), dim3(256), 0, ctx->stream, a_tiled, bias, \
                     m_dev, k, y2, (d_mean + 31) / 32, self_src, self_ids, d_mean, self_ld, hs_scale, f2, img1, img2, ablate, \
                     n_root_rows)
      if (ablate) GIGL_F2X_LAUNCH(true, false, true);
      else if (variant == 8) GIGL_F2X_LAUNCH(false, true, true);
      else if (variant == 9) GIGL_F2X_LAUNCH(false, false, false);
      else GIGL_F2X_LAUNCH(false, false, true);
#undef GIGL_F2X_LAUNCH
    } else if (ablate) GIGL_F2W_LAUNCH(1, 3, true);
    else GIGL_F2W_LAUNCH(1, 3, false);  // (two A chunks in flight at two waves per SIMD: 6.8 us per step against 5.9)
#undef GIGL_F2W_LAUNCH
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    return GIGL_OK;
  }
#define GIGL_F2_LAUNCH(PD, WPE)                                                                                              \
  hipLaunchKernelGGL((linear_fused2_kernel<PD, WPE>), dim3((unsigned)(bm * 2)), dim3(256), 0, ctx->stream, a_tiled, w, bias, \
                     m_dev, k, y2, plane_stride, (d_mean + 31) / 32, self_src, self_ids, d_mean, self_ld, hs_scale, f2,      \
                     (const _Float16*)w2h)
  // (measured and dropped, profiles/r06v_*: 2 / 3 chunks in flight at two waves per SIMD: 7.6 / 7.8 us per step against 6.3)
  GIGL_F2_LAUNCH(1, 3);
#undef GIGL_F2_LAUNCH
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_sage_fused_out(gigl_ctx* ctx, const float* p, int64_t plane_stride, const int32_t* rowptr,
                            const int32_t* rowend, const int32_t* col, const int32_t* root_local, int32_t b,
                            int32_t n_out, const float* bias, int32_t act, int32_t aggr, const int32_t* meta, float* out) {
  GIGL_REQUIRE(ctx, aggr == GIGL_AGGR_MEAN || aggr == GIGL_AGGR_SUM, "the fused last layer needs a linear reduction");
  if (b == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  const dim3 g((unsigned)(((int64_t)b + 3) / 4)), blk(256);
  const bool one = fused2_variant() >= 7;  // (linear_fused2x_kernel: whole p rows, one plane)
#define GIGL_FOUT(OP, PL)                                                                                                  \
  hipLaunchKernelGGL((sage_fused_out_kernel<OP, PL>), g, blk, 0, ctx->stream, p, plane_stride, rowptr, rowend, col, root_local, \
                     b, n_out, bias, act, meta, out)
  if (aggr == GIGL_AGGR_MEAN) {
    if (one) GIGL_FOUT(GIGL_AGGR_MEAN, 1);
    else GIGL_FOUT(GIGL_AGGR_MEAN, 2);
  } else {
    if (one) GIGL_FOUT(GIGL_AGGR_SUM, 1);
    else GIGL_FOUT(GIGL_AGGR_SUM, 2);
  }
#undef GIGL_FOUT
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// `batch` independent products of equal shape in one launch (row-major operands, k % 4 == 0): product b reads
// a + b * a_bstride ([m, k]) and w + b * w_bstride ([n, k]) and writes columns [b * n, (b + 1) * n) of y's rows
// (ldy >= batch * n floats apart).  The score blocks of a group of link-prediction batches (decoder.py:64-66, one
// torch.mm per batch in the reference) come out of one launch this way.
int32_t gigl_linear_batched(gigl_ctx* ctx, const float* a, const float* w, const float* bias, const int32_t* m_dev,
                            int64_t m_cap, int32_t k, int32_t n, int32_t act, int32_t batch, int64_t a_bstride,
                            int64_t w_bstride, int32_t ldy, float* y) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, a && w && m_dev && y, "null argument");
  GIGL_REQUIRE(ctx, k > 0 && (k & 3) == 0 && n > 0 && m_cap >= 0 && batch >= 1 && batch <= 65535 &&
                        (int64_t)ldy >= (int64_t)n * batch && a_bstride >= 0 && w_bstride >= 0,
               "bad sizes (k=%d n=%d batch=%d ldy=%d)", k, n, batch, ldy);
  GIGL_REQUIRE(ctx, act == 0 || act == 1, "bad act %d", act);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_LINEAR);
  const int64_t bm = (m_cap + 127) / 128;
  if (n > 64)
    hipLaunchKernelGGL((linear_split_kernel<2, true>), dim3((unsigned)(bm * ((n + 127) / 128)), (unsigned)batch),
                       dim3(256), 0, ctx->stream, a, w, bias, m_dev, k, n, act, y, 0, ldy, a_bstride, w_bstride);
  else
    hipLaunchKernelGGL((linear_split_kernel<1, true>), dim3((unsigned)(bm * ((n + 63) / 64)), (unsigned)batch),
                       dim3(256), 0, ctx->stream, a, w, bias, m_dev, k, n, act, y, 0, ldy, a_bstride, w_bstride);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_linear_grouped(gigl_ctx* ctx, const gigl_linear_group* groups_dev, int32_t n_groups, int64_t m_cap_max,
                            int32_t k, int32_t n, int32_t act) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, groups_dev || n_groups == 0, "null argument");
  GIGL_REQUIRE(ctx, k > 0 && (k & 3) == 0 && n > 0 && m_cap_max >= 0 && n_groups >= 0 && n_groups <= 65535,
               "bad sizes (k=%d n=%d groups=%d)", k, n, n_groups);
  GIGL_REQUIRE(ctx, act == 0 || act == 1, "bad act %d", act);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m_cap_max == 0 || n_groups == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_LINEAR);
  const int64_t bm = (m_cap_max + 127) / 128;
  const float* none = nullptr;
  const int32_t* none_i = nullptr;
  float* none_y = nullptr;
  const uint32_t* none_u = nullptr;
  if (n > 64)
    hipLaunchKernelGGL((linear_split_kernel<2, true>), dim3((unsigned)(bm * ((n + 127) / 128)), (unsigned)n_groups),
                       dim3(256), 0, ctx->stream, none, none, none, none_i, k, n, act, none_y, 0, n, (int64_t)0, (int64_t)0, none,
                       none_u, 0, 0, none, groups_dev);
  else
    hipLaunchKernelGGL((linear_split_kernel<1, true>), dim3((unsigned)(bm * ((n + 63) / 64)), (unsigned)n_groups),
                       dim3(256), 0, ctx->stream, none, none, none, none_i, k, n, act, none_y, 0, n, (int64_t)0, (int64_t)0, none,
                       none_u, 0, 0, none, groups_dev);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_linear_tiled(gigl_ctx* ctx, const float* a_tiled, const float* w, const float* bias, const int32_t* m_dev,
                          int64_t m_cap, int32_t k, int32_t n, int32_t act, float* y, const float* self_src,
                          const uint32_t* self_ids, int32_t d_mean, int32_t self_ld, const float* hs_scale, bool self_half) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  return linear_tiled_strided(ctx, a_tiled, w, bias, m_dev, m_cap, k, n, act, y, n, 1, 0, 0, self_src, self_ids, d_mean,
                              self_ld, hs_scale, self_half);
}

int64_t gigl_gat_input_layer_scratch(int32_t d, int32_t heads, int64_t cap_nodes, int64_t rows_cap, int64_t cap_edges) {
  const int64_t nkc = (d + 31) / 32, row_tiles = (rows_cap + 127) / 128, chunks = (d + 255) / 256;
  return (int64_t)2 * heads * d + chunks * cap_nodes * 2 * heads + cap_edges * (1 + heads) + rows_cap * heads +
         (int64_t)heads * row_tiles * nkc * 4096;
}

int32_t gigl_gat_input_layer(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* ids,
                             const float* w, const float* att_src, const float* att_dst, int32_t heads,
                             int32_t channels, float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                             const int32_t* col, int64_t cap_edges, const int32_t* n_src_dev, int64_t cap_nodes,
                             const int32_t* n_rows_dev, int64_t rows_cap, const float* bias, int32_t act,
                             float* scratch, float* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, src && ids && w && att_src && att_dst && rowptr && rowend && col && n_src_dev && n_rows_dev &&
                        scratch && out, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && heads > 0 && channels > 0 && cap_nodes >= rows_cap && rows_cap >= 0 && cap_edges >= 0,
               "bad sizes");
  GIGL_REQUIRE(ctx, src_dtype == GIGL_DTYPE_F32 || src_dtype == GIGL_DTYPE_F16, "bad dtype %d", src_dtype);
  GIGL_REQUIRE(ctx, act == 0 || act == 1, "bad act %d", act);
  if ((d & 3) || (heads != 1 && heads != 2 && heads != 4))
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "gigl_gat_input_layer: d=%d heads=%d outside the built shapes (d %% 4 == 0, "
                     "heads 1|2|4)", d, heads);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  hipStream_t st = ctx->stream;
  const int H = heads, C = channels;
  const int nkc = (d + 31) / 32, chunks = (d + 255) / 256;
  const int64_t row_tiles = (rows_cap + 127) / 128, head_stride = row_tiles * nkc * 4096;
  float* z = scratch;  // (first: its float4 accesses need the base's 16-byte alignment)
  float* u = z + H * head_stride;
  float* sp = u + (int64_t)2 * H * d;
  uint32_t* gid_e = reinterpret_cast<uint32_t*>(sp + (int64_t)chunks * cap_nodes * 2 * H);
  float* alpha_e = reinterpret_cast<float*>(gid_e + cap_edges);
  float* alpha_self = alpha_e + cap_edges * H;
  {
    gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
    hipLaunchKernelGGL(gat_fold_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)(2 * H)), dim3(256), 0, st, w, att_src,
                       att_dst, H, C, d, u);
    int64_t sblocks = (cap_nodes + 63) / 64, wblocks = (rows_cap + 3) / 4, gblocks = (rows_cap * chunks + 3) / 4;
    if (sblocks > 256 * 8) sblocks = 256 * 8;
    if (wblocks > 256 * 16) wblocks = 256 * 16;
    if (gblocks > 256 * 8) gblocks = 256 * 8;
#define GIGL_GAT_IN(TT, HH)                                                                                             \
  do {                                                                                                                  \
    hipLaunchKernelGGL((gat_score_kernel<TT, HH>), dim3((unsigned)sblocks, (unsigned)chunks), dim3(256), 0, st,        \
                       (const TT*)src, d, ids, n_src_dev, cap_nodes, u, sp);                                            \
    hipLaunchKernelGGL((gat_edge_weight_kernel<HH>), dim3((unsigned)wblocks), dim3(256), 0, st, ids, sp, chunks,       \
                       cap_nodes, rowptr, rowend, col, n_rows_dev, negative_slope, gid_e, alpha_e, alpha_self);         \
    hipLaunchKernelGGL((gat_input_gather_kernel<TT, HH>), dim3((unsigned)gblocks), dim3(256), 0, st, (const TT*)src,   \
                       d, ids, gid_e, alpha_e, alpha_self, chunks, rowptr, rowend, n_rows_dev, nkc, head_stride, z);    \
  } while (0)
#define GIGL_GAT_IN_H(TT)                                                                                               \
  do {                                                                                                                  \
    if (H == 1) GIGL_GAT_IN(TT, 1);                                                                                     \
    else if (H == 2) GIGL_GAT_IN(TT, 2);                                                                                \
    else GIGL_GAT_IN(TT, 4);                                                                                            \
  } while (0)
    if (src_dtype == GIGL_DTYPE_F32) GIGL_GAT_IN_H(float);
    else GIGL_GAT_IN_H(__half);
#undef GIGL_GAT_IN_H
#undef GIGL_GAT_IN
    GIGL_HIP_CHECK(ctx, hipGetLastError());
  }
  // the heads' projections in one launch (grid.y = head): z_h [rows][d] (tiled) x W_h^T -> columns [h*C, +C) of out
  return linear_tiled_strided(ctx, z, w, bias, n_rows_dev, rows_cap, d, C, act, out, H * C, H, head_stride,
                              (int64_t)C * d);
}

int64_t gigl_gat_input_layer_fused_scratch(int32_t d, int32_t heads, int64_t rows_cap) {
  const int64_t nkc = (d + 31) / 32, row_tiles = (rows_cap + 127) / 128;
  return (int64_t)heads * row_tiles * nkc * 4096 + (int64_t)2 * heads * d;
}

// the training path's two halves of that layer: the aggregation alone over folded vectors the caller supplies (plain
// rows out), and its backward w.r.t. those vectors
static int32_t gat_input_shape_ok(gigl_ctx* ctx, int32_t d, int32_t heads, int32_t src_dtype, const char* who) {
  if ((d & 3) || (heads != 1 && heads != 2 && heads != 4) || d > 1024)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "%s: d=%d heads=%d outside the built shapes (d %% 4 == 0, d <= 1024, heads 1|2|4)",
                     who, d, heads);
  if (src_dtype != GIGL_DTYPE_F32 && src_dtype != GIGL_DTYPE_F16)
    return gigl_fail(ctx, GIGL_E_INVALID_ARG, "%s: bad dtype %d", who, src_dtype);
  return GIGL_OK;
}

int32_t gigl_gat_input_aggregate(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* gather_ids,
                                 const float* u, int32_t heads, float negative_slope, const int32_t* rowptr,
                                 const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap,
                                 float* z) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, src && gather_ids && u && rowptr && rowend && col && n_rows_dev && z, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && rows_cap >= 0, "bad sizes");
  int32_t rc = gat_input_shape_ok(ctx, d, heads, src_dtype, "gigl_gat_input_aggregate");
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
  hipStream_t st = ctx->stream;
  const int P = (d + 255) / 256, H = heads;
  const int64_t head_stride = rows_cap * d;
  const int32_t* n_local_dev = nullptr;
  const int nkc = 0;
  const float slope = negative_slope;
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
#define GIGL_GAT_AG(TT, PP, HH)                                                                                        \
  hipLaunchKernelGGL((gat_input_online_kernel<TT, PP, HH>), dim3((unsigned)blocks), dim3(256), 0, st, (const TT*)src, d, \
                     gather_ids, n_local_dev, u, rowptr, rowend, col, n_rows_dev, slope, nkc, head_stride, z)
#define GIGL_GAT_AG_P(TT, HH)                                                                                          \
  do {                                                                                                                  \
    if (P == 1) GIGL_GAT_AG(TT, 1, HH);                                                                                \
    else if (P == 2) GIGL_GAT_AG(TT, 2, HH);                                                                           \
    else if (P == 3) GIGL_GAT_AG(TT, 3, HH);                                                                           \
    else GIGL_GAT_AG(TT, 4, HH);                                                                                       \
  } while (0)
#define GIGL_GAT_AG_H(TT)                                                                                              \
  do {                                                                                                                  \
    if (H == 1) GIGL_GAT_AG_P(TT, 1);                                                                                  \
    else if (H == 2) GIGL_GAT_AG_P(TT, 2);                                                                             \
    else GIGL_GAT_AG_P(TT, 4);                                                                                         \
  } while (0)
  if (src_dtype == GIGL_DTYPE_F32) GIGL_GAT_AG_H(float);
  else GIGL_GAT_AG_H(__half);
#undef GIGL_GAT_AG_H
#undef GIGL_GAT_AG_P
#undef GIGL_GAT_AG
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gat_input_aggregate_backward(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d,
                                          const uint32_t* gather_ids, const float* u, int32_t heads, float negative_slope,
                                          const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                          const int32_t* n_rows_dev, int64_t rows_cap, const float* dz, float* edge_scratch,
                                          float* du) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, src && gather_ids && u && rowptr && rowend && col && n_rows_dev && dz && edge_scratch && du, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && rows_cap >= 0, "bad sizes");
  int32_t rc = gat_input_shape_ok(ctx, d, heads, src_dtype, "gigl_gat_input_aggregate_backward");
  if (rc != GIGL_OK) return rc;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  gigl_prof_scope ps(ctx, GIGL_K_GATHER_BWD);
  hipStream_t st = ctx->stream;
  const int P = (d + 255) / 256, H = heads;
  const int64_t head_stride = rows_cap * d;
  const float slope = negative_slope;
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 2) blocks = 256 * 2;  // (two workgroups per CU fill its registers; fewer workgroups = fewer final atomics)
  // (A/B knob: GIGL_GAT_BWD_TWO_SWEEPS=1 keeps the two-sweep kernel — every source row read twice, the edges' logits parked)
  static const bool two_sweeps = getenv("GIGL_GAT_BWD_TWO_SWEEPS") != nullptr;
#define GIGL_GAT_BW(TT, PP, HH)                                                                                        \
  do {                                                                                                                  \
    if (two_sweeps)                                                                                                     \
      hipLaunchKernelGGL((gat_input_backward_kernel<TT, PP, HH>), dim3((unsigned)blocks), dim3(256), 0, st,            \
                         (const TT*)src, d, gather_ids, u, rowptr, rowend, col, n_rows_dev, slope, dz, head_stride,     \
                         edge_scratch, du);                                                                            \
    else                                                                                                                \
      hipLaunchKernelGGL((gat_input_backward_onepass_kernel<TT, PP, HH>), dim3((unsigned)blocks), dim3(256), 0, st,    \
                         (const TT*)src, d, gather_ids, u, rowptr, rowend, col, n_rows_dev, slope, dz, head_stride, du); \
  } while (0)
#define GIGL_GAT_BW_P(TT, HH)                                                                                          \
  do {                                                                                                                  \
    if (P == 1) GIGL_GAT_BW(TT, 1, HH);                                                                                \
    else if (P == 2) GIGL_GAT_BW(TT, 2, HH);                                                                           \
    else if (P == 3) GIGL_GAT_BW(TT, 3, HH);                                                                           \
    else GIGL_GAT_BW(TT, 4, HH);                                                                                       \
  } while (0)
#define GIGL_GAT_BW_H(TT)                                                                                              \
  do {                                                                                                                  \
    if (H == 1) GIGL_GAT_BW_P(TT, 1);                                                                                  \
    else if (H == 2) GIGL_GAT_BW_P(TT, 2);                                                                             \
    else GIGL_GAT_BW_P(TT, 4);                                                                                         \
  } while (0)
  if (src_dtype == GIGL_DTYPE_F32) GIGL_GAT_BW_H(float);
  else GIGL_GAT_BW_H(__half);
#undef GIGL_GAT_BW_H
#undef GIGL_GAT_BW_P
#undef GIGL_GAT_BW
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_gat_input_layer_fused(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d,
                                   const uint32_t* gather_ids, const int32_t* n_local_dev, const float* w,
                                   const float* att_src, const float* att_dst, int32_t heads, int32_t channels,
                                   float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                                   const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, const float* bias,
                                   int32_t act, float* scratch, float* out) {
  return gigl_gat_input_layer_fused_hs(ctx, src, src_dtype, d, gather_ids, n_local_dev, w, att_src, att_dst, heads, channels,
                                       negative_slope, rowptr, rowend, col, n_rows_dev, rows_cap, bias, act, scratch, out,
                                       nullptr);
}

// hs_scale != NULL: the projection of the aggregated rows over two fp16 planes per operand — the rows are convex
// combinations of table rows (softmax weights), so the table's largest magnitude bounds them: s_a comes from it, s_w from the weights'
int32_t gigl_gat_input_layer_fused_hs(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d,
                                      const uint32_t* gather_ids, const int32_t* n_local_dev, const float* w,
                                      const float* att_src, const float* att_dst, int32_t heads, int32_t channels,
                                      float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                                      const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, const float* bias,
                                      int32_t act, float* scratch, float* out, const float* hs_scale) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, src && gather_ids && w && att_src && att_dst && rowptr && rowend && col && n_rows_dev && scratch && out,
               "null argument");
  GIGL_REQUIRE(ctx, d > 0 && heads > 0 && channels > 0 && rows_cap >= 0, "bad sizes");
  GIGL_REQUIRE(ctx, src_dtype == GIGL_DTYPE_F32 || src_dtype == GIGL_DTYPE_F16, "bad dtype %d", src_dtype);
  GIGL_REQUIRE(ctx, act == 0 || act == 1, "bad act %d", act);
  const int P = (d + 255) / 256;
  if ((d & 3) || (heads != 1 && heads != 2 && heads != 4) || P > 4)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "gigl_gat_input_layer_fused: d=%d heads=%d outside the built shapes (d %% 4 "
                     "== 0, d <= 1024, heads 1|2|4)", d, heads);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  hipStream_t st = ctx->stream;
  const int H = heads, C = channels;
  const int nkc = (d + 31) / 32;
  const int64_t row_tiles = (rows_cap + 127) / 128, head_stride = row_tiles * nkc * 4096;
  float* z = scratch;  // (first: 16-byte aligned)
  float* u = z + H * head_stride;
  {
    gigl_prof_scope ps(ctx, GIGL_K_GATHER_MEAN);
    hipLaunchKernelGGL(gat_fold_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)(2 * H)), dim3(256), 0, st, w, att_src,
                       att_dst, H, C, d, u);
    int64_t blocks = (rows_cap + 3) / 4;  // one wave per row, four rows per workgroup
    if (blocks > 256 * 16) blocks = 256 * 16;
#define GIGL_GAT_ON(TT, PP, HH)                                                                                         \
  hipLaunchKernelGGL((gat_input_online_kernel<TT, PP, HH>), dim3((unsigned)blocks), dim3(256), 0, st,                 \
                     (const TT*)src, d, gather_ids, n_local_dev, u, rowptr, rowend, col, n_rows_dev, negative_slope,    \
                     nkc, head_stride, z)
#define GIGL_GAT_ON_P(TT, HH)                                                                                           \
  do {                                                                                                                  \
    if (P == 1) GIGL_GAT_ON(TT, 1, HH);                                                                                 \
    else if (P == 2) GIGL_GAT_ON(TT, 2, HH);                                                                            \
    else if (P == 3) GIGL_GAT_ON(TT, 3, HH);                                                                            \
    else GIGL_GAT_ON(TT, 4, HH);                                                                                        \
  } while (0)
#define GIGL_GAT_ON_H(TT)                                                                                               \
  do {                                                                                                                  \
    if (H == 1) GIGL_GAT_ON_P(TT, 1);                                                                                   \
    else if (H == 2) GIGL_GAT_ON_P(TT, 2);                                                                              \
    else GIGL_GAT_ON_P(TT, 4);                                                                                          \
  } while (0)
    if (src_dtype == GIGL_DTYPE_F32) GIGL_GAT_ON_H(float);
    else GIGL_GAT_ON_H(__half);
#undef GIGL_GAT_ON_H
#undef GIGL_GAT_ON_P
#undef GIGL_GAT_ON
    GIGL_HIP_CHECK(ctx, hipGetLastError());
  }
  return linear_tiled_strided(ctx, z, w, bias, n_rows_dev, rows_cap, d, C, act, out, H * C, H, head_stride, (int64_t)C * d,
                              nullptr, nullptr, 0, 0, hs_scale);
}
