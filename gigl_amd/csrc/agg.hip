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
#include "common.h"

#include <hip/hip_fp16.h>

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
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void gather_mean_kernel(const T* __restrict__ src, int d,
                                                          const uint32_t* __restrict__ gather_ids,
                                                          const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ rowend,
                                                          const int32_t* __restrict__ col,
                                                          const int32_t* __restrict__ n_rows_dev,
                                                          float* __restrict__ out) {
  constexpr int G = 64 / LPR;  // source rows per wave-instruction
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR;  // which source row of the instruction
  const int sl = lane % LPR;   // lane within the row
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_rows = *n_rows_dev;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    const int self = gather_ids ? (int)gather_ids[i] : i;
    float4_t acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = zero4;
    for (int c0 = 0; c0 < m; c0 += 64) {
      const int mm = min(64, m - c0);
      int my = 0;
      if (lane < mm) {
        my = col[e0 + c0 + lane];
        if (gather_ids) my = (int)gather_ids[my];
      }
      for (int e = 0; e < mm; e += 4 * G) {  // wave-uniform trip count (shuffles need every lane)
        const int ea = e + sub, eb = ea + G, ec = ea + 2 * G, ed = ea + 3 * G;
        const int ja = __shfl(my, ea & 63, 64), jb = __shfl(my, eb & 63, 64), jc = __shfl(my, ec & 63, 64),
                  jd = __shfl(my, ed & 63, 64);
        const T* pa = src + (int64_t)ja * d;
        const T* pb = src + (int64_t)jb * d;
        const T* pc = src + (int64_t)jc * d;
        const T* pd = src + (int64_t)jd * d;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int el = (v * LPR + sl) * 4;
          if (el < d) {
            float4_t a = ea < mm ? RowLoader<T>::load4(pa, el) : zero4;
            float4_t b = eb < mm ? RowLoader<T>::load4(pb, el) : zero4;
            float4_t c = ec < mm ? RowLoader<T>::load4(pc, el) : zero4;
            float4_t dd = ed < mm ? RowLoader<T>::load4(pd, el) : zero4;
            acc[v] += (a + b) + (c + dd);
          }
        }
      }
    }
    // combine the G partial sums (lanes sl, sl+LPR, ...)
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        acc[v].x += __shfl_xor(acc[v].x, off, 64);
        acc[v].y += __shfl_xor(acc[v].y, off, 64);
        acc[v].z += __shfl_xor(acc[v].z, off, 64);
        acc[v].w += __shfl_xor(acc[v].w, off, 64);
      }
    }
    float* o = out + (int64_t)i * 2 * d;
    const T* ps = src + (int64_t)self * d;
    if (sub == 0) {
      // mean = sum / deg (a true division, like torch's scatter-mean), 0 for an empty row
      const float dv = m > 0 ? (float)m : 1.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int el = (v * LPR + sl) * 4;
        if (el < d) *reinterpret_cast<float4_t*>(o + el) = acc[v] / dv;
      }
    }
    // self row copy, spread over all lanes of the wave
    for (int el = lane * 4; el < d; el += 64 * 4)
      *reinterpret_cast<float4_t*>(o + d + el) = RowLoader<T>::load4(ps, el);
  }
}

// backward of gather_mean w.r.t. a dense local source matrix (layers >= 2; raw input features need no
// gradient):  dsrc[i] += dout[i][d:2d];  dsrc[col[e]] += dout[i][0:d] / deg_i for every edge e of row i.
// One wave per destination row; fp32 atomics (several rows can share a source).
__global__ __launch_bounds__(256) void gather_mean_backward_kernel(const float* __restrict__ dout, int d,
                                                                   const int32_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ rowend,
                                                                   const int32_t* __restrict__ col,
                                                                   const int32_t* __restrict__ n_rows_dev,
                                                                   float* __restrict__ dsrc) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_rows = *n_rows_dev;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], m = rowend[i] - e0;
    const float* g = dout + (int64_t)i * 2 * d;
    for (int el = lane; el < d; el += 64) atomicAdd(&dsrc[(int64_t)i * d + el], g[d + el]);
    if (m == 0) continue;
    const float inv = 1.0f / (float)m;
    for (int e = 0; e < m; ++e) {
      const int j = col[e0 + e];
      for (int el = lane; el < d; el += 64) atomicAdd(&dsrc[(int64_t)j * d + el], g[el] * inv);
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
                                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_rows = *n_rows_dev;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n_rows; i += waves_total) {
    const int e0 = rowptr[i], e1 = rowend[i];
    const int deg = e1 - e0;
    float* o = out + (int64_t)i * 2 * d;
    const int self = gather_ids ? (int)gather_ids[i] : i;
    for (int el = lane; el < d; el += 64) {
      float acc = 0.f;
      for (int e = e0; e < e1; ++e) {
        int j = col[e];
        if (gather_ids) j = (int)gather_ids[j];
        acc += (float)src[(int64_t)j * d + el];
      }
      o[el] = deg > 0 ? acc / (float)deg : 0.f;
      o[d + el] = (float)src[(int64_t)self * d + el];
    }
  }
}

// ------------------------------------------------------------------------------------------
// y[m][n] = act(sum_k a[m][k] * w[n][k] + bias[n])       a: [M][K], w: [N][K] row-major, fp32
// ------------------------------------------------------------------------------------------
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

  float4_t ga[4], gw[NT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + lr + 8 * i, kk = k0 + lc * 4;
      ga[i] = (row < M && kk < K) ? *reinterpret_cast<const float4_t*>(a + (int64_t)row * K + kk) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int row = n0 + wr + 32 * i, kk = k0 + wc * 4;
      gw[i] = (row < N && kk < K) ? *reinterpret_cast<const float4_t*>(w + (int64_t)row * K + kk) : zero4;
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // previous chunk's LDS reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4_t*>(&s_a[wv_id][(lr + 8 * i) * LDK + lc * 4]) = ga[i];
#pragma unroll
    for (int i = 0; i < NT; ++i) *reinterpret_cast<float4_t*>(&s_w[(wr + 32 * i) * LDK + wc * 4]) = gw[i];
    __syncthreads();
    if (k0 + BK < K) gload(k0 + BK);  // in flight under the MFMAs below
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
  }
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

template <typename T>
int32_t launch_gather(gigl_ctx* ctx, const T* src, int d, const uint32_t* gather_ids,
                      const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                      const int32_t* n_rows_dev, int64_t rows_cap, float* out) {
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  dim3 g((unsigned)blocks), b(256);
  hipStream_t st = ctx->stream;
  const int vecs = d / 4;
#define GL(LPR, VPL)                                                                             \
  hipLaunchKernelGGL((gather_mean_kernel<T, LPR, VPL>), g, b, 0, st, src, d, gather_ids, rowptr, \
                     rowend, col, n_rows_dev, out)
  if ((d & 3) != 0 || vecs > 512) {
    hipLaunchKernelGGL((gather_mean_generic_kernel<T>), g, b, 0, st, src, d, gather_ids, rowptr, rowend,
                       col, n_rows_dev, out);
  } else if (vecs <= 8) GL(8, 1);
  else if (vecs <= 16) GL(16, 1);
  else if (vecs <= 32) GL(32, 1);
  else if (vecs <= 64) GL(64, 1);
  else if (vecs <= 128) GL(64, 2);
  else if (vecs <= 256) GL(64, 4);
  else GL(64, 8);
#undef GL
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // namespace

extern "C" {

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

int32_t gigl_gather_mean_backward(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                  const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                  int64_t rows_cap, float* dsrc) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, dout && rowptr && rowend && col && n_rows_dev && dsrc, "null argument");
  GIGL_REQUIRE(ctx, d > 0 && rows_cap >= 0, "bad sizes");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows_cap == 0) return GIGL_OK;
  int64_t blocks = (rows_cap + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gather_mean_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dout, d,
                     rowptr, rowend, col, n_rows_dev, dsrc);
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
  if ((k & 3) == 0) {  // LDS-staged, coalesced operand fetch
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
