// loss.hip — retrieval loss of the link-prediction head, fused: temperature -> sampling-probability correction ->
// duplicate / accidental-hit masking -> log-softmax -> cross-entropy against the diagonal, one pass over the scores.
//
// Replaces RetrievalLoss.calculate_batch_retrieval_loss and its two mask builders
// (python/gigl/src/common/models/layers/loss.py:209-277, :279-305, :307-331), which materialise four [Q, C]
// tensors (labels, duplicates, two masks) around CrossEntropyLoss(reduction="sum").  Here row i of the score matrix is
// read once by one 256-thread workgroup:
//     s_ij  = scores_ij / temperature - log(max(p_j, 1e-10))
//     column j != i is EXCLUDED when it is another row of the same query (j < Q and query_ids[j] == query_ids[i]) or,
//     with accidental-hit removal, holds the positive's candidate (candidate_ids[j] == candidate_ids[i]) — the
//     reference adds finfo.min to such a logit, which makes its softmax term exactly 0 in fp32
//     loss_i = logsumexp_j s_ij - s_ii
// Row results go to row_lse / row_loss; a second single-workgroup launch adds the rows in a fixed order (the sum is
// reproducible run to run).  Backward: dscores_ij = g * (softmax_ij - [i == j]) / temperature, 0 for excluded columns.
// HBM-bound: 4*Q*C bytes read forward, read + written backward.
#include "common.h"

#include <cfloat>

namespace {

struct LossArgs {
  const float* scores;
  int64_t ld;
  int32_t q, c;
  float temperature;  // <= 0: none
  const float* cand_prob;
  const int64_t* query_ids;
  const int64_t* cand_ids;
  int64_t g_scores;  // grid.y = independent batches: batch g's score block starts g_scores floats on, its ids / prob
                     // / per-row outputs follow the previous batch's (q or c entries each)
};

__device__ __forceinline__ float logit(const LossArgs& a, float raw, int j) {
  float s = a.temperature > 0.f ? raw / a.temperature : raw;
  if (a.cand_prob) s -= logf(fmaxf(a.cand_prob[j], 1e-10f));
  return s;
}

__device__ __forceinline__ bool excluded(const LossArgs& a, int i, int j, int64_t qid_i, int64_t cid_i) {
  if (j == i) return false;
  if (a.query_ids && j < a.q && a.query_ids[j] == qid_i) return true;
  return a.cand_ids && a.cand_ids[j] == cid_i;
}

// merge two (max, sum of exp relative to max) states
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  const float mm = fmaxf(m, m2);
  if (mm == -INFINITY) return;
  s = s * expf(m - mm) + s2 * expf(m2 - mm);
  m = mm;
}

__global__ __launch_bounds__(256) void retrieval_rows_kernel(LossArgs a0, float* __restrict__ masked_out,
                                                             float* __restrict__ row_lse,
                                                             float* __restrict__ row_loss) {
  __shared__ float s_m[4], s_s[4];
  LossArgs a = a0;
  if (const int64_t g = blockIdx.y) {
    a.scores += g * a.g_scores;
    if (a.cand_prob) a.cand_prob += g * a.c;
    if (a.query_ids) a.query_ids += g * a.q;
    if (a.cand_ids) a.cand_ids += g * a.c;
    if (masked_out) masked_out += g * a.q * a.c;
    row_lse += g * a.q;
    row_loss += g * a.q;
  }
  const int i = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* row = a.scores + (int64_t)i * a.ld;
  const int64_t qid_i = a.query_ids ? a.query_ids[i] : 0;
  const int64_t cid_i = a.cand_ids ? a.cand_ids[i] : 0;
  float m = -INFINITY, s = 0.f;
  for (int j = tid; j < a.c; j += 256) {
    const float v = logit(a, row[j], j);
    const bool ex = excluded(a, i, j, qid_i, cid_i);
    if (masked_out) masked_out[(int64_t)i * a.c + j] = ex ? v + (-FLT_MAX) : v;  // the reference's masked logits
    if (ex) continue;
    if (v > m) {
      s = s * expf(m - v) + 1.f;
      m = v;
    } else {
      s += expf(v - m);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
    lse_merge(m, s, m2, s2);
  }
  if (lane == 0) {
    s_m[w] = m;
    s_s[w] = s;
  }
  __syncthreads();
  if (tid == 0) {
    float mm = s_m[0], ss = s_s[0];
    for (int k = 1; k < 4; ++k) lse_merge(mm, ss, s_m[k], s_s[k]);
    const float lse = mm + logf(ss);
    row_lse[i] = lse;
    row_loss[i] = lse - logit(a, row[i], i);
  }
}

// fixed-order sum of n floats into *out (one workgroup; accumulates in double)
__global__ __launch_bounds__(1024) void sum_rows_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ double s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  v += (int64_t)blockIdx.x * n;  // (one workgroup per batch)
  out += blockIdx.x;
  double acc = 0.0;
  for (int i = tid; i < n; i += 1024) acc += (double)v[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) s_w[w] = acc;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += s_w[k];
    *out = (float)t;
  }
}

__global__ __launch_bounds__(256) void retrieval_backward_kernel(LossArgs a, const float* __restrict__ row_lse,
                                                                 const float* __restrict__ grad_loss,
                                                                 float* __restrict__ dscores) {
  const int i = blockIdx.x;
  const float* row = a.scores + (int64_t)i * a.ld;
  const int64_t qid_i = a.query_ids ? a.query_ids[i] : 0;
  const int64_t cid_i = a.cand_ids ? a.cand_ids[i] : 0;
  const float lse = row_lse[i];
  const float g = grad_loss ? *grad_loss : 1.f;
  const float scale = a.temperature > 0.f ? g / a.temperature : g;
  for (int j = threadIdx.x; j < a.c; j += 256) {
    float d = 0.f;
    if (!excluded(a, i, j, qid_i, cid_i)) d = (expf(logit(a, row[j], j) - lse) - (j == i ? 1.f : 0.f)) * scale;
    dscores[(int64_t)i * a.c + j] = d;
  }
}

int32_t make_args(gigl_ctx* ctx, LossArgs& a, const float* scores, int64_t ld, int32_t q, int32_t c, float temperature,
                  const float* cand_prob, const int64_t* query_ids, const int64_t* cand_ids) {
  GIGL_REQUIRE(ctx, scores && q >= 1 && c >= q && ld >= c,
               "Number of queries should be less than or equal to number of candidates in a batch (q=%d, c=%d)", q, c);
  a.scores = scores;
  a.ld = ld;
  a.q = q;
  a.c = c;
  a.temperature = temperature;
  a.cand_prob = cand_prob;
  a.query_ids = query_ids;
  a.cand_ids = cand_ids;
  return GIGL_OK;
}


// ---- count-min sketch of candidate ids (the Retrieval task's candidate-sampling correction,
// python/gigl/src/common/models/layers/count_min_sketch.py:11-95 used by task.py:140-205).  The reference keeps a
// depth x width int32 table on the host and walks ids one at a time through Python's hash((item, i)) % width; here
// the table lives in HBM, one thread per (id, row) adds with an atomic, one thread per id takes the minimum.  The
// hash is CPython's (>= 3.8) tuple hash of two ints restated — lanes hash(int) = x mod (2^61 - 1) with the sign kept,
// xxHash-style accumulation — so a table built here equals the reference's for the same ids, cell for cell.
__device__ __forceinline__ uint64_t py_int_hash(int64_t x) {
  const uint64_t P = (1ull << 61) - 1;
  const uint64_t ax = x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x;
  int64_t h = (int64_t)(ax % P);
  if (x < 0) h = -h;
  if (h == -1) h = -2;
  return (uint64_t)h;
}
__device__ __forceinline__ int64_t py_tuple2_hash(int64_t x, int64_t i) {
  const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P5 = 2870177450012600261ull;
  uint64_t acc = P5;
  acc += py_int_hash(x) * P2;
  acc = (acc << 31) | (acc >> 33);
  acc *= P1;
  acc += py_int_hash(i) * P2;
  acc = (acc << 31) | (acc >> 33);
  acc *= P1;
  acc += 2ull ^ (P5 ^ 3527539ull);
  if (acc == ~0ull) return 1546275796;
  return (int64_t)acc;
}
__device__ __forceinline__ int cms_cell(int64_t id, int row, int width) {
  const int64_t h = py_tuple2_hash(id, row);
  int64_t r = h % width;  // Python's %: the result takes the divisor's sign
  if (r < 0) r += width;
  return (int)r;
}
__global__ __launch_bounds__(256) void cms_add_kernel(int32_t* __restrict__ table, int width, int depth,
                                                      const int64_t* __restrict__ ids, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * depth) return;
  const int64_t k = t / depth;
  const int row = (int)(t - k * depth);
  atomicAdd(&table[(int64_t)row * width + cms_cell(ids[k], row, width)], 1);
}
__global__ __launch_bounds__(256) void cms_estimate_kernel(const int32_t* __restrict__ table, int width, int depth,
                                                           const int64_t* __restrict__ ids, int64_t n,
                                                           int64_t* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int64_t id = ids[k];
  int32_t best = 0x7FFFFFFF;
  for (int row = 0; row < depth; ++row) best = min(best, table[(int64_t)row * width + cms_cell(id, row, width)]);
  out[k] = best;
}

}  // namespace

extern "C" {

int32_t gigl_retrieval_loss(gigl_ctx* ctx, const float* scores, int64_t ld, int32_t q, int32_t c, float temperature,
                            const float* cand_prob, const int64_t* query_ids, const int64_t* cand_ids,
                            float* masked_scores, float* row_lse, float* row_loss, float* loss) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  LossArgs a{};
  int32_t rc = make_args(ctx, a, scores, ld, q, c, temperature, cand_prob, query_ids, cand_ids);
  if (rc != GIGL_OK) return rc;
  GIGL_REQUIRE(ctx, row_lse && row_loss && loss, "null output");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(retrieval_rows_kernel, dim3((unsigned)q), dim3(256), 0, ctx->stream, a, masked_scores, row_lse,
                     row_loss);
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(1024), 0, ctx->stream, row_loss, q, loss);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_retrieval_loss_batched(gigl_ctx* ctx, const float* scores, int64_t ld, int64_t batch_stride, int32_t q,
                                    int32_t c, int32_t batches, float temperature, const float* cand_prob,
                                    const int64_t* query_ids, const int64_t* cand_ids, float* row_lse, float* row_loss,
                                    float* loss) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  LossArgs a{};
  int32_t rc = make_args(ctx, a, scores, ld, q, c, temperature, cand_prob, query_ids, cand_ids);
  if (rc != GIGL_OK) return rc;
  GIGL_REQUIRE(ctx, row_lse && row_loss && loss && batches >= 1 && batches <= 65535 && batch_stride >= 0,
               "bad arguments (batches=%d)", batches);
  a.g_scores = batch_stride;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(retrieval_rows_kernel, dim3((unsigned)q, (unsigned)batches), dim3(256), 0, ctx->stream, a,
                     (float*)nullptr, row_lse, row_loss);
  hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)batches), dim3(1024), 0, ctx->stream, row_loss, q, loss);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_retrieval_loss_backward(gigl_ctx* ctx, const float* scores, int64_t ld, int32_t q, int32_t c,
                                     float temperature, const float* cand_prob, const int64_t* query_ids,
                                     const int64_t* cand_ids, const float* row_lse, const float* grad_loss,
                                     float* dscores) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  LossArgs a{};
  int32_t rc = make_args(ctx, a, scores, ld, q, c, temperature, cand_prob, query_ids, cand_ids);
  if (rc != GIGL_OK) return rc;
  GIGL_REQUIRE(ctx, row_lse && dscores, "null argument");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(retrieval_backward_kernel, dim3((unsigned)q), dim3(256), 0, ctx->stream, a, row_lse, grad_loss,
                     dscores);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_cms_add(gigl_ctx* ctx, int32_t* table, int32_t width, int32_t depth, const int64_t* ids, int64_t n) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, table && width > 0 && depth > 0 && n >= 0 && (ids || n == 0), "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n == 0) return GIGL_OK;
  hipLaunchKernelGGL(cms_add_kernel, dim3((unsigned)((n * depth + 255) / 256)), dim3(256), 0, ctx->stream, table, width,
                     depth, ids, n);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_cms_estimate(gigl_ctx* ctx, const int32_t* table, int32_t width, int32_t depth, const int64_t* ids,
                          int64_t n, int64_t* counts) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, table && width > 0 && depth > 0 && n >= 0 && ((ids && counts) || n == 0), "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n == 0) return GIGL_OK;
  hipLaunchKernelGGL(cms_estimate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, table, width,
                     depth, ids, n, counts);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
