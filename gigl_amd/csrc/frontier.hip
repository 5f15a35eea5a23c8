// frontier.hip — the two device-side ends of the per-hop frontier exchange on a hash-partitioned graph.
//
// The reference's distributed loader asks the owner of every frontier node for its sampled neighbours through one
// RPC per batch and partition (GLT DistNeighborLoader, python/gigl/distributed/distributed_neighborloader.py:26-192;
// owner(v) = v % world, dist_link_prediction_data_partitioner.py:692-695).  Here a hop is ONE equal-split
// all_to_all of fixed-capacity request buckets out and one of the answers back; these kernels fill the buckets and
// scatter the answers into the tree layout without any host-side bookkeeping (no sort, no counts on the host, no
// synchronisation: the step stays a pure stream of device work + two collectives per hop):
//   gigl_frontier_bucket   slot i (node v, path sum K) -> bucket owner(v): position by atomicAdd, the slot number is
//                          remembered locally; empty slots (GIGL_INVALID) are not sent; unused entries stay INVALID
//   (all_to_all of the buckets -> owners run gigl_expand_frontier on what they received -> all_to_all back)
//   gigl_frontier_scatter  answer p of bucket r -> tree slot slot_idx[r][p]: f ids, their count, and the path sums
//                          of the children (K accumulates along the path with uint32 wrap == the reference's int32 add)
// A bucket holds `cap` requests per peer; counts[world] is raised if one overflows (the caller sizes cap = m for
// small worlds, else 1.5 m / world + 512: owner(v) is a uniform hash).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void frontier_bucket_kernel(const uint32_t* __restrict__ nodes,
                                                              const uint32_t* __restrict__ ksums, int64_t m,
                                                              uint32_t world, int64_t cap, uint32_t* __restrict__ req,
                                                              int32_t* __restrict__ slot_idx,
                                                              int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t v = i < m ? nodes[i] : GIGL_INVALID;
  const uint32_t r = v == GIGL_INVALID ? 0xFFFFFFFFu : v % world;
  // One global atomic per (workgroup, owner) instead of one per slot: same-address atomics serialise (~13 ns each)
  // and with a small world every slot of a hop hits one of `world` counters.  Slots take a rank inside the
  // workgroup from LDS counters, one thread per owner reserves the workgroup's range.  (world > 64: per wave — the
  // lanes of an owner are served together by their first lane.)
  const int lane = threadIdx.x & 63;
  int32_t p = 0;
  if (world <= 64) {
    __shared__ int32_t s_cnt[64], s_base[64];
    if (threadIdx.x < world) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int32_t rank = 0;
    if (r != 0xFFFFFFFFu) rank = atomicAdd(&s_cnt[r], 1);
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x] > 0)
      s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (r != 0xFFFFFFFFu) p = s_base[r] + rank;
  } else {
    unsigned long long todo = __ballot(r != 0xFFFFFFFFu);
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const uint32_t r_lead = __shfl(r, lead, 64);
      const unsigned long long same = __ballot(r == r_lead);
      int32_t base = 0;
      if (lane == lead) base = atomicAdd(&counts[r_lead], (int32_t)__popcll(same));
      base = __shfl(base, lead, 64);
      if (r == r_lead) p = base + (int32_t)__popcll(same & ((1ull << lane) - 1ull));
      todo &= ~same;
    }
  }
  if (r == 0xFFFFFFFFu) return;
  if (p >= cap) {
    atomicOr(&counts[world], 1);
    return;
  }
  req[((int64_t)r * 2 + 0) * cap + p] = v;
  req[((int64_t)r * 2 + 1) * cap + p] = ksums ? ksums[i] : v;
  slot_idx[(int64_t)r * cap + p] = (int32_t)i;
}

__global__ __launch_bounds__(256) void frontier_scatter_kernel(const uint32_t* __restrict__ resp,
                                                               const int32_t* __restrict__ slot_idx,
                                                               const int32_t* __restrict__ counts,
                                                               const uint32_t* __restrict__ parent_ksums,
                                                               uint32_t world, int64_t cap, int f,
                                                               uint32_t* __restrict__ out_nbr,
                                                               int32_t* __restrict__ out_cnt,
                                                               uint32_t* __restrict__ child_ksums) {
  // one thread per (bucket entry, j): consecutive threads copy consecutive ids of one answer
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t entry = t / f;
  const int j = (int)(t - entry * f);
  if (entry >= (int64_t)world * cap) return;
  const uint32_t r = (uint32_t)(entry / cap);
  const int64_t p = entry - (int64_t)r * cap;
  int32_t c = counts[r];
  if (c > cap) c = (int32_t)cap;
  if (p >= c) return;
  const int32_t i = slot_idx[entry];
  const uint32_t v = resp[entry * f + j];
  out_nbr[(int64_t)i * f + j] = v;
  if (child_ksums) child_ksums[(int64_t)i * f + j] = v == GIGL_INVALID ? 0u : parent_ksums[i] + v;
  if (j == 0) {
    int n = 0;
    for (int q = 0; q < f; ++q) n += resp[entry * f + q] != GIGL_INVALID ? 1 : 0;
    out_cnt[i] = n;
  }
}

}  // namespace

namespace {
// SamplingOp DAG frontiers: row r of `ids` ([rows][width], GIGL_INVALID = empty) is the concatenation of the node
// sets the parent ops returned for root r; the op's input frontier is their set union
// (GraphDBSampler.getKHopSubgraphForRootNode, scala_spark35/.../sampler/GraphDBSampler.scala:78-82): every later
// occurrence of an id inside a row becomes GIGL_INVALID (the first one stays where it is).  One workgroup per row,
// LDS hash of (id -> first position).
__global__ __launch_bounds__(256) void rows_dedup_kernel(uint32_t* __restrict__ ids, int64_t rows, int32_t width,
                                                         uint32_t cap) {
  extern __shared__ uint32_t s_tab[];  // keys[cap] | pos[cap]
  uint32_t* keys = s_tab;
  uint32_t* pos = s_tab + cap;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    uint32_t* row = ids + r * width;
    for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) {
      keys[i] = GIGL_INVALID;
      pos[i] = GIGL_INVALID;
    }
    __syncthreads();
    for (int32_t q = threadIdx.x; q < width; q += blockDim.x) {
      const uint32_t id = row[q];
      if (id == GIGL_INVALID) continue;
      uint32_t h = (id * 0x9E3779B1u) >> 7 & (cap - 1);
      for (;;) {
        const uint32_t prev = atomicCAS(&keys[h], GIGL_INVALID, id);
        if (prev == GIGL_INVALID || prev == id) break;
        h = (h + 1) & (cap - 1);
      }
      atomicMin(&pos[h], (uint32_t)q);
    }
    __syncthreads();
    for (int32_t q = threadIdx.x; q < width; q += blockDim.x) {
      const uint32_t id = row[q];
      if (id == GIGL_INVALID) continue;
      uint32_t h = (id * 0x9E3779B1u) >> 7 & (cap - 1);
      while (keys[h] != id) h = (h + 1) & (cap - 1);
      if (pos[h] != (uint32_t)q) row[q] = GIGL_INVALID;
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" {

int32_t gigl_frontier_bucket(gigl_ctx* ctx, const uint32_t* nodes, const uint32_t* ksums, int64_t m, int32_t world,
                             int64_t cap, uint32_t* req, int32_t* slot_idx, int32_t* counts) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, (nodes || m == 0) && req && slot_idx && counts, "null argument");
  GIGL_REQUIRE(ctx, world >= 1 && m >= 0 && cap >= 1 && m < ((int64_t)1 << 31) && cap < ((int64_t)1 << 31),
               "bad sizes");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipMemsetAsync(req, 0xFF, (size_t)world * 2 * cap * 4, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipMemsetAsync(counts, 0, (size_t)(world + 1) * 4, ctx->stream));
  if (m > 0)
    hipLaunchKernelGGL(frontier_bucket_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, nodes,
                       ksums, m, (uint32_t)world, cap, req, slot_idx, counts);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_frontier_scatter(gigl_ctx* ctx, const uint32_t* resp, const int32_t* slot_idx, const int32_t* counts,
                              const uint32_t* parent_ksums, int64_t m, int32_t world, int64_t cap, int32_t f,
                              uint32_t* out_nbr, int32_t* out_cnt, uint32_t* child_ksums) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, resp && slot_idx && counts && out_nbr && out_cnt, "null argument");
  GIGL_REQUIRE(ctx, !child_ksums || parent_ksums, "child path sums need the parents' path sums");
  GIGL_REQUIRE(ctx, world >= 1 && m >= 0 && cap >= 1 && f >= 1 && f <= GIGL_MAX_FANOUT, "bad sizes");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m == 0) return GIGL_OK;
  // slots that were not sent (empty parents) have no children
  GIGL_HIP_CHECK(ctx, hipMemsetAsync(out_nbr, 0xFF, (size_t)m * f * 4, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipMemsetAsync(out_cnt, 0, (size_t)m * 4, ctx->stream));
  if (child_ksums) GIGL_HIP_CHECK(ctx, hipMemsetAsync(child_ksums, 0, (size_t)m * f * 4, ctx->stream));
  const int64_t threads = (int64_t)world * cap * f;
  hipLaunchKernelGGL(frontier_scatter_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, resp,
                     slot_idx, counts, parent_ksums, (uint32_t)world, cap, f, out_nbr, out_cnt, child_ksums);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_rows_dedup(gigl_ctx* ctx, uint32_t* ids, int64_t rows, int32_t width) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, rows >= 0 && width >= 0 && (ids || rows * width == 0), "bad arguments");
  if (rows == 0 || width == 0) return GIGL_OK;
  uint32_t cap = 64;
  while (cap < 2u * (uint32_t)width) cap <<= 1;
  if (cap > 16384)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "frontier rows of %d ids exceed the %d the in-LDS set holds", width, 8192);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (cap * 8 > 48 * 1024)
    GIGL_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)rows_dedup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)(cap * 8)));
  const int64_t grid = rows < 256 * 32 ? rows : 256 * 32;
  hipLaunchKernelGGL(rows_dedup_kernel, dim3((unsigned)grid), dim3(256), cap * 8, ctx->stream, ids, rows, width, cap);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
