// graph_build.hip — edge-list ingest: COO (src -> dst) to the resident CSC-by-destination.
//
// Replaces loadEdgeDataframeIntoSparkSql + enforceBidirectionalization
//   scala/subgraph_sampler/src/main/scala/libs/task/pureSpark/SGSPureSparkV1Task.scala:120-286 (:218-258)
// and the per-destination `array_sort(collect_list(_src_node))` of :338-345.
//   undirected: {(min,max)} distinct, UNION (distinct) with the reversed copy == sort+unique of both
//   orientations of every input edge.  Rows come out ascending and duplicate-free.
// Not on the per-batch path (runs once per graph): keys are radix-sorted with rocPRIM.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace {

__global__ void coo_keys_kernel(const uint32_t* src, const uint32_t* dst, int64_t e, int64_t n,
                                int directed, uint64_t* keys, int32_t* bad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= e) return;
  uint32_t s = src[i], d = dst[i];
  if ((int64_t)s >= n || (int64_t)d >= n) {
    atomicAdd(bad, 1);
    s = d = 0;
  }
  if (directed) {
    keys[i] = ((uint64_t)d << 32) | s;
  } else {
    keys[2 * i] = ((uint64_t)d << 32) | s;
    keys[2 * i + 1] = ((uint64_t)s << 32) | d;
  }
}

// ---- sharded ingest (owner(v) = v % world, edges follow the DESTINATION): every input record in one (directed) or both
// (undirected) orientations; an orientation s -> d is kept by the rank that owns d, as row d / world of its shard, the
// source id stays global.  Two passes over the input: count the kept records, then append them (order is irrelevant:
// the keys are sorted afterwards).
__device__ __forceinline__ bool shard_key(uint32_t s, uint32_t d, int rank, int world, uint64_t* key) {
  if ((int)(d % (uint32_t)world) != rank) return false;
  *key = ((uint64_t)(d / (uint32_t)world) << 32) | s;
  return true;
}

template <bool FILL>
__global__ __launch_bounds__(256) void coo_shard_keys_kernel(const uint32_t* __restrict__ src,
                                                             const uint32_t* __restrict__ dst, int64_t e, int64_t n,
                                                             int directed, int rank, int world,
                                                             unsigned long long* __restrict__ count,
                                                             uint64_t* __restrict__ keys, int32_t* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t k[2] = {0, 0};
  int nk = 0;
  if (i < e) {
    const uint32_t s = src[i], d = dst[i];
    if ((int64_t)s >= n || (int64_t)d >= n) {
      if (!FILL) atomicAdd(bad, 1);
    } else {
      if (shard_key(s, d, rank, world, &k[nk])) ++nk;
      if (!directed && shard_key(d, s, rank, world, &k[nk])) ++nk;
    }
  }
  // one atomic per wave: lanes take consecutive places
  int incl = nk;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if ((int)(threadIdx.x & 63) >= off) incl += o;
  }
  const int total = __shfl(incl, 63, 64);
  if (total == 0) return;
  unsigned long long base = 0;
  if ((threadIdx.x & 63) == 63) base = atomicAdd(count, (unsigned long long)total);
  base = __shfl(base, 63, 64);
  if (FILL) {
    const unsigned long long at = base + (unsigned long long)(incl - nk);
    for (int q = 0; q < nk; ++q) keys[at + q] = k[q];
  }
}

// keep_all != 0: every record stays (directed multi-edges: the reference's collect_list keeps repeated (src, dst)
// rows, SGSPureSparkV1Task.scala:337,442); *any_dup reports whether a record repeats
__global__ void uniq_flag_kernel(const uint64_t* sorted, int64_t m, int32_t* flags, int keep_all, int32_t* any_dup) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const bool first = i == 0 || sorted[i] != sorted[i - 1];
  if (!first && keep_all) *any_dup = 1;
  flags[i] = (first || keep_all) ? 1 : 0;
}

__global__ void uniq_write_kernel(const uint64_t* sorted, const int32_t* flags, const int32_t* scan,
                                  int64_t m, uint64_t* ukeys, int64_t* count) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  if (flags[i]) ukeys[scan[i]] = sorted[i];
  if (i == m - 1) *count = (int64_t)scan[i] + flags[i];
}

__global__ void csc_col_kernel(const uint64_t* ukeys, int64_t u, uint32_t* col) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < u) col[i] = (uint32_t)(ukeys[i] & 0xFFFFFFFFu);
}

__global__ void csc_rowptr_kernel(const uint64_t* ukeys, int64_t u, int64_t n, int64_t* rowptr) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > n) return;
  uint64_t target = (uint64_t)v << 32;
  int64_t lo = 0, hi = u;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (ukeys[mid] < target) lo = mid + 1;
    else hi = mid;
  }
  rowptr[v] = lo;
}

__global__ void maxdeg_kernel(const int64_t* rowptr, int64_t n, unsigned long long* out) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long d = v < n ? (unsigned long long)(rowptr[v + 1] - rowptr[v]) : 0ull;
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(d, off, 64);
    d = o > d ? o : d;
  }
  if ((threadIdx.x & 63) == 0 && d) atomicMax(out, d);
}

// does a row repeat an id?  Rows are ascending, so a repeat is two equal neighbours of `col` that are not separated by
// a row boundary (checked by a search of rowptr only when the ids are equal)
__global__ void multi_kernel(const int64_t* rowptr, const uint32_t* col, int64_t n, int64_t e, unsigned long long* out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p + 1 >= e || col[p] != col[p + 1]) return;
  int64_t lo = 0, hi = n;  // first row whose end is > p: the row that holds position p
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid + 1] > p) hi = mid; else lo = mid + 1;
  }
  if (rowptr[lo + 1] > p + 1) out[1] = 1ull;  // position p + 1 is in the same row
}

}  // namespace

int32_t gigl_graph_compute_maxdeg(gigl_ctx* ctx, gigl_graph* g) {
  g->maxdeg = 0;
  if (g->n == 0) return GIGL_OK;
  unsigned long long* d_out = nullptr;
  GIGL_HIP_CHECK(ctx, hipMalloc((void**)&d_out, 16));
  hipMemsetAsync(d_out, 0, 16, ctx->stream);
  hipLaunchKernelGGL(maxdeg_kernel, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, ctx->stream, g->rowptr,
                     g->n, d_out);
  if (g->e > 1)
    hipLaunchKernelGGL(multi_kernel, dim3((unsigned)((g->e + 255) / 256)), dim3(256), 0, ctx->stream, g->rowptr, g->col,
                       g->n, g->e, d_out);
  unsigned long long h[2] = {0, 0};
  hipError_t e = hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_out);
  if (e != hipSuccess) return gigl_fail(ctx, GIGL_E_HIP, "maxdeg reduction failed: %s", hipGetErrorString(e));
  g->maxdeg = (int64_t)h[0];
  g->multi = h[1] != 0;
  return GIGL_OK;
}


namespace {
// position of edge (src -> dst) in `col`: rowptr[dst] + index of src in the ascending row; -1 when absent
__global__ __launch_bounds__(256) void edge_ids_kernel(const int64_t* __restrict__ rowptr,
                                                       const uint32_t* __restrict__ col, int64_t n,
                                                       const uint32_t* __restrict__ src,
                                                       const uint32_t* __restrict__ dst, int64_t m,
                                                       int64_t* __restrict__ eid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t s = src[i], d = dst[i];
  int64_t r = -1;
  if ((int64_t)d < n && s != GIGL_INVALID) {
    int64_t lo = rowptr[d], hi = rowptr[d + 1];
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (col[mid] < s) lo = mid + 1;
      else hi = mid;
    }
    if (lo < rowptr[d + 1] && col[lo] == s) r = lo;
  }
  eid[i] = r;
}

// the same lookup for every edge of a union graph, by position in its `col` buffer: one thread per row
__global__ __launch_bounds__(256) void union_edge_ids_kernel(const int64_t* __restrict__ rowptr,
                                                             const uint32_t* __restrict__ col, int64_t n,
                                                             gigl_union u, int64_t* __restrict__ eid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= u.meta[GIGL_META_N_NODES]) return;
  const uint32_t d = u.nodes[i];
  const int32_t p0 = u.rowptr[i], p1 = u.rowend[i];
  if (p1 <= p0 || (int64_t)d >= n) return;
  const int64_t r0 = rowptr[d], r1 = rowptr[d + 1];
  int64_t from = r0;  // the union row is ascending in LOCAL ids, not in global ids: search the whole row each time
  for (int32_t p = p0; p < p1; ++p) {
    const uint32_t s = u.nodes[u.col[p]];
    int64_t lo = from, hi = r1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (col[mid] < s) lo = mid + 1;
      else hi = mid;
    }
    eid[p] = (lo < r1 && col[lo] == s) ? lo : -1;
  }
}
}  // namespace

extern "C" int32_t gigl_graph_build_from_coo(gigl_ctx* ctx, int64_t n, int64_t e, const uint32_t* src,
                                             const uint32_t* dst, int32_t loc, int32_t is_directed,
                                             gigl_graph** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, n >= 0 && e >= 0 && (e == 0 || (src && dst)), "bad COO arguments");
  GIGL_REQUIRE(ctx, n < (int64_t)GIGL_INVALID, "node ids must fit uint32");
  GIGL_REQUIRE(ctx, loc == GIGL_LOC_HOST || loc == GIGL_LOC_DEVICE, "bad loc %d", loc);
  const int64_t m = is_directed ? e : 2 * e;
  if (m >= ((int64_t)1 << 31))
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "more than 2^31 directed edge records in one shard");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;

  gigl_graph* g = new (std::nothrow) gigl_graph();
  if (!g) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  g->ctx = ctx;
  g->n = n;
  struct Tmp {
    void* p[8] = {nullptr};
    int k = 0;
    ~Tmp() { for (int i = 0; i < k; ++i) if (p[i]) hipFree(p[i]); }
    hipError_t alloc(void** q, size_t bytes) {
      hipError_t r = hipMalloc(q, bytes ? bytes : 16);
      if (r == hipSuccess) p[k++] = *q;
      return r;
    }
  } tmp;
#define BUILD_CHECK(expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      gigl_graph_destroy(g);                                                               \
      return gigl_fail(ctx, _e == hipErrorOutOfMemory ? GIGL_E_OOM : GIGL_E_HIP,           \
                       "%s failed: %s", #expr, hipGetErrorString(_e));                     \
    }                                                                                      \
  } while (0)

  BUILD_CHECK(hipMalloc((void**)&g->rowptr, (size_t)(n + 1) * sizeof(int64_t)));
  if (m == 0) {
    BUILD_CHECK(hipMalloc((void**)&g->col, 16));
    BUILD_CHECK(hipMemsetAsync(g->rowptr, 0, (size_t)(n + 1) * sizeof(int64_t), st));
    BUILD_CHECK(hipStreamSynchronize(st));
    g->e = 0;
    *out = g;
    return GIGL_OK;
  }
  const uint32_t *dsrc = src, *ddst = dst;
  if (loc == GIGL_LOC_HOST) {
    uint32_t *a, *b;
    BUILD_CHECK(tmp.alloc((void**)&a, (size_t)e * 4));
    BUILD_CHECK(tmp.alloc((void**)&b, (size_t)e * 4));
    BUILD_CHECK(hipMemcpyAsync(a, src, (size_t)e * 4, hipMemcpyHostToDevice, st));
    BUILD_CHECK(hipMemcpyAsync(b, dst, (size_t)e * 4, hipMemcpyHostToDevice, st));
    dsrc = a;
    ddst = b;
  }
  uint64_t *keys, *sorted;
  int32_t *flags, *scan, *bad;
  int64_t* count;
  BUILD_CHECK(tmp.alloc((void**)&keys, (size_t)m * 8));
  BUILD_CHECK(tmp.alloc((void**)&sorted, (size_t)m * 8));
  BUILD_CHECK(tmp.alloc((void**)&flags, (size_t)m * 4));
  BUILD_CHECK(tmp.alloc((void**)&scan, (size_t)m * 4));
  BUILD_CHECK(tmp.alloc((void**)&bad, 256));
  count = (int64_t*)((char*)bad + 64);
  BUILD_CHECK(hipMemsetAsync(bad, 0, 256, st));
  const int TB = 256;
  auto grid = [&](int64_t c) { return dim3((unsigned)((c + TB - 1) / TB)); };
  hipLaunchKernelGGL(coo_keys_kernel, grid(e), dim3(TB), 0, st, dsrc, ddst, e, n, is_directed ? 1 : 0,
                     keys, bad);
  int key_bits = 33;
  while (key_bits < 64 && (1LL << (key_bits - 32)) < n) ++key_bits;
  size_t t1 = 0, t2 = 0;
  hipcub::DeviceRadixSort::SortKeys((void*)nullptr, t1, keys, sorted, (int)m, 0, key_bits, st);
  hipcub::DeviceScan::ExclusiveSum((void*)nullptr, t2, flags, scan, (int)m, st);
  size_t tb = t1 > t2 ? t1 : t2;
  void* work;
  BUILD_CHECK(tmp.alloc(&work, tb));
  BUILD_CHECK(hipcub::DeviceRadixSort::SortKeys(work, tb, keys, sorted, (int)m, 0, key_bits, st));
  const int keep_all = is_directed == 2 ? 1 : 0;
  hipLaunchKernelGGL(uniq_flag_kernel, grid(m), dim3(TB), 0, st, sorted, m, flags, keep_all, bad + 1);
  BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(work, tb, flags, scan, (int)m, st));
  // unique keys overwrite `keys`
  hipLaunchKernelGGL(uniq_write_kernel, grid(m), dim3(TB), 0, st, sorted, flags, scan, m, keys, count);
  int64_t h_count = 0;
  int32_t h_bad = 0, h_dup = 0;
  BUILD_CHECK(hipMemcpyAsync(&h_count, count, 8, hipMemcpyDeviceToHost, st));
  BUILD_CHECK(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st));
  BUILD_CHECK(hipMemcpyAsync(&h_dup, bad + 1, 4, hipMemcpyDeviceToHost, st));
  BUILD_CHECK(hipStreamSynchronize(st));
  (void)h_dup;  // (gigl_graph_compute_maxdeg below looks at the finished rows)
  if (h_bad) {
    gigl_graph_destroy(g);
    return gigl_fail(ctx, GIGL_E_INVALID_ARG, "%d edges reference node ids >= n=%lld", h_bad,
                     (long long)n);
  }
  g->e = h_count;
  BUILD_CHECK(hipMalloc((void**)&g->col, (size_t)(h_count > 0 ? h_count : 1) * 4));
  hipLaunchKernelGGL(csc_col_kernel, grid(h_count), dim3(TB), 0, st, keys, h_count, g->col);
  hipLaunchKernelGGL(csc_rowptr_kernel, grid(n + 1), dim3(TB), 0, st, keys, h_count, n, g->rowptr);
  BUILD_CHECK(hipGetLastError());
  BUILD_CHECK(hipStreamSynchronize(st));
#undef BUILD_CHECK
  {
    int32_t rc = gigl_graph_compute_maxdeg(ctx, g);
    if (rc != GIGL_OK) {
      gigl_graph_destroy(g);
      return rc;
    }
  }
  *out = g;
  return GIGL_OK;
}

// Sharded ingest: the shard of rank `rank` of a graph hash-partitioned over `world` ranks (owner(v) = v % world, edges
// follow the destination: dist_link_prediction_data_partitioner.py:692-695), built straight from the whole edge list:
// both orientations of every record when the graph is undirected (enforceBidirectionalization,
// SGSPureSparkV1Task.scala:218-258), the orientations whose destination this rank owns are kept as row dst / world,
// source ids stay global.  Equal to partitioning the rows of gigl_graph_build_from_coo's result.
extern "C" int32_t gigl_graph_build_shard_from_coo(gigl_ctx* ctx, int64_t n, int32_t rank, int32_t world, int64_t e,
                                                   const uint32_t* src, const uint32_t* dst, int32_t loc,
                                                   int32_t is_directed, gigl_graph** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, n >= 0 && e >= 0 && (e == 0 || (src && dst)), "bad COO arguments");
  GIGL_REQUIRE(ctx, n < (int64_t)GIGL_INVALID, "node ids must fit uint32");
  GIGL_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
  GIGL_REQUIRE(ctx, loc == GIGL_LOC_HOST || loc == GIGL_LOC_DEVICE, "bad loc %d", loc);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t n_rows = n > rank ? (n - rank + world - 1) / world : 0;

  gigl_graph* g = new (std::nothrow) gigl_graph();
  if (!g) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  g->ctx = ctx;
  g->n = n_rows;
  struct Tmp {
    void* p[10] = {nullptr};
    int k = 0;
    ~Tmp() { for (int i = 0; i < k; ++i) if (p[i]) hipFree(p[i]); }
    hipError_t alloc(void** q, size_t bytes) {
      hipError_t r = hipMalloc(q, bytes ? bytes : 16);
      if (r == hipSuccess) p[k++] = *q;
      return r;
    }
  } tmp;
#define BUILD_CHECK(expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      gigl_graph_destroy(g);                                                               \
      return gigl_fail(ctx, _e == hipErrorOutOfMemory ? GIGL_E_OOM : GIGL_E_HIP,           \
                       "%s failed: %s", #expr, hipGetErrorString(_e));                     \
    }                                                                                      \
  } while (0)
  BUILD_CHECK(hipMalloc((void**)&g->rowptr, (size_t)(n_rows + 1) * sizeof(int64_t)));
  const uint32_t *dsrc = src, *ddst = dst;
  if (loc == GIGL_LOC_HOST && e > 0) {
    uint32_t *a, *b;
    BUILD_CHECK(tmp.alloc((void**)&a, (size_t)e * 4));
    BUILD_CHECK(tmp.alloc((void**)&b, (size_t)e * 4));
    BUILD_CHECK(hipMemcpyAsync(a, src, (size_t)e * 4, hipMemcpyHostToDevice, st));
    BUILD_CHECK(hipMemcpyAsync(b, dst, (size_t)e * 4, hipMemcpyHostToDevice, st));
    dsrc = a;
    ddst = b;
  }
  int32_t* bad;
  BUILD_CHECK(tmp.alloc((void**)&bad, 256));
  unsigned long long* cnt = (unsigned long long*)((char*)bad + 64);
  int64_t* count = (int64_t*)((char*)bad + 128);
  BUILD_CHECK(hipMemsetAsync(bad, 0, 256, st));
  const int TB = 256;
  auto grid = [&](int64_t c) { return dim3((unsigned)((c + TB - 1) / TB)); };
  unsigned long long h_kept = 0;
  int32_t h_bad = 0;
  if (e > 0) {
    hipLaunchKernelGGL(coo_shard_keys_kernel<false>, grid(e), dim3(TB), 0, st, dsrc, ddst, e, n, is_directed ? 1 : 0,
                       (int)rank, (int)world, cnt, (uint64_t*)nullptr, bad);
    BUILD_CHECK(hipMemcpyAsync(&h_kept, cnt, 8, hipMemcpyDeviceToHost, st));
    BUILD_CHECK(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st));
    BUILD_CHECK(hipStreamSynchronize(st));
  }
  if (h_bad) {
    gigl_graph_destroy(g);
    return gigl_fail(ctx, GIGL_E_INVALID_ARG, "%d edges reference node ids >= n=%lld", h_bad, (long long)n);
  }
  const int64_t m = (int64_t)h_kept;
  if (m >= ((int64_t)1 << 31)) {
    gigl_graph_destroy(g);
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "more than 2^31 directed edge records in one shard");
  }
  if (m == 0) {
    BUILD_CHECK(hipMalloc((void**)&g->col, 16));
    BUILD_CHECK(hipMemsetAsync(g->rowptr, 0, (size_t)(n_rows + 1) * sizeof(int64_t), st));
    BUILD_CHECK(hipStreamSynchronize(st));
    g->e = 0;
    *out = g;
    return GIGL_OK;
  }
  uint64_t *keys, *sorted;
  int32_t *flags, *scan;
  BUILD_CHECK(tmp.alloc((void**)&keys, (size_t)m * 8));
  BUILD_CHECK(tmp.alloc((void**)&sorted, (size_t)m * 8));
  BUILD_CHECK(tmp.alloc((void**)&flags, (size_t)m * 4));
  BUILD_CHECK(tmp.alloc((void**)&scan, (size_t)m * 4));
  BUILD_CHECK(hipMemsetAsync(cnt, 0, 8, st));
  hipLaunchKernelGGL(coo_shard_keys_kernel<true>, grid(e), dim3(TB), 0, st, dsrc, ddst, e, n, is_directed ? 1 : 0,
                     (int)rank, (int)world, cnt, keys, bad);
  int key_bits = 33;
  while (key_bits < 64 && (1LL << (key_bits - 32)) < n_rows) ++key_bits;
  size_t t1 = 0, t2 = 0;
  hipcub::DeviceRadixSort::SortKeys((void*)nullptr, t1, keys, sorted, (int)m, 0, key_bits, st);
  hipcub::DeviceScan::ExclusiveSum((void*)nullptr, t2, flags, scan, (int)m, st);
  size_t tb = t1 > t2 ? t1 : t2;
  void* work;
  BUILD_CHECK(tmp.alloc(&work, tb));
  BUILD_CHECK(hipcub::DeviceRadixSort::SortKeys(work, tb, keys, sorted, (int)m, 0, key_bits, st));
  const int keep_all = is_directed == 2 ? 1 : 0;
  hipLaunchKernelGGL(uniq_flag_kernel, grid(m), dim3(TB), 0, st, sorted, m, flags, keep_all, bad + 1);
  BUILD_CHECK(hipcub::DeviceScan::ExclusiveSum(work, tb, flags, scan, (int)m, st));
  hipLaunchKernelGGL(uniq_write_kernel, grid(m), dim3(TB), 0, st, sorted, flags, scan, m, keys, count);
  int64_t h_count = 0;
  BUILD_CHECK(hipMemcpyAsync(&h_count, count, 8, hipMemcpyDeviceToHost, st));
  BUILD_CHECK(hipStreamSynchronize(st));
  g->e = h_count;
  BUILD_CHECK(hipMalloc((void**)&g->col, (size_t)(h_count > 0 ? h_count : 1) * 4));
  hipLaunchKernelGGL(csc_col_kernel, grid(h_count), dim3(TB), 0, st, keys, h_count, g->col);
  hipLaunchKernelGGL(csc_rowptr_kernel, grid(n_rows + 1), dim3(TB), 0, st, keys, h_count, n_rows, g->rowptr);
  BUILD_CHECK(hipGetLastError());
  BUILD_CHECK(hipStreamSynchronize(st));
#undef BUILD_CHECK
  {
    int32_t rc = gigl_graph_compute_maxdeg(ctx, g);
    if (rc != GIGL_OK) {
      gigl_graph_destroy(g);
      return rc;
    }
  }
  *out = g;
  return GIGL_OK;
}

extern "C" int32_t gigl_edge_ids(gigl_ctx* ctx, gigl_graph* g, const uint32_t* src, const uint32_t* dst, int64_t m,
                      int64_t* eid) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, g && m >= 0 && (m == 0 || (src && dst && eid)), "null argument");
  if (m == 0) return GIGL_OK;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(edge_ids_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, g->rowptr,
                     g->col, g->n, src, dst, m, eid);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

extern "C" int32_t gigl_union_edge_ids(gigl_ctx* ctx, gigl_graph* g, const gigl_union* u, int64_t* eid) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, g && u && eid && u->meta && u->nodes && u->rowptr && u->rowend && u->col, "null argument");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipMemsetAsync(eid, 0xFF, (size_t)u->cap_edges * 8, ctx->stream));
  hipLaunchKernelGGL(union_edge_ids_kernel, dim3((unsigned)((u->cap_nodes + 255) / 256)), dim3(256), 0, ctx->stream,
                     g->rowptr, g->col, g->n, *u, eid);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}
