// sample.hip — k-hop rooted neighbour sampling on the HBM-resident CSC (frontier-expand kernels).
//
// Replaces (paths relative to the reference root):
//   sampleOnehopSrcNodesUniformly / sampleTwohopSrcNodesUniformly
//       scala/subgraph_sampler/src/main/scala/libs/task/pureSpark/SGSPureSparkV1Task.scala:313-494
//   SamplingStrategy.hashBasedUniformPermutation
//       scala/subgraph_sampler/src/main/scala/libs/task/SamplingStrategy.scala:16-82
//
// Parity contract (mode GIGL_MODE_SPARK_HASH): for a parent with ascending in-neighbour list
// A[1..n] and fanout f the sampled SET is {A[i]} for the f indices with the smallest
// (xxhash64_int32(i + K + seed*counter) as signed int64, i); K = int32-wrapping sum of the ids on the
// path root..parent, counter = hop number (1-based).  n <= f copies the row through.
//
// Kernel shape (gfx950): one 64-lane wavefront per parent slot.  Adjacency is read coalesced, 64
// ids per wave-instruction; each lane hashes its own index.  Top-f selection never leaves the
// wave: the first 64 candidates are bitonic-sorted across lanes (shuffles), later chunks are
// filtered against the current f-th key with one ballot and the (rare) survivors are inserted by
// a one-step lane shift.  The selected indices are re-sorted ascending so that output is the
// canonical (ascending id) form of the set.  Rows with n > HEAVY_DEG are handled by a whole
// workgroup (4 waves striding the row, merge through LDS).
#include "common.h"

namespace {

constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL;
constexpr uint64_t P2 = 0xC2B2AE3D27D4EB4FULL;
constexpr uint64_t P3 = 0x165667B19E3779F9ULL;
constexpr uint64_t P5 = 0x27D4EB2F165667C5ULL;
constexpr uint64_t SPARK_SEED = 42ULL;  // Spark SQL xxhash64() expression seed
constexpr uint64_t SIGN = 0x8000000000000000ULL;

// XXH64 of one little-endian int32 (== Spark XXH64.hashInt), returned with the sign bit flipped so
// that UNSIGNED comparison orders like Spark's signed LongType.
__device__ __forceinline__ uint64_t xxh64_i32_ordered(uint32_t x) {
  uint64_t h = (SPARK_SEED + P5 + 4ULL) ^ ((uint64_t)x * P1);
  h = ((h << 23) | (h >> 41)) * P2 + P3;
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  h *= P3;
  h ^= h >> 32;
  return h ^ SIGN;
}

__device__ __forceinline__ bool less96(uint64_t k1, uint32_t i1, uint64_t k2, uint32_t i2) {
  return k1 < k2 || (k1 == k2 && i1 < i2);
}

// ascending bitonic sort of one (key, idx) per lane across the 64-lane wave
__device__ __forceinline__ void wave_sort96(uint64_t& key, uint32_t& idx, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      uint64_t ok = __shfl_xor(key, j, 64);
      uint32_t oi = __shfl_xor(idx, j, 64);
      bool up = (lane & k) == 0;
      bool lower = (lane & j) == 0;
      bool mine_less = less96(key, idx, ok, oi);
      bool keep_mine = (lower == up) ? mine_less : !mine_less;
      key = keep_mine ? key : ok;
      idx = keep_mine ? idx : oi;
    }
  }
}

__device__ __forceinline__ void wave_sort32(uint32_t& v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      uint32_t o = __shfl_xor(v, j, 64);
      bool up = (lane & k) == 0;
      bool lower = (lane & j) == 0;
      uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
      v = (lower == up) ? mn : mx;
    }
  }
}

// insert wave-uniform candidate (ck, ci) into the ascending per-lane list (key, idx); the last
// lane's element falls off
__device__ __forceinline__ void wave_insert96(uint64_t& key, uint32_t& idx, uint64_t ck, uint32_t ci,
                                              int lane) {
  bool gt = less96(ck, ci, key, idx);  // my element is greater than the candidate
  uint64_t uk = __shfl_up(key, 1, 64);
  uint32_t ui = __shfl_up(idx, 1, 64);
  int gtu = __shfl_up((int)gt, 1, 64);
  if (lane == 0) gtu = 0;
  if (gtu) {
    key = uk;
    idx = ui;
  } else if (gt) {
    key = ck;
    idx = ci;
  }
}

struct ExpandArgs {
  const int64_t* rowptr;
  const uint32_t* col;
  int64_t n_nodes;
  const uint32_t* roots;
  const uint32_t* anc[GIGL_MAX_HOPS];  // nbr arrays of the hops already expanded
  int32_t fan[GIGL_MAX_HOPS];
  int32_t hop;  // 0-based hop being expanded: parents are roots (hop 0) or anc[hop-1] slots
  int64_t n_parents;
  int32_t f;
  int32_t hash_add;  // sampling_seed * (hop+1), int32 wrap
  uint32_t* out_nbr;
  int32_t* out_cnt;
};

// parent node id and K (wrapping int32 sum of the path ids) for parent slot p — wave-uniform
__device__ __forceinline__ void parent_of(const ExpandArgs& a, int64_t p, uint32_t& v, uint32_t& ksum) {
  if (a.hop == 0) {
    v = a.roots[p];
    ksum = v;
    return;
  }
  v = a.anc[a.hop - 1][p];
  uint32_t s = v;
  int64_t q = p;
  for (int l = a.hop - 1; l >= 1; --l) {
    q /= a.fan[l];
    s += a.anc[l - 1][q];
  }
  q /= a.fan[0];
  s += a.roots[q];
  ksum = s;
}

constexpr int64_t HEAVY_DEG = 4096;

// Scan indices [first, deg) in steps of `stride` chunks of 64 starting at chunk c0, maintaining the
// wave's ascending best list (key, idx) whose f-th element is the running threshold.
__device__ __forceinline__ void scan_chunks(uint64_t& key, uint32_t& idx, int64_t deg, int64_t c0,
                                            int64_t cstride, uint32_t base, int f, int lane) {
  uint64_t tk = __shfl(key, f - 1, 64);
  uint32_t ti = __shfl(idx, f - 1, 64);
  int64_t nchunks = (deg + 63) >> 6;
  for (int64_t c = c0; c < nchunks; c += cstride) {
    int64_t i0 = c * 64 + lane;  // 0-based position
    uint32_t i1 = (uint32_t)(i0 + 1);
    bool valid = i0 < deg;
    uint64_t ck = xxh64_i32_ordered(i1 + base);
    bool pass = valid && less96(ck, i1, tk, ti);
    unsigned long long m = __ballot(pass);
    while (m) {
      int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      uint64_t sk = __shfl(ck, src, 64);
      uint32_t si = __shfl(i1, src, 64);
      if (less96(sk, si, tk, ti)) {  // threshold may have tightened since the ballot
        wave_insert96(key, idx, sk, si, lane);
        tk = __shfl(key, f - 1, 64);
        ti = __shfl(idx, f - 1, 64);
      }
    }
  }
}

__global__ __launch_bounds__(256) void expand_kernel(ExpandArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int64_t waves_total = (int64_t)gridDim.x * 4;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave_in_block; p < a.n_parents; p += waves_total) {
    uint32_t v, ksum;
    parent_of(a, p, v, ksum);
    uint32_t* out = a.out_nbr + p * a.f;
    const int f = a.f;
    if (v == GIGL_INVALID || (int64_t)v >= a.n_nodes) {
      if (lane < f) out[lane] = GIGL_INVALID;
      if (lane == 0) a.out_cnt[p] = 0;
      continue;
    }
    const int64_t s = a.rowptr[v];
    const int64_t deg = a.rowptr[v + 1] - s;
    const uint32_t* row = a.col + s;
    if (deg <= f) {  // copy-through: the row is already the canonical ascending set
      if (lane < f) out[lane] = lane < deg ? row[lane] : GIGL_INVALID;
      if (lane == 0) a.out_cnt[p] = (int32_t)deg;
      continue;
    }
    if (deg > HEAVY_DEG) {  // left to expand_heavy_kernel (whole workgroup per parent)
      continue;
    }
    const uint32_t base = ksum + (uint32_t)a.hash_add;  // int32 wrap == uint32 wrap
    // chunk 0: one candidate per lane, full sort
    uint32_t idx = lane < deg ? (uint32_t)(lane + 1) : 0xFFFFFFFFu;
    uint64_t key = lane < deg ? xxh64_i32_ordered(idx + base) : ~0ULL;
    wave_sort96(key, idx, lane);
    if (deg > 64) scan_chunks(key, idx, deg, 1, 1, base, f, lane);
    // lanes [0,f) hold the selected 1-based indices; emit ascending
    uint32_t sel = lane < f ? idx : 0xFFFFFFFFu;
    wave_sort32(sel, lane);
    if (lane < f) out[lane] = row[sel - 1];
    if (lane == 0) a.out_cnt[p] = f;
  }
}

// heavy rows: one workgroup (4 waves) per parent; wave w scans chunks w, w+4, ...; the four best
// lists are merged by wave 0 through LDS.
__global__ __launch_bounds__(256) void expand_heavy_kernel(ExpandArgs a, const int64_t* heavy_list,
                                                           const int32_t* heavy_count) {
  __shared__ uint64_t s_key[4][64];
  __shared__ uint32_t s_idx[4][64];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int32_t n_heavy = *heavy_count;
  for (int32_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {
    const int64_t p = heavy_list[h];
    uint32_t v, ksum;
    parent_of(a, p, v, ksum);
    const int f = a.f;
    const int64_t s = a.rowptr[v];
    const int64_t deg = a.rowptr[v + 1] - s;
    const uint32_t* row = a.col + s;
    const uint32_t base = ksum + (uint32_t)a.hash_add;
    // first chunk of this wave: chunk index w (deg > HEAVY_DEG >= 256 so it is full)
    uint32_t idx = (uint32_t)(w * 64 + lane + 1);
    uint64_t key = xxh64_i32_ordered(idx + base);
    wave_sort96(key, idx, lane);
    scan_chunks(key, idx, deg, w + 4, 4, base, f, lane);
    s_key[w][lane] = key;
    s_idx[w][lane] = idx;
    __syncthreads();
    if (w == 0) {
      uint64_t tk = __shfl(key, f - 1, 64);
      uint32_t ti = __shfl(idx, f - 1, 64);
      for (int ow = 1; ow < 4; ++ow) {
        for (int j = 0; j < f; ++j) {  // other waves' lists are ascending: stop at first non-improving
          uint64_t ck = s_key[ow][j];
          uint32_t ci = s_idx[ow][j];
          if (!less96(ck, ci, tk, ti)) break;
          wave_insert96(key, idx, ck, ci, lane);
          tk = __shfl(key, f - 1, 64);
          ti = __shfl(idx, f - 1, 64);
        }
      }
      uint32_t sel = lane < f ? idx : 0xFFFFFFFFu;
      wave_sort32(sel, lane);
      if (lane < f) a.out_nbr[p * f + lane] = row[sel - 1];
      if (lane == 0) a.out_cnt[p] = f;
    }
    __syncthreads();
  }
}

// compacts the parent slots whose degree exceeds HEAVY_DEG
__global__ void find_heavy_kernel(ExpandArgs a, int64_t* heavy_list, int32_t* heavy_count) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n_parents) return;
  uint32_t v = a.hop == 0 ? a.roots[p] : a.anc[a.hop - 1][p];
  if (v == GIGL_INVALID || (int64_t)v >= a.n_nodes) return;
  int64_t deg = a.rowptr[v + 1] - a.rowptr[v];
  if (deg > HEAVY_DEG && deg > a.f) {
    int32_t at = atomicAdd(heavy_count, 1);
    heavy_list[at] = p;
  }
}

// ---- fast (non-parity) mode: f distinct positions by a counter-based RNG (Floyd's algorithm per
// lane-0 loop would serialise; instead: stratified offsets).  Labelled NOT parity everywhere.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void expand_fast_kernel(ExpandArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int64_t waves_total = (int64_t)gridDim.x * 4;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave_in_block; p < a.n_parents; p += waves_total) {
    uint32_t v, ksum;
    parent_of(a, p, v, ksum);
    uint32_t* out = a.out_nbr + p * a.f;
    const int f = a.f;
    if (v == GIGL_INVALID || (int64_t)v >= a.n_nodes) {
      if (lane < f) out[lane] = GIGL_INVALID;
      if (lane == 0) a.out_cnt[p] = 0;
      continue;
    }
    const int64_t s = a.rowptr[v];
    const int64_t deg = a.rowptr[v + 1] - s;
    const uint32_t* row = a.col + s;
    if (deg <= f) {
      if (lane < f) out[lane] = lane < deg ? row[lane] : GIGL_INVALID;
      if (lane == 0) a.out_cnt[p] = (int32_t)deg;
      continue;
    }
    // stratified: stratum j covers [j*deg/f, (j+1)*deg/f); one uniform draw inside each -> f distinct,
    // ascending positions; reads f ids instead of deg.
    if (lane < f) {
      int64_t lo = (int64_t)lane * deg / f, hi = (int64_t)(lane + 1) * deg / f;
      uint32_t r = mix32(mix32(ksum + (uint32_t)a.hash_add) ^ (uint32_t)(lane * 0x9E3779B9u));
      int64_t pos = lo + (int64_t)(((uint64_t)r * (uint64_t)(hi - lo)) >> 32);
      out[lane] = row[pos];
    }
    if (lane == 0) a.out_cnt[p] = f;
  }
}

}  // namespace

extern "C" {

int32_t gigl_sample_khop(gigl_ctx* ctx, gigl_graph* g, const uint32_t* roots, int32_t b,
                         const int32_t* fanouts, int32_t hops, int32_t sampling_seed, int32_t mode,
                         gigl_tree* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, g && (roots || b == 0) && fanouts && out, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS, "hops must be in [1,%d]", GIGL_MAX_HOPS);
  GIGL_REQUIRE(ctx, b >= 0, "negative batch");
  GIGL_REQUIRE(ctx, mode == GIGL_MODE_SPARK_HASH || mode == GIGL_MODE_FAST, "bad mode %d", mode);
  int64_t parents = b;
  for (int k = 0; k < hops; ++k) {
    if (fanouts[k] < 1 || fanouts[k] > GIGL_MAX_FANOUT)
      return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "fanout[%d]=%d outside [1,%d]", k, fanouts[k],
                       GIGL_MAX_FANOUT);
    GIGL_REQUIRE(ctx, out->nbr[k] && out->cnt[k], "tree buffers for hop %d are null", k);
    parents *= fanouts[k];
    GIGL_REQUIRE(ctx, parents < (int64_t)1 << 31, "tree too large");
  }
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  out->hops = hops;
  out->b = b;
  for (int k = 0; k < hops; ++k) out->fanouts[k] = fanouts[k];
  if (b == 0) return GIGL_OK;

  // scratch: heavy list per hop (worst case every parent) + counter
  int64_t max_parents = b;
  {
    int64_t q = b;
    for (int k = 0; k + 1 < hops; ++k) {
      q *= fanouts[k];
      if (q > max_parents) max_parents = q;
    }
  }
  int32_t rc = gigl_arena_reset(ctx, max_parents * 8 + 256 * 4);
  if (rc != GIGL_OK) return rc;
  int64_t* heavy_list = (int64_t*)gigl_arena_alloc(ctx, max_parents * 8);
  int32_t* heavy_count = (int32_t*)gigl_arena_alloc(ctx, 256);
  if (!heavy_list || !heavy_count) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");

  ExpandArgs a{};
  a.rowptr = g->rowptr;
  a.col = g->col;
  a.n_nodes = g->n;
  a.roots = roots;
  parents = b;
  for (int k = 0; k < hops; ++k) {
    a.hop = k;
    a.n_parents = parents;
    a.f = fanouts[k];
    a.fan[k] = fanouts[k];
    a.hash_add = (int32_t)((uint32_t)sampling_seed * (uint32_t)(k + 1));
    a.out_nbr = out->nbr[k];
    a.out_cnt = out->cnt[k];
    int64_t blocks = (parents + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (mode == GIGL_MODE_FAST) {
      hipLaunchKernelGGL(expand_fast_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    } else {
      {
        gigl_prof_scope ps(ctx, GIGL_K_FIND_HEAVY);
        GIGL_HIP_CHECK(ctx, hipMemsetAsync(heavy_count, 0, 4, ctx->stream));
        hipLaunchKernelGGL(find_heavy_kernel, dim3((unsigned)((parents + 255) / 256)), dim3(256), 0,
                           ctx->stream, a, heavy_list, heavy_count);
      }
      {
        gigl_prof_scope ps(ctx, GIGL_K_EXPAND);
        hipLaunchKernelGGL(expand_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
      }
      {
        gigl_prof_scope ps(ctx, GIGL_K_EXPAND_HEAVY);
        int64_t hb = parents < 2048 ? parents : 2048;
        hipLaunchKernelGGL(expand_heavy_kernel, dim3((unsigned)hb), dim3(256), 0, ctx->stream, a,
                           heavy_list, heavy_count);
      }
    }
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    a.anc[k] = out->nbr[k];
    parents *= fanouts[k];
  }
  return GIGL_OK;
}

int32_t gigl_sample_positives(gigl_ctx* ctx, gigl_graph* g_out, const uint32_t* roots, int32_t b,
                              int32_t f, int32_t sampling_seed, int32_t mode, uint32_t* pos,
                              int32_t* cnt) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, g_out && roots && pos && cnt, "null argument");
  GIGL_REQUIRE(ctx, mode == GIGL_MODE_SPARK_HASH, "positives are parity-mode only");
  if (f < 1 || f > GIGL_MAX_FANOUT)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "num positives %d outside [1,%d]", f, GIGL_MAX_FANOUT);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (b == 0) return GIGL_OK;
  int32_t rc = gigl_arena_reset(ctx, (int64_t)b * 8 + 256 * 4);
  if (rc != GIGL_OK) return rc;
  int64_t* heavy_list = (int64_t*)gigl_arena_alloc(ctx, (int64_t)b * 8);
  int32_t* heavy_count = (int32_t*)gigl_arena_alloc(ctx, 256);
  ExpandArgs a{};
  a.rowptr = g_out->rowptr;
  a.col = g_out->col;
  a.n_nodes = g_out->n;
  a.roots = roots;
  a.hop = 0;
  a.n_parents = b;
  a.f = f;
  a.fan[0] = f;
  // sampleDstNodesUniformly is the third hashBasedUniformPermutation call of the job: _counter = 3
  // (NodeAnchorBasedLinkPredictionTask.scala:171-172)
  a.hash_add = (int32_t)((uint32_t)sampling_seed * 3u);
  a.out_nbr = pos;
  a.out_cnt = cnt;
  GIGL_HIP_CHECK(ctx, hipMemsetAsync(heavy_count, 0, 4, ctx->stream));
  hipLaunchKernelGGL(find_heavy_kernel, dim3((unsigned)((b + 255) / 256)), dim3(256), 0, ctx->stream,
                     a, heavy_list, heavy_count);
  int64_t blocks = (b + 3) / 4;
  hipLaunchKernelGGL(expand_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
  hipLaunchKernelGGL(expand_heavy_kernel, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0,
                     ctx->stream, a, heavy_list, heavy_count);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
