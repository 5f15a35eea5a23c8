// sortscan.h — the library's own device sort / scan / distinct primitives for per-batch paths (typed_plan.hip).
//
// Why not rocPRIM here: its radix sort clears its counters with hipMemsetAsync, which becomes a MEMSET NODE when the
// calling plan is captured into a hipGraph — and memset nodes of a replayed graph were seen to run out of order with
// their neighbouring kernels on this runtime ("Two findings about captured graphs").  Everything below is
// kernels only, sized by capacities, with the real counts on the device: capturable, no host read.
//
//   gigl_radix_sort<K>      stable LSD radix sort of n keys over a list of 8-bit digit positions (a key's bit fields
//                           that can differ: the callers know their bounds), 3 launches per digit:
//                             rs_hist     per tile of 2,048 keys: digit histogram in LDS -> hist[digit][tile]
//                             rs_scan     ONE workgroup: exclusive prefix over (digit, tile) — where each tile's run
//                                         of each digit starts in the output
//                             rs_scatter  per tile: keys re-read in order, rank among the tile's earlier keys of the
//                                         same digit by wave match (8 ballots) + per-wave counters in LDS -> out
//   gigl_unique_compact<K>  the distinct keys of a SORTED array except a pad value, order kept, count on the device:
//                           heads per tile -> rs_scan -> ordered compaction (3 launches; replaces head flags +
//                           DeviceScan + compaction)
//   gigl_exclusive_scan_small  out-of-place exclusive sum by one workgroup (a few thousand entries)
// Integer work, bound by launch latency at the sizes of a batch (10^5..10^6 keys): ~3-5 us per launch.
#pragma once
#include "common.h"

namespace gigl_sort {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;

inline int64_t tiles_of(int64_t n) { return n > 0 ? (n + RS_TILE - 1) / RS_TILE : 1; }
// int32 words of scratch a sort / compaction of up to n keys needs
inline int64_t scratch_words(int64_t n) { return 256 * tiles_of(n) + 1024; }

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const K* __restrict__ in, int64_t n, int shift, int32_t tiles,
                                                             int32_t* __restrict__ hist) {
  __shared__ int32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = base + i * RS_THREADS + threadIdx.x;
    if (p < n) atomicAdd(&h[(uint32_t)(in[p] >> shift) & 0xFFu], 1);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];
}

// in-place exclusive prefix over v[0 .. total) by ONE workgroup of 1,024 threads (a thread owns a contiguous run);
// total_out (may be null) receives the sum of all entries
static __global__ __launch_bounds__(1024) void rs_scan_kernel(int32_t* __restrict__ v, int64_t total, int32_t* __restrict__ total_out) {
  __shared__ int32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t per = (total + 1023) / 1024;
  const int64_t lo = tid * per, hi = lo + per < total ? lo + per : total;
  int32_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += v[i];
  const int32_t incl = gigl_wave_incl_scan(s);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int32_t wbase = 0, all = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int32_t x = wsum[j];
    if (j < w) wbase += x;
    all += x;
  }
  int32_t run = wbase + incl - s;
  for (int64_t i = lo; i < hi; ++i) {
    const int32_t t = v[i];
    v[i] = run;
    run += t;
  }
  if (total_out && tid == 0) *total_out = all;
}

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const K* __restrict__ in, K* __restrict__ out, int64_t n,
                                                                int shift, int32_t tiles, const int32_t* __restrict__ hist) {
  __shared__ int32_t base[256];   // where this tile's run of digit d starts in `out`
  __shared__ int32_t run[256];    // keys of digit d this tile has placed so far
  __shared__ int32_t wc[4][256];  // per wave: count of digit d in this round, then its start within the round
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  base[tid] = hist[(int64_t)tid * tiles + blockIdx.x];
  run[tid] = 0;
  const int64_t tb = (int64_t)blockIdx.x * RS_TILE;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = tb + i * RS_THREADS + tid;
    const bool valid = p < n;
    const K k = valid ? in[p] : (K)0;
    const uint32_t d = (uint32_t)(k >> shift) & 0xFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) wc[j][tid] = 0;
    __syncthreads();
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int rank = __popcll(peers & lt);
    if (valid && rank == 0) wc[w][d] = __popcll(peers);
    __syncthreads();
    {
      const int32_t c0 = wc[0][tid], c1 = wc[1][tid], c2 = wc[2][tid], c3 = wc[3][tid], r = run[tid];
      wc[0][tid] = r;
      wc[1][tid] = r + c0;
      wc[2][tid] = r + c0 + c1;
      wc[3][tid] = r + c0 + c1 + c2;
      run[tid] = r + c0 + c1 + c2 + c3;
    }
    __syncthreads();
    if (valid) out[(int64_t)base[d] + wc[w][d] + rank] = k;
    __syncthreads();
  }
}

template <typename K>
__global__ __launch_bounds__(256) void rs_copy_kernel(const K* __restrict__ in, K* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}

// stable sort of in[0 .. n) into out by the 8-bit digits at bit positions shifts[0 .. n_digits) (least significant
// first).  tmp: n keys; scratch: scratch_words(n) int32.  `in` is left untouched; n_digits == 0 copies.
template <typename K>
inline void gigl_radix_sort(hipStream_t st, const K* in, K* out, K* tmp, int64_t n, const int* shifts, int n_digits,
                            int32_t* scratch) {
  if (n <= 0) return;
  const int32_t tiles = (int32_t)tiles_of(n);
  if (n_digits == 0) {
    hipLaunchKernelGGL((rs_copy_kernel<K>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
    return;
  }
  const K* src = in;
  for (int d = 0; d < n_digits; ++d) {
    K* dst = ((n_digits - 1 - d) & 1) ? tmp : out;  // the last pass lands in `out`
    hipLaunchKernelGGL((rs_hist_kernel<K>), dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, src, n, shifts[d], tiles, scratch);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, st, scratch, (int64_t)256 * tiles, (int32_t*)nullptr);
    hipLaunchKernelGGL((rs_scatter_kernel<K>), dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, src, dst, n, shifts[d], tiles,
                       (const int32_t*)scratch);
    src = dst;
  }
}

// digit positions covering the bit field [lo, lo + bits) appended to shifts; returns the new count
inline int add_digits(int* shifts, int count, int lo, int bits) {
  for (int b = 0; b < bits; b += 8) shifts[count++] = lo + b;
  return count;
}

// ---- batched variant for the sizes of a batch graph (every segment <= 256 tiles = 524,288 keys): up to 8 independent
// sorts share every launch (grid.y = segment), and a pass is TWO launches — the tile histograms are kept tile-major
// ([tile][digit]), so a scatter workgroup finds where its runs start by itself: it adds the tiles' histograms up
// (coalesced 1-KB rows, <= 256 of them) — all of them for the digit totals and their exclusive prefix over the digits,
// the earlier ones for its own offset; no scan launch, no atomics.  64-bit keys are sorted by a VIRTUAL key that packs their two bit fields
// ([0, lo_bits) and [32, ..)) side by side: ceil((lo_bits + hi_bits) / 8) passes instead of one set per field.
constexpr int RS_MAX_SEGS = 8;
constexpr int RS_FUSED_MAX_TILES = 256;
constexpr int RS_MAX_PASSES = 8;

template <typename K>
struct SortSegs {
  const K* in[RS_MAX_SEGS];
  K* out[RS_MAX_SEGS];
  K* tmp[RS_MAX_SEGS];
  int64_t n[RS_MAX_SEGS];
  int32_t lo_bits[RS_MAX_SEGS];  // 64-bit keys: width of the low field (>= 32: the key as it is)
  int32_t nseg;
};
// scratch words per segment: tile histograms (+ slack)
constexpr int64_t RS_SEG_WORDS = 256 * (int64_t)RS_FUSED_MAX_TILES + 256 * RS_MAX_PASSES;

template <typename K>
__device__ __forceinline__ uint32_t vdigit(K k, int lo_bits, int shift) {
  if constexpr (sizeof(K) == 8) {
    if (lo_bits < 32) {
      const unsigned long long v = ((unsigned long long)(k >> 32) << lo_bits) | ((unsigned long long)k & ((1ull << lo_bits) - 1ull));
      return (uint32_t)(v >> shift) & 0xFFu;
    }
  }
  return (uint32_t)(k >> shift) & 0xFFu;
}

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rsb_hist_kernel(SortSegs<K> sg, int src_sel, int shift, int pass,
                                                              int32_t* __restrict__ scratch) {
  __shared__ int32_t h[256];
  const int seg = blockIdx.y;
  const int64_t n = sg.n[seg];
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  if (base >= n) return;
  const K* in = src_sel == 0 ? sg.in[seg] : (src_sel == 1 ? sg.out[seg] : sg.tmp[seg]);
  int32_t* hist = scratch + seg * RS_SEG_WORDS;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int lo = sg.lo_bits[seg];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = base + i * RS_THREADS + threadIdx.x;
    if (p < n) atomicAdd(&h[vdigit<K>(in[p], lo, shift)], 1);
  }
  __syncthreads();
  hist[(int64_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rsb_scatter_kernel(SortSegs<K> sg, int src_sel, int dst_sel, int shift, int pass,
                                                                 const int32_t* __restrict__ scratch) {
  __shared__ int32_t base[256];
  __shared__ int32_t run[256];
  __shared__ int32_t wc[4][256];
  __shared__ int32_t wsum[4];
  const int seg = blockIdx.y;
  const int64_t n = sg.n[seg];
  const int64_t tb = (int64_t)blockIdx.x * RS_TILE;
  if (tb >= n) return;
  const K* in = src_sel == 0 ? sg.in[seg] : (src_sel == 1 ? sg.out[seg] : sg.tmp[seg]);
  K* out = dst_sel == 1 ? sg.out[seg] : sg.tmp[seg];
  const int32_t* hist = scratch + seg * RS_SEG_WORDS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // the tile's keys: all RS_ITEMS loads in flight before anything waits for one
  K keys[RS_ITEMS];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = tb + i * RS_THREADS + tid;
    keys[i] = p < n ? in[p] : (K)0;
  }
  {  // where this tile's run of digit `tid` starts: digits below it (all tiles) + the same digit in the earlier tiles.
     // Every workgroup adds the tiles' histograms up itself (coalesced 1-KB rows, <= 256 of them, four in flight): no
     // totals array, no atomics, no scan launch
    const int n_tiles = (int)((n + RS_TILE - 1) / RS_TILE), me = (int)blockIdx.x;
    int32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int tt = 0;
    for (; tt + 3 < n_tiles; tt += 4) {
      const int32_t v0 = hist[(int64_t)tt * 256 + tid], v1 = hist[(int64_t)(tt + 1) * 256 + tid],
                    v2 = hist[(int64_t)(tt + 2) * 256 + tid], v3 = hist[(int64_t)(tt + 3) * 256 + tid];
      a0 += v0;
      a1 += v1;
      a2 += v2;
      a3 += v3;
      e0 += tt < me ? v0 : 0;
      e1 += tt + 1 < me ? v1 : 0;
      e2 += tt + 2 < me ? v2 : 0;
      e3 += tt + 3 < me ? v3 : 0;
    }
    for (; tt < n_tiles; ++tt) {
      const int32_t v0 = hist[(int64_t)tt * 256 + tid];
      a0 += v0;
      e0 += tt < me ? v0 : 0;
    }
    const int32_t t = a0 + a1 + a2 + a3, earlier = e0 + e1 + e2 + e3;
    const int32_t incl = gigl_wave_incl_scan(t);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int32_t b0 = incl - t + earlier;
    for (int j = 0; j < w; ++j) b0 += wsum[j];
    base[tid] = b0;
    run[tid] = 0;
  }
  const int lo = sg.lo_bits[seg];
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = tb + i * RS_THREADS + tid;
    const bool valid = p < n;
    const K k = keys[i];
    const uint32_t d = vdigit<K>(k, lo, shift);
#pragma unroll
    for (int j = 0; j < 4; ++j) wc[j][tid] = 0;
    __syncthreads();
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int rank = __popcll(peers & lt);
    if (valid && rank == 0) wc[w][d] = __popcll(peers);
    __syncthreads();
    {
      const int32_t c0 = wc[0][tid], c1 = wc[1][tid], c2 = wc[2][tid], c3 = wc[3][tid], r = run[tid];
      wc[0][tid] = r;
      wc[1][tid] = r + c0;
      wc[2][tid] = r + c0 + c1;
      wc[3][tid] = r + c0 + c1 + c2;
      run[tid] = r + c0 + c1 + c2 + c3;
    }
    __syncthreads();
    if (valid) out[(int64_t)base[d] + wc[w][d] + rank] = k;
    __syncthreads();
  }
}

template <typename K>
__global__ __launch_bounds__(256) void rsb_copy_kernel(SortSegs<K> sg) {
  const int seg = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < sg.n[seg]) sg.out[seg][i] = sg.in[seg][i];
}

inline int64_t batch_scratch_words(int nseg) { return (int64_t)nseg * RS_SEG_WORDS; }

// stable sorts of sg.in[s][0 .. n[s]) into sg.out[s] by the low `bits` bits of the (virtual) keys, all segments in the
// same launches.  Every n[s] <= RS_FUSED_MAX_TILES * RS_TILE; scratch: batch_scratch_words(nseg) int32.
template <typename K>
inline void gigl_radix_sort_batch(hipStream_t st, const SortSegs<K>& sg, int bits, int32_t* scratch) {
  int64_t nmax = 0;
  for (int s = 0; s < sg.nseg; ++s) nmax = sg.n[s] > nmax ? sg.n[s] : nmax;
  if (nmax <= 0 || sg.nseg <= 0) return;
  const int passes = (bits + 7) / 8;
  const dim3 grid((unsigned)tiles_of(nmax), (unsigned)sg.nseg);
  if (passes == 0) {
    hipLaunchKernelGGL((rsb_copy_kernel<K>), dim3((unsigned)((nmax + 255) / 256), (unsigned)sg.nseg), dim3(256), 0, st, sg);
    return;
  }
  int src = 0;  // 0 = in, 1 = out, 2 = tmp
  for (int d = 0; d < passes; ++d) {
    const int dst = ((passes - 1 - d) & 1) ? 2 : 1;  // the last pass lands in `out`
    hipLaunchKernelGGL((rsb_hist_kernel<K>), grid, dim3(RS_THREADS), 0, st, sg, src, 8 * d, d, scratch);
    hipLaunchKernelGGL((rsb_scatter_kernel<K>), grid, dim3(RS_THREADS), 0, st, sg, src, dst, 8 * d, d, (const int32_t*)scratch);
    src = dst;
  }
}

// ---- distinct keys of a sorted array (pad excluded), order kept
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void uc_count_kernel(const K* __restrict__ sorted, int64_t n, K pad,
                                                              int32_t* __restrict__ tile_cnt) {
  __shared__ int32_t wsum[4];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = base + i * RS_THREADS + tid;
    if (p < n) {
      const K k = sorted[p];
      c += (k != pad && (p == 0 || sorted[p - 1] != k)) ? 1 : 0;
    }
  }
  const int32_t incl = gigl_wave_incl_scan(c);
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  if (tid == 0) tile_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void uc_write_kernel(const K* __restrict__ sorted, int64_t n, K pad,
                                                              const int32_t* __restrict__ tile_off, K* __restrict__ out,
                                                              int32_t* __restrict__ count) {
  __shared__ int32_t wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // a thread owns RS_ITEMS CONSECUTIVE keys: the compaction keeps the order
  const int64_t p0 = (int64_t)blockIdx.x * RS_TILE + (int64_t)tid * RS_ITEMS;
  K k[RS_ITEMS];
  bool head[RS_ITEMS];
  int32_t c = 0;
  K prev = p0 > 0 && p0 - 1 < n ? sorted[p0 - 1] : pad;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = p0 + i;
    k[i] = p < n ? sorted[p] : pad;
    head[i] = p < n && k[i] != pad && (p == 0 || prev != k[i]);
    c += head[i] ? 1 : 0;
    prev = k[i];
  }
  const int32_t incl = gigl_wave_incl_scan(c);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int32_t off = tile_off[blockIdx.x] + incl - c;
  for (int j = 0; j < w; ++j) off += wsum[j];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i)
    if (head[i]) out[off++] = k[i];
  if (blockIdx.x == gridDim.x - 1 && tid == RS_THREADS - 1) *count = off;
}

// out[0 .. *count) = the distinct keys of sorted[0 .. n) other than pad, in order.  scratch: scratch_words(n) int32.
template <typename K>
inline void gigl_unique_compact(hipStream_t st, const K* sorted, int64_t n, K pad, K* out, int32_t* count, int32_t* scratch) {
  const int32_t tiles = (int32_t)tiles_of(n);
  hipLaunchKernelGGL((uc_count_kernel<K>), dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, sorted, n, pad, scratch);
  hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, st, scratch, (int64_t)tiles, (int32_t*)nullptr);
  hipLaunchKernelGGL((uc_write_kernel<K>), dim3((unsigned)tiles), dim3(RS_THREADS), 0, st, sorted, n, pad,
                     (const int32_t*)scratch, out, count);
}

// batched: up to 8 segments per launch, two launches (tile counts, then ordered compaction with every workgroup summing
// the earlier tiles' counts itself — <= 256 of them)
template <typename K>
struct UniqSegs {
  const K* sorted[RS_MAX_SEGS];
  K* out[RS_MAX_SEGS];
  int32_t* count[RS_MAX_SEGS];
  int64_t n[RS_MAX_SEGS];
  int32_t nseg;
};

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void ucb_count_kernel(UniqSegs<K> sg, K pad, int32_t* __restrict__ scratch) {
  __shared__ int32_t wsum[4];
  const int seg = blockIdx.y, tid = threadIdx.x;
  const int64_t n = sg.n[seg];
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  if (base >= n) return;
  const K* sorted = sg.sorted[seg];
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = base + i * RS_THREADS + tid;
    if (p < n) {
      const K k = sorted[p];
      c += (k != pad && (p == 0 || sorted[p - 1] != k)) ? 1 : 0;
    }
  }
  const int32_t incl = gigl_wave_incl_scan(c);
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  if (tid == 0) scratch[seg * RS_SEG_WORDS + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

template <typename K>
__global__ __launch_bounds__(RS_THREADS) void ucb_write_kernel(UniqSegs<K> sg, K pad, const int32_t* __restrict__ scratch) {
  __shared__ int32_t wsum[4];
  __shared__ int32_t tsum[4];
  const int seg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t n = sg.n[seg];
  if ((int64_t)blockIdx.x * RS_TILE >= n) return;
  const K* sorted = sg.sorted[seg];
  const int32_t* tile_cnt = scratch + seg * RS_SEG_WORDS;
  int32_t e = (int)tid < (int)blockIdx.x ? tile_cnt[tid] : 0;  // (<= 256 tiles: one per thread)
  e = gigl_wave_incl_scan(e);
  if (lane == 63) tsum[w] = e;
  const int64_t p0 = (int64_t)blockIdx.x * RS_TILE + (int64_t)tid * RS_ITEMS;
  K k[RS_ITEMS];
  bool head[RS_ITEMS];
  int32_t c = 0;
  K prev = p0 > 0 && p0 - 1 < n ? sorted[p0 - 1] : pad;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t p = p0 + i;
    k[i] = p < n ? sorted[p] : pad;
    head[i] = p < n && k[i] != pad && (p == 0 || prev != k[i]);
    c += head[i] ? 1 : 0;
    prev = k[i];
  }
  const int32_t incl = gigl_wave_incl_scan(c);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int32_t off = tsum[0] + tsum[1] + tsum[2] + tsum[3] + incl - c;
  for (int j = 0; j < w; ++j) off += wsum[j];
  K* out = sg.out[seg];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i)
    if (head[i]) out[off++] = k[i];
  const int64_t last_tile = (n - 1) / RS_TILE;
  if ((int64_t)blockIdx.x == last_tile && tid == RS_THREADS - 1) *sg.count[seg] = off;
}

template <typename K>
inline void gigl_unique_compact_batch(hipStream_t st, const UniqSegs<K>& sg, K pad, int32_t* scratch) {
  int64_t nmax = 0;
  for (int s = 0; s < sg.nseg; ++s) nmax = sg.n[s] > nmax ? sg.n[s] : nmax;
  if (nmax <= 0 || sg.nseg <= 0) return;
  const dim3 grid((unsigned)tiles_of(nmax), (unsigned)sg.nseg);
  hipLaunchKernelGGL((ucb_count_kernel<K>), grid, dim3(RS_THREADS), 0, st, sg, pad, scratch);
  hipLaunchKernelGGL((ucb_write_kernel<K>), grid, dim3(RS_THREADS), 0, st, sg, pad, (const int32_t*)scratch);
}

// out[i] = in[0] + .. + in[i-1], i < n, by one workgroup
static __global__ __launch_bounds__(1024) void scan_small_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
  __shared__ int32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t per = (n + 1023) / 1024;
  const int64_t lo = tid * per, hi = lo + per < n ? lo + per : n;
  int32_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += in[i];
  const int32_t incl = gigl_wave_incl_scan(s);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int32_t run = incl - s;
  for (int j = 0; j < w; ++j) run += wsum[j];
  for (int64_t i = lo; i < hi; ++i) {
    const int32_t t = in[i];
    out[i] = run;
    run += t;
  }
}
inline void gigl_exclusive_scan_small(hipStream_t st, const int32_t* in, int32_t* out, int64_t n) {
  hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, in, out, n);
}

}  // namespace gigl_sort
