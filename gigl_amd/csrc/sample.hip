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
// Kernel shape (gfx950): a hop is two launches — plan_rows_kernel (one THREAD per parent slot: parent, row, hash window,
// where its candidates are -> a 32-byte descriptor) and expand_rows_kernel (one wave per TWO parent slots).  A row
// never hashes its adjacency and never reads it: the ~lambda positions whose hash lies under the row's threshold come
// from the device-wide "threshold lists" table (below) in one or two coalesced chunks, already in position order; the
// wave compacts the survivors into LDS (ballot + mbcnt), ranks them by counting, and the f lanes with rank < f ARE
// the sample in ascending-id order.  Rows the filter cannot settle (too few / too many survivors, a proxy tie at the
// f-th place) take a serial path: candidates inserted into a wave-resident sorted list by DPP lane shifts, 64-bit
// keys when a tie was seen.
#include "common.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

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

// ---- wave-level primitives.  Everything below is VALU/SALU only (DPP lane shifts and
// v_readlane broadcasts): no LDS crossbar (ds_bpermute) round trips on the insertion path.
__device__ __forceinline__ uint32_t dpp_shr1(uint32_t x) {  // lane i <- lane i-1 (whole wave), lane 0 <- 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)x, l);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(x >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t readlane32(uint32_t x, int l) { return __builtin_amdgcn_readlane(x, l); }
// LDS hand-off between the lanes of ONE wave: drain the wave's LDS traffic, keep compiler and lanes in step
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// insert wave-uniform candidate (ck, ci) into the ascending per-lane list (key, idx); the last
// lane's element falls off
__device__ __forceinline__ void wave_insert96(uint64_t& key, uint32_t& idx, uint64_t ck, uint32_t ci) {
  const bool gt = less96(ck, ci, key, idx);  // my element is greater than the candidate
  const uint32_t ulo = dpp_shr1((uint32_t)key), uhi = dpp_shr1((uint32_t)(key >> 32));
  const uint32_t ui = dpp_shr1(idx);
  const uint32_t gtu = dpp_shr1(gt ? 1u : 0u);  // lane 0 receives 0
  if (gtu) {
    key = ((uint64_t)uhi << 32) | ulo;
    idx = ui;
  } else if (gt) {
    key = ck;
    idx = ci;
  }
}

// merge one candidate per lane into the wave's ascending best list; (tk, ti) = current f-th element
__device__ __forceinline__ void merge_candidates(uint64_t& key, uint32_t& idx, uint64_t ck, uint32_t ci,
                                                 bool valid, uint64_t& tk, uint32_t& ti, int f, int lane) {
  bool pass = valid && less96(ck, ci, tk, ti);
  unsigned long long m = __ballot(pass);
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const uint64_t sk = readlane64(ck, src);
    const uint32_t si = readlane32(ci, src);
    if (less96(sk, si, tk, ti)) {  // threshold may have tightened since the ballot
      wave_insert96(key, idx, sk, si);
      tk = readlane64(key, f - 1);
      ti = readlane32(idx, f - 1);
    }
  }
}

// lanes [0, f) hold distinct 1-based positions: write row[pos-1] in ascending position order
__device__ __forceinline__ void emit_sorted(const uint32_t* row, uint32_t idx, int f, int lane, uint32_t* out) {
  int rank = 0;
  for (int l = 0; l < f; ++l) rank += readlane32(idx, l) < idx ? 1 : 0;
  if (lane < f) out[rank] = row[idx - 1];
}

// the same on a row that may repeat ids (ascending, so repeats are adjacent in position order): every id once;
// returns the number of ids written, the rest of the f slots become GIGL_INVALID
__device__ __forceinline__ int emit_sorted_distinct(const uint32_t* row, uint32_t idx, int f, int lane, uint32_t* out) {
  const bool mine = lane < f && idx != 0xFFFFFFFFu;
  const uint32_t val = mine ? row[idx - 1] : GIGL_INVALID;
  bool dup = false;  // an equal id sits at a smaller selected position
  for (int l = 0; l < f; ++l) {
    const uint32_t il = readlane32(idx, l), vl = readlane32(val, l);
    if (mine && vl == val && il < idx) dup = true;
  }
  const unsigned long long keep = __ballot(mine && !dup);
  int rank = 0;
  for (int l = 0; l < f; ++l)
    if ((keep >> l) & 1ull) rank += readlane32(idx, l) < idx ? 1 : 0;
  const int n = (int)__popcll(keep);
  if (lane < f) out[lane] = GIGL_INVALID;
  if (mine && !dup) out[rank] = val;
  return n;
}

// direct hashing of positions i in [i_lo, i_hi] (1-based, inclusive); chunks c0, c0+cstride, ...
// Four chunks are hashed before their merges so the multiply chains overlap.
__device__ __forceinline__ void scan_direct(uint64_t& key, uint32_t& idx, uint64_t& tk, uint32_t& ti,
                                            int64_t i_lo, int64_t i_hi, int64_t c0, int64_t cstride,
                                            uint32_t base, int f, int lane) {
  const int64_t n = i_hi - i_lo + 1;
  const int64_t nchunks = (n + 63) >> 6;
  int64_t c = c0;
  for (; c + 3 * cstride < nchunks; c += 4 * cstride) {
    const int64_t ia = i_lo + c * 64 + lane, ib = ia + cstride * 64, ic = ib + cstride * 64,
                  id = ic + cstride * 64;
    const uint64_t ka = xxh64_i32_ordered((uint32_t)ia + base), kb = xxh64_i32_ordered((uint32_t)ib + base),
                   kc = xxh64_i32_ordered((uint32_t)ic + base), kd = xxh64_i32_ordered((uint32_t)id + base);
    merge_candidates(key, idx, ka, (uint32_t)ia, ia <= i_hi, tk, ti, f, lane);
    merge_candidates(key, idx, kb, (uint32_t)ib, ib <= i_hi, tk, ti, f, lane);
    merge_candidates(key, idx, kc, (uint32_t)ic, ic <= i_hi, tk, ti, f, lane);
    merge_candidates(key, idx, kd, (uint32_t)id, id <= i_hi, tk, ti, f, lane);
  }
  for (; c < nchunks; c += cstride) {
    const int64_t i = i_lo + c * 64 + lane;
    merge_candidates(key, idx, xxh64_i32_ordered((uint32_t)i + base), (uint32_t)i, i <= i_hi, tk, ti, f, lane);
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
  // explicit frontier (hash-partitioned graphs): node ids + their K sums arrive from other ranks; the
  // resident CSC holds only the rows of the nodes this rank owns, row = id / row_div
  const uint32_t* ex_nodes;
  const uint32_t* ex_ksum;
  uint32_t row_div;
  int32_t proxy_drop;  // test knob (env GIGL_SAMPLER_PROXY_BITS): 32 - bits kept by the fast path's proxy keys
  int32_t flat_max;    // rows up to this length read their hash window from the table's flat array
  int32_t multi;       // rows may repeat an id (directed multi-edges): sampled positions are drawn over the multiset,
                       // the ids they hold are written once each (the reference's output is a set of edges)
  // PEER-MAPPED graph shards (gigl_sample_khop_peer; the sharded plan's peer-sampled route): node v's row is row v /
  // peer_world of rank (v % peer_world)'s CSC shard, whose rowptr / col arrays are mapped into this process — the
  // requester expands its own frontier and reads the owners' adjacency where it lives; a descriptor's `s` then carries the
  // rank in its top 16 bits.  peer_world == 0: off (rowptr / col above)
  const int64_t* const* peer_rowptr;
  const uint32_t* const* peer_col;
  uint32_t peer_world;
};

// the adjacency row a descriptor's `s` names
__device__ __forceinline__ const uint32_t* row_of_desc(const ExpandArgs& a, int64_t s) {
  if (a.peer_world) return a.peer_col[(uint64_t)s >> 48] + (s & (((int64_t)1 << 48) - 1));
  return a.col + s;
}

// CSC row of the parent in slot p and K (wrapping int32 sum of the path ids) — wave-uniform.  Slot numbers are
// < 2^31 (checked by the callers): 32-bit divisions (a 64-bit one costs ~150 scalar instructions, and the CU's
// single scalar ALU was as busy as the vector units).
__device__ __forceinline__ void parent_of(const ExpandArgs& a, uint32_t p, uint32_t& v, uint32_t& ksum) {
  if (a.ex_nodes) {
    v = a.ex_nodes[p];
    ksum = a.ex_ksum[p];
    if (v != GIGL_INVALID) v /= a.row_div;
    return;
  }
  if (a.hop == 0) {
    v = a.roots[p];
    ksum = v;
    return;
  }
  if (a.hop == 1) {  // the common two-hop case without the generic loop
    v = a.anc[0][p];
    ksum = v + a.roots[p / (uint32_t)a.fan[0]];
    return;
  }
  v = a.anc[a.hop - 1][p];
  uint32_t s = v;
  uint32_t q = p;
  for (int l = a.hop - 1; l >= 1; --l) {
    q /= (uint32_t)a.fan[l];
    s += a.anc[l - 1][q];
  }
  q /= (uint32_t)a.fan[0];
  s += a.roots[q];
  ksum = s;
}

constexpr int64_t HEAVY_DEG = 4096;

// ------------------------------------------------------------------------------------------
// Threshold lists over the hash sequence.
//
// Every parity-mode query is "the f smallest (g(j), j) for j in [base+1, base+deg]" where
// g(j) = xxhash64_int32(j) is ONE fixed function of the integer j = i + K + seed*counter — the graph,
// the roots and the sampling seed only move the window.  Hash values are uniform, so the f smallest of a window of
// deg elements lie, with overwhelming probability, below T = lambda/deg * 2^32 (lambda = f + 4 sqrt(f) + 4), and only
// ~lambda elements of the window do.  So g is tabulated once per device, by threshold:
//   flat[j]        = top 32 bits of the ordered hash of j ("proxy"), every j of the covered domain, in j order;
//   level l >= 1   = the (j, proxy) pairs of every j whose proxy is < 2^(32-l), ascending j (half of level l-1);
//   off[l][b]      = number of level-l pairs with j < b * 16 * 2^l (where a window starts and ends in the list).
//   memory: 4 + 8 (1 - 2^-24) + 0.25 (+ 0.1 build scratch) = 12.4 B per covered j.
// A row picks the deepest level whose threshold still covers its T (l = clz(T - 1): deg/2^l in [lambda, 2 lambda)),
// reads the pairs between two `off` entries — its window's members of that level plus at most one index block of
// slack at either end, one or two 64-pair chunks, coalesced — and keeps those inside the window and below T.  A short
// row reads flat[base+1 .. base+deg] instead.  Either way the survivors arrive in POSITION order.  No row hashes
// anything, and a row of a million neighbours costs what a row of a hundred does.  Results are bit-identical to the
// direct evaluation (see Sel::finish); windows that leave the covered domain or wrap around 2^32 hash directly.
// ------------------------------------------------------------------------------------------
constexpr int TBL_MAX_LEVELS = 24;
constexpr int TBL_TILE_SHIFT = 10;  // the table is built (and its domain rounded) in tiles of 1024 j
constexpr int TBL_IDX_SHIFT = 4;    // index block of level l = 16 * 2^l consecutive j (16 expected pairs)

struct RangeTable {
  int levels;
  uint64_t dom;          // covered j domain [0, dom), multiple of 1024
  const uint32_t* flat;  // [dom]
  const uint2* lvl[TBL_MAX_LEVELS + 1];     // [l] -> pairs (j, proxy), ascending j; [0] unused
  const uint32_t* off[TBL_MAX_LEVELS + 1];  // [l] -> ceil(dom / (16 * 2^l)) + 1 pair offsets
};

// ------------------------------------------------------------------------------------------
// Per-row selection, in two precisions.
//   FAST  keys are the top 32 bits of the ordered hash ("proxy").  Orders exactly like the 64-bit key
//         whenever two proxies differ; any proxy EQUALITY that could matter raises `tie` and the row is redone by
//         the exact path.
//   exact 64-bit key + position tie-break: SamplingStrategy.scala:63 verbatim (the 64-bit keys of the few
//         candidates under the threshold are hashed on the spot).
// ------------------------------------------------------------------------------------------
enum { SRC_HASH = 0, SRC_FLAT = 1, SRC_LEVEL = 2 };

template <bool FAST>
struct Sel {
  using key_t = typename std::conditional<FAST, uint32_t, uint64_t>::type;
  key_t key, tk;      // serial path: per-lane best key (ascending over lanes), threshold = f-th key
  uint32_t idx, ti;   // per-lane position (serial path: lanes [0, f), by key), threshold position
  bool tie;           // FAST only: a proxy equality was seen -> result not trustworthy
  bool ordered;       // FAST only: the result is in position order: the lanes of `selmask` hold the f positions, ascending
  unsigned long long selmask;
  int f, lane;
  int drop;           // FAST only: low proxy bits discarded (0 in production; tests raise it to force ties)
  uint32_t dmask;     // ~0 << drop

  static __device__ __forceinline__ key_t inf() { return (key_t)~(key_t)0; }
  __device__ __forceinline__ key_t mk(uint64_t ordered_hash) const {
    if constexpr (FAST) return (uint32_t)(ordered_hash >> 32) & dmask;
    else return ordered_hash;
  }
  __device__ __forceinline__ void reset_best() {
    key = tk = inf();
    idx = ti = 0xFFFFFFFFu;
  }
  __device__ __forceinline__ void init(int f_, int lane_, int drop_ = 0) {
    drop = drop_;
    dmask = drop_ >= 32 ? 0u : (0xFFFFFFFFu << drop_);
    reset_best();
    tie = false;
    ordered = false;
    selmask = 0;
    f = f_;
    lane = lane_;
  }
  __device__ __forceinline__ bool lt(key_t k1, uint32_t i1, key_t k2, uint32_t i2) const {
    if constexpr (FAST) return k1 < k2;
    else return less96(k1, i1, k2, i2);
  }
  __device__ __forceinline__ key_t bcast(key_t x, int l) const {
    if constexpr (FAST) return readlane32(x, l);
    else return readlane64(x, l);
  }
  __device__ __forceinline__ void refresh_threshold() {
    tk = bcast(key, f - 1);
    ti = readlane32(idx, f - 1);
  }
  // insert the wave-uniform candidate; the last lane's element falls off
  __device__ __forceinline__ void insert(key_t ck, uint32_t ci) {
    const bool gt = lt(ck, ci, key, idx);  // my element is greater than the candidate
    if constexpr (FAST) {
      if (__ballot(key == ck && idx != 0xFFFFFFFFu)) tie = true;
      const uint32_t uk = dpp_shr1(key), ui = dpp_shr1(idx), gtu = dpp_shr1(gt ? 1u : 0u);
      if (gtu) {
        key = uk;
        idx = ui;
      } else if (gt) {
        key = ck;
        idx = ci;
      }
    } else {
      const uint32_t ulo = dpp_shr1((uint32_t)key), uhi = dpp_shr1((uint32_t)(key >> 32));
      const uint32_t ui = dpp_shr1(idx), gtu = dpp_shr1(gt ? 1u : 0u);
      if (gtu) {
        key = ((uint64_t)uhi << 32) | ulo;
        idx = ui;
      } else if (gt) {
        key = ck;
        idx = ci;
      }
    }
  }
  // serial path: merge one candidate per lane into the best list
  __device__ __forceinline__ void merge(key_t ck, uint32_t ci, bool valid) {
    if constexpr (FAST) {
      if (__ballot(valid && ck == tk && tk != inf())) tie = true;
    }
    unsigned long long m = __ballot(valid && lt(ck, ci, tk, ti));
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const key_t sk = bcast(ck, src);
      const uint32_t si = readlane32(ci, src);
      if constexpr (FAST) {
        if (sk == tk) tie = true;
      }
      if (lt(sk, si, tk, ti)) {  // threshold may have tightened since the ballot
        insert(sk, si);
        refresh_threshold();
      }
    }
  }
  // serial path over the whole row: positions [1, deg], keys hashed on the spot (flat == nullptr) or, FAST only,
  // read from the flat array; 4 chunks ahead of their merges
  __device__ __forceinline__ void scan_row(const uint32_t* flat, int64_t deg, uint32_t base) {
    const int64_t nchunks = (deg + 63) >> 6;
    const uint32_t* src = flat + base;
    auto key_at = [&](int64_t i) -> key_t {
      if constexpr (FAST) {
        if (flat) return i <= deg ? (key_t)(src[i] & dmask) : inf();
      }
      return mk(xxh64_i32_ordered((uint32_t)i + base));
    };
    int64_t c = 0;
    for (; c + 3 < nchunks; c += 4) {
      const int64_t ia = 1 + c * 64 + lane, ib = ia + 64, ic = ib + 64, id = ic + 64;
      const key_t ka = key_at(ia), kb = key_at(ib), kc = key_at(ic), kd = key_at(id);
      merge(ka, (uint32_t)ia, ia <= deg);
      merge(kb, (uint32_t)ib, ib <= deg);
      merge(kc, (uint32_t)ic, ic <= deg);
      merge(kd, (uint32_t)id, id <= deg);
    }
    for (; c < nchunks; ++c) {
      const int64_t i = 1 + c * 64 + lane;
      merge(key_at(i), (uint32_t)i, i <= deg);
    }
  }
  // serial path over the level-l pairs of the window whose proxy is below T (T <= 2^(32-l): every element of the
  // window with a proxy below T is in the list); returns whether f elements were found (then they are the row's f best:
  // everything left out has a larger proxy, hence a larger key)
  __device__ __forceinline__ bool scan_level(const RangeTable& tb, int l, uint32_t T, uint32_t n, uint32_t base) {
    const int gs = TBL_IDX_SHIFT + l;
    const uint32_t* off = tb.off[l];
    const uint2* ent = tb.lvl[l];
    const uint32_t start = off[(base + 1u) >> gs], end = off[((base + n) >> gs) + 1u];
    for (uint32_t e0 = start; e0 < end; e0 += 64u) {
      const uint32_t e = e0 + (uint32_t)lane;
      uint2 x = make_uint2(0u, 0xFFFFFFFFu);
      if (e < end) x = ent[e];
      const uint32_t i = x.x - base;
      const bool pass = e < end && (uint32_t)(i - 1u) < n && x.y < T;
      key_t ck;
      if constexpr (FAST) ck = x.y & dmask;
      else ck = xxh64_i32_ordered(x.x);
      merge(ck, i, pass);
    }
    return readlane32(idx, f - 1) != 0xFFFFFFFFu;
  }
  // FAST only: the tail of the filter selections.  `count` survivors (every element of the window with a proxy below
  // the row's threshold) sit in the wave's LDS scratch lk (proxies) / li (positions), in ascending position order.
  // Every survivor finds its rank among them by counting (keys read back four at a time as LDS broadcasts); the f
  // best of the window are the f best survivors — every non-survivor has a larger proxy — provided there are at
  // least f of them and no proxy tie straddles the f-th place (-> tie, redone exactly).  Too few / too many survivors
  // (about 1 row in 10^3) -> false, the caller runs the serial path.  On success `ordered` is set: the lanes of
  // `selmask` hold the selected positions in ascending order, so the caller's output slot is a popcount.
  __device__ __forceinline__ bool finish(uint32_t count, uint32_t* lk, uint32_t* li) {
    if (count < (uint32_t)f || count > 64u) {
      wave_lds_sync();
      return false;
    }
    if (lane < 7) lk[count + (uint32_t)lane] = 0xFFFFFFFFu;  // padding of the last 8-wide read
    wave_lds_sync();
    const bool valid = (uint32_t)lane < count;
    const uint32_t k = valid ? lk[lane] : 0xFFFFFFFFu;
    const uint32_t i = valid ? li[lane] : 0xFFFFFFFFu;
    uint32_t rank = 0;  // survivors with a strictly smaller proxy
    for (uint32_t j = 0; j < count; j += 8) {
      const uint4 q = *(const uint4*)(lk + j), r = *(const uint4*)(lk + j + 4);  // same address in every lane: broadcasts
      rank += (q.x < k ? 1u : 0u) + (q.y < k ? 1u : 0u) + (q.z < k ? 1u : 0u) + (q.w < k ? 1u : 0u);
      rank += (r.x < k ? 1u : 0u) + (r.y < k ? 1u : 0u) + (r.z < k ? 1u : 0u) + (r.w < k ? 1u : 0u);
    }
    wave_lds_sync();  // (the scratch is free for the next row)
    // Survivors that share a proxy share a rank.  A group of equals that lies wholly inside or wholly outside the best
    // f leaves exactly f lanes with rank < f, and the selected SET is the exact one either way; a group that
    // straddles the boundary — the only tie that matters — puts more than f lanes below f.
    selmask = __ballot(valid && rank < (uint32_t)f);
    if (__popcll(selmask) != f) {
      tie = true;
      return true;
    }
    idx = i;
    ordered = true;
    return true;
  }
  // survivors of one 64-candidate chunk appended to the scratch, order kept
  __device__ __forceinline__ void offer(uint32_t k, uint32_t i, bool pass, uint32_t& count, uint32_t* lk, uint32_t* li) {
    const unsigned long long m = __ballot(pass);
    const uint32_t pos =
        count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (pass && pos < 64u) {
      lk[pos] = k;
      li[pos] = i;
    }
    count += (uint32_t)__popcll(m);
  }
  // FAST only.  The window's proxies in position order, hashed on the spot (SRC_HASH: windows outside the table) or
  // read from the flat array, filtered by T; four chunks' loads (or hashes) in flight before any is consumed.
  template <int SRC>
  __device__ __forceinline__ bool filter_positions(const uint32_t* flat, uint32_t n, uint32_t T, uint32_t base,
                                                   uint32_t* lk, uint32_t* li) {
    const bool all = T == 0xFFFFFFFFu;  // (rows of <= 64 positions)
    uint32_t count = 0;
    const uint32_t nchunks = (n + 63u) >> 6;
    const uint32_t* src = flat + base;  // position i -> j = base + i (inside the table: the planner checked)
    for (uint32_t c = 0; c < nchunks; c += 4) {
      uint32_t kk[4];
#pragma unroll
      for (uint32_t t = 0; t < 4; ++t) {
        const uint32_t i = (c + t) * 64u + (uint32_t)lane + 1u;
        kk[t] = 0xFFFFFFFFu;
        // (a chunk's last lanes may read up to 63 proxies past the window: inside the table's allocation, dropped below)
        if constexpr (SRC == SRC_FLAT) {
          if (c + t < nchunks) kk[t] = src[i];
        } else {
          kk[t] = (uint32_t)(xxh64_i32_ordered(i + base) >> 32);
        }
      }
#pragma unroll
      for (uint32_t t = 0; t < 4; ++t) {
        if (c + t >= nchunks) break;
        const uint32_t i = (c + t) * 64u + (uint32_t)lane + 1u;
        const uint32_t k = kk[t] & dmask;
        offer(k, i, i <= n && (all || k < T), count, lk, li);
      }
    }
    return finish(count, lk, li);
  }
  // FAST only.  The window's members of its level (T <= 2^(32-l)) — `cnt` pairs from `ent`: the run between two index
  // entries — filtered by window and T.
  __device__ __forceinline__ bool filter_level(const uint2* ent, uint32_t cnt, uint32_t n, uint32_t T, uint32_t base,
                                               uint32_t* lk, uint32_t* li) {
    uint32_t count = 0;
    for (uint32_t e0 = 0; e0 < cnt; e0 += 128u) {  // two chunks' loads in flight
      const uint32_t ea = e0 + (uint32_t)lane, eb = ea + 64u;
      uint2 xa = make_uint2(0u, 0xFFFFFFFFu), xb = xa;
      if (ea < cnt) xa = ent[ea];
      if (eb < cnt) xb = ent[eb];
      const uint32_t ia = xa.x - base, ib = xb.x - base;
      offer(xa.y & dmask, ia, (uint32_t)(ia - 1u) < n && xa.y < T, count, lk, li);
      if (e0 + 64u < cnt) offer(xb.y & dmask, ib, (uint32_t)(ib - 1u) < n && xb.y < T, count, lk, li);
    }
    return finish(count, lk, li);
  }
  // serial path (exact keys; rows the filter could not settle; fanouts too large for the filter): the candidates
  // under four times the row's threshold T, best f kept by insertion; in the (practically impossible) case that fewer
  // than f lie below it, the whole row.  Windows outside the table hash every position.
  __device__ __forceinline__ void serial(const RangeTable& tb, uint32_t n, uint32_t base, uint32_t T, bool in_table) {
    reset_best();
    if (!in_table) {
      scan_row(nullptr, (int64_t)n, base);
      return;
    }
    const uint32_t T4 = T >= (1u << 30) ? 0xFFFFFFFFu : 4u * T;
    const int l4 = T4 == 0xFFFFFFFFu ? 0 : min((int)__builtin_clz(T4 - 1u), tb.levels);
    if (l4 > 0) {
      if (scan_level(tb, l4, T4, n, base)) return;
      reset_best();
    }
    if constexpr (FAST) {
      scan_row(tb.flat, (int64_t)n, base);
      if (readlane32(idx, f - 1) == 0xFFFFFFFFu) tie = true;  // (a proxy of all ones never enters the list)
    } else {
      scan_row(nullptr, (int64_t)n, base);
    }
  }
};

// ------------------------------------------------------------------------------------------
// One hop = two launches.
//   plan_rows_kernel    one THREAD per parent slot: parent id and K (the path sums), the CSC row, the hash window and
//                       where its candidates are (threshold T, level, the run of pairs between two index entries):
//                       a chain of three dependent loads that a wave-per-row kernel would sit through serially, here
//                       issued for 64 rows at once; 32-byte descriptor per slot.
//   expand_rows_kernel  one WAVE per parent slot, driven by the descriptor: candidates -> survivors -> f best ->
//                       neighbour ids; what is left of the dependent chain is descriptor -> pairs -> ids.
// ------------------------------------------------------------------------------------------
enum { ROW_INVALID = 0, ROW_COPY = 1, ROW_FLAT = 2, ROW_LEVEL = 3, ROW_HASH = 4, ROW_SERIAL = 5, ROW_SERIAL_HASH = 6,
       ROW_SKIP = 7 };

struct __attribute__((aligned(32))) RowDesc {
  int64_t s;         // first position of the row in col
  uint32_t n;        // row length
  uint32_t base;     // hash offset of the window: position i -> j = base + i
  uint32_t T;        // proxy threshold (2^32 - 1: every position is a candidate)
  uint32_t kind_cnt; // ROW_* in bits 0..2, pair count of a ROW_LEVEL run above
  const void* ptr;   // ROW_LEVEL: first pair of the run; ROW_FLAT: the flat array
};

__device__ __forceinline__ float filter_lambda(int f) { return (float)f + 4.f * __builtin_sqrtf((float)f) + 4.f; }

// Work lists of the expansion pass (round 4).  On a power-law graph most parent slots of the last hop are empty (a
// root with 8 neighbours leaves 17 of its 25 hop-1 slots invalid) and many rows are no longer than the fanout
// (copy-through): a wave per slot pair spent two thirds of the waves of a [25,10] hop on slots with nothing to select.
// The planning pass — a thread per slot — now finishes those rows itself (f stores) and appends only the rows that need
// a selection to a work list; the expansion pass is a persistent grid that walks the lists.  WORK_LISTS lists, each
// with its counter on its own 128-byte line (same-line atomics serialise in L2), one atomic per planning workgroup.
constexpr int WORK_LISTS = 64;
constexpr int WORK_CNT_STRIDE = 32;  // int32 words between counters

struct WorkLists {
  uint32_t* items;  // [WORK_LISTS][cap]
  int32_t* count;   // [WORK_LISTS * WORK_CNT_STRIDE]
  int64_t cap;
  int32_t iters;  // 0: persistent waves; > 0: steps per copy of a wave (expand_rows_kernel)
  int32_t wgs;    // workgroups of ONE copy of the grid
};

__host__ __device__ static inline int64_t work_list_cap(int64_t n_parents) {
  return ((n_parents / 256 + 1 + WORK_LISTS - 1) / WORK_LISTS) * 256;
}
// scratch behind the descriptors of a hop with n parents (the callers size their arena with the widest hop)
static inline int64_t expand_desc_bytes(int64_t n_parents) {
  return (n_parents + 1) * (int64_t)sizeof(RowDesc) + WORK_LISTS * work_list_cap(n_parents) * 4 +
         WORK_LISTS * WORK_CNT_STRIDE * 4 + 512;
}

__global__ void work_counts_zero_kernel(int32_t* count) { count[threadIdx.x * WORK_CNT_STRIDE] = 0; }

__global__ __launch_bounds__(256) void plan_rows_kernel(ExpandArgs a, RangeTable tb, RowDesc* __restrict__ desc,
                                                        int64_t* heavy_list, int32_t* heavy_count, WorkLists wl) {
  __shared__ int32_t s_wave[4];
  __shared__ int32_t s_base;
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  bool real = false;  // the row needs the expansion pass
  if (p <= (uint32_t)a.n_parents) {
    RowDesc d{};
    if (p == (uint32_t)a.n_parents) {  // the slot after the last: the partner of an odd last row
      d.kind_cnt = ROW_SKIP;
      desc[p] = d;
    } else {
      uint32_t v, ksum;
      parent_of(a, p, v, ksum);
      uint32_t* out = a.out_nbr + (int64_t)p * a.f;
      if (v == GIGL_INVALID || (int64_t)v >= a.n_nodes) {  // finished here: an empty row
        for (int j = 0; j < a.f; ++j) out[j] = GIGL_INVALID;
        a.out_cnt[p] = 0;
      } else {
        int64_t deg;
        if (a.peer_world) {
          const uint32_t q = v / a.peer_world, rk = v - q * a.peer_world;
          const int64_t* rp = a.peer_rowptr[rk];
          const int64_t s0 = rp[q];
          deg = rp[q + 1] - s0;
          d.s = ((int64_t)rk << 48) | s0;
        } else {
          d.s = a.rowptr[v];
          deg = a.rowptr[v + 1] - d.s;
        }
        d.n = (uint32_t)deg;
        d.base = ksum + (uint32_t)a.hash_add;  // int32 wrap == uint32 wrap
        if (deg <= a.f && !a.multi) {  // finished here: copy-through, the row is already the canonical ascending set
          const uint32_t* row = row_of_desc(a, d.s);
          for (int j = 0; j < a.f; ++j) out[j] = j < (int)deg ? row[j] : GIGL_INVALID;
          a.out_cnt[p] = (int32_t)deg;
        } else if (deg <= a.f) {
          d.kind_cnt = ROW_COPY;
          desc[p] = d;
          real = true;
        } else {
          const bool in_table = (uint64_t)d.base + (uint64_t)deg < tb.dom;  // also excludes 2^32 wrap-around
          const float lam = filter_lambda(a.f);
          // (any T is a valid filter: the approximate reciprocal is fine)
          d.T = d.n <= 64u ? 0xFFFFFFFFu
                           : (uint32_t)fminf(lam * 4294967296.f * __builtin_amdgcn_rcpf((float)d.n), 4294967040.f);
          uint32_t kind;
          if (!in_table) {
            if (deg > HEAVY_DEG) {  // left to expand_heavy_kernel (workgroup per parent)
              kind = ROW_SKIP;
              if (heavy_list) heavy_list[atomicAdd(heavy_count, 1)] = (int64_t)p;
            } else {
              kind = lam <= 56.f ? ROW_HASH : ROW_SERIAL_HASH;
            }
          } else if (lam > 56.f) {
            kind = ROW_SERIAL;
          } else {
            const int l = min((int)__builtin_clz(d.T - 1u), tb.levels);  // deepest level with 2^(32-l) >= T  (T >= 2)
            if (l == 0 || d.n <= (uint32_t)a.flat_max) {
              kind = ROW_FLAT;
              d.ptr = tb.flat;
            } else {
              const int gs = TBL_IDX_SHIFT + l;
              const uint32_t* off = tb.off[l];
              const uint32_t start = off[(d.base + 1u) >> gs], end = off[((d.base + d.n) >> gs) + 1u];
              if (end - start > 256u) {  // (a row longer than the deepest level is made for)
                kind = ROW_SERIAL;
              } else {
                kind = ROW_LEVEL | ((end - start) << 3);
                d.ptr = tb.lvl[l] + start;
              }
            }
          }
          d.kind_cnt = kind;
          desc[p] = d;
          real = (kind & 7u) != ROW_SKIP;
        }
      }
    }
  }
  // append this workgroup's rows to one of the lists: rank inside the workgroup by ballot, one atomic per workgroup
  const unsigned long long m = __ballot(real);
  if (lane == 0) s_wave[w] = (int32_t)__popcll(m);
  __syncthreads();
  const int list = (int)(blockIdx.x % WORK_LISTS);
  if (threadIdx.x == 0) {
    const int32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = tot ? atomicAdd(&wl.count[list * WORK_CNT_STRIDE], tot) : 0;
  }
  __syncthreads();
  if (real) {
    int32_t pos = s_base + (int32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (int k = 0; k < w; ++k) pos += s_wave[k];
    wl.items[(int64_t)list * wl.cap + pos] = p;
  }
}

// candidates of one row fetched ahead of their use: up to two 64-wide chunks of (proxy, j)
struct Cand {
  uint32_t k0, j0, k1, j1;
  bool ahead;  // (wave-uniform) all of the row's candidates are here
};

__device__ __forceinline__ Cand fetch_candidates(const RowDesc& d, int lane) {
  Cand c;
  c.k0 = c.k1 = 0xFFFFFFFFu;
  c.j0 = c.j1 = d.base;  // position 0: never inside a window
  c.ahead = false;
  const uint32_t kind = d.kind_cnt & 7u, cnt = d.kind_cnt >> 3;
  if (kind == ROW_LEVEL && cnt <= 128u) {
    const uint2* ent = (const uint2*)d.ptr;
    if ((uint32_t)lane < cnt) {
      const uint2 x = ent[lane];
      c.j0 = x.x;
      c.k0 = x.y;
    }
    if ((uint32_t)lane + 64u < cnt) {
      const uint2 x = ent[lane + 64];
      c.j1 = x.x;
      c.k1 = x.y;
    }
    c.ahead = true;
  } else if (kind == ROW_FLAT && d.n <= 128u) {
    const uint32_t* src = (const uint32_t*)d.ptr + d.base;
    c.j0 = d.base + (uint32_t)lane + 1u;
    c.k0 = src[lane + 1];  // (the last lanes may read past the window: inside the table's allocation, dropped later)
    if (d.n > 64u) {
      c.j1 = c.j0 + 64u;
      c.k1 = src[lane + 65];
    }
    c.ahead = true;
  }
  return c;
}

// the filter selection over candidates fetched ahead; false: the row goes to the general path
__device__ __forceinline__ bool select_ahead(Sel<true>& sel, const RowDesc& d, const Cand& c, uint32_t* lk, uint32_t* li) {
  const uint32_t n = d.n, T = d.T, base = d.base;
  const bool all = T == 0xFFFFFFFFu;
  uint32_t count = 0;
  const uint32_t i0 = c.j0 - base, i1 = c.j1 - base;
  sel.offer(c.k0 & sel.dmask, i0, (uint32_t)(i0 - 1u) < n && (all || c.k0 < T), count, lk, li);
  if ((d.kind_cnt & 7u) == ROW_LEVEL ? (d.kind_cnt >> 3) > 64u : n > 64u)
    sel.offer(c.k1 & sel.dmask, i1, (uint32_t)(i1 - 1u) < n && (all || c.k1 < T), count, lk, li);
  return sel.finish(count, lk, li);
}

// Two parent slots per wave: both descriptors, then both rows' candidates, are requested before anything is consumed,
// and the first row's neighbour ids are in flight while the second row is ranked — the kernel is bound by the latency
// of its dependent loads (SQ counters: waves wait 3/4 of their cycles at the 8 waves a SIMD holds), so two rows per
// wave is twice the loads in flight.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void expand_rows_kernel(
    ExpandArgs a, RangeTable tb, const RowDesc* __restrict__ desc, WorkLists wl) {
  __shared__ __attribute__((aligned(16))) uint32_t s_lk[4][72], s_li[4][64];  // per-wave survivor scratch
  const int lane = threadIdx.x & 63;
  // the parent slots and everything in their descriptors are the same for all lanes: say so (readfirstlane), and the
  // per-row control flow below runs on the scalar unit with scalar loads
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int f = a.f;
  uint32_t* lk = s_lk[wave_in_block];
  uint32_t* li = s_li[wave_in_block];
  // persistent grid (a multiple of WORK_LISTS waves): wave g walks list g % WORK_LISTS, two rows per step
  // (wl.iters > 0: the grid is `rounds` copies of the persistent one — copy r of a wave walks steps [r iters, (r + 1) iters) of
  // that wave's sequence and retires, so that other streams' workgroups get wave slots while a hop is expanded)
  const uint32_t wgs_p = wl.iters > 0 ? (uint32_t)wl.wgs : gridDim.x;
  const uint32_t round = blockIdx.x / wgs_p;
  const uint32_t gw = (blockIdx.x % wgs_p) * 4u + (uint32_t)wave_in_block, n_waves = wgs_p * 4u;
  const uint32_t list = gw % WORK_LISTS, stride = n_waves / WORK_LISTS;
  const int32_t n_items = __builtin_amdgcn_readfirstlane(wl.count[list * WORK_CNT_STRIDE]);
  const uint32_t* items = wl.items + (int64_t)list * wl.cap;
  const uint32_t k_first = gw / WORK_LISTS + stride * round * (uint32_t)wl.iters;
  const uint32_t k_end = wl.iters > 0 ? k_first + stride * (uint32_t)wl.iters : 0xFFFFFFFFu;
  for (uint32_t k = k_first; (int32_t)(2u * k) < n_items && k < k_end; k += stride) {
    const uint32_t p0 = __builtin_amdgcn_readfirstlane(items[2u * k]);
    const uint32_t p1 = (int32_t)(2u * k + 1u) < n_items ? __builtin_amdgcn_readfirstlane(items[2u * k + 1u])
                                                         : (uint32_t)a.n_parents;  // (that slot holds a ROW_SKIP descriptor)
    const RowDesc d0 = desc[p0], d1 = desc[p1];
    const Cand c0 = fetch_candidates(d0, lane), c1 = fetch_candidates(d1, lane);
    Sel<true> s0, s1;
    s0.init(f, lane, a.proxy_drop);
    s1.init(f, lane, a.proxy_drop);
    // fast path: ordered selections straight to the output
    const bool done0 = c0.ahead && !a.multi && select_ahead(s0, d0, c0, lk, li) && !s0.tie;
    uint32_t slot0 = 0, val0 = 0;
    bool sel0 = false;
    if (done0) {
      const unsigned long long m = s0.selmask;
      sel0 = (m >> lane) & 1ull;
      slot0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (sel0) val0 = row_of_desc(a, d0.s)[s0.idx - 1];
    }
    const bool done1 = c1.ahead && !a.multi && select_ahead(s1, d1, c1, lk, li) && !s1.tie;
    uint32_t slot1 = 0, val1 = 0;
    bool sel1 = false;
    if (done1) {
      const unsigned long long m = s1.selmask;
      sel1 = (m >> lane) & 1ull;
      slot1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (sel1) val1 = row_of_desc(a, d1.s)[s1.idx - 1];
    }
    if (done0) {
      if (sel0) a.out_nbr[(int64_t)p0 * f + slot0] = val0;
      if (lane == 0) a.out_cnt[p0] = f;
    }
    if (done1) {
      if (sel1) a.out_nbr[(int64_t)p1 * f + slot1] = val1;
      if (lane == 0) a.out_cnt[p1] = f;
    }
    if (done0 && done1) continue;
    // general path: everything else (copies, invalid parents, long runs, windows outside the table, multi-edge graphs,
    // rows the filter could not settle, ties), one row after the other
  #pragma nounroll
    for (int r = 0; r < 2; ++r) {
      if (r == 0 ? done0 : done1) continue;
      const RowDesc d = r == 0 ? d0 : d1;
      const uint32_t p = r == 0 ? p0 : p1;
      const uint32_t kind = d.kind_cnt & 7u;
      if (kind == ROW_SKIP) continue;
      uint32_t* out = a.out_nbr + (int64_t)p * f;
      if (kind == ROW_INVALID) {
        if (lane < f) out[lane] = GIGL_INVALID;
        if (lane == 0) a.out_cnt[p] = 0;
        continue;
      }
      const uint32_t* row = row_of_desc(a, d.s);
      const uint32_t n = d.n, base = d.base, T = d.T;
      if (kind == ROW_COPY) {  // copy-through: the row is already the canonical ascending set
        if (a.multi) {
          const int nw = emit_sorted_distinct(row, (uint32_t)lane < n ? (uint32_t)lane + 1u : 0xFFFFFFFFu, f, lane, out);
          if (lane == 0) a.out_cnt[p] = nw;
          continue;
        }
        if (lane < f) out[lane] = (uint32_t)lane < n ? row[lane] : GIGL_INVALID;
        if (lane == 0) a.out_cnt[p] = (int32_t)n;
        continue;
      }
      const bool in_table = kind != ROW_HASH && kind != ROW_SERIAL_HASH;
      uint32_t sel_idx;
      {
        Sel<true> fast;
        fast.init(f, lane, a.proxy_drop);
        bool done = false;
        if (kind == ROW_LEVEL) done = fast.filter_level((const uint2*)d.ptr, d.kind_cnt >> 3, n, T, base, lk, li);
        else if (kind == ROW_FLAT) done = fast.template filter_positions<SRC_FLAT>((const uint32_t*)d.ptr, n, T, base, lk, li);
        else if (kind == ROW_HASH) done = fast.template filter_positions<SRC_HASH>(nullptr, n, T, base, lk, li);
        if (!done) fast.serial(tb, n, base, T, in_table);
        sel_idx = fast.idx;
        if (fast.tie) {  // a 32-bit proxy tie touched the result (about once per 10^7 rows): redo exactly
          Sel<false> exact;
          exact.init(f, lane);
          exact.serial(tb, n, base, T, in_table);
          sel_idx = exact.idx;
        } else if (fast.ordered) {  // selected lanes hold the positions in ascending order
          const unsigned long long m = fast.selmask;
          const bool sel = (m >> lane) & 1ull;
          const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          const uint32_t val = sel ? row[sel_idx - 1] : GIGL_INVALID;
          if (!a.multi) {
            if (sel) out[slot] = val;
            if (lane == 0) a.out_cnt[p] = f;
            continue;
          }
          // ascending positions of an ascending row: a repeated id sits right after its first copy
          if (sel) lk[slot] = val;
          wave_lds_sync();
          const bool keep = sel && (slot == 0 || lk[slot - 1] != val);
          wave_lds_sync();
          const unsigned long long km = __ballot(keep);
          if (lane < f) out[lane] = GIGL_INVALID;
          if (keep) out[__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u))] = val;
          if (lane == 0) a.out_cnt[p] = (int32_t)__popcll(km);
          continue;
        }
      }
      if (a.multi) {
        const int nw = emit_sorted_distinct(row, sel_idx, f, lane, out);
        if (lane == 0) a.out_cnt[p] = nw;
        continue;
      }
      emit_sorted(row, sel_idx, f, lane, out);
      if (lane == 0) a.out_cnt[p] = f;
    }
    wave_lds_sync();  // (the next pair reuses this wave's scratch)
  }
}

// ---- table construction.  A workgroup owns a tile of 1024 consecutive j; wave w the 256 j from tile*1024 + w*256,
// as 4 rows of 64 (lane = j offset inside the row), so ballots enumerate a level's members in j order.
__global__ __launch_bounds__(256) void table_flat_kernel(uint32_t* flat, uint64_t dom) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < dom) flat[j] = (uint32_t)(xxh64_i32_ordered((uint32_t)j) >> 32);
}

// deepest level that lists a proxy: its leading zeros (level l lists the proxies < 2^(32-l))
__device__ __forceinline__ int level_of(uint32_t proxy, int levels) { return min((int)__clz((int)proxy), levels); }

// per tile and level: number of members.  cnt[(l-1) * n_tiles + tile]
__global__ __launch_bounds__(256) void table_count_kernel(const uint32_t* flat, uint32_t* cnt, uint32_t n_tiles,
                                                          int levels) {
  __shared__ uint32_t s_cnt[TBL_MAX_LEVELS + 1][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x;
  const uint64_t j0 = ((uint64_t)tile << TBL_TILE_SHIFT) + (uint64_t)w * 256 + lane;
  int m[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = level_of(flat[j0 + r * 64], levels);
  for (int l = 1; l <= levels; ++l) {  // (members of level l+1 are members of level l)
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) c += (uint32_t)__popcll(__ballot(m[r] >= l));
    if (lane == 0) s_cnt[l][w] = c;
  }
  __syncthreads();
  const int l = threadIdx.x + 1;
  if (l <= levels) cnt[(uint64_t)(l - 1) * n_tiles + tile] = s_cnt[l][0] + s_cnt[l][1] + s_cnt[l][2] + s_cnt[l][3];
}

// per level (blockIdx.x = l - 1): exclusive scan of the tile counts, in place; total[l] = members of the level
__global__ __launch_bounds__(1024) void table_scan_kernel(uint32_t* cnt, uint32_t n_tiles, uint32_t* total) {
  __shared__ uint32_t s_sum[1024];
  uint32_t* c = cnt + (uint64_t)blockIdx.x * n_tiles;
  const uint32_t span = (n_tiles + 1023u) / 1024u;
  const uint32_t lo = min(threadIdx.x * span, n_tiles), hi = min(lo + span, n_tiles);
  uint32_t s = 0;
  for (uint32_t t = lo; t < hi; ++t) s += c[t];
  s_sum[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // (Hillis-Steele: built once per table)
    const uint32_t v = threadIdx.x >= (unsigned)d ? s_sum[threadIdx.x - d] : 0u;
    __syncthreads();
    s_sum[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = s_sum[threadIdx.x] - s;
  for (uint32_t t = lo; t < hi; ++t) {
    const uint32_t v = c[t];
    c[t] = run;
    run += v;
  }
  if (threadIdx.x == 1023) total[blockIdx.x + 1] = s_sum[1023];
}

struct TableBuild {
  const uint32_t* flat;
  const uint32_t* tile_off;  // scanned counts, [(l-1) * n_tiles + tile]
  const uint32_t* total;     // [l]
  uint2* lvl[TBL_MAX_LEVELS + 1];
  uint32_t* off[TBL_MAX_LEVELS + 1];
  uint32_t n_tiles;
  int levels;
  uint64_t dom;
};

// members written in j order; index entries of the blocks that start inside the tile
__global__ __launch_bounds__(256) void table_fill_kernel(TableBuild t) {
  __shared__ uint32_t s_cnt[TBL_MAX_LEVELS + 1][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t tile = blockIdx.x;
  const uint64_t jw = ((uint64_t)tile << TBL_TILE_SHIFT) + (uint64_t)w * 256;
  uint32_t px[4];
  int m[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    px[r] = t.flat[jw + r * 64 + lane];
    m[r] = level_of(px[r], t.levels);
  }
  for (int l = 1; l <= t.levels; ++l) {
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) c += (uint32_t)__popcll(__ballot(m[r] >= l));
    if (lane == 0) s_cnt[l][w] = c;
  }
  __syncthreads();
  for (int l = 1; l <= t.levels; ++l) {
    uint32_t run = t.tile_off[(uint64_t)(l - 1) * t.n_tiles + tile];  // members before this wave's first j
    for (int ww = 0; ww < w; ++ww) run += s_cnt[l][ww];
    const int gs = TBL_IDX_SHIFT + l;
    if (gs > TBL_TILE_SHIFT && s_cnt[l][w] == 0) continue;  // (index blocks of these levels start at tile starts)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned long long mm = __ballot(m[r] >= l);
      const uint32_t before = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
      const uint64_t j = jw + r * 64 + lane;
      if (m[r] >= l) t.lvl[l][before] = make_uint2((uint32_t)j, px[r]);
      if (gs <= TBL_TILE_SHIFT && (j & ((1ull << gs) - 1)) == 0) t.off[l][j >> gs] = before;
      run += (uint32_t)__popcll(mm);
    }
  }
  if (threadIdx.x == 0) {
    const uint64_t j = (uint64_t)tile << TBL_TILE_SHIFT;
    for (int l = 1; l <= t.levels; ++l) {
      const int gs = TBL_IDX_SHIFT + l;
      if (gs > TBL_TILE_SHIFT && (j & ((1ull << gs) - 1)) == 0)
        t.off[l][j >> gs] = t.tile_off[(uint64_t)(l - 1) * t.n_tiles + tile];
      if (tile == t.n_tiles - 1) t.off[l][(t.dom + (1ull << gs) - 1) >> gs] = t.total[l];  // the sentinel
    }
  }
}

// fallback for rows with deg > HEAVY_DEG whose window is outside the table: one workgroup (4 waves)
// per parent; wave w scans chunks w, w+4, ...; the four best lists are merged by wave 0 through LDS.
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
    uint32_t idx = 0xFFFFFFFFu;
    uint64_t key = ~0ULL;
    {
      uint64_t tk0 = ~0ULL;
      uint32_t ti0 = 0xFFFFFFFFu;
      scan_direct(key, idx, tk0, ti0, 1, deg, w, 4, base, f, lane);
    }
    s_key[w][lane] = key;
    s_idx[w][lane] = idx;
    __syncthreads();
    if (w == 0) {
      uint64_t tk = readlane64(key, f - 1);
      uint32_t ti = readlane32(idx, f - 1);
      for (int ow = 1; ow < 4; ++ow)
        merge_candidates(key, idx, s_key[ow][lane], s_idx[ow][lane], true, tk, ti, f, lane);
      if (a.multi) {
        const int nw = emit_sorted_distinct(row, idx, f, lane, a.out_nbr + p * f);
        if (lane == 0) a.out_cnt[p] = nw;
      } else {
        emit_sorted(row, idx, f, lane, a.out_nbr + p * f);
        if (lane == 0) a.out_cnt[p] = f;
      }
    }
    __syncthreads();
  }
}

// ---- fanouts beyond the wave-resident selection (f > GIGL_FAST_FANOUT = 64, up to GIGL_MAX_FANOUT): one WORKGROUP per
// parent slot.  The reference's numNeighborsToSample is any integer (SGSPureSparkV1Task.scala:313-388); the wave kernels
// above keep one candidate per lane.  Same contract, evaluated directly: the f smallest (xxhash64(i + K + seed*counter),
// i) over the row's positions.  The positions whose ordered hash lies under a threshold T ~ (f + 6 sqrt(f) + 16) / deg
// are collected into LDS (hash values are uniform: between f and 2,048 of them with overwhelming probability; T is
// doubled / halved and the row redone otherwise), sorted by (key, position) — bitonic, in LDS — the first f kept and
// sorted by position: ascending position = ascending id, the order every other path writes.  Rows of <= f neighbours are
// copied through.  Exact by construction (64-bit keys + position tie-break, SamplingStrategy.scala:63).
constexpr int WIDE_CAP = 2048;

// ascending bitonic sort of n <= cap (cap a power of two) (key, idx) pairs in LDS by (key, idx); entries [n, cap) must
// be padded with (~0, ~0) by the caller
__device__ __forceinline__ void wide_bitonic(uint64_t* k, uint32_t* x, int cap, bool by_idx) {
  for (int size = 2; size <= cap; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < cap / 2; t += blockDim.x) {
        const int lo = (t / stride) * (stride << 1) + (t % stride), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint64_t ka = k[lo], kb = k[hi];
        const uint32_t xa = x[lo], xb = x[hi];
        const bool a_gt_b = by_idx ? (xa > xb) : less96(kb, xb, ka, xa);
        if (a_gt_b == up) {
          k[lo] = kb;
          k[hi] = ka;
          x[lo] = xb;
          x[hi] = xa;
        }
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void expand_wide_kernel(ExpandArgs a) {
  __shared__ uint64_t s_key[WIDE_CAP];
  __shared__ uint32_t s_idx[WIDE_CAP];
  __shared__ int32_t s_count;
  __shared__ int32_t s_scan[256];
  const int tid = threadIdx.x;
  const int f = a.f;
  for (int64_t p = blockIdx.x; p < a.n_parents; p += gridDim.x) {
    uint32_t v, ksum;
    parent_of(a, (uint32_t)p, v, ksum);
    uint32_t* out = a.out_nbr + p * f;
    __syncthreads();  // (the previous row's LDS state is dead)
    if (v == GIGL_INVALID || (int64_t)v >= a.n_nodes) {
      for (int j = tid; j < f; j += 256) out[j] = GIGL_INVALID;
      if (tid == 0) a.out_cnt[p] = 0;
      continue;
    }
    const int64_t s = a.rowptr[v];
    const int64_t deg = a.rowptr[v + 1] - s;
    const uint32_t* row = a.col + s;
    const uint32_t base = ksum + (uint32_t)a.hash_add;
    int n_sel;  // positions selected, in s_idx[0 .. n_sel) ascending
    if (deg <= f) {
      for (int j = tid; j < f; j += 256) s_idx[j] = j < deg ? (uint32_t)(j + 1) : 0xFFFFFFFFu;
      n_sel = (int)deg;
      __syncthreads();
    } else {
      const double lam = (double)f + 6.0 * sqrt((double)f) + 16.0;
      double frac = lam / (double)deg;
      for (;;) {
        const uint64_t T = frac >= 1.0 ? ~0ULL : (uint64_t)(frac * 18446744073709551616.0);
        if (tid == 0) s_count = 0;
        __syncthreads();
        for (int64_t i = 1 + tid; i <= deg; i += 256) {
          const uint64_t key = xxh64_i32_ordered((uint32_t)i + base);
          if (key < T || T == ~0ULL) {
            const int slot = atomicAdd(&s_count, 1);
            if (slot < WIDE_CAP) {
              s_key[slot] = key;
              s_idx[slot] = (uint32_t)i;
            }
          }
        }
        __syncthreads();
        const int count = s_count;
        __syncthreads();
        if (count > WIDE_CAP) {
          frac *= 0.5;
          continue;
        }
        if (count < f) {
          frac *= 2.0;
          continue;
        }
        for (int j = count + tid; j < WIDE_CAP; j += 256) {
          s_key[j] = ~0ULL;
          s_idx[j] = 0xFFFFFFFFu;
        }
        int cap = 64;
        while (cap < count) cap <<= 1;
        wide_bitonic(s_key, s_idx, cap, false);  // by (key, position): the first f are the sample
        int cap2 = 64;
        while (cap2 < f) cap2 <<= 1;
        for (int j = f + tid; j < cap2; j += 256) {  // (cap2 <= cap: everything past f is dropped)
          s_key[j] = ~0ULL;
          s_idx[j] = 0xFFFFFFFFu;
        }
        wide_bitonic(s_key, s_idx, cap2, true);  // by position
        break;
      }
      n_sel = f;
    }
    // ids of the selected positions in position order; rows that repeat ids (directed multi-edges) write every id once
    int kept = 0;
    if (!a.multi) {
      for (int j = tid; j < f; j += 256) out[j] = j < n_sel ? row[s_idx[j] - 1] : GIGL_INVALID;
      kept = n_sel;
    } else {
      int run = 0;  // (f <= 1024: four slots per thread, consecutive)
      const int per = (f + 255) / 256;
      const int j0 = tid * per;
      for (int j = j0; j < j0 + per && j < n_sel; ++j) {
        const uint32_t val = row[s_idx[j] - 1];
        run += (j == 0 || row[s_idx[j - 1] - 1] != val) ? 1 : 0;
      }
      s_scan[tid] = run;
      __syncthreads();
      int off = 0;
      for (int t = 0; t < tid; ++t) off += s_scan[t];
      int total = 0;
      for (int t = 0; t < 256; ++t) total += s_scan[t];
      for (int j = j0; j < j0 + per && j < n_sel; ++j) {
        const uint32_t val = row[s_idx[j] - 1];
        if (j == 0 || row[s_idx[j - 1] - 1] != val) out[off++] = val;
      }
      for (int j = total + tid; j < f; j += 256) out[j] = GIGL_INVALID;
      kept = total;
    }
    if (tid == 0) a.out_cnt[p] = kept;
  }
}

// The table is a function of the integer axis alone (not of the graph, the roots or the seed), so ONE table per device
// serves every ctx of the process: ctxs hold a reference to the device's current table; a request beyond its domain
// builds a larger one, which replaces it in the registry, and the old one is freed when the last ctx that still
// points to it has moved on (a ctx only moves on after synchronising its own stream, so no kernel reads freed memory).
struct TableOwner {
  RangeTable t{};
  void* mem = nullptr;   // flat array, index arrays
  void* mem2 = nullptr;  // level lists
  int device = 0;
  int refs = 0;
};

std::mutex g_table_mu;
TableOwner* g_device_table[64] = {nullptr};

void table_unref(TableOwner* own) {  // g_table_mu held
  if (!own || --own->refs > 0) return;
  if (g_device_table[own->device] == own) g_device_table[own->device] = nullptr;
  if (own->mem) hipFree(own->mem);
  if (own->mem2) hipFree(own->mem2);
  delete own;
}

// Largest j domain a table may cover.  12.4 B per j: at least 2^30 j (13.3 GB, as sized for a 2-hop MAG240M job) and
// up to the whole uint32 axis (53 GB) when a quarter of the device's free memory allows it — a scale-30 RMAT job
// (windows up to ~3.2e9) then never leaves the table path; windows beyond it are hashed directly (same result).
// GIGL_TABLE_MAX_J overrides (tests).
uint64_t table_cap_j() {
  if (const char* e = getenv("GIGL_TABLE_MAX_J")) {
    const unsigned long long v = strtoull(e, nullptr, 10);
    if (v >= 64) return (uint64_t)v;
  }
  uint64_t cap = 1ull << 30;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    const uint64_t by_mem = (uint64_t)(free_b / 4) / 13;
    if (by_mem > cap) cap = by_mem;
  }
  return cap < (1ull << 32) ? cap : (1ull << 32);
}

// make ctx->sampler_table a table that covers [0, want_dom)
int32_t ensure_table(gigl_ctx* ctx, uint64_t want_dom) {
  constexpr uint64_t S0 = 1ull << TBL_TILE_SHIFT;
  constexpr uint64_t DOM_CAP = 1ull << 32;  // j is a uint32
  if (want_dom > DOM_CAP) want_dom = DOM_CAP;
  want_dom = (want_dom + S0 - 1) / S0 * S0;
  TableOwner* own = (TableOwner*)ctx->sampler_table;
  if (own && own->t.dom >= want_dom) return GIGL_OK;
  std::lock_guard<std::mutex> lock(g_table_mu);
  if (own) {  // this ctx moves on to a larger table: its in-flight kernels still read the old one
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    table_unref(own);
    ctx->sampler_table = nullptr;
  }
  const int dev = ctx->device & 63;
  TableOwner* shared = g_device_table[dev];
  if (shared && shared->t.dom >= want_dom) {
    ++shared->refs;
    ctx->sampler_table = shared;
    return GIGL_OK;
  }
  own = new TableOwner();
  own->device = dev;
  // grow geometrically so that slightly larger requests do not rebuild
  uint64_t dom = want_dom + want_dom / 4;
  if (dom > DOM_CAP) dom = DOM_CAP;
  dom = dom / S0 * S0;
  if (dom < S0) dom = S0;
  const int L = TBL_MAX_LEVELS;
  const uint32_t n_tiles = (uint32_t)(dom >> TBL_TILE_SHIFT);
  // first allocation: flat | per-level index arrays | tile counts (build scratch) | level totals
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off_elems[TBL_MAX_LEVELS + 1] = {0}, off_at[TBL_MAX_LEVELS + 1] = {0};
  size_t bytes = up((size_t)dom * 4);
  for (int l = 1; l <= L; ++l) {
    const int gs = TBL_IDX_SHIFT + l;
    off_elems[l] = (size_t)((dom + (1ull << gs) - 1) >> gs) + 1;
    off_at[l] = bytes;
    bytes = up(bytes + off_elems[l] * 4);
  }
  const size_t cnt_at = bytes;
  bytes = up(bytes + (size_t)L * n_tiles * 4);
  const size_t total_at = bytes;
  bytes += 256;
  auto fail_build = [&](int32_t code, const char* what, hipError_t e) {
    if (own->mem) hipFree(own->mem);
    if (own->mem2) hipFree(own->mem2);
    delete own;
    return gigl_fail(ctx, code, "%s (hash threshold table over %llu j): %s", what, (unsigned long long)dom,
                     hipGetErrorString(e));
  };
  hipError_t e = hipMalloc(&own->mem, bytes);
  if (e != hipSuccess) {
    own->mem = nullptr;
    return fail_build(GIGL_E_OOM, "hipMalloc failed", e);
  }
  char* m0 = (char*)own->mem;
  uint32_t* flat = (uint32_t*)m0;
  uint32_t* cnt = (uint32_t*)(m0 + cnt_at);
  uint32_t* total = (uint32_t*)(m0 + total_at);
  hipLaunchKernelGGL(table_flat_kernel, dim3((unsigned)((dom + 255) / 256)), dim3(256), 0, ctx->stream, flat, dom);
  hipLaunchKernelGGL(table_count_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, flat, cnt, n_tiles, L);
  hipLaunchKernelGGL(table_scan_kernel, dim3((unsigned)L), dim3(1024), 0, ctx->stream, cnt, n_tiles, total);
  uint32_t h_total[TBL_MAX_LEVELS + 1] = {0};
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(h_total, total, sizeof(h_total), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail_build(GIGL_E_HIP, "counting failed", e);
  size_t lvl_at[TBL_MAX_LEVELS + 1] = {0}, bytes2 = 0;
  for (int l = 1; l <= L; ++l) {
    lvl_at[l] = bytes2;
    bytes2 = up(bytes2 + (size_t)h_total[l] * 8 + 8);
  }
  e = hipMalloc(&own->mem2, bytes2);
  if (e != hipSuccess) {
    own->mem2 = nullptr;
    return fail_build(GIGL_E_OOM, "hipMalloc of the level lists failed", e);
  }
  RangeTable t{};
  t.dom = dom;
  t.levels = L;
  t.flat = flat;
  TableBuild tbuild{};
  tbuild.flat = flat;
  tbuild.tile_off = cnt;
  tbuild.total = total;
  tbuild.n_tiles = n_tiles;
  tbuild.levels = L;
  tbuild.dom = dom;
  for (int l = 1; l <= L; ++l) {
    tbuild.lvl[l] = (uint2*)((char*)own->mem2 + lvl_at[l]);
    tbuild.off[l] = (uint32_t*)(m0 + off_at[l]);
    t.lvl[l] = tbuild.lvl[l];
    t.off[l] = tbuild.off[l];
  }
  hipLaunchKernelGGL(table_fill_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, tbuild);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail_build(GIGL_E_HIP, "filling failed", e);
  own->t = t;
  own->refs = 1;  // this ctx
  // the registry points to the largest table; earlier ones live on until their ctxs let go
  g_device_table[dev] = own;
  ctx->sampler_table = own;
  return GIGL_OK;
}

// one hop of parity-mode expansion.  `covered` = the host proved every window lies inside the table.
int32_t run_expand(gigl_ctx* ctx, const ExpandArgs& a_in, const RangeTable& tb, bool covered,
                   int64_t* heavy_list, int32_t* heavy_count, RowDesc* desc) {
  ExpandArgs a = a_in;
  if (a.f > GIGL_FAST_FANOUT) {  // fanouts beyond the wave-resident selection: a workgroup per parent slot
    gigl_prof_scope ps(ctx, GIGL_K_EXPAND);
    int64_t wgs = a.n_parents < 8192 ? a.n_parents : 8192;
    hipLaunchKernelGGL(expand_wide_kernel, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, a);
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    return GIGL_OK;
  }
  a.proxy_drop = 0;
  if (const char* e = getenv("GIGL_SAMPLER_PROXY_BITS")) {  // test knob: fewer proxy bits -> forced ties
    int bits = atoi(e);
    if (bits >= 1 && bits <= 32) a.proxy_drop = 32 - bits;
  }
  static const int32_t flat_max = [] {  // (tuning knob: longest row that reads its hash window flat)
    const char* e = getenv("GIGL_EXPAND_FLAT_MAX");
    return e ? atoi(e) : 128;
  }();
  a.flat_max = flat_max;
  if (!covered) gigl_fill_u32(ctx->stream, heavy_count, 0u, 1);
  // the work lists sit behind this hop's descriptors (expand_desc_bytes)
  WorkLists wl{};
  wl.cap = work_list_cap(a.n_parents);
  wl.items = reinterpret_cast<uint32_t*>(desc + a.n_parents + 1);
  wl.count = reinterpret_cast<int32_t*>(wl.items + (int64_t)WORK_LISTS * wl.cap);
  wl.count = reinterpret_cast<int32_t*>(((uintptr_t)wl.count + 127) & ~(uintptr_t)127);
  // (a kernel, not hipMemsetAsync: a memset NODE of a captured graph was seen to run out of order with the planning
  // pass's atomics on this runtime — stale counts made the expansion pass read slots past a list's end)
  hipLaunchKernelGGL(work_counts_zero_kernel, dim3(1), dim3(WORK_LISTS), 0, ctx->stream, wl.count);
  {
    gigl_prof_scope ps(ctx, GIGL_K_EXPAND);
    hipLaunchKernelGGL(plan_rows_kernel, dim3((unsigned)(a.n_parents / 256 + 1)), dim3(256), 0, ctx->stream, a, tb,
                       desc, covered ? nullptr : heavy_list, heavy_count, wl);
    // persistent: 8 waves per SIMD on every CU at most, a multiple of WORK_LISTS waves
    int64_t wgs = ((a.n_parents + 7) / 8 + 15) / 16 * 16;
    static const int64_t wgs_max = [] {  // (tuning knob: the persistent grid's size, a multiple of 16)
      const char* e = getenv("GIGL_EXPAND_WGS");
      const int64_t v = e ? atoll(e) : 2048;
      return v < 16 ? (int64_t)16 : v / 16 * 16;
    }();
    if (wgs > wgs_max) wgs = wgs_max;
    // (round 6) waves that retire after 16 steps instead of persistent ones: a hop of 1.6 M rows is 13 copies of the 2,048-
    // workgroup grid, and the other streams' workgroups get wave slots as the copies retire — products 19.5 -> 19.3 us per
    // step on one box, 18.9 -> 18.7 on another, `expand` 4.8 -> 4.7 us alone (profiles/r06am_expand_iters.txt);
    // GIGL_EXPAND_ITERS=0 keeps the persistent grid (A/B).  Hops of fewer steps than that are one copy: unchanged.
    static const int32_t iters = [] {
      const char* e = getenv("GIGL_EXPAND_ITERS");
      const int v = e ? atoi(e) : 16;
      return v < 0 ? 0 : v;
    }();
    wl.iters = iters;
    wl.wgs = (int32_t)wgs;
    if (iters > 0) {
      const int64_t stride = wgs * 4 / WORK_LISTS;                       // steps of a list taken per pass of the grid
      const int64_t steps = (wl.cap / 2 + stride - 1) / stride;          // a wave's longest possible sequence
      wgs *= (steps + iters - 1) / iters;
    }
    hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, a, tb, (const RowDesc*)desc, wl);
  }
  if (!covered) {
    gigl_prof_scope ps(ctx, GIGL_K_EXPAND_HEAVY);
    int64_t hb = a.n_parents < 2048 ? a.n_parents : 2048;
    hipLaunchKernelGGL(expand_heavy_kernel, dim3((unsigned)hb), dim3(256), 0, ctx->stream, a, heavy_list,
                       heavy_count);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

// largest j any window of this call can reach, or UINT64_MAX if windows may wrap / go negative
uint64_t window_bound(const gigl_graph* g, int32_t hops, int32_t first_counter, int32_t sampling_seed) {
  uint64_t worst = 0;
  for (int k = 0; k < hops; ++k) {
    int64_t add = (int64_t)sampling_seed * (int64_t)(first_counter + k);
    if (add < 0 || add > (int64_t)0x7FFFFFFF) return ~0ULL;
    // K <= (k+1) ids each < n ; window end = K + add + deg
    uint64_t hi = (uint64_t)(k + 1) * (uint64_t)(g->n > 0 ? g->n - 1 : 0) + (uint64_t)add + (uint64_t)g->maxdeg;
    if (hi > worst) worst = hi;
  }
  return worst;
}

// ---- fast (non-parity) mode: stratified positions from a counter-based RNG: reads f ids instead of
// deg.  Labelled NOT parity everywhere.
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
    // stratum j covers [j*deg/f, (j+1)*deg/f); one uniform draw inside each -> f distinct ascending positions
    if (lane < f) {
      int64_t lo = (int64_t)lane * deg / f, hi = (int64_t)(lane + 1) * deg / f;
      uint32_t r = mix32(mix32(ksum + (uint32_t)a.hash_add) ^ (uint32_t)(lane * 0x9E3779B9u));
      int64_t pos = lo + (int64_t)(((uint64_t)r * (uint64_t)(hi - lo)) >> 32);
      out[lane] = row[pos];
    }
    if (lane == 0) a.out_cnt[p] = f;
  }
}

// with-replacement mode (the reference's sampleWithReplacementUDF, SGSPureSparkV1Task.scala:42-50: `numSamples`
// independent uniform draws from the neighbour list, an UNSEEDED java.util.Random there): f draws per parent from a
// counter-based generator keyed by (K + seed*counter, draw number) — reproducible; every parent with in-edges gets
// exactly f entries (repeats possible, also when deg < f), written in ascending id order.  Not a parity mode: the
// reference's draws cannot be reproduced.
__global__ __launch_bounds__(256) void expand_replace_kernel(ExpandArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int64_t waves_total = (int64_t)gridDim.x * 4;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave_in_block; p < a.n_parents; p += waves_total) {
    uint32_t v, ksum;
    parent_of(a, p, v, ksum);
    uint32_t* out = a.out_nbr + p * a.f;
    const int f = a.f;
    int64_t deg = 0, s = 0;
    if (v != GIGL_INVALID && (int64_t)v < a.n_nodes) {
      s = a.rowptr[v];
      deg = a.rowptr[v + 1] - s;
    }
    if (deg == 0) {
      if (lane < f) out[lane] = GIGL_INVALID;
      if (lane == 0) a.out_cnt[p] = 0;
      continue;
    }
    uint32_t id = GIGL_INVALID;
    if (lane < f) {
      const uint32_t r = mix32(mix32(ksum + (uint32_t)a.hash_add) ^ (uint32_t)((lane + 1) * 0x9E3779B9u));
      id = a.col[s + (int64_t)(((uint64_t)r * (uint64_t)deg) >> 32)];
    }
    int rank = 0;  // ascending ids, equal ids in lane order
    for (int l = 0; l < f; ++l) {
      const uint32_t o = readlane32(id, l);
      rank += (o < id || (o == id && l < lane)) ? 1 : 0;
    }
    if (lane < f) out[rank] = id;
    if (lane == 0) a.out_cnt[p] = f;
  }
}

}  // namespace

void gigl_sampler_table_free(gigl_ctx* ctx) {  // (the caller has synchronised the ctx stream)
  TableOwner* own = (TableOwner*)ctx->sampler_table;
  if (!own) return;
  std::lock_guard<std::mutex> lock(g_table_mu);
  table_unref(own);
  ctx->sampler_table = nullptr;
}

extern "C" {

int32_t gigl_sample_khop(gigl_ctx* ctx, gigl_graph* g, const uint32_t* roots, int32_t b,
                         const int32_t* fanouts, int32_t hops, int32_t sampling_seed, int32_t mode,
                         gigl_tree* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, g && (roots || b == 0) && fanouts && out, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS, "hops must be in [1,%d]", GIGL_MAX_HOPS);
  GIGL_REQUIRE(ctx, b >= 0, "negative batch");
  GIGL_REQUIRE(ctx, mode == GIGL_MODE_SPARK_HASH || mode == GIGL_MODE_FAST || mode == GIGL_MODE_REPLACE,
               "bad mode %d", mode);
  int64_t parents = b;
  for (int k = 0; k < hops; ++k) {
    if (fanouts[k] < 1 || fanouts[k] > GIGL_MAX_FANOUT)
      return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "fanout[%d]=%d outside [1,%d]", k, fanouts[k],
                       GIGL_MAX_FANOUT);
    if (mode != GIGL_MODE_SPARK_HASH && fanouts[k] > GIGL_FAST_FANOUT)
      return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "fanout[%d]=%d: the non-parity modes take fanouts up to %d", k, fanouts[k],
                       GIGL_FAST_FANOUT);
    GIGL_REQUIRE(ctx, out->nbr[k] && out->cnt[k], "tree buffers for hop %d are null", k);
    parents *= fanouts[k];
    GIGL_REQUIRE(ctx, parents < (int64_t)1 << 31, "tree too large");
  }
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  out->hops = hops;
  out->b = b;
  for (int k = 0; k < hops; ++k) out->fanouts[k] = fanouts[k];
  if (b == 0) return GIGL_OK;

  RangeTable tb{};
  bool covered = false;
  int32_t rc = GIGL_OK;
  if (mode == GIGL_MODE_SPARK_HASH) {
    // size the hash threshold table for this graph/seed (built once, reused by every later call)
    const uint64_t bound = window_bound(g, hops, 1, sampling_seed);
    const uint64_t cap = table_cap_j();  // windows beyond the table fall back to direct hashing
    rc = ensure_table(ctx, bound == ~0ULL ? (1ull << 20) : (bound + 1 < cap ? bound + 1 : cap));
    if (rc != GIGL_OK) return rc;
    tb = ((TableOwner*)ctx->sampler_table)->t;
    covered = bound != ~0ULL && bound < tb.dom;
  }

  // scratch: one row descriptor per parent slot of the widest hop; heavy list + counter when some window may fall
  // outside the table
  int64_t* heavy_list = nullptr;
  int32_t* heavy_count = nullptr;
  RowDesc* desc = nullptr;
  if (mode == GIGL_MODE_SPARK_HASH) {
    int64_t max_parents = b, q = b;
    for (int k = 0; k + 1 < hops; ++k) {
      q *= fanouts[k];
      if (q > max_parents) max_parents = q;
    }
    rc = gigl_arena_reset(ctx, expand_desc_bytes(max_parents) + max_parents * 8 + 1024);
    if (rc != GIGL_OK) return rc;
    desc = (RowDesc*)gigl_arena_alloc(ctx, expand_desc_bytes(max_parents));
    if (!covered) {
      heavy_list = (int64_t*)gigl_arena_alloc(ctx, max_parents * 8);
      heavy_count = (int32_t*)gigl_arena_alloc(ctx, 256);
    }
    if (!desc || (!covered && (!heavy_list || !heavy_count))) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  }

  ExpandArgs a{};
  a.rowptr = g->rowptr;
  a.col = g->col;
  a.n_nodes = g->n;
  a.multi = g->multi ? 1 : 0;
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
    if (mode == GIGL_MODE_FAST || mode == GIGL_MODE_REPLACE) {
      int64_t blocks = (parents + 3) / 4;
      if (blocks > 256 * 32) blocks = 256 * 32;
      gigl_prof_scope ps(ctx, GIGL_K_EXPAND);
      if (mode == GIGL_MODE_FAST)
        hipLaunchKernelGGL(expand_fast_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
      else
        hipLaunchKernelGGL(expand_replace_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
      GIGL_HIP_CHECK(ctx, hipGetLastError());
    } else {
      rc = run_expand(ctx, a, tb, covered, heavy_list, heavy_count, desc);
      if (rc != GIGL_OK) return rc;
    }
    a.anc[k] = out->nbr[k];
    parents *= fanouts[k];
  }
  return GIGL_OK;
}

int32_t gigl_sample_khop_peer(gigl_ctx* ctx, const int64_t* const* peer_rowptr, const uint32_t* const* peer_col,
                              int32_t world, int64_t n_global, int64_t max_window_end, const uint32_t* roots, int32_t b,
                              const int32_t* fanouts, int32_t hops, int32_t sampling_seed, gigl_tree* out) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, peer_rowptr && peer_col && (roots || b == 0) && fanouts && out, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS && b >= 0 && world >= 1 && world < (1 << 15), "bad shape");
  int64_t parents = b, max_parents = b;
  for (int k = 0; k < hops; ++k) {
    if (fanouts[k] < 1 || fanouts[k] > GIGL_FAST_FANOUT)
      return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "peer-mapped sampling takes fan-outs up to %d (fanout[%d]=%d)", GIGL_FAST_FANOUT, k,
                       fanouts[k]);
    GIGL_REQUIRE(ctx, out->nbr[k] && out->cnt[k], "tree buffers for hop %d are null", k);
    if (parents > max_parents) max_parents = parents;
    parents *= fanouts[k];
    GIGL_REQUIRE(ctx, parents < (int64_t)1 << 31, "tree too large");
  }
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  out->hops = hops;
  out->b = b;
  for (int k = 0; k < hops; ++k) out->fanouts[k] = fanouts[k];
  if (b == 0) return GIGL_OK;
  // every hash window of the job must lie inside the threshold table (the caller's bound): the rows beyond it would take the
  // workgroup-per-row paths, which read the adjacency of ONE resident graph
  const uint64_t cap = table_cap_j();
  if (max_window_end < 0 || (uint64_t)max_window_end + 1 > cap)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "peer-mapped sampling needs a window bound inside the threshold table (max_window_end)");
  int32_t rc = ensure_table(ctx, (uint64_t)max_window_end + 1);
  if (rc != GIGL_OK) return rc;
  const RangeTable tb = ((TableOwner*)ctx->sampler_table)->t;
  if ((uint64_t)max_window_end >= tb.dom)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "peer-mapped sampling: the window bound exceeds the threshold table");
  rc = gigl_arena_reset(ctx, expand_desc_bytes(max_parents) + 1024);
  if (rc != GIGL_OK) return rc;
  RowDesc* desc = (RowDesc*)gigl_arena_alloc(ctx, expand_desc_bytes(max_parents));
  if (!desc) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  ExpandArgs a{};
  a.peer_rowptr = peer_rowptr;
  a.peer_col = peer_col;
  a.peer_world = (uint32_t)world;
  a.n_nodes = n_global;
  a.multi = 0;
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
    rc = run_expand(ctx, a, tb, true, nullptr, nullptr, desc);
    if (rc != GIGL_OK) return rc;
    a.anc[k] = out->nbr[k];
    parents *= fanouts[k];
  }
  return GIGL_OK;
}

int32_t gigl_expand_frontier(gigl_ctx* ctx, gigl_graph* shard, const uint32_t* nodes, const uint32_t* ksums,
                             int64_t m, int32_t f, int32_t hash_add, int32_t world, int64_t max_window_end,
                             uint32_t* out_nbr, int32_t* out_cnt) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, shard && (m == 0 || (nodes && ksums && out_nbr && out_cnt)), "null argument");
  GIGL_REQUIRE(ctx, world >= 1 && m >= 0 && m < ((int64_t)1 << 31), "bad sizes");
  if (f < 1 || f > GIGL_MAX_FANOUT)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "fanout %d outside [1,%d]", f, GIGL_MAX_FANOUT);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m == 0) return GIGL_OK;
  const uint64_t cap = table_cap_j();
  const bool bounded = max_window_end >= 0;
  const uint64_t bound = bounded ? (uint64_t)max_window_end : ~0ULL;
  int32_t rc = ensure_table(ctx, !bounded ? (1ull << 20) : (bound + 1 < cap ? bound + 1 : cap));
  if (rc != GIGL_OK) return rc;
  const RangeTable tb = ((TableOwner*)ctx->sampler_table)->t;
  const bool covered = bounded && bound < tb.dom;
  int64_t* heavy_list = nullptr;
  int32_t* heavy_count = nullptr;
  rc = gigl_arena_reset(ctx, expand_desc_bytes(m) + m * 8 + 1024);
  if (rc != GIGL_OK) return rc;
  RowDesc* desc = (RowDesc*)gigl_arena_alloc(ctx, expand_desc_bytes(m));
  if (!covered) {
    heavy_list = (int64_t*)gigl_arena_alloc(ctx, m * 8);
    heavy_count = (int32_t*)gigl_arena_alloc(ctx, 256);
  }
  if (!desc || (!covered && (!heavy_list || !heavy_count))) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  ExpandArgs a{};
  a.rowptr = shard->rowptr;
  a.col = shard->col;
  a.n_nodes = shard->n;  // local rows
  a.multi = shard->multi ? 1 : 0;
  a.hop = 0;
  a.n_parents = m;
  a.f = f;
  a.fan[0] = f;
  a.hash_add = hash_add;
  a.out_nbr = out_nbr;
  a.out_cnt = out_cnt;
  a.ex_nodes = nodes;
  a.ex_ksum = ksums;
  a.row_div = (uint32_t)world;
  return run_expand(ctx, a, tb, covered, heavy_list, heavy_count, desc);
}

int32_t gigl_sample_positives(gigl_ctx* ctx, gigl_graph* g_out, const uint32_t* roots, int32_t b,
                              int32_t f, int32_t sampling_seed, int32_t mode, uint32_t* pos,
                              int32_t* cnt) {
  // sampleDstNodesUniformly is the third hashBasedUniformPermutation call of the job: _counter = 3
  // (NodeAnchorBasedLinkPredictionTask.scala:171-172)
  return gigl_sample_out_neighbors(ctx, g_out, roots, b, f, sampling_seed, 3, mode, pos, cnt);
}

int32_t gigl_sample_out_neighbors(gigl_ctx* ctx, gigl_graph* g_out, const uint32_t* roots, int32_t b,
                                  int32_t f, int32_t sampling_seed, int32_t counter, int32_t mode, uint32_t* pos,
                                  int32_t* cnt) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, counter >= 1 && counter <= 64, "permutation call counter %d outside [1,64]", counter);
  GIGL_REQUIRE(ctx, g_out && (roots || b == 0) && pos && cnt, "null argument");
  GIGL_REQUIRE(ctx, mode == GIGL_MODE_SPARK_HASH, "positives are parity-mode only");
  if (f < 1 || f > GIGL_MAX_FANOUT)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "num positives %d outside [1,%d]", f, GIGL_MAX_FANOUT);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (b == 0) return GIGL_OK;
  int32_t rc = gigl_arena_reset(ctx, expand_desc_bytes(b) + (int64_t)(b + 1) * 8 + 1024);
  if (rc != GIGL_OK) return rc;
  RowDesc* desc = (RowDesc*)gigl_arena_alloc(ctx, expand_desc_bytes(b));
  int64_t* heavy_list = (int64_t*)gigl_arena_alloc(ctx, (int64_t)b * 8);
  int32_t* heavy_count = (int32_t*)gigl_arena_alloc(ctx, 256);
  if (!desc || !heavy_list || !heavy_count) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  ExpandArgs a{};
  a.rowptr = g_out->rowptr;
  a.col = g_out->col;
  a.n_nodes = g_out->n;
  a.multi = g_out->multi ? 1 : 0;
  a.roots = roots;
  a.hop = 0;
  a.n_parents = b;
  a.f = f;
  a.fan[0] = f;
  a.hash_add = (int32_t)((uint32_t)sampling_seed * (uint32_t)counter);
  a.out_nbr = pos;
  a.out_cnt = cnt;
  const uint64_t bound = window_bound(g_out, 1, counter, sampling_seed);
  const uint64_t cap = table_cap_j();
  rc = ensure_table(ctx, bound == ~0ULL ? (1ull << 20) : (bound + 1 < cap ? bound + 1 : cap));
  if (rc != GIGL_OK) return rc;
  const RangeTable tb = ((TableOwner*)ctx->sampler_table)->t;
  return run_expand(ctx, a, tb, bound != ~0ULL && bound < tb.dom, heavy_list, heavy_count, desc);
}

}  // extern "C"
