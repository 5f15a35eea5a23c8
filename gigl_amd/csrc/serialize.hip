// serialize.hip — sampled trees -> serialized training-sample protos in TFRecord framing, on the device.
//
// Replaces the per-root assembly + hydration + proto encoding + record writing of the Spark sampler
// (paths relative to the reference root):
//   hydrateNodes / hydrateEdges / createSubgraph        scala/subgraph_sampler/src/main/scala/libs/task/pureSpark/
//                                                        SGSPureSparkV1Task.scala:496-593, :671-820
//   createNodeAnchorBasedLinkPredictionSubgraph         .../pureSpark/NodeAnchorBasedLinkPredictionTask.scala:146-312
//   castToRootedNodeNeighborhoodProtoSchema             SGSPureSparkV1Task.scala:1019-1040
//   TFRecordIO.writeDatasetToTfrecord                   scala/common/src/main/scala/utils/TFRecordIO.scala:53-69
// Messages: proto/snapchat/research/gbml/graph_schema.proto:5-62 (Node, Edge, Graph),
//           proto/snapchat/research/gbml/training_samples_schema.proto:16-43.
//
// Byte/integer work, HBM-bound: a record is ~ nodes*(4*D + 10) + edges*12 bytes and the feature payload of
// every DISTINCT node of a root's neighbourhood is copied once from the resident table (the reference joins D
// floats per sampled OCCURRENCE and dedups afterwards).
//
// Three launches (details at record_plan_kernel): a plan pass — one WAVE per record: the record's node stream read
// once into LDS, an LDS hash set for the first occurrences, wave scans for the byte offset of every field, the
// record's size; the part of the plan the writers need is saved per record —, a scan of the record sizes, and a
// write pass whose workgroups come in two kinds that run side by side: field writers (one wave per record: frame
// header, field headers, edges, label edges, suffix and BOTH checksums, bytes gathered in registers and stored 8 at a
// time at whatever address they belong) and row copiers (every (node field, 16-byte chunk) an independent item:
// 16-byte load, unaligned 16-byte store).  The payload is never read back for its CRC-32C: CRC is linear, a node's
// 4*D payload bytes contribute a state tabulated once per feature table (gigl_features_row_crc) after one table-driven
// shift of the running state, and a lane shifts its chained state to the end of the payload with one multiply mod P.
// Measured (MI355X, products-shaped graph, [25,10], D=100, ~90 distinct nodes and 40.4 KB per record; device time of
// back-to-back calls, scripts/micro_records.py --device-only): 4,096 records per call 182 us (plan 54 + scan ~5 + write
// ~120) = 0.91 TB/s of finished TFRecord bytes; 32,768 per call 1.05 ms (plan 0.19 + write 0.83) = 1.22 TB/s =
// 0.31 of the HBM peak counting the bytes written and the row bytes read (round 2: 289 us per 4,096 = 0.57 TB/s,
// 0.14).  What the pattern allows: the row copy alone as dense items runs at 4.5 TB/s read + written
// (scripts/micro_rowcopy.py: 66 us per 4,096 records).  What was learnt on the way (profiles/r03g_encoder.md): per-byte
// stores and a CRC pass over the payload each cost more than the payload copy; one ticket counter or one look-back
// chain shared by thousands of waves serialises them; the copy inside a per-record wave runs at half the dense rate.
#include "common.h"

#include <hip/hip_fp16.h>

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t CRC_POLY = 0x82F63B78u;  // CRC-32C (Castagnoli), reflected

struct RecArgs {
  const uint32_t* roots;  // [n_records * trees_per_record]
  const uint32_t* nbr[GIGL_MAX_HOPS];
  int32_t fan[GIGL_MAX_HOPS];
  int32_t slots[GIGL_MAX_HOPS];  // slots per tree at hop k
  int32_t hops;
  int32_t trees;      // trees per record
  int32_t tree_len;   // node-stream positions per tree = sum slots + 1
  int32_t edge_len;   // edge-stream positions per tree = sum slots
  int32_t kind;
  int32_t node_type, edge_type;
  int32_t frame;
  const void* feat;
  int32_t d, feat_dtype;
  int64_t feat_n;
  const uint8_t* emit;
  const uint8_t* suffix;
  const int64_t* suffix_off;
  int64_t n_records;
  uint32_t hash_cap;   // node hash entries (pow2)
  uint32_t ehash_cap;  // edge hash entries (pow2; 0 when trees == 1: edges are distinct by construction)
  const uint32_t* shift_tbl;  // [3][256]: x^(8*b*256^j) mod P, reflected (CRC combine)
  // edge features (Edge.feature_values = 4): row p of `efeat` belongs to the edge at position p of the resident CSC
  const float* efeat;
  int32_t de;
  const int64_t* g_rowptr;
  const uint32_t* g_col;
  int64_t g_n;
  // user-defined label edges (UserDefinedLabelsNodeAnchorBasedLinkPredictionTask.scala): the last neg_trees trees of
  // a record are the hard negatives; lab[0] / lab[1] = where the features of the pos / hard-neg edges come from
  int32_t neg_trees;
  struct LabelEdges {
    const int64_t* rowptr;  // CSR by SOURCE of the label edge list (NULL: pos edges are main edges, see edge_pos)
    const uint32_t* col;
    int64_t n;
    const float* feat;  // rows in `col` order
    int32_t de;
  } lab[2];
};

__device__ __forceinline__ int vlen(uint32_t v) {
  return v < (1u << 7) ? 1 : v < (1u << 14) ? 2 : v < (1u << 21) ? 3 : v < (1u << 28) ? 4 : 5;
}
__device__ __forceinline__ uint8_t* put_varint(uint8_t* p, uint64_t v) {
  while (v >= 128) {
    *p++ = (uint8_t)(v | 0x80);
    v >>= 7;
  }
  *p++ = (uint8_t)v;
  return p;
}

// Node message body length (graph_schema.proto:5-12): node_id elided when 0 (proto3 scalar), condensed_node_type
// is `optional` (explicit presence), feature_values packed
__device__ __forceinline__ uint32_t node_body_len(const RecArgs& a, uint32_t id) {
  uint32_t n = 0;
  if (id) n += 1 + vlen(id);
  if (a.node_type >= 0) n += 1 + vlen((uint32_t)a.node_type);
  if (a.d > 0) n += 1 + vlen(4u * (uint32_t)a.d) + 4u * (uint32_t)a.d;
  return n;
}
__device__ __forceinline__ uint32_t edge_body_len_de(const RecArgs& a, uint32_t s, uint32_t d, int32_t de) {
  uint32_t n = 0;
  if (s) n += 1 + vlen(s);
  if (d) n += 1 + vlen(d);
  if (a.edge_type >= 0) n += 1 + vlen((uint32_t)a.edge_type);
  if (de > 0) n += 1 + vlen(4u * (uint32_t)de) + 4u * (uint32_t)de;
  return n;
}
__device__ __forceinline__ uint32_t edge_body_len(const RecArgs& a, uint32_t s, uint32_t d) {
  return edge_body_len_de(a, s, d, a.de);
}
// feature dimension of the pos (which = 0) / hard-neg (1) label edges
__device__ __forceinline__ int32_t label_de(const RecArgs& a, int which) {
  return a.lab[which].rowptr ? a.lab[which].de : (which == 0 ? a.de : 0);
}
__device__ __forceinline__ uint32_t field_len(uint32_t body) { return 1 + vlen(body) + body; }
// position of the edge s -> d in the resident CSC (row d ascending), NONE when absent
__device__ __forceinline__ uint32_t edge_pos(const RecArgs& a, uint32_t s, uint32_t d) {
  if ((int64_t)d >= a.g_n) return NONE;
  int64_t lo = a.g_rowptr[d];
  const int64_t end = a.g_rowptr[d + 1];
  int64_t hi = end;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a.g_col[mid] < s) lo = mid + 1;
    else hi = mid;
  }
  return (lo < end && a.g_col[lo] == s) ? (uint32_t)lo : NONE;
}
// bytes of a Node field before its float payload: tag, length, node_id, condensed_node_type, feature_values header
__device__ __forceinline__ uint32_t node_hdr_len(const RecArgs& a, uint32_t id) {
  uint32_t n = 1 + vlen(node_body_len(a, id));
  if (id) n += 1 + vlen(id);
  if (a.node_type >= 0) n += 1 + vlen((uint32_t)a.node_type);
  if (a.d > 0) n += 1 + vlen(4u * (uint32_t)a.d);
  return n;
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns the exclusive prefix,
// total through `total`.  s_w: 8 uint32 of LDS.
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t* s_w, uint32_t& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  __syncthreads();
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int i = 0; i < w; ++i) base += s_w[i];
  total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return base + inc - v;
}

// ---- wave-level pieces: one WAVE owns a record, so nothing below needs a workgroup barrier ----
// LDS written by some lanes of a wave and read by others: the LDS queue is in order per wave; the fence keeps the
// compiler from moving accesses across it
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// exclusive scan of one 64-bit value per lane; total = sum over the wave
__device__ __forceinline__ unsigned long long wave_exscan64(unsigned long long v, unsigned long long& total) {
  const int lane = threadIdx.x & 63;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  total = __shfl(inc, 63, 64);
  return inc - v;
}

// The per-record plan: built in LDS (one segment per wave) — or, for streams too long for that, straight in the record's
// plan segment in scratch (same code, the arrays are then global memory).  Kept for the write pass (kept_words()):
//   hdr      [4]         n_uniq, nodes_bytes, edges_bytes, bytes of the hard_neg_edges fields
//   uid, fld [n_s+1]     the record's Node fields in RECORD order — entry 0 = the root_node field (offset 0 of the
//                        payload), entry 1 + u = the u-th distinct node in stream order: id, byte offset of the field
//                        inside the Graph body (low FLD_BITS bits; the bits above: length of the field's header,
//                        <= 24 bytes)
//   rc       [n_s+1]     the tabulated CRC state of every entry's feature row (gigl_features_row_crc), fetched for
//                        the whole record at once
//   edge_off [n_e]       byte offset of the edge inside the edge region, NONE = duplicate / empty
//   ids      [n_s]       the record's node stream, tree by tree: hop-0 slots, hop-1 slots, ..., the tree's root;
//                        NONE = empty slot.  Read from the sample ONCE; everything works on this copy.
// Scratch of the plan pass:
//   hslot    [hash_cap]  open-addressing set over the stream: a slot holds the FIRST stream position of its id
//                        (the key is ids[position]: 4 bytes per slot)
//   eslot    [ehash_cap] the same over the edge stream (records that merge several trees only)
constexpr uint32_t FLD_BITS = 27, FLD_MASK = (1u << FLD_BITS) - 1;
constexpr uint32_t PLAN_HDR = 4;
struct Plan {
  uint32_t *hdr, *uid, *fld, *rc, *edge_off, *ids, *hslot, *eslot;
  uint32_t n_uniq, nodes_bytes, edges_bytes;
};

__host__ __device__ inline uint32_t kept_words(const RecArgs& a) {
  const uint32_t n_s = (uint32_t)(a.trees * a.tree_len), n_e = (uint32_t)(a.trees * a.edge_len);
  return PLAN_HDR + 3 * (n_s + 1) + n_e + n_s;
}
size_t kept_bytes(const RecArgs& a) { return ((size_t)4 * kept_words(a) + 15) & ~(size_t)15; }
size_t plan_bytes(const RecArgs& a) {
  return ((size_t)4 * (kept_words(a) + a.hash_cap + a.ehash_cap) + 15) & ~(size_t)15;
}

__device__ __forceinline__ void carve(const RecArgs& a, unsigned char* base, Plan& pl) {
  const uint32_t n_s = (uint32_t)(a.trees * a.tree_len), n_e = (uint32_t)(a.trees * a.edge_len);
  uint32_t* p = (uint32_t*)base;
  pl.hdr = p;
  p += PLAN_HDR;
  pl.uid = p;
  p += n_s + 1;
  pl.fld = p;
  p += n_s + 1;
  pl.rc = p;
  p += n_s + 1;
  pl.edge_off = p;
  p += n_e;
  pl.ids = p;
  p += n_s;
  pl.hslot = p;
  p += a.hash_cap;
  pl.eslot = p;
}

// x / f for x < 2^20 and 1 <= f <= 64 without the (emulated) integer division: x * ceil(2^26 / f) >> 26 is exact there
__device__ __forceinline__ uint32_t div_fanout(uint32_t x, uint32_t f) {
  if (f == 1 || f > 64u || x >= (1u << 20)) return x / f;  // (fanouts past 64: the plain division)
  return __umulhi(x, (((1u << 26) + f - 1u) / f) << 6);
}
// edge at edge-stream position q, from the plan's copy of the stream: src NONE = no edge
__device__ __forceinline__ void plan_edge(const RecArgs& a, const uint32_t* ids, uint32_t q, uint32_t& s, uint32_t& d) {
  uint32_t tt = 0, local = q;
  if (a.trees > 1) {
    tt = q / (uint32_t)a.edge_len;
    local = q - tt * (uint32_t)a.edge_len;
  }
  const uint32_t* t_ids = ids + tt * (uint32_t)a.tree_len;
  s = t_ids[local];
  d = NONE;
  if (s == NONE) return;
  uint32_t prev_base = 0, base = 0;
  for (int k = 0; k < a.hops; ++k) {
    const uint32_t sl = (uint32_t)a.slots[k];
    if (local < base + sl) {
      d = k == 0 ? t_ids[a.tree_len - 1] : t_ids[prev_base + div_fanout(local - base, (uint32_t)a.fan[k])];
      return;
    }
    prev_base = base;
    base += sl;
  }
}
__device__ __forceinline__ uint32_t edge_hash(uint32_t s, uint32_t d) { return hash32(s * 0x9E3779B1u ^ hash32(d)); }

// first stream position of `id` (which is in the set)
__device__ __forceinline__ uint32_t node_first(const RecArgs& a, const Plan& pl, uint32_t id) {
  uint32_t h = hash32(id) & (a.hash_cap - 1);
  for (;;) {
    const uint32_t p = pl.hslot[h];
    if (pl.ids[p] == id) return p;
    h = (h + 1) & (a.hash_cap - 1);
  }
}
__device__ __forceinline__ uint32_t edge_first(const RecArgs& a, const Plan& pl, uint32_t s, uint32_t d) {
  uint32_t h = edge_hash(s, d) & (a.ehash_cap - 1);
  for (;;) {
    const uint32_t p = pl.eslot[h];
    uint32_t ps, pd;
    plan_edge(a, pl.ids, p, ps, pd);
    if (ps == s && pd == d) return p;
    h = (h + 1) & (a.ehash_cap - 1);
  }
}

// barrier + scan of the NT threads that build one record's plan: NT = 64 the lanes of one wave (nothing but wave-local
// ordering), NT = 256 a whole workgroup (s_w: 8 x uint64 of LDS for the waves' totals)
template <int NT>
__device__ __forceinline__ void grp_sync() {
  if constexpr (NT == 64) wave_sync();
  else __syncthreads();
}
template <int NT>
__device__ __forceinline__ unsigned long long grp_exscan64(unsigned long long v, unsigned long long& total,
                                                           unsigned long long* s_w) {
  unsigned long long wt;
  const unsigned long long ex = wave_exscan64(v, wt);
  if constexpr (NT == 64) {
    total = wt;
    return ex;
  } else {
    const int w = threadIdx.x >> 6;
    __syncthreads();  // (s_w may still be read from the previous scan)
    if ((threadIdx.x & 63) == 0) s_w[w] = wt;
    __syncthreads();
    unsigned long long base = 0, all = 0;
#pragma unroll
    for (int q = 0; q < NT / 64; ++q) {
      const unsigned long long x = s_w[q];
      if (q < w) base += x;
      all += x;
    }
    total = all;
    return base + ex;
  }
}

// builds the plan of record r: the 64 lanes of one wave (NT = 64), or a workgroup of 256 threads (NT = 256: the same phases
// with a position or two per thread instead of five per lane — a record's chain of dependent steps gets that much shorter)
template <int NT>
__device__ __attribute__((always_inline)) void build_plan(const RecArgs& a, int64_t r, Plan& pl,
                                                          const uint32_t* row_crc, unsigned long long* s_w = nullptr) {
  const uint32_t n_s = (uint32_t)(a.trees * a.tree_len), n_e = (uint32_t)(a.trees * a.edge_len);
  const uint32_t lane = NT == 64 ? (threadIdx.x & 63) : threadIdx.x;
  // the stream, tree by tree: one (uniform) root load, then every slot load is independent of the others
  for (int tt = 0; tt < a.trees; ++tt) {
    const int64_t t = r * a.trees + tt;
    const uint32_t root = a.roots[t];
    uint32_t* t_ids = pl.ids + (uint32_t)tt * (uint32_t)a.tree_len;
    uint32_t base = 0;
    for (int k = 0; k < a.hops; ++k) {
      const uint32_t sl = (uint32_t)a.slots[k];
      const uint32_t* src = a.nbr[k] + t * sl;
      for (uint32_t i = lane; i < sl; i += NT) t_ids[base + i] = root == NONE ? NONE : src[i];
      base += sl;
    }
    if (lane == 0) t_ids[base] = root;
  }
  for (uint32_t i = lane; i < a.hash_cap; i += NT) pl.hslot[i] = NONE;
  for (uint32_t i = lane; i < a.ehash_cap; i += NT) pl.eslot[i] = NONE;
  grp_sync<NT>();
  // first occurrence of every id in stream order
  for (uint32_t q = lane; q < n_s; q += NT) {
    const uint32_t id = pl.ids[q];
    if (id == NONE) continue;
    uint32_t h = hash32(id) & (a.hash_cap - 1);
    for (;;) {
      const uint32_t prev = atomicCAS(&pl.hslot[h], NONE, q);
      if (prev == NONE) break;
      if (pl.ids[prev] == id) {
        if (prev > q) atomicMin(&pl.hslot[h], q);
        break;
      }
      h = (h + 1) & (a.hash_cap - 1);
    }
  }
  if (a.ehash_cap) {
    for (uint32_t q = lane; q < n_e; q += NT) {
      uint32_t s, d;
      plan_edge(a, pl.ids, q, s, d);
      if (s == NONE) continue;
      uint32_t h = edge_hash(s, d) & (a.ehash_cap - 1);
      for (;;) {
        const uint32_t prev = atomicCAS(&pl.eslot[h], NONE, q);
        if (prev == NONE) break;
        uint32_t ps, pd;
        plan_edge(a, pl.ids, prev, ps, pd);
        if (ps == s && pd == d) {
          if (prev > q) atomicMin(&pl.eslot[h], q);
          break;
        }
        h = (h + 1) & (a.ehash_cap - 1);
      }
    }
  }
  grp_sync<NT>();
  // nodes: field sizes of first occurrences, scanned in stream order (a lane owns a contiguous run); bytes in the
  // low word, count in the high word of one 64-bit scan
  {
    const uint32_t per = (n_s + NT - 1) / NT, lo = min(lane * per, n_s), hi = min(lo + per, n_s);
    unsigned long long mine = 0;
    for (uint32_t q = lo; q < hi; ++q) {
      const uint32_t id = pl.ids[q];
      if (id != NONE && node_first(a, pl, id) == q) mine += (1ull << 32) | field_len(node_body_len(a, id));
    }
    unsigned long long tot;
    const unsigned long long off = grp_exscan64<NT>(mine, tot, s_w);
    uint32_t off_b = (uint32_t)off, off_c = 1u + (uint32_t)(off >> 32);
    for (uint32_t q = lo; q < hi; ++q) {
      const uint32_t id = pl.ids[q];
      if (id != NONE && node_first(a, pl, id) == q) {
        pl.uid[off_c] = id;
        pl.fld[off_c] = off_b | (node_hdr_len(a, id) << FLD_BITS);
        ++off_c;
        off_b += field_len(node_body_len(a, id));
      }
    }
    pl.nodes_bytes = (uint32_t)tot;
    pl.n_uniq = (uint32_t)(tot >> 32);
    if (lane == 0) {  // entry 0: the record's root_node field
      const uint32_t root = pl.ids[a.tree_len - 1];
      pl.uid[0] = root;
      pl.fld[0] = node_hdr_len(a, root) << FLD_BITS;
    }
  }
  // edges
  {
    const uint32_t per = (n_e + NT - 1) / NT, lo = min(lane * per, n_e), hi = min(lo + per, n_e);
    unsigned long long bytes = 0;
    for (uint32_t q = lo; q < hi; ++q) {
      uint32_t s, d;
      plan_edge(a, pl.ids, q, s, d);
      bool first = s != NONE;
      if (first && a.ehash_cap) first = edge_first(a, pl, s, d) == q;
      const uint32_t sz = first ? field_len(edge_body_len(a, s, d)) : NONE;
      pl.edge_off[q] = sz;
      if (first) bytes += sz;
    }
    unsigned long long tot;
    uint32_t off_b = (uint32_t)grp_exscan64<NT>(bytes, tot, s_w);
    for (uint32_t q = lo; q < hi; ++q) {
      const uint32_t sz = pl.edge_off[q];
      if (sz != NONE) {
        pl.edge_off[q] = off_b;
        off_b += sz;
      }
    }
    pl.edges_bytes = (uint32_t)tot;
  }
  grp_sync<NT>();
  // the rows' tabulated CRC states: independent loads, one round trip for the whole record
  if (row_crc)
    for (uint32_t i = lane; i <= pl.n_uniq; i += NT) {
      const uint32_t id = pl.uid[i];
      pl.rc[i] = (int64_t)id < a.feat_n ? row_crc[id] : 0u;
    }
  grp_sync<NT>();
}

// sizes of the fixed parts of record r (uniform over the workgroup)
struct Layout {
  uint32_t root_id, root_body, graph_body, pos_bytes, neg_bytes;
  uint64_t suffix_len, payload;
};
__device__ __forceinline__ Layout layout_of(const RecArgs& a, int64_t r, const Plan& pl) {
  Layout L;
  L.root_id = a.roots[r * a.trees];
  L.root_body = node_body_len(a, L.root_id);
  L.graph_body = pl.nodes_bytes + pl.edges_bytes;
  L.pos_bytes = L.neg_bytes = 0;
  if (a.kind == GIGL_REC_NODE_ANCHOR_LINK_PRED)
    for (int tt = 1; tt < a.trees; ++tt) {
      const uint32_t p = a.roots[r * a.trees + tt];
      if (p == NONE) continue;
      const int which = tt >= a.trees - a.neg_trees ? 1 : 0;
      (which ? L.neg_bytes : L.pos_bytes) += field_len(edge_body_len_de(a, L.root_id, p, label_de(a, which)));
    }
  L.suffix_len = a.suffix_off ? (uint64_t)(a.suffix_off[r + 1] - a.suffix_off[r]) : 0;
  L.payload = (uint64_t)field_len(L.root_body) + L.neg_bytes + field_len(L.graph_body) + L.pos_bytes + L.suffix_len;
  return L;
}

extern __shared__ __align__(16) unsigned char s_dyn[];

// exclusive scan of the record sizes (one workgroup, 8 consecutive sizes per thread and round); status = 1 when the
// output does not fit
// record sizes -> offsets by the NT threads of one workgroup (s_w: NT / 64 words, s_run: one word of LDS)
template <int NT>
__device__ __forceinline__ void scan_sizes(const int64_t* rec_size, int64_t n, int64_t cap, int64_t* rec_off, int32_t* status,
                                           int64_t* s_w, int64_t& s_run) {
  constexpr int PER = 8;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += NT * PER) {
    const int64_t i0 = base + (int64_t)threadIdx.x * PER;
    int64_t v[PER], mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      v[k] = i0 + k < n ? rec_size[i0 + k] : 0;
      mine += v[k];
    }
    int64_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int64_t pre = s_run;
    for (int j = 0; j < w; ++j) pre += s_w[j];
    int64_t run = pre + inc - mine;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (i0 + k < n) rec_off[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (threadIdx.x == NT - 1) s_run = pre + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    rec_off[n] = s_run;
    *status = s_run > cap ? 1 : 0;
  }
}

__global__ __launch_bounds__(1024) void record_scan_kernel(const int64_t* rec_size, int64_t n, int64_t cap,
                                                           int64_t* rec_off, int32_t* status) {
  __shared__ int64_t s_w[16];
  __shared__ int64_t s_run;
  scan_sizes<1024>(rec_size, n, cap, rec_off, status, s_w, s_run);
}

// feature row element k of node `id` as the fp32 bit pattern the proto carries
__device__ __forceinline__ uint32_t feat_word(const RecArgs& a, uint32_t id, uint32_t k) {
  if ((int64_t)id >= a.feat_n) return 0u;  // id outside the table: zeros (never happens for a consistent ingest)
  if (a.feat_dtype == GIGL_DTYPE_F32) return ((const uint32_t*)a.feat)[(int64_t)id * a.d + k];
  return __float_as_uint(__half2float(((const __half*)a.feat)[(int64_t)id * a.d + k]));
}

// ---- CRC-32C pieces (reflected domain: bit 31 of a word is the coefficient of x^0)
__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b) {
  uint32_t m = 1u << 31, p = 0;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1)) == 0) break;
    }
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
  }
  return p;
}
// x^(8*n) mod P for n < 2^24 from the byte-indexed tables (two multiplies instead of one per bit of n)
__device__ __forceinline__ uint32_t x8n_modp(const uint32_t* tbl, uint64_t n) {
  uint32_t p = tbl[n & 0xFF];
  if (n >> 8) p = multmodp(tbl[256 + ((n >> 8) & 0xFF)], p);
  if (n >> 16) p = multmodp(tbl[512 + ((n >> 16) & 0xFF)], p);
  return p;
}
__device__ __forceinline__ uint32_t crc_byte(const uint32_t* t, uint32_t c, uint32_t b) {
  return t[(c ^ b) & 0xFF] ^ (c >> 8);
}
__device__ __forceinline__ uint32_t crc_word(const uint32_t* t, uint32_t c, uint32_t w) {
  c ^= w;
  return t[768 + (c & 0xFF)] ^ t[512 + ((c >> 8) & 0xFF)] ^ t[256 + ((c >> 16) & 0xFF)] ^ t[c >> 24];
}
__device__ __forceinline__ uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }
// slicing-by-4 tables [4][256] built by the 256 threads of a workgroup
__device__ __forceinline__ void build_crc_tables(uint32_t* crc_t) {
  const uint32_t tid = threadIdx.x;
  uint32_t c = tid;
  for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
  crc_t[tid] = c;
  __syncthreads();
  for (int j = 1; j < 4; ++j) {
    const uint32_t prev = crc_t[(j - 1) * 256 + tid];
    crc_t[j * 256 + tid] = (prev >> 8) ^ crc_t[prev & 0xFF];
    __syncthreads();
  }
}

// A byte writer that folds what it writes into a CRC state (the bytes between the float payloads of a record are
// CRC'd from registers while they are written, never re-read).  Bytes gather in a 64-bit register and leave as
// 8-byte stores at whatever byte address they belong (global memory takes unaligned accesses; a byte store per byte
// costs the memory pipeline a request each), the rest as one 4-, 2- and 1-byte store.
// the record bytes are global memory, and the stores below say so (a generic pointer makes them FLAT instructions)
#define GIGL_GLOBAL __attribute__((address_space(1)))
typedef GIGL_GLOBAL uint8_t* gptr_t;
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_unaligned;
struct CrcOut {
  gptr_t p;  // where the next stored byte goes; `nb` more bytes wait in `acc` (not yet folded into c)
  uint32_t c;
  const uint32_t* t;
  uint64_t acc = 0;
  uint32_t nb = 0;
  __device__ __forceinline__ void byte(uint32_t b) {
    acc |= (uint64_t)(b & 0xFFu) << (8u * nb);
    if (++nb == 8) {  // two word steps of the checksum instead of eight dependent byte steps
      c = crc_word(t, crc_word(t, c, (uint32_t)acc), (uint32_t)(acc >> 32));
      *(GIGL_GLOBAL u64_unaligned*)p = acc;
      p += 8;
      acc = 0;
      nb = 0;
    }
  }
  // n <= 8 bytes at once (the low n bytes of w, lowest first)
  __device__ __forceinline__ void bytes(uint64_t w, uint32_t n) {
    const uint32_t sh = 8u * nb;
    acc |= w << sh;
    nb += n;
    if (nb >= 8) {  // two word steps of the checksum, one 8-byte store
      c = crc_word(t, crc_word(t, c, (uint32_t)acc), (uint32_t)(acc >> 32));
      *(GIGL_GLOBAL u64_unaligned*)p = acc;
      p += 8;
      nb -= 8;
      acc = sh ? w >> (64u - sh) : 0ull;  // what did not fit
    }
  }
  // a key byte followed by the varint of a 32-bit value: up to 6 bytes composed in registers
  __device__ __forceinline__ void key_varint(uint32_t key, uint32_t v) {
    const uint32_t n = (uint32_t)vlen(v);
    uint64_t w = (uint64_t)(v & 0x7Fu) | ((uint64_t)((v >> 7) & 0x7Fu) << 8) | ((uint64_t)((v >> 14) & 0x7Fu) << 16) |
                 ((uint64_t)((v >> 21) & 0x7Fu) << 24) | ((uint64_t)(v >> 28) << 32);
    w |= 0x0000008080808080ull & ((1ull << (8u * (n - 1u))) - 1ull);  // continuation bits of all but the last byte
    bytes((uint64_t)key | (w << 8), n + 1u);
  }
  __device__ __forceinline__ void varint(uint64_t v) {
    if (v < (1ull << 32)) {
      const uint32_t x = (uint32_t)v, n = (uint32_t)vlen(x);
      uint64_t w = (uint64_t)(x & 0x7Fu) | ((uint64_t)((x >> 7) & 0x7Fu) << 8) | ((uint64_t)((x >> 14) & 0x7Fu) << 16) |
                   ((uint64_t)((x >> 21) & 0x7Fu) << 24) | ((uint64_t)(x >> 28) << 32);
      w |= 0x0000008080808080ull & ((1ull << (8u * (n - 1u))) - 1ull);
      bytes(w, n);
      return;
    }
    while (v >= 128) {
      byte((uint32_t)(v | 0x80));
      v >>= 7;
    }
    byte((uint32_t)v);
  }
  __device__ __forceinline__ void word(uint32_t v) { bytes(v, 4); }
  __device__ __forceinline__ void flush() {
    if (nb & 4) {
      c = crc_word(t, c, (uint32_t)acc);
      *(GIGL_GLOBAL u32_unaligned*)p = (uint32_t)acc;
      p += 4;
      acc >>= 32;
    }
    if (nb & 2) {
      c = crc_byte(t, crc_byte(t, c, (uint32_t)acc & 0xFFu), ((uint32_t)acc >> 8) & 0xFFu);
      *(GIGL_GLOBAL u16_unaligned*)p = (uint16_t)acc;
      p += 2;
      acc >>= 16;
    }
    if (nb & 1) {
      c = crc_byte(t, c, (uint32_t)acc & 0xFFu);
      *p++ = (uint8_t)acc;
    }
    acc = 0;
    nb = 0;
  }
  __device__ __forceinline__ void seek(gptr_t q) {
    flush();
    p = q;
  }
  __device__ __forceinline__ gptr_t pos() const { return p + nb; }
};

// header of a Node field (everything before the float payload), composed in registers by one thread
__device__ __forceinline__ void write_node_header(const RecArgs& a, CrcOut& o, uint8_t tag, uint32_t id) {
  o.key_varint(tag, node_body_len(a, id));
  if (id) o.key_varint(0x08, id);
  if (a.node_type >= 0) o.key_varint(0x10, (uint32_t)a.node_type);
  if (a.d > 0) o.key_varint(0x1A, 4u * (uint32_t)a.d);
}

// header of an Edge field (everything before the float payload; the whole field without edge features)
__device__ __forceinline__ void write_edge(const RecArgs& a, CrcOut& o, uint8_t tag, uint32_t s, uint32_t d, int32_t de) {
  o.key_varint(tag, edge_body_len_de(a, s, d, de));
  if (s) o.key_varint(0x08, s);
  if (d) o.key_varint(0x10, d);
  if (a.edge_type >= 0) o.key_varint(0x18, (uint32_t)a.edge_type);
  if (de > 0) o.key_varint(0x22, 4u * (uint32_t)de);
}
// a label edge root -> t (pos_edges = 4 / hard_neg_edges = 2) with its features
__device__ __forceinline__ void write_label_edge(const RecArgs& a, CrcOut& o, int which, uint32_t root, uint32_t t) {
  const int32_t de = label_de(a, which);
  write_edge(a, o, which == 0 ? 0x22 : 0x12, root, t, de);
  if (de <= 0) return;
  const RecArgs::LabelEdges& lb = a.lab[which];
  const float* row = nullptr;
  if (lb.rowptr) {  // CSR by source: row `root`, ascending destinations
    if ((int64_t)root < lb.n) {
      int64_t lo = lb.rowptr[root];
      const int64_t end = lb.rowptr[root + 1];
      int64_t hi = end;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (lb.col[mid] < t) lo = mid + 1;
        else hi = mid;
      }
      if (lo < end && lb.col[lo] == t) row = lb.feat + lo * de;
    }
  } else {
    const uint32_t at = edge_pos(a, root, t);
    if (at != NONE) row = a.efeat + (int64_t)at * de;
  }
  for (int k = 0; k < de; ++k) o.word(row ? __float_as_uint(row[k]) : 0u);
}
// feature word k of the edge at CSC position `pos` as the bit pattern the proto carries (zeros for an unknown edge)
__device__ __forceinline__ uint32_t efeat_word(const RecArgs& a, uint32_t pos, uint32_t k) {
  return pos == NONE ? 0u : __float_as_uint(a.efeat[(int64_t)pos * a.de + k]);
}

// ---- per-row CRC table of a feature table ------------------------------------------------------------------------
// out[id] = the raw CRC-32C state (initial state 0, no final inversion) of the 4*d bytes node id's packed
// feature_values carry.  CRC-32C is linear over GF(2): the state after a message is (state before) * x^(8*len) +
// (state of the message alone), so a record's checksum takes one shift + one xor per node instead of a pass over the
// node's 4*d payload bytes — and a node's payload is the same in every record that holds it.  One wave per row.
__global__ __launch_bounds__(256) void row_crc_kernel(const void* feat, int32_t dtype, int64_t n, int32_t d,
                                                      const uint32_t* shift_tbl, uint32_t* out) {
  __shared__ uint32_t crc_t[1024];
  build_crc_tables(crc_t);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t per = ((uint32_t)d + 63u) / 64u;
  const uint32_t lo = min((uint32_t)lane * per, (uint32_t)d), hi = min(lo + per, (uint32_t)d);
  const uint32_t shift = hi < (uint32_t)d ? x8n_modp(shift_tbl, 4ull * ((uint32_t)d - hi)) : 0u;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < n; row += (int64_t)gridDim.x * 4) {
    uint32_t c = 0;
    if (dtype == GIGL_DTYPE_F32) {
      const uint32_t* src = (const uint32_t*)feat + row * d;
      for (uint32_t k = lo; k < hi; ++k) c = crc_word(crc_t, c, src[k]);
    } else {
      const __half* src = (const __half*)feat + row * d;
      for (uint32_t k = lo; k < hi; ++k) c = crc_word(crc_t, c, __float_as_uint(__half2float(src[k])));
    }
    uint32_t part = c;
    if (c && hi < (uint32_t)d) part = multmodp(shift, c);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part ^= __shfl_xor(part, o, 64);
    if (lane == 0) out[row] = part;
  }
}

// ---- the encoder: three launches ---------------------------------------------------------------------------------
// record_plan_kernel   one WAVE per record: the plan (in the wave's LDS segment, or — long streams — straight in the
//                      record's plan segment in scratch), the record's size, and the part of the plan the write pass
//                      needs saved to the record's segment (kept_words(): ~3 KB for a [25,10] record).  The work per
//                      record is a chain of short dependent steps (ids -> hash set -> scans); waves that own a record
//                      each and never meet at a barrier keep many such chains in flight per CU.
// record_scan_kernel   record sizes -> offsets, status = 1 when the output does not fit (nothing is written then).
// record_write_kernel  ONE launch, two kinds of workgroups dealt alternately over the grid so that both are in flight
//                      together (one kind is bound by latency, the other by bandwidth):
//   fields  one wave per record: frame header, field headers, edges, label edges, suffix, both checksums, from the
//           saved plan — everything of the record except the float payloads of its nodes;
//   rows    the payloads: ROWS_WPR waves per record, every (node field, 16-byte chunk) one independent item: a
//           16-byte load from the row, a 16-byte store at whatever byte address the payload starts at (global memory
//           takes unaligned accesses).  Alone this pattern moves rows at ~4.5 TB/s (scripts/micro_rowcopy.py).
// Checksum: nothing is read back.  The bytes between the float payloads are folded into a CRC state while they are
// written (CrcOut); a node's 4*d payload bytes contribute their tabulated state (gigl_features_row_crc) after one
// table-driven shift of the running state; a lane chains the fields of a contiguous run and shifts its state to the
// end of the payload once (state * x^(8*bytes_after) mod P), the wave xors the lanes' shares.
struct EncArgs {
  unsigned char* plans;      // [n_records] plan segments
  size_t plan_stride;        // bytes between the segments of two records
  size_t lds_stride;         // bytes between the LDS segments of two waves
  const uint32_t* row_crc;   // [feat_n] (row_crc_kernel); NULL when d == 0 or the records are not framed
  uint32_t x_row;            // x^(8 * 4d) mod P
  int64_t* rec_size;         // [n_records]
  const uint32_t* tables;    // crc_t[1024] | row_t[1024] | shift_t[768], tabulated once per ctx and row width
};

// the state after 4d more bytes of zeros: c * x^(8*4d), from the byte-sliced tables of that constant
__device__ __forceinline__ uint32_t shift_row(const uint32_t* t, uint32_t c) {
  return t[c & 0xFF] ^ t[256 + ((c >> 8) & 0xFF)] ^ t[512 + ((c >> 16) & 0xFF)] ^ t[768 + (c >> 24)];
}
// a state that ends `after` bytes before the end of the payload, as its share of the final state
__device__ __forceinline__ uint32_t crc_share(const uint32_t* shift_tbl, uint32_t c, uint64_t after) {
  return c ? multmodp(x8n_modp(shift_tbl, after), c) : 0u;
}

template <bool BIG>
__global__ __launch_bounds__(256) void record_plan_kernel(RecArgs a, EncArgs e) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + w;
  if (r >= a.n_records) return;
  uint32_t* const seg = (uint32_t*)(e.plans + (size_t)r * e.plan_stride);
  if (a.emit && !a.emit[r]) {
    if (lane == 0) {
      e.rec_size[r] = 0;
      seg[0] = NONE;  // no record
    }
    return;
  }
  Plan pl;
  if constexpr (BIG) carve(a, (unsigned char*)seg, pl);
  else carve(a, s_dyn + (size_t)w * e.lds_stride, pl);
  build_plan<64>(a, r, pl, e.row_crc);
  const Layout L = layout_of(a, r, pl);
  if (lane == 0) {
    e.rec_size[r] = (int64_t)L.payload + (a.frame ? 16 : 0);
    pl.hdr[0] = pl.n_uniq;
    pl.hdr[1] = pl.nodes_bytes;
    pl.hdr[2] = pl.edges_bytes;
    pl.hdr[3] = L.neg_bytes;
  }
  if constexpr (!BIG) {  // the part the write pass reads, out of LDS
    wave_sync();
    const uint32_t kw = kept_words(a);
    for (uint32_t i = lane; i < kw; i += 64) seg[i] = pl.hdr[i];
  }
}

// the same plan by a WORKGROUP per record (plans that fit LDS): calls of a few thousand records are bound by the length of one
// record's chain of dependent steps, not by throughput — 256 threads walk it with a position or two each
__global__ __launch_bounds__(256) void record_plan_wg_kernel(RecArgs a, EncArgs e) {
  // (the sizes are scanned by record_scan_kernel: a ticket per workgroup + the last one scanning was measured at 291 us per
  // 4,096 records against 169 — every workgroup's release fence writes the L2 back)
  __shared__ unsigned long long s_w[8];
  const int64_t r = blockIdx.x;
  uint32_t* const seg = (uint32_t*)(e.plans + (size_t)r * e.plan_stride);
  if (a.emit && !a.emit[r]) {
    if (threadIdx.x == 0) {
      e.rec_size[r] = 0;
      seg[0] = NONE;  // no record
    }
    return;
  }
  Plan pl;
  carve(a, s_dyn, pl);
  build_plan<256>(a, r, pl, e.row_crc, s_w);
  const Layout L = layout_of(a, r, pl);
  if (threadIdx.x == 0) {
    e.rec_size[r] = (int64_t)L.payload + (a.frame ? 16 : 0);
    pl.hdr[0] = pl.n_uniq;
    pl.hdr[1] = pl.nodes_bytes;
    pl.hdr[2] = pl.edges_bytes;
    pl.hdr[3] = L.neg_bytes;
  }
  __syncthreads();
  const uint32_t kw = kept_words(a);
  for (uint32_t i = threadIdx.x; i < kw; i += 256) seg[i] = pl.hdr[i];
}

__device__ __attribute__((always_inline)) void write_fields(const RecArgs& a, const EncArgs& e, const Plan& pl,
                                                            const Layout& L, int64_t r, uint8_t* rec,
                                                            const uint32_t* crc_t, const uint32_t* row_t,
                                                            const uint32_t* shift_t) {
  const uint32_t lane = threadIdx.x & 63;
  const gptr_t grec = (gptr_t)rec;
  const gptr_t payload = grec + (a.frame ? 12 : 0);
  const gptr_t hard_neg = payload + field_len(L.root_body);  // hard_neg_edges = 2 sits between root_node and neighborhood
  const gptr_t graph_hdr = hard_neg + L.neg_bytes;
  const gptr_t graph = graph_hdr + 1 + vlen(L.graph_body);
  const gptr_t edges = graph + pl.nodes_bytes;
  const gptr_t pos = graph + L.graph_body;
  const gptr_t suffix = pos + L.pos_bytes;
  const gptr_t payload_end = suffix + L.suffix_len;
  const uint64_t n = L.payload;
  uint32_t part = 0;  // this lane's share of the payload's CRC state
  if (lane == 0 && a.frame) {  // u64 length, masked CRC-32C of the length
    CrcOut o{grec, 0xFFFFFFFFu, crc_t};
    o.word((uint32_t)L.payload);
    o.word((uint32_t)(L.payload >> 32));  // (the eighth byte folds and stores the length)
    *(GIGL_GLOBAL u32_unaligned*)(grec + 8) = mask_crc(o.c ^ 0xFFFFFFFFu);
  }
  // root_node = 1 (entry 0) and the distinct nodes of the neighbourhood (Graph.nodes = 2; entries 1..n_uniq), in
  // record order: a lane writes the headers of a contiguous run of entries and chains their checksum
  const uint32_t n_items = pl.n_uniq + 1;
  const uint32_t D4 = 4u * (uint32_t)a.d;
  {
    const uint32_t per = (n_items + 63) / 64, lo = min(lane * per, n_items), hi = min(lo + per, n_items);
    if (lo < hi) {
      CrcOut o{payload, lo == 0 ? 0xFFFFFFFFu : 0u, crc_t};
      for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t id = pl.uid[i];
        if (i == 1) {  // hard_neg_edges = 2 and the Graph header sit between root_node and the first node
          o.seek(hard_neg);
          if (a.kind == GIGL_REC_NODE_ANCHOR_LINK_PRED)
            for (int tt = a.trees - a.neg_trees; tt < a.trees; ++tt) {
              const uint32_t p = a.roots[r * a.trees + tt];
              if (p != NONE) write_label_edge(a, o, 1, L.root_id, p);
            }
          o.byte(a.kind == GIGL_REC_NODE_ANCHOR_LINK_PRED ? 0x1A : 0x12);  // neighborhood = 3 / 2
          o.varint(L.graph_body);
        }
        o.seek((i == 0 ? payload : graph) + (pl.fld[i] & FLD_MASK));
        write_node_header(a, o, i == 0 ? 0x0A : 0x12, id);
        o.flush();
        if (a.d > 0) {
          if (e.row_crc) o.c = shift_row(row_t, o.c) ^ pl.rc[i];
          o.p += D4;
        }
      }
      part ^= crc_share(shift_t, o.c, n - (uint64_t)(o.p - payload));
    }
  }
  // Graph.edges = 3: a lane writes a contiguous run of the edge stream (fields are contiguous in stream order) —
  // header, then the edge's feature row — and chains the checksum
  {
    const uint32_t n_e = (uint32_t)(a.trees * a.edge_len);
    const uint32_t per = (n_e + 63) / 64, lo = min(lane * per, n_e), hi = min(lo + per, n_e);
    CrcOut o{nullptr, 0u, crc_t};
    for (uint32_t q = lo; q < hi; ++q) {
      const uint32_t eo = pl.edge_off[q];
      if (eo == NONE) continue;
      uint32_t s, d;
      plan_edge(a, pl.ids, q, s, d);
      if (!o.p) o.p = edges + eo;  // (the fields of a run are contiguous)
      write_edge(a, o, 0x1A, s, d, a.de);
      if (a.de > 0) {
        const uint32_t at = edge_pos(a, s, d);
        for (int k = 0; k < a.de; ++k) o.word(efeat_word(a, at, (uint32_t)k));
      }
    }
    if (o.p) {
      o.flush();
      part ^= crc_share(shift_t, o.c, n - (uint64_t)(o.p - payload));
    }
  }
  if (lane == 63 && L.pos_bytes) {  // pos_edges = 4, after the graph
    CrcOut o{pos, 0u, crc_t};
    for (int tt = 1; tt < a.trees - a.neg_trees; ++tt) {
      const uint32_t p = a.roots[r * a.trees + tt];
      if (p != NONE) write_label_edge(a, o, 0, L.root_id, p);
    }
    o.flush();
    part ^= crc_share(shift_t, o.c, n - (uint64_t)(o.p - payload));
  }
  if (L.suffix_len) {  // a contiguous chunk per lane
    const uint8_t* src = a.suffix + a.suffix_off[r];
    const uint64_t C = (L.suffix_len + 63) / 64, lo = min((uint64_t)lane * C, L.suffix_len), hi = min(lo + C, L.suffix_len);
    if (lo < hi) {
      CrcOut o{suffix + lo, 0u, crc_t};
      for (uint64_t i = lo; i < hi; ++i) o.byte(src[i]);
      o.flush();
      part ^= crc_share(shift_t, o.c, L.suffix_len - hi);
    }
  }
  if (!a.frame) return;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part ^= __shfl_xor(part, o, 64);
  if (lane == 0) *(GIGL_GLOBAL u32_unaligned*)payload_end = mask_crc(part ^ 0xFFFFFFFFu);
}

// the float payloads of record r's node fields: this wave's share (wave q of ROWS_WPR) of the (entry, 16-byte chunk)
// items, dealt in groups of 64 * UNR
constexpr int ROWS_WGS = 1;             // row-copier workgroups per record (2: no better; more loads in flight per wave: worse)
constexpr int ROWS_WPR = 4 * ROWS_WGS;  // = waves per record
__device__ __attribute__((always_inline)) void write_rows(const RecArgs& a, const uint32_t* seg, uint8_t* rec_start,
                                                          uint32_t q) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t n_s = (uint32_t)(a.trees * a.tree_len);
  const uint32_t n_items = seg[0] + 1;
  const uint32_t* uid = seg + PLAN_HDR;
  const uint32_t* fld = uid + n_s + 1;
  const uint32_t root_body = node_body_len(a, uid[0]);
  const gptr_t payload = (gptr_t)rec_start + (a.frame ? 12 : 0);
  // (the Graph body starts after root_node, the hard negatives and the Graph header)
  const uint32_t neg_bytes = seg[3];
  const uint32_t graph_body = seg[1] + seg[2];
  const gptr_t graph = payload + field_len(root_body) + neg_bytes + 1 + vlen(graph_body);
  const uint32_t D = (uint32_t)a.d, nch = (D + 3) / 4;
  const uint32_t total = n_items * nch;
  const bool vec_rows = a.feat_dtype == GIGL_DTYPE_F32 && (D & 3u) == 0;  // rows are 16-byte aligned
  constexpr int UNR = 2;  // row loads of UNR items are issued before any of them is consumed
  for (uint32_t i0 = q * 64 * UNR; i0 < total; i0 += ROWS_WPR * 64 * UNR) {
    uint32_t k0[UNR], wv[UNR][4];
    gptr_t dst[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      const uint32_t i = i0 + j * 64 + lane;
      const uint32_t u = i / nch, c = i - u * nch;
      k0[j] = 4u * c;
      wv[j][0] = wv[j][1] = wv[j][2] = wv[j][3] = 0;
      dst[j] = nullptr;
      if (i < total) {
        const uint32_t id = uid[u], f = fld[u];
        dst[j] = (u == 0 ? payload : graph) + (f & FLD_MASK) + (f >> FLD_BITS) + 4u * k0[j];
        if (vec_rows) {
          if ((int64_t)id < a.feat_n) {
            const uint4 v = *(const uint4*)((const uint32_t*)a.feat + (int64_t)id * D + k0[j]);
            wv[j][0] = v.x;
            wv[j][1] = v.y;
            wv[j][2] = v.z;
            wv[j][3] = v.w;
          }
        } else {
#pragma unroll
          for (uint32_t t = 0; t < 4; ++t)
            if (k0[j] + t < D) wv[j][t] = feat_word(a, id, k0[j] + t);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      if (!dst[j]) continue;
      if (k0[j] + 3 < D) {
        *(GIGL_GLOBAL u32x4_unaligned*)dst[j] = u32x4_unaligned{wv[j][0], wv[j][1], wv[j][2], wv[j][3]};
      } else {
#pragma unroll
        for (uint32_t t = 0; t < 3; ++t)
          if (k0[j] + t < D) *(GIGL_GLOBAL u32_unaligned*)(dst[j] + 4u * t) = wv[j][t];
      }
    }
  }
}

// the first ceil(n / wf) workgroups write the fields of wf records each (one wave per record; wf = as many plans as fit
// LDS), the n workgroups after them the rows of one record each (ROWS_WPR waves).  The field writers are the long,
// latency-bound ones: they are dispatched first and the row copiers fill the CUs around them
template <bool BIG>
__global__ __launch_bounds__(256) void record_write_kernel(RecArgs a, EncArgs e, const int64_t* rec_off,
                                                           const int32_t* status, uint8_t* out, uint32_t wf) {
  if (*status != 0) return;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, w = tid >> 6;
  const uint32_t n_grp = (uint32_t)((a.n_records + wf - 1) / wf);
  const uint32_t grp = blockIdx.x;
  if (blockIdx.x >= n_grp) {  // rows
    if (a.d <= 0) return;
    const int64_t r = (int64_t)((blockIdx.x - n_grp) / ROWS_WGS);
    if (r >= a.n_records) return;
    const uint32_t* seg = (const uint32_t*)(e.plans + (size_t)r * e.plan_stride);
    if (seg[0] == NONE) return;
    write_rows(a, seg, out + rec_off[r], ((blockIdx.x - n_grp) % ROWS_WGS) * 4 + w);
    return;
  }
  __shared__ uint32_t crc_t[1024 + 1024 + 768];
  uint32_t* const row_t = crc_t + 1024;
  uint32_t* const shift_t = crc_t + 2048;
  for (uint32_t i = tid; i < 2816; i += 256) crc_t[i] = e.tables[i];
  __syncthreads();
  // from here on the waves go their own ways
  if (w >= wf) return;
  const int64_t r = (int64_t)grp * wf + w;
  if (r >= a.n_records) return;
  uint32_t* seg = (uint32_t*)(e.plans + (size_t)r * e.plan_stride);
  if (seg[0] == NONE) return;
  Plan pl;
  carve(a, (unsigned char*)seg, pl);  // the plan is read where the plan pass left it ...
  if constexpr (!BIG) {  // ... but for the stream, which every edge looks into twice: into this wave's LDS segment
    uint32_t* mine = (uint32_t*)(s_dyn + (size_t)w * e.lds_stride);
    const uint32_t n_s = (uint32_t)(a.trees * a.tree_len);
    for (uint32_t i = lane; i < n_s; i += 64) mine[i] = pl.ids[i];
    wave_sync();
    pl.ids = mine;
  }
  pl.n_uniq = pl.hdr[0];
  pl.nodes_bytes = pl.hdr[1];
  pl.edges_bytes = pl.hdr[2];
  const Layout L = layout_of(a, r, pl);
  write_fields(a, e, pl, L, r, out + rec_off[r], crc_t, row_t, shift_t);
}

// ------------------------------------------------------------------------------------------
// Typed (heterogeneous) RootedNodeNeighborhood records: the result of a SamplingOp DAG on a typed graph — per op a
// frontier [b, w] and its sampled neighbours [b, w, f] — encoded on the device.  Replaces the per-root assembly of
// GraphDBSampler (scala_spark35/common/src/main/scala/graphdb/GraphDBSampler.scala:45-113: the union of all ops'
// edges and nodes + the root, as sets) and the proto cast / TFRecord write of the task
// (GraphDBNodeAnchorBasedLinkPredictionTask.scala:62-78).  Per root, one workgroup:
//   plan   every (node id, node type) and (src, dst, edge type) the ops produced goes into LDS, is bitonic-sorted and
//          de-duplicated — nodes ascending by (id, type), edges by (src, dst, type): the canonical order of the host
//          assembly (gigl_amd/graphdb_sampler.py) — and written to a per-root scratch segment with the record's size;
//   scan   record offsets (record_scan_kernel);
//   write  headers by one thread per node / edge (offsets from block scans of the field lengths), feature rows by one
//          wave per node, TFRecord framing + CRC-32C as in record_write_kernel.
// Edge features are not carried by this path (a sampler that hydrates typed edge features assembles on the host).
// ------------------------------------------------------------------------------------------
constexpr int TYPED_MAX_OPS = 16, TYPED_MAX_NODE_TYPES = 16, TYPED_MAX_EDGE_TYPES = 16;
constexpr uint32_t TYPED_MAX_ITEMS = 4096;  // candidate nodes / edges per root the LDS sort is sized for
constexpr uint32_t TYPED_BIG_MAX_ITEMS = 1u << 20;  // ... and the sort staged in global scratch (typed_plan_kernel<true>)
constexpr int64_t TYPED_BIG_STAGE_BYTES = 1ll << 30;  // bound of that staging area (it sets the number of workgroups)
constexpr unsigned long long PAD64 = ~0ull;

constexpr uint32_t TYPED_POS_BIT = 0x80000000u;  // class bit of a sorted edge's type word: a positive edge

struct TypedArgs {
  gigl_typed_op ops[TYPED_MAX_OPS];
  int32_t n_ops;
  const uint32_t* roots;
  int32_t root_type;
  gigl_typed_feat feat[TYPED_MAX_NODE_TYPES];
  int32_t n_node_types;
  int32_t frame;
  int32_t kind;    // GIGL_REC_ROOTED_NODE_NEIGHBORHOOD | GIGL_REC_NODE_ANCHOR_LINK_PRED (pos_edges after the graph)
  uint32_t items;  // candidates per root = sum of w * f over the ops
  uint32_t pow2;   // sort size: next power of two >= items + 1
  // per-root scratch
  unsigned long long* u_nodes;  // [b][items + 1]  (id << 32 | type), ascending
  unsigned long long* u_edges;  // [b][items]      (src << 32 | dst), ascending with u_etype as the minor key
  uint32_t* u_etype;            // [b][items]
  uint32_t* u_info;             // [b][8]: distinct nodes, distinct edges (both classes), bytes of the node fields,
                                //         graph body bytes, bytes of the pos_edges fields
  const uint32_t* shift_tbl;
  // more than TYPED_MAX_ITEMS - 1 candidates per root: the sort and the payload offsets are staged in global scratch, one
  // segment per WORKGROUP (a workgroup takes roots blockIdx.x, blockIdx.x + gridDim.x, ...)
  unsigned long long* g_nk;  // [workgroups][pow2]
  unsigned long long* g_ek;  // [workgroups][pow2]
  uint32_t* g_et;            // [workgroups][pow2]
  uint32_t* g_pay;           // [workgroups][items + 2]
  // Edge.feature_values per condensed edge type: the type's edge list as CSR by SOURCE + one fp32 row per edge in its
  // `col` order (feat == NULL: edges of the type carry no features)
  struct EdgeFeat {
    const int64_t* rowptr;
    const uint32_t* col;
    int64_t n_rows;
    const float* feat;
    int32_t d;
  } efeat[TYPED_MAX_EDGE_TYPES];
  int32_t n_edge_types;
};

// position of edge (s -> d) in its type's CSR by source, NONE when the type has no features or the edge is not listed
__device__ __forceinline__ uint32_t typed_edge_pos(const TypedArgs& a, unsigned long long k1, uint32_t cet) {
  if (cet >= (uint32_t)a.n_edge_types || !a.efeat[cet].feat) return NONE;
  const TypedArgs::EdgeFeat& g = a.efeat[cet];
  const uint32_t s = (uint32_t)(k1 >> 32), d = (uint32_t)k1;
  if ((int64_t)s >= g.n_rows) return NONE;
  int64_t lo = g.rowptr[s];
  const int64_t end = g.rowptr[s + 1];
  int64_t hi = end;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (g.col[mid] < d) lo = mid + 1;
    else hi = mid;
  }
  return (lo < end && g.col[lo] == d) ? (uint32_t)lo : NONE;
}

__device__ __forceinline__ uint32_t typed_node_body(const TypedArgs& a, unsigned long long key) {
  const uint32_t id = (uint32_t)(key >> 32), t = (uint32_t)key;
  uint32_t n = 1 + vlen(t);  // condensed_node_type is `optional`: written even when 0
  if (id) n += 1 + vlen(id);
  const int32_t d = t < (uint32_t)a.n_node_types && a.feat[t].x ? a.feat[t].d : 0;
  if (d > 0) n += 1 + vlen(4u * (uint32_t)d) + 4u * (uint32_t)d;
  return n;
}
__device__ __forceinline__ uint32_t typed_edge_body(const TypedArgs& a, unsigned long long k1, uint32_t cet) {
  const uint32_t s = (uint32_t)(k1 >> 32), d = (uint32_t)k1;
  uint32_t n = 1 + vlen(cet);
  if (s) n += 1 + vlen(s);
  if (d) n += 1 + vlen(d);
  if (typed_edge_pos(a, k1, cet) != NONE) {
    const uint32_t de = (uint32_t)a.efeat[cet].d;
    n += 1 + vlen(4u * de) + 4u * de;
  }
  return n;
}

// bitonic sort of n = 2^k (key, minor) pairs in LDS, ascending by (key, minor); minor may be null
__device__ void lds_bitonic(unsigned long long* key, uint32_t* minor, uint32_t n) {
  for (uint32_t k = 2; k <= n; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t p = i ^ j;
        if (p > i) {
          const unsigned long long a = key[i], b = key[p];
          const uint32_t ma = minor ? minor[i] : 0u, mb = minor ? minor[p] : 0u;
          const bool gt = a > b || (a == b && ma > mb);
          if (gt == ((i & k) == 0)) {
            key[i] = b;
            key[p] = a;
            if (minor) {
              minor[i] = mb;
              minor[p] = ma;
            }
          }
        }
      }
      __syncthreads();
    }
  }
}

// one compare-exchange pass of the bitonic network at distance j inside runs of k; idx0 = position of element 0 in the
// whole sequence (it decides the run's direction)
__device__ __forceinline__ void bitonic_pass(unsigned long long* key, uint32_t* minor, uint32_t n, uint32_t k, uint32_t j,
                                             uint32_t idx0) {
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t p = i ^ j;
    if (p > i) {
      const unsigned long long a = key[i], b = key[p];
      const uint32_t ma = minor ? minor[i] : 0u, mb = minor ? minor[p] : 0u;
      const bool gt = a > b || (a == b && ma > mb);
      if (gt == (((idx0 + i) & k) == 0)) {
        key[i] = b;
        key[p] = a;
        if (minor) {
          minor[i] = mb;
          minor[p] = ma;
        }
      }
    }
  }
  __syncthreads();
}

// the same network over a sequence in GLOBAL scratch (n = 2^k >= 2 * TYPED_MAX_ITEMS): passes at distances below the
// chunk size run on chunks staged in LDS (lk / lm: TYPED_MAX_ITEMS entries), the few longer ones in place
__device__ void big_bitonic(unsigned long long* key, uint32_t* minor, uint32_t n, unsigned long long* lk, uint32_t* lm) {
  constexpr uint32_t CH = TYPED_MAX_ITEMS;
  uint32_t* const lminor = minor ? lm : nullptr;
  for (uint32_t k0 = CH; k0 <= n; k0 <<= 1) {
    // k0 == CH: every run up to the chunk size, chunk by chunk; above: the long passes of run k0, then the short ones
    if (k0 > CH)
      for (uint32_t j = k0 >> 1; j >= CH; j >>= 1) bitonic_pass(key, minor, n, k0, j, 0);
    for (uint32_t c = 0; c < n; c += CH) {
      for (uint32_t i = threadIdx.x; i < CH; i += blockDim.x) {
        lk[i] = key[c + i];
        if (minor) lm[i] = minor[c + i];
      }
      __syncthreads();
      if (k0 == CH) {
        for (uint32_t k = 2; k <= CH; k <<= 1)
          for (uint32_t j = k >> 1; j > 0; j >>= 1) bitonic_pass(lk, lminor, CH, k, j, c);
      } else {
        for (uint32_t j = CH >> 1; j > 0; j >>= 1) bitonic_pass(lk, lminor, CH, k0, j, c);
      }
      for (uint32_t i = threadIdx.x; i < CH; i += blockDim.x) {
        key[c + i] = lk[i];
        if (minor) minor[c + i] = lm[i];
      }
      __syncthreads();
    }
  }
}

template <bool BIG>
__global__ __launch_bounds__(256) void typed_plan_kernel(TypedArgs a, int64_t* rec_size, uint32_t n_records) {
  __shared__ uint32_t s_w[8];
  const uint32_t tid = threadIdx.x, P = a.pow2;
  // [P] each: in LDS, or (BIG) this workgroup's segments of the global staging area
  unsigned long long* nk = BIG ? a.g_nk + (int64_t)blockIdx.x * P : (unsigned long long*)s_dyn;
  unsigned long long* ek = BIG ? a.g_ek + (int64_t)blockIdx.x * P : nk + P;
  uint32_t* et = BIG ? a.g_et + (int64_t)blockIdx.x * P : (uint32_t*)(ek + P);
  for (uint32_t r = blockIdx.x; r < n_records; r += gridDim.x) {
  for (uint32_t i = tid; i < P; i += 256) {
    nk[i] = PAD64;
    ek[i] = PAD64;
    et[i] = 0xFFFFFFFFu;
  }
  __syncthreads();
  const uint32_t root = a.roots[r];
  if (tid == 0) nk[0] = ((unsigned long long)root << 32) | (uint32_t)a.root_type;
  uint32_t base = 0;
  for (int o = 0; o < a.n_ops; ++o) {
    const gigl_typed_op& op = a.ops[o];
    const uint32_t wf = (uint32_t)(op.w * op.f);
    for (uint32_t idx = tid; idx < wf; idx += 256) {
      const uint32_t q = idx / (uint32_t)op.f;
      const uint32_t fr = op.frontier[(int64_t)r * op.w + q];
      const uint32_t v = op.nbr[((int64_t)r * op.w + q) * op.f + (idx - q * (uint32_t)op.f)];
      if (fr == GIGL_INVALID || v == GIGL_INVALID) continue;
      nk[1 + base + idx] = ((unsigned long long)v << 32) | (uint32_t)op.result_node_type;
      ek[base + idx] = (op.outgoing & 1) ? ((unsigned long long)fr << 32) | v : ((unsigned long long)v << 32) | fr;
      // (bit 31: an edge of a positive-edge op — listed as a pos_edges field, not inside the neighbourhood graph)
      et[base + idx] = (uint32_t)op.condensed_edge_type | ((op.outgoing & GIGL_TYPED_OP_POSITIVE) ? TYPED_POS_BIT : 0u);
    }
    base += wf;
  }
  __syncthreads();
  if (BIG) {
    big_bitonic(nk, nullptr, P, (unsigned long long*)s_dyn, nullptr);
    big_bitonic(ek, et, P, (unsigned long long*)s_dyn, (uint32_t*)((unsigned long long*)s_dyn + TYPED_MAX_ITEMS));
  } else {
    lds_bitonic(nk, nullptr, P);
    lds_bitonic(ek, et, P);
  }
  // distinct items -> scratch, field bytes summed
  unsigned long long* un = a.u_nodes + (int64_t)r * (a.items + 1);
  unsigned long long* ue = a.u_edges + (int64_t)r * a.items;
  uint32_t* ut = a.u_etype + (int64_t)r * a.items;
  uint32_t n_nodes = 0, n_edges = 0, node_bytes = 0, edge_bytes = 0, pos_bytes = 0;
  for (uint32_t c0 = 0; c0 < P; c0 += 256) {
    const uint32_t i = c0 + tid;
    const unsigned long long k = i < P ? nk[i] : PAD64;
    const bool keep = k != PAD64 && (i == 0 || nk[i - 1] != k);
    uint32_t tot, tot_b;
    const uint32_t pos = block_exscan(keep ? 1u : 0u, s_w, tot);
    const uint32_t fb = keep ? field_len(typed_node_body(a, k)) : 0u;
    block_exscan(fb, s_w, tot_b);
    if (keep) un[n_nodes + pos] = k;
    n_nodes += tot;
    node_bytes += tot_b;
  }
  for (uint32_t c0 = 0; c0 < P; c0 += 256) {
    const uint32_t i = c0 + tid;
    const unsigned long long k = i < P ? ek[i] : PAD64;
    const uint32_t t = i < P ? et[i] : 0u;
    const bool keep = k != PAD64 && (i == 0 || ek[i - 1] != k || et[i - 1] != t);
    uint32_t tot, tot_b, tot_p;
    const uint32_t pos = block_exscan(keep ? 1u : 0u, s_w, tot);
    const uint32_t fb = keep ? field_len(typed_edge_body(a, k, t & ~TYPED_POS_BIT)) : 0u;
    block_exscan((t & TYPED_POS_BIT) ? 0u : fb, s_w, tot_b);
    block_exscan((t & TYPED_POS_BIT) ? fb : 0u, s_w, tot_p);
    if (keep) {
      ue[n_edges + pos] = k;
      ut[n_edges + pos] = t;
    }
    n_edges += tot;
    edge_bytes += tot_b;
    pos_bytes += tot_p;
  }
  if (tid == 0) {
    const uint32_t graph_body = node_bytes + edge_bytes;
    const uint32_t root_body = typed_node_body(a, ((unsigned long long)root << 32) | (uint32_t)a.root_type);
    uint32_t* info = a.u_info + (int64_t)r * 8;
    info[0] = n_nodes;
    info[1] = n_edges;
    info[2] = node_bytes;
    info[3] = graph_body;
    info[4] = pos_bytes;
    rec_size[r] = (int64_t)field_len(root_body) + field_len(graph_body) + pos_bytes + (a.frame ? 16 : 0);
  }
  __syncthreads();  // (the staging buffers and s_w are reused by the workgroup's next root)
  }
}

__device__ __forceinline__ uint8_t* typed_write_node_header(const TypedArgs& a, uint8_t* q, uint8_t tag,
                                                            unsigned long long key) {
  const uint32_t id = (uint32_t)(key >> 32), t = (uint32_t)key;
  *q++ = tag;
  q = put_varint(q, typed_node_body(a, key));
  if (id) {
    *q++ = 0x08;
    q = put_varint(q, id);
  }
  *q++ = 0x10;
  q = put_varint(q, t);
  const int32_t d = t < (uint32_t)a.n_node_types && a.feat[t].x ? a.feat[t].d : 0;
  if (d > 0) {
    *q++ = 0x1A;
    q = put_varint(q, 4u * (uint32_t)d);
  }
  return q;  // the float payload starts here
}

template <bool BIG>
__global__ __launch_bounds__(256) void typed_write_kernel(TypedArgs a, const int64_t* rec_off, const int32_t* status,
                                                          uint8_t* out, uint32_t n_records) {
  __shared__ uint32_t s_w[8];
  __shared__ uint32_t s_x[4];
  __shared__ uint32_t crc_t[1024];
  if (*status != 0) return;
  const uint32_t tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  // [items + 2]: offset of every node's float payload from the record start (LDS, or the workgroup's global segment)
  uint32_t* pay = BIG ? a.g_pay + (int64_t)blockIdx.x * (a.items + 2) : (uint32_t*)s_dyn;
  {
    uint32_t c = tid;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
    crc_t[tid] = c;
  }
  __syncthreads();
  for (int j = 1; j < 4; ++j) {
    const uint32_t prev = crc_t[(j - 1) * 256 + tid];
    crc_t[j * 256 + tid] = (prev >> 8) ^ crc_t[prev & 0xFF];
    __syncthreads();
  }
  for (uint32_t r = blockIdx.x; r < n_records; r += gridDim.x) {
  const uint32_t* info = a.u_info + (int64_t)r * 8;
  const uint32_t n_nodes = info[0], n_edges = info[1], node_bytes = info[2], graph_body = info[3], pos_bytes = info[4];
  const unsigned long long* un = a.u_nodes + (int64_t)r * (a.items + 1);
  const unsigned long long* ue = a.u_edges + (int64_t)r * a.items;
  const uint32_t* ut = a.u_etype + (int64_t)r * a.items;
  const unsigned long long root_key = ((unsigned long long)a.roots[r] << 32) | (uint32_t)a.root_type;
  const uint32_t root_body = typed_node_body(a, root_key);
  const uint64_t payload_len = (uint64_t)field_len(root_body) + field_len(graph_body) + pos_bytes;
  uint8_t* const rec = out + rec_off[r];
  uint8_t* const payload = rec + (a.frame ? 12 : 0);
  uint8_t* const graph_hdr = payload + field_len(root_body);
  uint8_t* const graph = graph_hdr + 1 + vlen(graph_body);
  uint8_t* const edges = graph + node_bytes;
  uint8_t* const pos_edges = graph + graph_body;  // NodeAnchorBasedLinkPredictionSample.pos_edges = 4, after the graph
  uint8_t* const payload_end = payload + payload_len;
  if (tid == 0) {
    if (a.frame) {
      uint32_t c = 0xFFFFFFFFu;
      for (int b = 0; b < 8; ++b) {
        const uint32_t byte = (uint32_t)((payload_len >> (8 * b)) & 0xFF);
        rec[b] = (uint8_t)byte;
        c = crc_byte(crc_t, c, byte);
      }
      const uint32_t m = mask_crc(c ^ 0xFFFFFFFFu);
      for (int b = 0; b < 4; ++b) rec[8 + b] = (uint8_t)(m >> (8 * b));
    }
    uint8_t* q = typed_write_node_header(a, payload, 0x0A, root_key);  // root_node = 1
    pay[n_nodes] = (uint32_t)(q - rec);
    q = graph_hdr;
    *q++ = a.kind == GIGL_REC_NODE_ANCHOR_LINK_PRED ? 0x1A : 0x12;  // neighborhood = 3 | 2
    put_varint(q, graph_body);
  }
  // Graph.nodes = 2: one thread per node writes the header; offsets from block scans of the field lengths
  uint32_t run = 0;
  for (uint32_t c0 = 0; c0 < n_nodes; c0 += 256) {
    const uint32_t i = c0 + tid;
    const unsigned long long k = i < n_nodes ? un[i] : 0ull;
    uint32_t tot;
    const uint32_t off = block_exscan(i < n_nodes ? field_len(typed_node_body(a, k)) : 0u, s_w, tot);
    if (i < n_nodes) pay[i] = (uint32_t)(typed_write_node_header(a, graph + run + off, 0x12, k) - rec);
    run += tot;
  }
  run = 0;
  uint32_t run_p = 0;
  for (uint32_t c0 = 0; c0 < n_edges; c0 += 256) {  // Graph.edges = 3; the positive-edge class: pos_edges = 4
    const uint32_t i = c0 + tid;
    const unsigned long long k = i < n_edges ? ue[i] : 0ull;
    const uint32_t tc = i < n_edges ? ut[i] : 0u;
    const uint32_t t = tc & ~TYPED_POS_BIT;
    const bool is_pos = (tc & TYPED_POS_BIT) != 0;
    const uint32_t fl = i < n_edges ? field_len(typed_edge_body(a, k, t)) : 0u;
    uint32_t tot, tot_p;
    const uint32_t off = block_exscan(is_pos ? 0u : fl, s_w, tot);
    const uint32_t off_p = block_exscan(is_pos ? fl : 0u, s_w, tot_p);
    if (i < n_edges) {
      uint8_t* q = is_pos ? pos_edges + run_p + off_p : edges + run + off;
      const uint32_t s = (uint32_t)(k >> 32), d = (uint32_t)k;
      *q++ = is_pos ? 0x22 : 0x1A;
      q = put_varint(q, typed_edge_body(a, k, t));
      if (s) {
        *q++ = 0x08;
        q = put_varint(q, s);
      }
      if (d) {
        *q++ = 0x10;
        q = put_varint(q, d);
      }
      *q++ = 0x18;
      q = put_varint(q, t);
      const uint32_t pe = typed_edge_pos(a, k, t);
      if (pe != NONE) {  // feature_values = 4, packed floats (a few words per edge: written by the edge's thread)
        const uint32_t de = (uint32_t)a.efeat[t].d;
        *q++ = 0x22;
        q = put_varint(q, 4u * de);
        const float* fr = a.efeat[t].feat + (int64_t)pe * de;
        for (uint32_t w2 = 0; w2 < de; ++w2) *(u32_unaligned*)(q + 4u * w2) = __float_as_uint(fr[w2]);
      }
    }
    run += tot;
    run_p += tot_p;
  }
  __syncthreads();
  // feature rows: one wave per node (entry n_nodes = the root_node field)
  for (uint32_t i = (uint32_t)w; i <= n_nodes; i += 4) {
    const unsigned long long k = i < n_nodes ? un[i] : root_key;
    const uint32_t id = (uint32_t)(k >> 32), t = (uint32_t)k;
    if (t >= (uint32_t)a.n_node_types || !a.feat[t].x || a.feat[t].d <= 0) continue;
    const float* src = a.feat[t].x + (int64_t)id * a.feat[t].d;
    uint8_t* dst = rec + pay[i];
    for (int e = lane; e < a.feat[t].d; e += 64) *(u32_unaligned*)(dst + 4 * e) = __float_as_uint(src[e]);  // (unaligned stores)
  }
  if (a.frame) {
  __threadfence_block();
  __syncthreads();
  {  // CRC-32C of the payload (as record_write_kernel: per-thread slices combined by x^(8*bytes after the slice))
    const uint64_t n = payload_len;
    const uint64_t C = ((n + 255) / 256 + 3) & ~3ull;
    const uint64_t lo = min((uint64_t)tid * C, n), hi = min(lo + C, n);
    uint32_t c = tid == 0 ? 0xFFFFFFFFu : 0u;
    const uint8_t* p = payload + lo;
    const uint8_t* const e = payload + hi;
    while (p < e && ((uintptr_t)p & 3u)) c = crc_byte(crc_t, c, *p++);
    for (; p + 4 <= e; p += 4) c = crc_word(crc_t, c, *(const uint32_t*)p);
    while (p < e) c = crc_byte(crc_t, c, *p++);
    uint32_t part = 0;
    if (c) part = multmodp(x8n_modp(a.shift_tbl, n - hi), c);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part ^= __shfl_xor(part, o, 64);
    if (lane == 0) s_x[w] = part;
    __syncthreads();
    if (tid == 0) {
      const uint32_t m = mask_crc((s_x[0] ^ s_x[1] ^ s_x[2] ^ s_x[3]) ^ 0xFFFFFFFFu);
      for (int b = 0; b < 4; ++b) payload_end[b] = (uint8_t)(m >> (8 * b));
    }
  }
  }
  __syncthreads();  // (pay / s_w / s_x are reused by the workgroup's next root)
  }
}


uint32_t host_multmodp(uint32_t a, uint32_t b) {
  uint32_t m = 1u << 31, p = 0;
  for (;;) {
    if (a & m) {
      p ^= b;
      if ((a & (m - 1)) == 0) break;
    }
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
  }
  return p;
}

uint32_t next_pow2(uint32_t x) {
  uint32_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// node-stream positions per record: the plan is held in LDS while it fits (PLAN_LDS_BYTES: ~5,000 positions of a
// one-tree record), in a per-workgroup scratch segment beyond that
constexpr int64_t MAX_STREAM = 1 << 20;
constexpr size_t PLAN_LDS_BYTES = 144 * 1024;

int32_t fill_args(gigl_ctx* ctx, const int32_t* fanouts, int32_t hops, const gigl_record_opts* o, RecArgs& a) {
  GIGL_REQUIRE(ctx, fanouts && o, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS, "hops must be in [1,%d]", GIGL_MAX_HOPS);
  GIGL_REQUIRE(ctx, o->kind == GIGL_REC_ROOTED_NODE_NEIGHBORHOOD || o->kind == GIGL_REC_NODE_ANCHOR_LINK_PRED,
               "bad record kind %d", o->kind);
  GIGL_REQUIRE(ctx, o->trees_per_record >= 1, "trees_per_record must be >= 1");
  GIGL_REQUIRE(ctx, o->kind == GIGL_REC_NODE_ANCHOR_LINK_PRED || o->trees_per_record == 1,
               "a RootedNodeNeighborhood record is built from exactly one tree");
  GIGL_REQUIRE(ctx, (o->suffix == nullptr) == (o->suffix_off == nullptr), "suffix and suffix_off go together");
  int64_t s = 1, sum = 0;
  for (int k = 0; k < hops; ++k) {
    GIGL_REQUIRE(ctx, fanouts[k] >= 1 && fanouts[k] <= GIGL_MAX_FANOUT, "fanout[%d]=%d outside [1,%d]", k,
                 fanouts[k], GIGL_MAX_FANOUT);
    s *= fanouts[k];
    sum += s;
    if (sum > MAX_STREAM) break;
    a.fan[k] = fanouts[k];
    a.slots[k] = (int32_t)s;
  }
  if ((sum + 1) * o->trees_per_record > MAX_STREAM)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "record of %lld tree slots exceeds the %lld the record plan holds",
                     (long long)((sum + 1) * o->trees_per_record), (long long)MAX_STREAM);
  a.hops = hops;
  a.trees = o->trees_per_record;
  a.tree_len = (int32_t)sum + 1;
  a.edge_len = (int32_t)sum;
  a.kind = o->kind;
  a.node_type = o->condensed_node_type;
  a.edge_type = o->condensed_edge_type;
  a.frame = o->tfrecord_frame;
  a.emit = o->emit;
  a.suffix = o->suffix;
  a.suffix_off = o->suffix_off;
  a.hash_cap = next_pow2((uint32_t)(3 * a.trees * a.tree_len / 2 + 8));  // load factor 1/3 .. 2/3
  a.ehash_cap = a.trees > 1 ? next_pow2((uint32_t)(3 * a.trees * a.edge_len / 2 + 8)) : 0;
  return GIGL_OK;
}

// x^(8*n) mod P
uint32_t host_x8n(uint64_t n) {
  uint32_t sq = 1u << 30;  // x^1
  for (int k = 0; k < 3; ++k) sq = host_multmodp(sq, sq);  // x^8
  uint32_t p = 1u << 31;  // x^0
  for (; n; n >>= 1) {
    if (n & 1) p = host_multmodp(sq, p);
    sq = host_multmodp(sq, sq);
  }
  return p;
}

// x^(8*b*256^j) mod P for j = 0..2, b = 0..255 (ctx-owned device table, built once)
int32_t ensure_shift_table(gigl_ctx* ctx) {
  if (ctx->crc_shift_tbl) return GIGL_OK;
  std::vector<uint32_t> t(768);
  uint32_t x2n[32];
  uint32_t p = 1u << 30;  // x^1
  x2n[0] = p;
  for (int k = 1; k < 32; ++k) x2n[k] = p = host_multmodp(p, p);
  for (int j = 0; j < 3; ++j) {
    const uint32_t step = x2n[3 + 8 * j];  // x^(8*256^j)
    uint32_t acc = 1u << 31;               // x^0
    for (int b = 0; b < 256; ++b) {
      t[j * 256 + b] = acc;
      acc = host_multmodp(step, acc);
    }
  }
  GIGL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->crc_shift_tbl, 768 * 4));
  GIGL_HIP_CHECK(ctx, hipMemcpy(ctx->crc_shift_tbl, t.data(), 768 * 4, hipMemcpyHostToDevice));
  return GIGL_OK;
}

// crc_t | row_t | shift_t as the write pass holds them in LDS (ctx-owned device table, rebuilt when the row width changes)
int32_t ensure_enc_tables(gigl_ctx* ctx, int32_t d, uint32_t x_row) {
  if (ctx->enc_tables && ctx->enc_tables_d == d) return GIGL_OK;
  std::vector<uint32_t> t(2816);
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
    t[i] = c;
  }
  for (int j = 1; j < 4; ++j)
    for (uint32_t i = 0; i < 256; ++i) {
      const uint32_t prev = t[(j - 1) * 256 + i];
      t[j * 256 + i] = (prev >> 8) ^ t[prev & 0xFF];
    }
  for (uint32_t i = 0; i < 1024; ++i) t[1024 + i] = host_multmodp(x_row, (i & 255u) << (8 * (i >> 8)));
  GIGL_HIP_CHECK(ctx, hipMemcpy(t.data() + 2048, ctx->crc_shift_tbl, 768 * 4, hipMemcpyDeviceToHost));
  if (!ctx->enc_tables) GIGL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->enc_tables, 2816 * 4));
  else GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // (a write pass in flight still reads the old ones)
  GIGL_HIP_CHECK(ctx, hipMemcpy(ctx->enc_tables, t.data(), 2816 * 4, hipMemcpyHostToDevice));
  ctx->enc_tables_d = d;
  return GIGL_OK;
}

int hvlen(uint64_t v) {
  int n = 1;
  while (v >= 128) {
    v >>= 7;
    ++n;
  }
  return n;
}

}  // namespace

void gigl_scan_i64(gigl_ctx* ctx, const int64_t* sizes, int64_t n, int64_t cap, int64_t* out, int32_t* status) {
  hipLaunchKernelGGL(record_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, sizes, n, cap, out, status);
}

extern "C" {

int32_t gigl_records_capacity(const int32_t* fanouts, int32_t hops, int32_t d, const gigl_record_opts* opts,
                              int64_t n_records, int64_t suffix_total, int64_t* bytes) {
  if (!fanouts || !opts || !bytes || hops < 1 || hops > GIGL_MAX_HOPS || d < 0 || n_records < 0 ||
      opts->trees_per_record < 1)
    return GIGL_E_INVALID_ARG;
  int64_t s = 1, sum = 0;
  for (int k = 0; k < hops; ++k) {
    if (fanouts[k] < 1) return GIGL_E_INVALID_ARG;
    s *= fanouts[k];
    sum += s;
  }
  const int64_t node_body = 6 + 6 + (d > 0 ? 1 + hvlen(4ull * d) + 4ll * d : 0);
  const int64_t node_field = 1 + hvlen(node_body) + node_body;
  int64_t de = opts->edge_feat ? opts->edge_feat->d : 0;
  auto edge_bytes = [](int64_t k) { return 1 + 5 + 6 + 6 + 6 + (k > 0 ? 1 + hvlen(4ull * k) + 4 * k : 0); };
  const int64_t edge_field = edge_bytes(de);
  if (opts->pos_edge_feat && opts->pos_edge_feat->d > de) de = opts->pos_edge_feat->d;
  if (opts->neg_edge_feat && opts->neg_edge_feat->d > de) de = opts->neg_edge_feat->d;
  const int64_t label_field = edge_bytes(de);
  const int64_t t = opts->trees_per_record;
  const int64_t graph = t * ((sum + 1) * node_field + sum * edge_field);
  const int64_t per = node_field + 1 + hvlen(graph) + graph + (t - 1) * label_field + 16;
  *bytes = n_records * per + suffix_total;
  return GIGL_OK;
}

int32_t gigl_records_encode(gigl_ctx* ctx, const uint32_t* tree_roots, const gigl_tree* tree, gigl_feat* feat,
                            const gigl_record_opts* opts, int64_t n_records, uint8_t* out, int64_t out_cap,
                            int64_t* rec_off, int32_t* status) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, tree && opts && rec_off && status && (out || out_cap == 0), "null argument");
  GIGL_REQUIRE(ctx, n_records >= 0 && n_records < ((int64_t)1 << 31) && out_cap >= 0, "bad sizes");
  RecArgs a{};
  int32_t rc = fill_args(ctx, tree->fanouts, tree->hops, opts, a);
  if (rc != GIGL_OK) return rc;
  GIGL_REQUIRE(ctx, (int64_t)tree->b == n_records * a.trees, "tree holds %d roots, %lld records x %d trees expected",
               tree->b, (long long)n_records, a.trees);
  GIGL_REQUIRE(ctx, tree_roots || n_records == 0, "null roots");
  a.roots = tree_roots;
  for (int k = 0; k < a.hops; ++k) {
    GIGL_REQUIRE(ctx, tree->nbr[k], "tree buffers for hop %d are null", k);
    a.nbr[k] = tree->nbr[k];
  }
  if (feat) {
    a.feat = feat->rows;
    a.d = feat->d;
    a.feat_dtype = feat->dtype;
    a.feat_n = feat->n;
  }
  if (opts->edge_feat && opts->edge_feat->d > 0) {
    GIGL_REQUIRE(ctx, opts->graph, "edge features need the resident graph (opts->graph) for the edge lookup");
    GIGL_REQUIRE(ctx, opts->edge_feat->dtype == GIGL_DTYPE_F32, "edge features must be fp32");
    GIGL_REQUIRE(ctx, opts->edge_feat->n == opts->graph->e && opts->graph->e < ((int64_t)1 << 32) - 1,
                 "edge feature table of %lld rows for a graph of %lld edges", (long long)opts->edge_feat->n,
                 (long long)opts->graph->e);
    a.efeat = (const float*)opts->edge_feat->rows;
    a.de = opts->edge_feat->d;
    a.g_rowptr = opts->graph->rowptr;
    a.g_col = opts->graph->col;
    a.g_n = opts->graph->n;
  }
  GIGL_REQUIRE(ctx, opts->n_neg_trees >= 0 && opts->n_neg_trees < a.trees, "n_neg_trees %d outside [0,%d)",
               opts->n_neg_trees, a.trees);
  GIGL_REQUIRE(ctx, opts->n_neg_trees == 0 || opts->kind == GIGL_REC_NODE_ANCHOR_LINK_PRED,
               "hard negatives belong to NodeAnchorBasedLinkPredictionSample records");
  a.neg_trees = opts->n_neg_trees;
  for (int which = 0; which < 2; ++which) {
    gigl_graph* lg = which ? opts->neg_edges_graph : opts->pos_edges_graph;
    gigl_feat* lf = which ? opts->neg_edge_feat : opts->pos_edge_feat;
    if (!lf || lf->d <= 0) continue;
    GIGL_REQUIRE(ctx, lg, "label-edge features need the label edge list as a CSR-by-source graph");
    GIGL_REQUIRE(ctx, lf->dtype == GIGL_DTYPE_F32 && lf->n == lg->e, "label-edge feature table does not match its graph");
    a.lab[which].rowptr = lg->rowptr;
    a.lab[which].col = lg->col;
    a.lab[which].n = lg->n;
    a.lab[which].feat = (const float*)lf->rows;
    a.lab[which].de = lf->d;
  }
  // user-defined positives without features: pos_edges must not pick up the MAIN edge table
  if (opts->pos_edges_graph && !a.lab[0].rowptr) {
    a.lab[0].rowptr = opts->pos_edges_graph->rowptr;
    a.lab[0].col = opts->pos_edges_graph->col;
    a.lab[0].n = opts->pos_edges_graph->n;
    a.lab[0].de = 0;
  }
  a.n_records = n_records;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  rc = ensure_shift_table(ctx);
  if (rc != GIGL_OK) return rc;
  a.shift_tbl = ctx->crc_shift_tbl;
  {  // offsets inside a record are 27-bit (Plan::fld) and its checksum shifts are tabulated below 2^24 bytes
    int64_t one = 0;
    gigl_records_capacity(tree->fanouts, tree->hops, a.d, opts, 1, 0, &one);
    GIGL_REQUIRE(ctx, one < ((int64_t)1 << 24), "a record of up to %lld bytes: the encoder holds records below 16 MiB",
                 (long long)one);
  }
  EncArgs e{};
  e.x_row = host_x8n(4ull * (uint64_t)a.d);
  if (feat && a.frame && a.d > 0) {
    const uint32_t* rc_tbl = nullptr;
    rc = gigl_features_row_crc(ctx, feat, &rc_tbl);
    if (rc != GIGL_OK) return rc;
    e.row_crc = rc_tbl;
  }
  rc = ensure_enc_tables(ctx, a.d, e.x_row);
  if (rc != GIGL_OK) return rc;
  e.tables = ctx->enc_tables;
  // waves per workgroup: as many plans as fit LDS (4, 2 or 1); plans beyond LDS are built in scratch
  const size_t plan = plan_bytes(a), kept = kept_bytes(a);
  const bool big = plan > PLAN_LDS_BYTES;
  // (plan workgroups of ONE wave: LDS is granted per workgroup, and single-wave workgroups fill the CUs' LDS with 18 plans
  // of a [25,10] record instead of 16 in groups of four — 987 -> 931 us per 32,768 records, nothing at 4,096)
  int wp = 1, wf = 4;
  while (!big && (size_t)wp * plan > PLAN_LDS_BYTES) wp >>= 1;
  const size_t stream_bytes = ((size_t)4 * a.trees * a.tree_len + 15) & ~(size_t)15;  // (below 17 KB whenever !big)
  const size_t lds_p = big ? 0 : (size_t)wp * plan, lds_w = big ? 0 : (size_t)wf * stream_bytes;
  if (lds_p > 48 * 1024)
    GIGL_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)record_plan_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
  if (lds_w > 32 * 1024)
    GIGL_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)record_write_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w));
  e.plan_stride = big ? plan : kept;
  e.lds_stride = 0;
  const int64_t nb = n_records > 0 ? n_records : 1;
  const int64_t plans_bytes = nb * (int64_t)e.plan_stride;
  rc = gigl_arena_reset(ctx, (n_records + 1) * 8 + plans_bytes + 1024);
  if (rc != GIGL_OK) return rc;
  e.rec_size = (int64_t*)gigl_arena_alloc(ctx, (n_records + 1) * 8);
  e.plans = (unsigned char*)gigl_arena_alloc(ctx, plans_bytes);
  if (!e.rec_size || !e.plans) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  if (n_records > 0) {
    const unsigned gp = (unsigned)((n_records + wp - 1) / wp);
    e.lds_stride = plan;
    // (a workgroup per record while the call is latency-bound; GIGL_REC_PLAN_WG = 0 | 1 forces either shape: A/B)
    static const int wg_knob = getenv("GIGL_REC_PLAN_WG") ? atoi(getenv("GIGL_REC_PLAN_WG")) : -1;
    const bool plan_wg = !big && plan <= 64 * 1024 && (wg_knob >= 0 ? wg_knob != 0 : n_records <= 6144);
    if (plan_wg && plan > 48 * 1024)
      GIGL_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)record_plan_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)plan));
    if (big) hipLaunchKernelGGL(record_plan_kernel<true>, dim3(gp), dim3(64 * wp), 0, ctx->stream, a, e);
    else if (plan_wg) hipLaunchKernelGGL(record_plan_wg_kernel, dim3((unsigned)n_records), dim3(256), plan, ctx->stream, a, e);
    else hipLaunchKernelGGL(record_plan_kernel<false>, dim3(gp), dim3(64 * wp), lds_p, ctx->stream, a, e);
  }
  hipLaunchKernelGGL(record_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const int64_t*)e.rec_size, n_records,
                     out_cap, rec_off, status);
  if (n_records > 0) {
    const unsigned gw = (unsigned)((n_records + wf - 1) / wf + (a.d > 0 ? n_records * ROWS_WGS : 0));
    e.lds_stride = stream_bytes;
    if (big)
      hipLaunchKernelGGL(record_write_kernel<true>, dim3(gw), dim3(256), 0, ctx->stream, a, e, (const int64_t*)rec_off,
                         (const int32_t*)status, out, (uint32_t)wf);
    else
      hipLaunchKernelGGL(record_write_kernel<false>, dim3(gw), dim3(256), lds_w, ctx->stream, a, e,
                         (const int64_t*)rec_off, (const int32_t*)status, out, (uint32_t)wf);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_features_row_crc(gigl_ctx* ctx, gigl_feat* f, const uint32_t** table) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, f && table, "null argument");
  std::lock_guard<std::mutex> lock(f->row_crc_mu);
  if (!f->row_crc) {
    GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int32_t rc = ensure_shift_table(ctx);
    if (rc != GIGL_OK) return rc;
    uint32_t* t = nullptr;
    GIGL_HIP_CHECK(ctx, hipMalloc((void**)&t, (size_t)(f->n > 0 ? f->n : 1) * 4));
    if (f->n > 0) {
      const int64_t blocks = (f->n + 3) / 4;
      hipLaunchKernelGGL(row_crc_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, ctx->stream,
                         (const void*)f->rows, f->dtype, f->n, f->d, (const uint32_t*)ctx->crc_shift_tbl, t);
    }
    // finished before anybody else's stream can use it (once per table; not inside a stream capture)
    const hipError_t err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) {
      hipFree(t);
      return gigl_fail(ctx, GIGL_E_HIP, "row CRC table: %s", hipGetErrorString(err));
    }
    f->row_crc = t;
  }
  *table = f->row_crc;
  return GIGL_OK;
}

int32_t gigl_typed_records_capacity(const gigl_typed_op* ops, int32_t n_ops, const gigl_typed_feat* feats,
                                    int32_t n_node_types, const gigl_typed_edge_feat* efeats, int32_t n_edge_types,
                                    int64_t n_records, int32_t tfrecord_frame, int64_t* bytes) {
  if (!ops || n_ops < 1 || n_ops > TYPED_MAX_OPS || n_node_types < 1 || n_node_types > TYPED_MAX_NODE_TYPES || !bytes)
    return GIGL_E_INVALID_ARG;
  int64_t items = 0, dmax = 0;
  for (int o = 0; o < n_ops; ++o) items += (int64_t)ops[o].w * ops[o].f;
  for (int t = 0; t < n_node_types; ++t)
    if (feats && feats[t].x && feats[t].d > dmax) dmax = feats[t].d;
  // node field <= 2 + 6 + 6 + 6 + 4 d (+ length bytes), edge field <= 2 + 6 + 6 + 6; root field, graph header, frame
  int64_t demax = 0;
  for (int t = 0; efeats && t < n_edge_types; ++t)
    if (efeats[t].feat && efeats[t].d > demax) demax = efeats[t].d;
  const int64_t node = 26 + 4 * dmax, edge = 26 + 4 * demax;
  *bytes = n_records * ((items + 2) * node + items * edge + 16 + (tfrecord_frame ? 16 : 0));
  return GIGL_OK;
}

int32_t gigl_typed_records_encode(gigl_ctx* ctx, const uint32_t* roots, int32_t root_node_type, const gigl_typed_op* ops,
                                  int32_t n_ops, const gigl_typed_feat* feats, int32_t n_node_types,
                                  const gigl_typed_edge_feat* efeats, int32_t n_edge_types, int64_t n_records,
                                  int32_t tfrecord_frame, uint8_t* out, int64_t out_cap, int64_t* rec_off,
                                  int32_t* status) {
  return gigl_typed_samples_encode(ctx, GIGL_REC_ROOTED_NODE_NEIGHBORHOOD, roots, root_node_type, ops, n_ops, feats,
                                   n_node_types, efeats, n_edge_types, n_records, tfrecord_frame, out, out_cap, rec_off,
                                   status);
}

int32_t gigl_typed_samples_encode(gigl_ctx* ctx, int32_t kind, const uint32_t* roots, int32_t root_node_type,
                                  const gigl_typed_op* ops, int32_t n_ops, const gigl_typed_feat* feats,
                                  int32_t n_node_types, const gigl_typed_edge_feat* efeats, int32_t n_edge_types,
                                  int64_t n_records, int32_t tfrecord_frame, uint8_t* out, int64_t out_cap,
                                  int64_t* rec_off, int32_t* status) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, kind == GIGL_REC_ROOTED_NODE_NEIGHBORHOOD || kind == GIGL_REC_NODE_ANCHOR_LINK_PRED, "bad kind %d", kind);
  GIGL_REQUIRE(ctx, roots && ops && out && rec_off && status && n_records >= 0, "null argument");
  GIGL_REQUIRE(ctx, n_ops >= 1 && n_ops <= TYPED_MAX_OPS, "between 1 and %d sampling ops", TYPED_MAX_OPS);
  GIGL_REQUIRE(ctx, n_node_types >= 1 && n_node_types <= TYPED_MAX_NODE_TYPES && root_node_type >= 0 &&
                        root_node_type < n_node_types,
               "node types outside [1,%d]", TYPED_MAX_NODE_TYPES);
  TypedArgs a{};
  int64_t items = 0;
  for (int o = 0; o < n_ops; ++o) {
    GIGL_REQUIRE(ctx, ops[o].frontier && ops[o].nbr && ops[o].w >= 1 && ops[o].f >= 1, "op %d has no buffers", o);
    GIGL_REQUIRE(ctx, ops[o].result_node_type >= 0 && ops[o].result_node_type < n_node_types && ops[o].condensed_edge_type >= 0,
                 "op %d has a type outside the metadata", o);
    GIGL_REQUIRE(ctx, !(ops[o].outgoing & GIGL_TYPED_OP_POSITIVE) || kind == GIGL_REC_NODE_ANCHOR_LINK_PRED,
                 "op %d: a positive-edge op needs kind GIGL_REC_NODE_ANCHOR_LINK_PRED", o);
    a.ops[o] = ops[o];
    items += (int64_t)ops[o].w * ops[o].f;
  }
  if (items + 1 > (int64_t)TYPED_BIG_MAX_ITEMS)
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "%lld sampled slots per root: the typed record encoder holds up to %u",
                     (long long)items, TYPED_BIG_MAX_ITEMS - 1);
  a.n_ops = n_ops;
  a.roots = roots;
  a.root_type = root_node_type;
  for (int t = 0; t < n_node_types; ++t)
    if (feats) a.feat[t] = feats[t];
  a.n_node_types = n_node_types;
  GIGL_REQUIRE(ctx, n_edge_types >= 0 && n_edge_types <= TYPED_MAX_EDGE_TYPES, "edge types outside [0,%d]",
               TYPED_MAX_EDGE_TYPES);
  a.n_edge_types = efeats ? n_edge_types : 0;
  for (int t = 0; t < a.n_edge_types; ++t) {
    if (!efeats[t].feat) continue;
    GIGL_REQUIRE(ctx, efeats[t].by_source && efeats[t].d >= 1, "edge type %d: features need the edge list as CSR by source", t);
    a.efeat[t].rowptr = efeats[t].by_source->rowptr;
    a.efeat[t].col = efeats[t].by_source->col;
    a.efeat[t].n_rows = efeats[t].by_source->n;
    a.efeat[t].feat = efeats[t].feat;
    a.efeat[t].d = efeats[t].d;
  }
  a.frame = tfrecord_frame ? 1 : 0;
  a.kind = kind;
  a.items = (uint32_t)items;
  a.pow2 = next_pow2((uint32_t)items + 1);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int32_t rc = ensure_shift_table(ctx);
  if (rc != GIGL_OK) return rc;
  a.shift_tbl = ctx->crc_shift_tbl;
  const int64_t nb = n_records > 0 ? n_records : 1;
  // up to TYPED_MAX_ITEMS - 1 slots per root the sort runs in LDS, a workgroup per root; beyond, in global staging
  // segments owned by a bounded number of workgroups that walk the roots
  const bool big = items + 1 > (int64_t)TYPED_MAX_ITEMS;
  const int64_t wgs = big ? std::min<int64_t>(nb, std::max<int64_t>(64, (TYPED_BIG_STAGE_BYTES / 24) / a.pow2)) : 0;
  rc = gigl_arena_reset(ctx, (n_records + 1) * 8 + nb * ((items + 1) * 8 + items * 12 + 32) + 4096 +
                                 wgs * ((int64_t)a.pow2 * 20 + (items + 2) * 4) + 1024);
  if (rc != GIGL_OK) return rc;
  int64_t* rec_size = (int64_t*)gigl_arena_alloc(ctx, (n_records + 1) * 8);
  a.u_nodes = (unsigned long long*)gigl_arena_alloc(ctx, nb * (items + 1) * 8);
  a.u_edges = (unsigned long long*)gigl_arena_alloc(ctx, nb * (items > 0 ? items : 1) * 8);
  a.u_etype = (uint32_t*)gigl_arena_alloc(ctx, nb * (items > 0 ? items : 1) * 4);
  a.u_info = (uint32_t*)gigl_arena_alloc(ctx, nb * 32);
  if (!rec_size || !a.u_nodes || !a.u_edges || !a.u_etype || !a.u_info) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  if (big) {
    a.g_nk = (unsigned long long*)gigl_arena_alloc(ctx, wgs * (int64_t)a.pow2 * 8);
    a.g_ek = (unsigned long long*)gigl_arena_alloc(ctx, wgs * (int64_t)a.pow2 * 8);
    a.g_et = (uint32_t*)gigl_arena_alloc(ctx, wgs * (int64_t)a.pow2 * 4);
    a.g_pay = (uint32_t*)gigl_arena_alloc(ctx, wgs * (items + 2) * 4);
    if (!a.g_nk || !a.g_ek || !a.g_et || !a.g_pay) return gigl_fail(ctx, GIGL_E_OOM, "arena exhausted");
  }
  const size_t lds_plan = big ? (size_t)TYPED_MAX_ITEMS * 12 : (size_t)a.pow2 * 20;
  const size_t lds_write = big ? 0 : ((size_t)items + 2) * 4;
  const uint32_t nrec = (uint32_t)n_records;
  if (!big && lds_plan > 60 * 1024)
    GIGL_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)typed_plan_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_plan));
  if (n_records > 0) {
    if (big)
      hipLaunchKernelGGL(typed_plan_kernel<true>, dim3((unsigned)wgs), dim3(256), lds_plan, ctx->stream, a, rec_size, nrec);
    else
      hipLaunchKernelGGL(typed_plan_kernel<false>, dim3((unsigned)n_records), dim3(256), lds_plan, ctx->stream, a, rec_size,
                         nrec);
  }
  hipLaunchKernelGGL(record_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, rec_size, n_records, out_cap, rec_off,
                     status);
  if (n_records > 0) {
    if (big)
      hipLaunchKernelGGL(typed_write_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, a, rec_off, status, out,
                         nrec);
    else
      hipLaunchKernelGGL(typed_write_kernel<false>, dim3((unsigned)n_records), dim3(256), lds_write, ctx->stream, a, rec_off,
                         status, out, nrec);
  }
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}


}  // extern "C"
