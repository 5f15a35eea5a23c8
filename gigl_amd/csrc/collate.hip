// collate.hip — serialized training samples (TFRecord payloads) -> one batch graph, host-side native code.
//
// Replaces the per-node / per-edge Python loops of the reference's data loaders (paths relative to the reference
// root; SURVEY.md 8(a) rows T2/T3 — "the CPU bottleneck of the trainer"):
//   GbmlProtosTranslator.graph_data_from_GraphPb           python/gigl/src/common/translators/gbml_protos_translator.py:101-121
//   GraphBuilder.add_node / add_edge / add_graph_data        python/gigl/src/common/graph_builder/abstract_graph_builder.py:16-24,49-150
//   PygGraphBuilder.build                                    python/gigl/src/common/graph_builder/pyg_graph_builder.py:20-69
//   collate functions + coalesce()                           python/gigl/src/training/v1/lib/data_loaders/
//        rooted_node_neighborhood_data_loader.py:78-158, supervised_node_classification_data_loader.py:74-117,
//        node_anchor_based_link_prediction_data_loader.py:90-221
// Semantics kept: global -> local ids in first-seen order (a sample's nodes first, then its edges), a node seen
// again must carry the same features (np.allclose: rtol 1e-5, atol 1e-8), an edge is skipped if its (local src,
// local dst) pair is already present, edges are finally sorted by (src, dst); a positive / hard-negative / root
// that is not in the batch graph is an error.
// Records are parsed in parallel (one thread per slice of the batch), numbering is one sequential pass over
// compact per-record arrays, feature rows are copied in parallel.  This is the path for samples that arrive as
// TFRecords; batches sampled in HBM never leave the device (union.hip).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

struct gigl_collated {
  int32_t feat_dim = 0, edge_dim = 0;
  std::vector<uint32_t> node_ids;
  std::vector<float> x, edge_attr;
  std::vector<int64_t> edge_src, edge_dst;
  std::vector<int64_t> root_local, labels, pos_off, pos_dst, neg_off, neg_dst;
  std::vector<uint8_t> has_label;
};

// typed (heterogeneous) batch: one numbering, feature matrix per condensed node type; one edge list per condensed
// edge type, its endpoints local to the edge type's source / destination node types
struct gigl_collated_typed {
  struct NodeType {
    int32_t feat_dim = 0;
    std::vector<uint32_t> ids;
    std::vector<float> x;
  };
  struct EdgeType {
    int32_t edge_dim = 0;
    std::vector<int64_t> src, dst;
    std::vector<float> attr;
  };
  std::vector<NodeType> nt;
  std::vector<EdgeType> et;
  std::vector<int32_t> root_type;
  std::vector<int64_t> root_local, labels, pos_off, pos_dst, neg_off, neg_dst;
  std::vector<int32_t> pos_type, neg_type;
  std::vector<uint8_t> has_label;
};

namespace {

struct Span {
  const uint8_t* p;
  const uint8_t* e;
};
bool varint(Span& s, uint64_t& v) {
  v = 0;
  for (int shift = 0; shift < 70 && s.p < s.e; shift += 7) {
    const uint8_t b = *s.p++;
    v |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) return true;
  }
  return false;
}
bool field(Span& s, uint32_t& fno, uint32_t& wt, Span& sub, uint64_t& val) {
  uint64_t key;
  if (!varint(s, key)) return false;
  fno = (uint32_t)(key >> 3);
  wt = (uint32_t)(key & 7);
  if (wt == 0) return varint(s, val);
  if (wt == 2) {
    uint64_t ln;
    if (!varint(s, ln) || ln > (uint64_t)(s.e - s.p)) return false;
    sub = Span{s.p, s.p + ln};
    s.p += ln;
    return true;
  }
  const size_t fix = wt == 1 ? 8 : wt == 5 ? 4 : 0;
  if (!fix || (size_t)(s.e - s.p) < fix) return false;
  sub = Span{s.p, s.p + fix};
  s.p += fix;
  return true;
}

struct NodeRef {
  uint32_t id;
  int32_t n;          // floats
  const uint8_t* f;   // packed little-endian floats (record bytes, or the record's side arena)
  int32_t type = 0;   // condensed_node_type (0 when absent)
};
struct EdgeRef {
  uint32_t src, dst;
  int32_t n;          // floats (Edge.feature_values)
  const uint8_t* f;
  int32_t type = 0;   // condensed_edge_type (0 when absent)
};
struct Rec {
  NodeRef root{0, 0, nullptr, 0};
  bool has_root = false, has_graph = false, has_label = false, ok = true;
  int64_t label = 0;
  std::vector<NodeRef> nodes;
  std::vector<EdgeRef> edges;
  std::vector<uint32_t> pos, neg;
  std::vector<int32_t> pos_type, neg_type;  // condensed edge type of every supervision edge
  std::vector<std::vector<uint8_t>> arena;  // features that were not one packed run
};

bool parse_node(Span s, Rec& r, NodeRef& out) {
  out = NodeRef{0, 0, nullptr, 0};
  uint32_t fno, wt;
  Span x{nullptr, nullptr};
  uint64_t v;
  int runs = 0;
  std::vector<uint8_t> acc;
  while (s.p < s.e) {
    if (!field(s, fno, wt, x, v)) return false;
    if (fno == 1 && wt == 0) out.id = (uint32_t)v;
    else if (fno == 2 && wt == 0) out.type = (int32_t)v;
    else if (fno == 3 && (wt == 2 || wt == 5)) {
      if (runs == 0 && wt == 2) {
        out.f = x.p;
        out.n = (int32_t)((x.e - x.p) / 4);
      } else {
        if (runs == 1 && acc.empty() && out.f) acc.assign(out.f, out.f + 4 * (size_t)out.n);
        acc.insert(acc.end(), x.p, x.p + (wt == 5 ? 4 : ((x.e - x.p) / 4) * 4));
      }
      ++runs;
    }
  }
  if (!acc.empty()) {
    r.arena.push_back(std::move(acc));
    out.f = r.arena.back().data();
    out.n = (int32_t)(r.arena.back().size() / 4);
  }
  return true;
}
bool parse_edge(Span s, Rec& r, EdgeRef& out) {
  out = EdgeRef{0, 0, 0, nullptr, 0};
  uint32_t fno, wt;
  Span x{nullptr, nullptr};
  uint64_t v;
  int runs = 0;
  std::vector<uint8_t> acc;
  while (s.p < s.e) {
    if (!field(s, fno, wt, x, v)) return false;
    if (wt == 0) {
      if (fno == 1) out.src = (uint32_t)v;
      else if (fno == 2) out.dst = (uint32_t)v;
      else if (fno == 3) out.type = (int32_t)v;
    } else if (fno == 4 && (wt == 2 || wt == 5)) {
      if (runs == 0 && wt == 2) {
        out.f = x.p;
        out.n = (int32_t)((x.e - x.p) / 4);
      } else {
        if (runs == 1 && acc.empty() && out.f) acc.assign(out.f, out.f + 4 * (size_t)out.n);
        acc.insert(acc.end(), x.p, x.p + (wt == 5 ? 4 : ((x.e - x.p) / 4) * 4));
      }
      ++runs;
    }
  }
  if (!acc.empty()) {
    r.arena.push_back(std::move(acc));
    out.f = r.arena.back().data();
    out.n = (int32_t)(r.arena.back().size() / 4);
  }
  return true;
}
bool parse_graph(Span s, Rec& r) {
  uint32_t fno, wt;
  Span x{nullptr, nullptr};
  uint64_t v;
  while (s.p < s.e) {
    if (!field(s, fno, wt, x, v)) return false;
    if (wt != 2) continue;
    if (fno == 2) {
      NodeRef n;
      if (!parse_node(x, r, n)) return false;
      r.nodes.push_back(n);
    } else if (fno == 3) {
      EdgeRef e;
      if (!parse_edge(x, r, e)) return false;
      r.edges.push_back(e);
    }
  }
  return true;
}
bool parse_record(Span s, int32_t kind, Rec& r) {
  const uint32_t f_graph = kind == GIGL_REC_NODE_ANCHOR_LINK_PRED ? 3 : 2;
  uint32_t fno, wt;
  Span x{nullptr, nullptr};
  uint64_t v;
  while (s.p < s.e) {
    if (!field(s, fno, wt, x, v)) return false;
    if (wt != 2) continue;
    if (fno == 1) {
      if (!parse_node(x, r, r.root)) return false;
      r.has_root = true;
    } else if (fno == f_graph) {
      if (!parse_graph(x, r)) return false;
      r.has_graph = true;
    } else if (kind == GIGL_REC_ROOTED_NODE_NEIGHBORHOOD && fno == 3) {  // root_node_labels: Label{type=1, label=2}
      if (r.has_label) continue;  // the loaders use root_node_labels[0]
      Span l = x, y{nullptr, nullptr};
      uint32_t f2, w2;
      uint64_t lv = 0;
      while (l.p < l.e) {
        if (!field(l, f2, w2, y, v)) return false;
        if (f2 == 2 && w2 == 0) lv = v;
      }
      r.has_label = true;
      r.label = (int64_t)(int32_t)(uint32_t)lv;  // int32 field: sign-extended 64-bit varint
    } else if (kind == GIGL_REC_NODE_ANCHOR_LINK_PRED && (fno == 4 || fno == 2)) {
      EdgeRef e;
      if (!parse_edge(x, r, e)) return false;
      (fno == 4 ? r.pos : r.neg).push_back(e.dst);
      (fno == 4 ? r.pos_type : r.neg_type).push_back(e.type);
    }
  }
  return true;
}

struct IdMap {  // open addressing uint32 -> int32
  std::vector<uint32_t> keys;
  std::vector<int32_t> vals;
  uint32_t mask;
  explicit IdMap(size_t n) {
    size_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    keys.assign(cap, 0);
    vals.assign(cap, -1);
    mask = (uint32_t)cap - 1;
  }
  static uint32_t h(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
  }
  int32_t* slot(uint32_t k) {
    uint32_t s = h(k) & mask;
    while (vals[s] >= 0 && keys[s] != k) s = (s + 1) & mask;
    keys[s] = k;
    return &vals[s];
  }
  int32_t find(uint32_t k) const {
    uint32_t s = h(k) & mask;
    while (vals[s] >= 0) {
      if (keys[s] == k) return vals[s];
      s = (s + 1) & mask;
    }
    return -1;
  }
};

bool allclose(const uint8_t* a, const uint8_t* b, int32_t n) {
  if (memcmp(a, b, 4 * (size_t)n) == 0) return true;
  for (int32_t i = 0; i < n; ++i) {
    float x, y;
    memcpy(&x, a + 4 * i, 4);
    memcpy(&y, b + 4 * i, 4);
    if (!(std::fabs(x - y) <= 1e-8 + 1e-5 * std::fabs((double)y))) return false;  // numpy.allclose defaults
  }
  return true;
}

void set_err(char* err, int32_t cap, const char* fmt, long long a = 0, long long b = 0) {
  if (err && cap > 0) snprintf(err, (size_t)cap, fmt, a, b);
}

}  // namespace

extern "C" {

int32_t gigl_collate_records(const uint8_t* buf, const int64_t* payload_off, const int64_t* payload_len, int64_t b,
                             int32_t kind, int32_t n_threads, gigl_collated** out, char* err, int32_t err_cap) {
  if (!buf || !payload_off || !payload_len || b < 0 || !out) return GIGL_E_INVALID_ARG;
  if (kind != GIGL_REC_ROOTED_NODE_NEIGHBORHOOD && kind != GIGL_REC_NODE_ANCHOR_LINK_PRED) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  std::vector<Rec> recs((size_t)b);
  auto parse = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i)
      recs[i].ok = parse_record(Span{buf + payload_off[i], buf + payload_off[i] + payload_len[i]}, kind, recs[i]);
  };
  auto parallel = [&](auto fn, int64_t n) {
    if (n_threads == 1 || n < 64) {
      fn(0, n);
      return;
    }
    std::vector<std::thread> th;
    const int64_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
      const int64_t lo = t * per, hi = std::min(n, lo + per);
      if (lo < hi) th.emplace_back(fn, lo, hi);
    }
    for (auto& t : th) t.join();
  };
  parallel(parse, b);
  size_t total_nodes = 0, total_edges = 0;
  for (int64_t i = 0; i < b; ++i) {
    if (!recs[i].ok || !recs[i].has_root) {
      set_err(err, err_cap, "record %lld is not a well-formed training sample", (long long)i);
      return GIGL_E_INVALID_ARG;
    }
    // a RootedNodeNeighborhood without a neighborhood collates as the root alone
    // (rooted_node_neighborhood_data_loader.py:96-110)
    if (!recs[i].has_graph) recs[i].nodes.push_back(recs[i].root);
    total_nodes += recs[i].nodes.size();
    total_edges += recs[i].edges.size();
  }
  gigl_collated* c = new (std::nothrow) gigl_collated();
  if (!c) return GIGL_E_OOM;
  // ---- first-seen numbering + edge dedup (sequential: the order IS the semantics)
  IdMap ids(total_nodes + 1);
  std::vector<const uint8_t*> src_rows;
  src_rows.reserve(total_nodes);
  int32_t dim = -1;
  struct EdgeSet {
    std::vector<uint64_t> k;
    uint64_t mask;
    explicit EdgeSet(size_t n) {
      size_t cap = 16;
      while (cap < 2 * n) cap <<= 1;
      k.assign(cap, ~0ull);
      mask = cap - 1;
    }
    bool insert(uint64_t key) {
      uint64_t s = (key * 0x9E3779B97F4A7C15ull) >> 17 & mask;
      while (k[s] != ~0ull) {
        if (k[s] == key) return false;
        s = (s + 1) & mask;
      }
      k[s] = key;
      return true;
    }
  } eset(total_edges + 1);
  std::vector<uint64_t> ekeys;
  std::vector<const uint8_t*> erows;  // feature run of the edge that registered the key (first one wins)
  ekeys.reserve(total_edges);
  erows.reserve(total_edges);
  int32_t edim = -1;  // set by the first edge added (GraphBuilder.should_register_edge_features)
  for (int64_t i = 0; i < b; ++i) {
    for (const NodeRef& n : recs[i].nodes) {
      int32_t* s = ids.slot(n.id);
      if (*s >= 0) {
        const int32_t have = dim < 0 ? 0 : dim;
        if (n.n != have || !allclose(src_rows[*s], n.f ? n.f : buf, n.n)) {
          set_err(err, err_cap, "node %lld re-added with different features (record %lld)", (long long)n.id, (long long)i);
          delete c;
          return GIGL_E_INVALID_ARG;
        }
        continue;
      }
      if (dim < 0) dim = n.n;
      if (n.n != dim) {
        set_err(err, err_cap, "node %lld has %lld feature values, the batch's first node has another count",
                (long long)n.id, (long long)n.n);
        delete c;
        return GIGL_E_INVALID_ARG;
      }
      *s = (int32_t)c->node_ids.size();
      c->node_ids.push_back(n.id);
      src_rows.push_back(n.f ? n.f : buf);
    }
    for (const EdgeRef& e : recs[i].edges) {
      const int32_t ls = ids.find(e.src), ld = ids.find(e.dst);
      if (ls < 0 || ld < 0) {
        set_err(err, err_cap, "Tried to fetch a node which we have no information on (edge %lld->%lld)",
                (long long)e.src, (long long)e.dst);
        delete c;
        return GIGL_E_INVALID_ARG;
      }
      if (edim < 0) edim = e.n;
      if (e.n != edim) {  // abstract_graph_builder.py:121-132 (features on some edges only), torch.stack otherwise
        set_err(err, err_cap, "edge feature registration is inconsistent: edge %lld->%lld differs from the first edge",
                (long long)e.src, (long long)e.dst);
        delete c;
        return GIGL_E_INVALID_ARG;
      }
      const uint64_t key = ((uint64_t)(uint32_t)ls << 32) | (uint32_t)ld;
      if (eset.insert(key)) {
        ekeys.push_back(key);
        erows.push_back(e.f ? e.f : buf);
      }
    }
  }
  c->feat_dim = dim < 0 ? 0 : dim;
  // ---- feature rows (parallel copy), edges sorted by (src, dst) = coalesce()
  const int64_t n = (int64_t)c->node_ids.size();
  c->x.resize((size_t)n * c->feat_dim);
  if (c->feat_dim)
    parallel([&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) memcpy(&c->x[(size_t)i * c->feat_dim], src_rows[i], 4 * (size_t)c->feat_dim);
    }, n);
  c->edge_dim = edim < 0 ? 0 : edim;
  std::vector<uint32_t> eorder(ekeys.size());
  for (size_t i = 0; i < eorder.size(); ++i) eorder[i] = (uint32_t)i;
  std::sort(eorder.begin(), eorder.end(), [&](uint32_t a, uint32_t b) { return ekeys[a] < ekeys[b]; });
  c->edge_src.resize(ekeys.size());
  c->edge_dst.resize(ekeys.size());
  for (size_t i = 0; i < ekeys.size(); ++i) {
    c->edge_src[i] = (int64_t)(ekeys[eorder[i]] >> 32);
    c->edge_dst[i] = (int64_t)(ekeys[eorder[i]] & 0xFFFFFFFFull);
  }
  c->edge_attr.resize(ekeys.size() * (size_t)c->edge_dim);
  if (c->edge_dim)
    parallel([&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i)
        memcpy(&c->edge_attr[(size_t)i * c->edge_dim], erows[eorder[i]], 4 * (size_t)c->edge_dim);
    }, (int64_t)ekeys.size());
  // ---- per-sample outputs
  c->root_local.resize((size_t)b);
  c->labels.assign((size_t)b, 0);
  c->has_label.assign((size_t)b, 0);
  c->pos_off.assign((size_t)b + 1, 0);
  c->neg_off.assign((size_t)b + 1, 0);
  for (int64_t i = 0; i < b; ++i) {
    const int32_t r = ids.find(recs[i].root.id);
    if (r < 0) {
      set_err(err, err_cap, "root node %lld of record %lld is not in the batch graph", (long long)recs[i].root.id,
              (long long)i);
      delete c;
      return GIGL_E_INVALID_ARG;
    }
    c->root_local[i] = r;
    c->labels[i] = recs[i].label;
    c->has_label[i] = recs[i].has_label ? 1 : 0;
    for (int which = 0; which < 2; ++which) {
      const std::vector<uint32_t>& v = which ? recs[i].neg : recs[i].pos;
      std::vector<int64_t>& dst = which ? c->neg_dst : c->pos_dst;
      for (uint32_t g : v) {
        const int32_t l = ids.find(g);
        if (l < 0) {
          set_err(err, err_cap, "supervision edge target %lld of record %lld is not in the batch graph", (long long)g,
                  (long long)i);
          delete c;
          return GIGL_E_INVALID_ARG;
        }
        dst.push_back(l);
      }
      (which ? c->neg_off : c->pos_off)[i + 1] = (int64_t)dst.size();
    }
  }
  *out = c;
  return GIGL_OK;
}

// Typed (heterogeneous) collate: the same semantics per type — GraphBuilder keeps ONE first-seen counter per node type
// (abstract_graph_builder.py:16-24) and one ordered, de-duplicated edge list per edge type (:100-150); an edge's
// endpoints are nodes of the edge type's source / destination node types (gbml_protos_translator.py:101-121 through
// GraphMetadataPbWrapper.condensed_edge_type_to_edge_type_map), which the caller passes as et_src_nt / et_dst_nt.
int32_t gigl_collate_typed_records(const uint8_t* buf, const int64_t* payload_off, const int64_t* payload_len, int64_t b,
                                   int32_t kind, int32_t n_node_types, int32_t n_edge_types, const int32_t* et_src_nt,
                                   const int32_t* et_dst_nt, int32_t n_threads, gigl_collated_typed** out, char* err,
                                   int32_t err_cap) {
  if (!buf || !payload_off || !payload_len || b < 0 || !out || n_node_types < 1 || n_edge_types < 0) return GIGL_E_INVALID_ARG;
  if (n_edge_types > 0 && (!et_src_nt || !et_dst_nt)) return GIGL_E_INVALID_ARG;
  if (kind != GIGL_REC_ROOTED_NODE_NEIGHBORHOOD && kind != GIGL_REC_NODE_ANCHOR_LINK_PRED) return GIGL_E_INVALID_ARG;
  for (int32_t t = 0; t < n_edge_types; ++t)
    if (et_src_nt[t] < 0 || et_src_nt[t] >= n_node_types || et_dst_nt[t] < 0 || et_dst_nt[t] >= n_node_types)
      return GIGL_E_INVALID_ARG;
  *out = nullptr;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 64) n_threads = 64;
  std::vector<Rec> recs((size_t)b);
  auto parallel = [&](auto fn, int64_t n) {
    if (n_threads == 1 || n < 64) {
      fn(0, n);
      return;
    }
    std::vector<std::thread> th;
    const int64_t per = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
      const int64_t lo = t * per, hi = std::min(n, lo + per);
      if (lo < hi) th.emplace_back(fn, lo, hi);
    }
    for (auto& t : th) t.join();
  };
  parallel([&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i)
      recs[i].ok = parse_record(Span{buf + payload_off[i], buf + payload_off[i] + payload_len[i]}, kind, recs[i]);
  }, b);
  std::vector<size_t> nodes_of((size_t)n_node_types, 0), edges_of((size_t)std::max(n_edge_types, 1), 0);
  for (int64_t i = 0; i < b; ++i) {
    Rec& r = recs[i];
    if (!r.ok || !r.has_root) {
      set_err(err, err_cap, "record %lld is not a well-formed training sample", (long long)i);
      return GIGL_E_INVALID_ARG;
    }
    if (!r.has_graph) r.nodes.push_back(r.root);
    for (const NodeRef& n : r.nodes) {
      if (n.type < 0 || n.type >= n_node_types) {
        set_err(err, err_cap, "node %lld has condensed node type %lld outside the graph metadata", (long long)n.id,
                (long long)n.type);
        return GIGL_E_INVALID_ARG;
      }
      ++nodes_of[n.type];
    }
    for (const EdgeRef& e : r.edges) {
      if (e.type < 0 || e.type >= n_edge_types) {
        set_err(err, err_cap, "edge %lld->... has a condensed edge type outside the graph metadata (record %lld)",
                (long long)e.src, (long long)i);
        return GIGL_E_INVALID_ARG;
      }
      ++edges_of[e.type];
    }
  }
  gigl_collated_typed* c = new (std::nothrow) gigl_collated_typed();
  if (!c) return GIGL_E_OOM;
  c->nt.resize((size_t)n_node_types);
  c->et.resize((size_t)n_edge_types);
  auto fail = [&](int32_t code) {
    delete c;
    return code;
  };
  struct KeySet {
    std::vector<uint64_t> k;
    uint64_t mask;
    explicit KeySet(size_t n) {
      size_t cap = 16;
      while (cap < 2 * n) cap <<= 1;
      k.assign(cap, ~0ull);
      mask = cap - 1;
    }
    bool insert(uint64_t key) {
      uint64_t s = (key * 0x9E3779B97F4A7C15ull) >> 17 & mask;
      while (k[s] != ~0ull) {
        if (k[s] == key) return false;
        s = (s + 1) & mask;
      }
      k[s] = key;
      return true;
    }
  };
  std::vector<IdMap> ids;
  std::vector<std::vector<const uint8_t*>> src_rows((size_t)n_node_types);
  std::vector<int32_t> dim((size_t)n_node_types, -1);
  for (int32_t t = 0; t < n_node_types; ++t) ids.emplace_back(nodes_of[t] + 1);
  std::vector<KeySet> esets;
  std::vector<std::vector<uint64_t>> ekeys((size_t)n_edge_types);
  std::vector<std::vector<const uint8_t*>> erows((size_t)n_edge_types);
  std::vector<int32_t> edim((size_t)n_edge_types, -1);
  for (int32_t t = 0; t < n_edge_types; ++t) esets.emplace_back(edges_of[t] + 1);
  int has_efeat = -1;
  // ---- first-seen numbering per node type + edge dedup per edge type (sequential: the order IS the semantics)
  for (int64_t i = 0; i < b; ++i) {
    for (const NodeRef& n : recs[i].nodes) {
      const int32_t t = n.type;
      int32_t* s = ids[t].slot(n.id);
      if (*s >= 0) {
        const int32_t have = dim[t] < 0 ? 0 : dim[t];
        if (n.n != have || !allclose(src_rows[t][*s], n.f ? n.f : buf, n.n)) {
          set_err(err, err_cap, "node %lld re-added with different features (record %lld)", (long long)n.id, (long long)i);
          return fail(GIGL_E_INVALID_ARG);
        }
        continue;
      }
      if (dim[t] < 0) dim[t] = n.n;
      if (n.n != dim[t]) {
        set_err(err, err_cap, "node %lld has %lld feature values, the first node of its type has another count",
                (long long)n.id, (long long)n.n);
        return fail(GIGL_E_INVALID_ARG);
      }
      *s = (int32_t)c->nt[t].ids.size();
      c->nt[t].ids.push_back(n.id);
      src_rows[t].push_back(n.f ? n.f : buf);
    }
    for (const EdgeRef& e : recs[i].edges) {
      const int32_t t = e.type;
      const int32_t ls = ids[et_src_nt[t]].find(e.src), ld = ids[et_dst_nt[t]].find(e.dst);
      if (ls < 0 || ld < 0) {
        set_err(err, err_cap, "Tried to fetch a node which we have no information on (edge %lld->%lld)",
                (long long)e.src, (long long)e.dst);
        return fail(GIGL_E_INVALID_ARG);
      }
      // (should_register_edge_features is ONE flag of the builder: features on every edge of the batch or on none)
      if (has_efeat < 0) has_efeat = e.n > 0 ? 1 : 0;
      if ((e.n > 0) != (has_efeat == 1)) {
        set_err(err, err_cap, "edge feature registration is inconsistent: edge %lld->%lld differs from the first edge",
                (long long)e.src, (long long)e.dst);
        return fail(GIGL_E_INVALID_ARG);
      }
      if (edim[t] < 0) edim[t] = e.n;
      if (e.n != edim[t]) {
        set_err(err, err_cap, "edge %lld->%lld carries another number of feature values than the first edge of its type",
                (long long)e.src, (long long)e.dst);
        return fail(GIGL_E_INVALID_ARG);
      }
      const uint64_t key = ((uint64_t)(uint32_t)ls << 32) | (uint32_t)ld;
      if (esets[t].insert(key)) {
        ekeys[t].push_back(key);
        erows[t].push_back(e.f ? e.f : buf);
      }
    }
  }
  // ---- feature rows, edges of every type sorted by (src, dst) = coalesce()
  for (int32_t t = 0; t < n_node_types; ++t) {
    gigl_collated_typed::NodeType& T = c->nt[t];
    T.feat_dim = dim[t] < 0 ? 0 : dim[t];
    const int64_t n = (int64_t)T.ids.size();
    T.x.resize((size_t)n * T.feat_dim);
    if (T.feat_dim)
      parallel([&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) memcpy(&T.x[(size_t)i * T.feat_dim], src_rows[t][i], 4 * (size_t)T.feat_dim);
      }, n);
  }
  for (int32_t t = 0; t < n_edge_types; ++t) {
    gigl_collated_typed::EdgeType& E = c->et[t];
    E.edge_dim = edim[t] < 0 ? 0 : edim[t];
    const std::vector<uint64_t>& k = ekeys[t];
    std::vector<uint32_t> order(k.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return k[x] < k[y]; });
    E.src.resize(k.size());
    E.dst.resize(k.size());
    E.attr.resize(k.size() * (size_t)E.edge_dim);
    for (size_t i = 0; i < k.size(); ++i) {
      E.src[i] = (int64_t)(k[order[i]] >> 32);
      E.dst[i] = (int64_t)(k[order[i]] & 0xFFFFFFFFull);
      if (E.edge_dim) memcpy(&E.attr[i * (size_t)E.edge_dim], erows[t][order[i]], 4 * (size_t)E.edge_dim);
    }
  }
  // ---- per-sample outputs
  c->root_type.resize((size_t)b);
  c->root_local.resize((size_t)b);
  c->labels.assign((size_t)b, 0);
  c->has_label.assign((size_t)b, 0);
  c->pos_off.assign((size_t)b + 1, 0);
  c->neg_off.assign((size_t)b + 1, 0);
  for (int64_t i = 0; i < b; ++i) {
    const Rec& r = recs[i];
    const int32_t rt = r.root.type;
    const int32_t l = rt >= 0 && rt < n_node_types ? ids[rt].find(r.root.id) : -1;
    if (l < 0) {
      set_err(err, err_cap, "root node %lld of record %lld is not in the batch graph", (long long)r.root.id, (long long)i);
      return fail(GIGL_E_INVALID_ARG);
    }
    c->root_type[i] = rt;
    c->root_local[i] = l;
    c->labels[i] = r.label;
    c->has_label[i] = r.has_label ? 1 : 0;
    for (int which = 0; which < 2; ++which) {
      const std::vector<uint32_t>& v = which ? r.neg : r.pos;
      const std::vector<int32_t>& vt = which ? r.neg_type : r.pos_type;
      std::vector<int64_t>& dst = which ? c->neg_dst : c->pos_dst;
      std::vector<int32_t>& dty = which ? c->neg_type : c->pos_type;
      for (size_t q = 0; q < v.size(); ++q) {
        const int32_t t = vt[q];
        const int32_t ll = t >= 0 && t < n_edge_types ? ids[et_dst_nt[t]].find(v[q]) : -1;
        if (ll < 0) {
          set_err(err, err_cap, "supervision edge target %lld of record %lld is not in the batch graph", (long long)v[q],
                  (long long)i);
          return fail(GIGL_E_INVALID_ARG);
        }
        dst.push_back(ll);
        dty.push_back(t);
      }
      (which ? c->neg_off : c->pos_off)[i + 1] = (int64_t)dst.size();
    }
  }
  *out = c;
  return GIGL_OK;
}

int32_t gigl_collated_typed_info(const gigl_collated_typed* c, int64_t* nodes_per_type, int32_t* feat_dim_per_type,
                                 int64_t* edges_per_type, int32_t* edge_dim_per_type, int64_t* n_pos, int64_t* n_hard_neg) {
  if (!c) return GIGL_E_INVALID_ARG;
  for (size_t t = 0; t < c->nt.size(); ++t) {
    if (nodes_per_type) nodes_per_type[t] = (int64_t)c->nt[t].ids.size();
    if (feat_dim_per_type) feat_dim_per_type[t] = c->nt[t].feat_dim;
  }
  for (size_t t = 0; t < c->et.size(); ++t) {
    if (edges_per_type) edges_per_type[t] = (int64_t)c->et[t].src.size();
    if (edge_dim_per_type) edge_dim_per_type[t] = c->et[t].edge_dim;
  }
  if (n_pos) *n_pos = (int64_t)c->pos_dst.size();
  if (n_hard_neg) *n_hard_neg = (int64_t)c->neg_dst.size();
  return GIGL_OK;
}

int32_t gigl_collated_typed_nodes(const gigl_collated_typed* c, int32_t node_type, uint32_t* node_ids, float* x) {
  if (!c || node_type < 0 || (size_t)node_type >= c->nt.size()) return GIGL_E_INVALID_ARG;
  const gigl_collated_typed::NodeType& T = c->nt[node_type];
  if (node_ids && !T.ids.empty()) memcpy(node_ids, T.ids.data(), T.ids.size() * 4);
  if (x && !T.x.empty()) memcpy(x, T.x.data(), T.x.size() * 4);
  return GIGL_OK;
}

int32_t gigl_collated_typed_edges(const gigl_collated_typed* c, int32_t edge_type, int64_t* edge_index, float* edge_attr) {
  if (!c || edge_type < 0 || (size_t)edge_type >= c->et.size()) return GIGL_E_INVALID_ARG;
  const gigl_collated_typed::EdgeType& E = c->et[edge_type];
  if (edge_index && !E.src.empty()) {
    memcpy(edge_index, E.src.data(), E.src.size() * 8);
    memcpy(edge_index + E.src.size(), E.dst.data(), E.dst.size() * 8);
  }
  if (edge_attr && !E.attr.empty()) memcpy(edge_attr, E.attr.data(), E.attr.size() * 4);
  return GIGL_OK;
}

int32_t gigl_collated_typed_samples(const gigl_collated_typed* c, int32_t* root_type, int64_t* root_local, int64_t* labels,
                                    uint8_t* has_label, int64_t* pos_off, int64_t* pos_dst, int32_t* pos_type,
                                    int64_t* neg_off, int64_t* neg_dst, int32_t* neg_type) {
  if (!c) return GIGL_E_INVALID_ARG;
  auto cp = [](void* dst, const void* src, size_t bytes) {
    if (dst && bytes) memcpy(dst, src, bytes);
  };
  cp(root_type, c->root_type.data(), c->root_type.size() * 4);
  cp(root_local, c->root_local.data(), c->root_local.size() * 8);
  cp(labels, c->labels.data(), c->labels.size() * 8);
  cp(has_label, c->has_label.data(), c->has_label.size());
  cp(pos_off, c->pos_off.data(), c->pos_off.size() * 8);
  cp(pos_dst, c->pos_dst.data(), c->pos_dst.size() * 8);
  cp(pos_type, c->pos_type.data(), c->pos_type.size() * 4);
  cp(neg_off, c->neg_off.data(), c->neg_off.size() * 8);
  cp(neg_dst, c->neg_dst.data(), c->neg_dst.size() * 8);
  cp(neg_type, c->neg_type.data(), c->neg_type.size() * 4);
  return GIGL_OK;
}

int32_t gigl_collated_typed_destroy(gigl_collated_typed* c) {
  delete c;
  return GIGL_OK;
}

int32_t gigl_collated_info(const gigl_collated* c, int64_t* n_nodes, int64_t* n_edges, int32_t* feat_dim,
                           int64_t* n_pos, int64_t* n_hard_neg) {
  if (!c) return GIGL_E_INVALID_ARG;
  if (n_nodes) *n_nodes = (int64_t)c->node_ids.size();
  if (n_edges) *n_edges = (int64_t)c->edge_src.size();
  if (feat_dim) *feat_dim = c->feat_dim;
  if (n_pos) *n_pos = (int64_t)c->pos_dst.size();
  if (n_hard_neg) *n_hard_neg = (int64_t)c->neg_dst.size();
  return GIGL_OK;
}

int32_t gigl_collated_copy(const gigl_collated* c, uint32_t* node_ids, float* x, int64_t* edge_index,
                           int64_t* root_local, int64_t* labels, uint8_t* has_label, int64_t* pos_off,
                           int64_t* pos_dst, int64_t* neg_off, int64_t* neg_dst) {
  if (!c) return GIGL_E_INVALID_ARG;
  auto cp = [](void* dst, const void* src, size_t bytes) {
    if (dst && bytes) memcpy(dst, src, bytes);
  };
  cp(node_ids, c->node_ids.data(), c->node_ids.size() * 4);
  cp(x, c->x.data(), c->x.size() * 4);
  if (edge_index) {
    cp(edge_index, c->edge_src.data(), c->edge_src.size() * 8);
    cp(edge_index + c->edge_src.size(), c->edge_dst.data(), c->edge_dst.size() * 8);
  }
  cp(root_local, c->root_local.data(), c->root_local.size() * 8);
  cp(labels, c->labels.data(), c->labels.size() * 8);
  cp(has_label, c->has_label.data(), c->has_label.size());
  cp(pos_off, c->pos_off.data(), c->pos_off.size() * 8);
  cp(pos_dst, c->pos_dst.data(), c->pos_dst.size() * 8);
  cp(neg_off, c->neg_off.data(), c->neg_off.size() * 8);
  cp(neg_dst, c->neg_dst.data(), c->neg_dst.size() * 8);
  return GIGL_OK;
}

int32_t gigl_collated_edge_attr(const gigl_collated* c, int32_t* edge_dim, float* edge_attr) {
  if (!c) return GIGL_E_INVALID_ARG;
  if (edge_dim) *edge_dim = c->edge_dim;
  if (edge_attr && !c->edge_attr.empty()) memcpy(edge_attr, c->edge_attr.data(), c->edge_attr.size() * 4);
  return GIGL_OK;
}

int32_t gigl_collated_destroy(gigl_collated* c) {
  delete c;
  return GIGL_OK;
}

}  // extern "C"
