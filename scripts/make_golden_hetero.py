#!/usr/bin/env python3
"""Generate tests/golden/graph_builder_hetero_traces.json from the reference checkout (BUILD container only).

Seeded heterogeneous samples (3 node types, 4 edge types, some with edge features) pushed through the REFERENCE
GraphBuilder (python/gigl/src/common/graph_builder/abstract_graph_builder.py, imported) in the order its collate uses:
a sample's nodes first (first-seen numbering per node type), then its edges with skip_if_exists.  The file holds DATA
only: the samples (inputs) and, per node type, the global -> local map; per edge type, the ordered local edges
(expected outputs)."""
import json
import os
import sys

import numpy as np

REF = os.environ.get("GIGL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, os.path.join(REF, "python"))


def main():
    import torch
    from gigl.src.common.graph_builder.abstract_graph_builder import GraphBuilder
    from gigl.src.common.types.graph_data import Edge, EdgeType, Node, NodeId, NodeType, Relation

    class TraceBuilder(GraphBuilder):
        def __init__(self):
            self.reset()

        def build(self):
            return None

    nts = [NodeType("author"), NodeType("paper"), NodeType("venue")]
    # condensed edge type -> (src node type, dst node type)
    ets_idx = [(0, 1), (1, 0), (1, 2), (1, 1)]
    rels = ["writes", "written_by", "published_in", "cites"]
    ets = [EdgeType(nts[s], Relation(r), nts[d]) for (s, d), r in zip(ets_idx, rels)]
    traces = []
    for case in range(10):
        r = np.random.default_rng(7000 + case)
        with_efeat = case % 2 == 1
        n_samples = int(r.integers(1, 7))
        universe = [int(r.integers(3, 30)) for _ in nts]
        feat_dim = [int(r.integers(1, 5)) for _ in nts]
        edge_dim = [int(r.integers(1, 4)) for _ in ets]
        node_feat = [{} for _ in nts]
        samples = []
        bld = TraceBuilder()
        for _ in range(n_samples):
            nodes = []
            for t in range(len(nts)):
                k = int(r.integers(0 if t else 1, min(universe[t], 9) + 1))
                for v in r.choice(universe[t], size=k, replace=False):
                    v = int(v)
                    if v not in node_feat[t]:
                        node_feat[t][v] = [float(np.float32(x)) for x in r.standard_normal(feat_dim[t])]
                    nodes.append([t, v])
            order = r.permutation(len(nodes))
            nodes = [nodes[i] for i in order]  # types interleaved, as a sampler's output may be
            by_type = [[v for t, v in nodes if t == q] for q in range(len(nts))]
            edges = []
            for c, (s_t, d_t) in enumerate(ets_idx):
                if not by_type[s_t] or not by_type[d_t]:
                    continue
                for _ in range(int(r.integers(0, 2 * len(by_type[s_t]) + 1))):
                    s = by_type[s_t][int(r.integers(0, len(by_type[s_t])))]
                    d = by_type[d_t][int(r.integers(0, len(by_type[d_t])))]
                    ef = [float(np.float32(x)) for x in r.standard_normal(edge_dim[c])] if with_efeat else []
                    edges.append([c, s, d, ef])
            order = r.permutation(len(edges))
            edges = [edges[i] for i in order]
            samples.append({"nodes": [[t, v, node_feat[t][v]] for t, v in nodes], "edges": edges})
            for t, v in nodes:
                g = Node(type=nts[t], id=NodeId(v))
                if g not in bld.global_node_to_subgraph_node_map:
                    bld.add_node(node=g, feature_values=torch.tensor(node_feat[t][v]))
            seen = set()
            for c, s, d, ef in edges:
                if (c, s, d) in seen:  # a per-sample graph holds each edge once (first registration)
                    continue
                seen.add((c, s, d))
                e = Edge.from_nodes(Node(type=nts[ets_idx[c][0]], id=NodeId(s)), Node(type=nts[ets_idx[c][1]], id=NodeId(d)),
                                    Relation(rels[c]))
                bld.add_edge(edge=e, feature_values=torch.tensor(ef) if with_efeat else None, skip_if_exists=True)
        g2l = [{} for _ in nts]
        for g, l in bld.global_node_to_subgraph_node_map.items():
            g2l[nts.index(g.type)][int(g.id)] = int(l.id)
        ordered = []
        efeat = []
        for c, et in enumerate(ets):
            es = bld.ordered_edges.get(et, [])
            ordered.append([[int(e.src_node_id), int(e.dst_node_id)] for e in es])
            efeat.append([[float(x) for x in bld.subgraph_edge_feature_dict[e]] if with_efeat else [] for e in es])
        traces.append({"n_node_types": len(nts), "edge_type_endpoints": ets_idx, "samples": samples,
                       "global_to_local": [{str(k): v for k, v in m.items()} for m in g2l],
                       "ordered_edges_local": ordered, "ordered_edge_features": efeat})
    json.dump(traces, open(os.path.join(OUT, "graph_builder_hetero_traces.json"), "w"))
    print("written", os.path.join(OUT, "graph_builder_hetero_traces.json"), len(traces), "cases")


if __name__ == "__main__":
    main()
