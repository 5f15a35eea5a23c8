#!/usr/bin/env python3
"""Generate tests/golden/graph_builder_edge_feature_traces.json from the reference checkout (BUILD container only).

DATA only: seeded samples (nodes, edges with feature rows) and what the REFERENCE GraphBuilder
(python/gigl/src/common/graph_builder/abstract_graph_builder.py, imported) registered for them — the global->local
map, the ordered local edges and the feature row kept for each (first registration wins, :144-145) — plus the error
the builder raises when only some edges carry features (:121-132)."""
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GIGL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, os.path.join(REF, "python"))

from gigl.src.common.graph_builder.abstract_graph_builder import GraphBuilder  # noqa: E402
from gigl.src.common.types.graph_data import Edge, EdgeType, Node, NodeId, NodeType, Relation  # noqa: E402


class TraceBuilder(GraphBuilder):
    def __init__(self):
        self.reset()

    def build(self):
        return None


nt = NodeType("n")
et = EdgeType(nt, Relation("r"), nt)
traces = []
for case in range(10):
    r = np.random.default_rng(7000 + case)
    de = int(r.integers(1, 5))
    n_samples = int(r.integers(1, 6))
    universe = int(r.integers(4, 30))
    samples = []
    bld = TraceBuilder()
    for _ in range(n_samples):
        k = int(r.integers(1, min(universe, 10) + 1))
        nodes = [int(x) for x in r.choice(universe, size=k, replace=False)]
        ne = int(r.integers(0, 3 * k + 1))
        edges, seen = [], set()
        for _ in range(ne):
            s, d = int(nodes[int(r.integers(0, k))]), int(nodes[int(r.integers(0, k))])
            if (s, d) in seen:  # a per-sample PygGraphData holds each edge once
                continue
            seen.add((s, d))
            # the same global edge gets DIFFERENT features in different samples: shows which registration is kept
            edges.append([s, d, [float(np.float32(v)) for v in r.standard_normal(de)]])
        samples.append({"nodes": nodes, "edges": edges})
        for v in nodes:
            g = Node(type=nt, id=NodeId(v))
            if g not in bld.global_node_to_subgraph_node_map:
                bld.add_node(node=g)
        for s, d, f in edges:
            bld.add_edge(edge=Edge.from_nodes(Node(type=nt, id=NodeId(s)), Node(type=nt, id=NodeId(d)), Relation("r")),
                         feature_values=torch.tensor(f, dtype=torch.float32), skip_if_exists=True)
    mapping = {int(g.id): int(l.id) for g, l in bld.global_node_to_subgraph_node_map.items()}
    ordered = [[int(e.src_node_id), int(e.dst_node_id)] for e in bld.ordered_edges[et]]
    kept = [[float(v) for v in bld.subgraph_edge_feature_dict[e].tolist()] for e in bld.ordered_edges[et]]
    traces.append({"edge_dim": de, "samples": samples, "global_to_local": mapping, "ordered_edges_local": ordered,
                   "ordered_edge_features": kept})

# mixed registration: features on the first edge, none on the second (and the other way round)
errors = []
for first_has in (True, False):
    bld = TraceBuilder()
    a, b = Node(type=nt, id=NodeId(1)), Node(type=nt, id=NodeId(2))
    bld.add_node(node=a)
    bld.add_node(node=b)
    f = torch.tensor([1.0, 2.0])
    bld.add_edge(edge=Edge.from_nodes(a, b, Relation("r")), feature_values=f if first_has else None)
    try:
        bld.add_edge(edge=Edge.from_nodes(b, a, Relation("r")), feature_values=None if first_has else f)
        errors.append({"first_has_features": first_has, "raises": None})
    except Exception as e:  # noqa: BLE001
        errors.append({"first_has_features": first_has, "raises": type(e).__name__})
json.dump({"traces": traces, "mixed_registration": errors},
          open(os.path.join(OUT, "graph_builder_edge_feature_traces.json"), "w"))
print(len(traces), errors)
