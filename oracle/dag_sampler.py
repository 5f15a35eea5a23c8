"""TEST INFRASTRUCTURE ONLY: per-root CPU restatement of the reference's SamplingOp-DAG traversal
(GraphDBSampler.getKHopSubgraphForRootNode, scala_spark35/subgraph_sampler/src/main/scala/libs/sampler/
GraphDBSampler.scala:40-148; LocalDbClient.executeQuery, scala_spark35/common/src/main/scala/graphdb/local/
LocalDbClient.scala:156-237), one root at a time with Python sets and an explicit queue, as the Scala code does.

Parity status: the traversal (queue discipline, frontier = union of the parents' node sets, skipped ops, edge
orientation, final union + root) follows the reference line by line; WHICH neighbours a query returns is not defined
by the reference (LocalDbClient: first n of a HashSet; Nebula: server-side) -> UNPINNED.  The rule restated here is the
Spark sampler's hash permutation (oracle.hash_permutation) with K = root + frontier node, counter = 1 + op position."""
from collections import deque

import numpy as np

from .oracle import hash_permutation

MASK = 0xFFFFFFFF


def neighbour_lists(edges):
    """edges: {edge_type: (src, dst)} -> {(edge_type, 'INCOMING'|'OUTGOING'): {node: sorted distinct neighbours}}"""
    out = {}
    for et, (src, dst) in edges.items():
        inc, outg = {}, {}
        for s, d in zip(np.asarray(src).tolist(), np.asarray(dst).tolist()):
            inc.setdefault(d, set()).add(s)
            outg.setdefault(s, set()).add(d)
        out[(et, "INCOMING")] = {k: np.array(sorted(v), dtype=np.uint32) for k, v in inc.items()}
        out[(et, "OUTGOING")] = {k: np.array(sorted(v), dtype=np.uint32) for k, v in outg.items()}
    return out


def sample_for_root(root, ops, nbrs, node_types, condensed_edge_types, root_type, sampling_seed=42):
    """ops: list of objects with op_name, edge_type, num_nodes_to_sample, input_op_names, sampling_direction.
    -> (set of (src, dst, condensed_edge_type), set of (node_id, condensed_node_type))"""
    by_name = {op.op_name: op for op in ops}
    children = {op.op_name: [] for op in ops}
    for op in ops:
        for p in op.input_op_names:
            if p in by_name:
                children[p].append(op.op_name)
    parents = {op.op_name: [p for p in op.input_op_names if p in by_name] for op in ops}
    queue = deque(op.op_name for op in ops if not op.input_op_names)
    results = {}
    while queue:
        name = queue.popleft()
        op = by_name[name]
        if name in results:
            continue
        if not parents[name]:
            frontier = [root]
        else:
            if not all(p in results for p in parents[name]):
                continue  # re-queued by its last parent
            frontier = sorted({v for p in parents[name] for v, _ in results[p][1]})
        if not frontier:
            continue  # "There were no parent nodes to process. Our traversal ends here."
        outgoing = op.sampling_direction == "OUTGOING"
        lists = nbrs[(op.edge_type, "OUTGOING" if outgoing else "INCOMING")]
        cet = condensed_edge_types[op.edge_type]
        got_type = node_types[op.edge_type.dst_node_type if outgoing else op.edge_type.src_node_type]
        counter = 1 + [o.op_name for o in ops].index(name)
        e_set, n_set = set(), set()
        for v in frontier:
            row = lists.get(v)
            if row is None:
                continue
            pick = hash_permutation(row, (root + v) & MASK, sampling_seed=sampling_seed, counter=counter)
            for u in pick[: op.num_nodes_to_sample].tolist():
                e_set.add((v, u, cet) if outgoing else (u, v, cet))
                n_set.add((u, got_type))
        results[name] = (e_set, n_set)
        queue.extend(children[name])
    edges, nodes = set(), {(root, node_types[root_type])}
    for e_set, n_set in results.values():
        edges |= e_set
        nodes |= n_set
    return edges, nodes
