#!/usr/bin/env python3
"""Topology of a captured hipGraph, read through the HIP runtime (hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType on the handle that
torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph() returns; ROCm's debug_dump writes nothing).  A single-stream capture must come out as
ONE chain: edges = nodes - 1, one root, no node with two successors.  HAVATAR_GRAPH_SHAPE=1 makes GraphedTrainStep print this for its graph."""
import collections
import ctypes as C

_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}


def describe(raw_graph):
    hip = C.CDLL("libamdhip64.so")
    g = C.c_void_p(int(raw_graph))
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(g, None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(g, nodes, C.byref(n)) == 0
    e = C.c_size_t(0)
    assert hip.hipGraphGetEdges(g, None, None, C.byref(e)) == 0
    src, dst = (C.c_void_p * max(e.value, 1))(), (C.c_void_p * max(e.value, 1))()
    if e.value:
        assert hip.hipGraphGetEdges(g, src, dst, C.byref(e)) == 0
    succ, pred = collections.Counter(), collections.Counter()
    for i in range(e.value):
        succ[src[i]] += 1
        pred[dst[i]] += 1
    kinds = collections.Counter()
    for nd in nodes:
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(t))
        kinds[_TYPES.get(t.value, str(t.value))] += 1
    roots = sum(1 for nd in nodes if pred[nd] == 0)
    leaves = sum(1 for nd in nodes if succ[nd] == 0)
    forks = sum(1 for nd in nodes if succ[nd] > 1)
    joins = sum(1 for nd in nodes if pred[nd] > 1)
    return {"nodes": n.value, "edges": e.value, "roots": roots, "leaves": leaves, "forks": forks, "joins": joins, "kinds": dict(kinds),
            "chain": e.value == n.value - 1 and roots == 1 and forks == 0}
