"""hipGraph hygiene for the captured steps (bench.py, the training-step benches, the tests).

Root cause of round 2's "stale PyTorch reductions from the second replay on" (DESIGN.md section 5): on this stack
(ROCm 7.2, PyTorch 2.10) a captured `hipMemsetAsync` -- a MEMSET node -- executes correctly on the first replay of the
instantiated graph and writes a garbage 32-bit pattern from the second replay on (scripts/debug_graph_memset.py:
`memset(buf, 0); buf += 1` gives 1, then -1291841535 ever after).  PyTorch's multi-block reductions zero their
semaphore buffer with cudaMemsetAsync before every launch (ATen/native/cuda/Reduce.cuh) and never reset it in the
kernel, so under replay their "last block" is never recognised again and the output keeps its old value; the same
defect behind a larger memset made a replay abort outright.  This build's kernels never use memsets (zero fills
are kernels: `apn_zero_fill`, the producers' own clears); what PyTorch captures around them is checked here.

`node_census(graph)` counts the nodes of a captured graph by type through the HIP graph API;
`assert_replayable(graph)` raises `MemsetNodeInGraph` when a memset node is present, so that a caller can fall
back to eager execution BEFORE the first replay instead of training on stale values (or aborting).
"""
import ctypes

import torch

_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
               7: "event_record", 8: "ext_semaphore_signal", 9: "ext_semaphore_wait", 10: "mem_alloc", 11: "mem_free"}
_hip = None


class MemsetNodeInGraph(RuntimeError):
    pass


def _lib():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        _hip.hipGraphGetNodes.restype = ctypes.c_int
        _hip.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _hip.hipGraphNodeGetType.restype = ctypes.c_int
    return _hip


def new_graph():
    """A CUDAGraph that keeps its captured (un-instantiated) graph, so that its nodes can be inspected."""
    return torch.cuda.CUDAGraph(keep_graph=True)


def node_census(graph):
    """{node type: count} of a graph captured into `new_graph()`."""
    hip = _lib()
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    rc = hip.hipGraphGetNodes(raw, None, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed ({rc})")
    nodes = (ctypes.c_void_p * max(1, n.value))()
    rc = hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed ({rc})")
    census = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        rc = hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        if rc != 0:
            raise RuntimeError(f"hipGraphNodeGetType failed ({rc})")
        name = _NODE_TYPES.get(t.value, f"type{t.value}")
        census[name] = census.get(name, 0) + 1
    return census


def assert_replayable(graph, what="captured graph"):
    """Raise MemsetNodeInGraph if the graph holds a memset node (unsafe to replay more than once on this stack)."""
    census = node_census(graph)
    if census.get("memset", 0):
        raise MemsetNodeInGraph(f"{what} holds {census['memset']} memset node(s) of {sum(census.values())}: a captured "
                                "hipMemsetAsync writes garbage from the second replay on (adaptpoint_amd/graphs.py); "
                                "run this step eagerly or replace the memset's producer")
    return census
