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


class PhaseStamps:
    """Diagnostic: device wall-clock stamps at named points of a step, taken by one-thread launches on whatever stream
    is current (`apn_debug_stamp`) -- capturable, so a replayed graph reports when each of its branches reached each
    point, with no profiler attached.  Install with `graphs.STAMPS = PhaseStamps(dev)`; the steps call
    `graphs.mark(name)`, which does nothing while STAMPS is None."""

    def __init__(self, dev, slots=256):
        self.dev = dev
        self.buf = torch.zeros(slots, dtype=torch.int64, device=dev)
        self.names = []

    def mark(self, name):
        from .fused import _call
        if len(self.names) >= self.buf.numel():
            raise RuntimeError("PhaseStamps: out of slots")
        _call("apn_debug_stamp", self.dev, self.buf.data_ptr(), len(self.names))
        self.names.append(name)

    def report(self):
        """[(name, microseconds since the first stamp)] in time order (the clock runs at 100 MHz)."""
        t = self.buf[:len(self.names)].cpu().tolist()
        t0 = min(t)
        return sorted(((n, (v - t0) / 100.0) for n, v in zip(self.names, t)), key=lambda x: x[1])


STAMPS = None


def mark(name):
    if STAMPS is not None:
        STAMPS.mark(name)


def mark_grad(tensor, name):
    """Stamp the moment the backward pass has formed the gradient of `tensor` (a hook; only while STAMPS is set)."""
    if STAMPS is not None and torch.is_tensor(tensor) and tensor.requires_grad:
        def hook(g, _name=name):
            mark(_name)
        tensor.register_hook(hook)
    return tensor


# ------------------------------------------------------------------ side streams (parallel branches of a captured step)
# A replayed hipGraph runs independent branches concurrently (scripts/experiment_graph_branches.py: two chains of 40
# low-occupancy kernels on two forked streams replay in half the time of one chain).  The training steps use that for
# work that depends on coordinates only (FPS chains, ball queries, three_nn, tile / inverse maps) and for sub-networks
# that do not feed each other.  Autograd runs a backward node on the stream its forward ran on, so a branch forked in
# the forward pass is a branch of the backward pass too.  Everything here is a no-op unless `overlapping(True)` is active.
_OVERLAP = False
_SIDE = {}
LANE2 = "lane2"          # the ONE side stream the steps fork onto (two concurrent branches is what replays concurrently)


def overlap_enabled():
    return _OVERLAP


class overlapping:
    """`with overlapping(on):` -- the modules called inside fork their independent parts onto side streams."""

    def __init__(self, on=True):
        self.on = on if isinstance(on, str) else bool(on)

    def __enter__(self):
        global _OVERLAP
        self.prev, _OVERLAP = _OVERLAP, self.on
        return self

    def __exit__(self, *exc):
        global _OVERLAP
        _OVERLAP = self.prev


def side_stream(key, dev):
    dev = torch.device(dev)
    k = (key, dev.index if dev.index is not None else torch.cuda.current_device())
    if k not in _SIDE:
        _SIDE[k] = torch.cuda.Stream(dev)
    return _SIDE[k]


def fork(key, dev, *inputs):
    """The side stream `key` of `dev`, made to wait for everything queued on the current stream; `inputs`: tensors of
    the current stream that the branch reads (the caching allocator is told, so that a block freed by the caller is
    not handed out again before the branch has read it)."""
    main = torch.cuda.current_stream(dev)
    s = side_stream(key, dev)
    s.wait_stream(main)
    for t in inputs:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(s)
    return s


def join(s, *outputs):
    """The current stream waits for the side stream `s`; `outputs`: tensors the branch allocated that the caller goes on
    to read."""
    main = torch.cuda.current_stream(s.device)
    main.wait_stream(s)
    for t in outputs:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(main)


def ready_event():
    """An event recorded on the current stream (a branch's 'this part is ready' signal; `wait_ready` is its other end)."""
    ev = torch.cuda.Event()
    ev.record()
    return ev


def wait_ready(ev, *tensors):
    if ev is not None:
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
