"""hipGraph hygiene for the captured steps (bench.py, the training-step benches, the tests).

Root cause of round 2's "stale PyTorch reductions from the second replay on" (DESIGN.md section 5): on this stack
(ROCm 7.2, PyTorch 2.10) a captured `hipMemsetAsync` -- a MEMSET node -- executes correctly on the first replay of the
instantiated graph and writes a garbage 32-bit pattern from the second replay on (scripts/debug_graph_memset.py:
`memset(buf, 0); buf += 1` gives 1, then -1291841535 ever after).  PyTorch's multi-block reductions zero their
semaphore buffer with cudaMemsetAsync before every launch (ATen/native/cuda/Reduce.cuh) and never reset it in the
kernel, so under replay their "last block" is never recognised again and the output keeps its old value; the same
defect behind a larger memset made a replay abort outright.  This build's kernels never use memsets (zero fills
are kernels: `apn_zero_fill`, the producers' own clears); what PyTorch captures around them is checked here.

`node_census(graph)` counts the nodes of a captured graph by type through the HIP graph API (child graphs included);
`assert_replayable(graph)` raises `MemsetNodeInGraph` when a memset node is present, so that a caller can fall
back to eager execution BEFORE the first replay instead of training on stale values (or aborting).

Two more traps of this stack END A CAPTURE WITH A HOST SEGFAULT inside `hipStreamEndCapture` (round 3: DESIGN.md
sections 5 and 7c; logs gpurun_out/tnew.log, tc.log, wg2.log); both are refused here with a Python error instead:

* an autograd graph of an EARLIER eager step kept alive across the capture (a loss / logits tensor with a grad_fn
  that the caller still holds): its AccumulateGrad nodes belong to the stream of that earlier step, the backward under
  capture synchronises with that stream -- the default stream in the crashing case -- and the capture dies.
  `held_accumulators(leaves)` finds such nodes, `capture(fn, leaves=...)` raises `StaleAutogradGraph` before capturing.
* a side lane re-forked from the main stream after the main stream waited for an event recorded INSIDE that lane
  (`wait_ready`) in the same capture: `fork` raises `LaneTopology`.  Fork and join by `wait_stream` only.
"""
import ctypes
import gc

import torch

_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
               7: "event_record", 8: "ext_semaphore_signal", 9: "ext_semaphore_wait", 10: "mem_alloc", 11: "mem_free"}
_hip = None


class MemsetNodeInGraph(RuntimeError):
    pass


class StaleAutogradGraph(RuntimeError):
    pass


class LaneTopology(RuntimeError):
    pass


def _lib():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        _hip.hipGraphGetNodes.restype = ctypes.c_int
        _hip.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _hip.hipGraphNodeGetType.restype = ctypes.c_int
        _hip.hipGraphChildGraphNodeGetGraph.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        _hip.hipGraphChildGraphNodeGetGraph.restype = ctypes.c_int
    return _hip


def new_graph():
    """A CUDAGraph that keeps its captured (un-instantiated) graph, so that its nodes can be inspected."""
    return torch.cuda.CUDAGraph(keep_graph=True)


def node_census(graph):
    """{node type: count} of a graph captured into `new_graph()`; the nodes of child graphs count with their own
    types (a memset inside a child graph is as unsafe as one at the top level)."""
    return _census_raw(ctypes.c_void_p(graph.raw_cuda_graph()), {}, 0)


def _census_raw(raw, census, depth):
    hip = _lib()
    n = ctypes.c_size_t(0)
    rc = hip.hipGraphGetNodes(raw, None, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed ({rc})")
    nodes = (ctypes.c_void_p * max(1, n.value))()
    rc = hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed ({rc})")
    for i in range(n.value):
        t = ctypes.c_int(-1)
        rc = hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        if rc != 0:
            raise RuntimeError(f"hipGraphNodeGetType failed ({rc})")
        name = _NODE_TYPES.get(t.value, f"type{t.value}")
        census[name] = census.get(name, 0) + 1
        if t.value == 4 and depth < 8:
            child = ctypes.c_void_p()
            rc = hip.hipGraphChildGraphNodeGetGraph(ctypes.c_void_p(nodes[i]), ctypes.byref(child))
            if rc != 0:
                raise RuntimeError(f"hipGraphChildGraphNodeGetGraph failed ({rc})")
            _census_raw(child, census, depth + 1)
    return census


def assert_replayable(graph, what="captured graph"):
    """Raise MemsetNodeInGraph if the graph holds a memset node (unsafe to replay more than once on this stack)."""
    census = node_census(graph)
    if census.get("memset", 0):
        raise MemsetNodeInGraph(f"{what} holds {census['memset']} memset node(s) of {sum(census.values())}: a captured "
                                "hipMemsetAsync writes garbage from the second replay on (adaptpoint_amd/graphs.py); "
                                "run this step eagerly or replace the memset's producer")
    return census


def held_accumulators(leaves):
    """Indices of the leaf tensors in `leaves` whose AccumulateGrad node is being kept alive by an autograd graph that
    still exists (somebody holds a tensor with a grad_fn from an earlier forward pass).  How: a leaf hands out its
    accumulator node through any fresh view; nobody holding it, the node dies with that view and the next request makes
    a new one -- so a mark left on the node (its `metadata` dict) is still there on the second request only when the
    node outlived the first.  Works on any device; makes no launch."""
    held = []
    token = object()
    for i, q in enumerate(leaves):
        if not (torch.is_tensor(q) and q.requires_grad and q.is_leaf):
            continue
        with torch.enable_grad():
            node = q.view_as(q).grad_fn.next_functions[0][0]
            node.metadata["apn_probe"] = token
            del node
            node = q.view_as(q).grad_fn.next_functions[0][0]
            if node.metadata.pop("apn_probe", None) is token:
                held.append(i)
            del node
    return held


def capture(fn, leaves=(), what="the captured step", capture_error_mode="global", names=None):
    """Capture `fn()` into a hipGraph with the checks this stack needs; returns (graph, fn's result, node census).
    Before: Python garbage is collected and `leaves` (the parameters / inputs that receive gradients inside fn) are
    checked for autograd graphs of earlier steps still alive -> `StaleAutogradGraph` (a segfault inside
    hipStreamEndCapture otherwise).  After: the node census (child graphs included) -> `MemsetNodeInGraph` when the
    graph is unsafe to replay.  The lane bookkeeping of `fork` / `wait_ready` starts fresh."""
    gc.collect()
    leaves = list(leaves)
    held = held_accumulators(leaves)
    if held:
        shown = [names[i] if names else f"#{i}" for i in held[:6]]
        raise StaleAutogradGraph(
            f"{what}: {len(held)} of {len(leaves)} leaf tensors ({', '.join(map(str, shown))}{', ...' if len(held) > 6 else ''}) "
            "still belong to an autograd graph of an earlier step -- a tensor with a grad_fn from that step (its loss, "
            "logits, an output) is being kept alive.  Under capture the backward pass would synchronise with that step's "
            "stream and hipStreamEndCapture crashes the process: drop or .detach() those tensors first")
    _MIDWAIT.clear()
    g = new_graph()
    with torch.cuda.graph(g, capture_error_mode=capture_error_mode):
        out = fn()
    _MIDWAIT.clear()
    return g, out, assert_replayable(g, what)


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
_JOINS = {}              # lane key -> how often it has been joined (a caller can tell whether somebody else joined it)
_MIDWAIT = set()         # streams (ids) on which the main stream waited for an event recorded mid-lane, this capture


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
    if torch.cuda.is_current_stream_capturing():
        if s.cuda_stream in _MIDWAIT:
            raise LaneTopology(f"lane '{key}' is forked again after the main stream waited for an event recorded inside "
                               "it (graphs.wait_ready) in the same capture: on this stack that topology ends the capture "
                               "with a host segfault in hipStreamEndCapture (DESIGN.md 7c).  Join the lane (wait_stream) "
                               "instead of waiting for events inside it, or use one lane per role")
    else:
        _MIDWAIT.clear()
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
    _JOINS[s.cuda_stream] = _JOINS.get(s.cuda_stream, 0) + 1
    for t in outputs:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(main)


def ready_event():
    """An event recorded on the current stream (a branch's 'this part is ready' signal; `wait_ready` is its other end)."""
    ev = torch.cuda.Event()
    ev.record()
    ev.apn_stream = torch.cuda.current_stream().cuda_stream
    return ev


def joins(s):
    """How often the side stream `s` has been joined so far (`join`)."""
    return _JOINS.get(s.cuda_stream, 0)


def wait_ready(ev, *tensors):
    if ev is not None:
        main = torch.cuda.current_stream()
        if torch.cuda.is_current_stream_capturing() and getattr(ev, "apn_stream", None) not in (None, main.cuda_stream):
            _MIDWAIT.add(ev.apn_stream)
        main.wait_event(ev)
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
