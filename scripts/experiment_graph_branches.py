"""How does a replayed hipGraph run PARALLEL BRANCHES on this stack?  Two independent chains of K low-occupancy kernels
(FPS of 4 clouds: 4 workgroups, ~40 us each) captured (A) on one stream, (B) on two forked streams, one branch
captured after the other, (C) on two forked streams, captured alternately, (D) as B with a short common tail that
waits for both, (E) branch 2 forked late: behind half of branch 1.  Prints ms per replay; ideal for B-E is A / 2.
(F) three chains: the main chain waits, quarter by quarter, for EVENTS recorded inside side chain 1 (a producer consumed
part by part), side chain 2 is independent: ideal ~ K kernels + a quarter; (G) as F with one wait for all of chain 1
before the main chain: ideal 2K; (J) the producer's parts each on a stream of their own, joined by wait_stream where
they are consumed; (K) = J without the independent chain.  (L) a lane with work queued is re-forked from the main stream: part 2 must start behind
part 1.  (Recording an event in chain 1, waiting for it on the main
stream and then continuing chain 1 -- capture in the order of use -- crashed the process in hipGraph capture.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from adaptpoint_amd.layers import furthest_point_sample
from adaptpoint_amd import graphs

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
xa = torch.rand(4, 1024, 3, device=dev)
xb = torch.rand(4, 1024, 3, device=dev)
xc = torch.rand(4, 1024, 3, device=dev)
xd = torch.rand(4, 1024, 3, device=dev)
extra = [torch.cuda.Stream() for _ in range(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def work(x):
    return furthest_point_sample(x, 128)


def variant(kind):
    main = torch.cuda.current_stream()
    keep = []
    if kind == "A":
        for i in range(K):
            keep.append(work(xa)); keep.append(work(xb))
        return keep
    s1.wait_stream(main); s2.wait_stream(main)
    if kind in ("B", "D"):
        with torch.cuda.stream(s1):
            for i in range(K):
                keep.append(work(xa))
        with torch.cuda.stream(s2):
            for i in range(K):
                keep.append(work(xb))
    elif kind == "C":
        for i in range(K):
            with torch.cuda.stream(s1):
                keep.append(work(xa))
            with torch.cuda.stream(s2):
                keep.append(work(xb))
    elif kind == "E":
        with torch.cuda.stream(s1):
            for i in range(K // 2):
                keep.append(work(xa))
        s2.wait_stream(s1)
        with torch.cuda.stream(s1):
            for i in range(K // 2):
                keep.append(work(xa))
        with torch.cuda.stream(s2):
            for i in range(K // 2):
                keep.append(work(xb))
    elif kind in ("F", "G"):
        evs = []
        with torch.cuda.stream(s1):
            for i in range(K):
                keep.append(work(xa))
                if (i + 1) % (K // 4) == 0:
                    graphs.mark(f"chain 1: quarter {(i + 1) // (K // 4)} done")
                    ev = torch.cuda.Event(); ev.record(); evs.append(ev)
        with torch.cuda.stream(s2):
            for i in range(K):
                keep.append(work(xb))
            graphs.mark("chain 2 done")
        if kind == "G":
            main.wait_stream(s1)
        for i in range(K):
            if kind == "F" and i % (K // 4) == 0:
                main.wait_event(evs[i // (K // 4)])
            keep.append(work(xc))
            if (i + 1) % (K // 4) == 0:
                graphs.mark(f"main: quarter {(i + 1) // (K // 4)} done")
    elif kind in ("J", "K"):
        # a producer chain consumed part by part WITHOUT events waited for later: after every quarter of chain 1 a
        # stream of its own forks from it (wait_stream right there), runs the part's tail (2 kernels), and the main
        # chain joins THAT stream before its quarter.  (K: without the independent chain 2.)  ideal ~ 1.25 K + 2
        if kind == "J":
            with torch.cuda.stream(s2):
                for i in range(K):
                    keep.append(work(xb))
                graphs.mark("chain 2 done")
        parts = []
        for q in range(4):
            with torch.cuda.stream(s1):
                for i in range(K // 4):
                    keep.append(work(xa))
                graphs.mark(f"chain 1: quarter {q + 1} done")
            lq = extra[q]
            lq.wait_stream(s1)
            with torch.cuda.stream(lq):
                keep.append(work(xd)); keep.append(work(xd))
            parts.append(lq)
        for q in range(4):
            main.wait_stream(parts[q])
            for i in range(K // 4):
                keep.append(work(xc))
            graphs.mark(f"main: quarter {q + 1} done")
    elif kind == "L":
        # a lane that is re-forked from the main stream while it still has work of its own queued: does the captured
        # graph keep the lane's own order (part 1 -> part 2) next to the new dependency on the main stream?
        with torch.cuda.stream(s1):
            for i in range(K // 2):
                keep.append(work(xa))
            graphs.mark("lane: part 1 done")
        for i in range(K // 8):
            keep.append(work(xc))
        graphs.mark("main: its own part done")
        s1.wait_stream(main)
        with torch.cuda.stream(s1):
            graphs.mark("lane: part 2 starts")
            for i in range(K // 8):
                keep.append(work(xa))
    main.wait_stream(s1); main.wait_stream(s2)
    if kind == "D":
        for i in range(4):
            keep.append(work(xa))
    return keep


for kind in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("A", "B", "F", "G", "J", "K")):
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        variant(kind)
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    graphs.STAMPS = graphs.PhaseStamps(dev)
    with torch.cuda.graph(g):
        graphs.mark("start")
        keep = variant(kind)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print(kind, f"{(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per replay ({len(keep)} kernels + fills)", flush=True)
    if len(graphs.STAMPS.names) > 1:
        print("   ", ", ".join(f"{n} {us:.0f}" for n, us in graphs.STAMPS.report()))
    graphs.STAMPS = None
