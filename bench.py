#!/usr/bin/env python3
"""bench.py -- set-abstraction forward+backward throughput on MI355X.

Metric (BASELINE.json): set-abstraction fwd+bwd point-clouds/sec at B=32, N=1024
(npoint=512, nsample=32), at 1/2/4/8 GPUs.  Workload = BASELINE.json configs[1]:
PointNeXt-S stage-1 SetAbstraction (cfgs/scanobjectnn/pointnext-s.yaml:5-36;
openpoints/models/backbone/pointnext.py:82-170): FPS 1024->512, ball query r=0.15
K=32, group xyz + 32 features, dp/radius, Conv2d 35->32 BN ReLU, Conv2d 32->64 BN,
max over K, skip Conv1d 32->64, ReLU; loss = out.sum(); grads w.r.t. features and
weights.  A "step" is one such pass over one batch of 32 synthetic clouds per GPU
(distribution D1 of SURVEY.md section 8d: uniform cube -> centred -> unit sphere).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU, batch sharded by cloud (weak scaling, 32 per
GPU), SyncBatchNorm + gradient all-reduce over RCCL as the reference does at
world_size > 1 (examples/classification/main.py:27, train_autoaug.py:275-282):
`--sync-bn auto` is ON at world_size > 1; the per-rank-BatchNorm figure (not what the
reference runs) is reported beside it as `value_no_syncbn`.

Prints ONE JSON line on rank 0 (see the contract in the task description).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B_PER_GPU, N_PTS, NPOINT, NSAMPLE, C_IN, C_OUT, RADIUS = 32, 1024, 512, 32, 32, 64, 0.15
# What the DEFAULT configuration launches (short names of the per-kernel pass): seven on the MLP stream per step, five on
# the index stream per 20 batches.  The committed PMC traffic file (profiles/r0N_traffic.json) must cover exactly this
# set -- tests/test_host_cpu.py checks it, and the line says whether the kernels seen at run time equal it.
DEFAULT_KERNELS = ("sa_prep_stats", "sa_fwd_main", "sa_fwd_out", "sa_bwd_prep", "sa_bwd_main", "sa_bwd_point_grads",
                   "sa_bwd_finalize", "fps", "ball_query", "sa_point_geo", "sa_wide_tilemap_many", "sa_rowmap_many")
TRAFFIC_FILE = "r05_traffic.json"        # PMC bytes per launch of the kernels above (scripts/collect_profiles.sh pmc)
MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_block(fused=False, sync_bn=False):
    from adaptpoint_amd.set_abstraction import SetAbstraction
    return SetAbstraction(C_IN, C_OUT, layers=2, stride=2, fused=fused, sync_bn=sync_bn,
                          group_args={'NAME': 'ballquery', 'radius': RADIUS, 'nsample': NSAMPLE,
                                      'normalize_dp': True},
                          norm_args={'norm': 'bn'}, act_args={'act': 'relu'},
                          conv_args={'order': 'conv-norm-act'}, sampler='fps',
                          feature_type='dp_fj', use_res=True)


def make_inputs(batch, seed, distribution="D1"):
    """D1 = uniform cube, centred, scaled to the unit sphere (~9 neighbours in r=0.15);
    D2 = unit-sphere surface + N(0, 0.01) jitter (scan-like, ~6)."""
    from adaptpoint_amd import synthetic as GI
    cloud = GI.unit_sphere_cloud if distribution == "D1" else GI.sphere_surface_cloud
    p = torch.from_numpy(cloud(batch, N_PTS, seed=seed))
    f = torch.from_numpy(GI.seeded_normal((batch, C_IN, N_PTS), seed=seed + 7))
    return p, f


# SURVEY 8d, per cloud through config 2 (fwd+bwd): compulsory HBM bytes and shared-MLP flops
STEP_BYTES_PER_CLOUD = 14336 + 83968 + 346112 + 458752          # FPS + ball query + fused fwd + fused bwd
STEP_FLOPS_PER_CLOUD = 3 * (2 * NPOINT * NSAMPLE * (35 * 32 + 32 * 64) + 2 * NPOINT * 32 * 64)


# Algorithmic bytes one launch of each kernel must move (per cloud
# x the clouds of one launch); see DESIGN.md "Kernels and rooflines".
def algorithmic_bytes(batch, fused=True):
    mk = NPOINT * NSAMPLE
    return {
        # xyz in, idx + sampled xyz out; the drop-in op also reads and writes temp (the reference's
        # scratch, part of its contract), the fused block's sampler keeps it in registers
        "fps": batch * (N_PTS * 12 + (0 if fused else 2 * N_PTS * 4) + NPOINT * 4 + NPOINT * 12),
        "ball_query": batch * (N_PTS * 12 + NPOINT * 12 + mk * 4),         # xyz, queries, idx out
        "group_xyz": batch * (3 * N_PTS * 4 + mk * 4 + 3 * mk * 4),        # rows, idx, out
        "group_feat": batch * (C_IN * N_PTS * 4 + mk * 4 + C_IN * mk * 4),
        "group_feat_grad": batch * (C_IN * mk * 4 + mk * 4 + 2 * C_IN * N_PTS * 4),
        # index stage's occurrence statistics: idx + coordinates in, int64 geo (B,N,4) out
        "sa_point_geo": batch * (mk * 4 + N_PTS * 12 + NPOINT * 12 + N_PTS * 32),
        # per-point pass: f + geo in, the bf16 operand table(s) out
        "sa_prep_stats": batch * (C_IN * N_PTS * 4 + N_PTS * 32 + (2 if fused else 1) * N_PTS * 64),
        # fused passes: xyz + queries + idx + bf16 feature table in; pooled outputs / G, H out
        "sa_fwd_main": batch * (N_PTS * 12 + NPOINT * 12 + mk * 4 + N_PTS * 64 + NPOINT * 64 * 5),
        # + goa, ksel in; per-point sums A (128 B), per-query sums HA, HB out
        "sa_bwd_main": batch * (N_PTS * 12 + NPOINT * 12 + mk * 4 + N_PTS * 64 + NPOINT * 64 * 5
                                + N_PTS * 128 + NPOINT * 256),
    }


class KernelTimer:
    """HIP-event timing of the extension's launches, on the stream they run on
    (ops.py launches on torch's current stream, which is what these events see)."""

    def __init__(self, burst=None, burst_n=10):
        self.pairs = {}
        self.burst, self.burst_n, self.bursts = burst, burst_n, []

    def wrap(self, name, fn):
        def timed(*a, **k):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            self.pairs.setdefault(name, []).append((s, e))
            if name == self.burst:
                # the same launch again, burst_n times between ONE pair of events, while its buffers are alive (what the
                # pass computes after it is not used): an event pair around a single 30 us launch also measures the
                # dispatch latency behind the start event, 2-5 us that vary from pass to pass -- the single-launch
                # figure read 32-36 us where rocprofv3 said 31
                s2 = torch.cuda.Event(enable_timing=True)
                e2 = torch.cuda.Event(enable_timing=True)
                s2.record()
                for _ in range(self.burst_n):
                    fn(*a, **k)
                e2.record()
                self.bursts.append((s2, e2))
            return r
        return timed

    def burst_us(self):
        torch.cuda.synchronize()
        t = sorted(1e3 * s.elapsed_time(e) / self.burst_n for s, e in self.bursts)
        return t[len(t) // 2]

    def stats_us(self):
        """Per kernel: (median, lower quartile) of its event pairs, in us.  The MEDIAN is what `roofline` and the per-kernel
        table report as the average launch duration (it agrees with rocprofv3's average within the run-to-run spread); the
        lower quartile is printed beside it as `p25_us` -- a pair can only read long (it sits behind a late host launch, or
        the dispatch behind its start event is slow), never short, so it is the better estimate of the kernel alone, but it
        is a selected statistic and never feeds `frac`."""
        torch.cuda.synchronize()
        out = {}
        for k, v in self.pairs.items():
            t = sorted(s.elapsed_time(e) for s, e in v)
            out[k] = (1e3 * t[len(t) // 2], 1e3 * t[len(t) // 4])
        return out

    def mean_us(self):
        return {k: v[0] for k, v in self.stats_us().items()}


def instrument(timer, only=None):
    """Route adaptpoint_amd.layers' calls into the extension through event pairs."""
    from adaptpoint_amd import ops
    saved = {}

    def key_group(b, c, *a):
        return "group_xyz" if c == 3 else "group_feat"

    def patch(attr, name_fn):
        orig = getattr(ops, attr)
        saved[attr] = orig

        def call(*a, **k):
            name = name_fn(*a) if callable(name_fn) else name_fn
            if only is not None and name not in only:
                return orig(*a, **k)
            return timer.wrap(name, orig)(*a, **k)
        setattr(ops, attr, call)

    patch("furthest_point_sampling_wrapper", "fps")
    patch("ball_query_wrapper", "ball_query")
    patch("group_points_wrapper", key_group)
    patch("group_points_grad_wrapper", lambda b, c, *a: "group_xyz_grad" if c == 3 else "group_feat_grad")

    from adaptpoint_amd import fused
    orig_call = fused._call
    fused.PER_KERNEL_LAUNCH = True        # one foreign call per kernel, so each can carry events

    alias = {"furthest_point_sampling_xyz": "fps", "ball_query_zero": "ball_query"}

    def fused_call(name, dev, *a, **kw):
        short = name.replace("apn_", "")
        short = alias.get(short, short)
        if short == "sa_sample_overlap":          # one launch, two roles: name it by the roles present
            short = {(True, False): "fps", (False, True): "ball_query",
                     (True, True): "fps+ball_query"}[(a[5] is not None, a[8] is not None)]
        if only is not None and short not in only:
            return orig_call(name, dev, *a, **kw)
        return timer.wrap(short, orig_call)(name, dev, *a, **kw)
    from adaptpoint_amd import fused_wide
    fused._call = fused_wide._call = fused_call     # (the width-generic family launches through its own import of it)

    def restore():
        for k, v in saved.items():
            setattr(ops, k, v)
        fused._call = fused_wide._call = orig_call
        fused.PER_KERNEL_LAUNCH = False
    return restore


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget=14.0):
    """The oracle port of the same block on the host cores: bounded samples, all threads and 1 thread."""
    from oracle import cpu_block as CB
    from oracle import oracle as O
    from adaptpoint_amd import dp
    threads = dp.host_threads(cap=64)            # affinity / cgroup aware, not os.cpu_count()
    p, f = make_inputs(B_PER_GPU, seed=0)

    def sample(nthreads, budget):
        torch.set_num_threads(nthreads)
        O.set_threads(nthreads)
        torch.manual_seed(0)
        blk = CB.build_cpu_block(make_block)
        blk.train()
        CB.run_step(blk, p, f)                   # warm-up (page-in, thread pools)
        t0 = time.perf_counter()
        iters = 0
        while True:
            CB.run_step(blk, p, f)
            iters += 1
            el = time.perf_counter() - t0
            if el >= budget or iters >= 60:
                return B_PER_GPU * iters / el, iters, el
    v_all, it_all, el_all = sample(threads, seconds_budget * 0.6)
    v_one, it_one, el_one = sample(1, seconds_budget * 0.4)
    torch.set_num_threads(threads)
    return {"value": v_all, "unit": "point-clouds/s", "cores": threads, "kind": "port",
            "value_1_thread": v_one, "cpu_model": cpu_model(),
            "sample": f"{it_all} fwd+bwd steps ({el_all:.1f} s) at {threads} threads and {it_one} "
                      f"({el_one:.1f} s) at 1 thread of the same block (B={B_PER_GPU}, N={N_PTS}, "
                      f"npoint={NPOINT}, nsample={NSAMPLE}), 1 warm-up each: C oracle (OpenMP) for "
                      f"FPS/ball/group, torch-CPU for conv/BN/max"}


class Measured:
    pass


def measure(args, dev, world, rank, local_rank, distributed, mlp, sync_bn, steps, warmup, repeats=1):
    """Build the block for one configuration, time `steps` steps per the contract, and return the
    elapsed seconds plus the eager single-step closure (for the per-kernel passes)."""
    from adaptpoint_amd import dp
    r = Measured()
    torch.manual_seed(0)                          # identical initial weights on every rank
    fused_mlp = mlp.startswith("fused")
    if fused_mlp:
        from adaptpoint_amd import fused as _f
        _f.PRECISION = mlp.split("-", 1)[1]
    blk = make_block(fused=fused_mlp, sync_bn=sync_bn).to(dev)
    blk.train()
    # Gradient exchange.  Without SyncBatchNorm forward+backward contain no collective, so the
    # step is the same captured hipGraph as at one GPU, followed by ONE flat-bucket all-reduce of
    # the block's gradients (adaptpoint_amd.dp.allreduce_mean_).  With SyncBatchNorm collectives sit
    # inside forward and backward: the fused block runs eagerly (its four statistics all-reduces
    # between the launch phases) with the same flat gradient all-reduce; the unfused PyTorch path
    # converts to adaptpoint_amd.dp.SyncBatchNormAllReduce (ONE all-reduce per layer and direction)
    # and averages the gradients with the same flat all-reduce.
    if sync_bn and not fused_mlp:
        blk = dp.convert_sync_batchnorm(blk)
    model = blk
    # --graph-collectives on: the statistics all-reduces and the gradient all-reduce are captured INTO the
    # hipGraph (RCCL kernels are ordinary stream work).  Capture runs with capture_error_mode="thread_local":
    # in the default "global" mode ProcessGroupNCCL's watchdog thread querying an event while any stream
    # captures invalidates the capture and aborts the process (round 1, hipErrorStreamCaptureUnsupported).
    capture_collectives = distributed and args.graph_collectives != "off"
    eager_collectives = distributed and sync_bn and not capture_collectives
    params = [q for q in blk.parameters()]

    # the two-stream pipeline pays only when the step is GPU-bound, i.e. under graph replay
    pipelined = fused_mlp and args.pipeline == "on"
    use_graph = ((args.graph == "on") or (args.graph == "auto")) and not eager_collectives
    # steps per graph: several whole steps per replay when no collective sits between steps
    spg = 1
    if use_graph and (not distributed or capture_collectives) and args.steps_per_graph != 1:
        for cand in ((args.steps_per_graph,) if args.steps_per_graph else (20, 10, 4, 2)):
            if cand > 1 and steps % cand == 0:
                spg = cand
                break
    # inputs: every one of the spg steps of a launch has its own clouds (the ball query's cost is
    # data dependent); each rank draws its own shard
    seed0 = dp.shard_seed(args.seed, rank)
    pf = [make_inputs(B_PER_GPU, seed=seed0 + 31 * i, distribution=args.distribution) for i in range(spg)]
    p_all = torch.cat([a for a, _ in pf]).to(dev)                       # (spg*B, N, 3)
    ps = [p_all[i * B_PER_GPU:(i + 1) * B_PER_GPU] for i in range(spg)]
    fs = [b.to(dev).requires_grad_(True) for _, b in pf]
    index_batch = spg if args.index_batch == 0 else max(1, min(spg, args.index_batch))
    while spg % index_batch:
        index_batch -= 1
    if pipelined:
        from adaptpoint_amd.fused import Sampling
        side_stream = torch.cuda.Stream(priority=int(os.environ.get("APN_BENCH_SIDE_PRIORITY", "0")))
        # Two sets of spg index-stage results: a launch of spg steps consumes set `cur` on the
        # main stream while the side stream fills the other set for the NEXT launch -- the
        # streams meet once per launch (fork at its start, join at its end), not once per step:
        # inside a hipGraph every cross-queue dependency costs ~10 us of idle queue.
        # Each set is ONE stacked buffer: the index stages of `index_batch` batches run as one
        # FPS launch + one ball-query launch over index_batch * B clouds (each cloud still one
        # workgroup: the serial chains of different batches are independent and a batch of 32
        # keeps only 32 of 256 CUs busy).
        big = [Sampling(spg * B_PER_GPU, NPOINT, NSAMPLE, dev) for _ in range(2)]
        for b in big:
            b.alloc_geo(N_PTS)                       # the index stage also leaves the neighbourhoods' occurrence statistics
        sets = [[b.clouds(i * B_PER_GPU, (i + 1) * B_PER_GPU) for i in range(spg)] for b in big]
        blk.sample(p_all, out=big[0])                # prologue: index stages of the first launch
        big[1].buf.copy_(big[0].buf)
        big[1].geo.copy_(big[0].geo)
        big[1].dd.copy_(big[0].dd)
        # the width-generic kernels also take a tile map + inverse map of the neighbourhoods (index-stage work too)
        for st in sets:
            for i in range(spg):
                blk.index_for(st[i], N_PTS, C_IN)
        # register-resident kernels: the tile maps of a launch's spg batches live in ONE buffer per set and are built by
        # one pair of launches over the stacked neighbour array (forty launches of 12-15 us per replay before)
        big_maps = None
        if all(x.tmap is not None and x.index is None for st in sets for x in st):
            from adaptpoint_amd import fused_wide
            big_maps = [fused_wide.tile_maps(b.idx, spg) for b in big]
            # ... and their row maps (the backward pass stores g_u's rows in point-sorted order: no float atomics), the twenty
            # of a replay in ONE launch
            big_rows = [fused_wide.row_maps(maps, spg, B_PER_GPU, N_PTS, NPOINT, fidx=b.fidx) for maps, b in zip(big_maps, big)]
            for st, maps, rows_ in zip(sets, big_maps, big_rows):
                for i in range(spg):
                    st[i].tmap = maps[i]
                    st[i].rowmap = (rows_[0][i], rows_[1][i])
    cur_set = [0]
    ones = torch.ones(1, 1, 1, device=dev)
    graph_grads, last_grads = {}, [None]

    def clear_grads():
        for f in fs:
            f.grad = None
        for q in params:
            q.grad = None

    def mlp_steps(count, cur, first=0):
        """`count` consecutive MLP forward+backward steps on the current stream; pipelined: step i
        takes its index stage from sets[cur][i], else the block computes it in line."""
        for i in range(first, first + count):
            clear_grads()
            new_p, out = model([ps[i], fs[i]], sampling=sets[cur][i]) if pipelined else model([ps[i], fs[i]])
            if fused_mlp:
                # loss = out.sum(): its gradient is a constant 1 for every element, handed to the backward
                # entry as a stride-0 broadcast (the fused backward reads the upstream gradient with its
                # strides); the scalar itself is not formed -- no PyTorch reduce / fill launch in the step
                torch.autograd.backward([out], [ones.expand_as(out)])
            else:
                out.sum().backward()

    def index_steps(count, dst):
        if args.index_overlap == "on":
            # FPS of batch i and ball query of batch i-1 in one launch, batch by batch
            blk.sample_many(ps[:count], outs=sets[dst][:count])
            return
        nb = index_batch * B_PER_GPU
        for g in range(0, count * B_PER_GPU, nb):
            hi = min(g + nb, count * B_PER_GPU)
            blk.sample(p_all[g:hi], out=big[dst].clouds(g, hi))
        if big_maps is not None and count == spg:
            fused_wide.tile_maps(big[dst].idx, spg, out=big_maps[dst])
            fused_wide.row_maps(big_maps[dst], spg, B_PER_GPU, N_PTS, NPOINT, out=big_rows[dst], fidx=big[dst].fidx)
            return
        for i in range(count):
            if sets[dst][i].index is not None or sets[dst][i].tmap is not None:
                blk.index_for(sets[dst][i], N_PTS, C_IN, out=sets[dst][i].index)

    # Pipelined launch of `count` steps: the MLP steps consume set `cur` on the main stream while
    # the side stream fills the other set for the NEXT launch.  mlp_done / index_done order a
    # launch after the previous launch of the OTHER stream, whose set it is about to touch.
    mlp_done, index_done = torch.cuda.Event(), torch.cuda.Event()

    def launch(count, cur, run_mlp, run_index):
        main = torch.cuda.current_stream()
        if not pipelined:
            run_mlp(count, cur)
            return
        side_stream.wait_event(mlp_done)              # the set to refill was read by the last launch
        with torch.cuda.stream(side_stream):
            run_index(count, 1 - cur)
            next_index_done = torch.cuda.Event()
            next_index_done.record(side_stream)
        main.wait_event(index_done_box[0])            # the set to read was filled by the last launch
        run_mlp(count, cur)
        mlp_done.record(main)
        index_done_box[0] = next_index_done

    index_done_box = [index_done]
    index_done.record(torch.cuda.current_stream())    # prologue above filled set 0
    mlp_done.record(torch.cuda.current_stream())

    def step():
        launch(1, cur_set[0], mlp_steps, index_steps)
        cur_set[0] ^= 1

    eager_step = step
    if use_graph:
        # Whole-step capture: every launch of spg steps (extension kernels through ctypes on the
        # capture stream, PyTorch ops, autograd) becomes one hipGraph; replay has no host work,
        # and the ~20-30 us the GPU idles between two graph launches is paid once per spg steps.
        # Pipelined: the MLP steps and the index stages are SEPARATE single-branch graphs (one per
        # orientation of the two sets), replayed side by side on the two streams -- a two-branch
        # graph is not reliably run on two queues, and every cross-queue edge inside a graph
        # costs ~10 us of idle queue.
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            # Nothing may be in flight when capture starts: at N>1 the process group exists
            # already, and ProcessGroupNCCL's watchdog querying an event of outstanding collective
            # work on a capturing stream invalidates the capture (the abort of round 1,
            # hipErrorStreamCaptureUnsupported).  Drain the device, then all ranks meet.
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
                torch.cuda.synchronize()
            from adaptpoint_amd import graphs
            mlp_graphs, index_graphs = {}, {}
            r.graph_nodes = {}
            for cur in ((0, 1) if pipelined else (0,)):
                clear_grads()

                def mlp_capture():
                    if capture_collectives:
                        for i in range(spg):          # every step: fwd + bwd (+ statistics exchanges) + gradient all-reduce
                            mlp_steps(1, cur, first=i)
                            dp.allreduce_mean_([q.grad for q in params if q.grad is not None])
                    else:
                        mlp_steps(spg, cur)
                # graphs.capture: no autograd graph of an earlier step may be alive (a crash inside hipStreamEndCapture
                # otherwise), and a memset node (PyTorch's zero fills and multi-block reductions lower to one) writes
                # garbage from the second replay on, or aborts the process, on this stack: found HERE, before any replay,
                # either only costs the graph (the except below falls back to eager execution)
                g, _, r.graph_nodes["mlp"] = graphs.capture(mlp_capture, leaves=params + fs, what="the MLP steps' graph",
                                                            capture_error_mode="thread_local" if distributed else "global")
                mlp_graphs[cur] = g
                # each capture owns its gradient tensors; a replay refreshes them in place
                graph_grads[cur] = [q.grad for q in params if q.grad is not None]
                if pipelined:
                    index_graphs[cur], _, r.graph_nodes["index"] = graphs.capture(lambda: index_steps(spg, cur),
                                                                                  what="the index stages' graph")

            def step():
                cur = cur_set[0]
                launch(spg, cur, lambda n, c: mlp_graphs[c].replay(),
                       (lambda n, d: None) if args.diag_freeze_index else (lambda n, d: index_graphs[d].replay()))
                last_grads[0] = graph_grads[cur]
                if pipelined:
                    cur_set[0] ^= 1
        except Exception as exc:                      # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly",
                  file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            use_graph = False
            spg = 1
            step = eager_step
    r.capture_collectives = capture_collectives and use_graph
    if distributed and not r.capture_collectives:
        local_step = step

        def step():
            local_step()
            dp.allreduce_mean_(last_grads[0] if use_graph
                               else [q.grad for q in params if q.grad is not None])

    # W warm-up steps, barrier + synchronize, K timed steps, barrier + synchronize, MAX over ranks
    # (a replay of an spg-step graph counts as spg steps: exactly `steps` steps are timed)
    # (warm-up is rounded UP to whole replays: at least `warmup` untimed steps)
    r.elapsed = dp.timed_steps(step, steps // spg, -(-warmup // spg), dev)
    # A short timed region (the driver's --steps 20 is ONE graph replay of 3 ms) is repeated: the same K-step block,
    # fenced on both sides every time, `repeats` times; the MEDIAN block is reported, min / max beside it.
    r.blocks = [r.elapsed]
    if repeats > 1:
        for _ in range(repeats - 1):
            r.blocks.append(dp.timed_steps(step, steps // spg, 0, dev))
        r.elapsed = sorted(r.blocks)[len(r.blocks) // 2]
    r.steps, r.spg, r.use_graph, r.pipelined, r.fused_mlp = steps, spg, use_graph, pipelined, fused_mlp
    r.index_batch = index_batch if pipelined and args.index_overlap != "on" else 1
    r.index_verified = None
    if pipelined:
        # The index stages ran BESIDE the MLP kernels all through the timed region (two graphs on two streams).  Their
        # inputs never change, so both sets must hold exactly what the same index stage yields on an idle device:
        # FPS picks, sampled coordinates, neighbours and occurrence counts, bit for bit.
        torch.cuda.synchronize()
        ref = Sampling(spg * B_PER_GPU, NPOINT, NSAMPLE, dev)
        ref.alloc_geo(N_PTS)
        blk.sample(p_all, out=ref)
        torch.cuda.synchronize()
        r.index_verified = bool(all(torch.equal(b.buf, ref.buf) and torch.equal(b.geo, ref.geo) for b in big))
        if not r.index_verified:
            # say what differs (clouds per set): a wrong index stage voids the run's parity claim
            r.index_verified = {"sets": [{"clouds_with_other_fps_picks": int((b.fidx != ref.fidx).any(1).sum()),
                                          "clouds_with_other_neighbours": int((b.idx != ref.idx).flatten(1).any(1).sum()),
                                          "clouds_with_other_counts": int((b.geo != ref.geo).flatten(1).any(1).sum())}
                                         for b in big], "clouds": spg * B_PER_GPU}
            if os.environ.get("APN_BENCH_INDEX_DIAG"):
                import ctypes
                from adaptpoint_amd import _lib as _L
                cells = (ctypes.c_uint * 8)()
                fn = getattr(ctypes.CDLL(_L.LIB_PATH), "apn_fps_debug_cells", None)
                if fn is not None:
                    fn(cells)
                    print("[index diag] FPS threads whose LDS table entry changed under them:", cells[0], "of", cells[1],
                          "| steps where a wave's maximum beat the slot's:", cells[2], "| where no lane of a wave fired:", cells[3],
                          "| where several fired:", cells[4],
                          "| steps whose slot paired one lane's rank with another's distance:", cells[5], "of", cells[6], file=sys.stderr)
                for b in big:
                    wrong = (b.fidx != ref.fidx).any(1).nonzero().flatten().tolist()
                    for c in wrong[:6]:
                        j = int((b.fidx[c] != ref.fidx[c]).nonzero()[0])
                        print(f"[index diag] cloud {c}: first differing pick at step {j}: got {int(b.fidx[c, j])}, alone "
                              f"{int(ref.fidx[c, j])}; previous picks {b.fidx[c, max(0, j - 3):j].tolist()}; got-pick seen "
                              f"before: {int(b.fidx[c, j]) in b.fidx[c, :j].tolist()}; next {b.fidx[c, j + 1:j + 4].tolist()} vs "
                              f"{ref.fidx[c, j + 1:j + 4].tolist()}", file=sys.stderr, flush=True)
        del ref

    r.mlp_verified = None
    if pipelined and use_graph and fused_mlp and not distributed and r.index_verified is True:
        # ... and the MLP kernels ran beside the index kernels: the parameter gradients the LAST replayed step left
        # against the same step launched eagerly on the idle device (same index set): BIT FOR BIT since round 5 (no float
        # atomics on the chain: the per-point sums are ordered sums of stored rows).
        torch.cuda.synchronize()
        got = [g_.detach().clone() for g_ in last_grads[0]]
        mlp_steps(1, cur_set[0] ^ 1, first=spg - 1)
        torch.cuda.synchronize()
        want = [q.grad for q in params if q.grad is not None]
        dev_ = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(got, want))
        r.mlp_verified = {"max_gradient_deviation_rel": dev_, "bit_identical": bool(all(torch.equal(a, b) for a, b in zip(got, want)))}

    def eager_launch():
        """One whole launch (spg steps and, pipelined, the index stages of the next spg) issued
        eagerly, kernel by kernel: what the per-kernel event passes time."""
        launch(spg, cur_set[0], mlp_steps, index_steps)
        if pipelined:
            cur_set[0] ^= 1
    r.eager_step = eager_launch
    return r


def workload_main(args, dev, world, rank, distributed):
    """--workload classifier | gan | adaptpoint: BASELINE configs[2] / [3] / [4] per GPU (B=32 clouds each), the same
    contract as the block: W warm-up steps, K timed steps between barrier + synchronize fences, MAX over ranks, one
    JSON line.  Under torch.distributed the networks are wrapped as the reference wraps them
    (adaptpoint_amd/workloads.py); the SAFE launch structure (the step run eagerly around its collectives) is measured
    FIRST, the captured one (collectives inside the hipGraph, thread-local capture) after it; `value` is the captured
    figure when the capture succeeds, else the eager one."""
    from adaptpoint_amd import dp, graphs, workloads
    from adaptpoint_amd import set_abstraction as _sa
    steps = args.steps if args.steps is not None else 40
    warmup = args.warmup if args.warmup is not None else 5
    job = workloads.build(args.workload, dev, batch=B_PER_GPU, npoints=args.points, fused=True, distributed=distributed,
                          capturable=True, overlap=(args.overlap == "on"), seed=args.seed)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up off the default stream (optimizer state, autotuning)
        for _ in range(3):
            job.step()
        coll = workloads.collectives_per_step(job) if distributed else {}
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        el_eager = dp.timed_steps(job.step, steps, warmup, dev)
    value_eager = B_PER_GPU * world * steps / el_eager
    launch, value, elapsed, nodes, last = "eager", value_eager, el_eager, None, None
    if args.graph != "off":
        try:
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
                torch.cuda.synchronize()
            for q in job.parameters():
                q.grad = None
            g, last, nodes = graphs.capture(job.step, leaves=job.parameters(), what=f"the {args.workload} step's graph",
                                            capture_error_mode="thread_local" if distributed else "global")
            elapsed = dp.timed_steps(g.replay, steps, warmup, dev)
            value = B_PER_GPU * world * steps / elapsed
            launch = "hipGraph replay" + (", collectives captured" if distributed else "")
        except Exception as exc:                      # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture of the {args.workload} step failed ({type(exc).__name__}: {exc}); "
                  "reporting the eager figure", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
    names = {"classifier": "PointNeXt-S classifier training step (fwd + SmoothCE + bwd + clip + AdamW), BASELINE configs[2]",
             "gan": "AdaptPoint joint step: generator (Deformation + Mask controllers) + discriminator + feedback through the "
                    "eval-mode classifier, two Adam steps, BASELINE configs[3]",
             "adaptpoint": "PointNeXt-S + AdaptPoint end-to-end training step: one train_gan iteration, then one classifier "
                           "iteration on the generated clouds (train_autoaug.py:368-394), BASELINE configs[4] per GPU"}
    result = {
        "metric": f"{args.workload} training-step point-clouds/sec (B={B_PER_GPU}/GPU, N={args.points})",
        "value": round(value, 2), "unit": "point-clouds/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "value_eager_collectives" if distributed else "value_eager": round(value_eager, 2),
        "config": {"workload": names[args.workload] + f", B={B_PER_GPU}/GPU N={args.points}, random 15-class labels, "
                               "random-init weights", "global_batch": B_PER_GPU * world, "launch": launch,
                   "graph_nodes": nodes, "two_lane_joint_step": args.overlap == "on" and args.workload != "classifier",
                   "fused_fallbacks": sum(_sa.FUSED_FALLBACKS.values()),
                   "resampler": ("FPS to 1200 + a random 1024 of them, the subset drawn ONCE per run and kept on the device "
                                 "(the reference draws it per batch on the host)" if job.choice is not None else None),
                   "syncbn": job.syncbn,
                   "collectives_per_step": coll or None,
                   "parallelism": f"dp{world}" + (("+syncbn(classifier)" if job.syncbn else "") + "+flat-allreduce per network: "
                                                  f"{coll.get('all_reduce', 0)} all-reduces, {coll.get('bytes', 0)} B per step"
                                                  if distributed else "")},
        "roofline": None, "cpu_baseline": None,
        "losses": None if last is None else {k: round(float(v), 5) for k, v in last.items() if v is not None and v.dim() == 0},
    }
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of ~0.4 s (2000 steps of ~0.19 ms); 200 steps (40 ms) sat inside the clock ramp-up and
    # moved by 5 % from run to run (the other workloads: 40 steps of 2.5-10 ms, 5 warm-up)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["block", "classifier", "gan", "adaptpoint"], default="block",
                    help="block (default, the metric): one set-abstraction block fwd+bwd, BASELINE configs[1].  classifier / gan "
                         "/ adaptpoint: the training steps of configs[2] / [3] / [4] (per GPU B=32), wrapped for data "
                         "parallelism as the reference wraps them when launched through torch.distributed.run")
    ap.add_argument("--points", type=int, default=1024, help="points per cloud of the training-step workloads")
    ap.add_argument("--overlap", choices=["on", "off"], default="on",
                    help="training-step workloads: the joint step as two lanes of one captured graph (GanStep(overlap=True))")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed blocks of --steps steps each (median reported; 0 = auto: 25 when --steps < 400, else 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (value_f32_dropin at 1 GPU, value_no_syncbn at N>1)")
    ap.add_argument("--sync-bn", choices=["auto", "on", "off"], default="auto",
                    help="BatchNorm statistics at world_size>1.  auto = on: SyncBatchNorm, what the "
                         "reference forces (examples/classification/main.py:27): four small statistics "
                         "all-reduces per step inside the fused block + the gradient all-reduce.  off: "
                         "per-rank statistics (NOT the reference's semantics), one exchange per step")
    ap.add_argument("--deterministic", action="store_true",
                    help="bit-reproducible gradients (adaptpoint_amd.fused.DETERMINISTIC): the backward pass adds its "
                         "per-point sums as 64-bit fixed-point integers instead of float atomics; reported beside the "
                         "default as value_deterministic")
    ap.add_argument("--diag-freeze-index", action="store_true",
                    help="DIAGNOSTIC, not the metric: the index stages are computed once and the timed replays run the MLP "
                         "steps only (kernel tuning without the sampler beside them); the line says so in config.workload")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--distribution", choices=["D1", "D2"], default="D1",
                    help="synthetic clouds (SURVEY 8d): D1 uniform cube -> unit sphere; D2 sphere surface + jitter")
    ap.add_argument("--seed", type=int, default=0, help="base seed of the synthetic clouds (SURVEY 8d: 0..4)")
    ap.add_argument("--pipeline", choices=["on", "off"], default="on",
                    help="software-pipeline the index stage: FPS + ball query of the next launch's batches "
                         "run on a second HIP stream beside the MLP forward+backward of the current ones "
                         "(the index stage depends on coordinates only); fused path only")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step from a captured HIP graph (auto: on unless collectives sit inside the step)")
    ap.add_argument("--graph-collectives", choices=["auto", "on", "off"], default="auto",
                    help="N>1: capture the SyncBatchNorm statistics all-reduces and the gradient all-reduce into the "
                         "hipGraph (thread-local capture mode; auto = on) instead of running the step eagerly around "
                         "them (off: round 1's launch structure, ~2.4x slower at world_size 1)")
    ap.add_argument("--index-batch", type=int, default=0,
                    help="pipelined index stage: how many batches' FPS chains share one launch "
                         "(0 = all the batches of a graph replay; 1 = batch by batch as in round 1)")
    ap.add_argument("--index-overlap", choices=["on", "off"], default="off",
                    help="pipelined index stage batch by batch with FPS of one batch and ball query of the "
                         "previous one in ONE two-role launch (round 1's variant; superseded by --index-batch)")
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="whole steps captured per hipGraph (0 = auto: the largest of 20, 10, 4, 2 dividing --steps, "
                         "single GPU; warm-up is rounded up to whole replays; 1 = one step per replay)")
    ap.add_argument("--kernels", choices=["resident", "wide"], default="resident",
                    help="fused grouped MLP: the register-resident 32->32->64 kernels (csrc/sa_fused.hip) or the "
                         "width-generic ones with conv1 hoisted to the points (csrc/sa_wide.hip)")
    ap.add_argument("--mlp", choices=["fused-bf16x3", "fused-bf16", "torch-f32"], default="fused-bf16x3",
                    help="grouped shared-MLP: fused bf16-MFMA kernels (csrc/sa_fused.hip) with split "
                         "hi+lo operands (fp32-grade, default) or plain bf16 operands, or the unfused "
                         "drop-in path (nine extension ops + PyTorch conv/BN in fp32)")
    args = ap.parse_args()
    if args.workload == "block":
        args.steps = 2000 if args.steps is None else args.steps
        args.warmup = 200 if args.warmup is None else args.warmup

    from adaptpoint_amd import dp
    world, rank, local_rank = dp.env_world()
    # APN_BENCH_FORCE_DISTRIBUTED=1: take the N>1 code path (process group, SyncBatchNorm
    # phases, eager launch) even at world_size 1 -- lets a one-GPU box exercise it.
    force_dist = os.environ.get("APN_BENCH_FORCE_DISTRIBUTED") == "1"
    distributed = world > 1 or force_dist
    if distributed and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not distributed and args.gpus != 1:
        raise SystemExit("launch N>1 through torch.distributed.run (one process per GPU)")

    from adaptpoint_amd import _lib
    _lib.load()                                   # the HIP extension or nothing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if os.environ.get("APN_BENCH_MAIN_PRIORITY"):            # (experiment: the MLP stream as a high-priority stream)
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ["APN_BENCH_MAIN_PRIORITY"])))
    dev = torch.device("cuda", local_rank)
    if distributed and args.graph_collectives != "off":
        # collectives inside hipGraphs: the watchdog must not poll events of in-flight work (PyTorch's
        # documented requirement for capturing NCCL work), set before the process group exists
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    dp.init(args.backend, dev, force=force_dist)
    if force_dist:
        from adaptpoint_amd import fused as _fused
        _fused.FORCE_PHASED = True
        dp.FORCE_COLLECTIVES = True
    if args.workload != "block":
        return workload_main(args, dev, world, rank, distributed)

    if args.kernels == "wide":
        from adaptpoint_amd import set_abstraction as _sa_mod
        _sa_mod.PREFER_WIDE = True
    sync_bn = distributed and args.sync_bn != "off"          # auto = on, as the reference (main.py:27)
    # fewer than ~400 steps are less than 0.1 s of work: repeat the K-step block and report the median (keeps `steps` = K)
    repeats = args.repeats if args.repeats > 0 else (25 if args.steps < 400 else 1)
    # N>1: the SAFE launch structure is measured FIRST (the step run eagerly around its collectives: nothing of RCCL
    # inside a hipGraph), the captured-collectives structure after it: a capture that fails falls back to the former,
    # and the line carries both figures.  (No multi-rank run of either exists: one-GPU boxes only.)
    value_eager_collectives = None
    if distributed and sync_bn and args.graph_collectives != "off" and not args.no_secondary:
        import copy as _copy
        safe_args = _copy.copy(args)
        safe_args.graph_collectives = "off"
        sec = max(4, min(args.steps, 40))
        ms = measure(safe_args, dev, world, rank, local_rank, distributed, args.mlp, sync_bn, sec, 4)
        value_eager_collectives = round(B_PER_GPU * world * sec / ms.elapsed, 2)
    m = measure(args, dev, world, rank, local_rank, distributed, args.mlp, sync_bn, args.steps, args.warmup, repeats)
    elapsed, spg, use_graph, pipelined, fused_mlp, eager_step = (m.elapsed, m.spg, m.use_graph, m.pipelined,
                                                                 m.fused_mlp, m.eager_step)
    # The timed step issues whole launch sequences (one C call per direction, or one hipGraph),
    # which cannot carry per-kernel events.  The dominant kernel is therefore timed right after
    # the timed region: the same launch, same inputs, same stream, HIP events around it.
    t2 = KernelTimer()
    r2 = instrument(t2, only={"fps"})
    for _ in range(max(2, 20 // spg)):
        eager_step()
    r2()
    fps_us = t2.mean_us()["fps"]
    fps_clouds = B_PER_GPU * m.index_batch            # clouds (= workgroups = serial chains) per sampler launch

    # per-kernel view (un-timed extra pass): events around every extension launch
    # (every pass is enqueued behind a ~1 ms spin on the launch stream: the GPU then runs the launches back to back, as in
    # the replayed graph -- issued onto an idle queue each event pair also measured the host's launch latency, and the
    # dominant kernel read 45 us against rocprof's 39.8 of the replay)
    timer_all = KernelTimer()
    restore = instrument(timer_all)
    for _ in range(max(12, min(args.steps, 20) // spg)):
        torch.cuda._sleep(2_000_000)
        eager_step()
        torch.cuda.synchronize()
    per_kernel_stats = timer_all.stats_us()
    per_kernel_us = {k: v[0] for k, v in per_kernel_stats.items()}
    restore()

    total_clouds = B_PER_GPU * world * args.steps
    value = total_clouds / elapsed
    ab = algorithmic_bytes(B_PER_GPU, fused=fused_mlp)
    for k in ("fps", "ball_query", "sa_point_geo"):   # index launches cover index_batch batches
        ab[k] *= m.index_batch
    kernels = {}
    for k, us in sorted(per_kernel_us.items()):
        ent = {"avg_us": round(us, 2), "p25_us": round(per_kernel_stats[k][1], 2)}
        if k in ab:
            ent["algorithmic_bytes"] = ab[k]
            ent["achieved_GBps"] = round(ab[k] / us * 1e-3, 2)
            ent["frac_hbm"] = round(ab[k] / us * 1e-3 / HBM_PEAK_GBS, 5)
        kernels[k] = ent
    # PMC-measured HBM bytes per launch (rocprofv3 --pmc passes over THIS command's default launch structure,
    # corrected as MI355X_MICROARCH.md prescribes; scripts/collect_profiles.sh pmc -> scripts/make_traffic_json.py).
    # A committed constant, labelled as such: bench.py cannot run the profiler on itself.
    traffic, traffic_src, traffic_stamp = {}, None, {}
    for name in (TRAFFIC_FILE,):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("structure", "default") == "default":
                traffic, traffic_src = tj.get("bytes_per_launch", {}), "profiles/" + name
                # which tree the constant belongs to: the commit it was collected at, and whether the kernel sources it was
                # collected from are THIS tree's (scripts/make_traffic_json.py hashes them)
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                try:
                    import make_traffic_json as _mt
                    traffic_stamp = {"collected_at_commit": tj.get("collected_at_commit"),
                                     "sources_match": tj.get("kernel_sources_sha16") == _mt.sources_sha16(ROOT)
                                     if tj.get("kernel_sources_sha16") else None}
                finally:
                    sys.path.pop(0)
            break
    for k, ent in kernels.items():
        if k in traffic:
            ent["traffic_bytes_pmc"] = traffic[k]
    step_s = elapsed / args.steps
    step_flops, step_bytes = STEP_FLOPS_PER_CLOUD * B_PER_GPU, STEP_BYTES_PER_CLOUD * B_PER_GPU
    # The dominant kernel = the largest TOTAL time per step on the stream that bounds `value`.  With the index
    # stage on its own stream (one sampler launch per `index_batch` steps) that is the MLP stream; its kernels
    # launch once per step, so total = average launch duration.  Without the pipeline everything is one stream and
    # the index kernels count with their per-step share.
    index_names = {"fps", "ball_query", "fps+ball_query", "sa_point_geo", "sa_wide_tilemap", "sa_wide_tilemap_many", "sa_wide_csr",
                   "sa_rowmap_many"}
    per_step_us = {k: (us / m.index_batch if k in index_names else us) for k, us in per_kernel_us.items()}
    cand = {k: v for k, v in per_step_us.items() if not (pipelined and k in index_names)} or per_step_us
    dominant = max(cand, key=cand.get)
    # the dominant kernel again, ten launches of the same arguments between ONE pair of events (KernelTimer.burst): reported
    # beside the figure the roofline uses, not instead of it -- the repeats find their inputs in the caches, which a step's
    # launch does not (27.7 us against 31-32)
    if dominant in ("sa_bwd_main", "sa_fwd_main", "sa_wide_bwd_main", "sa_wide_fwd_main"):
        timer_b = KernelTimer(burst=dominant)
        restore = instrument(timer_b)
        for _ in range(6):
            torch.cuda._sleep(2_000_000)
            eager_step()
            torch.cuda.synchronize()
        restore()
        kernels[dominant]["avg_us_ten_back_to_back_same_inputs"] = round(timer_b.burst_us(), 2)
    dom_us = per_kernel_us[dominant]
    split = 3 if args.mlp.endswith("x3") else 1
    # SURVEY 8d per-cloud MLP flops, by pass: conv1 2*35*32 and conv2 2*32*64 per position forward; the backward pass
    # re-runs conv1 in both orientations 2*(2*35*32), dL/da1 = a1*Qm (2*32*32) + sparse*W2^T (2*64*32) and the Gram
    # product a1^T a1 (2*32*32) per position (after the Gram reformulation of dL/dW2: DESIGN.md section 4).
    pos = B_PER_GPU * NPOINT * NSAMPLE
    mfma_flops = {"sa_fwd_main": pos * (2 * 35 * 32 + 2 * 32 * 64),
                  "sa_bwd_main": pos * (2 * 2 * 35 * 32 + 2 * 2 * 32 * 32 + 2 * 64 * 32),
                  # the width-generic family: conv1 hoisted to the points, y2 = a1 W2^T forward; dL/da1 over k = [C_out; C_mid]
                  # and the Gram / pooled-row products backward
                  "sa_wide_fwd_main": pos * (2 * 32 * 64),
                  "sa_wide_bwd_main": pos * (2 * (64 + 32) * 32 + 2 * 32 * 32)}
    if dominant in mfma_flops:
        tf = mfma_flops[dominant] / dom_us * 1e-6
        roofline = {"kernel": dominant, "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 5),
                    "algorithmic_flops_per_launch": mfma_flops[dominant],
                    "algorithmic_bytes_per_launch": ab.get(dominant),
                    "note": (f"algorithmic flops of the 32 clouds of one launch (every product is issued as {split} bf16 "
                             f"MFMA(s): MFMA issue = {split} x this fraction) over the kernel's average launch duration, "
                             "HIP events on its launch stream in an eager pass of the same launches right after the "
                             "timed region (median of twelve pairs; `p25_launch_us` = their lower quartile); the kernel is bound by per-tile latency (two waves per SIMD at 256 VGPRs), "
                             "not by MFMA issue or HBM (DESIGN.md section 5)")}
    else:
        gb = ab.get(dominant, 0) / dom_us * 1e-3
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": round(gb, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gb / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": ab.get(dominant)}
    roofline.update({
        "avg_launch_us": round(dom_us, 2), "p25_launch_us": round(per_kernel_stats[dominant][1], 2),
        "traffic": traffic.get(dominant), "traffic_source": traffic_src if dominant in traffic else None,
        "traffic_collected_at_commit": traffic_stamp.get("collected_at_commit"),
        "traffic_sources_match": traffic_stamp.get("sources_match"),
        # the whole step (one batch of 32 clouds through FPS, ball query, fused forward and backward)
        # against the two peaks: SURVEY 8d's per-cloud figures x 32 over the measured step time
        "step": {"flops": step_flops, "bytes": step_bytes,
                 "frac_mfma": round(step_flops / step_s / (MFMA_BF16_PEAK_TFLOPS * 1e12), 5),
                 "frac_hbm": round(step_bytes / step_s / (HBM_PEAK_GBS * 1e9), 5)},
        # FPS is a serial chain of npoint-1 dependent arg-max steps, one workgroup per cloud: bound by per-step
        # latency, neither HBM nor MFMA; it runs on the index stream
        "index_stream": {"fps_step_ns": round(fps_us * 1e3 / (NPOINT - 1), 1), "fps_clouds_per_launch": fps_clouds,
                         "fps_us_per_batch": round(fps_us / m.index_batch, 2),
                         # both index sets after the timed region == the index stage run alone (bit for bit)
                         "verified_after_timed_region": m.index_verified,
                         # the last replayed step's parameter gradients (formed beside the index kernels) against the same
                         # step launched alone
                         "mlp_stream_verified_after_timed_region": m.mlp_verified},
        "kernels": kernels,
        # the kernels this run launched == the set the committed traffic file was collected for
        "kernel_set_is_default": sorted(kernels) == sorted(DEFAULT_KERNELS),
        "traffic_file_covers_kernel_set": bool(traffic) and sorted(traffic) == sorted(kernels),
    })

    from adaptpoint_amd import set_abstraction as _sa
    result = {
        "metric": "set-abstraction fwd+bwd point-clouds/sec (B=32,N=1024)",
        "value": round(value, 2), "unit": "point-clouds/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "timed_blocks": len(m.blocks),       # blocks of `steps` steps, each fenced on both sides; median reported
        "ms_per_step_min": round(1e3 * min(m.blocks) / args.steps, 4),
        "ms_per_step_max": round(1e3 * max(m.blocks) / args.steps, 4),
        "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if fused_mlp else "f32", "data": "synthetic",
        "config": {"workload": ("DIAGNOSTIC (--diag-freeze-index: index stages NOT in the timed region; not the metric): "
                                if args.diag_freeze_index else "")
                               + "PointNeXt-S stage-1 SetAbstraction fwd+bwd, B=32/GPU N=1024 "
                                 "npoint=512 nsample=32 r=0.15 C 32->64 (BASELINE configs[1])",
                   "distribution": {"D1": "D1: uniform cube centred+scaled to the unit sphere",
                                    "D2": "D2: unit-sphere surface + N(0,0.01) jitter"}[args.distribution],
                   "seed": args.seed,
                   "mlp": ({"fused-bf16x3": "fused bf16 MFMA on split hi+lo operands (3 MFMAs per product, "
                                            "f32 accumulate; forward within 7e-5 of an fp32 chain), "
                                            "f32 BatchNorm partial sums added exactly (integer accumulators) or in f64",
                            "fused-bf16": "fused bf16 MFMA (operands rounded to bf16, f32 accumulate), "
                                          "f32 BatchNorm partial sums added exactly (integer accumulators) or in f64",
                            "torch-f32": "unfused: extension ops + PyTorch fp32 conv/BN"}[args.mlp]),
                   "kernels": args.kernels,
                   "tile_map": ("both passes over the index stage's distinct-hit tile map (ball-query fill copies "
                                "folded into one row with a multiplicity: 3.7x fewer MFMA tiles); the backward pass stores its "
                                "rows of g_u through the map's row map (point-sorted order) and the per-point kernel sums a "
                                "point's consecutive rows in ascending order: no float atomics, gradients bit-reproducible"
                                if (fused_mlp and args.kernels == "resident") else
                                ("all passes over the distinct-hit tile map; per-point sums through its inverse map "
                                 "(no float atomics: gradients bit-reproducible)" if fused_mlp else "none")),
                   "launch": (f"hipGraph replay, {spg} step(s) per graph" if use_graph else "eager"),
                   "graph_nodes": getattr(m, "graph_nodes", None),     # node census of the captured graphs (no memset nodes)
                   "launches_per_step": ("3 forward + 4 backward on the MLP stream (BatchNorm folds and per-channel "
                                         "constants are prologues of their consumer kernels)" if fused_mlp and args.kernels == "resident" else None),
                   "pipeline": (f"index stages (FPS + ball query + occurrence statistics + tile map + row map) of the NEXT launch's batches on a second stream, "
                                f"{m.index_batch} batch(es) per sampler launch, beside the MLP fwd+bwd of the "
                                "current batch(es); the two streams meet once per launch" if pipelined else "none"),
                   "global_batch": B_PER_GPU * world,
                   "fused_fallbacks": sum(_sa.FUSED_FALLBACKS.values()),
                   "deterministic_gradients": bool(fused_mlp),       # no float atomics on the fused chain (round 5)
                   "parallelism": f"dp{world}" + ("+syncbn" if sync_bn else "")
                                  + ("+flat-allreduce" if distributed else "")
                                  + ("+collectives-in-graph" if getattr(m, "capture_collectives", False) else "")},
        "roofline": roofline,
    }
    if value_eager_collectives is not None:
        result["value_eager_collectives"] = value_eager_collectives
    if not args.no_secondary:
        sec_steps = max(4, min(args.steps, 40))
        if distributed and sync_bn:
            # the cheaper step the reference never runs: per-rank BatchNorm statistics, graph replay + one all-reduce
            m2 = measure(args, dev, world, rank, local_rank, distributed, args.mlp, False, sec_steps, 4)
            result["value_no_syncbn"] = round(B_PER_GPU * world * sec_steps / m2.elapsed, 2)
        elif not distributed and fused_mlp:
            # the bit-for-bit interoperable path: nine drop-in operators + PyTorch fp32 conv/BN (eager)
            m2 = measure(args, dev, world, rank, local_rank, distributed, "torch-f32", False, sec_steps, 4)
            result["value_f32_dropin"] = round(B_PER_GPU * sec_steps / m2.elapsed, 2)
            if pipelined and args.index_batch == 0 and args.steps % 20 == 0:
                # What the headline assumes, stated beside it: `value` runs the index stages (FPS + ball query) of 20 batches
                # in ONE stacked launch on a second stream, a replay ahead of the MLP steps that consume them.  The same
                # run batch by batch (one index launch per step, still a replay ahead), and with no second stream at all
                # (index stage in line, in front of each step's MLP -- what a consumer of freshly GENERATED coordinates,
                # e.g. the AdaptPoint feedback pass, gets):
                import copy as _copy
                v_steps = min(args.steps, 400)
                a1 = _copy.copy(args)
                a1.index_batch = 1
                m4 = measure(a1, dev, world, rank, local_rank, distributed, args.mlp, False, v_steps, 40, repeats=5)
                result["value_index_batch_1"] = round(B_PER_GPU * v_steps / m4.elapsed, 2)
                result["index_batch_1_verified"] = m4.index_verified
                a2 = _copy.copy(args)
                a2.pipeline = "off"
                m5 = measure(a2, dev, world, rank, local_rank, distributed, args.mlp, False, v_steps, 40, repeats=5)
                result["value_no_pipeline"] = round(B_PER_GPU * v_steps / m5.elapsed, 2)
            if args.mlp == "fused-bf16x3" and fused_mlp:
                # bit-reproducible gradients are the only mode since round 5 (rounds 3-4: a separate, slower mode with 64-bit
                # fixed-point atomics, reported here): the figure IS `value`
                result["value_deterministic"] = result["value"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
