"""BASELINE configs[3]: the AdaptPoint joint step -- generator (Deformation + Mask controllers)
+ discriminator + feedback through the eval-mode PointNeXt-S classifier -- one `train_gan`
iteration (examples/classification/train_autoaug.py:133-204), B=32, one MI355X.

    python scripts/bench_gan_step.py [--points 1024|2048] [--batch 32] [--mode fused|composed|both]

N=1024 is what BASELINE.json states; N=2048 is what the reference trains at (SURVEY header).
`fused`: the fused operators everywhere they exist (grouper, attention, set-abstraction blocks, one
2B feedback pass); `composed`: the same mirrors grouping / attending / convolving the way the
reference composes them in PyTorch over the nine drop-in operators, two feedback passes.
Also times the classifier training step (`train_one_epoch`, :471-512) with its resampler.
Prints one JSON line per configuration.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import golden_inputs as GI
from adaptpoint_amd.augmentor import AdaptPointAugmentor
from adaptpoint_amd.discriminator import PointDiscriminator1
from adaptpoint_amd.gan import ClassifierStep, GanStep
from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy


def timed(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--mode", default="both", choices=["fused", "composed", "both"])
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole joint step from a hipGraph (random draws of the generator then come "
                         "from the device generator, capturable Adam)")
    ap.add_argument("--overlap", action="store_true",
                    help="GanStep(overlap=True): the feedback pass's index pyramid and the discriminator's own step on "
                         "side streams (parallel branches of the captured graph)")
    ap.add_argument("--overlap-parts", default="",
                    help="comma-separated subset of imitator,real (default: both)")
    ap.add_argument("--stamps", action="store_true",
                    help="capture device wall-clock stamps at the step's phase boundaries (adaptpoint_amd.graphs."
                         "PhaseStamps) and print them after the run")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pos = torch.from_numpy(GI.unit_sphere_cloud(a.batch, a.points, seed=0))
    points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).to(dev)
    label = (torch.arange(a.batch) % 15).to(dev)
    for mode in (("fused", "composed") if a.mode == "both" else (a.mode,)):
        fused = mode == "fused"
        torch.manual_seed(0)
        G = AdaptPointAugmentor(fused=fused).to(dev)
        D = PointDiscriminator1(num_classes=15, fused=fused).to(dev)
        C = PointNextSClassifier(fused=fused).to(dev)
        step = GanStep(G, D, C, SmoothCrossEntropy(0.3), batched_feedback=fused, capturable=a.graph,
                       overlap=(frozenset(a.overlap_parts.split(",")) if a.overlap_parts else True) if (a.overlap and fused) else False)
        run = lambda: step(points, label)
        if a.graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step(points, label, device_noise=True)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            from adaptpoint_amd import graphs as apn_graphs
            if a.stamps:
                apn_graphs.STAMPS = apn_graphs.PhaseStamps(dev)
            # (no memset nodes, no autograd graph of an earlier step alive: adaptpoint_amd/graphs.py)
            graph, captured, _ = apn_graphs.capture(lambda: step(points, label, device_noise=True), what="the joint step's graph",
                                                    leaves=[q for net in (G, D, C) for q in net.parameters()])
            run = graph.replay
        torch.cuda.reset_peak_memory_stats()
        sec = timed(run, a.iters, a.warmup)
        res = {"config": "train_gan step (BASELINE configs[3])", "mode": mode, "B": a.batch, "N": a.points,
               "ms_per_step": round(sec * 1e3, 3), "clouds_per_s": round(a.batch / sec, 1),
               "peak_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
               "launch": "hipGraph replay" if a.graph else "eager", "overlap": bool(a.overlap and fused)}
        if a.graph:
            # host time of a replay call alone (hipGraphLaunch enqueues the nodes one by one on this stack: a step
            # of ~900 nodes can be bound by the enqueue, not by the device)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run()
            res["host_ms_per_replay_call"] = round((time.perf_counter() - t0) * 1e3, 3)
            torch.cuda.synchronize()
        out = captured if a.graph else step(points, label)
        res["losses"] = {k: round(float(out[k]), 5) for k in ("g_loss_raw", "feedback_loss", "d_loss")}
        print(json.dumps(res), flush=True)
        if a.graph and a.stamps:
            prev = 0.0
            for name, us in apn_graphs.STAMPS.report():
                print(f"  {us:9.1f} us  (+{us - prev:8.1f})  {name}")
                prev = us
            apn_graphs.STAMPS = None
        if a.points > 1024 and not a.graph:
            cstep = ClassifierStep(C)
            sec = timed(lambda: cstep(points, label), a.iters, a.warmup)
            print(json.dumps({"config": "train_one_epoch step with resampler (BASELINE configs[2] as trained)",
                              "mode": mode, "B": a.batch, "N": a.points, "ms_per_step": round(sec * 1e3, 3),
                              "clouds_per_s": round(a.batch / sec, 1), "launch": "eager"}), flush=True)


if __name__ == "__main__":
    main()
