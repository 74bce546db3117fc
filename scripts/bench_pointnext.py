"""BASELINE configs[2]: full PointNeXt-S classifier training step (fwd + bwd + AdamW) on
random 15-class clouds, B=32, N=1024, one MI355X.  Not the headline metric (bench.py is);
a second data point for DESIGN.md.

    python scripts/bench_pointnext.py [--fused] [--steps 50]
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
from adaptpoint_amd.pointnext import PointNextSClassifier


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", action="store_true", help="fused kernels in all four grouped stages")
    ap.add_argument("--wide-first", action="store_true", help="stage 1 on the width-generic kernels too")
    ap.add_argument("--graph", action="store_true", help="replay the whole step from a hipGraph")
    ap.add_argument("--pipeline", action="store_true",
                    help="index pyramid (FPS + ball query of all four blocks: coordinates only) of the NEXT batch on a "
                         "second stream beside the current step; needs --graph")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if a.wide_first:
        from adaptpoint_amd import set_abstraction as SA
        SA.PREFER_WIDE = True
    model = PointNextSClassifier(fused=a.fused).to(dev).train()
    # cfgs/scanobjectnn/default.yaml; under capture PyTorch's fused multi-tensor AdamW: the capturable foreach form
    # falls back to two `div_` launches per parameter (~110 launches of 4 us, 0.47 ms of the step), same update rule
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05,
                            **(dict(capturable=True, fused=True) if a.graph else {}))
    pos = torch.from_numpy(GI.unit_sphere_cloud(a.batch, 1024, seed=0)).to(dev)
    x = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1).transpose(1, 2).contiguous()
    gt = torch.randint(0, 15, (a.batch,), device=dev, generator=torch.Generator(dev).manual_seed(0))

    data = {'pos': pos, 'x': x}
    pyr = [None, None]
    if a.pipeline:
        pyr = [model.encoder.index_pyramid(pos) for _ in range(2)]

    def step(cur=0):
        opt.zero_grad(set_to_none=True)
        logits, loss = model.get_logits_loss(data, gt, pyramid=pyr[cur])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10, norm_type=2)     # train_autoaug.py:505-508
        opt.step()
        return loss.detach()

    if a.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs, losses, igraphs = {}, {}, {}
        for cur in ((0, 1) if a.pipeline else (0,)):
            from adaptpoint_amd import graphs as apn_graphs
            opt.zero_grad(set_to_none=True)
            # (no memset nodes, no autograd graph of an earlier step alive: adaptpoint_amd/graphs.py)
            g, losses[cur], _ = apn_graphs.capture(lambda: step(cur).detach(), leaves=list(model.parameters()),
                                                   what="the classifier step's graph")
            graphs[cur] = g
            if a.pipeline:
                igraphs[cur], _, _ = apn_graphs.capture(lambda: model.encoder.index_pyramid(pos, out=pyr[cur]),
                                                        what="the index pyramid's graph")
        state = {"cur": 0}
        index_stream = torch.cuda.Stream()
        main_done, index_done = torch.cuda.Event(), torch.cuda.Event()
        main_done.record(); index_done.record()
        box = [index_done]

        def step():
            cur = state["cur"]
            main = torch.cuda.current_stream()
            if a.pipeline:
                # the side stream refills the OTHER set (read by the previous step) while this step consumes `cur`
                index_stream.wait_event(main_done)
                with torch.cuda.stream(index_stream):
                    igraphs[1 - cur].replay()
                    nxt = torch.cuda.Event()
                    nxt.record(index_stream)
                main.wait_event(box[0])
                box[0] = nxt
            graphs[cur].replay()
            if a.pipeline:
                main_done.record(main)
                state["cur"] = 1 - cur
            return losses[cur]
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    verified = None
    if a.pipeline and a.graph:
        # the pyramids were filled BESIDE the feature stream all along: both sets against a pyramid built on the idle device
        ref = model.encoder.index_pyramid(pos)
        torch.cuda.synchronize()
        verified = all(torch.equal(x.buf, y.buf) for st in pyr for x, y in zip(st, ref) if x is not None)
    print(json.dumps({"config": "PointNeXt-S classifier train step, B=%d N=1024" % a.batch,
                      "stages": ("fused (1: %s kernels; 2-4: width-generic kernels)" % ("width-generic" if a.wide_first else "register-resident")
                                 if a.fused else "unfused ops + PyTorch fp32"),
                      "launch": ("hipGraph replay" + (", index pyramid of the next batch on a second stream" if a.pipeline else "")
                                 if a.graph else "eager"),
                      "ms_per_step": round(1e3 * el / a.steps, 3),
                      "clouds_per_s": round(a.batch * a.steps / el, 1), "loss": float(loss),
                      **({"index_pyramids_verified_after_the_run": bool(verified)} if verified is not None else {})}))


if __name__ == "__main__":
    main()
