"""Which piece of the classifier step breaks hipGraph capture: each piece captured in its own process."""
import subprocess
import sys

PIECES = ["full", "full+eager", "full+eager+load", "full+snap", "full+eager+snap"]

if len(sys.argv) > 1:
    import os
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from adaptpoint_amd import synthetic as GI
    from adaptpoint_amd.gan import ClassifierStep, resample
    from adaptpoint_amd.pointnext import PointNextSClassifier, fill_parameters_by_name
    what = sys.argv[1]
    dev = torch.device("cuda:0")
    B, N = 4, (1024 if what == "full_no_resample" else 2048)
    C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev)
    for m in C.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    opt = torch.optim.AdamW(C.parameters(), lr=2e-3, weight_decay=0.05, capturable=True, fused=True)
    step = ClassifierStep(C, optimizer=opt)
    choice = torch.from_numpy(np.random.RandomState(3).choice(1200, 1024, False).astype(np.int32)).to(dev)
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, N, seed=1)).to(dev)
    points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1)
    target = torch.randint(0, 15, (B,), device=dev)

    def piece():
        if what == "resample":
            return resample(points, 1024, 4, choice)
        if what.startswith("full"):
            return step(points, target, choice=choice)[1]
        C.train()
        p, x = resample(points, 1024, 4, choice)
        logits, loss = C.get_logits_loss({'pos': p, 'x': x}, target)
        if what == "forward":
            return loss
        loss.backward()
        if what == "fwd_bwd_clip":
            torch.nn.utils.clip_grad_norm_(C.parameters(), 10.0, norm_type=2)
        return loss
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            piece()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    import copy
    if "snap" in what:
        snap = copy.deepcopy(C.state_dict())
        ost = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state[p].items()} for p in opt.state}
    if "eager" in what:
        for _ in range(3):
            if "load" in what:
                points.copy_(points.clone())
            piece()
        w = copy.deepcopy(C.state_dict())
    if "snap" in what:
        with torch.no_grad():
            own = C.state_dict()
            for k, v in snap.items():
                own[k].copy_(v)
            for p in opt.state:
                for k, v in ost[id(p)].items():
                    if torch.is_tensor(v):
                        opt.state[p][k].copy_(v)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        piece()
    g.replay()
    torch.cuda.synchronize()
    print("captured + replayed:", what)
    sys.exit(0)

for w in PIECES:
    r = subprocess.run([sys.executable, __file__, w], capture_output=True, text=True, timeout=300)
    print(w, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], "|", (r.stderr.strip().splitlines() or [""])[-1][:200])
