"""Out-of-bounds WRITES of the extension's kernels: every tensor allocated through torch.empty / empty_like / zeros / full
inside the region gets a canary pad on both sides; the pads are checked afterwards.  (A write a few elements beyond a
buffer lands in a dead neighbour when kernels run one after the other -- and in live data of the OTHER lane when two
lanes of a graph share one allocator pool.)
    python scripts/redzone_run.py [classifier-eval | classifier-train | generator | discriminator]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import traceback
import torch
import golden_inputs as GI

PAD = 8192          # bytes on each side
CANARY = 0x5A
_real_empty, _real_empty_like, _real_zeros, _real_full = torch.empty, torch.empty_like, torch.zeros, torch.full
guards = []


def _guarded(shape, dtype, device, fill=None):
    dtype = dtype or torch.float32
    n = 1
    for d in shape:
        n *= int(d)
    es = torch.empty(0, dtype=dtype).element_size()
    raw = _real_empty(2 * PAD + n * es + 64, dtype=torch.uint8, device=device)
    off = (-raw.data_ptr()) % 64                      # keep the interior 64-byte aligned (+PAD is a multiple of 64)
    raw[:off + PAD] = CANARY
    raw[off + PAD + n * es:] = CANARY
    t = raw[off + PAD: off + PAD + n * es].view(dtype).view(*shape)
    if fill is not None:
        t.fill_(fill)
    where = "".join(traceback.format_stack(limit=6)[:-2][-3:])
    guards.append((raw, off, n * es, tuple(shape), dtype, where))
    return t


def _shape_of(args):
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return tuple(args[0])
    return tuple(args)


def empty(*args, dtype=None, device=None, **kw):
    dev = torch.device(device) if device is not None else None
    if dev is None or dev.type != "cuda" or kw.get("pin_memory"):
        return _real_empty(*args, dtype=dtype, device=device, **kw)
    return _guarded(_shape_of(args), dtype, dev)


def empty_like(t, dtype=None, device=None, **kw):
    dev = torch.device(device) if device is not None else t.device
    if dev.type != "cuda" or not t.is_contiguous():
        return _real_empty_like(t, dtype=dtype, device=device, **kw)
    return _guarded(tuple(t.shape), dtype or t.dtype, dev)


def zeros(*args, dtype=None, device=None, **kw):
    dev = torch.device(device) if device is not None else None
    if dev is None or dev.type != "cuda":
        return _real_zeros(*args, dtype=dtype, device=device, **kw)
    return _guarded(_shape_of(args), dtype, dev, fill=0)


def full(size, value, dtype=None, device=None, **kw):
    dev = torch.device(device) if device is not None else None
    if dev is None or dev.type != "cuda":
        return _real_full(size, value, dtype=dtype, device=device, **kw)
    if dtype is None:
        dtype = torch.float32 if isinstance(value, float) else torch.int64 if isinstance(value, int) else torch.bool
    return _guarded(tuple(size), dtype, dev, fill=value)


class redzones:
    def __enter__(self):
        torch.empty, torch.empty_like, torch.zeros, torch.full = empty, empty_like, zeros, full
        return self

    def __exit__(self, *exc):
        torch.empty, torch.empty_like, torch.zeros, torch.full = _real_empty, _real_empty_like, _real_zeros, _real_full


def check(tag):
    torch.cuda.synchronize()
    bad = 0
    for raw, off, nbytes, shape, dtype, where in guards:
        lo = raw[:off + PAD]
        hi = raw[off + PAD + nbytes:]
        nlo = int((lo != CANARY).sum())
        nhi = int((hi != CANARY).sum())
        if nlo or nhi:
            bad += 1
            first_hi = int((hi != CANARY).nonzero()[0]) if nhi else -1
            last_lo = int((lo != CANARY).nonzero()[-1]) - (off + PAD) if nlo else 0
            print(f"[{tag}] OUT-OF-BOUNDS WRITE around a {dtype} tensor of shape {shape}: {nlo} bytes below (nearest {last_lo}), "
                  f"{nhi} bytes above (first at +{first_hi}); allocated at\n{where}")
    print(f"[{tag}] {len(guards)} guarded tensors, {bad} with damaged pads")
    guards.clear()
    return bad


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "classifier-eval"
    dev = torch.device("cuda:0")
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    pos = torch.from_numpy(GI.unit_sphere_cloud(B, 1024, seed=700)).to(dev)
    points = torch.cat([pos, pos[:, :, 1:2] - pos[:, :, 1:2].min(1, keepdim=True)[0]], -1)
    label = (torch.arange(B) % 15).to(dev)
    from adaptpoint_amd.pointnext import PointNextSClassifier, SmoothCrossEntropy, fill_parameters_by_name
    if what.startswith("classifier"):
        C = fill_parameters_by_name(PointNextSClassifier(fused=True)).to(dev)
        data = {'pos': pos, 'x': points.transpose(1, 2).contiguous()}
        if what == "classifier-eval":
            C.eval()
            with torch.no_grad():
                C(data)
                torch.cuda.synchronize()
                with redzones():
                    C(data)
            return check(what)
        C.train()
        logits, loss = C.get_logits_loss(data, label)
        loss.backward()
        torch.cuda.synchronize()
        with redzones():
            logits, loss = C.get_logits_loss(data, label)
            loss.backward()
        return check(what)
    if what == "generator":
        from adaptpoint_amd.augmentor import AdaptPointAugmentor, draw_noise_on
        G = fill_parameters_by_name(AdaptPointAugmentor(fused=True)).to(dev).train()
        noise = draw_noise_on(dev, B, 1024, G.num_anchor)
        G(pos, noise)[1].sum().backward()
        torch.cuda.synchronize()
        with redzones():
            G(pos, noise)[1].sum().backward()
        return check(what)
    if what == "discriminator":
        from adaptpoint_amd.discriminator import PointDiscriminator1
        D = fill_parameters_by_name(PointDiscriminator1(num_classes=15, fused=True)).to(dev).train()
        x = pos.clone().requires_grad_(True)
        D(x).sum().backward()
        torch.cuda.synchronize()
        with redzones():
            D(x).sum().backward()
        return check(what)
    raise SystemExit("unknown target")


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
