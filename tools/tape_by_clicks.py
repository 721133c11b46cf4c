#!/usr/bin/env python3
"""Training-mode decoder (DecoderTape) forward / backward time on 4 x 80 k voxels for given click counts per sample, and the
tape's primitive ops by device time (each call bracketed by a device synchronisation, so the sum is stream-serial):
  python tools/tape_by_clicks.py 96,48,80,48 26,104,13,91 ..."""
import collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agile3d_amd import build_model, default_args
from agile3d_amd.synthetic import make_scene
from agile3d_amd import train_decoder as TD
from agile3d_amd.train_decoder import DecoderTape

if os.environ.get("SPIN") == "1":      # host waits spin instead of sleeping on the completion interrupt
    import ctypes
    _hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", _hip.hipSetDeviceFlags(1))
torch.manual_seed(0)
REPEATS = int(os.environ.get("REPEATS", "8"))
if os.environ.get("NOGC") == "1":
    import gc
    gc.disable()
model = build_model(default_args()).cuda().train()
scenes = [make_scene(80_000, seed=s) for s in range(4)]
ns = [len(sc["coords"]) for sc in scenes]
pcd = [torch.randn(n, 128, device="cuda") * 0.3 for n in ns]
pos = [torch.randn(n, 128, device="cuda") * 0.3 for n in ns]


def clicks_for(sc, total, seed):
    """`total` clicks spread over up to 10 objects (+ none on the background), the shape train_one_step ends a round with"""
    rng = np.random.default_rng(seed)
    lab = sc["labels"]
    ids = [i for i in np.unique(lab) if i > 0][:min(10, total)]
    ci, ct = {"0": []}, {"0": []}
    for k in range(1, len(ids) + 1):
        ci[str(k)], ct[str(k)] = [], []
    for t in range(total):
        k = 1 + t % len(ids)
        rows = np.flatnonzero(lab == ids[k - 1])
        ci[str(k)].append(int(rng.choice(rows)))
        ct[str(k)].append(t)
    return ci, ct


def run(counts, by_op=False):
    cs = [clicks_for(sc, c, 7 + i) for i, (sc, c) in enumerate(zip(scenes, counts))]
    ci, ct = [c[0] for c in cs], [c[1] for c in cs]
    R = None
    def once():
        nonlocal R
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tp = DecoderTape(model, pcd, pos, ci, ct)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if R is None:
            R = [[torch.randn_like(x) / 8 for x in lvl] for lvl in tp.logits]
        tp.backward(R)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tp.release()
        return 1e3 * (t1 - t0), 1e3 * (t2 - t1)
    for _ in range(2):
        once()
    f, b = zip(*[once() for _ in range(3)])
    print(f"clicks {counts}: forward {min(f):.1f} ms, backward {min(b):.1f} ms")
    if by_op:
        acc = collections.defaultdict(lambda: [0, 0.0])
        slow = []
        stalls = []
        lib = TD.L.load()
        names = list(TD.L.SYMBOLS)
        orig = {}
        for n in names:
            fn = getattr(lib, n)
            orig[n] = fn
            def wrap(*a, _fn=fn, _n=n):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t = time.perf_counter()
                e0.record()
                r = _fn(*a)
                e1.record()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
                e = acc[_n]; e[0] += 1; e[1] += dt
                if dt > 3e-3:       # a stall: in the launch call (host), on the device (events), or in the wait after it?
                    stalls.append((_n, 1e3 * dt, 1e3 * (t1 - t), e0.elapsed_time(e1)))
                if _n == "a3d_linear":
                    slow.append((dt, int(a[4]), int(a[5]), int(a[6]), a[0].value or 0, a[13].value or 0))
                return r
            try:
                setattr(lib, n, wrap)
            except Exception:
                pass
        t0 = time.perf_counter()
        for _ in range(REPEATS):
            once()
        wall = 1e3 * (time.perf_counter() - t0) / REPEATS
        for n, fn in orig.items():
            try:
                setattr(lib, n, fn)
            except Exception:
                pass
        tot = sum(v[1] for v in acc.values())
        print(f"  synchronised wall {wall:.1f} ms, of which inside library calls {1e3 * tot:.1f} ms")
        for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"  {n:34s} {c:4d} calls {1e3 * t:8.2f} ms")
        for n, dt, host, devt in stalls:
            print(f"  STALL {n}: {dt:.2f} ms wall = {host:.2f} ms inside the call + the wait; {devt:.2f} ms between the events on the stream")
        for dt, n, cin, cout, pi, po in sorted(slow, reverse=True)[:2]:
            print(f"  slowest a3d_linear: {1e3 * dt:7.3f} ms  rows {n} {cin} -> {cout} (in {pi:#x}, out {po:#x})")


for a in sys.argv[1:] or ["96,48,80,48", "26,104,13,91"]:
    run([int(x) for x in a.split(",")], by_op=True)
