#!/usr/bin/env python3
"""Does the time of one N-row linear (k_dense, 321 k rows, 128 -> 128) depend on WHICH device buffers it runs on?
Allocates `nbuf` activations through torch's caching allocator (optionally after fragmenting it the way a training iteration
does) and times every (input, output) neighbour pair with events:  python tools/linear_by_buffer.py [nbuf] [fragment]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agile3d_amd import lib as L
from agile3d_amd.train_decoder import _pack, _ptr, _stream

nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 24
fragment = len(sys.argv) > 2 and sys.argv[2] == "1"
lib = L.load()
n = 321107
torch.manual_seed(0)
junk = []
if fragment:                                  # odd-sized blocks allocated and freed in between
    for i in range(40):
        junk.append(torch.empty(int(1e6 * (3 + 7 * (i % 5))), device="cuda"))
    junk = junk[::2]
bufs = [torch.randn(n, 128, device="cuda") for _ in range(nbuf)]
W = torch.randn(128, 128, device="cuda") / 11
wp = _pack(W)[0]
bias = torch.zeros(128, device="cuda")
def run(x, y):
    L.check(lib.a3d_linear(_ptr(x), 128, None, 0, n, 128, 128, _ptr(wp), None, _ptr(bias), None, 0, 0, _ptr(y), 128,
                           None, 0, _stream()), "a3d_linear")
for i in range(nbuf):
    run(bufs[i], bufs[(i + 1) % nbuf])
torch.cuda.synchronize()
ts = []
for i in range(nbuf):
    x, y = bufs[i], bufs[(i + 1) % nbuf]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(x, y)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5 * 1e3)
print("us per linear by buffer pair:", " ".join(f"{t:.0f}" for t in ts))
print("addresses mod 2 MiB (KiB):", " ".join(str((b.data_ptr() % (2 << 20)) >> 10) for b in bufs))
print("min %.0f max %.0f us" % (min(ts), max(ts)))
