#!/usr/bin/env python3
"""Micro-benchmark of single backbone ops through the C ABI (for rocprofv3 --pmc runs and A/B
work on spconv.hip).  Usage: python tools/conv_bench.py [--voxels N] [--reps R] [--only NAME]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from agile3d_amd import lib as L  # noqa: E402
from agile3d_amd.engine import Scene  # noqa: E402
from agile3d_amd.synthetic import make_scene  # noqa: E402
from gpu_util import OneOp, pack_weight  # noqa: E402

CASES = [  # name, kind, level_in, cin, cout, kvol
    ("L0_conv3_96_96", L.OP_CONV3, 0, 96, 96, 27),
    ("L0_conv3_128_96", L.OP_CONV3, 0, 128, 96, 27),
    ("L1_conv3_96_96", L.OP_CONV3, 1, 96, 96, 27),
    ("L1_conv3_32_32", L.OP_CONV3, 1, 32, 32, 27),
    ("L2_conv3_64_64", L.OP_CONV3, 2, 64, 64, 27),
    ("L2_conv3_128_128", L.OP_CONV3, 2, 128, 128, 27),
    ("L3_conv3_256_256", L.OP_CONV3, 3, 256, 256, 27),
    ("L4_conv3_256_256", L.OP_CONV3, 4, 256, 256, 27),
    ("L3_conv3_128_128", L.OP_CONV3, 3, 128, 128, 27),
    ("L4_conv3_128_256", L.OP_CONV3, 4, 128, 256, 27),
    ("L3_conv3_384_256", L.OP_CONV3, 3, 384, 256, 27),
    ("L2_conv3_192_128", L.OP_CONV3, 2, 192, 128, 27),
    ("L2_conv3_32_64", L.OP_CONV3, 2, 32, 64, 27),
    ("L3_down_128_128", L.OP_DOWN, 3, 128, 128, 8),
    ("L2_down_64_64", L.OP_DOWN, 2, 64, 64, 8),
    ("L4_up_256_256", L.OP_UP, 4, 256, 256, 8),
    ("L3_up_256_128", L.OP_UP, 3, 256, 128, 8),
    ("L2_up_128_96", L.OP_UP, 2, 128, 96, 8),
    ("L0_linear_128_128", L.OP_LINEAR, 0, 128, 128, 1),
    ("L0_linear_128_96", L.OP_LINEAR, 0, 128, 96, 1),
    ("L1_up_96_96", L.OP_UP, 1, 96, 96, 8),
    ("L0_down_32_32", L.OP_DOWN, 0, 32, 32, 8),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=80_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--sweep", action="store_true", help="small-level kernel: every geometry (bn, ch, parts) per case, against stream-K")
    a = ap.parse_args()
    lib = L.load()
    sc = make_scene(a.voxels, seed=0)
    scene = Scene(torch.from_numpy(sc["coords"]).cuda())
    print("levels", scene.n)
    pairs3 = []
    for lvl in range(5):
        npad = (max(scene.n[lvl], 1) + 127) // 128 * 128
        nb = scene.table(lvl, L.TAB_NBR27).reshape(27, npad)
        gm = scene.table(lvl, L.TAB_GMASK27)
        pairs3.append((int((nb[:, :scene.n[lvl]] < scene.n[lvl]).sum()),
                       sum(bin(int(m)).count("1") for m in gm) * 16))
    for name, kind, lvl, cin, cout, kvol in CASES:
        if a.only and not __import__("re").search(a.only, name):
            continue
        g = torch.Generator().manual_seed(1)
        W = torch.randn(kvol, cin, cout, generator=g) / (cin * 4) ** 0.5
        def measure():
            op = OneOp(scene, kind, lvl, cin, cout, kvol, pack_weight(W.cuda()), None, None, relu=True)
            op.buffer(0).normal_()
            op.buffer(0)[-1].zero_()
            for _ in range(3):
                op.run()
            lib.a3d_profile_read(None, 0)
            lib.a3d_profile_enable(1)
            for _ in range(a.reps):
                op.run()
            lib.a3d_profile_enable(0)
            buf = (L.ProfEntry * 4096)()
            n = lib.a3d_profile_read(buf, 4096)
            ms = {}
            for i in range(n):
                ms.setdefault(buf[i].id, []).append(buf[i].ms)
            return buf[0].bn, buf[0].ksplit, float(np.median(ms[0] if 0 in ms else ms[L.PROF_DENSE])), float(np.median(ms.get(1, [0.0])))
        if kind == L.OP_CONV3:
            pairs, slots = pairs3[lvl]
        elif kind == L.OP_UP:
            pairs = slots = scene.n[lvl - 1]
        elif kind == L.OP_DOWN:
            pairs = slots = scene.n[lvl]
        else:
            pairs = slots = scene.n[lvl]
        fl = 2.0 * pairs * cin * cout
        if a.sweep:
            before = lib.a3d_conv_deep_mode(0)
            _, _, t_sk, _ = measure()
            lib.a3d_conv_deep_mode(1)
            bn1, ch1, t_auto, _ = measure()
            row = [f"{name:20s} stream-K {t_sk * 1e3:6.1f} | model ({bn1 % 1000},{ch1}) {t_auto * 1e3:6.1f} |"]
            for bn, ch in ((128, 32), (64, 64), (64, 32), (32, 64), (32, 32)):
                if cout % bn or cin % ch:
                    continue
                best = None
                for parts in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24):
                    lib.a3d_conv_deep_mode((bn // 32) | (ch // 32) << 4 | parts << 8)
                    b, _, t, _ = measure()
                    if b < 1000:
                        continue
                    if best is None or t < best[0]:
                        best = (t, parts)
                if best:
                    row.append(f"({bn},{ch}) {best[0] * 1e3:6.1f} P={best[1]:2d} |")
            lib.a3d_conv_deep_mode(before)
            print(" ".join(row), flush=True)
            continue
        bn_, ks_, t_conv, t_epi = measure()
        print(f"{name:20s} bn={bn_:3d} ksplit={ks_:2d} conv {t_conv * 1e3:8.1f} us  epi {t_epi * 1e3:6.1f} us  "
              f"algorithmic {fl / t_conv / 1e9:7.2f} TF/s  issued(group-active) {2.0 * slots * cin * cout / t_conv / 1e9:7.2f} TF/s")


if __name__ == "__main__":
    main()
