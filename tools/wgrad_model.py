#!/usr/bin/env python3
"""What bounds k_wgrad (csrc/wgrad.hip) on the training batch: from the scene tables of 4 x 80 k voxels, per level,
  * issued 16-position groups per offset (gmask27) against the algorithmic pairs -> the share of MFMA work that multiplies zeros;
  * the (chunk, offset) work items of the launch plan: largest item, and the makespan of a greedy schedule over the device's
    workgroup slots against the perfectly balanced one.
Usage: python tools/wgrad_model.py [--voxels N] [--batch B] [--slots 512]"""
import argparse
import heapq
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agile3d_amd import lib as L  # noqa: E402
from agile3d_amd.engine import Scene  # noqa: E402
from agile3d_amd.synthetic import make_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=80_000)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--slots", type=int, default=512)
    a = ap.parse_args()
    coords = np.concatenate([make_scene(a.voxels, seed=b, batch_index=b)["coords"] for b in range(a.batch)])
    sc = Scene(torch.from_numpy(coords).cuda())
    print("levels", sc.n)
    for lvl in range(5):
        n = sc.n[lvl]
        npad = (max(n, 1) + 127) // 128 * 128
        nb = sc.table(lvl, L.TAB_NBR27).reshape(27, npad)[:, :n]
        gm = sc.table(lvl, L.TAB_GMASK27)
        ng = (n + 15) // 16
        gm = gm[:ng]
        present = ((gm[None, :] >> np.arange(27, dtype=np.uint32)[:, None]) & 1).astype(np.int64)   # [27][ng]
        pairs = int((nb < n).sum())
        issued = int(present.sum()) * 16
        # the round-6 plan: equal segments of each offset's group list, column j of the launch on XCD j % 8
        cnt = present.sum(1)
        for tgt in (512, 3072):
            seg = max(8, -(-int(cnt.sum()) // tgt))
            nseg = -(-cnt // seg)
            smax = int(nseg.max())
            per_xcd = np.zeros(8, dtype=np.int64)
            for k in range(27):
                for col in range(smax):
                    sgm = col * int(nseg[k]) // smax
                    if sgm < nseg[k] and (col == 0 or sgm != (col - 1) * int(nseg[k]) // smax):
                        per_xcd[col % 8] += min(seg, int(cnt[k]) - sgm * seg)
            print(f"L{lvl} segments: tgt {tgt} seg_len {seg} smax {smax} slots {int(nseg.sum())} groups per XCD {per_xcd.tolist()} "
                  f"max / mean {per_xcd.max() / per_xcd.mean():.3f}")
        for tgt in (1536, 3072, 6144):
            chunks = max(1, min(256, (tgt + 26) // 27, (ng + 3) // 4))
            cg = (ng + chunks - 1) // chunks
            chunks = (ng + cg - 1) // cg
            # work of item (chunk, k) = the busiest wave's groups (wave w takes groups w, w + 4, ... of the chunk)
            items = []
            for c in range(chunks):
                pr = present[:, c * cg:min(ng, (c + 1) * cg)]
                per_wave = np.stack([pr[:, w::4].sum(1) for w in range(4)], 0)      # [4][27]
                items.extend(per_wave.max(0).tolist())
            items = np.array(items, dtype=np.int64)
            wave_sum = items.sum()
            # greedy list schedule in launch order over `slots` workgroup slots
            h = [0] * a.slots
            heapq.heapify(h)
            for t in items:
                heapq.heappush(h, heapq.heappop(h) + int(t) + 2)         # + 2 groups' worth: fold + partial write
            mk = max(h)
            ideal = present.sum() / 4.0 / a.slots
            print(f"L{lvl} tgt {tgt}: n {n} pairs {pairs} issued rows {issued} ({pairs / max(issued, 1):.3f} useful), chunks {chunks} x {cg} groups, "
                  f"items {len(items)}, busiest-wave sum / balanced {wave_sum / (present.sum() / 4.0):.3f}, makespan {mk} vs ideal {ideal:.1f} "
                  f"-> {ideal / mk:.3f}")


if __name__ == "__main__":
    main()
