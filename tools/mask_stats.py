import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from agile3d_amd import lib as L
from agile3d_amd.engine import Scene
from agile3d_amd.synthetic import make_scene
sc = make_scene(80_000, seed=0)
s = Scene(torch.from_numpy(sc["coords"]).cuda())
for lvl in range(3):
    gm = s.table(lvl, L.TAB_GMASK27)
    n = s.n[lvl]
    ng = (n + 15) // 16
    gm = gm[:ng]
    pad = (-ng) % 4
    g4 = np.concatenate([gm, np.zeros(pad, np.uint32)]).reshape(-1, 4)
    un = g4[:, 0] | g4[:, 1] | g4[:, 2] | g4[:, 3]
    pc = lambda x: np.array([bin(int(v)).count("1") for v in x.ravel()]).reshape(x.shape)
    p_g, p_u = pc(g4), pc(un)
    print(f"L{lvl}: groups {ng}, mean offsets/group {p_g.mean():.2f}, mean union/tile {p_u.mean():.2f}, presence = {p_g.sum() / (4 * p_u.sum()):.3f}")
    # per-stage distribution of number of present waves
    cnt = np.zeros(5)
    for k in range(27):
        pres = ((g4 >> k) & 1).sum(1)
        act = ((un >> k) & 1) == 1
        for c in range(5):
            cnt[c] += (pres[act] == c).sum()
    print("   waves present per union stage:", (cnt / cnt.sum()).round(3))
