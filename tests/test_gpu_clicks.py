"""-m gpu: the interactive loop around forward_mask through the C ABI:
label argmax, IoU counts, the click simulator and the Evaluate loop.
  * against vectors captured from the reference's utils/seg.py (tests/golden/clicks_cases.npz);
  * against the CPU oracle on fresh seeded inputs;
  * at the benchmark size (80 k voxels) against an exact float64 k-d tree.
Integer results (labels, counts, click rows, click order) are compared bit-exactly; a click row may
differ from the reference only where two candidates tie within the fp32 error of torch.cdist's
matmul formula (checked explicitly), distances within 2e-4 m."""
import os
import random
import types

import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

from agile3d_amd import SparseTensor, build_model, clicks as pc, default_args, randomize_bn_stats
from agile3d_amd.evaluate import Evaluate
from agile3d_amd.synthetic import make_scene
from oracle import backbone as ob, clicks as oc, decoder as od

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "clicks_cases.npz"))
NAMES = [str(n) for n in G["names"]]
DIST_TOL = 2e-4


def unflat(keys, lens, vals):
    out, p = {}, 0
    for k, n in zip(keys, lens):
        out[str(int(k))] = [int(v) for v in vals[p:p + n]]
        p += n
    return out


def exact_outside(xyz, cid):
    """float64 reference: per error point the distance to the nearest point of another cluster."""
    d = np.full(len(xyz), -1.0)
    x64 = xyz.astype(np.float64)
    for c in np.unique(cid[cid >= 0]):
        m = cid == c
        d[m] = cKDTree(x64[~m]).query(x64[m])[0]
    return d


def check_clusters(got, xyz, pred, labels, want_rows=None):
    cid = np.where(pred != labels, 96 * labels.astype(np.int64) + 11 * pred.astype(np.int64), -1)
    d = exact_outside(xyz, cid)
    ids = sorted(np.unique(cid[cid >= 0]).tolist())
    assert [c["cluster_id"] for c in got] == ids
    for k, c in enumerate(got):
        m = cid == c["cluster_id"]
        assert cid[c["row"]] == c["cluster_id"] and c["label"] == labels[c["row"]] and c["pred"] == pred[c["row"]]
        assert abs(c["error_size"] - d[m].max()) <= 1e-5, (c, d[m].max())
        assert d[c["row"]] >= d[m].max() - 1e-6          # it IS a farthest point
        first = int(np.flatnonzero(m & (d >= d[c["row"]] - 1e-7))[0])
        assert c["row"] <= first or d[c["row"]] > d[first]          # lowest row among exact ties
        if want_rows is not None and want_rows[k] != c["row"]:
            assert abs(d[want_rows[k]] - d[c["row"]]) <= DIST_TOL, "differs from the reference beyond a cdist tie"


@pytest.mark.parametrize("name", NAMES)
def test_cluster_table_vs_reference(name):
    xyz, pred, lab = G[f"{name}/xyz"], G[f"{name}/pred"], G[f"{name}/labels"]
    got = pc.error_clusters(torch.from_numpy(pred).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(xyz).cuda())
    want = G[f"{name}/clusters"]
    assert [c["cluster_id"] for c in got] == [int(v) for v in want[:, 0]]
    check_clusters(got, xyz, pred, lab, [int(v) for v in want[:, 1]])
    assert np.abs(np.array([c["error_size"] for c in got]) - want[:, 2]).max() <= DIST_TOL
    same = sum(c["row"] == int(w) for c, w in zip(got, want[:, 1]))
    print(f"{name}: {len(got)} clusters, {same} identical click rows")


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("mode", ["eval0", "evalk", "train"])
def test_simulated_clicks_vs_reference(name, mode):
    xyz, pred, lab = (torch.from_numpy(G[f"{name}/{k}"]).cuda() for k in ("xyz", "pred", "labels"))
    seed = int(G[f"{name}/seed"])
    cur, training, p = {"eval0": (0, False, torch.zeros(len(lab), device="cuda")), "evalk": (7, False, pred.long()),
                        "train": (None, True, pred.long())}[mode]
    random.seed(seed)
    clicks, num, pos, times = pc.get_simulated_clicks(p, lab.long(), xyz, cur, training=training)
    want_c = unflat(G[f"{name}/{mode}/keys"], G[f"{name}/{mode}/lens"], G[f"{name}/{mode}/rows"])
    want_t = unflat(G[f"{name}/{mode}/keys"], G[f"{name}/{mode}/lens"], G[f"{name}/{mode}/times"])
    assert num == int(G[f"{name}/{mode}/num"])
    assert clicks == want_c and list(clicks) == list(want_c) and times == want_t
    got_pos = np.stack([x.cpu().numpy() for k in clicks for x in pos[k]])
    assert np.array_equal(got_pos, G[f"{name}/{mode}/pos"])


@pytest.mark.parametrize("mode", ["eval0", "evalk", "train"])
def test_batched_round_equals_the_per_sample_loop(mode):
    """get_simulated_clicks_batch / mean_iou_scene_batch over all golden cases as ONE batch (error clusters on side
    streams, one host round trip) against the per-sample calls in the same order with the same ``random`` seed:
    identical clicks, order, positions and IoU bits -- and therefore identical to the reference's goldens."""
    xs = [torch.from_numpy(G[f"{n}/xyz"]).cuda() for n in NAMES]
    labs = [torch.from_numpy(G[f"{n}/labels"]).cuda().long() for n in NAMES]
    prs = [torch.from_numpy(G[f"{n}/pred"]).cuda().long() for n in NAMES]
    cur, training = {"eval0": (0, False), "evalk": (7, False), "train": (None, True)}[mode]
    if mode == "eval0":
        prs = [torch.zeros(len(l), device="cuda") for l in labs]
    random.seed(123)
    single = [pc.get_simulated_clicks(p, l, x, cur, training=training) for p, l, x in zip(prs, labs, xs)]
    random.seed(123)
    batch = pc.get_simulated_clicks_batch(prs, labs, xs, cur, training=training)
    nobj = [int((torch.unique(l) != 0).sum()) for l in labs]
    random.seed(123)
    batch_n = pc.get_simulated_clicks_batch(prs, labs, xs, cur, training=training, num_objs=nobj)
    for a, b, c in zip(single, batch, batch_n):
        for other in (b, c):
            assert a[0] == other[0] and (a[0] is None or list(a[0]) == list(other[0])) and a[1] == other[1] and a[3] == other[3]
            if a[2] is not None:
                assert all(torch.equal(u, v) for k in a[2] for u, v in zip(a[2][k], other[2][k]))
    ious = pc.mean_iou_scene_batch(prs, labs)
    for (iou_b, per_b), p, l in zip(ious, prs, labs):
        iou_s, per_s = pc.mean_iou_scene(p, l)
        assert (np.float32(iou_b.numpy()) == np.float32(iou_s.numpy()) or (np.isnan(iou_b.numpy()) and np.isnan(iou_s.numpy()))) and per_b == per_s


@pytest.mark.parametrize("name", NAMES)
def test_iou_and_weights_vs_reference(name):
    xyz, pred, lab = (torch.from_numpy(G[f"{name}/{k}"]).cuda() for k in ("xyz", "pred", "labels"))
    iou, per = pc.mean_iou_scene(pred, lab)
    assert np.float32(iou.numpy()) == G[f"{name}/iou"]                       # same fp32 bits
    assert list(per) == [int(v) for v in G[f"{name}/iou_ids"]]
    assert np.array_equal(np.array(list(per.values())), G[f"{name}/iou_vals"])
    c2 = unflat(G[f"{name}/extend/keys"], G[f"{name}/extend/lens"], G[f"{name}/extend/rows"])
    w = pc.cal_click_loss_weights(torch.zeros(len(lab), dtype=torch.long, device="cuda"), xyz, [lab], [c2])[0]
    # exact float64 statement of utils/seg.py:62-70 ...
    x64 = G[f"{name}/xyz"].astype(np.float64)
    rows = [r for v in c2.values() for r in v]
    d = cKDTree(x64[rows]).query(x64)[0]
    assert np.abs(w.cpu().numpy() - (0.8 + 1.2 * (1 - np.minimum(d, 0.3) / 0.3))).max() <= 2e-6
    # ... and the reference's own output, whose torch.cdist (matmul formula) is off by up to
    # sqrt(eps * |x|^2) ~ 1e-3 m next to a click: 1.2 / 0.3 * 1e-3 = 4e-3 on the weight
    assert np.abs(w.cpu().numpy() - G[f"{name}/weights"]).max() <= 4e-3


def test_iou_counts_through_inverse_map():
    rng = np.random.default_rng(3)
    n, nf = 5000, 23000
    pred = rng.integers(0, 7, n).astype(np.int32)
    inv = rng.integers(0, n, nf)
    lab_full = rng.integers(0, 9, nf).astype(np.int32)
    c = pc.iou_counts(torch.from_numpy(pred).cuda(), torch.from_numpy(lab_full).cuda(), torch.from_numpy(inv).cuda(), 16)
    pf = pred[inv]
    for i in range(16):
        assert c[0, i] == ((pf == i) & (lab_full == i)).sum() and c[1, i] == (pf == i).sum() and c[2, i] == (lab_full == i).sum()
    iou, per = pc.mean_iou_scene(torch.from_numpy(pred).cuda(), torch.from_numpy(lab_full).cuda(), torch.from_numpy(inv).cuda())
    want, wper = oc.mean_iou_scene(torch.from_numpy(pf).long(), torch.from_numpy(lab_full).long())
    assert np.float32(iou.numpy()) == np.float32(want.numpy()) and per == wper
    with pytest.raises(RuntimeError):
        pc.iou_counts(torch.from_numpy(pred).cuda(), torch.from_numpy(lab_full).cuda(), torch.from_numpy(inv + n).cuda())


def test_argmax_labels_and_click_overwrite():
    torch.manual_seed(1)
    logits = torch.randn(7001, 6, device="cuda")
    logits[5] = 0.25                       # all equal -> first index
    logits[6, 2] = logits[6, 4] = 9.0      # tie -> first of the two
    clicks = {"0": [11], "1": [3, 4], "2": [], "3": [7000, 3]}          # row 3 clicked twice: last wins
    got = pc.argmax_labels(logits, clicks).cpu()
    want = logits.cpu().argmax(-1)
    assert got[5] == 0 and got[6] == 2
    for k, rows in clicks.items():
        want[rows] = int(k)
    assert torch.equal(got.long(), want)
    assert torch.equal(pc.argmax_labels(logits).cpu().long(), logits.cpu().argmax(-1))


@pytest.mark.parametrize("voxels", [80_000, 300_000])
def test_full_size_scene_vs_kdtree(voxels):
    """BASELINE sizes (configs 2 and 5): 80 k / 300 k voxels, 10 objects, a prediction with large wrong regions.  From
    1 k points on the simulator runs its bounded search (sampled upper bounds, champions, survivors); the float64 k-d tree is exact."""
    sc = make_scene(voxels, seed=0)
    rng = np.random.default_rng(0)
    labels = np.where(sc["labels"] <= 10, sc["labels"], 0).astype(np.int32)
    xyz = sc["raw_xyz"]
    pred = labels.copy()
    for _ in range(12):
        i = rng.integers(len(xyz))
        pred[np.linalg.norm(xyz - xyz[i], axis=1) < rng.uniform(0.2, 1.2)] = rng.integers(0, 11)
    t = lambda a: torch.from_numpy(a).cuda()
    got = pc.error_clusters(t(pred), t(labels), t(xyz))
    assert len(got) > 5
    check_clusters(got, xyz, pred, labels)
    zero = pc.error_clusters(torch.zeros(len(labels), device="cuda"), t(labels), t(xyz))     # round 0 of the loop
    check_clusters(zero, xyz, np.zeros_like(labels), labels)
    assert pc.error_clusters(t(labels), t(labels), t(xyz)) == []
    assert pc.get_simulated_clicks(t(labels), t(labels), t(xyz), 3, training=False) == (None, None, None, None)


def _prune_cases():
    """Seeded (pred, labels, xyz) triples that stress the bounded search: salt-and-pepper errors (hundreds of tiny clusters),
    large blobs, everything wrong, sizes around the sampling stride, exact distance ties (points on an integer lattice)."""
    rng = np.random.default_rng(7)
    out = []
    # from kCoarseFrom = 20 000 points on (clicks.hip; 150 000 until round 6) a first bounding stage runs in front of the fine one
    # (every 16th point) -- upper bounds from a row's neighbours in a Morton order of the coordinates (error_clusters hands
    # one over from that size on), or from every 256th point without one (A3D_CLICK_ORDER=0) --: coherent regions, everything
    # wrong, exact distance ties on a lattice at 20 k .. 180 k points; the cases below that size exercise the one-stage
    # search around its own size thresholds
    for n, kind in [(1024, "noise"), (1500, "blobs"), (4099, "noise"), (20_011, "blobs"), (20_011, "all_wrong"),
                    (8192, "lattice"), (16_383, "blobs"), (16_384, "noise"), (65_537, "all_wrong"), (70_001, "half_wrong"),
                    (60_000, "lattice_big"), (150_001, "half_wrong"), (163_841, "all_wrong"), (180_000, "lattice_huge")]:
        xyz = rng.uniform(0, 4, (n, 3)).astype(np.float32)
        if kind == "lattice":
            xyz = rng.integers(0, 24, (n, 3)).astype(np.float32) * 0.05
        if kind == "lattice_big":
            xyz = rng.integers(0, 48, (n, 3)).astype(np.float32) * 0.05
        if kind == "lattice_huge":
            xyz = rng.integers(0, 72, (n, 3)).astype(np.float32) * 0.05
        labels = (xyz[:, 0] // 1).astype(np.int32) % 5
        pred = labels.copy()
        if kind == "noise":
            m = rng.random(n) < 0.3
            pred[m] = rng.integers(0, 7, m.sum())
        elif kind in ("blobs", "lattice"):
            for _ in range(6):
                c = xyz[rng.integers(n)]
                pred[np.linalg.norm(xyz - c, axis=1) < rng.uniform(0.3, 1.5)] = rng.integers(0, 7)
        elif kind == "all_wrong":
            pred = (labels + 1 + rng.integers(0, 2, n)).astype(np.int32)
        elif kind == "half_wrong":          # a few huge coherent regions (what a prediction of early weights looks like)
            m = (np.sin(2.1 * xyz[:, 0]) + np.cos(1.7 * xyz[:, 1]) + 0.3 * xyz[:, 2]) > 0.4
            pred[m] = (labels[m] + 1 + (xyz[m, 1] // 2).astype(np.int32)) % 6
        elif kind in ("lattice_big", "lattice_huge"):         # exact distance ties everywhere, most points wrong
            m = xyz.sum(1) > 1.0
            pred[m] = (labels[m] + 1) % 5
        out.append((pred.astype(np.int32), labels.astype(np.int32), xyz))
    return out


_PLAIN_SCRIPT = """
import sys, pickle, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import test_gpu_clicks as T
from agile3d_amd import clicks as pc
res = [pc.error_clusters(*(torch.from_numpy(a).cuda() for a in case)) for case in T._prune_cases()]
pickle.dump(res, open(sys.argv[1], "wb"))
"""


def test_bounded_search_equals_plain_pass(tmp_path):
    """The click simulator's bounded search (a coarse and a fine bounding stage: upper bounds from every 256th / 16th point, one
    exact champion per cluster, the survivors on to the next stage and finally the full pass) against the plain pass over all points in a second interpreter (A3D_CLICK_PRUNE=0): identical cluster
    lists -- rows and the bits of the error sizes -- and both against the float64 k-d tree."""
    import pickle, subprocess, sys
    out = tmp_path / "plain.pkl"
    env = dict(os.environ, A3D_CLICK_PRUNE="0")
    subprocess.run([sys.executable, "-c", _PLAIN_SCRIPT.format(root=os.path.dirname(HERE), tests=HERE), str(out)],
                   check=True, env=env, timeout=600)
    plain = pickle.load(open(out, "rb"))
    for (pred, labels, xyz), want in zip(_prune_cases(), plain):
        got = pc.error_clusters(*(torch.from_numpy(a).cuda() for a in (pred, labels, xyz)))
        assert len(got) == len(want) and len(got) >= 1
        for g, w in zip(got, want):
            assert g["cluster_id"] == w["cluster_id"] and g["row"] == w["row"], (g, w)
            assert np.float32(g["error_size"]).tobytes() == np.float32(w["error_size"]).tobytes(), (g, w)
        check_clusters(got, xyz, pred, labels)
    # one cluster covering the whole sample has no outside point at all: every bound is infinite, nothing is pruned,
    # and the call fails like the reference does
    n = 3000
    with pytest.raises(RuntimeError):
        pc.error_clusters(torch.full((n,), 3, device="cuda"), torch.full((n,), 2, device="cuda"), torch.rand(n, 3, device="cuda"))


def test_batched_clusters_and_spatial_order(monkeypatch):
    """a3d_click_clusters_batch: the samples of a round through ONE set of launches (device table of samples; sizes from
    1 k to 70 k points side by side, so every stage is on for some samples and off for others) == the per-sample calls; and
    the first bounding stage's spatial order (a3d_click_spatial_order: Morton order of the coordinates, cached per
    coordinate tensor) is a permutation with its inverse -- ANY permutation in its place gives the same clusters (the
    bounds stay bounds), as does none at all."""
    cases = [c for c in _prune_cases() if len(c[2]) <= 70_001]
    t = lambda a: torch.from_numpy(a).cuda()
    preds, labs, xyzs = [t(c[0]) for c in cases], [t(c[1]) for c in cases], [t(c[2]) for c in cases]
    single = [pc.error_clusters(p, l, x) for p, l, x in zip(preds, labs, xyzs)]
    assert pc.error_clusters_batch(preds, labs, xyzs) == single
    big = max(range(len(cases)), key=lambda i: len(cases[i][2]))
    order, inv = pc._spatial_order(xyzs[big])
    n = len(cases[big][2])
    assert sorted(order.cpu().tolist()) == list(range(n)) and torch.equal(inv[order.long()].cpu(), torch.arange(n, dtype=torch.int32))
    # neighbours in the order are neighbours in space: median distance of consecutive rows far below that of random pairs
    x = xyzs[big][order.long()]
    near = (x[1:] - x[:-1]).norm(dim=1).median().item()
    rand = (xyzs[big][torch.randperm(n, device="cuda")] - xyzs[big]).norm(dim=1).median().item()
    assert near < 0.2 * rand, (near, rand)
    perm = torch.randperm(n, device="cuda").to(torch.int32)
    pinv = torch.empty_like(perm)
    pinv[perm.long()] = torch.arange(n, dtype=torch.int32, device="cuda")
    for fake in ((perm, pinv), None):
        monkeypatch.setattr(pc, "_spatial_order", lambda xyz, fake=fake: fake if xyz.shape[0] == n else None)
        assert pc.error_clusters_batch(preds, labs, xyzs) == single


def test_one_sync_round_equals_the_two_calls():
    """mean_iou_and_clusters_batch (IoU counts on the caller's stream, error clusters on side streams, one host
    synchronisation) returns what mean_iou_scene_batch and error_clusters_batch return, inverse maps included."""
    rng = np.random.default_rng(3)
    preds, labs, labs_full, invs, xyzs = [], [], [], [], []
    for n in (1500, 4097, 2600):
        xyz = rng.uniform(0, 3, (n, 3)).astype(np.float32)
        lab = (xyz[:, 0] // 0.75).astype(np.int32)
        pred = lab.copy()
        m = rng.random(n) < 0.2
        pred[m] = rng.integers(0, 5, m.sum())
        inv = rng.integers(0, n, 2 * n)
        preds.append(torch.from_numpy(pred).cuda())
        labs.append(torch.from_numpy(lab).cuda())
        labs_full.append(torch.from_numpy(lab[inv]).cuda())
        invs.append(torch.from_numpy(inv).cuda())
        xyzs.append(torch.from_numpy(xyz).cuda())
    ious, clusters = pc.mean_iou_and_clusters_batch(preds, labs_full, invs, labs, xyzs)
    want_iou = pc.mean_iou_scene_batch(preds, labs_full, invs)
    want_cl = pc.error_clusters_batch(preds, labs, xyzs)
    assert clusters == want_cl
    for (a, pa), (b, pb) in zip(ious, want_iou):
        assert a.numpy().tobytes() == b.numpy().tobytes() and pa == pb


def test_bad_inputs_fail_loudly():
    with pytest.raises(RuntimeError):
        pc.error_clusters(torch.zeros(10), torch.zeros(10), torch.zeros(10, 3))          # CPU tensors
    x = torch.rand(100, 3, device="cuda")
    with pytest.raises(RuntimeError):
        pc.error_clusters(torch.full((100,), 300, device="cuda"), torch.zeros(100, device="cuda"), x)
    with pytest.raises(RuntimeError):                                                   # one cluster = everything
        pc.error_clusters(torch.ones(100, device="cuda"), torch.zeros(100, device="cuda"), x)


def test_evaluate_loop_teacher_forced(tmp_path):
    """Run the product's Evaluate on one synthetic scene and re-derive every round with the oracle from the
    product's own state (prediction, clicks, RNG state): CSV rows, IoU bits, next clicks, and the final
    round's logits (<= 1e-3)."""
    torch.manual_seed(0)
    model = randomize_bn_stats(build_model(default_args())).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    sc = make_scene(4000, seed=5)
    n = len(sc["coords"])
    ids = [i for i in np.unique(sc["labels"]) if i > 0][:3]
    labels = np.zeros(n, np.int64)
    for k, i in enumerate(ids, start=1):
        labels[sc["labels"] == i] = k
    rng = np.random.default_rng(1)
    inv = np.concatenate([np.arange(n), rng.integers(0, n, n)])             # full-res cloud = 2n points
    labels_full = labels[inv]
    batch = (torch.from_numpy(sc["coords"]), torch.from_numpy(sc["raw_xyz"]), torch.from_numpy(sc["feats"]),
             [torch.from_numpy(labels)], [torch.from_numpy(labels_full)], [torch.from_numpy(inv)],
             [{str(k): [1, 2] for k in range(4)}], ["scene0042_00"], [3])
    args = types.SimpleNamespace(output_dir=str(tmp_path), max_num_clicks=2, val_list=None)
    log = []

    def on_round(idx, current, pred, iou, ci, ct):
        log.append({"current": current, "pred": pred.cpu().clone(), "iou": np.float32(iou.numpy()),
                    "ci": {k: list(v) for k, v in ci.items()}, "ct": {k: list(v) for k, v in ct.items()},
                    "rng": random.getstate()})

    random.seed(9)
    csv = Evaluate(model, [batch], args, torch.device("cuda"), on_round)
    lines = open(csv).read().strip().split("\n")
    assert [r["current"] for r in log] == [0, 3, 4, 5, 6] and len(lines) == 5
    xyz = torch.from_numpy(sc["raw_xyz"])
    for r, line in zip(log, lines):
        want_iou, _ = oc.mean_iou_scene(r["pred"].long()[torch.from_numpy(inv)], torch.from_numpy(labels_full))
        assert r["iou"] == np.float32(want_iou.numpy())
        assert line == f"0 0042_00 3 {r['current'] / 3} {r['iou']}"
    assert log[0]["ci"] == {str(k): [] for k in range(4)}                    # click ids set null
    for a, b in zip(log[:-1], log[1:]):                                      # the clicks added after round a
        random.setstate(a["rng"])
        new, _, _, new_t = oc.get_simulated_clicks(a["pred"].long(), torch.from_numpy(labels), xyz, a["current"], False)
        ci, ct = oc.extend_clicks({k: list(v) for k, v in a["ci"].items()}, {k: list(v) for k, v in a["ct"].items()},
                                  new, new_t)
        assert (ci, ct) == (b["ci"], b["ct"])
    last = log[-1]
    ref_b = ob.forward_backbone(sd, sc["coords"], torch.from_numpy(sc["feats"]), xyz)
    ref = od.forward_mask(sd, ref_b["pcd_features"], xyz, ref_b["pos_enc"], last["ci"], last["ct"])[-1]
    want = ref.argmax(-1)
    for k, rows in last["ci"].items():
        want[rows] = int(k)
    agree = (want == last["pred"].long()).float().mean().item()
    top2 = ref.topk(2, dim=-1).values
    unsure = (top2[:, 0] - top2[:, 1]) < 2e-3                                # labels may differ only at near-ties
    assert bool(((want == last["pred"].long()) | unsure).all()), agree
    print(f"Evaluate: {len(lines)} rounds, final IoU {last['iou']:.4f}, label agreement with the oracle {agree:.5f}")


def test_evaluate_full_protocol(tmp_path):
    """The reference's protocol end to end: up to 20 clicks per object (eval_multi_obj.py:70,114) on a batch
    of one 5-object scene -> 100 clicks + 10 learned background queries (two 64-query blocks in the
    decoder), results CSV -> EvaluatorMO tables."""
    import json
    torch.manual_seed(0)
    model = randomize_bn_stats(build_model(default_args())).eval().cuda()
    sc = make_scene(5000, seed=8)
    n = len(sc["coords"])
    sizes = sorted(((int((sc["labels"] == i).sum()), i) for i in np.unique(sc["labels"]) if i > 0), reverse=True)
    labels = np.zeros(n, np.int64)
    for k, (_, i) in enumerate(sizes[:5], start=1):
        labels[sc["labels"] == i] = k
    batch = (torch.from_numpy(sc["coords"]), torch.from_numpy(sc["raw_xyz"]), torch.from_numpy(sc["feats"]),
             [torch.from_numpy(labels)], [torch.from_numpy(labels)], [torch.arange(n)],
             [{str(k): [] for k in range(6)}], ["scene0007_01"], [5])
    (tmp_path / "val.json").write_text(json.dumps({"scene0007_01_obj_5": {}}))
    args = types.SimpleNamespace(output_dir=str(tmp_path), max_num_clicks=20, val_list=str(tmp_path / "val.json"))
    seen = []
    random.seed(3)
    res = Evaluate(model, [batch], args, torch.device("cuda"),
                   lambda idx, cur, pred, iou, ci, ct: seen.append((cur, sum(len(v) for v in ci.values()))))
    lines = open(tmp_path / "val_results_multi.csv").read().strip().split("\n")
    assert [c for c, _ in seen] == [0] + list(range(5, 101)) and len(lines) == 97
    assert all(have <= cur for cur, have in seen) and seen[1][1] == 5      # one click per object after round 0
    assert seen[-1][1] > 64                                                # the wide (>64 query) path ran
    assert set(res) == {"NoC@50", "NoC@65", "NoC@80", "NoC@85", "NoC@90", "IoU@1", "IoU@3", "IoU@5", "IoU@10", "IoU@15"}
    assert all(0.0 <= res[k] <= 1.0 for k in res if k.startswith("IoU")) and all(1.0 <= res[k] <= 20.0 for k in res if k.startswith("NoC"))


def test_evaluate_single_object_protocol(tmp_path):
    """eval_single_obj.py's loop: binary labels, one click per round, rows '<idx> <scene> <object id> <clicks> <IoU>';
    every row's IoU is re-derived with the oracle from the prediction the product used."""
    from agile3d_amd.evaluate import EvaluateSingle
    torch.manual_seed(0)
    model = randomize_bn_stats(build_model(default_args())).eval().cuda()
    sc = make_scene(4000, seed=6)
    n = len(sc["coords"])
    big = max((i for i in np.unique(sc["labels"]) if i > 0), key=lambda i: (sc["labels"] == i).sum())
    labels = (sc["labels"] == big).astype(np.int64)
    batch = (torch.from_numpy(sc["coords"]), torch.from_numpy(sc["raw_xyz"]), torch.from_numpy(sc["feats"]),
             [torch.from_numpy(labels)], [torch.from_numpy(labels)], [torch.arange(n)], None, ["scene0011_00"], ["7"])
    args = types.SimpleNamespace(output_dir=str(tmp_path), max_num_clicks=6, val_list=None)
    log = []
    random.seed(4)
    csv = EvaluateSingle(model, [batch], args, torch.device("cuda"),
                         lambda idx, cur, pred, iou, ci, ct: log.append((cur, pred.cpu().clone(), np.float32(iou.numpy()),
                                                                        {k: list(v) for k, v in ci.items()})))
    lines = open(csv).read().strip().split("\n")
    assert [r[0] for r in log] == list(range(7)) and len(lines) == 7
    for (cur, pred, iou, ci), line in zip(log, lines):
        want, _ = oc.mean_iou_scene(pred.long(), torch.from_numpy(labels))
        assert iou == np.float32(want.numpy()) and line == f"0 0011_00 7 {cur} {iou}"
        assert sum(len(v) for v in ci.values()) == cur and set(ci) == {"0", "1"}
    assert log[1][3]["1"] and not log[1][3]["0"]            # the first click lands on the object


def test_batched_argmax_and_iou_counts_equal_the_per_sample_calls():
    """a3d_argmax_labels_batch / a3d_iou_counts_batch (all samples of a round in two launches / one launch) against the
    per-sample entry points: ragged sizes, different class counts, an empty click list, a row clicked for two objects (the
    LAST one wins, as in the reference's dict loop, eval_multi_obj.py:119-134), an inverse map on one sample."""
    from agile3d_amd import clicks as pc
    g = torch.Generator().manual_seed(11)
    sizes, classes = [5000, 1, 777, 20000], [4, 2, 7, 11]
    logits = [torch.randn(n, c, generator=g).cuda() for n, c in zip(sizes, classes)]
    clicks = [{"0": [3, 9], "1": [4999, 3], "2": [17]}, {}, {"0": [], "3": [5, 6, 7], "6": [776]}, {"10": [1, 19999], "0": [1]}]
    batch = pc.argmax_labels_batch(logits, clicks)
    for lg, ck, got in zip(logits, clicks, batch):
        want = pc.argmax_labels(lg, ck)
        assert got.dtype == torch.int32 and torch.equal(got, want)
        ref = lg.argmax(-1).to(torch.int32)
        for o, rows in ck.items():
            for r in rows:
                ref[r] = int(o)
        assert torch.equal(got, ref)
    assert int(batch[0][3]) == 1 and int(batch[3][1]) == 0          # the later dict entry overwrote the earlier one
    assert pc.argmax_labels_batch([], []) == []
    labels = [torch.randint(0, c, (n,), generator=g).to(torch.int32).cuda() for n, c in zip(sizes, classes)]
    inv = [None, None, torch.randint(0, 777, (1500,), generator=g).cuda(), None]
    labels[2] = torch.randint(0, 7, (1500,), generator=g).to(torch.int32).cuda()
    got = pc.iou_counts_batch(batch, labels, inv)
    for i in range(4):
        want = pc.iou_counts(batch[i], labels[i], inv[i], n_ids=256)
        assert np.array_equal(got[i], want), i
