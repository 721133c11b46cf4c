"""Scan ingest in front of the hot path: binary PLY reader / writer and the dataset
classes + collate, against outputs of the REFERENCE's own code on the committed fixture files
(tests/golden/make_dataset_goldens.py; tests/golden/data/ holds PLY files the reference's writer produced)."""
import json
import os

import numpy as np
import pytest
import torch

from agile3d_amd import datasets as D
from agile3d_amd.ply import read_ply, write_ply

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")
G = np.load(os.path.join(HERE, "golden", "dataset_cases.npz"))
SCAN = os.path.join(DATA, "scans", "scene0001_00.ply")
NAMES = ["coords_qv", "raw_coords_qv", "feats_qv", "labels_qv", "labels_full", "inverse_map"]


def _check_fields(rec, prefix):
    keys = [k[len(prefix) + 1:] for k in G.files if k.startswith(prefix + "/") and k != prefix + "/faces"]
    assert list(rec.dtype.names) == keys
    for f in keys:
        assert rec[f].dtype.itemsize == G[f"{prefix}/{f}"].dtype.itemsize and rec[f].dtype.kind == G[f"{prefix}/{f}"].dtype.kind
        assert np.array_equal(rec[f], G[f"{prefix}/{f}"]), f


def _check_sample(tup, key, exact_float=True):
    assert len(tup) == 9
    for n, v in zip(NAMES, tup[:6]):
        ref = G[f"{key}/{n}"]
        v = np.asarray(v)
        assert v.shape == ref.shape and v.dtype == ref.dtype, (key, n, v.dtype, ref.dtype)
        assert np.array_equal(v, ref), (key, n)
    assert tup[6] == json.loads(str(G[f"{key}/click_json"]))
    assert str(tup[7]) == str(G[f"{key}/scene_name"]) and str(tup[8]) == str(G[f"{key}/last"])


def test_read_ply_matches_reference_reader():
    _check_fields(read_ply(SCAN), "scan")
    _check_fields(read_ply(os.path.join(DATA, "scan_big_endian.ply")), "scan_be")
    vert, faces = read_ply(os.path.join(DATA, "mesh_small.ply"), triangular_mesh=True)
    _check_fields(vert, "mesh")
    assert faces.dtype == np.int32 and np.array_equal(faces, G["mesh/faces"])


def test_write_ply_produces_the_reference_writers_bytes(tmp_path):
    rec = read_ply(SCAN)
    xyz = np.column_stack([rec["x"], rec["y"], rec["z"]])
    rgb = np.column_stack([rec["R"], rec["G"], rec["B"]])
    out = str(tmp_path / "again")                      # extension is appended
    assert write_ply(out, [xyz, rgb, rec["label"]], ["x", "y", "z", "R", "G", "B", "label"]) is True
    assert open(out + ".ply", "rb").read() == open(SCAN, "rb").read()
    vert, faces = read_ply(os.path.join(DATA, "mesh_small.ply"), triangular_mesh=True)
    mesh = str(tmp_path / "mesh.ply")
    assert write_ply(mesh, (np.column_stack([vert["x"], vert["y"], vert["z"]]),
                            np.column_stack([vert["red"], vert["green"], vert["blue"]])),
                     ["x", "y", "z", "red", "green", "blue"], triangular_faces=faces)
    assert open(mesh, "rb").read() == open(os.path.join(DATA, "mesh_small.ply"), "rb").read()
    # a single 2-D array, columns -> fields
    assert write_ply(str(tmp_path / "pts.ply"), xyz[:10], ["x", "y", "z"])
    back = read_ply(str(tmp_path / "pts.ply"))
    assert np.array_equal(np.column_stack([back["x"], back["y"], back["z"]]), xyz[:10])


def test_ply_error_cases(tmp_path, capsys):
    pts = np.zeros((4, 3), np.float32)
    assert write_ply(str(tmp_path / "a.ply"), [pts, np.zeros(5)], ["x", "y", "z", "v"]) is False       # lengths differ
    assert write_ply(str(tmp_path / "a.ply"), [pts], ["x", "y"]) is False                              # too few names
    assert write_ply(str(tmp_path / "a.ply"), [np.zeros((2, 2, 2))], ["x"]) is False                   # > 2 dimensions
    assert "wrong" in capsys.readouterr().out
    asc = tmp_path / "ascii.ply"
    asc.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0.5\n")
    with pytest.raises(ValueError, match="not binary"):
        read_ply(str(asc))
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"plx\nformat binary_little_endian 1.0\nend_header\n")
    with pytest.raises(ValueError):
        read_ply(str(bad))
    with pytest.raises(ValueError):                      # a mesh needs triangular_mesh=True
        read_ply(os.path.join(DATA, "mesh_small.ply"))
    empty = tmp_path / "empty.ply"
    assert write_ply(str(empty), np.zeros((0, 3), np.float32), ["x", "y", "z"])
    assert len(read_ply(str(empty))) == 0


def test_multi_object_dataset_matches_reference():
    ds = D.InterMultiObj3DSegDataset(os.path.join(DATA, "scans"), os.path.join(DATA, "val_list.json"), 0.05)
    assert len(ds) == 2 and ds.dataset_list == ["scene0001_00_obj_2", "scene0001_00_obj_3"]
    s0, s1 = ds[0], ds[1]
    _check_sample(s0, "multi/0")
    _check_sample(s1, "multi/1")
    assert s1[8] == 3 and isinstance(s1[8], int) and set(s1[6]) == {"0", "1", "2", "3"}
    col = D.collation_fn([s0, s1])
    assert col[0].dtype == torch.int32 and np.array_equal(col[0].numpy(), G["multi/collate/coords"])
    assert col[1].dtype == torch.float32 and np.array_equal(col[1].numpy(), G["multi/collate/raw"])
    assert col[2].dtype == torch.float32 and np.array_equal(col[2].numpy(), G["multi/collate/feats"])
    assert [t.shape[0] for t in col[3]] == [len(s0[0]), len(s1[0])] and col[7] == ("scene0001_00", "scene0001_00")
    assert col[8] == (2, 3) and col[6][1] == s1[6]


def test_click_sanity_check_fires_on_a_wrong_click(tmp_path):
    lst = json.load(open(os.path.join(DATA, "val_list.json")))
    lst["scene0001_00_obj_3"]["clicks"]["1"][0] = lst["scene0001_00_obj_3"]["clicks"]["2"][0]
    p = tmp_path / "list.json"
    p.write_text(json.dumps(lst))
    ds = D.InterMultiObj3DSegDataset(os.path.join(DATA, "scans"), str(p), 0.05)
    with pytest.raises(AssertionError, match="data sample not match"):
        ds[1]


@pytest.mark.parametrize("seed", [3, 4])
def test_training_augmentation_consumes_numpy_rng_like_the_reference(seed):
    ds = D.InterMultiObj3DSegDataset(os.path.join(DATA, "scans"), os.path.join(DATA, "val_list.json"), 0.05,
                                     transforms=D.make_scan_transforms("train"))
    np.random.seed(seed)
    _check_sample(ds[0], f"multi/aug{seed}")


def test_single_object_dataset_matches_reference():
    olist = os.path.join(DATA, "object_ids.npy")
    ds = D.InterSingleObj3DSegDataset(os.path.join(DATA, "scans"), olist, 0.05)
    s0, s1 = ds[0], ds[1]
    _check_sample(s0, "single/0")
    _check_sample(s1, "single/1")
    _check_sample(D.InterSingleObj3DSegDataset(os.path.join(DATA, "scans"), olist, 0.05, crop=True)[0], "single/crop0")
    assert np.array_equal(D.collation_fn([s0, s1])[0].numpy(), G["single/collate/coords"])


def test_build_dataset_dispatch():
    from types import SimpleNamespace
    args = SimpleNamespace(dataset_mode="multi_obj", scan_folder=os.path.join(DATA, "scans"), train_list="",
                           val_list=os.path.join(DATA, "val_list.json"), voxel_size=0.05, crop=False)
    ds, fn = D.build_dataset("val", args)
    assert isinstance(ds, D.InterMultiObj3DSegDataset) and fn is D.collation_fn and ds.transforms is False
    args.dataset_mode, args.val_list = "single_obj", os.path.join(DATA, "object_ids.npy")
    ds, _ = D.build_dataset("val", args)
    assert isinstance(ds, D.InterSingleObj3DSegDataset) and len(ds) == 2
    args.dataset_mode = "other"
    with pytest.raises(ValueError):
        D.build_dataset("val", args)


def test_dataloader_with_workers_and_collate():
    ds = D.InterMultiObj3DSegDataset(os.path.join(DATA, "scans"), os.path.join(DATA, "val_list.json"), 0.05)
    dl = torch.utils.data.DataLoader(ds, batch_size=2, collate_fn=D.collation_fn, num_workers=2, shuffle=False)
    (batch,) = list(dl)
    assert np.array_equal(batch[0].numpy(), G["multi/collate/coords"])


@pytest.mark.gpu
def test_gpu_voxelisation_gives_the_same_tuple():
    ds = D.InterMultiObj3DSegDataset(os.path.join(DATA, "scans"), os.path.join(DATA, "val_list.json"), 0.05,
                                     voxelize_on="cuda")
    _check_sample(ds[0], "multi/0")
    _check_sample(ds[1], "multi/1")


@pytest.mark.gpu
def test_ply_to_evaluate_end_to_end(tmp_path):
    """File on disk -> dataset -> DataLoader(collate) -> Evaluate (backbone once, decoder + click simulator per round)
    -> results CSV -> EvaluatorMO: the whole chain of eval_multi_obj.py on the fixture scan, 3 objects x 20 clicks
    (the evaluator needs the 20-clicks-per-object rows: with fewer it has "no objects to eval", like the reference)."""
    import random
    import types

    from agile3d_amd import build_model, default_args, randomize_bn_stats
    from agile3d_amd.evaluate import Evaluate
    lst = json.load(open(os.path.join(DATA, "val_list.json")))
    vl = tmp_path / "val.json"
    vl.write_text(json.dumps({"scene0001_00_obj_3": lst["scene0001_00_obj_3"]}))
    ds, fn = D.build_dataset("val", types.SimpleNamespace(dataset_mode="multi_obj", scan_folder=os.path.join(DATA, "scans"),
                                                         train_list="", val_list=str(vl), voxel_size=0.05, crop=False))
    loader = torch.utils.data.DataLoader(ds, batch_size=1, collate_fn=fn, num_workers=1)
    torch.manual_seed(0)
    model = randomize_bn_stats(build_model(default_args())).eval().cuda()
    args = types.SimpleNamespace(output_dir=str(tmp_path), max_num_clicks=20, val_list=str(vl))
    random.seed(5)
    seen = []
    res = Evaluate(model, loader, args, torch.device("cuda"),
                   lambda idx, cur, pred, iou, ci, ct: seen.append((cur, float(iou), sum(len(v) for v in ci.values()))))
    rows = [l.split() for l in open(tmp_path / "val_results_multi.csv").read().strip().split("\n")]
    assert [int(r[0]) for r in rows] == [0] * 59 and all(r[1] == "0001_00" and r[2] == "3" for r in rows)
    assert [c for c, _, _ in seen] == [0] + list(range(3, 61))
    assert [float(r[3]) for r in rows] == [c / 3 for c, _, _ in seen]
    assert seen[0][1] == 0.0 and all(0.0 <= iou <= 1.0 for _, iou, _ in seen) and seen[1][2] == 3
    assert set(k for k in res if k.startswith("IoU")) == {"IoU@1", "IoU@3", "IoU@5", "IoU@10", "IoU@15"}
    assert all(1.0 <= res[k] <= 20.0 for k in res if k.startswith("NoC"))


@pytest.mark.gpu
def test_ply_to_training_epoch(tmp_path):
    """File on disk -> training dataset (augmentation on) -> DataLoader(collate) -> train_one_epoch: two iterations of
    the reference's loop (engine.py:26-179) on the fixture scan, then the LR schedule steps."""
    import random
    import types

    from agile3d_amd import build_model, default_args
    from agile3d_amd.criterion import build_mask_criterion
    from agile3d_amd.optim import AdamW
    from agile3d_amd.train_step import MultiStepLR, train_one_epoch
    tl = tmp_path / "train.json"                # training lists carry no pre-recorded clicks (the augmentation re-voxelises)
    tl.write_text(json.dumps({"scene0001_00_obj_3": {}, "scene0001_00_obj_2": {}}))
    ds, fn = D.build_dataset("train", types.SimpleNamespace(dataset_mode="multi_obj", scan_folder=os.path.join(DATA, "scans"),
                                                           train_list=str(tl), val_list="", voxel_size=0.05, crop=False))
    assert ds.transforms is True
    loader = torch.utils.data.DataLoader(ds, batch_size=1, collate_fn=fn, shuffle=False)
    np.random.seed(2), torch.manual_seed(2), random.seed(2)
    args = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])
    model = build_model(args).cuda()
    opt = AdamW(model.named_parameters(), lr=1e-4, weight_decay=1e-4)
    sched = MultiStepLR(opt, [1])
    lines = []
    stats, it = train_one_epoch(model, build_mask_criterion(args), loader, opt, torch.device("cuda"), 0, 0, 0.1, 1, lines.append)
    assert it == 2 and len(lines) == 2 and np.isfinite(stats["loss"]) and stats["grad_norm"] > 0
    assert {"loss_bce", "loss_dice", "loss_bce_1", "loss_dice_1"} <= set(stats)
    sched.step()
    assert abs(opt.lr - 1e-5) < 1e-12


@pytest.mark.gpu
def test_training_script_validation_pass(tmp_path):
    """engine.evaluate (engine.py:182-300): interactive protocol + per-round losses + the results table, from the PLY
    fixture through the DataLoader."""
    import random
    import types

    from agile3d_amd import build_model, default_args
    from agile3d_amd.criterion import build_mask_criterion
    from agile3d_amd.train_step import evaluate
    lst = json.load(open(os.path.join(DATA, "val_list.json")))
    vl = tmp_path / "val.json"
    vl.write_text(json.dumps({"scene0001_00_obj_3": lst["scene0001_00_obj_3"]}))
    ds, fn = D.build_dataset("val", types.SimpleNamespace(dataset_mode="multi_obj", scan_folder=os.path.join(DATA, "scans"),
                                                         train_list="", val_list=str(vl), voxel_size=0.05, crop=False))
    loader = torch.utils.data.DataLoader(ds, batch_size=1, collate_fn=fn)
    args = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"], max_num_clicks=20,
                        valResults_dir=str(tmp_path / "val"), val_list=str(vl))
    torch.manual_seed(0)
    random.seed(1)
    model = build_model(args).cuda()
    stats = evaluate(model, build_mask_criterion(args), loader, args, 7, torch.device("cuda"))
    rows = open(tmp_path / "val" / "val_results_epoch_7.csv").read().strip().split("\n")
    assert len(rows) == 59 and rows[0].split()[3] == "0.0"
    assert {"loss", "mIoU", "loss_bce", "loss_dice_1", "loss_bce_unscaled", "NoC@80", "IoU@5"} <= set(stats)
    assert np.isfinite(stats["loss"]) and 0.0 <= stats["mIoU"] <= 1.0
