"""The scan datasets in front of the hot path, with the reference's interface (row f-4 of SURVEY.md section 8):

    InterMultiObj3DSegDataset, collation_fn        datasets/InterMultiObj3DSegDataset.py:21-136
    InterSingleObj3DSegDataset, collation_fn_single datasets/InterSingleObj3DSegDataset.py:21-122
    build_dataset(split, args)                      datasets/__init__.py:4-10

A sample is a binary PLY scan (fields x, y, z, R, G, B, label; ``agile3d_amd.ply.read_ply``), min-shifted, colours
scaled to [0, 1], voxelised with ``sparse_quantize`` and returned as the same 9-tuple the reference's training and
evaluation loops unpack.  ``voxelize_on`` moves the voxelisation onto a GPU (``a3d_sparse_quantize``: identical
integers, results copied back so the tuple keeps its numpy types); by default it runs on the host like the
reference's loader workers.  The augmentation consumes ``np.random`` in the reference's order (two flips, a
quarter-turn choice, a free rotation about z).
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .ply import read_ply
from .sparse import batched_coordinates, sparse_quantize


def _rotz(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def _augment(points):
    """datasets/InterMultiObj3DSegDataset.py:97-116 (in place, like the reference)."""
    if np.random.random() > 0.5:        # flip along the YZ plane
        points[:, 0] = -1 * points[:, 0]
    if np.random.random() > 0.5:        # flip along the XZ plane
        points[:, 1] = -1 * points[:, 1]
    quarter = np.random.choice([0, np.pi / 2, np.pi, np.pi / 2 * 3])
    points[:, 0:3] = np.dot(points[:, 0:3], np.transpose(_rotz(quarter)))
    angle = (np.random.random() * 2 * np.pi) - np.pi
    points[:, 0:3] = np.dot(points[:, 0:3], np.transpose(_rotz(angle)))
    return points


def _load_scan(path):
    """-> (coords float32 [N,3] shifted so each axis starts at 0, colours float64 [N,3] in [0,1], raw record array)"""
    pc = read_ply(path)
    coords = np.column_stack([pc["x"] - pc["x"].min(), pc["y"] - pc["y"].min(), pc["z"] - pc["z"].min()])
    colors = np.column_stack([pc["R"], pc["G"], pc["B"]]) / 255
    return coords.astype(np.float32), colors, pc


def _voxelize(coords_full, quantization_size, device):
    """coords_qv, unique_map, inverse_map as numpy arrays (host path) -- or through the GPU kernel when a device is given."""
    if device is None:
        return sparse_quantize(coordinates=coords_full, quantization_size=quantization_size, return_index=True,
                               return_inverse=True)
    q, umap, inv = sparse_quantize(coordinates=torch.from_numpy(coords_full).to(device),
                                   quantization_size=quantization_size, return_index=True, return_inverse=True)
    return q.cpu().numpy(), umap.cpu().numpy(), inv.cpu().numpy()


class InterMultiObj3DSegDataset(Dataset):
    """Multi-object samples ``<scene>_obj_<K>`` of a JSON scene list (datasets/InterMultiObj3DSegDataset.py:21-95)."""

    def __init__(self, scan_folder, scene_list, quantization_size, transforms=None, voxelize_on=None):
        super().__init__()
        self.quantization_size = quantization_size
        self.scan_folder = scan_folder
        with open(scene_list) as f:
            self.data_samples = json.load(f)
        self.dataset_list = list(self.data_samples.keys())
        self.dataset_size = len(self.dataset_list)
        self.transforms = transforms
        self.voxelize_on = voxelize_on

    def __len__(self):
        return len(self.dataset_list)

    def __getitem__(self, i):
        sample_name = self.dataset_list[i]
        scene_name, num_obj = sample_name.split("_obj_")
        num_obj = int(num_obj)
        coords_full, colors_full, pc = _load_scan(os.path.join(self.scan_folder, scene_name + ".ply"))
        labels_full = pc["label"].astype(np.int32)
        if self.transforms:
            coords_full = self.augment(coords_full)
        data_sample = self.data_samples[sample_name]
        labels_full_new = self.compute_labels(labels_full, data_sample["obj"]) if data_sample else labels_full
        coords_qv, unique_map, inverse_map = _voxelize(coords_full, self.quantization_size, self.voxelize_on)
        raw_coords_qv = coords_full[unique_map]
        feats_qv = colors_full[unique_map]
        labels_qv = labels_full_new[unique_map]
        click_idx_qv = {}
        if data_sample:                      # pre-recorded clicks: every click must sit on its object
            click_idx_qv = data_sample["clicks"]
            for obj_id, click_id in click_idx_qv.items():
                assert np.all(labels_qv[click_id] == int(obj_id)), "data sample not match!"
        return (coords_qv, raw_coords_qv, feats_qv, labels_qv, labels_full_new, inverse_map, click_idx_qv, scene_name,
                num_obj)

    def compute_labels(self, ori_labels, correspondence):
        """{new object id: original instance id} -> labels 1..K (float64 zeros elsewhere, as the reference builds them)."""
        new_labels = np.zeros(ori_labels.shape)
        for new_obj_id, ori_obj_id in correspondence.items():
            new_labels[ori_labels == ori_obj_id] = int(new_obj_id)
        return new_labels

    def augment(self, point_cloud):
        return _augment(point_cloud)

    def rotz(self, t):
        return _rotz(t)


class InterSingleObj3DSegDataset(Dataset):
    """Single-object samples: rows (scene, object id) of an ``.npy`` list (datasets/InterSingleObj3DSegDataset.py:21-77);
    ``crop=True`` reads ``<scene>/<scene>_crop_<id>.ply`` whose label field is already binary."""

    def __init__(self, scan_folder, object_list, quantization_size, crop=False, transforms=None, voxelize_on=None):
        super().__init__()
        self.quantization_size = quantization_size
        self.scan_folder = scan_folder
        self.dataset_list = np.load(object_list)
        self.dataset_size = len(self.dataset_list)
        self.crop = crop
        self.transforms = transforms
        self.voxelize_on = voxelize_on

    def __len__(self):
        return len(self.dataset_list)

    def __getitem__(self, i):
        scene_name = self.dataset_list[i, 0]
        object_id = self.dataset_list[i, 1]
        if self.crop:
            path = os.path.join(self.scan_folder, scene_name, scene_name + "_crop_" + object_id + ".ply")
        else:
            path = os.path.join(self.scan_folder, scene_name + ".ply")
        coords_full, colors_full, pc = _load_scan(path)
        if self.crop:
            labels_full = pc["label"].astype(np.int32)
        else:
            labels_full = (pc["label"] == int(object_id)).astype(np.int32)
        if self.transforms:
            coords_full = self.augment(coords_full)
        coords_qv, unique_map, inverse_map = _voxelize(coords_full, self.quantization_size, self.voxelize_on)
        return (coords_qv, coords_full[unique_map], colors_full[unique_map], labels_full[unique_map], labels_full,
                inverse_map, {}, scene_name, object_id)

    def augment(self, point_cloud):
        return _augment(point_cloud)

    def rotz(self, t):
        return _rotz(t)


def collation_fn(data_labels):
    """Batch of dataset tuples -> what ``train_one_epoch`` / ``Evaluate`` unpack (datasets/...Dataset.py:126-136):
    batched int32 coordinates [sum n, 4] (batch index first), concatenated raw coordinates and colours as float32
    tensors, per-sample label tensors; inverse maps, click dicts, scene names and the 9th field stay tuples."""
    coords, raw_coords, feats, labels, labels_full, inverse_map, click_idx, scene_name, last = list(zip(*data_labels))
    coords_batch = batched_coordinates(coords)
    feats_batch = torch.from_numpy(np.concatenate(feats, 0)).float()
    labels_batch = [torch.from_numpy(l) for l in labels]
    raw_coords_batch = torch.from_numpy(np.concatenate(raw_coords, 0)).float()
    labels_full = [torch.from_numpy(l) for l in labels_full]
    return coords_batch, raw_coords_batch, feats_batch, labels_batch, labels_full, inverse_map, click_idx, scene_name, last


def make_scan_transforms(split):
    return split == "train"


def build_multi_obj_dataset(split, args):
    lists = {"train": args.train_list, "val": args.val_list}
    dataset = InterMultiObj3DSegDataset(args.scan_folder, lists[split], args.voxel_size,
                                        transforms=make_scan_transforms(split))
    return dataset, collation_fn


def build_single_obj_dataset(split, args):
    lists = {"train": args.train_list, "val": args.val_list}
    dataset = InterSingleObj3DSegDataset(args.scan_folder, lists[split], args.voxel_size, crop=args.crop,
                                         transforms=make_scan_transforms(split))
    return dataset, collation_fn


def build_dataset(split, args):
    """datasets/__init__.py:4-10."""
    if args.dataset_mode == "multi_obj":
        return build_multi_obj_dataset(split, args)
    if args.dataset_mode == "single_obj":
        return build_single_obj_dataset(split, args)
    raise ValueError(f"dataset mode {args.dataset_mode} not supported")
