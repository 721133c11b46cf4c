"""-m gpu: the full hot path through the reference-shaped API.
  * forward_backbone vs the CPU oracle on seeded synthetic scenes (fp32, |diff| <= 1e-3);
  * forward_mask vs the golden vectors captured from the REFERENCE's own forward_mask
    (tests/golden, |diff| <= 1e-3 as north_star states; the oracle itself is <= 1e-5);
  * size-independent properties at the benchmark size (80 k voxels)."""
import os

import numpy as np
import pytest
import torch

from agile3d_amd import SparseTensor, build_model, default_args, randomize_bn_stats
from agile3d_amd.synthetic import make_clicks, make_scene
from conftest import arrays_to_clicks, golden_cases, load_case
from oracle import backbone as ob, decoder as od

pytestmark = pytest.mark.gpu
TOL = 1e-3   # north_star: per-point mask logits within 1e-3 fp32


@pytest.fixture(scope="module")
def model_and_sd():
    torch.manual_seed(0)
    m = randomize_bn_stats(build_model(default_args())).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.cuda(), sd


def _run_backbone(model, sc):
    x = SparseTensor(features=torch.from_numpy(sc["feats"]), coordinates=torch.from_numpy(sc["coords"]),
                     device="cuda")
    return model.forward_backbone(x, raw_coordinates=torch.from_numpy(sc["raw_xyz"]).cuda())


@pytest.mark.parametrize("n,seed", [(3000, 1), (9000, 2)])
def test_forward_backbone_matches_oracle(model_and_sd, n, seed):
    model, sd = model_and_sd
    sc = make_scene(n, seed=seed)
    pcd, aux, coords, pos = _run_backbone(model, sc)
    ref = ob.forward_backbone(sd, sc["coords"], torch.from_numpy(sc["feats"]), torch.from_numpy(sc["raw_xyz"]))
    err = (pcd.F.cpu() - ref["pcd_features"]).abs().max().item()
    perr = (pos[4][0][0].cpu() - ref["pos_enc"]).abs().max().item()
    print(f"backbone n={len(sc['coords'])}: pcd_features max|diff|={err:.3e} (scale "
          f"{ref['pcd_features'].abs().max():.2f}), pos_enc max|diff|={perr:.3e}")
    assert err <= TOL * max(1.0, ref["pcd_features"].abs().max().item())
    assert perr <= 1e-4
    assert pcd.F.shape == (len(sc["coords"]), 128) and torch.equal(pcd.C.cpu(), torch.from_numpy(sc["coords"]))
    # aux feature maps: same values as the oracle's, rows matched by coordinates
    for i, fm in enumerate(aux):
        F, Cc = fm.F.cpu(), fm.C.cpu().numpy()
        level = 4 - i
        oc = ref["levels"].levels[level].copy()
        oc[:, 1:] *= (1 << level)
        key = lambda c: [tuple(r) for r in c.tolist()]
        pos_of = {k: j for j, k in enumerate(key(oc))}
        rows = [pos_of[k] for k in key(Cc)]
        e = (F - ref["feature_maps"][i][rows]).abs().max().item()
        assert e <= TOL * max(1.0, ref["feature_maps"][i].abs().max().item()), (i, e)


@pytest.mark.parametrize("n,batch", [(3000, 1), (20000, 2)])
def test_fused_residual_projections_equal_the_separate_launches(model_and_sd, n, batch):
    """The BasicBlock downsample projections run inside the blocks' second convs (a3d_op.proj_*: one more "offset" with the
    block input as its source, both BatchNorm scales folded into the packed weights).  Against the program with separate
    1x1 launches (BackboneProgram(fuse_proj=False)): every feature map and the output, to rounding (the scale is applied
    to the weights instead of to the sum)."""
    from agile3d_amd.engine import BackboneProgram
    model, _ = model_and_sd
    eng = model._get_engine()
    scs = [make_scene(n, seed=10 + b, batch_index=b) for b in range(batch)]
    sc = {k: np.concatenate([s_[k] for s_ in scs]) for k in ("coords", "feats", "raw_xyz")}
    outs = []
    for fuse in (True, False):
        eng.refresh_weights_if_stale()
        with torch.no_grad():
            eng.program = BackboneProgram(model, eng.device, fuse_proj=fuse)
        assert eng.program.fused_projections == (7 if fuse else 0)      # all seven BasicBlock.downsample projections
        pcd, aux, _, _ = _run_backbone(model, sc)
        outs.append([pcd.F.clone()] + [a.F.clone() for a in aux])
    eng.mark_stale()
    for a, b in zip(*outs):
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= 2e-5 * scale, (a - b).abs().max().item()


@pytest.mark.parametrize("n,batch", [(3000, 1), (20000, 2), (80000, 1)])
def test_small_level_kernel_equals_stream_k_and_is_deterministic(model_and_sd, n, batch):
    """Levels with a few stages of work per CU run on k_conv_deep (static parts, both operands by LDS-DMA, a cut tile finished
    by its last arriver; spconv.hip) instead of the stream-K kernel (a3d_conv_deep_mode).  Same arithmetic, another
    partition of the sums: every feature map and the output agree to rounding; two runs of either are bit-identical (the
    order in which the parts of a tile are added does not depend on which part arrives last); and the profile shows that
    the small-level kernel really ran."""
    from agile3d_amd import lib as L
    model, _ = model_and_sd
    lib = L.load()
    scs = [make_scene(n, seed=20 + b, batch_index=b) for b in range(batch)]
    sc = {k: np.concatenate([s_[k] for s_ in scs]) for k in ("coords", "feats", "raw_xyz")}
    before = lib.a3d_conv_deep_mode(-1)
    outs = {}
    try:
        for mode in (1, 0):
            lib.a3d_conv_deep_mode(mode)
            runs = []
            for rep in range(2):
                if rep == 1:
                    lib.a3d_profile_read(None, 0)
                    lib.a3d_profile_enable(1)
                pcd, aux, _, _ = _run_backbone(model, sc)
                torch.cuda.synchronize()
                runs.append([pcd.F.clone()] + [a.F.clone() for a in aux])
            lib.a3d_profile_enable(0)
            buf = (L.ProfEntry * 4096)()
            cnt = lib.a3d_profile_read(buf, 4096)
            deep = sum(1 for i in range(cnt) if buf[i].id == 0 and buf[i].bn >= 1000)
            assert (deep > 0) == (mode == 1), (mode, deep)
            print(f"n={n} x {batch}, mode {mode}: {deep} of {cnt} profiled launches on k_conv_deep")
            for a, b in zip(*runs):
                assert torch.equal(a, b), "two runs differ"
            outs[mode] = runs[0]
    finally:
        lib.a3d_conv_deep_mode(before)
    for a, b in zip(outs[1], outs[0]):
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= 2e-5 * scale, (a - b).abs().max().item()


def test_forward_mask_matches_oracle_end_to_end(model_and_sd):
    model, sd = model_and_sd
    sc = make_scene(4000, seed=3)
    ci, ct = make_clicks(sc["labels"], n_objects=4, clicks_per_object=2, n_bg_clicks=2, seed=3)
    pcd, aux, coords, pos = _run_backbone(model, sc)
    out = model.forward_mask(pcd, aux, coords, pos, click_idx=[ci], click_time_idx=[ct])
    ref_b = ob.forward_backbone(sd, sc["coords"], torch.from_numpy(sc["feats"]), torch.from_numpy(sc["raw_xyz"]))
    ref = od.forward_mask(sd, ref_b["pcd_features"], torch.from_numpy(sc["raw_xyz"]), ref_b["pos_enc"], ci, ct)
    got = [a["pred_masks"][0] for a in out["aux_outputs"]] + [out["pred_masks"][0]]
    for i in range(3):
        err = (got[i].cpu() - ref[i]).abs().max().item()
        print(f"end-to-end iteration {i}: logits max|diff|={err:.3e} (scale {ref[i].abs().max():.2f})")
        assert err <= TOL * max(1.0, ref[i].abs().max().item())
    p = out["pred_masks"][0]
    assert p.shape == (len(sc["coords"]), 5) and p.is_cuda
    p.argmax(-1)[ci["1"]] = 1   # callers write into the result (eval_multi_obj.py:140)


@pytest.mark.parametrize("name", golden_cases())
def test_forward_mask_matches_reference_goldens(model_and_sd, name, decoder_weights):
    """Inputs and expected logits were produced by the reference's own Agile3d.forward_mask."""
    model, sd = model_and_sd
    # the goldens were generated with exactly these decoder weights (seed 0); verify, then run
    for k, v in decoder_weights.items():
        assert torch.equal(sd[k], v), k
    c = load_case(name)
    K = int(c["K"])
    ci, ct = arrays_to_clicks(c["click_rows"], c["click_objs"], c["click_times"], K)
    eng = model._get_engine()
    pcd, aux, coords, pos = eng.decoder_inputs(torch.from_numpy(c["feats128"]), torch.from_numpy(c["xyz"]))
    perr = np.abs(pos[4][0][0].cpu().numpy() - c["pos_enc"]).max()
    out = model.forward_mask(pcd, aux, coords, pos, click_idx=[ci], click_time_idx=[ct])
    got = [a["pred_masks"][0] for a in out["aux_outputs"]] + [out["pred_masks"][0]]
    worst = 0.0
    for i in range(3):
        ref = c[f"logits{i}"]
        err = np.abs(got[i].cpu().numpy() - ref).max()
        worst = max(worst, err / max(1.0, np.abs(ref).max()))
        print(f"{name} iteration {i}: max|diff| vs REFERENCE = {err:.3e} (scale {np.abs(ref).max():.2f})")
    print(f"{name}: pos_enc max|diff| = {perr:.3e}")
    assert perr <= 1e-4 and worst <= TOL


@pytest.mark.parametrize("n_obj,per_obj,n_bg", [(5, 7, 3), (6, 8, 5), (6, 9, 1), (8, 15, 10), (8, 24, 8)])
def test_forward_mask_many_clicks(model_and_sd, decoder_weights, n_obj, per_obj, n_bg):
    """48 / 63 queries (three / four 16-query tiles in one workgroup) and 65 / 140 / 210 queries (two to four
    64-query blocks: the full evaluation protocol reaches 20 clicks per object, eval_multi_obj.py:114)
    against the oracle's forward_mask on the same decoder inputs."""
    model, sd = model_and_sd
    g = torch.Generator().manual_seed(n_obj * 100 + per_obj)
    n = 6000
    feats = torch.randn(n, 128, generator=g) * 0.5
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([8.0, 6.0, 2.6])
    rows = torch.randperm(n, generator=g)[:n_obj * per_obj + n_bg].tolist()
    order = torch.randperm(len(rows), generator=g).tolist()           # click times are a global order
    ci = {str(o): rows[(o - 1) * per_obj:o * per_obj] for o in range(1, n_obj + 1)}
    ct = {str(o): order[(o - 1) * per_obj:o * per_obj] for o in range(1, n_obj + 1)}
    ci["0"], ct["0"] = rows[n_obj * per_obj:], order[n_obj * per_obj:]
    eng = model._get_engine()
    pcd, aux, coords, pos = eng.decoder_inputs(feats, xyz)
    out = model.forward_mask(pcd, aux, coords, pos, click_idx=[ci], click_time_idx=[ct])
    got = [a["pred_masks"][0] for a in out["aux_outputs"]] + [out["pred_masks"][0]]
    ref = od.forward_mask(sd, feats, xyz, pos[4][0][0].cpu(), ci, ct)
    for i in range(3):
        err = (got[i].cpu() - ref[i]).abs().max().item()
        print(f"{len(rows) + 10} queries, iteration {i}: logits max|diff|={err:.3e} (scale {ref[i].abs().max():.2f})")
        assert err <= TOL * max(1.0, ref[i].abs().max().item())


def test_batched_position_encoding_equals_per_sample_calls(model_and_sd):
    """a3d_posenc_fourier_batch (three launches for the whole batch, what forward_backbone calls) against one
    a3d_posenc_fourier per sample: every sample is normalised by ITS OWN min / max (agile3d.py:141-161) -- same bits."""
    import ctypes as C
    from agile3d_amd import lib as L
    model, _ = model_and_sd
    eng = model._get_engine()
    eng.refresh_decoder_if_stale(check_versions=True)
    lib = L.load()
    g = torch.Generator().manual_seed(77)
    sizes = [1, 37, 5000, 256, 12345]
    xyz = torch.cat([torch.rand(n, 3, generator=g) * torch.tensor([8.0, 6.0, 2.6]) + 3.0 * i for i, n in enumerate(sizes)]).cuda()
    ranges, s = [], 0
    for n in sizes:
        ranges.append((s, s + n))
        s += n
    pes, mms = eng._posenc_batch(xyz, ranges)
    tmp = torch.empty(256 * 6 * 4, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (a, b), pe, mm in zip(ranges, pes, mms):
        if b - a < 2:
            continue          # a single point has max == min: both paths divide by zero alike, nothing to compare
        ref = torch.empty((b - a, 128), dtype=torch.float32, device="cuda")
        rmm = torch.empty(6, dtype=torch.float32, device="cuda")
        L.check(lib.a3d_posenc_fourier(C.c_void_p(xyz[a:b].data_ptr()), b - a, eng.decoder.gauss_B_ptr, C.c_void_p(rmm.data_ptr()),
                                       C.c_void_p(ref.data_ptr()), C.c_void_p(tmp.data_ptr()), tmp.numel(), st), "a3d_posenc_fourier")
        assert torch.equal(mm, rmm) and torch.equal(pe, ref), (a, b)


def test_too_many_clicks_is_an_error(model_and_sd):
    model, _ = model_and_sd
    eng = model._get_engine()
    pcd, aux, coords, pos = eng.decoder_inputs(torch.randn(3000, 128), torch.rand(3000, 3))
    ci = {"0": [], "1": list(range(201))}                              # the time table has 200 entries
    with pytest.raises(Exception):
        model.forward_mask(pcd, aux, coords, pos, click_idx=[ci], click_time_idx=[{"0": [], "1": list(range(201))}])


def test_batch_of_two_equals_two_single_scenes(model_and_sd):
    """Inference has no cross-scene coupling (BatchNorm uses running stats; agile3d.py:192 loops over
    samples): a 2-scene batch must reproduce the two single-scene results."""
    model, sd = model_and_sd
    a, b = make_scene(2500, seed=4), make_scene(3500, seed=5)
    ca, ta = make_clicks(a["labels"], 2, 1, 0, seed=4)
    cb, tb = make_clicks(b["labels"], 3, 2, 1, seed=5)
    singles = []
    for sc, ci, ct in ((a, ca, ta), (b, cb, tb)):
        r = _run_backbone(model, sc)
        singles.append((r[0].F.clone(), model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])["pred_masks"][0]))
    cb2 = b["coords"].copy()
    cb2[:, 0] = 1
    x = SparseTensor(features=torch.from_numpy(np.concatenate([a["feats"], b["feats"]])),
                     coordinates=torch.from_numpy(np.concatenate([a["coords"], cb2])), device="cuda")
    raw = torch.from_numpy(np.concatenate([a["raw_xyz"], b["raw_xyz"]])).cuda()
    r = model.forward_backbone(x, raw_coordinates=raw)
    out = model.forward_mask(*r, click_idx=[ca, cb], click_time_idx=[ta, tb])
    na = len(a["coords"])
    assert (r[0].F[:na] - singles[0][0]).abs().max().item() <= 1e-5
    assert (r[0].F[na:] - singles[1][0]).abs().max().item() <= 1e-5
    assert (out["pred_masks"][0] - singles[0][1]).abs().max().item() <= 1e-4
    assert (out["pred_masks"][1] - singles[1][1]).abs().max().item() <= 1e-4


def test_batched_decoder_equals_per_sample_runs(model_and_sd):
    """a3d_decoder_forward_batch: samples with the same padded query count share one launch of each wide kernel per
    layer (sample table), others go separately -- as independent launch chains on side streams, joined to the caller's
    stream before the call returns.  Five samples -- 14, 22 and 26 queries (the last two share launches), one with 44, one
    with 85 (two query blocks) -- different object counts: every layer's logits must equal the single-sample results."""
    model, sd = model_and_sd
    specs = [(2600, 11, 2, 2, 0), (3100, 12, 3, 4, 0), (2800, 13, 4, 3, 4), (3000, 14, 2, 15, 4), (3200, 15, 5, 15, 0)]
    scenes = [make_scene(n, seed=sd_) for n, sd_, *_ in specs]
    clicks = [make_clicks(sc["labels"], sp[2], sp[3], sp[4], seed=sp[1]) for sc, sp in zip(scenes, specs)]
    singles = []
    for sc, (ci, ct) in zip(scenes, clicks):
        r = _run_backbone(model, sc)
        o = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
        singles.append([o["pred_masks"][0]] + [a["pred_masks"][0] for a in o["aux_outputs"]])
    coords = []
    for b, sc in enumerate(scenes):
        c = sc["coords"].copy()
        c[:, 0] = b
        coords.append(c)
    x = SparseTensor(features=torch.from_numpy(np.concatenate([sc["feats"] for sc in scenes])),
                     coordinates=torch.from_numpy(np.concatenate(coords)), device="cuda")
    raw = torch.from_numpy(np.concatenate([sc["raw_xyz"] for sc in scenes])).cuda()
    r = model.forward_backbone(x, raw_coordinates=raw)
    out = model.forward_mask(*r, click_idx=[c[0] for c in clicks], click_time_idx=[c[1] for c in clicks])
    for b in range(len(scenes)):
        got = [out["pred_masks"][b]] + [a["pred_masks"][b] for a in out["aux_outputs"]]
        for g, ref in zip(got, singles[b]):
            assert g.shape == ref.shape
            assert (g - ref).abs().max().item() <= 1e-4, (b, (g - ref).abs().max().item())


@pytest.mark.parametrize("n,n_obj,per_obj,n_bg", [(5000, 50, 1, 0), (4000, 100, 2, 0), (300, 30, 5, 3), (17, 4, 6, 1)])
def test_wide_tier_many_objects_and_tiny_samples(model_and_sd, n, n_obj, per_obj, n_bg):
    """The fused wide tier at the edges of its shapes: many objects with a click or two each (the per-object maxima of
    k_out_w are an LDS table of 2 x 32 x (objects + 1) floats: past the default 64 KB from ~43 objects on), a sample with
    fewer points than one workgroup iteration takes (32) and one with 300 -- against the oracle on the same inputs."""
    model, sd = model_and_sd
    g = torch.Generator().manual_seed(n * 3 + n_obj)
    feats = torch.randn(n, 128, generator=g) * 0.5
    xyz = torch.rand(n, 3, generator=g) * torch.tensor([8.0, 6.0, 2.6])
    need = n_obj * per_obj + n_bg
    rows = (torch.randperm(n, generator=g)[:need] if need <= n else torch.randint(0, n, (need,), generator=g)).tolist()
    order = torch.randperm(len(rows), generator=g).tolist()
    ci = {str(o): rows[(o - 1) * per_obj:o * per_obj] for o in range(1, n_obj + 1)}
    ct = {str(o): order[(o - 1) * per_obj:o * per_obj] for o in range(1, n_obj + 1)}
    ci["0"], ct["0"] = rows[n_obj * per_obj:], order[n_obj * per_obj:]
    eng = model._get_engine()
    pcd, aux, coords, pos = eng.decoder_inputs(feats, xyz)
    out = model.forward_mask(pcd, aux, coords, pos, click_idx=[ci], click_time_idx=[ct])
    got = [a["pred_masks"][0] for a in out["aux_outputs"]] + [out["pred_masks"][0]]
    ref = od.forward_mask(sd, feats, xyz, pos[4][0][0].cpu(), ci, ct)
    for i in range(3):
        err = (got[i].cpu() - ref[i]).abs().max().item()
        print(f"{n} points, {n_obj} objects, {len(rows) + 10} queries, iteration {i}: logits max|diff|={err:.3e}")
        assert got[i].shape == (n, n_obj + 1) and err <= TOL * max(1.0, ref[i].abs().max().item())


def test_wide_tier_samples_of_different_query_counts_share_one_launch_group(model_and_sd):
    """More than 64 queries: the samples of a call go through the fused wide kernels (csrc/decoder_wide.h) TOGETHER, each with
    its own tile count from the sample table, under the build that holds the longest list -- 70, 97, 139, 171 and 210
    queries here (4 to 11 objects), plus one 20-query sample that keeps the <= 64-query kernels on a side stream.  Every
    layer's logits must equal the single-sample results (where each sample runs the build of ITS size) and the oracle's;
    the second call on the same backbone output reads the first layer's keys / values / queries from the scene cache."""
    model, sd = model_and_sd
    specs = [(2600, 31, 4, 15, 0), (3100, 32, 11, 7, 10), (2800, 33, 7, 18, 3), (3000, 34, 5, 2, 0), (3300, 35, 8, 20, 1),
             (2900, 36, 10, 19, 10)]
    scenes = [make_scene(n, seed=sd_) for n, sd_, *_ in specs]
    clicks = [make_clicks(sc["labels"], sp[2], sp[3], sp[4], seed=sp[1]) for sc, sp in zip(scenes, specs)]
    assert [sum(len(v) for v in c[0].values()) + 10 for c in clicks] == [70, 97, 139, 20, 171, 210]
    singles = []
    for sc, (ci, ct) in zip(scenes, clicks):
        r = _run_backbone(model, sc)
        o = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
        singles.append([o["pred_masks"][0]] + [a["pred_masks"][0] for a in o["aux_outputs"]])
        ref = od.forward_mask(sd, r[0].F.cpu(), torch.from_numpy(sc["raw_xyz"]), r[3][4][0][0].cpu(), ci, ct)
        err = (singles[-1][0].cpu() - ref[-1]).abs().max().item()
        assert err <= TOL * max(1.0, ref[-1].abs().max().item()), err
    coords = []
    for b, sc in enumerate(scenes):
        c = sc["coords"].copy()
        c[:, 0] = b
        coords.append(c)
    x = SparseTensor(features=torch.from_numpy(np.concatenate([sc["feats"] for sc in scenes])),
                     coordinates=torch.from_numpy(np.concatenate(coords)), device="cuda")
    raw = torch.from_numpy(np.concatenate([sc["raw_xyz"] for sc in scenes])).cuda()
    r = model.forward_backbone(x, raw_coordinates=raw)
    for call in range(3):   # fused first layer, the call that fills the scene cache, a call that reads it
        out = model.forward_mask(*r, click_idx=[c[0] for c in clicks], click_time_idx=[c[1] for c in clicks])
        for b in range(len(scenes)):
            got = [out["pred_masks"][b]] + [a["pred_masks"][b] for a in out["aux_outputs"]]
            for g, ref in zip(got, singles[b]):
                assert g.shape == ref.shape
                assert (g - ref).abs().max().item() <= 1e-4, (call, b, (g - ref).abs().max().item())


def test_scene_cache_of_first_layer_keys_and_values(model_and_sd):
    """forward_mask keeps the click-independent keys / values of the first layer's click-to-scene attention per scene from
    its second call on one backbone output (the interactive loop's ~100 passes, eval_multi_obj.py:112-160): the first call
    (fused kernel), the second (fills the cache) and the later ones (read it) agree, for a batch with a growing click set --
    20, then 44, then 85 queries (the multi-block path) --, each against a fresh backbone output that has no cache; the
    cache is dropped when the decoder's weights change."""
    model, sd = model_and_sd
    scenes = [make_scene(2700, seed=21), make_scene(3300, seed=22)]
    coords = []
    for b, sc in enumerate(scenes):
        c = sc["coords"].copy()
        c[:, 0] = b
        coords.append(c)
    x = SparseTensor(features=torch.from_numpy(np.concatenate([sc["feats"] for sc in scenes])),
                     coordinates=torch.from_numpy(np.concatenate(coords)), device="cuda")
    raw = torch.from_numpy(np.concatenate([sc["raw_xyz"] for sc in scenes])).cuda()
    r = model.forward_backbone(x, raw_coordinates=raw)
    st = r[0]._a3d
    rounds = [(2, 5, 0), (2, 15, 4), (5, 15, 0), (5, 15, 0)]         # (objects, clicks per object, background clicks)
    for k, (n_obj, per, n_bg) in enumerate(rounds):
        clicks = [make_clicks(sc["labels"], n_obj, per, n_bg, seed=40 + b) for b, sc in enumerate(scenes)]
        got = model.forward_mask(*r, click_idx=[c[0] for c in clicks], click_time_idx=[c[1] for c in clicks])
        fresh = model.forward_backbone(x, raw_coordinates=raw)       # first call on a new scene state: no cache
        want = model.forward_mask(*fresh, click_idx=[c[0] for c in clicks], click_time_idx=[c[1] for c in clicks])
        assert fresh[0]._a3d.kv0 is None
        assert (st.kv0 is None) == (k == 0)
        for b in range(2):
            assert (got["pred_masks"][b] - want["pred_masks"][b]).abs().max().item() <= 1e-4, (k, b)
            for ga, wa in zip(got["aux_outputs"], want["aux_outputs"]):
                assert (ga["pred_masks"][b] - wa["pred_masks"][b]).abs().max().item() <= 1e-4, (k, b)
    # new decoder weights: the cached keys / values are stale and must be refilled, not reused
    eng = model._get_engine()
    old_epoch = st.kv0_version
    saved = {name: prm.detach().clone() for name, prm in model.named_parameters() if name.startswith("c2s_attention")}
    assert saved
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name in saved:
                prm.mul_(1.5)
    eng.mark_stale()
    clicks = [make_clicks(sc["labels"], 2, 5, 0, seed=40 + b) for b, sc in enumerate(scenes)]
    got = model.forward_mask(*r, click_idx=[c[0] for c in clicks], click_time_idx=[c[1] for c in clicks])
    assert st.kv0_version != old_epoch
    fresh = model.forward_backbone(x, raw_coordinates=raw)
    want = model.forward_mask(*fresh, click_idx=[c[0] for c in clicks], click_time_idx=[c[1] for c in clicks])
    for b in range(2):
        assert (got["pred_masks"][b] - want["pred_masks"][b]).abs().max().item() <= 1e-4
    with torch.no_grad():                                            # the module-scoped model goes back to its weights
        for name, prm in model.named_parameters():
            if name in saved:
                prm.copy_(saved[name])
    eng.mark_stale()


def test_many_small_samples_in_one_batch(model_and_sd):
    """70 samples: more than one sample table (64 entries) -> the batch is cut into two groups of launches; the
    workgroup shares of tiny samples are clamped to what their few 16-point groups can use."""
    model, sd = model_and_sd
    scenes, clicks, coords = [], [], []
    for b in range(70):
        sc = make_scene(400 + 13 * b, seed=100 + b, n_boxes=4)
        scenes.append(sc)
        clicks.append(make_clicks(sc["labels"], 2, 1 + b % 2, 0, seed=b))
        c = sc["coords"].copy()
        c[:, 0] = b
        coords.append(c)
    x = SparseTensor(features=torch.from_numpy(np.concatenate([sc["feats"] for sc in scenes])),
                     coordinates=torch.from_numpy(np.concatenate(coords)), device="cuda")
    raw = torch.from_numpy(np.concatenate([sc["raw_xyz"] for sc in scenes])).cuda()
    r = model.forward_backbone(x, raw_coordinates=raw)
    out = model.forward_mask(*r, click_idx=[c[0] for c in clicks], click_time_idx=[c[1] for c in clicks])
    for b in (0, 1, 37, 63, 64, 69):
        rb = _run_backbone(model, scenes[b])
        ref = model.forward_mask(*rb, click_idx=[clicks[b][0]], click_time_idx=[clicks[b][1]])["pred_masks"][0]
        got = out["pred_masks"][b]
        assert got.shape == ref.shape == (len(scenes[b]["coords"]), 3)
        assert (got - ref).abs().max().item() <= 1e-4, b


def test_full_size_properties(model_and_sd):
    """BASELINE.json config 2 (80 k voxels, 10 clicks): size-independent properties.
    (a) row-permutation equivariance: shuffling the caller's row order permutes the outputs;
    (b) determinism: two runs are bit-identical;
    (c) translation of the voxel grid by a multiple of 16 leaves the features unchanged."""
    model, sd = model_and_sd
    sc = make_scene(80_000, seed=0)
    ci, ct = make_clicks(sc["labels"], 5, 2, 0, seed=0)
    r1 = _run_backbone(model, sc)
    o1 = model.forward_mask(*r1, click_idx=[ci], click_time_idx=[ct])["pred_masks"][0]
    r2 = _run_backbone(model, sc)
    o2 = model.forward_mask(*r2, click_idx=[ci], click_time_idx=[ct])["pred_masks"][0]
    assert torch.equal(r1[0].F, r2[0].F) and torch.equal(o1, o2), "non-deterministic"
    assert torch.isfinite(o1).all() and o1.shape == (len(sc["coords"]), 6)
    n = len(sc["coords"])
    perm = np.random.default_rng(0).permutation(n)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    sp = {k: v[perm] for k, v in sc.items()}
    cip = {k: [int(inv[r]) for r in v] for k, v in ci.items()}
    r3 = _run_backbone(model, sp)
    o3 = model.forward_mask(*r3, click_idx=[cip], click_time_idx=[ct])["pred_masks"][0]
    assert (r3[0].F - r1[0].F[perm]).abs().max().item() <= 1e-5
    assert (o3 - o1[perm]).abs().max().item() <= 1e-4
    st = dict(sc)
    st["coords"] = sc["coords"].copy()
    st["coords"][:, 1:] += np.array([32, -48, 16], np.int32)
    r4 = _run_backbone(model, st)
    assert (r4[0].F - r1[0].F).abs().max().item() <= 1e-5


@pytest.mark.parametrize("voxels,n_obj,per_obj,seed", [(80_000, 5, 2, 0), (300_000, 5, 4, 2)])
def test_benchmark_sizes_match_oracle(model_and_sd, voxels, n_obj, per_obj, seed):
    """BASELINE.json configs[1] (80 k voxels, 10 clicks) and configs[4] (300 k voxels, 20 clicks) against the CPU oracle
    itself, not only through size-independent properties: forward_backbone's features and the logits of all three
    decoder iterations within north_star's 1e-3 (fp32).  The oracle needs ~3 s / ~30 s at these sizes with 8 threads
    (torch's default of one thread per core is several times slower on a many-core host: the per-offset GEMMs are small)."""
    model, sd = model_and_sd
    sc = make_scene(voxels, seed=seed)
    ci, ct = make_clicks(sc["labels"], n_obj, per_obj, 0, seed=seed)
    pcd, aux, coords, pos = _run_backbone(model, sc)
    out = model.forward_mask(pcd, aux, coords, pos, click_idx=[ci], click_time_idx=[ct])
    got = [a["pred_masks"][0].cpu() for a in out["aux_outputs"]] + [out["pred_masks"][0].cpu()]
    feats_gpu = pcd.F.cpu()
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(8, nthr))
    try:
        ref_b = ob.forward_backbone(sd, sc["coords"], torch.from_numpy(sc["feats"]), torch.from_numpy(sc["raw_xyz"]))
        ref = od.forward_mask(sd, ref_b["pcd_features"], torch.from_numpy(sc["raw_xyz"]), ref_b["pos_enc"], ci, ct)
    finally:
        torch.set_num_threads(nthr)
    err_b = (feats_gpu - ref_b["pcd_features"]).abs().max().item()
    print(f"{len(sc['coords'])} voxels: pcd_features max|diff| = {err_b:.3e} (scale {ref_b['pcd_features'].abs().max():.2f})")
    assert err_b <= TOL * max(1.0, ref_b["pcd_features"].abs().max().item())
    for i in range(3):
        err = (got[i] - ref[i]).abs().max().item()
        print(f"{len(sc['coords'])} voxels, {n_obj * per_obj} clicks, iteration {i}: logits max|diff| = {err:.3e} "
              f"(scale {ref[i].abs().max():.2f})")
        assert err <= TOL * max(1.0, ref[i].abs().max().item())
    assert got[-1].shape == (len(sc["coords"]), n_obj + 1)


def test_two_scenes_in_flight_on_two_streams(model_and_sd):
    """bench.py issues consecutive scenes round-robin on two HIP streams.  The library keeps no state
    between calls that two in-flight scenes could share: results must be bit-identical to serial runs."""
    model, _ = model_and_sd
    jobs = []
    for seed, n in ((11, 3000), (12, 7000), (13, 4500), (14, 6000)):
        sc = make_scene(n, seed=seed)
        ci, ct = make_clicks(sc["labels"], 3, 2, 1, seed=seed)
        jobs.append((SparseTensor(features=torch.from_numpy(sc["feats"]), coordinates=torch.from_numpy(sc["coords"]),
                                  device="cuda"), torch.from_numpy(sc["raw_xyz"]).cuda(), ci, ct))

    def run(job):
        x, raw, ci, ct = job
        r = model.forward_backbone(x, raw_coordinates=raw)
        return r[0].F, model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])["pred_masks"][0]

    serial = [run(j) for j in jobs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        outs = []
        for i, j in enumerate(jobs * 2):
            with torch.cuda.stream(streams[i % 2]):
                outs.append(run(j))
        torch.cuda.synchronize()
        for i, (f, m) in enumerate(outs):
            assert torch.equal(f, serial[i % len(jobs)][0]) and torch.equal(m, serial[i % len(jobs)][1]), (rep, i)


def test_outdoor_scan_size_stress(model_and_sd):
    """BASELINE.json configs[4]: ~300 k voxels, 20 clicks.  Size-independent properties only (the oracle needs
    minutes at this size): bit-determinism, finite logits, and agreement with a run on a row-permuted copy."""
    model, _ = model_and_sd
    sc = make_scene(300_000, seed=2)
    n = len(sc["coords"])
    ci, ct = make_clicks(sc["labels"], 10, 2, 0, seed=2)
    r = _run_backbone(model, sc)
    o1 = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])["pred_masks"][0]
    o2 = model.forward_mask(*_run_backbone(model, sc), click_idx=[ci], click_time_idx=[ct])["pred_masks"][0]
    assert o1.shape == (n, 11) and torch.isfinite(o1).all() and torch.equal(o1, o2)
    perm = np.random.default_rng(0).permutation(n)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    sp = {k: (v[perm] if k != "labels" else v[perm]) for k, v in sc.items()}
    cip = {k: [int(inv[i]) for i in v] for k, v in ci.items()}
    op = model.forward_mask(*_run_backbone(model, sp), click_idx=[cip], click_time_idx=[ct])["pred_masks"][0]
    err = (op[torch.from_numpy(inv).cuda()] - o1).abs().max().item()
    print(f"300k voxels: permuted-input max|diff| = {err:.2e}")
    assert err <= 1e-4


@pytest.mark.parametrize("switch", ["A3D_FUSED_C2S=0", "A3D_WIDE_FROM=65", "A3D_FUSED_S2O=0"])
def test_unfused_decoder_path_still_matches_goldens(switch):
    """The decoder's other LIVE paths against the reference's goldens: A3D_FUSED_C2S=0 keeps the separate projection GEMMs +
    attention kernels at every query count (what 225 .. 256 queries run anyway); A3D_WIDE_FROM=65 keeps k_q_s2c +
    k_out_ln_mask for 33 .. 64 queries, which the wide tier serves by default; A3D_FUSED_S2O=0 runs k_s2c_w + k_out_w
    instead of the one-kernel scene-to-click half (k_s2o_w) at 33 .. 80 queries.  The library reads the
    variables once per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, "tests")
from agile3d_amd import build_model, default_args, randomize_bn_stats
from conftest import arrays_to_clicks, load_case
torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
worst = 0.0
for name in ("n4096_k5x2", "n3000_k10_bg", "n2048_k1", "n2000_k6_q46", "n1800_k8_q62", "n1500_k7_q75", "n150_k12_q144", "n220_k10_q205"):
    c = load_case(name)
    ci, ct = arrays_to_clicks(c["click_rows"], c["click_objs"], c["click_times"], int(c["K"]))
    r = model._get_engine().decoder_inputs(torch.from_numpy(c["feats128"]), torch.from_numpy(c["xyz"]))
    out = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
    got = [a["pred_masks"][0] for a in out["aux_outputs"]] + [out["pred_masks"][0]]
    for i in range(3):
        worst = max(worst, float(np.abs(got[i].cpu().numpy() - c[f"logits{i}"]).max() / max(1.0, np.abs(c[f"logits{i}"]).max())))
print("WORST", worst)
assert worst <= 1e-3
'''
    k_, v_ = switch.split("=")
    env = dict(os.environ, **{k_: v_})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "WORST" in res.stdout


def test_query_chain_helper_workgroups_match_single_workgroup(tmp_path):
    """The decoder's query-side layer with FFN / projection helper workgroups (default, A3D_QL_HELPERS=8) against the
    single-workgroup chain (A3D_QL_HELPERS=1): the switch is read once per process, so each setting runs in its own
    interpreter; the FFN's sum over hidden chunks is re-associated (eight partial sums), everything else is identical."""
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from agile3d_amd import SparseTensor, build_model, default_args, randomize_bn_stats
from agile3d_amd.synthetic import make_clicks, make_scene
torch.manual_seed(3)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
outs = []
for n, objs, cpo in ((6000, 3, 2), (9000, 5, 4)):          # 16 and 30+ queries: two query-tile counts
    sc = make_scene(n, seed=n)
    ci, ct = make_clicks(sc["labels"], objs, cpo, 1, seed=1)
    x = SparseTensor(features=torch.from_numpy(sc["feats"]).cuda(), coordinates=torch.from_numpy(sc["coords"]).cuda())
    r = model.forward_backbone(x, raw_coordinates=torch.from_numpy(sc["raw_xyz"]).cuda())
    o = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
    outs.append(o["pred_masks"][0].cpu().numpy())
    outs += [a["pred_masks"][0].cpu().numpy() for a in o["aux_outputs"]]
np.savez(sys.argv[2], *outs)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for nh in ("1", "8"):
        out = str(tmp_path / f"logits_{nh}.npz")
        env = dict(os.environ, A3D_QL_HELPERS=nh)
        subprocess.run([sys.executable, "-c", script, root, out], check=True, env=env, timeout=600)
        z = np.load(out)
        res[nh] = [z[k] for k in z.files]
    assert len(res["1"]) == len(res["8"]) == 6
    worst = max(float(np.abs(a - b).max()) for a, b in zip(res["1"], res["8"]))
    scale = max(float(np.abs(a).max()) for a in res["1"])
    print(f"helper workgroups vs single workgroup: max |diff| {worst:.2e} on logits of scale {scale:.1f}")
    assert worst <= 1e-4 * max(1.0, scale)


def test_training_mode_forward_invalidates_the_folded_backbone():
    """A train-mode forward moves the BatchNorm running statistics (through raw pointers); with no optimiser step after it
    (BN recalibration under no_grad) the next eval-mode forward must fold the NEW statistics, and num_batches_tracked
    counts the batch like nn.BatchNorm1d does."""
    torch.manual_seed(1)
    model = randomize_bn_stats(build_model(default_args())).cuda().eval()
    sc = make_scene(3000, seed=5)
    before = _run_backbone(model, sc)[0].F.clone()
    nbt0 = int(model.backbone.bn0.bn.num_batches_tracked)
    model.train()
    with torch.no_grad():
        _run_backbone(model, sc)
    model.eval()
    assert int(model.backbone.bn0.bn.num_batches_tracked) == nbt0 + 1
    after = _run_backbone(model, sc)[0].F
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref = ob.forward_backbone(sd, sc["coords"], torch.from_numpy(sc["feats"]), torch.from_numpy(sc["raw_xyz"]))
    err = (after.cpu() - ref["pcd_features"]).abs().max().item()
    moved = (after - before).abs().max().item()
    print(f"eval after a train-mode forward: vs oracle on the new statistics {err:.2e}; moved by {moved:.2e}")
    assert moved > 1e-3 and err <= TOL * max(1.0, ref["pcd_features"].abs().max().item())


@pytest.mark.parametrize("switch", ["A3D_FUSED_S2C"])
def test_one_pass_scene_to_click_half_matches_the_two_kernel_path(tmp_path, switch):
    """A3D_FUSED_S2C: k_s2c_out (scene-to-click attention + output projection + LayerNorm + mask head in one pass, <= ~24 queries,
    default) against k_q_s2c + k_out_ln_mask (A3D_FUSED_S2C=0): same arithmetic per element, so the logits, the label
    bytes feeding the next layer's attention mask and the intermediate (aux) logits agree to rounding; both against the
    reference's goldens.  The switch is read once per process -> two interpreters."""
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from agile3d_amd import SparseTensor, build_model, default_args, randomize_bn_stats
from agile3d_amd.synthetic import make_clicks, make_scene
from conftest import arrays_to_clicks, load_case
torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
outs, worst = [], 0.0
for name in ("n4096_k5x2", "n3000_k10_bg", "n2048_k1"):
    c = load_case(name)
    ci, ct = arrays_to_clicks(c["click_rows"], c["click_objs"], c["click_times"], int(c["K"]))
    r = model._get_engine().decoder_inputs(torch.from_numpy(c["feats128"]), torch.from_numpy(c["xyz"]))
    out = model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
    got = [a["pred_masks"][0] for a in out["aux_outputs"]] + [out["pred_masks"][0]]
    for i in range(3):
        worst = max(worst, float(np.abs(got[i].cpu().numpy() - c[f"logits{i}"]).max()))
        outs.append(got[i].cpu().numpy())
assert worst <= 1e-3, worst
# a batch of two scenes with different query counts (12 and 20 queries) through forward_backbone + forward_mask
scs = [make_scene(5000, seed=21, batch_index=0), make_scene(7000, seed=22, batch_index=1)]
cl = [make_clicks(scs[0]["labels"], 1, 2, 0, seed=1), make_clicks(scs[1]["labels"], 5, 2, 0, seed=2)]
x = SparseTensor(features=torch.from_numpy(np.concatenate([s["feats"] for s in scs])).cuda(),
                 coordinates=torch.from_numpy(np.concatenate([s["coords"] for s in scs])).cuda())
r = model.forward_backbone(x, raw_coordinates=torch.from_numpy(np.concatenate([s["raw_xyz"] for s in scs])).cuda())
o = model.forward_mask(*r, click_idx=[c[0] for c in cl], click_time_idx=[c[1] for c in cl])
outs += [p.cpu().numpy() for p in o["pred_masks"]]
np.savez(sys.argv[2], *outs)
print("WORST", worst)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for sw in ("0", "1"):
        out = str(tmp_path / f"logits_{sw}.npz")
        env = dict(os.environ, **{switch: sw})
        subprocess.run([sys.executable, "-c", script, root, out], check=True, env=env, timeout=600)
        z = np.load(out)
        res[sw] = [z[k] for k in z.files]
    assert len(res["0"]) == len(res["1"]) == 11
    worst = max(float(np.abs(a - b).max()) for a, b in zip(res["0"], res["1"]))
    scale = max(float(np.abs(a).max()) for a in res["0"])
    print(f"{switch} = 0 vs 1: max |diff| {worst:.2e} on logits of scale {scale:.1f}")
    assert worst <= 1e-4 * max(1.0, scale)
