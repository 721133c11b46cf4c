#!/usr/bin/env python3
"""bench.py -- scenes/s of the AGILE3D hot path on MI355X (BASELINE.json metric).

One *step* = one batch of --batch (default 4) independent scenes through the whole hot path, inputs already
resident in HBM: ONE batched SparseTensor as the reference's collate builds it (batch index in column 0),
coordinate manager build (a3d_scene_create) + forward_backbone over the batch + ONE forward_mask (one decoder
pass per sample).  Every scene is BASELINE.json configs[1]: a seeded synthetic 80k-voxel scene, 10 clicks
(5 objects x 2, no background clicks -> 20 queries), fp32, random-init weights with randomised BatchNorm
statistics.  `value` counts SCENES per second.
Consecutive steps are issued round-robin on --streams HIP streams (default 4 steps in flight), so one step's
latency-bound coarse levels and scene build overlap another's fine-level convolutions; `--batch 1 --streams 1`
is strictly one scene at a time.  N GPUs = N ranks with their own scenes (scene-sharded, no data-path
collective): weak scaling.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  At N=1 it also carries
  roofline      the dominant kernel's achieved fp32-MFMA TFLOP/s = algorithmic FLOPs per launch
                (2 * existing (input,output) pairs * Cin * Cout) / mean launch
                duration, measured live with HIP events on the launch stream in an instrumented
                pass after the timed region;
  cpu_baseline  the CPU oracle (restatement of the reference path the way MinkowskiEngine's CPU
                backend + torch-CPU execute it) timed on this host's cores on the same scene.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# consecutive scenes are issued on several HIP streams; the runtime's default of 4 hardware queues makes
# streams alias (measured: 4 streams 278 scenes/s with 4 queues, 333 with 8), so ask for 8 before HIP starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_HBM_GBS = 8000.0


def algorithmic_flops(entry, pairs):
    """2 * pairs * Cin * Cout for one conv launch."""
    from agile3d_amd import lib as L
    if entry.table == L.OP_CONV3:
        p = pairs["conv3"][entry.level]
    elif entry.table == L.OP_DOWN:
        p = pairs["n"][entry.level]            # every fine voxel contributes once
    elif entry.table == L.OP_UP:
        p = pairs["n"][entry.level - 1]        # every fine voxel receives once
    else:
        p = entry.n_out
    return 2.0 * p * entry.cin * entry.cout


def algorithmic_bytes(entry, pairs):
    """Compulsory HBM bytes of one conv launch: read input + weights once, write output once, read
    the kernel map once: 4 (N_in Cin + N_out Cout + K Cin Cout) + 8 pairs."""
    from agile3d_amd import lib as L
    n_out = entry.n_out
    if entry.table == L.OP_CONV3:
        n_in, p = pairs["n"][entry.level], pairs["conv3"][entry.level]
    elif entry.table == L.OP_DOWN:
        n_in = p = pairs["n"][entry.level]
    elif entry.table == L.OP_UP:
        n_in, p = pairs["n"][entry.level], pairs["n"][entry.level - 1]
    else:
        n_in, p = n_out, 0
    return 4.0 * (n_in * entry.cin + n_out * entry.cout + entry.kernel_volume * entry.cin * entry.cout) + 8.0 * p


def profile_pass(step, scene_pairs, n_steps):
    from agile3d_amd import lib as L
    lib = L.load()
    lib.a3d_profile_read(None, 0)
    lib.a3d_profile_enable(1)
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    lib.a3d_profile_enable(0)
    buf = (L.ProfEntry * 20000)()
    n = lib.a3d_profile_read(buf, 20000)
    agg = {}
    for i in range(n):
        e = buf[i]
        name = L.PROF_NAMES[e.id]
        if e.id == 0:
            ch = next(c for c in (96, 64, 32) if e.cin % c == 0 and 2 * c * e.bn * 4 <= 74 * 1024)
            name = f"k_spconv2<{e.bn},{ch}>"   # BN columns per workgroup, CH input channels per stage (plan_conv)
        elif e.id == L.PROF_DENSE:
            name = f"k_dense<{e.cin // 16},{e.cout // 16}>"
        a = agg.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
        a["ms"] += e.ms
        a["launches"] += 1
        if e.id == 0:
            a["flops"] += algorithmic_flops(e, scene_pairs)
            a["bytes"] += algorithmic_bytes(e, scene_pairs)
        elif e.id == L.PROF_DENSE:                 # [n,cin] x [cin,cout]: read X (+X2/res: not counted), write Y
            a["flops"] += 2.0 * e.n_out * e.cin * e.cout
            a["bytes"] += 4.0 * e.n_out * (e.cin + e.cout) + 4.0 * e.cin * e.cout
    for a in agg.values():
        a["ms_per_step"] = a["ms"] / n_steps
        a["launches_per_step"] = a["launches"] / n_steps
    return agg


def cpu_baseline(sd, sc, ci, ct, budget_s=12.0, max_scenes=3):
    from oracle import backbone as ob, decoder as od
    feats, raw = torch.from_numpy(sc["feats"]), torch.from_numpy(sc["raw_xyz"])
    t0 = time.time()
    done = 0
    while done < max_scenes and (done == 0 or time.time() - t0 < budget_s):
        r = ob.forward_backbone(sd, sc["coords"], feats, raw)
        od.forward_mask(sd, r["pcd_features"], raw, r["pos_enc"], ci, ct)
        done += 1
    dt = time.time() - t0
    return {"value": done / dt, "unit": "scenes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} full scene(s) of the same 80k-voxel/10-click workload through oracle/ "
                      f"(kernel maps + Res16UNet34C + 1 decoder pass) in {dt:.1f} s, torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--voxels", type=int, default=80_000)
    ap.add_argument("--objects", type=int, default=5)
    ap.add_argument("--clicks-per-object", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("A3D_BENCH_BATCH", "4")),
                    help="scenes per step and rank (one batched SparseTensor, as the reference's collate builds)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("A3D_BENCH_STREAMS", "4")),
                    help="scenes in flight per GPU: consecutive steps are issued round-robin on this many HIP streams")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as g
    if rank == 0:
        from agile3d_amd import build as _b
        if _b.needs_build():
            g.build()
    if world > 1:
        dist.barrier()
    from agile3d_amd import SparseTensor, build_model, default_args, lib as L, randomize_bn_stats
    from agile3d_amd.engine import Scene
    from agile3d_amd.synthetic import make_clicks, make_scene

    torch.manual_seed(0)
    model = randomize_bn_stats(build_model(default_args())).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    # one step = one batch of --batch scenes (different seeds) per rank, as ME.utils.batched_coordinates would
    # collate them (datasets/InterMultiObj3DSegDataset.py:129): batch index in column 0, samples contiguous
    scenes = [make_scene(args.voxels, seed=rank * args.batch + b, batch_index=b) for b in range(args.batch)]
    clicks = [make_clicks(s_["labels"], args.objects, args.clicks_per_object, 0, seed=rank * args.batch + b)
              for b, s_ in enumerate(scenes)]
    sc = scenes[0]
    ci, ct = clicks[0]
    cis, cts = [c[0] for c in clicks], [c[1] for c in clicks]
    coords = torch.from_numpy(np.concatenate([s_["coords"] for s_ in scenes])).to(dev)
    feats = torch.from_numpy(np.concatenate([s_["feats"] for s_ in scenes])).to(dev)
    raw = torch.from_numpy(np.concatenate([s_["raw_xyz"] for s_ in scenes])).to(dev)

    def one_scene():       # one STEP: the whole batch through scene build + backbone + one decoder pass per sample
        x = SparseTensor(features=feats, coordinates=coords)
        r = model.forward_backbone(x, raw_coordinates=raw)
        return model.forward_mask(*r, click_idx=cis, click_time_idx=cts)

    out0 = one_scene()                      # packs the weights once, on the default stream
    torch.cuda.synchronize()
    assert torch.isfinite(out0["pred_masks"][0]).all()
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    issued = [0]

    def step():
        if streams is None:
            return one_scene()
        with torch.cuda.stream(streams[issued[0] % len(streams)]):
            issued[0] += 1
            return one_scene()

    from agile3d_amd.sharding import timed_steps
    for _ in range(args.warmup):
        step()
    dt, out = timed_steps(step, args.steps, world, dev)
    assert torch.isfinite(out["pred_masks"][0]).all()

    res = {
        "metric": "scenes/s (80k-voxel, 10 clicks)", "value": world * args.batch * args.steps / dt, "unit": "scenes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"batch of {args.batch} synthetic {len(sc['coords'])}-voxel scenes, "
                               f"{args.objects * args.clicks_per_object} clicks each ({args.objects} objects x "
                               f"{args.clicks_per_object}), scene build + forward_backbone + 1 forward_mask per scene, fp32, eval",
                   "voxels": int(len(sc["coords"])), "queries": args.objects * args.clicks_per_object + 10,
                   "parallelism": f"scene-sharded x{world} (no data-path collective)",
                   "scenes_per_step_per_gpu": args.batch, "global_batch": world * args.batch,
                   "steps_in_flight_per_gpu": args.streams},
    }

    if rank == 0 and world == 1:
        if not args.no_profile:
            scn = Scene(coords)
            pairs = {"n": scn.n, "conv3": []}
            for lvl in range(5):
                npad = (max(scn.n[lvl], 1) + 127) // 128 * 128
                nb = scn.table(lvl, L.TAB_NBR27).reshape(27, npad)
                pairs["conv3"].append(int((nb[:, :scn.n[lvl]] < scn.n[lvl]).sum()))
            agg = profile_pass(one_scene, pairs, max(3, min(10, args.steps)))   # one scene at a time: clean kernel times
            conv = {k: v for k, v in agg.items() if k.startswith("k_spconv") or k.startswith("k_dense")}
            dom = max((k for k in conv if k.startswith("k_spconv")), key=lambda k: conv[k]["ms"])
            d = conv[dom]
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                               "avg_launch_ms": d["ms"] / d["launches"], "launches_per_step": d["launches_per_step"],
                               "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                               "algorithmic_bytes_per_launch": d["bytes"] / d["launches"]}
            tot_flops = sum(v["flops"] for v in conv.values()) / max(3, min(10, args.steps))
            res["kernels_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in sorted(agg.items())}
            res["conv_tflops"] = {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in conv.items()}
            res["scene_algorithmic_gflop_convs"] = round(tot_flops / 1e9, 2)
            res["gpu_ms_per_step_sum_of_kernels"] = round(sum(v["ms_per_step"] for v in agg.values()), 3)
            # the three phases on their own (one step at a time, one stream, wall clock with a device
            # sync around each) and the eval loop's number, decoder passes/s (backbone results reused per round)
            def wall(fn, reps=10):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    out = fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3, out
            t_scene, _ = wall(lambda: Scene(coords))
            t_bb, r = wall(lambda: model.forward_backbone(SparseTensor(features=feats, coordinates=coords),
                                                          raw_coordinates=raw))
            t_dec, _ = wall(lambda: model.forward_mask(*r, click_idx=cis, click_time_idx=cts))
            res["phases_ms_per_step"] = {"scene_build": round(t_scene, 3), "backbone": round(t_bb - t_scene, 3),
                                         "decoder_pass": round(t_dec, 3), "note": "single stream, host wall clock"}
            res["decoder_passes_per_s"] = round(args.batch / (t_dec * 1e-3), 1)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd, sc, ci, ct)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
