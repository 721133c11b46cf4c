"""-m gpu, world size 2: two data-parallel ranks as two processes sharing the one GPU of the test box (gloo backend;
RCCL refuses two ranks on one device -- the 8-GPU node runs the same code with backend nccl).
  * the real ``train_one_step`` with one scene per rank: the gradients every rank steps with are the MEAN of the two
    single-rank gradients, the clip uses the norm of that mean, both ranks end with identical parameters;
  * SyncBN (``BackboneTape(sync_bn=True)``): one scene on each of two ranks == both scenes in one batch on one rank
    (the reference normalises over all rows of the batch on one device, models/modules/common.py:20-22)."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_gpu_worker as W  # noqa: E402

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_world(mode, out_dir, world=2):
    port = str(29500 + (os.getpid() + hash(mode)) % 2000)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), mode, str(r), str(world), port,
                               str(out_dir)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-3000:]


def test_dp_train_step_two_ranks(tmp_path):
    from agile3d_amd import build_model, default_args
    from agile3d_amd.criterion import build_mask_criterion
    from agile3d_amd.optim import clip_grad_norm_
    from agile3d_amd.train_step import train_one_step
    _run_world("dp_step", tmp_path)
    r0 = torch.load(tmp_path / "dp_0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "dp_1.pt", weights_only=False)
    for k in r0["params"]:                                   # both ranks end with identical parameters
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    for k in r0["grads"]:                                    # ... because they stepped with identical gradients
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k
    assert r0["coef"] == r1["coef"]
    # the same two iterations on ONE rank each (no process group): mean of their gradients, clip on that mean, SGD
    dev = torch.device("cuda")
    args = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])
    single = []
    for rank in range(2):
        torch.manual_seed(3)
        model = build_model(args).to(dev)
        _, batch = W.scene_batch(70 + rank, 2500 + 200 * rank)
        opt = W.CaptureSGD(model, 0.0)                      # lr 0: gradients only
        np.random.seed(11 + rank), torch.manual_seed(11 + rank), random.seed(11 + rank)
        st = train_one_step(model, build_mask_criterion(args), opt, batch, dev, max_norm=0.1)
        single.append((opt.grads, st))
        assert st["clicks"] == [r0, r1][rank]["stats"]["clicks"]
    mean = {k: (single[0][0][k] + single[1][0][k]) / 2 for k in single[0][0]}
    worst = 0.0
    for k, g in mean.items():
        d = (r0["grads"][k].to(dev) - g).abs().max().item()
        worst = max(worst, d / max(1e-8, g.abs().max().item()))
    print(f"averaged gradients vs mean of the two single-rank runs: worst relative difference {worst:.2e}")
    assert worst <= 1e-5
    norm, coef = clip_grad_norm_({k: v.clone() for k, v in mean.items()}, 0.1)
    assert abs(coef - r0["coef"]) <= 1e-6 * coef
    torch.manual_seed(3)
    ref = build_model(args).to(dev)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            p.add_(mean[k].reshape(p.shape), alpha=-1e-2 * coef)
            assert torch.allclose(p.cpu(), r0["params"][k], rtol=1e-6, atol=1e-8), k


def test_syncbn_two_ranks_equal_one_rank_batch_of_two(tmp_path):
    from agile3d_amd import batched_coordinates, build_model, default_args
    from agile3d_amd.engine import Scene
    from agile3d_amd.train_backbone import BackboneTape
    _run_world("syncbn", tmp_path)
    r = [torch.load(tmp_path / f"syncbn_{i}.pt", weights_only=False) for i in range(2)]
    dev = torch.device("cuda")
    args = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])
    torch.manual_seed(3)
    model = build_model(args).to(dev).train()
    scenes = [W.scene_batch(60 + i, 30000 + 3000 * i)[0] for i in range(2)]
    coords = batched_coordinates([s["coords"][:, 1:] for s in scenes]).to(dev).to(torch.int32).contiguous()
    feats = torch.from_numpy(np.concatenate([s["feats"] for s in scenes])).to(dev)
    tape = BackboneTape(model, Scene(coords), feats, sync_bn=False)          # one rank, both scenes in the batch
    n0 = len(scenes[0]["coords"])
    grads = tape.backward(tape.output.clone())                               # L = |pcd_features|^2 / 2 on both sides
    out = tape.output.cpu()
    e0 = (out[:n0] - r[0]["out"]).abs().max().item()
    e1 = (out[n0:] - r[1]["out"]).abs().max().item()
    print(f"forward: one rank x 2 scenes vs two ranks x 1 scene (SyncBN): {e0:.2e} {e1:.2e} (scale {out.abs().max():.2f})")
    assert max(e0, e1) <= 1e-5 * max(1.0, out.abs().max().item())
    worst, rows = 0.0, []
    for k, g in grads.items():
        both = (r[0]["grads"][k] + r[1]["grads"][k]).to(dev)                # local sums of the two ranks
        e = (g - both).abs().max().item() / max(1e-6, g.abs().max().item())
        rows.append((e, k, g.abs().max().item()))
        worst = max(worst, e)
    rows.sort(reverse=True)
    from agile3d_amd import backward as B
    for name, (n, C, relu, with_res) in W.layer_cases().items():
        x, gamma, beta, res, dy = [t.to(dev) if t is not None else None for t in W.layer_data(n, C, with_res)]
        y, m, rs = B.bn_train_forward(x, gamma, beta, 1e-5, res, relu)
        dx, dg, db, dres = B.bn_train_backward(x, y, dy, gamma, m, rs, relu, with_res)
        cut = n // 3
        got = {k: torch.cat([r[0]["layer"][name][k], r[1]["layer"][name][k]]).to(dev) for k in ("y", "dx")}
        assert r[0]["layer"][name]["n_global"] == n
        assert (got["y"] - y).abs().max().item() <= 1e-5 * max(1.0, y.abs().max().item()), name
        assert (got["dx"] - dx).abs().max().item() <= 2e-5 * max(1.0, dx.abs().max().item()), name
        for k, ref in (("dgamma", dg), ("dbeta", db)):            # local sums of the ranks add up to the batch's
            both = (r[0]["layer"][name][k] + r[1]["layer"][name][k]).to(dev)
            assert (both - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), (name, k)
        if with_res:
            gr = torch.cat([r[0]["layer"][name]["dres"], r[1]["layer"][name]["dres"]]).to(dev)
            assert torch.equal(gr, dres), name
        assert cut > 0
    med = float(np.median([e for e, _, _ in rows]))
    print("largest differences:", [(f"{e:.1e}", k, f"{sc:.1e}") for e, k, sc in rows[:4]])
    print(f"gradients of the whole backbone: median relative difference {med:.2e}, worst {worst:.2e}")
    # 62 BatchNorm + ReLU layers deep, the elements whose pre-activation changes sign between the two (4e-6 apart)
    # forward passes move every sum behind them: the same 1-4 % the fp32-vs-fp64 comparison of the backbone shows
    # without shared ReLU masks (test_gpu_backward.py); the layer on its own (above) is compared exactly
    assert med <= 2e-2 and worst <= 0.2
    sd = model.state_dict()
    for k, v in r[0]["bn"].items():
        assert torch.allclose(v, r[1]["bn"][k])                              # same statistics on both ranks
        assert torch.allclose(sd[k].cpu().float(), v.float(), rtol=1e-4, atol=1e-6), k


def test_dp_epoch_overlapped_allreduce_keeps_ranks_identical(tmp_path, monkeypatch):
    """train_one_epoch (three iterations, AdamW, clip) on two ranks with the gradient all-reduce overlapped with the
    backbone backward in many small buckets: after the epoch both ranks hold bit-identical parameters (they stepped with
    identical averaged gradients every time), they moved, and both saw the same averaged-loss bookkeeping shape."""
    monkeypatch.setenv("A3D_DP_BUCKET_MB", "0.5")
    _run_world("dp_epoch", tmp_path)
    r0 = torch.load(tmp_path / "epoch_0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "epoch_1.pt", weights_only=False)
    assert r0["iters"] == r1["iters"] == 3
    from agile3d_amd import build_model, default_args
    torch.manual_seed(3)
    init = dict(build_model(default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])).named_parameters())
    moved = 0
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
        moved += int(not torch.equal(r0["params"][k], init[k].detach()))
    assert moved >= 260, moved
    assert set(r0["stats"]) == set(r1["stats"]) and np.isfinite(r0["stats"]["loss"])


def test_bench_multi_rank_code_path_on_one_gpu():
    """The literal `python bench.py --gpus 2`: bench.py launches its own two ranks (torch.distributed.run), which on this
    one-GPU box share the device and talk over gloo -- barriers, MAX over ranks, ranks_seen, one JSON line from rank 0.
    Checks the plumbing the driver's 2/4/8-GPU runs go through, not a number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "A3D_BENCH_ONE_GPU")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reps", "1",
           "--batch", "4", "--no-profile"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["global_batch"] == 8
    assert d["ranks_seen"] == 2 and len(d["ms_per_step_per_rank"]) == 2
    if torch.cuda.device_count() < 2:
        assert d["ranks_share_gpus"] is True and d["config"]["backend"] == "gloo"


def test_eight_ranks_share_the_one_gpu_host_side():
    """`A3D_BENCH_ONE_GPU=1 python bench.py --gpus 8 --batch 2`: EIGHT launch loops (one per rank, four streams each)
    on one host under the container's CPU quota, sharing the one GPU over gloo -- the same total device load as the
    single-rank headline (16 scenes in flight per step).  A plumbing + host-contention check of what an 8-GPU node's host
    side has to sustain (hostcpu.py: eight Python launch threads, torch pools capped to quota / 16 each), NOT a scaling
    number: every rank progresses at the same pace, the kernel does not throttle the container during the timed region
    (cpu.stat nr_throttled, printed in the line), and the ranks together are not much slower than one rank driving the
    same load."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["A3D_BENCH_ONE_GPU"] = "1"

    def run(extra):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--reps", "3", "--no-profile",
               "--steps-only", "--no-train", "--no-cpu-baseline"] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=root)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        return json.loads(lines[0])
    d8 = run(["--gpus", "8", "--batch", "2"])
    per = d8["ms_per_step_per_rank"]
    cfs = d8["config"]["host_cpu"]["cfs_during_timed_region"]
    print("8 ranks on one GPU:", round(d8["value"], 1), "scenes/s; per rank ms/step", per, "cfs", cfs,
          "host", d8["config"]["host_cpu"])
    assert d8["n_gpus"] == 8 and d8["ranks_seen"] == 8 and len(per) == 8 and d8["ranks_share_gpus"] is True
    assert d8["config"]["global_batch"] == 16
    assert max(per) <= 1.3 * min(per), per                     # nobody is starved (observed spread: a few per cent)
    if cfs and "nr_throttled" in cfs and "nr_periods" in cfs:
        assert cfs["nr_throttled"] <= max(2, cfs["nr_periods"] // 20), cfs      # the container is not being frozen
    env.pop("A3D_BENCH_ONE_GPU")
    d1 = run(["--gpus", "1", "--batch", "16"])
    print("1 rank, same load:", round(d1["value"], 1), "scenes/s")
    assert d8["value"] >= 0.7 * d1["value"], (d8["value"], d1["value"])


def test_rccl_backend_single_rank_on_the_one_gpu(tmp_path):
    """The `nccl` (= RCCL) branch that the 8-GPU node runs, at world size 1 on this box's one GPU (RCCL refuses two ranks
    per device, so every multi-rank test above is gloo): init_process_group("nccl", device_id=...), the all-reduce of
    ones that bench.py prints as ranks_seen, OverlappedAllReduce's asynchronous buckets on the communicator's own stream
    ordered behind a non-default producing stream (values exact), and the real train_one_step with the reducer forced on
    ending with bit-identical parameters to the run without it (a mean over one rank is the identity)."""
    _run_world("rccl1", tmp_path, world=1)
    r = torch.load(tmp_path / "rccl1.pt", weights_only=False)
    assert r["ranks_seen"] == 1
    assert r["bucket_values_ok"] and r["flights"] >= 2
    assert r["train_step_identical"] and r["n_params"] >= 260 and np.isfinite(r["loss"])


def test_bench_rccl_branch_at_world_size_one():
    """bench.py's own `nccl` branch (init with device_id, ranks_seen, barriers, MAX over ranks through the backend) under
    torch.distributed.run with ONE rank: A3D_BENCH_DIST_AT_1=1 keeps the distributed path on at world size 1."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "A3D_BENCH_ONE_GPU")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["A3D_BENCH_DIST_AT_1"] = "1"
    port = str(31500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--reps", "1",
           "--batch", "4", "--no-profile", "--no-cpu-baseline", "--steps-only"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["config"]["backend"] == "nccl (RCCL)"
    assert d["value"] > 0 and len(d["ms_per_step_per_rank"]) == 1


def test_eight_rank_dp_training_on_the_one_gpu(tmp_path, monkeypatch):
    """BASELINE.json config 4 (bs 8, one scene per rank, gradient all-reduce only; /root/reference main.py:115-127,
    engine.py:26-179) at ITS world size, host side: eight gloo ranks share the one GPU.
      * three iterations of the real train_one_step (train_one_epoch, AdamW + clip) with the gradient all-reduce overlapped
        with the backward in 0.5 MB buckets on its own communicator and SyncBN over the eight ranks: all eight ranks end with
        BIT-IDENTICAL parameters and BatchNorm statistics, the parameters moved;
      * the digest check with one rank's gradient list perturbed: every rank fails with an error, nobody hangs, and the
        process group is usable afterwards;
      * one iteration without SyncBN: the gradients every rank steps with are the MEAN of the eight single-rank gradients;
      * SyncBN's forward over eight ranks == one rank with the eight scenes in one batch.
    No RCCL here (it refuses two ranks per device) and no scaling number: the eight-GPU node runs the same code over nccl."""
    from agile3d_amd import batched_coordinates, build_model, default_args
    from agile3d_amd.criterion import build_mask_criterion
    from agile3d_amd.engine import Scene
    from agile3d_amd.train_backbone import BackboneTape
    from agile3d_amd.train_step import train_one_step
    args = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])
    dev = torch.device("cuda")
    monkeypatch.setenv("A3D_DP_BUCKET_MB", "0.5")
    monkeypatch.setenv("A3D_SYNC_BN", "1")
    monkeypatch.setenv("A3D_HOST_THREADS", "1")
    _run_world("dp8_epoch", tmp_path, world=8)
    r = [torch.load(tmp_path / f"epoch8_{i}.pt", weights_only=False) for i in range(8)]
    torch.manual_seed(3)
    init = dict(build_model(args).named_parameters())
    moved = 0
    for k in r[0]["params"]:
        for i in range(1, 8):
            assert torch.equal(r[0]["params"][k], r[i]["params"][k]), (k, i)
        moved += int(not torch.equal(r[0]["params"][k], init[k].detach()))
    for k in r[0]["bn"]:
        for i in range(1, 8):
            assert torch.equal(r[0]["bn"][k], r[i]["bn"][k]), (k, i)
    assert moved >= 260, moved
    assert all(x["iters"] == 3 and x["digest_raised"] and x["reducer_after"] for x in r), [(x["iters"], x["digest_raised"]) for x in r]
    assert all(np.isfinite(x["stats"]["loss"]) for x in r)
    # ---- one iteration without SyncBN: averaged gradients == mean of the eight single-rank gradients
    monkeypatch.setenv("A3D_SYNC_BN", "0")
    _run_world("dp_step", tmp_path, world=8)
    d = [torch.load(tmp_path / f"dp_{i}.pt", weights_only=False) for i in range(8)]
    for i in range(1, 8):
        assert d[i]["coef"] == d[0]["coef"]
        for k in d[0]["grads"]:
            assert torch.equal(d[0]["grads"][k], d[i]["grads"][k]), (k, i)
    mean = None
    for rank in range(8):
        torch.manual_seed(3)
        model = build_model(args).to(dev)
        _, batch = W.scene_batch(70 + rank, 2500 + 200 * rank)
        opt = W.CaptureSGD(model, 0.0)
        np.random.seed(11 + rank), torch.manual_seed(11 + rank), random.seed(11 + rank)
        st = train_one_step(model, build_mask_criterion(args), opt, batch, dev, max_norm=0.1)
        assert st["clicks"] == d[rank]["stats"]["clicks"]
        mean = {k: g.double() / 8 for k, g in opt.grads.items()} if mean is None else {k: mean[k] + g.double() / 8 for k, g in opt.grads.items()}
    worst = max((d[0]["grads"][k].to(dev).double() - g).abs().max().item() / max(1e-8, g.abs().max().item()) for k, g in mean.items())
    print(f"eight ranks: averaged gradients vs mean of the eight single-rank runs, worst relative difference {worst:.2e}")
    assert worst <= 2e-5
    # ---- SyncBN forward: eight ranks x one scene == one rank x eight scenes
    _run_world("syncbn8", tmp_path, world=8)
    s8 = [torch.load(tmp_path / f"syncbn8_{i}.pt", weights_only=False) for i in range(8)]
    scenes = [W.scene_batch(400 + i, 2600 + 90 * i)[0] for i in range(8)]
    torch.manual_seed(3)
    model = build_model(args).to(dev).train()
    coords = batched_coordinates([s_["coords"][:, 1:] for s_ in scenes]).to(dev).to(torch.int32).contiguous()
    feats = torch.from_numpy(np.concatenate([s_["feats"] for s_ in scenes])).to(dev)
    out = BackboneTape(model, Scene(coords), feats, sync_bn=False).output.cpu()
    row = 0
    for i, s_ in enumerate(scenes):
        n = len(s_["coords"])
        e = (out[row:row + n] - s8[i]["out"]).abs().max().item()
        assert e <= 2e-5 * max(1.0, out.abs().max().item()), (i, e)
        row += n
    sd = model.state_dict()
    for k, v in s8[0]["bn"].items():
        assert all(torch.equal(v, s8[i]["bn"][k]) for i in range(1, 8)), k
        assert torch.allclose(sd[k].cpu().float(), v.float(), rtol=1e-4, atol=1e-6), k
