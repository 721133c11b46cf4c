"""Worker of tests/test_gpu_distributed.py: one data-parallel rank (two of them share the one GPU of the test box;
gloo backend, rendezvous on 127.0.0.1).  Usage: python dist_gpu_worker.py <mode> <rank> <world> <port> <out_dir>"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def scene_batch(seed, n):
    from agile3d_amd import batched_coordinates
    from agile3d_amd.synthetic import make_scene
    s = make_scene(n, seed=seed)
    return s, (batched_coordinates([s["coords"][:, 1:]]), torch.from_numpy(s["raw_xyz"]), torch.from_numpy(s["feats"]),
               [torch.from_numpy(s["labels"].astype(np.int64))], None, None, [{}], (f"scene{seed:04d}_00",), (0,))


def layer_cases():
    return {"n5000_c64_relu_res": (5000, 64, True, True), "n301_c256": (301, 256, False, False), "n1000_c96_relu": (1000, 96, True, False)}


def layer_data(n, C, with_res):
    g = torch.Generator().manual_seed(n * 7 + C)
    x = torch.randn(n, C, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    res = torch.randn(n, C, generator=g) if with_res else None
    return x, gamma, beta, res, torch.randn(n, C, generator=g)


class CaptureSGD:
    """train_one_step's optimiser interface; keeps the (averaged, unclipped) gradients and the clip coefficient."""

    def __init__(self, model, lr):
        self.params, self.lr, self.grads, self.coef = dict(model.named_parameters()), lr, None, None

    def step(self, grads, coef):
        self.grads, self.coef = {k: g.detach().clone() for k, g in grads.items()}, coef
        with torch.no_grad():
            for k, g in grads.items():
                self.params[k].add_(g.reshape(self.params[k].shape), alpha=-self.lr * coef)


def out_dir_(p):
    os.makedirs(p, exist_ok=True)
    return p


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    if mode == "rccl1":
        # ONE rank on the one GPU over the real backend: nccl (= RCCL) builds its communicator, owns its stream, and the
        # bucketed all-reduce is ordered behind the producing stream -- what the 8-GPU node runs, at world size 1
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from agile3d_amd import build_model, default_args
    dev = torch.device("cuda")
    args = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])
    torch.manual_seed(3)
    model = build_model(args).to(dev)
    if mode == "syncbn":
        from agile3d_amd.engine import Scene
        from agile3d_amd.train_backbone import BackboneTape
        s, batch = scene_batch(60 + rank, 30000 + 3000 * rank)
        model.train()
        tape = BackboneTape(model, Scene(batch[0].to(dev).to(torch.int32).contiguous()), batch[2].to(dev), sync_bn=True)
        grads = tape.backward(tape.output.clone())        # L = |pcd_features|^2 / 2: a coherent gradient (random ones
                                                          # cancel to sums that one flipped ReLU element dominates)
        # one layer on its own: this rank's rows of a fixed matrix (no chain of ReLUs behind it: exact comparison)
        from agile3d_amd import backward as B
        layer = {}
        for name, (n, C, relu, with_res) in layer_cases().items():
            x, gamma, beta, res, dy = layer_data(n, C, with_res)
            lo, hi = (0, n // 3) if rank == 0 else (n // 3, n)
            xs, rs, ds = x[lo:hi].to(dev), (res[lo:hi].to(dev) if res is not None else None), dy[lo:hi].to(dev)
            y, m, r, ng = B.bn_sync_forward(xs, gamma.to(dev), beta.to(dev), 1e-5, rs, relu)
            dx, dg, db, dres = B.bn_sync_backward(xs, y, ds, gamma.to(dev), m, r, ng, relu, with_res)
            layer[name] = {"y": y.cpu(), "dx": dx.cpu(), "dgamma": dg.cpu(), "dbeta": db.cpu(), "n_global": ng,
                           "dres": dres.cpu() if dres is not None else None}
        torch.save({"out": tape.output.cpu(), "grads": {k: v.cpu() for k, v in grads.items()}, "layer": layer,
                    "bn": {k: v.cpu() for k, v in model.state_dict().items() if "running" in k}},
                   os.path.join(out, f"syncbn_{rank}.pt"))
    elif mode == "rccl1":
        from agile3d_amd.criterion import build_mask_criterion
        from agile3d_amd.optim import OverlappedAllReduce, allreduce_mean_, dist_all_reduce
        from agile3d_amd.train_step import train_one_step
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                   # bench.py's ranks_seen probe
        # (a) the overlapped reducer on RCCL: buckets produced by kernels still in flight on the current stream
        g = torch.Generator(device="cpu").manual_seed(5)
        shapes = [(27, 96, 96), (96,), (8, 128, 96), (1, 128), (27, 32, 32), (256,)]
        host = {f"p{i}": torch.randn(sh, generator=g) for i, sh in enumerate(shapes)}
        side = torch.cuda.Stream()
        grads = {}
        with torch.cuda.stream(side):                           # a non-default producing stream
            big = torch.randn(4096, 4096, device=dev)
            for k, v in host.items():
                x = v.to(dev, non_blocking=True)
                big = big @ big.clamp(-1e-3, 1e-3)              # work queued in front of the producer
                grads[k] = x * 2.0 + big[0, 0] * 0.0            # the gradient depends on that queued work
            red = OverlappedAllReduce(bucket_bytes=256 * 1024, expected={k: v.numel() for k, v in host.items()},
                                      single_rank=True)
            assert red.active and red.world == 1 and not red.gloo and red.group is not None
            for k in ["p3", "p1", "p0", "p5", "p4", "p2"]:
                red.add(k, grads[k])
            n_flights = len(red.flights)
            reduced = red.finish({})
        side.synchronize()
        ok = all(torch.equal(reduced[k].cpu(), host[k] * 2.0) for k in host)
        # (b) the real training iteration with the reducer forced on (A3D_DP_SINGLE_RANK=1) vs the plain iteration
        res = {}
        for tag, flag in (("plain", "0"), ("rccl", "1")):
            os.environ["A3D_DP_SINGLE_RANK"] = flag
            torch.manual_seed(3)
            m = build_model(args).to(dev)
            s, batch = scene_batch(70, 2500)
            opt = CaptureSGD(m, 1e-2)
            np.random.seed(11), torch.manual_seed(11), random.seed(11)
            st = train_one_step(m, build_mask_criterion(args), opt, batch, dev, max_norm=0.1)
            res[tag] = ({k: v.detach().cpu() for k, v in m.named_parameters()}, st)
        same = all(torch.equal(res["plain"][0][k], res["rccl"][0][k]) for k in res["plain"][0])
        torch.save({"ranks_seen": int(ones.item()), "bucket_values_ok": ok, "flights": n_flights, "train_step_identical": same,
                    "n_params": len(res["plain"][0]), "loss": res["rccl"][1]["loss"]}, os.path.join(out_dir_(out), "rccl1.pt"))
    elif mode == "dp_step":
        from agile3d_amd.criterion import build_mask_criterion
        from agile3d_amd.train_step import train_one_step
        s, batch = scene_batch(70 + rank, 2500 + 200 * rank)
        opt = CaptureSGD(model, 1e-2)
        np.random.seed(11 + rank), torch.manual_seed(11 + rank), random.seed(11 + rank)     # main.py:97: seed + rank
        st = train_one_step(model, build_mask_criterion(args), opt, batch, dev, max_norm=0.1)
        torch.save({"params": {k: v.detach().cpu() for k, v in model.named_parameters()}, "stats": st,
                    "grads": {k: v.cpu() for k, v in opt.grads.items()}, "coef": opt.coef},
                   os.path.join(out, f"dp_{rank}.pt"))
    elif mode == "dp_epoch":
        # train_one_epoch over three batches with the library's AdamW; A3D_DP_BUCKET_MB (set by the test) makes the
        # overlapped all-reduce issue many small buckets while the backbone backward is still running
        from agile3d_amd.criterion import build_mask_criterion
        from agile3d_amd.optim import AdamW
        from agile3d_amd.train_step import train_one_epoch
        loader = [scene_batch(80 + 10 * i + rank, 2300 + 150 * i + 100 * rank)[1] for i in range(3)]
        opt = AdamW(model.named_parameters(), lr=1e-3, weight_decay=1e-4)
        np.random.seed(21 + rank), torch.manual_seed(21 + rank), random.seed(21 + rank)
        stats, it = train_one_epoch(model, build_mask_criterion(args), loader, opt, dev, epoch=0, max_norm=0.1, log=None)
        torch.save({"params": {k: v.detach().cpu() for k, v in model.named_parameters()}, "stats": stats, "iters": it},
                   os.path.join(out, f"epoch_{rank}.pt"))
    elif mode == "dp8_epoch":
        # BASELINE config 4's collective path at its world size, host side: EIGHT ranks (one ~3 k-voxel scene each) sharing the
        # one GPU over gloo -- three iterations of the real train_one_step through train_one_epoch with the overlapped gradient
        # all-reduce in 0.5 MB buckets (A3D_DP_BUCKET_MB, set by the test) and SyncBN over the eight ranks (A3D_SYNC_BN)
        from agile3d_amd.criterion import build_mask_criterion
        from agile3d_amd.optim import AdamW, OverlappedAllReduce
        from agile3d_amd.train_step import train_one_epoch
        loader = [scene_batch(300 + 10 * i + rank, 2800 + 120 * i + 60 * rank)[1] for i in range(3)]
        opt = AdamW(model.named_parameters(), lr=1e-3, weight_decay=1e-4)
        np.random.seed(21 + rank), torch.manual_seed(21 + rank), random.seed(21 + rank)
        stats, it = train_one_epoch(model, build_mask_criterion(args), loader, opt, dev, epoch=0, max_norm=0.1, log=None)
        # the digest check with ONE rank out of line: every rank must fail with an error (none may enter a collective alone)
        expected = {k: v.numel() for k, v in model.named_parameters()}
        if rank == 5:
            expected[sorted(expected)[17]] += 1
        raised = False
        try:
            OverlappedAllReduce(bucket_bytes=1 << 19, expected=expected)
        except RuntimeError as e:
            raised = "disagree" in str(e)
        ok = OverlappedAllReduce(bucket_bytes=1 << 19, expected={k: v.numel() for k, v in model.named_parameters()})   # and the group still works
        torch.save({"params": {k: v.detach().cpu() for k, v in model.named_parameters()}, "stats": stats, "iters": it,
                    "digest_raised": raised, "reducer_after": bool(ok.active), "bn": {k: v.cpu() for k, v in model.state_dict().items() if "running" in k}},
                   os.path.join(out, f"epoch8_{rank}.pt"))
    elif mode == "syncbn8":
        from agile3d_amd.engine import Scene
        from agile3d_amd.train_backbone import BackboneTape
        s, batch = scene_batch(400 + rank, 2600 + 90 * rank)
        model.train()
        tape = BackboneTape(model, Scene(batch[0].to(dev).to(torch.int32).contiguous()), batch[2].to(dev), sync_bn=True)
        torch.save({"out": tape.output.cpu(), "bn": {k: v.cpu() for k, v in model.state_dict().items() if "running" in k}},
                   os.path.join(out, f"syncbn8_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
