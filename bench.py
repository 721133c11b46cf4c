#!/usr/bin/env python3
"""bench.py -- scenes/s of the AGILE3D hot path on MI355X (BASELINE.json metric).

One *step* = one batch of --batch (default 16) independent scenes through the whole hot path, inputs already
resident in HBM: ONE batched SparseTensor as the reference's collate builds it (batch index in column 0),
coordinate manager build (a3d_scene_create) + forward_backbone over the batch + ONE forward_mask (one decoder
pass per sample).  Every scene is BASELINE.json configs[1]: a seeded synthetic 80k-voxel scene, 10 clicks
(5 objects x 2, no background clicks -> 20 queries), fp32, random-init weights with randomised BatchNorm
statistics.  `value` counts SCENES per second.
Consecutive steps are issued round-robin on --streams HIP streams (default 4 steps in flight), so one step's
latency-bound coarse levels and scene build overlap another's fine-level convolutions; `--batch 1 --streams 1`
is strictly one scene at a time.  N GPUs = N ranks with their own scenes (scene-sharded, no data-path
collective): weak scaling.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launches its own N ranks through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The timed region (exactly --steps steps between barrier + device sync, MAX over ranks) is repeated --reps
times and the MEDIAN repetition is reported (all repetitions are listed in `timed_region`).

Rank 0 prints ONE JSON line.  At N=1 it also carries
  roofline      the dominant kernel's achieved fp32-MFMA TFLOP/s = algorithmic FLOPs per launch
                (2 * existing (input,output) pairs * Cin * Cout) / launch duration,
                measured live with HIP events on the launch stream in an instrumented pass after
                the timed region: per launch position the median of 10 steps, a kernel's time = the
                sum over its launches, dominant = the largest sum; compared with the committed
                rocprofv3 average (profiles/kernel_avg_us.json);
  pipeline_frac algorithmic FLOPs of one whole step / ms_per_step / the same peak;
  latency_ms_per_scene   SURVEY 8(d)'s batch 1 x 1 stream protocol (median of 50 after 10 warm-ups);
  cpu_baseline  the CPU oracle (restatement of the reference path the way MinkowskiEngine's CPU
                backend + torch-CPU execute it) timed on this host's cores on the same scene
                (median of 3), and `parity_vs_oracle` / `max_abs_diff`: the GPU's output for that
                scene against the oracle's.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: bf16 dense MFMA peak (the emulated-fp32 line's roofline divides by this / 6)
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
    # a fused residual projection (a3d_op.proj_cin, reported in bits 8-19 of the kernel-volume field): one more product per
    # row; a fused head (a3d_op.head_cout, bits 20+): one more GEMM on the finished rows
    cin2, head = (entry.kernel_volume >> 8) & 0xfff, entry.kernel_volume >> 20
    return 2.0 * p * entry.cin * entry.cout + 2.0 * entry.n_out * cin2 * entry.cout + 2.0 * entry.n_out * entry.cout * head


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
    kv, cin2, head = entry.kernel_volume & 0xff, (entry.kernel_volume >> 8) & 0xfff, entry.kernel_volume >> 20
    return 4.0 * (n_in * entry.cin + n_out * (entry.cout + cin2 + head) + (kv * entry.cin + cin2 + head) * entry.cout) + 8.0 * p


def workload_key(voxels, batch, queries):
    """Key of a workload in profiles/pmc_summary.json and profiles/kernel_avg_us.json."""
    return f"{int(float(f'{voxels:.2g}')) // 1000}k_b{batch}_q{queries}"      # two significant digits: 79 736 -> 80k, 301 193 -> 300k


def profile_file(name, wkey):
    """The per-kernel table of workload `wkey` in a committed profiles/ file ({} when the file has none for it)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get("workloads", {}).get(wkey, {}) or {}
    except Exception:
        return {}


def hbm_traffic_frac(traffic, avg_launch_ms):
    """Measured fabric bytes per launch / launch duration / peak; None without counters of THIS workload, and never a
    fraction above 1 (that would mean the counters are of something else)."""
    if not traffic:
        return None
    f = traffic / (avg_launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
    return f if f <= 1.0 else None


def kernel_name(e):
    from agile3d_amd import lib as L
    name = L.PROF_NAMES[e.id]
    if e.id == 0 and e.ksplit == 0:
        name = f"k_conv_wl<{e.cin // 16},{e.cout // 16}>"   # whole weight set resident in LDS (spconv.hip)
    elif e.id == 0 and e.bn >= 1000:
        name = f"k_conv_deep<{e.bn - 1000},{e.ksplit}>"   # the small-level kernel (spconv.hip: static parts, LDS-DMA ring)
    elif e.id == 0:
        name = f"k_conv_sk<{e.bn},{e.ksplit}>"   # BN columns per workgroup, CH input channels per stage (plan_sk; the
                                                  # profile entry's last integer field carries CH)
        if e.kernel_volume >> 20:                 # the fused-head instantiation (k_conv_sk<..., HEAD = true>): its own kernel
            name += "+head"
    elif e.id == L.PROF_DENSE:
        name = f"k_dense<{e.cin // 16},{e.cout // 16}>"
    return name


def profile_pass(step, scene_pairs, n_steps):
    """HIP-event durations of every launch of `n_steps` steps (events on the launch stream, a3d_profile_*).
    Every step issues the same launch sequence, so launch i of a step has n_steps samples: its duration is their
    MEDIAN (one bad event pair -- a preempted queue, a clock ramp -- cannot flip the dominant kernel), and a
    kernel's time per step is the sum of its launches' medians."""
    from agile3d_amd import lib as L
    lib = L.load()
    lib.a3d_profile_read(None, 0)
    lib.a3d_profile_enable(1)
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    lib.a3d_profile_enable(0)
    cap = 4000 * n_steps
    buf = (L.ProfEntry * cap)()
    n = lib.a3d_profile_read(buf, cap)
    if n >= cap or n % n_steps:
        raise RuntimeError(f"profile_pass: {n} instrumented launches in {n_steps} steps (buffer holds {cap}; every step must "
                           f"launch the same sequence) -- raise the cap or check that nothing else launched in between")
    per = n // n_steps
    agg = {}
    for i in range(per):
        e = buf[i]
        for r in range(1, n_steps):
            assert buf[i + r * per].id == e.id and buf[i + r * per].n_out == e.n_out, "launch sequences differ"
        ms = float(np.median([buf[i + r * per].ms for r in range(n_steps)]))
        name = kernel_name(e)
        a = agg.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
        a["ms"] += ms
        a["launches"] += 1
        if e.id == 0:
            a["flops"] += algorithmic_flops(e, scene_pairs)
            a["bytes"] += algorithmic_bytes(e, scene_pairs)
        elif e.id == L.PROF_DENSE:                 # [n,cin] x [cin,cout]: read X (+X2/res: not counted), write Y
            a["flops"] += 2.0 * e.n_out * e.cin * e.cout
            a["bytes"] += 4.0 * e.n_out * (e.cin + e.cout) + 4.0 * e.cin * e.cout
    for a in agg.values():                         # everything is per step now
        a["ms_per_step"] = a["ms"]
        a["launches_per_step"] = a["launches"]
    return agg


def cpu_baseline(sd, sc, ci, ct, gpu_logits=None, gpu_feats=None, samples=5):
    """The CPU oracle timed on this host's cores on scene 0 of the GPU workload: one probe sample at 8 and at 32
    threads (torch's default of one thread per core is slower on a 128-core host: the per-offset GEMMs are small),
    then `samples` scenes at the better setting with the coordinate manager's kernel maps BUILT ONCE outside the timed
    part (MinkowskiEngine caches them in its coordinate manager too); `value` is the median of those.  The time of a
    scene INCLUDING the (pure numpy) kernel-map construction is reported next to it.  The oracle's output is not
    thrown away: it is compared with what the GPU produced for the same scene (max_abs_diff)."""
    from oracle import backbone as ob, decoder as od
    feats, raw = torch.from_numpy(sc["feats"]), torch.from_numpy(sc["raw_xyz"])

    def one(levels=None):
        t0 = time.time()
        r = ob.forward_backbone(sd, sc["coords"], feats, raw, levels=levels)
        lg = od.forward_mask(sd, r["pcd_features"], raw, r["pos_enc"], ci, ct)
        return time.time() - t0, r, lg

    from agile3d_amd.hostcpu import cpu_quota
    ncpu = max(1, int(cpu_quota()))               # what the container lets this process use, not the CPUs it can see
    threads_before = torch.get_num_threads()
    probes = {}
    levels = None
    for nt in sorted({min(8, ncpu), min(32, ncpu)}):
        torch.set_num_threads(nt)
        dt0, r0, _ = one()                        # builds its own kernel maps: the all-inclusive time
        probes[nt] = dt0
        levels = r0["levels"]
    best = min(probes, key=probes.get)
    torch.set_num_threads(best)
    times = []
    for _ in range(samples):
        dt, r, lg = one(levels)
        times.append(dt)
    med = float(np.median(times))
    res = {"value": 1.0 / med, "unit": "scenes/s", "cores": best, "kind": "port",
           "sample": f"median of {samples} full scenes of the same {len(sc['coords'])}-voxel/"
                     f"{sum(len(v) for v in ci.values())}-click workload through oracle/ (Res16UNet34C + 1 decoder pass; "
                     f"kernel maps built once outside the timed part), {best} threads of the {ncpu} CPUs of this container's quota ({os.cpu_count()} visible; probes incl. "
                     f"kernel-map construction: " + ", ".join(f"{k} thr {v:.1f} s" for k, v in probes.items())
                     + f"), samples {[round(t, 2) for t in times]} s, torch {torch.__version__} CPU",
           "seconds_per_scene": med, "seconds_per_scene_incl_kernel_maps": float(probes[best]),
           "value_incl_kernel_maps": 1.0 / float(probes[best]), "samples": samples}
    diff = None
    if gpu_logits is not None:
        diff = {"logits_max_abs_diff": float((gpu_logits.cpu() - lg[-1]).abs().max()),
                "logits_scale": float(lg[-1].abs().max()),
                "pcd_features_max_abs_diff": float((gpu_feats.cpu() - r["pcd_features"]).abs().max()),
                "pcd_features_scale": float(r["pcd_features"].abs().max()),
                "note": "GPU output of scene 0 of the timed workload vs the CPU oracle on the same inputs (bar 1e-3)"}
    cpu_baseline.last_logits = lg[-1]      # the oracle's logits of that scene (the emulated-fp32 pass is compared with them too)
    torch.set_num_threads(threads_before)         # back to what was in force before this leg (hostcpu.cap_host_threads's policy)
    return res, diff


def _size64(x64, member):
    """float64 error size of a cluster (largest distance of a member to the nearest non-member) and the row attaining it"""
    rows = torch.nonzero(member).flatten()
    best, arg = -1.0, -1
    other = x64[~member]
    for s0 in range(0, len(rows), 1024):                    # chunks: [1024 x N] float64 at a time
        d = torch.cdist(x64[rows[s0:s0 + 1024]], other).min(1).values
        k = int(torch.argmax(d))
        if float(d[k]) > best:
            best, arg = float(d[k]), int(rows[s0 + k])
    return best, arg


def oracle_protocol_synced(model, sd, items, gpu_log, objects, max_clicks, dev, logit_tol=1e-4):
    """The CPU oracle's interactive protocol (oracle.clicks over oracle backbone + decoder, utils/seg.py:173-226 +
    eval_multi_obj.py:112-160) run NEXT TO the GPU product's log, round by round, with the same `random` stream.  Every
    round is compared: the oracle holds the clicks the GPU run holds, computes ITS labels / IoU / next clicks, and
    whatever differs must be PROVEN one of
      * "logit tie": labels differ only at points whose two best logits are within `logit_tol` on both sides (the GPU's
        logits differ from the oracle's by ~1e-5; eval_multi_obj.py:126 takes the argmax at face value);
      * "mask tie in layer l": the same coin flip one or two decoder layers earlier -- the arg-max labels of layer l's mask
        logits pick the points the NEXT layer's click-to-scene attention may see (agile3d.py:367-380), so one flipped point
        moves the final logits by ~0.1 although every layer's arithmetic agrees to 1e-5 (the float64 oracle flips too);
      * "distance tie" inside a cluster: both sides' rows are the arg-max of the outside distance IN THEIR OWN ARITHMETIC
        (the reference / oracle: torch.cdist, the matmul formula, error ~sqrt(eps)|x| near zero, utils/seg.py:157-171;
        clicks.hip: the exact difference expression) and their float64 distances differ by no more than cdist's own error
        on those two rows (+ 2 ulp);
      * "rank tie" between the two largest clusters: each side's top cluster is the larger one in its own arithmetic and
        the two clusters' float64 sizes differ by no more than cdist's own error on those two sizes (+ 2 ulp);
      * "follows a logit tie": the clicks were picked from labels that differed in the (proven) logit tie of that round;
    anything else counts as "unexplained" (a bug).  After a differing click the oracle CONTINUES FROM THE GPU's CLICKS, so
    no round is left uncompared.  Returns (oracle_log, oracle_bb, forks)."""
    import random
    from agile3d_amd import SparseTensor
    from agile3d_amd import clicks as pc
    from oracle import backbone as ob, clicks as oc, decoder as od
    per_scene = len(gpu_log) // len(items)
    forks = {"scenes": [], "unexplained": 0, "compared_rounds": 0, "identical_rounds": 0, "click_forks": 0,
             "logit_tol": logit_tol}
    oracle_log, oracle_bb = [], []

    def gpu_logits(i, ci, ct):
        sc = items[i]["scene"]
        x = SparseTensor(features=torch.from_numpy(sc["feats"]), coordinates=torch.from_numpy(sc["coords"]), device=dev)
        bo = model.forward_backbone(x, raw_coordinates=torch.from_numpy(sc["raw_xyz"]).to(dev))
        out_ = model.forward_mask(*bo, click_idx=[ci], click_time_idx=[ct])
        return [a_["pred_masks"][0].cpu() for a_ in out_["aux_outputs"]] + [out_["pred_masks"][0].cpu()]

    for i, it in enumerate(items):
        g = gpu_log[i * per_scene:(i + 1) * per_scene]
        sc, lab = it["scene"], torch.from_numpy(it["labels"])
        xyz = torch.from_numpy(sc["raw_xyz"])
        x64 = xyz.double()
        rb = ob.forward_backbone(sd, sc["coords"], torch.from_numpy(sc["feats"]), xyz)
        oracle_bb.append(rb)
        rec = {"scene": it["name"], "rounds": len(g), "events": []}
        ci = {str(k): [] for k in range(objects + 1)}
        ct = {str(k): [] for k in range(objects + 1)}
        n_clicks = 0
        for j, a in enumerate(g):
            assert a[0] == n_clicks and a[2] == ci and a[4] == ct        # the oracle holds the GPU run's clicks
            lo = None
            if n_clicks == 0:
                pred = torch.zeros(lab.shape)
            else:
                lo_all = od.forward_mask(sd, rb["pcd_features"], xyz, rb["pos_enc"], ci, ct)
                lo = lo_all[-1]
                pred = lo.argmax(-1)
                for obj_id, rows in ci.items():
                    pred[rows] = int(obj_id)
            iou, _ = oc.mean_iou_scene(pred, lab)
            pred = pred.long()
            oracle_log.append((n_clicks, float(np.float32(iou)), {k: list(v) for k, v in ci.items()}, pred.clone(),
                               {k: list(v) for k, v in ct.items()}))
            forks["compared_rounds"] += 1
            same_pred = torch.equal(a[3], pred)
            label_tie = False
            if same_pred and abs(a[1] - float(iou)) <= 1e-6:
                forks["identical_rounds"] += 1
            else:
                ev = {"round": j, "iou_gpu": a[1], "iou_oracle": float(iou)}
                if same_pred:
                    ev.update({"kind": "unexplained", "proven": False})      # same labels, different IoU
                else:
                    # the FIRST decoder layer whose arg-max labels differ: the last one is the prediction itself, an earlier
                    # one feeds the next layer's attention mask (agile3d.py:367-380: a discrete choice, so one flipped point
                    # moves the later logits by far more than rounding) -- either way the flip must sit where the two best
                    # logits of that layer are within logit_tol on BOTH sides, with that layer's logits in agreement
                    lg_all = gpu_logits(i, a[2], a[4])
                    first = next((l_ for l_ in range(len(lo_all)) if not torch.equal(lg_all[l_].argmax(-1), lo_all[l_].argmax(-1))),
                                 len(lo_all) - 1)
                    lg_f, lo_f = lg_all[first], lo_all[first]
                    rows = torch.nonzero(lg_f.argmax(-1) != lo_f.argmax(-1)).flatten()
                    if rows.numel() == 0:       # the labels differ only through the clicked rows' overwrite: cannot happen with equal clicks
                        rows = torch.nonzero(a[3] != pred).flatten()
                    mg = lg_f[rows].topk(2, dim=-1).values
                    mo = lo_f[rows].topk(2, dim=-1).values
                    margin = torch.maximum(mg[:, 0] - mg[:, 1], mo[:, 0] - mo[:, 1])
                    agree = float((lg_f - lo_f).abs().max())
                    label_tie = float(margin.max()) <= logit_tol and agree <= logit_tol
                    last = first == len(lo_all) - 1
                    ev.update({"points_with_different_labels": int((a[3] != pred).sum()), "first_layer_with_different_labels": first,
                               "points_flipped_in_that_layer": int(rows.numel()), "largest_top2_margin_there": float(margin.max()),
                               "max_logit_diff_gpu_vs_oracle_there": agree,
                               "max_logit_diff_gpu_vs_oracle_final": float((lg_all[-1] - lo).abs().max()),
                               "kind": ("logit tie" if last else f"mask tie in layer {first}") if label_tie else "unexplained",
                               "proven": bool(label_tie)})
                forks["unexplained"] += 0 if ev["proven"] else 1
                rec["events"].append(ev)
            # ---- the oracle's next clicks from ITS labels, against the GPU run's
            new_clicks, _, _, new_time = oc.get_simulated_clicks(pred, lab, xyz, n_clicks, training=False)
            nci, nct = {k: list(v) for k, v in ci.items()}, {k: list(v) for k, v in ct.items()}
            if new_clicks is not None:
                nci, nct = oc.extend_clicks(nci, nct, new_clicks, new_time)
            if j + 1 < len(g) and (g[j + 1][2] != nci or g[j + 1][4] != nct):
                forks["click_forks"] += 1
                fresh = lambda d: {k: v[len(ci[k]):] for k, v in d.items() if len(v) > len(ci[k])}   # the clicks of this round
                ev = {"round": j + 1, "kind": None, "proven": False, "gpu_clicks": fresh(g[j + 1][2]), "oracle_clicks": fresh(nci)}
                if not same_pred:
                    ev.update({"kind": "follows a logit tie", "proven": bool(label_tie)})
                else:
                    cg = {c["cluster_id"]: c for c in pc.error_clusters(pred.to(dev), lab.to(dev), xyz.to(dev))}
                    co = {c["cluster_id"]: c for c in oc.error_clusters(pred, lab, xyz)}
                    cid_all = lab.float() * 96 + pred.float() * 11
                    wrong = pred != lab
                    ties, ok = [], set(cg) == set(co)
                    for cid in sorted(set(cg) & set(co)):
                        ra, rb_ = cg[cid]["row"], co[cid]["row"]
                        if ra == rb_:
                            continue
                        member = wrong & (cid_all == cid)
                        d64 = torch.cdist(x64[[ra, rb_]], x64[~member]).min(1).values           # float64: exact to 1e-16
                        dcd = torch.cdist(xyz[~member], xyz[[ra, rb_]]).min(0).values.double()   # the reference's arithmetic
                        err_cd = float((dcd - d64).abs().sum())
                        ulp = float(np.spacing(np.float32(d64.max())))
                        proven = bool(d64[0] >= d64[1] - 2 * ulp and dcd[1] >= dcd[0]
                                      and float((d64[0] - d64[1]).abs()) <= err_cd + 2 * ulp)
                        ties.append({"cluster": cid, "gpu_row": ra, "oracle_row": rb_, "float64": [float(d64[0]), float(d64[1])],
                                     "torch_cdist": [float(dcd[0]), float(dcd[1])], "float64_gap": float((d64[0] - d64[1]).abs()),
                                     "cdist_error_on_the_two": err_cd, "proven": proven})
                        ok = ok and proven
                    ev["ties"] = ties
                    # the ranking of the clusters (largest error size first, stable): the top cluster is what the rounds
                    # after the first click on (utils/seg.py:213-221)
                    rank_g = sorted(cg, key=lambda c: cg[c]["error_size"], reverse=True)
                    rank_o = sorted(co, key=lambda c: co[c]["error_size"], reverse=True)
                    rank = None
                    if n_clicks > 0 and rank_g[:1] != rank_o[:1] and set(cg) == set(co):
                        c1, c2 = rank_g[0], rank_o[0]
                        s64 = {c: _size64(x64, wrong & (cid_all == c)) for c in (c1, c2)}
                        err_cd = sum(abs(co[c]["error_size"] - s64[c][0]) for c in (c1, c2))
                        ulp = float(np.spacing(np.float32(max(s64[c1][0], s64[c2][0]))))
                        gap = abs(s64[c1][0] - s64[c2][0])
                        proven = bool(cg[c1]["error_size"] >= cg[c2]["error_size"] and co[c2]["error_size"] >= co[c1]["error_size"]
                                      and gap <= err_cd + 2 * ulp)
                        rank = {"gpu_top": c1, "oracle_top": c2, "float64_sizes": [s64[c1][0], s64[c2][0]],
                                "gpu_sizes": [cg[c1]["error_size"], cg[c2]["error_size"]],
                                "torch_cdist_sizes": [co[c1]["error_size"], co[c2]["error_size"]], "float64_gap": gap,
                                "cdist_error_on_the_two": err_cd, "proven": proven}
                        ok = ok and proven
                    ev["rank"] = rank
                    explained = ok and bool(ties or rank)
                    ev.update({"kind": ("rank tie" if rank else "distance tie") if explained else "unexplained", "proven": explained})
                forks["unexplained"] += 0 if ev["proven"] else 1
                rec["events"].append(ev)
                nci, nct = {k: list(v) for k, v in g[j + 1][2].items()}, {k: list(v) for k, v in g[j + 1][4].items()}
            ci, ct = nci, nct
            n_clicks += objects if n_clicks == 0 else 1
        forks["scenes"].append(rec)
    return oracle_log, oracle_bb, forks


def iou_at_k(dev, n_scenes=4, voxels=5000, objects=3, max_clicks=20, fit_iters=120, lr=1e-3, min_iou5=0.5, more_iters=40,
             max_fit_iters=360):
    """BASELINE.json's "IoU@k vs ref" on what exists offline, in the regime the reference operates in.  ScanNet and the
    authors' checkpoint cannot be had here, so the state dict is FITTED first: `fit_iters` iterations of the repository's
    own training path (agile3d_amd.fit = train_step.train_one_step = the reference's engine.py:38-150 on the HIP kernels,
    AdamW + clip) on `n_scenes` seeded labelled synthetic scenes -- stopped early on purpose, so that the interactive
    protocol starts around IoU@1 0.5-0.6 and climbs over the rounds (small error clusters, NoC thresholds crossed
    mid-run) instead of random-init noise (IoU 0.05) or a memorised 1.0.  Then the protocol (eval_multi_obj.py:76-173 ->
    evaluation/evaluator_MO.py IoU@k / NoC@q) runs twice on those scenes with that state dict and the same `random`
    seed: the GPU product (agile3d_amd.Evaluate: HIP backbone + decoder, label argmax, IoU counters, click simulator)
    and the CPU oracle (oracle.clicks.interactive_rounds over oracle backbone + decoder); both CSVs go through the same
    EvaluatorMO, and the rounds are compared one by one (clicks chosen, IoU)."""
    import contextlib
    import io
    import random
    import tempfile
    import types
    from agile3d_amd import build_model, default_args
    from agile3d_amd.evaluate import Evaluate, EvaluatorMO
    from agile3d_amd.fit import eval_loader, fit, labelled_scenes
    from oracle import backbone as ob, clicks as oc, decoder as od
    torch.manual_seed(0)
    model = build_model(default_args()).to(dev)
    items = labelled_scenes(n_scenes, voxels, objects)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = fit(model, items, dev, iters=fit_iters, lr=lr, batch=2, seed=7)
    # training is deterministic per build, not across builds (a kernel that rounds differently in the last bits fits to
    # another point of the same curve: round 4 saw IoU@5 0.71 and 0.86 from two builds): fit on in steps of `more_iters`
    # until the GPU protocol reaches `min_iou5` at 5 clicks per object, so that the comparison below always runs in the
    # regime it is meant for; the line reports the iterations it took
    loader, val = eval_loader(items)
    tmp = tempfile.mkdtemp(prefix="a3d_iou_")
    json.dump(val, open(os.path.join(tmp, "val.json"), "w"))
    while min_iou5 is not None and len(losses) < max_fit_iters:
        probe = types.SimpleNamespace(output_dir=os.path.join(tmp, f"probe{len(losses)}"), max_num_clicks=5, val_list=None)
        at5 = []
        random.seed(11)
        with contextlib.redirect_stdout(io.StringIO()):
            Evaluate(model, loader, probe, dev,
                     lambda idx, cur, pred, iou, ci_, ct_: at5.append(float(iou)) if cur == 5 * objects else None)
        if float(np.mean(at5)) >= min_iou5:
            break
        losses += fit(model, items, dev, iters=more_iters, lr=lr, batch=2, seed=7 + len(losses), optimizer=fit.optimizer)
    fit_iters = len(losses)
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    model.eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # ---- GPU product
    args = types.SimpleNamespace(output_dir=os.path.join(tmp, "gpu"), max_num_clicks=max_clicks, val_list=os.path.join(tmp, "val.json"))
    gpu_log = []
    random.seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        res_gpu = Evaluate(model, loader, args, dev,
                           lambda idx, cur, pred, iou, ci_, ct_: gpu_log.append(
                               (cur, float(iou), {k: list(v) for k, v in ci_.items()}, pred.cpu().clone().long(),
                                {k: list(v) for k, v in ct_.items()})))
    # ---- CPU oracle, same state dict, same seed, next to the GPU run's log: every round compared (oracle_protocol_synced)
    os.makedirs(os.path.join(tmp, "cpu"), exist_ok=True)
    random.seed(11)
    oracle_log, oracle_bb, forks = oracle_protocol_synced(model, sd, items, gpu_log, objects, max_clicks, dev)
    per_scene = len(oracle_log) // len(items)
    with open(os.path.join(tmp, "cpu", "val_results_multi.csv"), "w") as f:
        for k_, r in enumerate(oracle_log):
            i = k_ // per_scene
            f.write(f"{i} {items[i]['name'].replace('scene', '')} {objects} {r[0] / objects} {np.float32(r[1])}\n")
    with contextlib.redirect_stdout(io.StringIO()):
        res_cpu = EvaluatorMO(os.path.join(tmp, "val.json"), os.path.join(tmp, "cpu", "val_results_multi.csv"),
                              [0.5, 0.65, 0.8, 0.85, 0.9]).eval_results()
    rounds = min(len(gpu_log), len(oracle_log))
    same_iou = [abs(a_[1] - b_[1]) <= 1e-6 for a_, b_ in zip(gpu_log, oracle_log)]
    fork_rounds = sorted(si_ * per_scene + e_["round"] for si_, s_ in enumerate(forks["scenes"]) for e_ in s_["events"] if "gpu_clicks" in e_)
    same_clicks = [k_ not in fork_rounds for k_ in range(rounds)]     # a round whose clicks the oracle would have picked differently
    first_div = next((i for i, (c_, u_) in enumerate(zip(same_clicks, same_iou)) if not (c_ and u_)), None)
    ks = (1, 3, 5, 10, 15)
    noc_g = {k_: round(float(v), 4) for k_, v in res_gpu.items() if k_.startswith("NoC")}
    noc_c = {k_: round(float(v), 4) for k_, v in res_cpu.items() if k_.startswith("NoC")}
    del model
    torch.cuda.empty_cache()
    return {"k": list(ks), "gpu": [round(float(res_gpu[f"IoU@{k_}"]), 6) for k_ in ks],
            "oracle": [round(float(res_cpu[f"IoU@{k_}"]), 6) for k_ in ks],
            "max_abs_diff": max(abs(float(res_gpu[f"IoU@{k_}"]) - float(res_cpu[f"IoU@{k_}"])) for k_ in ks),
            "noc_gpu": noc_g, "noc_oracle": noc_c,
            "noc_thresholds_crossed_before_max_clicks": sorted(k_ for k_, v in noc_g.items() if v < max_clicks),
            "rounds": rounds, "rounds_with_identical_clicks": int(sum(same_clicks)), "rounds_with_identical_iou": int(sum(same_iou)),
            "first_differing_round": first_div, "forks": forks,
            "weights": {"kind": "fitted", "fit_iterations": fit_iters, "lr": lr, "fit_seconds": round(fit_s, 1),
                        "adaptive_fit": min_iou5 is not None,     # fitted ON in steps until the GPU protocol reaches min_iou5 on the scenes
                        "min_iou5": min_iou5,                     # it is then evaluated on: a best case, not a fixed-iteration protocol
                        "ms_per_iteration": round(1e3 * fit_s / fit_iters, 1),
                        "loss_first5_mean": round(float(np.mean(losses[:5])), 4), "loss_last5_mean": round(float(np.mean(losses[-5:])), 4)},
            "note": f"interactive protocol on {n_scenes} seeded synthetic scenes ({voxels} voxels, {objects} objects, up to "
                    f"{max_clicks} clicks per object) with a state dict fitted on them by the repository's own training path: GPU "
                    "product (Evaluate) vs CPU oracle (interactive_rounds) round by round, both CSVs through EvaluatorMO; ScanNet "
                    "+ the authors' checkpoint are not available offline"}


def train_iter_ms(dev, voxels=80_000, batch=4, iters=5, warm=3):
    """BASELINE.json config 4's iteration (engine.py:38-150: training-mode backbone, no-grad click rounds, training-mode
    decoder, losses, backward, clip, AdamW) on ONE GPU: `iters` seeded iterations on a batch of 4 x 80 k-voxel labelled
    synthetic scenes after `warm` warm-ups (the first iterations grow the allocator's pools).  The number of click rounds
    is drawn per iteration (0..19, engine.py:83), so they are stated separately: `ms_without_click_rounds` is the median
    of (iteration - click rounds), phases device-synchronised."""
    import random
    from agile3d_amd import batched_coordinates, build_model, default_args
    from agile3d_amd.criterion import build_mask_criterion
    from agile3d_amd.optim import AdamW
    from agile3d_amd.train_step import train_one_step
    torch.manual_seed(0)
    targs = default_args(bce_loss_coef=1.0, dice_loss_coef=2.0, losses=["bce", "dice"])
    model = build_model(targs).to(dev)
    crit = build_mask_criterion(targs)
    scenes = [make_scene_(voxels, seed=b) for b in range(batch)]
    b = (batched_coordinates([s["coords"][:, 1:] for s in scenes]),
         torch.from_numpy(np.concatenate([s["raw_xyz"] for s in scenes])),
         torch.from_numpy(np.concatenate([s["feats"] for s in scenes])),
         [torch.from_numpy(s["labels"].astype(np.int64)) for s in scenes], None, None, [{} for _ in scenes],
         tuple(f"scene{i:04d}_00" for i in range(batch)), tuple(0 for _ in scenes))
    opt = AdamW(model.named_parameters(), lr=1e-4, weight_decay=1e-4)
    np.random.seed(1), random.seed(1)
    old = os.environ.get("A3D_TRAIN_TIMING")
    os.environ["A3D_TRAIN_TIMING"] = "quiet"
    rows = []
    try:
        # warm-ups, then the seeded sequence ONCE untimed (its iterations with 15-18 click rounds reach buffer sizes -- decoder
        # workspaces of 150-200 queries, the per-scene key / value / query cache -- the warm-ups do not: their first occurrence
        # paid 30-50 ms of device allocations inside the timed iterations), then the same draws again, timed
        for it in range(warm + iters):
            if it == warm:
                draws = (np.random.get_state(), random.getstate())
            train_one_step(model, crit, opt, b, dev, 0.1)
        np.random.set_state(draws[0]), random.setstate(draws[1])
        for it in range(warm, warm + iters):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = train_one_step(model, crit, opt, b, dev, 0.1)
            torch.cuda.synchronize()
            if it >= warm:
                ph = st["phases_ms"]
                click = ph.get("click simulation", 0.0)
                rows.append({"ms": round(1e3 * (time.perf_counter() - t0), 2), "click_rounds": st["click_rounds"],
                             "click_rounds_ms": round(click, 2), "phases_ms": {k: round(v, 2) for k, v in ph.items()}})
    finally:
        if old is None:
            os.environ.pop("A3D_TRAIN_TIMING", None)
        else:
            os.environ["A3D_TRAIN_TIMING"] = old
    del model, opt
    torch.cuda.empty_cache()
    wo = [r["ms"] - r["click_rounds_ms"] for r in rows]
    per_round = [r["click_rounds_ms"] / r["click_rounds"] for r in rows if r["click_rounds"]]
    return {"ms_without_click_rounds": round(float(np.median(wo)), 2), "ms_all": [r["ms"] for r in rows],
            "click_rounds": [r["click_rounds"] for r in rows],
            "ms_per_click_round": round(float(np.median(per_round)), 2) if per_round else None,
            "phases_ms_median": {k: round(float(np.median([r["phases_ms"].get(k, 0.0) for r in rows])), 2)
                                 for k in rows[0]["phases_ms"]},
            "prewarmed": True,     # the seeded draws run once untimed first: allocations of the long click schedules are warm
            "workload": f"{batch} x {voxels}-voxel labelled synthetic scenes per iteration, 1 GPU, fp32, AdamW + clip 0.1; "
                        f"median of {iters} seeded iterations after {warm} warm-ups and one untimed run of the same draws"}


def make_scene_(voxels, seed):
    from agile3d_amd.synthetic import make_scene
    return make_scene(voxels, seed=seed)


def pin_to_gpu_numa_node(local_rank):
    """Keep this rank's launch thread (about 130 launches per 8 ms step) on the cores of its GPU's NUMA node."""
    try:
        bdf = torch.cuda.get_device_properties(local_rank).pci_bus_id
    except Exception:
        try:
            import subprocess
            bdf = None
            out = subprocess.run(["rocm-smi", "--showbus"], capture_output=True, text=True, timeout=20).stdout
            for line in out.splitlines():
                if line.startswith(f"GPU[{local_rank}]"):
                    bdf = line.split()[-1]
        except Exception:
            bdf = None
    try:
        if not bdf:
            return None
        node = int(open(f"/sys/bus/pci/devices/{bdf.lower()}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return node
    except Exception:
        pass
    return None


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks under torch.distributed.run
    (--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>) and hand its output through."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from agile3d_amd.hostcpu import cpu_quota
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, int(cpu_quota() // (2 * n))))))   # the ranks share the container's CPU quota
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=15,
                    help="the timed region of --steps steps is repeated this many times; the MEDIAN repetition is reported")
    ap.add_argument("--voxels", type=int, default=80_000)
    ap.add_argument("--objects", type=int, default=5)
    ap.add_argument("--clicks-per-object", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-iteration timing (train_iter)")
    ap.add_argument("--dump-logits", default="", help="write scene 0's logits of the first step to this file (torch.save)")
    ap.add_argument("--steps-only", action="store_true",
                    help="only the batched steps (warm-up, timed region, instrumented pass): no latency / phase / eval-round "
                         "/ CPU passes -- the command tools/profile_round.sh traces for profiles/kernel_avg_us.json, so that "
                         "every launch of a kernel in the trace is a launch of the step the roofline object describes")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("A3D_BENCH_BATCH", "16")),
                    help="scenes per step and rank (one batched SparseTensor, as the reference's collate builds)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("A3D_BENCH_STREAMS", "4")),
                    help="scenes in flight per GPU: consecutive steps are issued round-robin on this many HIP streams")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` launches its own N ranks (one process per GPU) through torch.distributed.run,
        # exactly the command the driver would have typed; rank 0's JSON line is this process's only stdout line
        raise SystemExit(self_launch(args.gpus))
    # fewer GPUs on the box than ranks (or A3D_BENCH_ONE_GPU=1): ranks share devices round-robin and talk over gloo
    # instead of RCCL -- exercises the N > 1 code path (barriers, MAX over ranks, the line rank 0 prints) on a
    # single-GPU box; the line says so (`ranks_share_gpus`) and is never a scaling measurement
    n_dev = torch.cuda.device_count()
    one_gpu = os.environ.get("A3D_BENCH_ONE_GPU", "0") == "1" or (world > 1 and n_dev < world)
    if one_gpu:
        local_rank = local_rank % max(1, n_dev) if os.environ.get("A3D_BENCH_ONE_GPU", "0") != "1" else 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    numa = pin_to_gpu_numa_node(local_rank) if world > 1 else None
    ranks_seen = 1
    # A3D_BENCH_DIST_AT_1=1 under torch.distributed.run with ONE rank: the distributed path (RCCL communicator, barriers,
    # MAX over ranks, ranks_seen) stays on at world size 1 -- how the nccl branch is exercised on a one-GPU box
    dist_on = world > 1 or (os.environ.get("A3D_BENCH_DIST_AT_1", "0") == "1" and "WORLD_SIZE" in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # how many ranks the collective backend really connects: an all-reduce of ones (over RCCL on a multi-GPU node)
        ones = torch.ones(1, dtype=torch.float32, device="cpu" if one_gpu else dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    import __graft_entry__ as g
    if rank == 0:
        from agile3d_amd import build as _b
        if _b.needs_build():
            g.build()
    if dist_on:
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
    n0 = len(sc["coords"])

    def one_scene():       # one STEP: the whole batch through scene build + backbone + one decoder pass per sample
        x = SparseTensor(features=feats, coordinates=coords)
        r = model.forward_backbone(x, raw_coordinates=raw)
        return r, model.forward_mask(*r, click_idx=cis, click_time_idx=cts)

    r0, out0 = one_scene()                  # packs the weights once, on the default stream
    torch.cuda.synchronize()
    assert torch.isfinite(out0["pred_masks"][0]).all()
    gpu_logits0 = out0["pred_masks"][0].clone()
    gpu_feats0 = r0[0].F[:n0].clone()
    if args.dump_logits and rank == 0:
        torch.save(gpu_logits0.cpu(), args.dump_logits)
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    issued = [0]

    def step():
        if streams is None:
            return one_scene()[1]
        with torch.cuda.stream(streams[issued[0] % len(streams)]):
            issued[0] += 1
            return one_scene()[1]

    from agile3d_amd.sharding import timed_steps
    for _ in range(args.warmup):
        step()
    # the timed region (exactly --steps steps between barrier + device sync, MAX over ranks) is repeated --reps
    # times; the line reports the MEDIAN repetition and lists them all
    from agile3d_amd.hostcpu import cpu_throttle_stats
    thr0 = cpu_throttle_stats()
    rep_dt, own_dt = [], []
    for _ in range(max(1, args.reps)):
        dt_, out = timed_steps(step, args.steps, world, dev, own=own_dt)
        rep_dt.append(dt_)
    thr1 = cpu_throttle_stats()
    dt = float(np.median(rep_dt))
    per_rank_ms = [1e3 * float(np.median(own_dt)) / args.steps]
    if dist_on:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank_ms[0])
        per_rank_ms = [float(x) for x in gathered]
    assert torch.isfinite(out["pred_masks"][0]).all()

    res = {
        "metric": f"scenes/s ({round(n0 / 1000)}k-voxel, {args.objects * args.clicks_per_object} clicks)",
        "value": world * args.batch * args.steps / dt, "unit": "scenes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"batch of {args.batch} synthetic {n0}-voxel scenes, "
                               f"{args.objects * args.clicks_per_object} clicks each ({args.objects} objects x "
                               f"{args.clicks_per_object}), scene build + forward_backbone + 1 forward_mask per scene, fp32, eval",
                   "voxels": int(n0), "queries": args.objects * args.clicks_per_object + 10,
                   "parallelism": f"scene-sharded x{world} (no data-path collective)",
                   "scenes_per_step_per_gpu": args.batch, "global_batch": world * args.batch,
                   "steps_in_flight_per_gpu": args.streams},
        "timed_region": {"repetitions": len(rep_dt), "reported": "median",
                         "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in rep_dt]},
    }
    if numa is not None:
        res["config"]["launch_thread_numa_node"] = numa
    from agile3d_amd.hostcpu import cpu_quota
    res["config"]["host_cpu"] = {"visible": os.cpu_count(), "container_quota": round(cpu_quota(), 2),
                                 "torch_threads": torch.get_num_threads(),
                                 # CFS throttling of the container during the timed region (all ranks share the cgroup):
                                 # nr_throttled > 0 = the kernel froze every thread, launch threads included, for part of a period
                                 "cfs_during_timed_region": {k: thr1[k] - thr0[k] for k in thr1 if k in thr0} or None}
    if dist_on:
        res["ranks_seen"] = ranks_seen
        res["ms_per_step_per_rank"] = [round(x, 4) for x in per_rank_ms]
        res["config"]["backend"] = "gloo" if one_gpu else "nccl (RCCL)"
        res["config"]["gpus_visible"] = n_dev
        if one_gpu:
            res["ranks_share_gpus"] = True
            res["note_multi_rank"] = (f"{world} ranks on {max(1, n_dev)} visible GPU(s): plumbing check of the N > 1 path, "
                                      "not a scaling measurement")

    if rank == 0:
        # (N = 1: everything below; N > 1: rank 0 still measures the roofline object of its own GPU while the other
        # ranks wait at the closing barrier -- the line of every N carries it; latency / phase / CPU passes are N = 1 only)
        # SURVEY 8(d) latency configuration: batch 1, one stream, every scene timed on its own (device sync on
        # both sides), median of 50 after 10 warm-ups -- the interactive product's operating point
        c1, f1, w1 = coords[:n0].contiguous(), feats[:n0].contiguous(), raw[:n0].contiguous()

        def single():
            r = model.forward_backbone(SparseTensor(features=f1, coordinates=c1), raw_coordinates=w1)
            return model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
        lat = []
        for i in range(0 if (args.steps_only or world > 1) else 60):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            single()
            torch.cuda.synchronize()
            if i >= 10:
                lat.append(1e3 * (time.perf_counter() - t0))
        if lat:
            res["latency_ms_per_scene"] = round(float(np.median(lat)), 4)
            res["latency_note"] = "batch 1, 1 stream, one scene at a time: median of 50 after 10 warm-ups (host wall, device sync both sides)"
        if not args.no_profile:
            scn = Scene(coords)
            pairs = {"n": scn.n, "conv3": []}
            for lvl in range(5):
                npad = (max(scn.n[lvl], 1) + 127) // 128 * 128
                nb = scn.table(lvl, L.TAB_NBR27).reshape(27, npad)
                pairs["conv3"].append(int((nb[:, :scn.n[lvl]] < scn.n[lvl]).sum()))
            n_prof = 10
            agg = profile_pass(lambda: one_scene()[1], pairs, n_prof)   # one step at a time: clean kernel times
            conv = {k: v for k, v in agg.items() if k.startswith("k_conv") or k.startswith("k_dense")}
            dom = max((k for k in conv if k.startswith("k_conv")), key=lambda k: conv[k]["ms"])
            d = conv[dom]
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            # counters / rocprofv3 averages are only evidence for the workload they were collected on: the committed files
            # are keyed by workload ("<k voxels>k_b<batch>_q<queries>"), anything else prints null
            Q = args.objects * args.clicks_per_object + 10
            wkey = workload_key(n0, args.batch, Q)
            traffic = (profile_file("pmc_summary.json", wkey).get(dom) or {}).get("hbm_bytes_per_launch")
            hbm_gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9          # algorithmic (compulsory) bytes / measured time
            res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                               "hbm_achieved_gbs": hbm_gbs, "hbm_peak_gbs": PEAK_HBM_GBS, "hbm_frac": hbm_gbs / PEAK_HBM_GBS,
                               "hbm_traffic_frac": hbm_traffic_frac(traffic, d["ms"] / d["launches"]),
                               "avg_launch_ms": d["ms"] / d["launches"], "launches_per_step": d["launches_per_step"],
                               "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
                               "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                               "how": f"per launch position: median of {n_prof} HIP-event samples; kernel = sum over its "
                                      f"launches; dominant = largest sum"}
            # agreement with the committed rocprofv3 --kernel-trace --stats summary of the same command on the same workload
            res["roofline"]["profiles_workload"] = wkey if (traffic is not None or profile_file("kernel_avg_us.json", wkey)) else None
            want = profile_file("kernel_avg_us.json", wkey).get(dom)
            if want:
                got = 1e3 * d["ms"] / d["launches"]
                res["roofline"]["rocprof_avg_launch_us"] = want
                res["roofline"]["rocprof_command"] = ("rocprofv3 --kernel-trace --stats -- python bench.py --steps-only --streams 1"
                                                      + ("" if wkey == workload_key(80_000, 16, 20) else
                                                         f" --voxels {args.voxels} --clicks-per-object {args.clicks_per_object} --batch {args.batch}"))
                res["roofline"]["agrees_with_profiles_within_10pct"] = bool(abs(got - want) <= 0.1 * want)
            else:
                res["roofline"]["rocprof_avg_launch_us"] = None
            tot_flops = sum(v["flops"] for v in conv.values())
            res["kernels_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in sorted(agg.items())}
            res["conv_tflops"] = {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in conv.items()}
            res["scene_algorithmic_gflop_convs"] = round(tot_flops / 1e9, 2)
            res["gpu_ms_per_step_sum_of_kernels"] = round(sum(v["ms_per_step"] for v in agg.values()), 3)
            # whole timed pipeline against the same peak: algorithmic FLOPs per step (convs as counted above +
            # SURVEY 8(d)'s decoder formula per scene) / measured ms per step
            Q = args.objects * args.clicks_per_object + 10
            dec_flops = args.batch * 3 * (2.0 * n0 * 128 * 128 * 2 + 4.0 * Q * n0 * 128 + 2.0 * n0 * 128 * 128 * 2
                                          + 4.0 * n0 * Q * 128 + 2.0 * n0 * 128 * Q)
            conv_only = sum(v["flops"] for k, v in conv.items() if k.startswith("k_conv") or k in ("k_dense<6,8>", "k_dense<8,6>"))
            step_gf = (conv_only + dec_flops) / 1e9
            res["pipeline_algorithmic_gflop_per_step"] = round(step_gf, 1)
            res["pipeline_frac"] = round(step_gf / res["ms_per_step"] / PEAK_FP32_MFMA_TFLOPS, 4)   # GF / ms = TF/s
            # both fractions.  Compulsory bytes of a step = the convs' (as counted per launch above) + the
            # decoder's fused-ideal figure L 4 (2 N d + 3 N d) + 4 N (1 + K) per scene
            dec_bytes = args.batch * (3 * 4.0 * 5 * n0 * 128 + 4.0 * n0 * (1 + args.objects))
            step_bytes = sum(v["bytes"] for v in conv.values()) + dec_bytes
            res["pipeline_algorithmic_gbyte_per_step"] = round(step_bytes / 1e9, 3)
            res["pipeline_hbm_frac"] = round(step_bytes / 1e9 / (res["ms_per_step"] * 1e-3) / PEAK_HBM_GBS, 4)
        if not args.no_profile and not args.steps_only and world == 1:
            # the three phases on their own (one step at a time, one stream, wall clock with a device
            # sync around each) and the eval loop's number, decoder passes/s (backbone results reused per round)
            def wall(fn, reps=10):
                fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    out = fn()
                    torch.cuda.synchronize()
                    ts.append(1e3 * (time.perf_counter() - t0))
                return float(np.median(ts)), out
            t_scene, _ = wall(lambda: Scene(coords))
            t_bb, r = wall(lambda: model.forward_backbone(SparseTensor(features=feats, coordinates=coords),
                                                          raw_coordinates=raw))
            t_dec, _ = wall(lambda: model.forward_mask(*r, click_idx=cis, click_time_idx=cts))
            res["phases_ms_per_step"] = {"scene_build": round(t_scene, 3), "backbone": round(t_bb - t_scene, 3),
                                         "decoder_pass": round(t_dec, 3), "note": "single stream, host wall clock, medians of 10"}
            res["decoder_passes_per_s"] = round(args.batch / (t_dec * 1e-3), 1)
            r1 = model.forward_backbone(SparseTensor(features=f1, coordinates=c1), raw_coordinates=w1)
            t_dec1, _ = wall(lambda: model.forward_mask(*r1, click_idx=[ci], click_time_idx=[ct]), reps=30)
            res["decoder_pass_ms_single"] = round(t_dec1, 4)
            # one decoder pass on that scene by query count (5 objects x k clicks + 10 learned queries): 20 / 60 run the
            # <= 64-query kernels, 85 / 160 the fused wide tier (decoder_wide.h); the reference's protocol adds clicks up
            # to num_obj x 20 (eval_multi_obj.py:116-118), training up to 19 click rounds (engine.py:83-93)
            by_q = {}
            for cpo in (2, 10, 15, 30):
                ciq, ctq = make_clicks(sc["labels"], args.objects, cpo, 0, seed=11 + cpo)
                tq, _ = wall(lambda: model.forward_mask(*r1, click_idx=[ciq], click_time_idx=[ctq]), reps=20)
                by_q[str(args.objects * cpo + 10)] = round(tq, 4)
            res["decoder_pass_ms_by_queries"] = by_q
            res["decoder_pass_note"] = ("repeated passes on ONE backbone output, as the interactive loop runs them: from the third pass on "
                                        "the first layer's click-to-scene keys / values come from the per-scene cache (A3D_KV_CACHE_MB=0 "
                                        "switches it off); the timed steps of `value` run one pass per fresh scene and never use it")
            # one round of the evaluation protocol (eval_multi_obj.py:112-160) on that scene: forward_mask -> label argmax
            # with the clicked rows overwritten -> IoU -> click simulator -> extend_clicks; median of 20 rounds
            import random as _random
            from agile3d_amd import clicks as pc
            lab_np = np.zeros(n0, np.int64)
            sizes = sorted(((int((sc["labels"] == i).sum()), i) for i in np.unique(sc["labels"]) if i > 0), reverse=True)
            for k_, (_, i) in enumerate(sizes[:args.objects], start=1):
                lab_np[sc["labels"] == i] = k_
            lab = torch.from_numpy(lab_np).to(dev).to(torch.int32)     # ids as int32 once per scene, as evaluate.py holds them
            eci = {str(k_): [] for k_ in range(args.objects + 1)}
            ect = {str(k_): [] for k_ in range(args.objects + 1)}
            pred = torch.zeros(n0, dtype=torch.int32, device=dev)
            _random.seed(0)
            rounds = []
            for rnd in range(24):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if rnd:
                    o = model.forward_mask(*r1, click_idx=[eci], click_time_idx=[ect])["pred_masks"][0]
                    pred = pc.argmax_labels(o, eci)
                _, cl_ = pc.mean_iou_and_clusters_batch([pred], [lab], None, [lab], [w1])     # IoU + error clusters, one host sync
                new, _, _, nt = pc.pick_clicks_batch(cl_, [lab], [w1], rnd, training=False)[0]
                if new is not None:
                    pc.extend_clicks(eci, ect, new, nt)
                torch.cuda.synchronize()
                if rnd >= 4:
                    rounds.append(1e3 * (time.perf_counter() - t0))
            res["eval_round_ms"] = round(float(np.median(rounds)), 4)
            # the same protocol with the whole batch in flight: ONE batched forward_mask per round (a launch per decoder
            # kernel for all scenes), then label argmax / IoU / click simulator per scene; scene-rounds per second
            rB = model.forward_backbone(SparseTensor(features=feats, coordinates=coords), raw_coordinates=raw)
            labs, raws = [], []
            for b_, s_ in enumerate(scenes):
                lb = np.zeros(len(s_["coords"]), np.int64)
                sz = sorted(((int((s_["labels"] == i).sum()), i) for i in np.unique(s_["labels"]) if i > 0), reverse=True)
                for k_, (_, i) in enumerate(sz[:args.objects], start=1):
                    lb[s_["labels"] == i] = k_
                labs.append(torch.from_numpy(lb).to(dev).to(torch.int32))
                raws.append(torch.from_numpy(s_["raw_xyz"]).to(dev))
            ecis = [{str(k_): [] for k_ in range(args.objects + 1)} for _ in scenes]
            ects = [{str(k_): [] for k_ in range(args.objects + 1)} for _ in scenes]
            preds = [torch.zeros(len(s_["coords"]), dtype=torch.int32, device=dev) for s_ in scenes]
            _random.seed(0)
            brounds = []
            for rnd in range(16):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if rnd:
                    outs = model.forward_mask(*rB, click_idx=ecis, click_time_idx=ects)["pred_masks"]
                if rnd:
                    preds = pc.argmax_labels_batch(outs, ecis)
                _, cls_ = pc.mean_iou_and_clusters_batch(preds, labs, None, labs, raws)
                for b_, (new, _, _, nt) in enumerate(pc.pick_clicks_batch(cls_, labs, raws, rnd, training=False)):
                    if new is not None:
                        pc.extend_clicks(ecis[b_], ects[b_], new, nt)
                torch.cuda.synchronize()
                if rnd >= 3:
                    brounds.append(time.perf_counter() - t0)
            res["eval_rounds_per_s"] = round(len(scenes) / float(np.median(brounds)), 1)
            res["eval_rounds_note"] = (f"{len(scenes)} scenes advance in lock-step (eval_multi_obj.py:114,162-166 with a batch): one "
                                       "batched forward_mask, then the scenes' label argmax / IoU counts / error clusters side by side ("
                                       + ("one host round trip" if len(scenes) <= 8 else "two host round trips: more than eight samples")
                                       + " per round); scene-rounds per second")
        if not args.steps_only and world == 1 and args.batch > 4:
            # the SAME scenes through round 2's protocol (4 scenes per step, the same number of steps in flight), so that a
            # change of `value` can be told apart from a change of protocol: `value_batch4`
            nb4 = sum(len(s_["coords"]) for s_ in scenes[:4])
            c4, f4, w4 = coords[:nb4].contiguous(), feats[:nb4].contiguous(), raw[:nb4].contiguous()

            def step4():
                stq = streams[issued[0] % len(streams)] if streams else torch.cuda.current_stream()
                with torch.cuda.stream(stq):
                    issued[0] += 1
                    r = model.forward_backbone(SparseTensor(features=f4, coordinates=c4), raw_coordinates=w4)
                    return model.forward_mask(*r, click_idx=cis[:4], click_time_idx=cts[:4])
            for _ in range(args.warmup):
                step4()
            rep4 = [timed_steps(step4, args.steps, 1, dev)[0] for _ in range(max(1, min(args.reps, 7)))]
            res["value_batch4"] = round(4 * args.steps / float(np.median(rep4)), 2)
            res["value_batch4_note"] = (f"same scenes, 4 per step (round 2's protocol), {args.streams} steps in flight, median of "
                                        f"{len(rep4)} repetitions of {args.steps} steps")
        if not args.steps_only and world == 1 and not args.no_train:
            try:
                res["train_iter"] = train_iter_ms(dev)
                res["train_iter_ms"] = res["train_iter"]["ms_without_click_rounds"]
                # the backbone's training passes against the fp32-MFMA peak: forward + input gradient + weight gradient = 3 x the
                # algorithmic conv FLOPs of a scene (counted above on the same 80 k-voxel scenes) x 4 scenes / (forward + backward ms)
                ph = res["train_iter"]["phases_ms_median"]
                if res.get("scene_algorithmic_gflop_convs") and abs(args.voxels - 80_000) < 1000:
                    gf = 3.0 * res["scene_algorithmic_gflop_convs"] / args.batch * 4
                    ms = ph["backbone forward"] + ph["backbone backward"]
                    res["train_iter"]["backbone_fwd_bwd"] = {"ms": round(ms, 2), "algorithmic_gflop": round(gf, 1),
                                                            "tflops": round(gf / ms, 1),
                                                            "frac_of_fp32_mfma_peak": round(gf / ms / PEAK_FP32_MFMA_TFLOPS, 3)}
            except Exception as e:   # never lose the headline line over the extra
                res["train_iter"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_cpu_baseline and not args.steps_only and world == 1:
            res["cpu_baseline"], diff = cpu_baseline(sd, sc, ci, ct, gpu_logits0, gpu_feats0)
            res["parity_vs_oracle"] = diff
            res["max_abs_diff"] = diff["logits_max_abs_diff"] if diff else None
            try:
                res["iou_at_k"] = iou_at_k(dev)
            except Exception as e:   # never lose the headline line over the extra
                res["iou_at_k"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and world == 1 and not args.steps_only and not args.no_profile and os.environ.get("A3D_CONV_EMU", "0") == "0":
        # the opt-in emulated-fp32 build of the dominant conv kernel (A3D_CONV_EMU=1: every fp32 product from six bf16 MFMAs,
        # DESIGN.md 4.1) measured next to the headline in its own process -- reported, never `value`
        try:
            import subprocess
            env = dict(os.environ, A3D_CONV_EMU="2")
            import tempfile
            dump = os.path.join(tempfile.gettempdir(), f"a3d_bench_emu_logits_{os.getpid()}.pt")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps-only", "--reps", "5",
                                  "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch),
                                  "--streams", str(args.streams), "--dump-logits", dump], env=env, capture_output=True,
                                 text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            emu_logits = torch.load(dump)
            os.remove(dump)
            ref = getattr(cpu_baseline, "last_logits", None)
            res["emulated_fp32_products"] = {
                "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                "max_abs_diff": float((emu_logits - ref).abs().max()) if ref is not None else None,
                "max_abs_diff_vs_exact_build": float((emu_logits - gpu_logits0.cpu()).abs().max()),
                "note": "same workload with A3D_CONV_EMU=2: the gathered conv kernels form each fp32 product from 6 bf16-MFMA terms "
                        "(3-way operand split, fp32 accumulation; error bounded shape class by shape class on adversarial inputs in "
                        "tests/test_gpu_conv.py::test_emulated_fp32_products_error_bound, domain |x| >= 2^-100 or 0; parity tests "
                        "unchanged). Opt-in: not the arithmetic `value` is measured with"}
            rf = d.get("roofline")
            if rf:   # its own roofline: an emulated product costs six bf16 MFMA terms, so the bound is the bf16 peak / 6 -- never 157.3
                res["emulated_fp32_products"]["roofline"] = {
                    "bound": "mfma", "kernel": rf["kernel"], "achieved": rf["achieved"], "peak": PEAK_BF16_MFMA_TFLOPS / 6,
                    "unit": "TFLOP/s of fp32-equivalent products (6 bf16 MFMA terms each; peak = bf16 dense peak / 6)",
                    "frac": rf["achieved"] / (PEAK_BF16_MFMA_TFLOPS / 6), "traffic": None}
        except Exception as e:   # never lose the headline line over the extra
            res["emulated_fp32_products"] = {"error": str(e)[:200]}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
