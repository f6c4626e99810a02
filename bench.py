#!/usr/bin/env python3
"""Benchmark of the Flood-complex coverage sweep on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full coverage sweep of the workload, as SURVEY.md section 8d defines t_sweep: from the
sorted (here: Morton-sorted and box-tree-indexed) cloud and the simplices resident in HBM to the per-face
filtration values in HBM - every top-dimensional Delaunay simplex of the landmarks against the whole cloud
(sweep -> exact finish -> [all_reduce(MIN) across ranks] -> per-face maxima).  FPS, Delaunay, the cloud sort /
index build (the counterpart of the reference's argsort, core.py:140-144) and the Python dict / SimplexTree
hand-off are outside the step; the index build is timed separately ("ms_index_build") and
"value_including_index_build" charges it to every step.

value = N_points x S_top / t_step / 1e6  [M points x simplices / s], whole job over all ranks.
With N ranks the simplices are interleaved over the ranks (every rank holds the whole cloud) and the per-face
values are combined with one RCCL all_reduce(MIN); --shard points shards the cloud instead.  Total work is
fixed: "scaling": "strong".
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak fp32 vector

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(desc="1M-point 3D Gaussian, 1k landmarks, points_per_edge 30, max_dimension 3, fp32 "
                      "coverage sweep (BASELINE.json configs[1])",
                 gen="gauss", n=1_000_000, dim=3, n_lms=1000, ppe=30),
    "cfg3": dict(desc="1M-point 3D noisy torus, 1k landmarks, points_per_edge 30 (BASELINE.json configs[2])",
                 gen="torus", n=1_000_000, dim=3, n_lms=1000, ppe=30),
    "cfg5": dict(desc="16M-point 3D swiss cheese (6 voids), 4k landmarks, points_per_edge 30 (BASELINE.json configs[4], "
                      "cloud resident in HBM)",
                 gen="cheese", n=16_000_000, dim=3, n_lms=4000, ppe=30),
    "small": dict(desc="100k-point 3D Gaussian, 300 landmarks, points_per_edge 12 (debug)",
                  gen="gauss", n=100_000, dim=3, n_lms=300, ppe=12),
}


def make_points(w):
    torch.manual_seed(42)
    if w["gen"] == "gauss":
        return torch.randn(w["n"], w["dim"])
    if w["gen"] == "torus":
        theta = torch.rand(w["n"]) * 2 * torch.pi
        phi = torch.rand(w["n"]) * 2 * torch.pi
        x = (3.0 + torch.cos(phi)) * torch.cos(theta)
        y = (3.0 + torch.cos(phi)) * torch.sin(theta)
        z = torch.sin(phi)
        p = torch.stack((x, y, z), dim=1)
        return p + torch.randn_like(p) * 0.02
    if w["gen"] == "cheese":
        from flooder_amd.synthetic import generate_swiss_cheese_points
        return generate_swiss_cheese_points(w["n"], k=6, seed=42)[0]
    raise ValueError(w["gen"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1200, help="simplices in the CPU baseline sample")
    ap.add_argument("--variant", type=int, default=None, help="sweep_variant option of the library")
    ap.add_argument("--bvh-ks", type=int, default=None, help="samples per lane of the culled sweep (1,2,4,8)")
    ap.add_argument("--bvh-refine-pct", type=int, default=None,
                    help="tree sweep: cost-model threshold of the transposed refine in percent (100 = model, 1000000 = off)")
    ap.add_argument("--bvh-leaf-batch", type=int, default=None, help="exact finish: leaves fetched per step (1 or 4)")
    ap.add_argument("--curve", type=int, default=None, help="order of the cloud in the index: 0 Morton, 1 Hilbert (default)")
    ap.add_argument("--cell-exh-sparse", type=int, default=None,
                    help="cell sweep: most kept points a chunk with an (almost) empty box evaluates exhaustively")
    ap.add_argument("--cell-exh-dense", type=int, default=None,
                    help="cell sweep: most kept points a dense chunk evaluates exhaustively before it goes to the tree sweep")
    ap.add_argument("--cell-grid", type=int, default=None, help="persistent blocks of the cell sweep")
    ap.add_argument("--bvh-grid", type=int, default=None, help="persistent blocks of the tree sweep")
    ap.add_argument("--bvh-subs", type=int, default=None, help="waves per flagged tile in the exact finish")
    ap.add_argument("--shard", default="simplices", choices=["simplices", "points"],
                    help="multi-GPU decomposition: simplices (full cloud per rank, every W-th simplex; default) "
                         "or points (interleaved rows of the cloud, all_reduce(MIN) on the (S,R) minima)")
    ap.add_argument("--emulate-shard", default=None, metavar="r/W",
                    help="diagnostic (N=1 only): time rank r's share of a W-rank simplex-sharded step, no collective")
    ap.add_argument("--alpha", type=float, default=None, help="cell size of the cell sweep in units of the local spacing")
    ap.add_argument("--method", default="cell", choices=["cell", "bvh", "ball"],
                    help="cell: LDS cell-grid sweep + exact tree finish (default); bvh: box-tree culled sweep; "
                         "ball: the reference's formulation")
    args = ap.parse_args()

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # (test rigs may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("FLOODER_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import flooder_amd as fa
    from flooder_amd import _native, core
    from flooder_amd.distributed import min_reduce_hook

    lib = _native.load()  # no fallback: fails here if the HIP library is missing
    if args.variant is not None:
        _native.check(lib.flooder_set_option(b"sweep_variant", args.variant), "set_option")

    if args.bvh_ks is not None:
        _native.check(lib.flooder_set_option(b"bvh_ks", args.bvh_ks), "set_option")
    if args.bvh_refine_pct is not None:
        _native.check(lib.flooder_set_option(b"bvh_refine_pct", args.bvh_refine_pct), "set_option")
    if args.bvh_leaf_batch is not None:
        _native.check(lib.flooder_set_option(b"bvh_leaf_batch", args.bvh_leaf_batch), "set_option")
    if args.curve is not None:
        _native.check(lib.flooder_set_option(b"curve", args.curve), "set_option")
    if args.cell_exh_sparse is not None:
        _native.check(lib.flooder_set_option(b"cell_exh_sparse", args.cell_exh_sparse), "set_option")
    if args.cell_exh_dense is not None:
        _native.check(lib.flooder_set_option(b"cell_exh_dense", args.cell_exh_dense), "set_option")
    if args.cell_grid is not None:
        _native.check(lib.flooder_set_option(b"cell_grid", args.cell_grid), "set_option")
    if args.bvh_grid is not None:
        _native.check(lib.flooder_set_option(b"bvh_grid", args.bvh_grid), "set_option")
    if args.bvh_subs is not None:
        _native.check(lib.flooder_set_option(b"bvh_subs", args.bvh_subs), "set_option")
    if args.alpha is not None:
        core.CELL_ALPHA = args.alpha
    w = WORKLOADS[args.workload]
    # ------------------------------------------------------------------ untimed setup
    pts_cpu = make_points(w)
    pts_full = pts_cpu.to(dev)
    fa.generate_landmarks(pts_full, 8, start_idx=0)  # warm-up
    torch.cuda.synchronize()
    t_fps0 = time.perf_counter()
    lms = fa.generate_landmarks(pts_full, w["n_lms"], start_idx=0)
    torch.cuda.synchronize()
    t_fps = time.perf_counter() - t_fps0
    stree, simplices = core._build_complex(lms, w["dim"])
    d = w["dim"]
    simp = torch.as_tensor(simplices[d], device=dev)
    verts = lms[simp]
    centers, radii = core._ball_prep(verts, d)
    axis = int(torch.argmax(pts_full.max(dim=0).values - pts_full.min(dim=0).values).item())
    order_s = torch.argsort(centers[:, axis])
    verts, centers, radii, simp = verts[order_s], centers[order_s], radii[order_s], simp[order_s]
    weights, vertex_idxs, face_idxs = core.generate_grid(w["ppe"], d, dev, torch.float32)
    faces = core._FaceTable(face_idxs, weights.shape[0], dev)
    S_all = verts.shape[0]
    if world > 1 and args.shard == "points":
        order_p = torch.argsort(pts_full[:, axis])
        shard_raw = pts_full[order_p][rank::world].contiguous()  # this rank's interleaved share (raw rows)
        mine = None
        hook = min_reduce_hook()
    else:
        shard_raw = pts_full.contiguous()                          # whole cloud on every rank
        mine = torch.arange(rank, S_all, world, device=dev) if world > 1 else None
        hook = None
    face_hook = min_reduce_hook() if (world > 1 and mine is not None) else None
    if args.emulate_shard and world == 1:
        er, ew = (int(v) for v in args.emulate_shard.split("/"))
        mine = torch.arange(er, S_all, ew, device=dev)
        face_hook = lambda t: t  # noqa: E731  (the 360 KB all_reduce is not emulated)
    if mine is not None:
        verts, centers, radii = verts[mine].contiguous(), centers[mine].contiguous(), radii[mine].contiguous()
    dp = lib.flooder_padded_dim(w["dim"])
    del pts_full
    S, R = verts.shape[0], weights.shape[0]

    # reference-defined work of this input (untimed): candidate pairs P = sum_s |X n ball_s|
    pts_pad0 = core._pad_rows(shard_raw[torch.argsort(shard_raw[:, axis])], dp)
    search0 = pts_pad0[:, axis].contiguous()
    lo0 = torch.searchsorted(search0, (centers[:, axis] - radii).contiguous(), right=False)
    hi0 = torch.searchsorted(search0, (centers[:, axis] + radii).contiguous(), right=True)
    cnt0 = torch.zeros(S, dtype=torch.int32, device=dev)
    _native.check(lib.flooder_ball_count_f32(_native.ptr(pts_pad0), pts_pad0.shape[0], w["dim"], dp,
                                             _native.ptr(centers.contiguous()), _native.ptr(radii.contiguous()),
                                             _native.ptr(lo0), _native.ptr(hi0), S, _native.ptr(cnt0),
                                             _native.current_stream_ptr(dev)), "ball_count")
    P_local = int(cnt0.sum().item())
    del pts_pad0, search0, lo0, hi0, cnt0
    stats = torch.zeros(16, dtype=torch.int64, device=dev)
    plan = core.SamplePlan(weights, faces)

    def build_index():
        if args.method in ("cell", "bvh"):
            return core.PointIndex(shard_raw)
        o = torch.argsort(shard_raw[:, axis])
        pts_pad = core._pad_rows(shard_raw[o], dp)
        return (pts_pad, pts_pad[:, axis].contiguous())

    # the sort / index build: once per flood_complex call, like the reference's argsort (timed on its own)
    index = build_index()
    torch.cuda.synchronize()
    t_i0 = time.perf_counter()
    for _ in range(5):
        index = build_index()
    torch.cuda.synchronize()
    ms_index = (time.perf_counter() - t_i0) / 5 * 1e3

    def step(timer=None, with_stats=False):
        """indexed cloud + simplices in HBM -> per-face filtration values in HBM (t_sweep).
        The work counters are collected by ONE extra, untimed step: flood_complex() never asks for them, and
        thousands of waves adding to the same few words cost ~0.1 ms."""
        core.LAST_STATS.reset()
        st = stats if with_stats else None
        if args.method == "cell":
            stats.zero_()
            out, _ = core._sweep_dimension_cell(index, verts, weights, faces, hook, timer=timer, stats=st, plan=plan)
        elif args.method == "bvh":
            stats.zero_()
            out, _ = core._sweep_dimension_bvh(index, verts, weights, faces, hook, timer=timer,
                                               stats=None if st is None else st[:4], plan=plan)
        else:
            out, _ = core._sweep_dimension_hip(index[0], index[1], axis, w["dim"], verts, centers, radii, weights,
                                               faces, hook, timer=timer)
        if mine is not None:  # simplex sharding: every rank ends with all (S_all, F) values
            full = torch.full((S_all, out.shape[1]), float("inf"), dtype=out.dtype, device=dev)
            full[mine] = out
            with core._span(timer, "reduce"):
                face_hook(full)
            out = full
        return out

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:  # create the communicator outside the timed region even with --warmup 0
        dist.all_reduce(torch.zeros(1, device=dev), op=dist.ReduceOp.MIN)
    for _ in range(max(args.warmup, 0)):
        out = step()
    sync_all()
    timer = core._KernelTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(timer)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = w["n"] * S_all / (elapsed / args.steps) / 1e6

    out = step(None, with_stats=True)  # untimed: work counters for the report
    torch.cuda.synchronize()

    # ------------------------------------------------------------------ per-kernel numbers
    k_ms = timer.totals_ms()
    k_n = timer.counts()
    sweep_ms = k_ms["sweep"] / k_n["sweep"]
    slab_local = core.LAST_STATS.slab_points
    # algorithmic bytes of one sweep launch on this rank: every candidate row read
    # once per simplex, vertices, weights, and the (S, R) minimum buffer written once
    alg_bytes = P_local * w["dim"] * 4 + S * (d + 1) * w["dim"] * 4 + R * (d + 1) * 4 + S * R * 4
    achieved_gbs = alg_bytes / (sweep_ms * 1e-3) / 1e9
    pair_evals = P_local * R                       # what the reference's formulation evaluates
    if args.method == "bvh":
        st_h = stats.cpu().tolist()
        ks = args.bvh_ks or (2 if R > 64 else 1)
        done_evals = st_h[0] * 16 * 64 * ks        # leaves evaluated x 16 points x tile samples
        st_h = {"leaves_evaluated": st_h[0], "leaves_tested": st_h[1], "nodes_expanded": st_h[2]}
    elif args.method == "cell":
        st_h = stats.cpu().tolist()
        done_evals = st_h[0] + st_h[9] * 16 * 64
        st_h = {"cell_pairs": st_h[0], "points_staged": st_h[1], "tiles_flagged": st_h[2],
                "restage_rounds": st_h[3], "tiles_total": S * ((R + 63) // 64),
                "chunks_total": S * ((R + 255) // 256),
                "giveup_gather_density": st_h[4], "giveup_gather_stage": st_h[5], "giveup_lds_full": st_h[6],
                "giveup_doublings": st_h[7], "exhaustive_rounds": st_h[8],
                "fallback_leaves_evaluated": st_h[9], "fallback_leaves_tested": st_h[10],
                "fallback_nodes_expanded": st_h[11], "fallback_max_tests_one_tile": st_h[12],
                "fine_rows_swept": st_h[13], "fine_rows_total": st_h[14]}
    else:
        st_h = None
        done_evals = pair_evals
    valu_tflops = 10.0 * done_evals / (sweep_ms * 1e-3) / 1e12  # 3d+1 flop per evaluated pair, d = 3

    # measured HBM traffic of the dominant kernel (rocprofv3 PMC passes, tools/collect_profiles.sh ->
    # profiles/traffic.json; FETCH_SIZE doubled for 16 B/lane loads as MI355X_MICROARCH.md prescribes)
    traffic = None
    try:
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(f"{args.workload}:{args.method}", {}).get("bytes_per_launch")
    except Exception:
        traffic = None
    result = {
        "metric": "M points×simplices/s (coverage sweep)",
        "value": round(value, 3),
        "unit": "M points×simplices/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "ms_index_build": round(ms_index, 3),
        "value_including_index_build": round(w["n"] * S_all / ((ms_per_step + ms_index) * 1e-3) / 1e6, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": w["desc"], "points": w["n"], "landmarks": w["n_lms"], "top_simplices": S_all,
            "top_simplices_rank0": S,
            "samples_per_simplex": R, "candidate_pairs_rank0": P_local, "pair_evals_rank0": pair_evals,
            "ball_tests_rank0": slab_local, "parallelism": (f"{args.shard}-shard x{world}" if world > 1 else "single GPU"),
            "method": args.method, "pair_evals_done_rank0": done_evals,
            "sweep_stats_rank0": st_h,
            "sweep_stats_note": "work counters come from one extra untimed step (the timed steps run without them, as flood_complex does)",
        },
        "roofline": {
            "kernel": {"cell": "cell_sweep_kernel", "bvh": "sweep_bvh_kernel", "ball": "sweep_kernel"}[args.method], "bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
            "algorithmic_bytes": alg_bytes, "avg_launch_ms": round(sweep_ms, 4),
            "note": "algorithmic bytes = reference candidate pairs P x 4*dim + vertices + weights + (S,R) minima "
                    "(SURVEY.md 8d); the kernel itself is fp32-VALU/latency-bound, see valu",
            "valu": {"achieved": round(valu_tflops, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4), "flop_per_pair": 10},
        },
        "emulated_shard": args.emulate_shard,
        "kernels_ms_per_step": {k: round(v / args.steps, 4) for k, v in k_ms.items()},
        # landmark selection (generate_landmarks, outside the step): one distance-update + arg-max sweep of
        # the cloud per landmark; algorithmic bytes = (4*dim + 8) B per point and iteration
        "fps": {"points": w["n"], "landmarks": w["n_lms"], "ms": round(t_fps * 1e3, 3),
                "us_per_landmark": round(t_fps / w["n_lms"] * 1e6, 3),
                "algorithmic_GBs": round((4 * w["dim"] + 8) * w["n"] * w["n_lms"] / t_fps / 1e9, 1),
                "hbm_peak_GBs": HBM_PEAK_GBS},
    }

    # ------------------------------------------------------------------ CPU baseline + parity (rank 0, N=1)
    if world == 1 and not args.no_cpu_baseline:
        from oracle import flood_oracle as fo

        cb = fo.kdtree_sweep_sample(pts_cpu.numpy(), lms.cpu().numpy(), simp.cpu().numpy(), w["ppe"], d,
                                    n_sample=min(args.cpu_sample, S_all), seed=0, workers=1)
        got = out.cpu().numpy()[cb["picked"]]
        ref = cb["face_max"]
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6 * float(pts_cpu.abs().max()))
        cpu_value = w["n"] * cb["n_sample"] / (cb["query_s"] + cb["build_s"] * cb["n_sample"] / S_all) / 1e6
        result["cpu_baseline"] = {
            "value": round(cpu_value, 4), "unit": "M points×simplices/s", "cores": 1, "kind": "port",
            "sample": f"oracle kd-tree sweep (scipy KDTree.query, workers=1, as reference core.py:197-199) of "
                      f"{cb['n_sample']} of {S_all} tetrahedra x {R} samples; tree build {cb['build_s']:.2f}s "
                      f"(charged pro rata), query {cb['query_s']:.2f}s; host has {os.cpu_count()} cores",
        }
        result["parity"] = {"checked_simplices": int(cb["n_sample"]), "values": int(got.size),
                            "max_abs_err": float(np.abs(got - ref).max()),
                            "max_rel_err": float(rel.max())}

    if rank == 0:
        print(json.dumps(result, ensure_ascii=False))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
