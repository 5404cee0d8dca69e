#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: aligned pairs/s (+GCUPS), all-vs-all
64 x 5 kb (configs[1], "C2"), hot path = SeqRush::new UF state -> all-vs-all
biWFA alignment -> match-run extraction -> union-find unite (-> label merge
across GPUs).  One step = one pass of that path over the whole pair list with
the packed sequences already resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Rank 0 prints ONE JSON line.  The pair list is sharded over ranks (strong
scaling: the n^2 list is fixed); per-rank forests are merged with one RCCL
all-gather of canonical labels + replay-unite.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_CELL_2P = 28    # 9 loads + 5 stores of 2-byte offsets per (score,diagonal) cell
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(recs, sample_pairs):
    """oracle ("port") timed on the host cores over a bounded sample of the same pair list"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    o = ob.OracleSeqRush(records=recs)
    p = ob.default_params()
    p.threads = cores
    t0 = time.perf_counter()
    done, cells = o.align_and_unite(p, 0, sample_pairs)
    dt = time.perf_counter() - t0
    o.close()
    return {"value": done / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"first {done} ordered pairs (row-major) of the same 64x5kb set, "
                      f"{dt:.1f} s wall, oracle/ C restatement with OpenMP over pairs",
            "gcups": cells / dt / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nseq", type=int, default=64)
    ap.add_argument("--cpu-sample-pairs", type=int, default=0,
                    help="pairs in the CPU-baseline sample (0 = 8 per host core, at least 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus}")
    if local_rank == 0:
        ge.build()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: seqrush_amd has no CPU fallback")
    # SR_BENCH_SINGLE_DEVICE=1 (testing only): all ranks share GPU 0 and talk over gloo, so the
    # sharded path can be exercised end to end on a one-GPU box
    single_dev = os.environ.get("SR_BENCH_SINGLE_DEVICE") == "1"
    dev = 0 if single_dev else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if single_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        dist.barrier()

    from seqrush_amd import synth
    from seqrush_amd.seqrush import SeqSet, Params, Context

    recs = synth.config_c2(args.nseq)
    ss = SeqSet(recs)
    prm = Params()
    prm.c.device = dev
    prm.c.shard_rank, prm.c.shard_count = rank, world
    ctx = Context(dev)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.load(ss, prm)                      # pack + upload: inputs resident before timing
    ufn = ctx.uf_size
    lab = torch.empty(ufn, dtype=torch.int64, device="cuda")
    gathered = torch.empty(ufn * world, dtype=torch.int64, device="cuda") if world > 1 else None
    total_pairs = args.nseq * args.nseq
    total_cells = sum(len(a[1]) for a in recs) ** 2

    def step():
        ctx.reset_uf()
        ctx.align()
        ctx.unite()
        if world > 1:
            ctx.labels_device(lab.data_ptr())
            if single_dev:
                parts = [torch.empty(ufn, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(parts, lab.cpu())
                gathered.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(gathered, lab)
            ctx.merge_labels(gathered.data_ptr(), world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.sync()
    align_ms = []
    unite_ms = []
    orient_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # event timers are read after the timed region for the last step only;
        # per-step values are collected without host sync below
    fence()
    dt = time.perf_counter() - t0
    ctx.sync()                              # raises on device fault bits
    # per-kernel durations: re-run K untimed steps with event reads (keeps the
    # timed region free of host syncs)
    for _ in range(args.steps):
        step()
        torch.cuda.synchronize()
        align_ms.append(ctx.kernel_ms(0))
        unite_ms.append(ctx.kernel_ms(1))
        try:
            orient_ms.append(ctx.kernel_ms(4))          # orientation runs as its own kernel (sr_orient_kernel)
        except Exception:
            pass
    cnt = ctx.counters()
    labels_sha = None
    if os.environ.get("SR_BENCH_LABEL_SHA") == "1":
        import hashlib
        labels_sha = hashlib.sha256(ctx.download_labels().tobytes()).hexdigest()
    if world > 1:
        tdt = torch.tensor([dt], dtype=torch.float64, device="cpu" if single_dev else "cuda")
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt.item())
    ms_per_step = dt / args.steps * 1e3
    out = None
    if rank == 0:
        a_ms = sum(align_ms) / len(align_ms)
        u_ms = sum(unite_ms) / len(unite_ms)
        o_ms = sum(orient_ms) / len(orient_ms) if orient_ms else None
        # wavefront cells of the dominant kernel: the orientation kernel's cells are counted apart ([6])
        ori_cells = cnt["ticks_orientation"] if orient_ms else 0
        cells = cnt["wf_cells"] - ori_cells
        alg_bytes = cells * BYTES_PER_CELL_2P + ctx.num_pairs * (2 * 1250 * 2 + 8 * 1024)
        achieved = alg_bytes / (a_ms * 1e-3) / 1e9
        align_kernel_name = ctx.align_kernel
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if world == 1 and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("sr_align_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "aligned pairs/sec all-vs-all 64x5kb",
            "value": total_pairs * args.steps / dt,
            "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "gcups": total_cells * args.steps / dt / 1e9,
            "config": {"workload": f"C2: {args.nseq} x 5 kb synthetic (5% SNP, seed 2001), all-vs-all "
                                   f"incl. self = {total_pairs} ordered pairs, -k 0 -S 0,5,8,2,24,1 "
                                   f"--orientation-scores 0,1,1,1, biWFA (Ultralow)",
                       "pairs_per_gpu": ctx.num_pairs, "parallelism": f"pair-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": align_kernel_name, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": a_ms, "wf_cells_per_launch": cells,
                         "bytes_per_cell": BYTES_PER_CELL_2P, "unite_kernel_ms": u_ms,
                         "orient_kernel_ms": o_ms, "orient_wf_cells_per_launch": ori_cells,
                         "wf_cells_per_s": cells / (a_ms * 1e-3)},
            "kernels": cnt,
        }
        if labels_sha:
            out["labels_sha256"] = labels_sha
        if world == 1 and not args.no_cpu_baseline:
            ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            sample = args.cpu_sample_pairs or max(256, 8 * ncore)
            out["cpu_baseline"] = cpu_baseline(recs, min(sample, total_pairs))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
