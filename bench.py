#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: aligned pairs/s (+GCUPS), all-vs-all 64 x 5 kb (configs[1], "C2").
Hot path = SeqRush::new UF state -> (orientation ->) all-vs-all biWFA alignment -> match-run extraction ->
union-find unite (-> label merge across GPUs).  One step = one pass of that path over the whole pair list with the
packed sequences already resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5]
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Rank 0 prints ONE JSON line.  The pair list is sharded over ranks (cost-balanced; strong scaling: the list is
fixed); per-rank forests are merged with one RCCL all-gather of canonical u32 labels + replay-unite.

roofline (dominant kernel = the alignment kernel), all per launch, everything measured in THIS run unless it sits under
`recorded`:
  wf_level_diagonals   (score level, diagonal) units the kernel computed; every unit is five component cells (M I1 D1 I2 D2)
  wf_cells             the same in SURVEY 8(d) units: sum over levels and components of the range width = 5 x the above
                       (3 x for one-piece penalties)
  achieved / frac      bytes.rows_counted -- what the kernel really moved as wavefront rows, counted on the device lane
                       access by lane access -- / the kernel's average launch time (hipEvents on its stream) / 8 TB/s
  model_GBps / _frac   SURVEY 8(d)'s model of a ring that lives in HBM -- wf_cells x cell bytes x (1 write + ~1 effective
                       read) -- over the same time.  The blocked tile moves fewer bytes than the model, so this can exceed 1
  bytes.tile_ideal     what the blocked tile needs at least: (row loads + row stores of one B-level tile) x cell bytes
                       per level-diagonal (B = 10: 26 + 16 rows x 2 B / 10 levels = 8.4 B), x wf_level_diagonals
  bytes.lds_rows       the part of the rows that stayed in LDS (LDS-resident history of small base cases), not in rows_counted
  recorded             the newest rocprofv3 PMC summary (profiles/rNN_counters.json) taken for the same kernel name on
                       this workload: HBM-side traffic (FETCH_SIZE x 2.00 + WRITE_SIZE x 1.00, profiles/r02_calibration.json),
                       VALU issue and wait fractions, with the kernel time, box and commit they were measured at;
                       `traffic` repeats recorded.traffic (null when nothing matches)
  bound / bound_basis  the resource closest to what it can attain (hbm: traffic behind the L2 or device-counted rows against the
                       6.3 TB/s of a streaming copy; valu-issue: against 0.39 / cycle / SIMD) when within a quarter of it, else
                       "latency"; computed from the numbers above (never a literal)
  unite                the second kernel: united bases x 24 B (2 parent loads + 1 CAS of 8-byte nodes) / its time
h2h_ms = SURVEY 8(d)'s `t` (host sequences -> host UF array: pack + upload + step + download), outside `value`.
"""
import argparse
import json
import os
import statistics
import sys
import time
import glob
import threading

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s measured for streaming copies)


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota (cpu.max)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    return max(1, min(n, int(quota) if quota and quota >= 1 else n)), n, quota


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(recs, pairs, total_pairs_label, budget_s=24.0):
    """the oracle ("port": C restatement, OpenMP over pairs, one aligner arena per thread,
    lock-free UF) on the host, legs -t 1, -t 4 (the reference's default, src/seqrush.rs:37) and all usable cores
    (cgroup quota honoured), threads confined to as many distinct cores.  Task shape = the reference's: one task per
    ordered pair (src/seqrush.rs:738).  Every leg aligns a prefix of the SAME ordered pair list the GPU runs.
    Round 4: every leg is warmed first (its threads' arenas are touched and grown by an untimed run of 4 pairs per
    thread, twice: round 3's -t 1 leg timed the first touch of its arena), the sample is >= 16 pairs per thread and sized from
    the warm-up's rate so that a leg's three timed runs take about budget_s / legs seconds (the whole list when it fits),
    and a leg whose per-thread rate is more than 20 % off the median of the legs is flagged."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    usable, affinity_n, quota = usable_cpus()
    all_cpus = sorted(os.sched_getaffinity(0))
    legs = sorted({1, min(4, usable), usable})
    out_legs = {}
    saved = os.sched_getaffinity(0)

    def run(t, sample):
        o = ob.OracleSeqRush(records=recs)
        p = ob.default_params()
        p.threads = t
        t0 = time.perf_counter()
        done, cells = o.align_and_unite_list(p, sample)
        dt = time.perf_counter() - t0
        o.close()
        return done, cells, dt
    try:
        for t in legs:
            os.sched_setaffinity(0, set(all_cpus[:t]))
            nwarm = min(len(pairs), 4 * t)
            # (self pairs cost nothing: skip them when sizing from the warm-up, keep them in the sample -- they are
            # part of the list the GPU runs)
            run(t, pairs[:nwarm])                                  # first touch of the arenas, untimed
            done, _, dtw = run(t, pairs[:nwarm])
            rate = done / dtw if dtw > 0 else 1.0
            leg_s = budget_s / len(legs)
            want = int(rate * leg_s / 3.0)
            lo = 16 * t                                            # >= 16 pairs per thread ...
            if 3.0 * lo / rate > 3.0 * leg_s:                      # ... unless that alone is three budgets (C5: seconds per pair)
                lo = max(t, want)
            npairs = min(len(pairs), max(lo, want))
            sample = pairs[:npairs]
            vals, cells_v = [], []
            for _ in range(3):
                done, cells, dt = run(t, sample)
                vals.append(done / dt); cells_v.append(cells / dt / 1e9)
            med = statistics.median(vals)
            out_legs[f"t{t}"] = {"threads": t, "pairs_per_s": med, "pairs_per_s_per_thread": med / t,
                                 "gcups": statistics.median(cells_v), "runs": vals, "sample_pairs": npairs,
                                 "warmup_pairs": 2 * nwarm, "min_16_per_thread": npairs >= min(len(pairs), 16 * t), "whole_list": npairs == len(pairs)}
    finally:
        os.sched_setaffinity(0, saved)
    ptr = statistics.median(d["pairs_per_s_per_thread"] for d in out_legs.values())
    for d in out_legs.values():
        d["per_thread_vs_median_of_legs"] = d["pairs_per_s_per_thread"] / ptr if ptr > 0 else None
        d["deviates_over_20pct"] = bool(ptr > 0 and abs(d["pairs_per_s_per_thread"] / ptr - 1.0) > 0.20)
    best = max(out_legs.values(), key=lambda d: d["pairs_per_s"])
    return {"value": best["pairs_per_s"], "unit": "pairs/s", "cores": best["threads"], "kind": "port",
            "sample": f"first {best['sample_pairs']} ordered pairs of the same list ({total_pairs_label}), legs warmed first "
                      f"(untimed run), median of 3 runs per leg, oracle/ C restatement (OpenMP dynamic,1 over pairs = one rayon "
                      f"task per pair, per-thread arenas, 64-bit word extension, vectorised score-only step), threads "
                      f"confined to {best['threads']} cpus; NOT the seqrush binary (no Rust toolchain / crates here)",
            "legs": out_legs,
            "host": {"cpu_model": cpu_model(), "logical_cpus_visible": affinity_n, "cgroup_cpu_quota": quota,
                     "usable_cpus": usable}}


def host_stages(recs, ctx):
    """the stages either side of the timed path, reported separately and never part of `value`:
    FASTA parse (host loader, src/seqrush.rs:1801-1837), graph induction on the device + GFA text
    (src/bidirected_builder.rs:17-289), compaction + renumber (host C++, src/bidirected_ops.rs:75-490) and the file write.
    One measurement each on the union-find the timed steps left behind."""
    import tempfile
    from seqrush_amd.seqrush import load_sequences
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "in.fa")
        with open(fa, "wb") as fh:
            for n, sq in recs:
                fh.write(b">" + n.encode() + b"\n" + sq + b"\n")
        t0 = time.perf_counter(); seqs = load_sequences(fa); out["fasta_parse"] = (time.perf_counter() - t0) * 1e3
        assert len(seqs) == len(recs)
        t0 = time.perf_counter(); gfa, nn, ne = ctx.build_gfa(compact=False); t_ind = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); gfa_c, nc, ec = ctx.build_gfa(compact=True); t_cmp = (time.perf_counter() - t0) * 1e3
        out["induction_and_gfa_text"] = t_ind
        out["compaction_and_renumber"] = max(0.0, t_cmp - t_ind)        # the compact call repeats the induction
        t0 = time.perf_counter()
        with open(os.path.join(td, "out.gfa"), "w") as fh:
            fh.write(gfa_c)
        out["gfa_write"] = (time.perf_counter() - t0) * 1e3
        out["nodes_edges_uncompacted"] = [nn, ne]; out["nodes_edges_compacted"] = [nc, ec]
        out["gfa_bytes"] = len(gfa_c)
    return out


class Telemetry:
    """Shader clock and socket power of the card during the timed steps, sampled from the amdgpu hwmon files
    (freq1_input in Hz, power1_input in uW) by a thread of this process -- context for the several per cent by which
    boxes, and one box over a call, differ on a kernel whose distance from the memory ceiling is latency (clock-bound).  Only when
    exactly one card exposes the files (the one-GPU box); never part of the timed work (two small sysfs reads per 5 ms)."""

    def __init__(self, pci=None):
        f = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        if pci and len(f) > 1:                                   # several cards in sysfs: the one at the device's PCI address
            f = [x for x in f if os.path.basename(os.path.realpath(x.split("/hwmon/")[0])).lower() == pci.lower()]
        self.freq = f[0] if len(f) == 1 else None
        self.why = None if self.freq else f"{len(f)} cards expose hwmon freq1_input (pci {pci})"
        self.power = os.path.join(os.path.dirname(self.freq), "power1_input") if self.freq else None
        self.mhz, self.watt, self._stop, self._th = [], [], threading.Event(), None

    def _run(self):
        while not self._stop.is_set():
            try:
                with open(self.freq) as fh:
                    self.mhz.append(int(fh.read()) / 1e6)
                with open(self.power) as fh:
                    self.watt.append(int(fh.read()) / 1e6)
            except Exception:
                pass
            self._stop.wait(0.005)

    def start(self):
        if self.freq:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        if not self._th:
            return {"unavailable": self.why}
        self._stop.set()
        self._th.join()
        if not self.mhz:
            return None
        st = lambda v: [min(v), sum(v) / len(v), max(v)] if v else None
        return {"sclk_mhz_min_mean_max": st(self.mhz), "power_w_min_mean_max": st(self.watt), "samples": len(self.mhz),
                "source": os.path.dirname(self.freq)}


def host_to_host(ss, prm, dev, config):
    """SURVEY 8(d)'s `t`: sequences resident on the HOST -> union-find array resident on the HOST, i.e. what one call of
    the Seam-2 entry point costs a host that holds FASTA records: pack + upload (+ pair list, workspace sizing and
    allocation) + one step + download of the uf_rush node array.  Never part of `value` (the contract's timed region
    starts with the inputs in HBM); measured once on a fresh context after the timed steps (twice for the short
    configs: the second run has the allocator warm), plus the one-shot sr_align_and_unite() call that also creates and
    destroys its context."""
    import ctypes as C
    import numpy as np
    from seqrush_amd import _lib
    from seqrush_amd.seqrush import Context
    runs = []
    for _ in range(1 if config == "C5" else 2):
        c2 = Context(dev)
        t0 = time.perf_counter()
        c2.load(ss, prm)
        c2.sync()
        t1 = time.perf_counter()
        c2.run()
        c2.sync()
        t2 = time.perf_counter()
        nodes = c2.download_uf()
        c2.sync()
        t3 = time.perf_counter()
        runs.append({"load_ms": (t1 - t0) * 1e3, "step_ms": (t2 - t1) * 1e3, "uf_download_ms": (t3 - t2) * 1e3,
                     "total_ms": (t3 - t0) * 1e3, "uf_bytes": int(nodes.nbytes)})
        c2.close()
    best = min(runs, key=lambda r: r["total_ms"])
    res = dict(best)
    res["runs_total_ms"] = [r["total_ms"] for r in runs]
    res["load_is"] = "symbol map + packing + upload + pair list (+ sketches) + workspace sizing and hipMalloc + UF init"
    if config != "C5":
        L = _lib.load()
        out = np.zeros(2 * ss.total_length + 2, dtype=np.uint64)
        shots = []
        for _ in range(2):                                      # (a 25 GB hipMalloc costs 1 ms or 1 s, box and moment decide: both runs are kept)
            t0 = time.perf_counter()
            _lib.check(L.sr_align_and_unite(C.byref(ss.c), C.byref(prm.c), out.ctypes.data_as(C.POINTER(C.c_uint64))))
            shots.append((time.perf_counter() - t0) * 1e3)
        res["one_shot_sr_align_and_unite_ms"] = min(shots)
        res["one_shot_runs_ms"] = shots
    return res


def build_config(name, nseq, fasta=None):
    from seqrush_amd import synth
    if name == "C2":
        recs = synth.config_c2(nseq)
        return recs, "none", (f"C2: {nseq} x 5 kb synthetic (5% SNP, seed 2001), all-vs-all incl. self = {nseq * nseq} "
                              f"ordered pairs"), f"aligned pairs/sec all-vs-all {nseq}x5kb"
    if name == "C3":
        fasta = fasta or os.environ.get("SR_C3_FASTA")
        if fasta:
            # the real HLA-zoo DRB1 gene set (BASELINE.json configs[2]) for a user who has the file: the reference's path
            # convention is HLA-zoo/seqs/DRB1-3123.fa (src/bin/test_range_paf.rs:34); an empty submodule in this pipeline
            if not os.path.exists(fasta):
                raise SystemExit(f"--fasta / SR_C3_FASTA: {fasta} does not exist")
            from seqrush_amd.seqrush import load_sequences
            recs = [(s.id, bytes(s.data)) for s in load_sequences(fasta)]
            n = len(recs)
            return recs, "none", (f"C3: HLA-zoo DRB1 gene set from {os.path.basename(fasta)} ({n} sequences, "
                                  f"{sum(len(r[1]) for r in recs)} bp), all-vs-all incl. self = {n * n} ordered pairs"), \
                f"aligned pairs/sec all-vs-all HLA-zoo DRB1 ({n} sequences)"
        recs = synth.config_c3_surrogate()
        return recs, "none", ("C3 surrogate: 12 x ~14 kb (3% substitutions, 0.3% indels, two 200-800 bp insertions each; "
                              "HLA-zoo DRB1 is not in the container), all-vs-all incl. self = 144 ordered pairs"), \
            "aligned pairs/sec all-vs-all C3 surrogate 12x14kb"
    if name == "C4":
        recs = synth.config_c4()
        return recs, "tree:3,3,0.1", "C4: 1024 x 2 kb synthetic (16 clades), -x tree:3,3,0.1", \
            "aligned pairs/sec 1024x2kb tree:3,3,0.1"
    if name == "C5":
        recs = synth.config_c5(nseq)
        return recs, "none", (f"C5: {nseq} x 50 kb synthetic (2% substitutions, 0.1% indels, in-place inversions in 25%, 10% "
                              f"reverse-complemented), all-vs-all incl. self = {nseq * nseq} ordered pairs"), \
            f"aligned pairs/sec all-vs-all {nseq}x50kb"
    raise SystemExit(f"unknown --config {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--nseq", type=int, default=None, help="C2 / C5: number of sequences (default 64 / 256)")
    ap.add_argument("--fasta", default=None, help="C3 only: the real HLA-zoo DRB1 FASTA (also SR_C3_FASTA); default: the surrogate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2h", action="store_true", help="skip the host-to-host (pack + upload + step + download) measurement")
    ap.add_argument("--no-host-stages", action="store_true", help="skip the FASTA / induction / compaction / GFA-write timings")
    args = ap.parse_args()
    if args.nseq is None:
        args.nseq = 256 if args.config == "C5" else 64
    if args.steps is None:
        args.steps = 1 if args.config == "C5" else 5
    if args.warmup is None:
        args.warmup = 0 if args.config == "C5" else 1

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` started plainly: one process per GPU under torch.distributed.run, started as a
        # CHILD before anything in this process touches the GPU (never an exec); its stdout (rank 0's JSON line) and
        # exit code are relayed
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

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

    from seqrush_amd.seqrush import SeqSet, Params, Context

    recs, spars, workload, metric = build_config(args.config, args.nseq, args.fasta)
    ss = SeqSet(recs)
    prm = Params(sparsification=spars)
    prm.c.device = dev
    prm.c.shard_rank, prm.c.shard_count = rank, world
    ctx = Context(dev)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.load(ss, prm)                      # pack + upload (+ sketches): inputs resident before timing
    ufn = ctx.uf_size
    u32 = ufn < (1 << 32)                  # u32 labels while 2N+2 < 2^32
    ldt = torch.int32 if u32 else torch.int64
    lab = torch.empty(ufn, dtype=ldt, device="cuda") if world > 1 else None
    gathered = torch.empty(ufn * world, dtype=ldt, device="cuda") if world > 1 else None
    my_pairs = ctx.pairs()
    tp = torch.tensor([len(my_pairs), ctx.dp_cells], dtype=torch.float64)
    if world > 1:
        tpd = tp.to("cpu" if single_dev else "cuda")
        dist.all_reduce(tpd)
        tp = tpd.cpu()
    total_pairs, total_cells = int(tp[0].item()), float(tp[1].item())

    def step():
        ctx.reset_uf()
        ctx.run()                          # (orientation +) alignment + unite, batch after batch
        if world > 1:
            (ctx.labels_device_u32 if u32 else ctx.labels_device)(lab.data_ptr())
            if single_dev:
                parts = [torch.empty(ufn, dtype=ldt) for _ in range(world)]
                dist.all_gather(parts, lab.cpu())
                gathered.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(gathered, lab)
            (ctx.merge_labels_u32 if u32 else ctx.merge_labels)(gathered.data_ptr(), world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.sync()
    tel = None
    if world == 1:
        try:
            pr = torch.cuda.get_device_properties(dev)
            pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pci = None
        tel = Telemetry(pci)
    if tel:
        tel.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    telemetry = tel.stop() if tel else None
    ctx.sync()                              # raises on device fault bits
    # per-kernel durations (hipEvents on the context's stream): for the long configs the timed steps' own events
    # are read; otherwise K more untimed steps with event reads keep the timed region free of host syncs
    align_ms, unite_ms, orient_ms = [], [], []
    reread = args.config != "C5"
    for _ in range(args.steps if reread else 1):
        if reread:
            step()
            torch.cuda.synchronize()
        align_ms.append(ctx.kernel_ms(0))
        unite_ms.append(ctx.kernel_ms(1))
        try:
            orient_ms.append(ctx.kernel_ms(4))          # orientation runs as its own kernel (sr_orient_kernel)
        except Exception:
            pass
    cnt = ctx.counters()
    rep = ctx.workspace_report()
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
        ori_cells = cnt["ticks_orientation"] if orient_ms else 0      # [6] = the orientation kernel's level-diagonals
        ldiag = cnt["wf_cells"] - ori_cells                           # (level, diagonal) units of the alignment kernel
        ncomp = 5 if rep.get("two_piece", 1) else 3
        cell_b = int(rep.get("ring_cell_bytes", 2))
        cells_8d = ldiag * ncomp                                      # sum over levels and components
        alg_bytes = cells_8d * cell_b * 2                             # 8(d): 1 write + ~1 effective read per cell
        rows_counted = cnt["row_bytes_loaded"] + cnt["row_bytes_stored"]
        lds_rows = cnt.get("lds_row_bytes", 0)
        B = int(rep.get("block_levels", 1))
        # row loads + stores of one B-level tile of the blocked kernel (sr_align_blk.inc): exact 10-level instance
        # 26 + 16, generic 5-level instance 21 + 12; level-per-pass kernels: 7 source rows + 5 stored per level
        tile_rows = {10: 42, 5: 33}.get(B, 12 * B)
        tile_ideal = ldiag * tile_rows * cell_b / max(B, 1)
        sec = a_ms * 1e-3
        model_GBps = alg_bytes / sec / 1e9 if sec > 0 else 0.0
        rows_GBps = rows_counted / sec / 1e9 if sec > 0 else 0.0
        # Headline (round 4): achieved = the row bytes the kernel moved, counted on the device lane access
        # by lane access in this run (counters[16..17]), / its hipEvent time.  The SURVEY 8(d) ring model (every cell written
        # once + ~1 effective read) is kept as model_*: the blocked tile moves fewer bytes than that model, so model_frac
        # can exceed what the memory system did (C5: > 1) and is not a roofline for this kernel.
        roof = {"bound": None, "kernel": ctx.align_kernel, "achieved": rows_GBps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": rows_GBps / HBM_PEAK_GBS, "traffic": None, "kernel_ms": a_ms,
                "kernel_ms_min_max": [min(align_ms), max(align_ms)],      # (boxes and their thermal state differ by several %)
                "achieved_is": "row bytes counted on the device in this run (bytes.rows_counted) / kernel time (hipEvents on "
                               "the context's stream); model_frac prices SURVEY 8(d)'s ring-in-HBM model instead",
                "model_GBps": model_GBps, "model_frac": model_GBps / HBM_PEAK_GBS,
                "wf_level_diagonals": ldiag, "wf_cells": cells_8d, "components": ncomp, "cell_bytes": cell_b,
                "bytes": {"algorithmic_8d": alg_bytes, "tile_ideal": tile_ideal, "rows_counted": rows_counted,
                          "rows_loaded": cnt["row_bytes_loaded"], "rows_stored": cnt["row_bytes_stored"],
                          "lds_rows": lds_rows,
                          "rows_over_tile_ideal": rows_counted / tile_ideal if tile_ideal else None},
                "rows_GBps": rows_GBps, "rows_frac": rows_GBps / HBM_PEAK_GBS,
                "bytes_per_level_diagonal": rows_counted / ldiag if ldiag else None,
                "wf_cells_per_s": cells_8d / sec if sec > 0 else None,
                "level_diagonals_per_s": ldiag / sec if sec > 0 else None,
                "orient_kernel_ms": o_ms, "orient_level_diagonals": ori_cells,
                "unite": None}
        if rep.get("fused_unite"):
            # since round 4 the blocked alignment kernel unites a pair's match runs right after its CIGAR (sr_ctx_run):
            # no unite kernel, its atomics are inside roofline.kernel_ms
            roof["unite"] = {"kernel": None, "fused_into": ctx.align_kernel, "kernel_ms": 0.0,
                             "united_bases": cnt["united_bases"], "bytes": cnt["united_bases"] * 24}
        else:
            roof["unite"] = {"kernel": "sr_unite_kernel", "kernel_ms": u_ms, "united_bases": cnt["united_bases"],
                             "bytes": cnt["united_bases"] * 24,
                             "achieved": cnt["united_bases"] * 24 / (u_ms * 1e-3) / 1e9 if u_ms > 0 else None,
                             "unit": "GB/s", "bound": "hbm (random access, atomics)",
                             "frac": cnt["united_bases"] * 24 / (u_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if u_ms > 0 else None}
        # PMC passes cannot run inside this process: the newest recorded summary (profiles/rNN_counters.json, written by
        # scripts/profile_round.sh) is attached under ONE object, `recorded`, and only when it was taken for this kernel
        # name on this workload; it carries the kernel time, box and commit it was measured at
        rec = None
        if world == 1 and args.config == "C2" and args.nseq == 64:
            import glob
            for ppath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_counters.json")), reverse=True):
                try:
                    pmc = json.load(open(ppath))
                except Exception:
                    continue
                if ctx.align_kernel and ctx.align_kernel in str(pmc.get("align_kernel", "")) and pmc.get("align_hbm_bytes_per_launch") \
                        and pmc.get("align_kernel_instance", rep.get("kernel_instance")) == rep.get("kernel_instance"):
                    rec = {"file": os.path.relpath(ppath, ROOT), "kernel": pmc.get("align_kernel"),
                           "kernel_ms": pmc.get("align_kernel_ms"), "host": pmc.get("host"), "git": pmc.get("git"),
                           "traffic": pmc["align_hbm_bytes_per_launch"],
                           "traffic_GBps": pmc["align_hbm_bytes_per_launch"] / (pmc["align_kernel_ms"] * 1e-3) / 1e9 if pmc.get("align_kernel_ms") else None,
                           "traffic_over_rows_counted": pmc["align_hbm_bytes_per_launch"] / rows_counted if rows_counted else None,
                           "dram_bytes": pmc.get("align_dram_bytes_per_launch"),
                           "valu_issue_frac": pmc.get("align_valu_issue_frac"), "wait_any_frac": pmc.get("align_wait_any_frac"),
                           "valu_insts": pmc.get("align_valu_insts"), "salu_insts": pmc.get("align_salu_insts")}
                    break
        roof["recorded"] = rec
        if rec:
            roof["traffic"] = rec["traffic"]
        # `bound`: whichever resource of THIS run sits closest to what it can attain.  hbm: device-counted rows / time or -- when
        # a PMC pass of this kernel is on record -- the traffic it measured behind the L2, whichever is larger, against the
        # 6.3 TB/s a streaming copy sustains (MI355X_MICROARCH.md); issue: the recorded VALU-issue fraction against the 0.39
        # wave-instructions / cycle / SIMD four resident integer waves attain (scripts/calib) -- known only from a
        # PMC pass, so it takes part only when `recorded` matches this kernel; a kernel near neither is "latency".
        # Ceilings a kernel can attain, not data-sheet peaks: 6.3 TB/s is the guide's measured streaming copy (HBM3E 8 TB/s spec),
        # 0.39 wave-instructions / cycle / SIMD what four resident integer waves issue.  `frac` itself stays against the 8 TB/s peak.
        HBM_ATTAINABLE_GBS = 6300.0
        closeness = {"hbm": rows_GBps / HBM_ATTAINABLE_GBS}
        basis = {"hbm_frac_of_peak": rows_GBps / HBM_PEAK_GBS, "hbm_frac_of_streaming_copy_6300": rows_GBps / HBM_ATTAINABLE_GBS}
        if rec and rec.get("traffic_GBps"):
            # what the memory system behind the L2 carried (DRAM-destined requests incl. Infinity-Cache hits: gfx950 has no
            # counter that separates them), from the recorded PMC pass at ITS kernel time
            basis["recorded_traffic_frac_of_peak"] = rec["traffic_GBps"] / HBM_PEAK_GBS
            basis["recorded_traffic_frac_of_streaming_copy_6300"] = rec["traffic_GBps"] / HBM_ATTAINABLE_GBS
            closeness["hbm"] = max(closeness["hbm"], rec["traffic_GBps"] / HBM_ATTAINABLE_GBS)
        if rec and rec.get("valu_issue_frac") is not None:
            basis["valu_issue_frac_of_peak_0.5"] = rec["valu_issue_frac"]
            basis["valu_issue_frac_of_attainable_0.39"] = rec["valu_issue_frac"] * 0.5 / 0.39
            basis["wait_any_frac"] = rec.get("wait_any_frac")
            closeness["valu-issue"] = rec["valu_issue_frac"] * 0.5 / 0.39
        top = max(closeness, key=closeness.get)
        # The resource closest to what it can attain names the bound when it is within a quarter of it; otherwise the kernel is
        # bound by the latency of its dependent steps.  C2 (round 4): the traffic behind the L2 runs at 0.79 of a streaming copy
        # while the waves are parked half of their cycles -- a latency-bandwidth curve, T(pair) = F + S * (workgroups per CU) with
        # S = the pair's bytes at ~6.5 TB/s and F = its serial critical path (more bytes cost time in
        # proportion -- eager I/D rows +35 % bytes, +38 % -- while bytes that hit the caches, instructions and cheap tiles do not).
        if closeness[top] >= 0.75:
            roof["bound"] = top
            basis["note"] = ("%s at %.2f of what it can attain (hbm: 6.3 TB/s streaming copy; valu-issue: 0.39 / cycle / SIMD)%s"
                             % (top, closeness[top], "; the waves are parked %.0f %% of their cycles: the rest of the way to the ceiling is "
                                "latency (a pair's serial sections)" % (100.0 * rec["wait_any_frac"]) if rec and rec.get("wait_any_frac") else ""))
        else:
            roof["bound"] = "latency"
            basis["note"] = ("no resource within a quarter of what it can attain (closest: %s at %.2f%s): latency of dependent steps"
                             % (top, closeness[top], "" if rec else "; no PMC pass recorded for this workload, device-counted rows only"))
        roof["bound_basis"] = basis
        out = {
            "metric": metric, "value": total_pairs * args.steps / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "i16" if rep.get("offset_bytes") == 2 else "i32", "data": "synthetic",
            "gcups": total_cells * args.steps / dt / 1e9,
            "config": {"workload": workload + ", -k 0 -S 0,5,8,2,24,1 --orientation-scores 0,1,1,1, biWFA (Ultralow)",
                       "pairs_total": total_pairs, "pairs_per_gpu": ctx.num_pairs,
                       "parallelism": f"pair-shard x{world} (cost-balanced), u32 label all-gather" if world > 1 else "pair-shard x1",
                       "workspace": rep},
            "roofline": roof,
            "kernels": cnt,
        }
        if labels_sha:
            out["labels_sha256"] = labels_sha
        if telemetry:
            out["telemetry"] = telemetry
        if world == 1 and not args.no_host_stages:
            out["host_stages_ms"] = host_stages(recs, ctx)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(recs, my_pairs, workload.split(":")[0])
            out["speedup_vs_cpu_best_leg"] = out["value"] / out["cpu_baseline"]["value"]
    ctx.close()
    if out is not None and world == 1 and not args.no_h2h:
        out["h2h"] = host_to_host(ss, prm, dev, args.config)
        out["h2h_ms"] = out["h2h"]["total_ms"]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # a rank that fails before a collective must not leave the others waiting in it until the RCCL timeout:
        # report, then leave at once with a non-zero code -- torch.distributed.run ends the remaining ranks
        import traceback
        traceback.print_exc()
        sys.stderr.flush(); sys.stdout.flush()
        os._exit(1) if int(os.environ.get("WORLD_SIZE", "1")) > 1 else sys.exit(1)
