"""time one rank's share of C2 for shard counts 1..8 on one GPU (what each GPU of an N-GPU run does, minus the
label exchange): ms per pass for the workgroup shapes / orientation modes the host could pick"""
import os, sys, time, itertools, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime order, see tests/conftest.py)
import __graft_entry__ as ge; ge.build()
from seqrush_amd import synth
from seqrush_amd.seqrush import SeqSet, Params, Context
recs = synth.config_c2(64); ss = SeqSet(recs)
out = []
ENVS = [{}, {"SR_PREORIENT": "1"}, {"SR_PREORIENT": "0"}, {"SR_ALIGN_THREADS": "256"}, {"SR_ALIGN_THREADS": "512"},
        {"SR_ALIGN_THREADS": "512", "SR_PREORIENT": "1"}, {"SR_ALIGN_THREADS": "256", "SR_PREORIENT": "1"}]
if len(sys.argv) > 1:                      # extra variants: "K=V,K=V;K=V"
    ENVS = [{}] + [dict(kv.split("=") for kv in grp.split(",")) for grp in sys.argv[1].split(";")]
for world in (8, 4, 2):
    for env in ENVS:
        old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
        p = Params(); p.c.shard_rank, p.c.shard_count = 0, world
        ctx = Context(0); ctx.load(ss, p)
        for _ in range(2):
            ctx.reset_uf(); ctx.run(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.reset_uf(); ctx.run()
        ctx.sync(); dt = (time.perf_counter() - t0) / 5 * 1e3
        rep = ctx.workspace_report(); ctx.close()
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        rec = {"shards": world, "pairs": rep["pairs"], "env": env, "ms": round(dt, 2), "threads": rep["threads_per_workgroup"], "wgs": rep["workgroups"]}
        print(json.dumps(rec), flush=True); out.append(rec)
