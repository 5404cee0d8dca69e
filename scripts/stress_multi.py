"""Stress for workgroups that align many pairs one after the other: the blocked kernel (default) against the
level-per-pass kernel (SR_ALIGN_IMPL=1; independent tile code) on the same
inputs with few workgroups -- scores, strands and the union-find partition must agree (the number of device run-length
ops is not compared: kernels may split a run at a segment boundary, the CIGAR is the same).  usage: python scripts/stress_multi.py [rounds]"""
import sys, os, random, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT)
import torch  # noqa: F401
import numpy as np
from seqrush_amd import synth
from seqrush_amd.seqrush import SeqSet, Context, Params

def run(recs, env, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ss = SeqSet(recs); ctx = Context(0); ctx.load(ss, Params(**kw))
        ctx.run(); ctx.sync()
        sc, rv, co = ctx.pair_results(); lab = ctx.download_labels()
        ctx.close()
        return sc.copy(), rv.copy(), co.copy(), lab
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

bad = 0
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for r in range(first, rounds):
    rng = random.Random(4400 + r)
    n = rng.randint(6, 40); L = rng.choice([600, 1500, 3000, 5000])
    if n * n * L > 40 * 40 * 3000: n = max(6, int((40 * 40 * 3000 / L) ** 0.5))
    kind = rng.choice(["snp", "indel", "rc"])
    recs = (synth.snp_family(n, L, rng.choice([0.02, 0.05, 0.1]), 5000 + r) if kind == "snp" else
            synth.indel_family(n, L, 0.05, 0.02, 5000 + r) if kind == "indel" else
            synth.snp_family(n, L, 0.05, 5000 + r, rc_every=3))
    kw = {} if rng.random() < 0.7 else {"scores": rng.choice(["0,5,8,2", "0,6,9,2,30,1", "0,4,6,2,12,1"])}
    env = {"SR_NWG": str(rng.choice([1, 2, 3, 7, 16, 64]))}
    if rng.random() < 0.4: env["SR_ALIGN_THREADS"] = rng.choice(["256", "512"])
    if rng.random() < 0.3: env["SR_PREORIENT"] = rng.choice(["0", "1"])
    t0 = time.time()
    a = run(recs, env, **kw)
    b = run(recs, {"SR_ALIGN_IMPL": "1"}, **kw)
    diff = int(((a[0] != b[0]) | (a[1] != b[1])).sum())
    ok = diff == 0 and np.array_equal(a[3], b[3]) and int((a[0] < 0).sum()) == 0
    print("round", r, "n", n, "L", L, kind, kw, env, "pairs", len(a[0]), "diff", diff, "labels", bool(np.array_equal(a[3], b[3])),
          "failed", int((a[0] < 0).sum()), int((b[0] < 0).sum()), "OK" if ok else "FAIL", "%.1fs" % (time.time() - t0), flush=True)
    bad += 0 if ok else 1
print("stress done, failures:", bad)
sys.exit(1 if bad else 0)
