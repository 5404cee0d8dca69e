"""Differential run of one input under two environments (kernel knobs): scores, strands, CIGAR lengths and the
partition must agree.  No oracle: for inputs too large for the CPU restatement, or to localise a hang.
usage: python scripts/diff_probe.py NSEQ 'A=1 B=2' 'C=3' [config]"""
import sys, os, hashlib, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime of the torch wheel first)
import numpy as np
from seqrush_amd import synth
from seqrush_amd.seqrush import SeqSet, Context, Params

def run(recs, envs):
    env = dict(kv.split("=", 1) for kv in envs.split()) if envs.strip() else {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ss = SeqSet(recs); ctx = Context(0); ctx.load(ss, Params())
        t0 = time.time(); ctx.run(); ctx.sync(); dt = time.time() - t0
        sc, rv, co = ctx.pair_results(); lab = ctx.download_labels(); ctx.close()
        print("env", env, "pairs", len(sc), "%.1f ms" % (dt * 1e3), "score sum", int(sc.astype(np.int64).sum()), flush=True)
        return sc.copy(), rv.copy(), co.copy(), lab
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

if __name__ == "__main__":
    n = int(sys.argv[1]); ea, eb = sys.argv[2], sys.argv[3]
    cfg = sys.argv[4] if len(sys.argv) > 4 else "c2"
    recs = synth.config_c2(n) if cfg == "c2" else synth.indel_family(n, 3000, 0.06, 0.02, 4242)
    a = run(recs, ea); b = run(recs, eb)
    bad = np.nonzero((a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]))[0]
    print("pairs that differ:", len(bad), bad[:20].tolist())
    print("labels equal:", bool(np.array_equal(a[3], b[3])))
    sys.exit(1 if len(bad) or not np.array_equal(a[3], b[3]) else 0)
