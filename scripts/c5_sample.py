"""A larger oracle sample of the full C5 configuration than the test suite carries (tests/test_gpu_parity.py compares 12
pairs): N pairs of the 256 x 50 kb set drawn at random (seeded), a quarter of them among the pairs with an inverted or
reverse-complemented member, aligned by the product through an explicit pair list and by the oracle's biWFA on the host
threads -- strand, score and the raw CIGAR bytes must be equal.  usage: c5_sample.py [N=240] [seed=1]"""
import concurrent.futures as cf
import os
import sys
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge; ge.build()
import oracle_binding as ob
from conftest import usable_cpus
from seqrush_amd import synth
from seqrush_amd.seqrush import SeqSet, Params, Context

N = int(sys.argv[1]) if len(sys.argv) > 1 else 240
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
recs = synth.config_c5()
n = len(recs)
rng = np.random.default_rng(seed)
pick = set()
while len(pick) < N:
    q, t = int(rng.integers(n)), int(rng.integers(n))
    if q != t:
        pick.add((q, t))
pick = sorted(pick)
ss = SeqSet(recs)
t0 = time.perf_counter()
c = Context(0); c.load_pairs(ss, Params(), pick); c.run(); c.sync(); al = c.alignments(); rep = c.workspace_report(); c.close()
print(f"product: {N} pairs in {time.perf_counter() - t0:.1f} s, offset_bytes {rep['offset_bytes']}, ring_cell_bytes {rep['ring_cell_bytes']}, "
      f"threads {rep['threads_per_workgroup']}, build {rep['kernel_build']} {rep['source_digest']}", flush=True)
o = ob.OracleSeqRush(records=recs)
op = ob.default_params(); op.threads = 1
bad = []
t0 = time.perf_counter()
done = [0]


def one(i):
    q, t = pick[i]
    w = o.align_pair(op, q, t)
    ok = (int(al.score[i]) == w["score"] and bool(al.is_reverse[i]) == w["is_reverse"] and al.raw_cigar_bytes(i) == w["cigar"])
    done[0] += 1
    if done[0] % 20 == 0:
        print(f"oracle {done[0]} / {N} ({time.perf_counter() - t0:.0f} s)", flush=True)
    return ok, w["score"], w["is_reverse"]


with cf.ThreadPoolExecutor(max_workers=usable_cpus()) as ex:
    res = list(ex.map(one, range(N)))
for i, (ok, sc, rv) in enumerate(res):
    if not ok:
        bad.append(pick[i]); print("DIFF", pick[i], flush=True)
print(f"oracle: {N} pairs in {time.perf_counter() - t0:.1f} s; reversed {sum(1 for r in res if r[2])}, scores {min(r[1] for r in res)}..{max(r[1] for r in res)}")
print("c5 sample done, differing pairs:", bad)
o.close()
sys.exit(1 if bad else 0)
