import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge; ge.build()
import test_gpu_parity as t
bad = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 260):
    try:
        t.test_randomised_small_sets(None, seed)
    except Exception as e:
        bad.append((seed, repr(e)[:200])); print("FAIL", seed, repr(e)[:200], flush=True)
    if seed % 40 == 0: print("seed", seed, flush=True)
print("done, failures:", bad)

# medium sets: lengths 800..7000, divergence 1..20 %, indels, RC: breakpoint recursion 2-5 levels deep
import random
bad = []
for seed in range(60):
    rng = random.Random(5000 + seed)
    L = rng.randint(800, 7000)
    n = rng.randint(2, 3)
    sub, ind = rng.choice([0.01, 0.03, 0.06, 0.1, 0.2]), rng.choice([0.0, 0.005, 0.02])
    recs = t.synth.indel_family(n, L, sub, ind, 7000 + seed) if ind > 0 else t.synth.snp_family(n, L, sub, 7000 + seed, rc_every=rng.choice([0, 2, 3]) or None)
    if rng.random() < 0.3:
        recs = [(nm, sq[rng.randint(0, len(sq) // 5):]) for nm, sq in recs]
    try:
        sc = [None, "0,5,8,2", "0,6,9,2,30,1", "0,7,12,2,20,1"][seed % 4]       # blocked+lazy, one-piece, blocked eager, blocked+lazy
        t.check_parity(recs, **({"scores": sc} if sc else {}))
    except Exception as e:
        bad.append((seed, repr(e)[:200])); print("FAIL medium", seed, repr(e)[:200], flush=True)
    if seed % 10 == 0: print("medium", seed, L, sub, ind, flush=True)
print("medium done, failures:", bad)
