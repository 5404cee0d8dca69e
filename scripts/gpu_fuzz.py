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
MOFF = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # (argv[4]: first seed of the medium sets, 60 of them)
for seed in range(MOFF, MOFF + 60):
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

# round 2: raw-byte alphabets x kernel family x orientation mode, sparsified lists, batched arenas, compaction
import numpy as np
from conftest import canon_gfa
ob, synth = t.ob, t.synth
bad = []
for seed in range(int(sys.argv[3]) if len(sys.argv) > 3 else 80):
    rng = random.Random(9000 + seed)
    L = rng.choice([40, 150, 400, 900, 2500])
    n = rng.randint(2, 6)
    fam = synth.indel_family(n, L, rng.choice([0.02, 0.06, 0.15]), rng.choice([0.0, 0.01, 0.03]) or 0.001, 9100 + seed)
    recs = []
    for i, (nm, sq) in enumerate(fam):
        b = bytearray(sq)
        mode = rng.choice(["acgt", "n", "lower", "iupac", "mixed"])
        for _ in range(rng.randint(0, 4)):
            p = rng.randrange(len(b)); ln = rng.randint(1, max(1, len(b) // 8))
            if mode in ("n", "mixed"):
                b[p:p + ln] = b"N" * len(b[p:p + ln])
            if mode in ("lower", "mixed"):
                q = rng.randrange(len(b)); b[q:q + ln] = bytes(b[q:q + ln]).lower()
            if mode == "iupac":
                b[p:p + 3] = rng.choice([b"RYK", b"ryk", b"SWM", b"BDH"])[:len(b[p:p + 3])]
        sq = bytes(b)
        recs.append((nm, sq))
    if rng.random() < 0.4:                        # reverse-complement one member with the reference's complement table
        i = rng.randrange(len(recs)); o_ = ob.OracleSeqRush(records=[recs[i]]); import ctypes as C
        buf = C.create_string_buffer(len(recs[i][1])); ob.lib().sro_reverse_complement(recs[i][1], len(recs[i][1]), buf); recs[i] = (recs[i][0], buf.raw)
    kw = {}
    if rng.random() < 0.3: kw["scores"] = rng.choice(["0,5,8,2", "0,4,6,2,24,1", "0,6,9,2,30,1"])
    if rng.random() < 0.3: kw["min_match_len"] = rng.choice([1, 4, 12])
    env = {}
    if rng.random() < 0.3: env["SR_PREORIENT"] = rng.choice(["0", "1"])
    if rng.random() < 0.3: env["SR_ALIGN_THREADS"] = rng.choice(["64", "128", "512"])
    if rng.random() < 0.2: env["SR_BLK_LEVELS"] = "5"
    if rng.random() < 0.35: env["SR_NWG"] = rng.choice(["1", "2", "3"])      # several pairs per workgroup
    if rng.random() < 0.2: env["SR_CIGAR_ARENA_OPS"] = str(2 * L + 50)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        if "SR_CIGAR_ARENA_OPS" in env:           # batches: partition only (CIGARs are not resident)
            ss = t.SeqSet(recs); ctx = t.Context(0); ctx.load(ss, t.Params(**kw)); ctx.run(); ctx.sync()
            labels = ctx.download_labels(); dev = ctx.build_gfa(compact=True); ctx.close()
            o = ob.OracleSeqRush(records=recs); o.align_and_unite(t.oracle_params(**kw))
            assert np.array_equal(labels, o.canonical_labels())
        else:
            al, labels, cnt = t.check_parity(recs, **kw)
            ss = t.SeqSet(recs)
            o = ob.OracleSeqRush(records=recs); o.align_and_unite(t.oracle_params(**kw))
            dev = t.build_gfa(ss, labels, compact=True)
        assert canon_gfa(dev[0]) == canon_gfa(ob.compact_gfa(o.gfa(canonical=True)[0])[0])
        spec = rng.choice(["tree:2,1,0.2,8", "connectivity:0.8", "random:0.5", "tree:1,1,0.0"])
        ss = t.SeqSet(recs); ctx = t.Context(0); ctx.load(ss, t.Params(sparsification=spec, **kw)); pairs = ctx.pairs()
        assert pairs == o.sparsified_pairs(spec), spec
        ctx.run(); ctx.sync(); lab2 = ctx.download_labels(); ctx.close()
        o2 = ob.OracleSeqRush(records=recs); o2.align_and_unite_list(t.oracle_params(**kw), pairs)
        assert np.array_equal(lab2, o2.canonical_labels())
    except Exception as e:
        bad.append((seed, repr(e)[:300], env, kw)); print("FAIL r2", seed, repr(e)[:300], env, kw, flush=True)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    if seed % 10 == 0: print("r2", seed, L, n, flush=True)
print("round-2 fuzz done, failures:", bad)
