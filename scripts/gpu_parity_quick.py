import sys, random, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
import seqrush_amd as sa
from seqrush_amd.seqrush import SeqSet, Params, Context
import oracle_binding as ob

def synth(n, L, sub, seed, indel=0.0, rc_frac=0.0):
    rng = random.Random(seed)
    base = bytes(rng.choice(b"ACGT") for _ in range(L))
    out = []
    for i in range(n):
        r = random.Random(seed + 1 + i)
        s = bytearray()
        j = 0
        while j < L:
            x = r.random()
            if x < sub: s.append(r.choice([c for c in b"ACGT" if c != base[j]])); j += 1
            elif x < sub + indel/2: j += r.randint(1, 8)
            elif x < sub + indel:
                for _ in range(r.randint(1, 8)): s.append(r.choice(b"ACGT"))
            else: s.append(base[j]); j += 1
        s = bytes(s)
        if r.random() < rc_frac:
            s = bytes({65:84,84:65,67:71,71:67}[c] for c in reversed(s))
        out.append((f"seq{i}", s))
    return out

def run_case(name, recs, **kw):
    ss = SeqSet(recs)
    p = Params(**kw)
    ctx = Context(0)
    t0 = time.time()
    ctx.load(ss, p)
    t1 = time.time()
    ctx.align(); ctx.unite(); ctx.sync()
    t2 = time.time()
    al = ctx.alignments()
    labels = ctx.download_labels()
    ctx.sync()
    print(f"[{name}] load {t1-t0:.3f}s run {t2-t1:.3f}s align_ms {ctx.kernel_ms(0):.2f} unite_ms {ctx.kernel_ms(1):.2f} counters {ctx.counters()}")
    # oracle
    o = ob.OracleSeqRush(records=recs)
    op = ob.default_params(); op.threads = 8
    if 'scores' in kw:
        r, pen = ob.parse_scores(kw['scores']); op.pen = pen
    if 'min_match_len' in kw: op.min_match_len = kw['min_match_len']
    n = len(recs)
    bad = 0
    for i in range(al.n):
        q, t = int(al.query_idx[i]), int(al.target_idx[i])
        oa = o.align_pair(op, q, t)
        g = al.raw_cigar_bytes(i)
        if g != oa['cigar'] or bool(al.is_reverse[i]) != oa['is_reverse'] or int(al.score[i]) != oa['score']:
            bad += 1
            if bad < 5:
                print("  MISMATCH pair", q, t, "gpu score", al.score[i], "oracle", oa['score'], "rev", al.is_reverse[i], oa['is_reverse'])
                print("   gpu", ob.cigar_bytes_to_string(g)[:200])
                print("   orc", ob.cigar_bytes_to_string(oa['cigar'])[:200])
    o.align_and_unite(op)
    ol = o.canonical_labels()
    same = np.array_equal(ol, labels)
    gfa_o, nn, ne = o.gfa(canonical=True)
    gfa_g, gn, ge = sa.build_gfa(ss, labels)
    def canon(g):
        lines = g.strip().split("\n")
        return [l for l in lines if l[0] != 'L'], sorted(l for l in lines if l[0] == 'L')
    print(f"[{name}] pairs {al.n} cigar mismatches {bad}; labels equal {same}; gfa equal {canon(gfa_o)==canon(gfa_g)} nodes {nn}/{gn} edges {ne}/{ge}")
    ctx.close()
    return bad == 0 and same

ok = True
ok &= run_case("tiny", [("a", b"ATCGATCG"), ("b", b"ATCGATCGATCG"), ("c", b"ATTGATCGATCG")])
ok &= run_case("8x300", synth(8, 300, 0.05, 11))
ok &= run_case("8x1k", synth(8, 1000, 0.05, 1001))
ok &= run_case("6x1k-indel", synth(6, 1000, 0.03, 77, indel=0.01))
ok &= run_case("6x800-rc", synth(6, 800, 0.04, 99, indel=0.005, rc_frac=0.5))
ok &= run_case("4x2k-1p", synth(4, 2000, 0.05, 5), scores="0,5,8,2")
ok &= run_case("4x1k-k8", synth(4, 1000, 0.05, 6), min_match_len=8)
print("ALL OK" if ok else "FAILURES")
