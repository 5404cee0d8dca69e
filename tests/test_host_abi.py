"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol the header
declares, host-side logic (parsers, pair list, FASTA mirror, graph induction / GFA writer)
against the oracle, and loud failure without a device.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_binding as ob
import seqrush_amd as sa
from seqrush_amd import _lib, synth
from seqrush_amd.seqrush import SeqSet, Params, pair_list, build_gfa, load_sequences
from conftest import canon_gfa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "seqrush_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", hdr))
    L = _lib.load()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)
    assert L.sr_abi_version() == 2


def test_no_device_fails_loudly():
    L = _lib.load()
    if L.sr_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sa.SeqRushError) as e:
        sa.Context(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(sa.SeqRushError):
        sa.create_aligner("allwave").align_sequences([sa.AlignmentSequence("a", b"ACGT")])


def test_product_does_not_touch_the_oracle():
    """the product package must not import, link or mention oracle/"""
    pkg = os.path.join(ROOT, "seqrush_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                assert "oracle_binding" not in txt and "liboracle" not in txt and "sro_" not in txt, f


def test_parsers_match_oracle():
    cases = ["0,5,8,2,24,1", "0,5,8,2", "0,5,8,2,24", "0,5,8", "1,2,3,4,5,6,7", "0,a,8,2", "0, 5,8,2", "0,+5,8,2"]
    for s in cases:
        r, po = ob.parse_scores(s)
        p = Params()
        rc = _lib.load().sr_parse_scores(s.encode(), C.byref(p.c))
        assert (rc == 0) == (r == 0), s
        if r == 0:
            assert (p.c.match_score, p.c.mismatch_penalty, p.c.gap_open1, p.c.gap_ext1) == (po.match, po.mismatch, po.gap_open1, po.gap_ext1)
            assert (p.c.gap_open2 >= 0) == (po.gap_open2 >= 0)
    for s in ["none", "1.0", "auto", "random:0.5", "random:0", "connectivity:0.3", "tree:3,3,0.1", "tree:", "0.5", "x",
              # Rust's parsers (src/seqrush.rs:356-431): hex floats are errors, a 20-digit usize is fine, 2^64 overflows
              "0x.8", "random:0x1p-1", "tree:3,3,0X.1", "tree:18446744073709551615", "tree:18446744073709551616",
              "tree:+3,+2", "tree:3,3,0.1,99999999999999999999"]:
        sp = ob.Sparsification()
        r = ob.lib().sro_parse_sparsification(s.encode(), C.byref(sp))
        p = Params()
        rc = _lib.load().sr_parse_sparsification(s.encode(), C.byref(p.c))
        assert (rc == 0) == (r == 0), s
        if r == 0:
            assert p.c.sparsify_kind == sp.kind
        assert (r == 0) == (s in ("none", "1.0", "auto", "random:0.5", "connectivity:0.3", "tree:3,3,0.1", "0.5",
                                  "tree:18446744073709551615", "tree:+3,+2")), s
    assert sa.AlignmentScores.parse("0,5,8,2,24,1").gap2_extend == 1
    assert sa.AlignmentScores.parse_orientation("0,1,1,1").gap1_open == 1
    with pytest.raises(sa.SeqRushError):
        sa.AlignmentScores.parse_orientation("0,1,1")


def test_pair_list_and_sharding():
    p = Params()
    full = pair_list(5, p)
    assert full == [(q, t) for q in range(5) for t in range(5)]      # "all-vs-all including self", seqrush.rs:718-734
    p.c.exclude_self = 1
    assert pair_list(4, p) == [(q, t) for q in range(4) for t in range(4) if q != t]
    p = Params()
    shards = []
    for r in range(3):
        p.c.shard_rank, p.c.shard_count = r, 3
        shards.append(pair_list(7, p))
    assert sorted(sum(shards, [])) == [(q, t) for q in range(7) for t in range(7)]
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    p = Params(sparsification="random:0.3")
    sub = pair_list(20, p)
    assert all((q, q) in sub for q in range(20))
    assert 0.15 < (len(sub) - 20) / 380 < 0.45
    assert sub == pair_list(20, p)        # deterministic


def test_sparsified_pair_lists_host_kinds_match_oracle():
    """connectivity / auto / random need no sketches: the product's host enumeration equals the oracle's restatement
    of the same (own, unpinned) definition; tree needs the sequences and is refused by the host-only helper"""
    for n in (3, 9, 10, 37):
        recs = [(f"s{i}", b"ACGT" * 3) for i in range(n)]
        o = ob.OracleSeqRush(records=recs)
        for spec in ("none", "auto", "connectivity:0.5", "connectivity:0.999", "connectivity:1.0", "random:0.25", "0.7"):
            for ex in (0, 1):
                p = Params(sparsification=spec)
                p.c.exclude_self = ex
                assert pair_list(n, p) == o.sparsified_pairs(spec, exclude_self=bool(ex)), (n, spec, ex)
    p = Params(sparsification="connectivity:0.5")
    got = pair_list(200, p)
    assert all((t, q) in set(got) for q, t in got)                  # both directions of a kept unordered pair
    assert 200 < len(got) < 200 * 200 // 4                          # (ln 200 + 0.37) / 200 = 2.8 % of the pairs
    with pytest.raises(sa.SeqRushError) as e:
        pair_list(5, Params(sparsification="tree:3,3,0.1"))
    assert e.value.code == -6
    for spec, want in [("tree:3", (3, 0, 0.0, 16)), ("tree:3,2", (3, 2, 0.0, 16)), ("tree:1,0,0.25", (1, 0, 0.25, 16)),
                       ("tree:4,4,1.0,21", (4, 4, 1.0, 21))]:
        p = Params(sparsification=spec)
        sp = ob.Sparsification(); assert ob.lib().sro_parse_sparsification(spec.encode(), C.byref(sp)) == 0
        assert (p.c.tree_k_nearest, p.c.tree_k_farthest, p.c.tree_rand_frac, p.c.tree_kmer) == want
        assert (sp.k_nearest, sp.k_farthest, sp.rand_frac, sp.kmer_size) == want
    for bad in ("tree:a", "tree:1,b", "tree:1,1,1.5", "tree:1,1,0.1,0", "tree:1,2,3,4,5"):
        with pytest.raises(sa.SeqRushError):
            Params(sparsification=bad)


def test_shards_partition_the_list_for_every_world_size():
    for world in (1, 2, 3, 4, 8):
        for spec in ("none", "connectivity:0.6"):
            shards = []
            for r in range(world):
                p = Params(sparsification=spec); p.c.shard_rank, p.c.shard_count = r, world
                shards.append(pair_list(23, p))
            full = pair_list(23, Params(sparsification=spec))
            assert sorted(sum(shards, [])) == sorted(full)
            assert all(s == sorted(s) for s in shards)              # a rank keeps enumeration order
            off = [sum(1 for q, t in s if q != t) for s in shards]
            assert max(off) - min(off) <= 1                         # equal costs: round-robin; self pairs spread apart
            slf = [sum(1 for q, t in s if q == t) for s in shards]
            cost = [4 * a + 2 * b for a, b in zip(off, slf)]            # |q||t| for pairs, |q| for self pairs (len 2 here)
            assert max(cost) - min(cost) <= 4 and max(slf) - min(slf) <= 3


def test_fasta_loader_mirror(tmp_path):
    txt = b">s1 desc here\nACGT\n  GG  \n>s2\tx\nTT\r\n>\nAAAA\n>s3\nC\n"
    f = tmp_path / "x.fa"
    f.write_bytes(txt)
    seqs = load_sequences(str(f))
    o = ob.OracleSeqRush(fasta_text=txt)
    assert [(s.id, s.data, s.offset) for s in seqs] == [o.seq(i) for i in range(o.n)]


def _oracle_labels(recs, k=0, threads=4):
    o = ob.OracleSeqRush(records=recs)
    p = ob.default_params(); p.min_match_len = k; p.threads = threads
    o.align_and_unite(p)
    return o


@pytest.mark.parametrize("name,recs", [
    ("snp", synth.snp_family(5, 250, 0.05, 3)),
    ("indel", synth.indel_family(4, 300, 0.03, 0.02, 4)),
    ("rc", synth.snp_family(6, 200, 0.04, 5, rc_every=2)),
    ("tiny", [("single", b"A"), ("double", b"AT")]),
])
def test_product_gfa_builder_matches_oracle(name, recs):
    """sr_build_gfa (host C++, O(N)) vs the oracle's restatement of
    bidirected_builder.rs:17-289 + write_gfa, on the oracle's canonical labels"""
    o = _oracle_labels(recs)
    labels = o.canonical_labels()
    ss = SeqSet(recs)
    g_prod, nn, ne = build_gfa(ss, labels)
    g_orc, on, oe = o.gfa(canonical=True)
    assert (nn, ne) == (on, oe)
    assert g_prod == g_orc                       # identical incl. L-line order (first insertion)
    assert canon_gfa(g_prod) == canon_gfa(o.gfa(canonical=True, faithful_scan=True)[0])


def _base(L, seed):
    return synth.to_bytes(synth.base_sequence(L, seed))


COMPACT_CASES = {
    "identical": lambda: [(f"s{i}", _base(150, 777)) for i in range(5)],
    "snp": lambda: synth.snp_family(5, 400, 0.05, 13),
    "indel": lambda: synth.indel_family(4, 500, 0.03, 0.02, 14),
    "rc": lambda: synth.snp_family(6, 300, 0.04, 15, rc_every=2),
    "rc-indel": lambda: [(n, s if i % 2 else synth.reverse_complement(s)) for i, (n, s) in enumerate(synth.indel_family(4, 400, 0.03, 0.03, 16))],
    # a path that starts / ends in the middle of a chain: merge_component_v2's validation refuses the whole chain
    "suffix-prefix": lambda: [("full", _base(300, 17)), ("suffix", _base(300, 17)[120:]), ("prefix", _base(300, 17)[:77]),
                              ("mid", _base(300, 17)[40:260])],
    "suffix-rc": lambda: [("full", _base(260, 18)), ("rcsuffix", synth.reverse_complement(_base(260, 18)[100:])),
                          ("snp", synth.to_bytes(synth.substitute(synth.base_sequence(260, 18), 0.05, 19)))],
    "tandem-dup": lambda: [("s1", _base(100, 999)), ("s2", (lambda b: b[:40] + b[40:50] * 2 + b[50:])(_base(100, 999))),
                           ("s3", (lambda b: b[:60] + b[60:65] * 4 + b[65:])(_base(100, 999)))],
    "homopolymer": lambda: [("seq1", b"AAAAAAAA"), ("seq2", b"AAAAAAAA")],          # tests/test_edge_traversal.rs:105-176
    "tiny": lambda: [("single", b"A"), ("double", b"AT")],
    "n-and-lowercase": lambda: [(n, s[:60] + b"NNNN" + s[64:150] + s[150:200].lower() + s[200:]) if i != 1 else (n, s)
                                for i, (n, s) in enumerate(synth.snp_family(3, 260, 0.04, 20))],
}


@pytest.mark.parametrize("name", sorted(COMPACT_CASES))
@pytest.mark.parametrize("k", [0, 6])
def test_compaction_matches_oracle(name, k):
    """compact() + renumber_nodes_sequentially() (src/bidirected_ops.rs:75-490) -- the product's
    table-driven rounds (sr_compact.cpp) against the oracle's literal restatement, on the canonical GFA; every path of
    the compacted graph still spells its input (src/bidirected_gfa_writer.rs:143-148)"""
    recs = COMPACT_CASES[name]()
    o = _oracle_labels(recs, k=k)
    labels = o.canonical_labels()
    ss = SeqSet(recs)
    g_prod, nn, ne = build_gfa(ss, labels, compact=True)
    g_orc, on, oe = ob.compact_gfa(o.gfa(canonical=True)[0])
    assert (nn, ne) == (on, oe)
    assert canon_gfa(g_prod) == canon_gfa(g_orc)
    plain, pn, _ = build_gfa(ss, labels)
    assert nn <= pn
    seg = {}
    rc = bytes.maketrans(b"ACGTacgtNn", b"TGCATGCANN")
    for l in g_prod.split("\n"):
        if l.startswith("S\t"):
            f = l.split("\t"); seg[f[1]] = f[2].encode()
    for l in g_prod.split("\n"):
        if l.startswith("P\t"):
            f = l.split("\t")
            spelled = b"".join(seg[s[:-1]] if s[-1] == "+" else seg[s[:-1]].translate(rc)[::-1] for s in f[2].split(","))
            want = dict(recs)[f[1]]
            if name != "n-and-lowercase":                       # (case / N differences are the reference's own quirk there)
                assert spelled == want
            else:
                assert spelled.upper() == want.upper()
    if name == "identical":
        assert (nn, ne) == (1, 0)
    if name == "homopolymer":
        loops = [l for l in g_prod.split("\n") if l.startswith("L\t") and l.split("\t")[1] == l.split("\t")[3]]
        assert len(loops) <= 2                                  # the reference's own assertion


def test_uf_find_helper():
    o = _oracle_labels(synth.snp_family(3, 120, 0.05, 9))
    nodes = o.nodes()
    from seqrush_amd.seqrush import uf_find
    for x in (0, 1, 77, 200, len(nodes) - 1):
        assert uf_find(nodes, x) == o.find(x)


def test_host_union_find_is_uf_rush_bit_for_bit():
    """sr_uf_init_host / sr_uf_unite_host against the oracle's uf_rush restatement (uf_rush-0.2.1/src/lib.rs:112-208): after
    SeqRush::new and the same sequence of unites the NODE ARRAYS are equal word for word (parent, rank bits, halving)"""
    from seqrush_amd.seqrush import HostUnionFind
    total = 300
    h = HostUnionFind(total)
    L = ob.lib()
    u = L.sro_buf_new(total)
    for i in range(total):                                   # SeqRush::new (src/seqrush.rs:324-328)
        L.sro_buf_unite(u, 2 * i, 2 * i + 1)
    n = L.sro_uf_size(u)
    assert n == h.n
    grab = lambda: np.ctypeslib.as_array(L.sro_uf_nodes(u), shape=(n,)).copy()
    assert np.array_equal(grab(), h.nodes)
    rng = np.random.default_rng(77)
    for _ in range(2000):
        x, y = (int(v) for v in rng.integers(0, 2 * total, 2))
        before = bool(L.sro_buf_same(u, x, y))
        L.sro_buf_unite(u, x, y)
        assert h.unite(x, y) == (not before)
    assert np.array_equal(grab(), h.nodes)
    with pytest.raises(sa.SeqRushError):
        h.unite(0, h.n)                                      # the reference panics on an index out of range (lib.rs:113)
    lab = h.canonical_labels()
    for x in range(0, h.n, 7):
        assert lab[x] == min(y for y in range(h.n) if L.sro_buf_same(u, x, y)) if x < 40 else lab[lab[x]] == lab[x]


def test_host_label_merge_reproduces_the_single_forest():
    """sr_uf_merge_labels_host (SURVEY 8e on the host): forests of the product's pair shards, exchanged as canonical label
    arrays and replayed into a fresh SeqRush::new forest, give the partition of the unsharded run -- for 2, 3 and 5 shards"""
    from seqrush_amd.seqrush import HostUnionFind
    recs = synth.snp_family(6, 160, 0.05, 31, rc_every=3)
    ref = _oracle_labels(recs).canonical_labels()
    total = sum(len(s) for _, s in recs)
    for world in (2, 3, 5):
        gathered = []
        for rank in range(world):
            prm = Params(); prm.c.shard_rank, prm.c.shard_count = rank, world
            o = ob.OracleSeqRush(records=recs)
            op = ob.default_params()
            for (qi, ti) in pair_list(len(recs), prm):
                a = o.align_pair(op, qi, ti)
                assert o.process_alignment(ob.cigar_bytes_to_string(a["cigar"]), qi, ti, 0, a["is_reverse"]) >= 0
            gathered.append(o.canonical_labels())
            o.close()
        h = HostUnionFind(total)
        h.merge_labels(gathered)
        assert np.array_equal(h.canonical_labels(), ref)


@pytest.mark.parametrize("name,recs", [
    ("rc", synth.snp_family(6, 200, 0.04, 5, rc_every=2)),
    ("rc-indel", [(n, s if i % 2 else synth.reverse_complement(s)) for i, (n, s) in enumerate(synth.indel_family(4, 300, 0.03, 0.03, 16))]),
    ("snp", synth.snp_family(4, 250, 0.05, 3)),
])
def test_gfa_from_raw_nodes_uses_the_root_rule(name, recs):
    """sr_build_gfa_from_nodes: node base = base at the offset of the component's uf_rush ROOT, the reference's own rule
    (src/bidirected_builder.rs:46-48, 176-182), so that a host which replays the unites in a fixed order gets the
    reference's node orientation too.  Against the oracle's non-canonical induction on its own (sequentially built)
    forest: byte-identical GFA, also after compaction; and with canonical labels as "nodes" it degenerates to sr_build_gfa."""
    from seqrush_amd.seqrush import build_gfa_from_nodes
    o = _oracle_labels(recs)
    ss = SeqSet(recs)
    g_prod, nn, ne = build_gfa_from_nodes(ss, o.nodes())
    g_orc, on, oe = o.gfa(canonical=False)
    assert (nn, ne) == (on, oe) and g_prod == g_orc
    gc_prod, cn, ce = build_gfa_from_nodes(ss, o.nodes(), compact=True)
    gc_orc, con, coe = ob.compact_gfa(g_orc)
    assert (cn, ce) == (con, coe) and canon_gfa(gc_prod) == canon_gfa(gc_orc)
    lab = o.canonical_labels()
    assert build_gfa_from_nodes(ss, lab) == build_gfa(ss, lab)       # a label array is a (flat) forest


def test_aligner_backend_names():
    """src/aligner.rs:42-52, 64-96"""
    assert sa.AlignerBackend.from_str("AllWave") == sa.AlignerBackend.AllWave
    assert sa.AlignerBackend.from_str("SWEEPGA") == sa.AlignerBackend.SweepGA
    with pytest.raises(ValueError, match="Unknown aligner"):
        sa.AlignerBackend.from_str("bwa")
    assert isinstance(sa.create_aligner("allwave", 4, False, None), sa.Aligner)
    with pytest.raises(RuntimeError, match="SweepGA aligner not available"):
        sa.create_aligner(sa.AlignerBackend.SweepGA)


def test_synth_is_deterministic():
    a = synth.config_c1()
    assert len(a) == 8 and all(len(s) == 1000 for _, s in a)
    assert a == synth.config_c1()
    import hashlib
    h = hashlib.sha256(b"".join(s for _, s in synth.snp_family(4, 500, 0.05, 2001))).hexdigest()
    assert h == hashlib.sha256(b"".join(s for _, s in synth.snp_family(4, 500, 0.05, 2001))).hexdigest()
