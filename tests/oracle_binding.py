"""ctypes binding to oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (seqrush_amd/) never does.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("uf.c", "wfa.c", "seqrush.c", "compact.c", "sr_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "-B"])
    return _LIB_PATH


class Penalties(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open1", C.c_int32),
                ("gap_ext1", C.c_int32), ("gap_open2", C.c_int32), ("gap_ext2", C.c_int32)]

    @staticmethod
    def of(m, x, o1, e1, o2=-1, e2=-1):
        return Penalties(m, x, o1, e1, o2, e2)


class Sparsification(C.Structure):
    _fields_ = [("kind", C.c_int), ("factor", C.c_double), ("k_nearest", C.c_uint64),
                ("k_farthest", C.c_uint64), ("rand_frac", C.c_double), ("kmer_size", C.c_uint64)]


class Sequence(C.Structure):
    _fields_ = [("id", C.c_char_p), ("data", C.POINTER(C.c_uint8)), ("len", C.c_uint64),
                ("offset", C.c_uint64)]


class SeqRushS(C.Structure):
    _fields_ = [("seqs", C.POINTER(Sequence)), ("n", C.c_uint64), ("total_length", C.c_uint64),
                ("uf", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("pen", Penalties), ("ori", Penalties), ("min_match_len", C.c_uint64),
                ("max_divergence", C.c_double), ("exclude_self", C.c_int),
                ("memory_mode", C.c_int), ("threads", C.c_int)]


class Alignment(C.Structure):
    _fields_ = [("query_idx", C.c_uint32), ("target_idx", C.c_uint32), ("is_reverse", C.c_int),
                ("score", C.c_int), ("cigar_bytes", C.POINTER(C.c_uint8)), ("cigar_len", C.c_int),
                ("query_start", C.c_uint64), ("query_end", C.c_uint64),
                ("target_start", C.c_uint64), ("target_end", C.c_uint64)]


MEM_HIGH, MEM_ULTRALOW = 0, 3
M, I1, I2, D1, D2 = range(5)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build_oracle())
    u64, i32, vp = C.c_uint64, C.c_int, C.c_void_p
    L.sro_make_pos.restype = u64; L.sro_make_pos.argtypes = [u64, i32]
    L.sro_is_rev.argtypes = [u64]
    L.sro_offset.restype = u64; L.sro_offset.argtypes = [u64]
    for f in ("sro_incr_pos", "sro_decr_pos", "sro_flip_orientation"):
        getattr(L, f).restype = u64; getattr(L, f).argtypes = [u64]
    L.sro_orientation_char.restype = C.c_char; L.sro_orientation_char.argtypes = [u64]
    L.sro_rc_base.restype = C.c_uint8; L.sro_rc_base.argtypes = [C.c_uint8]
    L.sro_uf_new.restype = vp; L.sro_uf_new.argtypes = [u64]
    L.sro_uf_free.argtypes = [vp]
    L.sro_uf_size.restype = u64; L.sro_uf_size.argtypes = [vp]
    L.sro_uf_find.restype = u64; L.sro_uf_find.argtypes = [vp, u64]
    L.sro_uf_unite.argtypes = [vp, u64, u64]
    L.sro_uf_same.argtypes = [vp, u64, u64]
    L.sro_uf_nodes.restype = C.POINTER(u64); L.sro_uf_nodes.argtypes = [vp]
    L.sro_buf_new.restype = vp; L.sro_buf_new.argtypes = [u64]
    L.sro_buf_find.restype = u64; L.sro_buf_find.argtypes = [vp, u64]
    L.sro_buf_unite.argtypes = [vp, u64, u64]; L.sro_buf_unite.restype = None
    L.sro_buf_same.argtypes = [vp, u64, u64]
    L.sro_buf_unite_matching_region.argtypes = [vp, u64, u64, u64, u64, u64, i32, u64]
    L.sro_buf_unite_matching_region.restype = None
    L.sro_buf_unite_matching_region_seq2_rc.argtypes = [vp, u64, u64, u64, u64, u64, i32, u64]
    L.sro_buf_unite_matching_region_seq2_rc.restype = None
    L.sro_parse_scores.argtypes = [C.c_char_p, C.POINTER(Penalties)]
    L.sro_parse_orientation_scores.argtypes = [C.c_char_p, C.POINTER(Penalties)]
    L.sro_max_score_for_divergence.argtypes = [C.POINTER(Penalties), u64, C.c_double]
    L.sro_max_score_for_divergence.restype = C.c_int32
    L.sro_parse_sparsification.argtypes = [C.c_char_p, C.POINTER(Sparsification)]
    L.sro_wfa_align.argtypes = [C.c_char_p, i32, C.c_char_p, i32, C.POINTER(Penalties), i32,
                                C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(i32), C.POINTER(i32)]
    L.sro_wfa_score.argtypes = [C.c_char_p, i32, C.c_char_p, i32, C.POINTER(Penalties), i32,
                                C.POINTER(i32)]
    L.sro_wfa_last_cells.restype = u64
    L.sro_gotoh_score.argtypes = [C.c_char_p, i32, C.c_char_p, i32, C.POINTER(Penalties)]
    L.sro_cigar_score.argtypes = [C.c_char_p, i32, C.c_char_p, i32, C.c_char_p, i32,
                                  C.POINTER(Penalties)]
    L.sro_cigar_bytes_to_string.restype = vp
    L.sro_cigar_bytes_to_string.argtypes = [C.c_char_p, i32]
    L.sro_load_fasta_mem.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(Sequence)),
                                     C.POINTER(u64)]
    L.sro_free_sequences.argtypes = [C.POINTER(Sequence), u64]; L.sro_free_sequences.restype = None
    L.sro_reverse_complement.argtypes = [C.c_char_p, u64, C.c_char_p]
    L.sro_reverse_complement.restype = None
    L.sro_seqrush_new.restype = C.POINTER(SeqRushS)
    L.sro_seqrush_new.argtypes = [C.POINTER(Sequence), u64, C.c_char_p, C.c_size_t]
    L.sro_seqrush_free.argtypes = [C.POINTER(SeqRushS)]; L.sro_seqrush_free.restype = None
    L.sro_count_components.restype = u64; L.sro_count_components.argtypes = [C.POINTER(SeqRushS)]
    L.sro_process_alignment.restype = C.c_int64
    L.sro_process_alignment.argtypes = [C.POINTER(SeqRushS), C.c_char_p, u64, u64, u64, i32,
                                        u64, u64, u64, u64]
    L.sro_default_params.argtypes = [C.POINTER(Params)]; L.sro_default_params.restype = None
    L.sro_align_pair.argtypes = [C.POINTER(SeqRushS), C.POINTER(Params), C.c_uint32, C.c_uint32,
                                 C.POINTER(Alignment)]
    L.sro_alignment_free.argtypes = [C.POINTER(Alignment)]; L.sro_alignment_free.restype = None
    L.sro_align_and_unite.restype = C.c_int64
    L.sro_align_and_unite.argtypes = [C.POINTER(SeqRushS), C.POINTER(Params), u64, u64,
                                      C.POINTER(u64)]
    L.sro_align_and_unite_list.restype = C.c_int64
    L.sro_align_and_unite_list.argtypes = [C.POINTER(SeqRushS), C.POINTER(Params), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32), u64, C.POINTER(u64)]
    L.sro_align_and_unite_list_collect.restype = C.c_int64
    L.sro_align_and_unite_list_collect.argtypes = [C.POINTER(SeqRushS), C.POINTER(Params), C.POINTER(C.c_uint32),
                                                   C.POINTER(C.c_uint32), u64, i32, C.POINTER(C.c_int32),
                                                   C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(u64)]
    L.sro_cigar_run_digest.restype = u64; L.sro_cigar_run_digest.argtypes = [C.c_char_p, u64]
    L.sro_sparsified_pairs.argtypes = [C.POINTER(SeqRushS), C.POINTER(Sparsification), u64, i32,
                                       C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)),
                                       C.POINTER(u64)]
    L.sro_compact_gfa.restype = vp
    L.sro_compact_gfa.argtypes = [C.c_char_p, C.POINTER(u64), C.POINTER(u64)]
    L.sro_rewrite_gfa.restype = vp
    L.sro_rewrite_gfa.argtypes = [C.c_char_p, C.POINTER(u64), C.POINTER(u64)]
    L.sro_gfa_path_sequence.restype = vp; L.sro_gfa_path_sequence.argtypes = [C.c_char_p, u64]
    L.sro_handle_new.restype = u64; L.sro_handle_new.argtypes = [u64, i32]
    L.sro_handle_node_id.restype = u64; L.sro_handle_node_id.argtypes = [u64]
    L.sro_handle_is_reverse.argtypes = [u64]
    L.sro_handle_orientation_char.restype = C.c_char; L.sro_handle_orientation_char.argtypes = [u64]
    L.sro_handle_flip.restype = u64; L.sro_handle_flip.argtypes = [u64]
    L.sro_build_gfa.restype = vp
    L.sro_build_gfa.argtypes = [C.POINTER(SeqRushS), i32, i32, C.POINTER(u64), C.POINTER(u64)]
    L.sro_canonical_labels.argtypes = [C.POINTER(SeqRushS), C.POINTER(u64)]
    L.sro_canonical_labels.restype = None
    L._libc = C.CDLL(None)
    L._libc.free.argtypes = [vp]
    _lib = L
    return L


# ---------------------------------------------------------------- helpers
def wfa_align(pattern: bytes, text: bytes, pen: Penalties, mode=MEM_ULTRALOW):
    """-> (raw cigar bytes 'MXID', score)"""
    L = lib()
    cig = C.POINTER(C.c_uint8)(); n = C.c_int(); sc = C.c_int()
    st = L.sro_wfa_align(pattern, len(pattern), text, len(text), C.byref(pen), mode,
                         C.byref(cig), C.byref(n), C.byref(sc))
    if st != 0:
        raise RuntimeError(f"sro_wfa_align failed: {st}")
    out = bytes(cig[: n.value])
    L._libc.free(C.cast(cig, C.c_void_p))
    return out, sc.value


def wfa_score(pattern: bytes, text: bytes, pen: Penalties, max_score=-1):
    L = lib()
    sc = C.c_int()
    if L.sro_wfa_score(pattern, len(pattern), text, len(text), C.byref(pen), max_score,
                       C.byref(sc)) != 0:
        raise RuntimeError("sro_wfa_score failed")
    return sc.value


def gotoh(pattern: bytes, text: bytes, pen: Penalties):
    return lib().sro_gotoh_score(pattern, len(pattern), text, len(text), C.byref(pen))


def cigar_score(cig: bytes, pattern: bytes, text: bytes, pen: Penalties):
    return lib().sro_cigar_score(cig, len(cig), pattern, len(pattern), text, len(text),
                                 C.byref(pen))


def cigar_bytes_to_string(cig: bytes) -> str:
    L = lib()
    p = L.sro_cigar_bytes_to_string(cig, len(cig))
    s = C.cast(p, C.c_char_p).value.decode()
    L._libc.free(p)
    return s


def cigar_run_digest(raw: bytes) -> int:
    return lib().sro_cigar_run_digest(raw, len(raw))


def cigar_run_digests(cigar_ops, cigar_off):
    """the oracle's run digest (oracle/seqrush.c cigar_run_digest) of every alignment of a device result, from its
    run-length op buffer (u32 (len << 4) | code, code 0 '=' 1 X 2 raw-D 3 raw-I) -- array operations only"""
    import numpy as np
    ops = np.asarray(cigar_ops, dtype=np.uint64)
    off = np.asarray(cigar_off, dtype=np.int64)
    n = len(off) - 1
    cnt = np.diff(off)
    idx = np.arange(len(ops), dtype=np.int64) - np.repeat(off[:-1], cnt)
    G = np.uint64(0x9e3779b97f4a7c15)
    with np.errstate(over="ignore"):
        x = ops ^ (idx.astype(np.uint64) * G)
        x = x + G
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        x = x ^ (x >> np.uint64(31))
        out = np.zeros(n, dtype=np.uint64)
        nz = cnt > 0
        if len(ops):
            sums = np.add.reduceat(x, off[:-1][nz])
            out[nz] = sums
    return out


def compact_gfa(gfa_text: str):
    """compact() + renumber of the reference (src/bidirected_ops.rs:75-490) on a --no-compact GFA -> (text, nodes, edges)"""
    L = lib()
    nn = C.c_uint64(); ne = C.c_uint64()
    p = L.sro_compact_gfa(gfa_text.encode(), C.byref(nn), C.byref(ne))
    s = C.cast(p, C.c_char_p).value.decode()
    L._libc.free(p)
    return s, nn.value, ne.value


def rewrite_gfa(gfa_text: str):
    """parse + write_gfa (src/bidirected_ops.rs:880-925), no compaction -> (text, nodes, edges)"""
    L = lib()
    nn = C.c_uint64(); ne = C.c_uint64()
    p = L.sro_rewrite_gfa(gfa_text.encode(), C.byref(nn), C.byref(ne))
    s = C.cast(p, C.c_char_p).value.decode()
    L._libc.free(p)
    return s, nn.value, ne.value


def gfa_path_sequence(gfa_text: str, index: int) -> str:
    L = lib()
    p = L.sro_gfa_path_sequence(gfa_text.encode(), index)
    s = C.cast(p, C.c_char_p).value.decode()
    L._libc.free(p)
    return s


def parse_scores(s: str):
    p = Penalties()
    r = lib().sro_parse_scores(s.encode(), C.byref(p))
    return r, p


def default_params() -> Params:
    p = Params()
    lib().sro_default_params(C.byref(p))
    return p


class OracleSeqRush:
    """Owns an sro_seqrush built from FASTA text or (name, bytes) records."""

    def __init__(self, fasta_text: bytes = None, records=None):
        L = lib()
        if records is not None:
            fasta_text = b"".join(b">" + n.encode() + b"\n" + s + b"\n" for n, s in records)
        seqs = C.POINTER(Sequence)(); cnt = C.c_uint64()
        L.sro_load_fasta_mem(fasta_text, len(fasta_text), C.byref(seqs), C.byref(cnt))
        err = C.create_string_buffer(512)
        self.ptr = L.sro_seqrush_new(seqs, cnt.value, err, 512)
        if not self.ptr:
            L.sro_free_sequences(seqs, cnt.value)
            raise ValueError(err.value.decode())
        self.L = L

    @property
    def n(self):
        return self.ptr.contents.n

    @property
    def total_length(self):
        return self.ptr.contents.total_length

    @property
    def uf(self):
        return self.ptr.contents.uf

    def seq(self, i):
        s = self.ptr.contents.seqs[i]
        return s.id.decode(), bytes(s.data[: s.len]), s.offset

    def process_alignment(self, cigar: str, q, t, k=0, rc=False, qs=0, qe=None, ts=0, te=None):
        sq = self.ptr.contents.seqs[q]; st = self.ptr.contents.seqs[t]
        return self.L.sro_process_alignment(self.ptr, cigar.encode(), q, t, k, int(rc), qs,
                                            sq.len if qe is None else qe, ts,
                                            st.len if te is None else te)

    def align_pair(self, params: Params, q, t):
        a = Alignment()
        if self.L.sro_align_pair(self.ptr, C.byref(params), q, t, C.byref(a)) != 0:
            raise RuntimeError("sro_align_pair failed")
        out = dict(q=a.query_idx, t=a.target_idx, is_reverse=bool(a.is_reverse), score=a.score,
                   cigar=bytes(a.cigar_bytes[: a.cigar_len]))
        self.L.sro_alignment_free(C.byref(a))
        return out

    def align_and_unite(self, params: Params, begin=0, end=None):
        n = self.n
        cells = C.c_uint64()
        r = self.L.sro_align_and_unite(self.ptr, C.byref(params), begin,
                                       n * n if end is None else end, C.byref(cells))
        if r < 0:
            raise RuntimeError("sro_align_and_unite failed")
        return r, cells.value

    def align_and_unite_list(self, params: Params, pairs):
        """the same over an explicit ordered (query, target) list"""
        import numpy as np
        q = np.ascontiguousarray([a for a, _ in pairs], dtype=np.uint32)
        t = np.ascontiguousarray([b for _, b in pairs], dtype=np.uint32)
        cells = C.c_uint64()
        r = self.L.sro_align_and_unite_list(self.ptr, C.byref(params), q.ctypes.data_as(C.POINTER(C.c_uint32)),
                                            t.ctypes.data_as(C.POINTER(C.c_uint32)), len(pairs), C.byref(cells))
        if r < 0:
            raise RuntimeError("sro_align_and_unite_list failed")
        return r, cells.value

    def align_list_collect(self, params: Params, pairs, unite=True):
        """align (and unite) an ordered pair list on params.threads threads -> per-pair (score, is_reverse, number of
        CIGAR runs, run digest) arrays: what a full-size comparison with the device needs (see cigar_run_digests)"""
        import numpy as np
        n = len(pairs)
        q = np.ascontiguousarray([a for a, _ in pairs], dtype=np.uint32)
        t = np.ascontiguousarray([b for _, b in pairs], dtype=np.uint32)
        sc = np.zeros(max(n, 1), dtype=np.int32); rv = np.zeros(max(n, 1), dtype=np.uint8)
        nr = np.zeros(max(n, 1), dtype=np.uint32); dg = np.zeros(max(n, 1), dtype=np.uint64)
        r = self.L.sro_align_and_unite_list_collect(
            self.ptr, C.byref(params), q.ctypes.data_as(C.POINTER(C.c_uint32)), t.ctypes.data_as(C.POINTER(C.c_uint32)),
            n, int(unite), sc.ctypes.data_as(C.POINTER(C.c_int32)), rv.ctypes.data_as(C.POINTER(C.c_uint8)),
            nr.ctypes.data_as(C.POINTER(C.c_uint32)), dg.ctypes.data_as(C.POINTER(C.c_uint64)))
        if r != n:
            raise RuntimeError("sro_align_and_unite_list_collect failed")
        return sc[:n], rv[:n], nr[:n], dg[:n]

    def sparsified_pairs(self, spec: str, seed=42, exclude_self=False):
        """ordered pair list of `-x spec` (own definition, unpinned: allwave's rules are not in the reference tree)"""
        sp = Sparsification()
        if self.L.sro_parse_sparsification(spec.encode(), C.byref(sp)) != 0:
            raise ValueError(spec)
        q = C.POINTER(C.c_uint32)(); t = C.POINTER(C.c_uint32)(); cnt = C.c_uint64()
        if self.L.sro_sparsified_pairs(self.ptr, C.byref(sp), seed, int(exclude_self), C.byref(q), C.byref(t),
                                       C.byref(cnt)) != 0:
            raise RuntimeError("sro_sparsified_pairs failed")
        out = [(int(q[i]), int(t[i])) for i in range(cnt.value)]
        self.L._libc.free(C.cast(q, C.c_void_p)); self.L._libc.free(C.cast(t, C.c_void_p))
        return out

    def find(self, pos):
        return self.L.sro_buf_find(self.uf, pos)

    def same(self, a, b):
        return bool(self.L.sro_buf_same(self.uf, a, b))

    def count_components(self):
        return self.L.sro_count_components(self.ptr)

    def nodes(self):
        import numpy as np
        n = self.L.sro_uf_size(self.uf)
        p = self.L.sro_uf_nodes(self.uf)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def canonical_labels(self):
        import numpy as np
        n = self.L.sro_uf_size(self.uf)
        out = np.zeros(n, dtype=np.uint64)
        self.L.sro_canonical_labels(self.ptr, out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out

    def gfa(self, canonical=True, faithful_scan=False):
        nn = C.c_uint64(); ne = C.c_uint64()
        p = self.L.sro_build_gfa(self.ptr, int(canonical), int(faithful_scan), C.byref(nn),
                                 C.byref(ne))
        s = C.cast(p, C.c_char_p).value.decode()
        self.L._libc.free(p)
        return s, nn.value, ne.value

    def close(self):
        if self.ptr:
            self.L.sro_seqrush_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
