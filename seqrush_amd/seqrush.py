"""Host-side mirror of the reference's production entry points for the hot path.

Reference (pangenome/seqrush): ``Args`` src/seqrush.rs:17-152, ``Sequence``
:272-277, ``load_sequences`` :1801-1837, ``SeqRush::new`` :308-336,
``build_graph`` :433-458, ``align_and_unite_with_allwave`` :611-757,
``run_seqrush`` :1839-1853.  Everything that computes calls the C ABI
(``include/seqrush_amd.h``); there is no Python or CPU alignment path.
"""
import ctypes as C
import dataclasses
from typing import List, Optional, Sequence as Seq

import numpy as np

from . import _lib
from ._lib import ParamsC, SeqSetC, AlignmentsC, check, SeqRushError

SR_MEM_HIGH, SR_MEM_ULTRALOW = 0, 3


# --------------------------------------------------------------------------- args
@dataclasses.dataclass
class Args:
    """Flag surface of the reference CLI that the hot path reads (src/seqrush.rs:17-152)."""
    sequences: str = ""                     # -s
    output: str = "output.gfa"              # -o
    min_match_length: int = 0               # -k
    threads: int = 4                        # -t (host threads; the device path ignores it)
    scores: str = "0,5,8,2,24,1"            # -S
    orientation_scores: str = "0,1,1,1"     # --orientation-scores
    max_divergence: Optional[float] = None  # -d
    sparsification: str = "none"            # -x
    paf: Optional[str] = None               # -p: replay alignments from a PAF file instead of aligning
    output_alignments: Optional[str] = None  # --output-alignments
    no_compact: bool = False                # --no-compact (default: compact + renumber, bidirected_gfa_writer.rs:39-51)
    no_sort: bool = True                    # only --no-sort is implemented (Ygs layout out of scope)
    aligner: str = "allwave"
    verbose: bool = False
    device: int = 0                         # added: HIP device ordinal
    shard_rank: int = 0                     # added: multi-GPU pair shard
    shard_count: int = 1
    gpus: int = 1                           # added: --gpus N, one process per GPU under torch.distributed.run


@dataclasses.dataclass
class AlignmentScores:
    """src/seqrush.rs:154-250"""
    match_score: int
    mismatch_penalty: int
    gap1_open: int
    gap1_extend: int
    gap2_open: Optional[int] = None
    gap2_extend: Optional[int] = None

    @staticmethod
    def parse(scores_str: str) -> "AlignmentScores":
        p = ParamsC()
        L = _lib.load()
        L.sr_default_params(C.byref(p))
        check(L.sr_parse_scores(scores_str.encode(), C.byref(p)))
        two = p.gap_open2 >= 0
        return AlignmentScores(p.match_score, p.mismatch_penalty, p.gap_open1, p.gap_ext1,
                               p.gap_open2 if two else None, p.gap_ext2 if two else None)

    @staticmethod
    def parse_orientation(scores_str: str) -> "AlignmentScores":
        p = ParamsC()
        L = _lib.load()
        L.sr_default_params(C.byref(p))
        check(L.sr_parse_orientation_scores(scores_str.encode(), C.byref(p)))
        return AlignmentScores(p.ori_match, p.ori_mismatch, p.ori_gap_open, p.ori_gap_ext)


@dataclasses.dataclass
class Sequence:
    """src/seqrush.rs:272-277"""
    id: str
    data: bytes
    offset: int = 0


def load_sequences(file_path: str) -> List[Sequence]:
    """FASTA loader with the reference's exact record rules (src/seqrush.rs:1801-1837):
    id = first whitespace token after '>', lines trimmed and concatenated verbatim,
    offset = running sum, records with an empty id are dropped."""
    sequences: List[Sequence] = []
    current_id = ""
    current = bytearray()
    offset = 0
    with open(file_path, "rb") as fh:
        text = fh.read()
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    ws = b" \t\n\r\x0b\x0c"
    for line in lines:
        if line.endswith(b"\r"):
            line = line[:-1]
        if line.startswith(b">"):
            if current_id:
                sequences.append(Sequence(current_id, bytes(current), offset))
                offset += len(current)
                current = bytearray()
            toks = line[1:].split()
            current_id = toks[0].decode() if toks else ""
        else:
            current.extend(line.strip(ws))
    if current_id:
        sequences.append(Sequence(current_id, bytes(current), offset))
    return sequences


# --------------------------------------------------------------------------- C ABI wrappers
class SeqSet:
    """Owns the buffers behind an ``sr_seqset``."""

    def __init__(self, records: Seq):
        """records: iterable of (name, bytes) or Sequence"""
        names, datas = [], []
        for r in records:
            if isinstance(r, Sequence):
                names.append(r.id); datas.append(bytes(r.data))
            else:
                names.append(r[0]); datas.append(bytes(r[1]))
        self.names = names
        self.lengths = [len(d) for d in datas]
        self._bases = b"".join(datas)
        self._offsets = np.zeros(len(datas) + 1, dtype=np.uint64)
        np.cumsum(self.lengths, out=self._offsets[1:])
        self._names_c = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        self.c = SeqSetC(len(datas), self._bases, self._offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                         C.cast(self._names_c, C.POINTER(C.c_char_p)))

    @property
    def n(self):
        return len(self.names)

    @property
    def total_length(self):
        return int(self._offsets[-1])

    def offset(self, i):
        return int(self._offsets[i])

    def seq(self, i):
        return self._bases[int(self._offsets[i]): int(self._offsets[i + 1])]


class Params:
    """``sr_params`` with the reference defaults (src/seqrush.rs:33-75)."""

    def __init__(self, **kw):
        self.c = ParamsC()
        _lib.load().sr_default_params(C.byref(self.c))
        for k, v in kw.items():
            self.set(k, v)

    def set(self, k, v):
        if k == "scores":
            check(_lib.load().sr_parse_scores(v.encode(), C.byref(self.c)))
        elif k == "orientation_scores":
            check(_lib.load().sr_parse_orientation_scores(v.encode(), C.byref(self.c)))
        elif k == "sparsification":
            check(_lib.load().sr_parse_sparsification(v.encode(), C.byref(self.c)))
        elif k == "max_divergence":
            self.c.max_divergence = -1.0 if v is None else float(v)
        else:
            if not hasattr(self.c, k):
                raise AttributeError(k)
            setattr(self.c, k, v)

    @staticmethod
    def from_args(args: Args) -> "Params":
        p = Params(scores=args.scores, orientation_scores=args.orientation_scores,
                   sparsification=args.sparsification, max_divergence=args.max_divergence)
        p.c.min_match_len = args.min_match_length
        p.c.device = args.device
        p.c.shard_rank, p.c.shard_count = args.shard_rank, args.shard_count
        return p


class Alignments:
    """Owned ``sr_alignments`` (Seam 1 result)."""

    def __init__(self, ptr):
        self._p = ptr
        a = ptr.contents
        n = int(a.n)
        self.n = n

        def arr(p, dt, m):
            return np.ctypeslib.as_array(p, shape=(max(m, 1),))[:m].astype(dt, copy=True)
        self.query_idx = arr(a.query_idx, np.uint32, n)
        self.target_idx = arr(a.target_idx, np.uint32, n)
        self.is_reverse = arr(a.is_reverse, np.uint8, n)
        self.score = arr(a.score, np.int32, n)
        self.query_start = arr(a.query_start, np.uint64, n)
        self.query_end = arr(a.query_end, np.uint64, n)
        self.target_start = arr(a.target_start, np.uint64, n)
        self.target_end = arr(a.target_end, np.uint64, n)
        self.cigar_off = arr(a.cigar_off, np.uint64, n + 1)
        self.cigar_ops = arr(a.cigar_ops, np.uint32, int(self.cigar_off[-1]) if n else 0)

    def cigar(self, i: int) -> str:
        ops = self.cigar_ops[int(self.cigar_off[i]): int(self.cigar_off[i + 1])]
        return "".join(f"{int(o) >> 4}{'=XID'[int(o) & 3]}" for o in ops)

    def raw_cigar_bytes(self, i: int) -> bytes:
        """per-column raw WFA2 alphabet (M X I D), i.e. allwave's ``cigar_bytes``"""
        ops = self.cigar_ops[int(self.cigar_off[i]): int(self.cigar_off[i + 1])]
        tr = {0: b"M", 1: b"X", 2: b"D", 3: b"I"}   # undo the I<->D swap of src/wfa.rs:25-31
        return b"".join(tr[int(o) & 3] * (int(o) >> 4) for o in ops)

    def write_paf(self, seqset: SeqSet, path: str):
        check(_lib.load().sr_write_paf(self._p, C.byref(seqset.c), path.encode()))

    def close(self):
        if self._p:
            _lib.load().sr_alignments_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """Resident device context (``sr_ctx``): what ``bench.py`` times."""

    def __init__(self, device: int = 0):
        self.L = _lib.load()
        self._h = C.c_void_p()
        check(self.L.sr_ctx_create(device, C.byref(self._h)))
        self.seqset = None

    def set_stream(self, stream_ptr: int):
        check(self.L.sr_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def load(self, seqset: SeqSet, params: Params):
        self.seqset = seqset
        check(self.L.sr_ctx_load(self._h, C.byref(seqset.c), C.byref(params.c)))

    def load_pairs(self, seqset: SeqSet, params: Params, pairs):
        """explicit ordered (query, target) pair list instead of the all-vs-all enumeration"""
        self.seqset = seqset
        q = np.ascontiguousarray([a for a, _ in pairs], dtype=np.uint32)
        t = np.ascontiguousarray([b for _, b in pairs], dtype=np.uint32)
        check(self.L.sr_ctx_load_pairs(self._h, C.byref(seqset.c), C.byref(params.c),
                                       q.ctypes.data_as(C.POINTER(C.c_uint32)), t.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       len(pairs)))

    def pairs(self):
        """this rank's (query, target) pair list after sparsification and sharding"""
        q = C.POINTER(C.c_uint32)(); t = C.POINTER(C.c_uint32)(); cnt = C.c_uint64()
        check(self.L.sr_ctx_pairs(self._h, C.byref(q), C.byref(t), C.byref(cnt)))
        m = int(cnt.value)
        qa = np.ctypeslib.as_array(q, shape=(max(m, 1),))[:m].copy()
        ta = np.ctypeslib.as_array(t, shape=(max(m, 1),))[:m].copy()
        self.L.sr_free(C.cast(q, C.c_void_p)); self.L.sr_free(C.cast(t, C.c_void_p))
        return list(zip(qa.tolist(), ta.tolist()))

    @property
    def num_batches(self):
        return int(self.L.sr_ctx_num_batches(self._h))

    def workspace_report(self):
        import json
        return json.loads(self.L.sr_ctx_workspace_report(self._h).decode() or "{}")

    def run(self):
        """align + unite of the whole shard, batch after batch (PAF context: unite only)"""
        check(self.L.sr_ctx_run(self._h))

    def align_all(self, unite: bool = False) -> "Alignments":
        p = C.POINTER(AlignmentsC)()
        check(self.L.sr_ctx_align_all(self._h, 1 if unite else 0, C.byref(p)))
        return Alignments(p)

    def pair_results(self):
        """(score, is_reverse, n_cigar_ops) arrays over this rank's pairs; resident for every batch"""
        n = self.num_pairs
        sc = np.zeros(max(n, 1), dtype=np.int32); rv = np.zeros(max(n, 1), dtype=np.uint8); co = np.zeros(max(n, 1), dtype=np.uint32)
        check(self.L.sr_ctx_pair_results(self._h, sc.ctypes.data_as(C.POINTER(C.c_int32)),
                                         rv.ctypes.data_as(C.POINTER(C.c_uint8)), co.ctypes.data_as(C.POINTER(C.c_uint32))))
        return sc[:n], rv[:n], co[:n]

    def load_paf(self, seqset: SeqSet, params: Params, paf_path: str):
        """`seqrush -p`: replay the records of a PAF file (then unite()); there is no alignment stage"""
        self.seqset = seqset
        check(self.L.sr_ctx_load_paf(self._h, C.byref(seqset.c), C.byref(params.c), paf_path.encode()))

    def build_gfa(self, compact: bool = False):
        """graph induction on the device from this context's union-find (+ compaction and
        renumbering, src/bidirected_gfa_writer.rs:39-51) + GFA text;
        -> (gfa_text, n_nodes, n_edges), byte-identical to build_gfa(seqset, download_labels(), compact)"""
        out = C.c_void_p(); nn = C.c_uint64(); ne = C.c_uint64()
        check(self.L.sr_ctx_build_gfa_opts(self._h, C.byref(self.seqset.c), 1 if compact else 0, C.byref(out),
                                           C.byref(nn), C.byref(ne)))
        text = C.cast(out, C.c_char_p).value.decode()
        self.L.sr_free(out)
        return text, int(nn.value), int(ne.value)

    def reset_uf(self):
        check(self.L.sr_ctx_reset_uf(self._h))

    def align(self):
        check(self.L.sr_ctx_align(self._h))

    def unite(self):
        check(self.L.sr_ctx_unite(self._h))

    def sync(self):
        check(self.L.sr_ctx_sync(self._h))

    def alignments(self) -> Alignments:
        p = C.POINTER(AlignmentsC)()
        check(self.L.sr_ctx_alignments(self._h, C.byref(p)))
        return Alignments(p)

    @property
    def uf_size(self):
        return int(self.L.sr_ctx_uf_size(self._h))

    @property
    def num_pairs(self):
        return int(self.L.sr_ctx_num_pairs(self._h))

    @property
    def dp_cells(self):
        return int(self.L.sr_ctx_dp_cells(self._h))

    def download_uf(self) -> np.ndarray:
        out = np.zeros(self.uf_size, dtype=np.uint64)
        check(self.L.sr_ctx_download_uf(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def download_labels(self) -> np.ndarray:
        out = np.zeros(self.uf_size, dtype=np.uint64)
        check(self.L.sr_ctx_download_labels(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def labels_device(self, dev_ptr: int):
        check(self.L.sr_ctx_labels_device(self._h, C.c_void_p(dev_ptr)))

    def merge_labels(self, dev_ptr: int, count: int):
        check(self.L.sr_ctx_merge_labels(self._h, C.c_void_p(dev_ptr), count))

    def labels_device_u32(self, dev_ptr: int):
        check(self.L.sr_ctx_labels_device_u32(self._h, C.c_void_p(dev_ptr)))

    def merge_labels_u32(self, dev_ptr: int, count: int):
        check(self.L.sr_ctx_merge_labels_u32(self._h, C.c_void_p(dev_ptr), count))

    @property
    def align_kernel(self) -> str:
        n = self.L.sr_ctx_align_kernel(self._h)
        return n.decode() if n else ""

    def kernel_ms(self, which: int) -> float:
        ms = C.c_float()
        check(self.L.sr_ctx_kernel_ms(self._h, which, C.byref(ms)))
        return float(ms.value)

    def counters(self):
        """device counters of the last align / unite (include/seqrush_amd.h sr_ctx_counters_all): every name maps to ONE slot"""
        out = (C.c_uint64 * 48)()
        n = self.L.sr_ctx_counters_all(self._h, out, 48)
        if n < 0:
            check(n)
        return dict(row_bytes_loaded=int(out[16]), row_bytes_stored=int(out[17]), tk_recompute=int(out[18]),
                    wf_cells=int(out[0]), wf_steps=int(out[1]), base_segments=int(out[2]),
                    breakpoint_searches=int(out[3]), united_bases=int(out[4]), match_runs=int(out[5]),
                    ticks_orientation=int(out[6]), ticks_breakpoint=int(out[7]), ticks_base=int(out[8]),
                    bp_passes=int(out[9]), ticks_pair=int(out[10]), tk_pass=int(out[11]),
                    tk_barrier=int(out[12]), tk_setup_first=int(out[13]), tk_phase2=int(out[14]),
                    tk_tail=int(out[15]), bp_filter_units=int(out[19]), bp_candidates=int(out[20]),
                    bp_exact_units=int(out[21]), bp_rounds=int(out[22]), tk_backtrace=int(out[23]), tk_emit=int(out[24]),
                    tk_ctl_section=int(out[25]), tk_ctl_mak=int(out[26]), tk_ctl_segments=int(out[27]), tk_p2_list=int(out[28]),
                    tk_p2_filter=int(out[29]), tk_p2_exact=int(out[30]), tk_p2_replay=int(out[31]),
                    st_wait_cycles=int(out[32]), st_body_cycles=int(out[33]), st_tiles=int(out[34]), st_ext_iters=int(out[35]),
                    lds_row_bytes=int(out[36]), base_requeues=int(out[37]),
                    bounds_first=[int(out[40]), int(out[41]), int(out[42]), int(out[43])],
                    experiment=[int(out[44]), int(out[45]), int(out[46]), int(out[47])])

    def close(self):
        if self._h:
            self.L.sr_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_gfa(seqset: SeqSet, labels: np.ndarray, compact: bool = False):
    """Graph induction + GFA text from canonical labels (consumer A9; --no-sort, with or without --no-compact).
    -> (gfa_text, n_nodes, n_edges)"""
    L = _lib.load()
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    out = C.c_void_p(); nn = C.c_uint64(); ne = C.c_uint64()
    check(L.sr_build_gfa_opts(C.byref(seqset.c), labels.ctypes.data_as(C.POINTER(C.c_uint64)), 1 if compact else 0,
                              C.byref(out), C.byref(nn), C.byref(ne)))
    text = C.cast(out, C.c_char_p).value.decode()
    L.sr_free(out)
    return text, int(nn.value), int(ne.value)


def build_gfa_from_nodes(seqset: SeqSet, nodes: np.ndarray, compact: bool = False):
    """Graph induction from a RAW uf_rush node array with the reference's root rule (src/bidirected_builder.rs:46-48,
    176-182: node base = base at the offset of the component's union-find root).  -> (gfa_text, n_nodes, n_edges)"""
    L = _lib.load()
    nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
    out = C.c_void_p(); nn = C.c_uint64(); ne = C.c_uint64()
    check(L.sr_build_gfa_from_nodes(C.byref(seqset.c), nodes.ctypes.data_as(C.POINTER(C.c_uint64)), 1 if compact else 0,
                                    C.byref(out), C.byref(nn), C.byref(ne)))
    text = C.cast(out, C.c_char_p).value.decode()
    L.sr_free(out)
    return text, int(nn.value), int(ne.value)


class HostUnionFind:
    """uf_rush node array on the host (sr_uf_*_host: same packing / halving / rank / tie rule as uf_rush lib.rs:112-208,
    one thread): SeqRush::new state, unite, merge of gathered canonical label arrays, canonical labels."""

    def __init__(self, total_len: int):
        self.n = 2 * total_len + 2
        self.nodes = np.zeros(self.n, dtype=np.uint64)
        check(_lib.load().sr_uf_init_host(self._p(self.nodes), self.n, total_len))

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.POINTER(C.c_uint64))

    def unite(self, x: int, y: int) -> bool:
        r = _lib.load().sr_uf_unite_host(self._p(self.nodes), self.n, x, y)
        if r < 0:
            check(r)
        return r == 1

    def merge_labels(self, label_arrays):
        lab = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.uint64) for a in label_arrays]))
        assert len(lab) % self.n == 0
        check(_lib.load().sr_uf_merge_labels_host(self._p(self.nodes), self.n, self._p(lab), len(lab) // self.n))

    def canonical_labels(self) -> np.ndarray:
        out = np.zeros(self.n, dtype=np.uint64)
        check(_lib.load().sr_uf_canonical_labels_host(self._p(self.nodes), self.n, self._p(out)))
        return out


def uf_find(nodes: np.ndarray, x: int) -> int:
    nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
    return int(_lib.load().sr_uf_find(nodes.ctypes.data_as(C.POINTER(C.c_uint64)), len(nodes), x))


# --------------------------------------------------------------------------- SeqRush
class SeqRush:
    """Mirror of ``SeqRush`` (src/seqrush.rs:298-336, 433-458) over the device path."""

    def __init__(self, sequences: List[Sequence], device: int = 0):
        for s in sequences:                                  # :310-317
            if len(s.data) == 0:
                raise ValueError(
                    f"Empty sequences are not allowed: sequence '{s.id}' has length 0")
        off = 0
        for s in sequences:
            s.offset = off
            off += len(s.data)
        self.sequences = sequences
        self.total_length = off
        self.seqset = SeqSet(sequences)
        self.ctx = Context(device)
        self.labels = None
        self.alignments = None

    def align_and_unite(self, args: Args):
        """align_and_unite_with_allwave (src/seqrush.rs:611-757) on the device"""
        if args.paf is not None:                      # align_and_unite_from_paf (src/seqrush.rs:510-609)
            print(f"Reading alignments from PAF file: {args.paf}")
            self.ctx.load_paf(self.seqset, Params.from_args(args), args.paf)
            self.ctx.run()
            self.ctx.sync()
            self.labels = self.ctx.download_labels()
            self.ctx.sync()
            return
        if args.aligner.lower() != "allwave":
            raise SeqRushError(-6, f"aligner '{args.aligner}' is out of scope; only 'allwave'")
        params = Params.from_args(args)
        self.ctx.load(self.seqset, params)
        n = len(self.sequences)
        print(f"Total sequence pairs: {n * n} (sparsification: {args.sparsification})")
        if args.output_alignments:
            al = self.ctx.align_all(unite=True)         # batches: align, copy the CIGARs out, unite
            print(f"Writing alignments to {args.output_alignments}")
            al.write_paf(self.seqset, args.output_alignments)
            al.close()
        else:
            self.ctx.run()
        self.ctx.sync()
        self.labels = self.ctx.download_labels()
        self.ctx.sync()

    def build_graph(self, args: Args):
        print(f"Building graph with {len(self.sequences)} sequences "
              f"(total length: {self.total_length})")
        self.align_and_unite(args)
        self.write_gfa(args)

    def write_gfa(self, args: Args):
        if not args.no_sort:
            raise SeqRushError(-6, "only --no-sort output is implemented (the Ygs layout -- path-guided SGD, grooming, "
                                   "topological sort -- is outside the hot path and not reproducible run to run, "
                                   "SURVEY 0.4); compaction runs unless --no-compact")
        text, _, _ = self.ctx.build_gfa(compact=not args.no_compact)   # graph induction on the device, compaction on the host
        with open(args.output, "w") as fh:
            fh.write(text)


def run_seqrush(args: Args):
    """src/seqrush.rs:1839-1853"""
    sequences = load_sequences(args.sequences)
    print(f"Loaded {len(sequences)} sequences")
    sr = SeqRush(sequences, device=args.device)
    sr.build_graph(args)
    print(f"Graph written to {args.output}")
    return sr


def run_seqrush_rank(args: Args):
    """One rank of `--gpus N` (started by torch.distributed.run, one process per GPU): this rank's cost-balanced
    shard of the pair list -> private forest -> ONE all-gather of canonical labels (u32 while 2N+2 < 2^32; RCCL over
    xGMI) -> replay-unite on every rank -> rank 0 induces the graph and writes the GFA."""
    import os
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    single_dev = os.environ.get("SR_BENCH_SINGLE_DEVICE") == "1"      # testing: all ranks on GPU 0, gloo
    dev = 0 if single_dev else local
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if single_dev:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    args.device, args.shard_rank, args.shard_count = dev, rank, world
    sequences = load_sequences(args.sequences)
    if rank == 0:
        print(f"Loaded {len(sequences)} sequences")
        print(f"Building graph with {len(sequences)} sequences (total length: {sum(len(s.data) for s in sequences)})")
    if os.environ.get("SR_TEST_FAIL_RANK") == str(rank):       # test hook: a rank that fails before the collective
        raise SeqRushError(-1, f"SR_TEST_FAIL_RANK: rank {rank} fails on purpose")
    sr = SeqRush(sequences, device=dev)
    ctx = sr.ctx
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.paf is not None:
        ctx.load_paf(sr.seqset, Params.from_args(args), args.paf)
        ctx.run()
    else:
        ctx.load(sr.seqset, Params.from_args(args))
        if rank == 0:
            n = len(sequences)
            print(f"Total sequence pairs: {n * n} (sparsification: {args.sparsification})")
        if args.output_alignments:
            al = ctx.align_all(unite=True)
            al.write_paf(sr.seqset, f"{args.output_alignments}.rank{rank}")
            al.close()
        else:
            ctx.run()
    ctx.sync()
    ufn = ctx.uf_size
    u32 = ufn < (1 << 32)
    ldt = torch.int32 if u32 else torch.int64
    lab = torch.empty(ufn, dtype=ldt, device="cuda")
    gathered = torch.empty(ufn * world, dtype=ldt, device="cuda")
    (ctx.labels_device_u32 if u32 else ctx.labels_device)(lab.data_ptr())
    torch.cuda.synchronize()
    if single_dev:
        parts = [torch.empty(ufn, dtype=ldt) for _ in range(world)]
        dist.all_gather(parts, lab.cpu())
        gathered.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(gathered, lab)
    (ctx.merge_labels_u32 if u32 else ctx.merge_labels)(gathered.data_ptr(), world)
    ctx.sync()
    dist.barrier()
    if rank == 0:
        if args.output_alignments:
            with open(args.output_alignments, "wb") as out:
                for r in range(world):
                    part = f"{args.output_alignments}.rank{r}"
                    with open(part, "rb") as fh:
                        out.write(fh.read())
                    os.remove(part)
            print(f"Writing alignments to {args.output_alignments}")
        sr.write_gfa(args)
        print(f"Graph written to {args.output}")
    dist.barrier()
    dist.destroy_process_group()
    return sr


def pair_list(n: int, params: Params):
    """This rank's ordered (query, target) pair list (host only)."""
    L = _lib.load()
    q = C.POINTER(C.c_uint32)(); t = C.POINTER(C.c_uint32)(); cnt = C.c_uint64()
    check(L.sr_pair_list(n, C.byref(params.c), C.byref(q), C.byref(t), C.byref(cnt)))
    m = int(cnt.value)
    out = [(int(q[i]), int(t[i])) for i in range(m)]
    L.sr_free(C.cast(q, C.c_void_p)); L.sr_free(C.cast(t, C.c_void_p))
    return out
