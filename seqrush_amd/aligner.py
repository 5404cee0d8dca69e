"""Mirror of the reference's ``Aligner`` plug-in seam (src/aligner.rs:1-103,
src/aligner/allwave_impl.rs:1-149) on top of ``sr_align_all``.

Same names, argument meaning and error behaviour: ``create_aligner(backend,
threads, verbose, frequency)`` returns an object whose
``align_sequences(sequences)`` yields one ``AlignmentRecord`` per ordered pair
(self pairs included, allwave_impl.rs:114-120), CIGAR in the ``= X I D``
alphabet with ``I`` = query-only (src/wfa.rs:25-31), strand '-' for a
reverse-complemented query (aligner.rs:18).
"""
import ctypes as C
import dataclasses
import enum
from typing import List, Optional

from . import _lib
from ._lib import AlignmentsC, check
from .seqrush import SeqSet, Params, Alignments


@dataclasses.dataclass
class AlignmentSequence:          # src/aligner.rs:5-9
    id: str
    seq: bytes


@dataclasses.dataclass
class AlignmentRecord:            # src/aligner.rs:12-24
    query_name: str
    query_len: int
    query_start: int
    query_end: int
    strand: str
    target_name: str
    target_len: int
    target_start: int
    target_end: int
    cigar: str


class AlignerBackend(enum.Enum):  # src/aligner.rs:36-61
    AllWave = "allwave"
    SweepGA = "sweepga"

    @staticmethod
    def from_str(s: str) -> "AlignerBackend":
        t = s.lower()
        if t == "allwave":
            return AlignerBackend.AllWave
        if t == "sweepga":
            return AlignerBackend.SweepGA
        raise ValueError(f"Unknown aligner: {s}")

    def __str__(self):
        return self.value


class Aligner:                    # trait Aligner, src/aligner.rs:27-33
    def align_sequences(self, sequences: List[AlignmentSequence]) -> List[AlignmentRecord]:
        raise NotImplementedError


class AllwaveAligner(Aligner):
    """AllwaveAligner (allwave_impl.rs:5-43).  The reference's ``new`` hard-codes
    penalties 2,4,4,2,24,1 with a positive match score that WFA2 cannot take
    as-is (its handling lives in the absent allwave crate); this mirror defaults
    to the production penalties 0,5,8,2,24,1 (seqrush.rs:45) and exposes
    ``with_params`` like the reference (allwave_impl.rs:32-43)."""

    def __init__(self, threads: int = 4, verbose: bool = False, params: Optional[Params] = None,
                 device: int = 0):
        self.threads = threads
        self.verbose = verbose
        self.params = params if params is not None else Params()
        self.params.c.device = device

    @staticmethod
    def with_params(threads: int, verbose: bool, params: Params) -> "AllwaveAligner":
        return AllwaveAligner(threads, verbose, params)

    def align_raw(self, sequences: List[AlignmentSequence]):
        ss = SeqSet([(s.id, s.seq) for s in sequences])
        p = C.POINTER(AlignmentsC)()
        check(_lib.load().sr_align_all(C.byref(ss.c), C.byref(self.params.c), C.byref(p)))
        return ss, Alignments(p)

    def align_sequences(self, sequences: List[AlignmentSequence]) -> List[AlignmentRecord]:
        ss, al = self.align_raw(sequences)
        out = []
        for i in range(al.n):
            q, t = int(al.query_idx[i]), int(al.target_idx[i])
            out.append(AlignmentRecord(
                query_name=sequences[q].id, query_len=len(sequences[q].seq),
                query_start=int(al.query_start[i]), query_end=int(al.query_end[i]),
                strand="-" if al.is_reverse[i] else "+",
                target_name=sequences[t].id, target_len=len(sequences[t].seq),
                target_start=int(al.target_start[i]), target_end=int(al.target_end[i]),
                cigar=al.cigar(i)))
        al.close()
        return out


def create_aligner(backend, threads: int = 4, verbose: bool = False,
                   frequency: Optional[int] = None) -> Aligner:
    """src/aligner.rs:64-96.  SweepGA is a different aligner (FastGA) and is out
    of scope: requesting it is an error, like a build without `use-sweepga`."""
    if isinstance(backend, str):
        backend = AlignerBackend.from_str(backend)
    if backend == AlignerBackend.AllWave:
        return AllwaveAligner(threads, verbose)
    raise RuntimeError("SweepGA aligner not available. Rebuild with --features use-sweepga")
