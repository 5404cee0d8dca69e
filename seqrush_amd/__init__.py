"""seqrush_amd -- MI355X-native implementation of the seqrush hot path.

All-vs-all WFA2/biWFA alignment -> match-run extraction -> lock-free bidirected
union-find, as hand-written HIP kernels behind the C ABI of
``include/seqrush_amd.h``.  This package is the host-side mirror of the
reference's interfaces for that path (``Aligner`` trait, ``SeqRush`` /
``run_seqrush``); it contains no CPU alignment path.
"""
from .aligner import (AlignmentSequence, AlignmentRecord, Aligner, AllwaveAligner,  # noqa: F401
                      AlignerBackend, create_aligner)
from .seqrush import (Args, Sequence, SeqRush, load_sequences, run_seqrush,  # noqa: F401
                      AlignmentScores, SeqSet, Params, Context, build_gfa)
from ._lib import SeqRushError  # noqa: F401

__all__ = ["AlignmentSequence", "AlignmentRecord", "Aligner", "AllwaveAligner", "AlignerBackend",
           "create_aligner", "Args", "Sequence", "SeqRush", "load_sequences", "run_seqrush",
           "AlignmentScores", "SeqSet", "Params", "Context", "build_gfa", "SeqRushError"]
