"""ctypes loader for libseqrush_amd.so (the C ABI of include/seqrush_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C seqrush_amd/csrc``.  There is no fallback: if the shared object is
missing the import of any compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEQRUSH_AMD_LIB: another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("SEQRUSH_AMD_LIB") or os.path.join(_HERE, "libseqrush_amd.so")

# every symbol include/seqrush_amd.h declares
EXPORTS = [
    "sr_default_params", "sr_parse_scores", "sr_parse_orientation_scores",
    "sr_parse_sparsification", "sr_align_all", "sr_alignments_free", "sr_alignment_cigar",
    "sr_align_and_unite", "sr_uf_find", "sr_uf_same", "sr_write_paf", "sr_ctx_create",
    "sr_ctx_destroy", "sr_ctx_set_stream", "sr_ctx_load", "sr_ctx_reset_uf", "sr_ctx_align",
    "sr_ctx_unite", "sr_ctx_sync", "sr_ctx_alignments", "sr_ctx_download_uf", "sr_ctx_uf_size",
    "sr_ctx_num_pairs", "sr_ctx_dp_cells", "sr_ctx_labels_device", "sr_ctx_merge_labels",
    "sr_ctx_download_labels", "sr_ctx_kernel_ms", "sr_ctx_counters", "sr_build_gfa", "sr_free",
    "sr_last_error", "sr_abi_version", "sr_device_count", "sr_pair_list", "sr_ctx_align_kernel",
    "sr_ctx_load_paf", "sr_unite_paf", "sr_ctx_build_gfa", "sr_ctx_load_pairs", "sr_ctx_pairs",
    "sr_ctx_num_batches", "sr_ctx_workspace_report", "sr_ctx_run", "sr_ctx_align_all", "sr_ctx_pair_results",
    "sr_ctx_labels_device_u32", "sr_ctx_merge_labels_u32", "sr_ctx_counters_ext",
    "sr_build_gfa_opts", "sr_ctx_build_gfa_opts", "sr_ctx_merge_labels_host",
    "sr_uf_init_host", "sr_uf_unite_host", "sr_uf_merge_labels_host", "sr_uf_canonical_labels_host",
    "sr_build_gfa_from_nodes", "sr_ctx_counters_all", "sr_knobs_doc",
]


class SeqSetC(C.Structure):
    _fields_ = [("n", C.c_uint32), ("bases", C.c_char_p), ("offsets", C.POINTER(C.c_uint64)),
                ("names", C.POINTER(C.c_char_p))]


class ParamsC(C.Structure):
    _fields_ = [
        ("match_score", C.c_int32), ("mismatch_penalty", C.c_int32),
        ("gap_open1", C.c_int32), ("gap_ext1", C.c_int32),
        ("gap_open2", C.c_int32), ("gap_ext2", C.c_int32),
        ("ori_match", C.c_int32), ("ori_mismatch", C.c_int32),
        ("ori_gap_open", C.c_int32), ("ori_gap_ext", C.c_int32),
        ("min_match_len", C.c_uint64), ("max_divergence", C.c_double),
        ("exclude_self", C.c_int32), ("memory_mode", C.c_int32),
        ("sparsify_kind", C.c_int32), ("sparsify_factor", C.c_double),
        ("sparsify_seed", C.c_uint64), ("canonical_labels", C.c_int32), ("device", C.c_int32),
        ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32),
        ("tree_k_nearest", C.c_uint32), ("tree_k_farthest", C.c_uint32), ("tree_rand_frac", C.c_double),
        ("tree_kmer", C.c_uint32),
    ]


class AlignmentsC(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("query_idx", C.POINTER(C.c_uint32)), ("target_idx", C.POINTER(C.c_uint32)),
        ("is_reverse", C.POINTER(C.c_uint8)), ("score", C.POINTER(C.c_int32)),
        ("query_start", C.POINTER(C.c_uint64)), ("query_end", C.POINTER(C.c_uint64)),
        ("target_start", C.POINTER(C.c_uint64)), ("target_end", C.POINTER(C.c_uint64)),
        ("cigar_off", C.POINTER(C.c_uint64)), ("cigar_ops", C.POINTER(C.c_uint32)),
    ]


class SeqRushError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"seqrush_amd error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load the shared library and declare signatures.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(seqrush_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    PP, PS = C.POINTER(ParamsC), C.POINTER(SeqSetC)
    L.sr_default_params.argtypes = [PP]; L.sr_default_params.restype = None
    L.sr_parse_scores.argtypes = [C.c_char_p, PP]
    L.sr_parse_orientation_scores.argtypes = [C.c_char_p, PP]
    L.sr_parse_sparsification.argtypes = [C.c_char_p, PP]
    L.sr_align_all.argtypes = [PS, PP, C.POINTER(C.POINTER(AlignmentsC))]
    L.sr_alignments_free.argtypes = [C.POINTER(AlignmentsC)]; L.sr_alignments_free.restype = None
    L.sr_alignment_cigar.argtypes = [C.POINTER(AlignmentsC), u64, C.c_char_p, C.c_size_t]
    L.sr_alignment_cigar.restype = C.c_size_t
    L.sr_align_and_unite.argtypes = [PS, PP, C.POINTER(u64)]
    L.sr_uf_find.argtypes = [C.POINTER(u64), u64, u64]; L.sr_uf_find.restype = u64
    L.sr_uf_same.argtypes = [C.POINTER(u64), u64, u64, u64]
    L.sr_write_paf.argtypes = [C.POINTER(AlignmentsC), PS, C.c_char_p]
    L.sr_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.sr_ctx_destroy.argtypes = [vp]; L.sr_ctx_destroy.restype = None
    L.sr_ctx_set_stream.argtypes = [vp, vp]
    L.sr_ctx_load.argtypes = [vp, PS, PP]
    for f in ("sr_ctx_reset_uf", "sr_ctx_align", "sr_ctx_unite", "sr_ctx_sync"):
        getattr(L, f).argtypes = [vp]
    L.sr_ctx_alignments.argtypes = [vp, C.POINTER(C.POINTER(AlignmentsC))]
    L.sr_ctx_download_uf.argtypes = [vp, C.POINTER(u64)]
    for f in ("sr_ctx_uf_size", "sr_ctx_num_pairs", "sr_ctx_dp_cells"):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = u64
    L.sr_ctx_labels_device.argtypes = [vp, vp]
    L.sr_ctx_merge_labels.argtypes = [vp, vp, C.c_uint32]
    L.sr_ctx_download_labels.argtypes = [vp, C.POINTER(u64)]
    L.sr_ctx_merge_labels_host.argtypes = [vp, C.POINTER(u64), C.c_uint32]
    L.sr_ctx_kernel_ms.argtypes = [vp, i32, C.POINTER(C.c_float)]
    L.sr_ctx_align_kernel.argtypes = [vp]; L.sr_ctx_align_kernel.restype = C.c_char_p
    L.sr_ctx_load_paf.argtypes = [vp, PS, PP, C.c_char_p]
    L.sr_unite_paf.argtypes = [PS, PP, C.c_char_p, C.POINTER(u64)]
    L.sr_ctx_build_gfa.argtypes = [vp, PS, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.sr_ctx_counters.argtypes = [vp, C.POINTER(u64)]
    L.sr_ctx_counters_ext.argtypes = [vp, C.POINTER(u64)]
    L.sr_ctx_counters_all.argtypes = [vp, C.POINTER(u64), C.c_uint32]
    L.sr_knobs_doc.restype = C.c_char_p
    L.sr_ctx_load_pairs.argtypes = [vp, PS, PP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u64]
    L.sr_ctx_pairs.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64)]
    L.sr_ctx_num_batches.argtypes = [vp]; L.sr_ctx_num_batches.restype = C.c_uint32
    L.sr_ctx_workspace_report.argtypes = [vp]; L.sr_ctx_workspace_report.restype = C.c_char_p
    L.sr_ctx_run.argtypes = [vp]
    L.sr_ctx_align_all.argtypes = [vp, i32, C.POINTER(C.POINTER(AlignmentsC))]
    L.sr_ctx_pair_results.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
    L.sr_ctx_labels_device_u32.argtypes = [vp, vp]
    L.sr_ctx_merge_labels_u32.argtypes = [vp, vp, C.c_uint32]
    L.sr_build_gfa.argtypes = [PS, C.POINTER(u64), C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.sr_build_gfa_opts.argtypes = [PS, C.POINTER(u64), i32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.sr_ctx_build_gfa_opts.argtypes = [vp, PS, i32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.sr_uf_init_host.argtypes = [C.POINTER(u64), u64, u64]
    L.sr_uf_unite_host.argtypes = [C.POINTER(u64), u64, u64, u64]
    L.sr_uf_merge_labels_host.argtypes = [C.POINTER(u64), u64, C.POINTER(u64), C.c_uint32]
    L.sr_uf_canonical_labels_host.argtypes = [C.POINTER(u64), u64, C.POINTER(u64)]
    L.sr_build_gfa_from_nodes.argtypes = [PS, C.POINTER(u64), i32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.sr_pair_list.argtypes = [C.c_uint32, PP, C.POINTER(C.POINTER(C.c_uint32)),
                               C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64)]
    L.sr_free.argtypes = [vp]; L.sr_free.restype = None
    L.sr_last_error.restype = C.c_char_p
    L.sr_abi_version.restype = i32
    L.sr_device_count.restype = i32
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise SeqRushError(rc, load().sr_last_error().decode(errors="replace"))
