"""`python -m seqrush_amd` -- thin CLI with the reference's flag surface for the hot path
(src/seqrush.rs:17-152, src/main.rs:4-7).  Everything that computes runs on the GPU."""
import argparse
import os
import subprocess
import sys

from .seqrush import Args, run_seqrush, run_seqrush_rank
from ._lib import SeqRushError


def main(argv=None):
    ap = argparse.ArgumentParser(prog="seqrush", description="MI355X-native seqrush hot path")
    ap.add_argument("-s", "--sequences", required=True)
    ap.add_argument("-o", "--output", default="output.gfa")
    ap.add_argument("-k", "--min-match-length", type=int, default=0)
    ap.add_argument("-t", "--threads", type=int, default=4)
    ap.add_argument("-S", "--scores", default="0,5,8,2,24,1")
    ap.add_argument("--orientation-scores", default="0,1,1,1")
    ap.add_argument("-d", "--max-divergence", type=float, default=None)
    ap.add_argument("-x", "--sparsify", dest="sparsification", default="none")
    ap.add_argument("-p", "--paf", default=None)
    ap.add_argument("--output-alignments", default=None)
    ap.add_argument("--no-compact", action="store_true")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--aligner", default="allwave")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node: the pair list is sharded, one process per GPU (torch.distributed.run, RCCL)")
    ns = ap.parse_args(argv)
    if ns.gpus > 1 and "RANK" not in os.environ:
        # start one process per GPU BEFORE anything here touches the GPU (never exec from a process that has)
        port = os.environ.get("MASTER_PORT", str(29400 + os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ns.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, "-m", "seqrush_amd"] + list(argv if argv is not None else sys.argv[1:])
        env = dict(os.environ)
        env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
        return subprocess.call(cmd, env=env)
    args = Args(sequences=ns.sequences, output=ns.output, min_match_length=ns.min_match_length,
                threads=ns.threads, scores=ns.scores, orientation_scores=ns.orientation_scores,
                max_divergence=ns.max_divergence, sparsification=ns.sparsification, paf=ns.paf,
                output_alignments=ns.output_alignments, no_compact=ns.no_compact, no_sort=ns.no_sort,
                aligner=ns.aligner, verbose=ns.verbose, device=ns.device, gpus=ns.gpus)
    try:
        if ns.gpus > 1:
            if int(os.environ.get("WORLD_SIZE", "1")) != ns.gpus:
                raise ValueError(f"--gpus {ns.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
            run_seqrush_rank(args)
        else:
            run_seqrush(args)
    except (SeqRushError, ValueError) as e:
        print(f"Error: {e}", file=sys.stderr)
        if ns.gpus > 1:
            _leave_ranks()
        return 1
    except BaseException:
        if ns.gpus > 1:                 # any other failure of a rank: the same quick exit
            import traceback
            traceback.print_exc()
            _leave_ranks()
        raise
    return 0


def _leave_ranks():
    """A rank that failed (load error, device fault bits from ctx.sync()) leaves at once with a non-zero code instead of
    finalising the interpreter with a live process group: the other ranks may already sit in the label all-gather or a
    barrier, and torch.distributed.run ends them as soon as one worker has failed -- not after the RCCL timeout."""
    sys.stderr.flush(); sys.stdout.flush()
    os._exit(1)


if __name__ == "__main__":
    sys.exit(main())
