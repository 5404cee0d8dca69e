"""CPU test of the multi-GPU scheme with world_size 2 over gloo: the product's
pair sharding (sr_pair_list) + per-rank forests + ONE all-gather of canonical min-Pos labels +
replay-unite must reproduce the single-process partition.  The per-rank alignment work is done
by the oracle here (no GPU in this container); the SHARD and the MERGE are the product's: sr_pair_list and
sr_uf_init_host + sr_uf_merge_labels_host + sr_uf_canonical_labels_host (the host twins of the device merge;
sr_ctx_merge_labels_host itself uploads into the context's device forest, so it needs a GPU and is covered by
tests/test_gpu_parity.py::test_label_merge_* and the multi-rank bench / CLI tests there)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from seqrush_amd import synth
    from seqrush_amd.seqrush import Params, pair_list
    recs = synth.snp_family(6, 180, 0.05, 21, rc_every=3)
    prm = Params(); prm.c.shard_rank, prm.c.shard_count = rank, world
    mine = pair_list(len(recs), prm)
    o = ob.OracleSeqRush(records=recs)
    op = ob.default_params()
    for (qi, ti) in mine:
        a = o.align_pair(op, qi, ti)
        assert o.process_alignment(ob.cigar_bytes_to_string(a["cigar"]), qi, ti, 0, a["is_reverse"]) >= 0
    lab = torch.from_numpy(o.canonical_labels().astype(np.int64))
    gathered = [torch.empty_like(lab) for _ in range(world)]
    dist.all_gather(gathered, lab)                 # the single exchange
    from seqrush_amd.seqrush import HostUnionFind
    h = HostUnionFind(sum(len(sq) for _, sq in recs))          # product: SeqRush::new forest on the host
    h.merge_labels([g.numpy().astype(np.uint64) for g in gathered])   # product: replay-unite of the gathered label arrays
    merged = h.canonical_labels()
    if rank == 0:
        ref = ob.OracleSeqRush(records=recs)
        ref.align_and_unite(op)
        q.put((bool(np.array_equal(ref.canonical_labels(), merged)), len(mine)))
    else:
        q.put((True, len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_pair_shard_label_merge_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for ok, _ in res)
    assert sum(n for _, n in res) == 36


def test_elementwise_min_is_not_enough():
    """SURVEY 8e counter-example: an all-reduce(min) of label arrays loses 4~5."""
    a = np.array([0, 1, 2, 3, 4, 3]); b = np.array([0, 1, 2, 3, 4, 4])
    m = np.minimum(a, b)
    assert m[5] == 3 and m[4] == 4           # 4 and 5 end up in different sets: wrong
