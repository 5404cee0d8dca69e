"""Tile statistics of one step (experiment build: scripts/build_variant.sh stats "-DSR_BLK_ONLY_PROD -DSR_TILE_STATS",
SEQRUSH_AMD_LIB=seqrush_amd/libseqrush_amd_stats.so): aligner-passes, tiles, owned groups, aligners whose last tile would fit
a half wave, segments whose two aligners both do, passes, tile rounds as run and if such pairs of remainders shared a wave.
usage: tile_stats.py [C2|C4]"""
import json
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from seqrush_amd.seqrush import SeqSet, Params, Context

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
recs, spars, workload, _ = bench.build_config(cfg, 64 if cfg == "C2" else None)
ss = SeqSet(recs); prm = Params(sparsification=spars)
ctx = Context(0); ctx.load(ss, prm); ctx.reset_uf(); ctx.run(); ctx.sync()
c = ctx.counters(); rep = ctx.workspace_report(); ctx.close()
al, tiles, groups, small = c["st_wait_cycles"], c["st_body_cycles"], c["st_tiles"], c["st_ext_iters"]
both, passes, rounds, rounds_m = c["experiment"]
own = 58
out = {"config": cfg, "build": rep["kernel_build"], "threads": rep["threads_per_workgroup"], "aligner_passes": al, "tiles": tiles,
       "tiles_per_aligner_pass": tiles / max(al, 1), "owned_groups": groups, "owned_lane_fill": groups / max(tiles * own, 1),
       "aligners_with_last_tile_le_26_groups": small, "segments_with_both_small": both,
       "tiles_saved_if_merged": both, "tiles_saved_frac": both / max(tiles, 1),
       "passes": passes, "tile_rounds": rounds, "tile_rounds_if_merged": rounds_m, "rounds_saved_frac": 1 - rounds_m / max(rounds, 1),
       "wave_slot_fill": tiles / max(rounds * (rep["threads_per_workgroup"] // 64), 1)}
print(json.dumps(out))
