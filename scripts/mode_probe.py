"""does the fast / slow mode of the alignment kernel depend on the process or on the allocation?"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.build()
from seqrush_amd import synth
from seqrush_amd.seqrush import SeqSet, Params, Context
recs = synth.config_c2(64); ss = SeqSet(recs)
for rep in range(5):
    ctx = Context(0); ctx.load(ss, Params())
    ts = []
    for i in range(3):
        ctx.reset_uf(); ctx.align(); ctx.unite(); ctx.sync(); ts.append(round(ctx.kernel_ms(0), 1))
    print("context", rep, "align kernel ms", ts, flush=True)
    ctx.close()
