import sys, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from graphaligner_amd import binding, synth
import oracle_binding as ob, parity_common as pc
g = synth.SynthGraph(synth.random_genome(8000000, 47), node_len=32, snp_every=45, indel_every=500, seed=48)
reads, seeds = synth.simulate_reads(g, 8000, 15000, sub=0.02, ins=0.08, dele=0.05, seed=43)
gg = binding.Graph(gfa=g.gfa())
b = gg.prepare(reads, seeds, 35, 0); b.run(); s = b.collect(summary=True)
import collections
print(collections.Counter(s['status'].tolist()), collections.Counter(s['failed'].tolist()))
bad = [i for i in range(len(reads)) if s['failed'][i] or s['status'][i]]
print('bad', bad[:10])
og = ob.OracleGraph(g.nodes, g.edges)
for i in bad[:3]:
    o = og.align(reads[i], [seeds[i]], 35)
    print(i, 'dev status', s['status'][i], 'oracle', o['status'], o['failed'], o['score'], o['message'])
idx = list(range(0, 8000, 400))
res = gg.align([reads[i] for i in idx], [seeds[i] for i in idx], 35)
for i, r in zip(idx, res):
    pc.compare_read(dict(r, trace=np.zeros((0,7),dtype=np.int64)), dict(og.align(reads[i],[seeds[i]],35), trace=np.zeros((0,7),dtype=np.int64)), 'dense %d' % i)
print('20 dense reads identical to oracle')
