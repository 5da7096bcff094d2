import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from corintho_ai_amd import Trainer
from oracle import oracle as O
from tests import harness as H
import numpy as np
G,S,spe=[int(x) for x in sys.argv[1:4]]
t=Trainer(G,"",12345,S,spe,1.0,0.25,0,1,False,trace=True)
o=O.Trainer(G,seed=12345,max_searches=S,searches_per_eval=spe); o.enable_trace()
ra=H.play_generation(t,G,spe,H.hash_net,record=True)
rb=H.play_generation(o,G,spe,H.hash_net,record=True)
print('iters',ra['iterations'],rb['iterations'])
for i,(a,b) in enumerate(zip(ra['log'],rb['log'])):
    if a[1].shape!=b[1].shape or a[1].tobytes()!=b[1].tobytes():
        print('first diff at iteration',i,a[1].shape,b[1].shape)
        if a[1].shape==b[1].shape:
            rows=np.nonzero((a[1]!=b[1]).any(axis=1))[0]; print('rows',rows[:10])
            r=rows[0]; print(a[1][r]); print(b[1][r])
        break
for g in range(G):
    ta,tb=t.trace(g),o.trace(g)
    if not np.array_equal(ta,tb):
        n=min(len(ta),len(tb)); d=np.nonzero(ta[:n]!=tb[:n])[0]
        print('game',g,'trace len',len(ta),len(tb),'first diff idx',d[:3], ta[max(0,d[0]-8):d[0]+8] if len(d) else None, tb[max(0,d[0]-8):d[0]+8] if len(d) else None)
        break
print(t.stats()); print(o.counters())
