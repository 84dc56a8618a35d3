# condensed vs MS kernel on the unstable plants (GPU): errors vs the MS oracle + timing
import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, warnings
from tests.parity_util import run_unstable_plant
for rho in ((1.12,1.05),(1.2,1.1),(1.3,1.2)):
    for tr in ("MultipleShooting","SingleShooting"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                r=run_unstable_plant(B=16, rho=rho, transcription=tr, check=[0,5,9])
                print(rho,tr,'kind',r['kind'],'status',np.bincount(r['status'],minlength=3),'iters %.1f'%r['iters'].mean(),'err',r['err'],'cond %.1e'%r['cond'].max(),flush=True)
            except Exception as ex:
                print(rho,tr,'EXC',repr(ex)[:300],flush=True)
