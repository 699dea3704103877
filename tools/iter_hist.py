import sys, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
pkg=g._load_pkg(); lib=pkg.load_library()
for (B,N,cfg) in ((32768,10,4),(65536,20,3)):
    p=pkg.default_params(N,0,lib); rec=pkg.random_go1_trot_states(B,config_id=cfg)
    s=pkg.Solver(p,B,0,lib); f,i=s.solve(rec); ms=s.last_kernel_ms(); s.close()
    it=i['iterations']; h=np.bincount(it)
    tail=[(k,int((it>k).sum()), round(100*(it>k).mean(),2)) for k in range(12,it.max()+1)]
    print(B,N,'ms',ms,'mean',it.mean(),'max',it.max(),'\n  >k:',tail)
