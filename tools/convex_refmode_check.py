import sys, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as g
from oracle import pyoracle as po
pkg=g._load_pkg(); lib=pkg.load_library()
for N in (10,20):
    p=pkg.default_convex_params(N, pkg.MODE_REFERENCE, lib)
    rec=pkg.random_go1_convex_states(512, config_id=13 if N==20 else 12)
    s=pkg.Solver(p,1024,0,lib)
    f,i=s.convex_solve(rec)
    fam=s.kernel_for_batch(512)
    fo,io=po.convex_solve(po.default_convex_params(N,1), rec, threads=8)
    e=np.abs(f-fo).max(axis=1)
    print(f"convex reference mode N={N} [{fam}]: status equal {int((i['status']==io['status']).sum())}/512, iterations equal {int((i['iterations']==io['iterations']).sum())}/512 (max {i['iterations'].max()}), forces within 1e-6 N {int((e<1e-6).sum())}/512, median {np.median(e):.2e}, worst {e.max():.2e}", flush=True)
    for Bt in (1,512,1024):
        rt=pkg.random_go1_convex_states(Bt, config_id=13)
        ms=[]
        for r in range(6):
            s.convex_solve(rt)
            if r>=2: ms.append(s.last_kernel_ms())
        print(f"    B={Bt} [{s.kernel_for_batch(Bt)}]: {np.median(ms):.3f} ms -> {Bt/np.median(ms)/1e3:.3f} M/s", flush=True)
    s.close()
