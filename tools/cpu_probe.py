import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import pyoracle as po
pkg = po.pkg
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
os.system("grep -m1 'model name' /proc/cpuinfo; nproc")
p = po.default_params(10, 0)
for th in (1, 4, 16, 32, 64, 128, 256):
    n = 64 * th if th < 64 else 4096
    rec = pkg.random_go1_trot_states(n, config_id=2)
    t = time.perf_counter(); po.solve(p, rec, threads=th); dt = time.perf_counter() - t
    print('threads', th, 'n', n, 'solves/s %.0f' % (n / dt), 'per thread %.1f' % (n / dt / th))
