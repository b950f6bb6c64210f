import os, time, numpy as np, scipy.linalg as sl
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
import threadpoolctl
print(threadpoolctl.threadpool_info())
n=8192
rng=np.random.default_rng(0)
B=rng.standard_normal((n,n)); K=B@B.T/n+np.eye(n)
for th in (128, 64, 32, 16, 8):
    with threadpoolctl.threadpool_limits(th):
        A=K.copy(order='F'); t=time.perf_counter(); L,info=sl.lapack.dpotrf(A,lower=True,overwrite_a=True); dt=time.perf_counter()-t
        print('threads',th,'dpotrf n=8192 %.2fs %.0f GFLOP/s'%(dt,n**3/3/dt/1e9), flush=True)
