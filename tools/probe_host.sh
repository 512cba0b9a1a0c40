nproc; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
grep -m1 "model name" /proc/cpuinfo; free -g | head -2
python3 - <<'PY'
import numpy as np, time, os
from threadpoolctl import threadpool_limits, threadpool_info
print([ (i['internal_api'], i['num_threads'], i.get('version')) for i in threadpool_info()])
import scipy.linalg as sl
for nt in (1,8,32,64):
    with threadpool_limits(limits=nt):
        a=np.random.rand(4096,4096); b=np.random.rand(4096,4096)
        a@b
        t=time.time(); a@b; dt=time.time()-t
        print('threads',nt,'dgemm 4096^3 %.3f s  %.1f GFLOP/s'%(dt,2*4096**3/dt/1e9))
        x=np.random.rand(16384,2048)
        t=time.time(); sl.qr(x,mode='economic',overwrite_a=True,check_finite=False); dt=time.time()-t
        print('   qr 16384x2048 %.3f s'%dt)
PY
