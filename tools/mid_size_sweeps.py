import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import run_configs as rc
for N,R in ((2000,10000),(4000,20000),(6000,30000)):
    r=rc.sweep_record("mid",N,R,4096); print(N, "%.4f ms"%r["ms"], "%.1f %%"%(100*r["frac_of_8TBps"]))
