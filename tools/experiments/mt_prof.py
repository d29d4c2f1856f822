import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from enstop_amd.engine import Engine
with Engine() as eng:
    for (n,m,nnz,k) in ((18846,173762,2950000,20),(1000000,100000,100000000,64)):
        eng.generate_synthetic(n,m,nnz,seed=0)
        rs=np.random.RandomState(1); eng.init_factors_numpy_stream(k, rs)
        for rep in range(3):
            rs=np.random.RandomState(1)
            t=time.perf_counter(); eng.init_factors_numpy_stream(k, rs); print("wall %.2f ms"%((time.perf_counter()-t)*1e3))
        eng.timing(True); eng.timing_reset()
        rs=np.random.RandomState(1)
        t=time.perf_counter(); eng.init_factors_numpy_stream(k, rs); print("timed wall %.2f ms"%((time.perf_counter()-t)*1e3))
        print(eng.timing_report())
        eng.timing(False)
