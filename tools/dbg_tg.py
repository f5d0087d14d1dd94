import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import test_schedules as T
argv=['--workload','taylor_green','--n1','48']
on, c_on, r_on = T._run(argv, {}, steps=3)
off, c_off, r_off = T._run(argv, {'mass_fuse': 0}, steps=3)
print(c_on, r_on)
print(c_off, r_off)
for k in on:
    d=np.abs(on[k]-off[k]); print(k, d.max(), np.abs(off[k]).max(), np.argmax(d))
