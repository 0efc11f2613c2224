import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys,time,numpy as np;sys.path.insert(0,%r);from oracle import oracle as orc;from arap_flow_amd import synth;"
        "f=synth.make_frame(854,480,seed=0);t=time.time();"
        "orc.frame(f['mask_red'],f['constraints'],numIter=1,nIterations=2,lIterations=400,dtype=np.float32,mode=1,trig=1);"
        "print(time.time()-t)" % ROOT)
for pol in ("active", "passive"):
    for n in (1, 4, 8, 16, 32):
        env = dict(os.environ, OMP_NUM_THREADS=str(n), OMP_WAIT_POLICY=pol, OMP_PROC_BIND="false")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(pol, n, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:])
