#!/usr/bin/env python3
"""How many of a step's contacts were contacts in the step before (C2 run to a given step):
usage: contact_persistence.py [step]"""
import sys
sys.path[:0]=['.','particle-simulator_amd']
import numpy as np, bench
from psim_amd.particle_system import ParticleSystem
label,n,bounds,state,mask=bench.workload("c2")
ps=ParticleSystem(n,dict(bench.SIM),np.random.default_rng(42),bounds,stage_mask=mask,state=state,debug=True)
ps._calculate_gravity()
steps=int(sys.argv[1]) if len(sys.argv)>1 else 600
prev=None
for k in range(1,steps+3):
    ps.update()
    if k>=steps:
        ij=ps.debug_array("contact_ij").astype(np.int64)
        keys=np.unique(np.minimum(ij[:,0],ij[:,1])*(1<<32)+np.maximum(ij[:,0],ij[:,1]))
        if prev is not None:
            common=np.intersect1d(prev,keys,assume_unique=True).size
            print("step",k,"contacts",keys.size,"common with previous",common,"= %.3f of this, %.3f of previous"%(common/keys.size, common/prev.size), ps.counts()["solver_iterations"], flush=True)
        prev=keys
