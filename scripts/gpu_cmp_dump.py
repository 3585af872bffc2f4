import sys; sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import numpy as np, pyoracle
from window_io import read_dump
from limo_amd import default_options, ba
ctx=ba.Context(0)
o=default_options()
for k in range(2):
    w=read_dump('scratch_dump/solve_%06d.bin'%k)
    wo=w.copy(); ro,_=pyoracle.solve(wo,o)
    wg=w.copy(); rg=ctx.solve(wg,o)
    print(k,"oracle cost %.6f it %d term %d | gpu cost %.6f it %d term %d | dpose %.3e"%(ro["final_cost"],ro["iterations_total"],ro["termination"],rg["final_cost"],rg["iterations_total"],rg["termination"],np.abs(wo.kf_pose[:,4:]-wg.kf_pose[:,4:]).max()))
