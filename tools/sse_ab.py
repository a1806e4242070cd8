import sys, json, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import bench as b
import zen_amd
zen_amd.init(0)
hop5, M5 = 512, 51680
x5 = b.s_music(M5*hop5, seed=0)[None,:]
for opt in (0,1):
    zen_amd.set_option("no_sse_block", opt)
    r5 = b.block_run(zen_amd, b._Solo(), x5, zen_amd.OUTPUT_PERCUSSIVE, M5, 20, 3, 100.0, zen_amd.synchronize, hop=hop5, sse=True, copy_bord=False)
    print("no_sse_block", opt, "hops/s", M5*20/r5["dt"], {k: round(v["ms"]/20,4) for k,v in r5["breakdown"].items() if v["launches"]}, r5["checksum"])
    b.free_run(r5)
