import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, bench as b, zen_amd as z
z.init(0)
n = 26460000
x = b.s_music(n, seed=5)
din, dh, dp = z.DeviceBuffer(n), z.DeviceBuffer(n), z.DeviceBuffer(n)
din.upload(x)
for rep in range(2):
    for opt in (2, 0):
        z.set_option("no_istft_runs", opt)
        g = z.HPRIOffline(44100.0, 4096, 256, 2.0, 2.0)
        for i in range(2):
            g.process_device(din.ptr, n, n, harm=dh.ptr, perc=dp.ptr, out_stride=n)
        z.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            g.process_device(din.ptr, n, n, harm=dh.ptr, perc=dp.ptr, out_stride=n)
        z.synchronize()
        print("no_istft_runs", opt, "ms per 10-minute clip", round((time.perf_counter() - t0) / 5 * 1e3, 3), flush=True)
        del g
