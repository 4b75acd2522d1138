import numpy as np, time
T = time.perf_counter
a = np.random.rand(480, 640, 2)
for rep in range(3):
    t0 = T(); b = np.empty_like(a); t1 = T(); b[...] = 1.0; t2 = T(); del b; t3 = T()
    print('empty %.0f us, first touch fill %.0f us, free %.0f us' % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
b = np.empty_like(a); b[...] = 1
t0 = T(); b[...] = 2.0; print('warm fill %.0f us' % ((T() - t0) * 1e6))
t0 = T(); c = np.ascontiguousarray(np.asarray(a, dtype=np.float64))[None]; print('asarray view %.0f us' % ((T() - t0) * 1e6))
t0 = T(); x = a * 1.01; print('mul (alloc + compute) %.0f us' % ((T() - t0) * 1e6))
t0 = T(); x = a * 1.02; print('mul again (frees the previous result) %.0f us' % ((T() - t0) * 1e6))
import os; print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
