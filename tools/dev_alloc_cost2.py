import numpy as np, time, ctypes
T = time.perf_counter
shape = (1, 480, 640, 2)
def loop(tag, n=30):
    ts = []
    g = None
    for k in range(n):
        t0 = T(); g = np.empty(shape); g[...] = 1.0; ts.append(T() - t0)      # alloc, first touch, and free of the previous one
    print(tag, 'median %.0f us min %.0f us' % (np.median(ts) * 1e6, min(ts) * 1e6), flush=True)
loop('default')
try:
    import numpy.core.multiarray as ma
    old = ma._set_madvise_hugepage(False); print('madvise hugepage was', old)
    loop('no hugepage madvise')
    ma._set_madvise_hugepage(True)
except Exception as e:
    print('no _set_madvise_hugepage', e)
libc = ctypes.CDLL(None)
print('mallopt', libc.mallopt(-3, 1 << 30), libc.mallopt(-1, 1 << 30))
loop('heap-resident')
ma._set_madvise_hugepage(False)
loop('heap-resident + no hugepage')
buf = np.empty(shape)
ts = []
for k in range(30):
    t0 = T(); buf[...] = 1.0; ts.append(T() - t0)
print('warm buffer fill median %.0f us' % (np.median(ts) * 1e6))
