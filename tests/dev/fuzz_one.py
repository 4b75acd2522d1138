"""Re-run one case of fuzz_gpu.py: python tests/dev/fuzz_one.py <n_cases_seed> <index>  (prints per-window detail)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_gpu as F
seed0, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
for i in range(idx + 1):
    c = F.draw_case(rng)
print(c)
for single in range(c['B']):
    c1 = dict(c); c1['B'] = 1; c1['N'] = [c['N'][single]]
    print('window', single, 'N', c['N'][single], F.run_case(c1, 1000 * seed0 + idx + single))
print('batch', F.run_case(c, 1000 * seed0 + idx))
