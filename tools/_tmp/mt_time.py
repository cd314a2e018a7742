import time, numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from magphase_amd.engine import get_engine
e = get_engine()
for n in (7_700_000, 30_000_000):
    for rep in range(4):
        np.random.seed(1)
        torch.cuda.synchronize(); t = time.perf_counter()
        x = e.numpy_global_uniform(n)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(n, "samples: %.2f ms" % (dt * 1e3))
    t = time.perf_counter(); np.random.uniform(-1, 1, n); print("  numpy host: %.1f ms" % ((time.perf_counter() - t) * 1e3))
