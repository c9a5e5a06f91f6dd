"""Time simple_knn._C.distCUDA2 on synthetic clouds (GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gaussian-mesh-splatting_amd"))
import numpy as np, torch
from simple_knn._C import distCUDA2
rng = np.random.default_rng(0)
for n in (100_000, 300_000, 1_000_000, 5_000_000):
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    for name, p in (("shell", d * (1 + 0.01 * rng.normal(size=(n, 1)))), ("uniform", rng.uniform(-1, 1, size=(n, 3)))):
        t = torch.from_numpy(p.astype(np.float32)).cuda()
        distCUDA2(t); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): distCUDA2(t)
        torch.cuda.synchronize()
        print(f"N={n} {name}: {(time.perf_counter()-t0)/5*1e3:.3f} ms", flush=True)
