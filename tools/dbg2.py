import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gpu_sort_amd as gs
from oracle import oracle as O
dev = torch.device("cuda:0")
n = 1000
for name, fn, ref in (("uniform", lambda: gs.generate_uniform_keys(n, seed=3, start=5, device=dev), O.gen_uniform(n, 3, 5)),
                 ("zipf", lambda: gs.generate_zipf_keys(n, seed=3, start=5, device=dev), O.gen_zipf(n, 3, 5)),
                 ("and3", lambda: gs.generate_random_keys(n, seed=3, entropy_level=3, start=5, device=dev), O.gen_entropy_and(n, 3, 3, 5)),
                 ("enum", lambda: gs.generate_enumerated_values(n, start=5, device=dev), O.gen_enumerated(n, 5))):
    out = torch.full((n,), -7, dtype=torch.int32, device=dev)
    t = fn()
    torch.cuda.synchronize()
    g = t.cpu().numpy().view(np.uint32)
    print(name, np.array_equal(g, ref), g[:4], ref[:4])
