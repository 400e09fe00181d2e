"""Wall / host time of pa_linreg_solve (bandit cfg5: d = 64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pearl_amd import _native as N
dev = torch.device("cuda", 0)
d = 64; D = d + 1
x = torch.randn(4096, D, device=dev)
A = (x.t() @ x).contiguous(); b = torch.randn(D, device=dev)
work = torch.empty(D * 2 * D, dtype=torch.float64, device=dev)
inv = torch.empty(D, D, device=dev); coefs = torch.empty(D, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
s = N.stream_ptr(dev)
def call():
    N.check(N.lib().pa_linreg_solve(A.data_ptr(), b.data_ptr(), 1.0, d, work.data_ptr(), inv.data_ptr(),
                                    coefs.data_ptr(), flag.data_ptr(), s))
for _ in range(5): call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): call()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host {1e6*(t1-t0)/200:.1f} us/call, wall {1e6*(t2-t0)/200:.1f} us/call")
ref = torch.linalg.inv(A.double() + torch.eye(D, device=dev, dtype=torch.float64)).float()
print("max rel err vs fp64 inverse:", float(((inv - ref).abs().max() / ref.abs().max())))
