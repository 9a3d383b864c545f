#!/usr/bin/env python3
"""rocSOLVER's symmetric eigensolver under rocprofv3 counter collection.  Runs the solver twice -- through this
library's C ABI (eagle_sym_eig -> rocsolver_dsyevd) and through torch.linalg.eigh -- and says which survive; run it as
`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/eigh_under_pmc.py <which>` and keep stderr."""
import faulthandler
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.enable()
import numpy as np

which = sys.argv[1] if len(sys.argv) > 1 else "abi"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n))
A = B @ B.T / n + np.eye(n)
if which == "abi":
    from eagleeverything_amd import rcpp_api
    w, U = rcpp_api.sym_eig(A)
    print("eagle_sym_eig ok: max |A U - U w| = %.3e" % np.abs(A @ U - U * w).max(), flush=True)
else:
    import torch
    t = torch.as_tensor(A, device="cuda")
    w, U = torch.linalg.eigh(t)
    torch.cuda.synchronize()
    print("torch.linalg.eigh ok: max |A U - U w| = %.3e" % (t @ U - U * w).abs().max().item(), flush=True)
