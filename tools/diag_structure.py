#!/usr/bin/env python3
"""How the digit rule behaves on genotypes WITH population structure (GPU diagnostic): K sub-populations whose allele frequencies drift
from a common ancestral frequency (Balding-Nichols, Fst), operands from the model algebra on the actual MM^T as in bench.py.
Prints digits used / cut, the spectral bound, the certificate (flagged, re-evaluated, overflow) and the spread of b_i / |vara_i|."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from eagleeverything_amd.sharded import Collectives

n, L = int(os.environ.get("N", 4096)), int(os.environ.get("LM", 262144))
K, fst = int(os.environ.get("K", 3)), float(os.environ.get("FST", 0.1))
PMIN = float(os.environ.get("PMIN", 0.05))
args = bench.parse_args(["--n", str(n), "--markers", str(L)])
run = bench.Run(args, torch, None, Collectives(None), n, L, 0, 1, 0, "none")
sh = run.sh
dev = sh.dev
sh.fill_structured(K=K, fst=fst, seed=5, pmin=PMIN)
torch.cuda.synchronize()
MMt, _, _ = run.mmt_build(1)
ev = torch.linalg.eigvalsh(MMt)
print("top eigenvalues of MMt/max + 0.95 I:", [round(float(x), 3) for x in ev[-5:]], " median", round(float(ev[n // 2]), 3))
run.make_operands(MMt)
sh.mode = 1
for tune, name in ((29, "worst-case count"), (0, "with the spectral bound")):
    sh.L.eagle_dev_set_tune(sh.ctx, tune)
    sel, el, parts = run.timed(3, 1)
    torch.cuda.synchronize()
    S_used = sh.vara_i8_info()[0]
    print("%-24s digits used %d cut %d specH %.3g  step %.2f ms (vara kernel + extension %.2f, certify %.2f)  certificate %s  selected %s"
          % (name, S_used, sh.last_sliced, sh.last_specH, el / 3 * 1e3, parts["kern"] * 1e3, parts["cert"] * 1e3, sh.certificate(), sel))
    ct = sh.certificate()
    print("    certificate enforced 1.8 x %.0e (over the tight threshold: %d markers, switch above 512)"
          % (sh.last_budget_loose if ct["over_tight"] > 512 else sh.last_budget, ct["over_tight"]))
    wi = sh.w_info()
    print("    budget in force %.0e level %d;  W engine: int8=%d declined=%d (k,T)=(%d,%d)+(%d,%d) eta/mean|W_kk| %.2e  W %.2f ms"
          % (sh.last_budget, sh.last_level, wi["int8"], wi["declined"], wi["k1"], wi["T1"], wi["k2"], wi["T2"],
             wi["eta"] / wi["mean_diag"] if wi["mean_diag"] else 0.0, parts["w"] * 1e3))
sh.L.eagle_dev_set_tune(sh.ctx, 0)
if sh.last_specH > 0:
    q2 = sh.l1[:L, 1].double(); l1 = sh.l1[:L, 0].double()
    v = sh.vara[:L].abs()
    b = torch.minimum(sh.last_specH * q2, 0.5 * l1 * l1 * 2.0 ** (sh.last_e + 1 - 8 * S_used))
    r = (b / v)[v > 0]
    qs = torch.quantile(r, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64, device=dev))
    print("b_i/|vara_i| median %.3g  90%% %.3g  99%% %.3g  99.9%% %.3g  max %.3g   (enforced 9e-7); above: %d" % (*[float(x) for x in qs], int((r > 9e-7).sum())))
