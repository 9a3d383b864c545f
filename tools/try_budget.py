#!/usr/bin/env python3
"""GPU: the digit rule at budgets 5e-7 / 2e-7 / 1e-7 on the bench's operands (headline and C2): digits used / cut, level of the spectral
bound that took the digit off, step and phase times, certificate."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from eagleeverything_amd.sharded import Collectives
for (n, L) in ((10000, 1000000), (5000, 500000)):
    args = bench.parse_args(["--n", str(n), "--markers", str(L)])
    run = bench.Run(args, torch, None, Collectives(None), n, L, 0, 1, 0, "none")
    MMt, _, _ = run.mmt_build(1)
    run.make_operands(MMt)
    sh = run.sh
    sh.mode = 1
    for budget in (5e-7, 2e-7, 1e-7):
        sh._check(sh.L.eagle_set_scan_budget(sh.ctx, budget))
        sel, el, parts = run.timed(3, 1)
        S = sh.vara_i8_info()[0]
        print("n %d budget %.0e: digits used %d cut %d level %d specH %.3g  step %.2f ms (prep %.2f kern %.2f)  cert %s" % (n, budget, S, sh.last_sliced, sh.last_level, sh.last_specH, el / 3 * 1e3, parts["prep"] * 1e3, parts["kern"] * 1e3, sh.certificate()), flush=True)
    sh._check(sh.L.eagle_set_scan_budget(sh.ctx, 5e-7))
    del run, sh
    torch.cuda.empty_cache()
