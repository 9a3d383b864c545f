#!/usr/bin/env python3
"""Feasibility of a SPECTRAL per-marker bound for the digit-slice scan (GPU, torch fp64 for the diagnostics only).
The truncation error of marker i after S digits is m'^T H m' with H = sym(R)/2, R the residual of the folded W: |err_i| <= ||H||_2 * q2_i
(q2 = sum m'^2) next to the shipped worst case l1_i^2/2 * delta.  ||H||_2 <= (u/2) (||Ds||_2 + (n-1)/2), Ds = the symmetrised NEXT
digit (int8), u its weight, and ||Ds||_2^2 = lambda_max(Ds Ds) <= max row sum |Ds Ds| -- exact integer arithmetic an int8 MFMA SYRK
can do.  Prints, for the bench's operands: the norms, how loose each rigorous bound is, and how many markers a scan with S-1 digits
would flag / re-evaluate under either bound."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import numpy as np
import torch
import bench
from eagleeverything_amd.sharded import Collectives

n, L = int(os.environ.get("N", 10000)), int(os.environ.get("LM", 1000000))
args = bench.parse_args(["--n", str(n), "--markers", str(L)])
run = bench.Run(args, torch, None, Collectives(None), n, L, 0, 1, 0, "none")
MMt, _, _ = run.mmt_build(1)
run.make_operands(MMt)
sh = run.sh
sh.mode = 1
run.step()
torch.cuda.synchronize()
info = sh.vara_i8_info()
print("library: S = %s, certificate = %s" % (info, sh.certificate()))
vara = sh.vara[:L].clone()
a = sh.a[:L].clone()
l1 = sh.l1[:L, 0].double()
q2 = sh.l1[:L, 1].double()
np_ = sh.np_
Wu = sh.Wu                                   # folded upper triangle (diagonal = W_kk)
off = Wu - torch.diag(torch.diag(Wu))
mx = float(off.abs().max())
e = int(np.frexp(mx)[1])
S = 4
print("n_pad %d  max|Wu offdiag| %.4g  e %d  sum|W_kk| %.6g  mean W_kk %.4g" % (np_, mx, e, float(torch.diag(Wu).abs().sum()), float(torch.diag(Wu)[:n].mean())))
Q = torch.round(off * 2.0 ** (8 * S - e - 2))
d_last = torch.remainder(Q + 128, 256) - 128
del Q
Ds = d_last + d_last.T
del d_last
u = 2.0 ** (e + 2 - 8 * S)
delta_prev = 2.0 ** (e + 1 - 8 * (S - 1))     # the S-1 digit residual bound (round to nearest)
G = Ds @ Ds
gersh = float(G.abs().sum(1).max())
fro = float((Ds * Ds).sum().sqrt())
x = torch.randn(np_, dtype=torch.float64, device=Ds.device)
for _ in range(200):
    x = Ds @ x
    x /= x.norm()
lam = float((x @ (Ds @ x)).abs())
G2 = G @ G
p4 = float(G2.abs().sum(1).max()) ** 0.25
del G2
print("||Ds||_2 (power iteration, lower estimate) %.1f   2 sqrt(n) sigma = %.1f" % (lam, 2 * np.sqrt(np_) * float(Ds.std())))
print("rigorous: Frobenius %.1f (x%.1f)   sqrt(max row sum |Ds Ds|) %.1f (x%.2f)   (max row sum |(Ds Ds)^2|)^(1/4) %.1f (x%.2f)" %
      (fro, fro / lam, gersh ** 0.5, gersh ** 0.5 / lam, p4, p4 / lam))
for name, nb in (("power-iteration value (NOT rigorous)", lam), ("row sums of Ds Ds", gersh ** 0.5), ("row sums of (Ds Ds)^2", p4)):
    H = 0.5 * u * (nb + 0.5 * (np_ - 1))
    b_spec = H * q2
    b_wc = 0.5 * l1 * l1 * delta_prev * (1 + 2.0 ** -8)
    b = torch.minimum(b_spec, b_wc)
    flagged = int((b > 1e-7 * vara.abs()).sum())
    flagged_wc = int((b_wc > 1e-7 * vara.abs()).sum())
    lb = float((a * a / (vara + b)).max())
    cand = int(((a * a / (vara - b)) >= lb * (1 - 1e-9)).sum())
    print("%-40s ||H|| <= %.3g = %.1f delta_{S-1};  S-1 digits: flagged %d (worst-case bound alone: %d), candidates %d, max b/|vara| %.3g, median %.3g"
          % (name, H, H / delta_prev, flagged, flagged_wc, cand, float((b / vara.abs()).max()), float((b / vara.abs()).median())))
    # chooser analogue: a marker with q2 = n_pad against the diagonal term of a marker with n/2 non-zero entries
    print("    chooser: ||H|| n_pad = %.3g  vs  1e-7 * sum|W_kk| / 2 = %.3g" % (H * np_, 0.5e-7 * float(torch.diag(Wu).abs().sum())))
