import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from eagleeverything_amd import _lib
from eagleeverything_amd.sharded import DeviceShard
n, L = 10000, 262144
lib = _lib.load()
sh = DeviceShard(n, L); sh.fill_synthetic(); sh.mode, sh.nslices = 1, 4
gen = torch.Generator(device=sh.dev); gen.manual_seed(1)
A = torch.randn((n, 64), generator=gen, device=sh.dev, dtype=torch.float64) / 64.0
Sm = torch.eye(n, dtype=torch.float64, device=sh.dev) * 0.4 + A @ A.T
V = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev)
sh.set_operands(Sm, V, torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64))
sh.scan_operands(); sh.vara_prepare()
host = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
dev = torch.empty(256 << 20, dtype=torch.uint8, device=sh.dev)
side = torch.cuda.Stream()
for with_kernel in (False, True, True):
    torch.cuda.synchronize()
    sh.vara_prepare(with_a=False)
    torch.cuda.synchronize()
    e0, e1, k0, k1, f0, f1 = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    if with_kernel:
        k0.record(); sh.vara_kernel(); k1.record()
    with torch.cuda.stream(side):
        e0.record(side)
        dev.copy_(host, non_blocking=True)
        e1.record(side)
        f0.record(side)
        dev.zero_()     # a fill kernel: needs CUs
        f1.record(side)
    torch.cuda.synchronize()
    print("vara kernel running: %s  H2D 256 MiB: %.2f ms (%.1f GB/s)  fill 256 MiB: %.2f ms  kernel: %s ms" % (
        with_kernel, e0.elapsed_time(e1), 0.268 / e0.elapsed_time(e1) * 1e3, f0.elapsed_time(f1), "%.1f" % k0.elapsed_time(k1) if with_kernel else "-"))
