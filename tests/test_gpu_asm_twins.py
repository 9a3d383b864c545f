"""Every hand-scheduled (inline-asm) kernel against its compiler-scheduled twin, bit for bit, inside the GPU suite (VERDICT r2
item 8c): the asm blocks are opaque to hipcc's hazard recogniser and register allocator -- round 2 found an undeclared SCC clobber
that way -- so any change to them, or to the compiler, is caught by an exact A/B of integer results, at two shapes each:

  k_vara_i8p  (384 x 256 tile, asm-pipelined k-step)      vs  k_vara_i8w (tune 9: the same tile, hipcc's schedule)  and
                                                              k_vara_i8  (tune 8: the 256 x 256 tile)
  k_syrk_f4w  (384 x 256 tiles, asm-pipelined; n_pad >= 3072) and k_syrk_f4p (tune 10: 256 x 256, asm-pipelined)
                                                          vs  k_syrk_f4  (tune 9: 256 x 256, hipcc's schedule)
The compared quantities are exact integers (q of the digit slices, the int32 MM^T accumulator): equality is the only pass."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tune(sh, v):
    sh.L.eagle_dev_set_tune(sh.ctx, int(v))


@pytest.mark.parametrize("n,L", [(1900, 9001), (5000, 40000)])
def test_vara_digit_kernel_asm_against_compiler_scheduled_twins(n, L):
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=n)
    gen = torch.Generator(device=sh.dev)
    gen.manual_seed(n)
    A = torch.randn((n, 40), generator=gen, device=sh.dev, dtype=torch.float64) / 30.0
    S = 0.5 * torch.eye(n, dtype=torch.float64, device=sh.dev) + A @ A.T
    V = 0.6 * torch.eye(n, dtype=torch.float64, device=sh.dev) - 0.03 * (A[:, :6] @ A[:, :6].T)
    sh.set_operands(S, V, torch.randn(n, generator=gen, device=sh.dev, dtype=torch.float64))
    sh.mode = 1
    sh.nslices = 4
    sh.scan_operands(None)
    out = {}
    try:
        for tune in (0, 9, 8):
            _tune(sh, tune)
            sh.vara_prepare()       # the digit slices are stored in the layout the selected kernel reads (perm128 for the asm form)
            sh.vara_kernel()
            torch.cuda.synchronize()
            S_used = sh.vara_i8_info()[0]
            assert S_used == 4
            q = sh.ws[256:256 + 8 * S_used * sh.Lp].view(torch.int64).clone()   # [S][L_pad] exact integer row-dots (behind the 256-byte header)
            out[tune] = (q, sh.vara[:L].clone())
    finally:
        _tune(sh, 0)
    assert int(out[0][0].abs().max()) > 0
    for tune in (9, 8):
        assert torch.equal(out[0][0], out[tune][0]), "q differs between the asm kernel and tune %d" % tune
        assert torch.equal(out[0][1], out[tune][1]), "vara differs between the asm kernel and tune %d" % tune


@pytest.mark.parametrize("n,L", [(1500, 30000), (3300, 20000)])
def test_syrk_fp4_asm_against_compiler_scheduled_twin(n, L):
    import torch
    from eagleeverything_amd.sharded import DeviceShard
    sh = DeviceShard(n, L)
    sh.fill_synthetic(seed=L)
    M4 = sh.individual_major_fp4()
    out = {}
    try:
        for tune in (0, 10, 11, 9):      # 0: shipped choice for this size; 10: k_syrk_f4p; 11: k_syrk_f4w forced; 9: hipcc's schedule
            _tune(sh, tune)
            c32 = torch.zeros((sh.np_, sh.np_), dtype=torch.int32, device=sh.dev)
            assert sh.L.eagle_dev_mmt_accumulate_f4(sh.ctx, M4.data_ptr(), sh.np_, sh.Lp, sh.Lp // 2, c32.data_ptr(), sh._stream()) == 0
            torch.cuda.synchronize()
            out[tune] = c32
    finally:
        _tune(sh, 0)
    for tune in (0, 10, 11):
        assert torch.equal(out[tune], out[9]), "int32 MM^T accumulator differs between tune %d and the compiler-scheduled kernel" % tune
    G = sh.Mt8[:, :256].to(torch.float64)
    assert torch.equal((G.T @ G).to(torch.int32), out[9][:256, :256])
