"""Marker-sharded multi-GPU driver: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The reference is single-process and has no collective; SURVEY.md section 8(e) derives the sharding:
markers are independent units for the scan (calculate_a_and_vara_rcpp.cpp:103-112) and a k-split for
MM^T = sum_s M_s M_s^T (calculateMMt_rcpp.cpp:95).

  rank r owns the contiguous marker range shard_range(L, r, world): a row range of Mt.ascii, a column window of
  M.ascii, resident in its HBM as int8.
  MM^T : each rank accumulates its exact int32 partial -> ONE all-reduce (sum) -> identical fp64 MM^T everywhere.
         Integer sum: order independent, bit-exact.
  scan : every rank holds S, V, a_hat (n x n host algebra, broadcast from rank 0), scans its shard, and the
         per-shard (max tsq, first global index) pairs are all-gathered (16 B/rank); the winner is the largest
         tsq, ties broken by the smallest global index = which(tsq == max)[1] of E/R/find_qtl.R:76-80.

`Collectives` only needs torch.distributed (works with gloo on CPU, which is how the N>1 logic is tested);
`DeviceShard` needs a GPU and calls the device-resident C ABI (section 2 of include/eagle_hip.h).
"""
import ctypes as C

import numpy as np


def shard_range(L, rank, world):
    """Contiguous marker range [m0, m1) of `rank`: floor split, remainder to the first ranks."""
    base, rem = divmod(int(L), int(world))
    m0 = rank * base + min(rank, rem)
    return m0, m0 + base + (1 if rank < rem else 0)


class Collectives:
    """The two exchange steps of the sharded path."""

    def __init__(self, dist=None, force=False):
        """force: issue the collectives even in a group of ONE rank (the RCCL calls of a one-GPU box: tests/test_sharded_rccl1.py)."""
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist is not None and dist.is_initialized())
        self.rank = dist.get_rank() if self.active else 0
        # gloo (CPU tests, and the 2-ranks-on-one-GPU rehearsal of bench.py) moves device tensors through the host
        self.via_host = self.active and dist.get_backend() == "gloo"

    def _staged(self, t, fn):
        if self.via_host and t.is_cuda:
            h = t.cpu()
            fn(h)
            t.copy_(h)
        else:
            fn(t)
        return t

    def sum_partial_mmt(self, c32, tile=256, dst=None):
        """In-place sum of the int32 partial MM^T tensors of all ranks (exact).  Only the 256 x 256 tiles on or above the
        diagonal are live (the kernels never write the others), so only those travel: half the bytes of the matrix.
        dst = None: all-reduce (every rank holds the sum); dst = r: reduce to rank r only -- MM^T goes back to ONE host
        process (calculateMMt_rcpp returns it to R), so the other ranks need not receive it: half the link traffic; their
        tensors are unspecified afterwards."""
        if self.active:
            np_ = c32.shape[0]
            if dst is None:
                red = lambda x: self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
            else:
                red = lambda x: self.dist.reduce(x, dst=dst, op=self.dist.ReduceOp.SUM)
            if c32.dim() == 2 and c32.shape[1] == np_ and np_ % tile == 0 and np_ // tile > 1:
                nt = np_ // tile
                import torch
                iu = torch.triu_indices(nt, nt, device=c32.device)
                tiles = c32.view(nt, tile, nt, tile).permute(0, 2, 1, 3)       # [ti][tj][tile][tile] view
                packed = tiles[iu[0], iu[1]].contiguous()                      # upper tiles only
                self._staged(packed, red)
                if dst is None or dst == self.rank:
                    tiles[iu[0], iu[1]] = packed
            else:
                self._staged(c32, red)
        return c32

    def all_gather_rows(self, full, mine):
        """full (world * rows x cols, contiguous) <- the ranks' row blocks `mine` (rows x cols), in rank order."""
        if not self.active:
            return full
        if self.via_host and full.is_cuda:
            import torch
            parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(self.world)]
            self.dist.all_gather(parts, mine.cpu())
            full.copy_(torch.cat(parts, dim=0))
        else:
            self.dist.all_gather_into_tensor(full, mine)
        return full

    def broadcast_(self, t, src=0):
        if self.active:
            self._staged(t, lambda x: self.dist.broadcast(x, src=src))
        return t

    def best_marker(self, local_tsqmax, local_index0_global, device=None):
        """All-gather (tsqmax, global 0-based index or -1) and pick find_qtl.R:76-80's marker.
        Returns (1-based global index or 0, tsqmax)."""
        import torch
        mine = torch.tensor([float(local_tsqmax), float(local_index0_global)], dtype=torch.float64,
                            device=None if self.via_host else device)
        if self.active:
            allv = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(allv, mine)
            allv = torch.stack(allv).cpu().numpy()
        else:
            allv = mine.cpu().numpy()[None, :]
        return pick_best(allv[:, 0], allv[:, 1].astype(np.int64))


def pick_best(tsqmax, index0):
    """Largest tsq among shards that have a valid index; ties -> smallest global index. NaN shards skipped."""
    best_v, best_i = np.nan, -1
    for v, i in zip(tsqmax, index0):
        if i < 0 or np.isnan(v):
            continue
        if best_i < 0 or v > best_v or (v == best_v and i < best_i):
            best_v, best_i = float(v), int(i)
    return best_i + 1, best_v


class DeviceShard:
    """One rank's genotype shard resident in HBM + the device-resident hot path on it."""

    def __init__(self, n, L_local, first_marker=0, device=0):
        import torch

        from . import _lib, rcpp_api
        self.torch = torch
        self.L = _lib.load()
        self.ctx = rcpp_api.context(device)
        self.dev = torch.device("cuda", device)
        self.n, self.Lloc, self.first = int(n), int(L_local), int(first_marker)
        self.np_ = int(self.L.eagle_pad(self.n))
        self.Lp = int(self.L.eagle_pad(self.Lloc))
        self.Mt8 = torch.zeros((self.Lp, self.np_), dtype=torch.int8, device=self.dev)  # marker-major
        self.M8 = None                                                                   # individual-major
        self.M4 = None                                                                   # individual-major, fp4
        self.a = torch.zeros(self.Lp, dtype=torch.float64, device=self.dev)
        self.vara = torch.zeros(self.Lp, dtype=torch.float64, device=self.dev)
        self._scratch = torch.zeros(3 * 1024, dtype=torch.float64, device=self.dev)
        self._best = torch.zeros(3, dtype=torch.int64, device=self.dev)  # eagle_best: {f64, i64, i64}
        self.Sa = self.Va = self.ahat = self.v = self.Wu = self.tmp = None
        self.W0 = self.v0 = None
        self.ws = None
        self._ws_mode = None
        self.Mt4 = None
        self.Mt8s = self.cshift = self.l1 = None
        self.cert_ws = None
        self.certified = True  # digit-slice scans are certified (eagle_dev_scan_certify) before the arg-max
        self.share_w = True  # multi-rank runs split the n^3 part of the scan operands (scan_operands)
        self.mode = 0
        self.nslices = 0  # 0 = chosen by the library from its error bound
        self.stochastic = False  # digits of W rounded at random (eagle_set_scan_rounding): probabilistic certificate, one digit fewer
        self.extend = True       # eagle_dev_vara_i8_extend after the vara kernel (tests switch it off to see the raw values)
        self.w_mode = 1          # eagle_set_w_mode for the digit-slice scan: 1 = W on the int8 engine from 4,096 padded individuals up, 0 = fp64 GEMM, 2 = int8 always

    # ---- plumbing -------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def _check(self, rc):
        if rc != 0:
            from ._lib import EagleError
            raise EagleError(rc, self.L.eagle_last_error(self.ctx).decode())

    # ---- genotype shard -------------------------------------------------------------------------
    def fill_synthetic(self, seed=20240601, chunk=8192):
        """HWE genotypes generated on the device: p_j ~ U(0.05,0.5), g_ij ~ Bin(2,p_j).
        The stream is keyed by the GLOBAL chunk of 8192 markers a marker belongs to, so a shard holds exactly the markers the
        whole problem holds at those positions: the same data, MM^T and selected marker at every number of GPUs."""
        torch = self.torch
        gen = torch.Generator(device=self.dev)
        g0 = (self.first // chunk) * chunk
        for c0 in range(g0, self.first + self.Lloc, chunk):
            gen.manual_seed(int(seed) * 1000003 + c0)
            p = torch.rand((chunk, 1), generator=gen, device=self.dev) * 0.45 + 0.05
            u1 = torch.rand((chunk, self.n), generator=gen, device=self.dev)
            u2 = torch.rand((chunk, self.n), generator=gen, device=self.dev)
            g = (u1 < p).to(torch.int8) + (u2 < p).to(torch.int8) - 1
            lo, hi = max(c0, self.first), min(c0 + chunk, self.first + self.Lloc)
            self.Mt8[lo - self.first:hi - self.first, : self.n] = g[lo - c0:hi - c0]
        self.M8 = None

    def fill_structured(self, K=2, fst=0.5, seed=5, pmin=0.05, chunk=16384):
        """Synthetic genotypes WITH population structure (Balding-Nichols): K equal sub-populations whose allele frequencies drift from a
        common ancestral frequency by Fst.  The operands of such a panel leave many markers whose quadratic form cancels against its
        diagonal term -- the case the certificate's two-tier threshold is for (tools/diag_structure.py, tests/test_gpu_structure.py)."""
        torch = self.torch
        n, L, dev = self.n, self.Lloc, self.dev
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        pop = torch.arange(n, device=dev) * K // n
        for r0 in range(0, L, chunk):
            r1 = min(L, r0 + chunk)
            p0 = pmin + (1 - 2 * pmin) * torch.rand(r1 - r0, 1, generator=gen, device=dev)
            a, b = p0 * (1 - fst) / fst, (1 - p0) * (1 - fst) / fst
            # Beta(a, b) per (marker, population) through two gammas (torch.distributions draws from the global generator: seeded here)
            torch.manual_seed(seed + 1000003 * (r0 // chunk + 1))
            ga = torch.distributions.Gamma(a.expand(-1, K), 1.0).sample()
            gb = torch.distributions.Gamma(b.expand(-1, K), 1.0).sample()
            p = (ga / (ga + gb)).clamp(0.001, 0.999)[:, pop]          # (markers, n)
            g = (torch.rand(p.shape, generator=gen, device=dev) < p).to(torch.int8) + (torch.rand(p.shape, generator=gen, device=dev) < p).to(torch.int8) - 1
            self.Mt8[r0:r1, :n] = g
        self.M8 = self.M4 = None
        self.Mt8s = None

    def load_Mt_ascii(self, path, max_mem_gb=8.0, threads=8):
        """Rows [first, first+Lloc) of Mt.ascii (the shard is a contiguous byte range of the file)."""
        self.Mt8.zero_()
        self.torch.cuda.synchronize(self.dev)
        self._check(self.L.eagle_dev_load_ascii(self.ctx, path.encode(), self.first, self.Lloc, 0, self.n,
                                                self.Mt8.data_ptr(), self.np_, float(max_mem_gb), int(threads)))
        self.M8 = None

    def load_M_ascii(self, path, max_mem_gb=8.0, threads=8):
        """Column window [first, first+Lloc) of every line of M.ascii."""
        torch = self.torch
        self.M8 = torch.zeros((self.np_, self.Lp), dtype=torch.int8, device=self.dev)
        torch.cuda.synchronize(self.dev)
        self._check(self.L.eagle_dev_load_ascii(self.ctx, path.encode(), 0, self.n, self.first, self.Lloc,
                                                self.M8.data_ptr(), self.Lp, float(max_mem_gb), int(threads)))

    def individual_major(self):
        if self.M8 is None:
            self.M8 = self.torch.empty((self.np_, self.Lp), dtype=self.torch.int8, device=self.dev)
            self._check(self.L.eagle_dev_transpose_i8(self.ctx, self.Mt8.data_ptr(), self.Lp, self.np_, self.np_,
                                                      self.M8.data_ptr(), self.Lp, self._stream()))
        return self.M8

    def individual_major_fp4(self):
        """M4: the individual-major genotypes as fp4 (two per byte), the operand image of the MM^T kernel."""
        if self.M4 is None:
            self.M4 = self.torch.empty((self.np_, self.Lp // 2), dtype=self.torch.uint8, device=self.dev)
            if self.M8 is not None:   # a column window of M.ascii was loaded: pack it
                self._check(self.L.eagle_dev_pack_fp4(self.ctx, self.M8.data_ptr(), self.np_, self.Lp, self.Lp, self.M4.data_ptr(),
                                                      self._stream()))
            else:                     # straight from the marker-major image, one pass, no individual-major int8 image
                self._check(self.L.eagle_dev_transpose_pack_fp4(self.ctx, self.Mt8.data_ptr(), self.Lp, self.np_, self.np_,
                                                                self.M4.data_ptr(), self.Lp // 2, self._stream()))
        return self.M4

    # ---- MM^T -----------------------------------------------------------------------------------
    def mmt_partial(self, out=None):
        """Exact int32 partial M_s M_s^T of this shard (upper-triangular 256-tiles live)."""
        torch = self.torch
        M4 = self.individual_major_fp4()
        c32 = out if out is not None else torch.empty((self.np_, self.np_), dtype=torch.int32, device=self.dev)
        c32.zero_()
        self._check(self.L.eagle_dev_mmt_accumulate_f4(self.ctx, M4.data_ptr(), self.np_, self.Lp, self.Lp // 2, c32.data_ptr(),
                                                       self._stream()))
        return c32

    def mmt_finish(self, c32, normalise=False):
        torch = self.torch
        out = torch.empty((self.n, self.n), dtype=torch.float64, device=self.dev)
        mx = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self._check(self.L.eagle_dev_mmt_finish(self.ctx, c32.data_ptr(), self.n, self.np_, out.data_ptr(), self.n,
                                                mx.data_ptr(), self._stream()))
        if normalise:  # calcMMt.R:13
            self._check(self.L.eagle_dev_mmt_normalise(self.ctx, out.data_ptr(), self.n, self.n, mx.data_ptr(),
                                                       self._stream()))
        return out, mx

    # ---- scan -----------------------------------------------------------------------------------
    def set_operands(self, S, V, ahat):
        """S = inv_MMt_sqrt, V = dim_reduced_vara (n x n, any torch/numpy layout, as MATRICES), a_hat (n)."""
        torch = self.torch
        n, np_ = self.n, self.np_

        def pad_t(Mx):  # row-major image of the transpose == the column-major R matrix, zero padded
            Mx = torch.as_tensor(Mx, dtype=torch.float64, device=self.dev)
            out = torch.zeros((np_, np_), dtype=torch.float64, device=self.dev)
            out[:n, :n] = Mx.t()
            return out
        self.Sa, self.Va = pad_t(S), pad_t(V)
        self.ahat = torch.zeros(np_, dtype=torch.float64, device=self.dev)
        self.ahat[:n] = torch.as_tensor(ahat, dtype=torch.float64, device=self.dev).reshape(-1)
        if self.v is None:
            self.v = torch.zeros(np_, dtype=torch.float64, device=self.dev)
            self.Wu = torch.zeros((np_, np_), dtype=torch.float64, device=self.dev)
            self.tmp = torch.zeros((np_, np_), dtype=torch.float64, device=self.dev)

    def set_W(self, W, v):
        """eagle_scan_with_W's operands: W = S V S (n x n, any layout, as a MATRIX) and v = S a_hat handed over ready-made
        (inside AM(): varG^2 P and varG P y); scan_operands() then only folds W instead of forming two n^3 products."""
        torch = self.torch
        n, np_ = self.n, self.np_
        self.W0 = torch.zeros((np_, np_), dtype=torch.float64, device=self.dev)
        self.W0[:n, :n] = torch.as_tensor(W, dtype=torch.float64, device=self.dev).t()
        if self.v is None:
            self.v = torch.zeros(np_, dtype=torch.float64, device=self.dev)
            self.Wu = torch.zeros((np_, np_), dtype=torch.float64, device=self.dev)
        self.v.zero_()
        self.v[:n] = torch.as_tensor(v, dtype=torch.float64, device=self.dev).reshape(-1)
        self.v0 = self.v.clone()

    def release_operands(self):
        """Drop the n x n operand images and the vara workspace (a and vara of the last scan stay)."""
        self.Sa = self.Va = self.ahat = self.v = self.Wu = self.tmp = self.W0 = self.v0 = None
        self.ws = None
        self.torch.cuda.empty_cache()

    def scan_operands(self, coll=None):
        """v = S a_hat and Wu = fold(S V S).  With a Collectives of world > 1 whose size divides the 128-row tiles of W, the
        n^3 work is shared: every rank computes its row block of the W^T image, one all-gather completes it (the
        replicated computation is the fallback)."""
        if getattr(self, "W0", None) is not None:  # W and v were handed over (set_W): copy + fold, no product
            self.Wu.copy_(self.W0)
            self.v.copy_(self.v0)
            self._check(self.L.eagle_dev_fold_upper(self.ctx, self.Wu.data_ptr(), self.np_, self._stream()))
            return
        world = coll.world if coll is not None else 1
        nt = self.np_ // 128
        if coll is not None and coll.active and nt % world == 0 and self.share_w:
            rows = self.np_ // world
            r0 = coll.rank * rows
            self._check(self.L.eagle_dev_scan_operands_rows(self.ctx, self.Sa.data_ptr(), self.Va.data_ptr(), self.ahat.data_ptr(),
                                                            self.n, self.np_, r0, r0 + rows, self.v.data_ptr(), self.Wu.data_ptr(),
                                                            self.tmp.data_ptr(), self._stream()))
            mine = self.Wu[r0:r0 + rows].clone()
            coll.all_gather_rows(self.Wu, mine)
            self._check(self.L.eagle_dev_fold_upper(self.ctx, self.Wu.data_ptr(), self.np_, self._stream()))
            return
        # W from int8 digit slices only feeds the digit-slice scan, whose certificate carries its error bound (csrc/eagle_w8.hip)
        self._check(self.L.eagle_set_w_mode(self.ctx, int(self.w_mode) if self.mode == 1 else 0))
        self._check(self.L.eagle_dev_scan_operands(self.ctx, self.Sa.data_ptr(), self.Va.data_ptr(), self.ahat.data_ptr(),
                                                   self.n, self.np_, self.v.data_ptr(), self.Wu.data_ptr(),
                                                   self.tmp.data_ptr(), self._stream()))

    def w_info(self):
        """eagle_last_w_info as a dict: which engine formed the last W, its configuration and error bound."""
        import ctypes as C

        class Info(C.Structure):
            _fields_ = [("int8", C.c_int), ("declined", C.c_int), ("k1", C.c_int), ("T1", C.c_int), ("pairs1", C.c_int), ("k2", C.c_int),
                        ("T2", C.c_int), ("pairs2", C.c_int), ("eta", C.c_double), ("eta_x", C.c_double), ("target", C.c_double),
                        ("mean_diag", C.c_double), ("asym_term", C.c_double), ("pipelined", C.c_int), ("pad", C.c_int)]
        i = Info()
        self._check(self.L.eagle_last_w_info(self.ctx, C.byref(i)))
        return {k: getattr(i, k) for k, _ in Info._fields_}

    def gemv_a(self):
        self._check(self.L.eagle_dev_gemv_i8(self.ctx, self.Mt8.data_ptr(), self.Lp, self.np_, self.np_, self.v.data_ptr(),
                                             1.0, self.a.data_ptr(), self._stream()))

    def _ns(self):
        """The `nslices` argument of the digit-slice entry points: count | EAGLE_SLICES_STOCHASTIC."""
        return int(self.nslices) | (0x100 if (self.stochastic and self.mode == 1) else 0)

    def _ws(self):
        if self.ws is None or self._ws_mode != self.mode:
            fn = self.L.eagle_vara_f6_workspace_bytes if self.mode == 2 else self.L.eagle_vara_i8_workspace_bytes
            nb = int(fn(self.np_, self.Lp, self.nslices))
            self.ws = None
            self.ws = self.torch.empty(nb, dtype=self.torch.uint8, device=self.dev)
            self._ws_mode = self.mode
        return self.ws

    def shifted_image(self):
        """(Mt8s, cshift): the markers re-centred on their majority genotype, made once per shard for the digit-slice kernel."""
        if self.Mt8s is None:
            self.Mt8s = self.torch.empty_like(self.Mt8)
            self.cshift = self.torch.empty(self.Lp, dtype=self.torch.int8, device=self.dev)
            self.l1 = self.torch.empty((self.Lp, 2), dtype=self.torch.int32, device=self.dev)  # {sum |m'|, sum m'^2} per marker
            self._check(self.L.eagle_dev_marker_shift(self.ctx, self.Mt8.data_ptr(), self.Lp, self.n, self.np_, self.np_,
                                                      self.Mt8s.data_ptr(), self.cshift.data_ptr(), self.l1.data_ptr(), self._stream()))
        return self.Mt8s, self.cshift

    def fp4_image(self):
        """Mt4: the genotypes as fp4 (two per byte), made once per shard for the fp4 x fp6 vara kernel (mode 2)."""
        if self.Mt4 is None:
            self.Mt4 = self.torch.empty((self.Lp, self.np_ // 2), dtype=self.torch.uint8, device=self.dev)
            self._check(self.L.eagle_dev_pack_fp4(self.ctx, self.Mt8.data_ptr(), self.Lp, self.np_, self.np_, self.Mt4.data_ptr(),
                                                  self._stream()))
        return self.Mt4

    def vara_prepare(self, with_a=True):
        """int8 path, phase 1: slice W, and ONE pass over the genotypes for a = Mt v and the diagonal term of vara."""
        ws = self._ws()
        prep = self.L.eagle_dev_vara_f6_prepare if self.mode == 2 else self.L.eagle_dev_vara_i8_prepare
        self._check(prep(self.ctx, self.Mt8.data_ptr(), self.Lp, self.np_, self.np_,
                    self.Wu.data_ptr(), self._ns(), ws.data_ptr(), self.v.data_ptr() if with_a else None,
                    self.a.data_ptr() if with_a else None, self._stream()))

    def vara_kernel(self):
        """The dominant kernel alone: fp64 MFMA vara kernel (mode 0) or the int8 MFMA kernel + finish (mode 1)."""
        if self.mode == 0:
            self._check(self.L.eagle_dev_vara_f64(self.ctx, self.Mt8.data_ptr(), self.Lp, self.np_, self.np_,
                                                  self.Wu.data_ptr(), self.vara.data_ptr(), self._stream()))
        elif self.mode == 2:
            self._check(self.L.eagle_dev_vara_f6_mfma(self.ctx, self.Mt8.data_ptr(), self.fp4_image().data_ptr(), self.Lp, self.np_,
                                                      self.np_, self.nslices, self._ws().data_ptr(), self.vara.data_ptr(), None,
                                                      self._stream()))
        else:
            Ms, cs = self.shifted_image()
            self._check(self.L.eagle_dev_vara_i8_mfma_shifted(self.ctx, Ms.data_ptr(), cs.data_ptr(), self.Lp, self.np_, self.np_,
                                                              self._ns(), self._ws().data_ptr(), self.vara.data_ptr(), None,
                                                              self._stream()))
            if self.extend:   # markers that fail their budget under the spectral bound get the dropped digit back (dropped on the device otherwise)
                self._check(self.L.eagle_dev_vara_i8_extend(self.ctx, Ms.data_ptr(), cs.data_ptr(), self.l1.data_ptr(), self.Lloc, self.Lp,
                                                            self.np_, self.np_, self._ns(), self._ws().data_ptr(), self.vara.data_ptr(),
                                                            self._stream()))

    def certify(self):
        """Digit-slice mode: re-evaluate in fp64 every marker whose error bound exceeds 1.8 x budget (0.9e-6) of |vara| and every marker the
        bounds cannot exclude from being the arg-max, so that the arg-max below is the fp64 scan's (find_qtl.R:71-83)."""
        if self.mode != 1:
            return
        self.shifted_image()
        if self.cert_ws is None:
            nb = int(self.L.eagle_scan_certify_workspace_bytes(self.np_))
            self.cert_ws = self.torch.empty(nb, dtype=self.torch.uint8, device=self.dev)
        self._check(self.L.eagle_dev_scan_certify(self.ctx, self.Mt8.data_ptr(), self.Lloc, self.Lp, self.np_, self.np_,
                                                  self.cshift.data_ptr(), self.l1.data_ptr(), self._ns(), self._ws().data_ptr(),
                                                  self.Wu.data_ptr(), self.a.data_ptr(), self.vara.data_ptr(), self.cert_ws.data_ptr(),
                                                  self._stream()))

    def certificate(self):
        """{lower_bound, reevaluated, overflow, flagged, over_tight} of the last certify() (synchronises).  over_tight: markers over 1.8 x the
        budget in force; more than 512 of them under the tight budget and `flagged` is counted against 1.8 x the default one."""
        h = self.cert_ws[:24].cpu().numpy().tobytes()
        i = np.frombuffer(h[8:24], dtype=np.int32)
        return {"lower_bound": float(np.frombuffer(h[0:8], dtype=np.float64)[0]), "reevaluated": int(min(i[0], 2048)),
                "overflow": int(i[1]), "flagged": int(i[2]), "over_tight": int(i[3])}

    def vara_i8_info(self):
        """(slices used, absolute error bound, max |off-diagonal W|) of the last int8-slice vara launch (synchronises)."""
        h = self.ws[:120].cpu().numpy().tobytes()
        self.last_budget_loose = float(np.frombuffer(h[112:120], dtype=np.float64)[0])   # the default behind a tight budget in force
        self.last_budget = float(np.frombuffer(h[56:64], dtype=np.float64)[0])   # the budget in force (tight one first: eagle_last_scan_budget)
        self.last_wErr = float(np.frombuffer(h[96:104], dtype=np.float64)[0])    # || W - S V S ||_F bound of a W from the int8 engine (0: fp64 products)
        self.last_level = int(np.frombuffer(h[92:96], dtype=np.int32)[0])   # which level of the spectral bound took the digit off (0: none)
        self.last_e = int(np.frombuffer(h[64:68], dtype=np.int32)[0])   # scale exponent of the digits: unit of digit s = 2^(e + 2 - 8 (s + 1))
        self.last_sumdiag = float(np.frombuffer(h[24:32], dtype=np.float64)[0])  # sum_k |W_kk|
        # round 3: the spectral bound of the last digit (0: not in use) and the digits that were cut (= used, or one more)
        self.last_specH = float(np.frombuffer(h[40:48], dtype=np.float64)[0])
        self.last_sliced = int(np.frombuffer(h[48:52], dtype=np.int32)[0])
        return (int(np.frombuffer(h[8:12], dtype=np.int32)[0]), float(np.frombuffer(h[16:24], dtype=np.float64)[0]),
                float(np.frombuffer(h[0:8], dtype=np.float64)[0]))

    def argmax(self):
        self._check(self.L.eagle_dev_tsq_argmax(self.ctx, self.a.data_ptr(), self.vara.data_ptr(), self.Lloc, None,
                                                self._best.data_ptr(), self._scratch.data_ptr(), self._stream()))

    def scan(self, coll=None):
        """calculate_a_and_vara_rcpp.cpp:90-112 + find_qtl.R:71-83 on this shard, all on the current stream."""
        self.scan_operands(coll)
        if self.mode == 0:
            self.gemv_a()
        else:
            self.vara_prepare()
        self.vara_kernel()
        if self.certified:
            self.certify()
        self.argmax()

    def best(self):
        """(tsqmax, GLOBAL 0-based index or -1, near ties) of the last scan (synchronises)."""
        if self.mode == 1 and self.certified and self.cert_ws is not None and self.ws is not None:
            # the same copy brings the certificate's overflow flag and the spectral bound: a scan that took a digit off under the
            # spectral bound and then had to redo the block in fp64 makes this context keep the worst-case digit count (as
            # eagle_calculate_a_and_vara does for the reference-shaped call; eagle_set_scan_budget re-arms)
            u8 = self.torch.uint8
            raw = self.torch.cat([self._best.view(u8), self.cert_ws[:24], self.ws[40:48]]).cpu().numpy()
            b = np.frombuffer(raw[:24].tobytes(), dtype=np.int64)
            overflow = int(np.frombuffer(raw[24:48].tobytes()[12:16], dtype=np.int32)[0])
            spec = float(np.frombuffer(raw[48:56].tobytes(), dtype=np.float64)[0])
            if overflow and spec > 0.0:
                self.L.eagle_dev_set_spectral(self.ctx, 0)
        else:
            b = self._best.cpu().numpy()
        tsqmax = float(np.frombuffer(b[:1].tobytes(), dtype=np.float64)[0])
        idx0 = int(b[1])
        return tsqmax, (idx0 + self.first if idx0 >= 0 else -1), int(b[2])
