"""Operator-level parity of the HIP kernels (through the C ABI) against plain fp32/fp64 PyTorch
on the CPU.  fp32 kernels: 2e-5 relative to the output scale (fp32 MFMA is an exact fmaf chain);
bf16 kernels: inputs are rounded to bf16 first, the reference is computed from those rounded
values in fp64, and the tolerance is bf16's output rounding (2^-8 relative) plus accumulation."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32_TOL = 2e-5
BF16_TOL = 1.2e-2


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from bist_amd import _lib, ops as o
    assert _lib.lib.bist_device_ok() == 1, _lib.lib.bist_last_error()
    return o


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _cmp(got, ref, tol, what):
    got = got.detach().float().cpu().double()
    ref = ref.double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} (scale {scale:.3e}, tol {tol})"


def _q(x, dtype):
    """value as the kernel sees it (rounded to the storage dtype), in fp64 on the CPU"""
    return x.to(dtype).double()


CASES = [  # (M, N, K) -- K multiples of 8 take the LDS-DMA kernels for both dtypes
    (128, 128, 64), (300, 200, 128), (77, 513, 192), (1, 5, 64), (257, 129, 512), (320, 192, 1056), (1300, 1200, 96),
    (320, 192, 3000), (130, 70, 200), (64, 64, 8),     # K tails: the stager zero-fills the trailing partial tile
    (320, 512, 1536), (320, 512, 2048), (100, 130, 1000),      # long K on few 64-tiles (ring kernel)
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("M,N,K", CASES)
def test_gemm_fast_linear(ops, dtype, tol, M, N, K):
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    r = _rand(M, N, seed=4)
    xd, wd, bd, rd = (t.to(dtype).cuda() for t in (x, w, b, r))
    from bist_amd import _lib
    g = ops.gemm_desc(xd, wd, torch.empty(M, N, device="cuda", dtype=dtype), M=M, N=N, K=K, a_rs=K, b_rs=K, ldc=N)
    assert _lib.lib.bist_gemm_is_fast(g) >= 1          # 1 = LDS-DMA kernel, 2 = the same with split-K
    ref = _q(x, dtype) @ _q(w, dtype).t() + _q(b, dtype)
    _cmp(ops.linear(xd, wd, bd), ref, tol, "linear")
    _cmp(ops.linear(xd, wd, bd, act=ops.ACT_RELU), ref.clamp_min(0), tol, "linear+relu")
    _cmp(ops.linear(xd, wd, bd, residual=rd, alpha=0.5),
         0.5 * (_q(x, dtype) @ _q(w, dtype).t()) + _q(b, dtype) + _q(r, dtype), tol, "linear+residual")
    _cmp(ops.linear(xd, wd, None, out_dtype=torch.float32), _q(x, dtype) @ _q(w, dtype).t(),
         F32_TOL if dtype == torch.float32 else 2e-3, "linear f32 out")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("rows,n_out,k_in", [(320, 512, 512), (320, 1536, 512), (400, 1024, 512), (72, 64, 200), (25088, 512, 512)])
def test_gemm_pair_linear_backward(ops, dtype, tol, rows, n_out, k_in):
    """bist_gemm_pair: dX = dZ.W and dW += dZ^T.X of one linear layer in ONE launch when both are small
    (the last shape is too big for that and must fall back to two launches with the same results)."""
    dz, w, x = _rand(rows, n_out, seed=70, scale=n_out ** -0.5), _rand(n_out, k_in, seed=71), _rand(rows, k_in, seed=72, scale=rows ** -0.5)
    dzd, wd, xd = (t.to(dtype).cuda() for t in (dz, w, x))
    dx = torch.empty(rows, k_in, device="cuda", dtype=dtype)
    dw0 = _rand(n_out, k_in, seed=73)
    dw = dw0.cuda()                                                    # fp32 accumulator, like the trainer's acc32 views
    ga = ops.gemm_desc(dzd, wd, dx, M=rows, N=k_in, K=n_out, a_rs=n_out, a_ks=1, b_rs=1, b_ks=k_in, ldc=k_in, alpha=0.5)
    gb = ops.gemm_desc(dzd, xd, dw, M=n_out, N=k_in, K=rows, a_rs=1, a_ks=n_out, b_rs=1, b_ks=k_in, ldc=k_in, residual=dw, ldr=k_in)
    ops.gemm_pair(ga, gb)
    _cmp(dx, 0.5 * (_q(dz, dtype) @ _q(w, dtype)), tol, "pair dX")
    _cmp(dw, dw0 + _q(dz, dtype).t() @ _q(x, dtype), tol * 2, "pair dW (accumulated)")
    dwb = torch.empty(n_out, k_in, device="cuda", dtype=dtype)
    gb2 = ops.gemm_desc(dzd, xd, dwb, M=n_out, N=k_in, K=rows, a_rs=1, a_ks=n_out, b_rs=1, b_ks=k_in, ldc=k_in)
    ops.gemm_pair(ga, gb2)
    _cmp(dwb, _q(dz, dtype).t() @ _q(x, dtype), tol * 2, "pair dW (operand dtype)")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("M,N,K,lay", [(320, 512, 1536, "NT"), (320, 512, 2048, "NN"), (100, 130, 1000, "NT"), (192, 264, 832, "TT")])
def test_gemm_in_launch_split_k(ops, dtype, tol, M, N, K, lay):
    """hint BIST_GEMM_SPLIT64: K cut over neighbouring workgroups of a 64-tile product, combined inside the launch by the
    last slice to finish (release / ticket / acquire); repeated launches reuse the self-resetting ticket counters."""
    import ctypes
    from bist_amd import _lib
    a, b = _rand(M, K, seed=90), _rand(N, K, seed=91, scale=K ** -0.5)
    ref = _q(a, dtype) @ _q(b, dtype).t()
    ad, bd = a.to(dtype).cuda(), b.to(dtype).cuda()
    at, bt = a.t().contiguous().to(dtype).cuda(), b.t().contiguous().to(dtype).cuda()
    A, ars, aks = (ad, K, 1) if lay[0] == "N" else (at, 1, M)
    Bm, brs, bks = (bd, K, 1) if lay[1] == "T" and lay != "TT" else ((bt, 1, N) if lay in ("NN", "TT") else (bd, K, 1))
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    g = ops.gemm_desc(A, Bm, out, M=M, N=N, K=K, a_rs=ars, a_ks=aks, b_rs=brs, b_ks=bks, ldc=N)
    g.hint = 4
    assert _lib.lib.bist_gemm_is_fast(g) == 2
    for rep in range(3):
        out.zero_()
        _lib.check(_lib.lib.bist_gemm(ctypes.byref(g), torch.cuda.current_stream().cuda_stream), "bist_gemm")
        _cmp(out, ref, tol, f"in-launch split-K {lay} run {rep}")


@pytest.mark.parametrize("M,N,K", [(1024, 512, 256), (1100, 300, 192), (256, 256, 128), (700, 1024, 832), (160, 1568, 512), (470, 300, 128)])
def test_gemm_tile256_kernel(ops, M, N, K):
    """the 256x256-tile deep-pipelined kernel (hint BIST_GEMM_TILE256; automatic only for far larger products than
    this model has): ragged row / column tiles, bias + ReLU, residual with accumulate, fp32 output."""
    dtype, tol = torch.bfloat16, BF16_TOL
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    r = _rand(M, N, seed=4)
    xd, wd, bd, rd = (t.to(dtype).cuda() for t in (x, w, b, r))
    ref = _q(x, dtype) @ _q(w, dtype).t()

    def run(out, **kw):
        g = ops.gemm_desc(xd, wd, out, M=M, N=N, K=K, a_rs=K, b_rs=K, ldc=N, **kw)
        g.hint = 2
        from bist_amd import _lib
        import ctypes
        _lib.check(_lib.lib.bist_gemm(ctypes.byref(g), torch.cuda.current_stream().cuda_stream), "bist_gemm")
        return out
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    _cmp(run(out, bias=bd), ref + _q(b, dtype), tol, "tile256 linear")
    _cmp(run(out, bias=bd, act=ops.ACT_RELU), (ref + _q(b, dtype)).clamp_min(0), tol, "tile256 relu")
    _cmp(run(out, bias=bd, residual=rd, ldr=N, alpha=0.5), 0.5 * ref + _q(b, dtype) + _q(r, dtype), tol, "tile256 residual")
    o32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    _cmp(run(o32), ref, 2e-3, "tile256 f32 out")


def test_gemm_tile256_is_automatic_for_the_frame_grid_products(ops):
    """K-contiguous bf16 products of 140+ tiles of 256x256 (the M = B*T*S products of the path at B >= 12) take the
    256-tile kernel on their own; ragged last row tile, bias + ReLU + residual, and the dropout mask of its epilogue is
    the one the generic kernels draw (same seed, same element index)."""
    from bist_amd import _lib
    M, N, K = 18100, 512, 512
    dtype, tol = torch.bfloat16, BF16_TOL
    x, w, b = _rand(M, K, seed=11), _rand(N, K, seed=12, scale=K ** -0.5), _rand(N, seed=13)
    r = _rand(M, N, seed=14)
    xd, wd, bd, rd = (t.to(dtype).cuda() for t in (x, w, b, r))
    ref = _q(x, dtype) @ _q(w, dtype).t() + _q(b, dtype)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    kw = dict(M=M, N=N, K=K, a_rs=K, b_rs=K, ldc=N)
    assert _lib.lib.bist_gemm_is_fast(ops.gemm_desc(xd, wd, out, **kw)) == 4
    _cmp(ops.gemm(xd, wd, out, bias=bd, act=ops.ACT_RELU, residual=rd, ldr=N, **kw), ref.clamp_min(0) + _q(r, dtype), tol, "auto tile256")
    # dropout: kept elements are scaled by 1/(1-p), the rest are exactly the residual; the same call on a row slice that
    # takes the 128-tile kernel draws the same mask for the same elements
    p = 0.25
    full = ops.gemm(xd, wd, out, bias=bd, drop_p=p, drop_seed=77, **kw).float().cpu()
    kept = full != 0
    assert abs(kept.float().mean().item() - (1 - p)) < 5e-3
    _cmp(torch.where(kept, full * (1 - p), torch.zeros(())), torch.where(kept, ref.float(), torch.zeros(())), tol, "tile256 dropout")
    Ms = 1024
    small = torch.empty(Ms, N, device="cuda", dtype=dtype)
    kws = dict(kw, M=Ms)
    assert _lib.lib.bist_gemm_is_fast(ops.gemm_desc(xd[:Ms], wd, small, **kws)) == 1
    ops.gemm(xd[:Ms], wd, small, bias=bd, drop_p=p, drop_seed=77, **kws)
    assert torch.equal((small.float().cpu() != 0), kept[:Ms])


@pytest.mark.parametrize("M,N,K", [(320, 2048, 512), (77, 130, 96), (18100, 512, 512)])
def test_gemm_gate_epilogue(ops, M, N, K):
    """BIST_ACT_GATE: y = gate > 0 ? alpha * x.W^T : 0 -- the backward of dropout(relu(z)) fused into the product that
    computes its output gradient (64-tile, generic and 256-tile kernels)."""
    from bist_amd._lib import ACT_GATE
    dtype, tol = torch.bfloat16, BF16_TOL
    x, w = _rand(M, K, seed=31), _rand(N, K, seed=32, scale=K ** -0.5)
    gate = _rand(M, N, seed=33).clamp_min(0)                   # ~half zeros, like a ReLU output
    xd, wd, gd = (t.to(dtype).cuda() for t in (x, w, gate))
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(xd, wd, out, M=M, N=N, K=K, a_rs=K, b_rs=K, ldc=N, alpha=1.25, act=ACT_GATE, residual=gd, ldr=N)
    ref = torch.where(_q(gate, dtype) > 0, 1.25 * (_q(x, dtype) @ _q(w, dtype).t()), torch.zeros(()))
    _cmp(out, ref, tol, "gate epilogue")
    assert torch.equal(out.float().cpu() == 0, ref == 0) or (out.float().cpu()[ref == 0] == 0).all()


def test_gemm_tile256_race_screen(ops):
    """The 256-tile kernel orders its LDS-DMA stagings against the fragment reads by counted vmcnt waits and barriers only
    (no data dependence the hardware could see): a misplaced wait shows as a rare wrong tile that comes and goes with
    the memory system's timing.  Screen: the same product 40 times over operand sets larger than the 256 MiB MALL (so
    the A stream arrives from HBM with varying latency), at 2, 3, 8 and 32 K tiles -- every result bit-identical to the
    first, and the first equal to the 128-tile kernel's (different schedule, same accumulation order per output)."""
    import ctypes
    from bist_amd import _lib
    dtype = torch.bfloat16
    for (M, N, K) in [(25088, 512, 2048), (25088, 512, 512), (18100, 768, 192), (4096, 1024, 128)]:
        nset = max(2, int(400e6 // (2 * M * K)))
        g = torch.Generator(device="cuda").manual_seed(M + K)
        A = [(torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(dtype) for _ in range(nset)]
        w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).to(dtype)
        kw = dict(M=M, N=N, K=K, a_rs=K, b_rs=K, ldc=N)

        def run(a, hint):
            out = torch.empty(M, N, device="cuda", dtype=dtype)
            d = ops.gemm_desc(a, w, out, **kw)
            d.hint = hint
            _lib.check(_lib.lib.bist_gemm(ctypes.byref(d), torch.cuda.current_stream().cuda_stream), "bist_gemm")
            return out
        first = [run(a, 2) for a in A]
        for rep in range(40):
            i = rep % nset
            assert torch.equal(run(A[i], 2), first[i]), f"{(M, N, K)}: run {rep} differs from the first"
        # against the 128-tile kernel on a row slice small enough to take it
        Ms = 1024
        small = torch.empty(Ms, N, device="cuda", dtype=dtype)
        d = ops.gemm_desc(A[0][:Ms], w, small, **dict(kw, M=Ms))
        assert _lib.lib.bist_gemm_is_fast(d) == 1
        _lib.check(_lib.lib.bist_gemm(ctypes.byref(d), torch.cuda.current_stream().cuda_stream), "bist_gemm")
        assert (small.float() - first[0][:Ms].float()).abs().max().item() <= 2e-2 * max(1.0, first[0].float().abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
def test_gemm_generic_strides(ops, dtype, tol):
    # K tail (50), unaligned leading dims, transposed operands: all take the generic kernel
    M, N, K = 150, 70, 50
    a, b = _rand(M, K, seed=5), _rand(N, K, seed=6)
    ref = _q(a, dtype) @ _q(b, dtype).t()
    ad, bd = a.to(dtype).cuda(), b.to(dtype).cuda()
    _cmp(ops.linear(ad, bd), ref, tol, "K tail")
    at = a.t().contiguous().to(dtype).cuda()      # A stored [K, M]: a_rs = 1, a_ks = M
    bt = b.t().contiguous().to(dtype).cuda()      # B stored [K, N]
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(at, bd, out, M=M, N=N, K=K, a_rs=1, a_ks=M, b_rs=K, b_ks=1, ldc=N)
    _cmp(out, ref, tol, "A transposed (TN)")
    ops.gemm(ad, bt, out, M=M, N=N, K=K, a_rs=K, a_ks=1, b_rs=1, b_ks=N, ldc=N)
    _cmp(out, ref, tol, "B transposed (NN)")
    ops.gemm(at, bt, out, M=M, N=N, K=K, a_rs=1, a_ks=M, b_rs=1, b_ks=N, ldc=N)
    _cmp(out, ref, tol, "both transposed")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("M,N,K", [(256, 192, 128), (136, 520, 320), (128, 128, 64), (192, 264, 832), (1288, 1408, 160),
                                   (200, 136, 256), (136, 264, 512), (72, 200, 160), (1024, 512, 400), (72, 136, 40)])
def test_gemm_fast_transposed_operands(ops, dtype, tol, M, N, K):
    """row-contiguous operands (the backward products) take the LDS-DMA kernel with the transposing
    fragment reads (ds_read_b64_tr_b16 for bf16, ds_read_b32 for f32) -- no generic fallback."""
    from bist_amd import _lib
    a, b = _rand(M, K, seed=50), _rand(N, K, seed=51, scale=K ** -0.5)
    ref = _q(a, dtype) @ _q(b, dtype).t()
    ad, bd = a.to(dtype).cuda(), b.to(dtype).cuda()
    at, bt = a.t().contiguous().to(dtype).cuda(), b.t().contiguous().to(dtype).cuda()      # stored [K, M] / [K, N]
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    for name, (A, ars, aks), (Bm, brs, bks) in [("TN", (at, 1, M), (bd, K, 1)), ("NT'", (ad, K, 1), (bt, 1, N)),
                                               ("TT", (at, 1, M), (bt, 1, N))]:
        g = ops.gemm_desc(A, Bm, out, M=M, N=N, K=K, a_rs=ars, a_ks=aks, b_rs=brs, b_ks=bks, ldc=N)
        assert _lib.lib.bist_gemm_is_fast(g) >= 1, name
        out.zero_()
        ops.gemm(A, Bm, out, M=M, N=N, K=K, a_rs=ars, a_ks=aks, b_rs=brs, b_ks=bks, ldc=N)
        _cmp(out, ref, tol, f"transposed {name}")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, BF16_TOL)])
def test_gemm_split_k_weight_gradient_shape(ops, dtype, tol):
    """dW = dY^T X with a long reduction (K = rows): few output tiles -> split-K over the workspace."""
    from bist_amd import _lib
    rows, N, Kd = 6400, 128, 192
    dy, x = _rand(rows, N, seed=52, scale=rows ** -0.5), _rand(rows, Kd, seed=53)
    ref = _q(dy, dtype).t() @ _q(x, dtype)
    dyd, xd = dy.to(dtype).cuda(), x.to(dtype).cuda()
    out = torch.empty(N, Kd, device="cuda", dtype=dtype)
    kw = dict(M=N, N=Kd, K=rows, a_rs=1, a_ks=N, b_rs=1, b_ks=Kd, ldc=Kd, alpha=0.5)
    assert _lib.lib.bist_gemm_is_fast(ops.gemm_desc(dyd, xd, out, **kw)) == 2
    ops.gemm(dyd, xd, out, **kw)
    _cmp(out, 0.5 * ref, tol, "split-K dW")
    bias = _rand(Kd, seed=54)
    ops.gemm(dyd, xd, out, bias=bias.to(dtype).cuda(), act=ops.ACT_RELU, **kw)
    _cmp(out, (0.5 * ref + _q(bias, dtype)).clamp_min(0), tol, "split-K + epilogue")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("rows,d,n", [(320, 512, 3), (77, 100, 1), (1280, 512, 4), (5, 512, 8)])
def test_gemm_skinny_products(ops, dtype, tol, rows, d, n):
    """the fusion / switch logits (N <= 8), their input gradient (K <= 8) and weight gradient (M <= 8,
    K = rows) skip the 128x128 MFMA tile (decoder.py:153, generator.py:71/117)."""
    from bist_amd import _lib
    x, w, bias = _rand(rows, d, seed=60), _rand(n, d, seed=61, scale=d ** -0.5), _rand(n, seed=62)
    dl = _rand(rows, n, seed=63)
    xd, wd, bd, dld = x.to(dtype).cuda(), w.to(dtype).cuda(), bias.to(dtype).cuda(), dl.to(dtype).cuda()
    xq, wq, dlq = _q(x, dtype), _q(w, dtype), _q(dl, dtype)
    # logits = x W^T + b, accumulated on a previous partial sum
    prev = _rand(rows, n, seed=64)
    out = torch.empty(rows, n, device="cuda", dtype=torch.float32)
    kw = dict(M=rows, N=n, K=d, a_rs=d, a_ks=1, b_rs=d, b_ks=1, ldc=n)
    assert _lib.lib.bist_gemm_is_fast(ops.gemm_desc(xd, wd, out, **kw)) == 3
    ops.gemm(xd, wd, out, bias=bd, residual=prev.cuda(), ldr=n, **kw)
    _cmp(out, xq @ wq.t() + _q(bias, dtype) + prev, tol, "skinny N")
    # column slice of a wider weight (the concat-free fusion): row stride 3d
    wide = _rand(n, 3 * d, seed=65, scale=d ** -0.5)
    wided = wide.to(dtype).cuda()
    ops.gemm(xd, wided[:, d:2 * d], out, M=rows, N=n, K=d, a_rs=d, a_ks=1, b_rs=3 * d, b_ks=1, ldc=n)
    _cmp(out, xq @ _q(wide, dtype)[:, d:2 * d].t(), tol, "skinny N, sliced weight")
    # dx = dl W  (K = n)
    dx = torch.empty(rows, d, device="cuda", dtype=dtype)
    ops.gemm(dld, wd, dx, M=rows, N=d, K=n, a_rs=n, a_ks=1, b_rs=1, b_ks=d, ldc=d, alpha=0.5)
    _cmp(dx, 0.5 * (dlq @ wq), tol, "skinny K")
    # dW = dl^T x  (M = n, K = rows), accumulated in place in fp32
    acc = _rand(n, d, seed=66).cuda()
    want = acc.cpu() + dlq.t() @ xq
    ops.gemm(dld, xd, acc, M=n, N=d, K=rows, a_rs=1, a_ks=n, b_rs=1, b_ks=d, ldc=d, residual=acc, ldr=d)
    _cmp(acc, want, tol * max(1.0, rows ** 0.5 / 4), "skinny M")
    # the mirrored orientation: C[d, n] = x^T dl
    outT = torch.empty(d, n, device="cuda", dtype=dtype)
    ops.gemm(xd, dld, outT, M=d, N=n, K=rows, a_rs=1, a_ks=d, b_rs=1, b_ks=n, ldc=n)
    _cmp(outT, xq.t() @ dlq, tol * max(1.0, rows ** 0.5 / 4), "skinny N, transposed A")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
def test_gemm_batched_head_fold(ops, dtype, tol):
    # the K-fold: Qf[(b,i), hh*d + n] = sum_c Q[(b,i), hh*dk + c] * Wk[hh*dk + c, n]   (batch over heads)
    Bq, h, dk = 37, 4, 16
    d = h * dk
    q, wk = _rand(Bq, d, seed=7), _rand(d, d, seed=8, scale=d ** -0.5)
    qd, wd = q.to(dtype).cuda(), wk.to(dtype).cuda()
    out = torch.empty(Bq, h * d, device="cuda", dtype=dtype)
    ops.gemm(qd, wd, out, M=Bq, N=d, K=dk, a_rs=d, a_ks=1, b_rs=1, b_ks=d, ldc=h * d, batch=(1, h),
             a_bs=(0, dk), b_bs=(0, dk * d), c_bs=(0, d), alpha=0.25)
    ref = torch.einsum("mhc,hcn->mhn", _q(q, dtype).view(Bq, h, dk), _q(wk, dtype).view(h, dk, d)).reshape(Bq, h * d) * 0.25
    _cmp(out, ref, tol, "head fold")


def test_gemm_residual_row_map_and_accumulate(ops):
    B, G, Lq, d = 3, 5, 7, 64
    o, w, bias, x = _rand(B * G * Lq, d, seed=9), _rand(d, d, seed=10, scale=d ** -0.5), _rand(d, seed=11), _rand(B * Lq, d, seed=12)
    y = ops.linear(o.cuda(), w.cuda(), bias.cuda(), residual=x.cuda(), res_map=(G * Lq, Lq))
    ref = (o.double() @ w.double().t() + bias.double()).view(B, G, Lq, d) + x.double().view(B, 1, Lq, d)
    _cmp(y, ref.reshape(-1, d), F32_TOL, "expanded-query residual")
    acc = ops.linear(o.cuda(), w.cuda(), bias.cuda())
    ops.linear(o.cuda(), w.cuda(), None, out=acc, accumulate=True)
    _cmp(acc, 2 * (o.double() @ w.double().t()) + bias.double(), F32_TOL, "accumulate")


def test_gemm_dropout_mask_statistics(ops):
    M = N = 256
    x, w = torch.ones(M, 64), torch.zeros(N, 64)
    w[:, 0] = 1.0
    y = ops.linear(x.cuda(), w.cuda(), None, drop_p=0.25, drop_seed=1234).cpu()
    kept = (y != 0)
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    assert torch.allclose(y[kept], torch.full_like(y[kept], 1 / 0.75), atol=1e-6)
    y2 = ops.linear(x.cuda(), w.cuda(), None, drop_p=0.25, drop_seed=1234).cpu()
    assert torch.equal(y, y2)                                   # same seed -> same mask
    y3 = ops.linear(x.cuda(), w.cuda(), None, drop_p=0.25, drop_seed=99).cpu()
    assert not torch.equal(y, y3)


def test_gemm_rejects_bad_arguments(ops):
    from bist_amd._lib import BistError
    x = torch.zeros(4, 64, device="cuda")
    with pytest.raises(BistError):
        ops.gemm(x, x, x, M=0, N=4, K=64, a_rs=64, b_rs=64, ldc=4)
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 64), torch.zeros(4, 64))       # CPU tensors: no fallback


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("d", [64, 512, 100])
def test_layernorm(ops, dtype, tol, d):
    x, a, b = _rand(37, d, seed=13, scale=3.0) + 0.5, 1 + 0.1 * _rand(d, seed=14), 0.1 * _rand(d, seed=15)
    xq, aq, bq = _q(x, dtype), _q(a, dtype), _q(b, dtype)
    mean = xq.mean(-1, keepdim=True)
    ref = aq * (xq - mean) / (xq.std(-1, keepdim=True) + 1e-6) + bq          # torch.std is unbiased
    _cmp(ops.layernorm(x.to(dtype).cuda(), a.to(dtype).cuda(), b.to(dtype).cuda()), ref, tol, "layernorm")
    # the reference LayerNorm is NOT F.layer_norm: the two differ by > 1e-3 on this data
    assert (torch.nn.functional.layer_norm(xq, (d,), aq, bq, 1e-6) - ref).abs().max() > 1e-3


def _ref_attn(q, k, v, mask, h):
    N, Lq, d = q.shape
    dk = d // h
    qs, ks, vs = (t.view(N, -1, h, dk).transpose(1, 2) for t in (q, k, v))
    sc = qs @ ks.transpose(-1, -2) / math.sqrt(dk)
    if mask is not None:
        sc = sc.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    p = torch.softmax(sc, -1)
    return (p @ vs).transpose(1, 2).reshape(N, Lq, d), p


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("N,Lq,Lk,h,dk,mk", [
    (3, 5, 7, 4, 16, "key"), (2, 20, 20, 8, 64, "key"), (2, 9, 9, 8, 8, "causal"),
    (2, 6, 130, 1, 64, "key"), (2, 4, 11, 2, 32, None)])
def test_mha_core(ops, dtype, tol, N, Lq, Lk, h, dk, mk):
    d = h * dk
    q, k, v = _rand(N, Lq, d, seed=16), _rand(N, Lk, d, seed=17), _rand(N, Lk, d, seed=18)
    mask = None
    if mk == "key":
        mask = torch.ones(N, 1, Lk, dtype=torch.bool)
        mask[0, 0, Lk // 2:] = False
        mask[N - 1, 0, :] = False                  # fully masked -> uniform, not NaN
    elif mk == "causal":
        mask = torch.tril(torch.ones(1, Lq, Lk, dtype=torch.bool)).expand(N, Lq, Lk).clone()
        mask[1, :, 0] = False
    ref, pref = _ref_attn(_q(q, dtype), _q(k, dtype), _q(v, dtype), mask, h)
    out, p = ops.mha_core(q.to(dtype).cuda(), k.to(dtype).cuda(), v.to(dtype).cuda(),
                          None if mask is None else mask.cuda(), h, want_p=True)
    _cmp(out, ref, tol, "mha_core out")
    _cmp(p, pref, 1e-5 if dtype == torch.float32 else 1e-2, "mha_core p_attn")
    if mk == "key":
        assert torch.allclose(p[N - 1].cpu(), torch.full_like(p[N - 1].cpu(), 1.0 / Lk), atol=1e-6)
    # strided views: q/k/v as column slices of one packed projection
    packed = torch.cat([q, q, q], -1).to(dtype).cuda()
    if Lq == Lk:
        kk = torch.cat([q, k, v], -1).to(dtype).cuda()
        out2, _ = ops.mha_core(kk[..., :d], kk[..., d:2 * d], kk[..., 2 * d:], None if mask is None else mask.cuda(), h)
        _cmp(out2, ref, tol, "mha_core strided views")
    del packed


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("B,T,S,Lq,h,dk", [(2, 6, 9, 7, 4, 16), (2, 8, 49, 20, 8, 8), (1, 32, 49, 20, 8, 64), (2, 128, 49, 20, 8, 64),
                                         (2, 20, 49, 32, 2, 64)])
@pytest.mark.parametrize("direction", [0, 1])
def test_st_stage1_pv(ops, dtype, tol, B, T, S, Lq, h, dk, direction):
    d = h * dk
    sc = _rand(B, Lq * h, T * S, seed=19, scale=2.0)
    v = _rand(B, T, S, 2 * d, seed=20)                      # V is a column slice of a wider buffer
    tm = torch.ones(B, 1, T, dtype=torch.bool)
    tm[0, 0, T // 2:] = False
    if B > 1:
        tm[B - 1, 0, :] = False                              # one fully masked clip -> uniform over t
    scq = _q(sc, torch.float32).view(B, Lq, h, T, S)
    vq = _q(v[..., d:], dtype).view(B, T, S, h, dk)
    if direction == 0:
        s_ = scq.masked_fill(tm.view(B, 1, 1, T, 1) == 0, -1e9)
        p = torch.softmax(s_, dim=3)
        ref = torch.einsum("bihts,btshc->bsihc", p, vq).reshape(B, S, Lq, d)
    else:
        p = torch.softmax(scq, dim=4)
        ref = torch.einsum("bihts,btshc->btihc", p, vq).reshape(B, T, Lq, d)
    vd = v.to(dtype).cuda()
    out = ops.st_stage1_pv(sc.cuda(), vd[..., d:], tm.cuda() if direction == 0 else None, B=B, T=T, S=S, Lq=Lq, h=h, dk=dk,
                           direction=direction)
    _cmp(out, ref, tol, f"st_stage1_pv dir{direction}")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("B,G,Lq,h,d,masked", [(2, 9, 7, 4, 64, False), (2, 49, 20, 8, 64, False), (2, 32, 20, 8, 512, True),
                                               (1, 128, 5, 8, 512, True)])
def test_st_stage2(ops, dtype, tol, B, G, Lq, h, d, masked):
    q2f, y = _rand(B, Lq, h, d, seed=21, scale=d ** -0.5), _rand(B, G, Lq, d, seed=22)
    gm = None
    if masked:
        gm = torch.ones(B, 1, G, dtype=torch.bool)
        gm[0, 0, G // 3:] = False
        if B > 1:
            gm[B - 1, 0, :] = False
    qq, yq = _q(q2f, dtype), _q(y, dtype)
    sc = torch.einsum("bihe,bgie->bihg", qq, yq)
    if gm is not None:
        sc = sc.masked_fill(gm.view(B, 1, 1, G) == 0, -1e9)
    ref = torch.einsum("bihg,bgie->bihe", torch.softmax(sc, -1), yq)
    out = ops.st_stage2(q2f.to(dtype).cuda(), y.to(dtype).cuda(), None if gm is None else gm.cuda(), h=h)
    _cmp(out, ref, tol, "st_stage2")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, BF16_TOL)])
def test_embed_temporal_mask_fuse_cast(ops, dtype, tol):
    V, d, B, L = 50, 64, 3, 9
    lut = _rand(V, d, seed=23)
    ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(24))
    pos = torch.arange(0.0, 20).unsqueeze(1)
    div = torch.exp(torch.arange(0.0, d, 2) * -(math.log(10000.0) / d))
    pe = torch.zeros(20, d); pe[:, 0::2] = torch.sin(pos * div); pe[:, 1::2] = torch.cos(pos * div)
    ref = _q(lut, dtype)[ids] * math.sqrt(d) + pe[:L].double()
    _cmp(ops.embed_pe(ids.cuda(), lut.to(dtype).cuda(), pe.cuda()), ref, tol, "embed_pe")

    fts = _rand(B, 6, 5, 16, seed=25)
    fts[0, 4:] = 0; fts[2] = 0
    m = ops.temporal_mask(fts.to(dtype).cuda()).cpu()
    assert torch.equal(m, (fts.sum(2).sum(-1) != 0).unsqueeze(-2))

    xs = [_rand(B, L, d, seed=30 + j) for j in range(3)]
    score = _rand(B, L, 3, seed=26)
    w = torch.softmax(_q(score, dtype), -1)
    ref = sum(w[..., j:j + 1] * _q(xs[j], dtype) for j in range(3))
    _cmp(ops.fuse_modalities(score.to(dtype).cuda(), [x.to(dtype).cuda() for x in xs]), ref, tol, "fuse")

    x = _rand(1000, seed=27)
    assert torch.equal(ops.cast(x.cuda(), torch.bfloat16).cpu(), x.to(torch.bfloat16))
    assert torch.equal(ops.cast(x.to(torch.bfloat16).cuda(), torch.float32).cpu(), x.to(torch.bfloat16).float())


@pytest.mark.parametrize("B,T,S,Lq", [(2, 32, 49, 20), (1, 128, 49, 20), (2, 8, 9, 7), (2, 20, 49, 32), (3, 40, 5, 1), (1, 33, 64, 24)])
@pytest.mark.parametrize("direction", [0, 1])
def test_st_stage1_fused_matches_textbook_attention(ops, B, T, S, Lq, direction):
    """bist_st_stage1_fused_fwd against modules.py:54-64 / 81-100 + the residual of modules.py:44 written out per group in fp64 on the
    bf16-rounded operands (K, V and the expanded query materialised, as the reference does): x + W_o MHA(.) + b_o for every (b, g)."""
    d, h, dk = 512, 8, 64
    dt = torch.bfloat16
    assert ops.st_stage1_fused_ok(T, S, Lq, d, h, direction, dt)
    G, K = (S, T) if direction == 0 else (T, S)
    vft = _rand(B, T, S, d, seed=31)
    qf = _rand(B, Lq * h, d, seed=32, scale=1.5 * d ** -0.5)          # folded, pre-scaled query rows (i, hh): scores of O(1.5)
    wv, bv = _rand(d, d, seed=33, scale=d ** -0.5), _rand(d, seed=34, scale=0.1)
    wo, bo = _rand(d, d, seed=35, scale=d ** -0.5), _rand(d, seed=36, scale=0.1)
    x = _rand(B, Lq, d, seed=37)
    km = None
    if direction == 0 or B > 1:
        km = torch.ones(B, K, dtype=torch.bool)
        km[0, K // 2:] = False
        if B > 1:
            km[B - 1, :] = False                                       # fully masked clip -> uniform over the keys
    X = _q(vft, dt)
    Xg = X.permute(0, 2, 1, 3) if direction == 0 else X                # [B, G, K, d]
    sc = torch.einsum("bihe,bgke->bgihk", _q(qf, dt).view(B, Lq, h, d), Xg)
    if km is not None:
        sc = sc.masked_fill(km.view(B, 1, 1, 1, K) == 0, -1e9)
    p = torch.softmax(sc, -1)
    v = (Xg @ _q(wv, dt).t() + _q(bv, dt)).view(B, G, K, h, dk)
    ctx = torch.einsum("bgihk,bgkhc->bgihc", p, v).reshape(B, G, Lq, d)
    ref = _q(x, dt).view(B, 1, Lq, d) + ctx @ _q(wo, dt).t() + _q(bo, dt)
    c = lambda t: t.to(dt).cuda()
    out = ops.st_stage1_fused(c(qf), c(vft), None if km is None else km.cuda(), ops.pack_frag_rows(c(wv)), c(bv), ops.pack_frag_rows(c(wo)), c(bo), c(x),
                              h=h, direction=direction)
    assert out.shape == (B, G, Lq, d)
    _cmp(out, ref, 2e-2, f"st_stage1_fused dir{direction}")


def test_multi_set_row_ops_equal_their_single_set_forms(ops):
    """bist_layernorm_fwd_multi / bist_layernorm_bwd_multi / bist_scaled_bias_fwd_z (the C ABI's several-sets-per-launch row operations:
    the two directions' instances of a sublayer norm, include/bist_hip.h) against the reference's LayerNorm (modules.py:28-31) and its
    autograd gradients per set."""
    from oracle import bist_oracle as O
    rows, d, h = 40, 128, 8
    xs = [_rand(rows, d, seed=70 + i).cuda() for i in range(2)]
    ga = [(1 + 0.2 * _rand(d, seed=72 + i)).cuda() for i in range(2)]
    gb = [(0.1 * _rand(d, seed=74 + i)).cuda() for i in range(2)]
    outs = [torch.empty_like(x) for x in xs]
    ops.layernorm_multi(xs, ga, gb, outs, 1e-6)
    dys = [_rand(rows, d, seed=76 + i).cuda() for i in range(2)]
    dxs = [torch.empty_like(x) for x in xs]
    das, dbs = [torch.zeros(d, device="cuda") for _ in range(2)], [torch.zeros(d, device="cuda") for _ in range(2)]
    ops.layernorm_bwd_multi([(dys[i], xs[i], ga[i], dxs[i], das[i], dbs[i], None, None, 0) for i in range(2)], rows, d, d, d, d, 1e-6, 0, None, torch.float32)
    for i in range(2):
        x, a, b = xs[i].double().cpu().requires_grad_(True), ga[i].double().cpu().requires_grad_(True), gb[i].double().cpu().requires_grad_(True)
        y = O.layer_norm(x, a, b)
        y.backward(dys[i].double().cpu())
        _cmp(outs[i], y.detach(), 1e-5, f"layernorm_multi set {i}")
        _cmp(dxs[i], x.grad, 1e-4, f"layernorm_bwd_multi dx set {i}")
        _cmp(das[i], a.grad, 1e-4, f"layernorm_bwd_multi da set {i}")
        _cmp(dbs[i], b.grad, 1e-4, f"layernorm_bwd_multi db set {i}")
    # scaled bias over two stacked row blocks, each with its own bias (at bias0 + z * stride)
    x = _rand(2 * rows, d, seed=80).cuda()
    sc = torch.rand(2 * rows, h, device="cuda")
    bias = _rand(2, d, seed=81).cuda()
    got = ops.scaled_bias_z(x, sc, bias, d, h, 2)
    want = torch.cat([ops.scaled_bias(x[z * rows:(z + 1) * rows].contiguous(), sc[z * rows:(z + 1) * rows].contiguous(), bias[z], h) for z in range(2)])
    assert torch.equal(got, want)


def test_pack_frag_rows_layout(ops):
    """bist_pack_frag_rows: block (tile nt, pair kp, parity e), lane (x, kg) holds W[16 nt + x][64 kp + 16 kg + 8 e .. +7]."""
    R, Ccols = 48, 192
    w = torch.arange(R * Ccols, dtype=torch.float32).reshape(R, Ccols) % 251
    got = ops.pack_frag_rows(w.to(torch.bfloat16).cuda()).float().cpu().reshape(R // 16, Ccols // 64, 2, 64, 8)
    for nt, kp, e, lane in [(0, 0, 0, 0), (1, 2, 1, 17), (2, 1, 0, 63), (2, 2, 1, 38)]:
        x, kg = lane & 15, lane >> 4
        c0 = 64 * kp + 16 * kg + 8 * e
        assert torch.equal(got[nt, kp, e, lane], w[16 * nt + x, c0:c0 + 8]), (nt, kp, e, lane)
    assert torch.equal(got.reshape(-1).sort().values, w.reshape(-1).sort().values)          # a permutation


@pytest.mark.parametrize("M,N,act,res", [(320, 512, 0, True), (320, 1536, 0, False), (320, 2048, 1, False), (400, 512, 0, True), (37, 512, 0, False)])
def test_layernorm_prologue_of_the_projection(M, N, act, res, monkeypatch):
    """bist_gemm's LayerNorm prologue (K = 512, bf16): x.W^T of LayerNorm(x) with the norm inside the GEMM, against
    bist_layernorm_fwd followed by the plain GEMM; also the normalised rows it stores (ln_out) and the pending-tensor plumbing
    of ops.layernorm(lazy=True) -> ops.linear."""
    from bist_amd import ops
    from bist_amd._lib import lib
    monkeypatch.setattr(ops, "LAZY_LN", True)          # (opt-in: BIST_LAZY_LN=1)
    torch.manual_seed(M + N)
    x = (torch.randn(M, 512) * 1.7 + 0.3).cuda().bfloat16()
    a = (1 + 0.2 * torch.randn(512)).cuda().bfloat16()
    b = (0.1 * torch.randn(512)).cuda().bfloat16()
    w = (torch.randn(N, 512) * 0.05).cuda().bfloat16()
    bias = (torch.randn(N) * 0.1).cuda().bfloat16()
    r = torch.randn(M, N).cuda().bfloat16() if res else None
    assert ops.ln_lazy_ok(x, a, b)
    want_n = ops.layernorm(x, a, b, 1e-6)
    want = ops.linear(want_n, w, bias, act=act, residual=r)
    pend = ops.layernorm(x, a, b, 1e-6, lazy=True)
    assert getattr(pend, "_bist_ln", None) is not None
    lib.bist_launch_count_reset()
    got = ops.linear(pend, w, bias, act=act, residual=r)
    assert getattr(pend, "_bist_ln", None) is None
    torch.cuda.synchronize()
    # the normalised rows: same arithmetic, another summation order -> at most one bf16 step on a few elements
    dn = (pend.float() - want_n.float()).abs()
    assert dn.max().item() <= 2 ** -6 * max(1.0, want_n.float().abs().max().item()) and (dn > 0).float().mean().item() < 0.02, (dn.max(), (dn > 0).float().mean())
    d = (got.float() - want.float()).abs().max().item()
    assert d <= 3e-2 * max(1.0, want.float().abs().max().item()), d
    # pending output consumed by something that is not a fusable projection: materialised on demand
    pend2 = ops.layernorm(x, a, b, 1e-6, lazy=True)
    assert torch.equal(ops.ensure_ln(pend2), want_n)


def test_gemm_256_tile_launch_sheds_its_underfilled_last_round():
    """[100352 x 2048] x [2048 -> 512] (P0 at B = 64) is 784 tiles of 256 x 256 = 3 rounds of 256 CUs + 16 tiles: the last 8 row blocks
    go through bist_gemm as a product of their own.  Every row against a float64 reference of sampled rows (also across the seam)."""
    from bist_amd import ops
    torch.manual_seed(5)
    M, K, N = 100352, 2048, 512
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
    bias = (torch.randn(N, device="cuda") * 0.1).bfloat16()
    res = torch.randn(M, N, device="cuda").bfloat16()
    y = ops.linear(x, w, bias, residual=res)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 64), torch.arange(98304 - 64, 98304 + 64), torch.arange(M - 64, M), torch.randint(0, M, (256,))]).cuda()
    want = x[rows].double() @ w.double().t() + bias.double() + res[rows].double()
    err = (y[rows].double() - want).abs().max().item()
    assert err <= 2e-2 * max(1.0, want.abs().max().item()), err
    assert torch.isfinite(y.float()).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_text_vector_and_pointer_scatter_with_repeated_ids(dtype):
    """bist_text_vector_fwd against an fp64 einsum, and the pointer generator's scatter (bist_pointer_mix_fwd) with token ids that repeat
    inside a source and across sources against a plain index_add reference."""
    from bist_amd import ops
    torch.manual_seed(11)
    B, Lt, L, d, V = 3, 4, 37, 512, 200
    p = torch.softmax(torch.randn(B, Lt, L), -1).cuda()
    enc = torch.randn(B, L, d).cuda().to(dtype)
    got = ops.text_vector(p, enc).double().cpu()
    want = torch.einsum("bil,bld->bid", p.double().cpu(), enc.double().cpu())
    assert (got - want).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 1e-5)
    # scatter: two sources, ids drawn from a small set so that they repeat
    logits = torch.randn(B * Lt, V).cuda()
    sw = torch.randn(B * Lt, 3).cuda()
    ps = [torch.softmax(torch.randn(B, Lt, L_), -1).cuda() for L_ in (L, 19)]
    texts = [torch.randint(0, 12, (B, L_)).cuda() for L_ in (L, 19)]
    out = ops.pointer_mix(logits, sw, ps, texts, Lt).double().cpu()
    s = torch.softmax(sw.double().cpu(), -1)
    mix = s[:, 2:3] * torch.softmax(logits.double().cpu(), -1)
    for j in range(2):
        pj = ps[j].double().cpu().reshape(B * Lt, -1)
        ids = texts[j].cpu().repeat_interleave(Lt, dim=0)
        mix = mix.scatter_add(1, ids, s[:, j:j + 1] * pj)
    assert (out - mix.log()).abs().max().item() <= 1e-5


def test_new_entry_points_refuse_calls_outside_their_envelope():
    """bist_decoder_stack_fwd (slots beyond the pool, LkS not 32 / 64), bist_gemm with a LayerNorm prologue outside bist_gemm_ln_ok,
    bist_adam_apply_dev with a range that is not a multiple of 4: BIST_EINVAL and a message, no launch."""
    import ctypes as C
    from bist_amd import ops
    from bist_amd._lib import lib
    z = lambda *s, dt=torch.bfloat16: torch.zeros(*s, device="cuda", dtype=dt)
    bufs = {"x0": z(64, 512), "x1": z(64, 512), "q": z(64, 512), "kc": z(1, 64, 512), "vc": z(1, 64, 512), "h": z(64, 2048),
            "sync": z(8, dt=torch.int32)}
    desc = z(int(lib.bist_decoder_layer_desc_bytes()), dt=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    def stack(R, LkS, slot0, lk_pad_max=64):
        return lib.bist_decoder_stack_fwd(desc.data_ptr(), 1, bufs["x0"].data_ptr(), bufs["x0"].data_ptr(), bufs["x1"].data_ptr(), bufs["q"].data_ptr(),
                                          bufs["kc"].data_ptr(), bufs["vc"].data_ptr(), bufs["h"].data_ptr(), z(64, 64, dt=torch.uint8).data_ptr(),
                                          R, LkS, slot0, lk_pad_max, bufs["sync"].data_ptr(), None, 1, st)
    assert stack(5, 64, 60) != 0 and b"slot" in lib.bist_last_error()          # slots 60..64 leave the pool
    assert stack(5, 48, 0) != 0                                                  # LkS must be 32 or 64
    assert stack(40, 32, 0) != 0                                                 # more rows than key slots
    assert stack(5, 32, 0, lk_pad_max=96) != 0 and b"lk_pad_max" in lib.bist_last_error()      # padded memory lengths are 32, 64, 128, 256 or 512
    assert stack(5, 32, 0, lk_pad_max=1024) != 0
    # LayerNorm prologue with K != 512
    x, w, y = z(64, 256), z(512, 256), z(64, 512)
    g = ops.gemm_desc(x, w, y, M=64, N=512, K=256, a_rs=256, b_rs=256, ldc=512)
    a = z(256)
    g.ln_gain, g.ln_offset, g.ln_eps = a.data_ptr(), a.data_ptr(), 1e-6
    assert lib.bist_gemm_ln_ok(C.byref(g)) == 0 and lib.bist_gemm(C.byref(g), st) != 0
    f = lambda n: torch.zeros(n, device="cuda")
    assert lib.bist_adam_apply_dev(f(8).data_ptr(), f(8).data_ptr(), f(8).data_ptr(), f(8).data_ptr(), None, 6, f(8).data_ptr(), 0.9, 0.98, 1e-9, 0, 0, st) != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,K", [(25088, 2048), (100352, 2048), (1024, 512)])
def test_layernorm_epilogue_of_the_256_tile_product(M, K, monkeypatch):
    """LayerNorm(ReLU(x.W^T + b)) over N = 512 columns as ONE launch (BistGemm.ln_mode = 1: the two column-tile workgroups of a row
    block exchange their row statistics) against the product followed by bist_layernorm_fwd -- VidEncoder8's input projection at
    B = 16 and B = 64 (the latter also sheds its under-filled last round, whose rows get a LayerNorm launch of their own)."""
    import ctypes as C
    from bist_amd import ops
    from bist_amd._lib import lib
    monkeypatch.setattr(ops, "LN_EPILOGUE", True)          # (opt-in: BIST_LN_EPILOGUE=1)
    torch.manual_seed(M % 1000 + K)
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(512, K, device="cuda") * (1.0 / K ** 0.5)).bfloat16()
    bias = (torch.randn(512, device="cuda") * 0.1).bfloat16()
    a = (1 + 0.2 * torch.randn(512, device="cuda")).bfloat16()
    b = (0.1 * torch.randn(512, device="cuda")).bfloat16()
    want = ops.layernorm(ops.linear(x, w, bias, act=ops.ACT_RELU), a, b, 1e-6)
    y = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
    g = ops.gemm_desc(x, w, y, M=M, N=512, K=K, a_rs=K, b_rs=K, ldc=512, bias=bias, act=ops.ACT_RELU)
    g.ln_gain, g.ln_offset, g.ln_eps, g.ln_mode = a.data_ptr(), b.data_ptr(), 1e-6, 1
    assert lib.bist_gemm_ln_ok(C.byref(g)) == 1
    got = ops.linear(x, w, bias, act=ops.ACT_RELU, ln_out=(a, b, 1e-6))
    torch.cuda.synchronize()
    d = (got.float() - want.float()).abs()
    # same values up to the rounding of the bf16 activations the separate LayerNorm reads (the epilogue normalises the fp32 sums)
    assert bool((d <= 2 ** -5 * torch.clamp(want.float().abs(), min=1.0)).all()) and d.mean().item() <= 4e-3, (d.max().item(), d.mean().item())
    ws = ops._workspace(x.device)
    assert int(ws.view(torch.int32)[:1024].abs().max().item()) == 0, "exchange flags / error word must be zero after the launch"


@pytest.mark.parametrize("B,G,Lq,d,with_add", [(16, 49, 20, 512, True), (3, 32, 20, 512, False), (2, 5, 7, 64, True), (1, 128, 20, 512, True)])
def test_group_sum_mask_equals_the_two_separate_passes(B, G, Lq, d, with_add):
    """bist_group_sum_mask (the head of the fused stage-1 launch's backward: sum of dy over the groups + the other gradient of the
    un-expanded query, and the dropout-masked dy, in ONE pass; four group slices per workgroup) against the two passes it replaces:
    bist_group_sum_add and bist_epilogue_bwd with the same (p, seed) -- the masked copy bit-identical (same counter-based mask, same
    element index), the sum within bf16 rounding of the f32 sum of the bf16 addends."""
    from bist_amd import _lib, ops as O_
    from bist_amd._lib import check, lib
    g = torch.Generator().manual_seed(B * 1000 + G)
    dy = torch.randn(B, G, Lq, d, generator=g).to(torch.bfloat16).cuda()
    add = torch.randn(B * Lq, d, generator=g).to(torch.bfloat16).cuda() if with_add else None
    drop = (0.1, 12345)
    M = B * G * Lq
    out1, dz1 = torch.empty(B * Lq, d, device="cuda", dtype=torch.bfloat16), torch.empty_like(dy)
    check(lib.bist_group_sum_mask(dy.data_ptr(), add.data_ptr() if add is not None else None, out1.data_ptr(), dz1.data_ptr(), B, G, Lq * d,
                                  O_.drop_ref(drop), O_.dtype_code(dy.dtype), O_._stream()), "bist_group_sum_mask")
    dz2 = torch.empty_like(dy)
    check(lib.bist_epilogue_bwd(dy.data_ptr(), dy.data_ptr(), dz2.data_ptr(), M, d, d, d, d, O_.ACT_NONE, drop[0], drop[1], O_._ptr(O_.DROP_CTR),      # (the same step counter, if a trainer of an earlier test left one)
                               
                                O_.dtype_code(dy.dtype), O_._stream()), "bist_epilogue_bwd")
    torch.cuda.synchronize()
    assert torch.equal(dz1, dz2)
    kept = (dz1 != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.01
    ref = dy.float().sum(1).reshape(B * Lq, d) + (add.float() if add is not None else 0)
    err = (out1.float() - ref).abs().max().item()
    assert err <= 2.0 ** -7 * ref.abs().max().item(), err


def test_decoder_cache_fill_copies_keys_and_transposes_values():
    """bist_decoder_cache_fill (the persistent decoder kernel's per-turn caches out of the packed [k | v] projections, decoder.py:42-55):
    for jobs of different lengths and paddings, K rows = the k halves, VT = the v halves transposed with columns Lk..LkP-1 zero -- bit for
    bit; rows of K beyond Lk are left alone."""
    import ctypes as C
    from bist_amd import ops as O_
    from bist_amd._lib import BistKvFill, check, lib
    g = torch.Generator().manual_seed(3)
    specs = [(20, 32, 1024), (60, 64, 6144), (33, 64, 1024), (1, 32, 2048)]
    jobs = (BistKvFill * len(specs))()
    keep = []
    for j, (Lk, LkP, ld) in enumerate(specs):
        src = torch.randn(Lk, ld, generator=g).to(torch.bfloat16).cuda()
        off = 1024 * (j % 2) if ld > 1024 else 0                       # a [k | v] pair somewhere inside a wider packed row
        K = torch.full((LkP, 512), 7.0, dtype=torch.bfloat16).cuda()
        VT = torch.full((512, LkP), 7.0, dtype=torch.bfloat16).cuda()
        keep.append((src, off, K, VT, Lk, LkP))
        jobs[j].src, jobs[j].K, jobs[j].VT = src.data_ptr() + off * 2, K.data_ptr(), VT.data_ptr()
        jobs[j].Lk, jobs[j].LkP, jobs[j].ld = Lk, LkP, ld
    check(lib.bist_decoder_cache_fill(jobs, len(specs), O_.dtype_code(torch.bfloat16), O_._stream()), "bist_decoder_cache_fill")
    torch.cuda.synchronize()
    for src, off, K, VT, Lk, LkP in keep:
        assert torch.equal(K[:Lk], src[:, off:off + 512])
        assert bool((K[Lk:] == 7.0).all())
        assert torch.equal(VT[:, :Lk], src[:, off + 512:off + 1024].t())
        assert bool((VT[:, Lk:] == 0).all())


@pytest.mark.gpu
def test_stage_inputs_copies_and_pads_every_field_in_one_launch():
    """bist_stage_inputs (the inputs of a dialogue turn into the buffers its hipGraphs read): int64 token rows padded with the pad id, bool
    mask rows with False, several rows per job, a 16-byte-aligned bulk copy (the feature tensor) and an odd-sized byte copy, 17 jobs (two
    launches) -- against torch's pad / copy."""
    from bist_amd import ops
    g = torch.Generator().manual_seed(3)
    tok = torch.randint(2, 900, (1, 21), generator=g).cuda()
    msk = (torch.rand(2, 1, 21, generator=g) > 0.3).cuda()
    fts = torch.randn(1, 40, 64, generator=g).to(torch.bfloat16).cuda()
    odd = torch.randint(0, 255, (3, 7), generator=g).to(torch.uint8).cuda()
    h16 = torch.randn(5, 3, generator=g).to(torch.bfloat16).cuda()
    jobs, want = [], []
    for k in range(13):                                   # thirteen token tensors of different bucket sizes
        dst = torch.full((1, 24 + 8 * (k % 3)), -7, dtype=torch.long, device="cuda")
        jobs.append((tok, dst, 1)); want.append(torch.nn.functional.pad(tok, (0, dst.shape[1] - 21), value=1))
    for src, width, pad in ((msk, 24, 0), (fts, 64, 0), (odd, 7, 0), (h16, 8, 0x3F80)):       # 0x3F80 = bf16 1.0
        dst = torch.empty(src.shape[:-1] + (width,), dtype=src.dtype, device="cuda")
        if src.dtype != torch.bool:
            dst.fill_(3)
        jobs.append((src, dst, pad))
        val = {0: 0, 0x3F80: 1.0}[pad]
        want.append(torch.nn.functional.pad(src, (0, width - src.shape[-1]), value=bool(val) if src.dtype == torch.bool else val))
    ops.stage_inputs(jobs)
    torch.cuda.synchronize()
    for (src, dst, _), w in zip(jobs, want):
        assert torch.equal(dst, w), (tuple(src.shape), tuple(dst.shape), src.dtype)
    with pytest.raises(ValueError):
        ops.stage_inputs([(tok, torch.empty(1, 16, dtype=torch.long, device="cuda"), 1)])      # destination rows shorter than the source's


@pytest.mark.gpu
def test_captures_run_with_the_cyclic_collector_paused():
    """functional.capture_graph: between capture begin and end Python's cyclic collector must not run -- garbage it finds may own hipGraphs
    or device buffers of earlier captures, and destroying those inside a stream capture aborts the process (seen with torch 2.10, which
    no longer collects before a capture).  The collector's state is restored afterwards, also when the captured code raises."""
    import gc
    from bist_amd import functional as Fn, ops
    x = torch.ones(64, 512, device="cuda", dtype=torch.bfloat16)
    a = torch.ones(512, device="cuda", dtype=torch.bfloat16)
    ops.layernorm(x, a, a)                                  # (warm)
    assert gc.isenabled()
    g = torch.cuda.CUDAGraph()
    with Fn.capture_graph(g):
        assert not gc.isenabled()
        y = ops.layernorm(x, a, a)
    assert gc.isenabled()
    g.replay(); torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError):
        with Fn.capture_graph(g2):
            ops.layernorm(x, a, a)
            raise RuntimeError("inside the capture")
    assert gc.isenabled()
    gc.disable()
    try:
        with Fn.capture_graph(torch.cuda.CUDAGraph()):      # a caller that had the collector off keeps it off
            ops.layernorm(x, a, a)
        assert not gc.isenabled()
    finally:
        gc.enable()
