"""GPU parity tests: the HIP path (through the C ABI) against golden vectors exported from the reference and against
the oracle on the same seeded inputs.

Bars (BASELINE.json north_star): index tensors bit-exact; fp32 forward/backward within 1e-5 relative.  "Relative" is
read against the scale of each tensor (max |ref|); gradient tensors that are analytically ~0 are judged on 5 % of the
largest gradient tensor (helpers.rel_err).  Where the reference's own fp32 result sits further than 1e-5 from the
fp64 oracle (long reductions, the tiny ill-conditioned e2e fixture), the bound is stated next to the assertion.
"""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _glist_dev(mols, device, atoms=None):
    gl = H.oracle_glist(mols, atoms)
    return [t.to(device) if torch.is_tensor(t) else t for t in gl]


def _load_sd(module, z, prefix):
    sd = H.sd(z, prefix)
    module.load_state_dict(sd)
    return module


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("layout", ["kc_kc", "kc_rc", "rc_kc", "rc_rc"])
@pytest.mark.parametrize("shape", [(5, 7, 3), (128, 128, 16), (130, 67, 75), (300, 194, 64), (257, 512, 242), (64, 1, 637),
                                   (128, 128, 32), (256, 384, 96), (1280, 640, 160)])   # last three: aligned fast form
def test_gemm_layouts(device, layout, shape):
    from jova_amd.nn import functional as F_
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    want = torch.relu((a.double() @ b.double()) * 0.5 + bias.double() + res.double()).float()
    a_kc, b_kc = layout.split("_")
    A = (a if a_kc == "kc" else a.t().contiguous()).to(device)         # [M,K] or [K,M]
    B = (b.t().contiguous() if b_kc == "kc" else b).to(device)         # [N,K] or [K,N]
    sam, sak = (K, 1) if a_kc == "kc" else (1, M)
    sbk, sbn = (1, K) if b_kc == "kc" else (N, 1)
    C = torch.full((M, N), float("nan"), device=device)
    bias_d, res_d = bias.to(device), res.to(device)            # keep alive: the ABI takes raw pointers
    F_.gemm(F_._addr(A), sam, sak, F_._addr(B), sbk, sbn, F_._addr(C), N, M, N, K, alpha=0.5, bias=F_.P(bias_d),
            residual=F_._addr(res_d), ldr=N, relu=True)
    H.assert_close(C, want, tol=2e-6, what="gemm " + layout)


@pytest.mark.parametrize("shape", [(128, 512, 1024), (256, 128, 4096), (128, 128, 256)])
def test_gemm_few_tiles_long_k_is_split_with_bias(device, shape):
    """The aligned form splits a product of a few tiles along K by itself (ecfp view projection, 128 x 512 x 1024):
    partial sums onto a zeroed C, the bias added once; C starts as NaN, alpha rides along."""
    from jova_amd.nn import functional as F_
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    a, b, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    want = ((a.double() @ b.double().t()) * 0.25 + bias.double()).float()
    A, B, bias_d = a.to(device), b.to(device), bias.to(device)
    C = torch.full((M, N), float("nan"), device=device)
    F_.gemm(F_._addr(A), K, 1, F_._addr(B), 1, K, F_._addr(C), N, M, N, K, alpha=0.25, bias=F_.P(bias_d))
    H.assert_close(C, want, tol=2e-6, what="few-tile split")


@pytest.mark.parametrize("shape", [(3840, 2048, 512), (1280, 512, 512), (256, 128, 128), (640, 242, 242), (77, 22, 512),
                                   (23168, 512, 22), (4099, 22, 242), (2048, 22, 22), (2500, 1, 300), (6000, 32, 512),
                                   (128, 4096, 2048)])          # (the last: bf16-pieces form without split-K, dw through the residual)
def test_linear_bwd_weight_fuses_the_bias_gradient(device, shape):
    """jova_linear_bwd_weight: dw += dy^T x and db += colsum(dy), both ACCUMULATING, on the aligned form (bias gradient
    summed from the product's own A fragments, with and without split-K) and on the generic form (separate pass); one side <= 32 wide over >= 2048 rows takes the streaming reduction."""
    from jova_amd.nn import functional as F_
    rows, n_out, k_in = shape
    g = torch.Generator().manual_seed(rows + n_out)
    dy, x = torch.randn(rows, n_out, generator=g), torch.randn(rows, k_in, generator=g)
    dw0, db0 = torch.randn(n_out, k_in, generator=g), torch.randn(n_out, generator=g)
    want_dw = (dw0.double() + dy.double().t() @ x.double()).float()
    want_db = (db0.double() + dy.double().sum(0)).float()
    dyd, xd, dw, db = dy.to(device), x.to(device), dw0.to(device), db0.to(device)
    F_.linear_bwd_weight(dyd, xd, dw, db)
    H.assert_close(dw, want_dw, tol=2e-6, what="dw")
    H.assert_close(db, want_db, tol=5e-6, what="db")             # fp32 sums of up to 23 k terms
    dw2 = dw0.to(device)
    F_.linear_bwd_weight(dyd, xd, dw2, None)                    # without a bias
    H.assert_close(dw2, want_dw, tol=2e-6, what="dw (no bias)")


@pytest.mark.parametrize("mode", ["plain", "relu_mask", "residual"])
def test_backward_products_from_bf16_pieces_are_fp32_accurate(device, mode):
    """dX = dY W of a GFLOP or more runs on the bf16 matrix pipe, every fp32 operand split into three bf16 pieces
    (gemm_bf16x3_kernel).  Held to the error an fp32 FMA chain makes on the same product (measured here, against fp64), on
    operands whose magnitudes spread over twelve decades; the ReLU-mask and residual epilogues are exact selections /
    additions on top."""
    from jova_amd.nn import functional as F_
    M, N, K = 4096, 512, 1024                                     # dy [M, N], w [N, K] -> dx [M, K]; contraction over N
    g = torch.Generator().manual_seed(31)
    dy = torch.randn(M, N, generator=g) * torch.exp(6.0 * torch.randn(M, 1, generator=g))       # rows of very different size
    w = torch.randn(N, K, generator=g) * torch.exp(3.0 * torch.randn(1, K, generator=g))
    ref64 = dy.double() @ w.double()
    fp32_chain = torch.zeros(M, K)
    for n0 in range(0, N, 1):                                     # one fp32 rounding per term, in contraction order
        fp32_chain += dy[:, n0:n0 + 1] * w[n0:n0 + 1, :]
    scale = ref64.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)       # per output row: the rows differ by decades
    err32 = ((fp32_chain.double() - ref64).abs() / scale).max().item()
    extra = torch.randn(M, K, generator=g)
    dyd, wd, dx = dy.to(device), w.to(device), torch.full((M, K), float("nan"), device=device)
    if mode == "plain":
        F_.linear_bwd_data(dyd, wd, dx)
        want = ref64
    elif mode == "relu_mask":
        F_.linear_bwd_data(dyd, wd, dx, relu_mask=extra.to(device))
        want = torch.where(extra.double() > 0, ref64, torch.zeros_like(ref64))
    else:
        F_.linear_bwd_data(dyd, wd, dx, residual=(extra * scale.float()).to(device))
        want = ref64 + (extra * scale.float()).double()
    err = ((dx.cpu().double() - want).abs() / scale).max().item()
    assert err <= max(2e-6, 1.5 * err32), "bf16-pieces product: %.3e of the row scale, fp32 chain %.3e" % (err, err32)
    if mode == "plain":                                           # the switch: same product on the fp32 MFMA kernel
        from jova_amd._lib import lib
        was = lib().jova_gemm_get_bf16x3()                       # (0 under JOVA_GEMM_NO_BF16X3: put back what was found)
        lib().jova_gemm_set_bf16x3(0)
        try:
            dx32 = torch.full((M, K), float("nan"), device=device)
            F_.linear_bwd_data(dyd, wd, dx32)
        finally:
            lib().jova_gemm_set_bf16x3(was)
        e32k = ((dx32.cpu().double() - want).abs() / scale).max().item()
        assert e32k <= max(2e-6, 1.5 * err32), "fp32 MFMA kernel: %.3e" % e32k
        if was:
            assert not torch.equal(dx32, dx), "the switch changed nothing: both calls took the same kernel"


def _fp32_chain_error(a, b, ref64, scale):
    """Error of the plain fp32 multiply-add chain  sum_k a[:, k] b[k, :]  (one rounding per term) against fp64, in units of
    `scale` per output row: the yard-stick of the bf16-pieces products."""
    acc = torch.zeros(a.shape[0], b.shape[1])
    for k in range(a.shape[1]):
        acc += a[:, k:k + 1] * b[k:k + 1, :]
    return ((acc.double() - ref64).abs() / scale).max().item()


@pytest.mark.parametrize("case", ["dx_k2048", "dw_splitk", "dw_one_part"])
def test_bf16_pieces_second_shapes(device, case):
    """a second shape per kernel form.  dX with a 2048-long contraction (LDS-DMA form: the FFN
    linear2 data gradient), dW with split-K over 8192 rows and dW in one part (register-staged form)."""
    from jova_amd.nn import functional as F_
    from jova_amd._lib import lib
    if not lib().jova_gemm_get_bf16x3():
        pytest.skip("bf16-pieces products are switched off (JOVA_GEMM_NO_BF16X3)")
    g = torch.Generator().manual_seed(77)
    if case == "dx_k2048":
        M, N, K = 2048, 2048, 512                                  # dy [M, N] w [N, K]: contraction over N = 2048
        dy = torch.randn(M, N, generator=g) * torch.exp(4.0 * torch.randn(M, 1, generator=g))
        w = torch.randn(N, K, generator=g)
        ref64 = dy.double() @ w.double()
        scale = ref64.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        err32 = _fp32_chain_error(dy, w, ref64, scale)
        dx = torch.full((M, K), float("nan"), device=device)
        F_.linear_bwd_data(dy.to(device), w.to(device), dx)
        err = ((dx.cpu().double() - ref64).abs() / scale).max().item()
    else:
        rows, n_out, k_in = (8192, 512, 2048) if case == "dw_splitk" else (1024, 2048, 2048)
        dy = torch.randn(rows, n_out, generator=g) * torch.exp(3.0 * torch.randn(1, n_out, generator=g))
        x = torch.randn(rows, k_in, generator=g)
        ref64 = dy.double().t() @ x.double()
        scale = ref64.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        err32 = _fp32_chain_error(dy.t().contiguous(), x, ref64, scale)
        dw = torch.zeros(n_out, k_in, device=device)
        F_.linear_bwd_weight(dy.to(device), x.to(device), dw, None)
        err = ((dw.cpu().double() - ref64).abs() / scale).max().item()
    assert err <= max(2e-6, 1.5 * err32), "%s: %.3e of the row scale, fp32 chain %.3e" % (case, err, err32)


@pytest.mark.parametrize("form", ["dx", "dw"])
def test_bf16_pieces_operand_range(device, form):
    """The input range of the bf16-pieces products, pinned at both ends (include/jova_hip.h, jova_gemm_set_bf16x3):
      * top: operands up to 3.38e38 (the largest bfloat16 is 3.3895e38; fp32 goes to 3.4028e38) are handled at fp32
        accuracy; an operand above the bfloat16 range rounds its top piece to infinity and the affected outputs are NaN
        where an fp32 chain stays finite -- the documented limit, asserted here so that it cannot change silently;
      * bottom: operands down to 1e-32 keep all three pieces normal (the smallest piece is 2^-16 of the operand) and are
        handled at fp32 accuracy; exact zeros stay exact."""
    from jova_amd.nn import functional as F_
    from jova_amd._lib import lib
    if not lib().jova_gemm_get_bf16x3():
        pytest.skip("bf16-pieces products are switched off (JOVA_GEMM_NO_BF16X3)")
    g = torch.Generator().manual_seed(5)
    M, N, K = 2048, 1024, 512

    def run(a, b):                                                 # a [M, N], b [N, K] -> a b through the chosen form
        if form == "dx":
            out = torch.full((M, K), float("nan"), device=device)
            F_.linear_bwd_data(a.to(device), b.to(device), out)
        else:                                                      # dw[n_out = M, k_in = K] = dy^T x with dy = a^T
            out = torch.zeros(M, K, device=device)
            F_.linear_bwd_weight(a.t().contiguous().to(device), b.to(device), out, None)
        return out.cpu().double()

    sign = torch.where(torch.rand(M, N, generator=g) < 0.5, -1.0, 1.0)
    for name, lo, hi, bscale in (("top", 1.0e38, 3.38e38, 1e-6), ("bottom", 1.0e-32, 4.0e-32, 1e3)):
        a = sign * (lo + (hi - lo) * torch.rand(M, N, generator=g))
        a[:, 7] = 0.0                                              # exact zeros ride along
        b = torch.randn(N, K, generator=g) * bscale
        ref64 = a.double() @ b.double()
        assert float(ref64.abs().max()) < 3.0e38 and float(ref64.abs().max()) > 1e-36
        scale = ref64.abs().amax(dim=1, keepdim=True)
        err32 = _fp32_chain_error(a, b, ref64, scale)
        got = run(a, b)
        assert torch.isfinite(got).all(), name
        err = ((got - ref64).abs() / scale).max().item()
        assert err <= max(2e-6, 1.5 * err32), "%s of the range: %.3e of the row scale, fp32 chain %.3e" % (name, err, err32)
    # above the bfloat16 range: the documented limit
    a = torch.randn(M, N, generator=g)
    a[5, 11] = 3.4e38                                              # finite in fp32, beyond the largest bfloat16
    b = torch.randn(N, K, generator=g) * 1e-3
    got = run(a, b)
    assert not torch.isfinite(got[5]).all(), "an operand above 3.3895e38 is expected to poison its output row"
    keep = torch.ones(M, dtype=torch.bool)
    keep[5] = False
    assert torch.isfinite(got[keep]).all()
    was = lib().jova_gemm_get_bf16x3()
    lib().jova_gemm_set_bf16x3(0)                                 # the fp32 MFMA kernel handles it: the way out for such data
    try:
        assert torch.isfinite(run(a, b)).all()
    finally:
        lib().jova_gemm_set_bf16x3(was)


@pytest.mark.parametrize("shape", [(241, 241, 256, 32, 2), (256, 203, 64, 64, 1), (130, 250, 16, 3, 40)])
def test_gemm_ragged_batched_aligned_form(device, shape):
    """The per-(b,h) attention products with K = head_dim (scores = Q K^T, dP = dO V^T) take the aligned kernel although S
    is no multiple of 128: edge tiles read clamped rows and store guarded.  Operands laid out like qkv ([S, B, 3L])
    with NaN everywhere the product must not read or write."""
    from jova_amd.nn import functional as F_
    S1, S2, hd, nb1, nb2 = shape
    g = torch.Generator().manual_seed(S1 + S2)
    L = nb2 * hd
    q = torch.randn(S1, nb1, L, generator=g)
    k = torch.randn(S2, nb1, L, generator=g)
    want = torch.einsum("sbhd,tbhd->bhst", q.view(S1, nb1, nb2, hd).double(), k.view(S2, nb1, nb2, hd).double()) * 0.125
    Sp = (S2 + 3) // 4 * 4
    qd, kd = q.to(device), k.to(device)
    C = torch.full((nb1 * nb2, S1, Sp), float("nan"), device=device)
    F_.gemm(F_._addr(qd), nb1 * L, 1, F_._addr(kd), 1, nb1 * L, F_._addr(C), Sp, S1, S2, hd, alpha=0.125, nb1=nb1, nb2=nb2,
            sA=(L, hd), sB=(L, hd), sC=(nb2 * S1 * Sp, S1 * Sp))
    got = C.view(nb1, nb2, S1, Sp)
    H.assert_close(got[..., :S2], want.float(), tol=2e-6, what="ragged batched product")
    assert torch.isnan(got[..., S2:]).all(), "wrote into the row padding"


@pytest.mark.parametrize("layout", ["kc", "rc"])
def test_gemm_padded_rows_with_garbage_in_the_padding(device, layout):
    """Operands whose rows are padded to a multiple of 4 floats (the attention probabilities [S, S4]) take the 16-byte
    loads although the extent (278) is not a multiple of 4: whatever sits in the padding, NaN included, is dropped."""
    from jova_amd.nn import functional as F_
    g = torch.Generator().manual_seed(9)
    nb, M, N, K, ld = 3, 278, 64, 278, 280
    a = torch.randn(nb, M, K, generator=g)
    b = torch.randn(nb, K, N, generator=g)
    want = torch.einsum("xmk,xkn->xmn", a.double(), b.double()).float()
    if layout == "kc":                                           # A[m, k] at m * ld + k
        A = torch.full((nb, M, ld), float("nan"))
        A[:, :, :K] = a
        sam, sak, sA = ld, 1, M * ld
    else:                                                        # A[m, k] at k * ld + m
        A = torch.full((nb, K, ld), float("nan"))
        A[:, :, :M] = a.transpose(1, 2)
        sam, sak, sA = 1, ld, K * ld
    A, B = A.to(device), b.contiguous().to(device)
    C = torch.full((nb, M, N), float("nan"), device=device)
    F_.gemm(F_._addr(A), sam, sak, F_._addr(B), N, 1, F_._addr(C), N, M, N, K, nb1=nb, sA=(sA, 0), sB=(K * N, 0),
            sC=(M * N, 0))
    H.assert_close(C, want, tol=2e-6, what="padded " + layout)


def test_gemm_batched_strided_and_splitk(device):
    from jova_amd.nn import functional as F_
    g = torch.Generator().manual_seed(5)
    nb1, nb2, M, N, K = 3, 2, 70, 50, 40
    a = torch.randn(nb1, nb2, M, K, generator=g)
    b = torch.randn(nb1, nb2, N, K, generator=g)
    want = torch.einsum("xymk,xynk->xymn", a.double(), b.double()).float()
    A, B = a.to(device), b.to(device)
    C = torch.zeros(nb1, nb2, M, N, device=device)
    F_.gemm(F_._addr(A), K, 1, F_._addr(B), 1, K, F_._addr(C), N, M, N, K, nb1=nb1, nb2=nb2,
            sA=(nb2 * M * K, M * K), sB=(nb2 * N * K, N * K), sC=(nb2 * M * N, M * N))
    H.assert_close(C, want, tol=2e-6, what="batched")
    # split-K accumulates into C: start from ones
    M, N, K = 40, 30, 5000
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    C = torch.ones(M, N, device=device)
    a_d, b_d = a.to(device), b.to(device)
    F_.gemm(F_._addr(a_d), K, 1, F_._addr(b_d), 1, K, F_._addr(C), N, M, N, K, splitk=7)
    H.assert_close(C, (a.double() @ b.double().t() + 1).float(), tol=2e-6, what="splitk")
    # aligned fast form, weight-gradient shape: more work items than persistent workgroups, ragged last K split
    M, N, K = 1024, 1152, 32 * 37
    a = torch.randn(K, M, generator=g)
    b = torch.randn(K, N, generator=g)
    C = torch.ones(M, N, device=device)
    a_d, b_d = a.to(device), b.to(device)
    F_.gemm(F_._addr(a_d), 1, M, F_._addr(b_d), N, 1, F_._addr(C), N, M, N, K, splitk=8)
    H.assert_close(C, (a.double().t() @ b.double() + 1).float(), tol=2e-6, what="splitk fast")
    # aligned fast form, ragged last round: 520 tiles on 512 persistent workgroups -> the last 8 tiles are split along K
    # and accumulated with atomics onto zeroed rows; bias + residual come from part 0
    M, N, K = 8320, 1024, 512
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    a_d, b_d, bias_d, res_d = a.to(device), b.to(device), bias.to(device), res.to(device)
    C = torch.full((M, N), float("nan"), device=device)
    F_.gemm(F_._addr(a_d), K, 1, F_._addr(b_d), 1, K, F_._addr(C), N, M, N, K, bias=F_.P(bias_d),
            residual=F_._addr(res_d), ldr=N)
    H.assert_close(C, (a.double() @ b.double().t() + bias.double() + res.double()).float(), tol=2e-6, what="ragged tail")


# ------------------------------------------------------------------------------------------------ graph stack
@pytest.mark.parametrize("tag", ["l1", "l2", "lin"])
def test_graphconv_vs_reference_golden(device, tag):
    from jova_amd.nn.layers import GraphConvLayer
    z = H.load("op_graphconv")
    mols = H.unpack_mols(z)
    layer = _load_sd(GraphConvLayer(75 if tag == "l1" else 64, 64, activation=None if tag == "lin" else "relu"), z,
                     tag + ".sd.").to(device)
    x = None
    if tag != "l1":
        x = torch.from_numpy(z[tag + ".x"]).to(device).requires_grad_(True)
    gl = _glist_dev(mols, device, x)
    stats = torch.empty(2, 64, dtype=torch.float64, device=device)
    out = layer(gl, bn_stats=stats)
    H.assert_close(out, z[tag + ".out"], tol=1e-5, what="out")
    ref = torch.from_numpy(z[tag + ".out"]).double()
    H.assert_close(stats[0], ref.sum(0).numpy(), tol=1e-5, what="fused BN column sums")
    H.assert_close(stats[1], (ref * ref).sum(0).numpy(), tol=1e-5, what="fused BN column sums of squares")
    (out * torch.from_numpy(z[tag + ".gout"]).to(device)).sum().backward()
    for k in range(21):
        H.assert_close(layer.linear_layers[k].weight.grad, z["%s.grad.linear_layers.%d.weight" % (tag, k)], tol=1e-5,
                       what="dW%d" % k)
        H.assert_close(layer.linear_layers[k].bias.grad, z["%s.grad.linear_layers.%d.bias" % (tag, k)], tol=1e-5,
                       what="db%d" % k)
    if x is not None:
        H.assert_close(x.grad, z[tag + ".gx"], tol=1e-5, what="dx")


@pytest.mark.parametrize("mode", ["plan", "tile32_plan", "noplan", "two_buffers", "generic"])
@pytest.mark.parametrize("layer_kind", ["l1_relu", "l1_lin", "l2_relu", "l2_lin"])
def test_graphconv_backward_all_degrees_vs_oracle(device, layer_kind, mode, monkeypatch):
    """Backward of GraphConvLayer on a batch that populates every degree bucket 0..10 (multi-round gathers, short last
    rounds, partial last tiles) against fp64 autograd of the oracle: the wave-private form (dZ handed in: the "lin" kinds)
    / the 32-atom-tile LDS-DMA form with the host work plan, the latter also without it and with two LDS buffers, and the
    generic form.  75 -> 64 from int64 atoms (dW only) and 64 -> 64 (dW + dX), with the
    ReLU mask applied in the kernel (relu) and without (lin)."""
    from jova_amd import synth
    from jova_amd.nn.layers import GraphConvLayer
    from oracle import nn_ref as R
    rng = np.random.RandomState(64)
    mols = [synth.make_molecule(rng, int(n), deg_cap=4, hub_frac=0.15) for n in rng.randint(14, 60, size=150)]
    mols.append(synth.make_molecule(rng, 1))             # a degree-0 atom
    if mode == "tile32_plan":                            # the workgroup-synchronous 32-atom-tile form, also without a mask
        monkeypatch.setenv("JOVA_BWD_OLD", "1")
    elif mode == "noplan":
        monkeypatch.setenv("JOVA_BWD_NOPLAN", "1")
        monkeypatch.setenv("JOVA_FWD_NOPLAN", "1")
    elif mode == "two_buffers":
        monkeypatch.setenv("JOVA_BWD_DB", "1")
    elif mode == "generic":
        monkeypatch.setenv("JOVA_GRAPHCONV_NO_DMA", "1")
    first = layer_kind.startswith("l1")
    relu = layer_kind.endswith("relu")
    torch.manual_seed(8)
    layer = GraphConvLayer(75 if first else 64, 64, activation="relu" if relu else None)
    gl_cpu = H.oracle_glist(mols)
    n = gl_cpu[0].shape[0]
    assert all(int(a.shape[0]) > 0 for a in gl_cpu[4:]) and int(gl_cpu[1][0, 1]) > 0, "buckets 0..10 populated"
    x = None if first else torch.randn(n, 64)
    gout = torch.randn(n, 64)
    w = [l.weight.detach().double().requires_grad_(True) for l in layer.linear_layers]
    b = [l.bias.detach().double().requires_grad_(True) for l in layer.linear_layers]
    xo = gl_cpu[0].double() if first else x.double().requires_grad_(True)
    ref = R.graph_conv(xo, gl_cpu[1], gl_cpu[4:], w, b, relu=relu)
    (ref * gout.double()).sum().backward()
    layer = layer.to(device)
    xd = None if first else x.to(device).requires_grad_(True)
    gl = _glist_dev(mols, device, xd)
    out = layer(gl)
    (out * gout.to(device)).sum().backward()
    H.assert_close(out, ref.detach().float(), tol=1e-5, what="out")
    gmax = max(float(t.grad.abs().max()) for t in w)
    for k in range(21):
        H.assert_close(layer.linear_layers[k].weight.grad, w[k].grad.float(), tol=1e-5, what="dW%d" % k, floor=0.05 * gmax)
        H.assert_close(layer.linear_layers[k].bias.grad, b[k].grad.float(), tol=1e-5, what="db%d" % k,
                       floor=0.05 * float(max(t.grad.abs().max() for t in b)))
    if not first:
        H.assert_close(xd.grad, xo.grad.float(), tol=1e-5, what="dx")


def test_graphconv_backward_atomic_path_for_asymmetric_adjacency(device):
    """A directed edge list (not a bond graph) must take the scatter path and still match the oracle."""
    from jova_amd.nn.layers import GraphConvLayer, GraphPool
    from oracle import nn_ref as R
    torch.manual_seed(3)
    n = 40
    deg_slice = torch.tensor([[0, 4]] + [[4, 12], [16, 12], [28, 12]] + [[40, 0]] * 7)
    adj = [torch.randint(0, n, (12, d), dtype=torch.int32) for d in (1, 2, 3)] + \
          [torch.zeros((0, d), dtype=torch.int32) for d in range(4, 11)]
    x = torch.randn(n, 64)
    layer = GraphConvLayer(64, 64)
    w = [l.weight.detach().double().requires_grad_(True) for l in layer.linear_layers]
    b = [l.bias.detach().double().requires_grad_(True) for l in layer.linear_layers]
    xo = x.double().requires_grad_(True)
    pooled = R.graph_pool(R.graph_conv(xo, deg_slice, adj, w, b), deg_slice, adj)
    gout = torch.randn(n, 64)
    (pooled * gout.double()).sum().backward()
    layer = layer.to(device)
    xd = x.to(device).requires_grad_(True)
    gl = [xd, deg_slice.to(device), None, [n]] + [a.to(device) for a in adj]
    out = GraphPool()([layer(gl)] + gl[1:])
    assert int(gl[1]._jova_topology.sym_flag.item()) == 0
    (out * gout.to(device)).sum().backward()
    H.assert_close(out, pooled.detach().float(), tol=1e-5, what="out")
    H.assert_close(xd.grad, xo.grad.float(), tol=1e-5, what="dx (atomic scatter)")
    H.assert_close(layer.linear_layers[2].weight.grad, w[2].grad.float(), tol=1e-5, what="dW")


def test_graphpool_ties_vs_reference_golden(device):
    from jova_amd.nn.layers import GraphPool
    z = H.load("op_graphpool")
    x = torch.from_numpy(z["x"]).to(device).requires_grad_(True)
    gl = _glist_dev(H.unpack_mols(z), device, x)
    out = GraphPool()(gl)
    assert int(gl[1]._jova_topology.sym_flag.item()) == 1
    assert np.array_equal(out.detach().cpu().numpy(), z["out"])                 # max is exact
    (out * torch.from_numpy(z["gout"]).to(device)).sum().backward()
    H.assert_close(x.grad, z["gx"], tol=1e-6, what="dx (tie routing: first maximal index, self first)")


def test_graphpool_all_degrees_with_and_without_the_row_table(device):
    """GraphPool forward + backward on a batch that populates every degree bucket 0..10 (the row table holds four
    neighbours: atoms of degree 5..10 read the rest from the adjacency lists) against the oracle, and bit-equal between
    the table form (what the module runs) and the forms without it."""
    from jova_amd import synth
    from jova_amd.nn import functional as F_
    from jova_amd.nn.layers import GraphPool
    from oracle import nn_ref as R
    rng = np.random.RandomState(65)
    mols = [synth.make_molecule(rng, int(n), deg_cap=4, hub_frac=0.15) for n in rng.randint(14, 60, size=150)]
    mols.append(synth.make_molecule(rng, 1))
    gl_cpu = H.oracle_glist(mols)
    n = gl_cpu[0].shape[0]
    assert all(int(a.shape[0]) > 0 for a in gl_cpu[4:]) and int(gl_cpu[1][0, 1]) > 0, "buckets 0..10 populated"
    torch.manual_seed(9)
    x = torch.randn(n, 64).round(decimals=1)                      # coarse values: plenty of ties
    gout = torch.randn(n, 64)
    xo = x.double().requires_grad_(True)
    ref = R.graph_pool(xo, gl_cpu[1], gl_cpu[4:])
    (ref * gout.double()).sum().backward()
    xd = x.to(device).requires_grad_(True)
    gl = _glist_dev(mols, device, xd)
    out = GraphPool()(gl)
    (out * gout.to(device)).sum().backward()
    assert torch.equal(out.detach().cpu(), ref.detach().float())
    H.assert_close(xd.grad, xo.grad.float(), tol=1e-6, what="dx")
    topo = gl[1]._jova_topology
    assert topo.symmetric and topo.row_table is not None and int((topo.row_table[:, 1] > 4).sum()) > 0
    o2, a1, a2 = torch.empty_like(out), torch.empty(n, 64, dtype=torch.uint8, device=device), \
        torch.empty(n, 64, dtype=torch.uint8, device=device)
    F_.call("jova_graphpool_fwd", F_.P(xd.detach()), n, 64, F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(o2), F_.P(a1),
            F_.P(topo.block_order), None, F_.stream())
    assert torch.equal(o2, out.detach())
    F_.call("jova_graphpool_fwd", F_.P(xd.detach()), n, 64, F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(o2), F_.P(a2),
            None, F_.P(topo.row_table), F_.stream())
    assert torch.equal(a1, a2)
    dy = gout.to(device)
    dx_plain = torch.empty_like(out)
    F_.call("jova_graphpool_bwd", F_.P(dy), F_.P(a1), n, 64, F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(dx_plain),
            F_.P(topo.sym_flag), 1, F_.P(topo.rev_pos), F_.P(topo.block_order), None, F_.stream())
    assert torch.equal(dx_plain, xd.grad)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gather2d_nonsat_vs_reference_golden(device, tag):
    from jova_amd.nn.layers import GraphGather2D
    from jova_amd.nn import functional as F_
    from oracle import nn_ref as R
    z = H.load("op_gather2d_nonsat")
    n_atoms = [int(v) for v in z[tag + ".n_atoms"]]
    x = torch.from_numpy(z[tag + ".x"]).to(device).requires_grad_(True)
    out = GraphGather2D(activation="nonsat")([x, None, None, n_atoms], len(n_atoms))
    _, n_eval = R.nonsat(R.gather2d(torch.from_numpy(z[tag + ".x"]), n_atoms))
    ctrl = F_.NonsatFunction.last_n_eval.cpu().tolist()
    assert ctrl[0] == n_eval and ctrl[2] == 0, "evaluations %s (oracle %d)" % (ctrl, n_eval)
    H.assert_close(out, z[tag + ".out"], tol=1e-5, what="out")
    (out * torch.from_numpy(z[tag + ".gout"]).to(device)).sum().backward()
    H.assert_close(x.grad, z[tag + ".gx"], tol=1e-5, what="dx")
    tanh = GraphGather2D()([x.detach(), None, None, n_atoms], len(n_atoms))
    H.assert_close(tanh, z[tag + ".out_tanh"], tol=1e-6, what="tanh flavour (padding layout)")


def test_nonsat_iteration_cap(device):
    """max_iter is honoured: with ep = 0 the loop runs until the evaluation index exceeds max_iter (max_iter + 2 evaluations)."""
    from jova_amd.nn import functional as F_
    from oracle import nn_ref as R
    x = torch.randn(3000) * 3
    want, n_eval = R.nonsat(x, ep=-1.0, max_iter=5)
    out = F_.NonsatFunction.apply(x.to(device), -1.0, 5)
    assert F_.NonsatFunction.last_n_eval.cpu().tolist()[0] == n_eval == 7
    H.assert_close(out, want, tol=1e-5)


def test_nonsat_barrier_give_up_is_visible(device):
    """A barrier that gives up (workgroups not co-resident) must not hand a torn tensor to training: with the spin
    bound cut to one tick the first arrivals give up, every workgroup follows (collective), the flag is raised and the
    outputs are NaN; NonsatFunction.raise_if_timed_out reports it.  The next call with the default bound is clean."""
    from jova_amd import _lib
    from jova_amd.nn import functional as F_
    from oracle import nn_ref as R
    x = torch.randn(64 * 4096 * 8, generator=torch.Generator().manual_seed(3)) * 2
    xd = x.to(device)
    _lib.lib().jova_nonsat_set_spin_limit(1)
    try:
        out = F_.NonsatFunction.apply(xd, 1e-4, 100)
        ctrl = F_.NonsatFunction.last_n_eval.cpu().tolist()
    finally:
        _lib.lib().jova_nonsat_set_spin_limit(0)
    assert ctrl[2] == 1, "give-up flag not raised: %s" % ctrl
    assert torch.isnan(out).any(), "a barrier that gave up must poison its outputs"
    with pytest.raises(RuntimeError, match="barrier gave up"):
        F_.NonsatFunction.raise_if_timed_out()
    out = F_.NonsatFunction.apply(xd, 1e-4, 100)
    want, n_eval = R.nonsat(x)
    assert F_.NonsatFunction.last_n_eval.cpu().tolist() [0] == n_eval and F_.NonsatFunction.last_n_eval.cpu().tolist()[2] == 0
    F_.NonsatFunction.raise_if_timed_out()
    H.assert_close(out, want, tol=1e-5)


# ------------------------------------------------------------------------------------------------ protein branch
def test_prot2vec_vs_reference_golden(device):
    from jova_amd.nn.models import Prot2Vec, _construct_embedding_indices
    z = H.load("op_prot2vec")
    prof = H.profile(z)
    batch = [("davis", "P%03d" % i) for i in z["batch_prot"]]
    idx = _construct_embedding_indices(batch, prof, device, 60)
    assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), z["idx"])      # index cube bit-exact
    p2v = _load_sd(Prot2Vec(prof, 60, 6), z, "sd.").to(device)
    out = p2v(batch)
    assert np.array_equal(out.detach().cpu().numpy(), z["out"])
    (out * torch.from_numpy(z["gout"]).to(device)).sum().backward()
    H.assert_close(p2v.embedding.weight.grad, z["grad.embedding.weight"], tol=1e-6, what="dE")
    assert float(p2v.embedding.weight.grad[60].abs().max()) == 0.0


def test_pcnn2d_vs_reference_golden(device):
    from jova_amd.nn.models import ProteinCNN2D
    z = H.load("op_pcnn2d")
    net = _load_sd(ProteinCNN2D(dim=6, window=11, num_layers=3), z, "sd.").to(device)
    x = torch.from_numpy(z["x"]).to(device).requires_grad_(True)
    out = net(x)
    H.assert_close(out, z["out"], tol=1e-5, what="out")
    (out * torch.from_numpy(z["gout"]).to(device)).sum().backward()
    H.assert_close(x.grad, z["gx"], tol=1e-5, what="dx")
    for i in range(3):
        H.assert_close(net.lin_kernels[i].weight.grad, z["grad.lin_kernels.%d.weight" % i], tol=1e-5, what="dW")
        H.assert_close(net.lin_kernels[i].bias.grad, z["grad.lin_kernels.%d.bias" % i], tol=1e-5, what="db")


@pytest.mark.parametrize("layers", [2, 3])
def test_pcnn2d_padded_aligned_form_vs_fp64_and_generic(device, layers):
    """ProteinCNN2D at the tuned shape (dim 22 x window 11 = 242 columns, rows a multiple of 128) takes the padded form
    (ProteinCnnFunction: 256-column buffers on the aligned GEMM); outputs and every gradient against fp64 torch, and
    against the layer-by-layer generic form of the same module."""
    from jova_amd.nn import functional as F_
    from jova_amd.nn.models import ProteinCNN2D
    torch.manual_seed(layers)
    net = ProteinCNN2D(dim=22, window=11, num_layers=layers)
    x = torch.randn(5, 128, 242)
    gout = torch.randn(5, 128, 22)
    ws = [k.weight.detach().double().requires_grad_(True) for k in net.lin_kernels]
    bs = [k.bias.detach().double().requires_grad_(True) for k in net.lin_kernels]
    xr = x.double().requires_grad_(True)
    h = xr
    for w, b in zip(ws, bs):
        h = torch.relu(h @ w.t() + b)
    (h * gout.double()).sum().backward()
    net = net.to(device)
    assert F_.ProteinCnnFunction.usable(x, [k.weight for k in net.lin_kernels])
    res = {}
    for form in ("padded", "generic"):
        net.zero_grad()
        xd = x.to(device).requires_grad_(True)
        if form == "generic":
            saved = F_.ProteinCnnFunction.usable
            F_.ProteinCnnFunction.usable = staticmethod(lambda x_, w_: False)
        try:
            out = net(xd)
            (out * gout.to(device)).sum().backward()
        finally:
            if form == "generic":
                F_.ProteinCnnFunction.usable = saved
        res[form] = (out.detach(), xd.grad, [k.weight.grad.clone() for k in net.lin_kernels],
                     [k.bias.grad.clone() for k in net.lin_kernels])
        H.assert_close(out, h.detach().float(), tol=1e-5, what=form + " out")
        H.assert_close(xd.grad, xr.grad.float(), tol=1e-5, what=form + " dx")
        for k in range(layers):
            H.assert_close(res[form][2][k], ws[k].grad.float(), tol=1e-5, what=form + " dW%d" % k)
            H.assert_close(res[form][3][k], bs[k].grad.float(), tol=1e-5, what=form + " db%d" % k)
    assert res["padded"][0].grad_fn is None and res["padded"][0].shape == (5, 128, 22)


# ------------------------------------------------------------------------------------------------ joint attention
def test_joint_attention_vs_reference_golden(device):
    from jova_amd.nn.models import JointAttention
    z = H.load("op_joint_attention")
    ja = _load_sd(JointAttention(d_dims=[40, 12, 6], latent_dim=32, num_heads=2, num_layers=1, dprob=0.,
                                 inner_layer_dim=48), z, "sd.").to(device)
    ja.train()
    xs = [torch.from_numpy(z["x%d" % i]).to(device).requires_grad_(True) for i in range(3)]
    out = ja(xs)
    H.assert_close(out, z["out"], tol=1e-5, what="out")
    (out * torch.from_numpy(z["gout"]).to(device)).sum().backward()
    for i in range(3):
        H.assert_close(xs[i].grad, z["gx%d" % i], tol=2e-5, what="dx%d" % i)
    floor = H.grad_floor(z, "grad.")
    named = dict(ja.named_parameters())
    for k in z.files:
        if k.startswith("grad."):
            # 2e-5: the reference's own fp32 gradients sit up to 2e-5 from the fp64 oracle here (test_oracle_golden)
            H.assert_close(named[k[5:]].grad, z[k], tol=2e-5, what=k, floor=floor)


def test_attention_dropout_is_consistent_between_forward_and_backward(device):
    """With dprob > 0 the mask is regenerated in backward from (seed, offset): a finite-difference check along one
    direction must agree with the analytic gradient when the same mask is replayed."""
    from jova_amd.nn import functional as F_
    torch.manual_seed(11)
    x = torch.randn(4, 1000, device=device)
    F_._DropoutState.offset = 100
    y1 = F_.DropoutFunction.apply(x, 0.3)
    F_._DropoutState.offset = 100
    y2 = F_.DropoutFunction.apply(x, 0.3)
    assert torch.equal(y1, y2)
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.03
    H.assert_close(y1[y1 != 0], (x / 0.7)[y1 != 0], tol=1e-6)
    xg = x.clone().requires_grad_(True)
    F_._DropoutState.offset = 100
    F_.DropoutFunction.apply(xg, 0.3).sum().backward()
    H.assert_close(xg.grad, (y1 != 0).float() / 0.7, tol=1e-6)


def test_softmax_backward_with_fused_dropout_mask(device):
    """jova_softmax_bwd with the dropped probabilities handed in == dropout backward followed by softmax backward."""
    from jova_amd.nn import functional as F_
    torch.manual_seed(5)
    rows, n, ld = 37, 279, 280
    logits = torch.randn(rows, ld, device=device)
    probs = torch.zeros(rows, ld, device=device)
    F_.call("jova_softmax_fwd", F_.P(logits), rows, n, ld, F_.P(probs), F_.stream())
    pd = torch.zeros_like(probs)
    F_.dropout_apply(F_.P(probs), rows, n, ld, 0.3, 1234, 7, F_.P(pd))
    dp = torch.randn(rows, ld, device=device)
    ds = torch.zeros_like(dp)
    F_.call("jova_softmax_bwd", F_.P(probs), F_.P(dp), F_.P(pd), 1.0 / 0.7, rows, n, ld, F_.P(ds), F_.stream())
    p64, d64 = probs[:, :n].double(), dp[:, :n].double()
    dpm = torch.where(pd[:, :n] != 0, d64 / 0.7, torch.zeros_like(d64))
    ref = p64 * (dpm - (p64 * dpm).sum(1, keepdim=True))
    H.assert_close(ds[:, :n], ref.float(), tol=2e-6, what="ds")
    keep = float((pd[:, :n] != 0).float().mean())
    assert abs(keep - 0.7) < 0.03


def test_joint_attention_gradient_with_attention_dropout(device):
    """dprob > 0: the forward keeps the dropped attention probabilities and the backward reuses them (no mask replay).
    Directional finite difference with the masks replayed (same Philox offsets) against the analytic gradient."""
    from jova_amd.nn import functional as F_
    from jova_amd.nn.models import JointAttention
    torch.manual_seed(3)
    ja = JointAttention(d_dims=[10, 6], latent_dim=16, num_heads=2, num_layers=1, dprob=0.25, inner_layer_dim=32).to(device)
    ja.train()
    xs = [torch.randn(5, 8, 10, device=device), torch.randn(7, 8, 6, device=device)]
    g = torch.randn(8, 32, device=device)

    def loss_of(inputs):
        F_._DropoutState.offset = 1000
        return float((ja(inputs) * g).double().sum())

    xg = [x.clone().requires_grad_(True) for x in xs]
    F_._DropoutState.offset = 1000
    (ja(xg) * g).sum().backward()
    dirs = [torch.randn_like(x) for x in xs]
    analytic = sum(float((x.grad.double() * d.double()).sum()) for x, d in zip(xg, dirs))
    eps = 2e-3
    with torch.no_grad():
        fd = (loss_of([x + eps * d for x, d in zip(xs, dirs)]) - loss_of([x - eps * d for x, d in zip(xs, dirs)])) / (2 * eps)
    assert abs(fd - analytic) <= 3e-2 * max(abs(analytic), 1.0), (fd, analytic)


@pytest.mark.parametrize("opt", ["sgd", "sgd_momentum", "adamax"])
def test_optimizer_step_vs_torch_optim(device, opt):
    """jova_sgd_step / jova_adamax_step against torch.optim on the host (the reference builds its optimizer from
    torch.optim: proj/dti/train_joint_attn.py:485-486 steps it), four steps with weight decay on a flat buffer whose
    length is not a multiple of the vector width; grad_scale = 1/2 stands for the mean over two ranks."""
    from jova_amd.nn import functional as F_
    torch.manual_seed(11)
    n, lr, wd = 100003, 0.05, 0.01
    p0 = torch.randn(n)
    grads = [torch.randn(n) * (0.5 + k) for k in range(4)]
    ref = torch.nn.Parameter(p0.clone())
    if opt == "adamax":
        o = torch.optim.Adamax([ref], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    else:
        o = torch.optim.SGD([ref], lr=lr, weight_decay=wd, momentum=0.9 if opt == "sgd_momentum" else 0.0)
    p = p0.clone().to(device)
    m1, m2 = torch.zeros(n, device=device), torch.zeros(n, device=device)
    for k, g in enumerate(grads):
        ref.grad = g.clone()
        o.step()
        g2 = (2.0 * g).to(device)                       # the sum over two ranks; the kernel scales by 1/2
        if opt == "adamax":
            F_.call("jova_adamax_step", F_.P(p), F_.P(g2), F_.P(m1), F_.P(m2), n, lr, 0.9, 0.999, 1e-8, wd, 0.5, k + 1,
                    None, 0, F_.stream())
        else:
            mom = 0.9 if opt == "sgd_momentum" else 0.0
            F_.call("jova_sgd_step", F_.P(p), F_.P(g2), F_.P(m1) if mom else None, n, lr, wd, mom, 0.5, int(k == 0),
                    None, 0, F_.stream())
    H.assert_close(p, ref.detach(), tol=2e-6, what="params after 4 %s steps" % opt)


def test_graph_kernels_on_empty_and_single_atom_batches(device):
    """Edge cases of the batch graph through the C ABI: no atoms at all (every launch is skipped, status OK), and one
    molecule of one atom (degree 0 only: the self weights of bucket 0, no neighbour term; pooling is the identity)."""
    from jova_amd.nn import functional as F_
    from jova_amd.nn.layers import GraphConvLayer, GraphPool
    torch.manual_seed(2)
    conv = GraphConvLayer(64, 64, activation=None).to(device)
    pool = GraphPool()

    def gconv_input(n):
        ds = torch.zeros(11, 2, dtype=torch.int64)
        ds[0, 1] = n                                             # all atoms in the degree-0 bucket
        adj = [torch.zeros(0, d, dtype=torch.int32, device=device) for d in range(1, 11)]
        return [torch.randn(n, 64, device=device, requires_grad=True), ds.to(device), None, [n] if n else []] + adj

    data = gconv_input(0)
    y = conv(data)
    assert tuple(y.shape) == (0, 64)
    yp = pool([y] + data[1:])
    assert tuple(yp.shape) == (0, 64)
    yp.sum().backward()                                          # backward of nothing: zero weight gradients
    assert all(l.weight.grad is None or float(l.weight.grad.abs().max()) == 0.0 for l in conv.linear_layers)
    conv.zero_grad()
    data = gconv_input(1)
    y = conv(data)
    lin = conv.linear_layers[20]                                 # the degree-0 self weights (layers.py:330-336)
    want = data[0].detach() @ lin.weight.detach().t() + lin.bias.detach()
    H.assert_close(y.detach(), want, tol=1e-5, what="degree-0 atom")
    yp = pool([y] + data[1:])
    assert torch.equal(yp.detach(), y.detach())
    yp.sum().backward()
    torch.cuda.synchronize()
    H.assert_close(data[0].grad, lin.weight.detach().sum(0, keepdim=True), tol=1e-5, what="dx of the degree-0 atom")
    H.assert_close(lin.weight.grad, data[0].detach().expand(64, 64).contiguous(), tol=1e-5, what="dW of bucket 0")
    assert all(float(l.weight.grad.abs().max()) == 0.0 for l in conv.linear_layers[:20] if l.weight.grad is not None)


# ------------------------------------------------------------------------------------------------ end to end
def _e2e_setup(device):
    from jova_amd.data.collate import DtiBatch, process_gconv_view_data
    from jova_amd.feat.mol_graphs import ConvMol
    from jova_amd.train import create_integrated_net
    z = H.load("e2e_small")
    hp = H.hp(z)
    hparams = {"attn_heads": hp["heads"], "attn_layers": hp["attn_layers"], "lin_dims": hp["lin_dims"], "output_dim": 1,
               "latent_dim": hp["latent"], "dprob": hp["dprob"], "optimizer": "sgd",
               "optimizer__global__lr": hp["lr"], "optimizer__global__weight_decay": hp["wd"],
               "prot": {"model_types": ["pcnn2d"], "vocab_size": hp["vocab"], "window": hp["window"],
                        "pcnn_num_layers": hp["pcnn_layers"], "embedding_dim": hp["emb"]},
               "gconv": {"dim": hp["gconv_dim"]}, "ecfp8": {"dim": hp["ecfp_dim"]}}
    prof = H.profile(z)
    net = create_integrated_net(hparams, prof)
    net.load_state_dict(H.sd(z, "sd0."))
    net = net.to(device).train()
    mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(H.unpack_mols(z))]
    glist = process_gconv_view_data(mols, device)
    names = sorted(prof)
    batch = DtiBatch(torch.from_numpy(z["ecfp"]).to(device), glist, [("davis", names[i]) for i in z["prot_ids"]],
                     torch.from_numpy(z["y"]).float().to(device), torch.from_numpy(z["w"]).float().to(device))
    return z, hp, hparams, net, batch


def test_e2e_training_step_vs_reference_golden(device):
    from jova_amd.train import Trainer
    z, hp, hparams, net, batch = _e2e_setup(device)
    # the collated index tensors are the reference's, bit for bit
    ref_gl = H.oracle_glist(H.unpack_mols(z))
    for got, want in zip(batch.glist, ref_gl):
        if torch.is_tensor(want):
            assert got.dtype == want.dtype and torch.equal(got.cpu(), want)
        else:
            assert list(got) == list(want)
    tr = Trainer(net, hparams)
    tr.flat.zero_grad()
    loss, out = tr.forward_loss(batch)
    loss.backward()
    H.assert_close(out, z["step0.out"], tol=2e-5, what="outputs")
    assert abs(loss.item() - float(z["step0.loss"])) <= 1e-5 * abs(float(z["step0.loss"]))
    floor = H.grad_floor(z, "step0.grad.")
    named = dict(net.named_parameters())
    for k in z.files:
        if k.startswith("step0.grad."):
            # 5e-4: the tiny fixture model (batch 8, BN/LN chains) is ill-conditioned; the reference's own fp32
            # gradients sit up to 1.3e-4 from the fp64 oracle on these tensors (tests/test_oracle_golden.py)
            H.assert_close(named[k[len("step0.grad."):]].grad, z[k], tol=5e-4, what=k, floor=floor)
    # optimizer step + second step: loss of step 1 pins the SGD update and the BatchNorm running statistics
    net.load_state_dict(H.sd(z, "sd0."))
    tr.step(batch)
    sd = net.state_dict()
    for k in z.files:
        if k.startswith("step0.sd."):
            name = k[len("step0.sd."):]
            if name.endswith("num_batches_tracked"):
                assert int(sd[name]) == int(z[k])
            elif "running" in name:
                # batch statistics of 8 rows: the column mean is a cancelling sum (|mean| ~ 0.03 |x|), so input
                # differences of 1e-6 show up as a few 1e-5 of the running mean
                H.assert_close(sd[name], z[k], tol=1e-4, what=name)
            else:
                gmax = max(float(np.abs(z["step0.grad." + name]).max()), floor)
                bound = 1e-6 * float(np.abs(z[k]).max()) + hp["lr"] * 5e-4 * gmax
                assert float(np.abs(sd[name].cpu().numpy() - z[k]).max()) <= bound, name
    loss1, out1 = tr.step(batch)
    assert abs(loss1.item() - float(z["step1.loss"])) <= 2e-5 * abs(float(z["step1.loss"]))
    # the step-1 outputs inherit the gradient differences of step 0 (bounded above at 5e-4 on this ill-conditioned
    # fixture) through the updated weights
    H.assert_close(out1, z["step1.out"], tol=5e-4, what="outputs after one SGD step")


# ------------------------------------------------------------------------------------------------ full sizes
def test_overlapped_gradient_sync_on_rccl_single_rank(device):
    """The data-parallel step's two-piece gradient all-reduce on the real backend (RCCL, one rank: sums are the
    identity): the hook fires during backward, the asynchronous collective is ordered behind the kernels that wrote
    its slice, and the gradients equal those of the plain step."""
    import os
    import socket
    import torch.distributed as dist
    from jova_amd.train import OverlappedGradSync, Trainer, encoder_param_split
    z, hp, hparams, net, batch = _e2e_setup(device)
    tr = Trainer(net, hparams)
    tr.flat.zero_grad()
    loss, _ = tr.forward_loss(batch)
    loss.backward()
    want = tr.flat.grad.clone()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        split = encoder_param_split(net, tr.flat)
        assert split is not None and 0 < split < tr.flat.numel
        tr.sync = OverlappedGradSync(tr.flat, split, dist.group.WORLD)
        tr.flat.zero_grad()
        loss2, _ = tr.forward_loss(batch)
        loss2.backward()
        assert tr.sync.work is not None, "the hook on the encoders' outputs did not fire"
        tr.sync.finish()
        torch.cuda.synchronize()
        # (float atomics in the split-K / dW kernels: two runs of the same step differ by a few 1e-6 of the scale)
        assert float((tr.flat.grad - want).abs().max()) <= 2e-5 * float(want.abs().max())
    finally:
        dist.destroy_process_group()


def _stage_grads(dtype, sd0, hp, pool, mi, pi, y, w, views_in=None, ja_in=None):
    """Oracle (dtype) of the tuned model, stage by stage: graph / protein view encoders, joint attention, FC head.  Each
    stage runs from GIVEN inputs when views_in / ja_in are handed in (the fp64 run's stage outputs), so stage errors do
    not compound.  -> dict(views, ja, out, loss, d_views, d_ja, grads per parameter)."""
    from oracle import mol_graphs as og
    from oracle import nn_ref as R
    b = og.agglomerate([og.SortedMol(*pool.mols[i]) for i in mi])
    gl = og.gconv_input_list(b)
    glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + \
            [torch.from_numpy(a) for a in gl[4:]]
    idx = og.embedding_indices([pool.prot_names[i] for i in pi], pool.protein_profile, fill_val=pool.vocab)
    p = {k: (v.to(dtype).requires_grad_("running" not in k) if v.is_floating_point() else v) for k, v in sd0.items()}
    n_pcnn, n_lin = hp["prot"]["pcnn_num_layers"], len(hp["lin_dims"])
    v_e = torch.from_numpy(pool.ecfp[mi]).to(dtype).unsqueeze(0)
    v_g, _ = R.gconv_net(glist, p, prefix="0.models.1.0.")
    v_p = R.pcnn2d(R.prot2vec(torch.from_numpy(idx), p["0.models.2.0.embedding.weight"]),
                   [p["0.models.2.1.lin_kernels.%d.weight" % i] for i in range(n_pcnn)],
                   [p["0.models.2.1.lin_kernels.%d.bias" % i] for i in range(n_pcnn)])
    src = [v_e, v_g, v_p] if views_in is None else views_in
    views = [t.detach().to(dtype).requires_grad_(True) for t in src]
    ja = R.joint_attention(views, p, hp["attn_heads"], prefix="1.")
    ja_leaf = (ja if ja_in is None else ja_in).detach().to(dtype).requires_grad_(True)
    x, k = ja_leaf, 2
    for _ in range(n_lin):
        x = R.linear(x, p["%d.weight" % k], p["%d.bias" % k])
        x = torch.relu(R.batch_norm_train(x, p["%d.weight" % (k + 1)], p["%d.bias" % (k + 1)]))
        k += 4
    out = R.linear(x, p["%d.weight" % k], p["%d.bias" % k])
    loss = R.mse_loss(out, torch.from_numpy(w).to(dtype), torch.from_numpy(y).to(dtype))
    return dict(p=p, glist=gl, v_g=v_g, v_p=v_p, views=views, ja=ja, ja_leaf=ja_leaf, out=out, loss=loss)


def _stage_backward(st, d_ja=None, d_views=None):
    """Backward of _stage_grads' stages, each from a GIVEN upstream gradient when one is handed in."""
    st["loss"].backward()
    st["ja"].backward(st["ja_leaf"].grad if d_ja is None else d_ja.to(st["ja"].dtype))
    dv = [v.grad for v in st["views"]] if d_views is None else [t.to(st["v_g"].dtype) for t in d_views]
    st["v_g"].backward(dv[1])
    st["v_p"].backward(dv[2])
    return {k: v.grad.detach().double() for k, v in st["p"].items()
            if torch.is_tensor(v) and v.requires_grad and v.grad is not None}


def test_davis_full_size_stages_vs_oracle(device):
    """BASELINE.json config 2, stage by stage: the tuned Davis model, B = 128, every stage (graph view, protein view,
    joint attention, FC head) run on the GPU from the fp64 ORACLE's stage inputs and upstream gradients, so that ReLU
    mask flips of one stage do not leak into the next.  (End to end, against the imported reference, with bounds from
    the reference's own fp32 run: tests/test_full_size_parity.py.)

    Bounds are derived, not picked: the fp32 oracle (torch CPU arithmetic, as the reference's fp32 run) goes through
    the same stages from the same fp64 inputs, and every HIP tensor is held to
        err(hip, oracle64) <= max(1e-5, 2 * err(oracle32, oracle64))          err = max|a - b| / max|oracle64|,
    with err(oracle32, oracle64) of a parameter gradient taken no smaller than its stage's whole-gradient L2 distance.
    Gradients that are analytically zero (< 1e-9 of the largest) are judged on 1e-5 of the largest gradient."""
    from jova_amd import synth
    from jova_amd.data.collate import collate_pairs
    from jova_amd.feat.mol_graphs import ConvMol
    from jova_amd.nn import functional as F_
    from jova_amd.train import TUNED_DAVIS_HPARAMS, Trainer, create_integrated_net
    hp = dict(TUNED_DAVIS_HPARAMS)
    hp["dprob"] = 0.0
    pool = synth.SyntheticPool("davis", seed=8)
    cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
    torch.manual_seed(8)
    net = create_integrated_net(hp, pool.protein_profile)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(device).train()
    named = dict(net.named_parameters())
    mi, pi, y, w = pool.sample_pairs(np.random.RandomState(64), 128)
    batch = collate_pairs(pool, cms, mi, pi, y, w, device, net[0].models[2][0])
    tr = Trainer(net, hp)

    torch.set_num_threads(8)
    s64 = _stage_grads(torch.float64, sd0, hp, pool, mi, pi, y, w)
    g64 = _stage_backward(s64)
    d_ja64 = s64["ja_leaf"].grad.detach()
    d_views64 = [v.grad.detach() for v in s64["views"]]
    s32 = _stage_grads(torch.float32, sd0, hp, pool, mi, pi, y, w, views_in=s64["views"], ja_in=s64["ja"])
    g32 = _stage_backward(s32, d_ja=d_ja64, d_views=d_views64)
    for got, want in zip(batch.glist, s64["glist"]):            # index tensors bit-exact at full size
        if isinstance(want, np.ndarray):
            assert np.array_equal(got.cpu().numpy(), want)
    vocab = hp["prot"]["vocab_size"]
    for g in (g64, g32):
        g["0.models.2.0.embedding.weight"][vocab] = 0            # the padding row is never updated
    gmax = max(float(g.abs().max()) for g in g64.values())

    def f32(t):
        return t.detach().float().to(device)

    def held(got, ref32, ref64, what, floor=0.0):
        e_ref = H.rel_err(ref32.detach().double().numpy(), ref64.detach().double().numpy())
        e = H.rel_err(got.detach().cpu().double().numpy(), ref64.detach().double().numpy())
        bound = max(1e-5, 2 * max(e_ref, floor))
        assert e <= bound, "%s: err %.3e > %.3e (fp32 oracle: %.3e)" % (what, e, bound, e_ref)

    def check(prefixes):
        keys = [k for k in g64 if any(k.startswith(q) for q in prefixes)]
        assert keys
        e_stage = (sum(float(((g32[k] - g64[k]) ** 2).sum()) for k in keys) /
                   sum(float((g64[k] ** 2).sum()) for k in keys)) ** 0.5
        for k in keys:
            if float(g64[k].abs().max()) < 1e-9 * gmax:
                assert float(named[k].grad.abs().max()) <= 1e-5 * gmax, k
            else:
                held(named[k].grad, g32[k], g64[k], k, floor=e_stage)

    # FC head
    tr.flat.zero_grad()
    xg = f32(s64["ja"]).requires_grad_(True)
    h = xg
    for i in range(2, len(net)):
        h = net[i](h)
    F_.MSELossFunction.apply(h, batch.w, batch.y).backward()
    held(h, s32["out"], s64["out"], "outputs from the oracle's joint-attention output")
    held(xg.grad, s32["ja_leaf"].grad, d_ja64, "d loss / d joint-attention output")
    check(["%d." % i for i in range(2, len(net))])
    # joint attention
    tr.flat.zero_grad()
    vg = [f32(v).requires_grad_(True) for v in s64["views"]]
    ja_hip = net[1](vg)
    held(ja_hip, s32["ja"], s64["ja"], "joint-attention output")
    ja_hip.backward(f32(d_ja64))
    # the view gradients: bound = twice the largest distance any fp32 execution on record shows for them -- the fp32
    # oracle here, and the imported reference's JointAttention with its summation order changed (thread counts; the
    # latent dimension re-indexed: stage_noise_davis_b128.npz, tests/test_full_size_parity.py explains the two families)
    noise = H.load("stage_noise_davis_b128")
    for i in range(3):
        e_ref = max(float(noise["d_view%d" % i].max()), float(noise["perm.d_view%d" % i].max()))
        e_o32 = H.rel_err(s32["views"][i].grad.detach().double().numpy(), d_views64[i].detach().double().numpy())
        e = H.rel_err(vg[i].grad.detach().cpu().double().numpy(), d_views64[i].detach().double().numpy())
        bound = max(1e-5, 2 * max(e_ref, e_o32))
        assert e <= bound, "d loss / d view %d: err %.3e > %.3e (reference fp32 scatter %.3e, fp32 oracle %.3e)" % (
            i, e, bound, e_ref, e_o32)
    check(["1."])
    # graph view and protein view
    tr.flat.zero_grad()
    got_views = net[0](batch.model_input())
    H.assert_close(got_views[1], s64["v_g"].detach(), tol=1e-5, what="graph view")
    H.assert_close(got_views[2], s64["v_p"].detach(), tol=1e-5, what="protein view")
    got_views[1].backward(f32(d_views64[1]))
    got_views[2].backward(f32(d_views64[2]))
    check(["0."])


def test_config4_graph_stack_vs_oracle(device):
    """BASELINE.json config 4 at full size (4096 molecules, ~246 k atoms, degrees up to 10): conv -> BN+ReLU -> pool
    -> conv -> pool forward against the fp32 oracle on the host, plus size-independent properties of the backward:
    the pool gradient conserves mass per column and the gather and scatter forms of it agree."""
    from jova_amd import synth
    from jova_amd.data.collate import process_gconv_view_data
    from jova_amd.feat.mol_graphs import ConvMol
    from jova_amd.nn.layers import BatchNorm1d, GraphConvLayer, GraphPool
    from jova_amd.nn import functional as F_
    from oracle import nn_ref as R
    pool = synth.SyntheticPool("graph4k", seed=8)
    mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
    glist = process_gconv_view_data(mols, device)
    assert all(int(a.shape[0]) > 0 for a in glist[4:]), "every degree bucket 1..10 populated"
    torch.manual_seed(1)
    c1, bn, c2, gp = GraphConvLayer(75, 64), BatchNorm1d(64), GraphConvLayer(64, 64), GraphPool()
    cpu_l = [t.cpu() if torch.is_tensor(t) else t for t in glist]
    w1 = [l.weight.detach() for l in c1.linear_layers]
    b1 = [l.bias.detach() for l in c1.linear_layers]
    w2 = [l.weight.detach() for l in c2.linear_layers]
    b2 = [l.bias.detach() for l in c2.linear_layers]
    torch.set_num_threads(8)
    r1 = R.graph_conv(cpu_l[0], cpu_l[1], cpu_l[4:], w1, b1)
    r2 = R.graph_pool(torch.relu(R.batch_norm_train(r1, bn.weight.detach(), bn.bias.detach())), cpu_l[1], cpu_l[4:])
    r3 = R.graph_pool(R.graph_conv(r2, cpu_l[1], cpu_l[4:], w2, b2), cpu_l[1], cpu_l[4:])
    c1, bn, c2 = c1.to(device), bn.to(device).train(), c2.to(device)
    o1 = c1(glist)
    bn.fused_relu = True
    o2 = gp([bn(o1)] + glist[1:])
    x3 = o2.detach().requires_grad_(True)
    o3 = gp([c2([x3] + glist[1:])] + glist[1:])
    H.assert_close(o1, r1, tol=1e-5, what="conv1")
    H.assert_close(o2, r2, tol=2e-5, what="bn+relu+pool")
    H.assert_close(o3, r3, tol=2e-5, what="conv2+pool")
    # pool backward: every dy element is routed to exactly one atom -> column sums are conserved
    topo = glist[1]._jova_topology
    assert topo._sym_host is True and int(topo.sym_flag.item()) == 1      # the collator's host check == the device flag
    assert topo.symmetric
    dy = torch.randn_like(o2)
    arg = torch.empty(o2.shape, dtype=torch.uint8, device=device)
    tmp = torch.empty_like(o2)
    F_.call("jova_graphpool_fwd", F_.P(o1), o1.shape[0], 64, F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(tmp), F_.P(arg),
            None, None, F_.stream())
    # the XCD-aware block order is a permutation of the 16-row blocks and changes nothing in the results
    order = topo.block_order
    assert torch.equal(torch.sort(order.long()).values.cpu(), torch.arange((o1.shape[0] + 15) // 16))
    tmp2, arg2 = torch.empty_like(tmp), torch.empty_like(arg)
    F_.call("jova_graphpool_fwd", F_.P(o1), o1.shape[0], 64, F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(tmp2), F_.P(arg2),
            F_.P(order), None, F_.stream())
    assert torch.equal(tmp, tmp2) and torch.equal(arg, arg2)
    # the row table (row, degree, first neighbours of every row in launch order) changes nothing either; built from
    # the batch's block order and, a second time, in plain row order
    table = topo.row_table
    plain = torch.empty_like(table)
    F_.call("jova_row_table", o1.shape[0], F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(topo.rev_pos), None, F_.P(plain),
            F_.stream())
    assert torch.equal(torch.sort(table[:, 0]).values, torch.sort(plain[:, 0]).values)
    live = plain[plain[:, 0] >= 0]
    assert torch.equal(live[:, 0].cpu(), torch.arange(o1.shape[0], dtype=torch.int32))
    for tb in (table, plain):
        tmp3, arg3 = torch.full_like(tmp, float("nan")), torch.full_like(arg, 77)
        F_.call("jova_graphpool_fwd", F_.P(o1), o1.shape[0], 64, F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(tmp3),
                F_.P(arg3), None, F_.P(tb), F_.stream())
        assert torch.equal(tmp, tmp3) and torch.equal(arg, arg3), "pool forward: the row table changes the result"
    dx_g, dx_s, dx_r = torch.empty_like(o2), torch.empty_like(o2), torch.empty_like(o2)
    F_.call("jova_graphpool_bwd", F_.P(dy), F_.P(arg), o1.shape[0], 64, F_.P(topo.deg_slice), topo.adj_ptrs,
            F_.P(dx_g), F_.P(topo.sym_flag), 1, None, None, None, F_.stream())
    F_.call("jova_graphpool_bwd", F_.P(dy), F_.P(arg), o1.shape[0], 64, F_.P(topo.deg_slice), topo.adj_ptrs,
            F_.P(dx_r), F_.P(topo.sym_flag), 1, F_.P(topo.rev_pos), None, None, F_.stream())
    dx_o = torch.empty_like(o2)
    F_.call("jova_graphpool_bwd", F_.P(dy), F_.P(arg), o1.shape[0], 64, F_.P(topo.deg_slice), topo.adj_ptrs,
            F_.P(dx_o), F_.P(topo.sym_flag), 1, F_.P(topo.rev_pos), F_.P(order), None, F_.stream())
    assert torch.equal(dx_o, dx_r), "pool backward: the block order changes the result"
    for tb in (table, plain):
        dx_t = torch.full_like(o2, float("nan"))
        F_.call("jova_graphpool_bwd", F_.P(dy), F_.P(arg), o1.shape[0], 64, F_.P(topo.deg_slice), topo.adj_ptrs,
                F_.P(dx_t), F_.P(topo.sym_flag), 1, F_.P(topo.rev_pos), None, F_.P(tb), F_.stream())
        assert torch.equal(dx_t, dx_r), "pool backward: the row table changes the result"
    F_.call("jova_graphpool_bwd", F_.P(dy), F_.P(arg), o1.shape[0], 64, F_.P(topo.deg_slice), topo.adj_ptrs,
            F_.P(dx_s), None, 0, None, None, None, F_.stream())
    H.assert_close(dx_g, dx_s, tol=1e-6, what="pool backward: gather form == atomic scatter form")
    assert torch.equal(dx_g, dx_r), "pool backward: precomputed reverse positions change the result"
    assert int((topo.rev_pos == 0xFF).sum()) == 0, "symmetric graph: every edge has its reverse"
    H.assert_close(dx_g.double().sum(0), dy.double().sum(0), tol=1e-6, what="pool backward conserves column sums")
    # ---- backward at full size against the fp64 oracle (autograd of oracle/nn_ref.py), bounds derived from the fp32
    # oracle on the same inputs: err(hip, o64) <= max(1e-5, 2 * err(o32, o64)) per tensor
    g3 = torch.randn(o3.shape, generator=torch.Generator().manual_seed(4))
    g1 = torch.randn(o1.shape, generator=torch.Generator().manual_seed(5))
    for q in list(c1.parameters()) + list(c2.parameters()):
        q.grad = None
    (o3 * g3.to(device)).sum().backward()
    (o1 * g1.to(device)).sum().backward()

    def oracle_bwd(dtype):
        x = o2.detach().cpu().to(dtype).requires_grad_(True)
        W2 = [t.to(dtype).requires_grad_(True) for t in w2]
        B2 = [t.to(dtype).requires_grad_(True) for t in b2]
        W1 = [t.to(dtype).requires_grad_(True) for t in w1]
        B1 = [t.to(dtype).requires_grad_(True) for t in b1]
        (R.graph_pool(R.graph_conv(x, cpu_l[1], cpu_l[4:], W2, B2), cpu_l[1], cpu_l[4:]) * g3.to(dtype)).sum().backward()
        (R.graph_conv(cpu_l[0], cpu_l[1], cpu_l[4:], W1, B1) * g1.to(dtype)).sum().backward()
        return dict(x=x.grad, w2=[t.grad for t in W2], b2=[t.grad for t in B2], w1=[t.grad for t in W1],
                    b1=[t.grad for t in B1])
    r64, r32 = oracle_bwd(torch.float64), oracle_bwd(torch.float32)

    def held(got, a32, a64, what):
        e_ref = H.rel_err(a32.double().numpy(), a64.numpy())
        e = H.rel_err(got.detach().cpu().double().numpy(), a64.numpy())
        assert e <= max(1e-5, 2 * e_ref), "%s: err %.3e, fp32 oracle %.3e" % (what, e, e_ref)
    held(x3.grad, r32["x"], r64["x"], "config 4: conv2+pool dX")
    for k in range(21):
        held(c2.linear_layers[k].weight.grad, r32["w2"][k], r64["w2"][k], "config 4: conv2 dW[%d]" % k)
        held(c2.linear_layers[k].bias.grad, r32["b2"][k], r64["b2"][k], "config 4: conv2 db[%d]" % k)
        held(c1.linear_layers[k].weight.grad, r32["w1"][k], r64["w1"][k], "config 4: conv1 dW[%d]" % k)
        held(c1.linear_layers[k].bias.grad, r32["b1"][k], r64["b1"][k], "config 4: conv1 db[%d]" % k)
    # pool backward alone, element-wise against autograd of the oracle's pool (exact routing: 1e-6)
    xo = o1.detach().cpu().double().requires_grad_(True)
    (R.graph_pool(xo, cpu_l[1], cpu_l[4:]) * dy.cpu().double()).sum().backward()
    H.assert_close(dx_r, xo.grad, tol=1e-6, what="config 4: pool backward vs oracle")
    # gather2d + nonsat at this size (g = 128): forward, evaluation count and backward
    from jova_amd.nn.layers import GraphGather2D
    xg = torch.randn(o1.shape[0], 128, generator=torch.Generator().manual_seed(6))
    gg = None
    xd = xg.to(device).requires_grad_(True)
    og_ = GraphGather2D(activation="nonsat")([xd] + glist[1:], len(glist[3]))
    gg = torch.randn(og_.shape, generator=torch.Generator().manual_seed(7))
    (og_ * gg.to(device)).sum().backward()
    xc = xg.double().requires_grad_(True)
    want, n_eval = R.nonsat(R.gather2d(xc, glist[3]))
    (want * gg.double()).sum().backward()
    assert F_.NonsatFunction.last_n_eval.cpu().tolist()[0] == n_eval
    H.assert_close(og_, want.detach(), tol=1e-5, what="config 4: gather2d + nonsat")
    H.assert_close(xd.grad, xc.grad, tol=1e-5, what="config 4: gather2d + nonsat backward")


@pytest.mark.parametrize("shape,n_mols", [("davis", 64), ("graph4k", 300)])
def test_collator_host_images_equal_device_kernels(device, shape, n_mols):
    """The native collator ships two derived images in its one upload: the fp32 copy of the atom features and the
    reverse edge positions.  They must be what the device kernels (jova_atoms_to_f32, jova_adj_reverse_pos) make."""
    from jova_amd import synth
    from jova_amd.data.collate import gconv_input_list
    from jova_amd.feat.mol_graphs import ConvMol
    from jova_amd.nn import functional as F_
    pool = synth.SyntheticPool(shape, seed=3, n_mols=n_mols, n_prots=2)
    mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
    gl = gconv_input_list(mols, device)
    topo = gl[1]._jova_topology
    n, f = gl[0].shape
    ld = F_._atoms_ld(f)
    want = torch.empty((n, ld), device=device)
    F_.call("jova_atoms_to_f32", F_.P(gl[0]), n, f, ld, F_.P(want), F_.stream())
    assert torch.equal(topo.atoms_f32(gl[0]), want)
    e = sum(int(a.numel()) for a in gl[4:])
    rp = torch.empty(e, dtype=torch.uint8, device=device)
    F_.call("jova_adj_reverse_pos", n, F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(rp), F_.stream())
    assert topo.symmetric and torch.equal(topo.rev_pos[:e], rp)
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    F_.call("jova_adj_symmetry", n, F_.P(topo.deg_slice), topo.adj_ptrs, F_.P(flag), F_.stream())
    assert int(flag.item()) == 1 == int(topo.sym_flag.item())


def test_entry_points_capture_into_a_hip_graph(device):
    """the entry points only enqueue kernels / memsets on the given stream.  Three of them (an
    aligned GEMM with its in-stream memset, the graph pooling, the allocation-free nonsat) are captured into ONE HIP
    graph, together with a BatchNorm statistics pass (whose scratch is zeroed in-stream); three replays on changed inputs must
    equal the eager calls.  (First version of this test: the nonsat call's two small hipMemsetAsync did not take effect on
    every replay as memset nodes -- stale barrier words, NaN-filled output; the library now zeroes with kernels.)"""
    from jova_amd import synth
    from jova_amd._lib import P, call, stream
    from jova_amd.nn import functional as F_
    rng = np.random.RandomState(8)
    mols = [synth.make_molecule(rng, int(n)) for n in rng.randint(8, 30, size=40)]
    gl = _glist_dev(mols, device)
    topo = F_.topology_of(gl)
    n = gl[0].shape[0]
    x = torch.randn(n, 64, device=device)
    pooled, arg = torch.empty(n, 64, device=device), torch.empty(n, 64, dtype=torch.uint8, device=device)
    a, b, bias = torch.randn(128, 1024, device=device), torch.randn(512, 1024, device=device), torch.randn(512, device=device)
    c = torch.empty(128, 512, device=device)
    ns_out, ns_df = torch.empty(n * 64, device=device), torch.empty(n * 64, device=device)
    nev = torch.zeros(3, dtype=torch.int32, device=device)
    stats = torch.empty(2, 64, dtype=torch.float64, device=device)

    def run():
        st = stream()
        call("jova_colstats", P(x), n, 64, P(stats), st)
        F_.gemm(F_._addr(a), 1024, 1, F_._addr(b), 1, 1024, F_._addr(c), 512, 128, 512, 1024, bias=P(bias))
        call("jova_graphpool_fwd", P(x), n, 64, P(topo.deg_slice), topo.adj_ptrs, P(pooled), P(arg), None, None, st)
        call("jova_nonsat_fwd_noalloc", P(pooled), n * 64, 1e-4, 100, P(ns_out), P(ns_df), P(nev), st)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()                                                    # warm-up outside the capture (lazy module loading)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    for trial in range(3):
        x.copy_(torch.randn(n, 64, device=device) * (trial + 1))
        a.copy_(torch.randn(128, 1024, device=device))
        graph.replay()
        torch.cuda.synchronize()
        got = (c.clone(), pooled.clone(), ns_out.clone(), stats.clone())
        assert int(nev[2].item()) == 0 and not torch.isnan(got[2]).any()
        run()
        torch.cuda.synchronize()
        assert torch.equal(got[1], pooled) and torch.equal(got[2], ns_out)
        assert torch.allclose(got[3], stats, rtol=1e-12, atol=1e-9)
        H.assert_close(got[0], c, tol=2e-6, what="captured gemm")   # (split-K partial sums: atomic order)


def test_new_entry_points_on_empty_and_ragged_inputs(device):
    """Edge cases of the round-2 entry points: zero rows / zero steps are no-ops that return cleanly, out-of-range indices
    of the row gather yield NaN rows and are skipped by the scatter, copy2d pads and crops."""
    from jova_amd._lib import P, call, stream
    from jova_amd.nn import functional as F_
    st = stream()
    e = torch.empty(0, 8, device=device)
    idx0 = torch.empty(0, dtype=torch.int64, device=device)
    assert F_.RowGatherFunction.apply(torch.randn(5, 8, device=device), idx0).shape == (0, 8)
    out = F_.RowScatterAddFunction.apply(e, idx0, 4)
    assert out.shape == (4, 8) and float(out.abs().max()) == 0.0
    src = torch.arange(12, dtype=torch.float32, device=device).view(3, 4)
    g = F_.RowGatherFunction.apply(src, torch.tensor([2, 0, 7, -1], device=device))
    assert torch.equal(g[:2], src[[2, 0]]) and torch.isnan(g[2:]).all()
    acc = torch.zeros(2, 4, device=device)
    call("jova_scatter_add_rows", P(src), P(torch.tensor([1, 5, 1], device=device)), 3, 4, 2, P(acc), st)
    assert torch.equal(acc[1], src[0] + src[2]) and float(acc[0].abs().max()) == 0.0
    # copy2d: pad, crop, accumulate
    dst = torch.full((4, 6), 9.0, device=device)
    F_.copy2d(src, 3, 4, 4, dst, 4, 6, 6)
    assert torch.equal(dst[:3, :4], src) and float(dst[3].abs().max()) == 0.0 and float(dst[:, 4:].abs().max()) == 0.0
    crop = torch.ones(2, 3, device=device)
    F_.copy2d(dst, 2, 3, 6, crop, 2, 3, 3, accumulate=True)
    assert torch.equal(crop, src[:2, :3] + 1.0)
    # LSTM: zero steps / zero batch
    w_ih, w_hh = torch.randn(8, 5, device=device), torch.randn(8, 2, device=device)
    b = torch.zeros(8, device=device)
    assert F_.LstmFunction.apply(torch.empty(0, 3, 5, device=device), w_ih, w_hh, b, b, True).shape == (0, 3, 2)
    assert F_.LstmFunction.apply(torch.empty(4, 0, 5, device=device), w_ih, w_hh, b, b, True).shape == (4, 0, 2)
    # gaussian histogram rows sum to one; embedding of nothing
    gh = F_.GaussHistFunction.apply(torch.randn(7, 3, device=device), 1e-7)
    assert gh.shape == (7, 33) and torch.allclose(gh.view(7, 3, 11).sum(-1), torch.ones(7, 3, device=device), atol=1e-5)
    assert F_.EmbeddingFunction.apply(torch.randn(4, 6, device=device), idx0).shape == (0, 6)
    # optimizer step and non-finite flag on empty buffers
    z = torch.empty(0, device=device)
    call("jova_optim_step", 2, P(z), P(z), P(z), P(z), 0, 1e-3, 0.0, 1.0, 1, 0.9, 0.999, 1e-8, 0.0, 0.0, None, 0, st)
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    call("jova_nonfinite_flag", P(z), 0, P(flag), st)
    assert int(flag.item()) == 0


@pytest.mark.parametrize("form", ["fwd_bias_relu", "fwd_residual", "dx_plain", "dx_relu_mask", "dx_residual_k2048",
                                  "fwd_k32", "fwd_k64", "dx_k96", "fwd_unaligned_c"])
def test_bf16_pieces_with_b_split_beforehand(device, form):
    """jova_gemm_f32_img (B split into its bf16 pieces once per call, A split once per element: gemm_x3w_kernel) against
    fp64, with the same bound as the in-kernel split -- no worse than twice an fp32 FMA chain's own distance, floor 1e-6
    of max|C| -- on the forward form (x W^T + bias, ReLU / residual) and the data-gradient form (dY W, with the ReLU mask
    / a residual), with one to 64 k32 blocks and a half-empty last row tile, and against the in-kernel form to a few ulps
    of the largest element (same pieces, other summation order).  The size query answers 0 for the shapes and modes that
    stay on the other forms."""
    from jova_amd import _lib
    from jova_amd.nn import functional as F_
    L = _lib.lib()
    torch.manual_seed(21)
    M, N, K = (1152, 512, 2048) if form == "dx_residual_k2048" else (2048, 2048, 512)     # (1152: a half-empty last row tile)
    if form in ("fwd_k32", "fwd_k64", "dx_k96"):             # one, two and three k32 blocks: the peeled ends of the block loop
        M, N, K = 16384 + 128, 2048, {"fwd_k32": 32, "fwd_k64": 64, "dx_k96": 96}[form]
    fwd = form.startswith("fwd")
    A = torch.randn(M, K, device=device)
    A[::7] *= 37.0
    W = torch.randn((N, K) if fwd else (K, N), device=device) * 0.05
    sbk, sbn = (1, K) if fwd else (N, 1)
    bias = torch.randn(N, device=device) if form in ("fwd_bias_relu", "fwd_unaligned_c") else None
    R = torch.randn(M, N, device=device) if form in ("fwd_residual", "dx_relu_mask", "dx_residual_k2048") else None
    relu = {"fwd_bias_relu": 1, "dx_relu_mask": 2}.get(form, 0)
    need = L.jova_gemm_b_image_bytes(F_._addr(A), K, 1, F_._addr(W), sbk, sbn, M, N, K, 1, 1, 1)
    assert need == 6 * N * K
    assert L.jova_gemm_b_image_bytes(F_._addr(A), K, 1, F_._addr(W), sbk, sbn, M, N, K, 1, 1, 2) == 0      # split-K
    assert L.jova_gemm_b_image_bytes(F_._addr(A), K, 1, F_._addr(W), sbk, sbn, M - 1, N, K, 1, 1, 1) == 0  # ragged rows
    assert L.jova_gemm_b_image_bytes(F_._addr(A), K, 1, F_._addr(W), sbk, sbn, 128, 128, 512, 1, 1, 1) == 0  # small
    img = torch.empty(need, dtype=torch.uint8, device=device)
    F_.call("jova_gemm_b_image", F_._addr(W), sbk, sbn, N, K, F_.P(img), F_.stream())
    out = torch.full((M, N), float("nan"), device=device)
    args = (F_._addr(A), K, 1, F_._addr(W), sbk, sbn, F_._addr(out), N, M, N, K, 1.0, F_.P(bias),
            F_._addr(R) if R is not None else None, N, relu, 1, 1, 1, 0, 0, 0, 0, 0, 0)
    if form == "fwd_unaligned_c":                            # C four bytes off a 16-byte boundary, rows of N + 1 floats:
        buf = torch.full((M * (N + 1) + 1,), float("nan"), device=device)      # the scalar-store path of the epilogue
        F_.call("jova_gemm_f32_img", *(args[:6] + (F_._addr(buf, 1), N + 1) + args[8:]), F_.P(img), F_.stream())
        out = buf[1:].view(M, N + 1)[:, :N].contiguous()
        assert bool(torch.isnan(buf[1:].view(M, N + 1)[:, N]).all()) and bool(torch.isnan(buf[0])), "stores outside C"
    else:
        F_.call("jova_gemm_f32_img", *args, F_.P(img), F_.stream())
    out2 = torch.full((M, N), float("nan"), device=device)
    args2 = args[:6] + (F_._addr(out2),) + args[7:]
    F_.call("jova_gemm_f32", *args2, F_.stream())
    mode = L.jova_gemm_get_bf16x3()
    try:
        L.jova_gemm_set_bf16x3(0)
        assert L.jova_gemm_b_image_bytes(F_._addr(A), K, 1, F_._addr(W), sbk, sbn, M, N, K, 1, 1, 1) == 0
        out32 = torch.full((M, N), float("nan"), device=device)
        F_.call("jova_gemm_f32_img", *(args[:6] + (F_._addr(out32),) + args[7:]), F_.P(img), F_.stream())   # image ignored
    finally:
        L.jova_gemm_set_bf16x3(mode)
    Wd = W.double().t() if fwd else W.double()
    ref = A.double() @ Wd
    if bias is not None:
        ref = ref + bias.double()
    if relu == 2:
        ref = torch.where(R.double() > 0, ref, torch.zeros_like(ref))
    elif R is not None:
        ref = ref + R.double()
    if relu == 1:
        ref = ref.clamp_min(0)
    scale = float(ref.abs().max())
    e_img = float((out.double() - ref).abs().max()) / scale
    e_x3 = float((out2.double() - ref).abs().max()) / scale
    e_32 = float((out32.double() - ref).abs().max()) / scale
    assert e_img <= max(1e-6, 2 * e_32), (form, e_img, e_x3, e_32)
    assert float((out - out2).abs().max()) / scale <= 4e-6, "image form against the in-kernel split"
    if relu == 2:
        assert bool(((out == 0) == (R <= 0)).all()) or bool((out[R <= 0] == 0).all())
