"""The slab-reduced weight gradient (csrc/wgrad.hip: asr_gemm_tn_ws; solver.py:119 -> nn.Linear weight.grad of module.py:48-53,
attention.py:33-60, transformer.py:148) against torch fp32 on the bf16-rounded operands: ragged M (a partial last 64-row step), an
output that is not a multiple of the tile (padded dY rows), accumulate, the bias-gradient side product in both modes, repeated launches
on one workspace (the arrival counters must come back to zero), bit-identical repeats, and the atomics kernel it replaces."""
import numpy as np
import pytest
import torch

from asr_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = lambda t: t.detach().float().cpu().numpy()


def _ops(M, Nn, K, seed, lda=None):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, Nn, generator=g).bfloat16()
    b = torch.randn(M, K, generator=g).bfloat16()
    if lda:
        buf = torch.full((M, lda), 3.0, dtype=torch.bfloat16)      # finite padding: only rows >= N of the (unwritten) output see it
        buf[:, :Nn] = a
        ad = buf.to(DEV)[:, :Nn]
    else:
        ad = a.to(DEV)
    return a, b, ad, b.to(DEV)


@pytest.mark.parametrize("M,Nn,K,lda", [(32000, 256, 2048, None), (32000, 2048, 256, None), (32000, 768, 256, None), (32000, 256, 256, None),
                                        (1632, 256, 256, None), (1632, 2048, 256, None), (1000, 256, 2048, None), (4000, 4234, 256, 4352),
                                        (64, 128, 128, None), (72, 384, 128, None), (8192, 128, 128, None)])
def test_gemm_tn_slab(M, Nn, K, lda, monkeypatch):
    a, b, ad, bd = _ops(M, Nn, K, M + Nn, lda)
    ref = a.float().t() @ b.float()
    tol = dict(atol=2e-2 * (M / 300) ** 0.5, rtol=2e-3)
    cs = torch.zeros(Nn, device=DEV)
    out = torch.full((Nn, K), 7.0, device=DEV)                       # not pre-zeroed: the kernel overwrites
    ops.gemm_tn(ad, bd, out=out, colsum=cs, max_wgs=256)
    np.testing.assert_allclose(N(out), ref.numpy(), **tol)
    np.testing.assert_allclose(N(cs), a.float().sum(0).numpy(), atol=2e-2 * (M / 300) ** 0.5, rtol=2e-3)
    first = out.clone()
    for _ in range(3):                                               # same workspace again: counters were left at zero
        ops.gemm_tn(ad, bd, out=out, max_wgs=256)
        assert torch.equal(out, first)                               # fixed summation order: bit-identical
    ops.gemm_tn(ad, bd, out=out, accumulate=True, max_wgs=256)
    np.testing.assert_allclose(N(out), 2 * ref.numpy(), atol=2 * tol["atol"], rtol=2e-3)
    # deterministic bias gradient: one writer per element, bit-identical across launches
    monkeypatch.setattr(ops, "DETERMINISTIC", True)
    c1, c2 = torch.zeros(Nn, device=DEV), torch.zeros(Nn, device=DEV)
    ops.gemm_tn(ad, bd, out=out, colsum=c1, max_wgs=256)
    ops.gemm_tn(ad, bd, out=out, colsum=c2, max_wgs=256)
    assert torch.equal(c1, c2)
    np.testing.assert_allclose(N(c1), a.float().sum(0).numpy(), atol=2e-2 * (M / 300) ** 0.5, rtol=2e-3)
    # the atomics kernel computes the same product
    monkeypatch.setattr(ops, "TN_SLAB", False)
    old = ops.gemm_tn(ad, bd, max_wgs=256)
    np.testing.assert_allclose(N(first), N(old), atol=tol["atol"] * 0.5, rtol=1e-3)


@pytest.mark.parametrize("M,Nn,K", [(32000, 2048, 256), (32000, 256, 2048), (8000, 2048, 256), (8000, 256, 2048), (32000, 768, 256), (16000, 512, 512)])
@pytest.mark.parametrize("wgs", [128, 64, 32])
def test_gemm_tn_slab_fewer_m_ranges_than_xcds(M, Nn, K, wgs):
    """The side stream's launches (max_wgs = 128 and less: 1, 2 or 4 M-ranges): an M-range owns 8 / splits XCDs and its tiles are dealt
    to them by the larger operand's block index - every (tile, split) pair must still be computed exactly once."""
    a, b, ad, bd = _ops(M, Nn, K, M + Nn + wgs)
    ref = a.float().t() @ b.float()
    cs = torch.zeros(Nn, device=DEV)
    out = torch.full((Nn, K), float("nan"), device=DEV)
    ops.gemm_tn(ad, bd, out=out, colsum=cs, max_wgs=wgs)
    np.testing.assert_allclose(N(out), ref.numpy(), atol=2e-2 * (M / 300) ** 0.5, rtol=2e-3)
    np.testing.assert_allclose(N(cs), a.float().sum(0).numpy(), atol=2e-2 * (M / 300) ** 0.5, rtol=2e-3)
    same = ops.gemm_tn(ad, bd, max_wgs=256)
    np.testing.assert_allclose(N(out), N(same), atol=1e-2 * (M / 300) ** 0.5, rtol=1e-3)


def test_gemm_tn_slab_whole_chip_split():
    """max_wgs = 0 (the launch has the chip to itself) and a second destination: separate workspaces."""
    a, b, ad, bd = _ops(32000, 256, 2048, 5)
    ref = (a.float().t() @ b.float()).numpy()
    o1 = ops.gemm_tn(ad, bd)
    o2 = ops.gemm_tn(ad, bd)
    assert torch.equal(o1, o2)
    np.testing.assert_allclose(N(o1), ref, atol=0.25, rtol=2e-3)


def test_grouped_weight_gradients_equal_single_launches():
    """asr_gemm_tn_ws_group: the decoder's shapes (1632 rows; 256 / 768 / 2048 / 4234-in-4352 outputs, accumulate and overwrite, with and
    without the bias side product) in one pair of launches against one asr_gemm_tn_ws each."""
    M = 1632
    shapes = [(256, 256), (768, 256), (2048, 256), (256, 2048), (256, 256), (4234, 256), (256, 256), (768, 256)]
    g = torch.Generator().manual_seed(9)
    probs, refs = [], []
    for i, (Nn, K) in enumerate(shapes):
        lda = 4352 if Nn == 4234 else None
        a, b, ad, bd = _ops(M, Nn, K, 50 + i, lda)
        acc = i % 2 == 1
        out = (torch.randn(Nn, K, generator=g) if acc else torch.full((Nn, K), 3.0)).to(DEV)
        cs = torch.zeros(Nn, device=DEV) if i % 3 == 0 else None
        ref = a.float().t() @ b.float() + (out.cpu() if acc else 0.0)
        assert ops.gemm_tn_group_ok(ad, bd, out)
        probs.append((ad, bd, out, acc, cs))
        refs.append((ref, a.float().sum(0) if cs is not None else None))
    ops.gemm_tn_group(probs)
    for (ad, bd, out, acc, cs), (ref, csr) in zip(probs, refs):
        np.testing.assert_allclose(N(out), ref.numpy(), atol=6e-2, rtol=2e-3)
        if cs is not None:
            np.testing.assert_allclose(N(cs), csr.numpy(), atol=6e-2, rtol=2e-3)
    first = [p[2].clone() for p in probs]
    # again on the same workspaces, overwrite form: bit-identical to itself, and equal to the single launches
    probs2 = [(ad, bd, torch.empty_like(out), False, None) for ad, bd, out, acc, cs in probs]
    ops.gemm_tn_group(probs2)
    probs3 = [(ad, bd, torch.empty_like(out), False, None) for ad, bd, out, acc, cs in probs]
    ops.gemm_tn_group(probs3)
    for p2, p3 in zip(probs2, probs3):
        assert torch.equal(p2[2], p3[2])
        single = ops.gemm_tn(p2[0], p2[1], max_wgs=256)
        np.testing.assert_allclose(N(p2[2]), N(single), atol=2e-2, rtol=1e-3)


def test_batched_unsplit_weight_gradients_of_an_encoder_layer():
    """asr_gemm_tn_ws_group_wgs with an explicit workgroup budget: the four weight gradients of two encoder layers (Q/K/V [768 x 256],
    output projection [256 x 256], feed-forward [2048 x 256] and [256 x 2048]; module.py:48-53, attention.py:33-60) over 8 037 rows in
    ONE launch - budget = tiles: every problem unsplit over M (no slab, no reduce launch); budget = 2 x tiles: M-splits - accumulate
    and overwrite, with the bias side products, against fp32 matmuls of the same bf16 operands, bit-identical to itself, and equal
    to the single launches."""
    M = 8000 + 37
    shapes = [(768, 256), (256, 256), (2048, 256), (256, 2048)] * 2
    g = torch.Generator().manual_seed(19)
    probs, refs = [], []
    for i, (Nn, K) in enumerate(shapes):
        a, b, ad, bd = _ops(M, Nn, K, 80 + i, None)
        acc = i % 2 == 0
        out = (torch.randn(Nn, K, generator=g) if acc else torch.full((Nn, K), 3.0)).to(DEV)
        cs = torch.zeros(Nn, device=DEV) if i % 4 in (0, 2) else None
        ref = a.float().t() @ b.float() + (out.cpu() if acc else 0.0)
        probs.append((ad, bd, out, acc, cs))
        refs.append((ref, a.float().sum(0) if cs is not None else None))
    tiles = sum(ops.tn_tiles(p[0], p[1]) for p in probs)
    assert tiles == 160
    ops.gemm_tn_group(probs, group_wgs=tiles)
    for (ad, bd, out, acc, cs), (ref, csr) in zip(probs, refs):
        np.testing.assert_allclose(N(out), ref.numpy(), atol=0.15, rtol=2e-3)     # sums of 8 037 products of O(1) bf16 values
        if cs is not None:
            np.testing.assert_allclose(N(cs), csr.numpy(), atol=6e-2, rtol=2e-3)
    again = [(ad, bd, torch.empty_like(out), False, None) for ad, bd, out, acc, cs in probs]
    ops.gemm_tn_group(again, group_wgs=tiles)
    again2 = [(ad, bd, torch.empty_like(out), False, None) for ad, bd, out, acc, cs in probs]
    ops.gemm_tn_group(again2, group_wgs=tiles)
    split = [(ad, bd, torch.empty_like(out), False, None) for ad, bd, out, acc, cs in probs]
    ops.gemm_tn_group(split, group_wgs=2 * tiles)
    for p2, ps in zip(again, split):
        np.testing.assert_allclose(N(p2[2]), N(ps[2]), atol=3e-2, rtol=1e-3)
    for p2, p3 in zip(again, again2):
        assert torch.equal(p2[2], p3[2])
        single = ops.gemm_tn(p2[0], p2[1], max_wgs=256)
        np.testing.assert_allclose(N(p2[2]), N(single), atol=3e-2, rtol=1e-3)
