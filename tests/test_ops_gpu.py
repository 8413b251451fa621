"""GPU parity tests of every C-ABI kernel against plain torch fp32 / the numpy oracle on the same seeded inputs.
Integer / index work must be bit-exact; floating point tolerances are written at each assert."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pvpuformer_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


TD = {0: torch.bfloat16, 1: torch.float32}


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("shape", [(3, 448, 448), (2, 97, 131), (1, 5, 1)])
@pytest.mark.parametrize("border", [True, False])
def test_edt_bit_exact_vs_scipy(ops, shape, border):
    """vpu_edt == float32(scipy.ndimage.distance_transform_edt) bit for bit: blobs, thin lines, an all-ones mask (only
    the zero border -- or nothing: +inf -- to measure against), an all-zero mask; with the implicit zero border it
    equals the transform of the mask padded by one pixel (the reference's np.pad + [1:-1, 1:-1])."""
    from scipy import ndimage
    g = np.random.RandomState(7)
    B, H, W = shape
    masks = np.zeros(shape, np.uint8)
    for b in range(B):
        m = g.rand(H, W) > (0.02 if b == 0 else 0.5)
        if H > 20:
            m[H // 4:H // 2, W // 3:W // 3 + 2] = False
        masks[b] = m
    if B > 1:
        masks[1] = 1
    if B > 2:
        masks[2] = 0
    got = ops.edt(dev(torch.from_numpy(masks)), zero_border=border).cpu().numpy()
    for b in range(B):
        if border:
            ref = ndimage.distance_transform_edt(np.pad(masks[b], 1, "constant")).astype(np.float32)[1:-1, 1:-1]
        elif masks[b].all():
            ref = np.full((H, W), np.inf, np.float32)
        else:
            ref = ndimage.distance_transform_edt(masks[b]).astype(np.float32)
        assert np.array_equal(got[b], ref), (b, np.abs(got[b] - ref).max())


@pytest.mark.parametrize("as_allmask,gpu_cc", [(False, True), (False, False), (True, True)])      # gpu_cc = True is the default
def test_get_next_promts_gpu_equals_host(ops, as_allmask, gpu_cc, monkeypatch):
    """The device path of get_next_promts (masks, distance transforms, maxima and the k-th-candidate lookup on the GPU)
    against the host path (numpy + scipy) on the same inputs and the same random streams, three rounds deep: identical
    points (click coordinates, slots, orders), boxes, P2CL slot table and override masks."""
    import random
    from pvpuformer_amd.isegm.engine import prompt_sim
    from pvpuformer_amd.isegm.engine.prompt_sim import PromptState, get_next_promts
    monkeypatch.setattr(prompt_sim, "GPU_CC", gpu_cc)     # connected components of cal_box on the host / on the GPU
    B, H, n = 5, 96, 24
    g = torch.Generator().manual_seed(9)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
    gt = torch.zeros(B, 1, H, H)
    for b in range(B):
        cy, cx, r = (torch.rand(3, generator=g) * torch.tensor([H * 0.6, H * 0.6, H * 0.25]) + torch.tensor([H * 0.2, H * 0.2, 6.0])).tolist()
        gt[b, 0] = ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r).float()
    gt[4] = 0                                             # an empty object: no click, no box
    points0 = -torch.ones(B, 2 * n, 3)
    points0[:, 0] = torch.tensor([10.0, 12.0, 0.0])
    results = []
    for device in ("cpu", "cuda"):
        rng, np_rng = random.Random(3), np.random.RandomState(3)
        pts = points0.clone().to(device)
        state = PromptState(B, 2 * n, H, H, device, max_rounds=3)
        gg = torch.Generator().manual_seed(10)
        trace = []
        for rnd in range(3):
            noise = torch.rand(B, 1, H, H, generator=gg)
            pred = (0.75 * gt + 0.25 * noise if rnd else torch.zeros(B, 1, H, H)).to(device)
            pts, boxes = get_next_promts(pred, gt.to(device), pts, state, as_allmask=as_allmask, np_rng=np_rng, rng=rng)
            trace.append((pts.cpu().clone(), boxes.cpu().clone()))
        results.append((trace, state.slot_idx.cpu().clone(), state.override.cpu().clone(), state.used))
    (ta, sa, oa, ua), (tb, sb, ob, ub) = results
    assert ua == ub and ua > 0
    for (pa, ba), (pb, bb) in zip(ta, tb):
        assert torch.equal(pa, pb) and torch.equal(ba, bb)
    assert torch.equal(sa, sb) and torch.equal(oa[:ua], ob[:ub])      # (planes beyond `used` are never read: uninitialised)


def test_cc_roots_equal_scipy_components(ops):
    """vpu_cc_roots: the partition into 8-connected components and their order (by smallest pixel index) equal
    scipy.ndimage.label with the 3x3 structure, on noise at several densities, blobs, a spiral, a full and an empty
    mask; root = smallest linear index of the component."""
    from scipy import ndimage
    g = np.random.RandomState(11)
    H, W = 97, 131
    masks = []
    for dens in (0.1, 0.45, 0.6, 0.9):
        masks.append(g.rand(H, W) < dens)
    masks.append(ndimage.binary_opening(g.rand(H, W) < 0.6, iterations=2))
    sp = np.zeros((H, W), bool)                      # a spiral: one long thin component
    y0, y1, x0, x1 = 0, H - 1, 0, W - 1
    while y1 - y0 > 3 and x1 - x0 > 3:
        sp[y0, x0:x1 + 1] = True; sp[y0:y1 + 1, x1] = True; sp[y1, x0 + 2:x1 + 1] = True; sp[y0 + 2:y1 + 1, x0 + 2] = True
        y0, y1, x0, x1 = y0 + 2, y1 - 2, x0 + 2, x1 - 2   # (touching rings: 8-connectivity joins them)
    masks += [sp, np.ones((H, W), bool), np.zeros((H, W), bool)]
    m = np.stack(masks).astype(np.uint8)
    roots = ops.cc_roots(dev(torch.from_numpy(m))).cpu().numpy()
    for b in range(len(masks)):
        lab, n = ndimage.label(masks[b], structure=np.ones((3, 3), bool))
        r = roots[b]
        assert ((r >= 0) == masks[b]).all()
        if n == 0:
            continue
        rr = r[masks[b]] - b * H * W
        uniq = np.unique(rr)
        assert len(uniq) == n
        # ascending roots <-> scipy labels 1..n, and every root is the smallest index of its component
        flat = np.flatnonzero(masks[b].ravel())
        for k, u in enumerate(uniq):
            members = flat[rr == u]
            assert members.min() == u
            assert (lab.ravel()[members] == k + 1).all()


def test_cc_table_and_kept_region_boxes(ops):
    """vpu_cc_table over the labels of vpu_cc_roots: per component the exact pixel count and bounding box (against
    scipy.ndimage on noise, blobs, a 300-pixel-wide run that spans several waves, a full and an empty mask), the component
    count behind the last row, the overflow signal (count > kmax); and ``_kept_region_boxes`` -- what cal_box takes of
    max_connected_regions -- equal to the host function on the same masks, with and without the overflow fallback."""
    from scipy import ndimage
    from pvpuformer_amd.isegm.engine import prompt_sim as ps
    g = np.random.RandomState(12)
    H, W = 120, 333
    masks = [g.rand(H, W) < d for d in (0.05, 0.4, 0.62)]
    masks.append(ndimage.binary_opening(g.rand(H, W) < 0.65, iterations=2))
    wide = np.zeros((H, W), bool); wide[40:44, 10:310] = True; wide[80:100, 200:230] = True; wide[5, 5] = True
    masks += [wide, np.ones((H, W), bool), np.zeros((H, W), bool)]
    m = np.stack(masks).astype(np.uint8)
    md = dev(torch.from_numpy(m))
    roots = ops.cc_roots(md)
    kmax = 1 << 14
    flat = ops.cc_table(roots, kmax).cpu().numpy()
    K = int(flat[-1])
    table = flat[:6 * K].reshape(K, 6)
    table = table[np.argsort(table[:, 0])]
    want = []
    for b, mk in enumerate(masks):
        lab, n = ndimage.label(mk, structure=np.ones((3, 3), bool))
        for k, sl in enumerate(ndimage.find_objects(lab)):
            ys, xs = np.nonzero(lab == k + 1)
            want.append((b * H * W + int((ys * W + xs).min()), len(ys), ys.min(), ys.max(), xs.min(), xs.max()))
    want = np.array(sorted(want), np.int64)
    assert K == len(want) and K < kmax and np.array_equal(table, want)
    assert int(ops.cc_table(roots, 8).cpu().numpy()[-1]) == K          # more components than rows: the count says so
    host = []
    for mk in masks:
        region = ps.max_connected_regions(mk) == 1
        rows, cols = np.flatnonzero(region.any(1)), np.flatnonzero(region.any(0))
        host.append(None if len(rows) == 0 else (rows[0], rows[-1], cols[0], cols[-1]))
    assert ps._kept_region_boxes(md, kmax) == host
    assert ps._kept_region_boxes(md, 8) == host                          # overflow: the host labelling takes over


def test_chamfer5_bit_exact(ops):
    """vpu_chamfer5 (cv2.distanceTransform(DIST_L2, 5) of the training simulators, restated from OpenCV's two-pass fixed-point
    algorithm) == oracle/vpu_oracle.py::chamfer_l2_5x5 bit for bit: blobs, speckle, an empty and a full mask, 448 x 448 and a
    ragged 96 x 131, with the implicit zero border the reference's np.pad gives and without."""
    import vpu_oracle as vo
    g = np.random.RandomState(3)
    for H, W in ((448, 448), (96, 131)):
        yy, xx = np.mgrid[0:H, 0:W]
        masks = [((yy - H * 0.4) ** 2 / (H * 0.3) ** 2 + (xx - W * 0.5) ** 2 / (W * 0.35) ** 2 <= 1.0),
                 g.rand(H, W) > 0.03, g.rand(H, W) > 0.6, np.zeros((H, W), bool), np.ones((H, W), bool)]
        masks[0][H // 3: H // 3 + 7, W // 4: W // 2] = False
        m = np.stack(masks).astype(np.uint8)
        for border in (True, False):
            got = ops.chamfer5(dev(torch.from_numpy(m)), zero_border=border).cpu().numpy()
            for i in range(len(masks)):
                want = vo.chamfer_l2_5x5(np.pad(m[i], 1))[1:-1, 1:-1] if border else vo.chamfer_l2_5x5(m[i])
                assert np.array_equal(got[i], want), (H, W, border, i, np.abs(got[i] - want).max())
    # against the exact transform: the chamfer metric (a = 1, b = 1.4 < sqrt 2, c = 2.1969 < sqrt 5) is within ~2-3 % of it
    e = ops.edt(dev(torch.from_numpy(m[:1])), zero_border=True).cpu().numpy()[0]
    c = ops.chamfer5(dev(torch.from_numpy(m[:1])), zero_border=True).cpu().numpy()[0]
    assert np.all(c >= e * 0.97 - 1e-3) and np.all(c <= e * 1.03 + 1e-3)


def test_colsum_batched(ops):
    """70 independent column sums (more than one launch's 64) of different shapes accumulate into slices of one flat
    buffer exactly like per-job fp32 sums (integer data: exact)."""
    g = torch.Generator().manual_seed(5)
    flat = torch.arange(0, 70 * 4096, dtype=torch.float32).remainder(7).cuda()
    ref = flat.clone().cpu()
    jobs = []
    for i in range(70):
        rows, C = 1 + (i * 37) % 300, 8 * (1 + (i * 13) % 400)
        part = torch.randint(-4, 5, (rows, C), generator=g).float()
        jobs.append((dev(part), (flat, i * 4096), rows, C))
        ref[i * 4096:i * 4096 + C] += part.sum(0)
    ops.colsum_batched(jobs)
    torch.cuda.synchronize()
    assert torch.equal(flat.cpu(), ref)
    # strided output blocks (row_len, out_ld): S slabs of a [24, 256] gradient summed into a column block of a [24, 1024]
    # matrix (the head's fusion weight), few-row 16-byte form and (row_len 36: not 16-byte rows) the generic form
    for S, N, K, ld, col in ((5, 24, 256, 1024, 512), (40, 24, 256, 1024, 256), (3, 10, 36, 100, 12)):
        wide = torch.arange(0, N * ld, dtype=torch.float32).remainder(5).cuda()
        slabs = torch.randint(-3, 4, (S, N, K), generator=g).float()
        want = wide.clone().cpu().view(N, ld)
        want[:, col:col + K] += slabs.sum(0)
        ops.colsum_batched([(dev(slabs), (wide, col), S, N * K, K, ld)])
        torch.cuda.synchronize()
        assert torch.equal(wide.cpu().view(N, ld), want), (S, N, K)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_exact_integers(ops, dtype, tA, tB):
    """Small-integer operands: every product and sum is exact in bf16 x bf16 -> fp32, so the result must be
    bit-identical to an fp32 matmul.  Asymmetric data, ragged M/N, K not a multiple of the tile."""
    M, N, K = 200, 136, 72
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    Bm = torch.randint(-3, 4, (N, K), generator=g).float()
    A[0, 1] = 3; A[1, 0] = -2; Bm[0, 1] = 1; Bm[1, 0] = -3
    ref = A @ Bm.t()
    lda = 208 if tA else 80
    ldb = 144 if tB else 80
    Am = torch.zeros((K, lda) if tA else (M, lda))
    Bs = torch.zeros((K, ldb) if tB else (N, ldb))
    if tA: Am[:, :M] = A.t()
    else: Am[:, :K] = A
    if tB: Bs[:, :N] = Bm.t()
    else: Bs[:, :K] = Bm
    Ad, Bd = dev(Am).to(TD[dtype]), dev(Bs).to(TD[dtype])
    Cd = torch.full((M, 152), 7.0, device="cuda", dtype=TD[dtype])
    ops.gemm(Ad, Bd, Cd, M, N, K, lda, ldb, 152, dtype, transA=bool(tA), transB=bool(tB))
    torch.cuda.synchronize()
    out = Cd.float().cpu()
    assert torch.equal(out[:, :N], ref), f"max diff {(out[:, :N] - ref).abs().max()}"
    assert torch.all(out[:, N:] == 7.0), "wrote outside the N range"


@pytest.mark.lab
@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("K", [72, 200, 648])
def test_gemm_ring_pipeline_exact(ops, tA, tB, K):
    """The three-stage LDS-DMA ring main loop (vpu_gemm_set_option("ring", 2)): exact-integer operands must give the
    fp32 matmul bit for bit for every operand layout, ragged M / N, K shorter than / equal to / longer than the ring
    (2, 4 and 11 K-tiles: the out-of-range tail stages are requested as zeros), plus the fused bias-gradient column sums
    of the weight-gradient form."""
    M, N = 200, 136
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    Bm = torch.randint(-3, 4, (N, K), generator=g).float()
    ref = A @ Bm.t()
    lda = 208 if tA else (K + 8)
    ldb = 144 if tB else (K + 8)
    Am = torch.zeros((K, lda) if tA else (M, lda))
    Bs = torch.zeros((K, ldb) if tB else (N, ldb))
    if tA: Am[:, :M] = A.t()
    else: Am[:, :K] = A
    if tB: Bs[:, :N] = Bm.t()
    else: Bs[:, :K] = Bm
    Ad, Bd = dev(Am).to(torch.bfloat16), dev(Bs).to(torch.bfloat16)
    Cd = torch.full((M, 152), 7.0, device="cuda", dtype=torch.float32)
    cs = torch.zeros(M, device="cuda") if (tA and tB) else None
    ops.gemm_set_option("ring", 2)
    try:
        ops.gemm(Ad, Bd, Cd, M, N, K, lda, ldb, 152, 0, transA=bool(tA), transB=bool(tB), flags=ops.EPI_OUT_F32,
                 workspace=None, colsum=cs)
        torch.cuda.synchronize()
    finally:
        ops.gemm_set_option("ring", -1)
    out = Cd.cpu()
    assert torch.equal(out[:, :N], ref), f"max diff {(out[:, :N] - ref).abs().max()}"
    assert torch.all(out[:, N:] == 7.0), "wrote outside the N range"
    if cs is not None:
        assert torch.equal(cs.cpu(), A.sum(1))


@pytest.mark.parametrize("dtype", [0, 1])
def test_gemm_epilogues(ops, dtype):
    M, N, K = 300, 192, 128
    A, W = rnd(M, K, seed=2), rnd(N, K, seed=3, scale=0.2)
    bias, R = rnd(N, seed=4), rnd(M, N, seed=5)
    td = TD[dtype]
    Ad, Wd, Rd, bd = dev(A).to(td), dev(W).to(td), dev(R).to(td), dev(bias)
    Af, Wf, Rf = Ad.float(), Wd.float(), Rd.float()
    base = Af @ Wf.t()
    tol = dict(atol=3e-2, rtol=2e-2) if dtype == 0 else dict(atol=2e-5, rtol=1e-5)

    def run(flags, **kw):
        Cd = torch.zeros(M, N, device="cuda", dtype=torch.float32 if flags & ops.EPI_OUT_F32 else td)
        ops.gemm(Ad, Wd, Cd, M, N, K, K, K, N, dtype, flags=flags, **kw)
        torch.cuda.synchronize()
        return Cd.float()

    # bias + GELU with pre-activation side output + residual
    pre = torch.zeros(M, N, device="cuda", dtype=td)
    out = run(ops.EPI_BIAS | ops.EPI_PREACT | ops.EPI_GELU | ops.EPI_RESID, bias=bd, preact=pre, resid=Rd, ldr=N)
    torch.testing.assert_close(pre.float(), base + bd, **tol)
    torch.testing.assert_close(out, F.gelu(base + bd) + Rf, **tol)
    out = run(ops.EPI_BIAS | ops.EPI_RELU, bias=bd)
    torch.testing.assert_close(out, F.relu(base + bd), **tol)
    # activation backward epilogues
    aux = dev(rnd(M, N, seed=6, scale=2.0)).to(td)
    x = aux.float().clone().requires_grad_(True)
    F.gelu(x).sum().backward()
    out = run(ops.EPI_DGELU, aux=aux, ldaux=N)
    torch.testing.assert_close(out, base * x.grad, **tol)
    out = run(ops.EPI_DRELU, aux=aux, ldaux=N)
    torch.testing.assert_close(out, base * (aux.float() > 0).float(), **tol)
    # alpha, affine, fp32 output with accumulation
    Cd = torch.full((M, N), 2.0, device="cuda", dtype=torch.float32)
    ops.gemm(Ad, Wd, Cd, M, N, K, K, K, N, dtype, flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM | ops.EPI_AFFINE, alpha=0.5,
             post_mul=0.5, post_add=0.5)
    torch.testing.assert_close(Cd, (0.5 * base) * 0.5 + 0.5 + 2.0, **tol)
    # broadcast residual with a row period (pos_embed)
    Rp = dev(rnd(100, N, seed=7)).to(td)
    out = run(ops.EPI_RESID, resid=Rp, ldr=N, resid_period=100)
    torch.testing.assert_close(out, base + Rp.float().repeat(3, 1), **tol)


@pytest.mark.parametrize("dtype", [0, 1])
def test_gemm_attention_layout(ops, dtype):
    """Batched two-level strides exactly as the backbone uses them: qkv [B*n, 3*H*d] -> S, P.V, and all backward
    products, n = 196 (ragged: leading dimension padded to 200)."""
    Bw, n, H, d = 3, 196, 2, 64
    D = H * d
    td = TD[dtype]
    qkv = dev(rnd(Bw * n, 3 * D, seed=8)).to(td)
    q = qkv.float().view(Bw, n, 3, H, d).permute(2, 0, 3, 1, 4)
    ldS = 200
    S = torch.zeros(Bw * H, n, ldS, device="cuda", dtype=torch.float32)
    ops.gemm(qkv, (qkv, D), S, n, n, d, 3 * D, 3 * D, ldS, dtype, flags=ops.EPI_OUT_F32, alpha=0.125, batch=Bw * H,
             inner=H, sA=(n * 3 * D, d), sB=(n * 3 * D, d), sC=(H * n * ldS, n * ldS))
    Sref = (q[0] @ q[1].transpose(-1, -2)) * 0.125
    tol = dict(atol=5e-2, rtol=2e-2) if dtype == 0 else dict(atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(S.view(Bw, H, n, ldS)[..., :n], Sref, **tol)
    P = torch.zeros(Bw * H, n, ldS, device="cuda", dtype=td)
    ops.softmax_fwd(S, ldS, P, ldS, Bw * H * n, n)
    Pref = torch.softmax(Sref, -1)
    torch.testing.assert_close(P.float().view(Bw, H, n, ldS)[..., :n], Pref, atol=2e-3 if dtype == 0 else 1e-6, rtol=1e-2)
    assert torch.all(P[..., n:] == 0)
    # O = P V  (B operand K-major with ld 3D), written head-interleaved
    O = torch.zeros(Bw * n, D, device="cuda", dtype=td)
    ops.gemm(P, (qkv, 2 * D), O, n, d, n, ldS, 3 * D, D, dtype, transB=True, batch=Bw * H, inner=H,
             sA=(H * n * ldS, n * ldS), sB=(n * 3 * D, d), sC=(n * D, d))
    Oref = (P.float().view(Bw, H, n, ldS)[..., :n] @ q[2]).transpose(1, 2).reshape(Bw * n, D)
    torch.testing.assert_close(O.float(), Oref, **tol)
    # backward products
    dO = dev(rnd(Bw * n, D, seed=9)).to(td)
    dOh = dO.float().view(Bw, n, H, d).transpose(1, 2)
    dP = torch.zeros(Bw * H, n, ldS, device="cuda", dtype=torch.float32)
    ops.gemm(dO, (qkv, 2 * D), dP, n, n, d, D, 3 * D, ldS, dtype, flags=ops.EPI_OUT_F32, batch=Bw * H, inner=H,
             sA=(n * D, d), sB=(n * 3 * D, d), sC=(H * n * ldS, n * ldS))
    torch.testing.assert_close(dP.view(Bw, H, n, ldS)[..., :n], dOh @ q[2].transpose(-1, -2), **tol)
    dqkv = torch.zeros(Bw * n, 3 * D, device="cuda", dtype=td)
    # dV = P^T dO : A = P stored [k = nq][m = nk] (K-major), B = dO [k = nq][d] (K-major)
    ops.gemm(P, dO, (dqkv, 2 * D), n, d, n, ldS, D, 3 * D, dtype, transA=True, transB=True, batch=Bw * H, inner=H,
             sA=(H * n * ldS, n * ldS), sB=(n * D, d), sC=(n * 3 * D, d))
    dVref = (P.float().view(Bw, H, n, ldS)[..., :n].transpose(-1, -2) @ dOh)
    torch.testing.assert_close(dqkv.float().view(Bw, n, 3, H, d)[:, :, 2].transpose(1, 2), dVref, **tol)
    dS = torch.zeros(Bw * H, n, ldS, device="cuda", dtype=td)
    ops.softmax_bwd(P, ldS, dP, ldS, dS, Bw * H * n, n, 0.125)
    Pf = P.float().view(Bw, H, n, ldS)[..., :n]
    dPf = dP.view(Bw, H, n, ldS)[..., :n]
    dSref = Pf * (dPf - (Pf * dPf).sum(-1, keepdim=True)) * 0.125
    torch.testing.assert_close(dS.float().view(Bw, H, n, ldS)[..., :n], dSref, atol=3e-3 if dtype == 0 else 1e-6, rtol=2e-2)
    assert torch.all(dS[..., n:] == 0)
    # dQ = dS K ; dK = dS^T Q
    ops.gemm(dS, (qkv, D), dqkv, n, d, n, ldS, 3 * D, 3 * D, dtype, transB=True, batch=Bw * H, inner=H,
             sA=(H * n * ldS, n * ldS), sB=(n * 3 * D, d), sC=(n * 3 * D, d))
    ops.gemm(dS, qkv, (dqkv, D), n, d, n, ldS, 3 * D, 3 * D, dtype, transA=True, transB=True, batch=Bw * H, inner=H,
             sA=(H * n * ldS, n * ldS), sB=(n * 3 * D, d), sC=(n * 3 * D, d))
    dSf = dS.float().view(Bw, H, n, ldS)[..., :n]
    got = dqkv.float().view(Bw, n, 3, H, d)
    torch.testing.assert_close(got[:, :, 0].transpose(1, 2), dSf @ q[1], **tol)
    torch.testing.assert_close(got[:, :, 1].transpose(1, 2), dSf.transpose(-1, -2) @ q[0], **tol)


# ------------------------------------------------------------------------------------------------ row ops
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("C", [64, 768, 1280])
def test_layernorm(ops, dtype, C):
    rows = 333
    td = TD[dtype]
    x = dev(rnd(rows, C, seed=10, scale=2.0) + 0.3).to(td)
    w, b = dev(1 + rnd(C, seed=11, scale=0.2)), dev(rnd(C, seed=12, scale=0.2))
    y = torch.empty_like(x)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm_fwd(x, w, b, y, mean, rstd, rows, C, 1e-6)
    xf = x.float().requires_grad_(True)
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (C,), wf, bf, 1e-6)
    tol = dict(atol=3e-2, rtol=2e-2) if dtype == 0 else dict(atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(y.float(), ref, **tol)
    dy = dev(rnd(rows, C, seed=13)).to(td)
    dres = dev(rnd(rows, C, seed=14)).to(td)
    ref.backward(dy.float())
    nblk = ops.layernorm_bwd_nblk(rows)
    part = torch.zeros(nblk, 2, C, device="cuda")
    dx = torch.empty_like(x)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dres, dx, part, rows, C)
    dwb = torch.zeros(2 * C, device="cuda")
    ops.colsum_f32(part, dwb, nblk, 2 * C)
    dw, db = dwb[:C], dwb[C:]
    torch.testing.assert_close(dx.float(), xf.grad + dres.float(), **tol)
    gtol = dict(atol=0.15, rtol=3e-2) if dtype == 0 else dict(atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(dw, wf.grad, **gtol)
    torch.testing.assert_close(db, bf.grad, **gtol)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("C,pe_rows", [(768, 333), (768, 37), (1280, 111), (256, 9)])
def test_layernorm_with_position_embedding_add(ops, dtype, C, pe_rows):
    """vpu_layernorm_fwd_pe / vpu_layernorm_bwd2 (the DMA neck's LayerNorm -> + position embedding, transformer.py:439-457, in
    one launch each way): both outputs bit-identical to the LayerNorm launch followed by the broadcast add (the sum is taken
    from the rounded y); the backward over the two outputs' gradients == the plain backward over their sum -- bit-identical
    in fp32, within one bf16 rounding of the summed gradient otherwise."""
    rows = 333
    td = TD[dtype]
    x = dev(rnd(rows, C, seed=10, scale=2.0) + 0.3).to(td)
    w, b = dev(1 + rnd(C, seed=11, scale=0.2)), dev(rnd(C, seed=12, scale=0.2))
    pe = dev(rnd(pe_rows, C, seed=16)).to(td)
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm_fwd_pe(x, w, b, y, mean, rstd, rows, C, 1e-5, pe, pe_rows, y2)
    y_ref, yy_ref = torch.empty_like(x), torch.empty_like(x)
    mean_r, rstd_r = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm_fwd(x, w, b, y_ref, mean_r, rstd_r, rows, C, 1e-5)
    ops.add_bcast(y_ref, pe, yy_ref, rows * C, pe_rows * C)
    assert torch.equal(y, y_ref) and torch.equal(y2, yy_ref) and torch.equal(mean, mean_r) and torch.equal(rstd, rstd_r)
    dy, dy2 = dev(rnd(rows, C, seed=13)).to(td), dev(rnd(rows, C, seed=17)).to(td)
    dres = dev(rnd(rows, C, seed=14)).to(td)
    nblk = ops.layernorm_bwd_nblk(rows)
    part, part_r = torch.zeros(nblk, 2, C, device="cuda"), torch.zeros(nblk, 2, C, device="cuda")
    dx, dx_r = torch.empty_like(x), torch.empty_like(x)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dres, dx, part, rows, C, dy2=dy2)
    dsum = torch.empty_like(dy)
    ops.add4(dy, dy2, None, None, dsum, rows * C)
    ops.layernorm_bwd(dsum, x, w, mean, rstd, dres, dx_r, part_r, rows, C)
    if dtype == 1:
        assert torch.equal(dx, dx_r) and torch.equal(part, part_r)
    else:
        torch.testing.assert_close(dx.float(), dx_r.float(), atol=4e-2, rtol=2e-2)
        torch.testing.assert_close(part.sum(0), part_r.sum(0), atol=0.3, rtol=3e-2)
        # against the exact sum: closer than the path that rounds the summed gradient
        xf = x.float().requires_grad_(True)
        F.layer_norm(xf, (C,), w, b, 1e-5).backward(dy.float() + dy2.float())
        exact = xf.grad + dres.float()
        assert (dx.float() - exact).abs().mean() <= (dx_r.float() - exact).abs().mean() * 1.05
    with pytest.raises(Exception):
        ops.layernorm_fwd_pe(x, w, b, y, mean, rstd, rows, C, 1e-5, pe, 0, y2)


@pytest.mark.parametrize("dtype", [0, 1])
def test_colsum_l2norm_add_cast(ops, dtype):
    td = TD[dtype]
    rows, C, ld = 1000, 256, 264
    x = dev(rnd(rows, ld, seed=15)).to(td)
    out = torch.ones(C, device="cuda")
    part = torch.zeros(64, C, device="cuda")
    ops.colsum(x, ld, out, part, rows, C, beta=1.0)
    torch.testing.assert_close(out, 1 + x.float()[:, :C].sum(0), atol=1e-2, rtol=1e-4)
    out2 = torch.ones(C, device="cuda")
    ops.colsum(x, ld, out2, None, 12, C, beta=1.0)   # <= 64 rows: single-pass kernel, no partial buffer
    torch.testing.assert_close(out2, 1 + x.float()[:12, :C].sum(0), atol=1e-3, rtol=1e-4)
    xs = x[:, :C].contiguous()
    y, inv = torch.empty_like(xs), torch.empty(rows, device="cuda")
    ops.l2norm_fwd(xs, y, inv, rows, C)
    xf = xs.float().requires_grad_(True)
    ref = F.normalize(xf, p=2, dim=1)
    tol = dict(atol=1e-2, rtol=2e-2) if dtype == 0 else dict(atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(y.float(), ref, **tol)
    dy = dev(rnd(rows, C, seed=16)).to(td)
    ref.backward(dy.float())
    dx = torch.empty_like(xs)
    ops.l2norm_bwd(dy, y, inv, dx, rows, C)
    torch.testing.assert_close(dx.float(), xf.grad, atol=2e-2 if dtype == 0 else 1e-5, rtol=5e-2 if dtype == 0 else 1e-4)
    x40 = dev(rnd(77, 40, seed=20)).to(td)          # C = 40: the one-element-per-lane kernels
    y40, inv40, dx40 = torch.empty_like(x40), torch.empty(77, device="cuda"), torch.empty_like(x40)
    ops.l2norm_fwd(x40, y40, inv40, 77, 40)
    x40f = x40.float().requires_grad_(True)
    r40 = F.normalize(x40f, p=2, dim=1)
    torch.testing.assert_close(y40.float(), r40, **tol)
    r40.backward(torch.ones_like(r40))
    ops.l2norm_bwd(torch.ones_like(x40), y40, inv40, dx40, 77, 40)
    torch.testing.assert_close(dx40.float(), x40f.grad, atol=3e-2 if dtype == 0 else 1e-5, rtol=5e-2 if dtype == 0 else 1e-4)
    a, pe = dev(rnd(6, 40, 64, seed=17)).to(td), dev(rnd(40, 64, seed=18)).to(td)
    o = torch.empty_like(a)
    ops.add_bcast(a, pe, o, a.numel(), pe.numel())
    torch.testing.assert_close(o.float(), (a.float() + pe.float()).to(td).float())
    o4 = torch.empty_like(a)
    ops.add4(a, o, None, a, o4, a.numel())
    torch.testing.assert_close(o4.float(), a.float() * 2 + o.float(), atol=5e-2 if dtype == 0 else 1e-6, rtol=1e-2)
    src = dev(rnd(10, 899, seed=19))
    dst = torch.full((10, 904), 5.0, device="cuda", dtype=td)
    ops.cast2d(src, 899, dst, 904, 10, 899, 904)
    assert torch.equal(dst[:, :899], src.to(td)) and torch.all(dst[:, 899:] == 0)


# ------------------------------------------------------------------------------------------------ prompts
def test_pue_and_disks_bit_exact(ops, golden_dir):
    import os
    import vpu_oracle as vo
    fx = np.load(os.path.join(golden_dir, "pue.npz"))
    lut = dev(torch.from_numpy(vo.click_lut()))
    for key_p, key_o, n in (("points", "pue_click", 24), ("points_n3", "pue_click_n3", 3)):
        P = fx[key_p]
        B = P.shape[0]
        out = torch.zeros(B, 48, 904, device="cuda")
        out64 = torch.zeros(B, 48, 899, device="cuda", dtype=torch.float64)
        ops.pue_encode(dev(torch.from_numpy(P)), None, lut, out, out64, B, n, 24, 448, 904)
        assert np.array_equal(out64.cpu().numpy(), fx[key_o]), "PuE click rows are not bit-exact"
        assert np.array_equal(out[:, :, :899].cpu().numpy().astype(np.float64), fx[key_o])
        assert torch.all(out[:, :, 899:] == 0)
    P, BX = fx["points"], fx["boxes"]
    out64 = torch.zeros(6, 48, 899, device="cuda", dtype=torch.float64)
    out = torch.zeros(6, 48, 904, device="cuda", dtype=torch.bfloat16)
    ops.pue_encode(dev(torch.from_numpy(P)), dev(torch.from_numpy(BX)), lut, out, out64, 6, 24, 24, 448, 904)
    got = out64.cpu().numpy()
    assert np.array_equal(got != 0, fx["pue_box"] != 0), "box rows: support differs"
    np.testing.assert_allclose(got, fx["pue_box"], atol=3e-7, rtol=0)  # float32 exp vs the reference's
    np.testing.assert_allclose(got, vo.pue_box(P, BX), atol=2e-7, rtol=0)
    # edge boxes the golden set does not hold: w in {0,1} (negative sigma quirk), and random ones vs the oracle
    rs = np.random.RandomState(0)
    BX2 = np.stack([rs.randint(0, 448, 64), rs.randint(0, 448, 64), rs.randint(0, 200, 64), rs.randint(0, 200, 64),
                    rs.randint(0, 48, 64)], 1).astype(np.int32)
    BX2[0] = (200, 200, 0, 60, 3); BX2[1] = (200, 200, 1, 60, 30); BX2[2] = (100, 100, 60, 1, 0); BX2[3] = (5, 5, 2, 2, 1)
    P2 = -np.ones((64, 48, 3), np.float32)
    P2[:, 0] = (100, 100, 0)
    out64 = torch.zeros(64, 48, 899, device="cuda", dtype=torch.float64)
    out = torch.zeros(64, 48, 904, device="cuda")
    ops.pue_encode(dev(torch.from_numpy(P2)), dev(torch.from_numpy(BX2)), lut, out, out64, 64, 24, 24, 448, 904)
    ref = vo.pue_box(P2, BX2)
    assert np.array_equal(out64.cpu().numpy() != 0, ref != 0)
    np.testing.assert_allclose(out64.cpu().numpy(), ref, atol=2e-7, rtol=0)

    fd = np.load(os.path.join(golden_dir, "disk.npz"))
    shape = tuple(fd["disks_shape"])
    ref = np.unpackbits(fd["disks_packed"])[:int(np.prod(shape))].reshape(shape).astype(np.float32)
    d = torch.zeros(6, 2, 448, 448, device="cuda")
    ops.disk_maps(dev(torch.from_numpy(fd["points"])), None, d, 6, 24, 448, 448, 5.0)
    assert np.array_equal(d.cpu().numpy(), ref), "disk maps are not bit-exact"
    Ps = fd["points_small"]
    d = torch.zeros(3, 2, 96, 131, device="cuda")
    ops.disk_maps(dev(torch.from_numpy(Ps)), None, d, 3, 5, 96, 131, 5.0)
    assert np.array_equal(d.cpu().numpy(), fd["disks_small"].astype(np.float32))
    # random fractional clicks at full size + box outline, against the oracle
    rs = np.random.RandomState(1)
    Pr = -np.ones((4, 48, 3), np.float32)
    for b in range(4):
        for i in list(range(5)) + list(range(24, 27)):
            Pr[b, i] = (rs.rand() * 447, rs.rand() * 447, i)
    bx = np.array([[224, 200, 101, 60, 2], [30, 400, 50, 50, 30], [0, 0, 0, 0, 0], [440, 440, 30, 30, 1]], np.int32)
    d = torch.zeros(4, 2, 448, 448, device="cuda")
    ops.disk_maps(dev(torch.from_numpy(Pr)), dev(torch.from_numpy(bx)), d, 4, 24, 448, 448, 5.0)
    exp = vo.disk_maps(Pr, 448, 448)
    for b in range(4):
        exp[b] = vo.box_outline(exp[b], bx[b], 24)
    assert np.array_equal(d.cpu().numpy(), exp)


def test_thick_line_rasteriser_equals_opencv_restatement(ops):
    """a3: vpu_draw_polyline / the box outline of vpu_disk_maps against the oracle's restatement of OpenCV's PolyLine ->
    ThickLine -> FillConvexPoly / Line2 / Circle (oracle/vpu_oracle.py), bit for bit: random open poly-lines (short steps as
    the stroke simulator makes them, long slanted ones, repeated points, points outside the canvas, one- and two-point
    curves) and boxes (degenerate, clipped by the border, 1-pixel sides)."""
    import vpu_oracle as vo
    rs = np.random.RandomState(7)
    H, W = 97, 131
    curves = []
    for k in range(24):
        P = 12
        if k < 8:       # a walk with small steps
            pts = np.cumsum(rs.randint(-4, 5, size=(P, 2)), 0) + np.array([W // 2, H // 2])
        elif k < 16:    # long slanted segments, some leaving the canvas
            pts = np.column_stack((rs.randint(-20, W + 20, P), rs.randint(-20, H + 20, P)))
        elif k < 20:    # repeated points and axis-aligned runs
            pts = np.repeat(np.column_stack((rs.randint(0, W, P // 2), rs.randint(0, H, P // 2))), 2, 0)
            pts[1::4, 0] = pts[0::4, 0]
        else:           # everything on the border / in a corner
            pts = np.column_stack((rs.choice([0, 1, W - 2, W - 1], P), rs.choice([0, 1, H - 2, H - 1], P)))
        curves.append(pts.astype(np.int32))
    curves = np.stack(curves)
    d = torch.zeros(len(curves), 2, H, W, device="cuda")
    ops.draw_polyline(torch.from_numpy(curves).cuda(), d, len(curves), curves.shape[1], H, W)
    got = d.cpu().numpy()
    for b in range(len(curves)):
        exp = vo.polyline_raster(np.zeros((2, H, W), np.float32), curves[b])
        assert np.array_equal(got[b], exp), (b, int((got[b] != exp).sum()))
    for P in (1, 2):   # a single point draws nothing; two points draw one segment with both end caps
        d = torch.zeros(1, 2, H, W, device="cuda")
        ops.draw_polyline(torch.from_numpy(curves[:1, :P].copy()).cuda(), d, 1, P, H, W)
        assert np.array_equal(d.cpu().numpy()[0], vo.polyline_raster(np.zeros((2, H, W), np.float32), curves[0, :P]))
    # known answers of the restated algorithm: a horizontal thickness-3 line covers rows y-2 .. y+2 (half width 2^17 in 16.16,
    # Line2 draws the outline rows) with radius-2 caps: 4 cap pixels beyond each end
    m = vo.cv_polyline_mask(20, 30, [(5, 5), (20, 5)], False)
    assert m[3:8, 5:21].all() and m.sum() == 5 * 16 + 8
    boxes = np.array([[60, 40, 50, 30, 0], [60, 40, 51, 31, 30], [3, 3, 20, 20, 1], [128, 95, 30, 30, 40], [50, 50, 0, 0, 2],
                      [50, 50, 1, 1, 2], [70, 20, 2, 40, 25], [65, 48, 200, 150, 3]], np.int32)
    nb = len(boxes)
    none = -torch.ones(nb, 48, 3, device="cuda")
    d = torch.zeros(nb, 2, H, W, device="cuda")
    ops.disk_maps(none, torch.from_numpy(boxes).cuda(), d, nb, 24, H, W, 5.0)
    got = d.cpu().numpy()
    for b in range(nb):
        exp = vo.box_outline(np.zeros((2, H, W), np.float32), boxes[b], 24)
        assert np.array_equal(got[b], exp), (b, boxes[b], int((got[b] != exp).sum()))


# ------------------------------------------------------------------------------------------------ spatial
@pytest.mark.parametrize("dtype", [0, 1])
def test_patch_im2col_and_permute(ops, dtype):
    td = TD[dtype]
    B, H, P, wg = 2, 448, 16, 14
    img4 = dev(rnd(B, 4, H, H, seed=20).abs())
    disks = dev((rnd(B, 2, H, H, seed=21) > 0.9).float())
    g = H // P
    cols = torch.empty(B * g * g, 6 * P * P, device="cuda", dtype=td)
    ops.patch_im2col(img4, disks, cols, B, H, H, P, wg)
    mean = torch.tensor([.485, .456, .406], device="cuda").view(1, 3, 1, 1)
    std = torch.tensor([.229, .224, .225], device="cuda").view(1, 3, 1, 1)
    full = torch.cat([(img4[:, :3] - mean) / std, img4[:, 3:], disks], 1)
    ref = F.unfold(full, P, stride=P).transpose(1, 2)  # [B, T(raster), 6*P*P]
    nw = g // wg
    refw = ref.view(B, nw, wg, nw, wg, -1).permute(0, 1, 3, 2, 4, 5).reshape(B * g * g, -1)
    assert torch.equal(cols.float(), refw.to(td).float())
    # patch 14 (ViT-H): each half (3*14*14 = 588 columns) is zero-padded to 592 so that it starts 16-byte aligned
    P2, wg2 = 14, 16
    g2 = H // P2
    cols2 = torch.empty(B * g2 * g2, 2 * 592, device="cuda", dtype=td)
    ops.patch_im2col(img4, disks, cols2, B, H, H, P2, wg2)
    ref2 = F.unfold(full, P2, stride=P2).transpose(1, 2)
    nw2 = g2 // wg2
    ref2w = ref2.view(B, nw2, wg2, nw2, wg2, -1).permute(0, 1, 3, 2, 4, 5).reshape(B * g2 * g2, -1).to(td).float()
    c2 = cols2.float()
    assert torch.equal(c2[:, :588], ref2w[:, :588]) and torch.equal(c2[:, 592:592 + 588], ref2w[:, 588:])
    assert c2[:, 588:592].abs().max().item() == 0.0 and c2[:, 592 + 588:].abs().max().item() == 0.0
    x = dev(rnd(B, g * g, 64, seed=22)).to(td)
    y, z = torch.empty_like(x), torch.empty_like(x)
    ops.window_permute(x, y, B, g, wg, 64, to_raster=False)
    xw = x.view(B, nw, wg, nw, wg, 64).permute(0, 1, 3, 2, 4, 5).reshape(B, g * g, 64)
    assert torch.equal(y, xw)
    ops.window_permute(y, z, B, g, wg, 64, to_raster=True)
    assert torch.equal(z, x)


@pytest.mark.parametrize("dtype", [0, 1])
def test_pixel_shuffle_convs(ops, dtype):
    """ConvTranspose2d(2,2) and Conv2d(2,2) expressed as GEMM + pixel shuffle vs torch."""
    td = TD[dtype]
    B, h, Cin, Cout = 2, 6, 32, 16
    x = dev(rnd(B, h * h, Cin, seed=23)).to(td)           # channels-last tokens
    Wt = dev(rnd(Cin, Cout, 2, 2, seed=24, scale=0.3)).to(td)
    bias = dev(rnd(Cout, seed=25))
    tmp = torch.empty(B * h * h, Cout * 4, device="cuda", dtype=td)
    ops.gemm(x, Wt, tmp, B * h * h, Cout * 4, Cin, Cin, Cout * 4, Cout * 4, dtype, transB=True)
    out = torch.empty(B, 2 * h, 2 * h, Cout, device="cuda", dtype=td)
    ops.pixel_shuffle2(tmp, out, bias, B, h, h, Cout)
    xin = x.float().view(B, h, h, Cin).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(xin, Wt.float(), bias, stride=2).permute(0, 2, 3, 1)
    tol = dict(atol=4e-2, rtol=2e-2) if dtype == 0 else dict(atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(out.float(), ref, **tol)
    # inverse shuffle (space to depth) == im2col of Conv2d(2, stride 2)
    Wc = dev(rnd(Cout, Cin, 2, 2, seed=26, scale=0.3)).to(td)
    fine = dev(rnd(B, 2 * h, 2 * h, Cin, seed=27)).to(td)
    s2d = torch.empty(B * h * h, Cin * 4, device="cuda", dtype=td)
    ops.pixel_shuffle2(fine, s2d, None, B, h, h, Cin, inverse=True)
    o2 = torch.empty(B * h * h, Cout, device="cuda", dtype=td)
    ops.gemm(s2d, Wc, o2, B * h * h, Cout, Cin * 4, Cin * 4, Cin * 4, Cout, dtype)
    ref2 = F.conv2d(fine.float().permute(0, 3, 1, 2), Wc.float(), None, stride=2).permute(0, 2, 3, 1).reshape(B * h * h, Cout)
    torch.testing.assert_close(o2.float(), ref2, **tol)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("C,gelu", [(16, 1), (192, 0), (384, 1), (1536, 0), (2560, 0), (2560, 1)])
def test_groupnorm(ops, dtype, C, gelu):
    td = TD[dtype]
    B, HW = 2, 14 * 14 if C > 1000 else 28 * 28
    x = dev(rnd(B, HW, C, seed=28, scale=1.5) + 0.2).to(td)
    w, b = dev(1 + rnd(C, seed=29, scale=0.2)), dev(rnd(C, seed=30, scale=0.2))
    nch = ops.groupnorm_nchunk()
    stats = torch.zeros(B, nch, 2, device="cuda", dtype=torch.float64)
    mean, rstd = torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    y = torch.empty_like(x)
    ops.groupnorm_fwd(x, w, b, y, mean, rstd, stats, B, HW, C, 1e-5, gelu)
    xf = x.float().requires_grad_(True)
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.group_norm(xf.transpose(1, 2), 1, wf, bf, 1e-5)
    if gelu:
        ref = F.gelu(ref)
    ref = ref.transpose(1, 2)
    tol = dict(atol=3e-2, rtol=2e-2) if dtype == 0 else dict(atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(y.float(), ref, **tol)
    dy = dev(rnd(B, HW, C, seed=31)).to(td)
    ref.backward(dy.float())
    part = torch.zeros(B * nch, 2, C, device="cuda")
    dx = torch.empty_like(x)
    ops.groupnorm_bwd(dy, x, w, b, mean, rstd, dx, part, stats, B, HW, C, gelu)
    dwb = torch.zeros(2 * C, device="cuda")
    ops.colsum_f32(part, dwb, B * nch, 2 * C)
    dw, db = dwb[:C], dwb[C:]
    torch.testing.assert_close(dx.float(), xf.grad, atol=3e-2 if dtype == 0 else 2e-5, rtol=3e-2 if dtype == 0 else 1e-4)
    gt = dict(atol=0.3, rtol=3e-2) if dtype == 0 else dict(atol=2e-4, rtol=1e-4)
    torch.testing.assert_close(dw, wf.grad, **gt)
    torch.testing.assert_close(db, bf.grad, **gt)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("h", [14, 28])
def test_bilinear_backward_sliced_windows(ops, dtype, h, monkeypatch):
    """C = 256, ratio 8 (and 4 with VPU_BILINEAR_SPLIT_R=4 set before the library is first used): the gather window of
    an input pixel is cut into four row slices summed in a fixed order -- same result as autograd of F.interpolate."""
    td = TD[dtype]
    B, C, H = 2, 256, 112
    dout = dev(rnd(B, H, H, C, seed=80)).to(td)
    xf = torch.zeros(B, C, h, h, device="cuda", requires_grad=True)
    F.interpolate(xf, size=(H, H), mode="bilinear", align_corners=False).backward(dout.float().permute(0, 3, 1, 2))
    din = torch.empty(B, h, h, C, device="cuda", dtype=td)
    ops.bilinear_cl_bwd(dout, C, din, C, B, h, h, H, H, C, dtype)
    torch.testing.assert_close(din.float(), xf.grad.permute(0, 2, 3, 1), atol=0.25 if dtype == 0 else 2e-5,
                               rtol=3e-2 if dtype == 0 else 1e-5)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("h", [14, 28, 56, 112])
def test_bilinear_channels_last(ops, dtype, h):
    td = TD[dtype]
    B, C, H = 2, 16, 112
    x = dev(rnd(B, h, h, C, seed=32)).to(td)
    out = torch.zeros(B, H, H, 48, device="cuda", dtype=td)  # lands in a channel slice [16:32] of a wider map
    ops.bilinear_cl_fwd(x, C, (out, 16), 48, B, h, h, H, H, C, dtype)
    xf = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.interpolate(xf, size=(H, H), mode="bilinear", align_corners=False)
    tol = dict(atol=2e-2, rtol=2e-2) if dtype == 0 else dict(atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(out[..., 16:32].float(), ref.permute(0, 2, 3, 1), **tol)
    assert torch.all(out[..., :16] == 0) and torch.all(out[..., 32:] == 0)
    dout = torch.zeros(B, H, H, 48, device="cuda", dtype=td)
    dout[..., 16:32] = dev(rnd(B, H, H, C, seed=33)).to(td)
    ref.backward(dout[..., 16:32].float().permute(0, 3, 1, 2))
    din = torch.empty_like(x)
    ops.bilinear_cl_bwd((dout, 16), 48, din, C, B, h, h, H, H, C, dtype)
    torch.testing.assert_close(din.float(), xf.grad.permute(0, 2, 3, 1), atol=0.15 if dtype == 0 else 1e-5, rtol=3e-2 if dtype == 0 else 1e-5)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("n", [0, 1, 2, 3])
def test_upsum_relu_equals_fusion_over_concat(ops, dtype, n):
    """vpu_upsum_relu (head fusion by linearity): relu(t0 + sum_i resize(z_i)) in place == the same sum built from
    F.interpolate(align_corners=False), and -- the identity the engine relies on -- a 1x1 convolution over the channel
    concat of resized maps == the sum of the resized per-map products (fp32: to rounding)."""
    td = TD[dtype]
    B, C, H = 2, 32, 112
    sizes = [48] if n == 1 else [56, 28, 14][:n]    # 112 / 48: not an integer ratio -> the one-pixel-per-lane kernel
    t0 = dev(rnd(B, H, H, C, seed=40)).to(td)
    zs = [dev(rnd(B, s, s, C, seed=41 + i)).to(td) for i, s in enumerate(sizes)]
    ref = t0.float()
    for z in zs:
        ref = ref + F.interpolate(z.float().permute(0, 3, 1, 2), size=(H, H), mode="bilinear",
                                  align_corners=False).permute(0, 2, 3, 1)
    ref = torch.relu(ref)
    io = t0.clone()
    ops.upsum_relu(io, [(z, s, s) for z, s in zip(zs, sizes)], B, H, H, C, dtype)
    tol = dict(atol=3e-2, rtol=2e-2) if dtype == 0 else dict(atol=2e-6, rtol=1e-6)
    torch.testing.assert_close(io.float(), ref, **tol)
    if dtype == 1 and n == 3:
        g = torch.Generator().manual_seed(50)
        ys = [dev(rnd(B, s, s, 16, seed=60 + i)) for i, s in enumerate([112] + sizes)]
        Wf = dev(torch.randn(C, 64, generator=g) * 0.2)
        bias = dev(torch.randn(C, generator=g))
        cat = torch.cat([F.interpolate(y.permute(0, 3, 1, 2), size=(H, H), mode="bilinear", align_corners=False)
                         for y in ys], 1)
        want = torch.relu(F.conv2d(cat, Wf[:, :, None, None], bias)).permute(0, 2, 3, 1)
        io = (ys[0] @ Wf[:, :16].t() + bias).contiguous()
        lows = [(ys[i] @ Wf[:, 16 * i:16 * (i + 1)].t()).contiguous() for i in (1, 2, 3)]
        ops.upsum_relu(io, [(z, s, s) for z, s in zip(lows, sizes)], B, H, H, C, dtype)
        torch.testing.assert_close(io, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("dtype", [0, 1])
def test_batched_cast_and_fanout_add(ops, dtype):
    """vpu_cast2d_batched (the engine's per-step derived operands in one launch: strided casts, a two-source sum, zero-filled pad
    columns, the raster -> window row map) against vpu_cast2d / vpu_add4 / vpu_window_permute job by job, bit for bit; and
    vpu_fanout_add (the adjoint of add4: up to four destinations overwritten or accumulated in one launch) against add4."""
    td = TD[dtype]
    D, k3, k3p, E, Ep, g, wg = 64, 147, 152, 203, 208, 8, 4
    w1, w2 = dev(rnd(D, k3, seed=1)), dev(rnd(D, k3, seed=2))
    b1, b2 = dev(rnd(D, seed=3)), dev(rnd(D, seed=4))
    lin, pos = dev(rnd(32, E, seed=5)), dev(rnd(g * g + 1, D, seed=6))
    fused, bsum = torch.zeros(D, 2 * k3p, device="cuda", dtype=td), torch.empty(D, device="cuda")
    linp, posw = torch.full((32, Ep), 7.0, device="cuda", dtype=td), torch.empty(g * g, D, device="cuda", dtype=td)
    ops.cast2d_batched([
        dict(src=w1, dst=(fused, 0), ld_src=k3, ld_dst=2 * k3p, rows=D, cols=k3),
        dict(src=w2, dst=(fused, k3p), ld_src=k3, ld_dst=2 * k3p, rows=D, cols=k3),
        dict(src=b1, src2=b2, dst=bsum, ld_src=D, ld_dst=D, rows=1, cols=D),
        dict(src=lin, dst=linp, ld_src=E, ld_dst=Ep, rows=32, cols=E, cols_pad=Ep),
        dict(src=(pos, D), dst=posw, ld_src=D, ld_dst=D, rows=g * g, cols=D, perm=(g, wg)),
    ])
    ref_f = torch.zeros_like(fused)
    ops.cast2d(w1, k3, (ref_f, 0), 2 * k3p, D, k3)
    ops.cast2d(w2, k3, (ref_f, k3p), 2 * k3p, D, k3)
    ref_b = torch.empty_like(bsum)
    ops.add4(b1, b2, None, None, ref_b, D)
    ref_l = torch.empty_like(linp)
    ops.cast2d(lin, E, ref_l, Ep, 32, E, Ep)
    tmp, ref_p = torch.empty(g * g, D, device="cuda", dtype=td), torch.empty_like(posw)
    ops.cast2d((pos, D), D, tmp, D, g * g, D)
    ops.window_permute(tmp, ref_p, 1, g, wg, D, to_raster=False)
    assert torch.equal(fused, ref_f) and torch.equal(bsum, ref_b) and torch.equal(linp, ref_l) and torch.equal(posw, ref_p)
    src = dev(rnd(48, 64, seed=7)).to(td)
    dsts = [dev(rnd(48, 64, seed=8 + i)).to(td) for i in range(4)]
    want = [d.clone() for d in dsts]
    acc = [True, False, True, False]
    for d, a in zip(want, acc):
        if a:
            ops.add4(d, src, None, None, d, d.numel())
        else:
            d.copy_(src)
    ops.fanout_add(src, dsts, acc, src.numel())
    assert all(torch.equal(d, w) for d, w in zip(dsts, want))
    one = dev(rnd(48, 64, seed=20)).to(td)
    w1_ = one.clone()
    ops.add4(w1_, src, None, None, w1_, one.numel())
    ops.fanout_add(src, [one], [True], src.numel())
    assert torch.equal(one, w1_)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("C", [64, 768, 1280])
def test_batched_gates_equal_single_gate_launches(ops, dtype, C):
    """vpu_gate_fwd_n / vpu_gate_bwd_n (round 5: the three gates of SimpleFPN in one launch per pass) against the single-gate
    kernels, gate for gate: statistics, arg-max indices and gated maps bit for bit; dQ / dK bit for bit (same operations in
    the same order); dx = the fp32 sum over the three gates rounded once -- equal to the chained single-gate launches in
    fp32, within their extra bf16 roundings otherwise -- and against torch autograd.  C = 768 / 1280: two / three 512-column
    chunks per row (ViT-B / ViT-H)."""
    td = TD[dtype]
    B, nq, N, n = 2, 48, 196, 3
    x = dev(rnd(B, N, C, seed=50)).to(td)
    Qs = [dev(rnd(B, nq, C, seed=51 + i)).to(td) for i in range(n)]
    Ks = [dev(rnd(B, N, C, seed=61 + i)).to(td) for i in range(n)]
    cg, sg = torch.empty(n, B, C, device="cuda"), torch.empty(n, B, N, device="cuda")
    aq = torch.empty(n, B, C, device="cuda", dtype=torch.int32)
    ac = torch.empty(n, B, N, device="cuda", dtype=torch.int32)
    outs = [torch.empty_like(x) for _ in range(n)]
    ops.gate_fwd_n(Qs, Ks, x, outs, cg, aq, sg, ac, B, nq, N, C)
    for i in range(n):
        cg1, sg1 = torch.empty(B, C, device="cuda"), torch.empty(B, N, device="cuda")
        aq1, ac1 = torch.empty(B, C, device="cuda", dtype=torch.int32), torch.empty(B, N, device="cuda", dtype=torch.int32)
        ops.gate_stats(Qs[i], Ks[i], cg1, aq1, sg1, ac1, B, nq, N, C)
        o1 = torch.empty_like(x)
        ops.gate_apply(x, cg1, sg1, o1, B, N, C)
        assert torch.equal(cg[i], cg1) and torch.equal(sg[i], sg1) and torch.equal(aq[i], aq1) and torch.equal(ac[i], ac1)
        assert torch.equal(outs[i], o1)
    douts = [dev(rnd(B, N, C, seed=71 + i)).to(td) for i in range(n)]
    for accum in (False, True):
        dx0 = dev(rnd(B, N, C, seed=80)).to(td)
        dx = dx0.clone()
        dQs = [dev(rnd(B, nq, C, seed=81 + i)).to(td) for i in range(n)]
        dKs = [dev(rnd(B, N, C, seed=91 + i)).to(td) for i in range(n)]
        dQ1, dK1 = [t.clone() for t in dQs], [t.clone() for t in dKs]
        ops.gate_bwd_n(douts, x, cg, aq, sg, ac, dx, accum, dQs, dKs, torch.empty(n, B, 64, C, device="cuda"), B, nq, N, C)
        dx1 = dx0.clone()
        for i in range(n):
            ops.gate_bwd(douts[i], x, cg[i], aq[i], sg[i], ac[i], dx1, accum or i > 0, dQ1[i], dK1[i],
                         torch.empty(B, 64, C, device="cuda"), B, nq, N, C)
            assert torch.equal(dQs[i], dQ1[i]) and torch.equal(dKs[i], dK1[i]), i
        xf = x.float().clone().requires_grad_(True)       # (a fresh leaf per pass: x.float() IS x in fp32)
        ref = sum((xf * (1 + cg[i].unsqueeze(1) + sg[i].unsqueeze(2)) * douts[i].float()).sum() for i in range(n))
        ref.backward()
        want = xf.grad + (dx0.float() if accum else 0)
        if dtype == 1:
            torch.testing.assert_close(dx, dx1, atol=1e-5, rtol=1e-5)
            torch.testing.assert_close(dx, want, atol=1e-5, rtol=1e-5)
        else:       # one rounding instead of three: closer to the fp32 sum than the chained launches are
            e_n, e_1 = (dx.float() - want).abs().max().item(), (dx1.float() - want).abs().max().item()
            assert e_n <= e_1 + 1e-6 and e_n < 4e-2 * want.abs().max().item(), (e_n, e_1)


@pytest.mark.parametrize("dtype", [0, 1])
def test_gates_and_convseg(ops, dtype):
    td = TD[dtype]
    B, nq, N, C = 2, 48, 784, 64
    x, Q, K = dev(rnd(B, N, C, seed=34)).to(td), dev(rnd(B, nq, C, seed=35)).to(td), dev(rnd(B, N, C, seed=36)).to(td)
    cg, sg = torch.empty(B, C, device="cuda"), torch.empty(B, N, device="cuda")
    aq = torch.empty(B, C, device="cuda", dtype=torch.int32)
    ac = torch.empty(B, N, device="cuda", dtype=torch.int32)
    ops.gate_stats(Q, K, cg, aq, sg, ac, B, nq, N, C)
    out = torch.empty_like(x)
    ops.gate_apply(x, cg, sg, out, B, N, C)
    xf, Qf, Kf = (t.float().requires_grad_(True) for t in (x, Q, K))
    ref = xf + xf * Qf.max(1).values.sigmoid().unsqueeze(1) + xf * Kf.max(2).values.sigmoid().unsqueeze(2)
    tol = dict(atol=3e-2, rtol=2e-2) if dtype == 0 else dict(atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(out.float(), ref, **tol)
    dout = dev(rnd(B, N, C, seed=37)).to(td)
    ref.backward(dout.float())
    dx = dev(rnd(B, N, C, seed=38)).to(td)
    dx0 = dx.float().clone()
    dQ, dK = torch.zeros_like(Q), torch.zeros_like(K)
    part = torch.zeros(B, 64, C, device="cuda")
    ops.gate_bwd(dout, x, cg, aq, sg, ac, dx, True, dQ, dK, part, B, nq, N, C)
    bt = dict(atol=6e-2, rtol=3e-2) if dtype == 0 else dict(atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(dx.float(), dx0 + xf.grad, **bt)
    torch.testing.assert_close(dQ.float(), Qf.grad, atol=0.2 if dtype == 0 else 1e-4, rtol=3e-2 if dtype == 0 else 1e-4)
    torch.testing.assert_close(dK.float(), Kf.grad, **bt)
    # conv_seg with a Dropout2d mask
    rows, HW, Cc = B * 100, 100, 256
    f = dev(rnd(rows, Cc, seed=39)).to(td)
    w, bias = dev(rnd(Cc, seed=40, scale=0.1)), dev(rnd(1, seed=41))
    mask = dev((rnd(B, Cc, seed=42) > -0.8).float() / 0.9)
    o = torch.empty(rows, device="cuda")
    ops.convseg_fwd(f, w, bias, mask, o, rows, HW, Cc)
    ff, wf = f.float().requires_grad_(True), w.clone().requires_grad_(True)
    refo = ((ff.view(B, HW, Cc) * mask.view(B, 1, Cc)) * wf).sum(-1).view(rows) + bias
    torch.testing.assert_close(o, refo, atol=1e-4, rtol=1e-4)
    f40, w40, o40 = dev(rnd(rows, 40, seed=45)).to(td), dev(rnd(40, seed=46, scale=0.1)), torch.empty(rows, device="cuda")
    ops.convseg_fwd(f40, w40, bias, None, o40, rows, HW, 40)       # C = 40: the one-element-per-lane form, no mask
    torch.testing.assert_close(o40, (f40.float() * w40).sum(-1) + bias, atol=1e-4, rtol=1e-4)
    do = dev(rnd(rows, seed=43))
    refo.backward(do)
    nb = ops.convseg_bwd_nblk(rows)
    part, part_b = torch.zeros(nb, Cc, device="cuda"), torch.zeros(nb, device="cuda")
    dfe = torch.empty_like(f)
    ops.convseg_bwd(do, f, w, mask, dfe, False, part, part_b, rows, HW, Cc)
    torch.testing.assert_close(dfe.float(), ff.grad, atol=2e-3 if dtype == 0 else 1e-6, rtol=2e-2 if dtype == 0 else 1e-5)
    torch.testing.assert_close(part.sum(0), wf.grad, atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(part_b.sum(), do.sum(), atol=1e-3, rtol=1e-4)
    if dtype == 0:   # vpu_head_grad_fused == l2norm_bwd followed by convseg_bwd(accum = 3), one pass
        yn, invn = torch.empty_like(f), torch.empty(rows, device="cuda")
        ops.l2norm_fwd(f, yn, invn, rows, Cc)
        dfn = dev(rnd(rows, Cc, seed=48)).to(td)
        two = torch.empty_like(f)
        ops.l2norm_bwd(dfn, yn, invn, two, rows, Cc)
        p2, pb2 = torch.zeros(nb, Cc, device="cuda"), torch.zeros(nb, device="cuda")
        ops.convseg_bwd(do, f, w, mask, two, 3, p2, pb2, rows, HW, Cc)
        one, p1, pb1 = torch.empty_like(f), torch.zeros(nb, Cc, device="cuda"), torch.zeros(nb, device="cuda")
        ops.head_grad_fused(dfn, yn, invn, do, f, w, mask, one, p1, pb1, rows, HW, Cc)
        torch.testing.assert_close(one.float(), two.float(), atol=2e-2, rtol=2e-2)   # (the two-pass form rounds to bf16 in between)
        torch.testing.assert_close(p1, p2, atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(pb1, pb2, atol=1e-5, rtol=1e-5)
    # accum bit 1: the gradient leaves multiplied by [x > 0] (x = output of a ReLU); bit 0 adds into dx first
    base = dev(rnd(rows, Cc, seed=47)).to(td)
    dfm = base.clone()
    ops.convseg_bwd(do, f, w, mask, dfm, 3, part, part_b, rows, HW, Cc)
    want = (ff.grad + base.float()) * (f.float() > 0)
    torch.testing.assert_close(dfm.float(), want, atol=2e-2 if dtype == 0 else 1e-6, rtol=2e-2 if dtype == 0 else 1e-5)


def test_upsample_and_losses(ops):
    import vpu_oracle as vo
    B, S, h, H = 2, 8, 28, 112
    low = dev(torch.sigmoid(rnd(B, S, h, h, seed=44, scale=3.0)))
    up = torch.empty(B, S, H, H, device="cuda")
    ops.upsample_ac_fwd(low, up, B * S, h, h, H, H)
    lf = low.clone().requires_grad_(True)
    ref = F.interpolate(lf, size=(H, H), mode="bilinear", align_corners=True)
    torch.testing.assert_close(up, ref, atol=1e-6, rtol=1e-6)
    gt = dev((rnd(B, 1, H, H, seed=45) > 0.2).float())
    ed = vo.ed_mask_label(gt, S // 2)
    ov = dev((rnd(2, H, H, seed=46) > 0.5).float())
    idx = -torch.ones(B, S, dtype=torch.int32)
    idx[0, 1] = 0; idx[1, 6] = 1
    ed[0, 1] = ov[0]; ed[1, 6] = ov[1]
    loss = vo.bce_from_sigmoid(ref, ed).mean() * 2.0
    loss.backward()
    part = torch.empty(B, S, device="cuda")
    dprob = torch.empty_like(up)
    ops.p2cl_fwd_bwd(up, gt, dev(idx), ov, part, dprob, 2.0 / (B * S * H * H), B, S, H, H)
    torch.testing.assert_close(part.sum() * 2.0 / (B * S * H * H), loss.detach(), rtol=1e-5, atol=1e-6)
    dlow = torch.empty_like(low)
    ops.upsample_ac_bwd(dprob, dlow, B * S, h, h, H, H)
    torch.testing.assert_close(dlow, lf.grad, rtol=1e-4, atol=1e-9)
    # NFL + Dice
    logits = dev(rnd(B, 1, H, H, seed=47, scale=4.0)).requires_grad_(True)
    l_n, l_d = vo.nfl_loss(logits, gt).mean(), vo.dice_loss_naive(logits, gt)
    (1.5 * l_n + 0.7 * l_d).backward()
    out = torch.empty(B, 2, device="cuda")
    dl = torch.empty(B, H * H, device="cuda")
    ops.nfl_dice_fwd_bwd(logits.detach(), gt, None, out, dl, 1.5 / B, 0.7 / B, B, H * H)
    torch.testing.assert_close(out[:, 0].mean(), l_n.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(out[:, 1].mean(), l_d.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(dl.view_as(logits), logits.grad, rtol=2e-4, atol=1e-9)


def test_adam_matches_torch(ops):
    n = 10007
    p0, g = rnd(n, seed=48), rnd(n, seed=49, scale=0.1)
    p = dev(p0.clone())
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sh = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=5e-5, betas=(0.9, 0.999), eps=1e-8)
    for step in range(1, 4):
        pt.grad = g.clone() * step
        opt.step()
        ops.adam_step(p, dev(g * step), m, v, sh, n, 5e-5, 0.9, 0.999, 1e-8, 0.0, step)
    torch.testing.assert_close(p.cpu(), pt.detach(), rtol=1e-6, atol=1e-7)
    assert torch.equal(sh, p.to(torch.bfloat16))


@pytest.mark.parametrize("decoupled", [0, 1])
def test_adam_groups_matches_torch(ops, decoupled):
    """Per-tensor learning rate / weight decay in ONE launch == torch.optim.Adam / AdamW with one param group per tensor
    (three tensors at 8-element-aligned offsets of a flat buffer, the padding in between stays untouched)."""
    sizes, offs = [1000, 37, 4096], [0, 1000, 1040]
    total = 1040 + 4096
    lrs, wds = [5e-5, 5e-5 * 0.75 ** 3, 0.0], [0.02, 0.0, 0.02]
    flat0 = rnd(total, seed=52)
    g = rnd(total, seed=53, scale=0.1)
    p = dev(flat0.clone())
    m, v = torch.zeros(total, device="cuda"), torch.zeros(total, device="cuda")
    sh = torch.zeros(total, device="cuda", dtype=torch.bfloat16)
    seg_end = torch.tensor([1000, 1040, total], dtype=torch.int64, device="cuda")
    seg_lr, seg_wd = torch.tensor(lrs, device="cuda"), torch.tensor(wds, device="cuda")
    params = [torch.nn.Parameter(flat0[o:o + n].clone()) for o, n in zip(offs, sizes)]
    cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    opt = cls([{"params": [q], "lr": lr, "weight_decay": wd} for q, lr, wd in zip(params, lrs, wds)],
              lr=5e-5, betas=(0.9, 0.999), eps=1e-8)
    for step in range(1, 4):
        for q, o, n in zip(params, offs, sizes):
            q.grad = g[o:o + n].clone() * step
        opt.step()
        ops.adam_step_groups(p, dev(g * step), m, v, sh, total, seg_end, seg_lr, seg_wd, 3, 0.9, 0.999, 1e-8, decoupled,
                             step)
    for q, o, n in zip(params, offs, sizes):
        torch.testing.assert_close(p[o:o + n].cpu(), q.detach(), rtol=1e-6, atol=1e-7)
    assert torch.equal(sh, p.to(torch.bfloat16))


def test_adam_hyper_replayed_from_a_graph_matches_torch(ops):
    """vpu_adam_step_hyper reads {lr, 1-b1^t, sqrt(1-b2^t), grad_scale} from device memory: ONE captured launch, replayed
    with the scalars (and the gradient buffer) refreshed before every replay, must equal torch.optim.Adam with a
    per-tensor lr scale / weight decay table -- the capturable path of pvpuformer_amd.optim.FusedAdam."""
    sizes, offs = [1000, 37, 4096], [0, 1000, 1040]
    total = 1040 + 4096
    scales, wds, base_lr = [1.0, 0.75 ** 3, 0.5], [0.02, 0.0, 0.02], 5e-5
    flat0 = rnd(total, seed=54)
    g = rnd(total, seed=55, scale=0.1)
    p = dev(flat0.clone())
    gbuf = torch.zeros(total, device="cuda")
    m, v = torch.zeros(total, device="cuda"), torch.zeros(total, device="cuda")
    sh = torch.zeros(total, device="cuda", dtype=torch.bfloat16)
    seg_end = torch.tensor([1000, 1040, total], dtype=torch.int64, device="cuda")
    seg_sc, seg_wd = torch.tensor(scales, device="cuda"), torch.tensor(wds, device="cuda")
    hyper = torch.zeros(4, device="cuda")
    params = [torch.nn.Parameter(flat0[o:o + n].clone()) for o, n in zip(offs, sizes)]
    opt = torch.optim.Adam([{"params": [q], "lr": base_lr * sc, "weight_decay": wd} for q, sc, wd in zip(params, scales, wds)],
                           lr=base_lr, betas=(0.9, 0.999), eps=1e-8)

    def launch():
        ops.adam_step_hyper(p, gbuf, m, v, sh, total, hyper, seg_end, seg_sc, seg_wd, 3, 0.9, 0.999, 1e-8, 0.0, False)

    graph = None
    for step in range(1, 5):
        for q, o, n in zip(params, offs, sizes):
            q.grad = g[o:o + n].clone() * step
        opt.step()
        # the summed gradient of "2 ranks": grad_scale 0.5 undoes it
        gbuf.copy_(dev(g * step * 2.0))
        hyper.copy_(torch.tensor([base_lr, 1 - 0.9 ** step, (1 - 0.999 ** step) ** 0.5, 0.5]))
        if step == 1:
            launch()                      # eager first: nothing to replay yet
        elif graph is None:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            keep = [t.clone() for t in (p, m, v)]
            with torch.cuda.graph(graph):
                launch()
            for t, k in zip((p, m, v), keep):   # capture enqueues nothing; state unchanged
                assert torch.equal(t, k)
            graph.replay()
        else:
            graph.replay()
    torch.cuda.synchronize()
    for q, o, n in zip(params, offs, sizes):
        torch.testing.assert_close(p[o:o + n].cpu(), q.detach(), rtol=2e-6, atol=1e-7)
    assert torch.equal(sh, p.to(torch.bfloat16))


@pytest.mark.parametrize("tA,tB", [(1, 1), (0, 0), (0, 1)])
def test_gemm_split_k_and_vector_epilogue(ops, tA, tB):
    """Long-K / few-tile shapes take the split-K path (fp32 slabs + ordered reduce); the result must equal the
    unsplit launch bit for bit on exact-integer data and stay within bf16 tolerance on random data, for plain, ACCUM
    and full-epilogue launches.  Also ragged N (scalar tail of the vector epilogue)."""
    g = torch.Generator().manual_seed(3)
    M, N, K = 256, 136, 8192
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    Bm = torch.randint(-2, 3, (N, K), generator=g).float()
    ref = A @ Bm.t()
    Ad = dev(A.t().contiguous() if tA else A).to(torch.bfloat16)
    Bd = dev(Bm.t().contiguous() if tB else Bm).to(torch.bfloat16)
    lda, ldb = (M if tA else K), (N if tB else K)
    for ws in ("auto", None):
        C32 = torch.full((M, N), 1.0, device="cuda")
        ops.gemm(Ad, Bd, C32, M, N, K, lda, ldb, N, 0, transA=bool(tA), transB=bool(tB),
                 flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, workspace=ws)
        assert torch.equal(C32.cpu(), ref + 1.0), ws
    bias, R = dev(rnd(N, seed=50)), dev(rnd(M, N, seed=51)).to(torch.bfloat16)
    outs = []
    for ws in ("auto", None):
        Cb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(Ad, Bd, Cb, M, N, K, lda, ldb, N, 0, transA=bool(tA), transB=bool(tB),
                 flags=ops.EPI_BIAS | ops.EPI_RESID, bias=bias, resid=R, ldr=N, alpha=1.0 / 64, workspace=ws)
        outs.append(Cb.float().cpu())
    expect = (ref / 64 + bias.cpu() + R.float().cpu())
    torch.testing.assert_close(outs[0], expect, atol=0.2, rtol=2e-2)
    torch.testing.assert_close(outs[0], outs[1], atol=0.13, rtol=1e-2)


@pytest.mark.parametrize("shape", [(1, 1, 768, 768, 9408), (1, 1, 384, 768, 2048), (0, 0, 576, 768, 768), (0, 1, 576, 384, 1024)])
def test_gemm_split_k_inlaunch_combine_stress(ops, shape):
    """Split-K combined inside the launch (the last-arriving slice of a tile sums the write-through slabs after an
    agent-scope acquire) against the separate reduce launch: 25 back-to-back launches on FRESH exact-integer operands
    that reuse the same workspace (so a stale cached slab line would show), every result bit-identical to the fp32
    matmul, including the fused bias-gradient column sums of the weight-gradient form."""
    tA, tB, M, N, K = shape
    g = torch.Generator().manual_seed(11)
    for it in range(25):
        A = torch.randint(-2, 3, (M, K), generator=g).float()
        Bm = torch.randint(-2, 3, (N, K), generator=g).float()
        ref = A @ Bm.t()
        Ad = dev(A.t().contiguous() if tA else A).to(torch.bfloat16)
        Bd = dev(Bm.t().contiguous() if tB else Bm).to(torch.bfloat16)
        lda, ldb = (M if tA else K), (N if tB else K)
        outs = []
        for mode in (1, 0):
            ops.gemm_set_option("splitk_inlaunch", mode)
            ops.gemm_set_option("skinny", 0)      # (the 576-row shapes would otherwise take the one-launch skinny kernel)
            try:
                C32 = torch.full((M, N), float(it), device="cuda")
                cs = torch.full((M,), 2.0, device="cuda") if (tA and tB) else None
                ops.gemm(Ad, Bd, C32, M, N, K, lda, ldb, N, 0, transA=bool(tA), transB=bool(tB),
                         flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=cs)
            finally:
                ops.gemm_set_option("splitk_inlaunch", -1)
                ops.gemm_set_option("skinny", -1)
            outs.append((C32, cs))
        torch.cuda.synchronize()
        for C32, cs in outs:
            assert torch.equal(C32.cpu(), ref + float(it)), (it, (C32.cpu() - ref - it).abs().max())
            if cs is not None:
                assert torch.equal(cs.cpu(), A.sum(1) + 2.0), it


@pytest.mark.parametrize("tB", [0, 1])
@pytest.mark.parametrize("shape", [(576, 768, 768), (576, 384, 768), (576, 2048, 904), (576, 768, 2048), (96, 768, 384),
                                   (50, 136, 200), (1000, 72, 4096)])
def test_gemm_skinny_kernel(ops, tB, shape):
    """The one-launch kernel for few-row problems (64x64 tile per workgroup, its four waves split K, partial tiles summed
    through LDS in wave order): bit-exact on exact-integer data for plain / K-major B, ragged M, N and K (K % 8 == 0),
    fp32 ACCUM output; and equal to the split-K path (same fp32 sums up to their order) through the full epilogue
    (bias, ReLU / residual, bf16 output) on random data."""
    M, N, K = shape
    g = torch.Generator().manual_seed(17)
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    Bm = torch.randint(-2, 3, (N, K), generator=g).float()
    ref = A @ Bm.t()
    Ad = dev(A).to(torch.bfloat16)
    Bd = dev(Bm.t().contiguous() if tB else Bm).to(torch.bfloat16)
    ldb = N if tB else K
    try:
        for sk in (1, 0):
            ops.gemm_set_option("skinny", sk)
            C32 = torch.full((M, N), 3.0, device="cuda")
            ops.gemm(Ad, Bd, C32, M, N, K, K, ldb, N, 0, transB=bool(tB), flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM)
            assert torch.equal(C32.cpu(), ref + 3.0), (sk, float((C32.cpu() - ref - 3.0).abs().max()))
        Ar, Br = dev(rnd(M, K, seed=70)).to(torch.bfloat16), dev(rnd(K, N, seed=71) if tB else rnd(N, K, seed=71)).to(torch.bfloat16)
        bias, R = dev(rnd(N, seed=72)), dev(rnd(M, N, seed=73)).to(torch.bfloat16)
        for flags in (ops.EPI_BIAS | ops.EPI_RELU, ops.EPI_BIAS | ops.EPI_RESID, ops.EPI_DRELU):
            outs = []
            for sk in (1, 0):
                ops.gemm_set_option("skinny", sk)
                Cb = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
                ops.gemm(Ar, Br, Cb, M, N, K, K, ldb, N, 0, transB=bool(tB), flags=flags, bias=bias, resid=R, ldr=N,
                         aux=R, ldaux=N)
                outs.append(Cb.float())
            torch.testing.assert_close(outs[0], outs[1], atol=0.07, rtol=1e-2)
            want = Ar.float() @ (Br.float() if tB else Br.float().t())
            if flags & ops.EPI_BIAS:
                want = want + bias
            if flags & ops.EPI_RELU:
                want = torch.relu(want)
            if flags & ops.EPI_RESID:
                want = want + R.float()
            if flags & ops.EPI_DRELU:
                want = want * (R.float() > 0)
            torch.testing.assert_close(outs[0], want, atol=0.15, rtol=2e-2)
    finally:
        ops.gemm_set_option("skinny", -1)


@pytest.mark.parametrize("tA,tB", [(1, 1), (0, 0), (0, 1)])
def test_gemm_grouped_matches_single_launches(ops, tA, tB):
    """vpu_gemm_grouped: several independent problems (different M, N, K, ragged edges, long and short K) in ONE
    persistent launch, each un-split over its whole K == the fp32 matmul bit for bit on exact-integer operands, incl.
    accumulation into a pre-filled fp32 output and the fused bias-gradient column sums of the weight-gradient form."""
    g = torch.Generator().manual_seed(21)
    shapes = [(768, 768, 1152), (200, 136, 72), (384, 256, 4608), (130, 520, 640)]
    problems, checks = [], []
    for i, (M, N, K) in enumerate(shapes):
        A = torch.randint(-2, 3, (M, K), generator=g).float()
        Bm = torch.randint(-2, 3, (N, K), generator=g).float()
        lda, ldb = ((M + 7) // 8 * 8 if tA else K), ((N + 7) // 8 * 8 if tB else K)
        Ah = torch.zeros((K, lda) if tA else (M, lda)); Bh = torch.zeros((K, ldb) if tB else (N, ldb))
        if tA: Ah[:, :M] = A.t()
        else: Ah[:, :K] = A
        if tB: Bh[:, :N] = Bm.t()
        else: Bh[:, :K] = Bm
        Ad, Bd = dev(Ah).to(torch.bfloat16), dev(Bh).to(torch.bfloat16)
        ldc = (N + 7) // 8 * 8
        Cd = torch.full((M, ldc), float(i + 1), device="cuda")
        cs = torch.full((M,), 5.0, device="cuda") if (tA and tB and i != 1) else None
        problems.append(((Ad, Bd, Cd, M, N, K, lda, ldb, ldc, 0),
                         dict(transA=bool(tA), transB=bool(tB), flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=cs)))
        checks.append((Cd, cs, A @ Bm.t() + float(i + 1), A.sum(1) + 5.0, N))
    ops.gemm_grouped(problems)
    torch.cuda.synchronize()
    for Cd, cs, ref, csref, N in checks:
        assert torch.equal(Cd.cpu()[:, :N], ref), (Cd.cpu()[:, :N] - ref).abs().max()
        if cs is not None:
            assert torch.equal(cs.cpu(), csref)


@pytest.mark.parametrize("K", [512, 8192])
def test_gemm_fused_bias_gradient(ops, K):
    """wgrad GEMM with the bias-gradient column sums fused in (split-K for the long K, direct for the short one)."""
    g = torch.Generator().manual_seed(4)
    M, N = 200, 136                      # M = out features (rows of dW), N = in features
    dY = torch.randint(-2, 3, (K, M), generator=g).float()   # [tokens, out]  -> A, K-major
    X = torch.randint(-2, 3, (K, N), generator=g).float()    # [tokens, in]   -> B, K-major
    dYd, Xd = torch.zeros(K, 208), torch.zeros(K, 136)
    dYd[:, :M], Xd[:, :N] = dY, X
    dW = torch.full((M, N), 2.0, device="cuda")
    db = torch.full((M,), 3.0, device="cuda")
    ops.gemm(dev(dYd).to(torch.bfloat16), dev(Xd).to(torch.bfloat16), dW, M, N, K, 208, 136, N, 0, transA=True,
             transB=True, flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=db)
    assert torch.equal(dW.cpu(), dY.t() @ X + 2.0)
    assert torch.equal(db.cpu(), dY.sum(0) + 3.0)


@pytest.fixture(params=[0, 1], ids=["step32", "lean"])
def attn_form(request, ops):
    """both kernel families behind the attention entry points (vpu_attn_set_option)"""
    ops.attn_set_option("lean", request.param)
    yield request.param
    ops.attn_set_option("lean", -1)


@pytest.mark.parametrize("hd,n,nb", [(64, 196, 3), (64, 784, 1), (32, 196, 2), (64, 50, 2), (64, 256, 2), (80, 256, 1)])
def test_flash_attention_fwd_bwd(ops, attn_form, hd, n, nb):
    """Fused attention vs torch fp32 on the bf16-rounded inputs: forward within bf16 output rounding, gradients
    within 2e-2 of their scale.  n = 196 (window), 784 (global: four chunks of the whole-chunk form), 50 (ragged: masks
    keys and queries), 256 (ViT-H windows: two chunks, head dim 80 in the 128-column image)."""
    H = 3
    D = H * hd
    qkv = dev(rnd(nb * n, 3 * D, seed=60, scale=1.5)).to(torch.bfloat16)
    O = torch.zeros(nb * n, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(nb * H, n, device="cuda")
    scale = hd ** -0.5
    ops.attn_fwd(qkv, (qkv, D), (qkv, 2 * D), O, lse, nb, H, n, hd, 3 * D, D, scale)
    x = qkv.float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)
    S = (x[0] @ x[1].transpose(-1, -2)) * scale
    ref = (torch.softmax(S, -1) @ x[2])
    torch.testing.assert_close(O.float().view(nb, n, H, hd).transpose(1, 2), ref, atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(lse.view(nb, H, n), torch.logsumexp(S, -1), atol=2e-3, rtol=1e-4)
    dO = dev(rnd(nb * n, D, seed=61)).to(torch.bfloat16)
    ref.backward(dO.float().view(nb, n, H, hd).transpose(1, 2))
    dqkv = torch.zeros_like(qkv)
    delta = torch.zeros(nb * H, n, device="cuda")
    ops.attn_bwd(qkv, (qkv, D), (qkv, 2 * D), O, dO, lse, delta, dqkv, (dqkv, D), (dqkv, 2 * D), nb, H, n, hd, 3 * D, D,
                 3 * D, scale)
    got = dqkv.float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4)
    for i, name in enumerate("qkv"):
        ref_g = x.grad[i]
        tol = 2e-2 * ref_g.abs().max().item()
        assert (got[i] - ref_g).abs().max().item() < tol, (name, (got[i] - ref_g).abs().max().item(), tol)


@pytest.mark.parametrize("nq,nk,hd,S", [(48, 784, 48, 4), (784, 48, 48, 4), (48, 1024, 80, 4), (1024, 48, 80, 2), (64, 256, 64, 2)])
def test_split_attention_equals_unsplit(ops, nq, nk, hd, S):
    """vpu_xattn_fwd_split / _bwd_split + vpu_attn_combine / vpu_sum_groups (round 5: the long side of the DMA neck's prompt <->
    image attentions cut into S ranges that run as batch entries of one launch) against the unsplit launch and torch fp32:
    long keys (qdiv = S: partial softmaxes merged through their log-sum-exp, dQ as partial sums) and long queries (kdiv = S: dK /
    dV as partial sums); ViT-B's 48 x 784 at head dim 48 and ViT-H's 48 x 1024 at head dim 80."""
    nb, H = 3, 8
    ld = H * hd
    scale = hd ** -0.5
    Q = dev(rnd(nb * nq, ld, seed=300 + nq)).to(torch.bfloat16)
    K = dev(rnd(nb * nk, ld, seed=301 + nk)).to(torch.bfloat16)
    V = dev(rnd(nb * nk, ld, seed=302 + nk)).to(torch.bfloat16)
    dO = dev(rnd(nb * nq, ld, seed=303 + nq)).to(torch.bfloat16)
    O0, lse0 = torch.empty_like(Q), torch.empty(nb * H, nq, device="cuda")
    ops.xattn_fwd(Q, K, V, O0, lse0, nb, H, nq, nk, hd, ld, ld, ld, scale)
    dQ0, dK0, dV0 = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    ops.xattn_bwd(Q, K, V, O0, dO, lse0, torch.empty(nb * H, nq, device="cuda"), dQ0, dK0, dV0, nb, H, nq, nk, hd, ld, ld, ld, ld, ld, scale)
    O1, dQ1, dK1, dV1 = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    if nk > nq:     # long keys: the queries are shared
        o_s, lse_s = torch.empty(nb * S * nq, ld, device="cuda", dtype=torch.bfloat16), torch.empty(nb * S * H, nq, device="cuda")
        ops.xattn_fwd_split(Q, K, V, o_s, lse_s, nb * S, H, nq, nk // S, hd, ld, ld, ld, scale, S, 1)
        lse1 = torch.empty(nb * H, nq, device="cuda")
        ops.attn_combine(o_s, lse_s, O1, lse1, nb, H, nq, hd, S, ld, ld)
        dqp = torch.empty(nb * S * nq, ld, device="cuda", dtype=torch.bfloat16)
        ops.xattn_bwd_split(Q, K, V, O1, dO, lse1, torch.empty(nb * H, nq, device="cuda"), dqp, dK1, dV1, nb * S, H, nq, nk // S, hd,
                            ld, ld, ld, ld, ld, scale, S, 1)
        ops.sum_groups(dqp, dQ1, nb, S, nq * ld)
        assert torch.allclose(lse1, lse0, atol=2e-3, rtol=1e-4)
    else:           # long queries: the keys are shared
        lse1 = torch.empty(nb * S * H, nq // S, device="cuda")
        ops.xattn_fwd_split(Q, K, V, O1, lse1, nb * S, H, nq // S, nk, hd, ld, ld, ld, scale, 1, S)
        dkp, dvp = torch.empty(nb * S * nk, ld, device="cuda", dtype=torch.bfloat16), torch.empty(nb * S * nk, ld, device="cuda", dtype=torch.bfloat16)
        ops.xattn_bwd_split(Q, K, V, O1, dO, lse1, torch.empty(nb * S * H, nq // S, device="cuda"), dQ1, dkp, dvp, nb * S, H, nq // S, nk,
                            hd, ld, ld, ld, ld, ld, scale, 1, S)
        ops.sum_groups(dkp, dK1, nb, S, nk * ld)
        ops.sum_groups(dvp, dV1, nb, S, nk * ld)
        assert torch.equal(O1, O0) and torch.equal(dQ1, dQ0)       # (each query range is its own softmax: the same arithmetic)
    # torch fp32 on the bf16-rounded operands
    q_, k_, v_ = (t.float().view(nb, -1, H, hd).transpose(1, 2).clone().requires_grad_(True) for t in (Q, K, V))
    ref = torch.softmax((q_ @ k_.transpose(-1, -2)) * scale, -1) @ v_
    ref.backward(dO.float().view(nb, nq, H, hd).transpose(1, 2))
    back = lambda t: t.transpose(1, 2).reshape(-1, ld)
    for name, got, base, want in (("O", O1, O0, back(ref.detach())), ("dQ", dQ1, dQ0, back(q_.grad)), ("dK", dK1, dK0, back(k_.grad)),
                                  ("dV", dV1, dV0, back(v_.grad))):
        sc = want.abs().max().item()
        e1, e0 = (got.float() - want).abs().max().item(), (base.float() - want).abs().max().item()
        print(f"[split attention] {nq}x{nk} hd {hd} {name}: split {e1 / sc:.2e}, unsplit {e0 / sc:.2e} of the scale")
        assert e1 < 2e-2 * sc, (name, e1, sc)
        assert (got.float() - base.float()).abs().max().item() < 1.5e-2 * sc, name


@pytest.mark.parametrize("n,nb,H", [(196, 5, 3), (256, 2, 2), (208, 1, 1), (100, 3, 2), (64, 2, 4), (16, 3, 1), (4, 2, 1), (132, 2, 12), (50, 2, 2), (197, 1, 2)])
def test_one_pass_window_backward(ops, n, nb, H):
    """The one-pass (window, head) backward kernels (head dim 64, n <= 256) against torch fp32 on the bf16-rounded inputs AND
    against the two-kernel backward they replace -- the round-5 form in key passes (attn_bwd_winp_kernel<1>: four 16-key tiles
    per pass, one per wave, four workgroups per CU) and the round-3 form (attn_bwd_win_kernel: one workgroup per CU): key-tile
    counts 13 (four passes of 4, 4, 4, 1 tiles), 16, 13 exact, 7 (ragged last tile), 4, 1 and a quarter tile; every gradient
    within 2e-2 of its scale and within bf16 rounding of the two-kernel result (dQ of the pass form: one more bf16 rounding
    of the partial sum per key pass, hence 1.5e-2); the launch really was the kernel asked for; bitwise reproducible."""
    hd = 64
    D = H * hd
    qkv = dev(rnd(nb * n, 3 * D, seed=160 + n, scale=1.5)).to(torch.bfloat16)
    O = torch.zeros(nb * n, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(nb * H, n, device="cuda")
    scale = hd ** -0.5
    ops.attn_fwd(qkv, (qkv, D), (qkv, 2 * D), O, lse, nb, H, n, hd, 3 * D, D, scale)
    x = qkv.float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)
    ref = torch.softmax((x[0] @ x[1].transpose(-1, -2)) * scale, -1) @ x[2]
    dO = dev(rnd(nb * n, D, seed=161 + n)).to(torch.bfloat16)
    ref.backward(dO.float().view(nb, n, H, hd).transpose(1, 2))
    names = {3: "attn_bwd_winx_kernel" if 65 <= n <= 224 else "attn_bwd_win_kernel", 2: "attn_bwd_winp_kernel<1>", 1: "attn_bwd_win_kernel"}
    res = {}
    try:
        for onepass in (3, 2, 1, 0):
            ops.attn_set_option("onepass", onepass)
            dqkv = torch.full_like(qkv, float("nan"))
            delta = torch.zeros(nb * H, n, device="cuda")
            ops.attn_bwd(qkv, (qkv, D), (qkv, 2 * D), O, dO, lse, delta, dqkv, (dqkv, D), (dqkv, 2 * D), nb, H, n, hd, 3 * D, D,
                         3 * D, scale)
            assert ops.attn_last_kernel() == names.get(onepass, ops.attn_last_kernel()) and \
                (onepass > 0 or "attn_bwd_dq" in ops.attn_last_kernel()), ops.attn_last_kernel()
            res[onepass] = dqkv.float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4)
            # no atomics, no cross-wave reduction whose order could vary: the same launch gives the same bits every time
            for _ in range(3 if onepass else 0):
                again = torch.full_like(qkv, float("nan"))
                ops.attn_bwd(qkv, (qkv, D), (qkv, 2 * D), O, dO, lse, torch.zeros(nb * H, n, device="cuda"), again, (again, D),
                             (again, 2 * D), nb, H, n, hd, 3 * D, D, 3 * D, scale)
                assert torch.equal(again.float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4), res[onepass])
    finally:
        ops.attn_set_option("onepass", -1)
    assert torch.equal(res[3], res[1])       # the persistent form runs the per-problem form's arithmetic: same bits
    for mode in (2, 1):
        for i, name in enumerate("qkv"):
            ref_g = x.grad[i]
            sc = ref_g.abs().max().item()
            assert torch.isfinite(res[mode][i]).all(), (mode, name)
            err, dif = (res[mode][i] - ref_g).abs().max().item(), (res[mode][i] - res[0][i]).abs().max().item()
            print(f"[one-pass {mode}] n={n} d{name}: vs torch {err / sc:.2e}, vs two kernels {dif / sc:.2e} of the scale")
            assert err < 2e-2 * sc, (mode, name, err, sc)
            assert dif < (1.5e-2 if mode == 2 and name == "q" else 1e-2) * sc, (mode, name, dif, sc)


@pytest.mark.parametrize("n,nb,H", [(196, 48, 12), (100, 130, 4), (160, 90, 6), (70, 300, 2), (200, 45, 12), (176, 100, 3), (224, 33, 8)])
def test_persistent_window_backward_walks_several_problems_per_workgroup(ops, n, nb, H):
    """The persistent window backward (attn_bwd_winx_kernel: "onepass" = 3, the default) with MORE problems than the chip has CUs,
    so that every workgroup goes through the switch -- the next problem's Q / dO ring slots, K / V images, -lse and delta all come
    from the prefetch -- two or three times (576, 520, 540, 600, 540, 300, 264 problems; 7, 4, 5, 3, 7, 6, 7 blocks of 32 keys;
    odd and even tile counts, a ragged last tile): bit-identical to the per-problem kernel ("onepass" = 1) on the same inputs,
    twice in a row (no stale ring slot from the previous launch), and -- first case, the ViT-B bs-12 launch -- within 2e-2 of torch
    fp32 on the bf16-rounded inputs."""
    hd = 64
    D = H * hd
    qkv = dev(rnd(nb * n, 3 * D, seed=260 + n, scale=1.5)).to(torch.bfloat16)
    O = torch.zeros(nb * n, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(nb * H, n, device="cuda")
    scale = hd ** -0.5
    ops.attn_fwd(qkv, (qkv, D), (qkv, 2 * D), O, lse, nb, H, n, hd, 3 * D, D, scale)
    dO = dev(rnd(nb * n, D, seed=261 + n)).to(torch.bfloat16)
    res = {}
    try:
        for onepass in (1, 3, 3):
            ops.attn_set_option("onepass", onepass)
            dqkv = torch.full_like(qkv, float("nan"))
            delta = torch.zeros(nb * H, n, device="cuda")
            ops.attn_bwd(qkv, (qkv, D), (qkv, 2 * D), O, dO, lse, delta, dqkv, (dqkv, D), (dqkv, 2 * D), nb, H, n, hd, 3 * D, D,
                         3 * D, scale)
            assert ops.attn_last_kernel() == ("attn_bwd_winx_kernel" if onepass == 3 else "attn_bwd_win_kernel"), ops.attn_last_kernel()
            assert torch.isfinite(dqkv.float()).all()
            if onepass == 3:
                assert torch.equal(dqkv, res[1]), (n, nb, H, (dqkv.float() - res[1].float()).abs().max().item())
            res[onepass] = dqkv
        if n == 196:
            # a data-parallel step keeps CUs out of the persistent grids ("reserve_cus"): the walk over the problems changes, not the bits
            ops.gemm_set_option("reserve_cus", 16)
            try:
                dqkv = torch.full_like(qkv, float("nan"))
                ops.attn_bwd(qkv, (qkv, D), (qkv, 2 * D), O, dO, lse, torch.zeros(nb * H, n, device="cuda"), dqkv, (dqkv, D), (dqkv, 2 * D),
                             nb, H, n, hd, 3 * D, D, 3 * D, scale)
                assert ops.attn_last_kernel() == "attn_bwd_winx_kernel" and torch.equal(dqkv, res[1])
            finally:
                ops.gemm_set_option("reserve_cus", 0)
    finally:
        ops.attn_set_option("onepass", -1)
    if n == 196:
        x = qkv.float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)
        ref = torch.softmax((x[0] @ x[1].transpose(-1, -2)) * scale, -1) @ x[2]
        ref.backward(dO.float().view(nb, n, H, hd).transpose(1, 2))
        got = res[3].float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4)
        for i, name in enumerate("qkv"):
            sc = x.grad[i].abs().max().item()
            err = (got[i] - x.grad[i]).abs().max().item()
            print(f"[persistent window backward] d{name}: {err / sc:.2e} of the scale")
            assert err < 2e-2 * sc, (name, err, sc)


def test_attention_large_scores_raise_the_reference(ops, attn_form):
    """Scores with a spread of ~ +-60 (inputs scaled by 4): the running maximum of the forward is raised several times
    along the 784 keys (the lean kernels only do so when a step exceeds the reference by more than 2^8) and rows are close
    to one-hot; output, log-sum-exp and gradients still match torch fp32."""
    hd, n, nb, H = 64, 784, 1, 2
    D = H * hd
    qkv = dev(rnd(nb * n, 3 * D, seed=66, scale=4.0)).to(torch.bfloat16)
    O = torch.zeros(nb * n, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(nb * H, n, device="cuda")
    scale = hd ** -0.5
    ops.attn_fwd(qkv, (qkv, D), (qkv, 2 * D), O, lse, nb, H, n, hd, 3 * D, D, scale)
    x = qkv.float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)
    S = (x[0] @ x[1].transpose(-1, -2)) * scale
    ref = torch.softmax(S, -1) @ x[2]
    assert (S.max(-1).values - S[..., :32].max(-1).values).max().item() > 8 * 0.6931   # the reference does move
    torch.testing.assert_close(O.float().view(nb, n, H, hd).transpose(1, 2), ref, atol=6e-2, rtol=2e-2)
    torch.testing.assert_close(lse.view(nb, H, n), torch.logsumexp(S, -1), atol=2e-2, rtol=1e-4)
    dO = dev(rnd(nb * n, D, seed=67)).to(torch.bfloat16)
    ref.backward(dO.float().view(nb, n, H, hd).transpose(1, 2))
    dqkv = torch.zeros_like(qkv)
    delta = torch.zeros(nb * H, n, device="cuda")
    ops.attn_bwd(qkv, (qkv, D), (qkv, 2 * D), O, dO, lse, delta, dqkv, (dqkv, D), (dqkv, 2 * D), nb, H, n, hd, 3 * D, D,
                 3 * D, scale)
    got = dqkv.float().view(nb, n, 3, H, hd).permute(2, 0, 3, 1, 4)
    for i, name in enumerate("qkv"):
        ref_g = x.grad[i]
        tol = 3e-2 * ref_g.abs().max().item()
        assert (got[i] - ref_g).abs().max().item() < tol, (name, (got[i] - ref_g).abs().max().item(), tol)


@pytest.mark.parametrize("hd,nq,nk,nb", [(96, 48, 48, 2), (48, 48, 784, 2), (48, 784, 48, 2), (48, 50, 70, 1), (16, 20, 33, 3)])
def test_cross_attention_fwd_bwd(ops, attn_form, hd, nq, nk, nb):
    """The DMA neck's attention (transformer.py:499-521): separate query and key/value matrices, head dims 48 / 96 (run in
    the 64 / 128-column instantiation with zero-staged padding), ragged nq / nk.  Same tolerances as the self-attention."""
    H = 8
    D = H * hd
    Q = dev(rnd(nb * nq, D, seed=62, scale=1.5)).to(torch.bfloat16)
    K = dev(rnd(nb * nk, D, seed=63, scale=1.5)).to(torch.bfloat16)
    V = dev(rnd(nb * nk, D, seed=64, scale=1.5)).to(torch.bfloat16)
    O = torch.zeros(nb * nq, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(nb * H, nq, device="cuda")
    scale = hd ** -0.5
    ops.xattn_fwd(Q, K, V, O, lse, nb, H, nq, nk, hd, D, D, D, scale)
    q = Q.float().view(nb, nq, H, hd).transpose(1, 2).clone().requires_grad_(True)
    k = K.float().view(nb, nk, H, hd).transpose(1, 2).clone().requires_grad_(True)
    v = V.float().view(nb, nk, H, hd).transpose(1, 2).clone().requires_grad_(True)
    S = (q @ k.transpose(-1, -2)) * scale
    ref = torch.softmax(S, -1) @ v
    torch.testing.assert_close(O.float().view(nb, nq, H, hd).transpose(1, 2), ref, atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(lse.view(nb, H, nq), torch.logsumexp(S, -1), atol=2e-3, rtol=1e-4)
    dO = dev(rnd(nb * nq, D, seed=65)).to(torch.bfloat16)
    ref.backward(dO.float().view(nb, nq, H, hd).transpose(1, 2))
    dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
    delta = torch.zeros(nb * H, nq, device="cuda")
    ops.xattn_bwd(Q, K, V, O, dO, lse, delta, dQ, dK, dV, nb, H, nq, nk, hd, D, D, D, D, D, scale)
    for got, ref_g, n_, name in ((dQ, q.grad, nq, "q"), (dK, k.grad, nk, "k"), (dV, v.grad, nk, "v")):
        got = got.float().view(nb, n_, H, hd).transpose(1, 2)
        tol = 2e-2 * ref_g.abs().max().item()
        assert (got - ref_g).abs().max().item() < tol, (name, (got - ref_g).abs().max().item(), tol)


@pytest.mark.parametrize("hd,nq,nk", [(80, 48, 1024), (96, 48, 1024), (128, 48, 1024), (48, 48, 784), (64, 48, 784),
                                      (80, 1024, 1024), (64, 784, 784), (64, 196, 196), (80, 256, 256), (48, 784, 48),
                                      (80, 1024, 48)])
@pytest.mark.parametrize("in_scale", [3.0, 8.0])
def test_attention_with_large_scores_is_reproducible_and_finite(ops, hd, nq, nk, in_scale):
    """Scores far apart (|q.k| of tens: the running-maximum rescaling of the streaming softmax is exercised at every key
    chunk) over many key chunks, at the neck's and the backbones' problem shapes: the output matches torch, is finite, and
    six launches on the same inputs -- with different memory around them -- agree bit for bit, forward and backward.  (The
    forward kernels' running maximum once read its score MFMAs' results without the wait states an XDL result needs -- inline
    asm, invisible to the compiler's hazard recognizer --: the one-tile 128-column form failed all three at 48 queries, ViT-H's
    prompt tokens attending to the image: csrc/attention.hip, max8.)"""
    nb, H = 12, 8
    ld = H * hd
    g = torch.Generator(device="cuda").manual_seed(1234)
    Q = (torch.randn(nb * nq, ld, device="cuda", generator=g) * in_scale).to(torch.bfloat16)
    K = (torch.randn(nb * nk, ld, device="cuda", generator=g) * in_scale).to(torch.bfloat16)
    V = torch.randn(nb * nk, ld, device="cuda", generator=g).to(torch.bfloat16)
    dO = torch.randn(nb * nq, ld, device="cuda", generator=g).to(torch.bfloat16)
    sc = hd ** -0.5
    outs = []
    for r in range(6):
        junk = torch.full((1 << 20,), float(r + 1), device="cuda")       # (different neighbours in memory each time)
        O = torch.full((nb * nq, ld), float("nan"), device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(nb * H, nq, device="cuda")
        ops.xattn_fwd(Q, K, V, O, lse, nb, H, nq, nk, hd, ld, ld, ld, sc)
        dq, dk, dv = (torch.full_like(t, float("nan")) for t in (Q, K, V))
        delta = torch.empty(nb * H, nq, device="cuda")
        ops.xattn_bwd(Q, K, V, O, dO, lse, delta, dq, dk, dv, nb, H, nq, nk, hd, ld, ld, ld, ld, ld, sc)
        torch.cuda.synchronize()
        outs.append((O, lse, dq, dk, dv))
        del junk
    for t in outs[0]:
        assert torch.isfinite(t.float()).all()
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o)), "attention is not reproducible from launch to launch"
    q = Q.float().view(nb, nq, H, hd).permute(0, 2, 1, 3)
    k = K.float().view(nb, nk, H, hd).permute(0, 2, 1, 3)
    v = V.float().view(nb, nk, H, hd).permute(0, 2, 1, 3)
    S = q @ k.transpose(-1, -2) * sc
    ref = (torch.softmax(S, -1) @ v).permute(0, 2, 1, 3).reshape(nb * nq, ld)
    assert (outs[0][0].float() - ref).abs().max().item() < 3e-2          # (values of order 1, rounded to bf16)
    torch.testing.assert_close(outs[0][1].view(nb, H, nq), torch.logsumexp(S, -1), atol=2e-2, rtol=1e-3)


@pytest.mark.parametrize("B,S,h,H", [(2, 8, 28, 112), (1, 4, 128, 448), (1, 2, 112, 448), (2, 2, 9, 32)])
def test_fused_upsample_p2cl_matches_unfused(ops, B, S, h, H):
    """p2cl_up (upsample + loss + both backward passes, one kernel) == upsample_ac_fwd -> p2cl -> upsample_ac_bwd,
    with per-slot override masks, and it is bitwise reproducible.  Shapes: the test size, ViT-H's 128-wide head map
    (7-row bands), ViT-B's 112 -> 448, and a non-integer ratio with a short last band."""
    low = dev(torch.sigmoid(rnd(B, S, h, h, seed=70, scale=3.0)))
    gt = dev((rnd(B, 1, H, H, seed=71) > 0.2).float())
    ov = dev((rnd(2, H, H, seed=72) > 0.5).float())
    idx = -torch.ones(B, S, dtype=torch.int32)
    idx[0, 1] = 0; idx[B - 1, S - 2] = 1
    idx = dev(idx)
    gs = 2.0 / (B * S * H * H)
    up = torch.empty(B, S, H, H, device="cuda")
    ops.upsample_ac_fwd(low, up, B * S, h, h, H, H)
    part, dprob, dlow = torch.empty(B, S, device="cuda"), torch.empty_like(up), torch.empty_like(low)
    ops.p2cl_fwd_bwd(up, gt, idx, ov, part, dprob, gs, B, S, H, H)
    ops.upsample_ac_bwd(dprob, dlow, B * S, h, h, H, H)
    part2, dlow2 = torch.empty(B, S, device="cuda"), torch.empty_like(low)
    ops.p2cl_up_fwd_bwd(low, gt, idx, ov, part2, dlow2, gs, B, S, h, h, H, H)
    torch.testing.assert_close(part2, part, rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(dlow2, dlow, rtol=1e-4, atol=1e-9)
    part3, dlow3 = torch.empty(B, S, device="cuda"), torch.empty_like(low)
    ops.p2cl_up_fwd_bwd(low, gt, idx, ov, part3, dlow3, gs, B, S, h, h, H, H)
    assert torch.equal(dlow3, dlow2) and torch.equal(part3, part2)


@pytest.mark.parametrize("tB", [0, 1])
@pytest.mark.parametrize("k2", [1, 3])
@pytest.mark.parametrize("shape", [(2300, 1288, 256), (3000, 1024, 640), (1568, 3072, 1024), (9408, 768, 768)])
def test_gemm_k2_exact_and_epilogues(ops, tB, k2, shape):
    """The round-2 kernels (256-row tiles, 128 x 64 outputs per wave, K halves exchanged through LDS in the 256 x 128
    form; vpu_gemm_set_option("k2", 1 / 3) = 256 x 128 / 256 x 256 tiles): exact-integer operands must give the fp32
    matmul bit for bit (ragged M and N, several tiles per persistent workgroup, K of 4 and 10 K-tiles so that the ring's
    out-of-range tail stages occur); then the flag sets of the ViT blocks against the 128 x 128 kernel's output."""
    M, N, K = shape
    g = torch.Generator().manual_seed(11)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    Bm = torch.randint(-3, 4, (N, K), generator=g).float()
    A[0, 1] = 3; A[1, 0] = -2; Bm[0, 1] = 1; Bm[1, 0] = -3
    ref = A @ Bm.t()
    Ad = dev(A).to(torch.bfloat16)
    Bd = dev(Bm.t().contiguous() if tB else Bm).to(torch.bfloat16)
    ldb = N if tB else K
    bias = dev(torch.randint(-4, 5, (N,), generator=g).float())
    R = dev(torch.randint(-4, 5, (M, N), generator=g).float()).to(torch.bfloat16)
    aux = dev(torch.randint(-2, 3, (M, N), generator=g).float()).to(torch.bfloat16)

    def run(opt, flags, **kw):
        Cd = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        pre = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm_set_option("k2", opt)
        try:
            ops.gemm(Ad, Bd, Cd, M, N, K, K, ldb, N, 0, transB=bool(tB), flags=flags, preact=pre, **kw)
            torch.cuda.synchronize()
        finally:
            ops.gemm_set_option("k2", -1)
        return Cd.float().cpu(), pre.float().cpu()

    if tB == 0:
        out, _ = run(k2, 0)                      # (round 4: the bias-free form of the head's fusion / FPN convolutions)
        assert torch.equal(out, ref.to(torch.bfloat16).float())
        out, _ = run(k2, ops.EPI_BIAS, bias=bias)
        assert torch.equal(out, (ref + bias.cpu()).to(torch.bfloat16).float())
        out, _ = run(k2, ops.EPI_BIAS | ops.EPI_RESID, bias=bias, resid=R, ldr=N)
        assert torch.equal(out, (ref + bias.cpu() + R.float().cpu()).to(torch.bfloat16).float())
        fl = ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_SAVE_DGELU
        # GELU: small-magnitude operands so that the activation is exercised away from saturation
        Ad_s, Bd_s = Ad, Bd
        Ad = (Ad.float() * 0.125).to(torch.bfloat16)
        o_new, p_new = run(k2, fl, bias=bias * 0.25)
        o_old, p_old = run(0, fl, bias=bias * 0.25)
        Ad = Ad_s
        assert torch.equal(o_new, o_old) and torch.equal(p_new, p_old)
    else:
        out, _ = run(k2, 0)
        assert torch.equal(out, ref.to(torch.bfloat16).float())
        out, _ = run(k2, ops.EPI_MULAUX, aux=aux, ldaux=N)
        assert torch.equal(out, (ref * aux.float().cpu()).to(torch.bfloat16).float())


@pytest.mark.parametrize("tB", [0, 1])
@pytest.mark.parametrize("shape,grid", [((6001, 1544, 640), 0), ((6001, 1544, 640), 24), ((9408, 2304, 768), 0), ((5000, 1288, 1024), 7),
                                        ((2300, 1288, 576), 3), ((9408, 768, 3072), 40), ((6144, 2688, 640), 0), ((6100, 2680, 1280), 50)])
def test_gemm_k5_exact_and_epilogues(ops, tB, shape, grid):
    """Round 6, K5 (gemm_k5.hip): the two 4-wave groups of a workgroup own alternate 128-column tiles; one group's LDS-DMA +
    direct epilogue run beside the other group's main loop, the K-steps of consecutive tiles form one stream through a three-stage
    ring.  Exact-integer operands must give the fp32 matmul bit for bit in every flag set of the ViT blocks -- ragged M and N,
    K of 9 .. 48 K-steps, one to dozens of tiles per workgroup (`k5_grid` caps the grid so that every workgroup walks many tiles:
    odd and even counts, both groups ending a launch), tile heights 256 / 224 / 192 by the rounds rule (256 rows: the form with one
    fragment set, ViT-H's 12288 rows; (6144 | 6100) x (2688 | 2680) take it) -- and GELU + GELU' equal the 128 x 128 kernel's output."""
    M, N, K = shape
    g = torch.Generator().manual_seed(61)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    Bm = torch.randint(-3, 4, (N, K), generator=g).float()
    A[0, 1] = 3; A[1, 0] = -2; Bm[0, 1] = 1; Bm[1, 0] = -3
    ref = A @ Bm.t()
    Ad = dev(A).to(torch.bfloat16)
    Bd = dev(Bm.t().contiguous() if tB else Bm).to(torch.bfloat16)
    ldb = N if tB else K
    bias = dev(torch.randint(-4, 5, (N,), generator=g).float())
    R = dev(torch.randint(-4, 5, (M, N), generator=g).float()).to(torch.bfloat16)
    aux = dev(torch.randint(-2, 3, (M, N), generator=g).float()).to(torch.bfloat16)

    def run(k5, flags, A_=None, split=-1, **kw):
        Cd = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        pre = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm_set_option("k5", 2 if k5 else 0)
        ops.gemm_set_option("k5_grid", grid if k5 else 0)
        ops.gemm_set_option("k5_split", split)
        if not k5:
            ops.gemm_set_option("k2", 0)
        try:
            ops.gemm(Ad if A_ is None else A_, Bd, Cd, M, N, K, K, ldb, N, 0, transB=bool(tB), flags=flags, preact=pre, **kw)
            name = ops.gemm_last_kernel()
            torch.cuda.synchronize()
        finally:
            ops.gemm_set_option("k5", -1)
            ops.gemm_set_option("k5_grid", 0)
            ops.gemm_set_option("k5_split", -1)
            ops.gemm_set_option("k2", -1)
        # (the residual / aux forms take ten epilogue intervals: K >= 640; below that the call falls through to K2)
        want = bool(k5) and not (K < 640 and (flags & (ops.EPI_RESID | ops.EPI_MULAUX)))
        assert name.startswith("gemm_bf16_k5_kernel<%d, " % tB) == want, name
        return Cd.float().cpu(), pre.float().cpu()

    # k5_split: the LDS-DMA pieces issued by the producer waves alone (0) or half by each wave group (1; the default for GELU)
    for split in (0, 1):
        if tB == 0:
            out, _ = run(1, 0, split=split)
            assert torch.equal(out, ref.to(torch.bfloat16).float())
            out, _ = run(1, ops.EPI_BIAS, split=split, bias=bias)
            assert torch.equal(out, (ref + bias.cpu()).to(torch.bfloat16).float())
            out, _ = run(1, ops.EPI_BIAS | ops.EPI_RESID, split=split, bias=bias, resid=R, ldr=N)
            assert torch.equal(out, (ref + bias.cpu() + R.float().cpu()).to(torch.bfloat16).float())
            fl = ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_SAVE_DGELU
            As = (Ad.float() * 0.125).to(torch.bfloat16)   # small magnitudes: the activation away from saturation
            o_new, p_new = run(1, fl, A_=As, split=split, bias=bias * 0.25)
            o_old, p_old = run(0, fl, A_=As, bias=bias * 0.25)
            assert torch.equal(o_new, o_old) and torch.equal(p_new, p_old)
        else:
            out, _ = run(1, 0, split=split)
            assert torch.equal(out, ref.to(torch.bfloat16).float())
            out, _ = run(1, ops.EPI_MULAUX, split=split, aux=aux, ldaux=N)
            assert torch.equal(out, (ref * aux.float().cpu()).to(torch.bfloat16).float())


@pytest.mark.parametrize("shape,rb", [((6272, 768, 768), 6), ((6235, 768, 1024), 6), ((9408, 1280, 1280), 6), ((3136, 768, 1024), 5),
                                      ((9408, 768, 768), 7), ((2300, 1288, 256), 6), ((5120, 1536, 256), 8)])
def test_gemm_k2_tile_height_by_rounds(ops, shape, rb):
    """Round 5: the 256 x 128 K2 form picks 256 / 224 / 192-row tiles (8 / 7 / 6 row blocks per wave) by rounds of tiles x rows per
    tile, and 160 rows where nothing taller reaches the K2 forms' tile count -- batch 8's 6272 rows and ViT-H's N = 1280 take 192,
    batch 4's 3136 rows 160, ViT-B's 9408 x 768 keeps 224.  The
    instantiation the call took is checked by name; exact-integer operands give the fp32 product bit for bit in every form the
    ViT blocks use (plain, + bias, + bias + residual, dgrad, dgrad x aux), ragged last tiles included."""
    M, N, K = shape
    g = torch.Generator().manual_seed(41)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    Bm = torch.randint(-3, 4, (N, K), generator=g).float()
    ref = A @ Bm.t()
    Ad, Bd, Bt = dev(A).to(torch.bfloat16), dev(Bm).to(torch.bfloat16), dev(Bm.t().contiguous()).to(torch.bfloat16)
    bias = dev(torch.randint(-4, 5, (N,), generator=g).float())
    R = dev(torch.randint(-4, 5, (M, N), generator=g).float()).to(torch.bfloat16)
    aux = dev(torch.randint(-2, 3, (M, N), generator=g).float()).to(torch.bfloat16)
    ops.gemm_set_option("k2", 1)
    try:
        for tB, flags, kw, want in ((0, 0, {}, ref), (0, ops.EPI_BIAS, dict(bias=bias), ref + bias.cpu()),
                                    (0, ops.EPI_BIAS | ops.EPI_RESID, dict(bias=bias, resid=R, ldr=N), ref + bias.cpu() + R.float().cpu()),
                                    (1, 0, {}, ref), (1, ops.EPI_MULAUX, dict(aux=aux, ldaux=N), ref * aux.float().cpu())):
            Cd = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
            ops.gemm(Ad, Bt if tB else Bd, Cd, M, N, K, K, N if tB else K, N, 0, transB=bool(tB), flags=flags, **kw)
            name = ops.gemm_last_kernel()
            assert name.startswith("gemm_bf16_k2_kernel<0, %d, 2, " % tB) and name.endswith(", %d, 1>" % rb), name
            torch.cuda.synchronize()
            assert torch.equal(Cd.float().cpu(), want.to(torch.bfloat16).float()), (tB, flags)
    finally:
        ops.gemm_set_option("k2", -1)


@pytest.mark.lab
@pytest.mark.parametrize("tB", [0, 1])
@pytest.mark.parametrize("shape", [(6001, 1544, 256), (5000, 2304, 640), (9408, 2304, 768)])
def test_gemm_k3_exact_and_epilogues(ops, tB, shape):
    """The round-4 kernels (vpu_gemm_set_option("k3", 3): 256 / 224 x 128 tiles in 256-thread workgroups, two per CU, 32-deep
    K-steps, K-contiguous operands as [128 rows][32 k] images): exact-integer operands must give the fp32 matmul bit for bit
    (ragged M and N, more tiles than workgroup slots, K of 8, 20 and 24 K-steps); the flag sets of the ViT blocks against
    the 128 x 128 kernel's output."""
    M, N, K = shape
    g = torch.Generator().manual_seed(13)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    Bm = torch.randint(-3, 4, (N, K), generator=g).float()
    A[0, 1] = 3; A[1, 0] = -2; Bm[0, 1] = 1; Bm[1, 0] = -3
    ref = A @ Bm.t()
    Ad = dev(A).to(torch.bfloat16)
    Bd = dev(Bm.t().contiguous() if tB else Bm).to(torch.bfloat16)
    ldb = N if tB else K
    bias = dev(torch.randint(-4, 5, (N,), generator=g).float())
    R = dev(torch.randint(-4, 5, (M, N), generator=g).float()).to(torch.bfloat16)
    aux = dev(torch.randint(-2, 3, (M, N), generator=g).float()).to(torch.bfloat16)

    def run(k3, flags, A_=None, **kw):
        Cd = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        pre = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm_set_option("k3", k3)
        if not k3:
            ops.gemm_set_option("k2", 0)
        try:
            ops.gemm(Ad if A_ is None else A_, Bd, Cd, M, N, K, K, ldb, N, 0, transB=bool(tB), flags=flags, preact=pre, **kw)
            name = ops.gemm_last_kernel()
            torch.cuda.synchronize()
        finally:
            ops.gemm_set_option("k3", -1)
            ops.gemm_set_option("k2", -1)
        assert ("k3_kernel" in name) == bool(k3), name
        return Cd.float().cpu(), pre.float().cpu()

    if tB == 0:
        out, _ = run(3, ops.EPI_BIAS, bias=bias)
        assert torch.equal(out, (ref + bias.cpu()).to(torch.bfloat16).float())
        out, _ = run(3, ops.EPI_BIAS | ops.EPI_RESID, bias=bias, resid=R, ldr=N)
        assert torch.equal(out, (ref + bias.cpu() + R.float().cpu()).to(torch.bfloat16).float())
        fl = ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_SAVE_DGELU
        As = (Ad.float() * 0.125).to(torch.bfloat16)      # GELU away from saturation
        o_new, p_new = run(3, fl, A_=As, bias=bias * 0.25)
        o_old, p_old = run(0, fl, A_=As, bias=bias * 0.25)
        assert torch.equal(o_new, o_old) and torch.equal(p_new, p_old)
    else:
        out, _ = run(3, 0)
        assert torch.equal(out, ref.to(torch.bfloat16).float())
        out, _ = run(3, ops.EPI_MULAUX, aux=aux, ldaux=N)
        assert torch.equal(out, (ref * aux.float().cpu()).to(torch.bfloat16).float())


@pytest.mark.parametrize("tB", [0, 1])
@pytest.mark.parametrize("shape", [(3000, 1288, 256), (9408, 768, 768), (2500, 1544, 160)])
def test_gemm_k3s_exact_and_epilogues(ops, tB, shape):
    """The 128 x 128 ring kernel (vpu_gemm_set_option("k3", 4): 64 x 64 per wave, 32-deep K-steps through a three-stage ring,
    three workgroups per CU) on the shapes the round-1 128 x 128 kernel serves: exact-integer operands give the fp32 matmul bit
    for bit (ragged M and N, more tiles than workgroup slots, K of 5, 8 and 24 K-steps), every compile-time flag set."""
    M, N, K = shape
    g = torch.Generator().manual_seed(17)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    Bm = torch.randint(-3, 4, (N, K), generator=g).float()
    A[0, 1] = 3; A[1, 0] = -2; Bm[0, 1] = 1; Bm[1, 0] = -3
    ref = A @ Bm.t()
    Ad = dev(A).to(torch.bfloat16)
    Bd = dev(Bm.t().contiguous() if tB else Bm).to(torch.bfloat16)
    ldb = N if tB else K
    bias = dev(torch.randint(-4, 5, (N,), generator=g).float())
    R = dev(torch.randint(-4, 5, (M, N), generator=g).float()).to(torch.bfloat16)
    aux = dev(torch.randint(-2, 3, (M, N), generator=g).float()).to(torch.bfloat16)

    def run(k3, flags, A_=None, **kw):
        Cd = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        pre = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm_set_option("k3", k3)
        ops.gemm_set_option("k2", 0)
        try:
            ops.gemm(Ad if A_ is None else A_, Bd, Cd, M, N, K, K, ldb, N, 0, transB=bool(tB), flags=flags, preact=pre, **kw)
            name = ops.gemm_last_kernel()
            torch.cuda.synchronize()
        finally:
            ops.gemm_set_option("k3", -1)
            ops.gemm_set_option("k2", -1)
        assert ("k3s_kernel" in name) == bool(k3), name
        return Cd.float().cpu(), pre.float().cpu()

    bf = lambda t: t.to(torch.bfloat16).float()
    if tB == 0:
        out, _ = run(36, 0)
        assert torch.equal(out, bf(ref))
        out, _ = run(36, ops.EPI_BIAS, bias=bias)
        assert torch.equal(out, bf(ref + bias.cpu()))
        out, _ = run(36, ops.EPI_BIAS | ops.EPI_RELU, bias=bias)
        assert torch.equal(out, bf(torch.relu(ref + bias.cpu())))
        out, _ = run(36, ops.EPI_BIAS | ops.EPI_RESID, bias=bias, resid=R, ldr=N)
        assert torch.equal(out, bf(ref + bias.cpu() + R.float().cpu()))
        fl = ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_SAVE_DGELU
        As = (Ad.float() * 0.125).to(torch.bfloat16)      # GELU away from saturation
        o_new, p_new = run(36, fl, A_=As, bias=bias * 0.25)
        o_old, p_old = run(0, fl, A_=As, bias=bias * 0.25)
        assert torch.equal(o_new, o_old) and torch.equal(p_new, p_old)
    else:
        out, _ = run(36, 0)
        assert torch.equal(out, bf(ref))
        out, _ = run(36, ops.EPI_MULAUX, aux=aux, ldaux=N)
        assert torch.equal(out, bf(ref * aux.float().cpu()))
        out, _ = run(36, ops.EPI_DRELU, aux=aux, ldaux=N)
        assert torch.equal(out, bf(ref * (aux.float().cpu() > 0).float()))


@pytest.mark.parametrize("tB", [0, 1])
def test_gemm_grouped_skinny_form(ops, tB):
    """Groups whose problems are all skinny (<= 2560 rows, K <= 4096) run as 64 x 64 tiles with the four waves splitting K
    (gemm_bf16_skinny_grouped_kernel): bit-exact vs fp32 matmul on exact-integer operands -- fp32 accumulate output, bf16
    output with bias, ragged M / N / K, both B layouts."""
    g = torch.Generator().manual_seed(29)
    shapes = [(576, 768, 768), (576, 384, 768), (200, 136, 72), (96, 1024, 384), (50, 72, 200)]
    problems, checks = [], []
    for i, (M, N, K) in enumerate(shapes):
        Kp = (K + 7) // 8 * 8
        A = torch.randint(-2, 3, (M, K), generator=g).float()
        Bm = torch.randint(-2, 3, (N, K), generator=g).float()
        ldb = (N + 7) // 8 * 8 if tB else Kp
        Ah = torch.zeros(M, Kp); Ah[:, :K] = A
        Bh = torch.zeros((K, ldb) if tB else (N, ldb))
        if tB: Bh[:, :N] = Bm.t()
        else: Bh[:, :K] = Bm
        ldc = (N + 7) // 8 * 8
        if i % 2 == 0:
            Cd = torch.full((M, ldc), float(i + 1), device="cuda")
            kw = dict(transB=bool(tB), flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM)
            ref = A @ Bm.t() + float(i + 1)
        else:
            bias = torch.randint(-3, 4, (ldc,), generator=g).float()
            Cd = torch.zeros(M, ldc, device="cuda", dtype=torch.bfloat16)
            kw = dict(transB=bool(tB), flags=ops.EPI_BIAS, bias=dev(bias))
            ref = A @ Bm.t() + bias[:N]
        problems.append(((dev(Ah).to(torch.bfloat16), dev(Bh).to(torch.bfloat16), Cd, M, N, K, Kp, ldb, ldc, 0), kw))
        checks.append((Cd, ref, N))
    ops.gemm_grouped(problems)
    assert "skinny_grouped" in ops.gemm_last_kernel()
    torch.cuda.synchronize()
    for Cd, ref, N in checks:
        assert torch.equal(Cd[:, :N].float().cpu(), ref)


def test_gemm_grouped_skinny_wgrad_form(ops):
    """The weight-gradient orientation (both operands K-major) of the grouped skinny kernel, short reductions (the
    prompt-token layers: 576 rows) with the fused bias-gradient column sums: bit-exact accumulation into pre-filled fp32
    outputs, ragged M / N / K."""
    g = torch.Generator().manual_seed(31)
    shapes = [(768, 768, 576), (1024, 768, 576), (384, 768, 576), (200, 136, 72), (72, 520, 130)]   # (M, N, K)
    problems, checks = [], []
    for i, (M, N, K) in enumerate(shapes):
        lda, ldb = (M + 7) // 8 * 8, (N + 7) // 8 * 8
        A = torch.randint(-2, 3, (K, M), generator=g).float()
        Bm = torch.randint(-2, 3, (K, N), generator=g).float()
        Ah = torch.zeros(K, lda); Ah[:, :M] = A
        Bh = torch.zeros(K, ldb); Bh[:, :N] = Bm
        Cd = torch.full((M, ldb), float(i + 1), device="cuda")
        cs = torch.full((M,), 5.0, device="cuda") if i != 1 else None
        problems.append(((dev(Ah).to(torch.bfloat16), dev(Bh).to(torch.bfloat16), Cd, M, N, K, lda, ldb, ldb, 0),
                         dict(transA=True, transB=True, flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=cs)))
        checks.append((Cd, cs, A.t() @ Bm + float(i + 1), A.sum(0) + 5.0, N))
    ops.gemm_set_option("skinny_group", 2)      # (off by default for this orientation: measured slower in the step)
    try:
        ops.gemm_grouped(problems)
        assert "skinny_grouped_kernel<1, 1>" in ops.gemm_last_kernel()
        torch.cuda.synchronize()
    finally:
        ops.gemm_set_option("skinny_group", -1)
    for Cd, cs, ref, csref, N in checks:
        assert torch.equal(Cd[:, :N].cpu(), ref)
        if cs is not None:
            assert torch.equal(cs.cpu(), csref)


@pytest.mark.lab
def test_gemm_k2_grouped_wgrad(ops):
    """Weight-gradient groups over a long reduction in the K2 form (vpu_gemm_grouped: one global tile order cut into a
    contiguous range per XCD): bit-exact accumulation into pre-filled fp32 outputs + the fused bias-gradient column sums,
    for problems whose short tile dimension is M or N, ragged edges included."""
    g = torch.Generator().manual_seed(23)
    shapes = [(3072, 768, 2048), (768, 3072, 2048), (2304, 776, 2048), (520, 760, 2048)]   # (M, N, K): 72 + 72 + 63 + 18 tiles
    problems, checks = [], []
    for i, (M, N, K) in enumerate(shapes):
        A = torch.randint(-2, 3, (K, M), generator=g).float()
        Bm = torch.randint(-2, 3, (K, N), generator=g).float()
        Ad, Bd = dev(A).to(torch.bfloat16), dev(Bm).to(torch.bfloat16)
        Cd = torch.full((M, N), float(i + 1), device="cuda")
        cs = torch.full((M,), 5.0, device="cuda") if i != 1 else None
        problems.append(((Ad, Bd, Cd, M, N, K, M, N, N, 0),
                         dict(transA=True, transB=True, flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=cs)))
        checks.append((Cd, cs, A.t() @ Bm + float(i + 1), A.sum(0) + 5.0))
    ops.gemm_set_option("k2", 1)
    ops.gemm_set_option("k3", 0)
    try:
        ops.gemm_grouped(problems)
        assert "k2_grouped" in ops.gemm_last_kernel()
        torch.cuda.synchronize()
    finally:
        ops.gemm_set_option("k2", -1)
        ops.gemm_set_option("k3", -1)
    for Cd, cs, ref, csref in checks:
        assert torch.equal(Cd.cpu(), ref), (Cd.cpu() - ref).abs().max()
        if cs is not None:
            assert torch.equal(cs.cpu(), csref)


@pytest.mark.parametrize("opt", [pytest.param(1, marks=pytest.mark.lab), pytest.param(8, marks=pytest.mark.lab), 24])
def test_gemm_k3_grouped_wgrad(ops, opt):
    """The round-4 forms of the same groups.  vpu_gemm_set_option("k3", 1): 256 x 128 tiles in 256-thread workgroups, two per
    CU, 32-deep K-steps through a three-stage ring, no K split ("K3"); 8: 256 x 256 tiles, one 512-thread workgroup per CU,
    five-stage ring ("K4"); 24: K4 with the software-pipelined K-step (fragments of the next half-step read and the DMA issued
    between the MFMAs).  Bit-exact accumulation into pre-filled fp32 outputs + the fused bias-gradient column sums; ragged
    edges; more tiles than workgroup slots (a persistent workgroup walks several tiles of different problems); K of 64 and 66
    K-steps; and a problem given as BATCH ENTRIES -- the reduction slices of one gradient, each writing its own slab."""
    g = torch.Generator().manual_seed(29)
    shapes = [(3072, 768, 2048), (768, 3072, 2048), (2304, 776, 2048), (520, 760, 2048),         # (M, N, K): 72+72+63+18 tiles
              (3072, 768, 2112), (768, 3072, 2112), (2304, 768, 2112), (2056, 2056, 2112), (3072, 768, 2112),
              (768, 3072, 2112), (2304, 768, 2112)]                                               # 72+72+54+153+72+72+54 = 549
    problems, checks = [], []
    for i, (M, N, K) in enumerate(shapes):
        A = torch.randint(-2, 3, (K, M), generator=g).float()
        Bm = torch.randint(-2, 3, (K, N), generator=g).float()
        Ad, Bd = dev(A).to(torch.bfloat16), dev(Bm).to(torch.bfloat16)
        Cd = torch.full((M, N), float(i + 1), device="cuda")
        cs = torch.full((M,), 5.0, device="cuda") if i != 1 else None
        problems.append(((Ad, Bd, Cd, M, N, K, M, N, N, 0),
                         dict(transA=True, transB=True, flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=cs)))
        checks.append((Cd, cs, A.t() @ Bm + float(i + 1), A.sum(0) + 5.0))
    # one gradient [392, 264] over 3 x 2048 rows as three batch entries writing raw slabs (+ their column-sum slabs)
    S, M, N, L = 3, 392, 264, 2048
    A = torch.randint(-2, 3, (S * L, M), generator=g).float()
    Bm = torch.randint(-2, 3, (S * L, N), generator=g).float()
    Ad, Bd = dev(A).to(torch.bfloat16), dev(Bm).to(torch.bfloat16)
    slab = torch.full((S, M, N), -7.0, device="cuda")
    cslab = torch.zeros(S, M, device="cuda")
    sliced = ((Ad, Bd, slab, M, N, L, M, N, N, 0),
              dict(transA=True, transB=True, flags=ops.EPI_OUT_F32, colsum=cslab, batch=S, sA=(L * M, 0), sB=(L * N, 0), sC=(M * N, 0)))
    ops.gemm_set_option("k3", opt)
    try:
        ops.gemm_grouped(problems[:4])           # 225 tiles of 256 x 128
        if opt == 1:                             # (one workgroup per CU would have nobody to overlap with -> K2)
            assert "k2_grouped" in ops.gemm_last_kernel()
        ops.gemm_grouped(problems[4:] + [sliced])
        assert {1: "k3_grouped", 8: "k4_grouped", 24: "k4p_grouped"}[opt] in ops.gemm_last_kernel()
        torch.cuda.synchronize()
    finally:
        ops.gemm_set_option("k3", -1)
    for Cd, cs, ref, csref in checks:
        assert torch.equal(Cd.cpu(), ref), (Cd.cpu() - ref).abs().max()
        if cs is not None:
            assert torch.equal(cs.cpu(), csref)
    for z in range(S):
        az, bz = A[z * L:(z + 1) * L], Bm[z * L:(z + 1) * L]
        assert torch.equal(slab[z].cpu(), az.t() @ bz), z
        assert torch.equal(cslab[z].cpu(), az.sum(0)), z


def test_dropout_mask_and_fill(ops):
    """vpu_dropout_mask: values in {0, 1 / keep}, kept fraction ~ keep, a new mask per call (the call number lives on the
    device and is advanced by the launch -- also inside a replayed hipGraph), the same stream of masks for the same seed;
    ops.zero_ (vpu_fill_f32 with 16-byte stores) on aligned and unaligned, odd-sized buffers."""
    from pvpuformer_amd import ops as O
    O._drop_state.clear()
    a = [O.dropout_mask(12, 256, 0.9, "cuda", seed=77) for _ in range(3)]
    assert all(tuple(m.shape) == (12, 256) for m in a)
    for m in a:
        vals = set(torch.unique(m).tolist())
        assert vals <= {0.0, float(torch.tensor(1 / 0.9, dtype=torch.float32))} and len(vals) == 2
        assert 0.85 < float((m > 0).float().mean()) < 0.95
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[1], a[2])
    g = torch.cuda.CUDAGraph()
    out = []
    with torch.cuda.graph(g):
        captured = O.dropout_mask(12, 256, 0.9, "cuda")
    for _ in range(2):
        g.replay(); torch.cuda.synchronize(); out.append(captured.clone())
    assert not torch.equal(out[0], out[1])
    O._drop_state.clear()
    b = [O.dropout_mask(12, 256, 0.9, "cuda", seed=77) for _ in range(3)]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    # (seed, call number) live in device memory: a launch captured in a graph follows a re-seeding, and the pair can be
    # saved and restored (FusedAdam.state_dict carries it)
    O.dropout_seed("cuda", 77)
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(captured, a[0])
    assert O.dropout_state("cuda") == {"seed": 77, "calls": 1}
    saved = O.dropout_state("cuda")
    nxt = O.dropout_mask(12, 256, 0.9, "cuda")
    assert torch.equal(nxt, a[1])
    O.dropout_seed("cuda", 5)
    O.set_dropout_state("cuda", saved)
    assert torch.equal(O.dropout_mask(12, 256, 0.9, "cuda"), a[1])
    O._drop_state.clear()
    for n, off in ((1 << 20, 0), (1000003, 1), (7, 3), (3, 0)):
        buf = torch.full((n + 8,), 5.0, device="cuda")
        O.zero_(buf[off:off + n])
        assert float(buf[off:off + n].abs().sum()) == 0.0 and float(buf[:off].sum()) == 5.0 * off and float(buf[off + n:].sum()) == 5.0 * (8 - off)
    h = torch.full((1001,), 3.0, device="cuda", dtype=torch.bfloat16)[:1000]
    assert float(O.zero_(h).float().abs().sum()) == 0.0


def test_fill_ranges(ops):
    """vpu_fill_ranges_f32: every listed range is zeroed, nothing else is touched (ragged chunk tails, 150 ranges)."""
    g = torch.Generator().manual_seed(5)
    n = 3_000_000
    t = torch.full((n,), 7.0, device="cuda")
    ranges, pos = [], 0
    for i in range(150):
        gap = 4 * int(torch.randint(1, 300, (1,), generator=g))
        ln = 4 * int(torch.randint(1, 9000, (1,), generator=g))
        ranges.append((pos + gap, ln))
        pos += gap + ln
    assert pos < n
    ops.zero_ranges_(t, ranges)
    torch.cuda.synchronize()
    exp = torch.full((n,), 7.0)
    for o, ln in ranges:
        exp[o:o + ln] = 0.0
    assert torch.equal(t.cpu(), exp)


def test_gemm_k4_distributed_column_sums(ops):
    """cs_tn > 1 (round 4): every column tile of a packed weight-gradient problem sums the bias gradient over its own slice
    of the reduction and WRITES its partial into its row of a [cs_tn, M] slab -- also for a problem cut along its columns
    into two descriptors (cs_t0) and for reduction slices (batch entries: rows z * cs_tn + t).  Exact-integer operands: the
    slab rows must add up to the column sums of dY bit for bit, C must be the plain product, untouched slab words stay."""
    R, M, N = 4224, 512, 768               # reduction rows (132 K-steps of 32: 44 per column tile; slices of 2112), G is [M, N]
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-2, 3, (R, M), generator=g).float()
    B = torch.randint(-2, 3, (R, N), generator=g).float()
    Ad, Bd = A.cuda().to(torch.bfloat16), B.cuda().to(torch.bfloat16)
    ref, cref = A.t() @ B, A.sum(0)
    tn = N // 256
    ops.gemm_set_option("k3", 24)
    try:
        # one descriptor
        C = torch.full((M, N), 9.0, device="cuda")
        slab = torch.full((tn, M), 5.0, device="cuda")
        ops.gemm_grouped([((Ad, Bd, C, M, N, R, M, N, N, 0), dict(transA=True, transB=True, flags=ops.EPI_OUT_F32, colsum=slab, cs_tn=tn, cs_t0=0, cs_ld=M))])
        torch.cuda.synchronize()
        assert "k4p" in ops.gemm_last_kernel()
        assert torch.equal(C.cpu(), ref) and torch.equal(slab.sum(0).cpu(), cref)
        assert (slab.abs().sum(1) > 0).all()          # every column tile contributed a slice
        # the same problem cut along its columns: 256 + 512 columns
        C2 = torch.full((M, N), 9.0, device="cuda")
        slab2 = torch.full((tn, M), 5.0, device="cuda")
        kw = dict(transA=True, transB=True, flags=ops.EPI_OUT_F32, colsum=slab2, cs_tn=tn, cs_ld=M)
        ops.gemm_grouped([((Ad, Bd, C2, M, 256, R, M, N, N, 0), dict(kw, cs_t0=0)),
                          ((Ad, (Bd, 256), (C2, 256), M, N - 256, R, M, N, N, 0), dict(kw, cs_t0=1))])
        torch.cuda.synchronize()
        assert torch.equal(C2.cpu(), ref) and torch.equal(slab2.cpu(), slab.cpu())
        # two reduction slices as batch entries: slabs of C and rows z * tn + t of the column-sum slab
        L = R // 2
        if L % 64 == 0:
            C3 = torch.zeros(2, M, N, device="cuda")
            slab3 = torch.full((2 * tn, M), 5.0, device="cuda")
            ops.gemm_grouped([((Ad, Bd, C3, M, N, L, M, N, N, 0), dict(kw, colsum=slab3, cs_t0=0, batch=2, sA=(L * M, 0), sB=(L * N, 0), sC=(M * N, 0)))])
            torch.cuda.synchronize()
            assert torch.equal(C3.sum(0).cpu(), ref) and torch.equal(slab3.sum(0).cpu(), cref)
            assert torch.equal(slab3[:tn].sum(0).cpu(), A[:L].sum(0))
        # the classic kernels refuse the form
        ops.gemm_set_option("k3", 0)
        with pytest.raises(Exception, match="cs_tn"):
            ops.gemm_grouped([((Ad, Bd, C, M, N, R, M, N, N, 0), dict(transA=True, transB=True, flags=ops.EPI_OUT_F32, colsum=slab, cs_tn=tn, cs_t0=0, cs_ld=M))])
    finally:
        ops.gemm_set_option("k3", -1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("C", [64, 192, 200])
def test_pixel_unshuffle2_sums(ops, dtype, C):
    """vpu_pixel_unshuffle2_sums: the same bytes as pixel_shuffle2(inverse=True), and partial rows whose sum is the per-channel
    sum of the fine map (the ConvTranspose2d bias gradient) -- exact with small-integer data, for channel counts whose C / 8 does
    and does not divide 256."""
    B, h, w = 3, 5, 7
    g = torch.Generator().manual_seed(2)
    x = torch.randint(-3, 4, (B * 2 * h * 2 * w, C), generator=g).float().cuda().to(dtype)
    ref = torch.empty(B * h * w, 4 * C, device="cuda", dtype=dtype)
    ops.pixel_shuffle2(x, ref, None, B, h, w, C, inverse=True)
    out = torch.full_like(ref, 9.0)
    part, nb = ops.pixel_unshuffle2_sums(x, out, B, h, w, C)
    torch.cuda.synchronize()
    assert tuple(part.shape) == (nb, C) and nb % (C // 8) == 0
    assert torch.equal(out, ref)
    assert torch.equal(part.sum(0).cpu(), x.float().sum(0).cpu())


def test_gemm_bf16_accumulate_in_place_on_k3s(ops):
    """A bf16 output accumulated in place (flags = ACCUM, transB, K <= 384: the neck's second gradient into an activation
    gradient) runs as the residual form of the 128 x 128 ring kernel with the residual = C: exact-integer operands give
    C_old + A B bit for bit, also with a ragged last tile."""
    for M, N, K in ((9408, 768, 384), (4200, 264, 256), (1000, 264, 256)):
        g = torch.Generator().manual_seed(7)
        A = torch.randint(-2, 3, (M, K), generator=g).float()
        B = torch.randint(-2, 3, (K, N), generator=g).float()
        C0 = torch.randint(-8, 9, (M, N), generator=g).float()
        Ad, Bd = A.cuda().to(torch.bfloat16), B.cuda().to(torch.bfloat16)
        C = C0.cuda().to(torch.bfloat16)
        ops.gemm(Ad, Bd, C, M, N, K, K, N, N, 0, transB=True, flags=ops.EPI_ACCUM)
        torch.cuda.synchronize()
        if M >= 4096:        # (the small problem goes to the skinny kernel: same contract)
            assert "k3s_kernel<0, 1, 64>" in ops.gemm_last_kernel(), ops.gemm_last_kernel()
        assert torch.equal(C.float().cpu(), (C0 + A @ B).to(torch.bfloat16).float())


@pytest.mark.parametrize("tB", [0, 1])
def test_gemm_k2_grouped_forward_forms(ops, tB):
    """Forward / dgrad GROUPS on the 256 x 128 kernel (the DMA neck's image-side K / V projections: two 9408 x 384 x 768
    problems in one launch).  Problems that share one flag set -- bias only (forward) / none (dgrad) -- take the compile-time
    form with the direct epilogue (round 4), any other mix the run-time epilogue: both must give the fp32 product of
    exact-integer operands bit for bit, with a different bias per problem and a ragged last row tile."""
    g = torch.Generator().manual_seed(9)
    probs, refs = [], []
    for (M, N, K) in ((9408, 384, 768), (9408, 384, 768), (2000, 256, 512)):
        A = torch.randint(-2, 3, (M, K), generator=g).float()
        W = torch.randint(-2, 3, (N, K), generator=g).float()
        b = torch.randint(-4, 5, (N,), generator=g).float()
        Ad = A.cuda().to(torch.bfloat16)
        Wd = (W.t().contiguous() if tB else W).cuda().to(torch.bfloat16)
        C = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        kw = dict(transB=bool(tB), flags=0 if tB else ops.EPI_BIAS, bias=None if tB else b.cuda())
        probs.append(((Ad, Wd, C, M, N, K, K, N if tB else K, N, 0), kw))
        refs.append((A @ W.t() + (0 if tB else b)).to(torch.bfloat16).float())
    ops.gemm_grouped(probs)
    torch.cuda.synchronize()
    assert f"k2_grouped_fl_kernel<0, {tB}, {0 if tB else 1}>" in ops.gemm_last_kernel(), ops.gemm_last_kernel()
    for (args, _), ref in zip(probs, refs):
        assert torch.equal(args[2].float().cpu(), ref)
    if not tB:      # a mixed group (one problem with a ReLU): the run-time epilogue
        for (args, kw) in probs:
            args[2].fill_(7.0)
        probs[2][1]["flags"] = ops.EPI_BIAS | ops.EPI_RELU
        ops.gemm_grouped(probs)
        torch.cuda.synchronize()
        assert "k2_grouped_kernel<0, 0, false>" in ops.gemm_last_kernel(), ops.gemm_last_kernel()
        refs[2] = refs[2].clamp_min(0)
        for (args, _), ref in zip(probs, refs):
            assert torch.equal(args[2].float().cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pixel_shuffle2_gn_stats(ops, dtype):
    """vpu_pixel_shuffle2_gn_stats + vpu_groupnorm_apply == vpu_pixel_shuffle2 + vpu_groupnorm_fwd: the same map bit for bit,
    the same mean / rstd / normalised output up to the summation order of the statistics (float64 partials)."""
    B, h, w, C = 3, 9, 11, 48
    g = torch.Generator().manual_seed(4)
    t = (torch.randn(B * h * w, 4 * C, generator=g)).cuda().to(dtype)
    bias = torch.randn(C, generator=g).cuda()
    gw, gb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    HW = 4 * h * w
    nch = ops.groupnorm_nchunk()
    y0 = torch.empty(B * HW, C, device="cuda", dtype=dtype)
    ops.pixel_shuffle2(t, y0, bias, B, h, w, C)
    o0, m0, r0 = torch.empty_like(y0), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    st0 = torch.empty(B, nch, 2, device="cuda", dtype=torch.float64)
    ops.groupnorm_fwd(y0, gw, gb, o0, m0, r0, st0, B, HW, C, 1e-5, True)
    y1 = torch.full_like(y0, 3.0)
    st1 = torch.empty(B, nch, 2, device="cuda", dtype=torch.float64)
    ops.pixel_shuffle2_gn_stats(t, y1, bias, st1, B, h, w, C)
    o1, m1, r1 = torch.empty_like(y0), torch.empty(B, device="cuda"), torch.empty(B, device="cuda")
    ops.groupnorm_apply(y1, gw, gb, o1, m1, r1, st1, B, HW, C, 1e-5, True)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0)
    assert torch.allclose(st1.sum(1), st0.sum(1), rtol=1e-6, atol=1e-4)      # (fp32 partials of 32 values against 8, then float64)
    assert torch.allclose(m1, m0, rtol=1e-6, atol=1e-7) and torch.allclose(r1, r0, rtol=1e-6, atol=1e-7)
    assert torch.allclose(o1.float(), o0.float(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-5, atol=1e-2 if dtype == torch.bfloat16 else 1e-5)


def test_lab_library_families():
    """The kernel families the engine does not dispatch (K3 forward / grouped forms, the non-pipelined K4, the K2 grouped weight
    gradient, the three-stage ring forms) live in the laboratory library only (`csrc/build.sh diag`: -DVPU_DIAG -DVPU_LAB); the
    product library rejects their options.  Their exactness tests (marked `lab`) run here in a child process that loads that
    library (VPU_LIB_DIAG=1), so they stay covered by the same `pytest -m gpu` run."""
    import subprocess
    import sys
    from pvpuformer_amd import ops as pops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("VPU_LIB_DIAG", "0") != "1":
        for name, v in (("ring", 2), ("k3", 3), ("k3", 1)):
            with pytest.raises(RuntimeError, match="laboratory library"):
                pops.gemm_set_option(name, v)
    assert os.path.exists(os.path.join(root, "pvpuformer_amd", "libvpu_hip_diag.so")), "build it: bash pvpuformer_amd/csrc/build.sh diag"
    env = dict(os.environ, VPU_LIB_DIAG="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_ops_gpu.py"), "-q", "-x", "-m", "gpu and lab",
                        "-p", "no:cacheprovider"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail


def test_argument_errors_answer_before_any_launch(ops):
    """Error behaviour of the C ABI: a call whose arguments the kernels cannot take returns a negative VPU_ERR_* with a message
    naming the rule (raised as VpuError by the wrappers) BEFORE anything is launched -- the stream stays clean (a following good
    call works, nothing is pending in hipGetLastError) and output buffers are untouched."""
    import ctypes as C
    from pvpuformer_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    z = lambda *shape, dt=torch.bfloat16: torch.zeros(*shape, device="cuda", dtype=dt)
    sentinel = torch.full((64, 64), 7.0, device="cuda", dtype=torch.bfloat16)
    P = lambda t: C.c_void_p(t.data_ptr())
    cases = {
        "sum_groups: n % 8": lambda: lib.vpu_sum_groups(P(sentinel), P(sentinel), 2, 2, 12, _lib.BF16, st),
        "sum_groups: S < 1": lambda: lib.vpu_sum_groups(P(sentinel), P(sentinel), 2, 0, 16, _lib.BF16, st),
        "fanout_add: 5 destinations": lambda: lib.vpu_fanout_add(P(sentinel), (C.c_void_p * 5)(*[sentinel.data_ptr()] * 5), (C.c_int32 * 5)(), 5, 64, _lib.BF16, st),
        "fanout_add: null destination": lambda: lib.vpu_fanout_add(P(sentinel), (C.c_void_p * 2)(sentinel.data_ptr(), None), (C.c_int32 * 2)(), 2, 64, _lib.BF16, st),
        "cast2d_batched: no job": lambda: lib.vpu_cast2d_batched((_lib.CastJob * 1)(), 0, st),
        "cast2d_batched: empty job": lambda: lib.vpu_cast2d_batched((_lib.CastJob * 1)(), 1, st),
        "attn_combine: S = 9": lambda: lib.vpu_attn_combine(P(sentinel), P(sentinel), P(sentinel), P(sentinel), 1, 1, 8, 8, 9, 8, 8, st),
        "attn_combine: head dim % 8": lambda: lib.vpu_attn_combine(P(sentinel), P(sentinel), P(sentinel), P(sentinel), 1, 1, 8, 12, 2, 16, 16, st),
        "gemm_grouped: 0 problems": lambda: lib.vpu_gemm_grouped((_lib.GemmDesc * 1)(), 0, st),
        "gemm_grouped: 17 problems": lambda: lib.vpu_gemm_grouped((_lib.GemmDesc * 17)(), 17, st),
        "adam: step 0": lambda: lib.vpu_adam_step(P(sentinel), P(sentinel), P(sentinel), P(sentinel), None, 64, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 1.0, st),
        "p2cl_up: W % 4": lambda: lib.vpu_p2cl_up_fwd_bwd(P(sentinel), P(sentinel), None, None, P(sentinel), P(sentinel), 1.0, 1, 2, 8, 8, 30, 30, st),
        "bilinear_cl_bwd: C % 8": lambda: lib.vpu_bilinear_cl_bwd(P(sentinel), 12, P(sentinel), 12, 1, 2, 2, 4, 4, 12, _lib.BF16, st),
        "convseg_bwd: C > 512": lambda: lib.vpu_convseg_bwd(P(sentinel), P(sentinel), P(sentinel), None, P(sentinel), 0, P(sentinel), P(sentinel), 64, 64, 1024, _lib.BF16, st),
        "head_grad_fused: fp32": lambda: lib.vpu_head_grad_fused(P(sentinel), P(sentinel), P(sentinel), P(sentinel), P(sentinel), P(sentinel), None, P(sentinel), P(sentinel), P(sentinel), 64, 64, 64, _lib.F32, st),
        "patch_im2col_prenorm: H % P": lambda: lib.vpu_patch_im2col_prenorm(P(sentinel), P(sentinel), P(sentinel), 1, 30, 30, 16, 2, _lib.BF16, st),
        "gemm_set_option: unknown": lambda: lib.vpu_gemm_set_option(b"no_such_option", 1),
        "attn_set_option: unknown": lambda: lib.vpu_attn_set_option(b"no_such_option", 1),
        "gemm_set_option: k2 out of range": lambda: lib.vpu_gemm_set_option(b"k2", 9),
    }
    if os.environ.get("VPU_LIB_DIAG", "0") != "1":
        cases["debug_gemm_times: product library"] = lambda: lib.vpu_debug_gemm_times(P(sentinel))
    for what, call in cases.items():
        rc = call()
        msg = lib.vpu_last_error().decode()
        assert rc < 0 and msg, (what, rc, msg)
        print(f"[abi errors] {what:40s} -> {rc}: {msg[:90]}")
    # a misaligned GEMM operand has its own code
    A = z(64, 72)
    with pytest.raises(_lib.VpuError, match="16-byte"):
        ops.gemm(A.view(-1)[1:64 * 64 + 1].view(64, 64), A[:, :64].contiguous(), z(64, 64), 64, 64, 64, 64, 64, 64, _lib.BF16)
    torch.cuda.synchronize()
    assert torch.all(sentinel == 7.0)
    # ... and the library still works
    a, b = (torch.randint(-2, 3, (64, 64), device="cuda").to(torch.bfloat16) for _ in range(2))
    c = z(64, 64)
    ops.gemm(a, b, c, 64, 64, 64, 64, 64, 64, _lib.BF16)
    torch.cuda.synchronize()
    assert torch.equal(c.float(), a.float() @ b.float().t())
