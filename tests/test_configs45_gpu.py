"""Configs 4 and 5 of BASELINE.json at their OWN workloads: ViT-L (D = 1024, depth 24, 16 heads; click / box / scribble) and
ViT-H (D = 1280, depth 32, 16 heads of 80, patch 14 -> 1024 tokens, 16 x 16 windows), against fixtures recorded from the
reference itself at full depth (tests/golden/vitl.npz, vith.npz: oracle/make_golden.py vitl vith).

(a) exact-fp32 engine mode vs the reference: mask logits within 1e-3 relative, the loss scalars within 2e-4, EVERY gradient
    norm within 2e-3 and the stored gradients / slices element-wise, per prompt mode;
(b) bf16 at the benchmark's per-GPU batch (ViT-L B = 8 -> M = 6272 token rows, ViT-H B = 12 -> M = 12288): samples 0-1 of
    the batch are the fixture's, so their logits are compared with the reference's (samples are independent); the flat
    gradient of the B-sample step equals the sum of its B / 2 two-sample micro-batches (other tile shapes, other kernels);
    a spy asserts that the long-reduction weight gradients leave in groups that fill their rounds of 256 tiles (256-tile
    groups for ViT-L, 450 / 750 for ViT-H) and that the head-dim-80 attention instantiation really ran;
(c) one mixed click / box / scribble ``VPUTrainStep`` on the ViT-L model."""
import os
import random

import numpy as np
import pytest
import torch

import vpu_oracle as vo
from test_api_cpu import make_model
from test_model_gpu import _check_fp32_grads_against_fixture, _relerr
from test_oracle_golden import cfg_from_fixture

pytestmark = pytest.mark.gpu
_cache = {}


def _model(golden_dir, name):
    """One model per fixture for the whole module (a ViT-H engine holds ~9 GB of parameters, gradients and shadows)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if name not in _cache:
        _cache.clear()
        torch.cuda.empty_cache()
        fx = np.load(os.path.join(golden_dir, name))
        cfg = cfg_from_fixture(fx)
        sd = vo.synth_state_dict(vo.param_shapes(cfg), seed=0)
        model = make_model(cfg).cuda()
        model.load_state_dict(sd, strict=True)
        del sd
        B = int(fx["B"])
        batch = vo.synth_batch(B, cfg["img"], seed=int(fx["images_seed"]))
        img4 = torch.cat([batch["images"], torch.zeros(B, 1, cfg["img"], cfg["img"])], 1)
        img4[0, 3] = torch.sigmoid(4 * (batch["instances"][0, 0] - 0.5))
        _cache[name] = (fx, cfg, model, batch, img4)
    return _cache[name]


def _prompts(fx, batch, ptype):
    pts, boxes = batch["points"].cuda(), batch["boxes"].cuda()
    if ptype == 0:
        return pts, None
    return pts, (pts, boxes, [fx["scribbles"], fx["rects"]] if ptype == 2 else None)


CASES = [("vitl.npz", "click", 0), ("vitl.npz", "box", 1), ("vitl.npz", "scribble", 2), ("vith.npz", "click", 0), ("vith.npz", "box", 1)]


@pytest.mark.parametrize("fixture,mode,ptype", CASES)
def test_full_depth_fp32_matches_reference(golden_dir, fixture, mode, ptype):
    fx, cfg, model, batch, img4 = _model(golden_dir, fixture)
    assert (cfg["embed_dim"], cfg["depth"]) in ((1024, 24), (1280, 32))
    model.set_compute_dtype("f32")
    model.eval()
    model.zero_grad()
    pts, prompts = _prompts(fx, batch, ptype)
    if ptype == 2:
        random.seed(int(fx["scribble_seed"]))        # the scribble vectors draw from the global ``random`` state (ops.py:274,290)
    out = model(img4.cuda(), pts, prompts, ptype)
    assert _relerr(out["instances"][..., ::7, ::7].detach().cpu().numpy(), fx[f"{mode}_instances_sub"]) < 1e-3
    assert _relerr(out["instances_aux"][:, ::6, ::7, ::7].detach().cpu().numpy(), fx[f"{mode}_instances_aux_sub"]) < 1e-3
    gt = batch["instances"].cuda()
    total, parts = vo.step_loss(out, gt, vo.ed_mask_label(gt))
    np.testing.assert_allclose([total.item(), parts["nfl"].item(), parts["dice"].item(), parts["p2cl"].item()],
                               fx[f"{mode}_loss"], rtol=2e-4)
    total.backward()
    _check_fp32_grads_against_fixture(model, fx, mode)


def _step(eng, x, pts, gt, ptype=0, boxes=None):
    from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses
    inst, _ = eng.forward(x, pts, boxes, ptype, None, training=True, materialize_aux=False)
    losses, d_inst, d_sim = vpu_step_losses(inst, None, gt, None, None, iter_weight=1.0, sim_low=eng.sim_low)
    eng.backward(d_inst, None, d_sim_low=d_sim)
    return inst, losses


@pytest.mark.parametrize("fixture,B,logit_tol", [("vitl.npz", 8, 3.5e-2), ("vith.npz", 12, 6e-2)])
def test_bench_batch_bf16_step(golden_dir, fixture, B, logit_tol):
    from pvpuformer_amd import ops
    fx, cfg, model, batch, img4 = _model(golden_dir, fixture)
    model.set_compute_dtype("bf16")
    model.train()
    D, depth = cfg["embed_dim"], cfg["depth"]
    more = vo.synth_batch(B - 2, cfg["img"], seed=200)
    x = torch.cat([img4, torch.cat([more["images"], torch.sigmoid(3 * (more["instances"] - 0.4))], 1)], 0).cuda().contiguous()
    pts = torch.cat([batch["points"], more["points"]], 0).cuda()
    gt = torch.cat([batch["instances"], more["instances"]], 0).cuda()
    eng = model._ensure_engine()
    eng.refresh_weights()
    eng.zero_grad()
    gemms, attns = [], []
    eng_ops = __import__("pvpuformer_amd.engine", fromlist=["ops"]).ops
    orig = {n: getattr(eng_ops, n) for n in ("gemm", "gemm_grouped", "attn_fwd", "attn_bwd")}

    def wrap(name, sink, last):
        def f(*a, **k):
            r = orig[name](*a, **k)
            sink.append(last())
            return r
        return f
    descs = []
    def grouped(p):
        # problems are (args, kwargs) of ops.gemm: args = (A, B, C, M, N, K, ...); 256 x 128 tiles of the long reductions
        # (round 4: the packed queue cuts problems at 256-row / 256-column blocks, so pieces of 256 rows count too)
        descs.append(sum(-(-int(a[3]) // 256) * -(-int(a[4]) // 128) for a, _ in p if int(a[5]) >= 2048 and int(a[3]) >= 256 and int(a[4]) >= 256))
        orig["gemm_grouped"](p)
        gemms.append(ops.gemm_last_kernel())
    try:
        eng_ops.gemm = wrap("gemm", gemms, ops.gemm_last_kernel)
        eng_ops.gemm_grouped = grouped
        eng_ops.attn_fwd = wrap("attn_fwd", attns, ops.attn_last_kernel)
        eng_ops.attn_bwd = wrap("attn_bwd", attns, ops.attn_last_kernel)
        inst, losses = _step(eng, x, pts, gt)
        torch.cuda.synchronize()
    finally:
        for n, f in orig.items():
            setattr(eng_ops, n, f)
    # ---- the instantiations this configuration is meant to select really ran
    used = set(gemms)
    assert any(k.startswith("gemm_bf16_k4p_grouped_kernel<1, 1, true") for k in used) or "gemm_bf16_k2_grouped_kernel<1, 1, true>" in used, sorted(used)
    # (round 6: the forward / dgrad forms of the blocks run on K5, the two-tile ping-pong family -- all but fc1's bias + GELU + GELU',
    # which keeps the 256-column K2 form)
    assert any(k.startswith("gemm_bf16_k2_kernel<0, 0,") for k in used), sorted(used)
    assert any(k.startswith("gemm_bf16_k5_kernel<0, ") for k in used) and any(k.startswith("gemm_bf16_k5_kernel<1, ") for k in used), sorted(used)
    # the long-reduction weight gradients (M = 6272 / 12288 token rows: 98+ K-steps per output tile) leave in PACKED launches
    # (round 4, Engine._wgrad: every geometry packs under the K4 form): one round of 256 tiles of 256 x 256 = 512 of the
    # 256 x 128 units counted here, big problems cut at 256-row / 256-column blocks so that the round is full, up to nine small
    # ones riding (the counter above skips pieces narrower than 256, so a launch may show fewer than its 512).  Every ViT
    # block's tiles must be in such launches, and most launches must be near-full rounds.
    per_block = {1024: 384, 1280: 600}[D]
    big = [t for t in descs if t >= 200]
    assert sum(big) >= per_block * (depth - 1), (sorted(set(descs)), per_block * depth)
    assert max(big) >= 450
    assert sum(t >= 400 for t in big) >= 0.75 * len(big), sorted(big)
    hd = D // cfg["num_heads"]
    if hd == 80:
        want = {"attn_fwd_lean_kernel<128, 2, 96>", "attn_bwd_dq_lean_kernel<128, 2, 96> attn_bwd_dkdv_lean_kernel<128, 1, 96>"}
    else:
        want = {"attn_fwd_lean_kernel<64, 2, 64>", "attn_bwd_dq_lean_kernel<64, 2, 64> attn_bwd_dkdv_lean_kernel<64, 2, 64>"}
    assert want <= set(attns), sorted(set(attns))
    assert sum(a.startswith("attn_fwd") for a in attns) == depth            # one per ViT block (the neck's go through xattn_*)
    # ---- samples 0-1 are the fixture's: their logits against the REFERENCE's (bf16 bound, measured value printed)
    err = _relerr(inst[:2, :, ::7, ::7].float().cpu().numpy(), fx["click_instances_sub"])
    print(f"{fixture} B={B} bf16 logits vs reference: {err:.3e}")
    assert err < logit_tol, err          # measured 1.85e-2 (ViT-L, 24 blocks) / 3.43e-2 (ViT-H, 32 blocks): bound = 1.8 x that
    for v in losses.values():
        assert torch.isfinite(v).all()
    # ---- the same B samples as B / 2 micro-batches of two, accumulated: sum_j grad(mean over 2) = (B / 2) * grad(mean over B)
    total, rows = _micro_vs_big(eng, x, pts, gt, B)
    _assert_independent(total, rows, 4e-2, f"{fixture} B={B}:")


@pytest.mark.parametrize("fixture,B", [("vitl.npz", 8), ("vith.npz", 12)])
def test_training_step_is_bitwise_reproducible(golden_dir, fixture, B):
    """The same step three times (the reference-shaped synthetic weights of the fixtures: scores large enough for the
    attention kernels' rescaling path), freed memory poisoned with NaNs in between: logits, low-resolution similarities and
    every gradient agree bit for bit, nothing is NaN -- no kernel depends on timing or reads what it has not written.
    (This is the check that exposed the attention forward's missing wait states: ViT-H's neck output changed from launch
    to launch.)"""
    fx, cfg, model, batch, img4 = _model(golden_dir, fixture)
    model.set_compute_dtype("bf16")
    model.train()
    more = vo.synth_batch(B - 2, cfg["img"], seed=200)
    x = torch.cat([img4, torch.cat([more["images"], torch.sigmoid(3 * (more["instances"] - 0.4))], 1)], 0).cuda().contiguous()
    pts = torch.cat([batch["points"], more["points"]], 0).cuda()
    gt = torch.cat([batch["instances"], more["instances"]], 0).cuda()
    eng = model._ensure_engine()
    eng.refresh_weights()
    runs = []
    for r in range(3):
        if r:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(16)]
            junk += [torch.full((n,), float("nan"), device="cuda") for n in (256, 4096, 65536, 200000) for _ in range(300)]
            del junk
        eng.zero_grad()
        inst, _ = _step(eng, x, pts, gt)
        torch.cuda.synchronize()
        runs.append((inst.clone(), eng.sim_low.clone(), eng.gflat.clone()))
    for t in runs[0]:
        assert torch.isfinite(t).all()
    for other in runs[1:]:
        for name, a, b in zip(("logits", "sim_low", "gradients"), runs[0], other):
            assert torch.equal(a, b), f"{name} differ between two runs of the same step"


TOKEN_PATH = ("neck.att", "neck.ffn_layer", "head.ffn_layer")     # the 48-prompt-token side: q / k projection gradients are small
                                                                  # differences of large terms (near-uniform softmax at random init)


def _micro_vs_big(eng, x, pts, gt, B):
    """(whole-buffer relative L2 distance, [(relative L2 distance, cosine, name, norm)] worst first) between the gradient of
    one B-sample step (times B / 2) and the accumulated gradients of its two-sample micro-batches."""
    eng.zero_grad()
    _step(eng, x, pts, gt)
    big = eng.gflat.double() * (B / 2)
    eng.zero_grad()
    for j in range(0, B, 2):
        _step(eng, x[j:j + 2].contiguous(), pts[j:j + 2].contiguous(), gt[j:j + 2].contiguous())
    torch.cuda.synchronize()
    micro = eng.gflat.double()
    out = []
    for n, (off, shape, numel) in eng.names.items():
        a, b = big[off:off + numel], micro[off:off + numel]
        if float(b.norm()) > 0:
            out.append((float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm())), n, float(b.norm())))
    return float((big - micro).norm() / micro.norm()), sorted(out, reverse=True)


def _assert_independent(total, rows, image_tol, report, total_tol=1e-2):
    """bf16 bound: the image path (backbone, patch embeddings, FPN, head convolutions) per tensor within ``image_tol``; the
    prompt-token path by direction (cosine > 0.98) for every tensor that is not at its noise floor (norm above 1e-3 of the
    path's largest); the whole buffer within ``total_tol``.  Measured (printed): ViT-L B = 8 0.55 % whole buffer, 3.4 %
    worst image-path tensor (pos_embed), lowest token-path cosine 0.992; ViT-H B = 12 0.41 % / 2.3 % / 0.985 (the same in every
    process since the forward attention kernels' running maximum waits for its score MFMAs: csrc/attention.hip, max8)."""
    image = [r for r in rows if not r[2].startswith(TOKEN_PATH) and r[3] > 1e-3]
    token = [r for r in rows if r[2].startswith(TOKEN_PATH)]
    floor = 1e-3 * max(r[3] for r in token)
    token = [r for r in token if r[3] > floor]
    print(report, "whole buffer", round(total, 5), "| image path worst", [(round(r[0], 4), r[2]) for r in image[:2]],
          "| token path lowest cosine", sorted((round(r[1], 4), r[2]) for r in token)[:2])
    assert total < total_tol, total
    assert len(image) > 100 and image[0][0] < image_tol, image[:4]
    assert len(token) > 40 and min(r[1] for r in token) > 0.98, sorted((r[1], r[2]) for r in token)[:4]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_samples_are_independent_tiny(golden_dir, dtype):
    """Linearity over samples (a size-independent property of the path: no batch statistics anywhere): the
    gradient of a B = 8 step equals the accumulated gradients of its four two-sample micro-batches.  Exact-fp32 engine
    mode: to fp32 summation order; bf16: to the rounding noise of different kernel selections for M = 8 x 784 vs 2 x 784
    rows (the prompt-token path, 384 vs 96 rows, is the noisiest)."""
    from test_model_gpu import _setup
    fx, cfg, sd, model, batch, img4 = _setup(golden_dir, "tiny.npz", dtype)
    model.train()
    B = 8
    data = vo.synth_batch(B, cfg["img"], seed=77)
    x = torch.cat([data["images"], torch.sigmoid(3 * (data["instances"] - 0.4))], 1).cuda().contiguous()
    eng = model._ensure_engine()
    eng.refresh_weights()
    total, rows = _micro_vs_big(eng, x, data["points"].cuda(), data["instances"].cuda(), B)
    if dtype == "f32":
        sizeable = [r for r in rows if r[3] > 1e-4]        # (k_proj.bias gradients are analytically zero: rounding noise only)
        print("f32 big vs micro-batches:", total, sizeable[:3])
        assert total < 1e-5 and len(sizeable) > 100 and sizeable[0][0] < 2e-4, sizeable[:6]
    else:
        # (D = 128, K = 128 reductions: fewer terms average the roundings out -- measured 3.6 % whole buffer, 8 % pos_embed)
        _assert_independent(total, rows, 0.16, "tiny bf16:", total_tol=7e-2)


def test_vitl_mixed_prompt_train_step(golden_dir):
    """Config 4's prompt mix on its own model: ``VPUTrainStep`` with prompt types drawn from {click, box, scribble} and 1-3
    click iterations (trainer.py:339-454), ViT-L, B = 8, bf16, with the fused optimizer: three steps, every prompt type
    drawn at least once, losses finite and decreasing parameters move."""
    from pvpuformer_amd.isegm.engine.trainer import VPUTrainStep
    from pvpuformer_amd.optim import FusedAdam
    fx, cfg, model, batch, img4 = _model(golden_dir, "vitl.npz")
    model.set_compute_dtype("bf16")
    model.train()
    B = 8
    data = vo.synth_batch(B, cfg["img"], seed=300)
    opt = FusedAdam(model, lr=1e-5, betas=(0.9, 0.999), eps=1e-8)
    step = VPUTrainStep(model, opt, None, max_num_next_clicks=3, iterloss_weights=(1, 2, 3), prompt_types=(0, 1, 2))
    rng, np_rng = random.Random(11), np.random.RandomState(12)
    before = model.backbone.blocks[23].mlp.fc2.weight.detach().clone()
    seen = set()
    for s in range(3):
        rec = []
        logged, pts = step.batch_forward(step.upload({k: data[k] for k in ("images", "instances", "points")}, "cuda"),
                                         rng=rng, np_rng=np_rng, record=rec)
        seen |= {r["ptype"] for r in rec}
        tot = [float(v) for k, v in logged.items() if k.startswith("total_")]
        assert len(tot) == logged["num_iters"] and all(np.isfinite(t) for t in tot), logged
        assert tuple(pts.shape) == (B, 48, 3)
    torch.cuda.synchronize()
    assert opt.step_count == 3 and not torch.equal(before, model.backbone.blocks[23].mlp.fc2.weight.detach())
    assert 2 in seen and len(seen) >= 2, seen
    model.weights_frozen = False
    _cache.clear()
