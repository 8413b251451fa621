"""CPU: the nn.Module boundary mirrors the reference's (state-dict keys / shapes / order, _config capture, checkpoint
round trip, loud failure without a GPU)."""
import io

import numpy as np
import pytest
import torch

import vpu_oracle as vo


def make_model(cfg=None, **extra):
    from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model
    cfg = cfg or vo.make_cfg()
    bp = dict(img_size=(cfg["img"],) * 2, patch_size=(cfg["patch"],) * 2, in_chans=3, embed_dim=cfg["embed_dim"],
              depth=cfg["depth"], num_heads=cfg["num_heads"], mlp_ratio=cfg["mlp_ratio"], qkv_bias=True)
    npar = dict(in_dim=cfg["embed_dim"], out_dims=list(cfg["out_dims"]), img_size=(cfg["img"],) * 2)
    hp = dict(in_channels=list(cfg["out_dims"]), in_index=[0, 1, 2, 3], dropout_ratio=0.1, num_classes=1,
              loss_decode=None, align_corners=False, upsample='x1', ed_loss=True, channels=cfg["head_channels"])
    return VitMultiGaussianVector_ed_Model(use_disks=True, norm_radius=5, with_prev_mask=True, backbone_params=bp,
                                           neck_params=npar, head_params=hp, random_split=False, residual=True,
                                           with_aux_output=True, **extra)


TINY = dict(embed_dim=128, depth=8, num_heads=4, out_dims=(16, 32, 64, 128), head_channels=32)


@pytest.mark.parametrize("cfg_kw", [{}, TINY, dict(embed_dim=1024, depth=24, num_heads=16)])
def test_state_dict_matches_reference_table(cfg_kw):
    cfg = vo.make_cfg(**cfg_kw)
    m = make_model(cfg)
    shapes = vo.param_shapes(cfg)  # verified against the reference's own state_dict by oracle/make_golden.py
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    if not cfg_kw:
        assert len(sd) == 349
        assert sum(p.numel() for p in m.parameters()) == 123122987


def test_config_capture_and_checkpoint_roundtrip():
    from pvpuformer_amd.isegm.utils.serialization import load_model
    cfg = vo.make_cfg(**TINY)
    m = make_model(cfg)
    c = m._config
    assert c["class"] == "isegm.model.is_vpu_model.VitMultiGaussianVector_ed_Model"
    assert c["params"]["num_max_points"]["value"] == 24 and not c["params"]["num_max_points"]["specified"]
    assert c["params"]["use_disks"]["specified"] and c["params"]["backbone_params"]["value"]["embed_dim"] == 128
    buf = io.BytesIO()
    torch.save({"state_dict": m.state_dict(), "config": m._config}, buf)  # isegm/utils/misc.py:32-33 format
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    import pvpuformer_amd
    pvpuformer_amd.install()
    m2 = load_model(ck["config"])
    m2.load_state_dict(ck["state_dict"], strict=True)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_init_distributions():
    torch.manual_seed(0)
    m = make_model(vo.make_cfg(**TINY))
    sd = m.state_dict()
    assert torch.all(sd["backbone.blocks.0.norm1.weight"] == 1) and torch.all(sd["backbone.blocks.0.attn.qkv.bias"] == 0)
    w = sd["backbone.blocks.0.mlp.fc1.weight"]
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5
    assert w.abs().max() <= bound and w.abs().max() > 0.9 * bound
    assert abs(sd["backbone.pos_embed"].std().item() - 0.02) < 0.002
    assert abs(sd["head.logit_scale"].item() - np.log(1 / 0.07)) < 1e-6
    assert sd["neck.att.layers.0.norm1.weight"].eq(1).all() and sd["neck.down_4.1.bias"].eq(0).all()
    assert sd["pe_layer.positional_encoding_gaussian_matrix"].std() > 0.5


def test_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = make_model(vo.make_cfg(**TINY))
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 4, 448, 448), -torch.ones(1, 48, 3))
    assert m.backbone.no_weight_decay() == {"pos_embed", "cls_token", "dist_token"}
    assert m.backbone.patch_embed.grid_size == (28, 28) and m.with_prev_mask


def test_layerwise_lr_decay_groups_match_reference(golden_dir):
    """f3: the per-tensor (learning rate, weight decay) table of get_optimizer_with_layerwise_decay equals what the
    reference's param_groups_lrd produced for the same model (tests/golden/lrd.npz); tensors the reference leaves out of
    every group get learning-rate scale 0."""
    import os
    from pvpuformer_amd.isegm.utils import lr_decay as lrd
    fx = np.load(os.path.join(golden_dir, "lrd.npz"))
    m = make_model(vo.make_cfg(**TINY))
    groups = lrd.param_groups_lrd(m, 5e-5, weight_decay=0.02, no_weight_decay_list=m.backbone.no_weight_decay(),
                                  layer_decay=0.75)
    table = lrd.per_param_table(groups, 5e-5)
    ref = {str(n): (float(l), float(w)) for n, l, w in zip(fx["names"], fx["lr"], fx["wd"])}
    assert set(table) == set(ref)
    for n, (scale, wd) in table.items():
        assert abs(scale * 5e-5 - ref[n][0]) <= 1e-18 + 1e-12 * ref[n][0] and wd == ref[n][1], n
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in fx["all_names"]]
    assert lrd.get_layer_id_for_vit("blocks.3.attn.qkv.weight", 9) == 4 and lrd.get_layer_id_for_vit("fc_norm.weight", 9) == 9


def test_multistep_lr():
    from pvpuformer_amd.optim import MultiStepLR

    class Opt:
        lr = 5e-5
    o = Opt()
    s = MultiStepLR(o, milestones=[50, 55], gamma=0.1)
    lrs = []
    for _ in range(57):
        lrs.append(o.lr)
        s.step()
    assert lrs[0] == 5e-5 and lrs[49] == 5e-5 and abs(lrs[50] - 5e-6) < 1e-18 and abs(lrs[55] - 5e-7) < 1e-18


def test_mae_checkpoint_import_interpolates_pos_embed(tmp_path):
    """f4: backbone.init_weights_from_pretrained (models_vit.py:150-166 + pos_embed.py:75-96): an MAE-style checkpoint
    ({'model': ...}, 14x14 position grid + cls token, extra decoder keys) is loaded non-strictly, its position embedding
    bicubically re-gridded to the model's 28x28 grid with the cls token kept."""
    import torch.nn.functional as F
    cfg = vo.make_cfg(**TINY)
    m = make_model(cfg)
    D = cfg["embed_dim"]
    g = torch.Generator().manual_seed(5)
    mae = {"pos_embed": torch.randn(1, 1 + 14 * 14, D, generator=g), "cls_token": torch.randn(1, 1, D, generator=g),
           "blocks.0.attn.qkv.weight": torch.randn(3 * D, D, generator=g), "decoder_embed.weight": torch.randn(8, D, generator=g)}
    path = tmp_path / "mae.pth"
    torch.save({"model": {k: v.clone() for k, v in mae.items()}}, path)
    msg = m.backbone.init_weights_from_pretrained(str(path))
    assert "decoder_embed.weight" in msg.unexpected_keys and "blocks.1.attn.qkv.weight" in msg.missing_keys
    sd = m.backbone.state_dict()
    assert torch.equal(sd["blocks.0.attn.qkv.weight"], mae["blocks.0.attn.qkv.weight"])
    assert torch.equal(sd["cls_token"], mae["cls_token"])
    grid = mae["pos_embed"][:, 1:].reshape(1, 14, 14, D).permute(0, 3, 1, 2)
    want = F.interpolate(grid, size=(28, 28), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
    assert torch.equal(sd["pos_embed"][:, :1], mae["pos_embed"][:, :1])
    torch.testing.assert_close(sd["pos_embed"][:, 1:], want, rtol=0, atol=0)
    assert m.backbone.init_weights_from_pretrained("") is None


def test_predictor_side_branches_fail_loudly_and_callback_gets_the_box_prompt():
    """isegm/inference/predictors/base.py:109-125,154-164: the cascade re-prediction, the per-click model list and the
    ``as_multi_prompts=False`` branch are not mirrored -- they raise instead of silently running something else (INTEGRATION.md
    section 2).  For click prompts the mirror derives the box prompt only when somebody reads it: ``always_simulate_prompts``,
    which ``evaluate_sample`` switches on while a visualisation callback is attached (the reference computes it on every click,
    base.py:176, vpu_evaluation.py:84-97)."""
    from pvpuformer_amd.isegm.inference.predictors.base import BasePredictor
    from pvpuformer_amd.isegm.inference import vpu_evaluation as ve

    class Net:
        with_prev_mask = True

        def __call__(self, image, points, prompts=None, as_prompt_type=0):
            self.last_prompts = prompts
            logits = torch.zeros(image.shape[0], 1, *image.shape[2:])
            logits[:, :, 8:24, 8:24] = 4.0
            return {"instances": logits, "instances_aux": None}

    with pytest.raises(NotImplementedError, match="cascade"):
        BasePredictor(Net(), "cpu", cascade_step=2)
    with pytest.raises(NotImplementedError, match="click_models"):
        BasePredictor((Net(), [Net()]), "cpu")
    net = Net()
    pred = BasePredictor(net, "cpu", with_flip=False)
    assert pred.always_simulate_prompts is False
    image = np.zeros((32, 32, 3), np.uint8)
    gt = np.zeros((32, 32), np.int32)
    gt[8:24, 8:24] = 1
    clicker = ve.Clicker(gt_mask=gt)
    pred.set_input_image(image)
    clicker.make_next_click(np.zeros_like(gt, dtype=bool))
    with pytest.raises(NotImplementedError, match="as_multi_prompts"):
        pred.get_vqu_prediction(clicker, gt_mask=gt, as_multi_prompts=False)
    _, prompts = pred.get_vqu_prediction(clicker, gt_mask=gt)
    assert prompts[1] is None                       # click prompts, nobody reads the box: not simulated
    seen = []
    ve.evaluate_sample(image, gt, pred, max_iou_thr=2.0, max_clicks=2,
                       callback=lambda *a: seen.append(a[9]))      # (..., zoom_in, prompts, as_prompt_type)
    assert len(seen) == 2 and all(p[1] is not None and tuple(p[1].shape) == (1, 5) for p in seen)
    assert pred.always_simulate_prompts is False    # restored
