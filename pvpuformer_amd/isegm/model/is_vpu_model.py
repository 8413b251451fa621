"""``VitMultiGaussianVector_ed_Model`` -- drop-in for the reference class of the same name and module path
(isegm/model/is_vpu_model.py:140-449): same constructor keywords, ``_config`` capture, state-dict keys (349 for ViT-B),
positional call signature ``model(image, points, prompts, as_prompt_type, edloss, pclout)`` and output dict
``{'instances', 'instances_aux'}``.

All arithmetic runs in the HIP kernels of ``libvpu_hip.so`` through ``pvpuformer_amd.engine.Engine``; the module tree
below only holds parameters.  There is no CPU / eager fallback: calling the model without a GPU (or without the built
extension) raises.
"""
import os

import torch
import torch.nn as nn

from pvpuformer_amd import ops
from ..utils.serialization import serialize
from .is_model import ISModel
from .modeling.models_vit import PatchEmbed, VisionTransformer
from .param_table import Container, head_shapes, init_like_reference, neck_shapes, register_tree


class SimpleFPN(Container):
    """Parameters of the DMA neck (is_vpu_model.py:18-91)."""

    def __init__(self, in_dim=768, out_dims=[128, 256, 512, 1024], img_size=(448, 448), decoder_type=''):
        super().__init__()
        self.d_model, self.hide_dim, self.img_size, self.out_dims = in_dim, 1024, tuple(img_size), list(out_dims)
        register_tree(self, neck_shapes(in_dim, list(out_dims), tuple(img_size)))


class SwinTransfomerSegHead(Container):
    """Parameters of the segmentation head (swin_transformer.py:654-721, decode_head.py:47-86).  ``d_model`` follows the
    backbone width (the reference hard-codes 768, swin_transformer.py:668, which only fits ViT-B)."""

    def __init__(self, in_channels, channels, num_classes, in_index=(0, 1, 2, 3), dropout_ratio=0.1, loss_decode=None,
                 align_corners=False, upsample='x1', ed_loss=True, interpolate_mode='bilinear', d_model=768, **kwargs):
        super().__init__()
        if upsample != 'x1' or not ed_loss or align_corners or num_classes != 1 or list(in_index) != [0, 1, 2, 3]:
            raise NotImplementedError("only the VPU configuration (upsample='x1', ed_loss, 4 inputs) is on the hot path")
        self.in_channels, self.channels, self.num_classes = list(in_channels), channels, num_classes
        self.dropout_ratio, self.align_corners, self.unsample = dropout_ratio, align_corners, upsample
        self.loss_decode = loss_decode  # kept only so that checkpoints' pickled configs round-trip
        self.d_model = d_model
        register_tree(self, head_shapes(list(in_channels), channels, num_classes, d_model))


class PositionEmbeddingRandom(Container):
    """Buffer only (is_vpu_model.py:453-465); unused by forward, kept for state-dict compatibility."""

    def __init__(self, num_pos_feats=64, scale=None):
        super().__init__()
        self.register_buffer("positional_encoding_gaussian_matrix", torch.zeros(2, num_pos_feats))


class _VPUFunction(torch.autograd.Function):
    """Bridges the engine's tape into torch autograd so that ``loss.backward()`` of an unmodified trainer works.
    Parameter gradients are accumulated directly into ``param.grad`` (views of the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, anchor, model, image, points, boxes, prompt_type, drop_mask, scribble):
        ctx.engine = model._engine
        inst, aux = model._engine.forward(image, points, boxes, prompt_type, drop_mask, training=True, scribble=scribble)
        ctx.tape = model._engine.last_tape      # THIS call's tape: several forwards may be pending before one backward
        return inst, aux

    @staticmethod
    def backward(ctx, d_inst, d_aux):
        ctx.engine.backward(d_inst, d_aux, tape=ctx.tape)
        ctx.tape = None
        return (None,) * 8


class VitMultiGaussianVector_ed_Model(ISModel):
    @serialize
    def __init__(self, num_max_points=24, backbone_params={}, neck_params={}, head_params={}, random_split=False,
                 residual=False, **kwargs):
        super().__init__(**kwargs)
        if random_split:
            raise NotImplementedError("random_split (models_vit.py:193-222) is off in every VPU configuration")
        self.random_split, self.residual = random_split, residual
        self.num_max_points = num_max_points
        self.image_size = tuple(backbone_params['img_size'])
        self.vit_patch_size = tuple(backbone_params['patch_size'])
        self.embed_dim = backbone_params['embed_dim']
        self.out_chans = 256
        self.patch_embed_coords = PatchEmbed(img_size=self.image_size, patch_size=self.vit_patch_size,
                                             in_chans=3 if self.with_prev_mask else 2, embed_dim=self.embed_dim)
        self.backbone = VisionTransformer(**backbone_params)
        self.neck = SimpleFPN(**neck_params)
        hp = dict(head_params)
        hp.setdefault('d_model', self.embed_dim)
        self.head = SwinTransfomerSegHead(**hp)
        self.pe_layer = PositionEmbeddingRandom(self.embed_dim // 2)
        self.num_point_embeddings = 4
        self.point_embeddings = Container()
        for i in range(4):
            m = Container()
            m.weight = nn.Parameter(torch.zeros(1, self.embed_dim))
            self.point_embeddings.add_module(str(i), m)
        self.not_a_point_embed = Container()
        self.not_a_point_embed.weight = nn.Parameter(torch.zeros(1, self.embed_dim))
        if self.with_aux_output:
            self.head_aux = Container()
            self.head_aux.weight = nn.Parameter(torch.zeros(1, 128, 1, 1))
            self.head_aux.bias = nn.Parameter(torch.zeros(1))
        if not self.with_prev_mask:
            raise NotImplementedError("the VPU configuration feeds the previous mask (with_prev_mask=True)")
        init_like_reference(self)
        self._engine = None
        self._compute_dtype = os.environ.get("VPU_COMPUTE_DTYPE", "bf16")
        self._anchor = None
        self.weights_frozen = False
        self.graph_inference = os.environ.get("VPU_INFER_GRAPH", "0") == "1"   # see _graph_forward
        self._graphs = {}

    # ---------------------------------------------------------------------------------------------- engine plumbing
    def engine_cfg(self):
        b = self.backbone
        return dict(embed_dim=self.embed_dim, depth=b.depth, num_heads=b.num_heads, img=self.image_size[0],
                    patch=self.vit_patch_size[0], mlp_ratio=int(b.mlp_ratio), out_dims=tuple(self.neck.out_dims),
                    head_channels=self.head.channels, num_max_points=self.num_max_points,
                    head_d_model=self.head.d_model, norm_radius=self.norm_radius)

    def set_compute_dtype(self, dtype):
        """'bf16' (MFMA, default) or 'f32' (exact-fp32 parity mode)."""
        assert dtype in ("bf16", "f32")
        if dtype != self._compute_dtype:
            self._compute_dtype = dtype
            self._engine = None
        return self

    def _ensure_engine(self):
        from pvpuformer_amd.engine import Engine
        first = next(self.parameters())
        if not first.is_cuda:
            raise RuntimeError("VitMultiGaussianVector_ed_Model runs on an MI355X only (move it with .cuda()); "
                               "there is no CPU path in this package")
        eng = self._engine
        stale = eng is None or eng.flat.device != first.device or \
            first.data_ptr() != eng.flat.data_ptr() + 4 * eng.names[next(iter(eng.names))][0]
        if stale:
            eng = Engine(self.engine_cfg(), self._compute_dtype, device=first.device)
            eng.bind(dict(self.named_parameters()))
            self._engine = eng
            self._anchor = torch.zeros((), device=first.device, requires_grad=True)
        return eng

    def sync_weights(self):
        """Re-derives the compute-dtype weight copies from the fp32 parameters (after an external optimizer step or a
        state-dict load).  Called automatically at each forward unless ``weights_frozen`` is set."""
        self._ensure_engine().refresh_weights()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.shadow_valid = False
        return out

    def zero_grad(self, set_to_none=False):
        # gradients live in the engine's flat buffer: zero it in one pass instead of per-parameter
        if self._engine is not None:
            self._engine.zero_grad()
        else:
            super().zero_grad(set_to_none=False)

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, image, points=None, prompts=None, as_prompt_type=0, edloss=True, pclout=False):
        """image [B,4,H,W] (rgb in [0,1] + previous mask), points [B,2n,3] (row, col, order; -1 pad),
        prompts = (points, boxes[B,5] int32, scribbles) for as_prompt_type 1 and 2; scribbles = [points array [B,1,P,2] of
        (x, y), bounding rectangles [B,1,4] of (x_center, y_center, width, height)] as ``cal_scribble`` returns them
        (numpy, not tensors: trainer.py:1192-1243) (is_vpu_model.py:383-438)."""
        eng = self._ensure_engine()
        points, boxes, scribble = self._unpack_prompts(points, prompts, as_prompt_type, edloss)
        image = image.contiguous().float()
        if not self.weights_frozen or not eng.shadow_valid:
            eng.refresh_weights()
        drop_mask = None
        if self.training and self.head.dropout_ratio > 0:
            keep = 1.0 - self.head.dropout_ratio
            drop_mask = ops.dropout_mask(image.shape[0], self.head.channels, keep, image.device)
        if torch.is_grad_enabled():
            inst, aux = _VPUFunction.apply(self._anchor, self, image, points, boxes, as_prompt_type, drop_mask, scribble)
        elif self.graph_inference and self.weights_frozen and drop_mask is None and image.is_cuda and scribble is None:
            inst, aux = self._graph_forward(eng, image, points, boxes, as_prompt_type)
        else:
            inst, aux = eng.forward(image, points, boxes, as_prompt_type, drop_mask, training=False, scribble=scribble)
        return {'instances': inst, 'instances_aux': aux if self.with_aux_output else None}

    def _graph_forward(self, eng, image, points, boxes, ptype):
        """No-grad forward replayed from a captured hipGraph (``graph_inference``; env VPU_INFER_GRAPH=1 turns it on at
        construction): the ~230 launches of a batch-2 NoBRS forward are launch-bound when enqueued one by one.  One graph
        per (image shape, prompt rows, prompt type, dtype); inputs are copied into the graph's static buffers, the
        returned tensors are the graph's output buffers (valid until the next call with the same key, which is how the
        predictor uses them).  Needs ``weights_frozen`` (the compute-dtype operands are not rebuilt per call)."""
        points = points.to(image.device).float().contiguous()
        key = (tuple(image.shape), tuple(points.shape), int(ptype), eng.dt)
        ent = self._graphs.get(key)
        if ent is None:
            s_img, s_pts = image.clone(), points.clone()
            s_box = None if boxes is None else boxes.to(device=image.device, dtype=torch.int32).contiguous().clone()
            eng.forward(s_img, s_pts, s_box, ptype, None, training=False)          # eager once: lazily created state
            torch.cuda.synchronize()
            from pvpuformer_amd.graphs import capture
            g = torch.cuda.CUDAGraph()
            with capture(g, device=image.device):
                inst, aux = eng.forward(s_img, s_pts, s_box, ptype, None, training=False)
            ent = self._graphs[key] = (g, s_img, s_pts, s_box, inst, aux)
        g, s_img, s_pts, s_box, inst, aux = ent
        s_img.copy_(image)
        s_pts.copy_(points)
        if s_box is not None:
            s_box.copy_(boxes.to(device=image.device, dtype=torch.int32))
        g.replay()
        return inst, aux

    def _unpack_prompts(self, points, prompts, as_prompt_type, edloss):
        """(points, boxes, scribble operands of the engine) from the reference's ``prompts`` tuple (is_vpu_model.py:401-408)."""
        if not edloss:
            raise NotImplementedError("edloss=False (plain head.forward) is not used by the VPU trainer / predictor")
        boxes, scribble = None, None
        if as_prompt_type == 1:
            points, boxes, _ = prompts
        elif as_prompt_type == 2:
            from .scribble import scribble_curves, scribble_profiles
            points, _, (scr_pts, scr_rects) = prompts
            if torch.is_tensor(scr_pts):
                scr_pts, scr_rects = scr_pts.detach().cpu().numpy(), scr_rects.detach().cpu().numpy()
            scribble = (torch.from_numpy(scribble_curves(scr_pts)),
                        torch.from_numpy(scribble_profiles(scr_pts, scr_rects, self.image_size[0])))   # draws from `random`
        elif as_prompt_type != 0:
            raise ValueError(f"as_prompt_type must be 0 (clicks), 1 (box) or 2 (scribble), got {as_prompt_type}")
        return points, boxes, scribble

    def backbone_forward(self, image, coord_features=None, points=None, prompts=None, as_prompt_type=0, edloss=True, pclout=False):
        """is_vpu_model.py:383-419: everything between ``get_coord_features_with_prompt`` and the final x4 upsample, as a public
        method: ``image`` [B,3,H,W] normalised (``prepare_input``'s), ``coord_features`` [B,3,H,W] (previous mask + the two
        click maps, exactly as given: nothing is drawn here) -> {'instances': [B,1,H/4,W/4] logits, 'instances_aux':
        [B,2n,H/4,W/4] similarities}.  The stages stay fused inside the engine: the caller's planes go into the patch
        embedding's operand through ``vpu_patch_im2col_prenorm``, the low-resolution maps come out through the engine's taps.
        Inference only (the outputs carry no autograd graph: training goes through ``forward``): where the reference's method is
        the differentiable body of its ``forward`` (is_vpu_model.py:383-419), a caller that tries to train through this one gets
        a RuntimeError here instead of silent zero gradients.  ``pclout`` is accepted for signature parity only."""
        if self.training and torch.is_grad_enabled():
            raise RuntimeError("backbone_forward is inference-only in this build (no autograd graph behind its outputs): call it under "
                               "model.eval() / torch.no_grad(), and train through forward()")
        eng = self._ensure_engine()
        points, boxes, scribble = self._unpack_prompts(points, prompts, as_prompt_type, edloss)
        if coord_features is None or coord_features.shape[1] != 3:
            raise ValueError("backbone_forward: coord_features must be [B, 3, H, W] (previous mask + two click maps)")
        coord = coord_features.to(device=image.device, dtype=torch.float32)
        image4 = torch.cat([image.float(), coord[:, :1]], dim=1).contiguous()
        if not self.weights_frozen or not eng.shadow_valid:
            eng.refresh_weights()
        taps = {}
        with torch.no_grad():
            eng.forward(image4, points, boxes, as_prompt_type, None, training=False, taps=taps, materialize_aux=False,
                        scribble=scribble, coord_override=coord[:, 1:3], prenorm=True)
        return {'instances': taps["seg_lowres"].float(), 'instances_aux': taps["sim_lowres"].float() if self.with_aux_output else None}
