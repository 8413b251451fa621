"""ISModel base, API-compatible with isegm/model/is_model.py:9-146 for the VPU path (constructor arguments,
``with_prev_mask`` / ``with_aux_output`` attributes, ``prepare_input``).  RITM-only options are rejected."""
import torch
import torch.nn as nn


class ISModel(nn.Module):
    def __init__(self, with_aux_output=False, norm_radius=5, use_disks=False, cpu_dist_maps=False, use_rgb_conv=False,
                 use_leaky_relu=False, with_prev_mask=False, norm_mean_std=([.485, .456, .406], [.229, .224, .225])):
        super().__init__()
        if use_rgb_conv or cpu_dist_maps:
            raise NotImplementedError("RITM-only options (use_rgb_conv / cpu_dist_maps) are outside the VPU hot path")
        if not use_disks:
            raise NotImplementedError("the VPU configuration uses disk maps (use_disks=True)")
        ms = ([float(x) for x in norm_mean_std[0]], [float(x) for x in norm_mean_std[1]])
        if ms != ([.485, .456, .406], [.229, .224, .225]):
            raise NotImplementedError("normalisation constants are compiled into the patch im2col kernel")
        self.with_aux_output = with_aux_output
        self.with_prev_mask = with_prev_mask
        self.with_points = False
        self.norm_radius = norm_radius
        self.coord_feature_ch = 3 if with_prev_mask else 2
        self.maps_transform = nn.Identity()

    def prepare_input(self, image):
        """is_model.py:59-66: (normalised rgb, previous mask) -- ``(x - mean) / std`` as BatchImageNormalize does it
        (ops.py:398-407: a clone, sub_, div_).  The model's own ``forward`` does not come through here (the normalisation is
        fused into the patch im2col); this serves callers of the public pieces: ``backbone_forward`` takes what it returns.
        NOTE: ``forward()`` expects the UN-normalised [B,4,H,W] input, as the reference's does -- do not chain
        ``prepare_input`` -> ``forward`` (the image would be normalised twice)."""
        prev_mask = None
        if self.with_prev_mask:
            prev_mask = image[:, 3:, :, :]
            image = image[:, :3, :, :]
        mean = torch.as_tensor([.485, .456, .406], dtype=torch.float, device=image.device)[None, :, None, None]
        std = torch.as_tensor([.229, .224, .225], dtype=torch.float, device=image.device)[None, :, None, None]
        image = image.clone().sub_(mean).div_(std)
        return image, prev_mask

    # ---- the coordinate-feature builders of is_model.py:71-146 as public methods (the model's own forward builds the same maps
    # inside the engine; these serve callers that use the pieces on their own).  Device tensors in, device tensors out: the
    # HIP kernels of csrc/prompt.hip, no host round trip.
    @staticmethod
    def _points_f32(points, device):
        return points.to(device=device, dtype=torch.float32).contiguous()

    def get_coord_features(self, image, prev_mask, points):
        """is_model.py:71-76: the two disk maps (positive / negative clicks, radius ``norm_radius``) [+ prev_mask in front]."""
        from pvpuformer_amd import ops
        if not image.is_cuda:
            raise RuntimeError("ISModel.get_coord_features: the HIP path needs CUDA tensors (there is no CPU path)")
        B, H, W = image.shape[0], image.shape[2], image.shape[3]
        pts = self._points_f32(points, image.device)
        maps = torch.empty(B, 2, H, W, device=image.device, dtype=torch.float32)
        ops.disk_maps(pts, None, maps, B, pts.shape[1] // 2, H, W, float(self.norm_radius))
        return maps if prev_mask is None else torch.cat((prev_mask.to(maps.dtype), maps), dim=1)

    def get_coord_features_with_prompt(self, image, prev_mask, points, prompts=None, as_prompt_type=0, gt_mask=None):
        """is_model.py:78-96: disk maps, then per sample the box outline (prompt type 1) or the scribble poly-line (type 2)."""
        coord = self.get_coord_features(image, None, points)
        if as_prompt_type != 0:
            _, boxes, scribbles = prompts
            scribble, rects = scribbles if scribbles is not None else (None, None)
            for b in range(coord.shape[0]):
                if as_prompt_type == 1:
                    coord[b] = self.draw_box(coord[b], boxes[b], points)
                elif as_prompt_type == 2:
                    coord[b] = self.draw_scribble(coord[b], scribble[b], rects[b])
        return coord if prev_mask is None else torch.cat((prev_mask.to(coord.dtype), coord), dim=1)

    def draw_box(self, image_, bounding_rectangle_, points, gt_mask=None):
        """is_model.py:98-121: the 3-pixel rectangle outline of (x_center, y_center, width, height, slot) OR-ed into the
        positive (slot < n) or negative channel of one sample's [2, H, W] map."""
        from pvpuformer_amd import ops
        n = points.shape[1] // 2
        H, W = image_.shape[-2:]
        box = torch.as_tensor(bounding_rectangle_).to(device=image_.device, dtype=torch.int32).reshape(1, 5).contiguous()
        none = torch.full((1, 2 * n, 3), -1.0, device=image_.device)
        tmp = torch.empty(1, 2, H, W, device=image_.device, dtype=torch.float32)
        ops.disk_maps(none, box, tmp, 1, n, H, W, float(self.norm_radius))       # no clicks: just the outline, in its channel
        image_.copy_(torch.maximum(image_.to(torch.float32), tmp[0]).to(image_.dtype))
        return image_

    def draw_scribble(self, image_, scribble_, bounding_rectangle_, gt_mask=None):
        """is_model.py:123-146: the open poly-line through scribble_[0] ((x, y) vertices) into the positive channel."""
        import numpy as np
        from pvpuformer_amd import ops
        H, W = image_.shape[-2:]
        pts = scribble_[0].detach().cpu().numpy() if torch.is_tensor(scribble_) else np.asarray(scribble_[0])
        curve = torch.from_numpy(np.ascontiguousarray(pts[:, :2].astype(np.int32))).to(image_.device).reshape(1, -1, 2)
        tmp = image_.to(torch.float32).reshape(1, 2, H, W).contiguous()
        ops.draw_polyline(curve, tmp, 1, curve.shape[1], H, W)
        image_.copy_(tmp[0].to(image_.dtype))
        return image_
