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
        """Kept for API parity (is_model.py:59-66); on the product path normalisation is fused into the patch im2col."""
        prev_mask = None
        if self.with_prev_mask:
            prev_mask = image[:, 3:, :, :]
            image = image[:, :3, :, :]
        return image, prev_mask
