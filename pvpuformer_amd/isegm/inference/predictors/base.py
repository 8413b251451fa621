"""NoBRS predictor, API-compatible with the hot calls of isegm/inference/predictors/base.py:10-223:
``set_input_image / get_prediction / get_vqu_prediction / get_points_nd / get_states / set_states``.  Supported
transforms: sigmoid and horizontal-flip test-time augmentation (flip.py:8-37).  ZoomIn (zoom_in.py) is the next row
(SURVEY.md section 8f): without it the model runs on the full image, which must have the constructed size."""
import numpy as np
import torch

from ...engine.prompt_sim import get_next_promts


class BasePredictor:
    def __init__(self, model, device, net_clicks_limit=None, with_flip=False, with_sigmoid=True, zoom_in=None,
                 max_size=None, **kwargs):
        if zoom_in is not None or max_size is not None:
            raise NotImplementedError("ZoomIn / LimitLongestSide transforms are not built yet")
        self.net, self.device = model, device
        self.net_clicks_limit, self.with_flip, self.with_sigmoid = net_clicks_limit, with_flip, with_sigmoid
        self.original_image = None
        self.prev_prediction = None
        self.zoom_in = None

    def set_input_image(self, image):
        """image: HxWx3 uint8 / float numpy array (torchvision ToTensor semantics) or a [3,H,W] / [1,3,H,W] tensor."""
        if isinstance(image, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1).float()
            image = t / 255.0 if image.dtype == np.uint8 else t
        self.original_image = image.to(self.device)
        if self.original_image.dim() == 3:
            self.original_image = self.original_image.unsqueeze(0)
        self.prev_prediction = torch.zeros_like(self.original_image[:, :1])

    def get_points_nd(self, clicks_lists):
        """base.py:195-213: positives first, then negatives, each padded with (-1,-1,-1) to the common count."""
        num_pos = [sum(c.is_positive for c in cl) for cl in clicks_lists]
        num_neg = [len(cl) - p for cl, p in zip(clicks_lists, num_pos)]
        n = max(num_pos + num_neg)
        if self.net_clicks_limit is not None:
            n = min(self.net_clicks_limit, n)
        n = max(1, n)
        total = []
        for cl in clicks_lists:
            cl = cl[:self.net_clicks_limit]
            pos = [c.coords_and_indx for c in cl if c.is_positive]
            neg = [c.coords_and_indx for c in cl if not c.is_positive]
            total.append(pos + (n - len(pos)) * [(-1, -1, -1)] + neg + (n - len(neg)) * [(-1, -1, -1)])
        return torch.tensor(total, device=self.device)

    def _flip_inputs(self, image_nd, clicks_lists):
        """AddHorizontalFlip.transform (flip.py:9-22): batch of [image, flipped image]; clicks mirrored in x."""
        w = image_nd.shape[3]
        image_nd = torch.cat([image_nd, torch.flip(image_nd, dims=[3])], dim=0)
        flipped = [[c.copy(coords=(c.coords[0], w - c.coords[1] - 1)) for c in cl] for cl in clicks_lists]
        return image_nd, clicks_lists + flipped

    def _net_input(self, clicker, prev_mask):
        clicks_list = clicker.get_clicks()
        prev_mask = self.prev_prediction if prev_mask is None else prev_mask
        image_nd = torch.cat((self.original_image, prev_mask), dim=1) if self.net.with_prev_mask else self.original_image
        clicks_lists = [clicks_list]
        if self.with_flip:
            image_nd, clicks_lists = self._flip_inputs(image_nd, clicks_lists)
        return image_nd, clicks_lists, prev_mask

    def _finish(self, logits):
        pred = torch.sigmoid(logits) if self.with_sigmoid else logits
        if self.with_flip:   # AddHorizontalFlip.inv_transform (flip.py:24-31)
            half = pred.shape[0] // 2
            pred = 0.5 * (pred[:half] + torch.flip(pred[half:], dims=[3]))
        self.prev_prediction = pred
        return pred

    @torch.no_grad()
    def get_prediction(self, clicker, prev_mask=None):
        image_nd, clicks_lists, _ = self._net_input(clicker, prev_mask)
        logits = self.net(image_nd, self.get_points_nd(clicks_lists).float())['instances']   # base.py:102-104
        return self._finish(logits).cpu().numpy()[0, 0]

    @torch.no_grad()
    def get_vqu_prediction(self, clicker, prev_mask=None, on_cascade=False, gt_mask=None, as_prompt_type=0,
                           click_indx=0, as_multi_prompts=True):
        """base.py:106-151,166-177: the model also receives the box prompt derived from (prev_mask, gt)."""
        image_nd, clicks_lists, prev = self._net_input(clicker, prev_mask)
        points_nd = self.get_points_nd(clicks_lists).float()
        gt = torch.from_numpy(np.asarray(gt_mask, dtype=np.float32))[None, None].to(self.device)
        if self.with_flip:
            gt = torch.cat([gt, torch.flip(gt, dims=[3])], dim=0)
            prev = torch.cat([prev, torch.flip(prev, dims=[3])], dim=0)
        _, boxes = get_next_promts(prev, gt, points_nd, None, as_allmask=False, jitter_box=False)
        prompts = (points_nd, boxes, None)
        logits = self.net(image_nd, points_nd, prompts, as_prompt_type)['instances']
        return self._finish(logits).cpu().numpy()[0, 0], prompts

    def get_states(self):
        return {'transform_states': [], 'prev_prediction': self.prev_prediction.clone()}

    def set_states(self, states):
        self.prev_prediction = states['prev_prediction']
