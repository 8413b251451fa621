"""NoBRS predictor, API-compatible with the hot calls of isegm/inference/predictors/base.py:10-223:
``set_input_image / get_prediction / get_vqu_prediction / get_points_nd / apply_transforms / get_states / set_states``.
Transform pipeline as in the reference (base.py:42-48): [ZoomIn] [LimitLongestSide] [SigmoidForPred] [AddHorizontalFlip],
inverted in reverse order (so the flip average is taken on logits, before the sigmoid)."""
import numpy as np
import torch

from ...engine.prompt_sim import get_next_promts
from ..transforms import AddHorizontalFlip, LimitLongestSide, SigmoidForPred, crop_resize, resize_align_corners


class BasePredictor:
    def __init__(self, model, device, net_clicks_limit=None, with_flip=False, with_sigmoid=True, zoom_in=None,
                 max_size=None, cascade_step=0, cascade_adaptive=False, cascade_clicks=1, **kwargs):
        # base.py:109-125 -- the cascade re-prediction of the first clicks and the per-click model list (`model` given as a
        # (net, click_models) tuple) are side branches the shipped evaluation never takes (scripts/evaluate_vpumodel.py:187-192
        # passes neither): not mirrored, and refused here rather than silently ignored
        if cascade_step:
            raise NotImplementedError("BasePredictor(cascade_step > 0): the cascade branch of base.py:109-119 is not mirrored")
        if isinstance(model, tuple):
            raise NotImplementedError("BasePredictor(model=(net, click_models)): the per-click model list of base.py:121-125 is not mirrored")
        self.cascade_step, self.cascade_adaptive, self.cascade_clicks = 0, bool(cascade_adaptive), cascade_clicks
        self.click_models, self.model_indx = None, 0
        self.net, self.device = model, device
        self.net_clicks_limit, self.with_flip, self.with_sigmoid = net_clicks_limit, with_flip, with_sigmoid
        self.original_image = None
        self.prev_prediction = None
        self.always_simulate_prompts = bool(kwargs.get('always_simulate_prompts', False))
        self.zoom_in = zoom_in
        self.transforms = [zoom_in] if zoom_in is not None else []
        if max_size is not None:
            self.transforms.append(LimitLongestSide(max_size=max_size))
        if with_sigmoid:
            self.transforms.append(SigmoidForPred())
        if with_flip:
            self.transforms.append(AddHorizontalFlip())

    def set_input_image(self, image):
        """image: HxWx3 uint8 / float numpy array (torchvision ToTensor semantics) or a [3,H,W] / [1,3,H,W] tensor."""
        if isinstance(image, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1).float()
            image = t / 255.0 if image.dtype == np.uint8 else t
        for t in self.transforms:
            t.reset()
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

    def apply_transforms(self, image_nd, clicks_lists):
        is_image_changed = False
        for t in self.transforms:
            image_nd, clicks_lists = t.transform(image_nd, clicks_lists)
            is_image_changed |= t.image_changed
        return image_nd, clicks_lists, is_image_changed

    def _net_input(self, clicker, prev_mask):
        clicks_list = clicker.get_clicks()
        prev_mask = self.prev_prediction if prev_mask is None else prev_mask
        image_nd = torch.cat((self.original_image, prev_mask), dim=1) if self.net.with_prev_mask else self.original_image
        image_nd, clicks_lists, _ = self.apply_transforms(image_nd, [clicks_list])
        return image_nd, clicks_lists, prev_mask

    def _finish(self, logits, image_nd):
        pred = logits
        if tuple(pred.shape[2:]) != tuple(image_nd.shape[2:]):   # base.py:93-94 (a no-op at the model's own size)
            pred = resize_align_corners(pred, image_nd.shape[2:])
        for t in reversed(self.transforms):
            pred = t.inv_transform(pred)
        return pred

    @torch.no_grad()
    def get_prediction(self, clicker, prev_mask=None):
        image_nd, clicks_lists, _ = self._net_input(clicker, prev_mask)
        logits = self.net(image_nd, self.get_points_nd(clicks_lists).float())['instances']   # base.py:102-104
        pred = self._finish(logits, image_nd)
        if self.zoom_in is not None and self.zoom_in.check_possible_recalculation():
            return self.get_prediction(clicker)
        self.prev_prediction = pred
        return pred.cpu().numpy()[0, 0]

    @torch.no_grad()
    def get_vqu_prediction(self, clicker, prev_mask=None, on_cascade=False, gt_mask=None, as_prompt_type=0,
                           click_indx=0, as_multi_prompts=True):
        """base.py:106-151,166-177: the model also receives the box prompt derived from (prev_mask, gt), both cropped to
        the ZoomIn region of interest."""
        if not as_multi_prompts:
            raise NotImplementedError("get_vqu_prediction(as_multi_prompts=False): the get_next_promts_inference branch of "
                                      "base.py:154-164 is not mirrored (the shipped evaluation passes True, vpu_evaluation.py:43)")
        image_nd, clicks_lists, prev = self._net_input(clicker, prev_mask)
        points_nd = self.get_points_nd(clicks_lists).float()
        boxes = None
        if as_prompt_type != 0 or self.always_simulate_prompts:
            # the reference derives the box prompt on EVERY click (base.py:176), also for click prompts, where the network
            # never reads it: here it is simulated only when it is consumed (~3 ms of a 7.5-ms click otherwise) -- by the
            # network (as_prompt_type != 0) or by a caller that reads ``prompts[1]``: ``always_simulate_prompts=True`` (a
            # get_predictor keyword / attribute; evaluate_sample sets it while a visualisation callback is attached)
            gt = torch.from_numpy(np.asarray(gt_mask, dtype=np.float32))[None, None].to(self.device)
            if self.with_flip:
                gt = torch.cat([gt, torch.flip(gt, dims=[3])], dim=0)
                prev = torch.cat([prev, torch.flip(prev, dims=[3])], dim=0)
            if self.zoom_in is not None and self.zoom_in._object_roi is not None:
                gt = crop_resize(gt, self.zoom_in._object_roi, self.zoom_in.target_size)
                prev = crop_resize(prev, self.zoom_in._object_roi, self.zoom_in.target_size)
            _, boxes = get_next_promts(prev, gt, points_nd, None, as_allmask=False, jitter_box=False)
        prompts = (points_nd, boxes, None)
        logits = self.net(image_nd, points_nd, prompts, as_prompt_type)['instances']
        pred = self._finish(logits, image_nd)
        if self.zoom_in is not None and self.zoom_in.check_possible_recalculation():
            return self.get_prediction(clicker), prompts
        self.prev_prediction = pred
        return pred.cpu().numpy()[0, 0], prompts

    def get_states(self):
        return {'transform_states': [t.get_state() for t in self.transforms],
                'prev_prediction': self.prev_prediction.clone()}

    def set_states(self, states):
        assert len(states['transform_states']) == len(self.transforms)
        for st, t in zip(states['transform_states'], self.transforms):
            t.set_state(st)
        self.prev_prediction = states['prev_prediction']
