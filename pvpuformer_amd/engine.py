"""Forward / backward executor of the VPUFormer hot path on MI355X.

Composes the C-ABI kernels (``include/vpu_hip.h``) into the model of
``isegm/model/is_vpu_model.py:383-438`` (reference paths) and its hand-scheduled backward.  Design points:

* one flat fp32 parameter buffer + one flat fp32 gradient buffer (all ``nn.Parameter``s are views), a flat bf16
  shadow for the MFMA GEMMs -> the optimizer and the data-parallel all-reduce work on a few large contiguous ranges;
* backbone tokens stay in WINDOW order from the patch embedding to the last block, so ``patchify``/``unpatchify``
  (``models_vit.py:225-255``) cost nothing: window attention is a batched GEMM over contiguous rows, global
  attention is permutation-equivariant; one un-permute feeds the neck;
* every map in the neck / head is channels-last, so each (transposed) convolution is a GEMM plus at most one
  pixel-shuffle pass, and the four resized head inputs are written straight into their slice of the concat buffer;
* a tape of backward closures with explicit gradient accumulation -- no autograd graph, no per-op torch dispatch;
  torch is used for device memory and streams only.
"""
import contextlib
import math
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from ._lib import (BF16, F32, EPI_BIAS, EPI_PREACT, EPI_GELU, EPI_RELU, EPI_DGELU, EPI_DRELU, EPI_RESID, EPI_AFFINE,
                   EPI_ACCUM, EPI_OUT_F32, EPI_SAVE_DGELU, EPI_MULAUX)


class Tape:
    """The backward closures of ONE forward call plus that call's output-gradient slots.  ``Engine.forward`` returns a new
    one per call (``Engine.last_tape``), so several forwards may be pending at once -- the reference trainer sums the
    losses of up to three click iterations before a single ``loss.backward()`` (trainer.py:342-455), which reaches the
    autograd bridge once per forward."""
    __slots__ = ("fns", "out_grads", "done", "sim_low", "lane")

    def __init__(self):
        self.fns = []
        self.out_grads = [None, None, None]
        self.done = False
        self.sim_low = None
        self.lane = None          # the HIP stream the closures recorded now belong to (Engine._lane), None = the caller's

    def append(self, fn):
        s = self.lane
        if s is not None:         # a closure of the token lane runs on the token lane's stream in backward as well
            inner = fn

            def fn():
                with torch.cuda.stream(s):
                    inner()
        self.fns.append(fn)

    def __len__(self):
        return len(self.fns)


class Var:
    """An activation and (lazily) its gradient."""
    __slots__ = ("t", "g", "gn_stats")

    def __init__(self, t):
        self.t = t
        self.g = None
        self.gn_stats = None      # GroupNorm(1, C) statistics partials left by the producer (conv_t2), or None


def _rup(x, m):
    return (x + m - 1) // m * m


def click_lut():
    """19-tap PuE clip, float32 (isegm/model/ops.py:51-61)."""
    t = np.arange(0, 19, 1, np.float32)
    lut = np.exp(-((t - 9) ** 2) / (2 * 3 ** 2))
    lut[9] += 1
    return lut


def pos2d_table(d_model, height, width):
    """TwoWayTransformer.pos2d (isegm/model/modeling/transformer.py:290-318) as a constant [H*W, d_model] table,
    built once on the host instead of seven times per forward."""
    pe = torch.zeros(d_model, height, width)
    half = d_model // 2
    div = torch.exp(torch.arange(0., half, 2) * -(math.log(10000.0) / half))
    pw = torch.arange(0., width).unsqueeze(1) * div
    ph = torch.arange(0., height).unsqueeze(1) * div
    pe[0:half:2] = torch.sin(pw).t().unsqueeze(1).expand(-1, height, -1)
    pe[1:half:2] = torch.cos(pw).t().unsqueeze(1).expand(-1, height, -1)
    pe[half::2] = torch.sin(ph).t().unsqueeze(2).expand(-1, -1, width)
    pe[half + 1::2] = torch.cos(ph).t().unsqueeze(2).expand(-1, -1, width)
    return pe.reshape(d_model, height * width).t().contiguous()


class Engine:
    def __init__(self, cfg, dtype="bf16", device="cuda"):
        self.cfg = dict(cfg)
        self.dt = BF16 if dtype in ("bf16", BF16) else F32
        self.td = torch.bfloat16 if self.dt == BF16 else torch.float32
        self.dev = torch.device(device)
        c = self.cfg
        self.D, self.P, self.img = c["embed_dim"], c["patch"], c["img"]
        self.g = self.img // self.P
        self.NT = self.g * self.g
        self.wg = 224 // self.P
        self.nw = self.g // self.wg
        self.heads = c["num_heads"]
        self.depth = c["depth"]
        self.group = 6 if self.depth == 12 else self.depth // 4
        self.nmax = c["num_max_points"]
        self.E = 2 * self.img + 3
        self.Epad = _rup(self.E, 8)
        self.C = c["head_channels"]
        self.out_dims = tuple(c["out_dims"])
        self.names = None
        self.tape = Tape()
        self.last_tape = None
        self.params = []
        self.norm_radius = float(c.get("norm_radius", 5))
        self.training = False
        self.shadow_valid = False
        self._const_ready = False
        import os
        self.use_flash = True    # False: the unfused S / P path (what fp32 and other head dimensions take anyway); an attribute, no environment knob
        # weight gradients are only consumed by the optimizer: VPU_WGRAD_STREAM=1 runs them on a second HIP stream.  Off by
        # default: since the GEMM launches became persistent (every launch fills the chip) the two streams only share the
        # CUs, and the fork/join events cost host time (measured: 19.45 ms/step on one stream, 19.9 ms on two)
        self.use_side = os.environ.get("VPU_WGRAD_STREAM", "0") == "1"
        self.side = None
        self.fuse_ln_pe = True   # LayerNorm + position-embedding add in one launch (neck); attribute for the tests
        # weight-gradient GEMMs are queued and launched in groups (ops.gemm_grouped): the four of a ViT block are 432
        # full-K tiles -- one round of the persistent grid, no split-K slabs, no reduce launches --, the neck's 576-row
        # ones (9 K-tiles each) go eight to a launch.  ``group_wgrad`` = False launches each one on its own.
        self.group_wgrad = True    # (the settled A/B switches of rounds 2-5 are attributes: the tests flip them, no environment knob)
        self.ride_wgrad = True      # small long-reduction gradients ride with the big groups
        self.wgrad_fill = 0.8     # a big group is launched once its rounds of 256 tiles are this full
        # VPU_WGRAD_PACK (default 1): the queued long-reduction gradients leave in launches of ONE FULL ROUND of 256-row x
        # 128-column tiles: every tile of a reduction length costs the same (147 K-tiles at 9408 rows), so a launch takes as
        # long as one tile whatever its tile count -- a ViT-B block's four gradients are 216 tiles on 256 CUs (84 %), 12
        # such launches per step.  Packed, a launch takes whole problems in queue order and the leading row / column blocks
        # of the next one (a sub-matrix of the gradient: pointer offsets, no kernel change), the rest stays queued:
        # 2592 backbone tiles = 10.1 full launches instead of 12 (-0.3 ms per step).  Not with a reducer attached (a block's
        # range must be final at its marker).
        # Only where a block's four gradients under-fill one round (216 of 256 tiles at D = 768): at D = 1024 the fill rule's
        # groups are whole rounds already (128 + 128, 32 + 96 + 128 tiles) and cutting them costs 1.7 % (ViT-L 340 -> 334
        # images/s); at D = 1280 (450 / 750-tile groups) it is neutral.  VPU_WGRAD_PACK=0 / 1 forces it off / on.
        hid = self.D * c["mlp_ratio"]
        blk = sum(((n + 255) // 256) * ((k + 127) // 128) for n, k in ((3 * self.D, self.D), (self.D, self.D), (hid, self.D), (self.D, hid)))
        pk = os.environ.get("VPU_WGRAD_PACK", "")
        # (round 4: with the K4 launches -- 256 x 256 tiles, rounds cut exactly, small problems riding -- every geometry packs)
        # (the EFFECTIVE k3 option of the loaded library -- environment default resolved, laboratory-only bits masked in the
        # product build -- not the raw environment variable: ADVICE r5, the engine must size its rounds for the kernels the
        # library really dispatches)
        k3 = ops.gemm_get_option("k3")
        k4_default = (k3 & 8) != 0
        self.pack_wgrad = pk == "1" or (pk != "0" and (blk < 0.95 * 256 or k4_default))
        # K3 weight-gradient launches (round 4: 256 x 128 tiles in 256-thread workgroups, TWO per CU, free-running): a full
        # round is 512 tiles.  VPU_GEMM_K3 (bit 0) selects them in the library; the engine sizes its packed launches for it.
        self.k3_wgrad = (k3 & 1) != 0
        self.k4_wgrad = (k3 & 8) != 0          # 256 x 256 tiles, one workgroup per CU
        self.wgrad_round = 256 if (self.k4_wgrad or not self.k3_wgrad) else 512
        self.wgrad_tn = 256 if self.k4_wgrad else 128      # tile width of the long-reduction weight-gradient launches
        # ``unify_wgrad`` (with the packed queue): every weight gradient over a multiple of the token rows -- the
        # FPN's and the head's maps: 4 x / 16 x the 9408 rows of a ViT block -- is cut into reduction slices of exactly the
        # token rows, given to the launch as BATCH ENTRIES of one problem (one descriptor), each writing its own fp32 slab; the
        # slices are tiles of the same cost as a ViT block's, so they pack into the same launches, and one batched column
        # sum adds the slabs to the gradients.  Before: a split-K launch + a reduce launch per problem, most of them at the
        # end of backward with the chip a quarter full (0.47 ms per step).
        self.unify_wgrad = True
        self.lazy_zero = True      # zero_grad(lazy=True) honoured
        # round 5 launch fusions (the three gates in one launch per pass, the derived operands in one batched cast, the q_out
        # gradient fan-out in one launch, conv_seg's partial sums through the batched column sums): ``r5_fused`` = False runs the
        # round-4 launches instead (same-box A/B runs)
        self.r5_fused = True
        # split launches of the neck's prompt<->image attentions (cross_attention): ranges per long side; 1 = off
        self.xattn_split = max(1, int(os.environ.get("VPU_XATTN_SPLIT", "1")))
        # round 5: the DMA neck's prompt-token chain (~90 dependent launches of <= 108 workgroups each) on its own HIP stream
        # beside the image-side launches (Engine._neck_lanes); VPU_NECK_LANES=0 = everything on the caller's stream
        self.neck_lanes = int(os.environ.get("VPU_NECK_LANES", "0"))     # (2: only the part that runs beside the backbone)
        self._tok_stream = None
        self._in_lanes = False    # backward is inside the two-lane section: the weight-gradient queue only collects
        # fused bias column sums of the packed K4 launches DISTRIBUTED over a problem's column tiles (vpu_hip.h: cs_tn): with
        # the classic form the tiles of the first column block -- a third of a ViT block's -- run 17 % longer than the
        # others, and a packed launch is one round of tiles (tools/k4_drift.py).  ``dist_colsum`` = False: classic form
        self.dist_colsum = True
        self._pack_seen, self._pack_total = {}, {}     # reduction length -> tiles queued in this / the previous backward pass
        self._pending_reports, self._reporting = [], False    # gradient ranges whose marker has been passed but not reported yet
        self.group_tiles = 256     # flush_group: largest problem (output tiles) grouped (256: the 9408-row K / V projections of the neck share one launch, +0.7 % step rate)
        self.split_wgrad = True   # _wgrad_sliced for few-tile long reductions
        self._wq = []          # queued weight gradients: (gemm args, gemm kwargs, output tiles, reduction length)
        self._csq = []         # queued column sums of norm-layer gradient partials: (part, out, rows, cols)
        self._gq = []          # deferred small GEMMs of one group (an attention's q / k / v projections or their dgrads)
        self._gq_out = set()   # data_ptr of the outputs the queued GEMMs will write
        self._frozen = set()   # data_ptr of buffers a queued / side-stream GEMM still reads: no in-place writes
        self._token_rows = 0
        self.grad_ready_hook = None   # callable(lo, hi): gflat[lo:hi] is final for this backward (data-parallel reducer)
        self.report_lag = max(1, int(os.environ.get("VPU_DIST_REPORT_LAG", "2")))   # blocks a finished range may wait for its launches to fill

    # ------------------------------------------------------------------------------------------ parameters
    def bind(self, named_params):
        """Packs the parameters into one flat fp32 buffer (views keep their identity) and allocates the flat
        gradient buffer and the bf16 shadow."""
        self.names = OrderedDict()
        off = 0
        for n, p in named_params.items():
            self.names[n] = (off, tuple(p.shape), p.numel())
            off = _rup(off + p.numel(), 8)
        self.total = off
        self.flat = torch.zeros(off, device=self.dev, dtype=torch.float32)
        self.gflat = torch.zeros(off, device=self.dev, dtype=torch.float32)
        self.params = []
        for n, p in named_params.items():
            o, shape, numel = self.names[n]
            view = self.flat[o:o + numel].view(shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.gflat[o:o + numel].view(shape)
            self.params.append((p, o, shape, numel))
        self.shadow = torch.zeros(off, device=self.dev, dtype=torch.bfloat16) if self.dt == BF16 else None
        self._lazy = set()
        self._lazy_names, self._lazy_ranges = self._lazy_plan()
        D, P = self.D, self.P
        self.k3p = _rup(3 * P * P, 8)   # each half of the fused patch-embed K is padded to 16 bytes (P = 14: 588 -> 592)
        self.w_patch = torch.zeros(D, 2 * self.k3p, device=self.dev, dtype=self.td)
        self.b_patch = torch.zeros(D, device=self.dev, dtype=torch.float32)
        self.w_lin1p = torch.zeros(2048, self.Epad, device=self.dev, dtype=self.td)
        self.pos_win = torch.zeros(self.NT, D, device=self.dev, dtype=self.td)
        self._pos_cache, self._kpe_cache, self._regrid_cache = {}, {}, {}
        self.shadow_valid = False
        if not self._const_ready:
            self.kpe = pos2d_table(D, self.g, self.g).to(self.dev).to(self.td)
            self.lut = torch.from_numpy(click_lut()).to(self.dev)
            self._const_ready = True

    def W(self, name):
        o = self.names[name][0]
        return (self.shadow, o) if self.dt == BF16 else (self.flat, o)

    def Pm(self, name):
        return (self.flat, self.names[name][0])

    def G(self, name):
        return (self.gflat, self.names[name][0])

    def refresh_weights(self, shadow_is_fresh=False):
        """fp32 master -> compute-dtype operands (bf16 shadow, fused patch-embed weight, K-padded PuE weight,
        window-ordered pos_embed).  ~0.75 GB of traffic for ViT-B; ``shadow_is_fresh``: the fused optimizer has already
        written the bf16 shadow, only the four small derived operands are rebuilt."""
        D, P = self.D, self.P
        if self.dt == BF16 and not shadow_is_fresh:
            ops.cast2d(self.flat, self.total, self.shadow, self.total, 1, self.total)
        k3 = 3 * P * P
        po = self.names["backbone.pos_embed"][0]
        if not self.r5_fused:
            ops.cast2d(self.Pm("backbone.patch_embed.proj.weight"), k3, (self.w_patch, 0), 2 * self.k3p, D, k3)
            ops.cast2d(self.Pm("patch_embed_coords.proj.weight"), k3, (self.w_patch, self.k3p), 2 * self.k3p, D, k3)
            ops.add4(self.Pm("backbone.patch_embed.proj.bias"), self.Pm("patch_embed_coords.proj.bias"), None, None, self.b_patch, D)
            ops.cast2d(self.Pm("neck.ffn_layer.lin1.weight"), self.E, self.w_lin1p, self.Epad, 2048, self.E, self.Epad)
            tmp = torch.empty(self.NT, D, device=self.dev, dtype=self.td)
            ops.cast2d((self.flat, po + D), D, tmp, D, self.NT, D)
            ops.window_permute(tmp, self.pos_win, 1, self.g, self.wg, D, to_raster=False)
            self._pos_cache = {self.g: self.pos_win}
            self.shadow_valid = True
            return
        # the derived operands in ONE launch (round 5: vpu_cast2d_batched; five launches before): the two patch embeddings side
        # by side in the fused weight, the sum of their biases, the K-padded PuE weight, pos_embed[:, 1:] in window order
        ops.cast2d_batched([
            dict(src=self.Pm("backbone.patch_embed.proj.weight"), dst=(self.w_patch, 0), ld_src=k3, ld_dst=2 * self.k3p, rows=D, cols=k3),
            dict(src=self.Pm("patch_embed_coords.proj.weight"), dst=(self.w_patch, self.k3p), ld_src=k3, ld_dst=2 * self.k3p, rows=D,
                 cols=k3),
            dict(src=self.Pm("backbone.patch_embed.proj.bias"), src2=self.Pm("patch_embed_coords.proj.bias"), dst=self.b_patch,
                 ld_src=D, ld_dst=D, rows=1, cols=D),
            dict(src=self.Pm("neck.ffn_layer.lin1.weight"), dst=self.w_lin1p, ld_src=self.E, ld_dst=self.Epad, rows=2048, cols=self.E,
                 cols_pad=self.Epad),
            dict(src=(self.flat, po + D), dst=self.pos_win, ld_src=D, ld_dst=D, rows=self.NT, cols=D, perm=(self.g, self.wg)),
        ])
        self._pos_cache = {self.g: self.pos_win}      # other grids are re-derived from the new weights on demand
        self.shadow_valid = True

    def _regrid_adjoint(self, g):
        """R^T fp32 [g0^2, g^2] of the bicubic re-gridding of the g0 x g0 position grid onto g x g (row s, column t: the
        weight of trained position s in re-gridded position t), built once per grid from torch's own interpolate applied
        to the unit maps on the host -- the same linear map ``_pos_for`` applies to the embedding."""
        Rt = self._regrid_cache.get(g)
        if Rt is None:
            import torch.nn.functional as F
            g0 = self.g
            basis = torch.eye(g0 * g0, dtype=torch.float32).view(1, g0 * g0, g0, g0)
            Rt = F.interpolate(basis, size=(g, g), mode="bicubic", align_corners=False).reshape(g0 * g0, g * g)
            Rt = Rt.contiguous().to(self.dev)
            self._regrid_cache[g] = Rt
        return Rt

    def _pos_for(self, g):
        """Window-ordered position embedding for a g x g token grid.  The trained grid is used as it is; any other one
        (evaluation at another input size: scripts/evaluate_vpumodel.py:125 -> pos_embed.py:99-128, e.g. DAVIS at 672^2 =
        42 x 42 tokens, 9 windows) is the bicubic re-gridding of the trained embedding, exactly the tensor the reference
        installs with interpolate_pos_embed_inference -- computed once per grid and weight state (torch's bicubic
        interpolate: load-time plumbing, not the hot path)."""
        pw = self._pos_cache.get(g)
        if pw is None:
            import torch.nn.functional as F
            D, g0 = self.D, self.g
            po = self.names["backbone.pos_embed"][0]
            grid = self.flat[po + D: po + D + g0 * g0 * D].view(1, g0, g0, D).permute(0, 3, 1, 2)
            new = F.interpolate(grid, size=(g, g), mode="bicubic", align_corners=False).permute(0, 2, 3, 1)
            tmp = new.reshape(g * g, D).to(self.td).contiguous()
            pw = torch.empty(g * g, D, device=self.dev, dtype=self.td)
            ops.window_permute(tmp, pw, 1, g, self.wg, D, to_raster=False)
            self._pos_cache[g] = pw
        return pw

    def _kpe_for(self, g):
        k = self._kpe_cache.get(g)
        if k is None:
            k = self.kpe if g == self.g else pos2d_table(self.D, g, g).to(self.dev).to(self.td)
            self._kpe_cache[g] = k
        return k

    def zero_grad(self, lazy=False):
        """The flat gradient buffer <- 0.  ``lazy``: the caller promises that ONE backward of a training-mode forward comes
        next and that nothing reads the gradients before it (the captured / timed training step: zero, forward, backward,
        optimizer).  The weights whose gradient a single GEMM produces -- the ViT blocks' four linears, 69 % of ViT-B's
        parameters -- are then left as they are: that GEMM WRITES their gradient instead of accumulating into it (0 + x = x:
        the same bits), so neither the 4 bytes per parameter of this fill nor the read of C in the GEMM's epilogue happen.
        What such a pass did not write by its end is zeroed then (``backward``, ``abort_pass``)."""
        self._lazy = set()
        if (lazy and self.lazy_zero and self.group_wgrad and self.dt == BF16 and not self.use_side and self.pack_wgrad
                and self._lazy_ranges is not None):
            ops.zero_ranges_(self.gflat, self._lazy_ranges)
            self._lazy = set(self._lazy_names)
        else:
            ops.zero_(self.gflat)

    def _lazy_plan(self):
        """(names, complement ranges) of zero_grad(lazy=True); None when a range is not a multiple of 4 floats."""
        names = [n for n in self.names if n.startswith("backbone.blocks.") and
                 n.endswith((".attn.qkv.weight", ".attn.proj.weight", ".mlp.fc1.weight", ".mlp.fc2.weight"))]
        ranges, pos = [], 0
        for n in names:                       # (self.names is in buffer order)
            o, _, numel = self.names[n]
            if o % 4 or numel % 4:
                return [], None
            if o > pos:
                ranges.append((pos, o - pos))
            pos = o + numel
        if self.total > pos:
            ranges.append((pos, self.total - pos))
        return names, (ranges if names else None)

    def _lazy_finish(self):
        """Zeroes what a lazily zeroed pass has not written (a weight that got no gradient in this pass)."""
        if self._lazy:
            ops.zero_ranges_(self.gflat, [(self.names[n][0], self.names[n][2]) for n in sorted(self._lazy, key=lambda k: self.names[k][0])])
            self._lazy = set()

    def attach_grads(self):
        """``param.grad`` must be the views of the flat gradient buffer the kernels accumulate into.  A caller that ran
        ``optimizer.zero_grad()`` (``set_to_none=True`` is torch's default, as used by the reference at trainer.py:197,202)
        has dropped them: the buffer is then zeroed -- that call's meaning -- and the views are attached again; a gradient
        tensor of the caller's own is copied in first.  Two sentinel parameters are checked per call, all of them only
        when a sentinel is off."""
        if not self.params:
            return
        def ok(e):
            p, o, shape, numel = e
            return p.grad is not None and p.grad.data_ptr() == self.gflat.data_ptr() + 4 * o
        if ok(self.params[0]) and ok(self.params[-1]):
            return
        for e in self.params:
            p, o, shape, numel = e
            if ok(e):
                continue
            seg = self.gflat[o:o + numel]
            if p.grad is None:
                seg.zero_()
            else:
                seg.copy_(p.grad.reshape(-1))
            p.grad = seg.view(shape)

    # ------------------------------------------------------------------------------------------ helpers
    def _new(self, *shape, dtype=None):
        return torch.empty(shape, device=self.dev, dtype=dtype or self.td)

    def _upload(self, t, dtype):
        """Prompt tensors as contiguous ``dtype`` on the device.  Host tensors (the stroke simulator's curves and profiles)
        go through pinned memory with an asynchronous copy: a pageable copy blocks the host until everything queued on
        the stream before it has run -- the whole previous step."""
        if t.is_cuda:
            return t.to(dtype=dtype).contiguous()
        return t.to(dtype).contiguous().pin_memory().to(self.dev, non_blocking=True)

    def acc(self, var, g, take=False):
        """var.g += g.  ``take``: g may become var.g itself (caller guarantees g is dead afterwards)."""
        if var.g is None:
            if take:
                var.g = g
            else:
                var.g = torch.empty_like(g)
                ops.add4(g, None, None, None, var.g, g.numel())
        else:
            self._writable(var.g)
            ops.add4(var.g, g, None, None, var.g, g.numel())

    def _reduce_wb(self, part, nrows, prefix, Cdim):
        """part [nrows][2][C] -> G[prefix.weight] += sum part[:,0], G[prefix.bias] += sum part[:,1]; ONE launch when the two
        gradients are adjacent in the flat buffer (always the case for the norm layers: C % 8 == 0)."""
        ow, ob = self.names[prefix + ".weight"][0], self.names[prefix + ".bias"][0]
        if ob == ow + Cdim:
            # queued: one batched launch reduces the partial rows of every norm layer (flush_colsums)
            self._csq.append((part, self.G(prefix + ".weight"), nrows, 2 * Cdim))
        else:
            ops.colsum_f32(part.view(nrows, 2 * Cdim)[:, :Cdim].contiguous(), self.G(prefix + ".weight"), nrows, Cdim, beta=1.0)
            ops.colsum_f32(part.view(nrows, 2 * Cdim)[:, Cdim:].contiguous(), self.G(prefix + ".bias"), nrows, Cdim, beta=1.0)

    def _convseg_sums(self, part, part_b, nb, Cc):
        if self.r5_fused:
            self._csq.append((part, self.G("head.conv_seg.weight"), nb, Cc))
            self._csq.append((part_b, self.G("head.conv_seg.bias"), nb, 1))
        else:
            ops.colsum_f32(part, self.G("head.conv_seg.weight"), nb, Cc, beta=1.0)
            ops.colsum_f32(part_b, self.G("head.conv_seg.bias"), nb, 1, beta=1.0)

    def _colsum_to(self, dy, ld, gname, rows, N):
        part = self._new(64, N, dtype=torch.float32)
        ops.colsum(dy, ld, self.G(gname), part, rows, N, beta=1.0)

    def _wgrad(self, dy, ld_dy, x, ld_x, gname, N, K, M, ldc=None, bias=None):
        """G[N,K] += dy[M,N]^T x[M,K];  optionally G[bias][N] += column sums of dy, fused into the same launch (bf16
        path; the fp32 parity path uses the stand-alone column-sum kernel)."""
        fuse = bias is not None and self.dt == BF16
        gout = self.G(gname) if isinstance(gname, str) else gname      # (flat gradient buffer, element offset)
        if self.group_wgrad and self.dt == BF16 and not self.use_side:
            ldc_ = K if ldc is None else ldc
            args = (dy, x, gout, N, K, M, ld_dy, ld_x, ldc_, self.dt)
            acc = EPI_ACCUM
            if isinstance(gname, str) and gname in self._lazy:    # zero_grad(lazy=True): this GEMM writes the gradient
                self._lazy.discard(gname)
                acc = 0
            kw = dict(transA=True, transB=True, flags=EPI_OUT_F32 | acc, colsum=self.G(bias) if fuse else None)
            L = self._token_rows
            ctn = (K + 255) // 256      # column tiles of the K4 launches
            al16 = lambda t: (ops.ptr(t) & 15) == 0
            dcs = (fuse and self.dist_colsum and self.k4_wgrad and self.pack_wgrad and ctn > 1 and M % 64 == 0 and M > 2048
                   and N % 8 == 0 and K % 8 == 0 and ld_dy % 8 == 0 and ld_x % 8 == 0 and al16(dy) and al16(x))
            # (long reductions with 16-byte-aligned operands only: what vpu_gemm_grouped gives to the K4 kernels)
            if (self.unify_wgrad and self.pack_wgrad and (self.k3_wgrad or self.k4_wgrad) and L > 2048 and M > L and M % L == 0
                    and L % 64 == 0 and N % 8 == 0 and K % 8 == 0 and ld_dy % 8 == 0 and ld_x % 8 == 0 and ldc_ % 4 == 0):
                # S = M / L reduction slices as the batch entries of one problem: entry z reads rows [z L, (z + 1) L) of dy and
                # x and writes slab z (and its slice of the fused bias column sums); "_post": the column-sum jobs that add the
                # slabs to the gradient once the launch that holds the entry has been enqueued, "_target": where they end up
                S = M // L
                slab = self._new(S * N * K, dtype=torch.float32)
                post = [((slab, 0), gout, S, N * K, K, ldc_)]
                cs = None
                if fuse and dcs:
                    cs = (self._new(S * ctn * N, dtype=torch.float32), 0)            # (every word is written by exactly one tile)
                    post.append((cs, self.G(bias), S * ctn, N))
                elif fuse:
                    cs = (ops.zero_(self._new(S * N, dtype=torch.float32)), 0)       # (the kernel accumulates into it)
                    post.append((cs, self.G(bias), S, N))
                args = (dy, x, (slab, 0), N, K, L, ld_dy, ld_x, K, self.dt)
                kw = dict(transA=True, transB=True, flags=EPI_OUT_F32, colsum=cs, batch=S, sA=(L * ld_dy, 0), sB=(L * ld_x, 0),
                          sC=(N * K, 0), _post=post, _target=[gout] + ([self.G(bias)] if fuse else []))
                if fuse and dcs:
                    kw.update(cs_tn=ctn, cs_t0=0, cs_ld=N)
                M = L
            elif dcs and self.unify_wgrad:
                # un-sliced (a ViT block's linears): the column tiles' partial sums go to a [ctn, N] slab, one batched column
                # sum adds it to the bias gradient after the launch that holds the LAST part of this problem
                cs = (self._new(ctn * N, dtype=torch.float32), 0)
                kw.update(colsum=cs, cs_tn=ctn, cs_t0=0, cs_ld=N, _post=[(cs, self.G(bias), ctn, N)],
                          _target=[gout, self.G(bias)])
            self._wq.append((args, kw, ((N + 127) // 128) * ((K + 127) // 128) * kw.get("batch", 1), M))   # (the queue keeps dy and x alive)
            for t in (dy, x):
                tt = t[0] if isinstance(t, tuple) else t
                self._frozen.add(tt.data_ptr())
            kind = 0 if M <= 2048 else M
            if kind and self.ride_wgrad and self.pack_wgrad:
                self._pack_seen[kind] = self._pack_seen.get(kind, 0) + self._k2_tiles(self._wq[-1])
            if not self._in_lanes:        # (two-lane section of backward: the queue collects, the section's end launches)
                self._wgrad_autoflush(kind)
            return
        if not self.use_side:
            ops.gemm(dy, x, gout, N, K, M, ld_dy, ld_x, K if ldc is None else ldc, self.dt, transA=True,
                     transB=True, flags=EPI_OUT_F32 | EPI_ACCUM, colsum=self.G(bias) if fuse else None)
            if bias is not None and not fuse:
                self._colsum_to(dy, ld_dy, bias, M, N)
            return
        if self.side is None:
            self.side = torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream(self.dev)
        self.side.wait_stream(main)                      # dy and x are produced by work already queued on the main stream
        for t in (dy, x):
            tt = t[0] if isinstance(t, tuple) else t
            tt.record_stream(self.side)                  # the allocator must not recycle them under the side stream
            self._frozen.add(tt.data_ptr())
        with torch.cuda.stream(self.side):
            ops.gemm(dy, x, gout, N, K, M, ld_dy, ld_x, K if ldc is None else ldc, self.dt, transA=True,
                     transB=True, flags=EPI_OUT_F32 | EPI_ACCUM, colsum=self.G(bias) if fuse else None)
            if bias is not None and not fuse:
                self._colsum_to(dy, ld_dy, bias, M, N)

    def _wgrad_autoflush(self, kind):
        """Launch rule of the weight-gradient queue, evaluated after every new entry (kind = 0: short reductions, else the
        reduction length): a full group of one kind is launched at once; the other kinds stay."""
        same = [e for e in self._wq if (0 if e[3] <= 2048 else e[3]) == kind]
        if kind and self.ride_wgrad:
            # long reductions: the big problems ("anchors": a ViT block's four gradients are 216 tiles of 256 x 128 on
            # 256 CUs) are launched once they fill a round; the small ones queued meanwhile (the neck's 768 x 384
            # projections over the same 9408 rows) ride in the CUs such a launch leaves idle
            # (launched when the rounds of 256 tiles they need are at least 80 % full -- ViT-B's 216 tiles per block are,
            # ViT-L's 384 / ViT-H's 600 wait for the next block's: 768 = 3 full rounds, 1200 = 94 % of 5 -- or when the
            # group holds eight problems)
            anchors = [e for e in same if not self._is_rider(e)]
            T = sum(self._k2_tiles(e) for e in anchors)
            if self.pack_wgrad:
                # (full rounds: with the round cut exactly, the even spread of round 3 only risks a leftover launch)
                budget = self.pack_tiles(None, self.wgrad_round - (self.wgrad_round // 256) * self._reserved_cus())
                # a launch leaves when it would be FULL: up to 9 small problems (the neck's projections, the reduction slices
                # of the head / FPN gradients: ~10 tiles each -- 16 descriptors of them alone fill half a round) plus big
                # ones, the last of them cut so that the round is filled to the brim; or when two rounds have piled up
                while True:
                    mine = [e for e in self._wq if e[3] == kind]
                    T = sum(self._k2_tiles(e) for e in mine)
                    t_small = sum(self._k2_tiles(e) for e in [e for e in mine if self._pack_small(e)][:9])
                    t_big = sum(self._k2_tiles(e) for e in mine if not self._pack_small(e))
                    if not (T >= budget and (t_small + t_big >= budget or T >= 2 * budget)):
                        break
                    self.flush_wgrads(kind, ride=True, budget=budget)
                    if sum(self._k2_tiles(e) for e in self._wq if e[3] == kind) >= T:
                        break
            elif (T >= 200 and T >= self.wgrad_fill * 256 * ((T + 255) // 256)) or len(anchors) >= 8:
                self.flush_wgrads(kind, ride=True)
        elif len(same) >= 16:       # (one launch holds 16 descriptors; round 3 launched them in eights)
            self.flush_wgrads(kind)

    def _dgrad(self, dy, ld_dy, w, ldw, xvar, M, K, N, flags=0, aux=None, ldaux=0, defer=False):
        """x.g (+)= dy[M,N] W[N,K].  ``defer``: queued for the group launch of flush_group (the caller's tape runs it)."""
        f = flags
        if xvar.g is None:
            xvar.g = torch.empty_like(xvar.t)
        else:
            self._writable(xvar.g)
            f |= EPI_ACCUM
        args = (dy, w, xvar.g, M, K, N, ld_dy, ldw, K, self.dt)
        kw = dict(transB=True, flags=f, aux=aux, ldaux=ldaux)
        if defer:
            self._gemm_deferred(xvar.g.data_ptr(), args, kw)
        else:
            ops.gemm(*args, **kw)

    def _gemm_deferred(self, out_ptr, args, kw):
        """Queues one GEMM of a group of independent ones (flush_group launches them: the small ones -- <= 64 output tiles,
        K <= 2048 -- in ONE grouped launch).  A GEMM that accumulates into an output already in the queue runs after it."""
        if out_ptr in self._gq_out:
            self.flush_group()
        self._gq.append((args, kw))
        self._gq_out.add(out_ptr)

    def flush_group(self):
        q, self._gq = self._gq, []
        self._gq_out = set()
        is_small = [((a[3] + 127) // 128) * ((a[4] + 127) // 128) <= self.group_tiles and a[5] <= 2048 for a, _ in q]
        if sum(is_small) >= 2:
            ops.gemm_grouped([e for e, sm in zip(q, is_small) if sm])
            q = [e for e, sm in zip(q, is_small) if not sm]
        for args, kw in q:
            ops.gemm(*args, **kw)

    # ------------------------------------------------------------------------------------------ two lanes (round 5)
    @contextlib.contextmanager
    def _lane(self, s):
        """Everything launched inside runs on HIP stream ``s`` and the backward closures recorded inside will (Tape.append);
        ``s`` None = no-op.
        Allocator lifetime: tensors created inside (torch.empty on stream ``s``) are handed to kernels of the OTHER
        stream at the crossings without ``record_stream``.  That is safe only because (a) every such tensor is kept alive by
        the tape / the engine's buffers until the pass has ended -- the caching allocator cannot hand its block to another
        stream while a reference exists -- and (b) every crossing is fenced by the ``_xrec`` / ``_xwait`` event pair.  A
        tensor that is created in a lane, crosses, and is DROPPED before the pass ends would need ``record_stream``."""
        if s is None:
            yield
            return
        prev, self.tape.lane = self.tape.lane, s
        try:
            with torch.cuda.stream(s):
                yield
        finally:
            self.tape.lane = prev

    def _cur(self, s):
        return torch.cuda.current_stream(self.dev) if s is None else s

    def _xrec(self, src, on_bwd=None):
        """Forward: marks what stream ``src`` (None = the caller's) has queued so far; the matching ``_xwait(h, dst)`` makes
        ``dst`` wait for it.  Backward, mirrored: the closure of _xwait records on dst (dst's consumers of the crossing
        tensors have produced their gradients by then), the closure left here makes src wait for that.  Call both OUTSIDE
        ``_lane`` blocks or pass the streams explicitly -- their closures are lane-less."""
        h = {"ev": torch.cuda.Event(), "bev": None}
        h["ev"].record(self._cur(src))
        if self.training:
            def bwd():
                if h["bev"] is not None:
                    self._cur(src).wait_event(h["bev"])
                if on_bwd is not None:
                    on_bwd()
            self.tape.fns.append(bwd)
        return h

    def _xwait(self, h, dst, on_bwd=None):
        self._cur(dst).wait_event(h["ev"])
        if self.training:
            def bwd():
                if on_bwd is not None:
                    on_bwd()
                h["bev"] = torch.cuda.Event()
                h["bev"].record(self._cur(dst))
            self.tape.fns.append(bwd)

    # ------------------------------------------------------------------------------------------ ops with backward
    def linear(self, x, wname, bname, M, N, K, act=None, resid=None, out=None, x_grad=True, group=False, pre_masked=None):
        """y = act(x W^T + b) [+ resid].  ``out`` = (Var, col_offset, ld) writes into a column slice of a wider map.
        ``pre_masked``: a one-element list the producer of y.g sets to True when it has already multiplied the gradient
        by the ReLU mask of y (e.g. in a dgrad epilogue, EPI_DRELU with aux = y) -- the separate mask pass is skipped."""
        assert not (act and resid is not None)
        if out is None:
            y = Var(self._new(M, N))
            yt, yoff, ldc = y.t, 0, N
        else:
            y, yoff, ldc = out
            yt = y.t
        flags = EPI_BIAS | (EPI_RELU if act == "relu" else 0) | (EPI_RESID if resid is not None else 0)
        group = group and self.dt == BF16 and act is None and resid is None and out is None
        g_args = (x.t, self.W(wname), (yt, yoff), M, N, K, K, K, ldc, self.dt)
        g_kw = dict(flags=flags, bias=self.Pm(bname), resid=None if resid is None else resid.t, ldr=N)
        if group:     # the caller flushes the group (flush_group) before anything reads y
            self._gemm_deferred(yt.data_ptr(), g_args, g_kw)
        else:
            ops.gemm(*g_args, **g_kw)
        if self.training:
            def bwd():
                if y.g is None:
                    return
                dy = (y.g, yoff)
                if resid is not None:
                    assert out is None
                    self.acc(resid, y.g, take=True)
                if act == "relu" and not (pre_masked is not None and pre_masked[0]):
                    dz = self._new(M, N)
                    ops.act_bwd(dy, ldc, (yt, yoff), ldc, dz, N, M, N, 0, self.dt)
                    dy, ld = dz, N
                else:
                    ld = ldc
                self._wgrad(dy, ld, x.t, K, wname, N, K, M, bias=bname)
                if x_grad:
                    self._dgrad(dy, ld, self.W(wname), K, x, M, K, N, defer=group)
            self.tape.append(bwd)
        return y

    def mlp(self, x, p1, p2, M, K, Hd, N, act, resid=None, w1=None, k_grad=None, x_grad=True):
        """y = act(x W1^T + b1) W2^T + b2 [+ resid], with the activation backward fused into fc2's dgrad epilogue."""
        Kp = K
        h = self._new(M, Hd)
        if act == "gelu":
            pre = self._new(M, Hd)
            ops.gemm(x.t, self.W(p1 + ".weight") if w1 is None else w1, h, M, Hd, Kp, Kp, Kp, Hd, self.dt,
                     flags=EPI_BIAS | EPI_SAVE_DGELU | EPI_GELU, bias=self.Pm(p1 + ".bias"), preact=pre)
        else:
            pre = None
            ops.gemm(x.t, self.W(p1 + ".weight") if w1 is None else w1, h, M, Hd, Kp, Kp, Kp, Hd, self.dt,
                     flags=EPI_BIAS | EPI_RELU, bias=self.Pm(p1 + ".bias"))
        y = Var(self._new(M, N))
        ops.gemm(h, self.W(p2 + ".weight"), y.t, M, N, Hd, Hd, Hd, N, self.dt,
                 flags=EPI_BIAS | (EPI_RESID if resid is not None else 0), bias=self.Pm(p2 + ".bias"),
                 resid=None if resid is None else resid.t, ldr=N)
        if self.training:
            def bwd():
                dy = y.g
                if dy is None:
                    return
                if resid is not None:
                    self.acc(resid, dy, take=True)
                self._wgrad(dy, N, h, Hd, p2 + ".weight", N, Hd, M, bias=p2 + ".bias")
                dpre = self._new(M, Hd)
                ops.gemm(dy, self.W(p2 + ".weight"), dpre, M, Hd, N, N, Hd, Hd, self.dt, transB=True,
                         flags=EPI_MULAUX if act == "gelu" else EPI_DRELU, aux=pre if act == "gelu" else h, ldaux=Hd)
                kg = K if k_grad is None else k_grad
                # dW1[Hd, kg] += dpre^T x  (x may be K-padded: ld Kp, only kg columns are real)
                self._wgrad(dpre, Hd, x.t, Kp, p1 + ".weight", Hd, kg, M, bias=p1 + ".bias")
                if x_grad:
                    self._dgrad(dpre, Hd, self.W(p1 + ".weight") if w1 is None else w1, Kp, x, M, Kp, Hd)
            self.tape.append(bwd)
        return y

    def layernorm(self, x, prefix, rows, Cdim, eps, pe=None, pe_rows=0, pe_var=None):
        """y = LN(x).  ``pe`` [pe_rows, C]: also yy = y + pe[row % pe_rows] from the same launch -- the position-embedding
        add of the DMA neck (transformer.py:439-457) -- and (y, yy) is returned; ``pe_var``: the Var behind ``pe`` when it
        takes a gradient (the prompt tokens).  The backward takes the two outputs' gradients in one launch."""
        if pe is not None and not self.fuse_ln_pe:       # ``fuse_ln_pe`` = False: the separate add launch
            y = self.layernorm(x, prefix, rows, Cdim, eps)
            return y, self.add_pe(y, pe, rows * Cdim, pe_rows * Cdim, pe_var=pe_var)
        y = Var(self._new(rows, Cdim))
        yy = Var(self._new(rows, Cdim)) if pe is not None else None
        mean, rstd = self._new(rows, dtype=torch.float32), self._new(rows, dtype=torch.float32)
        if pe is None:
            ops.layernorm_fwd(x.t, self.Pm(prefix + ".weight"), self.Pm(prefix + ".bias"), y.t, mean, rstd, rows, Cdim, eps)
        else:
            ops.layernorm_fwd_pe(x.t, self.Pm(prefix + ".weight"), self.Pm(prefix + ".bias"), y.t, mean, rstd, rows, Cdim,
                                 eps, pe, pe_rows, yy.t)
        if self.training:
            def bwd():
                dy, dy2 = y.g, (yy.g if yy is not None else None)
                if dy is None and dy2 is None:
                    return
                if dy2 is not None and pe_var is not None:
                    self.acc(pe_var, dy2)
                if dy is None:
                    dy, dy2 = dy2, None
                nblk = ops.layernorm_bwd_nblk(rows)
                part = self._new(nblk, 2, Cdim, dtype=torch.float32)
                dres = x.g
                if x.g is None or x.g.data_ptr() in self._frozen:
                    x.g = torch.empty_like(x.t)   # (a frozen dres is read, the sum goes to a fresh buffer: no wait needed)
                ops.layernorm_bwd(dy, x.t, self.Pm(prefix + ".weight"), mean, rstd, dres, x.g, part, rows, Cdim, dy2=dy2)
                self._reduce_wb(part, nblk, prefix, Cdim)
            self.tape.append(bwd)
        return y if pe is None else (y, yy)

    def sdpa(self, q, k, v, o, nb, H, nq, nk, hd, scale):
        """softmax(Q K^T * scale) V for nb*H independent (batch, head) problems.  q,k,v,o = (Var, col_off, ld, rows_per_b):
        head h of batch b starts at element (b*rows_per_b)*ld + col_off + h*hd.  Materialises S (fp32) and P."""
        (qv, qo, qld, qrows), (kv, ko, kld, krows), (vv, vo, vld, vrows), (ov, oo, old, orows) = q, k, v, o
        ldS = _rup(nk, 8)
        S = self._new(nb * H, nq, ldS, dtype=torch.float32)
        sS = (H * nq * ldS, nq * ldS)
        ops.gemm((qv.t, qo), (kv.t, ko), S, nq, nk, hd, qld, kld, ldS, self.dt, flags=EPI_OUT_F32, alpha=scale,
                 batch=nb * H, inner=H, sA=(qrows * qld, hd), sB=(krows * kld, hd), sC=sS)
        Pm = self._new(nb * H, nq, ldS)
        ops.softmax_fwd(S, ldS, Pm, ldS, nb * H * nq, nk)
        del S
        ops.gemm(Pm, (vv.t, vo), (ov.t, oo), nq, hd, nk, ldS, vld, old, self.dt, transB=True, batch=nb * H, inner=H,
                 sA=sS, sB=(vrows * vld, hd), sC=(orows * old, hd))
        if self.training:
            def bwd():
                if ov.g is None:
                    return
                for var in {id(qv): qv, id(kv): kv, id(vv): vv}.values():
                    assert var.g is None, "sdpa inputs must be single-use projections"
                    var.g = torch.empty_like(var.t)
                dO = (ov.g, oo)
                dP = self._new(nb * H, nq, ldS, dtype=torch.float32)
                ops.gemm(dO, (vv.t, vo), dP, nq, nk, hd, old, vld, ldS, self.dt, flags=EPI_OUT_F32, batch=nb * H,
                         inner=H, sA=(orows * old, hd), sB=(vrows * vld, hd), sC=sS)
                # dV = P^T dO
                ops.gemm(Pm, dO, (vv.g, vo), nk, hd, nq, ldS, old, vld, self.dt, transA=True, transB=True, batch=nb * H,
                         inner=H, sA=sS, sB=(orows * old, hd), sC=(vrows * vld, hd))
                dS = self._new(nb * H, nq, ldS)
                ops.softmax_bwd(Pm, ldS, dP, ldS, dS, nb * H * nq, nk, scale)
                del dP
                ops.gemm(dS, (kv.t, ko), (qv.g, qo), nq, hd, nk, ldS, kld, qld, self.dt, transB=True, batch=nb * H,
                         inner=H, sA=sS, sB=(krows * kld, hd), sC=(qrows * qld, hd))
                ops.gemm(dS, (qv.t, qo), (kv.g, ko), nk, hd, nq, ldS, qld, kld, self.dt, transA=True, transB=True,
                         batch=nb * H, inner=H, sA=sS, sB=(qrows * qld, hd), sC=(krows * kld, hd))
            self.tape.append(bwd)

    def flash_attention(self, qkv, O, nb, H, n, hd, D, scale):
        """Fused attention of one ViT block on the fused qkv activation [rows, 3D] -> O [rows, D] (bf16, hd 32/64)."""
        lse = self._new(nb * H, n, dtype=torch.float32)
        ops.attn_fwd((qkv.t, 0), (qkv.t, D), (qkv.t, 2 * D), O.t, lse, nb, H, n, hd, 3 * D, D, scale)
        if self.training:
            def bwd():
                if O.g is None:
                    return
                assert qkv.g is None
                qkv.g = torch.empty_like(qkv.t)
                delta = self._new(nb * H, n, dtype=torch.float32)
                ops.attn_bwd((qkv.t, 0), (qkv.t, D), (qkv.t, 2 * D), O.t, O.g, lse, delta, (qkv.g, 0), (qkv.g, D),
                             (qkv.g, 2 * D), nb, H, n, hd, 3 * D, D, 3 * D, scale)
            self.tape.append(bwd)

    def cross_attention(self, Qp, Kp, Vp, O, nb, H, nq, nk, hd, ld, scale):
        """Fused attention of the DMA neck: projected queries [nb*nq, ld], keys / values [nb*nk, ld] -> O [nb*nq, ld]
        (bf16; the [nb, H, nq, nk] scores are never materialised).
        Round 5, split launches (VPU_XATTN_SPLIT, default 1 = off; opt-in, measured level): the neck's attentions pair 48 prompt tokens with the image tokens --
        nb * H = 96 workgroups that walk 784 keys (tokens -> image) or 784 queries (the dK / dV of image -> tokens) in 25 chunks,
        latency-bound on a third of the chip.  The long side is cut into S ranges that run as S batch entries of one launch:
        * long QUERIES (image -> tokens): entries share the keys (kdiv = S); forward and dQ are complete per entry, dK / dV come
          out as S partial sums, added in order by vpu_sum_groups;
        * long KEYS (tokens -> image): entries share the queries (qdiv = S); the forward leaves S partial softmaxes, merged by
          vpu_attn_combine (out = sum_s exp(lse_s - lse) o_s); dQ is S partial sums; dK / dV are complete."""
        S = self.xattn_split
        long_q = S > 1 and nq >= 4 * nk and nq % (4 * S) == 0 and nq // S >= 64
        long_k = S > 1 and not long_q and nk >= 4 * nq and nk % S == 0 and nk // S >= 64 and nq % 4 == 0
        if long_q:
            lse = self._new(nb * S * H, nq // S, dtype=torch.float32)
            ops.xattn_fwd_split(Qp.t, Kp.t, Vp.t, O.t, lse, nb * S, H, nq // S, nk, hd, ld, ld, ld, scale, 1, S)
        elif long_k:
            lse = self._new(nb * H, nq, dtype=torch.float32)
            o_s, lse_s = self._new(nb * S * nq, ld), self._new(nb * S * H, nq, dtype=torch.float32)
            ops.xattn_fwd_split(Qp.t, Kp.t, Vp.t, o_s, lse_s, nb * S, H, nq, nk // S, hd, ld, ld, ld, scale, S, 1)
            ops.attn_combine(o_s, lse_s, O.t, lse, nb, H, nq, hd, S, ld, ld)
        else:
            lse = self._new(nb * H, nq, dtype=torch.float32)
            ops.xattn_fwd(Qp.t, Kp.t, Vp.t, O.t, lse, nb, H, nq, nk, hd, ld, ld, ld, scale)
        if self.training:
            def bwd():
                if O.g is None:
                    return
                for var in {id(Qp): Qp, id(Kp): Kp, id(Vp): Vp}.values():
                    assert var.g is None, "attention inputs must be single-use projections"
                    var.g = torch.empty_like(var.t)
                if long_q:
                    delta = self._new(nb * S * H, nq // S, dtype=torch.float32)
                    dkp, dvp = self._new(nb * S * nk, ld), self._new(nb * S * nk, ld)
                    ops.xattn_bwd_split(Qp.t, Kp.t, Vp.t, O.t, O.g, lse, delta, Qp.g, dkp, dvp, nb * S, H, nq // S, nk, hd, ld, ld,
                                        ld, ld, ld, scale, 1, S)
                    ops.sum_groups(dkp, Kp.g, nb, S, nk * ld)
                    ops.sum_groups(dvp, Vp.g, nb, S, nk * ld)
                elif long_k:
                    delta = self._new(nb * H, nq, dtype=torch.float32)
                    dqp = self._new(nb * S * nq, ld)
                    ops.xattn_bwd_split(Qp.t, Kp.t, Vp.t, O.t, O.g, lse, delta, dqp, Kp.g, Vp.g, nb * S, H, nq, nk // S, hd, ld, ld,
                                        ld, ld, ld, scale, S, 1)
                    ops.sum_groups(dqp, Qp.g, nb, S, nq * ld)
                else:
                    delta = self._new(nb * H, nq, dtype=torch.float32)
                    ops.xattn_bwd(Qp.t, Kp.t, Vp.t, O.t, O.g, lse, delta, Qp.g, Kp.g, Vp.g, nb, H, nq, nk, hd, ld, ld, ld,
                                  ld, ld, scale)
            self.tape.append(bwd)

    def add_pe(self, x, pe, n, period, pe_var=None):
        """y = x + pe (pe broadcast with ``period`` elements; transformer.py:320,430)."""
        y = Var(torch.empty_like(x.t))
        ops.add_bcast(x.t, pe, y.t, n, period)
        if self.training:
            def bwd():
                if y.g is None:
                    return
                if pe_var is not None:
                    self.acc(pe_var, y.g)
                self.acc(x, y.g, take=True)
            self.tape.append(bwd)
        return y

    def mha(self, prefix, xq, xk, xv, B, nq, nk, internal, resid=None):
        """transformer.py Attention.forward (:499-521) + optional residual of the caller."""
        D, H = self.D, 8
        hd = internal // H
        # the three projections are independent: their forward GEMMs -- and, in backward, their dgrads -- are queued and
        # launched together (the 576-row ones in ONE grouped launch instead of a split-K GEMM + reduce each)
        if self.training:
            self.tape.append(self.flush_group)       # runs AFTER the three backward closures below
        Qp = self.linear(xq, prefix + ".q_proj.weight", prefix + ".q_proj.bias", B * nq, internal, D, group=True)
        Kp = self.linear(xk, prefix + ".k_proj.weight", prefix + ".k_proj.bias", B * nk, internal, D, group=True)
        Vp = self.linear(xv, prefix + ".v_proj.weight", prefix + ".v_proj.bias", B * nk, internal, D, group=True)
        self.flush_group()
        O = Var(self._new(B * nq, internal))
        if self.dt == BF16 and self.use_flash and hd % 16 == 0 and hd <= 128:
            self.cross_attention(Qp, Kp, Vp, O, B, H, nq, nk, hd, internal, 1.0 / math.sqrt(hd))
        else:
            self.sdpa((Qp, 0, internal, nq), (Kp, 0, internal, nk), (Vp, 0, internal, nk), (O, 0, internal, nq), B, H,
                      nq, nk, hd, 1.0 / math.sqrt(hd))
        return self.linear(O, prefix + ".out_proj.weight", prefix + ".out_proj.bias", B * nq, D, internal, resid=resid)

    def groupnorm(self, x, prefix, B, HW, Cdim, gelu):
        y = Var(torch.empty_like(x.t))
        nch = ops.groupnorm_nchunk()
        mean, rstd = self._new(B, dtype=torch.float32), self._new(B, dtype=torch.float32)
        stats = getattr(x, "gn_stats", None)
        if stats is not None:        # (round 4: conv_t2 left the statistics partials of its output: no statistics pass)
            ops.groupnorm_apply(x.t, self.Pm(prefix + ".weight"), self.Pm(prefix + ".bias"), y.t, mean, rstd, stats, B, HW,
                                Cdim, 1e-5, gelu)
        else:
            stats = self._new(B, nch, 2, dtype=torch.float64)
            ops.groupnorm_fwd(x.t, self.Pm(prefix + ".weight"), self.Pm(prefix + ".bias"), y.t, mean, rstd, stats, B, HW,
                              Cdim, 1e-5, gelu)
        if self.training:
            def bwd():
                if y.g is None:
                    return
                assert x.g is None
                x.g = torch.empty_like(x.t)
                part = self._new(B * nch, 2, Cdim, dtype=torch.float32)
                ops.groupnorm_bwd(y.g, x.t, self.Pm(prefix + ".weight"), self.Pm(prefix + ".bias"), mean, rstd, x.g,
                                  part, stats, B, HW, Cdim, gelu)
                self._reduce_wb(part, B * nch, prefix, Cdim)
            self.tape.append(bwd)
        return y

    def conv_t2(self, x, prefix, B, h, w, Cin, Cout, gn_next=False):
        """ConvTranspose2d(Cin, Cout, 2, stride=2) on channels-last tokens = GEMM + depth-to-space.  ``gn_next``: a
        GroupNorm(1, C) follows -- its statistics partials come out of the depth-to-space pass (y.gn_stats)."""
        M = B * h * w
        t = self._new(M, 4 * Cout)
        ops.gemm(x.t, self.W(prefix + ".weight"), t, M, 4 * Cout, Cin, Cin, 4 * Cout, 4 * Cout, self.dt, transB=True)
        y = Var(self._new(B * 4 * h * w, Cout))
        if gn_next and Cout % 8 == 0:
            y.gn_stats = self._new(B, ops.groupnorm_nchunk(), 2, dtype=torch.float64)
            ops.pixel_shuffle2_gn_stats(t, y.t, self.Pm(prefix + ".bias"), y.gn_stats, B, h, w, Cout)
        else:
            ops.pixel_shuffle2(t, y.t, self.Pm(prefix + ".bias"), B, h, w, Cout)
        del t
        if self.training:
            def bwd():
                if y.g is None:
                    return
                dt = self._new(M, 4 * Cout)
                if Cout % 8 == 0 and Cout <= 2048:
                    # (round 4: the bias gradient's partial sums come out of the same pass; one more job of the batched
                    # column sums instead of a second read of the map and two launches)
                    part, nb = ops.pixel_unshuffle2_sums(y.g, dt, B, h, w, Cout)
                    self._csq.append((part, self.G(prefix + ".bias"), nb, Cout))
                else:
                    ops.pixel_shuffle2(y.g, dt, None, B, h, w, Cout, inverse=True)
                    self._colsum_to(y.g, Cout, prefix + ".bias", B * 4 * h * w, Cout)
                # dW[Cin, 4Cout] += x^T dt   (launched directly: as a 36-tile rider of the grouped launches it displaced
                # the neck's small problems and cost more than it saved, 858 vs 864 images/s)
                if self.unify_wgrad and self.pack_wgrad and self.dt == BF16 and self.group_wgrad and not self.use_side:
                    # (round 4: queued like every other long reduction -- in the packed launches every tile counts, so it
                    # displaces nobody; a 37632-row one goes as four reduction slices)
                    self._wgrad(x.t, Cin, dt, 4 * Cout, prefix + ".weight", Cin, 4 * Cout, M)
                else:
                    ops.gemm(x.t, dt, self.G(prefix + ".weight"), Cin, 4 * Cout, M, Cin, 4 * Cout, 4 * Cout, self.dt,
                             transA=True, transB=True, flags=EPI_OUT_F32 | EPI_ACCUM)
                f = 0
                if x.g is None:
                    x.g = torch.empty_like(x.t)
                else:
                    self._writable(x.g)
                    f = EPI_ACCUM
                ops.gemm(dt, self.W(prefix + ".weight"), x.g, M, Cin, 4 * Cout, 4 * Cout, 4 * Cout, Cin, self.dt, flags=f)
            self.tape.append(bwd)
        return y

    def conv_s2(self, x, prefix, B, h, w, Cin, Cout):
        """Conv2d(Cin, Cout, 2, stride=2) on a channels-last [B, 2h, 2w, Cin] map = space-to-depth + GEMM."""
        M = B * h * w
        s2d = self._new(M, 4 * Cin)
        ops.pixel_shuffle2(x.t, s2d, None, B, h, w, Cin, inverse=True)
        y = Var(self._new(M, Cout))
        ops.gemm(s2d, self.W(prefix + ".weight"), y.t, M, Cout, 4 * Cin, 4 * Cin, 4 * Cin, Cout, self.dt,
                 flags=EPI_BIAS, bias=self.Pm(prefix + ".bias"))
        if self.training:
            def bwd():
                if y.g is None:
                    return
                self._wgrad(y.g, Cout, s2d, 4 * Cin, prefix + ".weight", Cout, 4 * Cin, M, bias=prefix + ".bias")
                ds = self._new(M, 4 * Cin)
                ops.gemm(y.g, self.W(prefix + ".weight"), ds, M, 4 * Cin, Cout, Cout, 4 * Cin, 4 * Cin, self.dt,
                         transB=True)
                dx = self._new(B * 4 * h * w, Cin)
                ops.pixel_shuffle2(ds, dx, None, B, h, w, Cin)
                self.acc(x, dx, take=True)
            self.tape.append(bwd)
        return y

    # ------------------------------------------------------------------------------------------ model
    def forward(self, image4, points, boxes=None, prompt_type=0, drop_mask=None, training=False, taps=None,
                materialize_aux=True, scribble=None, coord_override=None, prenorm=False):
        """image4 fp32 [B,4,H,W]; points fp32 [B,2n,3]; boxes int32 [B,5] (prompt_type 1); scribble (prompt_type 2) =
        (curve int32 [B,P,2] poly-line vertices, profiles float64 [B, 2*img] from isegm/model/scribble.py).
        Returns instances fp32 [B,1,H,W] (logits) and instances_aux fp32 [B,S,H,W].  ``materialize_aux=False`` skips the
        38.5 MB/img upsample of the P2CL similarities: aux is None, the low-resolution planes stay in ``self.sim_low``
        [B,S,h,w] for the fused loss (ops.p2cl_up_fwd_bwd) and backward() takes their gradient as ``d_sim_low``.
        ``coord_override`` fp32 [B,2,H,W]: the two click-map channels of the coordinate features as the CALLER made them
        (no disk maps / outlines are drawn); ``prenorm``: the rgb planes of image4 are normalised already -- both for the
        public ``backbone_forward`` (is_vpu_model.py:383-419)."""
        assert image4.is_cuda and image4.dtype == torch.float32 and image4.is_contiguous()
        if not self.shadow_valid:
            self.refresh_weights()
        self.training = training
        self.tape = Tape()
        tape = self.tape
        c = self.cfg
        B, H, W_ = image4.shape[0], image4.shape[2], image4.shape[3]
        D, P, heads = self.D, self.P, self.heads
        # the token grid follows the INPUT (the prompt vectors keep the constructor's image size, as in the reference:
        # is_vpu_model.py:189-230 builds them from self.image_size)
        if H != W_ or H % (P * self.wg) != 0:
            raise ValueError(f"input {H}x{W_}: square inputs whose side is a multiple of the {P * self.wg}-pixel window")
        g = H // P
        NT, nw_side = g * g, g // self.wg
        pos_win, kpe_tab = self._pos_for(g), self._kpe_for(g)
        M = B * NT
        self._token_rows = M          # the reduction length of the ViT blocks' weight gradients (the packed queue)
        n = points.shape[1] // 2
        points = self._upload(points, torch.float32)
        use_box = prompt_type == 1
        if use_box:
            boxes = self._upload(boxes, torch.int32)
        use_scr = prompt_type == 2
        if use_scr:
            curve = self._upload(scribble[0], torch.int32)
            prof = self._upload(scribble[1], torch.float64)
            assert curve.shape[0] == B and tuple(prof.shape) == (B, 2 * self.img)
        nq = 2 * self.nmax
        # round 5, two lanes: the prompt tokens' side of the DMA neck on its own stream (see the neck below).  What needs the
        # prompts only -- the PuE vectors, their MLP, layer 0's token self-attention -- is launched now, beside the backbone;
        # its backward closures are kept aside and join the tape at the neck's place (they belong to the neck's gradient range)
        lanes = (self.neck_lanes and training and self.dt == BF16 and self.use_flash and not self.use_side and self.fuse_ln_pe
                 and (D // 2 // 8) % 16 == 0 and D // 8 <= 128)
        T = None
        if lanes:
            if self._tok_stream is None:
                self._tok_stream = torch.cuda.Stream(device=self.dev)
            T = self._tok_stream
        tok = lambda: self._lane(T)

        def group(fn):      # the projections queued by fn() leave in one grouped launch; so do their dgrads (backward)
            if training:
                self.tape.append(self.flush_group)
            r = fn()
            self.flush_group()
            return r

        def proj(prefix, which, x_, rows, width):
            return self.linear(x_, f"{prefix}.{which}_proj.weight", f"{prefix}.{which}_proj.bias", rows, width, D, group=True)

        def attend(prefix, Qp, Kp, Vp, nq_, nk_, width, resid):
            O_ = Var(self._new(B * nq_, width))
            self.cross_attention(Qp, Kp, Vp, O_, B, 8, nq_, nk_, width // 8, width, 1.0 / math.sqrt(width // 8))
            return self.linear(O_, prefix + ".out_proj.weight", prefix + ".out_proj.bias", B * nq_, D, width, resid=resid)

        def prompt_vectors():
            pue_ = Var(self._new(B * nq, self.Epad))
            ops.pue_encode(points, boxes if use_box else None, self.lut, pue_.t, None, B, n, self.nmax, self.img, self.Epad)
            if use_scr:     # _guassinvector_scribble (is_vpu_model.py:294-352): the last valid positive row becomes the scribble
                ops.pue_scribble_rows(points, prof, pue_.t, None, B, n, self.nmax, self.img, self.Epad)
            return self.mlp(pue_, "neck.ffn_layer.lin1", "neck.ffn_layer.lin2", B * nq, self.Epad, 2048, D, "relu",
                            w1=self.w_lin1p, k_grad=self.E, x_grad=False)

        def tok_pre(l, q_, qq_, q0_):      # token side of layer l up to the token -> image queries: needs no image token
            p_ = f"neck.att.layers.{l}"
            if l == 0:
                q_ = self.mha(p_ + ".self_attn", q_, q_, q_, B, nq, nq, D)
            else:
                q_ = self.mha(p_ + ".self_attn", qq_, qq_, q_, B, nq, nq, D, resid=q_)
            q_, qq_ = self.layernorm(q_, p_ + ".norm1", B * nq, D, 1e-5, pe=q0_.t, pe_rows=B * nq, pe_var=q0_)
            Qt_ = group(lambda: proj(p_ + ".cross_attn_token_to_image", "q", qq_, B * nq, D // 2))
            return q_, qq_, Qt_

        def start_token_lane():
            fns_main, self.tape.fns = self.tape.fns, []

            def lanes_done():       # (backward leaves the two-lane section: what the queue collected meanwhile may go)
                self._in_lanes = False
                for kind in sorted({0 if e[3] <= 2048 else e[3] for e in self._wq}):
                    self._wgrad_autoflush(kind)
            h_fork = self._xrec(None, on_bwd=lanes_done)
            self._xwait(h_fork, T)
            with tok():
                q0_ = prompt_vectors()
                pre_ = tok_pre(0, q0_, None, q0_)
            early_, self.tape.fns = self.tape.fns, fns_main
            return early_, q0_, pre_

        if lanes and self.neck_lanes != 3:
            early, q0, q_pre = start_token_lane()
        # ---- a1-a4: prompts -> coordinate features -> fused patch embedding (window token order)
        if coord_override is not None:
            disks = coord_override.to(device=self.dev, dtype=torch.float32).contiguous()
            assert tuple(disks.shape) == (B, 2, H, W_)
        else:
            disks = self._new(B, 2, H, W_, dtype=torch.float32)
            ops.disk_maps(points, boxes if use_box else None, disks, B, n, H, W_, self.norm_radius)
            if use_scr:     # ISModel.draw_scribble (is_model.py:123-146): poly-line into the positive channel
                ops.draw_polyline(curve, disks, B, curve.shape[1], H, W_)
        if taps is not None:
            taps["disks"] = disks
        KP = 2 * self.k3p
        cols = self._new(M, KP)
        ops.patch_im2col(image4, disks, cols, B, H, W_, P, self.wg, prenorm=prenorm)
        x = Var(self._new(M, D))
        ops.gemm(cols, self.w_patch, x.t, M, D, KP, KP, KP, D, self.dt,
                 flags=EPI_BIAS | EPI_RESID, bias=self.b_patch, resid=pos_win, ldr=D, resid_period=NT)
        if training:
            x0 = x

            def bwd_patch():
                if x0.g is None:
                    return
                k3 = 3 * P * P
                for nm, co in (("backbone.patch_embed.proj", 0), ("patch_embed_coords.proj", self.k3p)):
                    self._wgrad(x0.g, D, (cols, co), KP, nm + ".weight", D, k3, M, bias=nm + ".bias")
                # pos_embed[:, 1:] gradient: sum over the batch, back to raster order
                s = self._new(NT * D, dtype=torch.float32)
                ops.colsum(x0.g, NT * D, s, None, B, NT * D, beta=0.0)   # rows = B <= 64: single-pass kernel
                r = self._new(NT * D, dtype=torch.float32)
                ops.window_permute(s, r, 1, g, self.wg, D, to_raster=True)
                gp = (self.gflat, self.names["backbone.pos_embed"][0] + D)
                if g == self.g:
                    ops.add4(gp, r, None, None, gp, NT * D)
                else:
                    # another input size: the tokens saw the bicubic re-gridding of pos_embed (pos_embed.py:99-128), a fixed
                    # linear map R of the g0 x g0 grid onto the g x g one; its adjoint takes the gradient back:
                    # d pos[g0^2, D] += R^T[g0^2, g^2] . d regridded[g^2, D]   (exact-fp32 GEMM into the flat gradient buffer)
                    g0 = self.g
                    ops.gemm(self._regrid_adjoint(g), r, gp, g0 * g0, D, NT, NT, D, D, F32, transB=True,
                             flags=EPI_OUT_F32 | EPI_ACCUM)
            self.tape.append(bwd_patch)
        if taps is not None:
            taps["tokens0_win"] = x.t
        # ---- a5/a6: ViT blocks on window-ordered tokens
        hd = D // heads
        for i in range(1, self.depth + 1):
            if training:
                self._mark_ready(f"backbone.blocks.{i - 1}.norm1.weight",
                                 f"backbone.blocks.{i}.norm1.weight" if i < self.depth else "backbone.fc_norm.weight")
            is_global = (i % self.group) == 0
            nwin = 1 if is_global else nw_side * nw_side
            nt = NT // nwin
            p = f"backbone.blocks.{i - 1}."
            h1 = self.layernorm(x, p + "norm1", M, D, 1e-6)
            qkv = self.linear(h1, p + "attn.qkv.weight", p + "attn.qkv.bias", M, 3 * D, D)
            O = Var(self._new(M, D))
            if self.dt == BF16 and hd % 16 == 0 and hd <= 128 and self.use_flash:
                self.flash_attention(qkv, O, B * nwin, heads, nt, hd, D, hd ** -0.5)
            else:
                self.sdpa((qkv, 0, 3 * D, nt), (qkv, D, 3 * D, nt), (qkv, 2 * D, 3 * D, nt), (O, 0, D, nt), B * nwin,
                          heads, nt, nt, hd, hd ** -0.5)
            x1 = self.linear(O, p + "attn.proj.weight", p + "attn.proj.bias", M, D, D, resid=x)
            h2 = self.layernorm(x1, p + "norm2", M, D, 1e-6)
            x = self.mlp(h2, p + "mlp.fc1", p + "mlp.fc2", M, D, D * c["mlp_ratio"], D, "gelu", resid=x1)
        xr = Var(self._new(M, D))
        ops.window_permute(x.t, xr.t, B, g, self.wg, D, to_raster=True)
        if training:
            xw = x

            def bwd_perm():
                if xr.g is None:
                    return
                gw = torch.empty_like(xr.g)
                ops.window_permute(xr.g, gw, B, g, self.wg, D, to_raster=False)
                self.acc(xw, gw, take=True)
            self.tape.append(bwd_perm)
        if taps is not None:
            taps["backbone"] = xr.t
        if training:
            self._mark_ready("backbone.fc_norm.weight", None)  # neck + head + the unused tail: final once they are done
        # ---- a7/a8: PuE vectors, a10/a11: DMA neck
        nQ, nK = B * nq * D, M * D
        hs = []
        if lanes:
            # Two lanes.  The neck alternates between the 576 prompt tokens (self-attention, LayerNorms, MLP, their projections:
            # ~30 launches per layer of <= 108 workgroups, latency-bound) and the 9408 image tokens (projections, LayerNorm:
            # full-chip launches).  One stream serialises them; here the token side runs on its own stream and meets the image
            # side at the two attentions of a layer only:
            #   image lane (the caller's stream): K / V of tokens -> image, Q of image -> tokens | ... | image -> tokens, norm4
            #   token lane:  self-attention, norm1, Q | tokens -> image, norm2, MLP, norm3, K / V  | next layer's self-attention ..
            # Events order the crossings (_xrec / _xwait; their backward closures mirror them), the backward closures run on the
            # lane they were recorded on, and the weight-gradient queue only collects while backward is inside (its launches
            # read operands of both lanes).  Same kernels on the same operands: the results do not depend on the lanes.
            if self.neck_lanes == 3:      # (3: the token lane starts here, nothing runs beside the backbone)
                early, q0, q_pre = start_token_lane()
            self.tape.fns.extend(early)
            if self.neck_lanes == 2:      # the token lane ends here: the neck itself on the caller's stream
                h_early = self._xrec(T)
                self._xwait(h_early, None)
                T = None
            k = xr
            kk = self.add_pe(k, kpe_tab, nK, NT * D)
            q, qq, Qt = q_pre
            for l in range(4):          # (l == 3: final_attn_token_to_image)
                p = f"neck.att.layers.{l}"
                t2i = p + ".cross_attn_token_to_image" if l < 3 else "neck.att.final_attn_token_to_image"
                i2t = p + ".cross_attn_image_to_token"
                Kt, Vt = group(lambda: (proj(t2i, "k", kk, M, D // 2), proj(t2i, "v", k, M, D // 2)))
                h_kv = self._xrec(None)
                if l < 3:
                    Qi = group(lambda: proj(i2t, "q", kk, M, D // 2))
                with tok():
                    if l == 3:
                        Qt = group(lambda: proj(t2i, "q", qq, B * nq, D // 2))
                    elif l > 0:
                        q, qq, Qt = tok_pre(l, q, qq, q0)
                self._xwait(h_kv, T)
                with tok():
                    q = attend(t2i, Qt, Kt, Vt, nq, NT, D // 2, q)
                    if l == 3:
                        q = self.layernorm(q, "neck.att.norm_final_attn", B * nq, D, 1e-5)
                    else:
                        q = self.layernorm(q, p + ".norm2", B * nq, D, 1e-5)
                        q = self.mlp(q, p + ".mlp.lin1", p + ".mlp.lin2", B * nq, D, 1024, D, "relu", resid=q)
                        q, qq = self.layernorm(q, p + ".norm3", B * nq, D, 1e-5, pe=q0.t, pe_rows=B * nq, pe_var=q0)
                        Ki, Vi = group(lambda: (proj(i2t, "k", qq, B * nq, D // 2), proj(i2t, "v", q, B * nq, D // 2)))
                if l == 3:
                    break
                h_tok = self._xrec(T)
                self._xwait(h_tok, None)
                k2 = attend(i2t, Qi, Ki, Vi, NT, nq, D // 2, k)
                k, kk = self.layernorm(k2, p + ".norm4", M, D, 1e-5, pe=kpe_tab, pe_rows=NT)
                if l != 2:
                    hs.append((q, k))
            hs.append((q, k))

            def lanes_begin():
                self._in_lanes = True
            h_join = self._xrec(T)
            self._xwait(h_join, None, on_bwd=lanes_begin)
        else:
            q0 = prompt_vectors()
            q, k = q0, xr
            qq = None
            kk = self.add_pe(k, kpe_tab, nK, NT * D)
            for l in range(3):
                p = f"neck.att.layers.{l}"
                # queries + query_pe / keys + key_pe (transformer.py:439-457) come out of the LayerNorm launch that produces the
                # queries / keys (Engine.layernorm(pe=...)); the sum of unchanged queries is taken once -- the reference adds it
                # again at the next layer's start (:439 after :457, :374 after :457): same numbers
                if l == 0:
                    q = self.mha(p + ".self_attn", q, q, q, B, nq, nq, D)
                else:
                    q = self.mha(p + ".self_attn", qq, qq, q, B, nq, nq, D, resid=q)
                q, qq = self.layernorm(q, p + ".norm1", B * nq, D, 1e-5, pe=q0.t, pe_rows=B * nq, pe_var=q0)
                q = self.mha(p + ".cross_attn_token_to_image", qq, kk, k, B, nq, NT, D // 2, resid=q)
                q = self.layernorm(q, p + ".norm2", B * nq, D, 1e-5)
                q = self.mlp(q, p + ".mlp.lin1", p + ".mlp.lin2", B * nq, D, 1024, D, "relu", resid=q)
                q, qq = self.layernorm(q, p + ".norm3", B * nq, D, 1e-5, pe=q0.t, pe_rows=B * nq, pe_var=q0)
                k2 = self.mha(p + ".cross_attn_image_to_token", kk, qq, q, B, NT, nq, D // 2, resid=k)
                k, kk = self.layernorm(k2, p + ".norm4", M, D, 1e-5, pe=kpe_tab, pe_rows=NT)
                if l != 2:
                    hs.append((q, k))
            q = self.mha("neck.att.final_attn_token_to_image", qq, kk, k, B, nq, NT, D // 2, resid=q)
            q = self.layernorm(q, "neck.att.norm_final_attn", B * nq, D, 1e-5)
            hs.append((q, k))
        q_out = Var(self._new(B * nq, D))
        ops.add4(q0.t, hs[0][0].t, hs[1][0].t, hs[2][0].t, q_out.t, nQ)
        if training:
            def bwd_qout():       # the gradient of the sum goes to its four terms: one launch (vpu_fanout_add)
                if q_out.g is None:
                    return
                vs = (q0, hs[0][0], hs[1][0], hs[2][0])
                if not self.r5_fused:
                    for v in vs:
                        self.acc(v, q_out.g)
                    return
                had = [v.g is not None for v in vs]
                for v, h_ in zip(vs, had):
                    if h_:
                        self._writable(v.g)
                    else:
                        v.g = torch.empty_like(q_out.g)
                ops.fanout_add(q_out.g, [v.g for v in vs], had, q_out.g.numel())
            self.tape.append(bwd_qout)
        # gates (is_vpu_model.py:106-121): x_i = x (1 + cg_i + sg_i) for the three (queries, keys) pairs -- round 5: the three
        # gates in ONE statistics launch and ONE pass over x forward (vpu_gate_fwd_n), one row pass + one column launch backward
        # (vpu_gate_bwd_n: dx summed in fp32 over the three gates, rounded once) instead of nine / six launches
        maps = [xr]
        ng = len(hs) if self.r5_fused else 0
        for qi, ki in (() if self.r5_fused else hs):       # (``r5_fused`` = False: one gate at a time, three launches each way)
            cg, sg = self._new(B, D, dtype=torch.float32), self._new(B, NT, dtype=torch.float32)
            aq, ac = self._new(B, D, dtype=torch.int32), self._new(B, NT, dtype=torch.int32)
            ops.gate_stats(qi.t, ki.t, cg, aq, sg, ac, B, nq, NT, D)
            xg = Var(self._new(M, D))
            ops.gate_apply(xr.t, cg, sg, xg.t, B, NT, D)
            if training:
                def bwd_gate(xg=xg, qi=qi, ki=ki, cg=cg, sg=sg, aq=aq, ac=ac):
                    if xg.g is None:
                        return
                    if any(v.g is None for v in (qi, ki)):
                        pend = [v for pair in hs for v in pair if v.g is None]
                        pool = ops.zero_(torch.empty(sum(v.t.numel() for v in pend), device=self.dev, dtype=self.td))
                        o_ = 0
                        for v in pend:
                            v.g = pool[o_:o_ + v.t.numel()].view_as(v.t)
                            o_ += v.t.numel()
                    accum = xr.g is not None
                    if not accum:
                        xr.g = torch.empty_like(xr.t)
                    for t_ in (xr.g, qi.g, ki.g):
                        self._writable(t_)
                    part = self._new(B, 64, D, dtype=torch.float32)
                    ops.gate_bwd(xg.g, xr.t, cg, aq, sg, ac, xr.g, accum, qi.g, ki.g, part, B, nq, NT, D)
                self.tape.append(bwd_gate)
            maps.append(xg)
        cg, sg = self._new(ng, B, D, dtype=torch.float32), self._new(ng, B, NT, dtype=torch.float32)
        aq, ac = self._new(ng, B, D, dtype=torch.int32), self._new(ng, B, NT, dtype=torch.int32)
        xgs = [Var(self._new(M, D)) for _ in range(ng)]
        if ng:
            ops.gate_fwd_n([qi.t for qi, _ in hs], [ki.t for _, ki in hs], xr.t, [v.t for v in xgs], cg, aq, sg, ac, B, nq, NT, D)
        if training and ng:
            def bwd_gates():
                live = [i for i in range(ng) if xgs[i].g is not None]
                if not live:
                    return
                if any(v.g is None for i in live for v in hs[i]):
                    # the gate gradients are scattered at arg-max positions into zeroed buffers: ONE fill for all of them
                    pend = [v for i in live for v in hs[i] if v.g is None]
                    pool = ops.zero_(torch.empty(sum(v.t.numel() for v in pend), device=self.dev, dtype=self.td))
                    o_ = 0
                    for v in pend:
                        v.g = pool[o_:o_ + v.t.numel()].view_as(v.t)
                        o_ += v.t.numel()
                accum = xr.g is not None
                if not accum:
                    xr.g = torch.empty_like(xr.t)
                for t_ in [xr.g] + [v.g for i in live for v in hs[i]]:
                    self._writable(t_)
                part = self._new(ng, B, 64, D, dtype=torch.float32)
                if len(live) == ng:
                    ops.gate_bwd_n([v.g for v in xgs], xr.t, cg, aq, sg, ac, xr.g, accum, [hs[i][0].g for i in live],
                                   [hs[i][1].g for i in live], part, B, nq, NT, D)
                else:       # (a map without a gradient: the single-gate launches over the views of the batched statistics)
                    for i in live:
                        ops.gate_bwd(xgs[i].g, xr.t, cg[i], aq[i], sg[i], ac[i], xr.g, accum, hs[i][0].g, hs[i][1].g, part[i], B, nq,
                                     NT, D)
                        accum = True
            self.tape.append(bwd_gates)
        maps += xgs
        if taps is not None:
            taps["q_out"] = q_out.t
        # FPN branches (is_vpu_model.py:55-86), channels-last
        o = self.out_dims
        c4 = max(o[0] * 2, D // 2)
        a = self.conv_t2(maps[0], "neck.down_4.0", B, g, g, D, c4, gn_next=True)
        a = self.groupnorm(a, "neck.down_4.1", B, 4 * NT, c4, True)
        a = self.conv_t2(a, "neck.down_4.3", B, 2 * g, 2 * g, c4, c4 // 2, gn_next=True)
        a = self.groupnorm(a, "neck.down_4.4", B, 16 * NT, c4 // 2, False)
        a = self.linear(a, "neck.down_4.5.weight", "neck.down_4.5.bias", B * 16 * NT, o[0], c4 // 2)
        d4 = self.groupnorm(a, "neck.down_4.6", B, 16 * NT, o[0], True)
        c8 = max(o[1], D // 2)
        a = self.conv_t2(maps[1], "neck.down_8.0", B, g, g, D, c8, gn_next=True)
        a = self.groupnorm(a, "neck.down_8.1", B, 4 * NT, c8, False)
        a = self.linear(a, "neck.down_8.2.weight", "neck.down_8.2.bias", B * 4 * NT, o[1], c8)
        d8 = self.groupnorm(a, "neck.down_8.3", B, 4 * NT, o[1], True)
        a = self.linear(maps[2], "neck.down_16.0.weight", "neck.down_16.0.bias", M, o[2], D)
        d16 = self.groupnorm(a, "neck.down_16.1", B, NT, o[2], True)
        c32 = max(o[3], D * 2)
        a = self.conv_s2(maps[3], "neck.down_32.0", B, g // 2, g // 2, D, c32)
        a = self.groupnorm(a, "neck.down_32.1", B, NT // 4, c32, False)
        a = self.linear(a, "neck.down_32.2.weight", "neck.down_32.2.bias", B * NT // 4, o[3], c32)
        d32 = self.groupnorm(a, "neck.down_32.3", B, NT // 4, o[3], True)
        feats = [(d4, 4 * g), (d8, 2 * g), (d16, g), (d32, g // 2)]
        if taps is not None:
            for i, (f, s) in enumerate(feats):
                taps[f"fpn{i}"] = f.t
        # ---- a12: head (swin_transformer.py:723-767)
        Cc = self.C
        Hs = 4 * g
        HW4 = Hs * Hs
        # fusion_conv over the concat of the four resized maps, by linearity: each level's column block of the fusion weight
        # is applied at that level's own resolution, the low-resolution products are resized and summed into the
        # full-resolution one (vpu_upsum_relu) -- no [B*112^2, 1024] concat, a third of the fusion FLOPs
        wf, ldf = self.W("head.fusion_conv.conv.weight"), 4 * Cc
        gf = self.G("head.fusion_conv.conv.weight")
        ys, zs = [], []
        ymask = [[False] for _ in range(4)]     # set by bwd_fusion: its dgrads apply the ReLU' of ys[i] in their epilogue
        for i in range(4):
            f, s = feats[i]
            ys.append(self.linear(f, f"head.convs.{i}.conv.weight", f"head.convs.{i}.conv.bias", B * s * s, Cc, o[i],
                                  act="relu", pre_masked=ymask[i]))
        fused = Var(self._new(B * HW4, Cc))
        ops.gemm(ys[0].t, wf, fused.t, B * HW4, Cc, Cc, Cc, ldf, Cc, self.dt, flags=EPI_BIAS,
                 bias=self.Pm("head.fusion_conv.conv.bias"))
        for i in (1, 2, 3):
            s = feats[i][1]
            z = self._new(B * s * s, Cc)
            ops.gemm(ys[i].t, (wf[0], wf[1] + i * Cc), z, B * s * s, Cc, Cc, Cc, ldf, Cc, self.dt)
            zs.append((z, s, s))
        ops.upsum_relu(fused.t, zs, B, Hs, Hs, Cc, self.dt)
        relu_done = [False]     # conv_seg's backward (the last writer of fused.g) has already applied fused's ReLU'
        if training:
            def bwd_fusion():
                if fused.g is None:
                    return
                if relu_done[0]:
                    dt_ = fused.g
                else:
                    dt_ = self._new(B * HW4, Cc)
                    ops.act_bwd(fused.g, Cc, fused.t, Cc, dt_, Cc, B * HW4, Cc, 0, self.dt)
                self._wgrad(dt_, Cc, ys[0].t, Cc, gf, Cc, Cc, B * HW4, ldc=ldf, bias="head.fusion_conv.conv.bias")
                fuse_mask = self.dt == BF16      # (the fp32 parity path keeps the separate mask pass)
                mk = dict(flags=EPI_DRELU, ldaux=Cc) if fuse_mask else {}
                self._dgrad(dt_, Cc, wf, ldf, ys[0], B * HW4, Cc, Cc, **(dict(mk, aux=ys[0].t) if fuse_mask else {}))
                ymask[0][0] = fuse_mask
                for i in (1, 2, 3):
                    s = feats[i][1]
                    dz = self._new(B * s * s, Cc)
                    ops.bilinear_cl_bwd(dt_, Cc, dz, Cc, B, s, s, Hs, Hs, Cc, self.dt)
                    self._wgrad(dz, Cc, ys[i].t, Cc, (gf[0], gf[1] + i * Cc), Cc, Cc, B * s * s, ldc=ldf)
                    self._dgrad(dz, Cc, (wf[0], wf[1] + i * Cc), ldf, ys[i], B * s * s, Cc, Cc,
                                **(dict(mk, aux=ys[i].t) if fuse_mask else {}))
                    ymask[i][0] = fuse_mask
            self.tape.append(bwd_fusion)
        seg = self._new(B * HW4, dtype=torch.float32)
        ops.convseg_fwd(fused.t, self.Pm("head.conv_seg.weight"), self.Pm("head.conv_seg.bias"), drop_mask, seg,
                        B * HW4, HW4, Cc)
        dm = c["head_d_model"]
        query = self.mlp(q_out, "head.ffn_layer.lin1", "head.ffn_layer.lin2", B * nq, dm, 2 * dm, Cc, "relu")
        qn, fn = Var(torch.empty_like(query.t)), Var(torch.empty_like(fused.t))
        inv_q, inv_f = self._new(B * nq, dtype=torch.float32), self._new(B * HW4, dtype=torch.float32)
        ops.l2norm_fwd(query.t, qn.t, inv_q, B * nq, Cc)
        ops.l2norm_fwd(fused.t, fn.t, inv_f, B * HW4, Cc)
        sim = self._new(B, nq, HW4, dtype=torch.float32)
        ops.gemm(qn.t, fn.t, sim, nq, HW4, Cc, Cc, Cc, HW4, self.dt, flags=EPI_OUT_F32 | EPI_AFFINE, post_mul=0.5,
                 post_add=0.5, batch=B, inner=1, sA=(nq * Cc, 0), sB=(HW4 * Cc, 0), sC=(nq * HW4, 0))
        if taps is not None:
            taps["seg_lowres"] = seg.view(B, 1, Hs, Hs)
            taps["sim_lowres"] = sim.view(B, nq, Hs, Hs)
            taps["fused"] = fused.t
        # ---- a13: final align_corners=True upsample
        inst = self._new(B, 1, H, W_, dtype=torch.float32)
        ops.upsample_ac_fwd(seg, inst, B, Hs, Hs, H, W_)
        aux = None
        self.sim_low = sim.view(B, nq, Hs, Hs)
        if materialize_aux:
            aux = self._new(B, nq, H, W_, dtype=torch.float32)
            ops.upsample_ac_fwd(sim, aux, B * nq, Hs, Hs, H, W_)
        tape.sim_low = self.sim_low
        if training:
            def bwd_head():
                d_inst, d_aux, d_sim_low = tape.out_grads
                if d_aux is not None or d_sim_low is not None:
                    if d_sim_low is not None:
                        dsim = d_sim_low.view(B, nq, HW4)
                    else:
                        dsim = self._new(B, nq, HW4, dtype=torch.float32)
                        ops.upsample_ac_bwd(d_aux, dsim, B * nq, Hs, Hs, H, W_)
                    if self.dt == BF16:
                        dsim_t = self._new(B, nq, HW4)
                        ops.cast2d(dsim, HW4, dsim_t, HW4, B * nq, HW4)
                    else:
                        dsim_t = dsim
                    qn.g, fn.g = torch.empty_like(qn.t), torch.empty_like(fn.t)
                    # d qn = 0.5 * dsim fn ;  d fn = 0.5 * dsim^T qn
                    ops.gemm(dsim_t, fn.t, qn.g, nq, Cc, HW4, HW4, Cc, Cc, self.dt, transB=True, alpha=0.5, batch=B,
                             inner=1, sA=(nq * HW4, 0), sB=(HW4 * Cc, 0), sC=(nq * Cc, 0))
                    ops.gemm(dsim_t, qn.t, fn.g, HW4, Cc, nq, HW4, Cc, Cc, self.dt, transA=True, transB=True, alpha=0.5,
                             batch=B, inner=1, sA=(nq * HW4, 0), sB=(nq * Cc, 0), sC=(HW4 * Cc, 0))
                    query.g = torch.empty_like(query.t)
                    ops.l2norm_bwd(qn.g, qn.t, inv_q, query.g, B * nq, Cc)
                    # both gradient paths of the fused map in one pass (vpu_head_grad_fused) when the mask loss is there too
                    one_pass = d_inst is not None and self.dt == BF16 and Cc in (64, 128, 256, 512)
                    if not one_pass:
                        fused.g = torch.empty_like(fused.t)
                        ops.l2norm_bwd(fn.g, fn.t, inv_f, fused.g, B * HW4, Cc)
                if d_inst is not None:
                    dseg = self._new(B * HW4, dtype=torch.float32)
                    ops.upsample_ac_bwd(d_inst, dseg, B, Hs, Hs, H, W_)
                    nb = ops.convseg_bwd_nblk(B * HW4)
                    part, part_b = self._new(nb, Cc, dtype=torch.float32), self._new(nb, dtype=torch.float32)
                    if (d_aux is not None or d_sim_low is not None) and one_pass:
                        fused.g = torch.empty_like(fused.t)
                        ops.head_grad_fused(fn.g, fn.t, inv_f, dseg, fused.t, self.Pm("head.conv_seg.weight"), drop_mask,
                                            fused.g, part, part_b, B * HW4, HW4, Cc)
                        relu_done[0] = True
                        # (conv_seg's weight / bias partial rows join the batched column sums of the norm layers: flush_colsums)
                        self._convseg_sums(part, part_b, nb, Cc)
                        return
                    accum = fused.g is not None
                    if not accum:
                        fused.g = torch.empty_like(fused.t)
                    self._writable(fused.g)
                    # accum bit 1: fused.g is complete with this call -> it leaves multiplied by [fused > 0] (one pass less
                    # over the 77-MB map than a separate ReLU' kernel)
                    ops.convseg_bwd(dseg, fused.t, self.Pm("head.conv_seg.weight"), drop_mask, fused.g, int(accum) | 2, part,
                                    part_b, B * HW4, HW4, Cc)
                    relu_done[0] = True
                    self._convseg_sums(part, part_b, nb, Cc)
            # must run BEFORE the closures of query / fused: insert at the position just after they were recorded
            self.tape.append(bwd_head)
        self.last_tape = tape if training else None
        return inst, aux

    def _mark_ready(self, first_name, next_name):
        """Records a tape marker: when backward reaches it, gflat[offset(first_name) : offset(next_name)) is final."""
        lo = self.names[first_name][0]
        hi = self.total if next_name is None else self.names[next_name][0]

        def marker():
            if self.ride_wgrad and self.group_wgrad and self.dt == BF16 and not self.use_side:
                # the long-reduction queue launches itself when its rounds are full (_wgrad), the short reductions go now;
                # with a reducer attached the range is reported as soon as nothing queued writes into it any more
                self.flush_wgrads(0)
                if self.grad_ready_hook is not None:
                    # the other long reductions (the FPN's / head's maps: a handful of problems over 37632 / 150528 rows that
                    # never fill a round and would wait for the end of backward) go now as well: ranges are reported in
                    # order, and a gradient of the head still in the queue would hold every block's range back until
                    # nothing of the backward is left to overlap the exchange with
                    for k in sorted({e[3] for e in self._wq if e[3] > 2048} - {self._token_rows}):
                        self.flush_wgrads(k)
                    self._pending_reports.append((lo, hi))
                    self._report_ready()
                    if len(self._pending_reports) > self.report_lag:
                        # the oldest range has waited ``report_lag`` whole blocks: what still writes into it goes now -- the
                        # neck's small long-reduction gradients wait to ride in a block's launch, and where the blocks'
                        # launches are whole rounds already (D = 1024: 768 tiles) there is never room for them before the end
                        # of backward.  (Round 6: two blocks, not one.  A ViT-B block's four gradients are 108 tiles of
                        # 256 x 256; flushed block by block they left in 15 launches of 42 % of a round per step -- 42.6 ms of
                        # K4P kernels per 13 steps against 22.5 without a reducer, +1.55 ms per step, the whole cost of
                        # running under the reducer at world size 1 (tools/dp_stats_ab.sh).  With two blocks of slack the
                        # full-round rule launches 240 tiles before the flush is due; a range reaches its collective one
                        # block later.)
                        lo0, hi0 = self._pending_reports[0]
                        off_of = lambda t: t[1] if isinstance(t, tuple) else None
                        late = [e for e in self._wq
                                if any(off_of(t) is None or lo0 <= off_of(t) < hi0 for t in self._targets(e))]
                        if late:
                            self.flush_wgrads(entries=late)
                return
            self.flush_wgrads()
            if self.grad_ready_hook is not None:
                self._pending_reports.append((lo, hi))
                self._report_ready()
        self.tape.append(marker)

    def _pack_small(self, e):
        """Small problems of the packed long-reduction queue: at most 24 tiles, or given as reduction slices (never cut)."""
        return self._k2_tiles(e) <= 24 or e[1].get("batch", 1) > 1

    def _k2_tiles(self, e):
        """Output tiles (256 x 128, or 256 x 256 for the K4 form) of a queued weight gradient (args: dy, x, g, N, K, M, ...: the
        gradient is [N, K]), times its batch entries (reduction slices)."""
        tn = self.wgrad_tn
        return ((e[0][3] + 255) // 256) * ((e[0][4] + tn - 1) // tn) * e[1].get("batch", 1)

    def _is_rider(self, e):
        return e[3] > 2048 and self._k2_tiles(e) <= 16

    @staticmethod
    def pack_tiles(total=None, cap=256):
        """Tiles per packed launch: one full round of the persistent 256 x 128 grid (``cap`` workgroups: one per CU, minus the
        CUs a reducer keeps free for RCCL) -- or, once the previous backward pass has shown how many tiles of this reduction
        length a pass queues, that total spread evenly over the ceil(total / cap) launches it needs anyway (2672 tiles: 11
        launches of 243 instead of 10 of 256 and a leftover launch that takes as long as a full one)."""
        cap = max(64, cap & ~7)
        if not total or total < cap:
            return cap
        launches = (total + cap - 1) // cap
        return (total + launches - 1) // launches

    def _reserved_cus(self):
        """CUs the attached reducer keeps out of the persistent GEMM grids while its buckets are in flight."""
        red = getattr(self.grad_ready_hook, "__self__", None)
        return int(getattr(red, "reserve_cus", 0) or 0) if self.grad_ready_hook is not None else 0

    def _report_ready(self, final=False):
        """Hands the finished gradient ranges to the reducer, in the order their markers were passed: a range is final once
        no queued weight gradient writes into it any more (the queue rules -- packed launches, riders, groups held back for
        a full round -- may carry the tail of a block's gradients into the next block's launch: its range then goes out one
        block later, still inside backward).  Launches and collectives share the stream, so "enqueued" is "ordered"."""
        if self.grad_ready_hook is None:
            self._pending_reports = []
            return
        off_of = lambda t: t[1] if isinstance(t, tuple) else None
        while self._pending_reports:
            lo, hi = self._pending_reports[0]
            if not final:
                busy = False
                for e in self._wq:
                    for t in self._targets(e):
                        o = off_of(t)
                        if o is None or lo <= o < hi:
                            busy = True
                            break
                    if busy:
                        break
                if busy:
                    return
            self._pending_reports.pop(0)
            self.flush_colsums()       # the range must be final before it is handed to the reducer
            if self.use_side:
                self.join_side()
            self.grad_ready_hook(lo, hi)

    def _split_entry(self, e, budget):
        """Cuts a queued weight gradient G[N, K] += dy[:, :N]^T x[:, :K] into (head, tail): head = the leading 256-row blocks
        of G (columns of dy) or the leading 128-column blocks of G (columns of x), whichever gives more tiles <= ``budget``;
        both are ordinary problems of the grouped launch (same operands at an offset, same leading dimensions).  The fused
        bias column sums belong to the tiles of G's first column block: a column cut keeps them in the head.  None if no
        block fits."""
        args, kw, _, red = e
        if kw.get("batch", 1) > 1:
            return None                    # (a sliced problem goes whole or waits)
        dy, x, g, N, K, M, ld_dy, ld_x, ldc, dt = args
        tn = self.wgrad_tn
        R, Cn = (N + 255) // 256, (K + tn - 1) // tn
        rows, cols = min(R - 1, budget // Cn), min(Cn - 1, budget // R)     # (a cut leaves something on both sides)
        if max(rows * Cn, cols * R) <= 0:
            return None
        adv = lambda t, n: (t[0], t[1] + n) if isinstance(t, tuple) else (t, n)
        t128 = lambda n_, k_: ((n_ + 127) // 128) * ((k_ + 127) // 128)
        cs = kw.get("colsum")
        hkw = {k: v for k, v in kw.items() if k != "_post"}     # (the slab is added up after the LAST part's launch: the tail keeps "_post")
        if rows * Cn >= cols * R:
            r = rows * 256
            head = ((dy, x, g, r, K, M, ld_dy, ld_x, ldc, dt), hkw, t128(r, K), red)
            tail = ((adv(dy, r), x, adv(g, r * ldc), N - r, K, M, ld_dy, ld_x, ldc, dt),
                    dict(kw, colsum=None if cs is None else adv(cs, r)), t128(N - r, K), red)
        else:
            c = cols * tn
            head = ((dy, x, g, N, c, M, ld_dy, ld_x, ldc, dt), hkw, t128(N, c), red)
            # (distributed column sums: every column tile owns a slice of the reduction -- the tail keeps the slab and
            # starts at global column tile cs_t0 + c / 256; classic form: the sums belong to the first column block)
            tkw = dict(kw, cs_t0=kw["cs_t0"] + c // 256) if kw.get("cs_tn", 0) > 1 else dict(kw, colsum=None)
            tail = ((dy, adv(x, c), adv(g, c), N, K - c, M, ld_dy, ld_x, ldc, dt), tkw, t128(N, K - c), red)
        return head, tail

    def flush_wgrads(self, only_kind=None, ride=False, keep_riders=False, touching=None, budget=None, entries=None):
        """Launches the queued weight gradients (``only_kind``: just the entries of that kind, 0 = short reductions, else a
        reduction length).  Short reductions (<= 2048 rows) go into one grouped launch; long ones are grouped per reduction
        length when that beats one split-K launch + reduce each, otherwise they are launched one by one.
        ``ride``: the kind's big problems plus as many of its queued small ones as fit one round of 256 tiles and one
        launch's 16 descriptors; ``keep_riders``: the small long-reduction problems stay queued (for a later ride);
        ``touching``: only the entries that read the buffer at that address (it is about to be modified in place)."""
        kind_of = lambda e: 0 if e[3] <= 2048 else e[3]
        q = [e for e in self._wq if only_kind is None or kind_of(e) == only_kind]
        if entries is not None:          # exactly these queued entries
            ids = set(id(e) for e in entries)
            q = [e for e in self._wq if id(e) in ids]
        if touching is not None:
            ptr_of = lambda t: (t[0] if isinstance(t, tuple) else t).data_ptr()
            q = [e for e in q if touching in (ptr_of(e[0][0]), ptr_of(e[0][1]))]
        if ride and budget is not None:
            # one full round.  A launch holds 16 descriptors, and the small problems (the neck's projections, the reduction
            # slices of the head / FPN gradients: a dozen tiles each) would use them up with the round half empty (measured:
            # 157 of 256 tiles in the first launch of a backward pass, 114 left over for a launch of their own at its end).
            # So: first the BIG problems in queue order (whole, then the leading row / column blocks of the first one that does
            # not fit; what is cut off stays at its place in the queue), at most 10 descriptors of them; then small ones, whole,
            # in queue order, while tiles and descriptors last.
            smalls = [e for e in q if self._pack_small(e)][:9]
            take, room = [], budget
            for e in smalls:
                t = self._k2_tiles(e)
                if t <= room:
                    take.append(e); room -= t
            nbig = 0
            for e in [e for e in q if not self._pack_small(e)]:
                t = self._k2_tiles(e)
                if t <= room and len(take) < 16:
                    take.append(e); room -= t; nbig += 1
                    continue
                cut = self._split_entry(e, room) if len(take) < 16 else None
                if cut is not None:
                    head, tail = cut
                    self._wq[[id(w) for w in self._wq].index(id(e))] = tail
                    self._wq.insert(0, head)      # (leaves the queue with this launch: `chosen` below goes by identity)
                    take.append(head); room -= self._k2_tiles(head)
                break
            if room > 0:                          # no (more) big problems: further small ones, whole, while descriptors last
                for e in [e for e in q if self._pack_small(e)][9:]:
                    t = self._k2_tiles(e)
                    if len(take) >= 16:
                        break
                    if t <= room:
                        take.append(e); room -= t
            q = take
        elif ride:
            anchors = [e for e in q if not self._is_rider(e)]
            T = sum(self._k2_tiles(e) for e in anchors)
            room, slots, take = 256 * ((T + 255) // 256) - T, 16 - len(anchors), []
            for e in q:
                if self._is_rider(e) and slots > 0 and self._k2_tiles(e) <= room:
                    take.append(e); slots -= 1; room -= self._k2_tiles(e)
            q = anchors + take
        elif keep_riders:
            q = [e for e in q if not self._is_rider(e)]
        if not q:
            return
        chosen = set(id(e) for e in q)
        self._wq = [e for e in self._wq if id(e) not in chosen]
        parts = {}   # kind -> entries
        for e in q:
            parts.setdefault(kind_of(e), []).append(e)
        for red, part in parts.items():
            # un-split tiles of one launch take nk K-tiles each (~0.6 us per K-tile on a CU of their own): worth it when
            # the tiles fill the chip (the four of a ViT block: 432) or when the alternative -- a split-K launch + reduce
            # per problem, ~35 us -- costs more (three or more over 9408 rows; never for a 2-tile problem over 150528)
            nk = (red + 63) // 64
            tiles = sum(e[2] for e in part)
            # (reductions beyond ~16k rows -- the head / FPN maps -- keep the per-problem split-K launch with up to 128
            # slices: measured 18.5 vs 17.9 ms per step when they were cut into 8 slices here)
            # slabs are summed into contiguous gradients only -- and ADDED to them (colsum_batched: out +=), so an entry
            # that must WRITE its gradient (zero_grad(lazy=True): no EPI_ACCUM, the target still holds the previous step's
            # values) is never sliced: it stays with the GEMM launches below, which honour the flag
            elig = [e for e in part if e[0][8] == e[0][4] and e[1].get("batch", 1) == 1 and not e[1].get("cs_tn")
                    and (e[1].get("flags", 0) & EPI_ACCUM)]
            etiles = sum(e[2] for e in elig)
            if self.split_wgrad and 2048 < red <= 16384 and len(elig) >= 2 and etiles < 200:
                self._wgrad_sliced(elig, red, etiles)
                taken = set(id(e) for e in elig)
                part = [e for e in part if id(e) not in taken]
                tiles -= etiles
            if not part:
                pass
            elif len(part) >= 2 and (red == 0 or tiles >= 200 or nk * 0.6 < len(part) * 35.0):
                for i in range(0, len(part), 16):        # (one launch holds 16 descriptors)
                    chunk = part[i:i + 16]
                    if len(chunk) == 1 and chunk[0][1].get("batch", 1) == 1 and not chunk[0][1].get("cs_tn"):
                        ops.gemm(*chunk[0][0], **self._pub(chunk[0][1]))
                    else:
                        ops.gemm_grouped([(e[0], self._pub(e[1])) for e in chunk])
            else:
                for args, kw, _, _ in part:
                    if kw.get("batch", 1) > 1 or kw.get("cs_tn"):
                        ops.gemm_grouped([(args, self._pub(kw))])     # (reduction slices / distributed column sums: the grouped K4 form)
                    else:
                        ops.gemm(*args, **self._pub(kw))
            for e in part:                               # the slabs of a sliced problem are added to its gradient by the
                self._csq.extend(e[1].get("_post", ()))  # batched column sums (flush_colsums: after this launch, same stream)
        if not self.use_side:
            self._frozen = set()
            for args, _, _, _ in self._wq:   # operands of the entries still queued stay frozen
                for t in (args[0], args[1]):
                    self._frozen.add((t[0] if isinstance(t, tuple) else t).data_ptr())
        if self._pending_reports and not self._reporting:
            self._reporting = True       # (join_side inside _report_ready flushes again)
            try:
                self._report_ready()
            finally:
                self._reporting = False

    @staticmethod
    def _pub(kw):
        """The keyword arguments of a queued GEMM without the queue's own annotations ("_post", "_target")."""
        return {k: v for k, v in kw.items() if not k.startswith("_")} if any(k.startswith("_") for k in kw) else kw

    @staticmethod
    def _targets(e):
        """(tensor, offset) gradient locations a queued entry ends up in (a sliced entry writes slabs first: "_target")."""
        args, kw = e[0], e[1]
        if "_target" in kw:
            return list(kw["_target"])
        out = [args[2]]
        if kw.get("colsum") is not None:
            out.append(kw["colsum"])
        return out

    def _wgrad_sliced(self, part, red, tiles):
        """Few output tiles over a long reduction (the DMA neck's 768 x 384 projections over the 9408 image tokens: 72
        tiles walking 147 K-tiles each on 72 of the 256 CUs, 150 us): every problem is cut into S reduction slices that
        run as independent problems of the grouped launch, each writing its raw fp32 slab (and its slice of the bias
        column sums); one batched column-sum launch adds the slabs to the gradients in slice order (deterministic)."""
        S = max(2, min(8, 288 // tiles, 16 // len(part)))   # (one launch holds 16 descriptors)
        kchunk = ((red + S - 1) // S + 63) // 64 * 64
        S = (red + kchunk - 1) // kchunk
        slab = self._new(sum(S * e[0][3] * e[0][4] for e in part), dtype=torch.float32)
        nb = sum(S * e[0][3] for e in part if e[1].get("colsum") is not None)
        bslab = ops.zero_(torch.empty(max(nb, 1), device=self.dev, dtype=torch.float32))   # the fused column sums accumulate
        sub, jobs, so, bo = [], [], 0, 0
        adv = lambda t, n: (t[0], t[1] + n) if isinstance(t, tuple) else (t, n)
        for args, kw, _, _ in part:
            dy, x, g, N, K, M, ld_dy, ld_x, ldc, dt = args
            cs = kw.get("colsum")
            for s in range(S):
                k0 = s * kchunk
                k1 = min(red, k0 + kchunk)
                kw_s = dict(kw, flags=EPI_OUT_F32, colsum=None if cs is None else (bslab, bo + s * N))
                sub.append(((adv(dy, k0 * ld_dy), adv(x, k0 * ld_x), (slab, so + s * N * K), N, K, k1 - k0, ld_dy, ld_x, K, dt), kw_s))
            jobs.append(((slab, so), g, S, N * K))
            so += S * N * K
            if cs is not None:
                jobs.append(((bslab, bo), cs, S, N))
                bo += S * N
        for i in range(0, len(sub), 16):
            ops.gemm_grouped(sub[i:i + 16])
        ops.colsum_batched(jobs)

    def flush_colsums(self):
        q, self._csq = self._csq, []
        if q:
            ops.colsum_batched(q)

    def join_side(self):
        """Launches the queued weight gradients; the main stream then waits for the side stream (if one is in use)."""
        self.flush_wgrads()
        if self.side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
        self._frozen.clear()

    def abort_pass(self):
        """Forgets everything a forward / backward pass has queued but not launched: the weight-gradient queue, the batched
        column sums, the deferred group, the frozen-buffer set, the pending gradient-range reports, the packing counters and
        the tapes.  For a pass that died half way -- a hipGraph capture that raised: its queue entries point at capture-pool
        buffers that were never written (a capture enqueues nothing), and the host-enqueued retry that follows must not
        launch them."""
        self._wq, self._csq, self._gq = [], [], []
        self._lazy_finish()       # (what zero_grad(lazy=True) left for a backward that will not happen)
        self._gq_out, self._frozen = set(), set()
        self._pending_reports, self._reporting = [], False
        self._pack_seen = {}
        self._in_lanes = False
        self.tape = Tape()
        self.last_tape = None

    def _writable(self, t):
        """Call before modifying gradient buffer ``t`` in place: a queued weight gradient that still reads it is launched
        first (the rest of the queue stays); with the side stream, the streams are joined."""
        if t is not None and t.data_ptr() in self._frozen:
            if self.use_side:
                self.join_side()
            else:
                self.flush_wgrads(touching=t.data_ptr())

    def backward(self, d_inst, d_aux, d_sim_low=None, tape=None):
        """Runs a recorded tape (``tape``: the one ``forward`` left in ``last_tape`` at that call; default: the most recent
        forward's).  d_inst fp32 [B,1,H,W] or None; d_aux fp32 [B,S,H,W] or None (or d_sim_low fp32 [B,S,h,w], the
        gradient of the low-resolution similarities from the fused loss).  Parameter gradients are ACCUMULATED into the
        flat gradient buffer (call zero_grad() between optimizer steps).  A tape runs once: a second backward of the same
        forward, or a backward with no training-mode forward before it, raises."""
        tape = self.last_tape if tape is None else tape
        if tape is None or tape.done or len(tape) == 0:
            raise RuntimeError("Engine.backward: no pending training-mode forward to back-propagate (its tape was "
                               "already run, or forward ran with training=False)")
        self.attach_grads()
        tape.out_grads[:] = [None if d_inst is None else d_inst.contiguous(), None if d_aux is None else d_aux.contiguous(),
                             None if d_sim_low is None else d_sim_low.contiguous()]
        self._pack_seen = {}
        for fn in reversed(tape.fns):
            fn()
        self._pack_total = self._pack_seen      # what a backward pass queues per reduction length: the next pass's launch sizes
        tape.fns = []
        tape.done = True
        if tape is self.last_tape:
            self.last_tape = None
        self.tape = Tape()
        self.join_side()          # every queued weight gradient is launched ...
        self.flush_colsums()      # ... before the batched column sums (norm-layer partials, slabs of the sliced reductions)
        self._lazy_finish()
        self._report_ready(final=True)
        if self.grad_ready_hook is not None:  # patch embeddings, cls/pos tokens: everything before block 0
            self.grad_ready_hook(0, self.names["backbone.blocks.0.norm1.weight"][0])
