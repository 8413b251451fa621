"""Thin host wrappers over the C ABI: torch tensors in (device memory + current HIP stream = plumbing), raw
pointers out.  No arithmetic happens here."""
import ctypes as C

import torch

from . import _lib
from ._lib import (BF16, F32, EPI_BIAS, EPI_PREACT, EPI_GELU, EPI_RELU, EPI_DGELU, EPI_DRELU, EPI_RESID, EPI_AFFINE,
                   EPI_ACCUM, EPI_OUT_F32, EPI_SAVE_DGELU, EPI_MULAUX, GemmDesc)

TORCH_DTYPE = {BF16: torch.bfloat16, F32: torch.float32}


def code_of(t):
    if isinstance(t, tuple):
        t = t[0]
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


# The raw handle of torch's current stream: torch.cuda.current_stream() builds a Stream object through three layers of
# Python device-index helpers (~5 us, once per kernel launch: ~2 ms of host time per training step);
# torch._C._cuda_getCurrentRawStream is the C entry point the same value comes from.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_handle(device_index=None):
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device() if device_index is None else device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _stream():
    return C.c_void_p(_stream_handle())


def ptr(t, off=0):
    """device pointer of tensor ``t`` advanced by ``off`` elements (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, tuple):
        t, o = t
        off += o
    return t.data_ptr() + off * t.element_size()


_WORKSPACE = {}


def split_k_workspace(device, nbytes=128 << 20):
    """fp32 scratch shared by every GEMM launched on ``device`` (launches on one stream are serialised, so one buffer
    is enough).  Passed to vpu_gemm, whose host side decides per launch whether to split K."""
    key = (device.index if device.index is not None else -1, _stream_handle(device.index))   # one scratch buffer per stream
    ws = _WORKSPACE.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.zeros(nbytes // 4, device=device, dtype=torch.float32)   # (zeroed: its tail holds tile counters)
        _WORKSPACE[key] = ws
    return ws


def _fill_desc(d, A, B, Cout, M, N, K, lda, ldb, ldc, dtype, transA=False, transB=False, flags=0, bias=None, resid=None,
               ldr=0, aux=None, ldaux=0, preact=None, alpha=1.0, post_mul=1.0, post_add=0.0, batch=1, inner=1,
               sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), resid_period=0, workspace="auto", colsum=None,
               cs_tn=0, cs_t0=0, cs_ld=0):
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(Cout)
    d.bias, d.resid, d.aux, d.preact = ptr(bias), ptr(resid), ptr(aux), ptr(preact)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr, d.ldaux = lda, ldb, ldc, ldr, ldaux
    d.batch, d.inner = batch, inner
    d.sAo, d.sAi = sA
    d.sBo, d.sBi = sB
    d.sCo, d.sCi = sC
    d.sRo, d.sRi = sR
    d.transA, d.transB = int(transA), int(transB)
    d.dtype, d.flags = dtype, flags
    d.resid_period = resid_period
    d.alpha, d.post_mul, d.post_add = alpha, post_mul, post_add
    d.colsum = ptr(colsum)
    d.cs_tn, d.cs_t0, d.cs_ld = int(cs_tn), int(cs_t0), int(cs_ld)      # (distributed column sums: include/vpu_hip.h)
    if workspace == "auto":
        c0 = Cout[0] if isinstance(Cout, tuple) else Cout
        workspace = split_k_workspace(c0.device)
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, dtype, **kw):
    """C = epilogue(alpha * op(A) op(B)); see include/vpu_hip.h.  A/B/C/resid/aux/preact may be tensors or
    (tensor, element_offset) tuples.  Keywords: transA, transB, flags, bias, resid, ldr, aux, ldaux, preact, alpha,
    post_mul, post_add, batch, inner, sA, sB, sC, sR, resid_period, workspace ("auto" | tensor | None), colsum."""
    d = GemmDesc()
    _fill_desc(d, A, B, Cout, M, N, K, lda, ldb, ldc, dtype, **kw)
    _lib.call("vpu_gemm", C.byref(d), _stream())


def gemm_grouped(problems):
    """ONE launch for up to 8 independent bf16 GEMMs (vpu_gemm_grouped): ``problems`` is a list of (args, kwargs) of
    ``gemm`` -- same transA / transB for all, batch 1, no split-K (every problem runs over its whole K)."""
    n = len(problems)
    arr = (GemmDesc * n)()
    for d, (args, kw) in zip(arr, problems):
        kw = dict(kw)
        kw["workspace"] = None
        _fill_desc(d, *args, **kw)
    _lib.call("vpu_gemm_grouped", arr, n, _stream())


def gemm_last_kernel():
    """rocprofv3 name of the kernel instantiation the last gemm / gemm_grouped call of this thread launched."""
    return _lib.load().vpu_gemm_last_kernel().decode()


def attn_last_kernel():
    """rocprofv3 name(s) of the kernel instantiation(s) the last attention call of this thread launched."""
    return _lib.load().vpu_attn_last_kernel().decode()


def attn_set_option(name, value):
    _lib.call("vpu_attn_set_option", name.encode(), int(value))


def gemm_set_option(name, value):
    _lib.call("vpu_gemm_set_option", name.encode(), int(value))


def gemm_get_option(name):
    """Effective value of a kernel-family option ("k2" / "k3" / "k5") as the loaded library's dispatch reads it."""
    import ctypes
    v = ctypes.c_int32(0)
    _lib.call("vpu_gemm_get_option", name.encode(), ctypes.byref(v))
    return int(v.value)


def layernorm_fwd(x, w, b, y, mean, rstd, rows, Cdim, eps):
    _lib.call("vpu_layernorm_fwd", ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, Cdim, eps, code_of(x),
              _stream())


def layernorm_fwd_pe(x, w, b, y, mean, rstd, rows, Cdim, eps, pe, pe_rows, y2):
    """y = LN(x), y2 = y + pe[row % pe_rows] in one launch (vpu_layernorm_fwd_pe)."""
    _lib.call("vpu_layernorm_fwd_pe", ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, Cdim, eps, ptr(pe),
              int(pe_rows), ptr(y2), code_of(x), _stream())


def layernorm_bwd_nblk(rows):
    return _lib.load().vpu_layernorm_bwd_nblk(rows)


def layernorm_bwd(dy, x, w, mean, rstd, dres, dx, part, rows, Cdim, dy2=None):
    """``dy2``: a second gradient of the output, summed with dy in fp32 (vpu_layernorm_bwd2)."""
    if dy2 is None:
        _lib.call("vpu_layernorm_bwd", ptr(dy), ptr(x), ptr(w), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(part), rows,
                  Cdim, code_of(x), _stream())
    else:
        _lib.call("vpu_layernorm_bwd2", ptr(dy), ptr(dy2), ptr(x), ptr(w), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                  ptr(part), rows, Cdim, code_of(x), _stream())


def edt(mask_u8, zero_border=True):
    """Exact Euclidean distance transform of uint8 masks [B, H, W] on the GPU (vpu_edt) -> float32 [B, H, W]."""
    B, H, W = mask_u8.shape
    mask_u8 = mask_u8.contiguous()
    scratch = torch.empty(B, H, W, device=mask_u8.device, dtype=torch.int32)
    dist = torch.empty(B, H, W, device=mask_u8.device, dtype=torch.float32)
    _lib.call("vpu_edt", ptr(mask_u8), ptr(scratch), ptr(dist), B, H, W, int(zero_border), _stream())
    return dist


def chamfer5(mask_u8, zero_border=True):
    """cv2.distanceTransform(mask, DIST_L2, 5) -- the 5 x 5 chamfer approximation the training simulators use -- of uint8 masks
    [B, H, W] on the GPU (vpu_chamfer5) -> float32 [B, H, W]."""
    B, H, W = mask_u8.shape
    mask_u8 = mask_u8.contiguous()
    scratch = torch.empty(B, H + 2, W + 2, device=mask_u8.device, dtype=torch.int32)
    dist = torch.empty(B, H, W, device=mask_u8.device, dtype=torch.float32)
    _lib.call("vpu_chamfer5", ptr(mask_u8), ptr(scratch), ptr(dist), B, H, W, int(zero_border), _stream())
    return dist


def cc_roots(mask_u8):
    """8-connected components of uint8 masks [B, H, W] (vpu_cc_roots) -> int32 [B, H, W]: smallest linear index of the
    pixel's component, -1 on background."""
    B, H, W = mask_u8.shape
    mask_u8 = mask_u8.contiguous()
    roots = torch.empty(B, H, W, device=mask_u8.device, dtype=torch.int32)
    _lib.call("vpu_cc_roots", ptr(mask_u8), ptr(roots), B, H, W, _stream())
    return roots


def cc_table(roots, kmax=4096):
    """Per-component (root, pixel count, ymin, ymax, xmin, xmax) of a cc_roots labelling (vpu_cc_table) -> int32
    [kmax * 6 + 1] on the device: kmax rows in no particular order, then the component count."""
    B, H, W = roots.shape
    slots = torch.empty(B, H, W, device=roots.device, dtype=torch.int32)
    table = torch.empty(kmax * 6 + 1, device=roots.device, dtype=torch.int32)
    _lib.call("vpu_cc_table", ptr(roots), ptr(slots), ptr(table), kmax, B, H, W, _stream())
    return table


def colsum_batched(jobs):
    """jobs: list of (in fp32 [rows, C], out fp32 [C] (tensor or (tensor, offset)), rows, C[, row_len, out_ld]): out += column
    sums, 64 per launch.  row_len / out_ld: the output is a [C / row_len][row_len] block of a matrix with leading dimension
    out_ld."""
    for i in range(0, len(jobs), 64):
        chunk = jobs[i:i + 64]
        arr = (_lib.ColsumJob * len(chunk))()
        for j, job in zip(arr, chunk):
            inp, out, rows, Cdim = job[:4]
            j.inp, j.out, j.nrows, j.ncols = ptr(inp), ptr(out), rows, Cdim
            j.row_len, j.out_ld = (job[4], job[5]) if len(job) > 4 and job[4] and job[5] != job[4] else (0, 0)
        _lib.call("vpu_colsum_batched", arr, len(chunk), _stream())


def colsum_f32(inp, out, rows, Cdim, beta=0.0):
    _lib.call("vpu_colsum_f32", ptr(inp), ptr(out), rows, Cdim, beta, _stream())


def colsum(inp, ld, out, part, rows, Cdim, beta=0.0):
    _lib.call("vpu_colsum", ptr(inp), ld, ptr(out), ptr(part), rows, Cdim, beta, code_of(inp[0] if isinstance(inp, tuple) else inp),
              _stream())


def softmax_fwd(S, lds, P, ldp, rows, ncols):
    _lib.call("vpu_softmax_fwd", ptr(S), lds, ptr(P), ldp, rows, ncols, code_of(P), _stream())


def softmax_bwd(P, ldp, dP, lddp, dS, rows, ncols, scale):
    _lib.call("vpu_softmax_bwd", ptr(P), ldp, ptr(dP), lddp, ptr(dS), rows, ncols, scale, code_of(P), _stream())


def l2norm_fwd(x, y, inv, rows, Cdim):
    _lib.call("vpu_l2norm_fwd", ptr(x), ptr(y), ptr(inv), rows, Cdim, code_of(x), _stream())


def l2norm_bwd(dy, y, inv, dx, rows, Cdim):
    _lib.call("vpu_l2norm_bwd", ptr(dy), ptr(y), ptr(inv), ptr(dx), rows, Cdim, code_of(y), _stream())


def attn_fwd(q, k, v, out, lse, nb, H, n, hd, ld, ldo, scale):
    _lib.call("vpu_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), nb, H, n, hd, ld, ldo, scale, _stream())


def xattn_fwd(q, k, v, out, lse, nb, H, nq, nk, hd, ldq, ldk, ldo, scale):
    _lib.call("vpu_xattn_fwd", ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), nb, H, nq, nk, hd, ldq, ldk, ldo, scale,
              _stream())


def xattn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, nb, H, nq, nk, hd, ldq, ldk, ldo, ldgq, ldgk, scale):
    _lib.call("vpu_xattn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dq), ptr(dk),
              ptr(dv), nb, H, nq, nk, hd, ldq, ldk, ldo, ldgq, ldgk, scale, _stream())


def xattn_fwd_split(q, k, v, out, lse, nb, H, nq, nk, hd, ldq, ldk, ldo, scale, qdiv, kdiv):
    """vpu_xattn_fwd_split: entry b reads its queries from entry b // qdiv, its keys / values from entry b // kdiv (include/vpu_hip.h)."""
    _lib.call("vpu_xattn_fwd_split", ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), nb, H, nq, nk, hd, ldq, ldk, ldo, scale,
              int(qdiv), int(kdiv), _stream())


def xattn_bwd_split(q, k, v, o, d_o, lse, delta, dq, dk, dv, nb, H, nq, nk, hd, ldq, ldk, ldo, ldgq, ldgk, scale, qdiv, kdiv):
    _lib.call("vpu_xattn_bwd_split", ptr(q), ptr(k), ptr(v), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dq), ptr(dk),
              ptr(dv), nb, H, nq, nk, hd, ldq, ldk, ldo, ldgq, ldgk, scale, int(qdiv), int(kdiv), _stream())


def attn_combine(o_s, lse_s, out, lse, nb, H, nq, hd, S, ld_s, ldo):
    _lib.call("vpu_attn_combine", ptr(o_s), ptr(lse_s), ptr(out), ptr(lse), nb, H, nq, hd, S, ld_s, ldo, _stream())


def sum_groups(inp, out, G, S, n):
    """out[g][i] = sum_s inp[g][s][i] (fp32 sum in order)."""
    _lib.call("vpu_sum_groups", ptr(inp), ptr(out), G, S, n, code_of(inp), _stream())


def attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, nb, H, n, hd, ld, ldo, ldg, scale):
    _lib.call("vpu_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(d_o), ptr(lse), ptr(delta), ptr(dq), ptr(dk),
              ptr(dv), nb, H, n, hd, ld, ldo, ldg, scale, _stream())


def add_bcast(a, b, out, n, period):
    _lib.call("vpu_add_bcast", ptr(a), ptr(b), ptr(out), n, period, code_of(a), _stream())


def add4(a, b, c, d, out, n):
    _lib.call("vpu_add4", ptr(a), ptr(b), ptr(c), ptr(d), ptr(out), n, code_of(a), _stream())


def fanout_add(src, dsts, accum, n):
    """dsts[i] (+)= src for up to four tensors in one launch (accum[i]: add / overwrite)."""
    arr = (C.c_void_p * len(dsts))(*[ptr(t) for t in dsts])
    acc = (C.c_int32 * len(dsts))(*[int(bool(a)) for a in accum])
    _lib.call("vpu_fanout_add", ptr(src), arr, acc, len(dsts), n, code_of(src), _stream())


def cast2d(src, ld_src, dst, ld_dst, rows, cols, cols_pad=None):
    s = src[0] if isinstance(src, tuple) else src
    d = dst[0] if isinstance(dst, tuple) else dst
    _lib.call("vpu_cast2d", ptr(src), code_of(s), ld_src, ptr(dst), code_of(d), ld_dst, rows, cols,
              cols if cols_pad is None else cols_pad, _stream())


def cast2d_batched(jobs):
    """jobs: up to 8 dicts(src fp32 (tensor or (tensor, offset)), dst, ld_src, ld_dst, rows, cols[, cols_pad, src2, perm]):
    dst[map(r)][c] = src[r][c] (+ src2[r][c]) in dst's dtype, ONE launch; perm = (g, wg): raster -> window row order."""
    arr = (_lib.CastJob * len(jobs))()
    for j, q in zip(arr, jobs):
        d = q["dst"][0] if isinstance(q["dst"], tuple) else q["dst"]
        j.src, j.src2, j.dst = ptr(q["src"]), ptr(q.get("src2")), ptr(q["dst"])
        j.ld_src, j.ld_dst, j.rows = q["ld_src"], q["ld_dst"], q["rows"]
        j.cols, j.cols_pad, j.dst_dtype = q["cols"], q.get("cols_pad", q["cols"]), code_of(d)
        j.perm_g, j.perm_wg = q.get("perm", (0, 0))
    _lib.call("vpu_cast2d_batched", arr, len(jobs), _stream())


def act_bwd(dy, ld_dy, aux, ld_aux, dz, ld_dz, rows, cols, kind, dtype):
    _lib.call("vpu_act_bwd", ptr(dy), ld_dy, ptr(aux), ld_aux, ptr(dz), ld_dz, rows, cols, kind, dtype, _stream())


def sigmoid_to_channel(logits, out, B, HW, channels, channel):
    _lib.call("vpu_sigmoid_to_channel", ptr(logits), ptr(out), B, HW, channels, channel, _stream())


def debug_spin(sink, wgs, cycles):
    """diagnostic: `wgs` channel-sized workgroups spinning for ~`cycles` shader cycles on the current stream"""
    _lib.call("vpu_debug_spin", ptr(sink), int(wgs), int(cycles), _stream())


_drop_state = {}


def _drop_key(device):
    key = torch.device(device)
    if key.type == "cuda" and key.index is None:
        key = torch.device("cuda", torch.cuda.current_device())
    return key


def _drop_entry(key, seed=None):
    """state tensor of ``key``'s mask stream: int64 [call number, seed] in device memory (vpu_dropout_mask reads both)."""
    if key not in _drop_state:
        sd = int(torch.initial_seed() if seed is None else seed) & (2 ** 63 - 1)
        _drop_state[key] = torch.tensor([0, sd], dtype=torch.int64, device=key)
    return _drop_state[key]


def dropout_mask(B, channels, keep, device, seed=None):
    """Dropout2d mask [B, channels] fp32 (values 0 or 1 / keep) from vpu_dropout_mask: one launch, capturable; the stream of
    masks is a function of the seed (default: torch.initial_seed() at the first call on that device; ``dropout_seed`` /
    ``set_dropout_state`` change it, also for launches already captured in a hipGraph) and the call count."""
    key = _drop_key(device)
    state = _drop_entry(key, seed)
    out = torch.empty(B, channels, device=key, dtype=torch.float32)
    _lib.call("vpu_dropout_mask", ptr(out), B * channels, float(keep), 0, ptr(state), _stream())
    return out


def dropout_seed(device, seed):
    """Restarts the mask stream of ``device``: call count 0 and a new seed, both in place in device memory -- launches
    captured before follow."""
    set_dropout_state(device, {"seed": int(seed) & (2 ** 63 - 1), "calls": 0})


def dropout_state(device):
    """{"seed", "calls"} of ``device``'s mask stream (a device read: synchronises); part of FusedAdam.state_dict()."""
    st = _drop_entry(_drop_key(device)).cpu()
    return {"seed": int(st[1]), "calls": int(st[0])}


def set_dropout_state(device, d):
    st = _drop_entry(_drop_key(device))
    st.copy_(torch.tensor([int(d["calls"]), int(d["seed"]) & (2 ** 63 - 1)], dtype=torch.int64))


def zero_(t):
    """t <- 0 through vpu_fill_f32 (any dtype whose size in bytes is a multiple of 4 per buffer: zero bits are zero bits)."""
    nbytes = t.numel() * t.element_size()
    assert t.is_contiguous() and nbytes % 4 == 0
    _lib.call("vpu_fill_f32", ptr(t), 0.0, nbytes // 4, _stream())
    return t


def zero_ranges_(t, ranges):
    """t[off : off + n] <- 0 for every (off, n) of ``ranges`` (fp32 tensor, multiples of 4 floats), 160 ranges per launch."""
    assert t.dtype == torch.float32 and t.is_contiguous()
    for i in range(0, len(ranges), 160):
        chunk = ranges[i:i + 160]
        offs = (C.c_int64 * len(chunk))(*[int(o) for o, _ in chunk])
        lens = (C.c_int64 * len(chunk))(*[int(n) for _, n in chunk])
        _lib.call("vpu_fill_ranges_f32", ptr(t), offs, lens, len(chunk), 0.0, _stream())
    return t


def fill_f32(t, v, n=None):
    _lib.call("vpu_fill_f32", ptr(t), v, t.numel() if n is None else n, _stream())


def pue_encode(points, boxes, lut, out, out64, B, n, num_max, img, ld):
    _lib.call("vpu_pue_encode", ptr(points), ptr(boxes), ptr(lut), ptr(out), ptr(out64), B, n, num_max, img, ld,
              code_of(out), _stream())


def pue_scribble_rows(points, vec64, out, out64, B, n, num_max, img, ld):
    _lib.call("vpu_pue_scribble_rows", ptr(points), ptr(vec64), ptr(out), ptr(out64), B, n, num_max, img, ld, code_of(out),
              _stream())


def draw_polyline(curve_i32, disks, B, P, H, W):
    _lib.call("vpu_draw_polyline", ptr(curve_i32), ptr(disks), B, P, H, W, _stream())


def mask_bbox(prob, thr, pos_clicks=None):
    """{count, rmin, rmax, cmin, cmax} (int32 [B,5], on the device) of prob[b] > thr joined with the positive clicks
    ``pos_clicks`` (int32 [n,2] (row, col) device tensor or None); prob: fp32 [B,H,W] contiguous."""
    B, H, W = prob.shape
    out = torch.empty(B, 5, device=prob.device, dtype=torch.int32)
    n = 0 if pos_clicks is None else int(pos_clicks.shape[0])
    _lib.call("vpu_mask_bbox", ptr(prob), float(thr), ptr(pos_clicks) if n else None, n, ptr(out), B, H, W, _stream())
    return out


def error_masks(pred_u8, gt_u8, valid_u8):
    """uint8 [2,H,W] on the device: (false negatives, false positives) of pred against gt inside valid"""
    H, W = gt_u8.shape
    out = torch.empty(2, H, W, device=gt_u8.device, dtype=torch.uint8)
    _lib.call("vpu_error_masks", ptr(pred_u8), ptr(gt_u8), ptr(valid_u8), ptr(out), H, W, _stream())
    return out


def masked_argmax(dist, keep_u8):
    """per plane of dist [P,H,W]: packed (max of dist * keep, first raster index) keys, uint64 as int64 [P] on the device"""
    P, H, W = dist.shape
    out = torch.empty(P, device=dist.device, dtype=torch.int64)
    _lib.call("vpu_masked_argmax", ptr(dist), ptr(keep_u8), ptr(out), P, H, W, _stream())
    return out


def disk_maps(points, boxes, out, B, n, H, W, radius):
    _lib.call("vpu_disk_maps", ptr(points), ptr(boxes), ptr(out), B, n, H, W, float(radius), _stream())


def patch_im2col(image4, disks, cols, B, H, W, P, win_tokens, prenorm=False):
    """``prenorm``: the rgb planes are normalised already (vpu_patch_im2col_prenorm)."""
    _lib.call("vpu_patch_im2col_prenorm" if prenorm else "vpu_patch_im2col", ptr(image4), ptr(disks), ptr(cols), B, H, W, P,
              win_tokens, code_of(cols), _stream())


def window_permute(x, y, B, g, wg, Cdim, to_raster):
    _lib.call("vpu_window_permute", ptr(x), ptr(y), B, g, wg, Cdim, 1 if to_raster else 0, code_of(x), _stream())


def pixel_shuffle2(src, dst, bias, B, h, w, Cdim, inverse=False):
    _lib.call("vpu_pixel_shuffle2", ptr(src), ptr(dst), ptr(bias), B, h, w, Cdim, 1 if inverse else 0, code_of(src),
              _stream())


def pixel_shuffle2_gn_stats(src, dst, bias, stats, B, h, w, Cdim):
    """pixel_shuffle2 (depth-to-space + bias) that also leaves GroupNorm(1, C)'s statistics partials of its output in
    ``stats`` (float64 [B, groupnorm_nchunk(), 2]): groupnorm_apply then normalises without a statistics pass."""
    _lib.call("vpu_pixel_shuffle2_gn_stats", ptr(src), ptr(dst), ptr(bias), ptr(stats), B, h, w, Cdim, code_of(src), _stream())


def groupnorm_apply(x, w, b, y, mean, rstd, stats, B, HW, Cdim, eps, gelu):
    _lib.call("vpu_groupnorm_apply", ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), ptr(stats), B, HW, Cdim, eps,
              int(gelu), code_of(x), _stream())


def pixel_unshuffle2_sums(src, dst, B, h, w, Cdim):
    """dst = depth-to-space^-1(src) as pixel_shuffle2(inverse=True), plus the per-channel partial sums of src (the transposed
    convolution's bias gradient): returns (part fp32 [nblk, Cdim], nblk) for a batched column sum."""
    nblk = _lib.load().vpu_pixel_unshuffle2_nblk(Cdim)
    part = torch.empty(nblk, Cdim, device=dst.device, dtype=torch.float32)
    _lib.call("vpu_pixel_unshuffle2_sums", ptr(src), ptr(dst), ptr(part), B, h, w, Cdim, code_of(src), _stream())
    return part, nblk


def groupnorm_nchunk():
    return _lib.load().vpu_groupnorm_nchunk()


def groupnorm_fwd(x, w, b, y, mean, rstd, stats, B, HW, Cdim, eps, gelu):
    _lib.call("vpu_groupnorm_fwd", ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), ptr(stats), B, HW, Cdim, eps,
              int(gelu), code_of(x), _stream())


def groupnorm_bwd(dy, x, w, b, mean, rstd, dx, part, stats, B, HW, Cdim, gelu):
    _lib.call("vpu_groupnorm_bwd", ptr(dy), ptr(x), ptr(w), ptr(b), ptr(mean), ptr(rstd), ptr(dx), ptr(part),
              ptr(stats), B, HW, Cdim, int(gelu), code_of(x), _stream())


def bilinear_cl_fwd(inp, ld_in, out, ld_out, B, h, w, H, W, Cdim, dtype):
    _lib.call("vpu_bilinear_cl_fwd", ptr(inp), ld_in, ptr(out), ld_out, B, h, w, H, W, Cdim, dtype, _stream())


def bilinear_cl_bwd(dout, ld_out, din, ld_in, B, h, w, H, W, Cdim, dtype):
    _lib.call("vpu_bilinear_cl_bwd", ptr(dout), ld_out, ptr(din), ld_in, B, h, w, H, W, Cdim, dtype, _stream())


def upsum_relu(io, maps, B, H, W, Cdim, dtype):
    """io[B,H,W,C] = relu(io + sum_i bilinear(z_i)), in place; maps = [(z_i [B,h,w,C], h, w), ...] (at most 3)."""
    n = len(maps)
    zp = (C.c_void_p * 3)(*[ptr(m[0]) for m in maps])
    hs = (C.c_int32 * 3)(*[m[1] for m in maps])
    ws = (C.c_int32 * 3)(*[m[2] for m in maps])
    _lib.call("vpu_upsum_relu", ptr(io), zp, hs, ws, n, B, H, W, Cdim, dtype, _stream())


def gate_stats(Q, Kt, cg, argq, sg, argc, B, nq, N, Cdim):
    _lib.call("vpu_gate_stats", ptr(Q), ptr(Kt), ptr(cg), ptr(argq), ptr(sg), ptr(argc), B, nq, N, Cdim, code_of(Q),
              _stream())


def gate_apply(x, cg, sg, out, B, N, Cdim):
    _lib.call("vpu_gate_apply", ptr(x), ptr(cg), ptr(sg), ptr(out), B, N, Cdim, code_of(x), _stream())


def gate_bwd(dout, x, cg, argq, sg, argc, dx, accum, dQ, dK, part, B, nq, N, Cdim):
    _lib.call("vpu_gate_bwd", ptr(dout), ptr(x), ptr(cg), ptr(argq), ptr(sg), ptr(argc), ptr(dx), int(accum), ptr(dQ),
              ptr(dK), ptr(part), B, nq, N, Cdim, code_of(x), _stream())


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[ptr(t) for t in ts])


def gate_fwd_n(Qs, Ks, x, outs, cg, argq, sg, argc, B, nq, N, Cdim):
    """The n <= 3 gates of SimpleFPN against one x: statistics (one launch) + gated maps (one launch).  cg / argq [n, B, C],
    sg / argc [n, B, N]."""
    _lib.call("vpu_gate_fwd_n", _ptr_array(Qs), _ptr_array(Ks), ptr(x), _ptr_array(outs), ptr(cg), ptr(argq), ptr(sg), ptr(argc),
              len(Qs), B, nq, N, Cdim, code_of(x), _stream())


def gate_bwd_n(douts, x, cg, argq, sg, argc, dx, accum, dQs, dKs, part, B, nq, N, Cdim):
    _lib.call("vpu_gate_bwd_n", _ptr_array(douts), ptr(x), ptr(cg), ptr(argq), ptr(sg), ptr(argc), ptr(dx), int(accum),
              _ptr_array(dQs), _ptr_array(dKs), ptr(part), len(douts), B, nq, N, Cdim, code_of(x), _stream())


def convseg_fwd(x, w, bias, mask, out, rows, HW, Cdim):
    _lib.call("vpu_convseg_fwd", ptr(x), ptr(w), ptr(bias), ptr(mask), ptr(out), rows, HW, Cdim, code_of(x), _stream())


def convseg_bwd_nblk(rows):
    return _lib.load().vpu_convseg_bwd_nblk(rows)


def convseg_bwd(dout, x, w, mask, dx, accum, part, part_b, rows, HW, Cdim):
    _lib.call("vpu_convseg_bwd", ptr(dout), ptr(x), ptr(w), ptr(mask), ptr(dx), int(accum), ptr(part), ptr(part_b),
              rows, HW, Cdim, code_of(x), _stream())


def head_grad_fused(dfn, y, inv, dout, x, w, mask, dx, part, part_b, rows, HW, Cdim):
    _lib.call("vpu_head_grad_fused", ptr(dfn), ptr(y), ptr(inv), ptr(dout), ptr(x), ptr(w), ptr(mask), ptr(dx), ptr(part),
              ptr(part_b), rows, HW, Cdim, code_of(x), _stream())


def upsample_ac_fwd(inp, out, planes, h, w, H, W):
    _lib.call("vpu_upsample_ac_fwd", ptr(inp), ptr(out), planes, h, w, H, W, _stream())


def upsample_ac_bwd(dout, din, planes, h, w, H, W):
    _lib.call("vpu_upsample_ac_bwd", ptr(dout), ptr(din), planes, h, w, H, W, _stream())


def p2cl_fwd_bwd(prob, gt, slot_idx, override, loss_part, dprob, grad_scale, B, S, H, W):
    _lib.call("vpu_p2cl_fwd_bwd", ptr(prob), ptr(gt), ptr(slot_idx), ptr(override), ptr(loss_part), ptr(dprob),
              grad_scale, B, S, H, W, _stream())


def p2cl_up_fwd_bwd(sim_low, gt, slot_idx, override, loss_part, dsim_low, grad_scale, B, S, h, w, H, W):
    """loss_part fp32 [B, S]: per-plane sums (the kernel's per-band partials are summed here); ``loss_part`` None: the
    per-band partials [B * S, nband] themselves are returned (their total is all vpu_loss_finalize needs: no extra launch)."""
    nband = _lib.load().vpu_p2cl_up_nband(h, w)
    bands = torch.empty(B * S, nband, device=sim_low.device, dtype=torch.float32)
    _lib.call("vpu_p2cl_up_fwd_bwd", ptr(sim_low), ptr(gt), ptr(slot_idx), ptr(override), ptr(bands), ptr(dsim_low),
              grad_scale, B, S, h, w, H, W, _stream())
    if loss_part is None:
        return bands
    torch.sum(bands, dim=1, out=loss_part.view(B * S))
    return loss_part


def nfl_dice_fwd_bwd(logits, gt, sums, out, dlogits, w_nfl, w_dice, B, HW):
    """``sums``: float64 scratch of vpu_nfl_dice_scratch_doubles(B) elements, or None to allocate it here."""
    if sums is None:
        sums = torch.empty(_lib.load().vpu_nfl_dice_scratch_doubles(B), device=logits.device, dtype=torch.float64)
    _lib.call("vpu_nfl_dice_fwd_bwd", ptr(logits), ptr(gt), ptr(sums), ptr(out), ptr(dlogits), w_nfl, w_dice, B, HW,
              _stream())


def loss_finalize(out, part, B, npart, inv_count, w_nfl, w_dice, w_pcl, iter_weight, res):
    _lib.call("vpu_loss_finalize", ptr(out), ptr(part), B, npart, float(inv_count), w_nfl, w_dice, w_pcl, iter_weight,
              ptr(res), _stream())


def adam_step_groups(p, g, m, v, shadow, n, seg_end, seg_lr, seg_wd, nseg, b1, b2, eps, decoupled, step, grad_scale=1.0):
    _lib.call("vpu_adam_step_groups", ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), n, ptr(seg_end), ptr(seg_lr),
              ptr(seg_wd), nseg, b1, b2, eps, int(decoupled), step, grad_scale, _stream())


def adam_step_hyper(p, g, m, v, shadow, n, hyper, seg_end, seg_scale, seg_wd, nseg, b1, b2, eps, wd, decoupled):
    _lib.call("vpu_adam_step_hyper", ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), n, ptr(hyper), ptr(seg_end),
              ptr(seg_scale), ptr(seg_wd), nseg, b1, b2, eps, wd, int(decoupled), _stream())


def adam_step(p, g, m, v, shadow, n, lr, b1, b2, eps, wd, step, grad_scale=1.0):
    _lib.call("vpu_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), n, lr, b1, b2, eps, wd, step, grad_scale,
              _stream())
