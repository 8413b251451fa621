"""Calibration only: the library attention (torch.nn.functional.scaled_dot_product_attention: the flash / memory-efficient
kernels this torch build carries) on the shapes of the step's attention launches, forward and backward, beside this
repository's kernels (tools/op_bench.py measures those).  usage: python tools/sdpa_compare.py [reps]"""
import sys, torch
import torch.nn.functional as F
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda"
SHAPES = [("window 14x14 (9 of 12 blocks)", 48, 12, 196, 196, 64), ("global 28x28 (3 of 12 blocks)", 12, 12, 784, 784, 64),
          ("neck tokens -> image", 12, 8, 48, 784, 48), ("neck image -> tokens", 12, 8, 784, 48, 48)]
for name, nb, H, nq, nk, hd in SHAPES:
    q = torch.randn(nb, H, nq, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(nb, H, nk, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(nb, H, nk, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(nb, H, nq, hd, device=dev, dtype=torch.bfloat16)
    def fwd():
        return F.scaled_dot_product_attention(q, k, v)
    for _ in range(3):
        o = fwd(); o.backward(do)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(reps):
        with torch.no_grad():
            fwd()
    e1.record()
    torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / reps * 1e3
    outs = [fwd() for _ in range(reps)]
    torch.cuda.synchronize()
    e1.record()
    for o in outs:
        o.backward(do, retain_graph=False)
    e2.record()
    torch.cuda.synchronize()
    tb = e1.elapsed_time(e2) / reps * 1e3
    fl = 4.0 * nb * H * nq * nk * hd
    print(f"{name:32s} nb {nb:3d} H {H:2d} {nq:4d} x {nk:4d} hd {hd}: library forward {tf:7.1f} us ({fl / tf / 1e6:6.1f} TFLOP/s), "
          f"backward {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.1f} TFLOP/s)")
