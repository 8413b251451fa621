#!/usr/bin/env python
"""Modelled data-parallel step from ONE GPU's measurements (no multi-GPU box is available to the build; the driver's scaling
run is the measurement -- this is the estimate beside it).  Measured here: the GPU time at which the backward hands each
gradient range to the reducer (HIP events at the reports of a host-enqueued step with a stand-in reducer, 16 CUs reserved),
the time backward ends and the step's total.  Modelled: the reducer's buckets (>= 25 MB, as parallel.GradReducer forms them)
go through ONE queue of ring all-reduces, each taking 2 (N - 1) / N x bytes / busbw(bytes) with
busbw(bytes) = peak x bytes / (bytes + half) (half-bandwidth message size `half`); a bucket starts when it is complete and
the previous one has finished; the optimizer waits for the last one.  No interference between RCCL's kernels and the GEMMs
beyond the 16 reserved CUs is modelled (unknown).
usage: python tools/dp_timeline_model.py [vitb|vitl|vith] [batch] [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvpuformer_amd import ops                                                          # noqa: E402
from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses                        # noqa: E402
from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model   # noqa: E402
from pvpuformer_amd.optim import FusedAdam                                            # noqa: E402
from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs                       # noqa: E402

MODELS = {"vitb": dict(embed_dim=768, depth=12, num_heads=12, patch=16), "vitl": dict(embed_dim=1024, depth=24, num_heads=16, patch=16),
          "vith": dict(embed_dim=1280, depth=32, num_heads=16, patch=14)}


def model_exchange(reports, t_bwd_end, n, peak_gbs, half_mb, bucket_mb=25.0, wire_bytes=4):
    """reports: [(t_ms, elements)] in report order -> (time the last collective ends, ms; number of collectives)."""
    buckets, pend, t_ready = [], 0, 0.0
    for t, elems in reports:
        pend += elems
        t_ready = t
        if pend * 4 >= bucket_mb * 1e6:          # (the reducer counts fp32 elements, whatever the wire format)
            buckets.append((t_ready, pend * wire_bytes))
            pend = 0
    if pend:
        buckets.append((t_bwd_end, pend * wire_bytes))
    t = 0.0
    for ready, nbytes in buckets:
        bw = peak_gbs * 1e9 * nbytes / (nbytes + half_mb * 1e6)
        t = max(t, ready) + 2.0 * (n - 1) / n * nbytes / bw * 1e3
    return t, len(buckets)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vitb"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs(**MODELS[name])).to(dev)
    model.set_compute_dtype("bf16")
    model.train()
    eng = model._ensure_engine()
    eng.refresh_weights()
    opt = FusedAdam(model, lr=5e-5)
    b = synth_batch(B, 448, seed=100, device=dev)
    x = torch.cat([b["images"], torch.zeros(B, 1, 448, 448, device=dev)], 1).contiguous()

    class Red:
        reserve_cus = 16

        def __init__(self):
            self.ev = []

        def ready(self, lo, hi):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.ev.append((e, hi - lo))
    red = Red()
    runs = []
    for it in range(6):
        eng.grad_ready_hook = red.ready
        red.ev = []
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        torch.cuda.synchronize()
        e0.record()
        eng.zero_grad()
        inst, _ = eng.forward(x, b["points"], None, 0, None, training=True, materialize_aux=False)
        _, d_inst, d_sim = vpu_step_losses(inst, None, b["instances"], None, None, iter_weight=1.0, sim_low=eng.sim_low)
        ops.gemm_set_option("reserve_cus", 16)
        e1.record()
        eng.backward(d_inst, None, d_sim_low=d_sim)
        e2.record()
        ops.gemm_set_option("reserve_cus", 0)
        opt.step(grad_scale=1.0)
        e3.record()
        torch.cuda.synchronize()
        if it >= 3:
            runs.append(dict(t_bwd_start=e0.elapsed_time(e1), t_bwd_end=e0.elapsed_time(e2), t_step=e0.elapsed_time(e3),
                             reports=[(e0.elapsed_time(e), n) for e, n in red.ev]))
    eng.grad_ready_hook = None
    r = runs[-1]
    res = {"what": f"{name} 448 bs={B}/GPU bf16 step with a stand-in reducer on ONE MI355X (host-enqueued, HIP events): measured report "
                   "times, modelled ring all-reduce queue (see tools/dp_timeline_model.py); NOT a multi-GPU measurement",
           "ms_backward_starts": round(r["t_bwd_start"], 3), "ms_backward_ends": round(r["t_bwd_end"], 3), "ms_step": round(r["t_step"], 3),
           "reports_ms_and_MB": [(round(t, 3), round(n * 4 / 1e6, 1)) for t, n in r["reports"]], "modelled": []}
    adam = r["t_step"] - r["t_bwd_end"]
    for n in (2, 4, 8):
        for peak, half in ((150.0, 8.0), (300.0, 16.0), (400.0, 16.0)):
            for wire, wb in (("fp32", 4), ("bf16", 2)):
                t_last, nb = model_exchange(r["reports"], r["t_bwd_end"], n, peak, half, wire_bytes=wb)
                exposed = max(0.0, t_last - r["t_bwd_end"])
                step = r["t_bwd_end"] + exposed + adam + (0.37 if wire == "bf16" else 0.0)      # (+ the measured cost of the casts)
                res["modelled"].append({"n_gpus": n, "busbw_peak_GBs": peak, "half_bandwidth_MB": half, "wire": wire, "collectives": nb,
                                        "exposed_ms": round(exposed, 3), "ms_step": round(step, 3)})
    print(json.dumps(res))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
