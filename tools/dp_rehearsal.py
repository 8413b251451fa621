#!/usr/bin/env python
"""Data-parallel rehearsal on ONE GPU: the ViT-B B = 12 bench step with the bucketed gradient reducer forced on over a
world-size-1 RCCL communicator (backend "nccl"), against the same step without a reducer.  What this CAN show on one GPU:
the engine-side cost of running under a reducer (queues flushed at every tape marker, no riding weight gradients,
`reserve_cus` CUs kept out of the persistent GEMM grids, the all-reduce calls and stream waits) and, from a kernel trace of
this script, whether RCCL's kernels run BESIDE the GEMMs or behind them.  What it cannot show: xGMI traffic -- no
multi-GPU box is available to the build, the driver's scaling run is the first.
usage: python tools/dp_rehearsal.py [steps] [out.json]"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvpuformer_amd.parallel import GradReducer, configure_rccl_env, finish_and_step  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    channels = configure_rccl_env()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses
    from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model
    from pvpuformer_amd.optim import FusedAdam
    from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs()).to(dev)
    model.set_compute_dtype("bf16")
    model.train()
    eng = model._ensure_engine()
    eng.refresh_weights()
    opt = FusedAdam(model, lr=5e-5)
    B = int(os.environ.get("DP_BATCH", "12"))      # 12: bench.py's per-GPU batch; 4: the reference recipe's 32 images on 8 GPUs
    b = synth_batch(B, 448, seed=100, device=dev)
    x = torch.cat([b["images"], torch.zeros(B, 1, 448, 448, device=dev)], 1).contiguous()

    host_ms = [0.0]

    def run(red):
        eng.grad_ready_hook = red.ready if red is not None else None
        def one():
            eng.zero_grad()
            inst, _ = eng.forward(x, b["points"], None, 0, None, training=True, materialize_aux=False)
            _, d_inst, d_sim = vpu_step_losses(inst, None, b["instances"], None, None, iter_weight=1.0, sim_low=eng.sim_low)
            if red is not None:
                red.begin()
            eng.backward(d_inst, None, d_sim_low=d_sim)
            finish_and_step(red, opt)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        host_ms[0] = (time.perf_counter() - t0) / steps * 1e3      # the host's enqueue time per step (it does not block)
        torch.cuda.synchronize()
        eng.grad_ready_hook = None
        return (time.perf_counter() - t0) / steps * 1e3

    def run_chain(red):
        """the same step replayed as bench.py replays it at N > 1: forward graph + backward segments cut at the reported
        ranges, the reducer's collectives launched by the host between two segments, Adam host-enqueued"""
        from pvpuformer_amd.graphs import SegmentedBackward, capture
        held = {}

        def head_body():
            eng.zero_grad()
            inst, _ = eng.forward(x, b["points"], None, 0, None, training=True, materialize_aux=False)
            _, held["d_inst"], held["d_sim"] = vpu_step_losses(inst, None, b["instances"], None, None, iter_weight=1.0, sim_low=eng.sim_low)
        head = torch.cuda.CUDAGraph()
        with capture(head):
            head_body()
        red.begin()
        seg = SegmentedBackward.capture(eng, lambda: eng.backward(held["d_inst"], None, d_sim_low=held["d_sim"]),
                                        hook_owner=red, pool=head.pool())
        red.finish()

        def one():
            head.replay()
            red.begin()
            seg.replay(red.ready)
            finish_and_step(red, opt)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt / steps * 1e3, t_host / steps * 1e3, sum(1 for g, _ in seg.segments if g is not None)

    only = os.environ.get("DP_ONLY")        # "none" / "reducer": ONE configuration, eager, for a kernel-trace A/B (tools/dp_stats_ab.sh)
    if only:
        red1 = GradReducer(eng.gflat, force=True, reserve_cus=int(os.environ.get("DP_RESERVE", "16"))) if only == "reducer" else None
        print(only, round(run(red1), 3), "ms per step")
        dist.barrier()
        dist.destroy_process_group()
        return
    res = {"what": f"ViT-B 448 bs={B} bf16 training step on ONE MI355X, eager launch, RCCL world size 1", "steps": steps,
           "NCCL_MAX_NCHANNELS": channels, "host_threads_allowed": len(os.sched_getaffinity(0))}
    res["ms_no_reducer"] = round(run(None), 3)
    red = GradReducer(eng.gflat, force=True, reserve_cus=16)
    res["ms_reducer_fp32_wire_reserve16"] = round(run(red), 3)
    res["eager_host_ms_per_step_reserve16"] = round(host_ms[0], 3)
    res["collectives_per_step"] = len(red.launched)          # (reset by begin(): the last step's)
    res["rccl_kernels_at_world_size_1"] = "none: RCCL returns from an in-place all-reduce over one rank without launching (kernel trace: 0 nccl kernels)"
    res["ms_reducer_bf16_wire_reserve16"] = round(run(GradReducer(eng.gflat, force=True, wire="bf16", reserve_cus=16)), 3)
    res["ms_reducer_fp32_wire_reserve0"] = round(run(GradReducer(eng.gflat, force=True, reserve_cus=0)), 3)
    res["ms_reducer_fp32_wire_reserve0_again"] = round(run(GradReducer(eng.gflat, force=True, reserve_cus=0)), 3)
    ms, host, nseg = run_chain(GradReducer(eng.gflat, force=True, reserve_cus=16))
    res["ms_reducer_fp32_wire_reserve16_graph_chain"] = round(ms, 3)
    res["graph_chain_host_ms_per_step"] = round(host, 3)
    res["graph_chain_backward_segments"] = nseg
    ms0, host0, nseg0 = run_chain(GradReducer(eng.gflat, force=True, reserve_cus=0))
    res["ms_reducer_fp32_wire_reserve0_graph_chain"] = round(ms0, 3)
    res["ms_no_reducer_again"] = round(run(None), 3)
    print(json.dumps(res))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(res, f, indent=1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
