"""Child process of tests/test_dist_gpu.py: one rank of a data-parallel job on ONE GPU box.

  nccl1  world size 1 on RCCL (backend "nccl"): the bucketed reducer forced on -- asynchronous all-reduces launched from the
         backward tape's markers, stream waits in finish(), the reserve-CUs knob -- must leave the gradients bit-identical
         to a step without a reducer (a SUM over one rank is the identity); the bf16 wire format within bf16 rounding.
  gloo2  world size 2, both ranks on cuda:0, gradients exchanged through gloo (RCCL refuses two ranks on one device):
         parameter + buffer broadcast makes the replicas identical, and 2 ranks x 1 sample give the gradient of 1 rank x
         2 samples (the losses are per-sample means averaged over the batch).
usage: python tests/dist_worker.py <mode> <rank> <world> <port> <outdir>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import vpu_oracle as vo                                                    # noqa: E402  (synthetic weights / batches only)
from test_api_cpu import TINY, make_model                                  # noqa: E402
from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses            # noqa: E402
from pvpuformer_amd.parallel import GradReducer, broadcast_parameters      # noqa: E402


def build(dtype):
    cfg = vo.make_cfg(**TINY)
    model = make_model(cfg).cuda()
    model.load_state_dict(vo.synth_state_dict(vo.param_shapes(cfg), seed=0), strict=True)
    model.set_compute_dtype(dtype)
    model.train()
    model.head.dropout_ratio = 0.0
    eng = model._ensure_engine()
    eng.refresh_weights()
    return cfg, model, eng


def step(eng, img4, pts, gt, red=None):
    eng.zero_grad()
    if red is not None:
        red.begin()
        eng.grad_ready_hook = red.ready
    inst, _ = eng.forward(img4, pts, None, 0, None, training=True, materialize_aux=False)
    _, d_inst, d_sim = vpu_step_losses(inst, None, gt, None, None, iter_weight=1.0, sim_low=eng.sim_low)
    eng.backward(d_inst, None, d_sim_low=d_sim)
    scale = red.finish() if red is not None else 1.0
    eng.grad_ready_hook = None
    torch.cuda.synchronize()
    return eng.gflat.clone() * scale


def chain_step(eng, img4, pts, gt, red, replays=2):
    """The same step replayed as a chain of hipGraphs (pvpuformer_amd/graphs.py): zero-grad + forward + losses in one graph,
    the backward cut at the reported gradient ranges, the reducer's collectives launched by the host between the segments."""
    from pvpuformer_amd.graphs import SegmentedBackward, capture
    held = {}

    def head_body():
        eng.zero_grad()
        inst, _ = eng.forward(img4, pts, None, 0, None, training=True, materialize_aux=False)
        _, held["d_inst"], held["d_sim"] = vpu_step_losses(inst, None, gt, None, None, iter_weight=1.0, sim_low=eng.sim_low)
    head = torch.cuda.CUDAGraph()
    with capture(head):
        head_body()
    red.begin()
    seg = SegmentedBackward.capture(eng, lambda: eng.backward(held["d_inst"], None, d_sim_low=held["d_sim"]), hook_owner=red,
                                    pool=head.pool())
    red.finish()
    assert eng.grad_ready_hook is None
    outs = []
    for _ in range(replays):
        head.replay()
        red.begin()
        seg.replay(red.ready)
        scale = red.finish()
        torch.cuda.synchronize()
        outs.append(eng.gflat.clone() * scale)
    n_graphs = sum(1 for g, _ in seg.segments if g is not None)
    n_ranges = sum(len(r) for _, r in seg.segments)
    return outs, n_graphs, n_ranges


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    if mode == "nccl1":
        from pvpuformer_amd.parallel import configure_rccl_env
        assert configure_rccl_env() is not None          # NCCL_MAX_NCHANNELS set before the communicator exists (as bench.py does)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        cfg, model, eng = build("bf16")
        b = vo.synth_batch(2, cfg["img"], seed=5)
        img4 = torch.cat([b["images"], torch.zeros(2, 1, cfg["img"], cfg["img"])], 1).cuda()
        pts, gt = b["points"].cuda(), b["instances"].cuda()
        plain = step(eng, img4, pts, gt)
        red = GradReducer(eng.gflat, bucket_bytes=1 << 20, force=True, reserve_cus=16)
        assert red.enabled and red.world == 1
        with_red = step(eng, img4, pts, gt, red)
        launched = list(red.launched)
        chain, n_graphs, n_ranges = chain_step(eng, img4, pts, gt, red)
        chain_launched = list(red.launched)
        # every block's weight gradients launched at its marker (no queue across blocks): ranges are reported in the middle
        # of backward, the chain is really cut there
        eng.group_wgrad = False
        plain_ng = step(eng, img4, pts, gt)
        eager_ng = step(eng, img4, pts, gt, red)
        launched_ng = list(red.launched)
        chain_ng, n_graphs_ng, n_ranges_ng = chain_step(eng, img4, pts, gt, red)
        chain_launched_ng = list(red.launched)
        eng.group_wgrad = True
        red16 = GradReducer(eng.gflat, bucket_bytes=1 << 20, force=True, wire="bf16", reserve_cus=0)
        with_bf16 = step(eng, img4, pts, gt, red16)
        again = step(eng, img4, pts, gt)               # the reserve knob is back to 0: same kernels as the first step
        # the reference-faithful training step under the reducer: captured passes (the reporting backward as a graph chain)
        # against host-enqueued ones, eight optimizer steps of 1-3 click iterations from the same seeds
        import random
        from pvpuformer_amd.isegm.engine.trainer import VPUTrainStep
        from pvpuformer_amd.optim import FusedAdam
        ts = {}
        for use_graph in (False, True):
            _, model_t, eng_t = build("bf16")
            red_t = GradReducer(eng_t.gflat, bucket_bytes=1 << 20, force=True, reserve_cus=16)
            st = VPUTrainStep(model_t, FusedAdam(model_t, lr=1e-4), red_t)
            st.use_graph = use_graph
            rng, np_rng = random.Random(3), np.random.RandomState(4)
            dev_batch = {k: v.cuda() for k, v in b.items()}
            buckets = []
            for _ in range(8):
                st.batch_forward(dev_batch, rng=rng, np_rng=np_rng)
                buckets.append(len(red_t.launched))
            torch.cuda.synchronize()
            ts[use_graph] = (eng_t.flat.clone().cpu().numpy(), sum(1 for v in st._passes.values() if v not in ("seen", False)),
                             buckets)
        assert ts[False][1] == 0 and ts[True][1] >= 2, (ts[False][1], ts[True][1])
        assert ts[False][2] == ts[True][2] and min(ts[True][2]) >= 3, (ts[False][2], ts[True][2])
        # the optimizer step started before the last collectives have ended (parallel.finish_and_step) == the plain sequence
        from pvpuformer_amd.parallel import finish_and_step
        fs = {}
        for wire in ("fp32", "bf16"):
            for split in (False, True):
                _, model_s, eng_s = build("bf16")
                red_s = GradReducer(eng_s.gflat, bucket_bytes=1 << 20, force=True, reserve_cus=16, wire=wire)
                opt_s = FusedAdam(model_s, lr=1e-3)
                cuts = []
                for _ in range(3):
                    eng_s.zero_grad()
                    red_s.begin()
                    eng_s.grad_ready_hook = red_s.ready
                    inst, _ = eng_s.forward(img4, pts, None, 0, None, training=True, materialize_aux=False)
                    _, d_inst, d_sim = vpu_step_losses(inst, None, gt, None, None, iter_weight=1.0, sim_low=eng_s.sim_low)
                    eng_s.backward(d_inst, None, d_sim_low=d_sim)
                    if split:
                        finish_and_step(red_s, opt_s, 0.5, keep_last=2)
                        cuts.append(int(red_s.last_split))
                        assert not red_s._works and not red_s._staged
                    else:
                        opt_s.step(grad_scale=red_s.finish() * 0.5)
                    eng_s.grad_ready_hook = None
                torch.cuda.synchronize()
                if split:       # the update really was cut in two: the front of the buffer waited for the last collectives
                    assert all(0 < c < eng_s.total for c in cuts), (cuts, eng_s.total)
                fs[(wire, split)] = (eng_s.flat.clone().cpu().numpy(), opt_s.m.clone().cpu().numpy(), opt_s.v.clone().cpu().numpy(),
                                     eng_s.shadow.float().cpu().numpy(), opt_s.step_count)
            for a, b_ in zip(fs[(wire, False)][:4], fs[(wire, True)][:4]):
                assert np.array_equal(a, b_), f"split optimizer step differs from the plain one ({wire} wire)"
        fs = {False: fs[("fp32", False)], True: fs[("fp32", True)]}
        assert fs[False][4] == fs[True][4] == 3
        np.savez(os.path.join(out, "nccl1.npz"), plain=plain.cpu().numpy(), with_red=with_red.cpu().numpy(),
                 ts_eager=ts[False][0], ts_graph=ts[True][0],
                 fs_plain_p=fs[False][0], fs_split_p=fs[True][0], fs_plain_m=fs[False][1], fs_split_m=fs[True][1],
                 fs_plain_v=fs[False][2], fs_split_v=fs[True][2], fs_plain_s=fs[False][3], fs_split_s=fs[True][3],
                 fs_total=np.asarray(eng_s.total),
                 with_bf16=with_bf16.cpu().numpy(), again=again.cpu().numpy(), launched=np.asarray(launched),
                 total=np.asarray(eng.total), chain0=chain[0].cpu().numpy(), chain1=chain[1].cpu().numpy(),
                 chain_launched=np.asarray(chain_launched), chain_graphs=np.asarray([n_graphs, n_ranges]),
                 plain_ng=plain_ng.cpu().numpy(), eager_ng=eager_ng.cpu().numpy(), chain_ng0=chain_ng[0].cpu().numpy(),
                 chain_ng1=chain_ng[1].cpu().numpy(), launched_ng=np.asarray(launched_ng),
                 chain_launched_ng=np.asarray(chain_launched_ng), chain_graphs_ng=np.asarray([n_graphs_ng, n_ranges_ng]))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cfg, model, eng = build("f32")
        buf = model.pe_layer.positional_encoding_gaussian_matrix
        buf.copy_(torch.full_like(buf, float(rank + 1)))
        if rank == 1:
            eng.flat.mul_(1.5)
        broadcast_parameters(eng.flat)
        for t in model.buffers():
            broadcast_parameters(t)
        eng.shadow_valid = False
        eng.refresh_weights()
        b = vo.synth_batch(2, cfg["img"], seed=5)
        img4 = torch.cat([b["images"], torch.zeros(2, 1, cfg["img"], cfg["img"])], 1).cuda()
        pts, gt = b["points"].cuda(), b["instances"].cuda()
        red = GradReducer(eng.gflat, bucket_bytes=256 << 10)
        mine = step(eng, img4[rank:rank + 1].contiguous(), pts[rank:rank + 1].contiguous(), gt[rank:rank + 1].contiguous(), red)
        res = dict(flat=eng.flat.cpu().numpy(), buf=buf.cpu().numpy(), mine=mine.cpu().numpy(),
                   launched=np.asarray(red.launched))
        chain, _, _ = chain_step(eng, img4[rank:rank + 1].contiguous(), pts[rank:rank + 1].contiguous(),
                                 gt[rank:rank + 1].contiguous(), red, replays=1)
        res["chain"] = chain[0].cpu().numpy()
        if rank == 0:
            res["full"] = step(eng, img4, pts, gt).cpu().numpy()
        np.savez(os.path.join(out, f"gloo2_rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()
    print("WORKER-OK", mode, rank)


if __name__ == "__main__":
    main()
