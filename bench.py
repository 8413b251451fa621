#!/usr/bin/env python
"""Headline benchmark: 448x448 images/sec, forward+backward(+Adam), ViT-B VPUFormer, bs 12 per GPU, click prompts,
bf16 MFMA, synthetic data, random-init weights (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no WORLD_SIZE in the environment: bench.py starts the N ranks itself, as fresh
                                         child processes, BEFORE anything here has touched the GPU or RCCL -- the
                                         reference's launch is torch.distributed.launch, train.py:18,74)

One "step" = one ISTrainer.batch_forward iteration with num_iters fixed to 1 (isegm/engine/trainer.py:310-491) on a
resident batch: zero grads, forward (train mode, Dropout2d on), NFL + Dice + P2CL losses, backward, bucketed gradient
all-reduce (N > 1), fused Adam.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMG = {"vitb": 510.16e9, "vitl": 1613.59e9, "vith": 4265.17e9}  # fwd+bwd, SURVEY.md section 8d [probe]
MODELS = {"vitb": dict(embed_dim=768, depth=12, num_heads=12, patch=16),
          "vitl": dict(embed_dim=1024, depth=24, num_heads=16, patch=16),
          "vith": dict(embed_dim=1280, depth=32, num_heads=16, patch=14)}
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md
NAMES = {"vitb": "ViT-B", "vitl": "ViT-L", "vith": "ViT-H"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=12, help="per-GPU batch (README.md:49-54: 12 on one GPU)")
    ap.add_argument("--model", default="vitb", choices=sorted(MODELS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="NO GPU, NO model: only the launcher / rendezvous / bucketed reducer / JSON plumbing of an N-rank "
                         "run over gloo on the CPU with a stand-in gradient buffer (tests/test_bench_launch_cpu.py). "
                         "Prints value null: never a measurement")
    return ap.parse_args()


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: N fresh child processes through torch.distributed.run (one rank
    per GPU, 127.0.0.1 rendezvous on a free port).  Called before this process has made any HIP / RCCL call -- it never
    makes one: it waits for the children and exits with their code."""
    import socket
    import subprocess
    if not args.rehearse_launch:
        import __graft_entry__ as ge
        ge.build(lab=False)            # hipcc only (no GPU): the ranks find a fresh library instead of racing to build it
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write(f"bench.py: no WORLD_SIZE in the environment -- starting {args.gpus} ranks: {' '.join(cmd)}\n")
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def cpu_baseline(batch_size, gpu_batch=12, keep_ref=None):
    """The CPU oracle (torch-CPU fp32 restatement of the reference, oracle/vpu_oracle.py) timed on this box's host
    cores on a bounded sample: same step definition (fwd + NFL/Dice/P2CL + bwd), the FIRST ``batch_size`` images of the
    batch rank 0's GPU step runs on (pvpuformer_amd.synth.synth_batch(gpu_batch, 448, seed=100), same seed path).
    ``keep_ref``: a dict that receives the oracle's inputs, weights and mask logits for the parity check below."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vpu_oracle as vo
    from pvpuformer_amd.synth import synth_batch
    # torch-CPU scales badly past a few dozen threads on this model (256 threads: 233 s/step measured in round 1):
    # use at most 32 and say so.  The sample is bounded to ~30 s: stop as soon as the budget is spent.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = vo.make_cfg()
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in
          vo.synth_state_dict(vo.param_shapes(cfg), seed=0).items()}
    full = synth_batch(gpu_batch, cfg["img"], seed=100, device="cpu")
    b = {k: v[:batch_size].contiguous() for k, v in full.items()}
    img4 = torch.cat([b["images"], torch.zeros(batch_size, 1, cfg["img"], cfg["img"])], 1)
    ed = vo.ed_mask_label(b["instances"])
    # 3 warm-up + >= 5 timed steps; bounded to ~30 s of CPU work (~1.4 s per step at 2 images): the warm-up is
    # cut short, then the timed steps, if the budget runs out
    WARM, TIMED, BUDGET = 3, 5, 30.0
    times, t_begin = [], time.perf_counter()
    for it in range(WARM + TIMED):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = vo.vpu_forward(sd, cfg, img4, b["points"])
        total, _ = vo.step_loss(out, b["instances"], ed)
        total.backward()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_begin > BUDGET:
            break
    if keep_ref is not None:
        keep_ref.update(sd={k: v.detach() for k, v in sd.items()}, img4=img4, points=b["points"],
                        logits=out["instances"].detach(), loss=float(total.detach()))
    warm = min(WARM, max(0, len(times) - 1))
    timed = times[warm:]
    t = sum(timed) / len(timed)
    return {"value": round(batch_size / t, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/vpu_oracle.py fp32 torch-CPU on {cores} of {os.cpu_count()} host threads (more is slower: 256 "
                      f"threads took 233 s/step), ViT-B/448, the first {batch_size} images of the GPU step's batch "
                      f"(synth_batch seed 100) per step, fwd+bwd+losses, mean of {len(timed)} timed "
                      f"step(s) after {warm} warm-up ({t:.2f} s/step, best {min(timed):.2f})"}


def parity_vs_oracle(ref, model_kwargs, dev):
    """Mask logits of the HIP path against the CPU oracle's on the SAME weights (hash-generated, loaded into a second model
    instance) and the same images -- the first images of the timed batch --, in the exact-fp32 engine mode and in the bf16
    mode the timed step runs in: max |difference| / max |reference logit| (north_star's 1e-3 is the fp32 figure; the bf16
    mode is the reference's --amp analogue).  The oracle is the checker here, nothing of it is timed."""
    from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model
    m = VitMultiGaussianVector_ed_Model(**model_kwargs).to(dev)
    m.load_state_dict(ref["sd"], strict=True)
    m.eval()
    out = {}
    for dtype in ("f32", "bf16"):
        m.set_compute_dtype(dtype)
        eng = m._ensure_engine()
        eng.refresh_weights()
        inst, _ = eng.forward(ref["img4"].to(dev), ref["points"].to(dev), None, 0, None, training=False, materialize_aux=False)
        err = (inst.float().cpu() - ref["logits"]).abs().max().item() / ref["logits"].abs().max().item()
        out["fp32_rel" if dtype == "f32" else "bf16_rel"] = float(f"{err:.3e}")
    out["what"] = (f"max |mask logit - oracle| / max |oracle logit|, ViT-B/448, {ref['img4'].shape[0]} images of the timed batch, "
                   f"oracle weights loaded into the HIP model (eval mode); the timed step computes in bf16")
    del m
    torch.cuda.empty_cache()
    return out


def rehearse_launch(args, rank, world):
    """``--rehearse-launch``: the N-rank plumbing of this file WITHOUT a GPU and WITHOUT the model -- launcher, gloo
    rendezvous, parameter broadcast, the bucketed reducer fed tail-first ranges of a stand-in gradient buffer, barrier +
    max-over-ranks timing, the data-parallel diagnostics and the JSON line.  ``value`` is null: nothing is measured."""
    import torch.distributed as dist
    from pvpuformer_amd.parallel import GradReducer, broadcast_parameters
    if world > 1:
        dist.init_process_group("gloo")
    n = 1 << 20
    flat = torch.full((n,), float(rank + 1))
    broadcast_parameters(flat)
    assert float(flat[0]) == 1.0
    g = torch.zeros(n)
    red = GradReducer(g, bucket_bytes=1 << 20)
    cuts = [n, n * 3 // 4, n // 2, n // 4, n // 8, 0]

    def step(diag=False):
        g.fill_(float(rank + 1))
        red.trace = [] if diag else None
        red.begin()
        for hi, lo in zip(cuts, cuts[1:]):
            red.ready(lo, hi)
        red.mark("bwd_end")
        scale = red.finish()
        red.mark("step_end")
        return scale

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scale = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ok = bool(torch.all(g == world * (world + 1) / 2)) and scale == 1.0 / world
    step(diag=True)
    dp = dp_report(red, world, "gloo", "eager (rehearsal)")
    if rank == 0:
        print(json.dumps({"metric": "448x448 images/sec fwd+bwd, ViT-B VPUFormer", "value": None, "unit": "images/sec",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": args.dtype, "data": "none (launch rehearsal on the CPU: no model, no GPU)",
                          "config": {"workload": "launch rehearsal", "parallelism": f"dp{world}", "reduced_ok": ok},
                          "dp": dp}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py --rehearse-launch: the reduced gradient is wrong")


def dp_report(red, world, backend, launch_mode):
    """The data-parallel fields of the JSON line (N > 1): what the run looked like from the inside, so that the first
    multi-GPU measurement comes with an explanation -- ranks the communicator saw, bytes per step on the wire, where each
    bucket's collective was launched and completed relative to the end of backward (exposed communication), host time
    spent waiting, launch mode, the RCCL channel cap / reserved CUs in force."""
    import torch.distributed as dist
    d = red.summary() if red.trace is not None else {}
    ranks = None
    if world > 1 and dist.is_initialized():
        one = torch.ones(1, device=red.g.device)
        dist.all_reduce(one)
        ranks = int(one.item())
    d.update({"backend": backend, "ranks_seen_by_all_reduce": ranks, "launch_mode": launch_mode,
              "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
              "VPU_DIST_RESERVE_CUS": os.environ.get("VPU_DIST_RESERVE_CUS", "16 (default)"),
              "split_adam": int(os.environ.get("VPU_DIST_SPLIT_ADAM", "0"))})
    try:
        d["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:
        d["rccl_version"] = None
    return d


def pmc_traffic(kernel, args):
    """HBM bytes per launch of the dominant GEMM variant from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 on gfx950 +
    WRITE_SIZE, separate passes: tools/run_pmc_bench.sh + tools/pmc_traffic.py -> profiles/rNN_pmc_traffic.json).  Counters
    cannot be read from inside this process, so the number is the one measured for the default workload; any other
    workload reports null."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))     # the newest round's
    if not found or args.model != "vitb" or args.batch != 12 or args.dtype != "bf16":
        return None, None
    path = found[-1]
    tab = json.load(open(path))
    v = tab.get(kernel)
    if not v:
        return None, None
    n, tot = v["launches"], v["launches"] * (v["read_bytes_per_launch"] + v["write_bytes_per_launch"])
    # (a LOOKUP of the builder-recorded PMC passes, not a measurement of this run: the JSON line says so)
    return round(tot / n), f"profiles/{os.path.basename(path)} (builder-recorded rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not this run)"


class GemmProbe:
    """HIP-event timing of every GEMM launch of ONE extra (untimed) step, grouped by kernel instantiation."""

    def __init__(self, ops):
        self.ops, self.orig, self.orig_grouped, self.rec, self.shapes = ops, ops.gemm, ops.gemm_grouped, [], []

    def __enter__(self):
        def wrapped(A, B, C, M, N, K, lda, ldb, ldc, dtype, transA=False, transB=False, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig(A, B, C, M, N, K, lda, ldb, ldc, dtype, transA=transA, transB=transB, **kw)
            e1.record()
            name = self.ops.gemm_last_kernel()      # the instantiation the C dispatcher picked, as rocprofv3 prints it
            self.rec.append((name, 2.0 * M * N * K * kw.get("batch", 1), e0, e1))
            self.shapes.append((int(transA), int(transB), M, N, K, kw.get("batch", 1), kw.get("flags", 0)))
        def wrapped_grouped(problems):   # one launch for several problems (the queued weight gradients)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig_grouped(problems)
            e1.record()
            (a0, k0) = problems[0]
            fl = sum(2.0 * a[3] * a[4] * a[5] for a, _ in problems)
            self.rec.append((self.ops.gemm_last_kernel(), fl, e0, e1))
            self.shapes.append(("grouped", len(problems), sum(a[3] for a, _ in problems), a0[4], a0[5], 1, 0))
        self.ops.gemm = wrapped
        self.ops.gemm_grouped = wrapped_grouped
        return self

    def __exit__(self, *a):
        self.ops.gemm = self.orig
        self.ops.gemm_grouped = self.orig_grouped

    def summary(self):
        torch.cuda.synchronize()
        agg, by_shape = {}, {}
        for (name, fl, e0, e1), shp in zip(self.rec, self.shapes):
            sec = e0.elapsed_time(e1) * 1e-3
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += fl; a[1] += sec; a[2] += 1
            s = by_shape.setdefault(shp, [0.0, 0.0, 0])
            s[0] += fl; s[1] += sec; s[2] += 1
        dump = os.environ.get("VPU_GEMM_SHAPES")
        if dump:
            rows = sorted(by_shape.items(), key=lambda kv: -kv[1][1])
            with open(dump, "w") as f:
                f.write("tA tB M N K batch flags | launches total_ms TFLOP/s\n")
                for shp, (fl, sec, cnt) in rows:
                    f.write(f"{shp} | {cnt} {sec * 1e3:.3f} {fl / sec / 1e12:.1f}\n")
        return agg


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))    # (this process has not touched the GPU and never will)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.rehearse_launch:
        return rehearse_launch(args, rank, world)
    # the extension is (re)built BEFORE anything touches the GPU or RCCL: a stale .so means 8 hipcc children, which must
    # neither inherit a profiler preload nor keep the other ranks waiting inside a collective.  Rank 0 builds, the others
    # poll the file's freshness.
    import __graft_entry__ as ge
    srcdir = os.path.join(ROOT, "pvpuformer_amd", "csrc")
    lib = os.path.join(ROOT, "pvpuformer_amd", "libvpu_hip.so")
    failed = lib + ".build_failed"          # rank 0's verdict for the ranks that poll (build.sh renames the finished
    if rank == 0:                           # library into place, so a fresh file is a complete file)
        if os.path.exists(failed):
            os.remove(failed)
        try:
            ge.build(lab=False)      # (the laboratory library is test infrastructure: never built on the way to a measurement)
        except BaseException as e:
            with open(failed, "w") as f:
                f.write(f"{type(e).__name__}: {e}\n")
            raise
    else:
        t_wait = time.time()
        while ge._stale(lib, srcdir):
            if os.path.exists(failed) and os.path.getmtime(failed) >= t_wait - 5:
                raise SystemExit("bench.py: rank 0 failed to build libvpu_hip.so: " + open(failed).read().strip())
            if time.time() - t_wait > 900:
                raise SystemExit("bench.py: rank 0 did not finish building libvpu_hip.so within 15 minutes")
            time.sleep(1.0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists for the product)")
    # rehearsal on a one-GPU box (never used by the driver): VPU_DIST_SHARE_GPU=1 puts every rank on cuda:0 and
    # VPU_DIST_BACKEND=gloo exchanges the gradients through the host -- RCCL refuses two ranks on one device
    if os.environ.get("VPU_DIST_SHARE_GPU", "0") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        backend = os.environ.get("VPU_DIST_BACKEND", "nccl")
        if backend == "nccl":
            from pvpuformer_amd.parallel import configure_rccl_env
            configure_rccl_env()      # channel count = the CUs the persistent GEMM grids leave free (before the communicator exists)
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from pvpuformer_amd import ops
    from pvpuformer_amd.isegm.engine.trainer import vpu_step_losses
    from pvpuformer_amd.isegm.model.is_vpu_model import VitMultiGaussianVector_ed_Model
    from pvpuformer_amd.optim import FusedAdam
    from pvpuformer_amd.parallel import GradReducer, broadcast_parameters, finish_and_step
    from pvpuformer_amd.synth import synth_batch, vitb_model_kwargs

    mk = MODELS[args.model]
    torch.manual_seed(0)
    model = VitMultiGaussianVector_ed_Model(**vitb_model_kwargs(embed_dim=mk["embed_dim"], depth=mk["depth"],
                                                                num_heads=mk["num_heads"], patch=mk["patch"])).to(dev)
    model.set_compute_dtype(args.dtype)
    model.train()
    eng = model._ensure_engine()
    broadcast_parameters(eng.flat)
    eng.refresh_weights()
    # Single GPU: the whole step (zero-grad, forward, losses, backward, Adam: ~550 kernel launches) is captured once and
    # replayed as ONE hipGraph -- the same kernels in the same order, none of the host's ~10 ms of Python / ctypes per
    # step (measured equal on an idle 256-core box, 858-863 images/s either way: the step is GPU-bound; on a box with a
    # small CPU share the eager loop becomes host-bound, the replay does not).  VPU_BENCH_GRAPH=0 forces the eager loop;
    # a failed capture falls back to it and says so in the JSON line.  Multi-GPU runs launch their RCCL collectives from
    # the backward tape: there the step is replayed as a CHAIN of graphs -- zero-grad + forward + losses, then the backward
    # cut wherever the engine reports a finished gradient range (pvpuformer_amd/graphs.py), the reducer's collectives
    # launched by the host between two segments, Adam host-enqueued -- ~25 launches per step instead of ~530.
    # (N > 1 default, round 6: the CHAIN.  Decided on what one GPU can show -- the forced reducer at world size 1 with the
    # process pinned to 32 host threads, what one of 8 ranks gets (tools/dp_mode_job.sh): eager 15.8 ms per
    # step with 13.9 ms of it host enqueue time -- host-bound before any collective -- against 14.3 ms and 0.63 ms of host time for
    # the chain at B = 12; level at B = 4 (8.66 / 8.68 ms) with 7.6 against 0.64 ms of host time.  The chain has run under RCCL at
    # world size 1 and over two gloo ranks; a rank whose capture fails stays host-enqueued (same kernels, same collectives, same
    # order).  VPU_BENCH_DP_GRAPH=0 selects the eager loop.)
    use_graph = world == 1 and os.environ.get("VPU_BENCH_GRAPH", "1") != "0"
    use_chain = world > 1 and os.environ.get("VPU_BENCH_DP_GRAPH", "1") != "0"
    opt = FusedAdam(model, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, capturable=use_graph)
    red = GradReducer(eng.gflat, wire=os.environ.get("VPU_DIST_WIRE", "fp32"))   # VPU_DIST_WIRE=bf16: half the bytes per link
    eng.grad_ready_hook = red.ready if red.enabled else None
    # VPU_ADAM_OVERLAP=1: the optimizer step runs range by range on a second stream while backward continues
    # (OverlappedAdam; also zeroes the gradients it has consumed).  Off by default: bit-identical results, but measured
    # slower on one GPU (17.3-17.5 ms per step against 17.0 with one Adam launch after backward, whatever the Adam grid:
    # the HBM-bound stream takes more from the GEMMs it runs beside than its own 0.65 ms)
    overlap = None
    if not use_graph and os.environ.get("VPU_ADAM_OVERLAP", "0") == "1":
        from pvpuformer_amd.optim import OverlappedAdam
        overlap = OverlappedAdam(opt, eng, red)
        eng.zero_grad()

    B = args.batch
    batch = synth_batch(B, 448, seed=100 + rank, device=dev)   # each rank its own shard of the global batch
    image4 = torch.cat([batch["images"], torch.zeros(B, 1, 448, 448, device=dev)], 1).contiguous()  # trainer.py:324,384
    gt, points = batch["instances"], batch["points"]
    keep = 1.0 - model.head.dropout_ratio
    last = {}

    mode = {"overlap": overlap is not None}

    def step_body():
        if not mode["overlap"]:
            eng.zero_grad(lazy=True)        # (one backward follows: single-writer gradients are written, not accumulated)
        mask = ops.dropout_mask(B, model.head.channels, keep, dev)
        inst, _ = eng.forward(image4, points, None, 0, mask, training=True, materialize_aux=False)
        losses, d_inst, d_sim = vpu_step_losses(inst, None, gt, None, None, iter_weight=1.0, sim_low=eng.sim_low)
        if mode["overlap"]:
            overlap.begin()
            eng.backward(d_inst, None, d_sim_low=d_sim)
            overlap.finish()
        else:
            red.begin()
            eng.backward(d_inst, None, d_sim_low=d_sim)
            finish_and_step(red, opt)       # (N > 1: Adam on the reduced part while the last collectives are on the wire)
        last["loss"] = losses["total"]

    graph = [None]
    chain = [None]                     # (head graph, SegmentedBackward) of a data-parallel step
    graph_note = [None]
    held = {}

    def head_body():
        eng.zero_grad(lazy=True)
        mask = ops.dropout_mask(B, model.head.channels, keep, dev)
        inst, _ = eng.forward(image4, points, None, 0, mask, training=True, materialize_aux=False)
        losses, held["d_inst"], held["d_sim"] = vpu_step_losses(inst, None, gt, None, None, iter_weight=1.0, sim_low=eng.sim_low)
        held["loss"] = losses["total"]

    def step():
        if use_graph:
            opt.prepare_step(1.0)      # step count, bias corrections, lr -> device scalars read by the captured Adam launch
        if graph[0] is not None:
            graph[0].replay()
        elif chain[0] is not None:
            chain[0][0].replay()
            red.begin()
            chain[0][1].replay(red.ready if red.enabled else None)
            finish_and_step(red, opt)
            last["loss"] = held["loss"]
        else:
            step_body()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if use_graph:
        if args.warmup == 0:
            step()                     # lazily created streams / workspaces / kernel attributes must exist before capture
            sync()
        opt.prepare_step(1.0)
        try:
            from pvpuformer_amd.graphs import capture
            g = torch.cuda.CUDAGraph()
            with capture(g, device=dev):
                step_body()
            opt.step_count -= 1        # capturing enqueues nothing: that step was not taken
            graph[0] = g
            step()                     # first replay (untimed): graph upload
            sync()
        except Exception as e:         # (a capture that fails leaves the eager loop: same kernels, host-enqueued)
            graph_note[0] = f"eager (hipGraph capture failed: {type(e).__name__}: {str(e)[:120]})"
            graph[0] = None
            torch.cuda.synchronize()
            eng.abort_pass()           # nothing the aborted capture queued may reach the eager steps' launches
    if use_chain and overlap is None:
        # every rank captures for itself (a capture enqueues nothing and launches no collective); a rank whose capture
        # fails stays host-enqueued: the same kernels and the same collectives in the same order as the replaying ranks
        from pvpuformer_amd.graphs import SegmentedBackward
        if args.warmup == 0:
            step()
            sync()
        try:
            head = torch.cuda.CUDAGraph()
            from pvpuformer_amd.graphs import capture
            with capture(head, device=dev):
                head_body()
            red.begin()                # (the GEMM grids of a data-parallel backward leave the reducer's CUs free)
            seg = SegmentedBackward.capture(eng, lambda: eng.backward(held["d_inst"], None, d_sim_low=held["d_sim"]),
                                            hook_owner=red if red.enabled else None, pool=head.pool())
            red.finish()
            chain[0] = (head, seg)
            graph_note[0] = (f"hipGraph chain: forward graph + {sum(1 for g, _ in seg.segments if g is not None)} backward "
                             f"segments cut at the {sum(len(r) for _, r in seg.segments)} reported gradient ranges, collectives between them")
        except Exception as e:
            graph_note[0] = f"eager (hipGraph capture failed: {type(e).__name__}: {str(e)[:120]})"
            chain[0] = None
            try:
                red.finish()
            except Exception:
                pass
            torch.cuda.synchronize()
            eng.abort_pass()
        if world > 1:
            dist.barrier()
        step()                         # first replay (untimed): graph upload
        sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0   # host time to ENQUEUE the steps (diagnostic: close to dt means launch-bound)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(last["loss"].item())
    value = world * B * args.steps / dt

    roof = None
    # one extra, UNTIMED step with a HIP-event pair around every GEMM launch.  The weight-gradient side stream is
    # switched off for it so that every launch is timed back-to-back on one stream (with two streams an event pair also
    # spans the wait for the other stream).
    side_was = eng.use_side
    eng.use_side = False
    hook_was = eng.grad_ready_hook
    if overlap is not None:            # the instrumented step runs the optimizer after backward: nothing beside the GEMMs
        mode["overlap"] = False
        eng.grad_ready_hook = red.ready if red.enabled else None
        red.on_bucket = None
    # The host enqueues a step faster than the GPU runs it (~14 vs ~17 ms) but not once two event records are added per
    # GEMM: three plain steps are queued first (no sync), so that the GPU works off a backlog while the instrumented step
    # is enqueued -- an event pair then brackets the kernel alone instead of the kernel plus the host's lag.
    graph[0] = None                    # the instrumented step is enqueued eagerly (HIP events around every GEMM launch)
    chain[0] = None
    # what the host needs to enqueue ONE step when nothing holds it back (empty queue: no back-pressure from the GPU)
    step()                             # (the first eager step after a graph capture re-allocates its workspaces: not timed)
    sync()
    t1 = time.perf_counter()
    step()
    t_host = time.perf_counter() - t1
    for _ in range(3):
        step()
    with GemmProbe(ops) as probe:
        step()
    eng.use_side = side_was
    eng.grad_ready_hook = hook_was
    agg = probe.summary()
    dp = None
    if world > 1:
        # one more untimed, host-enqueued step with every bucket's launch / completion stamped (HIP events on the compute
        # stream): exposed communication = end of backward -> last bucket complete
        try:
            sync()
            red.trace = []
            eng.zero_grad()
            mask = ops.dropout_mask(B, model.head.channels, keep, dev)
            inst, _ = eng.forward(image4, points, None, 0, mask, training=True, materialize_aux=False)
            _, d_inst, d_sim = vpu_step_losses(inst, None, gt, None, None, iter_weight=1.0, sim_low=eng.sim_low)
            red.begin()
            eng.backward(d_inst, None, d_sim_low=d_sim)
            red.mark("bwd_end")
            finish_and_step(red, opt)
            red.mark("step_end")
            sync()
            dp = dp_report(red, world, os.environ.get("VPU_DIST_BACKEND", "nccl"),
                           graph_note[0] or "eager (host-enqueued kernels, collectives from the backward tape's markers)")
        except Exception as e:         # diagnostics must never cost the measurement
            dp = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        red.trace = None
    if agg:
        name, (fl, sec, cnt) = max(agg.items(), key=lambda kv: kv[1][1])
        peak = BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
        traffic, traffic_source = pmc_traffic(name, args)
        roof = {"bound": "mfma", "kernel": name, "achieved": round(fl / sec / 1e12, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(fl / sec / 1e12 / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                "launches_per_step": cnt, "avg_launch_us": round(sec / cnt * 1e6, 2),
                "all_gemm_variants": {k: {"TFLOP/s": round(v[0] / v[1] / 1e12, 2), "ms": round(v[1] * 1e3, 3),
                                           "launches": v[2]} for k, v in sorted(agg.items())}}
    if rank == 0:
        line = {"metric": f"448x448 images/sec fwd+bwd, {NAMES[args.model]} VPUFormer", "value": round(value, 2), "unit": "images/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"{args.model} 448x448 bs={B}/GPU click-only training step "
                                       f"(fwd + NFL/Dice/P2CL + bwd + fused Adam), random-init weights",
                           "global_batch": B * world, "parallelism": f"dp{world}",
                           "flop_per_image_fwd_bwd": FLOP_PER_IMG[args.model],
                           "mfma_roofline_frac_end_to_end":
                               round(value / world * FLOP_PER_IMG[args.model] / (BF16_PEAK_TFLOPS * 1e12), 4),
                           "launch": graph_note[0] or ("hipGraph replay of the captured step" if use_graph else "eager (host-enqueued)"),
                           "optimizer": "Adam per finished gradient range on a second stream, overlapped with backward"
                                        if overlap is not None else "one Adam launch after backward",
                           "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
                           "host_eager_enqueue_ms_unblocked": round(t_host * 1e3, 3),
                           "final_loss": round(loss_val, 5)},
                "roofline": roof}
        if dp is not None:
            line["dp"] = dp
        if world == 1 and not args.no_cpu_baseline:
            ref = {} if (args.model == "vitb" and args.dtype == "bf16") else None
            line["cpu_baseline"] = cpu_baseline(args.cpu_batch, gpu_batch=B, keep_ref=ref)
            if ref:
                try:
                    line["parity"] = parity_vs_oracle(ref, vitb_model_kwargs(), dev)
                except Exception as e:
                    line["parity"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
