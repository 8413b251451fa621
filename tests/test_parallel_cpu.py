"""CPU, gloo, world_size 2: the bucketed gradient reducer sums exactly once per element, launches few large
collectives in tail-first order, and the parameter broadcast / loss reduce behave like the reference's DDP glue."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pvpuformer_amd.parallel import GradReducer, broadcast_parameters, reduce_loss_dict
    n = 1000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradReducer(g, bucket_bytes=4 * 300)
    red.begin()
    # ranges arrive tail-first, as Engine's tape markers produce them
    for lo, hi in [(900, 1000), (700, 900), (640, 700), (300, 640), (120, 300), (0, 120)]:
        red.ready(lo, hi)
    scale = red.finish()
    expect = torch.arange(n, dtype=torch.float32) * 3
    ok = torch.equal(g, expect) and scale == 0.5
    covered = sorted(red.launched)
    ok = ok and covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    ok = ok and len(red.launched) <= 4
    p = torch.full((10,), float(rank + 5))
    broadcast_parameters(p, src=0)
    ok = ok and torch.all(p == 5.0).item()
    l = reduce_loss_dict({"a": torch.tensor(float(rank)), "b": torch.tensor(2.0 * rank)})
    ok = ok and abs(l["a"].item() - 0.5) < 1e-6 and abs(l["b"].item() - 1.0) < 1e-6
    # a second step re-uses the reducer
    g.copy_(torch.ones(n) * (rank + 1))
    red.begin()
    red.ready(500, 1000); red.ready(0, 500)
    red.finish()
    ok = ok and torch.all(g == 3.0).item()
    # bf16 wire format: each rank's bucket is rounded to bf16, summed in bf16, widened back (exact on these small integers)
    g.copy_(torch.arange(n, dtype=torch.float32).remainder(64) * (rank + 1))
    red16 = GradReducer(g, bucket_bytes=4 * 300, wire="bf16")
    red16.begin()
    red16.ready(500, 1000); red16.ready(0, 500)
    ok = ok and red16.finish() == 0.5 and torch.equal(g, torch.arange(n, dtype=torch.float32).remainder(64) * 3)
    q.put((rank, bool(ok), red.launched))
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


def _solo(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    from pvpuformer_amd.parallel import GradReducer
    g = torch.arange(100, dtype=torch.float32)
    plain = GradReducer(g)
    forced = GradReducer(g, bucket_bytes=4 * 30, force=True)
    forced.begin()
    for lo, hi in [(70, 100), (40, 70), (0, 40)]:
        forced.ready(lo, hi)
    scale = forced.finish()
    q.put((not plain.enabled) and forced.enabled and scale == 1.0 and len(forced.launched) == 3
          and torch.equal(g, torch.arange(100, dtype=torch.float32)))
    dist.destroy_process_group()


def test_forced_reducer_at_world_size_1_runs_real_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)


def test_reducer_is_a_noop_without_process_group():
    from pvpuformer_amd.parallel import GradReducer
    g = torch.ones(10)
    r = GradReducer(g)
    r.begin(); r.ready(0, 10)
    assert r.finish() == 1.0 and torch.all(g == 1)
