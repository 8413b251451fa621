"""Data parallelism on the one-GPU box (a17 / 8e): the real collective path at world size 1 on RCCL, and two ranks sharing
the GPU over gloo -- fresh child processes (tests/dist_worker.py), results compared here."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _spawn(mode, world, out):
    port = _port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), mode, str(r), str(world), port, str(out)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0 and "WORKER-OK" in o, o[-3000:]


def test_rccl_world_size_1_reducer_is_the_identity(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _spawn("nccl1", 1, tmp_path)
    r = np.load(tmp_path / "nccl1.npz")
    assert np.array_equal(r["with_red"], r["plain"]), "a SUM all-reduce over one rank must leave the gradients unchanged"
    assert np.array_equal(r["again"], r["plain"])
    spans = sorted(tuple(x) for x in r["launched"])
    assert len(spans) >= 3 and spans[0][0] == 0 and spans[-1][1] == int(r["total"])
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), "the buckets must tile the flat gradient buffer"
    # the same step as a chain of hipGraphs cut at the reported ranges, collectives launched between the segments
    assert np.array_equal(r["chain0"], r["plain"]) and np.array_equal(r["chain1"], r["plain"])
    assert np.array_equal(r["chain_launched"], r["launched"]), "the replayed chain launches the same buckets in the same order"
    assert r["chain_graphs"][0] >= 1 and r["chain_graphs"][1] >= 3, r["chain_graphs"]
    # ... and with the ranges reported in the middle of backward (no weight-gradient queue across blocks): really cut
    assert np.array_equal(r["eager_ng"], r["plain_ng"])
    assert np.array_equal(r["chain_ng0"], r["plain_ng"]) and np.array_equal(r["chain_ng1"], r["plain_ng"])
    assert np.array_equal(r["chain_launched_ng"], r["launched_ng"])
    assert r["chain_graphs_ng"][0] >= 3 and r["chain_graphs_ng"][1] >= 3, r["chain_graphs_ng"]
    # Adam started on the reduced part while the last collectives are in flight == one Adam launch after all of them
    for k in "pmvs":
        assert np.array_equal(r[f"fs_split_{k}"], r[f"fs_plain_{k}"]), k
    # VPUTrainStep under the reducer: captured passes == host-enqueued passes after eight optimizer steps
    assert np.array_equal(r["ts_graph"], r["ts_eager"])
    ref = r["plain"]
    err = np.abs(r["with_bf16"] - ref)
    assert np.all(err <= 2.0 ** -8 * np.abs(ref) + 1e-30), "bf16 wire format: one rounding to 8 significant bits"
    assert np.any(r["with_bf16"] != ref)


def test_two_ranks_equal_one_rank_with_the_whole_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _spawn("gloo2", 2, tmp_path)
    r0, r1 = np.load(tmp_path / "gloo2_rank0.npz"), np.load(tmp_path / "gloo2_rank1.npz")
    assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["buf"], r1["buf"]) and np.all(r0["buf"] == 1.0)
    assert np.array_equal(r0["mine"], r1["mine"]), "both ranks hold the same reduced gradient"
    full, dp = r0["full"], r0["mine"]
    scale = np.abs(full).max()
    assert np.abs(dp - full).max() <= 2e-5 * scale, np.abs(dp - full).max() / scale
    assert len(r0["launched"]) >= 3 and np.array_equal(r0["launched"], r1["launched"])
    assert np.array_equal(r0["chain"], r0["mine"]) and np.array_equal(r1["chain"], r1["mine"]), "graph chain == host-enqueued step"
