#!/usr/bin/env python
"""Micro-benchmark of vpu_gemm on the VPUFormer ViT-B bs=12 shapes (random bf16 data).  Prints TFLOP/s per shape.
usage: python tools/gemm_bench.py [reps]"""
import os
import sys

if "k3" in os.environ.get("GEMM_BENCH_K2", "").split(","):
    os.environ.setdefault("VPU_LIB_DIAG", "1")   # the K3 forward forms live in the laboratory library (bash pvpuformer_amd/csrc/build.sh diag)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvpuformer_amd import ops  # noqa: E402

M = int(os.environ.get("GEMM_BENCH_M", "9408"))      # token rows: 9408 (ViT-B bs 12), 6272 (ViT-L bs 8), 12288 (ViT-H bs 12)
D = int(os.environ.get("GEMM_BENCH_D", "768"))       # width: 768 / 1024 / 1280
SHAPES = [  # name, tA, tB, M, N, K, flags
    ("qkv fwd", 0, 0, M, 3 * D, D, ops.EPI_BIAS),
    ("proj fwd+res", 0, 0, M, D, D, ops.EPI_BIAS | ops.EPI_RESID),
    ("fc1 fwd gelu", 0, 0, M, 4 * D, D, ops.EPI_BIAS | ops.EPI_GELU | ops.EPI_SAVE_DGELU),
    ("fc2 fwd+res", 0, 0, M, D, 4 * D, ops.EPI_BIAS | ops.EPI_RESID),
    ("fc2 dgrad*aux", 0, 1, M, 4 * D, D, ops.EPI_MULAUX),
    ("fc1 dgrad", 0, 1, M, D, 4 * D, 0),
    ("qkv dgrad", 0, 1, M, D, 3 * D, 0),
    ("proj dgrad", 0, 1, M, D, D, 0),
    ("fc1 wgrad", 1, 1, 4 * D, D, M, ops.EPI_OUT_F32 | ops.EPI_ACCUM),
    ("fc2 wgrad", 1, 1, D, 4 * D, M, ops.EPI_OUT_F32 | ops.EPI_ACCUM),
    ("qkv wgrad", 1, 1, 3 * D, D, M, ops.EPI_OUT_F32 | ops.EPI_ACCUM),
    ("proj wgrad", 1, 1, D, D, M, ops.EPI_OUT_F32 | ops.EPI_ACCUM),
    ("square 4096", 0, 0, 4096, 4096, 4096, 0),
]
NECK = [  # the DMA neck's prompt-token (576-row) and 384-wide GEMMs
    ("tok lin 768", 0, 0, 576, 768, 768, ops.EPI_BIAS),
    ("tok lin 384", 0, 0, 576, 384, 768, ops.EPI_BIAS),
    ("tok out 384", 0, 0, 576, 768, 384, ops.EPI_BIAS | ops.EPI_RESID),
    ("tok dgrad", 0, 1, 576, 768, 768, 0),
    ("tok wgrad", 1, 1, 768, 768, 576, ops.EPI_OUT_F32 | ops.EPI_ACCUM),
    ("img kproj 384", 0, 0, M, 384, 768, ops.EPI_BIAS),
    ("img kproj dgrad", 0, 1, M, 768, 384, 0),
    ("img kproj wgrad", 1, 1, 384, 768, M, ops.EPI_OUT_F32 | ops.EPI_ACCUM),
    ("img out 384", 0, 0, M, 768, 384, ops.EPI_BIAS | ops.EPI_RESID),
    ("fpn lin 4.5", 0, 0, 150528, 128, 192, ops.EPI_BIAS),
    ("fpn lin 8.2", 0, 0, 37632, 256, 384, ops.EPI_BIAS),
    ("fpn lin 16", 0, 0, M, 512, 768, ops.EPI_BIAS),
    ("head conv0", 0, 0, 150528, 256, 128, ops.EPI_BIAS | ops.EPI_RELU),
    ("head conv1", 0, 0, 37632, 256, 256, ops.EPI_BIAS | ops.EPI_RELU),
    ("head fuse0", 0, 0, 150528, 256, 256, ops.EPI_BIAS),
    ("head fuse0 dgrad", 0, 1, 150528, 256, 256, ops.EPI_DRELU),
    ("fpn lin 4.5 dgrad", 0, 1, 150528, 192, 128, 0),
    ("fpn ct 4.3 dgrad", 0, 0, 37632, 384, 768, 0),
    ("fpn wgrad", 1, 1, 256, 1024, 150528, ops.EPI_OUT_F32 | ops.EPI_ACCUM),
    ("head wgrad", 1, 1, 256, 128, 150528, ops.EPI_OUT_F32 | ops.EPI_ACCUM),
]


def set_k2(opt):
    """k2 option value: 0 the 128 x 128 kernel, 1 the 256 x 128 form, 2 the default rule, 3 the 256 x 256 form wherever legal;
    "k3": the round-4 form (two 256-thread workgroups per CU) where it is legal, the default rule elsewhere"""
    ops.gemm_set_option("k5", 0)
    ops.gemm_set_option("k5_noepi", 0)
    if opt == "k5x":            # K5 main loops only (diagnostic: nothing is stored)
        ops.gemm_set_option("k5_noepi", 1)
        opt = "k5n"
    if opt == "k5a":            # K5 wherever it is legal (also one tile per workgroup)
        ops.gemm_set_option("k3", -1)
        ops.gemm_set_option("k2", -1)
        ops.gemm_set_option("k5_split", 0)
        ops.gemm_set_option("k5", 2)
        return
    if opt in ("k5s", "k5n"):   # K5 with the LDS-DMA issue always / never shared by both wave groups
        ops.gemm_set_option("k5_split", 1 if opt == "k5s" else 0)
        opt = "k5"
    else:
        ops.gemm_set_option("k5_split", -1)
    if opt == "k5":        # round 6: the two-tile ping-pong form wherever more than one tile per workgroup (default rule otherwise)
        ops.gemm_set_option("k3", -1)
        ops.gemm_set_option("k2", -1)
        ops.gemm_set_option("k5", 1)
        return
    if opt == "k3":
        ops.gemm_set_option("k2", -1)
        ops.gemm_set_option("k3", 3)
        return
    if opt == "k3s":       # the 128 x 128 ring kernel where the round-1 128 x 128 kernel would run
        ops.gemm_set_option("k2", -1)
        ops.gemm_set_option("k3", 4)
        return
    ops.gemm_set_option("k3", 0 if opt != "-1" else -1)
    ops.gemm_set_option("k2", int(opt.split(":")[0]))


def group_bench(reps):
    """The four weight gradients of one ViT block as ONE grouped launch, per k2 option."""
    dev = "cuda"
    shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768)]
    probs = []
    fl = 0.0
    for m, n in shapes:
        A = (torch.rand(M, m, device=dev) - 0.5).to(torch.bfloat16)
        Bm = (torch.rand(M, n, device=dev) - 0.5).to(torch.bfloat16)
        C = torch.zeros(m, n, device=dev)
        cs = torch.zeros(m, device=dev)
        probs.append(((A, Bm, C, m, n, M, m, n, n, 0), dict(transA=True, transB=True, flags=ops.EPI_OUT_F32 | ops.EPI_ACCUM, colsum=cs)))
        fl += 2.0 * m * n * M
    for opt in [t for t in os.environ.get("GEMM_BENCH_K2", "0,1").split(",")]:
        set_k2(opt)
        for _ in range(3):
            ops.gemm_grouped(probs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm_grouped(probs)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        print(f"wgrad group      k2={opt}  {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s")
    set_k2("-1")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    if os.environ.get("GEMM_BENCH_GROUP", "0") == "1":
        group_bench(reps)
    dev = "cuda"
    tot_t, tot_f = 0.0, 0.0
    with_torch = os.environ.get("GEMM_BENCH_TORCH", "0") == "1"
    only = [t for t in os.environ.get("GEMM_BENCH_ONLY", "").split(",") if t]
    for name, tA, tB, m, n, k, flags in SHAPES + NECK:
        if only and not any(t in name for t in only):
            continue
        A = (torch.rand((k, m) if tA else (m, k), device=dev) - 0.5).to(torch.bfloat16)
        Bm = (torch.rand((k, n) if tB else (n, k), device=dev) - 0.5).to(torch.bfloat16)
        out_f32 = bool(flags & ops.EPI_OUT_F32)
        C = torch.zeros(m, n, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
        bias = torch.rand(n, device=dev)
        R = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
        aux = torch.ones(m, n, device=dev, dtype=torch.bfloat16)
        pre = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
        kw = dict(transA=bool(tA), transB=bool(tB), flags=flags, bias=bias, resid=R, ldr=n, aux=aux, ldaux=n, preact=pre)
        lda, ldb = (m if tA else k), (n if tB else k)
        k2s = [t for t in os.environ.get("GEMM_BENCH_K2", "").split(",") if t]
        if k2s:   # interleaved A/B of the kernel families in one process (guide rule 24): median of 5 rounds per option
            res = {}
            for rnd_ in range(5):
                for opt in k2s:
                    set_k2(opt)
                    ops.gemm(A, Bm, C, m, n, k, lda, ldb, n, 0, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        ops.gemm(A, Bm, C, m, n, k, lda, ldb, n, 0, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault(opt, []).append(e0.elapsed_time(e1) * 1e-3 / reps)
            set_k2("-1")
            ops.gemm_set_option("k5", -1)
            ops.gemm_set_option("k5_noepi", 0)
            ops.gemm_set_option("k5_split", -1)
            fl = 2.0 * m * n * k
            txt = "  ".join(f"k2={o}: {sorted(v)[2] * 1e6:7.1f} us {fl / sorted(v)[2] / 1e12:6.0f} TF" for o, v in res.items())
            print(f"{name:16s} M={m:6d} N={n:5d} K={k:5d}  {txt}")
            continue
        for _ in range(3):
            ops.gemm(A, Bm, C, m, n, k, lda, ldb, n, 0, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if os.environ.get("GEMM_BENCH_GRAPH", "0") == "1":   # replay a captured hipGraph: GPU time without the host launch cost
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(reps):
                    ops.gemm(A, Bm, C, m, n, k, lda, ldb, n, 0, **kw)
            graph.replay()
            torch.cuda.synchronize()
            e0.record()
            graph.replay()
            e1.record()
        else:
            e0.record()
            for _ in range(reps):
                ops.gemm(A, Bm, C, m, n, k, lda, ldb, n, 0, **kw)
            e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        fl = 2.0 * m * n * k
        if (name, tA, tB, m, n, k, flags) in SHAPES and "square" not in name:
            tot_t += t; tot_f += fl
        extra = ""
        if with_torch:   # library GEMM (hipBLASLt/rocBLAS through torch.matmul) on the same operands, no epilogue: calibration only
            a2 = A.t() if tA else A
            b2 = Bm if tB else Bm.t()
            for _ in range(3):
                torch.matmul(a2, b2)
            e0.record()
            for _ in range(reps):
                torch.matmul(a2, b2)
            e1.record()
            torch.cuda.synchronize()
            tt = e0.elapsed_time(e1) * 1e-3 / reps
            extra = f"   | torch.matmul {tt * 1e6:8.1f} us {fl / tt / 1e12:7.1f} TFLOP/s"
        print(f"{name:16s} tA={tA} tB={tB} M={m:6d} N={n:5d} K={k:5d}  {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s{extra}")
    if tot_t > 0:
        print(f"{'block total':16s} {tot_t * 1e6:8.1f} us  {tot_f / tot_t / 1e12:7.1f} TFLOP/s (one ViT block's 12 GEMMs)")


if __name__ == "__main__":
    main()
