"""Do the parallel branches of a captured hipGraph run concurrently, and what does a branch / a crossing cost?  Chains of
spin kernels (one thread each, torch.cuda._sleep) captured on one stream or on two (fork / join through events), with and
without crossings (an event recorded on one branch and waited for on the other) every few kernels; replay times."""
import sys, time, torch

def build(two, n, cyc, cross=0):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if two:
            side.wait_stream(main)
            for i in range(n):
                with torch.cuda.stream(side):
                    torch.cuda._sleep(cyc)
                torch.cuda._sleep(cyc)
                if cross and i % cross == cross - 1:      # main -> side -> main
                    e = torch.cuda.Event(); e.record(main); side.wait_event(e)
                    with torch.cuda.stream(side):
                        torch.cuda._sleep(cyc)
                    e2 = torch.cuda.Event(); e2.record(side); main.wait_event(e2)
            main.wait_stream(side)
        else:
            for i in range(2 * n + (n // cross if cross else 0)):
                torch.cuda._sleep(cyc)
    return g

def timeit(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for cyc, n in ((200000, 20), (20000, 100), (2000, 200)):
    a = timeit(build(False, n, cyc))
    b = timeit(build(True, n, cyc))
    c1 = timeit(build(False, n, cyc, cross=5))
    c2 = timeit(build(True, n, cyc, cross=5))
    print(f"spin {cyc:7d} cycles x {2 * n:3d} kernels: one chain {a:7.3f} ms | two branches {b:7.3f} ms | "
          f"with a crossing every 5 kernels: one chain {c1:7.3f} ms, two branches {c2:7.3f} ms")
