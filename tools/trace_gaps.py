"""Per-step timeline summary from a rocprofv3 --kernel-trace CSV: busy time, gaps, per-kernel totals of the LAST full step."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step ends with adam_kernel
ends = [i for i, n in enumerate(names) if "adam_kernel" in n]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 2
lo, hi = ends[which - 1] + 1, ends[which] + 1
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = 0; cur_end = t0; gaps = []; union = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if s > cur_end:
        gaps.append(s - cur_end)
        union += e - s
    else:
        union += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
print(f"launches {len(step)}  span {(t1-t0)/1e6:.3f} ms  sum-of-durations {busy/1e6:.3f} ms  union-busy {union/1e6:.3f} ms  "
      f"gap total {sum(gaps)/1e6:.3f} ms over {len(gaps)} gaps (avg {sum(gaps)/max(1,len(gaps))/1e3:.2f} us)")
# time from prev step end
print("idle before step start: %.3f ms" % ((t0 - int(rows[lo-1]["End_Timestamp"]))/1e6))
agg = collections.OrderedDict()
for r in step:
    n = r["Kernel_Name"]
    n = n.split("(")[0][-60:]
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{n:62s} {c:5d} {t/1e6:8.3f} ms  {t/c/1e3:8.1f} us")
big = sorted(gaps)[-10:]
print("largest gaps us:", [round(g/1e3,1) for g in big])
# utilisation per 0.5 ms bucket and the launch count in it
W = 500_000
nb = (t1 - t0) // W + 1
bus = [0] * nb; cnt = [0] * nb
for r in step:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    cnt[s // W] += 1
    b = s // W
    while s < e:
        lim = min(e, (b + 1) * W)
        bus[b] += lim - s
        s = lim; b += 1
print("bucket(0.5ms): util% / launches")
print(" ".join(f"{100*b//W}/{c}" for b, c in zip(bus, cnt)))
