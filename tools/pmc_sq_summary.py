"""Per-kernel means (per launch, x1e6) of the SQ counters of rocprofv3 --pmc passes and the derived MFMA-pipe busy fraction
VALU_MFMA_BUSY_CYCLES / (BUSY_CYCLES x 32) and s_waitcnt share WAIT_INST_ANY / WAVE_CYCLES.
usage: python tools/pmc_sq_summary.py <dir> [name filter]"""
import collections, csv, glob, re, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob(f"{d}/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        n = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]
        if flt not in n:
            continue
        a = agg[n][r["Counter_Name"].replace("SQ_", "")]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for n, cs in sorted(agg.items()):
    mean = {k: v[1] / v[0] for k, v in cs.items()}
    print(f"   {n} " + str({k: round(v / 1e6, 2) for k, v in sorted(mean.items())}))
    if "VALU_MFMA_BUSY_CYCLES" in mean and "BUSY_CYCLES" in mean:
        print(f"      MFMA-pipe busy fraction {mean['VALU_MFMA_BUSY_CYCLES'] / (mean['BUSY_CYCLES'] * 32):.3f}   s_waitcnt share of wave cycles "
              f"{mean.get('WAIT_INST_ANY', 0) / max(mean.get('WAVE_CYCLES', 1), 1):.3f} (LDS {mean.get('WAIT_INST_LDS', 0) / max(mean.get('WAVE_CYCLES', 1), 1):.3f})")
