"""Where one N=32768 step goes, from a rocprofv3 kernel trace of tools/time_sizes.py: time before the first bulk
update, the bulk updates themselves, the gaps between consecutive bulk updates (early = event hand-off, late = the
exposed panel chain), what follows the last one.   python tools/step_budget.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])
fills = [i for i, r in enumerate(rows) if "fill" in r["Kernel_Name"]]
step = rows[fills[-1]:]
bulk = [r for r in step if "gemm_nt_f64_rs<4, 2, true" in r["Kernel_Name"]]
t0, t1 = S(step[0]), max(E(r) for r in step)
lastfac = max(E(r) for r in step if "potrf128" in r["Kernel_Name"] or "gemm_nt" in r["Kernel_Name"])
print("step %.2f ms; %d bulk launches, %.2f ms inside them" % ((t1 - t0) / 1e6, len(bulk), sum(E(r) - S(r) for r in bulk) / 1e6))
print("before the first bulk launch: %.2f ms (fill %.2f)" % ((S(bulk[0]) - t0) / 1e6, (E(step[0]) - S(step[0])) / 1e6))
gaps = [(S(bulk[i + 1]) - E(bulk[i])) / 1e3 for i in range(len(bulk) - 1)]
print("gaps between bulk launches: total %.2f ms; per gap (us):" % (sum(g for g in gaps if g > 0) / 1e3), " ".join("%.0f" % g for g in gaps))
print("bulk durations (ms):", " ".join("%.2f" % ((E(r) - S(r)) / 1e6) for r in bulk))
print("after the last bulk launch until the factorisation's last kernel: %.2f ms; then to the end of the step: %.2f ms" %
      ((lastfac - E(bulk[-1])) / 1e6, (t1 - lastfac) / 1e6))
