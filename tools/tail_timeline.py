"""Timeline of the last outer steps from a rocprofv3 kernel trace CSV (gaps between bulk updates = exposed panel chain)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "true>" in r["Kernel_Name"] and "gemm_nt" in r["Kernel_Name"]]
last = idx[-62:]
gaps = sum(max(0, int(rows[last[i + 1]]["Start_Timestamp"]) - int(rows[last[i]]["End_Timestamp"])) for i in range(61))
print("sum of gaps between the 62 bulk updates of the last step: %.2f ms" % (gaps / 1e6))
a, b = idx[-3], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print("%-36s q%s st %8.1f us dur %6.1f us grid %s" % (n[:36], r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]))
