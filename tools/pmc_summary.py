"""Aggregate rocprofv3 counter-collection CSVs (one directory per --pmc pass) into one JSON:
{pass: {kernel: {counter: {"sum": total over dispatches, "dispatches": n}}}}.  Kernel names are cut at '('."""
import csv, glob, json, os, sys

def short(name):
    return name.split("(")[0].strip()

def summarise(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k, c = short(row["Kernel_Name"]), row["Counter_Name"]
                e = out.setdefault(k, {}).setdefault(c, {"sum": 0.0, "dispatches": 0})
                e["sum"] += float(row["Counter_Value"])
                e["dispatches"] += 1
    return out

if __name__ == "__main__":
    res = {}
    for spec in sys.argv[2:]:
        name, d = spec.split("=", 1)
        res[name] = summarise(d)
    # fingerprint of the sources these counters belong to (bench.py compares it with the sources it runs on)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import bench
        res["sources_sha16"] = bench._sources_sha16()
    except Exception as e:   # noqa: BLE001
        res["sources_sha16"] = None
        res["sources_sha16_error"] = str(e)
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print("wrote", sys.argv[1], {k: len(v) for k, v in res.items() if isinstance(v, dict)})
