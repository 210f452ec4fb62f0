"""Per-launch medians of the PMC passes of scripts/profile_round2.sh, by kernel and SpMV launch mode -> JSON on stdout.
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 the counter tallies 128-byte requests at 64 bytes);
WRITE_SIZE is taken as reported (KiB)."""
import collections
import csv
import glob
import json
import re
import sys

prefix = sys.argv[1]
med = lambda a: sorted(a)[len(a) // 2]


def name_of(k):
    m = re.search(r"k_spmv_s<(\d)", k)
    if m:
        return f"k_spmv_s<mode {m.group(1)}>"
    m = re.search(r"(k_[a-z_0-9]+)", k)
    return m.group(1) if m else k[:40]


out = collections.defaultdict(dict)
for f in sorted(glob.glob(prefix + "_*/*/*counter_collection.csv")):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = name_of(r["Kernel_Name"])
        by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, cs in by.items():
        if not (k.startswith("k_spmv_s") or k in ("k_bicg_xrp", "k_bicg_s", "k_rhs_init")):
            continue
        for c, v in cs.items():
            # launches queued after convergence return at once: keep the launches that did work (upper half by value)
            v = sorted(v)
            v = v[len(v) // 4:]
            out[k][c] = {"median": med(v), "launches": len(v)}
        out[k].setdefault("duration_us_under_pmc", med(sorted(dur[k])[len(dur[k]) // 4:]) / 1e3)
res = {"source": "rocprofv3 --kernel-trace --pmc <one set per pass> -- python3 scripts/dev_perf.py 512 4", "kernels": out}
for k, cs in out.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        cs["hbm_bytes_per_launch"] = 2.0 * cs["FETCH_SIZE"]["median"] * 1024.0 + cs["WRITE_SIZE"]["median"] * 1024.0
# the launch bench.py reports as the dominant kernel: the chain launch of the preconditioner polynomial -- mode 8 (Horner step,
# x-space form of the loop, the default) or mode 4 (lean launch of the y-space form), whichever the run used more
def launches_of(k):
    return out.get(k, {}).get("FETCH_SIZE", {}).get("launches", 0)
chain_kernel = "k_spmv_s<mode 8>" if launches_of("k_spmv_s<mode 8>") > launches_of("k_spmv_s<mode 4>") else "k_spmv_s<mode 4>"
lean = out.get(chain_kernel, {})
if "hbm_bytes_per_launch" in lean:
    res["hbm_bytes_per_launch"] = lean["hbm_bytes_per_launch"]
    res["hbm_bytes_per_launch_note"] = chain_kernel + ": the chain launch of the preconditioner polynomial, the launch bench.py reports as the dominant kernel"
# kernel trace of the bench run itself (no counters): the per-kernel durations bench.py's HIP-event figures are cross-checked
# against.  MEDIAN over the launches that did work: launches queued past convergence return at the done flag after ~5 us
# and would pull an average down (the *_kernel_stats.csv averages include them).  argv[2] = <name>_kernel_trace.csv
if len(sys.argv) > 2:
    durs = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        m = re.search(r"k_spmv_s<(\d)", r["Kernel_Name"])
        if m:
            durs[f"k_spmv_s<mode {m.group(1)}>"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    stats = {}
    for k, v in durs.items():
        work = sorted(x for x in v if x > 15.0)
        if work:
            stats[k] = {"calls": len(v), "returned_at_the_done_flag": len(v) - len(work), "median_us": med(work),
                        "mean_us": sum(work) / len(work), "p10_us": work[len(work) // 10], "p90_us": work[9 * len(work) // 10]}
    res["kernel_trace_of_the_bench_run"] = stats
    res["kernel_trace_note"] = ("durations of the launches that did work; consecutive dependent launches are contiguous in the "
                                "trace (end = next start), so the gap between them is inside these figures as it is inside bench.py's events")
    if chain_kernel in stats:
        res["lean_launch_kernel_trace_avg_us"] = stats[chain_kernel]["median_us"]
print(json.dumps(res, indent=1))
