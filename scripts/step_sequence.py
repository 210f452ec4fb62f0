"""Launch sequence of one time step out of a rocprofv3 kernel trace (csv): kernel, count of consecutive launches, mean
duration, mean gap before.  usage: step_sequence.py <kernel_trace.csv> [which step from the end, default 2]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def nm(k):
    m = re.search(r"k_spmv_s<(\d)", k)
    if m:
        return f"spmv{m.group(1)}"
    m = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_[A-Za-z]+)", k)
    return m.group(1) if m else k[:30]


idx = [i for i, r in enumerate(rows) if "k_rhs_init" in r["Kernel_Name"]]
i0, i1 = idx[-back - 1], idx[-back]
prev_end, out = None, []
for r in rows[i0 - 2:i1 - 2]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((nm(r["Kernel_Name"]), (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
comp = []
for n, d, g in out:
    if comp and comp[-1][0] == n and abs(comp[-1][1] / comp[-1][3] - d) < 0.3 * d + 3:
        comp[-1][1] += d; comp[-1][2] += g; comp[-1][3] += 1
    else:
        comp.append([n, d, g, 1])
for n, d, g, c in comp:
    print(f"{n:30s} x{c:2d}  {d / c:7.1f} us  gap before {g / c:5.1f} us")
print(f"launches {len(out)}  kernel time {sum(d for _, d, _ in out):.0f} us  gaps {sum(g for _, _, g in out):.0f} us")
