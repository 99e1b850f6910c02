"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid size): python scripts/trace_by_grid.py <dir> [min_calls]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void dqnhip::", "").replace("void ", "")
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
    wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 256)) or 256)
    agg[(name, grid // max(wg, 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in agg.values())
for (name, blocks), v in rows[:40]:
    v2 = sorted(v)
    print("%-44s blocks %6d calls %6d  avg %8.2f us  med %8.2f  sum %5.1f %%" % (name[:44], blocks, len(v), sum(v) / len(v), v2[len(v2) // 2], 100 * sum(v) / tot))
