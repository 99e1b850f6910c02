"""rocprofv3 kernel_stats.csv -> markdown table for profiles/ (usage: stats_to_md.py csv n_updates title cmd)."""
import csv, sys
f, n_upd, title, cmd = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4]
rows = list(csv.DictReader(open(f)))
if n_upd <= 0:      # derive the number of updates (or env steps) in the trace from a once-or-twice-per-unit kernel
    calls = {r["Name"]: int(r["Calls"]) for r in rows}
    adam = sum(v for k, v in calls.items() if "k_adam_soft" in k)
    env = sum(v for k, v in calls.items() if "k_env_step" in k)
    n_upd = adam / 2.0 if adam else float(env)
print("# %s\n" % title)
print("Command: `%s` (MI355X, via gpurun).  %d updates (env probe: batched steps) in the trace (pre-warm + warm-up + timed + the 20-update event-timing pass; counted from the optimiser / env-step launches).\n" % (cmd, n_upd))
print("| kernel | calls | calls/update | avg (us) | us/update | % |")
print("|---|---|---|---|---|---|")
tot = 0.0
for r in rows:
    name = r["Name"].replace("dqnhip::", "").replace("void ", "")
    if "(" in name: name = name[:name.index("(")]
    calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    per = calls / n_upd
    if name.startswith("__amd_rocclr") or name.startswith("k_add_transitions") or per < 0.5:
        continue
    tot += avg * per
    print("| `%s` | %d | %.1f | %.2f | %.1f | %s |" % (name, calls, per, avg, avg * per, r["Percentage"]))
print("\nSum of kernel time per update: **%.0f us** (prefill / copies excluded)." % tot)
