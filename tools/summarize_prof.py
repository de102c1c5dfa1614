"""Summarise gpurun_out/prof (tools/profile.sh) into profiles/<round>_*.md + a
small JSON with the per-launch HBM traffic that bench.py reports as
`roofline.traffic` (measured in separate rocprofv3 --pmc passes)."""
import csv, json, os, sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
os.makedirs("profiles", exist_ok=True)

def stats(name):
    path = f"{src}/{name}_stats/{name}_kernel_stats.csv"
    rows = list(csv.DictReader(open(path)))
    out = [f"### {name}: rocprofv3 --kernel-trace --stats (top kernels)", "",
           "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:8]:
        out.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                   f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    return "\n".join(out), rows

def pmc(name, counter):
    path = f"{src}/{name}_{'fetch' if counter=='FETCH_SIZE' else 'write'}/{name}_counter_collection.csv"
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc

md = [f"# {rnd}: rocprofv3 summaries (MI355X, gfx950)", "",
      "Commands: `tools/profile_r06.sh` (kernel-trace/stats runs and, separately, one `--pmc` pass per counter).",
      "`api` = `tools/api_bench.py`: the drop-in call with evaluation_times Minimal / Full / 0.1 (one 14-atom sequence each).",
      "FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x",
      "(MI355X_MICROARCH.md, HBM section), so HBM read bytes = 2 x FETCH_SIZE x 1024.", ""]
traffic = {}
for name in ("ns", "nst", "nsk", "cfg3", "cfg5", "cfg5k", "cfg5t", "cfg5b", "cfg5c", "cfg2", "api"):
    try:
        text, rows = stats(name)
        md += [text, ""]
    except FileNotFoundError:
        continue
for name, kern_sub in (("ns", "k_split_reg"), ("nst", "k_split14_loop"), ("nsk", "k_ket"), ("cfg3", "k_split_reg"), ("cfg3", "k_transpose_conj"), ("cfg5", "k_split_s"), ("cfg5t", "k_apply<"), ("cfg5b", "k_split_s"), ("cfg5c", "k_split_s"), ("cfg2", "k_split_reg"), ("cfg2", "k_traj")):
    try:
        f = pmc(name, "FETCH_SIZE"); w = pmc(name, "WRITE_SIZE")
    except FileNotFoundError:
        continue
    md += [f"### {name}: HBM traffic of `{kern_sub}` from PMC passes", "",
           "| kernel | launches | FETCH_SIZE KiB/launch | read bytes/launch (x2 corrected) | WRITE_SIZE KiB/launch | total bytes/launch |",
           "|---|---|---|---|---|---|"]
    for k in f:
        if kern_sub not in k:
            continue
        nf, sf = f[k]; nw, sw = w.get(k, [1, 0.0])
        rd = 2 * sf / nf * 1024; wr = sw / max(nw, 1) * 1024
        md.append(f"| `{k[:60]}` | {nf} | {sf/nf:.1f} | {rd:.4g} | {sw/max(nw,1):.1f} | {rd+wr:.4g} |")
        key = {"ns": "north_star", "nst": "north_star", "nsk": "north_star", "cfg3": "cfg3", "cfg5": "cfg5", "cfg5t": "cfg5", "cfg5b": "cfg5_24atoms",
               "cfg5c": "cfg5_22atoms", "cfg2": "cfg2"}[name] + ":" + {"k_split_s": "k_split", "k_split_t": "k_split", "k_split14_loop": "k_split14"}.get(kern_sub, kern_sub.rstrip("<"))
        traffic[key] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                                         "total_bytes_per_launch": rd + wr, "launches_sampled": nf}
    md.append("")
open(f"profiles/{rnd}_rocprof_summary.md", "w").write("\n".join(md))
json.dump(traffic, open(f"profiles/{rnd}_traffic.json", "w"), indent=1)
print("\n".join(md))
