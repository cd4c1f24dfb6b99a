"""Per-kernel averages of the SQ counter passes that scripts/pmc_sq.sh leaves under gpurun_out/sq."""
import collections
import csv
import sys

for f in sys.argv[1:]:
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xrit::", "")[:46]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, v in per.items():
        if not any(s in k for s in ("clock_pass", "clock_output", "costas_pass", "fir_decim", "newton_apply_waves")):
            continue
        print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
