"""rocprofv3 --pmc pass directories (tools/pmc_passes.sh) -> one csv `pass_,counter,mean_per_dispatch,dispatches` for the
kernel whose name contains a substring (the format of profiles/r*/pmc_step_kernel.csv, which bench.py reads)."""
import csv, glob, os, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else "lob_step_kernel"
rows, kernels = [], set()
for d in sorted(os.listdir(root)):
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(f"{root}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
                kernels.add(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lobk::", ""))
    rows += [(d, k, acc[k] / n[k], n[k]) for k in sorted(acc)]
with open(out, "w") as f:
    # (which kernel the counters belong to: bench.py:pmc_traffic refuses a file whose kernel is not the one its handle launches)
    f.write("# kernel: " + " | ".join(sorted(kernels)) + "\n")
    f.write("pass_,counter,mean_per_dispatch,dispatches\n")
    for r in rows: f.write(f"{r[0]},{r[1]},{r[2]},{r[3]}\n")
print(open(out).read())
