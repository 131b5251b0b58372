"""Per-kernel SQ counter ratios from rocprofv3 --pmc passes over `tools/microbench.py pmc`.
usage: python tools/pmc_kernels.py <counter_collection.csv> [<counter_collection.csv> ...]
Counters are summed over the dispatches of a kernel (all XCDs / SEs); what is printed are RATIOS of them (MI355X_MICROARCH.md:
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES, disjoint)."""
import collections
import csv
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for path in sys.argv[1:]:
    seen = set()
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].replace("void aic::", "")
            if "verify_attn" not in k or "combine" in k:
                continue
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key)
                calls[k] += 1
for k in sorted(tot):
    c = tot[k]
    line = f"{k:46s}"
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        line += "  of wave cycles: parked (waitcnt/barrier) %4.1f%%  issue-stalled %4.1f%%  issuing %4.1f%%" % (
            100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc)
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
                 "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        if name in c:
            line += f"  {name}={c[name]:.3g}"
    print(line)
