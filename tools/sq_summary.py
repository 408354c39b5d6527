"""Summary of an SQ-counter pass (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) over tools/probe_kernels.py: per kernel, the
medians over its launches and the derived fractions.

    python tools/sq_summary.py <dir with *counter_collection.csv> out.json

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 4 SIMDs x CUs) (the gfx94x formula ROCm falls back to on
gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"); wave-cycle shares: ACTIVE_INST_ANY (issuing), WAIT_INST_ANY (issue stall),
WAIT_ANY (parked at s_waitcnt / barrier) over WAVE_CYCLES (quad-cycle units, disjoint)."""
import csv, glob, json, os, re, statistics, sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
n_cu = int(sys.argv[3]) if len(sys.argv) > 3 else 256
vals = defaultdict(lambda: defaultdict(dict))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not any(k in name for k in ("gemm_8phase", "gemm_pipe", "attn_full", "gemm_glds", "layernorm")):
            continue
        name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
        vals[name][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
res = {"how": __doc__.split("\n\n")[0], "cus": n_cu, "kernels": {}}
for name, disp in vals.items():
    rows = list(disp.values())
    med = {c: statistics.median(r[c] for r in rows if c in r) for c in rows[0]}
    k = {"launches": len(rows), "median": med}
    if med.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE comes summed over the 8 XCDs (profiles/r04_sq_encoder_kernels.json: how that was checked)
        k["mfma_util"] = round(med.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (med["GRBM_GUI_ACTIVE"] / 8 * 4 * n_cu), 4)
    wc = med.get("SQ_WAVE_CYCLES")
    if wc:
        for c, key in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "parked"),
                       ("SQ_ACTIVE_INST_VALU", "valu_active")):
            if c in med:
                k[key + "_of_wave_cycles"] = round(med[c] / wc, 4)
    res["kernels"][name] = k
json.dump(res, open(out, "w"), indent=1)
for n, k in res["kernels"].items():
    print(n, {a: b for a, b in k.items() if a != "median"})
