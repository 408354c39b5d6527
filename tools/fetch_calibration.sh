#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE on gfx950 for the two load forms the kernels use, on a kernel whose bytes are known
# (tools/stream_lab2.hip streams exactly 122.88 MB per launch): mode 3 = global_load_dwordx4 (nt) to registers,
# mode 4 = global_load_lds_dwordx4 (LDS-DMA, the GEMM's staging form).  Answers whether the x2 correction of
# MI355X_MICROARCH.md (HBM section) also applies to LDS-DMA traffic.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_r02/calib"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$R/tools/stream_lab2.hip" -o /tmp/stream_lab2 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/raw" -o c -- /tmp/stream_lab2 > "$O/run.log" 2>&1
python3 - "$O" <<'PY'
import csv, glob, json, statistics, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(O + "/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024)
known = 16 * 20 * 1500 * 64 * 2 * 2
out = {"known_bytes_per_launch": known, "kernels": {}}
for k, v in sorted(acc.items()):
    m = statistics.median(v)
    out["kernels"][k] = {"launches": len(v), "FETCH_SIZE_bytes_median": m, "reported_over_known": round(m / known, 4)}
json.dump(out, open(O + "/fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf "$O/raw"
