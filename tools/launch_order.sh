#!/bin/bash
# The kernel launches of one training step of the headline configuration, in order, with their durations
# (rocprofv3 kernel trace of eager steps; the last step of the run is printed).
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/launch_order
rm -rf $out && mkdir -p $out
cd /tmp
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $repo/tools/step_prof.py --steps 4 > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts with stage_batch_kernel
starts = [i for i, n in enumerate(names) if "stage_batch_kernel" in n]
a = starts[-1]
prev_end = None
tot = 0
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    tot += (e - s) / 1e3
    n = r["Kernel_Name"].replace("gtg::(anonymous namespace)::", "").replace("void ", "")
    print("%7.2f us  gap %6.2f  grid %8s  %s" % ((e - s) / 1e3, gap, r.get("Grid_Size", "?"), n[:110]))
print("kernels: %d, sum of durations %.1f us" % (len(rows) - a, tot))
PY
