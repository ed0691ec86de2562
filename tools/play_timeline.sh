#!/bin/bash
# One rollout step of cfg2 as the GPU sees it: kernel names, durations and the idle gap before each (rocprofv3 --kernel-trace).
#   gpurun -- 'bash tools/play_timeline.sh > gpurun_out/play_timeline.txt'
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -- python $GRAFT_REPO_ROOT/bench.py --config ${1:-cfg2} --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
f=$(find /tmp/ptl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last rollout: find the last 40 im_step launches, print from the 20th-from-last to the next one
idx = [i for i, n in enumerate(names) if "im_step_kernel" in n]
a, b = idx[-20], idx[-18]
prev_end = int(rows[a - 1]["End_Timestamp"])
tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"gap {max(0, s - prev_end) / 1e3:6.2f} us | {(e - s) / 1e3:7.2f} us | {r['Kernel_Name'][:110]}")
    prev_end = max(prev_end, e)
print(f"step: {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us, {b - a} launches")
# one minibatch of the update: from one optimiser launch to the next
ad = [i for i, n in enumerate(names) if "adam" in n]
a, b = ad[-3] + 1, ad[-2] + 1
prev_end = int(rows[a - 1]["End_Timestamp"])
print("--- one minibatch of the update")
busy = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"gap {max(0, s - prev_end) / 1e3:6.2f} us | {(e - s) / 1e3:7.2f} us | {r['Kernel_Name'][:110]}")
    busy += (e - s) / 1e3
    prev_end = max(prev_end, e)
print(f"minibatch: {(int(rows[b - 1]['End_Timestamp']) - int(rows[a - 1]['End_Timestamp'])) / 1e3:.1f} us wall, {busy:.1f} us of kernels, {b - a} launches")
PY
