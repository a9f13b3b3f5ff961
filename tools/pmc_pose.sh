#!/bin/bash
# tools/pmc_pose.sh [f32|bf16]  -- LDS counters of the render kernel per camera pose (tools/pose_sweep.py: 5 poses x 13
# launches each), one --pmc pass.  Run on the GPU box (via gpurun); output gpurun_out/pmc_pose_<dtype>.txt
set -u
DT=${1:-bf16}
OUT=gpurun_out/pmc_pose_$DT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
    --output-format csv -d $OUT -o p -- python tools/pose_sweep.py $DT > $OUT/run.log 2>&1
python - <<EOF > gpurun_out/pmc_pose_$DT.txt
import csv, glob
from collections import defaultdict
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
order = []
for row in csv.DictReader(open(f)):
    if "render_lds" not in row["Kernel_Name"]:
        continue
    d = int(row["Dispatch_Id"])
    if d not in acc:
        order.append(d)
    acc[d][row["Counter_Name"]] += float(row["Counter_Value"])
order.sort()
poses = ["yaw 0.000", "yaw 0.150", "yaw 0.300", "yaw 0.450", "yaw 0.578"]
per = len(order) // 5
for i, name in enumerate(poses):
    ds = order[i * per + 3:(i + 1) * per]
    m = {k: sum(acc[d][k] for d in ds) / len(ds) for k in acc[ds[0]]}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"{name}: kernel {cyc/1e6:.2f} M cycles; LDS active {m['SQ_LDS_IDX_ACTIVE']/256/cyc:.1%} of them, bank conflicts "
          f"{m['SQ_LDS_BANK_CONFLICT']/256/cyc:.1%}; VALU insts/SIMD {m['SQ_INSTS_VALU']/1024/1e3:.0f} K, LDS insts/CU {m['SQ_INSTS_LDS']/256/1e3:.0f} K")
EOF
cat gpurun_out/pmc_pose_$DT.txt
