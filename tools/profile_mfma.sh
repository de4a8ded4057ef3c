#!/bin/bash
# rocprofv3 evidence for the opt-in matrix-core variant (bench.py --variant ${VAR:-4}): kernel statistics, then SQ counters.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/${TAG}_mfma${VAR:-4}
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 $PWD/bench.py"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- $B --variant ${VAR:-4} --cpu-seconds 0 --no-extras > "$OUT/kt_bench.json" 2> "$OUT/kt.log"
echo "kt rc=$?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$OUT/sq" -o p --output-format csv -- $B --variant ${VAR:-4} --steps 1 --warmup 0 --cpu-seconds 0 --no-extras > /dev/null 2> "$OUT/sq.log"
echo "sq rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA -d "$OUT/sq2" -o p --output-format csv -- $B --variant ${VAR:-4} --steps 1 --warmup 0 --cpu-seconds 0 --no-extras > /dev/null 2> "$OUT/sq2.log"
echo "sq2 rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o p --output-format csv -- $B --variant ${VAR:-4} --steps 1 --warmup 0 --cpu-seconds 0 --no-extras > /dev/null 2> "$OUT/fetch.log"
echo "fetch rc=$?"
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
cat "$OUT/kt/kt_kernel_stats.csv" | cut -c1-160
python3 - "$OUT" <<'PY'
import csv, collections, sys
P=sys.argv[1]
for name in ("sq","sq2","fetch"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
    try:
        for r in csv.DictReader(open(f"{P}/{name}/p_counter_collection.csv")):
            k=r["Kernel_Name"]
            if "lcm::" not in k: continue
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
    except Exception as e:
        print(name, "ERR", e); continue
    for k,v in acc.items(): print(name, k[:40], {c: round(x/n[k][c]) for c,x in v.items()})
PY
tail -3 "$OUT/sq2.log"
