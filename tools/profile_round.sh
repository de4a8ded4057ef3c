#!/bin/bash
# Collects this round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <round-tag, e.g. r02>
# Kernel-trace statistics and PMC counters are separate runs (gpurun refuses --pmc combined with API traces), and
# FETCH_SIZE / WRITE_SIZE are separate passes (TCC has 4 counter slots: FETCH_SIZE costs 3, WRITE_SIZE 2).
# The program follows `--` directly (python3 ...): no env / bash -c hop under the profiler.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/${TAG}_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 $PWD/bench.py"
cd "$PWD"
# 1. per-kernel time of the default bench command (no CPU leg, no extras: the kernels of the timed region only)
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- $B --cpu-seconds 0 --no-extras > "$OUT/kt_bench.json" 2> "$OUT/kt.log"
echo "kt rc=$?"
# 2. HBM-side traffic of one launch of each kernel: two PMC passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o p --output-format csv -- $B --steps 1 --warmup 0 --cpu-seconds 0 --no-extras > /dev/null 2> "$OUT/fetch.log"
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o p --output-format csv -- $B --steps 1 --warmup 0 --cpu-seconds 0 --no-extras > /dev/null 2> "$OUT/write.log"
echo "write rc=$?"
# 3. instruction mix of the argmin kernel (SQ block: 8 slots)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/sq" -o p --output-format csv -- $B --steps 1 --warmup 0 --cpu-seconds 0 --no-extras > /dev/null 2> "$OUT/sq.log"
echo "sq rc=$?"
# 4. the online path: micro-batched streaming over 1000 frames, kernel statistics + bench's own busy fraction
rocprofv3 --kernel-trace --stats -d "$OUT/stream" -o st --output-format csv -- $B --mode stream --frames 1000 --steps 1 --warmup 1 --cpu-seconds 0 > "$OUT/stream_bench.json" 2> "$OUT/stream.log"
echo "stream rc=$?"
# 5. cfg4 fused path: score kernel + the three loop-test kernels
rocprofv3 --kernel-trace --stats -d "$OUT/cfg4" -o c4 --output-format csv -- $B --workload cfg4 --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/cfg4_bench.json" 2> "$OUT/cfg4.log"
echo "cfg4 rc=$?"
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete     # keep the merge-back small
ls -R "$OUT" | head -60
