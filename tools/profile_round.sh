#!/bin/bash
# Collects this round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <round-tag, e.g. r03> [quick]
# Kernel-trace statistics and PMC counters are separate runs (gpurun refuses --pmc combined with API traces), and
# FETCH_SIZE / WRITE_SIZE are separate passes (TCC has 4 counter slots: FETCH_SIZE costs 3, WRITE_SIZE 2).
# The program follows `--` directly (python3 ... / the binary): no env / bash -c hop under the profiler.
set -u
TAG=${1:-r03}
QUICK=${2:-}
OUT=$PWD/gpurun_out/${TAG}_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 $PWD/bench.py"
ONE="--steps 1 --warmup 0 --cpu-seconds 0 --no-extras"
cd "$PWD"
# 1. per-kernel time of the default bench command (no CPU leg, no extras: the kernels of the timed region only)
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- $B --cpu-seconds 0 --no-extras > "$OUT/kt_bench.json" 2> "$OUT/kt.log"
echo "kt rc=$?"
# 2. HBM-side traffic of one step of each kernel: two PMC passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o p --output-format csv -- $B $ONE > /dev/null 2> "$OUT/fetch.log"
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o p --output-format csv -- $B $ONE > /dev/null 2> "$OUT/write.log"
echo "write rc=$?"
# 2b. EXACT read bytes: the size-resolved request counters (FETCH_SIZE tallies a 128-byte request as 64 bytes), and where the
#     requests go; the same passes over tools/fetch_calib (2^30 bytes read once per access pattern) calibrate both
R1="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum"
R2="TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum"
rocprofv3 --kernel-trace --pmc $R1 -d "$OUT/rd" -o p --output-format csv -- $B $ONE > /dev/null 2> "$OUT/rd.log"
echo "rd rc=$?"
rocprofv3 --kernel-trace --pmc $R2 -d "$OUT/rd2" -o p --output-format csv -- $B $ONE > /dev/null 2> "$OUT/rd2.log"
echo "rd2 rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch_calib" -o p --output-format csv -- $PWD/tools/fetch_calib > /dev/null 2> "$OUT/fetch_calib.log"
rocprofv3 --kernel-trace --pmc $R1 -d "$OUT/rd_calib" -o p --output-format csv -- $PWD/tools/fetch_calib > /dev/null 2> "$OUT/rd_calib.log"
echo "calib rc=$?"
# 3. instruction mix (SQ block: 8 slots)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/sq" -o p --output-format csv -- $B $ONE > /dev/null 2> "$OUT/sq.log"
echo "sq rc=$?"
# 4. is the VALU pipe busy?  cycle counters of the product kernels ...
VC="SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $VC -d "$OUT/valu" -o p --output-format csv -- $B $ONE > /dev/null 2> "$OUT/valu.log"
echo "valu rc=$?"
# ... and of the reference loops (pure v_xor_b32, pure v_bcnt_u32_b32, the inner-loop mix without memory traffic)
rocprofv3 --kernel-trace --pmc $VC -d "$OUT/valu_ref" -o p --output-format csv -- $PWD/tools/valu_busy 0.05 > "$OUT/valu_ref.txt" 2> "$OUT/valu_ref.log"
echo "valu_ref rc=$?"
$PWD/tools/valu_busy 0.05 > "$OUT/valu_ref_unprofiled.txt" 2>&1
# ... and the gfx950 dual-issue counter: quad-cycles in which two VALU instructions were issued
V2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $V2 -d "$OUT/valu2" -o p --output-format csv -- $B $ONE > /dev/null 2> "$OUT/valu2.log"
echo "valu2 rc=$?"
rocprofv3 --kernel-trace --pmc $V2 -d "$OUT/valu2_ref" -o p --output-format csv -- $PWD/tools/valu_busy 0.05 > "$OUT/valu2_ref.txt" 2> "$OUT/valu2_ref.log"
echo "valu2_ref rc=$?"
if [ -z "$QUICK" ]; then
# 5. the online path: micro-batched streaming over 1000 frames, kernel statistics + bench's own busy fraction
rocprofv3 --kernel-trace --stats -d "$OUT/stream" -o st --output-format csv -- $B --mode stream --frames 1000 --steps 1 --warmup 1 --cpu-seconds 0 > "$OUT/stream_bench.json" 2> "$OUT/stream.log"
echo "stream rc=$?"
# 6. cfg4 fused path on the selective synthetic variant is part of the default line's extras; here: kernel statistics of
#    the fused call on the default variant (score kernels + fold kernels + the loop-test kernels)
rocprofv3 --kernel-trace --stats -d "$OUT/cfg4" -o c4 --output-format csv -- $B --workload cfg4 --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/cfg4_bench.json" 2> "$OUT/cfg4.log"
echo "cfg4 rc=$?"
fi
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete     # keep the merge-back small
ls -R "$OUT" | head -80
