#!/usr/bin/env python3
"""The online (streaming) path through its three forms, one table (profiles/r03_stream_group.txt):
    python tools/stream_table.py [--frames 2500,5000,20000] [--forms single,group1,group8] > profiles/rNN_stream_group.txt
Each cell is one `bench.py --mode stream --frames F --steps 1 --warmup 0` in its own process (a pass from an EMPTY
database, micro-batches of 8, 2 in flight, host rows over PCIe, wall clock):
    single  = one lcm_handle                      (lcm_query_submit_batch / _collect_batch)
    group1  = --gpus 1 --force-group              (lcm_group_query_submit_batch / _collect_batch, RCCL communicator of one)
    group8  = --gpus 8 --loopback                 (8 shards on the one GPU: a rehearsal of the N = 8 path, NOT a scaling number)
The CPU leg is shortened (--cpu-seconds 2) — the table is about the GPU forms; parity fields come from bench.py itself."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMS = {"single": [], "group1": ["--gpus", "1", "--force-group"], "group8": ["--gpus", "8", "--loopback"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="2500,5000,20000")
    ap.add_argument("--forms", default="single,group1,group8")
    args = ap.parse_args()
    print("\n".join("# " + line for line in __doc__.strip().splitlines()))
    print("# frames  form     distances/s   ms per pass   parity")
    sys.stdout.flush()
    for frames in [int(x) for x in args.frames.split(",")]:
        base = None
        for form in args.forms.split(","):
            if form == "group1" and frames >= 20000:
                continue                                  # same code path as group8 minus the threads: not worth 3 more minutes
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "stream", "--frames", str(frames), "--steps", "1",
                   "--warmup", "0", "--cpu-seconds", "2"] + FORMS[form]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                print(f"{frames:6d}  {form:7s}  FAILED rc={r.returncode}: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ''}")
                continue
            d = json.loads(r.stdout.strip().splitlines()[-1])
            parity = d.get("merged_vs_oracle_sample")
            if parity is None:
                parity = {"cpu_sample_mismatches": (d.get("cpu_baseline") or {}).get("gpu_vs_cpu_sample_mismatches")}
            base = base or d["value"]
            print(f"{frames:6d}  {form:7s}  {d['value']:.4e}   {d['ms_per_step']:10.1f}   {parity}   ({100 * d['value'] / base:.1f} % of single)")
            sys.stdout.flush()


if __name__ == "__main__":
    main()
