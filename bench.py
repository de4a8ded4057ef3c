#!/usr/bin/env python3
"""bench.py — Hamming distances/sec of the all-vs-all loop search (BASELINE.json's metric) on N MI355X GPUs.

One "step" = one full pass of the hot path over the workload: every frame as query against every stored frame at
least min_gap older (lcm_all_vs_all_argmin -> the gfx950 pair-scoring kernel: per-query min AND argmin, per-pair record
and index checksum), inputs resident in HBM, plus — for N > 1 — the gather of the per-shard records (RCCL over xGMI).

  N = 1 : BASELINE.json configs[1] — 1000 frames x 2000 x 256-bit descriptors, min_gap 30 (470,935 pairs,
          1.88e12 distances per step), through one lcm_handle.
  N > 1 : WEAK scaling of the same workload at every N — the frame count is raised so that every device still scores
          ~470,935 pairs per step; stored frames are sharded cyclically by frame (device = position mod N), every device
          sees every query frame.  Two launch forms, same arithmetic, same bytes:
            * `python3 bench.py --gpus N` (WORLD_SIZE unset): ONE process, the product's own multi-GPU path behind the
              C ABI — lcm_group_create(N), lcm_group_all_vs_all_argmin: RCCL all-gather of the shard arenas, one host
              thread + one search per device, grouped ncclSend / ncclRecv of records and index checksums to the first
              device, device merge, one download.  `--gpus 1 --force-group` runs N = 1 through the same path (same
              value, byte-identical records and checksums as the plain line); `--loopback` rehearses any N as N shards
              on ONE device (exchange steps as device-local copies: NOT a scaling measurement).
            * `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (WORLD_SIZE set): one process per
              GPU, torch.distributed (backend "nccl" = RCCL) all-gather of the records.
          (`--workload cfg3` runs BASELINE.json configs[2], 10000 x 2000, strong-scaled, instead.)

Prints ONE JSON line (rank 0).  `roofline` is the HBM view the contract asks for (algorithmic bytes / kernel time);
`roofline_valu` is the roofline that actually binds this integer path;
`cpu_baseline` is the oracle's tuned CPU path timed on this box's host cores on a bounded sample of the same pairs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Hamming distances/sec (256-bit ORB) for all-vs-all loop search, 1/2/4/8 GPUs"
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU roofline of the 16-instruction minimum (8 v_xor_b32 + 8 v_bcnt_u32_b32 per distance) on gfx950, from measured
# issue behaviour (tools/valu_class.hip, tools/prio_bench; profiles/r01_valu_class.txt, r03_prio_bench.txt, r03_valu_issue.json):
# v_bcnt_u32_b32 is a quarter-rate instruction — a SIMD issues at most one per quad-cycle — and a half-rate v_xor_b32 of
# ANOTHER wave can issue in the same quad-cycle.  The popcounts alone therefore bound a distance: 8 quad-cycles = 32
# SIMD-cycles per 64 distances; 256 CUs x 4 SIMDs at 2.4 GHz.
VALU_PEAK_DIST_PER_S = 256 * 4 * 64 / 32.0 * 2.4e9
# rounds 1-2 priced every instruction alone in its issue slot (xor 2 + bcnt 4 cycles = 48 per 64 distances): kept for
# comparison across rounds — the priority-steered inner loop of round 3 runs ABOVE it
VALU_SERIAL_DIST_PER_S = 256 * 4 * 64 / 48.0 * 2.4e9
# dense matrix-core peaks (MI355X_MICROARCH.md: bf16 ~2.5 PF dense; int8 = 2x bf16 per clock, fp4 = 4x)
MFMA_I8_PEAK_OPS = 5.0e15
MFMA_FP4_PEAK_OPS = 10.0e15

WORKLOADS = {
    # name: (frames, descriptors per frame, description)
    "cfg1": (100, 500, "cfg1: 100 frames x 500 x 256-bit descriptors (reference's CPU-runnable plumbing case)"),
    "cfg2": (1000, 2000, "cfg2: 1000 frames x 2000 x 256-bit ORB descriptors, all-vs-all loop search, min_gap 30"),
    "cfg3": (10000, 2000, "cfg3: 10000 frames x 2000 descriptors, database sharded by frame across the GPUs"),
    "cfg4": (5000, 2000, "cfg4: 5000 frames x 2000 descriptors, fused on-device filter + loop-test counts"),
    "cfg5": (20000, 2000, "cfg5: 20000 frames x 2000 descriptors (use --mode stream: per-frame append + query)"),
}


_REAL_STDOUT = None


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def host_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_sample(entry, pkg, fs, gap, offs, got, seconds, threads, shard=None, got_idx=None, max_pairs=None):
    """The oracle's tuned CPU path on a random sample of the workload's eligible pairs sized for ~`seconds` of CPU work;
    `got` (optional) = the GPU's records in single-device order, compared on the same pairs (0 mismatches required).
    shard = (rank, world): sample only pairs whose stored frame that rank owns; `got`/`offs` are then that shard's."""
    oracle = entry.load_oracle()
    oracle.build()
    n_frames = fs.n_frames
    rng = np.random.default_rng(123)
    qs = rng.integers(gap, n_frames, size=262144 if max_pairs is None else max_pairs)
    ts = np.array([rng.integers(0, q - gap + 1) for q in qs])
    if shard is not None:
        r, w = shard
        ts = ts - (ts % w) + r
        ok = (ts >= 0) & (ts <= qs - gap)
        qs, ts = qs[ok], ts[ok]
    op = oracle.default_params(min_gap=gap)
    n_cal = min(4 * threads, len(qs))
    _, secs, isa = oracle.fast_score_pairs(fs.rows, fs.counts, qs[:n_cal], ts[:n_cal], op, threads)
    n_s = int(min(len(qs), max(n_cal, seconds / max(secs / n_cal, 1e-9))))
    cs, secs, isa = oracle.fast_score_pairs(fs.rows, fs.counts, qs[:n_s], ts[:n_s], op, threads)
    cpu_dist = int(np.sum(fs.counts[qs[:n_s]].astype(np.int64) * fs.counts[ts[:n_s]].astype(np.int64)))
    # the scalar restatement (Part 1 of the oracle: the plain, deliberately untuned statement of the rules, one thread,
    # ~1 s per 2000 x 2000 pair) on three of the same pairs: the semantic reference's own speed, and a check that the
    # tuned path's records are its records
    n_sc = min(3, n_s)
    t_sc = time.perf_counter()
    sc_recs = [oracle.pair_score(fs.frame(int(q)), fs.frame(int(t)), op) for q, t in zip(qs[:n_sc], ts[:n_sc])]
    t_sc = time.perf_counter() - t_sc
    sc_dist = int(np.sum(fs.counts[qs[:n_sc]].astype(np.int64) * fs.counts[ts[:n_sc]].astype(np.int64)))
    scalar = {"value": sc_dist / max(t_sc, 1e-9), "unit": "distances/s", "cores": 1, "pairs": int(n_sc),
              "equals_tuned_path": bool(all(sc_recs[k] == cs[k] for k in range(n_sc)))}
    out = {"value": cpu_dist / secs, "unit": "distances/s", "cores": threads, "kind": "port", "scalar_oracle_1_thread": scalar,
           "sample": f"{n_s} random eligible pairs of the same workload ({cpu_dist:.3e} distances, {secs:.1f} s), "
                     f"oracle tuned path ({isa}, pthreads over pairs)"}
    if got is not None:
        local_t = ts[:n_s] if shard is None else (ts[:n_s] - shard[0]) // shard[1]
        idx = offs[qs[:n_s]].astype(np.int64) + local_t
        mismatch = int(np.sum(got[idx] != cs))
        if got_idx is not None:
            # match INDICES too: the per-pair checksum of the good matches' train indices (a subsample: the oracle's
            # index-tracking pass is run outside the timed sample)
            k = min(n_s, 4096)
            _, isum = oracle.fast_score_pairs_idx(fs.rows, fs.counts, qs[:k], ts[:k], op, threads)
            bad = int(np.sum(got_idx[idx[:k]] != isum))
            out["gpu_vs_cpu_index_checksum_mismatches"] = bad
            out["index_checksums_compared"] = int(k)
            mismatch += bad
        out["gpu_vs_cpu_sample_mismatches"] = mismatch
        if mismatch:
            print(f"PARITY FAILURE: {mismatch} of {n_s} sampled pairs differ from the CPU oracle", file=sys.stderr)
    return out


def oracle_spot_check(entry, fs, gap, offs, got, got_idx, n_pairs=200, seed=7):
    """A fixed-size parity sample (records + index checksums) against the CPU oracle — the checker, outside any timed region."""
    oracle = entry.load_oracle()
    oracle.build()
    rng = np.random.default_rng(seed)
    qs = rng.integers(gap, fs.n_frames, size=n_pairs)
    ts = np.array([rng.integers(0, q - gap + 1) for q in qs])
    cs, isum = oracle.fast_score_pairs_idx(fs.rows, fs.counts, qs, ts, oracle.default_params(min_gap=gap), host_cores())
    at = offs[qs].astype(np.int64) + ts
    bad = int(np.sum(got[at] != cs))
    bad_idx = int(np.sum(got_idx[at] != isum)) if got_idx is not None else None
    if bad or bad_idx:
        print(f"PARITY FAILURE: {bad} records / {bad_idx} index checksums of {n_pairs} sampled pairs differ from the CPU oracle", file=sys.stderr)
    return {"pairs": int(n_pairs), "record_mismatches": bad, "index_checksum_mismatches": bad_idx}


def kernel_name(packed, n_desc, argmin):
    return "k_score_rowlane<%d, 8, %d, false, %s>%s" % (
        256 if packed else (64 if n_desc <= 512 else 128 if n_desc <= 1024 else 192 if n_desc <= 1536 else 256),
        1 if argmin else 0, "true" if packed else "false", " (argmin)" if argmin else "")


def roofline_from_launches(pkg, infos, n_desc, argmin, traffic=None, traffic_source=None):
    """The two roofline blocks from the library's own HIP events (lcm_last_launch_info) of one or more individually timed
    steps of ONE device.  A packed bulk search runs as several (score, fold) chunk launches; consecutive chunks sit on
    two streams, so two score launches are in flight at any time.  `kernel_ms` is the dominant kernel's AVERAGE
    LAUNCH DURATION from events around every score launch on its own stream (what rocprofv3 --kernel-trace --stats
    reports as that kernel's average), `achieved` = algorithmic bytes per launch / that duration (the contract's formula,
    to the letter: ONE launch's own rate), `achieved_chip` = x launches in flight = what the chip moves while they run,
    `achieved_per_step` = algorithmic bytes per step / the step's kernel span (the cross-check)."""
    li = infos[-1]
    packed = (li.route == pkg.capi.ROUTE_PACKED)
    step_ms = float(np.mean([x.kernel_ms for x in infos]))
    if packed and li.score_launches:
        n_launch = int(li.score_launches)
        launch_ms = float(np.mean([x.score_ms_sum / max(x.score_launches, 1) for x in infos]))
        in_flight = int(li.launches_in_flight)
        fold_ms = float(np.mean([x.aux_kernel_ms for x in infos]))
    else:
        n_launch = max(int(li.launches), 1)
        launch_ms = step_ms / n_launch
        in_flight = 1
        fold_ms = None
    bytes_step = int(li.algo_bytes)
    bytes_launch = bytes_step / n_launch
    achieved = bytes_launch / (launch_ms * 1e-3) / 1e9
    per_step = bytes_step / (step_ms * 1e-3) / 1e9
    kern_rate = int(li.distances) / (step_ms * 1e-3)
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": kernel_name(packed, n_desc, argmin),
            "route": ("packed: query rows of a group of frames share full 2048-row workgroups; a launch = the group x a range of "
                      "stored slots (1 GiB of per-row scratch in two halves); records formed by k_finalize_bulk" if packed
                      else "one workgroup per (query frame, run of stored frames)"),
            "kernel_ms": launch_ms, "launches_per_step": n_launch, "launches_in_flight": in_flight,
            "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes_per_step": bytes_step,
            "step_kernel_ms": step_ms, "achieved_chip": achieved * in_flight, "achieved_per_step": per_step,
            "fold_kernels_ms_per_step": fold_ms,
            "note": "this path is VALU-bound by ~100x (0.016 algorithmic bytes per distance: 8 TB/s would feed 5e14 distances/s, the quarter-rate VALU pipe issues 4.9e12); see roofline_valu.  kernel_ms = "
                    "average duration of ONE score launch (HIP events around each on its stream; with 2 launches in flight "
                    "they overlap pairwise, so the launches' durations sum to more than step_kernel_ms); achieved = bytes per "
                    "launch / kernel_ms = one launch's own rate; achieved_chip = achieved x launches in flight; "
                    "achieved_per_step = bytes per step / step_kernel_ms (all score + fold kernels of a step, device clock): "
                    "the rate to compare across rounds.  traffic = L2-to-fabric bytes per score launch incl. Infinity-Cache "
                    "hits; fold_kernels_ms_per_step sums the folds' own durations, which include waiting for a free CU behind "
                    "the other stream's score kernel"}
    valu = {"bound": "valu", "achieved": kern_rate, "peak": VALU_PEAK_DIST_PER_S, "unit": "distances/s",
            "frac": kern_rate / VALU_PEAK_DIST_PER_S,
            "model": "quarter-rate pipe: 8 v_bcnt_u32_b32 per distance, at most one per quad-cycle per SIMD (32 SIMD-cycles per 64 "
                     "distances; the 8 half-rate v_xor_b32 issue beside them from other waves), 1024 SIMDs @ 2.4 GHz; "
                     "achieved = distances per step / step_kernel_ms",
            "serial_issue_peak": VALU_SERIAL_DIST_PER_S, "serial_issue_frac": kern_rate / VALU_SERIAL_DIST_PER_S,
            "serial_issue_model": "rounds 1-2's roofline: each instruction alone in its issue slot, v_xor_b32 2 + v_bcnt_u32_b32 4 "
                                  "cycles (48 per 64 distances); above 1.0 = xors issued beside popcounts"}
    vb = load_profile_json("valu_busy.json")
    if vb is not None:
        for k in ("quarter_rate_pipe_busy_frac", "quad_cycles_with_two_valu_issued_frac", "valu_issue_slots_busy_frac", "simd_cycles_per_64_distances"):
            valu[k] = vb.get(k)
        valu["counters_source"] = ("profiles/valu_busy.json (rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE of this "
                                   "command, NOT measured in this run): quarter_rate_pipe_busy_frac = the kernel's quarter-rate VALU "
                                   "instructions (all but its 8 xors per distance) over its quad-cycles; "
                                   "quad_cycles_with_two_valu_issued_frac = SQ_ACTIVE_INST_VALU2 over quad-cycles; "
                                   "valu_issue_slots_busy_frac = quad-cycles with at least one VALU issue")
    return roof, valu


def load_profile_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p))
    except Exception:
        return None


def lookup_traffic(wl, world, n_frames, variant, packed):
    t = load_profile_json("hbm_traffic.json")
    if t is None:
        return None, None
    if (t.get("workload") == wl and t.get("n_gpus") == world and t.get("frames") == n_frames
            and t.get("kernel_variant", 0) == variant and bool(t.get("packed", False)) == packed):
        return t.get("hbm_bytes_per_launch"), ("profiles/hbm_traffic.json: rocprofv3 --pmc passes of this same command, per score launch like "
                                               "`achieved`, NOT measured in this run; reads = 32 x TCC_EA0_RDREQ_32B + 64 x _64B + 128 x _128B "
                                               "(exact, calibrated on known byte counts: FETCH_SIZE prices a 128-byte request at 64 bytes), "
                                               "writes = WRITE_SIZE")
    return None, None


# ---------------------------------------------------------------------------------------------------------------------
# streaming (online) mode
# ---------------------------------------------------------------------------------------------------------------------
def stream_mode(args, pkg, torch, dist, fs, world, rank, local_rank, dev, multi, wl_desc, seed, entry):
    """Online mode: frames arrive one at a time as HOST rows and are scored in micro-batches of --stream-batch frames
    (lcm_query_submit_batch: pinned staging, ONE H2D of the batch's rows, ONE launch over this rank's shard, one D2H
    of the 8-byte records — all enqueued, no host wait), then the frames this rank owns are appended (pinned ring +
    hipMemcpyAsync on the copy stream, overlapping the launch just submitted), then the PREVIOUS batch is collected: one
    batch is always in flight while the host prepares the next.  A batch spans fewer ids than min_gap, so the records
    are exactly those of frame-by-frame processing.  One step = one pass over the whole sequence (database empty at the
    start).  Scores are gathered once per step (RCCL) when N > 1."""
    p = pkg.default_params()
    p.min_gap = args.gap
    B = max(1, min(args.stream_batch, 16))
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    m = pkg.Matcher(p, device=local_rank, stream=stream.cuda_stream)
    m.set_kernel_variant(args.variant if args.variant in (4, 5) else 0)     # 4 / 5: opt-in matrix-core variants online
    m.set_tuning(pkg.capi.TUNE_ONLINE_STREAMS, args.online_streams)
    m.set_tuning(pkg.capi.TUNE_ONLINE_SPLIT, args.online_split)
    n_frames = fs.n_frames
    assert B == 1 or int(fs.ids[min(B, n_frames) - 1] - fs.ids[0]) < max(args.gap, 1), "a batch must span fewer ids than min_gap"
    owned_n = len(pkg.sharding.owned_positions(n_frames, rank, world))
    m.reserve(owned_n, fs.stride_rows)
    frames = [np.ascontiguousarray(fs.frame(f)) for f in range(n_frames)]
    ids = [int(x) for x in fs.ids]
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    er = pkg.sharding.shard_eligible_counts(fs.ids, args.gap, rank, world)
    cap = int(er.max()) * B if len(er) else 1

    depth = max(1, min(args.stream_depth, 4))                   # batches in flight (the library has 4 query slots)

    def one_pass():
        m.clear()
        out = []
        pending = []
        for f0 in range(0, n_frames, B):
            fr = range(f0, min(f0 + B, n_frames))
            pending.append(m.query_submit_batch([frames[f] for f in fr], [ids[f] for f in fr]))   # enqueued; the host moves on
            for f in fr:
                if f % world == rank:
                    m.append(ids[f], frames[f])                 # copy stream: overlaps the launch just submitted
            if len(pending) == depth:
                out.append(m.query_collect_batch(pending.pop(0), cap)[0])  # results of the OLDEST batch in flight
        for t in pending:
            out.append(m.query_collect_batch(t, cap)[0])
        m.sync()
        return np.concatenate(out) if out else np.zeros(0, pkg.capi.SCORE_DTYPE)

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    m.online_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        local = one_pass()
        if multi:
            t = torch.from_numpy(local.view(np.int64).copy()).to(cdev)
            shards = pkg.sharding.all_gather_scores(t, len(local))
        else:
            shards = [local]
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    t1 = time.perf_counter()
    st = m.online_stats()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=cdev)
    tot = torch.tensor([int(st.distances), int(st.pairs), int(st.algo_bytes)], dtype=torch.int64, device=cdev)
    if multi:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_dist, total_pairs, total_bytes = (int(x) // args.steps for x in tot.tolist())     # per step, all ranks
    merged, moffs = pkg.sharding.merge_shard_scores(shards, fs.ids, args.gap)
    assert len(merged) == pkg.synth.n_pairs_all_vs_all(n_frames, args.gap) == total_pairs
    if rank == 0:
        cpu = None
        if not multi and args.cpu_seconds > 0:
            cpu = cpu_baseline_sample(entry, pkg, fs, args.gap, moffs, merged, args.cpu_seconds, args.cpu_threads or host_cores())
        emit(stream_line(args, st, elapsed, total_dist, total_pairs, world, world, n_frames, fs, seed, B, depth, wl_desc, cpu,
                         "process per GPU (torch.distributed)" if multi else "one lcm_handle"))
    m.close()
    if multi:
        dist.destroy_process_group()


def stream_line(args, st, elapsed, total_dist, total_pairs, n_gpus, world, n_frames, fs, seed, B, depth, wl_desc, cpu, form, extra=None):
    # One device's (rank 0's / the slowest shard's) launches over all timed steps.  With one stream per query slot
    # consecutive launches overlap, so the sum of their event-bracketed durations can exceed the wall time: the roofline
    # then uses the wall time (conservative).
    kern_sum_s = st.kernel_ms * 1e-3
    kern_s = min(kern_sum_s, elapsed)
    work_share = world if form.startswith("one process, lcm_group") else 1      # group stats sum the work of all devices
    kern_rate = int(st.distances) / work_share / max(kern_s, 1e-12)
    achieved = int(st.algo_bytes) / work_share / max(kern_s, 1e-12) / 1e9
    out = {
        "metric": METRIC, "value": total_dist * args.steps / elapsed, "unit": "distances/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak" if args.workload == "auto" else "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "STREAMING (online append + micro-batched queries, host rows over PCIe): " + wl_desc,
                   "frames": n_frames, "descriptors_per_frame": fs.stride_rows, "min_gap": args.gap,
                   "pairs_per_step": total_pairs, "distances_per_step": total_dist, "seed": seed,
                   "stream_batch": B, "batches_in_flight": depth, "sharding": "cyclic by frame" if world > 1 else "none",
                   "form": form, "kernel_variant": args.variant if args.variant in (4, 5) else 0},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "k_score_rowlane (online launches: score + k_finalize_pairs in split mode)",
                     "kernel_ms": st.kernel_ms / max(int(st.launches) // work_share, 1), "launches": int(st.launches),
                     "kernel_ms_total": st.kernel_ms, "algorithmic_bytes_total": int(st.algo_bytes),
                     "note": "HIP events around every online launch, summed by the library (lcm_online_stats_read; for a group: "
                             "work summed over the devices, kernel time of the slowest); per-device figures; VALU-bound path, "
                             "see roofline_valu"},
        "roofline_valu": {"bound": "valu", "achieved": kern_rate, "peak": VALU_PEAK_DIST_PER_S, "unit": "distances/s",
                          "frac": kern_rate / VALU_PEAK_DIST_PER_S, "serial_issue_peak": VALU_SERIAL_DIST_PER_S,
                          "serial_issue_frac": kern_rate / VALU_SERIAL_DIST_PER_S},
        "device_busy_frac": kern_s / elapsed, "launch_time_sum_over_wall": kern_sum_s / elapsed,
        "online_streams": "one per query slot (consecutive launches overlap)" if args.online_streams else "handle's stream only",
        "cpu_baseline": cpu,
        "note": "PCIe-inclusive online rate (value); the headline metric is the batch mode (inputs resident in HBM)"}
    if extra:
        out.update(extra)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# extra blocks of the default N = 1 line
# ---------------------------------------------------------------------------------------------------------------------
def cfg4_fused_extra(pkg, torch, dev, local_rank, gap, selective):
    """One step of BASELINE.json configs[3] at full size on this GPU: 5000 x 2000, lcm_all_vs_all_loops (score kernels,
    scores stay in HBM, loop-test kernels, candidate compaction).  selective = the synthetic variant on which the README
    filter rejects unrelated pairs (synth.make_frames_selective): candidates are then sparse, as in a real sequence."""
    n_frames, n_desc, desc = WORKLOADS["cfg4"]
    seed = pkg.synth.BASE_SEED + 4
    fs = (pkg.synth.make_frames_selective if selective else pkg.synth.make_frames)(n_frames, n_desc, seed=seed)
    d_rows = torch.from_numpy(fs.rows).to(dev)
    stream = torch.cuda.current_stream(dev)
    p = pkg.default_params()
    p.min_gap = gap
    m = pkg.Matcher(p, device=local_rank, stream=stream.cuda_stream)
    m.reserve(n_frames, n_desc)
    fb = fs.stride_rows * 32
    for f in range(n_frames):
        m.append_device(int(fs.ids[f]), d_rows.data_ptr() + f * fb, int(fs.counts[f]))
    m.sync()
    n_all = pkg.synth.n_pairs_all_vs_all(n_frames, gap)
    buf = np.zeros(1 << 16 if selective else n_all, pkg.capi.CANDIDATE_DTYPE)   # (the default variant: nearly every pair is one)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    cands, pairs = m.all_vs_all_loops(out=buf)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    info = m.launch_info()
    n_places = max(1, n_frames // 4)
    revisit = int(np.sum((cands["current_frame_id"] % n_places) == (cands["matched_frame_id"] % n_places))) if len(cands) else 0
    out = {"workload": desc, "synthetic_variant": "selective (30 shared pool descriptors per frame)" if selective else "default (place-structured)",
           "frames": n_frames, "descriptors_per_frame": n_desc, "min_gap": gap, "seed": seed,
           "api": "lcm_all_vs_all_loops", "steps": 1, "ms_per_step": (t1 - t0) * 1e3, "pairs": int(pairs),
           "distances": int(info.distances), "value": int(info.distances) / (t1 - t0), "unit": "distances/s",
           "score_kernel_ms": info.kernel_ms, "loop_test_ms": info.aux_kernel_ms, "loop_candidates": int(len(cands)),
           "candidate_fraction": len(cands) / max(int(pairs), 1), "candidates_that_are_revisits_of_a_place": revisit,
           "candidate_bytes_to_host": int(len(cands)) * 24,
           "loop_test_roofline": {"bound": "hbm", "bytes": int(pairs) * 16 + int(len(cands)) * 24,
                                  "achieved_GBps": (int(pairs) * 16 + int(len(cands)) * 24) / max(info.aux_kernel_ms, 1e-6) / 1e6,
                                  "peak_GBps": HBM_PEAK_GBPS,
                                  "note": "k_loop_count + k_block_scan + count read-back + k_loop_emit: each pair's record read twice"}}
    m.close()
    del d_rows
    return out


def cfg3_whole_extra(entry, pkg, torch, dev, local_rank, gap):
    """north_star's target shape on ONE GPU: 10000 frames x 2000 descriptors, one lcm_all_vs_all_argmin pass (49.7 M pairs,
    1.99e14 distances, ~69 s), with a 200-pair CPU-oracle sample of records and index checksums."""
    n_frames, n_desc, desc = WORKLOADS["cfg3"]
    seed = pkg.synth.BASE_SEED + 3
    fs = pkg.synth.make_frames(n_frames, n_desc, seed=seed)
    d_rows = torch.from_numpy(fs.rows).to(dev)
    stream = torch.cuda.current_stream(dev)
    p = pkg.default_params()
    p.min_gap = gap
    m = pkg.Matcher(p, device=local_rank, stream=stream.cuda_stream)
    m.reserve(n_frames, n_desc)
    fb = fs.stride_rows * 32
    for f in range(n_frames):
        m.append_device(int(fs.ids[f]), d_rows.data_ptr() + f * fb, int(fs.counts[f]))
    n, offs = m.all_vs_all_plan()
    scores = torch.zeros(n, dtype=torch.int64, device=dev)
    isums = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    m.all_vs_all_argmin(scores.data_ptr(), n, isums.data_ptr())
    m.sync()
    t1 = time.perf_counter()
    li = m.launch_info()
    got = scores.cpu().numpy().view(pkg.capi.SCORE_DTYPE)
    got_idx = isums.cpu().numpy().view(np.uint32)
    roof, valu = roofline_from_launches(pkg, [li], n_desc, True)
    out = {"workload": "cfg3 WHOLE on one GPU (BASELINE.json north_star's target shape): 10000 frames x 2000 descriptors, min_gap %d" % gap,
           "api": "lcm_all_vs_all_argmin", "seed": seed, "steps": 1, "ms_per_step": (t1 - t0) * 1e3, "pairs": int(n),
           "distances": int(li.distances), "value": int(li.distances) / (t1 - t0), "unit": "distances/s",
           "hbm_frac": roof["frac"], "hbm_achieved_GBps": roof["achieved"], "valu_frac": valu["frac"], "valu_serial_issue_frac": valu["serial_issue_frac"],
           "score_launches": int(li.score_launches), "step_kernel_ms": li.kernel_ms,
           "cpu_oracle_sample": oracle_spot_check(entry, fs, gap, offs, got, got_idx, n_pairs=200)}
    m.close()
    del d_rows, scores, isums
    return out


def pair_mode_extra(pkg, fs):
    """matchFeatures as one call (LoopClosingSystem::matchFeatures, include/loop_closing.hpp:40): wall clock of lcm_match_features
    for two 2000-row frames of the workload, host rows in pageable memory, through this process's ctypes binding (a C++
    caller saves ~7 us: tools/pair_latency.cpp)."""
    q, t = np.ascontiguousarray(fs.frame(fs.n_frames - 1)), np.ascontiguousarray(fs.frame(3))
    with pkg.Matcher() as m:
        for _ in range(5):
            m.match_features(q, t)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter()
            good, md = m.match_features(q, t)
            ts.append(time.perf_counter() - t0)
        li = m.launch_info()
    return {"api": "lcm_match_features", "rows": [int(len(q)), int(len(t))], "median_us": float(np.median(ts) * 1e6),
            "p10_us": float(np.percentile(ts, 10) * 1e6), "score_kernel_us": li.kernel_ms * 1e3, "workgroups": int(li.workgroups),
            "good_matches": int(len(good)), "min_dist": int(md)}


def group_rehearsal_extra(pkg, fs, gap, expect, expect_idx, world, loopback):
    """cfg2 through lcm_group_all_vs_all_argmin: a group of ONE device (ncclCommInitAll, all-gather of the shard arena, search,
    gather, device merge, one download) or W shards on this ONE GPU (lcm_group_create_loopback: exchange steps as device-
    local copies — rank-major query buffer, W host threads, W concurrent searches, gatherv offsets, device merge).  The
    merged records AND index checksums must equal the single handle's bytes."""
    p = pkg.default_params()
    p.min_gap = gap
    kw = dict(n_devices=world, loopback_device=0) if loopback else dict(n_devices=world)
    with pkg.Group(p, **kw) as g:
        g.reserve(fs.n_frames, fs.stride_rows)
        for f in range(fs.n_frames):
            g.append(int(fs.ids[f]), fs.frame(f))
        sc, ix, _ = g.all_vs_all_argmin()                       # warm-up (plans, buffers, arena all-gather)
        first = g.info()
        t0 = time.perf_counter()
        sc, ix, _ = g.all_vs_all_argmin(sc, ix)
        t1 = time.perf_counter()
        gi = g.info()
    return {"api": ("lcm_group_create_loopback + " if loopback else "") + "lcm_group_all_vs_all_argmin",
            "n_devices": 1, "shards": world, "rccl_ranks": int(gi.rccl_ranks), "ms": (t1 - t0) * 1e3, "pairs": int(gi.pairs),
            "value": int(gi.distances) / (t1 - t0), "unit": "distances/s", "kernel_ms_max": gi.kernel_ms_max,
            "gather_merge_ms": gi.gather_merge_ms, "download_ms": gi.download_ms,
            "arena_allgather": {"first_search_bytes_per_device": int(first.gathered_query_bytes), "first_search_ms": first.allgather_ms,
                                "second_search_skipped": bool(gi.arena_gather_skipped)},
            "equals_single_handle_records": bool(expect is not None and len(sc) == len(expect) and np.array_equal(sc, expect)),
            "equals_single_handle_index_checksums": bool(expect_idx is not None and np.array_equal(ix, expect_idx)),
            "note": ("rehearsal of the W = %d index arithmetic on one device, RCCL's transport replaced by device-local copies; "
                     "not a scaling measurement" % world) if loopback else "RCCL communicator of one rank"}


# ---------------------------------------------------------------------------------------------------------------------
# N devices in ONE process through lcm_group_* (the product's own multi-GPU path)
# ---------------------------------------------------------------------------------------------------------------------
def group_mode(args, pkg, torch, entry):
    n_dev_seen = pkg.load_library().lcm_device_count()
    W = args.gpus
    loopback = bool(args.loopback)
    if not loopback and W > n_dev_seen:
        raise SystemExit(f"--gpus {W} but this process sees {n_dev_seen} HIP device(s) (use --loopback to rehearse {W} shards on one)")
    wl = args.workload
    if wl == "auto":
        wl = "cfg2"
    base_frames, n_desc, wl_desc = WORKLOADS[wl]
    n_desc = args.desc or n_desc
    scaling = "weak"
    if args.frames:
        n_frames = args.frames
        wl_desc += f" [frame count overridden: {n_frames}]"
        scaling = "strong"
    elif args.workload == "auto" and W > 1:
        per_dev = pkg.synth.n_pairs_all_vs_all(base_frames, args.gap)
        n_frames = pkg.synth.frames_for_pairs(per_dev * W, args.gap)         # weak scaling: ~cfg2's pair count per device
        wl_desc = (f"cfg2 weak-scaled to {W} devices: {n_frames} frames x {n_desc} descriptors "
                   f"(~{per_dev} pairs per device per step), cyclic frame sharding, min_gap {args.gap}")
    else:
        n_frames = base_frames
        scaling = "weak" if W == 1 else "strong"
    seed = pkg.synth.BASE_SEED + 2
    fs = pkg.synth.make_frames(n_frames, n_desc, seed=seed)
    p = pkg.default_params()
    p.min_gap = args.gap
    kw = dict(n_devices=W, loopback_device=0) if loopback else dict(n_devices=W, peer_copies=bool(args.peer_copies))
    g = pkg.Group(p, **kw)
    transport = g.transport
    g.set_tuning(pkg.capi.TUNE_PACKED, args.packed)
    if args.mode == "stream":
        return group_stream_mode(args, pkg, entry, g, fs, W, loopback, wl_desc, seed)
    g.reserve(n_frames, n_desc)
    for f in range(n_frames):
        g.append(int(fs.ids[f]), fs.frame(f))              # host rows -> the owner device's arena (pinned ring + copy stream)
    g.sync()                                                # inputs resident in HBM when the timed region starts
    argmin_api = (args.variant == 1)
    if not argmin_api:
        g.set_kernel_variant(args.variant)
    sc = ix = None

    def step():
        nonlocal sc, ix
        if argmin_api:
            sc, ix, offs_ = g.all_vs_all_argmin(sc, ix)
        else:
            sc, offs_ = g.all_vs_all()
        return offs_

    first_info = None
    offs = None
    for k in range(max(args.warmup, 1)):                    # (the first search builds plans and all-gathers the arenas)
        offs = step()
        if k == 0:
            first_info = g.info()
            first_gather = (int(first_info.gathered_query_bytes), first_info.allgather_ms)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    g.sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    gi = g.info()
    total_dist, total_pairs = int(gi.distances), int(gi.pairs)
    value = total_dist * args.steps / elapsed
    # per-device evidence from the LAST timed step: the slowest shard's launches are the roofline's dominant kernel
    shard_infos = [g.shard_launch_info(r) for r in range(W)]
    slow = int(np.argmax([x.kernel_ms for x in shard_infos]))
    roof, valu = roofline_from_launches(pkg, [shard_infos[slow]], n_desc, argmin_api)
    roof["device"] = slow
    # parity: byte-identity with the single handle at N = 1 (--force-group), oracle sample otherwise
    extra = {}
    cpu = None
    if W == 1 and not loopback:
        with pkg.Matcher(p) as m:
            m.set_tuning(pkg.capi.TUNE_PACKED, args.packed)
            m.reserve(n_frames, n_desc)
            for f in range(n_frames):
                m.append(int(fs.ids[f]), fs.frame(f))
            n1, _ = m.all_vs_all_plan()
            d, di = m.dev_alloc(max(n1, 1) * 8), m.dev_alloc(max(n1, 1) * 4)
            s1, i1 = np.zeros(n1, pkg.capi.SCORE_DTYPE), np.zeros(n1, np.uint32)
            if argmin_api:
                m.all_vs_all_argmin(d, n1, di)
            else:
                m.all_vs_all(d, n1)
            m.sync(); m.dev_download(d, s1)
            if argmin_api:
                m.dev_download(di, i1)
            m.dev_free(d); m.dev_free(di)
        extra["equals_single_handle"] = {"records": bool(np.array_equal(sc, s1)),
                                         "index_checksums": bool(np.array_equal(ix, i1)) if argmin_api else None}
        if args.cpu_seconds > 0:
            cpu = cpu_baseline_sample(entry, pkg, fs, args.gap, offs, sc, args.cpu_seconds, args.cpu_threads or host_cores(),
                                      got_idx=ix if argmin_api else None)
    else:
        extra["merged_vs_oracle_sample"] = oracle_spot_check(entry, fs, args.gap, offs, sc, ix if argmin_api else None, n_pairs=96)
    out = {
        "metric": METRIC, "value": value, "unit": "distances/s", "n_gpus": 1 if loopback else W, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": wl_desc, "frames": n_frames, "descriptors_per_frame": n_desc, "min_gap": args.gap,
                   "pairs_per_step": total_pairs, "distances_per_step": total_dist, "seed": seed,
                   "sharding": "cyclic by frame" if W > 1 else "none", "kernel_variant": args.variant,
                   "form": "one process, lcm_group over %d %s" % (W, "shards on ONE device (loopback rehearsal)" if loopback else "device(s)"),
                   "api": "lcm_group_all_vs_all_argmin" if argmin_api else "lcm_group_all_vs_all",
                   "outputs": "merged score records on the host" + (" + merged per-pair index checksums" if argmin_api else "")},
        "group": {"n_devices": int(gi.n_devices), "rccl_ranks": int(gi.rccl_ranks), "loopback": bool(gi.loopback), "transport": transport,
                  "kernel_ms_max": gi.kernel_ms_max, "kernel_ms_per_device": [gi.kernel_ms[r] for r in range(W)],
                  "pairs_per_device": [int(gi.shard_pairs[r]) for r in range(W)],
                  "gather_merge_ms": gi.gather_merge_ms, "download_ms": gi.download_ms,
                  "gathered_score_bytes": int(gi.gathered_score_bytes),
                  "arena_allgather": {"first_search_bytes_per_device": first_gather[0], "first_search_ms": first_gather[1],
                                      "timed_steps_skipped_it": bool(gi.arena_gather_skipped)},
                  "step_breakdown_ms": {"slowest_device_kernels": gi.kernel_ms_max, "gather_merge": gi.gather_merge_ms,
                                        "download": gi.download_ms,
                                        "host_and_other": elapsed / args.steps * 1e3 - gi.kernel_ms_max - gi.gather_merge_ms - gi.download_ms}},
        "roofline": roof, "roofline_valu": valu, "cpu_baseline": cpu,
    }
    if loopback:
        out["note"] = ("LOOPBACK REHEARSAL: %d shards on ONE GPU, RCCL's transport replaced by device-local copies — the N-device "
                       "code path end to end, NOT a scaling measurement (n_gpus says 1)" % W)
    out.update(extra)
    if W == 8 and args.workload == "auto" and not args.no_extras and not args.frames:
        out["extra"] = {"cfg3_sharded": group_cfg3_extra(pkg, entry, g, args.gap, W, loopback)}
    g.close()
    emit(out)


def group_cfg3_extra(pkg, entry, g, gap, W, loopback):
    """BASELINE.json configs[2] as named: 10000 frames x 2000 descriptors, database sharded across 8 devices, records
    gathered over RCCL — one step through lcm_group_all_vs_all_argmin (on the group the headline just used)."""
    n_frames, n_desc, desc = WORKLOADS["cfg3"]
    seed = pkg.synth.BASE_SEED + 3
    fs = pkg.synth.make_frames(n_frames, n_desc, seed=seed)
    g.clear()
    g.reserve(n_frames, n_desc)
    for f in range(n_frames):
        g.append(int(fs.ids[f]), fs.frame(f))
    g.sync()
    t0 = time.perf_counter()
    sc, ix, offs = g.all_vs_all_argmin()
    t1 = time.perf_counter()
    gi = g.info()
    return {"workload": desc, "api": "lcm_group_all_vs_all_argmin", "seed": seed, "steps": 1, "ms_per_step": (t1 - t0) * 1e3,
            "includes": "plan build + arena all-gather (first search on this database)", "pairs": int(gi.pairs),
            "distances": int(gi.distances), "value": int(gi.distances) / (t1 - t0), "unit": "distances/s",
            "kernel_ms_per_device": [gi.kernel_ms[r] for r in range(W)], "gather_merge_ms": gi.gather_merge_ms,
            "download_ms": gi.download_ms, "arena_allgather_ms": gi.allgather_ms,
            "arena_allgather_bytes_per_device": int(gi.gathered_query_bytes), "loopback": bool(loopback),
            "cpu_oracle_sample": oracle_spot_check(entry, fs, gap, offs, sc, ix, n_pairs=96)}


def group_stream_mode(args, pkg, entry, g, fs, W, loopback, wl_desc, seed):
    """BASELINE.json configs[4] shape inside ONE process: frames arrive as host rows, every micro-batch goes to all devices
    through lcm_group_query_submit_batch (each device's submit on its own host thread, asynchronous), the batch's frames
    are appended to their owner devices, and the PREVIOUS batch is collected — up to --stream-depth group tickets in flight."""
    B = max(1, min(args.stream_batch, 16))
    depth = max(1, min(args.stream_depth, 4))
    g.set_tuning(pkg.capi.TUNE_ONLINE_STREAMS, args.online_streams)
    g.set_tuning(pkg.capi.TUNE_ONLINE_SPLIT, args.online_split)
    if args.variant in (4, 5):
        g.set_kernel_variant(args.variant)
    n_frames = fs.n_frames
    assert B == 1 or int(fs.ids[min(B, n_frames) - 1] - fs.ids[0]) < max(args.gap, 1), "a batch must span fewer ids than min_gap"
    g.reserve(n_frames, fs.stride_rows)
    frames = [np.ascontiguousarray(fs.frame(f)) for f in range(n_frames)]
    ids = [int(x) for x in fs.ids]
    e = pkg.sharding.eligible_counts(fs.ids, args.gap)
    cap = int(e.max()) * B if len(e) else 1

    def one_pass():
        g.clear()
        out, pending = [], []
        for f0 in range(0, n_frames, B):
            fr = range(f0, min(f0 + B, n_frames))
            pending.append((g.query_submit_batch([frames[f] for f in fr], [ids[f] for f in fr]), len(fr)))
            for f in fr:
                g.append(ids[f], frames[f])
            if len(pending) == depth:
                tk, nb = pending.pop(0)
                out.append(g.query_collect_batch(tk, cap, nb)[0])
        for tk, nb in pending:
            out.append(g.query_collect_batch(tk, cap, nb)[0])
        g.sync()
        return np.concatenate(out) if out else np.zeros(0, pkg.capi.SCORE_DTYPE)

    for _ in range(args.warmup):
        one_pass()
    g.online_stats(reset=True)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        merged = one_pass()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    st = g.online_stats()
    total_pairs = pkg.synth.n_pairs_all_vs_all(n_frames, args.gap)
    assert len(merged) == total_pairs == int(st.pairs) // args.steps
    total_dist = int(st.distances) // args.steps
    moffs = pkg.sharding.offsets_from_counts(e)
    cpu = None
    extra = {"note": "PCIe-inclusive online rate (value); the headline metric is the batch mode (inputs resident in HBM)"}
    if W == 1 and not loopback and args.cpu_seconds > 0:
        cpu = cpu_baseline_sample(entry, pkg, fs, args.gap, moffs, merged, args.cpu_seconds, args.cpu_threads or host_cores())
    else:
        extra["merged_vs_oracle_sample"] = oracle_spot_check(entry, fs, args.gap, moffs, merged, None, n_pairs=240)
    if loopback:
        extra["note"] = ("LOOPBACK REHEARSAL: %d shards on ONE GPU — the N-device streaming path end to end, NOT a scaling "
                         "measurement (n_gpus says 1); PCIe-inclusive online rate" % W)
    emit(stream_line(args, st, elapsed, total_dist, total_pairs, 1 if loopback else W, W, n_frames, fs, seed, B, depth, wl_desc, cpu,
                     "one process, lcm_group over %d %s, asynchronous group tickets" % (W, "shards on ONE device (loopback rehearsal)" if loopback else "device(s)"),
                     extra))
    g.close()


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="auto", help="auto | cfg1 | cfg2 | cfg3 | cfg4 | cfg5")
    ap.add_argument("--frames", type=int, default=0, help="override the frame count")
    ap.add_argument("--desc", type=int, default=0, help="override descriptors per frame")
    ap.add_argument("--gap", type=int, default=30)
    ap.add_argument("--variant", type=int, default=1, help="1 (default): the ARGMIN kernel through lcm_all_vs_all_argmin — "
                    "per-query min AND first-minimum train index, per-pair index checksum written; 0: distance-only kernel; "
                    "2 / 3: the train-row-per-lane mapping (A/B measurement)")
    ap.add_argument("--packed", type=int, default=-1, help="bulk search with query rows packed into full 2048-row workgroups "
                    "(LCM_TUNE_PACKED): -1 automatic (default), 0 never, 1 always")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline sample budget; 0 disables")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores this process may use")
    ap.add_argument("--mode", default="batch", help="batch (default: one all-vs-all pass per step) | stream (online: "
                    "per frame, score it against the database, then append it — BASELINE.json configs[4] shape)")
    ap.add_argument("--stream-batch", type=int, default=8, help="--mode stream: frames per micro-batch (1..16; 1 = frame by frame)")
    ap.add_argument("--stream-depth", type=int, default=2, help="--mode stream: micro-batches in flight before the oldest is collected (1..4)")
    ap.add_argument("--online-streams", type=int, default=1, help="--mode stream: 1 = one stream per query slot (default), 0 = the handle's stream only")
    ap.add_argument("--online-split", type=int, default=-1, help="--mode stream: query rows per lane of the split mode (1, 2, 4), 0 = never split, -1 = automatic")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra blocks of the default line (cfg4 fused steps, cfg3 whole, group rehearsals)")
    ap.add_argument("--force-group", action="store_true", help="N = 1 through the product's multi-GPU path (lcm_group_* with one device): "
                    "same value and bytes as the plain line")
    ap.add_argument("--loopback", action="store_true", help="rehearse --gpus N as N shards on ONE device (lcm_group_create_loopback): the "
                    "N-device code path end to end on a one-GPU box; NOT a scaling measurement")
    ap.add_argument("--peer-copies", action="store_true", help="--gpus N in one process: exchange steps as device-to-device copies over xGMI "
                    "(lcm_group_create_peer) instead of RCCL — what lcm_group_create falls back to when no communicator can be created")
    ap.add_argument("--force-dist", action="store_true", help="exercise the process-per-GPU code path (process group, all-gather) even at world size 1")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (CPU rehearsal of the process-per-GPU path)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries chat on fd 1 (RCCL prints a 5-line version banner when
    # a communicator is created), so fd 1 is pointed at stderr for the whole run and the JSON line goes to the saved fd.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)

    import torch  # first: the process then has ONE HIP runtime (torch's), which liblcm_hip.so binds to
    import torch.distributed as dist

    import __graft_entry__ as entry
    pkg = entry.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "WORLD_SIZE" in os.environ and world >= 1 and "RANK" in os.environ      # started by torch.distributed.run
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the matcher has no CPU fallback")
    if (not launched or world == 1) and (args.gpus > 1 or args.force_group or args.loopback) and not args.force_dist:
        # ONE process, N devices: the product's own multi-GPU path behind the C ABI (lcm_group_*)
        return group_mode(args, pkg, torch, entry)
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()          # rehearsal: ranks may share a card
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    # ---- workload ------------------------------------------------------------------------------------
    wl = args.workload
    if wl == "auto":
        wl = "cfg2"          # configs[1] ("1 MI355X") at N = 1, weak-scaled at every other N: one family for the scaling curve
    base_frames, n_desc, wl_desc = WORKLOADS[wl]
    n_desc = args.desc or n_desc
    fused = (wl == "cfg4")                      # configs[3]: filter + loop test on the device, only candidates leave HBM
    if args.frames:
        n_frames = args.frames
        wl_desc += f" [frame count overridden: {n_frames}]"
    elif args.workload == "auto" and world > 1:
        per_rank = pkg.synth.n_pairs_all_vs_all(base_frames, args.gap)
        n_frames = pkg.synth.frames_for_pairs(per_rank * world, args.gap)     # weak scaling
        wl_desc = (f"cfg2 weak-scaled to {world} GPUs: {n_frames} frames x {n_desc} descriptors "
                   f"(~{per_rank} pairs per rank per step), cyclic frame sharding, min_gap {args.gap}")
    else:
        n_frames = base_frames
    seed = pkg.synth.BASE_SEED + 2
    fs = pkg.synth.make_frames(n_frames, n_desc, seed=seed)

    if args.mode == "stream":
        return stream_mode(args, pkg, torch, dist, fs, world, rank, local_rank, dev, multi, wl_desc, seed, entry)

    # ---- inputs resident in HBM ----------------------------------------------------------------------
    d_rows = torch.from_numpy(fs.rows).to(dev)                 # (frames, stride, 32) uint8: the query stream
    d_counts = torch.from_numpy(fs.counts).to(dev)
    # An explicit torch stream (not the legacy default stream, whose handle is 0): the library enqueues its kernels on
    # it and torch.distributed orders the RCCL all-gather after them, because collectives wait on the CURRENT stream.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    p = pkg.default_params()
    p.min_gap = args.gap
    m = pkg.Matcher(p, device=local_rank, stream=stream.cuda_stream)
    argmin_api = (args.variant == 1)                 # headline: north_star's "per-query min/argmin + match count"
    m.set_kernel_variant(0 if argmin_api else args.variant)
    m.set_tuning(pkg.capi.TUNE_PACKED, args.packed)
    owned = pkg.sharding.owned_positions(n_frames, rank, world)
    m.reserve(len(owned), n_desc)
    frame_bytes = fs.stride_rows * 32
    for pos in owned:
        m.append_device(int(fs.ids[pos]), d_rows.data_ptr() + int(pos) * frame_bytes, int(fs.counts[pos]))
    if not multi:
        n_local, offs = m.all_vs_all_plan()
        q_args = dict()
    else:
        q_args = dict(d_query_rows=d_rows.data_ptr(), d_query_counts=d_counts.data_ptr(), q_ids=fs.ids,
                      q_stride_rows=fs.stride_rows)
        n_local, offs = m.all_vs_all_plan(**q_args)
    scores = torch.zeros(max(n_local, 1), dtype=torch.int64, device=dev)     # 8-byte lcm_score records
    idx_sums = torch.zeros(max(n_local, 1), dtype=torch.int32, device=dev)   # argmin kernel: per-pair index checksum
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    if multi:
        n_t = torch.tensor([n_local], dtype=torch.int64, device=cdev)
        lens = [torch.zeros_like(n_t) for _ in range(world)]
        dist.all_gather(lens, n_t)
        lens = [int(x.item()) for x in lens]
        cap = max(lens)
        send = torch.zeros(cap, dtype=torch.int64, device=dev)
        recv = torch.empty(world * cap, dtype=torch.int64, device=cdev)
    else:
        lens = [n_local]

    fused_out = {}
    fused_buf = np.zeros(max(n_local, 1), pkg.capi.CANDIDATE_DTYPE) if fused else None     # worst case: every pair

    def search(d_scores_ptr, n):
        if argmin_api:
            m.all_vs_all_argmin(d_scores_ptr, n, idx_sums.data_ptr(), **q_args)
        else:
            m.all_vs_all(d_scores_ptr, n, **q_args)

    def step():
        if fused:
            # lcm_all_vs_all_loops: score kernels -> scores stay in HBM -> loop-test kernels -> compacted candidates -> host
            fused_out["cands"], fused_out["pairs"] = m.all_vs_all_loops(out=fused_buf, **q_args)
        elif not multi:
            search(scores.data_ptr(), n_local)
        else:
            search(send.data_ptr(), cap)                          # kernel writes straight into the send buffer
            if args.backend == "nccl":
                dist.all_gather_into_tensor(recv, send)           # RCCL over xGMI: per-shard score records
            else:
                dist.all_gather(list(recv.view(world, cap).unbind(0)), send.cpu())

    def barrier():
        torch.cuda.synchronize(dev)
        m.sync()                                                  # (the library's second stream joins the handle's, but be explicit)
        if multi:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # (reading the previous launch's HIP events would sync; collect after the timed region instead)
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=cdev)
    if multi:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    infos = [m.launch_info()]                                     # HIP events around the LAST timed step's kernels
    loop_test_ms = infos[0].aux_kernel_ms if fused else None
    long_step = infos[0].kernel_ms > 2000.0      # cfg3-sized steps: the timed region's own launches are evidence enough
    # a few more individually timed steps for stable per-launch durations (outside the timed region)
    for _ in range(min(3, max(args.steps - 1, 0)) if not (fused or long_step) else 0):
        search((send if multi else scores).data_ptr(), cap if multi else n_local)
        infos.append(m.launch_info())
    info = infos[-1]
    packed = (info.route == pkg.capi.ROUTE_PACKED)

    local_dist = int(info.distances)
    # the same workload through the OTHER row-per-lane kernel, reported beside the headline number: distance-only when the
    # headline is the argmin kernel, and the other way round
    other_ms = None
    if args.variant in (0, 1) and not fused and not (multi and long_step):
        ms = []
        for _ in range(1 if long_step else 2):
            if argmin_api:
                m.all_vs_all((send if multi else scores).data_ptr(), cap if multi else n_local, **q_args)
            else:
                m.all_vs_all_argmin((send if multi else scores).data_ptr(), cap if multi else n_local, idx_sums.data_ptr(), **q_args)
            ms.append(m.launch_info().kernel_ms)
        other_ms = float(np.mean(ms))
        search((send if multi else scores).data_ptr(), cap if multi else n_local)      # leave the headline's outputs behind
        m.sync()

    # OPT-IN matrix-core variants (north_star rules MFMA out of the product path; these are measurements of what the
    # rule costs): same workload, same records — compared here — on v_mfma_i32_32x32x32_i8 / the block-scaled fp4 MFMA
    mfma = None
    if args.variant in (0, 1) and not fused and not multi and not args.no_extras:
        mfma = {}
        ref = scores.clone()
        ref_idx = idx_sums.clone()
        for v, name, peak in ((4, "int8", MFMA_I8_PEAK_OPS), (5, "fp4", MFMA_FP4_PEAK_OPS)):
            m.set_kernel_variant(v)
            ms = []
            for _ in range(3):
                m.all_vs_all(scores.data_ptr(), n_local, **q_args)
                ms.append(m.launch_info().kernel_ms)
            torch.cuda.synchronize(dev)
            t_ms = float(np.mean(ms[1:]))                        # the first call also builds the operand image
            ops = 2.0 * 256.0 * local_dist / (t_ms * 1e-3)       # one multiply-add per descriptor bit per distance
            mfma[name] = {"kernel_variant": v, "ms_per_pass": t_ms, "distances_per_s": local_dist / (t_ms * 1e-3),
                          "records_equal_to_headline": bool(torch.equal(scores, ref)),
                          "roofline": {"bound": "mfma", "achieved": ops / 1e12, "peak": peak / 1e12, "unit": "TOP/s",
                                       "frac": ops / peak}}
            if argmin_api:
                ms = []
                for _ in range(2):
                    m.all_vs_all_argmin(scores.data_ptr(), n_local, idx_sums.data_ptr(), **q_args)
                    ms.append(m.launch_info().kernel_ms)
                torch.cuda.synchronize(dev)
                a_ms = float(np.min(ms))
                mfma[name]["argmin"] = {"ms_per_pass": a_ms, "distances_per_s": local_dist / (a_ms * 1e-3),
                                        "records_equal_to_headline": bool(torch.equal(scores, ref)),
                                        "index_checksums_equal_to_headline": bool(torch.equal(idx_sums, ref_idx))}
        m.set_kernel_variant(0)
        search(scores.data_ptr(), n_local)
        m.sync()

    tot = torch.tensor([local_dist, int(info.pairs), int(info.algo_bytes)], dtype=torch.int64, device=cdev)
    if multi:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    total_dist, total_pairs, total_bytes = (int(x) for x in tot.tolist())
    value = total_dist * args.steps / elapsed

    # ---- parity spot check + CPU baseline (rank 0, N = 1 only) ---------------------------------------
    cpu = None
    merged_mismatch = None
    if multi:
        # merge check (outside the timed region): the gathered shards, un-permuted, must equal what a single device
        # would have written for a sample of query frames — here verified structurally (lengths, n_train fields)
        host = recv.view(world, cap).cpu().numpy()
        shards = [host[r, : lens[r]].view(pkg.capi.SCORE_DTYPE) for r in range(world)]
        merged, moffs = pkg.sharding.merge_shard_scores(shards, fs.ids, args.gap)
        exp = pkg.synth.n_pairs_all_vs_all(n_frames, args.gap)
        assert len(merged) == exp, (len(merged), exp)
        e = pkg.sharding.eligible_counts(fs.ids, args.gap)
        nt_expect = np.concatenate([fs.counts[: int(k)] for k in e]) if exp else np.zeros(0)
        assert np.array_equal(merged["n_train"].astype(np.int64), nt_expect.astype(np.int64)), "merged shard order is wrong"
        if rank == 0 and exp:
            merged_mismatch = oracle_spot_check(entry, fs, args.gap, moffs, merged, None, n_pairs=96)["record_mismatches"]
    if fused:
        fused_scores = m.last_bulk_scores()                  # what the fused call left in HBM (parity sample below)
    if rank == 0 and not multi and args.cpu_seconds > 0:
        torch.cuda.synchronize(dev)
        got = fused_scores if fused else scores.cpu().numpy().view(pkg.capi.SCORE_DTYPE)[:n_local]
        got_idx = idx_sums.cpu().numpy().view(np.uint32)[:n_local] if (argmin_api and not fused) else None
        cpu = cpu_baseline_sample(entry, pkg, fs, args.gap, offs, got, args.cpu_seconds, args.cpu_threads or host_cores(),
                                  got_idx=got_idx)

    if rank == 0:
        traffic, traffic_source = (None, None) if fused else lookup_traffic(wl, world, n_frames, args.variant, packed)
        if args.variant in (2, 3):
            roof, valu = roofline_from_launches(pkg, infos, n_desc, args.variant == 3, traffic, traffic_source)
            roof["kernel"] = "k_score_trainlane<%s, false>" % ("true" if args.variant == 3 else "false")
        else:
            roof, valu = roofline_from_launches(pkg, infos, n_desc, argmin_api and not fused, traffic, traffic_source)
        out = {
            "metric": METRIC, "value": value, "unit": "distances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if (args.workload == "auto" or world == 1) else "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": wl_desc, "frames": n_frames, "descriptors_per_frame": n_desc, "min_gap": args.gap,
                       "pairs_per_step": total_pairs, "distances_per_step": total_dist, "seed": seed,
                       "sharding": "cyclic by frame" if world > 1 else "none", "kernel_variant": args.variant,
                       "form": "process per GPU (torch.distributed)" if multi else "one lcm_handle",
                       "api": "lcm_all_vs_all_loops" if fused else ("lcm_all_vs_all_argmin" if argmin_api else "lcm_all_vs_all"),
                       "outputs": "score record (good-match count, min distance) per pair" +
                                  (" + checksum of the good matches' first-minimum train indices per pair" if argmin_api and not fused else "")},
            "roofline": roof, "roofline_valu": valu,
            ("distance_only_kernel" if argmin_api else "argmin_kernel"): None if other_ms is None else {
                "what": ("same workload through lcm_all_vs_all, kernel variant 0: best distance per query row only (what a "
                         "LoopCandidate needs; no train index); per-row scratch in 2-byte words" if argmin_api else
                         "same workload through lcm_all_vs_all_argmin: per-query min AND first-minimum train index "
                         "(8-row group keys in lane-private LDS + re-scan), per-pair index checksum written"),
                "step_kernel_ms": other_ms, "distances_per_s": local_dist / (other_ms * 1e-3)},
            "cpu_baseline": cpu,
        }
        if args.variant in (4, 5):
            # an explicit --variant 4 / 5 run: the dominant kernel is MFMA-bound, say so in the contract's roofline
            peak = MFMA_I8_PEAK_OPS if args.variant == 4 else MFMA_FP4_PEAK_OPS
            kern_ms = float(np.mean([x.kernel_ms for x in infos]))
            ops = 2.0 * 256.0 * local_dist / (kern_ms * 1e-3)
            out["roofline_hbm"] = out["roofline"]
            out["roofline"] = {"bound": "mfma", "achieved": ops / 1e12, "peak": peak / 1e12, "unit": "TOP/s", "frac": ops / peak,
                               "traffic": None, "kernel": "k_score_mfma" + ("_fp4" if args.variant == 5 else ""), "kernel_ms": kern_ms,
                               "note": "OPT-IN variant, not the product default (north_star: no MFMA); kernel_ms = score + fold kernels"}
        if mfma is not None:
            out["matrix_core_variants"] = {
                "note": "OPT-IN (lcm_set_kernel_variant 4 / 5), not the product path and not the headline: BASELINE.json's "
                        "north_star rules MFMA out; this is what that rule costs on this workload, bit-identical records",
                **mfma}
        if multi:
            out["merged_shards_vs_oracle_sample_mismatches"] = merged_mismatch
        if fused:
            out["fused"] = {"api": "lcm_all_vs_all_loops", "loop_test_ms": loop_test_ms,
                            "loop_candidates": int(len(fused_out["cands"])),
                            "note": "roofline covers the score kernels inside the fused call; ms_per_step covers score kernels + "
                                    "loop-test kernels + candidate download"}
        if not multi and args.workload == "auto" and not args.frames and not args.desc and not args.no_extras:
            single = scores.cpu().numpy().view(pkg.capi.SCORE_DTYPE)[:n_local].copy()
            single_idx = idx_sums.cpu().numpy().view(np.uint32)[:n_local].copy() if argmin_api else None
            m.close()
            del scores, d_rows, idx_sums
            torch.cuda.empty_cache()
            out["extra"] = {"pair_mode": pair_mode_extra(pkg, fs),
                            "group_of_one": group_rehearsal_extra(pkg, fs, args.gap, single, single_idx, 1, False),
                            "group_loopback_8": group_rehearsal_extra(pkg, fs, args.gap, single, single_idx, 8, True),
                            "cfg4_fused": cfg4_fused_extra(pkg, torch, dev, local_rank, args.gap, False),
                            "cfg4_fused_selective": cfg4_fused_extra(pkg, torch, dev, local_rank, args.gap, True),
                            "cfg3_whole_one_gpu": cfg3_whole_extra(entry, pkg, torch, dev, local_rank, args.gap)}
        emit(out)
    m.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
