#!/usr/bin/env python3
"""bench.py — Hamming distances/sec of the all-vs-all loop search (BASELINE.json's metric) on N MI355X GPUs.

One "step" = one full pass of the hot path over the workload: every frame as query against every stored frame at
least min_gap older (lcm_all_vs_all -> the gfx950 pair-scoring kernel), inputs resident in HBM, followed — for N > 1 —
by the RCCL all-gather of the per-shard 8-byte score records (the path's one real exchange step).

  N = 1 : BASELINE.json configs[1] — 1000 frames x 2000 x 256-bit descriptors, min_gap 30 (470,935 pairs,
          1.88e12 distances per step).
  N > 1 : weak scaling — the frame count is raised so that every rank still scores ~470,935 pairs per step; the
          stored frames are sharded cyclically by frame (rank = position mod N), every rank sees every query frame.
          (`--workload cfg3` runs BASELINE.json configs[2], 10000 x 2000, instead.)

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
# VALU roofline of the 16-instruction minimum (8 v_xor_b32 + 8 v_bcnt_u32_b32 per distance), from the measured
# issue costs on gfx950 (tools/valu_peak.hip, profiles/r01_valu_peak.txt): v_xor_b32 2 cycles and v_bcnt_u32_b32
# 4 cycles per wave64 instruction per SIMD => 48 SIMD-cycles per 64 distances; 256 CUs x 4 SIMDs at 2.4 GHz.
VALU_PEAK_DIST_PER_S = 256 * 4 * 64 / 48.0 * 2.4e9
VALU_NOMINAL_DIST_PER_S = 256 * 4 * 64 / 32.0 * 2.4e9      # if every one of the 16 instructions issued in 2 cycles
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


def cpu_baseline_sample(entry, pkg, fs, gap, offs, got, seconds, threads, shard=None, got_idx=None):
    """The oracle's tuned CPU path on a random sample of the workload's eligible pairs sized for ~`seconds` of CPU work;
    `got` (optional) = the GPU's records in single-device order, compared on the same pairs (0 mismatches required).
    shard = (rank, world): sample only pairs whose stored frame that rank owns; `got`/`offs` are then that shard's."""
    oracle = entry.load_oracle()
    oracle.build()
    n_frames = fs.n_frames
    rng = np.random.default_rng(123)
    qs = rng.integers(gap, n_frames, size=262144)
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
        # this rank, all timed steps.  With one stream per query slot consecutive launches overlap, so the sum of their
        # event-bracketed durations can exceed the wall time: the roofline then uses the wall time (conservative).
        kern_sum_s = st.kernel_ms * 1e-3
        kern_s = min(kern_sum_s, elapsed)
        kern_rate = int(st.distances) / max(kern_s, 1e-12)
        achieved = int(st.algo_bytes) / max(kern_s, 1e-12) / 1e9
        cpu = None
        if not multi and args.cpu_seconds > 0:
            cpu = cpu_baseline_sample(entry, pkg, fs, args.gap, moffs, merged, args.cpu_seconds,
                                      args.cpu_threads or host_cores())
        emit(({
            "metric": METRIC, "value": total_dist * args.steps / elapsed, "unit": "distances/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.workload == "auto" else "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "STREAMING (online append + micro-batched queries, host rows over PCIe): " + wl_desc,
                       "frames": n_frames, "descriptors_per_frame": fs.stride_rows, "min_gap": args.gap,
                       "pairs_per_step": total_pairs, "distances_per_step": total_dist, "seed": seed,
                       "stream_batch": B, "batches_in_flight": depth, "sharding": "cyclic by frame" if world > 1 else "none",
                       "kernel_variant": args.variant if args.variant in (4, 5) else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": "k_score_rowlane (online launches: score + k_finalize_pairs in split mode)",
                         "kernel_ms": st.kernel_ms / max(int(st.launches), 1), "launches": int(st.launches),
                         "kernel_ms_total": st.kernel_ms, "algorithmic_bytes_total": int(st.algo_bytes),
                         "note": "HIP events around every online launch, summed by the library (lcm_online_stats_read); "
                                 "VALU-bound path, see roofline_valu"},
            "roofline_valu": {"bound": "valu", "achieved": kern_rate, "peak": VALU_PEAK_DIST_PER_S, "unit": "distances/s",
                              "frac": kern_rate / VALU_PEAK_DIST_PER_S, "nominal_peak": VALU_NOMINAL_DIST_PER_S,
                              "nominal_frac": kern_rate / VALU_NOMINAL_DIST_PER_S},
            "device_busy_frac": kern_s / elapsed, "launch_time_sum_over_wall": kern_sum_s / elapsed,
            "online_streams": "one per query slot (consecutive launches overlap)" if args.online_streams else "handle's stream only",
            "cpu_baseline": cpu,
            "note": "PCIe-inclusive online rate (value); the headline metric is the batch mode (inputs resident in HBM)"}))
    m.close()
    if multi:
        dist.destroy_process_group()


def cfg4_fused_extra(pkg, torch, dev, local_rank, gap):
    """One step of BASELINE.json configs[3] at full size on this GPU: 5000 x 2000, lcm_all_vs_all_loops (score kernel,
    scores stay in HBM, k_loop_test, candidate compaction).  Reported as an `extra` block of the N = 1 line."""
    n_frames, n_desc, desc = WORKLOADS["cfg4"]
    seed = pkg.synth.BASE_SEED + 4
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
    m.sync()
    buf = np.zeros(pkg.synth.n_pairs_all_vs_all(n_frames, gap), pkg.capi.CANDIDATE_DTYPE)   # worst case: every pair
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    cands, pairs = m.all_vs_all_loops(out=buf)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    info = m.launch_info()
    out = {"workload": desc, "frames": n_frames, "descriptors_per_frame": n_desc, "min_gap": gap, "seed": seed,
           "api": "lcm_all_vs_all_loops", "steps": 1, "ms_per_step": (t1 - t0) * 1e3, "pairs": int(pairs),
           "distances": int(info.distances), "value": int(info.distances) / (t1 - t0), "unit": "distances/s",
           "score_kernel_ms": info.kernel_ms, "k_loop_test_ms": info.aux_kernel_ms, "loop_candidates": int(len(cands)),
           "k_loop_test_roofline": {"bound": "hbm", "bytes": int(pairs) * 8 + int(len(cands)) * 24,
                                    "achieved_GBps": (int(pairs) * 8 + int(len(cands)) * 24) / max(info.aux_kernel_ms, 1e-6) / 1e6,
                                    "peak_GBps": HBM_PEAK_GBPS}}
    m.close()
    del d_rows
    return out


def group_loopback_extra(pkg, fs, gap, expect, world=8):
    """cfg2 through the multi-device path's own code for W = 8 on this ONE GPU (lcm_group_create_loopback: 8 shards as 8
    matchers on the device, exchange steps as device-local copies): rank-major query buffer, 8 planning threads, 8
    concurrent shard searches, gatherv offsets, device merge.  The merged bytes must equal the single handle's."""
    p = pkg.default_params()
    p.min_gap = gap
    with pkg.Group(p, n_devices=world, loopback_device=0) as g:
        g.reserve(fs.n_frames, fs.stride_rows)
        for f in range(fs.n_frames):
            g.append(int(fs.ids[f]), fs.frame(f))
        g.all_vs_all()
        t0 = time.perf_counter()
        merged, _ = g.all_vs_all()
        t1 = time.perf_counter()
        gi = g.info()
    return {"api": "lcm_group_create_loopback + lcm_group_all_vs_all", "shards_on_this_gpu": world, "ms": (t1 - t0) * 1e3,
            "pairs": int(gi.pairs), "value": int(gi.distances) / (t1 - t0), "unit": "distances/s",
            "equals_single_handle_result": bool(expect is not None and len(merged) == len(expect) and np.array_equal(merged, expect)),
            "note": "rehearsal of the W = 8 index arithmetic on one device, RCCL's transport replaced by device-local copies; "
                    "not a scaling measurement"}


def group_extra(pkg, fs, gap, expect):
    """The same cfg2 search through lcm_group_* (the multi-GPU entry of the C ABI) with the devices this process can
    see used as ONE group of size 1: ncclCommInitAll, all-gather of the shard arena into the query buffer, search,
    gather, device merge, one download.  Wall time of the whole call, host rows already appended."""
    p = pkg.default_params()
    p.min_gap = gap
    with pkg.Group(p, n_devices=1) as g:
        g.reserve(fs.n_frames, fs.stride_rows)
        for f in range(fs.n_frames):
            g.append(int(fs.ids[f]), fs.frame(f))
        g.all_vs_all()                                           # warm-up (plan, buffers, RCCL channels)
        t0 = time.perf_counter()
        merged, _ = g.all_vs_all()
        t1 = time.perf_counter()
        gi = g.info()
    return {"api": "lcm_group_all_vs_all", "n_devices": 1, "ms": (t1 - t0) * 1e3, "pairs": int(gi.pairs),
            "value": int(gi.distances) / (t1 - t0), "unit": "distances/s", "kernel_ms_max": gi.kernel_ms_max,
            "gather_merge_ms": gi.gather_merge_ms, "download_ms": gi.download_ms,
            "allgather_query_bytes": int(gi.gathered_query_bytes),
            "equals_single_handle_result": bool(expect is not None and len(merged) == len(expect) and np.array_equal(merged, expect))}


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
    ap.add_argument("--no-extras", action="store_true", help="skip the extra blocks of the default N = 1 line (cfg4 fused step)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the N > 1 code path (process group, all-gather) even at world size 1")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (CPU rehearsal of the N > 1 path)")
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
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the matcher has no CPU fallback")
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
        # N = 8 is the configuration BASELINE.json's configs[2] / north_star name: 10000 x 2000 sharded over 8 GPUs.
        # Other N: cfg2 (configs[1], "1 MI355X"), weak-scaled so that every rank keeps cfg2's pair count.
        wl = "cfg3" if (world == 8 and not args.frames and args.mode == "batch") else "cfg2"
    base_frames, n_desc, wl_desc = WORKLOADS[wl]
    n_desc = args.desc or n_desc
    fused = (wl == "cfg4")                      # configs[3]: filter + loop test on the device, only candidates leave HBM
    if args.frames:
        n_frames = args.frames
        wl_desc += f" [frame count overridden: {n_frames}]"
    elif args.workload == "auto" and world > 1 and wl == "cfg2":
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
            # lcm_all_vs_all_loops: score kernel -> scores stay in HBM -> k_loop_test -> compacted candidates -> host
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
        if multi:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
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

    info = m.launch_info()                                        # HIP events around the LAST step's kernel(s)
    # Packed route (the automatic choice for 2000-row frames): a step is the score kernel + the fold kernel
    # (k_finalize_bulk).  kernel_ms covers both; the roofline is that of the DOMINANT kernel, so its own duration is
    # kernel_ms - fold when the call is one chunk (aux_kernel_ms times the last chunk's fold by itself).
    packed = (info.route == pkg.capi.ROUTE_PACKED)
    one_chunk = packed and info.launches == 2
    fold_ms = []

    def dominant_ms(li):
        if one_chunk and not fused:
            fold_ms.append(li.aux_kernel_ms)
            return li.kernel_ms - li.aux_kernel_ms
        return li.kernel_ms

    kernel_ms.append(dominant_ms(info))
    # a few more individually timed launches for a stable per-launch duration (outside the timed region)
    loop_test_ms = info.aux_kernel_ms if fused else None
    long_step = info.kernel_ms > 2000.0          # cfg3-sized steps: the timed region's own launches are evidence enough
    for _ in range(min(3, max(args.steps - 1, 0)) if not (fused or long_step) else 0):
        search((send if multi else scores).data_ptr(), cap if multi else n_local)
        kernel_ms.append(dominant_ms(m.launch_info()))
    kern_ms = float(np.mean(kernel_ms))

    local_dist = int(info.distances)
    # the same workload through the OTHER row-per-lane kernel, reported beside the headline number: distance-only when the
    # headline is the argmin kernel, and the other way round
    other_ms = None
    if args.variant in (0, 1) and not fused and not (multi and long_step):      # (N = 8 / cfg3: two more 8 s passes per rank buy nothing)
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
                # the argmin form on the matrix cores: first tile that reaches the best dot product, then an exact
                # XOR + popcount re-scan of that one tile -> the same index checksums as the headline kernel
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
        merged_mismatch = None
        if rank == 0 and exp:
            # ... and by value: a sample of merged records against the CPU oracle (checker only, not timed)
            oracle = entry.load_oracle()
            oracle.build()
            rng = np.random.default_rng(7)
            qs = rng.integers(args.gap, n_frames, size=96)
            ts = np.array([rng.integers(0, q - args.gap + 1) for q in qs])
            cs, _, _ = oracle.fast_score_pairs(fs.rows, fs.counts, qs, ts, oracle.default_params(min_gap=args.gap), host_cores())
            merged_mismatch = int(np.sum(merged[moffs[qs] + ts] != cs))
            if merged_mismatch:
                print(f"PARITY FAILURE: {merged_mismatch} of 96 sampled merged records differ from the CPU oracle", file=sys.stderr)
    if fused:
        fused_scores = m.last_bulk_scores()                  # what the fused call left in HBM (parity sample below)
    if rank == 0 and not multi and args.cpu_seconds > 0:
        torch.cuda.synchronize(dev)
        got = fused_scores if fused else scores.cpu().numpy().view(pkg.capi.SCORE_DTYPE)[:n_local]
        got_idx = idx_sums.cpu().numpy().view(np.uint32)[:n_local] if (argmin_api and not fused) else None
        cpu = cpu_baseline_sample(entry, pkg, fs, args.gap, offs, got, args.cpu_seconds, args.cpu_threads or host_cores(),
                                  got_idx=got_idx)

    if rank == 0:
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")      # PMC-derived bytes per launch, if collected
        if os.path.exists(tp):
            try:
                t = json.load(open(tp))
                if (t.get("workload") == wl and t.get("n_gpus") == world and t.get("frames") == n_frames
                        and t.get("kernel_variant", 0) == args.variant and bool(t.get("packed", False)) == packed and not fused):
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        algo_bytes_launch = int(info.algo_bytes)
        achieved = algo_bytes_launch / (kern_ms * 1e-3) / 1e9
        kern_rate = local_dist / (kern_ms * 1e-3)
        out = {
            "metric": METRIC, "value": value, "unit": "distances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if (args.workload == "auto" or world == 1) else "strong",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": wl_desc, "frames": n_frames, "descriptors_per_frame": n_desc, "min_gap": args.gap,
                       "pairs_per_step": total_pairs, "distances_per_step": total_dist, "seed": seed,
                       "sharding": "cyclic by frame" if world > 1 else "none", "kernel_variant": args.variant,
                       "api": "lcm_all_vs_all_loops" if fused else ("lcm_all_vs_all_argmin" if argmin_api else "lcm_all_vs_all"),
                       "outputs": "score record (good-match count, min distance) per pair" +
                                  (" + checksum of the good matches' first-minimum train indices per pair" if argmin_api and not fused else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": None if traffic is None else
                         "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command "
                         "(gfx950 corrections applied), NOT measured in this run",
                         "kernel": ("k_score_trainlane<%s, false>" % ("true" if args.variant == 3 else "false")) if args.variant in (2, 3) else
                                   "k_score_rowlane<%d, 8, %d, false, %s>%s" % (
                                       256 if packed else (64 if n_desc <= 512 else 128 if n_desc <= 1024 else 192 if n_desc <= 1536 else 256),
                                       1 if argmin_api and not fused else 0, "true" if packed else "false",
                                       " (argmin)" if argmin_api and not fused else ""),
                         "route": "packed: query rows of consecutive frames share full 2048-row workgroups; records formed by "
                                  "k_finalize_bulk" if packed else "one workgroup per (query frame, run of stored frames)",
                         "fold_kernel_ms": float(np.mean(fold_ms)) if fold_ms else None,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": algo_bytes_launch,
                         "note": "this path is VALU-bound by ~200x (0.016 algorithmic bytes per distance); see roofline_valu. "
                                 "traffic = L2-to-fabric bytes incl. Infinity-Cache hits; the argmin kernel's re-scan re-reads 8 "
                                 "of ~2000 rows per (query row, pair) and ~1/3 of those miss the XCD's L2; the packed route adds "
                                 "4 bytes written + read per (pair, query row) of scratch (8 KB per pair, < 1 % of HBM peak)"},
            "roofline_valu": {"bound": "valu", "achieved": kern_rate, "peak": VALU_PEAK_DIST_PER_S, "unit": "distances/s",
                              "frac": kern_rate / VALU_PEAK_DIST_PER_S,
                              "nominal_peak": VALU_NOMINAL_DIST_PER_S, "nominal_frac": kern_rate / VALU_NOMINAL_DIST_PER_S,
                              "nominal_model": "all 16 instructions at the guide's 2-cycle wave64 issue on a SIMD-32 "
                                               "(v_bcnt_u32_b32 measures 4.19: tools/valu_class.hip)",
                              "model": "8 v_xor_b32 (2 cyc) + 8 v_bcnt_u32_b32 (4 cyc) per 64 distances per SIMD, "
                                       "1024 SIMDs @ 2.4 GHz"},
            ("distance_only_kernel" if argmin_api else "argmin_kernel"): None if other_ms is None else {
                "what": ("same workload through lcm_all_vs_all, kernel variant 0: best distance per query row only (what a "
                         "LoopCandidate needs; no train index)" if argmin_api else
                         "same workload through lcm_all_vs_all_argmin: per-query min AND first-minimum train index "
                         "(8-row group keys in lane-private LDS + re-scan), per-pair index checksum written"),
                "kernel_ms": other_ms, "distances_per_s": local_dist / (other_ms * 1e-3)},
            "cpu_baseline": cpu,
        }
        if args.variant in (4, 5):
            # an explicit --variant 4 / 5 run: the dominant kernel is MFMA-bound, say so in the contract's roofline
            peak = MFMA_I8_PEAK_OPS if args.variant == 4 else MFMA_FP4_PEAK_OPS
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
            out["fused"] = {"api": "lcm_all_vs_all_loops", "k_loop_test_ms": loop_test_ms,
                            "loop_candidates": int(len(fused_out["cands"])),
                            "note": "roofline.kernel_ms is the score kernel inside the fused call; ms_per_step covers "
                                    "score kernel + k_loop_test + candidate download + host sort"}
        if not multi and args.workload == "auto" and not args.frames and not args.desc and not args.no_extras:
            single = scores.cpu().numpy().view(pkg.capi.SCORE_DTYPE)[:n_local].copy()
            m.close()
            del scores, d_rows
            torch.cuda.empty_cache()
            out["extra"] = {"group_of_one": group_extra(pkg, fs, args.gap, single),
                            "group_loopback_8": group_loopback_extra(pkg, fs, args.gap, single),
                            "cfg4_fused": cfg4_fused_extra(pkg, torch, dev, local_rank, args.gap)}
        emit(out)
    m.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
