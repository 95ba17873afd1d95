#!/usr/bin/env python3
"""bench.py — headline benchmark of the tvidz inspector hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher around it: this process touches no GPU and starts N fresh rank
  processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in
  their environment, exactly what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
  --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` gives them), relays rank 0's one
  JSON line and exits with the worst return code of the ranks.  With WORLD_SIZE set it is a rank.

Primary metric (BASELINE.json configs[1]): 1080p frames/s of scene-cut scoring.  A "step" is one
pass of the scene path (luma SAD kernel + finalize/select kernel, through the C ABI) over a
batch of T synthetic 1080p luma frames already resident in HBM.  One process per GPU; with N > 1
every rank scores its own T frames (independent videos shard with no collective: weak scaling)
and `value` = N*T*K / max-over-ranks time.

Secondary (BASELINE.json configs[2]/[3]), reported in the "match" object of the same JSON line:
timestamp-vector pair-compares/s of the corpus matcher, corpus sharded over the N ranks with one
RCCL all-gather of per-shard top-k per batch of Q queries (in the timed region).

"roofline" is for the dominant kernel (luma_sad_flat_kernel): algorithmic bytes = W*H per frame
scored / its launch duration measured with HIP events on the launch stream.
"cpu_baseline" times the CPU oracle (oracle/, a port — the reference's arithmetic is inside an
external ffmpeg binary that is not available) on a bounded sample of the same frames.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tvidz_amd import _lib, corpus as tc, scene, sharded, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
H, W = 1080, 1920
FRAME_BYTES = H * W


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=10000, help="1080p frames per GPU per step")
    ap.add_argument("--corpus", type=int, default=100000, help="corpus videos, sharded over the ranks (configs[3])")
    ap.add_argument("--queries", type=int, default=4096,
                    help="query videos per match batch (sized so that a 1/8 shard is still well above the fixed "
                         "per-batch cost: DESIGN.md 5, profiles/r2_predicted_scaling.json)")
    ap.add_argument("--match-steps", type=int, default=100, help="timed batches of the corpus match (20 warm-up batches before)")
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the bounded driver run (16 x 256 1080p uploads)")
    ap.add_argument("--match-depth", type=int, default=0,
                    help="query batches in flight in the corpus-match leg, one stream each (0: calibrate 2 against 3)")
    ap.add_argument("--e2e-timeout", type=float, default=240.0, help="abandon the e2e leg after this many seconds")
    ap.add_argument("--e2e-ranked", action="store_true",
                    help="run the e2e leg over service.RankCorpus (what --gpus N > 1 does on every rank) at any N: "
                         "the one way to exercise that code path on a one-GPU box")
    ap.add_argument("--cpu-frames", type=int, default=1536)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--e2e-shapes", default="8x256,64x256",
                    help="uploads x frames of the bounded driver runs; the first is the `e2e` object, the others go "
                         "to e2e.other_shapes (configs[4] is 64 concurrent uploads)")
    ap.add_argument("--force-launch", action="store_true",
                    help="start the rank process(es) through the launcher at --gpus 1 too (the path --gpus N > 1 takes)")
    ap.add_argument("--dry-launch", action="store_true", help="print the launcher's child command as JSON and exit")
    ap.add_argument("--launch-stub", action="store_true",
                    help="CPU self-test of the launcher: every rank joins a gloo group and all-reduces one number, rank 0 "
                         "prints a stub line; a rank named by TVZ_BENCH_STUB_FAIL_RANK exits 7 (tests/test_bench_contract_cpu.py)")
    ap.add_argument("--stub-hang-e2e", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def barrier_sync(world):
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x: float, world: int, dev) -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: int, world: int, dev) -> int:
    if not dist.is_initialized():
        return int(x)
    t = torch.tensor([int(x)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def bench_scene(args, rank, world, dev):
    T = args.frames
    frames = torch.empty((T, H, W), dtype=torch.uint8, device=dev)
    synth.synth_luma(T, H, W, device=dev, seed=synth.FRAME_SEED + rank, out=frames)
    scorer = scene.SceneScorer(H, W, T, dev, threshold=0.3)
    stream = torch.cuda.current_stream(dev)
    for _ in range(args.warmup):
        scorer.score_batch(frames, carry=False)
    barrier_sync(world)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        scorer.score_batch(frames, carry=False)   # enqueues sad + finalize kernels on `stream`
        ev[k][1].record(stream)
    barrier_sync(world)
    wall = time.perf_counter() - t0
    wall = max_over_ranks(wall, world, dev)
    step_ms = [a.elapsed_time(b) for a, b in ev]
    n_cuts = int(scorer.selected[:T].sum().item())
    return dict(frames=frames, scorer=scorer, wall=wall, step_ms=step_ms, n_cuts=n_cuts)


def cpu_baseline_scene(frames: torch.Tensor, n_frames: int, n_threads: int, min_seconds: float = 1.5):
    """Time the CPU oracle (port of ffmpeg's luma SAD + select) on the first n_frames."""
    from oracle import oracle  # checker / CPU baseline only
    n_frames = min(n_frames, frames.shape[0])
    host = frames[:n_frames].cpu().numpy()
    out = np.zeros(n_frames, dtype=np.uint64)
    n_threads = n_threads or min(os.cpu_count() or 1, 16)
    bounds = np.linspace(0, n_frames, n_threads + 1).astype(int)
    oracle.lib()
    passes, dt = 0, 0.0
    with ThreadPoolExecutor(n_threads) as ex:   # ctypes releases the GIL
        t0 = time.perf_counter()
        while dt < min_seconds:                 # bounded: ~min_seconds x n_threads of CPU work
            list(ex.map(lambda i: oracle.luma_sad_range(host, int(bounds[i]), int(bounds[i + 1]), out),
                        range(n_threads)))
            oracle.scene_select(out, H, W, 0.3)
            passes += 1
            dt = time.perf_counter() - t0
    return {"value": passes * n_frames / dt, "unit": "frames/s", "cores": n_threads, "kind": "port",
            "sample": f"oracle/tvz_oracle.c (gcc -O3) luma SAD + select, {passes} passes over the first {n_frames} "
                      f"of the same 1080p frames, {n_threads} threads, {dt:.2f} s wall = {dt * n_threads:.0f} "
                      "core-seconds (the reference's ffmpeg binary is not available: a port, not the reference)"}, out


def cpu_baseline_match(ids, offs, keys, queries, n_threads: int = 0):
    """(i) the pure-Python restatement with the reference's loop shape
    (db.py:85-91), 1 thread (GIL-bound) on a bounded sample; (ii) the fair CPU bound: C restatement
    on sorted-unique rows with binary search, all host threads."""
    from oracle import oracle  # checker / CPU baseline only
    n_threads = n_threads or min(os.cpu_count() or 1, 16)
    C = len(ids)
    rows = [(int(ids[c]), keys[offs[c]:offs[c + 1]].tolist()) for c in range(min(C, 5000))]
    t0 = time.perf_counter()
    n_py = 0
    while time.perf_counter() - t0 < 2.0:                  # ~2 s of single-thread CPython
        oracle.find_duplicates_py(rows, queries[1 + n_py % max(1, len(queries) - 1)].tolist(), 2)
        n_py += 1
    dt_py = time.perf_counter() - t0
    srt = keys.copy()
    for c in range(C):
        srt[offs[c]:offs[c + 1]].sort()
    nq = min(len(queries), 64)
    bounds = np.linspace(0, C, n_threads + 1).astype(int)
    cnt = np.zeros(C, dtype=np.int32); kth = np.zeros(C, dtype=np.int32)
    L = oracle.lib()
    qarr = [np.ascontiguousarray(x, dtype=np.float64) for x in queries[:nq]]

    def work(i):
        for qa in qarr:
            L.orc_match_kth_sorted(qa.ctypes.data, len(qa), offs.ctypes.data, srt.ctypes.data,
                                   int(bounds[i]), int(bounds[i + 1]), 2, cnt.ctypes.data, kth.ctypes.data)
    t0 = time.perf_counter()
    passes = 0
    with ThreadPoolExecutor(n_threads) as ex:
        while time.perf_counter() - t0 < 1.0:
            list(ex.map(work, range(n_threads)))
            passes += 1
    dt_c = time.perf_counter() - t0
    return {"python_restatement": {"value": n_py * len(rows) / dt_py, "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle.find_duplicates_py (loop shape of db.py:85-91), {n_py} queries vs "
                                             f"{len(rows)} rows, {dt_py:.2f} s"},
            "c_sorted_binary_search": {"value": passes * nq * C / dt_c, "unit": "pairs/s", "cores": n_threads,
                                       "kind": "port",
                                       "sample": f"oracle orc_match_kth_sorted, {passes} x {nq} queries vs {C} rows, "
                                                 f"{n_threads} threads, {dt_c:.2f} s = {dt_c * n_threads:.0f} core-seconds"}}


def kernel_ms(fn, stream, reps=12, skip=2):
    """Median duration of fn() in ms, HIP events on the stream fn launches on."""
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); fn(); b.record(stream)
        stream.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[skip:]))


def bench_match(args, rank, world, dev):
    C = args.corpus
    Q = args.queries
    ids, offs, keys = synth.synth_timestamp_corpus(C, seed=synth.CORPUS_SEED)
    # NB distinct query batches ROTATE through every timed loop below (a loop that
    # replays one batch measures directory lines and postings that are already in the caches)
    NB = 8
    q_sets = [synth.synth_queries(ids, offs, keys, Q, seed=synth.CORPUS_SEED + 1 + b) for b in range(NB)]
    queries = q_sets[0]
    s_ids, s_offs, s_keys = sharded.shard_csr(ids, offs, keys, rank, world)
    dc = tc.DeviceCorpus(dev.index)
    dc.upload_csr(s_ids, s_offs, s_keys)
    batches = [tc.pack_queries(qs, dev) for qs in q_sets]
    max_len = max(b[2] for b in batches)
    d_q, d_off, _ = batches[0]
    K_TOP = 16          # per-shard top-k travelling in the all-gather: [Q,17,3] int32 per rank
    CAP = 16384         # per-shard hit-list capacity per query (synthetic corpora: ~2,100 hits per
    #                     query at min_match=2 over 100k videos; overflows are counted below)
    # the collective lives behind the C ABI (tvz_match_sharded: match -> top-k -> ncclAllGather ->
    # merge) at every N, a single process included (one-rank communicator); torch.distributed only
    # ships the 128-byte RCCL id
    comm = sharded.make_comm(dev.index)
    rccl_ranks, rccl_rank = comm.info()                # what the communicator inside the library says (tvz_comm_info)
    # Batches in flight, each on its own stream.  A third one can fill the gaps the tails of two lookups leave
    # (55 -> 46 us per batch on a 1/8 shard in a fresh process) - or lose 20 %: how the runtime maps the streams
    # onto hardware queues decides, and that is not in the program's hands (profiles/r3_shard_pipeline.txt).  So
    # the depth is CALIBRATED: both pipelines answer 40 batches, the faster one runs the timed loop
    # (--match-depth 2|3 pins it).
    from collections import deque

    def run_stream(m, depth, steps, pick):
        inflight, last = deque(), None
        for i in range(steps):
            b = pick(i)
            inflight.append(m.submit(b[0], b[1], max_len, 2, inputs_ready=True))
            if len(inflight) >= depth:
                last = m.finish(inflight.popleft(), host=True)
        while inflight:
            last = m.finish(inflight.popleft(), host=True)
        return last
    calib = {}
    first_batch_ms = None
    cands = {}
    for d in ((args.match_depth,) if args.match_depth else (2, 3)):
        m = sharded.RcclShardedMatcher(dc, comm, k=K_TOP, cap=CAP, n_streams=d)
        if first_batch_ms is None:
            # the very first FULL batch this handle ever answers: nothing of the index is in any cache.  (A
            # one-query batch goes first: the communicator's first collective, the workspaces and the streams
            # are one-time set-up - tens of milliseconds on some boxes - not the cost of a cold index.)
            one = tc.pack_queries(q_sets[0][:1], dev)
            m.match_topk(one[0], one[1], max_len, 2)
            torch.cuda.synchronize()
            tc0 = time.perf_counter()
            m.match_topk(batches[NB - 1][0], batches[NB - 1][1], max_len, 2)
            torch.cuda.synchronize()
            first_batch_ms = (time.perf_counter() - tc0) * 1e3
        for i in range(2 * d + 4):
            m.match_topk(batches[i % NB][0], batches[i % NB][1], max_len, 2)
        torch.cuda.synchronize()
        tcal = time.perf_counter()
        run_stream(m, d, 40, lambda i: batches[i % NB])
        torch.cuda.synchronize()
        calib[d] = (time.perf_counter() - tcal) * 1e3 / 40
        cands[d] = m
    DEPTH = min(calib, key=calib.get)
    if world > 1:                        # every rank must run the same pipeline: rank 0's choice
        t = torch.tensor([DEPTH], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0)
        DEPTH = int(t.item())
    sm = cands[DEPTH]
    for i in range(20):
        merged, totals = sm.match_topk(batches[i % NB][0], batches[i % NB][1], max_len, 2)
    barrier_sync(world)

    def stream_of_batches(pick):
        """A stream of query batches over DEPTH HIP streams: the all-gather + merge of a batch overlap the
        lookups of the next ones; every batch is fully merged inside the timed region."""
        return run_stream(sm, DEPTH, args.match_steps, pick)
    t0 = time.perf_counter()
    merged, totals = stream_of_batches(lambda i: batches[i % NB])
    barrier_sync(world)
    wall = max_over_ranks(time.perf_counter() - t0, world, dev)
    merged, totals = merged.clone(), totals.clone()      # (the matcher's output slots are reused by the next loop)
    # the same loop replaying ONE batch (what rounds 1-3 timed), for the record
    t0 = time.perf_counter()
    stream_of_batches(lambda i: batches[0])
    torch.cuda.synchronize()
    replay_ms = (time.perf_counter() - t0) * 1e3 / args.match_steps
    pairs = Q * C * args.match_steps
    mean_len = float(offs[-1]) / C
    bytes_per_pair = 8.0 * mean_len + 8.0
    n_dups = int((totals != 0).sum().item())
    n_over = int((totals < 0).sum().item())            # negative total = a shard's list overflowed
    mean_hits = float(totals.abs().float().mean().item())
    # the dominant kernel of a batch, timed on its own with HIP events on its stream (rank-local
    # shard; every launch of one tvz_match call): once as the service runs it (AUTO: inverted-index
    # lookup, the delta table is empty here) and once forced onto the corpus sweep (hash join)
    st = torch.cuda.Stream(dev)
    ws = torch.empty(tc.workspace_bytes(Q, max_len, CAP, K_TOP), dtype=torch.uint8, device=dev)
    hits = torch.empty((Q, CAP, 3), dtype=torch.int32, device=dev)
    n_h = torch.empty(Q, dtype=torch.int32, device=dev)
    blk = torch.empty((Q, K_TOP + 1, 3), dtype=torch.int32, device=dev)
    rot = {"i": 0}

    def fused_call(handle):
        def f():
            b = batches[rot["i"] % NB]
            rot["i"] += 1
            handle.match_topk(b[0], b[1], max_len, 2, CAP, K_TOP, out=blk, workspace=ws, stream=st)
        return f
    # the kernel a batch spends its time in: the lookup with the top-k in its epilogue (one launch per call)
    index_ms = kernel_ms(fused_call(dc), st, reps=28, skip=4)
    # the unfused lookup (tvz_match: full hit lists + counter gather), for comparison and for the hit count
    unfused_ms = kernel_ms(lambda: dc.match(d_q, d_off, max_len, 2, CAP, out_hits=hits, out_n=n_h, stream=st,
                                            workspace=ws), st)
    n_hits = int(n_h.clamp(min=0).sum().item())
    sweep_ms = kernel_ms(lambda: dc.match(d_q, d_off, max_len, 2, CAP, out_hits=hits, out_n=n_h, stream=st,
                                          workspace=ws, algo=_lib.ALGO_JOIN), st)
    shard_rows, shard_keys, _ = dc.stats()
    ix = dc.index_stats()
    corpus_bytes = 16.0 * shard_rows + 8.0 * shard_keys
    # postings the batch walks on this shard: for every query element, the rows that hold its key
    uk, uc = np.unique(s_keys.view(np.int64), return_counts=True)
    qks = [np.concatenate([np.asarray(q, dtype=np.float64) for q in qs]).view(np.int64) for qs in q_sets]
    qk = qks[0]

    def postings_of(uk_, uc_):                        # mean over the rotating batches
        tot = 0
        for qk_ in qks:
            pos = np.searchsorted(uk_, qk_)
            pos[pos >= len(uk_)] = 0
            tot += int(uc_[pos][uk_[pos] == qk_].sum())
        return tot / len(qks)
    postings = postings_of(uk, uc)
    n_elems = float(np.mean([len(x) for x in qks]))
    n_sub = -(-shard_rows // 16384)
    # tvz_match.hip join_shape(): tiles of <= 1024 queries whose elements fit a 2 MiB table at load 0.55
    q_per_tile = max(1, min(1024, int(0.55 * (1 << 19)) // max(max_len, 1), Q))
    n_tiles = -(-Q // q_per_tile)
    comm.close()
    out = {"value": pairs / wall, "unit": "pairs/s", "rccl_ranks": rccl_ranks, "n_gpus": world,
           "corpus_videos": C, "queries_per_batch": Q,
           "mean_cuts_per_video": round(mean_len, 1), "min_match": 2, "steps": args.match_steps,
           "ms_per_batch": wall * 1e3 / args.match_steps,
           "ms_per_batch_one_batch_replayed": replay_ms, "first_batch_ms_cold": first_batch_ms,
           "batches_in_flight": DEPTH,
           "batches_in_flight_calibration_ms": {str(d): round(v, 4) for d, v in calib.items()},
           "distinct_query_batches_rotating": NB,
           "algo": "AUTO = inverted-index lookup, one block per query (two queries per block, probed together, on shards of "
                   "two sub-indexes) walking the query's sub-indexes of 16384 rows and keeping the per-shard top-k in its "
                   "epilogue (no hit lists, no top-k launch) + sweep of the delta table (empty here); on a shard of ONE "
                   "sub-index (N = 8) the bucket directory - a key's entry and its postings in one 128-byte line - and, for "
                   "a stream of batches, one WAVE per query (TVZ_ALGO_PREFER_WAVE); identical rows to the full sweep + top-k "
                   "(tests/test_index_topk_gpu.py)",
           "index": ix,
           "collective": (f"tvz_match_sharded (C ABI): one ncclAllGather of [Q,{K_TOP + 1},3] int32 per batch "
                          f"(top-{K_TOP} + hit totals) over {world} rank(s), overlapped with the next batch's match"),
           "queries_with_hits": n_dups, "mean_hits_per_query": round(mean_hits, 1),
           "hit_list_capacity": CAP, "queries_with_overflowed_shard_lists": n_over,
           "scaling": "strong (the same corpus is sharded over the ranks)",
           "match_ms_per_batch_rank0": index_ms, "unfused_lookup_ms_per_batch_rank0": unfused_ms,
           "sweep_ms_per_batch_rank0": sweep_ms,
           "predicted_scaling": "profiles/r5_predicted_scaling.json (single-GPU shard timings; no multi-GPU box was available)"}
    tag = f"C{C}_Q{Q}"
    # ---- roofline of the kernel the batch spends its time in: the index lookup ----
    # algorithmic bytes: every posting of the query's keys once (2 B), ONE directory entry
    # (16 B head + 2 B per sub-index, rounded up to 8 sub-indexes; 16 B on a one-sub-index shard) and the
    # 8-byte query key per query element, 12 B per hit
    def ix_alg_bytes(n_sub_, postings_):
        entry = 16 + (0 if n_sub_ <= 1 else 2 * ((n_sub_ + 7) // 8 * 8))
        return postings_ * 2.0 + n_elems * (entry + 8.0) + Q * (K_TOP + 1) * 12.0, entry
    alg_ix, entry_bytes = ix_alg_bytes(n_sub, postings)
    t_ix, t_src = pmc_traffic("ts_match_index_topk", tag="topk") if world == 1 else (None, None)
    out["roofline"] = {
        "bound": "hbm",
        "kernel": "ts_match_index_topk_kernel (one launch per batch; 8 query batches rotating)",
        "achieved": alg_ix / (index_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": alg_ix / (index_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": t_ix, "traffic_source": t_src,
        "algorithmic_bytes_per_launch": alg_ix, "avg_launch_ms": index_ms,
        "workload": f"rank {rank}'s shard of {C} videos ({shard_rows} rows, {n_sub} sub-index(es)) x {Q} queries",
        "pairs_per_s": Q * shard_rows / (index_ms * 1e-3),
        **(bound_fields("ts_match_index_topk", "topk") if world == 1 else {}),
        "algorithmic_bytes": f"{postings:.0f} postings x 2 B + one {entry_bytes} B directory entry and one 8 B query key per "
                             f"query element ({n_elems:.0f}) + {Q} x {K_TOP + 1} output rows x 12 B (means over the rotating "
                             f"batches; the {n_hits} hits of a batch are no longer written)",
        "limiter": "not HBM bandwidth: see frac_bound (the nearest of VALU issue, LDS arrays and the fabric's "
                   "random-line rate) and wave_cycles_waiting_share - a block is latency-bound (one block per query walks "
                   "its sub-indexes: per sub-index ~300 postings per wave through two LDS passes and five block barriers)"}
    # ---- the same batch against rank 0's share of an 8-way sharded corpus: what every GPU of configs[3] runs ----
    if world == 1:
        s8 = sharded.shard_csr(ids, offs, keys, 0, 8)
        dc8 = tc.DeviceCorpus(dev.index)
        dc8.upload_csr(*s8)
        # one sub-index: the bucket directory (a key's entry and its postings in ONE line) and two shapes of the lookup
        # that keeps the top-k - a block per query (the default: a lone batch is answered soonest) and a wave per query
        # (what a stream of batches in flight takes, sharded.RcclShardedMatcher: fewer instructions per batch)
        def shaped_call(handle, shape):
            def f():
                b = batches[rot["i"] % NB]
                rot["i"] += 1
                handle.match_topk(b[0], b[1], max_len, 2, CAP, K_TOP, out=blk, workspace=ws, stream=st, algo=shape)
            return f
        ms8 = kernel_ms(shaped_call(dc8, 0), st, reps=28, skip=4)
        ms8_wave = kernel_ms(shaped_call(dc8, _lib.ALGO_WAVE), st, reps=28, skip=4) if max_len <= 512 else None
        uk8, uc8 = np.unique(s8[2].view(np.int64), return_counts=True)
        post8 = postings_of(uk8, uc8)
        rows8 = dc8.stats()[0]
        alg8, entry8 = ix_alg_bytes(-(-rows8 // 16384), post8)
        t8, t8_src = pmc_traffic("ts_match_index_topk", tag="shard8")        # (keyed by workload: the r3 line looked up a row count)
        # the same shard as a STREAM of batches (three in flight on their own streams, one-rank communicator: the whole
        # tvz_match_sharded call per batch), per shape: what a GPU of configs[3] sustains
        pipe = {}
        comm8 = sharded.make_comm(dev.index)
        for name, fl in (("wave_per_query", _lib.ALGO_PREFER_WAVE), ("block_per_query", _lib.ALGO_NO_WAVE)):
            try:
                m8 = sharded.RcclShardedMatcher(dc8, comm8, k=K_TOP, cap=CAP, n_streams=3, algo=fl)
                for i in range(10):
                    m8.match_topk(batches[i % NB][0], batches[i % NB][1], max_len, 2)
                torch.cuda.synchronize()
                tp = time.perf_counter()
                run_stream(m8, 3, 120, lambda i: batches[i % NB])
                torch.cuda.synchronize()
                pipe[name] = round((time.perf_counter() - tp) * 1e3 / 120, 4)
            except Exception as e:            # noqa: BLE001 - a secondary figure
                pipe[name] = repr(e)
        comm8.close()
        out["shard8_roofline"] = {
            "bound": "hbm", "kernel": f"ts_match_index_topk_kernel on rank 0's 1/8 shard ({rows8} rows, one sub-index: bucket "
                                      f"directory), the same rotating batches of {Q} queries",
            "achieved": alg8 / (ms8 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg8 / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": t8, "traffic_source": t8_src,
            "algorithmic_bytes_per_launch": alg8, "avg_launch_ms": ms8,
            "workload": f"rank 0's 1/8 shard of {C} videos ({rows8} rows) x {Q} queries",
            "pairs_per_s": Q * rows8 / (ms8 * 1e-3),
            **bound_fields("ts_match_index_topk", "shard8"),
            "wave_per_query_kernel": {"name": "ts_match_wq_topk_kernel (TVZ_ALGO_WAVE)", "avg_launch_ms": ms8_wave,
                                      "traffic": pmc_traffic("ts_match_wq_topk", tag="shard8_wave")[0],
                                      **bound_fields("ts_match_wq_topk", "shard8_wave")},
            "ms_per_batch_as_a_stream_of_batches": dict(pipe, note="tvz_match_sharded, three batches in flight, Python host "
                                                                    "(~30 us of host time per batch: close to its floor)"),
            "limiter": "latency, not a throughput ceiling: see frac_bound (the nearest resource is the fabric's random-line "
                       "rate, one 128-byte bucket line per query timestamp) and wave_cycles_waiting_share"}
        dc8.close()
    # ---- the same batch forced onto the corpus sweep (what AUTO runs without an index) ----
    alg = corpus_bytes * n_tiles + n_hits * 12.0
    out["sweep"] = {
        "pairs_per_s_rank0": Q * shard_rows / (sweep_ms * 1e-3),
        "roofline": {
            "bound": "hbm",
            "kernel": f"ts_match_join_kernel: {n_tiles} corpus sweep(s), one per tile of <= {q_per_tile} queries (the "
                      "event pair also covers ts_prep + ts_join_build + the counter gather, < 10 % of it)",
            "achieved": alg / (sweep_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # the committed counter pass averages per LAUNCH; a batch is one launch per tile
            "traffic": (lambda t: t[0] * n_tiles if t[0] else None)(pmc_traffic("ts_match_join", tag=tag)) if world == 1 else None,
            "traffic_source": pmc_traffic("ts_match_join", tag=tag)[1] if world == 1 else None,
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": sweep_ms,
            "limiter": "not HBM: the sweep streams this rank's corpus image once per tile (16 B row entry + 8 B per "
                       "key) and is bound by the probe rate of the 2 MiB fingerprint table (one random 16 B bucket "
                       "per corpus key: L2 hit 78 %, 66 % of wave-cycles waiting, profiles/r2_match_pmc.txt)"}}
    out["nominal_hbm_equiv"] = {"GBps_per_gpu": pairs * bytes_per_pair / wall / 1e9 / world,
                                "x_hbm_peak": pairs * bytes_per_pair / wall / 1e9 / world / HBM_PEAK_GBS,
                                "note": "SURVEY 8d nominal 8*L+8 B per (query,row) pair; not a roofline: the index "
                                        "reads only the posting lists of the query's keys, the sweep serves a tile of "
                                        "up to 1024 queries per probe"}
    dc.close()
    # configs[2]: ONE query vs a 5k-video corpus on one GPU (+ the batch-size sweep SURVEY 8d asks for)
    if rank == 0 and world == 1:
        out["config2"] = bench_match_q1(args, dev, Q)
    return out


def bench_match_q1(args, dev, Q):
    res = {}
    for C5 in (5000, 100000):
        ids5, offs5, keys5 = synth.synth_timestamp_corpus(C5, seed=synth.CORPUS_SEED)
        q5 = synth.synth_queries(ids5, offs5, keys5, max(Q, 64), seed=synth.CORPUS_SEED + 1)
        dc5 = tc.DeviceCorpus(dev.index)
        dc5.upload_csr(ids5, offs5, keys5)
        rows5, nkeys5, _ = dc5.stats()
        image = 16.0 * rows5 + 8.0 * nkeys5
        st = torch.cuda.Stream(dev)
        by_q = {}
        for q_n in (1, 8, 64, 1024):
            if q_n > len(q5):
                continue
            dq, do, ml = tc.pack_queries(q5[:q_n], dev)
            hits = torch.empty((q_n, 4096, 3), dtype=torch.int32, device=dev)
            n_h = torch.empty(q_n, dtype=torch.int32, device=dev)
            ws = torch.empty(tc.workspace_bytes(q_n, ml), dtype=torch.uint8, device=dev)
            med = kernel_ms(lambda: dc5.match(dq, do, ml, 2, 4096, out_hits=hits, out_n=n_h, stream=st, workspace=ws),
                            st, reps=24, skip=4)
            by_q[str(q_n)] = {"kernel_ms": round(med, 4), "pairs_per_s": q_n * C5 / (med * 1e-3)}
        lat = []
        for i in range(60):
            t = time.perf_counter()
            dc5.find_duplicates(q5[i % len(q5)], 2)
            lat.append(time.perf_counter() - t)
        # the same single query forced onto the corpus sweep (what a corpus without an index runs), at
        # min_match 2 and at the reference's default 5 (a handful of hits instead of the thousands of
        # accidental min_match=2 collisions, each an append to one hit list)
        dq, do, ml = tc.pack_queries(q5[:1], dev)
        hits = torch.empty((1, 4096, 3), dtype=torch.int32, device=dev)
        n_h = torch.empty(1, dtype=torch.int32, device=dev)
        ws = torch.empty(tc.workspace_bytes(1, ml), dtype=torch.uint8, device=dev)
        q1_ms = kernel_ms(lambda: dc5.match(dq, do, ml, 2, 4096, out_hits=hits, out_n=n_h, stream=st, workspace=ws,
                                            algo=_lib.ALGO_Q1), st, reps=24, skip=4)
        q1_mm5 = kernel_ms(lambda: dc5.match(dq, do, ml, 5, 4096, out_hits=hits, out_n=n_h, stream=st, workspace=ws,
                                             algo=_lib.ALGO_Q1), st, reps=24, skip=4)
        entry = {"kernel_by_batch_size": by_q, "index": dc5.index_stats(),
                 "q1_sweep_kernel_ms": round(q1_ms, 4), "q1_sweep_kernel_ms_min_match5": round(q1_mm5, 4),
                 "find_duplicates_latency_us": round(float(np.median(lat[10:])) * 1e6, 1),
                 "roofline_q1": {"bound": "hbm", "kernel": "ts_match_q1_kernel, forced (one query, whole corpus image "
                                                           "streamed once; the event pair also covers the counter-zeroing "
                                                           "launch in front of it - the kernel alone, rocprofv3 average: "
                                                           "profiles/r3_match_q1_100k_kernel_stats.csv); AUTO answers "
                                                           "from the index instead, see kernel_by_batch_size",
                                 "algorithmic_bytes_per_launch": image,
                                 "achieved": image / (q1_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": image / (q1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "frac_min_match5": image / (q1_mm5 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": pmc_traffic("ts_match_q1", tag=f"C{C5}_Q1")[0],
                                 "traffic_source": pmc_traffic("ts_match_q1", tag=f"C{C5}_Q1")[1],
                                 "limiter": ("LATENCY, not bandwidth: an 8 MB image is ~1 us of HBM time; the event pair holds two "
                                             "launches (counter zeroing + sweep, the sweep alone ~11 us by rocprofv3) and their "
                                             "gaps - and the service never runs this: AUTO answers one query from the index "
                                             "(kernel_by_batch_size[\"1\"], find_duplicates_latency_us)") if C5 <= 5000 else
                                            ("HBM stream + the exact pass for rows with two Bloom positives + ~1,800 same-address "
                                             "returning atomics (profiles/r3_probe_experiment.txt); forced: AUTO takes the index")}}
        if C5 == 5000 and not args.no_cpu:
            entry["cpu_baseline"] = cpu_baseline_match(ids5, offs5, keys5, q5)
        res[f"c{C5}"] = entry
        dc5.close()
    return res


def bench_config0(dev, n_threads: int = 0):
    """BASELINE.json configs[0] / BASELINE.md section 3: the reference's own CPU-runnable case - two
    10 s 480p30 clips (854x480, 300 frames each; clip B = the same scenes with every cut shifted by 7
    frames), scene cuts + timestamp compare.  The reference's ffmpeg is not available, so the CPU
    figures are the oracle port (1 core and all cores) and its two matcher restatements; beside them
    the same two clips through the HIP path."""
    from oracle import oracle  # checker / CPU baseline only
    n_threads = n_threads or min(os.cpu_count() or 1, 16)
    T0, H0, W0 = 300, 480, 854
    a, _ = synth.synth_luma(T0 + 7, H0, W0, device=dev, seed=77, min_scene=40, max_scene=70, adversarial=False)
    clips = [a[7:].contiguous(), a[:T0].contiguous()]      # B: the same content 7 frames later
    host = [c.cpu().numpy() for c in clips]
    res = {"workload": "2 clips x 300 frames of 854x480 luma (10 s at 30 fps), clip B cut-shifted by 7 frames"}
    cuts = []
    for cores in (1, n_threads):
        bounds = np.linspace(0, T0, cores + 1).astype(int)
        out = np.zeros(T0, dtype=np.uint64)
        t0, passes = time.perf_counter(), 0
        with ThreadPoolExecutor(cores) as ex:
            while time.perf_counter() - t0 < 1.0:
                cuts = []
                for h in host:
                    list(ex.map(lambda i: oracle.luma_sad_range(h, int(bounds[i]), int(bounds[i + 1]), out), range(cores)))
                    sel, _, _, _ = oracle.scene_select(out, H0, W0, 0.3)
                    cuts.append([oracle.pts_time_value(int(i), 1, 30, 0) for i in np.flatnonzero(sel)])
                passes += 1
        dt = time.perf_counter() - t0
        res[f"cpu_scene_{cores}_core" + ("s" if cores > 1 else "")] = {
            "value": passes * 2 * T0 / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle luma SAD + select + pts_time, {passes} passes over the two clips, {dt:.2f} s"}
    # compare: clip B's fingerprint against a table holding clip A (db.py:85-91), min_match 2 (app.py:235)
    table = [(1, cuts[0])]
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.3:
        v_py = oracle.find_duplicates_py(table, cuts[1], 2)
        n += 1
    py_us = (time.perf_counter() - t0) / n * 1e6
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.3:
        v_c = oracle.find_duplicates_c(table, cuts[1], 2)
        n += 1
    c_us = (time.perf_counter() - t0) / n * 1e6
    res["cpu_compare"] = {"python_restatement_us": round(py_us, 2), "c_restatement_us": round(c_us, 2), "cores": 1,
                          "kind": "port", "verdict": v_py, "note": "a uniformly cut-shifted copy shares no exact timestamp "
                          "with the original: NOT a duplicate for the reference (db.py:79 exact match), nor here"}
    assert v_py == v_c
    # the same through the HIP path (frames resident in HBM)
    dc = tc.DeviceCorpus(dev.index)
    g_cuts = []
    torch.cuda.synchronize()
    t0, passes = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.5:
        g_cuts = [[ts for _, ts in scene.detect_scene_cuts(c, time_base=(1, 30), batch=T0)] for c in clips]
        passes += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dc.upsert(1, g_cuts[0])
    lat = []
    for _ in range(200):
        t1 = time.perf_counter()
        v_g = dc.find_duplicates(g_cuts[1], 2)
        lat.append(time.perf_counter() - t1)
    dc.close()
    res["gpu"] = {"scene_frames_per_s": passes * 2 * T0 / dt, "compare_us": round(float(np.median(lat)) * 1e6, 2),
                  "verdict": v_g, "cuts_equal_cpu": g_cuts == cuts,
                  "note": "detect_scene_cuts on two 300-frame clips, one micro-batch each (launch + one D2H per clip: "
                          "a 123 MB clip is latency-bound, not HBM-bound) + tvz_find_duplicates through Python"}
    return res


def bench_e2e(dev, rank: int = 0, world: int = 1, n_uploads: int = 8, n_frames: int = 256, ranked: bool = False):
    """A bounded run of the whole driver (BASELINE.json configs[4]'s shape at 1080p): N concurrent
    uploads as mono Y4M files in RAM -> reader threads -> pinned slot pool -> H2D -> scene kernels ->
    one match per micro-batch -> write-behind SQL.  PCIe-inclusive; never part of `value`.  The 16 /
    64-upload and 4K figures are in profiles/r4_e2e_service.txt (profiles/e2e_service.py).
    world > 1 (`--gpus N` under torch.distributed.run): every rank runs this leg over
    service.RankCorpus(RcclShardedMatcher) - its shard of the table, the collective tick exchange with
    the other ranks - on its own uploads; the figure is all ranks' frames over the slowest rank's time.
    UNMEASURED ON HARDWARE for N > 1: no box of this project has had two GPUs."""
    import shutil
    import tempfile
    from tvidz_amd import db as tdb, feeder, inspector as insp
    need = n_uploads * n_frames * FRAME_BYTES * 1.2
    root = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need else None
    tmp = tempfile.mkdtemp(prefix="tvz_bench_e2e_", dir=root)
    try:
        files = {}
        for i in range(n_uploads):
            fr, _ = synth.synth_luma(n_frames, H, W, device=dev, seed=300 + i, min_scene=20, max_scene=90)
            name = f"17000000{i:02d}-clip{i}.y4m"
            files[name] = os.path.join(tmp, name)
            feeder.write_y4m(files[name], fr.cpu().numpy(), fps=(30, 1), chroma="mono")
            del fr
        ids, offs, keys = synth.synth_timestamp_corpus(5000, seed=1)
        lib_rows = [(int(ids[c]) + 100000, keys[offs[c]:offs[c + 1]].tolist()) for c in range(len(ids))]
        rc = None
        if world > 1 or ranked:
            from tvidz_amd import service
            shard = tc.DeviceCorpus(dev.index)
            # the tick thread's own group; the asks are host data (a single process: no group at all)
            group = dist.new_group(backend="gloo") if dist.is_initialized() else None
            matcher = sharded.RcclShardedMatcher(shard, sharded.make_comm(dev.index), k=64, cap=4096)
            rc = service.RankCorpus(shard, matcher, group=group, xdev="cpu")
            store = tdb.Store(f"sqlite:///{tmp}/tvidz.db", corpus=rc, census=False)   # this rank's partition of the table

            def load():
                store.clear()
                # (separate SQLite files per rank: keep the video ids of the ranks apart, as one shared table would)
                s_ = store.SessionLocal()
                try:
                    s_.add(tdb.Video(id=(rank + 1) * 1000000, filename=f"rank{rank}-floor"))
                    s_.commit()
                finally:
                    s_.close()
                rc.upload(lib_rows)                                      # video_id mod world == rank stay here
        else:
            store = tdb.Store(f"sqlite:///{tmp}/tvidz.db", device=dev.index)

            def load():
                store.clear()
                store.corpus.upload_csr(ids + 100000, offs, keys)
        load()
        ins = insp.Inspector(store, device=str(dev), frame_source=lambda b, k, f, u: (feeder.Y4MReader(files[k]), None),
                             batch=256, max_workers=min(n_uploads, 16))
        [f.result() for f in [ins.submit("videos", k) for k in files]]          # warm-up: slots, scorers, SQL
        ok, dts, scored, dups = True, [], [], 0
        for _ in range(3):               # three timed passes, the median reported: a 0.15 s run is at the host's mercy
            load()
            barrier_sync(world)
            before = ins.frames_scored
            t0 = time.perf_counter()
            res = [f.result() for f in [ins.submit("videos", k) for k in files]]
            dts.append(max_over_ranks(time.perf_counter() - t0, world, dev))
            # frames the scene kernels actually scored: an upload that reaches a duplicate verdict stops there
            # (inspector/app.py:249-255), so it is NOT uploads x frames
            scored.append(sum_over_ranks(ins.frames_scored - before, world, dev))
            ok = ok and all(r["status"] == "done" for r in res)
            dups = sum(1 for r in res if r["duplicates"])
        mid = int(np.argsort(dts)[len(dts) // 2])
        dt = dts[mid]
        ins.close()
        tick = None
        if rc is not None:
            tick = {"ticks": rc.ticks, "busy_ticks": rc.busy_ticks, "exact_asks": rc.exact_asks,
                    "host_us_per_busy_tick": round(rc.tick_host_s * 1e6 / max(rc.busy_ticks, 1), 1)}
            barrier_sync(world)
        store.close()                    # (RankCorpus.close is collective: every rank gets here)
        submitted = world * n_uploads * n_frames
        return {"value": scored[mid] / dt, "unit": "frames/s", "uploads": world * n_uploads, "frames_per_upload": n_frames,
                "frames_scored": scored[mid], "frames_submitted": submitted,
                "submitted_frames_per_s": submitted / dt,
                "uploads_stopped_by_a_duplicate_verdict_rank0": dups,
                "ranks": world, "rank_tick": tick, "unmeasured_on_hardware_for_ranks_above_1": world > 1,
                "height": H, "width": W, "all_done": ok, "GBps_luma": scored[mid] * FRAME_BYTES / dt / 1e9,
                "passes_s": [round(x, 4) for x in dts], "passes_fps": [round(n / x) for n, x in zip(scored, dts)],
                "note": "whole Python driver, PCIe-inclusive, files in RAM, NO DECODER (the reference's ffmpeg decode, "
                        "inspector/app.py:202-216, is not in this figure: no ffmpeg binary exists here); value = frames the "
                        "scene kernels scored / wall (an upload stops at its duplicate verdict, app.py:249-255), "
                        "submitted_frames_per_s = uploads x frames / wall is NOT a throughput; short clips: per-upload "
                        "set-up (SQL insert, reader thread, first slot) is inside the wall time; one H2D copy in flight"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def h2d_cost(dev, n_frames: int = 256):
    """Separate H2D cost line: pinned host luma -> HBM, never part of `value`."""
    host = torch.empty((n_frames, H, W), dtype=torch.uint8, pin_memory=True)
    dst = torch.empty((n_frames, H, W), dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); dst.copy_(host, non_blocking=True); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts[1:]))
    gbs = n_frames * FRAME_BYTES / ms / 1e6
    return {"GBps": round(gbs, 1), "ms_per_256_frames": round(ms, 3), "frames_per_s_bound": round(n_frames / ms * 1e3),
            "note": "pinned host -> HBM copy of 256 x 1080p luma; the PCIe-inclusive ceiling of the scene path"}


def pmc_traffic(kernel: str, T: int = 0, tag: str = ""):
    """(HBM bytes per launch, provenance) from the COMMITTED rocprofv3 PMC passes
    (profiles/*_pmc_summary.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE) - counters
    cannot be collected inside this run; the newest committed pass that has the kernel wins.  Scene
    kernel: valid for the default T only; matcher kernels: looked up under their workload tag
    (e.g. "C100000_Q4096_index")."""
    import glob
    best, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
            e = None
            if tag:
                e = d.get(tag, {}).get(kernel)
            elif d.get("_frames_per_launch", 10000) == T and kernel in d:
                e = d[kernel]
            if e:
                best = e.get("hbm_read_bytes_corrected", 0) + e.get("hbm_write_bytes", 0)
                src = f"committed counter pass profiles/{os.path.basename(f)} (not measured in this run)"
        except Exception:
            pass
    return best, src


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launcher_cmd(argv):
    """The command of ONE rank process of the self-launcher (rank, world and rendezvous travel in the environment,
    as under torch.distributed.run: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)."""
    keep = [a for a in argv if a not in ("--force-launch", "--dry-launch")]
    return [sys.executable, os.path.abspath(__file__)] + keep


def rank_env(rank: int, world: int, port: int) -> dict:
    env = dict(os.environ)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def launch_ranks(args, argv) -> int:
    """--gpus N without a launcher around it (WORLD_SIZE unset): THIS process has not touched a GPU and never
    will - it starts N FRESH rank processes (one per GPU; nothing is exec'ed over a process that has initialised
    the GPU), relays rank 0's one JSON line and returns the WORST return code of the ranks.  When a rank fails the
    others get 10 s to leave by themselves (a watchdog's return code 3 is worth more than a SIGTERM's), then are ended
    (SIGTERM, and SIGKILL 10 s later for a rank that does not react)."""
    cmd, port, n = launcher_cmd(argv), free_port(), args.gpus
    if args.dry_launch:
        print(json.dumps({"launch": cmd, "ranks": n,
                          "env": {k: rank_env(0, n, port)[k] for k in ("WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}}))
        return 0
    procs = [subprocess.Popen(cmd, env=rank_env(r, n, port), stdout=subprocess.PIPE if r == 0 else sys.stderr)
             for r in range(n)]
    line = [None]

    def relay():                                              # ranks write everything but rank 0's line to stderr
        for raw in procs[0].stdout:
            txt = raw.decode(errors="replace")
            try:
                d = json.loads(txt)
                ok = isinstance(d, dict) and ("metric" in d or d.get("stub"))
            except ValueError:
                ok = False
            if ok and line[0] is None:
                line[0] = txt
            else:
                sys.stderr.write(txt)
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    first_fail = term_at = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.05)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and first_fail is None:
            first_fail = time.monotonic()
        if first_fail is not None and term_at is None and time.monotonic() - first_fail > 10.0:
            for p in procs:
                if p.poll() is None:
                    p.terminate()                             # the exact processes this launcher started
            term_at = time.monotonic()
        if term_at is not None and time.monotonic() - term_at > 10.0:
            for p in procs:                                   # a rank stuck in the driver does not see SIGTERM
                if p.poll() is None:
                    p.kill()
            term_at = time.monotonic() + 1e9
    th.join(5.0)
    rcs = [p.returncode if p.returncode >= 0 else 128 - p.returncode for p in procs]
    rc = max(rcs)
    if rc:
        print(f"[bench] return codes of the {n} rank processes: {rcs}", file=sys.stderr)
    if line[0] is not None:
        sys.stdout.write(line[0] if line[0].endswith("\n") else line[0] + "\n")
        sys.stdout.flush()
    if rc == 0 and line[0] is None:
        print("[bench] the rank processes printed no result line", file=sys.stderr)
        rc = 4
    return rc


def run_under_watchdog(leg, timeout: float, rank: int, world: int, out: dict, key: str, emit, store=None):
    """Run leg() - a collective leg that has never run on more than one GPU - under a watchdog.  If it does not
    finish in `timeout` seconds (a collective some rank never joins), rank 0 prints the line it has, with
    out[key] = {"error": ..., "ranks_that_never_finished": [...]} (from markers the ranks leave in the
    rendezvous store, a host TCP channel that does not depend on the hung collective), and EVERY rank leaves with
    return code 3: a process that abandons a hung collective on the GPU must not look like a successful run."""
    done = threading.Event()

    def mark(name):
        if store is not None:
            try:
                store.set(f"tvz_bench_{key}_{name}_{rank}", "1")
            except Exception:
                pass

    def watchdog():
        if done.wait(timeout):
            return
        missing = None
        if store is not None:
            try:
                missing = [r for r in range(world) if not store.check([f"tvz_bench_{key}_finished_{r}"])]
            except Exception:
                missing = None
        out[key] = {"error": f"{key} leg did not finish within {timeout:.0f} s on rank {rank} (abandoned; exit code 3)",
                    "ranks_that_never_finished": missing}
        print(f"[bench] rank {rank}: {key} leg timed out (ranks that never finished: {missing}); leaving with rc 3",
              file=sys.stderr)
        if rank != 0:
            time.sleep(2.0)               # rank 0 prints first: the launcher ends the others at the first failure
        emit()
        os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        res = leg()
    except Exception as e:                # the driver needs SQLAlchemy + a writable temp directory: report, don't die
        res = {"error": repr(e)}
        print(f"[bench] {key} leg failed on rank {rank}: {e!r}", file=sys.stderr)
    mark("finished")
    if dist.is_initialized():
        dist.barrier()                    # (still under the watchdog)
    done.set()
    return res


def pmc_entry(kernel: str, tag: str):
    """The committed counter summary's whole entry for (workload tag, kernel) - profiles/summarize_match.py: HBM
    traffic, and the resource the kernel is closest to (`frac_bound`: VALU issue, the LDS arrays, or the rate the
    fabric delivers random 128-byte lines) - or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_match_pmc_summary.json"))):
        try:
            e = json.load(open(f)).get(tag, {}).get(kernel)
            if e:
                best = dict(e, source=f"profiles/{os.path.basename(f)}")
        except Exception:
            pass
    return best


def bound_fields(kernel: str, tag: str):
    e = pmc_entry(kernel, tag) or {}
    fb = e.get("frac_bound")
    return {"frac_bound": dict(fb, bounds=e.get("bounds"), source=e.get("source"),
                               note="share of the resource's capacity over the kernel's own duration, committed counter pass "
                                    "(profiles/summarize_match.py: VALU issue, LDS-array cycles, random 128-byte lines "
                                    "against 30.3 G lines/s); the largest of the three") if fb else None,
            "lds_bank_conflict_share": e.get("lds_bank_conflict_share"),
            "wave_cycles_waiting_share": e.get("wave_cycles_waiting_share")}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    # Launcher: before anything touches a GPU (importing torch does not; torch.cuda.* is not called above)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_launch or args.dry_launch):
        raise SystemExit(launch_ranks(args, argv))
    # ONE JSON line on stdout, nothing else: native libraries print there too (RCCL's version banner at
    # communicator creation), so everything written to descriptor 1 during the run goes to stderr and the
    # result line is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N` alone: it launches the ranks itself)")
    out = {}
    printed = threading.Event()

    def emit():
        if rank == 0 and not printed.is_set():
            printed.set()
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())

    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ   # under torch.distributed.run
    if args.launch_stub:
        # CPU self-test of the launcher and of the watchdog's return code: no GPU, no kernels, gloo
        if launched:
            dist.init_process_group("gloo")
        t = torch.tensor([rank + 1.0])
        if dist.is_initialized():
            dist.all_reduce(t)
        if os.environ.get("TVZ_BENCH_STUB_FAIL_RANK", "") == str(rank):
            os._exit(7)
        out.update({"stub": True, "n_gpus": world, "sum_of_rank_numbers": float(t.item())})
        if args.stub_hang_e2e:
            store = dist.distributed_c10d._get_default_store() if dist.is_initialized() else None
            hang = (lambda: time.sleep(3600)) if rank == world - 1 else (lambda: {"ok": True})
            out["e2e"] = run_under_watchdog(hang, args.e2e_timeout, rank, world, out, "e2e", emit, store)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        emit()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or launched:
        dist.init_process_group("nccl", device_id=dev)
    _lib.load()

    # The corpus-match leg runs FIRST: its two alternating streams overlap one batch's small kernels
    # with the next batch's match only if the runtime maps them onto hardware queues that dispatch
    # concurrently.  Behind the scene / e2e legs (which use the default stream and a dozen others
    # first) the two streams landed on queues 2 and 6 and the batches serialised: 0.45 ms per batch
    # instead of 0.34 (profiles/r3_shard_pipeline.txt).
    match_leg = None
    if not args.no_match:
        try:
            match_leg = bench_match(args, rank, world, dev)
        except Exception as e:            # the headline (scene) leg below must not die with the secondary one
            match_leg = {"error": repr(e)}
            print(f"[bench] corpus-match leg failed on rank {rank}: {e!r}", file=sys.stderr)
    torch.cuda.empty_cache()

    res = bench_scene(args, rank, world, dev)
    T, K = args.frames, args.steps
    wall = res["wall"]
    fps = world * T * K / wall
    # ONE time base for value, ms_per_step and the roofline: the timed region (barrier + synchronize on both
    # sides) divided by its K launches; the per-launch HIP-event durations (which leave out the gaps between
    # launches) are reported beside it
    step_ms = wall * 1e3 / K
    ev_ms = float(np.mean(res["step_ms"]))
    alg_scene = (T - 1) * FRAME_BYTES
    achieved = alg_scene / (step_ms * 1e-3) / 1e9
    scene_traffic, scene_src = pmc_traffic("luma_sad", T)

    out.update({
        "metric": "1080p frames/sec scene-cut scoring (luma SAD + select), frames resident in HBM",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"configs[1]: {T} synthetic 1080p luma frames per GPU per step, "
                               "HIP luma-SAD scene-cut kernel + select epilogue",
                   "frames_per_gpu": T, "height": H, "width": W, "threshold": 0.3,
                   "parallelism": f"{world} independent video batches (no collective)"},
        "cuts_detected_per_step": res["n_cuts"],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": scene_traffic,
                     "traffic_source": scene_src,
                     "kernel": "luma_sad_flat_kernel<8,nt> (a step also holds scene_finalize_kernel, <1% of it)",
                     "algorithmic_bytes_per_launch": alg_scene,
                     "avg_launch_ms": step_ms,
                     "time_base": "the timed region / its K launches = ms_per_step (the launches run back to back on one stream)",
                     "per_launch_event_ms_mean": ev_ms, "median_launch_ms": float(np.median(res["step_ms"])),
                     "p10_p90_ms": [float(np.percentile(res["step_ms"], 10)),
                                    float(np.percentile(res["step_ms"], 90))]},
    })
    # both halves of BASELINE.json's metric in the objects the driver keeps whole: the dominant
    # kernel of each leg with everything a reader needs to recompute its fraction
    kernels = [{"name": "luma_sad_flat_kernel", "workload": out["config"]["workload"], "avg_launch_ms": step_ms,
                "algorithmic_bytes": alg_scene, "traffic": scene_traffic, "frac": achieved / HBM_PEAK_GBS,
                "frames_per_s": fps / world}]
    if isinstance(match_leg, dict) and "roofline" in match_leg:
        for r in (match_leg["roofline"], match_leg.get("shard8_roofline")):
            if r:
                kernels.append({"name": r["kernel"].split(" ")[0], "workload": r.get("workload"),
                                "avg_launch_ms": r["avg_launch_ms"], "algorithmic_bytes": r["algorithmic_bytes_per_launch"],
                                "traffic": r["traffic"], "frac": r["frac"], "pairs_per_s": r.get("pairs_per_s"),
                                "frac_bound": r.get("frac_bound")})
        wk = (match_leg.get("shard8_roofline") or {}).get("wave_per_query_kernel")
        if wk and wk.get("avg_launch_ms"):
            r8 = match_leg["shard8_roofline"]
            kernels.append({"name": "ts_match_wq_topk_kernel", "workload": r8.get("workload") + " (one wave per query)",
                            "avg_launch_ms": wk["avg_launch_ms"], "algorithmic_bytes": r8["algorithmic_bytes_per_launch"],
                            "traffic": wk.get("traffic"),
                            "frac": r8["algorithmic_bytes_per_launch"] / (wk["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "pairs_per_s": r8["pairs_per_s"] * r8["avg_launch_ms"] / wk["avg_launch_ms"],
                            "frac_bound": wk.get("frac_bound")})
        kernels.append({"name": "tvz_match_sharded (whole batch: lookup with top-k -> ncclAllGather -> merge, pipelined)",
                        "workload": f"{match_leg['corpus_videos']} videos x {match_leg['queries_per_batch']} queries over "
                                    f"{match_leg['rccl_ranks']} RCCL rank(s)",
                        "avg_launch_ms": match_leg["ms_per_batch"], "pairs_per_s": match_leg["value"]})
    out["roofline"]["kernels"] = kernels
    if rank == 0:
        out["h2d"] = h2d_cost(dev)
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu, cpu_sad = cpu_baseline_scene(res["frames"], args.cpu_frames, args.cpu_threads)
        gpu_sad = res["scorer"].sad[:len(cpu_sad)].cpu().numpy().view(np.uint64)
        cpu["agrees_with_gpu"] = bool((gpu_sad == cpu_sad).all())
        # the restatement on ONE core as well as on all of them
        one, _ = cpu_baseline_scene(res["frames"], min(args.cpu_frames, 256), 1, min_seconds=1.0)
        cpu["single_core"] = {"value": one["value"], "unit": "frames/s", "cores": 1, "sample": one["sample"]}
        m5 = (match_leg or {}).get("config2", {}).get("c5000", {}) if isinstance(match_leg, dict) else {}
        if "cpu_baseline" in m5:
            cpu["matcher"] = m5["cpu_baseline"]            # the second half of the metric, timed on the same host cores
        out["cpu_baseline"] = cpu
    else:
        out["cpu_baseline"] = None
    del res
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu:
        out["config0"] = bench_config0(dev, args.cpu_threads)
    if match_leg is not None:
        out["match"] = match_leg

    if not args.no_e2e:                   # every rank: the N-rank leg is collective (the tick exchange)
        # The leg that has never run on more than one GPU comes LAST and under a watchdog (run_under_watchdog)
        shapes = [tuple(int(x) for x in sh.split("x")) for sh in args.e2e_shapes.split(",") if sh]
        store = dist.distributed_c10d._get_default_store() if dist.is_initialized() else None

        def leg():
            first = None
            for j, (nu, nf) in enumerate(shapes):
                r = bench_e2e(dev, rank, world, n_uploads=nu, n_frames=nf, ranked=args.e2e_ranked)
                torch.cuda.empty_cache()
                if first is None:
                    first = r
                else:
                    first.setdefault("other_shapes", []).append(r)
            return first
        e2e = run_under_watchdog(leg, args.e2e_timeout, rank, world, out, "e2e", emit, store)
        if rank == 0:
            out["e2e"] = e2e
            if isinstance(e2e, dict) and "GBps_luma" in e2e and "h2d" in out:
                # frames that were scored crossed the link: the figure cannot exceed what the link moves
                e2e["h2d_bound_check"] = {"copies_in_flight": 1, "h2d_GBps": out["h2d"]["GBps"],
                                          "ok": all(x["GBps_luma"] <= out["h2d"]["GBps"] * 1.02
                                                    for x in [e2e] + e2e.get("other_shapes", []))}
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    emit()


if __name__ == "__main__":
    main()
