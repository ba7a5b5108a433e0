#!/usr/bin/env python3
"""bench.py — MCTS simulations/s of device-resident AlphaZero-Risk self-play on MI355X.

Workload (BASELINE.json configs[1]): 256 concurrent self-play games per GPU, 100 MCTS simulations per move,
random-init 20-block / 256-filter net, bf16 MFMA contractions with fp32 accumulation.  One "step" = one pass of
the hot path over the batch of games: a tree step (expand + backup of the previous leaves, PUCT descent to the
next leaves — and, when a search completes, the move, the record and possibly a game restart — for all G games)
followed by one batched net evaluation of the G leaves.  Everything is resident in HBM; nothing crosses PCIe in
the timed region except the final counters.  Multi-GPU: one process per GPU, games sharded, no data-path
collective; one RCCL all_gather of the finished (s, pi, z) records closes the timed region (weak scaling).

    python bench.py [--gpus N --steps K --warmup W] [--games 256 --sims 100 --blocks 20 --dtype bf16]
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SIM = {20: 2 * 797_976_348, 5: 2 * 200_288_028}  # valid (un-padded) taps only
PEAK_BF16 = 2.5e15   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32 = 157.3e12  # fp32 vector/matrix


def flop_per_sim(blocks):
    if blocks in FLOP_PER_SIM:
        return FLOP_PER_SIM[blocks]
    mac = 304 * 13 * 256 + blocks * 2 * 304 * 256 * 256 + 46_876
    return 2 * mac


def cpu_baseline(blocks, sims, mcts_threads, seconds_budget=25.0):
    """the oracle (CPU port of the reference path, one game per OS thread, the game's THREADS_PER_MCTS search threads in
    the same lock-step schedule as the device, fp32 CPU net) on this box's host cores, bounded sample: one decision
    (sims simulations + root expansion) per game"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import azr_testlib as T

    orc = T.oracle()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 16))  # a 1-GPU box has a 16-core CPU share; "cores" reported = threads actually used
    flat = T.make_net_flat(blocks)
    net = T.OrcNet(blocks, flat.ctypes.data_as(T.f32p))
    cfg = T.default_settings(mcts_simulations=sims, mcts_threads=mcts_threads)
    orc.orc_bench_selfplay.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
    s, e, sec = C.c_uint64(), C.c_uint64(), C.c_double()
    # calibrate with a short search so the real sample stays inside the budget
    cal = T.default_settings(mcts_simulations=4)
    orc.orc_bench_selfplay(C.byref(cal), C.byref(net), 20260001, threads, 1, C.byref(s), C.byref(e), C.byref(sec))
    per_eval = sec.value / max(1, e.value / threads)
    decisions = max(1, int(seconds_budget / (per_eval * (sims + 1))))
    decisions = min(decisions, 4)
    orc.orc_bench_selfplay(C.byref(cfg), C.byref(net), 20260001, threads, decisions, C.byref(s), C.byref(e),
                           C.byref(sec))
    return {"value": s.value / sec.value, "unit": "MCTS simulations/s", "cores": threads, "kind": "port",
            "sample": f"{threads} games (one per host thread, THREADS_PER_MCTS={mcts_threads}), {decisions} decision(s) x {sims} sims each, "
                      f"{blocks}-block fp32 CPU net, {sec.value:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--games", type=int, default=256, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--threads", type=int, default=2,
                    help="THREADS_PER_MCTS (-t): search threads per game; 2 is the reference's default (src/settings.h:44)")
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-games", action="store_true",
                    help="skip the untimed tail that plays on until G games have finished (self-play games/s)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only path (no CPU fallback)")
    # backend "nccl" IS RCCL on ROCm.  AZR_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than
    # ranks (ranks then share GPUs and the record gather runs on host tensors).
    backend = os.environ.get("AZR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("alphazero-risk_amd")
    from importlib import import_module
    shard = import_module("alphazero-risk_amd.shard")
    eng = pkg.Engine(a.games, blocks=a.blocks, sims=a.sims, dtype=pkg.NET_BF16 if a.dtype == "bf16" else pkg.NET_F32,
                     device=local, threads=a.threads)
    eng.init_random(20260002)
    eng.selfplay_start(shard.rank_base_seed(20260001, rank))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_start = time.perf_counter()
    eng.selfplay_run(a.warmup)
    eng.drain(1)  # reset the record ring
    c0 = eng.counters()
    barrier()
    t0 = time.perf_counter()
    # EXACTLY K passes, issued in 10 slices so that a median rate can be reported next to the mean; a slice
    # boundary is one counter read-back (a stream sync of a few microseconds)
    chunk_rates, profs, done, prev_sims, prev_t = [], [], 0, c0["simulations"], t0
    for i in range(10):
        n = (a.steps * (i + 1)) // 10 - done
        if n <= 0:
            continue
        eng.selfplay_run(n)
        done += n
        now, cs = time.perf_counter(), eng.counters()["simulations"]
        chunk_rates.append((cs - prev_sims) / max(now - prev_t, 1e-9))
        profs.append(eng.profile_last_run())
        prev_sims, prev_t = cs, now
    ptr, nrec = eng.samples_device_view()
    recs = shard.device_records_to_torch(ptr, nrec, dev)
    allrecs = shard.gather_records(recs.to(cdev), dist if world > 1 else None)   # the path's one exchange step
    barrier()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    # HIP-event timings of the kernels (on the engine's stream), launch-weighted over the slices of the timed region
    nl = max(1, sum(p_["launches"] for p_ in profs))
    prof = {"net_ms": sum(p_["net_ms"] * p_["launches"] for p_ in profs) / nl,
            "tree_ms": sum(p_["tree_ms"] * p_["launches"] for p_ in profs) / nl, "launches": sum(p_["launches"] for p_ in profs)}

    delta = {k: c1[k] - c0[k] for k in c1}
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    tot = torch.tensor([delta["simulations"], delta["evaluations"], delta["levels"], delta["decisions"],
                        delta["games_finished"], delta["samples"], delta["errors"], delta["nodes_dropped"]],
                       dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    sims, evals, levels, decisions, games, samples, errors, dropped = [int(x) for x in tot.tolist()]

    # untimed tail (N = 1 only): keep playing until G games have finished to measure whole self-play games/s
    games_rate = None
    if world == 1 and not a.no_full_games:
        t_tail = time.perf_counter()
        while time.perf_counter() - t_tail < 120.0:
            eng.selfplay_run(2000)
            eng.drain(1)
            ct = eng.counters()
            if ct["games_finished"] >= a.games:
                break
        ct = eng.counters()
        el = time.perf_counter() - t_start
        if ct["games_finished"] > 0:
            games_rate = {"games_per_s": ct["games_finished"] / el, "games": ct["games_finished"],
                          "decisions_per_finished_game": ct["samples"] / ct["games_finished"],
                          "window_s": el, "simulations_per_s_over_window": ct["simulations"] / el}

    if rank == 0:
        fps = flop_per_sim(a.blocks)
        net_s = prof["net_ms"] * 1e-3
        # algorithmic work of one launch = the leaves that were actually waiting for the net (idle slots are not counted)
        leaves_per_launch = evals / max(1, a.steps * world)
        achieved = leaves_per_launch * fps / net_s if net_s > 0 else 0.0
        peak = PEAK_BF16 if a.dtype == "bf16" else PEAK_F32
        out = {
            "metric": "MCTS simulations/s", "value": sims / dt, "unit": "simulations/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{a.games} concurrent self-play games/GPU x {a.sims} MCTS sims/move, "
                                   f"THREADS_PER_MCTS {a.threads}, {a.blocks}-block 256-filter random-init net (BASELINE configs[1])",
                       "games_per_gpu": a.games, "sims_per_move": a.sims, "mcts_threads": a.threads, "blocks": a.blocks,
                       "parallelism": f"games sharded x{world}, no data-path collective; 1 all_gather of records"},
            "simulations_per_s_p50_rank0": sorted(chunk_rates)[len(chunk_rates) // 2] if chunk_rates else None,
            "self_play_games_per_s": games_rate["games_per_s"] if games_rate else games / dt,
            "self_play_games_window": games_rate, "decisions_per_s": decisions / dt,
            "net_evals_per_s": evals / dt, "mean_depth": levels / max(1, sims),
            "games_finished": games, "records_gathered": int(allrecs.shape[0]), "errors": errors,
            "nodes_dropped": dropped,
            "roofline": {"bound": "mfma", "kernel": "k_tower_bf16 (one whole net forward of the G x T leaf slots: stem + 2B conv layers + both heads, one launch)"
                         if a.dtype == "bf16" else "fp32 conv chain", "achieved": achieved / 1e12,
                         "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak,
                         "flop_per_launch": leaves_per_launch * fps, "leaves_per_launch": leaves_per_launch,
                         "leaf_slots_per_launch": a.games * a.threads, "avg_launch_ms": prof["net_ms"],
                         "tree_step_avg_ms": prof["tree_ms"], "timed_launches": prof["launches"],
                         # HBM-side bytes per launch from rocprofv3 PMC (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction),
                         # profiles/r01_final_pmc_fetch_write_g256_s100_t{1,2}_b20.txt — measured for exactly these configurations
                         "traffic": {(256, 1, 20, "bf16"): (2 * 186413.07 + 56.0) * 1024,
                                     (256, 2, 20, "bf16"): (2 * 186194.60 + 96.0) * 1024}.get(
                                         (a.games, a.threads, a.blocks, a.dtype))},
        }
        # the HBM-side part of a simulation: node reads / backup writes per tree level, leaf record, prior and
        # node write per evaluation, state + control block per game and pass.  One wavefront walks one game's tree, so
        # this kernel is bound by dependent-access latency, not by bandwidth: the fraction below says how far from it.
        lv, ev = levels / max(1, world), evals / max(1, world)
        tree_bytes = lv * (640 + 64 + 12 + 4) + ev * (96 + 64 + 16 + 176 + 4 + 640 + 8) + a.steps * a.games * (64 + 128) * 2
        tree_s = prof["tree_ms"] * 1e-3 * a.steps
        out["tree_step"] = {"bound": "hbm", "kernel": "k_tree_step (select / expand / backup / decision for every game, one wavefront each)",
                            "achieved": tree_bytes / tree_s / 1e9 if tree_s > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                            "frac": tree_bytes / tree_s / 8e12 if tree_s > 0 else 0.0,
                            "bytes_per_simulation": tree_bytes / max(1.0, sims / max(1, world)), "avg_launch_ms": prof["tree_ms"],
                            "note": "latency-bound pointer chasing (mean depth %.2f); 4-5 %% of the step" % (levels / max(1, sims))}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.blocks, a.sims, a.threads)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
