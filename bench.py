#!/usr/bin/env python3
"""bench.py — MCTS simulations/s of device-resident AlphaZero-Risk self-play on MI355X.

Headline workload = the point BASELINE.json's north_star quotes the metric on: 512 concurrent self-play games per
GPU, 100 MCTS simulations per move, THREADS_PER_MCTS 2 (the reference's default, src/settings.h:44), random-init
20-block / 256-filter net, bf16 MFMA contractions with fp32 accumulation.  At N = 1 the other two single-GPU
configurations of BASELINE.json (configs[1] 256 x 100 and configs[2] 2048 x 400) run in the same invocation and are
reported under "extra_configs", each with its own roofline (+ the headline point on NET_F32X and NET_F16); "midgame_leg" = the headline
entered in the middle of games; "config0_play" = BASELINE configs[0] through the host CLI; "optimiser_step" and "arena_100" = the other
two legs of a learn iteration at their stated shapes (batch 512; 100 mirrored games).

One "step" = one MOVE's worth of the hot path for every game of the batch: S/T + 1 passes, a pass being one tree step
(expand + backup of the previous leaves, PUCT descents to the next leaves — and, when a search completes, the move,
the (s, pi) record and possibly a game restart — for all G games) followed by one batched net evaluation of the
G x T leaf slots.  --steps K times exactly K x (S/T + 1) passes.  Everything is resident in HBM; nothing crosses PCIe
in the timed region except the final counters.

Multi-GPU (weak scaling): one process per GPU, games sharded with disjoint seed streams, NO data-path collective.
`python bench.py --gpus N` starts the N ranks itself (before anything touches torch or HIP); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of them.  The path's one exchange — the
all_gather of finished (s, pi, z) records, alphazero_trainer.cpp:59-62 — is measured on real records after an untimed
tail that plays games to their end, and reported under "exchange" (records, bytes, gather_ms).

    python bench.py [--gpus N --steps K --warmup W] [--games 512 --sims 100 --threads 2 --blocks 20 --dtype bf16]
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SIM = {20: 2 * 797_976_348, 5: 2 * 200_288_028}  # valid (un-padded) taps only
PEAK_BF16 = 2.5e15   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32 = 157.3e12  # fp32 vector/matrix
# BASELINE.json configs[1], configs[2] (games, sims/move, THREADS_PER_MCTS), and configs[1] at -t 4: with 256 games the leaf
# batch of a pass is games x THREADS_PER_MCTS, and 1024 leaf slots fill every CU with a 4-board tile
EXTRA_CONFIGS = [(256, 100, 2), (2048, 400, 2), (256, 100, 4)]
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written by tools/pmc_summary.py from rocprofv3 --pmc runs


def flop_per_sim(blocks):
    if blocks in FLOP_PER_SIM:
        return FLOP_PER_SIM[blocks]
    mac = 304 * 13 * 256 + blocks * 2 * 304 * 256 * 256 + 46_876
    return 2 * mac


def config_key(games, sims, threads, blocks, dtype):
    return f"g{games}_s{sims}_t{threads}_b{blocks}_{dtype}"


def measured_traffic(games, sims, threads, blocks, dtype):
    """HBM-side bytes per launch of the dominant kernel for exactly this configuration, from the committed rocprofv3 PMC
    summary (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) — None when no profile of this
    configuration has been taken."""
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f).get(config_key(games, sims, threads, blocks, dtype), {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


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


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` outside a launcher: this parent has not imported torch nor touched HIP; it starts one
    fresh child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays rank 0's JSON line and fails if any rank
    fails.  (Never an exec from a process that has initialised the GPU.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    import threading

    procs, errs = [], []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stdout is the result line; every other rank's stdout + everybody's stderr is kept for the failure report
        errs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else errs[-1], stderr=errs[-1]))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()   # rank 0's pipe is drained while ALL ranks are watched
    deadline = time.monotonic() + float(os.environ.get("AZR_BENCH_DEADLINE_S", "3000"))
    why = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        if any(rc not in (None, 0) for rc in rcs):
            why = "a rank failed"
        elif time.monotonic() > deadline:
            why = "deadline passed"
        if why:   # the survivors would wait in a collective for ever: end them (terminate, then kill)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.monotonic() + 10
            while time.monotonic() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    out = b"".join(chunks)
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if why or any(rcs):
        for r, f in enumerate(errs):
            f.seek(0)
            tail = f.read().decode(errors="replace")[-4000:]
            if tail.strip():
                sys.stderr.write(f"---- rank {r} (exit code {rcs[r]}) ----\n{tail}\n")
        raise SystemExit(f"bench.py: {why or 'a rank failed'}; rank exit codes {rcs}")
    line = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    if not line or json.loads(line[-1]).get("n_gpus") != a.gpus:
        raise SystemExit("bench.py: rank 0 did not report n_gpus == --gpus")


def midgame_states(games, rank):
    """`games` running positions from the committed golden games (tests/golden/rules_games.npz: 20 seeded games of the REAL
    reference played with random legal moves, every phase and every stage of a game), evenly spread, with an RNG stream each"""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "rules_games.npz"))
    st = g["states"]
    ends = set(int(x) - 1 for x in g["starts"][1:])            # the last recorded state of a game precedes its final move
    live = np.array([i for i in range(len(st)) if i not in ends])
    idx = live[(np.arange(games) * len(live)) // games]
    phases = np.bincount(st[idx, 149], minlength=6).tolist()   # Data::roundPhase (state/state.h:86-105)
    seeds = (np.arange(games, dtype=np.uint64) * 7919 + 104729 * (rank + 1) + 1) % 2147483646 + 1
    return st[idx].copy(), seeds.astype(np.uint32), phases


def config0_play(blocks):
    """BASELINE configs[0], `-m play --mcts=16 --cg=100` (AlphaZero vs ScriptPlayer), at its stated size through the C++
    host CLI above the C-ABI: wall time of the whole process (engine creation, random-init checkpoint, 100 games)"""
    host = os.path.join(ROOT, "alphazero-risk_amd", "host")
    exe = os.path.join(host, "AlphaZero_Risk_hip")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-s", "-C", host], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            cmd = [exe, "-m", "play", "--mcts=16", "--cg=100", f"--blocks={blocks}"]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=900)
            dt = time.perf_counter() - t0
        tail = r.stdout.strip().split("\n")[-4:]
        if r.returncode != 0 or not tail or tail[0] != "Games: 100":
            return {"error": (r.stderr or r.stdout)[-300:]}
        d_, p1, p2 = (int(t.split(":")[1]) for t in tail[1:])
        return {"command": "AlphaZero_Risk_hip " + " ".join(cmd[1:]), "wall_s": dt, "games": 100, "draws": d_, "alphazero_wins": p1,
                "script_wins": p2, "games_per_s": 100 / dt,
                "note": "random-init net; the reference's own CPU run of this command with its _DEBUG random-NN fake took 3.4 s (BASELINE.md §2)"}
    except Exception as e:   # noqa: BLE001  (a missing compiler on the box must not cost the headline)
        return {"error": repr(e)[-300:]}


def optimiser_step_leg(pkg, blocks, bs=512, batches=10, epochs=12):
    """the native optimiser step (azr_nn_train) at the reference's training shape (BATCH_SIZE 512, settings.h:74; learn loop row f-2):
    `batches` minibatches of self-play records x `epochs` in one call (120 steps: the call's fixed part — record upload, the refold /
    repack of the weights for inference afterwards — is ~1 % of it), ms per minibatch step"""
    import numpy as np
    eng = pkg.Engine(64, blocks=blocks, sims=8, dtype=pkg.NET_BF16)
    eng.init_random(1)
    eng.selfplay_start(7)
    recs = []
    while sum(len(r) for r in recs) < bs * batches:
        eng.selfplay_run(64)
        recs.append(eng.drain())
    rec = np.concatenate(recs)[:bs * batches]
    eng.train(rec[:bs], 1, batch_size=bs, rng_state=1)   # allocate + warm up
    t0 = time.perf_counter()
    hist, _ = eng.train(rec, epochs, batch_size=bs, rng_state=1)
    dt = time.perf_counter() - t0
    eng.close()
    steps = epochs * batches
    return {"ms_per_step": 1e3 * dt / steps, "steps": steps, "batch": bs, "blocks": blocks,
            "what": "azr_nn_train: gather + forward + backward + Adam of one minibatch of 265-byte records, fp32-equivalent split arithmetic on the MFMA",
            "last_epoch_losses": [float(x) for x in hist[-1]]}


def arena_leg(pkg, blocks, games=100, slots=128, sims=100, threads=2):
    """the learn loop's new-vs-old comparison (GameGroup::playGames with two AlphaZero players and two networks, game.cpp:277-312;
    alphazero_trainer.cpp:147-152) on the device: wall time of `games` mirrored games — the two games of a pair at the same time on
    two slots (AZR_MIRROR_CONCURRENT, the learn loops' default), and the reference's thread-per-pair form (one after the other on one
    slot) beside it"""
    new = pkg.Engine(slots, blocks=blocks, sims=sims, dtype=pkg.NET_BF16, threads=threads)
    old = pkg.Engine(slots, blocks=blocks, sims=sims, dtype=pkg.NET_BF16, threads=threads)
    new.init_random(1)
    old.init_random(2)
    new.arena_set_opponent(old)
    out = {}
    for name, mode in (("concurrent", pkg.MIRROR_CONCURRENT), ("sequential", pkg.MIRROR_SEQUENTIAL)):
        new.arena_start(pkg.PLAYER_ALPHAZERO, pkg.PLAYER_ALPHAZERO_B, games, 0, mode, 20260001)
        t0 = time.perf_counter()
        while not new.arena_run(256):
            pass
        out[name] = (time.perf_counter() - t0, new.arena_results())
    fb = new.counters()["tower_fallbacks"]
    new.arena_set_opponent(None)
    new.close(); old.close()
    return {"wall_s": out["concurrent"][0], "games": games, "slots": slots, "sims_per_move": sims, "mcts_threads": threads, "blocks": blocks,
            "results": out["concurrent"][1], "sequential_wall_s": out["sequential"][0], "sequential_results": out["sequential"][1],
            "tower_fallbacks": fb,
            "what": "two-net arena, 50 mirrored pairs: `wall_s` with a pair's two games at the same time on two slots (100 slots busy), "
                    "`sequential_wall_s` with each pair's games one after the other on one slot (the reference's thread-per-pair form); "
                    "passes queued without a read-back; k_tower_sc while both nets' launches fit the CUs, one board per workgroup otherwise"}


def run_config(ctx, games, sims, threads, steps, warmup, tail, dtype=None, midgame=False):
    """K steps of one configuration on this rank's GPU; returns the per-rank measurements (reduced by the caller)"""
    a, pkg, shard, torch, dist = ctx["a"], ctx["pkg"], ctx["shard"], ctx["torch"], ctx["dist"]
    dtype = dtype or a.dtype
    rank, world, local, dev, cdev = ctx["rank"], ctx["world"], ctx["local"], ctx["dev"], ctx["cdev"]
    use_dist = ctx["use_dist"]   # world > 1, or AZR_FORCE_DIST=1: a one-rank world still goes through every collective
    passes_per_step = sims // threads + 1   # setRootState's root expansion + (S - S % T) / T lock-stepped rounds
    eng = pkg.Engine(games, blocks=a.blocks, sims=sims, dtype={"bf16": pkg.NET_BF16, "f16": pkg.NET_F16, "f32": pkg.NET_F32, "f32x": pkg.NET_F32X}[dtype],
                     device=local, threads=threads)
    eng.init_random(20260002)
    phases = None
    if midgame:   # the move loop entered in the middle of games: every phase, every stage of a game from the first pass on
        st, seeds, phases = midgame_states(games, rank)
        eng.set_states(st)
        eng.set_rng(seeds)
        eng.selfplay_start_from_states(shard.rank_base_seed(20260001, rank))
    else:
        eng.selfplay_start(shard.rank_base_seed(20260001, rank))
    tower_kernel_name = "fp32 conv chain"
    if dtype == "f32x":
        tower_kernel_name = (f"k_tower_fx<2> x {(games * threads + 1) // 2} workgroups of 2 boards (one whole net forward at fp32-equivalent precision: "
                             "fp16-pair operands, 3 MFMA passes per conv layer, fp32 accumulate / epilogue / residual / heads, one launch)")
    if dtype in ("bf16", "f16"):   # the tile plan of this configuration's launches (G x T leaf slots), from the library itself
        nb, wgs = eng.tower_plan(games * threads)
        tower_kernel_name = (f"k_tower_sb<{nb}>" if nb > 1 else "k_tower_bf16<1>") + \
            f" x {wgs} workgroups of {nb} board(s) (one whole net forward of the G x T leaf slots: stem + 2B conv layers + both heads, one launch" + \
            (", fp16 operands)" if dtype == "f16" else ")")

    def barrier():
        torch.cuda.synchronize()
        eng.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    t_start = time.perf_counter()
    eng.selfplay_run(warmup * passes_per_step)
    c0 = eng.counters()
    barrier()
    t0 = time.perf_counter()
    # EXACTLY K steps = K x passes_per_step passes, issued in up to 10 slices so that a median rate can be reported next
    # to the mean; a slice boundary is one counter read-back (a stream sync of a few microseconds)
    chunk_rates, profs, done, prev_sims, prev_t = [], [], 0, c0["simulations"], t0
    for i in range(10):
        n = (steps * (i + 1)) // 10 - done
        if n <= 0:
            continue
        eng.selfplay_run(n * passes_per_step)
        done += n
        now, cs = time.perf_counter(), eng.counters()["simulations"]
        chunk_rates.append((cs - prev_sims) / max(now - prev_t, 1e-9))
        profs.append(eng.profile_last_run())
        prev_sims, prev_t = cs, now
    barrier()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    # HIP-event timings of the kernels (on the engine's stream), launch-weighted over the slices of the timed region
    nl = max(1, sum(p_["launches"] for p_ in profs))
    prof = {"net_ms": sum(p_["net_ms"] * p_["launches"] for p_ in profs) / nl,
            "tree_ms": sum(p_["tree_ms"] * p_["launches"] for p_ in profs) / nl, "launches": sum(p_["launches"] for p_ in profs)}
    delta = {k: c1[k] - c0[k] for k in c1}
    keys = ["simulations", "evaluations", "levels", "decisions", "games_finished", "samples", "errors", "nodes_dropped",
            "records_dropped"]
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    tot = torch.tensor([delta[k] for k in keys], dtype=torch.int64, device=cdev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    tot = dict(zip(keys, (int(x) for x in tot.tolist())))

    # untimed tail: play on until a quarter of this rank's games have finished (bounded), which gives whole self-play
    # games/s and REAL finished records for the path's one exchange step; then time that exchange on its own
    games_rate, exchange = None, None
    if tail:
        t_tail = time.perf_counter()
        while time.perf_counter() - t_tail < a.tail_seconds:
            eng.selfplay_run(40 * passes_per_step)
            if eng.counters()["games_finished"] >= max(1, games // 4):
                break
        ct = eng.counters()
        el = time.perf_counter() - t_start
        fin = torch.tensor([ct["games_finished"], ct["samples"], ct["simulations"]], dtype=torch.float64, device=cdev)
        elt = torch.tensor([el], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(fin, op=dist.ReduceOp.SUM)
            dist.all_reduce(elt, op=dist.ReduceOp.MAX)
        gf, smp, sm = fin.tolist()
        el = float(elt.item())
        if gf > 0:
            games_rate = {"games_per_s": gf / el, "games": int(gf), "decisions_per_finished_game": smp / gf,
                          "window_s": el, "simulations_per_s_over_window": sm / el}
        # the exchange: every rank's finished records, device-to-device into the send buffer, padded all_gather (RCCL over
        # xGMI at N > 1; at N = 1 the same code path without the collective)
        barrier()
        tg = time.perf_counter()
        recs = shard.device_records_to_torch(eng, dev)
        allrecs = shard.gather_records(recs if cdev == dev else recs.to(cdev), dist if use_dist else None)
        barrier()
        gms = 1e3 * (time.perf_counter() - tg)
        gt = torch.tensor([gms], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        exchange = {"records_gathered": int(allrecs.shape[0]), "bytes": int(allrecs.shape[0]) * 265,
                    "records_this_rank": int(recs.shape[0]), "gather_ms": float(gt.item()),
                    "self_play_window_s": el, "fraction_of_window": float(gt.item()) * 1e-3 / el,
                    "gathered_on": str(allrecs.device),
                    "collective": "all_gather (counts) + padded all_gather (records), backend %s" % ctx["backend"]
                                  if use_dist else "none at N = 1 (device copy of the record ring only)"}
        assert ct["records_dropped"] == 0, "records were dropped: raise sample_capacity"
    eng.close()

    fps = flop_per_sim(a.blocks)
    net_s = prof["net_ms"] * 1e-3
    npass = steps * passes_per_step
    # algorithmic work of one launch = the leaves that were actually waiting for the net (idle slots are not counted)
    leaves_per_launch = tot["evaluations"] / max(1, npass * world)
    achieved = leaves_per_launch * fps / net_s if net_s > 0 else 0.0
    peak = PEAK_F32 if dtype == "f32" else PEAK_BF16   # f32x runs on the fp16 MFMA (same dense peak as bf16)
    sims_total, levels = tot["simulations"], tot["levels"]
    out = {
        "value": sims_total / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
        "passes_per_step": passes_per_step, "timed_region_s": dt,
        "config": {"workload": f"{games} concurrent self-play games/GPU x {sims} MCTS sims/move, THREADS_PER_MCTS {threads}, "
                               f"{a.blocks}-block 256-filter random-init net, {dtype}",
                   "games_per_gpu": games, "sims_per_move": sims, "mcts_threads": threads, "blocks": a.blocks,
                   "step": f"{passes_per_step} passes (pass = tree step + net forward of the {games * threads} leaf slots) = one decision of every "
                           f"game whose root has to be expanded first; a root that survived the trim needs {sims // threads} (see decisions_per_game_and_step)",
                   "parallelism": f"games sharded x{world}, no data-path collective; 1 all_gather of finished records per iteration"},
        "simulations_per_s_p50_rank0": sorted(chunk_rates)[len(chunk_rates) // 2] if chunk_rates else None,
        "self_play_games_per_s": games_rate["games_per_s"] if games_rate else None,
        "self_play_games_window": games_rate, "decisions_per_s": tot["decisions"] / dt,
        "decisions_per_game_and_step": tot["decisions"] / max(1, steps * games * world),
        "net_evals_per_s": tot["evaluations"] / dt, "mean_depth": levels / max(1, sims_total),
        "games_finished_in_timed_region": tot["games_finished"], "errors": tot["errors"],
        "nodes_dropped": tot["nodes_dropped"], "records_dropped": tot["records_dropped"],
        "dtype": dtype,
        "roofline": {"bound": "mfma", "kernel": tower_kernel_name, "achieved": achieved / 1e12,
                     "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak,
                     "flop_per_launch": leaves_per_launch * fps, "leaves_per_launch": leaves_per_launch,
                     "leaf_slots_per_launch": games * threads, "avg_launch_ms": prof["net_ms"],
                     "tree_step_avg_ms": prof["tree_ms"], "timed_launches": prof["launches"],
                     # HBM-side bytes per launch from the committed rocprofv3 PMC summary of THIS configuration (null if none)
                     "traffic": measured_traffic(games, sims, threads, a.blocks, dtype)},
    }
    if dtype == "f32x":   # achieved / frac count the ALGORITHMIC flops of an fp32 evaluation; the kernel issues 3 fp16 MFMA passes for them
        out["roofline"]["note"] = ("fp32-equivalent: `achieved` = algorithmic (valid-tap) flops of ONE evaluation per board / launch time; the kernel "
                                   "issues 3 x that on the fp16 MFMA, so the matrix pipe's own utilisation is ~3 x `frac` (issued_frac)")
        out["roofline"]["issued_frac"] = 3 * achieved / peak
    if midgame:
        out["start"] = {"from": "tests/golden/rules_games.npz (positions of 20 seeded reference games, random legal moves)",
                        "positions_by_phase": dict(zip(["SETUP", "SETUP_NEUTRAL", "REINFORCEMENT", "ATTACK", "ATTACK_MOBILIZATION", "FORTIFY"], phases))}
    if exchange:
        out["exchange"] = exchange
        out["records_gathered"] = exchange["records_gathered"]
    # the HBM-side part of a simulation: node reads / backup writes per tree level, leaf record, prior and
    # node write per evaluation, state + control block per game and pass.  One wavefront walks one game's tree, so
    # this kernel is bound by dependent-access latency, not by bandwidth: the fraction below says how far from it.
    lv, ev = levels / max(1, world), tot["evaluations"] / max(1, world)
    tree_bytes = lv * (640 + 64 + 12 + 4) + ev * (96 + 64 + 16 + 176 + 4 + 640 + 8) + npass * games * (64 + 128) * 2
    tree_s = prof["tree_ms"] * 1e-3 * npass
    out["tree_step"] = {"bound": "hbm", "kernel": "k_tree_step (select / expand / backup / decision for every game, one wavefront each)",
                        "achieved": tree_bytes / tree_s / 1e9 if tree_s > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                        "frac": tree_bytes / tree_s / 8e12 if tree_s > 0 else 0.0,
                        "bytes_per_simulation": tree_bytes / max(1.0, sims_total / max(1, world)), "avg_launch_ms": prof["tree_ms"],
                        "share_of_step": prof["tree_ms"] * npass / (1e3 * dt) if dt > 0 else None,
                        "note": "latency-bound pointer chasing (mean depth %.2f)" % (levels / max(1, sims_total))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; a step = one move for every game = sims/threads + 1 passes")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--games", type=int, default=512, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--threads", type=int, default=2,
                    help="THREADS_PER_MCTS (-t): search threads per game; 2 is the reference's default (src/settings.h:44)")
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "f32x"],
                    help="bf16 (the benchmarked tower) | f16 (the same kernels on fp16 operands) | f32x (fp32-equivalent fp16-pair MFMA tower) | f32 (fp32 VALU kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip BASELINE configs[1] / configs[2] (N = 1 runs them by default)")
    ap.add_argument("--no-full-games", action="store_true",
                    help="skip the untimed tail (whole self-play games/s, and the record exchange on real records)")
    ap.add_argument("--tail-seconds", type=float, default=75.0)
    a = ap.parse_args()
    if a.gpus < 1 or a.steps < 1 or a.warmup < 0:
        raise SystemExit("bench.py: --gpus >= 1, --steps >= 1, --warmup >= 0")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        return spawn_ranks(a, sys.argv[1:])          # before torch / HIP are touched
    world = int(env_world or "1")
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus}, or plain python bench.py --gpus {a.gpus})")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import importlib

    import torch                      # first: the C-ABI library then binds to the same HIP runtime (shard._one_hip_runtime)
    import torch.distributed as dist

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
    use_dist = world > 1 or os.environ.get("AZR_FORCE_DIST", "0") not in ("", "0")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # AZR_FORCE_DIST=1: the N > 1 code path (RCCL collectives on device tensors) with one rank
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("alphazero-risk_amd")
    shard = importlib.import_module("alphazero-risk_amd.shard")
    ctx = dict(a=a, pkg=pkg, shard=shard, torch=torch, dist=dist, rank=rank, world=world, local=local, dev=dev, cdev=cdev,
               backend=backend, use_dist=use_dist)
    head = run_config(ctx, a.games, a.sims, a.threads, a.steps, a.warmup, tail=not a.no_full_games)
    extras, mid = [], None
    if world == 1 and not a.no_extra:
        for g, s, t in EXTRA_CONFIGS:
            if (g, s, t) == (a.games, a.sims, a.threads):
                continue
            # the same number of net launches as the headline region, at least 2 steps
            k = max(2, (a.steps * (a.sims // a.threads + 1)) // (s // t + 1))
            w = max(1, (a.warmup * (a.sims // a.threads + 1)) // (s // t + 1))
            e = run_config(ctx, g, s, t, k, w, tail=False)
            e["metric"], e["unit"] = "MCTS simulations/s", "simulations/s"
            extras.append(e)
        # the headline configuration entered in the MIDDLE of games (all phases, deep trees) instead of after the deal
        e = run_config(ctx, a.games, a.sims, a.threads, a.steps, a.warmup, tail=False, midgame=True)
        e["metric"], e["unit"] = "MCTS simulations/s", "simulations/s"
        mid = e
        if a.dtype == "bf16":   # the north-star point again at the reference's precision (it evaluates in fp32, alphazero_nn.cpp:247-248)
            k = max(2, a.steps // 4)
            e = run_config(ctx, a.games, a.sims, a.threads, k, max(1, a.warmup // 4), tail=False, dtype="f32x")
            e["metric"], e["unit"] = "MCTS simulations/s", "simulations/s"
            extras.append(e)
            # ... and on fp16 operands: the bf16 tower's kernels and rate, ~9x closer to the fp32 evaluation (AZR_NET_F16)
            e = run_config(ctx, a.games, a.sims, a.threads, a.steps, a.warmup, tail=False, dtype="f16")
            e["metric"], e["unit"] = "MCTS simulations/s", "simulations/s"
            extras.append(e)
    if rank == 0:
        out = {"metric": "MCTS simulations/s", "value": head.pop("value"), "unit": "simulations/s", "n_gpus": world,
               "steps": head.pop("steps"), "warmup": head.pop("warmup"), "ms_per_step": head.pop("ms_per_step"),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic"}
        out.update(head)
        if extras:
            out["extra_configs"] = extras
        if mid is not None:
            out["midgame_leg"] = mid
            out["midgame_leg"]["vs_headline"] = mid["value"] / out["value"]
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.blocks, a.sims, a.threads)
        if world == 1 and not a.no_extra:
            out["config0_play"] = config0_play(a.blocks)
            # the two other legs of the reference's learn iteration, each at its stated shape, ~12 s together
            out["optimiser_step"] = optimiser_step_leg(pkg, a.blocks)
            out["arena_100"] = arena_leg(pkg, a.blocks)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
