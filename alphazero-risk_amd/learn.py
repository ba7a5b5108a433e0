"""`-m train` / `-m learn`: the reference's learn loop (AlphaZeroTrainer::train, alphazero_trainer.cpp:12-34,121-198)
over the MI355X engine:

    self-play (device-resident, azr_selfplay_*)  ->  replay buffer (trimOldExamples, alphazero_nn_data.cpp:67-84)
    ->  train step (azr_nn_train: the HIP optimiser step of csrc/azr_train.hip)  ->  arena new-vs-old resident on the
    device (two engines = two nets, one tree per player as in the reference)  ->  accept (>= COMPARE_TRESHOLD of decided games)
    / revert  ->  benchmark vs RandomPlayer(10) and ScriptPlayer(100) on the device arena
with the reference's log files (log/azr-improvement-log.txt, azr-benchmark-log.txt, azr-nn-training-log.txt) and
checkpoint names (checkpoints/{latest,best}-checkpoint.bin, checkpoint-iter-N.bin; AZRW container).

    python -m alphazero-risk_amd.learn ...     is not importable as a module name with a hyphen; run
    python alphazero-risk_amd/learn.py --ti 2 --tg 256 --mcts 100 --gpu-games 256 --blocks 20 --cg 100

Multi-GPU (BASELINE configs[4]): one process per GPU under torchrun,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 alphazero-risk_amd/learn.py ...
Self-play, the arena and the benchmark games are sharded over the ranks with no collective inside them; the exchange
steps of an iteration are: all_gather of the new (s, pi, z) records, broadcast of the trained weights from rank 0 (the
reference trains on GPU 0 and hands the weights over through temp.bin), all_reduce of the six GameResults counters.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
P = importlib.import_module("alphazero-risk_amd")
shard_mod = importlib.import_module("alphazero-risk_amd.shard")


def arena_device(eng_new, eng_old, games, mirror=P.MIRROR_CONCURRENT, base_seed=1, collect=False):
    """the same arena resident on the device (azr_arena_* with AZR_PLAYER_ALPHAZERO_B = eng_old's network): every slot
    advances on its own, each network is evaluated on its own player's leaves; optionally returns the (s, pi, z)
    records both players produced (INCLUDE_COMPARE_GAMES_TRAIN_SAMPLES)"""
    eng_new.arena_set_opponent(eng_old)
    eng_new.arena_collect_samples(collect)
    eng_new.arena_start(P.PLAYER_ALPHAZERO, P.PLAYER_ALPHAZERO_B, games, 0, mirror, base_seed)
    while not eng_new.arena_run(4 * (eng_new.settings.mcts_simulations + 2)):
        pass
    gr = eng_new.arena_results()
    recs = eng_new.drain() if collect else np.zeros((0, 265), np.uint8)
    eng_new.arena_collect_samples(False)
    eng_new.arena_set_opponent(None)
    return gr, recs


def is_model_improved(gr, threshold):
    """AlphaZeroTrainer::isModelImproved (alphazero_trainer.cpp:192-198): `int >= int * float` evaluated in fp32"""
    return bool(np.float32(gr["win"][0]) >= np.float32(gr["win"][0] + gr["win"][1]) * np.float32(threshold))


def gr_str(gr):  # operator<<(GameResults) (game.cpp:227-235)
    return f"{gr['draw']}, {gr['win'][0]}/{gr['win_and_started'][0]}, {gr['win'][1]}/{gr['win_and_started'][1]}"


def improvement_line(it, gr):
    """one line of log/azr-improvement-log.txt (alphazero_trainer.cpp:163): iteration,GameResults"""
    return f"{it},{gr_str(gr)}\n"


def benchmark_line(it, random_gr, script_gr):
    """one line of log/azr-benchmark-log.txt (alphazero_trainer.cpp:139): iteration,<vs Random>, <vs Script>"""
    return f"{it},{gr_str(random_gr)}, {gr_str(script_gr)}\n"


def nn_training_line(hist):
    """one line of log/azr-nn-training-log.txt (LOG_NN_TRAINING): "policy loss, value loss, " per epoch, as `ostream <<
    float` prints them (6 significant digits)"""
    return "".join(f"{lp:g}, {lv:g}, " for lp, lv in hist) + "\n"


def trim_old_examples(records, old_game_index, smin, smax):
    """NNTrainDataStorage::trimOldExamples (alphazero_nn_data.cpp:67-84)"""
    n = len(records)
    if n > smax:
        return records[n - smax:], old_game_index
    if n > smin and old_game_index > 0:
        excess = min(old_game_index, n - smin)
        return records[excess:], old_game_index - excess
    return records, old_game_index


def save_training_samples(path, records):
    """NNTrainDataStorage::saveTrainingSamples (alphazero_nn_data.cpp:112-136): size_t count (8 bytes) + 265 bytes per record;
    nothing is written for an empty buffer"""
    records = np.ascontiguousarray(records, np.uint8).reshape(-1, 265)
    if len(records) == 0:
        return False
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(np.uint64(len(records)).tobytes())
        f.write(records.tobytes())
    return True


def load_training_samples(path):
    """NNTrainDataStorage::loadTrainingSamples (alphazero_nn_data.cpp:86-110).  The reference's writer emits an 8-byte
    count while its reader consumes 4 bytes (so it mis-reads its own files by 4 bytes): both header widths are accepted
    here, chosen by the file size, and the records come back intact.  A missing file is an empty buffer."""
    if not os.path.exists(path):
        return np.zeros((0, 265), np.uint8)
    raw = np.fromfile(path, np.uint8)
    if len(raw) >= 8:
        n8 = int(raw[:8].view(np.uint64)[0])
        if len(raw) == 8 + n8 * 265:
            return raw[8:].reshape(n8, 265).copy()
    if len(raw) >= 4:
        n4 = int(raw[:4].view(np.int32)[0])
        if n4 >= 0 and len(raw) == 4 + n4 * 265:
            return raw[4:].reshape(n4, 265).copy()
    raise ValueError(f"unrecognised sample file {path}: {len(raw)} bytes")


def run_arena(eng, p1, p2, games, mirror, seed):
    """GameGroup::playGames of an AlphaZero player group against ScriptPlayer / RandomPlayer (the benchmark games)"""
    eng.arena_start(p1, p2, games, 0, mirror, seed)
    while not eng.arena_run(4 * (eng.settings.mcts_simulations + 2)):
        pass
    return eng.arena_results()


def reduce_results(gr, dist, dev):
    """GameResults::add over ranks"""
    flat = dict(count=gr["count"], draw=gr["draw"], w0=gr["win"][0], w1=gr["win"][1], s0=gr["win_and_started"][0],
                s1=gr["win_and_started"][1])
    r = shard_mod.reduce_counters(flat, dist, device=dev)
    return dict(count=r["count"], draw=r["draw"], win=[r["w0"], r["w1"]], win_and_started=[r["s0"], r["s1"]])


class Deadline:
    """watchdog of a multi-rank run: a collective whose partner died never returns (an in-stream ncclAllReduce blocks the next
    stream synchronisation, a gloo / RCCL call of torch blocks in the call), so a timer thread ends THIS process with a message
    and exit code 124 when a phase overruns; the launcher (torchrun, bench.py's parent) then ends the other ranks.
    arm(what) starts the clock of the next phase (and stops the previous one's), disarm() stops it."""

    def __init__(self, seconds, rank):
        self.seconds, self.rank, self.timer = seconds, rank, None

    def arm(self, what):
        self.disarm()
        if self.seconds and self.seconds > 0:
            import threading
            self.timer = threading.Timer(self.seconds, self._expired, args=(what,))
            self.timer.daemon = True
            self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _expired(self, what):
        sys.stderr.write(f"learn.py: rank {self.rank} spent more than {self.seconds:.0f} s in '{what}': a rank is missing from a "
                         "collective; ending this process (exit code 124)\n")
        sys.stderr.flush()
        os._exit(124)


def learn(a, log=print, dist=None, rank=0, world=1, cdev="cpu"):
    os.makedirs("log", exist_ok=True)
    os.makedirs("checkpoints", exist_ok=True)
    dtype = {"bf16": P.NET_BF16, "f16": P.NET_F16, "f32x": P.NET_F32X, "f32": P.NET_F32}[a.dtype]
    t = getattr(a, "t", 2)
    # MIRROR_GAMES (settings.h:52): mirrored pairs; --pair-halves 1 plays a pair's two games at the same time on two slots
    mirror = P.MIRROR_CONCURRENT if getattr(a, "pair_halves", 1) and a.gpu_games >= 2 else P.MIRROR_SEQUENTIAL
    if rank != 0:
        log = lambda *_: None   # noqa: E731  (rank 0 reports)
    gen = P.Engine(a.gpu_games, blocks=a.blocks, sims=a.mcts, dtype=dtype, device=a.device, threads=t)
    new = P.Engine(a.gpu_games, blocks=a.blocks, sims=a.mcts, dtype=dtype, device=a.device, threads=t)
    latest, best = "checkpoints/latest-checkpoint.bin", "checkpoints/best-checkpoint.bin"
    # loadCheckpoint: missing => init + save (alphazero_nn.cpp:197-202).  Rank 0 creates the file before anybody reads it.
    if rank == 0 and not os.path.exists(latest):
        new.init_random(20260002)
        new.save(latest)
    if dist is not None:
        dist.barrier()
    for e in (new, gen):
        e.load(latest)
    shuffle_state = a.seed % 2147483646 + 1   # raw minstd_rand0 state standing in for the reference's global RNG
    records = load_training_samples("data/training_samples.bin")   # trainStorage.loadTrainingSamples(DEFAULT_SAMPLES)
    old_game_index = 0
    games_started = 0   # self-play games this rank has started in earlier iterations (its position in its seed stream)
    dp_native_ready = False
    sink = "/dev/null" if rank else None
    imp_log = open(sink or "log/azr-improvement-log.txt", "a")
    bench_log = open(sink or "log/azr-benchmark-log.txt", "a")
    nn_log = open(sink or "log/azr-nn-training-log.txt", "a")
    summary = []
    wd = Deadline(getattr(a, "phase_deadline", 0) if dist is not None else 0, rank)
    for it in range(a.ti):
        log(f"Train iteration {it}")
        wd.arm("self-play and record exchange")
        # ---- generateTrainData (alphazero_trainer.cpp:36-78)
        t0 = time.time()
        share = shard_mod.split_count(a.tg, world, rank)   # one self-play shard per GPU (alphazero_trainer.cpp:41-57)
        new_recs = [np.zeros((0, 265), np.uint8)]
        c = dict(games_finished=0, simulations=0, errors=0)
        if share > 0:
            # exactly `share` games, each played to its end (Counter::hasNext, alphazero_trainer.cpp:83); this rank's seed
            # stream continues where its previous iteration stopped, so no (iteration, rank) pair ever replays a game
            gen.selfplay_start_games(shard_mod.selfplay_seed(a.seed, rank, games_started), share)
            games_started += share
            while c["games_finished"] + c["errors"] < share:
                gen.selfplay_run(4 * (a.mcts + 2))
                c = gen.counters()
                new_recs.append(gen.drain())
            if c["records_dropped"]:
                raise RuntimeError(f"{c['records_dropped']} self-play records were dropped: raise sample_capacity")
        new_recs = np.concatenate(new_recs)
        if dist is not None:   # the one exchange step of data generation: records of all shards, in rank order
            import torch
            new_recs = shard_mod.gather_records(torch.from_numpy(new_recs).to(cdev), dist).cpu().numpy()
            c = shard_mod.reduce_counters({k: c[k] for k in ("games_finished", "simulations")}, dist, device=cdev)
            log(f"Record exchange: all_gather + counter all_reduce, backend {dist.get_backend()}, tensors on {cdev}, world {world}")
        dt = time.time() - t0
        log(f"Generated {len(new_recs)} new samples for total {len(records) + len(new_recs)}  "
            f"[{c['games_finished']} games, {c['simulations'] / dt:.0f} simulations/s, {c['games_finished'] / dt:.2f} games/s]")
        records = np.concatenate([records, new_recs])
        records, old_game_index = trim_old_examples(records, old_game_index, a.s, 16384 * a.bs)
        # ---- trainGroup->train (alphazero_gpu_cluster.cpp:221-231)
        wd.arm("training and weight hand-over")
        t0 = time.time()
        hist = []
        # --dp 0 (default): rank 0 trains and the others receive its weights, the reference's AlphaZeroNNGroup::train.  --dp 1: data-parallel
        # optimiser step.  A rank's share of the reference's BATCH_SIZE 512 on 8 ranks is 64 records, which the small-batch conv kernels
        # (t_conv_q: one board x 64 channels per block, 8-board weight-gradient slices) spread over the whole GPU: 4.2 ms of kernels against
        # 11.1 ms for all 512 records on one GPU — measured on ONE GPU with a one-rank stand-in for the collectives (tools/train_bench.py
        # --dp-world 8); what 2B + 6 latency-bound all-reduces and one of 95 MB cost over xGMI has not been measured, and equality of the
        # ranks' weights has been shown over gloo and in-process (tests/test_gpu_train_dp.py), not over RCCL with more than one rank:
        # opt-in until a run on a multi-GPU node is on record.
        dp_flag = getattr(a, "dp", 0)
        dp = dist is not None and dp_flag == 1 and a.bs % world == 0 and a.bs // world >= 2
        if dp:
            # data-parallel optimiser step: every rank takes 1/world of each minibatch (same shuffle stream everywhere);
            # batch statistics, losses and the gradient vector are all-reduced (RCCL over xGMI), every rank takes the
            # same Adam step — no weight broadcast
            if cdev != "cpu" and not getattr(a, "dp_callback", 0):
                # RCCL over xGMI on the engine's own stream: the engine gets a communicator of its own (once), every sum of the
                # step is stream-ordered
                if not dp_native_ready:
                    shard_mod.native_dp_init(new, dist, rank, world, cdev)
                    dp_native_ready = True
                hist, shuffle_state = new.train_dp(records, a.e, None, rank, world, batch_size=a.bs, rng_state=shuffle_state)
                how = "in-stream ncclAllReduce on the engine's own communicator"
            else:
                hist, shuffle_state = new.train_dp(records, a.e, shard_mod.make_allreduce(dist, cdev != "cpu", a.device), rank, world,
                                                   batch_size=a.bs, rng_state=shuffle_state)
                how = ("torch.distributed all_reduce on device buffers, backend " + dist.get_backend()) if cdev != "cpu" else "host copies (gloo rehearsal)"
            hist = [h for h in hist if not np.isnan(h[0])]
            log(f"Data-parallel optimiser step: {how}")
            if rank == 0:
                nn_log.write(nn_training_line(hist)); nn_log.flush()
        else:
            if rank == 0:   # AlphaZeroNNGroup::train: the first GPU trains (alphazero_gpu_cluster.cpp:221-231)
                hist, shuffle_state = new.train(records, a.e, batch_size=a.bs, rng_state=shuffle_state)
                hist = [h for h in hist if not np.isnan(h[0])]
                nn_log.write(nn_training_line(hist)); nn_log.flush()
            if dist is not None:   # ... and the others receive its weights
                w = shard_mod.broadcast_flat(new.get_weights(), dist, src=0, device=cdev)
                log(f"Weight broadcast from rank 0: backend {dist.get_backend()}, tensor on {cdev}")
                if rank != 0:
                    new.set_weights(w)
        steps = a.e * (len(records) // a.bs)
        if hist:
            log(f"Loss Policy / Value: {hist[-1][0]:f} / {hist[-1][1]:f}   [{steps} steps, {1e3 * (time.time() - t0) / max(steps, 1):.1f} ms/step]")
        # ---- updateIfImprovement (alphazero_trainer.cpp:134-190)
        wd.arm("compare and benchmark games")
        improved = True
        gr = None
        t_arena = time.time()
        if a.cg > 0:
            share = 2 * shard_mod.split_count(a.cg // 2, world, rank)
            aseed = shard_mod.rank_base_seed(a.seed + 7919 * (it + 1), rank)
            gr, arecs = arena_device(new, gen, share, mirror, aseed, collect=bool(getattr(a, "include_compare_samples", 1)))
            if dist is not None:
                gr = reduce_results(gr, dist, cdev)
                import torch
                arecs = shard_mod.gather_records(torch.from_numpy(arecs).to(cdev), dist).cpu().numpy()
            if len(arecs):   # playGames(..., trainStorage): the compare games' samples join the replay buffer
                records = np.concatenate([records, arecs])
                log(f"New samples generated from compare games {len(arecs)}")
            imp_log.write(improvement_line(it, gr)); imp_log.flush()
            improved = is_model_improved(gr, a.ct)
            log(f"Compare games: {gr_str(gr)}   [{gr['count']} games in {time.time() - t_arena:.1f} s]")
        if improved:
            log("Model improved")
            if rank == 0:
                new.save(best); new.save(f"checkpoints/checkpoint-iter-{it}.bin")
            gen.set_weights(new.get_weights())   # generateGroup->loadCheckpoint(best): every rank already holds them
            t_bench = time.time()
            r = run_arena(gen, P.PLAYER_ALPHAZERO, P.PLAYER_RANDOM, 2 * shard_mod.split_count(5, world, rank), mirror,
                          shard_mod.rank_base_seed(a.seed + 11 + 104729 * (it + 1), rank))
            s = run_arena(gen, P.PLAYER_ALPHAZERO, P.PLAYER_SCRIPT, 2 * shard_mod.split_count(50, world, rank), mirror,
                          shard_mod.rank_base_seed(a.seed + 13 + 15485863 * (it + 1), rank))
            if dist is not None:
                r, s = reduce_results(r, dist, cdev), reduce_results(s, dist, cdev)
            bench_log.write(benchmark_line(it, r, s)); bench_log.flush()
            log(f"Model benchmark: vs Random {r['win'][0]}/{r['count']}, vs Script {s['win'][0]}/{s['count']}   [{time.time() - t_bench:.1f} s]")
            old_game_index = max(len(records) - 1, 0)   # updateOldGamesIndex
        else:
            log("Model did not improve\nModel reverted back old")
            new.load(latest)
        wd.disarm()
        summary.append(dict(iteration=it, samples=len(records), losses=hist, arena=gr, improved=improved))
    # saveTrainingSamples (reference writer layout: 8-byte count + 265-byte records)
    if rank == 0:
        save_training_samples("data/training_samples.bin", records)
    gen.close(); new.close()
    return summary


def main():
    ap = argparse.ArgumentParser(description="AlphaZero-Risk learn loop on MI355X (reference flags of src/settings.h)")
    ap.add_argument("--ti", type=int, default=10000)      # TRAIN_ITERATIONS
    ap.add_argument("--tg", type=int, default=1000)       # TRAIN_ITERATION_GAMES
    ap.add_argument("--mcts", type=int, default=32)
    ap.add_argument("-t", type=int, default=2)            # THREADS_PER_MCTS
    ap.add_argument("--gpu-games", type=int, default=512,
                    help="concurrent games per GPU; 512 x THREADS_PER_MCTS 2 = 1024 leaf slots fill every CU with a 4-board tile")
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("-e", type=int, default=10)           # EPOCHS
    ap.add_argument("--bs", type=int, default=512)
    ap.add_argument("--cg", type=int, default=1000)       # COMPARE_GAMES
    ap.add_argument("--ct", type=float, default=0.55)
    ap.add_argument("-s", type=int, default=1024 * 512)   # SAMPLES_STORAGE_MIN
    ap.add_argument("--seed", type=int, default=20260001)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32x", "f32"])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--include-compare-samples", type=int, default=1)   # INCLUDE_COMPARE_GAMES_TRAIN_SAMPLES
    ap.add_argument("--pair-halves", type=int, default=1,
                    help="mirrored arena pairs: 1 = both games of a pair at the same time on two slots (AZR_MIRROR_CONCURRENT), "
                         "0 = one after the other on one slot (the reference's thread-per-pair form)")
    ap.add_argument("--dp", type=int, default=0,
                    help="multi-rank runs: 0 (default) = rank 0 trains and broadcasts (the reference's AlphaZeroNNGroup::train), "
                         "1 = data-parallel optimiser step over all ranks (opt-in: not yet measured on a multi-GPU node)")
    ap.add_argument("--phase-deadline", type=float, default=float(os.environ.get("AZR_LEARN_DEADLINE_S", "3600")),
                    help="multi-rank runs: seconds a rank may spend in one phase of an iteration (self-play + exchange, training, arena) before "
                         "it ends the process — a rank that died leaves the others waiting in a collective for ever otherwise")
    ap.add_argument("--dp-callback", type=int, default=0,
                    help="1 = the data-parallel step's sums go through torch.distributed (one host hand-over each) instead of the engine's own "
                         "in-stream RCCL communicator")
    a = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and not shard_mod.force_dist():
        learn(a)
        return
    # (AZR_FORCE_DIST=1 with one rank: the whole multi-rank code path — RCCL gather, counter all-reduce, weight broadcast or
    # data-parallel optimiser step — on a single GPU)
    import torch
    import torch.distributed as dist
    backend = os.environ.get("AZR_LEARN_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 path on fewer GPUs than ranks
    ndev = torch.cuda.device_count()
    a.device = local % max(ndev, 1)
    torch.cuda.set_device(a.device)   # torch's HIP runtime first, then the C-ABI library's; also under gloo: the
    torch.cuda.init()                 # all-reduce callback aliases engine buffers on THIS device
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
        os.environ.setdefault(k, v)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", a.device))
    else:
        dist.init_process_group(backend)
    try:
        learn(a, dist=dist, rank=rank, world=world, cdev=f"cuda:{a.device}" if backend == "nccl" else "cpu")
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
