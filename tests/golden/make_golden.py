#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference (oracle/_ref/libazr_ref.so).

Run in the build container only (needs /root/reference to have been compiled by `make -C oracle ref`):

    python tests/golden/make_golden.py

The fixtures are data only: inputs (seeds, state byte images, masks, priors) and the reference's
outputs for them.  Struct padding bytes of the 160-byte `Data` image are zeroed so the files are
deterministic.  What each file pins:

  tables.npz        a1/a6   adjacency lists (order!), neighbour masks, continent masks, bonuses
  rng_kat.npz       a11     minstd_rand0 + libstdc++ distributions: dice / int / float streams, randomMask
  rules_games.npz   a2-a10  seeded random-legal-policy games: state before each move, legal mask, move, outcome
  moves_all.npz     a5,a8,a9  every policy index 0..43 (legal AND illegal) applied to sampled states with a
                            fixed dice seed: error class (0 ok / 1 invalid_argument / 2 logic_error) + next state
  encode.npz        a12     NNInputData 88-byte images of sampled states
  normalize.npz     a15     NNOutputData::normalize on random priors x legal masks
  update_values.npz a24     z back-fill per record
  players_games.npz f-1,f-3 ScriptPlayer / RandomPlayer / Game (mirrored pairs, alternating starts): results, final states
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import azr_testlib as T  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    R = T.ref()
    fm = T.data_field_mask()

    # ---- tables
    nb = np.full((42, 6), 255, np.uint8)
    nbn = np.zeros(42, np.uint8)
    nbm = np.zeros(42, np.uint64)
    buf = (C.c_uint8 * 8)()
    for i in range(42):
        n = R.ref_neighbours(i, buf)
        nbn[i] = n
        nb[i, :n] = [buf[j] for j in range(n)]
        nbm[i] = R.ref_neighbour_mask(i)
    cont = np.array([R.ref_continent_mask(c) for c in range(7)], np.uint64)
    reinf_in = np.array([0, 1, 0x1ff, 0x1e00, 0x3f00000, 0xfe000, 0x3ffc000000, 0x3c000000000, 0x3ffffffffff,
                         0x1ff | 0x1e00, 0x155555, 0x3ffffffffff ^ 0x1ff, 0xfff, 0x7], np.uint64)
    rng = np.random.default_rng(7)
    reinf_in = np.concatenate([reinf_in, rng.integers(0, 1 << 42, 200, dtype=np.uint64)])
    reinf_out = np.array([R.ref_reinforcement_value(int(x)) for x in reinf_in], np.int8)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), neighbours=nb, neighbour_count=nbn, neighbour_mask=nbm,
                        continent_mask=cont, reinf_in=reinf_in, reinf_out=reinf_out)

    # ---- rng known answers
    seeds = np.array([1, 1234, 20260001, 2147483646, 2147483647, 0, 4294967295], np.uint32)
    dice = np.zeros((len(seeds), 600), np.uint8)
    ints = np.zeros((len(seeds), 200), np.int64)
    flts = np.zeros((len(seeds), 200), np.float32)
    mixed_state = np.zeros(len(seeds), np.uint32)
    for k, s in enumerate(seeds):
        R.ref_seed(int(s))
        dice[k] = [R.ref_rng_dice() for _ in range(600)]
        R.ref_seed(int(s))
        ints[k] = [R.ref_rng_int() for _ in range(200)]
        R.ref_seed(int(s))
        flts[k] = [R.ref_rng_float() for _ in range(200)]
        R.ref_seed(int(s))
        for _ in range(10):
            R.ref_rng_dice(); R.ref_rng_int(); R.ref_rng_float()
        mixed_state[k] = R.ref_rng_state()
    rm_masks = rng.integers(1, 1 << 43, 300, dtype=np.uint64)
    R.ref_seed(99)
    rm_out = np.array([R.ref_random_mask(int(m)) for m in rm_masks], np.uint64)
    np.savez_compressed(os.path.join(HERE, "rng_kat.npz"), seeds=seeds, dice=dice, ints=ints, floats=flts,
                        mixed_state=mixed_state, rm_seed=np.uint32(99), rm_masks=rm_masks, rm_out=rm_out)

    # ---- seeded random-policy games
    game_seeds = np.array(list(range(1, 17)) + [1234, 20260001, 20260002, 20260003], np.uint32)
    states, masks, moves, starts, status, finals = [], [], [], [0], [], []
    for s in game_seeds:
        g = T.ref_random_game(int(s))
        st = g["states"].copy()
        st[:, ~fm] = 0
        states.append(st); masks.append(g["masks"]); moves.append(g["moves"])
        starts.append(starts[-1] + len(g["moves"]))
        status.append(g["status"])
        f = g["final"].copy(); f[~fm] = 0
        finals.append(f)
    np.savez_compressed(os.path.join(HERE, "rules_games.npz"), seeds=game_seeds,
                        states=np.concatenate(states), masks=np.concatenate(masks), moves=np.concatenate(moves),
                        starts=np.array(starts, np.int64), status=np.array(status, np.int8),
                        finals=np.stack(finals))
    allstates = np.concatenate(states)

    # ---- every move index on sampled states
    sel = allstates[::23]
    rc = np.zeros((len(sel), 44), np.uint8)
    nxt = np.zeros((len(sel), 44, 160), np.uint8)
    for i, st in enumerate(sel):
        for mv in range(44):
            d = st.copy()
            R.ref_seed(777 + mv)
            rc[i, mv] = R.ref_make_move(T.ptr(d), mv)
            if rc[i, mv] == 0:
                d[~fm] = 0
                nxt[i, mv] = d
    np.savez_compressed(os.path.join(HERE, "moves_all.npz"), states=sel, dice_seed_base=np.uint32(777), rc=rc,
                        next=nxt)

    # ---- encode
    sel = allstates[::5]
    enc = np.zeros((len(sel), 88), np.uint8)
    gstat = np.zeros(len(sel), np.int8)
    for i, st in enumerate(sel):
        R.ref_encode(T.ptr(st), T.ptr(enc[i]))
        gstat[i] = R.ref_game_status(T.ptr(st))
    np.savez_compressed(os.path.join(HERE, "encode.npz"), states=sel, in88=enc, status=gstat)

    # ---- normalize
    pri = rng.uniform(0.0, 1.0, (400, 43)).astype(np.float32)
    pri /= pri.sum(1, keepdims=True)
    vm = np.concatenate([np.concatenate(masks)[::17][:300], rng.integers(1, 1 << 43, 100, dtype=np.uint64)])
    out = pri.copy()
    for i in range(len(out)):
        R.ref_normalize(T.ptr(out[i]), int(vm[i]))
    np.savez_compressed(os.path.join(HERE, "normalize.npz"), priors=pri, valid=vm, out=out)

    # ---- update values
    pl = rng.integers(0, 2, 64).astype(np.int8)
    zs = {}
    for gs in (0, 1, -2):
        z = np.zeros(64, np.float32)
        R.ref_update_values(T.ptr(pl), 64, gs, 80, T.ptr(z))
        zs[f"z_{gs}"] = z
    np.savez_compressed(os.path.join(HERE, "update_values.npz"), players=pl, z_p0=zs["z_0"], z_p1=zs["z_1"],
                        z_draw=zs["z_-2"])
    # ---- opponents + host game driver (ScriptPlayer / RandomPlayer / Game mirrored pairs): f-1, f-3
    rows = []
    for (k0, k1) in ((1, 2), (2, 1), (1, 1), (2, 2)):
        for mirror in (1, 0):
            for seed in (1, 2, 3):
                r6, st, rd, fin, rs = T.ref_play_games(k0, k1, 6, mirror, seed)
                fin = fin.copy(); fin[:, ~fm] = 0
                rows.append((k0, k1, mirror, seed, r6, st, rd, fin, rs))
    np.savez_compressed(os.path.join(HERE, "players_games.npz"),
                        kinds=np.array([[r[0], r[1]] for r in rows], np.int8), mirror=np.array([r[2] for r in rows], np.int8),
                        seeds=np.array([r[3] for r in rows], np.uint32), results=np.array([r[4] for r in rows], np.int32),
                        status=np.stack([r[5] for r in rows]), rounds=np.stack([r[6] for r in rows]),
                        finals=np.stack([r[7] for r in rows]), rng_state=np.array([r[8] for r in rows], np.uint32))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
