"""TEST INFRASTRUCTURE: the new-vs-old arena stepped from the HOST through the batched Player seam of the binding
(AlphaZeroPlayerGroup::takeTurns over azr_engine_* / azr_mcts_*), kept as a cross-check of the device-resident arena
(azr_arena_*), which is the product path.  Not shipped in the package."""
import numpy as np


def invert_players(img):
    """State::invertPlayers (state.cpp:493-516) on [G,160] Data images"""
    out = img.copy()
    la = out[:, :42]
    owner = la >> 6
    out[:, :42] = np.where(owner < 2, (la & 63) | ((owner ^ 1) << 6), la)
    out[:, 48:96], out[:, 96:144] = img[:, 96:144], img[:, 48:96]
    return out


def take_turns(eng, states, me):
    """AlphaZeroPlayerGroup::takeTurns = batched AlphaZeroPlayer::takeTurn (alphazero_player.cpp:3-21)"""
    eng.mcts_trim()
    while True:
        eng.set_states(states)
        status = eng.status()
        mine = (status == -1) & (states[:, 146] == me)
        if not mine.any():
            return states
        eng.simulate()
        mv = eng.pick(sample=False)
        mv[~mine] = 255
        eng.make_moves(mv)
        states = eng.get_states()


def arena_two_nets(eng_new, eng_old, games, mirror=True, base_seed=1):
    """GameGroup::playGames(trainAZPG, generateAZPG, games) (game.cpp:277-312): player 0 = new net, player 1 = old net.
    The G slots play mirrored pairs in lock-step.  Returns a GameResults-like dict."""
    G = eng_new.G
    res = dict(count=0, draw=0, win=[0, 0], win_and_started=[0, 0])
    seed = base_seed
    while True:
        take = min(G, (games - res["count"]) // 2)   # Counter::hasNext(2): whole pairs only; slot k plays pair k
        if take == 0:
            return res
        eng_new.new_games(np.arange(seed, seed + G, dtype=np.uint32))
        seed += G
        start = eng_new.get_states()
        for player_start in (0, 1):   # Game::newGame (game.cpp:170-191) + incPlayerStart
            if player_start == 0:
                states = start.copy()
            elif mirror:
                states = invert_players(start)
            else:
                eng_new.new_games(np.arange(seed, seed + G, dtype=np.uint32))
                seed += G
                states = eng_new.get_states()
            states[:, 146] = player_start          # State::setCurrentPlayerTurn
            eng_new.mcts_clear()                   # AlphaZeroPlayer::newGame
            eng_old.mcts_clear()
            while True:                            # Game::gameLoop
                states = take_turns(eng_new, states, 0)
                states = take_turns(eng_old, states, 1)
                eng_new.set_states(states)
                status = eng_new.status()
                if (status != -1).all():
                    break
            for g in range(take):                  # GameResults::addGame (game.cpp:193-213)
                res["count"] += 1
                if status[g] == -2:
                    res["draw"] += 1
                else:
                    res["win"][status[g]] += 1
                    if status[g] == player_start:
                        res["win_and_started"][status[g]] += 1


