"""Builds T threads x K synthetic envs wired to actors (counterpart of pyrela/create_atari.py:64-98).

Real ALE is unreachable in this pipeline (no submodule, no ROMs), so `game` only selects the
action count; everything else -- per-env seed = seed + thread*K + game (create_atari.py:84),
per-env eps from the Ape-X schedule, VectorEnv -> BasicThreadLoop -> Context -- is as upstream.
"""
import os
import sys

PYBIND_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pybind")
if PYBIND_DIR not in sys.path:
    sys.path.insert(0, PYBIND_DIR)

import torch  # noqa: E402,F401  (must precede the native modules: one HIP runtime per process)
import rela  # noqa: E402
import synth_atari  # noqa: E402

NUM_ACTION = 18  # ALE's legal action set size, what atari_env.h:78-80 reports for every game


def get_num_action(game_name):
    return NUM_ACTION


def create_game(seed, eps, episode_len):
    # RELA_SYNTH_ENV=null: the zero-cost env (constant frames, reward 0): the runtime's own ceiling
    if os.environ.get("RELA_SYNTH_ENV") == "null":
        return synth_atari.NullAtariEnv(eps, NUM_ACTION, episode_len, seed)
    # RELA_SYNTH_SLIDING=1: Atari-like frame stacks (ONE new 84x84 plane per step, the first plane of an episode
    # repeated four times: atari/game_state.h:53-82) instead of four fresh planes per step -- a quarter of the host
    # work per env-step, and the stacks a de-duplicating replay (RELA_REPLAY_DEDUP=plane) expects
    sliding = os.environ.get("RELA_SYNTH_SLIDING", "0") == "1"
    if os.environ.get("RELA_REPLAY_DEDUP") == "plane" and not sliding:
        # plane mode stores ONE new plane per env-step and rebuilds a stack from the previous step's planes: with an
        # env that emits four fresh planes per step it would hand the learner stacks that never existed
        raise ValueError("RELA_REPLAY_DEDUP=plane needs an env whose frame stack slides by one plane per step: set "
                         "RELA_SYNTH_SLIDING=1 (or use RELA_REPLAY_DEDUP=stack, which is valid for any env)")
    return synth_atari.SyntheticAtariEnv(seed, eps, NUM_ACTION, episode_len, sliding)


def create_train_env(seed, eps, episode_len, num_thread, num_game_per_thread, actor_creator):
    context = rela.Context()
    games, actors = [], []
    for t in range(num_thread):
        vec = rela.VectorEnv()
        for g in range(num_game_per_thread):
            idx = t * num_game_per_thread + g
            game = create_game(seed + idx, eps[idx], episode_len)
            games.append(game)
            vec.append(game)
        actor = actor_creator(t)
        actors.append(actor)
        context.push_env_thread(rela.BasicThreadLoop(actor, vec, False))
    print("Finished creating environments with %d games" % len(games))
    return context, games, actors
