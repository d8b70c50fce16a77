"""End-to-end parity of the threaded drop-in (rows L1, M1, A1-A3, R1-R7 of SURVEY 8a together):
the `rela` module of this repo, driven exactly like the REAL reference was when
tests/golden/e2e_lockstep_apex.json was recorded (its own pybind module, its TorchScript
ApexAgent on the CPU, the same synthetic env source compiled against its rela/env.h)."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    sys.path.insert(0, os.path.join(ROOT, "rela_amd", "pybind"))
    import torch  # noqa: F401
    import rela
    import synth_atari

    assert "rela_amd/pybind" in rela.__file__
    return rela, synth_atari


def _check_apex_rounds(rounds, gold):
    assert len(rounds) == len(gold["expect"])
    for r, (got, exp) in enumerate(zip(rounds, gold["expect"])):
        for key in ("s_sum", "s_head", "next_s_sum", "s_planes", "next_s_planes", "a", "terminal", "bootstrap", "eps",
                    "legal_sum", "num_add"):
            assert got[key] == exp[key], (r, key)
        assert np.array_equal(np.float32(got["reward"]), np.float32(exp["reward"])), r  # n-step return: exact
        np.testing.assert_allclose(got["weight"], exp["weight"], rtol=1e-4, err_msg="IS weights, round %d" % r)


@pytest.fixture
def dedup_env():
    """RELA_REPLAY_DEDUP for the duration of one test (read when the replay partition is created)."""
    def set_(mode):
        if mode:
            os.environ["RELA_REPLAY_DEDUP"] = mode
            os.environ["RELA_REPLAY_DEDUP_GUARD"] = "512"
        else:
            os.environ.pop("RELA_REPLAY_DEDUP", None)
    yield set_
    os.environ.pop("RELA_REPLAY_DEDUP", None)
    os.environ.pop("RELA_REPLAY_DEDUP_GUARD", None)


@pytest.mark.parametrize("dedup", [None, "stack"])
def test_lockstep_matches_reference(mods, dedup_env, dedup):
    """dedup = "stack": the replay keeps every observation stack once (obs of one transition = next_obs of
    another, as the reference's tensor views share storage, rela/types.cc:48-67); the sampled batches must be the
    same rows the REAL reference returned (SURVEY 8f-3's parity definition)."""
    from e2e_lockstep import CFG, load_agent_params, run_lockstep
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_apex.json")))
    assert gold["cfg"] == CFG
    dedup_env(dedup)
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(CFG["num_action"]), CFG["multi_step"], CFG["gamma"]))
    _check_apex_rounds(run_lockstep(rela, synth, agent, "cuda:0", "cuda:0"), gold)


def test_lockstep_with_prefetch_matches_reference(mods, dedup_env):
    """FFPrioritizedReplay(..., prefetch = 2): update_priority queues the next sample behind itself on the replay's
    stream and sample() hands that batch over (rela/prioritized_replay.h:223-230 runs sampler futures next to the
    learner).  The library sees the same calls in the same order as with prefetch = 0, so every round must still equal
    what the REAL reference (prefetch 0) returned."""
    from e2e_lockstep import CFG, load_agent_params, run_lockstep
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_apex.json")))
    dedup_env(None)
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(CFG["num_action"]), CFG["multi_step"], CFG["gamma"]))
    _check_apex_rounds(run_lockstep(rela, synth, agent, "cuda:0", "cuda:0", prefetch=2), gold)


@pytest.mark.parametrize("dedup", [None, "stack", "plane"])
def test_lockstep_sliding_env_matches_reference(mods, dedup_env, dedup):
    """An Atari-like env (one new 84x84 plane per step, sliding stack, first frame of an episode repeated four
    times: atari/game_state.h:53-82) driven through the REAL reference gave tests/golden/e2e_lockstep_apex_sliding.json.
    With dedup = "plane" the replay stores 7,056 B per env-step instead of 56,448 B and rebuilds the stacks in its
    gather: every sampled frame stack (per-plane sums), action, reward, flag and weight must equal the reference's."""
    from e2e_lockstep import CFG_SLIDING as C, load_agent_params, run_lockstep
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_apex_sliding.json")))
    assert gold["cfg"] == C
    dedup_env(dedup)
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(C["num_action"]), C["multi_step"], C["gamma"]), C)
    _check_apex_rounds(run_lockstep(rela, synth, agent, "cuda:0", "cuda:0", C), gold)


@pytest.fixture
def precision_env():
    """RELA_PRECISION for the duration of one test (read when a ModelLocker creates its device nets)."""
    def set_(mode):
        os.environ["RELA_PRECISION"] = mode
    yield set_
    os.environ.pop("RELA_PRECISION", None)


FAST_KERNELS = {"conv_bf16s<Conv3F>", "fc_bf16s (split-K)"}
F32_KERNELS = {"conv1_bf16x3", "conv_mfma<Conv2> (f32)", "conv_mfma<Conv3> (f32)"}


@pytest.mark.parametrize("precision", ["f32", "bf16x2"])
def test_lockstep_k128_matches_reference_in_both_precision_modes(mods, dedup_env, precision_env, precision):
    """the configuration the headline runs -- the engine in bf16x2 mode WITH memoised Q tables at
    >= 128 rows -- pinned at ENGINE level.  One actor thread x 128 envs through rela.Context / BasicThreadLoop /
    DQNActor / FFPrioritizedReplay (alpha 0.6 / beta 0.4), driven exactly like the REAL reference was when
    tests/golden/e2e_lockstep_apex_k128.json was recorded (its C++ actor, its TorchScript ApexAgent on the CPU:
    rela/dqn_actor.h:153-203, pyrela/apex.py:30-78).  Integer fields, frames, n-step rewards: exact; IS weights (the
    TD priorities through pow and the scan): 1e-4 relative in f32 mode, 2e-3 in bf16x2 mode (priorities within
    6e-5 x max|Q| of the f32 ones, tests/test_ffnet_gpu.py).  Exact greedy actions are a CHECKED expectation: the
    golden records that the two best legal Q-values of every decision of the run are 1.39e-4 apart, 8 x the fast
    mode's |dQ| bound.  The launch census asserts which kernels really ran."""
    from e2e_lockstep import CFG_BIG as C, load_agent_params, run_lockstep
    from kernel_names import CONV12
    from rela_amd import _capi as capi
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_apex_k128.json")))
    assert gold["cfg"] == C
    tol_q = 2e-5 * gold["max_abs_q"]
    assert gold["min_top2_gap"] > 4 * tol_q, "the golden's decisions are too close for an exact action comparison"
    dedup_env(None)
    precision_env(precision)
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(C["num_action"]), C["multi_step"], C["gamma"]), C)
    with capi.launch_census() as census:
        rounds = run_lockstep(rela, synth, agent, "cuda:0", "cuda:0", C)
    ran = set(census.counts)
    if precision == "bf16x2":
        assert FAST_KERNELS | {CONV12} <= ran and not (F32_KERNELS & ran), census.counts
    else:
        assert F32_KERNELS <= ran and not ((FAST_KERNELS | {CONV12}) & ran), census.counts
    assert len(rounds) == len(gold["expect"])
    for r, (got, exp) in enumerate(zip(rounds, gold["expect"])):
        for key in ("s_sum", "s_head", "next_s_sum", "s_planes", "next_s_planes", "a", "terminal", "bootstrap", "eps",
                    "legal_sum", "num_add"):
            assert got[key] == exp[key], (precision, r, key)
        assert np.array_equal(np.float32(got["reward"]), np.float32(exp["reward"])), (precision, r)
        np.testing.assert_allclose(got["weight"], exp["weight"], rtol=1e-4 if precision == "f32" else 2e-3,
                                   err_msg="IS weights, round %d, %s" % (r, precision))


# kernels of the f32x3 mode (csrc/ffnet.hip: from 512 rows every dense layer of the trunk runs on the bf16 matrix cores with
# three-part operands) -- tests/test_ffnet_gpu.py expected_kernels is the per-layer statement
F32X3_KERNELS = {"conv12_s3", "conv3_img_s3", "gemm_s3<fc>"}


@pytest.mark.parametrize("precision", ["f32", "f32x3"])
def test_lockstep_k512_matches_reference_in_f32_and_f32x3(mods, dedup_env, precision_env, precision):
    """the HEADLINE arithmetic pinned at engine level to the REAL reference.  One actor thread x 512
    envs (one 512-row shard: the batch size from which every dense layer of the f32x3 mode runs its three-part bf16
    kernels) through rela.Context / BasicThreadLoop / DQNActor / FFPrioritizedReplay (alpha 0.6 / beta 0.4), driven
    exactly like the reference was when tests/golden/e2e_lockstep_apex_k512.json was recorded (its C++ actor, its
    TorchScript ApexAgent on the CPU: rela/dqn_actor.h:153-203, pyrela/apex.py:30-78, pyrela/net.py:42-53).  Frames,
    actions, flags, n-step rewards: exact; IS weights (TD priorities through pow and the scan): 1e-4 relative in BOTH
    modes -- f32x3 gets the f32 mode's tolerance.  Exact greedy actions are a checked expectation: the golden records
    the smallest top-two gap of the run (1e-4) against |dQ| of a few 1e-7.  The launch census asserts the kernels."""
    from e2e_lockstep import CFG_K512 as C, load_agent_params, run_lockstep
    from kernel_names import CONV12
    from rela_amd import _capi as capi
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_apex_k512.json")))
    assert gold["cfg"] == C
    assert gold["min_top2_gap"] > 20 * 1e-6 * max(1.0, gold["max_abs_q"]), "decisions too close for an exact action comparison"
    dedup_env(None)
    precision_env(precision)
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(C["num_action"]), C["multi_step"], C["gamma"]), C)
    with capi.launch_census() as census:
        rounds = run_lockstep(rela, synth, agent, "cuda:0", "cuda:0", C)
    ran = set(census.counts)
    if precision == "f32x3":
        assert F32X3_KERNELS <= ran and not ((FAST_KERNELS | {CONV12} | F32_KERNELS) & ran), census.counts
    else:
        assert F32_KERNELS <= ran and not ((FAST_KERNELS | {CONV12} | F32X3_KERNELS) & ran), census.counts
    assert len(rounds) == len(gold["expect"])
    for r, (got, exp) in enumerate(zip(rounds, gold["expect"])):
        for key in ("s_sum", "s_head", "next_s_sum", "s_planes", "next_s_planes", "a", "terminal", "bootstrap", "eps",
                    "legal_sum", "num_add"):
            assert got[key] == exp[key], (precision, r, key)
        assert np.array_equal(np.float32(got["reward"]), np.float32(exp["reward"])), (precision, r)
        np.testing.assert_allclose(got["weight"], exp["weight"], rtol=1e-4, err_msg="IS weights, round %d, %s" % (r, precision))


@pytest.mark.parametrize("plane_upload", ["1", "0"])
def test_cohort_plane_upload_matches_reference_sliding_golden(mods, dedup_env, plane_upload):
    """r4: a VectorEnv whose envs all declare a sliding frame stack (rela::FrameRowEnv) uploads only the NEWEST 84x84
    plane of every row and the actor shard completes the stacks on the device (rela_apex_actor_slide_stacks; restart
    flags on episode starts: atari/game_state.h:53-82).  The four envs of the sliding golden, driven as a cohort of
    2 threads x 2 envs, must reproduce what the REAL reference (one thread x four envs, whole frames through its own
    VectorEnv, rela/env.h:45-82) sampled: every frame stack (per-plane sums of s and next_s), action, n-step reward and
    flag exactly, IS weights to 1e-4.  RELA_PLANE_UPLOAD=0 runs the same cohort with whole-frame uploads; the launch
    census says which path ran."""
    from e2e_lockstep import CFG_SLIDING, CFG_SLIDING_COHORT as C, load_agent_params, run_lockstep
    from rela_amd import _capi as capi
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_apex_sliding.json")))
    assert gold["cfg"] == CFG_SLIDING
    dedup_env(None)
    os.environ["RELA_PLANE_UPLOAD"] = plane_upload
    try:
        agent = load_agent_params(ApexAgent(lambda: AtariFFNet(C["num_action"]), C["multi_step"], C["gamma"]), C)
        with capi.launch_census() as census:
            rounds = run_lockstep(rela, synth, agent, "cuda:0", "cuda:0", C)
    finally:
        os.environ.pop("RELA_PLANE_UPLOAD", None)
    assert ("slide_stacks" in census.counts) == (plane_upload == "1"), census.counts
    _check_apex_rounds(rounds, gold)


def test_two_lockers_in_one_process_give_two_partitions_with_reference_parity(mods, dedup_env):
    """The reference's single-process multi-device wiring (pyrela/main.py:131-136,155,166: one ModelLocker per act
    device, threads dealt round-robin, ONE replay object) with two lockers on cuda:0: the replay owns one partition per
    locker (capacity / 2, seeds 5 and 6) and a batch of 16 is 8 rows of each.  Partition 0 is fed by thread 0 -- the
    very envs, capacity, seed and per-round priorities of the single-replay golden recorded from the REAL reference --
    so rows 0..7 of every batch must be that golden's batch (SURVEY 8e's parity definition: each partition
    bit-identical to a reference PrioritizedReplay(capacity / G, seed_g) fed the same stream and asked for B / G), and
    their importance weights the golden's up to the common factor of the global normalisation."""
    from e2e_lockstep import CFG, CFG_TWO_LOCKERS as C, load_agent_params, run_lockstep
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_apex.json")))
    assert gold["cfg"] == CFG
    dedup_env(None)
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(C["num_action"]), C["multi_step"], C["gamma"]))
    rounds = run_lockstep(rela, synth, agent, "cuda:0", "cuda:0", C)
    assert len(rounds) == len(gold["expect"])
    for r, (got, exp) in enumerate(zip(rounds, gold["expect"])):
        for key in ("s_sum", "s_head", "next_s_sum", "a", "terminal", "bootstrap", "eps", "legal_sum"):
            assert got[key][:8] == exp[key], (r, key)
            assert len(got[key]) == 16
        assert got["num_add"] == 2 * exp["num_add"], r
        assert np.array_equal(np.float32(got["reward"][:8]), np.float32(exp["reward"])), r
        w = np.asarray(got["weight"])
        assert w.max() == 1.0 and (w > 0).all()
        np.testing.assert_allclose(w[:8] / w[:8].max(), np.asarray(exp["weight"]) / max(exp["weight"]), rtol=1e-4,
                                   err_msg="IS weights of partition 0, round %d" % r)


def test_reference_main_wiring_runs_in_one_process(mods, capsys):
    """pyrela/main.py's own control flow in ONE process: two act lockers (`--act_device
    cuda:0,cuda:0`: one ModelLocker per entry, threads dealt round-robin, main.py:131-136,155,166) feeding one replay
    object, the eval locker on "cpu" (main.py:116) and the eval cadence after every epoch (main.py:264-278) with
    evaluation actors on that "cpu" locker -- which run on the GPU in the f32 parity mode."""
    from rela_amd.pyrela import main as entry

    args = entry.parse_args(["--num_thread", "4", "--num_game_per_thread", "8", "--batchsize", "32", "--epoch_len", "20",
                             "--num_epoch", "2", "--burn_in_frames", "128", "--replay_buffer_size", "1024",
                             "--episode_len", "25", "--actor_sync_freq", "5", "--act_device", "cuda:0,cuda:0",
                             "--single_process", "1", "--num_eval_game", "2"])
    hist = entry.train(args)
    out = capsys.readouterr().out
    assert "Speed: train: " in out and "eval score:" in out
    assert len(hist) == 2 and all(np.isfinite(h["loss"]) and np.isfinite(h["eval_score"]) for h in hist)
    assert hist[-1]["act"] > 0 and hist[-1]["buffer_add"] > 0


def test_cohort_in_fast_mode_agrees_with_the_f32_cohort(mods, dedup_env, precision_env):
    """The ActorCohort (2 threads x 64 envs = ONE 128-row shard: every thread runs the reference loop on its 64 envs,
    the last arriver launches the batched act / post_step; q.min() and replay blocks per group of 64) in bf16x2 mode
    against the same cohort in f32 mode, in lock step: every sampled batch holds the same rows (frames), actions,
    n-step rewards and flags; IS weights within the fast mode's priority tolerance.  The envs are the K = 128 golden's
    (same seeds), so the decisions of the run are the ones whose gaps the golden recorded."""
    from e2e_lockstep import CFG_COHORT as C, load_agent_params, run_lockstep
    from kernel_names import CONV12
    from rela_amd import _capi as capi
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    dedup_env(None)
    runs = {}
    for precision in ("f32", "bf16x2"):
        precision_env(precision)
        agent = load_agent_params(ApexAgent(lambda: AtariFFNet(18), 3, 0.997), C)
        with capi.launch_census() as census:
            runs[precision] = run_lockstep(rela, synth, agent, "cuda:0", "cuda:0", C)
        ran = set(census.counts)
        if precision == "bf16x2":
            assert FAST_KERNELS | {CONV12} <= ran and not (F32_KERNELS & ran), census.counts
        else:
            assert F32_KERNELS <= ran and not ((FAST_KERNELS | {CONV12}) & ran), census.counts
    assert len(runs["f32"]) == len(runs["bf16x2"]) == C["rounds"]
    for r, (a, b) in enumerate(zip(runs["f32"], runs["bf16x2"])):
        for key in ("s_sum", "s_head", "next_s_sum", "a", "terminal", "bootstrap", "eps", "legal_sum", "num_add"):
            assert a[key] == b[key], (r, key)
        assert np.array_equal(np.float32(a["reward"]), np.float32(b["reward"])), r
        np.testing.assert_allclose(a["weight"], b["weight"], rtol=2e-3, err_msg="IS weights, round %d" % r)


def test_cohort_in_f32x3_mode_agrees_with_the_f32_cohort(mods, dedup_env, precision_env):
    """RELA_PRECISION=f32x3 end to end: a cohort of 4 threads x 128 envs (one 512-row shard: from there conv2 / conv3 of the
    actors' forwards run on the three-part bf16 kernels, csrc/gemm_f32emu.h -- asserted through the launch census)
    against the same cohort in the exact f32 mode, in lock step: every sampled batch holds the same rows, actions,
    n-step rewards and flags, and the IS weights (functions of the TD priorities) agree to 1e-5 -- f32 round-off, not
    the fast mode's 2e-3."""
    from e2e_lockstep import CFG_COHORT_512 as C, load_agent_params, run_lockstep
    from rela_amd import _capi as capi
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    dedup_env(None)
    emu = {"conv12_s3", "conv3_img_s3", "gemm_s3<fc>"}
    runs = {}
    for precision in ("f32", "f32x3"):
        precision_env(precision)
        agent = load_agent_params(ApexAgent(lambda: AtariFFNet(18), 3, 0.997), C)
        with capi.launch_census() as census:
            runs[precision] = run_lockstep(rela, synth, agent, "cuda:0", "cuda:0", C)
        ran = set(census.counts)
        f32_convs = F32_KERNELS - {"conv1_bf16x3"}  # (conv1 is the same exact kernel in both modes)
        if precision == "f32x3":
            assert emu <= ran and not (f32_convs & ran), census.counts
        else:
            assert F32_KERNELS <= ran and not (emu & ran), census.counts
    assert len(runs["f32"]) == len(runs["f32x3"]) == C["rounds"]
    for r, (a, b) in enumerate(zip(runs["f32"], runs["f32x3"])):
        for key in ("s_sum", "s_head", "next_s_sum", "a", "terminal", "bootstrap", "eps", "legal_sum", "num_add"):
            assert a[key] == b[key], (r, key)
        assert np.array_equal(np.float32(a["reward"]), np.float32(b["reward"])), r
        np.testing.assert_allclose(a["weight"], b["weight"], rtol=1e-5, err_msg="IS weights, round %d" % r)


@pytest.mark.parametrize("hip_learner", [1, 0])
def test_training_entry_point_runs(mods, capsys, hip_learner):
    """pyrela-style main loop on 2 threads x 8 envs for two tiny epochs: actors insert from C++
    threads while the learner samples / steps / updates (hand-written HIP learner step, or PyTorch
    autograd as in the reference); rates are printed in the reference's `Speed:` format and the
    loss is finite."""
    from rela_amd.pyrela import main as entry

    args = entry.parse_args(["--num_thread", "2", "--num_game_per_thread", "8", "--batchsize", "32", "--epoch_len", "20",
                             "--num_epoch", "2", "--burn_in_frames", "64", "--replay_buffer_size", "512",
                             "--episode_len", "25", "--actor_sync_freq", "5", "--hip_learner", str(hip_learner)])
    hist = entry.train(args)
    out = capsys.readouterr().out
    assert "Speed: train: " in out and "buffer_add: " in out
    assert len(hist) == 2 and all(np.isfinite(h["loss"]) for h in hist)
    assert hist[-1]["act"] > 0 and hist[-1]["buffer_add"] > 0


def test_lockstep_r2d2_matches_reference(mods):
    """R2D2 rows (T2, A4, A5, N2 of SURVEY 8a) end to end: R2D2Actor + RNNPrioritizedReplay of this
    repo against the REAL reference (H6-shimmed build, CPU TorchScript LSTM agent), same lock-step
    protocol.  Sequences, actions, n-step rewards, terminals, lengths exact; stored recurrent state
    to 1e-4; IS weights (functions of the aggregated TD priorities) to 1e-3."""
    from e2e_lockstep import CFG_R2D2, load_lstm_agent_params, run_lockstep_r2d2
    from rela_amd.pyrela.net import AtariLSTMNet
    from rela_amd.pyrela.r2d2 import R2D2Agent

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_r2d2.json")))
    C = CFG_R2D2
    assert gold["cfg"] == C
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, C["num_action"]), "cpu", C["multi_step"], C["gamma"], C["eta"],
                      C["seq_len"], C["burn_in"], 0)
    load_lstm_agent_params(agent)
    rounds = run_lockstep_r2d2(rela, synth, agent, "cuda:0", "cuda:0")
    assert len(rounds) == len(gold["expect"])
    for r, (got, exp) in enumerate(zip(rounds, gold["expect"])):
        for key in ("s_sum", "a", "terminal", "bootstrap", "legal_sum", "seq_len", "num_add", "size", "eps_sum"):
            assert got[key] == exp[key], (r, key)
        assert np.array_equal(np.float32(got["reward"]), np.float32(exp["reward"])), r
        np.testing.assert_allclose(got["h0_abs"], exp["h0_abs"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got["c0_abs"], exp["c0_abs"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got["weight"], exp["weight"], rtol=1e-3, err_msg="IS weights, round %d" % r)


def test_lockstep_r2d2_c4_shape_matches_reference(mods):
    """The same lock-step protocol at BASELINE config C4's window shape (seq 80 / burn-in 40 / n 3, 123 slots,
    3.47 MB per sequence; pyrela/scripts/ref_run_r2d2.sh:12-18) against the REAL reference: episodes shorter
    than the window, a terminal inside the carried region, consecutive carries, recurrent state captured at
    window index 80, and the time-major gather of [123, B, 4, 84, 84] batches."""
    from e2e_lockstep import CFG_R2D2_C4 as C, load_lstm_agent_params, run_lockstep_r2d2
    from rela_amd.pyrela.net import AtariLSTMNet
    from rela_amd.pyrela.r2d2 import R2D2Agent

    rela, synth = mods
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_lockstep_r2d2_c4.json")))
    assert gold["cfg"] == C
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, C["num_action"]), "cpu", C["multi_step"], C["gamma"], C["eta"],
                      C["seq_len"], C["burn_in"], 0)
    load_lstm_agent_params(agent, C)
    rounds = run_lockstep_r2d2(rela, synth, agent, "cuda:0", "cuda:0", C, quiet=3.0)
    assert len(rounds) == len(gold["expect"])
    lens = set()
    for r, (got, exp) in enumerate(zip(rounds, gold["expect"])):
        for key in ("s_sum", "a", "terminal", "bootstrap", "legal_sum", "seq_len", "num_add", "size", "eps_sum"):
            assert got[key] == exp[key], (r, key)
        assert np.array_equal(np.float32(got["reward"]), np.float32(exp["reward"])), r
        np.testing.assert_allclose(got["h0_abs"], exp["h0_abs"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(got["c0_abs"], exp["c0_abs"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(got["weight"], exp["weight"], rtol=2e-3, err_msg="IS weights, round %d" % r)
        lens.update(exp["seq_len"])
    assert max(lens) == C["burn_in"] + C["seq_len"] and min(lens) < max(lens)  # full and short sequences were sampled


@pytest.mark.parametrize("hip_learner", [1, 0])
def test_r2d2_training_entry_point_runs(mods, capsys, hip_learner):
    """--algo r2d2 on 2 threads x 4 envs: sequences (seq 8 / burn 4 / n 3) flow from C++ actor threads
    through RNNPrioritizedReplay into the R2D2 learner step (burn-in unroll, BPTT, Adam, aggregate priority):
    hand-written HIP (csrc/learner_r2d2.hip) or PyTorch autograd as in the reference."""
    from rela_amd.pyrela import main as entry

    # (epochs long enough for sequences to complete inside them: with the persistent recurrent kernels six learner
    # steps take a few milliseconds, less than one 8-step sequence of these 8 envs)
    args = entry.parse_args(["--algo", "r2d2", "--num_thread", "2", "--num_game_per_thread", "4", "--batchsize", "8",
                             "--epoch_len", "40", "--num_epoch", "2", "--burn_in_frames", "16",
                             "--replay_buffer_size", "64", "--episode_len", "30", "--actor_sync_freq", "3",
                             "--seq_len", "8", "--seq_burn_in", "4", "--priority_exponent", "0.9",
                             "--importance_exponent", "0.6", "--hip_learner", str(hip_learner)])
    hist = entry.train(args)
    out = capsys.readouterr().out
    assert "Speed: train: " in out
    assert len(hist) == 2 and all(np.isfinite(h["loss"]) for h in hist)
    assert hist[-1]["act"] > 0 and max(h["buffer_add"] for h in hist) > 0


def test_cohort_batches_threads_consistently(mods):
    """Two actor threads batched into one device shard (ActorCohort): every sampled transition must be
    self-consistent -- with eps = 0 its action is the greedy action of the online net on its own
    frame stack, its frames come from one of the envs' LCG streams, inserts arrive as whole rounds of
    T*K transitions -- and the learner-side TD error of the batch reproduces a finite priority."""
    import torch

    from e2e_lockstep import CFG, load_agent_params
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    T, K, A, n = 2, 4, CFG["num_action"], 3
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(A), n, 0.997)).to("cuda:0")
    replay = rela.FFPrioritizedReplay(64, 3, 1.0, 0.4, 0)
    locker = rela.ModelLocker([agent], "cuda:0")
    ctx = rela.Context()
    actors, games = [], []
    for t in range(T):
        vec = rela.VectorEnv()
        for g in range(K):
            game = synth.SyntheticAtariEnv(100 + t * K + g, 0.0, A, 11)
            games.append(game)
            vec.append(game)
        actor = rela.DQNActor(locker, n, K, 0.997, replay)
        actors.append(actor)
        ctx.push_env_thread(rela.BasicThreadLoop(actor, vec, False))
    ctx.start()
    import time

    t0 = time.time()
    while replay.size() < 80 and time.time() - t0 < 120:  # ring = 80: the cohort parks on the full ring
        time.sleep(0.01)
    assert replay.size() == 80
    assert replay.num_add() % (T * K) == 0  # whole rounds only
    assert actors[0].num_act() == actors[1].num_act() > 0  # lock-step
    for _ in range(3):
        batch, w = replay.sample(16, "cuda:0")
        with torch.no_grad():
            q = agent.online_net(batch.obs)
            greedy = ((1 + q - q.min()) * batch.obs["legal_move"]).argmax(1)
            _, prio = agent.loss(batch, sync_priority=False)
        assert torch.equal(greedy, batch.action["a"])
        assert torch.isfinite(prio).all() and torch.isfinite(w).all()
        replay.update_priority(prio)
        time.sleep(0.2)
    ctx.terminate()
    ctx.resume()
    t0 = time.time()
    while not ctx.terminated():
        if replay.size() >= 16:
            batch, w = replay.sample(16, "cuda:0")
            replay.update_priority(w)
        time.sleep(0.005)
        assert time.time() - t0 < 120


def test_eval_path_runs_one_episode_per_thread(mods):
    """BasicThreadLoop(eval=True) semantics (thread_loop.h:66-71,92-103): exactly one env, one episode,
    no replay traffic; evaluation actors act greedily (eps = 0) and the context terminates by itself."""
    from e2e_lockstep import CFG, load_agent_params
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.eval import evaluate
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(CFG["num_action"]), 3, 0.997))
    locker = rela.ModelLocker([agent], "cuda:0")
    score = evaluate(3, locker, rela.DQNActor, 77, 25)
    assert -25.0 <= score <= 25.0
    again = evaluate(3, locker, rela.DQNActor, 77, 25)
    assert again == score  # greedy + seeded envs: deterministic
    with pytest.raises(ValueError):
        v = rela.VectorEnv()
        v.append(synth.SyntheticAtariEnv(1, 0.0, 18, 5))
        v.append(synth.SyntheticAtariEnv(2, 0.0, 18, 5))
        rela.BasicThreadLoop(rela.DQNActor(locker), v, True)  # eval loops drive exactly one env


def test_r2d2_cohort_batches_threads_consistently(mods):
    """Two R2D2Actor threads batched into one device shard (ActorCohort, LSTM flavour): the threads run
    in lock-step, and every sampled sequence is self-consistent -- unrolling the PyTorch AtariLSTMNet
    from the stored h0 over the stored frames reproduces the stored (greedy, eps = 0) actions at every
    valid step, i.e. each env kept its own recurrent state and window inside the shared shard."""
    import time

    import torch

    from e2e_lockstep import CFG_R2D2, load_lstm_agent_params
    from rela_amd.pyrela.net import AtariLSTMNet
    from rela_amd.pyrela.r2d2 import R2D2Agent

    rela, synth = mods
    C = CFG_R2D2
    T, K, A = 2, 4, C["num_action"]
    seq_len, burn, n = 8, 4, 3
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, A), "cpu", n, C["gamma"], C["eta"], seq_len, burn, 0)
    load_lstm_agent_params(agent)
    gpu_agent = R2D2Agent.clone(agent, "cuda:0")
    replay = rela.RNNPrioritizedReplay(32, 3, 0.9, 0.6, 0)
    locker = rela.ModelLocker([agent], "cuda:0")
    ctx = rela.Context()
    actors = []
    for t in range(T):
        vec = rela.VectorEnv()
        for g in range(K):
            vec.append(synth.SyntheticAtariEnv(300 + t * K + g, 0.0, A, 23))
        actor = rela.R2D2Actor(locker, n, K, C["gamma"], seq_len, burn, replay)
        actors.append(actor)
        ctx.push_env_thread(rela.BasicThreadLoop(actor, vec, False))
    ctx.start()
    t0 = time.time()
    while replay.size() < 24 and time.time() - t0 < 120:
        time.sleep(0.01)
    assert replay.size() >= 24
    assert actors[0].num_act() == actors[1].num_act() > 0  # lock-step: one shard, one tick for both threads
    B = 6
    for _ in range(3):
        batch, w = replay.sample(B, "cuda:0")
        s = batch.obs["s"]                      # [T, B, 4, 84, 84]
        legal = batch.obs["legal_move"]         # [T, B, A]
        with torch.no_grad():
            o, _ = gpu_agent.online_net.unroll_rnn({"s": s}, {"h0": batch.h0["h0"], "c0": batch.h0["c0"]})
            adv = gpu_agent.online_net.fc_a(o)
            greedy = ((1 + adv - adv.min()) * legal).argmax(2)
        a = batch.action["a"]
        L = batch.seq_len.long()
        assert (L >= 1).all() and (L <= burn + seq_len).all()
        for b in range(B):
            lb = int(L[b])
            assert torch.equal(greedy[:lb, b].cpu(), a[:lb, b].cpu()), "sequence %d deviates from its own unroll" % b
        assert torch.isfinite(w).all()
        replay.update_priority(torch.ones(B))
        time.sleep(0.1)
    ctx.terminate()
    ctx.resume()
    t0 = time.time()
    while not ctx.terminated():
        if replay.size() >= B:
            batch, w = replay.sample(B, "cuda:0")
            replay.update_priority(torch.ones(B))
        time.sleep(0.005)
        assert time.time() - t0 < 120


def test_r2d2_eval_path_runs_one_episode_per_thread(mods):
    """The evaluation constructor R2D2Actor(locker) (r2d2_actor.h:208-215): batch 1, no replay, recurrent
    state carried over the episode; scores are deterministic at eps = 0."""
    import torch

    from rela_amd.pyrela import create_env
    from rela_amd.pyrela.eval import evaluate
    from rela_amd.pyrela.net import AtariLSTMNet
    from rela_amd.pyrela.r2d2 import R2D2Agent
    from synth import synth_lstm_params

    rela, synth = mods
    A = create_env.get_num_action("synthetic")  # the eval env factory's action count
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, A), "cpu", 3, 0.997, 0.9, 8, 4, 0)
    sd = {}
    for prefix, seed in (("online_net.", 31), ("target_net.", 32)):
        for k, v in synth_lstm_params(A, seed).items():
            sd[prefix + k] = torch.from_numpy(v)
    agent.load_state_dict(sd)
    locker = rela.ModelLocker([agent], "cuda:0")
    score = evaluate(3, locker, rela.R2D2Actor, seed=9, episode_len=17, eval_eps=0.0)
    assert -17.0 <= score <= 17.0
    assert evaluate(3, locker, rela.R2D2Actor, seed=9, episode_len=17, eval_eps=0.0) == score
    ev = rela.R2D2Actor(locker)
    assert ev.num_act() == 0


def _run_small_replay(rela, ctx, replay, actors, batch, target_adds, make_priority):
    """Samples / updates until `target_adds` insertions happened; fails instead of hanging."""
    import time

    ctx.start()
    t0 = time.time()
    while replay.num_add() < target_adds:
        if replay.size() >= batch:
            _, w = replay.sample(batch, "cuda:0")
            replay.update_priority(make_priority(w))
        time.sleep(0.002)
        assert time.time() - t0 < 120, "no progress: %d adds, size %d" % (replay.num_add(), replay.size())
    lockstep = len({a.num_act() for a in actors}) == 1
    ctx.terminate()
    ctx.resume()
    t0 = time.time()
    while not ctx.terminated():
        if replay.size() >= batch:
            _, w = replay.sample(batch, "cuda:0")
            replay.update_priority(make_priority(w))
        time.sleep(0.002)
        assert time.time() - t0 < 120
    return lockstep


def test_cohort_shard_larger_than_ring_slack_inserts_in_pieces(mods):
    """A batched shard of T*K = 8 rows on a replay whose ring leaves only ring - capacity = 4 free slots
    once full: one blocking append of 8 rows could never be satisfied (sampling evicts down to capacity
    only); the shard must insert K-group pieces, as the reference's separate threads would."""
    import torch

    from e2e_lockstep import CFG, load_agent_params
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    rela, synth = mods
    T, K, A, n = 2, 4, CFG["num_action"], 3
    agent = load_agent_params(ApexAgent(lambda: AtariFFNet(A), n, 0.997)).to("cuda:0")
    replay = rela.FFPrioritizedReplay(16, 3, 1.0, 0.4, 0)  # ring 20
    locker = rela.ModelLocker([agent], "cuda:0")
    ctx = rela.Context()
    actors = []
    for t in range(T):
        vec = rela.VectorEnv()
        for g in range(K):
            vec.append(synth.SyntheticAtariEnv(500 + t * K + g, 0.0, A, 11))
        actor = rela.DQNActor(locker, n, K, 0.997, replay)
        actors.append(actor)
        ctx.push_env_thread(rela.BasicThreadLoop(actor, vec, False))
    _run_small_replay(rela, ctx, replay, actors, 4, 240, lambda w: torch.ones_like(w))
    assert replay.num_add() >= 240 and replay.num_add() % K == 0


def test_r2d2_cohort_pop_larger_than_ring_slack_inserts_in_pieces(mods):
    """The same for sequences: 8 envs in one shard, replay capacity 8 (ring 10, 2 free slots once full);
    pops of up to 8 sequences go in pieces cut at env boundaries."""
    import torch

    from e2e_lockstep import CFG_R2D2, load_lstm_agent_params
    from rela_amd.pyrela.net import AtariLSTMNet
    from rela_amd.pyrela.r2d2 import R2D2Agent

    rela, synth = mods
    C = CFG_R2D2
    T, K, A = 2, 4, C["num_action"]
    seq_len, burn, n = 6, 2, 3
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, A), "cpu", n, C["gamma"], C["eta"], seq_len, burn, 0)
    load_lstm_agent_params(agent)
    replay = rela.RNNPrioritizedReplay(8, 3, 0.9, 0.6, 0)  # ring 10
    locker = rela.ModelLocker([agent], "cuda:0")
    ctx = rela.Context()
    actors = []
    for t in range(T):
        vec = rela.VectorEnv()
        for g in range(K):
            vec.append(synth.SyntheticAtariEnv(600 + t * K + g, 0.0, A, 9 + g))  # staggered episode ends
        actor = rela.R2D2Actor(locker, n, K, C["gamma"], seq_len, burn, replay)
        actors.append(actor)
        ctx.push_env_thread(rela.BasicThreadLoop(actor, vec, False))
    _run_small_replay(rela, ctx, replay, actors, 2, 120, lambda w: torch.ones_like(w).cpu())
    assert replay.num_add() >= 120


@pytest.mark.parametrize("algo", ["apex", "r2d2"])
def test_benchmark_driver_tiny_grid(mods, capsys, algo):
    """rela_amd/pyrela/benchmark.py (counterpart of pyrela/benchmark.py:19-126, the script that DEFINES the
    env-steps/s metric) on a tiny grid: both phases run, the per-epoch lines and the summary
    line have the reference's format, and actors keep acting next to the unthrottled sampler."""
    import re

    from rela_amd.pyrela import benchmark

    # the replay must not fill during the phase WITHOUT a sampler (back-pressure would park the actors, H10)
    argv = ["--grid", "2x4,3x2", "--epoch_sec", "0.5", "--num_epoch", "2", "--episode_len", "30"]
    if algo == "r2d2":
        argv += ["--algo", "r2d2", "--seq_len", "8", "--seq_burn_in", "4", "--replay_buffer_size", "8192",
                 "--burn_in_frames", "70"]
    else:
        argv += ["--replay_buffer_size", "65536", "--burn_in_frames", "600"]
    rows = benchmark.main(argv)
    out = capsys.readouterr().out
    assert [(r[0], r[1]) for r in rows] == [(2, 4), (3, 2)]
    per_epoch = re.findall(r"^(without|with) sample: epoch (\d+), act rate: (\d+), buffer size: (\d+)", out, re.M)
    assert len(per_epoch) == 2 * 2 * 2  # 2 cells x 2 phases x 2 epochs
    summary = re.findall(r"^act rate: without sample: ([0-9.]+), with sample: ([0-9.]+)$", out, re.M)
    assert len(summary) == 2
    for (t, k, without, with_), (sw, sws) in zip(rows, summary):
        assert abs(float(sw) - without) < 0.01 and abs(float(sws) - with_) < 0.01
        assert without > 0 and with_ > 0


@pytest.mark.parametrize("algo", ["apex", "r2d2"])
def test_multi_process_layout_rehearsal_on_one_gpu(algo):
    """(apex = BASELINE C3's layout, r2d2 = C4's: sequence replay partitions, LSTM nets, HIP R2D2 learner.)
    The reference's multi-GPU layout (pyrela/main.py:131-166: act devices next to one learner device) as
    one process per GPU (rela_amd/pyrela/main.py:train_multi, rela_amd/parallel.py), rehearsed with all three
    ranks on cuda:0 over gloo: two actor processes with their replay partitions and C++ actor threads, one
    learner process sampling B/G from each, gathering the rows, scattering the priorities and publishing the
    flat weights.  (On a multi-GPU node the same code runs over RCCL; that needs more than this one card.)"""
    import ast
    import subprocess

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "rela_amd", "pyrela", "main.py"), "--train_device", "cuda:0",
           "--act_device", "cuda:0,cuda:0", "--num_thread", "4", "--num_game_per_thread", "4", "--batchsize", "32",
           "--epoch_len", "10", "--num_epoch", "2", "--burn_in_frames", "64", "--replay_buffer_size", "1024",
           "--episode_len", "25", "--actor_sync_freq", "5"]
    if algo == "r2d2":
        cmd += ["--algo", "r2d2", "--seq_len", "10", "--seq_burn_in", "4", "--batchsize", "8", "--replay_buffer_size",
                "256", "--burn_in_frames", "16", "--epoch_len", "6"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    last = [l for l in out.stdout.splitlines() if l.startswith("{'history'")][-1]
    res = ast.literal_eval(last)
    assert len(res["history"]) == 2 and all(np.isfinite(h["loss"]) for h in res["history"])
    assert res["act"] > 0 and res["buffer_add"] > 0


def test_rccl_collectives_single_rank():
    """The collectives of the N > 1 paths (flat gradient all-reduce of HipApexLearner, the IS-weight normalisation,
    the flat weight broadcast) on a one-rank RCCL communicator, on the device buffers those paths use: RCCL itself
    runs on this GPU (more ranks need more GPUs; the multi-rank logic is covered over gloo)."""
    import socket
    import subprocess

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                       "rccl_single_rank_child.py"), str(port)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["backend"] == "nccl" and rec["same_grad"] and rec["bcast_same"] and rec["isw_err"] < 1e-6, rec
