"""Frame-stack de-duplication in the replay (the reference stores transitions as VIEWS so that obs
of one and next_obs of another share storage, rela/types.cc:48-67, and an Atari observation is a sliding stack
of four planes, atari/game_state.h:53-82).  Parity definition: with de-duplication on, every sampled batch
(frames, small fields, ids, importance weights, f64 running sum) is IDENTICAL to the batch of a replay that stores
s and next_s in full, fed by the same actor shard inputs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sliding_frames(rng, R, ticks, p_term):
    """Atari-like observation stream: one new plane per step, the stack slides; after a terminal step the next
    observation starts an episode with its first plane repeated four times (game_state.h:66-70)."""
    planes = rng.integers(0, 256, (ticks, R, 84, 84), dtype=np.uint8)
    planes[:, :, 0, 0] = (np.arange(ticks)[:, None] % 251).astype(np.uint8)  # a tag byte: tick
    planes[:, :, 0, 1] = (np.arange(R)[None, :] % 251).astype(np.uint8)      # ... and row
    term = (rng.uniform(size=(ticks, R)) < p_term).astype(np.uint8)
    stacks = np.zeros((ticks, R, 4, 84, 84), np.uint8)
    for t in range(ticks):
        for r in range(R):
            if t == 0 or term[t - 1, r]:
                stacks[t, r, :] = planes[t, r]
            else:
                stacks[t, r, :3] = stacks[t - 1, r, 1:]
                stacks[t, r, 3] = planes[t, r]
    return stacks, term


def _run(dedup, stacks, term, rewards, cap, batch, n, K, sample_every, nonblocking=False):
    import torch

    from rela_amd.engine import ApexActorEngine, FFNetHandle
    from rela_amd.replay import FFReplay
    from synth import synth_params

    ticks, R = stacks.shape[:2]
    A = 6
    dev = "cuda:0"
    on, tg = FFNetHandle(A, dev), FFNetHandle(A, dev)
    on.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 11).items()})
    tg.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 12).items()})
    replay = FFReplay(cap, 7, 0.6, 0.4, 0, A, dev, dedup=dedup, guard_units=(n + 8) * R)
    eng = ApexActorEngine(R, K, A, n, 0.997, replay, [0.0] * R, dev)
    out = []
    for t in range(ticks):
        eng.next_obs_slot().copy_(torch.from_numpy(stacks[t]))
        eng.act(on)
        eng.post_step(torch.from_numpy(rewards[t]).to(dev), torch.from_numpy(term[t]).to(dev), on, tg,
                      nonblocking=nonblocking)
        if t % sample_every == sample_every - 1 and replay.size() >= batch:
            b, w = replay.sample(batch)
            st = replay.debug_state()
            assert st["dev_error"] == 0
            out.append(dict(s=b.obs["s"].cpu().numpy().copy(), ns=b.next_obs["s"].cpu().numpy().copy(),
                            a=b.action["a"].cpu().numpy().copy(), r=b.reward.cpu().numpy().copy(),
                            t=b.terminal.cpu().numpy().copy(), boot=b.bootstrap.cpu().numpy().copy(),
                            w=w.cpu().numpy().copy(), sum=st["sum"], head=st["head"], size=st["size"],
                            num_add=st["num_add"]))
            replay.update_priority(torch.linspace(0.3, 1.7, batch, device=dev))
    eng.close()
    replay.close()
    return out


@pytest.mark.parametrize("mode", ["stack", "plane"])
def test_dedup_batches_identical_to_full_storage(mode):
    """Small capacity so that both the slot ring and the unit ring wrap many times and every sample evicts."""
    rng = np.random.default_rng(31 if mode == "stack" else 32)
    R, K, n, ticks = 12, 4, 3, 90
    if mode == "plane":
        stacks, term = _sliding_frames(rng, R, ticks, 0.08)
    else:  # unrelated stacks every step (what the LCG env of SURVEY 8d produces): only whole stacks are shared
        stacks = rng.integers(0, 256, (ticks, R, 4, 84, 84), dtype=np.uint8)
        term = (rng.uniform(size=(ticks, R)) < 0.08).astype(np.uint8)
    rewards = rng.integers(-1, 2, (ticks, R)).astype(np.float32)
    ref = _run(None, stacks, term, rewards, cap=96, batch=16, n=n, K=K, sample_every=2)
    got = _run(mode, stacks, term, rewards, cap=96, batch=16, n=n, K=K, sample_every=2)
    assert len(ref) == len(got) >= 30
    for i, (a, b) in enumerate(zip(ref, got)):
        for key in ("a", "r", "t", "boot", "w", "s", "ns"):
            assert np.array_equal(a[key], b[key]), (mode, i, key)
        assert a["sum"] == b["sum"] and a["head"] == b["head"] and a["size"] == b["size"] and a["num_add"] == b["num_add"]
    assert ref[-1]["head"] != ref[0]["head"]  # eviction happened: the rings wrapped


def test_dedup_plane_survives_dropped_blocks():
    """Non-blocking producers on a full ring drop blocks (bench mode).  In plane mode the stacks of dropped ticks
    never enter the unit ring; the shard restarts with a keyframe and drops the transitions that would refer to a
    missing stack.  Everything that IS sampled must still be a correct pair: obs = the env's stack at its tick,
    next_obs = the same env's stack n ticks later (the frames carry (tick, row) tags)."""
    rng = np.random.default_rng(5)
    R, K, n, ticks = 8, 4, 2, 120
    stacks, term = _sliding_frames(rng, R, ticks, 0.05)
    term[:] = 0  # keep the stack tags unambiguous: no episode starts after the first
    stacks, _ = _sliding_frames(np.random.default_rng(5), R, ticks, 0.0)
    rewards = np.zeros((ticks, R), np.float32)
    # ring = 40: full after 5 inserts; samples (which evict) only every 10 ticks -> many dropped blocks
    got = _run("plane", stacks, term, rewards, cap=32, batch=8, n=n, K=K, sample_every=10, nonblocking=True)
    assert len(got) >= 8 and got[-1]["num_add"] < (ticks - n) * R  # blocks were dropped
    for rec in got:
        for s, ns in zip(rec["s"], rec["ns"]):
            t0, r0 = int(s[3, 0, 0]), int(s[3, 0, 1])  # newest plane of obs: (tick, row)
            assert np.array_equal(s, stacks[t0, r0]) and np.array_equal(ns, stacks[t0 + n, r0])


def test_dedup_plane_ring_of_2p23_slots_on_one_gpu():
    """BASELINE C5's replay size on ONE GPU: capacity 2^23 (ring 10,485,760 slots) with de-duplicated planes -- 74 GB of
    unit storage instead of the 592 GB two full stacks per slot would take.  6,400 envs feed it through the actor shard
    until the slot ring AND the unit ring have wrapped.  Every plane's bytes are a function of (tick, row) and its first
    four bytes carry (tick, row) themselves, so every sampled transition is checked IN FULL on the device: obs must be
    the env's four planes up to its tick (the first plane repeated at the start of the run, game_state.h:66-70),
    next_obs the same env's stack n ticks later."""
    import torch

    from rela_amd.engine import ApexActorEngine, FFNetHandle
    from rela_amd.replay import FFReplay
    from synth import synth_params

    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~90 GB of free HBM")
    dev = "cuda:0"
    R, K, n, A, cap, B = 6400, 80, 3, 6, 1 << 23, 512
    on, tg = FFNetHandle(A, dev), FFNetHandle(A, dev)
    on.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 11).items()})
    tg.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 12).items()})
    on.set_precision("bf16x2")
    tg.set_precision("bf16x2")
    replay = FFReplay(cap, 7, 0.6, 0.4, 0, A, dev, dedup="plane", guard_units=(n + 8) * R)
    eng = ApexActorEngine(R, K, A, n, 0.997, replay, [0.0] * R, dev)
    pix = torch.arange(84 * 84, device=dev, dtype=torch.int32)[None, :]

    def plane(tick, row):  # tick, row: int32 [m, 1] -> [m, 7056] u8
        p = ((tick * 131 + row * 31 + pix * 7) & 255).to(torch.uint8)
        p[:, 0], p[:, 1] = (tick[:, 0] & 255).to(torch.uint8), (tick[:, 0] >> 8).to(torch.uint8)
        p[:, 2], p[:, 3] = (row[:, 0] & 255).to(torch.uint8), (row[:, 0] >> 8).to(torch.uint8)
        return p

    def stack_at(tick, row):  # the env's stack at `tick`: planes tick-3 .. tick, clamped to the first plane
        return torch.stack([plane(torch.clamp(tick - 3 + k, min=0), row) for k in range(4)], 1).reshape(-1, 4, 84, 84)

    all_rows = torch.arange(R, device=dev, dtype=torch.int32)[:, None]
    ring = int(1.25 * cap)
    ticks = ring // R + 420  # the slot ring is full after 1,638 ticks; 420 more wrap it and the unit ring behind it
    zeros_r, zeros_t = torch.zeros(R, device=dev), torch.zeros(R, dtype=torch.uint8, device=dev)
    cur = stack_at(torch.zeros(R, 1, dtype=torch.int32, device=dev), all_rows)
    checked = 0
    for t in range(ticks):
        if t > 0:
            tt = torch.full((R, 1), t, dtype=torch.int32, device=dev)
            cur = torch.cat([cur[:, 1:], plane(tt, all_rows).reshape(R, 1, 84, 84)], 1)
        eng.next_obs_slot().copy_(cur)
        eng.act(on)
        eng.post_step(zeros_r, zeros_t, on, tg, nonblocking=False)
        if replay.size() > ring - R or t == ticks - 1:  # sample (which evicts down to capacity) and verify the batch
            b, w = replay.sample(B)
            s, ns = b.obs["s"].reshape(B, 4, -1), b.next_obs["s"].reshape(B, 4, -1)
            t0 = (s[:, 3, 0].int() | (s[:, 3, 1].int() << 8))[:, None]  # the newest plane names its tick and row
            r0 = (s[:, 3, 2].int() | (s[:, 3, 3].int() << 8))[:, None]
            assert int(t0.max()) <= t - n and int(t0.min()) >= 0 and int(r0.max()) < R
            assert torch.equal(b.obs["s"], stack_at(t0, r0)), "obs is not the env's stack at its tick"
            assert torch.equal(b.next_obs["s"], stack_at(t0 + n, r0)), "next_obs is not the same env's stack n ticks later"
            assert replay.debug_state()["dev_error"] == 0
            checked += B
            replay.update_priority(torch.linspace(0.3, 1.7, B, device=dev))
    st = replay.debug_state()
    assert st["num_add"] > ring + 2 * R and checked >= 2 * B  # the rings wrapped and samples were checked
    assert st["size"] >= cap - R
    eng.close()
    replay.close()
