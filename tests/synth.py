"""Deterministic synthetic inputs shared by the golden generator and the tests (numpy only)."""
import numpy as np


def synth_params(num_action, seed, gain=1.0):
    """Deterministic fp32 parameters in state_dict order, reproducible with numpy alone.  `gain` multiplies every
    weight tensor (not the biases): 4.6 per layer gives |Q| of 30-60, the scale of a trained Atari agent, where the
    default initialisation gives |Q| <= 0.5."""
    rng = np.random.default_rng(seed)
    shapes = [
        ("net.0.weight", (32, 4, 8, 8)), ("net.0.bias", (32,)),
        ("net.2.weight", (64, 32, 4, 4)), ("net.2.bias", (64,)),
        ("net.4.weight", (64, 64, 3, 3)), ("net.4.bias", (64,)),
        ("linear.0.weight", (512, 3136)), ("linear.0.bias", (512,)),
        ("fc_v.weight", (1, 512)), ("fc_v.bias", (1,)),
        ("fc_a.weight", (num_action, 512)), ("fc_a.bias", (num_action,)),
    ]
    out = {}
    for name, shp in shapes:
        fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else int(shp[0])
        bound = 1.0 / np.sqrt(fan_in)
        out[name] = rng.uniform(-bound, bound, shp).astype(np.float32)
        if gain != 1.0 and len(shp) > 1:
            out[name] = (out[name] * np.float32(gain)).astype(np.float32)
    return out


def f32_to_hex(a):
    """float32 array -> one hex string of its little-endian bytes (compact golden storage, bit-exact)."""
    return np.ascontiguousarray(a, "<f4").tobytes().hex()


def hex_to_f32(h, shape=None):
    a = np.frombuffer(bytes.fromhex(h), "<f4").copy()
    return a if shape is None else a.reshape(shape)


def synth_obs(n, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, 4, 84, 84), dtype=np.uint8)


def synth_lstm_params(num_action, seed):
    """Deterministic AtariLSTMNet parameters in state_dict order."""
    rng = np.random.default_rng(seed)
    shapes = [
        ("net.0.weight", (32, 4, 8, 8)), ("net.0.bias", (32,)),
        ("net.2.weight", (64, 32, 4, 4)), ("net.2.bias", (64,)),
        ("net.4.weight", (64, 64, 3, 3)), ("net.4.bias", (64,)),
        ("lstm.weight_ih_l0", (2048, 3136)), ("lstm.weight_hh_l0", (2048, 512)),
        ("lstm.bias_ih_l0", (2048,)), ("lstm.bias_hh_l0", (2048,)),
        ("fc_v.weight", (1, 512)), ("fc_v.bias", (1,)),
        ("fc_a.weight", (num_action, 512)), ("fc_a.bias", (num_action,)),
    ]
    out = {}
    for name, shp in shapes:
        if name.startswith("lstm"):
            bound = 1.0 / np.sqrt(512)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else int(shp[0])
            bound = 1.0 / np.sqrt(fan_in)
        out[name] = rng.uniform(-bound, bound, shp).astype(np.float32)
    return out


def synth_r2d2_batch(seed, A, B, seq, burn, n):
    """Deterministic RNNTransition-shaped batch ([T,B,...], rela/types.cc:140-182) with consistent padding, as numpy
    arrays: sequence b has seq_len L_b; its train-part terminals are 1 from step L_b - 1 on (r2d2.py:169-173 asserts
    exactly this); sequence 0 starts an episode (padLike'd burn-in, r2d2_actor.h:55-66).  Shared by the golden
    generator (tests/golden/make_golden.py) and the GPU tests, so a large batch needs no stored inputs."""
    T = burn + seq + n
    rng = np.random.default_rng(seed)
    s = synth_obs(T * B, seed + 1).reshape(T, B, 4, 84, 84)
    legal = (rng.uniform(size=(T, B, A)) < 0.85).astype(np.float32)
    legal[:, :, 0] = 1.0
    base = [burn + seq, burn + 3, burn + seq - 1]
    lens = np.array([base[b % 3] if b < 3 else int(burn + 1 + (7 * b) % seq) for b in range(B)], np.float32)
    term = np.zeros((T, B), np.float32)
    for b in range(B):
        if lens[b] < burn + seq:
            term[int(lens[b]) - 1:, b] = 1.0
    term[:burn, 0] = 1.0
    boot = (1.0 - np.maximum.reduce([np.roll(term, -k, 0) for k in range(n)])).astype(np.float32)
    boot[T - n:] = 0.0
    u = rng.uniform(size=(T, B, A)) * legal
    action = u.argmax(2).astype(np.int64)  # a uniformly random LEGAL action
    reward = rng.normal(0, 1.2, (T, B)).astype(np.float32)
    h0 = rng.normal(0, 0.3, (1, B, 512)).astype(np.float32)
    c0 = rng.normal(0, 0.3, (1, B, 512)).astype(np.float32)
    weight = rng.uniform(0.2, 1.0, B).astype(np.float32)
    return dict(s=s, legal=legal, seq_len=lens, terminal=term, bootstrap=boot, action=action, reward=reward, h0=h0,
                c0=c0, weight=weight)
