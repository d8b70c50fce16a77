"""Deterministic synthetic inputs shared by the golden generator and the tests (numpy only)."""
import numpy as np


def synth_params(num_action, seed):
    """Deterministic fp32 parameters in state_dict order, reproducible with numpy alone."""
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
    return out


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
