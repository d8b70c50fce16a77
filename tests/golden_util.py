"""Decoding of the tests/golden/ffnet_*.json fixtures (shared by the CPU oracle tests and the GPU tests).

Two generations of files: the small ones (N = 5 / 3) store every input and output as JSON lists; the large ones
(`ffnet_big_cases` of make_golden.py: N = 1024 and the trained-scale N = 256) store the reference's Q tables
bit-exactly as hex strings and re-derive legal moves / actions / rewards / bootstrap flags from `misc_seed`
exactly as the generator drew them."""
import numpy as np

from synth import hex_to_f32


def table(g, key, shape):
    return hex_to_f32(g[key + "_hex"], shape) if key + "_hex" in g else np.array(g[key], np.float32).reshape(shape)


def ffnet_case(g):
    """-> dict(A, N, gain, legal, action, reward, bootstrap) of one ffnet golden"""
    A, N = g["num_action"], g["N"]
    out = dict(A=A, N=N, gain=g.get("gain", 1.0), big="q_hex" in g)
    if out["big"]:
        rng = np.random.default_rng(g["misc_seed"])
        legal = (rng.uniform(size=(N, A)) < 0.8).astype(np.float32)
        legal[:, 0] = 1.0
        out["action"] = (rng.uniform(size=(N, A)) * legal).argmax(1).astype(np.int64)
        out["reward"] = rng.integers(-1, 2, N).astype(np.float32)
        out["bootstrap"] = (rng.uniform(size=N) < 0.8).astype(np.float32)
    else:
        legal = np.ones((N, A), np.float32)
        if g["legal_mode"] == "mask":
            legal[:, 1::2] = 0.0
        out["action"] = np.array(g["action"], np.int64)
        out["reward"] = np.array(g["reward"], np.float32)
        out["bootstrap"] = np.array(g["bootstrap"], np.float32)
    out["legal"] = legal
    return out


def decisive_rows(q_ref, legal, tol):
    """rows whose top two legal Q-values of the reference are further apart than `tol`: only there is the arg-max
    (greedy action, bootstrap action of the TD target) determined at the comparison's tolerance"""
    top2 = np.sort(np.where(legal > 0, q_ref, -np.inf), axis=1)[:, -2:]
    return (top2[:, 1] - top2[:, 0]) >= tol
