"""GPU parity tests of the hand-written AtariFFNet forward (rela_amd/csrc/ffnet.hip), through the
C ABI.  fp32 tolerance 1e-4 abs + 1e-4 rel on Q-values (the reference's library convolutions
sum in a different order; our MFMA path is an exact fp32 fmaf chain)."""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = ATOL = 1e-4


class GpuNet:
    def __init__(self, params, A):
        from rela_amd import _capi as capi

        self.capi = capi
        h = C.c_void_p()
        capi.check(capi.lib.rela_ffnet_create(C.byref(h), A, 0), "rela_ffnet_create")
        self.h, self.A = h, A
        keys = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
                "linear.0.weight", "linear.0.bias", "fc_v.weight", "fc_v.bias", "fc_a.weight", "fc_a.bias"]
        p = capi.FFNetParams()
        keep = []
        for (field, _), k in zip(capi.FFNetParams._fields_, keys):
            a = np.ascontiguousarray(params[k], np.float32)
            keep.append(a)
            setattr(p, field, a.ctypes.data_as(C.c_void_p))
        capi.check(capi.lib.rela_ffnet_load(h, C.byref(p), 0, None), "rela_ffnet_load")

    def forward(self, s, legal):
        import torch

        from gpu_util import cur_stream, dev, ptr

        n = s.shape[0]
        d_s, d_l = dev(s), dev(legal)
        q = torch.empty((n, self.A), device="cuda")
        nbytes = self.capi.lib.rela_ffnet_workspace_bytes(self.h, n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        self.capi.check(self.capi.lib.rela_ffnet_forward(self.h, n, ptr(d_s), ptr(d_l), ptr(q), ptr(ws), nbytes,
                                                         cur_stream()), "rela_ffnet_forward")
        torch.cuda.synchronize()
        return q

    def close(self):
        if self.h:
            self.capi.lib.rela_ffnet_destroy(self.h)
            self.h = None


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ffnet_*.json"))), ids=os.path.basename)
def test_ffnet_golden(path):
    """Q(s), greedy action and TD priority vs the reference's own net.py/apex.py outputs."""
    import torch

    from gpu_util import cur_stream, dev, ptr
    from synth import synth_obs, synth_params

    g = json.load(open(path))
    A, N = g["num_action"], g["N"]
    on, tg = GpuNet(synth_params(A, g["online_seed"]), A), GpuNet(synth_params(A, g["target_seed"]), A)
    s, ns = synth_obs(N, g["obs_seed"]), synth_obs(N, g["next_obs_seed"])
    legal = np.ones((N, A), np.float32)
    if g["legal_mode"] == "mask":
        legal[:, 1::2] = 0.0
    q = on.forward(s, legal)
    np.testing.assert_allclose(q.cpu().numpy(), np.array(g["q"]), rtol=RTOL, atol=ATOL)
    qn = on.forward(ns, legal)
    qt = tg.forward(ns, legal)
    np.testing.assert_allclose(qn.cpu().numpy(), np.array(g["q_next_online"]), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(qt.cpu().numpy(), np.array(g["q_next_target"]), rtol=RTOL, atol=ATOL)
    capi = on.capi
    d_legal, d_eps = dev(legal), dev(np.zeros(N, np.float32))
    act = torch.empty(N, dtype=torch.int64, device="cuda")
    capi.check(capi.lib.rela_apex_act_from_q(N, A, 0, ptr(q), ptr(d_legal), ptr(d_eps), 0, 0, ptr(act), cur_stream()), "act")
    assert act.cpu().numpy().tolist() == g["greedy"] == g["act_eps0"]
    d_a, d_r, d_b = dev(np.array(g["action"], np.int64)), dev(np.array(g["reward"], np.float32)), dev(
        np.array(g["bootstrap"], np.float32))
    td, pr = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    gamma_n = np.float32(np.float64(g["gamma"]) ** g["multi_step"])
    capi.check(capi.lib.rela_apex_td_from_q(N, A, 0, ptr(q), ptr(qn), ptr(qt), ptr(d_legal), ptr(d_a), ptr(d_r), ptr(d_b),
                                            C.c_float(gamma_n), ptr(td), ptr(pr), cur_stream()), "td")
    np.testing.assert_allclose(td.cpu().numpy(), np.array(g["td_err"]), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pr.cpu().numpy(), np.array(g["priority"]), rtol=RTOL, atol=ATOL)
    on.close()
    tg.close()


@pytest.mark.parametrize("N", [1, 2, 3, 7, 80, 130, 257, 514, 1537, 2003, 6400])
def test_ffnet_vs_torch_fp32(N):
    """Ragged batch sizes (partial sample tiles in every kernel) vs a plain PyTorch fp32 forward of
    the same architecture on the same device.  N >= 512 / 1536 switches conv2 / conv3 to the
    weight-stationary persistent kernels (uneven groups per block, partial last group); N = 6400 is the
    shape of bench.py's actor tick (80 threads x 80 envs)."""
    import torch
    import torch.nn.functional as F

    from synth import synth_obs, synth_params

    A = 18
    p = synth_params(A, 77)
    net = GpuNet(p, A)
    s = synth_obs(N, 1000 + N)
    rng = np.random.default_rng(N)
    legal = (rng.uniform(size=(N, A)) < 0.8).astype(np.float32)
    q = net.forward(s, legal).cpu().numpy()
    t = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    torch.backends.cudnn.allow_tf32 = False
    with torch.no_grad():
        x = torch.from_numpy(s).cuda().float() / 255.0
        x = F.relu(F.conv2d(x, t["net.0.weight"], t["net.0.bias"], stride=4))
        x = F.relu(F.conv2d(x, t["net.2.weight"], t["net.2.bias"], stride=2))
        x = F.relu(F.conv2d(x, t["net.4.weight"], t["net.4.bias"], stride=1))
        h = F.relu(F.linear(x.reshape(N, 3136), t["linear.0.weight"], t["linear.0.bias"]))
        v = F.linear(h, t["fc_v.weight"], t["fc_v.bias"])
        a = F.linear(h, t["fc_a.weight"], t["fc_a.bias"]) * torch.from_numpy(legal).cuda()
        ref = (v + a - a.mean(1, keepdim=True)).cpu().numpy()
    np.testing.assert_allclose(q, ref, rtol=RTOL, atol=ATOL)
    net.close()


def test_ffnet_vs_oracle_c():
    """Same comparison against the plain-C oracle (oracle/dqn_oracle.c), N small."""
    from oracle_lib import load
    from synth import synth_obs, synth_params
    from test_oracle_golden import _ffnet_struct

    A, N = 18, 6
    p = synth_params(A, 5)
    net = GpuNet(p, A)
    s = synth_obs(N, 6)
    legal = np.ones((N, A), np.float32)
    q = net.forward(s, legal).cpu().numpy()
    lib = load()
    lib.oracle_set_threads(4)
    on = _ffnet_struct(lib, p, A)
    ref = np.zeros((N, A), np.float32)
    lib.oracle_ffnet_forward(C.byref(on), N, s.ctypes.data_as(C.POINTER(C.c_uint8)),
                             legal.ctypes.data_as(C.POINTER(C.c_float)), ref.ctypes.data_as(C.POINTER(C.c_float)))
    np.testing.assert_allclose(q, ref, rtol=RTOL, atol=ATOL)
    net.close()


def test_ffnet_errors():
    from rela_amd import _capi as capi

    h = C.c_void_p()
    assert capi.lib.rela_ffnet_create(C.byref(h), 40, 0) == capi.EINVAL
    assert capi.lib.rela_ffnet_create(C.byref(h), 18, 0) == 0
    # forward before load_state_dict
    assert capi.lib.rela_ffnet_forward(h, 1, None, None, None, None, 0, None) == capi.ESTATE
    capi.lib.rela_ffnet_destroy(h)


# ---- split-bf16 fast path (rela_ffnet_set_precision(net, 1)) ----------------------------------------
@pytest.mark.parametrize("N", [1, 130, 1024, 1025, 2003, 3333, 6400])
def test_ffnet_fast_mode_within_stated_tolerance(N, record_property):
    """conv2 / conv3 / fc on split-bf16 MFMA (hi + lo bf16 operands, three products, f32 accumulation): Q-values
    within the stated tolerance (|dQ| < 2e-6) of the exact f32 path (parity mode) on the same weights and
    frames, at ragged batch sizes and at bench.py's actor shape (N = 6400); the greedy-action agreement between
    the two modes is recorded."""
    from synth import synth_obs, synth_params

    A = 18
    p = synth_params(A, 77)
    net = GpuNet(p, A)
    s = synth_obs(N, 3000 + N)
    rng = np.random.default_rng(N)
    legal = (rng.uniform(size=(N, A)) < 0.8).astype(np.float32)
    legal[:, 0] = 1.0
    q_ref = net.forward(s, legal).cpu().numpy()
    net.capi.check(net.capi.lib.rela_ffnet_set_precision(net.h, 1), "set_precision")
    assert net.capi.lib.rela_ffnet_precision(net.h) == 1
    q_fast = net.forward(s, legal).cpu().numpy()
    tmo = C.c_uint(7)
    net.capi.check(net.capi.lib.rela_ffnet_debug_pipe_timeout(net.h, C.byref(tmo)), "pipe_timeout")
    assert tmo.value == 0, "a wave of the pipelined conv1 -> conv2 kernel gave up on a hand-off (code %d)" % tmo.value
    net.capi.check(net.capi.lib.rela_ffnet_set_precision(net.h, 0), "set_precision")
    q_back = net.forward(s, legal).cpu().numpy()
    assert np.array_equal(q_ref, q_back)  # the parity mode is untouched by the switch
    np.testing.assert_allclose(q_fast, q_ref, rtol=RTOL, atol=ATOL)
    err = float(np.abs(q_fast - q_ref).max())
    scale = float(np.abs(q_ref).max())
    masked = lambda q: np.where(legal > 0, q, -np.inf).argmax(1)
    agree = float((masked(q_fast) == masked(q_ref)).mean())
    record_property("max_abs_err", err)
    record_property("greedy_agreement", agree)
    print("N=%d split-bf16 vs f32: max |dQ| = %.3g (|Q| up to %.3g), greedy agreement %.5f" % (N, err, scale, agree))
    assert err < 2e-6  # the stated tolerance of the fast mode; measured 6e-7 at |Q| <= 0.41
    assert agree >= 0.995 or N < 200
    net.close()


@pytest.mark.parametrize("fuse", [0, 1, 2, 3], ids=["separate", "fused", "fused-pipelined", "fused-mfma+service-waves"])
def test_ffnet_fast_mode_fusion_variants(fuse):
    """The three forms of conv1 -> conv2 in the fast mode (separate kernels; fused with conv1's output kept in LDS,
    the default; fused with layer-specialised waves handing tiles over through LDS counters) give the same Q within
    the stated tolerance, and no wave of the pipelined form ever gives up on a hand-off."""
    import subprocess
    import sys

    env = dict(os.environ, RELA_FUSE12=str(fuse))
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ffnet_fast_child.py"),
                          "2003"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["timeout"] == 0 and rec["max_err"] < 2e-6 and rec["agree"] >= 0.995, rec


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ffnet_*.json"))), ids=os.path.basename)
def test_ffnet_fast_mode_vs_reference_golden(path):
    """The fast mode against the vectors recorded from the REAL reference (same tolerance as the parity mode)."""
    from synth import synth_obs, synth_params

    g = json.load(open(path))
    A, N = g["num_action"], g["N"]
    on = GpuNet(synth_params(A, g["online_seed"]), A)
    on.capi.check(on.capi.lib.rela_ffnet_set_precision(on.h, 1), "set_precision")
    s = synth_obs(N, g["obs_seed"])
    legal = np.ones((N, A), np.float32)
    if g["legal_mode"] == "mask":
        legal[:, 1::2] = 0.0
    q = on.forward(s, legal)
    np.testing.assert_allclose(q.cpu().numpy(), np.array(g["q"]), rtol=RTOL, atol=ATOL)
    on.close()
