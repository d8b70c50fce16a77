"""GPU parity of the AtariLSTMNet step (conv trunk -> fused LSTM gates/cell GEMM -> heads) against
a plain PyTorch fp32 forward of the same architecture (rela_amd/pyrela/net.py, itself checked to
be identical to the reference's pyrela/net.py).  Tolerance 1e-4 abs+rel."""
import ctypes as C

import numpy as np
import pytest

from kernel_names import CONV12, CONV12_JOBS  # noqa: F401

pytestmark = pytest.mark.gpu

KEYS = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
        "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "fc_v.weight", "fc_v.bias",
        "fc_a.weight", "fc_a.bias"]


class GpuLstmNet:
    def __init__(self, params, A):
        from rela_amd import _capi as capi

        self.capi, self.A = capi, A
        h = C.c_void_p()
        capi.check(capi.lib.rela_lstmnet_create(C.byref(h), A, 0), "rela_lstmnet_create")
        self.h = h
        p = capi.LSTMNetParams()
        keep = []
        for (field, _), k in zip(capi.LSTMNetParams._fields_, KEYS):
            a = np.ascontiguousarray(params[k], np.float32)
            keep.append(a)
            setattr(p, field, a.ctypes.data_as(C.c_void_p))
        capi.check(capi.lib.rela_lstmnet_load(h, C.byref(p), 0, None), "rela_lstmnet_load")

    def step(self, s, legal, h_in, c_in):
        import torch

        from gpu_util import cur_stream, dev, ptr

        n = s.shape[0]
        d = [dev(x) for x in (s, legal, h_in, c_in)]
        h_out, c_out = torch.empty((n, 512), device="cuda"), torch.empty((n, 512), device="cuda")
        q, adv = torch.empty((n, self.A), device="cuda"), torch.empty((n, self.A), device="cuda")
        nb = self.capi.lib.rela_lstmnet_workspace_bytes(self.h, n)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        self.capi.check(self.capi.lib.rela_lstmnet_step(self.h, n, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(h_out),
                                                        ptr(c_out), ptr(q), ptr(adv), ptr(ws), nb, cur_stream()), "step")
        torch.cuda.synchronize()
        return h_out.cpu().numpy(), c_out.cpu().numpy(), q.cpu().numpy(), adv.cpu().numpy()

    def close(self):
        self.capi.lib.rela_lstmnet_destroy(self.h)


@pytest.mark.parametrize("precision", ["f32", "f32x3", "bf16x2"])
@pytest.mark.parametrize("N,A", [(1, 18), (5, 6), (80, 18), (131, 18), (1537, 18), (3200, 18)])
def test_lstm_step_vs_torch(N, A, precision):
    """All three precision modes against the plain PyTorch fp32 step, with the launched kernels asserted: in bf16x2 mode
    the trunk's split-bf16 kernels from 128 rows and the rec64 gate GEMM (lstm_gates_x_bf16) from 1,024 rows are
    compared with torch-fp32 directly; in f32x3 mode, from 512 rows, the conv trunk and the x part of the gate GEMM run with
    three-part operands on the bf16 matrix cores (csrc/conv12_s3.h, conv_img_s3.h, gemm_s3.h), the rest on the f32 kernels."""
    import torch

    from rela_amd.pyrela.net import AtariLSTMNet, dueling_q
    from synth import synth_lstm_params, synth_obs

    p = synth_lstm_params(A, 40 + A)
    net = GpuLstmNet(p, A)
    rng = np.random.default_rng(N)
    s = synth_obs(N, 500 + N)
    legal = (rng.uniform(size=(N, A)) < 0.85).astype(np.float32)
    h_in = rng.normal(0, 0.3, (N, 512)).astype(np.float32)
    c_in = rng.normal(0, 0.5, (N, 512)).astype(np.float32)
    net.capi.check(net.capi.lib.rela_lstmnet_set_precision(net.h, {"f32": 0, "bf16x2": 1, "f32x3": 2}[precision]), "set_precision")
    with net.capi.launch_census() as census:
        h, c, q, adv = net.step(s, legal, h_in, c_in)
    fast = {CONV12, "conv_bf16s<Conv3F>"} | ({"gemm_rec64_nt"} if N >= 1024 else set())
    # f32x3 from 512 rows (r5): the trunk on split3 records and the x part of the gates as a three-part GEMM
    emu = {"conv12_s3", "conv3_img_s3", "gemm_s3<gates_x>"}
    x3 = precision == "f32x3" and N >= 512
    if precision == "bf16x2" and N >= 128:
        assert fast <= set(census.counts), sorted(census.counts)
    else:
        assert not (fast & set(census.counts)) and (x3 or "conv1_bf16x3" in census.counts), sorted(census.counts)
    assert (emu <= set(census.counts)) if x3 else not (emu & set(census.counts)), sorted(census.counts)
    ref = AtariLSTMNet("cpu", A)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    with torch.no_grad():
        x = ref._features(torch.from_numpy(s)).unsqueeze(0)
        o, (hr, cr) = ref.lstm(x, (torch.from_numpy(h_in).unsqueeze(0), torch.from_numpy(c_in).unsqueeze(0)))
        adv_r = ref.fc_a(o).squeeze(0)
        q_r = dueling_q(ref.fc_v(o), ref.fc_a(o), torch.from_numpy(legal).unsqueeze(0), 2).squeeze(0)
    tol = dict(rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(h, hr.squeeze(0).numpy(), **tol)
    np.testing.assert_allclose(c, cr.squeeze(0).numpy(), **tol)
    np.testing.assert_allclose(adv, adv_r.numpy(), **tol)
    np.testing.assert_allclose(q, q_r.numpy(), **tol)
    net.close()


@pytest.mark.parametrize("N", [127, 128, 131, 1537, 3200])
def test_lstm_step_fast_trunk_within_stated_tolerance(N):
    """rela_lstmnet_set_precision(1): the conv trunk on split-bf16 MFMA for batches of 128 rows and more (conv1 -> conv2
    fused, conv3); below 1,024 rows a3 goes back to f32 for the f32 gate GEMM, from 1,024 rows up (1537: ragged last
    row block, 3200: the bench's shape) the x part of the gates is one split-bf16 GEMM on a3's records and the f32 MFMA
    kernel adds h x W_hh and runs the cell; the heads stay f32.  h, c, Q and the advantages stay within the fast mode's
    stated tolerance of the exact f32 step (N = 127 takes the f32 kernels: identical)."""
    from synth import synth_lstm_params, synth_obs

    A = 18
    net = GpuLstmNet(synth_lstm_params(A, 51), A)
    rng = np.random.default_rng(N)
    s = synth_obs(N, 900 + N)
    legal = np.ones((N, A), np.float32)
    h_in = rng.normal(0, 0.3, (N, 512)).astype(np.float32)
    c_in = rng.normal(0, 0.5, (N, 512)).astype(np.float32)
    ref = net.step(s, legal, h_in, c_in)
    net.capi.check(net.capi.lib.rela_lstmnet_set_precision(net.h, 1), "set_precision")
    assert net.capi.lib.rela_lstmnet_precision(net.h) == 1
    fast = net.step(s, legal, h_in, c_in)
    for a, b, name in zip(fast, ref, ("h", "c", "q", "adv")):
        err = float(np.abs(a - b).max())
        assert err < 4e-6, (name, err)
        if N < 128:
            assert np.array_equal(a, b), name
    assert float((fast[2].argmax(1) == ref[2].argmax(1)).mean()) >= 0.995
    net.close()
