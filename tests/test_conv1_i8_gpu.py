"""conv1 of the split-bf16 mode on the int8 matrix cores (csrc/ffnet.hip: conv12_i8): the kernel's arithmetic is
integer up to one fixed sequence of f32 operations, so conv1's records can be checked BIT FOR BIT against a numpy
restatement -- digits of the 24-bit fixed-point weights, exact i64 sums, then
    u = f32(S_hi) * 65536 + f32(S_mid * 256 + S_lo);  y = u * s_c + b'_c;  relu;  bf16 hi (RNE), bf16 lo of y - hi
and against the exact f32 convolution within the bound the quantisation gives (2^-24 of the channel's largest weight
per weight)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
QMAX = 127 * 65536 + 127 * 256 + 127


def bf16_rne(x):
    """f32 array -> (bf16 bits as uint16, the bf16 values as f32); round to nearest even, finite inputs"""
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    h = ((u + r) >> 16).astype(np.uint16)
    return h, (h.astype(np.uint32) << 16).view(np.float32)


def quantise(w, b):
    """numpy restatement of pack_conv1_i8_block: per-channel scale (f32), integer weights, folded bias (f32)"""
    w = w.reshape(32, 256).astype(np.float32) / np.float32(255)  # net.py:46's s / 255, folded into the weights
    mx = np.abs(w).max(1)
    sc = np.ones(32, np.float32)
    for c in range(32):
        if mx[c] > 0:
            s = np.float32(np.float64(mx[c]) / QMAX)
            if s <= 0 or np.float64(mx[c]) / np.float64(s) > QMAX:
                s = np.nextafter(s, np.float32(np.inf))
            sc[c] = s
    q = np.rint(w.astype(np.float64) / sc.astype(np.float64)[:, None]).astype(np.int64).clip(-QMAX, QMAX)
    bq = (b.astype(np.float64) + 128.0 * sc.astype(np.float64) * q.sum(1)).astype(np.float32)
    lo = ((q + 128) & 255) - 128
    q1 = (q - lo) >> 8
    mid = ((q1 + 128) & 255) - 128
    hi = (q1 - mid) >> 8
    assert np.abs(hi).max() <= 127 and (hi * 65536 + mid * 256 + lo == q).all()
    return sc, q, bq, (hi, mid, lo)


def patches(frames):
    """[N][4][84][84] u8 -> [N][400][256] i64 of x - 128 in the weights' k order (p, ky, kx)"""
    N = frames.shape[0]
    x = frames.astype(np.int64) - 128
    v = np.lib.stride_tricks.sliding_window_view(x, (8, 8), axis=(2, 3))[:, :, ::4, ::4]  # [N][4][20][20][8][8]
    return v.transpose(0, 2, 3, 1, 4, 5).reshape(N, 400, 256)


def expected_records(frames, w, b):
    sc, q, bq, (hi, mid, lo) = quantise(w, b)
    P = patches(frames)
    S = lambda d: np.einsum("npk,ck->npc", P, d)  # exact: |S| <= 2^22
    u = S(hi).astype(np.float32) * np.float32(65536) + (S(mid) * 256 + S(lo)).astype(np.float32)  # (int64 -> f32: RNE)
    y = u * sc[None, None, :] + bq[None, None, :]
    y = np.maximum(y, np.float32(0))
    hb, hv = bf16_rne(y)
    lb, _ = bf16_rne(y - hv)
    return np.concatenate([hb, lb], axis=2), y, (sc, q, bq)


@pytest.mark.parametrize("N,seed,gain", [(3, 1, 1.0), (257, 2, 1.0), (300, 3, 2.6)])
def test_conv1_records_bit_exact(N, seed, gain):
    import torch

    from gpu_util import dev
    from rela_amd import _capi as capi
    from rela_amd.engine import FFNetHandle
    from synth import synth_obs, synth_params

    A = 18
    params = synth_params(A, seed, gain)
    net = FFNetHandle(A, "cuda:0")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    net.set_precision("bf16x2")
    frames = synth_obs(N, 10 + seed)
    s = dev(frames)
    a1 = torch.zeros((N, 400, 64), dtype=torch.int16, device="cuda")
    a2 = torch.zeros((N, 81, 128), dtype=torch.int16, device="cuda")
    sc = np.zeros(32, np.float32)
    bq = np.zeros(32, np.float32)
    with capi.launch_census() as census:
        capi.check(capi.lib.rela_ffnet_debug_conv12_records(net.h, N, C.c_void_p(s.data_ptr()), C.c_void_p(a1.data_ptr()),
                                                            C.c_void_p(a2.data_ptr()), sc.ctypes.data_as(C.c_void_p),
                                                            bq.ctypes.data_as(C.c_void_p), None), "records")
    assert "conv12_i8_jobs" in census.counts, census.counts
    want, y, (sc_ref, q, bq_ref) = expected_records(frames, params["net.0.weight"], params["net.0.bias"])
    np.testing.assert_array_equal(sc.view(np.uint32), sc_ref.view(np.uint32))
    np.testing.assert_array_equal(bq.view(np.uint32), bq_ref.view(np.uint32))
    got = a1.cpu().numpy().view(np.uint16)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first differing (frame, pixel, half-channel): %s of %d; got %s want %s" % (
        bad[:5].tolist(), len(bad), got[tuple(bad[0])], want[tuple(bad[0])])
    # against the exact convolution: every weight is within s_c / 2 of its fixed-point value
    w = (params["net.0.weight"].reshape(32, 256).astype(np.float32) / np.float32(255)).astype(np.float64)
    exact = np.maximum(np.einsum("npk,ck->npc", patches(frames) + 128, w) + params["net.0.bias"].astype(np.float64), 0)
    bound = 0.5 * sc_ref.astype(np.float64) * 255 * 256 + 4e-7 * np.abs(exact).max()
    assert (np.abs(y - exact) <= bound[None, None, :]).all()
    assert np.abs(y - exact).max() < 2e-6 * max(1.0, np.abs(exact).max())
