"""GPU tests of the sequence features of the replay C ABI: three-phase append (begin / write /
commit, prioritized_replay.h:43-78) and the time-major gather of RNNTransition::makeBatch
(types.cc:140-182)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_three_phase_append_and_time_major_gather():
    import torch

    from gpu_util import cur_stream
    from oracle_lib import OracleReplay
    from rela_amd import _capi as capi

    T, E = 5, 48  # 5 steps of 48 bytes per slot in field 0; field 1 is a plain 16-byte row
    h = C.c_void_p()
    capi.check(capi.lib.rela_replay_create(C.byref(h), 16, 3, 1.0, 1.0, 0, 0), "create")
    rb = (C.c_int64 * 2)(T * E, 16)
    st = (C.c_int32 * 2)(T, 1)
    capi.check(capi.lib.rela_replay_set_schema_seq(h, 2, rb, st), "schema")
    oracle = OracleReplay(16, 3, 1.0, 1.0)
    rng = np.random.default_rng(0)
    seqs = rng.integers(0, 256, (12, T, E), dtype=np.uint8)
    plain = rng.integers(0, 256, (12, 16), dtype=np.uint8)
    d_seq, d_plain = torch.from_numpy(seqs).cuda(), torch.from_numpy(plain).cuda()
    prio = rng.uniform(0.1, 2, 12).astype(np.float32)
    d_prio = torch.from_numpy(prio).cuda()
    # block 1: 5 slots written out of order and piecewise, visible only after commit
    slot = C.c_int()
    capi.check(capi.lib.rela_replay_begin_add(h, 5, 0, C.byref(slot)), "begin")
    assert slot.value == 0 and capi.lib.rela_replay_size(h) == 0
    for off in (3, 0, 4, 1, 2):
        rows = (C.c_void_p * 2)(d_seq[off].data_ptr(), d_plain[off].data_ptr())
        capi.check(capi.lib.rela_replay_write_rows(h, slot.value, off, 1, rows, cur_stream()), "write")
    assert capi.lib.rela_replay_size(h) == 0  # safeSize_: nothing committed yet
    capi.check(capi.lib.rela_replay_commit_add(h, slot.value, 5, C.c_void_p(d_prio.data_ptr()), cur_stream()), "commit")
    assert capi.lib.rela_replay_size(h) == 5 and capi.lib.rela_replay_num_add(h) == 5
    assert oracle.add(np.arange(5), prio[:5]) == 0
    # block 2 through the one-shot add
    rows = (C.c_void_p * 2)(d_seq[5:].data_ptr(), d_plain[5:].data_ptr())
    capi.check(capi.lib.rela_replay_add(h, 7, rows, C.c_void_p(d_prio[5:].data_ptr()), 0, cur_stream()), "add")
    assert oracle.add(np.arange(5, 12), prio[5:]) == 0
    B = 6
    out_seq = torch.empty((T, B, E), dtype=torch.uint8, device="cuda")
    out_plain = torch.empty((B, 16), dtype=torch.uint8, device="cuda")
    w = torch.empty(B, device="cuda")
    outs = (C.c_void_p * 2)(out_seq.data_ptr(), out_plain.data_ptr())
    capi.check(capi.lib.rela_replay_sample(h, B, outs, C.c_void_p(w.data_ptr()), cur_stream()), "sample")
    rc, ids, tags, ow = oracle.sample(B)
    assert rc == 0
    st_ = capi.ReplayState()
    ids_dev = np.zeros(B, np.int32)
    capi.check(capi.lib.rela_replay_debug_state(h, C.byref(st_), ids_dev.ctypes.data_as(C.c_void_p), None, None), "dbg")
    np.testing.assert_array_equal(ids_dev, ids)
    assert st_.sum == oracle.state()["sum"]
    got = out_seq.cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(got[:, b], seqs[ids[b]])  # out[t][b] = slot_b[t]
    np.testing.assert_array_equal(out_plain.cpu().numpy(), plain[ids])
    np.testing.assert_allclose(w.cpu().numpy(), ow, rtol=4e-7)
    capi.lib.rela_replay_destroy(h)


def test_gathered_row_write_matches_per_row_writes():
    """rela_replay_write_rows_gather (one launch per field for any number of rows, arbitrary source
    rows and destination offsets, wrap-around of the ring) stores exactly what per-row
    rela_replay_write_rows calls store."""
    import torch

    from gpu_util import cur_stream
    from rela_amd import _capi as capi

    T, E = 4, 32
    rng = np.random.default_rng(1)
    src_seq = torch.from_numpy(rng.integers(0, 256, (20, T * E), dtype=np.uint8)).cuda()   # 20 candidate rows
    src_plain = torch.from_numpy(rng.integers(0, 256, (20, 12), dtype=np.uint8)).cuda()    # not 16-byte rows
    prio = torch.from_numpy(rng.uniform(0.1, 2, 16).astype(np.float32)).cuda()
    results = []
    for gathered in (False, True):
        h = C.c_void_p()
        capi.check(capi.lib.rela_replay_create(C.byref(h), 8, 3, 1.0, 1.0, 0, 0), "create")  # ring = 10
        rb = (C.c_int64 * 2)(T * E, 12)
        st = (C.c_int32 * 2)(T, 1)
        capi.check(capi.lib.rela_replay_set_schema_seq(h, 2, rb, st), "schema")
        slot = C.c_int()
        # first block fills 7 of the 10 ring slots, second block (after a sample pops nothing) wraps around
        for blk, (n, pick_seq, pick_plain) in enumerate([(7, [3, 19, 0, 7, 7, 12, 5], [1, 2, 3, 4, 5, 6, 0])]):
            capi.check(capi.lib.rela_replay_begin_add(h, n, 0, C.byref(slot)), "begin")
            order = [4, 0, 6, 2, 1, 5, 3]  # destination offsets in scrambled order
            if gathered:
                dst = torch.tensor(order, dtype=torch.int32).cuda()
                i_seq = torch.tensor([pick_seq[o] for o in order], dtype=torch.int32).cuda()
                i_plain = torch.tensor([pick_plain[o] for o in order], dtype=torch.int32).cuda()
                bases = (C.c_void_p * 2)(src_seq.data_ptr(), src_plain.data_ptr())
                idx = (C.c_void_p * 2)(i_seq.data_ptr(), i_plain.data_ptr())
                capi.check(capi.lib.rela_replay_write_rows_gather(h, slot.value, n, C.c_void_p(dst.data_ptr()), bases,
                                                                  idx, cur_stream()), "gather")
            else:
                for o in order:
                    rows = (C.c_void_p * 2)(src_seq[pick_seq[o]].data_ptr(), src_plain[pick_plain[o]].data_ptr())
                    capi.check(capi.lib.rela_replay_write_rows(h, slot.value, o, 1, rows, cur_stream()), "write")
            capi.check(capi.lib.rela_replay_commit_add(h, slot.value, n, C.c_void_p(prio.data_ptr()), cur_stream()),
                       "commit")
        B = 8
        out_seq = torch.empty((T, B, E), dtype=torch.uint8, device="cuda")
        out_plain = torch.empty((B, 12), dtype=torch.uint8, device="cuda")
        w = torch.empty(B, device="cuda")
        outs = (C.c_void_p * 2)(out_seq.data_ptr(), out_plain.data_ptr())
        capi.check(capi.lib.rela_replay_sample(h, B, outs, C.c_void_p(w.data_ptr()), cur_stream()), "sample")
        torch.cuda.synchronize()
        st_ = capi.ReplayState()
        ids = np.zeros(B, np.int32)
        capi.check(capi.lib.rela_replay_debug_state(h, C.byref(st_), ids.ctypes.data_as(C.c_void_p), None, None), "dbg")
        results.append((ids.copy(), out_seq.cpu().numpy().copy(), out_plain.cpu().numpy().copy(), w.cpu().numpy().copy()))
        # the stored rows are the picked source rows
        seq_np, plain_np = src_seq.cpu().numpy(), src_plain.cpu().numpy()
        for b in range(B):
            q = int(ids[b])  # slot == destination offset (first block starts at slot 0)
            assert np.array_equal(out_seq.cpu().numpy()[:, b, :].reshape(-1), seq_np[[3, 19, 0, 7, 7, 12, 5][q]])
            assert np.array_equal(out_plain.cpu().numpy()[b], plain_np[[1, 2, 3, 4, 5, 6, 0][q]])
        capi.lib.rela_replay_update_priority(h, B, C.c_void_p(w.data_ptr()), 1, cur_stream())
        torch.cuda.synchronize()
        capi.lib.rela_replay_destroy(h)
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)


def test_abort_releases_a_reservation():
    """A block reserved by begin_add whose producer fails must not wedge the ring: after
    rela_replay_abort_add the blocks behind it commit, the aborted slots carry zero weight, are never
    drawn, and sum_ ignores them."""
    import torch

    from gpu_util import cur_stream
    from rela_amd import _capi as capi

    h = C.c_void_p()
    capi.check(capi.lib.rela_replay_create(C.byref(h), 32, 3, 1.0, 1.0, 0, 0), "create")
    rb = (C.c_int64 * 1)(8)
    capi.check(capi.lib.rela_replay_set_schema(h, 1, rb), "schema")
    tags = torch.arange(24, dtype=torch.int64, device="cuda")
    prio = torch.ones(24, device="cuda")
    s0, s1 = C.c_int(), C.c_int()
    capi.check(capi.lib.rela_replay_begin_add(h, 8, 0, C.byref(s0)), "begin")
    capi.check(capi.lib.rela_replay_begin_add(h, 8, 0, C.byref(s1)), "begin")
    rows = (C.c_void_p * 1)(tags[8:].data_ptr())
    capi.check(capi.lib.rela_replay_write_rows(h, s1.value, 0, 8, rows, cur_stream()), "write")
    capi.check(capi.lib.rela_replay_abort_add(h, s0.value, 8), "abort")  # the first producer gives up
    capi.check(capi.lib.rela_replay_commit_add(h, s1.value, 8, C.c_void_p(prio.data_ptr()), cur_stream()), "commit")
    assert capi.lib.rela_replay_size(h) == 16 and capi.lib.rela_replay_num_add(h) == 8
    out = torch.empty(16, dtype=torch.int64, device="cuda")
    w = torch.empty(16, device="cuda")
    outs = (C.c_void_p * 1)(out.data_ptr())
    capi.check(capi.lib.rela_replay_sample(h, 16, outs, C.c_void_p(w.data_ptr()), cur_stream()), "sample")
    st = capi.ReplayState()
    ids = np.zeros(16, np.int32)
    capi.check(capi.lib.rela_replay_debug_state(h, C.byref(st), ids.ctypes.data_as(C.c_void_p), None, None), "dbg")
    assert st.dev_error == 0 and st.sum == 8.0
    assert ids.min() >= 8 and sorted(set(out.cpu().tolist())) == list(range(8, 16))
    capi.lib.rela_replay_destroy(h)


def test_shutdown_releases_a_commit_waiting_for_its_turn():
    """rela_replay_shutdown wakes a commit that waits behind a block that will never commit."""
    import threading

    import torch

    from gpu_util import cur_stream
    from rela_amd import _capi as capi

    h = C.c_void_p()
    capi.check(capi.lib.rela_replay_create(C.byref(h), 32, 3, 1.0, 1.0, 0, 0), "create")
    s0, s1 = C.c_int(), C.c_int()
    capi.check(capi.lib.rela_replay_begin_add(h, 4, 0, C.byref(s0)), "begin")
    capi.check(capi.lib.rela_replay_begin_add(h, 4, 0, C.byref(s1)), "begin")
    prio = torch.ones(4, device="cuda")
    stream = cur_stream()
    rc = []
    th = threading.Thread(target=lambda: rc.append(capi.lib.rela_replay_commit_add(
        h, s1.value, 4, C.c_void_p(prio.data_ptr()), stream)))
    th.start()
    th.join(0.3)
    assert th.is_alive()  # parked behind the uncommitted first block (in-order commit, :69)
    capi.check(capi.lib.rela_replay_shutdown(h), "shutdown")
    th.join(10)
    assert not th.is_alive() and rc == [capi.EWOULDBLOCK]
    capi.lib.rela_replay_destroy(h)


def test_is_weights_use_size_including_reservations():
    """prioritized_replay.h:312,321: the IS weights use size_ (reserved, uncommitted blocks included), not
    the committed size the scan ran over.  beta = 1 makes the weights exact: w_i = 1/(size * w_i/sum) / max."""
    import torch

    from gpu_util import cur_stream
    from rela_amd import _capi as capi

    outs_w = []
    for reserve in (0, 5):
        h = C.c_void_p()
        capi.check(capi.lib.rela_replay_create(C.byref(h), 64, 3, 1.0, 0.5, 0, 0), "create")
        prio = torch.arange(1, 17, device="cuda", dtype=torch.float32)
        capi.check(capi.lib.rela_replay_add(h, 16, None, C.c_void_p(prio.data_ptr()), 0, cur_stream()), "add")
        slot = C.c_int()
        if reserve:
            capi.check(capi.lib.rela_replay_begin_add(h, reserve, 0, C.byref(slot)), "begin")
        w = torch.empty(8, device="cuda")
        capi.check(capi.lib.rela_replay_sample(h, 8, None, C.c_void_p(w.data_ptr()), cur_stream()), "sample")
        st = capi.ReplayState()
        raw = np.zeros(8, np.float32)
        capi.check(capi.lib.rela_replay_debug_state(h, C.byref(st), None, raw.ctypes.data_as(C.c_void_p), None), "dbg")
        size = np.float32(16 + reserve)
        q = (raw / np.float32(st.sum)).astype(np.float32)
        exp = np.power((size * q).astype(np.float32), np.float32(-0.5)).astype(np.float32)
        exp = exp / exp.max()
        np.testing.assert_allclose(w.cpu().numpy(), exp, rtol=3e-7)
        outs_w.append(w.cpu().numpy())
        if reserve:
            capi.check(capi.lib.rela_replay_abort_add(h, slot.value, reserve), "abort")
        capi.lib.rela_replay_destroy(h)
    # normalised by the maximum, the factor size^-beta cancels: both runs agree to rounding
    np.testing.assert_allclose(outs_w[0], outs_w[1], rtol=3e-7)


def test_time_major_gather_beyond_65535_rows():
    """batch * steps > 65,535 (the grid.y limit): B = 1100 sequences of T = 64 steps in one sample."""
    import torch

    from gpu_util import cur_stream
    from rela_amd import _capi as capi

    T, E, N, B = 64, 16, 1200, 1100
    h = C.c_void_p()
    capi.check(capi.lib.rela_replay_create(C.byref(h), 2048, 3, 1.0, 1.0, 0, 0), "create")
    rb = (C.c_int64 * 1)(T * E)
    st = (C.c_int32 * 1)(T)
    capi.check(capi.lib.rela_replay_set_schema_seq(h, 1, rb, st), "schema")
    rng = np.random.default_rng(5)
    seqs = rng.integers(0, 256, (N, T, E), dtype=np.uint8)
    d_seq = torch.from_numpy(seqs).cuda()
    prio = torch.ones(N, device="cuda")
    rows = (C.c_void_p * 1)(d_seq.data_ptr())
    capi.check(capi.lib.rela_replay_add(h, N, rows, C.c_void_p(prio.data_ptr()), 0, cur_stream()), "add")
    out = torch.zeros((T, B, E), dtype=torch.uint8, device="cuda")
    w = torch.empty(B, device="cuda")
    outs = (C.c_void_p * 1)(out.data_ptr())
    capi.check(capi.lib.rela_replay_sample(h, B, outs, C.c_void_p(w.data_ptr()), cur_stream()), "sample")
    st_ = capi.ReplayState()
    ids = np.zeros(B, np.int32)
    capi.check(capi.lib.rela_replay_debug_state(h, C.byref(st_), ids.ctypes.data_as(C.c_void_p), None, None), "dbg")
    assert st_.dev_error == 0
    np.testing.assert_array_equal(out.cpu().numpy(), seqs[ids].transpose(1, 0, 2))
    capi.lib.rela_replay_destroy(h)
