"""rela_amd/csrc/sleef_powf_core.h (restatement of SLEEF's Sleef_powf*_u10, the function behind the reference's
torch::pow(priority, alpha), rela/prioritized_replay.h:188,239) compiled for the host, against vectors recorded
from the real ATen op in the build container (tests/golden/aten_powf_vectors.json) -- bit for bit, including
the scalar n % 32 tail that ATen evaluates with double pow.  CPU only."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import f2h, h2f

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim():
    src = os.path.join(HERE, "cpu_shims", "sleef_powf_host.cpp")
    hdr = os.path.join(HERE, "..", "rela_amd", "csrc", "sleef_powf_core.h")
    so = os.path.join(HERE, "cpu_shims", "libsleef_powf_host.so")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-o", so, src], check=True)
    return C.CDLL(so)


def test_host_build_matches_aten_vectors(shim):
    cases = json.load(open(os.path.join(HERE, "golden", "aten_powf_vectors.json")))["expect"]
    assert len(cases) > 100
    total = 0
    for c in cases:
        x = np.array([h2f(v) for v in c["x"]], np.float32)
        out = np.zeros_like(x)
        shim.shim_aten_pow(x.ctypes.data_as(C.c_void_p), len(x), C.c_float(h2f(c["exponent"])), out.ctypes.data_as(C.c_void_p))
        assert [f2h(v) for v in out] == c["y"], (c["exponent"], c["n"])
        total += len(x)
    assert total > 15000


def test_against_torch_pow_when_available(shim):
    """A denser sweep against this machine's ATen (skipped where its CPU capability is not the recorded one:
    the vector part is SLEEF only in the vectorised builds)."""
    import torch

    if torch.backends.cpu.get_cpu_capability() not in ("AVX512", "AVX2"):
        pytest.skip("ATen CPU capability %s" % torch.backends.cpu.get_cpu_capability())
    torch.set_num_threads(1)
    rng = np.random.default_rng(3)
    for ex in (0.6, 0.9, -0.4):
        for n in (80, 4096, 1000):
            x = (np.abs(rng.normal(0, 1, n)) * 10 ** rng.uniform(-6, 3, n)).astype(np.float32)
            ref = torch.pow(torch.from_numpy(x), float(np.float32(ex))).numpy()
            out = np.zeros_like(x)
            shim.shim_aten_pow(x.ctypes.data_as(C.c_void_p), n, C.c_float(ex), out.ctypes.data_as(C.c_void_p))
            assert np.array_equal(ref.view(np.uint32), out.view(np.uint32)), (ex, n)
