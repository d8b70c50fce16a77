"""The C-ABI library loads (no GPU needed) and exports every function include/rela_amd.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from rela_amd import _capi as capi

    hdr = open(os.path.join(ROOT, "include", "rela_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(rela_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(capi.lib, name), "librela_amd.so does not export %s" % name
    assert capi.MISSING == []
    assert capi.lib.rela_abi_version() == 1


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without a HIP device the create calls return RELA_ENODEV."""
    import ctypes as C

    import torch

    from rela_amd import _capi as capi

    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    assert capi.lib.rela_replay_create(C.byref(h), 16, 1, 1.0, 1.0, 0, 0) == capi.ENODEV
    assert b"no CPU path" in capi.lib.rela_last_error()
    assert capi.lib.rela_ffnet_create(C.byref(h), 18, 0) == capi.ENODEV
