"""rela_amd -- MI355X-native actor-learner hot path behind rela's pybind surface.

`rela_amd._capi` is the ctypes view of the C ABI (include/rela_amd.h); `rela_amd.replay`,
`rela_amd.actor` ... are the host-side mirrors of the reference interface built on it.
Importing the package never falls back to a CPU implementation.
"""
__all__ = ["build"]
