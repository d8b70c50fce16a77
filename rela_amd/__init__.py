"""rela_amd -- MI355X-native actor-learner hot path behind rela's pybind surface.

`rela_amd._capi` is the ctypes view of the C ABI (include/rela_amd.h); `rela_amd.replay`, `rela_amd.engine`
(device nets and actor shards), `rela_amd.learner` (HIP learner steps, collectives of replicated learners) and
`rela_amd.parallel` (replay partitions behind one learner) are the host-side mirrors of the reference interface
built on it; `rela_amd.pybind` holds the drop-in `rela` extension module and `rela_amd.pyrela` the training entry
points.  Importing the package never falls back to a CPU implementation.
"""
__all__ = ["build"]
