"""Host-side pieces of bench.py that must work on a fresh box without a GPU in the loop: `roofline.traffic` comes from the
PMC passes TRACKED under profiles/ (it was null in the driver's line because the summary file had not been
committed), and the launch census the parity tests assert on is part of the C ABI."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_module"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_roofline_traffic_is_available_from_tracked_profiles():
    b = _bench()
    for label, algorithmic in (("conv12_fused", b.BYTES_PER_SAMPLE["conv12_fused"] * b.ROWS),
                               ("fc_mfma", b.BYTES_PER_SAMPLE["fc_mfma"] * b.ROWS)):
        nbytes, source = b.traffic_from_profiles(label)
        assert nbytes is not None and source.startswith("profiles/"), label
        assert 0.9 * algorithmic < nbytes < 6 * algorithmic, (label, nbytes, algorithmic)
    # conv1 -> conv2 fused moves what the algorithm needs and no more (within 2 %)
    nbytes, _ = b.traffic_from_profiles("conv12_fused")
    assert abs(nbytes / (b.BYTES_PER_SAMPLE["conv12_fused"] * b.ROWS) - 1.0) < 0.02
    assert b.traffic_from_profiles("no_such_kernel") == (None, None)


def test_traffic_falls_back_to_the_raw_pmc_csvs(tmp_path, monkeypatch):
    """without a summary json the bytes come straight from profiles/r0N_pmc/*.csv (FETCH_SIZE doubled, KB -> bytes)"""
    import glob
    import shutil

    b = _bench()
    fake = tmp_path / "repo"
    (fake / "profiles").mkdir(parents=True)
    src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc")))[-1]
    shutil.copytree(src, fake / "profiles" / os.path.basename(src))
    monkeypatch.setattr(b, "ROOT", str(fake))
    nbytes, source = b.traffic_from_profiles("conv12_fused")
    assert nbytes is not None and "counter_collection.csv" in source
    assert abs(nbytes / (b.BYTES_PER_SAMPLE["conv12_fused"] * b.ROWS) - 1.0) < 0.02


def test_launch_census_symbols_and_binding():
    from rela_amd import _capi as capi

    assert hasattr(capi.lib, "rela_prof_count_enable") and hasattr(capi.lib, "rela_prof_counts_json")
    with capi.launch_census() as c:
        pass
    assert c.counts == {}
