"""The C-ABI library builds, loads and exports every symbol include/gsfm.h declares (no compute
calls — there is no GPU in the CPU test tier)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from glomap_amd import build

    path = build.build_lib()
    return ctypes.CDLL(str(path))


def _declared_functions():
    text = (ROOT / "include" / "gsfm.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsfm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared_functions()
    for must in ("gsfm_ra_solve", "gsfm_gp_solve", "gsfm_ba_solve", "gsfm_ctx_create", "gsfm_comm_init"):
        assert must in names


def test_all_declared_symbols_exported(lib):
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in gsfm.h but not exported by libgsfm.so: {missing}"


def test_version_and_defaults(lib):
    from glomap_amd import _lib

    l = _lib.load()
    assert l.gsfm_version() == 100
    o = _lib.RaOptions()
    l.gsfm_ra_options_default(ctypes.byref(o))
    # defaults of glomap/estimators/global_rotation_averaging.h:41-74
    assert o.max_num_l1_iterations == 5 and o.max_num_irls_iterations == 100
    assert o.l1_step_convergence_threshold == 1e-3 and o.irls_loss_parameter_sigma == 5.0
    assert o.l1_admm_max_num_iterations == 10  # gra.cc:484


def test_struct_sizes_match_header(lib):
    """ctypes mirrors must have the same size as the C structs (checked through a tiny C probe)."""
    import subprocess, tempfile, json
    from glomap_amd import _lib

    src = r'''
    #include <stdio.h>
    #include "gsfm.h"
    int main(){printf("{\"report\":%zu,\"ra_options\":%zu,\"ra_problem\":%zu,\"lm\":%zu,\"gp_options\":%zu,\"gp_problem\":%zu,\"ba_options\":%zu,\"ba_problem\":%zu,\"scene_view\":%zu,\"match_graph\":%zu,\"track_options\":%zu,\"track_set\":%zu}",
      sizeof(gsfm_report),sizeof(gsfm_ra_options),sizeof(gsfm_ra_problem),sizeof(gsfm_lm_options),sizeof(gsfm_gp_options),sizeof(gsfm_gp_problem),sizeof(gsfm_ba_options),sizeof(gsfm_ba_problem),sizeof(gsfm_scene_view),sizeof(gsfm_match_graph),sizeof(gsfm_track_options),sizeof(gsfm_track_set));return 0;}
    '''
    with tempfile.TemporaryDirectory() as d:
        c = Path(d) / "probe.c"
        c.write_text(src)
        exe = Path(d) / "probe"
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(c), "-o", str(exe)], check=True)
        sizes = json.loads(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout)
    assert sizes["report"] == ctypes.sizeof(_lib.Report)
    assert sizes["ra_options"] == ctypes.sizeof(_lib.RaOptions)
    assert sizes["ra_problem"] == ctypes.sizeof(_lib.RaProblemC)
    assert sizes["lm"] == ctypes.sizeof(_lib.LmOptions)
    assert sizes["gp_options"] == ctypes.sizeof(_lib.GpOptions)
    assert sizes["gp_problem"] == ctypes.sizeof(_lib.GpProblemC)
    assert sizes["ba_options"] == ctypes.sizeof(_lib.BaOptions)
    assert sizes["ba_problem"] == ctypes.sizeof(_lib.BaProblemC)
    assert sizes["scene_view"] == ctypes.sizeof(_lib.SceneViewC)
    assert sizes["match_graph"] == ctypes.sizeof(_lib.MatchGraphC)
    assert sizes["track_options"] == ctypes.sizeof(_lib.TrackOptionsC)
    assert sizes["track_set"] == ctypes.sizeof(_lib.TrackSetC)


def test_no_device_is_a_hard_error():
    """Without a GPU the product path must fail loudly, not fall back to the CPU."""
    import torch
    from glomap_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.GsfmError):
        _lib.Context(-1)


def test_product_never_imports_oracle():
    for py in (ROOT / "glomap_amd").rglob("*.py"):
        text = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{py} imports the oracle"


def test_python_option_defaults_match_the_library():
    """The dataclasses of glomap_amd/estimators.py and tracks.py must start from the same defaults as
    gsfm_*_options_default (two copies of the reference's option structs would otherwise drift apart)."""
    import ctypes as C

    from glomap_amd import _lib, estimators
    from glomap_amd.tracks import TrackEstablishmentOptions

    lib = _lib.load()

    def fields(cobj, prefix=""):
        out = {}
        for name, ctype in cobj._fields_:
            v = getattr(cobj, name)
            if isinstance(v, C.Structure):
                out.update(fields(v, prefix + name + "."))
            else:
                out[prefix + name] = v
        return out

    for default_fn, ctype, pyobj in ((lib.gsfm_ra_options_default, _lib.RaOptions, estimators.RotationEstimatorOptions()),
                                     (lib.gsfm_gp_options_default, _lib.GpOptions, estimators.GlobalPositionerOptions()),
                                     (lib.gsfm_ba_options_default, _lib.BaOptions, estimators.BundleAdjusterOptions()),
                                     (lib.gsfm_track_options_default, _lib.TrackOptionsC, TrackEstablishmentOptions())):
        ref = ctype()
        default_fn(C.byref(ref))
        want, got = fields(ref), fields(pyobj.to_c())
        assert want.keys() == got.keys()
        for k in want:
            assert want[k] == got[k], (ctype.__name__, k, want[k], got[k])


def test_block_mt19937_reproduces_the_standard_library_stream(lib):
    """Global positioning's random start (gp.cc:135,261,449: std::mt19937 + std::uniform_real_distribution<double>(-1, 1))
    is drawn by a block generator (csrc/mt19937.hpp); host-only self test of the library: the two streams are bit-identical,
    across state refills, odd phases and discards."""
    import ctypes as C

    import numpy as np

    for seed, skip, count in ((1, 0, 5000), (1, 7, 4001), (12345, 623, 1300), (0, 6 * 1000003, 2000)):
        a, b = np.empty(count), np.empty(count)
        rc = lib.gsfm_selftest_mt19937(C.c_uint32(seed), C.c_uint64(skip), C.c_int64(count), C.c_double(100.0),
                                       a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0 and np.array_equal(a, b) and np.abs(a).max() <= 100.0 and a.std() > 50.0
