"""CPU suite: the C-ABI library loads and exports exactly what include/adaptpoint_amd.h
declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "adaptpoint_amd.h")


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"APN_API\s+(?:const\s+char\s*\*|int)\s*(apn_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        protos[m.group(1)] = args
    return protos


@pytest.fixture(scope="module")
def lib():
    from adaptpoint_amd import build as apn_build
    path = apn_build.build()           # hipcc cross-compiles gfx950 without a GPU
    return ctypes.CDLL(path)


def test_header_declares_the_nine_reference_entry_points():
    protos = header_prototypes()
    for name in ["apn_ball_query", "apn_group_points", "apn_group_points_grad", "apn_gather_points",
                 "apn_gather_points_grad", "apn_furthest_point_sampling", "apn_three_nn",
                 "apn_three_interpolate", "apn_three_interpolate_grad"]:
        assert name in protos
        assert protos[name][-1].replace(" ", "") == "void*stream"


def test_library_exports_every_declared_symbol(lib):
    for name in header_prototypes():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctypes_signatures_match_header_arity():
    from adaptpoint_amd import _lib
    protos = header_prototypes()
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in protos, name
        assert len(argtypes) == len(protos[name]), name
        for at, decl in zip(argtypes, protos[name]):
            if "*" in decl:
                assert at is ctypes.c_void_p, (name, decl)
            elif decl.startswith("float"):
                assert at is ctypes.c_float, (name, decl)
            elif decl.startswith("double"):
                assert at is ctypes.c_double, (name, decl)
            elif decl.startswith("size_t"):
                assert at is ctypes.c_size_t, (name, decl)
            elif decl.startswith("long long"):
                assert at is ctypes.c_longlong, (name, decl)
            else:
                assert at is ctypes.c_int, (name, decl)


def test_version_and_error_strings(lib):
    lib.apn_version.restype = ctypes.c_int
    assert lib.apn_version() >= 100
    lib.apn_error_string.restype = ctypes.c_char_p
    assert lib.apn_error_string(0) == b"success"
    assert b"invalid" in lib.apn_error_string(-1)


def test_library_carries_its_flag_line_and_the_loader_checks_it(lib, monkeypatch):
    """a library built with the vectoriser back on (or contraction on) must not load silently.  build.py
    stamps the flag line into the library, refuses flag sets that undo the two correctness flags, treats an object whose
    recorded flags differ from today's as stale; _lib.load() refuses a library whose line lacks them."""
    from adaptpoint_amd import _lib, build as apn_build
    lib.apn_build_flags.restype = ctypes.c_char_p
    flags = lib.apn_build_flags().decode().split()
    assert "-fno-slp-vectorize" in flags and "-ffp-contract=off" in flags and "-O3" in flags
    assert _lib.REQUIRED_BUILD_FLAGS == apn_build.REQUIRED_FLAGS
    obj = os.path.join(apn_build.OBJ, "fps.o")
    assert apn_build._same_flags(obj, os.path.join(apn_build.CSRC, "fps.hip"))
    monkeypatch.setattr(apn_build, "CXXFLAGS", apn_build.CXXFLAGS + ["-DAPN_SOMETHING_ELSE"])
    assert not apn_build._same_flags(obj, os.path.join(apn_build.CSRC, "fps.hip"))       # changed flags: stale object
    monkeypatch.setattr(apn_build, "CXXFLAGS", apn_build.CXXFLAGS + ["-fslp-vectorize"])
    with pytest.raises(RuntimeError, match="refused"):
        apn_build.build()
    monkeypatch.setattr(apn_build, "CXXFLAGS", [f for f in apn_build.CXXFLAGS if "slp" not in f])
    with pytest.raises(RuntimeError, match="refused"):
        apn_build.check_flags()


def test_argument_validation_needs_no_gpu(lib):
    """Bad sizes are rejected before any HIP call; zero sizes are no-ops."""
    f = lib.apn_furthest_point_sampling
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
    assert f(-1, 8, 4, None, None, None, None) == -1
    assert f(0, 8, 4, None, None, None, None) == 0
    assert f(2, 8, 0, None, None, None, None) == 0          # m <= 0: the reference kernel returns at once
    assert f(2, 0, 4, None, None, None, None) == -1
    assert f(2, 8, 4, None, None, None, None) == -1          # null pointers
    g = lib.apn_group_points
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4
    assert g(1, 4, 8, 0, 3, None, None, None, None) == 0
    assert g(1, 4, 8, 1 << 20, 1 << 20, None, None, None, None) == -1   # npoints*nsample overflows int
    t = lib.apn_furthest_point_sampling_tuned
    t.restype = ctypes.c_int
    t.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
    assert t(2, 8, 0, None, None, None, 3, 0, None) == 0      # m <= 0 returns before anything is read
    assert t(2, 8, 4, None, None, None, 0, 0, None) == -1
    assert not hasattr(lib, "apn_fps_set_waves") and not hasattr(lib, "apn_fps_set_algo")   # no process-wide state


def test_argument_validation_of_the_adaptpoint_entries_needs_no_gpu():
    """The per-point layer, spectral-norm, anchor-transform and deformation entries reject bad arguments (and accept
    empty work) before any HIP call, like the reference-boundary entries above."""
    from adaptpoint_amd import _lib
    lib = _lib.load()
    EINVAL = -1
    assert lib.apn_pw_conv_tiles(4, 1000) == 4 * 8
    assert lib.apn_pw_conv_forward(2, 8, 8, 16, 5, None, None, None, None, None) == EINVAL        # precision 2 or 3
    assert lib.apn_pw_conv_forward(0, 8, 8, 16, 3, None, None, None, None, None) == 0             # no clouds
    assert lib.apn_pw_conv_forward(2, 8, 8, 16, 3, None, None, None, None, None) == EINVAL        # null tensors
    assert lib.apn_pw_bn_act(2, 0, 16, None, None, 0, None, None, 1e-5, 0.1, 1, 1, None, None, None, None, None, None) == EINVAL
    assert lib.apn_pw_bn_act(0, 8, 16, None, None, 0, None, None, 1e-5, 0.1, 1, 1, None, None, None, None, None, None) == 0
    assert lib.apn_pw_conv_grad_weight(2, 8, 8, 16, 3, None, None, None, None, None) == EINVAL
    assert lib.apn_pw_conv_max_backward(2, 129, 8, 16, None, None, None, None, None, 1, None, None, None, None, None) == EINVAL
    assert lib.apn_pw_contract(1, 4, 4, 4, None, 0, 4, 1, None, 0, 4, 1, None, 0, 4, 0, None, 3, None) == EINVAL
    # split-K shares: short contractions down to two chunks per share, long ones at least eight, <= 256 workgroups (one per CU)
    assert lib.apn_pw_contract_splits(1, 256, 256, 512) == 8 and lib.apn_pw_contract_splits(32, 512, 1536, 256) == 5
    assert lib.apn_spectral_norm(0, 4, None, 1, 1e-12, None, None, None, None, None, None, None, None) == EINVAL
    assert lib.apn_spectral_norm_blocks(1024, 512) == 512 and lib.apn_spectral_norm_blocks(15, 1) == 1
    assert lib.apn_anchor_transforms(0, None, None, None, 10.0, 3.0, 0.25, None, None, None) == 0
    assert lib.apn_anchor_transforms(4, None, None, None, 10.0, 3.0, 0.25, None, None, None) == EINVAL
    assert lib.apn_deform_forward(2, 5000, 4, None, None, None, None, None, None, 0.5, None, None, None, None) == EINVAL   # n <= 4096
    assert lib.apn_deform_forward(2, 1024, 9, None, None, None, None, None, None, 0.5, None, None, None, None) == EINVAL   # m <= 8
    assert lib.apn_deform_forward(0, 1024, 4, None, None, None, None, None, None, 0.5, None, None, None, None) == 0
