"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/rvio_hip.h
declares, its POD layouts match the ctypes mirror, and it fails loudly without a GPU.
No compute is launched here."""
import ctypes as C
import os
import re

import pytest

import oracle as O

abi, rv = O.abi, O.rv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from rvio_amd import build, hip
    build.build()
    return hip.load()


def test_every_declared_symbol_is_exported(lib):
    from rvio_amd import hip
    hdr = open(os.path.join(ROOT, "include", "rvio_hip.h")).read()
    declared = set(re.findall(r"\b(rvio_(?:hip_)?[a-z_]+)\s*\(", hdr))
    declared -= {"rvio_hip"}  # the opaque struct tag
    assert declared, "header parse failed"
    assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_abi_version_and_struct_sizes(lib):
    assert lib.rvio_hip_abi_version() == abi.ABI_VERSION
    assert C.sizeof(abi.rvio_imu) == 64
    assert C.sizeof(abi.rvio_frame_info) == 64
    assert C.sizeof(abi.rvio_tracks) == 32
    assert C.sizeof(abi.rvio_config) == 312


def test_config_euroc_matches_python_mirror(lib):
    c = abi.rvio_config()
    lib.rvio_config_euroc(C.byref(c))
    assert bytes(c) == bytes(abi.config_euroc())
    assert c.n_features == 200 and c.max_track_len == 15 and abs(c.gravity - 9.8082) < 1e-12


def test_create_rejects_bad_configs(lib):
    h = C.c_void_p()
    bad = abi.config_euroc(enable_equalizer=0, max_track_len=40)
    assert lib.rvio_hip_create(C.byref(bad), 0, C.byref(h)) == -1
    assert lib.rvio_hip_create(None, 0, C.byref(h)) == -1


def test_product_path_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rvio_amd import hip
    with pytest.raises(hip.RvioHipError):
        hip.RvioHip(abi.config_named("B", enable_equalizer=0))


def test_product_never_touches_the_oracle():
    """the product package must not import, link or call anything under oracle/"""
    for pkg in (os.path.join(ROOT, "r-vio_amd"), os.path.join(ROOT, "host"), os.path.join(ROOT, "include")):
      for dp, _, fs in os.walk(pkg):
          for f in fs:
              if f.endswith((".py", ".hip", ".h", ".cpp")):
                  txt = open(os.path.join(dp, f)).read()
                  code = "\n".join(ln for ln in txt.splitlines() if not ln.lstrip().startswith(("//", "#", "*", "/*", '"""')))
                  assert "liborc" not in txt and not re.search(r"\borc_[a-z_]+\s*\(", txt), os.path.join(dp, f)
                  assert not re.search(r"^\s*(import|from)\s+oracle\b", code, re.M), os.path.join(dp, f)
                  assert not re.search(r"#include\s*[\"<][^\">]*oracle", txt), os.path.join(dp, f)


def test_bench_touches_the_oracle_only_in_its_cpu_baseline_leg():
    """bench.py may use oracle/ for the reported CPU baseline, never for the thing measured"""
    txt = open(os.path.join(ROOT, "bench.py")).read()
    a = txt.index("def cpu_baseline(")
    b = txt.index("\nif __name__", a)
    outside = txt[:a] + txt[b:]
    assert not re.search(r"^\s*(import|from)\s+oracle\b", outside, re.M)
    assert "liborc" not in outside.replace("oracle/liborc.so (g++ -O3, single thread)", "")
    assert re.search(r"^\s*import oracle as O", txt[a:b], re.M)
