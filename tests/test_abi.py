"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/orbslam_amd.h declares,
fails loudly without a device, and the C++ drop-in headers compile against it."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "orbslam_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    names = re.findall(r"\b((?:orb|ba)_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from orbslam_in_practice_amd import _lib
    L = _lib.lib()
    decl = declared_functions()
    assert len(decl) >= 30
    missing = [n for n in decl if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == decl, set(decl) ^ set(_lib.EXPORTS)


def test_no_device_fails_loudly():
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import _lib
    L = _lib.lib()
    if L.orb_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(oa.OrbError) as ei:
        oa.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ei.value.code == _lib.ORB_E_NO_DEVICE
    with pytest.raises(oa.OrbError):
        oa.ORBmatcher(0.9, True)
    with pytest.raises(oa.OrbError):
        oa.Optimizer()


def test_host_side_helpers_need_no_device(oracle):
    """DescriptorDistance / ComputeThreeMaxima are host functions of the C-ABI (reference
    src/ORBmatcher.cpp:128-188) and must agree with the oracle"""
    from orbslam_in_practice_amd import _lib, synth
    L = _lib.lib()
    d = synth.random_descriptors(1, 64)
    for i in range(63):
        assert L.orb_descriptor_distance(_lib.ptr(d[i]), _lib.ptr(d[i + 1])) == oracle.descriptor_distance(d[i], d[i + 1])
        assert L.orb_descriptor_distance(_lib.ptr(d[i]), _lib.ptr(d[i + 1])) == \
            int(np.unpackbits(d[i] ^ d[i + 1]).sum())
    rng = np.random.default_rng(5)
    for _ in range(100):
        h = rng.integers(0, 30, 30).astype(np.int32)
        a, b, c = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        L.orb_compute_three_maxima(_lib.ptr(h), 30, C.byref(a), C.byref(b), C.byref(c))
        assert (a.value, b.value, c.value) == oracle.compute_three_maxima(h)


def test_product_never_touches_the_oracle():
    """the shipped package / C sources / headers must not include, import, link or dlopen anything under
    oracle/ (comments may cite it)"""
    bad = [re.compile(r'#\s*include\s*[<"][^>"]*oracle'), re.compile(r"^\s*(from|import)\s+oracle", re.M),
           re.compile(r"liborbslam_oracle"), re.compile(r"-lorbslam_oracle"), re.compile(r"orc_[a-z_]+\s*\(")]
    roots = [os.path.join(ROOT, "orbslam_in_practice_amd"), os.path.join(ROOT, "include")]
    for root in roots:
        for dirpath, _, files in os.walk(root):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".inc", "Makefile", ".cpp")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    for rx in bad:
                        assert not rx.search(txt), (os.path.join(dirpath, f), rx.pattern)


def test_cpp_dropin_headers_compile(tmp_path):
    """the reference-signature C++ classes over the C-ABI compile WITH every template instantiated
    (tests/cpp/dropin_main.cpp calls each public method) and link against the library; -Wall -Wextra clean"""
    from tests.test_gpu_dropin import build_dropin
    exe = build_dropin(str(tmp_path), extra=["-Wall", "-Wextra", "-Werror"])
    assert os.path.exists(exe)
    # ... and the Tracking + LocalMapping style loop written on them (tests/cpp/tracking_main.cpp)
    assert os.path.exists(build_dropin(str(tmp_path), extra=["-O1", "-Wall", "-Wextra", "-Werror"], program="tracking_main"))


def test_dropin_headers_standalone():
    """without a Frame.h / KeyFrame.h on the include path the headers still compile: the templates
    (ORBmatcherT<FrameT>, OptimizerT) are the whole interface then"""
    src = "#include <orbslam_amd/ORBextractor.h>\n#include <orbslam_amd/ORBmatcher.h>\n#include <orbslam_amd/Optimizer.h>\n" \
          "int main() { ORBSlam::Optimizer::InvLevelSigma2(); return 0; }\n"
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-"],
                   input=src.encode(), check=True)
