"""The drop-in boundary: libdsrc_gpu.so loads and exports exactly what include/dsrc_gpu.h declares
(no compute calls here -- this runs without a GPU), and the product never reaches for a CPU codec."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "dsrc_gpu.h")).read()
    return sorted(set(re.findall(r"\b(dsrcgpu_[a-z_0-9]+)\s*\(", src)))


def test_header_and_library_agree():
    import __graft_entry__ as g
    path = g.build_gpu_lib()
    lib = C.CDLL(path)
    names = declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dsrc_gpu.h but not exported"
    from dsrc_amd import _lib
    assert sorted(_lib.EXPORTS) == names


def test_exports_are_c_abi():
    import __graft_entry__ as g
    out = subprocess.check_output(["nm", "-D", "--defined-only", g.build_gpu_lib()], text=True)
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert all(not s.startswith("_Z") or "dsrcgpu" not in s for s in syms)
    assert set(declared()) <= set(syms)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "dsrc_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"liboracle|dsrc_oracle\.h|orc_compress|tests\._oracle|libdsrc_ref", txt):
                    bad.append(f)
    assert not bad, bad


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    os.environ.pop("DSRC_GPU_LIB", None)
    from dsrc_amd import _lib
    _lib._lib = None
    try:
        _lib.Handle()
    except _lib.DsrcGpuError as e:
        assert "no CPU fallback" in str(e) or "HIP" in str(e)
    else:
        raise AssertionError("handle creation must fail without a GPU")
