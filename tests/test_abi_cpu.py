"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every
symbol include/metacache_amd.h declares.  No compute calls here (no GPU)."""
import ctypes
import os
import re

from metacache_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "metacache_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    path = build.build_library()
    L = ctypes.CDLL(path)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in metacache_amd.h but not exported"


def test_python_binding_lists_the_same_symbols():
    assert sorted(api.EXPORTS) == declared_functions()


def test_no_gpu_means_loud_failure():
    """Without a usable device mc_create must fail with MC_ERR_HIP -- never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        return
    cfg = api.default_config()
    h = ctypes.c_void_p()
    rc = api.lib().mc_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -2 and not h.value
    assert b"no CPU fallback" in api.lib().mc_last_error(None)


def test_no_gpu_means_loud_failure_for_the_multi_gpu_drivers(golden):
    """mc_partset_open / mc_keyset_open (parts / key shards over GPUs) fail loudly as well; bad arguments are refused before any device work"""
    import torch
    from metacache_amd.api import McError
    if torch.cuda.is_available():
        return
    for cls, kw in ((api.PartSet, dict(resident=1)), (api.KeySet, dict(shards=2))):
        try:
            cls(golden.db_path("toy32"), **kw)
        except McError as e:
            assert "no usable HIP device" in str(e) or "no CPU fallback" in str(e), str(e)
        else:
            raise AssertionError(f"{cls.__name__} opened without a GPU")
    L = api.lib()
    L.mc_keyset_open.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    assert L.mc_keyset_open(None, None, 0, None, 0, None) == -1                       # MC_ERR_INVALID
    L.mc_keyset_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert L.mc_keyset_info(None, None) == -1
    L.mc_keyset_close.argtypes = [ctypes.c_void_p]
    L.mc_keyset_close(None)                                                           # a no-op


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "metacache_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "mc_oracle" not in src and "libmcref" not in src and "oracle/" not in src, os.path.join(dp, f)
