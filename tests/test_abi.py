"""The C-ABI library loads and exports every symbol include/sthip.h declares (no GPU needed)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sthip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sthip_[a-z_]+)\s*\(", src)))


def test_header_and_loader_agree(built):
    from stratum_amd import _lib

    assert declared_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(built):
    from stratum_amd import _lib

    L = _lib.lib()
    for name in declared_functions():
        assert hasattr(L, name), name
    assert L.sthip_abi_version() == 11


def test_wire_struct_sizes():
    from stratum_amd import wire

    assert C.sizeof(wire.BDPTPushConstants) == 96
    assert wire.InstanceData.itemsize == 16
    assert wire.PackedVertexData.itemsize == 32
    assert wire.TransformData.itemsize == 48
    assert wire.ViewData.itemsize == 48
    assert wire.MaterialRecord.itemsize == 72
    assert wire.ShadingData.itemsize == 48


def test_no_device_fails_loudly(built):
    """Without a HIP device the product refuses to work: there is no CPU fallback to fall into."""
    import torch

    from stratum_amd import _lib

    if torch.cuda.is_available():
        return  # on the GPU box this path cannot be exercised
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.sthip_create(0, C.byref(h))
    assert rc == -2 and not h.value  # STHIP_ERR_NO_DEVICE
    assert b"no HIP device" in L.sthip_last_error(None)


def test_product_never_touches_the_oracle():
    """Nothing under stratum_amd/ may import, include or link anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "stratum_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|oracle_py|liboracle|#include\s+\"[^\"]*oracle", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
