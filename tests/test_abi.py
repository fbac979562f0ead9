"""The drop-in boundary: libcrychic_hip.so loads, exports every symbol include/crychic_hip.h declares, keeps the
reference's data ABI byte for byte, and fails loudly -- not silently on the CPU -- when no HIP
device is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "crychic_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crychic_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    names = declared_symbols()
    assert len(names) >= 20
    raw = C.CDLL(built_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libcrychic_hip.so does not export %s" % n
        assert n in built_lib.PROTOTYPES, "python binding lacks a prototype for %s" % n
    assert sorted(built_lib.PROTOTYPES) == names


def test_data_abi_layout(built_lib):
    L, P, S = built_lib.Light, built_lib.PassConstants, built_lib.SsaoConstants
    assert C.sizeof(L) == 48 and L.FalloffStart.offset == 12 and L.Direction.offset == 16 and L.FalloffEnd.offset == 28
    assert L.Position.offset == 32 and L.SpotPower.offset == 44
    assert C.sizeof(P) == 2048
    exp = {"View": 0, "InvView": 64, "Proj": 128, "InvProj": 192, "ViewProj": 256, "InvViewProj": 320, "ViewProjTex": 384,
           "ShadowTransforms": 448, "EyePosW": 1216, "cbPerObjectPad1": 1228, "RenderTargetSize": 1232,
           "InvRenderTargetSize": 1240, "NearZ": 1248, "FarZ": 1252, "TotalTime": 1256, "DeltaTime": 1260,
           "AmbientLight": 1264, "Lights": 1280}
    for k, v in exp.items():
        assert getattr(P, k).offset == v, k
    assert C.sizeof(S) == 496
    exp = {"Proj": 0, "InvProj": 64, "ProjTex": 128, "OffsetVectors": 192, "BlurWeights": 416, "RenderTargetSize": 464,
           "InvRenderTargetSize": 472, "OcclusionRadius": 480, "OcclusionFadeStart": 484, "OcclusionFadeEnd": 488,
           "SurfaceEpsilon": 492}
    for k, v in exp.items():
        assert getattr(S, k).offset == v, k


def test_no_device_fails_loudly(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = built_lib.lib.crychic_ctx_create(0, C.byref(h))
    assert rc == -2 and not h.value
    assert b"no CPU fallback" in built_lib.lib.crychic_last_error()
    with pytest.raises(built_lib.CrychicError):
        built_lib.check(rc)
    # compute entry points refuse a null context instead of computing anything on the host
    assert built_lib.lib.crychic_ssao(None, None, None, None, None, None, None, 64, 64, 0, 32, None) < 0


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import, link or execute oracle/."""
    use = re.compile(r"liboracle|crychic_oracle\.h|or_math\.h|or_samplers\.h|oracle_lib|hostsim_lib|libhostsim|import\s+oracle|from\s+oracle|"
                     r"[\"']oracle[\"'/]|\bor_[a-z0-9_]+\s*\(")
    for sub in ("crychic_renderer_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                    text = open(os.path.join(dp, f), errors="replace").read()
                    m = use.search(text)
                    assert not m, "%s references the oracle: %r" % (os.path.join(dp, f), m.group(0))
    # bench.py may only reach it inside cpu_baseline()
    bench = open(os.path.join(ROOT, "bench.py")).read()
    body = bench[bench.index("def cpu_baseline"):bench.index("def main")]
    rest = bench.replace(body, "")
    assert "oracle_lib" in body and "oracle_lib" not in rest and "liboracle" not in rest


def test_strip_rows(built_lib):
    lib = built_lib.lib
    for H, n in ((2160, 8), (2160, 4), (2160, 2), (2160, 1), (4320, 8), (256, 3), (34, 8)):
        got, nxt = [], 0
        for r in range(n):
            a, b = C.c_uint32(), C.c_uint32()
            assert lib.crychic_strip_rows(H, n, r, C.byref(a), C.byref(b)) == 0
            assert a.value == nxt and a.value % 2 == 0
            nxt = a.value + b.value
            got.append(b.value)
        assert nxt == H
        assert all(g % 2 == 0 for g in got)
    a, b = C.c_uint32(), C.c_uint32()
    assert lib.crychic_strip_rows(2161, 8, 0, C.byref(a), C.byref(b)) == -1
    assert lib.crychic_strip_rows(2160, 8, 8, C.byref(a), C.byref(b)) == -1
