"""C-ABI checks that need no GPU: the library loads, exports every symbol include/rmd_api.h
declares, mirrors the reference struct layouts, and validates arguments before touching HIP."""
import ctypes as C
import os
import re

import pytest


def header_functions(root):
    text = open(os.path.join(root, "include", "rmd_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rmd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rmd):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = header_functions(root)
    assert len(declared) >= 40
    raw = C.CDLL(rmd.LIB_PATH)
    missing = [n for n in declared if not hasattr(raw, n)]
    assert not missing, missing
    from raymarchdenoisercuda_amd._lib import SYMBOLS
    assert sorted(SYMBOLS) == declared           # the Python binding table tracks the header


def test_struct_layouts_match_the_reference(rmd):
    # reference include/gbuffer.h:6-14 and include/filter.cuh:11-23
    assert C.sizeof(rmd.GBuffer) == 56 and C.alignment(rmd.GBuffer) == 8
    assert [getattr(rmd.GBuffer, f).offset for f in ("shape", "render", "denoised", "normal", "albedo", "buffer")] == [0, 8, 16, 24, 32, 40]
    assert C.sizeof(rmd.FilterParams) == 36 and C.alignment(rmd.FilterParams) == 4
    offs = [getattr(rmd.FilterParams, f).offset for f in ("type", "depth", "level", "radius", "sigmaSpace", "sigmaColor",
                                                           "sigmaAlbedo", "sigmaNormal", "cacheInput", "cacheBuffer")]
    assert offs == [0, 4, 8, 12, 16, 20, 24, 28, 32, 33]
    p = rmd.FilterParams()
    assert p.cacheInput == 1 and p.cacheBuffer == 1 and p.type == rmd.FilterParams.AVERAGE


def test_python_binding_matches_the_c_layout_of_the_header(rmd, orc):
    """raymarchdenoisercuda_amd/_lib.py against include/rmd_api.h as a C compiler lays it out (oracle/abi_probe.c:
    offsetof / sizeof of every field): same field names in the same order at the same offsets with the same sizes.
    The oracle's front end builds its structures from that report, not from the binding (tests/oracle_lib.py)."""
    from raymarchdenoisercuda_amd import _lib
    pairs = {"rmd_int2": _lib.Int2, "rmd_gbuffer": _lib.GBuffer, "rmd_filter_params": _lib.FilterParams,
             "rmd_svgf_params": _lib.SvgfParams, "rmd_svgf_frame_desc": _lib.SvgfFrameDesc, "rmd_strip_plan": _lib.StripPlan,
             "rmd_halo_step": _lib.HaloStep, "rmd_synth_desc": _lib.SynthDesc}
    assert sorted(pairs) == sorted(orc.STRUCTS)
    kinds = {C.c_int: "i", C.c_uint32: "u", C.c_float: "f", C.c_ubyte: "b", C.c_void_p: "p"}
    struct, seen = None, {}
    for line in orc.ABI_LAYOUT.splitlines():
        parts = line.split()
        if line[0] != " ":
            struct = parts[0]
            assert C.sizeof(pairs[struct]) == int(parts[1]) and C.alignment(pairs[struct]) == int(parts[2]), struct
            seen[struct] = []
            continue
        name, off, size, kind = parts[0], int(parts[1]), int(parts[2]), parts[3]
        seen[struct].append(name)
        field = getattr(pairs[struct], name)
        assert (field.offset, field.size) == (off, size), f"{struct}.{name}: binding {(field.offset, field.size)} vs C {(off, size)}"
        ctype = dict(pairs[struct]._fields_)[name]
        ctype = getattr(ctype, "_type_", ctype) if hasattr(ctype, "_length_") else ctype
        if kind.startswith("s:"):
            assert ctype is pairs[kind[2:]], f"{struct}.{name}"
        else:
            assert kinds[ctype] == kind, f"{struct}.{name}: binding {ctype} vs C kind {kind}"
    for struct, names in seen.items():
        assert names == [n for n, _ in pairs[struct]._fields_], struct


def test_default_params_match_appendix_a(rmd, orc):
    a, b = rmd.default_params(), orc.default_params()
    for name, _ in a._fields_:
        assert getattr(a, name) == getattr(b, name), name


def test_reach_of_a_frame(rmd):
    p = rmd.default_params()
    assert rmd.svgf.frame_reach(p) == (66, 65 + p.max_motion_rows, 60, 65)
    p.iterations = 1
    assert rmd.svgf.frame_reach(p)[2] == 0


def test_argument_errors_are_reported_without_a_gpu(rmd):
    g = rmd.GBuffer()
    g.shape = rmd.Int2(0, 10)
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelBaseline(g, rmd.FilterParams())
    assert e.value.code == -2
    g.shape = rmd.Int2(8, 8)
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelTiled(g, rmd.FilterParams())
    assert e.value.code == -1 and "NULL" in str(e.value)
    g.render, g.denoised = 4096, 8192
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelTiled(g, rmd.FilterParams(depth=2))
    assert e.value.code == -4
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelTiled(g, rmd.FilterParams(radius=-1))
    assert e.value.code == -3
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelTiled(g, rmd.FilterParams(type=7))                       # not a FilterType
    assert e.value.code == -3
    with pytest.raises(rmd.RmdError) as e:
        rmd.filterKernelTiled(g, rmd.FilterParams(type=rmd.FilterParams.GAUSSIAN, sigmaSpace=0.0))
    assert e.value.code == -3
    d = rmd.SvgfFrameDesc()
    d.width, d.height, d.buf_row0, d.buf_rows = 16, 16, 4, 20
    with pytest.raises(rmd.RmdError) as e:
        rmd.svgf.temporal(d, rmd.default_params(), 0, 16)
    assert e.value.code == -5


def test_no_product_module_touches_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "raymarchdenoisercuda_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f), errors="replace").read()
                # comments may cite the oracle, code may not use it
                if f.endswith(".py"):
                    code = re.sub(r"#[^\n]*", "", text)
                else:
                    code = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))
                assert "liboracle" not in code and "oracle_lib" not in code and "oracle/" not in code, os.path.join(base, f)
                assert not re.search(r"\borc_[a-z]", code), os.path.join(base, f)
