"""CPU: the C-ABI library loads, exports every symbol include/bhray.h declares, and its layouts are the
reference's #[repr(C)] layouts.  No compute calls (there is no GPU here and no CPU path in the library)."""
import ctypes as C
import os
import re

import pytest

import bhusie_amd as B
from bhusie_amd import layouts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "bhray.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(bhray_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound():
    L = B.lib()
    declared = header_functions()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/bhray.h but not exported by libbhray.so"
    assert sorted(layouts.SYMBOLS) == declared, "python bindings and header disagree"


def test_layout_sizes_match_reference_structs():
    # ray_pipeline.rs:3-14, camera.rs:66-73, blackhole.rs:37-51, triangle.rs:45-63, 268-285
    assert C.sizeof(layouts.BhrayDetails) == 32
    assert C.sizeof(layouts.BhrayCameraUniform) == 32
    assert C.sizeof(layouts.BhrayBlackHoleUniform) == 132
    assert C.sizeof(layouts.BhrayNode) == 32 and C.sizeof(layouts.BhrayTriangle) == 24
    assert layouts.BhrayBlackHoleUniform.position.offset == 16
    assert layouts.BhrayBlackHoleUniform.normal.offset == 32
    assert layouts.BhrayBlackHoleUniform.rotation_matrix.offset == 48
    assert layouts.BhrayBlackHoleUniform.feather_amount.offset == 96
    assert layouts.BhrayCameraUniform.forward.offset == 16 and layouts.BhrayCameraUniform.fov.offset == 28
    assert layouts.MODEL_UNIFORM_BYTES == 48 + 92 * 524288 + 28 == 48234572
    hdr = open(os.path.join(ROOT, "include", "bhray.h")).read()
    assert "48234572u" in hdr


def test_reference_ladder_rule():
    cfg = B.ladder_from_base((72, 41), 3, 4)              # mod.rs:177-205
    assert cfg.sizes() == [(72, 41), (214, 121), (640, 361), (1918, 1081)]
    assert (cfg.crop_x, cfg.crop_y, cfg.frame_w, cfg.frame_h) == (0, 0, 1918, 1081)
    assert B.ladder_for_frame((1918, 1081), 3, 4).sizes() == cfg.sizes()
    c2 = B.ladder_for_frame((1920, 1080), 3, 4)
    assert c2.sizes() == [(73, 41), (217, 121), (649, 361), (1945, 1081)]
    assert (c2.crop_x, c2.crop_y, c2.frame_w, c2.frame_h) == (12, 0, 1920, 1080)
    for (w0, h0), (w1, h1) in zip(c2.sizes(), c2.sizes()[1:]):
        assert w1 - 1 == 3 * (w0 - 1) and h1 - 1 == 3 * (h0 - 1)
    c4 = B.ladder_for_frame((3840, 2160), 3, 4)
    assert c4.sizes()[-1][0] >= 3840 and c4.sizes()[-1][1] >= 2160
    c8 = B.ladder_for_frame((7680, 4320), 3, 5)
    assert c8.sizes()[-1][0] >= 7680 and c8.levels == 5


def test_errors_are_codes_not_crashes():
    L = B.lib()
    assert L.bhray_strerror(0) == b"ok"
    assert L.bhray_create(None, None) == -1
    cfg = B.ladder_from_base((72, 41), 3, 4)
    cfg.struct_size = 4
    h = C.c_void_p()
    assert L.bhray_create(C.byref(cfg), C.byref(h)) == -1
    assert b"struct_size" in L.bhray_last_error(None)
    bad = layouts.BhrayConfig()
    assert L.bhray_ladder_from_base(1, 1, 3, 4, C.byref(bad)) == -1
    assert L.bhray_ladder_from_base(72, 41, 3, 99, C.byref(bad)) == -1


def test_no_device_is_a_loud_error_not_a_fallback():
    """On a box without a GPU bhray_create must fail with BHRAY_E_NO_DEVICE (never render on the CPU)."""
    if B.lib().bhray_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(B.BhrayError) as e:
        B.RayPass(B.ladder_from_base((72, 41), 3, 1))
    assert e.value.code == -2


def test_product_does_not_touch_the_oracle():
    """libbhray and the bhusie_amd package must not link, load or import anything under oracle/."""
    import subprocess
    so = os.path.join(ROOT, "bhusie_amd", "libbhray.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bhusie_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_cpp_host_program_is_built_and_fails_loudly_without_a_gpu(tmp_path):
    import subprocess
    exe = os.path.join(ROOT, "bhusie_amd", "bhray_render")
    assert os.path.exists(exe), "make -C bhusie_amd/csrc builds the C++ host program"
    if B.lib().bhray_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, str(tmp_path / "o.f32")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no HIP device" in r.stderr


def _header_struct_fields(name):
    """[(field, ctype, array_len or None)] of `typedef struct name { ... } name;` in include/bhray.h"""
    text = open(os.path.join(ROOT, "include", "bhray.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    macros = dict(re.findall(r"#define (BHRAY_[A-Z_]+)\s+(\d+)", text))
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"([a-z0-9_]+)\s+(.*)$", decl)
        ctype, rest = m.group(1), m.group(2)
        for item in rest.split(","):
            item = item.strip()
            a = re.match(r"([a-zA-Z0-9_]+)\[([A-Z0-9_ +]+)\]$", item)
            if a:                                             # a length is a macro, a number, or a sum of those
                n = sum(int(macros.get(t.strip(), t.strip())) for t in a.group(2).split("+"))
                out.append((a.group(1), ctype, n))
            else:
                out.append((item, ctype, None))
    return out


def test_integration_md_rust_binding_matches_the_header():
    """INTEGRATION.md holds the Rust side a maintainer adds (it cannot be compiled here: no Rust toolchain).  Keep it in lock step:
    bhray_config field for field (name, type, array length, order), and every extern fn it declares exists in the header with the
    same number of parameters."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = re.search(r"pub struct bhray_config \{(.*?)\n\}", md, flags=re.S).group(1)
    rfields = []
    for name, ty in re.findall(r"pub ([a-z0-9_]+): (\[[a-z0-9]+; \d+\]|[a-z0-9]+)", rust):
        a = re.match(r"\[([a-z0-9]+); (\d+)\]", ty)
        rfields.append((name, a.group(1), int(a.group(2))) if a else (name, ty, None))
    cmap = {"uint32_t": "u32", "int32_t": "i32", "uint8_t": "u8", "float": "f32", "uint64_t": "u64"}
    hfields = [(n, cmap[t], k) for n, t, k in _header_struct_fields("bhray_config")]
    assert rfields == hfields, "INTEGRATION.md bhray_config differs from include/bhray.h"
    def py_field(t):
        n = getattr(t, "_length_", None)
        base = t._type_ if n is not None else t
        return {C.c_uint32: "u32", C.c_int32: "i32", C.c_uint8: "u8"}[base], n
    assert hfields == [(n,) + py_field(t) for n, t in layouts.BhrayConfig._fields_], "bhusie_amd/layouts.py BhrayConfig differs from include/bhray.h"
    hdr = open(os.path.join(ROOT, "include", "bhray.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    ext = re.search(r'extern "C" \{(.*?)\n\}', md, flags=re.S).group(1)
    fns = re.findall(r"pub fn (bhray_[a-z0-9_]+)\((.*?)\)", ext)
    assert len(fns) >= 10
    for name, params in fns:
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, hdr, flags=re.S)
        assert m, f"{name} is declared in INTEGRATION.md but not in include/bhray.h"
        nh = 0 if m.group(1).strip() in ("", "void") else m.group(1).count(",") + 1
        nr = 0 if not params.strip() else params.count(",") + 1
        assert nh == nr, f"{name}: {nr} parameters in the Rust declaration, {nh} in the header"


def test_header_layout_asserts_are_compiled_into_the_library():
    """include/bhray.h's offsets are static_assert-ed in bhusie_amd/csrc/bhray_layout.cpp, which the Makefile builds into libbhray.so."""
    src = open(os.path.join(ROOT, "bhusie_amd", "csrc", "bhray_layout.cpp")).read()
    mk = open(os.path.join(ROOT, "bhusie_amd", "csrc", "Makefile")).read()
    assert "bhray_layout.cpp" in mk and "bhray_layout.o" in mk
    for sym in ("BHRAY_MODEL_OFF_POINTS", "BHRAY_MODEL_OFF_NORMALS", "BHRAY_MODEL_OFF_TRIANGLES", "BHRAY_MODEL_OFF_NODES",
                "BHRAY_MODEL_OFF_LOOKUP", "BHRAY_MODEL_UNIFORM_BYTES", "bhray_node", "bhray_triangle", "bhray_model_header",
                "bhray_black_hole_uniform", "bhray_camera_uniform", "bhray_details"):
        assert sym in src, sym
    assert src.count("static_assert") + src.count("OFF(") + src.count("SZ(") >= 60
