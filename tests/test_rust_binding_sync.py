"""The Rust binding (bindings/rust/codec-eval-hip) cannot be compiled in this image (no Rust toolchain); this keeps
its raw declarations in step with the C header: every function include/ce_metrics.h declares has a `pub fn` with the
same number of parameters in src/sys.rs, the status / metric constants agree, and the two plain structs have the
header's field order."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "ce_metrics.h")).read() + open(os.path.join(ROOT, "include", "ce_metrics_debug.h")).read()
SYS = open(os.path.join(ROOT, "bindings", "rust", "codec-eval-hip", "src", "sys.rs")).read()


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_functions():
    out = {}
    for m in re.finditer(r"^[A-Za-z_][A-Za-z0-9_ \*]*?\b(ce_[a-z0-9_]+)\(([^;{]*?)\);", _strip_comments(HEADER), flags=re.M | re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def rust_functions():
    out = {}
    block = SYS[SYS.index('extern "C" {'):]
    for m in re.finditer(r"pub fn (ce_[a-z0-9_]+)\((.*?)\)\s*(?:->[^;]*)?;", block, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_every_exported_function_is_declared_with_the_same_arity():
    c, r = c_functions(), rust_functions()
    assert len(c) >= 40
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    assert {k: v for k, v in c.items() if r[k] != v} == {}


def test_constants_and_struct_layouts():
    for name in ("CE_OK", "CE_ERR_DIM_MISMATCH", "CE_ERR_BAD_LENGTH", "CE_ERR_TOO_SMALL", "CE_ERR_BACKEND", "CE_ERR_INVALID_ARG"):
        cv = int(re.search(name + r"\s*=\s*(\d+)", HEADER).group(1))
        rv = int(re.search(r"pub const " + name + r": c_int = (\d+);", SYS).group(1))
        assert cv == rv, name
    for name in ("CE_METRIC_DSSIM", "CE_METRIC_SSIMULACRA2", "CE_METRIC_BUTTERAUGLI", "CE_METRIC_PSNR", "CE_FLAG_XYB_ROUNDTRIP"):
        cs = int(re.search(name + r"\s*=\s*1u\s*<<\s*(\d+)", HEADER).group(1))
        rs = int(re.search(r"pub const " + name + r": u32 = 1 << (\d+);", SYS).group(1))
        assert cs == rs, name
    for name in ("CE_PIXEL_RGB8", "CE_PIXEL_RGBA8", "CE_PIXEL_RGB16_10BIT", "CE_PIXEL_RGBA16_10BIT"):
        cv = int(re.search(name + r"\s*=\s*(\d+)", HEADER).group(1))
        rv = int(re.search(r"pub const " + name + r": c_int = (\d+);", SYS).group(1))
        assert cv == rv, name

    def c_fields(struct):
        body = re.search(r"typedef struct " + struct + r" \{(.*?)\} " + struct + ";", HEADER, flags=re.S).group(1)
        return [re.split(r"[ \*]", f.strip())[-1] for f in _strip_comments(body).split(";") if f.strip()]

    def rust_fields(struct):
        body = re.search(r"pub struct " + struct + r" \{(.*?)\}", SYS, flags=re.S).group(1)
        return re.findall(r"pub ([a-z0-9_]+):", body)

    assert c_fields("ce_scores") == rust_fields("ce_scores") == ["dssim", "ssimulacra2", "butteraugli", "psnr", "valid", "status"]
    assert c_fields("ce_pair_desc") == rust_fields("ce_pair_desc") == ["reference", "reference_len", "test", "test_len", "width", "height"]
