"""The C-ABI library loads and exports every symbol include/tsod.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from two_stage_object_detection_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tsod.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsod_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_ffi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    raw = ctypes.CDLL(_ffi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(raw, name), name
    lib = _ffi.lib()
    assert lib.tsod_version() == 242
    assert lib.tsod_status_str(0) == b"ok"
    assert b"workspace" in lib.tsod_status_str(-4)


def test_conv_desc_layout_matches_header(tmp_path):
    """The ctypes mirror of tsod_conv2d_desc against the C compiler's own layout of include/tsod.h (size and the offsets
    of the first, a middle and the last field)."""
    import subprocess
    # 5 + 16 + 16 + 3 + 5 + 2 + 1 + 1(float) + 5 (.. precision) + 6 (second source) + 2 (fp16x2 scale exponents) int32-sized fields
    # + four pointers (range_flag, amax_in, amax_in2, amax_out)
    assert ctypes.sizeof(_ffi.ConvDesc) == 4 * (5 + 16 + 16 + 3 + 5 + 2 + 1 + 1 + 5 + 6 + 2) + 4 * 8
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tsod.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %d %d\\n", '
                   'sizeof(tsod_conv2d_desc), offsetof(tsod_conv2d_desc, Cout), offsetof(tsod_conv2d_desc, slope), '
                   'offsetof(tsod_conv2d_desc, split_k), offsetof(tsod_conv2d_desc, range_flag), offsetof(tsod_conv2d_desc, amax_in2), '
                   'offsetof(tsod_conv2d_desc, amax_out), TSOD_AMAX_WORDS, TSOD_AMAX_STRIDE); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, o_cout, o_slope, o_split, o_prec, o_in2, o_out, words, stride = (int(v) for v in subprocess.check_output([str(exe)]).split())
    D = _ffi.ConvDesc
    assert (size, o_cout, o_slope, o_split, o_prec, o_in2, o_out) == (ctypes.sizeof(D), D.Cout.offset, D.slope.offset, D.split_k.offset,
                                                                      D.range_flag.offset, D.amax_in2.offset, D.amax_out.offset)
    assert (words, stride) == (_ffi.AMAX_WORDS, _ffi.AMAX_STRIDE)


def test_fused_launch_desc_layouts_match_header(tmp_path):
    """The ctypes mirrors of tsod_bottleneck_desc and tsod_stem_desc against the C compiler's layout of include/tsod.h."""
    import subprocess
    src = tmp_path / "layout2.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tsod.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(tsod_bottleneck_desc), offsetof(tsod_bottleneck_desc, w_exp), offsetof(tsod_bottleneck_desc, projection) * 1000 + offsetof(tsod_bottleneck_desc, range_flag), '
                   'offsetof(tsod_bottleneck_desc, amax_out), sizeof(tsod_stem_desc), offsetof(tsod_stem_desc, slope), '
                   'offsetof(tsod_stem_desc, range_flag), offsetof(tsod_stem_desc, amax_out)); return 0; }\n')
    exe = tmp_path / "layout2"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = tuple(int(v) for v in subprocess.check_output([str(exe)]).split())
    B, S = _ffi.BottleneckDesc, _ffi.StemDesc
    assert got == (ctypes.sizeof(B), B.w_exp.offset, B.projection.offset * 1000 + B.range_flag.offset, B.amax_out.offset,
                   ctypes.sizeof(S), S.slope.offset, S.range_flag.offset, S.amax_out.offset)
    assert (_ffi.STEM_NCHW, _ffi.STEM_NHWC4) == (0, 1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_ffi.TsodError, match="no CPU fallback"):
        _ffi.lib()


def test_cpu_tensors_are_refused():
    import torch
    from two_stage_object_detection_amd import hip_ops
    with pytest.raises(_ffi.TsodError, match="HIP-only"):
        hip_ops.bbox_iou(torch.zeros(2, 4), torch.zeros(2, 4))
