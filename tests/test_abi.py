"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/dealyolo_hip.h declares (no compute
calls -- there is no GPU here)."""
import os
import re
import subprocess

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dealyolo_hip.h")
LIBDIR = os.path.join(ROOT, "experiment-yolo_amd", "csrc")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dy_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    subprocess.run(["make", "-C", LIBDIR, "-j8"], check=True, capture_output=True)
    from ultralytics.hip import SIGNATURES, lib
    L = lib()
    decl = _declared()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(L, name), f"{name} declared in dealyolo_hip.h but not exported"
        assert name in SIGNATURES, f"{name} has no ctypes signature in ultralytics/hip/__init__.py"
    assert sorted(SIGNATURES) == decl, set(SIGNATURES) ^ set(decl)
    assert L.dy_abi_version() >= 1


def test_geometry_helpers_run_on_host():
    """Pure host-side planning helpers may be called without a GPU."""
    import ctypes as C
    from ultralytics.hip import lib
    L = lib()
    g = [C.c_int() for _ in range(8)]
    assert L.dy_conv_geometry(64, 64, 3, 1, *[C.byref(x) for x in g]) == 0
    cin_p, cout_p, cc, nch, mt, ng, kst, pe = [x.value for x in g]
    # 64-channel 3x3: two 32-channel Cin chunks (so the halo tiles share LDS with the 72 KiB of weights)
    assert (cin_p, cout_p, cc, nch, mt, ng, kst) == (64, 64, 32, 2, 4, 1, 9) and pe == 18 * 64 * 32
    assert L.dy_conv_geometry(3, 16, 3, 2, *[C.byref(x) for x in g]) == 0 and g[0].value == 8
    assert L.dy_conv_geometry(64, 64, 5, 1, *[C.byref(x) for x in g]) == -1  # unsupported kernel size -> DY_ERR_ARG
    assert L.dy_loss_workspace_bytes(64, 33600, 8) > 64 * 33600 * 4 * 10


def test_no_fallback_on_cpu():
    """The product path fails loudly without a GPU instead of computing on the CPU."""
    import pytest
    import torch
    from ultralytics.nn.modules import Conv
    m = Conv(16, 16, 3)
    with pytest.raises(RuntimeError, match="GPU only|no CPU fallback"):
        m(torch.zeros(1, 16, 8, 8))


def test_conv_kernel_name_helper():
    """bench.py groups its live timings by the kernel instantiation dy_conv_forward launches (rocprofv3 spelling)."""
    import ctypes as C
    from ultralytics.hip import lib
    L = lib()
    buf = C.create_string_buffer(128)
    assert L.dy_conv_kernel_name(64, 64, 3, 1, buf, 128) == 0 and buf.value == b"conv_mfma_pp_kernel<32, 4, 3, 1, 2, false, 0>"
    assert L.dy_conv_kernel_name_at(64, 64, 3, 1, 40, 1, 0, buf, 128) == 0 and buf.value == b"conv_mfma_pp_kernel<32, 4, 3, 1, 2, false, 40>"
    assert L.dy_conv_kernel_name_at(64, 64, 3, 1, 48, 1, 0, buf, 128) == 0 and buf.value == b"conv_mfma_pp_kernel<32, 4, 3, 1, 2, false, 0>"
    assert L.dy_conv_kernel_name(64, 64, 1, 1, buf, 128) == 0 and buf.value.startswith(b"conv_mfma_pp_kernel<64, 4, 1, 1,")
    assert L.dy_conv_kernel_name(64, 64, 5, 1, buf, 128) != 0
    # a 1x1 conv over a never-materialised concatenation stages the chunk no segment boundary cuts: four 32-channel members -> 32
    # (the dense 128-channel input takes 64), three 16-channel members -> 16
    from ultralytics.hip import DySegs
    t = DySegs()
    t.nseg = 4
    for i in range(4):
        t.c_end[i], t.ld[i], t.ptr[i] = 32 * (i + 1), 32, 4096
    assert L.dy_conv1x1_segs_kernel_name(128, 64, C.byref(t), buf, 128) == 0 and buf.value == b"conv_mfma_pp_kernel<32, 4, 1, 1, 2, false, 0>"
    assert L.dy_conv_kernel_name(128, 64, 1, 1, buf, 128) == 0 and buf.value == b"conv_mfma_pp_kernel<64, 4, 1, 1, 2, false, 0>"
    t.nseg = 3
    for i in range(3):
        t.c_end[i], t.ld[i] = 16 * (i + 1), 16
    assert L.dy_conv1x1_segs_kernel_name(48, 32, C.byref(t), buf, 128) == 0 and buf.value == b"conv_mfma_pp_kernel<16, 2, 1, 1, 2, false, 0>"
    assert L.dy_conv1x1_segs_kernel_name(64, 32, C.byref(t), buf, 128) != 0  # the table does not cover the channels
    # the weight gradient of the same layer: one (64, 64) channel block per workgroup on a large map, (32, 32) blocks on a 40x40 one,
    # where the fp32 weight slabs of 256 workgroup columns would outweigh the activations
    assert L.dy_wgrad_kernel_name(64, 64, 3, 1, buf, 128) == 0 and buf.value == b"conv_wgrad_kernel<3, 1, 4, 4, 0>"
    assert L.dy_wgrad_kernel_name_at(64, 160, 160, 64, 64, 3, 1, buf, 128) == 0 and buf.value == b"conv_wgrad_kernel<3, 1, 4, 4, 0>"
    assert L.dy_wgrad_kernel_name_at(64, 40, 40, 64, 64, 3, 1, buf, 128) == 0 and buf.value == b"conv_wgrad_kernel<3, 1, 2, 2, 0>", buf.value
    assert L.dy_wgrad_reduce_desc_bytes() >= 64
    # the argument block of dy_detection_loss has the same size on both sides of the boundary (lib() refuses to load otherwise)
    from ultralytics.hip import DyLossArgs
    assert L.dy_loss_args_bytes() == C.sizeof(DyLossArgs)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 without HIP (what cgo / bindgen / cffi would feed a C compiler),
    and a C translation unit that takes the address of every declared function must link against the library."""
    import shutil
    import pytest
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "dealyolo_hip.h")
    names = sorted(set(re.findall(r"^\s*(?:int|long|void|const char\s*\*)\s+(dy_\w+)\s*\(", open(hdr).read(), flags=re.M)))
    assert len(names) > 40
    src = tmp_path / "abi.c"
    src.write_text('#include "dealyolo_hip.h"\n#include <stdio.h>\nint main(void) {\n  void* f[] = {' + ", ".join(f"(void*){n}" for n in names) +
                   '};\n  printf("%d %d\\n", (int)(sizeof f / sizeof f[0]), dy_abi_version());\n  return 0;\n}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-Wno-pedantic", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)],
                   check=True, capture_output=True)
    lib = os.path.join(root, "experiment-yolo_amd", "csrc", "libdealyolo_hip.so")
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), lib, "-Wl,-rpath," + os.path.dirname(lib),
                        "-Wl,--unresolved-symbols=ignore-in-shared-libs", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
