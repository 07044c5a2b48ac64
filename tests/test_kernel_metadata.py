"""The conv kernels of libtsod.so must not touch scratch memory: conv_dma_kernel counts its vmcnt by hand (a scratch access is
a VMEM operation the count does not know about - results stay right, the waits turn conservative, and the time is gone), and a
spill in any K loop is a performance bug nobody sees in a parity test.  Read from the code objects inside the shipped library
(the .hip_fatbin section), so this runs without a GPU."""
import os
import re
import shutil
import subprocess

import pytest

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "two_stage_object_detection_amd", "libtsod.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernel_notes(tmp_path):
    bundler, readelf = os.path.join(LLVM, "clang-offload-bundler"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(LIB) and os.path.exists(bundler) and os.path.exists(readelf) and shutil.which("objcopy")):
        pytest.skip("libtsod.so or the LLVM / binutils tools are not here")
    fat = tmp_path / "fat.bin"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, str(fat)], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in the library"
    kernels = {}
    for i, a in enumerate(starts):
        part = tmp_path / f"bundle{i}.bin"
        part.write_bytes(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / f"bundle{i}.co"
        r = subprocess.run([bundler, "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0 or not co.exists() or co.stat().st_size == 0:
            continue
        notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True, check=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
                kernels[name] = {}
                continue
            m = re.match(r"\s+\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s+(\d+)", line)
            if m and name is not None:
                kernels[name][m.group(1)] = int(m.group(2))
    return kernels


def test_conv_kernels_use_no_scratch_memory(tmp_path):
    kernels = _kernel_notes(tmp_path)
    conv = {k: v for k, v in kernels.items() if "conv_dma_kernel" in k or "conv_igemm_kernel" in k or "bottleneck_kernel" in k or "stem_kernel" in k}
    assert len(conv) >= 32, sorted(kernels)[:5]                    # 12 LDS-DMA + 23 register-staged instantiations + the fused bottleneck + stem
    assert any("bottleneck_kernel" in k and v.get("vgpr_count", 999) <= 256 for k, v in conv.items())   # two workgroups per CU
    assert any("stem_kernel" in k and v.get("vgpr_count", 999) <= 256 for k, v in conv.items())         # (its LDS: a static_assert)
    bad = {k: v for k, v in conv.items() if v.get("private_segment_fixed_size", 0) != 0 or v.get("vgpr_spill_count", 0) != 0}
    assert not bad, bad
    # two waves per SIMD for the LDS-DMA tiles (512-thread workgroups at one per CU, 256-thread ones at two): 256 registers each
    assert all(v.get("vgpr_count", 0) <= 256 for k, v in conv.items() if "conv_dma_kernel" in k)
