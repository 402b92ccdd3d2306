"""Compile-time resources of the solve kernel, read from the code object metadata hipcc emits for gfx950 (no GPU
needed): no register spills and no scratch for ANY horizon, and a register budget that lets two waves share a
SIMD (VGPR + AGPR <= 256).  DESIGN.md section 5 quotes these numbers; VERDICT r1 found the h = 20 instantiation
spilling while the document said otherwise -- this test is what keeps the two in step."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which(HIPCC)), reason="hipcc not available")
def test_no_spills_and_two_waves_per_simd(tmp_path):
    import __graft_entry__ as ge
    out = str(tmp_path / "bmpc.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "--cuda-device-only", "-S", os.path.join(ge.CSRC, "bmpc_capi.hip"), "-o", out] + ge.KERNEL_FLAGS,
                          cwd=ge.CSRC, stderr=subprocess.DEVNULL)
    text = open(out).read()
    seen = {}
    for entry in re.split(r"\n\s+- (?=\.agpr_count:)", text)[1:]:          # one metadata entry per kernel
        m = re.search(r"\.name:\s+\S*solve_kernelILi(\d+)EE", entry)
        if m:
            seen[int(m.group(1))] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\n", entry.split(".wavefront_size")[0])
                                     if k != "offset" and k != "size"}
    assert sorted(seen) == [10, 16, 20], seen
    for h, meta in seen.items():
        assert int(meta["vgpr_spill_count"]) == 0, (h, meta)
        assert int(meta["private_segment_fixed_size"]) == 0, (h, meta)       # no scratch
        assert int(meta["vgpr_count"]) + meta["agpr_count"] <= 256, (h, meta)      # two waves per SIMD
        lds = int(meta["group_segment_fixed_size"])
        waves = {10: 2, 16: 3, 20: 4}[h]
        assert (160 * 1024 // lds) * waves >= 8, (h, lds)                     # LDS lets 8 waves live on a CU
