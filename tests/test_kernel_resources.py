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
        assert 160 * 1024 // lds >= 8 // waves, (h, lds)       # LDS admits the instances the 8 wave slots of a CU can hold


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which(HIPCC)), reason="hipcc not available")
def test_no_dpp_hazard_behind_inline_asm(tmp_path):
    """Guard for inline-asm DPP instructions (the DPP-broadcast sweep variant of DESIGN.md section 9 used
    `v_fmac_f32_dpp`; the shipped kernel has none, the compiler's own DPP moves are hazard-checked by the
    compiler).  The compiler's hazard recogniser does not look inside
    inline asm, so nothing inserts the two wait states a DPP read needs after a VALU write of the same register.
    The kernels feed them from LDS loads only; this test scans the generated ISA to make sure no vector
    instruction writes a DPP source register within the two instructions before its DPP read."""
    import __graft_entry__ as ge
    out = str(tmp_path / "bmpc.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "--cuda-device-only", "-S", os.path.join(ge.CSRC, "bmpc_capi.hip"), "-o", out] + ge.KERNEL_FLAGS,
                          cwd=ge.CSRC, stderr=subprocess.DEVNULL)
    lines = [ln.split(";")[0].strip() for ln in open(out).read().splitlines()]
    lines = [ln for ln in lines if ln and not ln.startswith(".") and not ln.endswith(":")]

    def written(ln):
        parts = ln.split(None, 1)
        if len(parts) < 2 or not parts[0].startswith("v_"):
            return set()
        dst = parts[1].split(",")[0].strip()
        m = re.match(r"v\[(\d+):(\d+)\]", dst)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", dst)
        return {int(m.group(1))} if m else set()

    n = 0
    for i, ln in enumerate(lines):
        if ln.startswith("v_fmac_f32_dpp"):
            n += 1
            src = int(re.findall(r"v(\d+)", ln)[1])
            for k in (1, 2):
                assert src not in written(lines[i - k]), (lines[i - k], ln)
    print("inline-asm DPP instructions checked:", n)      # none in the shipped variant (the broadcast sweep is not kept)
