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


@pytest.fixture(scope="module")
def isa_text(tmp_path_factory):
    """The gfx950 ISA + code object metadata of every kernel, compiled once for the three tests below."""
    import __graft_entry__ as ge
    out = str(tmp_path_factory.mktemp("isa") / "bmpc.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "--cuda-device-only", "-S", os.path.join(ge.CSRC, "bmpc_capi.hip"), "-o", out] + ge.KERNEL_FLAGS,
                          cwd=ge.CSRC, stderr=subprocess.DEVNULL)
    return open(out).read()


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which(HIPCC)), reason="hipcc not available")
def test_no_spills_and_two_waves_per_simd(isa_text):
    text = isa_text
    seen = {}
    for entry in re.split(r"\n\s+- (?=\.agpr_count:)", text)[1:]:          # one metadata entry per kernel
        m = re.search(r"\.name:\s+\S*solve_kernelILi(\d+)EE", entry)       # (not the diagnostics build solve_kernel_prof)
        if m:
            seen[int(m.group(1))] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\n", entry.split(".wavefront_size")[0])
                                     if k != "offset" and k != "size"}
    assert sorted(seen) == [8, 10, 12, 14, 16, 18, 20], seen
    for h, meta in seen.items():
        # Round 6: NOTHING spills up to h = 18, and h = 20 keeps 4 dwords (a set-up index, an LDS address, one f64 -- reloaded at
        # stopping tests / on the way out; test_no_scratch_access_in_the_hot_loops) where round 5 had 13 / 34 spilled registers
        # at h = 18 / 20 and 132 bytes of scratch.  What moved it was not the allocator's mood but what the loop keeps alive:
        # uniform f32 products of the stopping test and the re-classification formed on the host (gfx950 has no scalar float
        # unit: formed in the kernel they are loop invariants in VECTOR registers), f64 copies of the bounds widened at their
        # use, the lane's indices formed again from the row where a rebuild needs them, the residual norms stored at the test
        # instead of carried to the end of the kernel, Gt[row][row] picked out of the row half by factor().  The bounds are
        # the measured values: they go down with the code, not up.
        assert int(meta["vgpr_spill_count"]) <= {8: 0, 10: 0, 12: 0, 14: 0, 16: 0, 18: 0, 20: 4}[h], (h, meta)
        # (up to h = 18 at most a 36-byte reservation the code never touches: test_no_scratch_access_in_the_hot_loops holds that
        #  those kernels contain no scratch instruction at all)
        assert int(meta["private_segment_fixed_size"]) <= (20 if h == 20 else 36), (h, meta)
        assert int(meta["vgpr_count"]) + meta["agpr_count"] <= 256, (h, meta)      # two waves per SIMD
        lds = int(meta["group_segment_fixed_size"])
        waves = (12 * h + 63) // 64 if h % 5 else 2 * (h // 5) * 64 // 64
        assert 160 * 1024 // lds >= 8 // waves, (h, lds)       # LDS admits the instances the 8 wave slots of a CU can hold
    # the stage-structured family: one wave per instance up to h = 24, two from h = 26; what an instance holds in LDS
    # decides how many share a CU, and no variant spills (two waves halve the steps a lane owns)
    stage = {}
    for entry in re.split(r"\n\s+- (?=\.agpr_count:)", text)[1:]:
        m = re.search(r"\.name:\s+\S*stage_kernelILi(\d+)ELi(\d+)EE", entry)
        if m:
            stage[(int(m.group(1)), int(m.group(2)))] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\n", entry.split(".wavefront_size")[0])
                                                         if k != "offset" and k != "size"}
    assert sorted(stage) == [(2, 1), (3, 1), (3, 2), (4, 1), (4, 2), (5, 1)], stage
    for (n_p, n_w), meta in stage.items():
        lds = int(meta["group_segment_fixed_size"])
        steps = 5 * n_p * n_w
        assert lds <= 2000 * steps + 6000 * n_w, (n_p, n_w, lds)    # ~2 KB per step + the block-algebra scratch of one pass
        assert 160 * 1024 // lds >= 2, (n_p, n_w, lds)              # at least two instances per CU at h = 40
        # (no variant spills a vector register -- round 5's five-steps-per-lane one (h = 21 .. 24, all 512 registers) kept 6 in
        #  scratch: uniform f32 values of the stopping test, formed on the host since round 6)
        assert int(meta["vgpr_spill_count"]) == 0 and int(meta["private_segment_fixed_size"]) == 0, (n_p, n_w, meta)
        # What the family costs in registers, stated as it is (the round-3 test asserted `vgpr_count <= 512`, the
        # hardware maximum): every variant needs MORE than 256 of the unified 512 registers (the count includes the AGPRs it
        # parks values in), i.e. one wave per SIMD, and keeps ~200 uniform values in lanes of VGPRs (SGPR spills: v_writelane /
        # v_readlane, no scratch).  These bounds are regression guards for DESIGN.md section 5b's numbers, not targets.
        assert 256 < int(meta["vgpr_count"]) <= 512, (n_p, n_w, meta)
        assert int(meta["sgpr_spill_count"]) <= 252, (n_p, n_w, meta)      # (193 .. 251 in round 6, 195 .. 249 in round 5; 174 .. 222 in round 4)
        # instances per CU: LDS admits 160 KB / lds, one wave per SIMD admits 4 / n_w -- the smaller one is what DESIGN.md quotes
        per_cu = min(160 * 1024 // lds, 4 // n_w)
        assert per_cu == {(2, 1): 4, (3, 1): 4, (4, 1): 4, (5, 1): 3, (3, 2): 2, (4, 2): 2}[(n_p, n_w)], (n_p, n_w, per_cu)

@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which(HIPCC)), reason="hipcc not available")
def test_hessian_block_gemm_runs_on_the_matrix_cores(isa_text):
    """north_star: "MFMA used only for the dense horizon-block GEMMs inside the condensed Hessian".  The one such GEMM of the
    wrench-space form -- the torque block Gt_tt = M' M of the set-up -- is formed with v_mfma_f64_16x16x4_f64 in every dense
    kernel (the Euler rows four per instruction, unrolled; the angular-velocity rows -- the same three at every state step -- as one
    rank-3 instruction and a count; f64: the row it fills is also the operator of the carried gradient's increments), and nowhere
    else: the stage family never forms Gt."""
    lines = isa_text.splitlines()
    seen = {}
    for i, ln in enumerate(lines):
        m = re.match(r"_ZN4bmpc\d+(solve|stage)_kernel(?:_prof)?ILi(\d+)E(?:Li\d+E)?E\S*:", ln)
        if not m:
            continue
        end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        n = sum(1 for x in lines[i + 1:end] if x.split(";")[0].strip().startswith("v_mfma_f64_16x16x4"))
        any_mfma = sum(1 for x in lines[i + 1:end] if x.split(";")[0].strip().startswith("v_mfma"))
        assert any_mfma == n, ln
        seen.setdefault(m.group(1), []).append(n)
    assert len(seen["solve"]) == 14 and all(n >= 2 for n in seen["solve"]), seen
    assert all(n == 0 for n in seen["stage"]), seen


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which(HIPCC)), reason="hipcc not available")
def test_no_scratch_access_in_the_hot_loops(isa_text):
    """Where a dense kernel spills (round 6: h = 20 only, 4 dwords), the spilled values are stored during the set-up and reloaded in code that
    runs at stopping tests / on the way out only: no scratch instruction sits in the body of the sweep loop or of the ADMM
    iteration.  Both hot bodies are recognisable in the ISA by their packed FMAs: the code between the first and the last
    `v_pk_fma_f32` of a kernel spans the sweep loop and the iteration's phases P0-P5 (mat-vec and gradient increment);
    the stopping test, the rebuild and the outputs follow the last one.  Scratch STORES in between are allowed only before
    the sweep (set-up / factor prologue executed once per factorisation); LOADS are not allowed inside the sweep loop, and at
    most the factor prologue's (one per factorisation) before the iteration phases."""
    lines = isa_text.splitlines()
    seen = 0
    for i, ln in enumerate(lines):
        m = re.match(r"_ZN4bmpc\d+solve_kernelILi(\d+)EE\S*:", ln)
        if not m:
            continue
        seen += 1
        end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        body = [x.split(";")[0].strip() for x in lines[i + 1:end]]
        pk = [k for k, x in enumerate(body) if x.startswith("v_pk_fma_f32")]
        bars = [k for k, x in enumerate(body) if x.startswith("s_barrier")]
        # the sweep loop: the backward branch that encloses the densest run of packed FMAs
        labels = {mm.group(1): k for k, x in enumerate(body) for mm in [re.match(r"(\.LBB\d+_\d+):", x)] if mm}
        loops = []
        for k, x in enumerate(body):
            mm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", x)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                a = labels[mm.group(1)]
                loops.append((k - a, sum(1 for q in pk if a <= q <= k), a, k))
        # (the innermost loop that holds a whole sweep group: three two-pivot steps = 6 packed FMAs per register pair)
        _, npk, a, b = min(x for x in loops if x[1] >= 60)
        inside = [x for x in body[a:b + 1] if x.startswith("scratch_")]
        assert not inside, (m.group(1), "scratch access inside the sweep loop", inside)
        # the iteration phases: from the end of the sweep loop to the last packed FMA (the gradient increment of P5)
        loads = [x for x in body[b:pk[-1] + 1] if x.startswith("scratch_load")]
        assert len(loads) <= 2, (m.group(1), "scratch loads between the sweep and the end of the iteration phases", loads)
        if int(m.group(1)) <= 18:               # nothing spills up to h = 18: no scratch instruction at all
            assert not any(x.startswith("scratch_") for x in body), m.group(1)
    assert seen == 7


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which(HIPCC)), reason="hipcc not available")
def test_every_barrier_waits_for_the_waves_lds_operations(isa_text):
    """An s_barrier that a wave can reach with an LDS store still in flight lets the other waves read the old value.
    ROCm 7.2's hipcc leaves the `s_waitcnt lgkmcnt(0)` of __syncthreads()' release fence out at the top of the sweep
    loop (the loop's back edge carries a pending ds_write): on MI355X that showed as results changing from run to
    run as soon as two waves shared a SIMD.  The kernels therefore write the wait out (bmpc::sync_workgroup); this
    test reads the ISA of every solve kernel and requires that, walking back from each s_barrier, an
    `s_waitcnt ... lgkmcnt(0)` comes before any LDS instruction, branch or block label."""
    lines = isa_text.splitlines()
    checked = 0
    for i, ln in enumerate(lines):
        m = re.match(r"(_ZN4bmpc\d+(?:solve|stage)_kernel\w*ILi\d+E(?:Li\d+E)?E\S*):", ln)
        if not m:
            continue
        end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
        body = [x.split(";")[0].strip() for x in lines[i + 1:end]]
        body = [x for x in body if x and (not x.startswith(".") or re.match(r"\.LBB\d+_\d+:", x))]
        for k, x in enumerate(body):
            if not x.startswith("s_barrier"):
                continue
            checked += 1
            j = k - 1
            while True:
                assert j >= 0, (m.group(1), "barrier at the top of the kernel")
                y = body[j]
                if y.startswith("s_waitcnt") and "lgkmcnt(0)" in y:
                    break
                assert not (y.startswith("ds_") or y.startswith(".LBB") or y.startswith("s_cbranch") or y.startswith("s_branch")), \
                    (m.group(1), "s_barrier reachable without lgkmcnt(0)", body[max(0, j - 3):k + 1])
                j -= 1
    assert checked >= 14 * 12, checked          # 7 horizons x {solve_kernel, solve_kernel_prof}
    # the one-wave stage-structured kernels must not contain a single s_barrier (the two-wave ones are covered by the walk
    # above: the pattern matches their names too)
    n_stage = 0
    for i, ln in enumerate(lines):
        m = re.match(r"(_ZN4bmpc\d+stage_kernel\w*ILi\d+ELi1EE\S*):", ln)
        if m:
            n_stage += 1
            end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
            assert not any(x.split(";")[0].strip().startswith("s_barrier") for x in lines[i + 1:end]), m.group(1)
    assert n_stage == 8


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which(HIPCC)), reason="hipcc not available")
def test_no_dpp_hazard_behind_inline_asm(isa_text):
    """Guard for inline-asm DPP instructions (the DPP-broadcast sweep variant of docs/history_r02_r03.md used
    `v_fmac_f32_dpp`; the shipped kernel has none, the compiler's own DPP moves are hazard-checked by the
    compiler).  The compiler's hazard recogniser does not look inside
    inline asm, so nothing inserts the two wait states a DPP read needs after a VALU write of the same register.
    The kernels feed them from LDS loads only; this test scans the generated ISA to make sure no vector
    instruction writes a DPP source register within the two instructions before its DPP read."""
    lines = [ln.split(";")[0].strip() for ln in isa_text.splitlines()]
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
            # two wait states between a VALU write of the source and its DPP read: an `s_nop N` counts N + 1, any other
            # instruction 1
            states, k = 0, i - 1
            while states < 2:
                prev = lines[k]
                assert src not in written(prev), (prev, ln)
                m = re.match(r"s_nop\s+(\d+)", prev)
                states += int(m.group(1)) + 1 if m else 1
                k -= 1
    print("inline-asm DPP instructions checked:", n)      # the row-broadcast mat-vecs of the stage-structured kernels
    assert n >= 6 * 2 * 24
