"""Register / scratch budget of the hot kernels, read from the code hipcc generates for gfx950 (no GPU needed).

A hot loop whose accumulator arrays the compiler could not keep in registers still computes the right numbers -- 25x
slower (private-segment arrays; it happened once when the attention tile body was wrapped in a second lambda).  The
parity tests cannot see that, this one can: every MFMA kernel must stay within the 256-VGPR budget of two waves per
SIMD with at most a few spilled dwords, none of them arrays.
"""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpt_image_edit_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# file -> (kernel-name substring, max private-segment bytes per lane)
BUDGET = {
    "attention_fwd.hip": [("attention_fwd_kernel", 64)],
    "attention_bwd.hip": [("attention_bwd_kernel", 64)],
    # the split-K instantiation (..., true) moves one accumulator tile through scratch around its rendezvous, outside the
    # K loop (at most 8 stores + 8 loads per workgroup); every other instantiation keeps everything in registers
    "gemm_pingpong_bf16.hip": [("gemm8_kernelILi", 0), ("gemm9_kernel", 0), ("gemm_mix_kernel", 0), ("gemm8_streamk_kernel", 0)],
}
# key -> [(name fragment, max scratch bytes, max spilled VGPRs)]; the fp32-output parity build of the split-K form (epilogue
# 64 = FK_EPI_F32DBG, test-only) keeps its bias quads live across the rendezvous as well: more of the same, still outside the K loop
# (template arguments <EPI, BN, SPLITK, LAY, M16>: the fragments below match both MFMA shapes)
EXCEPTIONS = {"gemm8_kernelILi": [("ILi64ELi256ELb1ELi0ELb", 320, 72), ("Lb1ELi0ELb", 136, 32)],
              # the stream-K form loops over passes (tile part, rendezvous, epilogue): what is live across a pass sits in
              # scratch around it -- ~90 scratch instructions per pass, NONE inside the K loop (checked below)
              "gemm8_streamk_kernel": [("ILi64E", 200, 300), ("", 260, 120)]}


@pytest.mark.parametrize("src", sorted(BUDGET))
def test_hot_kernels_keep_their_accumulators_in_registers(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-Wno-unused-result", "-S",
           "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = out.read_text()
    meta = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n"
                      r"\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert meta, "no kernel metadata found in the generated assembly"
    seen = 0
    for name, scratch, vgprs, spills in meta:
        for key, max_scratch in BUDGET[src]:
            if key in name:
                seen += 1
                assert int(vgprs) <= 256, f"{name}: {vgprs} VGPRs"      # two waves per SIMD -> 256 registers each
                max_spills = 16
                for frag, relaxed, relaxed_spills in EXCEPTIONS.get(key, []):
                    if frag in name:
                        max_scratch, max_spills = relaxed, relaxed_spills
                        break
                assert int(scratch) <= max_scratch, f"{name}: {scratch} B of scratch per lane (arrays in private memory?)"
                assert int(spills) <= max_spills, f"{name}: {spills} spilled VGPRs"
    assert seen >= len(BUDGET[src]), f"expected kernels {BUDGET[src]} in {src}"
    if src == "gemm_pingpong_bf16.hip":     # the stream-K kernels' spills must stay outside the K loop (the 64-MFMA loop body)
        lines = text.split("\n")
        starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN12_GLOBAL__N_120gemm8_streamk_kernel\w+:", l)]
        assert starts
        for a in starts:
            if "ILi64E" in lines[a]:       # the fp32-output parity build (test-only) may spill where it likes
                continue
            e = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
            body = lines[a:e]
            labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
            loops = []
            for i, l in enumerate(body):
                m = re.search(r"(?:s_cbranch_\w+|s_branch) (\.LBB\d+_\d+)", l)
                if m and m.group(1) in labels and labels[m.group(1)] < i and any("v_mfma" in x for x in body[labels[m.group(1)]:i]):
                    loops.append((labels[m.group(1)], i))
            assert loops, "no loop with MFMAs found"
            head = min(loops, key=lambda ab: ab[1] - ab[0])[0]      # header of the innermost loop that multiplies ...
            k_loop = body[head:max(b for a_, b in loops if a_ == head)]   # ... up to its last back edge: two K-tiles = 64 MFMAs
            n_mfma = sum("v_mfma" in x for x in k_loop)
            assert n_mfma == (128 if "16x16x32" in "".join(k_loop) else 64)      # 32 x 32 x 16: 64 per two K-tiles; 16 x 16 x 32: 128
            assert not any("scratch_" in x for x in k_loop), f"{lines[a][:70]}: scratch traffic inside the K loop"
        # gemm10_kernel: 512 registers per lane by design (one wave per SIMD, all 256 accumulators in the AGPR half, operands of
        # the asm statement pinned); its K loop is the statement gemm10_loop.inc, which must arrive in the code object untouched:
        # 3 tile bodies x 128 MFMAs, no compiler instruction inside, the hazard padding in front of the epilogue's first
        # v_accvgpr_read, and at most a few dwords of scratch outside it (the opaque thread id of the per-CU tile loop)
        g10 = [(name, int(scratch), int(vg)) for name, scratch, vg, _ in meta if "gemm10_kernel" in name]
        assert len(g10) >= 16, f"expected plain + one-workgroup-per-CU instantiations of every epilogue, got {len(g10)}"
        for name, scratch, vg in g10:
            assert vg == 512 and scratch <= 32, f"{name}: {vg} registers, {scratch} B scratch"
        for a in [i for i, l in enumerate(lines) if re.match(r"^_ZN12_GLOBAL__N_113gemm10_kernel\w+:", l)]:
            e = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
            body = lines[a:e]
            spans = [(i, next(j for j in range(i, len(body)) if "#ASMEND" in body[j])) for i, l in enumerate(body) if "#ASMSTART" in l]
            lo, hi = max(spans, key=lambda ab: ab[1] - ab[0])
            stmt = [l.strip() for l in body[lo + 1:hi] if l.strip()]
            assert sum(l.startswith("v_mfma_f32_16x16x32_bf16") for l in stmt) == 384
            assert sum(l.startswith("s_barrier") for l in stmt) == 4 and not any("scratch_" in l or "v_accvgpr" in l for l in stmt)
            assert stmt[-2:] == ["s_nop 15", "s_nop 15"], "hazard padding between the last MFMA and the epilogue's accumulator reads"
            assert not any("v_mfma" in l for l in body[:lo] + body[hi:]), "an MFMA outside the statement"
        # No instantiation's K loop may wait for ALL vector-memory requests: the loop's LDS-DMA prefetch is in flight there and
        # the kernels order their ring with counted waits.  hipcc inserts exactly that wait in front of an LDS read it cannot
        # prove disjoint from a builtin LDS-DMA request (the K-major forms' transpose reads, until round 5: 12-40 % of the kernel).
        starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN12_GLOBAL__N_1\d+gemm(8|9|_mix|8_streamk)_kernel\w+:", l)]
        assert len(starts) >= 10
        for a in starts:
            e = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
            body = lines[a:e]
            labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
            loops = []
            for i, l in enumerate(body):
                m = re.search(r"(?:s_cbranch_\w+|s_branch) (\.LBB\d+_\d+)", l)
                if m and m.group(1) in labels and labels[m.group(1)] < i and sum("v_mfma" in x for x in body[labels[m.group(1)]:i]) >= 32:
                    loops.append((labels[m.group(1)], i))
            if not loops:
                continue
            lo, hi = min(loops, key=lambda ab: ab[1] - ab[0])
            assert not any("vmcnt(0)" in x for x in body[lo:hi]), f"{lines[a][:80]}: s_waitcnt vmcnt(0) inside the K loop"


def test_attention_fwd4_hand_placed_hazards_and_steady_loop(tmp_path):
    """attention_fwd4_kernel issues its S^T chains as inline-asm MFMAs, whose hazards hipcc cannot see: an XDL result needs 12
    wait states before a vector instruction reads it.  Read the generated code: every chain's last MFMA is >= 12 issue states
    (instructions + s_nop states, the hazard recognizer's own count) away from the first VALU read of its accumulator; and the
    steady-state tile loop (64 MFMAs) carries no AGPR copies, no scratch traffic and no wait for ALL vector-memory requests
    (the LDS-DMA prefetch in flight)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    import importlib.util
    spec = importlib.util.spec_from_file_location("a4_census", os.path.join(os.path.dirname(CSRC), "..", "tools", "a4_census.py"))
    census = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(census)
    out = tmp_path / "attention_fwd4.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-Wno-unused-value", "-Wno-unused-result",
           "-S", "--cuda-device-only", os.path.join(CSRC, "attention_fwd4.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = out.read_text()
    steady = 0
    for tag in ("ILb0E", "ILb1E"):
        name = "_ZN12_GLOBAL__N_121attention_fwd4_kernel%sEEv10AttnParams" % tag
        i = text.index(name + ":")
        body = text[i:text.index(".Lfunc_end", i)]
        chains = 0
        for lab, ins in census.blocks_of(body):
            if sum(x.startswith("v_mfma") for x in ins) < 32:
                continue
            for at, states, reader in census.hazard_distances(ins):
                chains += 1
                assert states >= 12, f"{tag} {lab}: {states} states between the chain's last MFMA (#{at}) and `{reader}`"
            n_mfma = sum(x.startswith("v_mfma") for x in ins)
            if (n_mfma == 64 and not any("accvgpr" in x for x in ins) and not any("scratch_" in x for x in ins)
                    and not any("vmcnt(0)" in x for x in ins)):
                steady += 1
        assert chains >= 8, f"{tag}: only {chains} asm chains found"
    assert steady >= 2, "no clean 64-MFMA tile body found (AGPR copies, scratch or vmcnt(0) in every one)"


def test_gemm10_loop_file_is_what_its_generator_writes():
    """csrc/gemm10_loop.inc (the K loop of gemm10_kernel, one asm statement) is generated: the committed file must be the
    generator's output, so that the schedule documented in gemm10_gen.py is the schedule that ships."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(CSRC, "gemm10_gen.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "gemm10_loop.inc is stale: run `python gpt_image_edit_amd/csrc/gemm10_gen.py`"
    text = open(os.path.join(CSRC, "gemm10_loop.inc")).read()
    assert text.count("v_mfma_f32_16x16x32_bf16") == 384 and text.count("s_memtime") == 0
