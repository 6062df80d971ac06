"""Instruction census of the MFMA-carrying basic blocks of a kernel file (no GPU needed): compiles ONE .hip source of
gpt_image_edit_amd/csrc to gfx950 assembly with the Makefile's flags and prints, per kernel and per basic block with at
least MIN_MFMA matrix instructions, how many MFMAs, VALU address additions, exponentials, LDS reads, LDS-DMA requests,
waits and scratch accesses the block holds, plus the kernel's register / spill figures.

    python tools/isa_census.py attention_fwd.hip [MIN_MFMA=16] [extra hipcc flags ...]

This is how DESIGN.md section 7's issue-slot budget of the attention forward was counted (32 MFMA, 32 v_fma, 32 v_exp,
33 v_add_f32, 16 cvt_pk, 48 DS reads, 14 v_add_u32 per tile and wave) and how the stage-constant variant of round 4 was
checked before it went to the GPU (0 / 4 / 4 address additions, no scratch in the loop)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpt_image_edit_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value", "-Wno-unused-result"]


def compile_to_asm(src, extra):
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "k.s")
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def census(lines, min_mfma):
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+: ", l) or re.match(r"^_Z\w+:\s*;", l)]
    starts.append(len(lines))
    meta = {}
    for l in lines:
        m = re.match(r"\s*\.set (_Z\w+)\.(num_vgpr|num_agpr|private_seg_size), (\d+)", l)
        if m:
            meta.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    for s, e in zip(starts[:-1], starts[1:]):
        name = lines[s].split(":")[0]
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        blocks, cur, lab = [], [], "entry"
        for l in lines[s:e]:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                blocks.append((lab, cur))
                cur, lab = [], m.group(1)
            else:
                cur.append(l)
        blocks.append((lab, cur))
        hot = [(lab, b) for lab, b in blocks if sum("v_mfma" in x for x in b) >= min_mfma]
        if not hot:
            continue
        print(f"{demangled[:150]}\n  registers: {meta.get(name, {})}")
        for lab, b in hot:
            ops = collections.Counter()
            for x in b:
                m = re.match(r"^\s+([a-z_0-9]+)", x)
                if m:
                    ops[m.group(1)] += 1
            pick = lambda pre: sum(v for k, v in ops.items() if k.startswith(pre))
            print(f"  {lab:<10} instrs {sum(ops.values()):4d} | mfma {pick('v_mfma'):3d} v_fma_f32 {ops['v_fma_f32']:3d} v_exp {pick('v_exp'):3d} "
                  f"v_add_f32 {pick('v_add_f32'):3d} cvt_pk {pick('v_cvt_pk'):3d} v_add_u32 {pick('v_add_u32'):3d} v_pk_* {pick('v_pk_'):3d} | "
                  f"ds_read {pick('ds_read'):3d} ds_write {pick('ds_write'):3d} lds-dma/buffer_load {pick('buffer_load'):3d} global_load {pick('global_load'):3d} | "
                  f"s_waitcnt {ops['s_waitcnt']:3d} s_barrier {ops['s_barrier']:2d} s_nop {ops['s_nop']:2d} scratch {pick('scratch_'):2d}")


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    src = sys.argv[1] if os.path.isabs(sys.argv[1]) else os.path.join(CSRC, sys.argv[1])
    rest = sys.argv[2:]
    min_mfma = int(rest.pop(0)) if rest and rest[0].isdigit() else 16
    census(compile_to_asm(src, rest), min_mfma)
