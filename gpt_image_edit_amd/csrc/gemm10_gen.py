#!/usr/bin/env python3
"""Generator of the hand-placed main loop of gemm10_kernel (gemm10.hip): writes gemm10_loop.inc, the body of ONE asm statement.

gemm10 = 256 x 256 x 64 tile, FOUR waves (one per SIMD), 128 x 128 per wave, v_mfma_f32_16x16x32_bf16, all 256 accumulator
registers pinned in the AGPR half, every instruction of the K loop placed by hand (VERDICT r5 "next" #2).  hipcc cannot build this
loop: it rotates 16 x 16 accumulators through v_accvgpr copies once the AGPR half is full.  Here the
registers are fixed numbers, the asm statement's operands are pinned to them ({v[..]}, {a[..]}, {s[..]} constraints in gemm10.hip),
and the loop lives inside the statement.

Per K-tile and wave: 128 MFMAs (two k-steps of 64: 8 W fragments x 8 A fragments), 32 ds_read_b128 (0.25 per MFMA), 16 KiB of
operands to stage, one s_barrier.  LDS image = gemm8_kernel's (128-byte rows, 16-byte chunk c of row r at c ^ ((r >> 1) & 7);
A0 | A1 | W0 | W1 half-tiles of 16 KiB, two K-tile buffers = 128 KiB), same fragment -> k mapping, same K order: the sums are
gemm8_kernel<.., M16>'s bit for bit.

Operand staging (LOAD = "reg"): buffer_load_dwordx4 into 16 x 4 staging registers one K-tile ahead, ds_write_b128 into the ring.
An LDS-DMA request costs the issuing wave ~60 cycles (MI355X_MICROARCH.md), four MFMA slots of a wave that is alone on its SIMD;
a register load + ds_write_b128 pair is two ordinary issue slots.  LOAD = "dma" keeps the LDS-DMA form for the A/B.

Schedule of iteration T (tile T in buffer b = T & 1; X / Y = the two fragment register sets):
  k-step 0 (MFMAs on X):  16 ds_read  Y <- tile T, k-step 1 (buffer b)
                          8 x [vmcnt(15); ds_write piece j of tile T+1 -> buffer b^1; buffer_load piece j of tile T+2]   j = 8..15 (W)
                          lgkmcnt(0); s_barrier        -> tile T+1 published, buffer b free
  k-step 1 (MFMAs on Y):  16 ds_read  X <- tile T+1, k-step 0 (buffer b^1)
                          8 x [vmcnt(15); ds_write piece j of tile T+2 -> buffer b; buffer_load piece j of tile T+3]     j = 0..7 (A)
                          lgkmcnt(0)
Every load has one whole K-tile (~2 k cycles) to land; 16 loads are always in flight, so the counted wait is vmcnt(15) throughout.

Register map (gemm10.hip pins the statement's operands to it):
  a[0:255]    accumulators: 32 x 32 block (nf, mf) at a[16 (4 nf + mf)], 16 x 16 quad q = 2 n16 + m16 at + 4 q  (store_tile's layout)
  v[16:23]    global byte offsets of this wave's 8 A pieces (8 rows x 128 B each), v[24:31] of its 8 W pieces
  v[32:35]    LDS read address of A fragments: buffer 0 k-step 0 / 1, buffer 1 k-step 0 / 1;  v[36:39] the same for W
  v[40:43]    LDS write address: buffer 0 even / odd piece, buffer 1 even / odd piece
  v[64:127]   staging: piece j at v[64 + 4 j]
  v[128:191]  fragment set X: W fragment n at v[128 + 4 n], A fragment m at v[160 + 4 m];  v[192:255] set Y likewise
  s[40:43]    buffer descriptor of the A tile, s[44:47] of the W tile
  s48         byte offset of K-tile 0 in a row, s49 number of K-tiles (>= 1), s50 = s48 + 128 (nk - 1)
  s[52:55]    scratch (offsets of the next two load tiles, loop counter)
"""
import os
import sys

X_W, X_A, Y_W, Y_A = 128, 160, 192, 224
STG = 64


def acc(n16g, m16g):
    """First AGPR of the 16 x 16 accumulator of global 16-blocks (n16g, m16g) of the wave's 128 x 128 tile."""
    nf, n16, mf, m16 = n16g >> 1, n16g & 1, m16g >> 1, m16g & 1
    return (nf * 4 + mf) * 16 + (2 * n16 + m16) * 4


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(base, n=4):
    return f"a[{base}:{base + n - 1}]"


class Emit:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)

    def text(self):
        return "".join(f'"{l}\\n"\n' for l in self.lines)


def mfma(e, wbase, abase, n, m, zero_c):
    c = acc(n, m)
    e(f"v_mfma_f32_16x16x32_bf16 {ar(c)}, {vr(wbase + 4 * n)}, {vr(abase + 4 * m)}, {'0' if zero_c else ar(c)}")


def kstep(e, cfg, step, buf, zero_c, no_stage=False):
    """One k-step = 64 MFMAs + its share of reads, stores and loads.  step 0: MFMAs on X, reads into Y; step 1: the reverse.
    cfg["ablate"] (measurement only, results wrong): "reads" / "stage" / "barrier" / "vmcnt" drop that ingredient."""
    ab = cfg.get("ablate", ())
    if step == 0:
        m_w, m_a, r_w, r_a = X_W, X_A, Y_W, Y_A
        rd_a, rd_w = 32 + 2 * buf + 1, 36 + 2 * buf + 1          # this tile's k-step 1, buffer b
        wbuf, pieces, soff = buf ^ 1, range(8, 16), "s52"         # W pieces of tile T+1 -> buffer b^1; loads of tile T+2
    else:
        m_w, m_a, r_w, r_a = Y_W, Y_A, X_W, X_A
        rd_a, rd_w = 32 + 2 * (buf ^ 1), 36 + 2 * (buf ^ 1)      # next tile's k-step 0, buffer b^1
        wbuf, pieces, soff = buf, range(0, 8), "s53"              # A pieces of tile T+2 -> buffer b; loads of tile T+3
    fill = {s: [] for s in range(64)}
    # fragment reads in the order the next k-step's MFMAs need them: every a[m] and w[0] first (MFMA order: n outer, m inner)
    reads = [(r_a + 4 * i, rd_a, 2048 * i) for i in range(8)] + [(r_w + 4 * i, rd_w, 2048 * i) for i in range(8)]
    if "reads" not in ab:
        for i, (dst, addr, off) in enumerate(reads):
            fill[cfg["read_slots"][i]].append(f"ds_read_b128 {vr(dst)}, v{addr} offset:{off}")
    for k, j in enumerate(pieces):
        is_w = j >= 8
        i = j - 8 if is_w else j
        wr = 40 + 2 * wbuf + (i & 1)
        lds_off = (32768 if is_w else 0) + i * 1024
        voff = (24 if is_w else 16) + i
        desc = "s[44:47]" if is_w else "s[40:43]"
        if "stage" in ab or no_stage:
            continue
        if "vmcnt" not in ab and "loads" not in ab:
            fill[cfg["write_slots"][k]].append("s_waitcnt vmcnt(15)")
        if "writes" not in ab:
            fill[cfg["write_slots"][k]].append(f"ds_write_b128 v{wr}, {vr(STG + 4 * j)} offset:{lds_off}")
        if "loads" not in ab:
            fill[cfg["load_slots"][k]].append(f"buffer_load_dwordx4 {vr(STG + 4 * j)}, v{voff}, {desc}, {soff} offen")
    # scalar bookkeeping of the loop (once per tile, in k-step 1): advance the two load offsets, clamped to the last K-tile
    if step == 1:
        fill[61].append("s_add_u32 s52, s52, 128")
        fill[61].append("s_min_u32 s52, s52, s50")
        fill[63].append("s_add_u32 s53, s53, 128")
        fill[63].append("s_min_u32 s53, s53, s50")
    if cfg.get("m32_probe"):
        # MEASUREMENT ONLY (results wrong: the 32 x 32 x 16 MFMAs take the 16 x 16 x 32 fragments as they are): the first
        # `m32_probe` of the 16 accumulator blocks run as 2 x v_mfma_f32_32x32x16_bf16 per k-step instead of 4 x 16x16x32 (same
        # pipe time, ~20 cycles of issue shadow each instead of ~4), and every store / load sits behind one of them
        heavy = [x for sl in sorted(fill) for x in fill[sl] if x.startswith(("ds_write", "buffer_load", "s_waitcnt vmcnt"))]
        light = [x for sl in sorted(fill) for x in fill[sl] if not x.startswith(("ds_write", "buffer_load", "s_waitcnt vmcnt"))]
        units = []          # heavy instructions grouped with their counted wait
        for x in heavy:
            if units and units[-1][-1].startswith("s_waitcnt vmcnt"):
                units[-1].append(x)
            else:
                units.append([x])
        for b in range(16):
            nf, mf = b >> 2, b & 3
            if b < cfg["m32_probe"]:
                for half in range(2):
                    c = 16 * b
                    e(f"v_mfma_f32_32x32x16_bf16 {ar(c, 16)}, {vr(m_w + 4 * (2 * nf + half))}, {vr(m_a + 4 * (2 * mf + half))}, {'0' if zero_c and half == 0 else ar(c, 16)}")
                    if units:
                        for x in units.pop(0):
                            e(x)
            else:
                for q in range(4):
                    mfma(e, m_w, m_a, 2 * nf + (q >> 1), 2 * mf + (q & 1), zero_c)
                    if light:
                        e(light.pop(0))
        for u in units:
            for x in u:
                e(x)
        for x in light:
            e(x)
        e("s_waitcnt lgkmcnt(0)")
        if step == 0 and "barrier" not in ab:
            e("s_barrier")
        return
    slot = 0
    for n in range(8):
        for m in range(8):
            mfma(e, m_w, m_a, n, m, zero_c)
            for f in fill[slot]:
                e(f)
            if cfg.get("pad") and not fill[slot]:      # measurement: one cheap instruction in every otherwise empty gap
                e(cfg["pad"])
            slot += 1
    e("s_waitcnt lgkmcnt(0)")
    if step == 0 and "barrier" not in ab:
        e("s_barrier")


def tile(e, cfg, buf, first=False):
    kstep(e, cfg, 0, buf, zero_c=first, no_stage=first)   # tile 0: the prologue has published tile 1 whole and has tile 2 in flight
    kstep(e, cfg, 1, buf, zero_c=False)


def prologue(e, cfg):
    """Tiles 0 and 1 into the ring, tile 2 in flight, fragment set X of tile 0 -- ONE memory latency: all 32 loads of tiles 0 and 1
    are issued at once (tile 1 into the fragment registers of set Y, which nothing uses yet)."""
    def ld(dst, j, soff):
        is_w, i = j >= 8, j & 7
        e(f"buffer_load_dwordx4 {vr(dst)}, v{(24 if is_w else 16) + i}, {'s[44:47]' if is_w else 's[40:43]'}, {soff} offen")

    def st(src, j, buf):
        is_w, i = j >= 8, j & 7
        e(f"ds_write_b128 v{40 + 2 * buf + (i & 1)}, {vr(src)} offset:{(32768 if is_w else 0) + i * 1024}")
    e("s_nop 4")                                  # an operand SGPR written by VALU (v_readfirstlane) right before the statement
    if cfg.get("timed"):
        e("s_memtime s[60:61]")                   # measurement forms: statement entry (s[60:61] is an operand then)
    e("s_mov_b32 s52, s48")                       # K-tile 0
    e("s_add_u32 s53, s48, 128")
    e("s_min_u32 s53, s53, s50")                  # K-tile min(1, nk - 1)
    for j in range(16):
        ld(STG + 4 * j, j, "s52")                 # tile 0 -> staging
    for j in range(16):
        ld(Y_W + 4 * j, j, "s53")                 # tile 1 -> v[192:255]
    e("s_add_u32 s52, s53, 128")
    e("s_min_u32 s52, s52, s50")                  # K-tile min(2, nk - 1)
    e("s_add_u32 s53, s52, 128")
    e("s_min_u32 s53, s53, s50")                  # K-tile min(3, nk - 1)
    e("s_waitcnt vmcnt(16)")
    for j in range(16):
        st(STG + 4 * j, j, 0)                     # tile 0 -> buffer 0
    for j in range(16):
        ld(STG + 4 * j, j, "s52")                 # tile 2 -> staging (stays in flight into the loop)
    e("s_waitcnt vmcnt(16)")
    for j in range(16):
        st(Y_W + 4 * j, j, 1)                     # tile 1 -> buffer 1
    # loop invariant at the top of iteration T: 16 loads in flight (A pieces of the older tile first), s52 = row offset of tile
    # T+2 (its W pieces are loaded in k-step 0), s53 = of tile T+3 (A pieces, k-step 1).  Tile 0's k-step 0 stages nothing
    # (tile 1 is complete, tile 2 in flight); its k-step 1 bookkeeping moves s52 on to tile 3 like every other iteration's.
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    for i in range(8):
        e(f"ds_read_b128 {vr(X_W + 4 * i)}, v36 offset:{2048 * i}")
    for i in range(8):
        e(f"ds_read_b128 {vr(X_A + 4 * i)}, v32 offset:{2048 * i}")
    e("s_waitcnt lgkmcnt(0)")


def generate(cfg):
    e = Emit()
    timed = cfg.get("timed", False)               # measurement forms: s_memtime either side of the loop (s[56:59] are operands then)
    prologue(e, cfg)
    if timed:
        e("s_memtime s[56:57]")
    # tile 0 in buffer 0 with zero C; then buffers alternate
    e("s_sub_u32 s54, s49, 1")                    # tiles left after tile 0
    tile(e, cfg, 0, first=True)
    e("s_cmp_eq_u32 s54, 0")
    e("s_cbranch_scc1 .Lg10_done_%=")
    e(".Lg10_loop_%=:")
    tile(e, cfg, 1)
    e("s_sub_u32 s54, s54, 1")
    e("s_cmp_eq_u32 s54, 0")
    e("s_cbranch_scc1 .Lg10_done_%=")
    tile(e, cfg, 0)
    e("s_sub_u32 s54, s54, 1")
    e("s_cmp_lg_u32 s54, 0")
    e("s_cbranch_scc1 .Lg10_loop_%=")
    e(".Lg10_done_%=:")
    if timed:
        e("s_memtime s[58:59]")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")                       # the surplus loads still target staging registers the compiler owns again
    e("s_nop 15")                                 # XDL write -> v_accvgpr_read of the epilogue: hipcc's hazard recognizer does
    e("s_nop 15")                                 # not see into the statement
    return e.text()


DEFAULT = dict(
    # slot (0..63 = the MFMA it follows) of each of the 16 fragment reads, the 8 stores and the 8 loads of a k-step: at most one
    # of them per gap (a ds_write_b128 or a buffer_load_dwordx4 costs ~12 cycles of a gap that hides ~4: profiles/r06_gemm10_cycles_callE.txt)
    read_slots=[0, 2, 4, 8, 10, 12, 16, 18, 20, 24, 26, 28, 32, 34, 36, 40],
    write_slots=[6, 14, 22, 30, 38, 44, 48, 52], load_slots=[7, 15, 23, 31, 39, 46, 50, 54],
)
# measurement forms (built only with -DFK_G10_EXPERIMENTS, selected by FK_G10_X=<n>; results of 1..5 are wrong by construction)
T = dict(DEFAULT, timed=True)
P6 = dict(read_slots=[0, 2, 4, 8, 10, 12, 16, 18, 20, 24, 26, 28, 32, 34, 36, 40],
          write_slots=[6, 14, 22, 30, 38, 44, 48, 52], load_slots=[7, 15, 23, 31, 39, 46, 50, 54], timed=True)
P7 = dict(read_slots=list(range(16)), write_slots=[18, 22, 26, 30, 34, 38, 42, 46], load_slots=[20, 24, 28, 32, 36, 40, 44, 48], timed=True)
EXPERIMENTS = {
    1: T,                                                        # the shipped schedule, timed
    2: dict(T, ablate=("barrier",)),
    3: dict(T, ablate=("barrier", "stage", "reads")),            # MFMAs only
    4: dict(T, ablate=("barrier", "stage", "reads"), pad="s_nop 0"),
    5: dict(T, ablate=("barrier", "stage", "reads"), pad="v_mov_b32 v64, v64"),
    6: dict(T, ablate=("barrier", "stage")),                     # MFMAs + fragment reads
    7: dict(T, ablate=("barrier", "reads")),                     # MFMAs + loads + stores
    8: dict(T, ablate=("barrier", "reads", "loads")),            # MFMAs + ds_write_b128
    9: dict(T, ablate=("barrier", "reads", "writes")),           # MFMAs + buffer_load_dwordx4 (+ vmcnt)
    10: P6,
    11: P7,
    12: dict(T, pad="s_nop 0"),                                  # the shipped schedule with every empty gap padded
    13: dict(T, ablate=("vmcnt",)),
    14: dict(P6, pad="s_nop 0"),
    15: dict(T, ablate=("stage",)),                              # MFMAs + reads + barrier
    16: dict(T, ablate=("reads",)),                              # MFMAs + staging + barrier
}
# what a part of the MFMAs in the 32 x 32 x 16 form would buy (issue shadow for the stores / loads) and cost (power): forms 17-20
EXPERIMENTS.update({17: dict(T, m32_probe=4), 18: dict(T, m32_probe=8), 19: dict(T, m32_probe=16),
                    20: dict(T, m32_probe=8, ablate=("stage",))})


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm10_loop.inc")
    text = ("// GENERATED by gemm10_gen.py -- do not edit; regenerate with `python gemm10_gen.py` (tests/test_kernel_resources.py\n"
            "// checks that this file is what the generator writes).\n" + generate(DEFAULT))
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(0 if open(out).read() == text else 1)
    with open(out, "w") as f:
        f.write(text)
    print("wrote", out, text.count("\n"), "lines")
    if len(sys.argv) > 1 and sys.argv[1] == "--experiments":      # untracked files next to the shipped one
        for n, cfg in EXPERIMENTS.items():
            with open(out.replace(".inc", f"_x{n}.inc"), "w") as f:
                f.write(generate(cfg))


if __name__ == "__main__":
    main()
