"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/fk.h
declares (no compute calls without a GPU), and the product path fails loudly instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "fk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from gpt_image_edit_amd import libfk
    if not os.path.exists(libfk.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = libfk.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fk.h but not exported by libfk"
        assert name in libfk.SIGNATURES, f"{name} has no ctypes signature"
    assert set(libfk.SIGNATURES) == set(declared)
    assert lib.fk_version().decode().endswith("gfx950")
    assert lib.fk_groupnorm_ws_floats(2, 4096, 512) > 0


def test_struct_layouts_match_header():
    from gpt_image_edit_amd import libfk
    assert ctypes.sizeof(libfk.Rows) == 24
    # fk_gemm_args: 4 pointers + ... ; compare against the C compiler's view
    import subprocess
    import tempfile
    code = ('#include <stdio.h>\n#include "fk.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(fk_gemm_args), sizeof(fk_conv_args), '
            'sizeof(fk_rows), sizeof(fk_block_ws), sizeof(fk_double_block_weights), sizeof(fk_single_block_weights));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(code)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(libfk.GemmArgs), ctypes.sizeof(libfk.ConvArgs), ctypes.sizeof(libfk.Rows),
                     ctypes.sizeof(libfk.BlockWs), ctypes.sizeof(libfk.DoubleBlockWeights), ctypes.sizeof(libfk.SingleBlockWeights)]


def test_no_cpu_fallback():
    from gpt_image_edit_amd import ops
    a = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(a, a)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.silu(a)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gpt_image_edit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"


def test_pack_rope_is_the_per_pair_view_of_the_flux_tables():
    """ops.pack_rope (host-side, no GPU): FluxPosEmbed's [S,128] cos / sin -> the [S,64,2] table of the fused QKV
    epilogue; anything that is not repeated over the columns of a rotary pair is refused."""
    import pytest
    import torch
    from gpt_image_edit_amd import ops
    from gpt_image_edit_amd.transformer import rope_tables
    from oracle.helpers import prepare_latent_image_ids
    ids = torch.cat([torch.zeros(5, 3), prepare_latent_image_ids(3, 4)])
    cos, sin = rope_tables(ids)
    cs = ops.pack_rope(cos, sin)
    assert cs.shape == (17, 64, 2) and cs.dtype == torch.float32 and cs.is_contiguous()
    assert torch.equal(cs[..., 0], cos[:, 0::2]) and torch.equal(cs[..., 0], cos[:, 1::2])
    assert torch.equal(cs[..., 1], sin[:, 0::2]) and torch.equal(cs[..., 1], sin[:, 1::2])
    bad = cos.clone()
    bad[3, 1] += 1e-3
    with pytest.raises(ValueError):
        ops.pack_rope(bad, sin)
    with pytest.raises(ValueError):
        ops.pack_rope(cos[:, :64], sin[:, :64])


def test_second_stream_policy_follows_the_attention_grid():
    """HipFluxTransformer2DModel._overlap_pays (host logic): the single blocks' MLP-up GEMM goes to a second stream only
    where the attention grid wastes a good part of its last round (measured: pays at S = 5632 / 8704, B = 1; costs at
    S = 2560)."""
    from types import SimpleNamespace
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel as M
    m = SimpleNamespace(num_heads=24)
    assert not M._overlap_pays(m, 1, 2560)        # 240 workgroups: one round
    assert M._overlap_pays(m, 1, 5632)            # 528: 2.06 rounds
    assert M._overlap_pays(m, 1, 8704)            # 816: 3.19 rounds
    assert not M._overlap_pays(m, 4, 8704)        # 12.75 rounds: 2 % waste
    assert not M._overlap_pays(m, 32, 8704)


def test_committed_traffic_side_file_matches_its_source(tmp_path):
    """profiles/r02_traffic.json (what bench.py reports as roofline.traffic) must be exactly what tools/traffic_json.py
    derives from the committed counter summary profiles/r02_traffic_items.json -- no hand-edited numbers."""
    import json
    import shutil
    import subprocess
    import sys
    stems = [s for s in ("r06_traffic", "r05_traffic", "r04_traffic", "r03_traffic", "r02_traffic") if os.path.exists(os.path.join(ROOT, "profiles", s + "_items.json"))]
    for stem in stems:          # the newest one is what bench.py reads
        shutil.copy(os.path.join(ROOT, "profiles", stem + "_items.json"), tmp_path / "t_items.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_json.py"), str(tmp_path / "t")], check=True,
                       capture_output=True)
        got = json.load(open(tmp_path / "t.json"))
        want = json.load(open(os.path.join(ROOT, "profiles", stem + ".json")))
        got.pop("source"), want.pop("source")
        assert got == want, stem
        ent = want["gemm"]["cfg2_single_512x512_28step"]
        assert ent["hbm_bytes_per_launch"] > ent["algorithmic_bytes_per_launch"] > 0
        assert any(k in ent["note"] for k in ("gemm9_kernel", "gemm8_kernel", "gemm_mix_kernel"))   # counters of real kernels
    import bench
    assert os.path.basename(bench.TRAFFIC_FILE) == stems[0] + ".json"


def test_library_keeps_no_process_wide_launch_switches():
    """VERDICT r4 weak #8 / SURVEY section 8(b) ("no global mutable state except an init-once table"): nothing the library
    exports changes the numerics of LATER calls -- batch-invariant mode, forced launch forms, the MFMA shape and the
    attention grids are fields / arguments of the call (fk_gemm_args.variant / plan / group_m / mfma, `grid` / `passes`,
    fk_block_ws.gemm_* / attn_grid); the host's defaults live in ``ops.LAUNCH`` (Python, application layer)."""
    import subprocess
    from gpt_image_edit_amd import libfk, ops
    assert not [n for n in _declared_symbols() if re.search(r"_set_|_get_|_last_variant", n)]
    out = subprocess.run(["nm", "-D", "--defined-only", libfk.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert len([n for n in exported if n.startswith("fk_")]) >= 50
    assert not [n for n in exported if n.startswith("fk_") and ("_set_" in n or "_get_" in n or "_last_variant" in n)]
    # ... and no diagnostic record either (round 6): the launch form a call used comes back through the call's own OUT field
    assert "variant_used" in {f for f, _ in libfk.GemmArgs._fields_} and "gemm_variant_used" in {f for f, _ in libfk.BlockWs._fields_}
    assert {"variant", "plan", "group_m", "mfma"} <= {f for f, _ in libfk.GemmArgs._fields_}
    assert {"gemm_variant", "gemm_plan", "gemm_group_m", "gemm_mfma", "attn_grid"} <= {f for f, _ in libfk.BlockWs._fields_}
    # the setters of the host layer change ops.LAUNCH (and the epoch a captured graph keys on), nothing else
    before, e0 = dict(vars(ops.LAUNCH)), ops.launch_config_epoch()
    try:
        ops.gemm_set_plan(1); ops.gemm_set_variant(256); ops.gemm_set_mfma(32); ops.gemm_set_group_m(4); ops.attention_set_split(0)
        ops.attention_bwd_set_mode(0)
        assert (ops.LAUNCH.gemm_plan, ops.LAUNCH.gemm_variant, ops.LAUNCH.gemm_mfma, ops.LAUNCH.gemm_group_m) == (9, 256, 32, 4)
        assert (ops.LAUNCH.attn_grid, ops.LAUNCH.attn_bwd_passes) == (-1, 3) and ops.launch_config_epoch() == e0 + 6
        ops.attention_set_split(7)
        assert ops.LAUNCH.attn_grid == 7
        for bad in (lambda: ops.gemm_set_plan(8), lambda: ops.gemm_set_variant(100), lambda: ops.gemm_set_mfma(8),
                    lambda: ops.gemm_set_group_m(-1), lambda: ops.attention_set_split(-2), lambda: ops.attention_bwd_set_mode(2)):
            with pytest.raises(ValueError):
                bad()
    finally:
        for k, v in before.items():
            setattr(ops.LAUNCH, k, v)
