"""world_size-2 CPU test (gloo) of the data-parallel harness: strided shard + one final all-gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gpt_image_edit_amd import dp
    r, lr, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    items = list(range(n_items))
    assert dp.shard(items, rank, world) == items[rank::world]           # the reference's strided shard
    assert dp.shard_indices(n_items, rank, world) == items[rank::world]

    def edit(item):  # stand-in for pipe(...).latents: value encodes the item id
        return torch.full((1, 4, 8), float(item), dtype=torch.bfloat16)

    full = dp.generate_sharded(edit, items)
    ok = full.shape == (n_items, 4, 8) and all(float(full[i, 0, 0]) == float(i) for i in range(n_items))
    local = dp.generate_sharded(edit, items, gather=False)
    ok = ok and local.shape[0] == len(items[rank::world])
    g = dp.all_gather_latents(torch.full((2, 3, 5), float(rank), dtype=torch.bfloat16))
    ok = ok and g.shape == (2 * world, 3, 5) and float(g[0, 0, 0]) == 0.0 and float(g[-1, 0, 0]) == float(world - 1)
    out_q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_shard_and_gather_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_unshard_order_single_process():
    from gpt_image_edit_amd import dp
    for n, w in [(8, 2), (12, 4), (8, 8), (4, 1)]:
        gathered = [it for r in range(w) for it in list(range(n))[r::w]]   # rank-major gathered order
        order = dp.unshard_order(n, w)
        assert [gathered[order[i]] for i in range(n)] == list(range(n))
    x = torch.randn(2, 3, 4)
    assert dp.all_gather_latents(x) is x  # no process group: identity


def test_bench_launches_itself_for_more_than_one_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher becomes `python -m torch.distributed.run ... bench.py --gpus N ...`
    (the reference's generators are torchrun programs: univa/eval/gedit/step1_gen_samples.py:82-92)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    argv = bench.self_launch_argv(8, ["--gpus", "8", "--steps", "2", "--warmup", "1"], 29511)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29511"
    i = argv.index(os.path.abspath(bench.__file__))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"]

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extra"])
    try:
        bench.main()
        raise AssertionError("main() must hand over to the launcher")
    except SystemExit as e:
        assert e.code == 7                                    # the launcher's return code is the bench's
    assert "--nproc-per-node=2" in seen["cmd"] and seen["cmd"][-7:] == ["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extra"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher a mismatching --gpus is refused, not re-launched
    monkeypatch.setenv("WORLD_SIZE", "4")
    try:
        bench.main()
        raise AssertionError("WORLD_SIZE=4 with --gpus 2 must be refused")
    except SystemExit as e:
        assert "WORLD_SIZE=4" in str(e.code)


def _ranks_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dist.init_process_group("gloo")
    out_q.put((rank, bench.ranks_seen(world, torch.device("cpu"), "gloo")))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_ranks_seen_comes_from_a_collective_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ranks_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, [0, 1]), (1, [0, 1])]


def test_bench_roofline_object_carries_counted_rates_and_stays_small():
    """`roofline_of` (no GPU needed: family sums in, JSON object out): the contract fields, the committed PMC traffic per launch
    class, the COUNTED GB/s of the attention / VAE kernels beside the algorithmic ones, and the slim form used for the extras."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    fam = {"gemm": dict(launches=5419, ms=760.0, flops=926e12, tflops=1218.4, algorithmic_bytes=0.0, gbps=0.0),
           "attention": dict(launches=1596, ms=117.0, flops=128e12, tflops=1094.0, algorithmic_bytes=1e11, gbps=855.0),
           "conv": dict(launches=62, ms=6.4, flops=3.6e12, tflops=560.0, algorithmic_bytes=4.3e9, gbps=680.0)}
    rl = bench.roofline_of(fam, "cfg2_single_512x512_28step")
    assert rl["bound"] == "mfma" and rl["unit"] == "TFLOP/s" and rl["peak"] == 2500.0
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12 and 0 < rl["frac"] < 1
    assert rl["traffic"] and rl["traffic"] > 119e6                      # counted bytes per fused-QKV launch exceed the algorithmic 119.5 MB
    assert set(rl["traffic_x_algorithmic"]) >= {"qkv", "mlp_up", "k_long", "out_proj"}
    for k in ("attention", "conv"):
        ok = rl["other_kernels"][k]
        assert ok["hbm_gbps_counted"] > 0 and 0 < ok["frac_of_hbm_peak_counted"] < 1 and ok["counted_x_algorithmic"] >= 1.0
    slim = bench._slim_roofline(rl)
    assert "traffic_note" not in slim and set(slim["other_kernels"]["attention"]) == {"ms_per_edit", "tflops", "hbm_gbps_algorithmic"}
    assert len(json.dumps(bench._compact(slim))) < 450
