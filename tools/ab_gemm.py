"""A/B two builds/modes of the GEMM: run with env A and env B in subprocesses, compare outputs bitwise, and
print TF/s of both.  Usage: python tools/ab_gemm.py "FK_LIB_PATH=build_ab/libfk_prev.so" "" (an older build against the
current one), or "FK_GEMM_BN=128" "FK_GEMM_BN=256" (tile override).  FK_AB_SHAPES="M,N,K,epi;..." replaces the shape list."""
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHAPES = [(2560, 9216, 3072, 0), (2560, 3072, 15360, 3), (2560, 12288, 3072, 1), (2051, 3072, 3072, 3), (32768, 3072, 12288, 0),
          (8704, 12288, 3072, 1)]


if os.environ.get("FK_AB_SHAPES"):   # "M,N,K,epi;M,N,K,epi;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["FK_AB_SHAPES"].split(";")]


def child(tag):
    sys.path.insert(0, ROOT)
    from gpt_image_edit_amd import ops
    from tools.bench_kernels import timeit
    BF = torch.bfloat16
    outs = {}
    for (M, N, K, epi) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        a = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(BF)
        w = ((torch.rand(N, K, generator=g, device="cuda") * 2 - 1) * 0.05).to(BF)
        b = (torch.rand(N, generator=g, device="cuda") * 2 - 1).to(BF)
        res = (torch.rand(1, M, N, generator=g, device="cuda") * 2 - 1).to(BF)
        gate = (torch.rand(1, N, generator=g, device="cuda") * 2 - 1).to(BF)
        kw = dict(epilogue=epi)
        if epi == 3:
            kw.update(res=res, gate=gate)
        ref = None
        for rep in range(3):   # repeated runs must agree with each other too (race screen)
            out = ops.gemm(a.view(1, M, K), w, b, **kw).clone()
            if ref is None:
                ref = out
            elif not torch.equal(ref, out):
                print(f"[{tag}] NON-DETERMINISTIC {M}x{N}x{K}", flush=True)
        outs[(M, N, K, epi)] = ref.cpu()
        o2 = torch.empty(1, M, N, device="cuda", dtype=BF)
        t = timeit(lambda: ops.gemm(a.view(1, M, K), w, b, out=o2, **kw))
        print(f"[{tag}] {M}x{N}x{K} epi{epi}: {2.0 * M * N * K / t / 1e12:.0f} TF/s", flush=True)
    torch.save(outs, f"/tmp/ab_{tag}.pt")


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    envs = sys.argv[1:3]
    for tag, e in zip("AB", envs):
        env = dict(os.environ)
        for kv in e.split(","):
            if kv:
                k, v = kv.split("=")
                env[k] = v
        subprocess.run([sys.executable, __file__, "--child", tag], env=env, check=True)
    A, B = torch.load("/tmp/ab_A.pt"), torch.load("/tmp/ab_B.pt")
    for k in A:
        same = torch.equal(A[k], B[k])
        d = (A[k].float() - B[k].float()).abs().max().item()
        print(f"compare {k}: bitwise_equal={same} max_abs_diff={d:.3e}")
