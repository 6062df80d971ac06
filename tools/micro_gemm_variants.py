import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from gpt_image_edit_amd import ops, libfk
torch.manual_seed(0)
for (M, N, K) in [(8192, 12288, 3072), (32768, 3072, 12288), (2560, 12288, 3072)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): ops.gemm(a, w, b, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 60
    for _ in range(n): ops.gemm(a, w, b, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"variant {libfk.load().fk_gemm_last_variant()} {M}x{N}x{K}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.0f} TF/s", flush=True)
