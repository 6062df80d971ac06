# Round 6, call N: would running part of gemm10's MFMAs as 32x32x16 (20 cycles of issue shadow instead of 4) pay for its power?
# Measurement forms 17-20 (wrong results by construction: fragments of the other shape), timed, against form 1 (the shipped loop).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2; do for X in 1 17 18 19 20; do
  FK_G10_X=$X timeout 120 python tools/g10_cycles.py 32768 3072 12288 2>&1 | grep -v amdgpu.ids | head -1
done; done > gpurun_out/r06n_gemm10_m32_probe.txt 2>&1
for X in 1 18; do FK_G10_X=$X timeout 120 python tools/g10_cycles.py 32768 12288 3072 2>&1 | grep -v amdgpu.ids | head -1; done >> gpurun_out/r06n_gemm10_m32_probe.txt 2>&1
cat gpurun_out/r06n_gemm10_m32_probe.txt
