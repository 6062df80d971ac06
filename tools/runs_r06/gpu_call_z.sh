# Round 6, call Z: the library rebuilt from scratch from the last tree (make with no objects present): full GPU suite + smoke.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
sha256sum gpt_image_edit_amd/libfk_gfx950.so | tee gpurun_out/r06z_so.sha
( timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06z_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r06z_tests.log ); grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r06z_tests.log | tail -8
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06z_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r06z_smoke.log ); tail -3 gpurun_out/r06z_smoke.log
