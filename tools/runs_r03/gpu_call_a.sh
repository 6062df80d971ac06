cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r03a_tests.log 2>&1; echo "pytest rc=$?" ); tail -30 gpurun_out/r03a_tests.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03a_smoke.log 2>&1; echo "smoke rc=$?" ); tail -3 gpurun_out/r03a_smoke.log
