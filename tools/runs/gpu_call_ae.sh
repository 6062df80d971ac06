cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02ae_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02ae_tests.log ); tail -3 gpurun_out/r02ae_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
