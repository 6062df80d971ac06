cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -q -m gpu -x --durations=5 > gpurun_out/r03d_tests.log 2>&1; echo "pytest rc=$?" ); tail -10 gpurun_out/r03d_tests.log
( FK_ATTN_TAIL=0 timeout 200 python tools/ab_attention.py tail0 > gpurun_out/r03d_ab_attn.log 2>&1; FK_ATTN_TAIL=1 timeout 200 python tools/ab_attention.py tail1 >> gpurun_out/r03d_ab_attn.log 2>&1; echo "ab_attn rc=$?" ); grep attention gpurun_out/r03d_ab_attn.log
( AB_ARMS="tail=0;tail=1" timeout 500 python tools/ab_edit_plans.py single_1024x1024_28step 3 1 > gpurun_out/r03d_ab_edit_1024.log 2>&1; echo "ab_edit1024 rc=$?" ); tail -5 gpurun_out/r03d_ab_edit_1024.log
rm -rf gpurun_out/traffic
bash tools/pmc_traffic.sh > gpurun_out/r03d_traffic.log 2>&1
tail -6 gpurun_out/r03d_traffic.log
mkdir -p gpurun_out/profiles_new
python tools/pmc_traffic_summary.py gpurun_out/traffic gpurun_out/profiles_new/r03_traffic > gpurun_out/r03d_traffic_summary.log 2>&1 && python tools/traffic_json.py gpurun_out/profiles_new/r03_traffic > /dev/null 2>&1
echo "traffic summary rc=$?"
grep "^| gemm\|^| attention\|^| ln_mod" gpurun_out/profiles_new/r03_traffic.md | cut -c1-230
