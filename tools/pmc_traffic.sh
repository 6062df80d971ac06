# HBM-traffic counters of the path's kernels.  One TCC counter group per rocprofv3 pass
# (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: they cannot share a pass -- MI355X_MICROARCH.md), kernel-trace
# only, every pass under its own timeout so that a hung pass costs a minute, not the box.
#   bash tools/pmc_traffic.sh          -> gpurun_out/traffic/{items.json, fetch.txt, write.txt, hit.txt, time.txt}
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_pass() {  # tag, counters...
  tag=$1; shift
  rm -rf /tmp/tr_$tag
  if [ "$#" -gt 0 ]; then PMC="--pmc $*"; else PMC="--stats"; fi
  timeout 240 rocprofv3 --kernel-trace $PMC -d /tmp/tr_$tag -o r -- python $ROOT/tools/traffic_target.py $OUT/items.json > $OUT/$tag.log 2>&1
  echo "pass $tag rc=$?" >> $OUT/passes.txt
  python $ROOT/tools/pmc_dump.py /tmp/tr_$tag $OUT/$tag.txt >> $OUT/$tag.log 2>&1
}
: > $OUT/passes.txt
PASSES=${PMC_PASSES:-"time fetch write hit req"}   # subset to save GPU time, e.g. PMC_PASSES="time fetch write"
for pass in $PASSES; do
  case $pass in
    time) run_pass time ;;
    fetch) run_pass fetch FETCH_SIZE ;;
    write) run_pass write WRITE_SIZE ;;
    hit) run_pass hit TCC_HIT_sum TCC_MISS_sum ;;
    req) run_pass req TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum ;;
  esac
done
cat $OUT/passes.txt
