# What a fabric byte costs a power-bound GEMM: the SAME gemm8 launch under tile orders that change only the
# operand reuse inside an XCD's L2 (fk_gemm_args.group_m: depth of the grouped order; 1 = one row tile per group, >= row tiles =
# one column range per XCD), with FETCH_SIZE (bytes beyond the L2s, x2 calibration of MI355X_MICROARCH.md), the clock
# (GRBM_GUI_ACTIVE / duration) and the rate side by side.   usage: SHAPE="8704 9216 3072" bash tools/pmc_gemm_tile_order.sh 1 2 4 8 16 64
cd /tmp && export TMPDIR=/tmp
SHAPE=${SHAPE:-"8704 9216 3072"}
for gm in "$@"; do
  FK_GEMM_BN=${BN:-256} FK_GEMM_GROUP_M=$gm timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d /tmp/to_$gm -o r -- python /root/repo/tools/prof_one.py gemm $SHAPE > /dev/null 2>&1
  python - <<PY
import sqlite3
c = sqlite3.connect("/tmp/to_$gm/r_results.db")
name, calls, avg = [r for r in c.execute("select name, total_calls, average from top_kernels") if "gemm" in r[0]][0]
cur = c.execute("select * from pmc_events limit 1"); cols = [d[0] for d in cur.description]
ix = {n: i for i, n in enumerate(cols)}
ni = ix.get("name", ix.get("kernel_name")); ci = ix.get("counter_name", ix.get("pmc_name", ix.get("symbol"))); vi = ix.get("value", ix.get("counter_value"))
agg = {}
for r in c.execute("select * from pmc_events"):
    if name[:40] not in str(r[ni]): continue
    a = agg.setdefault(r[ci], [0.0, 0]); a[0] += float(r[vi]); a[1] += 1
v = {k: s / n for k, (s, n) in agg.items()}
M, N, K = (int(x) for x in "$SHAPE".split())
alg = (M * K + N * K) * 2.0
fetch = v["FETCH_SIZE"] * 1024 * 2          # FETCH_SIZE is in KiB; gfx950 reports half the bytes of wide streaming reads
print(f"group_m {int('$gm'):3d}  {M}x{N}x{K}: {avg:7.1f} us  {2.0 * M * N * K / avg / 1e6:6.0f} TF/s  clock {v['GRBM_GUI_ACTIVE'] / avg / 1e3:.3f} GHz  "
      f"fetched beyond L2 {fetch / 1e6:7.1f} MB = {fetch / alg:5.2f} x the operands ({fetch / avg / 1e3:5.0f} GB/s)  [{name[:34]}]")
PY
done
