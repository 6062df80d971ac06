# SQ counters of the attention kernels for one or more builds.  usage: bash tools/pmc_attention.sh "tag:ENV=.." ...
# (SHAPE="B S"; KIND=attention | attention_bwd; one --pmc pass with --kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SHAPE=${SHAPE:-"4 8704"}
KIND=${KIND:-attention}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  env $envs timeout 120 rocprofv3 --kernel-trace --pmc $C -d /tmp/pa_$tag -o r -- python $REPO/tools/prof_one.py $KIND $SHAPE > /dev/null 2>&1
  echo "== $tag ($envs) $KIND $SHAPE"
  python - <<PY
import sqlite3
c = sqlite3.connect("/tmp/pa_$tag/r_results.db")
kernels = [r for r in c.execute("select name, total_calls, average from top_kernels") if "attention_" in r[0] and ("bwd" in r[0]) == ("$KIND" == "attention_bwd")]
cur = c.execute("select * from pmc_events limit 1"); cols = [d[0] for d in cur.description]
ix = {n: i for i, n in enumerate(cols)}
ni = ix.get("name", ix.get("kernel_name")); ci = ix.get("counter_name", ix.get("pmc_name", ix.get("symbol"))); vi = ix.get("value", ix.get("counter_value"))
rows = list(c.execute("select * from pmc_events"))
for name, calls, avg in kernels:
    agg = {}
    for r in rows:
        if str(r[ni]) != name: continue
        a = agg.setdefault(r[ci], [0.0, 0]); a[0] += float(r[vi]); a[1] += 1
    v = {k: s / n for k, (s, n) in agg.items()}
    wc = v["SQ_WAVE_CYCLES"]
    print(f"{name[:60]}  avg {avg:.0f} us  clock {v['GRBM_GUI_ACTIVE'] / avg / 1e3:.2f} GHz  mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 32 / v['GRBM_GUI_ACTIVE'] * 100:.1f}%")
    print("  wave-time: " + "  ".join(f"{k[3:]} {v[k] / wc * 100:.1f}%" for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU")) + f"  wave_cycles {wc / 1e6:.1f}M  lds_active {v['SQ_LDS_IDX_ACTIVE'] / 1e6:.2f}M conflicts {v['SQ_LDS_BANK_CONFLICT'] / 1e6:.2f}M  gui {v['GRBM_GUI_ACTIVE'] / 1e6:.3f}M")
PY
done
