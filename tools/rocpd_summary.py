"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel stats table committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_bench_kernel_stats.md

Durations are in microseconds (rocprofv3 --kernel-trace --stats; `top_kernels` view of the database).
"""
import sqlite3
import sys


def main(db, out, title=""):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    fam = {}
    for name, calls, tot, avg, pct in rows:
        key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
        f = fam.setdefault(key, [0, 0.0])
        f[0] += calls
        f[1] += tot
    total = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary{(': ' + title) if title else ''}\n\n")
        f.write(f"source db: `{db}`; total kernel time {total / 1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches\n\n")
        f.write("## by kernel family\n\n| family | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for k, (calls, tot) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            f.write(f"| {k} | {calls} | {tot / 1e3:.2f} | {tot / calls:.1f} | {100 * tot / total:.1f} |\n")
        f.write("\n## by kernel\n\n| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")
            f.write(f"| `{short[:110]}` | {calls} | {tot:.0f} | {avg:.1f} | {pct:.2f} |\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main(*sys.argv[1:4])
