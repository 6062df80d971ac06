"""Dump a rocprofv3 result directory (rocpd sqlite `*_results.db`) as one line per kernel dispatch, in dispatch order:
    <index> <kernel name (100 chars)> <duration_us> <counter>=<value> ...
Version-tolerant: table / column names are discovered, not assumed."""
import glob
import sqlite3
import sys


def main(d, out):
    dbs = glob.glob(d + "/**/*results.db", recursive=True) + glob.glob(d + "/**/*.db", recursive=True)
    if not dbs:
        print("no database under", d)
        open(out, "w").write("# no database\n")
        return
    c = sqlite3.connect(dbs[0])
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]

    def cols(t):
        return [r[1] for r in c.execute(f"pragma table_info('{t}')")]

    kt = next((n for n in names if n == "kernels"), None) or next((n for n in names if "kernel_dispatch" in n and "rocpd" not in n), None) \
        or next((n for n in names if "kernel_dispatch" in n), None)
    lines = []
    if kt:
        kc = cols(kt)
        name_c = next((x for x in ("name", "kernel_name", "kernel") if x in kc), None)
        st = next((x for x in ("start", "start_timestamp") if x in kc), None)
        en = next((x for x in ("end", "end_timestamp") if x in kc), None)
        did = next((x for x in ("dispatch_id", "id") if x in kc), None)
        rows = list(c.execute(f"select {did}, {name_c}, {st}, {en} from {kt} order by {st}"))
    else:
        rows = []
    pm = {}
    if "pmc_events" in names:
        pc = cols("pmc_events")
        ix = {n: i for i, n in enumerate(pc)}
        di = next((ix[x] for x in ("dispatch_id", "event_id", "id") if x in ix), None)
        ci = next((ix[x] for x in ("counter_name", "pmc_name", "symbol", "name") if x in ix and x != "name"), None)
        if ci is None:
            ci = ix.get("counter_name")
        vi = next((ix[x] for x in ("value", "counter_value") if x in ix), None)
        for r in c.execute("select * from pmc_events"):
            pm.setdefault(r[di], {}).setdefault(str(r[ci]), 0.0)
            pm[r[di]][str(r[ci])] += float(r[vi])
    with open(out, "w") as f:
        f.write(f"# db {dbs[0]} kernel table {kt} ({len(rows)} dispatches), tables: {', '.join(names)[:400]}\n")
        if "pmc_events" in names:
            f.write(f"# pmc_events columns: {cols('pmc_events')}\n")
        for i, (d_id, name, s, e) in enumerate(rows):
            ctr = " ".join(f"{k}={v:.0f}" for k, v in sorted(pm.get(d_id, {}).items()))
            f.write(f"{i}\t{str(name)[:100]}\t{(e - s) / 1e3:.2f}\t{ctr}\n")
    print("wrote", out, len(rows), "dispatches", len(pm), "with counters")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
