"""Summarise the HBM-traffic PMC passes of tools/pmc_traffic.sh into profiles/r02_traffic.{md,json}.

    python tools/pmc_traffic_summary.py gpurun_out/traffic profiles/r02_traffic

Per item (tools/traffic_target.py): average duration, FETCH_SIZE / WRITE_SIZE per launch, the byte count they stand
for after the calibration on the 1 GiB copy, the ratio to the item's ALGORITHMIC bytes, achieved GB/s and L2 hit rate.
"""
import json
import os
import sys


def load(fn):
    rows = []
    for line in open(fn):
        if line.startswith("#"):
            continue
        p = line.rstrip("\n").split("\t")
        ctr = dict(kv.split("=") for kv in p[3].split()) if len(p) > 3 and p[3] else {}
        rows.append((p[1], float(p[2]), {k: float(v) for k, v in ctr.items()}))
    return rows


def groups_of(rows):
    out, cur = [], []
    for r in rows:
        if "silu_kernel" in r[0]:
            out.append(cur)
            cur = []
        else:
            cur.append(r)
    return out


def per_item(rows, items):
    """[{us, counters}] per item: the item's launches are the LAST `reps` rows of its group that match its kernel."""
    gs = groups_of(rows)[1:]          # group 0 = before the first separator
    res = []
    gi = 0
    for it in items:
        g = gs[gi]
        gi += 1
        if it["kernel"] == "*":       # whole-decoder item: the group before it is the warm-up decode
            g = gs[gi]
            gi += 1
            us = sum(r[1] for r in g) / it["reps"]
            ctr = {}
            for r in g:
                for k, v in r[2].items():
                    ctr[k] = ctr.get(k, 0.0) + v / it["reps"]
            fam = {}
            for r in g:
                key = r[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
                f = fam.setdefault(key, dict(us=0.0, n=0, ctr={}))
                f["us"] += r[1] / it["reps"]
                f["n"] += 1
                for k, v in r[2].items():
                    f["ctr"][k] = f["ctr"].get(k, 0.0) + v / it["reps"]
            res.append(dict(us=us, ctr=ctr, launches=len(g) // it["reps"], families=fam))
            continue
        sel = [r for r in g if it["kernel"] in r[0]]
        sel = sel[-it["reps"]:]
        us = sum(r[1] for r in sel) / it["reps"]
        ctr = {}
        for r in sel:
            for k, v in r[2].items():
                ctr[k] = ctr.get(k, 0.0) + v / it["reps"]
        res.append(dict(us=us, ctr=ctr, launches=len(sel) // it["reps"], kernel=sel[-1][0] if sel else ""))
    return res


def main(d, out_stem):
    items = json.load(open(os.path.join(d, "items.json")))
    passes = {p: per_item(load(os.path.join(d, p + ".txt")), items) for p in ("time", "fetch", "write", "hit")}
    try:   # the request-counter pass is optional (PMC_PASSES of tools/pmc_traffic.sh); it only feeds the calibration note
        passes["req"] = per_item(load(os.path.join(d, "req.txt")), items)
    except Exception:
        passes["req"] = None
    cal = items[0]
    f_cal = passes["fetch"][0]["ctr"]["FETCH_SIZE"]
    w_cal = passes["write"][0]["ctr"]["WRITE_SIZE"]
    # the calibration kernel (torch's vectorised `src * 1`, 16 B per lane) reads and writes exactly 1 GiB per launch
    kf = cal["alg_read_bytes"] / f_cal          # bytes per FETCH_SIZE unit
    kw = cal["alg_write_bytes"] / w_cal
    rq = passes["req"][0]["ctr"] if passes["req"] else {"(request-counter pass not run this round; round 3": 0, "TCC_EA0_RDREQ = bytes / 128)": 0}
    lines = ["# HBM traffic of the path's kernels from rocprofv3 PMC passes", "",
             "Command: `bash tools/pmc_traffic.sh` (one TCC counter group per pass, `--kernel-trace` only; target "
             "`tools/traffic_target.py`: every item launched 3x back to back, separated by a one-wave kernel). Raw "
             "per-dispatch rows: `gpurun_out/traffic/*.txt` on the build box.", "",
             f"**Calibration** on a known byte count (torch's vectorised `src * 1` over 1 GiB of bf16: 1 GiB read + 1 GiB "
             f"written per launch, 16 B per lane): FETCH_SIZE = {f_cal:.0f}, WRITE_SIZE = {w_cal:.0f} -> **{kf:.0f} B per "
             f"FETCH_SIZE unit, {kw:.0f} B per WRITE_SIZE unit** (the counters are nominally KiB; FETCH_SIZE tallies the "
             f"128-B read requests of a wide coalesced stream at 64 B each -- MI355X_MICROARCH.md section HBM -- hence 2048; "
             f"WRITE_SIZE is exact). Every kernel of this repo reads through 16-B-per-lane loads / LDS-DMA, the calibrated pattern. "
             f"Request counters of the same item: " + ", ".join(f"{k}={v:.4g}" for k, v in sorted(rq.items())) + ".",
             "All byte figures below apply these two factors. Reads served by the 256 MiB Infinity Cache are counted "
             "(MI355X_MICROARCH.md): for items whose operands fit it, 'HBM' means 'beyond the XCD L2'.", "",
             "| item | kernel(s) | avg us | algorithmic MB (rd + wr) | counted MB (rd + wr) | counted / algorithmic | "
             "achieved GB/s (counted) | algorithmic GB/s | TF/s | L2 hit |", "|---|---|---|---|---|---|---|---|---|---|"]
    js = {}
    for i, it in enumerate(items):
        us = passes["time"][i]["us"]
        rd = passes["fetch"][i]["ctr"].get("FETCH_SIZE", 0.0) * kf
        wr = passes["write"][i]["ctr"].get("WRITE_SIZE", 0.0) * kw
        h = passes["hit"][i]["ctr"]
        hit = h.get("TCC_HIT_sum", 0.0) / max(1.0, h.get("TCC_HIT_sum", 0.0) + h.get("TCC_MISS_sum", 0.0))
        alg = it["alg_read_bytes"] + it["alg_write_bytes"]
        kern = passes["time"][i].get("kernel", "")
        kern = kern.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
        if it["kernel"] == "*":
            kern = f"all {passes['time'][i]['launches']} launches"
        tf = it["flops"] / (us * 1e-6) / 1e12 if it["flops"] else 0.0
        lines.append(f"| {it['name']} | `{kern}` | {us:.1f} | {it['alg_read_bytes'] / 1e6:.1f} + {it['alg_write_bytes'] / 1e6:.1f} | "
                     f"{rd / 1e6:.1f} + {wr / 1e6:.1f} | {(rd + wr) / alg:.2f} | {(rd + wr) / (us * 1e-6) / 1e9:.0f} | "
                     f"{alg / (us * 1e-6) / 1e9:.0f} | {tf:.0f} | {hit * 100:.1f} % |")
        js[it["name"]] = dict(avg_us=us, alg_bytes=alg, hbm_read_bytes=rd, hbm_write_bytes=wr, ratio=(rd + wr) / alg,
                              l2_hit=hit, tflops=tf, note=it.get("note", ""), kernel=kern)
        if "families" in passes["time"][i]:
            lines_f = ["", f"### {it['name']}: by kernel family (per decode)", "",
                       "| family | launches | us | counted MB rd | counted MB wr | GB/s (counted) |", "|---|---|---|---|---|---|"]
            ft, ff, fw = passes["time"][i]["families"], passes["fetch"][i]["families"], passes["write"][i]["families"]
            for k, v in sorted(ft.items(), key=lambda kv: -kv[1]["us"]):
                r_ = ff.get(k, {"ctr": {}})["ctr"].get("FETCH_SIZE", 0.0) * kf
                w_ = fw.get(k, {"ctr": {}})["ctr"].get("WRITE_SIZE", 0.0) * kw
                lines_f.append(f"| {k[:60]} | {v['n'] // it['reps']} | {v['us']:.1f} | {r_ / 1e6:.1f} | {w_ / 1e6:.1f} | "
                               f"{(r_ + w_) / max(v['us'], 1e-9) / 1e3:.0f} |")
            js[it["name"]]["families"] = lines_f
    fam_lines = []
    for v in js.values():
        fam_lines += v.pop("families", [])
    lines += fam_lines
    lines += ["", "Notes per item:"] + [f"* {it['name']}: {it['note']}" for it in items if it.get("note")]
    open(out_stem + ".md", "w").write("\n".join(lines) + "\n")
    json.dump(dict(bytes_per_fetch_unit=kf, bytes_per_write_unit=kw, items=js), open(out_stem + "_items.json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
