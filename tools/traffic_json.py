"""profiles/r02_traffic_items.json (tools/pmc_traffic_summary.py) -> profiles/r02_traffic.json, the side file bench.py
reads for `roofline.traffic`: HBM bytes per launch of the dominant GEMM launch class of each workload.

    python tools/traffic_json.py profiles/r02_traffic
"""
import json
import sys

GEMM_OF = {   # workload -> (item-name prefix, what it is)
    "default": ("gemm 2560x9216x3072", "the fused-QKV shape 2560 x 9216 x 3072 (largest GEMM launch class of the 512^2 edit)"),
    "cfg2_single_512x512_28step": ("gemm 2560x9216x3072", "the fused-QKV shape 2560 x 9216 x 3072 (largest GEMM launch class of the 512^2 edit)"),
    "single_1024x1024_28step": ("gemm 8704x12288x3072", "the MLP-up shape 8704 x 12288 x 3072 (largest GEMM launch class of the 1024^2 edit)"),
    "cfg3_batch32_1024x1024_28step": ("gemm 278528x3072x3072", "the out-proj shape 278528 x 3072 x 3072 of the batch-32 edit"),
}


CLASSES = {   # workload -> {launch class: item-name prefix}
    "cfg2_single_512x512_28step": {"qkv": "gemm 2560x9216x3072", "mlp_up": "gemm 2560x12288x3072", "k_long": "gemm 2560x3072x15360",
                                   "out_proj": "gemm 2560x3072x3072"},
    "single_1024x1024_28step": {"qkv": "gemm 8704x9216x3072", "mlp_up": "gemm 8704x12288x3072", "k_long": "gemm 8704x3072x15360"},
}
CLASSES["default"] = CLASSES["cfg2_single_512x512_28step"]


def main(stem):
    src = json.load(open(stem + "_items.json"))
    items = src["items"]
    out = {"source": "profiles/" + stem.split("/")[-1] + ".md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_traffic.sh)",
           "gemm": {}, "attention": {}, "ln_modulate": {}, "vae": {}}
    for wl, (prefix, what) in GEMM_OF.items():
        name = next((k for k in items if k.startswith(prefix)), None)
        if name is None:      # item left out of this round's passes (TRAFFIC_SKIP of tools/traffic_target.py)
            continue
        it = items[name]
        b = it["hbm_read_bytes"] + it["hbm_write_bytes"]
        out["gemm"][wl] = dict(
            hbm_bytes_per_launch=b, algorithmic_bytes_per_launch=it["alg_bytes"], ratio=it["ratio"], l2_hit=it["l2_hit"],
            avg_us_profiled=it["avg_us"],
            note=f"{it.get('kernel', 'GEMM kernel')} on {what}: FETCH_SIZE x {src['bytes_per_fetch_unit']:.0f} B + WRITE_SIZE x "
                 f"{src['bytes_per_write_unit']:.0f} B per launch (units calibrated on a 1 GiB stream, same file), = "
                 f"{it['ratio']:.2f} x the algorithmic bytes (A + W read once, C written once); reads served by the 256 MiB "
                 f"Infinity Cache are counted, so this is traffic beyond the XCD L2s, an upper bound of the HBM bytes")
    for wl, cls in CLASSES.items():        # every launch class of the workload, not only its largest
        per = {}
        for cname, prefix in cls.items():
            hit = [k for k in items if k.startswith(prefix)]
            if hit:
                it = items[hit[0]]
                per[cname] = dict(hbm_bytes_per_launch=it["hbm_read_bytes"] + it["hbm_write_bytes"],
                                  algorithmic_bytes_per_launch=it["alg_bytes"], ratio=it["ratio"], l2_hit=it["l2_hit"],
                                  avg_us_profiled=it["avg_us"], tflops=it["tflops"], kernel=it.get("kernel", ""))
        if wl in out["gemm"]:
            out["gemm"][wl]["classes"] = per
    for fam in ("attention", "ln_modulate", "vae"):
        for name, it in items.items():
            if name.startswith(fam):
                b = it["hbm_read_bytes"] + it["hbm_write_bytes"]
                out[fam][name] = dict(hbm_bytes_per_launch=b, algorithmic_bytes_per_launch=it["alg_bytes"], ratio=it["ratio"],
                                      gbps_counted=b / (it["avg_us"] * 1e-6) / 1e9)
    json.dump(out, open(stem + ".json", "w"), indent=1)
    print(json.dumps(out["gemm"], indent=1)[:1500])


if __name__ == "__main__":
    main(sys.argv[1])
