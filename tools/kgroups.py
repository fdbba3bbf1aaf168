"""Group the kernels of a rocprofv3 --kernel-trace database of a TRAINING run into families and write the JSON that bench.py
attaches to its `train` object (profiles/r02_train_step_kernel_groups.json):

    python tools/kgroups.py <rocprof dir> <steps in the trace> <out.json>
"""
import glob
import json
import re
import sqlite3
import sys

FAMILIES = [
    ("wgrad (filter gradients)", r"wgrad"),
    ("bn / activation passes", r"bn_act|channel_reduce|reduce_partials|bn_finalize|bn_stats"),
    ("conv forward + data gradient (implicit GEMM)", r"conv_igemm|conv_1x1s|conv_strip|stem_conv|stem_pair|bneck_pair|conv_direct|conv_v10_reduce"),
    ("filter packing", r"pack_filter|pack_dgrad"),
    ("loss", r"loss_"),
    ("optimizer (fused SGD / clip / EMA)", r"sgd|grad_norm|clip_coef"),
    ("pool / upsample / layout / decode", r"maxpool|spp|upsample|nchw|nhwc|decode|detect_raw|copy_slice"),
]


def main(d, steps, out):
    db = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kt = "kernels" if "kernels" in tables else next(t for t in tables if "kernel" in t.lower())
    rows = db.execute(f"select name, count(*), sum(end-start) from {kt} group by name").fetchall()
    tot = sum(r[2] for r in rows)
    groups = {name: {"ms_per_step": 0.0, "launches_per_step": 0.0} for name, _ in FAMILIES}
    groups["other (torch glue)"] = {"ms_per_step": 0.0, "launches_per_step": 0.0}
    per_kernel = []
    for n, c, s in rows:
        fam = next((name for name, pat in FAMILIES if re.search(pat, n)), "other (torch glue)")
        groups[fam]["ms_per_step"] += s / 1e6 / steps
        groups[fam]["launches_per_step"] += c / steps
        per_kernel.append((s, re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:80], c))
    for g in groups.values():
        g["share"] = round(g["ms_per_step"] * steps * 1e6 / tot, 4)
        g["ms_per_step"] = round(g["ms_per_step"], 3)
        g["launches_per_step"] = round(g["launches_per_step"], 1)
    per_kernel.sort(reverse=True)
    dom = per_kernel[0]
    json.dump({"_doc": "rocprofv3 --kernel-trace of tools/train_bench.py --batch 64 --fused (yolov3 640x640 autocast fp16), kernel time per training step by family",
               "kernel_ms_per_step": round(tot / 1e6 / steps, 3), "groups": groups,
               "dominant": {"kernel": dom[1], "share": round(dom[0] / tot, 4), "avg_us": round(dom[0] / dom[2] / 1e3, 1), "launches_per_step": round(dom[2] / steps, 1)}},
              open(out, "w"), indent=1)
    print(json.dumps(groups, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]), sys.argv[3])
