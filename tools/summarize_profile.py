#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into a per-kernel table."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, suffix):
    return sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))


def short(name):
    for tok in ("(anonymous namespace)::", "void "):
        name = name.replace(tok, "")
    name = name.split("(")[0]
    return name[:70]


def main(root):
    # timing
    for f in find(os.path.join(root, "trace"), "kernel_trace.csv"):
        dur = defaultdict(list)
        for r in csv.DictReader(open(f)):
            # one template instance often serves several layers: keep launches on different grids apart
            grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            dur[(short(r["Kernel_Name"]), grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("== kernel trace: %s" % os.path.relpath(f, root))
        print("%-72s %-16s %8s %12s %12s %12s" % ("kernel", "grid (threads)", "calls", "avg_us", "min_us", "total_ms"))
        for (k, grid), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            print("%-72s %-16s %8d %12.2f %12.2f %12.3f" % (k, grid, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, sum(v) / 1e6))
    # counters
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for f in find(os.path.join(root, sub), "counter_collection.csv"):
            agg = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                key = "%s [grid %s]" % (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("== counters: %s (per-dispatch averages)" % os.path.relpath(f, root))
            for k, cs in sorted(agg.items()):
                parts = ["%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())]
                print("%-90s n=%d %s" % (k, len(next(iter(cs.values()))), " ".join(parts)))


# qnn_last_kernel() tag -> regular expression of the kernel's demangled rocprofv3 name
TAG_PATTERNS = [
    (r"^mfma_f32_first_cin(\d)$", r"^k_conv_first_(lds|mfma)<\1,"),
    (r"^mfma_f32_stem_cin(\d)$", r"^k_conv_stem<\1,"),
    (r"^pw_i4_f32$", r"^k_conv_pw_f32<"),
    (r"^mfma_i4_halo64x64\+dense$", r"^k_conv_mfma_halo<\d+, \d+, (true|1)[,>]"),
    (r"^mfma_i4_halo64x64$", r"^k_conv_mfma_halo<\d+, \d+, (false|0)[,>]"),
    (r"^mfma_i(\d)_areg64x64\+dense$", r"^k_conv_mfma_areg<\1, \d, \d, \d, (true|1)>"),
    (r"^mfma_i(\d)_areg64x64$", r"^k_conv_mfma_areg<\1, \d, \d, \d(, (false|0))?>"),
    (r"^mfma_i(\d)_wres256x64$", r"^k_conv_mfma_wres<\1,"),
    (r"^mfma_i4_small_c(\d+)$", r"^k_conv_mfma_small<\1,"),
    (r"^mfma_i8x3_first_fixed$", r"^k_conv_first_fixed<"),
    (r"^mfma_i8_first_u8$", r"^k_conv_first_u8(_full)?<\d, \d, \w+, (false|0)(, \w+)?>"),
    (r"^mfma_i8_first_img255$", r"^k_conv_first_u8(_full)?<\d, \d, \w+, (true|1)(, \w+)?>"),
    (r"^strip_i4_c16_lds:res_none$", r"^k_conv_strip16_lds<(false|0),"),
    (r"^strip_i4_c16_lds:res_packed$", r"^k_conv_strip16_lds<(true|1),"),
    (r"^strip_i4_c16_lds$", r"^k_conv_strip16_lds<"),
    (r"^strip_i4_c(\d+)_s2(:res_\w+)?$", r"^k_conv_strip_s2<\1,"),
    (r"^strip_i4_c(\d+):res_none$", r"^k_conv_strip<\1, \d, 0,"),
    (r"^strip_i4_c(\d+):res_packed$", r"^k_conv_strip<\1, \d, 1,"),
    (r"^strip_i4_c(\d+):res_f32$", r"^k_conv_strip<\1, \d, 2,"),
    (r"^strip_i4_c(\d+)_proj(:res_\w+)?$", r"^k_conv_strip<\1, \d, 3,"),
    (r"^strip_i4_c(\d+)$", r"^k_conv_strip<\1,"),
    (r"^mfma_i(\d)_(\d+)x(\d+)$", None),          # tile sizes -> waves, handled below
    (r"^dense_i(\d)$", r"^k_dense_packed(_split)?<\1,"),
    (r"^dense_bin$", r"^k_dense_packed(_split)?<1,"),
    (r"^dense_f32$", r"^k_dense_f32in"),
    (r"^xnor_f32_cw(\d+)$", r"^k_conv_xnor_f32<\1>"),
    (r"^xnor_pk_cw(\d+)", r"^k_conv_xnor_pk<\1,"),
    (r"^ps_", r"^k_conv_ps<"),
    (r"^generic$", r"^k_conv_generic"),
]


def rocprof_names_for(tag, names):
    """`tag` is what qnn_last_kernel() reports, optionally with the residual kind bench.py's launch groups carry
    (`strip_i4_c16:res_packed`); a suffix no pattern knows is ignored."""
    import re
    if ":" in tag and not any(re.match(t, tag) for t, _ in TAG_PATTERNS):
        tag = tag.split(":")[0]
    for tpat, kpat in TAG_PATTERNS:
        m = re.match(tpat, tag)
        if not m:
            continue
        if kpat is None:
            bits, bm, bn = m.group(1), int(m.group(2)) // 64, int(m.group(3)) // 64
            kre = re.compile(r"^k_conv_mfma(16)?(_dma)?<%s, %d, %d," % (bits, bm, bn))
        else:
            kre = re.compile(kpat.replace(r"\1", m.group(1)) if m.groups() else kpat)
        return [n for n in names if kre.match(n)]
    return []


def traffic_json(root, out_path, bench_json=None):
    """Per-kernel average FETCH_SIZE / WRITE_SIZE (KiB per dispatch) and trace duration, keyed by the rocprofv3
    kernel name (`by_kernel`) and by the tag qnn_last_kernel() reports (`by_tag`, what bench.py looks up; a tag
    is only listed when exactly one profiled kernel matches it)."""
    import json
    res = {}
    for sub, key in (("pmc_fetch", "fetch_kb"), ("pmc_write", "write_kb")):
        for f in find(os.path.join(root, sub), "counter_collection.csv"):
            agg = defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                res.setdefault(k, {})[key] = sum(v) / len(v)
    for f in find(os.path.join(root, "trace"), "kernel_trace.csv"):
        dur = defaultdict(list)
        grids = defaultdict(set)
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            grids[short(r["Kernel_Name"])].add(r.get("Grid_Size_X", "?"))
        for k, v in dur.items():
            res.setdefault(k, {})["avg_us"] = sum(v) / len(v) / 1e3
            res[k]["calls"] = len(v)
            res[k]["grids"] = len(grids[k])     # > 1: one template instance served several layers, averages mix them
    tags = []
    if bench_json and os.path.exists(bench_json):
        for line in open(bench_json):
            line = line.strip()
            if line.startswith("{"):
                try:
                    for k in json.loads(line).get("kernels", []):
                        tags.append(k["kernel"])
                        if k.get("shape") and str(k["shape"][-1]).startswith("res_"):
                            tags.append("%s:%s" % (k["kernel"], k["shape"][-1]))
                except ValueError:
                    pass
    by_tag = {}
    for tag in dict.fromkeys(tags):
        names = rocprof_names_for(tag, list(res))
        if len(names) == 1 and res[names[0]].get("grids", 1) == 1:     # a per-launch figure only where it is one layer's
            by_tag[tag] = dict(res[names[0]], rocprof_kernel=names[0])
    json.dump({"by_tag": by_tag, "by_kernel": res}, open(out_path, "w"), indent=1, sort_keys=True)


def retag(traffic_path, bench_log):
    """Rebuild `by_tag` of an existing traffic.json from its `by_kernel` (after TAG_PATTERNS changed)."""
    import json
    d = json.load(open(traffic_path))
    res = d["by_kernel"]
    tags = []
    for line in open(bench_log):
        line = line.strip()
        if line.startswith("{"):
            try:
                for k in json.loads(line).get("kernels", []):
                    tags.append(k["kernel"])
                    if k.get("shape") and str(k["shape"][-1]).startswith("res_"):
                        tags.append("%s:%s" % (k["kernel"], k["shape"][-1]))
            except ValueError:
                pass
    by_tag = {}
    for tag in dict.fromkeys(tags):
        names = rocprof_names_for(tag, list(res))
        if len(names) == 1 and res[names[0]].get("grids", 1) == 1:
            by_tag[tag] = dict(res[names[0]], rocprof_kernel=names[0])
    d["by_tag"] = by_tag
    json.dump(d, open(traffic_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "--retag":
        retag(sys.argv[2], sys.argv[3])
        sys.exit(0)
    main(sys.argv[1])
    if len(sys.argv) > 2:
        traffic_json(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
