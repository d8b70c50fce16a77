#!/usr/bin/env python3
"""Turns the rocprofv3 PMC passes of tools/final_evidence_r4b.sh into the table of profiles/README.md and
into profiles/r02_traffic.json (HBM bytes per launch, which bench.py reports as roofline.traffic).

  python tools/pmc_table.py gpurun_out/final profiles/r02_traffic.json

Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is
doubled for wide coalesced reads on gfx950; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration;
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)."""
import collections
import csv
import glob
import json
import re
import sys


def short(n):
    # f32x3 kernels (csrc/gemm_f32emu.h): label + "_f32x3", what bench.py looks up for a region of that precision
    m = re.search(r"gemm_f32emu<.*?Prob(Conv2|Conv3|Fc)", n)
    if m:
        return {"Conv2": "conv2_mfma_f32x3", "Conv3": "conv3_mfma_f32x3", "Fc": "fc_mfma_f32x3"}[m.group(1)]
    # r5: the f32x3 trunk on split3 records (csrc/conv12_s3.h, conv_img_s3.h, gemm_s3.h)
    if "conv12_s3" in n:
        return "conv12_fused_f32x3"
    if "conv3_img_s3" in n:
        return "conv3_mfma_f32x3"
    m = re.search(r"gemm_s3<.*?ProbFcT<(\d+)", n)
    if m:
        return "fc_mfma_f32x3" if m.group(1) == "512" else "lstm_gates_x_f32x3"
    # split-bf16 path (names = the rela_prof labels bench.py looks traffic up by)
    if "conv12_i8" in n:
        return "conv12_fused"
    if "GemmCfg<3648" in n:
        return "lstm_gates_mfma"
    if "gemm_rec64_nt" in n:
        return "lstm_gates_x_bf16"
    m = re.search(r"conv_bf16s<.*?ConvFastCfg<(\d+)", n)
    if m:
        return "conv2_mfma" if m.group(1) == "32" else "conv3_mfma"
    if "fc_bf16s" in n:
        return "fc_mfma"
    if "conv1_bf16x3" in n:
        return "conv1_bf16x3"
    m = re.search(r"conv_mfma_bstat<.*?ConvCfg<(\d+)", n)
    if m:
        return "conv2_mfma" if m.group(1) == "32" else "conv3_mfma"
    m = re.search(r"ConvCfg<(\d+)", n)
    if m:
        return "conv2_mfma" if m.group(1) == "32" else "conv3_mfma"
    m = re.search(r"gemm_mfma<.*?GemmCfg<(\d+), (\d+)", n)
    if m:
        return {"3136_512": "fc_mfma", "512_32": "heads_mfma"}.get("%s_%s" % m.groups(), "gemm_%s_%s" % m.groups())
    m = re.search(r"::(\w+)[<(]", n)
    return m.group(1) if m else n[:40]


def main(root, out_json):
    res = {}
    for name in ("pmc_sq", "pmc_fetch", "pmc_write"):
        tr = list(csv.DictReader(open(glob.glob("%s/%s/*/*_kernel_trace.csv" % (root, name))[0])))
        dur = collections.defaultdict(list)
        for r in tr:
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for r in csv.DictReader(open(glob.glob("%s/%s/*/*_counter_collection.csv" % (root, name))[0])):
            res.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if name == "pmc_sq":
            for k, v in dur.items():
                res.setdefault(k, {})["us"] = [x / 1e3 for x in v]
    keep = ("conv12_fused", "conv12_fused_f32x3", "conv1_bf16x3", "conv2_mfma", "conv3_mfma", "fc_mfma", "conv2_mfma_f32x3",
            "conv3_mfma_f32x3", "fc_mfma_f32x3", "heads_mfma", "lstm_gates_mfma", "lstm_gates_x_f32x3",
            "lstm_gates_x_bf16", "slide_stacks", "replay_scatter_rows",
            "replay_gather_big", "seq_chain", "replay_search", "replay_finish", "replay_update", "replay_append_weights")
    traffic = {}
    print("every row is PER LAUNCH: averages over the `launches` launches of the kernel in the profiled run\n")
    print("| kernel | launches | us | GHz | MFMA busy | HBM read MB | HBM write MB | LDS bank-conflict cycles |")
    print("|---|---|---|---|---|---|---|---|")
    for k in keep:
        d = res.get(k)
        if not d or "us" not in d or "GRBM_GUI_ACTIVE" not in d:
            continue
        avg = lambda c: sum(d[c]) / len(d[c]) if c in d else float("nan")
        us, cyc = avg("us"), avg("GRBM_GUI_ACTIVE") / 8
        rd, wr = 2 * avg("FETCH_SIZE") * 1024, avg("WRITE_SIZE") * 1024
        traffic[k] = {"read_bytes": rd, "write_bytes": wr, "launch_us": us}
        print("| %s | %d | %.1f | %.2f | %.1f %% | %.1f | %.1f | %.0f |" % (
            k, len(d["us"]), us, cyc / us / 1e3, 100 * avg("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024), rd / 1e6, wr / 1e6,
            avg("SQ_LDS_BANK_CONFLICT")))
    if out_json:
        json.dump({"source": "rocprofv3 --pmc passes of tools/profile_forward.py (N = 6400), see tools/pmc_table.py",
                   "kernels": traffic}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
