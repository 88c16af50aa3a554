#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile.sh output) into profiles/<tag>_*.{csv,md,json}.

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in separate
passes, both in KiB; on gfx950 FETCH_SIZE tallies 128-B requests of wide coalesced reads at 64 B,
so the read side is reported both raw and doubled (upper bound); WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest(pattern):
    """the newest run of a pass (gpurun merges every call's output into the same directory; process ids repeat across boxes)"""
    files = glob.glob(pattern)
    return sorted(files, key=os.path.getmtime)[-1:] if files else []


def counters(d, kernel_subs):
    """mean per projector launch = sum over the launch's kernels (group kernel + straggler kernel)"""
    per = {ks: collections.defaultdict(list) for ks in kernel_subs}
    meta = {}
    for f in latest(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            for ks in kernel_subs:
                if ks in r["Kernel_Name"]:
                    per[ks][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta[ks] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                                  "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    out, n = collections.defaultdict(float), {}
    for ks in kernel_subs:
        for k, v in per[ks].items():
            out[k] += sum(v) / len(v)
            n[k] = len(v)
    return dict(out), n, meta


def main():
    tag = sys.argv[1]
    kernel_subs = (sys.argv[2] if len(sys.argv) > 2 else "project_fd_kernel,project_fd_flat_kernel,scout_kernel").split(",")
    kernel_sub = "+".join(kernel_subs)
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 262144          # units (projections / edges) per launch
    unit_bytes = int(sys.argv[4]) if len(sys.argv) > 4 else 225         # algorithmic bytes per unit (225 per projection)
    workload = sys.argv[5] if len(sys.argv) > 5 else "bench.py (C3)"   # what tools/profile.sh ran
    headline = workload.startswith("bench.py")                          # only the headline profile feeds bench.py's roofline.traffic
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = latest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
    krow = None
    parts = []
    for r in csv.DictReader(open(stats)):
        if any(ks in r["Name"] for ks in kernel_subs):
            parts.append((next(ks for ks in kernel_subs if ks in r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e6))
            if krow is None:
                krow = dict(r)
            else:  # one projector launch = one call of each kernel: durations add
                for k in ("AverageNs", "MinNs", "MaxNs"):
                    krow[k] = str(float(krow[k]) + float(r[k]))
    allc, ns, meta = {}, {}, {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mix"):
        c, n, m = counters(os.path.join(src, sub), kernel_subs)
        allc.update(c)
        ns.update(n)
        meta = m or meta
    fetch_kib, write_kib = allc.get("FETCH_SIZE"), allc.get("WRITE_SIZE")
    avg_ms = float(krow["AverageNs"]) / 1e6
    algo = unit_bytes * batch
    traffic_lo = (fetch_kib + write_kib) * 1024 if fetch_kib is not None and write_kib is not None else None
    traffic_hi = (2 * fetch_kib + write_kib) * 1024 if traffic_lo is not None else None
    summary = {
        "tag": tag, "kernel": kernel_sub, "batch": batch, "calls": int(krow["Calls"]), "avg_ms": avg_ms,
        "min_ms": float(krow["MinNs"]) / 1e6, "max_ms": float(krow["MaxNs"]) / 1e6,
        "projections_per_s_kernel": batch / (avg_ms * 1e-3),
        "algorithmic_bytes_per_launch": algo, "achieved_GBps_algorithmic": algo / (avg_ms * 1e-3) / 1e9,
        "counters_mean_per_launch": allc, "launch": meta, "kernels": parts,
        "hbm_bytes_per_launch_raw": traffic_lo, "hbm_bytes_per_launch_fetch_doubled": traffic_hi,
    }
    json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
    if traffic_hi is not None and headline:
        json.dump({"kernel": kernel_sub, "batch": batch, "hbm_bytes_per_launch": traffic_hi, "tag": tag,
                   "valu_wave_insts_per_launch": allc.get("SQ_INSTS_VALU"),
                   "note": "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, gfx950 FETCH_SIZE x2 correction "
                           "(upper bound for this kernel's 8-B-per-lane loads)"},
                  open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
    with open(os.path.join(dst, tag + "_summary.md"), "w") as f:
        f.write("# rocprofv3 summary `%s` — kernel `%s`, %d units per launch, workload %s\n\n" % (tag, kernel_sub, batch, workload))
        f.write("Command: `bash tools/profile.sh %s%s` (= `rocprofv3 --kernel-trace --stats -- python3 ...`, then separate `--pmc` "
                "passes; launch-record VGPR_Count is rocprofv3's figure = allocated VGPRs / 2 here, the ISA metadata holds the "
                "register count quoted in DESIGN.md).\n\n" % (tag, "" if headline else " " + workload))
        f.write("| quantity | value |\n|---|---|\n")
        f.write("| calls / avg / min / max | %d / %.3f ms / %.3f ms / %.3f ms |\n" % (summary["calls"], avg_ms, summary["min_ms"], summary["max_ms"]))
        f.write("| units/s (kernel only) | %.3e |\n" % summary["projections_per_s_kernel"])
        f.write("| algorithmic bytes per launch (%d B x %d) | %.1f MB -> %.3f GB/s = %.5f %% of 8 TB/s |\n"
                % (unit_bytes, batch, algo / 1e6, summary["achieved_GBps_algorithmic"], summary["achieved_GBps_algorithmic"] / 80.0))
        if traffic_lo is not None:
            f.write("| HBM traffic per launch (FETCH+WRITE raw / FETCH doubled) | %.1f MB / %.1f MB |\n" % (traffic_lo / 1e6, traffic_hi / 1e6))
        for name, calls, ms in parts:
            f.write("| kernel `%s` | %d calls, avg %.3f ms |\n" % (name[-60:], calls, ms))
        for ks in sorted(meta):
            f.write("| launch config `%s` | %s |\n" % (ks, ", ".join("%s=%s" % kv for kv in sorted(meta[ks].items()))))
        f.write("\n| counter (mean per launch, %s) | value |\n|---|---|\n" % ", ".join("%s x%d" % kv for kv in list(ns.items())[:1]))
        for k in sorted(allc):
            f.write("| %s | %.4g |\n" % (k, allc[k]))
        if "SQ_INSTS_VALU" in allc:
            issue = allc["SQ_INSTS_VALU"] * 4.0  # FP64-dominated: 4 cycles per wave64 instruction on a SIMD-32 at 16 FP64 lanes/clk
            avail = avg_ms * 1e-3 * 2.4e9 * 1024
            f.write("\nVALU issue roofline: %.3e VALU wave-instructions per launch x 4 cycles = %.3e SIMD-cycles of %.3e available "
                    "(1024 SIMDs x 2.4 GHz x %.2f ms) = **%.1f %%** of the FP64 issue ceiling.\n"
                    % (allc["SQ_INSTS_VALU"], issue, avail, avg_ms, 100 * issue / avail))
            summary["valu_issue_frac"] = issue / avail
            json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
        if "SQ_INSTS_VALU_FMA_F64" in allc:
            fl = (2 * allc["SQ_INSTS_VALU_FMA_F64"] + allc.get("SQ_INSTS_VALU_MUL_F64", 0) + allc.get("SQ_INSTS_VALU_ADD_F64", 0)) * 64
            f.write("\nExecuted FP64: %.3e flop per launch (FMA x2 + MUL + ADD, x64 lanes) = **%.1f TFLOP/s = %.1f %% of the 78.6 TFLOP/s "
                    "FP64 vector peak**; FP64 share of VALU instructions %.1f %%.\n"
                    % (fl, fl / (avg_ms * 1e-3) / 1e12, 100 * fl / (avg_ms * 1e-3) / 78.6e12,
                       100 * (allc["SQ_INSTS_VALU_FMA_F64"] + allc.get("SQ_INSTS_VALU_MUL_F64", 0) + allc.get("SQ_INSTS_VALU_ADD_F64", 0)
                              + allc.get("SQ_INSTS_VALU_TRANS_F64", 0)) / allc["SQ_INSTS_VALU"]))
            summary["executed_fp64_flop_per_launch"] = fl
            json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
            if headline:
                tj = json.load(open(os.path.join(dst, "traffic_latest.json")))
                tj["executed_fp64_flop_per_launch"] = fl
                json.dump(tj, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
        if "SQ_WAVE_CYCLES" in allc and "SQ_ACTIVE_INST_VALU" in allc:
            f.write("\nDerived: VALU-active share of wave lifetime = %.3f; " % (allc["SQ_ACTIVE_INST_VALU"] / allc["SQ_WAVE_CYCLES"]))
            if "SQ_WAIT_ANY" in allc:
                f.write("wave parked (s_waitcnt) share = %.3f; " % (allc["SQ_WAIT_ANY"] / allc["SQ_WAVE_CYCLES"]))
            if "SQ_WAIT_INST_ANY" in allc:
                f.write("issue-stall share = %.3f; " % (allc["SQ_WAIT_INST_ANY"] / allc["SQ_WAVE_CYCLES"]))
            if "GRBM_GUI_ACTIVE" in allc:
                f.write("effective clock = %.2f GHz." % (allc["GRBM_GUI_ACTIVE"] / 8 / (avg_ms * 1e-3) / 1e9))
            f.write("\n")
            if "GRBM_GUI_ACTIVE" in allc:
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
                busy = allc["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * allc["GRBM_GUI_ACTIVE"] / 8)
                f.write("\nVector pipes busy: SQ_ACTIVE_INST_VALU x 4 = %.3e SIMD-cycles of %.3e (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = "
                        "**%.1f %%** of the launch at the clock the chip actually held.\n"
                        % (allc["SQ_ACTIVE_INST_VALU"] * 4.0, 1024 * allc["GRBM_GUI_ACTIVE"] / 8, 100 * busy))
                if headline and os.path.exists(os.path.join(dst, "traffic_latest.json")):
                    tj = json.load(open(os.path.join(dst, "traffic_latest.json")))
                    tj["vector_pipes_busy_frac"] = busy
                    tj["effective_clock_ghz"] = allc["GRBM_GUI_ACTIVE"] / 8 / (avg_ms * 1e-3) / 1e9
                    json.dump(tj, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
    print(open(os.path.join(dst, tag + "_summary.md")).read())


if __name__ == "__main__":
    main()
