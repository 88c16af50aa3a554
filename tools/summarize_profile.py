#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile.sh output) into profiles/<tag>_*.{csv,md,json}.

    python tools/summarize_profile.py <tag> <kernel,...> <units per call> [bytes per unit] [workload text]

Round 5: statistics are taken over the PROFILED calls only — the warm-up calls' dispatches are dropped — and are
medians with quartiles, per kernel (from the kernel trace's per-dispatch rows) and per call (HIP events around each call, printed
by the workload itself: `CALL_MS` of tools/measure.py run, `roofline.kernel_ms_per_step` of bench.py).  A call whose kernels run on
two streams at once (split launches, the extend step's bulk form) is priced with that per-call WALL time, never with the sum of its
kernels' durations; and because counter collection serialises what production overlaps, the same per-call time is read from a
`--pmc` pass too and shown beside it.

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in separate passes, both in KiB; on gfx950
FETCH_SIZE tallies 128-B requests of wide coalesced reads at 64 B, so the read side is reported both raw and doubled (upper bound);
WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_CLOCK_GHZ = 2.4


def latest(pattern):
    """the newest run of a pass (gpurun merges every call's output into the same directory; process ids repeat across boxes)"""
    files = glob.glob(pattern, recursive=True)
    return sorted(files, key=os.path.getmtime)[-1:] if files else []


def quartiles(v):
    v = sorted(v)
    if not v:
        return None

    def q(p):
        x = p * (len(v) - 1)
        lo = int(x)
        hi = min(lo + 1, len(v) - 1)
        return v[lo] + (v[hi] - v[lo]) * (x - lo)
    return {"n": len(v), "median": q(0.5), "q1": q(0.25), "q3": q(0.75), "min": v[0], "max": v[-1]}


def call_times(log_path):
    """per-call wall times (ms) the workload printed, and how many warm-up calls ran in front of them"""
    if not os.path.exists(log_path):
        return None, None
    text = open(log_path, errors="replace").read()
    m = re.search(r"^CALL_MS (\{.*\})$", text, re.M)
    if m:
        j = json.loads(m.group(1))
        return j["call_ms"], j["warmup_calls"]
    for ln in reversed(text.splitlines()):  # bench.py's line
        if ln.startswith("{") and '"roofline"' in ln:
            j = json.loads(ln)
            return j["roofline"].get("kernel_ms_per_step"), j.get("warmup")
    return None, None


def profiled_rows(rows, key, warm_frac):
    """rows sorted by `key`, the warm-up calls' share dropped from the front"""
    rows = sorted(rows, key=key)
    return rows[int(round(len(rows) * warm_frac)):]


def counters(d, kernel_subs, warm_frac):
    """mean per call = sum over the call's kernels of (mean per profiled dispatch x dispatches per call)"""
    per = {ks: collections.defaultdict(list) for ks in kernel_subs}
    meta = {}
    for f in latest(os.path.join(d, "**", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            for ks in kernel_subs:
                if ks in r["Kernel_Name"]:
                    per[ks][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
                    meta[ks] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                                  "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
                    break
    out, n = collections.defaultdict(float), {}
    return per, out, n, meta


def main():
    tag = sys.argv[1]
    kernel_subs = (sys.argv[2] if len(sys.argv) > 2 else "project_fd_kernel,project_fd_flat_kernel,scout_kernel").split(",")
    kernel_sub = "+".join(kernel_subs)
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 262144          # units (projections / edges) per call
    unit_bytes = int(sys.argv[4]) if len(sys.argv) > 4 else 225         # algorithmic bytes per unit (225 per projection)
    workload = sys.argv[5] if len(sys.argv) > 5 else "bench.py (C3)"   # what tools/profile.sh ran
    headline = workload.startswith("bench.py")                          # only the headline profile feeds bench.py's roofline
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = latest(os.path.join(src, "trace", "**", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))

    calls_ms, warm = call_times(os.path.join(src, "trace.log"))
    n_calls = len(calls_ms) if calls_ms else 12
    warm = 3 if warm is None else warm
    warm_frac = warm / float(warm + n_calls)

    # ---- per kernel, from the trace's per-dispatch rows (profiled calls only) -----------------------------------------------------
    per_kernel = {}
    trace = latest(os.path.join(src, "trace", "**", "*_kernel_trace.csv"))
    if trace:
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(trace[0])):
            for ks in kernel_subs:
                if ks in r["Kernel_Name"]:
                    by[ks].append(r)
                    break
        for ks, rows in by.items():
            rows = profiled_rows(rows, lambda r: int(r["Start_Timestamp"]), warm_frac)
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
            per_kernel[ks] = dict(quartiles(d), per_call=len(d) / float(n_calls))
    dominant = max(per_kernel, key=lambda k: per_kernel[k]["median"] * per_kernel[k]["per_call"]) if per_kernel else None
    call_q = quartiles(calls_ms) if calls_ms else None
    # the call's wall time: what the workload measured around it; for the headline (bench.py steps overlap nothing) that is the
    # kernel sequence's event time per step
    wall_ms = call_q["median"] if call_q else sum(v["median"] * v["per_call"] for v in per_kernel.values())
    pmc_calls, _ = call_times(os.path.join(src, "pmc_sq.log"))
    pmc_q = quartiles(pmc_calls) if pmc_calls else None

    # ---- counters: mean per call over the profiled dispatches ----------------------------------------------------------------------
    allc, meta = {}, {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mix"):
        per, _, _, m = counters(os.path.join(src, sub), kernel_subs, warm_frac)
        meta.update(m)
        acc = collections.defaultdict(float)
        for ks in kernel_subs:
            for cname, vals in per[ks].items():
                vals = [v for _, v in profiled_rows(vals, lambda t: t[0], warm_frac)]
                if vals:
                    acc[cname] += sum(vals) / float(n_calls)  # dispatches per call x mean per dispatch
        allc.update(acc)
    fetch_kib, write_kib = allc.get("FETCH_SIZE"), allc.get("WRITE_SIZE")
    algo = unit_bytes * batch
    traffic_lo = (fetch_kib + write_kib) * 1024 if fetch_kib is not None and write_kib is not None else None
    traffic_hi = (2 * fetch_kib + write_kib) * 1024 if traffic_lo is not None else None

    summary = {
        "tag": tag, "kernel": kernel_sub, "batch": batch, "workload": workload, "profiled_calls": n_calls, "warmup_calls_dropped": warm,
        "call_ms": call_q, "call_ms_under_pmc": pmc_q, "wall_ms_used": wall_ms, "kernels": per_kernel, "dominant_kernel": dominant,
        "units_per_s": batch / (wall_ms * 1e-3),
        "algorithmic_bytes_per_call": algo, "achieved_GBps_algorithmic": algo / (wall_ms * 1e-3) / 1e9,
        "counters_mean_per_call": allc, "launch": meta,
        "hbm_bytes_per_call_raw": traffic_lo, "hbm_bytes_per_call_fetch_doubled": traffic_hi,
    }
    lines = []
    w = lines.append
    w("# rocprofv3 summary `%s` — kernel `%s`, %d units per call, workload %s\n" % (tag, kernel_sub, batch, workload))
    w("Command: `bash tools/profile.sh %s%s` (= `rocprofv3 --kernel-trace --stats -- python3 ...`, then separate `--pmc` passes; %d profiled "
      "calls behind %d warm-up calls whose dispatches are dropped; launch-record VGPR_Count is rocprofv3's figure = allocated VGPRs / 2 "
      "here, the ISA metadata — `closed_chain_motion_planner_amd/build/*.resources.json` — holds the register counts quoted in DESIGN.md).\n"
      % (tag, "" if headline else " " + workload, n_calls, warm))
    w("| quantity | median | q1 – q3 | min – max | n |\n|---|---|---|---|---|")
    if call_q:
        w("| one call, HIP events around it (kernel trace pass: kernels overlap as in production) | **%.3f ms** | %.3f – %.3f | %.3f – %.3f | %d |"
          % (call_q["median"], call_q["q1"], call_q["q3"], call_q["min"], call_q["max"], call_q["n"]))
    if pmc_q:
        w("| the same under `--pmc` (counter collection serialises the dispatches) | %.3f ms | %.3f – %.3f | %.3f – %.3f | %d |"
          % (pmc_q["median"], pmc_q["q1"], pmc_q["q3"], pmc_q["min"], pmc_q["max"], pmc_q["n"]))
    for ks, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["median"] * kv[1]["per_call"]):
        w("| kernel `%s` (%.3g per call) | %.3f ms | %.3f – %.3f | %.3f – %.3f | %d |"
          % (ks[-60:], v["per_call"], v["median"], v["q1"], v["q3"], v["min"], v["max"], v["n"]))
    w("")
    w("| quantity | value |\n|---|---|")
    w("| units/s at the call's median | %.3e |" % summary["units_per_s"])
    w("| algorithmic bytes per call (%d B x %d) | %.1f MB -> %.3f GB/s = %.5f %% of 8 TB/s |"
      % (unit_bytes, batch, algo / 1e6, summary["achieved_GBps_algorithmic"], summary["achieved_GBps_algorithmic"] / 80.0))
    if traffic_lo is not None:
        w("| HBM traffic per call (FETCH+WRITE raw / FETCH doubled) | %.1f MB / %.1f MB |" % (traffic_lo / 1e6, traffic_hi / 1e6))
    for ks in sorted(meta):
        w("| launch config `%s` | %s |" % (ks, ", ".join("%s=%s" % kv for kv in sorted(meta[ks].items()))))
    w("\n| counter (mean per call over the profiled calls) | value |\n|---|---|")
    for k in sorted(allc):
        w("| %s | %.4g |" % (k, allc[k]))
    if "SQ_INSTS_VALU" in allc:
        issue = allc["SQ_INSTS_VALU"] * 4.0  # FP64-dominated: 4 cycles per wave64 instruction on a SIMD at 16 FP64 lanes/clk
        avail = wall_ms * 1e-3 * PEAK_CLOCK_GHZ * 1e9 * 1024
        w("\nVALU issue roofline: %.3e VALU wave-instructions per call x 4 cycles = %.3e SIMD-cycles of %.3e available "
          "(1024 SIMDs x %.1f GHz x %.3f ms, the call's median wall time) = **%.1f %%** of the FP64 issue ceiling."
          % (allc["SQ_INSTS_VALU"], issue, avail, PEAK_CLOCK_GHZ, wall_ms, 100 * issue / avail))
        summary["valu_issue_frac"] = issue / avail
    if "SQ_INSTS_VALU_FMA_F64" in allc:
        fl = (2 * allc["SQ_INSTS_VALU_FMA_F64"] + allc.get("SQ_INSTS_VALU_MUL_F64", 0) + allc.get("SQ_INSTS_VALU_ADD_F64", 0)) * 64
        w("\nExecuted FP64: %.3e flop per call (FMA x2 + MUL + ADD, x64 lanes) = **%.1f TFLOP/s = %.1f %% of the 78.6 TFLOP/s FP64 vector "
          "peak**; FP64 share of VALU instructions %.1f %%."
          % (fl, fl / (wall_ms * 1e-3) / 1e12, 100 * fl / (wall_ms * 1e-3) / 78.6e12,
             100 * (allc["SQ_INSTS_VALU_FMA_F64"] + allc.get("SQ_INSTS_VALU_MUL_F64", 0) + allc.get("SQ_INSTS_VALU_ADD_F64", 0)
                    + allc.get("SQ_INSTS_VALU_TRANS_F64", 0)) / allc["SQ_INSTS_VALU"]))
        summary["executed_fp64_flop_per_call"] = fl
    if "SQ_WAVE_CYCLES" in allc and "SQ_ACTIVE_INST_VALU" in allc:
        s = "\nDerived: VALU-active share of wave lifetime = %.3f" % (allc["SQ_ACTIVE_INST_VALU"] / allc["SQ_WAVE_CYCLES"])
        if "SQ_WAIT_ANY" in allc:
            s += "; wave parked (s_waitcnt) share = %.3f" % (allc["SQ_WAIT_ANY"] / allc["SQ_WAVE_CYCLES"])
        if "SQ_WAIT_INST_ANY" in allc:
            s += "; issue-stall share = %.3f" % (allc["SQ_WAIT_INST_ANY"] / allc["SQ_WAVE_CYCLES"])
        w(s + ".")
        if "GRBM_GUI_ACTIVE" in allc:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs — both are
            # collected with the dispatches SERIALISED: their ratio is the busy share of the kernels run one after the other
            gui = allc["GRBM_GUI_ACTIVE"] / 8
            busy_serial = allc["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * gui)
            pmc_wall = pmc_q["median"] if pmc_q else None
            clock = gui / (pmc_wall * 1e-3) / 1e9 if pmc_wall else None
            w("\nVector pipes busy, kernels one after the other (as the counters see them): SQ_ACTIVE_INST_VALU x 4 = %.3e SIMD-cycles of "
              "%.3e (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = **%.1f %%**%s."
              % (allc["SQ_ACTIVE_INST_VALU"] * 4.0, 1024 * gui, 100 * busy_serial,
                 (" at an effective clock of %.2f GHz (GRBM_GUI_ACTIVE / 8 over the call's %.3f ms under `--pmc`)" % (clock, pmc_wall)) if clock else ""))
            busy_wall = allc["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * wall_ms * 1e-3 * PEAK_CLOCK_GHZ * 1e9)
            w("\nVector pipes busy over the call as it runs in production (kernels overlapping): the same SIMD-cycles over 1024 SIMDs x "
              "%.1f GHz x %.3f ms = **%.1f %%** — a lower bound, the chip rarely holds %.1f GHz under FP64 load."
              % (PEAK_CLOCK_GHZ, wall_ms, 100 * busy_wall, PEAK_CLOCK_GHZ))
            summary.update(vector_pipes_busy_serialised=busy_serial, vector_pipes_busy_over_wall_at_peak_clock=busy_wall, effective_clock_ghz_under_pmc=clock)
            tl = os.path.join(dst, "traffic_latest.json")
            if not headline and os.path.exists(tl):
                ref_clock = json.load(open(tl)).get("effective_clock_ghz")
                if ref_clock:  # the one clock that IS measured: GRBM_GUI_ACTIVE over the headline's single long kernel (no gaps to dilute it)
                    w("At the %.2f GHz the chip holds under this arithmetic (measured on the headline kernel, `profiles/%s`): **%.1f %%**."
                      % (ref_clock, json.load(open(tl)).get("tag"), 100 * busy_wall * PEAK_CLOCK_GHZ / ref_clock))
                    summary["vector_pipes_busy_over_wall_at_headline_clock"] = busy_wall * PEAK_CLOCK_GHZ / ref_clock
    json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
    open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
    if headline and traffic_hi is not None:
        dk = per_kernel.get(dominant, {})
        json.dump({"kernel": kernel_sub, "batch": batch, "tag": tag, "hbm_bytes_per_launch": traffic_hi,
                   "valu_wave_insts_per_launch": allc.get("SQ_INSTS_VALU"), "executed_fp64_flop_per_launch": summary.get("executed_fp64_flop_per_call"),
                   "vector_pipes_busy_frac": summary.get("vector_pipes_busy_serialised"), "effective_clock_ghz": summary.get("effective_clock_ghz_under_pmc"),
                   "dominant_kernel": dominant, "kernel_median_ms": dk.get("median"), "kernel_q1_ms": dk.get("q1"), "kernel_q3_ms": dk.get("q3"),
                   "kernel_profiled_dispatches": dk.get("n"), "call_median_ms": call_q["median"] if call_q else None,
                   "note": "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, gfx950 FETCH_SIZE x2 correction (upper bound for this kernel's "
                           "8-B-per-lane loads); kernel_median_ms: the dominant kernel's median over the profiled dispatches of the kernel-trace pass"},
                  open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
