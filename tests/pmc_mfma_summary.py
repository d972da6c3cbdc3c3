"""Summarise rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES (+ a GRBM_GUI_ACTIVE pass for the clock) of bench.py per conv kernel family:
matrix-pipe utilisation = MFMA busy cycles / (SIMDs x kernel cycles).  SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip's
1024 SIMDs, the cycles a SIMD's matrix pipe is busy (16 per v_mfma_f32_16x16x32_bf16: DESIGN.md section 4).
    python tests/pmc_mfma_summary.py gpurun_out/pmc_mfma_ profiles/r02_pmc_mfma.json <commit>"""
import json, re, sqlite3, sys

prefix, out_path = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else None


def per_kernel(counter):
    c = sqlite3.connect("%s%s/pmc_results.db" % (prefix, counter))
    rows = c.execute("select kernel_name, count(*), sum(value), sum(duration) from counters_collection where counter_name = ? group by kernel_name", (counter,))
    return {re.sub(r"\(.*", "", r[0]): (r[1], r[2], r[3]) for r in rows}


busy, act = per_kernel("SQ_VALU_MFMA_BUSY_CYCLES"), per_kernel("GRBM_GUI_ACTIVE")
CONV = ("conv_mfma_kernel", "conv_mfma2_kernel", "conv1x1_stream_kernel", "conv3x3_direct_kernel", "conv_wino_kernel")
out = {"commit": commit, "command": "rocprofv3 --kernel-trace --pmc <SQ_VALU_MFMA_BUSY_CYCLES|GRBM_GUI_ACTIVE> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
       "note": "utilisation = MFMA busy cycles / (1024 SIMDs x duration x clock of the same kernel from the GRBM_GUI_ACTIVE pass / 8 XCDs); "
               "bf16x3 issues three MFMAs per algorithmic MAC, so this is the fraction of the dense-bf16 matrix peak AT THE CLOCK THE KERNEL RAN AT",
       "kernels": {}}
tb = tc = 0.0
for k, (n, b, dur) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    if not any(c in k for c in CONV) or k not in act or act[k][2] <= 0:
        continue
    clk = act[k][1] / 8.0 / act[k][2]                  # GHz (cycles per ns)
    cycles = dur * clk                                 # kernel cycles in the busy pass
    util = b / (1024.0 * cycles)
    out["kernels"][k] = {"launches": n, "mfma_busy_frac": util, "clock_ghz": clk, "kernel_ms": dur / 1e6}
    tb += b; tc += 1024.0 * cycles
    print("%-64s n %4d  %8.2f ms  clk %.2f GHz  matrix pipe busy %.3f" % (k[:64], n, dur / 1e6, clk, util))
out["all_conv_launches"] = {"mfma_busy_frac": tb / tc if tc else None}
print("all conv launches: matrix pipe busy %.3f" % (tb / tc))
json.dump(out, open(out_path, "w"), indent=1)
