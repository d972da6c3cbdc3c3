"""Summarise the three rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE; each
its own run, as MI355X_MICROARCH.md prescribes) into profiles/<round>_pmc_traffic.json.

    python tests/pmc_summary.py gpurun_out/pmc_r01c_ profiles/r01_pmc_traffic.json "<command line profiled>"

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide
coalesced reads).  Effective clock of a kernel = sum(GRBM_GUI_ACTIVE) / 8 XCDs / sum(duration)."""
import json, re, sqlite3, sys

prefix, out_path, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
commit = sys.argv[4] if len(sys.argv) > 4 else None


def per_kernel(counter):
    c = sqlite3.connect("%s%s/pmc_results.db" % (prefix, counter))
    rows = c.execute("select kernel_name, count(*), sum(value), sum(duration) from counters_collection where counter_name = ? group by kernel_name", (counter,))
    return {re.sub(r"\(.*", "", r[0]): (r[1], r[2], r[3]) for r in rows}


fetch, write, act = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE"), per_kernel("GRBM_GUI_ACTIVE")
kernels = {}
for k in fetch:
    n, f, _ = fetch[k]
    w = write.get(k, (0, 0.0, 0))[1]
    e = {"launches": n, "fetch_bytes_corrected": f * 1024.0 * 2.0, "write_bytes": w * 1024.0}
    if k in act and act[k][2] > 0:
        e["effective_clock_ghz"] = act[k][1] / 8.0 / act[k][2]
        e["kernel_ms_in_clock_pass"] = act[k][2] / 1e6
    kernels[k] = e
kernels = dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["fetch_bytes_corrected"] + kv[1]["write_bytes"])))
CONV = ("conv_mfma_kernel", "conv_mfma2_kernel", "conv1x1_stream_kernel", "conv3x3_direct_kernel", "conv_wino_kernel")   # every ppst_conv2d_mfma variant
is_conv = lambda k: any(c in k for c in CONV)
conv = [v for k, v in kernels.items() if is_conv(k)]
n = sum(v["launches"] for v in conv)
fb, wb = sum(v["fetch_bytes_corrected"] for v in conv), sum(v["write_bytes"] for v in conv)
cyc = sum(act[k][1] for k in act if is_conv(k))
dur = sum(act[k][2] for k in act if is_conv(k))
res = {
    "command": cmd,
    "commit": commit,
    "note": "one rocprofv3 --kernel-trace --pmc <counter> run per counter; FETCH_SIZE/WRITE_SIZE are KB; FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); effective clock = GRBM_GUI_ACTIVE / 8 / duration "
            "(reads high for dispatches shorter than ~0.3 ms); all steps of the run (warm-up included) are in the trace",
    "conv_mfma": {"launches": n, "hbm_bytes_per_launch": (fb + wb) / n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n,
                  "effective_clock_ghz": cyc / 8.0 / dur if dur else None},
    "kernels": kernels,
}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["conv_mfma"], indent=1))
for k, v in list(kernels.items())[:14]:
    print("%-50s n %4d  fetch %8.1f MB  write %8.1f MB  clk %s" % (k[:50], v["launches"], v["fetch_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6, "%.2f GHz" % v["effective_clock_ghz"] if "effective_clock_ghz" in v else "-"))
