# Tuning aid (GPU box, through gpurun): the guided filter alone -- timing, the parity check, per-kernel times and the HBM bytes per pixel
# from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE; the program itself after `--`, --kernel-trace only beside --pmc).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 100 python tests/gf_time.py 4 1024
for v in 64,64 32,32; do echo "GF_VS=$v"; GF_VS=$v timeout -k 10 100 python tests/gf_time.py 4 1024; GF_VS=$v timeout -k 10 100 python tests/gf_time.py 8 512; done
timeout -k 10 200 python tests/gpu_diag.py guided 2>&1 | tail -8
rm -rf /tmp/gfprof; rocprofv3 --kernel-trace --stats -d /tmp/gfprof -o gf -- python3 tests/gf_time.py 4 1024 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/gfpmc_$c; rocprofv3 --kernel-trace --pmc $c -d /tmp/gfpmc_$c -o pmc -- python3 tests/gf_time.py 4 1024 > /dev/null 2>&1; done
python3 - <<'PY'
import sqlite3, glob, re, json
def kernel_times():
    c = sqlite3.connect(glob.glob("/tmp/gfprof/*.db")[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = "select s.kernel_name, count(*), avg(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kt, ks)
    return {re.sub(r"\(.*", "", n).replace("void ", ""): (cnt, av / 1e3) for n, cnt, av in c.execute(q) if "gf_" in n}
def counter(cname):
    c = sqlite3.connect(glob.glob("/tmp/gfpmc_%s/*.db" % cname)[0])
    rows = c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (cname,))
    return {re.sub(r"\(.*", "", r[0]).replace("void ", ""): (r[1], r[2]) for r in rows if "gf_" in r[0]}
fetch, write = counter("FETCH_SIZE"), counter("WRITE_SIZE")
try:
    times = kernel_times()
except Exception:
    times = {}
px = 4 * 1024 * 1024
raw = dbl = 0.0
out = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 tests/gf_time.py 4 1024 (one run per counter)",
       "note": "FETCH_SIZE / WRITE_SIZE in KB.  MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of WIDE (16 B / lane) coalesced reads on gfx950 "
               "and other access widths are uncalibrated -- calibrate on a known byte count in your own access pattern.  These kernels read 1 B / lane "
               "(uint8 rows) and 2 B / lane (half planes), a wave-load = 192 / 128 contiguous bytes.  Calibration: the kernels REQUEST "
               "6 x 1.3125 x (61 + 2 x 31) / 32 = 30.3 B / pixel (first launch) and 24 x 1.3125 x (61 + 2 x 63) / 64 = 92 B / pixel (second) in whole "
               "cache lines; a fetch count cannot exceed the bytes requested, and the DOUBLED counter does (35.9 and 118) while the raw counter "
               "(18.0 and 59.0) sits at the bytes a block touches once (22.6 and 61.0 B / pixel: rows + halo, columns + halo).  So for these narrow "
               "loads the raw counter is the byte count; both figures are recorded.", "kernels": {}}
for k in fetch:
    n, f = fetch[k]
    w = write.get(k, (n, 0.0))[1]
    fb, wb = f * 1024.0 / n, w * 1024.0 / n
    raw += fb + wb
    dbl += 2 * fb + wb
    out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch_raw": fb, "write_bytes_per_launch": wb,
                         "bytes_per_pixel_raw": (fb + wb) / px, "bytes_per_pixel_fetch_doubled": (2 * fb + wb) / px}
    print("%-40s n %3d  fetch %7.1f MB (raw)  write %7.1f MB per launch = %6.1f B / pixel (%.1f with the fetch doubled)" % (
        k[:40], n, fb / 1e6, wb / 1e6, (fb + wb) / px, (2 * fb + wb) / px))
out["bytes_per_pixel"] = raw / px
out["bytes_per_pixel_fetch_doubled"] = dbl / px
print("guided filter, batch of four 1024^2 images: %.1f B / pixel of HBM traffic by the calibrated (raw) counter, %.1f with FETCH_SIZE doubled" % (raw / px, dbl / px))
json.dump(out, open("gpurun_out/r5_gf_pmc.json", "w"), indent=1)
PY
