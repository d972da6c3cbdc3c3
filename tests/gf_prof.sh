# Tuning aid (GPU box, through gpurun): the guided filter alone -- timing, the parity check, per-kernel times and the HBM bytes per pixel
# from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE; the program itself after `--`, --kernel-trace only beside --pmc).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 100 python tests/gf_time.py 4 1024
timeout -k 10 100 python tests/gf_time.py 8 512
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
    return {re.sub(r"\(.*", "", n): (cnt, av / 1e3) for n, cnt, av in c.execute(q) if "gf_" in n}
def counter(cname):
    c = sqlite3.connect(glob.glob("/tmp/gfpmc_%s/*.db" % cname)[0])
    rows = c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (cname,))
    return {re.sub(r"\(.*", "", r[0]): (r[1], r[2]) for r in rows if "gf_" in r[0]}
times, fetch, write = kernel_times(), counter("FETCH_SIZE"), counter("WRITE_SIZE")
px = 4 * 1024 * 1024
tot = 0.0
out = {"note": "FETCH_SIZE / WRITE_SIZE in KB, separate rocprofv3 --kernel-trace --pmc passes of tests/gf_time.py 4 1024; FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950)", "kernels": {}}
for k in fetch:
    n, f = fetch[k]
    w = write.get(k, (n, 0.0))[1]
    fb, wb = f * 1024.0 * 2.0 / n, w * 1024.0 / n
    tot += fb + wb
    out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "us": times.get(k, (0, 0.0))[1]}
    print("%-44s n %3d  fetch %7.1f MB  write %7.1f MB per launch = %6.1f B / pixel   %.1f us" % (k[:44], n, fb / 1e6, wb / 1e6, (fb + wb) / px, times.get(k, (0, 0.0))[1]))
out["bytes_per_pixel"] = tot / px
out["ms_per_batch"] = sum(v["us"] for v in out["kernels"].values()) / 1e3
print("guided filter, batch of four 1024^2 images: %.1f B / pixel of HBM traffic, %.3f ms of kernels" % (tot / px, out["ms_per_batch"]))
json.dump(out, open("gpurun_out/r5_gf_pmc.json", "w"), indent=1)
PY
