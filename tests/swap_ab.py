"""Tuning aid (GPU): same-process A/B of the swap step with an ops switch off / on.   python tests/swap_ab.py UP9 [steps]"""
import contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppst_amd import ops
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "UP9"
steps = sys.argv[2] if len(sys.argv) > 2 else "10"
for on in (False, True, False, True):
    getattr(ops, name)["value"] = on
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", steps, "--warmup", "2"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print("%s %-5s %.2f swaps/s  %.3f ms  conv %.4f  StyledConv %.4f" % (name, "on" if on else "off", d["value"], d["ms_per_step"],
                                                                         d["roofline"]["frac"], d["roofline_modulated_conv2d"]["frac"]), flush=True)
