"""Build ppst_amd/libppst_hip.so (gfx950) in-tree with hipcc.

    python -m ppst_amd.build          # rebuild if any source is newer than the .so

The .so is git-ignored but travels to the GPU box with the working tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libppst_hip.so")
# PPST_EXPERIMENTS=1: also compile the measured-and-off conv forms (conv_ksplit.hip, conv_mfma2.hip variants 1 / 3 / 7 / 9, the
# two-pass fp16 mode, the 8-row two-block tile): tuning builds only -- the production library carries what runs.
EXPERIMENTS = os.environ.get("PPST_EXPERIMENTS") == "1"
SOURCES = ["upfirdn2d.hip", "fused_bias_act.hip", "elementwise.hip", "linear.hip", "conv_mfma.hip", "conv_mfma2.hip", "conv1x1.hip", "conv_wino.hip", "conv_f32.hip",
           "corr.hip", "guided_filter.hip", "train.hip", "train_g.hip", "imageio.hip", "smooth_filter.hip"] + (["conv_ksplit.hip"] if EXPERIMENTS else [])
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"] + (["-DPPST_EXPERIMENTS"] if EXPERIMENTS else [])
# per-file flags.  conv_wino.hip: its staging arithmetic runs inside the MFMA stream and the kernel sits at 256 registers -- with SLP
# vectorisation hipcc packs it into v_pk_* (operand pairs assembled with moves, 1 100 packed instructions, spills in the
# normalise-on-load + activation build); without it: no spills, same speed
FILE_FLAGS = {"conv_wino.hip": ["-fno-slp-vectorize"]}
MODE_STAMP = os.path.join(HERE, ".libmode")       # flavour of the built library (git-ignored, travels with the .so)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "rscl_common.h"), os.path.join(HERE, "..", "include", "ppst_hip.h")]


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def _mode():
    return "experiments" if EXPERIMENTS else "production"


def _lib_flavour():
    """'experiments' / 'production' of the built library: the stamp file, or -- a library without one (built before the stamp
    existed, or copied alone) -- whether it carries the experiment kernels (conv_ksplit.hip is compiled only with PPST_EXPERIMENTS=1); None if unreadable."""
    try:
        return open(MODE_STAMP).read().strip()
    except OSError:
        pass
    if not os.path.exists(LIB):
        return None
    try:    # (no dlopen here: loading the library ahead of torch would bring the system's HIP runtime in first, ppst_amd/_lib.py)
        return "experiments" if b"conv_ksplit_kernel" in open(LIB, "rb").read() else "production"
    except OSError:
        return None


def _mode_changed():
    """True only when the library exists in the OTHER flavour (an up-to-date production library without a stamp is not
    rebuilt -- and is usable on a host without hipcc)."""
    have = _lib_flavour()
    if have is None:
        return os.path.exists(LIB)            # a library that cannot say what it is
    return have != _mode()


def needs_build():
    # (objects too: a source edited WHILE a build ran is older than the library that build linked, but newer than its own object)
    return (_mode_changed() or _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS) or
            any(_stale(os.path.join(OBJ, s.replace(".hip", ".o")), [os.path.join(CSRC, s)]) for s in SOURCES if os.path.isdir(OBJ)))


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    force = force or _mode_changed()              # objects of the other flavour are not reused
    hipcc = _hipcc()

    def cc(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        if not force and not _stale(obj, [os.path.join(CSRC, src)] + HEADERS):
            return obj                                   # this object is current: only what changed is recompiled
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        return obj

    # the three conv files dominate (minutes each): start them first, all files in parallel
    order = sorted(SOURCES, key=lambda f: 0 if f.startswith("conv") else 1)
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = dict(zip(order, ex.map(cc, order)))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [objs[s_] for s_ in SOURCES]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr)
    with open(MODE_STAMP, "w") as f:
        f.write(_mode() + "\n")
    if verbose:
        print("built", LIB, "(%s)" % _mode())
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
