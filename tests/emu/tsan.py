"""TEST INFRASTRUCTURE: build tests/emu/tsan_main.cpp with ThreadSanitizer and run it on a few fixture instances.
Usage: python tests/emu/tsan.py [fixture] [n]; prints the sanitizer's report (none = no LDS race between lanes)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
CLANG = os.environ.get("BMPC_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


def build(exe):
    subprocess.check_call([CLANG, "-std=c++20", "-O1", "-g", "-pthread", "-fsanitize=thread", "-D_GNU_SOURCE", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-I" + HERE, "-x", "c++", os.path.join(HERE, "tsan_main.cpp"), "-o", exe])


def dump(path, cp, x_fb, foot, contact, phase, x_cmd, mu):
    with open(path, "wb") as fd:
        fd.write(bytes(cp))
        fd.write(np.array([x_fb.shape[0], 0 if mu is None else 1], np.int32).tobytes())
        for a, t in ((x_fb, np.float32), (foot, np.float32), (contact, np.uint8), (phase, np.int32), (x_cmd, np.float32)):
            fd.write(np.ascontiguousarray(a, t).tobytes())
        if mu is not None:
            fd.write(np.ascontiguousarray(mu, np.float32).tobytes())


def races(stderr):
    """The sanitizer's reports as a list of access lists [(kind, 'file:line'), ...], one per report."""
    import re
    out = []
    for rep in stderr.split("WARNING: ThreadSanitizer: data race")[1:]:
        lines, acc = rep.split("\n"), []
        for i, ln in enumerate(lines):
            m = re.match(r"\s+(Write|Read|Previous write|Previous read|Atomic \w+|Previous atomic \w+) of size", ln)
            if m:                               # innermost frame that carries a line number (a lambda's may not)
                mm = None
                for fr in lines[i + 1:i + 4]:
                    mm = mm or re.search(r"(bmpc_kernels\.hip|bmpc_emu\.cpp):(\d+)", fr)
                acc.append((m.group(1), mm.group(0) if mm else "?"))
        out.append(acc)
    return out


def run(name="cfg2_standing_h10", n=1):
    """Returns (stdout, stderr, returncode) of the sanitized emulation of the first n instances of a fixture."""
    import biped_mpc_py_amd as bm
    from tests import util
    h, half = util.BATCH_FIXTURES[name]
    d = util.load(name)
    mpc = bm.MPC()
    mpc.h = h
    cp = bm.pack_params(mpc, bm.Biped(), half=half)
    mu = d["mu_steps"][:n] if "mu_steps" in d.files and d["mu_steps"].size else None
    out = os.environ.get("BMPC_TSAN_DIR", "/tmp")
    exe, inp = os.path.join(out, "bmpc_tsan"), os.path.join(out, "bmpc_tsan.in")
    build(exe)
    dump(inp, cp, d["x_fb"][:n], d["foot"][:n], d["contact"][:n], util.phases(d["t"][:n], mpc.dt, h), d["x_cmd"][:n], mu)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 exitcode=0")
    r = subprocess.run([exe, inp], env=env, capture_output=True, text=True)
    return r.stdout, r.stderr, r.returncode


if __name__ == "__main__":
    so, se, rc = run(sys.argv[1] if len(sys.argv) > 1 else "cfg2_standing_h10", int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(so)
    from collections import Counter
    for k, v in Counter(tuple(r) for r in races(se)).most_common():
        print(v, k)
    sys.exit(rc)
