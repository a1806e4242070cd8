"""Configure-time probe for Intel IPP (TEST INFRASTRUCTURE: used by bench.py's cpu_baseline leg and tests/ only).

The reference's CPU path is Intel IPP (closed source; cmake/FindIPP.cmake:17,55-69 looks in $HOME/intel/ipp).  Neither
the build container nor this pool's GPU boxes have it, so the CPU baseline is the oracle ("port") and box / FFT / the
waveforms are "parity unpinned".  BASELINE.md section 3 promises: if a host does have ipp.h and libipp*, say so and
use it.  probe() looks in the usual places; check() then builds oracle/ipp_check.c against it (outputs in oracle/_ref/)
and returns its report: the literal IPP calls of the reference next to the restatement, differences and timings."""
import glob
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIBS = ("ippi", "ipps", "ippvm", "ippcore")          # cmake/FindIPP.cmake:55-69


def _roots():
    roots = [os.environ.get(k) for k in ("IPPROOT", "IPP_ROOT", "IPP_PATH")]
    roots += [os.path.expanduser("~/intel/ipp"), "/opt/intel/ipp", "/opt/intel/oneapi/ipp/latest", "/usr", "/usr/local"]
    roots += sorted(glob.glob("/opt/intel/oneapi/ipp/*")) + sorted(glob.glob("/opt/intel/*/ipp"))
    seen, out = set(), []
    for r in roots:
        if r and r not in seen:
            seen.add(r)
            out.append(r)
    return out


def probe():
    """{"found": bool, "searched": [...], "include": dir, "libdir": dir} -- no compilation, no execution."""
    searched = _roots()
    for r in searched:
        inc = next((d for d in (os.path.join(r, "include"), os.path.join(r, "include", "ipp")) if os.path.exists(os.path.join(d, "ipp.h"))), None)
        if not inc:
            continue
        for sub in ("lib/intel64", "lib", "lib64", "lib/x86_64-linux-gnu"):
            ld = os.path.join(r, sub)
            if all(glob.glob(os.path.join(ld, "lib%s.*" % n)) for n in LIBS):
                return {"found": True, "searched": searched, "include": inc, "libdir": ld}
    return {"found": False, "searched": searched}


def check(timeout=300):
    """probe(); where IPP exists, build and run oracle/ipp_check.c and attach its report."""
    p = probe()
    if not p["found"]:
        return p
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "ipp_check")
    cmd = ["gcc", "-O2", "-std=gnu99", "-I", p["include"], "-I", HERE, os.path.join(HERE, "ipp_check.c"), "-o", exe,
           "-L", HERE, "-lzen_oracle", "-L", p["libdir"]] + ["-l" + n for n in LIBS] + \
          ["-lm", "-Wl,-rpath," + HERE, "-Wl,-rpath," + p["libdir"]]
    try:
        subprocess.check_call(cmd, stderr=subprocess.PIPE)
        txt = subprocess.run([exe], stdout=subprocess.PIPE, timeout=timeout, check=True).stdout.decode()
        p["report"] = json.loads(txt)
    except Exception as exc:                                   # say so; the port remains the baseline
        p["error"] = str(exc)[:300]
    return p


if __name__ == "__main__":
    print(json.dumps(check(), indent=1))
