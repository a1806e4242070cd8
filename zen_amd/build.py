"""Builds zen_amd/libzen_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract: every float
product and sum is rounded separately, as in oracle/zen_oracle.c.
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# A/B builds (another set of flags beside the shipped library): ZEN_HIP_EXTRA_FLAGS="-DZEN_FFT16K_V=32" ZEN_HIP_VARIANT=v32
# python zen_amd/build.py -> zen_amd/libzen_hip_v32.so (objects in build_v32/); run with ZEN_HIP_SO=zen_amd/libzen_hip_v32.so
_VARIANT = os.environ.get("ZEN_HIP_VARIANT", "")
OUT = os.path.join(HERE, "libzen_hip%s.so" % ("_" + _VARIANT if _VARIANT else ""))
OBJDIR = os.path.join(HERE, "build" + ("_" + _VARIANT if _VARIANT else ""))
RESOURCES = os.path.join(HERE, "kernel_resources%s.json" % ("_" + _VARIANT if _VARIANT else ""))   # per-kernel registers / scratch / LDS of the last build
SOURCES = ["api.hip", "hpr.hip", "hpri.hip", "stft.hip", "istft.hip", "median.hip", "median_net.hip", "median47.hip", "median_big.hip", "rt_fused.hip", "rt_fused_multi.hip", "rt_fused_multi_lean.hip", "rt_sse.hip", "rt_sse_lat.hip", "rt_hop_lat.hip", "rt_wide.hip", "box.hip", "fft_big.hip", "sse_block.hip", "rt_resident.hip", "memguard.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXTRA = os.environ.get("ZEN_HIP_EXTRA_FLAGS", "").split()
# A/B builds that differ in a few files only: ZEN_HIP_VARIANT_FILES="stft.hip,istft.hip" compiles just those with the extra
# flags (into build_<variant>/) and links them with the shipped build's other objects (zen_amd/build/, brought up to date first)
_VARIANT_FILES = [f for f in os.environ.get("ZEN_HIP_VARIANT_FILES", "").split(",") if f]
BASE_OBJDIR = os.path.join(HERE, "build")
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


# per-file additions.  rt_fused.hip: the max-ILP scheduling strategy hides more latency at the kernel's fixed
# 3 waves per SIMD (0.64 -> 0.62 ms per 25 840 hops, same registers, no spills); istft.hip: -4 %; measured
# no gain or a loss on stft and median_net.  median_big.hip: the minimum-register iterative scheduler keeps
# the 128-wide merge networks inside 256 VGPRs (187 taps: 56 spilled registers -> 4; 0.94 -> 0.73 ms).
# -amdgpu-use-amdgpu-trackers=1 (the scheduler tracks register pressure with the AMDGPU-specific trackers): the
# block build of the fused kernel spills 5 registers instead of 19 at its 168-VGPR limit (0.594 -> 0.566 ms per
# 25 840 hops), the half-row build of the 47-tap kernel needs 76 instead of 92 VGPRs (5 -> 6 workgroups per CU,
# 0.140 -> 0.115 ms); it makes istft.hip worse (nfft 16384: 3 -> 22 spilled registers, +20 %) and leaves the others
# where they are (A/B of every file on the three bench workloads, round 2).
FILE_FLAGS = {"rt_fused.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-use-amdgpu-trackers=1"],
              "rt_fused_multi.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              # -DZEN_FFT_FOLD_ADDR (fft_dev.h Plan::pad_off: the padding arithmetic of the LDS image once per base address, the rest
              # immediates): measured per translation unit in round 5 -- the three-output lean fused kernel gains 2.4 % (0.8205 ->
              # 0.8005 ms per 25 840 hops, 160 -> 149 VGPRs); the one-output headline build loses 1 % with it (its own image
              # arithmetic is folded by hand in rt_fused.hip), the synthesis kernels of istft.hip are neutral to -1 % at nfft 16384
              # (+3 % at nfft 1024) although they drop 10-20 registers and every spill: they do not wait for VALU issue
              "rt_fused_multi_lean.hip": ["-DZEN_FFT_FOLD_ADDR"],
              "median47.hip": ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"],
              "istft.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              # (round 3: with the loads of a frame in flight together the analysis kernels gain from it too: nfft 16384
              # 1.08 -> 0.99 ms, nfft 1024 0.74 -> 0.64 ms per offline batch step; it lost while they were serialised)
              "stft.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              "sse_block.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              "median_big.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
              "rt_wide.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
              # the latency-layout single-hop kernels (round 5; tools/ab_lat_flags.sh, same box, shipped build before and after):
              # resident hop 256 / 512 / 1024 6.3-6.5 / 8.4-8.7 / 11.1-11.6 -> 5.6 / 7.8 / 11.0 us, SSE hop 512 11.1-11.4 -> 10.6;
              # per launch 0.1-0.4 us.  max-ilp alone: about half of that.
              "rt_hop_lat.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-use-amdgpu-trackers=1"],
              "rt_sse_lat.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-use-amdgpu-trackers=1"]}
FILE_FLAGS_ENV = os.environ.get("ZEN_HIP_FILE_FLAGS", "")   # A/B hook: "median_net.hip=-mllvm,-amdgpu-sched-strategy=max-ilp"
for _item in filter(None, FILE_FLAGS_ENV.split(";")):
    _name, _, _fl = _item.partition("=")
    FILE_FLAGS[_name] = [f for f in _fl.split(",") if f]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "zen_hip.h"))
    hdrs.append(os.path.abspath(__file__))   # the flags live here
    return hdrs


def _compile(src):
    if _VARIANT and _VARIANT_FILES and src not in _VARIANT_FILES:
        return _compile_in(src, BASE_OBJDIR, FLAGS[len(EXTRA):])
    return _compile_in(src, OBJDIR, FLAGS)


def _compile_in(src, objdir, flags):
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    extra = [os.path.join(CSRC, "rt_fused.hip")] if src.startswith(("rt_fused_multi", "rt_resident")) else []   # they include that file
    newest = max(os.path.getmtime(p) for p in [srcp] + extra + _deps())
    if os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [HIPCC] + flags + FILE_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, universal_newlines=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise subprocess.CalledProcessError(r.returncode, cmd)
    with open(obj + ".usage.json", "w") as f:
        json.dump(_parse_usage(r.stderr), f, indent=0, sort_keys=True)
    return obj, True


def _parse_usage(remarks):
    """The compiler's per-kernel resource remarks -> {demangled kernel: {vgprs, agprs, sgprs, scratch, occupancy, lds}}.
    Several kernels sit at a register limit where a small edit changes what is spilled (rt_fused.hip's block
    builds: 19 spilled registers as measured, 69 after an unrelated cleanup): tests/test_build_resources.py reads
    these numbers, so such a change shows up in the CPU tier and not as a slower bench line."""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds"}
    for ln in remarks.splitlines():
        if "remark:" not in ln:
            continue
        body = ln.split("remark:", 1)[1].split("[-Rpass", 1)[0].strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            out[cur] = {}
        elif cur and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in keys:
                out[cur][keys[k.strip()]] = int(v)
    names = list(out)
    if names:
        try:
            dem = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, universal_newlines=True,
                                 check=True).stdout.splitlines()
            out = {d.replace("zen_hip_impl::(anonymous namespace)::", "").replace("zen_hip_impl::", ""): out[n]
                   for n, d in zip(names, dem)}
        except (OSError, subprocess.CalledProcessError):
            pass
    return out


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(BASE_OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [r[0] for r in res]
    if any(r[1] for r in res) or not os.path.exists(OUT):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
        if verbose:
            print("built", OUT)
    usage = {}
    for o in objs:
        if os.path.exists(o + ".usage.json"):
            with open(o + ".usage.json") as f:
                usage[os.path.basename(o).replace(".o", ".hip")] = json.load(f)
    with open(RESOURCES, "w") as f:
        json.dump(usage, f, indent=1, sort_keys=True)
    if not _VARIANT:
        _stamp_revision()
    return OUT


KERNEL_PATHS = ("--", "zen_amd/csrc", "include", "zen_amd/build.py")   # git pathspec of the library's sources


def _stamp_revision():
    """.build_rev (git-ignored, travels to the GPU box with the built library): the commit the library was built from, with
    a mark when the kernel sources differ from it.  tools/collect_profiles.sh labels every record it writes with this; the
    box has no .git, so the label is taken here, at build time, never written by hand."""
    root = os.path.dirname(HERE)
    try:
        rev = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             universal_newlines=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--", "zen_amd/csrc", "include", "zen_amd/build.py"], cwd=root,
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, check=True).stdout.split("\n")
        n = len([d for d in dirty if d.strip()])
        with open(os.path.join(root, ".build_rev"), "w") as f:
            f.write(rev + ("+%d-uncommitted-source-files" % n if n else "") + "\n")
        # the last commit that touched what the library is made of: what a fuzz summary must name (tools/fuzz_final.sh refuses
        # to write one for a build with uncommitted kernel sources; tests/test_profiles.py compares the name with the tree)
        krev = subprocess.run(["git", "log", "-1", "--format=%h"] + list(KERNEL_PATHS), cwd=root, stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, universal_newlines=True, check=True).stdout.strip()
        with open(os.path.join(root, ".build_kernel_rev"), "w") as f:
            f.write(krev + ("+%d-uncommitted-source-files" % n if n else "") + "\n")
    except (OSError, subprocess.CalledProcessError):
        pass          # no git here (the GPU box): the stamp written where the library was built stays


def build_host(verbose=False):
    """C++ host mirror of the reference's libzen (zen_amd/libzen -> zen_amd/libzen.so), the `zen` command
    line tool (zen_amd/bin/zen) and the C++ test program (tests/cpp/test_libzen).  Plain g++: the host
    layer reaches the GPU only through the C-ABI of libzen_hip.so."""
    root = os.path.dirname(HERE)
    inc = ["-I", os.path.join(root, "include"), "-I", os.path.join(HERE, "libzen")]
    cxx = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-fPIC", "-Wall", "-ffp-contract=off"]
    libzen = os.path.join(HERE, "libzen.so")
    src = os.path.join(HERE, "libzen", "hps.cpp")
    link_hip = ["-L", HERE, "-lzen_hip", "-Wl,-rpath,$ORIGIN"]

    def stale(out, deps):
        return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)

    hdrs = []
    for d, _, fs in os.walk(os.path.join(HERE, "libzen")):
        hdrs += [os.path.join(d, f) for f in fs]
    hdrs.append(os.path.join(root, "include", "zen_hip.h"))
    if stale(libzen, hdrs + [OUT]):
        subprocess.check_call(cxx + inc + ["-shared", "-pthread", src, "-o", libzen] + link_hip)
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    cli = os.path.join(HERE, "bin", "zen")
    cli_src = [os.path.join(HERE, "cli", "main.cpp"), os.path.join(HERE, "cli", "wav.h")]
    if stale(cli, cli_src + [libzen]):
        subprocess.check_call(cxx + inc + [cli_src[0], "-o", cli, "-L", HERE, "-lzen", "-lzen_hip",
                                           "-Wl,-rpath,$ORIGIN/.."])
    tdir = os.path.join(root, "tests", "cpp")
    texe = os.path.join(tdir, "test_libzen")
    oracle_dir = os.path.join(root, "oracle")
    oracle_so = os.path.join(oracle_dir, "libzen_oracle.so")
    if os.path.exists(oracle_so) and stale(texe, [os.path.join(tdir, "test_libzen.cpp"), libzen, oracle_so]):
        subprocess.check_call(cxx + inc + [os.path.join(tdir, "test_libzen.cpp"), "-o", texe, "-L", HERE, "-lzen",
                                           "-lzen_hip", "-L", oracle_dir, "-lzen_oracle",
                                           "-Wl,-rpath,$ORIGIN/../../zen_amd", "-Wl,-rpath,$ORIGIN/../../oracle"])
    if verbose:
        print("built", libzen, cli, texe)
    return libzen, cli, texe


def build_tools(verbose=False):
    """The small gfx950 programs the GPU tier and bench.py run beside the library (tools/bin/, git-ignored, travels with the tree):
    the exhaustive proof of exact_div.h, the tuned copy kernel that is bench.py's practical HBM denominator, the store-pattern
    micro-benchmark."""
    root = os.path.dirname(HERE)
    bindir = os.path.join(root, "tools", "bin")
    os.makedirs(bindir, exist_ok=True)
    jobs = {"check_div": ["-ffp-contract=off", "-fno-fast-math"], "ubench_copy": [], "ubench_rowwrite": []}
    for name, extra in jobs.items():
        src, exe = os.path.join(root, "tools", name + ".hip"), os.path.join(bindir, name)
        deps = [src, os.path.join(CSRC, "exact_div.h")] if name == "check_div" else [src]
        if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
            continue
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-Wno-unused-result"] + extra + [src, "-o", exe])
        if verbose:
            print("built", exe)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_host(verbose=True)
    build_tools(verbose=True)
