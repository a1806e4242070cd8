"""The evidence under profiles/ must describe the kernels that are in the tree (the last six kernel commits of round 4
were never fuzzed).  CPU tier."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_PATHS = ["--", "zen_amd/csrc", "include", "zen_amd/build.py"]


def _git(*args):
    return subprocess.run(["git"] + list(args), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                          universal_newlines=True).stdout.strip()


def test_latest_fuzz_summary_names_the_kernel_sources_of_this_tree():
    """tools/fuzz_final.sh writes profiles/rNN_fuzz_summary.txt with a last line `kernel_commit <hash>`: the last commit that
    touched zen_amd/csrc, include/ or zen_amd/build.py when the fuzzed library was built (and it refuses a library built from
    uncommitted sources).  From round 5 on, that commit must be the tree's: a kernel change after the fuzz run means another
    fuzz run before the round closes."""
    if not os.path.isdir(os.path.join(ROOT, ".git")) or not _git("rev-parse", "HEAD"):
        pytest.skip("no git history here")
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_fuzz_summary.txt"))
                   if int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)) >= 5)
    if not files:
        pytest.skip("no fuzz summary of round 5 or later yet")
    txt = open(files[-1]).read()
    m = re.search(r"^kernel_commit ([0-9a-f]{7,40})$", txt, re.M)
    assert m, "%s carries no kernel_commit line (written by tools/fuzz_final.sh)" % files[-1]
    assert "mismatches 0" in txt and "MISMATCH" not in txt and "MEMCHECK" not in txt
    fuzzed = m.group(1)
    assert _git("cat-file", "-t", fuzzed) == "commit", "unknown commit " + fuzzed
    changed = _git("diff", "--name-only", fuzzed, "HEAD", *KERNEL_PATHS)
    assert changed == "", "kernel sources changed since the fuzzed build %s:\n%s" % (fuzzed, changed)
