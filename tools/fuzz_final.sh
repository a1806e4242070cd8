#!/bin/bash
# The round's closing fuzz run, on the GPU box:   gpurun -- 'tools/fuzz_final.sh r05'
# >= 3 realtime + 3 --offline + 1 --resident + 1 --filters seeds x 150 s of tools/fuzz_parity.py (HIP engine vs CPU oracle, tolerance 0, red zones on, every
# case followed by zen_hip_memcheck) on the build that is in the tree.  REFUSES to write a summary for a library that was built
# from uncommitted kernel sources (.build_kernel_rev, written by zen_amd/build.py where the library is built): the summary names
# the last commit that touched zen_amd/csrc / include / build.py, and tests/test_profiles.py fails the CPU tier when that is
# not the tree's -- a kernel commit after the fuzz run means another fuzz run.
TAG=${1:-x}; SECS=${2:-150}
cd "$(dirname "$0")/.." || exit 1
KREV=$(cat .build_kernel_rev 2>/dev/null || echo unknown)
case "$KREV" in
  unknown|*uncommitted*) echo "fuzz_final: the library was built from uncommitted kernel sources ($KREV): commit, rebuild, then fuzz" >&2; exit 2;;
esac
export ZEN_HIP_REDZONE=4096 ZEN_HIP_POISON=1
OUT=gpurun_out/${TAG}_fuzz_summary.txt
{
  echo "tools/fuzz_parity.py on the final build of round ${TAG#r}: kernel sources at commit $KREV (HEAD at build time: $(cat .build_rev 2>/dev/null))"
  echo "(random sample rate 2 ... 128 kHz, hop 32 ... 4096 -- every configuration the oracle's constructor accepts, masks of up to 267 taps and"
  echo " sliding matrices of up to 534 rows included: 'oracle-rejected' are the draws the reference itself cannot run, 'GPU-refused' must be 0 --,"
  echo " flags, causality, hard / soft / SSE, streams, blocking, chunking; --offline: the two-pass"
  echo " driver with its options drawn per case; HIP engine vs CPU oracle, tolerance 0; 4 KB red zones around every allocation and NaN-poisoned"
  echo " interiors, zen_hip_memcheck after every case)"
} > $OUT
RC=0
for seed in ${FUZZ_SEEDS:-501 502 503}; do   # (FUZZ_SEEDS / FUZZ_OFFLINE_SEEDS: other seeds, for an extra run under another tag)
  L=$(python3 tools/fuzz_parity.py --seconds $SECS --seed $seed 2>&1 | tail -3 | tr '\n' ' ')
  echo "seed $seed ($SECS s):            $L" >> $OUT
  case "$L" in *"mismatches 0"*"GPU-refused 0"*"memcheck clean"*) ;; *) RC=1;; esac
done
for seed in ${FUZZ_OFFLINE_SEEDS:-511 512 513}; do
  L=$(python3 tools/fuzz_parity.py --offline --seconds $SECS --seed $seed 2>&1 | tail -3 | tr '\n' ' ')
  echo "seed $seed (--offline, $SECS s): $L" >> $OUT
  case "$L" in *"mismatches 0"*"GPU-refused 0"*"memcheck clean"*) ;; *) RC=1;; esac
done
for seed in ${FUZZ_RESIDENT_SEEDS:-521}; do   # the per-hop API through the resident kernels (round 5)
  L=$(python3 tools/fuzz_parity.py --resident --seconds $SECS --seed $seed 2>&1 | tail -3 | tr '\n' ' ')
  echo "seed $seed (--resident, $SECS s): $L" >> $OUT
  case "$L" in *"mismatches 0"*"GPU-refused 0"*"memcheck clean"*) ;; *) RC=1;; esac
done
for seed in ${FUZZ_FILTER_SEEDS:-531}; do   # the drop-in filter classes, any shape and length (round 6)
  L=$(python3 tools/fuzz_parity.py --filters --seconds $SECS --seed $seed 2>&1 | tail -3 | tr '\n' ' ')
  echo "seed $seed (--filters, $SECS s): $L" >> $OUT
  case "$L" in *"mismatches 0"*"GPU-refused 0"*"memcheck clean"*) ;; *) RC=1;; esac
done
echo "kernel_commit $KREV" >> $OUT
cat $OUT
exit $RC
