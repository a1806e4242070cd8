#!/bin/bash
# HPRIOffline<GPU>::process(std::vector<float>) (tools/offline_host.cpp, one hour of audio): the round-6 path (result vectors
# appended from the sink while the separation runs) against the path of rounds 4-5 (ZEN_PROCESS_PLAIN=1: vectors first), alternating.
cd "$(dirname "$0")/.." || exit 1
g++ -O2 -std=c++17 -I include -I zen_amd/libzen tools/offline_host.cpp -o /tmp/offline_host -L zen_amd -lzen -lzen_hip -Wl,-rpath,$PWD/zen_amd || exit 1
for r in 1 2; do
  echo "sink:";  ZEN_TRACE_PROCESS=${TRACE:-} /tmp/offline_host ${1:-3600} 3 2>&1 | tail -${TAILN:-1}
  echo "plain:"; ZEN_PROCESS_PLAIN=1 /tmp/offline_host ${1:-3600} 3 2>&1 | tail -1
done
