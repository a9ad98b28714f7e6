#!/usr/bin/env bash
# AddressSanitizer + UBSan build of the host-only CLIs (zkpoa-verify incl. --export-vkey, zkpoa-sanitize) and a
# mutation fuzz of their inputs: the reference's committed proof / public / vkey JSON and a golden .zkey, with bytes
# flipped, ranges cut, brackets / quotes / long digit runs inserted, section sizes overwritten. A finding is any
# sanitizer report, signal or exit code outside the documented ones. CPU only (GPU ASan is not available on the pool).
#   tools/fuzz_host_cli.sh [iterations] [seed]
set -euo pipefail
HERE=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
N=${1:-500}; SEED=${2:-7}
W=$(mktemp -d)
CL=${CLANGXX:-/opt/rocm/lib/llvm/bin/clang++}
for f in verify verify_main sanitize_main; do
  hipcc -O1 -g -std=c++17 --offload-host-only -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unused-result \
        -c "$HERE/zk-proof-of-assets_amd/csrc/$f.hip" -o "$W/$f.o"
done
$CL -fsanitize=address,undefined -o "$W/zkpoa-verify" "$W/verify_main.o" "$W/verify.o"
$CL -fsanitize=address,undefined -o "$W/zkpoa-sanitize" "$W/sanitize_main.o" "$W/verify.o"
python3 "$HERE/tools/fuzz_host_cli.py" "$W" "$N" "$SEED"
