#!/usr/bin/env bash
# Cross-check of this prover against snarkjs on a REAL zkey -- for a maintainer on a box that has snarkjs
# (the build container has neither snarkjs nor a real .zkey/.wtns: SURVEY.md 8c; this script was never run there).
#
#   tools/crosscheck_snarkjs.sh <circuit_final.zkey> <witness.wtns> [workdir]
#
# What it does, mirroring the reference's own prove -> verify sequence
# (scripts/g16_prove.sh:246-252, scripts/g16_verify.sh:213-216, scripts/g16_setup.sh:287-293):
#   1. `prover zkey wtns proof.json public.json`               (this repo's drop-in, self-check ON)
#   2. `snarkjs zkey export verificationkey zkey vkey.json`    (the reference's own export), byte-compared with
#      `zkpoa-verify --export-vkey` on the same zkey
#   3. `snarkjs groth16 verify vkey.json public.json proof.json`  -> must print "snarkJS: OK!"
#   4. the native verifier on the same three files            -> must agree
#   5. `snarkjs groth16 prove` on the same inputs; public.json must be byte-identical, and with
#      ZKPOA_JSON=snarkjs our public.json must equal snarkjs' byte for byte. (proof.json differs: r, s are random
#      on both sides; "bit-identical proofs" only exists for injected r, s -- BASELINE north_star)
# Exit code 0 = every step agreed.
set -euo pipefail

ZKEY=${1:?usage: crosscheck_snarkjs.sh <zkey> <wtns> [workdir]}
WTNS=${2:?usage: crosscheck_snarkjs.sh <zkey> <wtns> [workdir]}
WORK=${3:-$(mktemp -d)}
HERE=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
PROVER="$HERE/zk-proof-of-assets_amd/prover"
VERIFY="$HERE/zk-proof-of-assets_amd/zkpoa-verify"
SNARKJS=${SNARKJS:-npx snarkjs}

mkdir -p "$WORK"
echo "== 1. MI355X prover (self-check against the zkey's own vkey is on by default)"
ZKPOA_SELFCHECK=all ZKPOA_VERBOSE=1 "$PROVER" "$ZKEY" "$WTNS" "$WORK/proof.json" "$WORK/public.json"

echo "== 2. snarkjs zkey export verificationkey"
$SNARKJS zkey export verificationkey "$ZKEY" "$WORK/vkey.json"

echo "== 2b. our export of the same key must be byte-identical (zkey sections 2-3 decoded with the same conventions)"
"$VERIFY" --export-vkey "$ZKEY" "$WORK/vkey_ours.json"
cmp "$WORK/vkey.json" "$WORK/vkey_ours.json"

echo "== 3. snarkjs groth16 verify (the reference's acceptance check, g16_verify.sh:213-216)"
$SNARKJS groth16 verify "$WORK/vkey.json" "$WORK/public.json" "$WORK/proof.json" | tee "$WORK/verify.log"
grep -q "OK" "$WORK/verify.log"

echo "== 4. native verifier on the same files"
"$VERIFY" "$WORK/vkey.json" "$WORK/public.json" "$WORK/proof.json"

echo "== 5. snarkjs groth16 prove on the same inputs: public signals must match byte for byte"
$SNARKJS groth16 prove "$ZKEY" "$WTNS" "$WORK/proof_snarkjs.json" "$WORK/public_snarkjs.json"
ZKPOA_JSON=snarkjs "$PROVER" "$ZKEY" "$WTNS" "$WORK/proof_ours_snarkjs_style.json" "$WORK/public_ours_snarkjs_style.json"
cmp "$WORK/public_snarkjs.json" "$WORK/public_ours_snarkjs_style.json"
"$VERIFY" "$WORK/vkey.json" "$WORK/public_snarkjs.json" "$WORK/proof_snarkjs.json"

echo "crosscheck OK: snarkjs accepts our proof, the native verifier accepts snarkjs' proof, public signals identical"
echo "(files kept in $WORK)"
