"""Copies the Merkle fields (data only) of the reference's height-12 layer-two circuit inputs into tests/golden/ref/merkle/:
tests/4_sigs_2_batches_12_height/layer_two/batch_{0,1}/layer_two_batch_{0,1}_input.json -> leaf_addresses, leaf_balances,
merkle_root, path_elements, path_indices (two owned leaves per batch: four sibling paths of length 11 produced by the
reference's Rust binary, scripts/merkle_tree.rs:354-376). Run in the build container (the reference does not travel)."""
import json
import os

REF = "/root/reference/tests/4_sigs_2_batches_12_height/layer_two"
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref", "merkle")
KEYS = ("leaf_addresses", "leaf_balances", "merkle_root", "path_elements", "path_indices")
for b in (0, 1):
    d = json.load(open(os.path.join(REF, "batch_%d" % b, "layer_two_batch_%d_input.json" % b)))
    with open(os.path.join(HERE, "height12_layer_two_batch_%d_merkle_inputs.json" % b), "w") as f:
        json.dump({k: d[k] for k in KEYS}, f, indent=1)
        f.write("\n")
