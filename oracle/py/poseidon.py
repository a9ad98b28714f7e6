"""ORACLE (test infrastructure) -- big-int restatement of the Poseidon hash and the Merkle tree the reference builds
over its anonymity set (scripts/merkle_tree.rs:206-269 leaves, :138-178 node hash, :411 rs_merkle tree).

The reference calls light-poseidon 0.2.0 `Poseidon::<Fr>::new_circom(2)` (Cargo.toml:11), i.e. circomlib's Poseidon over
the BN254 scalar field: width t = 3, x^5 S-box, 8 full + 57 partial rounds, round constants and MDS matrix from the
Poseidon reference generator (Grain LFSR, `generate_parameters_grain.sage 1 0 254 3 8 57 <r>`: constants by rejection
sampling, Cauchy matrix 1 / (x_i + y_j)). Neither crate is vendored under /root/reference, so the parameters are
REGENERATED here from that published algorithm and pinned three ways: circomlib's first round constant, circomlib's
test vector poseidon([1, 2]), and the reference's own Merkle root for its committed anonymity set
(tests/1_sigs_1_batches_5_height/logs/merkle_tree.log:13)."""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
RF, RP = 8, 57

_cache = {}


def params(t=3):
    """(round constants [(RF + RP) * t], MDS matrix [t][t]) for width t, standard form."""
    if t in _cache:
        return _cache[t]
    n = 254
    bits = []

    def put(v, w):
        bits.extend(int(b) for b in bin(v)[2:].zfill(w))
    put(1, 2); put(0, 4); put(n, 12); put(t, 12); put(RF, 10); put(RP, 10)      # prime field, x^alpha S-box
    bits.extend([1] * 30)
    state = bits

    def step():
        nb = state[62] ^ state[51] ^ state[38] ^ state[23] ^ state[13] ^ state[0]
        state.pop(0)
        state.append(nb)
        return nb
    for _ in range(160):
        step()

    def rbits(k):
        v = 0
        for _ in range(k):
            nb = step()
            while nb == 0:            # self-shrinking: a 0 discards the following bit
                step()
                nb = step()
            v = (v << 1) | step()
        return v
    C = []
    while len(C) < (RF + RP) * t:
        v = rbits(n)
        if v < R:
            C.append(v)
    while True:
        rl = [rbits(n) % R for _ in range(2 * t)]
        if len(set(rl)) != 2 * t:
            continue
        xs, ys = rl[:t], rl[t:]
        if any((xs[i] + ys[j]) % R == 0 for i in range(t) for j in range(t)):
            continue
        M = [[pow((xs[i] + ys[j]) % R, -1, R) for j in range(t)] for i in range(t)]
        break
    _cache[t] = (C, M)
    return _cache[t]


def poseidon(inputs):
    """circomlib Poseidon(len(inputs)) -> field element (state[0] after the permutation; capacity element 0 first)."""
    t = len(inputs) + 1
    C, M = params(t)
    st = [0] + [int(x) % R for x in inputs]
    for r in range(RF + RP):
        st = [(st[i] + C[r * t + i]) % R for i in range(t)]
        if r < RF // 2 or r >= RF // 2 + RP:
            st = [pow(x, 5, R) for x in st]
        else:
            st[0] = pow(st[0], 5, R)
        st = [sum(M[i][j] * st[j] for j in range(t)) % R for i in range(t)]
    return st[0]


def merkle_levels(addresses, balances):
    """merkle_tree.rs: leaf = poseidon(address, balance); zero-valued (unhashed) leaves pad to a power of two
    (:261-266); node = poseidon(left, right). Returns the levels, leaves first, root last."""
    level = [poseidon([a, b]) for a, b in zip(addresses, balances)]
    n = 1
    while n < len(level):
        n *= 2
    level += [0] * (n - len(level))
    levels = [level]
    while len(level) > 1:
        level = [poseidon([level[i], level[i + 1]]) for i in range(0, len(level), 2)]
        levels.append(level)
    return levels


def merkle_path(levels, index):
    """(path_elements, path_indices) as merkle_tree.rs writes them (:280-288, :366-376): sibling per level from the
    leaves up, index bit per level."""
    elems, idx, i = [], [], index
    for lv in levels[:-1]:
        elems.append(lv[i ^ 1])
        idx.append(i & 1)
        i >>= 1
    return elems, idx
