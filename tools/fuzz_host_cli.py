"""Mutation fuzz driver for tools/fuzz_host_cli.sh (sanitizer builds of zkpoa-verify / zkpoa-sanitize in argv[1])."""
import os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, N, SEED = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
REF = os.path.join(ROOT, "tests", "golden", "ref")
case = os.path.join(REF, "1_sigs_1_batches_5_height__layer_one__batch_0")
vk = open(os.path.join(REF, "layer_one_vkey.json"), "rb").read()
pub = open(os.path.join(case, "public.json"), "rb").read()
proof = open(os.path.join(case, "proof.json"), "rb").read()
zkey = open(os.path.join(ROOT, "tests", "golden", "gen", "n8", "circuit.zkey"), "rb").read()
rng = random.Random(SEED)
INSERTS = [b'"', b'[', b']', b'{', b'}', b',', b':', b'-', b'1e9', b'\\u0000', b'9' * 80, b'[' * 50]


def mutate_text(s):
    b = bytearray(s)
    for _ in range(rng.choice([1, 1, 2, 5, 20])):
        if not b:
            break
        op, i = rng.randrange(6), rng.randrange(len(b))
        if op == 0: b[i] = rng.randrange(256)
        elif op == 1: del b[i:i + rng.randrange(1, 40)]
        elif op == 2: b[i:i] = bytes(rng.randrange(32, 127) for _ in range(rng.randrange(1, 30)))
        elif op == 3: b[i:i] = rng.choice(INSERTS)
        elif op == 4: b = b[:i]
        else:
            j = rng.randrange(len(b)); b[i], b[j] = b[j], b[i]
    return bytes(b)


def mutate_zkey(z):
    b = bytearray(z)
    for _ in range(rng.choice([1, 1, 2, 4])):
        op, i = rng.randrange(4), rng.randrange(min(len(b), 3000))
        if op == 0: b[i] = rng.randrange(256)
        elif op == 1: b[i:i + 4] = rng.choice([b"\xff\xff\xff\xff", b"\0\0\0\0", b"\xff\xff\xff\x7f", b"\x01\0\0\x80"])
        elif op == 2: b = b[:rng.randrange(len(b))]
        else: b[i:i + 8] = rng.randrange(1 << 64).to_bytes(8, "little")
    return bytes(b)


def run(argv):
    r = subprocess.run(argv, capture_output=True, timeout=120)
    if b"Sanitizer" in r.stderr or b"runtime error" in r.stderr or r.returncode < 0 or r.returncode > 100:
        print("FINDING", argv, r.returncode, r.stderr[-800:].decode(errors="replace"))
        return 1
    return 0


findings = 0
p = lambda n: os.path.join(W, n)
for it in range(N):
    trio = [vk, pub, proof]
    k = rng.randrange(3)
    trio[k] = mutate_text(trio[k])
    for n, c in zip(("vk.json", "pub.json", "proof.json"), trio):
        open(p(n), "wb").write(c)
    findings += run([p("zkpoa-verify"), p("vk.json"), p("pub.json"), p("proof.json")])
    findings += run([p("zkpoa-sanitize"), p("vk.json"), p("pub.json"), p("proof.json"), p("sanitized.json")])
    open(p("f.zkey"), "wb").write(mutate_zkey(zkey))
    findings += run([p("zkpoa-verify"), "--export-vkey", p("f.zkey"), p("vk_out.json")])
    if findings > 3:
        break
print("iterations %d, findings %d" % (N, findings))
sys.exit(1 if findings else 0)
