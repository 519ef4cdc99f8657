"""Mutated copies of a 2-D (the reference's square_nb.1) and a 3-D (tools/sanitize/drive.cpp writes it) TetGen file set for
tools/sanitize/fuzz_reader.cpp: truncations, flipped bytes, replaced tokens (negative, huge, NaN, fractions ...), dropped and
duplicated lines, valid-looking node indices in the wrong place (non-manifold / degenerate / duplicate topologies), headers
that lie about their counts.  python make_fuzz_files.py <repo root> <work dir> [seed] [count]"""
import os, random, shutil, sys
root, work = sys.argv[1], sys.argv[2]
random.seed(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
src = [(os.path.join(root, "tests/golden/mesh/square_nb.1"), 2, ("node", "ele", "edge")), (os.path.join(work, "box.1"), 3, ("node", "ele", "face", "edge"))]
out = os.path.join(work, "fz")
os.makedirs(out, exist_ok=True)
for f in os.listdir(out): os.remove(os.path.join(out, f))
N = int(sys.argv[4]) if len(sys.argv) > 4 else 200
BAD = ["-1", "0", "999999999", "99999999999999999999", "nan", "inf", "-inf", "1e400", "x", "", "3.5", "-0", "+7", "1e-320", "0x10", "2147483648", "9223372036854775807"]
with open(os.path.join(out, "list.txt"), "w") as lst:
    for k in range(N):
        prefix, dim, exts = random.choice(src)
        name = os.path.join(out, "m%04d.1" % k)
        victim = random.choice(exts)
        for e in exts:
            data = open(prefix + "." + e, "rb").read()
            if e == victim:
                kind = random.choice([0, 1, 2, 3, 4, 5, 6, 7, 7, 7, 7, 8, 8])
                if kind == 0:    # truncate
                    data = data[: random.randrange(len(data))]
                elif kind == 1:  # flip bytes
                    b = bytearray(data)
                    for _ in range(random.randrange(1, 20)):
                        b[random.randrange(len(b))] = random.randrange(256)
                    data = bytes(b)
                elif kind in (2, 3):  # replace tokens
                    toks = data.split()
                    for _ in range(random.randrange(1, 6)):
                        i = random.randrange(min(len(toks), 400) if kind == 2 else len(toks))
                        toks[i] = random.choice(BAD).encode()
                    data = b" ".join(toks)
                elif kind == 4:  # drop a line
                    lines = data.split(b"\n")
                    del lines[random.randrange(len(lines))]
                    data = b"\n".join(lines)
                elif kind == 5:  # duplicate a line
                    lines = data.split(b"\n")
                    i = random.randrange(len(lines))
                    lines.insert(i, lines[i])
                    data = b"\n".join(lines)
                elif kind in (7, 8):  # a valid-looking index somewhere else (non-manifold / degenerate / duplicate topologies)
                    lines = data.split(b"\n")
                    for _ in range(random.randrange(1, 4)):
                        i = random.randrange(1, max(2, len(lines) - 1))
                        t = lines[i].split()
                        if len(t) >= 3:
                            j = random.randrange(1, len(t))
                            t[j] = (t[random.randrange(1, len(t))] if kind == 8 else str(random.randrange(0, 50)).encode())
                            lines[i] = b" ".join(t)
                    data = b"\n".join(lines)
                else:            # header lies about the count
                    lines = data.split(b"\n")
                    h = lines[0].split()
                    if h: h[0] = random.choice([b"0", b"1", b"-5", str(int(h[0]) * 2).encode() if h[0].isdigit() else b"7", b"4611686018427387904"])
                    lines[0] = b" ".join(h)
                    data = b"\n".join(lines)
            open(name + "." + e, "wb").write(data)
        lst.write("%s. %d\n" % (name, random.choice([dim, dim, dim, 0, 0, 0, 0, 5 - dim])))
