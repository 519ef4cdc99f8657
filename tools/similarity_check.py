#!/usr/bin/env python3
"""Share of a file's tokens that lie in runs of >= 6 consecutive tokens also found in any file of a reference
directory (comments stripped) -- the measure VERDICT r1 used on include/storm_hip/Storm.hpp against
/root/reference/source/Storm/Solvers/*.hpp.  Reading the reference as text; nothing is copied.

    python tools/similarity_check.py include/storm_hip/Storm.hpp /root/reference/source/Storm/Solvers"""
import glob
import os
import re
import sys

TOKEN = re.compile(r"[A-Za-z_][A-Za-z_0-9]*|\d+\.?\d*(?:[eE][-+]?\d+)?|<<=|[-+*/]=|::|->|&&|\|\||[^\s]")


def tokens(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"#[^\n]*", " ", text)
    return TOKEN.findall(text)


def main(path, ref_dir, run=6):
    mine = tokens(open(path, encoding="utf-8", errors="replace").read())
    grams = set()
    for f in glob.glob(os.path.join(ref_dir, "*")):
        if os.path.isfile(f):
            t = tokens(open(f, encoding="utf-8", errors="replace").read())
            grams.update(tuple(t[i:i + run]) for i in range(len(t) - run + 1))
    covered = [False] * len(mine)
    for i in range(len(mine) - run + 1):
        if tuple(mine[i:i + run]) in grams:
            for j in range(i, i + run):
                covered[j] = True
    share = sum(covered) / max(len(mine), 1)
    print(f"{path}: {len(mine)} tokens, {100 * share:.1f} % in runs of >= {run} tokens shared with {ref_dir}")
    return share


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
