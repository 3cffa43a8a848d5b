#!/usr/bin/env python3
"""8-token shingle overlap of a repository file against reference files (the copy check the round-2 verdict describes).
Usage: python tools/shingle_check.py <repo file> <reference file> [...]   (runs only where /root/reference exists)"""
import re
import sys


def tokens(path):
    text = open(path, errors="replace").read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S); text = re.sub(r"//[^\n]*", " ", text)
    return re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\d+\.?\d*|\S", text)


def shingles(tok, n=8):
    return {tuple(tok[i:i + n]) for i in range(max(0, len(tok) - n + 1))}


def main():
    mine = tokens(sys.argv[1]); sm = shingles(mine)
    for ref in sys.argv[2:]:
        sr = shingles(tokens(ref))
        common = sm & sr
        print("%s vs %s: %d / %d shingles of the file also occur in the reference = %.1f %%" % (sys.argv[1], ref, len(common), len(sm), 100.0 * len(common) / max(1, len(sm))))
        # worst 15-line window
        lines = open(sys.argv[1], errors="replace").read().split("\n")
        worst = (0.0, 0)
        for i in range(0, max(1, len(lines) - 15)):
            tk = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\d+\.?\d*|\S", "\n".join(lines[i:i + 15]))
            sw = shingles(tk)
            if len(sw) >= 40:
                frac = len(sw & sr) / len(sw)
                if frac > worst[0]: worst = (frac, i + 1)
        print("   worst 15-line window: %.0f %% at line %d" % (100 * worst[0], worst[1]))


if __name__ == "__main__":
    main()
