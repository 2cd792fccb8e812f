#!/usr/bin/env python3
"""Per-kernel instruction mix and resource table from a hipcc -S device listing.

    python tools/asm_stats.py build/asm/dev.s [kernel-substring ...]
"""
import collections
import re
import sys


def kernels(path):
    name, body, meta = None, [], {}
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            pass
        s = line.strip()
        if s.startswith("; NumVgprs:") or s.startswith("; NumAgprs:") or s.startswith("; ScratchSize:") or s.startswith("; Occupancy:") or s.startswith("; NumSgprs:") or s.startswith("; codeLenInByte"):
            k, v = re.split(r"[:=]", s[2:], maxsplit=1)
            meta[k.strip()] = v.strip()
        if s.startswith("; Occupancy:"):
            yield name, body, meta
            name, body, meta = None, [], {}
            continue
        if s and not s.startswith(";") and not s.startswith(".") and not s.endswith(":"):
            body.append(s.split()[0])


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body, meta in kernels(path):
        if pats and not any(p in name for p in pats):
            continue
        c = collections.Counter(body)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print(f"== {name}\n   {meta}\n   total {len(body)}  valu {valu}  salu {sum(v for k, v in c.items() if k.startswith('s_'))}")
        for k, v in c.most_common(18):
            print(f"   {v:7d}  {k}")


if __name__ == "__main__":
    main()
