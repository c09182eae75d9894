#!/usr/bin/env python
"""Which kernel instantiations of librtfs_hip.so does a traced run launch?   tools/kernel_coverage.py <out.txt> <results.db> [<results.db> ...]
The library's kernels are its kernel descriptors (`*.kd` symbols, demangled); the launched ones come from rocprofv3 --kernel-trace databases (rocpd sqlite, one per traced
process).  Output: every instantiation that no traced process launched, grouped by kernel template - the candidates for deletion (or for a test)."""
import collections
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def library_kernels():
    so = os.path.join(ROOT, "rtfs_net_amd", "librtfs_hip.so")
    syms = sorted({s[:-3] for s in subprocess.run(["strings", "-n", "8", so], capture_output=True, text=True).stdout.split("\n") if s.startswith("_ZN4rtfs") and s.endswith(".kd")})
    dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
    return [d for d in dem if d]


def norm(name):  # rocprof truncates nothing but prints `void ` prefixes the same way c++filt does; compare on the text before the argument list
    name = name.strip()
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):  # the '(' that opens the argument list: the first one at template depth 0
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return re.sub(r"^void\s+", "", name[:cut]).replace(" ", "")


def main(out, *dbs):
    launched = collections.Counter()
    for p in dbs:
        try:
            db = sqlite3.connect(p)
            for name, n in db.execute("select name, count(*) from kernels group by name"):
                launched[norm(name)] += n
        except sqlite3.Error as e:
            print(f"# {p}: {e}", file=sys.stderr)
    lib = library_kernels()
    never = collections.defaultdict(list)
    for k in lib:
        if launched[norm(k)] == 0:
            never[norm(k).split("<")[0]].append(k)
    lines = [f"# {len(lib)} kernel instantiations in librtfs_hip.so; {sum(1 for k in lib if launched[norm(k)])} launched by the traced runs ({len(dbs)} process databases); never launched: {sum(len(v) for v in never.values())}"]
    for tmpl in sorted(never):
        lines.append(f"{tmpl}: {len(never[tmpl])}")
        for k in never[tmpl]:
            lines.append("    " + k[:200])
    open(out, "w").write("\n".join(lines) + "\n")
    print(lines[0])


if __name__ == "__main__":
    main(*sys.argv[1:])
