#!/usr/bin/env python3
"""Summarise a `gcov -b -c` file of oracle/bliss_oracle.c: line / branch coverage in total and per function, and every branch
outcome that was never taken with its source line.  Test infrastructure (see oracle/Makefile, target `coverage`).

gcov counts two outcomes per condition ("branch 0", "branch 1") and lists calls separately; a line `branch N never executed`
belongs to code that was not reached at all.  Coverage here = outcomes taken at least once / all outcomes."""
import re
import sys


def parse(path):
    funcs = []          # [name, first_line]
    lines = {}          # lineno -> (count or None, text)
    branches = []       # (lineno, index, taken_count or None)
    cur = 0
    for raw in open(path, errors="replace"):
        m = re.match(r"^function (\S+) called (\d+)", raw)
        if m:
            funcs.append([m.group(1), None, int(m.group(2))])
            continue
        m = re.match(r"^branch\s+(\d+)\s+(never executed|taken (\d+))", raw)
        if m:
            branches.append((cur, int(m.group(1)), None if m.group(3) is None else int(m.group(3))))
            continue
        if raw.startswith("call") or raw.startswith("unconditional"):
            continue
        m = re.match(r"^\s*([0-9#=\-]+\*?):\s*(\d+):(.*)$", raw)
        if m:
            cnt, no, text = m.group(1), int(m.group(2)), m.group(3)
            if no == 0:
                continue
            cur = no
            c = None if cnt == "-" else (0 if cnt.strip("*") in ("#####", "=====") else int(cnt.strip("*")))
            lines[no] = (c, text)
            if funcs and funcs[-1][1] is None and c is not None:
                funcs[-1][1] = no
    return funcs, lines, branches


def main():
    funcs, lines, branches = parse(sys.argv[1])
    ex = [c for c, _ in lines.values() if c is not None]
    taken = sum(1 for _, _, t in branches if t)
    print(f"lines executed {sum(1 for c in ex if c)} of {len(ex)} = {100.0 * sum(1 for c in ex if c) / max(1, len(ex)):.1f} %")
    print(f"branch outcomes taken {taken} of {len(branches)} = {100.0 * taken / max(1, len(branches)):.1f} %")
    print(f"functions never called: {', '.join(f[0] for f in funcs if f[2] == 0) or 'none'}")
    print()
    starts = sorted((f[1], f[0]) for f in funcs if f[1] is not None)

    def func_of(no):
        name = "?"
        for s, n in starts:
            if s <= no:
                name = n
            else:
                break
        return name

    per = {}
    for no, idx, t in branches:
        f = func_of(no)
        a = per.setdefault(f, [0, 0])
        a[1] += 1
        a[0] += 1 if t else 0
    print("per function (branch outcomes taken / all):")
    for f, (a, b) in sorted(per.items(), key=lambda kv: kv[1][0] / kv[1][1]):
        if a < b:
            print(f"  {f:40s} {a:4d} / {b:4d}")
    print()
    print("branch outcomes never taken (line: source):")
    seen = set()
    for no, idx, t in branches:
        if not t and no not in seen:
            seen.add(no)
            n_missing = sum(1 for n2, _, t2 in branches if n2 == no and not t2)
            state = "not reached" if lines.get(no, (0, ""))[0] in (0, None) else f"{n_missing} outcome(s) missing"
            print(f"  {no:5d} [{func_of(no)}] ({state}): {lines.get(no, (None, ''))[1].strip()[:150]}")


if __name__ == "__main__":
    main()
