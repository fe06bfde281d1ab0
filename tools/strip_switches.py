#!/usr/bin/env python3
"""One-off source clean-up: resolve the timing-experiment preprocessor switches of a kernel source as UNDEFINED
(keep `#ifndef X` / `#else` bodies, drop `#ifdef X` bodies), leaving every other directive alone.

    python tools/strip_switches.py <file> MACRO [MACRO ...]
"""
import re
import sys

path, macros = sys.argv[1], set(sys.argv[2:])
out, stack = [], []          # stack of (is_ours, keeping_now, seen_else)
for line in open(path).read().split("\n"):
    m = re.match(r"\s*#\s*(ifdef|ifndef|if|else|elif|endif)\b\s*(\w*)", line)
    if m:
        kind, name = m.group(1), m.group(2)
        if kind in ("ifdef", "ifndef", "if"):
            ours = kind != "if" and name in macros
            stack.append([ours, (kind == "ifndef") if ours else True, False])
            if ours:
                continue
        elif kind in ("else", "elif"):
            if stack and stack[-1][0]:
                stack[-1][1] = not stack[-1][1]
                continue
        elif kind == "endif":
            top = stack.pop()
            if top[0]:
                continue
    if all(keep for _, keep, _ in stack):
        out.append(line)
open(path, "w").write("\n".join(out))
