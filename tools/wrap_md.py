#!/usr/bin/env python3
"""Re-flow the prose of a Markdown file at WIDTH columns (default 128): paragraphs and list items are joined and re-wrapped with their
indentation / bullet kept; tables, headings, code fences, HTML and lines inside fences are left alone.   python tools/wrap_md.py FILE [WIDTH]"""
import re, sys, textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 128
BULLET = re.compile(r"^(\s*)((?:[-*+]|\d+\.)\s+)")
lines = open(path).read().split("\n")
out, block, in_code = [], [], False


def flush():
    if not block:
        return
    first = block[0]
    m = BULLET.match(first)
    if m:
        indent, bullet = m.group(1), m.group(2)
    else:
        indent, bullet = re.match(r"^(\s*)", first).group(1), ""
    text = " ".join([first[len(indent) + len(bullet):].strip()] + [l.strip() for l in block[1:]])
    out.extend(textwrap.wrap(text, width=width, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet),
                             break_long_words=False, break_on_hyphens=False))
    block.clear()


for line in lines:
    if line.lstrip().startswith("```"):
        flush(); in_code = not in_code; out.append(line); continue
    if in_code or line.strip() == "" or line.lstrip().startswith(("|", "#", "<")):
        flush(); out.append(line); continue
    if BULLET.match(line):
        flush()
    block.append(line)
flush()
open(path, "w").write("\n".join(out))
