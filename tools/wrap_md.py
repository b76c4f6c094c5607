#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file at WIDTH columns (default 128): paragraphs and list items are re-flowed with their
indentation / bullet kept; tables, headings, code fences and HTML are left alone.   python tools/wrap_md.py FILE [WIDTH]"""
import re, sys, textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 128
out, in_code = [], False
for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        in_code = not in_code
        out.append(line)
        continue
    if in_code or len(line) <= width or line.lstrip().startswith(("|", "#", "<")):
        out.append(line)
        continue
    m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", line)
    indent, bullet = m.group(1), m.group(2) or ""
    body = line[len(indent) + len(bullet):]
    first = indent + bullet
    rest = indent + " " * len(bullet)
    out.extend(textwrap.wrap(body, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False))
open(path, "w").write("\n".join(out))
