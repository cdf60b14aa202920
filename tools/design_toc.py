"""Rewrite the table of contents of DESIGN.md (between the <!-- toc --> markers) with the current line number of every heading."""
import os
import re

p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "DESIGN.md")
s = open(p).read()
a, b = s.index("<!-- toc -->"), s.index("<!-- /toc -->")
for _ in range(3):  # the numbers move with the length of the table: iterate to a fixed point
    lines = s.split("\n")
    toc = ["<!-- toc -->", "| line | section |", "|---|---|"]
    for i, ln in enumerate(lines, 1):
        m = re.match(r"^(##+) (.*)", ln)
        if m and i > s[:b].count("\n") + 1:
            toc.append(f"| {i} | {'&nbsp;&nbsp;' * (len(m.group(1)) - 2)}{m.group(2)[:110]} |")
    s = s[:a] + "\n".join(toc) + "\n" + s[b:]
    a, b = s.index("<!-- toc -->"), s.index("<!-- /toc -->")
open(p, "w").write(s)
