#!/usr/bin/env python3
"""Puts the current text of integration/raytrace_gpu.c into the ```c block of INTEGRATION.md that prints it
(tests/test_host_and_abi.py checks that the guide shows the file that is compiled, not a sketch of it)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stub = open(os.path.join(ROOT, "integration", "raytrace_gpu.c")).read()
path = os.path.join(ROOT, "INTEGRATION.md")
md = open(path).read()
m = re.search(r"```c\n/\* Drop-in replacement for the reference's src/raytrace\.c.*?\n```\n", md, re.S)
assert m, "the stub's block was not found"
md = md[:m.start()] + "```c\n" + stub + "```\n" + md[m.end():]
open(path, "w").write(md)
print("INTEGRATION.md: stub block updated (%d lines)" % stub.count("\n"))
