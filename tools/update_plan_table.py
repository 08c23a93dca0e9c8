"""Rewrite the planner-field table of INTEGRATION.md from maua-style_amd/plan.py::FIELDS (run after adding or changing a field)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "maua-style_amd"))
import plan  # noqa: E402

path = os.path.join(REPO, "INTEGRATION.md")
s = open(path).read()
head = "| field | default | read by | decides |\n|---|---|---|---|\n"
a = s.index(head) + len(head)
b = s.index("\n\n", a)
rows = [f"| `{k}` | `{d}` | {'library' if who == 'lib' else 'host'} | {what} |" for k, (d, who, what) in plan.FIELDS.items()]
open(path, "w").write(s[:a] + "\n".join(rows) + s[b:])
print(len(rows), "fields")
