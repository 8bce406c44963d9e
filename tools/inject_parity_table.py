"""Put profiles/<round>_parity_table.md between the parity-table markers of DESIGN.md (section 5).
    python tools/inject_parity_table.py profiles/r06_parity_table.md"""
import sys

src = open(sys.argv[1]).read().rstrip() + "\n"
p = "DESIGN.md"
s = open(p).read()
a, b = "<!-- parity-table:begin -->", "<!-- parity-table:end -->"
i, j = s.index(a) + len(a), s.index(b)
open(p, "w").write(s[:i] + "\n" + f"(`{sys.argv[1]}`)\n\n" + src + s[j:])
print("injected", len(src.splitlines()), "lines")
