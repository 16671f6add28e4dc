#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 rocpd database (the default output format): rocpd_stats.py results.db [name-filter]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in c.execute(f"pragma table_info('{ks}')")]
namecol = "kernel_name" if "kernel_name" in cols else "display_name"
names = {r[0]: r[1] for r in c.execute(f"select id, {namecol} from '{ks}'")}
d = defaultdict(list)
for kid, s, e in c.execute(f"select kernel_id, start, end from '{kd}' order by start"):
    d[names[kid]].append(e - s)
tot = 0
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if filt and filt not in n:
        continue
    tot += sum(v)
    print(f"   {n[:72]:72s} calls {len(v):4d} avg {sum(v) / len(v) / 1e3:9.1f} us  min {min(v) / 1e3:9.1f} max {max(v) / 1e3:9.1f}")
print(f"   total {tot / 1e6:.3f} ms")
