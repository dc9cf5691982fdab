#!/usr/bin/env python3
"""Per-kernel register / instruction-mix table from a gfx950 assembly listing.
usage: hipcc ... -S --cuda-device-only -o x.s file.hip && python tools/isa_stats.py x.s [name-filter]"""
import collections
import re
import subprocess
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, stats = None, collections.OrderedDict()
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        stats[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", line)
    if m:
        op = m.group(1)
        c = stats[cur]
        if op.startswith("v_pk_"): c["pk"] += 1
        if op.startswith("v_mov") or op.startswith("v_pk_mov") or op.startswith("v_xor"): c["mov/xor"] += 1
        if op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): c["vmem"] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
    for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy"):
        m = re.match(rf"^; {key}: (\d+)", line)
        if m:
            stats[cur][key] = int(m.group(1))
for k, c in stats.items():
    if flt not in k or "NumVgprs" not in c:
        continue
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("yagi::", "").replace("void ", "")
    print(f"{name:52s} vgpr {c['NumVgprs']:3d} agpr {c['NumAgprs']:3d} scratch {c['ScratchSize']:4d} occ {c['Occupancy']} | "
          f"valu {c['valu']:5d} (pk {c['pk']:4d}, mov/xor {c['mov/xor']:4d}) lds {c['lds']:4d} vmem {c['vmem']:3d} bar {c['barrier']}")
