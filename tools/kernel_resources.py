"""Registers / scratch of every kernel in an ISA listing (hipcc -S ... --cuda-device-only): tools/kernel_resources.py k.s"""
import re, sys
name = None
rows = {}
for l in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        name = m.group(1)
    m = re.match(r'^; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)', l)
    if m and name:
        rows.setdefault(name, {})[m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if v.get('NumVgprs', 0) >= 64:
        short = re.sub(r'^_ZN2ss\d+', '', k)[:70]
        print("%-72s vgpr %3d agpr %3d scratch %4d" % (short, v.get('NumVgprs', 0), v.get('NumAgprs', 0), v.get('ScratchSize', 0)))
