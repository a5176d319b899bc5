"""Joins the per-shape rows of one kernel family (--family=conv_gather by default) of `scripts/layer_table.py` runs made under different dispatch settings (one box) into one
table: which shapes prefer which form.  usage: short_form_table.py file1 file2 ... (column = file name)"""
import os
import re
import sys


FAMILY = 'conv_gather'


def load(f):
    d = {}
    for ln in open(f):
        if ln.startswith(FAMILY + ' ') and not ln.startswith('#'):
            parts = re.split(r'\s{2,}', ln.strip())
            if len(parts) >= 4 and parts[2].isdigit():
                d[parts[1]] = (int(parts[2]), float(parts[3]))
    return d


files = sys.argv[1:]
if files and files[0].startswith('--family='):
    FAMILY = files.pop(0).split('=', 1)[1]
T = [load(f) for f in files]
names = [os.path.basename(f).replace('.txt', '')[-9:] for f in files]
rows = sorted(T[0], key=lambda k: -T[0][k][0] * T[0][k][1])
print(f"{'shape (us per launch)':62s}  n " + " ".join(f"{n:>9s}" for n in names))
tot = [0.0] * len(T)
for k in rows:
    if all(k in t for t in T):
        print(f"{k:62s} {T[0][k][0]:2d} " + " ".join(f"{t[k][1]:9.1f}" for t in T))
        for i, t in enumerate(T):
            tot[i] += t[k][0] * t[k][1]
print(f"{'total us per step over the common rows':62s}    " + " ".join(f"{v:9.0f}" for v in tot))
