"""Joins the per-shape rows of `scripts/layer_table.py` runs made with different HC_CONV_SHORT settings (one box) into one table:
which 1 x 1 / short-loop gather-conv shapes prefer the four-workgroups-per-CU form.  usage: short_form_table.py prefix (files prefix_N.txt)"""
import re
import sys


def load(f):
    d = {}
    for ln in open(f):
        if ln.startswith('conv_gather') and 'taps' in ln:
            m = re.match(r'conv_gather\s+(.*?taps\d+.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)', ln)
            if m:
                d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return d


pre = sys.argv[1]
keys = (0, 128, 256, 1024)
T = {s: load(f'{pre}_{s}.txt') for s in keys}
rows = sorted(T[0], key=lambda k: -T[0][k][0] * T[0][k][1])
print(f"{'shape (us per launch at HC_CONV_SHORT = ...)':62s}  n      0    128    256   1024")
for k in rows:
    if all(k in T[s] for s in T):
        print(f"{k:62s} {T[0][k][0]:2d} " + " ".join(f"{T[s][k][1]:6.1f}" for s in keys))
