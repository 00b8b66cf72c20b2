"""Shared helpers for the tests: golden-vector loading."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.cfg = json.loads(str(z["cfg"]))
        self.a = {k: z[k] for k in z.files if k != "cfg"}

    def __getitem__(self, k):
        return self.a[k]

    def has(self, k):
        return k in self.a

    def sd(self, prefix="sd/"):
        return {k[len(prefix):]: v for k, v in self.a.items() if k.startswith(prefix)}


def unragged(flat, lens):
    out, o = [], 0
    for n in lens:
        out.append([int(v) for v in flat[o:o + int(n)]])
        o += int(n)
    return out


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))
