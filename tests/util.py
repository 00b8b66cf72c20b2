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


def apply_trained_gains(module, gains, fc_prefixes=("fully_connected",)):
    """The weight scaling of the trained-scale fixtures (tests/golden/gen_golden.py::apply_trained_gains), on one of this
    package's modules or on a plain {key: numpy array} state dict: rnn.weight_ih_* x gains['weight_ih'], rnn.weight_hh_* x
    gains['weight_hh'], fully-connected weights x gains['fully_connected']; biases and convolutions untouched."""
    items = module.items() if isinstance(module, dict) else module.state_dict().items()
    for k, v in items:
        if "weight_ih" in k:
            g = gains["weight_ih"]
        elif "weight_hh" in k:
            g = gains["weight_hh"]
        elif k.startswith(fc_prefixes) and k.endswith("weight"):
            g = gains["fully_connected"]
        else:
            continue
        if isinstance(v, np.ndarray):
            v *= np.float32(g)
        else:
            v.mul_(g)


def apply_ds1_trained_gains(module, gains):
    """tests/golden/gen_golden.py::apply_ds1_trained_gains: fc1-3 weights x fc_pre, fc4 / out weights x fc_post, BiLSTM
    weight_ih / weight_hh x their gains."""
    for k, v in module.state_dict().items():
        if "weight_ih" in k:
            v.mul_(gains["weight_ih"])
        elif "weight_hh" in k:
            v.mul_(gains["weight_hh"])
        elif k.startswith(("fc1", "fc2", "fc3")) and k.endswith("weight"):
            v.mul_(gains["fc_pre"])
        elif k.startswith(("fc4", "out")) and k.endswith("weight"):
            v.mul_(gains["fc_post"])
