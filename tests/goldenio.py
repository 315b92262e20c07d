"""Loader for the golden fixtures written by tests/golden/make_golden.py."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.name = os.path.basename(path)[:-4]
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.inputs, self.grad_inputs, self.outs, self.cots = {}, {}, [], []
        self.sd, self.grads, self.hasgrad = {}, {}, {}
        for k in z.files:
            if k.startswith("in."):
                self.inputs[k[3:]] = torch.from_numpy(z[k])
            elif k.startswith("grad_in."):
                self.grad_inputs[k[8:]] = torch.from_numpy(z[k])
            elif k.startswith("sd."):
                self.sd[k[3:]] = torch.from_numpy(z[k])
            elif k.startswith("grad."):
                self.grads[k[5:]] = torch.from_numpy(z[k])
            elif k.startswith("hasgrad."):
                self.hasgrad[k[8:]] = bool(z[k])
        n = len([k for k in z.files if k.startswith("out.")])
        self.outs = [torch.from_numpy(z["out.%d" % i]) for i in range(n)]
        self.cots = [torch.from_numpy(z["cot.%d" % i]) for i in range(n)]
        if "layers" in self.meta:
            self.meta["layers"] = [tuple(l) for l in self.meta["layers"]]


MODEL_PREFIXES = ("conv_", "block_", "wavenet_", "rawctc_", "classifier_")   # generator_*.npz holds data-generator stages, not modules


def names(prefix=""):
    found = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
    return [n for n in found if n.startswith(MODEL_PREFIXES)]


def load(name):
    return Golden(os.path.join(GOLDEN_DIR, name + ".npz"))
