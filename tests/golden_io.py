"""Helpers shared by the tests: load a golden case, rebuild operands for the oracle / the product."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


class Case(object):
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(str(z["meta"]))
        self.arr = {k: z[k] for k in z.files if k != "meta"}

    def tensor(self, key):
        return torch.from_numpy(np.array(self.arr[key])) if key in self.arr else None

    def state_dict(self):
        return {k[3:]: torch.from_numpy(np.array(v)) for k, v in self.arr.items() if k.startswith("sd/")}

    def strides(self):
        return {k[7:]: tuple(int(s) for s in v) for k, v in self.arr.items() if k.startswith("stride/")}

    def grads(self):
        return {k[5:]: torch.from_numpy(np.array(v)) for k, v in self.arr.items() if k.startswith("grad/")}


def build_module(meta, device):
    """Instantiate the PRODUCT module (tensorized-rnn_amd) described by a golden case's meta."""
    import contextlib
    import io
    from tensorized_rnn.gru import GRU, TTGRU
    from tensorized_rnn.lstm import LSTM
    from tensorized_rnn.tt_lstm import TTLSTM
    kind = meta["kind"]
    common = dict(input_size=meta["input_size"], hidden_size=meta["hidden_size"],
                  num_layers=meta["num_layers"], device=device)
    with contextlib.redirect_stdout(io.StringIO()):
        if kind in ("ttlstm", "ttgru"):
            cls = TTLSTM if kind == "ttlstm" else TTGRU
            return cls(n_cores=meta["n_cores"], tt_rank=meta["tt_rank"], bias=meta.get("bias", True),
                       is_naive=meta.get("is_naive", False), new_core=meta.get("new_core"), **common)
        cls = LSTM if kind == "lstm" else GRU
        return cls(bias=meta.get("bias", True), **common)
