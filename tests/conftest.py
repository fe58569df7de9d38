import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_case(name):
    """Rebuild (cfg, weights, frames, search, prefix) of a golden case from its recorded seeds."""
    from oracle import git_oracle as O
    g = load_golden(name)
    cfg = O.CONFIGS[str(g["config"])]
    import ast
    wkw = ast.literal_eval(str(g["weights_kw"]))          # repr() of a plain dict / tuple, written by oracle/make_golden.py
    kind, max_steps, k, pn, lp = ast.literal_eval(str(g["search"]))
    search = O.SearchConfig(kind, max_steps, k, pn, lp)
    w = O.make_weights(cfg, **wkw)
    hw = tuple(int(v) for v in g["hw"]) if "hw" in g and g["hw"].size else None      # non-native resolution cases
    frames = O.make_images(cfg, int(g["batch"]), int(g["frames"]), seed=int(g["image_seed"]), hw=hw)
    prefix = torch.tensor(g["prefix"], dtype=torch.long)[None] if g["prefix"].size else None
    return g, cfg, w, frames, search, prefix


@pytest.fixture
def experiment_build():
    """The measurement build (libgitmi_exp.so) for the duration of a test: the entry points of include/gitmi_experiment.h
    (slower schedules kept for re-measurement, debug hooks) are not exported by the product libraries."""
    from generativeimage2text_amd import engine
    engine.use_experiment_build(True)
    yield
    engine.use_experiment_build(False)
