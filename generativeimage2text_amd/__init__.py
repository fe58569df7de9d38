"""gitmi -- MI355X-native GIT captioning / VQA inference engine.

The compute path lives in ``libgitmi.so`` (hand-written HIP for gfx950, C ABI in include/gitmi.h);
this package is the thin host-side mirror of the reference's Python interface for that path.
Importing the package does not need a GPU; constructing an engine does.
"""
from .configs import GitModelConfig, MODEL_PARAMS, config_for_model, config_from_param  # noqa: F401

__all__ = ["GitModelConfig", "MODEL_PARAMS", "config_for_model", "config_from_param"]
__version__ = "0.1.0"
