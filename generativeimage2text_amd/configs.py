"""Model hyper-parameters of the GIT family.

Mirrors what the reference spreads over `model.py:9-61` (decoder: always 768 hidden / 6 layers /
12 heads / 3072 FFN / vocab 30522 / 1024 positions), `model.py:63-91` + CLIP `build_model`
(encoder: ViT-B/16 or ViT-L/14) and `aux_data/models/<name>/parameter.yaml` (per-model overrides).
"""
from __future__ import annotations

import dataclasses
import os
from typing import Dict, Optional


@dataclasses.dataclass(frozen=True)
class GitModelConfig:
    name: str = "GIT_BASE"
    image_size: int = 224           # param['test_crop_size']
    patch: int = 16
    vit_width: int = 768            # == param['visual_feature_size']
    vit_layers: int = 12
    vit_heads: int = 12
    dec_hidden: int = 768
    dec_layers: int = 6
    dec_heads: int = 12
    dec_ffn: int = 3072
    vocab: int = 30522
    max_pos: int = 1024
    num_frames: int = 0             # param['num_image_with_embedding']
    sos: int = 101                  # tokenizer.cls_token_id
    eos: int = 102                  # tokenizer.sep_token_id
    test_respect_ratio_max: Optional[int] = None

    @property
    def n_tok(self) -> int:
        return (self.image_size // self.patch) ** 2 + 1

    @property
    def max_image_hw(self):
        """Largest input the engine is sized for: MinMaxResizeForTest(test_crop_size, test_respect_ratio_max) yields
        short side <= test_crop_size and long side <= test_respect_ratio_max (inference.py:29-64, 111-117); models
        without it only see image_size x image_size."""
        if self.test_respect_ratio_max is None:
            return None
        return (int(self.image_size), int(self.test_respect_ratio_max))


_ENCODERS = {
    # model.py:64-67 name_map -> CLIP VisualTransformer(input_resolution, patch, width, layers, heads, out)
    "CLIPViT_B_16": dict(patch=16, vit_width=768, vit_layers=12, vit_heads=12),
    "CLIPViT_L_14": dict(patch=14, vit_width=1024, vit_layers=24, vit_heads=16),
}

# aux_data/models/*/parameter.yaml of the reference, restated (models without a yaml use defaults)
MODEL_PARAMS: Dict[str, dict] = {
    "GIT_BASE": {}, "GIT_BASE_COCO": {}, "GIT_BASE_TEXTCAPS": {},
    "GIT_BASE_VQAv2": {"test_crop_size": 480, "test_respect_ratio_max": 640},
    "GIT_BASE_TEXTVQA": {"test_crop_size": 480, "test_respect_ratio_max": 640},
    "GIT_BASE_VATEX": {"num_image_with_embedding": 6},
    "GIT_BASE_MSRVTT": {"num_image_with_embedding": 6},
    "GIT_BASE_MSRVTT_QA": {"num_image_with_embedding": 6},
    "GIT_LARGE": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024},
    "GIT_LARGE_COCO": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024},
    "GIT_LARGE_TEXTCAPS": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024},
    "GIT_LARGE_R": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024},
    "GIT_LARGE_R_COCO": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024},
    "GIT_LARGE_R_TEXTCAPS": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024},
    "GIT_LARGE_VQAv2": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024,
                        "test_crop_size": 420, "test_respect_ratio_max": 560},
    "GIT_LARGE_TEXTVQA": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024,
                          "test_crop_size": 420, "test_respect_ratio_max": 560},
    "GIT_LARGE_VATEX": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024,
                        "num_image_with_embedding": 6},
    "GIT_LARGE_MSRVTT": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024,
                         "num_image_with_embedding": 6},
    "GIT_LARGE_MSRVTT_QA": {"image_encoder_type": "CLIPViT_L_14", "visual_feature_size": 1024,
                            "num_image_with_embedding": 6},
}


def _merge_into(base: dict, over: dict) -> dict:
    """`over`'s leaves written into `base`, nested dicts merged key by key (what the reference's path-wise update of
    tsv_io.py:102-106 amounts to)."""
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge_into(base[k], v)
        else:
            base[k] = v
    return base


def load_from_yaml_file(file_name: str) -> dict:
    """A parameter.yaml with the reference's `_base_` include chain resolved (tsv_io.py:92-107): the file named by
    `_base_` (relative to the including file) is loaded first and this file's values are written over it."""
    import yaml
    with open(file_name, "r") as fp:
        data = yaml.safe_load(fp)
    while isinstance(data, dict) and "_base_" in data:
        base = load_from_yaml_file(os.path.join(os.path.dirname(file_name), data["_base_"]))
        assert isinstance(base, dict)
        del data["_base_"]
        data = _merge_into(base, data)
    return data if data is not None else {}


def load_model_param(model_name: str, yaml_dir: str = "aux_data/models") -> dict:
    """The `param` dict of a model: `<yaml_dir>/<model_name>/parameter.yaml` when that file exists -- the reference
    reads aux_data/models/<name>/parameter.yaml in the single-image task (inference.py:68-70) and
    output/<name>/parameter.yaml in the TSV task (inference.py:135-137) --, else the built-in restatement of the
    shipped yaml files (MODEL_PARAMS; the reference would silently fall back to GIT_BASE defaults there)."""
    path = os.path.join(yaml_dir, model_name, "parameter.yaml")
    if os.path.isfile(path):
        return load_from_yaml_file(path)
    if model_name not in MODEL_PARAMS:
        raise KeyError(f"unknown GIT model '{model_name}': no {path} and not one of {sorted(MODEL_PARAMS)}")
    return dict(MODEL_PARAMS[model_name])


def config_from_param(param: Optional[dict], name: str = "GIT") -> GitModelConfig:
    """The dict the reference reads from parameter.yaml -> a model config (model.py:9-26, 59)."""
    param = dict(param or {})
    enc = _ENCODERS[param.get("image_encoder_type", "CLIPViT_B_16")]
    vfs = int(param.get("visual_feature_size", 768))
    if vfs != enc["vit_width"]:
        raise ValueError(f"visual_feature_size {vfs} does not match encoder width {enc['vit_width']}")
    return GitModelConfig(
        name=name,
        image_size=int(param.get("test_crop_size", 224)),
        num_frames=int(param.get("num_image_with_embedding") or 0),
        test_respect_ratio_max=param.get("test_respect_ratio_max"),
        **enc,
    )


def config_for_model(model_name: str) -> GitModelConfig:
    if model_name not in MODEL_PARAMS:
        raise KeyError(f"unknown GIT model '{model_name}' (known: {sorted(MODEL_PARAMS)})")
    return config_from_param(MODEL_PARAMS[model_name], name=model_name)
