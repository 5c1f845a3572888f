"""Config surface of the hot path: `configs/vsitu_cfg.yml` + `--dotted.key=value`.

Mirrors the behaviour of the reference's `vidsitu_code/extended_config.py`
(`CfgProcessor`: :40-202) without yacs / slowfast / fairseq, none of which exist
here: the two third-party default tables the reference merges in
(`slowfast.config.defaults.get_cfg()` -> `cfg.sf_mdl`, fairseq `transformer`
arch defaults -> `cfg.tx_dec`, :146-195) are restated below for the keys the
SlowFast -> TxEncoder path reads.

Kept semantics: the name -> file maps (:14-24), existence + type assertion of
every overridden key (:83-111), float-looking YAML scalars such as `1e-4`
parse as float (`utils/_init_stuff.py:4-17`).
"""
import ast
import copy
import os
import re

import yaml

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sf_mdl_to_cfg_fpath_dct = {
    "slow_fast_nl_r50_8x8": "./configs/vsitu_mdl_cfgs/Kinetics_c2_SLOWFAST_8x8_R50.yaml",
    "i3d_r50_8x8": "./configs/vsitu_mdl_cfgs/Kinetics_c2_I3D_8x8_R50.yaml",
    "i3d_r50_nl_8x8": "./configs/vsitu_mdl_cfgs/Kinetics_c2_I3D_NLN_8x8_R50.yaml",
    "i3d_tiny_nl": "./configs/vsitu_mdl_cfgs/I3D_tiny_nl.yaml",
    "i3d_tiny": "./configs/vsitu_mdl_cfgs/I3D_tiny.yaml",
    "slow_fast_mini": "./configs/vsitu_mdl_cfgs/SLOWFAST_mini.yaml",
}
tx_to_cfg_fpath_dct = {"transformer": "./configs/vsitu_tx_cfgs/transformer.yaml"}

# scientific-notation floats without a dot ("1e-4") are floats, as in the reference
_loader = yaml.SafeLoader
_loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(
        r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
        |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
        |\.[0-9_]+(?:[eE][-+][0-9]+)?
        |[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$""",
        re.X,
    ),
    list("-+0123456789."),
)


class CfgNode(dict):
    """Attribute-access dict (the subset of yacs.CfgNode the hot path uses)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v
        self.__dict__["_frozen"] = False

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen", False):
            raise AttributeError(f"cfg is frozen; cannot set {k}")
        self[k] = v

    def freeze(self, flag=True):
        self.__dict__["_frozen"] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze(flag)

    def defrost(self):
        self.freeze(False)

    def clone(self):
        return copy.deepcopy(self)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}


def sf_defaults():
    """Keys of slowfast.config.defaults the reference reads but no YAML sets."""
    return CfgNode(
        {
            "DATA": {"NUM_FRAMES": 8, "SAMPLING_RATE": 8, "TRAIN_CROP_SIZE": 224,
                     "INPUT_CHANNEL_NUM": [3, 3], "MEAN": [0.45, 0.45, 0.45],
                     "STD": [0.225, 0.225, 0.225], "TARGET_FPS": 30,
                     "REVERSE_INPUT_CHANNEL": False},
            "SLOWFAST": {"ALPHA": 8, "BETA_INV": 8, "FUSION_CONV_CHANNEL_RATIO": 2,
                         "FUSION_KERNEL_SZ": 5},
            "RESNET": {"ZERO_INIT_FINAL_BN": False, "WIDTH_PER_GROUP": 64, "NUM_GROUPS": 1,
                       "DEPTH": 50, "TRANS_FUNC": "bottleneck_transform", "STRIDE_1X1": False,
                       "INPLACE_RELU": True, "NUM_BLOCK_TEMP_KERNEL": [[3], [4], [6], [3]],
                       "SPATIAL_STRIDES": [[1], [2], [2], [2]],
                       "SPATIAL_DILATIONS": [[1], [1], [1], [1]]},
            "NONLOCAL": {"LOCATION": [[[]], [[]], [[]], [[]]], "GROUP": [[1], [1], [1], [1]],
                         "INSTANTIATION": "dot_product",
                         "POOL": [[[1, 2, 2], [1, 2, 2]], [[1, 2, 2], [1, 2, 2]], [[1, 2, 2], [1, 2, 2]],
                                  [[1, 2, 2], [1, 2, 2]]]},
            "BN": {"EPSILON": 1e-5, "MOMENTUM": 0.1, "NORM_TYPE": "batchnorm"},
            "MODEL": {"ARCH": "slowfast", "MODEL_NAME": "SlowFast", "NUM_CLASSES": 400,
                      "SINGLE_PATHWAY_ARCH": ["c2d", "i3d", "slow"],
                      "MULTI_PATHWAY_ARCH": ["slowfast"], "FC_INIT_STD": 0.01},
            "DETECTION": {"ENABLE": False},
            # where `mdl.load_sf_pretrained` finds the Kinetics weights (utils/trn_utils.py:358-375); the model YAMLs
            # name the model-zoo Caffe2 pickles as the reference's do
            "TRAIN": {"CHECKPOINT_FILE_PATH": "", "CHECKPOINT_TYPE": "pytorch"},
        }
    )


def tx_defaults():
    """fairseq `transformer` arch defaults the reference merges under cfg.tx_dec."""
    return CfgNode(
        {"encoder_embed_dim": 512, "encoder_ffn_embed_dim": 2048, "encoder_layers": 6,
         "encoder_attention_heads": 8, "encoder_normalize_before": False,
         "decoder_embed_dim": 512, "decoder_ffn_embed_dim": 2048, "decoder_layers": 6,
         "decoder_attention_heads": 8, "decoder_normalize_before": False,
         "attention_dropout": 0.0, "activation_dropout": 0.0, "activation_fn": "relu",
         "dropout": 0.1, "decoder_output_dim": 512, "decoder_input_dim": 512,
         "max_source_positions": 1024, "max_target_positions": 1024}
    )


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = CfgNode(v) if isinstance(v, dict) else v


def _load_yaml(path):
    if not os.path.isabs(path):
        path = os.path.join(_REPO, path)
    assert os.path.exists(path), f"{path} does not exist"
    with open(path) as f:
        return yaml.load(f, Loader=_loader)


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


class CfgProcessor:
    def __init__(self, cfg_pth="./configs/vsitu_cfg.yml"):
        self.cfg_pth = cfg_pth

    def get_vsitu_default_cfg(self):
        return CfgNode(_load_yaml(self.cfg_pth))

    @staticmethod
    def update_one_full_key(cfg, full_key, v):
        d = cfg
        keys = full_key.split(".")
        for sub in keys[:-1]:
            assert sub in d, f"key {full_key} doesnot exist"
            d = d[sub]
        sub = keys[-1]
        assert sub in d, f"key {full_key} doesnot exist"
        value = _decode(v)
        if isinstance(d[sub], float) and isinstance(value, int):
            value = float(value)
        assert isinstance(value, type(d[sub])), (
            f"type mismatch for {full_key}: {type(value).__name__} vs {type(d[sub]).__name__}")
        d[sub] = value

    def update_from_dict(self, cfg, dct):
        for k, v in dct.items():
            self.update_one_full_key(cfg, k, v)
        return cfg

    def pre_proc_config(self, cfg, dct=None):
        """Select + merge the trunk and transformer sub-configs
        (extended_config.py:146-195)."""
        dct = dct or {}
        for k in ("mdl.sf_mdl_name", "mdl.tx_dec_mdl_name"):
            if k in dct:
                self.update_one_full_key(cfg, k, dct[k])
        sf = sf_defaults()
        _merge(sf, _load_yaml(sf_mdl_to_cfg_fpath_dct[cfg.mdl.sf_mdl_name]))
        cfg.sf_mdl = sf
        tx = tx_defaults()
        _merge(tx, _load_yaml(tx_to_cfg_fpath_dct[cfg.mdl.tx_dec_mdl_name]))
        cfg.tx_dec = tx
        return cfg

    def post_proc_config(self, cfg):
        return cfg


def get_cfg(overrides=None, cfg_pth="./configs/vsitu_cfg.yml"):
    """One call: defaults -> sub-config merge -> dotted overrides."""
    overrides = dict(overrides or {})
    cp = CfgProcessor(cfg_pth)
    cfg = cp.get_vsitu_default_cfg()
    cfg = cp.pre_proc_config(cfg, overrides)
    cfg = cp.update_from_dict(cfg, overrides)
    return cp.post_proc_config(cfg)
