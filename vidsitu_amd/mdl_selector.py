"""Plugin surface: `(task_type, mdl.mdl_name)` -> `{mdl, loss, evl}` classes.

Mirrors `vidsitu_code/mdl_selector.py:26-73` for the rows of the hot path; names
outside it raise NotImplementedError exactly like an unknown name does upstream
(:46,71,73).
"""
from .mdl_sf_base import (SFBase, SFBase_TxEnc, LossB, LossLambda, SFPreFeats_TxDec, SFPreFeats_TxEncDec,
                          Simple_GPT2_New, Simple_TxDec)
from .evl_vsitu import EvalB, EvalB_Gen


def get_mdl_loss_eval(cfg):
    assert cfg.task_type in set(["vb", "vb_arg", "evrel", "evforecast"])
    if cfg.task_type == "vb":
        if cfg.mdl.mdl_name == "sf_base":
            return {"mdl": SFBase, "loss": LossB, "evl": EvalB}
        if cfg.mdl.mdl_name == "sf_base_txenc":  # BASELINE config 3 composition
            return {"mdl": SFBase_TxEnc, "loss": LossB, "evl": EvalB}
        raise NotImplementedError
    elif cfg.task_type == "vb_arg":
        if cfg.mdl.mdl_name == "new_gpt2_only":
            return {"mdl": Simple_GPT2_New, "loss": LossLambda, "evl": EvalB_Gen}
        if cfg.mdl.mdl_name == "tx_only":
            return {"mdl": Simple_TxDec, "loss": LossLambda, "evl": EvalB_Gen}
        if cfg.mdl.mdl_name == "sfpret_txed_vbarg":
            return {"mdl": SFPreFeats_TxDec, "loss": LossLambda, "evl": EvalB_Gen}
        if cfg.mdl.mdl_name == "sfpret_txe_txd_vbarg":
            return {"mdl": SFPreFeats_TxEncDec, "loss": LossLambda, "evl": EvalB_Gen}
        raise NotImplementedError
    else:
        raise NotImplementedError
