"""Host-side logic that needs no GPU: config surface, plugin selector, synthetic batch
contract, frame-sampling golden vectors, state_dict key parity with the oracle."""
import pytest
import torch

from oracle.slowfast_ref import VideoTrunk as RefTrunk, default_sf_cfg
from vidsitu_amd import synth_data
from vidsitu_amd.extended_config import get_cfg
from vidsitu_amd.mdl_selector import get_mdl_loss_eval
from vidsitu_amd.mdl_sf_base import SFBase, SFBase_TxEnc, LossB
from vidsitu_amd.trunk import VideoTrunk


def test_cfg_defaults_and_overrides():
    cfg = get_cfg()
    assert cfg.train.lr == 1e-4 and isinstance(cfg.train.lr, float)  # _init_stuff.py float patch
    assert cfg.DIST_BACKEND == "nccl"
    assert cfg.sf_mdl.SLOWFAST.ALPHA == 4 and cfg.sf_mdl.BN.EPSILON == 1e-5
    assert cfg.sf_mdl.DATA.MEAN == [0.45, 0.45, 0.45]
    assert cfg.tx_dec.encoder_layers == 3 and cfg.tx_dec.max_source_positions == 1024
    cfg = get_cfg({"tx_dec.encoder_layers": 6, "train.bs": 8, "mdl.mdl_name": "sf_base_txenc"})
    assert cfg.tx_dec.encoder_layers == 6 and cfg.train.bs == 8
    with pytest.raises(AssertionError):
        get_cfg({"train.no_such_key": 1})
    with pytest.raises(AssertionError):
        get_cfg({"train.bs": "sixteen"})  # type assertion (extended_config.py:107)


def test_selector_rows():
    cfg = get_cfg()
    assert get_mdl_loss_eval(cfg)["mdl"] is SFBase and get_mdl_loss_eval(cfg)["loss"] is LossB
    cfg = get_cfg({"mdl.mdl_name": "sf_base_txenc"})
    assert get_mdl_loss_eval(cfg)["mdl"] is SFBase_TxEnc
    cfg = get_cfg({"mdl.mdl_name": "gpt2_only"})
    with pytest.raises(NotImplementedError):
        get_mdl_loss_eval(cfg)
    cfg.task_type = "nope"
    with pytest.raises(AssertionError):
        get_mdl_loss_eval(cfg)


def test_frame_sampling_golden_vectors():
    # dat_loader.py:70-72
    assert synth_data.cent_frm_per_ev() == {"Ev1": 30, "Ev2": 90, "Ev3": 150, "Ev4": 210, "Ev5": 270}
    # video_utils.py:18-38 with half_len = 32*2/2, rate 2, 300 frames: Ev1 clamps at 0
    seq = synth_data.get_sequence(30, 32, 2, 300)
    assert len(seq) == 32 and seq[:3] == [0, 0, 2] and seq[-1] == 60
    seq5 = synth_data.get_sequence(270, 32, 2, 300)
    assert seq5[0] == 238 and seq5[-1] == 299 and seq5[-2] == 298
    assert synth_data.slow_index(32, 4).tolist() == [0, 4, 8, 13, 17, 22, 26, 31]


def test_synth_batch_contract():
    cfg = get_cfg()
    comm = synth_data.make_comm(cfg)
    assert comm.path_type == "multi" and len(comm.vb_id_vocab) == 1564
    b = synth_data.synth_batch(cfg, comm, bs=1, n_ev=2, crop=32)
    assert tuple(b["frms_ev_fast_tensor"].shape) == (1, 2, 3, 32, 32, 32)
    assert tuple(b["frms_ev_slow_tensor"].shape) == (1, 2, 3, 8, 32, 32)
    assert b["label_tensor"].dtype == torch.int64 and tuple(b["label_tensor"].shape) == (1, 2)
    idx = synth_data.slow_index(32, 4)
    assert torch.equal(b["frms_ev_slow_tensor"], b["frms_ev_fast_tensor"].index_select(3, idx))


@pytest.mark.parametrize("arch,depth,width,frames", [("slowfast", 50, 64, 32), ("i3d", 50, 64, 8),
                                                     ("i3d", "tiny", 8, 8)])
def test_state_dict_keys_match_oracle(arch, depth, width, frames):
    cfg = default_sf_cfg(arch, depth, width, frames)
    with torch.device("meta"):
        ref = RefTrunk(cfg)
    ours = VideoTrunk(cfg)
    rk = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    ok = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    assert rk == ok


def test_sfbase_builds_with_reference_attribute_names():
    cfg = get_cfg({"mdl.sf_mdl_name": "i3d_tiny"})
    comm = synth_data.make_comm(cfg)
    mdl = SFBase(cfg=cfg, comm=comm)
    for name in ("sf_mdl", "head", "proj_head", "forward_encoder", "forward_decoder"):
        assert hasattr(mdl, name)
    sd = mdl.state_dict()
    assert "proj_head.0.weight" in sd and "proj_head.2.bias" in sd
    assert tuple(sd["proj_head.2.weight"].shape) == (1564, 128)
    with pytest.raises(Exception):  # no silent CPU fallback on the product path
        mdl(synth_data.synth_batch(cfg, comm, bs=1, n_ev=1, crop=32))


REF_CFG = "/root/reference/configs/vsitu_cfg.yml"


@pytest.mark.skipif(not __import__("os").path.exists(REF_CFG), reason="reference tree not present (GPU box)")
def test_reference_cfg_file_loads_unmodified():
    """Drop-in boundary: the reference's own `configs/vsitu_cfg.yml`, byte for byte, goes
    through `get_cfg` with the reference's dotted overrides (`README.md:32-42`), and selects the same
    plugin classes; the keys the hot path reads carry the reference's values."""
    cfg = get_cfg({"task_type": "vb", "mdl.mdl_name": "sf_base", "train.bs": 8, "train.bsv": 8},
                  cfg_pth=REF_CFG)
    assert cfg.mdl.mdl_name == "sf_base" and cfg.task_type == "vb"
    assert cfg.train.lr == 1e-4 and cfg.train.bs == 8
    assert cfg.mdl.sf_mdl_name == "slow_fast_nl_r50_8x8" and cfg.sf_mdl.MODEL.ARCH == "slowfast"
    assert cfg.sf_mdl.DATA.NUM_FRAMES == 32 and cfg.sf_mdl.SLOWFAST.ALPHA == 4
    assert get_mdl_loss_eval(cfg)["mdl"] is SFBase
    cfg = get_cfg({"task_type": "vb_arg", "mdl.mdl_name": "sfpret_txe_txd_vbarg", "mdl.tx_enc_type": "new"},
                  cfg_pth=REF_CFG)
    assert cfg.gen.beam_size >= 1 and cfg.tx_dec.encoder_layers >= 1
    from vidsitu_amd.mdl_sf_base import SFPreFeats_TxEncDec
    assert get_mdl_loss_eval(cfg)["mdl"] is SFPreFeats_TxEncDec
    with pytest.raises(AssertionError):
        get_cfg({"mdl.not_a_key": 1}, cfg_pth=REF_CFG)
