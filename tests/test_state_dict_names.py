"""Checkpoint-key compatibility without a GPU: the HIP modules expose exactly the parameter / buffer names
of the modules they replace (fairseq's TransformerDecoder as restated by the oracle; slowfast's ResNet with
non-local blocks as restated by the oracle), so the reference's checkpoints load by name."""
import torch

from oracle import slowfast_ref, txdec_ref


def test_transformer_decoder_keys_are_fairseqs():
    from vidsitu_amd.fseq_txdec import TransformerDecoderHip

    m = TransformerDecoderHip(vocab=50, d_model=32, ffn=48, n_head=4, n_layer=2, out_dim=16, pad=49, dropout=0.0)
    sd = m.state_dict()
    want = set(txdec_ref.param_names(2)) | {"embed_positions._float_tensor", "version"}
    assert set(sd) == want
    w = txdec_ref.make_weights(50, 32, 48, 2, 16, 49, seed=1)
    missing, unexpected = m.load_state_dict(w, strict=True)
    assert not missing and not unexpected
    for k, v in w.items():
        assert torch.equal(m.state_dict()[k], v), k
    assert float(m.P("embed_tokens.weight")[49].abs().max()) == 0.0  # padding row


def test_i3d_nonlocal_trunk_keys_are_slowfasts():
    from vidsitu_amd.trunk import VideoTrunk

    cfg = slowfast_ref.default_sf_cfg("i3d", 50, 8, 8)
    cfg.NONLOCAL.LOCATION = [[[]], [[1, 3]], [[1, 3, 5]], [[]]]
    cfg.NONLOCAL.INSTANTIATION = "softmax"
    ref = slowfast_ref.VideoTrunk(cfg)
    ours = VideoTrunk(cfg)
    sd_r, sd_o = ref.state_dict(), ours.state_dict()
    assert set(sd_r) == set(sd_o)
    for k in sd_r:
        assert tuple(sd_r[k].shape) == tuple(sd_o[k].shape), k
    nl = [k for k in sd_o if "_nonlocal" in k]
    assert len({k.split(".")[1] for k in nl}) == 3 and "s3.pathway0_nonlocal1.conv_theta.bias" in sd_o
    ours.load_state_dict(sd_r, strict=True)
