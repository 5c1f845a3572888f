"""Synthetic stand-in for the VidSitu frame dataset, producing the exact batch
dict of the reference (`VsituDS.get_frms_all`, `vidsitu_code/dat_loader.py:454-501`;
`vb_only_item_getter` :528-534; collate `utils/dat_utils.py:81-109`):

    frms_ev_fast_tensor  f32 [B, E, 3, T, H, W]      (E = 5 events per 10-s video)
    frms_ev_slow_tensor  f32 [B, E, 3, T/alpha, H, W]  (multi-pathway archs only)
    vseg_idx             i64 [B]
    label_tensor         i64 [B, E]

Frame sampling rules that the real loader applies are kept as pure functions so
they can be pinned by tests: clip centres 30/90/150/210/270 (`dat_loader.py:70-72`),
`get_sequence` clamping (`utils/video_utils.py:18-38`) and the slow-pathway index
`linspace(0, T-1, T//alpha).long()` (`utils/video_utils.py:59-65`).
"""
from types import SimpleNamespace

import torch


def cent_frm_per_ev(fps=30, num_ev=5):
    return {f"Ev{ix + 1}": int((ix + 1 / 2) * fps * 2) for ix in range(num_ev)}


def get_sequence(center_idx, half_len, sample_rate, max_num_frames):
    seq = list(range(center_idx - half_len, center_idx + half_len, sample_rate))
    return [min(max(s, 0), max_num_frames - 1) for s in seq]


def slow_index(t, alpha):
    return torch.linspace(0, t - 1, t // alpha).long()


class SynthGPT2Tok:
    """Stand-in for `comm.gpt2_hf_tok` (`dat_loader.py:87`: a GPT2TokenizerFast with added pad /
    SRL tokens): the members the decoder path uses -- `len()`, `pad()/eos()/unk()` (fairseq
    dictionary style, `hf_gpt2_fseq.py:155`, `seq_gen.py:81-83`), `pad_token_id`, `eos_token_id`,
    `decode`.  Token ids: eos = vocab-2 (GPT-2's <|endoftext|> is the last base token), pad =
    vocab-1 (the added token)."""

    def __init__(self, vocab_size):
        self.vocab_size = vocab_size
        self.eos_token_id = vocab_size - 2
        self.pad_token_id = vocab_size - 1
        self.unk_token_id = self.eos_token_id  # GPT-2: unk_token == eos_token

    def __len__(self):
        return self.vocab_size

    def pad(self):
        return self.pad_token_id

    def eos(self):
        return self.eos_token_id

    def unk(self):
        return self.unk_token_id

    def decode(self, ids, skip_special_tokens=True):
        special = {self.pad_token_id, self.eos_token_id} if skip_special_tokens else set()
        return " ".join(f"t{int(i)}" for i in ids if int(i) not in special)


def make_comm(cfg):
    sf = cfg.sf_mdl
    arch = sf.MODEL.ARCH
    path_type = "multi" if arch in sf.MODEL.MULTI_PATHWAY_ARCH else "single"
    nv = cfg.synth.num_verbs if "synth" in cfg else 1564
    ntok = cfg.synth.gpt2_vocab if ("synth" in cfg and "gpt2_vocab" in cfg.synth) else 50259
    return SimpleNamespace(path_type=path_type, vb_id_vocab=[f"verb_{i}" for i in range(nv)],
                           num_frms=sf.DATA.NUM_FRAMES, sampling_rate=sf.DATA.SAMPLING_RATE,
                           gpt2_hf_tok=SynthGPT2Tok(ntok))


def synth_u8_batch(cfg, comm, bs, n_ev=5, seed=1234, device="cpu", crop=None):
    """The A0 batch as the loader's PIL step leaves it (`dat_loader.py:183-191`): uint8 RGB frames
    `frms_ev_fast_u8` [B, E, T, H, W, 3] -- the optional input of the GPU normalise / pack kernel.
    `reference_tensors(batch, cfg)` gives the fp32 tensors the reference would have built."""
    sf = cfg.sf_mdl
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = sf.DATA.NUM_FRAMES
    hw = crop or sf.DATA.TRAIN_CROP_SIZE
    fr = torch.randint(0, 256, (bs, n_ev, t, hw, hw, 3), generator=g, dtype=torch.int32).to(torch.uint8)
    return {
        "frms_ev_fast_u8": fr.to(device),
        "vseg_idx": torch.arange(bs, dtype=torch.long, device=device),
        "label_tensor": torch.randint(0, len(comm.vb_id_vocab), (bs, n_ev), generator=g).to(device),
    }


def synth_video_u8_batch(cfg, comm, bs, n_ev=5, seed=1234, device="cpu", crop=None):
    """uint8 frames with the statistics of decoded video rather than of noise (the calibration-robustness test and
    `feat_extractor`'s stand-in dataset): a spatially and temporally low-pass colour field (coarse noise, trilinear
    upsampling: neighbouring pixels and frames are strongly correlated), per-VIDEO brightness and contrast (events of a
    video share them, videos differ), a little sensor noise, clipped to 0..255.  Same dict as `synth_u8_batch`."""
    import torch.nn.functional as F

    sf = cfg.sf_mdl
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = sf.DATA.NUM_FRAMES
    hw = crop or sf.DATA.TRAIN_CROP_SIZE
    coarse = torch.randn(bs * n_ev, 3, max(2, t // 8), max(2, hw // 16), max(2, hw // 16), generator=g)
    mid = torch.randn(bs * n_ev, 3, max(2, t // 4), max(2, hw // 4), max(2, hw // 4), generator=g)
    field = (F.interpolate(coarse, size=(t, hw, hw), mode="trilinear", align_corners=False)
             + 0.35 * F.interpolate(mid, size=(t, hw, hw), mode="trilinear", align_corners=False))
    field = (field / field.std()).view(bs, n_ev, 3, t, hw, hw)                # unit spread: `contrast` is in grey levels
    bright = 60.0 + 120.0 * torch.rand(bs, 1, 1, 1, 1, 1, generator=g)        # per video
    tint = 12.0 * torch.randn(bs, 1, 3, 1, 1, 1, generator=g)                 # per video and colour channel
    contrast = 25.0 + 55.0 * torch.rand(bs, 1, 1, 1, 1, 1, generator=g)       # per video
    noise = 3.0 * torch.randn(bs, n_ev, 3, t, hw, hw, generator=g)
    img = (bright + tint + contrast * field + noise).round().clamp_(0, 255).to(torch.uint8)
    fr = img.permute(0, 1, 3, 4, 5, 2).contiguous()                           # [B, E, T, H, W, 3]
    return {
        "frms_ev_fast_u8": fr.to(device),
        "vseg_idx": torch.arange(bs, dtype=torch.long, device=device),
        "label_tensor": torch.randint(0, len(comm.vb_id_vocab), (bs, n_ev), generator=g).to(device),
    }


def reference_tensors(batch_u8, cfg, comm):
    """uint8 frames -> the reference's fp32 batch tensors, same operation order as
    `tensor_normalize` (`utils/video_utils.py:147-164`) and `pack_pathway_output` (:41-74)."""
    sf = cfg.sf_mdl
    fr = batch_u8["frms_ev_fast_u8"].cpu()
    x = fr.float() / 255.0
    x = x - torch.tensor(list(sf.DATA.MEAN))
    x = x / torch.tensor(list(sf.DATA.STD))
    fast = x.permute(0, 1, 5, 2, 3, 4).contiguous()  # [B, E, C, T, H, W]
    out = {"frms_ev_fast_tensor": fast, "vseg_idx": batch_u8["vseg_idx"].cpu(),
           "label_tensor": batch_u8["label_tensor"].cpu()}
    if comm.path_type == "multi":
        out["frms_ev_slow_tensor"] = fast.index_select(3, slow_index(fast.shape[3], sf.SLOWFAST.ALPHA))
    return out


def synth_srl_batch(comm, bs, n_ev=5, n_ann=1, seq_len=60, feat_dim=2304, seed=1234, device="cpu"):
    """Batch of the vb_arg contract (`dat_loader.py:220-452,503-511`): `seq_out_by_ev`
    i64 [B,E,n_ann,60] right-padded SRL token ids, `seq_out_lens_by_ev` {0,1} mask of the same
    shape, `vb_out_by_ev` i64 [B,E,n_ann,5], `frm_feats` f32 [B,E,feat_dim], `vseg_idx`."""
    tok = comm.gpt2_hf_tok
    g = torch.Generator(device="cpu").manual_seed(seed)
    seq = torch.full((bs, n_ev, n_ann, seq_len), tok.pad_token_id, dtype=torch.long)
    mask = torch.zeros((bs, n_ev, n_ann, seq_len), dtype=torch.long)
    for b in range(bs):
        for e in range(n_ev):
            for a in range(n_ann):
                n = int(torch.randint(4, max(5, seq_len * 2 // 3), (1,), generator=g))
                seq[b, e, a, :n] = torch.randint(0, len(tok) - 2, (n,), generator=g)
                seq[b, e, a, n] = tok.eos_token_id
                mask[b, e, a, : n + 1] = 1
    return {
        "seq_out_by_ev": seq.to(device), "seq_out_lens_by_ev": mask.to(device),
        "vb_out_by_ev": seq[..., :5].contiguous().to(device),
        "frm_feats": torch.randn(bs, n_ev, feat_dim, generator=g).to(device),
        "vseg_idx": torch.arange(bs, dtype=torch.long, device=device),
    }


def synth_batch(cfg, comm, bs, n_ev=5, seed=1234, device="cpu", dtype=torch.float32,
                crop=None):
    """One batch of the A0 contract with seeded N(0,1) 'frames'."""
    sf = cfg.sf_mdl
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = sf.DATA.NUM_FRAMES
    hw = crop or sf.DATA.TRAIN_CROP_SIZE
    fast = torch.randn(bs, n_ev, 3, t, hw, hw, generator=g)
    batch = {
        "frms_ev_fast_tensor": fast.to(device=device, dtype=dtype),
        "vseg_idx": torch.arange(bs, dtype=torch.long, device=device),
        "label_tensor": torch.randint(0, len(comm.vb_id_vocab), (bs, n_ev), generator=g).to(device),
    }
    if comm.path_type == "multi":
        idx = slow_index(t, sf.SLOWFAST.ALPHA)
        batch["frms_ev_slow_tensor"] = fast.index_select(3, idx).to(device=device, dtype=dtype)
    return batch
