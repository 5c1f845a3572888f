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


def make_comm(cfg):
    sf = cfg.sf_mdl
    arch = sf.MODEL.ARCH
    path_type = "multi" if arch in sf.MODEL.MULTI_PATHWAY_ARCH else "single"
    nv = cfg.synth.num_verbs if "synth" in cfg else 1564
    return SimpleNamespace(path_type=path_type, vb_id_vocab=[f"verb_{i}" for i in range(nv)],
                           num_frms=sf.DATA.NUM_FRAMES, sampling_rate=sf.DATA.SAMPLING_RATE)


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
