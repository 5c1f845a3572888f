"""Host-side mirror of the reference's feature dump (`vidsitu_code/feat_extractor.py:76-112`,
SURVEY.md 8a row A8): trunk -> trimmed head -> one `[E, feat_dim]` float32 `.npy` per video, named
`<vseg_name>_feats.npy` under `<cfg.ds.vsitu.vsitu_frm_feats>/<mdl_name>/`, the format
`VsituDS.get_frm_feats_all` (`vidsitu_code/dat_loader.py:503-511`) reads back for the TxEncoder
models.  Same class / method names and the same loop; the model call runs on the HIP kernels.
"""
from pathlib import Path

import numpy as np
import torch

from . import synth_data


class SynthFrameDataset:
    """`VsituDS_All` stand-in (`feat_extractor.py:40-74`): `vseg_lst` names + `all_itemgetter`
    items of the A0 contract with seeded synthetic frames."""

    def __init__(self, cfg, comm, n_videos, n_ev=5, seed=0, crop=None, names=None):
        self.cfg, self.comm = cfg, comm
        self.n_ev, self.seed, self.crop = n_ev, seed, crop
        self.vseg_lst = names or [f"v_synth{ix:05d}_seg_0_10" for ix in range(n_videos)]

    def __len__(self):
        return len(self.vseg_lst)

    def __getitem__(self, idx):
        b = synth_data.synth_batch(self.cfg, self.comm, bs=1, n_ev=self.n_ev, seed=self.seed + idx,
                                   crop=self.crop)
        out = {k: v[0] for k, v in b.items() if k.startswith("frms_")}
        out["vseg_idx"] = torch.tensor(idx).long()
        return out


class SimpleLoader:
    """Sequential batches with `.dataset` (the two attributes `forward_all` uses of a DataLoader);
    collate = stack per key (`utils/dat_utils.py:81-109` for tensors)."""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, batch_size

    def __iter__(self):
        for i0 in range(0, len(self.dataset), self.batch_size):
            items = [self.dataset[i] for i in range(i0, min(len(self.dataset), i0 + self.batch_size))]
            yield {k: torch.stack([it[k] for it in items]) for k in items[0]}

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size


class FeatExtract:
    def __init__(self, cfg):
        self.cfg = cfg

    def set_mdl_dl(self, mdl, dl, mdl_name: str, split_name: str):
        self.mdl = mdl
        self.dl = dl
        self.mdl_name = mdl_name
        self.split_name = split_name
        out_tdir = Path(self.cfg.ds.vsitu.vsitu_frm_feats) / f"{mdl_name}"
        out_tdir.mkdir(exist_ok=True, parents=True)
        self.out_tdir = out_tdir

    @torch.no_grad()
    def forward_all(self, device=None, dtype=torch.bfloat16):
        """feat_extractor.py:90-112.  Frames go to the GPU as bf16 (the trunk's storage type)."""
        device = device or torch.device("cuda")
        vseg_lst = self.dl.dataset.vseg_lst
        written = []
        for batch in self.dl:
            batch_gpu = {k: (v.to(device=device, dtype=dtype) if v.is_floating_point() else v.to(device))
                         for k, v in batch.items()}
            feat_out = self.mdl.forward_encoder(batch_gpu)
            head_out = self.mdl.head(feat_out)
            head_out = head_out.permute((0, 2, 3, 4, 1))  # (N, C, 1, 1, 1) -> (N, 1, 1, 1, C)
            B = len(batch["vseg_idx"])
            assert head_out.size(1) == 1 and head_out.size(2) == 1 and head_out.size(3) == 1
            n_ev = head_out.size(0) // B  # the reference's literal 5
            out_np = head_out.reshape(B, n_ev, -1).float().cpu().numpy()
            for vix in range(B):
                vseg_name = vseg_lst[int(batch["vseg_idx"][vix])]
                out_np_name = self.out_tdir / f"{vseg_name}_feats.npy"
                np.save(out_np_name, out_np[vix])
                written.append(out_np_name)
        return written


def read_frm_feats(feats_dir, vseg_name):
    """`VsituDS.get_frm_feats_all` (`dat_loader.py:503-511`): -> {"frm_feats": f32 [E, D]}."""
    arr = np.load(Path(feats_dir) / f"{vseg_name}_feats.npy")
    return {"frm_feats": torch.from_numpy(arr).float()}


def main(mdl_resume_path: str, mdl_name_used: str, is_cu: bool = False, splits=("valid", "train"), n_videos=None,
         calibrate: int = 2, **kwargs):
    """`python -m vidsitu_amd.feat_extractor <weights> <name> [--is_cu=True] [--dotted.key=value ...]`
    (`feat_extractor.py:119-176`): build the configured model, load a TRAINED checkpoint (the trainer's file format,
    `module.` prefixes stripped) or -- `is_cu` -- the Kinetics model-zoo Caffe2 pickle into `mdl.sf_mdl`, and write
    `<vsitu_frm_feats>/<name>/<vseg>_feats.npy` ([E, 2304] / [E, 2048] float32) for every video of every split.
    The videos are the synthetic stand-in dataset (`SynthFrameDataset`; the 50 GB frame dataset is out of scope), so what this
    entry point pins is the flow: weights -> eval trunk on the HIP kernels -> head -> files the TxEncoder rows read back.
    `--calibrate=N` (default 2; 0 = off, the reference's behaviour: it has no such step): before the first split the
    model measures, on the clips of the first N videos OF THE DATASET BEING EXTRACTED (the first split's loader -- the
    evaluation distribution itself, never a stand-in: a constant measured on other data is a data-dependent bias, not a
    correction), the per-channel constants the bf16 rounding of its convolution weights adds and folds their correction
    into the BN shifts (`SFBase.calibrate_weight_rounding`): features within 1e-3 of the fp32 reference's instead of
    3e-3 (tests/test_gpu_parity_full.py; spread over clips and under a calibration / evaluation distribution shift:
    profiles/parity_eval.json), at no cost per forward.  Calibration must precede any hipGraph capture of the eval
    forward: a captured graph keeps the fold tensors it was recorded with."""
    from . import checkpoint, synth_data
    from .extended_config import get_cfg
    from .mdl_selector import get_mdl_loss_eval

    cfg = get_cfg(kwargs)
    cfg.num_gpus, cfg.do_dist = 1, False
    comm = synth_data.make_comm(cfg)
    mdl = get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm)
    if is_cu:
        print("Using Caffe2 checkpoint")
        cfg.sf_mdl.TRAIN.CHECKPOINT_FILE_PATH, cfg.sf_mdl.TRAIN.CHECKPOINT_TYPE = mdl_resume_path, "caffe2"
        checkpoint.load_sf_pretrained(cfg, mdl)
    else:
        got = checkpoint.load_model_dict(mdl_resume_path, mdl, None, load_opt=False, strict=True)
        if got is None:
            raise FileNotFoundError(mdl_resume_path)
    mdl = mdl.to(torch.device("cuda")).eval()
    feat_ext, written = FeatExtract(cfg), []
    n = int(n_videos) if n_videos is not None else int(cfg.synth.num_videos)
    for si, split in enumerate(splits):
        ds = SynthFrameDataset(cfg, comm, n, n_ev=cfg.ds.vsitu.num_ev, seed=cfg.synth.seed + 1000 * si,
                               names=[f"{split}_v{i:05d}_seg_0-10" for i in range(n)])
        if si == 0 and int(calibrate) > 0 and hasattr(mdl, "calibrate_weight_rounding"):
            n_vid = min(int(calibrate), len(ds))
            cal = next(iter(SimpleLoader(ds, n_vid)))
            cal = {k: (v.to(device="cuda", dtype=torch.bfloat16) if v.is_floating_point() else v.to("cuda"))
                   for k, v in cal.items()}
            n_cal = mdl.calibrate_weight_rounding(cal)
            print(f"weight-rounding correction calibrated on the first {n_vid} video(s) of split {split!r}: "
                  f"{n_cal} convolutions")
        feat_ext.set_mdl_dl(mdl, SimpleLoader(ds, max(1, int(cfg.train.bsv))), mdl_name=mdl_name_used, split_name=split)
        written += feat_ext.forward_all()
    print(f"wrote {len(written)} feature files under {feat_ext.out_tdir}")
    return written


if __name__ == "__main__":
    import sys

    if len(sys.argv) < 3:
        sys.exit("usage: python -m vidsitu_amd.feat_extractor <weights> <name> [--is_cu=True] [--dotted.key=value ...]")
    kw = {}
    for a in sys.argv[3:]:
        k, v = a[2:].split("=", 1)
        kw[k] = v
    is_cu = str(kw.pop("is_cu", "False")) in ("1", "True", "true")
    if "calibrate" in kw:
        kw["calibrate"] = int(kw["calibrate"])
    if "splits" in kw:
        kw["splits"] = tuple(kw["splits"].split(","))
    main(sys.argv[1], sys.argv[2], is_cu=is_cu, **kw)
