"""
ORACLE (test infrastructure, not product code) -- CPU fp32 restatement of the
SlowFast / ResNet video trunk that the reference reaches through
`slowfast.models.video_model_builder.{SlowFast,ResNet}`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this package.  The product path (`vidsitu_amd/`) never does.

PARITY UNPINNED for this file: the arithmetic lives in the third-party package
`slowfast` (github.com/facebookresearch/SlowFast), an un-vendored git submodule
of the reference with no pinned commit (`/root/reference/.gitmodules:9-11`;
era: torch 1.5.1, see `vsitu_pyt_env.yml:175`).  The reference holds no golden
vector, test or fixture for it.  What anchors this restatement:

  * the reference's own call sites -- attribute names `s1, s1_fuse, s2, s2_fuse,
    pathway{p}_pool, s3, s3_fuse, s4, s4_fuse, s5, num_pathways`
    (`vidsitu_code/mdl_sf_base.py:20-34, 45-55`);
  * output widths 32*W and 32*W/BETA_INV (`mdl_sf_base.py:147-150`);
  * the YAML hyper-parameters
    (`configs/vsitu_mdl_cfgs/Kinetics_c2_SLOWFAST_8x8_R50.yaml:10-35`);
  * the published cost of SlowFast-8x8-R50: 65.7 GMAC @256^2 -> 50.31 GMAC @224^2
    and 33,583,800 conv parameters, both re-derived by
    `count_conv_macs_params()` below and asserted in tests/test_oracle_slowfast.py.

State-dict names follow the public upstream module tree so
the reference's checkpoints (`sf_mdl.*`) address the same tensors.
"""
from types import SimpleNamespace

import torch
from torch import nn

# ---- published structure constants of the upstream network ------------------
STAGE_DEPTH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), "tiny": (1, 0, 0, 1), "mini": (1, 1, 1, 1)}

# temporal kernel of conv1 / res2 .. res5, per pathway
TEMPORAL_KERNEL_BASIS = {
    "c2d": [[[1]], [[1]], [[1]], [[1]], [[1]]],
    "i3d": [[[5]], [[3]], [[3, 1]], [[3, 1]], [[1, 3]]],
    "slow": [[[1]], [[1]], [[1]], [[3]], [[3]]],
    "slowfast": [[[1], [5]], [[1], [3]], [[1], [3]], [[3], [3]], [[3], [3]]],
}
POOL1 = {
    "c2d": [[2, 1, 1]],
    "i3d": [[2, 1, 1]],
    "slow": [[1, 1, 1]],
    "slowfast": [[1, 1, 1], [1, 1, 1]],
}


def default_sf_cfg(arch="slowfast", depth=50, width=64, num_frames=32):
    """The keys of `cfg.sf_mdl` the trunk reads (SURVEY.md section 5 'Config')."""
    multi = arch == "slowfast"
    nb = STAGE_DEPTH[depth]
    return SimpleNamespace(
        MODEL=SimpleNamespace(
            ARCH=arch,
            MODEL_NAME="SlowFast" if multi else "ResNet",
            SINGLE_PATHWAY_ARCH=["c2d", "i3d", "slow"],
            MULTI_PATHWAY_ARCH=["slowfast"],
        ),
        DATA=SimpleNamespace(
            NUM_FRAMES=num_frames, INPUT_CHANNEL_NUM=[3, 3] if multi else [3]
        ),
        SLOWFAST=SimpleNamespace(
            ALPHA=4, BETA_INV=8, FUSION_CONV_CHANNEL_RATIO=2, FUSION_KERNEL_SZ=7
        ),
        RESNET=SimpleNamespace(
            DEPTH=depth,
            WIDTH_PER_GROUP=width,
            NUM_GROUPS=1,
            ZERO_INIT_FINAL_BN=True,
            NUM_BLOCK_TEMP_KERNEL=[[n, n] if multi else [n] for n in nb],
            SPATIAL_STRIDES=[[1, 1], [2, 2], [2, 2], [2, 2]]
            if multi
            else [[1], [2], [2], [2]],
        ),
        BN=SimpleNamespace(EPSILON=1e-5, MOMENTUM=0.1),
        NONLOCAL=SimpleNamespace(
            LOCATION=[[[], []] if multi else [[]] for _ in range(4)],
            GROUP=[[1, 1] if multi else [1] for _ in range(4)],
            INSTANTIATION="dot_product",
            POOL=[[[1, 2, 2], [1, 2, 2]] for _ in range(4)],
        ),
    )


def _bn(c, cfg):
    return nn.BatchNorm3d(c, eps=cfg.BN.EPSILON, momentum=cfg.BN.MOMENTUM)


class Stem(nn.Module):
    """conv -> bn -> relu -> maxpool[1,3,3]/[1,2,2]."""

    def __init__(self, cin, cout, kt, cfg):
        super().__init__()
        self.conv = nn.Conv3d(
            cin, cout, (kt, 7, 7), stride=(1, 2, 2), padding=(kt // 2, 3, 3), bias=False
        )
        self.bn = _bn(cout, cfg)
        self.relu = nn.ReLU(inplace=True)
        self.pool_layer = nn.MaxPool3d((1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))

    def forward(self, x):
        return self.pool_layer(self.relu(self.bn(self.conv(x))))


class VideoModelStem(nn.Module):
    def __init__(self, cins, couts, kts, cfg):
        super().__init__()
        self.num_pathways = len(cins)
        for p in range(self.num_pathways):
            self.add_module(f"pathway{p}_stem", Stem(cins[p], couts[p], kts[p], cfg))

    def forward(self, xs):
        return [getattr(self, f"pathway{p}_stem")(xs[p]) for p in range(len(xs))]


class FuseFastToSlow(nn.Module):
    """slow' = cat([slow, relu(bn(conv[k,1,1]/[alpha,1,1](fast)))], 1)."""

    def __init__(self, cfast, ratio, ksz, alpha, cfg):
        super().__init__()
        self.conv_f2s = nn.Conv3d(
            cfast,
            cfast * ratio,
            (ksz, 1, 1),
            stride=(alpha, 1, 1),
            padding=(ksz // 2, 0, 0),
            bias=False,
        )
        self.bn = _bn(cfast * ratio, cfg)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, xs):
        slow, fast = xs
        fuse = self.relu(self.bn(self.conv_f2s(fast)))
        return [torch.cat([slow, fuse], 1), fast]


class Bottleneck(nn.Module):
    """a: [Tk,1,1]  b: [1,3,3] (carries the stride)  c: [1,1,1]; BN after each."""

    def __init__(self, cin, cout, cinner, tk, stride, cfg):
        super().__init__()
        self.a = nn.Conv3d(cin, cinner, (tk, 1, 1), padding=(tk // 2, 0, 0), bias=False)
        self.a_bn = _bn(cinner, cfg)
        self.a_relu = nn.ReLU(inplace=True)
        self.b = nn.Conv3d(
            cinner,
            cinner,
            (1, 3, 3),
            stride=(1, stride, stride),
            padding=(0, 1, 1),
            bias=False,
        )
        self.b_bn = _bn(cinner, cfg)
        self.b_relu = nn.ReLU(inplace=True)
        self.c = nn.Conv3d(cinner, cout, 1, bias=False)
        self.c_bn = _bn(cout, cfg)
        self.c_bn.transform_final_bn = True

    def forward(self, x):
        x = self.a_relu(self.a_bn(self.a(x)))
        x = self.b_relu(self.b_bn(self.b(x)))
        return self.c_bn(self.c(x))


class ResBlock(nn.Module):
    def __init__(self, cin, cout, cinner, tk, stride, cfg):
        super().__init__()
        if cin != cout or stride != 1:
            self.branch1 = nn.Conv3d(cin, cout, 1, stride=(1, stride, stride), bias=False)
            self.branch1_bn = _bn(cout, cfg)
        self.branch2 = Bottleneck(cin, cout, cinner, tk, stride, cfg)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        if hasattr(self, "branch1"):
            x = self.branch1_bn(self.branch1(x)) + self.branch2(x)
        else:
            x = x + self.branch2(x)
        return self.relu(x)


class Nonlocal(nn.Module):
    """slowfast/models/nonlocal_helper.py `Nonlocal` (un-vendored submodule, restated): theta / phi / g
    1x1x1 convs (with bias), phi and g on the max-pooled input, softmax(theta.phi / sqrt(dim_inner)) or
    theta.phi / positions ("dot_product"), 1x1x1 output conv, BatchNorm (gamma zero-initialised), residual."""

    def __init__(self, dim, dim_inner, pool_size, instantiation, cfg):
        super().__init__()
        self.dim, self.dim_inner, self.instantiation = dim, dim_inner, instantiation
        self.use_pool = pool_size is not None and any(s > 1 for s in pool_size)
        self.conv_theta = nn.Conv3d(dim, dim_inner, 1)
        self.conv_phi = nn.Conv3d(dim, dim_inner, 1)
        self.conv_g = nn.Conv3d(dim, dim_inner, 1)
        self.conv_out = nn.Conv3d(dim_inner, dim, 1)
        self.bn = _bn(dim, cfg)
        self.bn.transform_final_bn = True
        if self.use_pool:
            self.pool = nn.MaxPool3d(kernel_size=pool_size, stride=pool_size, padding=[0, 0, 0])

    def forward(self, x):
        x_identity = x
        n, c, t, h, w = x.size()
        theta = self.conv_theta(x)
        if self.use_pool:
            x = self.pool(x)
        phi, g = self.conv_phi(x), self.conv_g(x)
        theta = theta.view(n, self.dim_inner, -1)
        phi = phi.view(n, self.dim_inner, -1)
        g = g.view(n, self.dim_inner, -1)
        theta_phi = torch.einsum("nct,ncp->ntp", theta, phi)
        if self.instantiation == "softmax":
            theta_phi = torch.softmax(theta_phi * (self.dim_inner ** -0.5), dim=2)
        elif self.instantiation == "dot_product":
            theta_phi = theta_phi / theta_phi.shape[2]
        else:
            raise NotImplementedError(self.instantiation)
        out = torch.einsum("ntg,ncg->nct", theta_phi, g).view(n, self.dim_inner, t, h, w)
        return x_identity + self.bn(self.conv_out(out))


class ResStage(nn.Module):
    def __init__(self, cins, couts, cinners, tks, strides, nblocks, nblk_tk, cfg, nl_inds=None, nl_pool=None):
        super().__init__()
        self.num_pathways = len(cins)
        self.num_blocks = nblocks
        self.nl_inds = nl_inds if nl_inds is not None else [[] for _ in cins]
        for p in range(self.num_pathways):
            for i in self.nl_inds[p]:  # slowfast ResStage._construct: a Nonlocal after block i
                self.add_module(f"pathway{p}_nonlocal{i}",
                                Nonlocal(couts[p], couts[p] // 2, nl_pool[p], cfg.NONLOCAL.INSTANTIATION, cfg))
            n = nblocks[p]
            tk_list = (tks[p] * n)[: nblk_tk[p]] + [1] * (n - nblk_tk[p])
            for i in range(n):
                self.add_module(
                    f"pathway{p}_res{i}",
                    ResBlock(
                        cins[p] if i == 0 else couts[p],
                        couts[p],
                        cinners[p],
                        tk_list[i],
                        strides[p] if i == 0 else 1,
                        cfg,
                    ),
                )

    def forward(self, xs):
        out = []
        for p in range(self.num_pathways):
            x = xs[p]
            for i in range(self.num_blocks[p]):
                x = getattr(self, f"pathway{p}_res{i}")(x)
                if i in self.nl_inds[p]:
                    x = getattr(self, f"pathway{p}_nonlocal{i}")(x)
            out.append(x)
        return out


def init_weights(model, fc_init_std=0.01, zero_init_final_bn=True):
    """He-normal(fan_out) convs; BN 1/0, final BN of each bottleneck 0."""
    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2]
            m.weight.data.normal_(0.0, (2.0 / fan_out) ** 0.5)
        elif isinstance(m, nn.BatchNorm3d):
            final = getattr(m, "transform_final_bn", False) and zero_init_final_bn
            m.weight.data.fill_(0.0 if final else 1.0)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0.0, fc_init_std)
            if m.bias is not None:
                m.bias.data.zero_()


class VideoTrunk(nn.Module):
    """SlowFast (two pathways) or single-pathway ResNet (c2d / i3d / slow)."""

    def __init__(self, cfg):
        super().__init__()
        arch = cfg.MODEL.ARCH
        self.multi = arch in cfg.MODEL.MULTI_PATHWAY_ARCH
        self.num_pathways = 2 if self.multi else 1
        self.enable_detection = False
        w = cfg.RESNET.WIDTH_PER_GROUP
        inner = cfg.RESNET.NUM_GROUPS * w
        depths = STAGE_DEPTH[cfg.RESNET.DEPTH]
        tk = TEMPORAL_KERNEL_BASIS[arch]
        nbt = cfg.RESNET.NUM_BLOCK_TEMP_KERNEL
        ss = cfg.RESNET.SPATIAL_STRIDES
        if self.multi:
            binv = cfg.SLOWFAST.BETA_INV
            ratio = cfg.SLOWFAST.FUSION_CONV_CHANNEL_RATIO
            odr = binv // ratio
            fk, alpha = cfg.SLOWFAST.FUSION_KERNEL_SZ, cfg.SLOWFAST.ALPHA
            self.s1 = VideoModelStem(
                cfg.DATA.INPUT_CHANNEL_NUM, [w, w // binv], [tk[0][0][0], tk[0][1][0]], cfg
            )
            self.s1_fuse = FuseFastToSlow(w // binv, ratio, fk, alpha, cfg)
            cin_s, cin_f = w + w // odr, w // binv
            for k in range(4):
                mult = 4 * (2 ** k)
                cout_s, cout_f = w * mult, w * mult // binv
                stage = ResStage(
                    [cin_s, cin_f],
                    [cout_s, cout_f],
                    [inner * (2 ** k), inner * (2 ** k) // binv],
                    tk[k + 1],
                    ss[k],
                    [depths[k]] * 2,
                    nbt[k],
                    cfg,
                )
                setattr(self, f"s{k + 2}", stage)
                if k < 3:
                    setattr(
                        self, f"s{k + 2}_fuse", FuseFastToSlow(cout_f, ratio, fk, alpha, cfg)
                    )
                cin_s, cin_f = cout_s + cout_s // odr, cout_f
            self.dim_out = [w * 32, w * 32 // binv]
        else:
            self.s1 = VideoModelStem(cfg.DATA.INPUT_CHANNEL_NUM, [w], [tk[0][0][0]], cfg)
            cin = w
            for k in range(4):
                cout = w * 4 * (2 ** k) if depths[k] > 0 else cin
                stage = ResStage(
                    [cin], [cout], [inner * (2 ** k)], tk[k + 1], ss[k], [depths[k]], nbt[k], cfg,
                    nl_inds=cfg.NONLOCAL.LOCATION[k], nl_pool=cfg.NONLOCAL.POOL[k],
                )
                setattr(self, f"s{k + 2}", stage)
                cin = cout
            self.dim_out = [cin]
        for p in range(self.num_pathways):
            ps = POOL1[arch][p]
            self.add_module(f"pathway{p}_pool", nn.MaxPool3d(ps, stride=ps, padding=0))
        init_weights(self, 0.01, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def forward_features(self, x):
        # order of calls: vidsitu_code/mdl_sf_base.py:21-34 (multi) / :46-55 (single)
        x = self.s1(x)
        if self.multi:
            x = self.s1_fuse(x)
        x = self.s2(x)
        if self.multi:
            x = self.s2_fuse(x)
        x = [getattr(self, f"pathway{p}_pool")(x[p]) for p in range(self.num_pathways)]
        x = self.s3(x)
        if self.multi:
            x = self.s3_fuse(x)
        x = self.s4(x)
        if self.multi:
            x = self.s4_fuse(x)
        return self.s5(x)


class TrimmedHead(nn.Module):
    """AdaptiveAvgPool3d(1) per pathway, cat on channels (mdl_sf_base.py:65-113)."""

    def forward(self, feats):
        return torch.cat([f.mean(dim=(2, 3, 4), keepdim=True) for f in feats], 1)


class SFBaseRef(nn.Module):
    """trunk -> head -> permute -> Linear/ReLU/Linear (mdl_sf_base.py:116-216)."""

    def __init__(self, cfg, n_vocab):
        super().__init__()
        self.sf_mdl = VideoTrunk(cfg)
        self.head = TrimmedHead()
        din = sum(self.sf_mdl.dim_out)
        self.proj_head = nn.Sequential(
            nn.Linear(din, din // 2), nn.ReLU(), nn.Linear(din // 2, n_vocab)
        )

    def forward_encoder(self, feats):
        return self.sf_mdl.forward_features(list(feats))

    def forward_feats(self, feats):
        """feat_extractor.py:94-102 -> [N, din]."""
        h = self.head(self.forward_encoder(feats)).permute(0, 2, 3, 4, 1)
        return h.reshape(h.shape[0], -1)

    def forward(self, feats):
        h = self.head(self.forward_encoder(feats)).permute(0, 2, 3, 4, 1)
        return self.proj_head(h).reshape(h.shape[0], -1)


def slow_index(t, alpha):
    """video_utils.py:59-65 -- linspace(0, T-1, T//alpha).long()."""
    return torch.linspace(0, t - 1, t // alpha).long()


def randomize_bn(model, seed=0):
    """Non-trivial BN affine + running stats so folding / zero-init bugs show."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm3d):
            c = m.num_features
            m.weight.data.copy_(0.5 + torch.rand(c, generator=g))
            m.bias.data.copy_(0.2 * torch.randn(c, generator=g))
            m.running_mean.copy_(0.1 * torch.randn(c, generator=g))
            m.running_var.copy_(0.5 + torch.rand(c, generator=g))


def count_conv_macs_params(model, inputs):
    """Analytic MAC / parameter count of every Conv3d on one forward."""
    macs, params, hooks = [0], [0], []

    def hook(m, inp, out):
        k = m.kernel_size
        macs[0] += out.numel() * m.in_channels * k[0] * k[1] * k[2]
        params[0] += m.weight.numel()

    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        model.forward_features(list(inputs))
    for h in hooks:
        h.remove()
    return macs[0], params[0]


# ---------------------------------------------------------------------------------
# bf16-storage emulation.  The HIP path stores every activation (and activation
# gradient) in bf16 between kernels while accumulating in fp32.  A random-weight
# ResNet in TRAIN mode (batch statistics, N = 2) amplifies a 2^-9 perturbation by
# ~100x over 50 layers: the fp32 oracle itself moves by 24 % when only its weights and
# inputs are rounded to bf16, 38-40 % with activations rounded too (measured,
# DESIGN.md "tolerance policy").  So train-mode parity is asserted against this same
# fp32 arithmetic with the rounding applied at the points where the kernels store bf16.
# ---------------------------------------------------------------------------------
def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def emulate_bf16_storage(model, grads=True):
    """Round conv weights to bf16 in place and install hooks that round, to bf16, the
    tensors the HIP trunk materialises: every conv output, every BN output that is not
    the final BN of a bottleneck, every ResBlock output; with `grads`, also the gradient
    arriving at those tensors.  Returns the hook handles."""
    handles = []

    def fwd_round(mod, inp, out):
        out = _rb(out)
        if grads and out.requires_grad:
            out.register_hook(_rb)
        return out

    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            m.weight.data = _rb(m.weight.data)
            handles.append(m.register_forward_hook(fwd_round))
        elif isinstance(m, nn.BatchNorm3d) and not getattr(m, "transform_final_bn", False):
            handles.append(m.register_forward_hook(fwd_round))
        elif isinstance(m, ResBlock):
            handles.append(m.register_forward_hook(fwd_round))
    return handles
