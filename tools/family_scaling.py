"""Kernel families of two rocprofv3 kernel_stats tables (tools/family_scaling.sh): ms per step and us per clip at each
batch size, and the ratio of the per-clip costs (1.0 = scales with the batch; 4.0 at 8 vs 32 clips = pure fixed cost)."""
import csv, re, sys

STEPS = 7  # bench.py --steps 5 --warmup 2 (eager: every step is profiled)
FAM = [("conv fwd/dgrad (igemm)", r"conv_igemm_kernel|conv_igemm_aol"), ("conv halo", r"conv_halo"), ("conv pw", r"conv_pw_kernel"),
       ("conv deep", r"conv_deep_kernel"), ("conv direct", r"conv_direct"), ("conv pair (dgrad+wgrad)", r"conv_pair_kernel"),
       ("wgrad deep", r"conv_wgrad_deep"), ("wgrad ring/tile", r"conv_wgrad_kernel|conv_wgrad_ring"),
       ("wgrad reduce / bn-bwd finalize", r"wgrad_reduce|bn_bwd_finalize|bn_finalize2"),
       ("bn finalize fwd", r"^bn_finalize_kernel|bn_partials_reduce"), ("bn apply", r"bn_apply_cols|bn_apply_kernel|bn_apply_maxpool"),
       ("bn bwd reduce", r"bn_bwd_reduce"), ("bn bwd apply", r"bn_bwd_apply"), ("stems", r"stem_"), ("pools / pack", r"pool|pack_input"),
       ("adam", r"adam"), ("weight images", r"weight_transpose|cast"), ("encoder / head", r"linear|layernorm|attn|softmax|xent"),
       ("torch (in the step)", r"at::|elementwise")]
# One-time work that is NOT part of a step (ADVICE / VERDICT r5): the ~1 000 `__amd_rocclr_copyBuffer` parameter uploads of model
# construction and the fills of first-use workspace allocations happen once per process -- reported on a line of their own,
# whole (not divided by the step count), and left out of the per-step sums.
ONE_TIME = r"rocclr|FillFunctor"


def load(path):
    fam = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"]
        key = "one-time (whole run)" if re.search(ONE_TIME, name) else next((f for f, pat in FAM if re.search(pat, name)), "other")
        c, t = fam.get(key, (0, 0.0))
        fam[key] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
    return fam


a, ca, b, cb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
print(f"{'family':34s} {'launches':>8s} {'ms/step@' + str(ca):>12s} {'ms/step@' + str(cb):>12s} {'us/clip@' + str(ca):>12s} {'us/clip@' + str(cb):>12s} {'ratio':>6s} {'excess ms@' + str(ca):>12s}")
tot = [0.0, 0.0, 0.0]
for key in ("one-time (whole run)",):
    if key in a or key in b:
        na, ta = a.get(key, (0, 0.0)); nb, tb = b.get(key, (0, 0.0))
        print(f"# {key}: {na} launches {ta / 1e6:.3f} ms at {ca} clips, {nb} launches {tb / 1e6:.3f} ms at {cb} clips -- not in the table below")
for key in [f for f, _ in FAM] + ["other"]:
    if key not in a and key not in b:
        continue
    na, ta = a.get(key, (0, 0.0)); nb, tb = b.get(key, (0, 0.0))
    ma, mb = ta / STEPS / 1e6, tb / STEPS / 1e6
    pa, pb = ma * 1e3 / ca, mb * 1e3 / cb
    ex = ma - mb * ca / cb
    tot[0] += ma; tot[1] += mb; tot[2] += ex
    print(f"{key:34s} {na // STEPS:8d} {ma:12.3f} {mb:12.3f} {pa:12.1f} {pb:12.1f} {pa / pb if pb else 0:6.2f} {ex:12.3f}")
print(f"{'sum':34s} {'':8s} {tot[0]:12.3f} {tot[1]:12.3f} {tot[0] * 1e3 / ca:12.1f} {tot[1] * 1e3 / cb:12.1f} {'':6s} {tot[2]:12.3f}")
