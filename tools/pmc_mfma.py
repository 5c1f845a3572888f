"""Aggregate the SQ / GRBM counter passes of tools/pmc_mfma.sh into profiles-style JSON: per (layer, kind)
the matrix-core busy fraction and the wave-cycle breakdown.
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles =
                   GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md 'DVFS give-back')
  wait_any       = SQ_WAIT_ANY / SQ_WAVE_CYCLES        (waves parked in s_waitcnt / barriers)
  wait_inst_any  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (issue stalls)
  wait_inst_lds  = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES   (LDS-issue stalls, a part of wait_inst_any)
  active         = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
usage: python tools/pmc_mfma.py <out_dir> <build tag>"""
import collections, csv, glob, json, sys

out_dir, build = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
plan = json.load(open(f"{out_dir}/plan.json"))
KEEP = ("conv_igemm_kernel", "conv_wgrad_ring_kernel", "conv_wgrad_kernel", "conv_direct_kernel", "conv_halo_kernel",
        "conv_pw_kernel", "conv_deep_kernel", "conv_wgrad_deep_kernel")


def rows_of(sub):
    by_dispatch = collections.OrderedDict()
    files = sorted(glob.glob(f"{out_dir}/{sub}/**/*counter_collection.csv", recursive=True))
    for f in files:
        for r in csv.DictReader(open(f)):
            if not any(k in r["Kernel_Name"] for k in KEEP):
                continue
            d = by_dispatch.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"].split("(")[0].replace("void ", ""),
                                                              "grid": r.get("Grid_Size"), "lds": r.get("LDS_Block_Size"),
                                                              "vgpr": r.get("VGPR_Count")})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [by_dispatch[k] for k in sorted(by_dispatch)]


passes = {sub: rows_of(sub) for sub in ("sq", "sq2")}
result = []
for sub, rows in passes.items():
    want = sum(p["launches"] for p in plan)
    if len(rows) != want:
        print(f"pass {sub}: {len(rows)} conv dispatches, plan has {want} -- skipped", file=sys.stderr)
        passes[sub] = None
i = 0
for p in plan:
    n = p["launches"]
    ent = {"layer": p["layer"], "kind": p["kind"]}
    for sub in ("sq", "sq2"):
        rows = passes.get(sub)
        if rows is None:
            continue
        seg = rows[i:i + n]
        ent["kernel"] = seg[0]["kernel"]
        ent["grid"], ent["lds_bytes"], ent["vgpr"] = seg[0]["grid"], seg[0]["lds"], seg[0]["vgpr"]
        for c in seg[0]:
            if c.isupper() or c.startswith(("SQ_", "GRBM_")):
                ent[c] = sum(r.get(c, 0.0) for r in seg) / n
    i += n
    if "GRBM_GUI_ACTIVE" in ent and ent["GRBM_GUI_ACTIVE"] > 0:
        cyc = ent["GRBM_GUI_ACTIVE"] / 8.0
        ent["kernel_cycles"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in ent:
            ent["mfma_busy"] = round(ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc), 4)
        ent["tflops_at_2p4ghz_equiv"] = round(p["flops"] / cyc * 2.4e9 / 1e12, 1)
    wc = ent.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for a, b in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"),
                     ("wait_inst_lds", "SQ_WAIT_INST_LDS"), ("active", "SQ_ACTIVE_INST_ANY")):
            if b in ent:
                ent[a] = round(ent[b] / wc, 4)
    if ent.get("SQ_LDS_IDX_ACTIVE"):
        ent["lds_conflict"] = round(ent.get("SQ_LDS_BANK_CONFLICT", 0.0) / ent["SQ_LDS_IDX_ACTIVE"], 4)
    result.append(ent)
meta = {"_meta": {"build": build, "method": "rocprofv3 --kernel-trace --pmc, two SQ passes + GRBM_GUI_ACTIVE; each (layer, kind) "
                  "launched alone, eager, 3x; counters averaged over the launches; see tools/pmc_mfma.sh",
                  "units": "mfma_busy: fraction of SIMD-cycles with the matrix pipe busy; wait_* / active: fraction of wave-cycles"},
        "layers": result}
json.dump(meta, open(f"{out_dir}/pmc_mfma.json", "w"), indent=1)
for e in result:
    print(f"{e['layer']:26s} {e['kind']:6s} mfma_busy {e.get('mfma_busy', float('nan')):6.3f} wait_any {e.get('wait_any', float('nan')):6.3f} "
          f"wait_inst {e.get('wait_inst_any', float('nan')):6.3f} lds {e.get('wait_inst_lds', float('nan')):6.3f} active {e.get('active', float('nan')):6.3f} "
          f"ldsconf {e.get('lds_conflict', float('nan')):5.2f} us {e.get('kernel_cycles', 0) / 2100.0:6.1f}  {e.get('kernel', '')[:58]}")
