import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total ms/step", round(tot / steps / 1e6, 3))
keys = ['bn_bwd_reduce', 'bn_bwd_apply', 'bn_bwd_finalize', 'bn_apply', 'bn_finalize', 'bn_partials', 'conv_igemm', 'conv_direct',
        'conv_wgrad', 'wgrad_reduce', 'stem_wgrad', 'stem_conv', 'stem_slab', 'linear_fwd', 'linear_bwd', 'transpose_f32',
        'layernorm', 'attn', 'adam', 'cast_f32', 'weight_transpose', 'pack_input', 'maxpool', 'avgpool', 'copyBuffer', 'softmax',
        'at::native']
groups = {}
for r in rows:
    n = r['Name']
    k = next((k for k in keys if k in n), n[:40])
    groups[k] = groups.get(k, 0) + float(r['TotalDurationNs']) / steps / 1e6
for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
    print(f"{v:7.3f} ms  {k}")
