"""GPU time of the Zero123 SDS step by kind of kernel, from the rocprofv3 kernel statistics tools/zero123_profile.sh wrote
(gpurun_out/zero123_<tag>/z_kernel_stats.csv): the table of profiles/r03_zero123.md.   usage: python tools/zero123_kinds.py r03g [steps]"""
import collections, csv, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03g"
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 21.0        # 8 warm-up + 10 steady state + 3 while capturing
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(os.path.join(root, "gpurun_out", f"zero123_{tag}", "z_kernel_stats.csv"))))
def kind(n):
    if "k_conv3x3" in n and "1, 1>" in n: return "linear layers, hand-written (`k_conv3x3<.., 1, 1>` = `dm4d_linear_f16`; incl. the 1x1 class of the stride-2 data gradient)"
    if "k_conv" in n: return "convolutions, hand-written (`k_conv3x3*`, `k_conv_s2_dgrad`, `k_conv_reduce`, `k_conv3x3_c128_small`)"
    if "Cijk" in n: return "GEMM (hipBLASLt; incl. the VAE attention's batched GEMMs)"
    if "groupnorm" in n: return "GroupNorm (+ SiLU + embedding / bias add), HIP `k_groupnorm*`"
    if "igemm" in n or "grouped_conv" in n or "naive_conv" in n or "SubTensorOp" in n: return "convolutions, library (conv_in / conv_out of both networks, the UNet's stride-2 Downsample pad 1 .. )"
    if "attn_fwd" in n or "attn_bwd" in n or "k_attention" in n: return "attention (UNet self-attention: HIP `k_attention`)"
    if "layernorm" in n or "layer_norm" in n: return "LayerNorm: HIP `k_add_layernorm_f16`"
    if "geglu" in n or "k_add_bias" in n: return "HIP `k_geglu`, `k_add_bias`"
    return "elementwise (residual adds, copies, casts, cat, softmax, rng ...)"
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = kind(r["Name"]); acc[k][0] += float(r["TotalDurationNs"]); acc[k][1] += int(r["Calls"])
tot = sum(v[0] for v in acc.values())
print(f"| kind | ms ({steps:.0f} steps) | ms per step | share | launches |\n|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda x: -x[1][0]):
    print(f"| {k} | {v[0] / 1e6:.1f} | {v[0] / 1e6 / steps:.2f} | {100 * v[0] / tot:.1f} % | {v[1]} |")
print(f"total {tot / 1e6:.1f} ms, {tot / 1e6 / steps:.2f} ms per step, {sum(v[1] for v in acc.values())} launches")
