"""Write profiles/r06_summary.md from the collected evidence set (tools/r06_collect.py first)."""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
b = json.loads(open(os.path.join(P, "r06_final_bench.json")).read().strip().splitlines()[-1])
r = b["roofline"]; g = r["gemm_family"]; w = b["with_tokenizer"]; f = w["fp16x2_mode"]; e = b["entrypoint"]; c5 = b["config5_vitl_1gpu"]
c4 = b["config4_end_to_end"]; ra = b["rasterizer_1m_events"]["roofline"]; att = r.get("attention_family", {})
tests = open(os.path.join(P, "r06_final_gpu_tests.txt")).read().strip().splitlines()[-1]
step = open(os.path.join(P, "r06_final_seq_step_kernels.txt")).read().splitlines()
def stat(path, pat):
    for row in csv.DictReader(open(os.path.join(P, path))):
        if pat in row["Name"]:
            return int(row["Calls"]), float(row["AverageNs"]) / 1e3
    return 0, 0.0
u = json.load(open(os.path.join(P, "r06_final_mfma_util.json"))); t = json.load(open(os.path.join(P, "r06_final_traffic.json")))
vu = json.load(open(os.path.join(P, "r06_final_vitl_mfma_util.json"))); vt = json.load(open(os.path.join(P, "r06_final_vitl_traffic.json")))
rt = json.load(open(os.path.join(P, "r06_final_raster_traffic.json")))
clk = json.load(open(os.path.join(P, "r06_clock.json")))["kernels"]
mu = lambda d, k: next((v["mfma_util"] for kk, v in d.items() if kk.startswith(k)), None)
mb = lambda d, k: next((v["hbm_bytes_per_launch"] / 1e6 for kk, v in d.items() if kk.startswith(k)), None)
rk = stat("r06_final_raster_kernel_stats.csv", "raster_bin_keys"); rc = stat("r06_final_raster_kernel_stats.csv", "raster_bin_accum")
ralg = 64 * (32 * 1_000_000 + 3 * 480 * 640)
L = []
L.append("# Round 6 -- 1x MI355X, ViT-B/16 C=2, B=256, bf16\n")
L.append("Every `r06_final_*` file (and `gemm_traffic.json`, `mfma_util.json`, `raster_traffic.json`, `r06_clock.json`) comes from ONE call of "
         "`tools/r06_final.sh` on one box after the last kernel commit, copied by `tools/r06_collect.py`; this text is generated from them by "
         "`tools/r06_summary.py`; `tests/test_bench_profiles.py::test_round6_evidence_set_is_consistent` checks that the files agree.\n")
L.append("## End of the round\n")
L.append(f"* `python -m pytest tests -m gpu` -> `r06_final_gpu_tests.txt`: {tests}.")
L.append(f"* `python bench.py` -> `r06_final_bench.json`: **{b['ms_per_step']} ms/step = {b['value']} samples/s** (p50 {b.get('ms_per_step_p50')}); "
         f"`roofline` (dominant kernel: the weight-gradient GEMM, `gemm_tn_p8_kernel` / `gemm_tn_p8_group_kernel` + reduction pass, HIP events): {r['avg_launch_us']} us per product = {r['achieved']} TFLOP/s = "
         f"**{r['frac']}** of 2.5 PFLOP/s, traffic {r['traffic'] / 1e6:.1f} MB per product; `gemm_family` {g['achieved']} TFLOP/s = **{g['frac']}** "
         f"(bias {g['per_epilogue']['0']['tflops']}, residual {g['per_epilogue']['2']['tflops']}, GELU + stored derivative {g['per_epilogue']['6']['tflops']}, "
         f"derivative product {g['per_epilogue']['7']['tflops']}, single weight-gradient launches (head, patch embedding) {g['per_epilogue']['100']['tflops']}, grouped pairs (proj + qkv, fc2 + fc1) {g['per_epilogue'].get('102', {}).get('tflops')}); `whole_step` {r['whole_step']['frac']} on executed "
         f"FLOPs, {r['whole_step']['frac_on_reference_flop_count']} on the reference's count; attention 14x14 in the step: forward {att.get('forward', {}).get('avg_us')} us, "
         f"backward {att.get('backward', {}).get('avg_us')} us per layer.")
L.append(f"* `with_tokenizer` (certified fp16x2 tokenizer, {f['label_mismatches']} label mismatches against the fp32 mode on {f['tokens_compared']} tokens; "
         f"{f['certification']['flagged_samples_in_label_sets']} of {f['certification']['samples_in_label_sets']} samples of the label sets recomputed in fp32): "
         f"**{w['value']} samples/s** ({w['ms_per_step']} ms; tokenizer {f['tokenizer_ms_per_step']} ms, raw fp16x2 forward {f['raw_fp16x2_tokenizer_ms']} ms; fp32 mode {w['fp32_mode']['value']} samples/s); "
         f"`entrypoint` **{e['value']} samples/s** ({e['ms_per_step']} ms; tokenizer stage {e['stages_alone_ms'].get('tokenizer_fp16x2')} ms, "
         f"{e['tokenizer_certification']['flagged_samples_per_batch']} samples per batch recomputed); configs[3] end to end {c4['value']} samples/s "
         f"(rasterizer {c4['rasterizer_ms_per_step']} ms of the step).")
L.append(f"* **`config5_vitl_1gpu`** (ViT-L/16 480x640, 1201 tokens, B = 64): **{c5['value']} samples/s** ({c5['ms_per_step']} ms), {c5['model_flops_frac_of_peak']} of peak on "
         f"executed FLOPs, {c5['model_flops_frac_of_peak_reference_count']} on the reference count; one-stream family split: "
         + ", ".join(f"{k} {v['ms']} ms ({v['tflops']} TFLOP/s)" for k, v in c5["family_split_one_stream_step"].items()) + ".")
sol = b["speed_of_light"]; lib = b["library"]
# the split of THIS call's trace (tools/r06_collect.py -> wgrad_split.json); the copy inside the bench line was read from the file
# that was committed when bench.py ran, i.e. from the previous call
if os.path.exists(os.path.join(P, "wgrad_split.json")):
    r = dict(r); r["kernel_vs_reduction"] = json.load(open(os.path.join(P, "wgrad_split.json")))
L.append(f"* `roofline.frac` per instrumented step: {r['frac_per_instrumented_step']} (pooled {r['frac']}); `kernel_vs_reduction` (rocprofv3, same call): "
         f"{r['kernel_vs_reduction']['kernel_us_per_product']} + {r['kernel_vs_reduction']['reduction_us_per_product']} us per product = {r['kernel_vs_reduction']['frac_kernel_alone']} kernel alone, "
         f"{r['kernel_vs_reduction']['frac_with_reduction']} with its reduction pass; `library`: {lib['path']}, build flags '{lib['build_flags']}', shipped build {lib['shipped_build']}, ABI {lib['abi']}."
         if r.get("kernel_vs_reduction") else f"* `roofline.frac` per instrumented step: {r['frac_per_instrumented_step']} (pooled {r['frac']}); `library`: {lib}.")
L.append("* `speed_of_light` (per launch max(FLOPs / at-clock MFMA peak " + str(sol["mfma_peak_at_clock_tflops"]) + " TFLOP/s, bytes / 6.3 TB/s), no overlap): "
         + "; ".join(f"{k} {v['floor_ms']} ms" for k, v in sol["families"].items())
         + f" -> **sum {sol['sum_floor_ms']} ms** (all-MFMA {sol['sum_mfma_ms']}, all-HBM {sol['sum_hbm_ms']}); step {sol['ms_per_step']} ms = **{sol['ms_per_step_over_sum_floor']} x** the floor.")
L.append(f"* `cpu_baseline`: {b['cpu_baseline'].get('value')} samples/s on {b['cpu_baseline'].get('cores')} CPUs.")
L.append(f"* sequential trace (`--no-side-stream`) -> `r06_final_seq_kernel_stats.csv`, per-step table from the same trace -> `r06_final_seq_step_kernels.txt`: {step[0]}; top rows:")
for l in step[1:13]:
    L.append("      " + re.sub(r"\s+", " ", l.strip()))
L.append(f"* counters of the ViT-B step -> `r06_final_mfma_util.json` (= `mfma_util.json`): `gemm_tn_p8` {mu(u, 'gemm_tn_p8_kernel')}, `gemm_p8_pair<0>` {mu(u, 'gemm_p8_pair_kernel<0,false')}, "
         f"`gemm_p8<0,256>` {mu(u, 'gemm_p8_kernel<0,256')}, attention forward {mu(u, 'attn16_fwd')}, backward {mu(u, 'attn16_bwd')}, whole step {u['_whole_step']['mfma_util']}; "
         f"`r06_final_traffic.json` (= `gemm_traffic.json`): `gemm_tn_p8` {mb(t, 'gemm_tn_p8_kernel'):.0f} MB per launch, `gemm_p8_pair` {mb(t, 'gemm_p8_pair_kernel'):.0f}, `gemm_p8` {mb(t, 'gemm_p8_kernel'):.0f}, "
         f"`attn16_bwd` {mb(t, 'attn16_bwd'):.0f}, `attn16_fwd` {mb(t, 'attn16_fwd'):.0f}, `ln_bwd_branch` {mb(t, 'ln_bwd_branch'):.0f}, AdamW {mb(t, 'adamw'):.0f}.")
# (the dS-storing backward -- the engine's default since the end of round 6 -- runs attn_bwd_kvs_win_kernel / attn_bwd_qs_win_kernel)
KV = "attn_bwd_kvs_win_kernel" if stat("r06_final_vitl_kernel_stats.csv", "attn_bwd_kvs_win_kernel") else "attn_bwd_kv_win_kernel"
DQ = "attn_bwd_qs_win_kernel" if stat("r06_final_vitl_kernel_stats.csv", "attn_bwd_qs_win_kernel") else "attn_bwd_q_win_kernel"
a1 = stat("r06_final_vitl_kernel_stats.csv", "attn_fwd_win_kernel"); a2 = stat("r06_final_vitl_kernel_stats.csv", KV); a3 = stat("r06_final_vitl_kernel_stats.csv", DQ)
L.append(f"* **config #5 counters (none existed before this round)**: `r06_final_vitl_kernel_stats.csv` (two-stream step under rocprofv3: forward {a1[1]:.0f} us, dK/dV {a2[1]:.0f}, "
         f"dQ {a3[1]:.0f} per layer); `r06_final_vitl_mfma_util.json`: `attn_fwd_win` {mu(vu, 'attn_fwd_win_kernel')}, `{KV[:-7]}` {mu(vu, KV)}, "
         f"`{DQ[:-7]}` {mu(vu, DQ)}, `gemm_tn_p8` {mu(vu, 'gemm_tn_p8_kernel')}, whole step {vu['_whole_step']['mfma_util']}; `r06_final_vitl_traffic.json`: forward "
         f"{mb(vt, 'attn_fwd_win_kernel'):.0f} MB per launch (Q, K, V read once + output written: 630 MB at B = 64; 1 852 MB before all groups of a (head, sample) pair were put on one XCD), dK/dV {mb(vt, KV):.0f}, dQ {mb(vt, DQ):.0f} (algorithmic, recomputing form: 944 / 787; the dS-storing form adds 2 x 3.0 GB of dS per layer at B = 64: written by dK/dV, read by dQ), `gemm_tn_p8` {mb(vt, 'gemm_tn_p8_kernel'):.0f}.")
L.append(f"* **config #4 rasterizer** (`tools/raster_bench.py`, 64 x 1 M events, 480x640): `r06_final_raster_kernel_stats.csv`: `raster_bin_keys` {rk[1]:.1f} us + `raster_bin_accum` {rc[1]:.1f} us "
         f"= {rk[1] + rc[1]:.1f} us for {ralg / 1e6:.0f} MB algorithmic = {ralg / ((rk[1] + rc[1]) * 1e-6) / 1e12:.2f} TB/s = **{ralg / ((rk[1] + rc[1]) * 1e-6) / 8e12:.3f}** of 8 TB/s; "
         f"`r06_final_raster_traffic.json` (= `raster_traffic.json`): {(rt['raster_bin_keys']['hbm_bytes_per_launch'] + rt['raster_bin_accum']['hbm_bytes_per_launch']) / 1e6:.0f} MB per launch = "
         f"{(rt['raster_bin_keys']['hbm_bytes_per_launch'] + rt['raster_bin_accum']['hbm_bytes_per_launch']) / ralg:.2f} x algorithmic; in the bench line (32 x 1 M): {ra['frac']} of 8 TB/s.")
L.append("* `r06_final_attn16.txt` / `r06_final_attn_win.txt` (kernels alone): " + " | ".join(x.strip() for x in open(os.path.join(P, "r06_final_attn16.txt")).read().strip().splitlines()[-1:])
         + " || window attention B = 64 x 16 heads x 1201 tokens: " + " | ".join(x.strip() for x in open(os.path.join(P, "r06_final_attn_win.txt")).read().strip().splitlines()[1::2]) + ".")
L.append("* `r06_clock.json` (stamp build): " + "; ".join(f"{k}: {v['clock_ghz']} GHz" for k, v in clk.items()) + ".\n")
L.append(open(os.path.join(P, "r06_during.md")).read() if os.path.exists(os.path.join(P, "r06_during.md")) else "")
open(os.path.join(P, "r06_summary.md"), "w").write("\n".join(L))
print("\n".join(L)[:3000])
