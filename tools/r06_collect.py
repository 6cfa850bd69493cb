"""Copy the outputs of ONE tools/r06_final.sh call (gpurun_out/r06fin_*) into profiles/r06_final_* and refresh the
un-prefixed summaries bench.py reads (gemm_traffic.json, mfma_util.json, raster_traffic.json, r06_clock.json).
    python tools/r06_collect.py"""
import json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
files = {"r06fin_tests.txt": "r06_final_gpu_tests.txt", "r06fin_bench.json": "r06_final_bench.json",
         "r06fin_seq_kernel_stats.csv": "r06_final_seq_kernel_stats.csv", "r06fin_seq_step_kernels.txt": "r06_final_seq_step_kernels.txt",
         "r06fin_two_kernel_stats.csv": "r06_final_two_kernel_stats.csv", "r06fin_gaps.txt": "r06_final_gaps.txt",
         "r06fin_mfma_util.json": "r06_final_mfma_util.json", "r06fin_traffic.json": "r06_final_traffic.json",
         "r06fin_vitl_kernel_stats.csv": "r06_final_vitl_kernel_stats.csv", "r06fin_vitl_mfma_util.json": "r06_final_vitl_mfma_util.json",
         "r06fin_vitl_traffic.json": "r06_final_vitl_traffic.json", "r06fin_raster_kernel_stats.csv": "r06_final_raster_kernel_stats.csv",
         "r06fin_raster_traffic.json": "r06_final_raster_traffic.json", "r06fin_attn16.txt": "r06_final_attn16.txt",
         "r06fin_attn_win.txt": "r06_final_attn_win.txt", "r06fin_clock.json": "r06_clock.json",
         "r06fin_conv_waves.txt": "r06_final_conv_waves.txt"}
for src, dst in files.items():
    shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
for src, dst in (("r06fin_traffic.json", "gemm_traffic.json"), ("r06fin_mfma_util.json", "mfma_util.json"),
                 ("r06fin_raster_traffic.json", "raster_traffic.json")):
    shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
b = json.loads(open(os.path.join(G, "r06fin_bench.json")).read().strip().splitlines()[-1])
print("bench:", b["value"], b["unit"], b["ms_per_step"], "ms; config5", b["config5_vitl_1gpu"]["value"], "; with_tokenizer",
      b["with_tokenizer"]["value"], "; entrypoint", b["entrypoint"]["value"])

# ---- the weight-gradient product split into its GEMM kernel and its reduction pass (bench.py: roofline.kernel_vs_reduction)
import csv, re
stats = {}
for r in csv.DictReader(open(os.path.join(P, "r06_final_seq_kernel_stats.csv"))):
    n = re.sub(r"^void ", "", r["Name"]); n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*$", "", n)
    stats[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
tot = lambda k: stats[k][0] * stats[k][1] if k in stats else 0.0
n_prod = stats.get("gemm_tn_p8_kernel<true>", (0, 0))[0] + 2 * stats.get("gemm_tn_p8_group_kernel", (0, 0))[0]
kern = (tot("gemm_tn_p8_kernel<true>") + tot("gemm_tn_p8_group_kernel")) / n_prod
red = (tot("tn_reduce_kernel") + tot("tn_reduce_group_kernel")) / n_prod
fl = b["roofline"]["algorithmic_flop_per_launch"]
json.dump({"kernel_us_per_product": round(kern, 2), "reduction_us_per_product": round(red, 2), "products": n_prod,
           "frac_kernel_alone": round(fl / (kern * 1e-6) / 2.5e15, 4), "frac_with_reduction": round(fl / ((kern + red) * 1e-6) / 2.5e15, 4),
           "source": "profiles/r06_final_seq_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the sequential step, same box and call as "
                     "r06_final_bench.json; algorithmic FLOPs per product from that bench line)"},
          open(os.path.join(P, "wgrad_split.json"), "w"), indent=1)
print("wgrad split:", json.load(open(os.path.join(P, "wgrad_split.json"))))
