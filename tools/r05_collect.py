"""Copy the outputs of ONE tools/r05_final.sh call (gpurun_out/r05fin_*) into profiles/r05_final_* and refresh the
un-prefixed summaries bench.py reads (gemm_traffic.json, mfma_util.json, raster_traffic.json, r05_clock.json).
    python tools/r05_collect.py"""
import json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
files = {"r05fin_tests.txt": "r05_final_gpu_tests.txt", "r05fin_bench.json": "r05_final_bench.json",
         "r05fin_seq_kernel_stats.csv": "r05_final_seq_kernel_stats.csv", "r05fin_seq_step_kernels.txt": "r05_final_seq_step_kernels.txt",
         "r05fin_two_kernel_stats.csv": "r05_final_two_kernel_stats.csv", "r05fin_gaps.txt": "r05_final_gaps.txt",
         "r05fin_mfma_util.json": "r05_final_mfma_util.json", "r05fin_traffic.json": "r05_final_traffic.json",
         "r05fin_vitl_kernel_stats.csv": "r05_final_vitl_kernel_stats.csv", "r05fin_vitl_mfma_util.json": "r05_final_vitl_mfma_util.json",
         "r05fin_vitl_traffic.json": "r05_final_vitl_traffic.json", "r05fin_raster_kernel_stats.csv": "r05_final_raster_kernel_stats.csv",
         "r05fin_raster_traffic.json": "r05_final_raster_traffic.json", "r05fin_attn16.txt": "r05_final_attn16.txt",
         "r05fin_attn_win.txt": "r05_final_attn_win.txt", "r05fin_clock.json": "r05_clock.json"}
for src, dst in files.items():
    shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
for src, dst in (("r05fin_traffic.json", "gemm_traffic.json"), ("r05fin_mfma_util.json", "mfma_util.json"),
                 ("r05fin_raster_traffic.json", "raster_traffic.json")):
    shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
b = json.loads(open(os.path.join(G, "r05fin_bench.json")).read().strip().splitlines()[-1])
print("bench:", b["value"], b["unit"], b["ms_per_step"], "ms; config5", b["config5_vitl_1gpu"]["value"], "; with_tokenizer",
      b["with_tokenizer"]["value"], "; entrypoint", b["entrypoint"]["value"])
