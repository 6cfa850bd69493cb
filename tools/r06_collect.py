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
         "r06fin_attn_win.txt": "r06_final_attn_win.txt", "r06fin_clock.json": "r06_clock.json"}
for src, dst in files.items():
    shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
for src, dst in (("r06fin_traffic.json", "gemm_traffic.json"), ("r06fin_mfma_util.json", "mfma_util.json"),
                 ("r06fin_raster_traffic.json", "raster_traffic.json")):
    shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
b = json.loads(open(os.path.join(G, "r06fin_bench.json")).read().strip().splitlines()[-1])
print("bench:", b["value"], b["unit"], b["ms_per_step"], "ms; config5", b["config5_vitl_1gpu"]["value"], "; with_tokenizer",
      b["with_tokenizer"]["value"], "; entrypoint", b["entrypoint"]["value"])
