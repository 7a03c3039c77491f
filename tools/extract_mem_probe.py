import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch
from bench_extract import measure_list
r = measure_list("resnet101", workers=8, short=12, mid=40, long=64)
print("descriptors/s", r["value"], "max reserved GB", round(torch.cuda.max_memory_reserved() / 1e9, 1), "max allocated GB", round(torch.cuda.max_memory_allocated() / 1e9, 1))
