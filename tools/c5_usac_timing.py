"""C5 with USAC on one GPU (bench_extras.c5_usac standing alone): 512 pairs through mlpl_pair_pose_batch_usac_dev."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import matchinglib_poselib_amd as mpa
import bench_extras
ctx = mpa.Context(0)
print(json.dumps(bench_extras.c5_usac(ctx, torch.device("cuda:0"), cpu_baseline=True, total=int(sys.argv[1]) if len(sys.argv) > 1 else 512), indent=1))
