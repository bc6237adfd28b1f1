"""C5 with USAC on one GPU (bench_extras.c5_usac standing alone): 512 pairs through mlpl_pair_pose_batch_usac_dev."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import matchinglib_poselib_amd as mpa
import bench_extras
ctx = mpa.Context(0)
if len(sys.argv) > 2:   # pairs per internal batch of the pair entries (option pair_batch_seq; default 512)
    ctx.set_option("pair_batch_seq", int(sys.argv[2]))
if len(sys.argv) > 3:   # runs per cohort (option hub_cohort; default 128)
    ctx.set_option("hub_cohort", int(sys.argv[3]))
if len(sys.argv) > 4:   # cohorts in flight (option hub_lanes; default 2)
    ctx.set_option("hub_lanes", int(sys.argv[4]))
print(json.dumps(bench_extras.c5_usac(ctx, torch.device("cuda:0"), cpu_baseline=True, total=int(sys.argv[1]) if len(sys.argv) > 1 else 512), indent=1))
