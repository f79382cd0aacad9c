import sys, json, torch
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.boundary_op_times(16, torch.device("cuda:0")), indent=0))
