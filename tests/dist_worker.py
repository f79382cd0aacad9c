"""world_size-2 gloo rehearsal of the sharded path's only collective (detector.gather_detections)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h3d_amd  # noqa: E402,F401
from h3d_amd import detector  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = detector.shard_batch(8, rank, world)
full = torch.arange(8 * 100 * 40, dtype=torch.float32).view(8, 100, 40)
mine = full[lo:hi].clone()
out = detector.gather_detections(mine)
assert out.shape == (8, 100, 40) and torch.equal(out, full), "all-gather mismatch"
dist.barrier()
if rank == 0:
    print("DIST_OK")
dist.destroy_process_group()
