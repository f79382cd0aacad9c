"""gloo rehearsal (CPU, world_size >= 2) of the sharded path's only collective, detector.gather_detections:
even shards (n = 8), uneven shards (n = 10 over 4 ranks / n = 9 over 2: the lowest ranks hold one image more)
and the error when a rank's shard does not match shard_batch."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h3d_amd  # noqa: E402,F401
from h3d_amd import detector  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
for n in (8, 9, 10, world, world + 1, 3 * world - 1):
    lo, hi = detector.shard_batch(n, rank, world)
    full = torch.arange(n * 100 * 40, dtype=torch.float32).view(n, 100, 40)
    mine = full[lo:hi].clone()
    out = detector.gather_detections(mine, n_images=n)
    assert out.shape == (n, 100, 40) and torch.equal(out, full), "all-gather mismatch (n = %d, world = %d)" % (n, world)
    if n % world == 0:
        out = detector.gather_detections(mine)           # equal shards: n_images may be omitted
        assert torch.equal(out, full)
    dist.barrier()
try:
    detector.gather_detections(torch.zeros(5, 100, 40), n_images=2 * world + 1)   # no rank holds 5 of 2w+1 images (w >= 2)
    raise SystemExit("expected a RuntimeError for a shard that shard_batch does not produce")
except RuntimeError as e:
    assert "shard_batch" in str(e)
dist.barrier()
if rank == 0:
    print("DIST_OK")
dist.destroy_process_group()
