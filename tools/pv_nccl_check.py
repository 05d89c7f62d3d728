"""tools/pv_nccl_check.py — shard.pv_pitch_shift_rank over the RCCL backend (run under torch.distributed.run)."""
import os, sys
import numpy as np
import torch, torch.distributed as dist
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import melonix_amd as mx
from melonix_amd import shard as sh
from conftest import accum_sweep, SR
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
w = accum_sweep(4 * SR)
ctx = mx.Context(local)
a = ctx.upload(w)
lo, hi, f32, i16 = sh.pv_pitch_shift_rank(ctx, a, 3.0, dist, rank, world)
whole, _ = ctx.pv_pitch_shift(a, 3.0)
ok = np.array_equal(f32.view(np.uint32), whole[lo:hi].view(np.uint32))
print(f"rank {rank}/{world}: outputs [{lo},{hi}) equal to the single-call slice: {ok}")
dist.destroy_process_group()
sys.exit(0 if ok else 1)
