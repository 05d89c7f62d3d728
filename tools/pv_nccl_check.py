"""tools/pv_nccl_check.py [seconds] — the sharded phase vocoder over RCCL, stand-alone (run under torch.distributed.run, one rank
per GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/pv_nccl_check.py 600`):
shard.pv_pitch_shift_rank_dev (three device stages, two small all-gathers: 12 KiB of phase maps and 30 KiB of overlap-add seams
per rank) against the single call on the same signal, per-rank stage and all-gather times.  bench.py --gpus N carries the same
check as `pv_shard_secondary`; this is the 30-second version for a first multi-GPU box."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import melonix_amd as mx  # noqa: E402
from bench import SR, gen_shard  # noqa: E402
from melonix_amd import shard as sh  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
dev_ord = local if local < torch.cuda.device_count() else 0
torch.cuda.set_device(dev_ord)
dev = torch.device("cuda", dev_ord)
dist.init_process_group("nccl", device_id=dev)
n = int(seconds * SR)
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(dev_ord)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
a = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)


def agree(ok_here):
    fl = torch.tensor([1 if ok_here else 0], device=dev)
    dist.all_reduce(fl, op=dist.ReduceOp.MIN)
    return bool(fl.item())


sh.pv_pitch_shift_rank_dev(ctx, a, 3.0, dist, rank, world, want_f32=False, agree=agree)  # (arena built, RCCL warm)
tm = {}
lo, hi, _, i16 = sh.pv_pitch_shift_rank_dev(ctx, a, 3.0, dist, rank, world, want_f32=False, timings=tm, agree=agree)
whole = torch.empty(n, dtype=torch.int16, device=dev)
ctx.pv_pitch_shift_dev(a, 3.0, None, whole.data_ptr())
torch.cuda.synchronize()
ok = bool(torch.equal(whole[lo:hi], i16))
print(f"rank {rank}/{world}: outputs [{lo},{hi}) equal to the single-call slice: {ok}; chunks {tm['chunks']}, arena {tm['arena_bytes'] / 1e9:.2f} GB; ms: stage 1 "
      f"{tm['stage1_s'] * 1e3:.2f}, all-gather {tm['gather_maps_s'] * 1e3:.3f}, stage 2 {tm['stage2_s'] * 1e3:.2f}, all-gather {tm['gather_seams_s'] * 1e3:.3f}, "
      f"stage 3 {tm['stage3_s'] * 1e3:.2f}", flush=True)
okt = torch.tensor([1 if ok else 0], device=dev)
dist.all_reduce(okt, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if bool(okt.item()) else 1)
