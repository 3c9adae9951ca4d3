"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-N gloo job on the CPU.
Each rank renders ITS rows of a small frame with the CPU oracle (standing in for the GPU
kernel, which needs a device), then runs the product's gather/assemble path
(trace-of-radiance_amd/distributed.py) and checks the assembled frame against the full render."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    row_tile = int(sys.argv[1])
    dist.init_process_group(backend="gloo")
    os.environ["TOR_NO_TORCH"] = "0"
    tdist = importlib.import_module("trace-of-radiance_amd.distributed")
    tor = importlib.import_module("trace-of-radiance_amd")
    from oracle import oracle as O
    H, W, spp = 22, 32, 4
    objs, _ = O.random_scene(0xFACADE)
    cam = O.camera()
    plan = tdist.ShardPlan(H, row_tile, world)
    # the C ABI's shard map and the Python plan must agree
    assert list(plan.rows_of(rank)) == list(tor.shard_rows(H, row_tile, rank, world))
    frame = tdist.DistributedFrame(plan, W, rank, torch.device("cpu"))
    full = O.render(H, W, spp, cam, objs, seeding=O.SEED_SAMPLE, math=O.MATH_PORTABLE, accum=O.ACCUM_QUANTIZED).pixels
    mine = np.stack([O.render(H, W, spp, cam, objs, seeding=O.SEED_SAMPLE, math=O.MATH_PORTABLE,
                              accum=O.ACCUM_QUANTIZED, rows=(int(r), int(r) + 1)).pixels[int(r)]
                     for r in plan.rows_of(rank)])
    frame.shard.copy_(torch.from_numpy(mine))
    out = frame.gather().numpy()
    assert np.array_equal(out, full), f"rank {rank}: assembled frame differs"
    # every rank holds the same frame
    t = torch.from_numpy(out.copy())
    ref = t.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(t, ref)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
