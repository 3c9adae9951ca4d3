"""N>1 path on the CPU: world_size-2 and -3 gloo jobs exercising row sharding, the single
all_gather of the framebuffer shards and the in-place row assembly."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,row_tile", [(2, 1), (2, 4), (3, 8)])
def test_gloo_gather_assembles_the_frame(world, row_tile, tor, oracle):
    port = 29500 + (os.getpid() % 2000) + world * 7 + row_tile
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(row_tile)]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count(" ok") == world


def test_shard_plan_matches_c_abi(tor):
    import importlib
    tdist = importlib.import_module("trace-of-radiance_amd.distributed")
    for nrows, tile, world in [(1080, 8, 8), (216, 1, 2), (45, 16, 4), (7, 3, 5), (2160, 16, 8)]:
        plan = tdist.ShardPlan(nrows, tile, world)
        seen = np.concatenate([plan.rows_of(k) for k in range(world)])
        assert sorted(seen.tolist()) == list(range(nrows))
        for k in range(world):
            assert plan.rows_of(k).tolist() == tor.shard_rows(nrows, tile, k, world).tolist()


def test_gather_mapping_inverts_shard_rows(tor):
    """The de-interleave step of the multi-GPU assembly (gather_rows_kernel, csrc/tor_kernels.hip) uses
    row -> (shard = tile mod N, local row = (tile div N) * row_tile + row mod row_tile) with tile = row div row_tile.
    It must be the inverse of tor_shard_rows for ragged splits too (row counts N does not divide, partial last tiles)."""
    import numpy as np
    for nrows in (2, 37, 216, 1080):
        for n in (1, 2, 3, 5, 8, 16):
            for tile in (1, 4, 7, 64):
                owner = np.full(nrows, -1)
                local = np.full(nrows, -1)
                for k in range(n):
                    rows = tor.shard_rows(nrows, tile, k, n)
                    assert np.all(np.diff(rows) > 0)
                    owner[rows] = k
                    local[rows] = np.arange(len(rows))
                r = np.arange(nrows)
                t = r // tile
                assert np.array_equal(owner, t % n) and np.array_equal(local, (t // n) * tile + r % tile), (nrows, n, tile)
