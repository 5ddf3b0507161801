"""One rank of the CPU rehearsal of bench.py's multi-rank path (started by sharding.launch_local_ranks from
tests/test_sharding_gloo.py): same entry sequence as bench.py -- env_rank_world, sharding.init (gloo), shard_indices, barrier,
timed loop, barrier, max_over_ranks, rank 0 prints one JSON line -- with the CPU oracle standing in for the HIP kernels."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    n_total = int(sys.argv[1])
    fail_rank = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    import torch
    from sudoku_vision_amd import sharding
    from sudoku_vision_amd.synth import random_state_dict, synth_frames
    import cnn_oracle
    import sv_oracle as o
    rank, local_rank, world = sharding.env_rank_world()
    if rank == fail_rank:
        raise SystemExit(7)
    sharding.init()
    frames, corners, _ = synth_frames(n_total, 135, 240, seed=5)
    mine = sharding.shard_indices(n_total, rank, world)
    sd = random_state_dict(3)
    sharding.barrier()
    t0 = time.perf_counter()
    digs = [cnn_oracle.predict(sd, o.cells_to_input(o.warp_cells(frames[i].numpy(), corners[i]))[:, None])[1] for i in mine]
    time.sleep(0.05 * (rank + 1))
    sharding.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
    local = torch.stack(digs) if digs else torch.zeros((0, 81), dtype=torch.uint8)
    full = sharding.gather_digits(local, n_total, rank, world)
    ranks = sharding.gather_objects({"rank": rank, "local_rank": local_rank, "pid": os.getpid()})    # what bench.py's `ranks` field is made of
    if rank == 0:
        print(json.dumps({"n_gpus": world, "elapsed": elapsed, "digits": full.numpy().tolist(), "ranks": ranks}), flush=True)
    sharding.shutdown()


if __name__ == "__main__":
    main()
