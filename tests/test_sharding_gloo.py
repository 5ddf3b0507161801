"""CPU, world_size 2 over gloo: the N>1 path of bench.py / the sharded pipeline (frame -> rank mapping, timing
barrier, MAX of elapsed, result gather).  Each rank runs the CPU oracle on its shard as the stand-in compute."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    import sudoku_vision_amd as sva
    from sudoku_vision_amd import sharding
    from sudoku_vision_amd.synth import synth_frames
    import cnn_oracle
    import sv_oracle as o
    r, lr, w = sharding.init(backend="gloo")
    assert (r, w) == (rank, world)
    frames, corners, _ = synth_frames(n_total, 135, 240, seed=5)        # every rank can regenerate the pool; it owns a shard
    mine = sharding.shard_indices(n_total, rank, world)
    sd = cnn_oracle.random_state_dict(3)
    sharding.barrier()
    t0 = time.perf_counter()
    digs = []
    for i in mine:
        cells = o.warp_cells(frames[i].numpy(), corners[i])
        digs.append(cnn_oracle.predict(sd, o.cells_to_input(cells)[:, None])[1])
    time.sleep(0.05 * (rank + 1))                                       # ranks finish at different times
    sharding.barrier()
    elapsed = time.perf_counter() - t0
    mx = sharding.max_over_ranks(elapsed)
    local = torch.stack(digs) if digs else torch.zeros((0, 81), dtype=torch.uint8)
    full = sharding.gather_digits(local, n_total, rank, world)
    q.put((rank, mine, elapsed, mx, full.numpy()))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_sharding():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    world, n_total = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]                # round-robin, disjoint, complete
    assert res[0][3] == res[1][3] >= max(res[0][2], res[1][2]) - 1e-9    # every rank reports the max over ranks
    assert (res[0][4] == res[1][4]).all()                                # gathered digits identical on both ranks
    # and equal to the single-process result
    import cnn_oracle
    import sv_oracle as o
    from sudoku_vision_amd.synth import synth_frames
    frames, corners, _ = synth_frames(n_total, 135, 240, seed=5)
    sd = cnn_oracle.random_state_dict(3)
    for i in range(n_total):
        d = cnn_oracle.predict(sd, o.cells_to_input(o.warp_cells(frames[i].numpy(), corners[i]))[:, None])[1].numpy()
        assert (res[0][4][i] == d).all()


def test_launch_local_ranks_world2(tmp_path):
    """bench.py --gpus N without a launcher starts its own ranks through sharding.launch_local_ranks: the same call here, world 2,
    CPU workers (tests/_rank_worker.py follows bench.py's entry sequence).  Rank 0's line carries n_gpus = 2 and the digits of
    the unsharded computation; a rank that dies takes the job down with its exit code instead of leaving the other at a barrier."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from sudoku_vision_amd import sharding
    from sudoku_vision_amd.synth import random_state_dict, synth_frames
    import cnn_oracle
    import sv_oracle as o
    worker = os.path.join(ROOT, "tests", "_rank_worker.py")
    out = tmp_path / "rank0.json"
    with open(out, "wb") as f:
        assert sharding.launch_local_ranks(2, [worker, "5"], rank0_stdout=f) == 0
    res = json.loads(out.read_text().strip().splitlines()[-1])
    assert res["n_gpus"] == 2 and res["elapsed"] >= 0.1                  # the slower rank sleeps 0.1 s: MAX over ranks
    assert [r["rank"] for r in res["ranks"]] == [0, 1] and [r["local_rank"] for r in res["ranks"]] == [0, 1]    # gather_objects: rank order
    assert len({r["pid"] for r in res["ranks"]}) == 2
    frames, corners, _ = synth_frames(5, 135, 240, seed=5)
    sd = random_state_dict(3)
    for i in range(5):
        d = cnn_oracle.predict(sd, o.cells_to_input(o.warp_cells(frames[i].numpy(), corners[i]))[:, None])[1].numpy()
        assert (np.array(res["digits"][i]) == d).all()
    with open(out, "wb") as f:
        assert sharding.launch_local_ranks(2, [worker, "5", "1"], rank0_stdout=f) == 7


def test_bench_refuses_world_size_mismatch():
    """`--gpus 8` under a WORLD_SIZE=2 launcher is an error, not a silent 2-GPU (or 1-GPU) benchmark."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_synth_state_dict_is_the_oracles():
    """bench.py takes its random-init weights from the package (no oracle import outside cpu_baseline): same values as the
    generator the CNN goldens were made with."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from sudoku_vision_amd.synth import random_state_dict
    import cnn_oracle
    a, b = random_state_dict(1234), cnn_oracle.random_state_dict(1234)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_bench_scripts_touch_the_oracle_only_in_cpu_baseline():
    import ast
    for name in ("bench.py", "bench_latency.py"):
        tree = ast.parse(open(os.path.join(ROOT, name)).read())
        for node in tree.body:                                           # top-level functions (cpu_baseline's helpers are nested in it)
            if isinstance(node, ast.FunctionDef):
                src = ast.get_source_segment(open(os.path.join(ROOT, name)).read(), node)
                if "oracle" in src:
                    assert node.name in ("cpu_baseline", "cpu_baseline_test_image"), f"{name}:{node.name} mentions the oracle"
        top = [n for n in tree.body if not isinstance(n, (ast.FunctionDef, ast.Expr))]
        for n in top:
            assert "oracle" not in ast.get_source_segment(open(os.path.join(ROOT, name)).read(), n)


def test_shard_indices_properties():
    from sudoku_vision_amd.sharding import shard_indices
    for n in (0, 1, 7, 100000):
        for w in (1, 2, 4, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            allidx = sorted(i for p in parts for i in p)
            assert allidx == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
