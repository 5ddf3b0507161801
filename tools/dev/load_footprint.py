"""sv_load_weights_f32 of the product library against the test-only superset (which still packs and uploads the round-1 Winograd and
split-bf16 weight images, as the round-2 product did): time per call and device memory held per context."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sudoku_vision_amd as sva
from sudoku_vision_amd.synth import random_state_dict
sd = random_state_dict(1234)
for lib in (sva._native.lib(), sva._native.lib_xcheck()):          # first contexts: code objects, HIP module load
    w = sva.Context(library=lib); w.load_state_dict(sd); w.close()
for name, lib in (("product", sva._native.lib()), ("xcheck (superset, = round-2 product's weight images)", sva._native.lib_xcheck())):
    torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    c = sva.Context(library=lib)
    c.load_state_dict(sd)
    t = time.perf_counter()
    for _ in range(5): c.load_state_dict(sd)
    dt = (time.perf_counter() - t) / 5
    torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]
    print(f"{name}: sv_load_weights_f32 {dt*1e3:.1f} ms per call; device memory held by the context's weight images {(free0-free1)/1e6:.2f} MB")
    c.close()
