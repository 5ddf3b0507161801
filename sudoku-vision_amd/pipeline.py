"""End-to-end frame -> digits pipeline with the host corner search in the loop -- the build's counterpart
of the hot section of pipeline/run.py:244-312 (call order, glue, output conventions), batched and
software-pipelined for a resident pool of frames:

    GPU  K1 (binary) -> despeckle  --D2H, pinned-->  CPU contour corner search (threads)  --Minv, H2D-->  GPU  K2 -> K3

Chunks of frames are double-buffered: while the host searches chunk i, the GPU thresholds chunk i+1 and
classifies chunk i-1.  `glue` selects what sits between extract_cells and the model (see include/sudoku_vision_hip.h sv_glue).
A frame whose grid is not found gets found=False and digits 0 (the reference
returns "Grid detection failed" for it, pipeline/run.py:268-272); so does a frame whose four corners do not define a
homography (order_points, cv/grid.py:79-91, returns one point twice for a quad rotated near 45 degrees; the reference
warps garbage for that image) -- one such frame never affects the others of its batch."""
import os
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

import numpy as np
import torch

from . import host
from .runtime import Context


def host_cpu_budget():
    """CPUs this process may actually use: the cgroup CPU quota when there is one (a box gives each GPU job 16 of the host's
    256 logical CPUs through cpu.max; running more runnable threads than the quota gets the whole group throttled for the rest
    of the period -- 30 ms stalls in the corner search), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    return n


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(device):
    """CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function) that this process may run on, or None when
    the platform does not say.  The pipeline's pinned buffers are first touched, hence placed, there; host threads that wander to the
    other socket of a two-socket box run the search 10-20 % slower and with 40-ms outliers (tools/dev/e2e_stall_trace.py)."""
    try:
        p = torch.cuda.get_device_properties(device)
        path = f"/sys/bus/pci/devices/{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0/local_cpulist"
        cpus = _parse_cpulist(open(path).read())
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None
    if hasattr(os, "sched_getaffinity"):
        cpus &= os.sched_getaffinity(0)
    return cpus or None


class FramePipeline:
    def __init__(self, ctx: Context, H: int, W: int, chunk: int = 32, host_threads=None, min_area_ratio=0.1, glue=0, despeckle=True, sparse=True, depth=5, cpu_affinity="auto", bits_direct=True):
        self.ctx, self.H, self.W, self.chunk = ctx, H, W, chunk
        # chunks in flight.  A chunk's chain is K1 -> D2H -> search -> K2/K3, and chunk i's K1 is only issued once chunk i-depth+1's search has
        # returned: with too few in flight the period is (K1 + D2H + search) / (depth - 1), not the slowest stage
        self.depth = depth = max(2, int(depth))
        self.host_threads = host_threads or max(1, host_cpu_budget() - 2)   # two CPUs left for this thread and the runtime's own
        self.min_area_ratio = min_area_ratio
        # the speck filter is exact only while nothing it can erase (bounding box <= 62x62 px, area <= 61*61) reaches the
        # search's area floor (include/sudoku_vision_hip.h, sv_despeckle_u8): small frames go through unfiltered
        self.despeckle = bool(despeckle) and min_area_ratio * H * W > 61 * 61
        self.glue = glue            # Context.GLUE_NORMALIZE, or GLUE_RUNPY for run.py's preprocess_cell (CLAHE + threshold)
        dev = ctx.device
        self.s_pre = torch.cuda.Stream(dev)       # K1 (+ despeckle)
        self.s_d2h = torch.cuda.Stream(dev)       # the D2H copy on its own stream: in s_pre it would hold back the next chunk's K1 for 0.3 ms
        self.s_cls = torch.cuda.Stream(dev)       # H2D + K2 + K3
        # D2H payload: with despeckle on and W % 32 == 0 the binary crosses PCIe as 1 bit per pixel (W/32 words per row)
        self.packed = self.despeckle and W % 32 == 0
        self.bits_direct = bool(bits_direct)        # K1 emits the bit image itself (sv_preprocess_bits_u8); False: byte image + packing despeckle
        # sparse=True (default): ... as sparse records (row masks + the non-zero words) with room for half of the words; a frame that does not
        # fit is fetched dense (sv_pack_sparse_bits).  2-3x fewer PCIe bytes on the synthetic feed for one small kernel; PCIe itself is not the
        # bound (the dense copy runs at 55 GB/s on its own stream) but the host side is: the search threads then read 138 KB of records per
        # frame instead of 259 KB of freshly DMA-written memory, and with K1 writing bits the pipeline is balanced enough for that to show
        # (128-frame chunks: 132 k frames/s sparse, 113 k dense; tools/e2e_breakdown.py)
        self.sparse = self.packed and bool(sparse) and H * ((W // 32 + 63) // 64) <= 16000
        if self.packed:
            self.dev_bits = [torch.empty((chunk, H, W // 32), dtype=torch.int32, device=dev) for _ in range(depth)]
        if self.sparse:
            # room for a third of the words (a synthetic 1080p frame keeps 13-15 k of its 64.8 k after the filter); an int: capacity in words
            # (tests force the fallback with it).  Smaller records = fewer D2H bytes: 192 k frames/s against 187 k with room for half (tools/dev/e2e_knobs.py)
            cap = H * (W // 32) // 3 if sparse is True else int(sparse)
            self.rec_bytes = host.sparse_bits_record_bytes(H, W, cap)
            self.pinned = [torch.empty((chunk, self.rec_bytes), dtype=torch.uint8).pin_memory() for _ in range(depth)]
            self.dev_rec = [torch.empty((chunk, self.rec_bytes), dtype=torch.uint8, device=dev) for _ in range(depth)]
            self.s_side = torch.cuda.Stream(dev)
        elif self.packed:
            self.pinned = [torch.empty((chunk, H, W // 32), dtype=torch.int32).pin_memory() for _ in range(depth)]
        else:
            self.pinned = [torch.empty((chunk, H, W), dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.dev_bin = None                       # byte images, only when K1 cannot write bits directly (allocated on first use)
        self.minv_pin = [torch.empty((chunk, 9), dtype=torch.float64).pin_memory() for _ in range(depth)]
        self.minv_dev = [torch.empty((chunk, 9), dtype=torch.float64, device=dev) for _ in range(depth)]
        # host threads next to the GPU: "auto" = the GPU's NUMA node, None = leave them alone, or an explicit set of CPUs.  Applies to the
        # search thread and the library's workers, not to the caller's thread
        self.cpus = gpu_local_cpus(dev) if cpu_affinity == "auto" else (set(cpu_affinity) if cpu_affinity else None)
        if self.cpus:
            host.set_pool_affinity(self.cpus)
        self.pool = ThreadPoolExecutor(1, initializer=(lambda: os.sched_setaffinity(0, self.cpus)) if self.cpus and hasattr(os, "sched_setaffinity") else None)
        self.dense_fallbacks = 0
        ctx.reserve(chunk * 81)

    def _search(self, slot, m, ev):
        ev.synchronize()
        if self.sparse:
            corners, found = host.find_grid_corners_sparse_batch(self.pinned[slot][:m].numpy(), self.H, self.W, self.min_area_ratio, 0.02, self.host_threads)
            over = np.nonzero(found == 2)[0]
            if over.size:                       # records that overflowed: fetch those frames dense (the slot's bit image is still there)
                with torch.cuda.stream(self.s_side):
                    dense = self.dev_bits[slot][torch.from_numpy(over).to(self.ctx.device)].cpu().numpy()
                c2, f2 = host.find_grid_corners_bits_batch(dense, self.H, self.W, self.min_area_ratio, 0.02, self.host_threads)
                corners[over], found[over] = c2, f2
                self.dense_fallbacks += int(over.size)
            found = found.astype(bool)
        elif self.packed:
            corners, found = host.find_grid_corners_bits_batch(self.pinned[slot][:m].numpy(), self.H, self.W, self.min_area_ratio, 0.02, self.host_threads)
        else:
            corners, found = host.find_grid_corners_batch(self.pinned[slot][:m].numpy(), self.min_area_ratio, 0.02, self.host_threads)
        minv, ok = Context.corners_to_minv_batch(corners.astype(np.float32))    # not-found frames hold zeros: degenerate, identity, masked
        self.minv_pin[slot][:m] = torch.from_numpy(minv.reshape(m, 9))
        return corners, found & ok

    def describe(self):
        d2h = (f"pinned D2H of sparse records of the bit-packed binary (row masks + non-zero words, {self.rec_bytes // 1000} KB/frame over PCIe, "
               f"dense fallback {self.H * self.W // 8 // 1000} KB)" if self.sparse
               else f"pinned D2H of the bit-packed binary ({self.H * self.W // 8 // 1000} KB/frame over PCIe)" if self.packed
               else f"pinned D2H of the binary ({self.H * self.W // 1000} KB/frame over PCIe)")
        k1 = ("K1 (bit image) -> despeckle in place (exact speck filter) -> " if self.packed and self.bits_direct
              else f"K1 -> {'despeckle (exact speck filter) -> ' if self.despeckle else ''}")
        return (f"{k1}{d2h} -> C++ contour corner search on "
                f"{self.host_threads} host threads -> K2 -> K3, {self.chunk}-frame chunks, {self.depth} in flight"
                + (f", host threads on the GPU's NUMA node ({len(self.cpus)} CPUs)" if self.cpus else ""))

    def run(self, frames, out=None, repeat=1, total=None):
        """frames u8 [n,H,W,3] on the context's device -> dict(digits u8[n,81], logits f32[n,81,10], conf f32[n,81],
        corners int32[n,4,2] (host), found bool[n] (host)).  repeat > 1 streams the pool that many times through the
        pipeline without draining it in between (steady-state throughput measurement); total = k streams exactly k frames,
        cycling the pool (the k-th frame is pool frame k mod n: BASELINE configs[3]'s shard of 100,000 frames); both need chunk | n."""
        n = frames.shape[0]
        if (repeat > 1 or total is not None) and n % self.chunk:
            raise ValueError("repeat / total need the chunk size to divide the number of frames")
        dev = self.ctx.device
        if out is None:
            out = {"logits": torch.empty((n, 81, 10), dtype=torch.float32, device=dev),
                   "digits": torch.empty((n, 81), dtype=torch.uint8, device=dev),
                   "conf": torch.empty((n, 81), dtype=torch.float32, device=dev)}
        corners_all = np.zeros((n, 4, 2), np.int32)
        found_all = np.zeros(n, bool)
        cur = torch.cuda.current_stream(dev)
        self.s_pre.wait_stream(cur)
        self.s_d2h.wait_stream(cur)
        self.s_cls.wait_stream(cur)
        # sv_preprocess_bits_u8's layout requirements (otherwise K1 writes bytes and the despeckle packs them)
        bits_direct = (self.packed and self.bits_direct and frames.is_contiguous() and frames.data_ptr() % 4 == 0 and (3 * self.W) % 4 == 0
                       and self.H >= 16 and self.W >= 16)
        if not bits_direct and self.dev_bin is None:
            self.dev_bin = [torch.empty((self.chunk, self.H, self.W), dtype=torch.uint8, device=dev) for _ in range(self.depth)]
        if total is None:
            starts = [(s0, min(self.chunk, n - s0)) for _ in range(repeat) for s0 in range(0, n, self.chunk)]
        else:
            starts = [((k * self.chunk) % n, min(self.chunk, total - k * self.chunk)) for k in range((total + self.chunk - 1) // self.chunk)]
        pending = []                                  # (future, slot, start, m)
        free_ev = [None] * self.depth                 # classification done with slot's buffers

        def classify(item):
            fut, slot, s, m = item
            corners, found = fut.result()
            corners_all[s:s + m], found_all[s:s + m] = corners, found
            with torch.cuda.stream(self.s_cls):
                self.minv_dev[slot][:m].copy_(self.minv_pin[slot][:m], non_blocking=True)
                sub = {k: out[k][s:s + m] for k in ("logits", "digits", "conf")}
                self.ctx.frames_to_digits(frames[s:s + m], self.minv_dev[slot][:m], out=sub, glue=self.glue)
                if not found.all():
                    out["digits"][s:s + m][torch.from_numpy(~found).to(dev)] = 0
                ev = torch.cuda.Event(blocking=True)      # the waiting thread sleeps instead of spinning: the box's CPU quota is for the search
                ev.record(self.s_cls)
                free_ev[slot] = ev

        for i, (s, m) in enumerate(starts):
            slot = i % self.depth
            if free_ev[slot] is not None:
                free_ev[slot].synchronize()
            with torch.cuda.stream(self.s_pre):
                # every buffer of a chunk belongs to its slot: an allocation in here (a 130-MB hipMalloc while the caching allocator's pool
                # grows) stalls the whole pipeline for tens of milliseconds
                if bits_direct:
                    # K1 writes the bit image itself and the speck filter works on it in place: no byte image at all
                    b = self.ctx.despeckle_bits(self.ctx.preprocess_bits(frames[s:s + m], out=self.dev_bits[slot][:m]))
                else:
                    b = self.ctx.preprocess(frames[s:s + m], out=self.dev_bin[slot][:m])
                # exact accelerator for the host search: erase the specks that cannot matter (csrc/k4_despeckle.hip), in place
                if self.packed:
                    if not bits_direct:
                        b = self.ctx.despeckle(b, out=b, packed=self.dev_bits[slot][:m])
                    if self.sparse:
                        b = self.ctx.pack_sparse_bits(b, self.dev_rec[slot])
                elif self.despeckle:
                    b = self.ctx.despeckle(b, out=b)
                ready = torch.cuda.Event()
                ready.record(self.s_pre)
            with torch.cuda.stream(self.s_d2h):
                self.s_d2h.wait_event(ready)
                self.pinned[slot][:m].copy_(b, non_blocking=True)
                ev = torch.cuda.Event(blocking=True)      # the waiting thread sleeps instead of spinning: the box's CPU quota is for the search
                ev.record(self.s_d2h)
            pending.append((self.pool.submit(self._search, slot, m, ev), slot, s, m))
            if len(pending) > self.depth - 2:
                classify(pending.pop(0))
        while pending:
            classify(pending.pop(0))
        cur.wait_stream(self.s_cls)
        out["corners"], out["found"] = corners_all, found_all
        return out


def recognize_image(image, model_state_dict=None, ctx=None, glue=Context.GLUE_RUNPY, top_k=0):
    """One BGR image (numpy uint8 [H,W,3], or a CUDA uint8 tensor of that shape) -> dict(grid 9x9 list, digits, confidences, corners) or None when no
    grid is found -- the call order of pipeline/run.py:261-312, preprocess_cell (:73-95) included by default.
    top_k > 1 adds run_v2's per-cell `alternatives` (pipeline/run_v2.py:165-178): 81 lists of (digit, prob), best excluded."""
    from .runtime import default_context
    ctx = ctx or default_context()
    if model_state_dict is not None:
        ctx.load_state_dict(model_state_dict)
    if isinstance(image, torch.Tensor):             # already in HBM (imgcodecs.imread(..., device=True))
        frames = image.contiguous()[None]
    else:
        frames = torch.from_numpy(np.ascontiguousarray(image)).to(ctx.device)[None]
    binary = ctx.preprocess(frames)[0].cpu().numpy()
    corners = host.find_grid_corners(binary)
    if corners is None:
        return None
    minv = ctx.minv_to_device(Context.corners_to_minv(corners[None].astype(np.float32)))
    out = ctx.frames_to_digits(frames, minv, glue=glue)
    digits = out["digits"][0].cpu().numpy()
    res = {"grid": [[int(digits[r * 9 + c]) for c in range(9)] for r in range(9)], "digits": digits,
           "confidence": out["conf"][0].cpu().numpy(), "logits": out["logits"][0].cpu().numpy(), "corners": corners}
    if top_k > 1:
        idx, prob = ctx.softmax_topk(out["logits"][0], top_k)
        idx, prob = idx.cpu().numpy(), prob.cpu().numpy()
        res["alternatives"] = [[(int(idx[i, j]), float(prob[i, j])) for j in range(1, top_k)] for i in range(81)]
    return res


def run_solver(grid):
    """The reference's run_solver (pipeline/run.py:163-202) without the subprocess: -> (success, solution) with
    solution == grid when the puzzle is invalid or has no solution."""
    code, sol = host.solve_sudoku(grid)
    g = [[int(v) for v in row] for row in np.asarray(grid).reshape(9, 9)]
    return (True, [[int(v) for v in row] for row in sol]) if code == 1 else (False, g)


# ---- the harness's own entry point (pipeline/run.py:36-70, 205-241, 244-355), same result fields and error strings ------------------
@dataclass
class CellPrediction:
    """pipeline/run.py:36-43"""
    row: int
    col: int
    digit: int                      # 0 = empty, 1-9 = digit
    confidence: float
    is_original: bool = True        # True if read from the image, False if filled in by the solver


@dataclass
class PipelineResult:
    """pipeline/run.py:46-66 (same fields; times in seconds)"""
    success: bool
    error: str = None
    time_cv: float = 0.0
    time_ml: float = 0.0
    time_solver: float = 0.0
    time_total: float = 0.0
    original_image: np.ndarray = None
    warped_grid: np.ndarray = None
    cells: list = field(default_factory=list)
    predictions: list = field(default_factory=list)
    recognized_grid: list = field(default_factory=list)
    solution: list = field(default_factory=list)
    low_confidence_cells: list = field(default_factory=list)
    constraint_violations: list = field(default_factory=list)


def check_constraints(grid):
    """The messages of pipeline/run.py:205-241 for a 9x9 grid: one entry per repeated value in a row, a column or a box
    (0 = empty is ignored); a value seen three times in a unit yields two entries, each naming the previous occurrence."""
    out = []
    units = [("row", r, [(r, c) for c in range(9)]) for r in range(9)]
    units += [("col", c, [(r, c) for r in range(9)]) for c in range(9)]
    units += [("box", b, [(3 * (b // 3) + i, 3 * (b % 3) + j) for i in range(3) for j in range(3)]) for b in range(9)]
    for kind, idx, members in units:
        last = {}
        for r, c in members:
            v = grid[r][c]
            if v <= 0:
                continue
            if v in last:
                pr, pc = last[v]
                if kind == "row":
                    out.append(f"Row {idx + 1}: duplicate {v} at columns {pc + 1} and {c + 1}")
                elif kind == "col":
                    out.append(f"Column {idx + 1}: duplicate {v} at rows {pr + 1} and {r + 1}")
                else:
                    out.append(f"Box ({idx // 3 + 1},{idx % 3 + 1}): duplicate {v}")
            last[v] = (r, c)
    return out


def run_pipeline(image_path, state_dict=None, ctx=None, debug=False, low_confidence=0.7):
    """run_pipeline(image_path) of pipeline/run.py:244-355 on the MI355X path: JPEG -> frame in HBM (imgcodecs) -> K1 -> host
    corner search -> K2 (warped 450x450 grid kept, as the reference keeps it) -> preprocess_cell + DigitCNN (K3) -> constraint
    check -> in-process solver.  Same PipelineResult fields, same error strings, same partial results on failure."""
    from . import imgcodecs
    from .runtime import default_context
    res = PipelineResult(success=False)
    t_total = time.time()
    ctx = ctx or default_context()
    if state_dict is not None:
        ctx.load_state_dict(state_dict)
    frame = imgcodecs.imread(image_path, device=True, ctx=ctx)
    if frame is None:
        res.error = f"Failed to load image: {image_path}"
        return res
    res.original_image = frame.cpu().numpy()

    t_cv = time.time()
    try:
        binary = ctx.preprocess(frame[None])[0].cpu().numpy()
    except Exception as e:                                   # noqa: BLE001 -- the reference reports, it does not raise
        res.error = f"Preprocessing failed: {e}"
        return res
    try:
        corners = host.find_grid_corners(binary)
        if corners is None:
            res.error = "Grid detection failed: no quadrilateral found"
            return res
    except Exception as e:                                   # noqa: BLE001
        res.error = f"Grid detection failed: {e}"
        return res
    try:
        minv = ctx.minv_to_device(Context.corners_to_minv(corners[None].astype(np.float32)))
        warped = ctx.warp_perspective(frame, minv[0], 450)
        res.warped_grid = warped.cpu().numpy()
    except Exception as e:                                   # noqa: BLE001
        res.error = f"Perspective warp failed: {e}"
        return res
    try:
        cells = ctx.extract_cells(warped, 28, 5, 5)          # cv/extract.py:13-56 defaults: 50-px cells, 10 % margin
        res.cells = list(cells.cpu().numpy())
        if len(res.cells) != 81:
            res.error = f"Cell extraction failed: expected 81 cells, got {len(res.cells)}"
            return res
    except Exception as e:                                   # noqa: BLE001
        res.error = f"Cell extraction failed: {e}"
        return res
    res.time_cv = time.time() - t_cv

    t_ml = time.time()
    try:
        logits, digits, conf = ctx.cnn_forward(cells, want_digits=True, glue=Context.GLUE_RUNPY)
        digits, conf = digits.cpu().numpy(), conf.cpu().numpy()
        res.predictions = [CellPrediction(row=i // 9, col=i % 9, digit=int(digits[i]), confidence=float(conf[i])) for i in range(81)]
        grid = [[int(digits[r * 9 + c]) for c in range(9)] for r in range(9)]
        res.recognized_grid = grid
        res.low_confidence_cells = [(p.row, p.col, p.confidence) for p in res.predictions if p.digit > 0 and p.confidence < low_confidence]
    except Exception as e:                                   # noqa: BLE001
        res.error = f"ML inference failed: {e}"
        return res
    res.time_ml = time.time() - t_ml

    res.constraint_violations = check_constraints(grid)
    if res.constraint_violations and debug:
        print(f"Warning: {len(res.constraint_violations)} constraint violations detected")
        for v in res.constraint_violations[:5]:
            print(f"  - {v}")

    t_solver = time.time()
    try:
        ok, solution = run_solver(grid)
        if ok:
            res.solution = solution
            for p in res.predictions:
                if p.digit == 0:
                    p.digit = solution[p.row][p.col]
                    p.is_original = False
        else:
            res.error = "Solver failed: puzzle may be invalid or have recognition errors"
            res.solution = grid                              # the reference still returns the partial result
    except Exception as e:                                   # noqa: BLE001
        res.error = f"Solver error: {e}"
        return res
    res.time_solver = time.time() - t_solver
    res.time_total = time.time() - t_total
    res.success = ok
    return res
