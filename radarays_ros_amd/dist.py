"""Multi-GPU: azimuth columns are independent (RadarCPU.cpp:155 runs them under
`#pragma omp parallel for`), so a frame shards over ranks by contiguous azimuth
blocks; ONE RCCL all-gather over xGMI per frame assembles the columns (no
reduction: blocks are disjoint).  The reference has no distributed code at all
(SURVEY.md §5); this is the MI355X design of BASELINE.json:north_star.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).  torch
is plumbing here: device buffers, streams, the collective.

Frames are pipelined over a few slots (own HIP stream + own column/image
buffers): the all-gather of frame k runs on RCCL's stream while the kernels of
frame k+1 already execute -- the collective is latency-, not bandwidth-bound
(1.37 MB per frame over 7 xGMI links), so hiding it is what matters.
"""
import torch
import torch.distributed as dist


def partition(n_angles, world, rank):
    """Contiguous azimuth block [begin, end) of `rank`; blocks differ by at most 1."""
    base, rem = divmod(n_angles, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_columns(cols_block, n_angles, world, group=None, out=None):
    """All-gather the per-rank column blocks [n_local][n_cells] into [n_angles][n_cells].
    Equal blocks use one all_gather_into_tensor; ragged blocks pad to the largest."""
    n_cells = cols_block.shape[1]
    sizes = [partition(n_angles, world, r) for r in range(world)]
    n_max = max(e - b for b, e in sizes)
    if all(e - b == n_max for b, e in sizes):
        if out is None:
            out = torch.empty((n_angles, n_cells), dtype=cols_block.dtype, device=cols_block.device)
        dist.all_gather_into_tensor(out, cols_block.contiguous(), group=group)
        return out
    pad = torch.zeros((n_max, n_cells), dtype=cols_block.dtype, device=cols_block.device)
    pad[:cols_block.shape[0]] = cols_block
    buf = torch.empty((world * n_max, n_cells), dtype=cols_block.dtype, device=cols_block.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * n_max:r * n_max + (e - b)] for r, (b, e) in enumerate(sizes)], 0)


class _Slot:
    def __init__(self, n_local, n_cells, n_angles, device):
        self.stream = torch.cuda.Stream(device=device)
        self.block = torch.zeros((n_local, n_cells), dtype=torch.uint8, device=device)
        self.cols = torch.zeros((n_angles, n_cells), dtype=torch.uint8, device=device)
        self.image = torch.zeros((n_cells, n_angles), dtype=torch.uint8, device=device)
        self.done = torch.cuda.Event()


class AzimuthShard:
    """Frame loop of one rank: simulate my azimuth block, all-gather, assemble mono8."""

    def __init__(self, ctx, n_cells, n_angles, rank, world, device, n_slots=3, force_slots=False):
        self.ctx, self.n_cells, self.n_angles = ctx, n_cells, n_angles
        self.rank, self.world, self.device = rank, world, device
        self.begin, self.end = partition(n_angles, world, rank)
        self.k = 0
        self.sharded = world > 1 or force_slots      # force_slots: exercise the N>1 path on one rank
        if not self.sharded:
            self.image = torch.zeros((n_cells, n_angles), dtype=torch.uint8, device=device)
        else:
            self.slots = [_Slot(self.end - self.begin, n_cells, n_angles, device) for _ in range(n_slots)]

    def frame(self, pose, stream=None):
        """Enqueue one frame and return the HBM tensor that will hold its mono8 image.

        world == 1: ordered on `stream` (default: current stream).
        world  > 1: asynchronous producer -- the frame runs on one of `n_slots` slot
        streams and NEVER waits on the caller's stream (that is what lets frame k+1
        overlap the all-gather of frame k).  Call `wait(stream)` before consuming;
        an image stays valid until `n_slots - 1` further frames have been enqueued."""
        stream = stream or torch.cuda.current_stream()
        if not self.sharded:
            # pipelining across frames happens inside the library (frame lanes)
            self.ctx.simulate_device(pose, self.image.data_ptr(), stream.cuda_stream)
            return self.image
        s = self.slots[self.k % len(self.slots)]
        self.k += 1
        with torch.cuda.stream(s.stream):
            sp = s.stream.cuda_stream
            self.ctx.simulate_columns_device(pose, self.begin, self.end, s.block.data_ptr(), None, sp)
            cols = gather_columns(s.block, self.n_angles, self.world, out=s.cols)
            self.ctx.assemble_image_device(cols.data_ptr(), s.image.data_ptr(), sp)
            s.done.record(s.stream)
        self.last = s
        return s.image

    def wait(self, stream=None):
        """Make `stream` wait for the most recently enqueued frame (world > 1)."""
        if self.sharded and getattr(self, "last", None) is not None:
            (stream or torch.cuda.current_stream()).wait_event(self.last.done)

    def close(self):
        self.image = None
        self.slots = None
