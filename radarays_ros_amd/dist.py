"""Multi-GPU: azimuth columns are independent (RadarCPU.cpp:155 runs them under
`#pragma omp parallel for`), so a frame shards over ranks by contiguous azimuth
blocks; ONE RCCL all-gather over xGMI per frame assembles the columns (no
reduction: blocks are disjoint).  The reference has no distributed code at all
(SURVEY.md §5); this is the MI355X design of BASELINE.json:north_star.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).  torch
is plumbing here: device buffers, the stream, the collective.
"""
import torch
import torch.distributed as dist


def partition(n_angles, world, rank):
    """Contiguous azimuth block [begin, end) of `rank`; blocks differ by at most 1."""
    base, rem = divmod(n_angles, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_columns(cols_block, n_angles, world, group=None):
    """All-gather the per-rank column blocks [n_local][n_cells] into [n_angles][n_cells].
    Equal blocks use one all_gather_into_tensor; ragged blocks pad to the largest."""
    n_cells = cols_block.shape[1]
    sizes = [partition(n_angles, world, r) for r in range(world)]
    n_max = max(e - b for b, e in sizes)
    if all(e - b == n_max for b, e in sizes):
        out = torch.empty((n_angles, n_cells), dtype=cols_block.dtype, device=cols_block.device)
        dist.all_gather_into_tensor(out, cols_block.contiguous(), group=group)
        return out
    pad = torch.zeros((n_max, n_cells), dtype=cols_block.dtype, device=cols_block.device)
    pad[:cols_block.shape[0]] = cols_block
    buf = torch.empty((world * n_max, n_cells), dtype=cols_block.dtype, device=cols_block.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * n_max:r * n_max + (e - b)] for r, (b, e) in enumerate(sizes)], 0)


class AzimuthShard:
    """Frame loop of one rank: simulate my azimuth block, all-gather, assemble mono8."""

    def __init__(self, ctx, n_cells, n_angles, rank, world, device):
        self.ctx, self.n_cells, self.n_angles = ctx, n_cells, n_angles
        self.rank, self.world, self.device = rank, world, device
        self.begin, self.end = partition(n_angles, world, rank)
        self.image = torch.zeros((n_cells, n_angles), dtype=torch.uint8, device=device)
        if world > 1:
            self.block = torch.zeros((self.end - self.begin, n_cells), dtype=torch.uint8, device=device)
            self.equal = n_angles % world == 0
            self.cols = torch.zeros((n_angles, n_cells), dtype=torch.uint8, device=device)

    def frame(self, pose, stream=None):
        """Enqueue one frame on `stream` (torch stream); the image lands in self.image."""
        stream = stream or torch.cuda.current_stream()
        sp = stream.cuda_stream
        if self.world == 1:
            self.ctx.simulate_device(pose, self.image.data_ptr(), sp)
            return self.image
        self.ctx.simulate_columns_device(pose, self.begin, self.end, self.block.data_ptr(), None, sp)
        with torch.cuda.stream(stream):
            if self.equal:
                dist.all_gather_into_tensor(self.cols, self.block)
                cols = self.cols
            else:
                cols = gather_columns(self.block, self.n_angles, self.world)
        self.ctx.assemble_image_device(cols.data_ptr(), self.image.data_ptr(), sp)
        return self.image

    def close(self):
        self.image = None
