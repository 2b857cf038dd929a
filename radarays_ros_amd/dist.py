"""Multi-GPU: azimuth columns are independent (RadarCPU.cpp:155 runs them under
`#pragma omp parallel for`), so frames shard over ranks by contiguous azimuth
blocks and ONE RCCL collective over xGMI per step assembles them (no reduction:
blocks are disjoint).  The reference has no distributed code at all;
this is the MI355X design of BASELINE.json:north_star.

Default (weak scaling): a step renders `world` frames.  Rank r simulates ITS azimuth
block of all `world` frames in one set of launches (rr_simulate_batch_columns_device:
world x 400/world = 400 segments, the same launch shape as one whole frame on one
GPU), then the per-frame gathers "columns of frame f -> rank f" are fused into a
single `all_to_all_single`; rank r assembles frame r.  Per-GPU work per step is
constant, each rank receives one frame (1.37 MB) per step.

`strong=True`: one frame per step sharded over all ranks + one all-gather (every
rank gets the frame) -- the latency mode for heavy frames (config 4).

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).  torch is
plumbing here: device buffers, streams, the collective.  Steps are pipelined over a
few slots (own HIP stream + buffers) so the collective of step k overlaps the
kernels of step k+1.

`host_out=True`: the reference's simulate() ends with the image in HOST memory
(m_polar_image, RadarCPU.cpp:542,555-561), so every rank also delivers the frames it
assembled to page-locked host memory.  A copy behind each step costs ~7 % of the frame
rate, so the images a slot assembled ride out on the later-pass trace
launches of the slot's NEXT step (rr_simulate_batch_columns_carry_device, same stream:
ordered behind the assemble that wrote them); `flush_host()` sends what is still waiting.
"""
import contextlib

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
    def __init__(self, n_frames, n_images, n_local, n_cells, n_angles, device):
        # (a CPU device is accepted only so that the step logic can be driven by a mock context in the
        # world-size-2 gloo tests; the product always runs on cuda devices)
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self.block = torch.zeros((n_frames, n_local, n_cells), dtype=torch.uint8, device=device)
        # weak: [source rank][fpr][n_local][cells] (same size as block); strong: the gathered [n_angles][cells]
        self.recv = torch.zeros((max(n_frames * n_local, n_angles), n_cells), dtype=torch.uint8, device=device)
        self.images = torch.zeros((n_images, n_cells, n_angles), dtype=torch.uint8, device=device)
        self.done = torch.cuda.Event() if device.type == "cuda" else None
        # host_out: where this slot's images end up, whether a set is still waiting in `images`, and the step it belongs to
        self.host = None
        self.host_pending = False
        self.step_no = -1
        self.host_step_no = -1


class AzimuthShard:
    """Step loop of one rank (see module docstring).

    A step always runs on one of `n_slots` slot streams (own buffers) and NEVER waits on the
    caller's stream -- that is what lets step k+1 overlap the tail (and, for world > 1, the
    collective) of step k.  `wait(stream)` orders a consumer after the last step; an image
    stays valid until `n_slots - 1` further steps have been enqueued.

    world == 1: a step renders `frames_per_rank` whole frames in one set of launches.
    world  > 1: weak mode renders `world * frames_per_rank` frames per step (rank r ends up
                with frames r*fpr .. r*fpr+fpr-1); strong mode renders one frame per step
                (every rank ends up with it)."""

    def __init__(self, ctx, n_cells, n_angles, rank, world, device, n_slots=4, strong=False,
                 frames_per_rank=1, force_collective=False, host_out=False):
        self.ctx, self.n_cells, self.n_angles = ctx, n_cells, n_angles
        self.rank, self.world, self.device = rank, world, device
        self.begin, self.end = partition(n_angles, world, rank)
        self.n_loc = self.end - self.begin
        self.k = 0
        self.collective = world > 1 or force_collective   # force_collective: run the N>1 code on one rank
        self.strong = self.collective and (strong or n_angles % world != 0)
        self.fpr = 1 if self.strong else int(frames_per_rank)
        self.frames_per_step = 1 if self.strong else self.fpr * world
        if self.frames_per_step > 64:      # RR_MAX_BATCH
            raise ValueError("at most 64 frames per step")
        self.slots = [_Slot(self.frames_per_step, self.fpr, self.n_loc, n_cells, n_angles, device)
                      for _ in range(n_slots)]
        self.last = None
        self.host_out = bool(host_out)
        if self.host_out:
            for s in self.slots:
                s.host = torch.zeros((self.fpr, n_cells, n_angles), dtype=torch.uint8, pin_memory=(device.type == "cuda"))

    def frame(self, pose, stream=None):
        """Single-frame step (frames_per_step == 1)."""
        return self.step([pose], stream)

    def step(self, poses, stream=None, done_event=None):
        """Enqueue one step (`frames_per_step` poses, the same list on every rank); returns the
        HBM tensor [fpr][n_cells][n_angles] that will hold this rank's mono8 image(s).
        `done_event` (a torch.cuda.Event) is recorded on the slot's stream behind the step."""
        assert len(poses) == self.frames_per_step
        s = self.slots[self.k % len(self.slots)]
        self.k += 1
        C, nl = self.n_cells, self.n_loc
        with (torch.cuda.stream(s.stream) if s.stream is not None else contextlib.nullcontext()):
            sp = s.stream.cuda_stream if s.stream is not None else None
            if self.strong:
                self.ctx.simulate_columns_device(poses[0], self.begin, self.end, s.block.data_ptr(), None, sp)
                cols = gather_columns(s.block[0], self.n_angles, self.world, out=s.recv[:self.n_angles])
                self.ctx.assemble_image_device(cols.data_ptr(), s.images[0].data_ptr(), sp)
            else:
                if self.host_out and s.host_pending:
                    # the images this slot assembled n_slots steps ago leave on this step's trace launches
                    self.ctx.simulate_batch_columns_carry_device(poses, self.begin, self.end, s.block.data_ptr(), sp,
                                                                 s.images.data_ptr(), s.host.data_ptr(), s.images.numel())
                    s.host_pending = False
                    s.host_step_no = s.step_no
                else:
                    self.ctx.simulate_batch_columns_device(poses, self.begin, self.end, s.block.data_ptr(), sp)
                if self.collective:
                    # frames d*fpr .. d*fpr+fpr-1 go to rank d; I receive [source rank][fpr][n_loc][C],
                    # source-rank order == azimuth order
                    dist.all_to_all_single(s.recv[:self.frames_per_step * nl].view(-1), s.block.view(-1))
                    src = s.recv
                else:
                    src = s.block
                # all fpr frames of this rank in one launch
                self.ctx.assemble_frames_device(src.data_ptr(), nl, self.fpr * nl * C, self.fpr, nl * C,
                                                s.images.data_ptr(), sp)
            s.step_no = self.k - 1
            if self.host_out:
                if self.strong:          # latency mode: one frame, delivered at once
                    self._deliver(s)
                    s.host_step_no = s.step_no
                else:
                    s.host_pending = True
            if s.done is not None:
                s.done.record(s.stream)
            if done_event is not None:
                if s.stream is not None:
                    done_event.record(s.stream)
                else:
                    done_event.record()
        self.last = s
        return s.images

    def _deliver(self, s):
        """slot.images -> slot.host behind the slot's stream, over the SDMA engines (rr_deliver_to_host_async: the delivered rate
        then does not depend on which engine the process' HIP runtime would pick for a hipMemcpyAsync); a context without it
        (the mock of the CPU tests) gets torch's copy"""
        if s.stream is not None and hasattr(self.ctx, "deliver_to_host_async"):
            self.ctx.deliver_to_host_async(s.images.data_ptr(), s.host.data_ptr(), s.images.numel(), s.stream.cuda_stream)
            self._delivered = True          # fenced by ctx.wait_host, not by the stream (the copy runs on the SDMA engines)
        else:
            s.host.copy_(s.images, non_blocking=True)

    def flush_host(self):
        """host_out: deliver the images that are still waiting on their slots (copies on the slots' streams) and wait
        for every delivery.  Afterwards slot.host holds the images of step slot.host_step_no for every slot."""
        if not self.host_out:
            return
        for s in self.slots:
            if s.host_pending:
                with (torch.cuda.stream(s.stream) if s.stream is not None else contextlib.nullcontext()):
                    self._deliver(s)
                s.host_pending = False
                s.host_step_no = s.step_no
        for s in self.slots:
            if s.stream is not None:
                s.stream.synchronize()
        if getattr(self, "_delivered", False):
            self.ctx.wait_host(None)
            self._delivered = False

    def host_images(self, step_no):
        """The host tensor [fpr][n_cells][n_angles] holding this rank's images of step `step_no`, or None when that step has
        not been delivered (yet) or its slot has been handed to a later step.  Valid after flush_host(), or once the step
        n_slots later on the same slot has completed."""
        for s in self.slots:
            if s.host is not None and s.host_step_no == step_no:
                return s.host
        return None

    def wait(self, stream=None):
        """Make `stream` wait for the most recently enqueued step."""
        if self.last is not None and self.last.done is not None:
            (stream or torch.cuda.current_stream()).wait_event(self.last.done)

    def close(self):
        self.slots = None
        self.last = None
