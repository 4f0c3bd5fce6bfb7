"""Frame-level sharding of the rcn back-end across GPUs: pictures are the independent
units (the reference's `--framethr`, ovdec.c:188-248), the only cross-GPU edge is a finished picture
becoming a reference picture of pictures decoded elsewhere.  One process per GPU; the exchange is a
point-to-point send/receive pair per picture over `torch.distributed` (RCCL on GPUs, gloo in the CPU
tests) -- there is no collective on the data path."""
from __future__ import annotations


def frame_owner(decode_order_idx: int, world: int) -> int:
    """Picture k in decode order is decoded by rank k mod world."""
    return decode_order_idx % world


def ring_exchange(dist, send_t, recv_t, rank: int, world: int):
    """Push this rank's reconstructed picture to rank+1 (which lists it as a reference picture of its next
    picture) and receive rank-1's into `recv_t`.  Blocks until both complete.  No-op for world == 1."""
    if world <= 1:
        return
    ops = [dist.P2POp(dist.isend, send_t, (rank + 1) % world),
           dist.P2POp(dist.irecv, recv_t, (rank - 1) % world)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
