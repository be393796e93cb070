"""Port-free rendezvous for the multi-process tests.

A test must never hand a closed ephemeral port to children it spawns seconds later: between the close and rank 0's
listen() anything may take the port -- including a peer rank's own retrying connect() to 127.0.0.1:<port>, which on
loopback can self-connect and occupy it (GPUTEST_r03: EADDRINUSE).  torch.distributed's FileStore needs no port: the
parent names a file that does not exist yet, every rank opens it.  gloo's own pair sockets are bound by gloo to port 0
and stay open, so there is no check-then-use window there either.
"""
import os
import tempfile


def new():
    """A rendezvous for one process group: the path of a file that does not exist yet, in a directory of its own."""
    return os.path.join(tempfile.mkdtemp(prefix="fdd_rdzv_"), "store")


def init(backend, rank, world, rdzv, **kw):
    """init_process_group on the file store at `rdzv` (a path from new()); MASTER_ADDR / MASTER_PORT are not read."""
    import torch.distributed as dist

    dist.init_process_group(backend, init_method="file://" + rdzv, rank=rank, world_size=world, **kw)
    return dist


def init_gloo(rank, world, rdzv):
    return init("gloo", rank, world, rdzv)


def done(rdzv):
    """Remove the store (the parent calls this after the workers have joined)."""
    try:
        os.remove(rdzv)
    except OSError:
        pass
    try:
        os.rmdir(os.path.dirname(rdzv))
    except OSError:
        pass
