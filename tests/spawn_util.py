"""mp.spawn of the ranks of a multi-process test on a FRESH rendezvous port, retried when the rendezvous itself fails (a port picked by
``bind(0)`` and released can be taken before the ranks bind it; a stale TIME_WAIT socket refuses the connection): a test of the data-parallel
program must not fail on the host's port table.  Failures of the workers themselves (assertions, device errors) are not retried."""
import socket

RENDEZVOUS_ERRORS = ("Address already in use", "EADDRINUSE", "Connection refused", "Connection reset", "Connection closed", "connect() timed out",
                     "Socket Timeout", "failed to connect", "Broken pipe", "client socket has timed out", "server socket has failed")


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(fn, nprocs: int, before=(), after=(), attempts: int = 3) -> int:
    """``mp.spawn(fn, args=(*before, port, *after), nprocs=nprocs, join=True)`` -> the port that worked."""
    import torch.multiprocessing as mp
    for attempt in range(attempts):
        port = free_port()
        try:
            mp.spawn(fn, args=(*before, port, *after), nprocs=nprocs, join=True)
            return port
        except Exception as e:   # ProcessRaisedException carries the worker's traceback as text
            if attempt + 1 < attempts and any(t in str(e) for t in RENDEZVOUS_ERRORS):
                continue
            raise
