"""Launch tape: one iteration's enqueue calls recorded once and replayed from a flat list.

An eager iteration costs the host about 8 ms (slicing views, argument checks, Python wrappers around ~550 launches),
close to its 9 ms of GPU time.  While `tape` is a list, the ctypes entry points of pgan_hip append (function, resolved
arguments) — raw device pointers, sizes, the stream handle — and the few non-library steps of an iteration (stream
fork/join, collectives) are appended through call().  Replaying the list repeats exactly those enqueues with none of
the Python in between.  Valid as long as the buffers the pointers refer to stay allocated (they are preallocated by
Engine / FusedTrainer) and the sequence of an iteration does not change.
"""
from __future__ import annotations

tape = None      # list of (callable, args, returns_status) while recording, else None
error_type = RuntimeError      # pgan_hip sets this to PganError: what replay() raises for a failed library call


def call(fn, *args, on_current_stream: bool = False):
    """Run fn(*args) now; while recording also note it for replay.  on_current_stream: at replay, run it with
    torch's current stream set to the one that is current now (for code that enqueues on "the current stream",
    e.g. torch.distributed collectives)."""
    global tape
    t = tape
    if t is None:
        return fn(*args)
    if on_current_stream:
        import torch
        t.append((_under, (torch.cuda.current_stream(), fn, args), False))
    else:
        t.append((fn, args, False))
    tape = None                      # whatever fn enqueues itself is part of this one entry
    try:
        return fn(*args)
    finally:
        tape = t


def _under(stream, fn, args):
    import torch
    with torch.cuda.stream(stream):
        return fn(*args)


def replay(entries) -> None:
    """Library entry points return a status (0 = enqueued); anything else stops the replay with an error, as the same
    call would have while recording."""
    for fn, args, status in entries:
        rc = fn(*args)
        if status and rc:
            raise error_type(f"{getattr(fn, '__name__', fn)} failed with code {rc} during tape replay")
