"""Launch tape: one iteration's enqueue calls recorded once and replayed from a flat list.

An eager iteration costs the host about 8 ms (slicing views, argument checks, Python wrappers around ~700 calls),
close to its 7 ms of GPU time.  While `tape` is a list, the ctypes entry points of pgan_hip append (function, resolved
arguments) -- raw device pointers, sizes, the stream handle; the stream fork / join edges are library calls too
(pgan_stream_edge on events that exist beforehand) and land on the list the same way -- and the few steps of an
iteration that are not library calls (collectives of a data-parallel run) are appended through call().

compile() turns the list into its replay form: every run of consecutive library calls becomes one native tape
(pgan_tape_create / _append in the library), replayed by a single pgan_tape_replay call that walks a C array -- no
interpreter between the launches; the non-library steps stay Python callables between those runs.  A single-GPU
iteration is one native tape.  Valid as long as the buffers the pointers refer to stay allocated (they are preallocated
by Engine / FusedTrainer) and the sequence of an iteration does not change.
"""
from __future__ import annotations

tape = None      # list of (callable, args, returns_status) while recording, else None
error_type = RuntimeError      # pgan_hip sets this to PganError: what replay() raises for a failed library call
native_tape = None             # pgan_hip sets this to its NativeTape class


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
    """Replay of an uncompiled list, call by call from Python.  Library entry points return a status (0 = enqueued);
    anything else stops the replay with an error, as the same call would have while recording."""
    for fn, args, status in entries:
        rc = fn(*args)
        if status and rc:
            raise error_type(f"{getattr(fn, '__name__', fn)} failed with code {rc} during tape replay")


class CompiledTape:
    """Replay form of a recorded list: native tapes (runs of library calls) and the Python steps between them."""

    def __init__(self, entries):
        self.entries = entries
        self.segments = []          # NativeTape | (fn, args)
        run = []

        def close():
            if run:
                self.segments.append(native_tape(list(run)))
                run.clear()
        for ent in entries:
            fn, args, status = ent
            if status and native_tape is not None and native_tape.accepts(fn):
                run.append(ent)
            else:
                close()
                self.segments.append((fn, args, status))
        close()

    @property
    def native_calls(self) -> int:
        return sum(len(s) for s in self.segments if not isinstance(s, tuple))

    @property
    def python_steps(self) -> int:
        return sum(1 for s in self.segments if isinstance(s, tuple))

    def replay(self) -> None:
        for seg in self.segments:
            if isinstance(seg, tuple):
                fn, args, status = seg
                rc = fn(*args)
                if status and rc:
                    raise error_type(f"{getattr(fn, '__name__', fn)} failed with code {rc} during tape replay")
            else:
                seg.replay()


def compile(entries) -> CompiledTape:      # noqa: A001 (the name says what it does to a tape)
    return CompiledTape(entries)
